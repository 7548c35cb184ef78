#!/bin/bash
# Check (GPU, round 6): the render tests of the GPU suite once per A/B switch of the tuning block set through the environment - every switch must
# leave every frame bit-identical (rectangle schedules, fill_outside and the computed start order included).
T="tests/test_gpu_render_batch.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_exchange.py tests/test_gpu_c5_virtual_ranks.py"
for e in "VKV_RAYMARCH_TILE_ORDER=linear" "VKV_RAYMARCH_FEEDBACK=0" "VKV_RAYMARCH_BATCH=pull" "VKV_RAYMARCH_SCHEDULER=persistent" "VKV_RAYMARCH_BATCH_ORDER=sequential" \
         "VKV_RAYMARCH_LUT=0" "VKV_RAYMARCH_CULL=0" "VKV_RAYMARCH_CLAMP=always" "VKV_RAYMARCH_WAVE_SHAPE=4" "VKV_RAYMARCH_WAVE_SHAPE=16" "VKV_RAYMARCH_FEEDBACK_PERIOD=1"; do
  printf "%-40s " "$e"; env $e python -m pytest $T -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -1
done
