R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "render or feedback or batch" > $O/pytest_ro.log 2>&1; tail -3 $O/pytest_ro.log
for v in 1 0; do export VKV_RAYMARCH_RAY_ORDER=$v; for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v$i.json 2> $O/bench_$v$i.err; python -c "
import json; d=json.load(open('$O/bench_$v$i.json')); print('ray_order $v bench', d['ms_per_step'], d['roofline']['frac'], d['single_frame']['ms_per_launch'], d.get('ms_per_step_with_depth'))"; done; done
