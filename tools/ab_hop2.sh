#!/bin/bash
# usage (GPU box, repo root): tools/ab_hop2.sh <tag>  - A/B of VkvTuning.probe_hops = 2: lab variants, the GPU parity suite with it on, three alternating bench runs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 600 python tools/lab/run_lab.py c3 0,30,32 > $O/lab.txt 2>&1; grep variant $O/lab.txt
VKV_RAYMARCH_PROBE_HOPS=2 timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_hop2.log 2>&1; tail -2 $O/pytest_gpu_hop2.log
for i in 1 2 3; do
  for h in 1 2; do
    VKV_RAYMARCH_PROBE_HOPS=$h timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_h${h}_$i.json 2>> $O/bench.err
    python3 -c "
import json,sys
d=json.load(open('$O/bench_h${h}_$i.json')); print('hops $h run $i', d['ms_per_step'], d['roofline']['frac'], 'single', d['single_frame']['ms_per_launch'])"
  done
done
