R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "divisions or render" > $O/pytest_div.log 2>&1; tail -3 $O/pytest_div.log
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print('bench', d['ms_per_step'], d['roofline']['frac'], d['single_frame']['ms_per_launch'], d.get('ms_per_step_with_depth'))"; done
