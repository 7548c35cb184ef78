R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "pull or tuning or feedback" > $O/pytest_rg.log 2>&1; tail -3 $O/pytest_rg.log
for v in region tiles; do export VKV_RAYMARCH_BATCH=$v; for a in "" "--batch-streams 1"; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $a > $O/bench_$v.json 2> $O/bench_$v.err; python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('batch=$v $a:', d['ms_per_step'], d['roofline']['frac'], d['single_frame']['ms_per_launch'], d.get('ms_per_step_with_depth'))"; done; done
