R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "not c5" > $O/pytest_l0.log 2>&1; tail -3 $O/pytest_l0.log
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; python -c "
import json; d=json.load(open('$O/bench_$i.json')); print('bench', d['ms_per_step'], d['roofline']['frac'], d['single_frame']['ms_per_launch'], d.get('ms_per_step_with_depth'))"; done
timeout 300 python bench.py --skip none --no-ert --frames-per-launch 1 --batch-streams 1 --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_d.json 2> $O/bench_d.err; python -c "
import json; d=json.load(open('$O/bench_d.json')); print('dense single', d['ms_per_step'], d['roofline']['frac'])"
timeout 300 python bench.py --frames-per-launch 1 --batch-streams 1 --no-cpu-baseline > $O/bench_s.json 2> $O/bench_s.err; python -c "
import json; d=json.load(open('$O/bench_s.json')); print('single launches', d['ms_per_step'], d['roofline']['frac'])"
