R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "not c5" > $O/pytest_r3.log 2>&1; tail -3 $O/pytest_r3.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<PY
import json
d=json.load(open("$O/bench_driver.json"))
print("driver-like:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_per_launch"], d.get("ms_per_step_with_depth"), d["single_frame"]["ms_per_launch"], d.get("verified_against_cpu"), d["cpu_baseline"]["sample"][-40:])
PY
tail -2 $O/bench_driver.err
for x in torch native; do timeout 300 python bench.py --force-gather --exchange $x --no-cpu-baseline --verify > $O/bench_gather_$x.json 2> $O/bench_gather_$x.err; python -c "
import json; d=json.load(open('$O/bench_gather_$x.json')); print('force-gather $x', d['ms_per_step'], d['host_enqueue_ms_per_step'])"; tail -1 $O/bench_gather_$x.err; done
