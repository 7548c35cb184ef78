#!/bin/bash
# usage (on the GPU box, from the repo root): tools/full_check.sh <tag>   - the whole GPU suite, smoke(), and a driver-like bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; python3 - $O/bench_driver_like.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], 'frac', r['frac'], 'per launch', r['frac_per_launch'], 'traffic', r.get('traffic'), 'valu_issue_frac', r.get('valu_issue_frac'), 'wait_frac', r.get('wait_frac'), 'waves_per_simd', r.get('waves_per_simd'), 'depth', d.get('ms_per_step_with_depth'), 'single', d['single_frame']['ms_per_launch'], 'cpu', d['cpu_baseline']['value'], '1-thread', d['cpu_baseline'].get('value_1_thread'), d.get('verified_against_cpu'))
PY
