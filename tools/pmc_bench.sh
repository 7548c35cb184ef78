#!/bin/bash
# usage: tools/pmc_bench.sh <tag> "<counters>" [extra bench args]   (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift; ctr=$1; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
python3 - "$R/gpurun_out/pmc_$tag" <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_raymarch' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in sorted(acc.items()): print('%-28s n=%d avg=%.4g'%(c,len(v),sum(v)/len(v)))
PY
