#!/bin/bash
# usage (GPU box, repo root): tools/pmc_pass.sh <tag> "<counters>" [bench args]    one rocprofv3 --pmc pass (counters only) over bench.py;
# prints the per-kernel averages of the ray-march kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift; ctr=$1; shift
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc/$tag -- python3 $R/bench.py --steps 16 --warmup 2 --min-seconds 0.1 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/pmc/$tag.err
python3 - "$R/gpurun_out/pmc/$tag" "$tag" <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_raymarch' in r['Kernel_Name']:
            acc[(r['Kernel_Name'].split('(')[0][:44], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(acc.items()): print('%-12s %-44s %-28s n=%d avg=%.5g'%(sys.argv[2],k,c,len(v),sum(v)/len(v)))
PY
