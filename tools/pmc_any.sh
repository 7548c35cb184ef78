#!/bin/bash
# usage: tools/pmc_any.sh <tag> "<counters>" <kernel-substring> <python script> [args]   (run on the GPU box from the repo root)
# One rocprofv3 --pmc pass (counters only, no traces) of an arbitrary script; prints per-kernel averages of the counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift; ctr=$1; shift; pat=$1; shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
python3 - "$R/gpurun_out/pmc_$tag" "$pat" <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:60],r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(acc.items()): print('%-60s %-26s n=%d avg=%.4g'%(k,c,len(v),sum(v)/len(v)))
PY
