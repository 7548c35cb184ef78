#!/bin/bash
# usage (GPU box, repo root): tools/ab_brick.sh <tag>  - VERDICT r3 'Next #2': dense vs 4x4x4-bricked distance map in the integrator (lab variants 21 / 40),
# probes-only variants 41 / 42 under the L2 counters, per-iteration stamps 26 / 43
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout 900 python tools/lab/run_lab.py c3 0,21,40,41,42,21,40 > $O/lab_c3.txt 2>&1; grep -E "variant|differs" $O/lab_c3.txt
timeout 900 python tools/lab/run_lab.py c4 0,21,40 > $O/lab_c4.txt 2>&1; grep -E "variant|differs" $O/lab_c4.txt
for v in 26 43; do
  LAB_STAMP_VARIANT=$v timeout 600 python tools/lab/run_lab.py c3 0 --lean-stamps 2>&1 | grep '"view"' > $O/stamps_v$v.jsonl
done
cd /tmp && export TMPDIR=/tmp
for v in 41 42 21 40; do
  for ctr in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum"; do
    tag=v${v}_$(echo $ctr | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc/$tag -- python3 $R/tools/lab/run_lab.py c3 $v > /dev/null 2> $O/pmc_$tag.err
    python3 - "$O/pmc/$tag" "variant$v" <<'PY' >> $O/pmc_summary.txt
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_lab_lean' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in sorted(acc.items()): print('%-10s %-30s n=%d avg=%.6g'%(sys.argv[2],c,len(v),sum(v)/len(v)))
PY
  done
done
rm -rf $O/pmc
cat $O/pmc_summary.txt
