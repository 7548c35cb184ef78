R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_precompute; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pre -- python3 $R/tools/time_precompute.py ${1:-c3} > $O/precompute_${1:-c3}.txt 2>&1
cat $O/precompute_${1:-c3}.txt | tail -14
python3 - $O/stats_pre <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:40]:
        print("%-90s calls %5s avg %10.1f us  total %8.2f ms"%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
