R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O
cd $R && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py -m gpu -x -q -k "distance or anisotropic or c3_full or c4_full" > $O/pytest_dm.log 2>&1; tail -2 $O/pytest_dm.log
cd /tmp && export TMPDIR=/tmp
for w in c3 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${w}_f -- python3 $R/tools/time_precompute.py $w > $O/pre_${w}_f.txt 2>&1
  echo "== $w"; grep "distance_map" $O/pre_${w}_f.txt
  python3 - $O/stats_${w}_f <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_dm' in r['Name']: print("   %-60s calls %5s avg %8.1f us"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
