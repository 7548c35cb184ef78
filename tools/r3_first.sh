R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3a
cd $R && timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc $?" 
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench_driver.json 2> gpurun_out/r3a/bench_driver.err; tail -c 600 gpurun_out/r3a/bench_driver.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3a/pmc_gather_fetch -- $R/tools/micro/gather_fetch > $R/gpurun_out/r3a/gather_fetch.txt 2>&1
cat $R/gpurun_out/r3a/gather_fetch.txt | tail -8
python3 - $R/gpurun_out/r3a/pmc_gather_fetch <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Kernel_Name'][:50], r['Counter_Name'], r['Counter_Value'])
PY
