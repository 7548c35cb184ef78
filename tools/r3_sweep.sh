R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O; cd $R
timeout 1500 python tools/benchmark_sweep.py --out $O --frames 100 --frames-in-flight 8 > $O/sweep8.log 2>&1; tail -2 $O/sweep8.log
timeout 1500 python tools/benchmark_sweep.py --out $O --frames 100 --frames-in-flight 1 > $O/sweep1.log 2>&1; tail -2 $O/sweep1.log
ls $O
