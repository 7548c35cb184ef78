#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/collect_profiles.sh <tag> [workload]      (r6_d c3 -> profiles/r6_traffic.json; r6_c4 c4 -> profiles/r6_c4_traffic.json)
# rocprofv3 evidence for profiles/: kernel-trace stats of bench.py in its submission modes (batch, one frame per launch, streams, dense
# sampling), FETCH_SIZE / WRITE_SIZE in PMC passes of their own (gpurun refuses traces + counters in one run), and the bench lines.
# Everything lands in gpurun_out/<tag>/; tools/summarize_profile.py condenses it into <tag>_rocprof.txt.
tag=$1; wl=${2:-c3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $wl --steps 64 --warmup 16 --no-cpu-baseline --no-depth-block --extras off --min-seconds 0.5"
run_stats() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -- $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run_pmc() { name=$1; ctr=$2; shift; shift; timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${name}_$ctr -- $B --steps 16 --warmup 2 --min-seconds 0.1 "$@" > /dev/null 2> $O/pmc_${name}_$ctr.err; }
run_stats default
run_stats batch8 --frames-per-launch 8 --batch-streams 1
run_stats single --frames-per-launch 1 --batch-streams 1
run_stats streams3 --submit streams
run_stats dense_single --skip none --no-ert --frames-per-launch 1 --batch-streams 1 --steps 16 --warmup 4
run_pmc batch8 FETCH_SIZE --frames-per-launch 8 --batch-streams 1
run_pmc batch8 WRITE_SIZE --frames-per-launch 8 --batch-streams 1
run_pmc single FETCH_SIZE --frames-per-launch 1
run_pmc single WRITE_SIZE --frames-per-launch 1
# instruction mix of the ray-march kernels (SQ block, one pass): VALU / SALU / LDS / VMEM instructions, wave cycles and their wait shares
run_pmc batch8 SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CU_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_SALU --frames-per-launch 8 --batch-streams 1
run_pmc batch8 SQ_INSTS_VMEM_RD,SQ_INSTS_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_THREAD_CYCLES_VALU,SQ_INSTS_VALU_CVT,SQ_INSTS_VALU_FMA_F32,SQ_INSTS_VALU_INT32 --frames-per-launch 8 --batch-streams 1
# VALU busy of the ray-march kernel (the binding limit next to the HBM fraction): SQ_ACTIVE_INST_VALU * 4 / (SIMDs * GRBM_GUI_ACTIVE)
run_pmc batch8 SQ_ACTIVE_INST_VALU,SQ_BUSY_CYCLES,GRBM_GUI_ACTIVE --frames-per-launch 8 --batch-streams 1
run_stats dense_batch8 --skip none --no-ert --steps 16 --warmup 8 --frames-per-launch 8 --batch-streams 1
# check of valu_active_lanes (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU) on a launch whose lanes all sample at every step: should read close to 64
run_pmc dense8 SQ_THREAD_CYCLES_VALU,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU --skip none --no-ert --steps 8 --warmup 2 --frames-per-launch 8 --batch-streams 1
if [ "$wl" = c3 ]; then
  for w in c2 c3cube c4 c5; do timeout 300 python3 $R/bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; done
  timeout 600 python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err
  timeout 300 python3 $R/tools/time_precompute.py c3 > $O/precompute_times_c3.txt 2>&1
fi
cd $R
{
  for m in default batch8 single streams3 dense_single; do
    python3 tools/summarize_profile.py "${tag}_$m" $O/stats_$m
    echo "bench line ($m):"; cat $O/bench_$m.json; echo
  done
  python3 tools/summarize_profile.py "${tag}_pmc_batch8" /nonexistent $O/pmc_batch8_FETCH_SIZE $O/pmc_batch8_WRITE_SIZE
  python3 tools/summarize_profile.py "${tag}_pmc_single" /nonexistent $O/pmc_single_FETCH_SIZE $O/pmc_single_WRITE_SIZE
  python3 tools/summarize_profile.py "${tag}_pmc_sq_batch8" /nonexistent $O/pmc_batch8_SQ_WAVES* $O/pmc_batch8_SQ_INSTS_VMEM_RD* $O/pmc_batch8_SQ_ACTIVE_INST_VALU*
  python3 tools/summarize_profile.py "${tag}_pmc_dense8" /nonexistent $O/pmc_dense8_SQ_THREAD_CYCLES_VALU*
  python3 tools/summarize_profile.py "${tag}_dense_batch8" $O/stats_dense_batch8
  echo "bench line (dense_batch8):"; cat $O/bench_dense_batch8.json; echo
  echo "# commit: $(cat $R/.commit_id 2>/dev/null)"
} > $O/${tag}_rocprof.txt
python3 tools/make_traffic_json.py $O ${tag%%_*} $wl > $O/${tag}_traffic.json; cat $O/${tag}_traffic.json
ls -la $O | head -30
