#!/bin/bash
# usage (on the GPU box, from the repo root):  tools/collect_profiles.sh <tag> [workload]
# rocprofv3 evidence for profiles/: kernel-trace stats of bench.py in its submission modes (batch, one frame per launch, streams, dense
# sampling), FETCH_SIZE / WRITE_SIZE in PMC passes of their own (gpurun refuses traces + counters in one run), and the bench lines.
# Everything lands in gpurun_out/<tag>/; tools/summarize_profile.py condenses it into <tag>_rocprof.txt.
tag=$1; wl=${2:-c3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $wl --steps 64 --warmup 16 --no-cpu-baseline --min-seconds 0.5"
run_stats() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -- $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run_pmc() { name=$1; ctr=$2; shift; shift; timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${name}_$ctr -- $B --steps 16 --warmup 2 --min-seconds 0.1 "$@" > /dev/null 2> $O/pmc_${name}_$ctr.err; }
run_stats batch8
run_stats single --frames-per-launch 1
run_stats streams3 --submit streams
run_stats dense_single --skip none --no-ert --frames-per-launch 1 --steps 16 --warmup 4
run_pmc batch8 FETCH_SIZE
run_pmc batch8 WRITE_SIZE
run_pmc single FETCH_SIZE --frames-per-launch 1
run_pmc single WRITE_SIZE --frames-per-launch 1
cd $R
{
  for m in batch8 single streams3 dense_single; do
    python3 tools/summarize_profile.py "${tag}_$m" $O/stats_$m
    echo "bench line ($m):"; cat $O/bench_$m.json; echo
  done
  python3 tools/summarize_profile.py "${tag}_pmc_batch8" /nonexistent $O/pmc_batch8_FETCH_SIZE $O/pmc_batch8_WRITE_SIZE
  python3 tools/summarize_profile.py "${tag}_pmc_single" /nonexistent $O/pmc_single_FETCH_SIZE $O/pmc_single_WRITE_SIZE
  echo "# commit: $(cat $R/.commit_id 2>/dev/null)"
} > $O/${tag}_rocprof.txt
ls -la $O | head -30
