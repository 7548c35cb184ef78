#!/bin/bash
# Same-box A/B of the bench headline: alternating runs of bench.py with two builds of the library (and, optionally, environment switches).
# usage: tools/ab_bench.sh <tag> <rounds> "<name>=<env assignments...>" ...      e.g.
#   tools/ab_bench.sh r5_clampfree 3 "r4=VKV_LIB_PATH=tools/lab/ab/libvkvolume_amd_r4.so" "new=" "new_clamp=VKV_RAYMARCH_CLAMP=always"
# Writes gpurun_out/<tag>/<name>_<round>.json (the bench line) and a summary table gpurun_out/<tag>/summary.txt.
set -u
tag=$1; rounds=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
BENCH_ARGS=${BENCH_ARGS:---steps 20 --warmup 5}
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    env $envs python bench.py $BENCH_ARGS > $out/${name}_$r.json 2> $out/${name}_$r.err || echo "FAILED $name $r" >> $out/summary.txt
  done
done
python - "$out" <<'PY' >> $out/summary.txt
import glob, json, os, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    name, r = os.path.basename(f)[:-5].rsplit("_", 1)
    try:
        line = [l for l in open(f) if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:
        print("unreadable", f, e); continue
    rows[name].append((int(r), d))
print("%-14s %5s %10s %8s %12s %12s" % ("build", "round", "ms/frame", "frac", "single ms", "verified"))
for name, lst in rows.items():
    for r, d in sorted(lst):
        sf = d.get("single_frame", {})
        print("%-14s %5d %10.4f %8.4f %12s %12s" % (name, r, d["ms_per_step"], d["roofline"]["frac"], sf.get("ms_per_launch", sf.get("ms", "")), d.get("verified_against_cpu", "")))
    ms = sorted(x[1]["ms_per_step"] for x in lst)
    print("%-14s median %8.4f" % (name, ms[len(ms) // 2]))
PY
cat $out/summary.txt
