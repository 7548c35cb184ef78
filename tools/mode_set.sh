#!/bin/bash
# usage (GPU box, repo root): tools/mode_set.sh <tag> [workload]
# The mode set of SURVEY.md section 8(d) on one workload: every empty-space-skipping type with and without early ray termination, and the
# intensity-only transfer function.  One bench line per mode into gpurun_out/<tag>/modes.jsonl, a table on stdout.
tag=$1; wl=${2:-c3}
O=gpurun_out/$tag; mkdir -p $O; : > $O/modes.jsonl
run() { name=$1; shift; python bench.py --workload $wl --no-cpu-baseline --no-depth-block --extras off --min-seconds 1.0 "$@" 2> $O/$name.err | grep '^{' | tail -1 > $O/$name.json; cat $O/$name.json >> $O/modes.jsonl; }
for skip in distance anisotropic block none; do
  run ${skip}_ert --skip $skip
  run ${skip}_noert --skip $skip --no-ert
done
run distance_ert_intensity_tf --skip distance --tf intensity
python - $O <<'PY'
import json, sys, os
names = [s + e for s in ("distance", "anisotropic", "block", "none") for e in ("_ert", "_noert")] + ["distance_ert_intensity_tf"]
print("%-28s %10s %8s %14s %14s" % ("mode", "ms/frame", "frac", "G samples/s", "G probes/s"))
for n in names:
    d = json.load(open(os.path.join(sys.argv[1], n + ".json")))
    print("%-28s %10.4f %8.3f %14.1f %14.1f" % (n, d["ms_per_step"], d["roofline"]["frac"], d["volume_samples_per_s"] / 1e9, d["distance_probes_per_s"] / 1e9))
PY
