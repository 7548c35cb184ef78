#!/bin/bash
# Experiment (GPU, round 6): does a coarser tile rectangle (align_tiles) let the start-order feedback follow a moving camera?  3 swap-chain images as
# one launch on one stream / two such launches in flight, orbit of D degrees per frame.
run() { python bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-depth-block --extras off --min-seconds 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-100s %.4f  %.3f' % ('$*', d['ms_per_step'], d['roofline']['frac']))"; }
for fs in "--frames-per-launch 3 --batch-streams 1" "--frames-per-launch 3 --batch-streams 2"; do
  for cam in "--camera moving --camera-step 0.25" "--camera moving --camera-step 1"; do
    for al in 1 4 8; do for fb in "" "--no-feedback"; do run $fs $cam --rect-align $al $fb; done; done
  done
done
