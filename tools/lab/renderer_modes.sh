#!/bin/bash
# Experiment (GPU, round 6): the submissions a renderer with a swap chain can make - 3 images as one launch on one stream, double buffering (one
# frame per launch on two streams), two launches of 3 in flight - with a camera that orbits by D degrees per frame, start-order feedback on / off.
run() { python bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-depth-block --extras off --min-seconds 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-88s %.4f  %.3f' % ('$*', d['ms_per_step'], d['roofline']['frac']))"; }
for fs in "--frames-per-launch 3 --batch-streams 1" "--frames-per-launch 1 --batch-streams 2" "--frames-per-launch 3 --batch-streams 2"; do
  for cam in "--camera static" "--camera moving --camera-step 0.25" "--camera moving --camera-step 1" "--camera moving --camera-step 3"; do
    for fb in "" "--no-feedback"; do run $fs $cam $fb; done
  done
done
