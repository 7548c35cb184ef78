#!/bin/bash
# Experiment (GPU, round 6): launches of few frames on one stream (what a renderer with a swap chain submits) - whole-image schedule against the
# screen rectangle + fill_outside, start-order feedback on / off.  usage: tools/lab/few_frames.sh
run() { f=$1; s=$2; shift 2; python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-depth-block --extras off --min-seconds 1 --frames-per-launch $f --batch-streams $s "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('F $f S $s %-32s %.4f  %.3f' % ('$*', d['ms_per_step'], d['roofline']['frac']))"; }
for rep in 1 2; do
for fs in "1 1" "2 1" "3 1" "4 1" "3 2"; do
  for opt in "--tile-rect on" "--tile-rect off" "--tile-rect on --no-feedback" "--tile-rect off --no-feedback"; do run $fs $opt; done
done; done
