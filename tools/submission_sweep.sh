#!/bin/bash
# What a renderer with few frames in flight gets, and the headline's neighbourhood: bench.py with F frames per launch on S streams (C3).
# usage: tools/submission_sweep.sh > profiles/rN_submission_sweep.txt
run() { python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-depth-block --extras off --min-seconds 1 --frames-per-launch $1 --batch-streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('frames per launch $1, launches in flight $2: %.4f  %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
echo "# bench.py --steps 24 --warmup 8 --frames-per-launch F --batch-streams S (C3, MI355X, $(git -C . log -1 --format=%h 2>/dev/null || echo this tree)): ms per frame, roofline.frac"
for f in 1 2 3 4 6 8; do for s in 1 2 3 4; do run $f $s; done; done
