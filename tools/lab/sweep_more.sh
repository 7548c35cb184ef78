#!/bin/bash
# Experiment (GPU, round 6): frames per launch x launches in flight beyond tools/submission_sweep.sh, block lengths that are multiples of the launch size
run() { f=$1; s=$2; st=$3; shift 3; python bench.py --steps $st --warmup $st --no-cpu-baseline --no-depth-block --extras off --min-seconds 1 --frames-per-launch $f --batch-streams $s "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('F $f S $s steps $st %-16s %.4f  %.3f' % ('$*', d['ms_per_step'], d['roofline']['frac']))"; }
for fb in "" "--no-feedback"; do
run 5 4 40 $fb; run 6 4 48 $fb; run 7 4 56 $fb; run 8 4 64 $fb; run 7 3 42 $fb; run 5 3 30 $fb; run 6 3 36 $fb; run 8 3 48 $fb
done
