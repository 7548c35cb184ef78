run() { python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-depth-block --extras off --min-seconds 1 --frames-per-launch $1 --batch-streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('frames per launch $1, launches in flight $2: %.4f  %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
for fs in "6 4" "4 6" "5 5" "6 5" "6 6" "3 8" "12 2" "5 4" "7 4" "6 3" "4 8" "6 4"; do run $fs; done
