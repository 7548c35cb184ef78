for s in 3 4 5; do for f in 4 5 7 8 10; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-depth-block --extras off --min-seconds 1.5 --frames-per-launch $f --batch-streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('steps 20: frames per launch $f, streams $s: %.4f  %.4f' % (d['ms_per_step'], d['roofline']['frac']))"; done; done
