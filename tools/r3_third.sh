R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O; cd $R
timeout 300 python bench.py --skip none --no-ert --steps 16 --warmup 8 --batch-streams 1 --no-cpu-baseline --min-seconds 1 > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('dense batch8', d['ms_per_step'], d['roofline']['frac'], 'single', d['single_frame']['ms_per_launch'])"
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('c3', d['ms_per_step'], d['roofline']['frac'], 'single', d['single_frame']['ms_per_launch'])"; done
