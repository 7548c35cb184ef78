cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_l; mkdir -p $O
for i in 1 2 3; do for f in 0 1; do
  VKV_RAYMARCH_FORMAT_ROWS=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b_f${f}_$i.json 2>> $O/err.txt
  python3 -c "
import json; d=json.load(open('$O/b_f${f}_$i.json')); print('format_rows $f run $i', d['ms_per_step'], d['roofline']['frac'], 'single', d['single_frame']['ms_per_launch'], 'depth', d.get('ms_per_step_with_depth'))"
done; done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
