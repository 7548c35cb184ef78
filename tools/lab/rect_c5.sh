B="python bench.py --steps 16 --warmup 4 --no-cpu-baseline --extras off --no-depth-block --min-seconds 1.0"
run() { label=$1; shift; $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %.4f' % ('$label', d['ms_per_step']))"; }
for w in c5 c4 c2 c3cube; do
run "$w whole" --workload $w --tile-rect off
run "$w fill" --workload $w
VKV_RAYMARCH_TILE_ORDER=linear run "$w whole linear" --workload $w --tile-rect off
done
