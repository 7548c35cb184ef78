B="python bench.py --steps 24 --warmup 6 --no-cpu-baseline --extras off --no-depth-block --min-seconds 1.0"
run() { label=$1; shift; $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-52s %.4f' % ('$label', d['ms_per_step']))"; }
for fb in "" "--no-feedback"; do
  run "static whole $fb" --tile-rect off $fb
  run "static fill $fb" $fb
  run "static compact rect (virtual 0/1) $fb" --virtual-rank 0/1 $fb
  run "moving whole $fb" --camera moving --tile-rect off $fb
  run "moving fill $fb" --camera moving $fb
  run "moving compact rect (virtual 0/1) $fb" --camera moving --virtual-rank 0/1 $fb
done
