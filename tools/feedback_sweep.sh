#!/bin/bash
# usage (GPU box, repo root): tools/feedback_sweep.sh      the start-order feedback against a camera that moves: bench.py's submission (6 frames per
# launch, 4 streams, 24 targets taking the frames in turn) with --camera moving at 0.25 / 1 / 3 degrees of orbit per frame and with the camera
# held still (the headline), each with VkvTuning.feedback on and off; 24- and 96-step blocks.  profiles/r6_feedback_moving_camera.txt keeps a run.
B="python bench.py --no-cpu-baseline --extras off --no-depth-block --min-seconds 1.0"
for steps in 24 96; do
for st in 0.25 1 3; do
  $B --steps $steps --warmup 6 --camera moving --camera-step $st 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $steps moving $st deg/frame  feedback on : %.4f ms per frame' % d['ms_per_step'])"
  $B --steps $steps --warmup 6 --camera moving --camera-step $st --no-feedback 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $steps moving $st deg/frame  feedback off: %.4f ms per frame' % d['ms_per_step'])"
done
$B --steps $steps --warmup 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $steps static (8 views, one per target)  feedback on : %.4f ms per frame' % d['ms_per_step'])"
$B --steps $steps --warmup 6 --no-feedback 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $steps static (8 views, one per target)  feedback off: %.4f ms per frame' % d['ms_per_step'])"
done
