"""Diagnostic (GPU): the pack kernel alone (vkv_pack_volume on the C3 scene), HIP events, best and median of 5 blocks of 10 launches."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vkvolume_amd import lib
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
run = lambda: v.pack()
run(); torch.cuda.synchronize()
times = []
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    times.append(s.elapsed_time(e) / 10)
times.sort()
print("pack_volume c3: best %.4f median %.4f ms" % (times[0], times[2]))
