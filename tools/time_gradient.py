"""Diagnostic (GPU): the gradient-map kernel alone, HIP events, best and median of several blocks of 20 launches.
    python tools/time_gradient.py [c3|c4]      (VKV_GRADIENT_KERNEL / VKV_GRADIENT_SEG select the experimental variants)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vkvolume_amd import lib
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, name)
st = torch.cuda.current_stream().cuda_stream
run = lambda: ctx.gradient_map(v.volume.data_ptr(), v.gradient.data_ptr(), v.extent, tf, st)
run(); torch.cuda.synchronize()
times = []
for _ in range(7):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    times.append(s.elapsed_time(e) / 20)
times.sort()
print("gradient_map %s: best %.4f ms, median %.4f ms (%.0f GB/s algorithmic at the median)" % (name, times[0], times[3], 2 * v.extent.count / times[3] / 1e6))
