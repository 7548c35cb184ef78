"""Diagnostic (GPU): per-wave timeline of one ray-march launch (start, end, iterations) via vkv_debug_trace."""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vkvolume_amd import abi, lib, volume as V
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, name)
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
p = sp.make_params(*views[view])
rgba8 = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
nblocks = ((p.tiles.tile_count + 7) // 8) * 8
trace = torch.zeros((nblocks * 4, 10), dtype=torch.int64, device="cuda")  # kTraceWords = 10 per wave
for _ in range(3): sp.draw(p, rgba8=rgba8)
torch.cuda.synchronize()
L = lib.load(); L.vkv_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
L.vkv_debug_trace(ctx.handle, trace.data_ptr())
sp.draw(p, rgba8=rgba8); torch.cuda.synchronize()
L.vkv_debug_trace(ctx.handle, None)
t = trace.cpu().numpy()
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
start, end, it = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2]   # microseconds (100 MHz)
xcc = t[:, 3] >> 32
dur = end - start
print("waves traced", len(t), "kernel span %.1f us" % end.max())
print("start time percentiles (us):", np.percentile(start, [50, 90, 99, 100]).round(1))
print("end time percentiles (us):", np.percentile(end, [50, 90, 99, 99.9, 100]).round(1))
m = it > 0
print("marching waves", m.sum(), "iterations percentiles", np.percentile(it[m], [50, 90, 99, 100]))
print("us per iteration (waves with >=64 iters): median %.3f  p10 %.3f p90 %.3f" % tuple(np.percentile((dur / np.maximum(it, 1))[it >= 64], [50, 10, 90])))
order = np.argsort(-end)[:12]
print("last finishing waves: (start, end, dur, iters, us/iter, xcc)")
for i in order:
    print("   %.1f %.1f %.1f %d %.3f %d" % (start[i], end[i], dur[i], it[i], dur[i] / max(it[i], 1), xcc[i]))
order = np.argsort(-it)[:8]
print("longest waves by iterations:")
for i in order:
    print("   %.1f %.1f %.1f %d %.3f %d" % (start[i], end[i], dur[i], it[i], dur[i] / max(it[i], 1), xcc[i]))
ts = np.arange(0, end.max(), 10.0)
occ = [(int(((start <= x) & (end > x) & m).sum())) for x in ts]
print("resident marching waves every 10us:", occ)
print("per-XCD end time:", [round(float(end[xcc == x].max()), 1) if (xcc == x).any() else None for x in range(8)])
