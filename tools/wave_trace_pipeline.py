"""Diagnostic (GPU): resident waves over time with several frames in flight (per-launch vkv_debug_trace buffers)."""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vkvolume_amd import abi, lib, volume as V
fif = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_frames = 12
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
params = [sp.make_params(*vw) for vw in views]
bufs = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(fif)]
nblocks = ((params[0].tiles.tile_count + 7) // 8) * 8
traces = [torch.zeros((nblocks * 4, 10), dtype=torch.int64, device="cuda") for _ in range(n_frames)]  # kTraceWords = 10 per wave
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(fif - 1)]
L = lib.load(); L.vkv_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
for k in range(16):
    with torch.cuda.stream(streams[k % fif]):
        sp.draw(params[k % 8], rgba8=bufs[k % fif])
torch.cuda.synchronize()
for k in range(n_frames):
    L.vkv_debug_trace(ctx.handle, traces[k].data_ptr())
    with torch.cuda.stream(streams[k % fif]):
        sp.draw(params[k % 8], rgba8=bufs[k % fif])
L.vkv_debug_trace(ctx.handle, None)
torch.cuda.synchronize()
recs = []
for k in range(n_frames):
    t = traces[k].cpu().numpy()
    t = t[t[:, 1] > 0]
    recs.append(np.column_stack([t[:, 0], t[:, 1], t[:, 2], np.full(len(t), k)]))
r = np.concatenate(recs)
t0 = r[:, 0].min()
start, end, it, fr = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0, r[:, 2], r[:, 3]
print("frames in flight", fif, ": %d frames span %.1f us -> %.1f us per frame" % (n_frames, end.max(), end.max() / n_frames))
for k in range(n_frames):
    m = fr == k
    print("  frame %2d: first wave starts %.1f, last wave ends %.1f (span %.1f)" % (k, start[m].min(), end[m].max(), end[m].max() - start[m].min()))
ts = np.arange(0, end.max(), 20.0)
m = it > 0
occ_all = [int(((start <= x) & (end > x)).sum()) for x in ts]
occ_m = [int(((start <= x) & (end > x) & m).sum()) for x in ts]
print("resident waves every 20 us (all / marching), capacity 8192:")
print("  ", list(zip(occ_all, occ_m)))
mid = (ts > end.max() * 0.25) & (ts < end.max() * 0.75)
print("steady-state mean resident waves: %.0f (marching %.0f)" % (np.mean(np.array(occ_all)[mid]), np.mean(np.array(occ_m)[mid])))
