"""Diagnostic (GPU): resident waves over time inside ONE vkv_render_batch launch (per-wave records of vkv_debug_trace)."""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vkvolume_amd import abi, lib, volume as V
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
bufs = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
plist = []
for j in range(n):
    q = sp.make_params(*views[j % 8])
    q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = bufs[j].data_ptr(), None, None, None, None, 0
    plist.append(q)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    ctx.render_batch(plist, st)
torch.cuda.synchronize()
nblocks = ((plist[0].tiles.tile_count + 7) // 8) * 8 * n
trace = torch.zeros((nblocks * 4, 10), dtype=torch.int64, device="cuda")
L = lib.load(); L.vkv_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
L.vkv_debug_trace(ctx.handle, trace.data_ptr())
ctx.render_batch(plist, st); torch.cuda.synchronize()
L.vkv_debug_trace(ctx.handle, None)
t = trace.cpu().numpy()
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
start, end, it = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2]
print("waves traced", len(t), "launch span %.1f us = %.1f us per frame" % (end.max(), end.max() / n))
m = it > 0
print("marching waves", int(m.sum()), "wave iterations", int(it.sum()), "= %.0f per frame" % (it.sum() / n))
cl = t[:, 9]
print("clamp-free loop: %d of those iterations took the rare clamp branch (%.2f %%), %.2f per marching wave; waves that never left the loop with the clamps count 0 here" % (
    int(cl.sum()), 100.0 * cl.sum() / max(it.sum(), 1), cl.sum() / max(m.sum(), 1)))
ts = np.arange(0, end.max(), 25.0)
occ = [int(((start <= x) & (end > x)).sum()) for x in ts]
occ_m = [int(((start <= x) & (end > x) & m).sum()) for x in ts]
print("resident waves every 25 us (all / marching), capacity 8192:")
print("  ", list(zip(occ, occ_m)))
print("mean resident waves over the launch: %.0f" % (np.sum(end - start) / end.max()))
busy = np.sum((end - start)[m]) / end.max()
print("mean resident marching waves: %.0f; wave-microseconds per wave iteration: %.3f" % (busy, np.sum((end - start)[m]) / it.sum()))
order = np.argsort(-end)[:16]
print("last finishing units (start, end, duration, iterations, us per iteration, xcc, unit id):")
for i in order:
    print("   %.1f %.1f %.1f %d %.3f %d %d" % (start[i], end[i], end[i] - start[i], it[i], (end[i] - start[i]) / max(it[i], 1), t[i, 3] >> 32, t[i, 3] & 0xffffffff))
late = start > 0.6 * end.max()
print("units started in the last 40 %% of the launch: %d, of them marching %d, with >= 100 iterations %d" % (late.sum(), (late & m).sum(), (late & (it >= 100)).sum()))
for lo in (0, 0.2, 0.4, 0.6, 0.8):
    sel = (start >= lo * end.max()) & (start < (lo + 0.2) * end.max())
    print("  started in [%.0f%%, %.0f%%): %6d units, mean iterations %.1f, max %d" % (lo * 100, lo * 100 + 20, sel.sum(), it[sel].mean() if sel.any() else 0, it[sel].max() if sel.any() else 0))
if t[:, 4].max() > 0:
    su, ma = (t[:, 4] - t0) / 100.0, (t[:, 5] - t0) / 100.0
    big = it >= 100
    print("units with >= 100 iterations: set-up %.1f us, march %.1f us (%.3f us per iteration), write-out %.1f us (means)" % (
        (su - start)[big].mean(), (ma - su)[big].mean(), ((ma - su)[big] / it[big]).mean(), (end - ma)[big].mean()))
    for i in order[:6]:
        print("   tail unit: set-up %.1f march %.1f write-out %.1f" % (su[i] - start[i], ma[i] - su[i], end[i] - ma[i]))
if t.shape[1] > 7 and t[:, 7].max() > 0:
    long = (end - start) > 50.0
    if long.any():
        mhz = (t[long, 7] - t[long, 6]) / (end[long] - start[long])  # shader-clock ticks per microsecond
        print("shader clock seen by waves that ran > 50 us: median %.0f MHz (p10 %.0f, p90 %.0f)" % tuple(np.percentile(mhz, [50, 10, 90])))
if len(sys.argv) > 2:
    # per-wave records for offline scheduling experiments: start, end (us), wave iterations, launch-wide wave index
    np.save(sys.argv[2], np.column_stack([start, end, it, np.nonzero(trace.cpu().numpy()[:, 1] > 0)[0]]))
