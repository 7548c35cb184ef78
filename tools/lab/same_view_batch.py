"""Experiment (GPU): how much of a C3 frame's time is the FIRST touch of its bricks (HBM latency)?  An 8-frame vkv_render_batch launch of
eight DIFFERENT views (the bench) against eight copies of ONE view: the copies march side by side, so seven of eight first touches of a brick
find it in the L2 / the MALL.  Same kernels, same work per frame (per view)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("VKV_RAYMARCH_FEEDBACK", "0")
import bench
from vkvolume_amd import abi, lib, volume as V
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
bufs = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(8)]


def params(view_ids):
    out = []
    for j, k in enumerate(view_ids):
        q = sp.make_params(*views[k])
        q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = bufs[j].data_ptr(), None, None, None, None, 0
        out.append(q)
    return out


def timed(plist, reps=60):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        ctx.render_batch(plist, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.render_batch(plist, s)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * len(plist)) * 1e3


mixed = timed(params(range(8)))
print("eight different views per launch (one launch at a time): %.4f ms per frame" % mixed)
same = []
for k in range(8):
    t = timed(params([k] * 8))
    same.append(t)
    print("  eight copies of view %d: %.4f ms per frame" % (k, t))
print("eight copies of one view, mean over the views: %.4f ms per frame (%.1f %% of the mixed launch)" % (np.mean(same), 100.0 * np.mean(same) / mixed))
