"""Diagnostic (GPU): distribution of per-ray event counts on a bench workload (what the longest rays are made of)."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vkvolume_amd import abi, lib, volume as V
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, name)
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
counts = torch.zeros((fh, fw, 3), dtype=torch.int32, device="cuda")
for i in (0, 3):
    p = sp.make_params(*views[i])
    sp.draw(p, counts=counts); torch.cuda.synchronize()
    c = counts.cpu().numpy().astype(np.int64)
    ev = (c[..., 0] + c[..., 1]).ravel(); vol = c[..., 0].ravel(); pr = c[..., 1].ravel()
    cov = ev > 0
    qs = [50, 90, 99, 99.9, 100]
    print("view", i, "covered", cov.mean(), "events/ray percentiles", dict(zip(qs, np.percentile(ev[cov], qs).round(1))))
    for thr in (50, 100, 150, 200):
        m = ev >= thr
        print("  rays with >= %d events: %d (%.3f%% of covered), their share of all events %.1f%%, vol share inside them %.2f" % (
            thr, m.sum(), 100 * m.sum() / cov.sum(), 100 * ev[m].sum() / ev.sum(), vol[m].sum() / max(ev[m].sum(), 1)))
    hh, ww = (fh // 8) * 8, (fw // 8) * 8
    t = (c[..., 0] + c[..., 1])[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8).max(axis=(1, 3)).ravel()
    print("  8x8 tiles: n=%d, max-events percentiles" % t.size, dict(zip(qs, np.percentile(t[t > 0], qs).round(1))), "tiles>=150:", int((t >= 150).sum()))
    b = (c[..., 0] + c[..., 1])[: (fh // 16) * 16, : (fw // 16) * 16].reshape(fh // 16, 16, fw // 16, 16)
    bm = b.max(axis=(1, 3)); print("  16x16 blocks max-events map rows (coarse):", [int(x) for x in bm.max(axis=1)])
