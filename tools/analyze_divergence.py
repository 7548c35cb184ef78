"""Diagnostic (GPU): per-view kernel time and SIMD-lane efficiency of the ray-march on a bench workload.
efficiency = sum over rays of loop events / (64 * sum over 8x8 wave tiles of the longest ray's events)."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vkvolume_amd import abi, lib, volume as V

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
TILE = int(os.environ.get("VKV_TILE", "16"))
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, name)
fw, fh = frame
views = bench.cameras(v, fw / fh)
cases = ((skip, True), (skip, False), (abi.SKIP_NONE, True), (abi.SKIP_NONE, False), (abi.SKIP_BLOCK, True), (abi.SKIP_ANISOTROPIC_DISTANCE, True))
for mode, ert in (cases[:1] + cases[3:4] if quick else cases):
    V.ComputeDistanceMap(ctx).compute(v, tf, mode)
    sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=mode, clip_distance=1.0, early_ray_termination=ert), (fw, fh))
    counts = torch.zeros((fh, fw, 3), dtype=torch.int32, device="cuda")
    rgba8 = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
    rows = []
    for i, (view, proj) in enumerate(views):
        p = sp.make_params(view, proj, abi.full_frame_tiles(fw, fh, TILE, TILE))
        sp.draw(p, counts=counts); torch.cuda.synchronize()
        c = counts.cpu().numpy().astype(np.int64)
        ev = c[..., 0] + c[..., 1]
        hh, ww = (fh // 8) * 8, (fw // 8) * 8
        t = ev[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8)
        tile_max = t.max(axis=(1, 3)); tile_sum = t.sum(axis=(1, 3))
        eff = tile_sum.sum() / (64.0 * tile_max.sum())
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2): sp.draw(p, rgba8=rgba8)
        s.record()
        for _ in range(5): sp.draw(p, rgba8=rgba8)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        if quick:
            rows.append(dict(view=i, ms=round(ms, 4)))
            continue
        rows.append(dict(view=i, ms=round(ms, 4), vol=int(c[..., 0].sum()), probes=int(c[..., 1].sum()), empty=int(c[..., 2].sum()),
                         covered=float((ev > 0).mean()), lane_eff=round(float(eff), 4), max_events=int(ev.max()),
                         wave_iters=int(tile_max.sum())))
    tot_ms = sum(r["ms"] for r in rows)
    print(json.dumps(dict(mode=mode, ert=ert, mean_ms=round(tot_ms / len(rows), 4), Mray_s=round(fw * fh * len(rows) / tot_ms / 1e3, 1),
                          sched=os.environ.get("VKV_RAYMARCH_SCHEDULER", "tiles"), tile=TILE,
                          Gsamples_s=None if quick else round(sum(r["vol"] for r in rows) / tot_ms / 1e6, 2), rows=rows)))
