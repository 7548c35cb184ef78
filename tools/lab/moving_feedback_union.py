"""Experiment (GPU, round 6): can the start-order feedback follow a MOVING camera when its state is not keyed by the frame's rectangle?  3 render
targets, launches of 3 frames on one stream, an orbit of STEP degrees per frame (96 views, back and forth).  Schedules: every view its own
screen rectangle (a target's feedback entry only matches frames with the same rectangle) against ONE rectangle for all views (the union:
every frame into a target finds the target's entry) - feedback on / off.   usage: moving_feedback_union.py [STEP] [F]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vkvolume_amd import abi, lib, volume as V
STEP = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
F = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NV = 96
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh, NV, STEP)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
whole = abi.full_frame_tiles(fw, fh, 16, 16)
st = torch.cuda.current_stream().cuda_stream
targets = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(F)]
base = [sp.make_params(view, proj, whole) for view, proj in views]
rects = [lib.screen_tile_rect(p.ray_cast, p.ray_gen, (fw, fh), (16, 16)) for p in base]
x0, y0 = min(r.x0 for r in rects), min(r.y0 for r in rects)
x1, y1 = max(r.x0 + r.w for r in rects), max(r.y0 + r.h for r in rects)
union = abi.TileRect(x0, y0, x1 - x0, y1 - y0)
print("step %.2f deg: %d distinct rectangles over %d views (tiles %d .. %d), union %d tiles" % (STEP, len({r.as_tuple() for r in rects}), NV, min(r.w * r.h for r in rects),
                                                                                             max(r.w * r.h for r in rects), union.w * union.h))
seq = list(range(NV)) + list(range(NV - 2, 0, -1))  # back and forth: consecutive frames always STEP apart


def run(kind, feedback):
    ctx.set_tuning(feedback=feedback)
    tiles = [abi.full_frame_tiles(fw, fh, 16, 16, rect=(union if kind == "union" else rects[i]), fill_outside=True) for i in range(NV)]
    if feedback:
        seen = set()
        for k, i in enumerate(seq):
            key = (k % F, tiles[i].rect.x0, tiles[i].rect.y0, tiles[i].rect.w, tiles[i].rect.h)
            if key not in seen:
                seen.add(key)
                ctx.register_target(targets[k % F].data_ptr(), (fw, fh), tiles[i])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = []
    for rnd in range(3):
        torch.cuda.synchronize()
        e0.record()
        n = 0
        for k0 in range(0, len(seq) - F + 1, F):
            ps = []
            for j in range(F):
                i = seq[k0 + j]
                q = abi.RenderParams.from_buffer_copy(base[i])
                q.tiles = tiles[i]
                q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = targets[(k0 + j) % F].data_ptr(), None, None, None, None, 0
                ps.append(q)
            ctx.render_batch(ps, st)
            n += F
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / n)
    return min(res[1:])


for rep in range(2):
    for kind in ("own", "union"):
        for fb in (0, 1):
            print("F %d step %.2f  rectangle %-6s feedback %d   %.4f ms per frame" % (F, STEP, kind, fb, run(kind, fb)))
