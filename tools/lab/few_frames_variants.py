"""Experiment (GPU, round 6): why launches of 2 - 3 frames on ONE stream lose with the screen rectangle + fill_outside (tools/lab/few_frames.sh).
Back-to-back vkv_render_batch launches of F frames on one stream, start-order feedback off, variants of the schedule:
  whole        every tile of the image (centre-first table)
  whole-linear every tile, plain order (VkvTuning.tile_order_linear)
  fill         the view's rectangle + fill_outside (plain order)
  rect-nofill  the rectangle into the image, nothing outside written (not a whole frame: timing only)
  fill-union   one rectangle (the union over the views) for every frame + fill_outside: equal tile counts
usage: few_frames_variants.py [F] [workload]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vkvolume_amd import abi, lib, volume as V
F = int(sys.argv[1]) if len(sys.argv) > 1 else 3
torch.cuda.set_device(0)
ctx = lib.Context(0)
ctx.set_tuning(feedback=0)
v, tf, frame, skip = bench.build_scene(ctx, sys.argv[2] if len(sys.argv) > 2 else "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
whole = abi.full_frame_tiles(fw, fh, 16, 16)
st = torch.cuda.current_stream().cuda_stream
targets = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(8)]
base = [sp.make_params(view, proj, whole) for view, proj in views]
rects = [lib.screen_tile_rect(p.ray_cast, p.ray_gen, (fw, fh), (16, 16)) for p in base]
x0, y0 = min(r.x0 for r in rects), min(r.y0 for r in rects)
x1, y1 = max(r.x0 + r.w for r in rects), max(r.y0 + r.h for r in rects)
union = abi.TileRect(x0, y0, x1 - x0, y1 - y0)
print("tiles: whole %d, rectangles %s, union %d" % (whole.tile_count, [r.w * r.h for r in rects], union.w * union.h))


def params(kind):
    out = []
    for i, p in enumerate(base):
        q = abi.RenderParams.from_buffer_copy(p)
        if kind in ("fill", "fill-table", "fill-midout", "fill-rings"):
            q.tiles = abi.full_frame_tiles(fw, fh, 16, 16, rect=rects[i], fill_outside=True)
        elif kind == "rect-nofill":
            q.tiles = abi.full_frame_tiles(fw, fh, 16, 16, rect=rects[i])
        elif kind == "fill-union":
            q.tiles = abi.full_frame_tiles(fw, fh, 16, 16, rect=union, fill_outside=True)
        q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = targets[i].data_ptr(), None, None, None, None, 0
        out.append(q)
    return out


def timed(ps, launches=16):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k = 0
    for rnd in range(2):
        torch.cuda.synchronize()
        e0.record()
        for _ in range(launches):
            ctx.render_batch([ps[(k + j) % 8] for j in range(F)], st)
            k += F
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (launches * F)


FEEDBACK = int(os.environ.get("FEEDBACK", "0"))
for rep in range(2):
    for kind in ("whole", "fill", "fill-table", "fill-midout", "fill-rings"):
        ctx.set_tuning(tile_order_linear={"whole-linear": 1, "fill-table": 2, "fill-midout": 3, "fill-rings": 4}.get(kind, 0), feedback=FEEDBACK)
        ps = params(kind)
        if FEEDBACK:
            for i, q in enumerate(ps):
                ctx.register_target(targets[i].data_ptr(), (fw, fh), q.tiles)
        print("F %d feedback %d %-13s %.4f ms per frame" % (F, FEEDBACK, kind, timed(ps)))
