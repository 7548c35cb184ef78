"""Experiment (GPU, round 6): one frame at a time through vkv_render (argument block by value, every tile scheduled) against vkv_render_batch with ONE frame
(argument block uploaded, the frame's screen rectangle + fill_outside)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vkvolume_amd import abi, lib, volume as V
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, sys.argv[1] if len(sys.argv) > 1 else "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
whole = abi.full_frame_tiles(fw, fh, 16, 16)
st = torch.cuda.current_stream().cuda_stream
res = {"render whole": [], "batch1 whole": [], "batch1 fill": []}
for i, (view, proj) in enumerate(views):
    t = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
    p = sp.make_params(view, proj, whole)
    rect = lib.screen_tile_rect(p.ray_cast, p.ray_gen, (fw, fh), (16, 16))
    for name, tiles, fn in (("render whole", whole, "r"), ("batch1 whole", whole, "b"), ("batch1 fill", abi.full_frame_tiles(fw, fh, 16, 16, rect=rect, fill_outside=True), "b")):
        q = abi.RenderParams.from_buffer_copy(p)
        q.tiles = tiles
        q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = t.data_ptr(), None, None, None, None, 0
        ctx.forget_target(t.data_ptr()) if False else None
        ctx.register_target(t.data_ptr(), (fw, fh), tiles)
        ts = []
        for rnd in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            ctx.render(q, st) if fn == "r" else ctx.render_batch([q], st)
            e1.record()
            torch.cuda.synchronize()
            if rnd >= 3:
                ts.append(e0.elapsed_time(e1))
        res[name].append(float(np.median(ts)))
for k, vals in res.items():
    print("%-14s mean %.4f ms   per view %s" % (k, float(np.mean(vals)), " ".join("%.3f" % x for x in vals)))
