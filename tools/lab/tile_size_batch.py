"""Experiment (GPU): schedule tile size of the bench's batch launches (16x16 = one workgroup per tile, the default; larger tiles keep the
16x16 blocks of a tile on one XCD, i.e. behind one L2).  Three launches of 7 + 7 + 6 frames on three streams, registered targets (start-order
feedback per tile), ms per frame."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vkvolume_amd import abi, lib, volume as V
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
streams = [torch.cuda.Stream() for _ in range(3)]
for tw, th in ((16, 16), (32, 16), (32, 32), (64, 32), (64, 64), (128, 64)):
    tiles = abi.full_frame_tiles(fw, fh, tw, th)
    sizes, k, sets = [7, 7, 6], 0, []
    for n in sizes:
        bufs = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
        plist = []
        for j in range(n):
            q = sp.make_params(*views[(k + j) % 8], tiles=tiles)
            q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = bufs[j].data_ptr(), None, None, None, None, 0
            ctx.register_target(bufs[j].data_ptr(), (fw, fh), tiles)
            plist.append(q)
        sets.append((bufs, plist))
        k += n
    ctx.prepare_render(sets[0][1], streams[0].cuda_stream)

    def block():
        for s in range(3):
            ctx.render_batch(sets[s][1], streams[s].cuda_stream)
        torch.cuda.synchronize()
    for _ in range(20):
        block()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(40):
            block()
        ts.append((time.perf_counter() - t0) / (40 * 20) * 1e3)
    print("tiles %3d x %3d: %.4f ms per frame (min %.4f)" % (tw, th, float(np.median(ts)), min(ts)))
    for b, _p in sets:
        for t in b:
            ctx.forget_target(t.data_ptr())
