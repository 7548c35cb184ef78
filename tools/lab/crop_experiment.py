"""What do the workgroups OUTSIDE the volume's screen rectangle cost a frame?  (GPU; lab only)

The same rays twice, submitted the way bench.py submits them (7 + 7 + 6 frames per launch on three streams): once as the full 1920x1080
frame, once with the image cropped to the union of the eight views' covered rectangles (rounded out to 16 pixels; the ray generator shifted
by the crop origin), which is the frame minus every workgroup that only tests the screen bound and stores the clear value.  The difference is
the upper bound of what a launch grid restricted to the screen bound could win.
usage: crop_experiment.py [c3] [--blocks 30]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vkvolume_amd import abi, lib  # noqa: E402
from vkvolume_amd import volume as V  # noqa: E402


def main():
    workload = next((a for a in sys.argv[1:] if not a.startswith("--")), "c3")
    blocks = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 30
    ctx = lib.Context(0)
    v, tf, frame, skip = bench.build_scene(ctx, workload)
    fw, fh = frame
    views = bench.cameras(v, fw / fh)
    opts = abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True)
    sp = V.VolumeRenderSubpass(ctx, v, opts, (fw, fh))
    T = bench.TILE
    full = [sp.make_params(view, proj, abi.full_frame_tiles(fw, fh, T, T, 0, 1, compact=False)) for view, proj in views]
    # covered rectangle of every view from the frag counters
    counts = torch.zeros((fh * fw, 3), dtype=torch.int32, device="cuda")
    x0, y0, x1, y1 = fw, fh, 0, 0
    for p in full:
        counts.zero_()
        sp.draw(p, counts=counts)
        torch.cuda.synchronize()
        cov = ((counts[:, 0] + counts[:, 1]) > 0).view(fh, fw)
        ys, xs = cov.any(1).nonzero().flatten(), cov.any(0).nonzero().flatten()
        print("view rect: x %d..%d y %d..%d  covered %.1f %% of the frame" % (xs[0], xs[-1], ys[0], ys[-1], 100.0 * cov.float().mean().item()))
        x0, y0, x1, y1 = min(x0, int(xs[0])), min(y0, int(ys[0])), max(x1, int(xs[-1])), max(y1, int(ys[-1]))
    x0, y0 = x0 // T * T, y0 // T * T
    cw, ch = -(-(x1 + 1 - x0) // T) * T, -(-(y1 + 1 - y0) // T) * T
    cw, ch = min(cw, fw - x0), min(ch, fh - y0)
    print("union rect: origin (%d, %d) size %d x %d = %.1f %% of the frame's pixels" % (x0, y0, cw, ch, 100.0 * cw * ch / (fw * fh)))
    spc = V.VolumeRenderSubpass(ctx, v, opts, (cw, ch))
    crop = []
    for p in full:
        q = abi.RenderParams.from_buffer_copy(p)
        for i in range(3):
            q.ray_gen.dir00[i] = float(np.float32(np.float64(p.ray_gen.dir00[i]) + x0 * np.float64(p.ray_gen.ddx[i]) + y0 * np.float64(p.ray_gen.ddy[i])))
        q.image_width, q.image_height = cw, ch
        q.tiles = abi.full_frame_tiles(cw, ch, T, T, 0, 1, compact=False)
        crop.append(spc.bind(q))

    def timed(params, w, h, label):
        fpl, nbs, steps = 8, 3, 20
        nbuf = fpl * nbs
        bufs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        tiles = params[0].tiles
        for t in bufs:
            ctx.register_target(t.data_ptr(), (w, h), tiles)
        rows = []
        for view_i in range(len(params)):
            row = []
            for j in range(nbuf):
                q = abi.RenderParams.from_buffer_copy(params[view_i])
                q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth = bufs[j].data_ptr(), None, None, None
                q.d_in_depth, q.blend_over_target = None, 0
                row.append(q)
            rows.append(row)
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nbs - 1)]

        def block():
            k, launch = 0, 0
            for n in bench.split_frames(steps, fpl):
                st, slot = streams[launch % nbs], launch % nbs
                launch += 1
                plist = [rows[(k + j) % len(params)][slot * fpl + j] for j in range(n)]
                with torch.cuda.stream(st):
                    ctx.render_batch(plist, st.cuda_stream)
                k += n
        for _ in range(5):
            block()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(blocks):
            block()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / (blocks * steps)
        print("%-28s %4d x %4d: %.4f ms per frame" % (label, w, h, ms))
        for t in bufs:
            ctx.forget_target(t.data_ptr())
        return ms

    for _ in range(2):
        a = timed(full, fw, fh, "full frame")
        b = timed(crop, cw, ch, "cropped to the union rect")
        print("  -> the workgroups outside the rectangle cost %.4f ms per frame (%.1f %%)" % (a - b, 100.0 * (a - b) / a))


if __name__ == "__main__":
    main()
