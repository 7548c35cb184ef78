"""Experiment (GPU): a SCOUT launch in front of a single frame - the same view at a quarter of the resolution with four times the step (1 / 64 of
the work), so that its rays touch the bricks the frame is about to need (they then come from the MALL or an L2 instead of HBM).  Loose coupling
through two launches: what a renderer could do with the boundary as it is.  (For many frames in flight the MALL is far too small: DESIGN.md 5.1.)"""
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
div = int(sys.argv[1]) if len(sys.argv) > 1 else 4
step = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
views = bench.cameras(v, fw / fh)
ro = abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True)
sp = V.VolumeRenderSubpass(ctx, v, ro, (fw, fh))
sw, sh = (fw // div + 15) // 16 * 16, (fh // div + 15) // 16 * 16
ss = V.VolumeRenderSubpass(ctx, v, ro, (sw, sh))
real_buf = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
scout_buf = torch.zeros((sh, sw, 4), dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
R, S = [], []
for k in range(8):
    q = sp.make_params(*views[k])
    q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = real_buf.data_ptr(), None, None, None, None, 0
    R.append(q)
    q = ss.make_params(*views[k])
    q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = scout_buf.data_ptr(), None, None, None, None, 0
    q.transfer_function.sampling_factor = 1.0 / step
    S.append(q)
ctx.prepare_render(R + S, s1.cuda_stream)
torch.cuda.synchronize()


def run(mode, reps=6):
    ts = []
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        for k in range(8):
            if mode == "real":
                ctx.render(R[k], s2.cuda_stream)
            elif mode == "scout":
                ctx.render(S[k], s1.cuda_stream)
            elif mode == "both":      # scout and frame at the same time on two streams
                ctx.render(S[k], s1.cuda_stream)
                ctx.render(R[k], s2.cuda_stream)
            elif mode == "serial":    # the frame behind its scout on one stream
                ctx.render(S[k], s2.cuda_stream)
                ctx.render(R[k], s2.cuda_stream)
            torch.cuda.synchronize()
        if rep:
            ts.append((time.perf_counter() - t0) / 8 * 1e3)
    return float(np.median(ts))


print("scout = %dx%d pixels, step x %.0f; one frame at a time, host-synchronised after every frame (ms per frame):" % (sw, sh, step))
for m in ("real", "scout", "both", "serial", "real"):
    print("  %-7s %.4f" % (m, run(m)))
