"""Experiment (GPU): start-order feedback with a camera that MOVES between the frames into a target (bench.py's targets always show the
same view).  Three streams, 7 frames per launch; target j of a stream shows azimuth 45 j + step * launch degrees.
    python tools/feedback_moving_camera.py [degrees per launch]      (run once with VKV_RAYMARCH_FEEDBACK=0 for the reference)"""
import sys, os, time, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vkvolume_amd import abi, camera, lib, volume as V
step = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
m = (v.node_transform.astype(np.float64).T @ v.image_transform.astype(np.float64).T)[:3, :3]
radius = 1.5 * 0.5 * math.sqrt(sum(float(np.linalg.norm(m[:, i])) ** 2 for i in range(3)))
proj = camera.perspective_vulkan(60.0, fw / fh, 0.1, 1000.0)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
n, launches = 7, 120
bufs = [[torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(n)] for _ in range(3)]
for row in bufs:
    for t in row:
        ctx.register_target(t.data_ptr(), (fw, fh), abi.full_frame_tiles(fw, fh))  # the per-target feedback state (vkv_render* never allocate)
# parameter blocks prepared ahead (the uniforms are host work outside the timed loop)
plists = []
for l in range(launches):
    s = l % 3
    pl = []
    for j in range(n):
        q = sp.make_params(camera.orbit_camera(45.0 * (j + 7 * s) + step * (l // 3), 20.0, radius), proj)
        q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = bufs[s][j].data_ptr(), None, None, None, None, 0
        pl.append(q)
    plists.append(pl)
for l in range(6):
    ctx.render_batch(plists[l], streams[l % 3].cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for l in range(6, launches):
    ctx.render_batch(plists[l], streams[l % 3].cuda_stream)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / ((launches - 6) * n) * 1e3
print("camera moves %.2f degrees between two frames into a target, feedback %s: %.4f ms per frame" % (
    step, "off" if os.environ.get("VKV_RAYMARCH_FEEDBACK") == "0" else "on (period %s)" % os.environ.get("VKV_RAYMARCH_FEEDBACK_PERIOD", "8"), ms))
