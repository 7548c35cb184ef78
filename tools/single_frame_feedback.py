import sys, os, time, math
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from vkvolume_amd import abi, camera, lib, volume as V
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
m = (v.node_transform.astype(np.float64).T @ v.image_transform.astype(np.float64).T)[:3, :3]
radius = 1.5 * 0.5 * math.sqrt(sum(float(np.linalg.norm(m[:, i])) ** 2 for i in range(3)))
proj = camera.perspective_vulkan(60.0, fw / fh, 0.1, 1000.0)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
bufs = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(8)]
for _t in bufs:
    ctx.register_target(_t.data_ptr(), (fw, fh), abi.full_frame_tiles(fw, fh))  # feedback state of the targets
ps = []
for j in range(8):
    q = sp.make_params(camera.orbit_camera(45.0 * j, 20.0, radius), proj)
    q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = bufs[j].data_ptr(), None, None, None, None, 0
    ps.append(q)
st = torch.cuda.current_stream().cuda_stream
def run(fn, reps=40):
    for j in range(16): fn(ps[j % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for j in range(8): fn(ps[j])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * 8) * 1e3
print("vkv_render, one frame per launch:        %.4f ms" % run(lambda p: ctx.render(p, st)))
print("vkv_render_batch of 1 (measured order):  %.4f ms" % run(lambda p: ctx.render_batch([p], st)))
print("vkv_render again:                        %.4f ms" % run(lambda p: ctx.render(p, st)))
