import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from vkvolume_amd import abi, lib, volume as V
torch.cuda.set_device(0)
ctx = lib.Context(0)
ctx.set_tuning(feedback_period=1)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
t = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
tiles = abi.full_frame_tiles(fw, fh, 16, 16)
ctx.register_target(t.data_ptr(), (fw, fh), tiles)
p = sp.make_params(*views[0], tiles)
for _ in range(40):
    sp.draw(p, rgba8=t)
torch.cuda.synchronize()
