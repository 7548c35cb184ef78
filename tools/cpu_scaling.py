"""Thread scaling of the CPU oracle's ray-marcher on this host (the bench line's cpu_baseline): Mray/s of view 0 of a workload at 1 .. all threads.
    python tools/cpu_scaling.py [workload] [pixel stride]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from oracle import vkv_oracle as O  # noqa: E402
from vkvolume_amd import abi, lib, volume as V  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 2
try:
    print(subprocess.check_output("lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA|L3'", shell=True, text=True))
except Exception as e:  # noqa: BLE001
    print("lscpu:", e)
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, name)
views = bench.cameras(v, frame[0] / frame[1])
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), frame)
p = sp.make_params(*views[0], abi.full_frame_tiles(frame[0], frame[1], 16, 16))
vol, grad, tex = v.volume.cpu().numpy(), v.gradient.cpu().numpy(), v.transfer_function.cpu().numpy()
maps = [m.cpu().numpy() for m in v.distance_maps]
cores = os.cpu_count()
r = O.render(p, vol, grad, tex, maps, n_threads=cores, pixel_stride=stride)  # warm: pool threads, page faults
base = None
for n in [1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256]:
    if n > cores:
        break
    best = 0.0
    for _ in range(3 if n > 1 else 1):
        r = O.render(p, vol, grad, tex, maps, n_threads=n, pixel_stride=stride if n > 4 else stride * 4, reuse=r if n > 4 else None)
        best = max(best, r.rays / r.seconds / 1e6)
    base = base or best
    print("threads %3d: %8.3f Mray/s  = %6.1f x one thread" % (n, best, best / base), flush=True)

# the bench's own call pattern: all 8 views, every pixel, outputs reused between passes
params = [sp.make_params(*vw, abi.full_frame_tiles(frame[0], frame[1], 16, 16)) for vw in views]
last = [O.render(q, vol, grad, tex, maps, n_threads=cores, pixel_stride=4, want_rgba8=True) for q in params]
for ps in range(3):
    secs = []
    for i, q in enumerate(params):
        last[i] = O.render(q, vol, grad, tex, maps, n_threads=cores, pixel_stride=1, want_rgba8=True, reuse=last[i])
        secs.append(last[i].seconds)
    print("pass %d, every pixel of 8 views on %d threads: %s s per view = %.1f Mray/s" % (ps, cores, " ".join("%.3f" % x for x in secs),
                                                                                        sum(r.rays for r in last) / sum(secs) / 1e6), flush=True)
