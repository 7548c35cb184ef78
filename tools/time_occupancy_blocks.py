"""Diagnostic (GPU): the map update of the reference's sweep block sizes on a bench volume - occupancy pass alone and occupancy + isotropic
distance transform (what the reference's benchmark mode logs as "Updated occupancy/distance map in X ms", src/volume_render.cpp:422-430).
usage: python tools/time_occupancy_blocks.py [workload]      (VKV_LIB_PATH selects another build of the library)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vkvolume_amd import abi, lib, volume as V  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
torch.cuda.set_device(0)
ctx = lib.Context(0)
extent, seed = bench.WORKLOADS[name][0], bench.WORKLOADS[name][1]
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


print("# %s %dx%dx%d, library %s, VKV_OCCUPANCY_KERNEL=%s" % (name, *extent, os.environ.get("VKV_LIB_PATH", "(this tree)"), os.environ.get("VKV_OCCUPANCY_KERNEL", "(default)")))
print("%5s %14s %10s %12s %12s %12s" % ("block", "map", "occ ms", "occ GB/s", "update iso", "update aniso"))
for b in (2, 3, 4, 5, 6):
    v = V.Volume(ctx)
    v.options = abi.VolumeOptions(intensity_min=0.1, intensity_max=1.0, gradient_min=0.0, gradient_max=0.2)
    v.load_synthetic(extent, kind=1, seed=seed, distance_map_block_size=b)
    V.default_scene(v)
    tf = v.get_transfer_function_uniform()
    V.ComputeGradientMap(ctx).compute(v, tf)
    v.update_transfer_function_texture()
    occ = torch.empty_like(v.distance_map_swap)
    p = lambda t: t.data_ptr()  # noqa: E731
    t_occ = timeit(lambda: ctx.occupancy_map(p(v.volume), p(v.gradient), p(v.transfer_function), tf, v.extent, p(occ), v.map_extent, st))
    cdm = V.ComputeDistanceMap(ctx)
    t_iso = timeit(lambda: cdm.compute(v, tf, abi.SKIP_DISTANCE))
    t_an = timeit(lambda: cdm.compute(v, tf, abi.SKIP_ANISOTROPIC_DISTANCE), n=5)
    me = v.map_extent
    print("%5d %14s %10.3f %12.1f %12.3f %12.3f" % (b, "%dx%dx%d" % (me.width, me.height, me.depth), t_occ, (2 * v.extent.count + me.count) / t_occ / 1e6, t_iso, t_an))
    del v, occ
    torch.cuda.empty_cache()
