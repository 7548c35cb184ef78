"""Diagnostic (GPU): HIP-event timing of the precompute kernels on a bench workload, with algorithmic GB/s."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vkvolume_amd import abi, lib, volume as V
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, name)
st = torch.cuda.current_stream().cuda_stream
nvox = v.extent.count; ncell = v.map_extent.count
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def report(label, ms, bytes_):
    print("%-34s %8.3f ms   %8.1f GB/s algorithmic" % (label, ms, bytes_ / ms / 1e6))
p = lambda t: None if t is None else t.data_ptr()
report("gradient_map (2 B/voxel)", timeit(lambda: ctx.gradient_map(p(v.volume), p(v.gradient), v.extent, tf, st)), 2 * nvox)
occ = torch.empty_like(v.distance_map_swap)
report("occupancy_map (2 B/voxel + 1 B/cell)", timeit(lambda: ctx.occupancy_map(p(v.volume), p(v.gradient), p(v.transfer_function), tf, v.extent, p(occ), v.map_extent, st)), 2 * nvox + ncell)
m = occ.clone(); sw = torch.empty_like(occ)
def dm():
    m.copy_(occ); ctx.distance_map(p(m), p(sw), v.map_extent, st)
def cp():
    m.copy_(occ)
t_cp = timeit(cp)
report("distance_map iso (6 B/cell)", timeit(dm) - t_cp, 6 * ncell)
maps = [torch.empty_like(occ) for _ in range(8)]
def dma():
    maps[7].copy_(occ); ctx.distance_map_anisotropic([p(x) for x in maps], p(sw), v.map_extent, st)
report("distance_map aniso (28 B/cell)", timeit(dma) - t_cp, 28 * ncell)
report("compute_distance_map mode 2 (update)", timeit(lambda: V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)), 2 * nvox + 7 * ncell)
report("compute_distance_map mode 3 (update)", timeit(lambda: V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_ANISOTROPIC_DISTANCE)), 2 * nvox + 29 * ncell)
report("pack_volume (2 B read + 4.3 B written/voxel)", timeit(lambda: v.pack()), 2 * nvox + ctx.packed_volume_bytes(v.extent))
cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
report("occupied_voxel_count (2 B/voxel)", timeit(lambda: ctx.occupied_voxel_count(p(v.volume), p(v.gradient), tf, v.extent, p(cnt), st)), 2 * nvox)
print("occupied voxels: %.4f %%" % (100.0 * cnt.item() / nvox), " occupied cells: %.3f %%" % (100.0 * (occ == 0).float().mean().item()),
      " mean distance %.2f max %d" % (m.float().mean().item(), int(m.max().item())))
