"""Diagnostic (GPU): what the y / z passes of the distance transform see - distribution of the input values per workgroup tile (8 lines x the
whole axis), to size the sparse table by the largest value of a tile.   python tools/dm_tile_stats.py [workload]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vkvolume_amd import lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, name)
st = torch.cuda.current_stream().cuda_stream
occ = torch.empty_like(v.distance_map_swap)
ctx.occupancy_map(v.volume.data_ptr(), v.gradient.data_ptr(), v.transfer_function.data_ptr(), tf, v.extent, occ.data_ptr(), v.map_extent, st)
torch.cuda.synchronize()
md, mh, mw = occ.shape
print("map", (mw, mh, md), "occupied cells %.2f %%" % (100.0 * (occ == 0).float().mean().item()))


def one_d(g, axis):
    """min over q of max(|p - q|, g(q)) along `axis`, brute force in chunks (int16)"""
    g = g.movedim(axis, -1).contiguous().to(torch.int16)
    n = g.shape[-1]
    idx = torch.arange(n, device=g.device, dtype=torch.int16)
    dist = (idx[:, None] - idx[None, :]).abs()  # [p, q]
    flat = g.reshape(-1, n)
    out = torch.empty_like(flat)
    step = max(1, (1 << 28) // (n * n))
    for a in range(0, flat.shape[0], step):
        blk = flat[a:a + step]
        out[a:a + step] = torch.maximum(dist[None], blk[:, None, :]).min(-1).values
    return out.reshape(g.shape).movedim(-1, axis).clamp(max=255).to(torch.uint8)


def stats(label, g, axis):
    # tile = 8 consecutive x, the whole `axis`, one index of the remaining dimension
    t = g.movedim(axis, 0)  # [axis, other, x]
    n, o, w = t.shape
    w8 = (w // 8) * 8
    tiles = t[:, :, :w8].reshape(n, o, w8 // 8, 8)
    gmax = tiles.amax((0, 3)).flatten().float()
    gmin = tiles.amin((0, 3)).flatten().float()
    print("%s: input mean %.1f, cells == 255: %.1f %%;  tiles: all 255: %.1f %%, gmax < 16: %.1f %%, < 32: %.1f %%, < 64: %.1f %%, < 128: %.1f %%, == 255 (mixed): %.1f %%" % (
        label, g.float().mean().item(), 100.0 * (g == 255).float().mean().item(), 100.0 * (gmin == 255).float().mean().item(),
        100.0 * (gmax < 16).float().mean().item(), 100.0 * (gmax < 32).float().mean().item(), 100.0 * (gmax < 64).float().mean().item(),
        100.0 * (gmax < 128).float().mean().item(), 100.0 * ((gmax == 255) & (gmin < 255)).float().mean().item()))


dx = one_d(occ, 2)
stats("y pass input (after x)", dx, 1)
dxy = one_d(dx, 1)
stats("z pass input (after x, y)", dxy, 0)
dxyz = one_d(dxy, 0)
m = occ.clone()
sw = torch.empty_like(occ)
ctx.distance_map(m.data_ptr(), sw.data_ptr(), v.map_extent, st)
torch.cuda.synchronize()
print("brute force == vkv_distance_map:", bool(torch.equal(dxyz, m)), " result mean %.2f" % m.float().mean().item())
