"""Experiment (GPU, round 6): HYBRID start orders - the heaviest p % of a frame's tiles first (by measured cost), the rest centre-first - against the
pure centre-first and the pure longest-first order, in bench.py's submission (6 frames per launch, 4 streams).  Derived from tile_order_feedback.py:
(the rest of this docstring is the parent's)
Experiment (GPU): start order of the tiles of a vkv_render_batch launch from MEASURED tile costs.
A first launch (centre-of-image-first order) is traced; every frame's tiles are then re-ordered longest first by the largest wave
iteration count of the tile's four waves (what a renderer could feed back from the previous frame rendered into the same target),
handed to the library through the diagnostic hook vkv_debug_tile_orders, and the same launch is timed with both orders.

    python tools/tile_order_feedback.py [frames per launch] [launches in flight on as many streams]
"""
import sys, os, ctypes as C, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from vkvolume_amd import abi, lib, volume as V
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True), (fw, fh))
L = lib.load()
L.vkv_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
L.vkv_debug_tile_orders.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
sets = []
for s in range(nstreams):
    bufs = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
    plist = []
    for j in range(n):
        q = sp.make_params(*views[j % 8])
        q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = bufs[j].data_ptr(), None, None, None, None, 0
        plist.append(q)
    sets.append((bufs, plist))
tiles = plist[0].tiles.tile_count
nblocks = ((tiles + 7) // 8) * 8 * n


def timed(reps=40):
    for _ in range(4):
        for s in range(nstreams):
            ctx.render_batch(sets[s][1], streams[s].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for s in range(nstreams):
            ctx.render_batch(sets[s][1], streams[s].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * nstreams * n) * 1e3


def traced():
    trace = torch.zeros((nblocks * 4, 10), dtype=torch.int64, device="cuda")
    L.vkv_debug_trace(ctx.handle, trace.data_ptr())
    ctx.render_batch(sets[0][1], streams[0].cuda_stream)
    torch.cuda.synchronize()
    L.vkv_debug_trace(ctx.handle, None)
    return trace.cpu().numpy()



tiles_x = (fw + 15) // 16
tid = np.arange(tiles)
cx, cy = ((tid % tiles_x) + 0.5) * 16 - 0.5 * fw, ((tid // tiles_x) + 0.5) * 16 - 0.5 * fh
centre = np.argsort(cx * cx + cy * cy, kind="stable")        # the library's default order
rank_in_centre = np.empty(tiles, np.int64); rank_in_centre[centre] = np.arange(tiles)
base_ms = min(timed() for _ in range(3))
t = traced()
live = t[:, 1] > 0
widx = np.nonzero(live)[0]
wg = widx // 4
f_of = (wg >> 3) % n
tile_of = (t[live, 3] & 0xffffffff).astype(np.int64)
cost = np.zeros((n, tiles), np.int64)
np.maximum.at(cost, (f_of, tile_of), t[live, 2].astype(np.int64))
ref = [b.clone() for b in sets[0][0]]
print("centre-first: %.4f ms per frame" % base_ms)


def run(name, orders):
    d = torch.from_numpy(np.stack(orders).astype(np.int64)).to(torch.int32).cuda().contiguous()
    L.vkv_debug_tile_orders(ctx.handle, d.data_ptr(), n, tiles)
    ms = min(timed() for _ in range(3))
    for a, b in zip(ref, sets[0][0]):
        assert torch.equal(a, b), "the start order changed a frame"
    print("%-60s %.4f ms per frame (%+.1f %%)" % (name, ms, (ms / base_ms - 1) * 100))


run("centre-first through the hook (check)", [centre for f in range(n)])
run("longest-first (the shipped feedback order)", [np.argsort(-cost[f], kind="stable") for f in range(n)])
for p in (1, 2, 5, 10, 20, 40):
    orders = []
    for f in range(n):
        k = max(1, tiles * p // 100)
        heavy = np.argsort(-cost[f], kind="stable")[:k]
        mask = np.ones(tiles, bool); mask[heavy] = False
        rest = centre[mask[centre]]
        orders.append(np.concatenate([heavy, rest]))
    run("heaviest %d %% first, the rest centre-first" % p, orders)
# heaviest first but in centre order among themselves (locality inside the heavy set)
for p in (5, 20):
    orders = []
    for f in range(n):
        k = max(1, tiles * p // 100)
        heavy = np.argsort(-cost[f], kind="stable")[:k]
        heavy = heavy[np.argsort(rank_in_centre[heavy], kind="stable")]
        mask = np.ones(tiles, bool); mask[heavy] = False
        orders.append(np.concatenate([heavy, centre[mask[centre]]]))
    run("heaviest %d %% first in centre order, the rest centre-first" % p, orders)
# covered tiles first (cost > 0) in centre order, empty tiles last
orders = []
for f in range(n):
    cov = centre[cost[f][centre] > 0]; emp = centre[cost[f][centre] == 0]
    orders.append(np.concatenate([cov, emp]))
run("covered tiles centre-first, empty tiles last", orders)
L.vkv_debug_tile_orders(ctx.handle, None, 0, 0)
