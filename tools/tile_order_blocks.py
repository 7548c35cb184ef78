"""Experiment (GPU): candidate tile start orders in the driver's setting - fenced blocks of 20 frames, launches of 7 + 7 + 6 frames on
three streams - with the orders handed in through vkv_debug_tile_orders.  Costs come from a traced launch per stream set."""
import sys, os, ctypes as C, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VKV_RAYMARCH_FEEDBACK", "0")
import bench
from vkvolume_amd import abi, lib, volume as V
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=not os.environ.get("NO_ERT")), (fw, fh))
L = lib.load()
L.vkv_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
L.vkv_debug_tile_orders.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
sizes = [7, 7, 6]
streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
sets, k = [], 0
for s, n in enumerate(sizes):
    bufs = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(n)]
    plist = []
    for j in range(n):
        q = sp.make_params(*views[(k + j) % 8])
        q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = bufs[j].data_ptr(), None, None, None, None, 0
        plist.append(q)
    sets.append((bufs, plist, [(k + j) % 8 for j in range(n)]))
    k += n
tiles = sets[0][1][0].tiles.tile_count


def block():
    for s in range(3):
        ctx.render_batch(sets[s][1], streams[s].cuda_stream)
    torch.cuda.synchronize()


def timed(orders_by_view, reps=150):
    # the hook holds one order table for "frame i of the launch": launches differ, so set it per launch
    tabs = None
    if orders_by_view is not None:
        tabs = [torch.from_numpy(np.stack([orders_by_view[vw] for vw in sets[s][2]]).astype(np.int64)).to(torch.int32).cuda().contiguous() for s in range(3)]

    def run():
        for s in range(3):
            if tabs is not None:
                L.vkv_debug_tile_orders(ctx.handle, tabs[s].data_ptr(), len(sets[s][2]), tiles)
            ctx.render_batch(sets[s][1], streams[s].cuda_stream)
        torch.cuda.synchronize()
    for _ in range(5):
        run()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    dt = (time.perf_counter() - t0) / (reps * 20) * 1e3
    L.vkv_debug_tile_orders(ctx.handle, None, 0, 0)
    return dt


# costs per view from traced 8-frame launches of the 8 views
bufs8 = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(8)]
for _t in bufs8:
    ctx.register_target(_t.data_ptr(), (fw, fh), abi.full_frame_tiles(fw, fh))  # feedback state of the targets (the library-feedback rows)
p8 = []
for j in range(8):
    q = sp.make_params(*views[j])
    q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = bufs8[j].data_ptr(), None, None, None, None, 0
    p8.append(q)
nblocks = ((tiles + 7) // 8) * 8 * 8
trace = torch.zeros((nblocks * 4, 10), dtype=torch.int64, device="cuda")
ctx.render_batch(p8, streams[0].cuda_stream); torch.cuda.synchronize()
L.vkv_debug_trace(ctx.handle, trace.data_ptr())
ctx.render_batch(p8, streams[0].cuda_stream); torch.cuda.synchronize()
L.vkv_debug_trace(ctx.handle, None)
t = trace.cpu().numpy(); live = t[:, 1] > 0
widx = np.nonzero(live)[0]; f_of = ((widx // 4) >> 3) % 8; tile_of = (t[live, 3] & 0xffffffff).astype(np.int64)
cost = np.zeros((8, tiles), np.int64)
np.maximum.at(cost, (f_of, tile_of), t[live, 2].astype(np.int64))

tx, ty = (fw + 15) // 16, (fh + 15) // 16
cx = (np.arange(tiles) % tx + 0.5) * 16 - 0.5 * fw; cy = (np.arange(tiles) // tx + 0.5) * 16 - 0.5 * fh
centre = np.argsort(cx * cx + cy * cy, kind="stable")


def lpt(vw):
    return np.argsort(-cost[vw], kind="stable")


def mixed(vw, heavy_per_group=8):
    o = lpt(vw)
    nh = int((cost[vw] > 0).sum())
    heavy, light = o[:nh], o[nh:]
    per = max(8, int(round(len(light) / max(nh, 1) * heavy_per_group / 8.0)) * 8)
    out, h, l = [], 0, 0
    while h < nh or l < len(light):
        out.extend(heavy[h:h + heavy_per_group]); h += heavy_per_group
        out.extend(light[l:l + per]); l += per
    return np.array(out[:tiles])


def lpt_band(vw, bands=4):
    # longest first inside bands: the order visits cost quantiles round-robin (heavy, medium, light, heavy, ...), in groups of 8
    o = lpt(vw)
    nh = int((cost[vw] > 0).sum())
    heavy, light = o[:nh], o[nh:]
    parts = np.array_split(heavy, bands)
    out, idx = [], [0] * bands
    while any(idx[b] < len(parts[b]) for b in range(bands)):
        for b in range(bands):
            out.extend(parts[b][idx[b]:idx[b] + 8]); idx[b] += 8
    return np.concatenate([np.array(out, dtype=np.int64), light])


def lpt_random_ties(vw, seed=1):
    rng = np.random.default_rng(seed + vw)
    o = lpt(vw)
    nh = int((cost[vw] > 0).sum())
    light = o[nh:].copy(); rng.shuffle(light)
    return np.concatenate([o[:nh], light])


def lpt_centre_ties(vw):
    # ties (and the tiles without a marching ray) in centre-first order, as a stable sort of the library's order gives
    return centre[np.argsort(-cost[vw][centre], kind="stable")]


def classes(vw, per_octave=1):
    # coarse cost classes (log2 scale), the library's centre-first order kept inside a class: longest first only roughly, spatially coherent
    c = cost[vw][centre].astype(np.float64)
    cls = np.where(c > 0, np.floor(np.log2(np.maximum(c, 1)) * per_octave) + 1, 0)
    return centre[np.argsort(-cls, kind="stable")]


def xcd_groups(vw, key):
    """Position p of the order runs on XCD p & 7 (own L2): give every XCD ONE spatial group of tiles - the tiles sorted by `key` (an angle
    around the screen centre, or x) and cut into 8 runs of equal total cost - longest first inside the group."""
    o = np.argsort(key, kind="stable")
    c = np.maximum(cost[vw][o], 1).astype(np.float64)        # tiles without a marching ray still cost a launch
    cuts = np.searchsorted(np.cumsum(c), np.linspace(0, c.sum(), 9)[1:-1])
    groups = np.split(o, cuts)
    groups = [g[np.argsort(-cost[vw][g], kind="stable")] for g in groups]
    n = max(len(g) for g in groups)
    out = np.full((n, 8), -1, np.int64)
    for x, g in enumerate(groups):
        out[:len(g), x] = g
    flat = out.reshape(-1)
    # the kernel needs a permutation of all tiles in `tiles` positions: holes of shorter groups are filled from the longest groups' tails
    spare = [t for t in flat[tiles:] if t >= 0]
    flat = flat[:tiles].copy()
    holes = np.nonzero(flat < 0)[0]
    assert len(holes) == len(spare), (len(holes), len(spare))
    flat[holes] = spare
    assert len(set(flat.tolist())) == tiles
    return flat


angle = np.arctan2(cy, cx)
print("fenced blocks of 20 frames (7 + 7 + 6 on three streams), ms per frame:")
if os.environ.get("XCD_GROUPS"):
    print("  longest first, ties centre-first      %.4f" % timed([lpt_centre_ties(vw) for vw in range(8)]))
    print("  one angular sector per XCD            %.4f" % timed([xcd_groups(vw, angle) for vw in range(8)]))
    print("  one vertical stripe per XCD           %.4f" % timed([xcd_groups(vw, cx) for vw in range(8)]))
    print("  one horizontal stripe per XCD         %.4f" % timed([xcd_groups(vw, cy) for vw in range(8)]))
    print("  longest first, ties centre-first      %.4f" % timed([lpt_centre_ties(vw) for vw in range(8)]))
    sys.exit(0)
print("  log2 cost classes, centre-first inside  %.4f" % timed([classes(vw) for vw in range(8)]))
print("  half-octave classes                     %.4f" % timed([classes(vw, 2) for vw in range(8)]))
print("  quarter-octave classes                  %.4f" % timed([classes(vw, 4) for vw in range(8)]))
print("  longest first, empty tiles shuffled   %.4f" % timed([lpt_random_ties(vw) for vw in range(8)]))
print("  longest first, ties centre-first      %.4f" % timed([lpt_centre_ties(vw) for vw in range(8)]))
print("  centre first (library)        %.4f" % timed(None))
print("  centre first (via the hook)   %.4f" % timed([centre] * 8))
print("  longest first (measured)      %.4f" % timed([lpt(vw) for vw in range(8)]))
print("  longest first + light mixed   %.4f" % timed([mixed(vw) for vw in range(8)]))
print("  longest first, 16 heavy/group %.4f" % timed([mixed(vw, 16) for vw in range(8)]))
print("  cost quantiles round-robin    %.4f" % timed([lpt_band(vw) for vw in range(8)]))
print("  plain order                   %.4f" % timed([np.arange(tiles)] * 8))
