"""Offline experiment (CPU, oracle): wave-iterations of the ray-march under "window look-ahead" scheduling.

Static scheduling: one lane marches one ray, an 8x8 wave runs max_r(events of ray r) iterations.
Look-ahead: in every wave iteration the 64 lanes are shared out over the LIVE rays of the wave; a ray that gets w lanes has the
w loop positions i0 .. i0+w-1 evaluated in parallel (sample + probe of each position), then its owner lane replays the frag's
state machine over those entries in order and consumes every event whose position lies inside the window.  The event sequence of
a ray is unchanged (same counters, same blend order); only the number of dependent memory round trips shrinks.

usage: lookahead_sim.py [scale] [tile_stride]   (scene arrays from /tmp/sim/*.npy, built by tools/build_scene_cpu.py)
"""
import sys, os, math, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vkv_oracle as O
from vkvolume_amd import abi, camera

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
tstride = int(sys.argv[2]) if len(sys.argv) > 2 else 4
views = [float(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0.0, 135.0]
W, H, D = int(1024 * scale), int(1024 * scale), int(795 * scale)
iw, ih = int(1920 * scale), int(1080 * scale)
vol, grad, maps = (np.load("/tmp/sim/%s_%g.npy" % (n, scale), mmap_mode="r") for n in ("vol", "grad", "maps"))
vol, grad, maps = np.ascontiguousarray(vol), np.ascontiguousarray(grad), np.ascontiguousarray(maps)
opt = abi.VolumeOptions(intensity_min=0.1, intensity_max=1.0, gradient_min=0.0, gradient_max=0.2)
tf, tex = O.transfer_function_uniform(opt), O.transfer_function_texture(opt)
ext = abi.Extent3D(W, H, D); me = O.map_extent(ext, 4)
ixf = camera.image_transform((0.0003, 0.0003, 0.0007), (W, H, D), (1, 0, 0, 90)); node = camera.benchmark_node_transform(ixf)
m = (node.astype(np.float64).T @ ixf.astype(np.float64).T)[:3, :3]
radius = 1.5 * 0.5 * math.sqrt(sum(float(np.linalg.norm(m[:, i])) ** 2 for i in range(3)))


def simulate(rays, policy, cap=64):
    """rays: list of step arrays (one per live ray of the wave).  Returns (iterations, lane evaluations, consumed, sum of longest replay)."""
    ptr = [0] * len(rays)
    last = [1] * len(rays)       # events consumed in the previous window (adaptive policy)
    iters = evals = used = replay = 0
    live = [r for r in range(len(rays)) if len(rays[r])]
    while live:
        L = len(live)
        if policy == "static":
            w = {r: 1 for r in live}
        elif policy == "fixed":
            w = {r: 64 // len(rays) for r in live}
        elif policy == "equal":
            base, rem = divmod(64, L)
            w = {r: min(cap, base + (1 if k < rem else 0)) for k, r in enumerate(live)}
        else:  # adaptive: ask for twice what the last window consumed, scale down to 64 lanes
            want = {r: min(cap, max(1, 2 * last[r])) for r in live}
            tot = sum(want.values())
            if tot > 64:
                f = 64.0 / tot
                want = {r: max(1, int(want[r] * f)) for r in live}
                while sum(want.values()) > 64:
                    k = max(want, key=want.get); want[k] -= 1
            w = want
        longest = 0
        for r in live:
            st = rays[r]; p = ptr[r]; i0 = st[p]; n = 0
            while p < len(st) and i0 <= st[p] < i0 + w[r]:
                p += 1; n += 1
            ptr[r] = p; last[r] = n; used += n; longest = max(longest, n)
        evals += sum(w.values()); iters += 1; replay += longest
        live = [r for r in live if ptr[r] < len(rays[r])]
    return iters, evals, used, replay


for az in views:
    view, proj = camera.orbit_camera(az, 20.0, radius), camera.perspective_vulkan(60.0, iw / ih, 0.1, 1000.0)
    cam, rc, rg = O.build_uniforms(view, proj, node, ixf, 1.0, (iw, ih), ext, me)
    p = abi.RenderParams(); p.camera, p.ray_cast, p.ray_gen, p.transfer_function = cam, rc, rg, tf
    p.options = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0, early_ray_termination=1)
    p.use_precomputed_gradient = 1; p.image_width, p.image_height = iw, ih
    p.tiles = abi.full_frame_tiles(iw, ih); p.volume_extent, p.map_extent = ext, me
    t0 = time.time()
    tot = {k: np.zeros(4, np.int64) for k in ("static", "equal", "fixed/32", "fixed/16", "fixed/8", "equal/32", "equal/16", "equal/8")}
    worst = {k: 0 for k in tot}
    kinds = np.zeros(256, np.int64)
    true_static = [0, 0]
    runs = []
    ntiles = 0
    for ty in range(0, ih // 8, tstride):
        for tx in range((ty // tstride) % tstride, iw // 8, tstride):
            rays = []
            for ly in range(8):
                for lx in range(8):
                    ev, st = O.trace_ray_steps(p, vol, grad, tex, maps, tx * 8 + lx, ty * 8 + ly)
                    rays.append(st.tolist())
                    if len(ev):
                        kinds += np.bincount(ev, minlength=256)
            if not any(len(x) for x in rays):
                continue
            ntiles += 1
            true_static[0] += max(len(x) for x in rays); true_static[1] = max(true_static[1], max(len(x) for x in rays))
            for k in tot:
                pol, rpw = (k.split("/")[0], int(k.split("/")[1])) if "/" in k else (k, 64)
                # rpw rays per wave: the 8x8 tile is cut into 64/rpw sub-tiles (rows of the tile), one wave each
                for g in range(0, 64, rpw):
                    sub = [rays[i] for i in range(g, min(g + rpw, 64))]
                    if not any(len(x) for x in sub):
                        continue
                    r = simulate(sub, pol, 64)
                    tot[k] += r; worst[k] = max(worst[k], r[0])
    print("az %g: %d tiles traced in %.0f s; events P/O/S/A = %d/%d/%d/%d" % (az, ntiles, time.time() - t0, kinds[ord('P')], kinds[ord('O')], kinds[ord('S')], kinds[ord('A')]))
    print("  one lane per ray, one event per iteration (round 1 kernel): wave-iterations %d (longest wave %d)" % tuple(true_static))
    for k, v in tot.items():
        print("  %-10s wave-iterations %8d (longest wave %4d)  lane evaluations %9d  consumed %9d  sum of longest replay %8d" % (k, v[0], worst[k], v[1], v[2], v[3]))
