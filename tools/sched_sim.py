"""Offline experiment (CPU, oracle): cost of different intra-wave scheduling policies on real per-ray event sequences.
Costs are in instruction units per wave-iteration: H = loop header, P = probe path, S = sample path, A = extra for alpha>0."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vkv_oracle as O
from vkvolume_amd import abi, camera
import math

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
W, H, D = int(1024 * scale), int(1024 * scale), int(795 * scale)
iw, ih = int(1920 * scale), int(1080 * scale)
vol = O.synth_volume((W, H, D), 1, 0xC0FFEE03)
opt = abi.VolumeOptions(intensity_min=0.1, gradient_min=0.0, gradient_max=0.2)
tf, tex = O.transfer_function_uniform(opt), O.transfer_function_texture(opt)
grad = O.gradient_map(vol, tf)
maps = O.compute_distance_map(vol, grad, tex, tf, 4, abi.SKIP_DISTANCE)
ext = abi.Extent3D(W, H, D); me = O.map_extent(ext, 4)
ixf = camera.image_transform((0.0003, 0.0003, 0.0007), (W, H, D), (1, 0, 0, 90)); node = camera.benchmark_node_transform(ixf)
m = (node.astype(np.float64).T @ ixf.astype(np.float64).T)[:3, :3]
radius = 1.5 * 0.5 * math.sqrt(sum(float(np.linalg.norm(m[:, i])) ** 2 for i in range(3)))
view, proj = camera.orbit_camera(0.0, 20.0, radius), camera.perspective_vulkan(60.0, iw / ih)
cam, rc, rg = O.build_uniforms(view, proj, node, ixf, 1.0, (iw, ih), ext, me)
p = abi.RenderParams(); p.camera, p.ray_cast, p.ray_gen, p.transfer_function = cam, rc, rg, tf
p.options = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
p.use_precomputed_gradient = 1; p.image_width, p.image_height = iw, ih
p.tiles = abi.full_frame_tiles(iw, ih); p.volume_extent, p.map_extent = ext, me
r = O.render(p, vol, grad, tex, maps)
ev = (r.counts[..., 0] + r.counts[..., 1]).astype(np.int64)
print("frame", iw, ih, "events", ev.sum(), "max", ev.max())
# pick the heaviest 8x8 tiles + a random sample of covered tiles
th, tw = ih // 8, iw // 8
tmax = ev[:th * 8, :tw * 8].reshape(th, 8, tw, 8).max(axis=(1, 3))
heavy = np.argsort(-tmax.ravel())[:40]
rng = np.random.default_rng(0)
covered = np.flatnonzero(tmax.ravel() > 0)
sample = np.concatenate([heavy, rng.choice(covered, 160, replace=False)])
L = O.lib(); L.vkvo_trace_ray.argtypes = [C.POINTER(abi.RenderParams), C.c_int, C.c_int, C.c_void_p, C.c_uint32]; L.vkvo_trace_ray.restype = C.c_uint32
pp = abi.RenderParams.from_buffer_copy(p)
keep = [np.ascontiguousarray(x) for x in (vol, grad, tex, maps[0])]
pp.d_volume, pp.d_gradient, pp.d_transfer_function = keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data
pp.d_distance_maps[0] = keep[3].ctypes.data
buf = np.zeros(4096, np.uint8)
def tile_seqs(t):
    ty, tx = divmod(int(t), tw)
    out = []
    for ly in range(8):
        for lx in range(8):
            n = L.vkvo_trace_ray(C.byref(pp), tx * 8 + lx, ty * 8 + ly, buf.ctypes.data, 4096)
            out.append(bytes(buf[:n]))
    return out
Hc, Pc, Sc, Ac = 25, 65, 120, 45
def cost_lockstep(seqs):
    n = max(len(s) for s in seqs); c = 0
    for k in range(n):
        evs = [s[k] for s in seqs if k < len(s)]
        c += Hc + (Pc if any(e in b"PO" for e in evs) else 0) + (Sc if any(e in b"SA" for e in evs) else 0) + (Ac if any(e == ord("A") for e in evs) else 0)
    return c, n
def cost_probe_inner(seqs):
    pos = [0] * len(seqs); c = 0; it = 0
    while any(pos[i] < len(s) for i, s in enumerate(seqs)):
        # inner: probe lanes hop until nobody wants a probe
        while any(pos[i] < len(s) and s[pos[i]] in b"PO" for i, s in enumerate(seqs)):
            c += Hc + Pc; it += 1
            for i, s in enumerate(seqs):
                if pos[i] < len(s) and s[pos[i]] in b"PO": pos[i] += 1
        if any(pos[i] < len(s) for i, s in enumerate(seqs)):
            evs = [s[pos[i]] for i, s in enumerate(seqs) if pos[i] < len(s)]
            c += Hc + Sc + (Ac if any(e == ord("A") for e in evs) else 0); it += 1
            for i, s in enumerate(seqs):
                if pos[i] < len(s): pos[i] += 1
    return c, it
def cost_majority(seqs):
    pos = [0] * len(seqs); c = 0; it = 0
    while True:
        want = [s[pos[i]] for i, s in enumerate(seqs) if pos[i] < len(s)]
        if not want: break
        npr = sum(1 for e in want if e in b"PO"); ns = len(want) - npr
        do_probe = npr > ns
        c += Hc + (Pc if do_probe else Sc + (Ac if any(e == ord("A") for e in want) else 0)); it += 1
        for i, s in enumerate(seqs):
            if pos[i] < len(s) and ((s[pos[i]] in b"PO") == do_probe): pos[i] += 1
    return c, it
def cost_sample2(seqs):
    """lockstep, but a sampling lane consumes up to 2 consecutive sample events per iteration (speculative second sample)"""
    pos = [0] * len(seqs); c = 0; it = 0
    while any(pos[i] < len(s) for i, s in enumerate(seqs)):
        evs = [s[pos[i]] for i, s in enumerate(seqs) if pos[i] < len(s)]
        anyp, anys = any(e in b"PO" for e in evs), any(e in b"SA" for e in evs)
        c += Hc + (Pc if anyp else 0) + ((Sc + 70) if anys else 0) + Ac * (1 if any(e == ord("A") for e in evs) else 0); it += 1
        for i, s in enumerate(seqs):
            if pos[i] < len(s):
                if s[pos[i]] in b"SA":
                    pos[i] += 1
                    if pos[i] < len(s) and s[pos[i]] in b"SA" and s[pos[i] - 1] != ord("X"): pos[i] += 1
                else: pos[i] += 1
    return c, it
def make_sample_k(K):
    def cost(seqs):
        """lockstep; a sampling lane consumes up to K consecutive sample events per iteration; the j-th sub-step costs S if any lane reaches it"""
        pos = [0] * len(seqs); c = 0; it = 0
        while any(pos[i] < len(s) for i, s in enumerate(seqs)):
            evs = [s[pos[i]] for i, s in enumerate(seqs) if pos[i] < len(s)]
            c += Hc + (Pc if any(e in b"PO" for e in evs) else 0); it += 1
            depth = [0] * K; occ = [False] * K
            for i, s in enumerate(seqs):
                if pos[i] >= len(s): continue
                if s[pos[i]] in b"PO":
                    pos[i] += 1; continue
                j = 0
                while j < K and pos[i] < len(s) and s[pos[i]] in b"SA":
                    depth[j] = 1; occ[j] = occ[j] or s[pos[i]] == ord("A"); pos[i] += 1; j += 1
            c += sum(Sc * d for d in depth) + sum(Ac for o in occ if o)
        return c, it
    return cost
policies = [("lockstep", cost_lockstep), ("sample2", make_sample_k(2)), ("sample3", make_sample_k(3)), ("sample4", make_sample_k(4)), ("sample6", make_sample_k(6))]
tot = {k: [0, 0] for k, _ in policies}
heavy_tot = {k: [0, 0] for k in tot}
for j, t in enumerate(sample):
    seqs = [s for s in tile_seqs(t) if len(s)]
    if not seqs: continue
    for name, fn in policies:
        c, it = fn(seqs)
        tot[name][0] += c; tot[name][1] += it
        if j < 40: heavy_tot[name][0] += c; heavy_tot[name][1] += it
    if j < 3:
        print("tile", t, "lens", sorted(len(s) for s in seqs)[-5:], "example", seqs[int(np.argmax([len(s) for s in seqs]))][:120].decode())
print("ALL sampled tiles  (cost units, iterations):", {k: tuple(v) for k, v in tot.items()})
print("40 HEAVIEST tiles  (cost units, iterations):", {k: tuple(v) for k, v in heavy_tot.items()})
