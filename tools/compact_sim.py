"""Offline experiment (CPU, oracle): how many wave-iterations would lane compaction save?  Compares the static 8x8 waves with an
ideal repack of the active rays of a 16x16 workgroup every K iterations, on per-ray event counts of the half-scale C3 scene."""
import sys, os, math, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vkv_oracle as O
from vkvolume_amd import abi, camera
scale = 0.5
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
tot = {}
for az in (0.0, 45.0, 90.0):
    view, proj = camera.orbit_camera(az, 20.0, radius), camera.perspective_vulkan(60.0, iw / ih)
    cam, rc, rg = O.build_uniforms(view, proj, node, ixf, 1.0, (iw, ih), ext, me)
    p = abi.RenderParams(); p.camera, p.ray_cast, p.ray_gen, p.transfer_function = cam, rc, rg, tf
    p.options = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0, early_ray_termination=1)
    p.use_precomputed_gradient = 1; p.image_width, p.image_height = iw, ih
    p.tiles = abi.full_frame_tiles(iw, ih); p.volume_extent, p.map_extent = ext, me
    r = O.render(p, vol, grad, tex, maps)
    ev = (r.counts[..., 0] + r.counts[..., 1]).astype(np.int64)
    hh, ww = (ih // 16) * 16, (iw // 16) * 16
    e = ev[:hh, :ww]
    # static 8x8 waves
    w8 = e.reshape(hh // 8, 8, ww // 8, 8).max(axis=(1, 3)).sum()
    ideal = int(np.ceil(e.sum() / 64))
    # block-level (16x16) repack: sum_t ceil(active(t)/64), repack only every K iterations
    b = e.reshape(hh // 16, 16, ww // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 256)
    res = {}
    for K in (1, 8, 16, 32):
        total = 0
        for rays in b:
            mx = rays.max()
            if mx == 0: continue
            t = 0
            s = np.sort(rays)
            while t < mx:
                active = 256 - np.searchsorted(s, t, side='right')   # rays with n > t
                waves = -(-active // 64)
                # run K iterations (or until those rays end): cost = waves * min(K, remaining of the packed waves' longest)
                step = min(K, mx - t)
                total += waves * step
                t += step
        res[K] = total
    # wave-level (8x8) early exit already counted in w8 (a wave ends when its longest ray ends)
    print("az", az, "events", e.sum(), "static 8x8 wave-iters", w8, "ideal", ideal, "block repack", res)
