"""Shared scene construction for the parity tests: the same call order as the reference's
VolumeRender::prepare (src/volume_render.cpp:186-238): load -> gradient -> TF texture -> occupancy/distance -> render."""
import numpy as np

from oracle import vkv_oracle as O
from vkvolume_amd import abi, camera, lib

APP_TF = dict(intensity_min=0.1, intensity_max=1.0, gradient_min=0.0, gradient_max=0.2)  # src/volume_render.cpp:67-70


def random_volume(shape_whd, seed, sparsity=0.0):
    """Random bytes with some smooth structure; `sparsity` = fraction of voxels forced below the app TF threshold."""
    w, h, d = shape_whd
    rng = np.random.default_rng(seed)
    v = rng.integers(0, 256, size=(d, h, w), dtype=np.uint8)
    if sparsity > 0:
        mask = rng.random((d, h, w)) < sparsity
        v[mask] = rng.integers(0, 20, size=int(mask.sum()), dtype=np.uint8)
    return v


class OracleScene:
    """CPU-side scene evaluated with the oracle (the expected values of every parity test)."""

    def __init__(self, vol_dhw, options, block=4, voxel_size=(1, 1, 1), axis_angle=(1, 0, 0, 0)):
        self.vol = np.ascontiguousarray(vol_dhw, np.uint8)
        d, h, w = self.vol.shape
        self.extent = abi.Extent3D(w, h, d)
        self.block = block
        self.map_extent = O.map_extent(self.extent, block)
        self.options = options
        self.tf = O.transfer_function_uniform(options)
        self.tex = O.transfer_function_texture(options)
        self.grad = O.gradient_map(self.vol, self.tf) if options.use_precomputed_gradient else None
        self.image_transform = camera.image_transform(voxel_size, (w, h, d), axis_angle)
        self.node_transform = camera.benchmark_node_transform(self.image_transform)
        self._maps = {}

    def maps(self, skipping_type):
        if skipping_type not in self._maps:
            self._maps[skipping_type] = O.compute_distance_map(self.vol, self.grad, self.tex, self.tf, self.block, skipping_type)
        return self._maps[skipping_type]

    def params(self, view, proj, image_size, render_options, tiles=None, uniforms=None):
        w, h = image_size
        if uniforms is None:
            uniforms = lib.build_uniforms(view, proj, self.node_transform, self.image_transform, render_options.clip_distance,
                                          (w, h), self.extent, self.map_extent)
        p = abi.RenderParams()
        p.camera, p.ray_cast, p.ray_gen = uniforms
        p.transfer_function = self.tf
        p.options = render_options
        p.use_precomputed_gradient = self.options.use_precomputed_gradient
        p.image_width, p.image_height = w, h
        p.tiles = tiles if tiles is not None else abi.full_frame_tiles(w, h)
        p.volume_extent, p.map_extent = self.extent, self.map_extent
        return p

    def render(self, params, **kw):
        st = params.options.skipping_type
        maps = None if st == abi.SKIP_NONE else self.maps(st)
        return O.render(params, self.vol, self.grad, self.tex, maps, **kw)


def orbit(azimuth, elevation=20.0, radius=150.0, fov=60.0, image_size=(256, 256)):
    view = camera.orbit_camera(azimuth, elevation, radius)
    proj = camera.perspective_vulkan(fov, image_size[0] / image_size[1])
    return view, proj


def brute_force_chebyshev(occ):
    """min(255, Chebyshev distance to the nearest occupied (== 0) cell); SURVEY.md §4 KAT 1."""
    occ = np.asarray(occ)
    pts = np.argwhere(occ == 0)
    out = np.full(occ.shape, 255, np.int64)
    if len(pts) == 0:
        return out.astype(np.uint8)
    zz, yy, xx = np.indices(occ.shape)
    for (z, y, x) in pts:
        d = np.maximum(np.maximum(np.abs(zz - z), np.abs(yy - y)), np.abs(xx - x))
        out = np.minimum(out, d)
    return np.minimum(out, 255).astype(np.uint8)


def brute_force_chebyshev_octant(occ, k):
    """One-sided KAT: nearest occupied cell o with s*(o - c) >= 0 per axis, s = octant signs of map index k
    (k = (dz<0) + 2(dy<0) + 4(dx<0)); SURVEY.md §4 KAT 1."""
    sz = -1 if (k & 1) else 1
    sy = -1 if (k & 2) else 1
    sx = -1 if (k & 4) else 1
    occ = np.asarray(occ)
    pts = np.argwhere(occ == 0)
    out = np.full(occ.shape, 255, np.int64)
    zz, yy, xx = np.indices(occ.shape)
    for (z, y, x) in pts:
        dz, dy, dx = sz * (z - zz), sy * (y - yy), sx * (x - xx)
        ok = (dz >= 0) & (dy >= 0) & (dx >= 0)
        d = np.maximum(np.maximum(dz, dy), dx)
        out = np.where(ok, np.minimum(out, d), out)
    return np.minimum(out, 255).astype(np.uint8)
