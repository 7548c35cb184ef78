"""Host-side mirror of the reference's operator interface for the hot path, on torch device tensors.

Same names, argument meaning and call order as the reference (LDeakin/VkVolume):

* ``Volume``                — src/volume_component.h:31-93 (resource owner + ``Options`` + TF texture)
* ``ComputeGradientMap``    — src/compute_gradient_map.h:38
* ``ComputeDistanceMap``    — src/compute_distance_map.h:38
* ``VolumeRenderSubpass``   — src/volume_render_subpass.h:58-88 (``Options``, ``draw``)

torch is plumbing only (device memory + the current HIP stream); every computation goes through the C ABI.
"""
import ctypes as C

import numpy as np
import torch

from . import abi, camera, lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Volume:
    """Owns the volume / gradient / transfer-function / distance-map device buffers
    (Volume::Image members, src/volume_component.h:87-90)."""

    def __init__(self, ctx, name="volume", device=None):
        self.ctx = ctx
        self.name = name
        self.device = torch.device("cuda", ctx.device) if device is None else device
        self.options = abi.VolumeOptions()  # Volume::Options defaults
        self.image_transform = np.eye(4, dtype=np.float32)
        self.node_transform = np.eye(4, dtype=np.float32)
        self.volume = self.gradient = self.transfer_function = None
        self.packed = self.transfer_function_bits = None  # device-internal accelerators (vkv_pack_volume, vkv_transfer_function_bits)
        self.use_packed = True
        self.distance_maps = []
        self.distance_map_swap = None
        self.extent = self.map_extent = None

    # -- load_from_file's device half (src/volume_component.cpp:55-153): take voxels, allocate images --
    def load_from_array(self, voxels_dhw, distance_map_block_size=4, image_transform=None):
        v = torch.as_tensor(np.ascontiguousarray(voxels_dhw, np.uint8)) if not torch.is_tensor(voxels_dhw) else voxels_dhw
        self.volume = v.to(self.device).contiguous()
        return self._allocate(distance_map_block_size, image_transform)

    def load_synthetic(self, extent_whd, kind, seed, distance_map_block_size=4, image_transform=None):
        w, h, d = extent_whd
        self.volume = torch.empty((d, h, w), dtype=torch.uint8, device=self.device)
        self.ctx.synth_volume(_ptr(self.volume), abi.Extent3D(w, h, d), kind, seed, _stream())
        return self._allocate(distance_map_block_size, image_transform)

    def _allocate(self, block, image_transform):
        d, h, w = self.volume.shape
        self.extent = abi.Extent3D(w, h, d)
        rnd = lambda x: (x + block - 1) // block  # noqa: E731  (volume_component.cpp:91-92)
        self.map_extent = abi.Extent3D(rnd(w), rnd(h), rnd(d))
        if self.options.use_precomputed_gradient:
            self.gradient = torch.zeros_like(self.volume)
        self.gradient_valid = False  # set by ComputeGradientMap.compute; the packed image must not be built from an empty map
        self.transfer_function = torch.zeros((256, 256, 4), dtype=torch.uint8, device=self.device)
        self.transfer_function_bits = torch.zeros(abi.TF_BITS_WORDS, dtype=torch.int32, device=self.device)
        self.packed = None
        self.distance_map_swap = torch.empty((self.map_extent.depth, self.map_extent.height, self.map_extent.width),
                                             dtype=torch.uint8, device=self.device)
        self.distance_maps = []
        if image_transform is not None:
            self.set_image_transform(image_transform)
        return True

    def set_image_transform(self, mat):
        self.image_transform = np.ascontiguousarray(mat, np.float32).reshape(4, 4)

    def set_number_of_distance_maps(self, n):
        """grow-only, re-creates all maps when growing (src/volume_component.cpp:155-184)"""
        if n <= len(self.distance_maps):
            return
        self.distance_maps = [torch.empty_like(self.distance_map_swap) for _ in range(n)]

    def get_transfer_function_uniform(self):
        return lib.transfer_function_uniform(self.options)

    def update_transfer_function_texture(self):
        tex = lib.transfer_function_texture(self.options)  # CPU builds the LUT (volume_component.cpp:242-261)
        self.transfer_function.copy_(torch.from_numpy(tex), non_blocking=False)
        # acceleration tables of the texture; the separable-product claim made with the uniform is checked on the device
        self.ctx.transfer_function_tables(_ptr(self.transfer_function), self.get_transfer_function_uniform(), _ptr(self.transfer_function_bits),
                                          _stream())

    def pack(self):
        """(Re)build the bricked sampling image from the linear volume (+ gradient map).  Call after the gradient map
        is computed — the counterpart of the driver's swizzle into an optimally tiled VkImage.  The packed image is a COPY:
        call pack() again after writing to ``volume`` / ``gradient`` in place (load_* and ComputeGradientMap do it themselves)."""
        if self.options.use_precomputed_gradient and not self.gradient_valid:
            raise RuntimeError("Volume.pack: use_precomputed_gradient is set but no gradient map has been computed "
                               "(run ComputeGradientMap.compute first)")
        n = self.ctx.packed_volume_bytes(self.extent)
        if self.packed is None or self.packed.numel() != n:
            self.packed = torch.empty(n, dtype=torch.uint8, device=self.device)
        grad = self.gradient if self.options.use_precomputed_gradient else None
        self.ctx.pack_volume(_ptr(self.volume), _ptr(grad), self.extent, _ptr(self.packed), _stream())


class ComputeGradientMap:
    def __init__(self, ctx):
        self.ctx = ctx

    def compute(self, volume, transfer_function_uniform):
        self.ctx.gradient_map(_ptr(volume.volume), _ptr(volume.gradient), volume.extent, transfer_function_uniform, _stream())
        volume.gradient_valid = True
        if volume.use_packed:
            volume.pack()


class ComputeDistanceMap:
    def __init__(self, ctx):
        self.ctx = ctx

    def compute(self, volume, transfer_function_uniform, skipping_type):
        n = 8 if skipping_type == abi.SKIP_ANISOTROPIC_DISTANCE else 1
        volume.set_number_of_distance_maps(n)
        grad = volume.gradient if volume.options.use_precomputed_gradient else None
        self.ctx.compute_distance_map(_ptr(volume.volume), _ptr(grad), _ptr(volume.transfer_function), transfer_function_uniform,
                                      volume.extent, [_ptr(m) for m in volume.distance_maps], _ptr(volume.distance_map_swap),
                                      volume.map_extent, skipping_type, _stream())


class VolumeRenderSubpass:
    """Offscreen counterpart of the reference subpass: ``draw`` renders one volume into caller-provided buffers."""

    Options = abi.RenderOptions

    def __init__(self, ctx, volume, options=None, image_size=(256, 256)):
        self.ctx, self.volume = ctx, volume
        self.options = options if options is not None else abi.RenderOptions()
        self.image_size = image_size

    def make_params(self, view, proj, tiles=None, uniforms=None):
        """``uniforms``: an already built (CameraUniform, RayCastUniform, RayGen) triple instead of vkv_build_uniforms' (tests)."""
        v = self.volume
        w, h = self.image_size
        cam, rc, rg = uniforms if uniforms is not None else lib.build_uniforms(
            view, proj, v.node_transform, v.image_transform, self.options.clip_distance, (w, h), v.extent, v.map_extent)
        p = abi.RenderParams()
        p.camera, p.ray_cast, p.ray_gen = cam, rc, rg
        p.transfer_function = v.get_transfer_function_uniform()
        p.options = self.options
        p.use_precomputed_gradient = v.options.use_precomputed_gradient
        p.image_width, p.image_height = w, h
        p.tiles = tiles if tiles is not None else abi.full_frame_tiles(w, h)
        p.volume_extent, p.map_extent = v.extent, v.map_extent
        return self.bind(p)

    def bind(self, params):
        """Copy ``params`` and point it at this volume's device buffers (what draw() binds at
        src/volume_render_subpass.cpp:263-284)."""
        v = self.volume
        p = abi.RenderParams.from_buffer_copy(params)
        p.d_volume = _ptr(v.volume)
        p.d_gradient = _ptr(v.gradient) if v.options.use_precomputed_gradient else None
        p.d_transfer_function = _ptr(v.transfer_function)
        for i in range(8):
            p.d_distance_maps[i] = _ptr(v.distance_maps[i]) if i < len(v.distance_maps) else None
        if v.use_packed and v.packed is None:
            v.pack()  # no gradient pass ran (on-the-fly / no-gradient variants): pack the volume channel alone
        p.d_packed_volume = _ptr(v.packed) if v.use_packed else None
        p.d_transfer_function_bits = _ptr(v.transfer_function_bits) if v.use_packed else None
        return p

    def draw(self, params, color=None, rgba8=None, counts=None, depth=None, in_depth=None, blend=False):
        """``in_depth``: scene depth for options.depth_attachment; ``blend``: blend onto the contents of color / rgba8
        (the subpass's blend state) instead of overwriting them."""
        params.d_out_color, params.d_out_rgba8 = _ptr(color), _ptr(rgba8)
        params.d_out_counts, params.d_out_depth = _ptr(counts), _ptr(depth)
        params.d_in_depth, params.blend_over_target = _ptr(in_depth), 1 if blend else 0
        self.ctx.render(params, _stream())


def default_scene(volume, voxel_size=(1.0, 1.0, 1.0), axis_angle=(1.0, 0.0, 0.0, 0.0)):
    """image transform from the header fields + benchmark-mode node scale (src/load_volume.cpp:82-83,
    src/volume_render.cpp:224-238)."""
    e = volume.extent
    volume.set_image_transform(camera.image_transform(voxel_size, (e.width, e.height, e.depth), axis_angle))
    volume.node_transform = camera.benchmark_node_transform(volume.image_transform)
