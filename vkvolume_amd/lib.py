"""ctypes binding of ``csrc/libvkvolume_amd.so`` (the C ABI of ``include/vkvolume_amd.h``).

There is no fallback: if the shared library is missing or a call fails, a ``VkvError`` is raised.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VKV_LIB_PATH") or os.path.join(_HERE, "csrc", "libvkvolume_amd.so")  # override: kernel experiments only

# every symbol include/vkvolume_amd.h declares
EXPORTS = [
    "vkv_create", "vkv_destroy", "vkv_last_error", "vkv_version",
    "vkv_transfer_function_uniform", "vkv_transfer_function_texture", "vkv_build_uniforms",
    "vkv_gradient_map", "vkv_occupancy_map", "vkv_distance_map", "vkv_distance_map_anisotropic",
    "vkv_compute_distance_map", "vkv_render", "vkv_render_batch", "vkv_scatter_tiles", "vkv_synth_volume",
    "vkv_packed_volume_bytes", "vkv_pack_volume", "vkv_transfer_function_bits", "vkv_transfer_function_tables",
    "vkv_occupied_voxel_count", "vkv_load_header", "vkv_load_data", "vkv_convert_volume", "vkv_gather_tiles", "vkv_assemble_frame",
    "vkv_assemble_frames", "vkv_get_tuning", "vkv_set_tuning", "vkv_prepare_render", "vkv_register_target", "vkv_forget_target",
    "vkv_release_stream", "vkv_trim", "vkv_release_captured", "vkv_screen_tile_rect",
]
# include/vkvolume_amd_debug.h (diagnostics: tools/ and the exhaustive numerics tests)
DEBUG_EXPORTS = ["vkv_debug_trace", "vkv_debug_tile_orders", "vkv_debug_check"]


class VkvError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("vkvolume_amd error %d: %s" % (code, message))
        self.code = code


_LIB = None


def load():
    """Load the HIP library (once).  Raises VkvError if it has not been built — there is no CPU path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise VkvError(abi.VKV_E_NO_DEVICE, "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # torch bundles its own HIP runtime (same soname as /opt/rocm's).  It must be the one already mapped when this
    # library is dlopen()ed, so that streams and device pointers handed over from torch belong to the same runtime.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    P = C.POINTER
    L.vkv_create.argtypes = [C.c_int, P(vp)]
    L.vkv_destroy.argtypes = [vp]
    L.vkv_destroy.restype = None
    L.vkv_last_error.argtypes = [vp]
    L.vkv_last_error.restype = C.c_char_p
    L.vkv_version.restype = C.c_char_p
    L.vkv_transfer_function_uniform.argtypes = [P(abi.VolumeOptions), P(abi.TransferFunctionUniform)]
    L.vkv_transfer_function_texture.argtypes = [P(abi.VolumeOptions), vp]
    L.vkv_build_uniforms.argtypes = [vp, vp, vp, vp, C.c_float, u32, u32, abi.Extent3D, abi.Extent3D,
                                     P(abi.CameraUniform), P(abi.RayCastUniform), P(abi.RayGen)]
    L.vkv_gradient_map.argtypes = [vp, vp, vp, abi.Extent3D, P(abi.TransferFunctionUniform), vp]
    L.vkv_occupancy_map.argtypes = [vp, vp, vp, vp, P(abi.TransferFunctionUniform), abi.Extent3D, vp, abi.Extent3D, vp]
    L.vkv_distance_map.argtypes = [vp, vp, vp, abi.Extent3D, vp]
    L.vkv_distance_map_anisotropic.argtypes = [vp, P(vp), vp, abi.Extent3D, vp]
    L.vkv_compute_distance_map.argtypes = [vp, vp, vp, vp, P(abi.TransferFunctionUniform), abi.Extent3D, P(vp), vp,
                                           abi.Extent3D, i32, vp]
    L.vkv_render.argtypes = [vp, P(abi.RenderParams), vp]
    L.vkv_render_batch.argtypes = [vp, P(abi.RenderParams), C.c_uint32, vp]
    L.vkv_gather_tiles.argtypes = [vp, vp, vp, C.c_size_t, i32, vp, vp]
    L.vkv_assemble_frame.argtypes = [vp, vp, vp, vp, u32, u32, u32, u32, P(abi.TileRect), u32, u32, u32, i32, vp, vp]
    L.vkv_scatter_tiles.argtypes = [vp, vp, vp, u32, u32, u32, u32, P(abi.TileRect), u32, u32, u32, vp]
    L.vkv_assemble_frames.argtypes = [vp, vp, vp, P(vp), u32, u32, u32, u32, u32, P(abi.TileRect), u32, u32, u32, i32, P(i32), vp, vp]
    L.vkv_screen_tile_rect.argtypes = [P(abi.RayCastUniform), P(abi.RayGen), u32, u32, u32, u32, u32, P(abi.TileRect)]
    L.vkv_get_tuning.argtypes = [vp, P(abi.Tuning)]
    L.vkv_set_tuning.argtypes = [vp, P(abi.Tuning)]
    L.vkv_prepare_render.argtypes = [vp, P(abi.RenderParams), u32, vp]
    L.vkv_register_target.argtypes = [vp, vp, u32, u32, P(abi.TileSchedule)]
    L.vkv_forget_target.argtypes = [vp, vp]
    L.vkv_release_stream.argtypes = [vp, vp]
    L.vkv_trim.argtypes = [vp]
    L.vkv_release_captured.argtypes = [vp, vp]
    L.vkv_debug_trace.argtypes = [vp, vp]
    L.vkv_debug_tile_orders.argtypes = [vp, vp, u32, u32]
    L.vkv_debug_check.argtypes = [vp, i32, u32, C.c_uint64, vp, vp]
    L.vkv_synth_volume.argtypes = [vp, vp, abi.Extent3D, u32, u32, vp]
    L.vkv_packed_volume_bytes.argtypes = [abi.Extent3D]
    L.vkv_packed_volume_bytes.restype = C.c_size_t
    L.vkv_pack_volume.argtypes = [vp, vp, vp, abi.Extent3D, vp, vp]
    L.vkv_transfer_function_bits.argtypes = [vp, vp, vp, vp]
    L.vkv_transfer_function_tables.argtypes = [vp, vp, P(abi.TransferFunctionUniform), vp, vp]
    L.vkv_occupied_voxel_count.argtypes = [vp, vp, vp, P(abi.TransferFunctionUniform), abi.Extent3D, vp, vp]
    L.vkv_convert_volume.argtypes = [vp, vp, i32, i32, C.c_float, C.c_float, C.c_uint64, vp, vp]
    L.vkv_load_header.argtypes = [C.c_char_p, P(abi.VolumeHeader)]
    L.vkv_load_data.argtypes = [C.c_char_p, P(abi.VolumeHeader), vp, C.c_size_t]
    for name in EXPORTS + DEBUG_EXPORTS:
        getattr(L, name)  # AttributeError here means the library does not export what the headers declare
    _LIB = L
    return L


class Context:
    """RAII wrapper of ``vkv_ctx`` for one device."""

    def __init__(self, device=0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.vkv_create(int(device), C.byref(h))
        if rc != 0:
            raise VkvError(rc, "vkv_create(device=%d) failed (no gfx950 device?)" % device)
        self.handle = h
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            self._lib.vkv_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise VkvError(rc, self._lib.vkv_last_error(self.handle).decode(errors="replace"))

    # ---- device entry points (pointers are ints / None; stream is a hipStream_t handle as int) ----
    def gradient_map(self, d_volume, d_gradient, extent, tf, stream=0):
        self.check(self._lib.vkv_gradient_map(self.handle, d_volume, d_gradient, extent, C.byref(tf), stream))

    def occupancy_map(self, d_volume, d_gradient, d_tf, tf, extent, d_map, map_extent, stream=0):
        self.check(self._lib.vkv_occupancy_map(self.handle, d_volume, d_gradient, d_tf, C.byref(tf), extent, d_map,
                                               map_extent, stream))

    def distance_map(self, d_map, d_swap, map_extent, stream=0):
        self.check(self._lib.vkv_distance_map(self.handle, d_map, d_swap, map_extent, stream))

    def distance_map_anisotropic(self, d_maps, d_swap, map_extent, stream=0):
        arr = (C.c_void_p * 8)(*d_maps)
        self.check(self._lib.vkv_distance_map_anisotropic(self.handle, arr, d_swap, map_extent, stream))

    def compute_distance_map(self, d_volume, d_gradient, d_tf, tf, extent, d_maps, d_swap, map_extent, skipping_type, stream=0):
        arr = (C.c_void_p * 8)(*(list(d_maps) + [None] * (8 - len(d_maps))))
        self.check(self._lib.vkv_compute_distance_map(self.handle, d_volume, d_gradient, d_tf, C.byref(tf), extent, arr,
                                                      d_swap, map_extent, skipping_type, stream))

    def render(self, params, stream=0):
        self.check(self._lib.vkv_render(self.handle, C.byref(params), stream))

    def render_batch(self, params_list, stream=0):
        """vkv_render_batch: several frames (same kernel variant, same tile-schedule size) in one launch."""
        arr = (abi.RenderParams * len(params_list))(*params_list)
        self.check(self._lib.vkv_render_batch(self.handle, arr, len(params_list), stream))

    def gather_tiles(self, d_tiles, d_gathered, bytes_per_rank, root, nccl_comm, stream=0):
        self.check(self._lib.vkv_gather_tiles(self.handle, d_tiles, d_gathered, bytes_per_rank, root, nccl_comm, stream))

    def assemble_frame(self, d_tiles, d_gathered, d_image, image_size, tile_size, n_ranks, rank, bytes_per_pixel, root, nccl_comm, stream=0, rect=None):
        """vkv_assemble_frame: ncclGather of the compact tile buffers (the tiles of `rect`, a TileRect; None = the whole image) to `root` +
        de-interleave there, on `stream`."""
        self.check(self._lib.vkv_assemble_frame(self.handle, d_tiles, d_gathered, d_image, image_size[0], image_size[1], tile_size[0], tile_size[1],
                                                None if rect is None else C.byref(rect), n_ranks, rank, bytes_per_pixel, root, nccl_comm, stream))

    def assemble_frames(self, d_tiles, d_gathered, d_images, n_frames, image_size, tile_size, n_ranks, rank, bytes_per_pixel, root, nccl_comm, stream=0,
                        rects=None, roots=None):
        """vkv_assemble_frames: the exchange of a whole launch - frame f with tile rectangle rects[f] (None: the whole image) to its owner
        roots[f] (None: all to `root`, ONE ncclGather of the launch's block; else one group of gathers) + one de-interleave kernel on every owner
        (d_images: n_frames image pointers, entry f read on the owner of frame f only; None on ranks that own nothing)."""
        arr = (C.c_void_p * n_frames)(*d_images) if d_images is not None else None
        rarr = (abi.TileRect * n_frames)(*rects) if rects is not None else None
        oarr = (C.c_int32 * n_frames)(*roots) if roots is not None else None
        self.check(self._lib.vkv_assemble_frames(self.handle, d_tiles, d_gathered, arr, n_frames, image_size[0], image_size[1], tile_size[0], tile_size[1],
                                                 rarr, n_ranks, rank, bytes_per_pixel, root, oarr, nccl_comm, stream))

    # ---- set-up calls ----
    def get_tuning(self):
        t = abi.Tuning()
        self.check(self._lib.vkv_get_tuning(self.handle, C.byref(t)))
        return t

    def set_tuning(self, **fields):
        """read-modify-write of the context's VkvTuning block: ctx.set_tuning(scheduler=1)"""
        t = self.get_tuning()
        for k, v in fields.items():
            if not hasattr(t, k):
                raise AttributeError("VkvTuning has no field %r" % k)
            setattr(t, k, v)
        self.check(self._lib.vkv_set_tuning(self.handle, C.byref(t)))
        return t

    def prepare_render(self, params_list, stream=0):
        arr = (abi.RenderParams * len(params_list))(*params_list)
        self.check(self._lib.vkv_prepare_render(self.handle, arr, len(params_list), stream))

    def register_target(self, d_target, image_size, tiles):
        self.check(self._lib.vkv_register_target(self.handle, d_target, image_size[0], image_size[1], C.byref(tiles)))

    def forget_target(self, d_target):
        self.check(self._lib.vkv_forget_target(self.handle, d_target))

    def release_stream(self, stream):
        self.check(self._lib.vkv_release_stream(self.handle, stream))

    def trim(self):
        """vkv_trim: wait for the device, drop every cached table, empty the arena's table region"""
        self.check(self._lib.vkv_trim(self.handle))

    def release_captured(self, stream):
        """vkv_release_captured: the argument slots of the vkv_render_batch launches captured on `stream` return to the context"""
        self.check(self._lib.vkv_release_captured(self.handle, stream))

    def render_rc(self, params, stream=0):
        """Like render() but returns the status code instead of raising (error-path tests)."""
        return self._lib.vkv_render(self.handle, C.byref(params), stream)

    def scatter_tiles(self, d_gathered, d_image, image_size, tile_size, n_ranks, rank_stride_tiles, bytes_per_pixel, stream=0, rect=None):
        self.check(self._lib.vkv_scatter_tiles(self.handle, d_gathered, d_image, image_size[0], image_size[1], tile_size[0],
                                               tile_size[1], None if rect is None else C.byref(rect), n_ranks, rank_stride_tiles, bytes_per_pixel, stream))

    def synth_volume(self, d_volume, extent, kind, seed, stream=0):
        self.check(self._lib.vkv_synth_volume(self.handle, d_volume, extent, kind, seed, stream))

    def packed_volume_bytes(self, extent):
        return int(self._lib.vkv_packed_volume_bytes(extent))

    def pack_volume(self, d_volume, d_gradient, extent, d_packed, stream=0):
        self.check(self._lib.vkv_pack_volume(self.handle, d_volume, d_gradient, extent, d_packed, stream))

    def transfer_function_bits(self, d_tf, d_bits, stream=0):
        self.check(self._lib.vkv_transfer_function_bits(self.handle, d_tf, d_bits, stream))

    def transfer_function_tables(self, d_tf, tf, d_tables, stream=0):
        self.check(self._lib.vkv_transfer_function_tables(self.handle, d_tf, None if tf is None else C.byref(tf), d_tables, stream))

    def occupied_voxel_count(self, d_volume, d_gradient, tf, extent, d_count, stream=0):
        self.check(self._lib.vkv_occupied_voxel_count(self.handle, d_volume, d_gradient, C.byref(tf), extent, d_count, stream))

    def convert_volume(self, d_raw, voxel_type, big_endian, range_min, range_max, n_voxels, d_out, stream=0):
        self.check(self._lib.vkv_convert_volume(self.handle, d_raw, voxel_type, 1 if big_endian else 0, range_min, range_max, n_voxels, d_out, stream))

    def last_error(self):
        return self._lib.vkv_last_error(self.handle).decode(errors="replace")


# ---- host helpers (no device needed) --------------------------------------------------------------

def transfer_function_uniform(options):
    out = abi.TransferFunctionUniform()
    rc = load().vkv_transfer_function_uniform(C.byref(options), C.byref(out))
    if rc != 0:
        raise VkvError(rc, "vkv_transfer_function_uniform")
    return out


def transfer_function_texture(options):
    import numpy as np
    tex = np.zeros((256, 256, 4), np.uint8)
    rc = load().vkv_transfer_function_texture(C.byref(options), tex.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise VkvError(rc, "vkv_transfer_function_texture")
    return tex


def build_uniforms(view, proj, node_transform, image_transform, clip_distance, image_size, volume_extent, map_extent):
    import numpy as np
    cam, rc_, rg = abi.CameraUniform(), abi.RayCastUniform(), abi.RayGen()
    mats = [np.ascontiguousarray(m, np.float32).reshape(16) for m in (view, proj, node_transform, image_transform)]
    rc = load().vkv_build_uniforms(*[m.ctypes.data_as(C.c_void_p) for m in mats], clip_distance, image_size[0], image_size[1],
                                   volume_extent, map_extent, C.byref(cam), C.byref(rc_), C.byref(rg))
    if rc != 0:
        raise VkvError(rc, "vkv_build_uniforms")
    return cam, rc_, rg


def screen_tile_rect(ray_cast, ray_gen, image_size, tile_size=(16, 16), align_tiles=1):
    """vkv_screen_tile_rect: the tile rectangle the fragments of a frame with these uniforms can lie in (pure CPU; every rank derives the same)"""
    out = abi.TileRect()
    rc = load().vkv_screen_tile_rect(C.byref(ray_cast), C.byref(ray_gen), image_size[0], image_size[1], tile_size[0], tile_size[1], align_tiles, C.byref(out))
    if rc != 0:
        raise VkvError(rc, "vkv_screen_tile_rect")
    return out


def load_header(path):
    """LoadVolume::load_header; raises RuntimeError with the reference's message on failure."""
    h = abi.VolumeHeader()
    if load().vkv_load_header(os.fsencode(path), C.byref(h)) != 0:
        raise RuntimeError("Failed to open header file")
    return h


def load_data(path, header):
    """LoadVolume::load_data → uint8 array [D, H, W]."""
    import numpy as np
    e = header.extent
    out = np.empty((e.depth, e.height, e.width), np.uint8)
    rc = load().vkv_load_data(os.fsencode(path), C.byref(header), out.ctypes.data_as(C.c_void_p), out.nbytes)
    if rc != 0:
        raise RuntimeError("load_data failed (%d)" % rc)
    return out
