"""ctypes/numpy front-end of the CPU oracle (``oracle/vkv_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` — never by the product package.  PARITY UNPINNED (see ``vkv_oracle.h``).
"""
import ctypes as C
import os
import time
import subprocess

import numpy as np

from vkvolume_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    # make is a no-op when the library is newer than its sources and the ABI header
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    except OSError:
        # no make on this machine: a prebuilt library is acceptable only if it is not older than what it is built from
        so = os.path.join(_HERE, "libvkv_oracle.so")
        srcs = [os.path.join(_HERE, "vkv_oracle.c"), os.path.join(_HERE, "vkv_oracle.h"),
                os.path.join(os.path.dirname(_HERE), "include", "vkvolume_amd.h")]
        if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs):
            raise
    # a failed compile (CalledProcessError) always propagates: comparing against a stale oracle would hide real differences


def _has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    build()
    name = "libvkv_oracle_fma.so" if _has_fma() and os.path.exists(os.path.join(_HERE, "libvkv_oracle_fma.so")) \
        else "libvkv_oracle.so"
    # tests/test_sanitizers_cpu.py: the ASan + UBSan build (`make -C oracle asan`), in a process that has the ASan runtime preloaded
    name = os.environ.get("VKV_ORACLE_LIB", name)
    L = C.CDLL(os.path.join(_HERE, name))
    u8p, f32p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.c_void_p
    L.vkvo_transfer_function_uniform.argtypes = [C.POINTER(abi.VolumeOptions), C.POINTER(abi.TransferFunctionUniform)]
    L.vkvo_transfer_function_texture.argtypes = [C.POINTER(abi.VolumeOptions), vp]
    L.vkvo_build_uniforms.argtypes = [vp, vp, vp, vp, C.c_float, C.c_uint32, C.c_uint32, abi.Extent3D, abi.Extent3D,
                                      C.POINTER(abi.CameraUniform), C.POINTER(abi.RayCastUniform), C.POINTER(abi.RayGen)]
    L.vkvo_gradient_map.argtypes = [vp, vp, abi.Extent3D, C.POINTER(abi.TransferFunctionUniform)]
    L.vkvo_occupancy_map.argtypes = [vp, vp, vp, C.POINTER(abi.TransferFunctionUniform), abi.Extent3D, vp, abi.Extent3D]
    L.vkvo_distance_map.argtypes = [vp, vp, abi.Extent3D]
    L.vkvo_occupied_voxel_count.argtypes = [vp, vp, C.POINTER(abi.TransferFunctionUniform), abi.Extent3D]
    L.vkvo_occupied_voxel_count.restype = C.c_uint64
    L.vkvo_distance_map_anisotropic.argtypes = [C.POINTER(vp), vp, abi.Extent3D]
    L.vkvo_compute_distance_map.argtypes = [vp, vp, vp, C.POINTER(abi.TransferFunctionUniform), abi.Extent3D,
                                            C.POINTER(vp), vp, abi.Extent3D, C.c_int32]
    L.vkvo_render.argtypes = [C.POINTER(abi.RenderParams), C.c_int, C.c_uint32]
    L.vkvo_render.restype = C.c_uint64
    L.vkvo_synth_volume.argtypes = [vp, abi.Extent3D, C.c_uint32, C.c_uint32]
    L.vkvo_load_header.argtypes = [C.c_char_p, vp]
    L.vkvo_load_data.argtypes = [C.c_char_p, vp, vp]
    del u8p, f32p
    _LIB = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _extent_of(vol):
    d, h, w = vol.shape
    return abi.Extent3D(w, h, d)


def map_extent(extent, block):
    """ceil(extent / block), src/volume_component.cpp:91-92"""
    return abi.Extent3D(*[(v + block - 1) // block for v in extent.as_tuple()])


def transfer_function_uniform(options):
    out = abi.TransferFunctionUniform()
    lib().vkvo_transfer_function_uniform(C.byref(options), C.byref(out))
    return out


def transfer_function_texture(options):
    tex = np.zeros((256, 256, 4), np.uint8)
    lib().vkvo_transfer_function_texture(C.byref(options), _ptr(tex))
    return tex


def build_uniforms(view, proj, node_transform, image_transform, clip_distance, image_size, volume_extent, map_ext):
    cam, rc, rg = abi.CameraUniform(), abi.RayCastUniform(), abi.RayGen()
    mats = [np.ascontiguousarray(m, np.float32).reshape(16) for m in (view, proj, node_transform, image_transform)]
    lib().vkvo_build_uniforms(_ptr(mats[0]), _ptr(mats[1]), _ptr(mats[2]), _ptr(mats[3]), clip_distance,
                              image_size[0], image_size[1], volume_extent, map_ext, C.byref(cam), C.byref(rc), C.byref(rg))
    return cam, rc, rg


def gradient_map(vol, tf):
    """vol: uint8 [D,H,W] → gradient uint8 [D,H,W]"""
    vol = np.ascontiguousarray(vol, np.uint8)
    grad = np.empty_like(vol)
    lib().vkvo_gradient_map(_ptr(vol), _ptr(grad), _extent_of(vol), C.byref(tf))
    return grad


def occupancy_map(vol, grad, tf_tex, tf, block, map_extent_override=None):
    """map_extent_override: a map extent that is not ceil(extent / block) (the block size then follows from it, src/compute_distance_map.cpp:110-113)"""
    vol = np.ascontiguousarray(vol, np.uint8)
    me = map_extent_override if map_extent_override is not None else map_extent(_extent_of(vol), block)
    out = np.empty((me.depth, me.height, me.width), np.uint8)
    lib().vkvo_occupancy_map(_ptr(vol), _ptr(grad), _ptr(tf_tex), C.byref(tf), _extent_of(vol), _ptr(out), me)
    return out


def occupied_voxel_count(vol, grad, tf):
    vol = np.ascontiguousarray(vol, np.uint8)
    return int(lib().vkvo_occupied_voxel_count(_ptr(vol), _ptr(grad), C.byref(tf), _extent_of(vol)))


def distance_map(occ):
    m = np.ascontiguousarray(occ, np.uint8).copy()
    swap = np.zeros_like(m)
    lib().vkvo_distance_map(_ptr(m), _ptr(swap), _extent_of(m))
    return m


def distance_map_anisotropic(occ):
    """occ: occupancy [md,mh,mw] → uint8 [8,md,mh,mw], index = (dz<0) + 2(dy<0) + 4(dx<0)"""
    occ = np.ascontiguousarray(occ, np.uint8)
    maps = np.zeros((8,) + occ.shape, np.uint8)
    maps[7] = occ
    swap = np.zeros_like(occ)
    ptrs = (C.c_void_p * 8)(*[maps[i].ctypes.data for i in range(8)])
    lib().vkvo_distance_map_anisotropic(ptrs, _ptr(swap), _extent_of(occ))
    return maps


def compute_distance_map(vol, grad, tf_tex, tf, block, skipping_type):
    """ComputeDistanceMap::compute → array [n_maps, md, mh, mw] (n_maps = 8 for anisotropic, else 1)."""
    vol = np.ascontiguousarray(vol, np.uint8)
    me = map_extent(_extent_of(vol), block)
    n = 8 if skipping_type == abi.SKIP_ANISOTROPIC_DISTANCE else 1
    maps = np.zeros((8, me.depth, me.height, me.width), np.uint8)
    swap = np.zeros((me.depth, me.height, me.width), np.uint8)
    ptrs = (C.c_void_p * 8)(*[maps[i].ctypes.data for i in range(8)])
    lib().vkvo_compute_distance_map(_ptr(vol), _ptr(grad), _ptr(tf_tex), C.byref(tf), _extent_of(vol), ptrs, _ptr(swap),
                                    me, skipping_type)
    return maps[:n].copy()


def synth_volume(shape_whd, kind, seed):
    w, h, d = shape_whd
    vol = np.empty((d, h, w), np.uint8)
    lib().vkvo_synth_volume(_ptr(vol), abi.Extent3D(w, h, d), kind, seed)
    return vol


class RenderResult:
    def __init__(self, color, counts, depth, rgba8, rays, seconds=0.0):
        self.color, self.counts, self.depth, self.rgba8, self.rays = color, counts, depth, rgba8, rays
        self.seconds = seconds  # wall time of the vkvo_render call alone (outputs allocated and touched before it)


def render(params, vol, grad, tf_tex, maps, n_threads=None, pixel_stride=1, want_rgba8=False, in_depth=None, target_color=None,
           target_rgba8=None, reuse=None):
    """Run the oracle ray-marcher. ``params`` is an ``abi.RenderParams`` whose pointer fields are overwritten with
    host arrays; outputs are image-shaped (or compact, following ``params.tiles``).  ``reuse``: a RenderResult of the same shape whose
    arrays are cleared and written again (bench.py: no allocation, no first-touch page faults inside the timed call).  ``in_depth`` feeds the DEPTH_ATTACHMENT
    variant; ``target_color`` / ``target_rgba8`` are existing frames to blend onto (sets ``blend_over_target``)."""
    p = abi.RenderParams.from_buffer_copy(params)
    vol = np.ascontiguousarray(vol, np.uint8)
    tf_tex = np.ascontiguousarray(tf_tex, np.uint8)
    keep = [vol, tf_tex]
    p.d_volume = vol.ctypes.data
    p.d_gradient = None
    if grad is not None:
        grad = np.ascontiguousarray(grad, np.uint8)
        keep.append(grad)
        p.d_gradient = grad.ctypes.data
    p.d_transfer_function = tf_tex.ctypes.data
    for i in range(8):
        p.d_distance_maps[i] = None
    if maps is not None:
        for i in range(len(maps)):
            m = np.ascontiguousarray(maps[i], np.uint8)
            keep.append(m)
            p.d_distance_maps[i] = m.ctypes.data
    if p.tiles.compact:
        npix = p.tiles.tile_count * p.tiles.tile_width * p.tiles.tile_height
        shape = (npix,)
    else:
        shape = (p.image_height, p.image_width)
    blend = target_color is not None or target_rgba8 is not None
    want_rgba8 = want_rgba8 or target_rgba8 is not None
    rgba8 = None
    if reuse is not None and not blend and reuse.counts.shape == shape + (3,) and (reuse.rgba8 is not None) == want_rgba8:
        color, counts, depth, rgba8 = reuse.color, reuse.counts, reuse.depth, reuse.rgba8
        color.fill(0), counts.fill(0), depth.fill(0)
        if rgba8 is not None:
            rgba8.fill(0)
    else:
        color = np.zeros(shape + (4,), np.float32) if target_color is None else np.ascontiguousarray(target_color, np.float32).copy()
        counts = np.zeros(shape + (3,), np.uint32)
        depth = np.zeros(shape, np.float32)
        if want_rgba8:
            rgba8 = np.zeros(shape + (4,), np.uint8) if target_rgba8 is None else np.ascontiguousarray(target_rgba8, np.uint8).copy()
    p.blend_over_target = 1 if blend else 0
    p.d_in_depth = None
    if in_depth is not None:
        in_depth = np.ascontiguousarray(in_depth, np.float32)
        keep.append(in_depth)
        p.d_in_depth = in_depth.ctypes.data
    p.d_out_color, p.d_out_counts, p.d_out_depth = color.ctypes.data, counts.ctypes.data, depth.ctypes.data
    p.d_out_rgba8 = rgba8.ctypes.data if want_rgba8 else None
    if n_threads is None:
        n_threads = os.cpu_count() or 1
    t0 = time.perf_counter()
    rays = lib().vkvo_render(C.byref(p), n_threads, pixel_stride)
    seconds = time.perf_counter() - t0
    del keep
    return RenderResult(color, counts, depth, rgba8, rays, seconds)


class Header(C.Structure):
    """VkvoHeader (LoadVolume::Header, src/load_volume.h:29-39)"""
    _fields_ = [("extent", abi.Extent3D), ("voxel_size", C.c_float * 3), ("normalisation_range", C.c_float * 2),
                ("type", C.c_char * 16), ("endianness", C.c_char * 16), ("image_transform", C.c_float * 16)]


def load_header(path):
    h = Header()
    rc = lib().vkvo_load_header(path.encode(), C.byref(h))
    if rc != 0:
        raise RuntimeError("Failed to open header file")
    return h


def load_data(path, header):
    e = header.extent
    out = np.empty((e.depth, e.height, e.width), np.uint8)
    rc = lib().vkvo_load_data(path.encode(), C.byref(header), _ptr(out))
    if rc != 0:
        raise RuntimeError("load_data failed (%d)" % rc)
    return out


def trace_ray_steps(params, vol, grad, tf_tex, maps, px, py, cap=4096):
    """Diagnostics (tools/): the event kinds (b'P', b'O', b'S', b'A') and loop indices of one pixel's ray."""
    L = lib()
    if not getattr(L, "_trace_steps_bound", False):
        L.vkvo_trace_ray_steps.argtypes = [C.POINTER(abi.RenderParams), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
        L.vkvo_trace_ray_steps.restype = C.c_uint32
        L._trace_steps_bound = True
    p = abi.RenderParams.from_buffer_copy(params)
    p.d_volume, p.d_gradient, p.d_transfer_function = vol.ctypes.data, (grad.ctypes.data if grad is not None else None), tf_tex.ctypes.data
    for i in range(8):
        p.d_distance_maps[i] = maps[i].ctypes.data if maps is not None and i < len(maps) else None
    p.d_in_depth = None
    ev, st = np.zeros(cap, np.uint8), np.zeros(cap, np.int32)
    n = L.vkvo_trace_ray_steps(C.byref(p), px, py, _ptr(ev), _ptr(st), cap)
    n = min(int(n), cap)
    return ev[:n], st[:n]
