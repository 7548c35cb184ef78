"""CPU: the C-ABI library loads and exports every symbol the header declares, the ctypes mirror matches the C layout,
the host-side helpers agree with the oracle, and the product fails loudly (no fallback) without its library / a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from oracle import vkv_oracle as O
from tests import helpers as T
from vkvolume_amd import abi, camera, lib, multigpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vkvolume_amd.h")


DEBUG_HEADER = os.path.join(ROOT, "include", "vkvolume_amd_debug.h")


def declared_functions(header=HEADER):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vkv_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    names = declared_functions()
    assert len(names) >= 15
    assert sorted(names) == sorted(lib.EXPORTS)
    debug = declared_functions(DEBUG_HEADER)
    assert sorted(debug) == sorted(lib.DEBUG_EXPORTS)  # the diagnostic entry points have a header of their own
    for n in names + debug:
        assert hasattr(L, n), n
    # nothing else with the library's prefix is exported: no undeclared entry points
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib.LIB_PATH]).decode()
    exported = sorted(set(re.findall(r"\b(vkv_[a-z_0-9]+)$", out, flags=re.M)))
    assert exported == sorted(names + debug), set(exported) ^ set(names + debug)
    assert b"gfx950" in L.vkv_version()


def test_ctypes_mirror_matches_c_layout(tmp_path):
    exe = tmp_path / "abi_probe"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), os.path.join(ROOT, "tests", "abi_probe.c")])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    mirror = {"VkvExtent3D": abi.Extent3D, "VkvTransferFunctionUniform": abi.TransferFunctionUniform, "VkvVolumeOptions": abi.VolumeOptions,
              "VkvCameraUniform": abi.CameraUniform, "VkvRayCastUniform": abi.RayCastUniform, "VkvRayGen": abi.RayGen,
              "VkvRenderOptions": abi.RenderOptions, "VkvTileRect": abi.TileRect, "VkvTileSchedule": abi.TileSchedule, "VkvRenderParams": abi.RenderParams,
              "VkvTuning": abi.Tuning}
    n = 0
    for line in out:
        f = line.split()
        if not f:
            continue
        if f[0] == "sizeof":
            assert C.sizeof(mirror[f[1]]) == int(f[2]), line
        else:
            t, field = f[1].split(".")
            assert getattr(mirror[t], field).offset == int(f[2]), line
        n += 1
    assert n >= 25
    # the reference's uniform block sizes
    assert C.sizeof(abi.CameraUniform) == 320 and C.sizeof(abi.RayCastUniform) == 68 and C.sizeof(abi.TransferFunctionUniform) == 32


@pytest.mark.parametrize("opts", [dict(), T.APP_TF, dict(intensity_min=0.4, intensity_max=0.8, gradient_min=0.0, gradient_max=0.0),
                                  dict(intensity_min=0.2, intensity_max=0.8, gradient_min=0.06, gradient_max=0.12),
                                  dict(sampling_factor=2.5, voxel_alpha_factor=0.3, intensity_min=0.5, intensity_max=0.5)])
def test_transfer_function_helpers_match_oracle(opts):
    o = abi.VolumeOptions(**opts)
    assert bytes(lib.transfer_function_uniform(o)) == bytes(O.transfer_function_uniform(o))
    assert np.array_equal(lib.transfer_function_texture(o), O.transfer_function_texture(o))


@pytest.mark.parametrize("az,el,radius,clip", [(0.0, 0.0, 150.0, 1.0), (33.0, 20.0, 95.0, 1.0), (200.0, -40.0, 30.0, 12.0), (90.0, 89.0, 60.0, 50.0)])
def test_build_uniforms_matches_double_precision_oracle(az, el, radius, clip):
    ext = abi.Extent3D(1024, 1024, 795)
    me = O.map_extent(ext, 4)
    ixf = camera.image_transform((0.0003, 0.0003, 0.0007), ext.as_tuple(), (1, 0, 0, 90))
    node = camera.benchmark_node_transform(ixf)
    view, proj = camera.orbit_camera(az, el, radius), camera.perspective_vulkan(60.0, 16 / 9)
    a = lib.build_uniforms(view, proj, node, ixf, clip, (1920, 1080), ext, me)
    b = O.build_uniforms(view, proj, node, ixf, clip, (1920, 1080), ext, me)
    assert a[1].front_index == b[1].front_index
    assert list(a[1].block_size) == [4.0, 4.0, 4.0, 0.0]
    for x, y in zip(a, b):
        n = C.sizeof(x) // 4 * 4
        fa, fb = np.frombuffer(bytes(x)[:n], np.float32)[:16 * 5], np.frombuffer(bytes(y)[:n], np.float32)[:16 * 5]
        if isinstance(x, abi.RayCastUniform):
            fa, fb = fa[:16], fb[:16]
        assert np.abs(fa - fb).max() <= 2e-6 * max(1.0, np.abs(fb).max())
    # the clip plane sits clip_distance in front of the camera: plane . (cam_pos, 1) == -clip  (volume_render_subpass.cpp:238)
    cam_pos = np.linalg.inv(view.astype(np.float64).T)[:3, 3]
    assert abs(np.dot(list(a[1].plane)[:3], cam_pos) + a[1].plane[3] + clip) < 1e-3


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "24"))))
def test_build_uniforms_fuzz(seed):
    """vkv_build_uniforms (the product's fp32 statement of src/volume_render_subpass.cpp:219-249) against the oracle's double-precision one on
    random scenes: extent, block size, voxel size, rotation, camera position / field of view / aspect, clip distance.  Same front-face
    index, same block size, every float within 1e-5 of the largest entry of its block (fp32 products of 4x4 matrices and an inverse)."""
    rng = np.random.default_rng(55000 + seed)
    ext = abi.Extent3D(*(int(x) for x in rng.integers(8, 1200, size=3)))
    block = int(rng.integers(1, 9))
    me = O.map_extent(ext, block)
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    ixf = camera.image_transform(tuple(float(x) for x in rng.uniform(0.0003, 2.0, size=3)), ext.as_tuple(),
                                 (float(axis[0]), float(axis[1]), float(axis[2]), float(rng.uniform(0, 360))))
    node = camera.benchmark_node_transform(ixf)
    size = (int(rng.integers(16, 4000)), int(rng.integers(16, 2200)))
    view = camera.orbit_camera(float(rng.uniform(0, 360)), float(rng.uniform(-85, 85)), float(rng.choice([40.0, 95.0, 150.0, 400.0])))
    proj = camera.perspective_vulkan(float(rng.uniform(20, 110)), size[0] / size[1])
    clip = float(rng.choice([0.1, 1.0, 12.0, 50.0]))
    a = lib.build_uniforms(view, proj, node, ixf, clip, size, ext, me)
    b = O.build_uniforms(view, proj, node, ixf, clip, size, ext, me)
    assert a[1].front_index == b[1].front_index
    assert list(a[1].block_size) == list(b[1].block_size)
    for x, y in zip(a, b):
        n = C.sizeof(x) // 4 * 4
        fa, fb = np.frombuffer(bytes(x)[:n], np.float32)[:16 * 5], np.frombuffer(bytes(y)[:n], np.float32)[:16 * 5]
        if isinstance(x, abi.RayCastUniform):
            fa, fb = fa[:16], fb[:16]
        assert np.abs(fa - fb).max() <= 1e-5 * max(1.0, np.abs(fb).max()), type(x).__name__


def test_map_extent_is_ceil_div():
    assert O.map_extent(abi.Extent3D(1024, 1024, 795), 4).as_tuple() == (256, 256, 199)
    assert O.map_extent(abi.Extent3D(64, 64, 64), 4).as_tuple() == (16, 16, 16)
    assert O.map_extent(abi.Extent3D(9, 8, 7), 4).as_tuple() == (3, 2, 2)


@pytest.mark.parametrize("frame,tile,world", [((1920, 1080), 16, 1), ((3840, 1080), 16, 2), ((150, 70), 16, 3), ((7680, 2160), 16, 8), ((33, 17), 16, 8)])
def test_tile_schedules_partition_the_frame(frame, tile, world):
    tiles_x, tiles_y = -(-frame[0] // tile), -(-frame[1] // tile)
    seen = np.zeros(tiles_x * tiles_y, int)
    rays = 0
    for r in range(world):
        s = abi.full_frame_tiles(frame[0], frame[1], tile, tile, r, world, compact=True)
        for k in range(s.tile_count):
            seen[s.tile_first + k * s.tile_stride] += 1
        g = multigpu.TileGather(None, r, world, frame, tile, 4, device="cpu")
        assert g.schedule.tile_count == s.tile_count and s.tile_count <= g.tiles_per_rank
        rays += g.my_ray_count()
    assert (seen == 1).all() and rays == frame[0] * frame[1]
    # round 6: the same deal inside a tile rectangle (tiles numbered row-major inside it)
    rect = abi.TileRect(tiles_x // 3, tiles_y // 4, max(1, tiles_x // 2), max(1, tiles_y // 2))
    seen = np.zeros((tiles_y, tiles_x), int)
    rays = 0
    for r in range(world):
        g = multigpu.TileGather(None, r, world, frame, tile, 4, device="cpu")
        s = g.rect_schedule(rect)
        assert s.tile_count <= multigpu.tiles_per_rank(rect, world) <= g.tiles_per_rank and s.rect.as_tuple() == rect.as_tuple()
        for k in range(s.tile_count):
            t = s.tile_first + k * s.tile_stride
            seen[rect.y0 + t // rect.w, rect.x0 + t % rect.w] += 1
        rays += g.my_ray_count(rect)
    inside = np.zeros_like(seen, bool)
    inside[rect.y0:rect.y0 + rect.h, rect.x0:rect.x0 + rect.w] = True
    assert (seen[inside] == 1).all() and not seen[~inside].any()
    assert rays == (min(frame[0], (rect.x0 + rect.w) * tile) - rect.x0 * tile) * (min(frame[1], (rect.y0 + rect.h) * tile) - rect.y0 * tile)


def test_product_fails_loudly_without_library(monkeypatch):
    monkeypatch.setattr(lib, "_LIB", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libvkvolume_amd.so")
    with pytest.raises(lib.VkvError, match="no CPU fallback"):
        lib.Context(0)
    with pytest.raises(lib.VkvError):
        lib.transfer_function_texture(abi.VolumeOptions())


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_context_creation_fails_without_gpu():
    with pytest.raises(lib.VkvError) as e:
        lib.Context(0)
    assert e.value.code == abi.VKV_E_NO_DEVICE


def test_null_arguments_are_rejected_on_the_host_side():
    L = lib.load()
    assert L.vkv_transfer_function_uniform(None, None) == abi.VKV_E_INVALID_ARGUMENT
    assert L.vkv_transfer_function_texture(None, None) == abi.VKV_E_INVALID_ARGUMENT
    assert L.vkv_render(None, None, None) == abi.VKV_E_INVALID_ARGUMENT
    assert L.vkv_gradient_map(None, None, None, abi.Extent3D(1, 1, 1), None, None) == abi.VKV_E_INVALID_ARGUMENT
    L.vkv_destroy(None)  # no-op


def test_loader_c_abi_matches_oracle_loader(tmp_path):
    """LoadVolume behind vkv_load_header / vkv_load_data (the product's C++ loader) against the oracle's C loader."""
    rng = np.random.default_rng(9)
    cases = [("uint8_t", "u1", "little"), ("int8_t", "i1", "big"), ("uint16_t", "u2", "little"), ("uint16_t", "u2", "big"),
             ("int16_t", "i2", "little"), ("int16_t", "i2", "big")]
    for ctype, npt, endian in cases:
        info = np.iinfo(npt)
        raw = rng.integers(info.min, info.max + 1, size=(7, 5, 9)).astype(npt)
        f = tmp_path / ("vol_%s_%s.raw" % (ctype, endian))
        raw.astype(("<" if endian == "little" else ">") + npt).tofile(f)
        lo, hi = (10.0, 200.0) if npt[1] == "1" else (400.0, 25380.0)
        (tmp_path / (f.name + ".header")).write_text("9 5 7 # extents\n0.0003 0.0003 0.0007 # voxel size\n%g %g # range\n%s %s # type\n"
                                                     "1 0 0 90 # rotation\n" % (lo, hi, ctype, endian))
        h = lib.load_header(str(f) + ".header")
        ho = O.load_header(str(f) + ".header")
        assert h.extent.as_tuple() == ho.extent.as_tuple() == (9, 5, 7)
        assert h.type == ho.type == ctype.encode() and h.endianness == ho.endianness == endian.encode()
        assert list(h.voxel_size) == list(ho.voxel_size) and list(h.normalisation_range) == list(ho.normalisation_range)
        assert np.allclose(list(h.image_transform), list(ho.image_transform), rtol=0, atol=1e-7)
        assert np.array_equal(lib.load_data(str(f), h), O.load_data(str(f), ho))
    with pytest.raises(RuntimeError, match="Failed to open header file"):
        lib.load_header(str(tmp_path / "nope.header"))
    (tmp_path / "bad.raw").write_bytes(b"\0" * 11)
    (tmp_path / "bad.raw.header").write_text("9 5 7\n1 1 1\n0 255\nuint8_t little\n1 0 0 0\n")
    with pytest.raises(RuntimeError):
        lib.load_data(str(tmp_path / "bad.raw"), lib.load_header(str(tmp_path / "bad.raw.header")))  # size mismatch
    (tmp_path / "f32.raw.header").write_text("1 1 1\n1 1 1\n0 255\nfloat little\n1 0 0 0\n")
    (tmp_path / "f32.raw").write_bytes(b"\0" * 4)
    with pytest.raises(RuntimeError):
        lib.load_data(str(tmp_path / "f32.raw"), lib.load_header(str(tmp_path / "f32.raw.header")))  # unsupported type


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "24"))))
def test_loader_fuzz(tmp_path, seed):
    """The product's C++ LoadVolume (vkv_load_header / vkv_load_data) against the oracle's C loader on random files: voxel type, byte order,
    extent, voxel size, normalisation range (inside / around / beyond the type's range), rotation axis and angle, comments or none in the
    header: header fields equal, image transform within 1e-6 (two float evaluations of the same formula), every data byte equal."""
    rng = np.random.default_rng(33000 + seed)
    ctype, npt = [("uint8_t", "u1"), ("int8_t", "i1"), ("uint16_t", "u2"), ("int16_t", "i2")][int(rng.integers(0, 4))]
    endian = ("little", "big")[int(rng.integers(0, 2))]
    w, h, d = (int(x) for x in rng.integers(1, 24, size=3))
    info = np.iinfo(npt)
    span = float(info.max) - float(info.min)
    lo = float(np.round(rng.uniform(info.min - 0.2 * span, info.max), int(rng.integers(0, 3))))
    hi = lo + float(rng.choice([1.0, 0.25 * span, span, 3.0 * span])) * float(rng.uniform(0.2, 1.0))
    raw = rng.integers(info.min, info.max + 1, size=(d, h, w)).astype(npt)
    f = tmp_path / "v.raw"
    raw.astype(("<" if endian == "little" else ">") + npt).tofile(f)
    vs = [float(np.round(x, 5)) for x in rng.uniform(0.0002, 3.0, size=3)]
    axis = [float(np.round(x, 3)) for x in rng.normal(size=3)]
    if all(a == 0.0 for a in axis):
        axis = [1.0, 0.0, 0.0]
    angle = float(np.round(rng.uniform(-360, 360), 2))
    c = (lambda t: " # " + t) if rng.random() < 0.5 else (lambda t: "")
    (tmp_path / "v.raw.header").write_text("%d %d %d%s\n%r %r %r%s\n%r %r%s\n%s %s%s\n%r %r %r %r%s\n" % (
        w, h, d, c("extents"), vs[0], vs[1], vs[2], c("voxel size"), lo, hi, c("range"), ctype, endian, c("type"), axis[0], axis[1], axis[2], angle, c("rotation")))
    hp, ho = lib.load_header(str(f) + ".header"), O.load_header(str(f) + ".header")
    assert hp.extent.as_tuple() == ho.extent.as_tuple() == (w, h, d)
    assert hp.type == ho.type == ctype.encode() and hp.endianness == ho.endianness == endian.encode()
    assert list(hp.voxel_size) == list(ho.voxel_size) and list(hp.normalisation_range) == list(ho.normalisation_range)
    scale = max(1.0, float(np.abs(np.array(list(ho.image_transform))).max()))
    assert np.allclose(list(hp.image_transform), list(ho.image_transform), rtol=0, atol=1e-6 * scale)
    assert np.array_equal(lib.load_data(str(f), hp), O.load_data(str(f), ho))


def test_bench_spreads_the_frames_of_a_block_evenly_over_its_launches():
    import bench
    assert bench.split_frames(20, 8) == [7, 7, 6]
    assert bench.split_frames(256, 8) == [8] * 32
    assert bench.split_frames(5, 8) == [5]
    assert bench.split_frames(17, 8) == [6, 6, 5]
    for steps in range(1, 70):
        for fpl in (1, 3, 8, 12, 32):
            parts = bench.split_frames(steps, fpl)
            assert sum(parts) == steps and max(parts) <= fpl and max(parts) - min(parts) <= 1 and len(parts) == -(-steps // fpl)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "40"))))
def test_screen_tile_rect_holds_every_fragment(seed):
    """vkv_screen_tile_rect (the counterpart of the reference's rasteriser only shading the clipped box's faces, src/volume_render_subpass.cpp:262-293):
    a frame rendered through the rectangle's schedule into cleared buffers equals the full-frame render - no pixel outside the rectangle has a
    fragment - for cameras outside, at, inside and beside the box, clip planes in front of, through and behind it, rotated anisotropic volumes;
    and the rectangle is TIGHT where the box is well in front of the camera (within the 2-pixel margin and the tile rounding of the pixels
    that do have a fragment)."""
    rng = np.random.default_rng(7700 + seed)
    shape = tuple(int(x) for x in rng.integers(8, 30, size=3))
    vol = O.synth_volume(shape, 1, int(rng.integers(1, 1 << 30)))
    voxel = tuple(float(x) for x in rng.uniform(0.3, 2.0, size=3))
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    scene = T.OracleScene(vol, abi.VolumeOptions(**T.APP_TF), 4, voxel_size=voxel, axis_angle=(float(axis[0]), float(axis[1]), float(axis[2]), float(rng.uniform(0, 360))))
    size = (int(rng.integers(3, 12)) * 16 - int(rng.integers(0, 16)), int(rng.integers(2, 8)) * 16 - int(rng.integers(0, 16)))
    radius = float(rng.choice([30.0, 60.0, 110.0, 150.0, 260.0, 500.0]))  # the node is scaled to 100 units: 30 / 60 put the camera inside or at the box
    centre = (0.0, 0.0, 0.0) if seed % 3 else tuple(float(x) for x in rng.uniform(-90.0, 90.0, size=3))  # every third seed: the box off-centre / off screen
    view = camera.orbit_camera(float(rng.uniform(0, 360)), float(rng.uniform(-80, 80)), radius, centre)
    proj = camera.perspective_vulkan(float(rng.uniform(25, 100)), size[0] / size[1])
    clip = float(rng.choice([0.1, 1.0, 1.0, 20.0, 70.0, 120.0, 200.0]))
    opts = abi.RenderOptions(skipping_type=abi.SKIP_NONE, clip_distance=clip, test=abi.TEST_RAY_ENTRY)  # the ray entry of every fragment: no marching needed
    params = scene.params(view, proj, size, opts)
    align = int(rng.choice([1, 1, 2, 4]))
    rect = lib.screen_tile_rect(params.ray_cast, params.ray_gen, size, (16, 16), align)
    tiles_x, tiles_y = -(-size[0] // 16), -(-size[1] // 16)
    assert rect.w >= 1 and rect.h >= 1 and rect.x0 + rect.w <= tiles_x and rect.y0 + rect.h <= tiles_y
    full = scene.render(params)
    covered = full.color[..., 3] != 0  # TEST_RAY_ENTRY writes alpha 1 for every fragment
    p2 = scene.params(view, proj, size, opts, tiles=abi.full_frame_tiles(size[0], size[1], rect=rect))
    part = scene.render(p2)
    assert np.array_equal(part.color, full.color), "a fragment lies outside the rectangle %s (image %s tiles)" % (rect.as_tuple(), (tiles_x, tiles_y))
    ys, xs = np.nonzero(covered)
    if len(xs) and radius >= 150.0 and align == 1 and clip <= 20.0 and seed % 3:  # (an off-centre orbit can put the camera next to the box)
        # tight: the rectangle's pixel bound is within 2 pixels (the margin) + 1 (sampling at pixel centres) + a tile of the covered pixels' bound
        assert rect.x0 * 16 >= xs.min() - 19 and rect.y0 * 16 >= ys.min() - 19, (rect.as_tuple(), xs.min(), ys.min())
        assert (rect.x0 + rect.w) * 16 <= xs.max() + 1 + 19 and (rect.y0 + rect.h) * 16 <= ys.max() + 1 + 19, (rect.as_tuple(), xs.max(), ys.max())
    print("seed %d: image %dx%d tiles, rect %s, %d covered pixels, radius %g clip %g" % (seed, tiles_x, tiles_y, rect.as_tuple(), len(xs), radius, clip))
