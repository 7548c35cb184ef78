"""GPU: the occupied-voxel statistic and the C++ host mirror (Volume / ComputeGradientMap / ComputeDistanceMap /
ComputeOccupiedVoxelCount / VolumeRenderSubpass driven by the vkv_offscreen executable) against the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from oracle import vkv_oracle as O
from tests import helpers as T
from vkvolume_amd import abi, lib

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "vkvolume_amd", "csrc", "vkv_offscreen")


@pytest.mark.parametrize("opts,use_map", [(T.APP_TF, True), (T.APP_TF, False),
                                          (dict(intensity_min=0.4, intensity_max=0.8, gradient_min=0.0, gradient_max=0.0), True),
                                          (dict(intensity_min=0.2, intensity_max=0.8, gradient_min=0.06, gradient_max=0.12), True)])
@pytest.mark.parametrize("shape", [(64, 64, 64), (70, 33, 19), (1, 1, 1)])
def test_occupied_voxel_count_parity(ctx, opts, use_map, shape):
    vol = O.synth_volume(shape, 1, 6)
    opt = abi.VolumeOptions(use_precomputed_gradient=use_map, **opts)
    tf = lib.transfer_function_uniform(opt)
    grad = O.gradient_map(vol, tf) if use_map else None
    d_vol = torch.from_numpy(vol).cuda()
    d_grad = torch.from_numpy(grad).cuda() if grad is not None else None
    d_count = torch.full((1,), 12345, dtype=torch.int64, device="cuda")
    ctx.occupied_voxel_count(d_vol.data_ptr(), None if d_grad is None else d_grad.data_ptr(), tf, abi.Extent3D(*shape), d_count.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
    assert int(d_count.item()) == O.occupied_voxel_count(vol, grad, tf)


@pytest.mark.parametrize("imin", [-0.2, 0.0, 1.0 / 255.0, 0.3, 127.0 / 255.0, 0.5, 128.5 / 255.0, 0.75, 254.0 / 255.0, 1.0, 1.5])
@pytest.mark.parametrize("use_map", [True, False])
@pytest.mark.parametrize("shape", [(64, 24, 20), (37, 29, 23), (258, 9, 7)])
def test_occupied_voxel_count_thresholds(ctx, imin, use_map, shape):
    """The count kernel skips the gradient rows of a batch whose every intensity byte lies below the lowest byte with alpha (round 6): every
    threshold position - none (imin < 0), both halves of the byte range (the SWAR compare treats bit 7 separately), all (imin >= 1) - on volumes
    that are mostly below it, with a few bytes above in otherwise skipped batches."""
    rng = np.random.default_rng(int(imin * 1000) % 997 + shape[0])
    lo = int(np.clip(imin * 255.0, 0, 255))
    vol = rng.integers(0, max(lo, 1), size=shape[::-1], dtype=np.uint8)          # below the threshold ...
    hot = rng.random(vol.shape) < 0.002
    vol[hot] = rng.integers(0, 256, size=int(hot.sum()), dtype=np.uint8)         # ... but for a few voxels anywhere in the range
    vol[shape[2] // 2] = rng.integers(0, 256, size=vol.shape[1:], dtype=np.uint8)  # and one slice of plain noise
    opt = abi.VolumeOptions(use_precomputed_gradient=use_map, intensity_min=imin, intensity_max=imin + 0.2, gradient_min=0.02, gradient_max=0.3)
    tf = lib.transfer_function_uniform(opt)
    grad = O.gradient_map(vol, tf) if use_map else None
    d_vol = torch.from_numpy(vol).cuda()
    d_grad = torch.from_numpy(grad).cuda() if grad is not None else None
    d_count = torch.full((1,), 12345, dtype=torch.int64, device="cuda")
    ctx.occupied_voxel_count(d_vol.data_ptr(), None if d_grad is None else d_grad.data_ptr(), tf, abi.Extent3D(*shape), d_count.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
    expect = O.occupied_voxel_count(vol, grad, tf)
    assert int(d_count.item()) == expect
    if imin >= 1.0:
        assert expect == 0


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "16"))))
def test_device_loader_conversion_fuzz(ctx, tmp_path, seed):
    """vkv_convert_volume against the oracle's CPU loader (src/load_volume.cpp:112-172) on random files: voxel type, byte order, extent
    (odd voxel counts: the device path converts several voxels per thread), normalisation range - inside, around and beyond the type's
    range, sometimes a single value wide."""
    rng = np.random.default_rng(21000 + seed)
    ctype, npt = [("uint8_t", "u1"), ("int8_t", "i1"), ("uint16_t", "u2"), ("int16_t", "i2")][int(rng.integers(0, 4))]
    endian = ("little", "big")[int(rng.integers(0, 2))]
    w, h, d = (int(x) for x in rng.integers(1, 40, size=3))
    info = np.iinfo(npt)
    span = float(info.max) - float(info.min)
    lo = float(np.round(rng.uniform(info.min - 0.2 * span, info.max), int(rng.integers(0, 3))))
    hi = lo + float(rng.choice([1.0, 0.25 * span, span, 3.0 * span])) * float(rng.uniform(0.2, 1.0))
    raw = rng.integers(info.min, info.max + 1, size=(d, h, w)).astype(npt)
    if rng.random() < 0.3:        # a narrow band of values around the range's ends
        raw = np.clip(np.round(rng.normal(lo, 2.0, size=(d, h, w))), info.min, info.max).astype(npt)
    f = tmp_path / "v.raw"
    file_bytes = raw.astype(("<" if endian == "little" else ">") + npt)
    file_bytes.tofile(f)
    (tmp_path / "v.raw.header").write_text("%d %d %d\n1 1 1\n%r %r\n%s %s\n1 0 0 0\n" % (w, h, d, lo, hi, ctype, endian))
    hdr = O.load_header(str(f) + ".header")
    expect = O.load_data(str(f), hdr)
    d_raw = torch.from_numpy(np.frombuffer(file_bytes.tobytes(), np.uint8).copy()).cuda()
    d_out = torch.full((d, h, w), 7, dtype=torch.uint8, device="cuda")
    ctx.convert_volume(d_raw.data_ptr(), abi.VOXEL_TYPES[ctype], endian == "big", hdr.normalisation_range[0], hdr.normalisation_range[1], w * h * d,
                       d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(d_out.cpu().numpy(), expect), "%s %s %dx%dx%d range %r..%r" % (ctype, endian, w, h, d, lo, hi)


@pytest.mark.parametrize("ctype,npt", [("uint8_t", "u1"), ("int8_t", "i1"), ("uint16_t", "u2"), ("int16_t", "i2")])
@pytest.mark.parametrize("endian", ["little", "big"])
@pytest.mark.parametrize("n_shape", [(33, 21, 17), (64, 16, 4), (1, 1, 1)])
def test_device_loader_conversion_matches_oracle_loader(ctx, tmp_path, ctype, npt, endian, n_shape):
    """vkv_convert_volume (the loader's normalisation on the device) against the oracle's CPU loader on the same raw file."""
    rng = np.random.default_rng(5)
    info = np.iinfo(npt)
    w, h, d = n_shape
    raw = rng.integers(info.min, info.max + 1, size=(d, h, w)).astype(npt)
    f = tmp_path / "v.raw"
    file_bytes = raw.astype(("<" if endian == "little" else ">") + npt)
    file_bytes.tofile(f)
    lo, hi = (-20.0, 100.0) if npt[1] == "1" else (-400.0, 25380.0)
    (tmp_path / "v.raw.header").write_text("%d %d %d\n1 1 1\n%g %g\n%s %s\n1 0 0 0\n" % (w, h, d, lo, hi, ctype, endian))
    expect = O.load_data(str(f), O.load_header(str(f) + ".header"))
    d_raw = torch.from_numpy(np.frombuffer(file_bytes.tobytes(), np.uint8).copy()).cuda()
    d_out = torch.full((d, h, w), 7, dtype=torch.uint8, device="cuda")
    ctx.convert_volume(d_raw.data_ptr(), abi.VOXEL_TYPES[ctype], endian == "big", lo, hi, w * h * d, d_out.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(d_out.cpu().numpy(), expect)


def run_offscreen(tmp_path, *flags):
    assert os.path.exists(EXE), "vkv_offscreen not built (run __graft_entry__.build())"
    out = subprocess.run([EXE, *flags], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()
    return out.stdout.decode()


@pytest.mark.parametrize("skipmode", [0, 1, 2, 3])
def test_offscreen_driver_matches_oracle(tmp_path, skipmode):
    """vkv_offscreen = the reference application's call order in C++.  Its frame must equal the oracle's frame for the exact
    parameter block its VolumeRenderSubpass::draw built (dumped with --dump-params)."""
    w, h = 160, 96
    shape, kind, seed = (72, 60, 48), 1, 11
    run_offscreen(tmp_path, "--synthetic=%dx%dx%d:%d:%d" % (*shape, kind, seed), "--width=%d" % w, "--height=%d" % h, "--skipmode=%d" % skipmode,
                  "--azimuth=40", "--elevation=15", "--dump-counts=counts.raw", "--dump-rgba8=rgba8.raw", "--dump-params=params.raw")
    p = abi.RenderParams.from_buffer_copy(open(tmp_path / "params.raw", "rb").read())
    assert (p.image_width, p.image_height) == (w, h) and p.options.skipping_type == skipmode
    assert p.options.early_ray_termination == 1 and p.options.clip_distance == 50.0  # VolumeRenderSubpass::Options defaults
    assert p.d_packed_volume and p.d_transfer_function_bits
    vol = O.synth_volume(shape, kind, seed)
    opt = abi.VolumeOptions(**T.APP_TF)  # the application's flag defaults (src/volume_render.cpp:67-70)
    tf = O.transfer_function_uniform(opt)
    assert bytes(tf) == bytes(p.transfer_function)
    tex = O.transfer_function_texture(opt)
    grad = O.gradient_map(vol, tf)
    maps = None if skipmode == 0 else O.compute_distance_map(vol, grad, tex, tf, 4, skipmode)
    ref = O.render(p, vol, grad, tex, maps, want_rgba8=True)
    counts = np.fromfile(tmp_path / "counts.raw", np.uint32).reshape(h, w, 3)
    rgba8 = np.fromfile(tmp_path / "rgba8.raw", np.uint8).reshape(h, w, 4)
    assert ref.counts[..., 0].sum() > 0
    assert np.array_equal(counts, ref.counts)
    assert np.array_equal(rgba8, ref.rgba8)


def test_offscreen_benchmark_mode_prints_the_harness_lines(tmp_path):
    """--benchmark=N: ERT off + count output (src/volume_render.cpp:177-183) and the three lines scripts/benchmark.py parses."""
    shape = (64, 64, 64)
    out = run_offscreen(tmp_path, "--synthetic=64x64x64:1:3", "--width=128", "--height=128", "--benchmark=3", "--skipmode=2", "--blocksize=4",
                        "--imin=0.1", "--imax=1.0", "--gmin=0.0", "--gmax=0.2", "--dump-counts=c.raw", "--dump-params=p.raw")
    m_fps = re.search(r"ran [\d]+ frames, averaged ([\d\.e\+]+) fps", out)
    m_map = re.search(r"Updated occupancy/distance map in ([\d\.e\-\+]+)ms", out)
    m_occ = re.search(r"Occupied voxels: ([\d\.e\-\+]+)%", out)
    assert m_fps and m_map and m_occ, out
    vol = O.synth_volume(shape, 1, 3)
    tf = O.transfer_function_uniform(abi.VolumeOptions(**T.APP_TF))
    grad = O.gradient_map(vol, tf)
    expect = 100.0 * np.float32(O.occupied_voxel_count(vol, grad, tf)) / np.float32(vol.size)
    assert abs(float(m_occ.group(1)) - float(expect)) < 1e-3
    p = abi.RenderParams.from_buffer_copy(open(tmp_path / "p.raw", "rb").read())
    assert p.options.early_ray_termination == 0 and p.options.test == abi.TEST_NUM_TEXTURE_SAMPLES and p.options.clip_distance == 1.0


def test_offscreen_loads_a_volume_file(tmp_path):
    """load_from_file: .header side-car + big-endian uint16 raw file (README.md:58-70) through the C++ loader onto the device."""
    rng = np.random.default_rng(1)
    vol16 = (O.synth_volume((48, 40, 32), 1, 8).astype(np.uint16) * 8 + 400)
    vol16.astype(">u2").tofile(tmp_path / "scan.raw")
    (tmp_path / "scan.raw.header").write_text("48 40 32 # extents\n0.001 0.001 0.002 # voxel size\n400.0 2538.0 # normalisation range\n"
                                              "uint16_t big # type\n1 0 0 90 # rotation\n")
    run_offscreen(tmp_path, "scan.raw", "--width=96", "--height=64", "--skipmode=2", "--dump-counts=c.raw", "--dump-params=p.raw")
    p = abi.RenderParams.from_buffer_copy(open(tmp_path / "p.raw", "rb").read())
    h = O.load_header(str(tmp_path / "scan.raw.header"))
    vol = O.load_data(str(tmp_path / "scan.raw"), h)
    assert (p.volume_extent.width, p.volume_extent.height, p.volume_extent.depth) == (48, 40, 32)
    opt = abi.VolumeOptions(**T.APP_TF)
    tf, tex = O.transfer_function_uniform(opt), O.transfer_function_texture(opt)
    grad = O.gradient_map(vol, tf)
    maps = O.compute_distance_map(vol, grad, tex, tf, 4, abi.SKIP_DISTANCE)
    ref = O.render(p, vol, grad, tex, maps)
    counts = np.fromfile(tmp_path / "c.raw", np.uint32).reshape(64, 96, 3)
    assert ref.counts[..., 0].sum() > 0 and np.array_equal(counts, ref.counts)
    del rng


@pytest.mark.parametrize("ranks,skipmode", [(3, 2), (8, 3)])
def test_offscreen_virtual_ranks_assemble_the_frame(tmp_path, ranks, skipmode):
    """The multi-GPU decomposition through the C++ host mirror (VolumeRenderSubpass::rank_schedule -> draw -> vkv_scatter_tiles, INTEGRATION.md
    section 5): one GPU plays every rank in turn; each renders its share of the tiles of the frame's screen rectangle into a compact buffer,
    the owner's de-interleave assembles them and clears the rest - byte-equal to the frame one launch renders."""
    full, assembled = tmp_path / "full.bin", tmp_path / "assembled.bin"
    cmd = [EXE, "--width=330", "--height=200", "--skipmode=%d" % skipmode, "--synthetic=72x64x56:1:7", "--azimuth=35", "--elevation=15",
           "--dump-rgba8=%s" % full, "--virtual-ranks=%d" % ranks, "--dump-assembled=%s" % assembled]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.fromfile(full, np.uint8), np.fromfile(assembled, np.uint8)
    assert a.size == 330 * 200 * 4 and a.any() and np.array_equal(a, b)
    m = re.search(r"rectangle (\d+)x(\d+) tiles at \((\d+), (\d+)\) of (\d+)x(\d+)", r.stdout)
    assert m and int(m.group(1)) * int(m.group(2)) < int(m.group(5)) * int(m.group(6)), r.stdout  # the rectangle is smaller than the frame
