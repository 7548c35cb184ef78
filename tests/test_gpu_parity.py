"""GPU parity: every HIP kernel of the hot path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bar: bit-exact for all uint8 maps and the uint32 sample counters; the float RGBA /
depth outputs are compared with the tolerance COLOR_TOL below (the arithmetic is pinned op-for-op, so the
observed difference is expected to be 0; the tolerance is the contract)."""
import os

import numpy as np
import pytest
import torch

from oracle import vkv_oracle as O
from tests import helpers as T
from vkvolume_amd import abi, lib, volume as V

pytestmark = pytest.mark.gpu

COLOR_TOL = 1e-5  # max abs difference per premultiplied float channel (SURVEY.md §8c suggests 1e-4)
DEPTH_TOL = 1e-6


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_gpu_volume(ctx, scene):
    """Upload the oracle scene's voxels and run the product's precompute chain in the reference's call order."""
    v = V.Volume(ctx)
    v.options = scene.options
    v.load_from_array(scene.vol, scene.block, scene.image_transform)
    v.node_transform = scene.node_transform
    tf = v.get_transfer_function_uniform()
    if v.options.use_precomputed_gradient:
        V.ComputeGradientMap(ctx).compute(v, tf)
    v.update_transfer_function_texture()
    return v, tf


# ------------------------------------------------------------------------------------------------------
# synthetic generator
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,kind,seed", [((64, 64, 64), 0, 1), ((64, 64, 64), 1, 0xC0FFEE02), ((97, 50, 33), 1, 7),
                                             ((130, 70, 41), 0, 3),
                                             # the generator's knobs (kind = 1 | shells << 8 | thickness << 16): 12 shells at 0.7 x, all 40 at 2 x, 3 at the default thickness
                                             ((97, 50, 33), 1 | (12 << 8) | (179 << 16), 7), ((64, 64, 64), 1 | (512 << 16), 0xC0FFEE02), ((70, 60, 50), 1 | (3 << 8) | (9 << 28), 5)])
def test_synth_volume_matches_oracle(ctx, shape, kind, seed):
    w, h, d = shape
    t = torch.empty((d, h, w), dtype=torch.uint8, device="cuda")
    ctx.synth_volume(t.data_ptr(), abi.Extent3D(w, h, d), kind, seed, torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(t.cpu().numpy(), O.synth_volume(shape, kind, seed))


# ------------------------------------------------------------------------------------------------------
# gradient map
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(64, 64, 64), (37, 29, 23), (130, 5, 3), (1, 1, 1), (65, 1, 9), (68, 13, 11), (4, 3, 2), (132, 9, 17), (256, 8, 8)])
@pytest.mark.parametrize("use_gradient", [True, False])
def test_gradient_map_parity(ctx, shape, use_gradient):
    vol = T.random_volume(shape, seed=11)
    opt = abi.VolumeOptions(**T.APP_TF) if use_gradient else abi.VolumeOptions(intensity_min=0.1, gradient_min=0.0, gradient_max=0.0)
    tf = lib.transfer_function_uniform(opt)
    assert bool(tf.use_gradient) == use_gradient
    d_vol, d_grad = dev(vol), torch.empty(vol.shape, dtype=torch.uint8, device="cuda")
    ctx.gradient_map(d_vol.data_ptr(), d_grad.data_ptr(), abi.Extent3D(*shape), tf, torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(d_grad.cpu().numpy(), O.gradient_map(vol, tf))


def test_gradient_kernel_shortcuts_are_exact(ctx):
    """The tiled gradient kernel's two short-cuts, checked exhaustively on the device: (0) its short sqrt (x * rsq(x) and one step on the
    exact residual, no range rescaling) equals the compiler's correctly rounded sqrt for every float its argument can be - 0 and all
    of [2^-90, 16): the sum of the squares of three tap sums in [-2, 2] (the 0.25 factors are applied after the root); (1) its
    one-instruction clamped R8_UNORM store equals rint(clamp(g, 0, 1) * 255) for every non-NaN float."""
    import ctypes as C
    L = ctx._lib
    L.vkv_debug_check.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
    bad = torch.zeros(2, dtype=torch.int64, device="cuda")
    lo, hi = int(np.float32(2.0 ** -90).view(np.uint32)), int(np.float32(16.0).view(np.uint32))
    ctx.check(L.vkv_debug_check(ctx.handle, 0, lo, hi - lo, bad.data_ptr(), None))
    ctx.check(L.vkv_debug_check(ctx.handle, 0, 0, 1, bad.data_ptr(), None))  # x = 0
    inf = 0x7f800000
    ctx.check(L.vkv_debug_check(ctx.handle, 1, 0, inf + 1, bad.data_ptr() + 8, None))  # +0 .. +inf
    ctx.check(L.vkv_debug_check(ctx.handle, 1, 0x80000000, inf + 1, bad.data_ptr() + 8, None))  # -0 .. -inf
    torch.cuda.synchronize()
    assert bad.tolist() == [0, 0]


def test_ray_setup_divisions_are_correctly_rounded(ctx):
    """The ray set-up's divisions run through v_rcp_f32 + the refinement the compiler's own IEEE division uses, shared between the quotients
    of one denominator and without range scaling / special-case fix-up (vkv_device.hpp: recip_exact, div_by).  Checked on the device
    against the IEEE division for EVERY float as the denominator: (2) all reciprocals, (3) quotients with eight hashed numerators per
    denominator.  Operands outside [2^-40, 2^40] never reach these functions (ray_setup sends such waves down the IEEE path:
    the axis-parallel and degenerate-camera cases of test_render_parity_* cover that branch)."""
    L = ctx._lib
    bad = torch.zeros(2, dtype=torch.int64, device="cuda")
    for what in (2, 3):
        for first in (0, 0x80000000):  # positive and negative patterns, each 2^31 of them
            ctx.check(L.vkv_debug_check(ctx.handle, what, first, 1 << 31, bad.data_ptr() + 8 * (what - 2), None))
    torch.cuda.synchronize()
    assert bad.tolist() == [0, 0]


@pytest.mark.parametrize("shape,segment", [((260, 40, 50), 3), ((192, 24, 67), 2), ((132, 9, 33), 10), ((64, 8, 80), 4), ((200, 17, 26), 255)])
def test_gradient_map_marching_workgroups(ctx, shape, segment):
    """The tiled kernel's workgroups march over `segment` tiles in z with the next tile prefetched; volumes this small run one tile per
    workgroup unless VkvTuning.gradient_segment forces the march: interior and edge tiles in x, clamped first / last tiles in
    z, a last segment that is shorter, a depth that is no multiple of the tile."""
    vol = T.random_volume(shape, seed=21 + segment)
    tf = lib.transfer_function_uniform(abi.VolumeOptions(**T.APP_TF))
    d_vol, d_grad = dev(vol), torch.empty(vol.shape, dtype=torch.uint8, device="cuda")
    ctx.set_tuning(gradient_segment=segment)
    try:
        ctx.gradient_map(d_vol.data_ptr(), d_grad.data_ptr(), abi.Extent3D(*shape), tf, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        ctx.set_tuning(gradient_segment=0)
    assert np.array_equal(d_grad.cpu().numpy(), O.gradient_map(vol, tf))


def test_gradient_map_smooth_volume(ctx):
    vol = O.synth_volume((96, 80, 72), 1, 5)
    tf = lib.transfer_function_uniform(abi.VolumeOptions(**T.APP_TF))
    d_vol, d_grad = dev(vol), torch.empty(vol.shape, dtype=torch.uint8, device="cuda")
    ctx.gradient_map(d_vol.data_ptr(), d_grad.data_ptr(), abi.Extent3D(96, 80, 72), tf, 0)
    torch.cuda.synchronize()
    assert np.array_equal(d_grad.cpu().numpy(), O.gradient_map(vol, tf))


# ------------------------------------------------------------------------------------------------------
# occupancy + distance maps
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,block", [((64, 64, 64), 4), ((61, 45, 30), 4), ((50, 33, 21), 3), ((40, 40, 40), 2),
                                         ((67, 31, 18), 5), ((48, 48, 48), 6), ((20, 9, 7), 1), ((68, 30, 21), 4), ((132, 17, 9), 2), ((12, 5, 3), 1),
                                         ((1028, 9, 6), 4),
                                         # dword-aligned rows with block widths that are not 1, 2 or 4 (k_occupancy_map_dword_any), several workgroups per row
                                         ((52, 33, 21), 3), ((2052, 7, 5), 3), ((1300, 6, 7), 5), ((1040, 5, 4), 7), ((2048, 1, 1), 100), ((2056, 1, 1), 130)])
@pytest.mark.parametrize("variant", ["precomputed", "on_the_fly", "no_gradient"])
def test_occupancy_map_parity(ctx, shape, block, variant):
    vol = T.random_volume(shape, seed=5, sparsity=0.97 if block < 50 else 0.995)  # big cells: fewer candidates, else no cell is empty
    if variant == "no_gradient":
        opt = abi.VolumeOptions(intensity_min=0.1, gradient_min=0.0, gradient_max=0.0)
    else:
        opt = abi.VolumeOptions(use_precomputed_gradient=(variant == "precomputed"), **T.APP_TF)
    scene = T.OracleScene(vol, opt, block)
    expect = O.occupancy_map(scene.vol, scene.grad, scene.tex, scene.tf, block)
    assert 0 < (expect == 0).mean() < 1, "test volume must give a mixed occupancy map"
    v, tf = make_gpu_volume(ctx, scene)
    d_map = torch.empty(expect.shape, dtype=torch.uint8, device="cuda")
    grad = v.gradient if opt.use_precomputed_gradient else None
    ctx.occupancy_map(v.volume.data_ptr(), None if grad is None else grad.data_ptr(), v.transfer_function.data_ptr(), tf,
                      v.extent, d_map.data_ptr(), v.map_extent, torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(d_map.cpu().numpy(), expect)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "16"))))
def test_gradient_and_occupancy_fuzz(ctx, seed):
    """Random volume extents (widths around the dword / 64-voxel tile / 1024-voxel row boundaries of the kernels), content, transfer function
    window, gradient variant and block size: the gradient map and the occupancy map, every byte against the oracle."""
    rng = np.random.default_rng(9000 + seed)
    edges = [1, 2, 3, 4, 5, 8, 9, 63, 64, 65, 68, 127, 128, 132, 255, 256, 260, 1024, 1028, 2052]
    while True:
        shape = tuple(int(rng.choice(edges)) if rng.random() < 0.4 else int(rng.integers(1, 150)) for _ in range(3))  # w, h, d
        if shape[0] * shape[1] * shape[2] <= 1200000:
            break
    kind = int(rng.integers(0, 3))
    vol = T.random_volume(shape, seed=seed, sparsity=float(rng.uniform(0.5, 0.999))) if kind == 2 else O.synth_volume(shape, kind, int(rng.integers(1, 1 << 30)))
    variant = ("precomputed", "on_the_fly", "no_gradient")[int(rng.integers(0, 3))]
    imin = float(rng.uniform(0.0, 0.5))
    tfo = dict(intensity_min=imin, intensity_max=float(rng.uniform(imin + 0.02, 1.0)))
    if variant == "no_gradient":
        opt = abi.VolumeOptions(gradient_min=0.0, gradient_max=0.0, **tfo)
    else:
        gmin = float(rng.uniform(0.0, 0.2))
        opt = abi.VolumeOptions(gradient_min=gmin, gradient_max=float(rng.uniform(gmin + 0.02, 0.8)), use_precomputed_gradient=(variant == "precomputed"), **tfo)
    block = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 6, 7, 8, 9, 33]))
    scene = T.OracleScene(vol, opt, block)
    v, tf = make_gpu_volume(ctx, scene)
    what = "shape %s block %d %s" % (shape, block, variant)
    if opt.use_precomputed_gradient:
        assert np.array_equal(v.gradient.cpu().numpy(), scene.grad), "gradient map, " + what
    expect = O.occupancy_map(scene.vol, scene.grad, scene.tex, scene.tf, block)
    d_map = torch.empty(expect.shape, dtype=torch.uint8, device="cuda")
    grad = v.gradient if opt.use_precomputed_gradient else None
    ctx.occupancy_map(v.volume.data_ptr(), None if grad is None else grad.data_ptr(), v.transfer_function.data_ptr(), tf,
                      v.extent, d_map.data_ptr(), v.map_extent, torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(d_map.cpu().numpy(), expect), "occupancy map, " + what
    # the occupied-voxel count of the reference's benchmark mode (analytic transfer function, src/compute_occupied_voxel_count.cpp)
    d_count = torch.full((1,), 12345, dtype=torch.int64, device="cuda")
    ctx.occupied_voxel_count(v.volume.data_ptr(), None if grad is None else grad.data_ptr(), tf, v.extent, d_count.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert int(d_count.item()) == O.occupied_voxel_count(scene.vol, scene.grad, scene.tf), "occupied voxel count, " + what


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "16"))))
def test_occupancy_map_extents_beyond_the_volume(ctx, seed):
    """the ABI accepts any map extent <= the volume's; block = ceil(extent / map extent) can then leave whole cell rows / slices
    outside the volume (16^3 voxels under a 7^3 map: block 3, cells 6 of y and z hold no voxel).  Such cells stay EMPTY (the reference's loop
    over a cell's voxels is clipped at the volume edge, occupancy_map.comp:52-53) - in the wave kernel, the workgroup-per-cell-row kernels
    (VkvTuning.occupancy_kernel = 1) and the oracle alike, and nothing is read past the volume."""
    rng = np.random.default_rng(4200 + seed)
    if seed == 0:
        shape, me = (16, 16, 16), abi.Extent3D(7, 7, 7)
    elif seed == 1:
        shape, me = (256, 33, 21), abi.Extent3D(100, 12, 8)
    else:
        shape = (int(rng.choice([16, 64, 68, 132, 256, 260])), int(rng.integers(5, 70)), int(rng.integers(5, 50)))
        me = abi.Extent3D(*(int(rng.integers(max(1, -(-n // int(rng.integers(2, 7)))), n + 1)) for n in shape))
    vol = T.random_volume(shape, seed=seed, sparsity=0.9)
    variant = ("precomputed", "no_gradient")[seed % 2]
    opt = abi.VolumeOptions(intensity_min=0.1, gradient_min=0.0, gradient_max=0.0) if variant == "no_gradient" else abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(vol, opt, 4)
    expect = O.occupancy_map(scene.vol, scene.grad, scene.tex, scene.tf, 0, map_extent_override=me)
    v, tf = make_gpu_volume(ctx, scene)
    st = torch.cuda.current_stream().cuda_stream
    t0 = ctx.get_tuning()
    got = {}
    try:
        for kernel in (0, 1):
            ctx.set_tuning(occupancy_kernel=kernel)
            d_map = torch.full(expect.shape, 77, dtype=torch.uint8, device="cuda")
            ctx.occupancy_map(v.volume.data_ptr(), v.gradient.data_ptr() if variant == "precomputed" else None, v.transfer_function.data_ptr(), tf,
                              v.extent, d_map.data_ptr(), me, st)
            got[kernel] = d_map.cpu().numpy()
    finally:
        ctx.set_tuning(occupancy_kernel=t0.occupancy_kernel)
    what = "shape %s map %s %s" % (shape, me.as_tuple(), variant)
    assert np.array_equal(got[0], expect), "wave kernel, " + what
    assert np.array_equal(got[1], expect), "row kernels, " + what


def sparse_occupancy(shape_dhw, seed, p):
    rng = np.random.default_rng(seed)
    return np.where(rng.random(shape_dhw) < p, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape_dhw,p", [((16, 16, 16), 0.02), ((9, 10, 13), 0.01), ((5, 70, 3), 0.01), ((40, 33, 130), 0.0005),
                                         ((1, 1, 1), 1.0), ((7, 7, 7), 0.0), ((3, 300, 2), 0.002), ((600, 2, 3), 0.0006), ((2, 700, 1), 0.0015),
                                         ((300, 290, 70), 0.00001), ((64, 64, 64), 0.3),
                                         # rows > 1024 cells (serial x scan), axes of several chunks, widths around the kernel's line groups
                                         ((2, 3, 1100), 0.002), ((2, 1500, 3), 0.001), ((1300, 2, 17), 0.0004), ((30, 130, 33), 0.001),
                                         ((3, 5, 1024), 0.0008), ((2, 2, 257), 0.004),
                                         # 16 | width and 128 < axis <= 256: the 16-byte vector staging of the y / z passes
                                         ((200, 130, 32), 0.0002), ((3, 140, 48), 0.002), ((131, 2, 16), 0.004),
                                         # 8 | width and 256 < axis <= 512: whole-line table with 8-byte vectors
                                         ((2, 300, 24), 0.001), ((400, 3, 8), 0.001),
                                         # the wave-per-row x pass: 8 and 16 cells per lane, dword and byte rows, a last lane that is partly / wholly past the row
                                         ((3, 4, 260), 0.003), ((2, 3, 700), 0.002), ((2, 2, 1023), 0.002), ((4, 3, 512), 0.001), ((2, 5, 61), 0.01), ((3, 2, 1021), 0.0004)])
def test_distance_map_parity(ctx, shape_dhw, p):
    occ = sparse_occupancy(shape_dhw, 3, p)
    d, h, w = shape_dhw
    d_map, d_swap = dev(occ), torch.empty(shape_dhw, dtype=torch.uint8, device="cuda")
    ctx.distance_map(d_map.data_ptr(), d_swap.data_ptr(), abi.Extent3D(w, h, d), torch.cuda.current_stream().cuda_stream)
    got = d_map.cpu().numpy()
    assert np.array_equal(got, O.distance_map(occ))
    if occ.size <= 20000:
        assert np.array_equal(got, T.brute_force_chebyshev(occ))


@pytest.mark.parametrize("shape_dhw", [(6, 7, 50), (3, 5, 260), (2, 3, 1000), (2, 2, 1300), (20, 24, 32), (3, 70, 13)])
def test_distance_transforms_on_arbitrary_bytes(ctx, shape_dhw):
    """Stage 0 of both shaders is the recurrence g = min(g_prev + 1, input) (distance_map.comp:57-71, distance_map_anisotropic.comp:44-53),
    stages 1 / 2 the max-based search - all defined for ANY byte input, not only a 0 / 255 occupancy map.  The kernels follow them there
    too (the x pass as x + running minimum of g(q) - q): random bytes with a few small values, both transforms against the oracle."""
    rng = np.random.default_rng(sum(shape_dhw))
    raw = rng.integers(0, 256, size=shape_dhw, dtype=np.uint8)
    raw[rng.random(shape_dhw) < 0.7] = 255
    raw[rng.random(shape_dhw) < 0.01] = 0
    d, h, w = shape_dhw
    st = torch.cuda.current_stream().cuda_stream
    d_map, d_swap = dev(raw), torch.empty(shape_dhw, dtype=torch.uint8, device="cuda")
    ctx.distance_map(d_map.data_ptr(), d_swap.data_ptr(), abi.Extent3D(w, h, d), st)
    assert np.array_equal(d_map.cpu().numpy(), O.distance_map(raw))
    maps = [torch.empty(shape_dhw, dtype=torch.uint8, device="cuda") for _ in range(8)]
    maps[7].copy_(dev(raw))
    ctx.distance_map_anisotropic([m.data_ptr() for m in maps], d_swap.data_ptr(), abi.Extent3D(w, h, d), st)
    expect = O.distance_map_anisotropic(raw)
    for k in range(8):
        assert np.array_equal(maps[k].cpu().numpy(), expect[k]), "octant %d" % k


@pytest.mark.parametrize("shape_dhw,p", [((16, 16, 16), 0.02), ((9, 10, 13), 0.01), ((5, 70, 3), 0.01), ((40, 33, 130), 0.0005),
                                         ((1, 1, 1), 1.0), ((6, 6, 6), 0.0), ((600, 2, 3), 0.0006), ((2, 700, 1), 0.0015), ((90, 280, 70), 0.00002),
                                         ((2, 3, 1100), 0.002), ((2, 1500, 3), 0.001), ((1300, 2, 17), 0.0004), ((30, 130, 33), 0.001),
                                         ((3, 5, 1024), 0.0008), ((2, 2, 257), 0.004),
                                         ((200, 130, 32), 0.0002), ((3, 140, 48), 0.002), ((131, 2, 16), 0.004), ((2, 300, 24), 0.001), ((400, 3, 8), 0.001),
                                         ((3, 4, 260), 0.003), ((2, 3, 700), 0.002), ((2, 2, 1023), 0.002), ((4, 3, 512), 0.001), ((2, 5, 61), 0.01)])
def test_distance_map_anisotropic_parity(ctx, shape_dhw, p):
    occ = sparse_occupancy(shape_dhw, 4, p)
    d, h, w = shape_dhw
    maps = [torch.empty(shape_dhw, dtype=torch.uint8, device="cuda") for _ in range(8)]
    maps[7].copy_(dev(occ))
    swap = torch.empty(shape_dhw, dtype=torch.uint8, device="cuda")
    ctx.distance_map_anisotropic([m.data_ptr() for m in maps], swap.data_ptr(), abi.Extent3D(w, h, d),
                                 torch.cuda.current_stream().cuda_stream)
    expect = O.distance_map_anisotropic(occ)
    for k in range(8):
        got = maps[k].cpu().numpy()
        assert np.array_equal(got, expect[k]), "octant %d" % k
        if occ.size <= 3000:
            assert np.array_equal(got, T.brute_force_chebyshev_octant(occ, k)), "octant %d vs brute force" % k


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "16"))))
def test_distance_transform_fuzz(ctx, seed):
    """Random map extents across every launch configuration of the transforms (line groups of 16 / 8 / 4, whole lines up to 512 cells, chunked
    longer ones, rows beyond 1024 cells, odd widths and unaligned rows), random density, sometimes arbitrary bytes: both transforms against
    the oracle, every cell."""
    rng = np.random.default_rng(7000 + seed)
    edges = [1, 2, 3, 7, 16, 17, 61, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1024, 1025, 1400]
    while True:
        dims = [int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(1, 200)) for _ in range(3)]
        if dims[0] * dims[1] * dims[2] <= 1500000:
            break
    shape_dhw = tuple(dims)
    cells = dims[0] * dims[1] * dims[2]
    p = float(min(1.0, rng.choice([0.0, 1.0, 3.0, 30.0, 3000.0]) / cells + rng.choice([0.0, 0.0, 0.001, 0.05])))
    occ = np.where(rng.random(shape_dhw) < p, 0, 255).astype(np.uint8)
    if rng.random() < 0.25:        # any byte input is defined (test_distance_transforms_on_arbitrary_bytes)
        raw = rng.integers(0, 256, size=shape_dhw, dtype=np.uint8)
        occ = np.where(rng.random(shape_dhw) < 0.3, raw, occ).astype(np.uint8)
    d, h, w = shape_dhw
    st = torch.cuda.current_stream().cuda_stream
    d_map, d_swap = dev(occ), torch.empty(shape_dhw, dtype=torch.uint8, device="cuda")
    ctx.distance_map(d_map.data_ptr(), d_swap.data_ptr(), abi.Extent3D(w, h, d), st)
    assert np.array_equal(d_map.cpu().numpy(), O.distance_map(occ)), "isotropic, shape %s" % (shape_dhw,)
    maps = [torch.empty(shape_dhw, dtype=torch.uint8, device="cuda") for _ in range(8)]
    maps[7].copy_(dev(occ))
    ctx.distance_map_anisotropic([m.data_ptr() for m in maps], d_swap.data_ptr(), abi.Extent3D(w, h, d), st)
    expect = O.distance_map_anisotropic(occ)
    for k in range(8):
        assert np.array_equal(maps[k].cpu().numpy(), expect[k]), "octant %d, shape %s" % (k, shape_dhw)


@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_BLOCK, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
def test_compute_distance_map_chain(ctx, skipping_type):
    """ComputeDistanceMap::compute end to end (occupancy -> transform) on a synthetic shell volume."""
    scene = T.OracleScene(O.synth_volume((72, 64, 56), 1, 9), abi.VolumeOptions(**T.APP_TF), 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    expect = scene.maps(skipping_type)
    assert len(v.distance_maps) == expect.shape[0]
    for k in range(expect.shape[0]):
        assert np.array_equal(v.distance_maps[k].cpu().numpy(), expect[k]), "map %d" % k


# ------------------------------------------------------------------------------------------------------
# ray-march integrator
# ------------------------------------------------------------------------------------------------------
def gpu_render(ctx, v, params, want_rgba8=False):
    if params.tiles.compact:
        shape = (params.tiles.tile_count * params.tiles.tile_width * params.tiles.tile_height,)
    else:
        shape = (params.image_height, params.image_width)
    # image-shaped outputs start from sentinels (every pixel must be written); compact buffers have padding
    # pixels in partial edge tiles that nobody writes, so they start from zero like the oracle's
    s_col, s_cnt = (0.0, 0) if params.tiles.compact else (-1.0, 0xFFFF)
    color = torch.full(shape + (4,), s_col, dtype=torch.float32, device="cuda")
    counts = torch.full(shape + (3,), s_cnt, dtype=torch.int32, device="cuda")
    depth = torch.full(shape, s_col, dtype=torch.float32, device="cuda")
    rgba8 = torch.zeros(shape + (4,), dtype=torch.uint8, device="cuda") if want_rgba8 else None
    sp = V.VolumeRenderSubpass(ctx, v, params.options, (params.image_width, params.image_height))
    outs = []
    # every combination of {bricked image + TF bit table, plain linear buffers + texel fetch} x {persistent-wave scheduler,
    # static tile scheduler} must give the same bits
    for packed, sched in ((True, "persistent"), (False, "persistent"), (True, "tiles"), (False, "tiles")):
        v.use_packed = packed
        ctx.set_tuning(scheduler=1 if sched == "persistent" else 0)
        p = sp.bind(params)
        assert bool(p.d_packed_volume) == packed and bool(p.d_transfer_function_bits) == packed
        c, n, d = color.clone(), counts.clone(), depth.clone()
        q = None if rgba8 is None else rgba8.clone()
        sp.draw(p, c, q, n, d)
        torch.cuda.synchronize()
        outs.append((c.cpu().numpy(), n.cpu().numpy().astype(np.uint32), d.cpu().numpy(), None if q is None else q.cpu().numpy()))
    v.use_packed = True
    ctx.set_tuning(scheduler=0)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert (a is None and b is None) or np.array_equal(a, b), "sampling-layout / scheduler variants disagree"
    return outs[0]


def compare_render(got, ref, label):
    color, counts, depth, _ = got
    assert np.array_equal(counts, ref.counts), "%s: sample counters differ in %d pixels" % (
        label, int((counts != ref.counts).any(-1).sum()))
    dc = float(np.abs(color - ref.color).max())
    dd = float(np.abs(depth - ref.depth).max())
    assert dc <= COLOR_TOL, "%s: colour max abs diff %g" % (label, dc)
    assert dd <= DEPTH_TOL, "%s: depth max abs diff %g" % (label, dd)
    return dc, dd


@pytest.fixture(scope="module")
def shell_scene(ctx):
    scene = T.OracleScene(O.synth_volume((96, 80, 72), 1, 0xC0FFEE02), abi.VolumeOptions(**T.APP_TF), 4)
    v, tf = make_gpu_volume(ctx, scene)
    cdm = V.ComputeDistanceMap(ctx)
    cdm.compute(v, tf, abi.SKIP_ANISOTROPIC_DISTANCE)  # 8 maps; map 0 is overwritten per mode below
    return scene, v, tf, cdm


@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_BLOCK, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
@pytest.mark.parametrize("ert", [True, False])
def test_render_parity_modes(ctx, shell_scene, skipping_type, ert):
    scene, v, tf, cdm = shell_scene
    cdm.compute(v, tf, skipping_type)
    size = (160, 96)
    worst = 0.0
    for az in (0.0, 33.0, 90.0, 201.0):
        view, proj = T.orbit(az, image_size=size)
        opts = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0, early_ray_termination=ert)
        params = scene.params(view, proj, size, opts)
        ref = scene.render(params)
        assert ref.counts[..., 0].sum() > 0
        dc, _ = compare_render(gpu_render(ctx, v, params), ref, "mode %d ert %d az %g" % (skipping_type, ert, az))
        worst = max(worst, dc)
    print("max colour diff", worst)


@pytest.mark.parametrize("variant", ["on_the_fly", "no_gradient"])
@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_DISTANCE])
def test_render_parity_gradient_variants(ctx, variant, skipping_type):
    vol = O.synth_volume((64, 56, 48), 1, 21)
    opt = abi.VolumeOptions(intensity_min=0.1, gradient_min=0.0, gradient_max=0.0) if variant == "no_gradient" else abi.VolumeOptions(use_precomputed_gradient=False, **T.APP_TF)
    scene = T.OracleScene(vol, opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (96, 64)
    view, proj = T.orbit(40.0, image_size=size)
    params = scene.params(view, proj, size, abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0))
    compare_render(gpu_render(ctx, v, params), scene.render(params), variant)


@pytest.mark.parametrize("sampling_factor,alpha_factor,block", [(0.5, 1.0, 4), (2.0, 0.7, 3), (3.0, 2.0, 5), (1.0, 0.0, 2)])
def test_render_parity_sampling_and_blocks(ctx, sampling_factor, alpha_factor, block):
    vol = O.synth_volume((70, 61, 45), 1, 33)
    opt = abi.VolumeOptions(sampling_factor=sampling_factor, voxel_alpha_factor=alpha_factor, **T.APP_TF)
    scene = T.OracleScene(vol, opt, block, voxel_size=(0.0003, 0.0003, 0.0007), axis_angle=(1, 0, 0, 90))
    v, tf = make_gpu_volume(ctx, scene)
    for st in (abi.SKIP_BLOCK, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE):
        V.ComputeDistanceMap(ctx).compute(v, tf, st)
        size = (112, 80)
        view, proj = T.orbit(135.0, elevation=-15.0, image_size=size)
        params = scene.params(view, proj, size, abi.RenderOptions(skipping_type=st, clip_distance=1.0))
        compare_render(gpu_render(ctx, v, params), scene.render(params), "sf %g block %d mode %d" % (sampling_factor, block, st))


def test_render_camera_inside_volume_and_axis_aligned(ctx, shell_scene):
    """Clip plane cuts the box (the plane-intersection vertex shader's case) and exactly axis-parallel rays
    (SURVEY.md Appendix B 12)."""
    scene, v, tf, cdm = shell_scene
    cdm.compute(v, tf, abi.SKIP_DISTANCE)
    size = (96, 96)
    for radius, clip in ((20.0, 5.0), (55.0, 30.0)):
        view, proj = T.orbit(0.0, elevation=0.0, radius=radius, image_size=size)
        params = scene.params(view, proj, size, abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=clip))
        ref = scene.render(params)
        assert (ref.counts[..., 0] > 0).mean() > 0.5
        compare_render(gpu_render(ctx, v, params), ref, "inside r=%g" % radius)


@pytest.mark.parametrize("test", [abi.TEST_RAY_ENTRY, abi.TEST_RAY_EXIT, abi.TEST_NUM_TEXTURE_SAMPLES])
def test_render_test_modes(ctx, shell_scene, test):
    scene, v, tf, cdm = shell_scene
    cdm.compute(v, tf, abi.SKIP_DISTANCE)
    size = (96, 64)
    view, proj = T.orbit(60.0, image_size=size)
    # the reference's benchmark configuration: ERT off, count output (src/volume_render.cpp:177-183)
    opts = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0, early_ray_termination=False, test=test)
    params = scene.params(view, proj, size, opts)
    ref = scene.render(params)
    color, counts, depth, _ = gpu_render(ctx, v, params)
    assert np.array_equal(color, ref.color)  # entry/exit coordinates and the count grey level: exact
    assert np.array_equal(counts, ref.counts)


def test_render_compact_tiles_and_rgba8(ctx, shell_scene):
    """Interleaved tile schedule of rank 1 of 3 into a compact buffer + RGBA8 quantisation, then the root-side
    de-interleave of three ranks' buffers (vkv_scatter_tiles)."""
    scene, v, tf, cdm = shell_scene
    cdm.compute(v, tf, abi.SKIP_DISTANCE)
    size = (150, 70)  # not a multiple of the tile size: edge tiles are partial
    view, proj = T.orbit(300.0, image_size=size)
    opts = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    full = scene.params(view, proj, size, opts, tiles=abi.full_frame_tiles(size[0], size[1], 32, 16))
    ref_full = scene.render(full, want_rgba8=True)
    world = 3
    per_rank = []
    tiles_per_rank = 0
    for rank in range(world):
        tiles = abi.full_frame_tiles(size[0], size[1], 32, 16, rank, world, compact=True)
        tiles_per_rank = max(tiles_per_rank, tiles.tile_count)
        params = scene.params(view, proj, size, opts, tiles=tiles)
        ref = scene.render(params, want_rgba8=True)
        color, counts, depth, rgba8 = gpu_render(ctx, v, params, want_rgba8=True)
        inside = ref.counts[..., 0] + ref.counts[..., 1] > 0
        assert np.array_equal(counts, ref.counts)
        assert np.abs(color - ref.color).max() <= COLOR_TOL
        assert np.array_equal(rgba8[inside], ref.rgba8[inside])
        per_rank.append(rgba8)
    gathered = np.zeros((world, tiles_per_rank * 32 * 16, 4), np.uint8)
    for r, buf in enumerate(per_rank):
        gathered[r, :buf.shape[0]] = buf
    d_g, d_img = dev(gathered), torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
    ctx.scatter_tiles(d_g.data_ptr(), d_img.data_ptr(), size, (32, 16), world, tiles_per_rank, 4, torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(d_img.cpu().numpy(), ref_full.rgba8)
    # round 6: the same frame through the schedule of its screen rectangle (vkv_screen_tile_rect): fewer tiles per rank, the same image - on the
    # device (non-compact: pixels outside the rectangle are not touched) and against the oracle's rendering of the same schedule
    rect = lib.screen_tile_rect(full.ray_cast, full.ray_gen, size, (32, 16))
    assert 0 < rect.tiles < 5 * 5
    tpr = -(-rect.tiles // world)
    gathered = np.zeros((world, tpr * 32 * 16, 4), np.uint8)
    whole = np.full((size[1], size[0], 4), 7, np.uint8)
    for rank in range(world):
        tiles = abi.full_frame_tiles(size[0], size[1], 32, 16, rank, world, compact=True, rect=rect)
        params = scene.params(view, proj, size, opts, tiles=tiles)
        ref = scene.render(params, want_rgba8=True)
        color, counts, depth, rgba8 = gpu_render(ctx, v, params, want_rgba8=True)
        assert np.array_equal(counts, ref.counts)
        inside = ref.counts[..., 0] + ref.counts[..., 1] > 0
        assert np.array_equal(rgba8[inside], ref.rgba8[inside])
        gathered[rank, :rgba8.shape[0]] = rgba8
    d_g = dev(gathered)
    d_img.fill_(9)
    ctx.scatter_tiles(d_g.data_ptr(), d_img.data_ptr(), size, (32, 16), world, tpr, 4, torch.cuda.current_stream().cuda_stream, rect=rect)
    assert np.array_equal(d_img.cpu().numpy(), ref_full.rgba8)
    # one rank, image-indexed outputs: the launch writes the rectangle's pixels only
    params = scene.params(view, proj, size, opts, tiles=abi.full_frame_tiles(size[0], size[1], 32, 16, rect=rect))
    d_out = dev(whole)
    sp = V.VolumeRenderSubpass(ctx, v, opts, size)
    sp.draw(sp.bind(params), rgba8=d_out)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    ys, xs = slice(rect.y0 * 16, min(size[1], (rect.y0 + rect.h) * 16)), slice(rect.x0 * 32, min(size[0], (rect.x0 + rect.w) * 32))
    assert np.array_equal(got[ys, xs], ref_full.rgba8[ys, xs])
    mask = np.ones(size[::-1], bool)
    mask[ys, xs] = False
    assert (got[mask] == 7).all() and not ref_full.rgba8[mask].any()


def test_render_error_paths(ctx, shell_scene):
    scene, v, tf, cdm = shell_scene
    size = (64, 64)
    view, proj = T.orbit(0.0, image_size=size)
    sp = V.VolumeRenderSubpass(ctx, v, abi.RenderOptions(clip_distance=1.0), size)
    p = sp.make_params(view, proj)
    out = torch.zeros((64, 64, 4), dtype=torch.float32, device="cuda")
    p.d_out_color = out.data_ptr()
    assert ctx.render_rc(p) == 0
    bad = abi.RenderParams.from_buffer_copy(p)
    bad.options.depth_attachment = 1
    assert ctx.render_rc(bad) == abi.VKV_E_INVALID_ARGUMENT and "depth_attachment" in ctx.last_error()  # needs d_in_depth
    bad = abi.RenderParams.from_buffer_copy(p)
    bad.d_volume = None
    assert ctx.render_rc(bad) == abi.VKV_E_INVALID_ARGUMENT
    bad = abi.RenderParams.from_buffer_copy(p)
    bad.tiles.tile_width = 12
    assert ctx.render_rc(bad) == abi.VKV_E_INVALID_ARGUMENT
    bad = abi.RenderParams.from_buffer_copy(p)
    bad.options.skipping_type = 9
    assert ctx.render_rc(bad) == abi.VKV_E_INVALID_ARGUMENT
    bad = abi.RenderParams.from_buffer_copy(p)
    bad.d_out_color = None
    assert ctx.render_rc(bad) == abi.VKV_E_INVALID_ARGUMENT
    with pytest.raises(lib.VkvError):
        ctx.distance_map(out.data_ptr(), out.data_ptr(), abi.Extent3D(4, 4, 4))  # aliased buffers
    with pytest.raises(lib.VkvError):
        ctx.distance_map(out.data_ptr(), out.data_ptr() + 64, abi.Extent3D(4096, 4, 4))  # axis > 2048


# ------------------------------------------------------------------------------------------------------
# depth attachment + compositing (frag:122-165, blend state volume_render_subpass.cpp:176-190)
# ------------------------------------------------------------------------------------------------------
def scene_depth_plane(params, size, cut):
    """A synthetic scene depth buffer (reverse-Z): a wall in front of part of the volume.  Depth of a point at view distance
    d with near n (far >> n) is ~ n / d; left third: nothing (0 = far), middle: a wall through the volume, right: a wall in
    front of the whole volume (every fragment there is discarded)."""
    w, h = size
    depth = np.zeros((h, w), np.float32)
    depth[:, w // 3: 2 * w // 3] = cut
    depth[:, 2 * w // 3:] = 0.5
    return depth


@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_DISTANCE])
@pytest.mark.parametrize("blend", [False, True])
def test_render_depth_attachment_and_blending(ctx, shell_scene, skipping_type, blend):
    scene, v, tf, cdm = shell_scene
    cdm.compute(v, tf, skipping_type)
    size = (144, 80)
    view, proj = T.orbit(25.0, image_size=size)
    opts = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0, depth_attachment=True)
    params = scene.params(view, proj, size, opts)
    # reverse-Z depth of the volume centre is near / distance = 0.1 / 150
    in_depth = scene_depth_plane(params, size, 0.1 / 150.0)
    rng = np.random.default_rng(2)
    tgt_color = rng.random((size[1], size[0], 4), dtype=np.float32) if blend else None
    tgt_rgba8 = rng.integers(0, 256, (size[1], size[0], 4), dtype=np.uint8) if blend else None
    ref = scene.render(params, in_depth=in_depth, target_color=tgt_color, target_rgba8=tgt_rgba8, want_rgba8=True)
    plain = scene.render(scene.params(view, proj, size, abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0)))
    third = size[0] // 3
    # left third: scene depth = far, nothing is clipped -> same sample counters as without the depth attachment
    assert np.array_equal(ref.counts[:, :third], plain.counts[:, :third])
    # middle: rays stop at the wall -> fewer or equal samples, strictly fewer in total; right: every fragment discarded
    # (only in aggregate: the reference reconstructs the depth-buffer point with x and y scaled by the depth ratio (frag:153), so
    # it is not on the pixel's own ray and single rays can even get a few more steps)
    assert ref.counts[:, third:2 * third, 0].sum() < 0.7 * plain.counts[:, third:2 * third, 0].sum()
    assert ref.counts[:, 2 * third:].sum() == 0
    if blend:
        assert np.array_equal(ref.color[:, 2 * third:], tgt_color[:, 2 * third:])  # untouched where there is no fragment
        assert np.array_equal(ref.rgba8[:, 2 * third:], tgt_rgba8[:, 2 * third:])
    sp = V.VolumeRenderSubpass(ctx, v, opts, size)
    p = sp.bind(params)
    color = torch.from_numpy(tgt_color).cuda() if blend else torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda")
    rgba8 = torch.from_numpy(tgt_rgba8).cuda() if blend else torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
    counts = torch.full((size[1], size[0], 3), 77, dtype=torch.int32, device="cuda")
    depth = torch.full((size[1], size[0]), -1.0, dtype=torch.float32, device="cuda")
    sp.draw(p, color, rgba8, counts, depth, in_depth=dev(in_depth), blend=blend)
    torch.cuda.synchronize()
    assert np.array_equal(counts.cpu().numpy().astype(np.uint32), ref.counts)
    assert np.abs(color.cpu().numpy() - ref.color).max() <= COLOR_TOL
    assert np.array_equal(rgba8.cpu().numpy(), ref.rgba8)
    got_depth = depth.cpu().numpy()
    frag = got_depth != -1.0
    assert np.abs(got_depth[frag] - ref.depth[frag]).max() <= DEPTH_TOL
    if not blend:
        assert frag.all() and np.array_equal(got_depth[:, 2 * third:], in_depth[:, 2 * third:])  # pass-through where discarded


# ------------------------------------------------------------------------------------------------------
# randomised configurations: every knob of the path drawn at random, HIP vs oracle
# ------------------------------------------------------------------------------------------------------
def fuzz_case(ctx, seed):
    """One random configuration of the ray-march path (test_render_fuzz, tests/test_gpu_fuzz.py::test_launch_variants_fuzz): volume shape /
    content, voxel size and rotation, TF window, sampling and alpha factors, block size, skipping mode, ERT, gradient variant, clip distance,
    camera (sometimes inside the box, for seeds >= 24 beside it) and frame size.  Returns the oracle scene, the device volume, the parameter
    block, the oracle's frame and a label."""
    rng = np.random.default_rng(1000 + seed)
    shape = tuple(int(x) for x in rng.integers(5, 46, size=3))  # w, h, d
    kind = int(rng.integers(0, 3))
    if kind == 2:
        vol = T.random_volume(shape, seed=seed, sparsity=float(rng.uniform(0.6, 0.995)))
    else:
        vol = O.synth_volume(shape, kind, int(rng.integers(1, 1 << 30)))
    grad_variant = ("precomputed", "on_the_fly", "no_gradient")[int(rng.integers(0, 3))]
    imin = float(rng.uniform(0.0, 0.4))
    tfo = dict(intensity_min=imin, intensity_max=float(rng.uniform(imin + 0.05, 1.0)), sampling_factor=float(rng.choice([0.5, 1.0, 1.0, 1.7, 3.0])),
               voxel_alpha_factor=float(rng.choice([0.3, 1.0, 1.0, 2.5])))
    if grad_variant == "no_gradient":
        opt = abi.VolumeOptions(gradient_min=0.0, gradient_max=0.0, **tfo)
    else:
        gmin = float(rng.uniform(0.0, 0.1))
        opt = abi.VolumeOptions(gradient_min=gmin, gradient_max=float(rng.uniform(gmin + 0.05, 0.6)), use_precomputed_gradient=(grad_variant == "precomputed"), **tfo)
    block = int(rng.integers(1, 8))
    voxel = tuple(float(x) for x in rng.uniform(0.2, 2.0, size=3))
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    scene = T.OracleScene(vol, opt, block, voxel_size=voxel, axis_angle=(float(axis[0]), float(axis[1]), float(axis[2]), float(rng.uniform(0, 360))))
    v, tf = make_gpu_volume(ctx, scene)
    st = int(rng.integers(0, 4))
    V.ComputeDistanceMap(ctx).compute(v, tf, st)
    size = (int(rng.integers(3, 9)) * 16 - int(rng.integers(0, 16)), int(rng.integers(2, 6)) * 16 - int(rng.integers(0, 16)))
    radius = float(rng.choice([30.0, 60.0, 110.0, 150.0, 260.0]))  # the node is scaled to 100 units: 30 / 60 put the camera inside or at the box
    view, proj = T.orbit(float(rng.uniform(0, 360)), elevation=float(rng.uniform(-80, 80)), radius=radius, fov=float(rng.uniform(25, 100)), image_size=size)
    if seed >= 24:
        # the camera orbits (and looks at) a point beside the volume: the box is off-centre, partly or wholly outside the frame - what
        # the launcher's screen bound of the box has to get right
        from vkvolume_amd import camera as _camera
        centre = tuple(float(x) for x in rng.uniform(-75.0, 75.0, size=3))
        view = _camera.orbit_camera(float(rng.uniform(0, 360)), float(rng.uniform(-80, 80)), radius, centre)
    opts = abi.RenderOptions(skipping_type=st, clip_distance=float(rng.choice([0.1, 1.0, 1.0, 20.0, 70.0])), early_ray_termination=bool(rng.integers(0, 2)))
    params = scene.params(view, proj, size, opts)
    ref = scene.render(params)
    label = "fuzz %d: shape %s block %d mode %d %s ert %d" % (seed, shape, block, st, grad_variant, opts.early_ray_termination)
    return scene, v, params, ref, label


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "36"))))  # a soak run sets more (profiles/r4_fuzz_soak.txt)
def test_render_fuzz(ctx, seed):
    """Random configurations (fuzz_case): counters bit-exact, colour and depth within the stated tolerance (observed 0), and all four
    {sampling layout} x {scheduler} variants identical (gpu_render)."""
    scene, v, params, ref, label = fuzz_case(ctx, seed)
    compare_render(gpu_render(ctx, v, params), ref, label)
    print("%s: %d samples, %d probes, %d pixels with colour" % (label, int(ref.counts[..., 0].sum()), int(ref.counts[..., 1].sum()), int((ref.color[..., 3] > 0).sum())))
