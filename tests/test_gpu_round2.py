"""GPU: transfer-function acceleration tables (separable greyscale claim checked on the device, generic RGBA fallback),
stream re-entrancy of one context, several volumes in one subpass, a reloaded Volume object."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import vkv_oracle as O
from tests import helpers as T
from tests.test_gpu_parity import compare_render, gpu_render, make_gpu_volume
from vkvolume_amd import abi, lib, multigpu, volume as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "vkvolume_amd", "csrc", "vkv_offscreen")
FLAG_WORD, AI_WORD, AG_WORD = 2048, 2052, 2308


def tables_of(ctx, tex, tf):
    d_tex = torch.from_numpy(np.ascontiguousarray(tex)).cuda()
    d_tab = torch.full((abi.TF_BITS_WORDS,), -1, dtype=torch.int32, device="cuda")
    ctx.transfer_function_tables(d_tex.data_ptr(), tf, d_tab.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_tab.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("opts", [T.APP_TF, dict(intensity_min=0.4, intensity_max=0.8, gradient_min=0.0, gradient_max=0.0),
                                  dict(intensity_min=0.2, intensity_max=0.8, gradient_min=0.06, gradient_max=0.12), dict()])
def test_tf_tables_separable_claim_is_checked_on_the_device(ctx, opts):
    opt = abi.VolumeOptions(**opts)
    tf, tex = lib.transfer_function_uniform(opt), O.transfer_function_texture(opt)
    tab = tables_of(ctx, tex, tf)
    bits = np.unpackbits(tab[:2048].view(np.uint8), bitorder="little").reshape(256, 256)
    assert np.array_equal(bits, (tex[..., 3] > 0).astype(np.uint8)), "alpha > 0 bit table"
    assert tab[FLAG_WORD] & 1 == 1, "the reference's own texture is the separable product"
    ai, ag = tab[AI_WORD:AI_WORD + 256].view(np.float32), tab[AG_WORD:AG_WORD + 256].view(np.float32)
    prod = np.minimum((ai[None, :] * ag[:, None] * np.float32(255.0)).astype(np.uint32), 255)
    assert np.array_equal(prod, tex[..., 3]) and np.array_equal(tex[..., 0], tex[..., 3])
    # one texel off by one in one channel: the claim must be withdrawn, the bit table must still be right
    for channel in (0, 3):
        bad = tex.copy()
        bad[200, 77, channel] ^= 1
        t2 = tables_of(ctx, bad, tf)
        assert t2[FLAG_WORD] & 1 == 0
        assert np.array_equal(np.unpackbits(t2[:2048].view(np.uint8), bitorder="little").reshape(256, 256), (bad[..., 3] > 0).astype(np.uint8))
    # no uniform: no claim
    assert tables_of(ctx, tex, None)[FLAG_WORD] & 1 == 0


@pytest.mark.parametrize("kind", ["rgba_random", "one_texel_off", "uniform_of_another_tf"])
@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_DISTANCE])
def test_render_with_textures_that_are_not_the_separable_product(ctx, kind, skipping_type):
    """Any RGBA8 texture must render like the oracle: the integrator may only take the two-table short cut when the device check passed."""
    vol = O.synth_volume((72, 64, 56), 1, 5)
    scene = T.OracleScene(vol, abi.VolumeOptions(**T.APP_TF), 4)
    rng = np.random.default_rng(3)
    tex = scene.tex.copy()
    if kind == "rgba_random":
        tex = rng.integers(0, 256, size=(256, 256, 4), dtype=np.uint8)
        tex[:, :30, 3] = 0  # keep low intensities empty so empty-space skipping has something to skip
    elif kind == "one_texel_off":
        g, i = np.argwhere(tex[..., 3] > 0)[100]
        tex[g, i, 1] ^= 0x40
    else:
        tex = O.transfer_function_texture(abi.VolumeOptions(intensity_min=0.3, intensity_max=0.9, gradient_min=0.0, gradient_max=0.5))
    scene.tex = tex
    scene._maps = {}
    v, tf = make_gpu_volume(ctx, scene)
    v.transfer_function.copy_(torch.from_numpy(tex))
    ctx.transfer_function_tables(v.transfer_function.data_ptr(), tf, v.transfer_function_bits.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(v.transfer_function_bits[FLAG_WORD].item()) & 1 == 0
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (128, 80)
    for az in (20.0, 200.0):
        view, proj = T.orbit(az, image_size=size)
        params = scene.params(view, proj, size, abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0))
        ref = scene.render(params)
        assert ref.counts[..., 0].sum() > 0
        compare_render(gpu_render(ctx, v, params), ref, "%s az %g" % (kind, az))


def test_one_context_on_three_streams(ctx):
    """Two map updates and one voxel count run concurrently on three streams of ONE context (each call stages a transfer-function
    bit table in context scratch: per stream), plus persistent-scheduler renders on two streams (per-stream tile queues)."""
    opts = [abi.VolumeOptions(**T.APP_TF), abi.VolumeOptions(intensity_min=0.35, intensity_max=0.8, gradient_min=0.02, gradient_max=0.3)]
    scenes = [T.OracleScene(O.synth_volume((96, 80, 72), 1, 40 + k), opts[k], 4) for k in range(2)]
    vols = [make_gpu_volume(ctx, s) for s in scenes]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    expect_maps = [s.maps(abi.SKIP_ANISOTROPIC_DISTANCE) for s in scenes]
    expect_count = O.occupied_voxel_count(scenes[0].vol, scenes[0].grad, scenes[0].tf)
    for rep in range(6):
        for (v, tf) in vols:
            v.set_number_of_distance_maps(8)
            for m in v.distance_maps:
                m.fill_(77)
        count.fill_(-1)
        torch.cuda.synchronize()
        for k, (v, tf) in enumerate(vols):
            ctx.compute_distance_map(v.volume.data_ptr(), v.gradient.data_ptr(), v.transfer_function.data_ptr(), tf, v.extent,
                                     [m.data_ptr() for m in v.distance_maps], v.distance_map_swap.data_ptr(), v.map_extent,
                                     abi.SKIP_ANISOTROPIC_DISTANCE, streams[k].cuda_stream)
        v0, tf0 = vols[0]
        ctx.occupied_voxel_count(v0.volume.data_ptr(), v0.gradient.data_ptr(), tf0, v0.extent, count.data_ptr(), streams[2].cuda_stream)
        torch.cuda.synchronize()
        for k, (v, tf) in enumerate(vols):
            got = np.stack([m.cpu().numpy() for m in v.distance_maps])
            assert np.array_equal(got, expect_maps[k]), "concurrent map update %d, repetition %d" % (k, rep)
        assert int(count.item()) == expect_count
    # two persistent-scheduler renders in flight on two streams
    size = (160, 96)
    ctx.set_tuning(scheduler=1)
    try:
        refs, outs, params = [], [], []
        for k, (v, tf) in enumerate(vols):
            view, proj = T.orbit(30.0 + 100 * k, image_size=size)
            p = scenes[k].params(view, proj, size, abi.RenderOptions(skipping_type=abi.SKIP_ANISOTROPIC_DISTANCE, clip_distance=1.0))
            refs.append(scenes[k].render(p))
            sp = V.VolumeRenderSubpass(ctx, v, p.options, size)
            params.append(sp.bind(p))
            outs.append(torch.zeros((size[1], size[0], 3), dtype=torch.int32, device="cuda"))
        torch.cuda.synchronize()
        for rep in range(8):
            for k in range(2):
                params[k].d_out_counts = outs[k].data_ptr()
                ctx.render(params[k], streams[k].cuda_stream)
        torch.cuda.synchronize()
        for k in range(2):
            assert np.array_equal(outs[k].cpu().numpy().astype(np.uint32), refs[k].counts)
    finally:
        ctx.set_tuning(scheduler=0)


def run_offscreen(tmp_path, *flags):
    assert os.path.exists(EXE), "vkv_offscreen not built (run __graft_entry__.build())"
    out = subprocess.run([EXE, *flags], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()
    return out.stdout.decode()


@pytest.mark.parametrize("skipmode", [0, 2])
def test_two_volumes_in_one_subpass(tmp_path, skipmode):
    """VolumeRenderSubpass::draw loops over its volumes (src/volume_render_subpass.cpp:219): the second one is blended onto the
    first one's result with the subpass's blend state.  C++ host classes through vkv_offscreen, expected frame from the oracle."""
    w, h = 160, 96
    s1, s2 = ((72, 60, 48), 1, 11), ((48, 56, 40), 1, 23)
    run_offscreen(tmp_path, "--synthetic=%dx%dx%d:%d:%d" % (*s1[0], s1[1], s1[2]), "--second-synthetic=%dx%dx%d:%d:%d" % (*s2[0], s2[1], s2[2]),
                  "--second-offset=25,-10,30", "--width=%d" % w, "--height=%d" % h, "--skipmode=%d" % skipmode, "--azimuth=40", "--elevation=15",
                  "--dump-counts=counts.raw", "--dump-rgba8=rgba8.raw", "--dump-params=p1.raw", "--dump-params2=p2.raw", "--reload")
    p1 = abi.RenderParams.from_buffer_copy(open(tmp_path / "p1.raw", "rb").read())
    p2 = abi.RenderParams.from_buffer_copy(open(tmp_path / "p2.raw", "rb").read())
    assert p2.blend_over_target == 1 and p1.blend_over_target == 0
    opt = abi.VolumeOptions(**T.APP_TF)
    tf, tex = O.transfer_function_uniform(opt), O.transfer_function_texture(opt)
    frames = []
    target = None
    for (shape, kind, seed), p in ((s1, p1), (s2, p2)):
        vol = O.synth_volume(shape, kind, seed)
        grad = O.gradient_map(vol, tf)
        maps = None if skipmode == 0 else O.compute_distance_map(vol, grad, tex, tf, 4, skipmode)
        r = O.render(p, vol, grad, tex, maps, want_rgba8=True, target_rgba8=target)
        target = r.rgba8
        frames.append(r)
    counts = np.fromfile(tmp_path / "counts.raw", np.uint32).reshape(h, w, 3)
    rgba8 = np.fromfile(tmp_path / "rgba8.raw", np.uint8).reshape(h, w, 4)
    both = (frames[0].counts[..., 0] > 0) & (frames[1].counts[..., 0] > 0)
    assert both.sum() > 100, "the two volumes must overlap on screen for the blend to be exercised"
    assert np.array_equal(rgba8, frames[1].rgba8), "two-volume frame differs from the oracle's"
    assert np.array_equal(counts, frames[1].counts)


@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
def test_render_batch_equals_single_launches(ctx, skipping_type):
    """vkv_render_batch: n frames (different cameras, one of them another volume) in one launch == n vkv_render calls, bit for bit."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scenes = [T.OracleScene(O.synth_volume((80, 72, 64), 1, 70 + k), opt, 4) for k in range(2)]
    vols = [make_gpu_volume(ctx, s) for s in scenes]
    for v, tf in vols:
        V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (144, 80)
    ro = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0)
    frames = []
    for k, az in enumerate((0.0, 50.0, 111.0, 200.0, 290.0)):
        scene, (v, tf) = scenes[k % 2], vols[k % 2]
        view, proj = T.orbit(az, image_size=size)
        p = V.VolumeRenderSubpass(ctx, v, ro, size).bind(scene.params(view, proj, size, ro))
        frames.append((scene, p))
    st = torch.cuda.current_stream().cuda_stream

    def outputs():
        return [dict(color=torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda"),
                     counts=torch.full((size[1], size[0], 3), 9, dtype=torch.int32, device="cuda"),
                     depth=torch.full((size[1], size[0]), -1.0, dtype=torch.float32, device="cuda"),
                     rgba8=torch.full((size[1], size[0], 4), 7, dtype=torch.uint8, device="cuda")) for _ in frames]

    def point(p, o):
        p.d_out_color, p.d_out_counts, p.d_out_depth, p.d_out_rgba8 = (o[k].data_ptr() for k in ("color", "counts", "depth", "rgba8"))

    single, batch = outputs(), outputs()
    for (scene, p), o in zip(frames, single):
        point(p, o)
        ctx.render(p, st)
    plist = []
    for (scene, p), o in zip(frames, batch):
        q = abi.RenderParams.from_buffer_copy(p)
        point(q, o)
        plist.append(q)
    ctx.render_batch(plist, st)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(single, batch)):
        for k in a:
            assert torch.equal(a[k], b[k]), "frame %d: %s differs between the batch and the single launch" % (i, k)
    ref = frames[3][0].render(frames[3][1])
    assert np.array_equal(batch[3]["counts"].cpu().numpy().astype(np.uint32), ref.counts)
    # a frame that needs another kernel variant is refused
    bad = abi.RenderParams.from_buffer_copy(plist[1])
    bad.options.early_ray_termination = 0
    with pytest.raises(lib.VkvError):
        ctx.render_batch([plist[0], bad], st)


def test_native_rccl_gather_and_assemble(ctx):
    """vkv_assemble_frame: ncclGather on the caller's communicator + de-interleave, through the C ABI (no torch.distributed).
    One GPU here, so the communicator has one rank (created with the RCCL the process has loaded); the compact tile layout,
    the gather and the scatter are the N > 1 code path."""
    import ctypes as C
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        scene = T.OracleScene(O.synth_volume((64, 56, 48), 1, 9), abi.VolumeOptions(**T.APP_TF), 4)
        v, tf = make_gpu_volume(ctx, scene)
        V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
        size, tile = (150, 70), 16  # not a multiple of the tile
        ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        view, proj = T.orbit(25.0, image_size=size)
        sp = V.VolumeRenderSubpass(ctx, v, ro, size)
        sched = abi.full_frame_tiles(size[0], size[1], tile, tile, 0, 1, compact=True)
        p = sp.make_params(view, proj, sched)
        n = sched.tile_count * tile * tile
        mine = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
        gathered = torch.full((1, n, 4), 9, dtype=torch.uint8, device="cuda")
        image = torch.full((size[1], size[0], 4), 5, dtype=torch.uint8, device="cuda")
        direct = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        sp.draw(p, rgba8=mine)
        ctx.assemble_frame(mine.data_ptr(), gathered.data_ptr(), image.data_ptr(), size, (tile, tile), 1, 0, sched.tile_count, 4, 0, comm.value, st)
        sp.draw(sp.make_params(view, proj), rgba8=direct)
        torch.cuda.synchronize()
        assert int(direct.sum().item()) > 0 and torch.equal(image, direct)
        assert torch.equal(gathered[0], mine)
        with pytest.raises(lib.VkvError):
            ctx.assemble_frame(mine.data_ptr(), gathered.data_ptr(), image.data_ptr(), size, (tile, tile), 1, 0, sched.tile_count, 4, 3, comm.value, st)
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.gpu
@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
def test_render_batch_pull_kernel_equals_single_launches(ctx, skipping_type):
    """VkvTuning.batch_mode = 1 (resident workgroups, waves pull 8x8 units from per-XCD ticket counters): same frames, bit for bit."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((96, 80, 72), 1, 91), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (208, 112)
    ro = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0)
    st = torch.cuda.current_stream().cuda_stream
    params = []
    for az in (0.0, 40.0, 95.0, 170.0, 230.0, 300.0, 345.0):
        view, proj = T.orbit(az, image_size=size)
        params.append(V.VolumeRenderSubpass(ctx, v, ro, size).bind(scene.params(view, proj, size, ro)))

    def outputs():
        return [dict(color=torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda"),
                     counts=torch.full((size[1], size[0], 3), 9, dtype=torch.int32, device="cuda"),
                     depth=torch.full((size[1], size[0]), -1.0, dtype=torch.float32, device="cuda"),
                     rgba8=torch.full((size[1], size[0], 4), 7, dtype=torch.uint8, device="cuda")) for _ in params]

    def point(p, o):
        p.d_out_color, p.d_out_counts, p.d_out_depth, p.d_out_rgba8 = (o[k].data_ptr() for k in ("color", "counts", "depth", "rgba8"))

    single, pulled = outputs(), outputs()
    for p, o in zip(params, single):
        point(p, o)
        ctx.render(p, st)
    plist = []
    for p, o in zip(params, pulled):
        q = abi.RenderParams.from_buffer_copy(p)
        point(q, o)
        plist.append(q)
    ctx.set_tuning(batch_mode=1)
    try:
        for _ in range(2):  # twice: the ticket counters are re-armed by every launch
            ctx.render_batch(plist, st)
        torch.cuda.synchronize()
    finally:
        ctx.set_tuning(batch_mode=0)
    for i, (a, b) in enumerate(zip(single, pulled)):
        for k in a:
            assert torch.equal(a[k], b[k]), "frame %d: %s differs between the pull kernel and the single launch" % (i, k)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["inside", "off_centre", "zoomed", "away", "grazing_corner", "compact_rank"])
def test_render_parity_screen_bound_cases(ctx, case):
    """The launcher's screen bound of the volume's box (pixels outside skip the ray set-up) must never cut a pixel the oracle shades:
    camera inside the box (bound disabled), box partly off screen, box larger than the screen, box behind the camera, a box corner
    beside the camera plane, and a compact multi-GPU tile schedule."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((88, 72, 64), 1, 0xB0B0), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    skipping_type = abi.SKIP_DISTANCE
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (176, 96)
    from vkvolume_amd import camera
    fov, tiles = 60.0, None
    if case == "inside":
        view = camera.look_at((12.0, 5.0, 8.0), (-30.0, 0.0, -20.0))
    elif case == "off_centre":
        view = camera.look_at((150.0, 40.0, 20.0), (70.0, 10.0, -60.0))
    elif case == "zoomed":
        view, fov = camera.look_at((130.0, 30.0, 40.0), (0.0, 0.0, 0.0)), 14.0
    elif case == "away":
        view = camera.look_at((150.0, 40.0, 20.0), (400.0, 90.0, 60.0))
    elif case == "grazing_corner":
        view = camera.look_at((62.0, 10.0, 58.0), (62.0, 10.0, -100.0))  # looks along -z past the box: corners beside the camera plane
    else:
        view = camera.look_at((140.0, 50.0, -60.0), (20.0, 0.0, 10.0))
        tiles = abi.full_frame_tiles(size[0], size[1], 16, 16, 1, 3, compact=True)
    proj = camera.perspective_vulkan(fov, size[0] / size[1])
    ro = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0)
    p = scene.params(view, proj, size, ro, tiles=tiles)
    ref = scene.render(p)
    got = gpu_render(ctx, v, p)
    compare_render(got, ref, "screen bound case %s" % case)
    if case == "away":
        assert int(ref.counts.sum()) == 0
    elif case != "grazing_corner":
        assert int(ref.counts.sum()) > 0


@pytest.mark.gpu
def test_scatter_tiles_reads_one_frame_of_a_gathered_batch(ctx):
    """The exchange of a whole vkv_render_batch launch (multigpu.BatchTileGather): the owner receives [rank][frame][tiles] and
    de-interleaves frame f with vkv_scatter_tiles on the block's f-th slice, the rank stride being frames x tiles_per_rank."""
    world, frames, size, tile = 3, 4, (208, 112), 16
    g = multigpu.BatchTileGather(None, 1, world, size, tile, 4, device="cuda", frames=frames, n_sets=1, any_root=True)
    rng = np.random.default_rng(5)
    flat = torch.from_numpy(rng.integers(0, 256, size=tuple(g.flat[0].shape), dtype=np.uint8)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for f in range(frames):
        img = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
        src, stride = g.frame_source(flat, f)
        ctx.scatter_tiles(src, img.data_ptr(), size, (tile, tile), world, stride, 4, st)
        want = multigpu.deinterleave_reference(flat[:, f].cpu().numpy(), size, tile, world)
        assert np.array_equal(img.cpu().numpy(), want), "frame %d of the batch" % f


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["address_tables=0", "address_tables=1", "screen_cull=0", "tile_order_linear=1", "full_table_lds_limit=0"])
def test_render_parity_under_each_kernel_selection_switch(switch):
    """The launcher picks one of three instantiations of the integrator (footprint address in registers / two-level LDS tables / one
    entry per voxel index), a tile start order and the screen bound by itself; small test volumes always get the same choice.  The
    render parity tests are re-run in a child process whose contexts start from the matching environment default (vkv_create reads
    VKV_RAYMARCH_* once per context), and the same switches are flipped through vkv_set_tuning in
    test_tuning_block_switches_render_the_same_bits."""
    env_of = {"address_tables=0": ("VKV_RAYMARCH_LUT", "0"), "address_tables=1": ("VKV_RAYMARCH_LUT", "2"), "screen_cull=0": ("VKV_RAYMARCH_CULL", "0"),
              "tile_order_linear=1": ("VKV_RAYMARCH_TILE_ORDER", "linear"), "full_table_lds_limit=0": ("VKV_RAYMARCH_FULL_LIMIT", "1")}
    name, value = env_of[switch]
    env = dict(os.environ, **{name: value})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-k", "render", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "%s:\n%s\n%s" % (switch, r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_tuning_block_switches_render_the_same_bits():
    """vkv_get_tuning / vkv_set_tuning: the environment is read once by vkv_create; afterwards only the block decides.  Every switch of
    the launchers renders the same frame (RGBA8 + counters) through vkv_render and vkv_render_batch."""
    c = lib.Context(0)
    try:
        t0 = c.get_tuning()
        assert t0.struct_size == C.sizeof(abi.Tuning) and t0.address_tables == 2 and t0.feedback == 1 and t0.feedback_period == 8 and t0.screen_cull == 1
        assert t0.arena_bytes >= (1 << 20)
        bad = abi.Tuning.from_buffer_copy(t0)
        bad.struct_size = 8
        assert c._lib.vkv_set_tuning(c.handle, C.byref(bad)) == abi.VKV_E_INVALID_ARGUMENT
        bad = abi.Tuning.from_buffer_copy(t0)
        bad.address_tables = 7
        assert c._lib.vkv_set_tuning(c.handle, C.byref(bad)) == abi.VKV_E_INVALID_ARGUMENT
        os.environ["VKV_RAYMARCH_LUT"] = "0"  # too late for this context: it must not change anything
        try:
            assert c.get_tuning().address_tables == 2
        finally:
            del os.environ["VKV_RAYMARCH_LUT"]
        opt = abi.VolumeOptions(**T.APP_TF)
        scene = T.OracleScene(O.synth_volume((72, 64, 56), 1, 4242), opt, 4)
        v, tf = make_gpu_volume(c, scene)
        V.ComputeDistanceMap(c).compute(v, tf, abi.SKIP_DISTANCE)
        size = (208, 112)
        ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        sp = V.VolumeRenderSubpass(c, v, ro, size)
        st = torch.cuda.current_stream().cuda_stream
        views = [sp.bind(scene.params(*T.orbit(az, image_size=size), size, ro)) for az in (15.0, 140.0, 260.0)]

        def frames():
            out = []
            for p in views:
                rgba, cnt = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda"), torch.zeros((size[1], size[0], 3), dtype=torch.int32, device="cuda")
                q = abi.RenderParams.from_buffer_copy(p)
                q.d_out_rgba8, q.d_out_counts, q.d_out_color, q.d_out_depth = rgba.data_ptr(), cnt.data_ptr(), None, None
                c.render(q, st)
                out.append((rgba, cnt))
            plist, outs = [], []
            for p in views:
                rgba, cnt = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda"), torch.zeros((size[1], size[0], 3), dtype=torch.int32, device="cuda")
                q = abi.RenderParams.from_buffer_copy(p)
                q.d_out_rgba8, q.d_out_counts, q.d_out_color, q.d_out_depth = rgba.data_ptr(), cnt.data_ptr(), None, None
                plist.append(q)
                outs.append((rgba, cnt))
            c.render_batch(plist, st)
            torch.cuda.synchronize()
            return out + outs

        ref = frames()
        assert int(ref[0][1].sum().item()) > 0
        for fields in (dict(address_tables=0), dict(address_tables=1), dict(full_table_lds_limit=0), dict(screen_cull=0), dict(tile_order_linear=1),
                       dict(batch_sequential=1), dict(batch_mode=1), dict(scheduler=1), dict(feedback=0), dict(tile_mix_heavy=0.3, tile_mix_spread=0.6)):
            c.set_tuning(**fields)
            got = frames()
            reset = {k: getattr(t0, k) for k in fields}
            c.set_tuning(**reset)
            for i, ((a, b), (x, y)) in enumerate(zip(ref, got)):
                assert torch.equal(a, x) and torch.equal(b, y), "%r changes frame %d" % (fields, i)
    finally:
        c.close()


@pytest.mark.gpu
def test_render_start_order_feedback_against_the_oracle(ctx):
    """vkv_render (one frame per launch) measures tile costs on the first frame into a target and every 8th one after it and starts the
    tiles of the following frames longest first: 19 frames into ONE set of output buffers, two views alternating in blocks of three (so
    orders derived from the other view are used too), every frame compared with the oracle (counters bit-exact)."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((64, 56, 48), 1, 321), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
    size = (208, 112)
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    assert ro.early_ray_termination
    sp = V.VolumeRenderSubpass(ctx, v, ro, size)
    views, refs = [], []
    for az in (25.0, 205.0):
        view, proj = T.orbit(az, image_size=size)
        params = scene.params(view, proj, size, ro)
        views.append(sp.bind(params))
        refs.append(scene.render(params))
    color = torch.empty((size[1], size[0], 4), dtype=torch.float32, device="cuda")
    counts = torch.empty((size[1], size[0], 3), dtype=torch.int32, device="cuda")
    depth = torch.empty((size[1], size[0]), dtype=torch.float32, device="cuda")
    ctx.register_target(color.data_ptr(), size, views[0].tiles)  # the feedback state of this target (vkv_render itself never allocates)
    for frame in range(19):
        k = (frame // 3) % 2
        color.fill_(-1.0), counts.fill_(0xFFFF), depth.fill_(-1.0)
        sp.draw(views[k], color, None, counts, depth)
        torch.cuda.synchronize()
        got = (color.cpu().numpy(), counts.cpu().numpy().astype(np.uint32), depth.cpu().numpy(), None)
        compare_render(got, refs[k], "frame %d (view %d)" % (frame, k))
    ctx.forget_target(color.data_ptr())


@pytest.mark.gpu
def test_start_order_feedback_targets_are_registered_and_forgotten(ctx):
    """Feedback state exists only for targets handed to vkv_register_target: 300 targets in turn, twice, half of them registered (and
    forgotten, re-registered with another schedule, registered twice): every frame equals the first one, registered or not, and a
    target registered for ANOTHER schedule is rendered without feedback instead of with a stale order."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((48, 40, 36), 1, 77), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
    size = (128, 128)  # 64 tiles: the smallest schedule that takes part in the feedback
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    sp = V.VolumeRenderSubpass(ctx, v, ro, size)
    view, proj = T.orbit(40.0, image_size=size)
    p = sp.bind(scene.params(view, proj, size, ro))
    targets = [torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda") for _ in range(300)]
    other = abi.full_frame_tiles(size[0], size[1], 32, 32)
    for j, t in enumerate(targets):
        if j % 2 == 0:
            ctx.register_target(t.data_ptr(), size, p.tiles)
        if j % 10 == 0:
            ctx.register_target(t.data_ptr(), size, p.tiles)  # again: replaces the state
        if j % 14 == 0:
            ctx.register_target(t.data_ptr(), size, other)  # registered for another schedule: this render gets no feedback
    for rnd in range(3):
        for t in targets:
            t.fill_(7)
            sp.draw(p, None, t)
        torch.cuda.synchronize()
        assert int(targets[0].sum().item()) > 0
        for j, t in enumerate(targets[1:]):
            assert torch.equal(t, targets[0]), "round %d, target %d" % (rnd, j + 1)
        if rnd == 0:
            for t in targets[::4]:
                ctx.forget_target(t.data_ptr())
    for t in targets:
        ctx.forget_target(t.data_ptr())  # unknown targets are fine
    assert ctx._lib.vkv_register_target(ctx.handle, None, 128, 128, C.byref(p.tiles)) == abi.VKV_E_INVALID_ARGUMENT


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(208, 112), (2608, 1040)])
def test_render_batch_start_order_feedback_keeps_the_frames(ctx, size):
    """vkv_render_batch re-orders the tiles of a frame by the costs the previous frame into the same target measured (a counting sort
    behind the render; more than 10 240 tiles take its two-pass path).  Four launches into the same targets, the views of the targets
    swapped in between: every frame equals its single-launch render."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((64, 56, 48), 1, 123), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    st = torch.cuda.current_stream().cuda_stream
    sp = V.VolumeRenderSubpass(ctx, v, ro, size)
    params = []
    for az in (10.0, 130.0, 250.0):
        view, proj = T.orbit(az, image_size=size)
        params.append(sp.bind(scene.params(view, proj, size, ro)))
    ref = []
    for p in params:
        out = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
        p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = out.data_ptr(), None, None, None
        ctx.render(p, st)
        ref.append(out)
    targets = [torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda") for _ in params]
    for t in targets:
        ctx.register_target(t.data_ptr(), size, params[0].tiles)
    for launch in range(4):
        shift = launch // 2  # launches 2 and 3 put other views into the same targets: the remembered costs belong to another view
        plist = []
        for j in range(len(params)):
            q = abi.RenderParams.from_buffer_copy(params[(j + shift) % len(params)])
            q.d_out_rgba8 = targets[j].data_ptr()
            plist.append(q)
        for t in targets:
            t.fill_(9)
        ctx.render_batch(plist, st)
        torch.cuda.synchronize()
        for j in range(len(params)):
            assert torch.equal(targets[j], ref[(j + shift) % len(params)]), "launch %d, target %d" % (launch, j)
    for t in targets:
        ctx.forget_target(t.data_ptr())
