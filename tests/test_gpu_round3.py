"""Round 3 GPU tests (through the C ABI): BASELINE.json configs[4] (C5) as eight virtual ranks on one GPU, the batched native exchange
vkv_assemble_frames, the set-up calls (vkv_prepare_render, vkv_register_target, vkv_release_stream) and the arena policy."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import vkv_oracle as O
from tests import helpers as T
from tests.test_gpu_fullsize_oracle import build, orbit
from tests.test_gpu_parity import make_gpu_volume
from vkvolume_amd import abi, lib, multigpu, volume as V

pytestmark = pytest.mark.gpu


def one_rank_communicator():
    """an RCCL communicator with one rank, created with the RCCL copy the process has loaded (the one vkv_gather_tiles resolves)"""
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    return rccl, comm


def test_c5_frame_as_eight_virtual_ranks(ctx):
    """BASELINE.json configs[4]: 2048^3 uint8, anisotropic Chebyshev maps, the 7680x4320 frame cut into 16x16 tiles dealt round-robin to
    8 ranks, RCCL gather.  One GPU plays the eight ranks in turn (rays are independent: what rank r renders does not depend on who else
    renders): rank r renders VkvTileSchedule{16, 16, tile_first = r, tile_stride = 8, compact = 1} of two views with one vkv_render_batch
    launch into its compact buffers; the eight buffers are laid out [rank][tiles] as ncclGather delivers them - and are also sent through
    ncclGather itself (a one-rank communicator, 531 MB) - and de-interleaved by vkv_scatter_tiles(n_ranks = 8).
      (a) the assembled RGBA8 frame == a direct full-frame render, byte for byte (both views);
      (b) every 32nd pixel in x and y == the CPU oracle: three counters + RGBA8, bit-exact;
      (c) the eight ranks' event totals (volume samples + distance probes) are within +-15 % of each other (the load-balance claim
          behind round-robin tiles)."""
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2 ** 30:
        pytest.skip("needs ~70 GiB of HBM")
    try:
        import psutil
        if psutil.virtual_memory().available < 40 * 2 ** 30:
            pytest.skip("needs ~20 GiB of host memory for the oracle's copy of the scene")
    except ImportError:
        pass
    world, tile, size = 8, 16, (7680, 4320)
    fw, fh = size
    v, tf = build(ctx, (2048, 2048, 2048), 0xC0FFEE04, abi.SKIP_ANISOTROPIC_DISTANCE)
    opts = abi.RenderOptions(skipping_type=abi.SKIP_ANISOTROPIC_DISTANCE, clip_distance=1.0, early_ray_termination=True)
    sp = V.VolumeRenderSubpass(ctx, v, opts, size)
    st = torch.cuda.current_stream().cuda_stream
    views = [orbit(v, az, size) for az in (45.0, 200.0)]
    total_tiles = (fw // tile) * (fh // tile)
    assert total_tiles == 129600 and total_tiles % world == 0
    per_rank = total_tiles // world
    n = per_rank * tile * tile
    gathered = [torch.zeros((world, n, 4), dtype=torch.uint8, device="cuda") for _ in views]        # [rank][tiles] per view
    counts_r = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    events = np.zeros((len(views), world), np.int64)
    for r in range(world):
        sched = abi.full_frame_tiles(fw, fh, tile, tile, r, world, compact=True)
        assert sched.tile_count == per_rank
        plist = []
        for k, (view, proj) in enumerate(views):
            p = sp.make_params(view, proj, sched)
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = gathered[k][r].data_ptr(), None, None, None
            plist.append(p)
        ctx.render_batch(plist, st)        # ONE launch per rank: its tiles of both views
        for k, (view, proj) in enumerate(views):        # the rank's counters (a second, single-frame launch of the same tiles)
            p = sp.make_params(view, proj, sched)
            check = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
            sp.draw(p, rgba8=check, counts=counts_r)
            torch.cuda.synchronize()
            assert torch.equal(check, gathered[k][r]), "rank %d, view %d: vkv_render and vkv_render_batch disagree" % (r, k)
            events[k, r] = int(counts_r[:, :2].to(torch.int64).sum().item())
    torch.cuda.synchronize()
    # (c) load balance of round-robin 16x16 tiles
    for k in range(len(views)):
        mean = events[k].mean()
        assert mean > 1e6
        assert float(np.abs(events[k] - mean).max()) <= 0.15 * mean, "view %d: per-rank events %r" % (k, events[k].tolist())
    rccl, comm = one_rank_communicator()
    try:
        for k, (view, proj) in enumerate(views):
            p_full = sp.make_params(view, proj)
            direct = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
            counts = torch.zeros((fh, fw, 3), dtype=torch.int32, device="cuda")
            sp.draw(p_full, rgba8=direct, counts=counts)
            # (a) de-interleave of the [rank][tiles] block
            image = torch.full((fh, fw, 4), 3, dtype=torch.uint8, device="cuda")
            ctx.scatter_tiles(gathered[k].data_ptr(), image.data_ptr(), size, (tile, tile), world, per_rank, 4, st)
            torch.cuda.synchronize()
            assert int(direct.to(torch.int64).sum().item()) > 0
            assert torch.equal(image, direct), "view %d: the frame assembled from 8 ranks' tiles differs from the direct render" % k
            # the same bytes through the native path at this size: ncclGather (RCCL, one rank) of the whole block, then the de-interleave
            recv = torch.zeros_like(gathered[k])
            image.fill_(5)
            ctx.gather_tiles(gathered[k].data_ptr(), recv.data_ptr(), gathered[k].numel(), 0, comm.value, st)
            ctx.scatter_tiles(recv.data_ptr(), image.data_ptr(), size, (tile, tile), world, per_rank, 4, st)
            torch.cuda.synchronize()
            assert torch.equal(image, direct), "view %d: ncclGather + vkv_scatter_tiles differs from the direct render" % k
            if k == 0:
                # (b) the oracle on every 32nd pixel
                vol, grad, tex = v.volume.cpu().numpy(), v.gradient.cpu().numpy(), v.transfer_function.cpu().numpy()
                maps = [m.cpu().numpy() for m in v.distance_maps]
            stride = 32
            ref = O.render(p_full, vol, grad, tex, maps, pixel_stride=stride, want_rgba8=True)
            sel = (slice(0, fh, stride), slice(0, fw, stride))
            assert ref.counts[sel][..., 0].sum() > 1000, "the sampled pixels must hit the volume"
            assert np.array_equal(counts.cpu().numpy().astype(np.uint32)[sel], ref.counts[sel]), "view %d: counters differ from the oracle" % k
            assert np.array_equal(image.cpu().numpy()[sel], ref.rgba8[sel]), "view %d: assembled RGBA8 differs from the oracle" % k
            del direct, counts, image, recv
    finally:
        rccl.ncclCommDestroy(comm)


def test_assemble_frames_one_collective_per_launch(ctx):
    """vkv_assemble_frames: the frames of a vkv_render_batch launch travel as ONE ncclGather of [frame][tiles] per rank, and ONE kernel
    de-interleaves [rank][frame][tiles] into the launch's images.  (1) a one-rank communicator: three frames rendered compact by one
    launch, assembled natively, equal to direct renders, also with fewer frames than the buffers hold; (2) argument checks.  (The
    three-rank layout of the de-interleave: test_scatter_frames_kernel_three_ranks.)"""
    rccl, comm = one_rank_communicator()
    try:
        scene = T.OracleScene(O.synth_volume((64, 56, 48), 1, 9), abi.VolumeOptions(**T.APP_TF), 4)
        v, tf = make_gpu_volume(ctx, scene)
        V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
        size, tile = (208, 112), 16
        ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        sp = V.VolumeRenderSubpass(ctx, v, ro, size)
        sched = abi.full_frame_tiles(size[0], size[1], tile, tile, 0, 1, compact=True)
        n = sched.tile_count * tile * tile
        st = torch.cuda.current_stream().cuda_stream
        frames = 3
        mine = torch.zeros((frames, n, 4), dtype=torch.uint8, device="cuda")
        plist, direct = [], []
        for f, az in enumerate((10.0, 120.0, 250.0)):
            view, proj = T.orbit(az, image_size=size)
            p = sp.make_params(view, proj, sched)
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = mine[f].data_ptr(), None, None, None
            plist.append(p)
            d = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
            sp.draw(sp.make_params(view, proj), rgba8=d)
            direct.append(d)
        ctx.render_batch(plist, st)
        gathered = torch.full((1, frames, n, 4), 9, dtype=torch.uint8, device="cuda")
        images = [torch.full((size[1], size[0], 4), 5, dtype=torch.uint8, device="cuda") for _ in range(frames)]
        ctx.assemble_frames(mine.data_ptr(), gathered.data_ptr(), [i.data_ptr() for i in images], frames, size, (tile, tile), 1, 0, sched.tile_count, 4, 0,
                            comm.value, st)
        torch.cuda.synchronize()
        assert torch.equal(gathered[0], mine)
        for f in range(frames):
            assert int(direct[f].sum().item()) > 0 and torch.equal(images[f], direct[f]), "frame %d" % f
        # a launch with fewer frames than the buffers hold: only the first two images are written
        for i in images:
            i.fill_(5)
        ctx.assemble_frames(mine.data_ptr(), gathered.data_ptr(), [i.data_ptr() for i in images[:2]], 2, size, (tile, tile), 1, 0, sched.tile_count, 4, 0,
                            comm.value, st)
        torch.cuda.synchronize()
        assert torch.equal(images[0], direct[0]) and torch.equal(images[1], direct[1]) and int((images[2] != 5).sum().item()) == 0
        # (2) argument checks
        L = ctx._lib
        arr = (C.c_void_p * 3)(*[i.data_ptr() for i in images])
        args = lambda **kw: [kw.get("tiles", mine.data_ptr()), kw.get("gath", gathered.data_ptr()), kw.get("imgs", arr), kw.get("frames", 3), size[0], size[1],
                             tile, tile, kw.get("n_ranks", 1), kw.get("rank", 0), kw.get("tpr", sched.tile_count), 4, kw.get("root", 0), comm.value, st]
        assert L.vkv_assemble_frames(ctx.handle, *args(frames=0)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(frames=abi.MAX_BATCH + 1)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(root=2)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(imgs=None)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(tpr=sched.tile_count - 1)) == abi.VKV_E_INVALID_ARGUMENT
    finally:
        rccl.ncclCommDestroy(comm)


def test_scatter_frames_kernel_three_ranks(ctx):
    """k_scatter_tiles_frames (the kernel behind vkv_assemble_frames) on a [3 ranks][frames][tiles] block, through the ABI: this process is
    rank 0 of a layout of three (the communicator at hand has one rank, so the collective delivers rank 0's block only; the blocks of
    ranks 1 and 2 are already in d_gathered, as if they had arrived) - every frame must equal the numpy statement of the de-interleave.
    Frame sizes that are no multiple of the tile, 1 .. 4 frames."""
    rccl, comm = one_rank_communicator()
    try:
        world, size, tile = 3, (150, 70), 16
        tiles_x, tiles_y = -(-size[0] // tile), -(-size[1] // tile)
        per_rank = -(-(tiles_x * tiles_y) // world)
        npx = per_rank * tile * tile
        rng = np.random.default_rng(3)
        st = torch.cuda.current_stream().cuda_stream
        for frames in (1, 2, 4):
            flat = torch.from_numpy(rng.integers(0, 256, size=(world, frames, npx, 4), dtype=np.uint8)).cuda()
            gathered = flat.clone()
            gathered[0].zero_()        # rank 0's block comes through ncclGather
            images = [torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda") for _ in range(frames)]
            ctx.assemble_frames(flat[0].data_ptr(), gathered.data_ptr(), [i.data_ptr() for i in images], frames, size, (tile, tile), world, 0, per_rank, 4, 0,
                                comm.value, st)
            torch.cuda.synchronize()
            assert torch.equal(gathered, flat)
            for f in range(frames):
                want = multigpu.deinterleave_reference(flat[:, f].cpu().numpy(), size, tile, world)
                assert np.array_equal(images[f].cpu().numpy(), want), "%d frames, frame %d" % (frames, f)
    finally:
        rccl.ncclCommDestroy(comm)


def test_prepare_render_then_launches_take_nothing_new(ctx):
    """vkv_prepare_render (set-up) creates the stream's scratch block, the address tables and the tile start order; the launches after it
    find everything in place: the arena's fill level (read back through a second prepare call's idempotence and the device's free
    memory) does not move, on a new stream or on the old one, and the frames equal those of an unprepared context."""
    scene = T.OracleScene(O.synth_volume((80, 72, 64), 1, 515), abi.VolumeOptions(**T.APP_TF), 4)
    size = (208, 112)
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    results = []
    for prepared in (False, True):
        c = lib.Context(0)
        try:
            v, tf = make_gpu_volume(c, scene)
            V.ComputeDistanceMap(c).compute(v, tf, abi.SKIP_DISTANCE)
            sp = V.VolumeRenderSubpass(c, v, ro, size)
            streams = [torch.cuda.Stream() for _ in range(3)]
            targets = [torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda") for _ in range(6)]
            plist = []
            for j, az in enumerate((0.0, 60.0, 120.0, 180.0, 240.0, 300.0)):
                p = sp.make_params(*T.orbit(az, image_size=size))
                p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = targets[j].data_ptr(), None, None, None
                plist.append(p)
            torch.cuda.synchronize()
            if prepared:
                for s in streams:
                    c.prepare_render(plist, s.cuda_stream)
                for t in targets:
                    c.register_target(t.data_ptr(), size, plist[0].tiles)
                torch.cuda.synchronize()
            free0, _ = torch.cuda.mem_get_info()
            for rep in range(3):
                for k, s in enumerate(streams):
                    c.render_batch(plist[2 * k:2 * k + 2], s.cuda_stream)
                    c.render(plist[2 * k], s.cuda_stream)
            torch.cuda.synchronize()
            free1, _ = torch.cuda.mem_get_info()
            if prepared:
                assert free1 == free0, "launches after vkv_prepare_render / vkv_register_target must not allocate device memory (%d bytes)" % (free0 - free1)
            results.append([t.clone() for t in targets])
            for s in streams:
                c.release_stream(s.cuda_stream)
            for t in targets:
                c.forget_target(t.data_ptr())
        finally:
            c.close()
    for a, b in zip(*results):
        assert int(a.sum().item()) > 0 and torch.equal(a, b)


def test_arena_exhaustion_degrades_to_table_free_launches(monkeypatch):
    """A context whose arena has no room for a table still renders the same bits: the launch runs without its start order instead of
    allocating, and a batch launch on a stream that cannot get a scratch block says so (VKV_ARENA_BYTES is read by vkv_create; the
    minimum is 1 MiB: four 128 KiB scratch blocks + 512 KiB of tables - round 4 gave the two their own regions)."""
    scene = T.OracleScene(O.synth_volume((72, 64, 56), 1, 99), abi.VolumeOptions(**T.APP_TF), 4)
    size = (6144, 6144)        # 147 456 tiles: a start order of 576 KiB, more than the small arena's table region
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    frames = []
    for arena in (None, "1048576"):
        if arena:
            monkeypatch.setenv("VKV_ARENA_BYTES", arena)
        c = lib.Context(0)
        monkeypatch.delenv("VKV_ARENA_BYTES", raising=False)
        try:
            assert c.get_tuning().arena_bytes == (int(arena) if arena else 8 << 20)
            v, tf = make_gpu_volume(c, scene)
            V.ComputeDistanceMap(c).compute(v, tf, abi.SKIP_DISTANCE)        # the default stream's scratch block
            count = torch.zeros(1, dtype=torch.int64, device="cuda")
            for s_ in [torch.cuda.Stream() for _ in range(3)]:        # three more: the four scratch blocks of the small arena are gone
                c.occupied_voxel_count(v.volume.data_ptr(), v.gradient.data_ptr(), tf, v.extent, count.data_ptr(), s_.cuda_stream)
            torch.cuda.synchronize()
            sp = V.VolumeRenderSubpass(c, v, ro, size)
            t = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
            p = sp.make_params(*T.orbit(40.0, image_size=size))
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = t.data_ptr(), None, None, None
            st = torch.cuda.Stream()
            c.render(p, st.cuda_stream)        # small arena: no room for the 576 KiB start order - plain tile order, same frame
            torch.cuda.synchronize()
            frames.append(t)
            if arena:
                # a batch launch needs a scratch block for its argument blocks: none left on a new stream - the call says so instead of allocating
                q = abi.RenderParams.from_buffer_copy(p)
                rc = c._lib.vkv_render_batch(c.handle, (abi.RenderParams * 2)(p, q), 2, torch.cuda.Stream().cuda_stream)
                assert rc == abi.VKV_E_UNSUPPORTED and "arena" in c.last_error()
                # ... while the set-up call may allocate: afterwards the same launch works
                st2 = torch.cuda.Stream()
                c.prepare_render([p, q], st2.cuda_stream)
                t.zero_()
                c.render_batch([p, q], st2.cuda_stream)
                torch.cuda.synchronize()
                assert torch.equal(t, frames[0])
        finally:
            c.close()
    assert int(frames[0].to(torch.int64).sum().item()) > 0 and torch.equal(frames[0], frames[1])


@pytest.mark.parametrize("skipping_type", [abi.SKIP_BLOCK, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
def test_launch_without_counters_renders_the_same_frame(ctx, skipping_type):
    """A launch without d_out_counts (what a renderer submits: the reference keeps the per-pixel sample counters only in its test modes,
    src/volume_render_subpass.h Test::NumTextureSamples) runs the integrator WITHOUT the three counters (kLeanNoCounts).  Its float
    colour, depth and RGBA8 must be the counted launch's bits - through vkv_render and through vkv_render_batch - and the oracle's."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((88, 72, 64), 1, 0x5EED0003), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (160, 96)
    ro = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0, early_ray_termination=True)
    st = torch.cuda.current_stream().cuda_stream
    plist = []
    for az in (10.0, 77.0, 200.0):
        view, proj = T.orbit(az, image_size=size)
        plist.append(V.VolumeRenderSubpass(ctx, v, ro, size).bind(scene.params(view, proj, size, ro)))

    def outputs(with_counts):
        return [dict(color=torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda"),
                     depth=torch.full((size[1], size[0]), -1.0, dtype=torch.float32, device="cuda"),
                     rgba8=torch.full((size[1], size[0], 4), 7, dtype=torch.uint8, device="cuda"),
                     counts=torch.full((size[1], size[0], 3), 9, dtype=torch.int32, device="cuda") if with_counts else None) for _ in plist]

    def point(p, o):
        q = abi.RenderParams.from_buffer_copy(p)
        q.d_out_color, q.d_out_depth, q.d_out_rgba8 = o["color"].data_ptr(), o["depth"].data_ptr(), o["rgba8"].data_ptr()
        q.d_out_counts = o["counts"].data_ptr() if o["counts"] is not None else None
        return q

    counted = outputs(True)
    for p, o in zip(plist, counted):
        ctx.render(point(p, o), st)
    torch.cuda.synchronize()
    assert int(counted[0]["counts"][..., 0].sum().item()) > 0 and int(counted[0]["counts"][..., 1].sum().item()) > 0
    try:
        for tables in (2, 1):        # one table entry per voxel index / the two-level tables: each has its own kernel without counters
            ctx.set_tuning(address_tables=tables)
            single, batch = outputs(False), outputs(False)
            for p, o in zip(plist, single):
                ctx.render(point(p, o), st)
            ctx.render_batch([point(p, o) for p, o in zip(plist, batch)], st)
            torch.cuda.synchronize()
            for i in range(len(plist)):
                for k in ("color", "depth", "rgba8"):
                    assert torch.equal(counted[i][k], single[i][k]), "tables %d frame %d: %s of the launch without counters differs" % (tables, i, k)
                    assert torch.equal(counted[i][k], batch[i][k]), "tables %d frame %d: %s of the batch launch without counters differs" % (tables, i, k)
    finally:
        ctx.set_tuning(address_tables=2)
    ref = scene.render(plist[1], want_rgba8=True)
    assert np.array_equal(single[1]["rgba8"].cpu().numpy(), ref.rgba8), "RGBA8 of the launch without counters differs from the oracle's"
    assert np.array_equal(counted[1]["counts"].cpu().numpy().astype(np.uint32), ref.counts)
