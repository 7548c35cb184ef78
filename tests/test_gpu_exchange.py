"""GPU: the exchange step of the multi-GPU path through the C ABI - ncclGather + de-interleave per frame (vkv_assemble_frame) and per launch (vkv_assemble_frames), vkv_scatter_tiles."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import vkv_oracle as O
from tests import helpers as T
from tests.test_gpu_fullsize_oracle import build, orbit
from tests.test_gpu_parity import compare_render, gpu_render, make_gpu_volume
from vkvolume_amd import abi, lib, multigpu, volume as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "vkvolume_amd", "csrc", "vkv_offscreen")
FLAG_WORD, AI_WORD, AG_WORD = 2048, 2052, 2308


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
