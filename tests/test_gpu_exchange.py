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


@pytest.mark.parametrize("use_rect", [False, True])
def test_native_rccl_gather_and_assemble(ctx, use_rect):
    """vkv_assemble_frame: ncclGather on the caller's communicator + de-interleave, through the C ABI (no torch.distributed).
    One GPU here, so the communicator has one rank (created with the RCCL the process has loaded); the compact tile layout,
    the gather and the scatter are the N > 1 code path.  use_rect: only the tiles of the frame's screen rectangle
    (vkv_screen_tile_rect) are scheduled and gathered, the de-interleave clears the rest of the image."""
    rccl, comm = one_rank_communicator()
    try:
        scene = T.OracleScene(O.synth_volume((64, 56, 48), 1, 9), abi.VolumeOptions(**T.APP_TF), 4)
        v, tf = make_gpu_volume(ctx, scene)
        V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
        size, tile = (150, 70), 16  # not a multiple of the tile
        ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        view, proj = T.orbit(25.0, image_size=size)
        sp = V.VolumeRenderSubpass(ctx, v, ro, size)
        p0 = sp.make_params(view, proj)
        rect = lib.screen_tile_rect(p0.ray_cast, p0.ray_gen, size, (tile, tile)) if use_rect else None
        sched = abi.full_frame_tiles(size[0], size[1], tile, tile, 0, 1, compact=True, rect=rect)
        if use_rect:
            assert sched.tile_count == rect.tiles < 10 * 5
        p = sp.make_params(view, proj, sched)
        n = sched.tile_count * tile * tile
        mine = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
        gathered = torch.full((1, n, 4), 9, dtype=torch.uint8, device="cuda")
        image = torch.full((size[1], size[0], 4), 5, dtype=torch.uint8, device="cuda")
        direct = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        sp.draw(p, rgba8=mine)
        ctx.assemble_frame(mine.data_ptr(), gathered.data_ptr(), image.data_ptr(), size, (tile, tile), 1, 0, 4, 0, comm.value, st, rect=rect)
        sp.draw(p0, rgba8=direct)
        torch.cuda.synchronize()
        assert int(direct.sum().item()) > 0 and torch.equal(image, direct)
        assert torch.equal(gathered[0], mine)
        with pytest.raises(lib.VkvError):
            ctx.assemble_frame(mine.data_ptr(), gathered.data_ptr(), image.data_ptr(), size, (tile, tile), 1, 0, 4, 3, comm.value, st, rect=rect)
        with pytest.raises(lib.VkvError):  # a rectangle that runs past the image
            ctx.assemble_frame(mine.data_ptr(), gathered.data_ptr(), image.data_ptr(), size, (tile, tile), 1, 0, 4, 0, comm.value, st, rect=abi.TileRect(8, 0, 3, 2))
    finally:
        rccl.ncclCommDestroy(comm)


def random_rects(rng, size, tile, frames):
    """tile rectangles inside the image, some of them the whole image"""
    tx, ty = -(-size[0] // tile), -(-size[1] // tile)
    out = []
    for f in range(frames):
        if rng.random() < 0.25:
            out.append(abi.TileRect(0, 0, tx, ty))
            continue
        w, h = int(rng.integers(1, tx + 1)), int(rng.integers(1, ty + 1))
        out.append(abi.TileRect(int(rng.integers(0, tx - w + 1)), int(rng.integers(0, ty - h + 1)), w, h))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("use_rect", [False, True])
def test_scatter_tiles_reads_one_frame_of_a_gathered_batch(ctx, use_rect):
    """The exchange of a whole vkv_render_batch launch (multigpu.BatchTileGather, one owner): the owner receives [rank][block of the launch] and
    de-interleaves frame f with vkv_scatter_tiles on the block's f-th slice, the rank stride being the block's tiles; every frame with
    its own tile rectangle, pixels outside it cleared."""
    world, frames, size, tile = 3, 4, (208, 112), 16
    rng = np.random.default_rng(5)
    rects = random_rects(rng, size, tile, frames) if use_rect else [abi.whole_image_rect(size[0], size[1], tile, tile)] * frames
    tpr, off, total = multigpu.launch_layout(rects, world)
    flat = torch.from_numpy(rng.integers(0, 256, size=(world, total * tile * tile, 4), dtype=np.uint8)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for f in range(frames):
        img = torch.full((size[1], size[0], 4), 7, dtype=torch.uint8, device="cuda")
        ctx.scatter_tiles(flat.data_ptr() + off[f] * tile * tile * 4, img.data_ptr(), size, (tile, tile), world, total, 4, st, rect=rects[f])
        a, b = off[f] * tile * tile, (off[f] + tpr[f]) * tile * tile
        want = multigpu.deinterleave_reference(flat[:, a:b].cpu().numpy(), size, tile, world, rects[f])
        assert np.array_equal(img.cpu().numpy(), want), "frame %d of the batch, rect %s" % (f, rects[f].as_tuple())


@pytest.mark.parametrize("mode", ["whole", "rect", "rect_grouped"])
def test_assemble_frames_one_collective_per_launch(ctx, mode):
    """vkv_assemble_frames: the frames of a vkv_render_batch launch travel as ONE ncclGather of [frame][tiles] per rank (or, with `roots`, as one
    group of gathers, frame f to roots[f]), and ONE kernel de-interleaves them into the launch's images.  (1) a one-rank communicator:
    three frames rendered compact by one launch - each through its own screen rectangle in the `rect` modes, back to back in the
    rank's block -, assembled natively, equal to direct renders, also with fewer frames than the buffers hold; (2) argument checks.
    (The three-rank layouts of the de-interleave: test_scatter_frames_kernel_three_ranks.)"""
    rccl, comm = one_rank_communicator()
    try:
        scene = T.OracleScene(O.synth_volume((64, 56, 48), 1, 9), abi.VolumeOptions(**T.APP_TF), 4)
        v, tf = make_gpu_volume(ctx, scene)
        V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
        size, tile = (208, 112), 16
        ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        sp = V.VolumeRenderSubpass(ctx, v, ro, size)
        st = torch.cuda.current_stream().cuda_stream
        frames = 3
        views = [T.orbit(az, image_size=size) for az in (10.0, 120.0, 250.0)]
        full = [sp.make_params(view, proj) for view, proj in views]
        rects = [lib.screen_tile_rect(p.ray_cast, p.ray_gen, size, (tile, tile)) for p in full] if mode != "whole" else None
        scheds = [abi.full_frame_tiles(size[0], size[1], tile, tile, 0, 1, compact=True, rect=rects[f] if rects else None) for f in range(frames)]
        tpr, off, total = multigpu.launch_layout(rects if rects else [abi.whole_image_rect(size[0], size[1], tile, tile)] * frames, 1)
        assert tpr == [s.tile_count for s in scheds]
        if rects:
            assert total < frames * 13 * 7
        roots = [0] * frames if mode == "rect_grouped" else None
        mine = torch.zeros((total * tile * tile, 4), dtype=torch.uint8, device="cuda")
        plist, direct = [], []
        for f, (view, proj) in enumerate(views):
            p = sp.make_params(view, proj, scheds[f])
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = mine.data_ptr() + off[f] * tile * tile * 4, None, None, None
            plist.append(p)
            d = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
            sp.draw(full[f], rgba8=d)
            direct.append(d)
        ctx.render_batch(plist, st)  # frames of different tile counts in one launch
        gathered = torch.full((total * tile * tile, 4), 9, dtype=torch.uint8, device="cuda")
        images = [torch.full((size[1], size[0], 4), 5, dtype=torch.uint8, device="cuda") for _ in range(frames)]
        ctx.assemble_frames(mine.data_ptr(), gathered.data_ptr(), [i.data_ptr() for i in images], frames, size, (tile, tile), 1, 0, 4, 0,
                            comm.value, st, rects=rects, roots=roots)
        torch.cuda.synchronize()
        assert torch.equal(gathered, mine)  # (one rank: both layouts of d_gathered are the rank's block)
        for f in range(frames):
            assert int(direct[f].sum().item()) > 0 and torch.equal(images[f], direct[f]), "frame %d" % f
        # a launch with fewer frames than the buffers hold: only the first two images are written
        for i in images:
            i.fill_(5)
        ctx.assemble_frames(mine.data_ptr(), gathered.data_ptr(), [i.data_ptr() for i in images[:2]], 2, size, (tile, tile), 1, 0, 4, 0,
                            comm.value, st, rects=rects[:2] if rects else None, roots=roots[:2] if roots else None)
        torch.cuda.synchronize()
        assert torch.equal(images[0], direct[0]) and torch.equal(images[1], direct[1]) and int((images[2] != 5).sum().item()) == 0
        # (2) argument checks
        L = ctx._lib
        arr = (C.c_void_p * 3)(*[i.data_ptr() for i in images])
        bad_rects = (abi.TileRect * 3)(abi.TileRect(0, 0, 13, 7), abi.TileRect(12, 0, 2, 2), abi.TileRect(0, 0, 1, 1))
        bad_roots = (C.c_int32 * 3)(0, 1, 0)
        args = lambda **kw: [kw.get("tiles", mine.data_ptr()), kw.get("gath", gathered.data_ptr()), kw.get("imgs", arr), kw.get("frames", 3), size[0], size[1],
                             tile, tile, kw.get("rects", None), kw.get("n_ranks", 1), kw.get("rank", 0), kw.get("bpp", 4), kw.get("root", 0), kw.get("roots", None),
                             comm.value, st]
        assert L.vkv_assemble_frames(ctx.handle, *args(frames=0)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(frames=abi.MAX_BATCH + 1)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(root=2)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(imgs=None)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(gath=None)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(bpp=3)) == abi.VKV_E_INVALID_ARGUMENT
        assert L.vkv_assemble_frames(ctx.handle, *args(rects=bad_rects)) == abi.VKV_E_INVALID_ARGUMENT  # the second rectangle runs past the image
        assert L.vkv_assemble_frames(ctx.handle, *args(roots=bad_roots)) == abi.VKV_E_INVALID_ARGUMENT  # a root outside the communicator
    finally:
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("use_rect", [False, True])
@pytest.mark.parametrize("grouped", [False, True])
def test_scatter_frames_kernel_three_ranks(ctx, use_rect, grouped):
    """k_scatter_tiles_frames (the kernel behind vkv_assemble_frames) on the blocks of three ranks, through the ABI: this process is
    rank 0 of a layout of three (the communicator at hand has one rank, so the collective delivers rank 0's block only; the blocks of
    ranks 1 and 2 are already in d_gathered, as if they had arrived) - every frame must equal the numpy statement of the de-interleave.
    Frame sizes that are no multiple of the tile, 1 .. 4 frames, every frame with its own rectangle (use_rect); grouped: the layout the
    per-frame gathers of `roots` leave, [frame][rank][tiles], instead of [rank][frame][tiles]."""
    rccl, comm = one_rank_communicator()
    try:
        world, size, tile = 3, (150, 70), 16
        tp = tile * tile
        rng = np.random.default_rng(3)
        st = torch.cuda.current_stream().cuda_stream
        for frames in (1, 2, 4):
            rects = random_rects(rng, size, tile, frames) if use_rect else [abi.whole_image_rect(size[0], size[1], tile, tile)] * frames
            tpr, off, total = multigpu.launch_layout(rects, world)
            blocks = torch.from_numpy(rng.integers(0, 256, size=(world, total * tp, 4), dtype=np.uint8)).cuda()  # every rank's [frame][tiles] block
            if grouped:  # frame f as [rank][tiles of f] at tile world * off[f]
                want_gathered = torch.cat([blocks[r, off[f] * tp:(off[f] + tpr[f]) * tp] for f in range(frames) for r in range(world)])
            else:
                want_gathered = blocks.reshape(-1, 4).clone()
            gathered = want_gathered.clone()
            for f in range(frames if grouped else 1):  # rank 0's part comes through ncclGather
                a = world * off[f] * tp if grouped else 0
                gathered[a:a + (tpr[f] if grouped else total) * tp].zero_()
            images = [torch.full((size[1], size[0], 4), 3, dtype=torch.uint8, device="cuda") for _ in range(frames)]
            ctx.assemble_frames(blocks[0].data_ptr(), gathered.data_ptr(), [i.data_ptr() for i in images], frames, size, (tile, tile), world, 0, 4, 0,
                                comm.value, st, rects=rects, roots=[0] * frames if grouped else None)
            torch.cuda.synchronize()
            assert torch.equal(gathered, want_gathered)
            for f in range(frames):
                want = multigpu.deinterleave_reference(blocks[:, off[f] * tp:(off[f] + tpr[f]) * tp].cpu().numpy(), size, tile, world, rects[f])
                assert np.array_equal(images[f].cpu().numpy(), want), "%d frames, frame %d, rect %s" % (frames, f, rects[f].as_tuple())
    finally:
        rccl.ncclCommDestroy(comm)
