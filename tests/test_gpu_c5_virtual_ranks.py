"""GPU: BASELINE.json configs[4] (C5: 2048^3, anisotropic maps, 7680x4320) rendered by one GPU as eight ranks in turn, assembled, and compared with a direct render and the oracle."""
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


def test_c5_frame_as_eight_virtual_ranks(ctx):
    """BASELINE.json configs[4]: 2048^3 uint8, anisotropic Chebyshev maps, the 7680x4320 frame cut into 16x16 tiles dealt round-robin to
    8 ranks, RCCL gather.  One GPU plays the eight ranks in turn (rays are independent: what rank r renders does not depend on who else
    renders): rank r renders VkvTileSchedule{16, 16, tile_first = r, tile_stride = 8, compact = 1} of two views with one vkv_render_batch
    launch into its compact buffers; the eight buffers are laid out [rank][tiles] as ncclGather delivers them - and are also sent through
    ncclGather itself (a one-rank communicator, the whole block of a view) - and de-interleaved by vkv_scatter_tiles(n_ranks = 8).
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
    # round 6: only the tiles of each view's screen rectangle (vkv_screen_tile_rect: derived from the uniforms alone, the same on every rank)
    # are scheduled and exchanged; the two views of a rank's launch have different tile counts
    full = [sp.make_params(view, proj) for view, proj in views]
    rects = [lib.screen_tile_rect(p.ray_cast, p.ray_gen, size, (tile, tile)) for p in full]
    per_rank = [multigpu.tiles_per_rank(r, world) for r in rects]
    for r_, n_ in zip(rects, per_rank):
        assert n_ * world <= 0.70 * total_tiles, "rectangle %s: the exchange moves %d of %d tiles" % (r_.as_tuple(), n_ * world, total_tiles)
    n = [t * tile * tile for t in per_rank]
    gathered = [torch.zeros((world, n[k], 4), dtype=torch.uint8, device="cuda") for k in range(len(views))]        # [rank][tiles] per view
    counts_r = torch.zeros((max(n), 3), dtype=torch.int32, device="cuda")
    events = np.zeros((len(views), world), np.int64)
    for r in range(world):
        scheds = [abi.full_frame_tiles(fw, fh, tile, tile, r, world, compact=True, rect=rect) for rect in rects]
        plist = []
        for k, (view, proj) in enumerate(views):
            assert per_rank[k] - 1 <= scheds[k].tile_count <= per_rank[k]
            p = sp.make_params(view, proj, scheds[k])
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = gathered[k][r].data_ptr(), None, None, None
            plist.append(p)
        ctx.render_batch(plist, st)        # ONE launch per rank: its tiles of both views
        for k, (view, proj) in enumerate(views):        # the rank's counters (a second, single-frame launch of the same tiles)
            p = sp.make_params(view, proj, scheds[k])
            check = torch.zeros((n[k], 4), dtype=torch.uint8, device="cuda")
            sp.draw(p, rgba8=check, counts=counts_r)
            torch.cuda.synchronize()
            assert torch.equal(check, gathered[k][r]), "rank %d, view %d: vkv_render and vkv_render_batch disagree" % (r, k)
            events[k, r] = int(counts_r[:scheds[k].tile_count * tile * tile, :2].to(torch.int64).sum().item())
    torch.cuda.synchronize()
    # (c) load balance of round-robin 16x16 tiles
    for k in range(len(views)):
        mean = events[k].mean()
        assert mean > 1e6
        assert float(np.abs(events[k] - mean).max()) <= 0.15 * mean, "view %d: per-rank events %r" % (k, events[k].tolist())
    rccl, comm = one_rank_communicator()
    try:
        for k, (view, proj) in enumerate(views):
            p_full = full[k]
            direct = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
            counts = torch.zeros((fh, fw, 3), dtype=torch.int32, device="cuda")
            sp.draw(p_full, rgba8=direct, counts=counts)
            # (a) de-interleave of the [rank][tiles] block
            image = torch.full((fh, fw, 4), 3, dtype=torch.uint8, device="cuda")
            ctx.scatter_tiles(gathered[k].data_ptr(), image.data_ptr(), size, (tile, tile), world, per_rank[k], 4, st, rect=rects[k])
            torch.cuda.synchronize()
            assert int(direct.to(torch.int64).sum().item()) > 0
            assert torch.equal(image, direct), "view %d: the frame assembled from 8 ranks' tiles differs from the direct render" % k
            # the same bytes through the native path at this size: ncclGather (RCCL, one rank) of the whole block, then the de-interleave
            recv = torch.zeros_like(gathered[k])
            image.fill_(5)
            ctx.gather_tiles(gathered[k].data_ptr(), recv.data_ptr(), gathered[k].numel(), 0, comm.value, st)
            ctx.scatter_tiles(recv.data_ptr(), image.data_ptr(), size, (tile, tile), world, per_rank[k], 4, st, rect=rects[k])
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
