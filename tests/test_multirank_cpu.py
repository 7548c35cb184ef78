"""CPU, world_size 2 over gloo: the screen-tile sharding + gather + de-interleave used by bench.py for N > 1.
Each rank renders its interleaved compact tiles (the CPU oracle stands in for the device kernel here), the buffers are
gathered to rank 0 with vkvolume_amd.multigpu.TileGather, and the de-interleaved frame must equal the single-rank frame.
Round 6: only the tiles of the frame's screen rectangle (vkv_screen_tile_rect, derived by every rank from the uniforms) are scheduled and
exchanged; the frames of a launch have their own rectangles and, with `spread`, their own owners."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vkv_oracle as O
from tests import helpers as T
from vkvolume_amd import abi, lib, multigpu

FRAME = (150, 70)  # not a multiple of the 16-pixel tile
WORLD = 2


def _scene():
    scene = T.OracleScene(O.synth_volume((48, 40, 36), 1, 77), abi.VolumeOptions(**T.APP_TF), 4)
    view, proj = T.orbit(25.0, image_size=FRAME)
    uniforms = O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, FRAME, scene.extent, scene.map_extent)
    return scene, view, proj, uniforms


def _worker(rank, port, out_path, rotate, use_rect):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        scene, view, proj, uniforms = _scene()
        g = multigpu.TileGather(dist, rank, WORLD, FRAME, 16, 4, device="cpu", any_root=rotate)
        opts = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        # the tile rectangle of the frame, derived on every rank from the same uniforms (smaller than the image for this view)
        rect = lib.screen_tile_rect(uniforms[1], uniforms[2], FRAME) if use_rect else None
        if use_rect:
            assert rect.tiles < g.total_tiles
        frames = []
        for k in range(3):  # three frames through the two buffers: exercises buffer reuse + ordering
            b = k % 2
            flat = g.finish(b)
            if flat is not None:
                frames.append(multigpu.deinterleave_reference(flat.numpy(), FRAME, 16, WORLD, rect))
            p = scene.params(view, proj, FRAME, opts, tiles=g.rect_schedule(rect), uniforms=uniforms)
            r = scene.render(p, n_threads=2, want_rgba8=True)
            g.buffers[b][:r.rgba8.shape[0]].copy_(torch.from_numpy(r.rgba8))
            g.start(b, k % WORLD if rotate else 0, rect)  # rotate: frame k is assembled on rank k mod WORLD
        for b in (1, 0):
            flat = g.finish(b)
            if flat is not None:
                frames.append(multigpu.deinterleave_reference(flat.numpy(), FRAME, 16, WORLD, rect))
        np.save("%s.%d.npy" % (out_path, rank), np.stack(frames) if frames else np.zeros((0, FRAME[1], FRAME[0], 4), np.uint8))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rotate,use_rect", [(False, False), (True, False), (True, True)])
def test_two_rank_tile_gather_reassembles_the_frame(tmp_path, rotate, use_rect):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frames")
    mp.spawn(_worker, args=(port, out, rotate, use_rect), nprocs=WORLD, join=True)
    per_rank = [np.load("%s.%d.npy" % (out, r)) for r in range(WORLD)]
    assert [f.shape[0] for f in per_rank] == ([2, 1] if rotate else [3, 0])  # frames 0 and 2 on rank 0, frame 1 on rank 1
    frames = np.concatenate(per_rank)
    scene, view, proj, uniforms = _scene()
    full = scene.params(view, proj, FRAME, abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0),
                        tiles=abi.full_frame_tiles(FRAME[0], FRAME[1]), uniforms=uniforms)
    ref = scene.render(full, want_rgba8=True).rgba8
    assert ref[..., 3].max() > 0
    for f in frames:
        assert np.array_equal(f, ref)


LAUNCHES = [(0, (25.0, 100.0, 190.0)), (1, (280.0, 330.0))]  # (buffer set, azimuths of the launch's frames)


def _batch_worker(rank, port, out_path, host_staging, use_rect, spread):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        scene, _, proj, _ = _scene()
        frames_per_launch = 3
        g = multigpu.BatchTileGather(dist, rank, WORLD, FRAME, 16, 4, device="cpu", frames=frames_per_launch, n_sets=2, any_root=True,
                                     host_staging=host_staging)
        opts = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        done, rects_of = [], {}
        for launch, (b, azimuths) in enumerate(LAUNCHES):
            uniforms = []
            for az in azimuths:
                view, _ = T.orbit(az, image_size=FRAME)
                uniforms.append((view, O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, FRAME, scene.extent, scene.map_extent)))
            rects = [lib.screen_tile_rect(u[1], u[2], FRAME) if use_rect else g.whole for _, u in uniforms]
            rects_of[b] = rects
            tpr, off, total = multigpu.launch_layout(rects, WORLD)
            block = np.zeros((total * 256, 4), np.uint8)
            for j, (view, u) in enumerate(uniforms):
                r = scene.render(scene.params(view, proj, FRAME, opts, tiles=g.rect_schedule(rects[j]), uniforms=u), n_threads=2, want_rgba8=True)
                assert r.rgba8.shape[0] <= tpr[j] * 256
                block[off[j] * 256:off[j] * 256 + r.rgba8.shape[0]] = r.rgba8  # frame j starts off[j] tiles into this rank's block
            g.sets[b][:block.size].copy_(torch.from_numpy(block.reshape(-1)))
            # launch l is assembled on rank l mod WORLD (one collective for its frames) - or its frames on ranks (l + j) mod WORLD (one gather each)
            roots = [(launch + j) % WORLD for j in range(len(azimuths))] if spread else None
            g.start(b, launch % WORLD, rects, roots)
        for launch, (b, azimuths) in enumerate(LAUNCHES):
            for f, src, stride, rect in g.finish(b):
                # (pointer, rank stride) -> the numpy view the de-interleave reads: WORLD blocks of `stride` tiles, the frame's tiles first
                first = (src - g.flat[b].data_ptr()) // 4
                flat = g.flat[b].numpy().reshape(-1, 4)
                ranks = np.stack([flat[first + r * stride * 256:first + r * stride * 256 + multigpu.tiles_per_rank(rect, WORLD) * 256] for r in range(WORLD)])
                done.append((azimuths[f], multigpu.deinterleave_reference(ranks, FRAME, 16, WORLD, rect)))
        np.save("%s.%d.npy" % (out_path, rank), np.stack([f for _, f in done]) if done else np.zeros((0, FRAME[1], FRAME[0], 4), np.uint8))
        np.save("%s.%d.az.npy" % (out_path, rank), np.array([a for a, _ in done]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("host_staging,use_rect,spread", [(False, False, False), (True, False, False), (False, True, False), (True, True, True), (False, True, True)])
def test_two_rank_batch_gather_reassembles_every_frame_of_a_launch(tmp_path, host_staging, use_rect, spread):
    """BatchTileGather: the compact tile buffers of all frames of a launch travel in one collective to the launch's owner
    (host_staging: through host copies, the path bench.py --backend gloo takes for device buffers); use_rect: every frame with its own
    screen rectangle, the block then holds [frame][tiles of its rectangle / WORLD]; spread: the frames of a launch have different owners."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "batch")
    mp.spawn(_batch_worker, args=(port, out, host_staging, use_rect, spread), nprocs=WORLD, join=True)
    scene, _, proj, _ = _scene()
    opts = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    seen = 0
    expect_owner = {}
    for launch, (b, azimuths) in enumerate(LAUNCHES):
        for j, a in enumerate(azimuths):
            expect_owner[a] = (launch + j) % WORLD if spread else launch % WORLD
    for rank in range(WORLD):
        frames, az = np.load("%s.%d.npy" % (out, rank)), np.load("%s.%d.az.npy" % (out, rank))
        assert sorted(az) == sorted(a for a, o in expect_owner.items() if o == rank)
        for a, f in zip(az, frames):
            view, _ = T.orbit(float(a), image_size=FRAME)
            uniforms = O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, FRAME, scene.extent, scene.map_extent)
            full = scene.params(view, proj, FRAME, opts, tiles=abi.full_frame_tiles(FRAME[0], FRAME[1]), uniforms=uniforms)
            assert np.array_equal(f, scene.render(full, want_rgba8=True).rgba8)
            seen += 1
    assert seen == 5
