"""CPU, world_size 2 over gloo: the screen-tile sharding + gather + de-interleave used by bench.py for N > 1.
Each rank renders its interleaved compact tiles (the CPU oracle stands in for the device kernel here), the buffers are
gathered to rank 0 with vkvolume_amd.multigpu.TileGather, and the de-interleaved frame must equal the single-rank frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vkv_oracle as O
from tests import helpers as T
from vkvolume_amd import abi, multigpu

FRAME = (150, 70)  # not a multiple of the 16-pixel tile
WORLD = 2


def _scene():
    scene = T.OracleScene(O.synth_volume((48, 40, 36), 1, 77), abi.VolumeOptions(**T.APP_TF), 4)
    view, proj = T.orbit(25.0, image_size=FRAME)
    uniforms = O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, FRAME, scene.extent, scene.map_extent)
    return scene, view, proj, uniforms


def _worker(rank, port, out_path, rotate):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        scene, view, proj, uniforms = _scene()
        g = multigpu.TileGather(dist, rank, WORLD, FRAME, 16, 4, device="cpu", any_root=rotate)
        opts = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        frames = []
        for k in range(3):  # three frames through the two buffers: exercises buffer reuse + ordering
            b = k % 2
            flat = g.finish(b)
            if flat is not None:
                frames.append(multigpu.deinterleave_reference(flat.numpy(), FRAME, 16, WORLD))
            p = scene.params(view, proj, FRAME, opts, tiles=g.schedule, uniforms=uniforms)
            r = scene.render(p, n_threads=2, want_rgba8=True)
            g.buffers[b][:r.rgba8.shape[0]].copy_(torch.from_numpy(r.rgba8))
            g.start(b, k % WORLD if rotate else 0)  # rotate: frame k is assembled on rank k mod WORLD
        for b in (1, 0):
            flat = g.finish(b)
            if flat is not None:
                frames.append(multigpu.deinterleave_reference(flat.numpy(), FRAME, 16, WORLD))
        np.save("%s.%d.npy" % (out_path, rank), np.stack(frames) if frames else np.zeros((0, FRAME[1], FRAME[0], 4), np.uint8))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rotate", [False, True])
def test_two_rank_tile_gather_reassembles_the_frame(tmp_path, rotate):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "frames")
    mp.spawn(_worker, args=(port, out, rotate), nprocs=WORLD, join=True)
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


def _batch_worker(rank, port, out_path, host_staging=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        scene, _, proj, _ = _scene()
        frames_per_launch = 3
        g = multigpu.BatchTileGather(dist, rank, WORLD, FRAME, 16, 4, device="cpu", frames=frames_per_launch, n_sets=2, any_root=True,
                                     host_staging=host_staging)
        opts = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        done = []
        launches = [(0, 3, (25.0, 100.0, 190.0)), (1, 2, (280.0, 330.0))]  # (buffer set, frames in the launch, their azimuths)
        for launch, (b, n, azimuths) in enumerate(launches):
            for j, az in enumerate(azimuths):
                view, _ = T.orbit(az, image_size=FRAME)
                uniforms = O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, FRAME, scene.extent, scene.map_extent)
                r = scene.render(scene.params(view, proj, FRAME, opts, tiles=g.schedule, uniforms=uniforms), n_threads=2, want_rgba8=True)
                g.buffers[b * frames_per_launch + j][:r.rgba8.shape[0]].copy_(torch.from_numpy(r.rgba8))
            g.start(b, launch % WORLD, n)  # launch l is assembled on rank l mod WORLD: one collective for its n frames
        for launch, (b, n, azimuths) in enumerate(launches):
            got = g.finish(b)
            if got is not None:
                flat, nf = got
                assert nf == n
                for j in range(nf):
                    done.append((azimuths[j], multigpu.deinterleave_reference(flat[:, j].numpy(), FRAME, 16, WORLD)))
        np.save("%s.%d.npy" % (out_path, rank), np.stack([f for _, f in done]))
        np.save("%s.%d.az.npy" % (out_path, rank), np.array([a for a, _ in done]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("host_staging", [False, True])
def test_two_rank_batch_gather_reassembles_every_frame_of_a_launch(tmp_path, host_staging):
    """BatchTileGather: the compact tile buffers of all frames of a launch travel in one collective to the launch's owner
    (host_staging: through host copies, the path bench.py --backend gloo takes for device buffers)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "batch")
    mp.spawn(_batch_worker, args=(port, out, host_staging), nprocs=WORLD, join=True)
    scene, _, proj, _ = _scene()
    opts = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    seen = 0
    for rank, expect in ((0, [25.0, 100.0, 190.0]), (1, [280.0, 330.0])):
        frames, az = np.load("%s.%d.npy" % (out, rank)), np.load("%s.%d.az.npy" % (out, rank))
        assert list(az) == expect
        for a, f in zip(az, frames):
            view, _ = T.orbit(float(a), image_size=FRAME)
            uniforms = O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, FRAME, scene.extent, scene.map_extent)
            full = scene.params(view, proj, FRAME, opts, tiles=abi.full_frame_tiles(FRAME[0], FRAME[1]), uniforms=uniforms)
            assert np.array_equal(f, scene.render(full, want_rgba8=True).rgba8)
            seen += 1
    assert seen == 5
