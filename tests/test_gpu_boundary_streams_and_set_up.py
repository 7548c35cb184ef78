"""GPU: one context on several HIP streams; several volumes in one subpass (the C++ host mirror through vkv_offscreen)."""
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


def run_offscreen(tmp_path, *flags):
    assert os.path.exists(EXE), "vkv_offscreen not built (run __graft_entry__.build())"
    out = subprocess.run([EXE, *flags], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()
    return out.stdout.decode()


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
