"""GPU: the start-order feedback of registered targets (vkv_register_target / vkv_forget_target) against the oracle."""
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
