"""GPU: vkv_render / vkv_render_batch captured into hipGraphs and replayed."""
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


def test_render_launches_captured_into_a_hip_graph_replay_the_same_frames(ctx):
    """hipGraph capture of the render entry points (include/vkvolume_amd.h: vkv_prepare_render first): eight vkv_render launches, and one
    vkv_render_batch launch of eight frames, captured on a stream and replayed several times - with host allocations churned between the
    capture and the replays - must produce the frames of direct launches.  (The argument blocks of the captured batch launch live in a
    pinned copy the context keeps: the graph's copy node reads its source at every replay; round 4 found the temporary it used to point
    to.)"""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((96, 80, 72), 1, 0x5EED0008), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
    size = (320, 192)
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    sp = V.VolumeRenderSubpass(ctx, v, ro, size)
    plist, direct, bufs = [], [], []
    for k in range(8):
        view, proj = T.orbit(45.0 * k, image_size=size)
        p = sp.bind(scene.params(view, proj, size, ro))
        buf = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
        p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = buf.data_ptr(), None, None, None
        ctx.render(p, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        direct.append(buf.clone())
        plist.append(p)
        bufs.append(buf)
    assert int(direct[0].to(torch.int64).sum().item()) > 0
    ref = scene.render(plist[3], want_rgba8=True)
    assert np.array_equal(direct[3].cpu().numpy(), ref.rgba8)
    s = torch.cuda.Stream()
    ctx.prepare_render(plist, s.cuda_stream)  # tables + the stream's scratch block in place: a capture allows no event query
    torch.cuda.synchronize()
    graphs = []
    for batch in (False, True):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            st = torch.cuda.current_stream().cuda_stream
            if batch:
                ctx.render_batch(plist, st)
            else:
                for p in plist:
                    ctx.render(p, st)
        graphs.append(g)
    torch.cuda.synchronize()
    churn = [np.random.default_rng(i).integers(0, 255, size=200_000, dtype=np.uint8) for i in range(64)]  # reuse what the capture call freed
    for rep in range(3):
        for g, name in zip(graphs, ("8 x vkv_render", "vkv_render_batch")):
            for b in bufs:
                b.fill_(9)
            g.replay()
            torch.cuda.synchronize()
            for k in range(8):
                assert torch.equal(bufs[k], direct[k]), "replay %d of the captured %s: view %d differs from the direct launch" % (rep, name, k)
        churn = [c[::-1].copy() for c in churn]
    del graphs
    ctx.release_stream(s.cuda_stream)


def test_more_captured_batch_launches_than_pinned_slots_and_trim(ctx):
    """A renderer that re-captures when the camera moves: 40 vkv_render_batch launches captured by one context (vkv_create sets 32 pinned
    slots aside; the later ones allocate their block during the capture), every graph replays its own frames; vkv_trim gives the blocks
    back and a capture after it works again."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((64, 56, 48), 1, 0x5EED0009), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
    size = (160, 96)
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    sp = V.VolumeRenderSubpass(ctx, v, ro, size)
    bufs = [torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    s = torch.cuda.Stream()

    def pair(k):
        out = []
        for j in range(2):
            view, proj = T.orbit(9.0 * k + 4.0 * j, image_size=size)
            p = sp.bind(scene.params(view, proj, size, ro))
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = bufs[j].data_ptr(), None, None, None
            out.append(p)
        return out

    def capture(plist):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ctx.render_batch(plist, torch.cuda.current_stream().cuda_stream)
        return g

    def check(g, plist, what):
        for b in bufs:
            b.fill_(7)
        g.replay()
        torch.cuda.synchronize()
        for j in range(2):
            ref = scene.render(plist[j], want_rgba8=True)
            assert np.array_equal(bufs[j].cpu().numpy(), ref.rgba8), what

    ctx.prepare_render(pair(0), s.cuda_stream)
    torch.cuda.synchronize()
    graphs = [(capture(pl), pl) for pl in (pair(k) for k in range(40))]
    torch.cuda.synchronize()
    for k in (0, 31, 32, 39):
        check(graphs[k][0], graphs[k][1], "captured launch %d" % k)
    del graphs
    ctx.trim()
    ctx.prepare_render(pair(41), s.cuda_stream)
    torch.cuda.synchronize()
    pl = pair(41)
    check(capture(pl), pl, "capture after vkv_trim")
    ctx.release_stream(s.cuda_stream)


def test_captured_batch_launches_replay_concurrently_and_release_their_slots(ctx):
    """Graphs are replayed wherever the caller likes: two vkv_render_batch launches captured on ONE stream, replayed at the same time on two
    OTHER streams while a live vkv_render_batch runs on the capture stream - each launch has an argument block of its own on the device, so
    none of them may pick up another's cameras or targets (round 4's captured launches went through the capture stream's scratch block).
    Then the renderer's loop: capture, replay, destroy, vkv_release_captured, 48 times over - the slots come back, nothing is allocated."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((64, 56, 48), 1, 0x5EED000A), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
    size = (160, 96)
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    sp = V.VolumeRenderSubpass(ctx, v, ro, size)
    n_sets, per = 3, 4   # two captured launches + one live one, four frames each, every frame into a target of its own
    bufs = [[torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda") for _ in range(per)] for _ in range(n_sets)]

    def plist(k):
        out = []
        for j in range(per):
            view, proj = T.orbit(37.0 * k + 11.0 * j, image_size=size)
            p = sp.bind(scene.params(view, proj, size, ro))
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = bufs[k % n_sets][j].data_ptr(), None, None, None
            out.append(p)
        return out

    lists = [plist(k) for k in range(n_sets)]
    refs = [[scene.render(p, want_rgba8=True).rgba8 for p in pl] for pl in lists]
    s_cap, s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    ctx.prepare_render(lists[0], s_cap.cuda_stream)
    torch.cuda.synchronize()

    def capture(pl):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s_cap):
            ctx.render_batch(pl, torch.cuda.current_stream().cuda_stream)
        return g

    g0, g1 = capture(lists[0]), capture(lists[1])
    torch.cuda.synchronize()
    for rep in range(6):
        for set_ in bufs:
            for b in set_:
                b.fill_(3)
        torch.cuda.synchronize()
        with torch.cuda.stream(s_a):
            g0.replay()
        with torch.cuda.stream(s_b):
            g1.replay()
        ctx.render_batch(lists[2], s_cap.cuda_stream)   # a live launch on the capture stream, through that stream's scratch block
        with torch.cuda.stream(s_a):
            g0.replay()
        torch.cuda.synchronize()
        for k in range(n_sets):
            for j in range(per):
                assert np.array_equal(bufs[k][j].cpu().numpy(), refs[k][j]), "repetition %d: launch %d frame %d" % (rep, k, j)
    del g0, g1
    ctx.release_captured(s_cap.cuda_stream)
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(48):   # more captures than the context has slots: each gives its slot back
        g = capture(lists[k % 2])
        g.replay()
        torch.cuda.synchronize()
        del g
        ctx.release_captured(s_cap.cuda_stream)
    for j in range(per):
        assert np.array_equal(bufs[1][j].cpu().numpy(), refs[1][j])
    assert torch.cuda.mem_get_info()[0] >= free0 - (1 << 20), "captures that release their slots must not grow the context"
    ctx.release_stream(s_cap.cuda_stream)
