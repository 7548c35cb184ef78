"""GPU: randomised tile schedules, launch variants, depth attachment and blend state against the oracle (VKV_TEST_FUZZ_SEEDS=n for more seeds)."""
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


def T_bind(ctx, v, scene, view, proj, size, ro, sched):
    sp = V.VolumeRenderSubpass(ctx, v, ro, size)
    return sp.bind(scene.params(view, proj, size, ro, tiles=sched))


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "12"))))
def test_tile_schedule_fuzz(ctx, seed):
    """The multi-GPU decomposition with one GPU playing every rank, all knobs random: frame size (ragged against the tiles), tile size
    (multiples of 16, x and y apart), number of ranks 1..9 (more ranks than tiles included), frames per vkv_render_batch launch 1..4
    (different views), skipping mode, ERT.  Each rank renders its interleaved compact schedule; the [rank][frame][tiles] blocks are
    de-interleaved by vkv_scatter_tiles as the owner of the launch does after the gather.  Every assembled frame == the oracle's
    full-frame RGBA8, and each rank's counters == the oracle's for its schedule."""
    rng = np.random.default_rng(12000 + seed)
    shape = tuple(int(x) for x in rng.integers(24, 72, size=3))
    scene = T.OracleScene(O.synth_volume(shape, int(rng.integers(0, 2)), int(rng.integers(1, 1 << 30))), abi.VolumeOptions(**T.APP_TF), int(rng.integers(2, 6)))
    v, tf = make_gpu_volume(ctx, scene)
    st_mode = int(rng.integers(1, 4))
    V.ComputeDistanceMap(ctx).compute(v, tf, st_mode)
    tw, th = 16 * int(rng.integers(1, 4)), 16 * int(rng.integers(1, 3))
    size = (int(rng.integers(17, 200)), int(rng.integers(17, 120)))
    world, frames = int(rng.integers(1, 10)), int(rng.integers(1, 5))
    ro = abi.RenderOptions(skipping_type=st_mode, clip_distance=1.0, early_ray_termination=bool(rng.integers(0, 2)))
    views = [T.orbit(float(rng.uniform(0, 360)), elevation=float(rng.uniform(-60, 60)), image_size=size) for _ in range(frames)]
    # odd seeds: every frame through the schedule of its own screen rectangle (vkv_screen_tile_rect, aligned to 1, 2 or 4 tiles) - the frames of a
    # launch then have different tile counts and lie back to back in a rank's block; even seeds: every tile of the image
    full = [scene.params(view, proj, size, ro, tiles=abi.full_frame_tiles(size[0], size[1], tw, th)) for view, proj in views]
    if seed % 2:
        rects = [lib.screen_tile_rect(p.ray_cast, p.ray_gen, size, (tw, th), int(rng.choice([1, 1, 2, 4]))) for p in full]
    else:
        rects = [abi.whole_image_rect(size[0], size[1], tw, th)] * frames
    tpr, off, total = multigpu.launch_layout(rects, world)
    tp = tw * th
    st = torch.cuda.current_stream().cuda_stream
    gathered = torch.full((world, total * tp, 4), 0x5A, dtype=torch.uint8, device="cuda")        # as the launch's owner receives it: [rank][frame][tiles]
    what = "seed %d: frame %s tiles %dx%d world %d frames %d mode %d rects %s" % (seed, size, tw, th, world, frames, st_mode, [r.as_tuple() for r in rects])
    for r in range(world):
        scheds = [abi.full_frame_tiles(size[0], size[1], tw, th, r, world, compact=True, rect=rect) for rect in rects]
        if max(s.tile_count for s in scheds) == 0:
            continue        # more ranks than tiles: this one has nothing to render (bench.py's ranks pass such schedules too)
        plist, counts = [], []
        for f, (view, proj) in enumerate(views):
            p = T_bind(ctx, v, scene, view, proj, size, ro, scheds[f])
            c = torch.zeros((max(1, scheds[f].tile_count) * tp, 3), dtype=torch.int32, device="cuda")  # (slots beyond the image edge stay unwritten)
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = gathered[r].data_ptr() + off[f] * tp * 4, None, c.data_ptr(), None
            plist.append(p)
            counts.append(c)
        if frames == 1:
            ctx.render(plist[0], st)
        else:
            ctx.render_batch(plist, st)        # (frames without a tile for this rank ride along)
        torch.cuda.synchronize()
        for f, (view, proj) in enumerate(views):
            if scheds[f].tile_count == 0:
                continue
            ref = scene.render(scene.params(view, proj, size, ro, tiles=scheds[f]))
            assert np.array_equal(counts[f].cpu().numpy().astype(np.uint32).reshape(ref.counts.shape), ref.counts), what + ", rank %d frame %d counters" % (r, f)
    for f, (view, proj) in enumerate(views):
        image = torch.full((size[1], size[0], 4), 3, dtype=torch.uint8, device="cuda")
        ctx.scatter_tiles(gathered.data_ptr() + off[f] * tp * 4, image.data_ptr(), size, (tw, th), world, total, 4, st, rect=rects[f])
        torch.cuda.synchronize()
        ref_full = scene.render(full[f], want_rgba8=True)
        assert np.array_equal(image.cpu().numpy(), ref_full.rgba8), what + ", assembled frame %d" % f


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "24"))))
def test_launch_variants_fuzz(ctx, seed):
    """The kernels a renderer actually launches, on the random configurations of test_render_fuzz: for every address-table kind
    (VkvTuning.address_tables 2 = per-voxel tables, 1 = two-level tables, 0 = address arithmetic in registers) a counted single launch
    (three counters == the oracle, colour / depth within tolerance), then the launches WITHOUT a counter buffer - the loop without the
    per-pixel counters where that instantiation exists (ESS + ERT + precomputed gradient), the counted loop elsewhere - as vkv_render and
    as a vkv_render_batch of three frames (the frame between two copies of a second view): float colour, RGBA8 and depth must be the
    counted launch's bits."""
    from tests.test_gpu_parity import COLOR_TOL, DEPTH_TOL, fuzz_case
    scene, v, params, ref, label = fuzz_case(ctx, 5000 + seed)
    size = (params.image_width, params.image_height)
    sp = V.VolumeRenderSubpass(ctx, v, params.options, size)
    st = torch.cuda.current_stream().cuda_stream
    other = abi.RenderParams.from_buffer_copy(params)        # a second view for the batch: the same camera mirrored in x (ddx negated)
    for i in range(3):
        other.ray_gen.dir00[i] = params.ray_gen.dir00[i] + (size[0] - 1) * params.ray_gen.ddx[i]
        other.ray_gen.ddx[i] = -params.ray_gen.ddx[i]

    def outputs():
        return (torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda"), torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda"),
                torch.full((size[1], size[0]), -1.0, dtype=torch.float32, device="cuda"))

    try:
        for tables in (2, 1, 0):
            ctx.set_tuning(address_tables=tables)
            what = "%s, address_tables %d" % (label, tables)
            p = sp.bind(params)
            col, rgba, dep = outputs()
            cnt = torch.full((size[1], size[0], 3), 0xFFFF, dtype=torch.int32, device="cuda")
            sp.draw(p, col, rgba, cnt, dep)
            torch.cuda.synchronize()
            assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), ref.counts), what + ": counters"
            assert float(np.abs(col.cpu().numpy() - ref.color).max()) <= COLOR_TOL and float(np.abs(dep.cpu().numpy() - ref.depth).max()) <= DEPTH_TOL, what
            col2, rgba2, dep2 = outputs()
            sp.draw(sp.bind(params), col2, rgba2, None, dep2)        # no counter buffer
            torch.cuda.synchronize()
            assert torch.equal(col2, col) and torch.equal(rgba2, rgba) and torch.equal(dep2, dep), what + ": launch without counters"
            plist, outs = [], []
            for q in (other, params, other):
                b = sp.bind(q)
                o = outputs()
                b.d_out_color, b.d_out_rgba8, b.d_out_depth, b.d_out_counts = o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), None
                b.d_in_depth, b.blend_over_target = None, 0
                plist.append(b)
                outs.append(o)
            ctx.render_batch(plist, st)
            torch.cuda.synchronize()
            assert torch.equal(outs[1][0], col) and torch.equal(outs[1][1], rgba) and torch.equal(outs[1][2], dep), what + ": batch launch without counters"
            assert torch.equal(outs[0][1], outs[2][1]) and torch.equal(outs[0][0], outs[2][0]), what + ": the two copies of the second view differ"
    finally:
        ctx.set_tuning(address_tables=2)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "16"))))
def test_depth_attachment_and_blend_fuzz(ctx, seed):
    """options.depth_attachment (frag:122-165: fragments behind the scene depth are discarded, rays end at it) and the subpass's blend state
    (volume_render_subpass.cpp:176-190: premultiplied 'over' onto the target's contents) on the random configurations of test_render_fuzz,
    with a random scene depth buffer (reverse-Z walls of random depth in random columns, 'far' elsewhere) and random target contents:
    counters, float colour, RGBA8 and depth against the oracle, with and without a counter buffer."""
    from tests.test_gpu_parity import COLOR_TOL, DEPTH_TOL, dev, fuzz_case
    scene, v, params, _, label = fuzz_case(ctx, 9000 + seed)
    rng = np.random.default_rng(77000 + seed)
    size = (params.image_width, params.image_height)
    p = abi.RenderParams.from_buffer_copy(params)
    p.options.depth_attachment = 1
    in_depth = np.zeros((size[1], size[0]), np.float32)
    for _ in range(int(rng.integers(1, 4))):
        x0 = int(rng.integers(0, size[0]))
        in_depth[:, x0:x0 + int(rng.integers(1, size[0]))] = float(rng.choice([0.1 / 90.0, 0.1 / 110.0, 0.1 / 150.0, 0.5, 1e-6]))
    blend = bool(rng.integers(0, 2))
    tgt_color = rng.random((size[1], size[0], 4), dtype=np.float32) if blend else None
    tgt_rgba8 = rng.integers(0, 256, (size[1], size[0], 4), dtype=np.uint8) if blend else None
    ref = scene.render(p, in_depth=in_depth, target_color=tgt_color, target_rgba8=tgt_rgba8, want_rgba8=True)
    sp = V.VolumeRenderSubpass(ctx, v, p.options, size)
    for with_counts in (True, False):
        q = sp.bind(p)
        color = torch.from_numpy(tgt_color).cuda() if blend else torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda")
        rgba8 = torch.from_numpy(tgt_rgba8).cuda() if blend else torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
        counts = torch.full((size[1], size[0], 3), 77, dtype=torch.int32, device="cuda") if with_counts else None
        depth = torch.full((size[1], size[0]), -1.0, dtype=torch.float32, device="cuda")
        sp.draw(q, color, rgba8, counts, depth, in_depth=dev(in_depth), blend=blend)
        torch.cuda.synchronize()
        what = "%s, blend %d, counters %d" % (label, blend, with_counts)
        if with_counts:
            assert np.array_equal(counts.cpu().numpy().astype(np.uint32), ref.counts), what
        assert float(np.abs(color.cpu().numpy() - ref.color).max()) <= COLOR_TOL, what
        assert np.array_equal(rgba8.cpu().numpy(), ref.rgba8), what
        got = depth.cpu().numpy()
        frag = got != -1.0
        if frag.any():
            assert float(np.abs(got[frag] - ref.depth[frag]).max()) <= DEPTH_TOL, what
