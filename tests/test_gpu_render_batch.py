"""GPU: vkv_render_batch against single launches (tile-per-workgroup and pulling kernels), launches without the per-pixel counters, the sample-count test output."""
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


@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
def test_render_batch_equals_single_launches(ctx, skipping_type):
    """vkv_render_batch: n frames (different cameras, one of them another volume) in one launch == n vkv_render calls, bit for bit."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scenes = [T.OracleScene(O.synth_volume((80, 72, 64), 1, 70 + k), opt, 4) for k in range(2)]
    vols = [make_gpu_volume(ctx, s) for s in scenes]
    for v, tf in vols:
        V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (144, 80)
    ro = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0)
    frames = []
    for k, az in enumerate((0.0, 50.0, 111.0, 200.0, 290.0)):
        scene, (v, tf) = scenes[k % 2], vols[k % 2]
        view, proj = T.orbit(az, image_size=size)
        p = V.VolumeRenderSubpass(ctx, v, ro, size).bind(scene.params(view, proj, size, ro))
        frames.append((scene, p))
    st = torch.cuda.current_stream().cuda_stream

    def outputs():
        return [dict(color=torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda"),
                     counts=torch.full((size[1], size[0], 3), 9, dtype=torch.int32, device="cuda"),
                     depth=torch.full((size[1], size[0]), -1.0, dtype=torch.float32, device="cuda"),
                     rgba8=torch.full((size[1], size[0], 4), 7, dtype=torch.uint8, device="cuda")) for _ in frames]

    def point(p, o):
        p.d_out_color, p.d_out_counts, p.d_out_depth, p.d_out_rgba8 = (o[k].data_ptr() for k in ("color", "counts", "depth", "rgba8"))

    single, batch = outputs(), outputs()
    for (scene, p), o in zip(frames, single):
        point(p, o)
        ctx.render(p, st)
    plist = []
    for (scene, p), o in zip(frames, batch):
        q = abi.RenderParams.from_buffer_copy(p)
        point(q, o)
        plist.append(q)
    ctx.render_batch(plist, st)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(single, batch)):
        for k in a:
            assert torch.equal(a[k], b[k]), "frame %d: %s differs between the batch and the single launch" % (i, k)
    ref = frames[3][0].render(frames[3][1])
    assert np.array_equal(batch[3]["counts"].cpu().numpy().astype(np.uint32), ref.counts)
    # a frame that needs another kernel variant is refused
    bad = abi.RenderParams.from_buffer_copy(plist[1])
    bad.options.early_ray_termination = 0
    with pytest.raises(lib.VkvError):
        ctx.render_batch([plist[0], bad], st)


@pytest.mark.gpu
@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
def test_render_batch_pull_kernel_equals_single_launches(ctx, skipping_type):
    """VkvTuning.batch_mode = 1 (resident workgroups, waves pull 8x8 units from per-XCD ticket counters): same frames, bit for bit."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((96, 80, 72), 1, 91), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (208, 112)
    ro = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0)
    st = torch.cuda.current_stream().cuda_stream
    params = []
    for az in (0.0, 40.0, 95.0, 170.0, 230.0, 300.0, 345.0):
        view, proj = T.orbit(az, image_size=size)
        params.append(V.VolumeRenderSubpass(ctx, v, ro, size).bind(scene.params(view, proj, size, ro)))

    def outputs():
        return [dict(color=torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda"),
                     counts=torch.full((size[1], size[0], 3), 9, dtype=torch.int32, device="cuda"),
                     depth=torch.full((size[1], size[0]), -1.0, dtype=torch.float32, device="cuda"),
                     rgba8=torch.full((size[1], size[0], 4), 7, dtype=torch.uint8, device="cuda")) for _ in params]

    def point(p, o):
        p.d_out_color, p.d_out_counts, p.d_out_depth, p.d_out_rgba8 = (o[k].data_ptr() for k in ("color", "counts", "depth", "rgba8"))

    single, pulled = outputs(), outputs()
    for p, o in zip(params, single):
        point(p, o)
        ctx.render(p, st)
    plist = []
    for p, o in zip(params, pulled):
        q = abi.RenderParams.from_buffer_copy(p)
        point(q, o)
        plist.append(q)
    ctx.set_tuning(batch_mode=1)
    try:
        for _ in range(2):  # twice: the ticket counters are re-armed by every launch
            ctx.render_batch(plist, st)
        torch.cuda.synchronize()
    finally:
        ctx.set_tuning(batch_mode=0)
    for i, (a, b) in enumerate(zip(single, pulled)):
        for k in a:
            assert torch.equal(a[k], b[k]), "frame %d: %s differs between the pull kernel and the single launch" % (i, k)


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


@pytest.mark.parametrize("skipping_type", [abi.SKIP_BLOCK, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
def test_sample_count_test_mode_without_a_counter_buffer(ctx, skipping_type):
    """Test::NumTextureSamples (frag:324-334: the pixel's colour is (volume samples + map probes) / n_steps_max) with early ray
    termination ON and NO d_out_counts - the reference's GUI switches the test mode independently of ERT and of the skipping type
    (src/volume_render.cpp:539).  The launchers run the loop without the per-pixel counters for launches that have no counter buffer;
    this mode needs them whatever the outputs are (every marched pixel came out (0, 0, 0, 1)).  Single launch and batch
    launch, both address-table kinds, float colour and RGBA8 against the oracle."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((88, 72, 64), 1, 0x5EED0004), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (160, 96)
    ro = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0, early_ray_termination=True, test=abi.TEST_NUM_TEXTURE_SAMPLES)
    st = torch.cuda.current_stream().cuda_stream
    plist, refs = [], []
    for az in (10.0, 200.0):
        view, proj = T.orbit(az, image_size=size)
        params = scene.params(view, proj, size, ro)
        plist.append(V.VolumeRenderSubpass(ctx, v, ro, size).bind(params))
        refs.append(scene.render(params, want_rgba8=True))
    assert refs[0].counts[..., 0].sum() > 0 and refs[0].counts[..., 1].sum() > 0
    assert float(refs[0].color[..., 0].max()) > 0.0  # the grey level of the count output

    def outputs():
        return [dict(color=torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda"),
                     rgba8=torch.full((size[1], size[0], 4), 7, dtype=torch.uint8, device="cuda")) for _ in plist]

    def point(p, o):
        q = abi.RenderParams.from_buffer_copy(p)
        q.d_out_color, q.d_out_rgba8, q.d_out_counts, q.d_out_depth = o["color"].data_ptr(), o["rgba8"].data_ptr(), None, None
        return q

    try:
        for tables in (2, 1, 0):
            ctx.set_tuning(address_tables=tables)
            single, batch = outputs(), outputs()
            for p, o in zip(plist, single):
                ctx.render(point(p, o), st)
            ctx.render_batch([point(p, o) for p, o in zip(plist, batch)], st)
            torch.cuda.synchronize()
            for i, ref in enumerate(refs):
                for name, got in (("vkv_render", single[i]), ("vkv_render_batch", batch[i])):
                    assert np.array_equal(got["color"].cpu().numpy(), ref.color), "%s, tables %d, view %d: count colour differs from the oracle" % (name, tables, i)
                    assert np.array_equal(got["rgba8"].cpu().numpy(), ref.rgba8), "%s, tables %d, view %d: RGBA8 differs from the oracle" % (name, tables, i)
    finally:
        ctx.set_tuning(address_tables=2)


@pytest.mark.parametrize("seed", range(int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "16"))))
def test_fill_outside_schedules_complete_the_frame(ctx, seed):
    """VkvTileSchedule.fill_outside (round 6): a frame rendered through the tile rectangle of its clipped box, with the workgroups of a
    vkv_render_batch launch writing the no-fragment result to every pixel outside it, equals the whole-image schedule's frame bit for bit - float
    colour, RGBA8, the three counters, depth; with the depth attachment (outside pixels carry the scene depth) and with the blend state (outside
    pixels keep the target's contents) - on the random configurations of test_render_fuzz (cameras inside / beside the box included), against
    the whole-image launch and against the oracle.  vkv_render (one frame, argument block by value) renders the same frame by scheduling the whole
    image; the oracle DEFINES a fill_outside schedule's result as the whole-image one."""
    from tests.test_gpu_parity import COLOR_TOL, dev, fuzz_case
    scene, v, params, _, label = fuzz_case(ctx, 300 + seed)
    rng = np.random.default_rng(5100 + seed)
    size = (params.image_width, params.image_height)
    variant = ("plain", "depth", "blend")[seed % 3]
    p = abi.RenderParams.from_buffer_copy(params)
    in_depth = tgt_color = tgt_rgba8 = None
    if variant != "plain":
        p.options.depth_attachment = 1
        in_depth = np.zeros((size[1], size[0]), np.float32)
        x0 = int(rng.integers(0, size[0]))
        in_depth[:, x0:x0 + int(rng.integers(1, size[0]))] = float(rng.choice([0.1 / 90.0, 0.1 / 150.0, 0.5]))
    if variant == "blend":
        tgt_color = rng.random((size[1], size[0], 4), dtype=np.float32)
        tgt_rgba8 = rng.integers(0, 256, (size[1], size[0], 4), dtype=np.uint8)
    blend = variant == "blend"
    rect = lib.screen_tile_rect(p.ray_cast, p.ray_gen, size, (16, 16), int(rng.choice([1, 1, 2])))
    pf = abi.RenderParams.from_buffer_copy(p)
    pf.tiles = abi.full_frame_tiles(size[0], size[1], 16, 16, rect=rect, fill_outside=True)
    assert pf.tiles.fill_outside == 1 and pf.tiles.tile_count == rect.tiles
    ref = scene.render(p, in_depth=in_depth, target_color=tgt_color, target_rgba8=tgt_rgba8, want_rgba8=True)
    ref_fill = scene.render(pf, in_depth=in_depth, target_color=tgt_color, target_rgba8=tgt_rgba8, want_rgba8=True)
    assert np.array_equal(ref.counts, ref_fill.counts) and np.array_equal(ref.color, ref_fill.color) and np.array_equal(ref.rgba8, ref_fill.rgba8)
    sp = V.VolumeRenderSubpass(ctx, v, p.options, size)
    st = torch.cuda.current_stream().cuda_stream

    def targets():
        color = torch.from_numpy(tgt_color).cuda() if blend else torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda")
        rgba8 = torch.from_numpy(tgt_rgba8).cuda() if blend else torch.full((size[1], size[0], 4), 0x5A, dtype=torch.uint8, device="cuda")
        counts = torch.full((size[1], size[0], 3), 77, dtype=torch.int32, device="cuda")
        depth = torch.full((size[1], size[0]), -1.0, dtype=torch.float32, device="cuda")
        return color, rgba8, counts, depth

    def bound(src, outs):
        q = sp.bind(src)
        q.d_out_color, q.d_out_rgba8, q.d_out_counts, q.d_out_depth = (t.data_ptr() for t in outs)
        q.d_in_depth, q.blend_over_target = (dev(in_depth).data_ptr() if in_depth is not None else None), 1 if blend else 0
        return q

    keep = dev(in_depth) if in_depth is not None else None  # (kept alive; bound() uploads its own copies per call)
    what = "%s, %s, rect %s of %dx%d tiles" % (label, variant, rect.as_tuple(), -(-size[0] // 16), -(-size[1] // 16))
    # the whole-image schedule: one launch of two frames (the reference result on the device)
    whole = [targets(), targets()]
    q_whole = [bound(p, o) for o in whole]
    depth_keep = [dev(in_depth) for _ in range(6)] if in_depth is not None else []
    for i, q in enumerate(q_whole):
        if in_depth is not None:
            q.d_in_depth = depth_keep[i].data_ptr()
    ctx.render_batch(q_whole, st)
    # the rectangle's schedule with fill_outside: the same launch shape
    fill = [targets(), targets()]
    q_fill = [bound(pf, o) for o in fill]
    for i, q in enumerate(q_fill):
        if in_depth is not None:
            q.d_in_depth = depth_keep[2 + i].data_ptr()
    ctx.render_batch(q_fill, st)
    # vkv_render of the same schedule (renders the whole image)
    single = targets()
    q_single = bound(pf, single)
    if in_depth is not None:
        q_single.d_in_depth = depth_keep[4].data_ptr()
    ctx.render(q_single, st)
    torch.cuda.synchronize()
    for name, outs in (("batch, frame 0", fill[0]), ("batch, frame 1", fill[1]), ("vkv_render", single)):
        for a, b, field in zip(outs, whole[0], ("color", "rgba8", "counts", "depth")):
            assert torch.equal(a, b), "%s: %s of the fill_outside schedule (%s) differs from the whole-image schedule's" % (what, field, name)
    assert np.array_equal(fill[0][2].cpu().numpy().astype(np.uint32), ref.counts), what
    assert float(np.abs(fill[0][0].cpu().numpy() - ref.color).max()) <= COLOR_TOL, what
    assert np.array_equal(fill[0][1].cpu().numpy(), ref.rgba8), what
    if not blend:
        # the production frame: RGBA8 only (the outside tiles are then cleared with 16-byte stores when the image rows allow it)
        only = [torch.full((size[1], size[0], 4), 0xC3, dtype=torch.uint8, device="cuda") for _ in range(2)]
        qs = []
        for i, t in enumerate(only):
            q = bound(pf, (None, t, None, None)) if False else sp.bind(pf)
            q.d_out_color, q.d_out_rgba8, q.d_out_counts, q.d_out_depth = None, t.data_ptr(), None, None
            q.d_in_depth, q.blend_over_target = (depth_keep[5].data_ptr() if in_depth is not None else None), 0
            qs.append(q)
        ctx.render_batch(qs, st)
        torch.cuda.synchronize()
        for t in only:
            assert torch.equal(t, whole[0][1]), "%s: RGBA8-only frame of the fill_outside schedule differs from the whole-image schedule's" % what
    # argument checks: fill_outside wants the whole rectangle, image-indexed
    bad = bound(pf, targets())
    bad.tiles.tile_count = max(0, rect.tiles - 1)
    if rect.tiles > 1 and (rect.w, rect.h) != (-(-size[0] // 16), -(-size[1] // 16)):
        assert ctx.render_rc(bad, st) == abi.VKV_E_INVALID_ARGUMENT
    del keep


RING_RECTS = [(0, 0, 1, 1), (3, 2, 1, 5), (2, 3, 7, 1), (1, 0, 2, 7), (0, 1, 9, 2), (4, 2, 3, 3), (1, 1, 5, 4), (0, 0, 10, 7), (2, 0, 8, 7), (0, 0, 10, 6),
              (3, 1, 6, 5), (5, 0, 4, 7)]


@pytest.mark.parametrize("rect", RING_RECTS, ids=lambda r: "%dx%d@%d,%d" % (r[2], r[3], r[0], r[1]))
def test_computed_start_order_reaches_every_tile_of_a_rectangle(ctx, rect):
    """A schedule that holds every tile of a tile rectangle starts its tiles ring by ring from the rectangle's middle (start_entry in
    raymarch_core.hpp: computed in the kernel, no table).  Every rectangle shape - one tile, one row, one column, two rows, odd and even sides,
    the whole image - renders exactly its tiles: inside equal to the whole-image frame, outside untouched (image-indexed) / every slot of the
    compact buffer written; the plain order (VkvTuning.tile_order_linear) gives the same bytes."""
    from tests.test_gpu_parity import fuzz_case
    scene, v, params, _, label = fuzz_case(ctx, 311)
    size = (160, 112)  # 10 x 7 tiles
    view, proj = T.orbit(35.0, image_size=size)
    sp = V.VolumeRenderSubpass(ctx, v, params.options, size)
    p = sp.make_params(view, proj, abi.full_frame_tiles(size[0], size[1]))
    st = torch.cuda.current_stream().cuda_stream
    r = abi.TileRect(*rect)

    def run(tiles, compact_pixels=None, n=2):
        outs, qs = [], []
        for _ in range(n):
            shape = (compact_pixels, 4) if compact_pixels else (size[1], size[0], 4)
            rgba8 = torch.full(shape, 0x5A, dtype=torch.uint8, device="cuda")
            counts = torch.full(shape[:-1] + (3,), 77, dtype=torch.int32, device="cuda")
            q = abi.RenderParams.from_buffer_copy(p)
            q.tiles = tiles
            q.d_out_color, q.d_out_rgba8, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = None, rgba8.data_ptr(), counts.data_ptr(), None, None, 0
            outs.append((rgba8, counts))
            qs.append(q)
        ctx.render_batch(qs, st)
        torch.cuda.synchronize()
        return outs

    whole = run(abi.full_frame_tiles(size[0], size[1]))[0]
    assert int(whole[0][..., 3].max()) > 0, label
    inside = torch.zeros((size[1], size[0]), dtype=torch.bool, device="cuda")
    inside[rect[1] * 16:(rect[1] + rect[3]) * 16, rect[0] * 16:(rect[0] + rect[2]) * 16] = True
    results = {}
    for linear in (0, 1):
        ctx.set_tuning(tile_order_linear=linear)
        try:
            img = run(abi.full_frame_tiles(size[0], size[1], rect=r))
            cmp = run(abi.full_frame_tiles(size[0], size[1], rect=r, compact=True), compact_pixels=rect[2] * rect[3] * 256)
        finally:
            ctx.set_tuning(tile_order_linear=0)
        for rgba8, counts in img:
            assert torch.equal(rgba8[inside], whole[0][inside]) and torch.equal(counts[inside], whole[1][inside]), "rect %s, linear %d" % (rect, linear)
            assert bool((rgba8[~inside] == 0x5A).all()) and bool((counts[~inside] == 77).all()), "rect %s, linear %d: wrote outside the rectangle" % (rect, linear)
        # the compact buffer: tile k of the rectangle (row-major) in slot k, every slot written
        tiles = whole[0][rect[1] * 16:(rect[1] + rect[3]) * 16, rect[0] * 16:(rect[0] + rect[2]) * 16].reshape(rect[3], 16, rect[2], 16, 4).permute(0, 2, 1, 3, 4).reshape(-1, 4)
        for rgba8, counts in cmp:
            assert torch.equal(rgba8, tiles), "rect %s, linear %d: compact buffer" % (rect, linear)
            assert bool((counts != 77).any(dim=-1).all())
        results[linear] = img[0][0]
    assert torch.equal(results[0], results[1])
