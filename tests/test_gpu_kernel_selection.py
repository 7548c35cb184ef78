"""GPU: every kernel-selection switch of the tuning block renders the same bits; screen-bound cases; range checks of vkv_set_tuning; the ray set-up's division dispatch."""
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
@pytest.mark.parametrize("case", ["inside", "off_centre", "zoomed", "away", "grazing_corner", "compact_rank"])
def test_render_parity_screen_bound_cases(ctx, case):
    """The launcher's screen bound of the volume's box (pixels outside skip the ray set-up) must never cut a pixel the oracle shades:
    camera inside the box (bound disabled), box partly off screen, box larger than the screen, box behind the camera, a box corner
    beside the camera plane, and a compact multi-GPU tile schedule."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((88, 72, 64), 1, 0xB0B0), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    skipping_type = abi.SKIP_DISTANCE
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (176, 96)
    from vkvolume_amd import camera
    fov, tiles = 60.0, None
    if case == "inside":
        view = camera.look_at((12.0, 5.0, 8.0), (-30.0, 0.0, -20.0))
    elif case == "off_centre":
        view = camera.look_at((150.0, 40.0, 20.0), (70.0, 10.0, -60.0))
    elif case == "zoomed":
        view, fov = camera.look_at((130.0, 30.0, 40.0), (0.0, 0.0, 0.0)), 14.0
    elif case == "away":
        view = camera.look_at((150.0, 40.0, 20.0), (400.0, 90.0, 60.0))
    elif case == "grazing_corner":
        view = camera.look_at((62.0, 10.0, 58.0), (62.0, 10.0, -100.0))  # looks along -z past the box: corners beside the camera plane
    else:
        view = camera.look_at((140.0, 50.0, -60.0), (20.0, 0.0, 10.0))
        tiles = abi.full_frame_tiles(size[0], size[1], 16, 16, 1, 3, compact=True)
    proj = camera.perspective_vulkan(fov, size[0] / size[1])
    ro = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0)
    p = scene.params(view, proj, size, ro, tiles=tiles)
    ref = scene.render(p)
    got = gpu_render(ctx, v, p)
    compare_render(got, ref, "screen bound case %s" % case)
    if case == "away":
        assert int(ref.counts.sum()) == 0
    elif case != "grazing_corner":
        assert int(ref.counts.sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["address_tables=0", "address_tables=1", "screen_cull=0", "tile_order_linear=1", "full_table_lds_limit=0", "clamp_always=1"])
def test_render_parity_under_each_kernel_selection_switch(switch):
    """The launcher picks one of three instantiations of the integrator (footprint address in registers / two-level LDS tables / one
    entry per voxel index), a tile start order and the screen bound by itself; small test volumes always get the same choice.  The
    render parity tests are re-run in a child process whose contexts start from the matching environment default (vkv_create reads
    VKV_RAYMARCH_* once per context), and the same switches are flipped through vkv_set_tuning in
    test_tuning_block_switches_render_the_same_bits."""
    env_of = {"address_tables=0": ("VKV_RAYMARCH_LUT", "0"), "address_tables=1": ("VKV_RAYMARCH_LUT", "2"), "screen_cull=0": ("VKV_RAYMARCH_CULL", "0"),
              "tile_order_linear=1": ("VKV_RAYMARCH_TILE_ORDER", "linear"), "full_table_lds_limit=0": ("VKV_RAYMARCH_FULL_LIMIT", "1"),
              # round 5: the march loop with its clamps in every iteration
              "clamp_always=1": ("VKV_RAYMARCH_CLAMP", "always")}
    name, value = env_of[switch]
    env = dict(os.environ, **{name: value})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-k", "render", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "%s:\n%s\n%s" % (switch, r.stdout[-3000:], r.stderr[-1000:])
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_tuning_block_switches_render_the_same_bits():
    """vkv_get_tuning / vkv_set_tuning: the environment is read once by vkv_create; afterwards only the block decides.  Every switch of
    the launchers renders the same frame (RGBA8 + counters) through vkv_render and vkv_render_batch."""
    c = lib.Context(0)
    try:
        t0 = c.get_tuning()
        assert t0.struct_size == C.sizeof(abi.Tuning) and t0.address_tables == 2 and t0.feedback == 1 and t0.feedback_period == 8 and t0.screen_cull == 1
        assert t0.arena_bytes >= (1 << 20)
        bad = abi.Tuning.from_buffer_copy(t0)
        bad.struct_size = 8
        assert c._lib.vkv_set_tuning(c.handle, C.byref(bad)) == abi.VKV_E_INVALID_ARGUMENT
        bad = abi.Tuning.from_buffer_copy(t0)
        bad.address_tables = 7
        assert c._lib.vkv_set_tuning(c.handle, C.byref(bad)) == abi.VKV_E_INVALID_ARGUMENT
        os.environ["VKV_RAYMARCH_LUT"] = "0"  # too late for this context: it must not change anything
        try:
            assert c.get_tuning().address_tables == 2
        finally:
            del os.environ["VKV_RAYMARCH_LUT"]
        opt = abi.VolumeOptions(**T.APP_TF)
        scene = T.OracleScene(O.synth_volume((72, 64, 56), 1, 4242), opt, 4)
        v, tf = make_gpu_volume(c, scene)
        V.ComputeDistanceMap(c).compute(v, tf, abi.SKIP_DISTANCE)
        size = (208, 112)
        ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        sp = V.VolumeRenderSubpass(c, v, ro, size)
        st = torch.cuda.current_stream().cuda_stream
        views = [sp.bind(scene.params(*T.orbit(az, image_size=size), size, ro)) for az in (15.0, 140.0, 260.0)]

        def frames():
            out = []
            for p in views:
                rgba, cnt = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda"), torch.zeros((size[1], size[0], 3), dtype=torch.int32, device="cuda")
                q = abi.RenderParams.from_buffer_copy(p)
                q.d_out_rgba8, q.d_out_counts, q.d_out_color, q.d_out_depth = rgba.data_ptr(), cnt.data_ptr(), None, None
                c.render(q, st)
                out.append((rgba, cnt))
            plist, outs = [], []
            for p in views:
                rgba, cnt = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda"), torch.zeros((size[1], size[0], 3), dtype=torch.int32, device="cuda")
                q = abi.RenderParams.from_buffer_copy(p)
                q.d_out_rgba8, q.d_out_counts, q.d_out_color, q.d_out_depth = rgba.data_ptr(), cnt.data_ptr(), None, None
                plist.append(q)
                outs.append((rgba, cnt))
            c.render_batch(plist, st)
            torch.cuda.synchronize()
            return out + outs

        ref = frames()
        assert int(ref[0][1].sum().item()) > 0
        for fields in (dict(address_tables=0), dict(address_tables=1), dict(full_table_lds_limit=0), dict(screen_cull=0), dict(tile_order_linear=1),
                       dict(batch_sequential=1), dict(batch_mode=1), dict(scheduler=1), dict(feedback=0), dict(tile_mix_heavy=0.3, tile_mix_spread=0.6),
                       dict(clamp_always=1), dict(occupancy_kernel=1), dict(wave_shape=4), dict(wave_shape=8), dict(wave_shape=16)):
            c.set_tuning(**fields)
            got = frames()
            reset = {k: getattr(t0, k) for k in fields}
            c.set_tuning(**reset)
            for i, ((a, b), (x, y)) in enumerate(zip(ref, got)):
                assert torch.equal(a, x) and torch.equal(b, y), "%r changes frame %d" % (fields, i)
    finally:
        c.close()


def test_zero_numerators_take_the_ieee_division(ctx):
    """the fast division of the ray set-up (v_rcp_f32 + refinement) loses the sign of -0 / d, so a zero numerator must not be
    'ordinary' (div_ordinary_num) - vkv_debug_check what = 4 runs the dispatch for +0 and -0 over EVERY float as the denominator: the fast
    path is never taken and the quotient is the IEEE one bit for bit."""
    import ctypes as C
    L = ctx._lib
    L.vkv_debug_check.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    for first in (0, 0x80000000):
        ctx.check(L.vkv_debug_check(ctx.handle, 4, first, 1 << 31, bad.data_ptr(), None))
    torch.cuda.synchronize()
    assert bad.tolist() == [0]


def test_set_tuning_rejects_values_that_would_break_every_launch(ctx):
    """a tile_mix that is not a number would never match a cached schedule (a new table per launch); a
    full_table_lds_limit above what a kernel may request as dynamic LDS would fail every launch instead of choosing smaller tables."""
    with pytest.raises(Exception):
        ctx.set_tuning(tile_mix_heavy=float("nan"))
    with pytest.raises(Exception):
        ctx.set_tuning(tile_mix_spread=1.5)
    try:
        ctx.set_tuning(full_table_lds_limit=1 << 30)
        assert ctx.get_tuning().full_table_lds_limit <= 64 * 1024
        opt = abi.VolumeOptions(**T.APP_TF)
        scene = T.OracleScene(O.synth_volume((40, 40, 40), 1, 0x5EED0007), opt, 4)
        v, tf = make_gpu_volume(ctx, scene)
        V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
        size = (96, 64)
        ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
        view, proj = T.orbit(10.0, image_size=size)
        params = scene.params(view, proj, size, ro)
        color, counts, _, _ = gpu_render(ctx, v, params)
        ref = scene.render(params)
        assert np.array_equal(counts, ref.counts) and np.abs(color - ref.color).max() <= 1e-5
    finally:
        ctx.set_tuning(full_table_lds_limit=17920)
