"""Round 4 GPU tests (through the C ABI)."""
import os

import numpy as np
import pytest
import torch

from oracle import vkv_oracle as O
from tests import helpers as T
from tests.test_gpu_parity import gpu_render, make_gpu_volume
from vkvolume_amd import abi, volume as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def test_table_region_fills_up_falls_back_and_trims(ctx):
    """tables are never evicted, so a renderer that keeps meeting new window sizes fills the arena's table region.
    300 distinct frame sizes: every frame is still the oracle-checked frame (once the region is full launches run without the start-order
    table: same bits), a NEW stream still gets its scratch block (its region is separate: vkv_render_batch works), vkv_trim empties the
    region and tables are created again; nothing of this allocates device memory (hipMemGetInfo)."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((48, 40, 36), 1, 0x5EED0006), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_DISTANCE)
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    st = torch.cuda.current_stream().cuda_stream
    big = (1920, 1088)  # 8 160 tiles = 32 KiB per table: the 6 MiB region holds ~190 of them
    rgba8 = torch.zeros((big[1] + 300, big[0], 4), dtype=torch.uint8, device="cuda")
    view, proj = T.orbit(40.0, image_size=(256, 160))
    small = (256, 160)
    ref = scene.render(scene.params(view, proj, small, ro), want_rgba8=True)

    def draw(size, stream=st):
        sp = V.VolumeRenderSubpass(ctx, v, ro, size)
        p = sp.bind(scene.params(view, proj, size, ro))
        q = abi.RenderParams.from_buffer_copy(p)
        q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth = rgba8.data_ptr(), None, None, None
        ctx.render(q, stream)
        return q

    draw(small)
    torch.cuda.synchronize()
    assert np.array_equal(rgba8.view(-1)[:small[0] * small[1] * 4].view(small[1], small[0], 4).cpu().numpy(), ref.rgba8)
    s2 = torch.cuda.Stream()  # created AND used before the reading below: a stream's first work makes the RUNTIME allocate its queue
    with torch.cuda.stream(s2):
        rgba8[:1].fill_(0)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(300):  # 300 x 32 KiB > the region: the later ones run without a start-order table
        draw((big[0], big[1] + i))
    torch.cuda.synchronize()
    # a stream the context has never seen still gets a scratch block and renders a batch
    q = draw(small, s2.cuda_stream)
    ctx.render_batch([q, q], s2.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(rgba8.view(-1)[:small[0] * small[1] * 4].view(small[1], small[0], 4).cpu().numpy(), ref.rgba8)
    ctx.trim()
    draw(small)
    draw((big[0], big[1] + 7))
    draw(small)
    torch.cuda.synchronize()
    assert np.array_equal(rgba8.view(-1)[:small[0] * small[1] * 4].view(small[1], small[0], 4).cpu().numpy(), ref.rgba8)
    assert torch.cuda.mem_get_info()[0] >= free0, "launches, the new stream or vkv_trim took device memory"
    ctx.release_stream(s2.cuda_stream)


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


def test_bench_two_ranks_share_one_device_over_gloo():
    """The N > 1 orchestration of bench.py with TWO REAL RANKS (the self-launching parent, torch.distributed.run, rank != owner paths, the
    rotating owner, buffer-set reuse behind the exchange's events, the max-over-ranks timing, --verify on the launch's owner) on a box with
    one GPU: both ranks use device 0 and the tile blocks travel over gloo through host memory (RCCL refuses two ranks on one device; what
    stays untested without a second GPU is RCCL moving the bytes).  The assembled frame of the last step must equal a direct render."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    for steps, fpl in ((12, 8), (7, 3)):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device", "--workload", "small", "--steps", str(steps),
               "--warmup", "3", "--frames-per-launch", str(fpl), "--min-seconds", "0.2", "--verify", "--c5-block", "off", "--no-cpu-baseline", "--launch-timeout", "600"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == steps and d["scaling"] == "weak"
        assert "gloo" in d["rccl_ranks_source"].lower() and "GLOO" in d["config"]["parallelism"]
        assert d["phases"]["launches_sampled"] > 0 and d["phases"]["render_ms"] > 0 and "gather_ms" in d["phases"] and "scatter_ms" in d["phases"]
        assert d["value"] > 0 and d["roofline"]["frac"] > 0
        assert "verify ok" in r.stderr, r.stderr[-2000:]


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



def test_first_launches_of_a_process_after_set_up_take_no_device_memory():
    """The runtime loads a translation unit's code object (device memory, milliseconds) at the first use of one of its kernels.
    vkv_prepare_render and vkv_register_target do that for the kernels the parameter blocks will launch, so the very first vkv_render /
    vkv_render_batch of a PROCESS neither allocates nor stalls on a load: checked in a fresh interpreter (in this one the kernels are
    long loaded) with hipMemGetInfo around the launches."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from oracle import vkv_oracle as O
from tests import helpers as T
from vkvolume_amd import abi, lib, volume as V
c = lib.Context(0)
scene = T.OracleScene(O.synth_volume((64, 56, 48), 1, 77), abi.VolumeOptions(**T.APP_TF), 4)
v = V.Volume(c); v.options = scene.options
v.load_from_array(scene.vol, scene.block, scene.image_transform); v.node_transform = scene.node_transform
tf = v.get_transfer_function_uniform()
V.ComputeGradientMap(c).compute(v, tf); v.update_transfer_function_texture()
V.ComputeDistanceMap(c).compute(v, tf, abi.SKIP_ANISOTROPIC_DISTANCE)
size = (208, 112)
ro = abi.RenderOptions(skipping_type=abi.SKIP_ANISOTROPIC_DISTANCE, clip_distance=1.0)
sp = V.VolumeRenderSubpass(c, v, ro, size)
s = torch.cuda.Stream()
targets = [torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
with torch.cuda.stream(s):
    targets[0][:1].fill_(0)  # the stream's queue exists
plist = []
for j, az in enumerate((10.0, 70.0)):
    p = sp.make_params(*T.orbit(az, image_size=size))
    p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = targets[j].data_ptr(), None, None, None
    plist.append(p)
c.prepare_render(plist, s.cuda_stream)
for t in targets:
    c.register_target(t.data_ptr(), size, plist[0].tiles)
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
c.render(plist[0], s.cuda_stream)
c.render_batch(plist, s.cuda_stream)
c.render_batch(plist, s.cuda_stream)
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
ref = scene.render(plist[1], want_rgba8=True)
assert np.array_equal(targets[1].cpu().numpy(), ref.rgba8)
print("device memory taken by the first launches:", free0 - free1)
assert free1 >= free0
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]


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
    tiles_x, tiles_y = -(-size[0] // tw), -(-size[1] // th)
    per_rank = -(-(tiles_x * tiles_y) // world)
    n = per_rank * tw * th
    st = torch.cuda.current_stream().cuda_stream
    gathered = torch.full((world, frames, n, 4), 0x5A, dtype=torch.uint8, device="cuda")        # as the launch's owner receives it
    what = "seed %d: frame %s tiles %dx%d world %d frames %d mode %d" % (seed, size, tw, th, world, frames, st_mode)
    for r in range(world):
        sched = abi.full_frame_tiles(size[0], size[1], tw, th, r, world, compact=True)
        if sched.tile_count == 0:
            continue        # more ranks than tiles: this one has nothing to render (bench.py's ranks pass such schedules too)
        plist, counts = [], []
        for f, (view, proj) in enumerate(views):
            p = T_bind(ctx, v, scene, view, proj, size, ro, sched)
            c = torch.zeros((sched.tile_count * tw * th, 3), dtype=torch.int32, device="cuda")  # (slots beyond the image edge stay unwritten)
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = gathered[r, f].data_ptr(), None, c.data_ptr(), None
            plist.append(p)
            counts.append(c)
        if frames == 1:
            ctx.render(plist[0], st)
        else:
            ctx.render_batch(plist, st)
        torch.cuda.synchronize()
        for f, (view, proj) in enumerate(views):
            ref = scene.render(scene.params(view, proj, size, ro, tiles=sched))
            assert np.array_equal(counts[f].cpu().numpy().astype(np.uint32).reshape(ref.counts.shape), ref.counts), what + ", rank %d frame %d counters" % (r, f)
    for f, (view, proj) in enumerate(views):
        image = torch.full((size[1], size[0], 4), 3, dtype=torch.uint8, device="cuda")
        ctx.scatter_tiles(gathered.data_ptr() + f * n * 4, image.data_ptr(), size, (tw, th), world, frames * per_rank, 4, st)
        torch.cuda.synchronize()
        full = scene.render(scene.params(view, proj, size, ro, tiles=abi.full_frame_tiles(size[0], size[1], tw, th)), want_rgba8=True)
        assert np.array_equal(image.cpu().numpy(), full.rgba8), what + ", assembled frame %d" % f


def T_bind(ctx, v, scene, view, proj, size, ro, sched):
    sp = V.VolumeRenderSubpass(ctx, v, ro, size)
    return sp.bind(scene.params(view, proj, size, ro, tiles=sched))


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
