"""GPU: the render entry points only enqueue - hipMemGetInfo before and after launches, the arena's table region filling up, vkv_trim, a fresh process."""
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


def test_prepare_render_then_launches_take_nothing_new(ctx):
    """vkv_prepare_render (set-up) creates the stream's scratch block, the address tables and the tile start order; the launches after it
    find everything in place: the arena's fill level (read back through a second prepare call's idempotence and the device's free
    memory) does not move, on a new stream or on the old one, and the frames equal those of an unprepared context."""
    scene = T.OracleScene(O.synth_volume((80, 72, 64), 1, 515), abi.VolumeOptions(**T.APP_TF), 4)
    size = (208, 112)
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    results = []
    for prepared in (False, True):
        c = lib.Context(0)
        try:
            v, tf = make_gpu_volume(c, scene)
            V.ComputeDistanceMap(c).compute(v, tf, abi.SKIP_DISTANCE)
            sp = V.VolumeRenderSubpass(c, v, ro, size)
            streams = [torch.cuda.Stream() for _ in range(3)]
            targets = [torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda") for _ in range(6)]
            plist = []
            for j, az in enumerate((0.0, 60.0, 120.0, 180.0, 240.0, 300.0)):
                p = sp.make_params(*T.orbit(az, image_size=size))
                p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = targets[j].data_ptr(), None, None, None
                plist.append(p)
            torch.cuda.synchronize()
            if prepared:
                for s in streams:
                    c.prepare_render(plist, s.cuda_stream)
                for t in targets:
                    c.register_target(t.data_ptr(), size, plist[0].tiles)
                torch.cuda.synchronize()
            free0, _ = torch.cuda.mem_get_info()
            for rep in range(3):
                for k, s in enumerate(streams):
                    c.render_batch(plist[2 * k:2 * k + 2], s.cuda_stream)
                    c.render(plist[2 * k], s.cuda_stream)
            torch.cuda.synchronize()
            free1, _ = torch.cuda.mem_get_info()
            if prepared:
                assert free1 == free0, "launches after vkv_prepare_render / vkv_register_target must not allocate device memory (%d bytes)" % (free0 - free1)
            results.append([t.clone() for t in targets])
            for s in streams:
                c.release_stream(s.cuda_stream)
            for t in targets:
                c.forget_target(t.data_ptr())
        finally:
            c.close()
    for a, b in zip(*results):
        assert int(a.sum().item()) > 0 and torch.equal(a, b)


def test_arena_exhaustion_degrades_to_table_free_launches(monkeypatch):
    """A context whose arena has no room for a table still renders the same bits: the launch runs without its start order instead of
    allocating, and a batch launch on a stream that cannot get a scratch block says so (VKV_ARENA_BYTES is read by vkv_create; the
    minimum is 1 MiB: four 128 KiB scratch blocks + 512 KiB of tables - round 4 gave the two their own regions)."""
    scene = T.OracleScene(O.synth_volume((72, 64, 56), 1, 99), abi.VolumeOptions(**T.APP_TF), 4)
    size = (6144, 6144)        # 147 456 tiles: a start order of 576 KiB, more than the small arena's table region
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0)
    frames = []
    for arena in (None, "1048576"):
        if arena:
            monkeypatch.setenv("VKV_ARENA_BYTES", arena)
        c = lib.Context(0)
        monkeypatch.delenv("VKV_ARENA_BYTES", raising=False)
        try:
            assert c.get_tuning().arena_bytes == (int(arena) if arena else 8 << 20)
            v, tf = make_gpu_volume(c, scene)
            V.ComputeDistanceMap(c).compute(v, tf, abi.SKIP_DISTANCE)        # the default stream's scratch block
            count = torch.zeros(1, dtype=torch.int64, device="cuda")
            for s_ in [torch.cuda.Stream() for _ in range(3)]:        # three more: the four scratch blocks of the small arena are gone
                c.occupied_voxel_count(v.volume.data_ptr(), v.gradient.data_ptr(), tf, v.extent, count.data_ptr(), s_.cuda_stream)
            torch.cuda.synchronize()
            sp = V.VolumeRenderSubpass(c, v, ro, size)
            t = torch.zeros((size[1], size[0], 4), dtype=torch.uint8, device="cuda")
            p = sp.make_params(*T.orbit(40.0, image_size=size))
            p.d_out_rgba8, p.d_out_color, p.d_out_counts, p.d_out_depth = t.data_ptr(), None, None, None
            st = torch.cuda.Stream()
            c.render(p, st.cuda_stream)        # small arena: no room for the 576 KiB start order - plain tile order, same frame
            torch.cuda.synchronize()
            frames.append(t)
            if arena:
                # a batch launch needs a scratch block for its argument blocks: none left on a new stream - the call says so instead of allocating
                q = abi.RenderParams.from_buffer_copy(p)
                rc = c._lib.vkv_render_batch(c.handle, (abi.RenderParams * 2)(p, q), 2, torch.cuda.Stream().cuda_stream)
                assert rc == abi.VKV_E_UNSUPPORTED and "arena" in c.last_error()
                # ... while the set-up call may allocate: afterwards the same launch works
                st2 = torch.cuda.Stream()
                c.prepare_render([p, q], st2.cuda_stream)
                t.zero_()
                c.render_batch([p, q], st2.cuda_stream)
                torch.cuda.synchronize()
                assert torch.equal(t, frames[0])
        finally:
            c.close()
    assert int(frames[0].to(torch.int64).sum().item()) > 0 and torch.equal(frames[0], frames[1])


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
