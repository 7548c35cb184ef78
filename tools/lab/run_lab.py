"""GPU A/B of experimental ray-march kernel variants (tools/lab/rm_lab.hip) against the product's vkv_render.

Every variant must reproduce the product's RGBA8, float colour, depth and the three counters bit for bit on all 8 bench views; then
it is timed one frame at a time (HIP events around single launches, median per view) and with three frames in flight.

    python tools/lab/run_lab.py [workload] [variants, comma separated] [--dense] [--stamps]
"""
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vkvolume_amd import abi, lib, volume as V  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
name = args[0] if args else "c3"
variants = [int(x) for x in args[1].split(",")] if len(args) > 1 else [0, 30, 21, 13]  # 30 = the product loop instantiated in the lab, 21 = round 3's flag set; older ids need -DLAB_ALL (tools/lab/Makefile)
dense = "--dense" in flags

torch.cuda.set_device(0)
ctx = lib.Context(0)
LAB = C.CDLL(os.path.join(ROOT, "tools", "lab", "libvkv_lab.so"))
LAB.vkv_lab_render.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_int, C.c_void_p]
L = lib.load()
L.vkv_debug_trace.argtypes = [C.c_void_p, C.c_void_p]

v, tf, frame, skip = bench.build_scene(ctx, name)
fw, fh = frame
views = bench.cameras(v, fw / fh)
if "--away" in flags:
    # every camera turned away from the volume: no ray enters it, the launch is ray set-up + pixel write-out only
    import math
    from vkvolume_amd import camera as _cam
    m = (v.node_transform.astype(np.float64).T @ v.image_transform.astype(np.float64).T)[:3, :3]
    r = 1.5 * 0.5 * math.sqrt(sum(float(np.linalg.norm(m[:, i])) ** 2 for i in range(3)))
    eyes = [r * np.array([math.cos(math.radians(20.0)) * math.sin(math.radians(45.0 * i)), math.sin(math.radians(20.0)),
                          math.cos(math.radians(20.0)) * math.cos(math.radians(45.0 * i))]) for i in range(8)]
    views = [(_cam.look_at(eyes[i], 3.0 * eyes[i]), views[i][1]) for i in range(8)]
if dense:
    opts = abi.RenderOptions(skipping_type=abi.SKIP_NONE, clip_distance=1.0, early_ray_termination=False)
else:
    opts = abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True)
sp = V.VolumeRenderSubpass(ctx, v, opts, (fw, fh))
params = [sp.make_params(view, proj) for view, proj in views]
flagword = int(v.transfer_function_bits[2048].item())
print("workload", name, "dense" if dense else "ESS+ERT", "TF separable flag:", flagword & 1, flush=True)


def bufs():
    return dict(rgba8=torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda"), color=torch.zeros((fh, fw, 4), dtype=torch.float32, device="cuda"),
                counts=torch.zeros((fh, fw, 3), dtype=torch.int32, device="cuda"), depth=torch.zeros((fh, fw), dtype=torch.float32, device="cuda"))


def set_outputs(p, b, only_rgba8=False):
    p.d_out_rgba8 = b["rgba8"].data_ptr()
    p.d_out_color = None if only_rgba8 else b["color"].data_ptr()
    p.d_out_counts = None if only_rgba8 else b["counts"].data_ptr()
    p.d_out_depth = None if only_rgba8 else b["depth"].data_ptr()
    p.d_in_depth, p.blend_over_target = None, 0


BRICK_VARIANTS = (40, 42, 43)   # read a bricked distance map (vkv_lab_brick_map); 41 / 42: probes only (the frame is wrong by design: not compared)
NOT_COMPARED = (41, 42, 44)   # 44: two of the four footprint gathers skipped
LAB.vkv_lab_brick_map.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
_me = v.map_extent
_bricked = []
for _m in v.distance_maps:
    _b = torch.zeros(64 * ((_me.width + 3) // 4) * ((_me.height + 3) // 4) * ((_me.depth + 3) // 4), dtype=torch.uint8, device="cuda")
    assert LAB.vkv_lab_brick_map(_m.data_ptr(), _b.data_ptr(), _me.width, _me.height, _me.depth, None) == 0
    _bricked.append(_b)
torch.cuda.synchronize()


def launch(variant, p, stream):
    if variant == 0:
        ctx.render(p, stream)
    else:
        if variant in BRICK_VARIANTS:
            p = abi.RenderParams.from_buffer_copy(p)
            for i, b in enumerate(_bricked):
                p.d_distance_maps[i] = b.data_ptr()
        rc = LAB.vkv_lab_render(ctx.handle, C.byref(p), variant, stream)
        if rc != 0:
            raise RuntimeError("lab variant %d: rc %d" % (variant, rc))


cur = torch.cuda.current_stream().cuda_stream
ref = []
for p in params:
    b = bufs()
    set_outputs(p, b)
    launch(0, p, cur)
    torch.cuda.synchronize()
    ref.append(b)
counts0 = ref[0]["counts"].to(torch.int64).sum((0, 1)).cpu().numpy()
print("view 0: volume samples %d, probes %d, empty %d" % tuple(counts0), flush=True)

results = {}
STREAMS = [torch.cuda.Stream() for _ in range(3)]  # the same three streams for every variant (stream -> hardware queue mapping is luck)
for var in variants:
    ok = True
    if var != 0 and var not in NOT_COMPARED:
        for i, p in enumerate(params):
            b = bufs()
            set_outputs(p, b)
            launch(var, p, cur)
            torch.cuda.synchronize()
            for k in ("rgba8", "counts", "color", "depth"):
                if not torch.equal(b[k], ref[i][k]):
                    ok = False
                    nbad = int((b[k] != ref[i][k]).sum().item())
                    print("  variant %d view %d: %s differs in %d elements" % (var, i, k, nbad), flush=True)
            del b
    # ---- one frame at a time ----
    b = bufs()
    per_view = []
    for p in params:
        set_outputs(p, b, only_rgba8=True)
        for _ in range(2):
            launch(var, p, cur)
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            launch(var, p, cur)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        per_view.append(float(np.median(ts)))
    single = float(np.mean(per_view))
    # ---- three frames in flight ----
    streams = STREAMS
    bb = [bufs() for _ in range(3)]
    pp = []
    for k in range(24):
        q = abi.RenderParams.from_buffer_copy(params[k % 8])
        set_outputs(q, bb[k % 3], only_rgba8=True)
        pp.append(q)
    def burst(n):
        for k in range(n):
            launch(var, pp[k % 24], streams[k % 3].cuda_stream)
    burst(24)
    torch.cuda.synchronize()
    reps = []
    for _ in range(7):
        t0 = time.perf_counter()
        burst(240)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / 240 * 1e3)
    fif3 = float(np.median(reps))
    results[var] = dict(bit_identical=ok, single_ms=round(single, 4), per_view_ms=[round(x, 3) for x in per_view], fif3_ms=round(fif3, 4), fif3_min_max=[round(min(reps), 4), round(max(reps), 4)])
    print(json.dumps({"variant": var, **results[var]}), flush=True)
    del bb, b

if "--stamps" in flags:
    for var in [int(x) for x in os.environ.get("LAB_STAMP_VARIANTS", "231,232,234").split(",")]:
        p = params[0]
        b = bufs()
        set_outputs(p, b, only_rgba8=True)
        W = var % 10
        nwaves = ((p.tiles.tile_count + 7) // 8) * 8 * W * 4
        trace = torch.zeros((nwaves, 10), dtype=torch.int64, device="cuda")
        for _ in range(2):
            launch(var, p, cur)
        torch.cuda.synchronize()
        L.vkv_debug_trace(ctx.handle, trace.data_ptr())
        launch(var, p, cur)
        torch.cuda.synchronize()
        L.vkv_debug_trace(ctx.handle, None)
        t = trace.cpu().numpy()
        t = t[t[:, 1] > 0]
        it = t[:, 2]
        dur_us = (t[:, 1] - t[:, 0]) / 100.0
        m = it >= 32
        ph = t[m][:, [8, 4, 5, 6, 7]].astype(np.float64) / it[m, None]
        print("stamped variant %d: waves %d (>=32 iterations: %d), kernel span %.1f us, longest wave %d iterations" % (var, len(t), m.sum(), (t[:, 1].max() - t[:, 0].min()) / 100.0, it.max()))
        print("  us per iteration (waves >= 32 it): median %.3f" % np.median(dur_us[m] / it[m]))
        print("  cycles per iteration, mean over those waves: addresses %.0f | load issue %.0f | memory wait %.0f | evaluate (filter, TF, skip length) %.0f | replay %.0f" % tuple(ph.mean(0)))
        print("  same, waves with >= 150 iterations:", (t[it >= 150][:, [8, 4, 5, 6, 7]].astype(np.float64) / it[it >= 150, None]).mean(0).round(0) if (it >= 150).any() else None)

if "--lean-stamps" in flags:
    # variant 31 = the product's single-frame kernel (raymarch_core.hpp, kLeanLut | kLeanFull) + s_memtime at the top of every iteration: cycles per
    # iteration by kind, per wave (LAB_STAMP_VARIANT=26: round 3's lab copy of the same loop)
    stamp_variant = int(os.environ.get("LAB_STAMP_VARIANT", "31"))
    for vi in range(len(params)):
        p = params[vi]
        b = bufs()
        set_outputs(p, b, only_rgba8=True)
        nwaves = ((p.tiles.tile_count + 7) // 8) * 8 * 4 * 4
        trace = torch.zeros((nwaves, 10), dtype=torch.int64, device="cuda")
        for _ in range(2):
            launch(stamp_variant, p, cur)
        torch.cuda.synchronize()
        L.vkv_debug_trace(ctx.handle, trace.data_ptr())
        launch(stamp_variant, p, cur)
        torch.cuda.synchronize()
        L.vkv_debug_trace(ctx.handle, None)
        t = trace.cpu().numpy().astype(np.uint64)
        t = t[t[:, 1] > 0]
        it = t[:, 2].astype(np.int64)
        real_us = (t[:, 1] - t[:, 0]).astype(np.float64) / 100.0
        clk = (t[:, 7] - t[:, 6]).astype(np.float64)
        long_ = it >= 32
        mhz = np.median(clk[long_] / real_us[long_])
        out = {"view": vi, "waves": int(len(t)), "longest_wave_iterations": int(it.max()), "kernel_span_us": round(float((t[:, 1].max() - t[:, 0].min()) / 100.0), 1),
               "s_memtime_MHz": round(float(mhz), 1)}
        for name, col in (("probe_only", 4), ("sample_only", 5), ("mixed", 8)):
            sums = (t[:, col] & np.uint64(0xffffffff)).astype(np.float64)
            cnts = (t[:, col] >> np.uint64(32)).astype(np.float64)
            out[name] = {"iterations": int(cnts.sum()), "ticks_per_iteration_all_waves": round(float(sums.sum() / max(cnts.sum(), 1)), 1),
                         "ticks_per_iteration_waves_ge_150": round(float(sums[it >= 150].sum() / max(cnts[it >= 150].sum(), 1)), 1) if (it >= 150).any() else None,
                         "ticks_per_iteration_longest_1pct_waves": round(float(sums[it >= np.percentile(it, 99)].sum() / max(cnts[it >= np.percentile(it, 99)].sum(), 1)), 1)}
        top = it >= np.percentile(it, 99)
        out["longest_1pct_waves"] = {"waves": int(top.sum()), "iterations_mean": round(float(it[top].mean()), 1), "ticks_per_iteration_whole_wave": round(float((clk[top] / it[top]).mean()), 1),
                                     "start_us_after_kernel_start_mean": round(float(((t[top, 0] - t[:, 0].min()).astype(np.float64) / 100.0).mean()), 1)}
        tot_s = sum((t[:, c] & np.uint64(0xffffffff)).astype(np.float64) for c in (4, 5, 8)); tot_c = sum((t[:, c] >> np.uint64(32)).astype(np.float64) for c in (4, 5, 8))
        out["check_longest_1pct"] = {"sum_of_counts_over_iterations": round(float((tot_c[top] / it[top]).mean()), 3), "sum_of_ticks_over_wave_ticks": round(float((tot_s[top] / clk[top]).mean()), 3),
                                     "kind_shares_of_iterations": [round(float(((t[top, c] >> np.uint64(32)).astype(np.float64).sum()) / tot_c[top].sum()), 3) for c in (4, 5, 8)]}
        print(json.dumps(out), flush=True)
