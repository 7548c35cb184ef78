#!/usr/bin/env python3
"""bench.py — Mray/s of the Chebyshev-ESS ray-march on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One *step* = one frame: every ray of the frame marched through the synthetic uint8 volume of the workload (default "c3":
1024x1024x795, BASELINE.json configs[2]) with the app-default transfer function, block size 4, Chebyshev distance-map
empty-space skipping and early ray termination, from one of 8 fixed orbit cameras (step k uses view k mod 8).  Inputs (volume,
gradient map, TF texture, distance map, packed sampling image) are resident in HBM before the timed region.

The timed block is EXACTLY K steps between two fences (barrier + synchronize); the block is repeated until at least
--min-seconds have been measured and the MEDIAN block is reported (`repeats`, `ms_per_step_min_max`).

N = 1 submits the frames `--frames-per-launch` at a time through vkv_render_batch (default 8: the eight views of one orbit in one
launch, the frames advance side by side inside one grid; a block's frames are spread evenly over its launches) and alternates
consecutive launches over `--batch-streams` HIP streams (default 3: the launches of a block overlap, the tail of one is covered by
the others; streams that happen to share a hardware queue simply serialise, which is the one-stream behaviour);
`--batch-streams 1` keeps one launch at a time; `--frames-per-launch 1` renders strictly one frame per launch, and
`--submit streams` keeps `--frames-in-flight` single-frame launches in flight on as many HIP streams (the round 1 scheme).
The one-frame-at-a-time launch duration is always measured too (`single_frame`, outside the timed region).

N > 1: the tiles of the frame's screen rectangle (the 16x16 tiles the clipped box projects into: vkv_screen_tile_rect, derived by every
rank from the uniforms; --tile-rect off = every tile of the frame) are dealt round-robin to the ranks (volume replicated); each rank renders
its tiles into a compact RGBA8 buffer, the buffers are gathered over RCCL to the owner of the launch (rank l mod N for launch l, so that
consecutive launches use disjoint xGMI links; --frame-owner spread: frame j of launch l to rank (l + j) mod N; rank0 pins it) and
de-interleaved there (the rest of the image is cleared).  `--scaling weak` (default for c2/c3/c4):
the frame grows to (W*sx)x(H*sy), sx*sy = N, sampling the SAME frustum; `--scaling strong` (default for c5, BASELINE.json
configs[4]): the frame stays 7680x4320 whatever N is.  value = rays of all ranks / max-over-ranks wall time.

Rank 0 prints ONE JSON line.  `roofline` prices the ray-march kernel by ALGORITHMIC bytes (16 B per volume sample,
1 B per distance probe, 4 B per ray of RGBA8 output) of one launch over that launch's HIP-event duration; `cpu_baseline` times the
CPU oracle (a scalar port of the reference shaders) on a pixel-strided sample of the same frames (at most three passes), and the
pixels it rendered (three counters + RGBA8) are then compared with the device's frames, bit for bit: `verified_against_cpu` in the
line, exit status 1 on a mismatch (`--no-verify-cpu` skips the comparison).
"""
import argparse
import json
import math
import os
import sys
import time



def self_launch(argv, child_cmd=None):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the PARENT of the job.  It starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` as a
    child BEFORE anything here has touched the GPU (no exec from a process that has initialised HIP: it never initialises it), relays
    the one JSON line rank 0 prints and exits with the children's status.  `--dry-launch` prints the child command line instead.
    `--launch-timeout S` (default 900): if the ranks have not finished after S seconds (an RCCL bring-up that hangs, a rank that died
    while the others wait in a collective) the parent kills the process group it started - fresh processes in a session of their own,
    never a re-exec - and exits with status 124.  `child_cmd` replaces the command (tests)."""
    import signal
    import socket
    import subprocess
    import threading
    n, dry, timeout = 1, False, 900.0
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
        elif a == "--dry-launch":
            dry = True
        elif a == "--launch-timeout" and i + 1 < len(argv):
            timeout = float(argv[i + 1])
        elif a.startswith("--launch-timeout="):
            timeout = float(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return  # a rank of an existing job (or N = 1): run in this process
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    child_args = [a for a in argv if a != "--dry-launch"]
    cmd = child_cmd or ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", port,
                         os.path.abspath(__file__)] + child_args)
    if dry:
        print(json.dumps({"dry_launch": cmd, "n_ranks": n, "launch_timeout_s": timeout}))
        raise SystemExit(0)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)  # own process group: killable as a whole
    timed_out = []

    def kill_job():
        timed_out.append(True)
        try:
            os.killpg(proc.pid, signal.SIGKILL)  # exactly the group this parent started
        except OSError:
            pass

    watchdog = threading.Timer(timeout, kill_job) if timeout > 0 else None
    if watchdog:
        watchdog.daemon = True
        watchdog.start()
    line = None
    for out_line in proc.stdout:  # rank 0 prints exactly one JSON line; anything else a child writes to stdout goes to stderr
        t = out_line.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)
    rc = proc.wait()
    if watchdog:
        watchdog.cancel()
    if timed_out:
        print("bench.py: the ranks did not finish within --launch-timeout %.0f s: killed" % timeout, file=sys.stderr)
        raise SystemExit(124)
    if rc != 0:
        raise SystemExit(rc if rc > 0 else 1)
    if line is None:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        raise SystemExit(1)
    print(line, flush=True)
    raise SystemExit(0)


if __name__ == "__main__":
    self_launch(sys.argv[1:])

# ROCclr multiplexes HIP streams onto 4 hardware queues by default; frames in flight, the assembly stream and RCCL's own
# stream need one each or they serialise behind each other (measured: 0.27 -> 0.20 ms per step on the gather path).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from vkvolume_amd import abi, camera, lib, multigpu, volume as V  # noqa: E402

def traffic_file(workload):
    """PMC-measured HBM bytes per launch (own rocprofv3 passes, tools/collect_profiles.sh): one file per workload that has been profiled"""
    return os.path.join(ROOT, "profiles", "r6_traffic.json" if workload == "c3" else "r6_%s_traffic.json" % workload)


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (extent WxHxD, seed, voxel size, axis-angle, frame, skipping type, scaling at N > 1)
    "c3": ((1024, 1024, 795), 0xC0FFEE03, (0.0003, 0.0003, 0.0007), (1.0, 0.0, 0.0, 90.0), (1920, 1080), abi.SKIP_DISTANCE, "weak"),
    "c3cube": ((1024, 1024, 1024), 0xC0FFEE03, (1.0, 1.0, 1.0), (1.0, 0.0, 0.0, 0.0), (1920, 1080), abi.SKIP_DISTANCE, "weak"),
    "c2": ((512, 512, 512), 0xC0FFEE02, (1.0, 1.0, 1.0), (1.0, 0.0, 0.0, 0.0), (1920, 1080), abi.SKIP_BLOCK, "weak"),
    "c4": ((2048, 2048, 2048), 0xC0FFEE04, (1.0, 1.0, 1.0), (1.0, 0.0, 0.0, 0.0), (3840, 2160), abi.SKIP_ANISOTROPIC_DISTANCE, "weak"),
    "c5": ((2048, 2048, 2048), 0xC0FFEE04, (1.0, 1.0, 1.0), (1.0, 0.0, 0.0, 0.0), (7680, 4320), abi.SKIP_ANISOTROPIC_DISTANCE, "strong"),
    "odd": ((493, 493, 443), 0xC0FFEE05, (1.0, 1.0, 1.0), (1.0, 0.0, 0.0, 0.0), (1920, 1080), abi.SKIP_DISTANCE, "weak"),
    "even": ((492, 492, 442), 0xC0FFEE05, (1.0, 1.0, 1.0), (1.0, 0.0, 0.0, 0.0), (1920, 1080), abi.SKIP_DISTANCE, "weak"),
    "small": ((128, 128, 100), 0xC0FFEE03, (0.0003, 0.0003, 0.0007), (1.0, 0.0, 0.0, 90.0), (320, 192), abi.SKIP_DISTANCE, "weak"),
}
WORKLOAD_NOTE = {
    "c3": "BASELINE.json configs[2] (stag-beetle shape)", "c3cube": "the literal 1024^3 of BASELINE.json's metric line",
    "c2": "BASELINE.json configs[1]", "c4": "BASELINE.json configs[3]", "c5": "BASELINE.json configs[4]: fixed 7680x4320 frame, screen tiles over the ranks",
    "small": "smoke size", "odd": "no extent a multiple of 4: rows at every byte alignment (tools/time_precompute.py odd; the byte-wise fallback kernels until round 5)",
    "even": "the odd workload's neighbour with every extent a multiple of 2 and the width one of 4 (tools/time_precompute.py even)",
}
GRID = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}
TILE = 16
N_VIEWS = 8
B_OUT = 4  # bytes per ray of the RGBA8 frame


def kernel_source_digest():
    """sha256 over the integrator's source files with comments and whitespace removed: ties profiles/*_traffic.json to the kernel CODE it
    was measured on (a comment edit does not invalidate a measurement)"""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in ("raymarch_core.hpp", "raymarch_inst.hpp", "raymarch.hip", "vkv_device.hpp", "Makefile"):
        with open(os.path.join(ROOT, "vkvolume_amd", "csrc", name), "r", errors="replace") as f:
            text = f.read()
        if not name.endswith("Makefile"):
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()


def build_scene(ctx, name, tf_preset="app", empty=False):
    extent, seed, voxel, axis_angle, frame, skip = WORKLOADS[name][:6]
    v = V.Volume(ctx)
    if tf_preset == "intensity":  # the reference CSVs' intensity-only rows (gmin = gmax = 0: no gradient term, 8 B per volume sample)
        v.options = abi.VolumeOptions(intensity_min=0.1, intensity_max=1.0, gradient_min=0.0, gradient_max=0.0)
    else:
        v.options = abi.VolumeOptions(intensity_min=0.1, intensity_max=1.0, gradient_min=0.0, gradient_max=0.2)  # volume_render.cpp:67-70
    v.load_synthetic(extent, kind=1, seed=seed, distance_map_block_size=4)
    if empty:  # the probe-only asymptote: every voxel below the transfer function's threshold, every map cell empty
        v.volume.zero_()
    V.default_scene(v, voxel, axis_angle)
    tf = v.get_transfer_function_uniform()
    V.ComputeGradientMap(ctx).compute(v, tf)
    v.update_transfer_function_texture()
    V.ComputeDistanceMap(ctx).compute(v, tf, skip)
    torch.cuda.synchronize()
    return v, tf, frame, skip


def split_frames(n_steps, frames_per_launch):
    """Frames of a block of n_steps spread evenly over ceil(n_steps / frames_per_launch) launches (20 steps, 6 per launch: 5 + 5 + 5 + 5; 8 per launch: 7 + 7 + 6)."""
    n_launches = -(-n_steps // frames_per_launch)
    out, k = [], 0
    for launch in range(n_launches):
        n = (n_steps - k + (n_launches - launch) - 1) // (n_launches - launch)
        out.append(n)
        k += n
    return out


def occupied_voxel_percent(ctx, v, tf):
    """the reference's benchmark-mode statistic (src/volume_render.cpp:399-414): % voxels with analytic TF alpha > 0"""
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    ctx.occupied_voxel_count(v.volume.data_ptr(), v.gradient.data_ptr(), tf, v.extent, count.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return 100.0 * float(count.item()) / float(v.extent.count)


def cameras(v, aspect, moving=0, step=1.0):
    """8 azimuths, elevation 20 deg, radius 1.5 x bounding-sphere radius of the scaled volume."""
    m = (v.node_transform.astype(np.float64).T @ v.image_transform.astype(np.float64).T)[:3, :3]
    half_diag = 0.5 * math.sqrt(sum(float(np.linalg.norm(m[:, i])) ** 2 for i in range(3)))
    radius = 1.5 * half_diag
    proj = camera.perspective_vulkan(60.0, aspect, 0.1, 1000.0)
    if moving:
        # a camera that moves: `moving` views one degree of orbit apart (a renderer's consecutive frames)
        return [(camera.orbit_camera(step * i, 20.0, radius), proj) for i in range(moving)]
    return [(camera.orbit_camera(45.0 * i, 20.0, radius), proj) for i in range(N_VIEWS)]


def main():
    # rank 0 must print exactly ONE line on stdout, but RCCL writes a version banner there when it creates a communicator: keep the real
    # stdout for the JSON line and send everything else that is written to file descriptor 1 to stderr
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--skip", default=None, choices=["none", "block", "distance", "anisotropic"], help="override the workload's empty-space-skipping mode")
    ap.add_argument("--tf", default="app", choices=["app", "intensity"], help="transfer function: the application's default (intensity 0.1..1, gradient "
                    "0..0.2) or the intensity-only rows of the reference CSVs (gmin = gmax = 0)")
    ap.add_argument("--no-ert", action="store_true", help="early ray termination off (with --skip none: dense sampling of every step of every ray)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="N > 1: grow the frame with N (weak) or keep it (strong); default per workload")
    ap.add_argument("--submit", default=None, choices=["batch", "streams"], help="N = 1: vkv_render_batch launches (default) or single-frame launches "
                    "on --frames-in-flight streams; N > 1 always uses streams")
    ap.add_argument("--batch-streams", type=int, default=4, help="batch submission: consecutive vkv_render_batch launches alternate over this many HIP streams "
                    "(the tail of one launch overlaps the head of the next)")
    ap.add_argument("--frames-per-launch", type=int, default=6, help="batch submission: frames per vkv_render_batch launch (1 = one frame per launch; 6 on 4 streams "
                    "measures 1 - 4 % better than 8 on 3 on every workload: profiles/r5_submission_sweep.txt)")
    ap.add_argument("--frames-in-flight", type=int, default=3, help="stream submission: consecutive frames render on this many HIP streams")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the timed block of K steps until this much time has been measured")
    ap.add_argument("--frame-owner", default="rotate", choices=["rotate", "spread", "rank0"], help="N > 1: rank that assembles a launch's frames: launch l on rank l mod N "
                    "(rotate, default: the inbound xGMI links and the de-interleave of consecutive launches are disjoint), frame j of launch l on rank (l + j) mod N "
                    "(spread: every GPU receives at once; batch submission), or always rank 0")
    ap.add_argument("--camera", default="static", choices=["static", "moving"], help="static (default): 8 orbit views 45 degrees apart, each render target showing one of "
                    "them (the reference's benchmark also holds its camera still, src/volume_render.cpp:224-234); moving: every frame a new view, one degree of "
                    "orbit from the last, the targets taking the frames in turn (what a renderer's swap chain sees)")
    ap.add_argument("--camera-step", type=float, default=1.0, help="--camera moving: degrees of orbit between two frames")
    ap.add_argument("--no-feedback", action="store_true", help="VkvTuning.feedback = 0: every frame starts its tiles centre-first, no start order from measured tile costs")
    ap.add_argument("--tile-rect", default="on", choices=["on", "off"], help="N > 1 / --virtual-rank / --force-gather: schedule and exchange only the tiles of each "
                    "frame's screen rectangle (vkv_screen_tile_rect: the clipped box's projection, derived by every rank from the uniforms; default) or every "
                    "tile of the frame (off: rounds 1-5)")
    ap.add_argument("--rect-align", type=int, default=1, help="tile rectangles rounded outwards to multiples of this many tiles (vkv_screen_tile_rect's align_tiles): "
                    "a camera that moves then changes the rectangle - and the start-order feedback state keyed by it - less often")
    ap.add_argument("--exchange", default="torch", choices=["native", "torch"], help="N > 1: the tile gather through torch.distributed.gather (default: "
                    "0.170 ms per step in the one-rank pipeline test) or through the C ABI (vkv_gather_tiles / vkv_scatter_tiles: ncclGather on a "
                    "communicator of our own, no torch.distributed on the data path; 0.190 ms per step in the same test)")
    ap.add_argument("--force-gather", action="store_true", help="exercise the tile gather / de-interleave path with a 1-rank process group")
    ap.add_argument("--verify", action="store_true", help="after timing, check the assembled frame of the last step against a direct render")
    ap.add_argument("--verify-cpu", action="store_true", help="(default since round 3; kept for old command lines) compare the CPU oracle's pixels with the device's")
    ap.add_argument("--no-verify-cpu", action="store_true", help="do not compare the pixels the CPU baseline rendered (counters + RGBA8) with the device's frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-depth-block", action="store_true", help="skip the extra block that also writes gl_FragDepth (ms_per_step_with_depth): the rocprofv3 "
                    "passes of tools/collect_profiles.sh use it so that the kernel's averages hold the headline launches only")
    ap.add_argument("--dry-launch", action="store_true", help="--gpus N > 1 without WORLD_SIZE: print the torch.distributed.run command line the "
                    "parent would start, and exit")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the ranks: nccl (= RCCL on ROCm, default) or gloo "
                    "(functional test of the N > 1 orchestration without RCCL: the tile blocks travel through host memory)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses device 0 (functional test of the N > 1 path on a one-GPU box; needs "
                    "--backend gloo and --exchange torch: RCCL refuses two ranks on one device)")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="--gpus N > 1 without WORLD_SIZE: seconds after which the parent kills the ranks it "
                    "started and exits with status 124 (0 = never)")
    ap.add_argument("--c5-block", default="auto", choices=["auto", "on", "off"], help="after the headline blocks, a short block of BASELINE.json configs[4] (c5: "
                    "2048^3, anisotropic maps, the fixed 7680x4320 frame dealt over the ranks), reported as \"c5_strong\" in the same line; auto = at N > 1")
    ap.add_argument("--virtual-rank", default=None, metavar="R/N", help="N = 1: render only the tile share rank R of N would render (compact schedule tile_first = R, "
                    "tile_stride = N, no exchange) - what one of N GPUs would be busy with per frame; tools/virtual_ranks.py turns the shares into a "
                    "PREDICTED scaling curve")
    ap.add_argument("--extras", default="auto", choices=["auto", "on", "off"], help="N = 1, after the headline blocks: a short block of the literal 1024^3 (\"c3cube\") and "
                    "the two asymptotes of the integrator - dense sampling and probes only (\"asymptotes\"); auto = with --workload c3 and the default options")
    ap.add_argument("--native-block", default="auto", choices=["auto", "on", "off"], help="N > 1 with --exchange torch: after each measurement a short block of the "
                    "same launches with the C ABI's exchange (vkv_assemble_frames), reported as \"native_exchange\"; a failure becomes an \"error\" string there")
    ap.add_argument("--native-timeout", type=float, default=120.0, help="seconds after which a native-exchange block that has not finished is given up: rank 0 "
                    "prints the line it has (with the timeout as the block's \"error\"), every rank exits")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if world not in GRID:
        raise SystemExit("--gpus must be 1, 2, 4 or 8")
    if args.one_device and (args.backend != "gloo" or args.exchange != "torch"):
        raise SystemExit("--one-device needs --backend gloo --exchange torch")
    device = 0 if args.one_device else local_rank
    torch.cuda.set_device(device)
    dist = None
    use_gather = world > 1 or args.force_gather
    # N > 1 (and the one-rank --force-gather proxy): batch launches with ONE gather per launch - through torch.distributed.gather or, with
    # --exchange native, through the C ABI (vkv_assemble_frames) - or round 1's single-frame launches with one exchange per frame
    # (--submit streams)
    submit = args.submit or "batch"
    if submit == "streams":
        # several single-frame launches in flight prefer the plain tile order (their heavy image centres then do not coincide: 0.134 vs
        # 0.141 ms per frame on C3); the library reads the switch once, before its first launch
        os.environ.setdefault("VKV_RAYMARCH_TILE_ORDER", "linear")
    if use_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # no device_id: an eagerly created communicator measured 20 % slower on this pipeline than the lazily created one
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    ctx = lib.Context(device)  # raises if the HIP library is missing: there is no fallback path
    env = {"world": world, "rank": rank, "local_rank": device, "dist": dist, "ctx": ctx, "use_gather": use_gather, "submit": submit}
    import copy
    import threading
    printed = threading.Lock()
    state = {"out": None, "done": False}

    def emit():
        """rank 0's ONE line (whoever gets here first: the main thread at the end, or a watchdog that gave a side block up)"""
        with printed:
            if not state["done"] and rank == 0 and state["out"] is not None:
                print(json.dumps(state["out"]), file=json_out, flush=True)
            state["done"] = True

    def side_block(key, holder, **over):
        """A short extra measurement with some options replaced, reported as holder[key].  It can never cost the headline: an exception becomes
        {"error": ...} (on every rank: the ranks agree on the outcome through one all-reduce), and a block that hangs - a collective one rank never
        enters - is given up after --native-timeout seconds by a watchdog that prints the line as it stands and ends the process with status 124
        (fresh processes were started by the launcher; nothing is re-executed)."""
        a2 = copy.copy(args)
        for k, val in over.items():
            setattr(a2, k, val)
        a2.no_cpu_baseline, a2.no_depth_block = True, True

        def give_up():
            if rank == 0 and holder is not None:
                holder[key] = {"error": "gave up after %.0f s (--native-timeout): the block did not finish" % args.native_timeout}
            emit()  # the headline first, so the measurement is not lost
            os._exit(124)  # non-zero like the --launch-timeout watchdog: a run whose collective hung is not a success

        dog = threading.Timer(args.native_timeout, give_up) if (world > 1 or use_gather) and args.native_timeout > 0 else None
        if dog:
            dog.daemon = True
            dog.start()
        res, err = None, None
        try:
            torch.cuda.empty_cache()
            res = job(a2, env)
        except BaseException as e:  # SystemExit (a failed --verify) as well
            err = "%s: %s" % (type(e).__name__, e)
        try:
            if dist is not None and world > 1:
                flag = torch.tensor([1.0 if err else 0.0], dtype=torch.float64, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if flag.item() > 0 and err is None:
                    err = "another rank failed"
        except BaseException as e:
            err = err or ("%s: %s" % (type(e).__name__, e))
        if dog:
            dog.cancel()
        if rank == 0 and holder is not None:
            holder[key] = {"error": err} if err else res
        return None if err else res

    def brief(o, extra=()):
        d = {k: o[k] for k in ("ms_per_step", "value", "unit", "steps", "warmup", "repeats", "scaling", "covered_Mray_per_s", "phases", "rccl_ranks",
                                "rccl_ranks_source", "volume_samples_per_s", "distance_probes_per_s", "exchange_bytes_per_frame") + tuple(extra) if k in o}
        d.update(frac=o["roofline"]["frac"], achieved=o["roofline"]["achieved"], workload=o["config"]["workload"], parallelism=o["config"]["parallelism"])
        return d

    def with_native(o, **over):
        """N > 1, torch exchange: the same launches once more, briefly, through the C ABI's own exchange (vkv_assemble_frames) - so that the first
        run on real links carries the product's gather next to torch.distributed's"""
        if not (use_gather and args.exchange == "torch" and submit == "batch" and (args.native_block == "on" or (args.native_block == "auto" and args.backend == "nccl"))):
            return
        # twice: the launch's owner rotating (launch l -> rank l mod N, ONE ncclGather per launch) and the owners spread inside a launch (frame j of
        # launch l -> rank (l + j) mod N, one RCCL group of gathers per launch: every GPU receives at once)
        for key, owner in (("native_exchange", args.frame_owner if args.frame_owner != "spread" else "rotate"), ("native_exchange_spread", "spread")):
            tmp = {}
            side_block("n", tmp if rank == 0 else None, exchange="native", frame_owner=owner, steps=min(args.steps, 16), warmup=min(args.warmup, 8),
                       min_seconds=min(args.min_seconds, 0.5), **over)
            if rank == 0 and o is not None:
                r = tmp.get("n")
                o[key] = r if (r is None or "error" in r) else brief(r)

    out = job(args, env)
    state["out"] = out
    with_native(out)
    # BASELINE.json configs[4] is the configuration north_star names for 8 GPUs: at N > 1 the same ranks then time a short block of it (strong
    # scaling: the 7680x4320 frame is fixed, its tiles are dealt over the ranks); c3 stays the headline so that N = 1 agrees with BENCH
    if args.c5_block == "on" or (args.c5_block == "auto" and world > 1 and args.workload != "c5"):
        c5 = dict(workload="c5", scaling="strong", skip=None, tf="app", no_ert=False, steps=min(args.steps, 16), warmup=min(args.warmup, 8),
                  min_seconds=min(args.min_seconds, 1.0))
        tmp = {}
        o2 = side_block("c5", tmp if rank == 0 else None, **c5)
        if rank == 0:
            out["c5_strong"] = brief(o2) if o2 is not None else tmp.get("c5")
        if o2 is not None or rank != 0:
            with_native(out["c5_strong"] if rank == 0 and o2 is not None else None, **{k: c5[k] for k in ("workload", "scaling", "skip", "tf", "no_ert")})
    # N = 1, behind the headline (each a short block that can never cost it): the other single-GPU configurations of BASELINE.json on the same
    # clock - the literal 1024^3 of the metric line, configs[1] (c2: 512^3, block ESS) and configs[3] (c4: 2048^3, anisotropic maps, 3840x2160,
    # "the HBM-roofline run") -, the same submission with a camera that MOVES and with the start-order feedback off, and the integrator's asymptotes
    default_run = args.workload == "c3" and args.skip is None and args.tf == "app" and not args.no_ert and submit == "batch" and args.camera == "static" and not args.no_feedback
    if world == 1 and not use_gather and not args.virtual_rank and (args.extras == "on" or (args.extras == "auto" and default_run)):
        short = dict(steps=min(args.steps, 16), warmup=min(args.warmup, 4), min_seconds=min(args.min_seconds, 0.5))
        for key, wl, note in (("c3cube", "c3cube", "BASELINE.json's metric line names 1024^3: the same submission on the literal cube"),
                              ("c2", "c2", "BASELINE.json configs[1]: 512^3, occupancy-grid (block) ESS only"),
                              ("c4", "c4", "BASELINE.json configs[3], 'the HBM-roofline run': 2048^3 (37 GB packed sampling image, 1.1 GiB of maps), "
                                           "anisotropic Chebyshev maps, 3840x2160")):
            o3 = side_block(key, out, workload=wl, **short)
            if o3 is not None:
                out[key] = brief(o3, ("single_frame",))
                out[key]["note"] = note + "; a short block behind the headline, on the driver's clock"
                for k in ("traffic", "traffic_over_algorithmic", "traffic_source", "wait_frac", "valu_issue_frac", "waves_per_simd", "valu_active_lanes"):
                    if o3["roofline"].get(k) is not None:
                        out[key][k] = o3["roofline"][k]
        # a camera that moves (every frame a new view, one degree of orbit from the last, targets in turn) and the start-order feedback switched
        # off, each measured here: the headline holds 8 views still, each target showing one of them - the best case for the feedback
        cam = {}
        for key, over in (("static", dict()), ("static_feedback_off", dict(no_feedback=True)), ("moving", dict(camera="moving")),
                          ("moving_feedback_off", dict(camera="moving", no_feedback=True))):
            o6 = side_block(key, cam, workload=args.workload, **over, **short)
            if o6 is not None:
                cam[key] = {"ms_per_step": o6["ms_per_step"], "frac": o6["roofline"]["frac"], "value": o6["value"], "steps": o6["steps"], "repeats": o6["repeats"]}
        cam["what"] = ("the headline's submission with --camera moving (one degree of orbit per frame; a block sweeps its %d frames forwards, the next one backwards, so the "
                       "camera never jumps; the targets take the frames in turn, a target's consecutive frames are up to %d degrees apart) and / or with "
                       "VkvTuning.feedback = 0 (every frame starts its tiles centre-first); `static` = the headline's configuration once more in these short blocks "
                       "(%d steps: a block's launches do not fill the headline's streams, compare the four with each other)" % (short["steps"], short["steps"] - 1, short["steps"]))
        out["camera_and_feedback"] = cam
        asym = {}
        o4 = side_block("dense", asym, workload=args.workload, skip="none", no_ert=True, steps=min(args.steps, 8), warmup=2, min_seconds=min(args.min_seconds, 0.3))
        if o4 is not None:
            asym["dense"] = {"ms_per_step": o4["ms_per_step"], "frac": o4["roofline"]["frac"], "volume_samples_per_s": o4["volume_samples_per_s"],
                             "what": "no empty-space skipping, no early termination: every lane samples at every step of its ray (16 B per sample)"}
        o5 = side_block("probe_only", asym, workload=args.workload, skip="block", empty_volume=True, **short)
        if o5 is not None:
            asym["probe_only"] = {"ms_per_step": o5["ms_per_step"], "frac": o5["roofline"]["frac"], "distance_probes_per_s": o5["distance_probes_per_s"],
                                  "volume_samples_per_s": o5["volume_samples_per_s"],
                                  "what": "the same box with every voxel empty and the 0 / 255 block map: a ray samples once, then walks one map cell per probe "
                                          "to the far side (1 B per probe)"}
        # lanes per VALU instruction of the headline's kernel, from the PMC passes behind roofline.traffic (digest-tied to the kernel sources like
        # it): with every instruction on 64 lanes the same instruction stream would deliver 64 / valu_active_lanes times the events - the ceiling of
        # lane = ray, unreachable (no re-packing scheme keeps the frag's event sequence for free, profiles/HISTORY.md section 5.2)
        lanes = out["roofline"].get("valu_active_lanes")
        if lanes:
            asym["full_lanes"] = {"valu_active_lanes": lanes, "of": 64, "speedup": round(64.0 / lanes, 3), "frac": round(64.0 / lanes * out["roofline"]["frac"], 4),
                                  "what": "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU of the headline's kernel (rocprofv3 --pmc, %s): the average number of live "
                                          "lanes of a VALU instruction; frac = the headline's fraction if every instruction ran on 64" % os.path.basename(traffic_file(args.workload))}
        out["asymptotes"] = asym
    emit()
    if dist is not None:
        dist.destroy_process_group()


def job(args, env):
    """One measurement (scene set-up, pre-pass, warm-up, the timed blocks) of args.workload on the ranks of `env`; returns the result
    line as a dict on rank 0, None elsewhere."""
    world, rank, local_rank, dist, ctx, use_gather, submit = (env[k] for k in ("world", "rank", "local_rank", "dist", "ctx", "use_gather", "submit"))
    if getattr(args, "no_feedback", False) and ctx.get_tuning().feedback:
        ctx.set_tuning(feedback=0)  # every frame starts its tiles centre-first
        try:
            return job(args, env)
        finally:
            ctx.set_tuning(feedback=1)
    v, tf, frame, skip = build_scene(ctx, args.workload, args.tf, empty=getattr(args, "empty_volume", False))
    if args.skip is not None:
        skip = {"none": abi.SKIP_NONE, "block": abi.SKIP_BLOCK, "distance": abi.SKIP_DISTANCE, "anisotropic": abi.SKIP_ANISOTROPIC_DISTANCE}[args.skip]
        V.ComputeDistanceMap(ctx).compute(v, tf, skip)
        torch.cuda.synchronize()
    scaling = args.scaling or WORKLOADS[args.workload][6]
    sx, sy = GRID[world] if scaling == "weak" else (1, 1)
    fw, fh = frame[0] * sx, frame[1] * sy
    # --camera moving: every frame of a block is a new view, one degree of orbit from the last (blocks sweep forwards and backwards in turn, so the
    # camera never jumps and every block renders the same views); the default holds 8 views still, each target showing one of them
    moving = getattr(args, "camera", "static") == "moving"
    views = cameras(v, frame[0] / frame[1], max(args.steps, args.warmup) if moving else 0, getattr(args, "camera_step", 1.0))  # the SAME frustum for every N
    n_views = len(views)
    direction = [0]  # moving camera: 0 = this block sweeps forwards, 1 = backwards

    def vidx(k):
        """view of step k of the current block"""
        if not moving:
            return k % n_views
        return k if direction[0] == 0 else args.steps - 1 - k
    opts = abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=not args.no_ert)
    sp = V.VolumeRenderSubpass(ctx, v, opts, (fw, fh))
    virtual = None
    if getattr(args, "virtual_rank", None):
        if world != 1 or use_gather:
            raise SystemExit("--virtual-rank is a one-GPU proxy: not with --gpus > 1 or --force-gather")
        virtual = tuple(int(x) for x in args.virtual_rank.split("/"))
        if len(virtual) != 2 or not (0 <= virtual[0] < virtual[1]):
            raise SystemExit("--virtual-rank R/N needs 0 <= R < N")
    compact = use_gather or virtual is not None
    my_rank, n_ranks = (virtual if virtual else (rank, world))
    # Sharded frames (N > 1, the one-rank proxies): only the tiles of each view's screen rectangle are scheduled and exchanged - the tile rectangle
    # the clipped box projects into, derived by every rank from the uniforms alone (vkv_screen_tile_rect); the de-interleave clears the rest.
    use_rect = compact and getattr(args, "tile_rect", "on") != "off"
    # One GPU, whole frames (round 6): the batch launches schedule the tiles of each view's screen rectangle too and their workgroups write the
    # no-fragment result to every pixel outside it (VkvTileSchedule.fill_outside): the same frame, complete, without a workgroup per empty tile
    use_fill = not compact and submit == "batch" and getattr(args, "tile_rect", "on") != "off"
    whole_rect = abi.whole_image_rect(fw, fh, TILE, TILE)
    whole_tiles = abi.full_frame_tiles(fw, fh, TILE, TILE)
    params, params_whole, rects, tiles_v = [], [], [], []
    for view, proj in views:
        p = sp.make_params(view, proj, whole_tiles)
        params_whole.append(abi.RenderParams.from_buffer_copy(p))  # the whole-image schedule: counter pre-pass, single-frame launches, CPU check
        r = lib.screen_tile_rect(p.ray_cast, p.ray_gen, (fw, fh), (TILE, TILE), max(1, getattr(args, "rect_align", 1))) if (use_rect or use_fill) else whole_rect
        if use_fill:
            t = abi.full_frame_tiles(fw, fh, TILE, TILE, rect=r, fill_outside=True)
        else:
            t = abi.full_frame_tiles(fw, fh, TILE, TILE, my_rank, n_ranks, compact=compact, rect=r if use_rect else None)
        p.tiles = t
        params.append(p)
        rects.append(r)
        tiles_v.append(t)
    tiles = whole_tiles  # (single-frame launches: the whole-image schedule)
    pixels_v = [t.tile_count * TILE * TILE if compact else fw * fh for t in tiles_v]  # output slots of this rank per view
    my_pixels = max(pixels_v)
    # every pixel of the frame is a ray (covered or not), summed over ranks - the assembled frame is complete, whatever was exchanged; a virtual
    # rank counts the share of the WHOLE frame's pixels it stands for (1 / N of them: its value stays comparable with the frame's)
    rays_per_frame_all = (fw * fh / n_ranks) if virtual else fw * fh

    # ---- pre-pass (untimed): frag counters per view -> algorithmic bytes per frame ---------------------------------
    counts = torch.zeros((max(my_pixels, 1), 3), dtype=torch.int32, device="cuda")
    n_vs, n_ds, n_cov = [], [], []
    for p in (params if compact else params_whole):
        counts.zero_()
        sp.draw(p, counts=counts)
        torch.cuda.synchronize()
        s = counts.to(torch.int64).sum(0).cpu().numpy()
        n_vs.append(int(s[0]))
        n_ds.append(int(s[1]))
        n_cov.append(int(((counts[:, 0] + counts[:, 1]) > 0).sum().item()))
    del counts

    # ---- outputs -------------------------------------------------------------------------------------------------
    fif = max(1, args.frames_in_flight)
    fpl = max(1, min(args.frames_per_launch, abi.MAX_BATCH, args.steps))
    gather, images, rotate, spread = None, [], False, False
    nbs = max(1, args.batch_streams) if submit == "batch" else 1
    nsets = nbs + 1  # gather path: one more buffer set than render streams, so a launch does not wait for the exchange nbs launches back
    batch_gather = use_gather and submit == "batch"
    if batch_gather:
        rotate = args.frame_owner in ("rotate", "spread") and world > 1
        # spread: frame j of launch l is assembled on rank (l + j) mod N (one gather per frame; natively all of a launch's in ONE RCCL group), so
        # every GPU receives at the same time over its own inbound links; the one-rank proxy takes the same code path with every owner 0
        spread = args.frame_owner == "spread"
        if args.exchange == "native":
            gather = multigpu.NativeBatchExchange(ctx, dist, rank, world, (fw, fh), TILE, 4, frames=fpl, n_sets=nsets, any_root=rotate)
            images = gather.images or []
        else:
            gather = multigpu.BatchTileGather(dist, rank, world, (fw, fh), TILE, 4, device="cuda", frames=fpl, n_sets=nsets, any_root=rotate,
                                              host_staging=args.backend == "gloo")
            if rank == 0 or rotate:
                images = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(nsets * fpl)]
        nbuf = nsets * fpl
        bufs = None  # the frames of a launch lie back to back in the launch's buffer set (gather.frame_pointer)
    elif use_gather:
        nbuf = fif + 1  # one more buffer than render streams: a render does not wait for the gather of the frame fif steps back
        rotate = args.frame_owner in ("rotate", "spread") and world > 1
        if args.exchange == "native":
            gather = multigpu.NativeExchange(ctx, dist, rank, world, (fw, fh), TILE, 4, n_buffers=nbuf, any_root=rotate)
            images = gather.images or []
        else:
            if args.backend == "gloo":
                raise SystemExit("--backend gloo supports the batch submission only")
            gather = multigpu.TileGather(dist, rank, world, (fw, fh), TILE, 4, device="cuda", n_buffers=nbuf, any_root=rotate)
            if rank == 0 or rotate:
                images = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        bufs = gather.buffers
    else:
        # per-frame submission: as many targets as keep every target on ONE of the orbit views (like the slots of the batch path)
        nbuf = fpl * nbs if submit == "batch" else -(-fif // N_VIEWS) * N_VIEWS
        if virtual:
            bufs = [torch.zeros((max(my_pixels, 1), 4), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        else:
            bufs = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    # in-image pixels this rank renders of every view (the 4 B per ray of the algorithmic bytes; edge tiles of the frame are partial)
    if compact:
        rays_v = [multigpu.rect_ray_count(r, my_rank, n_ranks, (fw, fh), TILE) for r in rects]
    else:
        rays_v = [fw * fh] * n_views
    # what the exchange moves: ceil(tiles of the view's rectangle / N) tiles of RGBA8 from every rank to the frame's owner
    exchange_bytes_v = [multigpu.tiles_per_rank(r, n_ranks) * TILE * TILE * B_OUT for r in rects]
    exchange_bytes_whole = multigpu.tiles_per_rank(whole_rect, n_ranks) * TILE * TILE * B_OUT
    # set-up: the per-target state of the start-order feedback (vkv_render itself never allocates); a target is registered once, for the
    # schedule of the first view it shows (frames of another schedule into it start centre-first)
    registered = set()

    def register(ptr, view_i):
        if ptr not in registered:
            registered.add(ptr)
            ctx.register_target(ptr, (fw, fh), tiles_v[view_i])

    # algorithmic bytes of one frame (this rank's part): 8 B per trilinear footprint of the volume, 8 more for the gradient map's when the
    # transfer function has a gradient term, 1 B per distance probe
    b_sample = 16 if tf.use_gradient else 8
    frame_bytes = [n_vs[i] * b_sample + n_ds[i] * 1 + rays_v[i] * B_OUT for i in range(n_views)]
    # the HIP streams live as long as the process (one pool for the headline and its side blocks: a stream a context has seen keeps a scratch block)
    pool = env.setdefault("streams", [])
    while len(pool) < max(fif, nbs) - 1:
        pool.append(torch.cuda.Stream())
    streams = [torch.cuda.current_stream()] + pool[:max(fif, nbs) - 1]
    # the exchange streams get the higher priority: their small kernels (RCCL's gather, the de-interleave) must not queue behind the render
    # workgroups of the next frames (native exchange with a one-rank group: 0.25 -> 0.19 ms per step)
    if gather and "side" not in env:
        env["side"], env["xchg"] = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)
    side = env["side"] if gather else None
    xchg = env["xchg"] if gather else None  # native exchange: RCCL's gather here, the de-interleave on `side`
    freed = [None] * (nsets if batch_gather else nbuf)
    launches = []  # (start event, stop event, algorithmic bytes) of the timed launches of the last block
    phases = []  # gather path, launches this rank owns: (render start, render end, gathered or None, de-interleaved) events
    phase_ms = {"render_ms": [], "gather_ms": [], "scatter_ms": [], "exchange_ms": []}

    def plan(n_steps):
        """the launches of a block of n_steps: (first step, frames, stream index, buffer set, owner of the launch, owners of its frames or None)"""
        k, out_ = 0, []
        for launch, n in enumerate(split_frames(n_steps, fpl)):  # 20 steps with 6 per launch: 5 + 5 + 5 + 5 (8 per launch: 7 + 7 + 6, not 8 + 8 + 4)
            owner = (launch % world) if rotate else 0  # gather path: the rank that assembles this launch's frames
            roots = [((launch + j) % world) if rotate else 0 for j in range(n)] if spread else None
            out_.append((k, n, launch % nbs, (launch % nsets) if gather else (launch % nbs), owner, roots))
            k += n
        return out_

    # per-launch parameter blocks of the batch path.  N = 1: slot j of a launch renders into bufs[j].  Gather path: the frames of a launch lie
    # back to back in the launch's buffer set - frame j at the tile offset its predecessors' rectangles leave (multigpu.launch_layout) - so the
    # blocks depend on which views a launch holds: built once per (first view, frames, set) before the warm-up
    depth_bufs = None
    want_depth = submit == "batch" and not use_gather and not args.no_depth_block and not virtual
    if want_depth:
        depth_bufs = [torch.zeros((fh, fw), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
    launch_cache = {}

    def launch_params(k, n, slot, depth):
        idx = tuple(vidx(k + j) for j in range(n))
        key = (idx, slot, depth)
        hit = launch_cache.get(key)
        if hit is not None:
            return hit
        rl = [rects[i] for i in idx]
        off = multigpu.launch_layout(rl, world)[1] if batch_gather else None
        plist = []
        for j, i in enumerate(idx):
            q = abi.RenderParams.from_buffer_copy(params[i])
            target = gather.frame_pointer(slot, off[j]) if batch_gather else bufs[slot * fpl + j].data_ptr()
            q.d_out_rgba8, q.d_out_color, q.d_out_counts = target, None, None
            q.d_out_depth = depth_bufs[slot * fpl + j].data_ptr() if depth else None  # gl_FragDepth as well (frag:315-321): ms_per_step_with_depth
            q.d_in_depth, q.blend_over_target = None, 0
            register(target, i)
            plist.append(q)
        launch_cache[key] = (plist, rl)
        return launch_cache[key]

    if submit == "batch":
        for n_steps, rev in ((args.warmup, 0), (args.steps, 0), (args.steps, 1)):
            direction[0] = rev if moving else 0
            for k, n, _, slot, _, _ in plan(n_steps):
                launch_params(k, n, slot, False)
                if want_depth:
                    launch_params(k, n, slot, True)
        direction[0] = 0
    else:
        for j, t in enumerate(bufs):
            register(t.data_ptr(), j % n_views)
    torch.cuda.synchronize()

    last_slot, last_owner, last_view = [0], [0], [0]
    with_depth = [False]
    native = gather is not None and args.exchange == "native"

    def run_batch(n_steps, timed):
        for k, n, si, slot, owner, roots in plan(n_steps):
            st = streams[si]  # stream of this launch
            plist, rl = launch_params(k, n, slot, with_depth[0])
            last_slot[0], last_view[0] = slot * fpl + n - 1, vidx(k + n - 1)  # output buffer and view of the block's last step (--verify)
            mine = (rank in roots) if roots else (rank == owner)  # this rank assembles a frame of the launch
            with torch.cuda.stream(st):
                if gather and freed[slot] is not None:
                    st.wait_event(freed[slot])  # the exchange that last used this buffer set has read it
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                if n == 1:
                    ctx.render(plist[0], st.cuda_stream)
                else:
                    ctx.render_batch(plist, st.cuda_stream)
                if timed:
                    e1.record(st)
                    launches.append((e0, e1, sum(frame_bytes[vidx(k + j)] for j in range(n)), n))
                if gather and native:
                    rendered = torch.cuda.Event()
                    rendered.record(st)
                elif gather:
                    gather.start(slot, owner, rl, roots)  # ONE collective for the launch's n frames (or one per frame), ordered after the render
            if gather and native:
                # vkv_assemble_frames on the exchange stream, behind the launch: one ncclGather (or one group) + one de-interleave kernel, nothing waits on the host
                xchg.wait_event(rendered)
                gather.assemble(slot, owner, n, xchg, rects=rl if use_rect else None, roots=roots)
                if timed and mine:  # phases of a launch on a rank that assembles part of it: render, then gather + de-interleave (one call here)
                    x1 = torch.cuda.Event(enable_timing=True)
                    x1.record(xchg)
                    phases.append((e0, e1, None, x1))
                freed[slot] = torch.cuda.Event()
                freed[slot].record(xchg)
                last_owner[0] = roots[-1] if roots else owner
            elif gather:
                with torch.cuda.stream(side):
                    got = gather.finish(slot)
                    if got:
                        g1 = None
                        if timed:
                            g1 = torch.cuda.Event(enable_timing=True)
                            g1.record(side)  # the launch's frames this rank assembles have arrived from every rank
                        for f, src, stride, r in got:
                            ctx.scatter_tiles(src, images[slot * fpl + f].data_ptr(), (fw, fh), (TILE, TILE), world, stride, 4, side.cuda_stream, rect=r)
                        if timed:
                            s1 = torch.cuda.Event(enable_timing=True)
                            s1.record(side)
                            phases.append((e0, e1, g1, s1))
                    freed[slot] = torch.cuda.Event()
                    freed[slot].record(side)
                last_owner[0] = roots[-1] if roots else owner

    # HIP events bracket every launch at N = 1; on the gather path every 7th (a timing event per launch costs ~25 us per step
    # there, more than 10 % of the step, once seven queues are busy)
    ev_every = 7 if use_gather else 1  # coprime with the 8 views

    def run_streams(n_steps, timed):
        for k in range(n_steps):
            b, st = k % nbuf, streams[k % fif]
            with torch.cuda.stream(st):
                if gather and freed[b] is not None:
                    st.wait_event(freed[b])  # frame k - nbuf: its gather has read bufs[b], its de-interleave has read flat[b]
                ev = timed and k % ev_every == 0
                if ev:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                sp.draw(params[vidx(k)], rgba8=bufs[b])
                last_view[0] = vidx(k)
                if ev:
                    e1.record(st)
                    launches.append((e0, e1, frame_bytes[vidx(k)], 1))
                if gather and native:
                    rendered = torch.cuda.Event()
                    rendered.record(st)
                elif gather:
                    # RCCL gather of frame k (ordered after the render) overlaps the renders of the next frames
                    gather.start(b, k % world if rotate else 0, rects[vidx(k)])
            if gather and native:
                # gather + de-interleave of frame k on the assembly stream, behind its render: nothing waits on the host
                xchg.wait_event(rendered)
                last = gather.assemble(b, k % world if rotate else 0, xchg, side, rect=rects[vidx(k)])
                freed[b] = torch.cuda.Event()
                freed[b].record(last)
            elif gather:
                with torch.cuda.stream(side):
                    flat = gather.finish(b)
                    if flat is not None:
                        ctx.scatter_tiles(flat.data_ptr(), images[b].data_ptr(), (fw, fh), (TILE, TILE), world, gather.tiles_per_rank, 4,
                                          side.cuda_stream, rect=rects[vidx(k)])
                    freed[b] = torch.cuda.Event()
                    freed[b].record(side)

    run = run_batch if submit == "batch" else run_streams

    def fence():
        if world > 1:
            if args.backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                torch.cuda.synchronize()
                dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup, False)
    fence()
    # ---- timed blocks: exactly K steps between two fences, repeated until min_seconds have been measured -----------
    blocks, kernel_ms, kernel_bytes, kernel_frames, enqueue_times = [], [], [], [], []
    total = 0.0
    while True:
        del launches[:]
        del phases[:]
        fence()
        direction[0] = (len(blocks) % 2) if moving else 0  # moving camera: forwards, backwards, forwards, ...
        t0 = time.perf_counter()
        run(args.steps, True)
        host_enqueue = time.perf_counter() - t0  # the host's share: how long the loop took to enqueue the block
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        blocks.append(elapsed)
        enqueue_times.append(host_enqueue)
        kernel_ms += [e0.elapsed_time(e1) for e0, e1, _, _ in launches]
        for p0, p1, pg, ps in phases:
            phase_ms["render_ms"].append(p0.elapsed_time(p1))
            if pg is not None:
                phase_ms["gather_ms"].append(p1.elapsed_time(pg))
                phase_ms["scatter_ms"].append(pg.elapsed_time(ps))
            else:
                phase_ms["exchange_ms"].append(p1.elapsed_time(ps))
        del phases[:]
        kernel_bytes += [b for _, _, b, _ in launches]
        kernel_frames += [n for _, _, _, n in launches]
        total += elapsed
        stop = total >= args.min_seconds or len(blocks) >= 4096
        if world > 1:
            t = torch.tensor([1.0 if stop else 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            stop = t.item() > 0
        if stop:
            break
    elapsed = float(np.median(blocks))
    kernel_ms_avg = sum(kernel_ms) / len(kernel_ms)

    # ---- the same block with gl_FragDepth written too (not the headline: BASELINE's metric is the colour frame) ------
    depth_ms = None
    if want_depth:
        with_depth[0] = True
        run(args.warmup, False)
        dblocks, dtotal = [], 0.0
        while dtotal < min(0.5, args.min_seconds) and len(dblocks) < 1024:
            fence()
            t0 = time.perf_counter()
            run(args.steps, False)
            fence()
            dblocks.append(time.perf_counter() - t0)
            dtotal += dblocks[-1]
        with_depth[0] = False
        depth_ms = float(np.median(dblocks)) / args.steps * 1e3
    alg_avg = sum(kernel_bytes) / len(kernel_bytes)
    achieved_gbs = alg_avg / (kernel_ms_avg * 1e-3) / 1e9

    # ---- one frame at a time (outside the timed region): the latency of a single frame's launch ---------------------
    single = None
    if not use_gather and not virtual and not moving:
        # one target per view (a static camera per target, as in the timed loop), 2 untimed + 5 timed launches of a view back to back as in
        # the earlier rounds' figure; the median per view (one frame in eight also measures tile costs and is followed by the 20 us sort
        # kernel: inside the bracket, outside the median)
        own = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(N_VIEWS)]
        for t in own:
            ctx.register_target(t.data_ptr(), (fw, fh), tiles)
        ts = [[] for _ in range(N_VIEWS)]
        for i in range(N_VIEWS):
            for rnd in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                sp.draw(params_whole[i], rgba8=own[i])
                e1.record()
                torch.cuda.synchronize()
                if rnd >= 2:
                    ts[i].append(e0.elapsed_time(e1))
        per_view = [float(np.median(t)) for t in ts]
        ms1 = float(np.mean(per_view))
        gb1 = float(np.mean(frame_bytes)) / (ms1 * 1e-3) / 1e9
        single = {"ms_per_launch": round(ms1, 4), "achieved": round(gb1, 2), "frac": round(gb1 / HBM_PEAK_GBS, 5),
                  "Mray_per_s": round(rays_per_frame_all / ms1 / 1e3, 1), "per_view_ms": [round(x, 4) for x in per_view],
                  "note": "one vkv_render launch per frame with nothing else on the GPU: a view rendered 7 times in a row into its own target (a static camera: start order from the costs measured on the first of them, caches warm from the same view; the views in turn measure about 0.245, VKV_RAYMARCH_FEEDBACK=0 about 0.25 / 0.245 ms), median of the last 5, mean over the 8 views"}

    # whole-job sample rates need every rank's counters
    direction[0] = 0
    tot = torch.tensor([sum(n_vs[vidx(k)] for k in range(args.steps)), sum(n_ds[vidx(k)] for k in range(args.steps)),
                        sum(n_cov[vidx(k)] for k in range(args.steps))], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tot)
    vs_total, ds_total, cov_total = float(tot[0].item()), float(tot[1].item()), float(tot[2].item())

    if args.verify:
        verify(ctx, sp, v, views[last_view[0]], params, args.steps, nbuf, (last_slot[0] + 1) if submit == "batch" else 0, (fw, fh), bufs, images, gather, rank,
               last_owner[0] if submit == "batch" else ((args.steps - 1) % world if rotate else 0))
    if rank != 0:
        if native:
            gather.close()
        return None

    value = rays_per_frame_all * args.steps / elapsed / 1e6
    aggregate_gbs = sum(frame_bytes[vidx(k)] for k in range(args.steps)) / elapsed / 1e9
    kernel_name = "k_raymarch_lean_batch" if (submit == "batch" and fpl > 1) else "k_raymarch_lean"
    concurrent = nbs if submit == "batch" else fif
    extent = WORKLOADS[args.workload][0]
    out = {
        "metric": "Mray/s", "value": round(value, 3), "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "rccl_ranks": (gather.comm_count() if native else (dist.get_world_size() if dist is not None else 1)),
        "rccl_ranks_source": ("ncclCommCount of the exchange's own communicator" if native else ("torch.distributed.get_world_size (backend %s)" % ("nccl = RCCL" if args.backend == "nccl" else args.backend) if dist is not None else "no communicator")),
        "repeats": len(blocks), "host_enqueue_ms_per_step": round(float(np.median(enqueue_times)) / args.steps * 1e3, 4), "ms_per_step_min_max": [round(min(blocks) / args.steps * 1e3, 4), round(max(blocks) / args.steps * 1e3, 4)],
        "config": {"workload": "%s (%s): %dx%dx%d uint8 synthetic shells, %dx%d frame, %s, block 4, TF %s, "
                               "8 orbit views" % (args.workload, WORKLOAD_NOTE[args.workload], *extent, fw, fh,
                                                  {0: "no ESS", 1: "block ESS", 2: "Chebyshev distance-map ESS", 3: "anisotropic Chebyshev distance-map ESS"}[skip]
                                                  + (" + ERT" if not args.no_ert else ", no ERT"),
                                                  "imin 0.1 imax 1 gmin 0 gmax 0.2" if args.tf == "app" else "imin 0.1 imax 1 gmin 0 gmax 0 (intensity only)"),
                   "parallelism": "16x16 screen tiles %s round-robin over %d GPU(s), RCCL gather to %s" % (
                       "of each frame's screen rectangle (vkv_screen_tile_rect)" if use_rect else "of the whole frame", world,
                       ("rank (l + j) mod N for frame j of launch l (one gather per frame, one RCCL group per launch)" if spread else
                        ("rank l mod N for launch l (one gather per launch)" if submit == "batch" else "rank k mod N for frame k")) if rotate else "rank 0")
                   + ((" (vkv_assemble_frames: one ncclGather + one de-interleave per launch)" if submit == "batch" else " (vkv_assemble_frame: ncclGather + de-interleave)") if native else (" (torch.distributed.gather)" if args.backend == "nccl" else " (torch.distributed.gather over GLOO through host memory: a functional run, not a measurement)"))
                   if world > 1 else "1 GPU",
                   "output": "RGBA8",
                   "schedule": ("the 16x16 tiles of each view's screen rectangle (vkv_screen_tile_rect: %s of the frame's %d tiles), the launch's workgroups write the no-fragment result to every pixel outside it (VkvTileSchedule.fill_outside): the whole frame, every pixel written" % (
                       "/".join(str(int(r.w * r.h)) for r in rects[:N_VIEWS]), int(whole_rect.w * whole_rect.h))) if use_fill else ("tiles of each view's screen rectangle, compact buffers" if use_rect else "every tile of the frame"),
                   "submission": ("vkv_render_batch, up to %d frames per launch, consecutive launches on %d HIP stream(s); tile start order from the "
                                  "tile costs measured on earlier frames into the same target%s" % (fpl, nbs, (" (switched off: VkvTuning.feedback = 0)" if args.no_feedback else "") +
                                                                                                    (" (a camera that moves %g degree(s) of orbit per frame, the targets taking the frames in turn)" % args.camera_step if moving else
                                                                                                     (" (each target shows the same orbit view every block: the best case for that feedback; the line's camera_and_feedback block measures the others)" if not args.no_feedback else "")))) if submit == "batch" else ("%d single-frame launches in flight on %d HIP streams (measured start order per target as well; each of the %d targets shows one orbit view)" % (fif, fif, nbuf)),
                   "occupied_voxel_percent": round(occupied_voxel_percent(ctx, v, tf), 4)},
        "covered_Mray_per_s": round(cov_total / elapsed / 1e6, 3), "covered_fraction": round(cov_total / (rays_per_frame_all * args.steps), 4),
        "volume_samples_per_s": round(vs_total / elapsed, 1), "distance_probes_per_s": round(ds_total / elapsed, 1),
        "roofline": {"bound": "hbm", "achieved": round(aggregate_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(aggregate_gbs / HBM_PEAK_GBS, 5), "traffic": None,
                     "kernel": kernel_name, "kernel_ms_avg": round(kernel_ms_avg, 4),
                     "frames_per_launch": round(sum(kernel_frames) / len(kernel_frames), 3),
                     "algorithmic_bytes_per_launch": int(alg_avg),
                     "achieved_per_launch": round(achieved_gbs, 2), "frac_per_launch": round(achieved_gbs / HBM_PEAK_GBS, 5),
                     "concurrent_launches": concurrent,
                     "note": "algorithmic (requested) bytes: 16 B/volume sample (8 with the intensity-only TF) + 1 B/distance probe + 4 B/ray, summed over the frames of a launch; NOT "
                             "DRAM traffic. achieved / frac = the DEVICE-level rate: this rank's algorithmic bytes of one timed block / the block's wall "
                             "time between its fences (what the chip delivered, whatever the launches' overlap). achieved_per_launch / frac_per_launch = "
                             "bytes of one launch / the HIP-event duration of that launch (kernel_ms_avg), averaged over the timed launches: with "
                             "concurrent_launches > 1 the launches of a block overlap, one launch's duration then covers work of the others and the "
                             "per-launch figure is only that launch's share of the device; with concurrent_launches = 1 the two agree up to the "
                             "gaps between launches. single_frame is the same kernel with nothing else running. HIP events bracket every launch at "
                             "N = 1 and every 7th launch on the gather path (where launches of consecutive frames overlap)"},
    }
    if compact:
        # what the exchange moves per frame: every rank sends ceil(tiles of the view's screen rectangle / N) tiles of RGBA8 to the frame's owner;
        # `whole_frame` = the same with every tile of the frame (rounds 1-5, --tile-rect off)
        per_rank = float(np.mean([exchange_bytes_v[vidx(k)] for k in range(args.steps)]))
        out["exchange_bytes_per_frame"] = {"per_rank": int(per_rank), "all_ranks": int(per_rank * n_ranks), "whole_frame_all_ranks": int(exchange_bytes_whole * n_ranks),
                                           "ratio": round(per_rank / exchange_bytes_whole, 4), "tile_rect": "on" if use_rect else "off",
                                           "frame_owner": ("spread: frame j of launch l on rank (l + j) mod N" if spread else ("rotate: launch l on rank l mod N" if rotate else "rank 0")) if use_gather else None,
                                           "tiles_of_rect_per_view": [int(r.w * r.h) for r in rects], "tiles_of_frame": int(whole_rect.w * whole_rect.h)}
    if any(phase_ms.values()):
        # HIP-event times of the launches rank 0 owned, from the render's end: gather_ms includes waiting for the slowest rank's render (torch
        # exchange: until the block has arrived; then scatter_ms = the de-interleave kernels), exchange_ms = both (native: one call)
        out["phases"] = {k: round(float(np.mean(x)), 4) for k, x in phase_ms.items() if x}
        out["phases"].update(launches_sampled=len(phase_ms["render_ms"]), rank=0, frames_per_launch=out["roofline"]["frames_per_launch"])
    if single is not None:
        out["single_frame"] = single
    if depth_ms is not None:
        out["ms_per_step_with_depth"] = round(depth_ms, 4)

    # HBM traffic cannot be read from inside the process; it comes from separate rocprofv3 --pmc passes over this same
    # command (FETCH_SIZE and WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes and calibrated on the integrator's own gather
    # pattern, tools/micro/gather_fetch.hip), committed under profiles/.  The file names the sources it was measured on (a digest of the
    # integrator's source files + the commit): a tree whose integrator differs gets "traffic": null instead of a stale figure.
    try:
        with open(traffic_file(args.workload)) as f:
            tr = json.load(f)
        out["roofline"]["traffic_commit"] = tr.get("commit")
        if (tr.get("workload") == args.workload and world == 1 and tr.get("kernel") == out["roofline"]["kernel"] and args.tf == "app"
                and args.skip is None and not args.no_ert and not moving and not args.no_feedback and not virtual):
            if tr.get("kernel_source_sha256") == kernel_source_digest():
                # measured with 8 frames per launch: scaled to this run's average launch
                out["roofline"]["traffic"] = int(tr["traffic_bytes_per_launch"] * out["roofline"]["frames_per_launch"] / tr.get("frames_per_launch", 8))
                out["roofline"]["traffic_source"] = tr["source"]
                out["roofline"]["traffic_over_algorithmic"] = round(out["roofline"]["traffic"] / max(1, out["roofline"]["algorithmic_bytes_per_launch"]), 3)
                # the binding limit (same PMC passes, DESIGN.md section 6 has the formulas): valu_issue_frac = the launch's VALU wave-instructions
                # by SQ counter class x the issue cost measured for the opcodes of that class on gfx950 / the SIMD cycles of the launch;
                # wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of a resident wave's time parked in s_waitcnt); waves_per_simd = resident waves
                for k in ("valu_issue_frac", "wait_frac", "waves_per_simd", "valu_active_lanes"):
                    if tr.get(k) is not None:
                        out["roofline"][k] = tr[k]
            else:
                out["roofline"]["traffic_source"] = "withheld: %s was measured on other integrator sources than this tree's" % os.path.basename(traffic_file(args.workload))
    except (OSError, ValueError, KeyError):
        pass

    if virtual:
        out["virtual_rank"] = {"rank": virtual[0], "of": virtual[1], "tiles_per_view": [int(t.tile_count) for t in tiles_v], "rays_per_frame": rays_per_frame_all,
                               "note": "one GPU rendering the tile share of rank %d of %d (compact schedule, no exchange): value and ms_per_step are this share's" % virtual}
    if world == 1 and not args.no_cpu_baseline and not virtual:
        out["cpu_baseline"] = cpu_baseline(ctx, sp, v, params_whole, (fw, fh), args.cpu_seconds, not args.no_verify_cpu, out,
                                           timed_params=params if use_fill else None)
    if native:
        gather.close()
    return out


def verify(ctx, sp, v, view, params, steps, nbuf, fpl, frame, bufs, images, gather, rank, owner):
    """The frame left in the last step's buffer (on the rank that owns it) must equal a direct single-launch render of the same
    view, bit for bit."""
    if rank != owner:
        return
    k = steps - 1
    fw, fh = frame
    direct = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
    sp.draw(sp.make_params(*view, abi.full_frame_tiles(fw, fh, TILE, TILE)), rgba8=direct)
    torch.cuda.synchronize()
    slot = (fpl - 1) if fpl else (k % nbuf)  # batch submission: `fpl` carries the last step's output buffer index + 1
    got = (images[slot] if fpl else images[k % nbuf]) if gather else bufs[slot].view(fh, fw, 4)
    if not torch.equal(got, direct):
        raise SystemExit("verify failed: assembled frame differs from the direct render in %d bytes" % int((got != direct).sum().item()))
    print("verify ok (rank %d): frame of step %d matches the direct render (%d non-zero bytes)" % (rank, k, int((direct != 0).sum().item())),
          file=sys.stderr)


def cpu_quota_cores():
    """CPUs' worth of run time per period the cgroup of this process may use (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us); None = no quota"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else round(float(q) / float(p), 2)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else round(q / p, 2)
    except (OSError, ValueError):
        return None


def cpu_baseline(ctx, sp, v, params, frame, target_seconds, verify_cpu, out, timed_params=None):
    """Time the CPU oracle (scalar port of the reference shaders, all host cores) on every s-th pixel in x and y of the
    same 8 frames.  The oracle is only the baseline being timed here; nothing it computes feeds the GPU result.  With
    --verify-cpu the pixels it rendered are then compared with the device's frames (the oracle as the checker)."""
    from oracle import vkv_oracle as O
    cores = os.cpu_count() or 1
    vol, grad = v.volume.cpu().numpy(), v.gradient.cpu().numpy()
    tex = v.transfer_function.cpu().numpy()
    maps = [m.cpu().numpy() for m in v.distance_maps]
    # calibrate on a sparse sample of view 0 (run twice: the first call pays page faults and thread start-up), pick the
    # pixel stride so that ONE pass over the 8 views fits the budget, then repeat passes until the budget is used
    O.render(params[0], vol, grad, tex, maps, n_threads=cores, pixel_stride=16)
    r = O.render(params[0], vol, grad, tex, maps, n_threads=cores, pixel_stride=8)
    rate = r.rays / max(r.seconds, 1e-6)
    want = rate * target_seconds / N_VIEWS
    stride = max(1, int(math.ceil(math.sqrt(frame[0] * frame[1] / max(want, 1.0)))))
    # one output set per view, allocated and touched BEFORE the timed passes (an untimed pass: first-touch page faults of 70 MB per view,
    # the pool's threads created); the timed figure is the time inside vkvo_render alone, summed over the calls
    last = [O.render(p, vol, grad, tex, maps, n_threads=cores, pixel_stride=max(stride, 4), want_rgba8=verify_cpu) for p in params]
    rays, passes, dt = 0, 0, 0.0
    while True:
        last = [O.render(p, vol, grad, tex, maps, n_threads=cores, pixel_stride=stride, want_rgba8=verify_cpu, reuse=last[i]) for i, p in enumerate(params)]
        rays += sum(r.rays for r in last)
        dt += sum(r.seconds for r in last)
        passes += 1
        if dt >= target_seconds or passes >= 3:  # a pass renders every sampled pixel of all 8 views: more than three only repeat it
            break
    if verify_cpu:
        fw, fh = frame
        counts = torch.zeros((fh, fw, 3), dtype=torch.int32, device="cuda")
        rgba8 = torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda")
        sel = (slice(0, fh, stride), slice(0, fw, stride))
        n = 0
        for i, p in enumerate(params):
            sp.draw(p, rgba8=rgba8, counts=counts)
            torch.cuda.synchronize()
            if not np.array_equal(counts.cpu().numpy().astype(np.uint32)[sel], last[i].counts[sel]):
                raise SystemExit("verify-cpu failed: view %d: the frag counters differ from the oracle's" % i)
            if not np.array_equal(rgba8.cpu().numpy()[sel], last[i].rgba8[sel]):
                raise SystemExit("verify-cpu failed: view %d: RGBA8 differs from the oracle's" % i)
            # the timed steps ask for RGBA8 only, which selects the integrator without the per-pixel counters: its frame is checked as well
            rgba8.zero_()
            sp.draw(p, rgba8=rgba8)
            torch.cuda.synchronize()
            if not np.array_equal(rgba8.cpu().numpy()[sel], last[i].rgba8[sel]):
                raise SystemExit("verify-cpu failed: view %d: RGBA8 of the launch without counters differs from the oracle's" % i)
            if timed_params is not None:
                # ... and exactly what the timed block submits: the view's fill_outside schedule (the tiles of its screen rectangle, the rest filled by the
                # rendering workgroups) in a vkv_render_batch launch, into a buffer full of garbage - every pixel of the frame, against the oracle
                a, b = torch.full_like(rgba8, 0x5A), torch.full_like(rgba8, 0xA5)
                qs = []
                for t in (a, b):
                    q = abi.RenderParams.from_buffer_copy(timed_params[i])
                    q.d_out_rgba8, q.d_out_color, q.d_out_counts, q.d_out_depth, q.d_in_depth, q.blend_over_target = t.data_ptr(), None, None, None, None, 0
                    qs.append(q)
                ctx.render_batch(qs, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                if not (np.array_equal(a.cpu().numpy()[sel], last[i].rgba8[sel]) and torch.equal(a, b)):
                    raise SystemExit("verify-cpu failed: view %d: RGBA8 of the batch launch through the view's tile rectangle (fill_outside) differs from the oracle's" % i)
            n += last[i].counts[sel].shape[0] * last[i].counts[sel].shape[1]
        out["verified_against_cpu"] = {"views": len(params), "pixels": n, "pixel_stride": stride,
                                       "what": "3 counters + RGBA8 per pixel, bit-exact; RGBA8 again from the launch without counters and from the timed configuration (vkv_render_batch through each view's tile rectangle with fill_outside, into a buffer of garbage)"}
        print("verify-cpu ok: %d pixels of %d views match the oracle (counters + RGBA8)" % (n, len(params)), file=sys.stderr)
    # the same port on ONE thread (SURVEY.md §8d asks for both figures): view 0 on a sparser sample, about two seconds
    s1 = max(stride, int(math.ceil(math.sqrt(frame[0] * frame[1] / max(rate / cores * 2.0, 1.0)))))
    r1 = O.render(params[0], vol, grad, tex, maps, n_threads=1, pixel_stride=s1)
    dt1 = max(r1.seconds, 1e-9)
    quota = cpu_quota_cores()
    best = max(r.rays / max(r.seconds, 1e-9) for r in last) / 1e6  # the fastest single call of the last pass: what the pool delivers while the quota lasts
    # `cores` = the CPUs the figure can actually use: the cgroup's quota when one applies (host_cpus = what the container sees, threads = what ran)
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mray/s", "cores": (min(cores, int(math.ceil(quota))) if quota else cores), "host_cpus": cores, "threads": cores, "kind": "port",
            "value_1_thread": round(r1.rays / dt1 / 1e6, 5),
            # a container may see every CPU of the host and still be held to a CFS quota (cpu.max): the sustained figure is then the quota's, whatever
            # the thread count (the GPU boxes of this pool: 256 CPUs visible, quota 16 - a 2 M-ray view takes 6 ms when it fits a 100 ms period's
            # allowance and 98 ms when it does not; tools/cpu_scaling.py shows both)
            "cpu_quota_cores": quota, "value_fastest_call": round(best, 3),
            "sample": "oracle/vkv_oracle.c (scalar C port of the shaders; a persistent pthread pool claims 4-row pieces of 16x16 tiles from an "
                      "atomic counter), every %d-th pixel in x and y of the same 8 frames, %d pass(es): %d rays in %.2f s inside vkvo_render "
                      "(outputs allocated and touched before)" % (stride, passes, rays, dt)}


if __name__ == "__main__":
    main()
