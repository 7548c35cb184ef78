# host-side cost of the multi-rank step loop at 1 rank (--force-gather equivalent)
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch.distributed as dist
import bench
from vkvolume_amd import abi, lib, multigpu, volume as V
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=0, world_size=1, **({"device_id": torch.device("cuda", 0)} if os.environ.get("HP_DEVICE_ID") else {}))
torch.cuda.set_device(0)
ctx = lib.Context(0)
v, tf, frame, skip = bench.build_scene(ctx, "c3")
fw, fh = frame
views = bench.cameras(v, fw / fh)
opts = abi.RenderOptions(skipping_type=skip, clip_distance=1.0, early_ray_termination=True)
sp = V.VolumeRenderSubpass(ctx, v, opts, (fw, fh))
fif = 3
tiles = abi.full_frame_tiles(fw, fh, 16, 16, 0, 1, compact=True)
params = [sp.make_params(view, proj, tiles) for view, proj in views]
gather = multigpu.TileGather(dist, 0, 1, (fw, fh), 16, 4, device="cuda", n_buffers=fif)
bufs = gather.buffers
images = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(fif)]
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(fif - 1)]
acc = {"collect": 0.0, "draw": 0.0, "start": 0.0, "ctx": 0.0}
def collect(b):
    flat = gather.finish(b)
    if flat is not None:
        ctx.scatter_tiles(flat.data_ptr(), images[b].data_ptr(), (fw, fh), (16, 16), 1, gather.tiles_per_rank, 4, torch.cuda.current_stream().cuda_stream)
def run(n, variant):
    for k in range(n):
        b = k % fif
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[b]):
            t1 = time.perf_counter()
            if variant >= 2: collect(b)
            t2 = time.perf_counter()
            sp.draw(params[k % 8], rgba8=bufs[b])
            t3 = time.perf_counter()
            if variant >= 1: gather.start(b)
            t4 = time.perf_counter()
        acc["ctx"] += t1 - t0; acc["collect"] += t2 - t1; acc["draw"] += t3 - t2; acc["start"] += t4 - t3
    for b in range(fif):
        with torch.cuda.stream(streams[b]): collect(b)
side = torch.cuda.Stream()
done = [None] * fif
def run3(n):
    """renders never queue behind the assembly: wait / de-interleave live on a side stream, render streams wait on events only"""
    for k in range(n):
        b = k % fif
        with torch.cuda.stream(streams[b]):
            if done[b] is not None:
                streams[b].wait_event(done[b])      # frame k-fif's gather read bufs[b] and its scatter read flat[b]
            sp.draw(params[k % 8], rgba8=bufs[b])
            gather.start(b)
        with torch.cuda.stream(side):
            collect(b)
            done[b] = torch.cuda.Event(); done[b].record(side)
run3(32); torch.cuda.synchronize()
t = time.perf_counter(); run3(256); th = time.perf_counter() - t; torch.cuda.synchronize(); tt = time.perf_counter() - t
print("variant 3 (side stream) host loop %.1f us/step, total %.1f us/step" % (th / 256 * 1e6, tt / 256 * 1e6))
for nb in (4, 6, 9):
    g2 = multigpu.TileGather(dist, 0, 1, (fw, fh), 16, 4, device="cuda", n_buffers=nb)
    img2 = [torch.zeros((fh, fw, 4), dtype=torch.uint8, device="cuda") for _ in range(nb)]
    done2 = [None] * nb
    def run4(n):
        for k in range(n):
            b, s_ = k % nb, streams[k % fif]
            with torch.cuda.stream(s_):
                if done2[b] is not None:
                    s_.wait_event(done2[b])
                sp.draw(params[k % 8], rgba8=g2.buffers[b])
                g2.start(b)
            with torch.cuda.stream(side):
                flat = g2.finish(b)
                ctx.scatter_tiles(flat.data_ptr(), img2[b].data_ptr(), (fw, fh), (16, 16), 1, g2.tiles_per_rank, 4, side.cuda_stream)
                done2[b] = torch.cuda.Event(); done2[b].record(side)
    run4(32); torch.cuda.synchronize()
    t = time.perf_counter(); run4(256); th = time.perf_counter() - t; torch.cuda.synchronize(); tt = time.perf_counter() - t
    print("variant 4 (3 render streams, %d buffers, side stream) host loop %.1f us/step, total %.1f us/step" % (nb, th / 256 * 1e6, tt / 256 * 1e6))
for variant in (0, 1, 2):
    run(32, variant); torch.cuda.synchronize()
    for k_ in acc: acc[k_] = 0.0
    t = time.perf_counter(); run(256, variant); th = time.perf_counter() - t; torch.cuda.synchronize(); tt = time.perf_counter() - t
    print("variant", variant, "host loop %.1f us/step, total %.1f us/step" % (th / 256 * 1e6, tt / 256 * 1e6), {k_: round(v_ / 256 * 1e6, 1) for k_, v_ in acc.items()})
dist.destroy_process_group()
