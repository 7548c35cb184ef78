"""PREDICTED strong-scaling curve of BASELINE.json configs[4] (c5: 2048^3, anisotropic maps, the fixed 7680x4320 frame dealt over the ranks in
16x16 tiles) from ONE GPU: for N = 2, 4, 8 the tile share of every rank r is rendered alone (`bench.py --workload c5 --virtual-rank r/N`,
the bench's own submission: 6 frames per launch, 4 streams, no exchange), and

    predicted_speedup(N) = t(N = 1) / (max_r t(r/N) + exchange),     exchange = bytes one rank sends per frame / 153 GB/s (one xGMI link)

The exchange term is an upper bound of what it adds (in the bench it overlaps the next launches' renders).  This cannot replace a run on N
devices - it knows nothing of link contention, of RCCL's launch costs or of the owner's de-interleave - it bounds the render side: if max_r is
already above t(1) / 6 at N = 8, the >= 6x target is missed before a byte has moved.
Round 6: a rank renders - and sends - only its share of the tiles of each frame's screen rectangle (vkv_screen_tile_rect); `off` as the third
argument schedules every tile of the frame as rounds 1-5 did.
usage: python tools/virtual_ranks.py [workload] [steps] [on|off]          (writes a table to stdout; profiles/r6_virtual_ranks.txt keeps one)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload = sys.argv[1] if len(sys.argv) > 1 else "c5"
steps = sys.argv[2] if len(sys.argv) > 2 else "16"
rect_flag = ["--tile-rect", sys.argv[3]] if len(sys.argv) > 3 else []        # "off": every tile of the frame scheduled and exchanged (rounds 1-5)
LINK_GBS = 153.0


def run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", steps, "--warmup", "4", "--min-seconds", "1.0", "--no-cpu-baseline",
           "--no-depth-block", "--extras", "off", "--c5-block", "off"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
    if r.returncode != 0:
        raise SystemExit("bench.py failed: %s" % r.stderr[-2000:])
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


one = run([])
t1 = one["ms_per_step"]
print("# PREDICTED strong scaling of %s from one GPU (tools/virtual_ranks.py): each rank's tile share rendered alone, no exchange" % workload)
print("# %s" % one["config"]["workload"])
print("N = 1: %.4f ms per frame (%.1f Mray/s, frac %.3f)" % (t1, one["value"], one["roofline"]["frac"]))
for n in (2, 4, 8):
    ts, sent, whole = [], 0, 0
    for r in range(n):
        d = run(["--virtual-rank", "%d/%d" % (r, n)] + rect_flag)
        ts.append(d["ms_per_step"])
        # bytes a rank sends per frame: ceil(tiles of the frame's screen rectangle / N) tiles of RGBA8 (bench.py: exchange_bytes_per_frame)
        sent = max(sent, d["exchange_bytes_per_frame"]["per_rank"])
        whole = d["exchange_bytes_per_frame"]["whole_frame_all_ranks"] / n
    exch_ms = sent / (LINK_GBS * 1e9) * 1e3
    worst = max(ts)
    print("N = %d: per-rank ms per frame %s   max %.4f  mean %.4f  (max / mean %.3f)   exchange %.1f MB per rank (%.2f of the whole frame's %.1f MB) = %.4f ms on one link" % (
        n, " ".join("%.4f" % t for t in ts), worst, sum(ts) / n, worst / (sum(ts) / n), sent / 1e6, sent / whole, whole / 1e6, exch_ms))
    print("        predicted speedup: render only %.2fx (efficiency %.2f), with the exchange serialised behind it %.2fx" % (
        t1 / worst, t1 / worst / n, t1 / (worst + exch_ms)))
