#!/usr/bin/env python3
"""Harness counterpart of the reference's scripts/benchmark.py, written for this build.

Sweeps skipmode {0,1,2,3} x blocksize {2..6} x the reference's six dataset/TF presets and writes one CSV per skip mode
with the reference's columns (image, skipmode, blocksize, occupancy, framerate, update, imin, imax, gmin, gmax) so that
MI355X results sit next to scripts/benchmark_results_{0..3}.csv.  It drives `vkv_offscreen --benchmark=N` and parses
the same three log lines the reference harness parses.

The reference's scans (present / stag beetle / kingsnake) are not published, so each preset uses a synthetic uint8
volume of the same extent (`--synthetic`); pass --assets DIR to use real `<name>` + `<name>.header` files instead.

    python tools/benchmark_sweep.py --out profiles/ [--frames 100] [--quick]
"""
import argparse
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "vkvolume_amd", "csrc", "vkv_offscreen")
WIDTH = HEIGHT = 1200  # scripts/benchmark.py:10-11

# scripts/benchmark.py:27-34 — (file name, extent, imin, imax, gmin, gmax)
PRESETS = [
    ("present_492x492x442.uint16", (492, 492, 442), 0.071, 1.0, 0.0, 0.0),
    ("present_492x492x442.uint16", (492, 492, 442), 0.071, 1.0, 0.06, 0.1),
    ("stag_beetle_832x832x494.uint16", (832, 832, 494), 0.086, 1.0, 0.0, 0.0),
    ("stag_beetle_832x832x494.uint16", (832, 832, 494), 0.086, 1.0, 0.1, 0.3),
    ("kingsnake_1024x1024x795.uint8", (1024, 1024, 795), 0.4, 0.8, 0.0, 0.0),
    ("kingsnake_1024x1024x795.uint8", (1024, 1024, 795), 0.2, 0.8, 0.06, 0.12),
]


def run(preset, blocksize, skipmode, frames, assets):
    name, extent, imin, imax, gmin, gmax = preset
    cmd = [APP, "--width=%d" % WIDTH, "--height=%d" % HEIGHT, "--benchmark=%d" % frames, "--imin=%g" % imin, "--imax=%g" % imax,
           "--gmin=%g" % gmin, "--gmax=%g" % gmax, "--blocksize=%d" % blocksize, "--skipmode=%d" % skipmode]
    if assets and os.path.exists(os.path.join(assets, name)):
        cmd.append(os.path.join(assets, name))
    else:
        cmd.append("--synthetic=%dx%dx%d:1:%d" % (*extent, 0xC0FFEE00 + extent[2] % 251))
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800).stdout.decode()
    fps = re.search(r"ran [\d]+ frames, averaged ([\d\.e\+\-]+) fps", out)
    upd = re.search(r"Updated occupancy/distance map in ([\d\.e\+\-]+)ms", out)
    occ = re.search(r"Occupied voxels: ([\d\.e\+\-]+)%", out)
    if not (fps and upd and occ):
        raise RuntimeError("unexpected output of %s:\n%s" % (" ".join(cmd), out))
    return float(fps.group(1)), float(upd.group(1)), float(occ.group(1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"))
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--assets", default=None)
    ap.add_argument("--quick", action="store_true", help="block sizes 2 and 4 only, first and last preset")
    args = ap.parse_args()
    presets = [PRESETS[0], PRESETS[-1]] if args.quick else PRESETS
    block_sizes = [2, 4] if args.quick else [2, 3, 4, 5, 6]
    for skipmode in (0, 1, 2, 3):
        rows = []
        for p in presets:
            for b in block_sizes:
                if skipmode == 0 and b != block_sizes[0]:
                    rows.append(dict(rows[-1], blocksize=b))  # the reference repeats the mode-0 row (benchmark.py:71)
                    continue
                fps, upd, occ = run(p, b, skipmode, args.frames, args.assets)
                rows.append(dict(image=p[0], skipmode=skipmode, blocksize=b, occupancy=occ, framerate=fps, update=upd,
                                 imin=p[2], imax=p[3], gmin=p[4], gmax=p[5]))
                print("\t", p[0], skipmode, b, fps, upd, occ, flush=True)
        path = os.path.join(args.out, "benchmark_results_%d_mi355x_synthetic.csv" % skipmode)
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["image", "skipmode", "blocksize", "occupancy", "framerate", "update", "imin", "imax", "gmin", "gmax"])
            w.writeheader()
            w.writerows(rows)
        print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
