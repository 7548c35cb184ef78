#!/usr/bin/env python3
"""Harness counterpart of the reference's scripts/benchmark.py, written for this build.

Sweeps skipmode {0,1,2,3} x blocksize {2..6} x the reference's six dataset/TF presets and writes one CSV per skip mode
with the reference's columns (image, skipmode, blocksize, occupancy, framerate, update, imin, imax, gmin, gmax) so that
MI355X results sit next to scripts/benchmark_results_{0..3}.csv.  It drives `vkv_offscreen --benchmark=N` and parses
the same three log lines the reference harness parses.

The reference's scans (present / stag beetle / kingsnake) are not published, so each preset uses a synthetic uint8
volume of the same extent (`--synthetic`) and the `image` column says so: `synthetic_<W>x<H>x<D>` - ellipsoid shells whose count,
thickness and noise floor are tuned per preset (SCENE_KIND, tools/tune_sweep_scenes.py) so that the `occupancy` column - the
analytic-TF voxel count of src/compute_occupied_voxel_count.cpp - matches the reference's row within 1 % (round 6; rounds 1-5 ran
the same seeds at 2 - 10 x the scans' occupied share).  The STRUCTURE of the scenes still differs from a CT scan's (thin closed shells
against a noise floor): same regime, not the same image.  Pass --assets DIR to use real `<name>` + `<name>.header` files instead (the
`image` column then carries the file name).

`--frames-in-flight N` (default 8): vkv_offscreen renders N frames per vkv_render_batch launch (the reference's swap-chain images in
flight), so `framerate` is batch throughput; `--frames-in-flight 1` is the frame-serial figure (one launch per frame, each waiting for
the one before: what the reference's benchmark loop measures).  The value is written to the `frames_in_flight` column and to the file
name (`..._fif<N>.csv`).

    python tools/benchmark_sweep.py --out profiles/ [--frames 100] [--quick] [--frames-in-flight 1]
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


# the `occupancy` column of the reference's rows for these presets (scripts/benchmark_results_0.csv:2, 7, 12, 17, 22, 27), in per cent
REFERENCE_OCCUPANCY = [7.13484, 1.84919, 3.96725, 1.31172, 0.669954, 0.547355]
# the generator knobs that bring each synthetic scene to that occupied share (tools/tune_sweep_scenes.py, within 10 % relative): vkv_synth_volume's
# kind = 1 | shells << 8 | thickness << 16 (the first `shells` of the seed's 40 ellipsoid shells, their thickness scaled by thickness / 256)
SCENE_KIND = [4047448065, 4040626177, 15800833, 16451841, 15667201, 16975617]  # -> 7.115 / 1.845 / 3.955 / 1.308 / 0.668 / 0.545 % (profiles/r6_tune_scenes.txt)


def preset_seed(extent):
    return 0xC0FFEE00 + extent[2] % 251


def image_label(preset, assets):
    """what the `image` column says: the scan's file name only when the real file was rendered"""
    name, extent = preset[0], preset[1]
    if assets and os.path.exists(os.path.join(assets, name)):
        return name
    return "synthetic_%dx%dx%d" % extent


def run(preset, blocksize, skipmode, frames, assets, frames_in_flight, kind=None, extra=()):
    """one vkv_offscreen --benchmark run -> (fps, map update ms, occupied voxels %); `kind`: generator word instead of the preset's"""
    name, extent, imin, imax, gmin, gmax = preset
    cmd = [APP, "--width=%d" % WIDTH, "--height=%d" % HEIGHT, "--benchmark=%d" % frames, "--imin=%g" % imin, "--imax=%g" % imax,
           "--gmin=%g" % gmin, "--gmax=%g" % gmax, "--blocksize=%d" % blocksize, "--skipmode=%d" % skipmode,
           "--frames-in-flight=%d" % frames_in_flight]
    if assets and os.path.exists(os.path.join(assets, name)):
        cmd.append(os.path.join(assets, name))
    else:
        cmd.append("--synthetic=%dx%dx%d:%d:%d" % (*extent, kind if kind is not None else SCENE_KIND[PRESETS.index(preset)], preset_seed(extent)))
    cmd += list(extra)
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
    ap.add_argument("--frames-in-flight", type=int, default=8, help="frames per vkv_render_batch launch (1 = frame-serial, like the reference's loop)")
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
                fps, upd, occ = run(p, b, skipmode, args.frames, args.assets, args.frames_in_flight)
                rows.append(dict(image=image_label(p, args.assets), skipmode=skipmode, blocksize=b, occupancy=occ, framerate=fps, update=upd,
                                 imin=p[2], imax=p[3], gmin=p[4], gmax=p[5], frames_in_flight=args.frames_in_flight))
                print("\t", image_label(p, args.assets), skipmode, b, fps, upd, occ, flush=True)
        path = os.path.join(args.out, "benchmark_results_%d_mi355x_synthetic_fif%d.csv" % (skipmode, args.frames_in_flight))
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["image", "skipmode", "blocksize", "occupancy", "framerate", "update", "imin", "imax", "gmin", "gmax",
                                              "frames_in_flight"])
            w.writeheader()
            w.writerows(rows)
        print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
