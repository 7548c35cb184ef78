#!/usr/bin/env python3
"""Where empty-space skipping pays on MI355X: frame rate of the four skipping modes against the occupied share of the volume (GPU).

One extent (the `present` preset's 492x492x442, intensity-only transfer function imin 0.071 as in scripts/benchmark.py:28), the seed of
the sweep's scene.  Two families: CLUSTERED occupancy (the generator's shell count / thickness knobs turned from ~0.6 % to ~50 % occupied voxels)
and SCATTERED occupancy (hash noise above the transfer function's threshold: nearly every 4^3 cell holds an occupied voxel); for each scene the
reference's benchmark mode (`vkv_offscreen --benchmark`: 1200x1200, ERT off, sample-count output, src/volume_render.cpp:177-183) and the
same with early ray termination on, block size 4, one frame per launch (--frames-in-flight 1: what the reference's loop measures) and 8.
Prints one table per (ERT, frames in flight); INTEGRATION.md section 6 carries a run (profiles/r6_ess_crossover.txt).
usage: python tools/ess_crossover.py [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.benchmark_sweep import PRESETS, run  # noqa: E402

PRESET = PRESETS[0]
NOISE = 15 << 28  # hash noise 0..15: below the transfer function's threshold (imin 0.071 = 18.1 of 255)
# family 1, CLUSTERED occupancy - (shells, thickness / 256, noise word, imin): thin few shells ... thick many shells, noise below the threshold
SCENES = [(4, 200, NOISE, 0.071), (10, 230, NOISE, 0.071), (24, 256, NOISE, 0.071), (40, 319, NOISE, 0.071), (40, 600, NOISE, 0.071), (40, 1100, NOISE, 0.071),
          (40, 2000, NOISE, 0.071), (40, 3600, NOISE, 0.071)]
# family 2, SCATTERED occupancy - four thin shells under hash noise 0..20 of which the transfer function's threshold lets 1, 2, 4 or 8 levels through:
# single occupied voxels everywhere (round 5's first sweep scene was of this kind: 2 of 21 levels, 9.5 % of the voxels, 99.8 % of the 4^3 cells)
SCATTERED = [(4, 200, 0, 0.0765), (4, 200, 0, 0.071), (4, 200, 0, 0.0648), (4, 200, 0, 0.0491)]


def occupied_cells_percent(ctx, extent, kind, seed, imin, block):
    """share of the occupancy map's cells that hold an occupied voxel (texture-path transfer function, shaders/occupancy_map.comp)"""
    import torch
    from vkvolume_amd import abi, volume as V
    v = V.Volume(ctx)
    v.options = abi.VolumeOptions(intensity_min=imin, intensity_max=PRESET[3], gradient_min=PRESET[4], gradient_max=PRESET[5])
    v.load_synthetic(extent, kind=kind, seed=seed, distance_map_block_size=block)
    tf = v.get_transfer_function_uniform()
    V.ComputeGradientMap(ctx).compute(v, tf)
    v.update_transfer_function_texture()
    V.ComputeDistanceMap(ctx).compute(v, tf, abi.SKIP_BLOCK)
    torch.cuda.synchronize()
    return 100.0 * float((v.distance_maps[0] == 0).float().mean().item())


def main():
    import torch
    from tools.benchmark_sweep import preset_seed
    from vkvolume_amd import lib
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    torch.cuda.set_device(0)
    ctx = lib.Context(0)
    for family, scenes in (("clustered occupancy (closed shells, noise below the transfer function's threshold)", SCENES),
                           ("scattered occupancy (hash noise above the threshold: single voxels everywhere)", SCATTERED)):
        for ert in (0, 1):
            for fif in (1, 8):
                print("# %s; early ray termination %s, %d frame(s) per launch: frames per second (block 4, 1200x1200, %d frames)" % (
                    family, "on" if ert else "off (the reference's benchmark mode)", fif, frames))
                print("# %-10s %-10s %10s %10s %10s %10s   %s" % ("voxels %", "cells %", "no ESS", "block", "Chebyshev", "anisotropic", "best ESS / no ESS"))
                for shells, tq, noise, imin in scenes:
                    kind = 1 | (shells << 8) | (tq << 16) | noise
                    preset = (PRESET[0], PRESET[1], imin, PRESET[3], PRESET[4], PRESET[5])
                    cells = occupied_cells_percent(ctx, PRESET[1], kind, preset_seed(PRESET[1]), imin, 4)
                    fps, occ = [], None
                    for mode in (0, 1, 2, 3):
                        f, _, occ = run(preset, 4, mode, frames, None, fif, kind=kind, extra=["--ert=%d" % ert])
                        fps.append(f)
                    print("  %-10.3f %-10.2f %10.1f %10.1f %10.1f %10.1f   %.2f" % (occ, cells, fps[0], fps[1], fps[2], fps[3], max(fps[1:]) / fps[0]), flush=True)
                print()


if __name__ == "__main__":
    main()
