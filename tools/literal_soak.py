"""Soak (CPU): the C oracle's integrator against the literal numpy transliteration of shaders/volume_render.frag (tests/golden/frag_literal.py)
on RANDOM configurations - volume shape and content, block size, transfer-function window, sampling and alpha factors, skipping type, early ray
termination, gradient variant, camera - where tests/test_frag_literal_cpu.py marches 24 fixed ones.  With the build's arithmetic pins the three
frag counters of every covered pixel must be equal; the premultiplied float colour of the two statements is compared as well (the literal
evaluates the opacity correction with numpy's float32 pow per sample, the oracle through its 256-entry table: reported, not asserted).  One process per worker; prints one line per configuration and a summary.
usage: literal_soak.py <first seed> <count> [workers]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(seed):
    from oracle import vkv_oracle as O
    from tests import helpers as T
    from tests.golden import frag_literal as F
    from tests.test_frag_literal_cpu import literal_counts
    from vkvolume_amd import abi
    rng = np.random.default_rng(31000 + seed)
    shape = tuple(int(x) for x in rng.integers(9, 44, size=3))
    kind = int(rng.integers(0, 3))
    vol = T.random_volume(shape, seed=seed, sparsity=float(rng.uniform(0.7, 0.99))) if kind == 2 else O.synth_volume(shape, kind, int(rng.integers(1, 1 << 30)))
    variant = ("precomputed", "on_the_fly", "no_gradient")[int(rng.integers(0, 3))]
    imin = float(rng.uniform(0.0, 0.4))
    tfo = dict(intensity_min=imin, intensity_max=float(rng.uniform(imin + 0.05, 1.0)), sampling_factor=float(rng.choice([0.5, 1.0, 1.0, 1.7, 3.0])),
               voxel_alpha_factor=float(rng.choice([0.3, 1.0, 1.0, 2.5])))
    if variant == "no_gradient":
        opt = abi.VolumeOptions(gradient_min=0.0, gradient_max=0.0, **tfo)
    else:
        gmin = float(rng.uniform(0.0, 0.1))
        opt = abi.VolumeOptions(gradient_min=gmin, gradient_max=float(rng.uniform(gmin + 0.05, 0.6)), use_precomputed_gradient=(variant == "precomputed"), **tfo)
    block = int(rng.integers(1, 7))
    voxel = tuple(float(x) for x in rng.uniform(0.3, 2.0, size=3))
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    scene = T.OracleScene(vol, opt, block, voxel_size=voxel, axis_angle=(float(axis[0]), float(axis[1]), float(axis[2]), float(rng.uniform(0, 360))))
    mode, ert = int(rng.integers(0, 4)), bool(rng.integers(0, 2))
    size = (int(rng.integers(10, 22)), int(rng.integers(10, 22)))
    view, proj = T.orbit(float(rng.uniform(0, 360)), elevation=float(rng.uniform(-70, 70)), radius=float(rng.choice([60.0, 110.0, 150.0, 260.0])),
                         fov=float(rng.uniform(25, 90)), image_size=size)
    ro = abi.RenderOptions(skipping_type=mode, clip_distance=float(rng.choice([0.1, 1.0, 1.0, 20.0])), early_ray_termination=ert)
    p = scene.params(view, proj, size, ro)
    ref = scene.render(p)
    pe = scene.params(view, proj, size, abi.RenderOptions(skipping_type=mode, clip_distance=ro.clip_distance, early_ray_termination=ert, test=abi.TEST_RAY_ENTRY))
    entry = scene.render(pe).color
    t = time.time()
    maps = None if mode == abi.SKIP_NONE else list(scene.maps(mode))
    U = F.Uniforms(list(p.ray_cast.camera_pos_tex), list(p.ray_cast.block_size), p.transfer_function.sampling_factor, p.transfer_function.voxel_alpha_factor,
                   p.transfer_function.grad_magnitude_modifier, p.transfer_function.use_gradient, scene.vol, scene.grad, scene.tex, maps,
                   mode, ert, precomputed_gradient=bool(p.use_precomputed_gradient))
    h, w = entry.shape[:2]
    build, colour = np.zeros((h, w, 3), np.uint32), np.zeros((h, w, 4), np.float32)
    for y in range(h):
        for x in range(w):
            if entry[y, x, 3] > 0:
                r = F.frag_main(entry[y, x, :3], U, "build")
                build[y, x], colour[y, x] = r[:3], r[3]
    bad = int((build != ref.counts).any(-1).sum())
    covered = int((entry[..., 3] > 0).sum())
    cdiff = float(np.abs(colour - ref.color)[entry[..., 3] > 0].max()) if covered else 0.0
    return "seed %d: shape %s block %d mode %d ert %d sf %g %s frame %s: %d covered pixels, %d events, %d pixels differ, max colour difference %.2e (%.0f s)" % (
        seed, shape, block, mode, ert, tfo["sampling_factor"], variant, size, covered, int(ref.counts[..., :2].sum()), bad, cdiff, time.time() - t), bad, covered, cdiff


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(workers) as pool:
        bad_total = pix = 0
        worst = 0.0
        for line, bad, covered, cdiff in pool.imap_unordered(one, range(first, first + count)):
            print(line, flush=True)
            bad_total += bad
            pix += covered
            worst = max(worst, cdiff)
    print("literal soak: %d configurations, %d covered pixels, %d pixels with different counters, largest colour difference %.3e" % (count, pix, bad_total, worst))
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
