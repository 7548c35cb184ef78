"""The C oracle's integrator against an independent literal transliteration of shaders/volume_render.frag:117-336
(tests/golden/frag_literal.py, numpy, written from the GLSL, no code shared with oracle/vkv_oracle.c): VERDICT round 2, next #7.

Per configuration (4 skipping types x early ray termination on / off x sampling factor 0.5 / 1 / 2) a small frame is marched by both:
  * with the build's arithmetic pins ("build": one fma for `ray_entry + float(i) * step_volume`, fma-lerp filter, fma blend) the three
    frag counters of EVERY pixel must equal the oracle's - the loop structure, the probe test, both skip formulas, the step back with
    i_min, the u_last_alpha bookkeeping, the counters and early ray termination are then the same in two independent statements;
  * with plain float32 arithmetic ("plain": multiply then add, the Vulkan specification's weighted-sum filter) the counters may differ
    only where a position lands within an ulp of a cell or texel boundary: every pixel that differs must be one that the "build"
    arithmetic reproduces, and they must be few: < 8 % of the covered pixels (a ray has 20 - 200 events, each one chance in a few thousand
    to sit on a boundary) - < 20 % with BLOCK skipping, whose skip length is by
    construction the distance to the exit face of the current cell (frag:239), so every skip lands ON a cell boundary and the cell the
    next position falls into hangs on the last bit of `ray_entry + float(i) * step_volume`.  The differing pixels are printed (pytest -s)."""
import numpy as np
import pytest

from oracle import vkv_oracle as O
from tests import helpers as T
from tests.golden import frag_literal as F
from vkvolume_amd import abi

import os

SHAPE = (36, 30, 26)  # W, H, D: no multiples of the block size
IMAGE = (32, 32) if os.environ.get("VKV_TEST_EXHAUSTIVE") else (16, 16)  # (the literal is pure Python: ~0.1 s per covered pixel and variant)


def scene_for(sf, on_the_fly=False):
    opt = abi.VolumeOptions(sampling_factor=sf, use_precomputed_gradient=not on_the_fly, **T.APP_TF)
    vol = O.synth_volume(SHAPE, 1, 0xC0FFEE02)
    return T.OracleScene(vol, opt, 4, voxel_size=(0.0003, 0.0003, 0.0007), axis_angle=(1, 0, 0, 90))


def literal_counts(scene, p, entry, pins):
    maps = None if p.options.skipping_type == abi.SKIP_NONE else list(scene.maps(p.options.skipping_type))
    U = F.Uniforms(list(p.ray_cast.camera_pos_tex), list(p.ray_cast.block_size), p.transfer_function.sampling_factor, p.transfer_function.voxel_alpha_factor,
                   p.transfer_function.grad_magnitude_modifier, p.transfer_function.use_gradient, scene.vol, scene.grad, scene.tex, maps,
                   p.options.skipping_type, bool(p.options.early_ray_termination), precomputed_gradient=bool(p.use_precomputed_gradient))
    h, w = entry.shape[:2]
    out = np.zeros((h, w, 3), np.uint32)
    for y in range(h):
        for x in range(w):
            if entry[y, x, 3] > 0:  # covered pixels: the RayEntry test output has alpha 1
                out[y, x] = F.frag_main(entry[y, x, :3], U, pins)[:3]
    return out


@pytest.mark.parametrize("sf", [0.5, 1.0, 2.0])
@pytest.mark.parametrize("ert", [True, False])
@pytest.mark.parametrize("mode", [abi.SKIP_NONE, abi.SKIP_BLOCK, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
def test_oracle_counters_equal_the_literal_transliteration(mode, ert, sf, capsys):
    scene = scene_for(sf)
    view, proj = T.orbit(33.0 + 40.0 * mode, image_size=IMAGE)
    ro = abi.RenderOptions(skipping_type=mode, clip_distance=1.0, early_ray_termination=ert)
    p = scene.params(view, proj, IMAGE, ro)
    ref = scene.render(p)
    pe = scene.params(view, proj, IMAGE, abi.RenderOptions(skipping_type=mode, clip_distance=1.0, early_ray_termination=ert, test=abi.TEST_RAY_ENTRY))
    entry = scene.render(pe).color  # rgb = ray_entry, a = 1 where the pixel has a fragment
    assert ref.counts[..., 0].sum() > 100
    build = literal_counts(scene, p, entry, "build")
    bad = np.argwhere((build != ref.counts).any(-1))
    assert len(bad) == 0, "build-pinned literal differs from the oracle at pixels %r: literal %r oracle %r" % (
        bad[:5].tolist(), [build[y, x].tolist() for y, x in bad[:5]], [ref.counts[y, x].tolist() for y, x in bad[:5]])
    plain = literal_counts(scene, p, entry, "plain")
    diff = np.argwhere((plain != ref.counts).any(-1))
    covered = int((entry[..., 3] > 0).sum())
    with capsys.disabled():
        if len(diff):
            print("\n  mode %d ert %d sf %g: %d of %d covered pixels differ under plain float32 arithmetic (an ulp at a cell / texel boundary): %s" % (
                mode, ert, sf, len(diff), covered, [(int(y), int(x), plain[y, x].tolist(), ref.counts[y, x].tolist()) for y, x in diff[:3]]))
    limit = 0.20 if mode == abi.SKIP_BLOCK else 0.08
    assert len(diff) <= max(2, limit * covered), "%d of %d pixels differ" % (len(diff), covered)


def test_literal_on_the_fly_gradient_variant():
    """the #ifndef PRECOMPUTED_GRADIENT branch of get_gradient (frag:92-97), Chebyshev skipping"""
    scene = scene_for(1.0, on_the_fly=True)
    view, proj = T.orbit(200.0, image_size=(12, 12))
    ro = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0, early_ray_termination=True)
    p = scene.params(view, proj, (12, 12), ro)
    ref = scene.render(p)
    pe = scene.params(view, proj, (12, 12), abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0, test=abi.TEST_RAY_ENTRY))
    entry = scene.render(pe).color
    plain = literal_counts(scene, p, entry, "plain")
    covered = int((entry[..., 3] > 0).sum())
    assert ref.counts[..., 0].sum() > 100
    assert int((plain != ref.counts).any(-1).sum()) <= max(2, 0.06 * covered)
