"""The C oracle's distance transforms against an independent literal transliteration of the reference's compute shaders and dispatch
schedule (tests/golden/comp_literal.py, plain Python from the GLSL / C++ text, no code shared with oracle/vkv_oracle.c): every cell of
every map, for 0 / 255 occupancy maps of random density AND for arbitrary byte inputs (the shaders are defined for any bytes; the
product's kernels follow them there, tests/test_gpu_parity.py::test_distance_transforms_on_arbitrary_bytes)."""
import os

import numpy as np
import pytest

from oracle import vkv_oracle as O
from tests.golden import comp_literal as L


def make_input(seed):
    rng = np.random.default_rng(4400 + seed)
    shape = tuple(int(x) for x in rng.integers(1, 14, size=3))        # d, h, w
    if seed % 5 == 4:
        shape = tuple(int(x) for x in rng.permutation([1, int(rng.integers(2, 6)), int(rng.integers(20, 40))]))        # a long line
    cells = shape[0] * shape[1] * shape[2]
    p = float(min(1.0, rng.choice([0.0, 1.0, 2.0, 10.0]) / cells + rng.choice([0.0, 0.01, 0.1, 0.5])))
    a = np.where(rng.random(shape) < p, 0, 255).astype(np.uint8)
    if seed % 3 == 2:        # arbitrary bytes
        raw = rng.integers(0, 256, size=shape, dtype=np.uint8)
        a = np.where(rng.random(shape) < 0.4, raw, a).astype(np.uint8)
    return a


N_SEEDS = int(os.environ.get("VKV_TEST_FUZZ_SEEDS", "40"))        # a soak sets more (profiles/r4_literal_soak.txt)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_isotropic_transform_equals_the_shader_transliteration(seed):
    a = make_input(seed)
    assert np.array_equal(O.distance_map(a), L.distance_map(a)), "shape %s" % (a.shape,)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_anisotropic_transform_equals_the_shader_transliteration(seed):
    a = make_input(1000003 + seed)
    got, want = O.distance_map_anisotropic(a), L.distance_map_anisotropic(a)
    for k in range(8):
        assert np.array_equal(got[k], want[k]), "map %d, shape %s" % (k, a.shape)
