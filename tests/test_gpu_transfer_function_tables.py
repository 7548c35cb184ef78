"""GPU: transfer-function acceleration tables (the separable greyscale claim checked on the device) and the generic RGBA texture path."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import vkv_oracle as O
from tests import helpers as T
from tests.test_gpu_fullsize_oracle import build, orbit
from tests.test_gpu_parity import compare_render, gpu_render, make_gpu_volume
from vkvolume_amd import abi, lib, multigpu, volume as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "vkvolume_amd", "csrc", "vkv_offscreen")
FLAG_WORD, AI_WORD, AG_WORD = 2048, 2052, 2308


def tables_of(ctx, tex, tf):
    d_tex = torch.from_numpy(np.ascontiguousarray(tex)).cuda()
    d_tab = torch.full((abi.TF_BITS_WORDS,), -1, dtype=torch.int32, device="cuda")
    ctx.transfer_function_tables(d_tex.data_ptr(), tf, d_tab.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_tab.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("opts", [T.APP_TF, dict(intensity_min=0.4, intensity_max=0.8, gradient_min=0.0, gradient_max=0.0),
                                  dict(intensity_min=0.2, intensity_max=0.8, gradient_min=0.06, gradient_max=0.12), dict()])
def test_tf_tables_separable_claim_is_checked_on_the_device(ctx, opts):
    opt = abi.VolumeOptions(**opts)
    tf, tex = lib.transfer_function_uniform(opt), O.transfer_function_texture(opt)
    tab = tables_of(ctx, tex, tf)
    bits = np.unpackbits(tab[:2048].view(np.uint8), bitorder="little").reshape(256, 256)
    assert np.array_equal(bits, (tex[..., 3] > 0).astype(np.uint8)), "alpha > 0 bit table"
    assert tab[FLAG_WORD] & 1 == 1, "the reference's own texture is the separable product"
    ai, ag = tab[AI_WORD:AI_WORD + 256].view(np.float32), tab[AG_WORD:AG_WORD + 256].view(np.float32)
    prod = np.minimum((ai[None, :] * ag[:, None] * np.float32(255.0)).astype(np.uint32), 255)
    assert np.array_equal(prod, tex[..., 3]) and np.array_equal(tex[..., 0], tex[..., 3])
    # one texel off by one in one channel: the claim must be withdrawn, the bit table must still be right
    for channel in (0, 3):
        bad = tex.copy()
        bad[200, 77, channel] ^= 1
        t2 = tables_of(ctx, bad, tf)
        assert t2[FLAG_WORD] & 1 == 0
        assert np.array_equal(np.unpackbits(t2[:2048].view(np.uint8), bitorder="little").reshape(256, 256), (bad[..., 3] > 0).astype(np.uint8))
    # no uniform: no claim
    assert tables_of(ctx, tex, None)[FLAG_WORD] & 1 == 0


@pytest.mark.parametrize("kind", ["rgba_random", "one_texel_off", "uniform_of_another_tf"])
@pytest.mark.parametrize("skipping_type", [abi.SKIP_NONE, abi.SKIP_DISTANCE])
def test_render_with_textures_that_are_not_the_separable_product(ctx, kind, skipping_type):
    """Any RGBA8 texture must render like the oracle: the integrator may only take the two-table short cut when the device check passed."""
    vol = O.synth_volume((72, 64, 56), 1, 5)
    scene = T.OracleScene(vol, abi.VolumeOptions(**T.APP_TF), 4)
    rng = np.random.default_rng(3)
    tex = scene.tex.copy()
    if kind == "rgba_random":
        tex = rng.integers(0, 256, size=(256, 256, 4), dtype=np.uint8)
        tex[:, :30, 3] = 0  # keep low intensities empty so empty-space skipping has something to skip
    elif kind == "one_texel_off":
        g, i = np.argwhere(tex[..., 3] > 0)[100]
        tex[g, i, 1] ^= 0x40
    else:
        tex = O.transfer_function_texture(abi.VolumeOptions(intensity_min=0.3, intensity_max=0.9, gradient_min=0.0, gradient_max=0.5))
    scene.tex = tex
    scene._maps = {}
    v, tf = make_gpu_volume(ctx, scene)
    v.transfer_function.copy_(torch.from_numpy(tex))
    ctx.transfer_function_tables(v.transfer_function.data_ptr(), tf, v.transfer_function_bits.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(v.transfer_function_bits[FLAG_WORD].item()) & 1 == 0
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (128, 80)
    for az in (20.0, 200.0):
        view, proj = T.orbit(az, image_size=size)
        params = scene.params(view, proj, size, abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0))
        ref = scene.render(params)
        assert ref.counts[..., 0].sum() > 0
        compare_render(gpu_render(ctx, v, params), ref, "%s az %g" % (kind, az))
