"""CPU: the oracle against the committed self-generated golden vectors (tests/golden/make_golden.py).
GPU: the HIP path against the same committed vectors."""
import os

import numpy as np
import pytest

from oracle import vkv_oracle as O
from tests.golden import make_golden as G
from vkvolume_amd import abi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_v1.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_oracle_reproduces_golden_precompute(gold):
    assert np.array_equal(O.synth_volume(G.SHAPE, 1, G.SEED), gold["volume"])
    scene, _ = G.scene_and_params(0, True, gold["volume"])
    assert np.array_equal(scene.grad, gold["gradient"])
    assert np.array_equal(scene.tex[..., 3], gold["tf_alpha"])
    assert np.array_equal(scene.maps(abi.SKIP_BLOCK)[0], gold["occupancy"])
    assert np.array_equal(scene.maps(abi.SKIP_DISTANCE)[0], gold["distance"])
    assert np.array_equal(scene.maps(abi.SKIP_ANISOTROPIC_DISTANCE), gold["distance_aniso"])


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("ert", [True, False])
def test_oracle_reproduces_golden_frames(gold, mode, ert):
    scene, p = G.scene_and_params(mode, ert, gold["volume"])
    r = scene.render(p)
    assert np.array_equal(r.counts, gold["counts_m%d_e%d" % (mode, ert)])
    assert np.array_equal(r.color, gold["color_m%d_e%d" % (mode, ert)])
    assert np.array_equal(r.depth, gold["depth_m%d_e%d" % (mode, ert)])
    if mode == 2 and ert:
        # the committed block has the VkvRenderParams layout of rounds 1-5; round 6 put VkvTileSchedule.rect and .fill_outside (16 + 4 bytes, zero here: the whole
        # image) behind the schedule's six words.  Every other byte - uniforms, ray generator, options, schedule, extents - must be the old one.
        import ctypes as C
        new, was = bytes(p), gold["params_m2_e1"].tobytes()
        rect_at = abi.RenderParams.tiles.offset + abi.TileSchedule.rect.offset
        ext_at, ptr_at = abi.RenderParams.volume_extent.offset, abi.RenderParams.d_volume.offset
        assert new[rect_at:rect_at + C.sizeof(abi.TileRect) + 4] == bytes(20) and ext_at == rect_at + 16 + 4  # the rectangle, then fill_outside (all zero here)
        assert new[:rect_at] == was[:rect_at]
        assert new[ext_at:ext_at + 24] == was[rect_at:rect_at + 24]  # volume and map extent
        was_ptr_at = (rect_at + 24 + 7) // 8 * 8
        assert new[ptr_at:] == was[was_ptr_at:] and len(new) - ptr_at == len(was) - was_ptr_at


@pytest.mark.gpu
def test_hip_reproduces_golden(gold, ctx):
    import torch
    from tests import helpers as T
    from tests.test_gpu_parity import COLOR_TOL, DEPTH_TOL, gpu_render, make_gpu_volume
    from vkvolume_amd import volume as V
    scene = T.OracleScene(gold["volume"], abi.VolumeOptions(**T.APP_TF), G.BLOCK, voxel_size=(0.0003, 0.0003, 0.0007), axis_angle=(1, 0, 0, 90))
    v, tf = make_gpu_volume(ctx, scene)
    assert np.array_equal(v.gradient.cpu().numpy(), gold["gradient"])
    assert np.array_equal(v.transfer_function.cpu().numpy()[..., 3], gold["tf_alpha"])
    cdm = V.ComputeDistanceMap(ctx)
    for mode, key in ((abi.SKIP_ANISOTROPIC_DISTANCE, "distance_aniso"), (abi.SKIP_BLOCK, "occupancy"), (abi.SKIP_DISTANCE, "distance")):
        cdm.compute(v, tf, mode)
        torch.cuda.synchronize()
        if mode == abi.SKIP_ANISOTROPIC_DISTANCE:
            for k in range(8):
                assert np.array_equal(v.distance_maps[k].cpu().numpy(), gold[key][k])
        else:
            assert np.array_equal(v.distance_maps[0].cpu().numpy(), gold[key])
    for mode in (0, 1, 2, 3):
        cdm.compute(v, tf, mode)
        for ert in (True, False):
            _, p = G.scene_and_params(mode, ert, gold["volume"])
            color, counts, depth, _ = gpu_render(ctx, v, p)
            assert np.array_equal(counts, gold["counts_m%d_e%d" % (mode, ert)])
            assert np.abs(color - gold["color_m%d_e%d" % (mode, ert)]).max() <= COLOR_TOL
            assert np.abs(depth - gold["depth_m%d_e%d" % (mode, ert)]).max() <= DEPTH_TOL
