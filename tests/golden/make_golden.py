"""Generates tests/golden/hotpath_v1.npz with the CPU oracle (oracle/vkv_oracle.c).

The reference (LDeakin/VkVolume) cannot run here and ships no golden vectors, so these are
SELF-GENERATED regression vectors: inputs + the oracle's outputs for every stage of the hot path.  They pin the oracle
against accidental change and give the GPU tests committed expected values; they do not pin parity with the reference.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import vkv_oracle as O  # noqa: E402
from tests import helpers as T  # noqa: E402
from vkvolume_amd import abi  # noqa: E402

SHAPE = (40, 36, 28)  # W, H, D — not multiples of the block size on purpose
SEED = 0xC0FFEE02
BLOCK = 4
IMAGE = (48, 32)
AZIMUTH = 33.0


def scene_and_params(mode, ert, vol=None):
    vol = O.synth_volume(SHAPE, 1, SEED) if vol is None else vol
    scene = T.OracleScene(vol, abi.VolumeOptions(**T.APP_TF), BLOCK, voxel_size=(0.0003, 0.0003, 0.0007), axis_angle=(1, 0, 0, 90))
    view, proj = T.orbit(AZIMUTH, image_size=IMAGE)
    uniforms = O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, IMAGE, scene.extent, scene.map_extent)
    p = scene.params(view, proj, IMAGE, abi.RenderOptions(skipping_type=mode, clip_distance=1.0, early_ray_termination=ert), uniforms=uniforms)
    return scene, p


def main():
    out = {}
    scene, _ = scene_and_params(0, True)
    out["volume"] = scene.vol
    out["gradient"] = scene.grad
    out["tf_alpha"] = scene.tex[..., 3]
    out["occupancy"] = scene.maps(abi.SKIP_BLOCK)[0]
    out["distance"] = scene.maps(abi.SKIP_DISTANCE)[0]
    out["distance_aniso"] = scene.maps(abi.SKIP_ANISOTROPIC_DISTANCE)
    for mode in (0, 1, 2, 3):
        for ert in (True, False):
            scene, p = scene_and_params(mode, ert, scene.vol)
            r = scene.render(p)
            out["counts_m%d_e%d" % (mode, ert)] = r.counts
            out["color_m%d_e%d" % (mode, ert)] = r.color
            out["depth_m%d_e%d" % (mode, ert)] = r.depth
            if mode == 2 and ert:
                out["params_m2_e1"] = np.frombuffer(bytes(p), np.uint8)  # the exact parameter block (pointers are null)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
