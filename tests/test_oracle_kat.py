"""CPU: pin the oracle.  The reference has no tests or golden vectors for this path (SURVEY.md §4, §8c — parity
unpinned), so the oracle is held to (1) brute-force mathematical known answers for the distance transforms,
(2) direct-formula checks of gradient / occupancy / TF texture written independently in numpy, (3) closed-form and
invariance properties of the integrator."""
import numpy as np
import pytest

from oracle import vkv_oracle as O
from tests import helpers as T
from vkvolume_amd import abi


def sparse(shape, seed, p):
    rng = np.random.default_rng(seed)
    return np.where(rng.random(shape) < p, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape,p,seed", [((9, 10, 13), 0.01, 1), ((9, 10, 13), 0.05, 2), ((16, 16, 16), 0.002, 3), ((4, 30, 5), 0.02, 4),
                                          ((1, 1, 40), 0.05, 5), ((6, 6, 6), 0.0, 6), ((5, 5, 5), 1.0, 7)])
def test_isotropic_distance_map_is_capped_chebyshev_distance(shape, p, seed):
    occ = sparse(shape, seed, p)
    assert np.array_equal(O.distance_map(occ), T.brute_force_chebyshev(occ))


@pytest.mark.parametrize("shape,p,seed", [((9, 10, 13), 0.01, 1), ((9, 10, 13), 0.05, 2), ((12, 12, 12), 0.004, 3), ((3, 20, 4), 0.03, 4)])
def test_anisotropic_maps_are_octant_chebyshev_distances(shape, p, seed):
    occ = sparse(shape, seed, p)
    maps = O.distance_map_anisotropic(occ)
    iso = O.distance_map(occ)
    for k in range(8):
        assert np.array_equal(maps[k], T.brute_force_chebyshev_octant(occ, k)), "octant %d" % k
        assert (maps[k] >= iso).all()  # SURVEY.md §4 KAT 3: one-sided distance >= two-sided
    assert np.array_equal(maps.min(axis=0), iso)  # the nearest occupied cell lies in one of the eight closed octants


def test_distance_saturates_at_255():
    occ = np.full((1, 1, 600), 255, np.uint8)
    occ[0, 0, 0] = 0
    d = O.distance_map(occ)[0, 0]
    assert np.array_equal(d, np.minimum(np.arange(600), 255).astype(np.uint8))
    m = O.distance_map_anisotropic(occ)
    assert np.array_equal(m[4][0, 0], np.minimum(np.arange(600), 255).astype(np.uint8))  # rays with dx < 0 look towards x = 0
    assert (m[0][0, 0, 1:] == 255).all() and m[0][0, 0, 0] == 0


def numpy_gradient(vol, modifier=1.0):
    """Independent numpy statement of shaders/get_gradient_compute.glsl:12-20 + RNE UNORM store."""
    v = vol.astype(np.float32) / np.float32(255.0)
    D, H, W = vol.shape
    z, y, x = np.indices(vol.shape)
    c = lambda a, n: np.clip(a, 0, n - 1)  # noqa: E731
    v1 = v[c(z - 1, D), c(y - 1, H), c(x + 1, W)]
    v2 = v[c(z + 1, D), c(y - 1, H), c(x - 1, W)]
    v3 = v[c(z - 1, D), c(y + 1, H), c(x - 1, W)]
    v4 = v[c(z + 1, D), c(y + 1, H), c(x + 1, W)]
    q = np.float32(0.25)
    gx, gy, gz = q * (((v1 - v2) - v3) + v4), q * (((-v1 - v2) + v3) + v4), q * (((-v1 + v2) - v3) + v4)
    g = np.clip(np.sqrt((gx * gx + gy * gy) + gz * gz) * np.float32(modifier), 0, 1).astype(np.float32)
    return np.rint(g * np.float32(255.0)).astype(np.uint8)


@pytest.mark.parametrize("shape", [(20, 17, 9), (5, 1, 3), (1, 1, 1)])
def test_gradient_map_formula(shape):
    vol = T.random_volume(shape, 2)
    tf = O.transfer_function_uniform(abi.VolumeOptions(**T.APP_TF))
    assert np.array_equal(O.gradient_map(vol, tf), numpy_gradient(vol))
    tf0 = O.transfer_function_uniform(abi.VolumeOptions(gradient_min=0.3, gradient_max=0.3))
    assert not tf0.use_gradient and (O.gradient_map(vol, tf0) == 255).all()  # get_gradient_compute.glsl:6-7


def test_transfer_function_texture_formula():
    opt = abi.VolumeOptions(**T.APP_TF)
    tex = O.transfer_function_texture(opt)
    i = np.arange(256, dtype=np.float32)
    f = np.float32
    ai = np.clip((i / f(255) - f(0.1)) * (f(1) / (f(1.0) - f(0.1))), 0, 1).astype(np.float32)
    ag = np.clip((i / f(255) - f(0.0)) * (f(1) / (f(0.2) - f(0.0))), 0, 1).astype(np.float32)
    expect = np.clip(ag[:, None] * ai[None, :] * f(255), 0, 255).astype(np.uint8)  # truncation (volume_component.cpp:259)
    assert np.array_equal(tex[..., 3], expect)
    assert all(np.array_equal(tex[..., c], expect) for c in range(3))  # greyscale: all channels = alpha
    # NEAREST lookup of byte/255 lands on texel `byte`
    assert np.array_equal(np.minimum(np.floor(i / f(255) * f(256)), 255).astype(int), np.arange(256))


@pytest.mark.parametrize("shape,block", [((20, 17, 9), 4), ((13, 13, 13), 3), ((8, 8, 8), 2), ((10, 7, 5), 5)])
def test_occupancy_map_formula(shape, block):
    vol = T.random_volume(shape, 8, sparsity=0.95)
    opt = abi.VolumeOptions(**T.APP_TF)
    tf, tex = O.transfer_function_uniform(opt), O.transfer_function_texture(opt)
    grad = O.gradient_map(vol, tf)
    occ = O.occupancy_map(vol, grad, tex, tf, block)
    alpha = tex[grad.astype(int), vol.astype(int), 3] > 0
    D, H, W = vol.shape
    md, mh, mw = occ.shape
    # block size is RE-DERIVED as ceil(volume / map) per axis (src/compute_distance_map.cpp:110-113): for 9 voxels and a
    # requested block of 4 the map has 3 cells and the effective block is 3, not 4
    bz, by, bx = -(-D // md), -(-H // mh), -(-W // mw)
    pad = np.zeros((md * bz, mh * by, mw * bx), bool)
    pad[:D, :H, :W] = alpha
    any_occ = pad.reshape(md, bz, mh, by, mw, bx).any(axis=(1, 3, 5))
    assert np.array_equal(occ, np.where(any_occ, 0, 255).astype(np.uint8))
    assert any_occ.any()


def sphere_scene(mode=abi.SKIP_NONE, size=(96, 96), ert=True, options=None, test=abi.TEST_NONE):
    vol = O.synth_volume((64, 64, 64), 0, 1)
    scene = T.OracleScene(vol, options or abi.VolumeOptions(**T.APP_TF), 4)
    view, proj = T.orbit(30.0, image_size=size)
    cam = O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, size, scene.extent, scene.map_extent)
    p = scene.params(view, proj, size, abi.RenderOptions(skipping_type=mode, clip_distance=1.0, early_ray_termination=ert, test=test),
                     uniforms=cam)
    return scene, p


def test_ray_entry_exit_lie_on_the_unit_cube():
    """Test::RayEntry / RayExit (frag:168-173): closed form — both points are on the cube surface and the camera, entry and exit
    are collinear."""
    scene, p = sphere_scene(test=abi.TEST_RAY_ENTRY)
    entry = scene.render(p).color
    scene, q = sphere_scene(test=abi.TEST_RAY_EXIT)
    exit_ = scene.render(q).color
    hit = entry[..., 3] == 1.0
    assert 0.3 < hit.mean() < 1.0 and np.array_equal(hit, exit_[..., 3] == 1.0)
    for pts in (entry[hit][:, :3], exit_[hit][:, :3]):
        on_face = np.minimum(np.abs(pts), np.abs(pts - 1.0)).min(axis=1)
        assert on_face.max() < 2e-6 and pts.min() > -2e-6 and pts.max() < 1 + 2e-6
    cam = np.array(list(p.ray_cast.camera_pos_tex)[:3], np.float64)
    a, b = entry[hit][:, :3] - cam, exit_[hit][:, :3] - cam
    cosang = (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
    assert cosang.min() > 1 - 1e-9 and (np.linalg.norm(b, axis=1) > np.linalg.norm(a, axis=1)).all()


def test_sample_count_matches_chord_length_without_ess():
    """No ESS, no ERT: every covered ray takes n_steps = ceil(dim_max * |exit - entry| * sf) volume samples (frag:176-178, 215)."""
    scene, p = sphere_scene(ert=False)
    r = scene.render(p)
    scene, pe = sphere_scene(test=abi.TEST_RAY_ENTRY)
    scene, px = sphere_scene(test=abi.TEST_RAY_EXIT)
    e, x = scene.render(pe).color, scene.render(px).color
    chord = np.linalg.norm((x[..., :3] - e[..., :3]).astype(np.float64), axis=-1)
    n = np.ceil(64 * chord)
    marched = r.counts[..., 0] > 0
    assert marched.mean() > 0.3
    assert np.abs(r.counts[..., 0][marched] - n[marched]).max() <= 1  # fp32 vs fp64 ceil at integer boundaries
    assert (r.counts[..., 1] == 0).all()


def test_ess_monotonic_and_nearly_invariant():
    """SURVEY.md §4 KAT 2/3: skipping changes sample counts monotonically, the image only by trilinear bleed."""
    imgs, totals = {}, {}
    for mode in (abi.SKIP_NONE, abi.SKIP_BLOCK, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE):
        scene, p = sphere_scene(mode)
        r = scene.render(p)
        imgs[mode], totals[mode] = r.color, int(r.counts[..., 0].sum() + r.counts[..., 1].sum())
    assert totals[3] <= totals[2] <= totals[1] <= totals[0]
    for mode in (1, 2, 3):
        assert np.abs(imgs[mode] - imgs[0]).max() < 5e-3
    assert imgs[0][..., 3].max() == 1.0  # ERT clamps alpha to exactly 1 (frag:296)


def test_intensity_only_tf_skipping_is_exact_on_binary_blocks():
    """Intensity-only TF, non-zero voxels kept one voxel inside their 4^3 map cells: the trilinear bleed of every
    non-zero voxel then stays inside an occupied cell, no skipped sample has alpha > 0, and every skipping mode must give
    the identical image bit for bit (SURVEY.md §4 KAT 2).  (Without the margin the reference algorithm itself is only
    approximately invariant: its one-sample step-back misses bleed on oblique rays, frag:254-256.)"""
    vol = np.zeros((32, 32, 32), np.uint8)
    vol[9:15, 13:19, 5:23] = 200
    vol[21:27, 5:7, 9:11] = 120
    opt = abi.VolumeOptions(intensity_min=0.1, gradient_min=0.0, gradient_max=0.0)
    scene = T.OracleScene(vol, opt, 4)
    size = (80, 64)
    view, proj = T.orbit(70.0, image_size=size)
    cam = O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, size, scene.extent, scene.map_extent)
    base = None
    for mode in (0, 1, 2, 3):
        for ert in (True, False):
            p = scene.params(view, proj, size, abi.RenderOptions(skipping_type=mode, clip_distance=1.0, early_ray_termination=ert), uniforms=cam)
            r = scene.render(p)
            key = r.color.copy()
            if ert:
                if base is None:
                    base = key
                assert np.array_equal(key, base), "mode %d" % mode
    assert base[..., 3].max() > 0.5


def test_loader_roundtrip(tmp_path):
    """LoadVolume (src/load_volume.cpp): header parsing, size check, endianness, normalisation with truncation."""
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 65536, size=(5, 4, 6), dtype=np.uint16)
    for endian in ("little", "big"):
        f = tmp_path / ("v_%s.raw" % endian)
        raw.astype("<u2" if endian == "little" else ">u2").tofile(f)
        (tmp_path / (f.name + ".header")).write_text("6 4 5 # extents\n0.001 0.002 0.003 # voxel size\n400.0 2538.0 # range\n"
                                                     "uint16_t %s # type\n1 0 0 90 # rotation\n" % endian)
        h = O.load_header(str(f) + ".header")
        assert h.extent.as_tuple() == (6, 4, 5) and h.type == b"uint16_t" and h.endianness == endian.encode()
        got = O.load_data(str(f), h)
        t = np.clip((raw.astype(np.float32) - np.float32(400)) / (np.float32(2538) - np.float32(400)), 0, 1)
        assert np.array_equal(got, (np.float32(255) * t).astype(np.uint8))
        m = np.array(list(h.image_transform), np.float64).reshape(4, 4).T  # rotate(90deg, x) * scale(physical size)
        assert np.allclose(m[:3, :3], [[0.006, 0, 0], [0, 0, -0.015], [0, 0.008, 0]], atol=1e-7)
    with pytest.raises(RuntimeError):
        O.load_header(str(tmp_path / "missing.header"))
    (tmp_path / "short.raw").write_bytes(b"\0" * 10)
    (tmp_path / "short.raw.header").write_text("6 4 5\n1 1 1\n0 255\nuint8_t little\n1 0 0 0\n")
    with pytest.raises(RuntimeError):
        O.load_data(str(tmp_path / "short.raw"), O.load_header(str(tmp_path / "short.raw.header")))


@pytest.mark.parametrize("opts,use_map", [(T.APP_TF, True), (T.APP_TF, False), (dict(intensity_min=0.4, intensity_max=0.8, gradient_min=0.0, gradient_max=0.0), True),
                                          (dict(intensity_min=0.2, intensity_max=0.8, gradient_min=0.06, gradient_max=0.12), True)])
def test_occupied_voxel_count_formula(opts, use_map):
    """shaders/occupied_voxel_count.comp uses the ANALYTIC transfer function (transfer_function.glsl:41-43)."""
    vol = O.synth_volume((30, 26, 22), 1, 4)
    opt = abi.VolumeOptions(use_precomputed_gradient=use_map, **opts)
    tf = O.transfer_function_uniform(opt)
    grad = O.gradient_map(vol, tf) if use_map else None
    f = np.float32
    i = vol.astype(np.float32) / f(255)
    if not tf.use_gradient:
        g = np.ones_like(i)
    elif use_map:
        g = grad.astype(np.float32) / f(255)
    else:
        v = i
        D, H, W = vol.shape
        z, y, x = np.indices(vol.shape)
        c = lambda a, n: np.clip(a, 0, n - 1)  # noqa: E731
        v1, v2 = v[c(z - 1, D), c(y - 1, H), c(x + 1, W)], v[c(z + 1, D), c(y - 1, H), c(x - 1, W)]
        v3, v4 = v[c(z - 1, D), c(y + 1, H), c(x - 1, W)], v[c(z + 1, D), c(y + 1, H), c(x + 1, W)]
        q = f(0.25)
        gx, gy, gz = q * (((v1 - v2) - v3) + v4), q * (((-v1 - v2) + v3) + v4), q * (((-v1 + v2) - v3) + v4)
        g = np.clip(np.sqrt((gx * gx + gy * gy) + gz * gz), 0, 1).astype(np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        ai = np.clip((i - f(tf.intensity_min)) * f(tf.intensity_range_inv), 0, 1)
        ag = np.clip((g - f(tf.gradient_min)) * f(tf.gradient_range_inv), 0, 1)
        expect = int(((ai * ag) > 0).sum())
    assert O.occupied_voxel_count(vol, grad, tf) == expect
    assert 0 < expect < vol.size


def test_c1_sphere_256_plumbing():
    """BASELINE.json configs[0]: 64^3 synthetic sphere, 256x256 offscreen, no ESS — the CPU scalar ray-marcher end to end
    (load -> gradient -> TF -> render), with closed-form checks: the image is symmetric under the sphere's symmetry about
    the view axis (left-right mirror for a camera in the x = 0 plane) and opaque in the middle, empty in the corners."""
    vol = O.synth_volume((64, 64, 64), 0, 1)
    assert vol[32, 32, 32] == 255 and vol[0, 0, 0] == 0 and np.array_equal(vol, vol[::-1, ::-1, ::-1])
    scene = T.OracleScene(vol, abi.VolumeOptions(**T.APP_TF), 4)
    size = (256, 256)
    view, proj = T.orbit(0.0, elevation=0.0, radius=150.0, image_size=size)
    cam = O.build_uniforms(view, proj, scene.node_transform, scene.image_transform, 1.0, size, scene.extent, scene.map_extent)
    p = scene.params(view, proj, size, abi.RenderOptions(skipping_type=abi.SKIP_NONE, clip_distance=1.0), uniforms=cam)
    r = scene.render(p)
    assert r.rays == 256 * 256
    a = r.color[..., 3]
    assert a[128, 128] == 1.0 and a[0, 0] == 0.0 and a[255, 255] == 0.0  # ERT clamps alpha to exactly 1 in the centre
    assert 0.1 < (a > 0).mean() < 0.6
    # mirror symmetry of the silhouette; the alpha values themselves differ a little because the 4-tap tetrahedron gradient
    # (get_gradient_compute.glsl:12-18) is not mirror-symmetric
    assert np.abs(a - a[:, ::-1]).max() < 0.1 and np.abs(a - a[::-1, :]).max() < 0.1
    assert ((a > 0) != (a[:, ::-1] > 0)).mean() < 2e-3
    assert (r.counts[..., 1] == 0).all() and r.counts[..., 0].max() <= 111  # n_steps_max = ceil(64 * sqrt(3))


def test_x_stage_recurrence_has_the_min_plus_closed_form():
    """Stage 0 of the distance transforms, g = min(g_prev + 1, input) forward and backward (distance_map.comp:57-71), equals
    out(x) = min over q of (input(q) + |x - q|) for ANY byte input - the closed form the wave-per-row x pass of the HIP build evaluates.
    Checked on a map of depth and height 1, where stages 1 and 2 are the identity."""
    rng = np.random.default_rng(17)
    for w in (1, 2, 7, 64, 300):
        raw = rng.integers(0, 256, size=(1, 1, w), dtype=np.uint8)
        raw[rng.random(raw.shape) < 0.5] = 255
        got = O.distance_map(raw)[0, 0].astype(np.int64)
        x = np.arange(w)
        want = (raw[0, 0].astype(np.int64)[None, :] + np.abs(x[:, None] - x[None, :])).min(axis=1)
        assert np.array_equal(got, want)
        aniso = O.distance_map_anisotropic(raw)
        up = np.where(x[None, :] >= x[:, None], raw[0, 0].astype(np.int64)[None, :] + (x[None, :] - x[:, None]), 1 << 30).min(axis=1)
        dn = np.where(x[None, :] <= x[:, None], raw[0, 0].astype(np.int64)[None, :] + (x[:, None] - x[None, :]), 1 << 30).min(axis=1)
        assert np.array_equal(aniso[0][0, 0], up) and np.array_equal(aniso[4][0, 0], dn)  # octant bit 4 = dx < 0
