"""Round 4 GPU tests (through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import vkv_oracle as O
from tests import helpers as T
from tests.test_gpu_parity import make_gpu_volume
from vkvolume_amd import abi, volume as V

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("skipping_type", [abi.SKIP_BLOCK, abi.SKIP_DISTANCE, abi.SKIP_ANISOTROPIC_DISTANCE])
def test_sample_count_test_mode_without_a_counter_buffer(ctx, skipping_type):
    """Test::NumTextureSamples (frag:324-334: the pixel's colour is (volume samples + map probes) / n_steps_max) with early ray
    termination ON and NO d_out_counts - the reference's GUI switches the test mode independently of ERT and of the skipping type
    (src/volume_render.cpp:539).  The launchers run the loop without the per-pixel counters for launches that have no counter buffer;
    this mode needs them whatever the outputs are (every marched pixel came out (0, 0, 0, 1)).  Single launch and batch
    launch, both address-table kinds, float colour and RGBA8 against the oracle."""
    opt = abi.VolumeOptions(**T.APP_TF)
    scene = T.OracleScene(O.synth_volume((88, 72, 64), 1, 0x5EED0004), opt, 4)
    v, tf = make_gpu_volume(ctx, scene)
    V.ComputeDistanceMap(ctx).compute(v, tf, skipping_type)
    size = (160, 96)
    ro = abi.RenderOptions(skipping_type=skipping_type, clip_distance=1.0, early_ray_termination=True, test=abi.TEST_NUM_TEXTURE_SAMPLES)
    st = torch.cuda.current_stream().cuda_stream
    plist, refs = [], []
    for az in (10.0, 200.0):
        view, proj = T.orbit(az, image_size=size)
        params = scene.params(view, proj, size, ro)
        plist.append(V.VolumeRenderSubpass(ctx, v, ro, size).bind(params))
        refs.append(scene.render(params, want_rgba8=True))
    assert refs[0].counts[..., 0].sum() > 0 and refs[0].counts[..., 1].sum() > 0
    assert float(refs[0].color[..., 0].max()) > 0.0  # the grey level of the count output

    def outputs():
        return [dict(color=torch.full((size[1], size[0], 4), -1.0, dtype=torch.float32, device="cuda"),
                     rgba8=torch.full((size[1], size[0], 4), 7, dtype=torch.uint8, device="cuda")) for _ in plist]

    def point(p, o):
        q = abi.RenderParams.from_buffer_copy(p)
        q.d_out_color, q.d_out_rgba8, q.d_out_counts, q.d_out_depth = o["color"].data_ptr(), o["rgba8"].data_ptr(), None, None
        return q

    try:
        for tables in (2, 1, 0):
            ctx.set_tuning(address_tables=tables)
            single, batch = outputs(), outputs()
            for p, o in zip(plist, single):
                ctx.render(point(p, o), st)
            ctx.render_batch([point(p, o) for p, o in zip(plist, batch)], st)
            torch.cuda.synchronize()
            for i, ref in enumerate(refs):
                for name, got in (("vkv_render", single[i]), ("vkv_render_batch", batch[i])):
                    assert np.array_equal(got["color"].cpu().numpy(), ref.color), "%s, tables %d, view %d: count colour differs from the oracle" % (name, tables, i)
                    assert np.array_equal(got["rgba8"].cpu().numpy(), ref.rgba8), "%s, tables %d, view %d: RGBA8 differs from the oracle" % (name, tables, i)
    finally:
        ctx.set_tuning(address_tables=2)
