"""Diagnostic (GPU): the fallback kernels odd shapes get - a volume with no extent a multiple of 4 (bench workload "odd", 493x493x443: k_gradient_map,
k_occupancy_map, k_pack_volume instead of the tiled / dword kernels) is timed by tools/time_precompute.py odd; this script adds the x pass of the
distance transform for map rows longer than 1024 cells (until round 5 the serial k_dm_x; round 6: k_dm_x_wave with 32 cells per lane up to 2048 cells)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vkvolume_amd import abi, lib  # noqa: E402

torch.cuda.set_device(0)
ctx = lib.Context(0)
st = torch.cuda.current_stream().cuda_stream
for mw, mh, md in ((1536, 128, 64), (1024, 128, 96)):
    g = torch.Generator(device="cuda").manual_seed(1)
    occ = torch.where(torch.rand((md, mh, mw), device="cuda", generator=g) < 0.004, 0, 255).to(torch.uint8)
    m, sw = occ.clone(), torch.empty_like(occ)
    ext = abi.Extent3D(mw, mh, md)

    def run():
        m.copy_(occ)
        ctx.distance_map(m.data_ptr(), sw.data_ptr(), ext, st)

    run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e.record()
    torch.cuda.synchronize()
    n = mw * mh * md
    print("isotropic transform of a %dx%dx%d map (%s x pass): %.3f ms per update incl. the copy of the occupancy map, %.1f Mcell" % (
        mw, mh, md, "k_dm_x_wave, 32 cells per lane (round 6; k_dm_x until round 5)" if mw > 1024 else "k_dm_x_wave", s.elapsed_time(e) / 10, n / 1e6))
