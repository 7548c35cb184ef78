"""Offline experiments (CPU, oracle traces): collect the per-ray event sequences of sampled 32x32 pixel regions of the C3 bench views and
cache them in /tmp/sim/traces_<view>.npz for the scheduling simulators (tools/sched_policies_sim.py).
usage: sim_traces.py [azimuths...]      (scene from tools/build_scene_cpu.py)
Events per ray: 0 = probe that skipped ('P'), 1 = probe that found the cell occupied ('O'), 2 = empty sample ('S'), 3 = sample with alpha ('A')."""
import sys, os, math, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vkv_oracle as O
from vkvolume_amd import abi, camera

REGION = 32
STRIDE = 5        # every STRIDE-th 32x32 region in x and y (staggered)


def scene():
    W, H, D = 1024, 1024, 795
    vol, grad, maps = (np.ascontiguousarray(np.load("/tmp/sim/%s_1.npy" % n, mmap_mode="r")) for n in ("vol", "grad", "maps"))
    opt = abi.VolumeOptions(intensity_min=0.1, intensity_max=1.0, gradient_min=0.0, gradient_max=0.2)
    tf, tex = O.transfer_function_uniform(opt), O.transfer_function_texture(opt)
    ext = abi.Extent3D(W, H, D)
    me = O.map_extent(ext, 4)
    ixf = camera.image_transform((0.0003, 0.0003, 0.0007), (W, H, D), (1, 0, 0, 90))
    node = camera.benchmark_node_transform(ixf)
    m = (node.astype(np.float64).T @ ixf.astype(np.float64).T)[:3, :3]
    radius = 1.5 * 0.5 * math.sqrt(sum(float(np.linalg.norm(m[:, i])) ** 2 for i in range(3)))
    return vol, grad, maps, tf, tex, ext, me, ixf, node, radius


def params_for(az, sc, iw=1920, ih=1080):
    vol, grad, maps, tf, tex, ext, me, ixf, node, radius = sc
    view, proj = camera.orbit_camera(az, 20.0, radius), camera.perspective_vulkan(60.0, iw / ih, 0.1, 1000.0)
    cam, rc, rg = O.build_uniforms(view, proj, node, ixf, 1.0, (iw, ih), ext, me)
    p = abi.RenderParams()
    p.camera, p.ray_cast, p.ray_gen, p.transfer_function = cam, rc, rg, tf
    p.options = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0, early_ray_termination=1)
    p.use_precomputed_gradient = 1
    p.image_width, p.image_height = iw, ih
    p.tiles = abi.full_frame_tiles(iw, ih)
    p.volume_extent, p.map_extent = ext, me
    return p


def collect(az, sc):
    vol, grad, maps, tf, tex = sc[:5]
    p = params_for(az, sc)
    code = np.zeros(256, np.uint8)
    code[ord('P')], code[ord('O')], code[ord('S')], code[ord('A')] = 0, 1, 2, 3
    regions, offs, lens, data = [], [], [], []
    n = 0
    t = time.time()
    for ry in range(0, 1080 // REGION, 1):
        for rx in range((ry * 2) % STRIDE, 1920 // REGION, STRIDE):
            for ly in range(REGION):
                for lx in range(REGION):
                    ev, st = O.trace_ray_steps(p, vol, grad, tex, maps, rx * REGION + lx, ry * REGION + ly)
                    offs.append(n)
                    lens.append(len(ev))
                    if len(ev):
                        data.append(code[ev])
                        n += len(ev)
            regions.append((rx, ry))
    data = np.concatenate(data) if data else np.zeros(0, np.uint8)
    print("az %g: %d regions, %d rays, %d events in %.0f s" % (az, len(regions), len(offs), n, time.time() - t))
    np.savez_compressed("/tmp/sim/traces_%g.npz" % az, regions=np.array(regions), offs=np.array(offs), lens=np.array(lens), data=data)


if __name__ == "__main__":
    sc = scene()
    for az in [float(a) for a in sys.argv[1:]] or [0.0, 135.0]:
        collect(az, sc)
