"""Offline experiment (CPU, oracle traces): what would sorting the rays of a 16x16 workgroup by the kind of their next event (probe / sample)
into pure waves every iteration save?  Cost model: SIMD cycles per wave iteration of a pure-probe (235), pure-sample (370) and mixed (540) wave,
from tools/isa_loop_stats.py.  Result on C3: 1.46x with free exchanges, 1.05x at 120 cycles of LDS exchange + barrier per wave iteration: not built.
usage: modesort_sim.py   (scene from tools/build_scene_cpu.py)"""
import sys, os, math, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vkv_oracle as O
from vkvolume_amd import abi, camera
tstride=5
W, H, D = 1024,1024,795; iw, ih = 1920,1080
vol, grad, maps = (np.ascontiguousarray(np.load("/tmp/sim/%s_1.npy" % n, mmap_mode="r")) for n in ("vol", "grad", "maps"))
opt = abi.VolumeOptions(intensity_min=0.1, intensity_max=1.0, gradient_min=0.0, gradient_max=0.2)
tf, tex = O.transfer_function_uniform(opt), O.transfer_function_texture(opt)
ext = abi.Extent3D(W, H, D); me = O.map_extent(ext, 4)
ixf = camera.image_transform((0.0003, 0.0003, 0.0007), (W, H, D), (1, 0, 0, 90)); node = camera.benchmark_node_transform(ixf)
m = (node.astype(np.float64).T @ ixf.astype(np.float64).T)[:3, :3]
radius = 1.5 * 0.5 * math.sqrt(sum(float(np.linalg.norm(m[:, i])) ** 2 for i in range(3)))
view, proj = camera.orbit_camera(0.0, 20.0, radius), camera.perspective_vulkan(60.0, iw / ih, 0.1, 1000.0)
cam, rc, rg = O.build_uniforms(view, proj, node, ixf, 1.0, (iw, ih), ext, me)
p = abi.RenderParams(); p.camera, p.ray_cast, p.ray_gen, p.transfer_function = cam, rc, rg, tf
p.options = abi.RenderOptions(skipping_type=abi.SKIP_DISTANCE, clip_distance=1.0, early_ray_termination=1)
p.use_precomputed_gradient = 1; p.image_width, p.image_height = iw, ih
p.tiles = abi.full_frame_tiles(iw, ih); p.volume_extent, p.map_extent = ext, me
CP, CS, CM = 235, 370, 540
cur = srt = 0; cur_it = srt_it = 0
for ty in range(0, ih // 16, tstride):
    for tx in range((ty // tstride) % tstride, iw // 16, tstride):
        waves = [[] for _ in range(4)]
        for ly in range(16):
            for lx in range(16):
                ev, st = O.trace_ray_steps(p, vol, grad, tex, maps, tx * 16 + lx, ty * 16 + ly)
                if len(ev): waves[(ly // 8) * 2 + lx // 8].append(((ev == ord('P')) | (ev == ord('O'))).tolist())
        rays = [r for w in waves for r in w]
        if not rays: continue
        # current: static 8x8 waves
        for w in waves:
            if not w: continue
            n = max(len(r) for r in w)
            for t in range(n):
                pr = any(len(r) > t and r[t] for r in w); sm = any(len(r) > t and not r[t] for r in w)
                cur += CM if (pr and sm) else (CP if pr else CS); cur_it += 1
        # mode-sorted within the workgroup, one event per ray per iteration
        n = max(len(r) for r in rays)
        for t in range(n):
            npr = sum(1 for r in rays if len(r) > t and r[t]); nsm = sum(1 for r in rays if len(r) > t and not r[t])
            wp, ws = -(-npr // 64), -(-nsm // 64)
            srt += wp * CP + ws * CS; srt_it += wp + ws
print("static 8x8 waves : cost %d, wave-iterations %d" % (cur, cur_it))
print("mode-sorted (ideal, no exchange cost): cost %d (%.2fx), wave-iterations %d" % (srt, cur / srt, srt_it))
for co in (60, 120, 200):
    print("  with %d cycles of exchange overhead per wave-iteration: %.2fx" % (co, cur / (srt + co * srt_it)))
