"""CPU-side copy of the bench scene for the offline scheduling experiments (tools/lookahead_sim.py, tools/modesort_sim.py): the oracle's
synthetic volume, gradient map and Chebyshev distance map at `scale` x the C3 extent, saved to /tmp/sim/*.npy (full scale: ~3 minutes).
usage: build_scene_cpu.py [scale]"""
import sys, os, math, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.makedirs("/tmp/sim", exist_ok=True)
from oracle import vkv_oracle as O
from vkvolume_amd import abi, camera
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
W, H, D = int(1024 * scale), int(1024 * scale), int(795 * scale)
t=time.time()
vol = O.synth_volume((W, H, D), 1, 0xC0FFEE03); print("synth", time.time()-t); t=time.time()
opt = abi.VolumeOptions(intensity_min=0.1, intensity_max=1.0, gradient_min=0.0, gradient_max=0.2)
tf, tex = O.transfer_function_uniform(opt), O.transfer_function_texture(opt)
grad = O.gradient_map(vol, tf); print("grad", time.time()-t); t=time.time()
maps = O.compute_distance_map(vol, grad, tex, tf, 4, abi.SKIP_DISTANCE); print("maps", time.time()-t)
np.save("/tmp/sim/vol_%g.npy" % scale, vol); np.save("/tmp/sim/grad_%g.npy" % scale, grad); np.save("/tmp/sim/maps_%g.npy" % scale, maps)
