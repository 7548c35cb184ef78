"""GPU: bench.py's N > 1 orchestration on a one-GPU box (two real ranks over gloo, a one-rank RCCL group) and the keys of its line."""
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


def test_bench_two_ranks_share_one_device_over_gloo():
    """The N > 1 orchestration of bench.py with TWO REAL RANKS (the self-launching parent, torch.distributed.run, rank != owner paths, the
    rotating owner, buffer-set reuse behind the exchange's events, the max-over-ranks timing, --verify on the launch's owner) on a box with
    one GPU: both ranks use device 0 and the tile blocks travel over gloo through host memory (RCCL refuses two ranks on one device; what
    stays untested without a second GPU is RCCL moving the bytes).  The assembled frame of the last step must equal a direct render."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    for steps, fpl, more in ((12, 8, []), (7, 3, []), (11, 4, ["--frame-owner", "spread"]), (6, 6, ["--tile-rect", "off"])):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device", "--workload", "small", "--steps", str(steps),
               "--warmup", "3", "--frames-per-launch", str(fpl), "--min-seconds", "0.2", "--verify", "--c5-block", "off", "--no-cpu-baseline", "--launch-timeout", "600"] + more
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == steps and d["scaling"] == "weak"
        assert "gloo" in d["rccl_ranks_source"].lower() and "GLOO" in d["config"]["parallelism"]
        assert d["phases"]["launches_sampled"] > 0 and d["phases"]["render_ms"] > 0 and "gather_ms" in d["phases"] and "scatter_ms" in d["phases"]
        assert d["value"] > 0 and d["roofline"]["frac"] > 0
        assert "verify ok" in r.stderr, r.stderr[-2000:]
        assert "native_exchange" not in d  # --native-block auto: only with the nccl backend
        # round 6: only the tiles of each frame's screen rectangle travel (the small workload's box fills about half of the frame)
        x = d["exchange_bytes_per_frame"]
        if "--tile-rect" in more:
            assert x["ratio"] == 1.0 and x["tile_rect"] == "off"
        else:
            assert 0.2 < x["ratio"] < 0.9 and x["tile_rect"] == "on" and max(x["tiles_of_rect_per_view"]) < x["tiles_of_frame"], x
        assert ("spread" in x["frame_owner"]) == ("spread" in more)
    # the C ABI's exchange asked for where RCCL cannot provide it (two ranks on one device): the block degrades to an "error" string in its
    # sub-object, the headline line is still printed, once, and the job ends with status 0
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device", "--workload", "small", "--steps", "8",
           "--warmup", "2", "--min-seconds", "0.2", "--c5-block", "off", "--no-cpu-baseline", "--launch-timeout", "600", "--native-block", "on", "--native-timeout", "60"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["value"] > 0 and isinstance(d["native_exchange"].get("error"), str) and d["native_exchange"]["error"], d.get("native_exchange")


def test_bench_one_rank_group_carries_both_exchanges():
    """bench.py on the gather path with a ONE-rank RCCL group (--force-gather): the headline with the torch.distributed exchange and, in the same
    line, the short block of the same launches through the C ABI's exchange ("native_exchange": ncclGather on a communicator of our own +
    one de-interleave kernel per launch), for the c3-like headline and for the c5-like strong block."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-gather", "--workload", "small", "--steps", "12", "--warmup", "3",
           "--min-seconds", "0.2", "--verify", "--c5-block", "off", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    ne = d["native_exchange"]
    assert "error" not in ne, ne
    assert ne["ms_per_step"] > 0 and ne["rccl_ranks"] == 1 and "ncclCommCount" in ne["rccl_ranks_source"] and "exchange_ms" in ne["phases"] and ne["phases"]["render_ms"] > 0
    assert "gather_ms" in d["phases"] and d["rccl_ranks"] == 1
    # the same launches with the owners spread inside a launch (one RCCL group of per-frame gathers: with one rank every owner is rank 0)
    ns = d["native_exchange_spread"]
    assert "error" not in ns, ns
    assert ns["ms_per_step"] > 0 and "exchange_ms" in ns["phases"] and "spread" in ns["exchange_bytes_per_frame"]["frame_owner"]
    assert d["exchange_bytes_per_frame"]["ratio"] < 0.9 and ne["exchange_bytes_per_frame"]["per_rank"] == d["exchange_bytes_per_frame"]["per_rank"]


def test_bench_moving_camera_and_feedback_switch():
    """bench.py --camera moving (every frame a new view, one degree of orbit from the last, blocks sweeping forwards and backwards) and
    --no-feedback (VkvTuning.feedback = 0), the blocks the default line reports as camera_and_feedback: the frame left by the last step must
    equal a direct render of that step's view (--verify), and the default line's side blocks come without an error."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    for more in (["--camera", "moving"], ["--camera", "moving", "--camera-step", "3", "--no-feedback"], ["--no-feedback"]):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", "small", "--steps", "11", "--warmup", "3", "--min-seconds", "0.3", "--verify",
               "--extras", "off", "--no-cpu-baseline"] + more
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert d["value"] > 0 and d["repeats"] >= 2 and "verify ok" in r.stderr
        assert ("moves" in d["config"]["submission"]) == ("--camera" in more) and ("switched off" in d["config"]["submission"]) == ("--no-feedback" in more)
        assert d["roofline"]["traffic"] is None  # the PMC figures belong to the static-camera headline only
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", "small", "--steps", "8", "--warmup", "2", "--min-seconds", "0.2", "--extras", "on",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    for key in ("static", "static_feedback_off", "moving", "moving_feedback_off"):
        assert d["camera_and_feedback"][key]["ms_per_step"] > 0, d["camera_and_feedback"]
    for key in ("dense", "probe_only"):
        assert "error" not in d["asymptotes"][key], d["asymptotes"]
