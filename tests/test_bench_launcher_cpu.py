"""bench.py --gpus N without WORLD_SIZE must start its own ranks (VERDICT r2 next #2a): the parent builds the torch.distributed.run command
line before anything touches the GPU.  --dry-launch prints that command line instead of running it."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dry(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=120)
    return r


def test_dry_launch_builds_the_torchrun_command_line():
    r = _dry(["--gpus", "2", "--steps", "5", "--warmup", "1", "--dry-launch"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    cmd = d["dry_launch"]
    assert d["n_ranks"] == 2
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    child = cmd[script + 1:]
    assert child == ["--gpus", "2", "--steps", "5", "--warmup", "1"]  # the ranks get the same arguments, minus --dry-launch


def test_dry_launch_honours_master_port_and_equals_form():
    r = _dry(["--gpus=8", "--dry-launch"], {"MASTER_PORT": "29777"})
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip())["dry_launch"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-port") + 1] == "29777"


def test_a_rank_of_an_existing_job_does_not_relaunch():
    # WORLD_SIZE set: the process is a rank; with a WORLD_SIZE that does not match --gpus it must stop with the mismatch message,
    # not start children (no GPU is needed to get that far)
    r = _dry(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=4 does not match --gpus 2" in r.stderr
    assert "dry_launch" not in r.stdout
