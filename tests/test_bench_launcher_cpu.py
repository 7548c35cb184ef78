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


def test_dry_launch_of_the_eight_gpu_c5_job():
    """The driver's scaling run is `python bench.py --gpus 8 ...`; BASELINE.json configs[4] (c5, strong) can also be asked for as the
    headline.  The ranks get the workload and the launch timeout unchanged."""
    r = _dry(["--gpus", "8", "--workload", "c5", "--steps", "16", "--warmup", "4", "--launch-timeout", "600", "--dry-launch"])
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout.strip())
    cmd = d["dry_launch"]
    assert d["n_ranks"] == 8 and d["launch_timeout_s"] == 600.0
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    child = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert child[child.index("--workload") + 1] == "c5"
    assert child[child.index("--launch-timeout") + 1] == "600"


def test_launch_timeout_kills_the_ranks_it_started(tmp_path):
    """A job whose ranks hang (RCCL bring-up on a broken node) must not hang the driver: after --launch-timeout seconds the parent kills
    the process group it started and exits with status 124.  The stand-in child spawns a grandchild; both must be gone."""
    import time
    sys.path.insert(0, ROOT)
    pidfile = tmp_path / "pids"
    child = [sys.executable, "-c",
             "import os, subprocess, sys, time\n"
             "g = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(120)'])\n"
             "open(%r, 'w').write('%%d %%d' %% (os.getpid(), g.pid))\n"
             "time.sleep(120)\n" % str(pidfile)]
    code = ("import sys; sys.path.insert(0, %r); import importlib.util as u\n"
            "spec = u.spec_from_file_location('bench_launcher', %r); src = open(%r).read().split('if __name__ == \"__main__\":')[0]\n"
            "ns = {}; exec(compile(src, 'bench_head', 'exec'), ns)\n"
            "ns['self_launch'](['--gpus', '2', '--launch-timeout', '2'], child_cmd=%r)\n" % (ROOT, os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "bench.py"), child))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 124, (r.returncode, r.stderr)
    assert "did not finish within --launch-timeout" in r.stderr
    assert time.time() - t0 < 30
    pids = [int(x) for x in pidfile.read_text().split()]
    time.sleep(0.5)
    for pid in pids:
        alive = True
        try:
            os.kill(pid, 0)
            # a killed child that nobody has reaped yet still answers signal 0: look at its state
            with open("/proc/%d/stat" % pid) as f:
                alive = f.read().split(")")[-1].split()[0] != "Z"
        except (OSError, IOError):
            alive = False
        assert not alive, "process %d of the timed-out job is still running" % pid
