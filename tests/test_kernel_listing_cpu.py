"""Properties of the compiled ray-march kernels that no run can show but the measurements depend on, read from the compiler's listing of
the bench's translation unit (hipcc cross-compiles gfx950 without a GPU; ~30 s):
  * the march loops with hand-set load waits start with the PROBE side (its byte is the oldest load in flight and its outcome runs while the
    footprint gathers are under way: lean_march; the result does not depend on that order, the frame time does), and the loads in front
    of the hand-set wait are the probe byte FOLLOWED by the four footprint dwords (the result does depend on this one: vmcnt(4) releases the oldest load);
  * the batch kernels of the packed-image path fit their 64-register budget without scratch (a kernel with scratch makes its first launch
    allocate device memory, which vkv_render / vkv_render_batch promise not to do)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vkvolume_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _makefile_flags():
    """the FLAGS line of the product Makefile (the listing must be built like the library)"""
    text = open(os.path.join(CSRC, "Makefile")).read().replace("\\\n", " ")
    m = re.search(r"^FLAGS\s*:=\s*(.*)$", text, flags=re.M)
    return [f for f in m.group(1).split() if not f.startswith("$(")]


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if not os.path.exists(HIPCC) and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("listing") / "raymarch_s2e1.s")
    flags = [f.replace("$(ARCH)", "gfx950") for f in _makefile_flags()]
    cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc"] + flags + ["--offload-arch=gfx950", "-S", "--cuda-device-only", os.path.join(CSRC, "raymarch_s2e1.hip"), "-o", out]
    subprocess.run(cmd, check=True, cwd=CSRC, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    return out


def test_march_loops_with_hand_set_waits_start_with_the_probe_side(listing):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_march_loop.py"), listing, "lean", "--order"], stdout=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert r.stdout.count("probe side first") >= 8, r.stdout[-2000:]        # the clamp-free loop and the loop with the clamps of every such kernel
    assert "SAMPLE side first" not in r.stdout
    # the hand-set vmcnt(4) releases the probe byte only when it was issued BEFORE the four footprint dwords
    assert "LOAD ORDER" not in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("loads in front of the wait: probe byte, then four footprint dwords") == r.stdout.count("probe side first")


def test_batch_kernels_have_no_scratch_and_the_bench_kernel_keeps_eight_waves(listing):
    kernels = {}
    name = None
    for line in open(listing):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
        m = re.match(r";\s*(NumVgprs|ScratchSize):\s*(\d+)", line)
        if m and name:
            kernels.setdefault(name, {})[m.group(1)] = int(m.group(2))
    batch = {k: v for k, v in kernels.items() if "k_raymarch_lean_batch" in k}
    assert len(batch) >= 8
    for k, v in batch.items():
        assert v["ScratchSize"] == 0, (k, v)
    bench_kernel = [v for k, v in batch.items() if "ILi2ELb1ELi1ELj55E" in k]        # <VKV_SKIP_DISTANCE, ERT, gradient map, kLfFullNc>
    assert len(bench_kernel) == 1 and bench_kernel[0]["NumVgprs"] <= 64, bench_kernel
    single = {k: v for k, v in kernels.items() if re.match(r"_Z15k_raymarch_leanI", k)}
    for k, v in single.items():
        assert v["ScratchSize"] == 0, (k, v)
