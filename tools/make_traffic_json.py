"""Build profiles/<round>_traffic.json (what bench.py's roofline.traffic / valu_busy replay) from the PMC passes of tools/collect_profiles.sh.

    python tools/make_traffic_json.py gpurun_out/<tag> r3 > profiles/r3_traffic.json

HBM bytes per launch of the ray-march kernel = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024: FETCH_SIZE tallies every 128-byte request at 64 bytes on
gfx950 (MI355X_MICROARCH.md, HBM) - calibrated on a coalesced stream in round 1 AND on the integrator's own 2-byte-aligned dword gathers in
round 3 (tools/micro/gather_fetch.hip, profiles/r3_micro_gather_fetch.txt: the same factor, traffic counted in whole 128-byte lines).
VALU busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8) of the same kernel: the gfx9 VALUBusy formula (rocprofv3 sums
GRBM_GUI_ACTIVE over the 8 XCDs).  It prices every wave instruction at 4 cycles; gfx950 issues the plain fp32 / integer-add / logic opcodes
in 2.4 - 2.9 (tools/micro/valu_mix.hip), so the figure is an upper bound of the SIMDs' busy share - read it as "the VALU is the pipe that
is full", next to SQ_INSTS_VALU per launch.
The file names the integrator sources it was measured on (sha256 over raymarch_core.hpp, raymarch.hip, vkv_device.hpp, Makefile: bench.py
recomputes it and withholds the figures when the tree differs; comments and whitespace do not count)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def averages(pattern, kernel):
    acc = collections.defaultdict(list)
    for d in glob.glob(pattern):
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    out_dir, rnd = sys.argv[1], sys.argv[2]
    kernel = "k_raymarch_lean_batch"
    avg, n = averages(os.path.join(out_dir, "pmc_batch8_*"), kernel)
    import bench
    digest = bench.kernel_source_digest()
    commit = open(os.path.join(ROOT, ".commit_id")).read().strip() if os.path.exists(os.path.join(ROOT, ".commit_id")) else None
    fetch, write = avg.get("FETCH_SIZE"), avg.get("WRITE_SIZE")
    valu = None
    if avg.get("SQ_ACTIVE_INST_VALU") and avg.get("GRBM_GUI_ACTIVE"):
        valu = avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * avg["GRBM_GUI_ACTIVE"] / 8.0)
    print(json.dumps({
        "_comment": __doc__.split("\n\n")[2].replace("\n", " "),
        "workload": "c3", "kernel": kernel, "frames_per_launch": 8,
        "fetch_size_kib_avg": fetch, "write_size_kib_avg": write,
        "traffic_bytes_per_launch": int((fetch * 2 + write) * 1024) if fetch and write else None,
        "valu_busy": round(valu, 4) if valu else None,
        "counters_avg": {k: round(v, 1) for k, v in sorted(avg.items())}, "launches_averaged": n,
        "kernel_source_sha256": digest, "commit": commit,
        "source": "profiles/%s_rocprof.txt" % os.path.basename(out_dir.rstrip("/")),
    }, indent=1))


if __name__ == "__main__":
    main()
