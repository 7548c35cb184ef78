"""Build profiles/<round>_traffic.json (what bench.py's roofline.traffic / valu_issue_frac / wait_frac / waves_per_simd replay) from the PMC passes of tools/collect_profiles.sh.

    python tools/make_traffic_json.py gpurun_out/<tag> r3 [workload] > profiles/r3_traffic.json

HBM bytes per launch of the ray-march kernel = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024: FETCH_SIZE tallies every 128-byte request at 64 bytes on
gfx950 (MI355X_MICROARCH.md, HBM) - calibrated on a coalesced stream in round 1 AND on the integrator's own 2-byte-aligned dword gathers in
round 3 (tools/micro/gather_fetch.hip, profiles/r3_micro_gather_fetch.txt: the same factor, traffic counted in whole 128-byte lines).
valu_issue_frac = sum over the SQ counter classes (SQ_INSTS_VALU_CVT, _FMA_F32, _INT32, the rest of SQ_INSTS_VALU) of instructions x the
issue cost measured for the opcodes of that class (tools/valu_issue_model.py: static mix of the shipped march loop x tools/micro/valu_mix.hip,
profiles/r2_micro_valu_mix.txt: 2.4 - 2.9 cycles for v_fma / v_add / v_mul / logic / v_mov, 4.1 - 4.5 for everything else the loop uses),
divided by the launch's SIMD cycles 1024 x GRBM_GUI_ACTIVE / 8 (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs).  It replaces round 3's
valu_busy (the gfx9 formula SQ_ACTIVE_INST_VALU x 4 / SIMDs / cycles, which prices every instruction at 4 cycles: 0.96 for a kernel whose
priced figure is 0.84).  wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES: the share of a resident wave's time parked in s_waitcnt;
waves_per_simd = SQ_WAVE_CYCLES x 4 / (1024 x GRBM_GUI_ACTIVE / 8) (the SQ wave counters tick in quad-cycles, MI355X_MICROARCH.md);
valu_active_lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU: the average number of live lanes (of 64) of a VALU instruction.
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
    workload = sys.argv[3] if len(sys.argv) > 3 else "c3"
    kernel = "k_raymarch_lean_batch"
    avg, n = averages(os.path.join(out_dir, "pmc_batch8_*"), kernel)
    import bench
    digest = bench.kernel_source_digest()
    commit = open(os.path.join(ROOT, ".commit_id")).read().strip() if os.path.exists(os.path.join(ROOT, ".commit_id")) else None
    fetch, write = avg.get("FETCH_SIZE"), avg.get("WRITE_SIZE")
    issue = wait = waves = model = None
    simd_cycles = 1024.0 * avg["GRBM_GUI_ACTIVE"] / 8.0 if avg.get("GRBM_GUI_ACTIVE") else None
    if simd_cycles and all(avg.get(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_INT32")):
        import subprocess
        model = json.loads(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "valu_issue_model.py")], text=True))
        insts = {"CVT": avg["SQ_INSTS_VALU_CVT"], "FMA_F32": avg["SQ_INSTS_VALU_FMA_F32"], "INT32": avg["SQ_INSTS_VALU_INT32"]}
        insts["OTHER"] = avg["SQ_INSTS_VALU"] - sum(insts.values())
        issue = sum(insts[k] * model[k]["cycles_per_instruction"] for k in insts) / simd_cycles
    # average number of live lanes of a VALU instruction (both counters tick in quad-cycles; the second is the first weighted by the threads that ran)
    lanes = avg["SQ_THREAD_CYCLES_VALU"] / avg["SQ_ACTIVE_INST_VALU"] if avg.get("SQ_THREAD_CYCLES_VALU") and avg.get("SQ_ACTIVE_INST_VALU") else None
    if avg.get("SQ_WAIT_ANY") and avg.get("SQ_WAVE_CYCLES"):
        wait = avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"]
        if simd_cycles:
            waves = avg["SQ_WAVE_CYCLES"] * 4.0 / simd_cycles
    print(json.dumps({
        "_comment": " ".join(__doc__.split("\n\n")[2:]).replace("\n", " "),
        "workload": workload, "kernel": kernel, "frames_per_launch": 8,
        "fetch_size_kib_avg": fetch, "write_size_kib_avg": write,
        "traffic_bytes_per_launch": int((fetch * 2 + write) * 1024) if fetch and write else None,
        "valu_issue_frac": round(issue, 4) if issue else None, "wait_frac": round(wait, 4) if wait else None,
        "valu_issue_frac_note": (None if workload == "c3" else "priced with the per-class instruction costs of C3's march loop (the distance-map variant <2, true, 1, 55>); this workload "
                                 "runs another variant of the same loop, the model is good to about +-15 %: a figure near or above 1 says that VALU issue is the limit, not by how much"),
        "waves_per_simd": round(waves, 2) if waves else None, "valu_active_lanes": round(lanes, 2) if lanes else None, "valu_issue_cost_model": model,
        "counters_avg": {k: round(v, 1) for k, v in sorted(avg.items())}, "launches_averaged": n,
        "kernel_source_sha256": digest, "commit": commit,
        "source": "profiles/%s_rocprof.txt" % os.path.basename(out_dir.rstrip("/")),
    }, indent=1))


if __name__ == "__main__":
    main()
