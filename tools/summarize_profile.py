"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into a small text summary for profiles/.

    python tools/summarize_profile.py <tag> <stats_dir> [<pmc_dir> ...]
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")[:60]


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = ["# rocprofv3 summary '%s'" % tag, ""]
    for f in glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True):
        lines.append("## kernel stats (rocprofv3 --kernel-trace --stats): %s" % os.path.relpath(f))
        lines.append("%-62s %6s %14s %12s %7s %12s %12s" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"))
        for r in csv.DictReader(open(f)):
            if r["Name"].startswith(("k_", "void k_")):
                lines.append("%-62s %6s %14s %12.0f %7s %12s %12s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                                                     r["Percentage"], r["MinNs"], r["MaxNs"]))
        lines.append("")
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(list)
            regs = {}
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith(("k_", "void k_")):
                    acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
                    regs[short(r["Kernel_Name"])] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
            lines.append("## PMC (own pass): %s" % os.path.relpath(f))
            lines.append("%-62s %-14s %6s %16s %16s %16s" % ("kernel", "counter", "n", "avg", "min", "max"))
            for (k, c), v in sorted(acc.items()):
                lines.append("%-62s %-14s %6d %16.1f %16.1f %16.1f" % (k, c, len(v), sum(v) / len(v), min(v), max(v)))
            lines.append("registers (VGPR, SGPR, LDS bytes, scratch): " + "; ".join("%s=%s" % kv for kv in sorted(regs.items())))
            lines.append("")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
