"""Opcode-weighted VALU issue cost per SQ counter class, for the bench line's `roofline.valu_issue_frac` (VERDICT r3 'Next #3').

The gfx9 VALUBusy formula (SQ_ACTIVE_INST_VALU x 4 / SIMDs / cycles) prices every wave64 VALU instruction at 4 cycles; on gfx950 a SIMD
issues v_fma / v_add / v_mul / logic / v_mov in 2.4 - 2.9 cycles and every other opcode this kernel uses in 4.1 - 4.5
(tools/micro/valu_mix.hip at 8 waves per SIMD, profiles/r2_micro_valu_mix.txt).  rocprofv3 splits SQ_INSTS_VALU into CVT, FMA_F32, INT32 and
the rest; this tool prices each class with the average measured cost of the opcodes of that class in the march loop of the shipped kernel
(static mix of the loop body in a `hipcc -S` listing), so that

    valu_issue_frac = sum_class(N_class x cost_class) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)

usage: valu_issue_model.py [listing.s]   (without a listing: compiles k_raymarch_lean_batch<DISTANCE, ERT, gradient map, full tables, no counters>)
prints a JSON object {class: {"instructions": n, "cycles_per_instruction": c}, ...}."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def measured_costs():
    cost = {}
    for line in open(os.path.join(ROOT, "profiles", "r2_micro_valu_mix.txt")):
        m = re.match(r"(v_\w+)(?: \(([^)]*)\))?\s+[\d.]+ ms -> ([\d.]+) cycles", line)
        if m and not (m.group(1) == "v_cndmask_b32" and m.group(2) == "vcc"):  # (the vcc form of that micro-benchmark serialises on VCC: not how the kernel uses it)
            cost.setdefault(m.group(1), float(m.group(3)))
    return cost


def classify(op):
    if op.startswith("v_cvt_"):
        return "CVT"
    if op in ("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mad_f32", "v_mac_f32"):
        return "FMA_F32"
    if re.match(r"v_(add|sub|subrev|mul|mad|min|max|med3|add3|lshl_add|mul_lo|mul_hi|bfe|sad)\w*_(u32|i32|u24|i24|u32_u24|i32_i24)\b", op) or op in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"):
        return "INT32"
    return "OTHER"


def march_loop(listing, name_filter):
    s = open(listing).read()
    for f in re.split(r"\n(?=_Z\w+:)", s):
        name = f.split(":")[0]
        if not name.startswith("_Z") or name_filter not in name:
            continue
        labels, ins = {}, []
        for l in f.split(".Lfunc_end")[0].split("\n"):
            t = l.strip()
            if not t or t.startswith(";") or (t.startswith(".") and not t.startswith(".LBB")):
                continue
            if t.startswith(".LBB") and t.split()[0].endswith(":"):
                labels[t.split(":")[0]] = len(ins)
                continue
            if t.endswith(":"):
                continue
            ins.append(t.split(";")[0].strip())
        loops = []
        for i, t in enumerate(ins):
            mm = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", t)
            if mm:
                lab = mm.group(1) or mm.group(2)
                if lab in labels and labels[lab] <= i:
                    loops.append((labels[lab], i))
        loads = lambda a, b: sum(1 for x in ins[a:b + 1] if x.startswith(("global_load", "buffer_load", "flat_load")))
        cand = [(a, b) for a, b in loops if loads(a, b) >= 4]
        cand = [(a, b) for a, b in cand if not any((a <= a2 and b2 <= b) and (a2, b2) != (a, b) for a2, b2 in cand)]
        if cand:
            # a kernel holds one copy of the march loop per transfer-function path; the bench's TF is the reference's separable product, whose
            # copy reads no RGBA texel (the generic copy's dependent texel fetch is the loop's only flat_load)
            sep = [(a, b) for a, b in cand if not any(x.startswith("flat_load") for x in ins[a:b + 1])]
            # round 5: a kernel of the separable path holds the clamp-free loop (the one waves of free rays run: it computes r with v_fract_f32)
            # next to the loop with the clamps; the bench's frames run the former
            free = [(a, b) for a, b in (sep or cand) if any(x.startswith("v_fract_f32") for x in ins[a:b + 1])]
            if free:        # (its rare clamp block sits behind the loop and branches back into it: the shortest region is the loop proper)
                a, b = min(free, key=lambda ab: ab[1] - ab[0])
            else:
                a, b = max(sep or cand, key=lambda ab: ab[1] - ab[0])
            return name, ins[a:b + 1]
    raise SystemExit("no march loop found for %r" % name_filter)


def main():
    if len(sys.argv) > 1:
        listing, name_filter = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_raymarch_lean_batch"
    else:
        d = tempfile.mkdtemp()
        src = os.path.join(d, "k.hip")
        open(src, "w").write('#include "%s/vkvolume_amd/csrc/raymarch_inst.hpp"\n'
                             "template __global__ void k_raymarch_lean_batch<VKV_SKIP_DISTANCE, true, 1, vkv::kLfFullNc>(const RayMarchArgs *__restrict__, uint32_t, uint32_t);\n" % ROOT)
        listing, name_filter = os.path.join(d, "k.s"), "k_raymarch_lean_batch"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
                               "-fno-fast-math", "--cuda-device-only", "-S", src, "-o", listing], stderr=subprocess.DEVNULL)
    cost = measured_costs()
    name, loop = march_loop(listing, name_filter)
    acc = {}
    unknown = set()
    for x in loop:
        if not x.startswith("v_"):
            continue
        op = x.split()[0]
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if base in cost:
            c = cost[base]
        elif base.startswith("v_cmp"):
            c = cost["v_cmp_lt_f32"]
        elif base.startswith("v_pk_"):
            c = cost["v_pk_add_f32"]
        else:
            c = 4.2  # every opcode of this kernel outside the full-rate list measured 4.1 - 4.5
            unknown.add(base)
        if op.endswith("_sdwa") or op.endswith("_dpp"):
            c = max(c, cost["v_sub_u32_sdwa"])
        k = classify(base)
        n, t = acc.get(k, (0, 0.0))
        acc[k] = (n + 1, t + c)
    out = {k: {"instructions": n, "cycles_per_instruction": round(t / n, 3)} for k, (n, t) in sorted(acc.items())}
    out["_kernel"] = name
    out["_loop_valu_instructions"] = sum(n for n, _ in acc.values())
    out["_loop_valu_cycles"] = round(sum(t for _, t in acc.values()), 1)
    out["_opcodes_priced_at_the_half_rate_default"] = sorted(unknown)
    out["_source"] = "static mix of the kernel's march loop x profiles/r2_micro_valu_mix.txt (tools/micro/valu_mix.hip, 8 waves per SIMD)"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
