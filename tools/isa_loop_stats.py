"""Static instruction mix of the hot loop of each kernel in a hipcc -S listing (offline, CPU): finds the innermost-but-largest
backward branch region that contains vector memory loads and counts VALU / SALU / VMEM / LDS / DPP instructions in it.
usage: isa_loop_stats.py file.s [name-filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\n(_Z[^\n:]*):[^\n]*\n(.*?)\.end_amdhsa_kernel|\n(_Z[^\n:]*):[^\n]*\n(.*?)s_endpgm', s, re.S):
    pass
funcs = re.split(r'\n(?=_Z\w+:)', s)
for f in funcs:
    name = f.split(':')[0]
    if not name.startswith('_Z') or flt not in name:
        continue
    body = f.split('.Lfunc_end')[0]
    lines = body.split('\n')
    labels, ins = {}, []
    for l in lines:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'):
            continue
        if t.startswith('.LBB') and t.split()[0].endswith(':'):
            labels[t.split(':')[0]] = len(ins)
            continue
        if t.endswith(':'):
            continue
        ins.append(t.split(';')[0].strip())
    loops = []
    for i, t in enumerate(ins):
        mm = re.match(r's_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)', t)
        if mm:
            lab = mm.group(1) or mm.group(2)
            if lab in labels and labels[lab] <= i:
                loops.append((labels[lab], i))
    # outermost loops that hold at least three vector loads and one LDS access: the march loops (one per transfer-function path)
    cand = [(a, b) for a, b in loops if sum(1 for x in ins[a:b + 1] if x.startswith(('global_load', 'buffer_load', 'flat_load'))) >= 3 and any(x.startswith('ds_') for x in ins[a:b + 1])]
    cand = [(a, b) for a, b in cand if not any((a2 <= a and b <= b2) and (a2, b2) != (a, b) for a2, b2 in cand)]
    for best in cand:
      seg = ins[best[0]:best[1] + 1]
      c = lambda p: sum(1 for x in seg if x.startswith(p))
      # SIMD cycles of the loop's VALU instructions at the issue costs measured by tools/micro/valu_mix.hip (8 waves per SIMD)
      full = ("v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32",
              "v_xor_b32", "v_mov_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_not_b32")
      cyc = 0.0
      for x in seg:
          if not x.startswith("v_"): continue
          op = x.split()[0]
          base = op.replace("_e32", "").replace("_e64", "")
          if "sdwa" in op or "dpp" in x: cyc += 4.2
          elif base in full: cyc += 2.5
          elif base in ("v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32"): cyc += 8.3
          elif base.startswith("v_pk_"): cyc += 4.3
          elif base.startswith(("v_mad_u64", "v_mad_i64")): cyc += 8.4
          else: cyc += 4.2
      print("%-70s loop %4d instr: valu %3d = %4.0f SIMD cycles (dpp %2d) salu %3d vmem-load %2d vmem-store %2d lds %2d waitcnt %2d branch %2d" % (
        name[:70], len(seg), c('v_'), cyc, sum(1 for x in seg if 'dpp' in x or 'quad_perm' in x), c('s_') - c('s_waitcnt') - c('s_cbranch') - c('s_branch'),
        c(('global_load', 'buffer_load', 'flat_load')), c(('global_store', 'buffer_store')), c('ds_'), c('s_waitcnt'), c(('s_cbranch', 's_branch'))))
