"""The march loop of each ray-march kernel in a `hipcc -S --cuda-device-only` listing: the SMALLEST backward-branch region that holds the
probe byte load and the four footprint loads.  Prints instruction counts by class; with --dump the loop's instructions; with --order
which outcome block comes first in the loops with hand-set load waits (the probe side must: its byte is the oldest load, lean_march) -
exit code 1 if a loop starts with the sample side or if the loads in front of the hand-set wait are not
"probe byte, then the four footprint dwords" (vmcnt counts loads in issue order: the wait releases the oldest).
usage: isa_march_loop.py file.s [name-filter] [--dump] [--order]"""
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
dump = "--dump" in sys.argv
order = "--order" in sys.argv
bad_order = 0
s = open(args[0]).read()
flt = args[1] if len(args) > 1 else ""
for f in re.split(r'\n(?=_Z\w+:)', s):
    name = f.split(':')[0]
    if not name.startswith('_Z') or flt not in name:
        continue
    body = f.split('.Lfunc_end')[0]
    labels, ins = {}, []
    for l in body.split('\n'):
        t = l.strip()
        if not t or t.startswith(';') or (t.startswith('.') and not t.startswith('.LBB')):
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
    def loads(a, b):
        return sum(1 for x in ins[a:b + 1] if x.startswith(('global_load', 'buffer_load', 'flat_load')))
    cand = [(a, b) for a, b in loops if loads(a, b) >= 4]
    # innermost qualifying loops
    cand = [(a, b) for a, b in cand if not any((a <= a2 and b2 <= b) and (a2, b2) != (a, b) for a2, b2 in cand)]
    for a, b in cand:
        seg = ins[a:b + 1]
        c = lambda p: sum(1 for x in seg if x.startswith(p))
        print("%-72s march loop %4d instr: valu %3d salu %3d vmem-load %2d lds %2d waitcnt %2d branch %2d" % (
            name[:72], len(seg), c('v_'), c('s_') - c('s_waitcnt') - c('s_cbranch') - c('s_branch'), c(('global_load', 'buffer_load', 'flat_load')), c('ds_'),
            c('s_waitcnt'), c(('s_cbranch', 's_branch'))))
        if order:
            first = lambda p: next((k for k, x in enumerate(seg) if x.startswith(p)), None)
            w4, cv = first('s_cmp_lg_u64 exec'), first('v_cvt_f32_ubyte')        # (the hand-set wait of the probe byte starts with this comparison)
            if w4 is not None and cv is not None:
                print("    %s side first" % ("probe" if w4 < cv else "SAMPLE"))
                bad_order += w4 > cv
                # the hand-set vmcnt(4) in front of the probe outcome releases the probe BYTE only if that byte is the oldest of the iteration's
                # loads: the vector-memory loads in front of the wait must be exactly one global_load_ubyte followed by the four footprint dwords
                # (nothing else pins the order of the two sides of the load if / else)
                vm = [x.split()[0] for x in seg[:w4] if x.startswith(('global_load', 'buffer_load', 'flat_load', 'global_store', 'buffer_store', 'flat_store'))]
                if vm == ['global_load_ubyte'] + ['global_load_dword'] * 4:
                    print("    loads in front of the wait: probe byte, then four footprint dwords")
                else:
                    print("    LOAD ORDER in front of the wait: %s" % " ".join(vm))
                    bad_order += 1
        if dump:
            print("\n".join("    " + x for x in seg))
if order and bad_order:
    sys.exit(1)
