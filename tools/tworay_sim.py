"""Offline experiment (CPU, oracle traces of tools/sim_traces.py): two rays per lane marched one after the other (a wave covers 8x16 or 16x8 pixels;
a lane whose first ray has ended starts its second), priced like tools/sched_policies_sim.py.  Result (round 4): 1.02x with a FREE switch, 0.85x when
every iteration in which some lane switches pays a ray set-up, 0.96x with two synchronised switch points - the rays of an 8x8 tile end too close
together for this to matter; the idle lanes of the shipped loop are lanes in the OTHER phase (probe / sample), not dead ones.
usage: tworay_sim.py [set-up cycles]"""
import sys, numpy as np
sys.path.insert(0, '/root/repo/tools')
import sched_policies_sim as SP
H, P, S = SP.H, SP.P, SP.S
SETUP = float(sys.argv[1]) if len(sys.argv) > 1 else 500.0   # ray set-up + finish of one 8x8 unit, SIMD cycles (149 + ~40 VALU, ~3 cycles each)
REGION = 32

def seqs(M):
    lens = (M >= 0).sum(1)
    return lens

def cost_lockstep_rows(rows):
    """rows: (64, L) int8 padded with -1: lock-step cost of one wave"""
    isP = (rows == 0) | (rows == 1); isS = (rows == 2) | (rows == 3)
    aP, aS = isP.any(0), isS.any(0)
    live = aP | aS
    return (live * H + aP * P + aS * S).sum(), live.sum()

def concat(A, B):
    """per lane: events of A then events of B"""
    la, lb = (A >= 0).sum(1), (B >= 0).sum(1)
    L = int((la + lb).max()) + 1
    out = np.full((64, L), -1, np.int8)
    for l in range(64):
        out[l, :la[l]] = A[l, :la[l]]
        out[l, la[l]:la[l] + lb[l]] = B[l, :lb[l]]
    return out, la

tot = {}
def add(k, c, i=0):
    a = tot.setdefault(k, [0.0, 0.0]); a[0] += c; a[1] += i
for az in (0.0, 135.0):
    for M in SP.load(az):
        y, x = np.divmod(np.arange(REGION * REGION), REGION)
        # baseline: 8x8 waves, each sets up once
        for ty in range(4):
            for tx in range(4):
                sel = ((y // 8) == ty) & ((x // 8) == tx)
                rows = M[sel]
                if (rows >= 0).any():
                    c, i = cost_lockstep_rows(rows); add("shipped: 8x8 wave, one ray per lane", c + SETUP, i)
                    add("shipped, loop only", c, i)
        # two rays per lane: wave covers 8x16 (A = upper 8x8, B = lower) or 16x8
        for name, pairs in (("8x16", [((2 * ty, tx), (2 * ty + 1, tx)) for ty in range(2) for tx in range(4)]),
                            ("16x8", [((ty, 2 * tx), (ty, 2 * tx + 1)) for ty in range(4) for tx in range(2)])):
            for (a, b) in pairs:
                A = M[((y // 8) == a[0]) & ((x // 8) == a[1])]; B = M[((y // 8) == b[0]) & ((x // 8) == b[1])]
                if not ((A >= 0).any() or (B >= 0).any()):
                    continue
                rows, la = concat(A, B)
                c, i = cost_lockstep_rows(rows)
                lb = (B >= 0).sum(1)
                switch_its = len(set(int(v) for v, w in zip(la, lb) if w > 0 and v > 0))   # iterations in which some lane starts its second ray
                add("two rays per lane %s, switch free" % name, c + SETUP, i)
                add("two rays per lane %s, every switch iteration costs a set-up" % name, c + SETUP * (1 + switch_its), i)
                # switches only at sync points: when half of the lanes are done with A, and when all are
                # lanes done with A wait (idle) until the sync point
                order = np.sort(la)
                t1, t2 = int(order[31]), int(order[63])
                la2 = np.where(la <= t1, t1, t2)
                L = int((la2 + lb).max()) + 1
                rows2 = np.full((64, L), -1, np.int8)
                for l in range(64):
                    rows2[l, :la[l]] = A[l, :la[l]]
                    rows2[l, la2[l]:la2[l] + lb[l]] = B[l, :lb[l]]
                c2, i2 = cost_lockstep_rows(rows2)
                add("two rays per lane %s, switch at 2 sync points (half done / all done)" % name, c2 + SETUP * 3, i2)
base = tot["shipped: 8x8 wave, one ray per lane"][0]
for k, (c, i) in tot.items():
    print("%-75s cycles %.3e (%.3fx) wave-iterations %d" % (k, c, base / c, i))
