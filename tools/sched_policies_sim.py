"""Offline experiment (CPU, oracle traces from tools/sim_traces.py): SIMD cycles of the ray-march loop under different intra-workgroup
scheduling policies, priced with the per-block issue costs of the shipped loop (tools/isa_loop_stats.py + tools/micro/valu_mix.hip):
  head 53 (position, cell index, probe test) + probe block 103 + sample block 286 (address 67 + filter / TF 168 + state update 51).
A block costs the SIMD the same whether 1 or 64 lanes of the wave run it, so the policies differ in HOW MANY block executions the
frame's fixed event sequence needs.
usage: sched_policies_sim.py [az ...]"""
import sys, os
import numpy as np

H, P, S_CORE, S_TAIL = 53.0, 103.0, 235.0, 51.0
S = S_CORE + S_TAIL
REGION = 32


def load(az):
    z = np.load("/tmp/sim/traces_%g.npz" % az)
    offs, lens, data = z["offs"], z["lens"], z["data"]
    nreg = len(z["regions"])
    per = REGION * REGION
    out = []
    for r in range(nreg):
        ln = lens[r * per:(r + 1) * per]
        if ln.max() == 0:
            continue
        M = np.full((per, int(ln.max()) + 1), -1, np.int8)
        for i in range(per):
            if ln[i]:
                M[i, :ln[i]] = data[offs[r * per + i]:offs[r * per + i] + ln[i]]
        out.append(M)
    return out


def wave_of(per_side=REGION):
    """ray index (row-major in the 32x32 region) -> wave id: 8x8 tiles, numbered so that waves 4g..4g+3 form the 16x16 block g"""
    y, x = np.divmod(np.arange(per_side * per_side), per_side)
    block = (y // 16) * 2 + (x // 16)
    sub = ((y % 16) // 8) * 2 + ((x % 16) // 8)
    return block * 4 + sub


WAVE = wave_of()
NW = 16


def lockstep(M):
    """the shipped kernel: every live lane takes its next event every iteration"""
    isP = (M == 0) | (M == 1)
    isS = (M == 2) | (M == 3)
    cost = its = 0.0
    for w in range(NW):
        sel = WAVE == w
        aP, aS = isP[sel].any(0), isS[sel].any(0)
        live = aP | aS
        cost += (live * H + aP * P + aS * S).sum()
        its += live.sum()
    return cost, its


def ideal(M):
    nP = ((M == 0) | (M == 1)).sum()
    nS = ((M == 2) | (M == 3)).sum()
    return nP / 64.0 * (H + P) + nS / 64.0 * (H + S), (nP + nS) / 64.0


def run_policy(M, policy, G=4, theta=1.0, X=40.0, K=0):
    """generic stepper.  policy:
       'burst'   per wave: the sample block runs only when >= theta of the wave's live lanes wait for it (or nobody probes)
       'shared'  groups of G waves: all sample requests of the group are served by ceil(n / 64) sample-core executions per step (synchronous)
       'shared_async'  the same, but only full batches of 64 are served unless no lane of the group can probe
       'compact' lockstep + the live rays of a group of G waves re-packed into as few waves as possible every K iterations"""
    n, L = M.shape
    ptr = np.zeros(n, np.int64)
    lens = (M >= 0).sum(1)
    wave = WAVE.copy()
    cost = its = 0.0
    step = 0
    rows = np.arange(n)
    while True:
        alive = ptr < lens
        if not alive.any():
            break
        k = np.where(alive, M[rows, np.minimum(ptr, L - 1)], -1)
        wantP = (k == 0) | (k == 1)
        wantS = (k == 2) | (k == 3)
        if policy == 'compact' and K and step % K == 0:
            # ideal re-pack inside each group of G waves: live rays first, in their current order
            for g in range(NW // G):
                sel = np.where((WAVE // G) == g)[0]
                order = sel[np.argsort(~alive[sel], kind='stable')]
                wave[order] = g * G + np.arange(len(sel)) // 64
            cost += 0.0  # exchange cost added by the caller per re-pack
        nP = np.bincount(wave[wantP], minlength=NW)
        nS = np.bincount(wave[wantS], minlength=NW)
        live = nP + nS
        if policy in ('burst',):
            execS = (nS > 0) & ((nS >= theta * live) | (nP == 0))
            execP = nP > 0
            cost += ((live > 0) * H + execP * P + execS * S).sum()
            its += (live > 0).sum()
            adv = (wantP & execP[wave]) | (wantS & execS[wave])
        elif policy == 'compact':
            cost += ((live > 0) * H + (nP > 0) * P + (nS > 0) * S).sum()
            its += (live > 0).sum()
            adv = alive
        elif policy == 'shared':
            gS = nS.reshape(-1, G).sum(1)
            cost += ((live > 0) * H + (nP > 0) * P + (nS > 0) * (X + S_TAIL)).sum() + (np.ceil(gS / 64.0) * S_CORE).sum()
            its += (live > 0).sum()
            adv = alive
        elif policy == 'shared_async':
            gS = nS.reshape(-1, G).sum(1)
            gP = nP.reshape(-1, G).sum(1)
            batches = np.where(gP == 0, np.ceil(gS / 64.0), np.floor(gS / (64.0 * theta))).astype(np.int64)
            batches = np.minimum(batches, np.ceil(gS / 64.0).astype(np.int64))
            served = np.minimum(batches * 64, gS)
            # requests are served oldest-wave-first inside the group (order does not matter for the cost)
            adv = wantP.copy()
            for g in range(NW // G):
                if served[g] > 0:
                    idx = np.where(wantS & ((wave // G) == g))[0][:served[g]]
                    adv[idx] = True
            anyServed = np.bincount(wave[adv & wantS], minlength=NW) > 0
            act = (nP > 0) | anyServed
            cost += (act * H + (nP > 0) * P + anyServed * (X + S_TAIL)).sum() + (batches * S_CORE).sum()
            its += act.sum()
        ptr = ptr + adv
        step += 1
    return cost, its, step


def main():
    azs = [float(a) for a in sys.argv[1:]] or [0.0, 135.0]
    for az in azs:
        regs = load(az)
        tot = {}
        def add(name, c, i):
            a = tot.setdefault(name, [0.0, 0.0])
            a[0] += c
            a[1] += i
        for M in regs:
            c, i = lockstep(M); add("lockstep (shipped)", c, i)
            c, i = ideal(M); add("ideal (every block execution full)", c, i)
            for th in (0.5, 0.75, 1.0):
                c, i, _ = run_policy(M, 'burst', theta=th); add("burst per wave, theta %.2f" % th, c, i)
            for G in (4, 16):
                for X in (40.0, 80.0):
                    c, i, _ = run_policy(M, 'shared', G=G, X=X); add("shared sample core, %2d waves, sync, exchange %d" % (G, X), c, i)
                c, i, _ = run_policy(M, 'shared_async', G=G, X=40.0, theta=1.0); add("shared sample core, %2d waves, full batches only" % G, c, i)
            for G in (4, 16):
                for K in (4, 16):
                    c, i, st = run_policy(M, 'compact', G=G, K=K); add("compaction in %2d waves every %2d its (exchange free)" % (G, K), c, i)
        base = tot["lockstep (shipped)"][0]
        print("az %g, %d regions" % (az, len(regs)))
        for name, (c, i) in tot.items():
            print("  %-58s cycles %.3e  (%.2fx)  wave-iterations %d" % (name, c, base / c, i))


if __name__ == "__main__":
    main()
