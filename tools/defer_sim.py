"""Offline experiment (CPU, oracle traces of tools/sim_traces.py): the march loop runs its probe block and its sample block back to back whenever
a wave holds lanes of both kinds.  What if the FEW lanes of the expensive kind waited (kept their state, did nothing) while the others go on,
until enough of them have gathered?  Per wave iteration: head H + probe block K (if it runs) + sample block S (if it runs), in VALU
instructions of the shipped loop (mixed 118 = H + K + S, dense 93 = H + S).  Policy defer(theta, hold): the sample block is skipped in an
iteration with fewer than theta sampling lanes while probing lanes exist, but no lane waits more than `hold` iterations in a row.
usage: defer_sim.py [H K S]"""
import sys
import numpy as np

H, K, S = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (18, 25, 75)
REGION = 32


def waves(path):
    z = np.load(path)
    offs, lens, data = z["offs"], z["lens"], z["data"]
    n_regions = len(z["regions"])
    for r in range(n_regions):
        for wy in range(REGION // 8):
            for wx in range(REGION // 8):
                lanes = []
                for ly in range(8):
                    for lx in range(8):
                        i = r * REGION * REGION + (wy * 8 + ly) * REGION + wx * 8 + lx
                        if lens[i]:
                            lanes.append(data[offs[i]:offs[i] + lens[i]] >= 2)  # True = sample-kind event
                if lanes:
                    yield lanes


def run(lanes, theta, hold):
    pos = [0] * len(lanes)
    held = [0] * len(lanes)
    n = [len(l) for l in lanes]
    cost = iters = lane_events = 0
    runs_p = runs_s = 0
    while True:
        act = [i for i in range(len(lanes)) if pos[i] < n[i]]
        if not act:
            break
        smp = [i for i in act if lanes[i][pos[i]]]
        prb = [i for i in act if not lanes[i][pos[i]]]
        defer = theta > 0 and prb and 0 < len(smp) < theta and all(held[i] < hold for i in smp)
        c = H
        if prb:
            c += K
            runs_p += 1
            for i in prb:
                pos[i] += 1
        if smp and not defer:
            c += S
            runs_s += 1
            for i in smp:
                pos[i] += 1
                held[i] = 0
        elif defer:
            for i in smp:
                held[i] += 1
        cost += c
        iters += 1
    return cost, iters, runs_p, runs_s


def main():
    ws = []
    for az in (0, 135):
        ws += list(waves("/tmp/sim/traces_%d.npz" % az))
    print("%d waves, H %d K %d S %d" % (len(ws), H, K, S))
    base = None
    for theta, hold in [(0, 0), (4, 2), (4, 4), (8, 2), (8, 4), (8, 8), (16, 2), (16, 4), (16, 8), (16, 16), (24, 4), (24, 8), (32, 4), (32, 8), (32, 16), (64, 8), (64, 1 << 30)]:
        tot = np.zeros(4, np.int64)
        longest = 0
        for l in ws:
            r = run(l, theta, hold)
            tot += r
        if base is None:
            base = tot.copy()
        print("theta %2d hold %10d: cost %.3f x, wave iterations %.3f x, probe-block runs %.3f x, sample-block runs %.3f x" % (
            theta, hold, tot[0] / base[0], tot[1] / base[1], tot[2] / base[2], tot[3] / base[3]))
    print("baseline: %.1f %% of the iterations run both blocks" % (100.0 * (base[2] + base[3] - base[1]) / base[1]))


main()
