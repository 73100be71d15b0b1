#!/usr/bin/env python3
"""Stress of wgs_em_fit (iterations enqueued ahead of the host, device-side decisions, parked fits) against the
step-by-step protocol of device.run_em on the device: random batches of population fits and leave-one-out fits,
random tolerances / iteration caps (exhaustion), guard floors that park every decision or none, special genotype
likelihoods (all-hom-ref SNPs -> f = 0 -> 0/0 = NaN patterns).  Iteration counts, activity flags and every frequency
vector must be identical bit for bit.    python tools/stress_em_fit.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth  # noqa: E402
from wgsassign_amd import device  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    bad = 0
    t0 = time.time()
    for c in range(cases):
        K = int(rng.integers(1, 9))
        sizes = rng.integers(1, 12, size=K)
        labels = np.repeat(np.arange(K), sizes)
        rng.shuffle(labels)
        n = len(labels)
        m = int(rng.choice([1, 64, 65, 1000, 4097, 50_000, 300_000]))
        L, _ = synth.make_beagle_for_labels(m, labels, K, seed=int(rng.integers(1 << 30)), depth=float(rng.choice([0.5, 2.0, 10.0])))
        if rng.random() < 0.5 and m > 8:                       # certain hom-ref rows and contradicting individuals: f -> 0, 0/0
            rows = rng.choice(m, size=max(1, m // 50), replace=False)
            L[rows, 0::2], L[rows, 1::2] = 1.0, 0.0
            L[rows[: len(rows) // 2], 0] = 0.0
        b = device.DeviceBeagle.from_host(L, labels.astype(np.int32), K)
        # fits: every population, plus leave-one-out fits of a random subset of individuals
        loo = rng.choice(n, size=int(rng.integers(0, n + 1)), replace=False)
        loo = np.array([i for i in loo if sizes[labels[i]] >= 1], dtype=np.int32)
        groups = np.concatenate([np.arange(K), labels[loo]]).astype(np.int32)
        skips = np.concatenate([-np.ones(K), loo]).astype(np.int32)
        order = rng.permutation(len(groups))
        groups, skips = groups[order], skips[order]
        tole = float(rng.choice([1e-4, 1e-3, 3e-5, 0.0]))
        max_iter = int(rng.choice([1, 2, 7, 40, 200]))
        guard = float(rng.choice([0.0, 0.25, 1e9]))
        res = {}
        for loop in ("c", "python"):
            os.environ["WGSASSIGN_EM_LOOP"] = loop
            device.EMBatch.GUARD = guard
            em = device.EMBatch(b, groups, skips)
            pre = rng.random(len(groups)) < 0.1 if loop == "c" else pre      # some fits frozen beforehand
            for j in np.flatnonzero(pre):
                em.set_active(int(j), False)
            with np.errstate(all="ignore"):
                iters = em.run(max_iter, tole)
            res[loop] = (iters.copy(), np.stack([em.get_f(j) for j in range(len(groups))]), em.active.copy())
            em.close()
        a, p = res["c"], res["python"]
        same_f = np.array_equal(np.isnan(a[1]), np.isnan(p[1])) and a[1][~np.isnan(a[1])].tobytes() == p[1][~np.isnan(p[1])].tobytes()
        ok = np.array_equal(a[0], p[0]) and same_f and np.array_equal(a[2], p[2])
        bad += not ok
        print("case %3d  m=%7d n=%3d K=%d fits=%3d tole=%g max_iter=%3d guard=%g  iters %d..%d  %s" %
              (c, m, n, K, len(groups), tole, max_iter, guard, a[0].min(), a[0].max(), "ok" if ok else "MISMATCH"), flush=True)
        b.close()
    device.EMBatch.GUARD = 0.0
    print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
