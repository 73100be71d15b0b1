#!/usr/bin/env python3
"""Stress of the block-parallel exact partition chains against the literal one-lane-per-chain kernel on the device:
random shapes (m, n, K, P, site0), carries from a "previous shard", population structures, injected zeros / NaNs /
constant columns / missing-data runs.  Prints one line per case and a summary; exit status 1 on any mismatch.
    python tools/stress_chains.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth  # noqa: E402
from wgsassign_amd import _lib, device  # noqa: E402


def same_nan(a, b):
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    ok = ~np.isnan(a)
    return a[ok].tobytes() == b[ok].tobytes()


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    lib = _lib.load()
    bad = 0
    t0 = time.time()
    for c in range(cases):
        K = int(rng.integers(1, 21))
        sizes = rng.integers(1, 9, size=K)
        labels = np.repeat(np.arange(K), sizes)
        rng.shuffle(labels)
        n = len(labels)
        m = int(rng.choice([1, 63, 4096, 4097, 20_000, 65_537, 200_000, 1_000_003]))
        if m * n > 6e7:
            m = int(6e7 // n)
        P = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 31, 64]))
        site0 = int(rng.choice([0, 1, 12345, 2**33 + 7]))
        L, _ = synth.make_beagle_for_labels(m, labels, K, seed=int(rng.integers(1 << 30)), depth=float(rng.choice([0.3, 2.0, 8.0])))
        A = np.clip(rng.beta(0.8, 0.8, size=(m, K)), 0.004, 0.996).astype(np.float32)
        flavour = int(rng.integers(0, 6))
        if flavour == 1:
            A[:, rng.integers(K)] = np.float32(0.25)                       # constant column: repeated addends
        if flavour == 2 and m > 10:
            A[rng.integers(m), rng.integers(K)] = np.nan                   # NaN from some site on
        if flavour == 3 and m > 10:
            s = int(rng.integers(m))
            A[s, 0] = 1.0
            L[s, 0:2] = (1.0, 0.0)                                         # likelihood exactly 0 -> -inf
        if flavour == 4:
            L[::3] = np.float32(0.333333)                                  # runs of missing data
        carry = None
        if rng.random() < 0.5:
            carry = -(rng.random((n * P, K)) * 10.0 ** rng.integers(0, 7)).astype(np.float32)
        per_ind = rng.random() < 0.5
        b = device.DeviceBeagle.from_host(L, labels.astype(np.int32), K, site0=site0)
        afs = device.AFSet.from_host(A)
        colptr, keep = None, None
        if per_ind:                                                         # per-individual columns: a second set, permuted
            afs2 = device.AFSet.from_host(np.ascontiguousarray(A[:, ::-1]))
            keep = afs2
            tab = np.empty((n, K), dtype=np.uint64)
            for i in range(n):
                for k in range(K):
                    tab[i, k] = (afs2 if (i + k) % 2 else afs).col_dev((k + i) % K)
            colptr = tab
        arr, cp = device._colptr_arg(colptr, n, K)
        got = np.zeros((n * P, K), dtype=np.float32)
        want = np.zeros((n * P, K), dtype=np.float32)
        with np.errstate(all="ignore"):
            _lib.check(lib.wgs_assign_parts_exact(b.handle, afs.handle, cp, P, _lib.f32p(carry) if carry is not None else None, _lib.f32p(got)))
            _lib.check(lib.wgs_debug_parts_exact_literal(b.handle, afs.handle, cp, P, _lib.f32p(carry) if carry is not None else None, _lib.f32p(want)))
        ok = same_nan(got, want)
        bad += not ok
        print("case %3d  m=%8d n=%3d K=%2d P=%2d site0=%d flavour=%d carry=%d per_ind=%d  %s" %
              (c, m, n, K, P, site0, flavour, carry is not None, per_ind, "ok" if ok else "MISMATCH"), flush=True)
        for x in (afs, b) + ((keep,) if keep is not None else ()):
            x.close()
    print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
