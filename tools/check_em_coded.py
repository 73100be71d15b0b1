#!/usr/bin/env python3
"""em_coded_kernel ALONE, every launch the same work (for profiles: round 4's coded profile averaged 23 launches of different work):
the class codes are built, then wgs_em_fit runs exactly `steps` iterations (tole = 0) -- two per launch -- and, with
WGSASSIGN_EM_FUSE=1, one per launch.  Prints the kernel's own algorithmic bytes per launch.
   python tools/check_em_coded.py [snps inds pops [steps [fused|single]]]     (a variant name runs only that one: a profile of uniform launches)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import device  # noqa: E402

SEED = 20260313


def main():
    m, n, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10_000_000, 1000, 10)
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
    only = sys.argv[5] if len(sys.argv) > 5 else None
    ctx = device.get_context()
    group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
    b = device.DeviceBeagle(m, n, group_of, K)
    b.synth(SEED, 2.0)
    ctx.sync()
    os.environ["WGSASSIGN_EM_CODES_SWEEPS"] = "0"
    info = b.codes_info()
    res = {"config": "%d x %d, K=%d" % (m, n, K), "steps": steps}
    for label, fuse in (("two_iterations_per_launch", None), ("one_iteration_per_launch", "1")):
        if only and (only == "fused") != (fuse is None):
            continue
        if fuse:
            os.environ["WGSASSIGN_EM_FUSE"] = fuse
        em = device.EMBatch(b, np.arange(K, dtype=np.int32))
        em.fit(steps, 0.0)                      # (warm-up with the same launches: every launch of the profile is the same work)
        em.fit(steps, 0.0)
        ms = em.fit_stats()[3]
        em.close()
        launches = steps if fuse else steps // 2
        # per SNP and launch: a code byte per individual, 8 bytes per class present per slab (the sample's mean), f in (4 K) and out (4 K per iteration)
        its = 1 if fuse else 2
        alg = (float(n) + 8.0 * K * info["sample_mean_classes_per_slab"] + 4.0 * K + 4.0 * K * its) * m
        res[label] = {"launches": launches, "kernel_ms_per_launch": round(ms / launches, 4), "algorithmic_bytes_per_launch": alg,
                      "hbm_frac_of_its_own_bytes": round(alg / (ms / launches * 1e-3) / 8e12, 4)}
    os.environ.pop("WGSASSIGN_EM_FUSE", None)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
