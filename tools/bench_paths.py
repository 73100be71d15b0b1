#!/usr/bin/env python3
"""Whole-path timings on device-generated data (not the driver's bench.py contract):
   full --get_reference_af fit (all populations to convergence), the exact convergence chain,
   --get_pop_like, and --loo, at a BASELINE.json configuration.

   python tools/bench_paths.py --snps 2000000 --inds 500 --pops 8 [--loo]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import device, glassy  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=2_000_000)
    ap.add_argument("--inds", type=int, default=500)
    ap.add_argument("--pops", type=int, default=8)
    ap.add_argument("--loo", action="store_true")
    ap.add_argument("--partitions", type=int, default=1)
    ap.add_argument("--exact-parts", action="store_true", help="also time the bit-exact partition chains when --partitions 1")
    a = ap.parse_args()
    m, n, K = a.snps, a.inds, a.pops
    ctx = device.get_context()
    group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    b = device.DeviceBeagle(m, n, group_of, K)
    b.synth(20260313, 2.0)
    ctx.sync()
    res = {"config": {"snps": m, "inds": n, "pops": K}, "device": ctx.info()["name"]}

    t0 = time.perf_counter()
    em = device.EMBatch(b, np.arange(K, dtype=np.int32))
    iters = em.run(200, 1e-4)
    ctx.sync()
    t_fit = time.perf_counter() - t0
    res["reference_af_fit"] = {"seconds": round(t_fit, 4), "iters": [int(x) for x in iters],
                               "snp_updates_per_s": float(m) * float(np.sum(iters)) / t_fit,
                               "iterations_enqueued, chain_batches, seconds_in_wgs_em_fit, sweep_kernels_ms": list(em.fit_stats())}
    t0 = time.perf_counter()
    c = em.rmse_chain(0, 0.0)
    res["rmse_chain"] = {"seconds": round(time.perf_counter() - t0, 5), "diff": device.chain_diff(c, m)}

    af = np.empty((m, K), dtype=np.float32)
    for k in range(K):
        em.clamp(k, int(counts[k]))
        af[:, k] = em.get_f(k)
    em.close()
    afs = device.AFSet.from_host(af)
    t0 = time.perf_counter()
    out, _ = device.assign(b, afs)
    res["pop_like"] = {"seconds": round(time.perf_counter() - t0, 4), "kernel_ms": round(device.assign.last_ms, 2),
                       "snps_per_s": m / (device.assign.last_ms * 1e-3), "self_assign_accuracy":
                       float(np.mean(np.argmax(out, axis=1) == group_of))}
    afs.close()
    if a.loo:
        tm, af0 = {}, af.copy()
        t0 = time.perf_counter()
        ll, parts = glassy.loo_device(b, b, af, group_of, 200, 1e-4, a.partitions, verbose=False, timings=tm,
                                      need_parts=a.partitions > 1 or a.exact_parts)      # as the command line does
        res["loo"] = {"seconds": round(time.perf_counter() - t0, 3), "fits": n,
                      "iters_min_max": [int(tm["iters"].min()), int(tm["iters"].max())],
                      "accuracy": float(np.mean(np.argmax(ll, axis=1) == group_of)), "one_call": bool(tm.get("one_call"))}
        # the same through the step-wise entry points, for the breakdown (EM batch / scoring kernels)
        os.environ["WGSASSIGN_LOO"] = "python"
        tm2, af2 = {}, af0.copy()
        t0 = time.perf_counter()
        ll2, parts2 = glassy.loo_device(b, b, af2, group_of, 200, 1e-4, a.partitions, verbose=False, timings=tm2,
                                        need_parts=a.partitions > 1 or a.exact_parts)
        res["loo_stepwise"] = {"seconds": round(time.perf_counter() - t0, 3), "em_seconds": round(tm2["em_seconds"], 3),
                               "score_seconds": round(tm2["score_seconds"], 3),
                               "score_kernels_ms": {k: round(v, 3) for k, v in tm2.items() if k.endswith("_ms")},
                               "chain_blocks_serial_of_walked": tm2.get("serial_blocks"),
                               "identical_to_one_call": bool(ll.tobytes() == ll2.tobytes() and parts.tobytes() == parts2.tobytes()
                                                             and af.tobytes() == af2.tobytes())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
