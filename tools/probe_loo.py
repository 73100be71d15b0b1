#!/usr/bin/env python3
"""Leave-one-out re-fits through the class codes (em_coded_kernel) against the float32 group kernel (em_sweep_group_kernel): seconds per
   phase and identical bits.   python tools/probe_loo.py [snps inds pops [partitions]]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import device, glassy  # noqa: E402


def main():
    m, n, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (2_000_000, 500, 8)
    P = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    ctx = device.get_context()
    group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    b = device.DeviceBeagle(m, n, group_of, K)
    b.synth(20260313, 2.0)
    ctx.sync()
    em = device.EMBatch(b, np.arange(K, dtype=np.int32))
    em.run(200, 1e-4)
    af = np.stack([(em.clamp(k, int(counts[k])), em.get_f(k))[1] for k in range(K)], axis=1)
    em.close()
    res, outs = {}, {}
    legs = (("float32", "0"), ("coded_cold", "1"), ("coded_warm", "1"), ("float32_again", "0"))
    if os.environ.get("PROBE_LOO_LEGS"):                     # (profiling: e.g. PROBE_LOO_LEGS=coded_warm,float32_again)
        legs = tuple(x for x in legs if x[0] in os.environ["PROBE_LOO_LEGS"].split(","))
    for leg, env in legs:
        os.environ["WGSASSIGN_LOO_CODES"] = env
        tm = {}
        t0 = time.perf_counter()
        ll, parts = glassy.loo_device(b, b, af.copy(), group_of, 200, 1e-4, P, verbose=False, timings=tm, need_parts=P > 1)
        dt = time.perf_counter() - t0
        it = tm["iters"]
        res[leg] = {"seconds": round(dt, 4), "em_seconds": round(tm.get("em_seconds", 0.0), 4), "em_sweep_kernels_ms": round(tm.get("em_sweep_kernel_ms", 0.0), 2),
                    "score_seconds": round(tm.get("score_seconds", 0.0), 4), "iterations_min_max": [int(it.min()), int(it.max())], "codes_state": b.codes_state()}
        outs[leg] = (ll.tobytes(), None if parts is None else parts.tobytes(), it.tobytes())
    res["identical"] = bool(len(set(outs.values())) == 1)
    if b.codes_state() == 1:
        info = b.codes_info()
        res["codes"] = {k: info[k] for k in ("build_ms", "em_table_rows", "em_direct_tile_share", "sample_mean_classes_per_slab", "mean_classes")}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
