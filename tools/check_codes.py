#!/usr/bin/env python3
"""The kernels that go through the class codes (csrc/common.h: wgs_codes) against the direct kernels, at a BASELINE
configuration on device-generated data: identical bits, and the timings of both.
   python tools/check_codes.py [snps inds pops]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import device  # noqa: E402

SEED = 20260313


def main():
    m, n, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10_000_000, 1000, 10)
    ctx = device.get_context()
    group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    b = device.DeviceBeagle(m, n, group_of, K)
    b.synth(SEED, 2.0)
    ctx.sync()
    res = {"config": "%d x %d, K=%d" % (m, n, K)}
    os.environ["WGSASSIGN_CODES"] = "0"
    em = device.EMBatch(b, np.arange(K, dtype=np.int32))
    t0 = time.perf_counter()
    iters = em.run(200, 1e-4)
    ctx.sync()
    res["fit_direct"] = {"seconds": round(time.perf_counter() - t0, 4), "iters": [int(x) for x in iters], "sweep_ms": round(em.fit_stats()[3], 2)}
    afs = device.AFSet(m, K)
    f_direct = []
    for k in range(K):
        em.clamp(k, int(counts[k]))
        afs.set_column_from_em(k, em, k)
        f_direct.append(em.get_f(k))
    ctx.sync()
    out0, _ = device.assign(b, afs)
    out0, _ = device.assign(b, afs)
    res["pop_like_direct_ms"] = round(device.assign.last_ms, 3)
    os.environ["WGSASSIGN_CODES"] = "1"
    t0 = time.perf_counter()
    res["codes"] = b.codes_info()
    res["codes"]["seconds_incl_info"] = round(time.perf_counter() - t0, 3)
    out1, _ = device.assign(b, afs)
    out1, _ = device.assign(b, afs)
    res["pop_like_coded_ms"] = round(device.assign.last_ms, 3)
    res["pop_like_identical"] = bool(out0.tobytes() == out1.tobytes())
    from wgsassign_amd._lib import MODE_FAST
    fast1, _ = device.assign(b, afs, mode=MODE_FAST)
    fast1, _ = device.assign(b, afs, mode=MODE_FAST)
    res["pop_like_coded_fast_ms"] = round(device.assign.last_ms, 3)
    res["variants"] = {}
    for v in os.environ.get("CHECK_CODES_VARIANTS", "").split(","):       # WGS_SCORE_CODED_TABLE values to compare: float,double
        if not v:
            continue
        os.environ["WGS_SCORE_CODED_TABLE"] = v
        o, _ = device.assign(b, afs)
        o, _ = device.assign(b, afs)
        ms = device.assign.last_ms
        f, _ = device.assign(b, afs, mode=MODE_FAST)
        f, _ = device.assign(b, afs, mode=MODE_FAST)
        res["variants"][v] = {"exact_ms": round(ms, 3), "identical": bool(o.tobytes() == out0.tobytes()),
                              "fast_ms": round(device.assign.last_ms, 3), "fast_identical": bool(f.tobytes() == fast1.tobytes())}
    os.environ.pop("WGS_SCORE_CODED_TABLE", None)
    em2 = device.EMBatch(b, np.arange(K, dtype=np.int32))
    t0 = time.perf_counter()
    iters2 = em2.run(200, 1e-4)
    ctx.sync()
    res["fit_coded"] = {"seconds": round(time.perf_counter() - t0, 4), "iters": [int(x) for x in iters2], "sweep_ms": round(em2.fit_stats()[3], 2)}
    same = True
    for k in range(K):
        em2.clamp(k, int(counts[k]))
        same &= em2.get_f(k).tobytes() == f_direct[k].tobytes()
    res["fit_identical"] = bool(same)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
