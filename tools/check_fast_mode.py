#!/usr/bin/env python3
"""WGS_MODE_FAST against WGS_MODE_EXACT (itself bit-pinned to the reference) at FULL size on the device:
configs[2] (10M x 1000, K=10: --get_reference_af + --get_pop_like) and configs[3] (2M x 500, K=8: --loo).
Prints iteration counts, the worst relative deviation of the clamped frequencies and of every n x K sum,
and the timings of both modes.   python tools/check_fast_mode.py [c3] [c4]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import device, glassy  # noqa: E402
from wgsassign_amd._lib import MODE_EXACT, MODE_FAST  # noqa: E402

SEED = 20260313


def blocks_of(n, K):
    return np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.abs(b)))


def c3():
    m, n, K = 10_000_000, 1000, 10
    group_of = blocks_of(n, K)
    b = device.DeviceBeagle(m, n, group_of, K)
    b.synth(SEED, 2.0)
    out = {"config": "10M x 1000, K=10"}
    fits = {}
    for name, mode in (("exact", MODE_EXACT), ("fast", MODE_FAST)):
        em = device.EMBatch(b, np.arange(K, dtype=np.int32), mode=mode)
        t0 = time.perf_counter()
        iters = em.run(200, 1e-4)
        b.ctx.sync()
        out[name + "_fit_seconds"] = round(time.perf_counter() - t0, 4)
        out[name + "_iters"] = [int(x) for x in iters]
        cols = []
        for k in range(K):
            em.clamp(k, n // K)
            cols.append(em.get_f(k))
        fits[name] = (em, np.stack(cols))
    fe, ff = fits["exact"][1], fits["fast"][1]
    d = np.abs(ff.astype(np.float64) - fe) / fe
    out["af_max_rel"] = float(d.max())
    out["af_frac_above_1e-6"] = float(np.mean(d > 1e-6))
    out["af_p999999_rel"] = float(np.quantile(d, 0.999999))
    afs = device.AFSet(m, K)
    for k in range(K):
        afs.set_column_from_em(k, fits["exact"][0], k)
    b.ctx.sync()
    sums = {}
    for name, mode in (("exact", MODE_EXACT), ("fast", MODE_FAST)):
        o, _ = device.assign(b, afs, mode=mode)
        sums[name] = o
        out[name + "_assign_ms"] = round(device.assign.last_ms, 2)
    out["sums_max_rel"] = rel(sums["fast"], sums["exact"])
    out["sums_max_rel_float32"] = rel(sums["fast"].astype(np.float32), sums["exact"].astype(np.float32))
    for em, _ in fits.values():
        em.close()
    afs.close()
    b.close()
    return out


def c4():
    m, n, K = 2_000_000, 500, 8
    group_of = blocks_of(n, K)
    b = device.DeviceBeagle(m, n, group_of, K)
    b.synth(SEED + 4, 2.0)
    em = device.EMBatch(b, np.arange(K, dtype=np.int32))
    em.run(200, 1e-4)
    counts = np.bincount(group_of, minlength=K)
    af0 = np.empty((m, K), dtype=np.float32)
    for k in range(K):
        em.clamp(k, int(counts[k]))
        af0[:, k] = em.get_f(k)
    em.close()
    out = {"config": "2M x 500, K=8 --loo"}
    res = {}
    for name in ("exact", "fast"):
        os.environ["WGSASSIGN_MODE"] = os.environ["WGSASSIGN_EM_MODE"] = name
        os.environ["WGSASSIGN_PARTS"] = name
        af = af0.copy()
        tm = {}
        t0 = time.perf_counter()
        ll, _ = glassy.loo_device(b, b, af, group_of, 200, 1e-4, 1, verbose=False, timings=tm, need_parts=False)
        out[name + "_loo_seconds"] = round(time.perf_counter() - t0, 3)
        res[name] = (ll, af, tm["iters"])
    out["iters_identical"] = bool(np.array_equal(res["exact"][2], res["fast"][2]))
    out["iters_differing"] = int(np.sum(res["exact"][2] != res["fast"][2]))
    out["loo_sums_max_rel"] = rel(res["fast"][0], res["exact"][0])
    d = np.abs(res["fast"][1].astype(np.float64) - res["exact"][1]) / res["exact"][1]
    out["af_after_max_rel"] = float(d.max())
    out["af_after_frac_above_1e-6"] = float(np.mean(d > 1e-6))
    b.close()
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c4"]
    for w in which:
        print(json.dumps({"c3": c3, "c4": c4}[w]()), flush=True)
