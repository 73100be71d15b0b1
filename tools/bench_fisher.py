#!/usr/bin/env python3
"""--ne_obs on device-resident data: python tools/bench_fisher.py --snps 2000000 --inds 500 --pops 8"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import device, fisher  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--snps", type=int, default=2_000_000)
ap.add_argument("--inds", type=int, default=500)
ap.add_argument("--pops", type=int, default=8)
a = ap.parse_args()
m, n, K = a.snps, a.inds, a.pops
group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
IDs = np.array([["Ind%d" % i, "pop%02d" % group_of[i]] for i in range(n)])
b = device.DeviceBeagle(m, n, group_of, K)
b.synth(20260313, 2.0)
em = device.EMBatch(b, np.arange(K, dtype=np.int32))
em.run(200, 1e-4)
af = np.stack([em.get_f(k) for k in range(K)], axis=1)
af = np.clip(af, 1 / (2 * (n // K + 1)), 1 - 1 / (2 * (n // K + 1))).astype(np.float32)
em.close()
res = {"config": {"snps": m, "inds": n, "pops": K}}
for name, fn in (("fisher_obs", lambda: fisher.fisher_obs(None, af, IDs, 1, beagle=b)),
                 ("fisher_obs_ind (np.mean formed on the device)", lambda: fisher.fisher_obs_ind(None, af, IDs, 1, beagle=b))):
    fn()
    b.ctx.sync()
    t0 = time.perf_counter()
    out = fn()
    b.ctx.sync()
    res[name] = round(time.perf_counter() - t0, 4)
if m * n <= 4e9:
    t0 = time.perf_counter()
    host = fisher.fisher_obs_ind(None, af, IDs, 1, beagle=b, host_mean=True)
    res["fisher_obs_ind (rows downloaded, np.mean on the host)"] = round(time.perf_counter() - t0, 4)
    res["identical"] = bool(host.tobytes() == out.tobytes())
print(json.dumps(res))
