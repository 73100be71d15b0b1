#!/usr/bin/env python3
"""The class encoder alone (csrc/codes_kernels.hip) on a device-generated matrix: build time, geometry, statistics.
   python tools/bench_encode.py [snps inds pops [repeats]]      (WGSASSIGN_CODES_TABLE=64|128|256 forces a geometry)"""
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
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    ctx = device.get_context()
    group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
    b = device.DeviceBeagle(m, n, group_of, K)
    out = []
    for r in range(reps):
        b.synth(SEED + r, 2.0)          # (a new matrix drops the codes)
        ctx.sync()
        t0 = time.perf_counter()
        info = b.codes_info()
        info["seconds"] = round(time.perf_counter() - t0, 4)
        if os.environ.get("BENCH_ENCODE_FIT"):                   # an EM fit through the codes, and what codes_info says afterwards
            em = device.EMBatch(b, np.arange(K, dtype=np.int32))
            it = em.run(200, 1e-4)
            info["fit_iterations"] = [int(x) for x in it]
            info["fit_sweep_ms"] = round(em.fit_stats()[3], 2)
            info["em_direct_tile_share_after_fit"] = b.codes_info()["em_direct_tile_share"]
            em.close()
        out.append(info)
    print(json.dumps({"config": "%d x %d, K=%d" % (m, n, K), "gl_bytes": b.nbytes(), "runs": out}))


if __name__ == "__main__":
    main()
