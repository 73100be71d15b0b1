#!/usr/bin/env python3
"""What does a cold coded fit wait for when the codes' hipMalloc is slow (VRAM used by an earlier process)?  Run after such a process,
   plain or under `rocprofv3 --hip-trace`: a warm direct fit first (as bench.py's headline leg), then the cold fit with the codes allowed.
   python tools/probe_stall.py [snps inds pops]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import device  # noqa: E402


def main():
    m, n, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10_000_000, 1000, 10)
    ctx = device.get_context()
    group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
    b = device.DeviceBeagle(m, n, group_of, K)
    b.synth(20260313, 2.0)
    ctx.sync()
    res = {}
    score = len(sys.argv) > 4 and sys.argv[4] == "score"
    for leg, env in (("direct", "0"), ("cold", "1"), ("warm", "1")):
        os.environ["WGSASSIGN_CODES"] = env
        em = device.EMBatch(b, np.arange(K, dtype=np.int32))
        ctx.sync()
        t0 = time.perf_counter()
        if not score or leg == "direct":
            it = em.run(200, 1e-4)
            res[leg] = {"seconds": round(time.perf_counter() - t0, 4), "iterations": int(max(it)), "sweep_ms_asked_at_once": round(em.fit_stats()[3], 2),
                        "codes_state": b.codes_state()}
            if leg == "cold":
                b.codes_info()                                   # (waits for the memory)
            res[leg]["sweep_ms"] = round(em.fit_stats()[3], 2)
            if leg == "direct":
                afs = device.AFSet(b.m, K, ctx=b.ctx)
                for k in range(K):
                    em.clamp(k, n // K)
                    afs.set_column_from_em(k, em, k)
        if score:
            t0 = time.perf_counter()
            out, _ = device.assign(b, afs)
            res[leg + "_pop_like"] = {"seconds": round(time.perf_counter() - t0, 4), "kernel_ms_asked_at_once": round(device.assign.last_ms, 2), "codes_state": b.codes_state(),
                                      "checksum": float(out.sum())}
            if leg == "cold":
                b.codes_info()
            res[leg + "_pop_like"]["kernel_ms"] = round(device.last_assign_ms(b.ctx), 2)
        em.close()
    info = b.codes_info()
    res["alloc_ms"], res["alloc_wait_ms"] = info["alloc_ms"], info["alloc_wait_ms"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
