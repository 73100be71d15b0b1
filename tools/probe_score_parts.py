#!/usr/bin/env python3
"""The coded scoring sweep at one size under every split of a block's tiles (WGS_SCORE_CODED_PARTS = 1..16 where usable) and under the
library's own choice: same n x K sums as the float32 sweep, and the kernel time of each.  python tools/probe_score_parts.py 1250944 1000 10"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import device  # noqa: E402

m, n, K = (int(x) for x in sys.argv[1:4])
group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
b = device.DeviceBeagle(m, n, group_of, K)
b.synth(20260313, 2.0)
os.environ["WGSASSIGN_CODES"] = "0"
em = device.EMBatch(b, np.arange(K, dtype=np.int32))
em.run(5, 1e-4)
afs = device.AFSet(m, K)
for k in range(K):
    afs.set_column_from_em(k, em, k)
out0, _ = device.assign(b, afs)
os.environ["WGSASSIGN_CODES"] = "1"
os.environ["WGSASSIGN_SCORE_CODES_ALWAYS"] = "1"
b.prepare_codes(False)
res = {}
for parts in [0] + list(range(1, 17)) + [0]:     # (the library's own choice first and last: the first calls after a build run slower)
    if parts:
        os.environ["WGS_SCORE_CODED_PARTS"] = str(parts)
    else:
        os.environ.pop("WGS_SCORE_CODED_PARTS", None)
    best = None
    for _ in range(3):
        out1, _ = device.assign(b, afs)
        best = device.assign.last_ms if best is None else min(best, device.assign.last_ms)
    key = str(parts) if parts else ("chosen_first" if "chosen_first" not in res else "chosen")
    res[key] = {"ms": round(best, 3), "identical": bool(out1.tobytes() == out0.tobytes())}
print(json.dumps({"m": m, "n": n, "K": K, "parts": res}))
