#!/usr/bin/env python3
"""The class-coded scoring sweep against the float32 one on a device-generated matrix, three times over: how many of the n x K sums
differ, where, by how much, and whether the repetitions agree (a race shows as repetitions that differ).  The launch shape changes
with the size (the parts a block's tiles are split over): python tools/check_coded_scoring.py 10000000 1000 10"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from wgsassign_amd import device
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
outs = []
for rep in range(3):
    out1, _ = device.assign(b, afs)
    outs.append(out1.copy())
    d = out1 != out0
    print(m, "rep", rep, "state", b.codes_state(), "differ", int(d.sum()), "of", d.size, "rows", np.flatnonzero(d.any(axis=1))[:12].tolist(), "cols", np.flatnonzero(d.any(axis=0)).tolist(),
          "max rel", float(np.max(np.abs(out1 - out0) / np.abs(out0))))
print("reps equal", all(o.tobytes() == outs[0].tobytes() for o in outs))
