#!/usr/bin/env python3
"""Does a process that ends while the class codes' memory is still being allocated (helper thread, VRAM an earlier process used) wait
for that allocation?  One cold fit, then the interpreter exits; run under `time`, after a process that used the memory.
   python tools/probe_exit.py [snps inds pops]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wgsassign_amd import device  # noqa: E402

t_start = time.perf_counter()
m, n, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10_000_000, 1000, 10)
ctx = device.get_context()
group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
b = device.DeviceBeagle(m, n, group_of, K)
b.synth(20260313, 2.0)
ctx.sync()
em = device.EMBatch(b, np.arange(K, dtype=np.int32))
t0 = time.perf_counter()
it = em.run(200, 1e-4)
print("fit %.3f s, iterations %d, codes_state %d, since start %.3f s" % (time.perf_counter() - t0, int(max(it)), b.codes_state(), time.perf_counter() - t_start), flush=True)
