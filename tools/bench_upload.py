#!/usr/bin/env python3
"""PCIe-inclusive rates: host matrix -> device slabs, and the thin mirror."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth
from wgsassign_amd import device, emMAF_cy
m, n, K = 1_000_000, 200, 5
rng = np.random.default_rng(1)
L = rng.random((m, 2 * n), dtype=np.float32) * np.float32(0.5)
group_of = (np.arange(n) // (n // K)).astype(np.int32)
device.get_context()
for rep in range(2):
    t0 = time.perf_counter(); b = device.DeviceBeagle.from_host(L, group_of, K); t1 = time.perf_counter() - t0
    b.close()
print("upload+permute %.2f GB in %.3f s = %.1f GB/s" % (L.nbytes / 1e9, t1, L.nbytes / 1e9 / t1))
f = np.full(m, 0.25, dtype=np.float32)
Lp = np.ascontiguousarray(L[:, :80])
t0 = time.perf_counter(); emMAF_cy.emMAF_update(Lp, f, 1); t1 = time.perf_counter() - t0
print("thin mirror emMAF_update (1M x 40 ind, %.2f GB): %.3f s = %.2f M SNP-updates/s" % (Lp.nbytes / 1e9, t1, m / t1 / 1e6))
