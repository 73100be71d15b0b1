#!/usr/bin/env python3
"""How many of the n x K assignment log-likelihoods (float32, as stored) are bit-identical to the oracle's
np.sum(dtype=float) of the per-site vectors: python tools/count_identical.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from oracle import oracle  # noqa: E402
from wgsassign_amd import device  # noqa: E402

for m, n, K in ((1501, 40, 6), (20_000, 60, 5), (70_001, 30, 10), (300_000, 24, 8)):
    labels = np.arange(n) % K
    L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=m)
    pops, af, _, _ = oracle.fit_reference_af(L, IDs, t=8)
    want = oracle.assignLL(L, af.copy(), 8)
    b = device.DeviceBeagle.from_host(L)                      # one slab: the --get_pop_like shape
    afs = device.AFSet.from_host(af)
    out, _ = device.assign(b, afs)
    got = out.astype(np.float32)
    g = device.DeviceBeagle.from_host(L, np.searchsorted(pops, IDs[:, 1]).astype(np.int32), K)
    out2, _ = device.assign(g, afs)
    print("m=%d n=%d K=%d: identical float32 entries %d of %d (one slab), %d (population slabs); float64 sums equal to the last bit: %s"
          % (m, n, K, int(np.sum(got.view(np.uint32) == want.view(np.uint32))), n * K,
             int(np.sum(out2.astype(np.float32).view(np.uint32) == want.view(np.uint32))), "n/a"))
    afs.close(); b.close(); g.close()
