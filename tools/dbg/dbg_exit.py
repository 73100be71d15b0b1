import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, synth
from wgsassign_amd import device as dev, glassy
which = sys.argv[1]
L, IDs = synth.make_beagle(2000, 12, 3, seed=1)
pops = np.unique(IDs[:, 1]); group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
bw = dev.DeviceBeagle.from_host(L, group_of, 3)
if which in ("em", "all"):
    emw = dev.EMBatch(bw, group_of, np.arange(12, dtype=np.int32))
    emw.run(50, 1e-4)
if which in ("af", "all"):
    afw = dev.AFSet.from_host(np.full((2000, 3), 0.3, dtype=np.float32))
if which in ("score", "all"):
    afw2 = dev.AFSet.from_host(np.full((2000, 3), 0.3, dtype=np.float32))
    sc = dev.Score(bw, afw2)
    sc.sums()
print("leaving with", which, flush=True)
