import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, synth
from wgsassign_amd import device
from wgsassign_amd._lib import MODE_FAST, MODE_EXACT
g = np.load("/root/repo/tests/golden/synth_mid.npz")
L, IDs = synth.make_beagle(int(g["m"]), int(g["n"]), int(g["K"]))
pops = np.unique(IDs[:, 1]); group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
b = device.DeviceBeagle.from_host(L, group_of, len(pops))
for mode in (MODE_EXACT, MODE_FAST):
    em = device.EMBatch(b, np.arange(5, dtype=np.int32), mode=mode)
    iters = em.run(200, 1e-4)
    errs = []
    for k in range(5):
        em.clamp(k, 20)
        f = em.get_f(k); ref = g["pop_af"][:, k]
        errs.append(np.max(np.abs(f.astype(np.float64) - ref) / ref))
    print("mode", mode, "iters", list(iters), "ref", list(g["iters"]), "max rel err per pop", ["%.2e" % e for e in errs])
    afs = device.AFSet.from_host(np.ascontiguousarray(g["pop_af"][:5000]))
bs = device.DeviceBeagle.from_host(np.ascontiguousarray(L[:5000]))
for mode in (MODE_EXACT, MODE_FAST):
    out, _ = device.assign(bs, afs, mode=mode)
    ref = g["logl_5000"].astype(np.float64)
    print("assign mode", mode, "max rel err", np.max(np.abs(out.astype(np.float32) - ref) / np.abs(ref)))
