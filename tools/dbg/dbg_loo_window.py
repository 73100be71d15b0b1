import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, synth, io, contextlib
from oracle import oracle as orc
from wgsassign_amd import device as dev, glassy, emMAF

def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)

n, K, P, m = 500, 8, 3, 512
group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
IDs = np.array([["Ind%d" % i, "pop%02d" % group_of[i]] for i in range(n)], dtype=str)
L, _ = synth.make_beagle(m, n, K, seed=1)
pops, af, _, _ = orc.fit_reference_af(L, IDs, t=8)
af0 = af.copy()
# 1. full pipeline vs oracle
a1, a2 = af.copy(), af.copy()
lo, po = orc.loo(L, a1, IDs, 8, 200, 1e-4, None, P)
ld, pd_ = quiet(glassy.loo, L, a2, IDs, 1, 200, 1e-4, None, P)
d = np.abs(ld.astype(np.float64) - lo) / np.abs(lo)
print("pipeline: max rel", d.max(), "parts same", pd_.tobytes() == po.tobytes(), "af same", a1.tobytes() == a2.tobytes())
# 2. columns given
cols = np.empty((n, m), np.float32)
for i in range(n):
    members = np.flatnonzero(group_of == group_of[i]); others = members[members != i]
    f, _ = orc.emMAF(orc.gather(L, others, 8), 200, 1e-4, 8)
    cols[i] = orc.clamp(f, len(others))
l2, p2 = orc.loo_score(L, af0.copy(), cols, group_of, 8, P)
print("oracle loo == loo_score", np.array_equal(lo, l2))
bw = dev.DeviceBeagle.from_host(L, group_of, K)
emw = dev.EMBatch(bw, group_of, np.arange(n, dtype=np.int32))
for i in range(n):
    emw.set_f(i, cols[i])
got = np.stack([emw.get_f(i) for i in range(n)])
print("set_f/get_f round trip", np.array_equal(got, cols))
afw = dev.AFSet.from_host(af0.copy())
o, pr = glassy.score_loo_batch(bw, afw, emw, group_of, 0, n, P)
d = np.abs(o - l2.astype(np.float64)) / np.abs(l2)
print("given columns: max rel", d.max(), "rows with diff", np.flatnonzero(d.max(axis=1) > 1e-6)[:20], "parts same", pr.tobytes() == p2.tobytes())
bad = np.argwhere(d > 1e-6)
print("bad cells (first 20):", bad[:20].tolist())
print("o[0]", o[0], "\nl2[0]", l2[0])
# shared-column scoring of the same matrix
o3, _ = dev.assign(bw, afw)
l3 = orc.assignLL(L, af0.copy(), 8)
print("shared columns: max rel", (np.abs(o3 - l3) / np.abs(l3)).max())
for x in (afw, emw, bw):
    x.close()
print("closed")
