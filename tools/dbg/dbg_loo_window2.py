import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, synth, io, contextlib
from oracle import oracle as orc
from wgsassign_amd import device as dev, glassy, emMAF

def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)

m, n, K, P = 200_000, 500, 8, 3
group_of = np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)
counts = np.bincount(group_of, minlength=K)
IDs = np.array([["Ind%d" % i, "pop%02d" % group_of[i]] for i in range(n)], dtype=str)
b = dev.DeviceBeagle(m, n, group_of, K)
b.synth(synth.SEED + 4, 2.0)
pops, af, iters_full = quiet(emMAF.emMAF_populations, None, IDs, 200, 1e-4, beagle=b)
af0 = af.copy()
r0, nr = 123_456, 512
cols = np.empty((n, nr), dtype=np.float32)
def grab(em, i0, i1):
    for i in range(i0, i1):
        cols[i] = em.get_f_range(i - i0, r0, nr)
timings = {}
ll, parts = quiet(glassy.loo_device, b, b, af, group_of, 200, 1e-4, P, timings=timings, inspect=grab)
rows = b.download_rows(r0, nr)
A0 = np.ascontiguousarray(af0[r0:r0 + nr])
print("af0 window stats", A0.min(), A0.max(), "af changed by loo:", not np.array_equal(af, af0))
# shared-column check on the window
bw = dev.DeviceBeagle.from_host(rows, group_of, K, site0=r0)
afw = dev.AFSet.from_host(A0)
o3, _ = dev.assign(bw, afw)
l3 = orc.assignLL(rows, A0.copy(), 8)
print("shared columns on window: max rel", (np.abs(o3 - l3) / np.abs(l3)).max())
print("afw round trip", np.array_equal(afw.to_host(), A0))
ll_o, parts_o = orc.loo_score(rows, A0.copy(), cols, group_of, 8, P, site0=r0)
emw = dev.EMBatch(bw, group_of, np.arange(n, dtype=np.int32))
for i in range(n):
    emw.set_f(i, cols[i])
o, pr = glassy.score_loo_batch(bw, afw, emw, group_of, 0, n, P)
d = np.abs(o - ll_o.astype(np.float64)) / np.abs(ll_o)
print("given columns: max rel", d.max(), "parts same", pr.tobytes() == parts_o.tobytes())
print("bad cells:", np.argwhere(d > 1e-6)[:12].tolist())
print("o[0]", o[0]); print("l[0]", ll_o[0])
# what would individual 0 get with the LAST columns (af after loo)?
A1 = np.ascontiguousarray(af[r0:r0 + nr])
l_last = orc.assignLL(rows[:, :2].copy(), A1.copy(), 8)
l_init = orc.assignLL(rows[:, :2].copy(), A0.copy(), 8)
print("ind0 with initial columns", l_init[0]); print("ind0 with final columns  ", l_last[0])
for x in (afw, emw, bw, b):
    x.close()
