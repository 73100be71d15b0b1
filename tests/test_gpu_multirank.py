"""Two ranks (gloo) sharing the one GPU of the test box, SNPs sharded in two contiguous ranges:
the real kernels + the real driver loop must reproduce the single-process results -- allele
frequencies and iteration counts bit for bit (incl. the rank-to-rank carry of the exact
convergence chain, forced on every iteration in one variant), assignment and leave-one-out
log-likelihoods (global partition labels) within tolerance."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
rank = int(sys.argv[1]); guard = float(sys.argv[2])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=2)
from wgsassign_amd import device, emMAF, glassy
from wgsassign_amd.comm import TorchComm, shard_range
device.EMBatch.GUARD = guard
comm = TorchComm()
G = os.path.join({root!r}, "tests", "golden")
fit, loo, asg = (np.load(os.path.join(G, f)) for f in ("amre_fit.npz", "amre_loo.npz", "amre_assign.npz"))
L, IDs = fit["L"], fit["IDs"]
m = L.shape[0]
lo, hi = shard_range(m, comm.rank, comm.world)
pops = np.unique(IDs[:, 1])
group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
ctx = device.Context(0)
ok = True
def close(a, b, rtol):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return bool(np.all(np.abs(a - b) <= rtol * np.abs(b)))
# --get_reference_af on the shard
b = device.DeviceBeagle.from_host(np.ascontiguousarray(L[lo:hi]), group_of, len(pops), site0=lo, ctx=ctx)
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    _, af, iters = emMAF.emMAF_populations(None, IDs, 200, 1e-4, beagle=b, comm=comm)
ok &= list(iters) == list(fit["iters"]) and af.tobytes() == np.ascontiguousarray(fit["pop_af"][lo:hi]).tobytes()
# --loo with 3 partitions (labels use GLOBAL site indices)
with contextlib.redirect_stdout(io.StringIO()):
    ll, parts = glassy.loo_device(b, b, af, group_of, 200, 1e-4, 3, comm=comm, verbose=False)
ok &= close(ll, loo["loo_P3"], 1e-6) and close(parts, loo["parts_P3"], 2e-5)
ok &= af.tobytes() == np.ascontiguousarray(loo["af_after_P3"][lo:hi]).tobytes()
# --get_pop_like on the other file
La = asg["L"]
ba = device.DeviceBeagle.from_host(np.ascontiguousarray(La[lo:hi]), None, 1, site0=lo, ctx=ctx)
afs = device.AFSet.from_host(np.ascontiguousarray(fit["pop_af"][lo:hi]), ctx=ctx)
out, _ = device.assign(ba, afs, comm=comm)
ok &= close(out.astype(np.float32), asg["logl"], 1e-6)
print("RANK", rank, "OK" if ok else "FAIL", list(iters), flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("guard", [0.25, 1e9])
def test_two_ranks_one_gpu(tmp_path, guard):
    port = 29600 + (os.getpid() % 1000) + (1 if guard > 1 else 0)
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(guard)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "RANK %d OK" % r in o
