"""Two ranks (gloo) sharing the one GPU of the test box, SNPs sharded in two contiguous ranges:
the real kernels + the real driver loop must reproduce the single-process results -- allele
frequencies and iteration counts bit for bit (incl. the rank-to-rank carry of the exact
convergence chain, forced on every iteration in one variant), assignment and leave-one-out
log-likelihoods (global partition labels) within tolerance."""
import os
import subprocess
import sys

import pytest



def free_port():
    """A TCP port p the OS reports free right now, with p + 1 free too (the communicators' TCP star listens there)."""
    from wgsassign_amd.comm import free_port_pair
    return free_port_pair()


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
ok &= close(ll, loo["loo_P3"], 1e-6) and parts.tobytes() == loo["parts_P3"].tobytes()   # float32 carries cross the ranks
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
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(guard)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "RANK %d OK" % r in o


def clean(stdout):
    """Program output without the launcher's noise: gloo prints a banner per rank and torchrun may
    emit a blank line before the first program line."""
    lines = [l for l in stdout.splitlines() if "[Gloo]" not in l and "connected peer ranks" not in l]   # ranks interleave
    # the two ranks' gloo banners can interleave mid-line and leave a fragment ("1") of their own: the program's
    # output starts at its banner line
    if "WGSassign" in lines:
        lines = lines[lines.index("WGSassign"):]
    while lines and lines[0] == "":
        lines.pop(0)
    return lines


def test_cli_two_ranks_matches_single_process(tmp_path, golden):
    """`torchrun --nproc-per-node 2 -m wgsassign_amd.WGSassign ...` (gloo, both ranks on the one GPU):
    same stdout and output files as the reference CLI run recorded in tests/golden/amre_cli.npz."""
    import gzip
    import numpy as np
    from conftest import GOLDEN
    g = golden("amre_cli.npz")
    data = os.path.join(GOLDEN, "data")
    env = dict(os.environ, WGSASSIGN_BACKEND="gloo", WGSASSIGN_DEVICE="0", PYTHONPATH=ROOT)   # both ranks on GPU 0
    port = free_port()
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), "-m", "wgsassign_amd.WGSassign"]
    r = subprocess.run(base + ["--beagle", os.path.join(data, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"),
                               "--pop_af_IDs", os.path.join(data, "amre.breeding.ind85.reference_k5.IDs.txt"),
                               "--get_reference_af", "--loo", "--partition_sites", "3", "--out", "ref", "--threads", "2"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert np.load(tmp_path / "ref.pop_af.npy").tobytes() == g["pop_af_npy"].tobytes()
    assert (tmp_path / "ref.pop_names.txt").read_text() == str(g["pop_names"])
    ref_lines = str(g["stdout_ref"]).replace("<TMP>/", "").splitlines()
    got_lines = clean(r.stdout)
    assert got_lines == ref_lines, "\n".join(got_lines[:12])

    def table(text):
        rows = [line.split("\t") for line in text.strip().split("\n")]
        return rows[0], rows[1:]
    h_ref, r_ref = table(str(g["loo_tsv"]))
    h_got, r_got = table((tmp_path / "ref.pop_like_LOO.tsv").read_text())
    assert h_got == h_ref and [x[:2] for x in r_got] == [x[:2] for x in r_ref]
    a = np.array([[float(v) for v in x[2:]] for x in r_got])
    b = np.array([[float(v) for v in x[2:]] for x in r_ref])
    assert np.all(np.abs(a - b) <= 1e-6 * np.abs(b) + 1.5e-6)
    h_ref, r_ref = table(str(g["parts_tsv"]))
    h_got, r_got = table(gzip.open(tmp_path / "ref.pop_like_LOO_partitions_3.tsv.gz", "rt").read())
    assert h_got == h_ref
    assert r_got == r_ref                       # partition sums are bit-exact -> identical text
    # --loo --loo_downsampled_beagle, sharded: names-only passes + keep masks
    gd = golden("amre_cli_downsampled.npz")
    r = subprocess.run(base + ["--beagle", os.path.join(data, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"),
                               "--pop_af_IDs", os.path.join(data, "amre.breeding.ind85.reference_k5.IDs.txt"),
                               "--get_reference_af", "--loo", "--loo_downsampled_beagle",
                               os.path.join(data, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each_subset_80percent_sites.beagle.gz"),
                               "--out", "ds", "--threads", "2"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert np.load(tmp_path / "ds.pop_af.npy").tobytes() == gd["pop_af_npy"].tobytes()
    assert clean(r.stdout) == str(gd["stdout"]).replace("<TMP>/", "").splitlines()
    h_ref, r_ref = table(str(gd["loo_tsv"]))
    h_got, r_got = table((tmp_path / "ds.pop_like_LOO_downsampled.tsv").read_text())
    assert h_got == h_ref and [x[:2] for x in r_got] == [x[:2] for x in r_ref]
    a = np.array([[float(v) for v in x[2:]] for x in r_got])
    b = np.array([[float(v) for v in x[2:]] for x in r_ref])
    assert np.all(np.abs(a - b) <= 1e-6 * np.abs(b) + 1.5e-6)
    # --ne_obs, sharded: per-SNP matrices bit-identical, per-individual means within 1e-6
    gf = golden("fisher.npz")
    r = subprocess.run(base + ["--beagle", os.path.join(data, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"),
                               "--pop_af_IDs", os.path.join(data, "amre.breeding.ind85.reference_k5.IDs.txt"),
                               "--get_reference_af", "--ne_obs", "--out", "ne", "--threads", "2"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert np.load(tmp_path / "ne.fisher_obs.npy").tobytes() == gf["f_obs"].tobytes()
    assert np.load(tmp_path / "ne.ne_obs.npy").tobytes() == gf["ne_obs"].tobytes()
    assert (tmp_path / "ne.ne_obs.txt").read_text() == str(gf["ne_obs_txt"])
    assert (tmp_path / "ne.ne_ind.txt").read_text() == str(gf["ne_ind_txt"])      # np.mean's running total crosses the ranks
    assert clean(r.stdout) == str(gf["stdout"]).replace("<TMP>/", "").splitlines()
    # --get_pop_like, sharded
    r = subprocess.run(base + ["--beagle", os.path.join(data, "amre.nonbreeding.ind34.ds_2x.sites-filter.top_50_each.beagle.gz"),
                               "--pop_af_file", "ref.pop_af.npy", "--get_pop_like", "--out", "nb", "--threads", "2"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    got = np.loadtxt(tmp_path / "nb.pop_like.txt")
    ref = np.loadtxt(__import__("io").StringIO(str(g["pop_like_txt"])))
    assert np.all(np.abs(got - ref) <= 1e-6 * np.abs(ref))
    assert clean(r.stdout) == str(g["stdout_like"]).replace("<TMP>/", "").splitlines()


def test_cli_gpus_flag_starts_its_own_ranks(tmp_path, golden):
    """`WGSassign --gpus 2` (no torchrun): the command line starts one process per rank itself; here both on the one
    GPU over the TCP all-reduce.  Same files and stdout as the reference CLI run recorded in tests/golden."""
    import gzip
    import numpy as np
    from conftest import GOLDEN
    g = golden("amre_cli.npz")
    data = os.path.join(GOLDEN, "data")
    env = dict(os.environ, WGSASSIGN_COMM="socket", WGSASSIGN_DEVICE="0", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "wgsassign_amd.WGSassign", "--gpus", "2",
                        "--beagle", os.path.join(data, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"),
                        "--pop_af_IDs", os.path.join(data, "amre.breeding.ind85.reference_k5.IDs.txt"),
                        "--get_reference_af", "--loo", "--partition_sites", "3", "--out", "ref", "--threads", "2"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert np.load(tmp_path / "ref.pop_af.npy").tobytes() == g["pop_af_npy"].tobytes()
    assert clean(r.stdout) == str(g["stdout_ref"]).replace("<TMP>/", "").splitlines()
    assert gzip.open(tmp_path / "ref.pop_like_LOO_partitions_3.tsv.gz", "rt").read() == str(g["parts_tsv"])
    assert "\t-gpus 2\n" in (tmp_path / "ref.args").read_text()
    # a rank that fails takes the run down with its status (here: an unreadable Beagle file on every rank)
    r = subprocess.run([sys.executable, "-m", "wgsassign_amd.WGSassign", "--gpus", "2", "--beagle", str(tmp_path / "missing.beagle.gz"),
                        "--pop_af_IDs", os.path.join(data, "amre.breeding.ind85.reference_k5.IDs.txt"), "--get_reference_af", "--out", "bad"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


_SUMS_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import synth
from oracle import oracle
from wgsassign_amd import device, glassy
from wgsassign_amd.comm import SocketComm, shard_range
rank, world = int(sys.argv[1]), 3
comm = SocketComm(rank, world, "127.0.0.1", {port})
m, n, K = 100_003, 20, 4
labels = np.arange(n) % K
L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=11)
pops, af, _, _ = oracle.fit_reference_af(L, IDs, t=4)
group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
lo, hi = shard_range(m, rank, world)
assert (lo % 8192 == 0 or rank == 0) and (hi % 8192 == 0 or rank == world - 1)
ctx = device.Context(0)
comm.attach(ctx)          # the library's own loops (wgs_em_fit, wgs_loo) run across the ranks, all-reducing through this transport
assert comm.handle is not None
from wgsassign_amd import emMAF
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    bq = device.DeviceBeagle.from_host(np.ascontiguousarray(L[lo:hi]), group_of, K, site0=lo, ctx=ctx)
    _, af_c, it_c = emMAF.emMAF_populations(None, IDs, 200, 1e-4, beagle=bq, comm=comm)          # wgs_em_fit over 3 ranks
    os.environ["WGSASSIGN_EM_LOOP"] = "python"
    _, af_p, it_p = emMAF.emMAF_populations(None, IDs, 200, 1e-4, beagle=bq, comm=comm)          # the step-by-step twin
    del os.environ["WGSASSIGN_EM_LOOP"]
    bq.close()
assert af_c.tobytes() == af_p.tobytes() == np.ascontiguousarray(af[lo:hi]).tobytes() and list(it_c) == list(it_p)
b = device.DeviceBeagle.from_host(np.ascontiguousarray(L[lo:hi]), group_of, K, site0=lo, ctx=ctx)
afs = device.AFSet.from_host(np.ascontiguousarray(af[lo:hi]), ctx=ctx)
import ctypes
from wgsassign_amd import _lib
def collectives():
    st = (ctypes.c_int64 * 4)()
    _lib.check(_lib.load().wgs_comm_stats(comm.handle, st))
    return np.array(st[:2])                                   # all-reduces, broadcasts through the library's communicator
c0 = collectives()
out, _ = device.assign(b, afs, comm=comm)                      # --get_pop_like, SNP-sharded over 3 ranks
# what crosses the ranks for the n x K totals in NumPy's order: `world` broadcasts (the running float64 total handed
# from shard to shard on the stream), no all-reduce
assert list(collectives() - c0) == [0, world], collectives() - c0
import io, contextlib
a1 = np.ascontiguousarray(af[lo:hi]).copy()
c0, tm = collectives(), {{}}
with contextlib.redirect_stdout(io.StringIO()):
    ll, parts = glassy.loo_device(b, b, a1, group_of, 200, 1e-4, 2, comm=comm, verbose=False, timings=tm)      # one C call (wgs_loo)
# --loo, one batch: the batch-size agreement + one all-reduce of the convergence sums per EM iteration enqueued + the one that
# closes the fit (the ranks compare what they found); `world` broadcasts per batched resolution of undecided fits, for the
# totals, and for the partition chains
assert tm["one_call"] and tm["em_batches"] == 1
assert list(collectives() - c0) == [2 + tm["em_iterations_enqueued"], world * (2 + tm["em_chain_resolutions"])], (collectives() - c0, tm)
os.environ["WGSASSIGN_LOO"] = "python"
a2 = np.ascontiguousarray(af[lo:hi]).copy()
with contextlib.redirect_stdout(io.StringIO()):
    ll_py, parts_py = glassy.loo_device(b, b, a2, group_of, 200, 1e-4, 2, comm=comm, verbose=False) # its Python twin
ok = ll.tobytes() == ll_py.tobytes() and parts.tobytes() == parts_py.tobytes()
if rank == 0:                                                 # the same on ONE shard, and NumPy itself
    bw = device.DeviceBeagle.from_host(L, group_of, K, ctx=ctx)
    aw = device.AFSet.from_host(af, ctx=ctx)
    whole, _ = device.assign(bw, aw)
    ok &= out.tobytes() == whole.tobytes()                    # float64, bit for bit
    for i, k in ((0, 0), (n - 1, K - 1), (7, 2)):
        vec = np.zeros(m, dtype=np.float32)
        oracle.loglike(L, af, vec, 4, i, k)
        ok &= out[i, k] == np.sum(vec, dtype=float)
    a3 = af.copy()
    with contextlib.redirect_stdout(io.StringIO()):
        ll_w, parts_w = glassy.loo_device(bw, bw, a3, group_of, 200, 1e-4, 2, verbose=False)
    ok &= ll.tobytes() == ll_w.tobytes() and parts.tobytes() == parts_w.tobytes()
    with np.errstate(all="ignore"):
        loo_o, parts_o = oracle.loo(L, af.copy(), IDs, 4, 200, 1e-4, None, 2)
    ok &= ll.tobytes() == loo_o.tobytes() and parts.tobytes() == parts_o.tobytes()
# a matrix on which the ORDER of the float64 additions shows (2M sites, per-site values from 2^-11 to 2^4): the running total
# handed from shard to shard must follow NumPy's chunk order; adding the shard totals would not give the same bits
m2, n2 = 2_000_003, 4
L2, _ = synth.make_beagle_for_labels(m2, np.arange(n2) % 2, 2, seed=12)
af2 = np.random.default_rng(3).choice(np.array([3e-4, 2e-3, 0.05, 0.4, 0.9, 1 - 3e-4], dtype=np.float32), size=(m2, 2))
lo2, hi2 = shard_range(m2, rank, world)
b2 = device.DeviceBeagle.from_host(np.ascontiguousarray(L2[lo2:hi2]), None, 1, site0=lo2, ctx=ctx)
afs2 = device.AFSet.from_host(np.ascontiguousarray(af2[lo2:hi2]), ctx=ctx)
out2, _ = device.assign(b2, afs2, comm=comm)
if rank == 0:
    cuts = [shard_range(m2, r, world)[0] for r in range(world)] + [m2]
    shard_totals_differ = 0
    for i in range(n2):
        for k in range(2):
            vec = np.zeros(m2, dtype=np.float32)
            oracle.loglike(L2, af2, vec, 4, i, k)
            ok &= out2[i, k] == np.sum(vec, dtype=float)
            v64 = vec.astype(np.float64)
            tot = 0.0
            for r in range(world):          # what summing per-shard totals would give
                part = 0.0
                for c in np.add.reduceat(v64[cuts[r]:cuts[r + 1]], np.arange(0, cuts[r + 1] - cuts[r], 8192)):
                    part = part + c
                tot = tot + part
            shard_totals_differ += tot != out2[i, k]
    ok &= shard_totals_differ > 0          # i.e. this check notices a different order
from wgsassign_amd import fisher
ne = fisher.fisher_obs_ind(None, np.ascontiguousarray(af[lo:hi]), IDs, 1, beagle=b, comm=comm, m_total=m)   # --ne_obs, sharded
if rank == 0:
    ok &= ne.tobytes() == oracle.fisher_obs_ind(L, af.copy(), IDs, 4).tobytes()           # np.mean itself
print("RANK", rank, "OK" if ok else "FAIL", flush=True)
comm.barrier()
sys.exit(0 if ok else 1)
'''


def test_three_ranks_sums_are_the_single_gpu_sums_bit_for_bit(tmp_path):
    """The n x K float64 sums of a matrix sharded over three ranks (shards cut at multiples of 8192 sites, NumPy's running
    total handed from shard to shard) equal the one-shard sums and np.sum(vec, dtype=float) itself to the last bit --
    for --get_pop_like and for --loo (C call and Python twin), whose float32 results then equal the oracle's exactly."""
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(_SUMS_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(3)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "RANK %d OK" % r in o


def test_the_tag_kernels_of_an_rccl_collective_on_made_up_ranks():
    """The device code an N > 1 RCCL job runs around every collective -- the row each rank writes behind the payload, and, behind the
    all-reduce, the comparison of ALL ranks' rows with one's own (csrc/rccl_comm.hip) -- on rows made up here for 2, 8 and 64 ranks
    (RCCL refuses two ranks on one GPU, and no multi-GPU box has been available): rows that agree pass for every rank; a difference in
    any compared word of any rank is reported by every other rank, with both rows; the free word is not compared."""
    import ctypes
    import numpy as np
    from wgsassign_amd import _lib, device
    lib = _lib.load()
    ctx = device.get_context()

    def check(rows, as_rank):
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        out = np.zeros(18, dtype=np.float64)
        _lib.check(lib.wgs_debug_comm_tag_kernels(ctx.handle, rows.shape[0], rows.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), as_rank,
                                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    base = np.array([41.0, 1.0, 7.0, 3.0, 10.0, 2.0, 20.0, 0.0])        # sequence number, step, generation, iteration, shapes, size, free word
    for world in (1, 2, 8, 64):
        rows = np.tile(base, (world, 1))
        rows[:, 7] = np.arange(world)                                    # the free word differs on purpose (the fuse agreement travels in it)
        for r in {0, world // 2, world - 1}:
            assert check(rows, r)[0] == 0.0, (world, r)
        if world == 1:
            continue
        for word in range(7):                                            # every compared word
            bad = world - 1 if word % 2 else world // 3
            other = rows.copy()
            other[bad, word] += 1.0
            for r in {0, world // 2, world - 1} - {bad}:
                out = check(other, r)
                assert out[0] == 1.0 and int(out[1]) == bad, (world, word, r, out)
                assert np.array_equal(out[2:10], other[r]) and np.array_equal(out[10:18], other[bad])
            out = check(other, bad)                                      # the odd one out sees somebody else's row differ
            assert out[0] == 1.0 and int(out[1]) != bad and np.array_equal(out[2:10], other[bad])


def test_native_rccl_single_rank():
    """The library's own RCCL communicator (dlopen'ed librccl, no torch): with one rank the all-reduce
    is the identity -- exercises loading, unique id, init, the stream-ordered collective, destroy."""
    import numpy as np
    from wgsassign_amd import comm as wcomm
    from wgsassign_amd import device
    c = wcomm.RcclComm(device.get_context(), 0, 1)
    x = np.array([1.5, -2.25, 1e300, 0.0])
    assert np.array_equal(c.allreduce_sum(x), x)
    assert c.allgather_object({"a": 1}) == [{"a": 1}]
    big = np.arange(200_000, dtype=np.float64)
    assert np.array_equal(c.allreduce_sum(big), big)
    # what RCCL itself says about the communicator it built (wgs_comm_init refuses one that differs from what was asked for)
    info = c.info()
    assert info["native"] and info["rccl_ranks_seen"] == 1 and info["rccl_rank"] == 0 and info["world"] == 1
    assert info["rccl_device"] == device.get_context().device
    t = c.time_collectives(reps=10, n=16)
    assert t["allreduce_us"] > 0 and t["bcast_us"] >= 0 and t["calls"] == 10
    c.barrier()
    c.close()


def test_cli_three_ranks_on_a_bgzf_file_equal_one_process(tmp_path, oracle):
    """A generated BGZF Beagle file (what ANGSD writes) through the command line: three ranks over the TCP-star
    communicator (no torch; all on the one GPU) each open the file at the access point of their own SNP range and
    produce, bit for bit, the files of the single-process run -- which in turn match the oracle on the parsed matrix."""
    import gzip
    import numpy as np
    import synth
    from test_reader_cpu import _bgzf_write
    m, K = 30_011, 6
    rng = np.random.default_rng(12)
    labels = np.repeat(np.arange(K), [5, 6, 7, 4, 8, 6])
    rng.shuffle(labels)
    L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=404)
    n = len(IDs)
    head = "marker\tallele1\tallele2\t" + "\t".join("%s\t%s\t%s" % (x, x, x) for x in IDs[:, 0])
    lines = [head]
    for s in range(m):
        vals = []
        for i in range(n):
            g0, g1 = L[s, 2 * i], L[s, 2 * i + 1]
            vals += ["%.6f" % g0, "%.6f" % g1, "%.6f" % max(0.0, 1 - g0 - g1)]
        lines.append("chr%d_%d\t0\t1\t" % (s % 3, s + 1) + "\t".join(vals))
    beagle = str(tmp_path / "gen.beagle.gz")
    _bgzf_write(beagle, "\n".join(lines) + "\n")
    ids = str(tmp_path / "ids.txt")
    np.savetxt(ids, IDs, fmt="%s", delimiter="\t")
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env0.update(PYTHONPATH=ROOT, WGSASSIGN_DEVICE="0", WGSASSIGN_INDEX_DIR=str(tmp_path))
    args = ["-m", "wgsassign_amd.WGSassign", "--beagle", beagle, "--pop_af_IDs", ids, "--get_reference_af", "--loo",
            "--partition_sites", "2"]
    one = subprocess.run([sys.executable] + args + ["--out", "one"], cwd=tmp_path, env=env0, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-3000:]
    port = free_port()
    procs = []
    for r in range(3):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   WGSASSIGN_COMM="socket")
        procs.append(subprocess.Popen([sys.executable] + args + ["--out", "three"], cwd=tmp_path, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
    assert outs[0].replace("three.", "one.") == one.stdout and outs[1].strip() == "" and outs[2].strip() == ""
    for suffix in (".pop_af.npy", ".pop_names.txt", ".pop_like_LOO.tsv"):
        assert (tmp_path / ("three" + suffix)).read_bytes() == (tmp_path / ("one" + suffix)).read_bytes(), suffix
    assert gzip.open(tmp_path / "three.pop_like_LOO_partitions_2.tsv.gz", "rt").read() == \
        gzip.open(tmp_path / "one.pop_like_LOO_partitions_2.tsv.gz", "rt").read()
    # ... and the single-process files are the oracle's
    pops, af_o, _, iters_o = oracle.fit_reference_af(L, IDs, t=8)
    assert np.load(tmp_path / "one.pop_af.npy").tobytes() == af_o.tobytes()
    assert [l for l in one.stdout.splitlines() if l.startswith("EM (MAF)")][:K] == ["EM (MAF) converged at iteration: %d" % i for i in iters_o]
    loo_o, parts_o = oracle.loo(L, af_o.copy(), IDs, 8, 200, 1e-4, None, 2)
    rows = [l.split("\t") for l in gzip.open(tmp_path / "one.pop_like_LOO_partitions_2.tsv.gz", "rt").read().strip().split("\n")][1:]
    assert [["%.6f" % v for v in row] for row in parts_o] == [r[3:] for r in rows]


_MIXED_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
rank, world, variant = int(sys.argv[1]), 2, sys.argv[2]
if variant == "one_rank_without_codes" and rank == 1:
    os.environ["WGSASSIGN_CODES"] = "0"                       # this rank sweeps the float32 slabs throughout
late = variant in ("memory_late_on_one_rank", "memory_late_and_no_agreement") and rank == 1
if late:
    os.environ["WGSASSIGN_CODES_ALLOC_WAIT_MS"] = "0"          # ... this one until its codes' memory arrives, in the middle of the fit
import synth
from oracle import oracle
from wgsassign_amd import device, emMAF
from wgsassign_amd.comm import SocketComm, shard_range, CollectiveMismatch, COMM_DIVERGED
import io, contextlib
if late:
    device.debug_hook("codes_alloc_release_after_sweeps", 4)   # handed over after the fourth sweep over the float32 slabs: a count, not a race
if variant == "memory_late_and_no_agreement":
    device.debug_hook("em_fuse_without_agreement", 1)          # the defect of commit 807a461: each rank runs two iterations per sweep as soon as IT can
comm = SocketComm(rank, world, "127.0.0.1", {port})
m, n, K = 150_000, 96, 3
L, IDs = synth.make_beagle(m, n, K, seed=21)
with contextlib.redirect_stdout(io.StringIO()):
    pops, af, _, iters = oracle.fit_reference_af(L, IDs, t=4)
group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
lo, hi = shard_range(m, rank, world)
ctx = device.Context(0)
comm.attach(ctx)
b = device.DeviceBeagle.from_host(np.ascontiguousarray(L[lo:hi]), group_of, K, site0=lo, ctx=ctx)
ok = True
fused = []
try:
    for fit in range(3):          # cold (codes built inside where allowed), then with whatever each rank has by then
        with contextlib.redirect_stdout(io.StringIO()):
            _, af_c, it_c = emMAF.emMAF_populations(None, IDs, 200, 1e-4, beagle=b, comm=comm)
        ok &= af_c.tobytes() == np.ascontiguousarray(af[lo:hi]).tobytes() and list(it_c) == [int(x) for x in iters]
        fused.append(emMAF.emMAF_populations.last_stats[0])         # sweeps enqueued: fewer than iterations once two run per sweep
except CollectiveMismatch as e:
    print("RANK", rank, "MISMATCH:", e, flush=True)
    os._exit(COMM_DIVERGED)
print("RANK", rank, "sweeps per fit", fused, "iterations", [int(x) for x in iters], flush=True)
# leave-one-out on a smaller matrix: the re-fits of one rank through the slab's class table, of the other over the float32 slabs
from wgsassign_amd import glassy
m2 = 12_000
L2, IDs2 = synth.make_beagle(m2, n, K, seed=22)
with contextlib.redirect_stdout(io.StringIO()):
    _, af2, _, _ = oracle.fit_reference_af(L2, IDs2, t=4)
    with np.errstate(all="ignore"):
        loo_o, parts_o = oracle.loo(L2, af2.copy(), IDs2, 4, 200, 1e-4, None, 2)
lo2, hi2 = shard_range(m2, rank, world)
b2 = device.DeviceBeagle.from_host(np.ascontiguousarray(L2[lo2:hi2]), group_of, K, site0=lo2, ctx=ctx)
a2 = np.ascontiguousarray(af2[lo2:hi2]).copy()
with contextlib.redirect_stdout(io.StringIO()):
    ll, parts = glassy.loo_device(b2, b2, a2, group_of, 200, 1e-4, 2, comm=comm, verbose=False)
ok &= ll.tobytes() == loo_o.tobytes() and parts.tobytes() == parts_o.tobytes()
print("RANK", rank, "OK" if ok else "FAIL", "codes_state", b.codes_state(), b2.codes_state(), flush=True)
comm.close()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("variant", ["both_with_codes", "one_rank_without_codes", "memory_late_on_one_rank"])
def test_ranks_that_disagree_about_the_codes(tmp_path, variant):
    """SNP shards on two ranks whose sweeps go different ways -- one through the class codes (two iterations per sweep when it may),
    the other over the float32 slabs, or through the codes only from the middle of the fit -- must still run the same number of
    iterations per sweep: the sums of a sweep are all-reduced.  Every sweep's all-reduce carries each rank's "I could run two"
    (the free word of its tag row); once all ranks have said so, all of them do -- also in the MIDDLE of a fit, which is what a
    cold fit across shards is (the codes are built inside it).  Frequencies and iteration counts equal the oracle's on three fits
    in a row."""
    import re
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(_MIXED_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), variant], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "RANK %d OK" % r in o
    sweeps = [eval(re.search(r"sweeps per fit (\[.*?\])", o).group(1)) for o in outs]
    iters = max(eval(re.search(r"iterations (\[.*?\])", outs[0]).group(1)))
    assert sweeps[0] == sweeps[1]                                   # the ranks enqueue the same sweeps
    if variant == "one_rank_without_codes":
        assert all(s >= iters for s in sweeps[0])                   # one rank never can: nobody ever runs two iterations per sweep
    else:
        assert sweeps[0][0] < iters, sweeps                         # fusion switched on inside the cold fit, on both ranks at the same sweep
        assert sweeps[0][2] <= sweeps[0][0]


def test_ranks_out_of_step_in_the_fit_fail_loudly(tmp_path):
    """The constellation of commit 807a461 replayed with the agreement switched off (test hook em_fuse_without_agreement): rank 0 has
    its codes from the first sweep and runs two iterations per sweep at once, rank 1 -- whose codes' memory arrives after its
    fourth sweep -- runs one.  Before the fix this paired the sums of different iterations and returned wrong frequencies without a
    word.  Now the tag of the very first all-reduce differs (3 fits running 6 iterations against 3 running 3): both ranks stop
    with status 76 and a message naming both tuples; nothing is retried."""
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(_MIXED_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "memory_late_and_no_agreement"], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 76, "rank %d: status %s\n%s" % (r, p.returncode, o[-3000:])
        assert "OK" not in o.replace("MISMATCH", "")
        assert "collective mismatch" in o and "EM convergence sums" in o
        assert "shape 3 / 6" in o and "shape 3 / 3" in o, o[-3000:]


def test_command_line_leaves_with_76_when_ranks_fall_out_of_step(tmp_path):
    """The same divergence through the COMMAND LINE (two ranks over the TCP star on the one GPU, the bundled AMRE file, populations
    of 13-23 individuals sent through the class codes): rank 1's codes arrive after its fourth sweep and the agreement is switched
    off, so the first all-reduce of the fit pairs a sweep of two iterations with a sweep of one.  Every rank must print the
    mismatch and leave with status 76 (wgsassign_amd.comm.COMM_DIVERGED) -- no output file of the fit, nothing retried."""
    from conftest import GOLDEN
    data = os.path.join(GOLDEN, "data")
    port = free_port()
    script = tmp_path / "rank.py"
    script.write_text(r'''
import os, sys
sys.path.insert(0, %r)
rank = int(os.environ["RANK"])
if rank == 1:
    os.environ["WGSASSIGN_CODES_ALLOC_WAIT_MS"] = "0"
from wgsassign_amd import device
if rank == 1:
    device.debug_hook("codes_alloc_release_after_sweeps", 4)
device.debug_hook("em_fuse_without_agreement", 1)
from wgsassign_amd.WGSassign import main
main(sys.argv[1:])
''' % ROOT)
    args = ["--beagle", os.path.join(data, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"),
            "--pop_af_IDs", os.path.join(data, "amre.breeding.ind85.reference_k5.IDs.txt"), "--get_reference_af", "--out", "diverged"]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   WGSASSIGN_COMM="socket", WGSASSIGN_DEVICE="0", WGSASSIGN_EM_CODES_MIN="8", WGSASSIGN_INDEX_DIR=str(tmp_path), PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)] + args, cwd=tmp_path, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 76, "rank %d: status %s\n%s" % (r, p.returncode, o[-3000:])
        assert "collective mismatch" in o and "EM convergence sums" in o, o[-3000:]
    assert not (tmp_path / "diverged.pop_af.npy").exists()
