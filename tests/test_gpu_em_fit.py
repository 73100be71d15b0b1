"""wgs_em_fit / wgs_loo (one C call, iterations enqueued ahead of the host, decisions on the device) against
the step-by-step protocol of wgsassign_amd.device.run_em / glassy.loo_device and the golden vectors:
identical iteration counts, frequencies, sums and partition sums."""
import numpy as np
import pytest

import synth
from test_gpu_parity import quiet, same, same_nan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wg():
    from wgsassign_amd import device, emMAF, glassy
    device.get_context()

    class NS:
        pass
    ns = NS()
    ns.device, ns.emMAF, ns.glassy = device, emMAF, glassy
    return ns


@pytest.mark.parametrize("guard", [0.0, 1e9])
def test_c_loop_equals_python_loop(wg, monkeypatch, guard):
    """40 leave-one-out fits + 4 population fits in one batch: wgs_em_fit and run_em stop every fit at the
    same iteration with the same bits -- with the exact chain deciding only inside the band, and (guard = 1e9)
    with EVERY decision parked for the exact chain, which exercises the park / resolve / re-activate path on
    every iteration."""
    dev = wg.device
    m, n, K = 30_000, 40, 4
    L, IDs = synth.make_beagle(m, n, K, seed=21)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    groups = np.concatenate([group_of, np.arange(K)]).astype(np.int32)
    skips = np.concatenate([np.arange(n), -np.ones(K)]).astype(np.int32)
    monkeypatch.setattr(dev.EMBatch, "GUARD", guard)
    res = {}
    for loop in ("c", "python"):
        monkeypatch.setenv("WGSASSIGN_EM_LOOP", loop)
        em = dev.EMBatch(b, groups, skips)
        iters = em.run(200, 1e-4)
        res[loop] = (iters.copy(), np.stack([em.get_f(j) for j in range(n + K)]), em.active.copy())
        if loop == "c":
            launched, batches, seconds, sweep_ms = em.fit_stats()
            assert 0 < sweep_ms < seconds * 1e3
            # one iteration of lookahead; a parked fit sweeps every other iteration
            assert launched >= iters.max() + 1 and (launched >= 2 * iters.max() if guard > 1 else launched <= iters.max() + 4)
            assert (batches >= iters.max()) if guard > 1 else (batches <= 6)
        em.close()
    assert np.array_equal(res["c"][0], res["python"][0]) and res["c"][0].min() > 0
    assert same(res["c"][1], res["python"][1])
    assert np.array_equal(res["c"][2], res["python"][2]) and not res["c"][2].any()
    b.close()


def test_c_loop_exhaustion_nan_and_preset_inactive(wg, golden):
    dev = wg.device
    g = golden("edge.npz")
    L = g["mixed_L"]
    b = dev.DeviceBeagle.from_host(L)
    em = dev.EMBatch(b, [0, 0, 0])
    em.set_active(1, False)                       # a frozen fit stays at 0.25 and reports 0
    iters = em.fit(3, 1e-12)
    assert list(iters) == [0, 0, 0]
    assert same(em.get_f(0), g["exhaust_f"]) and same(em.get_f(2), g["exhaust_f"])
    assert np.all(em.get_f(1) == np.float32(0.25))
    assert em.fit_stats()[0] == 3
    em.close()
    # tole = 0 / NaN: `diff < tole` never holds, every iteration runs
    em = dev.EMBatch(b, [0])
    assert list(em.fit(5, 0.0)) == [0] and em.fit_stats()[0] == 5
    em.close()
    b.close()
    # a population of size 0 under leave-one-out: NaN forever (golden single_* of test_loo_population_of_one)
    af = g["single_af"].copy()
    with np.errstate(all="ignore"):
        (ll, parts), _ = quiet(wg.glassy.loo, g["single_L"], af, g["single_IDs"], 1, 20, 1e-4, None, 2)
    assert same_nan(parts, g["single_parts"]) and same(af, g["single_af_after"])


@pytest.mark.parametrize("batch", [None, "7"])
def test_wgs_loo_equals_stepwise_orchestration(wg, golden, monkeypatch, batch):
    """glassy.loo through the single C entry (wgs_loo) and through the Python orchestration over the step-wise
    entry points: same bits, also when the re-fits run in batches."""
    if batch:
        monkeypatch.setenv("WGSASSIGN_LOO_BATCH", batch)
    g, fit = golden("amre_loo.npz"), golden("amre_fit.npz")
    out = {}
    for how in ("c", "python"):
        monkeypatch.setenv("WGSASSIGN_LOO", how)
        af = fit["pop_af"].copy()
        (ll, parts), text = quiet(wg.glassy.loo, fit["L"], af, fit["IDs"], 1, 200, 1e-4, None, 3)
        out[how] = (ll, parts, af, text)
        assert same(parts, g["parts_P3"]) and same(af, g["af_after_P3"])
    for x, y in zip(out["c"], out["python"]):
        assert (x == y) if isinstance(x, str) else same(x, y)


def test_native_communicator_handle_in_the_c_loops(wg, golden):
    """The RCCL code path of wgs_em_fit / wgs_loo (all-reduce enqueued behind the sums, carries broadcast
    through the communicator) with a one-rank communicator: same results as without."""
    from wgsassign_amd import _lib
    from wgsassign_amd import comm as wcomm
    dev = wg.device
    fit = golden("amre_fit.npz")
    pops = np.unique(fit["IDs"][:, 1])
    group_of = np.searchsorted(pops, fit["IDs"][:, 1]).astype(np.int32)
    c = wcomm.RcclComm(dev.get_context(), 0, 1)
    assert c.native and c.handle is not None
    b = dev.DeviceBeagle.from_host(fit["L"], group_of, len(pops))
    em = dev.EMBatch(b, np.arange(5, dtype=np.int32))
    em.GUARD = 1e9                                     # every decision through the chain + the carry broadcast
    iters = np.zeros(5, dtype=np.int32)
    _lib.check(_lib.load().wgs_em_fit(em.handle, 200, 1e-4, b.m, c.handle, 1e9, _lib.i32p(iters)))
    assert list(iters) == [17, 14, 16, 14, 13]
    for k in range(5):
        assert same(em.get_f(k), fit["f_raw"][k])
    em.close()
    g = golden("amre_loo.npz")
    afs = dev.AFSet.from_host(fit["pop_af"].copy())
    n, K, P = b.n, 5, 3
    ll, parts, it = np.zeros((n, K)), np.zeros((n * P, K), dtype=np.float32), np.zeros(n, dtype=np.int32)
    _lib.check(_lib.load().wgs_loo(b.handle, None, afs.handle, 200, 1e-4, b.m, c.handle, P, 30, 0, 0, _lib.f64p(ll),
                                   _lib.f32p(parts), _lib.i32p(it)))
    assert same(parts, g["parts_P3"]) and same(afs.to_host(), g["af_after_P3"]) and it.min() > 0
    afs.close()
    b.close()
    c.close()


@pytest.mark.parametrize("guard", [0.0, 1e9, "band"])
@pytest.mark.parametrize("max_iter", [200, 7, 8, 1, 2])
def test_two_iterations_per_sweep_equal_one(wg, oracle, monkeypatch, guard, max_iter):
    """The coded sweep runs two EM iterations per pass over the codes (csrc/em_kernels.hip: fused iterations; the update of a SNP
    needs only that SNP's frequency, the convergence test only the sums).  Against one iteration per sweep
    (WGSASSIGN_EM_FUSE=1), the Python step-by-step protocol and the oracle: the same iteration counts and bits -- when fits stop
    at the first or the second iteration of a sweep (populations of different size converge at different counts), with every
    decision parked for the exact chain (guard = 1e9: both iterations of every sweep go through park / resolve), with a band
    wide enough to park only the last iterations, with an iteration limit that is odd, even, 1 and 2 (exhaustion: iters = 0)."""
    dev = wg.device
    monkeypatch.setenv("WGSASSIGN_EM_CODES_MIN", "1")
    m, K = 40_000, 6
    labels = np.repeat(np.arange(K), [9, 31, 40, 57, 23, 64])
    L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=33)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    assert b.codes_info()["available"]
    monkeypatch.setattr(dev.EMBatch, "GUARD", 0.5 if guard == "band" else guard)
    res = {}
    for label, env in (("fused", {"WGSASSIGN_EM_FUSE": "2"}), ("single", {"WGSASSIGN_EM_FUSE": "1"}), ("python", {"WGSASSIGN_EM_LOOP": "python"})):
        for k in ("WGSASSIGN_EM_FUSE", "WGSASSIGN_EM_LOOP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        em = dev.EMBatch(b, np.arange(K, dtype=np.int32))
        iters = em.run(max_iter, 1e-4)
        res[label] = (iters.copy(), np.stack([em.get_f(j) for j in range(K)]), np.stack([em.get_f_range(j, 0, m, previous=True) for j in range(K)]),
                      em.fit_stats()[0])
        em.close()
    for other in ("single", "python"):
        assert np.array_equal(res["fused"][0], res[other][0]), (res["fused"][0], res[other][0])
        assert same(res["fused"][1], res[other][1])
    # the vector BEFORE the last update is the right one too (what the exact chain and a later wgs_em_rmse_chain read)
    assert same(res["fused"][2], res["single"][2])
    if max_iter >= 8 and guard == 0.0:
        assert res["fused"][3] < res["single"][3]            # fewer sweeps were enqueued
    if max_iter == 200:
        assert len(set(res["fused"][0].tolist())) > 1 and res["fused"][0].min() > 0
        with quiet_ctx():
            _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
        assert list(res["fused"][0]) == [int(x) for x in it_o]
        em = dev.EMBatch(b, np.arange(K, dtype=np.int32))
        em.run(200, 1e-4)
        counts = np.bincount(group_of, minlength=K)
        cols = []
        for k in range(K):
            em.clamp(k, int(counts[k]))
            cols.append(em.get_f(k))
        em.close()
        assert same(np.stack(cols, axis=1), af_o)
    b.close()


class quiet_ctx:
    def __enter__(self):
        import contextlib
        import io
        self.cm = contextlib.redirect_stdout(io.StringIO())
        self.cm.__enter__()

    def __exit__(self, *exc):
        return self.cm.__exit__(*exc)
