"""The kernels that go through the class codes (csrc/common.h: wgs_codes -- per-SNP dictionaries of the distinct (g0, g1)
pairs, one byte per (SNP, individual)) against the direct kernels and the oracle: the same bits everywhere, for matrices
that can be coded, and an unnoticed fall-back to the direct kernels for matrices that cannot."""
import os

import numpy as np
import pytest

import synth
from test_gpu_parity import same, same_nan

pytestmark = pytest.mark.gpu


class quiet:
    """swallow the drivers' progress lines"""

    def __enter__(self):
        import contextlib
        import io
        self.cm = contextlib.redirect_stdout(io.StringIO())
        self.cm.__enter__()

    def __exit__(self, *exc):
        return self.cm.__exit__(*exc)


@pytest.fixture()
def dev():
    from wgsassign_amd import device
    device.get_context()
    return device


class codes:
    """with codes(False): the direct kernels; with codes(True): the coded ones (when the matrix can be coded)"""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("WGSASSIGN_CODES")
        os.environ["WGSASSIGN_CODES"] = "1" if self.on else "0"

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("WGSASSIGN_CODES")
        else:
            os.environ["WGSASSIGN_CODES"] = self.old


def fit_and_score(dev, b, K, counts, mode=None):
    em = dev.EMBatch(b, np.arange(K, dtype=np.int32))
    iters = em.run(200, 1e-4)
    cols = []
    afs = dev.AFSet(b.m, K, ctx=b.ctx)
    for k in range(K):
        em.clamp(k, int(counts[k]))
        cols.append(em.get_f(k))
        afs.set_column_from_em(k, em, k)
    out, _ = dev.assign(b, afs, mode=mode)
    em.close()
    afs.close()
    return [int(x) for x in iters], np.stack(cols, axis=1), out


@pytest.mark.parametrize("m,n,K", [(1, 3, 1), (63, 9, 2), (4097, 37, 4), (20_011, 61, 5), (70_003, 103, 7), (300_017, 64, 10), (40_000, 130, 13),
                                   (9_000, 257, 20), (20_000, 230, 3), (6_000, 499, 4)])
def test_coded_kernels_equal_direct_kernels_and_oracle(dev, oracle, m, n, K, monkeypatch):
    """EM fit (all populations to convergence) and the n x K sums, through the codes and directly: identical iteration
    counts, frequencies and float64 sums; odd sizes, quads that straddle the end of a slab, short matrices (blocks split
    over several workgroups), K beyond one register batch."""
    if n < 200:
        monkeypatch.setenv("WGSASSIGN_EM_CODES_MIN", "1")      # populations below 28 individuals through the coded EM sweep too
    rng = np.random.default_rng(m + n)
    labels = rng.integers(0, K, size=n)
    labels[:K] = np.arange(K)                                  # no empty population
    L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=3)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    with codes(False):
        it0, af0, out0 = fit_and_score(dev, b, K, counts)
    with codes(True):
        info = b.codes_info()
        assert info["available"] and 1 <= info["max_classes"] <= 64 and info["mean_classes"] <= info["max_classes"]
        it1, af1, out1 = fit_and_score(dev, b, K, counts)
        from wgsassign_amd._lib import MODE_FAST
        _, _, fast1 = fit_and_score(dev, b, K, counts, mode=MODE_FAST)
    with codes(False):
        from wgsassign_amd._lib import MODE_FAST
        _, _, fast0 = fit_and_score(dev, b, K, counts, mode=MODE_FAST)
    assert it1 == it0 and same(af1, af0) and same_nan(out1, out0) and same_nan(fast1, fast0)
    b.close()
    if m <= 70_003:
        with quiet():
            _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
        assert it1 == [int(x) for x in it_o] and same(af1, af_o)
        with np.errstate(all="ignore"):
            assert same_nan(out1.astype(np.float32), oracle.assignLL(L, af_o.copy(), 4))


def test_special_values_through_the_class_table(dev, oracle):
    """Likelihood sums of exactly 0 (log = -inf), NaN data, frequencies of exactly 0 and 1: the class table holds libm's
    special values and the sums carry them exactly like the direct sweep."""
    m, n, K = 5000, 24, 3
    L, IDs = synth.make_beagle(m, n, K, seed=8)
    L[::7, 0::2] = 0.0
    L[::7, 1::2] = 0.0                                           # (0, 0): g2 = 1
    L[3::11, 0] = np.nan
    L[5::13, 2:4] = [1.0, 0.0]
    A = np.random.default_rng(1).random((m, K)).astype(np.float32)
    A[::5, 0] = 0.0
    A[1::5, 1] = 1.0
    A[2::9, 2] = np.nan
    group_of = (np.arange(n) % K).astype(np.int32)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    afs = dev.AFSet.from_host(A)
    with codes(False):
        out0, _ = dev.assign(b, afs)
    with codes(True):
        assert b.codes_info()["available"]
        out1, _ = dev.assign(b, afs)
    assert same_nan(out1, out0) and (np.isinf(out1).any() or np.isnan(out1).any())
    with np.errstate(all="ignore"):
        assert same_nan(out1.astype(np.float32), oracle.assignLL(L, A.copy(), 4))
    afs.close()
    b.close()


def test_matrices_that_cannot_be_coded_take_the_direct_kernels(dev, oracle):
    """More than 64 distinct (g0, g1) pairs in a SNP (deep coverage: every individual its own likelihoods): no codes, the
    direct kernels run, the results are the oracle's.  A matrix that becomes codable after its rows are replaced is coded
    then, and the other way round."""
    m, n, K = 3000, 80, 2
    rng = np.random.default_rng(4)
    g = rng.dirichlet((0.7, 0.7, 0.7), size=(m, n))
    L = np.empty((m, 2 * n), dtype=np.float32)
    L[:, 0::2] = np.round(g[:, :, 0], 6)
    L[:, 1::2] = np.round(g[:, :, 1], 6)
    IDs = synth.pop_labels(n, K)
    group_of = (np.arange(n) // (n // K)).astype(np.int32)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    with codes(True):
        assert not b.codes_info()["available"]
        it, af, out = fit_and_score(dev, b, K, np.bincount(group_of))
        with quiet():
            _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
        assert it == [int(x) for x in it_o] and same(af, af_o)
        assert same_nan(out.astype(np.float32), oracle.assignLL(L, af_o.copy(), 4))
        # replace the rows by low-depth data: codable now, and the codes follow the new contents
        L2, _ = synth.make_beagle(m, n, K, seed=2)
        b.upload_rows(L2, 0)
        assert b.codes_info()["available"]
        it2, af2, out2 = fit_and_score(dev, b, K, np.bincount(group_of))
        with quiet():
            _, af_o2, _, it_o2 = oracle.fit_reference_af(L2, IDs, t=4)
        assert it2 == [int(x) for x in it_o2] and same(af2, af_o2)
        assert same_nan(out2.astype(np.float32), oracle.assignLL(L2, af_o2.copy(), 4))
        b.upload_rows(L[:10], 5)                                 # ten deep-coverage rows: not codable any more
        assert not b.codes_info()["available"]
    b.close()


def test_row_ranges_and_leave_one_out_are_unchanged(dev, oracle):
    """A scoring object restricted to a range of individuals goes through the codes too; leave-one-out (per-individual
    columns, several fits per slab) keeps the direct kernels -- both give the bits they gave without codes."""
    from wgsassign_amd import glassy
    m, n, K = 30_000, 45, 3
    L, IDs = synth.make_beagle(m, n, K, seed=6)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    with quiet():
        _, af, _, _ = oracle.fit_reference_af(L, IDs, t=4)
    afs = dev.AFSet.from_host(af)
    res = {}
    for on in (False, True):
        with codes(on):
            sc = dev.Score(b, afs, rows=(7, 29))
            res[on] = sc.sums()
            sc.close()
            a = af.copy()
            with quiet():
                res[on, "loo"] = glassy.loo_device(b, b, a, group_of, 200, 1e-4, 2, verbose=False)
    assert same_nan(res[True], res[False]) and not res[True][:7].any() and not res[True][29:].any() and res[True][7:29].all()
    assert same_nan(res[True, "loo"][0], res[False, "loo"][0]) and same(res[True, "loo"][1], res[False, "loo"][1])
    afs.close()
    b.close()


def test_tiles_richer_than_the_quotient_table_are_swept_directly(dev, oracle, monkeypatch):
    """The coded EM sweep sizes its table (rows per SNP) so that ~1 % of the tiles at most have a SNP with more classes in a
    slab; those tiles are swept from the float32 slab inside the same kernel.  A matrix of low-depth sites with a few rich
    ones (every individual its own likelihoods, but few enough classes to stay codable): same iterations and frequencies as
    the direct kernels and the oracle, with and without a left-out individual."""
    monkeypatch.setenv("WGSASSIGN_EM_CODES_MIN", "1")
    m, n, K = 64 * 700 + 5, 90, 2
    L, IDs = synth.make_beagle(m, n, K, seed=12)
    rng = np.random.default_rng(5)
    rich = rng.choice(m, size=12, replace=False)                # 12 sites in 12 of 701 tiles: the 1 % rule leaves them out
    vals = np.round(rng.dirichlet((0.7, 0.7, 0.7), size=(len(rich), 30)), 6)      # 30 distinct pairs
    pick = rng.integers(0, 30, size=(len(rich), n))
    for a, s in enumerate(rich):
        L[s, 0::2] = vals[a, pick[a], 0]
        L[s, 1::2] = vals[a, pick[a], 1]
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    with codes(False):
        it0, af0, _ = fit_and_score(dev, b, K, counts)
    with codes(True):
        it1, af1, _ = fit_and_score(dev, b, K, counts)
        info = b.codes_info()
        assert info["available"] and 0 < info["em_direct_tile_share"] <= 0.01 and info["em_table_rows"] < 8 * ((info["max_classes"] + 7) // 8)
        # leave-one-out style fits (a skipped column) of single slabs through the coded sweep
        em = dev.EMBatch(b, np.array([0, 1], dtype=np.int32), skips=np.array([3, 50], dtype=np.int32))
        its = [int(x) for x in em.run(200, 1e-4)]
        f_skip = [em.get_f(0), em.get_f(1)]
        em.close()
    with codes(False):
        em = dev.EMBatch(b, np.array([0, 1], dtype=np.int32), skips=np.array([3, 50], dtype=np.int32))
        its0 = [int(x) for x in em.run(200, 1e-4)]
        assert its == its0 and same(f_skip[0], em.get_f(0)) and same(f_skip[1], em.get_f(1))
        em.close()
    assert it1 == it0 and same(af1, af0)
    with quiet():
        _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
    assert it1 == [int(x) for x in it_o] and same(af1, af_o)
    b.close()
