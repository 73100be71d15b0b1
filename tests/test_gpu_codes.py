"""The kernels that go through the class codes (csrc/common.h: wgs_codes -- per-SNP dictionaries of the distinct (g0, g1)
pairs, one byte per (SNP, individual)) against the direct kernels and the oracle: the same bits everywhere -- for every
geometry of the encoder's hash tables, for quality-dependent likelihoods with many classes per SNP, for SNPs and tiles
with more classes than the tables hold (taken from the float32 slabs, SNP by SNP / tile by tile), and an unnoticed
fall-back to the direct kernels for matrices that are not worth coding."""
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


@pytest.mark.parametrize("m,n,K,slots", [(1, 3, 1, 64), (63, 9, 2, 128), (4097, 37, 4, 256), (20_011, 61, 5, 64), (70_003, 103, 7, 128), (300_017, 64, 10, 64),
                                         (40_000, 130, 13, 256), (9_000, 257, 20, 64), (20_000, 230, 3, 128), (6_000, 499, 4, 256), (6_000, 499, 4, 0)])
def test_coded_kernels_equal_direct_kernels_and_oracle(dev, oracle, m, n, K, slots, monkeypatch):
    """EM fit (all populations to convergence) and the n x K sums, through the codes and directly: identical iteration
    counts, frequencies and float64 sums; odd sizes, quads that straddle the end of a slab, short matrices (blocks split
    over several workgroups), K beyond one register batch; every geometry of the encoder (64 / 128 / 256 hash slots per SNP =
    64 / 32 / 16 SNPs per wavefront; 0: its own choice)."""
    monkeypatch.setenv("WGSASSIGN_EM_CODES_SWEEPS", "0")       # the EM fit asks for the codes from its first sweep
    if slots:
        monkeypatch.setenv("WGSASSIGN_CODES_TABLE", str(slots))    # (also codes matrices too small to be worth it)
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
        assert info["rich_snp_share"] == 0 and info["hash_slots"] == (slots or 64) and info["em_table_rows"] > 0
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


def test_matrices_not_worth_coding_take_the_direct_kernels(dev, oracle, monkeypatch):
    """Deep coverage -- every individual its own likelihoods, as many classes per SNP as individuals: the sample pass says
    so, no codes are built, the direct kernels run, the results are the oracle's.  A matrix that becomes worth coding after
    its rows are replaced is coded then; ten deep-coverage rows among the others stay UNCODED rows of a coded matrix
    (round 3 dropped the whole matrix's codes for one such row) and every result still is the oracle's."""
    monkeypatch.setenv("WGSASSIGN_EM_CODES_SWEEPS", "0")
    monkeypatch.setenv("WGSASSIGN_EM_CODES_MIN", "1")
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
        # replace the rows by low-depth data: worth coding now, and the codes follow the new contents
        L2, _ = synth.make_beagle(m, n, K, seed=2)
        b.upload_rows(L2, 0)
        info = b.codes_info()
        assert info["available"] and info["rich_snp_share"] == 0
        it2, af2, out2 = fit_and_score(dev, b, K, np.bincount(group_of))
        with quiet():
            _, af_o2, _, it_o2 = oracle.fit_reference_af(L2, IDs, t=4)
        assert it2 == [int(x) for x in it_o2] and same(af2, af_o2)
        assert same_nan(out2.astype(np.float32), oracle.assignLL(L2, af_o2.copy(), 4))
        # ten deep-coverage rows: ten uncoded SNPs in ONE tile (rows 5..14), everything else stays coded
        b.upload_rows(L[:10], 5)
        L3 = L2.copy()
        L3[5:15] = L[:10]
        info = b.codes_info()
        assert info["available"] and abs(info["rich_snp_share"] - 10 / m) < 1e-9 and 0 < info["em_direct_tile_share"] <= 2 / 47
        it3, af3, out3 = fit_and_score(dev, b, K, np.bincount(group_of))
        with quiet():
            _, af_o3, _, it_o3 = oracle.fit_reference_af(L3, IDs, t=4)
        assert it3 == [int(x) for x in it_o3] and same(af3, af_o3)
        assert same_nan(out3.astype(np.float32), oracle.assignLL(L3, af_o3.copy(), 4))
    b.close()


@pytest.mark.parametrize("m,n,K,quals,coded", [(6_000, 400, 4, synth.QUAL_BINS, True), (3_000, 1000, 10, synth.QUAL_BINS, True),
                                               (40_000, 200, 5, synth.QUAL_BINS, True), (3_000, 600, 3, (30, 34), False)])
def test_quality_dependent_likelihoods(dev, oracle, m, n, K, quals, coded, monkeypatch):
    """Likelihoods as ANGSD writes them for real reads (every read its own error rate, tests/synth.py: make_beagle_quality):
    40-100 classes per SNP instead of the ~27 of the fixed-error generator -- larger hash tables, 8 or 4 SNPs per table of
    the coded scoring sweep, some SNPs and tiles beyond the tables.  Same iterations, frequencies and sums as the direct
    kernels and the oracle.  (The last case, five equally likely qualities: 150 classes among 600 individuals -- the sample
    pass finds the largest table overflowing and the matrix stays uncoded.)"""
    monkeypatch.setenv("WGSASSIGN_EM_CODES_SWEEPS", "0")
    L, IDs = synth.make_beagle_quality(m, n, K, seed=5, quals=quals)
    ncls = synth.classes_per_snp(L)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    with codes(False):
        it0, af0, out0 = fit_and_score(dev, b, K, counts)
    with codes(True):
        info = b.codes_info()
        assert info["available"] == coded
        if coded:
            assert info["hash_slots"] >= (128 if ncls.mean() > 40 else 64)
            # the dictionary agrees with a NumPy count wherever the SNP was coded
            assert info["max_classes"] <= ncls.max() and abs(info["mean_classes"] - ncls.mean()) < 0.05 * ncls.mean() + 20 * info["rich_snp_share"]
        it1, af1, out1 = fit_and_score(dev, b, K, counts)
    assert it1 == it0 and same(af1, af0) and same_nan(out1, out0)
    b.close()
    with quiet():
        _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
    assert it1 == [int(x) for x in it_o] and same(af1, af_o)
    with np.errstate(all="ignore"):
        assert same_nan(out1.astype(np.float32), oracle.assignLL(L, af_o.copy(), 4))


def test_snps_beyond_the_tables_are_scored_from_the_slab(dev, oracle, monkeypatch):
    """A low-depth matrix in which every 250th SNP has as many classes as individuals: those SNPs are left uncoded (ncls = 0)
    and the coded scoring sweep takes exactly them from the float32 slab, inside the same launch; the EM sweep takes
    their tiles directly.  The bits of the direct kernels and of the oracle, in both arithmetic modes."""
    monkeypatch.setenv("WGSASSIGN_EM_CODES_SWEEPS", "0")
    m, n, K = 20_000, 160, 4
    L, IDs = synth.make_beagle(m, n, K, seed=21)
    rng = np.random.default_rng(2)
    deep = np.arange(7, m, 250)
    g = rng.dirichlet((0.7, 0.7, 0.7), size=(len(deep), n))
    L[deep, 0::2] = np.round(g[:, :, 0], 6)
    L[deep, 1::2] = np.round(g[:, :, 1], 6)
    L[deep[3], 0] = np.float32(np.nan)                          # special values on the direct path of the coded sweep
    L[deep[4], 2:4] = [0.0, 0.0]
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    from wgsassign_amd._lib import MODE_FAST
    with codes(False):
        it0, af0, out0 = fit_and_score(dev, b, K, counts)
        _, _, fast0 = fit_and_score(dev, b, K, counts, mode=MODE_FAST)
    with codes(True):
        info = b.codes_info()
        assert info["available"] and abs(info["rich_snp_share"] - len(deep) / m) < 1e-9
        it1, af1, out1 = fit_and_score(dev, b, K, counts)
        _, _, fast1 = fit_and_score(dev, b, K, counts, mode=MODE_FAST)
    assert it1 == it0 and same(af1, af0) and same_nan(out1, out0) and same_nan(fast1, fast0)
    b.close()
    with quiet():
        _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
    assert it1 == [int(x) for x in it_o] and same_nan(af1, af_o)
    with np.errstate(all="ignore"):
        assert same_nan(out1.astype(np.float32), oracle.assignLL(L, af_o.copy(), 4))


@pytest.mark.parametrize("slots", [64, 128, 256])
def test_small_quotient_table_sends_many_tiles_to_the_direct_path(dev, oracle, slots, monkeypatch):
    """WGSASSIGN_EM_TABLE_ROWS=16 on 30 individuals per population (~10 classes per slab and SNP): one tile in thirteen has a SNP with more
    classes than rows and are swept from the float32 slab inside the coded sweep, the others through the table -- same
    iterations and frequencies as the direct kernels and the oracle, and the reported share of direct tiles says so."""
    monkeypatch.setenv("WGSASSIGN_EM_CODES_SWEEPS", "0")
    monkeypatch.setenv("WGSASSIGN_EM_TABLE_ROWS", "16")
    monkeypatch.setenv("WGSASSIGN_CODES_TABLE", str(slots))
    m, n, K = 50_000, 60, 2
    L, IDs = synth.make_beagle(m, n, K, seed=31)
    monkeypatch.setenv("WGSASSIGN_EM_CODES_MIN", "1")
    group_of = (np.arange(n) // (n // K)).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    # how many (slab, tile) pairs NumPy sees with more than 16 classes in some SNP
    tiles = (m + 63) // 64
    over = 0
    for k in range(K):
        cl = synth.classes_per_snp(L[:, 2 * 30 * k:2 * 30 * (k + 1)])
        cl = np.concatenate([cl, np.ones(tiles * 64 - m, dtype=cl.dtype)])
        over += int((cl.reshape(tiles, 64).max(axis=1) > 16).sum())
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    with codes(False):
        it0, af0, _ = fit_and_score(dev, b, K, counts)
    with codes(True):
        it1, af1, _ = fit_and_score(dev, b, K, counts)
        info = b.codes_info()
    assert info["em_table_rows"] == 16 and abs(info["em_direct_tile_share"] - over / (K * tiles)) < 1e-9 and 0.02 < info["em_direct_tile_share"] < 0.98
    assert it1 == it0 and same(af1, af0)
    with quiet():
        _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
    assert it1 == [int(x) for x in it_o] and same(af1, af_o)
    b.close()


@pytest.mark.parametrize("wait_ms", ["0", "0.05"])
def test_sweeps_do_not_wait_for_the_codes_memory(dev, oracle, wait_ms, monkeypatch):
    """The codes' device memory comes from a helper thread (hipMalloc of VRAM an earlier process used takes seconds); a sweep that
    wants the codes waits WGSASSIGN_CODES_ALLOC_WAIT_MS for it (3 by default; 0 here) and otherwise runs over the float32 slabs --
    the fit then switches to the coded sweep (two iterations per pass) in the middle, or never: same iterations and frequencies as
    the direct kernels and the oracle either way, and the codes are there once somebody waits for them."""
    monkeypatch.setenv("WGSASSIGN_CODES_ALLOC_WAIT_MS", wait_ms)
    monkeypatch.setenv("WGSASSIGN_EM_CODES_MIN", "1")
    m, n, K = 60_000, 150, 3
    L, IDs = synth.make_beagle(m, n, K, seed=44)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    with codes(False):
        b0 = dev.DeviceBeagle.from_host(L, group_of, K)
        it0, af0, out0 = fit_and_score(dev, b0, K, counts)
        b0.close()
    with codes(True):
        b = dev.DeviceBeagle.from_host(L, group_of, K)
        assert b.codes_state() == 0
        it1, af1, out1 = fit_and_score(dev, b, K, counts)
        assert b.codes_state() in (0, 1)                        # (built in the middle of the fit, by the scoring sweep, or not yet)
        assert b.codes_wait() >= 0.0 and b.codes_wait() == 0.0  # (waits for an allocation in flight; none is afterwards)
        info = b.codes_info()                                    # builds, if nothing has yet
        assert info["available"] and b.codes_state() == 1
        it2, af2, out2 = fit_and_score(dev, b, K, counts)
        b.close()
    assert it1 == it0 == it2 and same(af1, af0) and same(af2, af0) and same_nan(out1, out0) and same_nan(out2, out0)
    with quiet():
        _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
    assert it1 == [int(x) for x in it_o] and same(af1, af_o)


def test_allocation_time_is_accounted(dev):
    """wgs_malloc_seconds: a running total of what hipMalloc took through the library (the driver clears VRAM an earlier process
    used, so the same call costs 0.3 ms or seconds; bench.py reports the share of every whole path)."""
    before = dev.malloc_seconds()
    L, IDs = synth.make_beagle(5_000, 12, 2, seed=3)
    group_of = np.searchsorted(np.unique(IDs[:, 1]), IDs[:, 1]).astype(np.int32)
    b = dev.DeviceBeagle.from_host(L, group_of, 2)
    em = dev.EMBatch(b, np.arange(2, dtype=np.int32))
    after = dev.malloc_seconds()
    em.close()
    b.close()
    assert 0.0 <= before < after < before + 60.0 and dev.malloc_seconds() >= after


def test_a_scoring_sweep_builds_the_codes_without_the_slab_numbering(dev, oracle):
    """--get_pop_like alone never fits: a build that a scoring sweep asks for leaves out the slabs' own numbering (the coded EM sweep's
    tables: a third of the encode pass, two thirds of the memory).  An EM fit on the same matrix afterwards rebuilds in full; same
    results as the direct kernels in every order."""
    m, n, K = 40_000, 150, 3
    L, IDs = synth.make_beagle(m, n, K, seed=46)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    with quiet():
        _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
    afs = dev.AFSet.from_host(af_o)
    with codes(False):
        b0 = dev.DeviceBeagle.from_host(L, group_of, K)
        out0, _ = dev.assign(b0, afs)
        b0.close()
    with codes(True):
        b = dev.DeviceBeagle.from_host(L, group_of, K)
        out1, _ = dev.assign(b, afs)
        first = b.codes_info()
        assert b.codes_state() == 1 and first["em_table_rows"] == 0 and first["slab_numbering_bytes"] == 0
        it, af, out2 = fit_and_score(dev, b, K, counts)
        second = b.codes_info()
        assert second["em_table_rows"] > 0 and second["slab_numbering_bytes"] > 0 and second["bytes"] > first["bytes"]
        b.close()
    afs.close()
    assert same_nan(out1, out0) and it == [int(x) for x in it_o] and same(af, af_o)


def test_a_slow_allocation_is_not_waited_for(dev, oracle, monkeypatch):
    """The test hook codes_alloc_delay_ms makes the helper thread's hipMalloc take 400 ms (the driver takes seconds for VRAM an earlier
    process used).  With no waiting allowed the fit and the scoring call run over the float32 slabs and return long before the memory;
    kernel times read -1 meanwhile (hipEventElapsedTime would wait for the allocation) and are there after codes_wait(); the next fit
    builds the codes.  Same results throughout."""
    import time
    monkeypatch.setenv("WGSASSIGN_CODES_ALLOC_WAIT_MS", "0")
    dev.debug_hook("codes_alloc_delay_ms", 400)
    m, n, K = 50_000, 120, 3
    L, IDs = synth.make_beagle(m, n, K, seed=45)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    counts = np.bincount(group_of, minlength=K)
    with quiet():
        _, af_o, _, it_o = oracle.fit_reference_af(L, IDs, t=4)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    em = dev.EMBatch(b, np.arange(K, dtype=np.int32))
    t0 = time.perf_counter()
    it1 = em.run(200, 1e-4)
    dt = time.perf_counter() - t0
    assert dt < 0.3 and b.codes_state() == 0                 # did not wait; nothing built
    assert em.fit_stats()[3] == -1.0                         # sweep kernel time: not readable yet
    afs = dev.AFSet(b.m, K, ctx=b.ctx)
    for k in range(K):
        em.clamp(k, int(counts[k]))
        afs.set_column_from_em(k, em, k)
    out1, _ = dev.assign(b, afs)
    assert dev.assign.last_ms == -1.0 and b.codes_state() == 0
    waited = b.codes_wait()
    dev.debug_hook("codes_alloc_delay_ms", 0)
    assert waited >= 350.0 and em.fit_stats()[3] > 0.0 and dev.last_assign_ms(b.ctx) > 0.0
    af1 = np.stack([em.get_f(k) for k in range(K)], axis=1)
    em.close()
    em = dev.EMBatch(b, np.arange(K, dtype=np.int32))
    it2 = em.run(200, 1e-4)
    assert b.codes_state() == 1 and em.fit_stats()[3] > 0.0  # the memory is there: built by the first sweep
    for k in range(K):
        em.clamp(k, int(counts[k]))
    af2 = np.stack([em.get_f(k) for k in range(K)], axis=1)
    out2, _ = dev.assign(b, afs)
    assert dev.assign.last_ms > 0.0
    em.close()
    afs.close()
    b.close()
    assert list(it1) == list(it2) == [int(x) for x in it_o] and same(af1, af_o) and same(af2, af_o) and same_nan(out1, out2)


def test_codes_are_built_only_when_they_can_pay(dev, monkeypatch):
    """The cost model of csrc/api.hip: em_codes_pay (a 6.4 GB device-generated matrix, 100 individuals per population: the
    encode pass costs about four direct sweeps, a coded sweep saves half of one) -- a fit with three iterations ahead sweeps the
    float32 slabs and builds nothing; a fit with many builds the codes at its first sweep; a step-by-step caller gets them after
    three direct sweeps; a scoring sweep with shared columns always builds them; a small matrix (fixed costs dominate) never
    builds them for an EM fit.  Same frequencies either way (the tests above)."""
    monkeypatch.delenv("WGSASSIGN_EM_CODES_SWEEPS", raising=False)
    monkeypatch.delenv("WGSASSIGN_SCORE_CODES_ALWAYS", raising=False)
    m, n, K = 2_000_000, 400, 4
    group_of = (np.arange(n) // (n // K)).astype(np.int32)

    def matrix(m=m):
        b = dev.DeviceBeagle(m, n, group_of, K)
        b.synth(77, 2.0)
        return b

    with codes(True):
        b = matrix()
        em = dev.EMBatch(b, np.arange(K, dtype=np.int32))
        em.fit(3, 0.0)                                           # three iterations at most: not worth an encode pass
        assert b.codes_state() == 0
        em.fit(50, 0.0)                                          # fifty: built at the first sweep
        assert b.codes_state() == 1
        em.close()
        b.close()
        b = matrix()
        em = dev.EMBatch(b, np.arange(K, dtype=np.int32))
        for i in range(6):                                       # step by step: direct at first, coded once the run is long
            em.step()
            assert (b.codes_state() == 1) == (i >= 3), i
        em.close()
        b.close()
        b = matrix()
        afs = dev.AFSet.from_host(np.full((m, K), 0.3, dtype=np.float32))
        dev.assign(b, afs)
        assert b.codes_state() == 1
        afs.close()
        b.close()
        b = matrix(20_000)                                       # 64 MB: the encode pass's fixed costs exceed what 14 sweeps save
        em = dev.EMBatch(b, np.arange(K, dtype=np.int32))
        em.fit(200, 0.0)
        assert b.codes_state() == 0
        afs = dev.AFSet.from_host(np.full((20_000, K), 0.3, dtype=np.float32))
        dev.assign(b, afs)                                       # ... and what one scoring sweep saves
        assert b.codes_state() == 0
        afs.close()
        em.close()
        b.close()
        # quality-dependent likelihoods, 26 classes per slab among 100 individuals: a coded EM sweep saves a fifth of the direct
        # one and the encoder needs its largest tables -- the fit keeps the float32 slabs (the sample pass said so); ONE scoring
        # sweep over 4 populations saves less than that encode pass costs (the model's direct-sweep time was ten times too long
        # until round 5 put its predictions beside bench.py's measurements: it built here), one over 16 populations repays it
        b = dev.DeviceBeagle(m, n, group_of, K)
        b.synth_quality(77, 2.0)
        em = dev.EMBatch(b, np.arange(K, dtype=np.int32))
        em.fit(50, 0.0)
        assert b.codes_state() == 0
        afs = dev.AFSet.from_host(np.full((m, K), 0.3, dtype=np.float32))
        dev.assign(b, afs)
        assert b.codes_state() == 0
        md = b.codes_model(K)
        assert not md["builds_for_scoring"] and md["score_float32_sweep_ms"] * (1 - md["score_share_of_the_coded_sweep"]) < md["encode_for_scoring_ms"]
        afs.close()
        afs = dev.AFSet.from_host(np.full((m, 16), 0.3, dtype=np.float32))
        assert b.codes_model(16)["builds_for_scoring"]
        dev.assign(b, afs)
        assert b.codes_state() == 1
        afs.close()
        em.close()
        b.close()


def test_row_ranges_and_leave_one_out_are_unchanged(dev, oracle):
    """A scoring object restricted to a range of individuals goes through the codes too; leave-one-out (per-individual
    columns, several fits per slab: populations of 15 stay with the float32 group kernel) -- both give the bits they gave without codes."""
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


@pytest.mark.parametrize("rows,min_cols,n,P", [(None, None, 120, 2), ("8", None, 120, 1), (None, "1", 36, 3), (None, None, 200, 1)])
def test_leave_one_out_through_the_codes(dev, oracle, rows, min_cols, n, P, monkeypatch):
    """Leave-one-out re-fits (glassy.py:65-85: every individual's population without it) through the class codes: em_coded_group_kernel
    walks the fits of a slab one after the other over a tile whose dictionary rows stay in registers.  Same iterations, log-likelihoods
    and partition sums as the float32 group kernel (WGSASSIGN_LOO_CODES=0) and as the oracle; with an 8-row table most tiles take the
    kernel's term-by-term path; with the population limit lowered, populations of 12."""
    from wgsassign_amd import glassy
    if rows:
        monkeypatch.setenv("WGSASSIGN_EM_TABLE_ROWS", rows)
    if min_cols:
        monkeypatch.setenv("WGSASSIGN_EM_CODES_MIN", min_cols)
    m, K = 12_000, 3
    L, IDs = synth.make_beagle(m, n, K, seed=31 + n)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    with quiet():
        _, af, _, _ = oracle.fit_reference_af(L, IDs, t=4)
    res = {}
    for on in ("0", "1"):
        monkeypatch.setenv("WGSASSIGN_LOO_CODES", on)
        b = dev.DeviceBeagle.from_host(L, group_of, K)
        tm = {}
        a = af.copy()
        with quiet():
            ll, parts = glassy.loo_device(b, b, a, group_of, 200, 1e-4, P, verbose=False, timings=tm, need_parts=True)
        res[on] = (ll, parts, tm["iters"].copy(), a)
        assert b.codes_state() == (1 if on == "1" else 0)          # (only the re-fits' sweeps ask for codes here: per-individual columns are scored directly)
        if on == "1" and rows:
            assert b.codes_info()["em_direct_tile_share"] > 0.5
        b.close()
    assert same_nan(res["1"][0], res["0"][0]) and same(res["1"][1], res["0"][1]) and list(res["1"][2]) == list(res["0"][2]) and same(res["1"][3], res["0"][3])
    if n <= 120:
        with quiet():
            ll_o, parts_o = oracle.loo(L, af.copy(), IDs, 4, 200, 1e-4, None, P)
        assert same_nan(res["1"][0].astype(np.float32), ll_o) and same(res["1"][1], parts_o)


def test_tiles_richer_than_the_quotient_table_are_swept_directly(dev, oracle, monkeypatch):
    """The coded EM sweep sizes its table (rows per SNP) so that ~1 % of the tiles at most have a SNP with more classes in a
    slab; those tiles are swept from the float32 slab inside the same kernel.  A matrix of low-depth sites with a few rich
    ones (every individual its own likelihoods, but few enough classes to stay codable): same iterations and frequencies as
    the direct kernels and the oracle, with and without a left-out individual."""
    monkeypatch.setenv("WGSASSIGN_EM_CODES_MIN", "1")
    monkeypatch.setenv("WGSASSIGN_EM_CODES_SWEEPS", "0")
    m, n, K = 64 * 700 + 5, 90, 2
    L, IDs = synth.make_beagle(m, n, K, seed=12)
    rng = np.random.default_rng(5)
    rich = rng.choice(m, size=6, replace=False)                 # 6 sites in 6 of 701 tiles: the 1 % rule leaves them out
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
