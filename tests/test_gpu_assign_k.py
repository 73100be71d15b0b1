"""Assignment / leave-one-out scoring kernels at the population counts the BASELINE configs launch.

The scoring sweep (csrc/assign_kernels.hip) batches the K populations into register batches of KB <= 10:
K=6 -> KB=6, K=7 -> 7, K=8 -> 8 (config 4), K=9 -> 9, K=10 -> 10 (config 3, one pass), K=13 -> two passes of 7 with
a padded slot, K=17 -> two passes of 9 with a padded slot, K=20 -> two passes of 10 (config 5), K=23 -> three of 8, K=33 -> four of 9; NP = 2 individual
pairs per wave for KB <= 6, 1 above.  The chain kernel of the exact partition sums stays at KB <= 8 (K=10 -> two
passes of 5, K=20 -> three of 7).  Every one of them is held to the oracle here, in both
modes: shared frequency columns (glassy.py:31-42, --get_pop_like) and per-individual columns
(glassy.py:92-109, --loo), with odd population sizes and SNP counts that are not multiples of 64.
"""
import contextlib
import io

import numpy as np
import pytest

import synth
from test_gpu_parity import close, quiet, same, same_nan, RTOL_PARTS

pytestmark = pytest.mark.gpu

KS = [6, 7, 8, 9, 10, 13, 17, 20, 23, 33]


def odd_populations(K, rng):
    """Population labels: sizes 2..7 with both parities, individuals interleaved in file order."""
    sizes = [int(x) for x in rng.integers(2, 8, size=K)]
    sizes[0], sizes[-1] = 3, 5
    labels = np.repeat(np.arange(K), sizes)
    rng.shuffle(labels)
    return labels


def structured(m, K, rng, seed):
    """Genotype likelihoods drawn from K DIFFERENTIATED populations (a kernel that confuses columns cannot pass)."""
    return synth.make_beagle_for_labels(m, odd_populations(K, rng), K, seed=seed)


@pytest.fixture(scope="module")
def wg():
    from wgsassign_amd import device, emMAF, glassy
    device.get_context()

    class NS:
        pass
    ns = NS()
    ns.device, ns.emMAF, ns.glassy = device, emMAF, glassy
    return ns


@pytest.mark.parametrize("K", KS)
def test_assign_shared_columns_against_oracle(wg, oracle, K):
    rng = np.random.default_rng(500 + K)
    m = 1000 + 37 * K + 1
    L, IDs = structured(m, K, rng, 4000 + K)
    n = len(IDs)
    pops_o, af_o, _, iters_o = oracle.fit_reference_af(L, IDs, t=4)
    (pops, af, iters), _ = quiet(wg.emMAF.emMAF_populations, L, IDs, 200, 1e-4)
    assert list(iters) == list(iters_o) and same(af, af_o)
    ll_o = oracle.assignLL(L, af_o.copy(), 4)
    # (a) --get_pop_like shape: all individuals in one slab
    ll, text = quiet(wg.glassy.assignLL, L, af.copy(), 1)
    assert text.strip() == "%d individuals to assign to %d populations" % (n, K)
    assert ll.dtype == np.float32 and same_nan(ll, ll_o), (K, "one slab")
    # (b) the same sums from K population slabs of odd sizes (the layout --get_reference_af leaves on the device)
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    b = wg.device.DeviceBeagle.from_host(L, group_of, K)
    afs = wg.device.AFSet.from_host(af)
    out, _ = wg.device.assign(b, afs)
    assert same_nan(out.astype(np.float32), ll_o), (K, "population slabs")
    # run-to-run reproducible, bit for bit, in float64
    out2, _ = wg.device.assign(b, afs)
    assert same(out, out2)
    afs.close()
    b.close()


@pytest.mark.parametrize("P", [1, 3])
@pytest.mark.parametrize("K", KS)
def test_loo_per_individual_columns_against_oracle(wg, oracle, K, P):
    rng = np.random.default_rng(900 + K)
    m = 600 + 29 * K + 3
    L, IDs = structured(m, K, rng, 5000 + K)
    n = len(IDs)
    _, af_o, _, _ = oracle.fit_reference_af(L, IDs, t=4)
    af1, af2 = af_o.copy(), af_o.copy()
    with np.errstate(all="ignore"):
        loo_o, parts_o = oracle.loo(L, af1, IDs, 4, 200, 1e-4, None, P)
        (loo, parts), _ = quiet(wg.glassy.loo, L, af2, IDs, 1, 200, 1e-4, None, P)
    assert same_nan(loo, loo_o), (K, P)
    assert same_nan(parts, parts_o), (K, P)          # serial float32 partition sums: bit-identical
    assert same_nan(af2, af1), (K, P)                # the sticky column overwrite


def test_loo_with_more_partitions_than_the_block_parallel_chains_take(wg, oracle):
    """--partition_sites 100 (> 64: the leave-one-out falls back from the single C call to the step-by-step driver and
    the one-lane-per-chain kernel): same bits as the oracle's np.add.at sums, same sticky columns."""
    K, P = 6, 100
    rng = np.random.default_rng(77)
    L, IDs = structured(1500, K, rng, 7100)
    _, af_o, _, _ = oracle.fit_reference_af(L, IDs, t=4)
    af1, af2 = af_o.copy(), af_o.copy()
    with np.errstate(all="ignore"):
        loo_o, parts_o = oracle.loo(L, af1, IDs, 4, 200, 1e-4, None, P)
        (loo, parts), _ = quiet(wg.glassy.loo, L, af2, IDs, 1, 200, 1e-4, None, P)
    assert parts.shape == (len(IDs) * P, K)
    assert same_nan(loo, loo_o) and same_nan(parts, parts_o) and same_nan(af2, af1)


@pytest.mark.parametrize("K", [8, 10, 13])
def test_fast_partition_kernel_large_K(wg, oracle, monkeypatch, K):
    """WGSASSIGN_PARTS=fast routes P > 1 through assign_kernel<KB> (lane <-> individual pair): float64
    partition sums, within the reference's own float32 accumulation noise."""
    monkeypatch.setenv("WGSASSIGN_PARTS", "fast")
    rng = np.random.default_rng(1300 + K)
    m = 777
    L, IDs = structured(m, K, rng, 6000 + K)
    n = len(IDs)
    _, af_o, _, _ = oracle.fit_reference_af(L, IDs, t=4)
    af1, af2 = af_o.copy(), af_o.copy()
    with np.errstate(all="ignore"):
        loo_o, parts_o = oracle.loo(L, af1, IDs, 4, 200, 1e-4, None, 3)
        (loo, parts), _ = quiet(wg.glassy.loo, L, af2, IDs, 1, 200, 1e-4, None, 3)
    assert close(loo, loo_o) and close(parts, parts_o, RTOL_PARTS) and same_nan(af2, af1)


@pytest.mark.parametrize("m, n, K", [(1501, 40, 6), (8192 + 77, 26, 5), (70_001, 30, 10), (300_000, 24, 8)])
def test_sums_are_numpys_own_float64_sums(wg, oracle, m, n, K):
    """glassy.py:38 stores np.sum(vec, dtype=float): NumPy adds, chunk of 8192 sites after chunk, the chunk's pairwise
    sum to the running float64 total.  block_prefix_kernel adds the device's 4096-site block sums in that order, and
    inside a block every partial sum of these float32 values is exact in float64 -- so for a matrix that starts at
    site 0 the float64 sums are NumPy's to the last bit, and every stored float32 equals the oracle's."""
    labels = np.arange(n) % K
    L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=m)
    pops, af, _, _ = oracle.fit_reference_af(L, IDs, t=8)
    want = oracle.assignLL(L, af.copy(), 8)
    for group_of, n_groups in ((None, 1), (np.searchsorted(pops, IDs[:, 1]).astype(np.int32), K)):
        b = wg.device.DeviceBeagle.from_host(L, group_of, n_groups)
        afs = wg.device.AFSet.from_host(af)
        out, _ = wg.device.assign(b, afs)
        afs.close()
        b.close()
        assert same(out.astype(np.float32), want), (m, n_groups)
        for i, k in ((0, 0), (n - 1, K - 1), (n // 2, 1)):
            vec = np.zeros(m, dtype=np.float32)
            oracle.loglike(L, af, vec, 4, i, k)
            assert out[i, k] == np.sum(vec, dtype=float), (m, i, k)


def test_sum_order_matters_at_millions_of_sites_and_is_numpys(wg, oracle):
    """With frequencies down to 3e-4 (per-site values from 2^-11 to 2^4) and 3M sites the running float64 total no longer
    holds every partial sum exactly (24 + 15 + 22 bits; inside a 4096-site block 24 + 15 + 12 still fit), so the ORDER
    of the additions across blocks shows in the last bits: plain block-after-block accumulation gives other float64 values
    than np.sum(vec, dtype=float) -- the chunk order of block_prefix_kernel gives NumPy's, for every pair."""
    m, n, K = 3_000_001, 6, 2
    labels = np.arange(n) % K
    L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=99)
    rng = np.random.default_rng(5)
    af = rng.choice(np.array([3e-4, 2e-3, 0.05, 0.4, 0.9, 1 - 3e-4], dtype=np.float32), size=(m, K))
    b = wg.device.DeviceBeagle.from_host(L)
    afs = wg.device.AFSet.from_host(af)
    out, _ = wg.device.assign(b, afs)
    afs.close()
    b.close()
    plain_differs = 0
    for i in range(n):
        for k in range(K):
            vec = np.zeros(m, dtype=np.float32)
            oracle.loglike(L, af, vec, 8, i, k)
            assert out[i, k] == np.sum(vec, dtype=float), (i, k)
            v64 = vec.astype(np.float64)
            blocks = np.add.reduceat(v64, np.arange(0, m, 4096))          # exact inside a block
            plain = 0.0
            for s in blocks:
                plain = plain + s
            plain_differs += plain != out[i, k]
    assert plain_differs > 0          # i.e. this test would notice a different order
