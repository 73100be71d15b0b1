"""Degenerate shapes through the reference-shaped entry points, against the oracle: no SNPs at all, a single SNP,
SNP counts around the 64-SNP tile and the 4096-SNP block, one individual, one population, a single-column
likelihood matrix.  What the reference does there follows from its loops (emMAF_cy.pyx:16-23 runs zero times,
rmse1d divides 0 by 0 -> NaN -> never converges, np.sum of an empty vector is 0.0): the device path must agree and
must not fault on an empty launch."""
import numpy as np
import pytest

import synth
from test_gpu_parity import quiet, same, same_nan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wg():
    from wgsassign_amd import device, emMAF, emMAF_cy, glassy, glassy_cy
    device.get_context()

    class NS:
        pass
    ns = NS()
    ns.device, ns.emMAF, ns.emMAF_cy, ns.glassy, ns.glassy_cy = device, emMAF, emMAF_cy, glassy, glassy_cy
    return ns


def test_no_snps_at_all(wg, oracle):
    n, K = 6, 2
    L = np.empty((0, 2 * n), dtype=np.float32)
    f, text = quiet(wg.emMAF.emMAF, L, 7, 1e-4, 1)
    f_o, iters_o = oracle.emMAF(L, 7, 1e-4, 4)
    assert f.shape == f_o.shape == (0,) and f.dtype == np.float32 and text == "" and iters_o == 0   # NaN diff: never converges
    af = np.empty((0, K), dtype=np.float32)
    ll, text = quiet(wg.glassy.assignLL, L, af, 1)
    assert ll.shape == (n, K) and ll.dtype == np.float32 and not ll.any()                          # np.sum([]) = 0.0
    assert same(ll, oracle.assignLL(L, af, 4))
    assert text.strip() == "%d individuals to assign to %d populations" % (n, K)
    IDs = np.array([["i%d" % i, "p%d" % (i % K)] for i in range(n)])
    with np.errstate(all="ignore"):
        (loo, parts), _ = quiet(wg.glassy.loo, L, af.copy(), IDs, 1, 5, 1e-4, None, 3)
        loo_o, parts_o = oracle.loo(L, af.copy(), IDs, 4, 5, 1e-4, None, 3)
    assert same_nan(loo, loo_o) and same_nan(parts, parts_o) and parts.shape == (n * 3, K)


@pytest.mark.parametrize("m", [1, 2, 63, 64, 65, 4095, 4096, 4097, 8193])
def test_snp_counts_around_tile_and_block_edges(wg, oracle, m):
    n, K = 9, 3
    labels = np.arange(n) % K
    L, IDs = synth.make_beagle_for_labels(m, labels, K, seed=100 + m)
    pops_o, af_o, _, iters_o = oracle.fit_reference_af(L, IDs, t=4)
    (pops, af, iters), _ = quiet(wg.emMAF.emMAF_populations, L, IDs, 200, 1e-4)
    assert list(iters) == list(iters_o) and same_nan(af, af_o), m
    with np.errstate(all="ignore"):
        ll, _ = quiet(wg.glassy.assignLL, L, af.copy(), 1)
        assert same_nan(ll, oracle.assignLL(L, af_o.copy(), 4)), m
        a1, a2 = af_o.copy(), af_o.copy()
        loo_o, parts_o = oracle.loo(L, a1, IDs, 4, 200, 1e-4, None, 2)
        (loo, parts), _ = quiet(wg.glassy.loo, L, a2, IDs, 1, 200, 1e-4, None, 2)
    assert same_nan(loo, loo_o) and same_nan(parts, parts_o) and same_nan(a1, a2), m


def test_one_individual_one_population(wg, oracle):
    m = 777
    L, IDs = synth.make_beagle_for_labels(m, np.zeros(1, dtype=int), 1, seed=5)
    assert L.shape == (m, 2)
    f, _ = quiet(wg.emMAF.emMAF, L, 200, 1e-4, 1)
    f_o, _ = oracle.emMAF(L, 200, 1e-4, 4)
    assert same_nan(f, f_o)
    af = np.clip(np.nan_to_num(f_o, nan=0.5), 0.25, 0.75).reshape(m, 1).astype(np.float32)
    ll, _ = quiet(wg.glassy.assignLL, L, af.copy(), 1)
    assert ll.shape == (1, 1) and same_nan(ll, oracle.assignLL(L, af.copy(), 4))
    # leave-one-out of the only member: an empty re-fit -> NaN column, NaN likelihoods (glassy.py:69-89)
    with np.errstate(all="ignore"):
        a1, a2 = af.copy(), af.copy()
        loo_o, parts_o = oracle.loo(L, a1, IDs, 4, 5, 1e-4, None, 1)
        (loo, parts), _ = quiet(wg.glassy.loo, L, a2, IDs, 1, 5, 1e-4, None, 1)
    assert same_nan(loo, loo_o) and same_nan(parts, parts_o) and same_nan(a1, a2) and np.isnan(a2).all()


def test_thin_mirrors_on_empty_and_single_inputs(wg, oracle):
    # one SNP, one individual through the per-call mirrors (emMAF_cy.pyx:10, glassy_cy.pyx:12)
    L = np.array([[0.2, 0.5]], dtype=np.float32)
    f = np.array([0.25], dtype=np.float32)
    f_o = f.copy()
    wg.emMAF_cy.emMAF_update(L, f, 1)
    oracle.emMAF_update(L, f_o, 1)
    assert same(f, f_o)
    A = np.array([[0.4]], dtype=np.float32)
    v, v_o = np.zeros(1, dtype=np.float32), np.zeros(1, dtype=np.float32)
    wg.glassy_cy.loglike(L, A, v, 1, 0, 0)
    oracle.loglike(L, A, v_o, 1, 0, 0)
    assert same(v, v_o)
    # zero SNPs: nothing to do, nothing written, no fault
    L0 = np.empty((0, 2), dtype=np.float32)
    wg.emMAF_cy.emMAF_update(L0, np.empty(0, dtype=np.float32), 1)
    wg.glassy_cy.loglike(L0, np.empty((0, 1), dtype=np.float32), np.empty(0, dtype=np.float32), 1, 0, 0)
    assert np.isnan(wg.emMAF_cy.rmse1d(np.empty(0, dtype=np.float32), np.empty(0, dtype=np.float32)))   # 0/0 (emMAF_cy.pyx:32)
