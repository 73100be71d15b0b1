"""Pins the CPU oracle (oracle/) bit-for-bit to the real reference through tests/golden/.

Golden vectors come from tests/golden/make_golden.py run against the compiled reference.
CPU only: no GPU needed.
"""
import os

import numpy as np
import pytest

import synth
from conftest import GOLDEN

DATA = os.path.join(GOLDEN, "data")


def same(a, b):
    """bit-exact incl. NaN pattern"""
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def test_reader_matches_reference(oracle, golden):
    g = golden("amre_fit.npz")
    L, samples, sites = oracle.read_beagle(os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"))
    assert same(L, g["L"]) and synth.digest(L) == "432436039a568fb4"
    assert samples == list(g["samples"]) and sites == list(g["sites"])


def test_amre_fit(oracle, golden):
    g = golden("amre_fit.npz")
    pops, af, f_raw, iters = oracle.fit_reference_af(g["L"], g["IDs"], t=2)
    assert list(pops) == list(g["pops"])
    assert list(iters) == list(g["iters"]) == [17, 14, 16, 14, 13]
    assert same(f_raw, g["f_raw"]) and same(af, g["pop_af"])
    assert synth.digest(af) == "876aa2694aecb6f3"


def test_amre_trace(oracle, golden):
    g = golden("amre_trace.npz")
    L = g["L_pop"]
    f = np.full(L.shape[0], 0.25, dtype=np.float32)
    prev = f.copy()
    for it in range(len(g["trace"])):
        oracle.emMAF_update(L, f, 1)
        assert same(f, g["trace"][it]), it
        assert oracle.rmse1d(f, prev) == g["diffs"][it]
        prev = f.copy()


def test_thread_invariance(oracle, golden):
    L = golden("amre_trace.npz")["L_pop"]
    f1 = np.full(L.shape[0], 0.3, dtype=np.float32)
    f8 = f1.copy()
    oracle.emMAF_update(L, f1, 1)
    oracle.emMAF_update(L, f8, 8)
    assert same(f1, f8)


def test_amre_assign(oracle, golden):
    g = golden("amre_assign.npz")
    af = golden("amre_fit.npz")["pop_af"]
    assert same(oracle.assignLL(g["L"], af.copy(), 2), g["logl"])
    vec = np.zeros(g["L"].shape[0], dtype=np.float32)
    oracle.loglike(g["L"], af.copy(), vec, 1, 3, 2)
    assert same(vec, g["vec_i3_k2"])


@pytest.mark.parametrize("P", [1, 3])
def test_amre_loo(oracle, golden, P):
    g, fit = golden("amre_loo.npz"), golden("amre_fit.npz")
    af = fit["pop_af"].copy()
    ll, parts = oracle.loo(fit["L"], af, fit["IDs"], 2, 200, 1e-4, None, P)
    assert same(ll, g["loo_P%d" % P]) and same(parts, g["parts_P%d" % P]) and same(af, g["af_after_P%d" % P])


def test_amre_loo_downsampled(oracle, golden):
    g, fit = golden("amre_loo.npz"), golden("amre_fit.npz")
    mask = oracle.filter_sites_mask(list(fit["sites"]), list(g["sites_ds"]))
    assert mask.dtype == np.bool_ and same(mask, g["mask"]) and int((~mask).sum()) == 92
    L_f = np.ascontiguousarray(fit["L"][mask])
    mask2 = oracle.filter_sites_mask(list(g["sites_ds"]), list(fit["sites"][mask]))
    L_ds = np.ascontiguousarray(g["L_ds"][mask2])
    pops, af, f_raw, iters = oracle.fit_reference_af(L_f, fit["IDs"])
    assert same(af, g["pop_af_filtered"]) and list(iters) == list(g["iters_filtered"])
    ll, _ = oracle.loo(L_f, af, fit["IDs"], 2, 200, 1e-4, L_ds, 1)
    assert same(ll, g["loo_ds"]) and same(af, g["af_after_ds"])


def test_edge_cases(oracle, golden):
    g = golden("edge.npz")
    for n in (1, 2, 5):
        L = g["corner_n%d_L" % n]
        f = np.full(L.shape[0], 0.25, dtype=np.float32)
        for it in range(4):
            oracle.emMAF_update(L, f, 1)
            assert same(f, g["corner_n%d_f" % n][it]), (n, it)
    L = g["mixed_L"]
    f, it = oracle.emMAF(L, 200, 1e-4, 1)
    assert same(f, g["mixed_f"]) and (it or -1) == int(g["mixed_iters"][0])
    for name, f0 in (("zero", 0.0), ("one", 1.0)):
        f = np.full(12, f0, dtype=np.float32)
        oracle.emMAF_update(L, f, 1)
        assert same(f, g["mixed_from_" + name])
    f, it = oracle.emMAF(L, 3, 1e-12, 1)
    assert same(f, g["exhaust_f"]) and it == 0 and int(g["exhaust_printed"]) == 0
    v1 = np.array([0.25, 0.5, np.nan], dtype=np.float32)
    v2 = np.array([0.5, 0.5, 0.1], dtype=np.float32)
    assert np.isnan(oracle.rmse1d(v1, v2)) and np.isnan(g["rmse_nan"])
    assert oracle.rmse1d(v1[:2].copy(), v2[:2].copy()) == float(g["rmse_small"])
    A = g["ll_A"]
    n = L.shape[1] // 2
    with np.errstate(all="ignore"):
        for i in range(n):
            for k in range(5):
                vec = np.zeros(12, dtype=np.float32)
                oracle.loglike(L, A, vec, 1, i, k)
                assert same(vec, g["ll_vecs"][i, k])
    assert same(oracle.assignLL(L, A, 1), g["ll_mat"])
    v = np.zeros(12, dtype=np.float32)
    oracle.loglike(L, A, v, 1, 2, 1)
    oracle.loglike(L, A, v, 1, 4, 2)
    assert same(v, g["ll_accum"])
    # population of size 1 under LOO: NaN column that sticks
    af = g["single_af"].copy()
    ll, parts = oracle.loo(g["single_L"], af, g["single_IDs"], 1, 20, 1e-4, None, 2)
    assert same(ll, g["single_loo"]) and same(parts, g["single_parts"]) and same(af, g["single_af_after"])
    assert np.isnan(af[:, 1]).all()


@pytest.mark.parametrize("n", [85, 200, 1000, 2000])
def test_accumulation_order(oracle, golden, n):
    g = golden("accum.npz")
    m = int(g["n%d_m" % n])
    L, _ = synth.make_beagle(m, n, 1, seed=600 + n)
    assert synth.digest(L) == str(g["n%d_digest" % n])
    f1 = np.full(m, 0.25, dtype=np.float32)
    oracle.emMAF_update(L, f1, 4)
    assert same(f1, g["n%d_f1" % n])
    f, it = oracle.emMAF(L, 200, 1e-4, 4)
    assert same(f, g["n%d_f" % n]) and it == int(g["n%d_iters" % n][0])


def test_rmse(oracle, golden):
    g = golden("rmse.npz")
    for m in (449, 100_000, 1_000_000, 10_000_000):
        rng = np.random.Generator(np.random.PCG64(700 + m))
        v1 = rng.random(m, dtype=np.float32)
        v2 = (v1 + rng.normal(0, 1.2e-4, m).astype(np.float32)).astype(np.float32)
        assert oracle.rmse1d(v1, v2) == float(g["m%d" % m])
    rng = np.random.Generator(np.random.PCG64(77))
    m = 300_000
    v1 = rng.random(m, dtype=np.float32)
    v2 = (v1 + (rng.normal(0, 1, m) * 10.0 ** rng.uniform(-7, -1, m)).astype(np.float32)).astype(np.float32)
    assert oracle.rmse1d(v1, v2) == float(g["wide"])


def test_synth_mid(oracle, golden):
    g = golden("synth_mid.npz")
    L, IDs = synth.make_beagle(int(g["m"]), int(g["n"]), int(g["K"]))
    assert synth.digest(L) == str(g["digest"])
    pops, af, f_raw, iters = oracle.fit_reference_af(L, IDs, t=8)
    assert same(af, g["pop_af"]) and list(iters) == list(g["iters"])
    assert same(oracle.assignLL(np.ascontiguousarray(L[:5000]), np.ascontiguousarray(af[:5000]), 8), g["logl_5000"])
    ms = int(g["loo_ms"])
    Ls = np.ascontiguousarray(L[:ms])
    pops, af2, _, it2 = oracle.fit_reference_af(Ls, IDs, t=8)
    assert same(af2, g["loo_pop_af"]) and list(it2) == list(g["loo_iters"])
    ll, parts = oracle.loo(Ls, af2, IDs, 8, 200, 1e-4, None, 4)
    assert same(ll, g["loo"]) and same(parts, g["loo_parts"]) and same(af2, g["loo_af_after"])


def test_synth_interleaved(oracle, golden):
    g = golden("synth_interleaved.npz")
    L, IDs = synth.make_beagle(6000, 37, 4, seed=88, interleave=True)
    assert synth.digest(L) == str(g["digest"])
    pops, af, _, iters = oracle.fit_reference_af(L, IDs, t=8)
    assert same(af, g["pop_af"]) and list(iters) == list(g["iters"])
    ll, _ = oracle.loo(L, af, IDs, 8, 200, 1e-4, None, 1)
    assert same(ll, g["loo"]) and same(af, g["af_after"])


def test_fisher(oracle, golden):
    """--ne_obs: fisher.py / fisher_cy.pyx restated, bit-exact incl. np.mean."""
    g, fit = golden("fisher.npz"), golden("amre_fit.npz")
    f_obs, ne_obs = oracle.fisher_obs(fit["L"], fit["pop_af"].copy(), fit["IDs"], 2)
    assert same(f_obs, g["f_obs"]) and same(ne_obs, g["ne_obs"])
    assert same(oracle.fisher_obs_ind(fit["L"], fit["pop_af"].copy(), fit["IDs"], 2), g["ne_ind"])
    L, IDs = synth.make_beagle(5000, 61, 3, seed=31, interleave=True)
    assert synth.digest(L) == str(g["synth_digest"])
    f_obs, ne_obs = oracle.fisher_obs(L, g["synth_af"].copy(), IDs, 4)
    assert same(f_obs, g["synth_f_obs"]) and same(ne_obs, g["synth_ne_obs"])
    assert same(oracle.fisher_obs_ind(L, g["synth_af"].copy(), IDs, 4), g["synth_ne_ind"])
