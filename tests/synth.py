"""Seeded synthetic Beagle genotype-likelihood matrices for tests and golden vectors.

Follows the recipe in SURVEY.md section 8(d): per-SNP ancestral frequency, per-population
drift, Hardy-Weinberg genotypes, Poisson(2) depth, sequencing error e = 0.01, GLs
normalised, rounded to 6 decimals (what ANGSD writes as text) and cast to float32, stored
as (g0, g1) pairs per (SNP, individual) exactly like reader_cy.pyx:71-77 lays them out.

This is test infrastructure (NumPy); the at-scale generator used by bench.py runs on the
device (wgsassign_amd/csrc/synth.hip).
"""
import hashlib

import numpy as np

SEED = 20260313


def pop_labels(n, K):
    """Equal contiguous blocks of individuals per population: IDs array (n, 2) of str."""
    per = n // K
    labels = []
    for i in range(n):
        k = min(i // per, K - 1) if per > 0 else 0
        labels.append(("Ind%d" % i, "pop%02d" % k))
    return np.array(labels, dtype=str)


def make_beagle(m, n, K, seed=SEED, depth=2.0, interleave=False):
    """Return (L float32 (m, 2n) C-contiguous, IDs (n, 2) str).

    interleave=True assigns populations round-robin instead of in contiguous blocks, so
    per-population column gathers are strided (exercises the slab permutation).
    """
    rng = np.random.Generator(np.random.PCG64(seed + 7919 * m + 104729 * n + K))
    p_anc = rng.beta(0.8, 0.8, size=m)
    p_pop = np.clip(p_anc[:, None] + rng.normal(0.0, 0.08, size=(m, K)), 0.01, 0.99)
    IDs = pop_labels(n, K)
    if interleave:
        IDs[:, 1] = np.array(["pop%02d" % (i % K) for i in range(n)])
    pops = np.unique(IDs[:, 1])
    pop_of = np.searchsorted(pops, IDs[:, 1])
    p_ind = p_pop[:, pop_of]                                   # (m, n)
    geno = rng.binomial(2, p_ind)                              # (m, n)
    d = rng.poisson(depth, size=(m, n))
    e = 0.01
    p_alt = np.array([e, 0.5, 1.0 - e])[geno]
    alt = rng.binomial(d, p_alt)
    ref = d - alt
    l0 = (1 - e) ** ref * e ** alt
    l1 = 0.5 ** d
    l2 = (1 - e) ** alt * e ** ref
    tot = l0 + l1 + l2
    g0 = np.round(l0 / tot, 6)
    g1 = np.round(l1 / tot, 6)
    L = np.empty((m, 2 * n), dtype=np.float32)
    L[:, 0::2] = g0.astype(np.float32)
    L[:, 1::2] = g1.astype(np.float32)
    return np.ascontiguousarray(L), IDs


def make_beagle_for_labels(m, labels, K, seed=SEED, depth=2.0):
    """Like make_beagle, for an arbitrary assignment of individuals to K DIFFERENTIATED populations (labels[i] in
    0..K-1, any sizes, any order): a scoring kernel that mixes up frequency columns gives visibly different sums."""
    labels = np.asarray(labels)
    n = len(labels)
    rng = np.random.Generator(np.random.PCG64(seed + 7919 * m + 104729 * n + K))
    p_anc = rng.beta(0.8, 0.8, size=m)
    p_pop = np.clip(p_anc[:, None] + rng.normal(0.0, 0.12, size=(m, K)), 0.01, 0.99)
    geno = rng.binomial(2, p_pop[:, labels])
    d = rng.poisson(depth, size=(m, n))
    e = 0.01
    alt = rng.binomial(d, np.array([e, 0.5, 1.0 - e])[geno])
    ref = d - alt
    l0, l1, l2 = (1 - e) ** ref * e ** alt, 0.5 ** d, (1 - e) ** alt * e ** ref
    tot = l0 + l1 + l2
    L = np.empty((m, 2 * n), dtype=np.float32)
    L[:, 0::2] = np.round(l0 / tot, 6).astype(np.float32)
    L[:, 1::2] = np.round(l1 / tot, 6).astype(np.float32)
    IDs = np.array([["Ind%d" % i, "pop%02d" % labels[i]] for i in range(n)], dtype=str)
    return np.ascontiguousarray(L), IDs


def digest(a):
    """sha256[:16] of the array bytes (same convention as BASELINE.md section 2)."""
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
