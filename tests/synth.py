"""Seeded synthetic Beagle genotype-likelihood matrices for tests and golden vectors.

Follows the recipe in SURVEY.md section 8(d): per-SNP ancestral frequency, per-population
drift, Hardy-Weinberg genotypes, Poisson(2) depth, sequencing error e = 0.01, GLs
normalised, rounded to 6 decimals (what ANGSD writes as text) and cast to float32, stored
as (g0, g1) pairs per (SNP, individual) exactly like reader_cy.pyx:71-77 lays them out.

This is test infrastructure (NumPy); the at-scale generator used by bench.py runs on the
device (wgsassign_amd/csrc/synth.hip).
"""
import hashlib
import struct
import zlib

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


# Base-quality bins of a current Illumina instrument (RTA3 bins qualities to four values; share of bases per bin as in a
# typical 2 x 150 run) -- the per-read error rates of make_beagle_quality below.
QUAL_BINS = ((12, 23, 37), (0.03, 0.12, 0.85))


def make_beagle_quality(m, n, K, seed=SEED, depth=2.0, quals=QUAL_BINS, dmax=15, labels=None):
    """Like make_beagle, with QUALITY-DEPENDENT genotype likelihoods -- what ANGSD's -GL 2 (GATK model) writes for real
    reads: every read has its own error rate e = 10^(-Q/10), Q drawn per read (quals = (values, probabilities), or
    (lo, hi) for a uniform integer quality), P(base | genotype) = 1-e / e/3 for the homozygotes and their mean for the
    heterozygote, likelihoods multiplied over the reads, normalised, rounded to 6 decimals.  The fixed e = 0.01 of
    make_beagle gives ~27 distinct (g0, g1) pairs per SNP among 1000 individuals; binned qualities give ~80 (the bundled
    85-individual AMRE file has 29 on average, this model 26 at n = 85), a uniform Q20-40 draw ~530.
    Returns (L float32 (m, 2n), IDs (n, 2) str)."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919 * m + 104729 * n + K + 17))
    p_anc = rng.beta(0.8, 0.8, size=m)
    p_pop = np.clip(p_anc[:, None] + rng.normal(0.0, 0.08, size=(m, K)), 0.01, 0.99)
    if labels is None:
        IDs = pop_labels(n, K)
        pops = np.unique(IDs[:, 1])
        pop_of = np.searchsorted(pops, IDs[:, 1])
    else:
        pop_of = np.asarray(labels)
        IDs = np.array([["Ind%d" % i, "pop%02d" % pop_of[i]] for i in range(n)], dtype=str)
    L = np.empty((m, 2 * n), dtype=np.float32)
    step = max(1, 4_000_000 // max(1, n * dmax))              # rows per slice: the (rows, n, dmax) temporaries stay small
    for r0 in range(0, m, step):
        r1 = min(m, r0 + step)
        geno = rng.binomial(2, p_pop[r0:r1][:, pop_of])
        d = np.minimum(rng.poisson(depth, size=geno.shape), dmax)
        shape = geno.shape + (dmax,)
        if np.isscalar(quals[0]):
            Q = rng.integers(quals[0], quals[1] + 1, size=shape)
        else:
            Q = rng.choice(np.asarray(quals[0]), p=np.asarray(quals[1], dtype=float), size=shape)
        e = 10.0 ** (-Q / 10.0)
        is_alt = (rng.random(shape) < geno[:, :, None] * 0.5) ^ (rng.random(shape) < e)
        live = np.arange(dmax)[None, None, :] < d[:, :, None]
        pm, px = 1.0 - e, e / 3.0
        l0 = np.where(live, np.where(is_alt, px, pm), 1.0).prod(axis=2)
        l1 = np.where(live, 0.5 * pm + 0.5 * px, 1.0).prod(axis=2)
        l2 = np.where(live, np.where(is_alt, pm, px), 1.0).prod(axis=2)
        tot = l0 + l1 + l2
        L[r0:r1, 0::2] = np.round(l0 / tot, 6)
        L[r0:r1, 1::2] = np.round(l1 / tot, 6)
    return np.ascontiguousarray(L), IDs


def classes_per_snp(L):
    """Distinct (g0, g1) bit patterns per row of an (m, 2n) float32 matrix."""
    L = np.ascontiguousarray(L, dtype=np.float32)
    key = (L[:, 0::2].view(np.uint32).astype(np.uint64) << np.uint64(32)) | L[:, 1::2].view(np.uint32)
    key.sort(axis=1)
    return 1 + (key[:, 1:] != key[:, :-1]).sum(axis=1)


def digest(a):
    """sha256[:16] of the array bytes (same convention as BASELINE.md section 2)."""
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# ---- BGZF files (what ANGSD writes) for the reader / ingest tests and tools/bench_reader.py
def bgzf_block(chunk, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    payload = co.compress(chunk) + co.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(payload) + 8 - 1) +
            payload + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


def write_bgzf(path, data, block=60000):
    with open(path, "wb") as fh:
        for i in range(0, len(data), block):
            fh.write(bgzf_block(data[i:i + block]))
        fh.write(bgzf_block(b""))


def make_pool_file(path, n, m, pool=1024, seed=1):
    """A BGZF Beagle file of m sites x n individuals without formatting m x 3n numbers in Python: a pool of
    pre-compressed GL sections (one BGZF block each, the way low-depth ANGSD output looks: a third of the genotypes
    missing = 0.333333 three times), each line = a small block with its own site name + one pool block.
    Returns the pool's values (pool, 2n) float32 and the pool index of every line."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1_000_001, size=(pool, n))
    b = (rng.random((pool, n)) * (1_000_000 - a)).astype(np.int64)
    c = 1_000_000 - a - b
    missing = rng.random((pool, n)) < 0.35
    a[missing], b[missing], c[missing] = 333333, 333333, 333333
    v = np.stack([a, b, c], axis=2).reshape(pool, 3 * n)                      # micro-units
    txt = np.empty((pool, 3 * n, 9), dtype=np.uint8)
    txt[:, :, 0] = 9                                                          # '\t'
    txt[:, :, 1] = 48 + v // 1_000_000
    txt[:, :, 2] = 46
    r = v % 1_000_000
    for k in range(6):
        txt[:, :, 3 + k] = 48 + (r // 10 ** (5 - k)) % 10
    vals = (v.reshape(pool, n, 3)[:, :, :2].reshape(pool, 2 * n) / 1e6).astype(np.float32)
    blocks = [bgzf_block(txt[i].tobytes() + b"\n", level=4) for i in range(pool)]
    pick = rng.integers(0, pool, size=m)
    head = "marker\tallele1\tallele2\t" + "\t".join("I%d\tI%d\tI%d" % (i, i, i) for i in range(n)) + "\n"
    with open(path, "wb") as fh:
        fh.write(bgzf_block(head.encode()))
        for s in range(m):
            fh.write(bgzf_block(b"chr7_%d\tA\tC" % (s + 1), level=1))
            fh.write(blocks[pick[s]])
        fh.write(bgzf_block(b""))
    return vals, pick
