"""Block-parallel exact partition chains (csrc/assign_kernels.hip: chain_cand_kernel / chain_walk_kernel)
against the literal one-lane-per-chain kernel and the oracle's np.add.at (utils.py:147-149): bit-identical
float32 sums for every (individual, population, partition), for awkward partition counts, SNP shards
joined by float32 carries, global partition labels (site0 != 0), -inf / NaN values and repeated addends
(the float32 chain's systematic rounding, which an order-free sum would not reproduce)."""
import numpy as np
import pytest

import synth
from test_gpu_parity import same, same_nan

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from wgsassign_amd import device
    device.get_context()
    return device


def fitted(oracle, L, IDs):
    pops, af, _, _ = oracle.fit_reference_af(L, IDs, t=4)
    return np.searchsorted(pops, IDs[:, 1]).astype(np.int32), af


def oracle_parts(oracle, L, A, P, site0=0, carry=None):
    m, n, K = L.shape[0], L.shape[1] // 2, A.shape[1]
    labels = (site0 + np.arange(m)) % P
    out = np.zeros((n * P, K), dtype=np.float32) if carry is None else carry.copy()
    with np.errstate(all="ignore"):
        for i in range(n):
            for k in range(K):
                vec = np.zeros(m, dtype=np.float32)
                oracle.loglike(L, A, vec, 4, i, k)
                pr = out[i * P:(i + 1) * P, k].copy()
                np.add.at(pr, labels, vec)
                out[i * P:(i + 1) * P, k] = pr
    return out


@pytest.mark.parametrize("P", [1, 2, 3, 7, 10, 64, 65, 100])       # beyond 64 partitions: the one-lane chains
def test_chains_match_literal_kernel_and_oracle(dev, oracle, P):
    m, n, K = 40_000 + 37, 14, 3
    L, IDs = synth.make_beagle(m, n, K, seed=70 + P)
    group_of, af = fitted(oracle, L, IDs)
    site0 = 12_345_678
    b = dev.DeviceBeagle.from_host(L, group_of, K, site0=site0)
    afs = dev.AFSet.from_host(af)
    fast = dev.partition_sums_exact(b, afs, P=P)
    lit = dev.partition_sums_exact(b, afs, P=P, literal=True)
    assert same(fast, lit)
    assert same(fast, oracle_parts(oracle, L, af, P, site0=site0))
    # and continued from a carry (as the second of two SNP shards would)
    rng = np.random.default_rng(P)
    carry = -(rng.random((n * P, K)) * 3e4).astype(np.float32)
    lib = __import__("wgsassign_amd._lib", fromlist=["x"])
    got = np.zeros_like(carry)
    lib.check(lib.load().wgs_assign_parts_exact(b.handle, afs.handle, None, P, lib.f32p(carry), lib.f32p(got)))
    want = np.zeros_like(carry)
    lib.check(lib.load().wgs_debug_parts_exact_literal(b.handle, afs.handle, None, P, lib.f32p(carry), lib.f32p(want)))
    assert same(got, want)
    afs.close()
    b.close()


def test_chains_over_two_shards_equal_one(dev, oracle):
    """The float32 carry handed from the first SNP shard to the second reproduces the unsplit chain."""
    m, n, K, P = 300_000, 10, 2, 3
    L, IDs = synth.make_beagle(m, n, K, seed=99)
    group_of, af = fitted(oracle, L, IDs)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    afs = dev.AFSet.from_host(af)
    sc = dev.Score(b, afs)
    whole = sc.parts_exact(P)
    redone, walked = sc.serial_blocks()
    assert 0 < redone < 0.5 * walked, (redone, walked)          # most blocks take the parallel path (6 % at 2M SNPs)
    sc.close()
    assert same(whole, dev.partition_sums_exact(b, afs, P=P, literal=True))
    cut = 123_457
    lib = __import__("wgsassign_amd._lib", fromlist=["x"])
    carry = None
    for lo, hi in ((0, cut), (cut, m)):
        bs = dev.DeviceBeagle.from_host(np.ascontiguousarray(L[lo:hi]), group_of, K, site0=lo)
        a_s = dev.AFSet.from_host(np.ascontiguousarray(af[lo:hi]))
        out = np.zeros((n * P, K), dtype=np.float32)
        lib.check(lib.load().wgs_assign_parts_exact(bs.handle, a_s.handle, None, P, lib.f32p(carry) if carry is not None else None,
                                                    lib.f32p(out)))
        carry = out
        a_s.close()
        bs.close()
    assert same(carry, whole)
    afs.close()
    b.close()


def test_chains_special_values_and_repeated_addends(dev, oracle):
    """(a) a likelihood of exactly 0 (g = (1, 0), a = 1) makes the per-site value -inf, a NaN frequency makes
    it NaN: the chains carry them exactly as the serial loop does.  (b) half of the sites missing data
    (0.333333, 0.333333): thousands of IDENTICAL addends, each rounding the same way on the running sum's
    grid -- the float32 chain drifts from the float64 sum by far more than random rounding would."""
    m, n, K, P = 50_000, 8, 3, 2
    L, IDs = synth.make_beagle(m, n, K, seed=5)
    L = L.copy()
    L[::2, :] = np.float32(0.333333)                         # (b)
    group_of, af = fitted(oracle, L, IDs)
    af = af.copy()
    af[20_000, 0] = 1.0                                      # (a) -inf for individuals with g = (1, 0) there
    L[20_000, 0:2] = (1.0, 0.0)
    af[30_001, 1] = np.nan                                   # NaN from site 30001 on, population 1
    af[:, 2] = np.float32(0.25)                              # constant frequency: per-site values repeat exactly
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    afs = dev.AFSet.from_host(af)
    sc = dev.Score(b, afs)
    with np.errstate(all="ignore"):
        sums = sc.sums()
        parts = sc.parts_exact(P)
        want = oracle_parts(oracle, L, af, P)
    sc.close()
    assert same_nan(parts, want)
    assert np.isneginf(parts[0 * P + 0, 0]) and np.isnan(parts[:, 1]).any() and np.isfinite(parts[:, 2]).all()
    tot = parts.reshape(n, P, K).astype(np.float64).sum(axis=1)
    drift = np.abs(tot[:, 2] - sums[:, 2]) / np.abs(sums[:, 2])
    assert drift.max() > 1e-5                                # the systematic float32 drift is really there ...
    assert same(parts[:, 2], want[:, 2])                     # ... and reproduced to the bit
    afs.close()
    b.close()


def test_row_range_scores_only_its_individuals(dev, oracle):
    """A leave-one-out batch scores its own individuals only (rows outside the range stay 0), with the same
    bits as the full run."""
    m, n, K, P = 9_000, 23, 4, 3
    L, IDs = synth.make_beagle(m, n, K, seed=8, interleave=True)
    group_of, af = fitted(oracle, L, IDs)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    afs = dev.AFSet.from_host(af)
    full = dev.Score(b, afs)
    s_full, p_full = full.sums(), full.parts_exact(P)
    full.close()
    for lo, hi in ((0, 1), (5, 17), (22, 23), (7, 7)):
        sc = dev.Score(b, afs, rows=(lo, hi))
        s, p = sc.sums(), sc.parts_exact(P)
        sc.close()
        assert same(s[lo:hi], s_full[lo:hi]) and same(p[lo * P:hi * P], p_full[lo * P:hi * P])
        assert not s[:lo].any() and not s[hi:].any() and not p[:lo * P].any() and not p[hi * P:].any()
    afs.close()
    b.close()


def test_sums_are_reproducible_and_mode_fast_close(dev, oracle):
    from wgsassign_amd._lib import MODE_FAST
    m, n, K = 70_000, 31, 10
    L, IDs = synth.make_beagle(m, n, K, seed=77)
    group_of, af = fitted(oracle, L, IDs)
    b = dev.DeviceBeagle.from_host(L, group_of, K)
    afs = dev.AFSet.from_host(af)
    a1, _ = dev.assign(b, afs)
    a2, _ = dev.assign(b, afs)
    assert same(a1, a2)                                       # one writer per block sum, blocks added in order
    f1, _ = dev.assign(b, afs, mode=MODE_FAST)
    assert np.max(np.abs(f1 - a1) / np.abs(a1)) < 1e-6
    afs.close()
    b.close()


def test_random_shapes_against_the_literal_kernel():
    """tools/stress_chains.py, 16 random cases (shapes up to 10^6 SNPs, K <= 20, P <= 64, site0 beyond 2^33, carries,
    per-individual columns, injected -inf / NaN / constant columns / missing-data runs): 0 mismatches."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_chains.py"), "16", "2026"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "16 cases, 0 mismatches" in r.stdout
