"""The block-parallel exact emulation of the reference's serial float32 convergence sum
(emMAF_cy.pyx:26-33) against the oracle's literal loop and the golden values -- bit for bit."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from wgsassign_amd import _lib, device
    return _lib.load(), device.get_context(), _lib


def rmse(dev, v1, v2, serial=0):
    lib, ctx, _lib = dev
    out, nser = ctypes.c_double(), ctypes.c_int()
    _lib.check(lib.wgs_debug_rmse1d(ctx.handle, _lib.f32p(v1), _lib.f32p(v2), v1.shape[0], ctypes.byref(out), serial,
                                    ctypes.byref(nser)))
    return out.value, nser.value


def check(dev, oracle, v1, v2, what):
    want = oracle.rmse1d(v1, v2)
    got, nser = rmse(dev, v1, v2)
    nblocks = (len(v1) + 4095) // 4096
    assert got == want or (np.isnan(got) and np.isnan(want)), (what, got, want)
    return nser, nblocks


def test_golden_sizes_incl_10M(dev, golden, oracle):
    g = golden("rmse.npz")
    for m in (449, 100_000, 1_000_000, 10_000_000):
        rng = np.random.Generator(np.random.PCG64(700 + m))
        v1 = rng.random(m, dtype=np.float32)
        v2 = (v1 + rng.normal(0, 1.2e-4, m).astype(np.float32)).astype(np.float32)
        got, nser = rmse(dev, v1, v2)
        assert got == float(g["m%d" % m]), m
        nblocks = (m + 4095) // 4096
        print("m=%d: %d of %d blocks serial" % (m, nser, nblocks))
        assert nser <= 40 + nblocks // 50          # the parallel path carries almost everything
    # the one-lane serial kernel agrees as well (it is the in-device reference of the fast path)
    assert rmse(dev, v1[:300_000].copy(), v2[:300_000].copy(), serial=1)[0] == oracle.rmse1d(v1[:300_000].copy(), v2[:300_000].copy())


def test_wide_magnitudes_and_ties(dev, oracle):
    rng = np.random.Generator(np.random.PCG64(77))
    m = 300_000
    v1 = rng.random(m, dtype=np.float32)
    v2 = (v1 + (rng.normal(0, 1, m) * 10.0 ** rng.uniform(-7, -1, m)).astype(np.float32)).astype(np.float32)
    check(dev, oracle, v1, v2, "wide")
    # exact ties: differences j * 2^-12 -> squares j^2 * 2^-24 land exactly half way between grid points
    for scale in (2.0 ** -12, 2.0 ** -10, 2.0 ** -13):
        j = rng.integers(0, 64, size=200_003)
        v2 = np.zeros(len(j), dtype=np.float32)
        v1 = (j * scale).astype(np.float32)
        check(dev, oracle, v1, v2, "ties %g" % scale)
    # constant tiny increments: long runs absorbed (d < ulp/2), then exactly ulp/2
    v1 = np.full(1_000_001, 2.0 ** -13, dtype=np.float32)
    v2 = np.zeros_like(v1)
    check(dev, oracle, v1, v2, "constant")
    # a late spike forces binade jumps
    v1 = rng.random(500_000, dtype=np.float32) * np.float32(1e-3)
    v2 = np.zeros_like(v1)
    v1[123_457] = 30.0
    v1[400_000] = 1000.0
    check(dev, oracle, v1, v2, "spikes")


def test_zero_nan_and_small(dev, oracle):
    z = np.zeros(100_000, dtype=np.float32)
    got, nser = rmse(dev, z, z.copy())
    assert got == 0.0 and nser == 0                 # all-zero blocks are skipped, not walked serially
    v1 = np.random.default_rng(1).random(50_000, dtype=np.float32)
    v2 = v1.copy()
    v2[777] += np.float32(0.25)
    check(dev, oracle, v1, v2, "single")
    v2[40_000] = np.nan
    assert np.isnan(rmse(dev, v1, v2)[0])
    for m in (1, 2, 63, 64, 65, 4095, 4096, 4097):
        a = np.random.default_rng(m).random(m, dtype=np.float32)
        b = np.random.default_rng(m + 1).random(m, dtype=np.float32)
        check(dev, oracle, a, b, m)


def test_carry_across_shards(dev, oracle):
    """wgs_em_rmse_chain continues the chain from the previous shard's carry: two shards == one."""
    from wgsassign_amd import device
    rng = np.random.default_rng(5)
    m, split = 700_001, 300_017
    f_prev = rng.random(m, dtype=np.float32)
    f_cur = (f_prev + rng.normal(0, 3e-4, m).astype(np.float32)).astype(np.float32)

    def batch(lo, hi):
        L = np.zeros((hi - lo, 2), dtype=np.float32)
        b = device.DeviceBeagle.from_host(L)
        em = device.EMBatch(b, [0])
        em.set_f(0, np.ascontiguousarray(f_prev[lo:hi]))
        em.step()                                    # flips buffers: previous := what we set
        em.set_f(0, np.ascontiguousarray(f_cur[lo:hi]))
        return b, em
    b0, e0 = batch(0, m)
    whole = e0.rmse_chain(0, 0.0)
    assert device.chain_diff(whole, m) == oracle.rmse1d(f_cur, f_prev)
    b1, e1 = batch(0, split)
    b2, e2 = batch(split, m)
    c1 = e1.rmse_chain(0, 0.0)
    c2 = e2.rmse_chain(0, c1)
    assert np.float32(c2).tobytes() == np.float32(whole).tobytes()
    for x in (e0, e1, e2, b0, b1, b2):
        x.close()
