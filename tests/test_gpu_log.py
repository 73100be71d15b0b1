"""The assignment kernel's double-precision log of float32 arguments (csrc/assign_kernels.hip:
log_f32arg) against libm.  The reference stores (float)log((double)s) per site
(glassy_cy.pyx:21); the device function must give the same float32 for every float32 s, up to
the handful of arguments whose true log lies within a double ulp of a float32 rounding boundary."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from wgsassign_amd import _lib, device
    ctx = device.get_context()
    return _lib.load(), ctx, _lib


def values(dev, x, use_libm=0):
    lib, ctx, _lib = dev
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    _lib.check(lib.wgs_debug_log_values(ctx.handle, _lib.f32p(x), _lib.f32p(out), x.shape[0], use_libm))
    return out


def test_exhaustive_vs_device_libm(dev):
    """Every positive finite float32 (2^31 - 2^23 bit patterns): custom vs the device math library."""
    lib, ctx, _lib = dev
    count, first = ctypes.c_uint64(), ctypes.c_uint32()
    _lib.check(lib.wgs_debug_log_mismatch(ctx.handle, 1, 0x7F800000, ctypes.byref(count), ctypes.byref(first)))
    print("float32-rounded log mismatches vs ocml over all positive float32:", count.value, "first bits", hex(first.value))
    assert count.value <= 64, (count.value, hex(first.value))


def test_dense_sample_vs_glibc(dev, oracle):
    """Against the reference's own libm: every 61st bit pattern of the whole positive range, every
    pattern of [0.999, 1.001] (where log passes through zero) and the likelihood-sum range (0, 1]
    at stride 7."""
    total = mism = 0
    worst = []
    for bits in (np.arange(1, 0x7F800000, 61, dtype=np.uint32),
                 np.arange(np.float32(0.999).view(np.uint32), np.float32(1.001).view(np.uint32) + 1, dtype=np.uint32),
                 np.arange(0x2F000000, 0x3F800001, 7, dtype=np.uint32)):
        x = bits.view(np.float32)
        mine = values(dev, x)
        ref = oracle.log_f32(np.ascontiguousarray(x), 8)
        bad = mine.view(np.uint32) != ref.view(np.uint32)
        total += len(x)
        mism += int(bad.sum())
        if bad.any():
            worst.append(int(np.max(np.abs(mine[bad].view(np.int32).astype(np.int64) - ref[bad].view(np.int32)))))
    print("vs glibc: %d of %d float32-rounded values differ" % (mism, total))
    assert mism <= max(8, total // 20_000_000) and all(w <= 1 for w in worst)     # at most 1 float32 ulp, a handful of cases


def test_exhaustive_vs_glibc(dev, oracle):
    """EVERY positive finite float32 (2,139,095,039 arguments): the device function against
    (float)log((double)x) of the host's glibc -- the exact expression the reference evaluates
    (glassy_cy.pyx:21 with a zero-initialised vector)."""
    threads = min(len(__import__("os").sched_getaffinity(0)), 16)
    total = mism = 0
    first = []
    step = 1 << 26
    for b0 in range(0, 0x7F800000, step):
        bits = np.arange(max(b0, 1), min(b0 + step, 0x7F800000), dtype=np.uint32)
        x = bits.view(np.float32)
        mine = values(dev, x)
        ref = oracle.log_f32(x, threads)
        bad = np.flatnonzero(mine.view(np.uint32) != ref.view(np.uint32))
        total += len(x)
        mism += len(bad)
        first += [hex(int(bits[i])) for i in bad[:4]]
    print("exhaustive vs glibc: %d of %d float32-rounded logs differ %s" % (mism, total, first[:8]))
    assert total == 0x7F800000 - 1 and mism <= 16


def test_special_values(dev):
    x = np.array([0.0, -0.0, -1.0, np.inf, np.nan, 1.0, 1e-45, 1.1754944e-38, 3.4028235e38, -np.inf], dtype=np.float32)
    with np.errstate(all="ignore"):
        want = np.log(x.astype(np.float64)).astype(np.float32)
    got = values(dev, x)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok], want[ok])
    assert got[5] == 0.0 and got[0] == -np.inf and got[1] == -np.inf and got[3] == np.inf


def test_em_divide_is_ieee_exact(dev):
    """The EM kernel's Newton-core divide (no v_div_scale) against the compiler's IEEE double divide
    on 4.3e9 pseudo-random EM-shaped operand pairs (incl. 0/0 and 100 binades of magnitude)."""
    lib, ctx, _lib = dev
    bad = ctypes.c_uint64(123)
    _lib.check(lib.wgs_debug_div_mismatch(ctx.handle, 20260313, 4096, ctypes.byref(bad)))
    assert bad.value == 0, bad.value


def test_em_divide_reciprocal_bound_exhaustive(dev):
    """div_exact uses ONE Newton refinement of v_rcp_f64.  Its exactness argument (csrc/em_kernels.hip) needs the
    refined reciprocal within 2^-48 of 1/den: checked here for EVERY float32 mantissa of the denominator (2^23 values),
    at exponents from float32-denormal sums to the largest possible sum."""
    lib, ctx, _lib = dev
    for exponent in (-149, -126, -100, -30, -1, 0, 1, 2):
        worst = ctypes.c_double(-1.0)
        _lib.check(lib.wgs_debug_rcp_error(ctx.handle, exponent, ctypes.byref(worst)))
        assert 0 <= worst.value < 2.0 ** -48, (exponent, worst.value)
