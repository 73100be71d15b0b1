"""BASELINE.json configurations at FULL size, held to the oracle over the WHOLE matrix (round-4 review: every full-size test
compared one window of a few hundred rows; a wrong tile anywhere else would have passed).

Every SNP is independent (emMAF_cy.pyx:16-23, glassy_cy.pyx:17-21), so the oracle can follow a device-resident matrix chunk by
chunk through download_rows:
  * the fit (--get_reference_af): EVERY SNP of EVERY population slab -- the oracle's emMAF_update iterated as many times as the
    device reports, on all rows, equals the device's clamped frequencies bit for bit; and the reported iteration IS the
    reference's: its serial float32 convergence sum (emMAF_cy.pyx:30-31) is continued from chunk to chunk for every iteration, and
    `diff < tole` (emMAF.py:23) must hold at the reported iteration and at none before it;
  * the n x K sums (--get_pop_like): the device keeps the sums per chunk of 8192 sites (the addends of np.sum's running float64
    total, glassy.py:38); their fold in NumPy's order must be the returned totals, and at a sample of the chunks -- the first, the
    last, every 80th (1.25 % of the tiles of every slab) and those that straddle a multiple of 2^32 bytes, floats or float4
    elements of a slab -- every one of the n x K chunk sums must equal np.sum(dtype=float) of the oracle's per-site values, bit
    for bit.  configs[1] and configs[3] (1M x 200, 2M x 500) are small enough for EVERY chunk, i.e. the oracle's own n x K matrix.
  * and the class-coded sweep must return the float32 sweep's chunk sums in EVERY chunk (two independent kernels on all tiles).
Each configuration runs twice: with the class codes forced on (what tests/conftest.py does for the whole suite) and with the
cost models' own decisions (what a user gets).  The oracle's part is computed once per configuration.
"""
import os

import numpy as np
import pytest

import synth
from test_gpu_parity import quiet, same

pytestmark = pytest.mark.gpu

CHUNK = 8192
_oracle_cache = {}


@pytest.fixture(scope="module")
def wg():
    from wgsassign_amd import device, emMAF
    from wgsassign_amd.comm import usable_cpus
    device.get_context()

    class NS:
        pass
    ns = NS()
    ns.device, ns.emMAF, ns.threads = device, emMAF, usable_cpus()
    return ns


@pytest.fixture(params=["codes_forced", "cost_model_defaults"])
def codes_policy(request, monkeypatch):
    if request.param == "cost_model_defaults":
        for k in ("WGSASSIGN_EM_CODES_SWEEPS", "WGSASSIGN_SCORE_CODES_ALWAYS", "WGSASSIGN_CODES_ALLOC_WAIT_MS"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("WGSASSIGN_CODES_ALLOC_WAIT_MS", "-1")      # (a slow hipMalloc on a used box must not change which kernels the test exercises)
    return request.param


def blocks_of(n, K):
    return np.minimum(np.arange(n) // (n // K), K - 1).astype(np.int32)


def ids_of(group_of):
    return np.array([["Ind%d" % i, "pop%03d" % g] for i, g in enumerate(group_of)], dtype=str)


def serial_f32(carry, values):
    """The reference's `res = res + x` over float32 values in order, continued from `carry` (np.cumsum accumulates serially in
    the array's own type: tests/test_oracle_golden.py holds it to the oracle's rmse1d)."""
    return np.cumsum(np.concatenate((np.array([carry], dtype=np.float32), values)), dtype=np.float32)[-1]


def oracle_fit_every_snp(oracle, b, group_of, K, iters, threads, rows_per_pass=1 << 17):
    """(af (m, K) clamped, diff[k][t]): the oracle on ALL rows of the device's matrix, iters[k] updates of population k, and the
    reference's convergence metric after every one of them (serial float32 sum over all m SNPs, emMAF_cy.pyx:26-33)."""
    m = b.m
    members = [np.flatnonzero(group_of == k) for k in range(K)]
    T = int(max(iters))
    carry = np.zeros((K, T + 1), dtype=np.float32)
    af = np.empty((m, K), dtype=np.float32)
    for r0 in range(0, m, rows_per_pass):
        nr = min(rows_per_pass, m - r0)
        rows = b.download_rows(r0, nr)
        for k in range(K):
            Lp = oracle.gather(rows, members[k], threads)
            f = np.full(nr, 0.25, dtype=np.float32)
            prev = f.copy()
            for t in range(1, int(iters[k]) + 1):
                oracle.emMAF_update(Lp, f, threads)
                d = f - prev
                carry[k, t] = serial_f32(carry[k, t], d * d)
                prev[:] = f
            af[r0:r0 + nr, k] = oracle.clamp(f, len(members[k]))
    with np.errstate(invalid="ignore"):
        diff = np.sqrt((carry / np.float32(m)).astype(np.float64))
    return af, diff


def check_stopping_iterations(diff, iters, tole=1e-4):
    for k, it in enumerate(iters):
        assert it > 0
        assert diff[k, it] < tole, (k, it, diff[k, :it + 1])                    # emMAF.py:23 fires here ...
        assert not np.any(diff[k, 1:it] < tole), (k, it, diff[k, :it + 1])      # ... and nowhere before


def boundary_chunks(b_npairs, ntiles):
    """Chunks (128 tiles) holding a tile that straddles a multiple of 2^32 bytes / floats / float4 elements of a slab."""
    out = set()
    bytes_per_tile = b_npairs * 64 * 16
    for unit in (1, 4, 16):
        k = 1
        while k * (1 << 32) * unit < ntiles * bytes_per_tile:
            t = (k * (1 << 32) * unit) // bytes_per_tile
            for tt in (t - 1, t, t + 1):
                if 0 <= tt < ntiles:
                    out.add(tt // 128)
            k += 1
    return out


def sample_chunks(nchunks, npairs_list, ntiles, every):
    s = {0, nchunks - 1} | set(range(0, nchunks, every))
    for npairs in npairs_list:
        s |= boundary_chunks(npairs, ntiles)
    return sorted(c for c in s if 0 <= c < nchunks)


def oracle_chunk_sums(oracle, b, A, chunks, threads):
    """want[j, i, k] = np.sum(dtype=float) of the oracle's per-site values of (individual i, population k) over chunk chunks[j]."""
    m, n, K = b.m, b.n, A.shape[1]
    spans = [(c * CHUNK, min(CHUNK, m - c * CHUNK)) for c in chunks]
    rows = np.concatenate([b.download_rows(r0, nr) for r0, nr in spans])
    Asub = np.ascontiguousarray(np.concatenate([A[r0:r0 + nr] for r0, nr in spans]))
    offs = np.concatenate(([0], np.cumsum([nr for _, nr in spans])))
    want = np.empty((len(chunks), n, K), dtype=np.float64)
    with np.errstate(all="ignore"):
        for i in range(n):
            for k in range(K):
                vec = np.zeros(rows.shape[0], dtype=np.float32)
                oracle.loglike(rows, Asub, vec, threads, i, k)
                for j in range(len(chunks)):
                    want[j, i, k] = np.sum(vec[offs[j]:offs[j + 1]], dtype=float)
    return want


def score_and_check(dev, oracle, b, afs, A, key, every, threads, expect_coded=None):
    """The sweep the policy selects and the float32 sweep: chunk sums equal in EVERY chunk, their NumPy-order fold = the returned
    totals, and the sampled chunks equal the oracle's (computed once per configuration)."""
    sc = dev.Score(b, afs)
    tot = sc.sums()
    chunks = sc.chunk_sums()
    coded = b.codes_state() == 1
    if expect_coded is not None:
        assert coded == expect_coded
    run = np.zeros_like(tot)
    for c in range(chunks.shape[0]):
        run = run + chunks[c]                                  # np.sum's running total: chunk after chunk
    assert same(run, tot)
    old = os.environ.get("WGSASSIGN_CODES")
    os.environ["WGSASSIGN_CODES"] = "0"
    try:
        tot32 = sc.sums()
        chunks32 = sc.chunk_sums()
    finally:
        if old is None:
            os.environ.pop("WGSASSIGN_CODES")
        else:
            os.environ["WGSASSIGN_CODES"] = old
    assert same(chunks, chunks32) and same(tot, tot32)         # every tile of every slab, through two kernels
    sc.close()
    ntiles = (b.m + 63) // 64
    npairs = sorted({(int(np.sum(b.group_of == g)) + 1) // 2 for g in range(b.n_groups)})
    pick = sample_chunks(chunks.shape[0], npairs, ntiles, every)
    assert len(pick) * 128 >= 0.01 * ntiles
    if key not in _oracle_cache:
        _oracle_cache[key] = oracle_chunk_sums(oracle, b, A, pick, threads)
    want = _oracle_cache[key]
    bad = np.argwhere(chunks[pick] != want)
    assert bad.size == 0, "chunk %d, individual %d, population %d: %r != %r" % (
        pick[bad[0][0]], bad[0][1], bad[0][2], chunks[pick][tuple(bad[0])], want[tuple(bad[0])])
    return tot, coded


def fit_and_check(wg, oracle, b, group_of, K, key):
    """--get_reference_af on the device against the oracle on every SNP of every slab, stopping iterations included."""
    dev = wg.device
    (pops, af, iters), _ = quiet(wg.emMAF.emMAF_populations, None, ids_of(group_of), 200, 1e-4, beagle=b)
    iters = [int(x) for x in iters]
    if key not in _oracle_cache:
        _oracle_cache[key] = (iters,) + oracle_fit_every_snp(oracle, b, group_of, K, iters, wg.threads)
    iters_o, af_o, diff = _oracle_cache[key]
    assert iters == iters_o                                   # (both policies report the same iterations)
    check_stopping_iterations(diff, iters)
    if not same(af, af_o):
        bad = np.argwhere(af.view(np.uint32) != af_o.view(np.uint32))
        raise AssertionError("%d of %d frequencies differ from the oracle, first at SNP %d (tile %d), population %d: %r != %r"
                             % (len(bad), af.size, bad[0][0], bad[0][0] // 64, bad[0][1], af[tuple(bad[0])], af_o[tuple(bad[0])]))
    return af, iters


@pytest.mark.parametrize("m,n,K", [(1_000_000, 200, 5), (2_000_000, 500, 8)], ids=["config2_1Mx200_K5", "config4_2Mx500_K8"])
def test_every_snp_and_every_sum_config2_and_config4(wg, oracle, codes_policy, m, n, K):
    """configs[1] and configs[3]: the converged fit of EVERY SNP and EVERY one of the n x K log-likelihood sums against the oracle."""
    dev = wg.device
    group_of = blocks_of(n, K)
    b = dev.DeviceBeagle(m, n, group_of, K)
    b.synth(synth.SEED + m // 1_000_000, 2.0)
    af, iters = fit_and_check(wg, oracle, b, group_of, K, ("fit", m, n, K))
    if codes_policy == "cost_model_defaults" and m == 2_000_000:
        assert b.codes_state() == 1                            # the model builds the codes for 2M x 500 (1M x 200 is a near tie: either way)
    afs = dev.AFSet.from_host(af)
    tot, _ = score_and_check(dev, oracle, b, afs, af, ("sums", m, n, K), 1, wg.threads)        # every chunk: the oracle's whole matrix
    # ... which is glassy.assignLL's own output: the float64 totals rounded to float32 (glassy.py:38-42)
    want = _oracle_cache[("sums", m, n, K)]
    run = np.zeros((n, K))
    for c in range(want.shape[0]):
        run = run + want[c]
    assert same(tot.astype(np.float32), run.astype(np.float32)) and same(tot, run)
    assert np.array_equal(np.argmax(tot, axis=1), group_of)
    afs.close()
    b.close()


def test_every_snp_config3_10M_x_1000_K10(wg, oracle, codes_policy):
    """configs[2] (80 GB): the fit on every SNP of all ten slabs; the sums at 1.25 % of the chunks of every slab + the first, the last
    and the 2^32-byte boundary of a slab, through the class codes and over the float32 slabs in EVERY chunk."""
    dev = wg.device
    m, n, K = 10_000_000, 1000, 10
    group_of = blocks_of(n, K)
    b = dev.DeviceBeagle(m, n, group_of, K)
    b.synth(synth.SEED, 2.0)
    af, iters = fit_and_check(wg, oracle, b, group_of, K, ("fit", m, n, K))
    assert b.codes_state() == 1                                # forced, or built by the cost model inside the cold fit
    afs = dev.AFSet.from_host(af)
    pick = sample_chunks((m + CHUNK - 1) // CHUNK, [50], (m + 63) // 64, 80)
    assert 655 in pick                                         # tile 83 886 of a slab starts 2^32 bytes into it
    tot, coded = score_and_check(dev, oracle, b, afs, af, ("sums", m, n, K), 80, wg.threads, expect_coded=True)
    assert np.array_equal(np.argmax(tot, axis=1), group_of)
    afs.close()
    b.close()


def test_pop_like_shape_of_config3_one_slab(wg, oracle, codes_policy):
    """--get_pop_like as the command line meets configs[2]: all 1000 individuals in ONE slab of 5.0e9 float4 elements.  Chunk sums
    at 1.25 % of the chunks, the first, the last and those around element 2^32 and byte 2^32, 2^33, ... of the slab."""
    dev = wg.device
    m, n, K = 10_000_000, 1000, 10
    one = np.zeros(n, dtype=np.int32)
    b = dev.DeviceBeagle(m, n, one, 1)
    b.synth(synth.SEED, 2.0)
    rng = np.random.default_rng(7)
    A = rng.choice(np.array([0.004, 0.03, 0.11, 0.27, 0.5, 0.81, 0.96], dtype=np.float32), size=(m, K)).astype(np.float32)
    afs = dev.AFSet.from_host(A)
    pick = sample_chunks((m + CHUNK - 1) // CHUNK, [500], (m + 63) // 64, 80)
    assert (1 << 32) // (500 * 64) // 128 in pick              # float4 element 2^32 lies in tile 134 217
    tot, coded = score_and_check(dev, oracle, b, afs, A, ("one_slab", m, n, K), 80, wg.threads, expect_coded=True)
    assert np.all(np.isfinite(tot)) and np.all(tot < 0)
    afs.close()
    b.close()


def test_every_snp_config5_shard_6M25_x_2000_K20(wg, oracle, codes_policy):
    """configs[4]: one rank's 100 GB shard of 50M x 2000, K=20 (site0 != 0): the fit on every SNP of all twenty slabs, the sums at
    1.3 % of the chunks of every slab, first, last, 2^32-byte boundary."""
    dev = wg.device
    m, n, K = 6_250_000, 2000, 20
    from wgsassign_amd.comm import shard_range
    site0 = shard_range(50_000_000, 5, 8)[0]
    assert site0 % 8192 == 0 and site0 > 0
    group_of = blocks_of(n, K)
    b = dev.DeviceBeagle(m, n, group_of, K, site0=site0)
    b.synth(synth.SEED, 2.0)
    af, iters = fit_and_check(wg, oracle, b, group_of, K, ("fit", m, n, K))
    assert b.codes_state() == 1
    afs = dev.AFSet.from_host(af)
    tot, coded = score_and_check(dev, oracle, b, afs, af, ("sums", m, n, K), 75, wg.threads, expect_coded=True)
    assert np.array_equal(np.argmax(tot, axis=1), group_of)
    afs.close()
    b.close()
