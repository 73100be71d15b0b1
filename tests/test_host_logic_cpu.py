"""CPU-only tests of the product's host logic: the EM driver loop (single rank and two gloo
ranks), shard ranges, utils, reader, CLI argument handling.  The arithmetic primitives are
stood in for by the oracle (tests/cpu_standin.py); everything else is product code."""
import io
import contextlib
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import synth


def free_port():
    """A TCP port p the OS reports free right now, with p + 1 free too (the communicators' TCP star listens there)."""
    from wgsassign_amd.comm import free_port_pair
    return free_port_pair()


from conftest import GOLDEN, ROOT
from cpu_standin import OracleEMBatch

DATA = os.path.join(GOLDEN, "data")


def test_cumsum_standin_is_the_serial_chain(oracle):
    rng = np.random.Generator(np.random.PCG64(1))
    a = rng.random(50_000, dtype=np.float32)
    b = (a + rng.normal(0, 1e-4, 50_000).astype(np.float32)).astype(np.float32)
    em = OracleEMBatch(oracle, np.zeros((50_000, 2), np.float32), [np.array([0])])
    em.f[0], em.f_prev[0] = a, b
    from wgsassign_amd.device import chain_diff
    assert chain_diff(em.rmse_chain(0, 0.0), 50_000) == oracle.rmse1d(a, b)
    # split in two shards: carry hand-off reproduces the unsplit chain bit for bit
    em1 = OracleEMBatch(oracle, np.zeros((20_000, 2), np.float32), [np.array([0])])
    em2 = OracleEMBatch(oracle, np.zeros((30_000, 2), np.float32), [np.array([0])])
    em1.f[0], em1.f_prev[0] = a[:20_000], b[:20_000]
    em2.f[0], em2.f_prev[0] = a[20_000:], b[20_000:]
    assert em2.rmse_chain(0, em1.rmse_chain(0, 0.0)) == em.rmse_chain(0, 0.0)


def test_decide_converged():
    from wgsassign_amd.device import decide_converged, guard_band
    m, tole = 1000, 1e-4
    th = tole * tole * m
    assert decide_converged(th * 0.5, m, tole, 0.25) == 1
    assert decide_converged(th * 2.0, m, tole, 0.25) == -1
    assert decide_converged(th * 1.1, m, tole, 0.25) == 0 and decide_converged(th * 0.9, m, tole, 0.25) == 0
    assert decide_converged(float("nan"), m, tole, 0.25) == -1          # NaN < tole is False (emMAF.py:23)
    assert decide_converged(0.0, m, 0.0, 0.25) == -1                     # diff < 0 never holds
    # without a floor the band is m * 2^-24 (+1e-6): tiny for small m, +-0.6 at 10^7, one-sided beyond 2^24
    assert decide_converged(th * 0.999, m, tole) == 1 and decide_converged(th * 1.001, m, tole) == -1
    assert decide_converged(th * (1 + 1e-7), m, tole) == 0
    assert abs(guard_band(10_000_000) - 0.596) < 1e-3
    big = 100_000_000
    thb = tole * tole * big
    assert decide_converged(thb * 1e-3, big, tole) == 0       # float64 can never call "converged" at this m
    assert decide_converged(thb * 6.9, big, tole) == 0 and decide_converged(thb * 7.0, big, tole) == -1


def test_guard_band_covers_float32_stagnation():
    """At whole-genome SNP counts the reference's serial float32 sum (emMAF_cy.pyx:30-31) absorbs small
    terms and lands far below the exact sum: with m = 6*10^7 and squared differences whose float64 sum
    is 1.15x the threshold, the float32 chain says "converged".  The driver must not decide from the
    float64 sum there (the float32 sum is 17 % low here and 25 %+ at 10^8: no fixed band is safe)."""
    from wgsassign_amd.device import chain_diff, decide_converged
    m, tole = 60_000_000, 1e-4
    rng = np.random.Generator(np.random.PCG64(11))
    sq = (rng.standard_normal(m, dtype=np.float32) * np.float32(1.07e-4)) ** 2       # float32 squares, as emMAF_cy.pyx:31
    s64 = float(np.sum(sq, dtype=np.float64))
    ratio = s64 / (tole * tole * m)
    assert 1.1 < ratio < 1.2
    with np.errstate(all="ignore"):
        f32 = np.cumsum(sq, dtype=np.float32)[-1]          # the serial float32 accumulation
    assert chain_diff(f32, m) < tole                       # the reference stops here ...
    assert decide_converged(s64, m, tole) == 0             # ... so the float64 sum must defer to the exact chain
    # rigorous bound behind the band: |S - F| <= m * 2^-24 * F
    assert abs(s64 - float(f32)) <= m * 2.0 ** -24 * float(f32)


@pytest.mark.parametrize("guard", [0.0, 0.25, 1e9])
def test_run_em_single_rank_matches_reference_iterations(oracle, golden, guard):
    """The driver loop stops every population at the reference's iteration (17/14/16/14/13) with
    bit-identical frequencies -- with the default guard band and with the exact chain forced on
    every iteration (guard = inf)."""
    from wgsassign_amd.device import run_em
    g = golden("amre_fit.npz")
    pops = np.unique(g["IDs"][:, 1])
    groups = [np.flatnonzero(g["IDs"][:, 1] == p) for p in pops]
    em = OracleEMBatch(oracle, g["L"], groups, guard=guard)
    iters = run_em(em, 200, 1e-4, m_total=g["L"].shape[0])
    assert list(iters) == list(g["iters"])
    for k in range(len(pops)):
        assert em.f[k].tobytes() == g["f_raw"][k].tobytes()
    if guard > 1:
        assert em.chain_calls == int(np.sum(g["iters"]))


def test_run_em_exhausted_and_nan(oracle, golden):
    from wgsassign_amd.device import run_em
    g = golden("edge.npz")
    L = g["mixed_L"]
    em = OracleEMBatch(oracle, L, [np.arange(L.shape[1] // 2)])
    assert list(run_em(em, 3, 1e-12, m_total=L.shape[0])) == [0]
    assert em.f[0].tobytes() == g["exhaust_f"].tobytes()
    # a population of size 0 (LOO of a singleton): NaN forever, runs all iterations
    em = OracleEMBatch(oracle, L, [np.array([], dtype=np.int64)])
    with np.errstate(all="ignore"):
        assert list(run_em(em, 7, 1e-4, m_total=L.shape[0])) == [0]
    assert np.isnan(em.f[0]).all()


def test_shard_range_covers_everything():
    from wgsassign_amd.comm import shard_range
    for m in (1, 7, 449, 10_000_000):
        for world in (1, 2, 3, 8):
            edges = [shard_range(m, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            if m // world >= 8192:          # long shards are cut at multiples of 8192 sites (NumPy's summation chunks) ...
                assert all(lo % 8192 == 0 for lo, _ in edges)
                assert max(hi - lo for lo, hi in edges) - m / world < 2 * 8192          # ... and stay balanced


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
from oracle import oracle as orc
from cpu_standin import OracleEMBatch
from wgsassign_amd.comm import TorchComm, shard_range
from wgsassign_amd.device import run_em
g = np.load(os.path.join({root!r}, "tests", "golden", "amre_fit.npz"))
L, IDs = g["L"], g["IDs"]
comm = TorchComm()
lo, hi = shard_range(L.shape[0], comm.rank, comm.world)
pops = np.unique(IDs[:, 1])
groups = [np.flatnonzero(IDs[:, 1] == p) for p in pops]
guard = float(sys.argv[2])
em = OracleEMBatch(orc, np.ascontiguousarray(L[lo:hi]), groups, guard=guard)
iters = run_em(em, 200, 1e-4, comm=comm, m_total=L.shape[0])
ok = list(iters) == list(g["iters"])
for k in range(len(pops)):
    ok = ok and em.f[k].tobytes() == g["f_raw"][k][lo:hi].tobytes()
print("RANK", comm.rank, "OK" if ok else "FAIL", list(iters), em.chain_calls, flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("guard", [0.25, 1e9])
def test_run_em_two_gloo_ranks_snp_sharded(tmp_path, guard):
    """world_size 2 over gloo: SNPs sharded in two contiguous ranges; the all-reduced sums and the
    rank-to-rank carry of the serial float32 chain give the single-process iterations and
    frequencies on both ranks."""
    port = free_port()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(guard)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-2000:])
        assert "RANK %d OK" % r in o


def test_utils_and_reader(golden):
    from wgsassign_amd import reader_cy, utils
    g = golden("amre_fit.npz")
    L, samples, sites = reader_cy.readBeagle(os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"))
    assert L.tobytes() == g["L"].tobytes() and samples == list(g["samples"]) and sites == list(g["sites"])
    lo = golden("amre_loo.npz")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        Lf, names = utils.filter_sites_to_common(L, sites, list(lo["sites_ds"]))
    assert buf.getvalue() == str(lo["filter_text"]) and Lf.shape[0] == 357
    assert utils.site_mask(sites, list(lo["sites_ds"])).tobytes() == lo["mask"].tobytes()
    v = np.arange(10, dtype=np.float32)
    assert list(utils.partition_loglikes(v, 3)) == [18.0, 12.0, 15.0]
    with pytest.raises(ValueError):
        utils.partition_loglikes(np.zeros((2, 2), np.float32), 2)


def test_write_ass_mats_matches_reference_text(golden, tmp_path):
    from wgsassign_amd import utils
    cli, fit, loo = golden("amre_cli.npz"), golden("amre_fit.npz"), golden("amre_loo.npz")
    out = tmp_path / "x.tsv"
    with contextlib.redirect_stdout(io.StringIO()):
        utils.write_ass_mats(str(out), loo["loo_P1"], list(fit["samples"]), fit["pops"], print_part_column=False,
                             sample_locations=fit["IDs"][:, 1], doing_LOO=True)
    assert out.read_text() == str(cli["loo_tsv"])
    outz = tmp_path / "p.tsv.gz"
    with contextlib.redirect_stdout(io.StringIO()):
        utils.write_ass_mats(str(outz), loo["parts_P3"], list(fit["samples"]), fit["pops"], partition_count=3,
                             print_part_column=True, sample_locations=fit["IDs"][:, 1], doing_LOO=True)
    import gzip
    assert gzip.open(outz, "rt").read() == str(cli["parts_tsv"])
    with pytest.raises(ValueError, match="shape mismatch"):
        utils.write_ass_mats(str(out), loo["loo_P1"][:3], list(fit["samples"]), fit["pops"])
    # The format is whatever pandas' to_csv(sep="\t", index=False, float_format="%.6f") writes for such a table
    # (utils.py:96-121): NaN as an empty field (a population of one under LOO), +-inf, names that need quoting.
    import pandas as pd
    rng = np.random.default_rng(0)
    names = ["a b", 'q"uote', "plain", "tab\there", "x,y"]
    pops = ["north", "so uth", 'w"est']
    for P, part_col, locs, is_loo in ((1, False, ["north", "north", 'w"est', "so uth", "north"], True), (3, True, None, False),
                                      (2, True, ["l1", "l2", "l3", "l4", "l5"], False), (1, True, None, False)):
        mat = (rng.standard_normal((len(names) * P, len(pops))) * 1e5).astype(np.float32)
        mat[1, 0], mat[2, 1], mat[0, 2], mat[3, 1] = np.nan, np.inf, -np.inf, -0.0
        cols = {"sample": np.repeat(names, P)}
        if locs is not None:
            cols["source_pop" if is_loo else "location"] = np.repeat(locs, P)
        if part_col:
            cols["data_part"] = np.tile(np.arange(P), len(names))
        want = pd.concat([pd.DataFrame(cols), pd.DataFrame(mat, columns=pops)], axis=1).to_csv(sep="\t", index=False, float_format="%.6f")
        with contextlib.redirect_stdout(io.StringIO()):
            utils.write_ass_mats(str(out), mat, names, pops, partition_count=P, print_part_column=part_col, sample_locations=locs,
                                 doing_LOO=is_loo)
        assert out.read_text() == want, (P, part_col)


def test_cli_flags_and_refusals(tmp_path):
    from wgsassign_amd import WGSassign
    a = WGSassign.parser.parse_args([])
    assert (a.threads, a.out, a.maf_iter, a.maf_tole, a.partition_sites) == (1, "wgsassign", 200, 1e-4, 1)
    with pytest.raises(ValueError, match="requires that --loo"):
        with contextlib.redirect_stdout(io.StringIO()):
            WGSassign.main(["--loo_downsampled_beagle", "x.gz", "--out", str(tmp_path / "o")])
    with pytest.raises(SystemExit, match="outside the scope"):
        with contextlib.redirect_stdout(io.StringIO()):
            WGSassign.main(["--get_em_mix", "--out", str(tmp_path / "o")])
    # the .args log lists non-default options only (WGSassign.py:127-141)
    with contextlib.redirect_stdout(io.StringIO()):
        WGSassign.main(["--threads", "3", "--out", str(tmp_path / "log")])
    txt = (tmp_path / "log.args").read_text()
    assert txt.startswith("WGSassign\nTime: ") and "\t-threads 3\n" in txt and "maf_iter" not in txt


def test_reference_import_paths_resolve():
    """`from WGSassign import ...` (the reference's import paths, WGSassign.py:148-159) reach this build."""
    from WGSassign import emMAF, emMAF_cy, fisher, glassy, glassy_cy, reader_cy, utils
    import WGSassign.WGSassign as cli
    assert emMAF.emMAF.__module__ == "wgsassign_amd.emMAF" and glassy.loo.__module__ == "wgsassign_amd.glassy"
    assert emMAF_cy.emMAF_update.__module__ == "wgsassign_amd.emMAF_cy" and callable(glassy_cy.loglike)
    assert callable(reader_cy.readBeagle) and callable(utils.write_ass_mats) and callable(fisher.fisher_obs)
    assert cli.parser.prog == "WGSassign" and callable(cli.main)


def test_console_script_declared_like_the_reference():
    """setup.py:47-50 of the reference installs `WGSassign=WGSassign.WGSassign:main`; pyproject.toml declares the same
    script name on this build's main(), and the alias package keeps the reference's module path importable."""
    import importlib
    import tomli
    meta = tomli.load(open(os.path.join(ROOT, "pyproject.toml"), "rb"))
    target = meta["project"]["scripts"]["WGSassign"]
    mod, func = target.split(":")
    assert callable(getattr(importlib.import_module(mod), func))
    assert set(meta["tool"]["setuptools"]["packages"]) == {"wgsassign_amd", "WGSassign"}
    import WGSassign.WGSassign as ref_path
    assert ref_path.main is getattr(importlib.import_module(mod), func)


def test_installed_numpy_sums_float32_in_8192_element_chunks():
    """The device forms `np.sum(vec, dtype=float)` (glassy.py:38) and `np.mean(vec)` of float32 vectors (fisher.py:59) in
    NumPy's own order -- which the build measured on NumPy 2.2: the reduction hands its inner loop 8192 elements at a time,
    each chunk summed pairwise (leaves of at most 128 elements, eight interleaved accumulators) and added to a running
    total.  A NumPy that sums differently (older releases may sum a contiguous vector in one pairwise pass) would silently
    break the bit-for-bit claims: this test fails loudly instead.  (The reference pins numpy<=1.22.3 for its z-score code
    only; the bit-exact claim is stated for the NumPy this test passes on.)"""
    rng = np.random.default_rng(0)

    def pairwise(a):                  # NumPy's pairwise_sum for one chunk, in the accumulator type of `a`
        n = len(a)
        if n < 8:
            r = a.dtype.type(0)
            for x in a:
                r = a.dtype.type(r + x)
            return r
        if n <= 128:
            acc = [a[i] for i in range(8)]
            for i in range(8, n - n % 8, 8):
                for j in range(8):
                    acc[j] = a.dtype.type(acc[j] + a[i + j])
            t = a.dtype.type
            r = t(t(t(acc[0] + acc[1]) + t(acc[2] + acc[3])) + t(t(acc[4] + acc[5]) + t(acc[6] + acc[7])))
            for i in range(n - n % 8, n):
                r = t(r + a[i])
            return r
        half = n // 2
        half -= half % 8
        return a.dtype.type(pairwise(a[:half]) + pairwise(a[half:]))

    for m in (8192 * 3 + 77, 20_000, 8191, 8193):
        v = (rng.random(m) * np.exp(rng.normal(0, 6, size=m))).astype(np.float32)
        # float32 accumulation (np.mean / np.sum without dtype): chunks of 8192 added to a running float32 total
        tot = np.float32(0)
        for c in range(0, m, 8192):
            tot = np.float32(tot + pairwise(v[c:c + 8192]))
        assert np.sum(v).tobytes() == tot.tobytes(), ("float32 sum", m, np.__version__)
        # float64 accumulation of float32 data (np.sum(dtype=float)): the same chunking on the cast values
        v64 = v.astype(np.float64)
        tot = np.float64(0)
        for c in range(0, m, 8192):
            tot = np.float64(tot + pairwise(v64[c:c + 8192]))
        assert np.sum(v, dtype=float).tobytes() == tot.tobytes(), ("float64 sum of float32", m, np.__version__)
