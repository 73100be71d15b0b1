"""End-to-end: the `WGSassign` command line of this build against the reference CLI's recorded
outputs on the bundled AMRE data (tests/golden/amre_cli.npz, made by running the real reference)."""
import contextlib
import gzip
import io
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DATA = os.path.join(GOLDEN, "data")
BREED = os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz")
IDS = os.path.join(DATA, "amre.breeding.ind85.reference_k5.IDs.txt")
NONBREED = os.path.join(DATA, "amre.nonbreeding.ind34.ds_2x.sites-filter.top_50_each.beagle.gz")


def run_cli(argv):
    from wgsassign_amd import WGSassign
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        WGSassign.main(argv)
    return buf.getvalue()


def table(text):
    rows = [line.split("\t") for line in text.strip().split("\n")]
    return rows[0], rows[1:]


def test_reference_af_loo_partitions(tmp_path, golden):
    g = golden("amre_cli.npz")
    out = str(tmp_path / "ref")
    stdout = run_cli(["--beagle", BREED, "--pop_af_IDs", IDS, "--get_reference_af", "--loo", "--partition_sites", "3",
                      "--out", out, "--threads", "2"])
    # binary allele frequencies: identical bytes; population names file identical
    assert np.load(out + ".pop_af.npy").tobytes() == g["pop_af_npy"].tobytes()
    assert open(out + ".pop_names.txt").read() == str(g["pop_names"])
    # stdout: same lines as the reference except paths (the reference run used a temp dir)
    ref_lines = [l for l in str(g["stdout_ref"]).replace("<TMP>/", "").splitlines()]
    got_lines = [l.replace(str(tmp_path) + "/", "") for l in stdout.splitlines()]
    assert got_lines == ref_lines
    # LOO table: same header/labels; values within 1e-6 relative of the reference's printed numbers
    h_ref, r_ref = table(str(g["loo_tsv"]))
    h_got, r_got = table(open(out + ".pop_like_LOO.tsv").read())
    assert h_got == h_ref and [r[:2] for r in r_got] == [r[:2] for r in r_ref]
    a = np.array([[float(x) for x in r[2:]] for r in r_got])
    b = np.array([[float(x) for x in r[2:]] for r in r_ref])
    assert np.all(np.abs(a - b) <= 1e-6 * np.abs(b) + 1.5e-6)        # %.6f text: half a unit of the last digit each side
    h_ref, r_ref = table(str(g["parts_tsv"]))
    h_got, r_got = table(gzip.open(out + ".pop_like_LOO_partitions_3.tsv.gz", "rt").read())
    assert h_got == h_ref and [r[:3] for r in r_got] == [r[:3] for r in r_ref]
    assert r_got == r_ref                       # partition sums are bit-exact -> identical text
    assert os.path.exists(out + ".args")


def test_get_pop_like(tmp_path, golden):
    g = golden("amre_cli.npz")
    af = tmp_path / "ref.pop_af.npy"
    np.save(af, g["pop_af_npy"])
    out = str(tmp_path / "nb")
    stdout = run_cli(["--beagle", NONBREED, "--pop_af_file", str(af), "--get_pop_like", "--out", out, "--threads", "2"])
    got = np.loadtxt(out + ".pop_like.txt")
    ref = np.loadtxt(io.StringIO(str(g["pop_like_txt"])))
    assert got.shape == ref.shape == (34, 5)
    assert np.all(np.abs(got - ref) <= 1e-6 * np.abs(ref))
    same = open(out + ".pop_like.txt").read() == str(g["pop_like_txt"])
    print("pop_like.txt text identical to the reference's:", same)
    assert [l.replace(str(tmp_path) + "/", "") for l in stdout.splitlines()] == \
        str(g["stdout_like"]).replace("<TMP>/", "").splitlines()


def test_loo_downsampled_cli(tmp_path, golden):
    """--loo --loo_downsampled_beagle: stdout, filtered allele frequencies and the TSV of the reference CLI."""
    g = golden("amre_cli_downsampled.npz")
    ds = os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each_subset_80percent_sites.beagle.gz")
    out = str(tmp_path / "ds")
    stdout = run_cli(["--beagle", BREED, "--pop_af_IDs", IDS, "--get_reference_af", "--loo", "--loo_downsampled_beagle", ds,
                      "--out", out, "--threads", "2"])
    assert np.load(out + ".pop_af.npy").tobytes() == g["pop_af_npy"].tobytes()
    ref_lines = str(g["stdout"]).replace("<TMP>/", "").splitlines()
    assert [l.replace(str(tmp_path) + "/", "") for l in stdout.splitlines()] == ref_lines
    h_ref, r_ref = table(str(g["loo_tsv"]))
    h_got, r_got = table(open(out + ".pop_like_LOO_downsampled.tsv").read())
    assert h_got == h_ref and [r[:2] for r in r_got] == [r[:2] for r in r_ref]
    a = np.array([[float(x) for x in r[2:]] for r in r_got])
    b = np.array([[float(x) for x in r[2:]] for r in r_ref])
    assert np.all(np.abs(a - b) <= 1e-6 * np.abs(b) + 1.5e-6)


def test_cli_streams_in_many_chunks(tmp_path, golden, monkeypatch):
    """The CLI streams the file in chunks (here ~40 sites each): same outputs as in one piece."""
    monkeypatch.setenv("WGSASSIGN_CHUNK_BYTES", str(40 * 170 * 4))
    g = golden("amre_cli.npz")
    out = str(tmp_path / "ref")
    run_cli(["--beagle", BREED, "--pop_af_IDs", IDS, "--get_reference_af", "--loo", "--partition_sites", "3",
             "--out", out, "--threads", "2"])
    assert np.load(out + ".pop_af.npy").tobytes() == g["pop_af_npy"].tobytes()
    assert gzip.open(out + ".pop_like_LOO_partitions_3.tsv.gz", "rt").read() == str(g["parts_tsv"])


def test_cli_on_bgzf_copies_of_the_bundled_files(tmp_path, golden, monkeypatch):
    """The bundled files re-written as BGZF (what ANGSD writes): the command line then reads them through the
    device-resident ingest (members inflated, cut into lines and tokenised on the MI355X; a site filter applied while
    streaming) -- same `.pop_af.npy` bytes, same stdout, same partition sums and downsampled TSV as the reference CLI."""
    import synth
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    monkeypatch.setenv("WGSASSIGN_TEXT_CHUNK_BYTES", str(1 << 20))          # the 1 MB of text in two chunks
    ds_src = os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each_subset_80percent_sites.beagle.gz")
    paths = {}
    for name, src, block in (("breed", BREED, 20000), ("ds", ds_src, 60000), ("nonbreed", NONBREED, 3000)):
        paths[name] = str(tmp_path / (name + ".beagle.gz"))
        synth.write_bgzf(paths[name], gzip.open(src, "rb").read(), block=block)
    g = golden("amre_cli.npz")
    out = str(tmp_path / "ref")
    stdout = run_cli(["--beagle", paths["breed"], "--pop_af_IDs", IDS, "--get_reference_af", "--loo", "--partition_sites", "3",
                      "--out", out, "--threads", "2"])
    assert np.load(out + ".pop_af.npy").tobytes() == g["pop_af_npy"].tobytes()
    ref_lines = str(g["stdout_ref"]).replace("<TMP>/", "").splitlines()
    got_lines = [l.replace(str(tmp_path) + "/", "") for l in stdout.splitlines()]
    assert [l for l in got_lines if "Parsing" not in l] == [l for l in ref_lines if "Parsing" not in l]
    assert gzip.open(out + ".pop_like_LOO_partitions_3.tsv.gz", "rt").read() == str(g["parts_tsv"])
    out2 = str(tmp_path / "nb")
    af = tmp_path / "ref.pop_af.npy"
    np.save(af, g["pop_af_npy"])
    run_cli(["--beagle", paths["nonbreed"], "--pop_af_file", str(af), "--get_pop_like", "--out", out2, "--threads", "2"])
    got = np.loadtxt(out2 + ".pop_like.txt")
    ref = np.loadtxt(io.StringIO(str(g["pop_like_txt"])))
    assert got.shape == ref.shape and np.all(np.abs(got - ref) <= 1e-6 * np.abs(ref))
    gd = golden("amre_cli_downsampled.npz")
    out3 = str(tmp_path / "ds")
    run_cli(["--beagle", paths["breed"], "--pop_af_IDs", IDS, "--get_reference_af", "--loo", "--loo_downsampled_beagle", paths["ds"],
             "--out", out3, "--threads", "2"])
    assert np.load(out3 + ".pop_af.npy").tobytes() == gd["pop_af_npy"].tobytes()
    h_ref, r_ref = table(str(gd["loo_tsv"]))
    h_got, r_got = table(open(out3 + ".pop_like_LOO_downsampled.tsv").read())
    assert h_got == h_ref and [r[:2] for r in r_got] == [r[:2] for r in r_ref]
    a = np.array([[float(x) for x in r[2:]] for r in r_got])
    b = np.array([[float(x) for x in r[2:]] for r in r_ref])
    assert np.all(np.abs(a - b) <= 1e-6 * np.abs(b) + 1.5e-6)


@pytest.mark.parametrize("ranks, cases", [(1, 16), (3, 4)])
def test_random_cases_through_the_command_line(ranks, cases):
    """tools/fuzz_cli.py: random shapes (populations of one, one SNP, tile edges, 1-7 partitions, --ne_obs), plain gzip or
    BGZF files, one process or `--gpus 3` on the one card: every output file equals, byte for byte or character for
    character, what the oracle pipeline writes for the same matrix."""
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_cli.py"), str(cases), "2026", str(ranks)], capture_output=True,
                       text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "%d cases, 0 mismatches" % cases in r.stdout


def test_cli_on_quality_dependent_likelihoods(tmp_path, oracle, monkeypatch):
    """A BGZF Beagle file whose likelihoods come from per-read base qualities (tests/synth.py: make_beagle_quality: ~39 classes per
    SNP among 200 individuals -- 128-slot hash tables in the class encoder, 8 SNPs per table of the coded scoring sweep) through
    the command line: `--get_reference_af` then `--get_pop_like` give the oracle pipeline's `.pop_af.npy` bytes and `%.7f` text."""
    import sys
    import synth
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_cli
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    m, n, K = 2500, 200, 2
    L, IDs = synth.make_beagle_quality(m, n, K, seed=11)
    assert np.percentile(synth.classes_per_snp(L), 99) > 44            # beyond the 64-slot tables
    path, ids = str(tmp_path / "q.beagle.gz"), str(tmp_path / "ids.txt")
    bench_cli.write_beagle(path, L, ids, IDs, "bgzf")
    out = str(tmp_path / "q")
    run_cli(["--beagle", path, "--pop_af_IDs", ids, "--get_reference_af", "--out", out, "--threads", "2"])
    pops, af, _, _ = oracle.fit_reference_af(L, IDs, t=4)
    assert np.load(out + ".pop_af.npy").tobytes() == af.tobytes()
    run_cli(["--beagle", path, "--pop_af_file", out + ".pop_af.npy", "--get_pop_like", "--out", out, "--threads", "2"])
    with np.errstate(all="ignore"):
        ll = oracle.assignLL(L, af.copy(), 4)
    want = io.StringIO()
    np.savetxt(want, ll, fmt="%.7f")                          # WGSassign.py:306
    assert open(out + ".pop_like.txt").read() == want.getvalue()
