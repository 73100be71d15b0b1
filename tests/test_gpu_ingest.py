"""The device-side ingest (csrc/ingest.hip: Beagle text tokenised on the MI355X straight into the population slabs)
against the host parser (csrc/reader.cpp, itself pinned to the reference's parse by tests/test_reader_cpu.py) and the
golden parse of the bundled files (reader_cy.pyx:16-77)."""
import gzip
import os

import numpy as np
import pytest

import synth
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DATA = os.path.join(GOLDEN, "data")


from synth import bgzf_block, make_pool_file, write_bgzf  # noqa: E402


def device_rows(path, rank=0, world=1, keep=None, group_of=None, n_groups=1, env=None):
    """The matrix as the device holds it after stream_to_device, back in host layout."""
    from wgsassign_amd import reader_cy
    old = os.environ.get("WGSASSIGN_INGEST")
    if env:
        os.environ["WGSASSIGN_INGEST"] = env
    try:
        b, samples, sites, m_total = reader_cy.stream_to_device(path, group_of, n_groups, rank=rank, world=world, keep=keep)
    finally:
        if env:
            os.environ.pop("WGSASSIGN_INGEST")
            if old is not None:
                os.environ["WGSASSIGN_INGEST"] = old
    rows = b.download_rows(0, b.m)
    stats = b.ingest_stats
    b.close()
    return rows, samples, sites, stats


def same_bits(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def test_bundled_files_through_the_device_tokeniser(golden, tmp_path, monkeypatch):
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    g = golden("amre_fit.npz")
    rows, samples, sites, stats = device_rows(os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"),
                                              group_of=np.arange(85, dtype=np.int32) % 5, n_groups=5)
    assert same_bits(rows, g["L"]) and synth.digest(rows) == "432436039a568fb4"
    assert samples == list(g["samples"]) and sites == list(g["sites"])
    assert stats["host_lines"] == 0 and stats["lines"] == 449          # every value converted on the device
    a = golden("amre_assign.npz")
    rows, samples, sites, _ = device_rows(os.path.join(DATA, "amre.nonbreeding.ind34.ds_2x.sites-filter.top_50_each.beagle.gz"))
    assert same_bits(rows, a["L"]) and samples == list(a["samples"]) and sites == list(a["sites"])


@pytest.mark.parametrize("container", ["gzip", "bgzf"])
def test_awkward_tokens_flag_lines_for_the_host(tmp_path, monkeypatch, container):
    """gzip: text inflated and listed on the host; BGZF: inflated and listed on the device.  Decimal tokens (with signs, exponents, odd widths, several delimiters, CRLF, blank lines, no final newline) are
    converted on the device; inf/nan, hex floats, 16+ digits and junk flag their line for the strtod-backed host
    parser.  Either way the bits are the host parser's."""
    from wgsassign_amd import reader_cy
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    plain = ["1e-3", "+0.5", "-0.25", ".5", "5.", "1E2", "0.000001", "1.5e-003", "0.333333", "00.12345", "-0.12345", "9.999999",
             "0.1234567", "0.12345", "1.000000", "123456789012345", "1e22", "2.5E-21", "0", "7", "0.00000000000001"]
    hard = ["0.1234567890123456789", "nan", "inf", "123456789012345678", "0x1p-2", "1e", "0.5abc", "-.", "1e400", "1e-30", "2.5E-22",
            "0.33333333333333333", "0.0000000000000001"]
    rng = np.random.default_rng(2)
    n = 24

    def line(name, pool, sep="\t"):
        t = [pool[int(k)] for k in rng.integers(0, len(pool), size=3 * n)]
        return name + sep + "A" + sep + "C" + sep + sep.join(t)
    head = "marker allele1 allele2 " + " ".join("S%d S%d S%d" % (i, i, i) for i in range(n))
    lines = [head]
    kinds = []
    for s in range(400):
        hard_line = s % 7 == 3
        kinds.append(hard_line)
        lines.append(line("s%d" % s, plain + hard if hard_line else plain, sep=["\t", " ", " \t", "  "][s % 4]))
        if s % 50 == 9:
            lines.append("")
    text = "\r\n".join(lines)                                   # CRLF, no final newline
    p = str(tmp_path / "awkward.beagle.gz")

    def write(t):
        if container == "bgzf":
            write_bgzf(p, t.encode(), block=5000)
        else:
            with gzip.open(p, "wt", newline="") as fh:
                fh.write(t)
    write(text)
    want, samples_h, sites_h = reader_cy.readBeagle(p)
    rows, samples, sites, stats = device_rows(p)
    assert same_bits(rows, want) and samples == samples_h and sites == sites_h == ["s%d" % s for s in range(400)]
    assert 0 < stats["host_lines"] <= sum(kinds)                # only lines that hold a hard token went to the host
    # a short line is reported like the host reader reports it
    write(head + "\n" + line("s0", plain) + "\ns1\tA\tC\t0.1\t0.2\n")
    with pytest.raises(ValueError, match="Beagle data line 3 has fewer than %d" % (3 * n)):
        device_rows(p)


@pytest.mark.parametrize("layout", ["plain", "bgzf", "bgzf_tiny_blocks", "no_final_newline_blank_lines"])
def test_device_ingest_equals_host_parser(tmp_path, monkeypatch, layout):
    """Random matrices in every container the reader knows, small text chunks (many carries), populations interleaved
    over the slabs, ranks 0..2 of 3 (odd site0), a site mask: the slabs hold the host parser's floats, bit for bit."""
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    monkeypatch.setenv("WGSASSIGN_TEXT_CHUNK_BYTES", str(1 << 20))
    m, n, K = 30_011, 53, 4
    L, _ = synth.make_beagle(m, n, K, seed=17)
    head = "marker\tallele1\tallele2\t" + "\t".join("I%d\tI%d\tI%d" % (i, i, i) for i in range(n))
    body = []
    for s in range(m):
        vals = ["%.6f\t%.6f\t%.6f" % (L[s, 2 * i], L[s, 2 * i + 1], max(0.0, 1 - L[s, 2 * i] - L[s, 2 * i + 1])) for i in range(n)]
        body.append("chr%d_%d\tA\tG\t" % (s % 5, s) + "\t".join(vals))
        if "blank" in layout and s % 101 == 7:
            body.append("")
    text = (head + "\n" + "\n".join(body) + ("" if "no_final" in layout else "\n")).encode()
    p = str(tmp_path / ("f_%s.beagle.gz" % layout))
    if layout.startswith("bgzf"):
        write_bgzf(p, text, block=900 if "tiny" in layout else 60000)
    else:
        with gzip.open(p, "wb", compresslevel=6) as fh:
            fh.write(text)
    names = ["chr%d_%d" % (s % 5, s) for s in range(m)]
    group_of = (np.arange(n) * 7 % K).astype(np.int32)
    rows, _, sites, stats = device_rows(p, group_of=group_of, n_groups=K)
    assert same_bits(rows, L) and sites == names and stats["host_lines"] == 0
    assert stats["chunks"] > 5
    from wgsassign_amd.comm import shard_range
    for rank in range(3):
        lo, hi = shard_range(m, rank, 3)
        rows, _, sites, _ = device_rows(p, rank=rank, world=3, group_of=group_of, n_groups=K)
        assert same_bits(rows, L[lo:hi]) and sites == names[lo:hi], rank
    keep = np.random.default_rng(4).random(m) < 0.8
    for rank in range(2):
        lo, hi = shard_range(int(keep.sum()), rank, 2)
        rows, _, sites, _ = device_rows(p, rank=rank, world=2, keep=keep, group_of=group_of, n_groups=K)
        assert same_bits(rows, L[keep][lo:hi]) and sites == [x for x, k in zip(names, keep) if k][lo:hi], rank
        rows_h, _, sites_h, _ = device_rows(p, rank=rank, world=2, keep=keep, group_of=group_of, n_groups=K, env="host")
        assert same_bits(rows, rows_h) and sites == sites_h


def test_n2000_bgzf_two_ranks_on_one_card(tmp_path, monkeypatch):
    """A shard of the streamed configuration (BASELINE configs[4]: n = 2000 individuals, K = 20 populations, BGZF as ANGSD
    writes it): 200 k sites = 10.8 GB of text through the device tokeniser, as ranks 0 and 1 of 2 and as a reader started
    at an odd row, against the host parser and against the values the file was written from."""
    from wgsassign_amd import reader_cy
    from wgsassign_amd.comm import shard_range
    from wgsassign_amd.device import DeviceBeagle
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    n, m, K = 2000, 200_000, 20
    p = str(tmp_path / "c5.beagle.gz")
    vals, pick = make_pool_file(p, n, m)
    group_of = (np.arange(n) % K).astype(np.int32)
    idx, _, sites = reader_cy.ensure_index(p)
    assert sites == m
    report = {}
    for rank in range(2):
        lo, hi = shard_range(m, rank, 2)
        import time
        t0 = time.perf_counter()
        b, samples, names, _ = reader_cy.stream_to_device(p, group_of, K, rank=rank, world=2, names="ends")
        dt = time.perf_counter() - t0
        st = b.ingest_stats
        report[rank] = "%.0f sites/s (%.2f GB/s of text; waited %.1f s for inflate, device %.0f ms)" % (
            (hi - lo) / dt, st["text_bytes"] / dt / 1e9, st["wait_s"], st["device_ms"])
        assert b.m == hi - lo and b.site0 == lo and len(samples) == n and st["host_lines"] == 0
        assert names[:2] == ["chr7_%d" % (lo + 1), "chr7_%d" % (lo + 2)] and names[-1] == "chr7_%d" % hi
        # against the values the file was written from, 4096 rows at a time
        for r0 in range(0, hi - lo, 4096):
            k = min(4096, hi - lo - r0)
            got = b.download_rows(r0, k)
            assert got.tobytes() == vals[pick[lo + r0:lo + r0 + k]].tobytes(), (rank, r0)
        if rank == 1:
            # ... and against the host parser on a stretch of the same shard
            with reader_cy.BeagleStream(p, index=idx, first_row=lo + 12_345) as stream:
                rows, sn = next(stream.chunks(max_rows=3000))
            assert rows.tobytes() == b.download_rows(12_345, 3000).tobytes() and sn[0] == "chr7_%d" % (lo + 12_346)
        b.close()
    print("device ingest, n=2000 BGZF:", report)
    # a reader opened at an odd row, ingesting into rows 5.. of a matrix whose first global site is odd
    first, cnt = 77_777, 10_001
    b = DeviceBeagle(cnt + 5, n, group_of, K, site0=first - 5)
    with reader_cy.BeagleStream(p, index=idx, first_row=first) as stream:
        done = sum(k for k, _ in stream.ingest(b, row0=5, limit=cnt))
    assert done == cnt and b.download_rows(5, cnt).tobytes() == vals[pick[first:first + cnt]].tobytes()
    assert not b.download_rows(0, 5).any()
    b.close()


def test_members_the_device_rejects_and_damaged_files(tmp_path, monkeypatch):
    """Members the inflate kernel does not accept are inflated on the host and patched into the device text (forced here
    for every third member, whose device text is wiped first): the same slabs.  A member with a damaged deflate stream --
    rejected by the device AND by the host's inflater -- and a truncated file are errors, not short matrices."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import beagle_files
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    monkeypatch.setenv("WGSASSIGN_TEXT_CHUNK_BYTES", str(8 << 20))
    n, m = 300, 14000                                            # 113 MB of text, 14 chunks of 8 MiB
    p = str(tmp_path / "r.beagle.gz")
    _, vals, pick = beagle_files.write_lowdepth_bgzf(p, n, m, pool=128)
    want = vals[pick]
    rows, _, sites, stats = device_rows(p)
    assert same_bits(rows, want) and stats["blocks_left_to_host_inflater"] == 0 and stats["blocks_inflated_on_device"] > 0
    monkeypatch.setenv("WGSASSIGN_DEBUG_REJECT_MEMBERS", "3")
    rows, _, sites2, stats = device_rows(p)
    assert same_bits(rows, want) and sites2 == sites and stats["blocks_left_to_host_inflater"] >= stats["blocks_inflated_on_device"] // 3
    monkeypatch.delenv("WGSASSIGN_DEBUG_REJECT_MEMBERS")
    from wgsassign_amd import reader_cy
    import shutil
    reader_cy.ensure_index(p)
    raw = bytearray(open(p, "rb").read())
    k = bytes(raw).find(b"\x1f\x8b\x08\x04", len(raw) * 9 // 10)
    for i in range(k + 40, k + 60):
        raw[i] ^= 0x5A                                           # inside that member's deflate stream
    dmg = str(tmp_path / "dmg.beagle.gz")
    with open(dmg, "wb") as fh:
        fh.write(bytes(raw))
    # the damaged copy inherits the good file's index (same size, same modification time, same member layout), so that the
    # damage is met by the ingest and not by the index pass
    st = os.stat(p)
    os.utime(dmg, ns=(st.st_atime_ns, st.st_mtime_ns))
    for a, b in zip(reader_cy.index_paths(p), reader_cy.index_paths(dmg)):
        if os.path.exists(a):
            shutil.copyfile(a, b)
            os.chmod(b, 0o600)
    with pytest.raises(RuntimeError, match="corrupt block"):
        device_rows(dmg)
    cut = str(tmp_path / "cut.beagle.gz")
    with open(cut, "wb") as fh:
        fh.write(bytes(open(p, "rb").read()[:-3000]))
    with pytest.raises(RuntimeError, match="read error|corrupt|truncated"):
        device_rows(cut)


def test_cold_file_goes_through_in_one_pass(tmp_path, monkeypatch):
    """A BGZF file without an index, one rank (reader_cy._stream_cold_file): the index pass and the device ingest run at the same
    time, the matrix is created for an ESTIMATE of the site count and cut to the file's sites afterwards.  Same slabs, names,
    site count and cached index as the usual way (index pass first); an estimate that is too small by more than the margin
    falls back to the usual way; fits over the trimmed matrix have the bits of fits over a matrix created at its size."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import beagle_files
    from wgsassign_amd import device as dev
    from wgsassign_amd import reader_cy
    n, m, K = 120, 9000, 3
    p = str(tmp_path / "cold.beagle.gz")
    _, vals, pick = beagle_files.write_lowdepth_bgzf(p, n, m, pool=256)
    want = vals[pick]
    group_of = (np.arange(n) % K).astype(np.int32)
    est = reader_cy.estimate_sites(p)
    assert est is not None and abs(est - m) <= m // 50

    def run(index_dir, one_pass, estimate=None):
        monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(index_dir))
        monkeypatch.setenv("WGSASSIGN_COLD_ONE_PASS", "1" if one_pass else "0")
        os.makedirs(index_dir, mode=0o700, exist_ok=True)
        calls = []
        real, real_estimate = reader_cy._stream_cold_file, reader_cy.estimate_sites

        def spy(*a, **k):
            out = real(*a, **k)
            calls.append(out is not None)
            return out
        monkeypatch.setattr(reader_cy, "_stream_cold_file", spy)
        if estimate is not None:
            monkeypatch.setattr(reader_cy, "estimate_sites", lambda path: estimate)
        try:
            b, samples, sites, m_total = reader_cy.stream_to_device(p, group_of, K, names="all")
        finally:
            monkeypatch.setattr(reader_cy, "_stream_cold_file", real)
            if estimate is not None:
                monkeypatch.setattr(reader_cy, "estimate_sites", real_estimate)
        idx = reader_cy.index_paths(p)[0]
        return b, samples, sites, m_total, calls, os.path.exists(idx)

    b0, samples0, sites0, m0, calls0, idx0 = run(tmp_path / "usual", False)
    assert calls0 == [] and idx0 and m0 == m and b0.m == m
    b1, samples1, sites1, m1, calls1, idx1 = run(tmp_path / "cold", True)
    assert calls1 == [True] and idx1 and m1 == m and b1.m == m            # the one-pass way was taken, and left the index behind
    assert samples1 == samples0 and sites1 == sites0
    r0, r1 = b0.download_rows(0, m), b1.download_rows(0, m)
    assert same_bits(r0, want) and same_bits(r1, want)
    assert open(reader_cy.index_paths(p)[0], "rb").read() == open(str(tmp_path / "usual" / os.path.basename(reader_cy.index_paths(p)[0])), "rb").read()
    # a second call finds the index and goes the usual way
    b2, _, _, _, calls2, _ = run(tmp_path / "cold", True)
    assert calls2 == [False] and same_bits(b2.download_rows(0, m), want)
    b2.close()
    # an estimate of a third of the sites: the ingest runs out of rows, the usual way takes over
    b3, _, sites3, m3, calls3, _ = run(tmp_path / "small", True, estimate=m // 3)
    assert calls3 == [False] and m3 == m and sites3 == sites0 and same_bits(b3.download_rows(0, m), want)
    b3.close()
    # the trimmed matrix (created for ~11 k rows, cut to 9000; nbytes of the rows in use) fits like one created at its size
    assert b1.nbytes() == b0.nbytes()
    pops = np.arange(K, dtype=np.int32)
    f0 = dev.EMBatch(b0, pops)
    f1 = dev.EMBatch(b1, pops)
    f0.fit(40, 1e-4)
    f1.fit(40, 1e-4)
    for k in range(K):
        assert same_bits(f0.get_f(k), f1.get_f(k)), k
    with pytest.raises(ValueError, match="in use"):
        b1.set_rows(m - 1)
    f0.close()
    f1.close()
    with pytest.raises(ValueError, match="cannot be set"):
        b1.set_rows(m + 1)
    b1.set_rows(m - 64 * 20)                                               # fewer tiles: still the first rows
    assert b1.m == m - 1280 and same_bits(b1.download_rows(0, b1.m), want[:b1.m])
    b0.close()
    b1.close()


def test_cold_file_one_pass_at_the_streamed_configurations_width(tmp_path, monkeypatch):
    """The one-pass way on a shard as wide as BASELINE configs[4] (n = 2000, K = 20; 60 k sites = 3.2 GB of text, several chunks,
    an estimate from five samples of a 160 MB file): EVERY row of the trimmed matrix equals the matrix of the usual way (index
    first), which equals the source values; the estimate is within 2 %."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import beagle_files
    from wgsassign_amd import reader_cy
    n, m, K = 2000, 60_000, 20
    p = str(tmp_path / "wide.beagle.gz")
    _, vals, pick = beagle_files.write_lowdepth_bgzf(p, n, m)
    group_of = (np.arange(n) % K).astype(np.int32)
    est = reader_cy.estimate_sites(p)
    assert est is not None and abs(est - m) <= m // 50
    out = {}
    for label, one_pass in (("usual", "0"), ("cold", "1")):
        monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path / label))
        os.makedirs(str(tmp_path / label), mode=0o700, exist_ok=True)
        monkeypatch.setenv("WGSASSIGN_COLD_ONE_PASS", one_pass)
        b, samples, sites, m_total = reader_cy.stream_to_device(p, group_of, K, names="ends")
        assert m_total == m and b.m == m and os.path.exists(reader_cy.index_paths(p)[0])
        digests = []
        for r0 in range(0, m, 10_000):                              # 160 MB of rows at a time
            rows = b.download_rows(r0, min(10_000, m - r0))
            assert same_bits(rows, vals[pick[r0:r0 + rows.shape[0]]]), (label, r0)
            digests.append(synth.digest(rows))
        out[label] = (digests, samples, sites, b.nbytes(), b.ingest_stats["chunks"])
        b.close()
    assert out["usual"][:4] == out["cold"][:4] and out["cold"][4] >= 2
