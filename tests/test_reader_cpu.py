"""CPU-only: the native streamed Beagle reader (csrc/reader.cpp) against the reference's parsing
rules (reader_cy.pyx:16-77) -- golden parse of the bundled files, a pure-Python restatement,
awkward tokens, chunking."""
import gzip
import os
import time

import numpy as np
import pytest

import synth
from conftest import GOLDEN

DATA = os.path.join(GOLDEN, "data")


def write_beagle(path, L, names=None, gl2=None, line_end="\n", final_newline=True, fmt="%.6f"):
    m, n = L.shape[0], L.shape[1] // 2
    names = names or ["Ind%d" % i for i in range(n)]
    head = "marker\tallele1\tallele2\t" + "\t".join("%s\t%s\t%s" % (x, x, x) for x in names)
    lines = [head]
    for s in range(m):
        vals = []
        for i in range(n):
            g0, g1 = L[s, 2 * i], L[s, 2 * i + 1]
            vals += [fmt % g0, fmt % g1, fmt % max(0.0, 1 - g0 - g1)]
        lines.append("chr1_%d\t0\t1\t" % (s + 1) + "\t".join(vals))
    text = line_end.join(lines) + (line_end if final_newline else "")
    with gzip.open(path, "wt", newline="") as fh:
        fh.write(text)


def test_bundled_files_match_reference_parse(golden):
    from wgsassign_amd import reader_cy
    g = golden("amre_fit.npz")
    L, samples, sites = reader_cy.readBeagle(os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each.beagle.gz"))
    assert L.dtype == np.float32 and L.flags.c_contiguous
    assert L.tobytes() == g["L"].tobytes() and synth.digest(L) == "432436039a568fb4"
    assert samples == list(g["samples"]) and sites == list(g["sites"])
    a = golden("amre_assign.npz")
    L, samples, sites = reader_cy.readBeagle(os.path.join(DATA, "amre.nonbreeding.ind34.ds_2x.sites-filter.top_50_each.beagle.gz"))
    assert L.tobytes() == a["L"].tobytes() and samples == list(a["samples"]) and sites == list(a["sites"])
    lo = golden("amre_loo.npz")
    L, _, sites = reader_cy.readBeagle(os.path.join(DATA, "amre.breeding.ind85.ds_2x.sites-filter.top_50_each_subset_80percent_sites.beagle.gz"))
    assert L.tobytes() == lo["L_ds"].tobytes() and sites == list(lo["sites_ds"])


def test_native_equals_python_restatement_and_chunks(tmp_path):
    from wgsassign_amd import reader_cy
    L, _ = synth.make_beagle(3001, 23, 3, seed=9)
    p = str(tmp_path / "a.beagle.gz")
    write_beagle(p, L)
    Ln, sn, sites_n = reader_cy.readBeagle(p)
    Lp, sp, sites_p = reader_cy.readBeagle_py(p)
    assert Ln.tobytes() == Lp.tobytes() == L.tobytes() and sn == sp and sites_n == sites_p
    assert reader_cy.count_sites(p) == 3001
    with reader_cy.BeagleStream(p, threads=3) as st:
        parts = [(r.copy(), names) for r, names in st.chunks(max_rows=257)]
    assert [len(x[1]) for x in parts] == [257] * 11 + [3001 - 11 * 257]
    assert np.concatenate([x[0] for x in parts]).tobytes() == L.tobytes()
    assert sum((x[1] for x in parts), []) == sites_p


def test_awkward_tokens_and_line_ends(tmp_path):
    """atof semantics: exponents, long mantissas, signs, nan/inf, mixed blanks, CRLF, blank lines,
    no final newline."""
    from wgsassign_amd import reader_cy
    toks = ["1e-3", "0.1234567890123456789", "+0.5", "-0.25", "nan", "inf", ".5", "5.", "1E2", "0.000001",
            "0.3333333333333333", "123456789012345678", "0x1p-2", "  0.75",
            # eight characters like ANGSD's d.dddddd, but not that format -- and the format itself between them
            "1.5e-003", "0.333333", "00.12345", "-0.12345", "9.999999", "0.1234567", "0.12345", "1.000000"]
    vals = " \t".join(t.strip() for t in toks)
    n = len(toks) // 2
    head = "marker allele1 allele2 " + " ".join("S%d S%d S%d" % (i, i, i) for i in range(n))
    row = lambda name: name + "\tA\tC\t" + "\t".join(
        "%s\t%s\t0.0" % (toks[2 * i].strip(), toks[2 * i + 1].strip()) for i in range(n))
    row2 = row("s2").replace("\t", " \t")                                          # two delimiters between tokens
    text = head + "\r\n" + row("s1") + "\r\n\r\n" + row2 + "\n" + row("s3")        # no final newline
    p = str(tmp_path / "b.beagle.gz")
    with gzip.open(p, "wt", newline="") as fh:
        fh.write(text)
    L, samples, sites = reader_cy.readBeagle(p)
    want = np.array([float.fromhex(t.strip()) if "x" in t else float(t) for t in toks], dtype=np.float64).astype(np.float32)
    assert samples == ["S%d" % i for i in range(n)] and sites == ["s1", "s2", "s3"]
    assert L.shape == (3, 2 * n)
    for r in range(3):
        assert np.array_equal(L[r], want, equal_nan=True)
    assert reader_cy.count_sites(p) == 3
    # header only
    p2 = str(tmp_path / "c.beagle.gz")
    with gzip.open(p2, "wt") as fh:
        fh.write(head + "\n")
    L, samples, sites = reader_cy.readBeagle(p2)
    assert L.shape == (0, 2 * n) and sites == [] and reader_cy.count_sites(p2) == 0
    # short line -> error, not a wild read
    p3 = str(tmp_path / "d.beagle.gz")
    with gzip.open(p3, "wt") as fh:
        fh.write(head + "\ns1\tA\tC\t0.1\t0.2\n")
    with pytest.raises(ValueError, match="fewer than"):
        reader_cy.readBeagle(p3)
    with pytest.raises(ValueError, match="cannot open"):
        reader_cy.readBeagle(str(tmp_path / "missing.gz"))


def test_six_decimal_text_round_trip_exhaustive_sample():
    """The fast decimal path (mantissa / 10^k) must equal strtod for every ANGSD-style token:
    all 1,000,001 six-decimal values 0.000000 .. 1.000000."""
    import ctypes
    from wgsassign_amd import reader_cy
    import tempfile
    vals = np.arange(0, 1_000_001)
    toks = ["%d.%06d" % (v // 1_000_000, v % 1_000_000) for v in vals]
    want = np.array([float(t) for t in toks]).astype(np.float32)
    n = 1000
    head = "m a b " + " ".join("S%d S%d S%d" % (i, i, i) for i in range(n))
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "e.beagle.gz")
        with gzip.open(p, "wt", compresslevel=1) as fh:
            fh.write(head + "\n")
            for r in range(0, 1_000_000, 2 * n):
                t = toks[r:r + 2 * n]
                fh.write("s%d A C " % r + " ".join("%s %s 0" % (t[2 * i], t[2 * i + 1]) for i in range(n)) + "\n")
        L, _, _ = reader_cy.readBeagle(p)
    assert L.shape == (500, 2 * n) and np.array_equal(L.reshape(-1), want[:1_000_000])


def test_speed_report(tmp_path):
    from wgsassign_amd import reader_cy
    L, _ = synth.make_beagle(20_000, 100, 5, seed=4)
    p = str(tmp_path / "big.beagle.gz")
    write_beagle(p, L)
    t0 = time.perf_counter()
    Ln, _, sites = reader_cy.readBeagle(p)
    t_native = time.perf_counter() - t0
    assert Ln.tobytes() == L.tobytes() and len(sites) == 20_000
    print("native reader: %.0f sites/s at n=100 (%d threads)" % (20_000 / t_native, min(len(os.sched_getaffinity(0)), 16)))
    assert 20_000 / t_native > 30_500      # the reference's readBeagle: 30.5 k sites/s at n=100


# ------------------------------------------------------------------ random access (wgs_reader_build_index)
def _bgzf_like(path, text, block=7000):
    """Concatenated gzip members (what bgzip / ANGSD's BGZF writer produce: one member per <= 64 KiB block)."""
    data = text.encode()
    with open(path, "wb") as fh:
        for i in range(0, len(data), block):
            fh.write(gzip.compress(data[i:i + block], compresslevel=6))


def _text_of(L, blank_lines=False, final_newline=True):
    m, n = L.shape[0], L.shape[1] // 2
    head = "marker\tallele1\tallele2\t" + "\t".join("I%d\tI%d\tI%d" % (i, i, i) for i in range(n))
    lines = [head]
    for s in range(m):
        vals = []
        for i in range(n):
            g0, g1 = L[s, 2 * i], L[s, 2 * i + 1]
            vals += ["%.6f" % g0, "%.6f" % g1, "%.6f" % max(0.0, 1 - g0 - g1)]
        lines.append("ctg%d_%d\t0\t1\t" % (s % 7, s + 1) + "\t".join(vals))
        if blank_lines and s % 97 == 5:
            lines.append("")
    return "\n".join(lines) + ("\n" if final_newline else "")


@pytest.mark.parametrize("layout", ["plain", "members", "plain_no_final_newline_blank_lines"])
def test_indexed_open_starts_at_any_row(tmp_path, monkeypatch, layout):
    """An index built in one pass lets a reader start at ANY row and return exactly the rows a from-the-start
    reader returns there: plain gzip (access points at deflate-block boundaries, 32 KiB dictionaries), concatenated
    members (BGZF-like: member starts need no dictionary), blank lines and a missing final newline."""
    import ctypes
    from wgsassign_amd import _lib, reader_cy
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    m, n = 2500, 11
    L, _ = synth.make_beagle(m, n, 2, seed=31)
    text = _text_of(L, blank_lines="blank" in layout, final_newline="no_final" not in layout)
    p = str(tmp_path / ("x_%s.beagle.gz" % layout))
    if layout == "members":
        _bgzf_like(p, text)
    else:
        with gzip.open(p, "wt", newline="", compresslevel=6) as fh:
            fh.write(text)
    lib = _lib.load()
    idx = str(tmp_path / "i.idx")
    nam = str(tmp_path / "i.names")
    sites = ctypes.c_int64()
    _lib.check(lib.wgs_reader_build_index(p.encode(), idx.encode(), nam.encode(), 40_000, 1000, ctypes.byref(sites)))
    assert sites.value == m == reader_cy.count_sites(p)
    names = open(nam).read().split("\n")[:-1]
    assert names == ["ctg%d_%d" % (s % 7, s + 1) for s in range(m)]
    assert os.path.getsize(idx) > 3000 or layout == "members"     # access points with their (deflated) 32 KiB dictionaries
    for first in [0, 1, 2, 63, 64, 500, 1234, 2498, 2499]:
        with reader_cy.BeagleStream(p, threads=2, index=idx, first_row=first) as st:
            assert st.n == n and st.sample_names == ["I%d" % i for i in range(n)]
            rows, sn = next(st.chunks(max_rows=40))
        k = min(40, m - first)
        assert rows.shape[0] == k and sn == names[first:first + k], first
        assert rows.tobytes() == L[first:first + k].tobytes(), first
    # whole ranges: the stretches between access points are inflated in parallel, `threads` at a time, the one after
    # the last point by the serial stream -- the same rows and names as one stream from the first byte
    for first, threads in [(0, 4), (1234, 3), (2400, 16), (0, 1)]:
        with reader_cy.BeagleStream(p, threads=threads, index=idx, first_row=first) as st:
            got = list(st.chunks(max_rows=333))
        assert np.concatenate([r for r, _ in got]).tobytes() == L[first:].tobytes(), (first, threads)
        assert [x for _, ns in got for x in ns] == names[first:], (first, threads)
    # a file cut short after its index was built (same size and mtime faked): an error, not a short read
    if layout == "plain":
        cut = str(tmp_path / "cut.beagle.gz")
        raw = open(p, "rb").read()
        open(cut, "wb").write(raw[:len(raw) // 2] + bytes(len(raw) - len(raw) // 2))
        st_p = os.stat(p)
        os.utime(cut, ns=(st_p.st_atime_ns, st_p.st_mtime_ns))
        with pytest.raises(RuntimeError):
            with reader_cy.BeagleStream(cut, threads=4, index=idx, first_row=0) as st:
                list(st.chunks(max_rows=333))
    # an index of another file (or of an older version of this one) is refused
    os.utime(p, (1, 1))
    with pytest.raises(ValueError, match="is not an index of"):
        reader_cy.BeagleStream(p, index=idx, first_row=3)


def test_ensure_index_once_and_names_pass(tmp_path, monkeypatch):
    """read_site_names + the per-rank readers make ONE inflate pass over the file between them (the index
    pass); later opens reuse the cached index."""
    from wgsassign_amd import reader_cy
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path / "cache"))
    os.makedirs(tmp_path / "cache")
    L, _ = synth.make_beagle(900, 5, 1, seed=4)
    p = str(tmp_path / "y.beagle.gz")
    write_beagle(p, L)
    samples, names = reader_cy.read_site_names(p)
    assert samples == ["Ind%d" % i for i in range(5)] and names == ["chr1_%d" % (s + 1) for s in range(900)]
    idx, nam = reader_cy.index_paths(p)
    stamp = os.path.getmtime(idx)
    time.sleep(0.05)
    i2, _, sites = reader_cy.ensure_index(p)
    assert i2 == idx and sites == 900 and os.path.getmtime(idx) == stamp          # not rebuilt
    got = [r.copy() for r, _ in reader_cy.prefetched(reader_cy.BeagleStream(p, index=idx, first_row=450).chunks(max_rows=100))]
    assert np.concatenate(got).tobytes() == L[450:].tobytes()
    # errors inside the background producer reach the consumer
    def boom():
        yield 1
        raise RuntimeError("inside")
    it = reader_cy.prefetched(boom())
    assert next(it) == 1
    with pytest.raises(RuntimeError, match="inside"):
        next(it)


def test_header_with_a_partial_individual(tmp_path):
    """A header whose GL column count is not a multiple of 3: the reference keeps n = c // 3 individuals and the first
    2n values of each row (reader_cy.pyx:48-49, 71-75).  Every parsed row is exactly 2n floats -- the extra columns are
    not written anywhere (they used to spill into the next row and past the end of the last one)."""
    from wgsassign_amd import reader_cy
    n, m = 3, 200
    rng = np.random.default_rng(3)
    head = "marker\tallele1\tallele2\t" + "\t".join("S%d" % (i // 3) for i in range(3 * n + 2))     # 11 GL columns
    vals = np.round(rng.random((m, 3 * n + 2)), 6)
    p = str(tmp_path / "odd.beagle.gz")
    with gzip.open(p, "wt") as fh:
        fh.write(head + "\n")
        for s in range(m):
            fh.write("c_%d\tA\tC\t" % s + "\t".join("%.6f" % v for v in vals[s]) + "\n")
    guard = np.full((m + 1, 2 * n), -7.0, dtype=np.float32)
    with reader_cy.BeagleStream(p, threads=4) as st:
        assert st.n == n and st.sample_names == ["S0", "S1", "S2"]             # the complete individuals
        import ctypes
        from wgsassign_amd import _lib
        got = ctypes.c_int64()
        _lib.check(_lib.load().wgs_reader_next(st._h, _lib.f32p(guard), m, ctypes.byref(got)))
    assert got.value == m and np.all(guard[m] == -7.0)                           # nothing written past the last row
    want = np.stack([vals[:, 3 * i + j] for i in range(n) for j in (0, 1)], axis=1).astype(np.float32)
    assert guard[:m].tobytes() == want.tobytes()


def _bgzf_write(path, text, block=60000):
    """A real BGZF file (htslib layout): gzip members with the 'BC' extra subfield announcing their size, raw deflate
    payload, CRC32 + ISIZE trailer, and the empty end-of-file marker block."""
    import struct
    import zlib
    data = text.encode()
    with open(path, "wb") as fh:
        for i in list(range(0, len(data), block)) + [None]:
            chunk = b"" if i is None else data[i:i + block]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            payload = co.compress(chunk) + co.flush()
            bsize = 12 + 6 + len(payload) + 8
            fh.write(b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) +
                     b"BC" + struct.pack("<HH", 2, bsize - 1) + payload +
                     struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))


@pytest.mark.parametrize("variant", ["normal", "no_final_newline_blank_lines"])
def test_bgzf_parallel_index_and_block_parallel_inflate(tmp_path, monkeypatch, variant):
    """BGZF input (what ANGSD writes): the index pass inflates the blocks on all host threads and finds the same site
    count, header and line numbers as the serial pass over the same text; readers opened at any row inflate their
    blocks in parallel and return the same rows; gzip itself reads the file as ordinary multi-member gzip."""
    import ctypes
    from wgsassign_amd import _lib, reader_cy
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    m, n = 4000, 9
    L, _ = synth.make_beagle(m, n, 2, seed=77)
    text = _text_of(L, blank_lines="blank" in variant, final_newline="no_final" not in variant)
    p = str(tmp_path / "b.beagle.gz")
    _bgzf_write(p, text)
    assert gzip.open(p, "rt").read() == text
    lib = _lib.load()
    idx = str(tmp_path / "b.idx")
    sites = ctypes.c_int64()
    _lib.check(lib.wgs_reader_build_index(p.encode(), idx.encode(), None, 100_000, 4096, ctypes.byref(sites)))
    assert sites.value == m
    assert os.path.getsize(idx) < 20_000                      # block starts need no 32 KiB dictionaries
    names = ["ctg%d_%d" % (s % 7, s + 1) for s in range(m)]
    for first in [0, 1, 190, 191, 192, 1000, 2047, 3999]:
        with reader_cy.BeagleStream(p, threads=3, index=idx, first_row=first) as st:
            assert st.n == n
            got = list(st.chunks(max_rows=700))
        rows = np.concatenate([r for r, _ in got])
        sn = [x for _, ns in got for x in ns]
        assert rows.tobytes() == L[first:].tobytes() and sn == names[first:], first
    # without an index: from the start, block-parallel
    Lr, samples, sites_r = reader_cy.readBeagle(p)
    assert Lr.tobytes() == L.tobytes() and sites_r == names and samples == ["I%d" % i for i in range(n)]
    # the names pass (blocks inflated in parallel, first tokens scanned serially) agrees
    s2, n2 = reader_cy.read_site_names(p)
    assert n2 == names and s2 == samples
    # blocks are inflated by libdeflate when its shared library is installed, by zlib otherwise (or on request): same matrix
    if variant == "normal":
        import subprocess
        import sys
        from conftest import ROOT
        code = ("import sys, hashlib; sys.path.insert(0, %r); from wgsassign_amd import reader_cy; "
                "L, _, s = reader_cy.readBeagle(%r); print(hashlib.sha1(L.tobytes()).hexdigest(), len(s))" % (ROOT, p))
        import hashlib
        for backend in ("zlib", "libdeflate"):
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, WGSASSIGN_INFLATE=backend))
            assert r.returncode == 0, r.stderr[-2000:]
            assert r.stdout.split() == [hashlib.sha1(L.tobytes()).hexdigest(), str(m)], backend
    # a truncated file is an error, not a silent short read
    cut = str(tmp_path / "cut.beagle.gz")
    open(cut, "wb").write(open(p, "rb").read()[:-5000])
    with pytest.raises(RuntimeError):
        reader_cy.readBeagle(cut)


def test_reader_clean_under_address_and_thread_sanitizers():
    """csrc/reader.cpp is host code with worker threads (parallel parse, BGZF blocks, gzip stretches between access
    points): tools/reader_sanitize.sh builds it with -fsanitize=address,undefined and with -fsanitize=thread and drives
    the index pass, indexed opens at several rows and thread counts, and full reads whose checksums must agree."""
    import subprocess
    from conftest import ROOT
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "reader_sanitize.sh"), "3000", "20"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "clean under ASan+UBSan and TSan" in r.stdout and r.stdout.count("\nok\n") == 8
    assert "ERROR: AddressSanitizer" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr and "runtime error" not in r.stderr


# ------------------------------------------------------------------ text hand-over to the device tokeniser (reader_text.h)
def _text_rows(path, n, m_max, chunk_bytes, limit=-1, threads=3, index=None, first_row=0):
    import ctypes
    from wgsassign_amd import _lib, reader_cy
    with reader_cy.BeagleStream(path, threads=threads, index=index, first_row=first_row) as st:
        rows = np.full((m_max, 2 * n), -7.0, dtype=np.float32)
        got = ctypes.c_int64()
        lib = _lib.load()
        _lib.check(lib.wgs_debug_reader_text_rows(st._h, chunk_bytes, limit, _lib.f32p(rows), m_max, ctypes.byref(got)))
        nb = ctypes.c_int64()
        ptr = lib.wgs_reader_chunk_sites(st._h, ctypes.byref(nb))
        names = ctypes.string_at(ptr, nb.value).decode().split("\n")[:-1]
    return rows[:got.value], names


@pytest.mark.parametrize("layout", ["plain", "members", "plain_no_final_newline_blank_lines", "crlf"])
def test_text_hand_over_lists_the_same_rows(tmp_path, monkeypatch, layout):
    """The producer of the device ingest (inflate ahead, cut at the last newline, carry the partial line, list the
    non-blank lines in parallel, stop at the row limit) hands over exactly the lines wgs_reader_next parses: same
    rows, same names -- chunks far smaller than the file, from the start and from the middle through the index."""
    from wgsassign_amd import reader_cy
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    m, n = 9000, 37
    L, _ = synth.make_beagle(m, n, 2, seed=5)
    text = _text_of(L, blank_lines="blank" in layout, final_newline="no_final" not in layout)
    if layout == "crlf":
        text = text.replace("\n", "\r\n")
    p = str(tmp_path / ("t_%s.beagle.gz" % layout))
    if layout == "members":
        _bgzf_like(p, text)
    else:
        with gzip.open(p, "wt", newline="", compresslevel=6) as fh:
            fh.write(text)
    names = ["ctg%d_%d" % (s % 7, s + 1) for s in range(m)]
    rows, got = _text_rows(p, n, m, 1 << 20)                  # ~9 chunks of the 9 MB of text
    assert rows.shape[0] == m and rows.tobytes() == L.tobytes() and got == names
    rows, got = _text_rows(p, n, m, 1 << 20, limit=4321)
    assert rows.shape[0] == 4321 and rows.tobytes() == L[:4321].tobytes() and got == names[:4321]
    rows, got = _text_rows(p, n, m, 1 << 20, limit=0)
    assert rows.shape[0] == 0 and got == []
    idx, _, sites = reader_cy.ensure_index(p)
    assert sites == m
    for first, limit, threads in [(3333, 2000, 4), (8990, -1, 2), (1, 8999, 8)]:
        rows, got = _text_rows(p, n, m, 1 << 20, limit=limit, threads=threads, index=idx, first_row=first)
        want = L[first:] if limit < 0 else L[first:first + limit]
        assert rows.tobytes() == want.tobytes() and got == names[first:first + len(want)], (first, limit)


def test_text_hand_over_long_lines_grow_the_buffer(tmp_path):
    """A line longer than the chunk (n = 2000 individuals: 54 kB per line against 1 MiB chunks is fine, but a 3 MB
    line is not) makes the buffer grow instead of splitting the line."""
    n, m = 120_000, 3                                         # 3.2 MB per line
    rng = np.random.default_rng(3)
    L = (rng.integers(0, 1_000_001, size=(m, 2 * n)) / 1e6).astype(np.float32)
    head = "m a b " + " ".join("S S S" for _ in range(n))
    p = str(tmp_path / "long.beagle.gz")
    with gzip.open(p, "wt", compresslevel=1) as fh:
        fh.write(head + "\n")
        for s in range(m):
            fh.write("s%d A C " % s + " ".join("%.6f %.6f 0" % (L[s, 2 * i], L[s, 2 * i + 1]) for i in range(n)) + "\n")
    rows, got = _text_rows(p, n, m, 1 << 20)
    assert rows.tobytes() == L.tobytes() and got == ["s0", "s1", "s2"]


def test_decimal_tokens_with_exponents_take_the_exact_path(tmp_path):
    """[sign] digits [. digits] [e [sign] digits] with <= 15 significant digits and a net power of ten within +-22 is
    converted by one correctly rounded operation (host: parse_double; device: ingest.hip parse_token) -- float()'s
    value, which is strtod's; everything else goes to strtod itself."""
    from wgsassign_amd import reader_cy
    rng = np.random.default_rng(11)
    toks = ["1e-3", "1.5E+3", "-2.5e-7", "123456789012345e-22", "0.000001e6", "9.99999e22", "1e23", "1e-23", "4.9e-324",
            "1e", "1e+", "1.2.3", "-.", ".", "+.5e1", "0e999", "00012.5000e-01", "1e0005", "7E-0", "123456789012345678e-3"]
    for _ in range(300):
        digs = int(rng.integers(1, 16))
        mant = "".join(str(int(d)) for d in rng.integers(0, 10, size=digs))
        dot = int(rng.integers(0, digs + 1))
        t = mant[:dot] + "." + mant[dot:] if rng.random() < 0.8 else mant
        if rng.random() < 0.7:
            t += "eE"[int(rng.integers(0, 2))] + ["", "+", "-"][int(rng.integers(0, 3))] + str(int(rng.integers(0, 40)))
        toks.append(["", "-", "+"][int(rng.integers(0, 3))] + t)
    if len(toks) % 2:
        toks.append("0.5")
    n = len(toks) // 2
    p = str(tmp_path / "exp.beagle.gz")
    with gzip.open(p, "wt") as fh:
        fh.write("m a b " + " ".join("S S S" for _ in range(n)) + "\n")
        fh.write("s1 A C " + " ".join("%s %s 0" % (toks[2 * i], toks[2 * i + 1]) for i in range(n)) + "\n")

    def atof(t):                                              # C's atof: the longest valid prefix, 0.0 if there is none
        import re
        mm = re.match(r"[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)", t)
        return float(mm.group(0)) if mm else 0.0
    with np.errstate(over="ignore"):
        want = np.array([atof(t) for t in toks], dtype=np.float64).astype(np.float32)
    L, _, _ = reader_cy.readBeagle(p)
    assert L.shape == (1, 2 * n)
    bad = [(t, float(a), float(b)) for t, a, b in zip(toks, L[0], want) if a.tobytes() != b.tobytes()]
    assert not bad, bad[:5]


def test_text_hand_over_really_chunks(tmp_path, monkeypatch):
    """What the line-oriented calls had already inflated when the hand-over starts (up to 64 MiB) is cut into chunks
    like everything after it: 1 MiB chunks over 9 MB of text are at least nine chunks (seen through the debug hook's
    row order check and the wall clock of neither: counted here by the number of distinct first rows)."""
    import ctypes
    from wgsassign_amd import _lib, reader_cy
    m, n = 9000, 37
    L, _ = synth.make_beagle(m, n, 2, seed=5)
    p = str(tmp_path / "c.beagle.gz")
    with gzip.open(p, "wt", newline="", compresslevel=1) as fh:
        fh.write(_text_of(L))
    lib = _lib.load()
    with reader_cy.BeagleStream(p, threads=2) as st:
        rows = np.empty((m, 2 * n), dtype=np.float32)
        got, chunks = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(lib.wgs_debug_reader_text_rows(st._h, 1 << 20, -1, _lib.f32p(rows), m, ctypes.byref(got)))
        assert got.value == m and rows.tobytes() == L.tobytes()
        assert lib.wgs_debug_reader_text_chunks(st._h) >= 9


def test_index_cache_is_private_and_checked(tmp_path, monkeypatch):
    """The cache of indices and site names: a per-user 0700 directory by default, files created exclusively (a planted
    symlink is not followed), read only if they are regular files of this user; a names file with the wrong number of
    lines (a run killed mid-write in an earlier version) is rebuilt, a same-size rewrite of the Beagle file is noticed
    (modification time in nanoseconds)."""
    from wgsassign_amd import reader_cy
    monkeypatch.delenv("WGSASSIGN_INDEX_DIR", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    d = reader_cy.cache_dir()
    assert d == str(tmp_path / "cache" / "wgsassign") and (os.stat(d).st_mode & 0o777) == 0o700
    L, _ = synth.make_beagle(300, 5, 1, seed=2)
    p = str(tmp_path / "a.beagle.gz")
    write_beagle(p, L)
    idx, nam = reader_cy.index_paths(p)
    # a symlink planted where the temporary file would go must not be written through
    victim = tmp_path / "victim"
    victim.write_text("precious")
    os.symlink(str(victim), idx + ".tmp.%d.0" % os.getpid())
    samples, names = reader_cy.read_site_names(p)
    assert victim.read_text() == "precious" and names == ["chr1_%d" % (s + 1) for s in range(300)]
    assert (os.stat(idx).st_mode & 0o777) == 0o600 and (os.stat(nam).st_mode & 0o777) == 0o600
    # truncated names file: rebuilt, not trusted
    open(nam, "wb").write(open(nam, "rb").read()[:100])
    assert reader_cy.read_site_names(p)[1] == names
    # an index that is a symlink (to a perfectly good index) is not read
    real = idx + ".real"
    os.rename(idx, real)
    os.symlink(real, idx)
    assert not reader_cy._private_file(idx)
    os.remove(idx)
    os.rename(real, idx)
    # same size, same second, different content: the key changes with the nanoseconds
    st = os.stat(p)
    L2 = L.copy()
    L2[0, 0] = np.float32(0.5) if L[0, 0] != np.float32(0.5) else np.float32(0.25)
    write_beagle(p, L2)
    os.utime(p, ns=(st.st_atime_ns, st.st_mtime_ns + 1000))
    assert reader_cy.index_paths(p)[0] != idx
    # a cache directory others can write to is refused
    bad = tmp_path / "shared"
    bad.mkdir(mode=0o777)
    os.chmod(str(bad), 0o777)
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(bad))
    with pytest.raises(RuntimeError, match="not a private directory"):
        reader_cy.cache_dir()


# ------------------------------------------------------------------ the BGZF index pass in parts
def _index_fields(idx_path, path):
    """(sites, first rows readable through the index at a few positions) -- what an index is for"""
    import ctypes
    from wgsassign_amd import _lib, reader_cy
    n = ctypes.c_int64()
    _lib.check(_lib.load().wgs_reader_index_sites(os.fsencode(path), os.fsencode(idx_path), ctypes.byref(n)))
    return n.value


@pytest.mark.parametrize("block", [60000, 700])
def test_bgzf_index_in_parts_equals_the_single_pass(tmp_path, monkeypatch, block):
    """The index pass split into byte ranges (ranks of a node, threads of a rank): every part finds its first block by the
    BGZF signature, the merge chains them -- same site count, same access points as the single pass; readers opened through
    the merged index return the same rows.  Tiny blocks (700 bytes of text each) make thousands of hops and resyncs."""
    import ctypes
    from wgsassign_amd import _lib, reader_cy
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    m, n = 6000, 19
    L, _ = synth.make_beagle(m, n, 2, seed=13)
    p = str(tmp_path / "p.beagle.gz")
    _bgzf_write(p, _text_of(L, blank_lines=True), block=block)
    lib = _lib.load()
    one = str(tmp_path / "one.idx")
    sites = ctypes.c_int64()
    _lib.check(lib.wgs_reader_build_index(p.encode(), one.encode(), None, 40_000, 100_000, ctypes.byref(sites)))
    assert sites.value == m
    for nparts, threads in [(1, 1), (2, 3), (3, 1), (5, 2), (8, 4)]:
        prefix = str(tmp_path / ("parts_%d_%d" % (nparts, threads)))
        for k in range(nparts):
            _lib.check(lib.wgs_reader_index_part(p.encode(), ("%s.%d" % (prefix, k)).encode(), k, nparts, threads))
        merged = str(tmp_path / ("m_%d_%d.idx" % (nparts, threads)))
        got = ctypes.c_int64()
        _lib.check(lib.wgs_reader_index_merge(p.encode(), merged.encode(), prefix.encode(), nparts, 40_000, 100_000, ctypes.byref(got)))
        assert got.value == m
        assert open(merged, "rb").read() == open(one, "rb").read(), (nparts, threads)     # the very same index file
        assert not os.path.exists(prefix + ".0")                                          # parts are removed after the merge
    for first in (0, 1234, 5999):
        with reader_cy.BeagleStream(p, threads=2, index=merged, first_row=first) as st:
            rows, _ = next(st.chunks(max_rows=50))
        assert rows.tobytes() == L[first:first + 50].tobytes()


def test_bgzf_parts_reject_a_planted_signature_and_a_plain_gzip_file(tmp_path):
    """A block whose (stored) payload contains a chain of look-alike BGZF headers: a part that resynchronises onto it
    starts off the true chain; the merge notices (the parts do not chain) and reports rc 3, so the caller falls back to the
    serial pass.  Plain gzip cannot be done in parts at all."""
    import ctypes
    import struct
    import zlib
    from wgsassign_amd import _lib, reader_cy

    def member(chunk, level):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = co.compress(chunk) + co.flush()
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(payload) + 8 - 1) +
                payload + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    head = "marker a b " + " ".join("S S S" for _ in range(3)) + "\n"
    line = lambda s: "s%d A C " % s + " ".join("0.250000 0.500000 0.250000" for _ in range(3)) + "\n"
    # four fake members back to back, each announcing a size that leads exactly to the next: inside ONE stored block
    fake = b"".join(member(b"x" * 40, 0) for _ in range(6))
    text_before = (head + "".join(line(s) for s in range(300))).encode()
    evil_line = b"evil" + b"x" * 600 + b" A C " + fake.replace(b"\n", b"_").replace(b" ", b"_").replace(b"\t", b"_") + b" 0.1 0.2 0.7 0.1 0.2 0.7 0.1 0.2 0.7\n"
    text_after = "".join(line(s) for s in range(300, 900)).encode()
    p = str(tmp_path / "evil.beagle.gz")
    with open(p, "wb") as fh:
        for i in range(0, len(text_before), 3000):
            fh.write(member(text_before[i:i + 3000], 6))
        fh.write(member(evil_line, 0))                     # level 0 = stored: the look-alike headers appear verbatim in the file
        for i in range(0, len(text_after), 3000):
            fh.write(member(text_after[i:i + 3000], 6))
        fh.write(member(b"", 6))
    assert fake.replace(b"\n", b"_").replace(b" ", b"_").replace(b"\t", b"_") == fake     # the fakes survived the token rules
    lib = _lib.load()
    L, _, sites = reader_cy.readBeagle(p)
    assert len(sites) == 901 and sites[300] == "evil" + "x" * 600
    size = os.path.getsize(p)
    raw = open(p, "rb").read()
    evil_at = raw.index(fake)
    # choose a split whose second part starts just before the planted chain
    found_bad = False
    for nparts in range(2, 40):
        lo = [size * k // nparts for k in range(nparts)]
        if not any(evil_at - 580 < x <= evil_at for x in lo):         # inside the evil member, before the planted chain
            continue
        prefix = str(tmp_path / ("e%d" % nparts))
        rcs = [lib.wgs_reader_index_part(p.encode(), ("%s.%d" % (prefix, k)).encode(), k, nparts, 1) for k in range(nparts)]
        # the part that landed on the planted chain either runs into garbage behind it (rc 3 at once) or delivers blocks
        # that do not chain with its neighbours (rc 3 from the merge): never a wrong index
        assert all(r in (0, 3) for r in rcs)
        if all(r == 0 for r in rcs):
            got = ctypes.c_int64()
            rc = lib.wgs_reader_index_merge(p.encode(), str(tmp_path / "e.idx").encode(), prefix.encode(), nparts, 40_000, 1000, ctypes.byref(got))
            assert rc == 3 and b"do not chain" in lib.wgs_last_error()
        found_bad = True
        break
    assert found_bad
    # the single-process pass (threads resynchronise too) still gets it right: its fall-back is the serial hop
    n = ctypes.c_int64()
    _lib.check(lib.wgs_reader_build_index(p.encode(), str(tmp_path / "ok.idx").encode(), None, 40_000, 1000, ctypes.byref(n)))
    assert n.value == 901
    # plain gzip: rc 3
    g = str(tmp_path / "plain.beagle.gz")
    with gzip.open(g, "wb") as fh:
        fh.write(text_before)
    assert lib.wgs_reader_index_part(g.encode(), str(tmp_path / "g.0").encode(), 0, 2, 1) == 3


def test_damaged_part_files_are_refused_and_removed(tmp_path):
    """A part file of the split index pass whose block count or header length has been damaged (the counts sized
    two vectors before being checked against the file -- std::bad_alloc through an extern "C" call): the merge answers rc 3 with
    a message, and the part files are gone afterwards whether the merge succeeded or not."""
    import ctypes
    import struct
    from wgsassign_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    n, m = 6, 400
    head = "marker\tallele1\tallele2\t" + "\t".join("I%d\tI%d\tI%d" % (i, i, i) for i in range(n))
    lines = [head] + ["s%d\tA\tC\t" % s + "\t".join("%.6f" % v for v in rng.random(3 * n)) for s in range(m)]
    p = str(tmp_path / "d.beagle.gz")
    _bgzf_write(p, "\n".join(lines) + "\n", block=900)
    nparts = 3
    for damage in ("blocks", "header", "truncate", None):
        prefix = str(tmp_path / ("part_%s" % damage))
        for k in range(nparts):
            assert lib.wgs_reader_index_part(p.encode(), ("%s.%d" % (prefix, k)).encode(), k, nparts, 1) == 0
        victim = "%s.1" % prefix
        raw = bytearray(open(victim, "rb").read())
        # layout: magic[8], file size, mtime, first, next, nb, hl (six uint64)
        if damage == "blocks":
            raw[8 + 4 * 8:8 + 5 * 8] = struct.pack("<Q", (1 << 39) + 7)
        elif damage == "header":
            raw[8 + 5 * 8:8 + 6 * 8] = struct.pack("<Q", (1 << 29) + 3)
        elif damage == "truncate":
            raw = raw[:len(raw) - 5]
        if damage:
            os.chmod(victim, 0o600)
            with open(victim, "wb") as fh:
                fh.write(bytes(raw))
        got = ctypes.c_int64()
        rc = lib.wgs_reader_index_merge(p.encode(), str(tmp_path / ("m_%s.idx" % damage)).encode(), prefix.encode(), nparts, 40_000, 1000, ctypes.byref(got))
        if damage:
            assert rc == 3 and b"is not a part of the index" in lib.wgs_last_error()
        else:
            assert rc == 0 and got.value == m
        assert not [f for f in os.listdir(tmp_path) if f.startswith("part_%s." % damage)]


# ------------------------------------------------------------------ the compressed hand-over (device-resident BGZF ingest)
def _comp_text(path, comp_bytes, text_cap, nbuf=2, threads=3, index=None, first_row=0, cap=64 << 20):
    import ctypes
    from wgsassign_amd import _lib, reader_cy
    with reader_cy.BeagleStream(path, threads=threads, index=index, first_row=first_row) as st:
        buf = ctypes.create_string_buffer(cap)
        nbytes = ctypes.c_int64()
        info = (ctypes.c_int64 * 4)()
        _lib.check(_lib.load().wgs_debug_reader_comp_text(st._h, comp_bytes, text_cap, nbuf, buf, cap, ctypes.byref(nbytes), info))
        return buf.raw[:nbytes.value], list(info)


@pytest.mark.parametrize("block", [900, 60000])
def test_compressed_hand_over_delivers_the_rest_of_the_file(tmp_path, monkeypatch, block):
    """What the device-resident ingest consumes (reader.cpp: comp_producer): the text the header calls had inflated
    already, then whole BGZF members in staging buffers -- together exactly the file's text behind the header, whatever
    the staging and text limits (one buffer or two, slices that end inside members, limits below one slice), from the
    start and from a row in the middle through the index."""
    from wgsassign_amd import reader_cy
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    m, n = 6000, 41
    L, _ = synth.make_beagle(m, n, 2, seed=9)
    text = _text_of(L, blank_lines=True, final_newline=False)
    p = str(tmp_path / "c.beagle.gz")
    synth.write_bgzf(p, text.encode(), block=block)
    body = text.split("\n", 1)[1].encode()
    for comp_bytes, text_cap, nbuf in [(1 << 20, 1 << 20, 2), (1 << 20, 3 << 20, 1), (4 << 20, 1 << 20, 3), (64 << 20, 64 << 20, 1)]:
        got, info = _comp_text(p, comp_bytes, text_cap, nbuf)
        assert got == body, (comp_bytes, text_cap)
        assert info[2] <= max(text_cap, 1 << 20) and info[0] >= len(body) // max(text_cap, 1 << 20)
    idx, _, sites = reader_cy.ensure_index(p)
    assert sites == m
    lines = body.split(b"\n")
    data_lines = [i for i, x in enumerate(lines) if x.strip()]
    for first in (1, 2999, m - 1):
        got, info = _comp_text(p, 1 << 20, 1 << 20, 2, index=idx, first_row=first)
        want = b"\n".join(lines[data_lines[first]:])
        assert got.endswith(want) and len(got) - len(want) < 70000, first      # a reader opened at a row starts at its member
        assert got[len(got) - len(want) - 1:len(got) - len(want)] in (b"\n", b"")


def test_compressed_hand_over_reports_damaged_files(tmp_path, monkeypatch):
    """A truncated file and a member whose header was overwritten -- beyond what the reader inflates while it reads the
    header -- are errors of the hand-over, not silent ends; the undamaged file comes through whole."""
    import gzip as gz
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import beagle_files
    monkeypatch.setenv("WGSASSIGN_INDEX_DIR", str(tmp_path))
    p = str(tmp_path / "d.beagle.gz")
    total, _, _ = beagle_files.write_lowdepth_bgzf(p, 500, 12000, pool=64)       # 162 MB of text; the open call inflates ~1 MB of it on the host
    got, info = _comp_text(p, 2 << 20, 8 << 20, threads=2, cap=200 << 20)
    assert len(got) > 150 << 20 and got == gz.open(p).read()[-len(got):] and info[3] >= 1 and info[0] > 10
    raw = open(p, "rb").read()
    cut = str(tmp_path / "cut.beagle.gz")
    with open(cut, "wb") as fh:
        fh.write(raw[:len(raw) - 5000])
    with pytest.raises(RuntimeError, match="corrupt or truncated"):
        _comp_text(cut, 2 << 20, 8 << 20, threads=2, cap=200 << 20)
    bad = bytearray(raw)
    k = raw.find(b"\x1f\x8b\x08\x04", len(raw) * 3 // 4)
    bad[k + 1] = 0x00                                             # not a gzip member any more
    dmg = str(tmp_path / "dmg.beagle.gz")
    with open(dmg, "wb") as fh:
        fh.write(bytes(bad))
    with pytest.raises(RuntimeError, match="corrupt or truncated"):
        _comp_text(dmg, 2 << 20, 8 << 20, threads=2, cap=200 << 20)


@pytest.mark.parametrize("block", [700, 60000])
def test_site_estimate_of_a_bgzf_file(tmp_path, block):
    """wgs_reader_estimate_sites (what sizes the device matrix of a file met for the first time, before its index pass has counted
    the sites): five samples of a quarter megabyte, newlines per compressed byte times the file size -- within a few per cent of
    the count on homogeneous text, whatever the block size; nothing to say about a plain gzip file (rc 3 -> None)."""
    import gzip
    from wgsassign_amd import reader_cy
    m, n = 9000, 23
    L, _ = synth.make_beagle(m, n, 2, seed=29)
    text = _text_of(L)
    p = str(tmp_path / "e.beagle.gz")
    _bgzf_write(p, text, block=block)
    est = reader_cy.estimate_sites(p)
    assert est is not None and abs(est - m) <= m // 20, est
    assert reader_cy.count_sites(p) == m
    q = str(tmp_path / "plain.beagle.gz")
    with gzip.open(q, "wb") as fh:
        fh.write(text.encode())
    assert reader_cy.estimate_sites(q) is None
    tiny = str(tmp_path / "tiny.beagle.gz")                       # smaller than one sample: the first sample is all of it
    _bgzf_write(tiny, _text_of(L[:40]), block=block)
    t = reader_cy.estimate_sites(tiny)
    assert t is not None and 35 <= t <= 48, t
