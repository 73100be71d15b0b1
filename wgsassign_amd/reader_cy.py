"""Beagle reader: drop-in for `reader_cy.readBeagle` (reader_cy.pyx:16-77), backed by the
native streamed reader of libwgsassign_hip.so (csrc/reader.cpp; host code, needs no GPU).

Header: every 3rd token after the first three is a sample name; per line token 0 is the site
name, two allele columns are skipped, GL0 and GL1 are kept and GL2 dropped; values are
atof(token) rounded to float32.
"""
import ctypes
import gzip
import os

import numpy as np

from . import _lib


class BeagleStream:
    """Chunked reader: iterate (rows float32 (k, 2n), site_names list) until the file ends."""

    def __init__(self, path, threads=None):
        lib = _lib.load()
        if threads is None:
            threads = min(len(os.sched_getaffinity(0)), 16)
        h = ctypes.c_void_p()
        _lib.check(lib.wgs_reader_open(os.fsencode(path), int(threads), ctypes.byref(h)))
        self._h = h
        self.n = lib.wgs_reader_n_individuals(h)
        self.sample_names = [lib.wgs_reader_sample_name(h, i).decode() for i in range(self.n)]

    def skip(self, nrows):
        """Skip nrows sites without parsing; returns the number actually skipped."""
        got = ctypes.c_int64()
        _lib.check(_lib.load().wgs_reader_skip(self._h, int(nrows), ctypes.byref(got)))
        return got.value

    def chunks(self, max_rows=None, target_bytes=None, limit=None):
        """Yield (rows, site_names) chunks; at most `limit` sites in total when given.  Chunk size:
        max_rows sites, else target_bytes (default 256 MiB, or WGSASSIGN_CHUNK_BYTES) of float32."""
        lib = _lib.load()
        if target_bytes is None:
            target_bytes = int(os.environ.get("WGSASSIGN_CHUNK_BYTES", 256 << 20))
        if max_rows is None:
            max_rows = max(1, target_bytes // max(1, 8 * self.n))
        left = limit
        while True:
            want = max_rows if left is None else min(max_rows, left)
            if want <= 0:
                return
            rows = np.empty((want, 2 * self.n), dtype=np.float32)
            got = ctypes.c_int64()
            _lib.check(lib.wgs_reader_next(self._h, _lib.f32p(rows), want, ctypes.byref(got)))
            if left is not None:
                left -= got.value
            if got.value == 0:
                return
            nbytes = ctypes.c_int64()
            ptr = lib.wgs_reader_chunk_sites(self._h, ctypes.byref(nbytes))
            names = ctypes.string_at(ptr, nbytes.value).decode().split("\n")[:-1]
            yield rows[:got.value], names

    def close(self):
        if self._h:
            _lib.load().wgs_reader_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def count_sites(path):
    n = ctypes.c_int64()
    _lib.check(_lib.load().wgs_reader_count_sites(os.fsencode(path), ctypes.byref(n)))
    return n.value


def read_site_names(path, chunk=1 << 20):
    """(sample_names, site_names) of a Beagle file without parsing any likelihood (inflate-bound)."""
    lib = _lib.load()
    names = []
    with BeagleStream(path, threads=1) as st:
        while True:
            got = ctypes.c_int64()
            _lib.check(lib.wgs_reader_skip_names(st._h, chunk, ctypes.byref(got)))
            if got.value == 0:
                break
            nbytes = ctypes.c_int64()
            ptr = lib.wgs_reader_chunk_sites(st._h, ctypes.byref(nbytes))
            names.extend(ctypes.string_at(ptr, nbytes.value).decode().split("\n")[:-1])
        return list(st.sample_names), names


def readBeagle(beagle):
    """reader_cy.pyx:16-77: returns (L float32 (m, 2n) C-contiguous, sample_names, site_names)."""
    with BeagleStream(beagle) as st:
        parts, site_names = [], []
        for rows, names in st.chunks():
            parts.append(rows.copy() if rows.base is not None and rows.shape[0] * 4 < rows.base.shape[0] else rows)
            site_names.extend(names)
        if parts:
            L = np.ascontiguousarray(np.concatenate(parts, axis=0)) if len(parts) > 1 else np.ascontiguousarray(parts[0])
        else:
            L = np.empty((0, 2 * st.n), dtype=np.float32)
        return L, list(st.sample_names), site_names


def readBeagle_py(beagle):
    """Pure-Python restatement of the same rules (slow; kept as a cross-check of the native reader)."""
    with gzip.open(beagle, "rb") as fh:
        header = fh.readline().split()
        sample_names = [t.decode() for t in header[3::3]]
        n = len(header[3:]) // 3
        site_names, chunks = [], []
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            site_names.append(tok[0].decode())
            gl = np.array(tok[3:3 + 3 * n], dtype=np.float64).reshape(n, 3)
            chunks.append(gl[:, :2].astype(np.float32).reshape(-1))
    L = np.ascontiguousarray(np.array(chunks, dtype=np.float32).reshape(len(chunks), 2 * n))
    return L, sample_names, site_names


def stream_to_device(path, group_of=None, n_groups=1, ctx=None, threads=None, rank=0, world=1, m_total=None,
                     keep=None):
    """Two passes over the file: count the sites, then parse chunk by chunk straight into the
    device slabs -- host memory stays bounded by one chunk (the reference holds two
    full copies).  With world > 1 this rank skips to its contiguous SNP range
    (comm.shard_range) and parses only that.  group_of may be a callable(sample_names) ->
    (group_of, n_groups).  keep (bool array over the file's sites, e.g. the site mask of
    utils.filter_sites_to_common) restricts the matrix to the kept sites; shard ranges then count
    kept sites.  Returns (DeviceBeagle, sample_names, site_names of the range, m_total)."""
    from .comm import shard_range
    from .device import DeviceBeagle
    if keep is not None:
        keep = np.asarray(keep, dtype=bool)
        kept_rows = np.flatnonzero(keep)
        m_total = len(kept_rows)
    elif m_total is None:
        m_total = count_sites(path)
    lo, hi = shard_range(m_total, rank, world)
    # file rows [r0, r1) cover this rank's (kept) sites
    if keep is None:
        r0, r1 = lo, hi
    elif hi > lo:
        r0, r1 = int(kept_rows[lo]), int(kept_rows[hi - 1]) + 1
    else:
        r0 = r1 = 0
    with BeagleStream(path, threads) as st:
        if callable(group_of):
            group_of, n_groups = group_of(list(st.sample_names))
        if hi <= lo:
            raise ValueError("fewer sites (%d) than ranks (%d): nothing to shard" % (m_total, world))
        beagle = DeviceBeagle(hi - lo, st.n, group_of, n_groups, site0=lo, ctx=ctx)
        if st.skip(r0) != r0:
            raise RuntimeError("Beagle file shorter than counted")
        row0, frow, site_names = 0, r0, []
        for rows, names in st.chunks(limit=r1 - r0):
            if keep is not None:
                sel = keep[frow:frow + rows.shape[0]]
                frow += rows.shape[0]
                rows = rows[sel]
                names = [x for x, k in zip(names, sel) if k]
            if rows.shape[0]:
                beagle.upload_rows(np.ascontiguousarray(rows), row0)
            row0 += rows.shape[0]
            site_names.extend(names)
        if row0 != hi - lo:
            raise RuntimeError("Beagle file changed while reading: expected %d sites, parsed %d" % (hi - lo, row0))
        return beagle, list(st.sample_names), site_names, m_total
