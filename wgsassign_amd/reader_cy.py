"""Beagle reader: drop-in for `reader_cy.readBeagle` (reader_cy.pyx:16-77), backed by the
native streamed reader of libwgsassign_hip.so (csrc/reader.cpp; host code, needs no GPU).

Header: every 3rd token after the first three is a sample name; per line token 0 is the site
name, two allele columns are skipped, GL0 and GL1 are kept and GL2 dropped; values are
atof(token) rounded to float32.
"""
import ctypes
import gzip
import hashlib
import os
import queue
import tempfile
import threading

import numpy as np

from . import _lib

INDEX_SPAN_BYTES = 16 << 20       # text between two access points of the index (= one unit of parallel inflate)
INDEX_MAX_POINTS = 4096

_requested_threads = None         # -t / --threads of the command line (set_threads), when it asks for more than one


def set_threads(t):
    """-t/--threads of the command line (WGSassign.py:28-29, default 1): a value above 1 is the host-thread budget of
    this NODE's job -- inflate and newline scan of the reader; 1 (the reference's default) leaves the choice to
    host_threads()."""
    global _requested_threads
    _requested_threads = int(t) if t and int(t) > 1 else None


def host_threads():
    """Host threads ONE rank gives its reader: the node's budget -- WGSASSIGN_THREADS, else -t, else the CPUs this
    process can keep busy (comm.usable_cpus: affinity cut down to the control group's CPU quota; at most 32 per rank)
    -- divided by the ranks sharing the node (LOCAL_WORLD_SIZE, exported by torchrun and by comm.launch_local_ranks):
    eight ranks on a 128-thread node take 16 each instead of 8 x 16 on the one-GPU share."""
    local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    env = os.environ.get("WGSASSIGN_THREADS")
    if env:
        return max(1, int(env) // local)
    if _requested_threads:
        return max(1, _requested_threads // local)
    from .comm import usable_cpus
    return max(1, min(usable_cpus() // local, 32))


class BeagleStream:
    """Chunked reader: iterate (rows float32 (k, 2n), site_names list) until the file ends."""

    def __init__(self, path, threads=None, index=None, first_row=0):
        """From the first site, or -- given the index of wgs_reader_build_index -- positioned at `first_row`
        without inflating the file up to there."""
        lib = _lib.load()
        if threads is None:
            threads = host_threads()
        self.threads = int(threads)
        self._ingest = None
        h = ctypes.c_void_p()
        if index is None:
            _lib.check(lib.wgs_reader_open(os.fsencode(path), int(threads), ctypes.byref(h)))
            if first_row:
                got = ctypes.c_int64()
                _lib.check(lib.wgs_reader_skip(h, int(first_row), ctypes.byref(got)))
                if got.value != first_row:
                    raise RuntimeError("Beagle file shorter than counted")
        else:
            _lib.check(lib.wgs_reader_open_indexed(os.fsencode(path), os.fsencode(index), int(first_row), int(threads),
                                                   ctypes.byref(h)))
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

    def ingest(self, beagle, row0=0, limit=None, keep=None, chunk_bytes=None, names="all"):
        """Device-side ingest (csrc/ingest.hip): the rest of the file -- at most `limit` sites -- goes as TEXT to the
        GPU, which tokenises it straight into the slabs of `beagle` from row `row0` on.  keep: bool array over those
        sites (False = the site takes no row).  Yields (rows_written, site_names of the chunk's kept sites); names="ends" (and
        no keep mask): only the first and the last four names of a chunk of more than eight sites."""
        lib = _lib.load()
        if chunk_bytes is None:
            chunk_bytes = int(os.environ.get("WGSASSIGN_TEXT_CHUNK_BYTES", 0))
        g = ctypes.c_void_p()
        _lib.check(lib.wgs_ingest_create(beagle.handle, self._h, -1 if limit is None else int(limit), int(chunk_bytes),
                                         ctypes.byref(g)))
        self._ingest = g
        if keep is not None:
            keep = np.ascontiguousarray(keep, dtype=np.uint8)
        consumed = 0
        try:
            while True:
                nfile, nrows = ctypes.c_int64(), ctypes.c_int64()
                kp = ctypes.c_void_p(keep.ctypes.data + consumed) if keep is not None else None
                _lib.check(lib.wgs_ingest_next(g, int(row0), kp, (len(keep) - consumed) if keep is not None else 0,
                                               ctypes.byref(nfile), ctypes.byref(nrows)))
                if nfile.value == 0:
                    break
                nbytes = ctypes.c_int64()
                ptr = lib.wgs_ingest_chunk_sites(g, ctypes.byref(nbytes))
                raw = ctypes.string_at(ptr, nbytes.value)
                if names == "ends" and keep is None and nfile.value > 8:
                    # only the first and the last four names of the chunk are wanted (stream_to_device, names="ends"): a Python string
                    # per site is 10 ms per 100 000 sites, as much as the device needs for them
                    head = raw.split(b"\n", 4)[:4]
                    tail = raw[-4096:].split(b"\n")[-5:-1] if raw.count(b"\n", -4096) >= 5 else raw.split(b"\n")[-5:-1]
                    names_out = [x.decode() for x in head + tail]
                else:
                    names_out = raw.decode().split("\n")[:-1]
                    if keep is not None:
                        names_out = [x for x, k in zip(names_out, keep[consumed:consumed + nfile.value]) if k]
                consumed += nfile.value
                row0 += nrows.value
                yield nrows.value, names_out
            st = (ctypes.c_double * 14)()
            _lib.check(lib.wgs_ingest_stats(g, st))
            self.ingest_stats = dict(zip(("wait_s", "inflate_s", "scan_s", "device_ms", "host_lines", "text_bytes", "lines",
                                          "chunks", "device_inflate_kernel_ms", "blocks_inflated_on_device",
                                          "blocks_left_to_host_inflater", "read_s", "create_s", "next_s"), st))
        finally:
            if self._ingest is not None:     # close() may have destroyed it already
                self._ingest = None
                lib.wgs_ingest_destroy(g)

    def close(self):
        if self._h:
            if self._ingest:
                _lib.load().wgs_ingest_destroy(self._ingest)
                self._ingest = None
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


def estimate_sites(path):
    """About how many sites a BGZF Beagle file holds (five samples of a quarter megabyte: milliseconds), or None where that
    cannot be said (not BGZF)."""
    n = ctypes.c_int64()
    rc = _lib.load().wgs_reader_estimate_sites(os.fsencode(path), ctypes.byref(n))
    return n.value if rc == 0 and n.value > 0 else None


def cache_dir():
    """Where indices and site-name lists are cached: WGSASSIGN_INDEX_DIR, else a per-user directory
    ($XDG_CACHE_HOME/wgsassign or ~/.cache/wgsassign; the per-user temporary directory when there is no home), created
    with mode 0700.  A directory that belongs to somebody else, or that others may write to, is refused -- the access
    points and line numbers of an index are trusted by every later run."""
    d = os.environ.get("WGSASSIGN_INDEX_DIR")
    if not d:
        base = os.environ.get("XDG_CACHE_HOME") or (os.path.join(os.path.expanduser("~"), ".cache") if os.path.expanduser("~") != "~" else None)
        d = os.path.join(base, "wgsassign") if base else os.path.join(tempfile.gettempdir(), "wgsassign-%d" % os.geteuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.geteuid() or (st.st_mode & 0o022 and not st.st_mode & 0o1000):
        raise RuntimeError("index cache directory %s is not a private directory of this user (set WGSASSIGN_INDEX_DIR)" % d)
    return d


def index_paths(path):
    """Index and site-name cache files of a Beagle file, keyed by its absolute path, size and modification time (ns)."""
    st = os.stat(path)
    key = hashlib.sha1(("%s|%d|%d" % (os.path.abspath(path), st.st_size, st.st_mtime_ns)).encode()).hexdigest()[:20]
    base = os.path.join(cache_dir(), "wgsassign_" + key)
    return base + ".idx", base + ".names"


def _private_file(path):
    """A regular file (not a symlink) of this user."""
    try:
        st = os.lstat(path)
    except OSError:
        return False
    import stat as _stat
    return _stat.S_ISREG(st.st_mode) and st.st_uid == os.geteuid()


def _count_lines(path):
    n = 0
    with open(path, "rb") as fh:
        while True:
            buf = fh.read(16 << 20)
            if not buf:
                return n
            n += buf.count(b"\n")


def ensure_index(path, comm=None, names=False):
    """ONE inflate pass per file and node: the first rank OF EVERY NODE (LOCAL_RANK 0; the cache directory is node-local)
    counts the sites, records the access points (and the site names when asked) in the index cache; the other ranks
    wait and read the result.  A cached index is used only if it still describes the file (size, mtime in ns), belongs to
    this user, and -- when names are wanted -- comes with a names file holding exactly one name per site.
    Returns (index_path, names_path or None, sites)."""
    lib = _lib.load()
    idx, nam = index_paths(path)
    rank = comm.rank if comm is not None else 0
    local_rank = int(os.environ.get("LOCAL_RANK", rank)) if comm is not None and comm.world > 1 else 0

    def valid():
        n = ctypes.c_int64()
        ok = _private_file(idx) and lib.wgs_reader_index_sites(os.fsencode(path), os.fsencode(idx), ctypes.byref(n)) == 0
        if ok and names:
            ok = _private_file(nam) and _count_lines(nam) == n.value
        return ok, n.value

    ok, sites = valid()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", comm.world)) if comm is not None and comm.world > 1 else 1

    def build_alone():
        n = ctypes.c_int64()
        _lib.check(lib.wgs_reader_build_index(os.fsencode(path), os.fsencode(idx), os.fsencode(nam) if names else None,
                                              INDEX_SPAN_BYTES, INDEX_MAX_POINTS, ctypes.byref(n)))

    if comm is not None and comm.world > 1 and local_world > 1 and not names:
        # BGZF (what ANGSD writes): the pass is split over the ranks of the node -- every rank inflates and summarises the
        # blocks of its byte range, the first rank chains the parts (wgs_reader_index_part / _merge); anything else, or a
        # range that did not find the block chain: the first rank alone, as before
        need = comm.allreduce_sum(np.array([0.0 if ok else 1.0]))[0] > 0
        if need:
            prefix = idx + ".parts"
            rc = 0
            if not ok:
                rc = lib.wgs_reader_index_part(os.fsencode(path), os.fsencode("%s.%d" % (prefix, local_rank)), local_rank, local_world,
                                               host_threads())
            failed = comm.allreduce_sum(np.array([1.0 if rc else 0.0]))[0] > 0      # also the barrier: all parts are written
            if not ok and local_rank == 0:
                if not failed:
                    n = ctypes.c_int64()
                    failed = lib.wgs_reader_index_merge(os.fsencode(path), os.fsencode(idx), os.fsencode(prefix), local_world,
                                                        INDEX_SPAN_BYTES, INDEX_MAX_POINTS, ctypes.byref(n)) != 0
                if failed:
                    build_alone()
            comm.barrier()
    else:
        if local_rank == 0 and not ok:
            build_alone()
        if comm is not None and comm.world > 1:
            comm.barrier()
    ok, sites = valid()
    if not ok:
        raise RuntimeError("the index of %s could not be built or read (%s)" % (path, idx))
    return idx, (nam if names else None), sites


def read_site_names(path, comm=None):
    """(sample_names, site_names) of a Beagle file without parsing any likelihood: the names come out of the
    same single inflate pass that counts the sites and builds the index."""
    idx, nam, _ = ensure_index(path, comm, names=True)
    with BeagleStream(path, threads=1, index=idx, first_row=0) as st:
        samples = list(st.sample_names)
    with open(nam, "rb") as fh:
        names = fh.read().decode().split("\n")[:-1]
    return samples, names


def prefetched(gen, depth=1):
    """Run a generator one item ahead in a background thread: the next chunk is inflated and parsed (native code,
    GIL released) while the caller uploads the current one."""
    q = queue.Queue(maxsize=depth)
    done = object()

    def work():
        try:
            for item in gen:
                q.put(item)
            q.put(done)
        except BaseException as e:      # hand the error to the consumer
            q.put(e)

    threading.Thread(target=work, daemon=True).start()
    while True:
        item = q.get()
        if item is done:
            return
        if isinstance(item, BaseException):
            raise item
        yield item


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


def _index_is_cached(path):
    idx, _ = index_paths(path)
    n = ctypes.c_int64()
    return _private_file(idx) and _lib.load().wgs_reader_index_sites(os.fsencode(path), os.fsencode(idx), ctypes.byref(n)) == 0


def _stream_cold_file(path, group_of, n_groups, ctx, threads, names):
    """A BGZF file that has no index yet, one rank: the index pass (all host threads inflate and count: its result is the site
    count and the cached index, as ever) and the device ingest run AT THE SAME TIME instead of one after the other -- the host
    has little to do during the ingest (the device inflates) and the device nothing during the index pass.  The matrix is
    created for the estimate of wgs_reader_estimate_sites plus a quarter and cut to the file's sites afterwards
    (DeviceBeagle.set_rows); the two passes must agree about their number.  Returns None where this does not apply (an index
    exists, not BGZF, the host inflates, the estimate was too small): the caller then goes the usual way."""
    import threading
    from .device import DeviceBeagle
    if os.environ.get("WGSASSIGN_INFLATE", "device") in ("host", "zlib") or os.environ.get("WGSASSIGN_INGEST", "device") == "host":
        return None
    if _index_is_cached(path):
        return None
    est = estimate_sites(path)
    if est is None:
        return None
    box = {}

    def index_pass():
        try:
            box["index"] = ensure_index(path)
        except BaseException as e:          # handed to the caller's thread below
            box["error"] = e

    th = threading.Thread(target=index_pass, name="wgs-index-pass")
    th.start()
    beagle = None
    try:
        with BeagleStream(path, threads, index=None, first_row=0) as st:
            if callable(group_of):
                group_of, n_groups = group_of(list(st.sample_names))
            cap = est + est // 4 + 1024
            beagle = DeviceBeagle(cap, st.n, group_of, n_groups, site0=0, ctx=ctx)
            rows, site_names, tail = 0, [], []
            try:
                for nrows, chunk_names in st.ingest(beagle, 0, None, None, names=names):
                    rows += nrows
                    if names == "all":
                        site_names.extend(chunk_names)
                    else:
                        if len(site_names) < 4:
                            site_names = (site_names + chunk_names)[:4]
                        tail = (tail + chunk_names)[-4:]
            except (RuntimeError, ValueError) as e:
                if "outside the device matrix" not in str(e):
                    raise
                beagle.close()                  # more sites than estimated plus a quarter: the usual way, with the exact count
                beagle = None
                return None
            stats = getattr(st, "ingest_stats", None)
            samples = list(st.sample_names)
        th.join()
        if "error" in box:
            raise box["error"]
        m_file = box["index"][2]
        if rows != m_file:
            raise RuntimeError("Beagle file changed while reading: the index pass counted %d sites, the ingest parsed %d" % (m_file, rows))
        beagle.set_rows(rows)
        beagle.ingest_stats = stats
        if names != "all" and rows > 4:
            site_names = site_names + tail
        out, beagle = beagle, None
        return out, samples, site_names, m_file
    finally:
        th.join()
        if beagle is not None:
            beagle.close()


def stream_to_device(path, group_of=None, n_groups=1, ctx=None, threads=None, rank=0, world=1, m_total=None,
                     keep=None, comm=None, names="all"):
    """The file goes chunk by chunk straight into the device slabs -- host memory stays bounded by two chunks
    (the reference holds two full copies).  One indexing pass per file and node (ensure_index)
    counts the sites; with world > 1 every rank then starts at the access point before its contiguous SNP range
    (comm.shard_range) and inflates and parses only that range, the next chunk being prepared while the current
    one is uploaded.  group_of may be a callable(sample_names) -> (group_of, n_groups).  keep (bool array over the
    file's sites, e.g. the site mask of utils.filter_sites_to_common) restricts the matrix to the kept sites;
    shard ranges then count kept sites.  names="ends" keeps only the first and last four site names of the range
    (all the command line prints; a Python list of 50M names would cost gigabytes and minutes).
    Returns (DeviceBeagle, sample_names, site_names of the range, m_total)."""
    from .comm import shard_range
    from .device import DeviceBeagle
    if world == 1 and keep is None and m_total is None and os.environ.get("WGSASSIGN_COLD_ONE_PASS", "1") != "0":
        cold = _stream_cold_file(path, group_of, n_groups, ctx, threads, names)
        if cold is not None:
            return cold
    index, _, m_file = ensure_index(path, comm)
    if keep is not None:
        keep = np.asarray(keep, dtype=bool)
        kept_rows = np.flatnonzero(keep)
        m_total = len(kept_rows)
    elif m_total is None:
        m_total = m_file
    lo, hi = shard_range(m_total, rank, world)
    # file rows [r0, r1) cover this rank's (kept) sites
    if keep is None:
        r0, r1 = lo, hi
    elif hi > lo:
        r0, r1 = int(kept_rows[lo]), int(kept_rows[hi - 1]) + 1
    else:
        r0 = r1 = 0
    if hi <= lo:
        raise ValueError("fewer sites (%d) than ranks (%d): nothing to shard" % (m_total, world))
    with BeagleStream(path, threads, index=index, first_row=r0) as st:
        if callable(group_of):
            group_of, n_groups = group_of(list(st.sample_names))
        beagle = DeviceBeagle(hi - lo, st.n, group_of, n_groups, site0=lo, ctx=ctx)
        row0, frow, site_names, tail, names_mode = 0, r0, [], [], names
        if os.environ.get("WGSASSIGN_INGEST", "device") == "host":
            # the round-2 path, kept as the comparator: every value parsed on the host (wgs_reader_next), rows uploaded
            def chunks():
                nonlocal frow
                for rows, names in prefetched(st.chunks(limit=r1 - r0)):
                    if keep is not None:
                        sel = keep[frow:frow + rows.shape[0]]
                        frow += rows.shape[0]
                        rows = rows[sel]
                        names = [x for x, k in zip(names, sel) if k]
                    if rows.shape[0]:
                        beagle.upload_rows(np.ascontiguousarray(rows), row0)
                    yield rows.shape[0], names
        else:
            def chunks():
                return st.ingest(beagle, 0, r1 - r0, None if keep is None else keep[r0:r1], names=names_mode)
        for nrows, names in chunks():
            row0 += nrows
            if names_mode == "all":
                site_names.extend(names)
            else:
                if len(site_names) < 4:
                    site_names = (site_names + names)[:4]
                tail = (tail + names)[-4:]
        beagle.ingest_stats = getattr(st, "ingest_stats", None)
        if names_mode != "all" and row0 > 4:
            site_names = site_names + tail         # [:4] are the first four names of the range, [-4:] the last four
        if row0 != hi - lo:
            raise RuntimeError("Beagle file changed while reading: expected %d sites, parsed %d" % (hi - lo, row0))
        return beagle, list(st.sample_names), site_names, m_total
