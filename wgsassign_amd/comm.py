"""The one collective the SNP-sharded path needs: a sum all-reduce of a few float64.

SNPs shard trivially (every SNP's update and log-likelihood term is independent:
emMAF_cy.pyx:16-23, glassy_cy.pyx:17-21); ranks own contiguous SNP ranges in rank order.
Exchange steps: per EM iteration the per-fit sums of squared differences (n_fits doubles), per
undecided fit the float32 carry of the serial convergence chain, once per assignment the
n x K partial log-likelihood sums, and for exact partition sums the float32 carries of the chains.

Communicators (all expose rank, world, allreduce_sum, gather_rows, allgather_object, barrier):
  LocalComm   one process.
  RcclComm    default for N > 1: RCCL over xGMI through the library's own communicator
              (include/wgsassign_hip.h: wgs_comm_*; librccl dlopen'ed, no tensor framework).  Bootstrap
              and the few host-side gathers go over a persistent TCP star (SideChannel).
  SocketComm  the same TCP star also carries the all-reduce (host-staged, summed in rank order): the
              fallback when RCCL cannot initialise, and a CPU-testable multi-rank path without torch.
  TorchComm   torch.distributed process group (`nccl` == RCCL, or `gloo`): optional, used by the
              gloo rehearsals that put two ranks on one GPU.
N > 1 over RCCL has not been executed yet on real hardware (no multi-GPU box was available to the
build); the protocol is covered by gloo / socket multi-rank tests.
"""
import hashlib
import json
import os
import socket
import struct
import time

import numpy as np

COMM_INIT_FAILED = 75        # exit status of a worker whose communicator could not initialise (EX_TEMPFAIL)
COMM_DIVERGED = 76           # exit status of a worker that found the ranks issuing different collectives (never retried)


class CollectiveMismatch(RuntimeError):
    """The ranks have stopped issuing the same sequence of collectives: wrong numbers would follow, so every rank that sees it
    stops.  The message names what this rank issued and what the other one did."""


# ---- self-checking collectives -----------------------------------------------------------------
# Every all-reduce carries, behind its payload, a table of one row per rank: {sequence number on this communicator, op, generation,
# iteration, shape a, shape b, payload elements, aux}.  A rank fills its own row and zeros the rest, so the sum all-gathers the rows;
# afterwards every rank compares all rows with its own (aux, a free word, excepted).  The C library does the same for the
# collectives it issues itself (csrc/rccl_comm.hip); this is the host side's copy for the all-reduces Python issues -- over sockets,
# torch.distributed or, through wgs_comm_allreduce_host_tagged, RCCL.  Host all-reduces are told apart by their CALL SITE: the tag of
# one issued at device.py line 458 is op 8 (WGS_OP_HOST), shape a = 100000 * (index of "device.py") + 458.
TAG_WORDS = 8
OP_HOST = 8
_SITE_FILES = ["device.py", "glassy.py", "fisher.py", "reader_cy.py", "WGSassign.py", "comm.py", "emMAF.py", "bench.py", "utils.py"]


def _call_site(depth=1):
    """100000 * (1 + index of the file) + line of the first frame outside this module's all-reduce plumbing."""
    import sys
    f = sys._getframe(1)
    while f is not None and os.path.basename(f.f_code.co_filename) == "comm.py" and f.f_back is not None and \
            f.f_code.co_name in ("allreduce_sum", "barrier", "_tagged_allreduce", "fn", "allreduce_device"):
        f = f.f_back
    name = os.path.basename(f.f_code.co_filename)
    idx = _SITE_FILES.index(name) if name in _SITE_FILES else len(_SITE_FILES)
    return 100000 * (idx + 1) + min(int(f.f_lineno), 99999)


def site_name(code):
    idx, line = int(code) // 100000 - 1, int(code) % 100000
    return "%s:%d" % (_SITE_FILES[idx] if 0 <= idx < len(_SITE_FILES) else "<other file>", line)


def describe_row(rank, w):
    what = "host all-reduce at %s" % site_name(w[4]) if int(w[1]) == OP_HOST else "op %d, shape %d" % (int(w[1]), int(w[4]))
    return "rank %d issued collective #%d: %s, generation %d, iteration %d, shape b %d, %d payload elements" % (
        rank, int(w[0]), what, int(w[2]), int(w[3]), int(w[5]), int(w[6]))


def decode_sites(msg):
    """The library's own mismatch text with the call sites of host all-reduces (op 8: shape a) spelled out."""
    import re
    return re.sub(r"\(op 8\), generation (-?\d+), iteration (-?\d+), shape (\d+) /",
                  lambda mt: "(op 8), generation %s, iteration %s, shape %s [%s] /" % (mt.group(1), mt.group(2), mt.group(3), site_name(mt.group(3))), msg)


class _Tagged:
    """allreduce_sum(arr, tag=None) over a subclass's _allreduce_raw(flat float64) with the rows attached and compared.
    tag: None, or (generation, iteration, shape_b, aux) -- the call site supplies op and shape a."""
    _seq = 0
    last_rows = None         # the table of the last all-reduce: last_rows[r, 7] is rank r's aux

    def _tagged_allreduce(self, arr, tag=None, depth=3):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        gen, it, shape_b, aux = tag if tag is not None else (0, 0, 0, 0)
        self._seq += 1
        row = np.array([self._seq, OP_HOST, gen, it, _call_site(depth), shape_b, a.size, aux], dtype=np.float64)
        table = np.zeros((self.world, TAG_WORDS))
        table[self.rank] = row
        out = np.asarray(self._allreduce_raw(np.concatenate([a.reshape(-1), table.reshape(-1)])), dtype=np.float64)
        rows = out[a.size:].reshape(self.world, TAG_WORDS)
        for r in range(self.world):
            if not np.array_equal(rows[r, :7], row[:7]):
                raise CollectiveMismatch("collective mismatch: the ranks have stopped issuing the same sequence of collectives -- %s; %s"
                                         % (describe_row(self.rank, row), describe_row(r, rows[r])))
        self.last_rows = rows
        return out[:a.size].reshape(a.shape).copy()


class LocalComm:
    rank = 0
    world = 1

    def allreduce_sum(self, arr, tag=None):
        return arr

    def gather_rows(self, arr):
        return arr

    def allgather_object(self, obj):
        return [obj]

    def barrier(self):
        pass

    def close(self):
        pass


# ---------------------------------------------------------------------------------------------
class SideChannel:
    """Persistent TCP star: rank 0 listens once, every other rank keeps ONE connection for the
    communicator's lifetime.  All operations are collective and issued in the same order on every
    rank, so a connection carries them strictly in sequence (no per-call accept, nothing to race).
    Frames are length-prefixed raw bytes; objects travel as JSON, arrays as a JSON header + raw
    buffer -- nothing is unpickled.  A connection must open with magic | rank | world | token."""

    MAGIC = b"WGSCOMM3"
    ABORT = -1               # kind of the frame with which rank 0 tells the others that the ranks are out of step

    def __init__(self, rank, world, addr="127.0.0.1", port=29401, token=None, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.peers, self.hub = {}, None
        self._seq = 0            # collectives issued on this star: every frame says which one it belongs to, and of what kind
        if self.world == 1:
            return
        if token is None:
            token = os.environ.get("WGSASSIGN_COMM_TOKEN") or "%s:%d:%d:%d:%s" % (
                addr, port, world, os.getuid(), os.environ.get("TORCHELASTIC_RUN_ID", ""))
        tok = hashlib.sha256(token.encode()).digest()
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            bind_until = time.monotonic() + min(10.0, timeout)
            while True:                          # the port may be some connection's ephemeral local port for a moment
                try:
                    srv.bind((addr, int(port)))
                    break
                except OSError as e:
                    if time.monotonic() > bind_until:
                        raise RuntimeError("SideChannel: cannot listen on %s:%d (%s)" % (addr, port, e))
                    time.sleep(0.2)
            srv.listen(self.world)
            try:
                while len(self.peers) < self.world - 1:
                    srv.settimeout(max(0.1, deadline - time.monotonic()))
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        raise RuntimeError("SideChannel: %d of %d ranks connected within %.0f s" %
                                           (len(self.peers) + 1, self.world, timeout))
                    conn.settimeout(10.0)
                    try:
                        hello = self._take(conn, 8 + 8 + 32)
                    except Exception:
                        conn.close()
                        continue
                    r, w = struct.unpack("<ii", hello[8:16])
                    if hello[:8] != self.MAGIC or w != self.world or not 0 < r < self.world or r in self.peers \
                            or hello[16:] != tok:
                        conn.close()             # not one of ours
                        continue
                    conn.settimeout(None)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.peers[r] = conn
            finally:
                srv.close()
            for r in sorted(self.peers):
                self._send(self.peers[r], b"ok")
        else:
            last = None
            while time.monotonic() < deadline:
                try:
                    c = socket.create_connection((addr, int(port)), timeout=5.0)
                    break
                except OSError as e:
                    last = e
                    time.sleep(0.05)
            else:
                raise RuntimeError("SideChannel: cannot reach rank 0 at %s:%d (%s)" % (addr, port, last))
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.sendall(self.MAGIC + struct.pack("<ii", self.rank, self.world) + tok)
            c.settimeout(max(1.0, deadline - time.monotonic()))
            if self._recv(c)[2] != b"ok":
                raise RuntimeError("SideChannel: rank 0 refused the connection")
            c.settimeout(None)
            self.hub = c

    @staticmethod
    def _take(conn, k):
        buf = bytearray(k)
        view, got = memoryview(buf), 0
        while got < k:
            r = conn.recv_into(view[got:], k - got)
            if r == 0:
                raise RuntimeError("SideChannel: peer closed the connection")
            got += r
        return bytes(buf)

    @classmethod
    def _recv(cls, conn):
        """(kind, sequence number, payload) of the next frame."""
        size, kind, seq = struct.unpack("<qii", cls._take(conn, 16))
        return kind, seq, cls._take(conn, size)

    @staticmethod
    def _send(conn, data, kind=0, seq=0):
        conn.sendall(struct.pack("<qii", len(data), kind, seq))
        conn.sendall(data)

    def _expect(self, got, kind, sender):
        """A frame of another kind or number than this rank's own current collective: the ranks are out of step."""
        k, q, data = got
        if k == self.ABORT:
            raise CollectiveMismatch(data.decode(errors="replace"))
        if (k, q) != (kind, self._seq):
            msg = ("collective mismatch: the ranks have stopped issuing the same sequence of collectives -- rank %d is at host "
                   "collective #%d of kind %d; rank %d sent a frame of collective #%d, kind %d" % (self.rank, self._seq, kind, sender, q, k))
            self.abort(msg)
        return data

    def abort(self, msg):
        """Raise CollectiveMismatch(msg); rank 0 first tells every other rank, whatever it is waiting for."""
        if self.rank == 0:
            for c in self.peers.values():
                try:
                    self._send(c, msg.encode(), self.ABORT, self._seq)
                except OSError:
                    pass
        raise CollectiveMismatch(msg)

    # ---- collectives on bytes (kind: what the caller is doing -- 1 all-reduce, 2 rows, 3 objects, 4 bootstrap)
    def bcast(self, data, kind=0):
        """rank 0's bytes on every rank."""
        if self.world == 1:
            return data
        self._seq += 1
        if self.rank == 0:
            for r in sorted(self.peers):
                self._send(self.peers[r], data, kind, self._seq)
            return data
        return self._expect(self._recv(self.hub), kind, 0)

    def gather(self, data, kind=0):
        """list of every rank's bytes (rank order) on rank 0, None elsewhere."""
        if self.world == 1:
            return [data]
        self._seq += 1
        if self.rank == 0:
            return [data] + [self._expect(self._recv(self.peers[r]), kind, r) for r in range(1, self.world)]
        self._send(self.hub, data, kind, self._seq)
        return None

    def close(self):
        for c in list(self.peers.values()) + ([self.hub] if self.hub else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.hub = {}, None


def _pack_array(a):
    a = np.ascontiguousarray(a)
    head = json.dumps({"dtype": a.dtype.str, "shape": list(a.shape)}).encode()
    return struct.pack("<i", len(head)) + head + a.tobytes()


def _unpack_array(blob):
    k = struct.unpack("<i", blob[:4])[0]
    head = json.loads(blob[4:4 + k].decode())
    dt = np.dtype(head["dtype"])
    if dt.kind not in "fiub":
        raise RuntimeError("SideChannel: refusing array of dtype %r" % head["dtype"])
    return np.frombuffer(blob, dtype=dt, offset=4 + k).reshape(head["shape"])


class SocketComm(_Tagged):
    """All collectives over the TCP star (host-staged; sums formed on rank 0 in rank order, so every
    rank sees identical bits).  Latency ~0.1 ms per all-reduce: fine for the few float64 this path
    exchanges, and independent of any GPU library."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29400, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.ch = SideChannel(rank, world, addr, int(port) + 1, timeout=timeout)
        self._host_h = self._host_fn = None

    def attach(self, ctx):
        """Give the library's own loops (wgs_em_fit, wgs_loo) this transport: a wgs_comm whose all-reduce calls back into
        allreduce_sum below (wgs_comm_create_host).  After this `handle` is set and EMBatch.run / glassy.loo_device take
        their one-call C paths across the ranks, exactly as they do over RCCL."""
        import ctypes
        from . import _lib
        if self._host_h is not None or self.world == 1:
            return self

        def fn(buf, n, _user):
            try:
                view = np.ctypeslib.as_array(buf, shape=(int(n),))
                view[:] = SocketComm._tagged_allreduce(self, view.copy(), None, 2)
                return 0
            except BaseException as e:
                _lib.callback_error = e              # (the library reports a failed all-reduce; _lib.check() raises this instead)
                return 1
        self._host_fn = _lib.ALLREDUCE_FN(fn)                  # kept alive with the communicator
        h = ctypes.c_void_p()
        _lib.check(_lib.load().wgs_comm_create_host(ctx.handle, self.rank, self.world, self._host_fn, None, ctypes.byref(h)))
        self._host_h = h
        return self

    @property
    def handle(self):
        """wgs_comm* for the C entry points that run whole loops, or None (not attached: step-by-step Python drivers)."""
        return self._host_h

    def allreduce_sum(self, arr, tag=None):
        if self.world == 1:
            return np.ascontiguousarray(arr, dtype=np.float64)
        return self._tagged_allreduce(arr, tag)

    def _allreduce_raw(self, a):
        parts = self.ch.gather(a.tobytes(), 1)
        if self.rank == 0:
            if any(len(p) != len(parts[0]) for p in parts):
                self.ch.abort("collective mismatch: the ranks have stopped issuing the same sequence of collectives -- their all-reduce "
                              "payloads (with %d float64 of tag rows each) have %s float64, by rank" % (self.world * TAG_WORDS, [len(p) // 8 for p in parts]))
            tot = np.frombuffer(parts[0], dtype=np.float64).copy()
            for p in parts[1:]:
                tot += np.frombuffer(p, dtype=np.float64)
            blob = self.ch.bcast(tot.tobytes(), 1)
        else:
            blob = self.ch.bcast(None, 1)
        return np.frombuffer(blob, dtype=np.float64)

    def gather_rows(self, arr):
        """Every rank's rows (SNP shards, rank order) concatenated on rank 0 -- raw buffers, rank 0 only;
        the other ranks get None (only rank 0 writes output files)."""
        parts = self.ch.gather(_pack_array(arr), 2)
        if self.rank != 0:
            return None
        return np.concatenate([_unpack_array(p) for p in parts], axis=0)

    def allgather_object(self, obj):
        """JSON-serialisable objects (site-name previews, counts) from every rank, on every rank."""
        parts = self.ch.gather(json.dumps(obj).encode(), 3)
        blob = self.ch.bcast(json.dumps([json.loads(p.decode()) for p in parts]).encode() if self.rank == 0 else None, 3)
        return json.loads(blob.decode())

    def barrier(self):
        if self.world > 1:
            self._tagged_allreduce(np.zeros(1))

    def _detach(self):
        if getattr(self, "_host_h", None) is not None:
            from . import _lib
            _lib.load().wgs_comm_destroy(self._host_h)
            self._host_h = self._host_fn = None

    def close(self):
        self._detach()
        self.ch.close()


class RcclComm(SocketComm):
    """RCCL through the library's own communicator (include/wgsassign_hip.h: wgs_comm_*): no tensor
    framework involved.  The 128-byte unique id travels from rank 0 to the others over the TCP star on
    MASTER_ADDR:(MASTER_PORT + 1); the all-reduces run on the context's HIP stream over xGMI.  If any
    rank cannot initialise RCCL, ALL ranks agree (over the star) to keep the socket all-reduce."""

    def __init__(self, ctx, rank, world, addr="127.0.0.1", port=29400, timeout=120.0):
        import ctypes
        from . import _lib
        super().__init__(rank, world, addr, port, timeout)
        self._lib, self._ct, self.ctx = _lib, ctypes, ctx
        self._h, self.native, self.native_error = None, False, ""
        lib = _lib.load()
        ident = (ctypes.c_uint8 * 128)()
        ok = 1
        if self.rank == 0 and lib.wgs_comm_unique_id(ident) != 0:
            ok, self.native_error = 0, _lib.last_error()
        blob = self.ch.bcast(bytes([ok]) + bytes(ident) if self.rank == 0 else None, 4)
        if blob[0]:
            ident = (ctypes.c_uint8 * 128).from_buffer_copy(blob[1:129])
            h = ctypes.c_void_p()
            # ncclCommInitRank is collective: if a peer never arrives it does not return.  A rank stuck in it for
            # `timeout` seconds leaves with COMM_INIT_FAILED -- the launchers (bench.py, `WGSassign --gpus N`) then start
            # the ranks again over the socket all-reduce.
            import threading
            done = threading.Event()

            def watchdog():
                if not done.wait(timeout):
                    import sys
                    print("wgsassign_amd: rank %d: RCCL communicator did not initialise within %.0f s" % (self.rank, timeout),
                          file=sys.stderr, flush=True)
                    os._exit(COMM_INIT_FAILED)
            threading.Thread(target=watchdog, daemon=True).start()
            rc = lib.wgs_comm_init(ctx.handle, ident, self.rank, self.world, ctypes.byref(h))
            done.set()
            if rc == 0:
                self._h = h
            else:
                ok, self.native_error = 0, _lib.last_error()
        flags = SocketComm._tagged_allreduce(self, np.array([float(ok and blob[0])]), None, 2)
        self.native = int(flags[0]) == self.world
        if not self.native:
            if self._h:
                lib.wgs_comm_destroy(self._h)
                self._h = None
            self.attach(ctx)                     # the library's loops still run in one call, over the TCP all-reduce

    # ---- the collective
    def allreduce_sum(self, arr, tag=None):
        if not self.native:
            return SocketComm.allreduce_sum(self, arr, tag)
        a = np.ascontiguousarray(arr, dtype=np.float64).copy()
        gen, it, shape_b, aux = tag if tag is not None else (0, 0, 0, 0)
        ctag = self._lib.CollTag(OP_HOST, int(gen), int(it), _call_site(2), int(shape_b), int(aux))
        rows = np.zeros((self.world, TAG_WORDS))
        self._lib.check(self._lib.load().wgs_comm_allreduce_host_tagged(self._h, self._lib.f64p(a.reshape(-1)), a.size,
                                                                        self._ct.byref(ctag), self._lib.f64p(rows)))
        self.last_rows = rows
        return a

    def barrier(self):
        self.allreduce_sum(np.zeros(1))

    def step_reduced(self, em_handle, n_fits):
        """EM sweep into the communicator's device buffer, all-reduce behind it on the same stream,
        one readback (the per-iteration exchange of the sharded EM; see EMBatch.step_reduced)."""
        lib = self._lib.load()
        if not self.native:
            out = np.zeros(int(n_fits), dtype=np.float64)
            self._lib.check(lib.wgs_em_step(em_handle, self._lib.f64p(out)))
            return SocketComm._tagged_allreduce(self, out)
        buf = lib.wgs_comm_buffer(self._h, int(n_fits))
        if not buf:
            raise RuntimeError("wgsassign_amd HIP call failed: " + self._lib.last_error())
        self._lib.check(lib.wgs_em_step_dev(em_handle, self._ct.c_void_p(buf)))
        out = np.zeros(int(n_fits), dtype=np.float64)
        self._lib.check(lib.wgs_comm_allreduce_buffer(self._h, int(n_fits), self._lib.f64p(out)))
        return out

    def info(self):
        """What the communicator is, by RCCL's own account: dict with `native`, `rccl_ranks_seen` (ncclCommCount),
        `rccl_rank`, `rccl_device` (-1 for the socket fall-back), `world`, `rank`."""
        h = self.handle
        out = (self._ct.c_int64 * 8)()
        self._lib.check(self._lib.load().wgs_comm_info(h, out))
        return {"native": bool(out[0]), "rccl_ranks_seen": int(out[1]), "rccl_rank": int(out[2]), "rccl_device": int(out[3]),
                "world": int(out[4]), "rank": int(out[5]), "collectives_issued": int(out[6]), "out_of_step": bool(out[7])}

    def time_collectives(self, reps=20, n=16):
        """Mean device microseconds of a sum all-reduce / a broadcast of n float64 on the context's stream (collective)."""
        us = np.zeros(2)
        self._lib.check(self._lib.load().wgs_comm_time_collectives(self.handle, int(reps), int(n), self._lib.f64p(us)))
        return {"allreduce_us": float(us[0]), "bcast_us": float(us[1]), "float64_per_call": int(n), "calls": int(reps)}

    @property
    def handle(self):
        """wgs_comm* for the C entry points that run whole loops (wgs_em_fit, wgs_loo): RCCL, or the host-backed one."""
        return self._h if self.native else self._host_h

    def close(self):
        if self._h:
            self._lib.load().wgs_comm_destroy(self._h)
            self._h = None
        SocketComm.close(self)


class TorchComm(_Tagged):
    """torch.distributed process group (backend `nccl` == RCCL on ROCm, or `gloo` on CPU).

    Ordering note: torch bundles its own HIP runtime; import torch and initialise the process group
    BEFORE the first wgsassign_amd device call (init_from_env does), otherwise torch
    reports "No HIP GPUs are available"."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self._cuda = dist.get_backend() == "nccl"
        self._device = device

    def _dev(self):
        return self._device if self._device is not None else "cuda"

    def allreduce_sum(self, arr, tag=None):
        """Sum `arr` (float64) over all ranks; every rank gets identical bits (and has compared every rank's tag row with its own)."""
        return self._tagged_allreduce(arr, tag)

    def _allreduce_raw(self, a):
        torch, dist = self._torch, self._dist
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        if self._cuda:
            t = t.to(self._dev())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    # ---- device-resident variant: the sums never visit the host before the collective
    def device_buffer(self, n):
        """A float64 CUDA tensor of n elements (its data_ptr() is handed to wgs_em_step_dev) + room for the ranks' tag rows."""
        if not self._cuda:
            return None
        return self._torch.zeros(int(n) + self.world * TAG_WORDS, dtype=self._torch.float64, device=self._dev())

    def allreduce_device(self, t, stream_ptr, tag=None):
        """RCCL all-reduce of tensor t (a device_buffer), ordered after the work already enqueued on the library's
        HIP stream `stream_ptr`, the ranks' tag rows behind the payload; returns the reduced values as a NumPy array."""
        torch, dist = self._torch, self._dist
        n = t.numel() - self.world * TAG_WORDS
        gen, it, shape_b, aux = tag if tag is not None else (0, 0, 0, 0)
        self._seq += 1
        row = np.array([self._seq, OP_HOST, gen, it, _call_site(), shape_b, n, aux], dtype=np.float64)
        table = np.zeros((self.world, TAG_WORDS))
        table[self.rank] = row
        ext = torch.cuda.ExternalStream(int(stream_ptr), device=t.device)
        with torch.cuda.stream(ext):
            t[n:] = torch.from_numpy(table.reshape(-1)).to(t.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            out = t.cpu().numpy()    # enqueued on the same stream, synchronises it
        rows = out[n:].reshape(self.world, TAG_WORDS)
        for r in range(self.world):
            if not np.array_equal(rows[r, :7], row[:7]):
                raise CollectiveMismatch("collective mismatch: the ranks have stopped issuing the same sequence of collectives -- %s; %s"
                                         % (describe_row(self.rank, row), describe_row(r, rows[r])))
        return out[:n]

    def gather_rows(self, arr):
        """Every rank's rows (rank order) concatenated on rank 0 (None elsewhere): row counts by one
        all-reduce, then point-to-point tensor sends -- raw buffers, nothing pickled."""
        torch, dist = self._torch, self._dist
        a = np.ascontiguousarray(arr)
        counts = np.zeros(self.world)
        counts[self.rank] = a.shape[0]
        counts = self._tagged_allreduce(counts).astype(np.int64)
        if self.world == 1:
            return a
        dev = self._dev() if self._cuda else "cpu"
        if self.rank != 0:
            dist.send(torch.from_numpy(a).to(dev), dst=0)
            return None
        parts = [a]
        for r in range(1, self.world):
            t = torch.empty((int(counts[r]),) + a.shape[1:], dtype=torch.from_numpy(a[:0]).dtype, device=dev)
            dist.recv(t, src=r)
            parts.append(t.cpu().numpy())
        return np.concatenate(parts, axis=0)

    def allgather_object(self, obj):
        parts = [None] * self.world
        self._dist.all_gather_object(parts, obj)
        return parts

    def barrier(self):
        self._dist.barrier()

    def close(self):
        pass


def init_from_env(ctx=None):
    """Communicator from the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*, as set by
    torchrun or by bench.py's own launcher).  WGSASSIGN_COMM selects it: `rccl` (default: the library's
    own RCCL communicator, no torch), `socket` (TCP all-reduce), `torch` (torch.distributed; backend
    WGSASSIGN_BACKEND = nccl | gloo -- gloo lets several ranks share one GPU in rehearsals; setting
    WGSASSIGN_BACKEND=gloo alone implies torch).  Returns LocalComm() when WORLD_SIZE is 1 or unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return LocalComm()
    rank = int(os.environ.get("RANK", "0"))
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29400"))
    backend = os.environ.get("WGSASSIGN_BACKEND", "nccl")
    kind = os.environ.get("WGSASSIGN_COMM", "torch" if backend == "gloo" else "rccl")
    if kind == "socket":
        from .device import get_context
        return SocketComm(rank, world, addr, port).attach(ctx or get_context())
    if kind == "rccl":
        from .device import get_context
        return RcclComm(ctx or get_context(), rank, world, addr, port)
    if kind != "torch":
        raise ValueError("WGSASSIGN_COMM must be rccl, socket or torch, got %r" % kind)
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return TorchComm(device=torch.device("cuda", local_rank) if backend == "nccl" else None)


def usable_cpus():
    """CPUs this process can really keep busy: its affinity set, cut down to the CPU-time quota of its control group
    (cgroup v2 cpu.max / v1 cpu.cfs_quota_us) when there is one -- a container with 256 visible CPUs and a quota of 16
    runs 256 OpenMP threads 50x slower than 16."""
    n = len(os.sched_getaffinity(0))

    def read(p):
        try:
            return open(p).read().split()
        except OSError:
            return None
    v2 = read("/sys/fs/cgroup/cpu.max")
    quota = None
    if v2 and len(v2) == 2 and v2[0] != "max":
        quota = float(v2[0]) / float(v2[1])
    else:
        q, per = read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        if q and per and float(q[0]) > 0:
            quota = float(q[0]) / float(per[0])
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def comm_attempts(base):
    """What a launcher tries, in order, each with fresh child processes: RCCL with dmabuf IPC between the ranks
    (HSA_ENABLE_IPC_MODE_LEGACY=0 -- what this host driver supports, and the default here unless the caller's
    environment says otherwise); RCCL with that setting flipped (other drivers only do legacy IPC); the socket
    all-reduce, which needs nothing from the GPU runtime.  A rank leaves with COMM_INIT_FAILED (75) when its
    communicator cannot be built or does not come up within the watchdog's time; any other failure ends the job.
    Returns [(label, environment overrides)]."""
    first = base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if base.get("WGSASSIGN_COMM", "rccl") != "rccl" or base.get("WGSASSIGN_BACKEND") == "gloo":
        return [("as configured", {"HSA_ENABLE_IPC_MODE_LEGACY": first})]
    return [("rccl, HSA_ENABLE_IPC_MODE_LEGACY=%s" % first, {"HSA_ENABLE_IPC_MODE_LEGACY": first}),
            ("rccl, HSA_ENABLE_IPC_MODE_LEGACY=%s" % ("1" if first == "0" else "0"), {"HSA_ENABLE_IPC_MODE_LEGACY": "1" if first == "0" else "0"}),
            ("socket all-reduce", {"HSA_ENABLE_IPC_MODE_LEGACY": first, "WGSASSIGN_COMM": "socket"})]


def launch_local_ranks(n, argv, env=None, poll=0.05):
    """Start `argv` n times as ranks 0 .. n-1 of this node -- RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, a free port chosen -- and wait for them: what `WGSassign --gpus N` does instead of
    asking for torchrun.  Rank 0 writes to this process's stdout, the others only to stderr.  The caller must not have
    touched the GPU (children are separate processes; nothing is exec'ed).  When ANY rank reports that its
    communicator did not initialise (status 75) the ranks are started again with the next of comm_attempts().
    Returns 0, or a failed rank's own status."""
    import sys
    base = dict(os.environ if env is None else env)
    attempts = comm_attempts(base)
    codes = [1]
    for i, (label, over) in enumerate(attempts):
        codes = _run_ranks(n, argv, dict(base, **over), free_port_pair(), poll)
        if all(c == 0 for c in codes):
            return 0
        if not any(c == COMM_INIT_FAILED for c in codes) or i + 1 == len(attempts):
            break
        print("wgsassign_amd: the communicator did not initialise (%s); starting the ranks again: %s" % (label, attempts[i + 1][0]),
              file=sys.stderr, flush=True)
    return next((c for c in codes if c > 0), 1)      # a rank's own status rather than the -SIGTERM of the ones ended here


def free_port_pair(addr="127.0.0.1"):
    """A port p such that p and p + 1 can both be bound right now: MASTER_PORT for a launcher (the TCP star of the
    communicators listens on MASTER_PORT + 1)."""
    for _ in range(200):
        with socket.socket() as a:
            a.bind((addr, 0))
            p = a.getsockname()[1]
            if p >= 65535:
                continue
            with socket.socket() as b:
                try:
                    b.bind((addr, p + 1))
                except OSError:
                    continue
            return p
    raise RuntimeError("no two consecutive free ports on %s" % addr)


def _run_ranks(n, argv, base, port, poll, grace=3.0):
    """One attempt: every rank's exit status.  After the first failure the others get `grace` seconds to leave by
    themselves (a rank whose peer failed fast may still be on its way to its own status 75), then they are ended."""
    import subprocess
    import time
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        procs.append(subprocess.Popen(argv, env=e, stdout=None if r == 0 else subprocess.DEVNULL))
    try:
        failed_at = None
        while any(p.poll() is None for p in procs):
            if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
                failed_at = time.time()
            if failed_at is not None and time.time() - failed_at > grace:
                break
            time.sleep(poll)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return [p.returncode for p in procs]


SHARD_ALIGN = 8192      # NumPy's reductions work through 8192-element chunks (device.Score.sums)


def shard_range(m_total, rank, world):
    """Contiguous SNP range [lo, hi) of `rank`: GPU g owns [g*m/G, (g+1)*m/G) -- with the cuts moved down
    to multiples of 8192 sites when the shards are at least that long, so that every chunk of NumPy's float64
    summation (glassy.py:38) lies inside one shard and the running total can be handed from shard to shard."""
    def cut(r):
        c = (m_total * r) // world
        return c - c % SHARD_ALIGN if 0 < r < world and m_total // world >= SHARD_ALIGN else c
    return cut(rank), cut(rank + 1)
