"""The one collective the SNP-sharded path needs: a sum all-reduce of a few float64.

SNPs shard trivially (every SNP's update and log-likelihood term is independent:
emMAF_cy.pyx:16-23, glassy_cy.pyx:17-21); ranks own contiguous SNP ranges in rank order.
Exchange steps: per EM iteration the per-fit sums of squared differences (n_fits doubles), per
undecided fit the float32 carry of the serial convergence chain, and once per assignment the
n x K partial log-likelihood sums.  With the `nccl` backend this is RCCL over xGMI.
"""
import numpy as np


class LocalComm:
    rank = 0
    world = 1

    def allreduce_sum(self, arr):
        return arr

    def gather_rows(self, arr):
        return arr

    def allgather_object(self, obj):
        return [obj]

    def barrier(self):
        pass


class TorchComm:
    """torch.distributed process group (backend `nccl` == RCCL on ROCm, or `gloo` on CPU).

    Ordering note: torch bundles its own HIP runtime; import torch and initialise the process group
    BEFORE the first wgsassign_amd device call (init_from_env and bench.py do), otherwise torch
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

    def allreduce_sum(self, arr):
        """Sum `arr` (float64) over all ranks; every rank gets identical bits."""
        torch, dist = self._torch, self._dist
        a = np.ascontiguousarray(arr, dtype=np.float64)
        t = torch.from_numpy(a.copy())
        if self._cuda:
            t = t.to(self._device if self._device is not None else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy().reshape(a.shape)

    # ---- device-resident variant: the sums never visit the host before the collective
    def device_buffer(self, n):
        """A float64 CUDA tensor of n elements (its data_ptr() is handed to wgs_em_step_dev)."""
        if not self._cuda:
            return None
        return self._torch.zeros(int(n), dtype=self._torch.float64,
                                 device=self._device if self._device is not None else "cuda")

    def allreduce_device(self, t, stream_ptr):
        """RCCL all-reduce of tensor t, ordered after the work already enqueued on the library's
        HIP stream `stream_ptr`; returns the reduced values as a NumPy array."""
        torch, dist = self._torch, self._dist
        ext = torch.cuda.ExternalStream(int(stream_ptr), device=t.device)
        with torch.cuda.stream(ext):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            out = t.cpu()            # enqueued on the same stream, synchronises it
        return out.numpy()

    def gather_rows(self, arr):
        """Concatenate every rank's rows (SNP shards, rank order) -- the full array on every rank."""
        parts = [None] * self.world
        self._dist.all_gather_object(parts, np.ascontiguousarray(arr))
        return np.concatenate(parts, axis=0)

    def allgather_object(self, obj):
        parts = [None] * self.world
        self._dist.all_gather_object(parts, obj)
        return parts

    def barrier(self):
        self._dist.barrier()


class RcclComm:
    """RCCL through the library's own communicator (include/wgsassign_hip.h: wgs_comm_*): no tensor
    framework involved.  The 128-byte unique id travels from rank 0 to the others over a plain TCP
    socket on MASTER_ADDR:(MASTER_PORT + 1); objects (gather_rows) use the same channel pattern."""

    def __init__(self, ctx, rank, world, addr="127.0.0.1", port=29400):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        self.rank, self.world, self.ctx = int(rank), int(world), ctx
        self._addr, self._port = addr, int(port) + 1
        lib = _lib.load()
        ident = (ctypes.c_uint8 * 128)()
        if self.rank == 0:
            _lib.check(lib.wgs_comm_unique_id(ident))
        blob = self._bcast_bytes(bytes(ident) if self.rank == 0 else None)
        ident = (ctypes.c_uint8 * 128).from_buffer_copy(blob)
        h = ctypes.c_void_p()
        _lib.check(lib.wgs_comm_init(ctx.handle, ident, self.rank, self.world, ctypes.byref(h)))
        self._h = h
        self._dev = None

    # ---- tiny TCP helpers (bootstrap and object gathers only; the data path is RCCL)
    def _serve(self, handler):
        import socket
        with socket.socket() as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((self._addr, self._port))
            srv.listen(self.world)
            for _ in range(self.world - 1):
                conn, _a = srv.accept()
                with conn:
                    handler(conn)

    def _connect(self):
        import socket
        import time
        for _ in range(600):
            try:
                return socket.create_connection((self._addr, self._port), timeout=60)
            except OSError:
                time.sleep(0.1)
        raise RuntimeError("RcclComm: cannot reach rank 0 at %s:%d" % (self._addr, self._port))

    @staticmethod
    def _send(conn, data):
        conn.sendall(len(data).to_bytes(8, "little") + data)

    @staticmethod
    def _recv(conn):
        def take(k):
            buf = b""
            while len(buf) < k:
                chunk = conn.recv(k - len(buf))
                if not chunk:
                    raise RuntimeError("RcclComm: peer closed the connection")
                buf += chunk
            return buf
        return take(int.from_bytes(take(8), "little"))

    def _bcast_bytes(self, data):
        if self.world == 1:
            return data
        if self.rank == 0:
            self._serve(lambda c: self._send(c, data))
            return data
        with self._connect() as c:
            return self._recv(c)

    def allgather_object(self, obj):
        import pickle
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            parts = {0: obj}

            def take(c):
                r, o = pickle.loads(self._recv(c))
                parts[r] = o
            self._serve(take)
            out = [parts[r] for r in range(self.world)]
            blob = pickle.dumps(out)
            self._serve(lambda c: self._send(c, blob))
            return out
        with self._connect() as c:
            self._send(c, pickle.dumps((self.rank, obj)))
        with self._connect() as c:
            return pickle.loads(self._recv(c))

    def gather_rows(self, arr):
        return np.concatenate(self.allgather_object(np.ascontiguousarray(arr)), axis=0)

    # ---- the collective
    def allreduce_sum(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64).copy()
        self._lib.check(self._lib.load().wgs_comm_allreduce_f64(self._h, self._lib.f64p(a.reshape(-1)), a.size))
        return a

    def step_reduced(self, em_handle, n_fits):
        """EM sweep into the communicator's device buffer, all-reduce behind it on the same stream,
        one readback (the per-iteration exchange of the sharded EM; see EMBatch.step_reduced)."""
        lib = self._lib.load()
        buf = lib.wgs_comm_buffer(self._h, int(n_fits))
        if not buf:
            raise RuntimeError("wgsassign_amd HIP call failed: " + self._lib.last_error())
        self._lib.check(lib.wgs_em_step_dev(em_handle, self._ct.c_void_p(buf)))
        out = np.zeros(int(n_fits), dtype=np.float64)
        self._lib.check(lib.wgs_comm_allreduce_buffer(self._h, int(n_fits), self._lib.f64p(out)))
        return out

    def barrier(self):
        self.allreduce_sum(np.zeros(1))

    def close(self):
        if self._h:
            self._lib.load().wgs_comm_destroy(self._h)
            self._h = None


def init_from_env():
    """Process group from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*):
    RCCL (`nccl`) when a GPU per rank is available, `gloo` when WGSASSIGN_BACKEND=gloo (ranks
    sharing one GPU, CPU-only rehearsals).  Returns LocalComm() outside torchrun."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return LocalComm()
    if os.environ.get("WGSASSIGN_COMM", "torch") == "rccl":     # the library's own RCCL communicator
        from .device import get_context
        return RcclComm(get_context(), int(os.environ.get("RANK", "0")), world,
                        os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29400")))
    import torch
    import torch.distributed as dist
    backend = os.environ.get("WGSASSIGN_BACKEND", "nccl")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return TorchComm(device=torch.device("cuda", local_rank) if backend == "nccl" else None)


def shard_range(m_total, rank, world):
    """Contiguous SNP range [lo, hi) of `rank`: GPU g owns [g*m/G, (g+1)*m/G)."""
    lo = (m_total * rank) // world
    hi = (m_total * (rank + 1)) // world
    return lo, hi
