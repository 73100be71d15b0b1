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
    """torch.distributed process group (backend `nccl` == RCCL on ROCm, or `gloo` on CPU)."""

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


def init_from_env():
    """Process group from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*):
    RCCL (`nccl`) when a GPU per rank is available, `gloo` when WGSASSIGN_BACKEND=gloo (ranks
    sharing one GPU, CPU-only rehearsals).  Returns LocalComm() outside torchrun."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return LocalComm()
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
