"""Observed Fisher information / effective sample sizes: drop-in for the reference's `fisher.py`
(`--ne_obs`), on the device slabs."""
import numpy as np

from . import _lib
from .device import AFSet, DeviceBeagle


def _slabs(L, IDs):
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    return DeviceBeagle.from_host(np.asarray(L), group_of, len(pops)), pops


def fisher_obs(L, af, IDs, t=1, beagle=None):
    """fisher.py:11-44: (f_obs, ne_obs), both (m, K) float32 -- per population the per-SNP sum over
    its individuals of the observed-information term (serial float32, file order) and
    0.5 * f * a * (1 - a)."""
    own = beagle is None
    if own:
        beagle, _ = _slabs(L, np.asarray(IDs))
    afs = AFSet.from_host(np.ascontiguousarray(af, dtype=np.float32), ctx=beagle.ctx)
    f_obs = np.empty((beagle.m, afs.K), dtype=np.float32)
    ne_obs = np.empty((beagle.m, afs.K), dtype=np.float32)
    _lib.check(_lib.load().wgs_fisher_obs(beagle.handle, afs.handle, _lib.f32p(f_obs), _lib.f32p(ne_obs)))
    afs.close()
    if own:
        beagle.close()
    return f_obs, ne_obs


def fisher_obs_ind(L, af, IDs, t=1, beagle=None, comm=None, m_total=None, exact_budget_bytes=8 << 30, host_mean=False):
    """fisher.py:46-60: per individual the mean over SNPs of its effective-sample-size term under
    its own population's frequencies.

    Single shard (default): the per-site float32 terms are computed on the device in batches of
    individuals and np.mean of each row is formed there exactly as NumPy forms it (pairwise float32
    summation, float64 division; csrc/em_kernels.hip: pairwise_leaf_kernel) -- bit-identical to the
    reference without moving n x m floats over PCIe; host_mean=True downloads the rows and calls np.mean
    itself (the cross-check).  SNP-sharded (comm given): the same running float32 total continued from shard to
    shard in SNP order (wgs_fisher_ind_sums) -- bit-identical as well."""
    own = beagle is None
    if own:
        beagle, _ = _slabs(L, np.asarray(IDs))
    afs = AFSet.from_host(np.ascontiguousarray(af, dtype=np.float32), ctx=beagle.ctx)
    lib = _lib.load()
    if comm is not None and comm.world > 1:
        # np.mean's running float32 total handed from shard to shard in SNP order (shards start at multiples of NumPy's
        # 8192-element chunks: comm.shard_range); the batches must be the same on every rank
        from .comm import SHARD_ALIGN, shard_range
        out = np.zeros(beagle.n, dtype=np.float32)
        group_of = beagle.group_of
        aligned = m_total // comm.world >= SHARD_ALIGN          # else the shards cut through NumPy's summation tree
        batch = int(max(1, min(256, exact_budget_bytes // max(1, 4 * (m_total // comm.world + 8192)))))
        if not aligned:
            batch = int(max(1, min(batch, (64 << 20) // max(1, 8 * m_total))))
        lo, hi = shard_range(m_total, comm.rank, comm.world)
        i = 0
        while i < beagle.n:
            j = i + 1
            while j < beagle.n and j - i < batch and group_of[j] == group_of[i]:
                j += 1
            if not aligned:
                # few sites (< 8192 per rank): the rows themselves are small -- every rank contributes its columns of the
                # (individuals x all sites) matrix and np.mean is applied to whole rows, as on one shard
                rows = np.empty((j - i, beagle.m), dtype=np.float32)
                _lib.check(lib.wgs_fisher_ind_sites(beagle.handle, afs.handle, i, j - i, _lib.f32p(rows)))
                full = np.zeros((j - i, m_total), dtype=np.float64)
                full[:, lo:hi] = rows
                full = comm.allreduce_sum(full).astype(np.float32)             # float32 values: exact in float64
                for r in range(j - i):
                    out[i + r] = out[i + r] + np.mean(full[r])                  # fisher.py:59
                i = j
                continue
            run = None
            for r in range(comm.world):
                mine = np.zeros(j - i, dtype=np.float32)
                if r == comm.rank:
                    _lib.check(lib.wgs_fisher_ind_sums(beagle.handle, afs.handle, i, j - i, _lib.f32p(run) if run is not None else None,
                                                       _lib.f32p(mine)))
                # only rank r contributes: a broadcast of float32 values (exact in float64)
                run = np.ascontiguousarray(comm.allreduce_sum(mine.astype(np.float64)).astype(np.float32))
            out[i:j] = (run.astype(np.float64) / m_total).astype(np.float32)      # np.mean: float64 division, float32 result
            i = j
    else:
        out = np.zeros(beagle.n, dtype=np.float32)
        m, group_of = beagle.m, beagle.group_of
        budget = min(exact_budget_bytes, 1 << 30) if host_mean else exact_budget_bytes       # host rows vs device workspace
        batch = int(max(1, min(256, budget // max(1, 4 * m))))
        i = 0
        while i < beagle.n:
            j = i + 1
            while j < beagle.n and j - i < batch and group_of[j] == group_of[i]:
                j += 1
            if host_mean:
                rows = np.empty((j - i, m), dtype=np.float32)
                _lib.check(lib.wgs_fisher_ind_sites(beagle.handle, afs.handle, i, j - i, _lib.f32p(rows)))
                for r in range(j - i):
                    out[i + r] = out[i + r] + np.mean(rows[r])        # fisher.py:59
            else:
                means = np.empty(j - i, dtype=np.float32)
                _lib.check(lib.wgs_fisher_ind_means(beagle.handle, afs.handle, i, j - i, _lib.f32p(means)))
                out[i:j] = out[i:j] + means                           # 0 + np.mean(...)
            i = j
    afs.close()
    if own:
        beagle.close()
    return out
