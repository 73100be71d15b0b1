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


def fisher_obs_ind(L, af, IDs, t=1, beagle=None, comm=None, m_total=None):
    """fisher.py:46-60: per individual the mean over SNPs of its effective-sample-size term under
    its own population's frequencies (float64 sum on the device; the reference's np.mean
    accumulates in float32)."""
    own = beagle is None
    if own:
        beagle, _ = _slabs(L, np.asarray(IDs))
    afs = AFSet.from_host(np.ascontiguousarray(af, dtype=np.float32), ctx=beagle.ctx)
    sums = np.zeros(beagle.n, dtype=np.float64)
    _lib.check(_lib.load().wgs_fisher_obs_ind(beagle.handle, afs.handle, _lib.f64p(sums)))
    afs.close()
    m = beagle.m
    if comm is not None and comm.world > 1:
        sums = comm.allreduce_sum(sums)
        m = m_total
    if own:
        beagle.close()
    return (sums / m).astype(np.float32)
