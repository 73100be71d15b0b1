"""EM allele-frequency driver: drop-in for the reference's `emMAF.py`."""
import numpy as np

from .device import DeviceBeagle, EMBatch


def emMAF(L, iter, tole, t=1):
    """emMAF.py:15-27: start at f = 0.25, update until rmse(f, f_prev) < tole or `iter` updates.

    L is the (m, 2*n_pop) float32 matrix of one population.  It is uploaded once, the whole loop
    runs on the device, and only the convergence sums cross to the host each iteration.
    Returns the UNclamped float32 frequencies; prints the reference's convergence line.
    """
    L = np.asarray(L)
    m = L.shape[0]
    if m == 0:
        return np.empty(0, dtype=np.float32)
    if L.shape[1] // 2 == 0:
        # emMAF_cy.pyx:17,23 with no individuals: tmp/(float)0 = NaN for every SNP, never converges
        return np.full(m, np.nan, dtype=np.float32)
    beagle = DeviceBeagle.from_host(L)
    em = EMBatch(beagle, [0])
    iters = em.run(iter, tole)
    if iters[0] > 0:
        print("EM (MAF) converged at iteration: " + str(int(iters[0])))
    f = em.get_f(0)
    em.close()
    beagle.close()
    return f


def emMAF_populations(L, IDs, iter, tole, beagle=None, comm=None):
    """The per-population loop of WGSassign.py:211-242 as ONE batch of EM chains.

    The reference gathers each population's columns on the host and runs emMAF on the copy;
    here L is permuted once into population slabs on the device and all K fits advance together,
    each freezing at its own convergence iteration.  Returns (pops sorted, af (m, K) float32
    clamped per WGSassign.py:236-240, iters (K,)).  Prints one convergence line per population,
    in population order, like the reference does.
    """
    L = np.asarray(L) if L is not None else None
    IDs = np.asarray(IDs)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:, 1]).astype(np.int32)
    counts = np.bincount(group_of, minlength=len(pops))
    own = beagle is None
    if own:
        beagle = DeviceBeagle.from_host(L, group_of, len(pops))
    em = EMBatch(beagle, np.arange(len(pops), dtype=np.int32))
    iters = em.run(iter, tole, comm=comm)
    try:        # (sweeps enqueued, exact-chain resolutions, wall seconds, sweep-kernel ms) of the one-call fit, for whoever asks
        emMAF_populations.last_stats = em.fit_stats()
    except Exception:
        emMAF_populations.last_stats = None
    af = np.empty((beagle.m, len(pops)), dtype=np.float32)
    for k in range(len(pops)):
        if iters[k] > 0:
            print("EM (MAF) converged at iteration: " + str(int(iters[k])))
        em.clamp(k, int(counts[k]))
        af[:, k] = em.get_f(k)
    em.close()
    if own:
        beagle.close()
    return pops, af, iters
