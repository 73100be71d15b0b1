"""Assignment log-likelihoods and leave-one-out: drop-in for the reference's `glassy.py`."""
import numpy as np

import os

from .device import MAX_BLOCK_PARALLEL_PARTS, AFSet, DeviceBeagle, EMBatch, Score, assign, partition_sums_exact


def assignLL(L, af, t=1):
    """glassy.py:18-44: (n, K) float32 matrix of summed per-site log-likelihoods.

    One sweep over the device-resident matrix produces all n*K sums (the reference rescans L
    once per pair); sums are accumulated in float64 like np.sum(..., dtype=float) (glassy.py:38)
    and stored as float32 (glassy.py:42).
    """
    L = np.asarray(L)
    af = np.ascontiguousarray(af, dtype=np.float32)
    n = L.shape[1] // 2
    k = af.shape[1]
    print(str(n) + " individuals to assign to " + str(k) + " populations")
    if n == 0 or L.shape[0] == 0:
        return np.zeros((n, k), dtype=np.float32)
    beagle = DeviceBeagle.from_host(L)
    afset = AFSet.from_host(af[:L.shape[0]])
    out, _ = assign(beagle, afset)
    afset.close()
    beagle.close()
    with np.errstate(over="ignore"):
        return out.astype(np.float32)


def loo(L, af, IDs, t, maf_iter, maf_tole, downsampled_L=None, num_partitions=1, need_parts=True):
    """glassy.py:47-112: leave-one-out assignment log-likelihoods.

    Semantics kept from the reference:
      * individual i's own population column is re-estimated without i (glassy.py:65-78) and
        clamped with n_pop = |pop| - 1 (glassy.py:80-85);
      * that column OVERWRITES af[:, pop_col] and is never restored (glassy.py:87-89), so when i
        is scored every other column holds the leave-one-out estimate of the most recent earlier
        individual of that population (or the full-population estimate if there was none);
        `af` is mutated in place and ends up holding each population's LAST re-fit;
      * scoring uses downsampled_L when given (glassy.py:96-98);
      * per-partition sums use labels = site index % num_partitions (utils.py:147).
    All n re-fits run as one batch of EM chains on the device; one scoring sweep follows.
    Returns (logl_mat (n, K) float32, logl_parts_mat (n*P, K) float32).  The partition sums are the
    reference's serial float32 accumulations, bit for bit (WGSASSIGN_PARTS=fast: float64 sums, ~1e-5).
    need_parts=False with num_partitions == 1 skips that chain (the reference's CLI never writes the
    one-partition matrix) and returns the totals in its place.
    """
    L = np.asarray(L)
    IDs = np.asarray(IDs)
    m = L.shape[0]
    n = L.shape[1] // 2
    k = af.shape[1]
    P = int(num_partitions)
    print(str(n) + " individuals to assign to " + str(k) + " populations")
    if downsampled_L is not None:
        print("Using downsampled GLs for likelihood evaluation in LOO assignment.")
    if m == 0 or n == 0:
        # glassy.py:65-109 over no SNPs: every re-fit and every per-site vector is empty, np.sum([]) = 0.0
        return np.zeros((n, k), dtype=np.float32), np.zeros((n * P, k), dtype=np.float32)
    pops = np.unique(IDs[:, 1])
    group_of = np.searchsorted(pops, IDs[:n, 1]).astype(np.int32)
    beagle = DeviceBeagle.from_host(L, group_of, len(pops))
    scored = DeviceBeagle.from_host(np.asarray(downsampled_L), group_of, len(pops)) if downsampled_L is not None else beagle
    logl, logl_parts = loo_device(beagle, scored, af, group_of, maf_iter, maf_tole, P, need_parts=need_parts)
    if scored is not beagle:
        scored.close()
    beagle.close()
    return logl, logl_parts


def loo_column_table(afset, em, group_of, i0, i1):
    """(n, K) table of device addresses for scoring individuals [i0, i1): individual i is scored against
    its own re-fit (fit i - i0 of `em`) and, for every other population, the re-fit of the most recent
    earlier individual of that population, else the column of `afset` (glassy.py:87-105, the sticky
    overwrite).  Rows outside [i0, i1) hold valid placeholders."""
    n, k = len(group_of), afset.K
    cur = [afset.col_dev(j) for j in range(k)]
    colptr = np.empty((n, k), dtype=np.uint64)
    colptr[:] = cur
    for i in range(i0, i1):
        cur[group_of[i]] = em.f_dev(i - i0)
        colptr[i] = cur
    return colptr


def score_loo_batch(scored, afset, em, group_of, i0, i1, P=1, exact_parts=True, comm=None, timings=None):
    """Scoring step of glassy.py:92-109 for individuals [i0, i1) whose converged, clamped re-fits are the
    fits of `em`: returns (sums (n, K) float64, partition sums (n*P, K) or None).  Only rows [i0, i1) are
    computed (a batch scores its own individuals); the other rows are 0."""
    colptr = loo_column_table(afset, em, group_of, i0, i1)
    if not exact_parts:
        if P == 1:
            sc = Score(scored, afset, colptr, rows=(i0, i1))
            o = sc.sums(comm=comm)
            sc.close()
            return o, None
        return assign(scored, afset, colptr=colptr, P=P, comm=comm)      # float64 partition sums (WGSASSIGN_PARTS=fast)
    # sums over all sites in float64 (np.sum(dtype=float), glassy.py:101); partition sums literally as
    # utils.py:147-149 accumulates them (serial float32) -- for P == 1 too (glassy.py:108-109)
    from ._lib import MODE_EXACT
    if P > MAX_BLOCK_PARALLEL_PARTS:
        o, _ = assign(scored, afset, colptr=colptr, P=1, comm=comm)
        return o, partition_sums_exact(scored, afset, colptr=colptr, P=P, comm=comm)
    sc = Score(scored, afset, colptr, rows=(i0, i1))
    o = sc.sums(MODE_EXACT, comm)
    pr = sc.parts_exact(P, comm)
    if timings is not None:
        for k, v in sc.ms.items():
            timings[k + "_ms"] = timings.get(k + "_ms", 0.0) + v
        timings["serial_blocks"] = sc.serial_blocks()
    sc.close()
    return o, pr


def loo_device(beagle, scored, af, group_of, maf_iter, maf_tole, P=1, comm=None, verbose=True, timings=None,
               need_parts=True, inspect=None):
    """The body of loo() on device-resident matrices: `beagle` holds the GLs the frequencies are
    re-estimated from, `scored` the GLs that are scored (the same object unless a downsampled
    matrix is given), both with population slabs `group_of`.  `af` (m, K) float32 is mutated like
    glassy.py:87-89 does.  With `comm`, SNPs are sharded over ranks (af is this rank's shard).
    inspect(em, i0, i1), if given, is called with each batch's converged and clamped fits."""
    import time
    n, k = beagle.n, af.shape[1]
    counts = np.bincount(group_of, minlength=k)
    exact_parts = os.environ.get("WGSASSIGN_PARTS", "exact") != "fast" and (need_parts or P > 1)
    handle = getattr(comm, "handle", None) if comm is not None and comm.world > 1 else None
    one_call = (comm is None or comm.world == 1 or handle is not None) and inspect is None \
        and (exact_parts or P == 1) and P <= MAX_BLOCK_PARALLEL_PARTS and os.environ.get("WGSASSIGN_LOO", "c") != "python"
    if one_call:
        # the whole of glassy.py:65-109 behind one C entry (wgs_loo); the Python orchestration below does the
        # same through the step-wise entry points and serves the other communicators (gloo / socket)
        import ctypes
        from . import _lib
        from .device import default_em_mode, default_mode
        t0 = time.perf_counter()
        afset = AFSet.from_host(np.ascontiguousarray(af, dtype=np.float32))
        m_total = int(comm.allreduce_sum(np.array([float(beagle.m)]))[0]) if handle is not None else beagle.m
        out = np.zeros((n, k), dtype=np.float64)
        parts = np.zeros((n * P, k), dtype=np.float32) if exact_parts else None
        iters = np.zeros(n, dtype=np.int32)
        _lib.check(_lib.load().wgs_loo(beagle.handle, scored.handle if scored is not beagle else None, afset.handle,
                                       int(maf_iter), float(maf_tole), m_total, handle, P,
                                       int(os.environ.get("WGSASSIGN_LOO_BATCH", 0)), default_em_mode(), default_mode(), _lib.f64p(out),
                                       _lib.f32p(parts) if parts is not None else None, _lib.i32p(iters)))
        if verbose:
            for i in range(n):
                if iters[i] > 0:
                    print("EM (MAF) converged at iteration: " + str(int(iters[i])))
        af[:, :] = afset.to_host()
        afset.close()
        if timings is not None:
            st = (ctypes.c_double * 7)()
            _lib.check(_lib.load().wgs_loo_stats(st))
            timings.update(seconds=time.perf_counter() - t0, iters=iters, one_call=True, em_seconds=st[0], score_seconds=st[1],
                           chain_seconds=st[2], em_sweep_kernel_ms=st[3], em_batches=int(st[4]), em_chain_resolutions=int(st[5]),
                           em_iterations_enqueued=int(st[6]))
        with np.errstate(over="ignore"):
            logl = out.astype(np.float32)
        return logl, (parts if parts is not None else logl.copy())
    # The n re-fits need 2 float32 vectors + the per-tile partial sums each (~8.2 bytes per SNP and fit).
    # They run as ONE batch when that fits the free device memory, else in file-order batches: the
    # "current" columns (afset) carry the sticky overwrite from batch to batch.
    batch = loo_batch_size(beagle, n, comm)
    afset = AFSet.from_host(np.ascontiguousarray(af, dtype=np.float32))
    out = np.zeros((n, k), dtype=np.float64)
    parts = np.zeros((n * P, k), dtype=np.float64 if not exact_parts else np.float32) if (P > 1 or exact_parts) else None
    iters = np.zeros(n, dtype=np.int32)
    t_em = t_score = 0.0
    for i0 in range(0, n, batch):
        i1 = min(n, i0 + batch)
        t0 = time.perf_counter()
        em = EMBatch(beagle, group_of[i0:i1], np.arange(i0, i1, dtype=np.int32))
        iters[i0:i1] = em.run(maf_iter, maf_tole, comm=comm)
        beagle.ctx.sync()
        t1 = time.perf_counter()
        for i in range(i0, i1):
            if verbose and iters[i] > 0:
                print("EM (MAF) converged at iteration: " + str(int(iters[i])))
            em.clamp(i - i0, int(counts[group_of[i]]) - 1)
        if inspect is not None:
            inspect(em, i0, i1)
        o, pr = score_loo_batch(scored, afset, em, group_of, i0, i1, P, exact_parts, comm, timings)
        out[i0:i1] = o[i0:i1]
        if parts is not None and pr is not None:
            parts[i0 * P:i1 * P] = pr[i0 * P:i1 * P]
        # the last re-fit of each population in this batch becomes the current column
        last = {int(g): i for i, g in zip(range(i0, i1), group_of[i0:i1])}
        for g, i in last.items():
            afset.set_column_from_em(g, em, i - i0)
        beagle.ctx.sync()
        em.close()
        t_em += t1 - t0
        t_score += time.perf_counter() - t1
    af[:, :] = afset.to_host()           # glassy.py:89: the caller's af ends up holding each population's last re-fit
    afset.close()
    if timings is not None:
        timings.update(em_seconds=t_em, score_seconds=t_score, iters=iters, batch=batch)
    with np.errstate(over="ignore"):
        logl = out.astype(np.float32)
        logl_parts = parts.astype(np.float32) if parts is not None else logl.copy()
    return logl, logl_parts


def loo_batch_size(beagle, n, comm=None):
    """Number of leave-one-out re-fits per EM batch: what fits the free device memory (or
    WGSASSIGN_LOO_BATCH), agreed across SNP-shard ranks -- every rank must build batches of the same
    fits or the per-iteration all-reduces stop matching -- by taking the minimum over ranks."""
    free_bytes, _ = beagle.ctx.mem_info()
    per_fit = int(beagle.m * 8.2) + 4096
    batch = int(os.environ.get("WGSASSIGN_LOO_BATCH", max(1, min(n, int(0.8 * free_bytes) // per_fit))))
    batch = max(1, min(n, batch))
    if comm is not None and comm.world > 1:
        slots = np.zeros(comm.world)
        slots[comm.rank] = batch
        batch = int(np.min(comm.allreduce_sum(slots)))
    return batch
