"""Device-resident objects of the HIP path and the host-side EM driver loop.

Host code is Python/NumPy; all arithmetic of the hot path runs in the HIP kernels behind
include/wgsassign_hip.h.  SNP sharding over ranks needs exactly one collective (a sum
all-reduce of a few float64), supplied by a `comm` object (wgsassign_amd/comm.py).
"""
import atexit
import ctypes
import math
import os
import weakref

import numpy as np

from . import _lib
from ._lib import MODE_EXACT, MODE_FAST, check, f32p, f64p, i32p

_default_ctx = None
_live = weakref.WeakSet()       # device objects that still own library handles


@atexit.register
def _close_all():
    """Release every device object before the interpreter tears modules down: children (scores, EM batches)
    before the matrices they refer to, so no destructor runs against a freed parent or an unloaded library."""
    order = {"Score": 0, "EMBatch": 1, "AFSet": 2, "DeviceBeagle": 3}
    for obj in sorted(list(_live), key=lambda o: order.get(type(o).__name__, 9)):
        try:
            obj.close()
        except Exception:
            pass



def _mode_from(var):
    m = os.environ.get(var, "exact").lower()
    if m not in ("exact", "fast"):
        raise ValueError("%s must be 'exact' or 'fast', got %r" % (var, m))
    return MODE_EXACT if m == "exact" else MODE_FAST


def default_mode():
    """Arithmetic of the SCORING sweep (WGSASSIGN_MODE): exact (default; per-site values bit-identical to the
    reference) or fast (float32 evaluation, hardware log: n x K sums within 1.2e-7 relative of exact at 10M x 1000 x
    K=10 and 2M x 500 x K=8 -- tools/check_fast_mode.py -- inside the 1e-6 bar, 2.9x faster)."""
    return _mode_from("WGSASSIGN_MODE")


def default_em_mode():
    """Arithmetic of the EM update (WGSASSIGN_EM_MODE, default exact).  The float32 update keeps the reference's
    iteration counts but its frequencies drift up to 7e-6 relative at 10M SNPs (0.1 % of them beyond 1e-6): outside
    the 1e-6 bar, so it is an explicit opt-in and never implied by WGSASSIGN_MODE=fast."""
    return _mode_from("WGSASSIGN_EM_MODE")


def _as_f32c(a, name):
    a = np.asarray(a)
    if a.dtype != np.float32 or not a.flags.c_contiguous:
        raise ValueError("%s must be a C-contiguous float32 array" % name)
    return a


def default_device_index():
    """The GPU of this process: WGSASSIGN_DEVICE if set, else LOCAL_RANK.  When exactly ONE device is visible -- a launcher that
    narrows every rank's view to its own GPU (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank) and still sets LOCAL_RANK =
    0 .. N-1 -- that device is it.  With several devices visible a LOCAL_RANK beyond them is an error: taken modulo the count it
    would put two ranks on one GPU without a word."""
    if "WGSASSIGN_DEVICE" in os.environ:
        return int(os.environ["WGSASSIGN_DEVICE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = ctypes.c_int(0)
    if _lib.load().wgs_device_count(ctypes.byref(n)) == 0 and n.value > 0:
        if n.value == 1:
            return 0
        if local_rank >= n.value:
            raise RuntimeError("wgsassign_amd: LOCAL_RANK=%d but only %d GPUs are visible to this process (set WGSASSIGN_DEVICE to share "
                               "a device deliberately)" % (local_rank, n.value))
    return local_rank


class Context:
    """One HIP device + stream (wgs_ctx)."""

    def __init__(self, device=None):
        lib = _lib.load()
        if device is None:
            device = default_device_index()
        h = ctypes.c_void_p()
        check(lib.wgs_ctx_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    def sync(self):
        check(_lib.load().wgs_ctx_sync(self._h))

    def stream_ptr(self):
        """The context's hipStream_t as an integer (for torch.cuda.ExternalStream)."""
        return int(_lib.load().wgs_ctx_stream(self._h) or 0)

    def mem_info(self):
        """(free, total) device memory in bytes."""
        free, total = ctypes.c_int64(), ctypes.c_int64()
        check(_lib.load().wgs_ctx_mem_info(self._h, ctypes.byref(free), ctypes.byref(total)))
        return free.value, total.value

    def info(self):
        name = ctypes.create_string_buffer(256)
        cus = ctypes.c_int()
        mem = ctypes.c_int64()
        check(_lib.load().wgs_ctx_info(self._h, name, 256, ctypes.byref(cus), ctypes.byref(mem)))
        return {"name": name.value.decode(), "cus": cus.value, "mem_bytes": mem.value}

    def close(self):
        if self._h:
            _lib.load().wgs_ctx_destroy(self._h)
            self._h = None


def get_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


class DeviceBeagle:
    """Genotype-likelihood matrix (one SNP shard) as population slabs in HBM (wgs_beagle).

    group_of: int array (n,) with the group (population column) of every individual, or None for
    a single group.  Groups keep individuals in file order, as the reference's sorted column
    gather does (WGSassign.py:227-233).
    """

    def __init__(self, m, n, group_of=None, n_groups=1, site0=0, ctx=None):
        self.ctx = ctx or get_context()
        self.m, self.n, self.site0 = int(m), int(n), int(site0)
        if group_of is not None:
            group_of = np.ascontiguousarray(group_of, dtype=np.int32)
            if group_of.shape != (self.n,):
                raise ValueError("group_of must have one entry per individual")
            n_groups = int(n_groups)
            gp = i32p(group_of)
        else:
            n_groups, gp = 1, None
        self.group_of = group_of if group_of is not None else np.zeros(self.n, dtype=np.int32)
        self.n_groups = n_groups
        h = ctypes.c_void_p()
        check(_lib.load().wgs_beagle_create(self.ctx.handle, self.m, self.n, gp, n_groups, self.site0, ctypes.byref(h)))
        self._h = h
        self._children = weakref.WeakSet()      # EM batches and scores over this matrix: closed before it, whatever order a collector picks
        _live.add(self)

    @classmethod
    def from_host(cls, L, group_of=None, n_groups=1, site0=0, ctx=None):
        L = _as_f32c(L, "L")
        if L.ndim != 2:
            raise ValueError("L must be 2-dimensional (m, 2n)")
        b = cls(L.shape[0], L.shape[1] // 2, group_of, n_groups, site0, ctx)
        b.upload_rows(L, 0)
        return b

    @property
    def handle(self):
        return self._h

    def upload_rows(self, rows, row0=0):
        rows = _as_f32c(rows, "rows")
        if rows.ndim != 2 or rows.shape[1] < 2 * self.n:
            raise ValueError("rows must be (nrows, 2n)")
        if rows.shape[1] != 2 * self.n:      # odd trailing column is ignored like n = L.shape[1]//2
            rows = np.ascontiguousarray(rows[:, :2 * self.n])
        check(_lib.load().wgs_beagle_upload_rows(self._h, f32p(rows), int(row0), rows.shape[0]))

    def download_rows(self, row0, nrows):
        out = np.empty((int(nrows), 2 * self.n), dtype=np.float32)
        check(_lib.load().wgs_beagle_download_rows(self._h, f32p(out), int(row0), int(nrows)))
        return out

    def synth(self, seed, depth=2.0):
        check(_lib.load().wgs_beagle_synth(self._h, int(seed), float(depth)))

    def synth_quality(self, seed, depth=2.0, quals=(12, 23, 37), probs=(0.03, 0.12, 0.85)):
        """Synthetic matrix with quality-dependent likelihoods (every read its own error rate, drawn from the given bins)."""
        q = np.ascontiguousarray(quals, dtype=np.float64)
        p = np.ascontiguousarray(probs, dtype=np.float64)
        check(_lib.load().wgs_beagle_synth_quality(self._h, int(seed), float(depth), len(q), f64p(q), f64p(p)))

    def nbytes(self):
        return int(_lib.load().wgs_beagle_bytes(self._h))

    def set_rows(self, m):
        """A matrix created with room for more sites than its file held (reader_cy.estimate_sites) becomes one of m sites
        (include/wgsassign_hip.h: wgs_beagle_set_rows) -- before any EM batch or score is made from it."""
        check(_lib.load().wgs_beagle_set_rows(self._h, int(m)))
        self.m = int(m)

    def prepare_codes(self, em=True):
        """Build the class codes now (em: and the slabs' own numbering for the coded EM sweep) instead of at first use."""
        check(_lib.load().wgs_beagle_codes_prepare(self._h, 1 if em else 0))

    def codes_state(self):
        """1: the class codes exist, 0: nothing has asked for them yet, -1: not worth coding.  Builds nothing."""
        return int(_lib.load().wgs_beagle_codes_state(self._h))

    def codes_wait(self):
        """Wait for the codes' device memory if its allocation is under way on the library's helper thread (sweeps do not wait for it:
        they run over the float32 slabs meanwhile).  Returns the milliseconds that hipMalloc took (0.0: none was in flight)."""
        ms = ctypes.c_double()
        check(_lib.load().wgs_beagle_codes_wait(self._h, ctypes.byref(ms)))
        return ms.value

    def codes_model(self, K_score=0):
        """What the library's cost models predict for this matrix (wgs_codes_model): the numbers its decisions are made from."""
        o = (ctypes.c_double * 12)()
        check(_lib.load().wgs_codes_model(self._h, int(K_score), o))
        return {"em_float32_sweep_ms": o[0], "em_share_saved_by_a_coded_sweep": o[1], "encode_ms": o[2], "sweeps_counted": o[3],
                "builds_for_a_fit": bool(o[4]), "score_float32_sweep_ms": o[5], "score_share_of_the_coded_sweep": o[6],
                "encode_for_scoring_ms": o[7], "builds_for_scoring": bool(o[8]), "from_the_sample_pass": bool(o[9]),
                "sample_classes_per_slab_and_snp": o[10], "sample_classes_per_snp": o[11]}

    def codes_info(self):
        """Class codes of the matrix (built on first use; csrc/common.h: wgs_codes, csrc/codes.hip): what they hold and what
        they cost.  `build_ms` is the whole build (sample pass, one allocation, the encode pass); `rich_snp_share` the SNPs
        left uncoded (too many classes: every sweep takes them from the float32 slab), `em_direct_tile_share` the (slab, tile)
        pairs the coded EM sweep takes from the float32 slab."""
        info = (ctypes.c_double * 20)()
        check(_lib.load().wgs_beagle_codes_info(self._h, info))
        return {"available": bool(info[0]), "max_classes": int(info[1]), "bytes": int(info[2]), "build_ms": info[3],
                "encode_kernel_ms": info[5], "mean_classes": info[4], "sample_ms": info[6], "slab_numbering_bytes": int(info[7]),
                "em_table_rows": int(info[8]), "em_direct_tile_share": info[9], "hash_slots": int(info[10]), "rich_snp_share": info[11],
                "dict_rows": int(info[12]), "probe_rounds_per_buffer": info[13], "alloc_ms": info[14], "score_table_rows": int(info[15]),
                "sample_mean_classes": info[16], "sample_mean_classes_per_slab": info[17], "score_batch_snps": int(info[18]),
                "alloc_wait_ms": info[19]}

    def close(self):
        if self._h:
            for child in list(self._children):
                child.close()
            _lib.load().wgs_beagle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AFSet:
    """K allele-frequency vectors of m SNPs on the device (the (m, K) `.pop_af.npy` matrix)."""

    def __init__(self, m, K, ctx=None):
        self.ctx = ctx or get_context()
        self.m, self.K = int(m), int(K)
        h = ctypes.c_void_p()
        check(_lib.load().wgs_afset_create(self.ctx.handle, self.m, self.K, ctypes.byref(h)))
        self._h = h
        self._children = weakref.WeakSet()      # scores over these columns
        _live.add(self)

    @classmethod
    def from_host(cls, A, ctx=None):
        A = _as_f32c(A, "af")
        a = cls(A.shape[0], A.shape[1], ctx)
        check(_lib.load().wgs_afset_upload(a._h, f32p(A)))
        return a

    @property
    def handle(self):
        return self._h

    def to_host(self):
        out = np.empty((self.m, self.K), dtype=np.float32)
        check(_lib.load().wgs_afset_download(self._h, f32p(out)))
        return out

    def set_column_from_em(self, col, em, fit):
        check(_lib.load().wgs_afset_set_column_from_em(self._h, int(col), em.handle, int(fit)))

    def col_dev(self, col):
        return _lib.load().wgs_afset_col_dev(self._h, int(col))

    def close(self):
        if self._h:
            for child in list(self._children):
                child.close()
            _lib.load().wgs_afset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EMBatch:
    """A batch of EM fits over the slabs of one DeviceBeagle (wgs_em).

    fit j uses the individuals of group groups[j], leaving out individual skips[j] (global
    index) when >= 0 -- the leave-one-out re-fit of glassy.py:65-78.
    """

    GUARD = 0.0    # floor of the band in which the exact serial chain decides (see guard_band)

    def __init__(self, beagle, groups, skips=None, mode=None):
        self.b = beagle
        self.n_fits = len(groups)
        self.mode = default_em_mode() if mode is None else mode
        g = np.ascontiguousarray(groups, dtype=np.int32)
        s = np.ascontiguousarray(skips if skips is not None else np.full(self.n_fits, -1), dtype=np.int32)
        h = ctypes.c_void_p()
        check(_lib.load().wgs_em_create(beagle.handle, self.n_fits, i32p(g), i32p(s), self.mode, ctypes.byref(h)))
        self._h = h
        beagle._children.add(self)
        _live.add(self)
        self.active = np.ones(self.n_fits, dtype=bool)

    @property
    def handle(self):
        return self._h

    # ---- primitives (one C-ABI call each)
    def step(self):
        """One EM update of every active fit (emMAF_cy.pyx:10-23); returns the float64 sums of
        (f_new - f_old)^2 over this shard's SNPs per fit."""
        ssq = np.zeros(self.n_fits, dtype=np.float64)
        check(_lib.load().wgs_em_step(self._h, f64p(ssq)))
        return ssq

    def step_reduced(self, comm):
        """step() followed by the sum over SNP shards.  With an RCCL communicator the per-fit sums
        stay on the device: the sweep writes them into a device buffer (wgs_em_step_dev), the
        all-reduce is enqueued behind it on the same stream, one readback ends the iteration."""
        if comm is None or comm.world == 1 and not getattr(comm, "force_device", False):
            return self.step()
        if hasattr(comm, "step_reduced"):      # library-native RCCL: sweep, all-reduce and readback on one stream
            return comm.step_reduced(self._h, self.n_fits)
        buf = getattr(self, "_ssq_buf", None)
        if buf is None and hasattr(comm, "device_buffer"):
            buf = self._ssq_buf = comm.device_buffer(self.n_fits)
        if buf is None:
            return comm.allreduce_sum(self.step())
        check(_lib.load().wgs_em_step_dev(self._h, ctypes.c_void_p(buf.data_ptr())))
        return comm.allreduce_device(buf, self.b.ctx.stream_ptr())

    def last_sweep_ms(self):
        ms = ctypes.c_float()
        check(_lib.load().wgs_em_last_sweep_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def rmse_chain(self, fit, carry_in):
        out = ctypes.c_float()
        check(_lib.load().wgs_em_rmse_chain(self._h, int(fit), ctypes.c_float(carry_in), ctypes.byref(out)))
        return np.float32(out.value)

    def set_active(self, fit, active):
        check(_lib.load().wgs_em_set_active(self._h, int(fit), int(bool(active))))
        self.active[fit] = bool(active)

    def clamp(self, fit, n_pop):
        """WGSassign.py:236-240 / glassy.py:80-85."""
        lo = 1 / (2 * (n_pop + 1))
        hi = 1 - lo
        check(_lib.load().wgs_em_clamp(self._h, int(fit), np.float32(lo), np.float32(hi)))

    def get_f(self, fit):
        out = np.empty(self.b.m, dtype=np.float32)
        check(_lib.load().wgs_em_get_f(self._h, int(fit), f32p(out)))
        return out

    def get_f_range(self, fit, row0, nrows, previous=False):
        """Rows [row0, row0+nrows) of the current (or previous-iteration) frequencies of a fit."""
        out = np.empty(int(nrows), dtype=np.float32)
        check(_lib.load().wgs_em_get_f_range(self._h, int(fit), int(bool(previous)), int(row0), int(nrows), f32p(out)))
        return out

    def set_f(self, fit, f):
        f = _as_f32c(f, "f")
        check(_lib.load().wgs_em_set_f(self._h, int(fit), f32p(f)))

    def f_dev(self, fit):
        return _lib.load().wgs_em_f_dev(self._h, int(fit))

    # ---- the driver loop of emMAF.py:20-26, for all fits at once
    def run(self, max_iter, tole, comm=None, m_total=None):
        """Returns iters (n_fits,), the 1-based iteration at which each fit met `diff < tole`
        (0: max_iter exhausted -- the reference prints nothing then).  One C call (wgs_em_fit: iterations
        enqueued ahead of the host, decisions on the device) for one shard or RCCL shards; the
        step-by-step protocol of run_em for the other communicators (gloo / socket rehearsals) or when
        WGSASSIGN_EM_LOOP=python."""
        native = comm is None or comm.world == 1 or getattr(comm, "handle", None) is not None
        if native and os.environ.get("WGSASSIGN_EM_LOOP", "c") != "python":
            return self.fit(max_iter, tole, comm, m_total)
        return run_em(self, max_iter, tole, comm, m_total)

    def fit(self, max_iter, tole, comm=None, m_total=None):
        """wgs_em_fit: emMAF.py:15-27 for every fit in one call."""
        lib = _lib.load()
        forced = comm is not None and getattr(comm, "force_device", False)     # bench.py: rehearse the collective on one GPU
        handle = getattr(comm, "handle", None) if comm is not None and (comm.world > 1 or forced) else None
        if m_total is None:
            m_total = int(comm.allreduce_sum(np.array([float(self.b.m)]))[0]) if handle is not None else self.b.m
        iters = np.zeros(self.n_fits, dtype=np.int32)
        check(lib.wgs_em_fit(self._h, int(max_iter), float(tole), int(m_total), handle, float(self.GUARD), i32p(iters)))
        self.active = self.active & (iters == 0)
        return iters

    def fit_stats(self):
        """(iterations enqueued, batched exact-chain resolutions, wall seconds, summed sweep-kernel ms) of the last fit()."""
        it, ch, sec, ms = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
        check(_lib.load().wgs_em_fit_stats(self._h, ctypes.byref(it), ctypes.byref(ch), ctypes.byref(sec), ctypes.byref(ms)))
        return it.value, ch.value, sec.value, ms.value

    def close(self):
        if self._h:
            _lib.load().wgs_em_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def guard_band(m_total, guard=0.0):
    """Relative half-width of the band around tole^2 * m in which the float64 sum cannot decide for the
    reference's SERIAL FLOAT32 sum (emMAF_cy.pyx:30-31).  Every one of the m additions rounds by at most
    half an ulp of its result, i.e. by <= F * 2^-24 with F the final float32 sum, so
    |S - F| <= m * 2^-24 * F rigorously (S: exact sum): the band must grow with m -- at 10^7 SNPs it is
    +-0.6, from 1.7*10^7 SNPs on the float32 sum may sit arbitrarily far BELOW the exact one (small terms
    are absorbed) and only the upper edge remains.  `guard` is a floor (tests force the chain with 1e9);
    1e-6 covers the float32 divide by m and the float32 rounding of m itself."""
    return max(float(guard), float(m_total) * 2.0 ** -24) + 1e-6


def decide_converged(ssq, m_total, tole, guard=0.0):
    """Classify a float64 sum of squared differences against the reference's test
    `sqrt(float32_serial_sum / float32(m)) < tole` (emMAF_cy.pyx:26-33, emMAF.py:22-23):
    returns +1 converged, -1 not converged, 0 too close to call (needs the exact chain).
    NaN never converges (NaN < tole is False)."""
    if ssq != ssq or not tole > 0:
        return -1
    thresh = tole * tole * float(m_total)
    g = guard_band(m_total, guard)
    # F <= S / (1 - g) (g < 1) and F >= S / (1 + g)
    if g < 1.0 and ssq < thresh * (1.0 - g):
        return 1
    if ssq >= thresh * (1.0 + g):
        return -1
    return 0


def chain_diff(carry, m_total):
    """emMAF_cy.pyx:32-33: res / (float)n, then sqrt in double."""
    res = np.float32(carry) / np.float32(m_total)
    return math.sqrt(float(res))


def run_em(em, max_iter, tole, comm=None, m_total=None):
    """emMAF.py:20-26 for a batch of fits sharded by SNP over comm.world ranks.

    `em` needs: n_fits, active (bool array), step() -> ssq, rmse_chain(fit, carry_in) -> float32,
    set_active(fit, bool), GUARD.  (tests drive this loop with a CPU stand-in to cover the
    multi-rank protocol without a GPU.)
    """
    from .comm import LocalComm
    comm = comm or LocalComm()
    if m_total is None:
        m_total = int(comm.allreduce_sum(np.array([float(em.b.m)]))[0]) if comm.world > 1 else em.b.m
    iters = np.zeros(em.n_fits, dtype=np.int32)
    for it in range(1, int(max_iter) + 1):
        if not em.active.any():
            break
        if hasattr(em, "step_reduced"):
            ssq = em.step_reduced(comm)
        else:
            ssq = em.step()
            if comm.world > 1:
                ssq = comm.allreduce_sum(ssq)
        undecided = []
        for j in np.flatnonzero(em.active):
            d = decide_converged(ssq[j], m_total, tole, em.GUARD)
            if d > 0:
                iters[j] = it
                em.set_active(j, False)
            elif d == 0:
                undecided.append(int(j))
        for j in undecided:
            # the reference's float32 running sum walks all SNPs in index order: rank r continues
            # from the carry of rank r-1 (contiguous SNP shards in rank order)
            carry = np.float32(0.0)
            for r in range(comm.world):
                mine = np.zeros(1, dtype=np.float64)
                if r == comm.rank:
                    mine[0] = float(em.rmse_chain(j, carry))
                if comm.world > 1:
                    mine = comm.allreduce_sum(mine)
                carry = np.float32(mine[0])
            if chain_diff(carry, m_total) < tole:
                iters[j] = it
                em.set_active(j, False)
    return iters


def _colptr_arg(colptr, n, K):
    if colptr is None:
        return None, None
    arr = np.ascontiguousarray(colptr, dtype=np.uint64)
    if arr.shape != (n, K):
        raise ValueError("colptr must be (n, K)")
    return arr, arr.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p))


def assign(beagle, afset, colptr=None, P=1, mode=None, comm=None):
    """All n x K assignment log-likelihood sums in one sweep (glassy.py:18-44 / 87-109).

    Returns (out (n, K) float64, parts (n*P, K) float64 or None), already summed over ranks.
    colptr: optional (n, K) array of device addresses (per-individual frequency vectors).
    P > 1 also returns FLOAT64 partition sums through the library's cross-check entry (wgs_debug_assign_parts_f64:
    float64 atomics, ~1e-5 from the reference's serial float32 partition sums -- what WGSASSIGN_PARTS=fast selects and the
    tests compare against); the reference's own partition sums come from partition_sums_exact / Score.
    """
    mode = default_mode() if mode is None else mode
    n, K = beagle.n, afset.K
    if P == 1 and comm is not None and comm.world > 1:       # the totals in NumPy's order across the shards (Score.sums)
        sc = Score(beagle, afset, colptr)
        try:
            out = sc.sums(mode, comm)
            assign.last_ms = sc.ms["sweep"]
            return out, None
        finally:
            sc.close()
    out = np.zeros((n, K), dtype=np.float64)
    parts = np.zeros((n * P, K), dtype=np.float64) if P > 1 else None
    _keep, cp = _colptr_arg(colptr, n, K)
    if P == 1:
        check(_lib.load().wgs_assign(beagle.handle, afset.handle, cp, mode, f64p(out)))
    else:
        check(_lib.load().wgs_debug_assign_parts_f64(beagle.handle, afset.handle, cp, int(P), mode, f64p(out), f64p(parts)))
    assign.last_ms = last_assign_ms(beagle.ctx)
    if comm is not None and comm.world > 1:
        out = comm.allreduce_sum(out)
        if parts is not None:
            parts = comm.allreduce_sum(parts)
    return out, parts


def debug_hook(name, value):
    """A named process-wide test switch of the library (include/wgsassign_hip_debug.h: wgs_debug_hook); 0 switches it off."""
    check(_lib.load().wgs_debug_hook(name.encode(), int(value)))


def malloc_seconds():
    """Seconds this process has spent inside hipMalloc through the library so far (include/wgsassign_hip.h: wgs_malloc_seconds)."""
    return float(_lib.load().wgs_malloc_seconds())


def last_assign_ms(ctx):
    """Kernel time of the context's last scoring call (HIP events on its stream)."""
    ms = ctypes.c_float()
    check(_lib.load().wgs_assign_last_ms(ctx.handle, ctypes.byref(ms)))
    return ms.value


MAX_BLOCK_PARALLEL_PARTS = 64     # beyond this the chains are many and short: literal one-lane chains


class Score:
    """The scoring step of glassy.py:31-42 / 92-109 on device-resident data (wgs_score): the n x K float64
    sums of the individuals [rows) in one reproducible sweep, and the reference's serial float32
    partition sums (utils.py:147-149) from block functions prepared in parallel on every SNP shard."""

    def __init__(self, beagle, afset, colptr=None, rows=None):
        self.b, self.K, self.n = beagle, afset.K, beagle.n
        lo, hi = (0, self.n) if rows is None else (int(rows[0]), int(rows[1]))
        self.rows = (lo, hi)
        self._keep, cp = _colptr_arg(colptr, self.n, self.K)
        h = ctypes.c_void_p()
        check(_lib.load().wgs_score_create(beagle.handle, afset.handle, cp, lo, hi, ctypes.byref(h)))
        self._h = h
        beagle._children.add(self)
        afset._children.add(self)
        _live.add(self)
        self.local = None
        self.ms = {}

    def sums(self, mode=None, comm=None):
        """(n, K) float64 sums over all SNPs (all ranks); rows outside `rows` are 0."""
        mode = default_mode() if mode is None else mode
        out = np.zeros((self.n, self.K), dtype=np.float64)
        check(_lib.load().wgs_score_sums(self._h, mode, f64p(out)))
        self.ms["sweep"] = last_assign_ms(self.b.ctx)
        self.local, self.mode = out, mode
        if comm is not None and comm.world > 1:
            handle = getattr(comm, "handle", None)
            if handle is not None:
                # np.sum's running float64 total (glassy.py:38) handed from shard to shard in SNP order ON THE STREAM
                # (wgs_score_totals_all: `world` broadcasts, one readback); `before` = the total of the shards before this
                # one, from which this rank predicts its chains
                run, before = np.zeros_like(out), np.zeros_like(out)
                check(_lib.load().wgs_score_totals_all(self._h, handle, f64p(run), f64p(before)))
                self.before = before if comm.rank > 0 else None
                return run
            # communicators without a library handle (torch.distributed): the same hand-over through host all-reduces
            slots = np.zeros((comm.world,) + out.shape)
            slots[comm.rank] = out
            by_rank = comm.allreduce_sum(slots)
            self.before = np.ascontiguousarray(by_rank[:comm.rank].sum(axis=0)) if comm.rank > 0 else None
            run = None
            for r in range(comm.world):
                mine = np.zeros_like(out)
                if r == comm.rank:
                    check(_lib.load().wgs_score_total_from(self._h, f64p(run) if run is not None else None, f64p(mine)))
                run = np.ascontiguousarray(comm.allreduce_sum(mine))      # only rank r contributes: a broadcast
            return run
        self.before = None
        return out

    def parts_exact(self, P, comm=None):
        """(n*P, K) float32: utils.partition_loglikes for every (individual, population), bit-exact.
        Needs sums(MODE_EXACT) first.  Every rank prepares its block functions at once; only the walk
        (milliseconds) follows the previous shard's float32 carries."""
        if self.local is None or self.mode != MODE_EXACT:
            self.sums(MODE_EXACT, comm)
        lib = _lib.load()
        world = comm.world if comm is not None else 1
        rank = comm.rank if comm is not None else 0
        start = self.before
        check(lib.wgs_score_chains_prepare(self._h, int(P), f64p(start) if start is not None else None))
        self.ms["chains"] = last_assign_ms(self.b.ctx)
        parts = np.zeros((self.n * P, self.K), dtype=np.float32)
        handle = getattr(comm, "handle", None) if world > 1 else None
        if world == 1 or handle is not None:
            # all walks in one call: the float32 carries pass from shard to shard on the stream (`world` broadcasts)
            check(lib.wgs_score_chains_walk_all(self._h, handle, f32p(parts)))
            self.ms["walk"] = last_assign_ms(self.b.ctx)
            return parts
        carry = None
        for r in range(world):
            mine = np.zeros((self.n * P, self.K), dtype=np.float64)
            if r == rank:
                check(lib.wgs_score_chains_walk(self._h, f32p(carry) if carry is not None else None, f32p(parts)))
                self.ms["walk"] = last_assign_ms(self.b.ctx)
                mine = parts.astype(np.float64)
            mine = comm.allreduce_sum(mine)      # only rank r contributes: a broadcast of its float32 values
            carry = np.ascontiguousarray(mine.astype(np.float32))
        return carry

    def chunk_sums(self):
        """Test hook: the last sums() per chunk of 8192 sites, (chunks, n, K) float64 (wgs_debug_score_chunks)."""
        nc = ctypes.c_int64()
        check(_lib.load().wgs_debug_score_chunks(self._h, None, ctypes.byref(nc)))
        out = np.empty((nc.value, self.n, self.K), dtype=np.float64)
        check(_lib.load().wgs_debug_score_chunks(self._h, f64p(out), ctypes.byref(nc)))
        return out

    def serial_blocks(self):
        """(blocks redone with the literal serial loop, (chain, block) pairs walked) of the last walk."""
        tot = ctypes.c_int64()
        k = _lib.load().wgs_score_last_serial_blocks(self._h, ctypes.byref(tot))
        return k, tot.value

    def close(self):
        if self._h:
            _lib.load().wgs_score_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def partition_sums_exact(beagle, afset, colptr=None, P=1, comm=None, literal=False):
    """utils.partition_loglikes for all n x K pairs, bit-exact (serial float32 per partition in site
    order).  With SNP shards the float32 carries travel from rank to rank in SNP order.
    Returns (n*P, K) float32.  literal=True (or P > 64) uses one GPU lane per chain -- the slow
    cross-check of the block-parallel chains."""
    n, K = beagle.n, afset.K
    if not literal and P <= MAX_BLOCK_PARALLEL_PARTS:
        sc = Score(beagle, afset, colptr)
        try:
            return sc.parts_exact(P, comm)
        finally:
            sc.close()
    _keep, cp = _colptr_arg(colptr, n, K)
    lib = _lib.load()
    world = comm.world if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    carry = None
    parts = np.zeros((n * P, K), dtype=np.float32)
    for r in range(world):
        mine = np.zeros((n * P, K), dtype=np.float64)
        if r == rank:
            check(lib.wgs_debug_parts_exact_literal(beagle.handle, afset.handle, cp, int(P),
                                                    f32p(carry) if carry is not None else None, f32p(parts)))
            mine = parts.astype(np.float64)
        if world > 1:
            mine = comm.allreduce_sum(mine)      # only rank r contributes: a broadcast of its float32 values
        carry = np.ascontiguousarray(mine.astype(np.float32))
    return carry if world > 1 else parts
