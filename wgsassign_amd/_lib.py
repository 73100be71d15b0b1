"""ctypes binding of libwgsassign_hip.so (include/wgsassign_hip.h).

There is NO CPU fallback: if the HIP library is missing or no AMD GPU is visible, every entry
point raises.  (The CPU restatement under oracle/ is test infrastructure and is never imported
from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WGSASSIGN_LIB_PATH") or os.path.join(_HERE, "libwgsassign_hip.so")   # override: experimental builds

MODE_EXACT = 0
MODE_FAST = 1

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_vp = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_i32 = ctypes.c_int32
c_int = ctypes.c_int

# name -> (restype, argtypes); every symbol include/wgsassign_hip.h declares
# int fn(double *buf, int64 n, void *user): in-place sum over ranks (wgs_comm_create_host)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int64, ctypes.c_void_p)

SIGNATURES = {
    "wgs_last_error": (ctypes.c_char_p, []),
    "wgs_version": (c_int, []),
    "wgs_build_id": (ctypes.c_char_p, []),
    "wgs_kernels_id": (ctypes.c_char_p, []),
    "wgs_ingest_kernels_id": (ctypes.c_char_p, []),
    "wgs_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "wgs_ctx_create": (c_int, [c_int, ctypes.POINTER(c_vp)]),
    "wgs_ctx_destroy": (None, [c_vp]),
    "wgs_ctx_sync": (c_int, [c_vp]),
    "wgs_ctx_stream": (c_vp, [c_vp]),
    "wgs_ctx_mem_info": (c_int, [c_vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "wgs_ctx_info": (c_int, [c_vp, ctypes.c_char_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_i64)]),
    "wgs_emmaf_update": (c_int, [c_vp, c_f32p, c_i64, c_i64, c_f32p, c_int]),
    "wgs_rmse1d": (c_int, [c_vp, c_f32p, c_f32p, c_i64, c_f64p]),
    "wgs_loglike": (c_int, [c_vp, c_f32p, c_i64, c_i64, c_f32p, c_i64, c_f32p, c_i64, c_i64, c_int]),
    "wgs_beagle_create": (c_int, [c_vp, c_i64, c_i64, c_i32p, c_i32, c_i64, ctypes.POINTER(c_vp)]),
    "wgs_beagle_destroy": (None, [c_vp]),
    "wgs_beagle_upload_rows": (c_int, [c_vp, c_f32p, c_i64, c_i64]),
    "wgs_beagle_download_rows": (c_int, [c_vp, c_f32p, c_i64, c_i64]),
    "wgs_beagle_synth": (c_int, [c_vp, ctypes.c_uint64, ctypes.c_double]),
    "wgs_beagle_bytes": (c_i64, [c_vp]),
    "wgs_beagle_set_rows": (c_int, [c_vp, c_i64]),
    "wgs_beagle_synth_quality": (c_int, [c_vp, ctypes.c_uint64, ctypes.c_double, c_i32, c_f64p, c_f64p]),
    "wgs_beagle_codes_info": (c_int, [c_vp, c_f64p]),
    "wgs_beagle_codes_prepare": (c_int, [c_vp, c_int]),
    "wgs_codes_model": (c_int, [c_vp, c_i32, c_f64p]),
    "wgs_beagle_codes_state": (c_int, [c_vp]),
    "wgs_beagle_codes_wait": (c_int, [c_vp, c_f64p]),
    "wgs_malloc_seconds": (ctypes.c_double, []),
    "wgs_em_create": (c_int, [c_vp, c_i32, c_i32p, c_i32p, c_int, ctypes.POINTER(c_vp)]),
    "wgs_em_destroy": (None, [c_vp]),
    "wgs_em_step": (c_int, [c_vp, c_f64p]),
    "wgs_em_step_dev": (c_int, [c_vp, c_vp]),
    "wgs_em_rmse_chain": (c_int, [c_vp, c_i32, ctypes.c_float, c_f32p]),
    "wgs_em_fit": (c_int, [c_vp, c_i32, ctypes.c_double, c_i64, c_vp, ctypes.c_double, c_i32p]),
    "wgs_em_fit_stats": (c_int, [c_vp, c_i32p, c_i32p, c_f64p, c_f64p]),
    "wgs_comm_rank": (c_int, [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "wgs_em_last_sweep_ms": (c_int, [c_vp, c_f32p]),
    "wgs_em_set_active": (c_int, [c_vp, c_i32, c_int]),
    "wgs_em_n_active": (c_int, [c_vp]),
    "wgs_em_clamp": (c_int, [c_vp, c_i32, ctypes.c_float, ctypes.c_float]),
    "wgs_em_get_f": (c_int, [c_vp, c_i32, c_f32p]),
    "wgs_em_set_f": (c_int, [c_vp, c_i32, c_f32p]),
    "wgs_em_get_f_range": (c_int, [c_vp, c_i32, c_int, c_i64, c_i64, c_f32p]),
    "wgs_em_f_dev": (c_vp, [c_vp, c_i32]),
    "wgs_afset_create": (c_int, [c_vp, c_i64, c_i32, ctypes.POINTER(c_vp)]),
    "wgs_afset_destroy": (None, [c_vp]),
    "wgs_afset_upload": (c_int, [c_vp, c_f32p]),
    "wgs_afset_download": (c_int, [c_vp, c_f32p]),
    "wgs_afset_set_column_from_em": (c_int, [c_vp, c_i32, c_vp, c_i32]),
    "wgs_afset_col_dev": (c_vp, [c_vp, c_i32]),
    "wgs_assign_parts_exact": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_i32, c_f32p, c_f32p]),
    "wgs_comm_unique_id": (c_int, [ctypes.POINTER(ctypes.c_uint8)]),
    "wgs_comm_init": (c_int, [c_vp, ctypes.POINTER(ctypes.c_uint8), c_int, c_int, ctypes.POINTER(c_vp)]),
    "wgs_comm_create_host": (c_int, [c_vp, c_int, c_int, ALLREDUCE_FN, c_vp, ctypes.POINTER(c_vp)]),
    "wgs_comm_destroy": (None, [c_vp]),
    "wgs_comm_allreduce_f64_dev": (c_int, [c_vp, c_vp, c_i64]),
    "wgs_comm_allreduce_f64": (c_int, [c_vp, c_f64p, c_i64]),
    "wgs_comm_allreduce_host_tagged": (c_int, [c_vp, c_f64p, c_i64, c_vp, c_f64p]),
    "wgs_comm_next_generation": (c_i32, [c_vp]),
    "wgs_comm_check": (c_int, [c_vp]),
    "wgs_comm_bcast_dev": (c_int, [c_vp, c_vp, c_i64, c_int]),
    "wgs_comm_info": (c_int, [c_vp, ctypes.POINTER(c_i64)]),
    "wgs_comm_time_collectives": (c_int, [c_vp, c_i32, c_i64, c_f64p]),
    "wgs_comm_stats": (c_int, [c_vp, ctypes.POINTER(c_i64)]),
    "wgs_comm_buffer": (c_vp, [c_vp, c_i64]),
    "wgs_comm_allreduce_buffer": (c_int, [c_vp, c_i64, c_f64p]),
    "wgs_fisher_obs": (c_int, [c_vp, c_vp, c_f32p, c_f32p]),
    "wgs_fisher_ind_sites": (c_int, [c_vp, c_vp, c_i32, c_i32, c_f32p]),
    "wgs_fisher_ind_means": (c_int, [c_vp, c_vp, c_i32, c_i32, c_f32p]),
    "wgs_fisher_ind_sums": (c_int, [c_vp, c_vp, c_i32, c_i32, c_f32p, c_f32p]),
    "wgs_reader_open": (c_int, [ctypes.c_char_p, c_int, ctypes.POINTER(c_vp)]),
    "wgs_reader_close": (None, [c_vp]),
    "wgs_reader_n_individuals": (c_int, [c_vp]),
    "wgs_reader_sample_name": (ctypes.c_char_p, [c_vp, c_int]),
    "wgs_reader_next": (c_int, [c_vp, c_f32p, c_i64, ctypes.POINTER(c_i64)]),
    "wgs_reader_skip": (c_int, [c_vp, c_i64, ctypes.POINTER(c_i64)]),
    "wgs_reader_skip_names": (c_int, [c_vp, c_i64, ctypes.POINTER(c_i64)]),
    "wgs_reader_chunk_sites": (c_vp, [c_vp, ctypes.POINTER(c_i64)]),
    "wgs_reader_count_sites": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_i64)]),
    "wgs_reader_estimate_sites": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_i64)]),
    "wgs_reader_build_index": (c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, c_i64, c_i32, ctypes.POINTER(c_i64)]),
    "wgs_reader_index_sites": (c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(c_i64)]),
    "wgs_reader_index_part": (c_int, [ctypes.c_char_p, ctypes.c_char_p, c_int, c_int, c_int]),
    "wgs_reader_index_merge": (c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, c_int, c_i64, c_i32, ctypes.POINTER(c_i64)]),
    "wgs_reader_open_indexed": (c_int, [ctypes.c_char_p, ctypes.c_char_p, c_i64, c_int, ctypes.POINTER(c_vp)]),
    "wgs_ingest_create": (c_int, [c_vp, c_vp, c_i64, c_i64, ctypes.POINTER(c_vp)]),
    "wgs_ingest_destroy": (None, [c_vp]),
    "wgs_ingest_next": (c_int, [c_vp, c_i64, c_vp, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "wgs_ingest_chunk_sites": (c_vp, [c_vp, ctypes.POINTER(c_i64)]),
    "wgs_ingest_stats": (c_int, [c_vp, c_f64p]),
    "wgs_debug_inflate": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp, c_f32p]),
    "wgs_debug_reader_text_rows": (c_int, [c_vp, c_i64, c_i64, c_f32p, c_i64, ctypes.POINTER(c_i64)]),
    "wgs_debug_reader_text_chunks": (c_i64, [c_vp]),
    "wgs_debug_reader_comp_text": (c_int, [c_vp, c_i64, c_i64, c_int, c_vp, c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "wgs_debug_rmse1d": (c_int, [c_vp, c_f32p, c_f32p, c_i64, c_f64p, c_int, ctypes.POINTER(c_int)]),
    "wgs_em_last_chain_serial_blocks": (c_int, [c_vp]),
    "wgs_debug_rcp_error": (c_int, [c_vp, c_int, c_f64p]),
    "wgs_debug_div_mismatch": (c_int, [c_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    "wgs_debug_log_mismatch": (c_int, [c_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]),
    "wgs_debug_log_values": (c_int, [c_vp, c_f32p, c_f32p, c_i64, c_int]),
    "wgs_assign_last_ms": (c_int, [c_vp, c_f32p]),
    "wgs_score_create": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_i32, c_i32, ctypes.POINTER(c_vp)]),
    "wgs_score_destroy": (None, [c_vp]),
    "wgs_score_sums": (c_int, [c_vp, c_int, c_f64p]),
    "wgs_score_total_from": (c_int, [c_vp, c_f64p, c_f64p]),
    "wgs_score_totals_all": (c_int, [c_vp, c_vp, c_f64p, c_f64p]),
    "wgs_score_chains_walk_all": (c_int, [c_vp, c_vp, c_f32p]),
    "wgs_score_chains_prepare": (c_int, [c_vp, c_i32, c_f64p]),
    "wgs_score_chains_walk": (c_int, [c_vp, c_f32p, c_f32p]),
    "wgs_loo": (c_int, [c_vp, c_vp, c_vp, c_i32, ctypes.c_double, c_i64, c_vp, c_i32, c_i32, c_int, c_int, c_f64p, c_f32p, c_i32p]),
    "wgs_loo_stats": (c_int, [c_f64p]),
    "wgs_score_last_serial_blocks": (c_int, [c_vp, ctypes.POINTER(c_i64)]),
    "wgs_debug_hook": (c_int, [ctypes.c_char_p, c_i64]),
    "wgs_debug_score_chunks": (c_int, [c_vp, c_f64p, ctypes.POINTER(c_i64)]),
    "wgs_debug_comm_tag_kernels": (c_int, [c_vp, c_i32, c_f64p, c_i32, c_f64p]),
    "wgs_debug_parts_exact_literal": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_i32, c_f32p, c_f32p]),
    "wgs_assign": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_int, c_f64p]),
    "wgs_debug_assign_parts_f64": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp), c_i32, c_int, c_f64p, c_f64p]),
}

_lib = None
ABI_VERSION = 3      # = WGS_ABI_VERSION of include/wgsassign_hip.h


def load():
    """Load the shared library and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "wgsassign_amd: %s is missing. Build it with `python -m wgsassign_amd.build` "
                "(needs hipcc). There is no CPU fallback." % LIB_PATH)
        # RCCL between processes needs dmabuf IPC on this host driver; the runtime reads this at its first HIP call
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        if lib.wgs_version() != ABI_VERSION:      # signatures above are those of include/wgsassign_hip.h's WGS_ABI_VERSION
            raise RuntimeError("wgsassign_amd: %s has ABI version %d, this shim binds version %d -- rebuild the library "
                               "(python -m wgsassign_amd.build)" % (LIB_PATH, lib.wgs_version(), ABI_VERSION))
        _lib = lib
    return _lib


def last_error():
    msg = load().wgs_last_error()
    return msg.decode() if msg else ""


class CollTag(ctypes.Structure):
    """wgs_coll_tag of include/wgsassign_hip.h: what a collective is, said by its caller."""
    _fields_ = [("op", c_i32), ("generation", c_i32), ("iteration", c_i32), ("shape_a", c_i32), ("shape_b", c_i32), ("aux", c_i32)]


callback_error = None      # an exception raised inside a Python callback the library called (comm.SocketComm's all-reduce): check() re-raises it


def check(rc):
    global callback_error
    if rc != 0:
        msg = last_error()
        if callback_error is not None:
            e, callback_error = callback_error, None
            raise e
        if rc == 2:
            raise ValueError(msg)
        if msg.startswith("collective mismatch"):          # the ranks are out of step (csrc/rccl_comm.hip): its own exception, never retried
            from .comm import CollectiveMismatch, decode_sites
            raise CollectiveMismatch(decode_sites(msg))
        # HIP_TRY reports "<file>.hip:<line>: <call> failed: <hip error>"; other failures (the reader's) carry their own text
        raise RuntimeError(("wgsassign_amd HIP call failed: " if ".hip:" in msg else "wgsassign_amd: ") + msg)


def f32p(a):
    return a.ctypes.data_as(c_f32p)


def f64p(a):
    return a.ctypes.data_as(c_f64p)


def i32p(a):
    return a.ctypes.data_as(c_i32p)
