/*
 * wgsassign_hip_debug.h -- test hooks of libwgsassign_hip.so: entry points that exist for the test suite and the
 * micro-benchmarks (cross-check kernels, exhaustive checks of the arithmetic building blocks, host-only drivers of the
 * reader's hand-overs).  NOT part of the drop-in boundary (include/wgsassign_hip.h): nothing on the product path calls them,
 * and they may change without a bump of WGS_ABI_VERSION.
 */
#ifndef WGSASSIGN_HIP_DEBUG_H
#define WGSASSIGN_HIP_DEBUG_H

#include "wgsassign_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Named process-wide switches of the test suite (the product path reads no environment variable for them; 0 = off):
 *   codes_alloc_delay_ms              the helper thread's hipMalloc of the class codes' memory takes this much longer
 *   codes_alloc_release_after_sweeps  ... and is handed over only once the matrix has been swept directly this many times (or after
 *                                     2 s): which sweep of a fit finds the codes is then decided by a count, not by a race
 *   em_fuse_without_agreement         wgs_em_fit runs two iterations per sweep whenever THIS rank can, without the ranks' agreement
 *                                     (replays the defect of commit 807a461: the collectives' tags must catch it)
 *   em_coded_extra_lds                bytes of LDS em_coded_kernel requests beyond its table (occupancy experiment) */
int wgs_debug_hook(const char *name, int64_t value);

/* The literal one-lane-per-chain kernel behind wgs_assign_parts_exact, whatever P. */
int wgs_debug_parts_exact_literal(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t P,
                                  const float *carry_in, float *parts_out);

/* BGZF inflate on the device (csrc/inflate.hip; RFC 1951, one lane per block): `nblocks` raw deflate streams -- BGZF members
 * without header and trailer -- lying in the host buffer `comp` (in_off, in_len) are inflated into `out` (out_off, isize);
 * status[i] != 0 marks a stream the device did not accept (the host inflates those).  *kernel_ms: the kernel alone. */
int wgs_debug_inflate(wgs_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, const uint64_t *in_off, const uint32_t *in_len,
                      const uint64_t *out_off, const uint32_t *isize, int32_t nblocks, uint8_t *out, int64_t out_bytes,
                      uint8_t *status, float *kernel_ms);

/* Test hook, needs no GPU: drains the reader through the text hand-over (producer thread, carried partial lines,
 * parallel newline scan, row limit) with ordinary memory and the host parser in place of the device tokeniser. */
int wgs_debug_reader_text_rows(wgs_reader *r, int64_t chunk_bytes, int64_t limit_rows, float *rows, int64_t max_rows, int64_t *nrows);
int64_t wgs_debug_reader_text_chunks(wgs_reader *r);   /* chunks that hand-over produced */
/* Test hook, needs no GPU: the COMPRESSED hand-over of a BGZF file (what the device-resident ingest consumes: whole members
 * in caller-allocated staging + the text the header calls had inflated already), inflated on the host into text[0 .. cap).
 * info[0..3] = chunks, members, largest text of one chunk, chunks that carried pre-inflated text. */
int wgs_debug_reader_comp_text(wgs_reader *r, int64_t comp_bytes, int64_t text_cap, int nbuf, char *text, int64_t cap, int64_t *bytes,
                               int64_t *info);

/* Test hooks for the convergence chain: wgs_rmse1d's value through the literal one-lane serial
 * kernel (serial != 0) or through the block-parallel exact emulation, reporting the number of
 * 4096-element blocks that fell back to the serial loop; the same count for the last
 * wgs_em_rmse_chain of an EM batch. */
int wgs_debug_rmse1d(wgs_ctx *ctx, const float *v1, const float *v2, int64_t m, double *out, int serial,
                     int *serial_blocks);

/* Test hook for the EM kernel's correctly rounded divide (csrc/em_kernels.hip: div_exact): number of
 * 2^20 x per_thread pseudo-random EM-shaped operand pairs whose quotient differs bitwise from the
 * compiler's IEEE double divide. */
int wgs_debug_div_mismatch(wgs_ctx *ctx, uint64_t seed, uint64_t per_thread, uint64_t *mismatch);
/* ... and the accuracy of its once-refined reciprocal: the largest relative error over ALL 2^23 float32 mantissas of
 * the denominator at binary exponent `exponent` (its exactness argument needs < 2^-48; see em_kernels.hip). */
int wgs_debug_rcp_error(wgs_ctx *ctx, int exponent, double *max_rel);

/* Test hooks for the assignment kernel's double-precision log of float32 arguments
 * (csrc/assign_kernels.hip: log_f32arg): number of float32 bit patterns in [b0, b1) whose
 * float32-rounded log differs from the device math library's, and the values themselves. */
int wgs_debug_log_mismatch(wgs_ctx *ctx, uint32_t b0, uint32_t b1, uint64_t *count, uint32_t *first);
int wgs_debug_log_values(wgs_ctx *ctx, const float *x, float *out, int64_t n, int use_libm);

/* The sums of the last wgs_score_sums per chunk of 8192 sites (the addends of np.sum's running float64 total): out (NULL: only
 * *nchunks is set) receives ceil(blocks / 2) x n*K float64, out[c * n*K + i * K + k]. */
int wgs_debug_score_chunks(wgs_score *sc, double *out, int64_t *nchunks);

/* The two device kernels around a tagged RCCL collective (csrc/rccl_comm.hip: the row every rank writes behind the payload, the
 * comparison of all ranks' rows with one's own behind the all-reduce) on rows made up by the caller: what the ranks of a world-rank
 * job would run -- RCCL refuses two ranks on one GPU, so no test reaches them otherwise.  rows: world x 8 float64 {sequence number,
 * step, generation, iteration, shape, shape, payload elements, free word}; as_rank: whose check runs.  fault_out (18 float64):
 * [0] = 1 when the check reported a difference, [1] the other rank, [2..9] this rank's row, [10..17] the other's. */
int wgs_debug_comm_tag_kernels(wgs_ctx *ctx, int32_t world, const double *rows, int32_t as_rank, double *fault_out);

/* Cross-check only: FLOAT64 partition sums parts[(i*P + p)*K + k] (labels = global site index % P) and totals from the
 * round-1 kernel (lanes <-> pairs of individuals, tile ranges combined with float64 atomics): ~1e-5 from the
 * reference's serial float32 partition sums, not reproducible run to run.  Not on the product path. */
int wgs_debug_assign_parts_f64(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t P, int mode,
                               double *out, double *parts);

#ifdef __cplusplus
}
#endif
#endif /* WGSASSIGN_HIP_DEBUG_H */
