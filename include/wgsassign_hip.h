/*
 * wgsassign_hip.h -- C ABI of libwgsassign_hip.so, the MI355X (gfx950) implementation of
 * WGSassign's hot path: the per-population EM allele-frequency estimator and the per-SNP
 * assignment log-likelihood summation.
 *
 * The reference exposes this path as Python-callable Cython functions (no C plugin ABI), so
 * the boundary is: Python shim (the modules of wgsassign_amd/, same module/function names as the
 * reference) -> ctypes -> the functions below.  Each entry point names the reference
 * interface it stands in for (paths relative to the reference repository).
 *
 * Conventions
 *   - plain pointers and sizes only; all matrices are float32, C-contiguous, exactly as the
 *     reference's typed memoryviews require (emMAF_cy.pyx:10, glassy_cy.pyx:12);
 *   - every function returns 0 on success, non-zero on failure; wgs_last_error() then holds a
 *     message for the calling thread (the shim raises RuntimeError/ValueError from it);
 *   - "host" pointers are ordinary process memory; "dev" pointers are HIP device memory of the
 *     context's device (so a caller that owns device buffers -- e.g. an RCCL bounce buffer
 *     allocated elsewhere -- can hand them in directly);
 *   - arithmetic mode: WGS_MODE_EXACT reproduces the reference's rounding sequence operation
 *     by operation (double products rounded to float32, float32 (p0+p1)+p2, serial float32
 *     accumulation over individuals in file order) and is bit-identical to the reference for
 *     allele frequencies and EM iteration counts; WGS_MODE_FAST evaluates each term in float32: the
 *     n x K sums stay within 1.2e-7 relative at full size, the EM frequencies drift up to 7e-6 at 10M SNPs
 *     (iteration counts unchanged) -- see DESIGN.md; the host side only ever defaults to EXACT.
 */
#ifndef WGSASSIGN_HIP_H
#define WGSASSIGN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WGS_MODE_EXACT 0
#define WGS_MODE_FAST 1

typedef struct wgs_ctx wgs_ctx;       /* one HIP device + stream + workspace */
typedef struct wgs_beagle wgs_beagle; /* device-resident genotype-likelihood matrix (SNP shard) */
typedef struct wgs_em wgs_em;         /* a batch of EM fits over one wgs_beagle */

/* ------------------------------------------------------------------ context */
const char *wgs_last_error(void);
/* ABI version of this header (WGS_ABI_VERSION).  2 (round 4): wgs_assign lost its `P` / `parts` arguments and
 * wgs_fisher_obs_ind was removed in round 3 while the number stayed 1 -- a caller built against the older header must
 * compare wgs_version() with the WGS_ABI_VERSION it was compiled with and refuse to run on a mismatch (the ctypes shim does:
 * wgsassign_amd/_lib.py); wgs_beagle_codes_info fills 20 entries; wgs_comm_info is new.  3 (round 5): self-checking
 * collectives (wgs_coll_tag, wgs_comm_check, wgs_comm_next_generation, wgs_comm_allreduce_host_tagged). */
#define WGS_ABI_VERSION 3
int wgs_version(void);
/* sha256[:16] over every source of the library / over the sources of the EM and scoring kernels (em_kernels.hip,
 * assign_kernels.hip, beagle_kernels.hip, codes_kernels.hip, common.h, log_table.h), fixed at build time: profiles record them, bench.py quotes hardware
 * counters only from a profile whose kernels id equals the loaded library's. */
const char *wgs_build_id(void);
const char *wgs_kernels_id(void);
/* ... and over the sources of the ingest kernels (ingest.hip, inflate.hip, common.h): recorded with the ingest profiles. */
const char *wgs_ingest_kernels_id(void);
int wgs_device_count(int *count);
int wgs_ctx_create(int device, wgs_ctx **out);
void wgs_ctx_destroy(wgs_ctx *ctx);
int wgs_ctx_sync(wgs_ctx *ctx);
/* The context's hipStream_t (all work of the library is enqueued on it). */
void *wgs_ctx_stream(wgs_ctx *ctx);
/* Free / total device memory right now (hipMemGetInfo): used to size leave-one-out batches. */
int wgs_ctx_mem_info(wgs_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);
/* Device name / CU count / total memory of the context's device (for reports). */
int wgs_ctx_info(wgs_ctx *ctx, char *name, int name_len, int *cus, int64_t *mem_bytes);

/* ------------------------------------------------------------------ (1) thin mirrors
 * Same arguments, in-place semantics and results as the Cython kernels; host pointers.
 * They upload, run the HIP kernel and download -- meant for drop-in use and parity tests on
 * inputs the reference handles in seconds, not for the at-scale path (use (2)). */

/* emMAF_cy.emMAF_update(L, f, t)            -- emMAF_cy.pyx:10-23.  f is updated in place. */
int wgs_emmaf_update(wgs_ctx *ctx, const float *L, int64_t m, int64_t n, float *f, int mode);
/* emMAF_cy.rmse1d(v1, v2) -> float          -- emMAF_cy.pyx:26-33.  Bit-exact serial float32
 * accumulation (see wgs_em_rmse_chain). */
int wgs_rmse1d(wgs_ctx *ctx, const float *v1, const float *v2, int64_t m, double *out);
/* glassy_cy.loglike(L, A, loglike_vec, t, i, k) -- glassy_cy.pyx:12-21.  vec is accumulated
 * into, in place.  A is (m, K). */
int wgs_loglike(wgs_ctx *ctx, const float *L, int64_t m, int64_t n, const float *A, int64_t K,
                float *vec, int64_t i, int64_t k, int mode);

/* ------------------------------------------------------------------ (2) device-resident path */

/* Genotype-likelihood matrix of m SNPs x n individuals (reader_cy.pyx:71-77 layout on the
 * host: row s = g0_0 g1_0 g0_1 g1_1 ...).  On the device it is kept as one SNP-major slab of
 * (g0,g1) float2 pairs per group of individuals ("population slab"): slab g is m x ld_g, its
 * columns are the individuals with group_of[i] == g in file order -- the on-device equivalent
 * of the per-population column gather at WGSassign.py:227-233 / glassy.py:69-77, done once.
 * group_of == NULL puts every individual in one group (the --get_pop_like case).
 * site0 is the global index of the shard's first SNP (partition labels use global indices). */
int wgs_beagle_create(wgs_ctx *ctx, int64_t m, int64_t n, const int32_t *group_of, int32_t n_groups,
                      int64_t site0, wgs_beagle **out);
void wgs_beagle_destroy(wgs_beagle *b);
/* Copy host rows [row0, row0+nrows) of an (m, 2n) float32 matrix into the slabs. */
int wgs_beagle_upload_rows(wgs_beagle *b, const float *L_rows, int64_t row0, int64_t nrows);
/* Copy rows back into (nrows, 2n) host layout (tests, CPU-baseline sample). */
int wgs_beagle_download_rows(wgs_beagle *b, float *L_rows, int64_t row0, int64_t nrows);
/* Fill the slabs with synthetic low-depth genotype likelihoods on the device (SURVEY 8d:
 * Philox-4x32-10 counter RNG keyed by (seed, global SNP, individual), HWE genotypes from
 * per-group frequencies, Poisson(depth) reads, error 0.01, GLs rounded to 6 decimals). */
int wgs_beagle_synth(wgs_beagle *b, uint64_t seed, double depth);
/* The same with QUALITY-DEPENDENT likelihoods, as ANGSD's -GL 2 writes them for real reads: every read draws its base quality
 * from n_bins bins (Phred values quals[], probabilities probs[]; n_bins <= 8) and enters with its own error rate -- 80-100
 * distinct (g0, g1) per SNP among 1000 individuals with the four bins of current instruments instead of the 27 of the fixed
 * error of wgs_beagle_synth (bench.py: extra.realistic_gl; the NumPy twin of the model is tests/synth.py: make_beagle_quality). */
int wgs_beagle_synth_quality(wgs_beagle *b, uint64_t seed, double depth, int32_t n_bins, const double *quals, const double *probs);
int64_t wgs_beagle_bytes(const wgs_beagle *b);
/* A matrix created with room for more sites than the file turned out to hold (wgs_reader_estimate_sites): rows becomes its
 * number of sites (0 < rows <= the rows it was created with; the rows behind were never written).  Slabs more than a tenth
 * too large are moved into allocations of the right size.  Only before anything was made FROM the matrix (EM batches,
 * scores: rc 2 otherwise); its class codes are dropped. */
int wgs_beagle_set_rows(wgs_beagle *b, int64_t rows);
/* Class codes of the matrix: low-depth genotype likelihoods take few distinct (g0, g1) values per SNP (29 on average in
 * the bundled 85-individual data, 27 among 1000 individuals of the 2x synthetic matrices, ~80 with binned base qualities), so
 * the kernels evaluate the EM term's quotient / the per-site log-likelihood once per CLASS and SNP and look it up per
 * individual -- same values, same order of accumulation, same bits.  Built by ONE pass over the matrix when a sweep that
 * profits asks for them (a scoring sweep with shared columns; an EM fit with enough iterations ahead): one byte per (SNP,
 * individual) + a dictionary, and the same again in every population slab's own numbering (csrc/codes.hip).
 * WGSASSIGN_CODES=0 disables them.  A SNP with more classes than the tables hold is left uncoded and taken from the
 * float32 slab by every sweep (nothing matrix-wide depends on it); a matrix whose typical SNP has that many is not coded.
 * info[0..19] = available, classes of the richest coded SNP, bytes held, build milliseconds (sample pass, allocation and
 * encode pass), mean classes per coded SNP, milliseconds of the encode pass alone, of the sample pass, bytes of the slabs' own
 * numbering, rows of the coded EM sweep's quotient table, share of (slab, tile) pairs with more classes than rows (swept
 * directly), hash slots per SNP of the encoder (64 / 128 / 256), share of SNPs left uncoded, dictionary rows per tile, hash
 * probe rounds beyond the first per 16 lookups, milliseconds of the allocation, rows of the coded scoring sweep's table, mean classes
 * per SNP and per (population slab, SNP) in the sample pass, SNPs per table of the coded scoring sweep, milliseconds the building call
 * waited for the codes' memory ([14] is what its hipMalloc took on the helper thread). */
int wgs_beagle_codes_info(wgs_beagle *b, double *info);
/* 1: the codes exist, 0: nothing has asked for them yet, -1: the matrix was found not worth coding (or no memory).  Builds nothing. */
int wgs_beagle_codes_state(wgs_beagle *b);
/* The codes' device memory is allocated on a helper thread (hipMalloc of VRAM an earlier process used takes seconds on this driver);
 * a sweep that wants the codes waits WGSASSIGN_CODES_ALLOC_WAIT_MS (3) for it and otherwise runs over the float32 slabs.  This call
 * waits for an allocation in flight, builds nothing; *alloc_ms = what that hipMalloc took (0: none was in flight).  While one is in
 * flight wgs_em_fit_stats and wgs_assign_last_ms report -1 for kernel times (hipEventElapsedTime would wait for it). */
int wgs_beagle_codes_wait(wgs_beagle *b, double *alloc_ms);
/* Seconds the calling threads of this process have spent inside hipMalloc through the library so far (a running total; the
 * helper thread's allocation of the class codes' memory is not in it).  The driver clears VRAM an earlier process used when
 * it hands it out again, so the same allocation costs 0.3 ms or seconds by what the box did before; callers that time whole
 * paths (bench.py) report the share. */
double wgs_malloc_seconds(void);
/* What the library's cost models predict for this matrix (csrc/em_api.hip: em_codes_model; csrc/codes.hip: wgs_codes_pay_for_scoring),
 * so that a caller can put the prediction beside its measurement.  out[0..11]: one EM sweep of all population slabs over the float32
 * matrix in ms | share of it a coded sweep saves | encode pass (with the slabs' own numbering) in ms | sweeps the decision counts |
 * 1 = the model builds the codes for a fit | one scoring sweep over K_score populations over the float32 matrix in ms | share of
 * that the coded sweep costs | encode pass for scoring alone in ms | 1 = the model builds them for scoring | 1 = from the matrix's
 * own sample pass | classes per (slab, SNP) and per SNP in the sample.  K_score = 0: the EM entries only. */
int wgs_codes_model(wgs_beagle *b, int32_t K_score, double *out);
/* Builds the codes now rather than at the first sweep that asks.  (`em` is ignored since version 2: one pass builds all.) */
int wgs_beagle_codes_prepare(wgs_beagle *b, int em);

/* A batch of EM fits (emMAF.py:15-27) over slabs of `b`.  Fit j estimates the frequency of
 * every SNP from the individuals of group fit_group[j], leaving out individual fit_skip[j]
 * (global individual index, must belong to that group) or nobody when fit_skip[j] < 0 --
 * the leave-one-out re-fit of glassy.py:65-78.  All fits start at f = 0.25 (emMAF.py:17-18). */
int wgs_em_create(wgs_beagle *b, int32_t n_fits, const int32_t *fit_group, const int32_t *fit_skip,
                  int mode, wgs_em **out);
void wgs_em_destroy(wgs_em *em);
/* One EM update (emMAF_cy.pyx:10-23) of every still-active fit in ONE sweep over the slabs,
 * fused with the float64 sum over this shard's SNPs of (f_new - f_old)^2 per fit.  After the
 * call each swept fit's current frequencies are the updated ones and its previous ones are
 * kept (for wgs_em_rmse_chain).  ssq_host (n_fits doubles, may be NULL) receives the sums;
 * fits that were not swept report 0.  Synchronises the context's stream. */
int wgs_em_step(wgs_em *em, double *ssq_host);
/* Same, but leaves the sums in device memory at ssq_dev (n_fits doubles, zeroed by the call)
 * and does NOT synchronise -- for callers that all-reduce them on the device with RCCL. */
int wgs_em_step_dev(wgs_em *em, double *ssq_dev);
/* The reference's convergence metric (emMAF_cy.pyx:26-33) is a SERIAL float32 accumulation
 * over all m SNPs.  This continues that exact chain over this shard's SNPs for one fit:
 * carry_in is the float32 running sum after the preceding shards (0 for the first shard),
 * carry_out the running sum after this shard; diff = sqrt((double)(carry / (float)m_total)). */
int wgs_em_rmse_chain(wgs_em *em, int32_t fit, float carry_in, float *carry_out);
/* Duration in milliseconds of the sweep kernel(s) of the last wgs_em_step*, measured with HIP
 * events recorded on the context's stream around the launch (waits for the kernel). */
int wgs_em_last_sweep_ms(wgs_em *em, float *ms);
/* Freeze (active = 0) or re-activate a fit: frozen fits are skipped by wgs_em_step and keep
 * the frequencies of their last update -- emMAF.py:23-25 `break`s after the update. */
int wgs_em_set_active(wgs_em *em, int32_t fit, int active);
int wgs_em_n_active(wgs_em *em);
/* emMAF.emMAF(L, iter, tole, t) -- emMAF.py:15-27 -- for EVERY fit of the batch in one call: update,
 * `rmse1d(f, f_prev) < tole` -> stop (after the update), at most max_iter updates.  iters_out[j] = the
 * 1-based iteration at which fit j converged, 0 if max_iter was exhausted (the reference prints nothing
 * then).  m_total = SNPs of ALL shards (the metric divides by it); comm = the RCCL communicator of the
 * SNP shards or NULL for one shard.  Iterations are enqueued one ahead of the host: a device kernel
 * decides the clear cases from the (all-reduced) float64 sums; only sums within the band
 * max(guard_floor, m_total * 2^-24) of tole^2 * m_total -- where the reference's serial float32 sum can
 * fall on either side -- go through the exact chain (wgs_em_rmse_chain's, batched over fits, carries
 * handed from shard to shard in rank order).  Identical iteration counts and frequencies to the
 * step-by-step protocol of wgsassign_amd/device.py:run_em (kept for communicators other than RCCL). */
typedef struct wgs_comm wgs_comm;
int wgs_em_fit(wgs_em *em, int32_t max_iter, double tole, int64_t m_total, wgs_comm *comm, double guard_floor,
               int32_t *iters_out);
/* Iterations enqueued / batched exact-chain resolutions / wall seconds of the last wgs_em_fit, and the summed
 * duration in ms of its sweep kernels (HIP events recorded on the context's stream around every sweep; read by this
 * call, not inside the fit).  *sweep_ms = -1 while the class codes' memory is being allocated (wgs_beagle_codes_wait). */
int wgs_em_fit_stats(wgs_em *em, int32_t *iterations, int32_t *chain_batches, double *seconds, double *sweep_ms);
/* Clamp fit j's frequencies to [lo, hi] the way WGSassign.py:236-240 does (float32 compares,
 * NaN untouched). */
int wgs_em_clamp(wgs_em *em, int32_t fit, float lo, float hi);
/* Copy fit j's current frequencies (m floats) to the host / get their device address. */
int wgs_em_get_f(wgs_em *em, int32_t fit, float *f_host);
int wgs_em_set_f(wgs_em *em, int32_t fit, const float *f_host);
/* Rows [row0, row0+nrows) of fit j's current (previous == 0) or previous-iteration (previous != 0)
 * frequencies: the two vectors emMAF.py:22 hands to rmse1d. */
int wgs_em_get_f_range(wgs_em *em, int32_t fit, int previous, int64_t row0, int64_t nrows, float *f_host);
const float *wgs_em_f_dev(wgs_em *em, int32_t fit);

/* Frequency vectors kept on the device for the assignment kernels: K vectors of m floats
 * (population-major), uploaded from the reference's (m, K) matrix (`.pop_af.npy`,
 * WGSassign.py:243,303). */
typedef struct wgs_afset wgs_afset;
int wgs_afset_create(wgs_ctx *ctx, int64_t m, int32_t K, wgs_afset **out);
void wgs_afset_destroy(wgs_afset *a);
int wgs_afset_upload(wgs_afset *a, const float *A_mK);           /* host (m, K) -> K device vectors */
int wgs_afset_download(wgs_afset *a, float *A_mK);               /* K device vectors -> host (m, K) */
int wgs_afset_set_column_from_em(wgs_afset *a, int32_t col, wgs_em *em, int32_t fit); /* device copy */
const float *wgs_afset_col_dev(wgs_afset *a, int32_t col);

/* glassy.assignLL(L, af, t) -- glassy.py:18-44 -- for ALL n x K pairs in one sweep over the
 * slabs: out[i*K + k] = sum over this shard's SNPs of the float32 per-site log-likelihood
 * (glassy_cy.pyx:18-21), accumulated in float64 (the reference sums with np.sum(dtype=float),
 * glassy.py:38) in a fixed order (run-to-run reproducible).  colptr (may be NULL) overrides the frequency vector per (individual, k):
 * colptr[i*K + k] is a device pointer to m floats -- this is how the leave-one-out scoring of
 * glassy.py:87-105 (per-individual columns, sticky overwrite) is expressed.  out is a host float64
 * buffer [n*K], summed into (caller zeroes).  Partition sums (utils.py:129-151): wgs_assign_parts_exact. */
int wgs_assign(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int mode, double *out);

/* utils.partition_loglikes(per_site_ll, P) -- utils.py:129-151 -- for every (individual, population)
 * pair, BIT-EXACT: the reference accumulates each partition serially in float32 in site order
 * (np.add.at).  carry_in (float32 [n*P*K], NULL for the first SNP shard) is the running value after the
 * preceding shards, parts_out (float32 [n*P*K], index (i*P + p)*K + k) the value after this one.
 * Always exact-mode arithmetic.  One call = wgs_score_create + _sums + _chains_prepare + _chains_walk
 * below (block-parallel); more than ~64 partitions fall back to one literal chain per GPU lane. */
int wgs_assign_parts_exact(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t P,
                           const float *carry_in, float *parts_out);

/* The scoring step of glassy.py:31-42 / 92-109 as an object, for callers that need the pieces separately
 * (leave-one-out in batches, SNP shards whose chains are joined by float32 carries):
 *   wgs_score_create          b, a, colptr as for wgs_assign; only individuals [row_lo, row_hi) (file order)
 *                             are scored -- a leave-one-out batch scores its own individuals only;
 *   wgs_score_sums            the n x K float64 sums, ONE launch over all population slabs; every
 *                             (individual, population, block of 4096 SNPs) sum has one writer and the blocks
 *                             are added in order: reproducible bit for bit (host `out`, n*K, overwritten);
 *   wgs_score_chains_prepare  block functions of the exact float32 partition chains on the ulp grid predicted
 *                             from those sums; `start` (host n*K float64 or NULL) = the sums over the
 *                             preceding SNP shards.  All shards can prepare in parallel;
 *   wgs_score_chains_walk     walks this shard's chains from carry_in (host float32 [n*P*K] or NULL) -- the
 *                             only step that follows the previous shard -- into parts_out. */
typedef struct wgs_score wgs_score;
int wgs_score_create(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t row_lo, int32_t row_hi,
                     wgs_score **out);
void wgs_score_destroy(wgs_score *sc);
int wgs_score_sums(wgs_score *sc, int mode, double *out);
/* ... continued from the SNP shards before this one (NumPy's running total handed from shard to shard): see api.hip. */
int wgs_score_total_from(wgs_score *sc, const double *carry_in, double *out);
/* The same for ALL shards in one call: the running total goes from shard to shard in SNP order ON THE STREAM -- `world`
 * broadcasts of n*K float64, one readback.  totals_out: the totals (every rank); before_out (may be NULL): the total of
 * the shards before this one (wgs_score_chains_prepare's `start`).  comm NULL = one shard.  Needs wgs_score_sums first. */
int wgs_score_totals_all(wgs_score *sc, wgs_comm *comm, double *totals_out, double *before_out);
int wgs_score_chains_prepare(wgs_score *sc, int32_t P, const double *start);
int wgs_score_chains_walk(wgs_score *sc, const float *carry_in, float *parts_out);
/* The walks of ALL shards in one call (every rank has prepared its block functions): `world` broadcasts of n*P*K
 * float32 on the stream, one readback; parts_out receives the values after the last shard on every rank. */
int wgs_score_chains_walk_all(wgs_score *sc, wgs_comm *comm, float *parts_out);
/* glassy.loo(L, af, IDs, t, maf_iter, maf_tole, downsampled_L, num_partitions) -- glassy.py:47-112 -- in one
 * call on device-resident data: per individual (file order) the re-fit of its population without it
 * (wgs_em_fit, a batch of individuals at once), the clamp with n_pop - 1, the never-restored overwrite of
 * the population's column of `a` (in: full-population estimates, out: each population's last re-fit), the
 * K float64 sums and -- when parts_out is given -- the serial float32 partition sums.
 * scored: the matrix that is scored (NULL = b).  batch: re-fits per EM batch (0 = what fits the free
 * device memory, agreed across ranks).  em_mode / score_mode: arithmetic of the re-fits and of the sums
 * (the partition sums are always exact).  ll_out: host float64 [n*K]; parts_out: host float32 [n*P*K] or
 * NULL; iters_out: [n] convergence iterations of the re-fits (0 = max_iter exhausted). */
int wgs_loo(wgs_beagle *b, wgs_beagle *scored, wgs_afset *a, int32_t max_iter, double tole, int64_t m_total,
            wgs_comm *comm, int32_t P, int32_t batch, int em_mode, int score_mode, double *ll_out, float *parts_out,
            int32_t *iters_out);
/* Phases of the last wgs_loo of this process: stats[0..6] = seconds in the EM re-fits (wgs_em_fit incl. its exact
 * chains), in the scoring sweeps (+ cross-rank totals), in the exact partition chains; EM sweep kernel ms; EM batches;
 * batched chain resolutions of the re-fits; EM iterations enqueued. */
int wgs_loo_stats(double *stats);

/* (chain, block) pairs of the last walk that took the literal serial loop / walked in all (diagnostics). */
int wgs_score_last_serial_blocks(wgs_score *sc, int64_t *total_blocks);

/* ------------------------------------------------------------------ RCCL communicator (SNP shards)
 * The one collective of the sharded path -- a sum all-reduce of a few float64 over xGMI -- without
 * a tensor framework: librccl is dlopen'ed on first use.  Rank 0 creates the 128-byte unique id,
 * the host side distributes it (wgsassign_amd/comm.py: TCP on MASTER_ADDR), every rank inits. */
int wgs_comm_unique_id(uint8_t *id128);
int wgs_comm_init(wgs_ctx *ctx, const uint8_t *id128, int rank, int world, wgs_comm **out);
/* A communicator over the caller's own transport: fn sums n host doubles in place over all ranks (0 = success).  The
 * library's loops (wgs_em_fit, wgs_loo) stage their device buffers through pinned host memory for it. */
typedef int (*wgs_allreduce_fn)(double *buf, int64_t n, void *user);
int wgs_comm_create_host(wgs_ctx *ctx, int rank, int world, wgs_allreduce_fn fn, void *user, wgs_comm **out);
void wgs_comm_destroy(wgs_comm *c);
int wgs_comm_rank(wgs_comm *c, int *rank, int *world);
/* Self-checking collectives (ABI 3).  The ranks of the sharded path must issue the same collectives in the same order with the
 * same meaning; every collective of a wgs_comm therefore carries -- inside the SAME transfer as its payload -- a row per rank:
 * {sequence number on this communicator, op, generation, iteration, shape_a, shape_b, payload elements, aux}.  An all-reduce
 * all-gathers the rows (each rank fills its own, the sum fills in the rest) and every rank compares all of them with its own; a
 * broadcast carries the root's row and every receiver compares.  A rank that is out of step makes the NEXT wgs_comm_check -- every
 * entry point that takes a communicator calls it wherever it synchronises -- fail with rc 1 and a wgs_last_error() that starts
 * "collective mismatch:" and names both ranks' tuples; the communicator refuses all further collectives.  `aux` is not compared:
 * a free word per rank that all ranks get to see (wgs_em_fit: "this rank could run two iterations per sweep").  The reference has
 * no counterpart (its only parallelism is OpenMP inside one process, emMAF_cy.pyx:16). */
typedef struct wgs_coll_tag {
    int32_t op;                 /* WGS_OP_*: which exchange step of the path */
    int32_t generation;         /* which fit / scoring call / leave-one-out batch of this communicator (wgs_comm_next_generation) */
    int32_t iteration;          /* EM sweep number, rank hop, ... */
    int32_t shape_a, shape_b;   /* what the payload means: fits in the sweep and EM iterations they run, cells, root of a broadcast */
    int32_t aux;                /* not compared */
} wgs_coll_tag;
enum { WGS_OP_GENERIC = 0, WGS_OP_EM_SUMS = 1, WGS_OP_EM_CHAIN = 2, WGS_OP_EM_FIT_END = 3, WGS_OP_SCORE_TOTALS = 4,
       WGS_OP_PART_CHAINS = 5, WGS_OP_LOO_BATCH = 6, WGS_OP_TIMING = 7, WGS_OP_HOST = 8 };
int32_t wgs_comm_next_generation(wgs_comm *c);
int wgs_comm_check(wgs_comm *c);
/* Sum all-reduce of n host float64 with a caller-given tag; rows_out (NULL or world * 8 float64) receives every rank's row
 * (rows_out[r * 8 + 7] = rank r's aux).  Returns after the rows have been compared. */
int wgs_comm_allreduce_host_tagged(wgs_comm *c, double *host_buf, int64_t n, const wgs_coll_tag *tag, double *rows_out);
/* In-place sum of n float64 in device memory, enqueued on the context's stream (pairs with wgs_em_step_dev); tagged
 * WGS_OP_GENERIC (sequence number and size are still compared), staged through the communicator's bounce buffer. */
int wgs_comm_allreduce_f64_dev(wgs_comm *c, double *dev_buf, int64_t n);
/* Broadcast of `bytes` bytes (a multiple of 4) of DEVICE memory from rank `root`, enqueued on the context's stream: how
 * a running value -- np.sum's float64 total, a float32 chain carry -- passes from SNP shard to SNP shard without a host
 * round trip (ncclBroadcast; over a host-backed communicator: 32-bit words widened to float64 through its sum all-reduce). */
int wgs_comm_bcast_dev(wgs_comm *c, void *dev_buf, int64_t bytes, int root);
/* stats[0..3]: all-reduces, broadcasts, payload bytes, host round trips (stream synchronisations) of this
 * communicator's collectives so far. */
int wgs_comm_stats(wgs_comm *c, int64_t *stats);
/* info[0..7]: 1 = RCCL communicator / 0 = host-backed; the number of ranks, this rank and the device as RCCL ITSELF reports
 * them for the communicator it built (ncclCommCount / ncclCommUserRank / ncclCommCuDevice; -1 where unavailable) -- wgs_comm_init
 * fails when they differ from what was asked for --; world and rank as given at creation; [6] collectives issued so far, [7] 1 once
 * a rank was found out of step. */
int wgs_comm_info(wgs_comm *c, int64_t *info);
/* Mean device microseconds (HIP events on the context's stream) of `reps` sum all-reduces of n float64 and of `reps`
 * broadcasts of n float64 from rank 0: us_out[0..1].  Collective. */
int wgs_comm_time_collectives(wgs_comm *c, int32_t reps, int64_t n, double *us_out);
/* Same for a host buffer (staged through the device); returns when the result is back. */
int wgs_comm_allreduce_f64(wgs_comm *c, double *host_buf, int64_t n);
/* The communicator's device bounce buffer (>= n float64), e.g. as the target of wgs_em_step_dev;
 * wgs_comm_allreduce_buffer reduces its first n elements in place on the stream, copies them to
 * host_out and synchronises: sweep -> all-reduce -> one readback, nothing in between on the host. */
double *wgs_comm_buffer(wgs_comm *c, int64_t n);
int wgs_comm_allreduce_buffer(wgs_comm *c, int64_t n, double *host_out);

/* ------------------------------------------------------------------ Fisher information (--ne_obs)
 * fisher.fisher_obs(L, af, IDs, t) -- fisher.py:11-44 over fisher_cy.fisher_obs / ne_obs
 * (fisher_cy.pyx:12-39): per (SNP, population) the serial float32 sum over the population's
 * individuals of the observed-information term, and n_tilde = 0.5 * f * a * (1 - a).  The slabs of
 * `b` must be the populations of `a`'s columns.  Outputs are host (m, K) float32 matrices. */
int wgs_fisher_obs(wgs_beagle *b, wgs_afset *a, float *f_obs_mK, float *ne_obs_mK);
/* fisher.fisher_obs_ind -- fisher.py:46-60 over fisher_cy.pyx:41-65 -- is wgs_fisher_ind_means / _sums below.
 * The per-site float32 values themselves for individuals [i0, i0+count) of ONE population
 * (rows_out[(i - i0) * m + s]), so the host can apply np.mean to each row exactly as fisher.py:59. */
int wgs_fisher_ind_sites(wgs_beagle *b, wgs_afset *a, int32_t i0, int32_t count, float *rows_out);
/* ... and np.mean of each of those rows formed on the device exactly as NumPy forms it (pairwise float32 sum, float64
 * division, float32 result): means_out[i - i0], fisher.py:59 without moving the rows to the host. */
int wgs_fisher_ind_means(wgs_beagle *b, wgs_afset *a, int32_t i0, int32_t count, float *means_out);
/* ... over SNP shards: NumPy's running float32 total after this shard, continued from the one before (carry_in, NULL on the
 * first shard); the last shard's totals / number of sites (float64 division) are the means. */
int wgs_fisher_ind_sums(wgs_beagle *b, wgs_afset *a, int32_t i0, int32_t count, const float *carry_in, float *sums_out);

/* ------------------------------------------------------------------ streamed Beagle reader (host)
 * reader_cy.readBeagle(path) -- reader_cy.pyx:16-77 -- as a chunked native reader: gzip inflate,
 * lines parsed in parallel into float32 (rows, 2n) chunks the caller owns (and typically hands
 * to wgs_beagle_upload_rows), so files larger than host RAM can be streamed.  Needs no GPU. */
typedef struct wgs_reader wgs_reader;
int wgs_reader_open(const char *path, int threads, wgs_reader **out);      /* parses the header line */
void wgs_reader_close(wgs_reader *r);
int wgs_reader_n_individuals(wgs_reader *r);
const char *wgs_reader_sample_name(wgs_reader *r, int i);
/* Parse up to max_rows further sites into rows[max_rows][2n]; *nrows = sites parsed (0 at EOF). */
int wgs_reader_next(wgs_reader *r, float *rows, int64_t max_rows, int64_t *nrows);
/* Skip up to max_rows sites without parsing them (SNP-sharded reading: each rank parses only its range). */
int wgs_reader_skip(wgs_reader *r, int64_t max_rows, int64_t *nrows);
/* Same, keeping the skipped lines' site names (read them with wgs_reader_chunk_sites): a names-only pass. */
int wgs_reader_skip_names(wgs_reader *r, int64_t max_rows, int64_t *nrows);
/* Site names of the last chunk, '\n'-terminated each, *bytes long. */
const char *wgs_reader_chunk_sites(wgs_reader *r, int64_t *bytes);
/* Number of data lines (sites) of a gzipped Beagle file: one inflate pass, no parsing. */
int wgs_reader_count_sites(const char *path, int64_t *sites);
/* About how many sites a BGZF file holds, from five samples of a quarter megabyte (newlines per compressed byte x file size):
 * milliseconds.  For sizing a device matrix before the exact count is known (wgs_beagle_set_rows trims it afterwards).
 * rc 3: not BGZF. */
int wgs_reader_estimate_sites(const char *path, int64_t *estimate);
/* The same pass can leave an INDEX behind (index_path): header fields, the site count and at most max_points
 * access points about span_bytes of text apart (deflate-block boundaries with their 32 KiB dictionaries; member
 * boundaries of BGZF / concatenated gzip need none), and the site names, '\n'-terminated (names_path) -- so that
 * one pass per file serves the count, the names-only pass of the downsampled-LOO site masks and every rank's
 * start position.  wgs_reader_open_indexed returns a reader whose next row is first_row without inflating what
 * precedes the access point before it: a rank that owns a later SNP range reads only its own range. */
int wgs_reader_build_index(const char *path, const char *index_path, const char *names_path, int64_t span_bytes,
                           int32_t max_points, int64_t *sites);
int wgs_reader_index_sites(const char *path, const char *index_path, int64_t *sites);
/* The same pass over a BGZF file SPLIT over the ranks of a node (and the threads of each): rank `part` of `nparts` finds the
 * first block of its byte range by its 16-byte signature, inflates and summarises its blocks into part_path
 * (= parts_prefix + "." + part); after a barrier one rank chains the parts -- accepted only if every part starts exactly
 * where the one before ended -- into the index.  rc 3: the file cannot be done in parts (not BGZF, a range off the block
 * chain); one rank then calls wgs_reader_build_index. */
int wgs_reader_index_part(const char *path, const char *part_path, int part, int nparts, int threads);
int wgs_reader_index_merge(const char *path, const char *index_path, const char *parts_prefix, int nparts, int64_t span_bytes,
                           int32_t max_points, int64_t *sites);
int wgs_reader_open_indexed(const char *path, const char *index_path, int64_t first_row, int threads, wgs_reader **out);

/* ------------------------------------------------------------------ device-side ingest
 * reader_cy.pyx:52-66 (strtok + atof per value) on the MI355X: the reader only inflates, finds the newlines and keeps
 * the site names; the TEXT goes to the device through page-locked buffers and a HIP kernel tokenises it straight into
 * the population slabs of `b` (csrc/ingest.hip).  Values the kernel does not convert itself (inf/nan, hex, 16+ digits)
 * flag their line, which the host then parses with strtod -- the results equal wgs_reader_next's bit for bit.
 * `r` is a reader positioned at the first wanted row (wgs_reader_open / wgs_reader_open_indexed); after
 * wgs_ingest_create it must only be used through the ingest, which is destroyed BEFORE the reader is closed.
 * limit_rows < 0: to the end of the file.  chunk_bytes <= 0: 256 MiB of text per chunk. */
typedef struct wgs_ingest wgs_ingest;
int wgs_ingest_create(wgs_beagle *b, wgs_reader *r, int64_t limit_rows, int64_t chunk_bytes, wgs_ingest **out);
void wgs_ingest_destroy(wgs_ingest *g);
/* Next chunk: its lines fill the slab rows row0, row0 + 1, ... (keep != NULL: keep[i] != 0 keeps the i-th line of this
 * chunk, dropped lines take no row; keep_len = entries available).  *file_rows = lines consumed (0: end of the file or
 * of the row limit), *rows_written = rows filled.  Synchronises the context's stream. */
int wgs_ingest_next(wgs_ingest *g, int64_t row0, const uint8_t *keep, int64_t keep_len, int64_t *file_rows, int64_t *rows_written);
/* Site names of that chunk's lines (kept or not), '\n'-terminated each, *bytes long. */
const char *wgs_ingest_chunk_sites(wgs_ingest *g, int64_t *bytes);
/* stats[0..13]: seconds the caller waited for the producer thread; producer seconds in inflate / in the newline scan (host
 * inflate); device ms (copies + kernels); lines parsed on the host; text bytes; lines; chunks; ms of the device inflate
 * kernel; BGZF members inflated on the device (csrc/inflate.hip: BGZF files take the device-resident pipeline unless
 * WGSASSIGN_INFLATE=host); members the device left to the host's inflater; producer seconds reading compressed members; seconds inside
 * wgs_ingest_create and inside wgs_ingest_next. */
int wgs_ingest_stats(wgs_ingest *g, double *stats);

/* Blocks of 4096 elements that fell back to the serial loop in the last wgs_em_rmse_chain of an EM batch (diagnostics). */
int wgs_em_last_chain_serial_blocks(wgs_em *em);

/* Kernel time (HIP events on the context's stream) of the context's last wgs_assign / wgs_score_* call; -1 while the class
 * codes' memory is being allocated (wgs_beagle_codes_wait): ask again afterwards. */
int wgs_assign_last_ms(wgs_ctx *ctx, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* WGSASSIGN_HIP_H */
