// Internal declarations shared by the HIP translation units of libwgsassign_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/wgsassign_hip.h"
#include "../../include/wgsassign_hip_debug.h"

void wgs_set_error(const char *fmt, ...);

// (-DWGS_TRACE_STALLS=ms: runtime calls that took longer are named on stderr -- how the wait behind a slow hipMalloc was found)
#ifdef WGS_TRACE_STALLS
#include <chrono>
#include <cstdio>
struct wgs_stall_probe {
    const char *what, *file;
    int line;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~wgs_stall_probe() {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms >= WGS_TRACE_STALLS) fprintf(stderr, "[stall] %.1f ms in %s (%s:%d)\n", ms, what, file, line);
    }
};
#define WGS_STALL_PROBE(expr) wgs_stall_probe probe_{#expr, __FILE__, __LINE__}
#define WGS_STALL_SCOPE(name) wgs_stall_probe scope_probe_{name, __FILE__, __LINE__}
#else
#define WGS_STALL_PROBE(expr)
#define WGS_STALL_SCOPE(name)
#endif

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_;                                                                        \
        {                                                                                     \
            WGS_STALL_PROBE(expr);                                                            \
            e_ = (expr);                                                                      \
        }                                                                                     \
        if (e_ != hipSuccess) {                                                               \
            wgs_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

// hipMalloc with the time it took added to a process-wide total (wgs_malloc_seconds): on this driver an allocation of VRAM that an
// earlier process used is cleared when it is handed out again -- up to ~100 ms per GB, for the same call that takes 0.3 ms on
// untouched memory (profiles/r04_alloc_ubench.txt) -- and a caller timing whole paths wants to know what share that was.
extern std::atomic<long long> g_wgs_malloc_ns;
template <typename T>
static inline hipError_t wgs_malloc(T **p, size_t bytes)
{
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = hipMalloc(p, bytes);
    g_wgs_malloc_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return e;
}

#define WGS_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            wgs_set_error(__VA_ARGS__); \
            return 2;                   \
        }                               \
    } while (0)

// Runs `fn` when the scope is left unless dismissed: every early error return (HIP_TRY /
// WGS_REQUIRE) of a create/upload function then releases what was allocated so far.
template <typename F>
struct ScopeFail {
    F fn;
    bool armed = true;
    explicit ScopeFail(F f) : fn(f) {}
    ~ScopeFail() { if (armed) fn(); }
    void dismiss() { armed = false; }
};
template <typename F>
ScopeFail<F> on_failure(F f) { return ScopeFail<F>(f); }

struct wgs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void *pinned = nullptr;  // small pinned host scratch for async readbacks
    size_t pinned_bytes = 0;
    int cus = 0;
    void *ws = nullptr;      // grow-only device workspace (assignment outputs / pointer tables)
    size_t ws_bytes = 0;
    void *ws_b = nullptr;    // second, small one (slab tables of the encoder's sample pass)
    size_t ws_b_bytes = 0;
    // per-context state that must not be shared between contexts on different devices
    bool log_table_ready = false;               // the log table of assign_kernels.hip is uploaded to this device
    hipEvent_t ev0 = nullptr, ev1 = nullptr;    // bracket the scoring kernels of the last wgs_assign / wgs_score_*
    float last_assign_ms = 0.0f;
    bool assign_ms_pending = false;             // ev0 .. ev1 not read yet: hipEventElapsedTime WAITS for a hipMalloc in flight on another thread
                                                // (tools/ubench_alloc3.hip; kernel launches, copies and synchronisation do not), so elapsed times
                                                // are read when somebody asks and no allocation of this library is in flight (wgs_assign_last_ms)
    std::atomic<int> allocs_in_flight{0};       // helper-thread hipMallocs under way (codes.hip)
    struct PoolBlock {
        void *p;
        size_t bytes;
        bool used;
    };
    std::vector<PoolBlock> pool;                // wgs_pool_malloc's blocks, in use and idle
    int64_t dyn_lds_base = -1;                  // LDS address at which a kernel's dynamic allocation starts (em_coded_usable probes it once; -1: not yet)
    hipEvent_t enc_ev0 = nullptr, enc_ev1 = nullptr;   // bracket the class encoder (its own pair: ev0 / ev1 may still hold an unread scoring time)
};
// Device workspace of at least `bytes` (256-byte aligned); contents are not preserved across calls.
int wgs_ctx_workspace(wgs_ctx *ctx, size_t bytes, void **out);
// Buffers of short-lived objects (a wgs_score lives for one call of --get_pop_like or one leave-one-out batch) come from a small cache
// kept with the context: hipMalloc / hipFree cost ~50-100 us each and hipFree synchronises the device -- a dozen of them were 0.9 ms of
// a 2.7 ms scoring call on one rank's shard of 8 (bench.py: shard_projection).  At most 1 GiB of idle blocks is kept.
hipError_t wgs_pool_malloc(wgs_ctx *ctx, void **p, size_t bytes);
void wgs_pool_free(wgs_ctx *ctx, void *p);
template <typename T>
static inline hipError_t wgs_pool_malloc(wgs_ctx *ctx, T **p, size_t bytes)
{
    return wgs_pool_malloc(ctx, reinterpret_cast<void **>(p), bytes);
}

// One population slab: the (g0,g1) pairs of the individuals of one group (file order), stored
// TILE-INTERLEAVED for lane<->SNP kernels:
//     base[(tile * npairs + pair) * 64 + lane] = float4{g0,g1 of individual 2*pair, g0,g1 of 2*pair+1}
// for SNP 64*tile + lane.  One wave-wide 16-byte load is one aligned, contiguous 1 KiB; a tile
// (64 SNPs x all individuals of the group) is one contiguous npairs KiB block.
struct Slab {
    float4 *base = nullptr;
    int32_t npairs = 0;      // ceil(ncols / 2)
    int32_t ncols = 0;       // individuals in the group
    std::vector<int32_t> members;  // column -> global individual index
    int32_t *d_members = nullptr;
};
static inline int64_t wgs_ntiles(int64_t m) { return (m + 63) / 64; }

// Class codes of a matrix (wgs_beagle_codes): low-depth genotype likelihoods take few distinct (g0, g1) values per SNP
// -- the bundled 85-individual data 29 on average, the 2x synthetic matrices 27 among 1000 individuals, likelihoods from
// binned base qualities ~80 -- and both hot kernels evaluate an expensive function of (g0, g1, per-SNP parameter) per
// individual: the EM term's quotient, the per-site log-likelihood.  With the classes known, that function is evaluated once
// per CLASS and SNP and looked up per individual; the serial float32 accumulation / the float64 sums run over the
// individuals exactly as before, on the very same values, so every result keeps its bits.  Built by ONE pass of
// class_encode_kernel (codes_kernels.hip); policy (when to build, table geometry) in codes.hip.
//   dict   [(tile * drows + class) * 64 + lane]  the (g0, g1) of class `class` of SNP 64 * tile + lane (tile-interleaved)
//   ncls   [SNP]                                 classes of the SNP (<= drows <= 254); 0: the SNP is RICH (too many classes
//                                                for the tables) and every sweep takes it from the float32 slab
//   codes  per slab: [(tile * nquads + quad) * 64 + lane]  the classes of individuals 4 quad .. 4 quad + 3 (one byte
//          each, low byte first) of the slab for SNP 64 * tile + lane: coalesced for lane <-> SNP kernels, and 16
//          consecutive SNPs of one quad are one 64-byte line for lane <-> quad kernels
// and the slab's OWN numbering of the classes (the EM sweep through the codes looks quotients up in a table with one row per
// class PRESENT in the slab, not per class of the SNP):
//   lcodes like codes, a byte = rank of the individual's class among the classes present in the slab at that SNP
//   ldict  [(tile * lrows + rank) * 64 + lane]  (g0, g1) of the rank-th present class of SNP 64 * tile + lane
//   tile_rows [tile * 8 + k]  most classes present in the slab at one of SNPs 8 k .. 8 k + 7 of the tile (as the encoder's wavefronts,
//          which own 8, 16 or 32 SNPs, saw it); 255 when one of them is rich.  A tile whose largest entry exceeds lrows is swept
//          directly from the float32 slab by the coded EM sweep.
#ifndef WGS_ENC_SLOTS
#define WGS_ENC_SLOTS 1024         // hash slots per wavefront of the class encoder (codes_kernels.hip): SNPs x slots per SNP.  1024 = 10 KiB of LDS
                                   // and <= 128 VGPRs: four wavefronts per SIMD (2048: two; measured 45 ms against 33 at 10M x 1000)
#endif
constexpr int WGS_ENC_MIN_SNPS = WGS_ENC_SLOTS / 256;      // SNPs per wavefront with the largest tables (256 slots per SNP)
constexpr int WGS_TILE_ROWS_BYTES = 64 / WGS_ENC_MIN_SNPS;  // tile_rows entries per tile: one per WGS_ENC_MIN_SNPS SNPs
constexpr int WGS_BATCH_ROWS_CAP = 616;  // classes the SNPs of one batch of the coded scoring sweep may sum to: 616 rows of 80 bytes + their 8-byte dictionary entries + the log table fit 64 KiB of LDS
struct SlabCodes {
    uint32_t *codes = nullptr;
    uint32_t *lcodes = nullptr;
    float2 *ldict = nullptr;
    uint8_t *tile_rows = nullptr;  // [tile * WGS_TILE_ROWS_BYTES + k]: for SNPs WGS_ENC_MIN_SNPS k ... of the tile
    int32_t nquads = 0;
    int32_t quad0 = 0;             // first matrix-wide quad index of this slab
};
struct wgs_codes {
    void *pool = nullptr;          // one allocation behind every array below
    int32_t snps_per_wave = 16;    // encoder geometry: 16 / 8 / 4 SNPs per wavefront = hash tables of 64 / 128 / 256 slots
    int32_t drows = 0;             // dictionary rows per tile
    int32_t lrows = 0;             // rows of a slab's own dictionary per tile = rows of the coded EM sweep's table; 0: no local numbering
    int32_t cmax = 0;              // most classes of a coded SNP
    int32_t score_batch = 16;      // SNPs per table of the coded scoring sweep: 16, 8 or 4 (fewer when the SNPs have many classes)
    int32_t rows_batch = 0;        // most classes summed over an aligned group of score_batch SNPs (that sweep's table rows)
    int32_t total_quads = 0;
    float2 *dict = nullptr;
    uint8_t *ncls = nullptr;
    std::vector<SlabCodes> slabs;
    SlabCodes *d_slabs = nullptr;  // device copy
    uint4 *wave_stats = nullptr;   // the encoder's per-wavefront records (two per work unit)
    int64_t bytes = 0, local_bytes = 0;
    int64_t generation = 0;        // distinguishes this build from any earlier one of the same matrix (caches of derived tables)
    double build_ms = 0.0, kernel_ms = 0.0, sample_ms = 0.0, alloc_ms = 0.0, alloc_wait_ms = 0.0;
    double sum_ncls = 0.0;         // over the coded SNPs
    int64_t rich_snps = 0;
    double local_direct_share = 0.0;   // share of the (slab, tile) pairs the coded EM sweep takes from the float32 slab
    bool local_skipped = false;        // built for a scoring sweep: without the slabs' own numbering although the matrix could have one (an EM fit that
                                       // wants it rebuilds: em_api.hip)
    double probe_rounds = 0.0;     // hash probe rounds beyond the first per buffer of 16 lookups (encoder diagnostics)
    double sample_mean_g = 0.0, sample_mean_l = 0.0;   // classes per SNP / per (slab, SNP) in the sample
};

// What the encoder's sample pass found (codes.hip: wgs_beagle_codes_plan): state 0 = not sampled, 1 = worth coding, -1 = not.
struct wgs_codes_plan {
    int state = 0;
    int32_t slots = 64, drows = 0, lrows = 0, score_batch = 16;
    double mean_g = 0.0, mean_l = 0.0, sample_ms = 0.0;
};

struct wgs_beagle {
    wgs_ctx *ctx = nullptr;
    int64_t m = 0, n = 0, site0 = 0;
    int32_t n_groups = 0;
    std::vector<Slab> slabs;
    std::vector<int32_t> group_of, col_of;  // per individual
    // device-side lookup tables for the scatter/gather/synth kernels
    int32_t *d_group_of = nullptr, *d_col_of = nullptr, *d_npairs = nullptr;
    float4 **d_base = nullptr;
    int64_t bytes = 0;
    // class codes, built when a sweep that profits from them asks (wgs_beagle_codes) and dropped when rows change; codes_state:
    // 0 = not tried, 1 = available, -1 = not worth coding (too many classes per SNP in the sample, or no memory): the direct kernels run
    wgs_codes *codes = nullptr;
    int codes_state = 0;
    int64_t codes_generation = 0;  // counts builds and drops
    wgs_codes_plan plan;           // the sample pass's findings (reset when rows change)
    void *pool = nullptr;          // device memory of the class codes, kept across rebuilds
    size_t pool_bytes = 0;
    // ... and how it is come by: hipMalloc on a helper thread (codes.hip: pool_request) -- VRAM that an earlier process used is cleared
    // by the driver when it is handed out again, seconds for tens of GB, and sweeps over the float32 slabs can run meanwhile
    std::thread *pool_thread = nullptr;
    std::atomic<int> pool_state{0};   // 0: nothing requested, 1: in flight, 2: pool_new is ready, -1: no memory
    void *pool_new = nullptr;
    size_t pool_new_bytes = 0, pool_want = 0, pool_want_small = 0;
    double pool_request_s = 0.0, pool_alloc_ms = 0.0;
    std::atomic<int64_t> direct_sweeps{0};     // EM sweeps over the float32 slabs so far (wgs_em_step callers: the codes are built once a run is long)
};
// The matrix's class codes, or nullptr when they are switched off (WGSASSIGN_CODES=0) or the matrix is not worth coding (then the
// direct kernels are used).  build = false only returns codes that exist already.
// wait = false: when the codes' memory is not there yet the call returns nullptr for now (the caller sweeps the float32 slabs)
// and a later call finishes the build.
wgs_codes *wgs_beagle_codes(wgs_beagle *b, bool build = true, bool wait = true, bool for_scoring_only = false);
const wgs_codes_plan *wgs_beagle_codes_plan(wgs_beagle *b);
double wgs_codes_build_ms_estimate(const wgs_beagle *b, int slots, bool with_slab_numbering = true);
double wgs_em_codes_saving(double classes_per_slab, double cols, int lrows);     // share of a float32 fit's time the coded sweeps save (codes.hip)
bool wgs_codes_pay_for_scoring(wgs_beagle *b, int K);
// the numbers that decision is made from: the direct sweep's time, the share of it the coded sweep costs, the encode pass (false: not worth coding)
bool wgs_codes_scoring_model(wgs_beagle *b, int K, double *direct_ms, double *coded_share, double *build_ms);
int wgs_ctx_workspace_b(wgs_ctx *ctx, size_t bytes, void **out);     // a second small grow-only scratch (survives wgs_ctx_workspace calls)
void wgs_beagle_drop_codes(wgs_beagle *b);
void wgs_beagle_release_pool(wgs_beagle *b);
int launch_class_sample(wgs_beagle *b, wgs_codes *c, int max_units, unsigned long long *hist_g, unsigned long long *hist_l, double *rounds_per_buffer);
int launch_class_encode(wgs_beagle *b, wgs_codes *c);

// ---- self-checking collectives (rccl_comm.hip): the row every rank attaches to every collective
enum { WGS_TAG_SEQ = 0, WGS_TAG_OP, WGS_TAG_GEN, WGS_TAG_ITER, WGS_TAG_SHAPE_A, WGS_TAG_SHAPE_B, WGS_TAG_COUNT, WGS_TAG_AUX, WGS_TAG_WORDS };
struct CommRow {
    double w[WGS_TAG_WORDS];       // small integers, exact in float64: a sum all-reduce with zeros from the others all-gathers them
};
struct CommFault {                 // page-locked: written by comm_tag_check_kernel, read by wgs_comm_check
    int flag, claimed, rank, other;
    double mine[WGS_TAG_WORDS], theirs[WGS_TAG_WORDS];
};
constexpr int WGS_COMM_MAX_WORLD = 64;
// float64 a buffer handed to wgs_comm_allreduce_tagged / wgs_comm_bcast_tagged must have room for behind its payload
constexpr size_t wgs_comm_tail_doubles() { return (size_t)WGS_COMM_MAX_WORLD * WGS_TAG_WORDS + 1; }
constexpr size_t wgs_comm_tail_bytes() { return wgs_comm_tail_doubles() * sizeof(double); }
extern "C" int wgs_comm_allreduce_tagged(wgs_comm *c, double *dev_buf, int64_t n, const wgs_coll_tag *tag);
extern "C" int wgs_comm_bcast_tagged(wgs_comm *c, void *dev_buf, int64_t bytes, int root, const wgs_coll_tag *tag);

// ---- objects of the C ABI that point at a parent (an EM batch at its matrix, a score at its matrix and frequency set).  Destroying a
// parent destroys them first, and destroying something that is no longer alive returns at once: a caller's garbage collector may
// release its handles in any order (Python finalises the objects of a reference cycle in arbitrary order; before round 5 an EM batch
// finalised after its matrix was a use after free).
enum { WGS_LIVE_EM = 1, WGS_LIVE_SCORE = 2 };
void wgs_live_add(void *obj, int kind, void *parent, void *parent2 = nullptr);
bool wgs_live_remove(void *obj);               // false: not alive (never created through the ABI, or destroyed already)
void wgs_live_destroy_children(void *parent);

// ---- test hooks (include/wgsassign_hip_debug.h: wgs_debug_hook): process-wide switches only the test suite sets, by name
int64_t wgs_hook(const char *name);        // 0 unless a test set it

struct wgs_afset {
    wgs_ctx *ctx = nullptr;
    int64_t m = 0;
    int32_t K = 0;
    float *buf = nullptr;  // K vectors of m floats, population-major
};

// ---- kernel launchers implemented in the .hip files (all asynchronous on ctx->stream)

struct FitDesc {       // one EM fit as the sweep kernel sees it
    const float4 *slab;
    const float *f_old;
    float *f_new;
    double *ssq;       // = sum over SNPs of (f_new - f_old)^2 (written by the reduce kernel)
    double *ssq_part;  // per-tile partial sums of this fit [ntiles]
    // fused iterations (em_coded_kernel): fuse = 2 runs a second update from f_new into f_new2, its sums in ssq2 / ssq_part2
    float *f_new2;
    double *ssq2;
    double *ssq_part2;
    int32_t fuse;
    int32_t npairs, ncols;
    int32_t skip;      // local column left out (LOO) or -1
    int32_t n_eff;     // ncols - (skip >= 0)
    const int32_t *state;  // device: fit state (EM_ACTIVE / EM_CONVERGED / EM_UNDECIDED) or nullptr = always sweep
    // the slab's class codes in its own numbering (common.h: SlabLocal), or nullptr: then only the direct kernels can take this fit
    const uint32_t *lcodes;
    const float2 *ldict;
    const uint8_t *tile_rows;
    int32_t nquads, lrows;
};
// (_A: of the two iterations of a fused sweep the FIRST converged / is undecided; the plain values then speak of the second)
enum { EM_ACTIVE = 0, EM_CONVERGED = 1, EM_UNDECIDED = 2, EM_CONVERGED_A = 3, EM_UNDECIDED_A = 4 };
// One exact convergence chain (emMAF_cy.pyx:30-31 over this shard): float32 running sum of (a-b)^2 from carry_in.
struct ChainJob {
    const float *a, *b;
    float carry_in;
};

struct FisherDesc {    // one population of the --ne_obs sweep
    const float4 *slab;
    const float *th;       // the population's (clamped) allele frequencies, m floats
    float *f_out, *ne_out; // observed Fisher information / effective sample size per SNP
    int32_t npairs, ncols;
};
int launch_fisher_pop(wgs_ctx *ctx, const FisherDesc *d_descs, int32_t n_desc, int64_t m);
int launch_fisher_ind_sites(wgs_ctx *ctx, const float4 *slab, const int32_t *d_cols, const float *th, float *d_out, int64_t m,
                            int npairs, int count);
int launch_pairwise_mean(wgs_ctx *ctx, const float *d_rows, int count, int64_t m, int64_t divide_by, const int64_t *d_leaf_lo,
                         const int32_t *d_leaf_len, int nleaf, const int32_t *d_prog, int nprog, float *d_leaf_sums, const float *d_carry,
                         float *d_means);
int launch_em_sweep(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, int64_t m, int mode);
// the same sweep through the class codes (exact mode; every fit's descriptor carries its slab's local codes; rows = the
// largest SlabLocal::rows among the fits)
int launch_em_coded(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, int64_t m, int rows);
bool em_coded_usable(wgs_ctx *ctx);
int launch_em_coded_groups(wgs_ctx *ctx, const FitDesc *d_descs, const int32_t *d_groups, int32_t n_groups, int64_t m, int rows);
int em_fits_per_group(void);
int launch_em_sweep_groups(wgs_ctx *ctx, const FitDesc *d_descs, const int32_t *d_groups, int32_t n_groups, int64_t m, int mode);
int ssq_reduce_chunks(void);
int launch_ssq_reduce(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, int64_t m, double *part2, int second = 0);
// state[fit] of every listed fit that swept: ssq < lo -> EM_CONVERGED, ssq >= hi (or NaN) -> EM_ACTIVE, else EM_UNDECIDED
int launch_em_decide(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, double lo, double hi);
int launch_rcp_error(wgs_ctx *ctx, int exponent, unsigned long long *d_max_bits);
int launch_div_check(wgs_ctx *ctx, unsigned long long seed, unsigned long long per_thread, unsigned long long *d_mismatch);
int launch_fill(wgs_ctx *ctx, float *p, int64_t count, float v);
int launch_clamp(wgs_ctx *ctx, float *p, int64_t count, float lo, float hi);
int launch_rmse_chain_serial(wgs_ctx *ctx, const float *a, const float *b, int64_t m, float carry_in, float *d_out);
size_t rmse_chain_workspace_bytes(int64_t m);
int launch_rmse_chain(wgs_ctx *ctx, const float *a, const float *b, int64_t m, float carry_in, float *d_out, void *work,
                      int *d_serial);
// n_jobs chains at once (device array of jobs): d_out[j] / d_serial[j]; work: n_jobs * rmse_chain_workspace_bytes(m)
int launch_chain_set_carry(wgs_ctx *ctx, ChainJob *d_jobs, const float *d_carry, int n_jobs);
int launch_rmse_chain_batch(wgs_ctx *ctx, const ChainJob *d_jobs, int n_jobs, int64_t m, float *d_out, void *work, int *d_serial);

struct AssignArgs {
    const float4 *slab;
    const int32_t *members;        // device: slab column -> global individual
    const float *const *colptr;    // device: [n*K] per-(individual,k) vectors, or nullptr
    const float *const *acol;      // device: [K] shared vectors
    double *out;                   // device: [(n*P) * K]
    int64_t m, site0;
    int32_t npairs, ncols, K, P;
    int32_t tiles_per_wave;
};
int launch_assign(wgs_ctx *ctx, const AssignArgs &a, int mode);
struct PartsSlab {     // one population slab inside the single exact-partition launch
    const float4 *slab;
    const int32_t *members;
    int32_t npairs, ncols;
    int32_t block0;    // first workgroup (64 columns each) of this slab
};
int launch_parts_exact(wgs_ctx *ctx, const AssignArgs &a, const PartsSlab *d_slabs, int n_slabs, int total_blocks,
                       const float *d_carry, float *d_parts);

// ---- scoring: all n x K sums in one launch over a table of population slabs ----------------------
// A launch scores the individuals [row_lo, row_hi) (file order; all of them in the common case).  The
// work unit of a wavefront is (pair group, block): NP pairs of slab columns x one BLOCK of
// WGS_BLOCK_TILES tiles (4096 SNPs).  Its float64 sums over the block are STORED to S[block][cell]
// (cell = individual * K + population) -- one writer per element, no atomics, so the n x K sums
// (S added over blocks in block order) are reproducible bit for bit.  The same block structure
// carries the exact partition chains (see chain_cand_kernel).
constexpr int WGS_BLOCK_TILES = 64;            // tiles per block: 4096 SNPs
struct ScoreSlab {
    const float4 *slab;
    const int32_t *members;        // slab column -> global individual
    int32_t npairs, ncols;
    int32_t pair0;                 // first pair holding a scored column
    int32_t npg;                   // pair groups (NP pairs each) holding scored columns
    int32_t pg0;                   // first launch-wide pair-group index of this slab
    int32_t col_lo, col_hi;        // scored columns [col_lo, col_hi)
};
struct ScoreArgs {
    const ScoreSlab *slabs;        // device
    int32_t n_slabs, total_pg;
    const float *const *colptr;    // device: [n*K] per-(individual, k) vectors, or nullptr
    const float *const *acol;      // device: [K] shared vectors
    int64_t m, site0, cells;       // cells = n * K
    int32_t K, P, period;          // period = P / gcd(64, P): tiles t and t + period give a lane the same label
    int32_t nblocks;
    double *S;                     // device: [nblocks][cells] block sums, then exclusive prefixes
    const double *start;           // device: [cells] sum over the preceding SNP shards (or nullptr)
    uint32_t *cand;                // device: [cells * P][nblocks] packed block functions
};
struct WalkArgs {
    const uint32_t *cand;          // [chains][nblocks]
    const float *carry;            // [chains] running values after the preceding shards, or nullptr
    float *parts;                  // [chains] out
    const int32_t *group_of, *col_of, *npairs;   // per individual / per slab (wgs_beagle tables)
    float4 *const *base;
    const float *const *colptr, *const *acol;
    int64_t m, site0;
    int32_t n, K, P, nblocks, row_lo, row_hi;
    int32_t *n_serial;             // device counter: blocks that took the literal serial loop (or nullptr)
};
int score_pairs_per_wave(int K, bool per_ind); // NP of the sweep for K populations (depends on the register batch KB)
int chain_pairs_per_wave(int K, bool per_ind); // NP of the chain kernel (the slab table must be built for it)
int launch_score_sweep(wgs_ctx *ctx, const ScoreArgs &a, int mode);
struct CodedSlabHost {             // = CodedSlab of assign_kernels.hip
    const uint32_t *codes;
    const int32_t *members;
    const float4 *slab;            // the float32 slab: SNPs the encoder left uncoded (ncls = 0) are scored from it
    int32_t nquads, ncols, quad0, col_lo, col_hi, npairs;
};
int score_kb(int K);                                   // populations per pass of the scoring sweeps
size_t score_coded_lds_bytes(int rows, int kb, int batch);   // LDS of the coded sweep for a matrix whose richest batch of SNPs has `rows` classes
int launch_score_coded(wgs_ctx *ctx, const wgs_codes *c, const void *d_slabs, int n_slabs, int total_quads, const float *const *d_acol,
                       int64_t m, int64_t cells, int K, int nblocks, double *S, int mode);
int launch_block_prefix(wgs_ctx *ctx, double *S, int nblocks, int64_t cells, double *out, int keep_prefix, double *chunks);
int launch_chunk_total(wgs_ctx *ctx, const double *chunks, int nchunks, int64_t cells, const double *carry, double *out);
size_t chain_cand_lds_bytes(int K, int P, bool per_ind);
int launch_chain_cand(wgs_ctx *ctx, const ScoreArgs &a);
int launch_chain_walk(wgs_ctx *ctx, const WalkArgs &w);
int launch_log_mismatch(wgs_ctx *ctx, unsigned int b0, unsigned int b1, unsigned long long *d_count, unsigned int *d_first);
int launch_log_values(wgs_ctx *ctx, const float *d_x, float *d_out, int64_t n, int use_libm);
int launch_loglike_site(wgs_ctx *ctx, const float2 *g, const float *a, float *vec, int64_t m, int mode);

int launch_scatter_rows(wgs_beagle *b, const float *d_rows, int64_t row0, int64_t nrows);
int launch_gather_rows(wgs_beagle *b, float *d_rows, int64_t row0, int64_t nrows);
int launch_synth(wgs_beagle *b, uint64_t seed, double depth);
int launch_synth_quality(wgs_beagle *b, uint64_t seed, double depth, int nq, const double *quals, const double *probs);
int launch_transpose_mK_to_Km(wgs_ctx *ctx, const float *src_mK, float *dst_Km, int64_t m, int32_t K);
int launch_transpose_Km_to_mK(wgs_ctx *ctx, const float *src_Km, float *dst_mK, int64_t m, int32_t K);
