// C ABI of libwgsassign_hip.so (see include/wgsassign_hip.h).  Host-side orchestration only:
// all arithmetic of the hot path runs in the kernels of em_kernels.hip / assign_kernels.hip.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>

#include "build_id.h"
#include "common.h"

static thread_local std::string g_err;

void wgs_set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" {

const char *wgs_last_error(void) { return g_err.c_str(); }
int wgs_version(void) { return WGS_ABI_VERSION; }
const char *wgs_build_id(void) { return WGS_BUILD_ID; }
const char *wgs_kernels_id(void) { return WGS_KERNELS_ID; }

int wgs_device_count(int *count)
{
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}

int wgs_ctx_create(int device, wgs_ctx **out)
{
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    WGS_REQUIRE(n > 0, "no HIP device visible: libwgsassign_hip needs an AMD GPU (gfx950); there is no CPU fallback");
    WGS_REQUIRE(device >= 0 && device < n, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    wgs_ctx *c = new wgs_ctx();
    auto guard = on_failure([&] { wgs_ctx_destroy(c); });
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->pinned_bytes = 1 << 20;
    HIP_TRY(hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->cus = prop.multiProcessorCount;
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    guard.dismiss();
    *out = c;
    return 0;
}

void wgs_ctx_destroy(wgs_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

}   // extern "C"

int wgs_ctx_workspace(wgs_ctx *ctx, size_t bytes, void **out)
{
    if (bytes > ctx->ws_bytes) {
        if (ctx->ws) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(ctx->ws));
            ctx->ws = nullptr;
            ctx->ws_bytes = 0;
        }
        const size_t want = (bytes + (1u << 20)) & ~((size_t)(1u << 20) - 1);
        HIP_TRY(hipMalloc(&ctx->ws, want));
        ctx->ws_bytes = want;
    }
    *out = ctx->ws;
    return 0;
}

extern "C" {

int wgs_ctx_sync(wgs_ctx *ctx)
{
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

void *wgs_ctx_stream(wgs_ctx *ctx) { return (void *)ctx->stream; }

int wgs_ctx_mem_info(wgs_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes)
{
    WGS_REQUIRE(ctx, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return 0;
}

int wgs_ctx_info(wgs_ctx *ctx, char *name, int name_len, int *cus, int64_t *mem_bytes)
{
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (mem_bytes) *mem_bytes = (int64_t)prop.totalGlobalMem;
    return 0;
}

/* ------------------------------------------------------------------ beagle */

}   // extern "C"

// (class codes: codes.hip)

extern "C" {

void wgs_beagle_destroy(wgs_beagle *b)
{
    if (!b) return;
    wgs_beagle_drop_codes(b);
    (void)hipSetDevice(b->ctx->device);
    for (auto &s : b->slabs) {
        if (s.base) (void)hipFree(s.base);
        if (s.d_members) (void)hipFree(s.d_members);
    }
    if (b->d_group_of) (void)hipFree(b->d_group_of);
    if (b->d_col_of) (void)hipFree(b->d_col_of);
    if (b->d_npairs) (void)hipFree(b->d_npairs);
    if (b->d_base) (void)hipFree(b->d_base);
    delete b;
}

int wgs_beagle_create(wgs_ctx *ctx, int64_t m, int64_t n, const int32_t *group_of, int32_t n_groups, int64_t site0,
                      wgs_beagle **out)
{
    WGS_REQUIRE(ctx && out, "null argument");
    WGS_REQUIRE(m > 0 && n > 0, "beagle matrix must have m > 0 and n > 0 (got m=%lld n=%lld)", (long long)m, (long long)n);
    WGS_REQUIRE(n < (1 << 30), "too many individuals");
    if (!group_of) n_groups = 1;
    WGS_REQUIRE(n_groups >= 1, "n_groups must be >= 1");
    HIP_TRY(hipSetDevice(ctx->device));
    wgs_beagle *b = new wgs_beagle();
    auto guard = on_failure([&] { wgs_beagle_destroy(b); });
    b->ctx = ctx;
    b->m = m;
    b->n = n;
    b->site0 = site0;
    b->n_groups = n_groups;
    b->slabs.resize(n_groups);
    b->group_of.resize(n);
    b->col_of.resize(n);
    for (int64_t i = 0; i < n; ++i) {
        const int g = group_of ? group_of[i] : 0;
        if (g < 0 || g >= n_groups) {
            wgs_set_error("group_of[%lld] = %d out of range (0..%d)", (long long)i, g, n_groups - 1);
            return 2;
        }
        b->group_of[i] = g;
        b->col_of[i] = (int32_t)b->slabs[g].members.size();
        b->slabs[g].members.push_back((int32_t)i);
    }
    std::vector<float4 *> bases(n_groups, nullptr);
    std::vector<int32_t> nps(n_groups, 0);
    for (int g = 0; g < n_groups; ++g) {
        Slab &s = b->slabs[g];
        s.ncols = (int32_t)s.members.size();
        s.npairs = (s.ncols + 1) / 2;
        if (s.ncols == 0) continue;
        const size_t bytes = (size_t)wgs_ntiles(m) * s.npairs * 64 * sizeof(float4);
        if (hipMalloc(&s.base, bytes) != hipSuccess) {
            wgs_set_error("hipMalloc of %zu bytes for population slab %d failed", bytes, g);
            return 1;
        }
        (void)hipMemsetAsync(s.base, 0, bytes, ctx->stream);
        b->bytes += (int64_t)bytes;
        bases[g] = s.base;
        nps[g] = s.npairs;
        if (hipMalloc(&s.d_members, sizeof(int32_t) * s.ncols) != hipSuccess ||
            hipMemcpy(s.d_members, s.members.data(), sizeof(int32_t) * s.ncols, hipMemcpyHostToDevice) != hipSuccess) {
            wgs_set_error("could not upload the member table of slab %d", g);
            return 1;
        }
    }
    HIP_TRY(hipMalloc(&b->d_group_of, sizeof(int32_t) * n));
    HIP_TRY(hipMalloc(&b->d_col_of, sizeof(int32_t) * n));
    HIP_TRY(hipMalloc(&b->d_npairs, sizeof(int32_t) * n_groups));
    HIP_TRY(hipMalloc(&b->d_base, sizeof(float4 *) * n_groups));
    HIP_TRY(hipMemcpy(b->d_group_of, b->group_of.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_col_of, b->col_of.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_npairs, nps.data(), sizeof(int32_t) * n_groups, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_base, bases.data(), sizeof(float4 *) * n_groups, hipMemcpyHostToDevice));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    guard.dismiss();
    *out = b;
    return 0;
}

int64_t wgs_beagle_bytes(const wgs_beagle *b) { return b ? b->bytes : 0; }

static int64_t staging_rows(const wgs_beagle *b, int64_t nrows)
{
    const int64_t row_bytes = b->n * 2 * (int64_t)sizeof(float);
    int64_t r = (256ll << 20) / row_bytes;
    if (r < 1) r = 1;
    return std::min(r, nrows);
}

int wgs_beagle_upload_rows(wgs_beagle *b, const float *L_rows, int64_t row0, int64_t nrows)
{
    WGS_REQUIRE(b && L_rows, "null argument");
    WGS_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= b->m, "row range [%lld, %lld) outside 0..%lld", (long long)row0,
                (long long)(row0 + nrows), (long long)b->m);
    if (nrows == 0) return 0;
    HIP_TRY(hipSetDevice(b->ctx->device));
    wgs_beagle_drop_codes(b);                        // the matrix changes: its class codes are rebuilt on next use
    const int64_t chunk = staging_rows(b, nrows);
    const size_t row_bytes = (size_t)b->n * 2 * sizeof(float);
    void *ws = nullptr;
    if (wgs_ctx_workspace(b->ctx, chunk * row_bytes, &ws)) return 1;
    float *d_stage = reinterpret_cast<float *>(ws);
    int rc = 0;
    for (int64_t r = 0; r < nrows && !rc; r += chunk) {
        const int64_t cnt = std::min(chunk, nrows - r);
        if (hipMemcpyAsync(d_stage, L_rows + (size_t)r * b->n * 2, cnt * row_bytes, hipMemcpyHostToDevice, b->ctx->stream) != hipSuccess) {
            wgs_set_error("upload of rows failed");
            rc = 1;
            break;
        }
        rc = launch_scatter_rows(b, d_stage, row0 + r, cnt);
        if (!rc && hipStreamSynchronize(b->ctx->stream) != hipSuccess) {
            wgs_set_error("scatter kernel failed");
            rc = 1;
        }
    }
    return rc;
}

int wgs_beagle_download_rows(wgs_beagle *b, float *L_rows, int64_t row0, int64_t nrows)
{
    WGS_REQUIRE(b && L_rows, "null argument");
    WGS_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= b->m, "row range outside matrix");
    if (nrows == 0) return 0;
    HIP_TRY(hipSetDevice(b->ctx->device));
    const int64_t chunk = staging_rows(b, nrows);
    const size_t row_bytes = (size_t)b->n * 2 * sizeof(float);
    void *ws = nullptr;
    if (wgs_ctx_workspace(b->ctx, chunk * row_bytes, &ws)) return 1;
    float *d_stage = reinterpret_cast<float *>(ws);
    int rc = 0;
    for (int64_t r = 0; r < nrows && !rc; r += chunk) {
        const int64_t cnt = std::min(chunk, nrows - r);
        rc = launch_gather_rows(b, d_stage, row0 + r, cnt);
        if (rc) break;
        if (hipMemcpyAsync(L_rows + (size_t)r * b->n * 2, d_stage, cnt * row_bytes, hipMemcpyDeviceToHost, b->ctx->stream) != hipSuccess ||
            hipStreamSynchronize(b->ctx->stream) != hipSuccess) {
            wgs_set_error("download of rows failed");
            rc = 1;
        }
    }
    return rc;
}

int wgs_beagle_synth(wgs_beagle *b, uint64_t seed, double depth)
{
    WGS_REQUIRE(b, "null argument");
    HIP_TRY(hipSetDevice(b->ctx->device));
    wgs_beagle_drop_codes(b);
    return launch_synth(b, seed, depth);
}

int wgs_beagle_synth_quality(wgs_beagle *b, uint64_t seed, double depth, int32_t n_bins, const double *quals, const double *probs)
{
    WGS_REQUIRE(b, "null argument");
    HIP_TRY(hipSetDevice(b->ctx->device));
    wgs_beagle_drop_codes(b);
    return launch_synth_quality(b, seed, depth, n_bins, quals, probs);
}

/* ------------------------------------------------------------------ EM */

struct wgs_em {
    wgs_beagle *b = nullptr;
    int32_t n_fits = 0;
    int mode = WGS_MODE_EXACT;
    std::vector<int32_t> group, skip_local, n_eff;
    std::vector<uint8_t> cur, active;
    float *fbuf[2] = {nullptr, nullptr};  // 2 x n_fits x m
    FitDesc *d_descs = nullptr;
    FitDesc *h_descs = nullptr;           // pinned
    int32_t *d_groups = nullptr, *h_groups = nullptr;         // (first, count) pairs of the fit-group sweep, step path
    int32_t *d_groups2[2] = {nullptr, nullptr}, *h_groups2[2] = {nullptr, nullptr};   // ... wgs_em_fit slots
    double *d_ssq = nullptr;
    double *d_part = nullptr;             // n_fits x ntiles per-tile partial sums
    double *d_part2 = nullptr;            // n_fits x ssq_reduce_chunks() slice sums
    float *d_carry = nullptr;             // [0] carry out, [1] (as int) serial-block count
    void *d_chain_work = nullptr;
    std::vector<int32_t> last;            // fits swept by the last step
    int last_chain_serial_blocks = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // bracket the sweep kernel(s) of the last step
    // wgs_em_fit: device-side fit states, two-slot pinned rings for the one-iteration lookahead
    int32_t *d_state = nullptr;           // [n_fits] EM_ACTIVE / EM_CONVERGED / EM_UNDECIDED
    FitDesc *d_descs2[2] = {nullptr, nullptr}, *h_descs2[2] = {nullptr, nullptr};
    int32_t *h_state[2] = {nullptr, nullptr}, *h_setstate = nullptr;
    double *d_ssq2 = nullptr;             // [n_fits] sums of the iteration in flight
    hipEvent_t ev_it[2] = {nullptr, nullptr};
    hipEvent_t ev_sw0[2] = {nullptr, nullptr}, ev_sw1[2] = {nullptr, nullptr};   // bracket the sweep kernel(s) of a slot
    double fit_sweep_ms = 0.0;            // summed sweep-kernel time of the last wgs_em_fit (HIP events)
    ChainJob *d_jobs = nullptr, *h_jobs = nullptr;
    float *d_chain_out = nullptr, *h_chain_out = nullptr;     // [n_fits] carries | [n_fits] serial-block counts
    void *d_chain_batch = nullptr;
    size_t chain_batch_jobs = 0;
    double fit_seconds = 0.0;
    int fit_iterations = 0, fit_chain_batches = 0;
};

void wgs_em_destroy(wgs_em *em)
{
    if (!em) return;
    (void)hipSetDevice(em->b->ctx->device);
    for (int i = 0; i < 2; ++i)
        if (em->fbuf[i]) (void)hipFree(em->fbuf[i]);
    if (em->d_descs) (void)hipFree(em->d_descs);
    if (em->h_descs) (void)hipHostFree(em->h_descs);
    if (em->d_groups) (void)hipFree(em->d_groups);
    if (em->h_groups) (void)hipHostFree(em->h_groups);
    for (int i = 0; i < 2; ++i) {
        if (em->d_groups2[i]) (void)hipFree(em->d_groups2[i]);
        if (em->h_groups2[i]) (void)hipHostFree(em->h_groups2[i]);
    }
    if (em->d_ssq) (void)hipFree(em->d_ssq);
    if (em->d_part) (void)hipFree(em->d_part);
    if (em->d_part2) (void)hipFree(em->d_part2);
    if (em->d_carry) (void)hipFree(em->d_carry);
    if (em->d_chain_work) (void)hipFree(em->d_chain_work);
    if (em->ev0) (void)hipEventDestroy(em->ev0);
    if (em->ev1) (void)hipEventDestroy(em->ev1);
    for (int i = 0; i < 2; ++i) {
        if (em->d_descs2[i]) (void)hipFree(em->d_descs2[i]);
        if (em->h_descs2[i]) (void)hipHostFree(em->h_descs2[i]);
        if (em->h_state[i]) (void)hipHostFree(em->h_state[i]);
        if (em->ev_it[i]) (void)hipEventDestroy(em->ev_it[i]);
        if (em->ev_sw0[i]) (void)hipEventDestroy(em->ev_sw0[i]);
        if (em->ev_sw1[i]) (void)hipEventDestroy(em->ev_sw1[i]);
    }
    for (void *p : {(void *)em->d_state, (void *)em->d_ssq2, (void *)em->d_jobs, (void *)em->d_chain_out, em->d_chain_batch})
        if (p) (void)hipFree(p);
    for (void *p : {(void *)em->h_jobs, (void *)em->h_chain_out, (void *)em->h_setstate})
        if (p) (void)hipHostFree(p);
    delete em;
}

int wgs_em_create(wgs_beagle *b, int32_t n_fits, const int32_t *fit_group, const int32_t *fit_skip, int mode, wgs_em **out)
{
    WGS_REQUIRE(b && fit_group && out, "null argument");
    WGS_REQUIRE(n_fits > 0, "n_fits must be positive");
    WGS_REQUIRE(mode == WGS_MODE_EXACT || mode == WGS_MODE_FAST, "unknown mode %d", mode);
    HIP_TRY(hipSetDevice(b->ctx->device));
    wgs_em *em = new wgs_em();
    auto guard = on_failure([&] { wgs_em_destroy(em); });
    em->b = b;
    em->n_fits = n_fits;
    em->mode = mode;
    em->group.resize(n_fits);
    em->skip_local.resize(n_fits);
    em->n_eff.resize(n_fits);
    em->cur.assign(n_fits, 0);
    em->active.assign(n_fits, 1);
    for (int j = 0; j < n_fits; ++j) {
        const int g = fit_group[j];
        if (g < 0 || g >= b->n_groups || b->slabs[g].ncols == 0) {
            wgs_set_error("fit %d: group %d is out of range or empty", j, g);
            return 2;
        }
        int skip = -1;
        if (fit_skip && fit_skip[j] >= 0) {
            const int i = fit_skip[j];
            if (i >= b->n || b->group_of[i] != g) {
                wgs_set_error("fit %d: left-out individual %d does not belong to group %d", j, i, g);
                return 2;
            }
            skip = b->col_of[i];
        }
        em->group[j] = g;
        em->skip_local[j] = skip;
        em->n_eff[j] = b->slabs[g].ncols - (skip >= 0 ? 1 : 0);
    }
    const size_t fbytes = (size_t)n_fits * b->m * sizeof(float);
    for (int i = 0; i < 2; ++i) {
        if (hipMalloc(&em->fbuf[i], fbytes) != hipSuccess) {
            wgs_set_error("hipMalloc of %zu bytes for EM frequencies failed", fbytes);
            return 1;
        }
    }
    HIP_TRY(hipMalloc(&em->d_descs, sizeof(FitDesc) * n_fits));
    HIP_TRY(hipMalloc(&em->d_ssq, sizeof(double) * n_fits));
    HIP_TRY(hipMalloc(&em->d_part, sizeof(double) * (size_t)n_fits * wgs_ntiles(b->m)));
    HIP_TRY(hipMalloc(&em->d_part2, sizeof(double) * (size_t)n_fits * ssq_reduce_chunks()));
    HIP_TRY(hipMalloc(&em->d_carry, 2 * sizeof(float)));
    HIP_TRY(hipMalloc(&em->d_chain_work, rmse_chain_workspace_bytes(b->m)));
    HIP_TRY(hipEventCreate(&em->ev0));
    HIP_TRY(hipEventCreate(&em->ev1));
    HIP_TRY(hipHostMalloc(&em->h_descs, sizeof(FitDesc) * n_fits, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&em->d_groups, sizeof(int32_t) * 2 * n_fits));
    HIP_TRY(hipHostMalloc(&em->h_groups, sizeof(int32_t) * 2 * n_fits, hipHostMallocDefault));
    if (launch_fill(b->ctx, em->fbuf[0], (int64_t)n_fits * b->m, 0.25f)) return 1;   // emMAF.py:17-18
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    guard.dismiss();
    *out = em;
    return 0;
}

static float *em_f(wgs_em *em, int fit, int which) { return em->fbuf[which] + (size_t)fit * em->b->m; }

/* Whether building the class codes pays for the EM sweeps still to come (codes.hip builds them in one pass over the matrix):
 *   the encode pass costs about the matrix's bytes at 1.8 TB/s (measured: 80 GB in 44 ms, 1.6 GB in 2.3 ms) + 0.6 ms of
 *   sample pass, allocation and readbacks;
 *   a coded sweep saves a share of the direct sweep (the slabs' bytes at ~6 TB/s) that grows with the population size --
 *   measured 14 % at 30 individuals, 21 % at 36, 39 % at 62, 52 % at 100;
 *   sweeps to come: what the caller knows -- wgs_em_fit its iteration limit, of which a fit rarely uses more than ~14 (the
 *   reference's default tolerance: 13-17 iterations on every data set here); a step-by-step caller nothing, so there a matrix
 *   that has been swept directly three times is taken to be in a long run.
 * WGSASSIGN_EM_CODES_SWEEPS=k replaces the model by "k or more sweeps ahead" (0: always; tests). */
static bool em_codes_pay(const wgs_em *em, const std::vector<int32_t> &order, int fewest_cols, int sweeps_ahead)
{
    const wgs_beagle *b = em->b;
    if (const char *sw = getenv("WGSASSIGN_EM_CODES_SWEEPS")) return sweeps_ahead >= atoi(sw) || b->direct_sweeps >= 3;
    double ahead = std::min(sweeps_ahead, 14);
    if (sweeps_ahead <= 0 && b->direct_sweeps >= 3) ahead = 12;
    static const double at[5] = {28, 36, 62, 100, 1e9}, share[5] = {0.10, 0.21, 0.39, 0.52, 0.52};
    double saves = share[0];
    for (int i = 0; i + 1 < 5; ++i)
        if (fewest_cols >= at[i]) saves = share[i] + (share[i + 1] - share[i]) * std::min(1.0, (fewest_cols - at[i]) / (at[i + 1] - at[i]));
    double swept = 0.0;
    for (int j : order) swept += 8.0 * (double)b->slabs[em->group[j]].ncols * (double)b->m;
    const double direct_ms = swept / 6.0e9, build_ms = (double)b->bytes / 1.8e9 + 0.6;
    return ahead * saves * direct_ms > build_ms;
}

/* Enqueue one sweep (+ the fixed-order reduction of its sums) for the fits in `list`: descriptors into the pinned
 * array H and from there to D.  Fits of different populations stream their slabs once (nontemporal loads); when
 * several fits share a slab (leave-one-out batches) they are ordered by slab and swept in groups of up to
 * em_fits_per_group() per wavefront (group table Hg -> Dg), which share the tile's loads and conversions.
 * ssq_base[j] receives fit j's sum; state_base (device, may be NULL) holds the fit states a sweep honours. */
static int em_enqueue_sweep(wgs_em *em, const std::vector<int32_t> &list, FitDesc *H, FitDesc *D, int32_t *Hg, int32_t *Dg,
                            double *ssq_base, int32_t *state_base, hipEvent_t ev0, hipEvent_t ev1, int sweeps_ahead)
{
    wgs_ctx *ctx = em->b->ctx;
    const int64_t ntiles = wgs_ntiles(em->b->m);
    std::vector<int32_t> order(list);
    std::vector<char> seen(em->b->n_groups, 0);
    bool shared = false;
    for (int j : order) {
        shared = shared || seen[em->group[j]];
        seen[em->group[j]] = 1;
    }
    if (shared) std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return em->group[x] < em->group[y]; });
    // exact mode on a coded matrix: the sweep through the class codes (same frequencies, bit for bit)
    // -- for fits of different slabs; leave-one-out batches (several fits per slab) stay with em_sweep_group_kernel,
    // whose shared loads and conversions serve them better than a quotient table per fit
    // -- and small populations stay with em_sweep_kernel too: below ~28 individuals the table costs more than it saves
    // (measured: 20 individuals 0.98x, 30 1.16x, 36 1.26x, 62 1.64x, 100 2.1x)
    // -- and the codes are BUILT for it only when the sweeps still to come repay the encode pass (em_codes_pay below).
    // Codes that exist already (a scoring sweep built them, or wgs_beagle_codes_prepare) are used at once.
    bool worth = em->mode == WGS_MODE_EXACT && !shared;
    const char *min_env = getenv("WGSASSIGN_EM_CODES_MIN");    // tests lower it to run small populations through the codes
    const int min_cols = min_env ? atoi(min_env) : 28;
    int fewest = INT32_MAX;
    for (int j : order) fewest = std::min(fewest, (int)em->b->slabs[em->group[j]].ncols);
    worth = worth && fewest >= min_cols;
    const bool build = worth && em_codes_pay(em, order, fewest, sweeps_ahead);
    wgs_codes *codes = worth ? wgs_beagle_codes(em->b, build) : nullptr;
    if (codes && codes->lrows == 0) codes = nullptr;
    if (worth && !codes) ++em->b->direct_sweeps;          // (a sweep the codes could have served)
    int coded_rows_max = 0;
    for (size_t i = 0; i < order.size(); ++i) {
        const int j = order[i];
        const Slab &s = em->b->slabs[em->group[j]];
        FitDesc &d = H[i];
        d.lcodes = codes ? codes->slabs[em->group[j]].lcodes : nullptr;
        d.ldict = codes ? codes->slabs[em->group[j]].ldict : nullptr;
        d.lrows = codes ? codes->lrows : 0;
        d.tile_rows = codes ? codes->slabs[em->group[j]].tile_rows : nullptr;
        d.nquads = codes ? codes->slabs[em->group[j]].nquads : 0;
        coded_rows_max = std::max(coded_rows_max, (int)d.lrows);
        d.slab = s.base;
        d.f_old = em_f(em, j, em->cur[j]);
        d.f_new = em_f(em, j, em->cur[j] ^ 1);
        d.ssq = ssq_base + j;
        d.ssq_part = em->d_part + (size_t)j * ntiles;
        d.npairs = s.npairs;
        d.ncols = s.ncols;
        d.skip = em->skip_local[j];
        d.n_eff = em->n_eff[j];
        d.state = state_base ? state_base + j : nullptr;
    }
    // H (pinned) stays untouched until the caller has waited for this sweep
    HIP_TRY(hipMemcpyAsync(D, H, sizeof(FitDesc) * order.size(), hipMemcpyHostToDevice, ctx->stream));
    int32_t n_groups = 0;
    if (shared && !codes) {
        const int fg = em_fits_per_group();
        for (size_t i = 0; i < order.size();) {
            size_t k = i + 1;
            while (k < order.size() && (int)(k - i) < fg && em->group[order[k]] == em->group[order[i]]) ++k;
            Hg[2 * n_groups] = (int32_t)i;
            Hg[2 * n_groups + 1] = (int32_t)(k - i);
            ++n_groups;
            i = k;
        }
        HIP_TRY(hipMemcpyAsync(Dg, Hg, sizeof(int32_t) * 2 * n_groups, hipMemcpyHostToDevice, ctx->stream));
    }
    if (ev0) HIP_TRY(hipEventRecord(ev0, ctx->stream));
    const int64_t per_unit = ((ntiles + 3) / 4 + 7) / 8 * 8 + 8;       // workgroups per fit / per group: slices stay below 2^31
    const size_t max_units = (size_t)std::max<int64_t>(1, ((1ll << 31) - 1) / per_unit);
    if (codes) {
        const int64_t per_fit = (ntiles + 7) / 8 * 8 + 8;            // at least one tile per workgroup
        const size_t max_fits = (size_t)std::max<int64_t>(1, ((1ll << 31) - 1) / per_fit);
        for (size_t off = 0; off < order.size(); off += max_fits) {
            const int cnt = (int)std::min<size_t>(max_fits, order.size() - off);
            if (launch_em_coded(ctx, D + off, cnt, em->b->m, coded_rows_max)) return 1;
        }
    } else if (shared) {
        for (size_t off = 0; off < (size_t)n_groups; off += max_units) {
            const int cnt = (int)std::min<size_t>(max_units, (size_t)n_groups - off);
            if (launch_em_sweep_groups(ctx, D, Dg + 2 * off, cnt, em->b->m, em->mode)) return 1;
        }
    } else {
        for (size_t off = 0; off < order.size(); off += max_units) {
            const int cnt = (int)std::min<size_t>(max_units, order.size() - off);
            if (launch_em_sweep(ctx, D + off, cnt, em->b->m, em->mode)) return 1;
        }
    }
    if (ev1) HIP_TRY(hipEventRecord(ev1, ctx->stream));
    for (size_t off = 0; off < order.size(); off += 65535) {
        const int cnt = (int)std::min<size_t>(65535, order.size() - off);
        if (launch_ssq_reduce(ctx, D + off, cnt, em->b->m, em->d_part2 + off * ssq_reduce_chunks())) return 1;
    }
    return 0;
}

int wgs_em_step_dev(wgs_em *em, double *ssq_dev)
{
    WGS_REQUIRE(em && ssq_dev, "null argument");
    wgs_ctx *ctx = em->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    em->last.clear();
    for (int j = 0; j < em->n_fits; ++j)
        if (em->active[j]) em->last.push_back(j);
    HIP_TRY(hipMemsetAsync(ssq_dev, 0, sizeof(double) * em->n_fits, ctx->stream));
    if (em->last.empty()) return 0;
    // h_descs / h_groups (pinned) stay untouched until the next step, which the caller only starts after
    // consuming this step's sums
    if (em_enqueue_sweep(em, em->last, em->h_descs, em->d_descs, em->h_groups, em->d_groups, ssq_dev, nullptr, em->ev0, em->ev1, 0)) return 1;
    for (int j : em->last) em->cur[j] ^= 1;   // the new frequencies are now current; 1-cur holds f_prev
    return 0;
}

int wgs_em_step(wgs_em *em, double *ssq_host)
{
    WGS_REQUIRE(em, "null argument");
    if (wgs_em_step_dev(em, em->d_ssq)) return 1;
    wgs_ctx *ctx = em->b->ctx;
    if (ssq_host) {
        HIP_TRY(hipMemcpyAsync(ssq_host, em->d_ssq, sizeof(double) * em->n_fits, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int wgs_em_rmse_chain(wgs_em *em, int32_t fit, float carry_in, float *carry_out)
{
    WGS_REQUIRE(em && carry_out, "null argument");
    WGS_REQUIRE(fit >= 0 && fit < em->n_fits, "fit index out of range");
    wgs_ctx *ctx = em->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (launch_rmse_chain(ctx, em_f(em, fit, em->cur[fit]), em_f(em, fit, em->cur[fit] ^ 1), em->b->m, carry_in, em->d_carry,
                          em->d_chain_work, reinterpret_cast<int *>(em->d_carry + 1)))
        return 1;
    float host[2];
    HIP_TRY(hipMemcpyAsync(host, em->d_carry, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *carry_out = host[0];
    memcpy(&em->last_chain_serial_blocks, &host[1], sizeof(int));
    return 0;
}

/* ---- emMAF.py:15-27 for every fit of the batch in ONE call ------------------------------------------
 * The host enqueues iteration t (sweep, sum reduction, [RCCL all-reduce], decision kernel, state readback)
 * BEFORE it reads the decisions of iteration t-1, so the GPU never waits for the host:
 *   - the decision kernel settles the clear cases on the device (EM_CONVERGED / EM_ACTIVE) and parks the
 *     fits whose float64 sum lies in the guard band (EM_UNDECIDED);
 *   - a sweep skips every fit that is not EM_ACTIVE, so a fit that converged at t-1 keeps the frequencies of
 *     update t-1 (emMAF.py:23-25 breaks after the update) and a parked fit keeps both vectors its exact
 *     chain needs;
 *   - the host, one iteration behind, resolves parked fits with the exact serial float32 chain (all of them
 *     in one batched launch; across SNP shards the float32 carries travel in rank order) and either
 *     finishes them or re-activates them -- such a fit simply runs its next sweep one iteration later.
 * Decisions use only all-reduced sums, so every rank takes the same path. */
static int em_fit_alloc(wgs_em *em)
{
    if (em->d_state) return 0;
    const size_t n = (size_t)em->n_fits;
    HIP_TRY(hipMalloc(&em->d_state, sizeof(int32_t) * n));
    HIP_TRY(hipMalloc(&em->d_ssq2, sizeof(double) * n));
    HIP_TRY(hipMemset(em->d_ssq2, 0, sizeof(double) * n));
    HIP_TRY(hipMalloc(&em->d_jobs, sizeof(ChainJob) * n));
    HIP_TRY(hipMalloc(&em->d_chain_out, sizeof(float) * 2 * n));
    // workspace of the exact chains for all fits at once (60 bytes per fit and block of 4096 SNPs): no allocation
    // inside the convergence loop
    HIP_TRY(hipMalloc(&em->d_chain_batch, rmse_chain_workspace_bytes(em->b->m) * n));
    em->chain_batch_jobs = n;
    HIP_TRY(hipHostMalloc(&em->h_jobs, sizeof(ChainJob) * n, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&em->h_chain_out, sizeof(float) * 2 * n, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&em->h_setstate, sizeof(int32_t) * n, hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMalloc(&em->d_descs2[i], sizeof(FitDesc) * n));
        HIP_TRY(hipHostMalloc(&em->h_descs2[i], sizeof(FitDesc) * n, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&em->d_groups2[i], sizeof(int32_t) * 2 * n));
        HIP_TRY(hipHostMalloc(&em->h_groups2[i], sizeof(int32_t) * 2 * n, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(&em->h_state[i], sizeof(int32_t) * n, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&em->ev_it[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&em->ev_sw0[i]));
        HIP_TRY(hipEventCreate(&em->ev_sw1[i]));
    }
    return 0;
}

/* Exact chains of `fits` (all at once): converged[i] = the reference's `diff < tole` for fits[i]. */
static int em_resolve_chains(wgs_em *em, const std::vector<int32_t> &fits, double tole, int64_t m_total, wgs_comm *comm,
                             std::vector<char> &converged)
{
    wgs_ctx *ctx = em->b->ctx;
    const int nj = (int)fits.size();
    converged.assign(nj, 0);
    if (nj == 0) return 0;
    int world = 1, rank = 0;
    if (comm) wgs_comm_rank(comm, &rank, &world);
    // The serial float32 chain crosses the SNP shards in rank order ON THE STREAM: rank r walks its blocks from the
    // running values it received and broadcasts the result (`world` broadcasts of nj float32, one readback at the end).
    for (int i = 0; i < nj; ++i) {
        const int j = fits[i];
        em->h_jobs[i] = ChainJob{em_f(em, j, em->cur[j]), em_f(em, j, em->cur[j] ^ 1), 0.0f};
    }
    HIP_TRY(hipMemcpyAsync(em->d_jobs, em->h_jobs, sizeof(ChainJob) * nj, hipMemcpyHostToDevice, ctx->stream));
    for (int r = 0; r < world; ++r) {
        if (r == rank) {
            if (r > 0 && launch_chain_set_carry(ctx, em->d_jobs, em->d_chain_out, nj)) return 1;
            if (launch_rmse_chain_batch(ctx, em->d_jobs, nj, em->b->m, em->d_chain_out, em->d_chain_batch,
                                        reinterpret_cast<int *>(em->d_chain_out + em->n_fits)))
                return 1;
        }
        if (world > 1 && wgs_comm_bcast_dev(comm, em->d_chain_out, (int64_t)sizeof(float) * nj, r)) return 1;
    }
    HIP_TRY(hipMemcpyAsync(em->h_chain_out, em->d_chain_out, sizeof(float) * nj, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));       // also: h_jobs has been consumed
    const float *carry = em->h_chain_out;
    ++em->fit_chain_batches;
    for (int i = 0; i < nj; ++i) {
        const float res = carry[i] / (float)m_total;         // emMAF_cy.pyx:32
        converged[i] = sqrt((double)res) < tole;             // emMAF_cy.pyx:33, emMAF.py:23
    }
    return 0;
}

int wgs_em_fit(wgs_em *em, int32_t max_iter, double tole, int64_t m_total, wgs_comm *comm, double guard_floor, int32_t *iters_out)
{
    WGS_REQUIRE(em && iters_out, "null argument");
    WGS_REQUIRE(m_total >= em->b->m, "m_total (%lld) is smaller than this shard (%lld SNPs)", (long long)m_total, (long long)em->b->m);
    wgs_ctx *ctx = em->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (em_fit_alloc(em)) return 1;
    const int n = em->n_fits;
    // the band of device.py: guard_band / decide_converged
    double lo = -1.0, hi = -INFINITY;                        // tole <= 0 or NaN: `diff < tole` never holds
    if (tole > 0) {
        const double thresh = tole * tole * (double)m_total;
        const double g = std::max(guard_floor, (double)m_total * 0x1p-24) + 1e-6;
        lo = g < 1.0 ? thresh * (1.0 - g) : -1.0;
        hi = thresh * (1.0 + g);
    }
    std::vector<char> fin(n, 0), skipped(n, 0);
    std::vector<int32_t> sweeps(n, 0), init(n), ran, parked, lists[2];
    for (int j = 0; j < n; ++j) {
        iters_out[j] = 0;
        fin[j] = !em->active[j];
        init[j] = em->active[j] ? EM_ACTIVE : EM_CONVERGED;
    }
    HIP_TRY(hipMemcpyAsync(em->d_state, init.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    em->fit_iterations = em->fit_chain_batches = 0;
    em->fit_sweep_ms = 0.0;
    const auto t_begin = std::chrono::steady_clock::now();
    bool launched_prev = false;
    for (int t = 1;; ++t) {
        const int slot = t & 1;
        // Who ran at t-1 is known now: its list minus the fits found finished or parked when the decisions
        // of t-2 were read (those sweeps returned at once).
        ran.clear();
        for (int j : lists[slot ^ 1]) {
            if (skipped[j]) continue;
            ++sweeps[j];
            em->cur[j] ^= 1;                                 // the new frequencies are current; 1-cur holds f_prev
            ran.push_back(j);
        }
        std::fill(skipped.begin(), skipped.end(), 0);
        // ---- enqueue iteration t (fits that turn out to have converged at t-1 return at once)
        std::vector<int32_t> &L = lists[slot];
        L.clear();
        for (int j = 0; j < n; ++j)
            if (!fin[j] && sweeps[j] < max_iter) L.push_back(j);
        if (!L.empty()) {
            if (em_enqueue_sweep(em, L, em->h_descs2[slot], em->d_descs2[slot], em->h_groups2[slot], em->d_groups2[slot], em->d_ssq2,
                                 em->d_state, em->ev_sw0[slot], em->ev_sw1[slot], max_iter - t + 1))
                return 1;
            // Fits that skipped this sweep have stale sums; the decision kernel ignores them, and they are stale
            // in the same way on every rank (all ranks take the same decisions).
            if (comm && wgs_comm_allreduce_f64_dev(comm, em->d_ssq2, n)) return 1;
            if (launch_em_decide(ctx, em->d_descs2[slot], (int)L.size(), lo, hi)) return 1;
            HIP_TRY(hipMemcpyAsync(em->h_state[slot], em->d_state, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipEventRecord(em->ev_it[slot], ctx->stream));
            ++em->fit_iterations;
        }
        // ---- read the decisions of iteration t-1 while the GPU works on iteration t
        if (launched_prev) {
            const int ps = slot ^ 1;
            HIP_TRY(hipEventSynchronize(em->ev_it[ps]));     // also: the pinned descriptors of t-1 have been consumed
            float sweep_ms = 0.0f;
            if (hipEventElapsedTime(&sweep_ms, em->ev_sw0[ps], em->ev_sw1[ps]) == hipSuccess) em->fit_sweep_ms += sweep_ms;
            parked.clear();
            for (int j : ran) {
                const int st = em->h_state[ps][j];
                if (st == EM_CONVERGED) {
                    fin[j] = 1;
                    skipped[j] = 1;                          // its sweep t (if enqueued) returned at once
                    iters_out[j] = sweeps[j];
                } else if (st == EM_UNDECIDED) {
                    parked.push_back(j);
                    skipped[j] = 1;
                } else if (sweeps[j] >= max_iter) {
                    fin[j] = 1;                              // exhausted: the reference prints nothing, iters stays 0
                }
            }
            if (!parked.empty()) {
                std::vector<char> conv;
                if (em_resolve_chains(em, parked, tole, m_total, comm, conv)) return 1;
                for (size_t i = 0; i < parked.size(); ++i) {
                    const int j = parked[i];
                    if (conv[i]) {
                        fin[j] = 1;
                        iters_out[j] = sweeps[j];
                    } else if (sweeps[j] >= max_iter) {
                        fin[j] = 1;
                    }
                    // stream-ordered behind iteration t (whose sweep must see the fit parked throughout)
                    em->h_setstate[j] = conv[i] ? EM_CONVERGED : EM_ACTIVE;
                    HIP_TRY(hipMemcpyAsync(em->d_state + j, em->h_setstate + j, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
                }
            }
        }
        launched_prev = !L.empty();
        if (!launched_prev) break;                           // nothing in flight: every fit finished or exhausted
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < n; ++j)
        if (iters_out[j] > 0) em->active[j] = 0;             // frozen, as wgs_em_set_active(j, 0) would
    em->fit_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}

/* Diagnostics of the last wgs_em_fit: iterations enqueued, batched exact-chain resolutions, wall seconds, and the
 * summed duration of its sweep kernels (HIP events on the context's stream around each iteration's sweep). */
int wgs_em_fit_stats(wgs_em *em, int32_t *iterations, int32_t *chain_batches, double *seconds, double *sweep_ms)
{
    WGS_REQUIRE(em, "null argument");
    if (iterations) *iterations = em->fit_iterations;
    if (chain_batches) *chain_batches = em->fit_chain_batches;
    if (seconds) *seconds = em->fit_seconds;
    if (sweep_ms) *sweep_ms = em->fit_sweep_ms;
    return 0;
}

int wgs_em_last_chain_serial_blocks(wgs_em *em) { return em ? em->last_chain_serial_blocks : -1; }

int wgs_em_last_sweep_ms(wgs_em *em, float *ms)
{
    WGS_REQUIRE(em && ms, "null argument");
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    HIP_TRY(hipEventSynchronize(em->ev1));
    HIP_TRY(hipEventElapsedTime(ms, em->ev0, em->ev1));
    return 0;
}

int wgs_em_set_active(wgs_em *em, int32_t fit, int active)
{
    WGS_REQUIRE(em && fit >= 0 && fit < em->n_fits, "fit index out of range");
    em->active[fit] = active ? 1 : 0;
    return 0;
}

int wgs_em_n_active(wgs_em *em)
{
    int c = 0;
    for (int j = 0; j < em->n_fits; ++j) c += em->active[j];
    return c;
}

int wgs_em_clamp(wgs_em *em, int32_t fit, float lo, float hi)
{
    WGS_REQUIRE(em && fit >= 0 && fit < em->n_fits, "fit index out of range");
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    return launch_clamp(em->b->ctx, em_f(em, fit, em->cur[fit]), em->b->m, lo, hi);
}

int wgs_em_get_f(wgs_em *em, int32_t fit, float *f_host)
{
    WGS_REQUIRE(em && f_host && fit >= 0 && fit < em->n_fits, "bad argument");
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    HIP_TRY(hipMemcpyAsync(f_host, em_f(em, fit, em->cur[fit]), sizeof(float) * em->b->m, hipMemcpyDeviceToHost, em->b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(em->b->ctx->stream));
    return 0;
}

int wgs_em_get_f_range(wgs_em *em, int32_t fit, int previous, int64_t row0, int64_t nrows, float *f_host)
{
    WGS_REQUIRE(em && f_host && fit >= 0 && fit < em->n_fits, "bad argument");
    WGS_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= em->b->m, "row range [%lld, %lld) outside 0..%lld", (long long)row0,
                (long long)(row0 + nrows), (long long)em->b->m);
    if (nrows == 0) return 0;
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    const float *src = em_f(em, fit, previous ? em->cur[fit] ^ 1 : em->cur[fit]) + row0;
    HIP_TRY(hipMemcpyAsync(f_host, src, sizeof(float) * nrows, hipMemcpyDeviceToHost, em->b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(em->b->ctx->stream));
    return 0;
}

int wgs_em_set_f(wgs_em *em, int32_t fit, const float *f_host)
{
    WGS_REQUIRE(em && f_host && fit >= 0 && fit < em->n_fits, "bad argument");
    HIP_TRY(hipSetDevice(em->b->ctx->device));
    HIP_TRY(hipMemcpyAsync(em_f(em, fit, em->cur[fit]), f_host, sizeof(float) * em->b->m, hipMemcpyHostToDevice, em->b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(em->b->ctx->stream));
    return 0;
}

const float *wgs_em_f_dev(wgs_em *em, int32_t fit)
{
    if (!em || fit < 0 || fit >= em->n_fits) return nullptr;
    return em_f(em, fit, em->cur[fit]);
}

/* ------------------------------------------------------------------ allele-frequency sets */

int wgs_afset_create(wgs_ctx *ctx, int64_t m, int32_t K, wgs_afset **out)
{
    WGS_REQUIRE(ctx && out && m > 0 && K > 0, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    wgs_afset *a = new wgs_afset();
    a->ctx = ctx;
    a->m = m;
    a->K = K;
    if (hipMalloc(&a->buf, sizeof(float) * (size_t)m * K) != hipSuccess) {
        wgs_set_error("hipMalloc for allele frequencies failed");
        delete a;
        return 1;
    }
    *out = a;
    return 0;
}

void wgs_afset_destroy(wgs_afset *a)
{
    if (!a) return;
    (void)hipSetDevice(a->ctx->device);
    if (a->buf) (void)hipFree(a->buf);
    delete a;
}

int wgs_afset_upload(wgs_afset *a, const float *A_mK)
{
    WGS_REQUIRE(a && A_mK, "null argument");
    HIP_TRY(hipSetDevice(a->ctx->device));
    float *tmp = nullptr;
    const size_t bytes = sizeof(float) * (size_t)a->m * a->K;
    HIP_TRY(hipMalloc(&tmp, bytes));
    int rc = 0;
    if (hipMemcpyAsync(tmp, A_mK, bytes, hipMemcpyHostToDevice, a->ctx->stream) != hipSuccess) rc = 1;
    if (!rc) rc = launch_transpose_mK_to_Km(a->ctx, tmp, a->buf, a->m, a->K);
    if (hipStreamSynchronize(a->ctx->stream) != hipSuccess) rc = 1;
    (void)hipFree(tmp);
    if (rc) wgs_set_error("allele-frequency upload failed");
    return rc;
}

int wgs_afset_download(wgs_afset *a, float *A_mK)
{
    WGS_REQUIRE(a && A_mK, "null argument");
    HIP_TRY(hipSetDevice(a->ctx->device));
    float *tmp = nullptr;
    const size_t bytes = sizeof(float) * (size_t)a->m * a->K;
    HIP_TRY(hipMalloc(&tmp, bytes));
    int rc = launch_transpose_Km_to_mK(a->ctx, a->buf, tmp, a->m, a->K);
    if (!rc && hipMemcpyAsync(A_mK, tmp, bytes, hipMemcpyDeviceToHost, a->ctx->stream) != hipSuccess) rc = 1;
    if (hipStreamSynchronize(a->ctx->stream) != hipSuccess) rc = 1;
    (void)hipFree(tmp);
    if (rc) wgs_set_error("allele-frequency download failed");
    return rc;
}

int wgs_afset_set_column_from_em(wgs_afset *a, int32_t col, wgs_em *em, int32_t fit)
{
    WGS_REQUIRE(a && em && col >= 0 && col < a->K && fit >= 0 && fit < em->n_fits, "bad argument");
    WGS_REQUIRE(a->m == em->b->m, "SNP counts differ");
    HIP_TRY(hipSetDevice(a->ctx->device));
    HIP_TRY(hipMemcpyAsync(a->buf + (size_t)col * a->m, wgs_em_f_dev(em, fit), sizeof(float) * a->m, hipMemcpyDeviceToDevice, a->ctx->stream));
    return 0;
}

const float *wgs_afset_col_dev(wgs_afset *a, int32_t col)
{
    if (!a || col < 0 || col >= a->K) return nullptr;
    return a->buf + (size_t)col * a->m;
}

/* ------------------------------------------------------------------ assignment / scoring */

int wgs_assign_last_ms(wgs_ctx *ctx, float *ms)
{
    WGS_REQUIRE(ctx && ms, "null argument");
    *ms = ctx->last_assign_ms;
    return 0;
}

struct wgs_score {
    wgs_beagle *b = nullptr;
    wgs_afset *a = nullptr;
    int32_t K = 0, row_lo = 0, row_hi = 0, nblocks = 0, P = 0;
    int64_t n = 0, cells = 0;
    bool per_ind = false, have_prefix = false;
    const float **d_acol = nullptr, **d_colptr = nullptr;
    ScoreSlab *d_slabs[2] = {nullptr, nullptr};      // [0] table of the sweep, [1] table of the chain kernel
    int n_slabs[2] = {0, 0}, total_pg[2] = {0, 0};
    double *d_S = nullptr, *d_out = nullptr, *d_start = nullptr, *d_run = nullptr, *d_chunks = nullptr;     // d_chunks: [ceil(nblocks/2)][cells]
    uint32_t *d_cand = nullptr;
    float *d_carry = nullptr, *d_parts = nullptr;
    int32_t *d_nserial = nullptr;
    int32_t last_serial_blocks = 0;
    CodedSlabHost *d_coded = nullptr;     // slab table of the sweep through the class codes (shared columns)
    int n_coded = 0, coded_quads = 0;
    int64_t coded_generation = -1;        // wgs_codes::generation of the build d_coded was made from
    int last_path = 0;                    // 1: the last wgs_score_sums went through the class codes
};

void wgs_score_destroy(wgs_score *sc)
{
    if (!sc) return;
    (void)hipSetDevice(sc->b->ctx->device);
    (void)hipStreamSynchronize(sc->b->ctx->stream);
    void *bufs[] = {sc->d_acol, sc->d_colptr, sc->d_slabs[0], sc->d_slabs[1] == sc->d_slabs[0] ? nullptr : sc->d_slabs[1], sc->d_S,
                    sc->d_out, sc->d_start, sc->d_run, sc->d_coded, sc->d_cand, sc->d_carry, sc->d_parts, sc->d_nserial, sc->d_chunks};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    delete sc;
}

/* Slab table for NP pairs per wave, restricted to the individuals [row_lo, row_hi): the members of a slab
 * are in file order, so the scored ones are a contiguous column range. */
static int build_slab_table(wgs_score *sc, int np, int which)
{
    std::vector<ScoreSlab> tab;
    int pg = 0;
    for (int g = 0; g < sc->b->n_groups; ++g) {
        const Slab &s = sc->b->slabs[g];
        if (s.ncols == 0) continue;
        const int lo = (int)(std::lower_bound(s.members.begin(), s.members.end(), sc->row_lo) - s.members.begin());
        const int hi = (int)(std::lower_bound(s.members.begin(), s.members.end(), sc->row_hi) - s.members.begin());
        if (hi <= lo) continue;
        ScoreSlab e;
        e.slab = s.base;
        e.members = s.d_members;
        e.npairs = s.npairs;
        e.ncols = s.ncols;
        e.pair0 = lo / 2;
        e.npg = ((hi - 1) / 2 - e.pair0 + 1 + np - 1) / np;
        e.pg0 = pg;
        e.col_lo = lo;
        e.col_hi = hi;
        pg += e.npg;
        tab.push_back(e);
    }
    sc->n_slabs[which] = (int)tab.size();
    sc->total_pg[which] = pg;
    if (tab.empty()) return 0;
    HIP_TRY(hipMalloc(&sc->d_slabs[which], sizeof(ScoreSlab) * tab.size()));
    HIP_TRY(hipMemcpy(sc->d_slabs[which], tab.data(), sizeof(ScoreSlab) * tab.size(), hipMemcpyHostToDevice));
    return 0;
}

int wgs_score_create(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t row_lo, int32_t row_hi, wgs_score **out)
{
    WGS_REQUIRE(b && a && out, "null argument");
    WGS_REQUIRE(a->m == b->m, "allele frequencies cover %lld SNPs, the Beagle shard %lld", (long long)a->m, (long long)b->m);
    WGS_REQUIRE(row_lo >= 0 && row_lo <= row_hi && row_hi <= b->n, "individual range [%d, %d) outside 0..%lld", row_lo, row_hi,
                (long long)b->n);
    wgs_ctx *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    wgs_score *sc = new wgs_score();
    auto guard = on_failure([&] { wgs_score_destroy(sc); });
    sc->b = b;
    sc->a = a;
    sc->K = a->K;
    sc->n = b->n;
    sc->cells = b->n * (int64_t)a->K;
    sc->row_lo = row_lo;
    sc->row_hi = row_hi;
    sc->per_ind = colptr != nullptr;
    sc->nblocks = (int32_t)((wgs_ntiles(b->m) + WGS_BLOCK_TILES - 1) / WGS_BLOCK_TILES);
    std::vector<const float *> acol(a->K);
    for (int k = 0; k < a->K; ++k) acol[k] = a->buf + (size_t)k * a->m;
    HIP_TRY(hipMalloc(&sc->d_acol, sizeof(float *) * a->K));
    HIP_TRY(hipMemcpy(sc->d_acol, acol.data(), sizeof(float *) * a->K, hipMemcpyHostToDevice));
    if (colptr) {
        HIP_TRY(hipMalloc(&sc->d_colptr, sizeof(float *) * sc->cells));
        HIP_TRY(hipMemcpy(sc->d_colptr, colptr, sizeof(float *) * sc->cells, hipMemcpyHostToDevice));
    }
    const int np_sweep = score_pairs_per_wave(a->K, sc->per_ind), np_chain = chain_pairs_per_wave(a->K, sc->per_ind);
    if (build_slab_table(sc, np_sweep, 0)) return 1;
    if (np_chain == np_sweep) {
        sc->d_slabs[1] = sc->d_slabs[0];
        sc->n_slabs[1] = sc->n_slabs[0];
        sc->total_pg[1] = sc->total_pg[0];
    } else if (build_slab_table(sc, np_chain, 1)) {
        return 1;
    }
    if (hipMalloc(&sc->d_S, sizeof(double) * (size_t)sc->nblocks * sc->cells) != hipSuccess) {
        wgs_set_error("hipMalloc of %zu bytes for the block sums failed", sizeof(double) * (size_t)sc->nblocks * sc->cells);
        return 1;
    }
    HIP_TRY(hipMalloc(&sc->d_out, sizeof(double) * sc->cells));
    guard.dismiss();
    *out = sc;
    return 0;
}

static ScoreArgs score_args(const wgs_score *sc, int which)
{
    ScoreArgs A;
    A.slabs = sc->d_slabs[which];
    A.n_slabs = sc->n_slabs[which];
    A.total_pg = sc->total_pg[which];
    A.colptr = sc->d_colptr;
    A.acol = sc->d_acol;
    A.m = sc->b->m;
    A.site0 = sc->b->site0;
    A.cells = sc->cells;
    A.K = sc->K;
    A.P = 1;
    A.period = 1;
    A.nblocks = sc->nblocks;
    A.S = sc->d_S;
    A.start = nullptr;
    A.cand = nullptr;
    return A;
}

/* All n x K sums of glassy.py:31-42 / 92-105 for the scored individuals: out[i*K + k] (host, overwritten;
 * rows outside the scored range are 0) = the float64 sum over this shard's SNPs of the float32 per-site
 * values, formed in a fixed order (per lane over the tiles of a block, a fixed shuffle tree over lanes,
 * blocks in order): the same bits on every run. */
int wgs_score_sums(wgs_score *sc, int mode, double *out)
{
    WGS_REQUIRE(sc && out, "null argument");
    WGS_REQUIRE(mode == WGS_MODE_EXACT || mode == WGS_MODE_FAST, "unknown mode %d", mode);
    wgs_ctx *ctx = sc->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    // shared columns + a codable matrix: the sweep through the class codes (same S, bit for bit)
    wgs_codes *codes = sc->per_ind ? nullptr : wgs_beagle_codes(sc->b);
    if (codes && score_coded_lds_bytes(codes->rows_batch, score_kb(sc->K), codes->score_batch) > 64 * 1024) codes = nullptr;
    if (codes && sc->coded_generation != codes->generation) {     // (keyed on the build, not on the object's address: a rebuilt wgs_codes may reuse it)
        std::vector<CodedSlabHost> tab;
        int quad0 = 0;
        for (int g = 0; g < sc->b->n_groups; ++g) {
            const Slab &s = sc->b->slabs[g];
            if (s.ncols == 0) continue;
            const int lo = (int)(std::lower_bound(s.members.begin(), s.members.end(), sc->row_lo) - s.members.begin());
            const int hi = (int)(std::lower_bound(s.members.begin(), s.members.end(), sc->row_hi) - s.members.begin());
            if (hi <= lo) continue;
            CodedSlabHost e;
            e.codes = codes->slabs[g].codes;
            e.members = s.d_members;
            e.slab = s.base;
            e.npairs = s.npairs;
            e.nquads = codes->slabs[g].nquads;
            e.ncols = s.ncols;
            e.quad0 = quad0;
            e.col_lo = lo;
            e.col_hi = hi;
            quad0 += e.nquads;
            tab.push_back(e);
        }
        if (sc->d_coded) HIP_TRY(hipFree(sc->d_coded));
        sc->d_coded = nullptr;
        sc->n_coded = (int)tab.size();
        sc->coded_quads = quad0;
        if (!tab.empty()) {
            HIP_TRY(hipMalloc(&sc->d_coded, sizeof(CodedSlabHost) * tab.size()));
            HIP_TRY(hipMemcpy(sc->d_coded, tab.data(), sizeof(CodedSlabHost) * tab.size(), hipMemcpyHostToDevice));
        }
        sc->coded_generation = codes->generation;
    }
    HIP_TRY(hipMemsetAsync(sc->d_S, 0, sizeof(double) * (size_t)sc->nblocks * sc->cells, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    sc->last_path = codes ? 1 : 0;
    if (codes) {
        if (launch_score_coded(ctx, codes, sc->d_coded, sc->n_coded, sc->coded_quads, sc->d_acol, sc->b->m, sc->cells, sc->K, sc->nblocks,
                               sc->d_S, mode))
            return 1;
    } else if (launch_score_sweep(ctx, score_args(sc, 0), mode)) {
        return 1;
    }
    if (!sc->d_chunks && hipMalloc(&sc->d_chunks, sizeof(double) * (size_t)((sc->nblocks + 1) / 2) * sc->cells) != hipSuccess) {
        wgs_set_error("hipMalloc of the chunk sums failed");
        return 1;
    }
    if (launch_block_prefix(ctx, sc->d_S, sc->nblocks, sc->cells, sc->d_out, 1, sc->d_chunks)) return 1;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, sc->d_out, sizeof(double) * sc->cells, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    (void)hipEventElapsedTime(&ctx->last_assign_ms, ctx->ev0, ctx->ev1);
    sc->have_prefix = (mode == WGS_MODE_EXACT);
    return 0;
}

/* The same sums continued from the SNP shards before this one: out[i*K + k] = (((carry_in + C0) + C1) + ...) over this
 * shard's 8192-site chunk sums C (kept by wgs_score_sums), i.e. np.sum(vec, dtype=float) of glassy.py:38 carried on in
 * NumPy's own order when every shard starts at a multiple of 8192 sites (comm.shard_range sees to that).  carry_in (host,
 * n*K doubles, NULL = zeros) is the value returned for the preceding shard; needs wgs_score_sums first. */
int wgs_score_total_from(wgs_score *sc, const double *carry_in, double *out)
{
    WGS_REQUIRE(sc && out, "null argument");
    WGS_REQUIRE(sc->d_chunks, "wgs_score_total_from needs wgs_score_sums first");
    wgs_ctx *ctx = sc->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!sc->d_start) HIP_TRY(hipMalloc(&sc->d_start, sizeof(double) * sc->cells));
    if (carry_in) HIP_TRY(hipMemcpyAsync(sc->d_start, carry_in, sizeof(double) * sc->cells, hipMemcpyHostToDevice, ctx->stream));
    if (launch_chunk_total(ctx, sc->d_chunks, (sc->nblocks + 1) / 2, sc->cells, carry_in ? sc->d_start : nullptr, sc->d_out)) return 1;
    HIP_TRY(hipMemcpyAsync(out, sc->d_out, sizeof(double) * sc->cells, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* The n x K totals over ALL SNP shards in NumPy's order, one call: np.sum's running float64 total is handed from shard to
 * shard in SNP order ON THE STREAM -- rank r continues it over its chunk sums (chunk_total_kernel) and broadcasts the
 * result, rank r + 1 picks it up as its carry -- `world` broadcasts of n*K float64 enqueued back to back, ONE readback.
 * totals_out (host, n*K) receives the totals on every rank; before_out (host, n*K, may be NULL) the total over the shards
 * BEFORE this one (what wgs_score_chains_prepare wants as `start`).  comm == NULL or one rank: the local sums.
 * Needs wgs_score_sums first. */
int wgs_score_totals_all(wgs_score *sc, wgs_comm *comm, double *totals_out, double *before_out)
{
    WGS_REQUIRE(sc && totals_out, "null argument");
    WGS_REQUIRE(sc->d_chunks, "wgs_score_totals_all needs wgs_score_sums first");
    wgs_ctx *ctx = sc->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    int world = 1, rank = 0;
    if (comm) wgs_comm_rank(comm, &rank, &world);
    const size_t bytes = sizeof(double) * sc->cells;
    if (!sc->d_start) HIP_TRY(hipMalloc(&sc->d_start, bytes));
    if (!sc->d_run) HIP_TRY(hipMalloc(&sc->d_run, bytes));
    HIP_TRY(hipMemsetAsync(sc->d_start, 0, bytes, ctx->stream));
    for (int r = 0; r < world; ++r) {
        if (r == rank) {
            if (r > 0) HIP_TRY(hipMemcpyAsync(sc->d_start, sc->d_run, bytes, hipMemcpyDeviceToDevice, ctx->stream));   // what precedes this shard
            if (launch_chunk_total(ctx, sc->d_chunks, (sc->nblocks + 1) / 2, sc->cells, r > 0 ? sc->d_start : nullptr, sc->d_run)) return 1;
        }
        if (world > 1 && wgs_comm_bcast_dev(comm, sc->d_run, (int64_t)bytes, r)) return 1;
    }
    HIP_TRY(hipMemcpyAsync(totals_out, sc->d_run, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (before_out) HIP_TRY(hipMemcpyAsync(before_out, sc->d_start, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* Block functions of the exact partition chains (utils.py:147-149) for P partitions; needs the block
 * sums of wgs_score_sums(WGS_MODE_EXACT).  start (host, n*K doubles, may be NULL) = the float64 sums over
 * the SNP shards that precede this one (its partitions are predicted to hold equal shares).  rc 2 when P is
 * too large for the block-parallel kernel (use wgs_assign_parts_exact's literal chains then). */
int wgs_score_chains_prepare(wgs_score *sc, int32_t P, const double *start)
{
    WGS_REQUIRE(sc, "null argument");
    WGS_REQUIRE(P >= 1, "partition count must be >= 1");
    WGS_REQUIRE(sc->have_prefix, "wgs_score_chains_prepare needs wgs_score_sums(WGS_MODE_EXACT) first");
    WGS_REQUIRE(chain_cand_lds_bytes(sc->K, P, sc->per_ind) <= 64 * 1024 && (int64_t)sc->cells * P < (1ll << 31),
                "too many partitions (%d) for the block-parallel chains", P);
    wgs_ctx *ctx = sc->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t chains = (size_t)sc->cells * P;
    if (sc->P != P) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (void *p : {(void *)sc->d_cand, (void *)sc->d_carry, (void *)sc->d_parts})
            if (p) (void)hipFree(p);
        sc->d_cand = nullptr;
        sc->d_carry = sc->d_parts = nullptr;
        sc->P = 0;
        if (hipMalloc(&sc->d_cand, sizeof(uint32_t) * chains * sc->nblocks) != hipSuccess) {
            wgs_set_error("hipMalloc of %zu bytes for the partition-chain block functions failed", sizeof(uint32_t) * chains * sc->nblocks);
            return 1;
        }
        HIP_TRY(hipMalloc(&sc->d_carry, sizeof(float) * chains));
        HIP_TRY(hipMalloc(&sc->d_parts, sizeof(float) * chains));
        if (!sc->d_nserial) HIP_TRY(hipMalloc(&sc->d_nserial, sizeof(int32_t)));
        if (!sc->d_start) HIP_TRY(hipMalloc(&sc->d_start, sizeof(double) * sc->cells));
        sc->P = P;
    }
    HIP_TRY(hipMemsetAsync(sc->d_cand, 0, sizeof(uint32_t) * chains * sc->nblocks, ctx->stream));
    if (start) HIP_TRY(hipMemcpyAsync(sc->d_start, start, sizeof(double) * sc->cells, hipMemcpyHostToDevice, ctx->stream));
    ScoreArgs A = score_args(sc, 1);
    A.P = P;
    int g = 64, r = P;                       // gcd(64, P)
    while (r) {
        const int t = g % r;
        g = r;
        r = t;
    }
    A.period = P / g;
    A.start = start ? sc->d_start : nullptr;
    A.cand = sc->d_cand;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    if (launch_chain_cand(ctx, A)) return 1;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));       // `start` (host) has been consumed
    (void)hipEventElapsedTime(&ctx->last_assign_ms, ctx->ev0, ctx->ev1);
    return 0;
}

/* Walk the chains of this shard: carry_in (host float32 [n*P*K], NULL = zeros) is the running value after
 * the preceding shards, parts_out (host float32 [n*P*K], index (i*P + p)*K + k) the value after this one;
 * rows of individuals outside the scored range are 0. */
static int chains_walk_enqueue(wgs_score *sc, bool with_carry)
{
    wgs_ctx *ctx = sc->b->ctx;
    const size_t chains = (size_t)sc->cells * sc->P;
    HIP_TRY(hipMemsetAsync(sc->d_parts, 0, sizeof(float) * chains, ctx->stream));
    HIP_TRY(hipMemsetAsync(sc->d_nserial, 0, sizeof(int32_t), ctx->stream));
    WalkArgs W;
    W.cand = sc->d_cand;
    W.carry = with_carry ? sc->d_carry : nullptr;
    W.parts = sc->d_parts;
    W.group_of = sc->b->d_group_of;
    W.col_of = sc->b->d_col_of;
    W.npairs = sc->b->d_npairs;
    W.base = sc->b->d_base;
    W.colptr = sc->d_colptr;
    W.acol = sc->d_acol;
    W.m = sc->b->m;
    W.site0 = sc->b->site0;
    W.n = (int32_t)sc->n;
    W.K = sc->K;
    W.P = sc->P;
    W.nblocks = sc->nblocks;
    W.row_lo = sc->row_lo;
    W.row_hi = sc->row_hi;
    W.n_serial = sc->d_nserial;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    if (launch_chain_walk(ctx, W)) return 1;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    return 0;
}

int wgs_score_chains_walk(wgs_score *sc, const float *carry_in, float *parts_out)
{
    WGS_REQUIRE(sc && parts_out, "null argument");
    WGS_REQUIRE(sc->P >= 1 && sc->d_cand, "wgs_score_chains_walk needs wgs_score_chains_prepare first");
    wgs_ctx *ctx = sc->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t chains = (size_t)sc->cells * sc->P;
    if (carry_in) HIP_TRY(hipMemcpyAsync(sc->d_carry, carry_in, sizeof(float) * chains, hipMemcpyHostToDevice, ctx->stream));
    if (chains_walk_enqueue(sc, carry_in != nullptr)) return 1;
    HIP_TRY(hipMemcpyAsync(parts_out, sc->d_parts, sizeof(float) * chains, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&sc->last_serial_blocks, sc->d_nserial, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    (void)hipEventElapsedTime(&ctx->last_assign_ms, ctx->ev0, ctx->ev1);
    return 0;
}

/* The chains of ALL SNP shards, one call: rank 0 walks its blocks from zero, broadcasts its float32 values, rank 1 walks on
 * from them, ... -- `world` broadcasts of n*P*K float32 on the stream, ONE readback; parts_out (host) receives the
 * values after the last shard on every rank.  Every rank has prepared its block functions before (in parallel). */
int wgs_score_chains_walk_all(wgs_score *sc, wgs_comm *comm, float *parts_out)
{
    WGS_REQUIRE(sc && parts_out, "null argument");
    WGS_REQUIRE(sc->P >= 1 && sc->d_cand, "wgs_score_chains_walk_all needs wgs_score_chains_prepare first");
    wgs_ctx *ctx = sc->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    int world = 1, rank = 0;
    if (comm) wgs_comm_rank(comm, &rank, &world);
    const size_t bytes = sizeof(float) * (size_t)sc->cells * sc->P;
    for (int r = 0; r < world; ++r) {
        if (r == rank) {
            if (r > 0) HIP_TRY(hipMemcpyAsync(sc->d_carry, sc->d_parts, bytes, hipMemcpyDeviceToDevice, ctx->stream));
            if (chains_walk_enqueue(sc, r > 0)) return 1;
            HIP_TRY(hipMemcpyAsync(&sc->last_serial_blocks, sc->d_nserial, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        }
        if (world > 1 && wgs_comm_bcast_dev(comm, sc->d_parts, (int64_t)bytes, r)) return 1;
    }
    HIP_TRY(hipMemcpyAsync(parts_out, sc->d_parts, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    (void)hipEventElapsedTime(&ctx->last_assign_ms, ctx->ev0, ctx->ev1);
    return 0;
}

/* Test hook: blocks that took the literal serial loop in the last wgs_score_chains_walk, and the number of
 * (chain, block) pairs walked. */
int wgs_score_last_serial_blocks(wgs_score *sc, int64_t *total_blocks)
{
    if (!sc) return -1;
    if (total_blocks) *total_blocks = (int64_t)(sc->row_hi - sc->row_lo) * sc->K * sc->P * sc->nblocks;
    return sc->last_serial_blocks;
}

int wgs_assign(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int mode, double *out)
{
    WGS_REQUIRE(b && a && out, "null argument");
    WGS_REQUIRE(a->m == b->m, "allele frequencies cover %lld SNPs, the Beagle shard %lld", (long long)a->m, (long long)b->m);
    wgs_ctx *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t cells = (size_t)b->n * a->K;
    ctx->last_assign_ms = 0.0f;
    std::vector<double> h(cells);
    // one launch over all population slabs, reproducible sums (wgs_score_sums)
    wgs_score *sc = nullptr;
    int rc = wgs_score_create(b, a, colptr, 0, (int32_t)b->n, &sc);
    if (!rc) rc = wgs_score_sums(sc, mode, h.data());
    wgs_score_destroy(sc);
    if (rc) return rc;
    for (size_t c = 0; c < cells; ++c) out[c] += h[c];
    return 0;
}

/* Cross-check only (tests, tools/check_fast_mode.py): FLOAT64 partition sums (labels = global site index % P) from the
 * round-1 kernel that maps lanes to pairs of individuals and combines tile ranges with float64 atomics -- within
 * ~1e-5 of the reference's serial float32 partition sums and not reproducible run to run.  The product path is
 * wgs_assign_parts_exact / wgs_score_chains_*.  out [n*K] and parts [n*P*K] are accumulated into. */
int wgs_debug_assign_parts_f64(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t P, int mode, double *out, double *parts)
{
    WGS_REQUIRE(b && a && out && parts, "null argument");
    WGS_REQUIRE(a->m == b->m, "allele frequencies cover %lld SNPs, the Beagle shard %lld", (long long)a->m, (long long)b->m);
    WGS_REQUIRE(P >= 1, "partition count must be >= 1");
    wgs_ctx *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const int K = a->K;
    const int64_t n = b->n;
    const size_t cells = (size_t)n * P * K;
    ctx->last_assign_ms = 0.0f;
    std::vector<double> h(cells);
    // one grow-only workspace: [cells doubles | K shared pointers | n*K per-individual pointers]
    const size_t off_acol = (sizeof(double) * cells + 255) & ~(size_t)255;
    const size_t off_colptr = (off_acol + sizeof(float *) * K + 255) & ~(size_t)255;
    const size_t total = off_colptr + (colptr ? sizeof(float *) * n * K : 0);
    void *ws = nullptr;
    if (wgs_ctx_workspace(ctx, total, &ws)) return 1;
    double *d_out = reinterpret_cast<double *>(ws);
    const float **d_acol = reinterpret_cast<const float **>(reinterpret_cast<char *>(ws) + off_acol);
    const float **d_colptr = colptr ? reinterpret_cast<const float **>(reinterpret_cast<char *>(ws) + off_colptr) : nullptr;
    std::vector<const float *> acol(K);
    for (int k = 0; k < K; ++k) acol[k] = a->buf + (size_t)k * a->m;
    HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(double) * cells, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_acol, acol.data(), sizeof(float *) * K, hipMemcpyHostToDevice, ctx->stream));
    if (colptr) HIP_TRY(hipMemcpyAsync(d_colptr, colptr, sizeof(float *) * n * K, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));      // acol (a local vector) has been consumed
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    for (int g = 0; g < b->n_groups; ++g) {
        const Slab &s = b->slabs[g];
        if (s.ncols == 0) continue;
        AssignArgs args;
        args.slab = s.base;
        args.members = s.d_members;
        args.colptr = d_colptr;
        args.acol = d_acol;
        args.out = d_out;
        args.m = b->m;
        args.site0 = b->site0;
        args.npairs = s.npairs;
        args.ncols = s.ncols;
        args.K = K;
        args.P = P;
        args.tiles_per_wave = 0;
        if (launch_assign(ctx, args, mode)) return 1;
    }
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipMemcpyAsync(h.data(), d_out, sizeof(double) * cells, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    (void)hipEventElapsedTime(&ctx->last_assign_ms, ctx->ev0, ctx->ev1);
    for (size_t c = 0; c < cells; ++c) parts[c] += h[c];
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < K; ++k) {
            double t = 0.0;
            for (int p = 0; p < P; ++p) t += h[((size_t)i * P + p) * K + k];
            out[(size_t)i * K + k] += t;
        }
    return 0;
}

/* ---- glassy.loo -- glassy.py:47-112 -- in one call ---------------------------------------------------
 * For every individual i (file order): re-fit its population without it (emMAF.py:15-27 via wgs_em_fit, all
 * individuals of a batch at once), clamp with n_pop - 1 (glassy.py:80-85), OVERWRITE the population's column
 * (glassy.py:87-89: never restored, so every other column is the re-fit of the most recent earlier individual
 * of that population), score i against all K columns (float64 sums of the float32 per-site values,
 * glassy.py:92-105) and, if asked, accumulate the serial float32 partition sums (utils.py:147-149).
 *   b       the matrix the frequencies are estimated from (population slabs = columns of `a`);
 *   scored  the matrix that is scored (NULL = b; the downsampled matrix of --loo_downsampled_beagle);
 *   a       in: the full-population estimates; out: each population's LAST re-fit (glassy.py:89);
 *   batch   re-fits per EM batch, 0 = what fits the free device memory (agreed across ranks);
 *   ll_out  host float64 [n*K] (overwritten); parts_out host float32 [n*P*K] or NULL; iters_out [n]. */
static double g_loo_stats[7];      // of the last wgs_loo of this process: see wgs_loo_stats

static double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

/* Phases of the last wgs_loo: stats[0..5] = seconds in the EM re-fits (wgs_em_fit incl. its exact chains), in the
 * scoring sweeps (+ their cross-rank totals), in the exact partition chains; EM sweep kernel ms; EM batches; chain
 * resolutions of the re-fits; EM iterations enqueued (= all-reduces of the convergence sums across SNP shards). */
int wgs_loo_stats(double *stats)
{
    WGS_REQUIRE(stats, "null argument");
    for (int i = 0; i < 7; ++i) stats[i] = g_loo_stats[i];
    return 0;
}

int wgs_loo(wgs_beagle *b, wgs_beagle *scored, wgs_afset *a, int32_t max_iter, double tole, int64_t m_total, wgs_comm *comm,
            int32_t P, int32_t batch, int em_mode, int score_mode, double *ll_out, float *parts_out, int32_t *iters_out)
{
    for (double &x : g_loo_stats) x = 0.0;
    WGS_REQUIRE(b && a && ll_out && iters_out, "null argument");
    if (!scored) scored = b;
    WGS_REQUIRE(scored->n == b->n && scored->m == b->m && scored->n_groups == b->n_groups && scored->group_of == b->group_of,
                "the scored matrix must have the shape and population slabs of the fitted one");
    WGS_REQUIRE(a->m == b->m && a->K == b->n_groups, "allele frequencies (%lld x %d) do not match the population slabs (%lld x %d)",
                (long long)a->m, a->K, (long long)b->m, b->n_groups);
    WGS_REQUIRE(P >= 1, "partition count must be >= 1");
    wgs_ctx *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const int64_t n = b->n;
    const int K = a->K;
    const size_t cells = (size_t)n * K;
    int world = 1, rank = 0;
    if (comm) wgs_comm_rank(comm, &rank, &world);
    if (batch <= 0) {      // 2 float32 vectors + per-tile partial sums per fit: ~8.2 bytes per SNP and fit
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const double per_fit = (double)b->m * 8.2 + 4096.0;
        batch = (int32_t)std::max<double>(1.0, std::min<double>((double)n, 0.8 * (double)free_b / per_fit));
    }
    batch = (int32_t)std::min<int64_t>(n, batch);
    if (world > 1) {       // every rank must run the same batches: the minimum over ranks
        std::vector<double> slots(world, 0.0);
        slots[rank] = (double)batch;
        if (wgs_comm_allreduce_f64(comm, slots.data(), world)) return 1;
        batch = (int32_t)*std::min_element(slots.begin(), slots.end());
    }
    std::vector<int32_t> counts(K, 0);
    for (int64_t i = 0; i < n; ++i) ++counts[b->group_of[i]];
    std::fill(ll_out, ll_out + cells, 0.0);
    if (parts_out) std::fill(parts_out, parts_out + cells * P, 0.0f);
    std::vector<const float *> colptr(cells), cur(K);
    std::vector<double> sums(cells), start(cells);
    std::vector<float> parts;
    for (int64_t i0 = 0; i0 < n; i0 += batch) {
        const int64_t i1 = std::min<int64_t>(n, i0 + batch);
        const int nb = (int)(i1 - i0);
        std::vector<int32_t> grp(nb), skip(nb);
        for (int x = 0; x < nb; ++x) grp[x] = b->group_of[i0 + x], skip[x] = (int32_t)(i0 + x);
        wgs_em *em = nullptr;
        wgs_score *sc = nullptr;
        auto guard = on_failure([&] { wgs_score_destroy(sc); wgs_em_destroy(em); });
        double t_phase = wall_s();
        int rc = wgs_em_create(b, nb, grp.data(), skip.data(), em_mode, &em);
        if (rc) return rc;
        if ((rc = wgs_em_fit(em, max_iter, tole, m_total, comm, 0.0, iters_out + i0))) return rc;
        {
            int32_t it = 0, cb = 0;
            double sec = 0.0, sweep_ms = 0.0;
            wgs_em_fit_stats(em, &it, &cb, &sec, &sweep_ms);
            g_loo_stats[0] += wall_s() - t_phase;
            g_loo_stats[3] += sweep_ms;
            g_loo_stats[4] += 1.0;
            g_loo_stats[5] += cb;
            g_loo_stats[6] += it;
        }
        t_phase = wall_s();
        for (int x = 0; x < nb; ++x) {
            const int npop = counts[grp[x]] - 1;
            const double lo = 1.0 / (2.0 * (npop + 1));
            if ((rc = wgs_em_clamp(em, x, (float)lo, (float)(1.0 - lo)))) return rc;
        }
        // glassy.py:87-105: individual i's own re-fit, else the most recent earlier re-fit, else the column of `a`
        for (int k = 0; k < K; ++k) cur[k] = a->buf + (size_t)k * a->m;
        for (int64_t i = 0; i < n; ++i)
            for (int k = 0; k < K; ++k) colptr[(size_t)i * K + k] = cur[k];
        for (int64_t i = i0; i < i1; ++i) {
            cur[b->group_of[i]] = wgs_em_f_dev(em, (int32_t)(i - i0));
            for (int k = 0; k < K; ++k) colptr[(size_t)i * K + k] = cur[k];
        }
        if ((rc = wgs_score_create(scored, a, colptr.data(), (int32_t)i0, (int32_t)i1, &sc))) return rc;
        if ((rc = wgs_score_sums(sc, parts_out ? WGS_MODE_EXACT : score_mode, sums.data()))) return rc;
        if (world > 1) {
            // np.sum's running float64 total handed from shard to shard in SNP order on the stream (`world` broadcasts,
            // one readback); `start` = what precedes this shard, for the chain prediction
            if ((rc = wgs_score_totals_all(sc, comm, sums.data(), start.data()))) return rc;
        }
        for (size_t c = (size_t)i0 * K; c < (size_t)i1 * K; ++c) ll_out[c] = sums[c];
        g_loo_stats[1] += wall_s() - t_phase;
        t_phase = wall_s();
        if (parts_out) {
            if ((rc = wgs_score_chains_prepare(sc, P, world > 1 && rank > 0 ? start.data() : nullptr))) return rc;
            parts.assign(cells * P, 0.0f);
            // every rank has its block functions; the walks follow each other with the float32 carries (`world` broadcasts)
            if ((rc = wgs_score_chains_walk_all(sc, comm, parts.data()))) return rc;
            for (size_t c = (size_t)i0 * P * K; c < (size_t)i1 * P * K; ++c) parts_out[c] = parts[c];
            g_loo_stats[2] += wall_s() - t_phase;
        }
        // the last re-fit of each population in this batch becomes the current column
        std::vector<int32_t> last(K, -1);
        for (int x = 0; x < nb; ++x) last[grp[x]] = x;
        for (int k = 0; k < K; ++k)
            if (last[k] >= 0 && (rc = wgs_afset_set_column_from_em(a, k, em, last[k]))) return rc;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        guard.dismiss();
        wgs_score_destroy(sc);
        wgs_em_destroy(em);
    }
    return 0;
}

static int parts_exact_literal(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t P, const float *carry_in,
                               float *parts_out);

/* Exact partition sums: utils.partition_loglikes (utils.py:129-151) for every (individual,
 * population) -- serial float32 accumulation per partition in site order, continued from
 * carry_in (float32 [n*P*K], NULL = zeros: first shard) into parts_out (float32 [n*P*K]).
 * literal != 0 forces the one-lane-per-chain kernel (the cross-check of the block-parallel chains). */
int wgs_assign_parts_exact(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t P, const float *carry_in,
                           float *parts_out)
{
    WGS_REQUIRE(b && a && parts_out, "null argument");
    WGS_REQUIRE(a->m == b->m, "allele frequencies cover %lld SNPs, the Beagle shard %lld", (long long)a->m, (long long)b->m);
    WGS_REQUIRE(P >= 1, "partition count must be >= 1");
    if (chain_cand_lds_bytes(a->K, P, colptr != nullptr) > 64 * 1024 || b->n * (int64_t)a->K * P >= (1ll << 31))
        return parts_exact_literal(b, a, colptr, P, carry_in, parts_out);
    const size_t cells = (size_t)b->n * a->K;
    std::vector<double> sums(cells), start;
    if (carry_in) {           // the preceding shards' float64 sums are not known here: their float32 chains stand in
        start.assign(cells, 0.0);
        for (int64_t i = 0; i < b->n; ++i)
            for (int p = 0; p < P; ++p)
                for (int k = 0; k < a->K; ++k) start[(size_t)i * a->K + k] += (double)carry_in[((size_t)i * P + p) * a->K + k];
    }
    wgs_score *sc = nullptr;
    int rc = wgs_score_create(b, a, colptr, 0, (int32_t)b->n, &sc);
    if (!rc) rc = wgs_score_sums(sc, WGS_MODE_EXACT, sums.data());
    if (!rc) rc = wgs_score_chains_prepare(sc, P, carry_in ? start.data() : nullptr);
    if (!rc) rc = wgs_score_chains_walk(sc, carry_in, parts_out);
    wgs_score_destroy(sc);
    return rc;
}

int wgs_debug_parts_exact_literal(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t P, const float *carry_in,
                                  float *parts_out)
{
    WGS_REQUIRE(b && a && parts_out, "null argument");
    WGS_REQUIRE(a->m == b->m, "allele frequencies cover %lld SNPs, the Beagle shard %lld", (long long)a->m, (long long)b->m);
    WGS_REQUIRE(P >= 1, "partition count must be >= 1");
    return parts_exact_literal(b, a, colptr, P, carry_in, parts_out);
}

static int parts_exact_literal(wgs_beagle *b, wgs_afset *a, const float *const *colptr, int32_t P, const float *carry_in,
                               float *parts_out)
{
    wgs_ctx *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const int K = a->K;
    const int64_t n = b->n;
    const size_t cells = (size_t)n * P * K;
    const size_t off_carry = (sizeof(float) * cells + 255) & ~(size_t)255;
    const size_t off_acol = (off_carry + sizeof(float) * cells + 255) & ~(size_t)255;
    const size_t off_slabs = (off_acol + sizeof(float *) * K + 255) & ~(size_t)255;
    const size_t off_colptr = (off_slabs + sizeof(PartsSlab) * b->n_groups + 255) & ~(size_t)255;
    const size_t total = off_colptr + (colptr ? sizeof(float *) * n * K : 0);
    void *ws = nullptr;
    if (wgs_ctx_workspace(ctx, total, &ws)) return 1;
    char *base = reinterpret_cast<char *>(ws);
    float *d_parts = reinterpret_cast<float *>(base);
    float *d_carry = carry_in ? reinterpret_cast<float *>(base + off_carry) : nullptr;
    PartsSlab *d_slabs = reinterpret_cast<PartsSlab *>(base + off_slabs);
    const float **d_acol = reinterpret_cast<const float **>(base + off_acol);
    const float **d_colptr = colptr ? reinterpret_cast<const float **>(base + off_colptr) : nullptr;
    std::vector<const float *> acol(K);
    for (int k = 0; k < K; ++k) acol[k] = a->buf + (size_t)k * a->m;
    HIP_TRY(hipMemsetAsync(d_parts, 0, sizeof(float) * cells, ctx->stream));
    if (carry_in) HIP_TRY(hipMemcpyAsync(d_carry, carry_in, sizeof(float) * cells, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_acol, acol.data(), sizeof(float *) * K, hipMemcpyHostToDevice, ctx->stream));
    if (colptr) HIP_TRY(hipMemcpyAsync(d_colptr, colptr, sizeof(float *) * n * K, hipMemcpyHostToDevice, ctx->stream));
    std::vector<PartsSlab> slabs;
    int blocks = 0;
    for (int g = 0; g < b->n_groups; ++g) {
        const Slab &s = b->slabs[g];
        if (s.ncols == 0) continue;
        slabs.push_back({s.base, s.d_members, s.npairs, s.ncols, blocks});
        blocks += (s.ncols + 63) / 64;
    }
    HIP_TRY(hipMemcpyAsync(d_slabs, slabs.data(), sizeof(PartsSlab) * slabs.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    {
        AssignArgs args;
        args.slab = nullptr;
        args.members = nullptr;
        args.colptr = d_colptr;
        args.acol = d_acol;
        args.out = nullptr;
        args.m = b->m;
        args.site0 = b->site0;
        args.npairs = 0;
        args.ncols = 0;
        args.K = K;
        args.P = P;
        args.tiles_per_wave = 0;
        if (launch_parts_exact(ctx, args, d_slabs, (int)slabs.size(), blocks, d_carry, d_parts)) return 1;
    }
    HIP_TRY(hipMemcpyAsync(parts_out, d_parts, sizeof(float) * cells, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* ------------------------------------------------------------------ Fisher information (--ne_obs) */

int wgs_fisher_obs(wgs_beagle *b, wgs_afset *a, float *f_obs_mK, float *ne_obs_mK)
{
    WGS_REQUIRE(b && a && f_obs_mK && ne_obs_mK, "null argument");
    WGS_REQUIRE(a->m == b->m && a->K == b->n_groups, "allele frequencies (%lld x %d) do not match the population slabs (%lld x %d)",
                (long long)a->m, a->K, (long long)b->m, b->n_groups);
    wgs_ctx *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const int K = a->K;
    const size_t mk = (size_t)b->m * K;
    // workspace: [f (K x m) | ne (K x m) | transposed (m x K) | descs]
    const size_t off_desc = (3 * mk * sizeof(float) + 255) & ~(size_t)255;
    void *ws = nullptr;
    if (wgs_ctx_workspace(ctx, off_desc + sizeof(FisherDesc) * K, &ws)) return 1;
    float *d_f = reinterpret_cast<float *>(ws), *d_ne = d_f + mk, *d_t = d_ne + mk;
    FisherDesc *d_descs = reinterpret_cast<FisherDesc *>(reinterpret_cast<char *>(ws) + off_desc);
    std::vector<FisherDesc> descs;
    for (int g = 0; g < K; ++g) {
        const Slab &s = b->slabs[g];
        WGS_REQUIRE(s.ncols > 0, "population %d has no individuals", g);
        FisherDesc d;
        d.slab = s.base;
        d.th = a->buf + (size_t)g * a->m;
        d.f_out = d_f + (size_t)g * b->m;
        d.ne_out = d_ne + (size_t)g * b->m;
        d.npairs = s.npairs;
        d.ncols = s.ncols;
        descs.push_back(d);
    }
    HIP_TRY(hipMemcpyAsync(d_descs, descs.data(), sizeof(FisherDesc) * K, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (launch_fisher_pop(ctx, d_descs, K, b->m)) return 1;
    if (launch_transpose_Km_to_mK(ctx, d_f, d_t, b->m, K)) return 1;
    HIP_TRY(hipMemcpyAsync(f_obs_mK, d_t, mk * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (launch_transpose_Km_to_mK(ctx, d_ne, d_t, b->m, K)) return 1;
    HIP_TRY(hipMemcpyAsync(ne_obs_mK, d_t, mk * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* Per-site effective-sample-size terms (float32, fisher_cy.pyx:41-65) of individuals
 * [i0, i0 + count): rows_out[(i - i0) * m + s].  All of them must lie in one population slab. */
int wgs_fisher_ind_sites(wgs_beagle *b, wgs_afset *a, int32_t i0, int32_t count, float *rows_out)
{
    WGS_REQUIRE(b && a && rows_out && count > 0 && i0 >= 0 && (int64_t)i0 + count <= b->n, "bad argument");
    WGS_REQUIRE(a->m == b->m && a->K == b->n_groups, "allele frequencies do not match the population slabs");
    wgs_ctx *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const int g = b->group_of[i0];
    std::vector<int32_t> cols(count);
    for (int j = 0; j < count; ++j) {
        WGS_REQUIRE(b->group_of[i0 + j] == g, "individuals %d..%d span more than one population", i0, i0 + count - 1);
        cols[j] = b->col_of[i0 + j];
    }
    const size_t off_cols = ((size_t)count * b->m * sizeof(float) + 255) & ~(size_t)255;
    void *ws = nullptr;
    if (wgs_ctx_workspace(ctx, off_cols + sizeof(int32_t) * count, &ws)) return 1;
    float *d_out = reinterpret_cast<float *>(ws);
    int32_t *d_cols = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(ws) + off_cols);
    HIP_TRY(hipMemcpyAsync(d_cols, cols.data(), sizeof(int32_t) * count, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const Slab &s = b->slabs[g];
    if (launch_fisher_ind_sites(ctx, s.base, d_cols, a->buf + (size_t)g * a->m, d_out, b->m, s.npairs, count)) return 1;
    HIP_TRY(hipMemcpyAsync(rows_out, d_out, (size_t)count * b->m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* How NumPy sums a contiguous float32 vector of n elements (np.add.reduce behind np.mean, fisher.py:59): the reduction
 * hands its inner loop at most 8192 elements at a time (the iterator's buffer size -- measured: np.sum equals this
 * scheme and not one pairwise pass over the whole vector from n = 8193 on; np.setbufsize does not change it), each such
 * chunk is summed pairwise (em_kernels.hip: pairwise_leaf_kernel) and the chunk sums are added to the running float32
 * total in order.  The leaves and the order of the additions depend on n alone. */
namespace {
struct PairwisePlan {
    std::vector<int64_t> leaf_lo;
    std::vector<int32_t> leaf_len, prog;
};
void pairwise_plan(int64_t lo, int64_t n, PairwisePlan &p)
{
    if (n <= 128) {
        p.prog.push_back((int32_t)p.leaf_lo.size());
        p.leaf_lo.push_back(lo);
        p.leaf_len.push_back((int32_t)n);
        return;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    pairwise_plan(lo, n2, p);
    pairwise_plan(lo + n2, n - n2, p);
    p.prog.push_back(-1);
}
}  // namespace

/* fisher.py:52-59 for individuals [i0, i0 + count) of one population slab, entirely on the device:
 * means_out[i - i0] = np.mean of the individual's float32 per-site terms -- NumPy's pairwise float32 sum, divided by
 * the count in float64, stored as float32 -- without the count x m matrix ever crossing PCIe. */
static int fisher_ind_reduce(wgs_beagle *b, wgs_afset *a, int32_t i0, int32_t count, const float *carry_in, int64_t divide_by, float *means_out);

int wgs_fisher_ind_means(wgs_beagle *b, wgs_afset *a, int32_t i0, int32_t count, float *means_out)
{
    WGS_REQUIRE(b, "null argument");
    return fisher_ind_reduce(b, a, i0, count, nullptr, b->m, means_out);
}

/* The same reduction over SNP shards: sums_out[i - i0] = NumPy's running float32 total after this shard, continued
 * from carry_in (host, count floats, NULL = this is the first shard): total = total + pairwise(chunk) for every 8192-site
 * chunk of the shard (shards start at multiples of 8192 sites: comm.shard_range).  The last shard's totals divided by
 * the number of sites in float64 are np.mean's result. */
int wgs_fisher_ind_sums(wgs_beagle *b, wgs_afset *a, int32_t i0, int32_t count, const float *carry_in, float *sums_out)
{
    return fisher_ind_reduce(b, a, i0, count, carry_in, 0, sums_out);
}

static int fisher_ind_reduce(wgs_beagle *b, wgs_afset *a, int32_t i0, int32_t count, const float *carry_in, int64_t divide_by, float *means_out)
{
    WGS_REQUIRE(b && a && means_out && count > 0 && i0 >= 0 && (int64_t)i0 + count <= b->n, "bad argument");
    WGS_REQUIRE(a->m == b->m && a->K == b->n_groups, "allele frequencies do not match the population slabs");
    wgs_ctx *ctx = b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const int g = b->group_of[i0];
    std::vector<int32_t> cols(count);
    for (int j = 0; j < count; ++j) {
        WGS_REQUIRE(b->group_of[i0 + j] == g, "individuals %d..%d span more than one population", i0, i0 + count - 1);
        cols[j] = b->col_of[i0 + j];
    }
    PairwisePlan plan;
    for (int64_t lo = 0; lo < b->m; lo += 8192) {            // total = total + pairwise(chunk); the first chunk starts it
        pairwise_plan(lo, std::min<int64_t>(8192, b->m - lo), plan);
        if (lo > 0 || carry_in) plan.prog.push_back(-1);
    }
    const size_t nleaf = plan.leaf_lo.size(), nprog = plan.prog.size();
    WGS_REQUIRE(nleaf < (1u << 28), "too many SNPs for one pairwise plan");
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_rows = 0, o_sums = o_rows + up((size_t)count * b->m * sizeof(float)), o_means = o_sums + up((size_t)count * nleaf * sizeof(float)),
                 o_lo = o_means + up(sizeof(float) * count), o_len = o_lo + up(sizeof(int64_t) * nleaf), o_prog = o_len + up(sizeof(int32_t) * nleaf),
                 o_cols = o_prog + up(sizeof(int32_t) * nprog), o_carry = o_cols + up(sizeof(int32_t) * count),
                 total = o_carry + up(sizeof(float) * count);
    void *ws = nullptr;
    if (wgs_ctx_workspace(ctx, total, &ws)) return 1;
    char *w = reinterpret_cast<char *>(ws);
    float *d_rows = reinterpret_cast<float *>(w + o_rows), *d_sums = reinterpret_cast<float *>(w + o_sums), *d_means = reinterpret_cast<float *>(w + o_means);
    int64_t *d_lo = reinterpret_cast<int64_t *>(w + o_lo);
    int32_t *d_len = reinterpret_cast<int32_t *>(w + o_len), *d_prog = reinterpret_cast<int32_t *>(w + o_prog), *d_cols = reinterpret_cast<int32_t *>(w + o_cols);
    HIP_TRY(hipMemcpyAsync(d_lo, plan.leaf_lo.data(), sizeof(int64_t) * nleaf, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_len, plan.leaf_len.data(), sizeof(int32_t) * nleaf, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_prog, plan.prog.data(), sizeof(int32_t) * nprog, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_cols, cols.data(), sizeof(int32_t) * count, hipMemcpyHostToDevice, ctx->stream));
    float *d_carry = reinterpret_cast<float *>(w + o_carry);
    if (carry_in) HIP_TRY(hipMemcpyAsync(d_carry, carry_in, sizeof(float) * count, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));              // the host vectors above go out of use here
    const Slab &s = b->slabs[g];
    if (launch_fisher_ind_sites(ctx, s.base, d_cols, a->buf + (size_t)g * a->m, d_rows, b->m, s.npairs, count)) return 1;
    if (launch_pairwise_mean(ctx, d_rows, count, b->m, divide_by, d_lo, d_len, (int)nleaf, d_prog, (int)nprog, d_sums,
                             carry_in ? d_carry : nullptr, d_means))
        return 1;
    HIP_TRY(hipMemcpyAsync(means_out, d_means, sizeof(float) * count, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

/* ------------------------------------------------------------------ test hooks */

/* pairs = 2^20 threads x per_thread operand pairs; *mismatch = results of the kernel's Newton-core
 * divide that differ bitwise from the IEEE divide. */
int wgs_debug_div_mismatch(wgs_ctx *ctx, uint64_t seed, uint64_t per_thread, uint64_t *mismatch)
{
    WGS_REQUIRE(ctx && mismatch, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    void *ws = nullptr;
    if (wgs_ctx_workspace(ctx, 256, &ws)) return 1;
    unsigned long long *d = reinterpret_cast<unsigned long long *>(ws);
    HIP_TRY(hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
    if (launch_div_check(ctx, seed, per_thread, d)) return 1;
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *mismatch = h;
    return 0;
}

/* Largest relative error of the once-refined reciprocal of div_exact over all 2^23 float32 mantissas of the
 * denominator scaled by 2^exponent (the bound its exactness argument rests on: < 2^-48). */
int wgs_debug_rcp_error(wgs_ctx *ctx, int exponent, double *max_rel)
{
    WGS_REQUIRE(ctx && max_rel, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    void *ws = nullptr;
    if (wgs_ctx_workspace(ctx, 256, &ws)) return 1;
    unsigned long long *d = reinterpret_cast<unsigned long long *>(ws), h = 0;
    HIP_TRY(hipMemsetAsync(d, 0, sizeof h, ctx->stream));
    if (launch_rcp_error(ctx, exponent, d)) return 1;
    HIP_TRY(hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(max_rel, &h, sizeof h);
    return 0;
}

int wgs_debug_log_mismatch(wgs_ctx *ctx, uint32_t b0, uint32_t b1, uint64_t *count, uint32_t *first)
{
    WGS_REQUIRE(ctx && count && first, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long *d_count = nullptr;
    unsigned int *d_first = nullptr;
    HIP_TRY(hipMalloc(&d_count, sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&d_first, sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(hipMemsetAsync(d_first, 0xFF, sizeof(unsigned int), ctx->stream));
    int rc = launch_log_mismatch(ctx, b0, b1, d_count, d_first);
    unsigned long long c = 0;
    unsigned int f = 0;
    if (!rc && (hipMemcpyAsync(&c, d_count, sizeof c, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(&f, d_first, sizeof f, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        wgs_set_error("log self-test failed on the device");
        rc = 1;
    }
    (void)hipFree(d_count);
    (void)hipFree(d_first);
    *count = c;
    *first = f;
    return rc;
}

int wgs_debug_log_values(wgs_ctx *ctx, const float *x, float *out, int64_t n, int use_libm)
{
    WGS_REQUIRE(ctx && x && out && n >= 0, "bad argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    float *d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(float) * 2 * (size_t)n));
    int rc = 0;
    if (hipMemcpyAsync(d, x, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = 1;
    if (!rc) rc = launch_log_values(ctx, d, d + n, n, use_libm);
    if (!rc && (hipMemcpyAsync(out, d + n, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = 1;
    (void)hipFree(d);
    if (rc) wgs_set_error("log values: device operation failed");
    return rc;
}

/* ------------------------------------------------------------------ thin mirrors */

int wgs_emmaf_update(wgs_ctx *ctx, const float *L, int64_t m, int64_t n, float *f, int mode)
{
    WGS_REQUIRE(ctx && L && f, "null argument");
    wgs_beagle *b = nullptr;
    wgs_em *em = nullptr;
    int rc = 0;
    if (n == 0) {   // emMAF_cy.pyx:17,23 with an empty loop: tmp = 0.0, f[s] = 0.0/0.0
        for (int64_t s = 0; s < m; ++s) f[s] = nanf("");
        return 0;
    }
    if (m == 0) return 0;
    const int32_t grp = 0;
    rc = wgs_beagle_create(ctx, m, n, nullptr, 1, 0, &b);
    if (!rc) rc = wgs_beagle_upload_rows(b, L, 0, m);
    if (!rc) rc = wgs_em_create(b, 1, &grp, nullptr, mode, &em);
    if (!rc) rc = wgs_em_set_f(em, 0, f);
    if (!rc) rc = wgs_em_step(em, nullptr);
    if (!rc) rc = wgs_em_get_f(em, 0, f);
    wgs_em_destroy(em);
    wgs_beagle_destroy(b);
    return rc;
}

static int rmse1d_impl(wgs_ctx *ctx, const float *v1, const float *v2, int64_t m, double *out, int serial, int *serial_blocks);

int wgs_rmse1d(wgs_ctx *ctx, const float *v1, const float *v2, int64_t m, double *out)
{
    return rmse1d_impl(ctx, v1, v2, m, out, 0, nullptr);
}

/* Test hook: the same value through the literal one-lane serial kernel (serial != 0), or through
 * the block-parallel exact chain reporting how many blocks fell back to the serial loop. */
int wgs_debug_rmse1d(wgs_ctx *ctx, const float *v1, const float *v2, int64_t m, double *out, int serial, int *serial_blocks)
{
    return rmse1d_impl(ctx, v1, v2, m, out, serial, serial_blocks);
}

static int rmse1d_impl(wgs_ctx *ctx, const float *v1, const float *v2, int64_t m, double *out, int serial, int *serial_blocks)
{
    WGS_REQUIRE(ctx && v1 && v2 && out, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (m == 0) {   // 0.0f / 0.0f
        *out = nan("");
        return 0;
    }
    float *d = nullptr;
    void *work = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(float) * (2 * (size_t)m + 2)));
    HIP_TRY(hipMalloc(&work, rmse_chain_workspace_bytes(m)));
    int rc = 0;
    float res = 0.0f;
    int nser = 0;
    if (hipMemcpyAsync(d, v1, sizeof(float) * m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d + m, v2, sizeof(float) * m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = 1;
    if (!rc && hipMemsetAsync(d + 2 * m, 0, 2 * sizeof(float), ctx->stream) != hipSuccess) rc = 1;
    if (!rc) rc = serial ? launch_rmse_chain_serial(ctx, d, d + m, m, 0.0f, d + 2 * m)
                         : launch_rmse_chain(ctx, d, d + m, m, 0.0f, d + 2 * m, work, reinterpret_cast<int *>(d + 2 * m + 1));
    if (!rc && (hipMemcpyAsync(&res, d + 2 * m, sizeof(float), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(&nser, d + 2 * m + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = 1;
    (void)hipFree(d);
    (void)hipFree(work);
    if (serial_blocks) *serial_blocks = nser;
    if (rc) {
        wgs_set_error("rmse1d: device operation failed");
        return 1;
    }
    res = res / (float)m;           // emMAF_cy.pyx:32
    *out = sqrt((double)res);       // emMAF_cy.pyx:33
    return 0;
}

int wgs_loglike(wgs_ctx *ctx, const float *L, int64_t m, int64_t n, const float *A, int64_t K, float *vec, int64_t i,
                int64_t k, int mode)
{
    WGS_REQUIRE(ctx && L && A && vec, "null argument");
    WGS_REQUIRE(i >= 0 && i < n && k >= 0 && k < K, "individual %lld / population %lld out of range", (long long)i, (long long)k);
    if (m == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    float *d = nullptr;   // [2m g | m a | m vec]
    HIP_TRY(hipMalloc(&d, sizeof(float) * 4 * (size_t)m));
    int rc = 0;
    // strided host columns -> compact device vectors
    if (hipMemcpy2DAsync(d, 2 * sizeof(float), L + 2 * i, sizeof(float) * 2 * n, 2 * sizeof(float), m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpy2DAsync(d + 2 * m, sizeof(float), A + k, sizeof(float) * K, sizeof(float), m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d + 3 * m, vec, sizeof(float) * m, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = 1;
    if (!rc) rc = launch_loglike_site(ctx, reinterpret_cast<const float2 *>(d), d + 2 * m, d + 3 * m, m, mode);
    if (!rc && (hipMemcpyAsync(vec, d + 3 * m, sizeof(float) * m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = 1;
    (void)hipFree(d);
    if (rc) wgs_set_error("loglike: device operation failed");
    return rc;
}

}  // extern "C"
