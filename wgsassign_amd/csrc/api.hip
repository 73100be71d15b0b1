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
#include "em_state.h"

static thread_local std::string g_err;

void wgs_set_error(const char *fmt, ...)
{
    char buf[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

std::atomic<long long> g_wgs_malloc_ns{0};

// Test hooks: named process-wide switches that only wgs_debug_hook (include/wgsassign_hip_debug.h) sets -- the product path reads
// no environment variable for them.  A handful of names, looked up rarely (per allocation request, per fit), so a locked list does.
#include <mutex>
static std::mutex g_hook_mutex;
static std::vector<std::pair<std::string, int64_t>> g_hooks;
int64_t wgs_hook(const char *name)
{
    std::lock_guard<std::mutex> lock(g_hook_mutex);
    for (auto &h : g_hooks)
        if (h.first == name) return h.second;
    return 0;
}
struct LiveEntry {
    void *obj, *parent, *parent2;
    int kind;
};
static std::mutex g_live_mutex;
static std::vector<LiveEntry> g_live;
void wgs_live_add(void *obj, int kind, void *parent, void *parent2)
{
    std::lock_guard<std::mutex> lock(g_live_mutex);
    g_live.push_back(LiveEntry{obj, parent, parent2, kind});
}
bool wgs_live_remove(void *obj)
{
    std::lock_guard<std::mutex> lock(g_live_mutex);
    for (size_t i = 0; i < g_live.size(); ++i)
        if (g_live[i].obj == obj) {
            g_live[i] = g_live.back();
            g_live.pop_back();
            return true;
        }
    return false;
}
extern "C" void wgs_em_destroy(wgs_em *em);
extern "C" void wgs_score_destroy(wgs_score *sc);
void wgs_live_destroy_children(void *parent)
{
    for (;;) {
        LiveEntry e{nullptr, nullptr, nullptr, 0};
        {
            std::lock_guard<std::mutex> lock(g_live_mutex);
            for (const LiveEntry &x : g_live)
                if (x.parent == parent || x.parent2 == parent) {
                    e = x;
                    break;
                }
        }
        if (!e.obj) return;
        if (e.kind == WGS_LIVE_EM) wgs_em_destroy(reinterpret_cast<wgs_em *>(e.obj));
        else wgs_score_destroy(reinterpret_cast<wgs_score *>(e.obj));
    }
}

extern "C" int wgs_debug_hook(const char *name, int64_t value)
{
    WGS_REQUIRE(name, "null argument");
    static const char *known[] = {"codes_alloc_delay_ms", "codes_alloc_release_after_sweeps", "em_fuse_without_agreement", "em_coded_extra_lds"};
    bool ok = false;
    for (const char *k : known) ok = ok || strcmp(k, name) == 0;
    WGS_REQUIRE(ok, "unknown test hook '%s'", name);
    std::lock_guard<std::mutex> lock(g_hook_mutex);
    for (auto &h : g_hooks)
        if (h.first == name) {
            h.second = value;
            return 0;
        }
    g_hooks.emplace_back(name, value);
    return 0;
}

extern "C" {
double wgs_malloc_seconds(void) { return (double)g_wgs_malloc_ns.load() * 1e-9; }

const char *wgs_last_error(void) { return g_err.c_str(); }
int wgs_version(void) { return WGS_ABI_VERSION; }
const char *wgs_build_id(void) { return WGS_BUILD_ID; }
const char *wgs_kernels_id(void) { return WGS_KERNELS_ID; }
const char *wgs_ingest_kernels_id(void) { return WGS_INGEST_KERNELS_ID; }

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
    HIP_TRY(hipEventCreate(&c->enc_ev0));
    HIP_TRY(hipEventCreate(&c->enc_ev1));
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
    if (ctx->ws_b) (void)hipFree(ctx->ws_b);
    for (wgs_ctx::PoolBlock &b : ctx->pool) (void)hipFree(b.p);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->enc_ev0) (void)hipEventDestroy(ctx->enc_ev0);
    if (ctx->enc_ev1) (void)hipEventDestroy(ctx->enc_ev1);
    delete ctx;
}

}   // extern "C"

hipError_t wgs_pool_malloc(wgs_ctx *ctx, void **p, size_t bytes)
{
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    int best = -1;
    for (size_t i = 0; i < ctx->pool.size(); ++i) {           // the smallest idle block that holds it without wasting more than half
        const wgs_ctx::PoolBlock &b = ctx->pool[i];
        if (!b.used && b.bytes >= bytes && b.bytes <= 2 * bytes + 4096 && (best < 0 || b.bytes < ctx->pool[best].bytes)) best = (int)i;
    }
    if (best >= 0) {
        ctx->pool[best].used = true;
        *p = ctx->pool[best].p;
        return hipSuccess;
    }
    void *q = nullptr;
    hipError_t e = wgs_malloc(&q, bytes);
    if (e != hipSuccess) {                                     // no memory: give the idle blocks back and try once more
        (void)hipGetLastError();
        for (size_t i = 0; i < ctx->pool.size();) {
            if (!ctx->pool[i].used) {
                (void)hipFree(ctx->pool[i].p);
                ctx->pool[i] = ctx->pool.back();
                ctx->pool.pop_back();
            } else {
                ++i;
            }
        }
        e = wgs_malloc(&q, bytes);
        if (e != hipSuccess) return e;
    }
    ctx->pool.push_back(wgs_ctx::PoolBlock{q, bytes, true});
    *p = q;
    return hipSuccess;
}

void wgs_pool_free(wgs_ctx *ctx, void *p)
{
    if (!p) return;
    size_t idle = 0;
    bool found = false;
    for (wgs_ctx::PoolBlock &b : ctx->pool) {
        if (b.p == p) b.used = false, found = true;
        if (!b.used) idle += b.bytes;
    }
    if (!found) {
        (void)hipFree(p);
        return;
    }
    while (idle > ((size_t)1 << 30) || ctx->pool.size() > 64) {      // keep the cache small: the largest idle block goes first
        int big = -1;
        for (size_t i = 0; i < ctx->pool.size(); ++i)
            if (!ctx->pool[i].used && (big < 0 || ctx->pool[i].bytes > ctx->pool[big].bytes)) big = (int)i;
        if (big < 0) break;
        idle -= ctx->pool[big].bytes;
        (void)hipFree(ctx->pool[big].p);
        ctx->pool[big] = ctx->pool.back();
        ctx->pool.pop_back();
    }
}

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
        HIP_TRY(wgs_malloc(&ctx->ws, want));
        ctx->ws_bytes = want;
    }
    *out = ctx->ws;
    return 0;
}

int wgs_ctx_workspace_b(wgs_ctx *ctx, size_t bytes, void **out)
{
    if (bytes > ctx->ws_b_bytes) {
        if (ctx->ws_b) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(ctx->ws_b));
            ctx->ws_b = nullptr;
            ctx->ws_b_bytes = 0;
        }
        const size_t want = (bytes + 4095) & ~(size_t)4095;
        HIP_TRY(wgs_malloc(&ctx->ws_b, want));
        ctx->ws_b_bytes = want;
    }
    *out = ctx->ws_b;
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
    wgs_live_destroy_children(b);             // EM batches and scores over this matrix go first
    wgs_beagle_drop_codes(b);
    (void)hipSetDevice(b->ctx->device);
    wgs_beagle_release_pool(b);
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
        if (wgs_malloc(&s.base, bytes) != hipSuccess) {
            wgs_set_error("hipMalloc of %zu bytes for population slab %d failed", bytes, g);
            return 1;
        }
        (void)hipMemsetAsync(s.base, 0, bytes, ctx->stream);
        b->bytes += (int64_t)bytes;
        bases[g] = s.base;
        nps[g] = s.npairs;
        if (wgs_malloc(&s.d_members, sizeof(int32_t) * s.ncols) != hipSuccess ||
            hipMemcpy(s.d_members, s.members.data(), sizeof(int32_t) * s.ncols, hipMemcpyHostToDevice) != hipSuccess) {
            wgs_set_error("could not upload the member table of slab %d", g);
            return 1;
        }
    }
    HIP_TRY(wgs_malloc(&b->d_group_of, sizeof(int32_t) * n));
    HIP_TRY(wgs_malloc(&b->d_col_of, sizeof(int32_t) * n));
    HIP_TRY(wgs_malloc(&b->d_npairs, sizeof(int32_t) * n_groups));
    HIP_TRY(wgs_malloc(&b->d_base, sizeof(float4 *) * n_groups));
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

static bool wgs_live_has_children(void *parent)
{
    std::lock_guard<std::mutex> lock(g_live_mutex);
    for (const LiveEntry &x : g_live)
        if (x.parent == parent || x.parent2 == parent) return true;
    return false;
}

/* A matrix created with room for more sites than its file held (include/wgsassign_hip.h).  The slabs are tile-major (64 SNPs per
 * tile): the rows of a smaller matrix are the first tiles of the larger one, unchanged, and the rows of its last tile beyond
 * `rows` are the zeros they were created with. */
int wgs_beagle_set_rows(wgs_beagle *b, int64_t rows)
{
    WGS_REQUIRE(b, "null argument");
    WGS_REQUIRE(rows > 0 && rows <= b->m, "a matrix of %lld rows cannot be set to %lld", (long long)b->m, (long long)rows);
    WGS_REQUIRE(!wgs_live_has_children(b), "the matrix is in use (EM batches or scores were made from it)");
    if (rows == b->m) return 0;
    HIP_TRY(hipSetDevice(b->ctx->device));
    wgs_beagle_drop_codes(b);
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    const int64_t tiles_old = wgs_ntiles(b->m), tiles_new = wgs_ntiles(rows);
    if (tiles_old > tiles_new + tiles_new / 10) {
        std::vector<float4 *> bases(b->n_groups, nullptr);
        for (int g = 0; g < b->n_groups; ++g) {
            Slab &s = b->slabs[g];
            bases[g] = s.base;
            if (s.ncols == 0) continue;
            const size_t bytes = (size_t)tiles_new * s.npairs * 64 * sizeof(float4);
            float4 *fresh = nullptr;
            if (wgs_malloc(reinterpret_cast<void **>(&fresh), bytes) != hipSuccess) {
                (void)hipGetLastError();                    // no room for the copy: the larger allocation stays
                continue;
            }
            HIP_TRY(hipMemcpyAsync(fresh, s.base, bytes, hipMemcpyDeviceToDevice, b->ctx->stream));
            HIP_TRY(hipStreamSynchronize(b->ctx->stream));
            HIP_TRY(hipFree(s.base));
            s.base = fresh;
            bases[g] = fresh;
        }
        HIP_TRY(hipMemcpy(b->d_base, bases.data(), sizeof(float4 *) * b->n_groups, hipMemcpyHostToDevice));
    }
    b->m = rows;
    b->bytes = 0;                                           // (of the rows in use: what the sweeps read and the cost models go by)
    for (const Slab &s : b->slabs) b->bytes += (int64_t)((size_t)tiles_new * s.npairs * 64 * sizeof(float4));
    return 0;
}

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

/* ------------------------------------------------------------------ allele-frequency sets */

int wgs_afset_create(wgs_ctx *ctx, int64_t m, int32_t K, wgs_afset **out)
{
    WGS_REQUIRE(ctx && out && m > 0 && K > 0, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    wgs_afset *a = new wgs_afset();
    a->ctx = ctx;
    a->m = m;
    a->K = K;
    if (wgs_malloc(&a->buf, sizeof(float) * (size_t)m * K) != hipSuccess) {
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
    wgs_live_destroy_children(a);             // scores over these columns go first
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
    HIP_TRY(wgs_malloc(&tmp, bytes));
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
    HIP_TRY(wgs_malloc(&tmp, bytes));
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
    HIP_TRY(wgs_malloc(&d_count, sizeof(unsigned long long)));
    HIP_TRY(wgs_malloc(&d_first, sizeof(unsigned int)));
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
    HIP_TRY(wgs_malloc(&d, sizeof(float) * 2 * (size_t)n));
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
    HIP_TRY(wgs_malloc(&d, sizeof(float) * (2 * (size_t)m + 2)));
    HIP_TRY(wgs_malloc(&work, rmse_chain_workspace_bytes(m)));
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
    HIP_TRY(wgs_malloc(&d, sizeof(float) * 4 * (size_t)m));
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
