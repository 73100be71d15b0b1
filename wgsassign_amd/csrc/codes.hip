// Class codes of a device-resident matrix (common.h: wgs_codes): when they are built, with which geometry, and what they hold.
// The encoder itself is codes_kernels.hip.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "common.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The codes' device memory comes from a helper thread: hipMalloc of VRAM that an earlier PROCESS used is cleared by the driver when
// it is handed out again (profiles/r04_alloc_ubench.txt: 0.3 ms ... 6 s for 42 GB, by what the box did before), and kernels of this
// thread are not held up by it (762 launches during a 1.3 s hipMalloc, none slower than usual; nor are copies and synchronisation --
// hipEventElapsedTime is, which is why the library reads elapsed times lazily: ctx->allocs_in_flight).  A sweep that wants the codes waits
// a few milliseconds for the memory -- long enough for the fast case -- and otherwise goes over the float32 slabs; a later sweep
// finds the pool ready and builds the codes then.
static void pool_join(wgs_beagle *b)
{
    if (b->pool_thread) {
        b->pool_thread->join();
        delete b->pool_thread;
        b->pool_thread = nullptr;
    }
}

// What the helper thread came back with becomes b->pool.  1: done; 0: still allocating; -1: no memory
static int pool_adopt(wgs_beagle *b)
{
    const int st = b->pool_state.load();
    if (st == 0) return b->pool ? 1 : -1;
    if (st == 1) return 0;
    pool_join(b);
    b->pool_state.store(0);
    if (st < 0) return -1;
    if (b->pool) (void)hipFree(b->pool);                    // (a smaller one of an earlier build)
    b->pool = b->pool_new;
    b->pool_bytes = b->pool_new_bytes;
    b->pool_new = nullptr;
    return 1;
}

// 1: b->pool holds at least `want` (or, failing that, `want_small`) bytes; 0: not yet; -1: no memory
static int pool_request(wgs_beagle *b, size_t want, size_t want_small, double grace_ms)
{
    if (b->pool && b->pool_bytes >= want) return 1;
    bool started = false;
    if (b->pool_state.load() == 0) {
        started = true;
        pool_join(b);
        b->pool_want = want;
        b->pool_want_small = want_small;
        b->pool_request_s = now_s();
        b->pool_state.store(1);
        const int dev = b->ctx->device;
        wgs_ctx *ctx = b->ctx;
        ++ctx->allocs_in_flight;
        const int delay_ms = (int)wgs_hook("codes_alloc_delay_ms");                 // tests: a slow hipMalloc on demand ...
        const int64_t release_at = wgs_hook("codes_alloc_release_after_sweeps");    // ... handed over after this many direct sweeps
        auto work = [b, dev, ctx, delay_ms, release_at] {
            const double t0 = now_s();
            if (delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
            while (release_at > 0 && b->direct_sweeps.load() < release_at && now_s() - t0 < 2.0) std::this_thread::sleep_for(std::chrono::microseconds(50));
            void *p = nullptr;
            size_t got = 0;
            if (hipSetDevice(dev) == hipSuccess) {
                if (hipMalloc(&p, b->pool_want) == hipSuccess) got = b->pool_want;
                else if (b->pool_want_small && b->pool_want_small < b->pool_want && hipMalloc(&p, b->pool_want_small) == hipSuccess) got = b->pool_want_small;
                else p = nullptr;
                (void)hipGetLastError();
            }
            b->pool_new = p;
            b->pool_new_bytes = got;
            b->pool_alloc_ms = (now_s() - t0) * 1e3;
            --ctx->allocs_in_flight;
            b->pool_state.store(p ? 2 : -1);
        };
        try {
            b->pool_thread = new std::thread(work);
        } catch (...) {                                      // no thread to be had: allocate here
            b->pool_thread = nullptr;
            work();
        }
    }
    // the call that starts the allocation gives it `grace_ms` (the fast case takes 0.3 ms); later calls only look (a fit of short
    // sweeps would otherwise wait at every one of them)
    const double t0 = now_s();
    while (b->pool_state.load() == 1 && (grace_ms < 0 || (started && (now_s() - t0) * 1e3 < grace_ms))) std::this_thread::sleep_for(std::chrono::microseconds(50));
    const int st = pool_adopt(b);
    if (st <= 0) return st;
    if (b->pool_bytes >= want || (want_small && b->pool_bytes >= want_small)) return 1;
    // (the request in flight was for an earlier, smaller content of the matrix: ask again)
    (void)hipFree(b->pool);
    b->pool = nullptr;
    b->pool_bytes = 0;
    return pool_request(b, want, want_small, grace_ms);
}

extern "C" int wgs_beagle_codes_wait(wgs_beagle *b, double *alloc_ms)
{
    WGS_REQUIRE(b, "null argument");
    if (alloc_ms) *alloc_ms = 0.0;
    if (b->pool_state.load() == 0) return 0;
    HIP_TRY(hipSetDevice(b->ctx->device));
    while (b->pool_state.load() == 1) std::this_thread::sleep_for(std::chrono::microseconds(100));
    (void)pool_adopt(b);
    if (alloc_ms) *alloc_ms = b->pool_alloc_ms;
    return 0;
}

/* Ends the helper thread and releases the codes' memory (wgs_beagle_destroy). */
void wgs_beagle_release_pool(wgs_beagle *b)
{
    if (!b) return;
    pool_join(b);
    if (b->pool_new) (void)hipFree(b->pool_new);
    b->pool_new = nullptr;
    b->pool_state.store(0);
    if (b->pool) (void)hipFree(b->pool);
    b->pool = nullptr;
    b->pool_bytes = 0;
}

void wgs_beagle_drop_codes(wgs_beagle *b)
{
    if (!b) return;
    if (wgs_codes *c = b->codes) {
        (void)hipSetDevice(b->ctx->device);
        (void)hipStreamSynchronize(b->ctx->stream);
        delete c;                                          // (the pool stays with the wgs_beagle: wgs_beagle_codes uses it again)
    }
    b->codes = nullptr;
    b->codes_state = 0;
    b->plan = wgs_codes_plan();
    b->direct_sweeps = 0;
    ++b->codes_generation;
}

static bool codes_switched_off()
{
    const char *env = getenv("WGSASSIGN_CODES");           // read at every use, so one process can compare both paths
    return env && env[0] == '0';
}

// Share of a float32 fit's time that coded sweeps save once the codes are there (two iterations per sweep): what em_codes_pay and the
// plan below decide with.  Measured round 5, warm fits: 0.59 at 14.7 classes per (slab, SNP) among 100 individuals, 0.55 at 12.7 among
// 62, 0.31 at ~11 among 36 -- min(0.6, 1.0 - 2.2 x classes / individuals) (round 4's 0.92 - 2.72 x predated the fused sweeps and turned
// the 36-individual shape away, which gains a third).  The sweep is bound by how many wavefronts a CU holds (24.3 / 13.2 /
// 9.4 / 7.5 ms at 2 / 4 / 6 / 8 per CU), i.e. by its table of lrows x 512 bytes: beyond the 24 rows of fixed-error data (13 wavefronts
// per CU) what is NOT saved grows by (13 / wavefronts)^0.8 -- quality-dependent likelihoods (26 classes per slab, ~56 rows: 5 wavefronts)
// save nothing, as measured in round 4.
double wgs_em_codes_saving(double classes_per_slab, double cols, int lrows)
{
    const double saves = std::max(0.0, std::min(0.6, 1.0 - 2.2 * classes_per_slab / std::max(1.0, cols)));
    if (lrows <= 24) return saves;
    const double waves = std::max(1.0, std::min(13.0, floor(160.0 * 1024.0 / ((double)lrows * 512.0))));
    return std::max(0.0, 1.0 - (1.0 - saves) * pow(13.0 / waves, 0.8));
}

// Smallest c with at least `share` of the histogram's mass at or below it (the overflow bin 255 counts as 255).
static int hist_quantile(const unsigned long long *h, double share)
{
    unsigned long long total = 0, run = 0;
    for (int i = 0; i < 256; ++i) total += h[i];
    if (!total) return 0;
    const double want = share * (double)total;
    for (int i = 0; i < 256; ++i) {
        run += h[i];
        if ((double)run >= want) return i;
    }
    return 255;
}

static double hist_mean(const unsigned long long *h)
{
    unsigned long long total = 0;
    double sum = 0.0;
    for (int i = 0; i < 256; ++i) total += h[i], sum += (double)i * (double)h[i];
    return total ? sum / (double)total : 0.0;
}

// What the sample pass says about the matrix (cached in the wgs_beagle until its rows change): whether class codes are worth having at
// all, and with which geometry.  ~0.3-0.5 ms (a few thousand groups of 8 SNPs through the largest hash table, one readback).
const wgs_codes_plan *wgs_beagle_codes_plan(wgs_beagle *b)
{
    if (!b) return nullptr;
    wgs_codes_plan &P = b->plan;
    if (P.state != 0) return &P;
    P = wgs_codes_plan();
    P.state = -1;
    if (hipSetDevice(b->ctx->device) != hipSuccess) return &P;
    const double t0 = now_s();
    // (the sample pass needs the slab table on the device, with no arrays behind it)
    wgs_codes tmpc;
    tmpc.slabs.resize(b->n_groups);
    for (int g = 0; g < b->n_groups; ++g) tmpc.slabs[g].nquads = (b->slabs[g].ncols + 3) / 4;
    void *tmp = nullptr;
    if (wgs_ctx_workspace_b(b->ctx, sizeof(SlabCodes) * b->n_groups, &tmp)) return &P;
    tmpc.d_slabs = reinterpret_cast<SlabCodes *>(tmp);
    if (hipMemcpy(tmpc.d_slabs, tmpc.slabs.data(), sizeof(SlabCodes) * b->n_groups, hipMemcpyHostToDevice) != hipSuccess) return &P;
    unsigned long long hg[256], hl[256];
    if (launch_class_sample(b, &tmpc, 4096, hg, hl, nullptr)) {
        (void)hipGetLastError();
        return &P;
    }
    P.mean_g = hist_mean(hg);
    P.mean_l = hist_mean(hl);
    // geometry: the table should stay under ~60 % full for all but a few SNPs in a thousand (those become rich)
    const int g999 = hist_quantile(hg, 0.999), g99 = hist_quantile(hg, 0.99);
    const char *force = getenv("WGSASSIGN_CODES_TABLE");   // experiments / tests: 64, 128 or 256 slots per SNP
    P.slots = g99 <= 44 ? 64 : (g99 <= 88 ? 128 : 256);       // (a larger table costs 1.4 x / 2.4 x the encode time of the 64-slot one)
    if (force && (atoi(force) == 64 || atoi(force) == 128 || atoi(force) == 256)) P.slots = atoi(force);
    P.drows = std::min(std::min(254, P.slots - P.slots / 8), (g999 + 4 + 7) & ~7);
    // SNPs per table of the coded scoring sweep: as many as keep a typical batch inside its LDS table
    P.score_batch = g99 * 16 <= WGS_BATCH_ROWS_CAP ? 16 : (g99 * 8 <= WGS_BATCH_ROWS_CAP ? 8 : 4);
    P.score_batch = std::min(P.score_batch, WGS_ENC_SLOTS / P.slots);      // (the encoder checks a batch's rows inside one wavefront)
    // the coded EM sweep's table: a tile is swept directly when one of its 64 SNPs has more classes in the slab than rows,
    // so ~1 % of the tiles at most means ~1.5 in 10 000 (slab, SNP) pairs
    const int l_hi = hist_quantile(hl, 1.0 - 1.0 / 6400.0);
    P.lrows = (std::max(l_hi, 1) + 7) & ~7;
    if (P.lrows > 64 || l_hi >= 255) P.lrows = 0;
    if (const char *rows_env = getenv("WGSASSIGN_EM_TABLE_ROWS")) {    // experiments / tests: 8 .. 64, a multiple of 8
        const int r = atoi(rows_env);
        if (r >= 8 && r <= 64 && r % 8 == 0) P.lrows = r;
    }
    // the slabs' own numbering is for the coded EM sweep only: when even a long fit could not repay the encode pass (many classes per
    // slab: the sweep would save too little -- em_api.hip: em_codes_pay has the model), the pass leaves it out and is a third cheaper
    {
        int groups = 0;
        for (int g = 0; g < b->n_groups; ++g) groups += b->slabs[g].ncols > 0;
        const double cols = (double)b->n / std::max(1, groups);
        const double saves = wgs_em_codes_saving(P.mean_l, cols, P.lrows);
        if (!getenv("WGSASSIGN_EM_TABLE_ROWS") && !getenv("WGSASSIGN_EM_CODES_SWEEPS") &&
            14.0 * saves * ((double)b->bytes / 6.0e9) <= wgs_codes_build_ms_estimate(b, P.slots))
            P.lrows = 0;
    }
    P.sample_ms = (now_s() - t0) * 1e3;
    // not worth coding: the typical SNP overflows the largest table, or has hardly fewer classes than individuals
    P.state = (g99 >= 200 || (P.mean_g * 2.0 > (double)b->n && !force)) ? -1 : 1;
    return &P;
}

// The encode pass streams the matrix once and writes about half of it again (codes, the slabs' own codes and dictionaries): measured
// 80 GB in 33 ms, 8 GB in 3.8 ms, 1.6 GB in 1.1 ms with 64-slot tables, ~1.4 x / 2.4 x that with 128 / 256 slots, + ~0.5 ms of sample
// pass, allocation and readbacks.  Without the slabs' own numbering (a build for a scoring sweep) 0.78 x that: 27 ms, 3.1 ms.
double wgs_codes_build_ms_estimate(const wgs_beagle *b, int slots, bool with_slab_numbering)
{
    return (double)b->bytes / 2.3e9 * (slots <= 64 ? 1.0 : (slots == 128 ? 1.4 : 2.4)) * (with_slab_numbering ? 1.0 : 0.78) + 0.5;
}

// Whether a scoring sweep with shared columns over K populations should build the codes: the direct sweep costs ~1.2e-12 s per
// (SNP, individual, population) (119 ms at 10M x 1000 x 10, on the FP64 issue roof); the coded one: wgs_codes_scoring_model.
bool wgs_codes_scoring_model(wgs_beagle *b, int K, double *direct_ms, double *coded_share, double *build_ms)
{
    const wgs_codes_plan *P = wgs_beagle_codes_plan(b);
    *direct_ms = 1.2e-9 * (double)b->m * (double)b->n * (double)K;      // (1.2e-8 until round 5: ten times the sweep's time -- found when bench.py began to print this beside the measurement)
    *coded_share = 1.0;
    *build_ms = 0.0;
    if (!P || P->state <= 0) return false;
    // the coded sweep: per SNP and population the table work (one evaluation per class, 2.9e-9 ms) and the look-ups of one workgroup's
    // 1024 individuals, whether they exist or not (3.65e-8 ms) -- a workgroup per 1024 individuals, each filling its own table.  Fitted
    // to 10M x 1000 x 10 (26 classes: 11.3 ms since the kernel stopped waiting for its own loads, 13.0 before: both constants x 0.87),
    // 2M x 1000 x 10 with 73 classes (4.8 ms); 2M x 500 x 8 (2.35 ms), 6.25M x 2000 x 20 (32 ms) and 5M x 180 x 5 (where building
    // for ONE sweep loses a millisecond: round 4's share, 1.25 x classes / individuals + 0.09, built there) check it to 10-20 %.
    const double groups = (double)((b->n + 1023) / 1024);
    const double coded_ms = (double)b->m * (double)K * groups * (2.9e-9 * P->mean_g + 3.65e-8);
    *coded_share = std::min(1.0, coded_ms / std::max(1e-9, *direct_ms));
    *build_ms = wgs_codes_build_ms_estimate(b, P->slots, false);
    return true;
}

bool wgs_codes_pay_for_scoring(wgs_beagle *b, int K)
{
    if (getenv("WGSASSIGN_CODES_TABLE") || getenv("WGSASSIGN_SCORE_CODES_ALWAYS")) return true;      // experiments / tests
    double direct_ms, coded_share, build_ms;
    if (!wgs_codes_scoring_model(b, K, &direct_ms, &coded_share, &build_ms)) return false;
    return direct_ms * (1.0 - coded_share) > build_ms;
}

// Builds the class codes (the plan of the sample pass decides whether and how, then one pass over the matrix: ~2 x its streaming
// time).  Not worth coding -- most SNPs with more classes than the largest table holds, or hardly fewer classes than individuals --
// or no memory for the codes (half of the matrix): nullptr, and the direct kernels run.  WGSASSIGN_CODES=0 turns the codes off altogether.
wgs_codes *wgs_beagle_codes(wgs_beagle *b, bool build, bool wait, bool for_scoring_only)
{
    if (!b || b->codes_state < 0 || codes_switched_off()) return nullptr;
    if (b->codes_state > 0) return b->codes;
    if (!build) return nullptr;
    b->codes_state = -1;
    if (hipSetDevice(b->ctx->device) != hipSuccess) return nullptr;
    const double t0 = now_s();
    const wgs_codes_plan *P = wgs_beagle_codes_plan(b);
    if (!P || P->state <= 0) return nullptr;
    const int64_t tiles = wgs_ntiles(b->m);
    const size_t rows = (size_t)tiles * 64;
    wgs_codes *c = new wgs_codes();
    b->codes = c;
    auto fail = [&]() -> wgs_codes * {
        (void)hipGetLastError();
        const wgs_codes_plan keep = b->plan;
        wgs_beagle_drop_codes(b);
        b->plan = keep;
        b->codes_state = -1;
        return nullptr;
    };
    c->generation = ++b->codes_generation;
    c->slabs.resize(b->n_groups);
    int quad0 = 0;
    for (int g = 0; g < b->n_groups; ++g) {
        SlabCodes &s = c->slabs[g];
        s.nquads = (b->slabs[g].ncols + 3) / 4;
        s.quad0 = quad0;
        quad0 += s.nquads;
    }
    c->total_quads = quad0;
    c->sample_mean_g = P->mean_g;
    c->sample_mean_l = P->mean_l;
    c->sample_ms = P->sample_ms;
    c->snps_per_wave = WGS_ENC_SLOTS / P->slots;
    c->drows = P->drows;
    c->score_batch = P->score_batch;
    // a scoring sweep needs the class codes and the dictionary only: the slabs' own numbering (the coded EM sweep's) is left out of a
    // build it asks for -- a third of the pass and two thirds of the memory
    c->lrows = for_scoring_only ? 0 : P->lrows;
    c->local_skipped = for_scoring_only && P->lrows > 0;
    const size_t slab_tab = ((sizeof(SlabCodes) * b->n_groups + 255) / 256) * 256;
    // ---- one allocation for everything
    auto plan = [&](bool with_local, std::vector<size_t> &off) -> size_t {
        size_t at = 0;
        auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) / 256 * 256; return o; };
        off.clear();
        off.push_back(take(slab_tab));
        off.push_back(take(rows));                                             // ncls
        off.push_back(take((size_t)tiles * c->drows * 64 * sizeof(float2)));   // dict
        off.push_back(take(sizeof(uint4) * 2 * (size_t)tiles * (64 / c->snps_per_wave)));   // wave_stats
        for (int g = 0; g < b->n_groups; ++g) {
            const size_t words = (size_t)tiles * c->slabs[g].nquads * 64;
            off.push_back(take(words * sizeof(uint32_t)));                      // codes
            off.push_back(take(WGS_TILE_ROWS_BYTES * (size_t)tiles));           // tile_rows
            off.push_back(take(with_local ? words * sizeof(uint32_t) : 0));     // lcodes
            off.push_back(take(with_local ? (size_t)tiles * c->lrows * 64 * sizeof(float2) : 0));   // ldict
        }
        return at;
    };
    const double ta = now_s();
    std::vector<size_t> off;
    size_t total = plan(c->lrows > 0, off);
    {
        // the pool of an earlier build of this matrix (dropped because rows changed) is used again when it is large enough; else the
        // helper thread fetches one: at least without the slabs' own numbering when the full size is not to be had
        std::vector<size_t> off_small;
        const size_t small = c->lrows > 0 ? plan(false, off_small) : 0;
        const char *grace_env = getenv("WGSASSIGN_CODES_ALLOC_WAIT_MS");    // (< 0: wait however long it takes; the test suite does)
        const int got = pool_request(b, total, small, wait ? -1.0 : (grace_env ? atof(grace_env) : 3.0));
        if (got == 0) {
            // not there yet: nothing is decided -- this call goes without codes, the next one asks again
            delete c;
            b->codes = nullptr;
            b->codes_state = 0;
            --b->codes_generation;
            return nullptr;
        }
        if (got < 0) return fail();
        if (b->pool_bytes < total) {
            c->lrows = 0;                                  // without the slabs' own numbering: the scoring sweep can still use the codes
            total = plan(false, off);
        }
    }
    c->pool = b->pool;
    c->alloc_ms = b->pool_alloc_ms;                        // the helper thread's hipMalloc (0 when an earlier pool was used again)
    c->alloc_wait_ms = (now_s() - ta) * 1e3;               // what this call waited for it
    b->pool_alloc_ms = 0.0;
    char *base = reinterpret_cast<char *>(c->pool);
    c->d_slabs = reinterpret_cast<SlabCodes *>(base + off[0]);
    c->ncls = reinterpret_cast<uint8_t *>(base + off[1]);
    c->dict = reinterpret_cast<float2 *>(base + off[2]);
    c->wave_stats = reinterpret_cast<uint4 *>(base + off[3]);
    for (int g = 0; g < b->n_groups; ++g) {
        SlabCodes &s = c->slabs[g];
        if (s.nquads == 0) continue;
        s.codes = reinterpret_cast<uint32_t *>(base + off[4 + 4 * g]);
        s.tile_rows = reinterpret_cast<uint8_t *>(base + off[5 + 4 * g]);
        if (c->lrows > 0) {
            s.lcodes = reinterpret_cast<uint32_t *>(base + off[6 + 4 * g]);
            s.ldict = reinterpret_cast<float2 *>(base + off[7 + 4 * g]);
            c->local_bytes += (int64_t)((size_t)tiles * s.nquads * 64 * sizeof(uint32_t) + (size_t)tiles * c->lrows * 64 * sizeof(float2));
        }
    }
    c->bytes = (int64_t)total - c->local_bytes;
    if (hipMemcpy(c->d_slabs, c->slabs.data(), sizeof(SlabCodes) * b->n_groups, hipMemcpyHostToDevice) != hipSuccess) return fail();
    if (launch_class_encode(b, c)) return fail();          // (sets kernel_ms from HIP events around the kernel)
    c->build_ms = (now_s() - t0) * 1e3;
    b->codes_state = 1;
    return c;
}

extern "C" {

/* Class codes of the matrix (csrc/common.h: wgs_codes), see include/wgsassign_hip.h for info[0..19].  Builds the codes if
 * they have not been tried yet. */
int wgs_beagle_codes_info(wgs_beagle *b, double *info)
{
    WGS_REQUIRE(b && info, "null argument");
    wgs_codes *c = wgs_beagle_codes(b);
    for (int i = 0; i < 20; ++i) info[i] = 0.0;
    if (!c) return 0;
    const double coded = (double)b->m - (double)c->rich_snps;
    info[0] = 1.0;
    info[1] = c->cmax;
    info[2] = (double)(c->bytes + c->local_bytes);
    info[3] = c->build_ms;
    info[4] = coded > 0 ? c->sum_ncls / coded : 0.0;
    info[5] = c->kernel_ms;
    info[6] = c->sample_ms;
    info[7] = (double)c->local_bytes;
    info[8] = c->lrows;
    info[9] = c->local_direct_share;
    info[10] = WGS_ENC_SLOTS / c->snps_per_wave;
    info[11] = b->m > 0 ? (double)c->rich_snps / (double)b->m : 0.0;
    info[12] = c->drows;
    info[13] = c->probe_rounds;
    info[14] = c->alloc_ms;
    info[15] = c->rows_batch;
    info[16] = c->sample_mean_g;
    info[17] = c->sample_mean_l;
    info[18] = c->score_batch;
    info[19] = c->alloc_wait_ms;
    return 0;
}

/* 1: the class codes exist, 0: not tried yet (nothing has asked for them), -1: tried and not worth it / no memory.  Builds nothing. */
int wgs_beagle_codes_state(wgs_beagle *b) { return b ? b->codes_state : 0; }

/* Builds the class codes now instead of at the first sweep that asks for them, e.g. while the host is still busy with
 * something else.  Returns 0 also when the matrix is not worth coding.  (`em` is kept for callers of version 1: the slabs' own
 * numbering is part of the one encode pass now.) */
int wgs_beagle_codes_prepare(wgs_beagle *b, int em)
{
    WGS_REQUIRE(b, "null argument");
    (void)em;
    (void)wgs_beagle_codes(b);
    return 0;
}

}   // extern "C"
