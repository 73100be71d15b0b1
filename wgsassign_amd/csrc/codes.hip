// Class codes of a device-resident matrix (common.h: wgs_codes): when they are built, with which geometry, and what they hold.
// The encoder itself is codes_kernels.hip.
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "common.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void wgs_beagle_drop_codes(wgs_beagle *b)
{
    if (!b) return;
    if (wgs_codes *c = b->codes) {
        (void)hipSetDevice(b->ctx->device);
        (void)hipStreamSynchronize(b->ctx->stream);
        if (c->pool) (void)hipFree(c->pool);
        delete c;
    }
    b->codes = nullptr;
    b->codes_state = 0;
    b->direct_sweeps = 0;
    ++b->codes_generation;
}

static bool codes_switched_off()
{
    const char *env = getenv("WGSASSIGN_CODES");           // read at every use, so one process can compare both paths
    return env && env[0] == '0';
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

// Builds the class codes (a sample of the matrix decides whether and how, then one pass over it: ~2 x its streaming time).
// Not worth coding -- most SNPs with more classes than the largest table holds, or hardly fewer classes than individuals -- or
// no memory for the codes (a quarter of the matrix + the dictionaries): nullptr, and the direct kernels run.
// WGSASSIGN_CODES=0 turns the codes off altogether.
wgs_codes *wgs_beagle_codes(wgs_beagle *b, bool build)
{
    if (!b || b->codes_state < 0 || codes_switched_off()) return nullptr;
    if (b->codes_state > 0) return b->codes;
    if (!build) return nullptr;
    b->codes_state = -1;
    if (hipSetDevice(b->ctx->device) != hipSuccess) return nullptr;
    const double t0 = now_s();
    const int64_t tiles = wgs_ntiles(b->m);
    const size_t rows = (size_t)tiles * 64;
    wgs_codes *c = new wgs_codes();
    b->codes = c;
    auto fail = [&]() -> wgs_codes * {
        (void)hipGetLastError();
        wgs_beagle_drop_codes(b);
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
    // ---- the sample: classes per SNP over all individuals and per population slab
    // (the sample pass needs the slab table on the device, with no arrays behind it yet)
    const size_t slab_tab = ((sizeof(SlabCodes) * b->n_groups + 255) / 256) * 256;
    {
        void *tmp = nullptr;
        if (hipMalloc(&tmp, slab_tab) != hipSuccess) return fail();
        c->d_slabs = reinterpret_cast<SlabCodes *>(tmp);
        bool ok = hipMemcpy(c->d_slabs, c->slabs.data(), sizeof(SlabCodes) * b->n_groups, hipMemcpyHostToDevice) == hipSuccess;
        unsigned long long hg[256], hl[256];
        ok = ok && launch_class_sample(b, c, 4096, hg, hl, nullptr) == 0;
        (void)hipFree(tmp);
        c->d_slabs = nullptr;
        if (!ok) return fail();
        c->sample_mean_g = hist_mean(hg);
        c->sample_mean_l = hist_mean(hl);
        c->sample_ms = (now_s() - t0) * 1e3;
        // geometry: the table should stay under ~60 % full for all but a few SNPs in a thousand (those become rich)
        const int g999 = hist_quantile(hg, 0.999), g99 = hist_quantile(hg, 0.99);
        const char *force = getenv("WGSASSIGN_CODES_TABLE");   // experiments / tests: 64, 128 or 256 slots per SNP
        int slots = g99 <= 36 ? 64 : (g99 <= 80 ? 128 : 256);
        if (force && (atoi(force) == 64 || atoi(force) == 128 || atoi(force) == 256)) slots = atoi(force);
        c->snps_per_wave = 2048 / slots;
        // not worth coding: the typical SNP overflows the largest table, or has hardly fewer classes than individuals
        if (g99 >= 200 || (c->sample_mean_g * 2.0 > (double)b->n && !force)) return fail();
        c->drows = std::min(std::min(254, slots - slots / 8), (g999 + 4 + 7) & ~7);
        // SNPs per table of the coded scoring sweep: as many as keep a typical batch inside its LDS table
        c->score_batch = g99 * 16 <= WGS_BATCH_ROWS_CAP ? 16 : (g99 * 8 <= WGS_BATCH_ROWS_CAP ? 8 : 4);
        c->score_batch = std::min(c->score_batch, c->snps_per_wave);      // (the encoder checks a batch's rows inside one wavefront)
        // the coded EM sweep's table: a tile is swept directly when one of its 64 SNPs has more classes in the slab than rows,
        // so ~1 % of the tiles at most means ~1.5 in 10 000 (slab, SNP) pairs
        const int l_hi = hist_quantile(hl, 1.0 - 1.0 / 6400.0);
        c->lrows = (std::max(l_hi, 1) + 7) & ~7;
        if (c->lrows > 64 || l_hi >= 255) c->lrows = 0;
        if (const char *rows_env = getenv("WGSASSIGN_EM_TABLE_ROWS")) {    // experiments / tests: 8 .. 64, a multiple of 8
            const int r = atoi(rows_env);
            if (r >= 8 && r <= 64 && r % 8 == 0) c->lrows = r;
        }
    }
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
            off.push_back(take(8 * (size_t)tiles));                             // tile_rows
            off.push_back(take(with_local ? words * sizeof(uint32_t) : 0));     // lcodes
            off.push_back(take(with_local ? (size_t)tiles * c->lrows * 64 * sizeof(float2) : 0));   // ldict
        }
        return at;
    };
    const double ta = now_s();
    std::vector<size_t> off;
    size_t total = plan(c->lrows > 0, off);
    if (hipMalloc(&c->pool, total) != hipSuccess) {
        (void)hipGetLastError();
        c->pool = nullptr;
        if (c->lrows == 0) return fail();
        c->lrows = 0;                                      // without the slabs' own numbering: the scoring sweep can still use the codes
        total = plan(false, off);
        if (hipMalloc(&c->pool, total) != hipSuccess) {
            c->pool = nullptr;
            return fail();
        }
    }
    c->alloc_ms = (now_s() - ta) * 1e3;
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
    info[10] = 2048 / c->snps_per_wave;
    info[11] = b->m > 0 ? (double)c->rich_snps / (double)b->m : 0.0;
    info[12] = c->drows;
    info[13] = c->probe_rounds;
    info[14] = c->alloc_ms;
    info[15] = c->rows_batch;
    info[16] = c->sample_mean_g;
    info[17] = c->sample_mean_l;
    info[18] = c->score_batch;
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
