// Scoring of the C ABI (include/wgsassign_hip.h: wgs_score_*, wgs_assign, wgs_loo, exact partition sums): glassy.py:18-112 and
// utils.py:129-151 on device-resident data.  Host-side orchestration only; the arithmetic is in assign_kernels.hip.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>

#include "common.h"

extern "C" {

/* ------------------------------------------------------------------ assignment / scoring */

int wgs_assign_last_ms(wgs_ctx *ctx, float *ms)
{
    WGS_REQUIRE(ctx && ms, "null argument");
    if (ctx->assign_ms_pending) {
        if (ctx->allocs_in_flight.load() > 0) {              // the query would wait for that hipMalloc: not known yet (ask again later)
            *ms = -1.0f;
            return 0;
        }
        HIP_TRY(hipSetDevice(ctx->device));
        if (hipEventElapsedTime(&ctx->last_assign_ms, ctx->ev0, ctx->ev1) != hipSuccess) ctx->last_assign_ms = 0.0f;
        (void)hipGetLastError();
        ctx->assign_ms_pending = false;
    }
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
    if (!sc || !wgs_live_remove(sc)) return;          // (destroyed already, e.g. together with its matrix or frequency set)
    (void)hipSetDevice(sc->b->ctx->device);
    (void)hipStreamSynchronize(sc->b->ctx->stream);
    void *bufs[] = {sc->d_acol, sc->d_colptr, sc->d_slabs[0], sc->d_slabs[1] == sc->d_slabs[0] ? nullptr : sc->d_slabs[1], sc->d_S,
                    sc->d_out, sc->d_start, sc->d_run, sc->d_coded, sc->d_cand, sc->d_carry, sc->d_parts, sc->d_nserial, sc->d_chunks};
    for (void *p : bufs)
        if (p) wgs_pool_free(sc->b->ctx, p);
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
    HIP_TRY(wgs_pool_malloc(sc->b->ctx, &sc->d_slabs[which], sizeof(ScoreSlab) * tab.size()));
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
    wgs_live_add(sc, WGS_LIVE_SCORE, b, a);
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
    HIP_TRY(wgs_pool_malloc(ctx, &sc->d_acol, sizeof(float *) * a->K));
    HIP_TRY(hipMemcpy(sc->d_acol, acol.data(), sizeof(float *) * a->K, hipMemcpyHostToDevice));
    if (colptr) {
        HIP_TRY(wgs_pool_malloc(ctx, &sc->d_colptr, sizeof(float *) * sc->cells));
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
    if (wgs_pool_malloc(ctx, &sc->d_S, sizeof(double) * (size_t)sc->nblocks * sc->cells) != hipSuccess) {
        wgs_set_error("hipMalloc of %zu bytes for the block sums failed", sizeof(double) * (size_t)sc->nblocks * sc->cells);
        return 1;
    }
    HIP_TRY(wgs_pool_malloc(ctx, &sc->d_out, sizeof(double) * sc->cells));
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
    // (built for this sweep only when what it saves exceeds the encode pass: codes.hip: wgs_codes_pay_for_scoring)
    wgs_codes *codes = sc->per_ind ? nullptr : wgs_beagle_codes(sc->b, false);
    if (!codes && !sc->per_ind && wgs_codes_pay_for_scoring(sc->b, sc->K)) codes = wgs_beagle_codes(sc->b, true, false, true);
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
        if (sc->d_coded) wgs_pool_free(ctx, sc->d_coded);
        sc->d_coded = nullptr;
        sc->n_coded = (int)tab.size();
        sc->coded_quads = quad0;
        if (!tab.empty()) {
            HIP_TRY(wgs_pool_malloc(ctx, &sc->d_coded, sizeof(CodedSlabHost) * tab.size()));
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
    if (!sc->d_chunks && wgs_pool_malloc(ctx, &sc->d_chunks, sizeof(double) * (size_t)((sc->nblocks + 1) / 2) * sc->cells) != hipSuccess) {
        wgs_set_error("hipMalloc of the chunk sums failed");
        return 1;
    }
    if (launch_block_prefix(ctx, sc->d_S, sc->nblocks, sc->cells, sc->d_out, 1, sc->d_chunks)) return 1;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, sc->d_out, sizeof(double) * sc->cells, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->assign_ms_pending = true;
    sc->have_prefix = (mode == WGS_MODE_EXACT);
    return 0;
}

/* Test hook (include/wgsassign_hip_debug.h): the sums of the last wgs_score_sums per CHUNK of 8192 sites (two blocks; the addends of
 * np.sum's running total, what wgs_score_total_from folds): out[c * cells + i * K + k], ceil(nblocks / 2) x cells float64.  Lets a
 * test hold a full-size sweep to the oracle at any subset of the chunks: every float64 partial sum inside a chunk is exact. */
int wgs_debug_score_chunks(wgs_score *sc, double *out, int64_t *nchunks)
{
    WGS_REQUIRE(sc && nchunks, "null argument");
    WGS_REQUIRE(sc->d_chunks, "wgs_debug_score_chunks needs wgs_score_sums first");
    *nchunks = (sc->nblocks + 1) / 2;
    if (!out) return 0;
    wgs_ctx *ctx = sc->b->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(out, sc->d_chunks, sizeof(double) * (size_t)*nchunks * sc->cells, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
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
    if (!sc->d_start) HIP_TRY(wgs_pool_malloc(ctx, &sc->d_start, sizeof(double) * sc->cells));
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
    if (!sc->d_start) HIP_TRY(wgs_pool_malloc(ctx, &sc->d_start, bytes));
    if (!sc->d_run) HIP_TRY(wgs_pool_malloc(ctx, &sc->d_run, bytes + wgs_comm_tail_bytes()));      // (+ the sender's tag row behind the totals)
    HIP_TRY(hipMemsetAsync(sc->d_start, 0, bytes, ctx->stream));
    const int32_t generation = comm ? wgs_comm_next_generation(comm) : 0;
    for (int r = 0; r < world; ++r) {
        if (r == rank) {
            if (r > 0) HIP_TRY(hipMemcpyAsync(sc->d_start, sc->d_run, bytes, hipMemcpyDeviceToDevice, ctx->stream));   // what precedes this shard
            if (launch_chunk_total(ctx, sc->d_chunks, (sc->nblocks + 1) / 2, sc->cells, r > 0 ? sc->d_start : nullptr, sc->d_run)) return 1;
        }
        const wgs_coll_tag tag = {WGS_OP_SCORE_TOTALS, generation, r, (int32_t)(sc->row_hi - sc->row_lo), r, 0};
        if (world > 1 && wgs_comm_bcast_tagged(comm, sc->d_run, (int64_t)bytes, r, &tag)) return 1;
    }
    HIP_TRY(hipMemcpyAsync(totals_out, sc->d_run, bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (before_out) HIP_TRY(hipMemcpyAsync(before_out, sc->d_start, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return wgs_comm_check(comm);
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
            if (p) wgs_pool_free(ctx, p);
        sc->d_cand = nullptr;
        sc->d_carry = sc->d_parts = nullptr;
        sc->P = 0;
        if (wgs_pool_malloc(ctx, &sc->d_cand, sizeof(uint32_t) * chains * sc->nblocks) != hipSuccess) {
            wgs_set_error("hipMalloc of %zu bytes for the partition-chain block functions failed", sizeof(uint32_t) * chains * sc->nblocks);
            return 1;
        }
        HIP_TRY(wgs_pool_malloc(ctx, &sc->d_carry, sizeof(float) * chains));
        HIP_TRY(wgs_pool_malloc(ctx, &sc->d_parts, sizeof(float) * chains + wgs_comm_tail_bytes()));   // (+ the sender's tag row behind the carries)
        if (!sc->d_nserial) HIP_TRY(wgs_pool_malloc(ctx, &sc->d_nserial, sizeof(int32_t)));
        if (!sc->d_start) HIP_TRY(wgs_pool_malloc(ctx, &sc->d_start, sizeof(double) * sc->cells));
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
    ctx->assign_ms_pending = true;
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
    ctx->assign_ms_pending = true;
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
    const int32_t generation = comm ? wgs_comm_next_generation(comm) : 0;
    for (int r = 0; r < world; ++r) {
        if (r == rank) {
            if (r > 0) HIP_TRY(hipMemcpyAsync(sc->d_carry, sc->d_parts, bytes, hipMemcpyDeviceToDevice, ctx->stream));
            if (chains_walk_enqueue(sc, r > 0)) return 1;
            HIP_TRY(hipMemcpyAsync(&sc->last_serial_blocks, sc->d_nserial, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        }
        const wgs_coll_tag tag = {WGS_OP_PART_CHAINS, generation, r, sc->P, r, 0};
        if (world > 1 && wgs_comm_bcast_tagged(comm, sc->d_parts, (int64_t)bytes, r, &tag)) return 1;
    }
    HIP_TRY(hipMemcpyAsync(parts_out, sc->d_parts, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->assign_ms_pending = true;
    return wgs_comm_check(comm);
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
    ctx->assign_ms_pending = false;
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
    ctx->assign_ms_pending = false;
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
    ctx->assign_ms_pending = true;
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
        const wgs_coll_tag tag = {WGS_OP_LOO_BATCH, wgs_comm_next_generation(comm), 0, (int32_t)n, P, batch};
        if (wgs_comm_allreduce_host_tagged(comm, slots.data(), world, &tag, nullptr)) return 1;
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
            if (sweep_ms < 0) sweep_ms = 0.0;                // (not known while the codes' memory is being allocated)
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
    return wgs_comm_check(comm);
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

}   // extern "C"
