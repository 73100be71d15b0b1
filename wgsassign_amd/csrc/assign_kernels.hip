// Assignment log-likelihood (glassy_cy.pyx:12-21 summed as glassy.py:31-42) as CDNA4 kernels.
// Compiled with -ffp-contract=off (see em_kernels.hip).
//
// The reference scans L once per (individual, population) pair: n*K strided passes.  Here ONE
// sweep over a population slab produces every pair:
//   lane <-> PAIR of individuals (slab columns 2p, 2p+1); wave <-> 64 pairs x a range of tiles.
//   In the tile-interleaved slab a lane's (g0,g1,g0',g1') for consecutive SNPs of a tile are
//   consecutive 16-byte words (one 128-byte line per 8 SNPs), so each lane streams its own
//   lines.  The kernel is bound by float64 arithmetic (one double log per (SNP, individual,
//   population)), not by these loads.
//   Each lane keeps 2 x KB float64 accumulators (np.sum(..., dtype=float), glassy.py:38); the
//   per-SNP frequency of population k is a broadcast load, or a per-lane vector when a
//   per-individual column table is given (leave-one-out).
#include <type_traits>

#include "common.h"
#include "log_table.h"

namespace {

// ---- double-precision log of a float32 argument -------------------------------------------
// The reference calls libm's double log on (double)(float) values and stores the result as
// float32 (glassy_cy.pyx:21).  A general double log (ocml: ~80 VALU instructions, double-double
// arithmetic) bounds the assignment kernel; this one exploits that the argument has only 24
// significant bits: table-driven range reduction with an EXACT reduced argument, degree-8
// Taylor polynomial, compensated reconstruction -- 16 float64 instructions, error < 1 ulp of
// double, i.e. the float32-rounded result differs from a correctly rounded log's only when the
// true value lies within ~2^-29 relative of a float32 rounding boundary (tests/test_gpu_log.py
// counts the cases exhaustively over every positive float32).
__device__ double2 wgs_log_table_dev[WGS_LOG_N];
constexpr int WGS_LOG_REP = 16;   // LDS copies of the table (see load_log_table)

// c8 = -0.125, the leading polynomial coefficient, handed in by the hot loops from a VGPR pair they keep alive
// (log_c8()): the first Horner step then is ONE v_fma_f64 with a register and a scalar constant, instead of a
// v_mov_b64 of 1/7 into the accumulator of a v_fmac_f64 for every term (two non-inline constants do not fit one
// instruction's constant bus).
__device__ __forceinline__ double log_c8()
{
    double c = -0.125;
    asm volatile("" : "+v"(c));
    return c;
}

template <int REP = WGS_LOG_REP>
__device__ __forceinline__ double log_f32arg(double x, const double2 *tab, double c8 = -0.125)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const unsigned int hi = (unsigned int)(bits >> 32), lo = (unsigned int)bits;
    const unsigned int tmp = hi - WGS_LOG_OFF;
    const int k = (int)tmp >> 20;                          // x = z * 2^k, z in [0.6875, 1.375)
    const unsigned int i = (tmp >> 13) & (WGS_LOG_N - 1);
    const double z = __hiloint2double((int)(hi - (tmp & 0xFFF00000u)), (int)lo);
    const double2 t = tab[i * REP];                        // {invc, logc}; tab already points at this lane's copy
    const double r = __builtin_fma(z, t.x, -1.0);          // exact
    const double kd = (double)k;
    const double w = __builtin_fma(kd, WGS_LN2HI, t.y);    // kd*Ln2hi is exact
    const double hi_ = w + r;
    const double lo_ = __builtin_fma(kd, WGS_LN2LO, (w - hi_) + r);
    double q = __builtin_fma(r, c8, 1.0 / 7.0);            // log1p(r) = r + r^2 * q(r)
    q = __builtin_fma(r, q, -1.0 / 6.0);
    q = __builtin_fma(r, q, 0.2);
    q = __builtin_fma(r, q, -0.25);
    q = __builtin_fma(r, q, 1.0 / 3.0);
    q = __builtin_fma(r, q, -0.5);
    return __builtin_fma(r * r, q, lo_) + hi_;
}

// (float)log((double)s) for any float32 s, including the special values libm defines:
// log(+-0) = -inf, log(+inf) = +inf, log(negative) = log(NaN) = NaN.  The hardware's float32
// log2 returns exactly those for exactly those arguments, so one v_log_f32 supplies every
// special value and one class test selects it -- no branches in the per-term code.
__device__ __forceinline__ float logf_of_f32(float s, const double2 *tab)
{
    const float v = (float)log_f32arg((double)s, tab);
    const float special = __builtin_amdgcn_logf(s);
    return __builtin_isfpclass(s, 0x0100 | 0x0080) ? v : special;   // +normal | +subnormal
}

// LDS image of the table: WGS_LOG_REP = 16 interleaved copies, entry i of copy c at [i*16 + c].
// A ds_read_b128 is serviced in groups of 16 lanes whose (lane & 15) are all distinct; with lane l
// reading copy (l & 15) every lane of a group hits its own 16-byte slot of the 256-byte bank row,
// whatever its index i: the data-dependent lookup is bank-conflict free (a single copy measured
// 61 % of LDS cycles lost to conflicts).
__device__ __forceinline__ const double2 *load_log_table(double2 *tab)
{
    for (int e = threadIdx.x; e < WGS_LOG_N * WGS_LOG_REP; e += blockDim.x) tab[e] = wgs_log_table_dev[e / WGS_LOG_REP];
    __syncthreads();
    return tab + (threadIdx.x & (WGS_LOG_REP - 1));
}

// glassy_cy.pyx:18-21 for one (SNP, individual, population), exact rounding sequence; returns
// the float32 the reference stores into loglike_vec[s] (which starts at 0.0f, glassy.py:34).
__device__ __forceinline__ float site_ll_exact(double g0d, double g1d, double g2d, float a, const double2 *tab)
{
    const double ad = (double)a;
    const double oma = 1.0 - ad;
    const float like0 = (float)((g0d * oma) * oma);
    const float like1 = (float)(((g1d * 2.0) * oma) * ad);
    const float like2 = (float)((g2d * ad) * ad);
    return logf_of_f32((like0 + like1) + like2, tab);
}

__device__ __forceinline__ float site_ll_fast(float g0, float g1, float g2, float a)
{
    const float oma = 1.0f - a;
    const float like0 = g0 * oma * oma;
    const float like1 = g1 * 2.0f * oma * a;
    const float like2 = g2 * a * a;
    return __builtin_amdgcn_logf((like0 + like1) + like2) * 0.69314718055994530942f;   // v_log_f32 (log2) * ln 2
}

typedef const float __attribute__((address_space(1))) *gf32_ptr;

template <int KB, int MODE>
__global__ __launch_bounds__(256) void assign_kernel(AssignArgs A)
{
    __shared__ double2 tab_lds[WGS_LOG_N * WGS_LOG_REP];
    const double2 *tab = load_log_table(tab_lds);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pair = blockIdx.y * 64 + lane;
    const bool valid_a = 2 * pair < A.ncols, valid_b = 2 * pair + 1 < A.ncols;
    const int pairc = pair < A.npairs ? pair : A.npairs - 1;
    const int ind_a = A.members[valid_a ? 2 * pair : 0];
    const int ind_b = A.members[valid_b ? 2 * pair + 1 : 0];
    const int64_t ntiles = (A.m + 63) >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    const int64_t t0 = w * A.tiles_per_wave;
    int64_t t1 = t0 + A.tiles_per_wave;
    if (t1 > ntiles) t1 = ntiles;
    if (t0 >= t1) return;
    const int64_t s_begin = t0 << 6;
    const int64_t s_end = (t1 << 6) < A.m ? (t1 << 6) : A.m;
    const float4 *base = A.slab + (int64_t)pairc * 64;     // + tile * npairs * 64 + lane-in-tile

    for (int kb = 0; kb < A.K; kb += KB) {
        gf32_ptr pa[KB], pb[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int k = kb + j < A.K ? kb + j : A.K - 1;
            pa[j] = (gf32_ptr)(A.colptr ? A.colptr[(int64_t)ind_a * A.K + k] : A.acol[k]);
            pb[j] = (gf32_ptr)(A.colptr ? A.colptr[(int64_t)ind_b * A.K + k] : A.acol[k]);
        }
        // utils.py:147-149: partition label = global site index % P (P == 1: one pass over all sites).
        for (int p = 0; p < A.P; ++p) {
            double acc_a[KB], acc_b[KB];
#pragma unroll
            for (int j = 0; j < KB; ++j) acc_a[j] = 0.0, acc_b[j] = 0.0;
            const int64_t first = A.P == 1 ? s_begin : s_begin + ((p - (A.site0 + s_begin) % A.P) % A.P + A.P) % A.P;
            for (int64_t s = first; s < s_end; s += A.P) {
                const float4 g = base[((s >> 6) * A.npairs << 6) + (s & 63)];
                const double a0 = (double)g.x, a1 = (double)g.y, a2 = (1.0 - a0) - a1;
                const double b0 = (double)g.z, b1 = (double)g.w, b2 = (1.0 - b0) - b1;
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    if (kb + j < A.K) {
                        const float fa = pa[j][s], fb = pb[j][s];
                        float va, vb;
                        if (MODE == WGS_MODE_EXACT) {
                            va = site_ll_exact(a0, a1, a2, fa, tab);
                            vb = site_ll_exact(b0, b1, b2, fb, tab);
                        } else {
                            va = site_ll_fast(g.x, g.y, (1.0f - g.x) - g.y, fa);
                            vb = site_ll_fast(g.z, g.w, (1.0f - g.z) - g.w, fb);
                        }
                        acc_a[j] += (double)va;
                        acc_b[j] += (double)vb;
                    }
                }
            }
            if (first < s_end) {
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    if (kb + j < A.K) {
                        if (valid_a) atomicAdd(&A.out[((int64_t)ind_a * A.P + p) * A.K + kb + j], acc_a[j]);
                        if (valid_b) atomicAdd(&A.out[((int64_t)ind_b * A.P + p) * A.K + kb + j], acc_b[j]);
                    }
                }
            }
        }
    }
}

// ---- scoring sweep: lane <-> SNP, wave <-> (NP pairs of slab columns, one block of tiles) ---------
// GL loads are the slab's native 1 KiB wave loads, the K frequencies of a SNP are loaded once per
// tile and their double forms (a, 1-a) hoisted over the individuals (shared-A mode), and in
// leave-one-out mode the per-individual frequency vectors are read coalesced (lane = SNP).  Each lane
// keeps NP x 2 x KB float64 partial sums over the tiles of its block; one cross-lane reduction per
// (individual, population) ends the block and its result is STORED (one writer per element).  Pair
// groups of ALL population slabs share one launch, pair-group index fastest over waves, so the waves
// that need the same tile's frequencies run together (L2 hits) and the launch has one tail, not K.

// (like0 + like1) + like2 of glassy_cy.pyx:18-20 in the reference's rounding sequence.
__device__ __forceinline__ float like_sum_exact(double g0d, double g1d2, double g2d, double ad, double oma)
{
    const float like0 = (float)((g0d * oma) * oma);
    const float like1 = (float)((g1d2 * oma) * ad);        // ((g1*2.0)*(1-a))*a, g1*2.0 exact
    const float like2 = (float)((g2d * ad) * ad);
    return (like0 + like1) + like2;
}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

constexpr unsigned FP_POS_FINITE = 0x0100 | 0x0080;        // +normal | +subnormal: the arguments log_f32arg handles

// What one wavefront works on: decoded once, wave-uniform.
template <int NP>
struct WaveWork {
    ScoreSlab sl;
    int64_t blk, t0, t1;
    int pg;                        // pair group within the slab
    int ind[NP][2];
    bool ok[NP][2];
    int pairc[NP];
};

template <int NP>
__device__ __forceinline__ bool decode_wave(const ScoreArgs &A, WaveWork<NP> &w)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int pgg = (int)(wid % A.total_pg);
    w.blk = wid / A.total_pg;
    if (w.blk >= A.nblocks) return false;
    int g = 0;
    while (g + 1 < A.n_slabs && pgg >= A.slabs[g + 1].pg0) ++g;
    w.sl = A.slabs[g];
    const int pg = w.pg = pgg - w.sl.pg0;
    const int64_t ntiles = (A.m + 63) >> 6;
    w.t0 = w.blk * WGS_BLOCK_TILES;
    w.t1 = w.t0 + WGS_BLOCK_TILES < ntiles ? w.t0 + WGS_BLOCK_TILES : ntiles;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pair = w.sl.pair0 + pg * NP + q;
        w.pairc[q] = pair < w.sl.npairs ? pair : w.sl.npairs - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = 2 * pair + h;
            w.ok[q][h] = col >= w.sl.col_lo && col < w.sl.col_hi;
            w.ind[q][h] = w.sl.members[w.ok[q][h] ? col : w.sl.col_lo];
        }
    }
    return true;
}

// Register budget: 3 waves per SIMD (<= 168 VGPRs) where the instantiation fits it WITHOUT spilling -- shared
// columns with 8 (exact) or 10 (float32) accumulator pairs per lane, except KB = 8 whose ten hoisted frequencies in
// double push it over --, 2 (<= 256) otherwise and with the per-individual pointer and frequency tables of
// leave-one-out.  tools/kernel_resources.py prints what the compiler made of every instantiation
// (profiles/r03_kernel_resources.txt): none spills.
constexpr int sweep_waves_per_simd(int KB, int NP, int MODE, bool PER_IND)
{
    if (PER_IND) return 2;
    if (MODE == WGS_MODE_EXACT) return (KB * NP <= 8 && KB != 8) ? 3 : 2;
    return (KB * NP > 10 || KB > 8) ? 2 : 3;
}
// Pairs of slab columns per wave: two halve the per-tile frequency loads and conversions per term (measured
// 174 -> 150 ms at K = 10 as two passes of 5) while the accumulators leave room for it.
constexpr int sweep_pairs(int KB, bool PER_IND) { return KB <= (PER_IND ? 4 : 6) ? 2 : 1; }
constexpr int chain_pairs(int KB, bool PER_IND) { return !PER_IND && KB <= 4 ? 2 : 1; }

template <int KB, int NP, int MODE, bool PER_IND>
__global__ __launch_bounds__(256, sweep_waves_per_simd(KB, NP, MODE, PER_IND)) void score_sweep_kernel(ScoreArgs A)
{
    __shared__ double2 tab_lds[WGS_LOG_N * WGS_LOG_REP];
    const double2 *tab = load_log_table(tab_lds);
    const int lane = threadIdx.x & 63;
    WaveWork<NP> w;
    if (!decode_wave<NP>(A, w)) return;
    const float4 *slab = w.sl.slab;
    const int npairs = w.sl.npairs;
    const int64_t t0 = w.t0, t1 = w.t1;
    const double c8 = log_c8();

    for (int kb = 0; kb < A.K; kb += KB) {
        constexpr int NA = PER_IND ? NP * 2 * KB : KB;
        gf32_ptr ptr[NA];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int k = kb + j < A.K ? kb + j : A.K - 1;
            if (PER_IND) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) ptr[(q * 2 + h) * KB + j] = (gf32_ptr)A.colptr[(int64_t)w.ind[q][h] * A.K + k];
            } else {
                ptr[j] = (gf32_ptr)A.acol[k];
            }
        }
        double acc[NP][2][KB];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < KB; ++j) acc[q][h][j] = 0.0;

        // One tile ahead: the GLs and frequencies of tile t+1 are requested before tile t is consumed
        // (only ~3 waves share a SIMD, so nothing else hides the load latency).
        float4 g_cur[NP], g_nxt[NP];
        float a_cur[NA], a_nxt[NA];
        auto fetch = [&](int64_t t, float4 *gq, float *aq) {
            const int64_t s = (t << 6) + lane;
            const int64_t sc = s < A.m ? s : A.m - 1;          // clamped index; dead lanes are neutralised below
#pragma unroll
            for (int q = 0; q < NP; ++q) gq[q] = slab[((t * npairs + w.pairc[q]) << 6) + lane];
#pragma unroll
            for (int x = 0; x < NA; ++x) aq[x] = ptr[x][sc];
        };
        // One tile.  MASKED (only the last, partial tile of the matrix): lanes past the last SNP are given
        // g = (1, 0) and a = 0, for which the site likelihood is exactly 1 and its log exactly 0.
        // FIX = false is the hot path: log_f32arg on every sum, whatever it is; the sums that are not
        // positive finite numbers (0 -> -inf, negative or NaN -> NaN; +inf cannot arise) are noted in
        // `plain` and the tile is visited again with FIX = true, which adds libm's special value for
        // exactly those terms.  That is exact: log_f32arg returns a FINITE value for +-0 and +inf, and
        // finite + (+-inf or NaN) is the same whatever the finite part was; for negative or NaN sums
        // the result must be NaN and the special value added is NaN.
        auto tile = [&](int64_t t, auto masked_tag, auto fix_tag) -> bool {
            constexpr bool MASKED = decltype(masked_tag)::value, FIX = decltype(fix_tag)::value;
            const bool live = !MASKED || ((t << 6) + lane) < A.m;
            bool plain = true;
            double ad[KB], oma[KB];
            float af[KB];
            if (!PER_IND) {
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    af[j] = live ? a_cur[j] : 0.0f;
                    ad[j] = (double)af[j];
                    oma[j] = 1.0 - ad[j];
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const float4 g = g_cur[q];
                const float gl[2][2] = {{live ? g.x : 1.0f, live ? g.y : 0.0f}, {live ? g.z : 1.0f, live ? g.w : 0.0f}};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double g0d = (double)gl[h][0], g1d = (double)gl[h][1];
                    const double g1d2 = g1d * 2.0, g2d = (1.0 - g0d) - g1d;
                    const float g2f = (1.0f - gl[h][0]) - gl[h][1];
                    // all KB slots are computed (slots past K repeat population K-1 and are dropped in the
                    // epilogue): the tile body stays one basic block the scheduler can interleave
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        const float a = PER_IND ? (live ? a_cur[(q * 2 + h) * KB + j] : 0.0f) : af[j];
                        if (MODE == WGS_MODE_EXACT) {
                            const double a_d = PER_IND ? (double)a : ad[j];
                            const float s = like_sum_exact(g0d, g1d2, g2d, a_d, PER_IND ? 1.0 - a_d : oma[j]);
                            const bool fin = __builtin_isfpclass(s, FP_POS_FINITE);
                            if (FIX) {
                                if (!fin) acc[q][h][j] += (double)__builtin_amdgcn_logf(s);
                            } else {
                                plain = plain && fin;
                                acc[q][h][j] += (double)(float)log_f32arg((double)s, tab, c8);
                            }
                        } else {
                            acc[q][h][j] += (double)site_ll_fast(gl[h][0], gl[h][1], g2f, a);
                        }
                    }
                }
            }
            return plain;
        };
        using T = std::true_type;
        using F = std::false_type;
        fetch(t0, g_cur, a_cur);
        for (int64_t t = t0; t < t1; ++t) {
            if (t + 1 < t1) fetch(t + 1, g_nxt, a_nxt);
            const bool partial = ((t + 1) << 6) > A.m;
            const bool plain = partial ? tile(t, T{}, F{}) : tile(t, F{}, F{});
            if (MODE == WGS_MODE_EXACT && !__all(plain)) {       // rare: a likelihood sum of exactly 0, or NaN data
                if (partial) tile(t, T{}, T{}); else tile(t, F{}, T{});
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) g_cur[q] = g_nxt[q];
#pragma unroll
            for (int x = 0; x < NA; ++x) a_cur[x] = a_nxt[x];
        }
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    if (kb + j < A.K) {
                        const double tot = wave_sum(acc[q][h][j]);
                        if (lane == 0 && w.ok[q][h]) A.S[w.blk * A.cells + (int64_t)w.ind[q][h] * A.K + kb + j] = tot;
                    }
                }
    }
}

// ---- scoring through the class codes (common.h: wgs_codes) ---------------------------------------------------
// The per-site value (float)log(like0 + like1 + like2) depends on the individual only through its (g0, g1), and a SNP has
// few distinct (g0, g1): 27 on average among 1000 individuals at 2x.  So the values are computed ONCE per (SNP, class,
// population) into an LDS table and the individuals only look them up:
//   workgroup <-> one block of 4096 SNPs x up to 1024 individuals; lane <-> QUAD of individuals (4 x KB float64 sums in
//   registers, no cross-lane reduction at all); SNPs are taken 16 at a time -- the 16 code words of a quad are one
//   64-byte line:
//     phase 1  all threads fill vtab[snp in batch][class][population] (the reference's rounding sequence and the
//              table log of the direct sweep, special values through the hardware log as in logf_of_f32);
//     phase 2  every lane adds vtab[snp][code][.] of its four individuals to its sums.
// The per-site float32 values are the direct sweep's bit for bit and a block's float64 partial sums are exact in any
// order, so S[block][cell] -- and everything built on it: NumPy-order totals, chain predictions -- is unchanged.
// Cost per (SNP, individual, population): a quarter of an LDS read, one conversion, one add (41.6 instructions in the
// direct sweep) plus classes/individuals of the table work.
struct CodedSlab {
    const uint32_t *codes;
    const int32_t *members;
    const float4 *slab;            // the float32 slab, for the SNPs the encoder left uncoded (ncls = 0: too many classes)
    int32_t nquads, ncols, quad0, col_lo, col_hi, npairs;
};
struct CodedScoreArgs {
    const float2 *dict;
    const uint8_t *ncls;
    const CodedSlab *slabs;
    int32_t n_slabs, drows, total_quads;
    const float *const *acol;
    int64_t m, cells;
    int32_t K, nblocks;
    int32_t table_rows;            // rows of the table in LDS (wgs_codes::rows_batch: the richest aligned batch of SNPs of the matrix)
    int32_t parts;                 // a block's 64 tiles are shared by `parts` workgroups (short matrices: enough workgroups
                                   // to fill the chip); their partial sums go to Sp[part][block][cell] and are added in a
                                   // fixed order by combine_parts_kernel
    double *S;
    const double2 *logtab;         // the log table in device memory (wgs_log_table_dev)
};
constexpr int CODED_BATCH_MAX = 16; // SNPs per table: the code words of 16 SNPs of one quad are one 64-byte line; matrices with many
                                    // classes per SNP take 8 or 4 at a time (wgs_codes::score_batch) so that a batch's rows fit the table
constexpr int CODED_LOG_REP = 2;   // LDS copies of the log table here (its reads are a twentieth of the kernel's LDS traffic: 4 KiB instead of 32)

// Details of the table:
//   * it holds only the classes a SNP HAS: row of (SNP j of the batch, class c) = rowoff[j] + c, rowoff = running sum of
//     ncls over the batch's 16 SNPs (LDS sized for the richest aligned 16-SNP group of the matrix, wgs_codes::rows16; a
//     first version walked cmax rows per SNP and left a third of its phase-1 lanes idle: 20.3 ms -> 15.9 at 10M x 1000 x 10);
//   * a phase-1 work item is (row, half of the populations): the dictionary entry is fetched and widened once per item,
//     the frequencies of the batch come from LDS (staged for the NEXT batch while this one is computed, together with
//     the row offsets: two small buffers, no extra barrier);
//   * TV = double: the table holds the widened values and phase 2 is an LDS read and an add per term, no conversion
//     (-> 14.2 ms); TV = float where the float rows need no padding to 16 bytes (KB = 4, 8; 7 pads one): half the LDS
//     traffic wins there.  WGS_SCORE_CODED_TABLE=float|double forces one (experiments).
struct CodedPrep {
    int rowoff[20];                // [j] = rows before SNP j of the batch, [batch size] = rows of the batch, [17] = bit j: SNP j of the batch is uncoded
    float aval[CODED_BATCH_MAX][10];   // allele frequencies [SNP of the batch][population of this pass]
};

// Float64 number X .. HI-1 of an LDS row as ds_read_b64 (byte address `addr` of the row), and the point behind a wait at which they have
// arrived; v[0] is number LO
template <int HI, int LO, int X = LO>
struct CodedRow {
    static __device__ __forceinline__ void issue(double (&v)[HI - LO], unsigned addr)
    {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[X - LO]) : "v"(addr), "n"(8 * X));
        CodedRow<HI, LO, X + 1>::issue(v, addr);
    }
    static __device__ __forceinline__ void arrived(double (&v)[HI - LO])
    {
        asm volatile("" : "+v"(v[X - LO]));
        CodedRow<HI, LO, X + 1>::arrived(v);
    }
};
template <int HI, int LO>
struct CodedRow<HI, LO, HI> {
    static __device__ __forceinline__ void issue(double (&)[HI - LO], unsigned) {}
    static __device__ __forceinline__ void arrived(double (&)[HI - LO]) {}
};
template <int N>
__device__ __forceinline__ void lds_wait()
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int CODED_ROWS_MAX = (WGS_BATCH_ROWS_CAP + 255) / 256;     // table rows whose dictionary entries a thread requests per batch

#ifdef WGS_SCORE_STATS
// (experiments: -DWGS_SCORE_STATS adds up, per launch of score_coded_kernel and over all wavefronts, the clock cycles [0] from a batch's
// start to its phase 1, [1] in phase 1, [2] at the barrier behind it, [3] in phase 2, [4] at the barrier behind it; [5] batches,
// [6] wavefronts; printed by launch_score_coded.)
__device__ unsigned long long g_score_stats[8];
#define SCORE_CLOCK(i) do { const unsigned long long now_ = clock64(); stat_[i] += now_ - mark_; mark_ = now_; } while (0)
#else
#define SCORE_CLOCK(i) do { } while (0)
#endif

// the SNP of the batch that table row r belongs to: the largest j with rowoff[j] <= r (rowoff is the running sum of the SNPs' classes)
__device__ __forceinline__ int item_snp(const CodedPrep &P, int r, int batch)
{
    int j = 0;
    for (int step = batch / 2; step >= 1; step >>= 1) j += P.rowoff[j + step] <= r ? step : 0;
    return j;
}

template <int KB, int MODE, typename TV, int CODED_BATCH>
#ifndef WGS_SCORE_CODED_WAVES
#define WGS_SCORE_CODED_WAVES 3
#endif
__global__ __launch_bounds__(256, WGS_SCORE_CODED_WAVES) void score_coded_kernel(CodedScoreArgs A)
{
    static_assert(CODED_BATCH == 16 || CODED_BATCH == 8 || CODED_BATCH == 4, "SNPs per table");
    // Row stride of the table.  Phase 2 reads rows by class id -- data-dependent addresses -- and the LDS serves a wide read in lane
    // groups whose lanes must fall on different banks: with 16-byte reads of 80-byte rows (round 4) 16 lanes share 16 bank windows
    // that the class id selects modulo 16, so a rare class (id >= 16) collides with a frequent one in most groups (SQ_LDS_BANK_CONFLICT
    // was 36 % of the LDS-array cycles, 23 % once the encoder numbered the classes by first appearance).  Float64 rows are therefore
    // an ODD number of 8-byte words and read 8 bytes at a time: 32 lanes per group on 32 two-bank windows that the id selects modulo
    // 32 -- conflict-free up to 32 classes per SNP, the same LDS-array cycles per row.  (Float rows keep 16-byte reads.)
    constexpr int KBP = sizeof(TV) == 8 ? (KB | 1) : ((KB + 3) & ~3);
    constexpr int KG = (KB + 1) / 2;                                                // populations per phase-1 item
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double2 *tab_lds = reinterpret_cast<double2 *>(lds_raw);
    CodedPrep *prep = reinterpret_cast<CodedPrep *>(lds_raw + sizeof(double2) * WGS_LOG_N * CODED_LOG_REP);
    float2 *stage = reinterpret_cast<float2 *>(reinterpret_cast<unsigned char *>(prep) + ((2 * sizeof(CodedPrep) + 15) & ~(size_t)15));   // [A.table_rows]
    TV *vtab = reinterpret_cast<TV *>(stage + ((A.table_rows + 1) & ~1));
    if (MODE == WGS_MODE_EXACT) {
        for (int e = threadIdx.x; e < WGS_LOG_N * CODED_LOG_REP; e += blockDim.x) tab_lds[e] = A.logtab[e / CODED_LOG_REP];
    }
    const double2 *tab = tab_lds + (threadIdx.x & (CODED_LOG_REP - 1));
    if (sizeof(TV) == 8 && threadIdx.x < KBP) vtab[(size_t)A.table_rows * KBP + threadIdx.x] = (TV)0;     // the row of zeros behind the table (phase 2)
    const int tid = threadIdx.x;
    const int64_t blk = blockIdx.x;
    const int64_t ntiles = (A.m + 63) >> 6;
    // (a block's tiles in `parts` runs of ceil(64 / parts) tiles, the last one shorter: any count up to 16, not only the powers of two)
    const int tiles_per_part = (WGS_BLOCK_TILES + A.parts - 1) / A.parts;
    const int64_t t0 = blk * WGS_BLOCK_TILES + (int64_t)blockIdx.z * tiles_per_part;
    const int64_t block_end = (blk + 1) * WGS_BLOCK_TILES < ntiles ? (blk + 1) * WGS_BLOCK_TILES : ntiles;
    const int64_t t1 = t0 + tiles_per_part < block_end ? t0 + tiles_per_part : block_end;
    double *const Sout = A.S + ((int64_t)blockIdx.z * A.nblocks + blk) * A.cells;
    const int Q = (int)blockIdx.y * 256 + tid;
    const bool have = Q < A.total_quads;
    auto slab_of = [&]() -> int {                          // (found again where it is needed rather than kept: see below)
        int g = 0;
        if (have)
            while (g + 1 < A.n_slabs && Q >= A.slabs[g + 1].quad0) ++g;
        return g;
    };
    const int g = slab_of();
    // (Of the lane's slab only the address of its code words and their stride per tile stay in registers through the batches; the
    // record itself is read again where it is needed -- for a SNP the encoder left uncoded, and for the sums' destinations at the end.
    // Held throughout, its twelve registers and the four member indices were spilled, and a reload from scratch behind the batch's
    // loads waits for those loads: vector memory returns in order.)
    const uint4 *cptr = reinterpret_cast<const uint4 *>(A.slabs[g].codes + (int64_t)(have ? Q - A.slabs[g].quad0 : 0) * 64);
    const int tile_words = A.slabs[g].nquads * 16;                                    // uint4 per tile of this slab's code words
    const int64_t s_begin = t0 << 6;
    const int64_t s_end = (t1 << 6) < A.m ? (t1 << 6) : A.m;
    const int nbatch = s_end > s_begin ? (int)((s_end - s_begin + CODED_BATCH - 1) / CODED_BATCH) : 0;

    for (int kb = 0; kb < A.K; kb += KB) {
        // what phase 1 needs of batch b, written by threads 0..15 (row offsets) and 16.. (frequencies)
        // (in two steps: the loads are issued at the start of a batch, their values go to LDS behind its phase 1 -- written where they
        // are loaded they cost every wavefront a trip to memory per batch)
        // (Round 5, late: every thread requests ONE byte and ONE float per batch from addresses of its own -- the class count of "its"
        // SNP, the frequency of "its" (SNP, population) -- whether or not it has a use for them, through pointers typed as GLOBAL: a
        // load inside a conditional block is waited for where the block ends (the compiler merges the two paths' registers there), a
        // flat load counts as an LDS operation too (phase 1's LDS reads then wait for it), and the column pointer A.acol[k], fetched
        // from device memory per batch, made it two dependent trips to memory -- together a sixth of the kernel's time, spent between
        // a batch's start and its phase 1 (-DWGS_SCORE_STATS).  Not inline assembly: the compiler must know which registers have
        // loads in flight, or it moves them -- a version that requested the code words by asm gave wrong sums at some launch shapes.)
        typedef const __attribute__((address_space(1))) uint8_t *gl_u8_t;
        typedef const __attribute__((address_space(1))) float *gl_f32_t;
        const int pe = tid >= 64 && tid - 64 < CODED_BATCH * KB ? tid - 64 : 0, pj = pe / KB, pk = pe - pj * KB;
        const gl_f32_t my_col = (gl_f32_t)A.acol[kb + pk < A.K ? kb + pk : A.K - 1];
        const gl_u8_t all_ncls = (gl_u8_t)A.ncls;
        auto prepare_load = [&](int b, int &got_ncls, float &got_a) {
            const int64_t s0 = s_begin + (int64_t)b * CODED_BATCH;
            const int64_t sn = s0 + (tid & (CODED_BATCH - 1)) < s_end ? s0 + (tid & (CODED_BATCH - 1)) : s_end - 1;
            const int64_t sa = s0 + pj < s_end ? s0 + pj : s_end - 1;
            got_ncls = all_ncls[sn];
            got_a = my_col[sa];
        };
        // what the two values mean for this thread (after a wait for them: vmcnt)
        auto prepare_value = [&](int b, int got_ncls, float got_a) -> int {
            const int64_t s0 = s_begin + (int64_t)b * CODED_BATCH;
            if (tid < CODED_BATCH) return s0 + tid < s_end ? got_ncls : 0;
            if (tid >= 64 && tid - 64 < CODED_BATCH * KB) return __float_as_int(s0 + pj < s_end ? got_a : 0.5f);
            return 0;
        };
        auto prepare_store = [&](int b, int val) {
            CodedPrep &P = prep[b & 1];
            const int64_t s0 = s_begin + (int64_t)b * CODED_BATCH;
            if (tid < 64) {                                    // the first wavefront: a running sum of ncls over the batch's SNPs
                const int j = tid & 15;
                const int n = tid < CODED_BATCH ? val : 0;
                int incl = n;
#pragma unroll
                for (int off = 1; off < CODED_BATCH; off <<= 1) {
                    const int up = __shfl_up(incl, off, 64);
                    if (j >= off) incl += up;
                }
                if (tid < CODED_BATCH) {
                    P.rowoff[j] = incl - n;
                    if (j == CODED_BATCH - 1) P.rowoff[CODED_BATCH] = incl;
                }
                const unsigned long long uncoded = __ballot(tid < CODED_BATCH && s0 + j < s_end && n == 0);
                if (tid == 0) P.rowoff[17] = (int)(uncoded & 0xFFFFu);     // bit j: SNP j of the batch was left uncoded by the encoder
            } else if (tid - 64 < CODED_BATCH * KB) {
                const int e = tid - 64, j = e / KB, k = e - j * KB;
                P.aval[j][k] = __int_as_float(val);
            }
        };
        auto prepare = [&](int b) {
            int got_ncls;
            float got_a;
            prepare_load(b, got_ncls, got_a);
            prepare_store(b, prepare_value(b, got_ncls, got_a));
        };
        double acc[4][KB];
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int k = 0; k < KB; ++k) acc[h][k] = 0.0;
        __syncthreads();                                       // the log table (first pass); the previous pass's last phase 2
        if (nbatch > 0) prepare(0);
        __syncthreads();
        // The dictionary entries of a batch -- one per table row: [(tile * drows + class) * 64 + SNP of the tile] -- are fetched one
        // batch AHEAD: every batch touches a fresh cache line per class row, so a load issued where phase 1 needs it costs a trip to
        // HBM (three to four in a row per batch: 53 % of the wave-cycles of round 4's kernel were spent waiting).  They go straight
        // into LDS (global_load_lds: no registers held across phase 2, whose 80 accumulator registers leave none): wave w moves rows
        // 64 w + lane, 64 w + lane + 256, ... as two 4-byte transfers into the planes stage_x / stage_y, issued right behind the
        // barrier that ends phase 1 (the next batch's row offsets are complete then) and drained by the barrier that ends phase 2.
        typedef __attribute__((address_space(1))) const void *gl_src_t;
        typedef __attribute__((address_space(3))) void *lds_dst_t;
        float *const stage_x = reinterpret_cast<float *>(stage), *const stage_y = stage_x + ((A.table_rows + 1) & ~1);
        auto dma_rows = [&](int b) {
            const CodedPrep &Pn = prep[b & 1];
            const int64_t s0n = s_begin + (int64_t)b * CODED_BATCH;
            const int64_t tn = s0n >> 6;
            const int l0n = (int)(s0n & 63);
            const int rows_n = Pn.rowoff[CODED_BATCH];
            const int wave0 = (int)__builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
            for (int x = 0; x < CODED_ROWS_MAX; ++x) {
                const int r = tid + 256 * x;
                if (r < rows_n) {
                    const int j = item_snp(Pn, r, CODED_BATCH);
                    const float *src = reinterpret_cast<const float *>(A.dict + ((tn * A.drows + (r - Pn.rowoff[j])) * 64 + l0n + j));
                    __builtin_amdgcn_global_load_lds((gl_src_t)src, (lds_dst_t)(stage_x + wave0 + 256 * x), 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((gl_src_t)(src + 1), (lds_dst_t)(stage_y + wave0 + 256 * x), 4, 0, 0);
                }
            }
        };
        if (nbatch > 0) dma_rows(0);
        __syncthreads();
#ifdef WGS_SCORE_STATS
        unsigned long long stat_[8] = {0}, mark_ = clock64();
#endif
        for (int b = 0; b < nbatch; ++b) {
            const CodedPrep &P = prep[b & 1];
            const int64_t s0 = s_begin + (int64_t)b * CODED_BATCH;
            const int64_t t = s0 >> 6;
            const int l0 = (int)(s0 & 63);
            const int nj = s_end - s0 < CODED_BATCH ? (int)(s_end - s0) : CODED_BATCH;
            // the batch's code words (a lane without a quad reads quad 0's: valid classes, sums that nobody stores) and the next batch's
            // class counts and frequencies: requested here, used behind phase 1
            uint4 cw[CODED_BATCH / 4];
            {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const __attribute__((address_space(1))) u32x4 *line = (const __attribute__((address_space(1))) u32x4 *)(cptr + (t * (int64_t)tile_words + l0 / 4));
#pragma unroll
                for (int x = 0; x < CODED_BATCH / 4; ++x) {
                    const u32x4 v = line[x];
                    cw[x] = make_uint4(v.x, v.y, v.z, v.w);
                }
            }
            int next_ncls;
            float next_a;
            prepare_load(b + 1 < nbatch ? b + 1 : b, next_ncls, next_a);      // (unconditionally: a register with a load in flight must not be merged with another value)
            // phase 1: vtab[rowoff[j] + c][k] from the staged dictionary entries
            SCORE_CLOCK(0);
            const int items = 2 * P.rowoff[CODED_BATCH];
            for (int it = tid; it < items; it += 256) {
                const int r = it >> 1, half = it & 1;
                const int j = item_snp(P, r, CODED_BATCH);
                const float2 gl = make_float2(stage_x[r], stage_y[r]);
                const double g0d = (double)gl.x, g1d = (double)gl.y;
                const double g1x2 = g1d * 2.0, g2d = (1.0 - g0d) - g1d;
                const float g2f = (1.0f - gl.x) - gl.y;
#pragma unroll
                for (int kq = 0; kq < KG; ++kq) {
                    const int k = half * KG + kq;
                    if (k < KB) {
                        const float a = P.aval[j][k];
                        float v;
                        if (MODE == WGS_MODE_EXACT) {
                            const double ad = (double)a;
                            const float ssum = like_sum_exact(g0d, g1x2, g2d, ad, 1.0 - ad);
                            v = (float)log_f32arg<CODED_LOG_REP>((double)ssum, tab);
                            // a likelihood of exactly 0, NaN data: libm's special value (v_log_f32 returns exactly -inf / NaN there); a
                            // branch no lane of the wavefront takes on ordinary data, instead of a quarter-rate log and a select per element
                            if (__builtin_expect(!__builtin_isfpclass(ssum, FP_POS_FINITE), 0)) v = __builtin_amdgcn_logf(ssum);
                        } else {
                            v = site_ll_fast(gl.x, gl.y, g2f, a);
                        }
                        vtab[r * KBP + k] = (TV)v;
                    }
                }
            }
            // (what was requested at the batch's start has had phase 1 to arrive)
            if (b + 1 < nbatch) prepare_store(b + 1, prepare_value(b + 1, next_ncls, next_a));
            SCORE_CLOCK(1);
            __syncthreads();
            SCORE_CLOCK(2);
            // (the staged entries of this batch have been consumed; the code words, requested before phase 1, are here -- said explicitly,
            // because with a transfer to LDS in flight the compiler waits for ALL outstanding loads at the next use of a loaded value)
#pragma unroll
            for (int x = 0; x < CODED_BATCH / 4; ++x) asm volatile("" : "+v"(cw[x].x), "+v"(cw[x].y), "+v"(cw[x].z), "+v"(cw[x].w));
            if (b + 1 < nbatch) dma_rows(b + 1);
            // phase 2: look up and add
            const unsigned *cwv = reinterpret_cast<const unsigned *>(cw);
            const unsigned uncoded_snps = (unsigned)__builtin_amdgcn_readfirstlane(P.rowoff[17]);
            if (sizeof(TV) == 8) {
                // float64 rows: KB reads of 8 bytes per (SNP, individual), written as ds_read_b64 by hand -- the compiler pairs adjacent
                // 8-byte LDS loads into ds_read2_b64, which the LDS serves at half the rate and with the 32-bank mapping -- and pipelined by
                // hand: the reads of the next individual are in flight while the current one's values are added (counted lgkmcnt waits;
                // LDS operations complete in order, and anything else that counts on lgkmcnt only makes a counted wait wait longer)
                //
                // Round 5 (late): ONE pipeline over the batch's 4 x CODED_BATCH (SNP, individual) items instead of one per SNP.  The
                // per-SNP version read the SNP's row offset from LDS and waited for it with every earlier read drained, skipped absent /
                // uncoded SNPs by a (uniform) branch, and drained again behind the SNP's fourth individual: two exposed LDS round trips
                // per SNP and wavefront.  Now the row offsets are fetched once per batch into scalar registers, an absent or uncoded
                // SNP reads class 0 of a ROW OF ZEROS behind the table (x + 0.0 = x bit for bit; a sum that starts at +0.0 is never
                // -0.0) so that the sixteen SNPs are the same straight-line code, and the next item's reads are in flight while the
                // current one's values are added all the way through the batch.
                const unsigned vtab_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)reinterpret_cast<unsigned char *>(vtab);
                const unsigned zero_addr = vtab_addr + (unsigned)A.table_rows * (unsigned)(KBP * 8);
                unsigned base[CODED_BATCH], word[CODED_BATCH];
#pragma unroll
                for (int j = 0; j < CODED_BATCH; ++j) {
                    const bool skip = j >= nj || ((uncoded_snps >> j) & 1u);
                    const unsigned off = (unsigned)__builtin_amdgcn_readfirstlane(P.rowoff[j]);
                    base[j] = skip ? zero_addr : vtab_addr + off * (unsigned)(KBP * 8);
                    word[j] = skip ? 0u : cwv[j];
                }
                auto row_addr = [&](int i) -> unsigned { return base[i >> 2] + ((word[i >> 2] >> (8 * (i & 3))) & 255u) * (unsigned)(KBP * 8); };
                double va[KB], vb[KB];
                CodedRow<KB, 0>::issue(va, row_addr(0));
#pragma unroll
                for (int i = 0; i < 4 * CODED_BATCH; i += 2) {
                    CodedRow<KB, 0>::issue(vb, row_addr(i + 1));
                    lds_wait<KB>();
                    CodedRow<KB, 0>::arrived(va);
#pragma unroll
                    for (int k = 0; k < KB; ++k) acc[i & 3][k] += va[k];
                    if (i + 2 < 4 * CODED_BATCH) {
                        CodedRow<KB, 0>::issue(va, row_addr(i + 2));
                        lds_wait<KB>();
                    } else {
                        lds_wait<0>();
                    }
                    CodedRow<KB, 0>::arrived(vb);
#pragma unroll
                    for (int k = 0; k < KB; ++k) acc[(i + 1) & 3][k] += vb[k];
                }
            } else {
#pragma unroll
                for (int j = 0; j < CODED_BATCH; ++j) {
                    if (j < nj && !((uncoded_snps >> j) & 1u)) {
                        const unsigned w = cwv[j];
                        const TV *rows_j = vtab + P.rowoff[j] * KBP;
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            const int code = (w >> (8 * h)) & 255;
                            TV vals[KBP];
                            const float4 *row = reinterpret_cast<const float4 *>(rows_j + code * KBP);
#pragma unroll
                            for (int x = 0; x < KBP / 4; ++x) {
                                const float4 f = row[x];
                                vals[4 * x] = (TV)f.x, vals[4 * x + 1] = (TV)f.y, vals[4 * x + 2] = (TV)f.z, vals[4 * x + 3] = (TV)f.w;
                            }
#pragma unroll
                            for (int k = 0; k < KB; ++k) acc[h][k] += (double)vals[k];
                        }
                    }
                }
            }
            // SNPs the encoder left uncoded (more classes than its tables hold; ncls = 0): their terms straight from the float32
            // slab with the direct sweep's arithmetic -- the same float32 per-site values, added to the same float64 sums
#pragma unroll 1
            for (int j = 0; j < (uncoded_snps ? nj : 0); ++j) {
                if ((uncoded_snps >> j) & 1u) {
                    if (have) {
                        const int gu = slab_of();
                        const CodedSlab sl = A.slabs[gu];
                        const int q = Q - sl.quad0;
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) {
                            const int pair = 2 * q + pr < sl.npairs ? 2 * q + pr : sl.npairs - 1;
                            const float4 g = sl.slab[(t * sl.npairs + pair) * 64 + l0 + j];
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) {
                                const float g0 = hh ? g.z : g.x, g1 = hh ? g.w : g.y;
                                const double g0d = (double)g0, g1d = (double)g1;
                                const double g1x2 = g1d * 2.0, g2d = (1.0 - g0d) - g1d;
                                const float g2f = (1.0f - g0) - g1;
#pragma unroll
                                for (int k = 0; k < KB; ++k) {
                                    const float a = P.aval[j][k];
                                    float v;
                                    if (MODE == WGS_MODE_EXACT) {
                                        const double ad = (double)a;
                                        const float ssum = like_sum_exact(g0d, g1x2, g2d, ad, 1.0 - ad);
                                        const float plain = (float)log_f32arg<CODED_LOG_REP>((double)ssum, tab);
                                        v = __builtin_isfpclass(ssum, FP_POS_FINITE) ? plain : __builtin_amdgcn_logf(ssum);
                                    } else {
                                        v = site_ll_fast(g0, g1, g2f, a);
                                    }
                                    acc[2 * pr + hh][k] += (double)v;
                                    // (one evaluation after the other: interleaved, this rare path's forty evaluations set the
                                    // kernel's register need and made the common path spill)
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            }
                        }
                    }
                }
            }
            SCORE_CLOCK(3);
            __syncthreads();                                   // (also drains the transfers into the stage: vmcnt(0))
            SCORE_CLOCK(4);
        }
#ifdef WGS_SCORE_STATS
        if ((tid & 63) == 0) {
            for (int i = 0; i < 5; ++i) atomicAdd(&g_score_stats[i], stat_[i]);
            atomicAdd(&g_score_stats[5], (unsigned long long)nbatch);
            atomicAdd(&g_score_stats[6], 1ull);
        }
#endif
        {
            const CodedSlab sl = A.slabs[slab_of()];
            const int qs = have ? Q - sl.quad0 : 0;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int col = 4 * qs + h;
                const bool ok = have && col >= sl.col_lo && col < sl.col_hi;
                const int ind = sl.members[ok ? col : sl.col_lo];
#pragma unroll
                for (int k = 0; k < KB; ++k)
                    if (ok && kb + k < A.K) Sout[(int64_t)ind * A.K + kb + k] = acc[h][k];
            }
        }
    }
}

// out[cell] = sum over blocks of S[block][cell] in a fixed order (reproducible) -- the order of np.sum(vec, dtype=float)
// (glassy.py:38): NumPy's reduction hands its inner loop 8192 elements at a time, sums each such chunk pairwise (the top
// split of a full chunk is 4096 + 4096, i.e. two of the blocks here) and adds the chunk sums to the running total one
// after the other: total = total + (S[2c] + S[2c+1]).  Inside a block the sweep adds in its own order; that is the same
// number whenever the partial sums are exact, which 4096 float32 values within 2^14 of each other always are in float64
// (24 + 14 + 12 bits) -- so for a matrix that starts at site 0 the n x K sums are NumPy's own, not just close to them.
// With keep_prefix S[block][cell] is replaced by the sum of the blocks before it (what the chain prediction needs);
// chunks (may be NULL) receives the chunk sums C[c][cell] = S[2c] + S[2c+1], from which chunk_total_kernel continues a
// running total handed over by the SNP shard before this one.
__global__ __launch_bounds__(256) void block_prefix_kernel(double *__restrict__ S, int nblocks, int64_t cells, double *__restrict__ out,
                                                           int keep_prefix, double *__restrict__ chunks)
{
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cell >= cells) return;
    double run = 0.0;
    int b = 0;
    for (; b + 8 <= nblocks; b += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = S[(int64_t)(b + u) * cells + cell];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            if (keep_prefix) {
                S[(int64_t)(b + u) * cells + cell] = run;
                S[(int64_t)(b + u + 1) * cells + cell] = run + v[u];
            }
            if (chunks) chunks[(int64_t)((b + u) >> 1) * cells + cell] = v[u] + v[u + 1];
            run = run + (v[u] + v[u + 1]);
        }
    }
    for (; b < nblocks; b += 2) {
        const double v0 = S[(int64_t)b * cells + cell];
        const bool pair = b + 1 < nblocks;
        const double v1 = pair ? S[(int64_t)(b + 1) * cells + cell] : 0.0;
        if (keep_prefix) {
            S[(int64_t)b * cells + cell] = run;
            if (pair) S[(int64_t)(b + 1) * cells + cell] = run + v0;
        }
        if (chunks) chunks[(int64_t)(b >> 1) * cells + cell] = pair ? v0 + v1 : v0;
        run = pair ? run + (v0 + v1) : run + v0;
    }
    out[cell] = run;
}

// out[cell] = (((carry[cell] + C[0]) + C[1]) + ...): the running float64 total of np.sum continued over this shard's chunks.
__global__ __launch_bounds__(256) void chunk_total_kernel(const double *__restrict__ chunks, int nchunks, int64_t cells,
                                                          const double *__restrict__ carry, double *__restrict__ out)
{
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cell >= cells) return;
    double run = carry ? carry[cell] : 0.0;
    for (int c = 0; c < nchunks; ++c) run = run + chunks[(int64_t)c * cells + cell];
    out[cell] = run;
}

// ---- exact partition sums, block-parallel ----------------------------------------------------------
// utils.py:147-149: labels = arange(m) % P; np.add.at(zeros(P, float32), labels, per_site_ll) is a
// SERIAL float32 accumulation per partition in site order.  The per-site values v are <= 0 (logs of
// likelihoods <= 1), so the running sum only grows in magnitude and round-to-nearest-even is symmetric
// in sign: on magnitudes this is the chain of the convergence metric (em_kernels.hip, rmse_*), and the
// same decomposition applies.  While |R| stays in one binade it is M * u (u its ulp, M a 24-bit
// integer) and adding |v| adds the INTEGER RN(|v| / u), which depends on M only at exact ties.  A block
// of sites without a tie therefore adds a fixed integer D -- a plain, order-free sum -- to M.
//
//   chain_cand_kernel  (same sweep as above, one block per wave) forms D for every (individual,
//                      population, partition, block) on the ulp grid PREDICTED for that block from the
//                      float64 prefix sums of the sweep: RN(|v| / u) * u = fl(X0 + |v|) - X0 with
//                      X0 = 2^e, two float32 operations; the rounding error |v| - that is exact, and a
//                      tie is an error of exactly u / 2.  A block with a tie, a positive or non-finite
//                      value, or a sum that leaves the binade is marked.
//   chain_walk_kernel  one wavefront per chain walks the blocks: M += D when the running value really
//                      is in the predicted binade and stays in it, else that ONE block is redone with
//                      the literal serial loop (binade crossings, ties, the first blocks).
// Bit-identical to the serial loop by construction; a wrong prediction costs speed only.
constexpr unsigned CAND_BAD = 0x80000000u;      // packed word: bad << 31 | biased exponent << 23 | D (< 2^23); 0 = no sites

template <int KB, int NP, bool PER_IND>
__global__ __launch_bounds__(256, 2) void chain_cand_kernel(ScoreArgs A)
{
    __shared__ double2 tab_lds[WGS_LOG_N * WGS_LOG_REP];
    extern __shared__ unsigned cand_lds[];                   // [4 waves][2][CW * P]: D sums | exponent + flags
    const double2 *tab = load_log_table(tab_lds);
    const int lane = threadIdx.x & 63;
    WaveWork<NP> w;
    if (!decode_wave<NP>(A, w)) return;
    const float4 *slab = w.sl.slab;
    const int npairs = w.sl.npairs;
    const int64_t t0 = w.t0, t1 = w.t1;
    const int P = A.P, period = A.period;
    const double invP = 1.0 / (double)P;
    const double c8 = log_c8();
    constexpr int CW = NP * 2 * KB;
    unsigned *wD = cand_lds + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 2 * CW * P;
    unsigned *wF = wD + CW * P;

    for (int kb = 0; kb < A.K; kb += KB) {
        constexpr int NA = PER_IND ? NP * 2 * KB : KB;
        gf32_ptr ptr[NA];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int k = kb + j < A.K ? kb + j : A.K - 1;
            if (PER_IND) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) ptr[(q * 2 + h) * KB + j] = (gf32_ptr)A.colptr[(int64_t)w.ind[q][h] * A.K + k];
            } else {
                ptr[j] = (gf32_ptr)A.acol[k];
            }
        }
        for (int e = lane; e < 2 * CW * P; e += 64) wD[e] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // tiles t0 + c, t0 + c + period, ...: in each such class a lane's partition label is constant
        for (int c = 0; c < period && t0 + c < t1; ++c) {
            const int label = (int)((A.site0 + ((t0 + c) << 6) + lane) % P);
            float X0[NP][2][KB], acc[NP][2][KB], emax[NP][2][KB];
            float vmax = 0.0f;
            bool plain = true;          // every likelihood sum of this lane was a positive finite number
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        const int k = kb + j < A.K ? kb + j : A.K - 1;
                        const int64_t cell = (int64_t)w.ind[q][h] * A.K + k;
                        // running value predicted at the start of this block: an equal share of the
                        // float64 sums of everything before it (preceding shards + preceding blocks)
                        const double before = (A.start ? A.start[cell] : 0.0) + A.S[w.blk * A.cells + cell];
                        const float est = (float)(before * invP);
                        const unsigned eb = (__float_as_uint(est) >> 23) & 0xFF;
                        const bool known = est < 0.0f && eb >= 30 && eb <= 253;        // u/2 = 2^(eb-151) stays a normal float32
                        X0[q][h][j] = known ? __uint_as_float(eb << 23) : 0.0f;   // 2^e; 0 marks "no prediction"
                        acc[q][h][j] = 0.0f;
                        emax[q][h][j] = 0.0f;
                    }
            float4 g_cur[NP], g_nxt[NP];
            float a_cur[NA], a_nxt[NA];
            auto fetch = [&](int64_t t, float4 *gq, float *aq) {
                const int64_t s = (t << 6) + lane;
                const int64_t sc = s < A.m ? s : A.m - 1;
#pragma unroll
                for (int q = 0; q < NP; ++q) gq[q] = slab[((t * npairs + w.pairc[q]) << 6) + lane];
#pragma unroll
                for (int x = 0; x < NA; ++x) aq[x] = ptr[x][sc];
            };
            fetch(t0 + c, g_cur, a_cur);
            for (int64_t t = t0 + c; t < t1; t += period) {
                if (t + period < t1) fetch(t + period, g_nxt, a_nxt);
                const bool live = ((t << 6) + lane) < A.m;
                double ad[KB], oma[KB];
                if (!PER_IND) {
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        ad[j] = (double)(live ? a_cur[j] : 0.0f);
                        oma[j] = 1.0 - ad[j];
                    }
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const float4 g = g_cur[q];
                    const float gl[2][2] = {{live ? g.x : 1.0f, live ? g.y : 0.0f}, {live ? g.z : 1.0f, live ? g.w : 0.0f}};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const double g0d = (double)gl[h][0], g1d = (double)gl[h][1];
                        const double g1d2 = g1d * 2.0, g2d = (1.0 - g0d) - g1d;
#pragma unroll
                        for (int j = 0; j < KB; ++j) {
                            const double a_d = PER_IND ? (double)(live ? a_cur[(q * 2 + h) * KB + j] : 0.0f) : ad[j];
                            const float s = like_sum_exact(g0d, g1d2, g2d, a_d, PER_IND ? 1.0 - a_d : oma[j]);
                            // v = the float32 the reference stores for this site -- unless s is 0, negative or
                            // NaN (log_f32arg is not defined there): such a lane marks its blocks instead
                            plain = plain && __builtin_isfpclass(s, FP_POS_FINITE);
                            const float v = (float)log_f32arg((double)s, tab, c8);
                            const float mag = -v;                               // >= 0 for every likelihood <= 1
                            const float r = (X0[q][h][j] + mag) - X0[q][h][j];  // RN(mag / u) * u on the predicted grid
                            acc[q][h][j] += r;                                  // exact while the block stays in the binade
                            emax[q][h][j] = __builtin_fmaxf(emax[q][h][j], __builtin_fabsf(mag - r));
                            vmax = __builtin_fmaxf(vmax, v);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) g_cur[q] = g_nxt[q];
#pragma unroll
                for (int x = 0; x < NA; ++x) a_cur[x] = a_nxt[x];
            }
            // this lane's share of its (cell, label) block functions -> LDS (integer adds: order-free)
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        const float x0 = X0[q][h][j], a = acc[q][h][j];
                        const unsigned eb = __float_as_uint(x0) >> 23;
                        // bad: no prediction; left the binade / non-finite (a < x0 fails for NaN and inf too);
                        // a tie (rounding error exactly u/2 = x0 * 2^-24); a positive value
                        const bool bad = !(x0 > 0.0f) || !(a < x0) || emax[q][h][j] == x0 * 0x1p-24f || vmax > 0.0f || !plain;
                        const unsigned D = bad ? 0u : (unsigned)__builtin_ldexpf(a, 150 - (int)eb);
                        const int slot = ((q * 2 + h) * KB + j) * P + label;
                        atomicAdd(&wD[slot], D);
                        atomicOr(&wF[slot], (bad ? CAND_BAD : 0u) | (eb << 23) | 1u);     // bit 0: the label has sites here
                    }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < CW * P; e += 64) {
            const int cellslot = e / P, p = e - cellslot * P;
            const int j = cellslot % KB, qh = cellslot / KB;
            const int col = 2 * (w.sl.pair0 + w.pg * NP + qh / 2) + (qh & 1);
            if (kb + j >= A.K || col < w.sl.col_lo || col >= w.sl.col_hi) continue;
            const int ind = w.sl.members[col];
            const unsigned F = wF[e], D = wD[e];
            unsigned word = 0;
            if (F & 1u) word = ((F & CAND_BAD) || D >= (1u << 23)) ? CAND_BAD : ((F & 0x7F800000u) | D);
            A.cand[(((int64_t)ind * P + p) * A.K + kb + j) * A.nblocks + w.blk] = word;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// One wavefront per chain (individual, partition, population).  All 64 lanes hold the same running
// value; the serial fallback of a block computes its per-site values 64 at a time into LDS.
constexpr int WALK_CHUNK = 1024;
__global__ __launch_bounds__(256) void chain_walk_kernel(WalkArgs W)
{
    __shared__ double2 tab_lds[WGS_LOG_N * WGS_LOG_REP];
    __shared__ __attribute__((aligned(16))) float sq_all[4][WALK_CHUNK];
    const double2 *tab = load_log_table(tab_lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nchains = (int64_t)W.n * W.P * W.K;
    const int64_t chain = (int64_t)blockIdx.x * 4 + wave;
    if (chain >= nchains) return;
    const int k = (int)(chain % W.K);
    const int64_t ip = chain / W.K;
    const int p = (int)(ip % W.P), ind = (int)(ip / W.P);
    if (ind < W.row_lo || ind >= W.row_hi) return;
    float *sq = sq_all[wave];
    const int g = W.group_of[ind], col = W.col_of[ind], npairs = W.npairs[g];
    const float2 *slab2 = reinterpret_cast<const float2 *>(W.base[g]);
    gf32_ptr ptr = (gf32_ptr)(W.colptr ? W.colptr[(int64_t)ind * W.K + k] : W.acol[k]);
    const int64_t pair_off = (int64_t)(col >> 1) * 64, half = col & 1;
    float res = W.carry ? W.carry[chain] : 0.0f;
    const unsigned *cw = W.cand + chain * W.nblocks;
    int serial = 0;
    for (int b0 = 0; b0 < W.nblocks && res == res; b0 += 64) {      // NaN + anything = NaN: nothing left to do
        const unsigned mine = b0 + lane < W.nblocks ? cw[b0 + lane] : 0u;
        const int nb = W.nblocks - b0 < 64 ? W.nblocks - b0 : 64;
        for (int jb = 0; jb < nb && res == res; ++jb) {
            const unsigned word = __builtin_amdgcn_readlane(mine, jb);
            if (word == 0u) continue;                               // no site of this partition in the block
            const unsigned bits = __float_as_uint(res);
            if (!(word & CAND_BAD) && (bits >> 31) && ((bits >> 23) & 0xFF) == ((word >> 23) & 0xFF)) {
                const unsigned M = (bits & 0x7FFFFFu) | 0x800000u, D = word & 0x7FFFFFu;
                if (M + D < (1u << 24)) {
                    res = __uint_as_float((bits & 0xFF800000u) | ((M + D) & 0x7FFFFFu));
                    continue;
                }
            }
            // the literal serial loop for this block: sites s of [64 * 64 * b, ...) with (site0 + s) % P == p
            ++serial;
            const int64_t lo = (int64_t)(b0 + jb) * WGS_BLOCK_TILES * 64;
            int64_t hi = lo + WGS_BLOCK_TILES * 64;
            if (hi > W.m) hi = W.m;
            const int64_t first = lo + (((p - (W.site0 + lo)) % W.P) + W.P) % W.P;
            for (int64_t c0 = first; c0 < hi; c0 += (int64_t)WALK_CHUNK * W.P) {
                int cnt = 0;
#pragma unroll 8
                for (int r = 0; r < WALK_CHUNK / 64; ++r) {
                    const int64_t s = c0 + (int64_t)(r * 64 + lane) * W.P;
                    float v = 0.0f;
                    if (s < hi) {
                        const float2 gg = slab2[((((s >> 6) * npairs) << 6) + pair_off + (s & 63)) * 2 + half];
                        const double g0d = (double)gg.x, g1d = (double)gg.y;
                        v = site_ll_exact(g0d, g1d, (1.0 - g0d) - g1d, ptr[s], tab);
                    }
                    sq[r * 64 + lane] = v;
                }
                {
                    const int64_t left = (hi - c0 + W.P - 1) / W.P;
                    cnt = left < WALK_CHUNK ? (int)left : WALK_CHUNK;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // float32 += float32 in site order.  The adds are one dependent chain; the LDS reads are not:
                // 16 values per batch of four ds_read_b128, so their latency is paid once per batch, not per add
                int t = 0;
                for (; t + 16 <= cnt; t += 16) {
                    const float4 a = *reinterpret_cast<const float4 *>(sq + t), b = *reinterpret_cast<const float4 *>(sq + t + 4);
                    const float4 c = *reinterpret_cast<const float4 *>(sq + t + 8), d = *reinterpret_cast<const float4 *>(sq + t + 12);
                    res = res + a.x; res = res + a.y; res = res + a.z; res = res + a.w;
                    res = res + b.x; res = res + b.y; res = res + b.z; res = res + b.w;
                    res = res + c.x; res = res + c.y; res = res + c.z; res = res + c.w;
                    res = res + d.x; res = res + d.y; res = res + d.z; res = res + d.w;
                }
                for (; t < cnt; ++t) res = res + sq[t];
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (lane == 0) {
        W.parts[chain] = res;
        if (W.n_serial && serial) atomicAdd(W.n_serial, serial);
    }
}

// ---- exact partition sums, literal version -----------------------------------------------------------
// One lane owns one (individual, population, partition) chain and walks the partition's sites of this
// shard in order, starting from the float32 carry of the previous shard.  Parallelism is across the
// n x K x P chains only, so the time is ~ (m / P) x one site's latency.  Kept as the in-device
// cross-check of the block-parallel chains above and for P > 64 (many short chains).
__global__ __launch_bounds__(64) void parts_exact_kernel(AssignArgs A, const PartsSlab *__restrict__ slabs, int n_slabs,
                                                         const float *__restrict__ carry, float *__restrict__ parts)
{
    __shared__ double2 tab_lds[WGS_LOG_N * WGS_LOG_REP];
    const double2 *tab = load_log_table(tab_lds);
    const int lane = threadIdx.x;
    // all population slabs share ONE launch (their chains are independent and each slab alone would
    // put only a handful of waves on the chip): find this workgroup's slab
    int g = 0;
    while (g + 1 < n_slabs && (int)blockIdx.x >= slabs[g + 1].block0) ++g;
    A.slab = slabs[g].slab;
    A.members = slabs[g].members;
    A.npairs = slabs[g].npairs;
    A.ncols = slabs[g].ncols;
    const int c = ((int)blockIdx.x - slabs[g].block0) * 64 + lane;
    const bool valid = c < A.ncols;
    const int cc = valid ? c : A.ncols - 1;
    const int j = blockIdx.y, p = blockIdx.z;
    const int ind = A.members[cc];
    gf32_ptr ptr = (gf32_ptr)(A.colptr ? A.colptr[(int64_t)ind * A.K + j] : A.acol[j]);
    const int64_t cell = ((int64_t)ind * A.P + p) * A.K + j;
    float res = carry ? carry[cell] : 0.0f;
    const float2 *slab2 = reinterpret_cast<const float2 *>(A.slab);
    const int64_t first = ((p - A.site0 % A.P) % A.P + A.P) % A.P;
    const int64_t pair_off = (int64_t)(cc >> 1) * 64, half = cc & 1;
    // Only ~n*K*P/64 waves exist, so nothing hides memory latency but this wave's own loads: the GLs
    // and frequencies of the next PU sites are requested before the current PU are consumed.
    constexpr int PU = 16;
    for (int64_t s = first; s < A.m; s += (int64_t)A.P * PU) {
        float2 g[PU];
        float a[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            int64_t ss = s + (int64_t)u * A.P;
            if (ss >= A.m) ss = A.m - 1;                          // clamped address, value unused
            g[u] = slab2[((((ss >> 6) * A.npairs) << 6) + pair_off + (ss & 63)) * 2 + half];
            a[u] = ptr[ss];
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            if (s + (int64_t)u * A.P < A.m) {                     // wave-uniform
                const double g0d = (double)g[u].x, g1d = (double)g[u].y;
                const float v = site_ll_exact(g0d, g1d, (1.0 - g0d) - g1d, a[u], tab);
                res = res + v;                                    // float32 += float32, site order
            }
        }
    }
    if (valid) parts[cell] = res;
}

// The thin mirror of glassy_cy.loglike: vec[s] = (float)((double)vec[s] + log(...)), one
// individual (its (g0,g1) column compacted to g[m]) and one population (a[m]).
template <int MODE>
__global__ void loglike_site_kernel(const float2 *__restrict__ g, const float *__restrict__ a, float *vec, int64_t m)
{
    __shared__ double2 tab_lds[WGS_LOG_N * WGS_LOG_REP];
    const double2 *tab = load_log_table(tab_lds);
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; s < m; s += stride) {
        const float2 gg = g[s];
        const float av = a[s];
        const double g0d = (double)gg.x, g1d = (double)gg.y;
        if (MODE == WGS_MODE_EXACT) {
            const double ad = (double)av, oma = 1.0 - ad;
            const float like0 = (float)((g0d * oma) * oma);
            const float like1 = (float)(((g1d * 2.0) * oma) * ad);
            const float like2 = (float)((((1.0 - g0d) - g1d) * ad) * ad);
            // vec + log(.) in double, then float32 (glassy_cy.pyx:21); the log itself is needed to
            // double precision here because vec[s] may be non-zero (accumulate-into semantics)
            const float sum = (like0 + like1) + like2;
            double l = log_f32arg((double)sum, tab);
            l = sum == 0.0f ? -(double)__builtin_inff() : l;
            l = sum == __builtin_inff() ? (double)sum : l;
            l = !(sum >= 0.0f) ? (double)__builtin_nanf("") : l;
            vec[s] = (float)((double)vec[s] + l);
        } else {
            vec[s] = vec[s] + site_ll_fast(gg.x, gg.y, (1.0f - gg.x) - gg.y, av);
        }
    }
}

// Test hooks: the float32-rounded log of every float32 in a bit-pattern range, custom vs ocml.
__global__ void log_mismatch_kernel(unsigned int b0, unsigned int b1, unsigned long long *count, unsigned int *first)
{
    __shared__ double2 tab_lds[WGS_LOG_N * WGS_LOG_REP];
    const double2 *tab = load_log_table(tab_lds);
    unsigned long long local = 0;
    for (unsigned long long b = (unsigned long long)b0 + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; b < b1;
         b += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned int)b);
        const float mine = logf_of_f32(x, tab);
        const float ref = (float)log((double)x);
        if (__float_as_uint(mine) != __float_as_uint(ref) && !(mine != mine && ref != ref)) {
            ++local;
            atomicMin(first, (unsigned int)b);
        }
    }
    if (local) atomicAdd(count, local);
}

__global__ void log_values_kernel(const float *x, float *out, int64_t n, int use_libm)
{
    __shared__ double2 tab_lds[WGS_LOG_N * WGS_LOG_REP];
    const double2 *tab = load_log_table(tab_lds);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = use_libm ? (float)log((double)x[i]) : logf_of_f32(x[i], tab);
}

template <int KB>
int launch_assign_kb(wgs_ctx *ctx, const AssignArgs &a, int mode, dim3 grid)
{
    if (mode == WGS_MODE_EXACT)
        hipLaunchKernelGGL((assign_kernel<KB, WGS_MODE_EXACT>), grid, dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((assign_kernel<KB, WGS_MODE_FAST>), grid, dim3(256), 0, ctx->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

static int ensure_log_table(wgs_ctx *ctx)
{
    // one upload per context (the table lives in the code object's __device__ memory of the context's device)
    if (ctx->log_table_ready) return 0;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(wgs_log_table_dev), wgs_log_table_host, sizeof(double) * 2 * WGS_LOG_N));
    ctx->log_table_ready = true;
    return 0;
}

int launch_log_mismatch(wgs_ctx *ctx, unsigned int b0, unsigned int b1, unsigned long long *d_count, unsigned int *d_first)
{
    if (ensure_log_table(ctx)) return 1;
    hipLaunchKernelGGL(log_mismatch_kernel, dim3(4096), dim3(256), 0, ctx->stream, b0, b1, d_count, d_first);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_log_values(wgs_ctx *ctx, const float *d_x, float *d_out, int64_t n, int use_libm)
{
    if (ensure_log_table(ctx)) return 1;
    hipLaunchKernelGGL(log_values_kernel, dim3(1024), dim3(256), 0, ctx->stream, d_x, d_out, n, use_libm);
    HIP_TRY(hipGetLastError());
    return 0;
}

// KB = populations per register batch: the batch size with the fewest passes over K, then the least padding.
// Every pass re-reads the block's GLs, so K <= 10 is ONE pass (HBM traffic = algorithmic bytes) and K = 20 two;
// KB = 9, 10 run one pair per wave at 2 waves/SIMD (measured equal to two passes of 5 in exact mode -- the kernel is
// bound by FP64 issue either way -- and 11 % faster in float32 mode).
static int pick_kb(int K, int kb_max = 10)
{
    int best = 4, best_cost = 1 << 30;
    for (int kb = 4; kb <= kb_max; ++kb) {
        const int passes = (K + kb - 1) / kb;
        const int cost = passes * 1000 + passes * kb - K;
        if (cost < best_cost) best_cost = cost, best = kb;
    }
    return best;
}

int score_pairs_per_wave(int K, bool per_ind) { return sweep_pairs(pick_kb(K), per_ind); }
int score_kb(int K) { return pick_kb(K); }
// The chain kernel keeps three float32 per (cell, lane) instead of one float64; with per-individual columns
// its pointer and frequency tables leave room for one pair only.
// It stays with batches of at most 8 populations (9 and 10 would spill).
static int pick_kb_chain(int K) { return pick_kb(K, 8); }
int chain_pairs_per_wave(int K, bool per_ind) { return chain_pairs(pick_kb_chain(K), per_ind); }

// The float64 partition sums of WGSASSIGN_PARTS=fast (P > 1): lane <-> individual pair, one slab per launch.
int launch_assign(wgs_ctx *ctx, const AssignArgs &a_in, int mode)
{
    if (ensure_log_table(ctx)) return 1;
    AssignArgs a = a_in;
    if (a.m <= 0 || a.ncols <= 0 || a.K <= 0) return 0;
    // Enough waves to fill 256 CUs several times over, but tile ranges long enough to amortise the
    // per-wave prologue (pointer table) and the atomics.
    const int pairblocks = (a.npairs + 63) / 64;
    const int64_t ntiles = wgs_ntiles(a.m);
    int64_t want_waves = (int64_t)ctx->cus * 32 / pairblocks;
    if (want_waves < 4) want_waves = 4;
    int64_t tpw = (ntiles + want_waves - 1) / want_waves;
    if (tpw < 1) tpw = 1;
    a.tiles_per_wave = (int32_t)(tpw > 0x7fffffff ? 0x7fffffff : tpw);
    const int64_t waves = (ntiles + a.tiles_per_wave - 1) / a.tiles_per_wave;
    dim3 grid((unsigned)((waves + 3) / 4), (unsigned)pairblocks);
    switch (pick_kb(a.K, 8)) {
        case 4: return launch_assign_kb<4>(ctx, a, mode, grid);
        case 5: return launch_assign_kb<5>(ctx, a, mode, grid);
        case 6: return launch_assign_kb<6>(ctx, a, mode, grid);
        case 7: return launch_assign_kb<7>(ctx, a, mode, grid);
        default: return launch_assign_kb<8>(ctx, a, mode, grid);
    }
}

#define WGS_FOR_KB(X, K)                \
    switch (pick_kb(K)) {               \
        case 4: X(4); break;            \
        case 5: X(5); break;            \
        case 6: X(6); break;            \
        case 7: X(7); break;            \
        case 8: X(8); break;            \
        case 9: X(9); break;            \
        default: X(10); break;          \
    }

int launch_score_sweep(wgs_ctx *ctx, const ScoreArgs &a, int mode)
{
    if (a.m <= 0 || a.total_pg <= 0 || a.K <= 0 || a.nblocks <= 0) return 0;
    if (ensure_log_table(ctx)) return 1;
    const int64_t waves = (int64_t)a.total_pg * a.nblocks;
    WGS_REQUIRE(waves < (1ll << 32), "scoring sweep: too many work units for one launch");
    dim3 grid((unsigned)((waves + 3) / 4));
    const bool per_ind = a.colptr != nullptr;
#define WGS_SWEEP(KB)                                                                                                                          \
    do {                                                                                                                                       \
        if (mode == WGS_MODE_EXACT) {                                                                                                          \
            if (per_ind) hipLaunchKernelGGL((score_sweep_kernel<KB, sweep_pairs(KB, true), WGS_MODE_EXACT, true>), grid, dim3(256), 0, ctx->stream, a);   \
            else hipLaunchKernelGGL((score_sweep_kernel<KB, sweep_pairs(KB, false), WGS_MODE_EXACT, false>), grid, dim3(256), 0, ctx->stream, a);         \
        } else {                                                                                                                               \
            if (per_ind) hipLaunchKernelGGL((score_sweep_kernel<KB, sweep_pairs(KB, true), WGS_MODE_FAST, true>), grid, dim3(256), 0, ctx->stream, a);    \
            else hipLaunchKernelGGL((score_sweep_kernel<KB, sweep_pairs(KB, false), WGS_MODE_FAST, false>), grid, dim3(256), 0, ctx->stream, a);          \
        }                                                                                                                                      \
    } while (0)
    WGS_FOR_KB(WGS_SWEEP, a.K)
#undef WGS_SWEEP
    HIP_TRY(hipGetLastError());
    return 0;
}

// S[block][cell] = ((Sp[0] + Sp[1]) + Sp[2]) + ... over the parts of a block: a fixed order (and exact anyway while the
// block's partial sums are).
__global__ __launch_bounds__(256) void combine_parts_kernel(const double *__restrict__ Sp, double *__restrict__ S, int64_t total, int parts)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    double acc = Sp[e];
    for (int p = 1; p < parts; ++p) acc += Sp[(int64_t)p * total + e];
    S[e] = acc;
}

// float table rows where they need (almost) no padding to 16 bytes, float64 rows (no conversion in phase 2) elsewhere
static bool score_coded_wide(int kb)
{
    const char *e = getenv("WGS_SCORE_CODED_TABLE");
    if (e && e[0] == 'f') return false;
    if (e && e[0] == 'd') return true;
    return ((kb + 3) & ~3) - kb > 1;
}
size_t score_coded_lds_bytes(int rows, int kb, int batch)
{
    const size_t row = (batch < 16 || score_coded_wide(kb)) ? sizeof(double) * (kb | 1) : sizeof(float) * ((kb + 3) & ~3);
    return sizeof(double2) * WGS_LOG_N * CODED_LOG_REP + ((2 * sizeof(CodedPrep) + 15) & ~(size_t)15) + sizeof(float2) * (size_t)((rows + 1) & ~1) +
           row * (size_t)(rows + 1);                      // (+ the row of zeros that absent and uncoded SNPs read)
}

// The scoring sweep through the class codes (shared columns only).  d_slabs: n_slabs CodedSlab records in device memory.
int launch_score_coded(wgs_ctx *ctx, const wgs_codes *c, const void *d_slabs, int n_slabs, int total_quads, const float *const *d_acol,
                       int64_t m, int64_t cells, int K, int nblocks, double *S, int mode)
{
    if (m <= 0 || total_quads <= 0 || K <= 0 || nblocks <= 0) return 0;
    if (ensure_log_table(ctx)) return 1;
    CodedScoreArgs A;
    A.dict = c->dict;
    A.ncls = c->ncls;
    A.slabs = reinterpret_cast<const CodedSlab *>(d_slabs);
    A.n_slabs = n_slabs;
    A.drows = c->drows;
    A.total_quads = total_quads;
    A.acol = d_acol;
    A.m = m;
    A.cells = cells;
    A.K = K;
    A.nblocks = nblocks;
    A.S = S;
    A.table_rows = c->rows_batch;
    void *sym = nullptr;
    HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(wgs_log_table_dev)));
    A.logtab = reinterpret_cast<const double2 *>(sym);
    const int kb = pick_kb(K);
    const int batch = c->score_batch;
    const bool wide = batch < 16 || score_coded_wide(kb);          // (the 8- and 4-SNP tables exist with float64 rows only)
    const size_t lds = score_coded_lds_bytes(c->rows_batch, kb, batch);
    WGS_REQUIRE(lds <= 64 * 1024, "class table too large for LDS");
    const unsigned ygroups = (unsigned)((total_quads + 255) / 256);
    // A block's 64 tiles go to `parts` workgroups: enough of them to fill the chip (short matrices), and -- the workgroups all take
    // the same time -- a count that does not leave the last round of workgroups mostly empty: 2442 blocks on 768 places (3 per CU by
    // registers, fewer when the table is large) are 3.18 rounds, i.e. a fifth of the chip-time idle; in halves 6.36 of 7, in quarters
    // 12.7 of 13.
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / std::max<size_t>(lds, 1)));
    const double places = (double)std::max(1, ctx->cus) * per_cu;
    // Any count from 1 to 16 whose runs of ceil(64 / parts) tiles are all non-empty (round 5, late: the powers of two alone left a
    // shard of 1.25M SNPs -- 306 blocks -- at 6.4 rounds of 7 with 16 parts; 5 parts are 1.99 rounds of 2).  The workgroups all take
    // about the same time, a run's tiles plus what a workgroup costs before its first one (the log table, the first batch's trips to
    // memory: about a tile's worth), so the split with the fewest rounds x (tiles per run + 1) wins -- tools/probe_score_parts.py
    // times every split: 5 at 1.25M SNPs (1.44 ms; 16: 1.59), 10 at 300 k (0.45 ms; 16: 0.46, 4: 0.73), 3-5 or 12 at 10M (within 2 %).
    auto usable = [](int p) { return (p - 1) * ((WGS_BLOCK_TILES + p - 1) / p) < WGS_BLOCK_TILES; };
    int parts = 1;
    {
        double best_cost = 0.0;
        for (int p = 1; p <= 16; ++p) {
            if (!usable(p)) continue;
            const double rounds = ceil((double)nblocks * ygroups * p / places);
            const double cost = rounds * (double)((WGS_BLOCK_TILES + p - 1) / p + 1);
            if (p == 1 || cost < best_cost - 1e-9) parts = p, best_cost = cost;
        }
    }
    if (const char *pe = getenv("WGS_SCORE_CODED_PARTS")) {        // experiments: 1 .. 16
        const int v = atoi(pe);
        if (v >= 1 && v <= 16 && usable(v)) parts = v;
    }
    A.parts = parts;
    const int64_t total = (int64_t)nblocks * cells;
    if (parts > 1) {
        void *ws = nullptr;
        if (wgs_ctx_workspace(ctx, sizeof(double) * (size_t)total * parts, &ws)) return 1;
        HIP_TRY(hipMemsetAsync(ws, 0, sizeof(double) * (size_t)total * parts, ctx->stream));   // rows outside the scored range stay 0
        A.S = reinterpret_cast<double *>(ws);
    }
    dim3 grid((unsigned)nblocks, ygroups, (unsigned)parts);
#define WGS_CODED(KB)                                                                                                     \
    do {                                                                                                                  \
        if (batch == 8) {                                                                                                 \
            if (mode == WGS_MODE_EXACT) hipLaunchKernelGGL((score_coded_kernel<KB, WGS_MODE_EXACT, double, 8>), grid, dim3(256), lds, ctx->stream, A); \
            else hipLaunchKernelGGL((score_coded_kernel<KB, WGS_MODE_FAST, double, 8>), grid, dim3(256), lds, ctx->stream, A); \
        } else if (batch == 4) {                                                                                          \
            if (mode == WGS_MODE_EXACT) hipLaunchKernelGGL((score_coded_kernel<KB, WGS_MODE_EXACT, double, 4>), grid, dim3(256), lds, ctx->stream, A); \
            else hipLaunchKernelGGL((score_coded_kernel<KB, WGS_MODE_FAST, double, 4>), grid, dim3(256), lds, ctx->stream, A); \
        } else if (wide) {                                                                                                \
            if (mode == WGS_MODE_EXACT) hipLaunchKernelGGL((score_coded_kernel<KB, WGS_MODE_EXACT, double, 16>), grid, dim3(256), lds, ctx->stream, A); \
            else hipLaunchKernelGGL((score_coded_kernel<KB, WGS_MODE_FAST, double, 16>), grid, dim3(256), lds, ctx->stream, A); \
        } else {                                                                                                          \
            if (mode == WGS_MODE_EXACT) hipLaunchKernelGGL((score_coded_kernel<KB, WGS_MODE_EXACT, float, 16>), grid, dim3(256), lds, ctx->stream, A); \
            else hipLaunchKernelGGL((score_coded_kernel<KB, WGS_MODE_FAST, float, 16>), grid, dim3(256), lds, ctx->stream, A); \
        }                                                                                                                 \
    } while (0)
    WGS_FOR_KB(WGS_CODED, K)
#undef WGS_CODED
    HIP_TRY(hipGetLastError());
#ifdef WGS_SCORE_STATS
    {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        unsigned long long st[8] = {0}, zero[8] = {0};
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_score_stats), sizeof st);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_score_stats), zero, sizeof zero);
        const double w = st[6] ? (double)st[6] : 1.0, nb = st[5] ? (double)st[5] : 1.0;
        fprintf(stderr, "[score stats] %llu wavefronts, %.0f batches each; cycles per wavefront and batch: before phase 1 %.0f, phase 1 %.0f, barrier %.0f, phase 2 %.0f, "
                "barrier %.0f\n", st[6], nb / w, st[0] / nb, st[1] / nb, st[2] / nb, st[3] / nb, st[4] / nb);
    }
#endif
    if (parts > 1) {
        hipLaunchKernelGGL(combine_parts_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, A.S, S, total, parts);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int launch_block_prefix(wgs_ctx *ctx, double *S, int nblocks, int64_t cells, double *out, int keep_prefix, double *chunks)
{
    if (cells <= 0) return 0;
    hipLaunchKernelGGL(block_prefix_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream, S, nblocks, cells, out,
                       keep_prefix, chunks);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_chunk_total(wgs_ctx *ctx, const double *chunks, int nchunks, int64_t cells, const double *carry, double *out)
{
    if (cells <= 0) return 0;
    hipLaunchKernelGGL(chunk_total_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream, chunks, nchunks, cells, carry, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t chain_cand_lds_bytes(int K, int P, bool per_ind)
{
    const int kb = pick_kb_chain(K), np = chain_pairs_per_wave(K, per_ind);
    return (size_t)4 * 2 * (np * 2 * kb) * P * sizeof(unsigned);
}

int launch_chain_cand(wgs_ctx *ctx, const ScoreArgs &a)
{
    if (a.m <= 0 || a.total_pg <= 0 || a.K <= 0 || a.nblocks <= 0) return 0;
    if (ensure_log_table(ctx)) return 1;
    const int64_t waves = (int64_t)a.total_pg * a.nblocks;
    WGS_REQUIRE(waves < (1ll << 32), "partition chains: too many work units for one launch");
    dim3 grid((unsigned)((waves + 3) / 4));
    const bool per_ind = a.colptr != nullptr;
    const size_t lds = chain_cand_lds_bytes(a.K, a.P, per_ind);
    WGS_REQUIRE(lds <= 96 * 1024, "partition chains: too many partitions for the block-parallel kernel");
#define WGS_CAND(KB)                                                                                                                  \
    do {                                                                                                                              \
        if (per_ind) hipLaunchKernelGGL((chain_cand_kernel<KB, chain_pairs(KB, true), true>), grid, dim3(256), lds, ctx->stream, a);  \
        else hipLaunchKernelGGL((chain_cand_kernel<KB, chain_pairs(KB, false), false>), grid, dim3(256), lds, ctx->stream, a);        \
    } while (0)
    switch (pick_kb_chain(a.K)) {
        case 4: WGS_CAND(4); break;
        case 5: WGS_CAND(5); break;
        case 6: WGS_CAND(6); break;
        case 7: WGS_CAND(7); break;
        default: WGS_CAND(8); break;
    }
#undef WGS_CAND
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_chain_walk(wgs_ctx *ctx, const WalkArgs &w)
{
    const int64_t nchains = (int64_t)w.n * w.P * w.K;
    if (nchains <= 0 || w.m <= 0) return 0;
    if (ensure_log_table(ctx)) return 1;
    WGS_REQUIRE(nchains < (1ll << 32), "partition chains: too many chains for one launch");
    hipLaunchKernelGGL(chain_walk_kernel, dim3((unsigned)((nchains + 3) / 4)), dim3(256), 0, ctx->stream, w);
    HIP_TRY(hipGetLastError());
    return 0;
}

// a: the fields common to all slabs (colptr, acol, m, site0, K, P); d_slabs: n_slabs descriptors with
// block0 = first workgroup of each slab, total_blocks workgroups in all.
int launch_parts_exact(wgs_ctx *ctx, const AssignArgs &a, const PartsSlab *d_slabs, int n_slabs, int total_blocks,
                       const float *d_carry, float *d_parts)
{
    if (a.m <= 0 || n_slabs <= 0 || total_blocks <= 0 || a.K <= 0 || a.P <= 0) return 0;
    if (ensure_log_table(ctx)) return 1;
    WGS_REQUIRE(a.K <= 65535 && a.P <= 65535, "too many populations / partitions for one launch");
    dim3 grid((unsigned)total_blocks, (unsigned)a.K, (unsigned)a.P);
    hipLaunchKernelGGL(parts_exact_kernel, grid, dim3(64), 0, ctx->stream, a, d_slabs, n_slabs, d_carry, d_parts);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_loglike_site(wgs_ctx *ctx, const float2 *g, const float *a, float *vec, int64_t m, int mode)
{
    if (m <= 0) return 0;
    if (ensure_log_table(ctx)) return 1;
    int64_t blocks = (m + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (mode == WGS_MODE_EXACT)
        hipLaunchKernelGGL(loglike_site_kernel<WGS_MODE_EXACT>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g, a, vec, m);
    else
        hipLaunchKernelGGL(loglike_site_kernel<WGS_MODE_FAST>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g, a, vec, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
