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

__device__ __forceinline__ double log_f32arg(double x, const double2 *tab)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    const unsigned int hi = (unsigned int)(bits >> 32), lo = (unsigned int)bits;
    const unsigned int tmp = hi - WGS_LOG_OFF;
    const int k = (int)tmp >> 20;                          // x = z * 2^k, z in [0.6875, 1.375)
    const unsigned int i = (tmp >> 13) & (WGS_LOG_N - 1);
    const double z = __hiloint2double((int)(hi - (tmp & 0xFFF00000u)), (int)lo);
    const double2 t = tab[i * WGS_LOG_REP];                // {invc, logc}; tab already points at this lane's copy
    const double r = __builtin_fma(z, t.x, -1.0);          // exact
    const double kd = (double)k;
    const double w = __builtin_fma(kd, WGS_LN2HI, t.y);    // kd*Ln2hi is exact
    const double hi_ = w + r;
    const double lo_ = __builtin_fma(kd, WGS_LN2LO, (w - hi_) + r);
    double q = __builtin_fma(r, -0.125, 1.0 / 7.0);        // log1p(r) = r + r^2 * q(r)
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

// ---- P == 1 path: lane <-> SNP -----------------------------------------------------------------
// wave <-> (group of NP individual pairs, range of tiles); lane <-> SNP of the tile.  GL loads are
// the slab's native 1 KiB wave loads, the K frequencies of a SNP are loaded once per tile and
// their double forms (a, 1-a) hoisted over the individuals (shared-A mode), and in leave-one-out
// mode the per-individual frequency vectors are read coalesced (lane = SNP).  Each lane keeps
// NP x 2 x KB float64 partial sums over the tiles of its range; one cross-lane reduction per
// (individual, population) ends the range.  Pair-group index varies fastest over workgroups, so
// waves that need the same tile's frequencies run together (L2 hits).
__device__ __forceinline__ float site_ll_exact2(double g0d, double g1d2, double g2d, double ad, double oma, const double2 *tab)
{
    const float like0 = (float)((g0d * oma) * oma);
    const float like1 = (float)((g1d2 * oma) * ad);        // ((g1*2.0)*(1-a))*a, g1*2.0 exact
    const float like2 = (float)((g2d * ad) * ad);
    return logf_of_f32((like0 + like1) + like2, tab);
}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

template <int KB, int NP, int MODE, bool PER_IND>
__global__ __launch_bounds__(256) void assign_snp_kernel(AssignArgs A)
{
    __shared__ double2 tab_lds[WGS_LOG_N * WGS_LOG_REP];
    const double2 *tab = load_log_table(tab_lds);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int npg = (A.npairs + NP - 1) / NP;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int pg = (int)(wid % npg);
    const int64_t tr = wid / npg;
    const int64_t ntiles = (A.m + 63) >> 6;
    const int64_t t0 = tr * A.tiles_per_wave;
    int64_t t1 = t0 + A.tiles_per_wave;
    if (t1 > ntiles) t1 = ntiles;
    if (t0 >= t1) return;

    int ind[NP][2];
    bool ok[NP][2];
    int pairc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int pair = pg * NP + q;
        pairc[q] = pair < A.npairs ? pair : A.npairs - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ok[q][h] = 2 * pair + h < A.ncols;
            ind[q][h] = A.members[ok[q][h] ? 2 * pair + h : 0];
        }
    }

    for (int kb = 0; kb < A.K; kb += KB) {
        gf32_ptr ptr[PER_IND ? NP * 2 * KB : KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int k = kb + j < A.K ? kb + j : A.K - 1;
            if (PER_IND) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int h = 0; h < 2; ++h) ptr[(q * 2 + h) * KB + j] = (gf32_ptr)A.colptr[(int64_t)ind[q][h] * A.K + k];
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

        // One tile ahead: the GLs and frequencies of tile t+1 are requested before tile t is
        // consumed (a wave owns a long tile range and only ~4 waves share a SIMD, so nothing else
        // hides the load latency: un-prefetched, 46 % of the wave cycles were spent in s_waitcnt).
        constexpr int NA = PER_IND ? NP * 2 * KB : KB;
        float4 g_cur[NP], g_nxt[NP];
        float a_cur[NA], a_nxt[NA];
        auto fetch = [&](int64_t t, float4 *gq, float *aq) {
            const int64_t s = (t << 6) + lane;
            const int64_t sc = s < A.m ? s : A.m - 1;          // clamped index; dead lanes are neutralised below
#pragma unroll
            for (int q = 0; q < NP; ++q) gq[q] = A.slab[((t * A.npairs + pairc[q]) << 6) + lane];
#pragma unroll
            for (int x = 0; x < NA; ++x) aq[x] = ptr[x][sc];
        };
        fetch(t0, g_cur, a_cur);
        for (int64_t t = t0; t < t1; ++t) {
            if (t + 1 < t1) fetch(t + 1, g_nxt, a_nxt);        // wave-uniform
            const bool live = ((t << 6) + lane) < A.m;
            // Lanes past the last SNP (only in the final tile) are given g = (1, 0) and a = 0, for
            // which the site likelihood is exactly 1 and its log exactly 0: no masking per term.
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
                    // all KB slots are computed (slots past K repeat population K-1 and are dropped
                    // in the epilogue): the tile body stays one basic block the scheduler can interleave
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        float v;
                        if (PER_IND) {
                            const float a = live ? a_cur[(q * 2 + h) * KB + j] : 0.0f;
                            if (MODE == WGS_MODE_EXACT) {
                                const double a_d = (double)a;
                                v = site_ll_exact2(g0d, g1d2, g2d, a_d, 1.0 - a_d, tab);
                            } else {
                                v = site_ll_fast(gl[h][0], gl[h][1], g2f, a);
                            }
                        } else {
                            v = MODE == WGS_MODE_EXACT ? site_ll_exact2(g0d, g1d2, g2d, ad[j], oma[j], tab)
                                                       : site_ll_fast(gl[h][0], gl[h][1], g2f, af[j]);
                        }
                        acc[q][h][j] += (double)v;
                    }
                }
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
                        if (lane == 0 && ok[q][h]) atomicAdd(&A.out[(int64_t)ind[q][h] * A.K + kb + j], tot);
                    }
                }
    }
}

// ---- exact partition sums ----------------------------------------------------------------------
// utils.py:147-149: labels = arange(m) % P; np.add.at(zeros(P, float32), labels, per_site_ll) -- a
// SERIAL float32 accumulation per partition in site order.  Reproduced literally: one lane owns one
// (individual, population, partition) chain and walks the partition's sites of this shard in
// order, starting from the float32 carry of the previous shard.  Parallelism is across the
// n x K x P chains only, so the time is ~ (m / P) x one site's latency -- the price of bit-exact
// partition sums; the float64 sums of the sweeps above are the fast alternative.
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

template <int KB, int NP>
int launch_assign_snp(wgs_ctx *ctx, AssignArgs a, int mode)
{
    // waves = pair groups x tile ranges; aim at ~32 waves per CU with ranges of >= 8 tiles
    const int npg = (a.npairs + NP - 1) / NP;
    const int64_t ntiles = wgs_ntiles(a.m);
    int64_t ranges = ((int64_t)ctx->cus * 32 + npg - 1) / npg;
    if (ranges < 1) ranges = 1;
    int64_t tpw = (ntiles + ranges - 1) / ranges;
    if (tpw < 8) tpw = 8;
    if (tpw > ntiles) tpw = ntiles;
    a.tiles_per_wave = (int32_t)tpw;
    ranges = (ntiles + tpw - 1) / tpw;
    const int64_t waves = ranges * npg;
    dim3 grid((unsigned)((waves + 3) / 4));
    const bool per_ind = a.colptr != nullptr;
#define WGS_LAUNCH(M, PI) hipLaunchKernelGGL((assign_snp_kernel<KB, NP, M, PI>), grid, dim3(256), 0, ctx->stream, a)
    if (mode == WGS_MODE_EXACT) {
        if (per_ind) WGS_LAUNCH(WGS_MODE_EXACT, true); else WGS_LAUNCH(WGS_MODE_EXACT, false);
    } else {
        if (per_ind) WGS_LAUNCH(WGS_MODE_FAST, true); else WGS_LAUNCH(WGS_MODE_FAST, false);
    }
#undef WGS_LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
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
    // one upload per process and device (the table lives in the code object's __device__ memory)
    static thread_local int done_for = -1;
    if (done_for == ctx->device) return 0;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(wgs_log_table_dev), wgs_log_table_host, sizeof(double) * 2 * WGS_LOG_N));
    done_for = ctx->device;
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
    // KB = populations per register batch: pick the batch size with the fewest passes, then the least padding
    int best = 4, best_cost = 1 << 30;
    for (int kb = 4; kb <= 8; ++kb) {
        const int passes = (a.K + kb - 1) / kb;
        const int cost = passes * 1000 + passes * kb - a.K;
        if (cost < best_cost) best_cost = cost, best = kb;
    }
    if (a.P == 1) {
        // NP = 2 pairs per wave halves the per-tile frequency loads/conversions per term; measured
        // 174 -> 150 ms at K = 10 (KB = 5).  For KB >= 7 it only costs occupancy (measured equal at
        // K = 8; KB = 10 in one pass measured slower than two passes of 5), so NP = 1 there.
        const bool np2 = a.npairs >= 2;
        switch (best) {
            case 4: return np2 ? launch_assign_snp<4, 2>(ctx, a, mode) : launch_assign_snp<4, 1>(ctx, a, mode);
            case 5: return np2 ? launch_assign_snp<5, 2>(ctx, a, mode) : launch_assign_snp<5, 1>(ctx, a, mode);
            case 6: return np2 ? launch_assign_snp<6, 2>(ctx, a, mode) : launch_assign_snp<6, 1>(ctx, a, mode);
            case 7: return launch_assign_snp<7, 1>(ctx, a, mode);
            default: return launch_assign_snp<8, 1>(ctx, a, mode);
        }
    }
    switch (best) {
        case 4: return launch_assign_kb<4>(ctx, a, mode, grid);
        case 5: return launch_assign_kb<5>(ctx, a, mode, grid);
        case 6: return launch_assign_kb<6>(ctx, a, mode, grid);
        case 7: return launch_assign_kb<7>(ctx, a, mode, grid);
        default: return launch_assign_kb<8>(ctx, a, mode, grid);
    }
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
