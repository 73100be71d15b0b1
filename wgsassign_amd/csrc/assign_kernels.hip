// Assignment log-likelihood (glassy_cy.pyx:12-21 summed as glassy.py:31-42) as CDNA4 kernels.
// Compiled with -ffp-contract=off (see em_kernels.hip).
//
// The reference scans L once per (individual, population) pair: n*K strided passes.  Here ONE
// sweep over a population slab produces every pair:
//   lane <-> individual (slab column): row loads are coalesced 512-byte segments, no transpose;
//   each lane keeps K float64 accumulators (np.sum(..., dtype=float), glassy.py:38) and walks
//   its wave's SNP range; the per-SNP frequency of population k is a wave-uniform (broadcast)
//   load, or a per-lane vector when a per-individual column table is given (leave-one-out).
#include "common.h"

namespace {

// glassy_cy.pyx:18-21 for one (SNP, individual, population), exact rounding sequence; returns
// the float32 the reference stores into loglike_vec[s] (which starts at 0.0f, glassy.py:34).
__device__ __forceinline__ float site_ll_exact(double g0d, double g1d, double g2d, float a)
{
    const double ad = (double)a;
    const double oma = 1.0 - ad;
    const float like0 = (float)((g0d * oma) * oma);
    const float like1 = (float)(((g1d * 2.0) * oma) * ad);
    const float like2 = (float)((g2d * ad) * ad);
    return (float)log((double)((like0 + like1) + like2));
}

__device__ __forceinline__ float site_ll_fast(float g0, float g1, float g2, float a)
{
    const float oma = 1.0f - a;
    const float like0 = g0 * oma * oma;
    const float like1 = g1 * 2.0f * oma * a;
    const float like2 = g2 * a * a;
    return logf((like0 + like1) + like2);
}

template <int KB, int MODE>
__global__ __launch_bounds__(256) void assign_kernel(AssignArgs A)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = blockIdx.y * 64 + lane;
    const bool valid = col < A.ncols;
    const int colc = valid ? col : A.ncols - 1;
    const int ind = A.members[colc];
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    int64_t s0 = w * A.rows_per_wave;
    int64_t s1 = s0 + A.rows_per_wave;
    if (s1 > A.m) s1 = A.m;

    for (int kb = 0; kb < A.K; kb += KB) {
        const float *ptr[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            const int k = kb + j < A.K ? kb + j : A.K - 1;
            ptr[j] = A.colptr ? A.colptr[(int64_t)ind * A.K + k] : A.acol[k];
        }
        if (A.P <= 1) {
            double acc[KB];
#pragma unroll
            for (int j = 0; j < KB; ++j) acc[j] = 0.0;
            for (int64_t s = s0; s < s1; ++s) {
                const float2 g = A.slab[s * A.ld + colc];
                const double g0d = (double)g.x, g1d = (double)g.y;
                const double g2d = (1.0 - g0d) - g1d;
                const float g2f = (1.0f - g.x) - g.y;
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    if (kb + j < A.K) {
                        const float a = ptr[j][s];
                        const float v = MODE == WGS_MODE_EXACT ? site_ll_exact(g0d, g1d, g2d, a)
                                                               : site_ll_fast(g.x, g.y, g2f, a);
                        acc[j] += (double)v;
                    }
                }
            }
            if (valid && s0 < s1) {
#pragma unroll
                for (int j = 0; j < KB; ++j)
                    if (kb + j < A.K) atomicAdd(&A.out[(int64_t)ind * A.K + kb + j], acc[j]);
            }
        } else {
            // utils.py:147-149: label = global site index % P; one accumulator per partition.
            // Sites of one partition are visited in index order within the wave's range.
            for (int p = 0; p < A.P; ++p) {
                double acc[KB];
#pragma unroll
                for (int j = 0; j < KB; ++j) acc[j] = 0.0;
                int64_t first = s0 + ((p - (A.site0 + s0) % A.P) % A.P + A.P) % A.P;
                bool any = false;
                for (int64_t s = first; s < s1; s += A.P) {
                    any = true;
                    const float2 g = A.slab[s * A.ld + colc];
                    const double g0d = (double)g.x, g1d = (double)g.y;
                    const double g2d = (1.0 - g0d) - g1d;
                    const float g2f = (1.0f - g.x) - g.y;
#pragma unroll
                    for (int j = 0; j < KB; ++j) {
                        if (kb + j < A.K) {
                            const float a = ptr[j][s];
                            const float v = MODE == WGS_MODE_EXACT ? site_ll_exact(g0d, g1d, g2d, a)
                                                                   : site_ll_fast(g.x, g.y, g2f, a);
                            acc[j] += (double)v;
                        }
                    }
                }
                if (valid && any) {
#pragma unroll
                    for (int j = 0; j < KB; ++j)
                        if (kb + j < A.K)
                            atomicAdd(&A.out[((int64_t)ind * A.P + p) * A.K + kb + j], acc[j]);
                }
            }
        }
    }
}

// The thin mirror of glassy_cy.loglike: vec[s] = (float)((double)vec[s] + log(...)), one
// individual (its (g0,g1) column compacted to g[m]) and one population (a[m]).
template <int MODE>
__global__ void loglike_site_kernel(const float2 *__restrict__ g, const float *__restrict__ a, float *vec, int64_t m)
{
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
            vec[s] = (float)((double)vec[s] + log((double)((like0 + like1) + like2)));
        } else {
            vec[s] = vec[s] + site_ll_fast(gg.x, gg.y, (1.0f - gg.x) - gg.y, av);
        }
    }
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

int launch_assign(wgs_ctx *ctx, const AssignArgs &a_in, int mode)
{
    AssignArgs a = a_in;
    if (a.m <= 0 || a.ncols <= 0 || a.K <= 0) return 0;
    // Enough waves to fill 256 CUs several times over, but ranges long enough to amortise the
    // per-wave prologue (pointer table) and the atomics.
    const int colblocks = (a.ncols + 63) / 64;
    int64_t want_waves = (int64_t)ctx->cus * 32 / colblocks;
    if (want_waves < 4) want_waves = 4;
    int64_t rpw = (a.m + want_waves - 1) / want_waves;
    if (rpw < 64) rpw = 64;
    if (a.P > 1) rpw = ((rpw + a.P - 1) / a.P) * a.P;
    a.rows_per_wave = (int32_t)(rpw > 0x7fffffff ? 0x7fffffff : rpw);
    const int64_t waves = (a.m + a.rows_per_wave - 1) / a.rows_per_wave;
    dim3 grid((unsigned)((waves + 3) / 4), (unsigned)colblocks);
    if (a.K <= 4) return launch_assign_kb<4>(ctx, a, mode, grid);
    if (a.K <= 8) return launch_assign_kb<8>(ctx, a, mode, grid);
    return launch_assign_kb<16>(ctx, a, mode, grid);
}

int launch_loglike_site(wgs_ctx *ctx, const float2 *g, const float *a, float *vec, int64_t m, int mode)
{
    if (m <= 0) return 0;
    int64_t blocks = (m + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (mode == WGS_MODE_EXACT)
        hipLaunchKernelGGL(loglike_site_kernel<WGS_MODE_EXACT>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g, a, vec, m);
    else
        hipLaunchKernelGGL(loglike_site_kernel<WGS_MODE_FAST>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g, a, vec, m);
    HIP_TRY(hipGetLastError());
    return 0;
}
