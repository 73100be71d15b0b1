// EM allele-frequency update (emMAF_cy.pyx:10-23) and convergence metric (emMAF_cy.pyx:26-33)
// as CDNA4 (gfx950) kernels.  Compiled with -ffp-contract=off: in WGS_MODE_EXACT every
// floating-point operation below is one rounding of the reference's C expression, so no
// contraction or reassociation is allowed; fused operations are written explicitly.
//
// Mapping (exact mode must reproduce a SERIAL float32 accumulation over individuals):
//   lane  <-> SNP          (64 SNPs per wavefront; the serial chain lives in one lane)
//   wave  <-> tile of 64 SNPs x all individuals of one population slab
//   The wave streams its tile through LDS in chunks of T individuals: coalesced 16-byte
//   global loads (8 lanes x 16 B = one 128-byte row segment), ds_write_b128 into a padded
//   image (row stride 144 B -> conflict-free), then each lane walks ITS row with
//   ds_read_b128.  All lanes process the same individual at the same time, so the
//   leave-one-out skip and the column bound are wave-uniform branches.
#include "common.h"

namespace {

constexpr int T = 16;                 // individuals per chunk
constexpr int RS = T * 8 + 16;        // LDS row stride in bytes (odd multiple of 16)
constexpr int WAVES = 4;
typedef float f4 __attribute__((ext_vector_type(4)));   // plain vector type: stays in VGPRs

struct SnpState {
    double fd, omf, fd2;              // f, 1-f, 2f in double (per SNP, hoisted)
    float ff, omff, ff2;              // fast-mode float copies
};

// One (SNP, individual) term of emMAF_cy.pyx:19-22, exact rounding sequence:
//   p0 = (float)(((double)g0*(1.0-f))*(1.0-f))
//   p1 = (float)((((double)g1*2.0)*f)*(1.0-f))          (g1*2.0)*f == g1*(2f): exact scaling
//   p2 = (float)((((1.0-(double)g0)-(double)g1)*f)*f)
//   tmp = (float)((double)tmp + ((double)p1 + 2.0*(double)p2) / (2.0*(double)((p0+p1)+p2)))
__device__ __forceinline__ void term_exact(float g0, float g1, const SnpState &st, float &tmp)
{
    const double g0d = (double)g0, g1d = (double)g1;
    const float p0 = (float)((g0d * st.omf) * st.omf);
    const float p1 = (float)((g1d * st.fd2) * st.omf);
    const float p2 = (float)((((1.0 - g0d) - g1d) * st.fd) * st.fd);
    const float s = (p0 + p1) + p2;
    const double num = (double)p1 + 2.0 * (double)p2;
    const double den = 2.0 * (double)s;
    tmp = (float)((double)tmp + num / den);
}

// Fast mode: the same expression evaluated in float32 (one reciprocal), same accumulation order.
__device__ __forceinline__ void term_fast(float g0, float g1, const SnpState &st, float &tmp)
{
    const float p0 = g0 * st.omff * st.omff;
    const float p1 = g1 * st.ff2 * st.omff;
    const float p2 = ((1.0f - g0) - g1) * st.ff * st.ff;
    const float s = (p0 + p1) + p2;
    const float num = p1 + 2.0f * p2;
    tmp = tmp + num * __builtin_amdgcn_rcpf(2.0f * s);
}

template <int MODE>
__global__ __launch_bounds__(WAVES * 64) void em_sweep_kernel(const FitDesc *__restrict__ fits, int64_t m)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[WAVES * 64 * RS];
    const FitDesc fd = fits[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * WAVES + wave) * 64;
    unsigned char *wl = lds + wave * (64 * RS);

    const int64_t my_row = row0 + lane;
    const int64_t my_row_c = my_row < m ? my_row : m - 1;
    const float f_old = fd.f_old[my_row_c];
    SnpState st;
    st.fd = (double)f_old;
    st.omf = 1.0 - st.fd;
    st.fd2 = 2.0 * st.fd;
    st.ff = f_old;
    st.omff = 1.0f - f_old;
    st.ff2 = 2.0f * f_old;

    // staging: wave instruction q covers rows q*8 .. q*8+7, 8 lanes x 16 B per row
    const int lr = lane >> 3, lc = lane & 7;
    const float2 *src[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int64_t r = row0 + q * 8 + lr;
        if (r >= m) r = m - 1;
        src[q] = fd.slab + r * fd.ld + lc * 2;
    }
    const int nchunks = (fd.ncols + T - 1) / T;
    f4 stage[8];   // slabs are never empty (wgs_em_create rejects empty groups), so chunk 0 exists
#pragma unroll
    for (int q = 0; q < 8; ++q) stage[q] = *reinterpret_cast<const f4 *>(src[q]);

    float tmp = 0.0f;
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();   // previous chunk's LDS reads are done (WAR)
#pragma unroll
        for (int q = 0; q < 8; ++q)
            *reinterpret_cast<f4 *>(wl + (q * 8 + lr) * RS + lc * 16) = stage[q];
        if (c + 1 < nchunks) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                stage[q] = *reinterpret_cast<const f4 *>(src[q] + (c + 1) * T);
        }
        __syncthreads();   // tile visible (RAW)
        const int i0 = c * T;
#pragma unroll
        for (int jj = 0; jj < T / 2; ++jj) {
            const f4 v = *reinterpret_cast<const f4 *>(wl + lane * RS + jj * 16);
            const int ia = i0 + 2 * jj, ib = ia + 1;
            if (ia < fd.ncols && ia != fd.skip) {
                if (MODE == WGS_MODE_EXACT) term_exact(v.x, v.y, st, tmp); else term_fast(v.x, v.y, st, tmp);
            }
            if (ib < fd.ncols && ib != fd.skip) {
                if (MODE == WGS_MODE_EXACT) term_exact(v.z, v.w, st, tmp); else term_fast(v.z, v.w, st, tmp);
            }
        }
    }
    const float f_new = tmp / (float)fd.n_eff;           // emMAF_cy.pyx:23 (float32 divide)
    double sq = 0.0;
    if (my_row < m) {
        fd.f_new[my_row] = f_new;
        const float d = f_new - f_old;                    // emMAF_cy.pyx:31, float32
        sq = (double)(d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if (lane == 0) atomicAdd(fd.ssq, sq);
}

__global__ void fill_kernel(float *p, int64_t count, float v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) p[i] = v;
}

// WGSassign.py:236-240: af[af < lo] = lo; af[af > hi] = hi  (float32 compares; NaN untouched)
__global__ void clamp_kernel(float *p, int64_t count, float lo, float hi)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        float v = p[i];
        if (v < lo) v = lo;
        if (v > hi) v = hi;
        p[i] = v;
    }
}

// emMAF_cy.pyx:30-31 continued over [0, m): res = res + (a-b)*(a-b), float32, index order.
// One wavefront: 64 lanes form the squares of 1024 elements in parallel (each square is an
// independent float32 sub+mul, identical to the serial code), lane 0 adds them in order.
__global__ __launch_bounds__(64) void rmse_chain_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                         int64_t m, float carry, float *out)
{
    __shared__ float sq[1024];
    const int lane = threadIdx.x;
    float res = carry;
    for (int64_t base = 0; base < m; base += 1024) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t i = base + k * 64 + lane;
            float v = 0.0f;
            if (i < m) {
                const float d = a[i] - b[i];
                v = d * d;
            }
            sq[k * 64 + lane] = v;
        }
        __syncthreads();
        if (lane == 0) {
            const int cnt = (m - base) < 1024 ? (int)(m - base) : 1024;
            for (int t = 0; t < cnt; ++t) res = res + sq[t];
        }
        __syncthreads();
    }
    if (lane == 0) *out = res;
}

}  // namespace

int launch_em_sweep(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, int64_t m, int mode)
{
    if (n_fits <= 0 || m <= 0) return 0;
    const int64_t tiles = (m + 63) / 64;
    dim3 grid((unsigned)((tiles + WAVES - 1) / WAVES), (unsigned)n_fits);
    WGS_REQUIRE(n_fits <= 65535, "em sweep: more than 65535 fits in one launch (%d)", n_fits);
    if (mode == WGS_MODE_EXACT)
        hipLaunchKernelGGL(em_sweep_kernel<WGS_MODE_EXACT>, grid, dim3(WAVES * 64), 0, ctx->stream, d_descs, m);
    else
        hipLaunchKernelGGL(em_sweep_kernel<WGS_MODE_FAST>, grid, dim3(WAVES * 64), 0, ctx->stream, d_descs, m);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_fill(wgs_ctx *ctx, float *p, int64_t count, float v)
{
    if (count <= 0) return 0;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, count, v);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_clamp(wgs_ctx *ctx, float *p, int64_t count, float lo, float hi)
{
    if (count <= 0) return 0;
    int64_t blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, p, count, lo, hi);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_rmse_chain(wgs_ctx *ctx, const float *a, const float *b, int64_t m, float carry_in, float *d_out)
{
    hipLaunchKernelGGL(rmse_chain_kernel, dim3(1), dim3(64), 0, ctx->stream, a, b, m, carry_in, d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}
