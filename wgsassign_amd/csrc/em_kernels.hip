// EM allele-frequency update (emMAF_cy.pyx:10-23) and convergence metric (emMAF_cy.pyx:26-33)
// as CDNA4 (gfx950) kernels.  Compiled with -ffp-contract=off: in WGS_MODE_EXACT every
// floating-point operation below is one rounding of the reference's C expression, so no
// contraction or reassociation is allowed; fused operations are written explicitly.
//
// Mapping (exact mode must reproduce a SERIAL float32 accumulation over individuals):
//   lane <-> SNP           (64 SNPs per wavefront; the serial chain lives in one lane)
//   wave <-> tile of 64 SNPs x all individuals of one population slab
//   The slab is stored tile-interleaved (common.h: Slab), so the wave's k-th load instruction
//   is ONE aligned contiguous 1 KiB: lane l receives (g0,g1) of individuals 2k and 2k+1 of ITS
//   SNP.  No LDS, no barriers, no cross-lane traffic; HBM bytes == algorithmic bytes.  All
//   lanes process the same individual at the same time, so the leave-one-out skip and the
//   column bound are wave-uniform branches.  Loads for the next U pairs are issued before the
//   current U pairs are consumed (register double buffer).
#include "common.h"

namespace {

constexpr int U = 4;                  // individual pairs per register buffer (2 x 4 KiB in flight per wave)
constexpr int WAVES = 4;
typedef float f4 __attribute__((ext_vector_type(4)));   // plain vector type: stays in VGPRs
// Pointers read out of the descriptor table are generic to the compiler (-> flat_load, which
// cannot be pipelined with counted waits); they are device-global by construction.
typedef const f4 __attribute__((address_space(1))) *gf4_ptr;
typedef const float __attribute__((address_space(1))) *gf32_ptr;
typedef float __attribute__((address_space(1))) *gf32_wptr;

struct SnpState {
    double fd, omf, fd2;              // f, 1-f, 2f in double (per SNP, hoisted)
    float ff, omff, ff2;              // fast-mode float copies
};

// Correctly rounded double quotient num/den for den = a float32 value (or 0) and 0 <= num:
// the same Newton-Raphson core LLVM emits for an IEEE f64 divide (v_rcp_f64 seed, two
// refinements, quotient, one residual correction), minus the v_div_scale range scaling that
// operands of this magnitude never need; v_div_fixup keeps the 0/0, x/0 and NaN results of `/`.
__device__ __forceinline__ double div_exact(double num, double den)
{
#ifdef WGS_PLAIN_DIVIDE
    return num / den;
#else
    double r = __builtin_amdgcn_rcp(den);
    double e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-den, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = num * r;
    const double rem = __builtin_fma(-den, q, num);
    q = __builtin_fma(rem, r, q);
    return __builtin_amdgcn_div_fixup(q, den, num);
#endif
}

// One (SNP, individual) term of emMAF_cy.pyx:19-22, exact rounding sequence:
//   p0 = (float)(((double)g0*(1.0-f))*(1.0-f))
//   p1 = (float)((((double)g1*2.0)*f)*(1.0-f))          (g1*2.0)*f == g1*(2f): exact scaling
//   p2 = (float)((((1.0-(double)g0)-(double)g1)*f)*f)
//   tmp = (float)((double)tmp + ((double)p1 + 2.0*(double)p2) / (2.0*(double)((p0+p1)+p2)))
// Scalings by 2 are exact, so 2.0*p2 + p1 is one fma and x/(2s) == 0.5*(x/s) folds into the
// accumulation's fma: every remaining operation is one rounding of the reference's expression.
__device__ __forceinline__ void term_exact(float g0, float g1, const SnpState &st, float &tmp)
{
    const double g0d = (double)g0, g1d = (double)g1;
    const float p0 = (float)((g0d * st.omf) * st.omf);
    const float p1 = (float)((g1d * st.fd2) * st.omf);
    const float p2 = (float)((((1.0 - g0d) - g1d) * st.fd) * st.fd);
    const float s = (p0 + p1) + p2;
    const double num = __builtin_fma(2.0, (double)p2, (double)p1);
    const double q = div_exact(num, (double)s);
    tmp = (float)__builtin_fma(0.5, q, (double)tmp);
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
    const FitDesc fd = fits[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t row0 = tile * 64;
    if (row0 >= m) return;                       // wave-uniform; there are no barriers below

    const int64_t my_row = row0 + lane;
    const int64_t my_row_c = my_row < m ? my_row : m - 1;
    const float f_old = ((gf32_ptr)fd.f_old)[my_row_c];
    SnpState st;
    st.fd = (double)f_old;
    st.omf = 1.0 - st.fd;
    st.fd2 = 2.0 * st.fd;
    st.ff = f_old;
    st.omff = 1.0f - f_old;
    st.ff2 = 2.0f * f_old;

    const int npairs = fd.npairs;
    gf4_ptr src = (gf4_ptr)fd.slab + tile * npairs * 64 + lane;
    const int last = npairs - 1;
    f4 cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = src[(u < last ? u : last) * 64];   // clamped: tail re-reads hit cache
    float tmp = 0.0f;
    for (int p0 = 0; p0 < npairs; p0 += U) {
        if (p0 + U < npairs) {                   // one wave-uniform branch per buffer, loads unconditional
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = p0 + U + u;
                nxt[u] = src[(pp < last ? pp : last) * 64];
            }
        }
        // buffers whose 2U individuals are all present take the branch-free path; the buffer holding
        // the left-out individual (LOO) or the odd tail checks each individual (wave-uniform)
        const bool plain = 2 * (p0 + U) <= fd.ncols && (fd.skip < 2 * p0 || fd.skip >= 2 * (p0 + U));
        if (plain) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f4 v = cur[u];
                if (MODE == WGS_MODE_EXACT) {
                    term_exact(v.x, v.y, st, tmp);
                    term_exact(v.z, v.w, st, tmp);
                } else {
                    term_fast(v.x, v.y, st, tmp);
                    term_fast(v.z, v.w, st, tmp);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f4 v = cur[u];
                const int ia = 2 * (p0 + u), ib = ia + 1;
                if (ia < fd.ncols && ia != fd.skip) {
                    if (MODE == WGS_MODE_EXACT) term_exact(v.x, v.y, st, tmp); else term_fast(v.x, v.y, st, tmp);
                }
                if (ib < fd.ncols && ib != fd.skip) {
                    if (MODE == WGS_MODE_EXACT) term_exact(v.z, v.w, st, tmp); else term_fast(v.z, v.w, st, tmp);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    const float f_new = tmp / (float)fd.n_eff;           // emMAF_cy.pyx:23 (float32 divide)
    double sq = 0.0;
    if (my_row < m) {
        ((gf32_wptr)fd.f_new)[my_row] = f_new;
        const float d = f_new - f_old;                    // emMAF_cy.pyx:31, float32
        sq = (double)(d * d);
    }
    // Per-tile partial sum, stored (not atomically added: a million same-address float64 atomics
    // serialise at ~12 ns each and would dominate the sweep); ssq_reduce_kernel adds them in a
    // fixed order, so the sums are reproducible run to run.
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if (lane == 0) fd.ssq_part[tile] = sq;
}

// ssq[fit] = sum over tiles of the per-tile partials, fixed summation order.
__global__ __launch_bounds__(256) void ssq_reduce_kernel(const FitDesc *__restrict__ fits, int64_t ntiles)
{
    __shared__ double red[256];
    const FitDesc fd = fits[blockIdx.x];
    double acc = 0.0;
    for (int64_t t = threadIdx.x; t < ntiles; t += 256) acc += fd.ssq_part[t];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *fd.ssq = red[0];
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

int launch_ssq_reduce(wgs_ctx *ctx, const FitDesc *d_descs, int32_t n_fits, int64_t m)
{
    if (n_fits <= 0 || m <= 0) return 0;
    hipLaunchKernelGGL(ssq_reduce_kernel, dim3((unsigned)n_fits), dim3(256), 0, ctx->stream, d_descs, wgs_ntiles(m));
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
