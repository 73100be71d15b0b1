// Data-movement kernels: host-layout rows <-> population slabs, (m,K) <-> K vectors, and the
// on-device synthetic Beagle generator used by bench.py.
#include "common.h"

namespace {

// Slab element (SNP s, local column c) as a float2 index into the tile-interleaved slab
// (common.h: Slab): tile s/64, pair c/2, lane s%64, half c&1.
__device__ __forceinline__ int64_t slab_index(int64_t s, int c, int npairs)
{
    return ((((s >> 6) * npairs + (c >> 1)) << 6) + (s & 63)) * 2 + (c & 1);
}

// rows: (nrows, 2n) float32 in the reference's host layout (reader_cy.pyx:71-77);
// element (r, i) goes to slab[group_of[i]] at (SNP row0 + r, column col_of[i]).
__global__ void scatter_rows_kernel(const float2 *__restrict__ rows, int64_t nrows, int64_t n, int64_t row0,
                                    const int32_t *__restrict__ group_of, const int32_t *__restrict__ col_of,
                                    float4 *const *__restrict__ base, const int32_t *__restrict__ npairs)
{
    const int64_t total = nrows * n;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < total; e += stride) {
        const int64_t r = e / n;
        const int i = (int)(e - r * n);
        const int g = group_of[i];
        reinterpret_cast<float2 *>(base[g])[slab_index(row0 + r, col_of[i], npairs[g])] = rows[e];
    }
}

__global__ void gather_rows_kernel(float2 *__restrict__ rows, int64_t nrows, int64_t n, int64_t row0,
                                   const int32_t *__restrict__ group_of, const int32_t *__restrict__ col_of,
                                   float4 *const *__restrict__ base, const int32_t *__restrict__ npairs)
{
    const int64_t total = nrows * n;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < total; e += stride) {
        const int64_t r = e / n;
        const int i = (int)(e - r * n);
        const int g = group_of[i];
        rows[e] = reinterpret_cast<const float2 *>(base[g])[slab_index(row0 + r, col_of[i], npairs[g])];
    }
}

// (m, K) row-major <-> K vectors of m.  K is small (5..20): one thread per SNP.
__global__ void mK_to_Km_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t m, int K)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; s < m; s += stride)
        for (int k = 0; k < K; ++k) dst[(int64_t)k * m + s] = src[s * K + k];
}
__global__ void Km_to_mK_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t m, int K)
{
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; s < m; s += stride)
        for (int k = 0; k < K; ++k) dst[s * K + k] = src[(int64_t)k * m + s];
}

// ---- Philox-4x32-10 counter RNG
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// Per-(SNP, group) population frequency: ancestral p ~ arcsine-ish U-shape, drifted by
// N(0, 0.08^2) per group, clipped to [0.01, 0.99].  Pure function of (seed, global SNP, group).
__device__ __forceinline__ float pop_freq(uint64_t seed, int64_t gsnp, int g)
{
    uint32_t c[4] = {(uint32_t)gsnp, (uint32_t)(gsnp >> 32), 0xA5A5A5A5u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u = u01(c[0]);
    const float anc = 0.5f - 0.5f * cospif(u);                       // Beta(1/2,1/2)
    uint32_t d[4] = {(uint32_t)gsnp, (uint32_t)(gsnp >> 32), 0x5A5A5A5Au, (uint32_t)g};
    philox4x32_10(d, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float z = sqrtf(-2.0f * logf(u01(d[0]))) * cospif(2.0f * u01(d[1]));
    float p = anc + 0.08f * z;
    return fminf(fmaxf(p, 0.01f), 0.99f);
}

// (g0, g1) of one (SNP, individual): genotype ~ Binomial(2, p); depth ~ Poisson(depth)
// truncated at 15; alt reads ~ Binomial(depth, {e, 1/2, 1-e}[g]); GL_g ∝ P(reads | g), normalised,
// rounded to 6 decimals like the ANGSD text the reference parses with atof.
__device__ float2 synth_gl(uint64_t seed, int64_t gsnp, int ind, float p, float lam, float cdf0)
{
    uint32_t c[4] = {(uint32_t)gsnp, (uint32_t)(gsnp >> 32), (uint32_t)ind, 1u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int geno = (u01(c[0]) < p) + (u01(c[1]) < p);
    float cdf = cdf0, pm = cdf0;              // Poisson by inversion
    const float ud = u01(c[2]);
    int d = 0;
    while (ud > cdf && d < 15) { ++d; pm *= lam / (float)d; cdf += pm; }
    const float qalt = geno == 0 ? 0.01f : (geno == 1 ? 0.5f : 0.99f);
    // up to 15 read draws from the remaining 32 + 3*32 random bits: 8 bits each
    uint32_t bits[4] = {c[3], c[0] ^ 0x9E3779B9u, c[1] ^ 0xBB67AE85u, c[2] ^ 0x85EBCA6Bu};
    int alt = 0;
    for (int r = 0; r < d; ++r) {
        const uint32_t w = bits[r >> 2] >> ((r & 3) * 8);
        alt += ((float)(w & 0xFF) + 0.5f) * (1.0f / 256.0f) < qalt;
    }
    const int ref = d - alt;
    const double e1 = 0.01, e0 = 0.99;
    double l0 = 1.0, l1 = 1.0, l2 = 1.0;
    for (int r = 0; r < ref; ++r) { l0 *= e0; l2 *= e1; }
    for (int r = 0; r < alt; ++r) { l0 *= e1; l2 *= e0; }
    for (int r = 0; r < d; ++r) l1 *= 0.5;
    const double tot = l0 + l1 + l2;
    return make_float2((float)(rint(l0 / tot * 1e6) / 1e6), (float)(rint(l1 / tot * 1e6) / 1e6));
}

// The same with QUALITY-DEPENDENT likelihoods (what ANGSD -GL 2 writes for real reads; tests/synth.py: make_beagle_quality is the
// NumPy twin of the model, not of the random stream): every read draws its base quality from `nq` bins (error e = 10^(-Q/10),
// cumulative probabilities qcdf), P(read | genotype) = 1-e / e/3 for the homozygotes and their mean for the heterozygote.
struct QualBins {
    int nq;
    float e[8], cdf[8];
};
__device__ float2 synth_gl_quality(uint64_t seed, int64_t gsnp, int ind, float p, float lam, float cdf0, const QualBins &qb)
{
    uint32_t c[4] = {(uint32_t)gsnp, (uint32_t)(gsnp >> 32), (uint32_t)ind, 1u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int geno = (u01(c[0]) < p) + (u01(c[1]) < p);
    float cdf = cdf0, pm = cdf0;
    const float ud = u01(c[2]);
    int d = 0;
    while (ud > cdf && d < 15) { ++d; pm *= lam / (float)d; cdf += pm; }
    uint32_t r[4] = {(uint32_t)gsnp, (uint32_t)(gsnp >> 32), (uint32_t)ind, 2u};
    philox4x32_10(r, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t bits[8] = {c[3], c[0] ^ 0x9E3779B9u, c[1] ^ 0xBB67AE85u, c[2] ^ 0x85EBCA6Bu, r[0], r[1], r[2], r[3]};
    double l0 = 1.0, l1 = 1.0, l2 = 1.0;
    for (int k = 0; k < d; ++k) {
        const uint32_t w = bits[k >> 1] >> ((k & 1) * 16);               // 16 bits per read: 8 for the allele, 8 for the quality
        const float uq = ((float)((w >> 8) & 0xFF) + 0.5f) * (1.0f / 256.0f);
        int q = 0;
        while (q + 1 < qb.nq && uq > qb.cdf[q]) ++q;
        const float e = qb.e[q];
        const float from_alt = geno * 0.5f;
        const float ua = ((float)(w & 0xFF) + 0.5f) * (1.0f / 256.0f);
        // the read shows the alternative allele: drawn from the genotype, flipped with probability e
        const float p_alt = from_alt * (1.0f - e) + (1.0f - from_alt) * e;
        const bool alt = ua < p_alt;
        const double ok = 1.0 - (double)e, bad = (double)e / 3.0;
        l0 *= alt ? bad : ok;
        l1 *= 0.5 * ok + 0.5 * bad;
        l2 *= alt ? ok : bad;
    }
    const double tot = l0 + l1 + l2;
    return make_float2((float)(rint(l0 / tot * 1e6) / 1e6), (float)(rint(l1 / tot * 1e6) / 1e6));
}

__global__ void synth_quality_kernel(float4 *__restrict__ slab, int64_t m, int npairs, int ncols, const int32_t *__restrict__ members,
                                     int group, int64_t site0, uint64_t seed, float depth, QualBins qb)
{
    const int64_t total = ((m + 63) / 64) * npairs * 64;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float cdf0 = expf(-depth);
    for (; e < total; e += stride) {
        const int lane = (int)(e & 63);
        const int64_t tp = e >> 6;
        const int64_t t = tp / npairs;
        const int pr = (int)(tp - t * npairs);
        const int64_t s = t * 64 + lane;
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < m) {
            const int64_t gsnp = site0 + s;
            const float p = pop_freq(seed, gsnp, group);
            const float2 a = synth_gl_quality(seed, gsnp, members[2 * pr], p, depth, cdf0, qb);
            float2 b = make_float2(0.f, 0.f);
            if (2 * pr + 1 < ncols) b = synth_gl_quality(seed, gsnp, members[2 * pr + 1], p, depth, cdf0, qb);
            out = make_float4(a.x, a.y, b.x, b.y);
        }
        slab[e] = out;
    }
}

// One thread per slab float4 = (tile, pair, lane): both individuals of the pair for one SNP.
// Writes are one contiguous 1 KiB per wave.  SNPs beyond m (last tile) get zeros.
__global__ void synth_kernel(float4 *__restrict__ slab, int64_t m, int npairs, int ncols, const int32_t *__restrict__ members,
                             int group, int64_t site0, uint64_t seed, float depth)
{
    const int64_t total = ((m + 63) / 64) * npairs * 64;
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float cdf0 = expf(-depth);
    for (; e < total; e += stride) {
        const int lane = (int)(e & 63);
        const int64_t tp = e >> 6;
        const int64_t t = tp / npairs;
        const int pr = (int)(tp - t * npairs);
        const int64_t s = t * 64 + lane;
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < m) {
            const int64_t gsnp = site0 + s;
            const float p = pop_freq(seed, gsnp, group);
            const float2 a = synth_gl(seed, gsnp, members[2 * pr], p, depth, cdf0);
            float2 b = make_float2(0.f, 0.f);
            if (2 * pr + 1 < ncols) b = synth_gl(seed, gsnp, members[2 * pr + 1], p, depth, cdf0);
            out = make_float4(a.x, a.y, b.x, b.y);
        }
        slab[e] = out;
    }
}

inline unsigned grid_for(int64_t total)
{
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

}  // namespace

int launch_scatter_rows(wgs_beagle *b, const float *d_rows, int64_t row0, int64_t nrows)
{
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(nrows * b->n)), dim3(256), 0, b->ctx->stream,
                       reinterpret_cast<const float2 *>(d_rows), nrows, b->n, row0, b->d_group_of, b->d_col_of,
                       b->d_base, b->d_npairs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gather_rows(wgs_beagle *b, float *d_rows, int64_t row0, int64_t nrows)
{
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(nrows * b->n)), dim3(256), 0, b->ctx->stream,
                       reinterpret_cast<float2 *>(d_rows), nrows, b->n, row0, b->d_group_of, b->d_col_of, b->d_base,
                       b->d_npairs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_synth(wgs_beagle *b, uint64_t seed, double depth)
{
    for (int g = 0; g < b->n_groups; ++g) {
        Slab &sl = b->slabs[g];
        if (sl.ncols == 0) continue;
        hipLaunchKernelGGL(synth_kernel, dim3(grid_for(wgs_ntiles(b->m) * sl.npairs * 64)), dim3(256), 0, b->ctx->stream,
                           sl.base, b->m, sl.npairs, sl.ncols, sl.d_members, g, b->site0, seed, (float)depth);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

int launch_synth_quality(wgs_beagle *b, uint64_t seed, double depth, int nq, const double *quals, const double *probs)
{
    WGS_REQUIRE(nq >= 1 && nq <= 8 && quals && probs, "synthetic base qualities: 1 to 8 bins");
    QualBins qb;
    qb.nq = nq;
    double tot = 0.0, run = 0.0;
    for (int i = 0; i < nq; ++i) tot += probs[i];
    WGS_REQUIRE(tot > 0, "synthetic base qualities: probabilities sum to zero");
    for (int i = 0; i < 8; ++i) {
        if (i < nq) run += probs[i] / tot;
        qb.e[i] = i < nq ? (float)pow(10.0, -quals[i] / 10.0) : 0.0f;
        qb.cdf[i] = i < nq ? (float)run : 1.0f;
    }
    for (int g = 0; g < b->n_groups; ++g) {
        Slab &sl = b->slabs[g];
        if (sl.ncols == 0) continue;
        hipLaunchKernelGGL(synth_quality_kernel, dim3(grid_for(wgs_ntiles(b->m) * sl.npairs * 64)), dim3(256), 0, b->ctx->stream,
                           sl.base, b->m, sl.npairs, sl.ncols, sl.d_members, g, b->site0, seed, (float)depth, qb);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

int launch_transpose_mK_to_Km(wgs_ctx *ctx, const float *src_mK, float *dst_Km, int64_t m, int32_t K)
{
    if (m <= 0) return 0;
    hipLaunchKernelGGL(mK_to_Km_kernel, dim3(grid_for(m)), dim3(256), 0, ctx->stream, src_mK, dst_Km, m, K);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_transpose_Km_to_mK(wgs_ctx *ctx, const float *src_Km, float *dst_mK, int64_t m, int32_t K)
{
    if (m <= 0) return 0;
    hipLaunchKernelGGL(Km_to_mK_kernel, dim3(grid_for(m)), dim3(256), 0, ctx->stream, src_Km, dst_mK, m, K);
    HIP_TRY(hipGetLastError());
    return 0;
}
