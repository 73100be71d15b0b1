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

// ---- class codes (common.h: wgs_codes) ---------------------------------------------------------------------
// lane <-> SNP, one wavefront per tile: every lane walks the individuals of ITS SNP through all slabs (the slabs'
// native coalesced loads) and keeps the distinct (g0, g1) bit patterns in a private open-addressing table of 64 slots
// in LDS (slot-major, so a wave-wide access with per-lane slots is conflict-free); occupancy is a 64-bit mask in
// registers.  A class's id is the number of occupied slots below its slot -- known once the walk is complete: the walk
// over the slabs writes slot numbers into the code words, a second walk over the code words alone (an eighth of the
// bytes) turns them into ids.  More than 64 classes in one SNP: not codable (ncls = 255).
struct ClassTable {
    uint64_t *keys;                // LDS: [slot * 64 + lane]
    uint64_t mask = 0;
    bool overflow = false;
    __device__ __forceinline__ static int home(uint64_t key) { return (int)((key * 0x9E3779B97F4A7C15ull) >> 58); }
    __device__ __forceinline__ int find_or_insert(uint64_t key, int lane, bool insert)
    {
        int h = home(key);
        for (int probes = 0; probes < 64; ++probes) {
            if (!((mask >> h) & 1)) {
                if (!insert) return -1;
                keys[h * 64 + lane] = key;
                mask |= 1ull << h;
                return h;
            }
            if (keys[h * 64 + lane] == key) return h;
            h = (h + 1) & 63;
        }
        overflow = true;
        return -1;
    }
    __device__ __forceinline__ int id_of(int slot) const { return slot < 0 ? 0 : __popcll(mask & ((1ull << slot) - 1)); }
};

__device__ __forceinline__ uint64_t gl_key(float g0, float g1) { return ((uint64_t)__float_as_uint(g0) << 32) | __float_as_uint(g1); }

struct EncodeArgs {
    float4 *const *base;           // device: slab bases
    const int32_t *npairs, *ncols; // device: per slab
    int32_t n_slabs;
    int64_t m;
    // outputs (encode pass): nullptr in the counting pass
    float2 *dict;
    int32_t cmax;
    const SlabCodes *slabs;
    uint8_t *ncls;
};

__global__ __launch_bounds__(64) void class_encode_kernel(EncodeArgs A)
{
    __shared__ uint64_t keys[64 * 64];
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t snp = tile * 64 + lane;
    ClassTable T;
    T.keys = keys;
    constexpr int PF = 8;              // pair loads in flight (one wave per workgroup: nothing else hides their latency)
    // the walk over the float32 slabs: every (g0, g1) is looked up / inserted once and its SLOT goes into the code word; the
    // slots each slab uses are remembered in its `present` word
    for (int g = 0; g < A.n_slabs; ++g) {
        const int np = A.npairs[g], nc = A.ncols[g];
        const SlabCodes sc = A.slabs[g];
        const float4 *src = A.base[g] + tile * np * 64 + lane;
        uint64_t used = 0;
        for (int q0 = 0; q0 < sc.nquads; q0 += PF / 2) {
            float4 v[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) v[u] = src[(int64_t)(2 * q0 + u < np ? 2 * q0 + u : np - 1) * 64];
#pragma unroll
            for (int x = 0; x < PF / 2; ++x) {
                const int q = q0 + x;
                if (q >= sc.nquads) break;
                uint32_t word = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = 2 * q + h;
                    if (p >= np) break;
                    const float4 vv = v[2 * x + h];
                    int s0 = T.find_or_insert(gl_key(vv.x, vv.y), lane, true);
                    s0 = s0 < 0 ? 0 : s0;                        // (overflow: the codes are dropped by the host)
                    word |= (uint32_t)s0 << (16 * h);
                    used |= 1ull << s0;
                    if (2 * p + 1 < nc) {
                        int s1 = T.find_or_insert(gl_key(vv.z, vv.w), lane, true);
                        s1 = s1 < 0 ? 0 : s1;
                        word |= (uint32_t)s1 << (16 * h + 8);
                        used |= 1ull << s1;
                    }
                }
                sc.codes[(tile * sc.nquads + q) * 64 + lane] = word;
            }
        }
        sc.present[snp] = used;
    }
    const int n = T.overflow ? 255 : __popcll(T.mask);
    A.ncls[snp] = (uint8_t)n;          // the arrays cover whole tiles
    if (__any(T.overflow)) return;     // not codable: the host sees ncls = 255 and drops the codes
    // the dictionary, class id = rank of the slot among the occupied ones
    for (uint64_t left = T.mask; left;) {
        const int slot = __builtin_ctzll(left);
        left &= left - 1;
        const uint64_t key = keys[slot * 64 + lane];
        A.dict[(tile * WGS_CODE_ROWS + T.id_of(slot)) * 64 + lane] = make_float2(__uint_as_float((uint32_t)(key >> 32)), __uint_as_float((uint32_t)key));
    }
    // second walk, over the code words only (an eighth of the slabs' bytes, written moments ago): slot -> class id
    for (int g = 0; g < A.n_slabs; ++g) {
        const SlabCodes sc = A.slabs[g];
        uint64_t present = 0;
        for (uint64_t left = sc.present[snp]; left;) {
            const int slot = __builtin_ctzll(left);
            left &= left - 1;
            present |= 1ull << T.id_of(slot);
        }
        sc.present[snp] = present;
        uint32_t *cw = sc.codes + tile * sc.nquads * 64 + lane;
        for (int q0 = 0; q0 < sc.nquads; q0 += PF) {
            uint32_t w[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) w[u] = cw[(int64_t)(q0 + u < sc.nquads ? q0 + u : sc.nquads - 1) * 64];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (q0 + u >= sc.nquads) break;
                uint32_t o = 0;
#pragma unroll
                for (int h = 0; h < 4; ++h) o |= (uint32_t)T.id_of((int)((w[u] >> (8 * h)) & 63u)) << (8 * h);
                cw[(int64_t)(q0 + u) * 64] = o;
            }
        }
    }
}

// ---- the slabs' own class numbering (common.h: SlabLocal) --------------------------------------------------------
// hist[slab * 8 + k] = tiles of the slab whose richest SNP shows 8 k + 1 .. 8 k + 8 classes among the slab's individuals
__global__ __launch_bounds__(256) void slab_rows_hist_kernel(const SlabCodes *slabs, int64_t tiles, unsigned long long *hist)
{
    const SlabCodes sc = slabs[blockIdx.y];
    if (sc.nquads == 0) return;
    const int lane = threadIdx.x & 63;
    unsigned count[8] = {0, 0, 0, 0, 0, 0, 0, 0};              // this wavefront's tiles per bin (kept in lane 0; one atomic per bin at the end)
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += (int64_t)gridDim.x * 4) {
        int best = __popcll(sc.present[t * 64 + lane]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
        const int bin = best > 0 ? (best - 1) >> 3 : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) count[k] += bin == k ? 1u : 0u;
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (count[k]) atomicAdd(hist + blockIdx.y * 8 + k, (unsigned long long)count[k]);
    }
}

// One wavefront per tile of one slab, lane <-> SNP: the dictionary rows of the classes present, in rank order; the code
// words with every class replaced by its rank.
__global__ __launch_bounds__(64) void local_encode_kernel(SlabCodes sc, const float2 *__restrict__ dict, uint32_t *__restrict__ lcodes,
                                                          float2 *__restrict__ ldict, int rows)
{
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const uint64_t present = sc.present[tile * 64 + lane];
    uint64_t left = present;
    for (int r = 0; r < rows; ++r) {                           // (a SNP with more classes than rows: its tile is swept directly)
        float2 v = make_float2(0.0f, 0.0f);
        if (left) {
            const int c = __builtin_ctzll(left);
            left &= left - 1;
            v = dict[(tile * WGS_CODE_ROWS + c) * 64 + lane];
        }
        ldict[(tile * rows + r) * 64 + lane] = v;
    }
    const uint32_t *src = sc.codes + tile * sc.nquads * 64 + lane;
    uint32_t *dst = lcodes + tile * sc.nquads * 64 + lane;
    for (int q = 0; q < sc.nquads; ++q) {
        const uint32_t w = src[(int64_t)q * 64];
        uint32_t o = 0;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const unsigned c = (w >> (8 * h)) & 255u;
            o |= (uint32_t)__popcll(present & ((1ull << c) - 1ull)) << (8 * h);
        }
        dst[(int64_t)q * 64] = o;
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

static EncodeArgs encode_args(wgs_beagle *b, int32_t *d_ncols)
{
    EncodeArgs A;
    A.base = b->d_base;
    A.npairs = b->d_npairs;
    A.ncols = d_ncols;
    A.n_slabs = b->n_groups;
    A.m = b->m;
    A.dict = nullptr;
    A.cmax = 0;
    A.slabs = nullptr;
    A.ncls = nullptr;
    return A;
}

int launch_class_encode(wgs_beagle *b, wgs_codes *c)
{
    std::vector<int32_t> ncols(b->n_groups);
    for (int g = 0; g < b->n_groups; ++g) ncols[g] = b->slabs[g].ncols;
    void *ws = nullptr;
    if (wgs_ctx_workspace(b->ctx, sizeof(int32_t) * b->n_groups, &ws)) return 1;
    HIP_TRY(hipMemcpyAsync(ws, ncols.data(), sizeof(int32_t) * b->n_groups, hipMemcpyHostToDevice, b->ctx->stream));
    EncodeArgs A = encode_args(b, reinterpret_cast<int32_t *>(ws));
    A.dict = c->dict;
    A.cmax = c->cmax;
    A.slabs = c->d_slabs;
    A.ncls = c->ncls;
    hipLaunchKernelGGL(class_encode_kernel, dim3((unsigned)wgs_ntiles(b->m)), dim3(64), 0, b->ctx->stream, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return 0;
}

// Allocates and fills wgs_codes::local for every slab (0 = done; a failed allocation is reported, the caller falls back).
int launch_local_encode(wgs_beagle *b, wgs_codes *c)
{
    const int64_t tiles = wgs_ntiles(b->m);
    const int G = b->n_groups;
    void *ws = nullptr;
    if (wgs_ctx_workspace(b->ctx, sizeof(unsigned long long) * 8 * G, &ws)) return 1;
    unsigned long long *d_hist = reinterpret_cast<unsigned long long *>(ws);
    HIP_TRY(hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * 8 * G, b->ctx->stream));
    // (a few hundred workgroups per slab: every wavefront then counts dozens of tiles before its eight atomic adds)
    const unsigned hist_blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(512, (tiles + 3) / 4));
    hipLaunchKernelGGL(slab_rows_hist_kernel, dim3(hist_blocks, (unsigned)G), dim3(256), 0, b->ctx->stream, c->d_slabs, tiles, d_hist);
    HIP_TRY(hipGetLastError());
    std::vector<unsigned long long> hist((size_t)8 * G, 0);
    HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, sizeof(unsigned long long) * 8 * G, hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    // table rows: the fewest (in eights) that leave at most 1 % of the (slab, tile) pairs to the direct path -- every row costs
    // 512 bytes of LDS per wavefront, and the sweep is bound by the wavefronts a CU holds
    unsigned long long total = 0, by_rows[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = 0; g < G; ++g)
        for (int k = 0; k < 8; ++k) by_rows[k] += hist[(size_t)8 * g + k], total += hist[(size_t)8 * g + k];
    int k_rows = 7;
    unsigned long long above = 0;
    while (k_rows > 0 && (above + by_rows[k_rows]) * 100 <= total) above += by_rows[k_rows--];
    c->lrows = 8 * (k_rows + 1);
    c->local_direct_share = total ? (double)above / (double)total : 0.0;
    c->local.assign(G, SlabLocal());
    for (int g = 0; g < G; ++g) {
        const SlabCodes &sc = c->slabs[g];
        SlabLocal &L = c->local[g];
        if (sc.nquads == 0) continue;
        const size_t words = (size_t)tiles * sc.nquads * 64, entries = (size_t)tiles * c->lrows * 64;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&L.lcodes), words * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&L.ldict), entries * sizeof(float2)));
        c->local_bytes += (int64_t)(words * sizeof(uint32_t) + entries * sizeof(float2));
        hipLaunchKernelGGL(local_encode_kernel, dim3((unsigned)tiles), dim3(64), 0, b->ctx->stream, sc, c->dict, L.lcodes, L.ldict, c->lrows);
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
