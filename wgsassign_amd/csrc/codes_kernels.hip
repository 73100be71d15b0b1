// The class encoder (common.h: wgs_codes): ONE pass over the float32 slabs that leaves everything both coded sweeps need --
// the per-SNP dictionary and class ids of every individual (scoring), and per population slab its own numbering of the
// classes with its own dictionary rows (EM).  Round 3 did this in two kernels (77 + 14 ms at 10M x 1000) whose walk was a
// chain of dependent LDS probes, one wavefront per tile; here the probes are LDS compare-and-swaps issued eight at a
// time by four wavefronts per SIMD, and nothing is added to a shared counter.
//
//   hash table   per SNP, T slots of 8 bytes (the (g0, g1) bit pattern) in LDS, open addressing, linear probing, insert-only.
//                A lookup-or-insert at slot h is ONE ds_cmpst_rtn_b64 (expect EMPTY, store the key): it returns EMPTY (inserted),
//                the key (found) or another key (probe h + 1).  An insert-only table gives every key the same slot whatever the
//                order of the operations, and LDS operations of a wavefront execute in issue order, so the lookups of a
//                buffer are issued back to back without waiting for each other -- also when two of them insert the same new key.
//   geometry     a wavefront owns 1024 slots (10 KiB of LDS with the marks, <= 128 VGPRs: FOUR wavefronts per SIMD -- the walk is a
//                chain of LDS round trips, and only other wavefronts hide them: 2048 slots at two per SIMD took 45 ms where this
//                takes 33, 4096 at one 77): SNPS SNPs x T slots, (16, 64), (8, 128) or (4, 256), chosen per matrix from a sample
//                (low-depth data with fixed error has ~27 classes per SNP among 1000 individuals, likelihoods from binned base
//                qualities 70-100).  The 64 / SNPS lanes of a SNP share its table (real atomics) and take alternate quads.
//   walk         per slab: the lane's quads, two 16-byte loads each (the slab's native layout, lane <-> SNP), 4 lookups per
//                quad.  A lookup that finds its slot EMPTY has met a NEW class: classes are numbered in the order they appear
//                (round 5; a byte per slot holds the number, a byte per number the slot), so the quad's code word -- four class
//                ids -- is final when it is written, frequent classes have low numbers, and a bit per class in registers
//                remembers what the lane met in this slab.
//   slab end     the slab's OWN class number = the rank of a class among the classes met in the slab (a popcount of the bits
//                below it); the slab's dictionary rows are written coalesced (ldict), its code words re-read (they were
//                written moments ago) and written as local ranks (lcodes), the most local classes of the tile's SNPs goes to
//                tile_rows (what the coded EM sweep sizes its table by).
//   end          dictionary rows (dict) in class order, ncls.  (Round 4 numbered the classes by hash slot after the walk and
//                rewrote every slab's code words: a sixth of the pass's traffic and instructions.)
// A SNP whose table overflows (more than RMAX probes, i.e. too many classes for T), with more classes than the dictionary
// has rows, or holding the one bit pattern used as EMPTY is RICH: ncls = 0 and tile_rows = 255 tell the sweeps to take that
// SNP (scoring) / that tile of that slab (EM) from the float32 slab.  Nothing about a rich SNP affects the others.
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));   // plain vector type (nontemporal loads take it)
typedef const f4 __attribute__((address_space(1))) *gf4_ptr;
// Pointers read out of the slab table are generic to the compiler (-> flat_load / flat_store, which count as LDS operations
// too and stall the probes' counted waits); they are device-global by construction.
typedef uint32_t __attribute__((address_space(1))) *gu32_ptr;
typedef unsigned long long __attribute__((address_space(1))) *gu64_ptr;
typedef uint8_t __attribute__((address_space(1))) *gu8_ptr;

constexpr unsigned long long KEY_EMPTY = ~0ull;            // (g0, g1) = two NaNs with all payload bits set: no parser makes it
constexpr int ENC_SLOTS = WGS_ENC_SLOTS;                            // hash slots per wavefront (16 KiB of keys + 4 KiB of marks: 8 wavefronts per CU)
constexpr int ENC_RMAX = 24;                               // probe rounds per buffer before the SNP is given up as rich
constexpr int ENC_PD = ENC_SLOTS >= 2048 ? 2 : 1;                                  // buffers requested ahead of the one being hashed
constexpr int ENC_UQ = ENC_SLOTS >= 2048 ? 4 : 2;                                 // quads per lane and buffer: 8 loads of 16 bytes, 16 lookups in flight

__device__ __forceinline__ unsigned hash32(unsigned g0, unsigned g1)
{
    // two 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate) over all 32 bits of a mix of both values
    const unsigned x = g0 ^ __builtin_rotateleft32(g1, 13);
    return __umul24(x, 0x9E3779u) ^ __builtin_rotateleft32(__umul24(x >> 8, 0x85EBCBu), 3);
}

struct EncodeArgs {
    float4 *const *base;           // device: slab bases
    const int32_t *npairs, *ncols; // device: per slab
    const SlabCodes *slabs;        // device
    int32_t n_slabs;
    int64_t m, tiles;
    float2 *dict;
    uint8_t *ncls;
    int32_t drows, lrows;          // dictionary rows per tile / rows of a slab's own dictionary per tile (0: no local numbering)
    int32_t batch, batch_cap;      // the coded scoring sweep tables `batch` (16, 8 or 4) consecutive SNPs at a time: most classes such an
                                   // aligned group may sum to (its LDS table)
    int64_t unit_stride;           // sample pass: work unit = blockIdx.x * unit_stride
    uint8_t *sample;               // sample pass: [unit][slab, then all][SNP of the unit] classes found (0: beyond the last SNP, 255: overflow)
    unsigned long long *stats;     // see EncStat
    uint4 *wave_stats;             // full pass: two records per work unit (sum of ncls, rich SNPs, most classes, most rows of a batch | rounds, buffers)
};

// stats[]: what the host reads back after a pass
enum EncStat {
    ST_SUM_NCLS = 0,      // sum of ncls over the coded SNPs
    ST_RICH = 1,          // rich SNPs
    ST_CMAX = 2,          // most classes of a coded SNP
    ST_ROWS_BATCH = 3,    // most classes summed over an aligned group of `batch` SNPs
    ST_UNUSED = 4,
    ST_ROUNDS = 5,        // probe rounds beyond the first, summed over buffers
    ST_BUFFERS = 6,       // buffers walked
    ST_DIRECT = 7,        // (slab, tile) pairs with more classes than the EM sweep's table has rows
    ST_COUNT = 8
};

#ifdef WGS_ENC_STATS
// (experiments: -DWGS_ENC_STATS adds up over a sample of class_encode_kernel's wavefronts the clock cycles [0] before the first slab,
// [1] in the slabs' walks (hash, insert, code words), [2] at the slabs' ends (the slab's own numbering), [3] at the end (dictionary
// rows, records); [4] wavefronts sampled; printed and reset by launch_class_encode.)
__device__ unsigned long long g_enc_stats[8];
#define ENC_CLOCK(i) do { const unsigned long long now_ = clock64(); stat_[i] += now_ - mark_; mark_ = now_; } while (0)
#else
#define ENC_CLOCK(i) do { } while (0)
#endif

template <int SNPS, bool SAMPLE>
__global__ __launch_bounds__(64, ENC_SLOTS >= 2048 ? 2 : 4) void class_encode_kernel(EncodeArgs A)
{
#ifdef WGS_ENC_STATS
    unsigned long long stat_[4] = {0, 0, 0, 0}, mark_ = clock64();
#endif
    constexpr int COLS = 64 / SNPS, T = ENC_SLOTS / SNPS, NW = T / 64;              // NW 64-bit words hold a bit per class of a SNP
    constexpr unsigned TMASK = T - 1, HSHIFT = T == 64 ? 26 : (T == 128 ? 25 : 24);
    constexpr int NL = 4 * ENC_UQ;                         // lookups per lane and buffer
    __shared__ unsigned long long keys[ENC_SLOTS];         // [slot * SNPS + s]
    __shared__ __align__(16) uint8_t gid_of[ENC_SLOTS];    // [slot * SNPS + s]: the class id of the key in that slot
    __shared__ __align__(16) uint8_t slot_of[ENC_SLOTS];   // [class id * SNPS + s]: the slot of that class
    const int lane = threadIdx.x;
    const int s = lane & (SNPS - 1), col = lane / SNPS;
    const int64_t unit = (int64_t)blockIdx.x * A.unit_stride;
    const int64_t tile = unit / (64 / SNPS);
    const int sub = (int)(unit - tile * (64 / SNPS));
    if (tile >= A.tiles) return;
    const int ls = sub * SNPS + s;                         // this lane's SNP within the tile
    const int64_t snp = tile * 64 + ls;

#pragma unroll
    for (int i = 0; i < ENC_SLOTS / 64; ++i) keys[i * 64 + lane] = KEY_EMPTY;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

    bool rich = false;                                     // this lane gave up on its SNP (combined over the SNP's lanes at slab ends)
    unsigned n_rounds = 0, n_buffers = 0;
    int nid = 0;                                           // classes of this lane's SNP so far: the same number in all lanes of the SNP

    // lanes of one SNP: col = 0 .. COLS-1 at lane distance SNPS
    auto snp_or = [&](bool v) {
        int x = v ? 1 : 0;
#pragma unroll
        for (int d = SNPS; d < 64; d <<= 1) x |= __shfl_xor(x, d, 64);
        return x != 0;
    };
    auto snp_or64 = [&](unsigned long long v) {
        unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
        for (int d = SNPS; d < 64; d <<= 1) {
            lo |= (unsigned)__shfl_xor((int)lo, d, 64);
            hi |= (unsigned)__shfl_xor((int)hi, d, 64);
        }
        return ((unsigned long long)hi << 32) | lo;
    };
    // (exclusive prefix over the SNP's lanes in col order, total)
    auto snp_scan = [&](int cnt, int &pre, int &tot) {
        pre = 0;
        tot = 0;
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            const int v = __shfl(cnt, c * SNPS + s, 64);
            if (c < col) pre += v;
            tot += v;
        }
    };
    auto wave_max = [&](int v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
        return v;
    };
    // one probe of lookup `key` at `slot`: what the slot held (EMPTY: the key went in; the key: found; another key: go on)
    auto probe = [&](unsigned slot, unsigned long long key) { return atomicCAS(&keys[slot * SNPS + s], KEY_EMPTY, key); };
    // a bit per class id in NW words
    auto set_bit = [&](unsigned long long (&m)[NW], unsigned id) {
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if (NW == 1 || (id >> 6) == (unsigned)w) m[w] |= 1ull << (id & 63u);
    };
    auto bits_below = [&](const unsigned long long (&m)[NW], unsigned id) {      // set bits at positions < id: the class's rank among the set
        int r = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned long long below = (id >> 6) > (unsigned)w ? ~0ull : ((id >> 6) == (unsigned)w ? (1ull << (id & 63u)) - 1ull : 0ull);
            r += __popcll(m[w] & below);
        }
        return r;
    };
    auto count_bits = [&](const unsigned long long (&m)[NW]) {
        int r = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) r += __popcll(m[w]);
        return r;
    };
    // walks the set bits of m in ascending order: lowest set bit (position), then cleared
    auto pop_lowest = [&](unsigned long long (&m)[NW]) {
        int pos = 0;
        bool done = false;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (!done && m[w]) {
                pos = 64 * w + __builtin_ctzll(m[w]);
                m[w] &= m[w] - 1ull;
                done = true;
            }
        }
        return pos;
    };

    ENC_CLOCK(0);
    for (int g = 0; g < A.n_slabs; ++g) {
        const int np = A.npairs[g], nc = A.ncols[g];
        const SlabCodes sc = A.slabs[g];
        const int nquads = sc.nquads;
        if (nquads == 0) continue;
        gf4_ptr src = (gf4_ptr)A.base[g] + tile * np * 64 + ls;
        gu32_ptr cw = (gu32_ptr)sc.codes + tile * nquads * 64 + ls;
        unsigned long long seen[NW];                       // classes this lane met in this slab
#pragma unroll
        for (int w = 0; w < NW; ++w) seen[w] = 0ull;
        // ---- the walk: this lane's quads col, col + COLS, ...; the loads of the next buffer are in flight while this one is hashed
        // (two buffers ahead: the walk is bound by memory latency -- a wavefront has few loads in flight -- and that latency varies
        // with how the driver could place the matrix; 16 KiB in flight per wavefront while 8 KiB are being hashed)
        f4 nxt[ENC_PD][ENC_UQ][2];
        auto fetch = [&](int qb, f4 (&dst)[ENC_UQ][2]) {
#pragma unroll
            for (int u = 0; u < ENC_UQ; ++u) {
                const int q = qb + u * COLS + col;
                dst[u][0] = __builtin_nontemporal_load(src + (int64_t)min(2 * q, np - 1) * 64);
                dst[u][1] = __builtin_nontemporal_load(src + (int64_t)min(2 * q + 1, np - 1) * 64);
            }
        };
        constexpr int QSTEP = COLS * ENC_UQ;
#pragma unroll
        for (int d = 0; d < ENC_PD; ++d)
            if (d == 0 || d * QSTEP < nquads) fetch(d * QSTEP, nxt[d]);
        for (int qb = 0; qb < nquads; qb += QSTEP) {
            f4 v[ENC_UQ][2];
#pragma unroll
            for (int u = 0; u < ENC_UQ; ++u) v[u][0] = nxt[0][u][0], v[u][1] = nxt[0][u][1];
#pragma unroll
            for (int d = 0; d + 1 < ENC_PD; ++d)
#pragma unroll
                for (int u = 0; u < ENC_UQ; ++u) nxt[d][u][0] = nxt[d + 1][u][0], nxt[d][u][1] = nxt[d + 1][u][1];
            if (qb + ENC_PD * QSTEP < nquads) fetch(qb + ENC_PD * QSTEP, nxt[ENC_PD - 1]);
            unsigned long long key[NL], old[NL];
            unsigned slot[NL];
#pragma unroll
            for (int u = 0; u < ENC_UQ; ++u) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const f4 vv = v[u][h >> 1];
                    const unsigned g0 = __float_as_uint((h & 1) ? vv.z : vv.x), g1 = __float_as_uint((h & 1) ? vv.w : vv.y);
                    key[4 * u + h] = ((unsigned long long)g1 << 32) | g0;          // = the float2 (g0, g1) as the slab holds it
                    slot[4 * u + h] = hash32(g0, g1) >> HSHIFT;
                }
            }
            bool pend[NL], live[NL];                       // lane masks in scalar registers
            // a buffer whose 4 x ENC_UQ x COLS individuals all exist, in a wavefront without a rich SNP: no per-lookup predicates
            const bool plain = 4 * (qb + COLS * ENC_UQ) <= nc && !__any(rich);
            if (plain) {
#pragma unroll
                for (int i = 0; i < NL; ++i) old[i] = probe(slot[i], key[i]);
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    live[i] = true;
                    pend[i] = old[i] != KEY_EMPTY && old[i] != key[i];
                }
            } else {
#pragma unroll
                for (int u = 0; u < ENC_UQ; ++u) {
                    const int q = qb + u * COLS + col;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const int i = 4 * u + h;
                        live[i] = 4 * q + h < nc && !rich;
                        old[i] = key[i];
                        if (live[i]) old[i] = probe(slot[i], key[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < NL; ++i) pend[i] = live[i] && old[i] != KEY_EMPTY && old[i] != key[i];
            }
            bool any_pend = false;
#pragma unroll
            for (int i = 0; i < NL; ++i) any_pend = any_pend || pend[i];
            int rounds = 0;
            while (__any(any_pend)) {
                if (++rounds > ENC_RMAX) {                 // too many classes for this table: the SNP is rich
                    rich = rich || any_pend;
                    break;
                }
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    if (__any(pend[i])) {                  // (most lookups are settled by their first probe: skipped as a wave)
                        if (pend[i]) {
                            slot[i] = (slot[i] + 1) & TMASK;
                            old[i] = probe(slot[i], key[i]);
                        }
                    }
                }
                any_pend = false;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    pend[i] = pend[i] && old[i] != KEY_EMPTY && old[i] != key[i];
                    any_pend = any_pend || pend[i];
                }
            }
            n_rounds += (unsigned)rounds;
            ++n_buffers;
            // A lookup that found its slot EMPTY put a NEW class there: classes are numbered in the order they appear -- per SNP the
            // lanes of the SNP number this buffer's newcomers after the nid classes known so far (column order, then lookup order;
            // nid stays the same number in all lanes of a SNP).  New classes are rare once the first few buffers of the first slab
            // are through: the whole step is skipped as a wave.  Frequent classes come early, i.e. get low numbers -- the numbers
            // are what the sweeps index their tables by, and nothing has to be renumbered at the end (round 4 numbered the classes
            // by hash slot after the walk and rewrote every code word: 20 GB of the pass's 150).
            int ins = 0;
#pragma unroll
            for (int i = 0; i < NL; ++i) ins += (live[i] && !pend[i] && old[i] == KEY_EMPTY) ? 1 : 0;
            if (__any(ins != 0)) {
                int pre, tot;
                snp_scan(ins, pre, tot);
                int id = nid + pre;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    if (live[i] && !pend[i] && old[i] == KEY_EMPTY) {
                        gid_of[slot[i] * SNPS + s] = (uint8_t)min(id, 255);
                        if (id < T) slot_of[id * SNPS + s] = (uint8_t)slot[i];
                        ++id;
                    }
                }
                nid += tot;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            // the lookups' class ids: the code words, and the classes met in this slab (a rich SNP's are never read)
            unsigned cid[NL];
#pragma unroll
            for (int i = 0; i < NL; ++i) cid[i] = gid_of[slot[i] * SNPS + s];
            if (!plain) {                                  // (individuals beyond the slab's last, lookups given up: class 0 in the code word)
#pragma unroll
                for (int i = 0; i < NL; ++i) cid[i] = (live[i] && !pend[i]) ? cid[i] : 0u;
            }
#pragma unroll
            for (int i = 0; i < NL; ++i)
                if (plain || (live[i] && !pend[i])) set_bit(seen, cid[i] < (unsigned)T ? cid[i] : 0u);
            if (!SAMPLE) {
#pragma unroll
                for (int u = 0; u < ENC_UQ; ++u) {
                    const int q = qb + u * COLS + col;
                    if (plain || q < nquads) cw[(int64_t)q * 64] = cid[4 * u] | (cid[4 * u + 1] << 8) | (cid[4 * u + 2] << 16) | (cid[4 * u + 3] << 24);
                }
            }
        }
        // ---- slab end: the slab's own numbering = the rank of a class among the classes met in this slab
        ENC_CLOCK(1);
        if (nid > T || nid > 254) rich = true;
        rich = snp_or(rich);
#pragma unroll
        for (int w = 0; w < NW; ++w) seen[w] = snp_or64(seen[w]);
        const int nloc = count_bits(seen);
        if (SAMPLE) {
            if (col == 0) A.sample[((int64_t)blockIdx.x * (A.n_slabs + 1) + g) * SNPS + s] = (uint8_t)(snp < A.m ? (rich ? 255 : min(nloc, 254)) : 0);
        } else {
            const int wmax = wave_max(rich ? 255 : nloc);
            // one byte per group of WGS_ENC_MIN_SNPS SNPs of the tile (this wave's share of them): plain stores, the EM sweep takes the largest
            if (lane < SNPS / WGS_ENC_MIN_SNPS) ((gu8_ptr)sc.tile_rows)[tile * WGS_TILE_ROWS_BYTES + sub * (SNPS / WGS_ENC_MIN_SNPS) + lane] = (uint8_t)wmax;
            if (A.lrows > 0 && wmax <= A.lrows) {          // wave-uniform: this wave's SNPs fit the EM sweep's table
                // dictionary rows: as many as the richest of this wave's SNPs has (the EM sweep requests rows eight at a time and never
                // looks at a row beyond a SNP's own classes: what it finds in the unwritten ones does not matter); this lane writes
                // ranks col, col + COLS, ... of its SNP
                const int rows_w = wmax;
                gu64_ptr ld = (gu64_ptr)sc.ldict + (tile * A.lrows) * 64 + ls;
                unsigned long long walk[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) walk[w] = seen[w];
                for (int c = 0; c < col; ++c) (void)pop_lowest(walk);
                for (int r0 = col; r0 < rows_w; r0 += COLS) {
                    unsigned long long e = 0;
                    if (r0 < nloc) {
                        const int id = pop_lowest(walk);
                        e = keys[(unsigned)slot_of[id * SNPS + s] * SNPS + s];
                        for (int c = 1; c < COLS; ++c) (void)pop_lowest(walk);
                    }
                    ld[(int64_t)r0 * 64] = e;
                }
                // the code words again, as local ranks
                gu32_ptr lw = (gu32_ptr)sc.lcodes + tile * nquads * 64 + ls;
                constexpr int PF = 8;
                for (int q0 = col; q0 < nquads; q0 += COLS * PF) {
                    uint32_t w[PF];
#pragma unroll
                    for (int u = 0; u < PF; ++u) w[u] = cw[(int64_t)min(q0 + u * COLS, nquads - 1) * 64];
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        if (q0 + u * COLS < nquads) {
                            uint32_t o = 0;
#pragma unroll
                            for (int h = 0; h < 4; ++h) o |= (uint32_t)(bits_below(seen, (w[u] >> (8 * h)) & 255u) & 255) << (8 * h);
                            lw[(int64_t)(q0 + u * COLS) * 64] = o;
                        }
                    }
                }
            }
        }
    }

    // ---- end: the SNP's classes are numbered already; its dictionary rows, in that order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    ENC_CLOCK(2);
    const int ncls = nid;
    if (ncls > A.drows || ncls > 254 || ncls > T) rich = true;
    // the one bit pattern used as EMPTY, looked up: it "went in" without changing its slot -- that SNP cannot be coded
    for (int r0 = col; r0 < min(ncls, T); r0 += COLS)
        if (keys[(unsigned)slot_of[r0 * SNPS + s] * SNPS + s] == KEY_EMPTY) rich = true;
    rich = snp_or(rich);
    if (SAMPLE) {
        if (col == 0) A.sample[((int64_t)blockIdx.x * (A.n_slabs + 1) + A.n_slabs) * SNPS + s] = (uint8_t)(snp < A.m ? (rich ? 255 : min(ncls, 254)) : 0);
        return;
    }
    int eff = rich ? 0 : ncls;
    // the scoring sweep's table holds the classes of `batch` consecutive SNPs: a group that would not fit loses its
    // richest SNPs to the direct path.  (Every group of `batch` lanes holds `batch` consecutive SNPs of one column, and all
    // columns of a SNP see the same numbers and take the same decision.)
    auto group_sum = [&](int v) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1)
            if (off < A.batch && off < SNPS) v += __shfl_xor(v, off, 64);
        return v;
    };
    for (int it = 0; it < 16; ++it) {
        const int sum = group_sum(eff);
        if (!__any(sum > A.batch_cap)) break;
        int best = (eff << 8) | (63 - (lane & 15));        // the richest SNP of the group (lowest lane on ties)
#pragma unroll
        for (int off = 1; off < 16; off <<= 1)
            if (off < A.batch && off < SNPS) best = max(best, __shfl_xor(best, off, 64));
        if (sum > A.batch_cap && (best & 255) == 63 - (lane & 15)) {
            eff = 0;
            rich = true;
        }
    }
    if (col == 0) ((gu8_ptr)A.ncls)[snp] = (uint8_t)eff;
    {
        const int mb = wave_max(group_sum(eff)), mc = wave_max(eff);
        unsigned long long tot = (unsigned long long)((col == 0 && snp < A.m) ? eff : 0), nrich = (col == 0 && snp < A.m && rich) ? 1ull : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            tot += __shfl_xor(tot, off, 64);
            nrich += __shfl_xor(nrich, off, 64);
        }
        // this wavefront's record (no atomics on shared counters: 312 500 wavefronts adding to one cache line queue up behind each
        // other in its memory channel; encode_stats_kernel adds the records up)
        if (lane == 0) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            typedef u4 __attribute__((address_space(1))) *gu4_ptr;
            gu4_ptr rec = (gu4_ptr)A.wave_stats + 2 * (int64_t)blockIdx.x;
            rec[0] = u4{(unsigned)tot, (unsigned)nrich, (unsigned)mc, (unsigned)mb};
            rec[1] = u4{n_rounds, n_buffers, 0u, 0u};
        }
    }
    const int wmax = min(wave_max(eff), A.drows);
    gu64_ptr dd = (gu64_ptr)A.dict + (tile * A.drows) * 64 + ls;
    for (int r0 = col; r0 < wmax; r0 += COLS)
        if (r0 < eff) dd[(int64_t)r0 * 64] = keys[(unsigned)slot_of[r0 * SNPS + s] * SNPS + s];
#ifdef WGS_ENC_STATS
    ENC_CLOCK(3);
    if (lane == 0 && (blockIdx.x & 63u) == 0u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&g_enc_stats[i], stat_[i]);
        atomicAdd(&g_enc_stats[4], 1ull);
    }
#endif
}

// After the sample pass: hist[c] = sampled SNPs with c classes, hist[256 + c] = sampled (slab, SNP) pairs with c classes in the slab
// (slabs with individuals only).  One workgroup; the host reads back 4 KiB instead of the sample (the first device-to-host copy of
// more than a few KiB in a process costs ~7 ms, whatever its destination).
__global__ __launch_bounds__(256) void sample_hist_kernel(const uint8_t *sample, int64_t units, int n_slabs, const int32_t *ncols, unsigned long long *hist)
{
    __shared__ unsigned int hg[256], hl[256];
    hg[threadIdx.x] = hl[threadIdx.x] = 0;
    __syncthreads();
    constexpr int W = WGS_ENC_MIN_SNPS;
    for (int64_t i = threadIdx.x; i < units * W; i += 256) {
        const uint8_t *row = sample + (i / W) * (n_slabs + 1) * W;
        const int x = (int)(i % W);
        const unsigned g = row[n_slabs * W + x];
        if (g == 0) continue;                                // beyond the last SNP
        atomicAdd(&hg[g], 1u);
        for (int sl = 0; sl < n_slabs; ++sl)
            if (ncols[sl]) atomicAdd(&hl[row[sl * W + x]], 1u);
    }
    __syncthreads();
    hist[threadIdx.x] = hg[threadIdx.x];
    hist[256 + threadIdx.x] = hl[threadIdx.x];
}

// After the encode pass: the wavefronts' records added up, and the (slab, tile) pairs whose SNPs do not fit the EM sweep's table counted.
__global__ __launch_bounds__(256) void encode_stats_kernel(const uint4 *wave_stats, int64_t units, const SlabCodes *slabs, int n_slabs, int64_t tiles,
                                                           unsigned lrows, unsigned long long *out)
{
    unsigned long long sum = 0, rich = 0, rounds = 0, buffers = 0, direct = 0;
    unsigned cmax = 0, rows = 0;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += step) {
        const uint4 a = wave_stats[2 * u], b = wave_stats[2 * u + 1];
        sum += a.x, rich += a.y, cmax = max(cmax, a.z), rows = max(rows, a.w), rounds += b.x, buffers += b.y;
    }
    for (int g = 0; g < n_slabs; ++g) {
        if (slabs[g].nquads == 0) continue;
        const unsigned long long *tr = reinterpret_cast<const unsigned long long *>(slabs[g].tile_rows);
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < tiles; t += step) {
            unsigned mx = 0;
#pragma unroll
            for (int x = 0; x < WGS_TILE_ROWS_BYTES / 8; ++x) {
                const unsigned long long w = tr[t * (WGS_TILE_ROWS_BYTES / 8) + x];
#pragma unroll
                for (int k = 0; k < 8; ++k) mx = max(mx, (unsigned)((w >> (8 * k)) & 255u));
            }
            direct += mx > lrows ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64), rich += __shfl_xor(rich, off, 64), rounds += __shfl_xor(rounds, off, 64);
        buffers += __shfl_xor(buffers, off, 64), direct += __shfl_xor(direct, off, 64);
        cmax = max(cmax, (unsigned)__shfl_xor((int)cmax, off, 64)), rows = max(rows, (unsigned)__shfl_xor((int)rows, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + ST_SUM_NCLS, sum);
        atomicAdd(out + ST_RICH, rich);
        atomicMax(out + ST_CMAX, (unsigned long long)cmax);
        atomicMax(out + ST_ROWS_BATCH, (unsigned long long)rows);
        atomicAdd(out + ST_ROUNDS, rounds);
        atomicAdd(out + ST_BUFFERS, buffers);
        atomicAdd(out + ST_DIRECT, direct);
    }
}

}  // namespace

static EncodeArgs encode_args(wgs_beagle *b, wgs_codes *c, const int32_t *d_ncols, unsigned long long *d_stats)
{
    EncodeArgs A;
    A.base = b->d_base;
    A.npairs = b->d_npairs;
    A.ncols = d_ncols;
    A.slabs = c->d_slabs;
    A.n_slabs = b->n_groups;
    A.m = b->m;
    A.tiles = wgs_ntiles(b->m);
    A.dict = c->dict;
    A.ncls = c->ncls;
    A.drows = c->drows;
    A.lrows = c->lrows;
    A.batch = c->score_batch;
    A.batch_cap = WGS_BATCH_ROWS_CAP;
    A.unit_stride = 1;
    A.sample = nullptr;
    A.stats = d_stats;
    A.wave_stats = nullptr;
    return A;
}

// Device scratch of an encode pass: the slabs' column counts and the statistics block.
static int encode_scratch(wgs_beagle *b, int32_t **d_ncols, unsigned long long **d_stats, size_t extra = 0)
{
    std::vector<int32_t> ncols(b->n_groups);
    for (int g = 0; g < b->n_groups; ++g) ncols[g] = b->slabs[g].ncols;
    void *ws = nullptr;
    const size_t stat_bytes = sizeof(unsigned long long) * ST_COUNT;
    if (wgs_ctx_workspace(b->ctx, stat_bytes + sizeof(int32_t) * (b->n_groups + 64) + extra, &ws)) return 1;
    *d_stats = reinterpret_cast<unsigned long long *>(ws);
    *d_ncols = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(ws) + stat_bytes);
    HIP_TRY(hipMemsetAsync(*d_stats, 0, stat_bytes, b->ctx->stream));
    HIP_TRY(hipMemcpyAsync(*d_ncols, ncols.data(), sizeof(int32_t) * b->n_groups, hipMemcpyHostToDevice, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));        // (ncols is a local)
    return 0;
}

// The sample pass: up to `max_units` groups of 8 SNPs spread over the matrix through the (8, 256) table; hist_g[c] = SNPs
// with c classes, hist_l[c] = (slab, SNP) pairs with c classes in the slab (255 = overflow), 256 entries each.
int launch_class_sample(wgs_beagle *b, wgs_codes *c, int max_units, unsigned long long *hist_g, unsigned long long *hist_l, double *rounds_per_buffer)
{
    int32_t *d_ncols = nullptr;
    unsigned long long *d_stats = nullptr;
    const int64_t units = wgs_ntiles(b->m) * WGS_TILE_ROWS_BYTES;
    const int64_t stride = std::max<int64_t>(1, units / std::max(1, max_units));
    const int64_t grid = (units + stride - 1) / stride;
    const size_t sample_bytes = ((size_t)grid * (b->n_groups + 1) * WGS_ENC_MIN_SNPS + 7) / 8 * 8, hist_bytes = 512 * sizeof(unsigned long long);
    if (encode_scratch(b, &d_ncols, &d_stats, sample_bytes + hist_bytes + 8)) return 1;
    EncodeArgs A = encode_args(b, c, d_ncols, d_stats);
    A.drows = 254;
    A.lrows = 0;
    A.unit_stride = stride;
    A.sample = reinterpret_cast<uint8_t *>(d_ncols + b->n_groups + 64);
    unsigned long long *d_hist = reinterpret_cast<unsigned long long *>((reinterpret_cast<uintptr_t>(A.sample + sample_bytes) + 7) & ~(uintptr_t)7);
    hipLaunchKernelGGL((class_encode_kernel<WGS_ENC_MIN_SNPS, true>), dim3((unsigned)grid), dim3(64), 0, b->ctx->stream, A);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sample_hist_kernel, dim3(1), dim3(256), 0, b->ctx->stream, A.sample, grid, (int)b->n_groups, d_ncols, d_hist);
    HIP_TRY(hipGetLastError());
    unsigned long long *h = reinterpret_cast<unsigned long long *>(b->ctx->pinned);
    HIP_TRY(hipMemcpyAsync(h, d_hist, hist_bytes, hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    for (int i = 0; i < 256; ++i) {
        hist_g[i] = h[i];
        hist_l[i] = h[256 + i];
    }
    if (rounds_per_buffer) *rounds_per_buffer = 0.0;
    return 0;
}

// The encode pass over the whole matrix with c->snps_per_wave SNPs per wavefront; fills the arrays of `c` and its statistics.
int launch_class_encode(wgs_beagle *b, wgs_codes *c)
{
    int32_t *d_ncols = nullptr;
    unsigned long long *d_stats = nullptr;
    if (encode_scratch(b, &d_ncols, &d_stats)) return 1;
    const int64_t tiles = wgs_ntiles(b->m);
    EncodeArgs A = encode_args(b, c, d_ncols, d_stats);
    const int64_t units = tiles * (64 / c->snps_per_wave);
    A.wave_stats = c->wave_stats;
    WGS_REQUIRE(units < (1ll << 31), "class encoder: %lld work units exceed one launch", (long long)units);
    HIP_TRY(hipEventRecord(b->ctx->enc_ev0, b->ctx->stream));
    if (c->snps_per_wave == WGS_ENC_SLOTS / 64) hipLaunchKernelGGL((class_encode_kernel<WGS_ENC_SLOTS / 64, false>), dim3((unsigned)units), dim3(64), 0, b->ctx->stream, A);
    else if (c->snps_per_wave == WGS_ENC_SLOTS / 128) hipLaunchKernelGGL((class_encode_kernel<WGS_ENC_SLOTS / 128, false>), dim3((unsigned)units), dim3(64), 0, b->ctx->stream, A);
    else hipLaunchKernelGGL((class_encode_kernel<WGS_ENC_SLOTS / 256, false>), dim3((unsigned)units), dim3(64), 0, b->ctx->stream, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(b->ctx->enc_ev1, b->ctx->stream));
    {
        const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(512, (units + 255) / 256));
        hipLaunchKernelGGL(encode_stats_kernel, dim3(gx), dim3(256), 0, b->ctx->stream, c->wave_stats, units, c->d_slabs, (int)b->n_groups, tiles,
                           (unsigned)(c->lrows > 0 ? c->lrows : 255), d_stats);
        HIP_TRY(hipGetLastError());
    }
    unsigned long long *h = reinterpret_cast<unsigned long long *>(b->ctx->pinned);
    HIP_TRY(hipMemcpyAsync(h, d_stats, sizeof(unsigned long long) * 8, hipMemcpyDeviceToHost, b->ctx->stream));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
#ifdef WGS_ENC_STATS
    {
        unsigned long long st[8] = {0}, zero[8] = {0};
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_enc_stats), sizeof st);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_enc_stats), zero, sizeof zero);
        const double w = st[4] ? (double)st[4] : 1.0;
        fprintf(stderr, "[encode stats] %llu wavefronts sampled; cycles per wavefront: before the slabs %.0f, the slabs' walks %.0f, the slabs' ends %.0f, the end %.0f\n",
                st[4], st[0] / w, st[1] / w, st[2] / w, st[3] / w);
    }
#endif
    float ev_ms = 0.0f;
    if (hipEventElapsedTime(&ev_ms, b->ctx->enc_ev0, b->ctx->enc_ev1) == hipSuccess) c->kernel_ms = ev_ms;     // the encode kernel alone (HIP events)
    c->sum_ncls = (double)h[ST_SUM_NCLS];
    c->rich_snps = (int64_t)h[ST_RICH];
    c->cmax = (int32_t)h[ST_CMAX];
    c->rows_batch = std::max<int32_t>(c->score_batch, (int32_t)h[ST_ROWS_BATCH]);
    c->probe_rounds = h[ST_BUFFERS] ? (double)h[ST_ROUNDS] / (double)h[ST_BUFFERS] : 0.0;
    int64_t slab_tiles = 0;
    for (int g = 0; g < b->n_groups; ++g) slab_tiles += c->slabs[g].nquads ? tiles : 0;
    c->local_direct_share = c->lrows > 0 && slab_tiles ? (double)h[ST_DIRECT] / (double)slab_tiles : 1.0;
    return 0;
}
