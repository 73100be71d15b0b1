// Device-side ingest: Beagle TEXT -> population slabs, tokenised on the MI355X.
//
// The reference parses every value on the host (reader_cy.pyx:52-66: strtok + atof per token); so did this
// library's reader until round 3 (reader.cpp: wgs_reader_next), which left the GPU idle for > 99 % of an
// end-to-end run at scale.  Here the host only inflates, finds the newlines and keeps the site names
// (reader.cpp: text_producer); the text itself goes to the device through page-locked buffers and
// `tokenise_kernel` does what strtok/atof did, writing each kept value at its place in the tile-interleaved slab
// (no row-major intermediate, no scatter pass):
//   * one wavefront per line; per step the 64 lanes take 64 consecutive 16-byte words (1 KiB, coalesced);
//   * a byte is a delimiter if it is one of "\t \n\r" (reader.cpp: is_delim); a token starts at a non-delimiter
//     that follows a delimiter; a wave-wide prefix sum of the per-lane start counts numbers the tokens of the
//     line, so every lane knows which (individual, GL) each of its tokens is: token 0 is the site name, 1-2 the
//     alleles (reader_cy.pyx:56-60), then of every triple the first two are kept and the third dropped (:62-66);
//   * a kept token is converted by the lane that holds its first byte, from registers (its word and the next):
//     ANGSD's "d.dddddd" by the eight-digit SWAR trick, any other plain decimal (optional sign, <= 15 significant
//     digits, optional exponent, net power of ten within +-22) by integer mantissa and ONE correctly rounded
//     double multiplication or division by an exact power of ten -- the correctly rounded value of the token,
//     i.e. what atof returns -- then (float), as reader_cy.pyx:66 stores it;
//   * anything else (inf/nan, hex, 16+ digits, tokens longer than 16 bytes, trailing junk), and a line with too
//     few columns, only FLAGS the line: the host re-parses flagged lines with the strtod-backed parser
//     (reader_text_parse_line) and uploads those rows -- none for ANGSD output.
// Bound: the host's inflate rate; the kernel reads 27 bytes of text and writes 8 bytes per (SNP, individual).
#include <string.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "reader_text.h"

size_t inflate_table_bytes(void);
constexpr size_t INFLATE_PAD = 128;   // bytes the compressed chunk is padded by on the device (inflate.hip reads a 24-byte window ahead)
int launch_inflate(wgs_ctx *ctx, const uint8_t *d_comp, const uint64_t *d_in_off, const uint32_t *d_in_len, const uint64_t *d_out_off,
                   const uint32_t *d_isize, uint8_t *d_out, uint8_t *d_status, void *d_tables, int32_t nblocks);

namespace {

__constant__ double kPow10Dev[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                     1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

struct TokArgs {
    const uint4 *text;             // the chunk, 16-byte aligned, padded with newlines
    const uint32_t *begin, *end;   // per line: [begin, end) in bytes
    const int32_t *dst;            // per line: row relative to row0, or -1 (site filtered out)
    uint8_t *flags;                // per line: 1 = the host must parse this line
    uint32_t *nflagged;            // optional: += lines flagged
    int64_t row0;                  // slab row of dst == 0
    int32_t nlines, n_inds;
    const int32_t *group_of, *col_of, *npairs;
    float4 *const *base;
};

// bit i = byte i of v is NOT one of '\t' '\n' '\r' ' '
__device__ __forceinline__ uint32_t nondelim4(uint32_t v)
{
    auto zero_bytes = [](uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); };   // 0x80 per zero byte
    const uint32_t d = zero_bytes(v ^ 0x09090909u) | zero_bytes(v ^ 0x0A0A0A0Au) | zero_bytes(v ^ 0x0D0D0D0Du) | zero_bytes(v ^ 0x20202020u);
    return ((((d >> 7) * 0x01020408u) >> 24) & 0xFu) ^ 0xFu;
}
__device__ __forceinline__ uint32_t nondelim16(const uint4 &x)
{
    return nondelim4(x.x) | (nondelim4(x.y) << 4) | (nondelim4(x.z) << 8) | (nondelim4(x.w) << 12);
}

// The token of `len` (1..16) bytes held in lo (bytes 0-7) and hi (8-15), first character in the lowest byte.
// Returns false when the host has to convert it (see the file comment).
__device__ __forceinline__ bool parse_token(uint64_t lo, uint64_t hi, int len, float *out)
{
    if (len == 8) {                                            // "d.dddddd" (reader.cpp: parse_f6)
        uint64_t d = lo ^ 0x3030303030302E30ull;
        if (!((((d + 0x7676767676767676ull) | d) & 0x8080808080808080ull) || (d & 0xFF00ull))) {
            d = (d >> 8) | (d & 0xFF);
            d = d * 10 + (d >> 8);
            const uint64_t mask = 0x000000FF000000FFull;
            d = (((d & mask) * (100 + (1000000ull << 32))) + (((d >> 16) & mask) * (1 + (10000ull << 32)))) >> 32;
            *out = (float)((double)(uint32_t)d / 1e7);
            return true;
        }
    }
    auto at = [&](int i) { return (uint32_t)((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8))) & 0xFF); };
    int i = 0;
    bool neg = false;
    uint32_t c = at(0);
    if (c == '-' || c == '+') {
        neg = c == '-';
        i = 1;
    }
    uint64_t mant = 0;
    int digits = 0, frac = 0, seen = 0;
    bool dot = false;
    for (; i < len; ++i) {
        c = at(i);
        if (c >= '0' && c <= '9') {
            ++seen;
            if (mant == 0 && c == '0') {
                if (dot) ++frac;
                continue;
            }
            if (++digits > 15) return false;
            mant = mant * 10 + (uint64_t)(c - '0');
            if (dot) ++frac;
        } else if (c == '.' && !dot) {
            dot = true;
        } else {
            break;
        }
    }
    if (seen == 0) return false;
    int ex = 0;
    if (i < len) {
        if (c != 'e' && c != 'E') return false;
        ++i;
        bool xneg = false;
        if (i < len && (at(i) == '-' || at(i) == '+')) xneg = at(i++) == '-';
        int xd = 0;
        for (; i < len && at(i) >= '0' && at(i) <= '9' && xd < 4; ++i, ++xd) ex = ex * 10 + (int)(at(i) - '0');
        if (xd == 0 || xd > 3 || i < len) return false;
        if (xneg) ex = -ex;
    }
    const int net = ex - frac;
    if (net < -22 || net > 22) return false;
    const double v = net < 0 ? (double)mant / kPow10Dev[-net] : (double)mant * kPow10Dev[net];
    *out = (float)(neg ? -v : v);
    return true;
}

__global__ __launch_bounds__(256) void tokenise_kernel(TokArgs a)
{
    const int lane = threadIdx.x & 63;
    const int line = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (line >= a.nlines) return;                              // whole waves leave together
    const int32_t rel = a.dst[line];
    if (rel < 0) return;
    const uint32_t b = a.begin[line], e = a.end[line];
    const int64_t row = a.row0 + rel;
    const uint32_t need = 3u + 3u * (uint32_t)a.n_inds;
    const uint32_t w0 = b >> 4, w1 = (e + 15u) >> 4;           // the 16-byte words that hold the line
    uint32_t tok = 0;                                          // tokens that start before this step's words
    uint32_t prev_nd = 0;                                      // the byte before this step's words is a non-delimiter
    bool bad = false;
    for (uint32_t wb = w0; wb < w1 && tok < need; wb += 64) {
        const uint32_t w = wb + (uint32_t)lane;
        uint4 x = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au), y = x;
        if (w < w1) {
            x = a.text[w];
            y = a.text[w + 1];                                 // in bounds: the buffer is padded by TEXT_PAD bytes
        }
        uint32_t nd = nondelim16(x) | (nondelim16(y) << 16);
        // bytes outside [b, e) count as delimiters: bit i is byte 16 w + i
        const int64_t first = (int64_t)b - (int64_t)w * 16, last = (int64_t)e - (int64_t)w * 16;
        if (first > 0) nd &= first >= 32 ? 0u : ~0u << first;
        if (last < 32) nd &= last <= 0 ? 0u : ~0u >> (32 - last);
        const uint32_t own = nd & 0xFFFFu;
        uint32_t before = (uint32_t)__shfl_up((int)(own >> 15), 1);
        if (lane == 0) before = prev_nd;
        uint32_t starts = own & ~((own << 1) | before);
        const int cnt = __popc(starts);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        uint32_t k = tok + (uint32_t)(incl - cnt);
        while (starts) {
            const int bit = __ffs((int)starts) - 1;
            starts &= starts - 1;
            if (k >= 3 && k < need) {
                const uint32_t j = k - 3, ind = j / 3, which = j - ind * 3;
                if (which < 2) {
                    const uint32_t inv = ~(nd >> bit);
                    const int len = inv ? __ffs((int)inv) - 1 : 32;        // non-delimiters from `bit` on
                    float v = 0.0f;
                    bool ok = bit + len < 32 && len <= 16;                 // the token ends inside the two words
                    if (ok) {
                        const uint64_t q0 = x.x | ((uint64_t)x.y << 32), q1 = x.z | ((uint64_t)x.w << 32);
                        const uint64_t q2 = y.x | ((uint64_t)y.y << 32), q3 = y.z | ((uint64_t)y.w << 32);
                        const uint64_t A = bit & 8 ? q1 : q0, B = bit & 8 ? q2 : q1, C = bit & 8 ? q3 : q2;
                        const int s = (bit & 7) * 8;
                        const uint64_t lo = s ? (A >> s) | (B << (64 - s)) : A, hi = s ? (B >> s) | (C << (64 - s)) : B;
                        ok = parse_token(lo, hi, len, &v);
                    }
                    if (ok) {
                        const int g = a.group_of[ind], col = a.col_of[ind];
                        const int64_t at = ((((row >> 6) * a.npairs[g] + (col >> 1)) << 6) + (row & 63)) * 4 + (col & 1) * 2 + (int)which;
                        reinterpret_cast<float *>(a.base[g])[at] = v;
                    } else {
                        bad = true;
                    }
                }
            }
            ++k;
        }
        tok += (uint32_t)__shfl(incl, 63);
        prev_nd = (uint32_t)__shfl((int)(own >> 15), 63);
    }
    if (tok < need) bad = true;                                // too few columns: the host reports the line
    const bool any_bad = __any(bad);
    if (lane == 0) {
        a.flags[line] = any_bad ? 1 : 0;
        if (any_bad && a.nflagged) atomicAdd(a.nflagged, 1u);
    }
}

// ---- line listing on the device (BGZF members inflated there: the text never exists on the host) -----------------------
// What reader.cpp: list_lines does with memchr on the host: the positions of the newlines (count per 4 KiB block, scan,
// write), then per line its extent, whether it is blank, and its first token (the site name); exclusive scans number the
// non-blank lines and place the names in one blob.  T_* index the totals the host reads back.
enum { T_NEWLINES = 0, T_NONBLANK = 1, T_NAME_BYTES = 2, T_TAIL = 3, T_NAME_CUT = 4, T_FLAGGED = 5, T_COUNT = 8 };

__device__ __forceinline__ uint32_t newline_mask16(const uint4 &x)
{
    auto m4 = [](uint32_t v) {
        const uint32_t y = v ^ 0x0A0A0A0Au;
        const uint32_t z = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);       // 0x80 per newline
        return (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
    };
    return m4(x.x) | (m4(x.y) << 4) | (m4(x.z) << 8) | (m4(x.w) << 12);
}

// this thread's 16 bytes of text[0 .. total): bit i = byte i is a newline
__device__ __forceinline__ uint32_t newline_bits(const uint4 *text, uint64_t total, uint64_t word)
{
    if (word * 16 >= total) return 0;
    uint32_t m = newline_mask16(text[word]);
    const uint64_t left = total - word * 16;
    if (left < 16) m &= (1u << left) - 1u;
    return m;
}

__global__ __launch_bounds__(256) void count_newlines_kernel(const uint4 *text, uint64_t total, uint32_t *counts)
{
    __shared__ uint32_t part[4];
    const uint64_t word = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t c = (uint32_t)__popc(newline_bits(text, total, word));
#pragma unroll
    for (int off = 32; off; off >>= 1) c += (uint32_t)__shfl_down((int)c, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void write_newlines_kernel(const uint4 *text, uint64_t total, const uint32_t *offsets, uint32_t *nl_pos)
{
    __shared__ uint32_t part[4];
    const uint64_t word = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t m = newline_bits(text, total, word);
    const uint32_t c = (uint32_t)__popc(m);
    uint32_t incl = c;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) part[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t at = offsets[blockIdx.x] + incl - c;
    for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) at += part[w];
    while (m) {
        const int bit = __ffs((int)m) - 1;
        m &= m - 1;
        nl_pos[at++] = (uint32_t)(word * 16 + (uint64_t)bit);
    }
}

// out[i] = in[0] + ... + in[i-1]; *total = the whole sum (one workgroup: the arrays are small next to the text)
__global__ __launch_bounds__(1024) void scan_u32_kernel(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total)
{
    __shared__ uint32_t part[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 4096) {
        const uint32_t i0 = base + threadIdx.x * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0u;
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) part[wave] = incl;
        __syncthreads();
        uint32_t at = carry_s + incl - mine;
        for (int w = 0; w < wave; ++w) at += part[w];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) out[i0 + k] = at;
            at += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = at;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

__device__ __forceinline__ bool is_delim_dev(uint8_t c) { return c == '\t' || c == ' ' || c == '\n' || c == '\r'; }

struct LineArgs {
    const uint8_t *text;
    const uint32_t *nl_pos;
    uint32_t nlines;               // newline-terminated lines of the chunk (blank ones included)
    uint32_t *begin, *end;         // per line: [begin, end) without the newline
    uint32_t *nonblank;            // 1 = a data row
    uint32_t *name_start, *name_len1;   // first token; its length + 1 (0 for blank lines)
    uint32_t *totals;
};

__global__ __launch_bounds__(256) void line_info_kernel(LineArgs a)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.nlines) return;
    const uint32_t b = i ? a.nl_pos[i - 1] + 1u : 0u, e = a.nl_pos[i];
    uint32_t q = b;
    while (q < e && is_delim_dev(a.text[q])) ++q;
    uint32_t x = q;
    while (x < e && !is_delim_dev(a.text[x])) ++x;
    a.begin[i] = b;
    a.end[i] = e;
    a.nonblank[i] = q < e ? 1u : 0u;
    a.name_start[i] = q;
    a.name_len1[i] = q < e ? x - q + 1u : 0u;
    if (i == a.nlines - 1) a.totals[T_TAIL] = e + 1u;          // where the partial last line starts
}

struct DstArgs {
    uint32_t nlines, take;         // rows of this chunk: the first `take` non-blank lines
    const uint32_t *nonblank, *rank, *name_start, *name_len1, *name_off;
    const int32_t *dstmap;         // optional, per rank: the row (relative to row0) or -1 = the site is filtered out
    const uint8_t *text;
    int32_t *dst;
    uint8_t *names;
    uint32_t *totals;
};

__global__ __launch_bounds__(256) void dst_names_kernel(DstArgs a)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.nlines) return;
    const bool row = a.nonblank[i] != 0;
    const uint32_t r = a.rank[i];
    if (row && r == a.take) a.totals[T_NAME_CUT] = a.name_off[i];   // the row limit cuts the chunk here
    if (!row || r >= a.take) {
        a.dst[i] = -1;
        return;
    }
    a.dst[i] = a.dstmap ? a.dstmap[r] : (int32_t)r;
    const uint32_t n = a.name_len1[i] - 1u;
    const uint8_t *src = a.text + a.name_start[i];
    uint8_t *out = a.names + a.name_off[i];
    for (uint32_t k = 0; k < n; ++k) out[k] = src[k];
    out[n] = '\n';
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Page-locked staging.  hipHostMalloc takes 0.18 s per GB on this platform and hipHostFree 0.1 s (tools/ubench_pin.hip: 48 and
// 28 ms for the 272 MB of bench.py's ingest leg -- half of that leg's time); anonymous memory in transparent huge pages, faulted
// in by a few threads and then registered, takes 4 ms for the same 272 MB and copies to the device at the same 57 GB/s.  Without
// huge pages (the kernel's setting) it is 4 KiB pages as before, still three times cheaper; what mmap or the registration
// refuses falls back to hipHostMalloc.  The pages go back to the system on a thread of their own (munmap: 12-20 ms).
struct PinnedBlock {
    void *raw = nullptr;          // nullptr: from hipHostMalloc
    size_t raw_bytes = 0;
};
std::mutex g_pinned_mu;
std::unordered_map<void *, PinnedBlock> g_pinned;

void *pinned_alloc(size_t bytes, void *user)                  // called from the producer thread too
{
    if (hipSetDevice((int)(intptr_t)user) != hipSuccess) return nullptr;
    const size_t huge = (size_t)2 << 20;
    if (bytes >= 4 * huge) {
        const size_t len = (bytes + huge - 1) & ~(huge - 1);
        void *raw = mmap(nullptr, len + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (raw != MAP_FAILED) {
            char *p = reinterpret_cast<char *>(((uintptr_t)raw + huge - 1) & ~(uintptr_t)(huge - 1));
            (void)madvise(p, len, MADV_HUGEPAGE);
            const int T = (int)std::max<size_t>(1, std::min<size_t>(8, std::min<size_t>(std::thread::hardware_concurrency(), len / (16 * huge))));
            std::vector<std::thread> th;
            for (int t = 1; t < T; ++t)
                th.emplace_back([=] {
                    for (size_t i = len * (size_t)t / (size_t)T; i < len * (size_t)(t + 1) / (size_t)T; i += 4096) p[i] = 0;
                });
            for (size_t i = 0; i < len / (size_t)T; i += 4096) p[i] = 0;
            for (auto &x : th) x.join();
            if (hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> lk(g_pinned_mu);
                g_pinned[p] = PinnedBlock{raw, len + huge};
                return p;
            }
            (void)hipGetLastError();
            munmap(raw, len + huge);
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    g_pinned[p] = PinnedBlock{};
    return p;
}
void pinned_release(void *p, void *)
{
    if (!p) return;
    PinnedBlock blk;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        auto it = g_pinned.find(p);
        if (it == g_pinned.end()) return;
        blk = it->second;
        g_pinned.erase(it);
    }
    if (!blk.raw) {
        (void)hipHostFree(p);
        return;
    }
    (void)hipHostUnregister(p);
    std::thread([blk] { munmap(blk.raw, blk.raw_bytes); }).detach();
}

}  // namespace

struct wgs_ingest {
    wgs_beagle *b = nullptr;
    wgs_reader *r = nullptr;
    void *d_text = nullptr;
    size_t text_cap = 0;
    uint32_t *d_begin = nullptr, *d_end = nullptr;
    int32_t *d_dst = nullptr;
    uint8_t *d_flags = nullptr;
    size_t lines_cap = 0;
    std::vector<int32_t> dst;
    std::vector<uint8_t> flags;
    std::vector<float> rows;        // host-parsed rows of flagged lines
    std::string names;              // site names of the last chunk
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // statistics
    double wait_s = 0.0, inflate_s = 0.0, scan_s = 0.0, device_ms = 0.0;
    int64_t host_lines = 0, text_bytes = 0, lines = 0, chunks = 0;
    // ---- BGZF: the device-resident pipeline (compressed members in, slab rows out; see wgs_ingest_next) ----
    bool resident = false, done = false;
    int64_t limit = -1, rows_done = 0;
    size_t chunk_text = 0;          // text per chunk the producer aims at
    size_t carry = 0;               // d_text[0 .. carry) = the partial last line of the previous chunk
    uint8_t *d_carry = nullptr;
    size_t carry_cap = 0;
    uint8_t *d_comp = nullptr;
    size_t comp_cap = 0;
    uint64_t *d_in_off = nullptr, *d_out_off = nullptr;
    uint32_t *d_in_len = nullptr, *d_isize = nullptr;
    uint8_t *d_status = nullptr;
    void *d_tables = nullptr;
    size_t blocks_cap = 0;
    uint32_t *d_counts = nullptr, *d_offsets = nullptr;     // newlines per 4 KiB of text, and their exclusive scan
    size_t counts_cap = 0;
    uint32_t *d_nl_pos = nullptr, *d_nonblank = nullptr, *d_rank = nullptr, *d_name_start = nullptr, *d_name_len1 = nullptr, *d_name_off = nullptr;
    int32_t *d_dstmap = nullptr;
    uint8_t *d_names = nullptr;
    size_t names_cap = 0;
    uint32_t *d_totals = nullptr, *h_totals = nullptr;
    std::vector<uint64_t> out_off;
    std::vector<uint8_t> status;
    std::vector<int32_t> dstmap;
    std::vector<uint32_t> h_begin, h_end, h_rank;
    std::vector<char> line;
    hipEvent_t iev0 = nullptr, iev1 = nullptr;
    double inflate_kernel_ms = 0.0, read_s = 0.0, create_s = 0.0, next_s = 0.0;
    int64_t blocks_inflated = 0, blocks_host = 0;
};

namespace {

template <class T>
int regrow(hipStream_t st, T *&p, size_t count)
{
    HIP_TRY(hipStreamSynchronize(st));
    if (p) HIP_TRY(hipFree(p));
    p = nullptr;
    HIP_TRY(wgs_malloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
    return 0;
}

int ensure_line_arrays(wgs_ingest *g, hipStream_t st, size_t nl, bool resident)
{
    if (nl <= g->lines_cap) return 0;
    const size_t cap = nl + nl / 2 + 1024;
    g->lines_cap = 0;
    if (int rc = regrow(st, g->d_begin, cap)) return rc;
    if (int rc = regrow(st, g->d_end, cap)) return rc;
    if (int rc = regrow(st, g->d_dst, cap)) return rc;
    if (int rc = regrow(st, g->d_flags, cap)) return rc;
    if (resident) {
        if (int rc = regrow(st, g->d_nl_pos, cap)) return rc;
        if (int rc = regrow(st, g->d_nonblank, cap)) return rc;
        if (int rc = regrow(st, g->d_rank, cap)) return rc;
        if (int rc = regrow(st, g->d_name_start, cap)) return rc;
        if (int rc = regrow(st, g->d_name_len1, cap)) return rc;
        if (int rc = regrow(st, g->d_name_off, cap)) return rc;
        if (int rc = regrow(st, g->d_dstmap, cap)) return rc;
    }
    g->lines_cap = cap;
    return 0;
}

void launch_tokenise(wgs_ingest *g, hipStream_t st, const void *text, int64_t row0, size_t nl, uint32_t *nflagged)
{
    wgs_beagle *b = g->b;
    TokArgs a;
    a.text = reinterpret_cast<const uint4 *>(text);
    a.begin = g->d_begin;
    a.end = g->d_end;
    a.dst = g->d_dst;
    a.flags = g->d_flags;
    a.nflagged = nflagged;
    a.row0 = row0;
    a.nlines = (int32_t)nl;
    a.n_inds = (int32_t)b->n;
    a.group_of = b->d_group_of;
    a.col_of = b->d_col_of;
    a.npairs = b->d_npairs;
    a.base = b->d_base;
    hipLaunchKernelGGL(tokenise_kernel, dim3((unsigned)((nl + 3) / 4)), dim3(256), 0, st, a);
}

// Lines the device flagged (bit patterns it leaves to strtod), host-parsed and uploaded in runs of consecutive rows.
// text_of(t) returns the line t as [b, e) in host memory; number_of(t) its index among the chunk's data lines.
template <class TextOf, class NumberOf>
int host_parse_flagged(wgs_ingest *g, int64_t row0, size_t nl, const uint8_t *flags, const int32_t *dst, int64_t first_row, TextOf text_of, NumberOf number_of)
{
    wgs_beagle *b = g->b;
    const size_t row_floats = (size_t)2 * (size_t)b->n;
    for (size_t i = 0; i < nl;) {
        if (!flags[i] || dst[i] < 0) {
            ++i;
            continue;
        }
        size_t j = i;
        while (j < nl && flags[j] && dst[j] == dst[i] + (int32_t)(j - i) && j - i < 4096) ++j;
        g->rows.resize((j - i) * row_floats);
        for (size_t t = i; t < j; ++t) {
            const char *lb = nullptr, *le = nullptr;
            if (int rc = text_of(t, &lb, &le)) return rc;
            if (reader_text_parse_line(g->r, lb, le, g->rows.data() + (t - i) * row_floats)) {
                wgs_set_error("Beagle data line %lld has fewer than %d genotype-likelihood columns",
                              (long long)(reader_text_lines_read(g->r) + first_row + (int64_t)number_of(t) + 2), reader_text_gl_cols(g->r));
                return 2;
            }
        }
        if (int rc = wgs_beagle_upload_rows(b, g->rows.data(), row0 + dst[i], (int64_t)(j - i))) return rc;
        g->host_lines += (int64_t)(j - i);
        i = j;
    }
    return 0;
}

/* BGZF, device-resident: per chunk the producer thread only READS the next members (reader.cpp: comp_producer); here
 *   H2D of the compressed bytes -> inflate_kernel (inflate.hip; one lane per member) into d_text behind the carried partial
 *   line -> newline positions (count per 4 KiB, scan, write) -> per line extent / blank / site name -> scans number the data
 *   lines and place the names -> dst (row limit, site mask) + names blob -> tokenise_kernel -> the partial last line moves
 *   to the front for the next chunk.
 * The host sees three small read-backs per chunk (counts, the names, the flag count) and the text never exists there. */
int ingest_next_resident(wgs_ingest *g, int64_t row0, const uint8_t *keep, int64_t keep_len, int64_t *file_rows, int64_t *rows_written)
{
    wgs_beagle *b = g->b;
    hipStream_t st = b->ctx->stream;
    static const bool trace = getenv("WGSASSIGN_INGEST_TRACE") != nullptr;       // per-chunk wall times on stderr
    while (!g->done) {
        const double t_begin = now_s();
        CompChunk *c = nullptr;
        double waited = 0.0;
        if (int rc = reader_comp_next(g->r, &c, &waited)) return rc;
        g->wait_s += waited;
        if (!c) {
            g->done = true;
            break;
        }
        struct Release {
            wgs_ingest *g;
            CompChunk *c;
            ~Release() { if (c) reader_comp_release(g->r, c); }
        } release{g, c};
        const bool last = c->last;
        const int nb = (int)c->isize.size();
        const size_t lead = g->carry + c->pre_len;
        const size_t total = lead + c->text_bytes + (last ? 1 : 0);     // a newline closes an unterminated last line
        WGS_REQUIRE(total + TEXT_PAD + 64 < (1ull << 32), "a Beagle line longer than 4 GiB");
        g->read_s += c->read_s;
        const size_t need = ((total + TEXT_PAD + 15) & ~(size_t)15) + 64;
        if (need > g->text_cap) {                                         // keeps the carried partial line
            HIP_TRY(hipStreamSynchronize(st));
            void *q = nullptr;
            const size_t cap = need + need / 8;
            HIP_TRY(wgs_malloc(&q, cap));
            if (g->carry) HIP_TRY(hipMemcpy(q, g->d_text, g->carry, hipMemcpyDeviceToDevice));
            if (g->d_text) HIP_TRY(hipFree(g->d_text));
            g->d_text = q;
            g->text_cap = cap;
        }
        uint8_t *text = reinterpret_cast<uint8_t *>(g->d_text);
        if ((size_t)nb > g->blocks_cap) {
            const size_t cap = (size_t)nb + (size_t)nb / 4 + 256;
            g->blocks_cap = 0;
            if (int rc = regrow(st, g->d_in_off, cap)) return rc;
            if (int rc = regrow(st, g->d_out_off, cap)) return rc;
            if (int rc = regrow(st, g->d_in_len, cap)) return rc;
            if (int rc = regrow(st, g->d_isize, cap)) return rc;
            if (int rc = regrow(st, g->d_status, cap)) return rc;
            HIP_TRY(hipStreamSynchronize(st));
            if (g->d_tables) HIP_TRY(hipFree(g->d_tables));
            g->d_tables = nullptr;
            HIP_TRY(wgs_malloc(&g->d_tables, inflate_table_bytes() * cap));
            g->blocks_cap = cap;
        }
        if (c->len + INFLATE_PAD > g->comp_cap) {
            g->comp_cap = 0;
            if (int rc = regrow(st, g->d_comp, c->len + INFLATE_PAD)) return rc;
            g->comp_cap = c->len + INFLATE_PAD;
        }
        const double t_alloc = now_s();
        HIP_TRY(hipEventRecord(g->ev0, st));
        HIP_TRY(hipMemsetAsync(g->d_totals, 0, T_COUNT * sizeof(uint32_t), st));
        if (c->pre_len) HIP_TRY(hipMemcpyAsync(text + g->carry, c->pre_text, c->pre_len, hipMemcpyHostToDevice, st));
        if (nb) {
            g->out_off.resize((size_t)nb);
            uint64_t at = lead;
            for (int i = 0; i < nb; ++i) {
                g->out_off[(size_t)i] = at;
                at += c->isize[(size_t)i];
            }
            HIP_TRY(hipMemcpyAsync(g->d_comp, c->comp, c->len, hipMemcpyHostToDevice, st));
            // zeros behind the chunk: a damaged last member that reads on finds an invalid stored-block header there, not the
            // stale bytes of the chunk before (the kernel stops a lane a few bytes past its stream's end anyway)
            HIP_TRY(hipMemsetAsync(g->d_comp + c->len, 0, INFLATE_PAD, st));
            HIP_TRY(hipMemcpyAsync(g->d_in_off, c->in_off.data(), sizeof(uint64_t) * nb, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(g->d_out_off, g->out_off.data(), sizeof(uint64_t) * nb, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(g->d_in_len, c->in_len.data(), sizeof(uint32_t) * nb, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(g->d_isize, c->isize.data(), sizeof(uint32_t) * nb, hipMemcpyHostToDevice, st));
            HIP_TRY(hipEventRecord(g->iev0, st));
            if (launch_inflate(b->ctx, g->d_comp, g->d_in_off, g->d_in_len, g->d_out_off, g->d_isize, text, g->d_status, g->d_tables, nb)) return 1;
            HIP_TRY(hipEventRecord(g->iev1, st));
            g->status.resize((size_t)nb);
            HIP_TRY(hipMemcpyAsync(g->status.data(), g->d_status, (size_t)nb, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipMemsetAsync(text + lead + c->text_bytes, '\n', need - (lead + c->text_bytes), st));
        const size_t nblk = (total + 4095) / 4096;
        if (nblk > g->counts_cap) {
            const size_t cap = nblk + nblk / 4 + 64;
            g->counts_cap = 0;
            if (int rc = regrow(st, g->d_counts, cap)) return rc;
            if (int rc = regrow(st, g->d_offsets, cap)) return rc;
            g->counts_cap = cap;
        }
        for (int pass = 0; pass < 2; ++pass) {
            if (nblk) {
                hipLaunchKernelGGL(count_newlines_kernel, dim3((unsigned)nblk), dim3(256), 0, st, reinterpret_cast<const uint4 *>(text), (uint64_t)total, g->d_counts);
                hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, st, g->d_counts, g->d_offsets, (uint32_t)nblk, g->d_totals + T_NEWLINES);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipMemcpyAsync(g->h_totals, g->d_totals, T_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (pass) break;
            // members the device did not accept: the host's inflater, patched into the device text; the newlines are counted again
            // (WGSASSIGN_DEBUG_REJECT_MEMBERS=k, tests: every k-th member is treated as rejected and its device text wiped first)
            const char *reject_env = getenv("WGSASSIGN_DEBUG_REJECT_MEMBERS");
            const int reject_every = reject_env ? atoi(reject_env) : 0;
            for (int i = 0; reject_every > 0 && i < nb; i += reject_every) {
                g->status[(size_t)i] = 1;
                HIP_TRY(hipMemset(text + g->out_off[(size_t)i], '#', c->isize[(size_t)i]));
            }
            bool patched = false;
            for (int i = 0; i < nb; ++i) {
                if (!g->status[(size_t)i]) continue;
                g->line.resize(65536);
                if (!reader_inflate_member(c->comp + c->in_off[(size_t)i], c->in_len[(size_t)i], c->isize[(size_t)i], reinterpret_cast<unsigned char *>(g->line.data()))) {
                    wgs_set_error("read error in the BGZF file (corrupt block)");
                    return 1;
                }
                HIP_TRY(hipMemcpy(text + g->out_off[(size_t)i], g->line.data(), c->isize[(size_t)i], hipMemcpyHostToDevice));
                ++g->blocks_host;
                patched = true;
            }
            if (!patched) break;
        }
        if (nb) {
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, g->iev0, g->iev1);
            g->inflate_kernel_ms += ms;
            g->blocks_inflated += nb;
        }
        const double t_inflated = now_s();
        reader_comp_release(g->r, c);                                   // the producer may refill it while the device works on
        release.c = nullptr;
        c = nullptr;
        const size_t nl = g->h_totals[T_NEWLINES];
        g->text_bytes += (int64_t)(total - g->carry);
        g->chunks += 1;
        if (nl == 0) {                                                    // not one whole line yet
            g->carry = total;
            if (last) g->done = true;
            continue;
        }
        if (int rc = ensure_line_arrays(g, st, nl, true)) return rc;
        hipLaunchKernelGGL(write_newlines_kernel, dim3((unsigned)nblk), dim3(256), 0, st, reinterpret_cast<const uint4 *>(text), (uint64_t)total, g->d_offsets, g->d_nl_pos);
        LineArgs la;
        la.text = text;
        la.nl_pos = g->d_nl_pos;
        la.nlines = (uint32_t)nl;
        la.begin = g->d_begin;
        la.end = g->d_end;
        la.nonblank = g->d_nonblank;
        la.name_start = g->d_name_start;
        la.name_len1 = g->d_name_len1;
        la.totals = g->d_totals;
        hipLaunchKernelGGL(line_info_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, la);
        hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, st, g->d_nonblank, g->d_rank, (uint32_t)nl, g->d_totals + T_NONBLANK);
        hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, st, g->d_name_len1, g->d_name_off, (uint32_t)nl, g->d_totals + T_NAME_BYTES);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(g->h_totals, g->d_totals, T_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const double t_listed = now_s();
        const size_t rows_here = g->h_totals[T_NONBLANK];
        size_t take = rows_here;
        if (g->limit >= 0 && (int64_t)take >= g->limit - g->rows_done) {
            take = (size_t)(g->limit - g->rows_done);
            g->done = true;
        }
        if (last) g->done = true;
        int64_t written = (int64_t)take;
        if (keep) {
            WGS_REQUIRE((int64_t)take <= keep_len, "site mask shorter than the file (%lld lines left in it, %lld in the chunk)", (long long)keep_len, (long long)take);
            g->dstmap.resize(take);
            written = 0;
            for (size_t i = 0; i < take; ++i) g->dstmap[i] = keep[i] ? (int32_t)written++ : -1;
            if (take) HIP_TRY(hipMemcpyAsync(g->d_dstmap, g->dstmap.data(), take * sizeof(int32_t), hipMemcpyHostToDevice, st));
        }
        WGS_REQUIRE(row0 >= 0 && row0 + written <= b->m, "rows [%lld, %lld) outside the device matrix (%lld rows)", (long long)row0, (long long)(row0 + written), (long long)b->m);
        const size_t name_bytes = g->h_totals[T_NAME_BYTES];
        if (name_bytes + 16 > g->names_cap) {
            g->names_cap = 0;
            if (int rc = regrow(st, g->d_names, name_bytes + name_bytes / 2 + 4096)) return rc;
            g->names_cap = name_bytes + name_bytes / 2 + 4096;
        }
        DstArgs da;
        da.nlines = (uint32_t)nl;
        da.take = (uint32_t)take;
        da.nonblank = g->d_nonblank;
        da.rank = g->d_rank;
        da.name_start = g->d_name_start;
        da.name_len1 = g->d_name_len1;
        da.name_off = g->d_name_off;
        da.dstmap = keep ? g->d_dstmap : nullptr;
        da.text = text;
        da.dst = g->d_dst;
        da.names = g->d_names;
        da.totals = g->d_totals;
        hipLaunchKernelGGL(dst_names_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, da);
        HIP_TRY(hipMemsetAsync(g->d_flags, 0, nl, st));
        launch_tokenise(g, st, text, row0, nl, g->d_totals + T_FLAGGED);
        HIP_TRY(hipGetLastError());
        g->names.resize(name_bytes);
        if (name_bytes) HIP_TRY(hipMemcpyAsync(&g->names[0], g->d_names, name_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(g->h_totals, g->d_totals, T_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipEventRecord(g->ev1, st));
        HIP_TRY(hipStreamSynchronize(st));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, g->ev0, g->ev1));
        g->device_ms += ms;
        if (take < rows_here) g->names.resize(g->h_totals[T_NAME_CUT]);
        if (g->h_totals[T_FLAGGED]) {
            // rare: fetch what the host parser needs -- flags, extents, numbering -- and the flagged lines themselves
            g->flags.resize(nl);
            g->dst.resize(nl);
            g->h_begin.resize(nl);
            g->h_end.resize(nl);
            g->h_rank.resize(nl);
            HIP_TRY(hipMemcpy(g->flags.data(), g->d_flags, nl, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(g->dst.data(), g->d_dst, nl * sizeof(int32_t), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(g->h_begin.data(), g->d_begin, nl * sizeof(uint32_t), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(g->h_end.data(), g->d_end, nl * sizeof(uint32_t), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(g->h_rank.data(), g->d_rank, nl * sizeof(uint32_t), hipMemcpyDeviceToHost));
            auto text_of = [&](size_t t, const char **lb, const char **le) -> int {
                const size_t n = g->h_end[t] - g->h_begin[t];
                g->line.resize(n + 1);
                HIP_TRY(hipMemcpy(g->line.data(), text + g->h_begin[t], n, hipMemcpyDeviceToHost));
                *lb = g->line.data();
                *le = g->line.data() + n;
                return 0;
            };
            if (int rc = host_parse_flagged(g, row0, nl, g->flags.data(), g->dst.data(), g->rows_done, text_of, [&](size_t t) { return g->h_rank[t]; })) return rc;
        }
        // the partial last line moves to the front for the next chunk (through a side buffer: the two ranges may overlap);
        // queued behind the tokeniser, waited for by nobody but the next chunk's kernels
        const size_t tail = g->h_totals[T_TAIL];
        const size_t left = g->done ? 0 : total - tail;
        if (left) {
            if (left > g->carry_cap) {
                g->carry_cap = 0;
                if (int rc = regrow(st, g->d_carry, left + left / 2 + 65536)) return rc;
                g->carry_cap = left + left / 2 + 65536;
            }
            HIP_TRY(hipMemcpyAsync(g->d_carry, text + tail, left, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(text, g->d_carry, left, hipMemcpyDeviceToDevice, st));
        }
        g->carry = left;
        g->rows_done += (int64_t)take;
        g->lines += (int64_t)take;
        if (trace)
            fprintf(stderr, "ingest chunk %lld: %zu B text, %d members, %zu lines | wait %.1f ms, buffers %.1f, copy+inflate %.1f, list %.1f, rows+names+tokenise %.1f\n",
                    (long long)g->chunks, total, nb, nl, waited * 1e3, (t_alloc - t_begin - waited) * 1e3, (t_inflated - t_alloc) * 1e3,
                    (t_listed - t_inflated) * 1e3, (now_s() - t_listed) * 1e3);
        if (take == 0) continue;
        *file_rows = (int64_t)take;
        *rows_written = written;
        return 0;
    }
    return 0;
}

}  // namespace

extern "C" {

void wgs_ingest_destroy(wgs_ingest *g)
{
    if (!g) return;
    (void)hipSetDevice(g->b->ctx->device);
    (void)hipStreamSynchronize(g->b->ctx->stream);
    if (g->resident) {
        reader_add_lines_read(g->r, g->rows_done);
        reader_comp_stop(g->r);
    }
    reader_text_stop(g->r);                                    // joins the producer, frees the pinned buffers
    for (void *p : {g->d_text, (void *)g->d_begin, (void *)g->d_end, (void *)g->d_dst, (void *)g->d_flags, (void *)g->d_carry, (void *)g->d_comp,
                    (void *)g->d_in_off, (void *)g->d_out_off, (void *)g->d_in_len, (void *)g->d_isize, (void *)g->d_status, g->d_tables,
                    (void *)g->d_counts, (void *)g->d_offsets, (void *)g->d_nl_pos, (void *)g->d_nonblank, (void *)g->d_rank, (void *)g->d_name_start,
                    (void *)g->d_name_len1, (void *)g->d_name_off, (void *)g->d_dstmap, (void *)g->d_names, (void *)g->d_totals})
        if (p) (void)hipFree(p);
    if (g->h_totals) (void)hipHostFree(g->h_totals);
    for (hipEvent_t e : {g->ev0, g->ev1, g->iev0, g->iev1})
        if (e) (void)hipEventDestroy(e);
    delete g;
}

/* chunk_bytes: text per chunk (<= 0: the default -- 256 MiB through the host inflater; 3 GiB -- or the members one launch has lanes for -- when the device inflates, one
 * lane per BGZF member: the more members per launch, the better the chip is used). */
int wgs_ingest_create(wgs_beagle *b, wgs_reader *r, int64_t limit_rows, int64_t chunk_bytes, wgs_ingest **out)
{
    WGS_REQUIRE(b && r && out, "null argument");
    const double t_create = now_s();
    WGS_REQUIRE(reader_text_n_inds(r) == b->n, "the Beagle file has %d individuals, the device matrix %lld", reader_text_n_inds(r),
                (long long)b->n);
    HIP_TRY(hipSetDevice(b->ctx->device));
    // BGZF (what ANGSD writes): inflated on the device unless WGSASSIGN_INFLATE says host / zlib
    const char *how = getenv("WGSASSIGN_INFLATE");
    const bool resident = reader_text_is_bgzf(r) && !(how && (strcmp(how, "host") == 0 || strcmp(how, "zlib") == 0));
    // (one lane per member and three wavefronts per CU: a launch of up to 49 k members takes the time of one member, so a
    // chunk is that many members -- see reader_comp_start below -- or 3 GiB of text, whichever comes first)
    if (chunk_bytes <= 0) chunk_bytes = resident ? (3ll << 30) : (256ll << 20);
    chunk_bytes = std::min<int64_t>(chunk_bytes, resident ? (3ll << 30) : (1ll << 30));
    wgs_ingest *g = new wgs_ingest();
    g->b = b;
    g->r = r;
    g->limit = limit_rows;
    TextAllocator a;
    a.alloc = pinned_alloc;
    a.release = pinned_release;
    a.user = (void *)(intptr_t)b->ctx->device;
    auto guard = on_failure([&] { wgs_ingest_destroy(g); });
    HIP_TRY(hipEventCreate(&g->ev0));
    HIP_TRY(hipEventCreate(&g->ev1));
    if (resident) {
        HIP_TRY(hipEventCreate(&g->iev0));
        HIP_TRY(hipEventCreate(&g->iev1));
        HIP_TRY(wgs_malloc(reinterpret_cast<void **>(&g->d_totals), T_COUNT * sizeof(uint32_t)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g->h_totals), T_COUNT * sizeof(uint32_t), hipHostMallocDefault));
        g->chunk_text = (size_t)std::max<int64_t>(chunk_bytes, 1 << 20);
        g->resident = true;
        if (limit_rows == 0) g->done = true;
        // page-locked staging for the compressed members: an eighth of the text (low-depth ANGSD output deflates 10 : 1 and
        // more; where a file compresses less a chunk simply ends early), one buffer when the rest of the file fits into it
        else {
            size_t staging = std::max<size_t>(g->chunk_text / 8, 1u << 20);
            const int64_t left = reader_comp_bytes_left(r);
            int nbuf = 2;
            if (left >= 0 && (size_t)left + 4096 <= staging) {
                staging = std::max<size_t>(((size_t)left + 4096 + 0xFFFFF) & ~(size_t)0xFFFFF, 1u << 20);
                nbuf = 1;
            }
            // one lane per member and three wavefronts per CU (52 KiB of tables each): members beyond that many wait for a
            // second round of the launch
            const size_t lanes = (size_t)std::max(1, b->ctx->cus) * 3 * 64;
            if (int rc = reader_comp_start(r, staging, g->chunk_text, nbuf, a, lanes)) return rc;
        }
    } else if (int rc = reader_text_start(r, (size_t)chunk_bytes, 3, a, limit_rows)) {
        return rc;
    }
    guard.dismiss();
    g->create_s = now_s() - t_create;
    *out = g;
    return 0;
}

/* The next chunk of the file: its lines are tokenised on the device into the slab rows row0, row0 + 1, ...
 * (keep != NULL: keep[i] says whether the i-th line of THIS chunk is kept; dropped lines take no row).
 * *file_rows = lines of the chunk (0 at the end of the file / the row limit), *rows_written = rows they filled. */
int wgs_ingest_next(wgs_ingest *g, int64_t row0, const uint8_t *keep, int64_t keep_len, int64_t *file_rows, int64_t *rows_written)
{
    WGS_REQUIRE(g && file_rows && rows_written, "null argument");
    wgs_beagle *b = g->b;
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    *file_rows = *rows_written = 0;
    g->names.clear();
    wgs_beagle_drop_codes(b);                                  // the matrix changes: its class codes are rebuilt on next use
    struct Clock {
        wgs_ingest *g;
        double t0;
        ~Clock() { g->next_s += now_s() - t0; }
    } clock{g, now_s()};
    if (g->resident) return ingest_next_resident(g, row0, keep, keep_len, file_rows, rows_written);
    TextChunk *c = nullptr;
    double waited = 0.0;
    if (int rc = reader_text_next(g->r, &c, &waited)) return rc;
    g->wait_s += waited;
    if (!c) return 0;
    struct Release {
        wgs_ingest *g;
        TextChunk *c;
        ~Release() { reader_text_release(g->r, c); }
    } release{g, c};
    const size_t nl = c->begin.size();
    if (keep) WGS_REQUIRE((int64_t)nl <= keep_len, "site mask shorter than the file (%lld lines left in it, %lld in the chunk)", (long long)keep_len, (long long)nl);
    g->dst.resize(nl);
    int64_t written = 0;
    for (size_t i = 0; i < nl; ++i) g->dst[i] = (!keep || keep[i]) ? (int32_t)written++ : -1;
    WGS_REQUIRE(row0 >= 0 && row0 + written <= b->m, "rows [%lld, %lld) outside the device matrix (%lld rows)", (long long)row0,
                (long long)(row0 + written), (long long)b->m);
    // device buffers (grow only)
    const size_t text_bytes = (c->len + TEXT_PAD + 15) & ~(size_t)15;
    if (text_bytes > g->text_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (g->d_text) HIP_TRY(hipFree(g->d_text));
        g->d_text = nullptr;
        g->text_cap = 0;
        const size_t cap = std::max(text_bytes, c->cap);
        HIP_TRY(wgs_malloc(&g->d_text, cap));
        g->text_cap = cap;
    }
    if (int rc = ensure_line_arrays(g, st, nl, false)) return rc;
    HIP_TRY(hipEventRecord(g->ev0, st));
    HIP_TRY(hipMemcpyAsync(g->d_text, c->data, text_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g->d_begin, c->begin.data(), nl * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g->d_end, c->end.data(), nl * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g->d_dst, g->dst.data(), nl * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(g->d_flags, 0, nl, st));
    launch_tokenise(g, st, g->d_text, row0, nl, nullptr);
    HIP_TRY(hipGetLastError());
    g->flags.resize(nl);
    HIP_TRY(hipMemcpyAsync(g->flags.data(), g->d_flags, nl, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(g->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    g->device_ms += ms;
    auto text_of = [&](size_t t, const char **lb, const char **le) -> int {
        *lb = c->data + c->begin[t];
        *le = c->data + c->end[t];
        return 0;
    };
    if (int rc = host_parse_flagged(g, row0, nl, g->flags.data(), g->dst.data(), c->first_row, text_of, [](size_t t) { return t; })) return rc;
    g->names.swap(c->names);
    g->inflate_s += c->inflate_s;
    g->scan_s += c->scan_s;
    g->text_bytes += (int64_t)c->len;
    g->lines += (int64_t)nl;
    g->chunks += 1;
    *file_rows = (int64_t)nl;
    *rows_written = written;
    return 0;
}

/* Site names of the chunk wgs_ingest_next just returned (every line of it, kept or not), '\n'-terminated each. */
const char *wgs_ingest_chunk_sites(wgs_ingest *g, int64_t *bytes)
{
    if (!g) return nullptr;
    if (bytes) *bytes = (int64_t)g->names.size();
    return g->names.c_str();
}

/* stats[0..13]: seconds the consumer waited for the producer, producer seconds in inflate and in the newline scan (host
 * inflate), device milliseconds (copies + all kernels), lines parsed on the host, text bytes, lines, chunks, milliseconds of
 * the device inflate kernel, BGZF members inflated on the device, members the device left to the host's inflater, producer
 * seconds reading compressed members (device inflate), seconds in wgs_ingest_create and in wgs_ingest_next. */
int wgs_ingest_stats(wgs_ingest *g, double *stats)
{
    WGS_REQUIRE(g && stats, "null argument");
    stats[0] = g->wait_s;
    stats[1] = g->inflate_s;
    stats[2] = g->scan_s;
    stats[3] = g->device_ms;
    stats[4] = (double)g->host_lines;
    stats[5] = (double)g->text_bytes;
    stats[6] = (double)g->lines;
    stats[7] = (double)g->chunks;
    stats[8] = g->inflate_kernel_ms;
    stats[9] = (double)g->blocks_inflated;
    stats[10] = (double)g->blocks_host;
    stats[11] = g->read_s;
    stats[12] = g->create_s;
    stats[13] = g->next_s;
    return 0;
}

}  // extern "C"
