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

#include <chrono>
#include <string>
#include <vector>

#include "common.h"
#include "reader_text.h"

namespace {

__constant__ double kPow10Dev[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                     1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

struct TokArgs {
    const uint4 *text;             // the chunk, 16-byte aligned, padded with newlines
    const uint32_t *begin, *end;   // per line: [begin, end) in bytes
    const int32_t *dst;            // per line: row relative to row0, or -1 (site filtered out)
    uint8_t *flags;                // per line: 1 = the host must parse this line
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
    if (lane == 0) a.flags[line] = any_bad ? 1 : 0;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void *pinned_alloc(size_t bytes, void *user)                  // called from the producer thread too
{
    void *p = nullptr;
    if (hipSetDevice((int)(intptr_t)user) != hipSuccess) return nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
void pinned_release(void *p, void *) { (void)hipHostFree(p); }

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
};

extern "C" {

void wgs_ingest_destroy(wgs_ingest *g)
{
    if (!g) return;
    (void)hipSetDevice(g->b->ctx->device);
    (void)hipStreamSynchronize(g->b->ctx->stream);
    reader_text_stop(g->r);                                    // joins the producer, frees the pinned buffers
    if (g->d_text) (void)hipFree(g->d_text);
    if (g->d_begin) (void)hipFree(g->d_begin);
    if (g->d_end) (void)hipFree(g->d_end);
    if (g->d_dst) (void)hipFree(g->d_dst);
    if (g->d_flags) (void)hipFree(g->d_flags);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    delete g;
}

int wgs_ingest_create(wgs_beagle *b, wgs_reader *r, int64_t limit_rows, int64_t chunk_bytes, wgs_ingest **out)
{
    WGS_REQUIRE(b && r && out, "null argument");
    WGS_REQUIRE(reader_text_n_inds(r) == b->n, "the Beagle file has %d individuals, the device matrix %lld", reader_text_n_inds(r),
                (long long)b->n);
    HIP_TRY(hipSetDevice(b->ctx->device));
    if (chunk_bytes <= 0) chunk_bytes = 256ll << 20;
    chunk_bytes = std::min<int64_t>(chunk_bytes, 1ll << 30);
    wgs_ingest *g = new wgs_ingest();
    g->b = b;
    g->r = r;
    TextAllocator a;
    a.alloc = pinned_alloc;
    a.release = pinned_release;
    a.user = (void *)(intptr_t)b->ctx->device;
    if (int rc = reader_text_start(r, (size_t)chunk_bytes, 3, a, limit_rows)) {
        delete g;
        return rc;
    }
    auto guard = on_failure([&] { wgs_ingest_destroy(g); });
    HIP_TRY(hipEventCreate(&g->ev0));
    HIP_TRY(hipEventCreate(&g->ev1));
    guard.dismiss();
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
        HIP_TRY(hipMalloc(&g->d_text, cap));
        g->text_cap = cap;
    }
    if (nl > g->lines_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        for (void *p : {(void *)g->d_begin, (void *)g->d_end, (void *)g->d_dst, (void *)g->d_flags})
            if (p) HIP_TRY(hipFree(p));
        g->d_begin = g->d_end = nullptr;
        g->d_dst = nullptr;
        g->d_flags = nullptr;
        g->lines_cap = 0;
        const size_t cap = nl + nl / 2 + 1024;
        HIP_TRY(hipMalloc(&g->d_begin, cap * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&g->d_end, cap * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&g->d_dst, cap * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&g->d_flags, cap));
        g->lines_cap = cap;
    }
    HIP_TRY(hipEventRecord(g->ev0, st));
    HIP_TRY(hipMemcpyAsync(g->d_text, c->data, text_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g->d_begin, c->begin.data(), nl * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g->d_end, c->end.data(), nl * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g->d_dst, g->dst.data(), nl * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(g->d_flags, 0, nl, st));
    TokArgs a;
    a.text = reinterpret_cast<const uint4 *>(g->d_text);
    a.begin = g->d_begin;
    a.end = g->d_end;
    a.dst = g->d_dst;
    a.flags = g->d_flags;
    a.row0 = row0;
    a.nlines = (int32_t)nl;
    a.n_inds = (int32_t)b->n;
    a.group_of = b->d_group_of;
    a.col_of = b->d_col_of;
    a.npairs = b->d_npairs;
    a.base = b->d_base;
    hipLaunchKernelGGL(tokenise_kernel, dim3((unsigned)((nl + 3) / 4)), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    g->flags.resize(nl);
    HIP_TRY(hipMemcpyAsync(g->flags.data(), g->d_flags, nl, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(g->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    g->device_ms += ms;
    // flagged lines: the strtod-backed host parser, uploaded in runs of consecutive rows
    const size_t row_floats = (size_t)2 * (size_t)b->n;
    for (size_t i = 0; i < nl;) {
        if (!g->flags[i] || g->dst[i] < 0) {
            ++i;
            continue;
        }
        size_t j = i;
        while (j < nl && g->flags[j] && g->dst[j] == g->dst[i] + (int32_t)(j - i) && j - i < 4096) ++j;
        g->rows.resize((j - i) * row_floats);
        for (size_t t = i; t < j; ++t)
            if (reader_text_parse_line(g->r, c->data + c->begin[t], c->data + c->end[t], g->rows.data() + (t - i) * row_floats)) {
                wgs_set_error("Beagle data line %lld has fewer than %d genotype-likelihood columns",
                              (long long)(reader_text_lines_read(g->r) + c->first_row + (int64_t)t + 2), reader_text_gl_cols(g->r));
                return 2;
            }
        if (int rc = wgs_beagle_upload_rows(b, g->rows.data(), row0 + g->dst[i], (int64_t)(j - i))) return rc;
        g->host_lines += (int64_t)(j - i);
        i = j;
    }
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

/* stats[0..7]: seconds the consumer waited for text, producer seconds in inflate and in the newline scan, device
 * milliseconds (H2D + tokeniser), lines parsed on the host, text bytes, lines, chunks. */
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
    return 0;
}

}  // extern "C"
