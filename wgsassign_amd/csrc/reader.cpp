// Streamed Beagle reader -- the native counterpart of reader_cy.readBeagle (reader_cy.pyx:16-77).
//
// The reference pipes `gunzip -c` through Python, tokenises every line with strtok(" \t\n") and
// converts with atof into a vector<vector<float>> that is then copied into a NumPy array (two
// copies of the matrix in RAM, ~30 k sites/s).  Here: zlib inflate into a large buffer, lines
// parsed in parallel by worker threads straight into the caller's float32 (rows, 2n) chunk --
// chunk by chunk, so a file larger than host RAM can be streamed into device slabs.
//
// Parity rules kept (reader_cy.pyx:35-66): header tokens after the first three name the GL
// columns, every third one (c % 3 == 1) is a sample name; per line token 0 is the site name, two
// allele tokens are skipped, of every GL triple the first two values are kept and the third
// dropped; a value is atof(token) rounded to float32.  atof == strtod: decimal tokens with at
// most 15 significant digits and no exponent take the exact fast path (integer mantissa / power
// of ten, one correctly rounded division -- identical to strtod for such inputs); anything else
// (exponents, inf/nan, hex) goes through strtod itself.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wgsassign_hip.h"
#include "../../include/wgsassign_hip_debug.h"
#include "reader_text.h"

void wgs_set_error(const char *fmt, ...);

namespace {

template <typename F>
void run_threads(int T, F work)
{
    if (T <= 1) {
        work(0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
}

inline bool is_delim(char c) { return c == '\t' || c == ' ' || c == '\n' || c == '\r'; }

const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// atof of the token [p, e) (no delimiters inside).  Decimal tokens -- [sign] digits [. digits] [e [sign] digits] -- whose
// mantissa fits 53 bits and whose net power of ten is at most 22 in magnitude take the exact path: one correctly
// rounded multiplication or division of two exactly representable numbers, which is the correctly rounded value of
// the token, i.e. strtod's result.  Everything else (longer mantissas, inf/nan, hex, trailing junk) is strtod's.
// ingest.hip holds the device twin of this function (tokenise_kernel): the two must accept the same tokens.
inline double parse_double(const char *p, const char *e)
{
    const char *q = p;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) neg = *q++ == '-';
    uint64_t mant = 0;
    int digits = 0, frac = 0, seen = 0;
    bool dot = false, ok = true;
    for (; q < e; ++q) {
        const char c = *q;
        if (c >= '0' && c <= '9') {
            ++seen;
            if (mant == 0 && c == '0') {                     // zeros before the first significant digit
                if (dot) ++frac;
                continue;
            }
            if (++digits > 15) break;
            mant = mant * 10 + (uint64_t)(c - '0');
            if (dot) ++frac;
        } else if (c == '.' && !dot) {
            dot = true;
        } else {
            break;
        }
    }
    int ex = 0;
    if (q < e && (*q == 'e' || *q == 'E') && seen > 0 && digits <= 15) {
        const char *x = q + 1;
        bool xneg = false;
        if (x < e && (*x == '-' || *x == '+')) xneg = *x++ == '-';
        int xd = 0;
        for (; x < e && *x >= '0' && *x <= '9' && xd < 4; ++x, ++xd) ex = ex * 10 + (*x - '0');
        if (xd == 0 || xd > 3) ok = false;                   // "1e", "1e+": atof stops before the 'e'
        if (xneg) ex = -ex;
        q = x;
    }
    const int net = ex - frac;
    if (ok && q == e && seen > 0 && digits <= 15 && net >= -22 && net <= 22) {
        const double v = net < 0 ? (double)mant / kPow10[-net] : (double)mant * kPow10[net];
        return neg ? -v : v;
    }
    char tmp[64];
    const size_t len = (size_t)(e - p);
    if (len < sizeof tmp) {
        memcpy(tmp, p, len);
        tmp[len] = 0;
        return strtod(tmp, nullptr);
    }
    std::string str(p, e);
    return strtod(str.c_str(), nullptr);
}

// The token almost every Beagle file consists of: "d.dddddd" (ANGSD prints %f) followed by a delimiter or the end of
// the line.  Eight bytes are checked and converted at once: XOR with "0.000000" leaves the digit values (and 0 for the
// point), the pairwise multiply-adds of the usual eight-digit trick give d0 d2 .. d7 0 = 10 * mantissa, and the value is
// that integer / 1e7 -- one correctly rounded division of two exactly representable numbers, i.e. strtod's result.
inline bool parse_f6(const char *p, const char *e, double *v)
{
    if (e - p < 8 || (e - p > 8 && !is_delim(p[8]))) return false;
    uint64_t w;
    memcpy(&w, p, 8);
    uint64_t d = w ^ 0x3030303030302E30ull;
    if ((((d + 0x7676767676767676ull) | d) & 0x8080808080808080ull) || (d & 0xFF00ull)) return false;
    d = (d >> 8) | (d & 0xFF);                               // first digit, then the six decimals, then 0
    d = d * 10 + (d >> 8);
    const uint64_t mask = 0x000000FF000000FFull;
    d = (((d & mask) * (100 + (1000000ull << 32))) + (((d >> 16) & mask) * (1 + (10000ull << 32)))) >> 32;
    *v = (double)(uint32_t)d / 1e7;
    return true;
}

struct Line {
    const char *begin, *end;   // without the newline
};

}  // namespace

namespace {

// ---- gzip input with random access ------------------------------------------------------------------
// A gzip stream can only be inflated from its start -- unless one keeps, for chosen deflate-block
// boundaries, the 32 KiB of output that precede them (the dictionary the next block may refer to).
// wgs_reader_build_index makes ONE inflate pass over the file, counting the data lines (sites) and
// recording such access points (every `span` bytes of output; at the start of a gzip member -- BGZF
// files have one per 64 KiB -- no dictionary is needed); a reader opened from the index starts at the
// access point before its first row, so a rank that owns a later SNP range no longer inflates the
// ranges before it.
constexpr size_t GZ_WIN = 32768;

struct AccessPoint {
    uint64_t in = 0;            // compressed offset of the first byte not yet consumed
    uint64_t out = 0;           // uncompressed offset
    int64_t lines_before = 0;   // non-blank lines (header included) completed before `out`
    uint8_t bits = 0;           // bits of byte in-1 that belong to the next block (0: byte aligned)
    uint8_t content = 0;        // the line in progress at `out` already holds a non-delimiter character
    uint8_t at_line_start = 0;  // `out` is the first byte of a line
    uint8_t member_start = 0;   // a gzip member header starts at `in` (no dictionary needed)
    std::vector<unsigned char> window;   // GZ_WIN bytes of output before `out` (empty at a member start)
};

// BGZF (bgzip, htslib -- what ANGSD writes): a series of gzip members of at most 64 KiB, each announcing its own
// compressed size in a 'BC' extra subfield and its uncompressed size in the trailer, so the blocks can be found
// without inflating anything and inflated independently, in parallel.
struct BgzfBlock {
    uint64_t off = 0;       // file offset of the member
    uint32_t csize = 0;     // whole member, header and trailer included
    uint32_t hdr = 0;       // bytes before the deflate data
    uint32_t isize = 0;     // uncompressed bytes
    uint32_t unused = 0;    // (explicit padding: the part files of the split index pass hold these records verbatim)
};

// Parses the member header at p (n bytes available): 0 = not BGZF, -1 = need more bytes, else the member size.
inline long bgzf_member_size(const unsigned char *p, size_t n, uint32_t *hdr)
{
    if (n < 12) return -1;
    if (p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = p[10] | ((size_t)p[11] << 8);
    if (n < 12 + xlen) return -1;
    for (size_t q = 12; q + 4 <= 12 + xlen;) {
        const size_t slen = p[q + 2] | ((size_t)p[q + 3] << 8);
        if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) {
            if ((p[3] & ~4) != 0) return 0;      // names / comments / header CRC: not what bgzip writes
            *hdr = (uint32_t)(12 + xlen);
            return (long)(p[q + 4] | ((size_t)p[q + 5] << 8)) + 1;
        }
        q += 4 + slen;
    }
    return 0;
}

// Whole-buffer inflate of BGZF members.  libdeflate does that 2-3x faster than zlib; this image ships its shared
// library (libdeflate.so.0, 1.10) without the header, hence the three prototypes and dlopen.  Without the library, or
// with WGSASSIGN_INFLATE=zlib: zlib's raw inflate.  Neither checks the member's CRC32 (gzread does; a corrupt block
// that still inflates to ISIZE bytes would show up as unparsable text).
struct LibDeflate {
    void *(*alloc)() = nullptr;
    void (*release)(void *) = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    bool ok = false;
    LibDeflate()
    {
        const char *want = getenv("WGSASSIGN_INFLATE");
        if (want && strcmp(want, "zlib") == 0) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = reinterpret_cast<void *(*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
        release = reinterpret_cast<void (*)(void *)>(dlsym(h, "libdeflate_free_decompressor"));
        decompress = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *)>(dlsym(h, "libdeflate_deflate_decompress"));
        ok = alloc && release && decompress;
    }
};
inline const LibDeflate &libdeflate()
{
    static const LibDeflate L;
    return L;
}

struct BlockInflater {
    z_stream z;
    bool z_live = false;
    void *ld = nullptr;
    BlockInflater() { memset(&z, 0, sizeof z); }
    BlockInflater(const BlockInflater &) = delete;
    BlockInflater &operator=(const BlockInflater &) = delete;
    ~BlockInflater()
    {
        if (ld) libdeflate().release(ld);
        if (z_live) inflateEnd(&z);
    }
    bool init()
    {
        if (libdeflate().ok && (ld = libdeflate().alloc()) != nullptr) return true;
        z_live = inflateInit2(&z, -15) == Z_OK;
        return z_live;
    }
    // one member (raw deflate between its header and trailer) into out[0 .. isize)
    bool run(const unsigned char *member, const BgzfBlock &b, unsigned char *out)
    {
        if (b.isize == 0) return true;
        if (ld) return libdeflate().decompress(ld, member + b.hdr, b.csize - b.hdr - 8, out, b.isize, nullptr) == 0;
        if (inflateReset(&z) != Z_OK) return false;
        z.next_in = const_cast<unsigned char *>(member) + b.hdr;
        z.avail_in = b.csize - b.hdr - 8;
        z.next_out = out;
        z.avail_out = b.isize;
        return inflate(&z, Z_FINISH) == Z_STREAM_END && z.avail_out == 0;
    }
};

struct GzSource {
    FILE *fp = nullptr;
    z_stream z;
    bool z_live = false, raw = false, eof = false;
    std::vector<unsigned char> in;
    // BGZF mode
    bool bgzf = false;
    int threads = 1;
    // (not a std::vector: resize() would zero its 64 MiB at every open -- 13 ms on the GPU box, 30 ms elsewhere, a seventh of a whole
    // device ingest of 5.4 GB of text -- for a buffer that the device ingest fills with one member)
    struct RawBuf {
        unsigned char *p = nullptr;
        size_t n = 0;
        ~RawBuf() { free(p); }
        RawBuf() = default;
        RawBuf(const RawBuf &) = delete;
        RawBuf &operator=(const RawBuf &) = delete;
        unsigned char *data() const { return p; }
        size_t size() const { return n; }
        void resize(size_t want)
        {
            if (want <= n) return;
            unsigned char *q = (unsigned char *)realloc(p, want);
            if (!q) throw std::bad_alloc();
            p = q;
            n = want;
        }
    } cbuf;
    size_t clen = 0, cpos = 0;
    std::vector<BgzfBlock> batch;
    // segmented mode: a plain-gzip file read through its index -- the stretches between consecutive access points
    // are inflated independently, each from its own 32 KiB dictionary, `threads` at a time (BGZF blocks, megabytes long)
    bool segmented = false;
    std::string seg_path;
    std::vector<AccessPoint> segs;   // stretch i = output [segs[i].out, segs[i+1].out); the last point runs to the end
    size_t seg_next = 0;
    size_t in_bytes = 4u << 20;      // compressed staging buffer of the serial stream

    ~GzSource() { close(); }
    void close()
    {
        if (z_live) inflateEnd(&z);
        z_live = false;
        if (fp) fclose(fp);
        fp = nullptr;
    }
    bool refill()
    {
        if (z.avail_in != 0) return true;
        const size_t got = fread(in.data(), 1, in.size(), fp);
        z.next_in = in.data();
        z.avail_in = (unsigned)got;
        return got != 0;
    }
    // from the first byte (ap == nullptr) or from an access point
    bool open(const char *path, const AccessPoint *ap)
    {
        close();
        fp = fopen(path, "rb");
        if (!fp) return false;
        in.resize(in_bytes);
        memset(&z, 0, sizeof z);
        eof = false;
        raw = ap && !ap->member_start;
        if (inflateInit2(&z, raw ? -15 : 15 + 32) != Z_OK) return false;
        z_live = true;
        if (!ap) return true;
        if (fseeko(fp, (off_t)(ap->in - (ap->bits ? 1 : 0)), SEEK_SET) != 0) return false;
        if (ap->bits) {
            const int c = getc(fp);
            if (c == EOF) return false;
            if (inflatePrime(&z, ap->bits, c >> (8 - ap->bits)) != Z_OK) return false;
        }
        if (raw && inflateSetDictionary(&z, ap->window.data(), (unsigned)ap->window.size()) != Z_OK) return false;
        return true;
    }
    // After open() at the start of a member: switch to parallel block inflation if the file is BGZF from here on
    // (checked block by block as they are read; a non-BGZF member later is an error, such files do not exist).
    void try_bgzf(int nthreads)
    {
        if (raw) return;
        const off_t here = ftello(fp);
        unsigned char head[64];
        const size_t got = fread(head, 1, sizeof head, fp);
        fseeko(fp, here, SEEK_SET);
        uint32_t hdr = 0;
        if (bgzf_member_size(head, got, &hdr) > 0) {
            bgzf = true;
            threads = nthreads > 0 ? nthreads : 1;
            cbuf.resize(64u << 20);
            clen = cpos = 0;
        }
    }
    long read_bgzf(char *dst, size_t cap)
    {
        for (;;) {
            // whole members available in cbuf[cpos, clen) that fit into `cap`
            batch.clear();
            size_t p = cpos, out = 0;
            bool need_more = false;
            while (p < clen) {
                uint32_t hdr = 0;
                const long sz = bgzf_member_size(cbuf.data() + p, clen - p, &hdr);
                if (sz == 0) return -1;
                if (sz < 0 || p + (size_t)sz > clen) {
                    need_more = true;
                    break;
                }
                BgzfBlock b;
                b.off = p;
                b.csize = (uint32_t)sz;
                b.hdr = hdr;
                const unsigned char *t = cbuf.data() + p + sz - 4;
                b.isize = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                if (b.isize > 65536 || b.csize < hdr + 8) return -1;
                if (out + b.isize > cap) break;
                batch.push_back(b);
                out += b.isize;
                p += (size_t)sz;
            }
            if (!batch.empty()) {
                std::vector<size_t> ooff(batch.size());
                size_t acc = 0;
                for (size_t i = 0; i < batch.size(); ++i) ooff[i] = acc, acc += batch[i].isize;
                const int T = (int)std::min<size_t>((size_t)threads, batch.size());
                std::vector<char> ok(T, 1);
                auto work = [&](int t) {
                    BlockInflater inf;
                    if (!inf.init()) {
                        ok[t] = 0;
                        return;
                    }
                    for (size_t i = (size_t)t; i < batch.size(); i += (size_t)T)
                        if (!inf.run(cbuf.data() + batch[i].off, batch[i], reinterpret_cast<unsigned char *>(dst) + ooff[i])) ok[t] = 0;
                };
                if (T <= 1) {
                    work(0);
                } else {
                    std::vector<std::thread> th;
                    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
                    for (auto &x : th) x.join();
                }
                for (char c : ok)
                    if (!c) return -1;
                cpos = p;
                if (acc > 0) return (long)acc;
                continue;                            // only empty members (the BGZF end marker): look further
            }
            if (!need_more && p < clen) return -2;   // the next member does not fit into `cap`
            // refill the compressed staging buffer
            if (cpos > 0) {
                memmove(cbuf.data(), cbuf.data() + cpos, clen - cpos);
                clen -= cpos;
                cpos = 0;
            }
            // the next stretch of the file, its slices copied out of the page cache by all threads (one fread of 32 MiB
            // was a quarter of the whole pass on a 32-thread host)
            const size_t want = cbuf.size() - clen;
            const off_t at = ftello(fp);
            const int fd = fileno(fp);
            const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, want >> 20));
            std::vector<size_t> part(T, 0);
            auto rd = [&](int t) {
                const size_t a = want * (size_t)t / (size_t)T, b = want * (size_t)(t + 1) / (size_t)T;
                size_t done = 0;
                while (a + done < b) {
                    const ssize_t k = pread(fd, cbuf.data() + clen + a + done, b - a - done, at + (off_t)(a + done));
                    if (k <= 0) break;
                    done += (size_t)k;
                }
                part[t] = done;
            };
            if (T <= 1) {
                rd(0);
            } else {
                std::vector<std::thread> th;
                for (int t = 0; t < T; ++t) th.emplace_back(rd, t);
                for (auto &x : th) x.join();
            }
            size_t got = 0;                          // contiguous bytes from the start: a short slice is the end of the file
            for (int t = 0; t < T; ++t) {
                got += part[t];
                if (part[t] < want * (size_t)(t + 1) / (size_t)T - want * (size_t)t / (size_t)T) break;
            }
            fseeko(fp, at + (off_t)got, SEEK_SET);
            clen += got;
            if (got == 0) {
                eof = true;
                return clen == 0 ? 0 : -1;           // trailing bytes that are not a whole member
            }
        }
    }
    static constexpr size_t SEG_BATCH_MAX = 512u << 20;      // text inflated by one batch of stretches
    void use_segments(const char *path, std::vector<AccessPoint> points, int nthreads)
    {
        if (bgzf || points.size() < 2 || nthreads < 2) return;
        segmented = true;
        seg_path = path;
        segs = std::move(points);
        seg_next = 0;
        threads = nthreads;
    }
    size_t seg_len(size_t i) const { return (size_t)(segs[i + 1].out - segs[i].out); }
    // room the next read() needs at least / would like to have
    size_t min_room() const
    {
        if (bgzf) return 65536;
        if (segmented && seg_next + 1 < segs.size()) return std::max<size_t>(seg_len(seg_next), 1);
        return 1;
    }
    size_t want_room() const
    {
        if (!segmented) return min_room();
        size_t total = 0;
        for (size_t c = 0; c < (size_t)threads && seg_next + c + 1 < segs.size(); ++c) {
            if (c > 0 && total + seg_len(seg_next + c) > SEG_BATCH_MAX) break;
            total += seg_len(seg_next + c);
        }
        return std::max<size_t>(total, 1);
    }
    // -3: no whole stretch is left (the one after the last access point has no known end) -- the serial stream,
    // reopened at that point, takes over
    long read_segments(char *dst, size_t cap)
    {
        for (;;) {
            if (seg_next + 1 >= segs.size()) {
                segmented = false;
                const AccessPoint last = std::move(segs[seg_next]);
                segs.clear();
                return open(seg_path.c_str(), &last) ? -3 : -1;
            }
            size_t c = 0, total = 0;
            while (c < (size_t)threads && seg_next + c + 1 < segs.size()) {
                const size_t len = seg_len(seg_next + c);
                if (total + len > cap || (c > 0 && total + len > SEG_BATCH_MAX)) break;
                total += len;
                ++c;
            }
            if (c == 0) return -2;
            std::vector<size_t> ooff(c);
            for (size_t i = 0, acc = 0; i < c; ++i) ooff[i] = acc, acc += seg_len(seg_next + i);
            std::vector<char> ok(c, 1);
            auto work = [&](size_t i) {
                GzSource s;
                s.in_bytes = 1u << 20;
                const size_t len = seg_len(seg_next + i);
                size_t got = 0;
                if (!s.open(seg_path.c_str(), &segs[seg_next + i])) ok[i] = 0;
                while (ok[i] && got < len) {
                    const long k = s.read(dst + ooff[i] + got, len - got);
                    if (k <= 0) ok[i] = 0;       // the file ends before the index says it does
                    else got += (size_t)k;
                }
            };
            if (c == 1) {
                work(0);
            } else {
                std::vector<std::thread> th;
                for (size_t i = 0; i < c; ++i) th.emplace_back(work, i);
                for (auto &x : th) x.join();
            }
            for (char f : ok)
                if (!f) return -1;
            for (size_t i = 0; i < c; ++i) std::vector<unsigned char>().swap(segs[seg_next + i].window);   // done with it
            seg_next += c;
            if (total > 0) return (long)total;
        }
    }
    // up to `cap` bytes of output; 0 at the end of the file, -1 on a corrupt stream, -2: the next unit (BGZF block,
    // stretch between access points) does not fit into `cap`
    long read(char *dst, size_t cap)
    {
        if (segmented) {
            const long k = read_segments(dst, cap);
            if (k != -3) return k;
        }
        if (eof) return 0;
        if (bgzf) return read_bgzf(dst, cap);
        z.next_out = reinterpret_cast<unsigned char *>(dst);
        z.avail_out = (unsigned)std::min<size_t>(cap, 1u << 30);
        const unsigned want = z.avail_out;
        while (z.avail_out != 0) {
            if (!refill()) {
                eof = true;             // input exhausted (a truncated last member ends the data like gzread's EOF)
                break;
            }
            const int ret = inflate(&z, Z_NO_FLUSH);
            if (ret == Z_STREAM_END) {  // end of a gzip member: another one may follow (concatenated gzip, BGZF)
                if (raw) {              // raw inflate leaves the 8-byte CRC32 + ISIZE trailer in the input
                    for (int skip = 8; skip > 0;) {
                        if (!refill()) break;
                        const unsigned k = std::min<unsigned>((unsigned)skip, z.avail_in);
                        z.next_in += k;
                        z.avail_in -= k;
                        skip -= (int)k;
                    }
                    raw = false;
                }
                if (!refill()) {
                    eof = true;
                    break;
                }
                if (inflateReset2(&z, 15 + 32) != Z_OK) return -1;
            } else if (ret != Z_OK && ret != Z_BUF_ERROR) {
                return -1;
            }
        }
        return (long)(want - z.avail_out);
    }
};

}  // namespace

// Text buffer that grows without being zero-filled (std::vector's resize would touch every new page).
struct TextBuf {
    char *p = nullptr;
    size_t n = 0;
    ~TextBuf() { free(p); }
    TextBuf() = default;
    TextBuf(const TextBuf &) = delete;
    TextBuf &operator=(const TextBuf &) = delete;
    char *data() const { return p; }
    size_t size() const { return n; }
    bool resize(size_t want)
    {
        if (want <= n) return true;
        char *q = (char *)realloc(p, want);
        if (!q) return false;
        p = q;
        n = want;
        return true;
    }
};

struct wgs_reader {
    GzSource src;
    std::vector<std::string> samples;
    int gl_cols = 0;   // GL columns in the header (3 per individual)
    int n_inds = 0;
    TextBuf buf;
    size_t len = 0, pos = 0;
    size_t fill_cap = (size_t)-1;   // most bytes one fill() appends (BGZF, while a reader is opened: the header and the lines up to
                                    // the first row need kilobytes, and what is left in the buffer reaches the device ingest
                                    // as host-inflated text)
    bool eof = false;
    std::string chunk_sites;   // '\n'-joined site names of the last chunk
    int threads = 1;
    int64_t lines_read = 0;
    int64_t text_chunks = 0;           // chunks the last text hand-over produced
    struct CompPipe *cpipe = nullptr;  // the compressed hand-over (reader_text.h: CompChunk), when started
    struct TextPipe *pipe = nullptr;   // the text hand-over to the device tokeniser (reader_text.h), when started
};

static bool fill(wgs_reader *r)
{
    // keep the unconsumed tail, append more inflated bytes
    if (r->pos > 0) {
        memmove(r->buf.data(), r->buf.data() + r->pos, r->len - r->pos);
        r->len -= r->pos;
        r->pos = 0;
    }
    // a single line longer than the buffer; a whole 64 KiB BGZF block, or one batch of stretches between access
    // points, must fit behind the tail
    const size_t want = r->src.want_room();
    if (r->buf.size() - r->len < want && !r->buf.resize(r->src.segmented ? r->len + want : r->buf.size() * 2)) return false;
    const size_t room = r->buf.size() - r->len;
    const size_t limit = r->len + std::min(room, std::max(r->fill_cap, want));
    while (!r->eof && r->len < limit) {
        const bool batch = r->src.segmented;
        const long got = r->src.read(r->buf.data() + r->len, limit - r->len);
        if (got == -2) break;                                 // the next block does not fit any more; enough for now
        if (got < 0) return false;
        if (got == 0) {
            r->eof = true;
            break;
        }
        r->len += (size_t)got;
        if (batch) break;                                     // one parallel batch per call (a second would be a partial one)
    }
    return true;
}

namespace {
// Parse one data line into out[0 .. 2*n_inds); returns false when the line is short.
static bool parse_line(const wgs_reader *r, const Line &ln, float *out, std::string *site)
{
    const char *p = ln.begin, *e = ln.end;
    auto next = [&](const char *&tb, const char *&te) {
        while (p < e && is_delim(*p)) ++p;
        if (p >= e) return false;
        tb = p;
        while (p < e && !is_delim(*p)) ++p;
        te = p;
        return true;
    };
    const char *tb, *te;
    if (!next(tb, te)) return false;
    site->assign(tb, te);                                  // reader_cy.pyx:56-57
    if (!next(tb, te) || !next(tb, te)) return false;      // allele1, allele2 (reader_cy.pyx:59-60)
    // a header whose GL column count is not a multiple of 3 leaves a partial individual: the reference
    // parses those columns but only keeps the first 2 * (n // 3) values of each row (reader_cy.pyx:48-49,
    // 71-75), so they are not read at all here -- every row is exactly 2 * n_inds floats
    for (int i = 0; i < r->n_inds; ++i) {
        // fast path: "\td.dddddd\td.dddddd\t<anything>" -- one delimiter, two fixed-format values, the third skipped
        if (e - p >= 19 && is_delim(p[0])) {
            double a, b;
            if (parse_f6(p + 1, e, &a) && parse_f6(p + 10, e, &b)) {
                const char *q = p + 19;
                while (q < e && !is_delim(*q)) ++q;
                if (q > p + 19) {                                         // the third value is there
                    *out++ = (float)a;
                    *out++ = (float)b;
                    p = q;
                    continue;
                }
            }
        }
        for (int j = 0; j < 3; ++j) {
            if (!next(tb, te)) return false;
            if (j < 2) *out++ = (float)parse_double(tb, te);              // reader_cy.pyx:62-66
        }
    }
    return true;
}

// Header line (reader_cy.pyx:35-49): tokens after the first three are GL columns, every third names a sample.
void parse_header(const char *p, const char *hend, std::vector<std::string> &samples, int &gl_cols)
{
    int tok = 0;
    samples.clear();
    while (p < hend) {
        while (p < hend && is_delim(*p)) ++p;
        if (p >= hend) break;
        const char *tb = p;
        while (p < hend && !is_delim(*p)) ++p;
        ++tok;
        if (tok > 3 && (tok - 3) % 3 == 1) samples.emplace_back(tb, p);
    }
    gl_cols = tok > 3 ? tok - 3 : 0;
}

}  // namespace

extern "C" {

int wgs_reader_open(const char *path, int threads, wgs_reader **out)
{
    if (!path || !out) {
        wgs_set_error("null argument");
        return 2;
    }
    wgs_reader *r = new wgs_reader();
    if (!r->src.open(path, nullptr)) {
        wgs_set_error("cannot open Beagle file %s", path);
        delete r;
        return 2;
    }
    r->threads = threads > 0 ? threads : 1;
    r->src.try_bgzf(r->threads);
    if (!r->buf.resize(64u << 20)) {
        wgs_set_error("out of memory");
        delete r;
        return 1;
    }
    if (r->src.bgzf) r->fill_cap = 1u << 20;
    if (!fill(r)) {
        wgs_set_error("read error in %s", path);
        delete r;
        return 1;
    }
    const char *nl = (const char *)memchr(r->buf.data(), '\n', r->len);
    while (!nl && !r->eof) {
        if (!fill(r)) break;
        nl = (const char *)memchr(r->buf.data(), '\n', r->len);
    }
    const char *hend = nl ? nl : r->buf.data() + r->len;
    parse_header(r->buf.data(), hend, r->samples, r->gl_cols);
    r->n_inds = r->gl_cols / 3;
    r->pos = nl ? (size_t)(nl - r->buf.data()) + 1 : r->len;
    r->fill_cap = (size_t)-1;
    *out = r;
    return 0;
}

void wgs_reader_close(wgs_reader *r)
{
    if (!r) return;
    reader_comp_stop(r);
    reader_text_stop(r);
    delete r;
}

}  // extern "C"

namespace {

// Line bookkeeping of the index pass over consecutive pieces of output: non-blank lines completed, whether
// the line in progress has content, optional capture of every data line's first token (the site name).
struct LineScan {
    int64_t lines = 0;          // non-blank lines completed (the header is line 0)
    bool content = false, at_line_start = true, in_name = false, name_done = false;
    std::string header, *names = nullptr;
    bool header_done = false;

    void feed(const unsigned char *p, size_t n)
    {
        const unsigned char *e = p + n;
        while (p < e) {
            if (!header_done) {                      // keep the header line verbatim
                const unsigned char *nl = (const unsigned char *)memchr(p, '\n', (size_t)(e - p));
                header.append((const char *)p, (size_t)((nl ? nl : e) - p));
                if (!nl) return;
                header_done = true;
                lines += 1;                          // the header counts as a line even when empty
                content = false;
                at_line_start = true;
                p = nl + 1;
                continue;
            }
            if (!content) {                          // look for the first non-delimiter of the line
                while (p < e && *p != '\n' && is_delim((char)*p)) ++p, at_line_start = false;
                if (p == e) return;
                if (*p == '\n') {                    // blank line: not counted
                    at_line_start = true;
                    ++p;
                    continue;
                }
                content = true;
                at_line_start = false;
                in_name = names != nullptr;
                name_done = false;
            }
            if (in_name) {
                const unsigned char *t = p;
                while (t < e && !is_delim((char)*t)) ++t;
                names->append((const char *)p, (size_t)(t - p));
                p = t;
                if (p == e) return;
                names->push_back('\n');
                in_name = false;
            }
            const unsigned char *nl = (const unsigned char *)memchr(p, '\n', (size_t)(e - p));
            if (!nl) return;
            lines += 1;
            content = false;
            at_line_start = true;
            p = nl + 1;
        }
    }
    void finish()
    {
        if (!header_done) {
            header_done = true;
            lines += 1;
        } else if (content) {
            if (in_name) names->push_back('\n');
            lines += 1;                              // last line without a newline
        }
        content = false;
    }
};

struct BeagleIndex {
    uint64_t file_size = 0, mtime = 0;
    int64_t sites = 0;
    int32_t gl_cols = 0;
    std::vector<std::string> samples;
    std::vector<AccessPoint> points;
};

// size and modification time in NANOSECONDS: a same-size rewrite within one second must not reuse a stale index
bool file_identity(const char *path, uint64_t &size, uint64_t &mtime)
{
    struct stat st;
    if (stat(path, &st) != 0) return false;
    size = (uint64_t)st.st_size;
    mtime = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
    return true;
}

// Cache files (index, site names) are written under a fresh name that must not exist (no symlink is followed, nothing
// of another user is overwritten) and renamed into place; they are read only if they are regular files of this user.
FILE *create_private(const std::string &final_path, std::string &tmp)
{
    for (int attempt = 0; attempt < 16; ++attempt) {
        tmp = final_path + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(attempt);
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
        if (fd >= 0) {
            FILE *f = fdopen(fd, "wb");
            if (!f) close(fd);
            return f;
        }
        if (errno != EEXIST) return nullptr;
    }
    return nullptr;
}

FILE *open_private(const char *path)
{
    const int fd = open(path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid()) {
        close(fd);
        return nullptr;
    }
    FILE *f = fdopen(fd, "rb");
    if (!f) close(fd);
    return f;
}

// One inflate pass: count the sites, read the header, record access points every `span` output bytes.
int scan_file(const char *path, int64_t span, BeagleIndex &idx, std::string *names)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        wgs_set_error("cannot open Beagle file %s", path);
        return 2;
    }
    file_identity(path, idx.file_size, idx.mtime);
    // Output goes into 1 MiB laps of a buffer whose first GZ_WIN bytes always hold the 32 KiB that precede the lap
    // (copied there when a lap starts), so the dictionary of an access point is the GZ_WIN bytes before next_out,
    // contiguous -- and inflate is called per megabyte or deflate block, not per 32 KiB.
    constexpr size_t LAP = 1u << 20;
    std::vector<unsigned char> in(4u << 20), win(GZ_WIN + LAP);
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 32) != Z_OK) {
        fclose(fp);
        wgs_set_error("zlib initialisation failed");
        return 1;
    }
    LineScan ls;
    ls.names = names;
    uint64_t totin = 0, totout = 0, last = 0;
    bool member_start = true, bad = false;
    z.avail_out = 0;
    auto add_point = [&](bool at_member) {
        AccessPoint ap;
        ap.in = totin;
        ap.out = totout;
        ap.lines_before = ls.lines;
        ap.content = ls.content;
        ap.at_line_start = ls.at_line_start;
        ap.member_start = at_member;
        ap.bits = at_member ? 0 : (uint8_t)(z.data_type & 7);
        if (!at_member) ap.window.assign(z.next_out - GZ_WIN, z.next_out);       // the GZ_WIN bytes before `out`
        idx.points.push_back(std::move(ap));
        last = totout;
    };
    for (;;) {
        if (z.avail_in == 0) {
            const size_t got = fread(in.data(), 1, in.size(), fp);
            if (got == 0) break;
            z.next_in = in.data();
            z.avail_in = (unsigned)got;
        }
        if (member_start && span > 0 && totout - last >= (uint64_t)span && totout > 0) add_point(true);
        member_start = false;
        if (z.avail_out == 0) {                      // new lap: keep the last GZ_WIN bytes in front of it
            if (totout > 0) memmove(win.data(), z.next_out - GZ_WIN, GZ_WIN);
            z.avail_out = (unsigned)LAP;
            z.next_out = win.data() + GZ_WIN;
        }
        const unsigned char *produced_at = z.next_out;
        const unsigned in0 = z.avail_in, out0 = z.avail_out;
        const int ret = inflate(&z, Z_BLOCK);
        totin += in0 - z.avail_in;
        totout += out0 - z.avail_out;
        ls.feed(produced_at, out0 - z.avail_out);
        if (ret == Z_STREAM_END) {                   // next gzip member, if any
            if (z.avail_in == 0) {
                const size_t got = fread(in.data(), 1, in.size(), fp);
                z.next_in = in.data();
                z.avail_in = (unsigned)got;
                if (got == 0) break;
            }
            if (inflateReset2(&z, 15 + 32) != Z_OK) {
                bad = true;
                break;
            }
            member_start = true;
            continue;
        }
        if (ret != Z_OK && ret != Z_BUF_ERROR) {
            bad = true;
            break;
        }
        // at the end of a deflate block that is not the last of its member
        if (span > 0 && (z.data_type & 128) && !(z.data_type & 64) && totout - last >= (uint64_t)span && totout >= GZ_WIN)
            add_point(false);
    }
    inflateEnd(&z);
    fclose(fp);
    if (bad) {
        wgs_set_error("read error in %s (corrupt gzip stream)", path);
        return 1;
    }
    ls.finish();
    idx.sites = ls.lines > 0 ? ls.lines - 1 : 0;     // minus the header line
    parse_header(ls.header.data(), ls.header.data() + ls.header.size(), idx.samples, idx.gl_cols);
    return 0;
}

// ---- the same pass for BGZF files, in parallel -----------------------------------------------------------------
// The block table (offsets, compressed and uncompressed sizes) comes from hopping over the member headers and
// trailers -- nothing is inflated for it.  Worker threads then inflate disjoint ranges of blocks and summarise each:
// where its first newline is, whether there is text before it, how many non-blank lines end after it, and what is
// left open at its end.  A serial pass over the summaries (a few words per 64 KiB of input) turns them into global
// line numbers; access points are block starts (no dictionaries).
struct BlockLines {
    uint8_t has_nl = 0;             // the block contains a newline
    uint8_t content_before = 0;     // non-delimiter text before the first newline (or anywhere, if there is none)
    uint8_t content_after = 0;      // non-delimiter text after the last newline
    uint8_t last_is_nl = 0;         // the block ends with a newline
    int32_t lines_after_first = 0;  // non-blank lines that start AND end inside the block
};

bool bgzf_block_table(const char *path, std::vector<BgzfBlock> &blocks)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    const int fd = fileno(fp);
    uint64_t off = 0;
    unsigned char head[512], tail[4];
    bool ok = true;
    for (;;) {
        const ssize_t got = pread(fd, head, sizeof head, (off_t)off);
        if (got == 0) break;
        uint32_t hdr = 0;
        const long sz = got > 0 ? bgzf_member_size(head, (size_t)got, &hdr) : 0;
        if (sz <= 0 || pread(fd, tail, 4, (off_t)(off + (uint64_t)sz - 4)) != 4) {
            ok = false;
            break;
        }
        BgzfBlock b;
        b.off = off;
        b.csize = (uint32_t)sz;
        b.hdr = hdr;
        b.isize = tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
        if (b.isize > 65536 || b.csize < hdr + 8) {
            ok = false;
            break;
        }
        blocks.push_back(b);
        off += (uint64_t)sz;
    }
    fclose(fp);
    return ok && !blocks.empty();
}

void summarise_block(const unsigned char *p, size_t n, BlockLines &bl)
{
    const unsigned char *e = p + n;
    const unsigned char *nl = (const unsigned char *)memchr(p, '\n', n);
    const unsigned char *seg_end = nl ? nl : e;
    for (const unsigned char *t = p; t < seg_end && !bl.content_before; ++t) bl.content_before = !is_delim((char)*t);
    if (!nl) return;
    bl.has_nl = 1;
    const unsigned char *q = nl + 1;
    for (;;) {
        const unsigned char *nx = (const unsigned char *)memchr(q, '\n', (size_t)(e - q));
        const unsigned char *end = nx ? nx : e;
        bool content = false;
        for (const unsigned char *t = q; t < end && !content; ++t) content = !is_delim((char)*t);
        if (!nx) {
            bl.content_after = content;
            break;
        }
        bl.lines_after_first += content;
        q = nx + 1;
    }
    bl.last_is_nl = n > 0 && p[n - 1] == '\n';
}

// ---- the BGZF pass in parts: threads of one process, and processes of one node -------------------------------------
// Finding the blocks by hopping from header to header is serial (41 M hops for 2.7 TB of text), and so was the pass of
// round 2 across ranks (the first rank inflated the whole file).  A part -- any byte range of the file -- can find its own
// first block instead: BGZF members start with a 16-byte signature (gzip magic, FEXTRA, XLEN = 6, 'B' 'C' 2 0); the first
// position at or after the range's start where a header parses AND three further hops land on headers (or the end of the
// file) is taken as a block boundary.  That is a heuristic only for SPEED: every part reports where it started and where
// the block after its last one starts, and the merge accepts the parts only if they chain exactly (part 0 starts at byte
// 0, so by induction every part hopped the true chain); otherwise the classic serial hop takes over.
struct PartData {
    uint64_t first = 0, next = 0;          // file offset of the part's first block / of the first block after its last
    std::vector<BgzfBlock> blocks;
    std::vector<BlockLines> sum;
    std::string header;                    // part 0: the header line
    bool header_done = false;
};

// Parses the member at file offset `off`: 1 = a BGZF block, 0 = end of file exactly at off, -1 = anything else.
int bgzf_block_at(int fd, uint64_t file_size, uint64_t off, BgzfBlock &b)
{
    if (off == file_size) return 0;
    unsigned char head[64], tail[4];
    const ssize_t got = pread(fd, head, sizeof head, (off_t)off);
    uint32_t hdr = 0;
    const long sz = got > 0 ? bgzf_member_size(head, (size_t)got, &hdr) : 0;
    if (sz <= 0 || off + (uint64_t)sz > file_size || pread(fd, tail, 4, (off_t)(off + (uint64_t)sz - 4)) != 4) return -1;
    b.off = off;
    b.csize = (uint32_t)sz;
    b.hdr = hdr;
    b.isize = tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
    return b.isize <= 65536 && b.csize >= hdr + 8 ? 1 : -1;
}

// First block boundary at or after `from` (from > 0): see above.  Returns file_size when the rest holds no block start.
uint64_t bgzf_resync(int fd, uint64_t file_size, uint64_t from)
{
    std::vector<unsigned char> win(1u << 20);
    for (uint64_t base = from; base < file_size;) {
        const ssize_t got = pread(fd, win.data(), win.size(), (off_t)base);
        if (got < 16) return file_size;
        for (ssize_t i = 0; i + 16 <= got; ++i) {
            const unsigned char *q = win.data() + i;
            if (q[0] != 31 || q[1] != 139 || q[2] != 8 || q[3] != 4) continue;
            uint64_t at = base + (uint64_t)i;
            bool chain = true;
            for (int hop = 0; hop < 4 && chain; ++hop) {
                BgzfBlock b;
                const int k = bgzf_block_at(fd, file_size, at, b);
                if (k == 0) break;                            // the chain ends with the file: fine
                chain = k > 0;
                at += b.csize;
            }
            if (chain) return base + (uint64_t)i;
        }
        base += (uint64_t)got - 15;                           // a signature may straddle the window's end
    }
    return file_size;
}

// The blocks that START in [lo, hi), inflated and summarised: one thread's share.
bool bgzf_part_thread(const char *path, uint64_t file_size, uint64_t lo, uint64_t hi, bool want_header, PartData &out)
{
    FILE *fp = fopen(path, "rb");
    BlockInflater inf;
    if (!fp || !inf.init()) {
        if (fp) fclose(fp);
        return false;
    }
    const int fd = fileno(fp);
    uint64_t at = lo == 0 ? 0 : bgzf_resync(fd, file_size, lo);
    out.first = at;
    std::vector<unsigned char> cb(65536 + 4096), ob(65536);
    bool ok = true;
    while (at < hi && at < file_size) {
        BgzfBlock b;
        if (bgzf_block_at(fd, file_size, at, b) <= 0 || pread(fd, cb.data(), b.csize, (off_t)at) != (ssize_t)b.csize) {
            ok = false;
            break;
        }
        BgzfBlock rel = b;
        rel.off = 0;
        if (!inf.run(cb.data(), rel, ob.data())) {
            ok = false;
            break;
        }
        BlockLines bl;
        summarise_block(ob.data(), b.isize, bl);
        if (want_header && !out.header_done && b.isize > 0) {
            const unsigned char *nl = (const unsigned char *)memchr(ob.data(), '\n', b.isize);
            out.header.append((const char *)ob.data(), nl ? (size_t)(nl - ob.data()) : b.isize);
            out.header_done = nl != nullptr;
        }
        out.blocks.push_back(b);
        out.sum.push_back(bl);
        at += b.csize;
    }
    out.next = at;
    fclose(fp);
    return ok;
}

// Part `part` of `nparts` (equal byte ranges of the file) on `threads` threads.  0 = ok, -1 = not BGZF / the sub-ranges
// did not chain (the caller falls back to the serial hop), 1 = read error.
int bgzf_collect_part(const char *path, int part, int nparts, int threads, PartData &out)
{
    uint64_t file_size = 0, mtime = 0;
    if (!file_identity(path, file_size, mtime)) return 1;
    {
        FILE *fp = fopen(path, "rb");
        if (!fp) return 1;
        BgzfBlock b;
        const int k = bgzf_block_at(fileno(fp), file_size, 0, b);
        fclose(fp);
        if (k <= 0) return -1;
    }
    const uint64_t lo = file_size * (uint64_t)part / (uint64_t)nparts, hi = file_size * (uint64_t)(part + 1) / (uint64_t)nparts;
    const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(threads > 0 ? threads : 1), (hi - lo) >> 20));
    std::vector<PartData> sub(T);
    std::vector<char> ok(T, 1);
    run_threads(T, [&](int t) {
        const uint64_t a = lo + (hi - lo) * (uint64_t)t / (uint64_t)T, b = lo + (hi - lo) * (uint64_t)(t + 1) / (uint64_t)T;
        ok[t] = bgzf_part_thread(path, file_size, a, b, part == 0 && t == 0, sub[t]);
    });
    for (char c : ok)
        if (!c) return -1;
    out = PartData();
    out.first = sub[0].first;
    for (int t = 0; t < T; ++t) {
        if (t > 0 && sub[t].first != sub[t - 1].next) return -1;          // a sub-range started off the chain
        out.blocks.insert(out.blocks.end(), sub[t].blocks.begin(), sub[t].blocks.end());
        out.sum.insert(out.sum.end(), sub[t].sum.begin(), sub[t].sum.end());
    }
    out.next = sub[T - 1].next;
    out.header = sub[0].header;
    out.header_done = sub[0].header_done;
    // a header line longer than the first thread's share: finish it serially
    if (part == 0 && !out.header_done) {
        out.header.clear();
        FILE *fp = fopen(path, "rb");
        BlockInflater inf;
        if (!fp || !inf.init()) {
            if (fp) fclose(fp);
            return 1;
        }
        std::vector<unsigned char> cb(65536 + 4096), ob(65536);
        for (const BgzfBlock &b : out.blocks) {
            if (pread(fileno(fp), cb.data(), b.csize, (off_t)b.off) != (ssize_t)b.csize) break;
            BgzfBlock rel = b;
            rel.off = 0;
            if (!inf.run(cb.data(), rel, ob.data())) break;
            const unsigned char *nl = (const unsigned char *)memchr(ob.data(), '\n', b.isize);
            out.header.append((const char *)ob.data(), nl ? (size_t)(nl - ob.data()) : b.isize);
            if (nl) break;
        }
        fclose(fp);
    }
    return 0;
}

// The serial pass over the block summaries of all parts, in file order: line numbers at every block start, access points.
// -1 when the parts do not chain.
int bgzf_merge_parts(const char *path, const std::vector<PartData> &parts, int64_t span, BeagleIndex &idx)
{
    uint64_t file_size = 0;
    if (!file_identity(path, idx.file_size, idx.mtime)) return 1;
    file_size = idx.file_size;
    uint64_t expect = 0;
    for (const PartData &p : parts) {
        if (p.first != expect) return -1;
        expect = p.next;
    }
    if (expect != file_size) return -1;
    int64_t lines = 0;              // non-blank lines completed
    bool content = false, at_line_start = true, header_done = false;
    uint64_t out = 0, last = 0;
    for (const PartData &p : parts) {
        for (size_t i = 0; i < p.blocks.size(); ++i) {
            const BgzfBlock &blk = p.blocks[i];
            const BlockLines &bl = p.sum[i];
            if (span > 0 && out > 0 && out - last >= (uint64_t)span && blk.isize > 0 && header_done) {
                AccessPoint ap;
                ap.in = blk.off;
                ap.out = out;
                ap.lines_before = lines;
                ap.content = content;
                ap.at_line_start = at_line_start;
                ap.member_start = 1;
                idx.points.push_back(std::move(ap));
                last = out;
            }
            if (blk.isize == 0) continue;
            if (bl.has_nl) {
                // the header counts as a line even when it is empty (LineScan); every other line only when non-blank
                lines += !header_done ? 1 : (content || bl.content_before);
                header_done = true;
                lines += bl.lines_after_first;
                content = bl.content_after;
                at_line_start = bl.last_is_nl;
            } else {
                content = content || bl.content_before;
                at_line_start = false;
            }
            out += blk.isize;
        }
    }
    if (!header_done) lines += 1;
    else if (content) lines += 1;   // last line without a newline
    idx.sites = lines > 0 ? lines - 1 : 0;
    const std::string &header = parts.empty() ? std::string() : parts[0].header;
    parse_header(header.data(), header.data() + header.size(), idx.samples, idx.gl_cols);
    return 0;
}

// Returns 0 on success, -1 when the file is not (entirely) BGZF -- the caller then takes the serial pass.
int scan_file_bgzf(const char *path, int64_t span, int threads, BeagleIndex &idx)
{
    std::vector<PartData> parts(1);
    int rc = bgzf_collect_part(path, 0, 1, threads, parts[0]);
    if (rc == 0) rc = bgzf_merge_parts(path, parts, span, idx);
    if (rc == -1) {
        // the sub-ranges did not chain (or this is not BGZF at all): the classic serial hop decides
        std::vector<BgzfBlock> blocks;
        if (!bgzf_block_table(path, blocks)) return -1;
        idx = BeagleIndex();
        parts.assign(1, PartData());
        uint64_t file_size = 0, mtime = 0;
        file_identity(path, file_size, mtime);
        parts[0].first = 0;
        if (!bgzf_part_thread(path, file_size, 0, file_size, true, parts[0])) {
            wgs_set_error("read error in %s (corrupt BGZF block)", path);
            return 1;
        }
        rc = bgzf_merge_parts(path, parts, span, idx);
    }
    if (rc == 1) wgs_set_error("read error in %s (corrupt BGZF block)", path);
    return rc;
}

template <typename T>
void put(FILE *f, const T &v) { fwrite(&v, sizeof v, 1, f); }
template <typename T>
bool get(FILE *f, T &v) { return fread(&v, sizeof v, 1, f) == 1; }

const char kIndexMagic[8] = {'W', 'G', 'S', 'I', 'D', 'X', '3', 0};     // 3: modification time in nanoseconds

bool save_index(const char *path, const BeagleIndex &idx)
{
    std::string tmp;
    FILE *f = create_private(path, tmp);
    if (!f) return false;
    fwrite(kIndexMagic, 1, 8, f);
    put(f, idx.file_size);
    put(f, idx.mtime);
    put(f, idx.sites);
    put(f, idx.gl_cols);
    const int32_t ns = (int32_t)idx.samples.size(), np = (int32_t)idx.points.size();
    put(f, ns);
    for (const auto &s : idx.samples) {
        const int32_t l = (int32_t)s.size();
        put(f, l);
        fwrite(s.data(), 1, s.size(), f);
    }
    put(f, np);
    for (const auto &a : idx.points) {
        put(f, a.in);
        put(f, a.out);
        put(f, a.lines_before);
        put(f, a.bits);
        put(f, a.content);
        put(f, a.at_line_start);
        put(f, a.member_start);
        if (!a.member_start) {                                // the 32 KiB dictionary, deflated (text: ~4x smaller)
            std::vector<unsigned char> z(compressBound(GZ_WIN));
            uLongf zn = (uLongf)z.size();
            if (compress2(z.data(), &zn, a.window.data(), GZ_WIN, 1) != Z_OK) {
                fclose(f);
                return false;
            }
            const uint32_t n32 = (uint32_t)zn;
            put(f, n32);
            fwrite(z.data(), 1, zn, f);
        }
    }
    const bool ok = !ferror(f) && fflush(f) == 0;
    fclose(f);
    if (ok && rename(tmp.c_str(), path) == 0) return true;     // atomic: readers never see a partial index
    unlink(tmp.c_str());
    return false;
}

bool load_index(const char *path, BeagleIndex &idx)
{
    FILE *f = open_private(path);
    if (!f) return false;
    char magic[8];
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, kIndexMagic, 8) == 0;
    int32_t ns = 0, np = 0;
    ok = ok && get(f, idx.file_size) && get(f, idx.mtime) && get(f, idx.sites) && get(f, idx.gl_cols) && get(f, ns) && ns >= 0;
    for (int i = 0; ok && i < ns; ++i) {
        int32_t l = 0;
        ok = get(f, l) && l >= 0 && l < (1 << 20);
        if (!ok) break;
        std::string s((size_t)l, '\0');
        ok = fread(&s[0], 1, (size_t)l, f) == (size_t)l;
        idx.samples.push_back(std::move(s));
    }
    ok = ok && get(f, np) && np >= 0;
    for (int i = 0; ok && i < np; ++i) {
        AccessPoint a;
        ok = get(f, a.in) && get(f, a.out) && get(f, a.lines_before) && get(f, a.bits) && get(f, a.content) &&
             get(f, a.at_line_start) && get(f, a.member_start);
        if (ok && !a.member_start) {
            uint32_t zn = 0;
            ok = get(f, zn) && zn > 0 && zn <= compressBound(GZ_WIN);
            std::vector<unsigned char> z(ok ? zn : 0);
            ok = ok && fread(z.data(), 1, zn, f) == zn;
            a.window.resize(GZ_WIN);
            uLongf out = GZ_WIN;
            ok = ok && uncompress(a.window.data(), &out, z.data(), zn) == Z_OK && out == GZ_WIN;
        }
        if (ok) idx.points.push_back(std::move(a));
    }
    fclose(f);
    return ok;
}

}  // namespace

extern "C" {

/* ONE inflate pass over a gzipped Beagle file: counts its sites, and (index_path != NULL) writes an index --
 * header fields plus at most max_points access points, about `span_bytes` of text apart -- from which
 * wgs_reader_open_indexed starts at any row without inflating what precedes it; names_path != NULL also
 * receives every site name, '\n'-terminated (the names-only pass of the downsampled-LOO site masks). */
int wgs_reader_build_index(const char *path, const char *index_path, const char *names_path, int64_t span_bytes, int32_t max_points,
                           int64_t *sites)
{
    if (!path || !sites) {
        wgs_set_error("null argument");
        return 2;
    }
    BeagleIndex idx;
    std::string names;
    const int64_t span = index_path ? std::max<int64_t>(span_bytes, (int64_t)GZ_WIN) : 0;
    // BGZF input (what ANGSD writes): blocks are independent -> all host threads.  Site names then come from a second
    // pass whose blocks are inflated in parallel as well (GzSource::read_bgzf) and only scanned serially for the first
    // token of every line -- a memchr per line.  Anything else: one serial inflate pass that does everything.
    const int threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    int rc = scan_file_bgzf(path, span, threads, idx);
    if (rc == 0 && names_path) {
        GzSource src;
        if (!src.open(path, nullptr)) {
            wgs_set_error("cannot open Beagle file %s", path);
            return 2;
        }
        src.try_bgzf(threads);
        LineScan ls;
        ls.names = &names;
        std::vector<char> buf(64u << 20);
        for (;;) {
            const long got = src.read(buf.data(), buf.size());
            if (got < 0) {
                wgs_set_error("read error in %s (corrupt gzip stream)", path);
                return 1;
            }
            if (got == 0) break;
            ls.feed(reinterpret_cast<const unsigned char *>(buf.data()), (size_t)got);
        }
        ls.finish();
        if ((ls.lines > 0 ? ls.lines - 1 : 0) != idx.sites) {
            wgs_set_error("Beagle file %s: the two passes disagree about the number of sites", path);
            return 1;
        }
    }
    if (rc < 0) {
        idx = BeagleIndex();
        rc = scan_file(path, span, idx, names_path ? &names : nullptr);
    }
    if (rc) return rc;
    *sites = idx.sites;
    // the site names first, then the index: an index in place implies that the names written with it are complete
    // (both appear by rename; a run that is killed in between leaves a names file without an index -- rebuilt)
    if (names_path) {
        std::string tmp;
        FILE *f = create_private(names_path, tmp);
        const bool ok = f && fwrite(names.data(), 1, names.size(), f) == names.size() && fflush(f) == 0;
        if (f) fclose(f);
        if (!ok || rename(tmp.c_str(), names_path) != 0) {
            if (f) unlink(tmp.c_str());
            wgs_set_error("cannot write the site names to %s", names_path);
            return 1;
        }
    }
    if (index_path) {
        if (max_points > 0 && (int64_t)idx.points.size() > max_points) {       // thin out evenly
            std::vector<AccessPoint> keep;
            const double step = (double)idx.points.size() / max_points;
            for (int i = 0; i < max_points; ++i) keep.push_back(std::move(idx.points[(size_t)(i * step)]));
            idx.points.swap(keep);
        }
        if (!save_index(index_path, idx)) {
            wgs_set_error("cannot write the Beagle index %s", index_path);
            return 1;
        }
    }
    return 0;
}

}   // extern "C"

namespace {
const char kPartMagic[8] = {'W', 'G', 'S', 'P', 'R', 'T', '1', 0};

bool save_part(const char *path_out, const PartData &p, uint64_t file_size, uint64_t mtime)
{
    std::string tmp;
    FILE *f = create_private(path_out, tmp);
    if (!f) return false;
    fwrite(kPartMagic, 1, 8, f);
    put(f, file_size);
    put(f, mtime);
    put(f, p.first);
    put(f, p.next);
    const uint64_t nb = p.blocks.size(), hl = p.header.size();
    put(f, nb);
    put(f, hl);
    fwrite(p.header.data(), 1, p.header.size(), f);
    if (nb) {
        fwrite(p.blocks.data(), sizeof(BgzfBlock), nb, f);
        fwrite(p.sum.data(), sizeof(BlockLines), nb, f);
    }
    const bool ok = !ferror(f) && fflush(f) == 0;
    fclose(f);
    if (ok && rename(tmp.c_str(), path_out) == 0) return true;
    unlink(tmp.c_str());
    return false;
}

bool load_part(const char *path_in, PartData &p, uint64_t file_size, uint64_t mtime)
{
    FILE *f = open_private(path_in);
    if (!f) return false;
    char magic[8];
    uint64_t fs = 0, mt = 0, nb = 0, hl = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, kPartMagic, 8) == 0 && get(f, fs) && get(f, mt) && get(f, p.first) && get(f, p.next) &&
              get(f, nb) && get(f, hl) && fs == file_size && mt == mtime && hl < (1u << 30) && nb < (1ull << 40);
    if (ok) {
        // the counts must fit the part file itself BEFORE anything is sized by them: a damaged part file would otherwise ask for
        // terabytes (std::bad_alloc through an extern "C" entry point)
        struct stat sb;
        const uint64_t fixed = 8 + 6 * sizeof(uint64_t);
        ok = fstat(fileno(f), &sb) == 0 && (uint64_t)sb.st_size >= fixed && hl <= (uint64_t)sb.st_size - fixed &&
             nb <= ((uint64_t)sb.st_size - fixed - hl) / (sizeof(BgzfBlock) + sizeof(BlockLines));
    }
    if (ok) {
        p.header.resize((size_t)hl);
        ok = fread(&p.header[0], 1, (size_t)hl, f) == (size_t)hl || hl == 0;
        p.blocks.resize((size_t)nb);
        p.sum.resize((size_t)nb);
        if (ok && nb) ok = fread(p.blocks.data(), sizeof(BgzfBlock), (size_t)nb, f) == (size_t)nb && fread(p.sum.data(), sizeof(BlockLines), (size_t)nb, f) == (size_t)nb;
    }
    fclose(f);
    return ok;
}
}  // namespace

extern "C" {

/* The index pass of a BGZF file split over the ranks of a node: rank `part` of `nparts` inflates and summarises the blocks
 * of its byte range (on `threads` threads) into part_path; wgs_reader_index_merge (one rank, after a barrier) chains the
 * parts into the index.  rc 3 = this file cannot be done in parts (not BGZF, or a range did not find the block chain): one
 * rank then calls wgs_reader_build_index as before. */
int wgs_reader_index_part(const char *path, const char *part_path, int part, int nparts, int threads)
{
    if (!path || !part_path || part < 0 || nparts < 1 || part >= nparts) {
        wgs_set_error("bad argument");
        return 2;
    }
    PartData p;
    const int rc = bgzf_collect_part(path, part, nparts, threads, p);
    if (rc == -1) {
        wgs_set_error("%s cannot be indexed in parts", path);
        return 3;
    }
    if (rc) {
        wgs_set_error("read error in %s (corrupt BGZF block)", path);
        return 1;
    }
    uint64_t size = 0, mtime = 0;
    file_identity(path, size, mtime);
    if (!save_part(part_path, p, size, mtime)) {
        wgs_set_error("cannot write %s", part_path);
        return 1;
    }
    return 0;
}

int wgs_reader_index_merge(const char *path, const char *index_path, const char *parts_prefix, int nparts, int64_t span_bytes,
                           int32_t max_points, int64_t *sites)
{
    if (!path || !index_path || !parts_prefix || nparts < 1 || !sites) {
        wgs_set_error("bad argument");
        return 2;
    }
    uint64_t size = 0, mtime = 0;
    if (!file_identity(path, size, mtime)) {
        wgs_set_error("cannot open Beagle file %s", path);
        return 2;
    }
    // the part files are this call's to remove, however it ends (the caller falls back to the one-rank pass on rc 3)
    struct PartFiles {
        std::string prefix;
        int n;
        ~PartFiles() { for (int k = 0; k < n; ++k) unlink((prefix + "." + std::to_string(k)).c_str()); }
    } cleanup{std::string(parts_prefix), nparts};
    std::vector<PartData> parts((size_t)nparts);
    for (int k = 0; k < nparts; ++k) {
        const std::string pp = std::string(parts_prefix) + "." + std::to_string(k);
        if (!load_part(pp.c_str(), parts[(size_t)k], size, mtime)) {
            wgs_set_error("%s is not a part of the index of %s", pp.c_str(), path);
            return 3;
        }
    }
    BeagleIndex idx;
    const int rc = bgzf_merge_parts(path, parts, std::max<int64_t>(span_bytes, (int64_t)GZ_WIN), idx);
    if (rc == -1) {
        wgs_set_error("the parts of the index of %s do not chain", path);
        return 3;
    }
    if (rc) {
        wgs_set_error("read error in %s", path);
        return 1;
    }
    if (max_points > 0 && (int64_t)idx.points.size() > max_points) {
        std::vector<AccessPoint> keep;
        const double step = (double)idx.points.size() / max_points;
        for (int i = 0; i < max_points; ++i) keep.push_back(std::move(idx.points[(size_t)(i * step)]));
        idx.points.swap(keep);
    }
    if (!save_index(index_path, idx)) {
        wgs_set_error("cannot write the Beagle index %s", index_path);
        return 1;
    }
    *sites = idx.sites;
    return 0;
}

/* Sites of the file an index was built for (after checking that it still describes `path`: size and mtime). */
int wgs_reader_index_sites(const char *path, const char *index_path, int64_t *sites)
{
    if (!path || !index_path || !sites) {
        wgs_set_error("null argument");
        return 2;
    }
    BeagleIndex idx;
    uint64_t size = 0, mtime = 0;
    if (!load_index(index_path, idx) || !file_identity(path, size, mtime) || size != idx.file_size || mtime != idx.mtime) {
        wgs_set_error("%s is not an index of %s", index_path, path);
        return 2;
    }
    *sites = idx.sites;
    return 0;
}

/* A reader positioned so that the next row it returns is `first_row` (0-based site index): it starts inflating
 * at the last access point at or before that row and skips the few lines in between without parsing. */
int wgs_reader_open_indexed(const char *path, const char *index_path, int64_t first_row, int threads, wgs_reader **out)
{
    if (!path || !index_path || !out || first_row < 0) {
        wgs_set_error("bad argument");
        return 2;
    }
    BeagleIndex idx;
    uint64_t size = 0, mtime = 0;
    if (!load_index(index_path, idx) || !file_identity(path, size, mtime) || size != idx.file_size || mtime != idx.mtime) {
        wgs_set_error("%s is not an index of %s", index_path, path);
        return 2;
    }
    // data row r is non-blank line r + 1; an access point serves it when lines_before <= r + 1 (it may sit inside
    // line lines_before, whose remainder is then skipped: that costs one more line of margin)
    const AccessPoint *best = nullptr;
    for (const auto &a : idx.points) {
        const int64_t first_full = a.lines_before + (a.at_line_start ? 0 : 1);
        if (first_full <= first_row + 1 && (!best || a.out > best->out)) best = &a;
    }
    wgs_reader *r = new wgs_reader();
    r->samples = idx.samples;
    r->gl_cols = idx.gl_cols;
    r->n_inds = idx.gl_cols / 3;
    r->threads = threads > 0 ? threads : 1;
    if (!r->buf.resize(64u << 20)) {
        wgs_set_error("out of memory");
        delete r;
        return 1;
    }
    if (!r->src.open(path, best)) {
        wgs_set_error("cannot open Beagle file %s at its access point", path);
        delete r;
        return 2;
    }
    if (!best || best->member_start) r->src.try_bgzf(r->threads);
    if (r->src.bgzf) r->fill_cap = 1u << 20;
    const bool have_best = best != nullptr;
    const int64_t best_lines = best ? best->lines_before : 0;
    const bool best_line_start = best ? best->at_line_start != 0 : true, best_content = best ? best->content != 0 : false;
    if (!r->src.bgzf && r->threads > 1) {                    // plain gzip: inflate the stretches between access points in parallel
        std::vector<AccessPoint> segs;
        if (best) {
            segs.push_back(*best);
        } else {
            AccessPoint start;
            start.member_start = 1;
            start.at_line_start = 1;
            segs.push_back(start);
        }
        for (auto &a : idx.points)
            if (a.out > segs.back().out) segs.push_back(std::move(a));   // idx.points (and `best`) are spent from here on
        best = nullptr;
        r->src.use_segments(path, std::move(segs), r->threads);
    }
    int64_t line = 0;            // non-blank line index of the next complete line in the buffer
    if (have_best) {
        if (!fill(r)) {
            wgs_set_error("read error in %s", path);
            delete r;
            return 1;
        }
        line = best_lines;
        if (!best_line_start) {                          // drop the tail of the line the access point sits in
            bool content = best_content;
            for (;;) {
                const char *b = r->buf.data() + r->pos;
                const char *nl = (const char *)memchr(b, '\n', r->len - r->pos);
                const char *e = nl ? nl : r->buf.data() + r->len;
                for (const char *t = b; t < e && !content; ++t) content = !is_delim(*t);
                r->pos = (size_t)(e - r->buf.data()) + (nl ? 1 : 0);
                if (nl || r->eof) break;
                if (!fill(r)) {
                    wgs_set_error("read error in %s", path);
                    delete r;
                    return 1;
                }
                if (r->eof && r->pos >= r->len) break;
            }
            line += content ? 1 : 0;
        }
    } else {
        line = 0;                                        // from the first byte: the header is line 0
    }
    // skip whole lines up to the wanted row (line index first_row + 1); from the start that includes the header
    int64_t to_skip = first_row + 1 - line, got = 0;
    if (to_skip < 0) {
        wgs_set_error("index of %s is inconsistent", path);
        delete r;
        return 1;
    }
    if (to_skip > 0 && (wgs_reader_skip(r, to_skip, &got) != 0 || got != to_skip)) {
        if (got != to_skip) wgs_set_error("Beagle file %s is shorter than its index says", path);
        delete r;
        return 1;
    }
    r->lines_read = first_row;
    r->fill_cap = (size_t)-1;
    *out = r;
    return 0;
}

int wgs_reader_n_individuals(wgs_reader *r) { return r ? r->n_inds : 0; }

const char *wgs_reader_sample_name(wgs_reader *r, int i)
{
    if (!r || i < 0 || i >= (int)r->samples.size()) return nullptr;
    return r->samples[i].c_str();
}

int wgs_reader_next(wgs_reader *r, float *rows, int64_t max_rows, int64_t *nrows)
{
    if (!r || !rows || !nrows || max_rows < 0) {
        wgs_set_error("bad argument");
        return 2;
    }
    r->chunk_sites.clear();
    int64_t done = 0;
    const size_t row_floats = (size_t)2 * r->n_inds;
    std::vector<Line> lines;
    std::vector<std::string> sites;
    while (done < max_rows) {
        // complete lines available in the buffer
        lines.clear();
        size_t scan = r->pos;
        while ((int64_t)lines.size() < max_rows - done && scan < r->len) {
            const char *nl = (const char *)memchr(r->buf.data() + scan, '\n', r->len - scan);
            if (!nl) {
                if (!r->eof) break;
                nl = r->buf.data() + r->len;                 // last line without newline
            }
            const char *b = r->buf.data() + scan, *e = nl;
            scan = (size_t)(nl - r->buf.data()) + (nl < r->buf.data() + r->len ? 1 : 0);
            const char *t = b;
            while (t < e && is_delim(*t)) ++t;
            if (t < e) lines.push_back({b, e});              // blank lines are skipped
            if (nl == r->buf.data() + r->len) break;
        }
        if (lines.empty()) {
            r->pos = scan;
            if (r->eof && r->pos >= r->len) break;
            const size_t before = r->len - r->pos;
            if (!fill(r)) {
                wgs_set_error("read error while inflating the Beagle file");
                return 1;
            }
            if (r->eof && r->len - r->pos == before && before == 0) break;
            continue;
        }
        sites.assign(lines.size(), std::string());
        const int T = (int)std::min<size_t>((size_t)r->threads, lines.size());
        std::vector<int64_t> bad(T, -1);
        auto work = [&](int t) {
            for (size_t i = (size_t)t; i < lines.size(); i += (size_t)T)
                if (!parse_line(r, lines[i], rows + ((size_t)done + i) * row_floats, &sites[i]) && bad[t] < 0) bad[t] = (int64_t)i;
        };
        if (T <= 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(work, t);
            for (auto &x : th) x.join();
        }
        for (int t = 0; t < T; ++t)
            if (bad[t] >= 0) {
                wgs_set_error("Beagle data line %lld has fewer than %d genotype-likelihood columns",
                              (long long)(r->lines_read + done + bad[t] + 2), r->gl_cols);
                return 2;
            }
        for (auto &s : sites) {
            r->chunk_sites += s;
            r->chunk_sites += '\n';
        }
        done += (int64_t)lines.size();
        r->pos = scan;
    }
    r->lines_read += done;
    *nrows = done;
    return 0;
}

static int skip_impl(wgs_reader *r, int64_t max_rows, int64_t *nrows, bool names);

/* Skip up to max_rows data lines without parsing them (a rank that owns a later SNP range). */
int wgs_reader_skip(wgs_reader *r, int64_t max_rows, int64_t *nrows) { return skip_impl(r, max_rows, nrows, false); }

/* Same, but keep the site names of the skipped lines (wgs_reader_chunk_sites): a names-only pass. */
int wgs_reader_skip_names(wgs_reader *r, int64_t max_rows, int64_t *nrows) { return skip_impl(r, max_rows, nrows, true); }

static int skip_impl(wgs_reader *r, int64_t max_rows, int64_t *nrows, bool names)
{
    if (!r || !nrows || max_rows < 0) {
        wgs_set_error("bad argument");
        return 2;
    }
    if (names) r->chunk_sites.clear();
    int64_t done = 0;
    while (done < max_rows) {
        bool progressed = false;
        while (done < max_rows && r->pos < r->len) {
            const char *b = r->buf.data() + r->pos;
            const char *nl = (const char *)memchr(b, '\n', r->len - r->pos);
            if (!nl) {
                if (!r->eof) break;
                nl = r->buf.data() + r->len;
            }
            const char *t = b;
            while (t < nl && is_delim(*t)) ++t;
            done += t < nl;                                   // blank lines do not count
            if (names && t < nl) {
                const char *e = t;
                while (e < nl && !is_delim(*e)) ++e;
                r->chunk_sites.append(t, e);
                r->chunk_sites += '\n';
            }
            r->pos = (size_t)(nl - r->buf.data()) + (nl < r->buf.data() + r->len ? 1 : 0);
            progressed = true;
        }
        if (done >= max_rows) break;
        if (r->eof && r->pos >= r->len) break;
        const size_t before = r->len - r->pos;
        if (!fill(r)) {
            wgs_set_error("read error while inflating the Beagle file");
            return 1;
        }
        if (!progressed && r->eof && r->len - r->pos == before && before == 0) break;
    }
    r->lines_read += done;
    *nrows = done;
    return 0;
}

int wgs_reader_count_sites(const char *path, int64_t *sites) { return wgs_reader_build_index(path, nullptr, nullptr, 0, 0, sites); }

/* About how many sites a BGZF file holds, from five samples of a quarter megabyte each (start, quartiles, end): the newlines of the
 * blocks that start there, per compressed byte, times the file's size.  Milliseconds instead of the index pass's inflate of
 * everything -- for the caller that wants to size a device matrix BEFORE it knows (stream_to_device's one-pass cold path: the exact
 * count arrives with the index, which is built meanwhile).  rc 3: not BGZF, or a sample fell off the block chain. */
int wgs_reader_estimate_sites(const char *path, int64_t *estimate)
{
    if (!path || !estimate) {
        wgs_set_error("null argument");
        return 2;
    }
    *estimate = 0;
    uint64_t file_size = 0, mtime = 0;
    if (!file_identity(path, file_size, mtime)) {
        wgs_set_error("cannot open Beagle file %s", path);
        return 2;
    }
    {
        FILE *fp = fopen(path, "rb");
        if (!fp) {
            wgs_set_error("cannot open Beagle file %s", path);
            return 2;
        }
        BgzfBlock b;
        const int k = bgzf_block_at(fileno(fp), file_size, 0, b);
        fclose(fp);
        if (k <= 0) return 3;
    }
    const uint64_t window = 256u << 10;
    double newlines = 0.0, comp = 0.0;
    for (int q = 0; q < 5; ++q) {
        uint64_t lo = q == 4 ? (file_size > window ? file_size - window : 0) : file_size / 4 * (uint64_t)q;
        if (q > 0 && lo < window) continue;                           // a file smaller than the samples: the first one is all of it
        PartData pd;
        if (!bgzf_part_thread(path, file_size, lo, std::min(file_size, lo + window), false, pd)) return 3;
        for (size_t i = 0; i < pd.blocks.size(); ++i) {
            newlines += (double)pd.sum[i].lines_after_first + (pd.sum[i].has_nl ? 1.0 : 0.0);
            comp += (double)pd.blocks[i].csize;
        }
    }
    if (comp <= 0.0) return 3;
    *estimate = (int64_t)(newlines / comp * (double)file_size);
    return 0;
}

const char *wgs_reader_chunk_sites(wgs_reader *r, int64_t *bytes)
{
    if (!r) return nullptr;
    if (bytes) *bytes = (int64_t)r->chunk_sites.size();
    return r->chunk_sites.c_str();
}

}  // extern "C"

// ---- text hand-over to the device tokeniser (reader_text.h) ----------------------------------------------------
// A producer thread inflates ahead into caller-allocated (pinned) buffers, cuts each at its last newline, carries
// the partial line into the next buffer, and lists the non-blank lines -- newline scan and first tokens (site names)
// on all of the reader's threads.  Everything per VALUE happens on the GPU.
struct TextPipe {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<TextChunk *> free_q, ready_q;
    std::vector<TextChunk *> all;
    TextAllocator alloc;
    size_t chunk_bytes = 0;
    int64_t limit = -1, rows = 0;
    bool finished = false, stop = false;
    int rc = 0;
    std::string err;
    int64_t chunks = 0;
    std::vector<char> carry;    // text not handed out yet: [carry_pos, size)
    size_t carry_pos = 0;
};

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// room for `want` bytes of text plus the pad; keeps data[0 .. len)
bool chunk_reserve(TextPipe *p, TextChunk *c, size_t want)
{
    if (c->data && want + TEXT_PAD <= c->cap) return true;
    const size_t cap = want + TEXT_PAD;
    char *q = (char *)p->alloc.alloc(cap, p->alloc.user);
    if (!q) return false;
    if (c->len) memcpy(q, c->data, c->len);
    if (c->data) p->alloc.release(c->data, p->alloc.user);
    c->data = q;
    c->cap = cap;
    return true;
}

// The non-blank lines of data[0 .. len) -- every line ends with a newline, or (at the end of the file) with the buffer.
void list_lines(const char *data, size_t len, int threads, std::vector<uint32_t> &begin, std::vector<uint32_t> &end, std::string &names)
{
    begin.clear();
    end.clear();
    names.clear();
    if (len == 0) return;
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, len >> 20));
    std::vector<std::vector<uint32_t>> nl(T), lb(T), le(T);
    std::vector<std::string> nm(T);
    run_threads(T, [&](int t) {
        const size_t a = len * (size_t)t / (size_t)T, b = len * (size_t)(t + 1) / (size_t)T;
        for (const char *q = data + a, *e = data + b; q < e;) {
            q = (const char *)memchr(q, '\n', (size_t)(e - q));
            if (!q) break;
            nl[t].push_back((uint32_t)(q - data));
            ++q;
        }
    });
    // thread t lists the lines that END in its range; the first of them starts after the last newline before the range
    std::vector<size_t> start(T, 0);
    {
        size_t prev = 0;
        for (int t = 0; t < T; ++t) {
            start[t] = prev;
            if (!nl[t].empty()) prev = (size_t)nl[t].back() + 1;
        }
        if (prev < len) nl[T - 1].push_back((uint32_t)len);      // last line without a newline
    }
    run_threads(T, [&](int t) {
        size_t b = start[t];
        for (uint32_t stop : nl[t]) {
            const char *q = data + b, *e = data + stop;
            while (q < e && is_delim(*q)) ++q;
            if (q < e) {                                         // blank lines are not rows
                lb[t].push_back((uint32_t)b);
                le[t].push_back(stop);
                const char *x = q;
                while (x < e && !is_delim(*x)) ++x;
                nm[t].append(q, x);
                nm[t].push_back('\n');
            }
            b = (size_t)stop + 1;
        }
    });
    for (int t = 0; t < T; ++t) {
        begin.insert(begin.end(), lb[t].begin(), lb[t].end());
        end.insert(end.end(), le[t].begin(), le[t].end());
        names += nm[t];
    }
}

void text_producer(wgs_reader *r)
{
    TextPipe *p = r->pipe;
    auto fail = [&](int rc, const char *msg) {
        std::lock_guard<std::mutex> lk(p->mu);
        p->rc = rc;
        p->err = msg;
        p->finished = true;
        p->cv.notify_all();
    };
    for (;;) {
        TextChunk *c = nullptr;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || !p->free_q.empty(); });
            if (p->stop) return;
            c = p->free_q.front();
            p->free_q.pop_front();
        }
        const double t0 = now_s();
        c->len = 0;
        const size_t pending0 = p->carry.size() - p->carry_pos;
        if (!chunk_reserve(p, c, std::max(p->chunk_bytes, std::min(pending0, p->chunk_bytes) + r->src.want_room()))) return fail(1, "out of (pinned) memory for the text buffers");
        size_t complete = 0;
        bool from_carry_only = false;
        for (;;) {
            // first what is pending from before (the partial line of the last chunk; at the start, everything the
            // line-oriented calls had already inflated), then the inflater
            if (p->carry_pos < p->carry.size()) {
                const size_t take = std::min(p->carry.size() - p->carry_pos, c->cap - TEXT_PAD - c->len);
                memcpy(c->data + c->len, p->carry.data() + p->carry_pos, take);
                c->len += take;
                p->carry_pos += take;
                if (p->carry_pos == p->carry.size()) {
                    p->carry.clear();
                    p->carry_pos = 0;
                }
            }
            from_carry_only = p->carry_pos > 0;              // pending text left: nothing was inflated into this chunk
            while (!from_carry_only && !r->eof) {
                const size_t room = c->cap - TEXT_PAD - c->len;
                if (room < r->src.min_room()) break;
                const bool batch = r->src.segmented;
                const long got = r->src.read(c->data + c->len, room);
                if (got == -2) break;
                if (got < 0) return fail(1, "read error while inflating the Beagle file");
                if (got == 0) {
                    r->eof = true;
                    break;
                }
                c->len += (size_t)got;
                if (batch) break;
            }
            const char *last = c->len ? (const char *)memrchr(c->data, '\n', c->len) : nullptr;
            if (r->eof && !from_carry_only) {
                complete = c->len;
                break;
            }
            if (last && (from_carry_only || c->cap - TEXT_PAD - c->len < r->src.min_room() || r->src.segmented || c->len >= p->chunk_bytes / 2)) {
                complete = (size_t)(last - c->data) + 1;
                break;
            }
            // one line longer than the buffer, or the next unit does not fit behind what is there: grow
            if (c->len + std::max(p->chunk_bytes, r->src.want_room()) >= (1ull << 32) - TEXT_PAD)
                return fail(1, "a Beagle line longer than 4 GiB");
            if (!chunk_reserve(p, c, c->len + std::max(p->chunk_bytes, r->src.want_room()))) return fail(1, "out of (pinned) memory for the text buffers");
        }
        if (from_carry_only) p->carry_pos -= c->len - complete;      // the cut-off tail is still there, right before carry_pos
        else p->carry.assign(c->data + complete, c->data + c->len);
        c->len = complete;
        memset(c->data + c->len, '\n', TEXT_PAD);
        const double t1 = now_s();
        list_lines(c->data, c->len, r->threads, c->begin, c->end, c->names);
        c->inflate_s = t1 - t0;
        c->scan_s = now_s() - t1;
        bool last_chunk = r->eof && p->carry.empty();
        c->first_row = p->rows;
        if (p->limit >= 0 && p->rows + (int64_t)c->begin.size() >= p->limit) {
            const size_t keep = (size_t)(p->limit - p->rows);
            if (keep < c->begin.size()) {
                c->begin.resize(keep);
                c->end.resize(keep);
                size_t at = 0;
                for (size_t i = 0; i < keep; ++i) at = c->names.find('\n', at) + 1;
                c->names.resize(at);
            }
            last_chunk = true;
        }
        p->rows += (int64_t)c->begin.size();
        p->chunks += c->begin.empty() ? 0 : 1;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (c->begin.empty()) p->free_q.push_back(c);
            else p->ready_q.push_back(c);
            if (last_chunk) p->finished = true;
            p->cv.notify_all();
        }
        if (last_chunk) return;
    }
}

}  // namespace

bool reader_text_is_bgzf(const wgs_reader *r) { return r && r->src.bgzf; }

int reader_text_start(wgs_reader *r, size_t chunk_bytes, int nbuf, TextAllocator a, int64_t limit_rows)
{
    if (!r || r->pipe || !a.alloc || !a.release || nbuf < 1) {
        wgs_set_error("bad argument");
        return 2;
    }
    TextPipe *p = new TextPipe();
    p->alloc = a;
    p->chunk_bytes = std::min<size_t>(std::max<size_t>(chunk_bytes, 1u << 20), (size_t)2 << 30);
    p->limit = limit_rows;
    // what the line-oriented calls left in the reader's buffer comes first
    p->carry.assign(r->buf.data() + r->pos, r->buf.data() + r->len);
    r->pos = r->len = 0;
    for (int i = 0; i < nbuf; ++i) {
        TextChunk *c = new TextChunk();
        c->slot = i;
        p->all.push_back(c);
        p->free_q.push_back(c);
    }
    r->pipe = p;
    if (limit_rows == 0) p->finished = true;
    else p->th = std::thread(text_producer, r);
    return 0;
}

int reader_text_next(wgs_reader *r, TextChunk **out, double *waited_s)
{
    TextPipe *p = r ? r->pipe : nullptr;
    if (!p || !out) {
        wgs_set_error("bad argument");
        return 2;
    }
    const double t0 = now_s();
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return !p->ready_q.empty() || p->finished; });
    if (waited_s) *waited_s = now_s() - t0;
    *out = nullptr;
    if (!p->ready_q.empty()) {
        *out = p->ready_q.front();
        p->ready_q.pop_front();
        return 0;
    }
    if (p->rc) wgs_set_error("%s", p->err.c_str());
    return p->rc;
}

void reader_text_release(wgs_reader *r, TextChunk *c)
{
    TextPipe *p = r ? r->pipe : nullptr;
    if (!p || !c) return;
    std::lock_guard<std::mutex> lk(p->mu);
    p->free_q.push_back(c);
    p->cv.notify_all();
}

void reader_text_stop(wgs_reader *r)
{
    TextPipe *p = r ? r->pipe : nullptr;
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
        p->cv.notify_all();
    }
    if (p->th.joinable()) p->th.join();
    r->lines_read += p->rows;
    r->text_chunks = p->chunks;
    for (TextChunk *c : p->all) {
        if (c->data) p->alloc.release(c->data, p->alloc.user);
        delete c;
    }
    delete p;
    r->pipe = nullptr;
}

// ---- the compressed hand-over ----------------------------------------------------------------------------------
struct CompPipe {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<CompChunk *> free_q, ready_q;
    std::vector<CompChunk *> all;
    TextAllocator alloc;
    size_t text_cap = 0, max_members = 0;
    uint64_t file_off = 0;
    bool finished = false, stop = false;
    size_t pre_pos = 0, pre_end = 0;   // text the line-oriented calls had inflated already: [pre_pos, pre_end) of the reader's buffer
    int rc = 0;
    std::string err;
};

namespace {
void comp_producer(wgs_reader *r)
{
    CompPipe *p = r->cpipe;
    const int fd = fileno(r->src.fp);
    auto fail = [&](const char *msg) {
        std::lock_guard<std::mutex> lk(p->mu);
        p->rc = 1;
        p->err = msg;
        p->finished = true;
        p->cv.notify_all();
    };
    for (;;) {
        CompChunk *c = nullptr;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || !p->free_q.empty(); });
            if (p->stop) return;
            c = p->free_q.front();
            p->free_q.pop_front();
        }
        const double t0 = now_s();
        c->in_off.clear();
        c->in_len.clear();
        c->isize.clear();
        c->text_bytes = 0;
        c->len = 0;
        c->last = false;
        c->pre_text = nullptr;
        c->pre_len = 0;
        if (p->pre_pos < p->pre_end) {                                     // first what was inflated already, a chunk's worth at a time
            c->pre_text = r->buf.data() + p->pre_pos;
            c->pre_len = std::min(p->pre_end - p->pre_pos, p->text_cap);
            p->pre_pos += c->pre_len;
            std::lock_guard<std::mutex> lk(p->mu);
            p->ready_q.push_back(c);
            p->cv.notify_all();
            continue;
        }
        // The file in slices (all threads copy their share out of the page cache), whole members counted off as they arrive,
        // until the device's text buffer or this staging buffer is full; the slices shrink towards the end so that little
        // is read twice.
        size_t got = 0, at = 0;
        bool file_end = false, full = false, corrupt = false, partial = false;
        while (!full && !file_end && !corrupt && got < c->cap) {
            size_t slice = (size_t)64 << 20;
            if (at > 0 && c->text_bytes > 0) {
                const double per_text = (double)at / (double)c->text_bytes;
                const double est = (double)(p->text_cap - c->text_bytes) * per_text * 1.03 + 262144.0 - (double)(got - at);
                slice = (size_t)std::min<double>((double)slice, std::max<double>(est, (double)(1 << 20)));
            }
            slice = std::min(slice, c->cap - got);
            const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)r->threads, slice >> 20));
            std::vector<size_t> part(T, 0);
            run_threads(T, [&](int t) {
                const size_t a = slice * (size_t)t / (size_t)T, b = slice * (size_t)(t + 1) / (size_t)T;
                size_t done = 0;
                while (a + done < b) {
                    const ssize_t k = pread(fd, c->comp + got + a + done, b - a - done, (off_t)(p->file_off + got + a + done));
                    if (k <= 0) break;
                    done += (size_t)k;
                }
                part[t] = done;
            });
            size_t n = 0;
            for (int t = 0; t < T; ++t) {
                n += part[t];
                if (part[t] < slice * (size_t)(t + 1) / (size_t)T - slice * (size_t)t / (size_t)T) {
                    file_end = true;
                    break;
                }
            }
            got += n;
            partial = false;
            while (at < got) {
                uint32_t hdr = 0;
                const long sz = bgzf_member_size(c->comp + at, got - at, &hdr);
                if (sz == 0) {
                    corrupt = true;
                    break;
                }
                if (sz < 0 || at + (size_t)sz > got) {                         // a partial member: the next slice completes it
                    partial = true;
                    break;
                }
                const unsigned char *tl = c->comp + at + sz - 4;
                const uint32_t isz = tl[0] | ((uint32_t)tl[1] << 8) | ((uint32_t)tl[2] << 16) | ((uint32_t)tl[3] << 24);
                if (isz > 65536 || (uint32_t)sz < hdr + 8) {
                    corrupt = true;
                    break;
                }
                if (c->text_bytes + isz > p->text_cap || (p->max_members && isz && c->isize.size() >= p->max_members)) {
                    full = true;
                    break;
                }
                if (isz) {
                    c->in_off.push_back(at + hdr);
                    c->in_len.push_back((uint32_t)sz - hdr - 8);
                    c->isize.push_back(isz);
                }
                c->text_bytes += isz;
                at += (size_t)sz;
            }
        }
        if (corrupt || (file_end && partial)) return fail("read error in the BGZF file (corrupt or truncated member)");
        if (at == 0 && !file_end) return fail("a BGZF member larger than the staging buffer");
        c->len = at;
        p->file_off += at;
        c->last = file_end && at == got;                                   // the file ended with this chunk's last member
        c->read_s = now_s() - t0;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->ready_q.push_back(c);
            if (c->last) p->finished = true;
            p->cv.notify_all();
        }
        if (c->last) return;
    }
}
}  // namespace

int reader_comp_start(wgs_reader *r, size_t comp_bytes, size_t text_cap, int nbuf, TextAllocator a, size_t max_members)
{
    if (!r || r->pipe || r->cpipe || !r->src.bgzf || !a.alloc || !a.release || nbuf < 1) {
        wgs_set_error("bad argument");
        return 2;
    }
    CompPipe *p = new CompPipe();
    p->alloc = a;
    p->text_cap = std::max<size_t>(text_cap, 1u << 20);
    p->max_members = max_members;
    comp_bytes = std::max<size_t>(comp_bytes, 1u << 20);
    p->pre_pos = r->pos;
    p->pre_end = r->len;
    // what the block-parallel host path had read ahead but not inflated is read again from the file
    p->file_off = (uint64_t)ftello(r->src.fp) - (uint64_t)(r->src.clen - r->src.cpos);
    for (int i = 0; i < nbuf; ++i) {
        CompChunk *c = new CompChunk();
        c->comp = (unsigned char *)a.alloc(comp_bytes, a.user);
        c->cap = comp_bytes;
        p->all.push_back(c);
        if (!c->comp) {
            for (CompChunk *x : p->all) {
                if (x->comp) a.release(x->comp, a.user);
                delete x;
            }
            delete p;
            wgs_set_error("out of (pinned) memory for the compressed staging buffers");
            return 1;
        }
        p->free_q.push_back(c);
    }
    r->cpipe = p;
    p->th = std::thread(comp_producer, r);
    return 0;
}

int reader_comp_next(wgs_reader *r, CompChunk **out, double *waited_s)
{
    CompPipe *p = r ? r->cpipe : nullptr;
    if (!p || !out) {
        wgs_set_error("bad argument");
        return 2;
    }
    const double t0 = now_s();
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return !p->ready_q.empty() || p->finished; });
    if (waited_s) *waited_s = now_s() - t0;
    *out = nullptr;
    if (!p->ready_q.empty()) {
        *out = p->ready_q.front();
        p->ready_q.pop_front();
        return 0;
    }
    if (p->rc) wgs_set_error("%s", p->err.c_str());
    return p->rc;
}

void reader_comp_release(wgs_reader *r, CompChunk *c)
{
    CompPipe *p = r ? r->cpipe : nullptr;
    if (!p || !c) return;
    std::lock_guard<std::mutex> lk(p->mu);
    p->free_q.push_back(c);
    p->cv.notify_all();
}

void reader_comp_stop(wgs_reader *r)
{
    CompPipe *p = r ? r->cpipe : nullptr;
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
        p->cv.notify_all();
    }
    if (p->th.joinable()) p->th.join();
    for (CompChunk *c : p->all) {
        if (c->comp) p->alloc.release(c->comp, p->alloc.user);
        delete c;
    }
    delete p;
    r->cpipe = nullptr;
    r->pos = r->len = 0;               // the buffered text went to the device with the first chunk
    r->eof = true;
}

// compressed bytes from the first member not yet inflated to the end of the file (-1: unknown)
int64_t reader_comp_bytes_left(wgs_reader *r)
{
    if (!r || !r->src.fp) return -1;
    struct stat sb;
    if (fstat(fileno(r->src.fp), &sb) != 0) return -1;
    const int64_t at = (int64_t)ftello(r->src.fp) - (int64_t)(r->src.clen - r->src.cpos);
    return std::max<int64_t>(0, (int64_t)sb.st_size - at);
}

bool reader_inflate_member(const unsigned char *deflate, uint32_t in_len, uint32_t isize, unsigned char *out)
{
    BlockInflater inf;
    if (!inf.init()) return false;
    BgzfBlock b;
    b.off = 0;
    b.hdr = 0;
    b.csize = in_len + 8;
    b.isize = isize;
    return inf.run(deflate, b, out);
}

void reader_add_lines_read(wgs_reader *r, int64_t rows) { r->lines_read += rows; }

int reader_text_parse_line(const wgs_reader *r, const char *b, const char *e, float *out)
{
    std::string site;
    return parse_line(r, Line{b, e}, out, &site) ? 0 : 1;
}

int reader_text_n_inds(const wgs_reader *r) { return r->n_inds; }
int reader_text_gl_cols(const wgs_reader *r) { return r->gl_cols; }
int64_t reader_text_lines_read(const wgs_reader *r) { return r->lines_read; }

extern "C" {

/* Test hook (needs no GPU): the rows of wgs_reader_next, but through the text hand-over -- the producer thread, the
 * carried partial lines, the parallel newline scan, the row limit -- with ordinary memory for the buffers and the host
 * parser in place of the device tokeniser.  One call drains the reader: rows[max_rows][2n], site names through
 * wgs_reader_chunk_sites. */
int64_t wgs_debug_reader_text_chunks(wgs_reader *r) { return r ? r->text_chunks : 0; }

int wgs_debug_reader_text_rows(wgs_reader *r, int64_t chunk_bytes, int64_t limit_rows, float *rows, int64_t max_rows, int64_t *nrows)
{
    if (!r || !nrows) {                                    // rows == NULL: only drain the hand-over (inflate + line lists), for timing
        wgs_set_error("bad argument");
        return 2;
    }
    TextAllocator a;
    a.alloc = [](size_t n, void *) { return malloc(n); };
    a.release = [](void *p, void *) { free(p); };
    if (int rc = reader_text_start(r, (size_t)chunk_bytes, 2, a, limit_rows)) return rc;
    r->chunk_sites.clear();
    const size_t row_floats = (size_t)2 * r->n_inds;
    int64_t done = 0;
    int rc = 0;
    for (;;) {
        TextChunk *c = nullptr;
        if ((rc = reader_text_next(r, &c, nullptr)) != 0 || !c) break;
        if (c->first_row != done) rc = 1, wgs_set_error("text chunks out of order");
        if (!rows) done += (int64_t)c->begin.size();
        for (size_t i = 0; rows && i < c->begin.size() && !rc; ++i) {
            if (done >= max_rows) {
                rc = 2;
                wgs_set_error("more rows than the caller has room for");
            } else if (reader_text_parse_line(r, c->data + c->begin[i], c->data + c->end[i], rows + (size_t)done * row_floats)) {
                rc = 2;
                wgs_set_error("Beagle data line %lld has fewer than %d genotype-likelihood columns", (long long)(r->lines_read + done + 2), r->gl_cols);
            } else {
                ++done;
            }
        }
        r->chunk_sites += c->names;
        reader_text_release(r, c);
        if (rc) break;
    }
    reader_text_stop(r);
    *nrows = done;
    return rc;
}

/* Debug / tests: the COMPRESSED hand-over (reader_text.h: CompChunk -- what the device-resident ingest consumes) driven on the
 * host: every chunk's pre-inflated text and members (inflated here with the host's inflater) appended to text[0 .. cap);
 * info[0] = chunks, info[1] = members, info[2] = largest text of one chunk, info[3] = chunks of pre-inflated text. */
int wgs_debug_reader_comp_text(wgs_reader *r, int64_t comp_bytes, int64_t text_cap, int nbuf, char *text, int64_t cap, int64_t *bytes, int64_t *info)
{
    if (!r || !text || !bytes || !info) {
        wgs_set_error("bad argument");
        return 2;
    }
    TextAllocator a;
    a.alloc = [](size_t n, void *) { return malloc(n); };
    a.release = [](void *p, void *) { free(p); };
    if (int rc = reader_comp_start(r, (size_t)comp_bytes, (size_t)text_cap, nbuf, a)) return rc;
    int64_t at = 0;
    int rc = 0;
    info[0] = info[1] = info[2] = info[3] = 0;
    for (;;) {
        CompChunk *c = nullptr;
        if ((rc = reader_comp_next(r, &c, nullptr)) != 0 || !c) break;
        const int64_t here = (int64_t)c->pre_len + (int64_t)c->text_bytes;
        if (at + here > cap) {
            rc = 2;
            wgs_set_error("more text than the caller has room for");
        } else {
            if (c->pre_len) memcpy(text + at, c->pre_text, c->pre_len);
            at += (int64_t)c->pre_len;
            for (size_t i = 0; i < c->isize.size() && !rc; ++i) {
                if (!reader_inflate_member(c->comp + c->in_off[i], c->in_len[i], c->isize[i], reinterpret_cast<unsigned char *>(text + at))) {
                    rc = 1;
                    wgs_set_error("read error in the BGZF file (corrupt block)");
                }
                at += c->isize[i];
            }
        }
        info[0] += 1;
        info[1] += (int64_t)c->isize.size();
        info[2] = std::max<int64_t>(info[2], here);
        info[3] += c->pre_len ? 1 : 0;
        const bool last = c->last;
        reader_comp_release(r, c);
        if (rc || last) break;
    }
    reader_comp_stop(r);
    *bytes = at;
    return rc;
}

}  // extern "C"
