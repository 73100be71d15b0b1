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
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/wgsassign_hip.h"

void wgs_set_error(const char *fmt, ...);

namespace {

inline bool is_delim(char c) { return c == '\t' || c == ' ' || c == '\n' || c == '\r'; }

const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// atof of the token [p, e) (no delimiters inside).
inline double parse_double(const char *p, const char *e)
{
    const char *q = p;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) neg = *q++ == '-';
    uint64_t mant = 0;
    int digits = 0, frac = 0;
    bool dot = false, ok = q < e;
    for (; q < e; ++q) {
        const char c = *q;
        if (c >= '0' && c <= '9') {
            if (mant == 0 && c == '0' && !dot) continue;     // leading integer zeros
            if (mant != 0 || c != '0') ++digits;             // significant digits
            mant = mant * 10 + (uint64_t)(c - '0');
            if (dot) ++frac;
        } else if (c == '.' && !dot) {
            dot = true;
        } else {
            ok = false;
            break;
        }
    }
    // exact when the mantissa fits 53 bits and the power of ten is exactly representable
    if (ok && digits <= 15 && frac <= 22 && mant < (1ull << 53)) {
        const double v = (double)mant / kPow10[frac];
        return neg ? -v : v;
    }
    char tmp[64];
    const size_t len = (size_t)(e - p);
    if (len < sizeof tmp) {
        memcpy(tmp, p, len);
        tmp[len] = 0;
        return strtod(tmp, nullptr);
    }
    std::string s(p, e);
    return strtod(s.c_str(), nullptr);
}

struct Line {
    const char *begin, *end;   // without the newline
};

}  // namespace

struct wgs_reader {
    gzFile gz = nullptr;
    std::vector<std::string> samples;
    int gl_cols = 0;   // GL columns in the header (3 per individual)
    int n_inds = 0;
    std::vector<char> buf;
    size_t len = 0, pos = 0;
    bool eof = false;
    std::string chunk_sites;   // '\n'-joined site names of the last chunk
    int threads = 1;
    int64_t lines_read = 0;
};

static bool fill(wgs_reader *r)
{
    // keep the unconsumed tail, append more inflated bytes
    if (r->pos > 0) {
        memmove(r->buf.data(), r->buf.data() + r->pos, r->len - r->pos);
        r->len -= r->pos;
        r->pos = 0;
    }
    if (r->len == r->buf.size()) r->buf.resize(r->buf.size() * 2);   // a single line longer than the buffer
    while (!r->eof && r->len < r->buf.size()) {
        const int want = (int)std::min<size_t>(r->buf.size() - r->len, 1u << 30);
        const int got = gzread(r->gz, r->buf.data() + r->len, (unsigned)want);
        if (got < 0) return false;
        if (got == 0) {
            r->eof = true;
            break;
        }
        r->len += (size_t)got;
    }
    return true;
}

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
    for (int i = 0; i < 3 * r->n_inds; ++i) {
        if (!next(tb, te)) return false;
        if ((i + 1) % 3 != 0) *out++ = (float)parse_double(tb, te);   // reader_cy.pyx:62-66
    }
    return true;
}

extern "C" {

int wgs_reader_open(const char *path, int threads, wgs_reader **out)
{
    if (!path || !out) {
        wgs_set_error("null argument");
        return 2;
    }
    gzFile gz = gzopen(path, "rb");
    if (!gz) {
        wgs_set_error("cannot open Beagle file %s", path);
        return 2;
    }
    gzbuffer(gz, 1u << 20);
    wgs_reader *r = new wgs_reader();
    r->gz = gz;
    r->threads = threads > 0 ? threads : 1;
    r->buf.resize(64u << 20);
    if (!fill(r)) {
        wgs_set_error("read error in %s", path);
        gzclose(gz);
        delete r;
        return 1;
    }
    // header line (reader_cy.pyx:35-49)
    const char *nl = (const char *)memchr(r->buf.data(), '\n', r->len);
    while (!nl && !r->eof) {
        if (!fill(r)) break;
        nl = (const char *)memchr(r->buf.data(), '\n', r->len);
    }
    const char *hend = nl ? nl : r->buf.data() + r->len;
    const char *p = r->buf.data();
    int tok = 0;
    while (p < hend) {
        while (p < hend && is_delim(*p)) ++p;
        if (p >= hend) break;
        const char *tb = p;
        while (p < hend && !is_delim(*p)) ++p;
        ++tok;
        if (tok > 3) {
            const int c = tok - 3;
            if (c % 3 == 1) r->samples.emplace_back(tb, p);
        }
    }
    r->gl_cols = tok > 3 ? tok - 3 : 0;
    r->n_inds = r->gl_cols / 3;
    r->pos = nl ? (size_t)(nl - r->buf.data()) + 1 : r->len;
    *out = r;
    return 0;
}

void wgs_reader_close(wgs_reader *r)
{
    if (!r) return;
    if (r->gz) gzclose(r->gz);
    delete r;
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

int wgs_reader_count_sites(const char *path, int64_t *sites)
{
    if (!path || !sites) {
        wgs_set_error("null argument");
        return 2;
    }
    gzFile gz = gzopen(path, "rb");
    if (!gz) {
        wgs_set_error("cannot open Beagle file %s", path);
        return 2;
    }
    gzbuffer(gz, 1u << 20);
    std::vector<char> buf(16u << 20);
    int64_t lines = 0;
    bool content = false;   // the current line holds a non-delimiter character
    for (;;) {
        const int got = gzread(gz, buf.data(), (unsigned)buf.size());
        if (got < 0) {
            gzclose(gz);
            wgs_set_error("read error in %s", path);
            return 1;
        }
        if (got == 0) break;
        for (int i = 0; i < got; ++i) {
            const char c = buf[i];
            if (c == '\n') {
                lines += content;
                content = false;
            } else if (!is_delim(c)) {
                content = true;
            }
        }
    }
    lines += content;
    gzclose(gz);
    *sites = lines > 0 ? lines - 1 : 0;   // minus the header line
    return 0;
}

const char *wgs_reader_chunk_sites(wgs_reader *r, int64_t *bytes)
{
    if (!r) return nullptr;
    if (bytes) *bytes = (int64_t)r->chunk_sites.size();
    return r->chunk_sites.c_str();
}

}  // extern "C"
