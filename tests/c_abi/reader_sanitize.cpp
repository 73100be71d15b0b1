// Sanitizer driver for csrc/reader.cpp (host code only): index, indexed opens at several rows and thread counts,
// full reads; prints a checksum per configuration -- all must agree.  Then the two hand-overs of the device ingest, whose
// producer threads run beside the consumer: inflated text (wgs_debug_reader_text_rows) and, for BGZF files, compressed
// members (wgs_debug_reader_comp_text), both with small buffers so that many chunks change hands.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "wgsassign_hip.h"
#include "wgsassign_hip_debug.h"

static char g_err[1024];
void wgs_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

static uint64_t read_all(wgs_reader *r, int64_t chunk, int64_t *rows_out)
{
    const int n = wgs_reader_n_individuals(r);
    std::vector<float> buf((size_t)chunk * 2 * n);
    uint64_t h = 1469598103934665603ull;
    int64_t total = 0;
    for (;;) {
        int64_t got = 0;
        if (wgs_reader_next(r, buf.data(), chunk, &got)) { fprintf(stderr, "next failed: %s\n", g_err); exit(2); }
        if (got == 0) break;
        const unsigned char *p = (const unsigned char *)buf.data();
        for (size_t i = 0; i < (size_t)got * 2 * n * 4; ++i) h = (h ^ p[i]) * 1099511628211ull;
        int64_t nb = 0;
        const char *names = wgs_reader_chunk_sites(r, &nb);
        for (int64_t i = 0; i < nb; ++i) h = (h ^ (unsigned char)names[i]) * 1099511628211ull;
        total += got;
    }
    *rows_out = total;
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 1;
    const char *path = argv[1];
    const int64_t span = atoll(argv[2]);
    std::string idx = std::string(path) + ".idx", names = std::string(path) + ".names";
    int64_t sites = 0, sites2 = 0;
    if (wgs_reader_build_index(path, idx.c_str(), names.c_str(), span, 4096, &sites)) { fprintf(stderr, "index: %s\n", g_err); return 2; }
    if (wgs_reader_build_index(path, idx.c_str(), nullptr, span, 4096, &sites2) || sites2 != sites) { fprintf(stderr, "index 2: %s\n", g_err); return 2; }
    if (wgs_reader_count_sites(path, &sites2) || sites2 != sites) return 3;
    printf("sites %lld\n", (long long)sites);
    uint64_t want[3] = {0, 0, 0};
    const int64_t firsts[3] = {0, sites / 3, sites - 5 > 0 ? sites - 5 : 0};
    for (int threads : {1, 2, 7, 16}) {
        for (int f = 0; f < 3; ++f) {
            wgs_reader *r = nullptr;
            if (wgs_reader_open_indexed(path, idx.c_str(), firsts[f], threads, &r)) { fprintf(stderr, "open: %s\n", g_err); return 2; }
            int64_t rows = 0;
            const uint64_t h = read_all(r, 777, &rows);
            wgs_reader_close(r);
            if (rows != sites - firsts[f]) { fprintf(stderr, "rows %lld != %lld\n", (long long)rows, (long long)(sites - firsts[f])); return 4; }
            if (threads == 1) want[f] = h;
            else if (h != want[f]) { fprintf(stderr, "checksum differs: threads %d first %lld\n", threads, (long long)firsts[f]); return 5; }
        }
        wgs_reader *r = nullptr;
        if (wgs_reader_open(path, threads, &r)) return 2;
        int64_t rows = 0, skipped = 0;
        if (wgs_reader_skip_names(r, 10, &skipped)) return 2;
        const uint64_t h = read_all(r, 100000, &rows);
        (void)h;
        wgs_reader_close(r);
        if (rows + skipped != sites) return 6;
    }
    for (int threads : {1, 3, 8}) {
        const int64_t first = sites / 3;
        wgs_reader *r = nullptr;
        if (wgs_reader_open_indexed(path, idx.c_str(), first, threads, &r)) { fprintf(stderr, "open: %s\n", g_err); return 2; }
        const int n = wgs_reader_n_individuals(r);
        std::vector<float> rows((size_t)(sites - first) * 2 * n);
        int64_t got = 0;
        if (wgs_debug_reader_text_rows(r, 1 << 20, -1, rows.data(), sites - first, &got) || got != sites - first) {
            fprintf(stderr, "text hand-over: %s (%lld rows)\n", g_err, (long long)got);
            return 7;
        }
        wgs_reader_close(r);
        if (wgs_reader_open_indexed(path, idx.c_str(), 0, threads, &r)) return 2;      // from the start of the data
        std::vector<char> text((size_t)256 << 20);
        int64_t bytes = 0, info[4] = {0, 0, 0, 0};
        const int rc = wgs_debug_reader_comp_text(r, 1 << 20, 2 << 20, 2, text.data(), (int64_t)text.size(), &bytes, info);
        wgs_reader_close(r);
        if (rc == 0) {
            int64_t lines = 0;
            for (int64_t i = 0; i < bytes; ++i) lines += text[(size_t)i] == '\n';
            printf("threads %d: compressed hand-over %lld chunks, %lld members, %lld bytes, %lld lines\n", threads, (long long)info[0], (long long)info[1],
                   (long long)bytes, (long long)lines);
            if (lines < sites) return 8;
        } else if (!strstr(g_err, "bad argument")) {       // a plain gzip file has no members to hand over: refused
            fprintf(stderr, "compressed hand-over: %s\n", g_err);
            return 8;
        }
    }
    printf("ok\n");
    return 0;
}
