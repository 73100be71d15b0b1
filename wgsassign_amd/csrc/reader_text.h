// Internal interface between the streamed reader (reader.cpp, host only) and the device-side ingest
// (ingest.hip): the reader hands over inflated TEXT -- whole lines in a caller-allocated (pinned) buffer plus
// the offsets of every non-blank line -- and the MI355X tokenises it straight into the population slabs.
// The host keeps what has to stay serial or is cheap there: inflate, the newline scan, the site names.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

struct wgs_reader;

struct TextAllocator {
    void *(*alloc)(size_t bytes, void *user) = nullptr;    // page-locked host memory for the text buffers
    void (*release)(void *p, void *user) = nullptr;
    void *user = nullptr;
};

// One batch of complete lines.  data[0 .. len) is followed by at least TEXT_PAD bytes of '\n' (the tokeniser
// reads whole 16-byte words and one word ahead).
constexpr size_t TEXT_PAD = 64;
struct TextChunk {
    int slot = 0;                  // index of this chunk's buffers (0 .. nbuf-1)
    char *data = nullptr;
    size_t cap = 0, len = 0;
    std::vector<uint32_t> begin, end;   // per non-blank line: [begin, end) without the newline
    std::string names;                  // '\n'-terminated first tokens of those lines (site names)
    int64_t first_row = 0;              // data rows handed out before this chunk (since reader_text_start)
    double inflate_s = 0.0, scan_s = 0.0;
};

// ---- the compressed hand-over (BGZF, device-resident text): the producer only READS -- whole members into page-locked
// buffers, their offsets and sizes listed -- and the device inflates, lists the lines and tokenises (ingest.hip); the text
// never exists on the host.
struct CompChunk {
    unsigned char *comp = nullptr;      // page-locked; members back to back as in the file
    size_t cap = 0, len = 0;
    std::vector<uint64_t> in_off;       // per non-empty member: first byte of its deflate stream within comp
    std::vector<uint32_t> in_len, isize;
    size_t text_bytes = 0;              // sum of isize
    const char *pre_text = nullptr;     // first chunk only: text the line-oriented calls had already inflated (precedes the members)
    size_t pre_len = 0;
    bool last = false;                  // nothing follows
    double read_s = 0.0;
};
// a chunk ends before its text would exceed text_cap or its non-empty members max_members (0: no limit)
int reader_comp_start(wgs_reader *r, size_t comp_bytes, size_t text_cap, int nbuf, TextAllocator a, size_t max_members = 0);
int reader_comp_next(wgs_reader *r, CompChunk **out, double *waited_s);
void reader_comp_release(wgs_reader *r, CompChunk *c);
void reader_comp_stop(wgs_reader *r);
// One member through the host's inflater (for members the device rejects): comp + in_off .. -> out[isize]; false = corrupt.
bool reader_inflate_member(const unsigned char *deflate, uint32_t in_len, uint32_t isize, unsigned char *out);
void reader_add_lines_read(wgs_reader *r, int64_t rows);
int64_t reader_comp_bytes_left(wgs_reader *r);   // compressed bytes not yet inflated (-1: unknown)

// Starts the producer thread: it inflates ahead into `nbuf` buffers of `chunk_bytes` (grown when one line or one
// batch of parallel-inflated stretches needs more) and stops after `limit_rows` data rows (< 0: the whole file).
int reader_text_start(wgs_reader *r, size_t chunk_bytes, int nbuf, TextAllocator a, int64_t limit_rows);
bool reader_text_is_bgzf(const wgs_reader *r);
// Next chunk in file order (*out = nullptr at the end); blocks while the producer is still inflating it.
int reader_text_next(wgs_reader *r, TextChunk **out, double *waited_s);
void reader_text_release(wgs_reader *r, TextChunk *c);
void reader_text_stop(wgs_reader *r);
// The host parser for one line (the fallback for lines the device flags): 0 = ok, 1 = too few columns.
int reader_text_parse_line(const wgs_reader *r, const char *b, const char *e, float *out);
int reader_text_n_inds(const wgs_reader *r);
int reader_text_gl_cols(const wgs_reader *r);
int64_t reader_text_lines_read(const wgs_reader *r);
