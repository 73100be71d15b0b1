// BGZF inflate on the MI355X (RFC 1951, written from the specification).
//
// The ingest (ingest.hip) is bound by the host's inflate since the values are tokenised on the device; BGZF -- what ANGSD
// writes -- is a series of INDEPENDENT deflate streams of at most 64 KiB of output, thousands per chunk of text, so the
// streams can be decoded side by side: one LANE per block.  A deflate stream itself is serial (variable-length codes,
// back-references into its own output); a lane decodes one symbol at a time through its own tables:
//   * bits come from a 64-bit buffer refilled eight bytes at a time;
//   * literal/length codes through an 8-bit table, distance codes through a 7-bit table, both in LDS (entry = length << 9
//     | symbol, filled from the canonical code of the block's code lengths, bit-reversed because deflate packs codes LSB
//     first); the rare longer codes are found by the canonical count/first/offset walk of the remaining lengths;
//   * literals and copies are written byte by byte into the block's slot of the text buffer (a lane's consecutive bytes
//     share cache lines, L2 merges them); copies read the lane's own earlier output.
// Every block is checked: the stream must end with its final block exactly at ISIZE bytes of output and inside its input;
// anything else only marks the block, which the host then inflates itself (none for files bgzip or ANGSD wrote).
// CRC32 is not checked (neither does the host path, reader.cpp: BlockInflater).
#include <stdio.h>

#include "common.h"

namespace {

// The decode tables live in LDS, entry-major with the lane fastest ([entry][lane]: a wave-wide lookup with per-lane entries
// touches 32 banks evenly): device memory made every lookup a cache miss (tables of thousands of lanes, random entries --
// 2.3 us per symbol measured).  8 + 7 bits of primary table, the per-length counts for the canonical walk of longer codes.
constexpr int LIT_BITS = 8, DIST_BITS = 7;
constexpr int LIT_SIZE = 1 << LIT_BITS, DIST_SIZE = 1 << DIST_BITS;
constexpr int LDS_LIT = 0, LDS_DIST = LIT_SIZE * 64, LDS_LIT_COUNT = LDS_DIST + DIST_SIZE * 64, LDS_DIST_COUNT = LDS_LIT_COUNT + 16 * 64,
              LDS_WORDS = LDS_DIST_COUNT + 16 * 64;      // uint16 words per wavefront: 52 KiB

// per-lane scratch in device memory (one per block in flight): what is touched once per deflate block, not once per symbol
struct LaneTables {
    uint16_t lit_sorted[288];      // symbols ordered by (length, symbol): the canonical walk for codes beyond the table
    uint16_t dist_sorted[32];
    uint16_t offs[16], next[16];   // build_table's running offsets / next codes per length
    uint8_t lens[320];             // code lengths while a dynamic block's trees are read
    uint8_t cl[20], dl[32];        // code-length code lengths / distance code lengths
};

struct InflateArgs {
    const uint8_t *comp;           // compressed bytes of the chunk
    const uint64_t *in_off;        // per block: first byte of the raw deflate stream (member header skipped)
    const uint32_t *in_len;        // per block: bytes of the deflate stream
    const uint64_t *out_off;       // per block: where its text goes
    const uint32_t *isize;         // per block: bytes of text it must produce
    uint8_t *out;
    uint8_t *status;               // per block: 0 = ok, else the host inflates it
    LaneTables *tables;            // one per lane of the launch
    int32_t nblocks;
};

__constant__ uint8_t kClenOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct BitReader {
    const uint8_t *p, *end;
    uint64_t buf = 0;
    int cnt = 0;
    // three aligned 64-bit words of input held in registers: the one `p` points into and the two behind it.  A refill takes
    // its eight bytes from them and, when `p` moves on to the next word, requests the word after the window -- a load whose
    // result is not needed for at least eight more bytes of input, so the trips in between do not wait for memory.
    // (a pointer into GLOBAL memory by type: a generic one makes these flat loads, which count as LDS traffic too, and the table
    // lookup that follows a refill then waits for the word requested ahead)
    typedef const __attribute__((address_space(1))) uint64_t *GlobalWords;
    GlobalWords a = nullptr;
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    __device__ __forceinline__ void start(const uint8_t *first, const uint8_t *last)
    {
        p = first;
        end = last;
        a = (GlobalWords)((uintptr_t)first & ~(uintptr_t)7);
        w0 = a[0], w1 = a[1], w2 = a[2];
    }
    // At least 56 valid bits afterwards.  (The chunk is padded, so reading a little past a stream is harmless: a valid stream
    // never CONSUMES those bits, see `consumed_past_end`); bytes of a partly used word are ORed in again at the same position
    // by the next refill, which changes nothing.
    __device__ __forceinline__ void refill()
    {
        const int sh = (int)((uintptr_t)p & 7) * 8;
        const uint64_t v = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
        buf |= v << cnt;
        const int nbytes = (63 - cnt) >> 3;
        p += nbytes;
        cnt += nbytes * 8;
        if (((uintptr_t)p & ~(uintptr_t)7) != (uintptr_t)a) {     // at most one word further
            ++a;
            w0 = w1;
            w1 = w2;
            w2 = a[2];
        }
    }
    // The same for the decode loop, without a branch and in two halves.  `refill_take` (top of a trip; `on` says whether this lane
    // refills at all) moves bits from the window into `buf` and requests the word behind the window -- whether or not the window
    // moves (the same word again if not: a hit in L1), because a request that ends a conditional block is waited for at once,
    // where the compiler merges the two paths' registers.  `refill_commit` shifts the window and is the first use of the
    // requested word: the loop calls it between a copy's loads and its stores, where the wavefront waits for memory anyway and
    // the request (older than the copy's loads) has long been answered.  Used at the top of the next trip instead, it cost that
    // trip a wait for the previous trip's STORES (vmcnt counts loads and stores in one sequence on this architecture).
    uint64_t requested = 0;
    bool moves = false;
    __device__ __forceinline__ void refill_take(bool on)
    {
        const int sh = (int)((uintptr_t)p & 7) * 8;
        const uint64_t v = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
        const int nbytes = on ? (63 - cnt) >> 3 : 0;
        buf |= on ? v << cnt : 0;
        p += nbytes;
        cnt += nbytes * 8;
        moves = ((uintptr_t)p & ~(uintptr_t)7) != (uintptr_t)a;
        a += moves ? 1 : 0;
        requested = a[2];
    }
    __device__ __forceinline__ void refill_commit()
    {
        w0 = moves ? w1 : w0;
        w1 = moves ? w2 : w1;
        w2 = requested;
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    __device__ __forceinline__ void drop(int n)
    {
        buf >>= n;
        cnt -= n;
    }
    __device__ __forceinline__ uint32_t take(int n)              // n <= 32
    {
        if (cnt < n) refill();
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    __device__ __forceinline__ bool consumed_past_end() const { return p - (cnt >> 3) > end; }
    // The read pointer runs at most eight bytes ahead of the bits consumed: further than that behind the stream's end means the
    // stream has consumed input that is not its own (a damaged or truncated member) -- checked once per trip and per symbol of
    // a header, so that a lane never reads more than a few words past its stream (the chunk is padded by 128 zero bytes).
    __device__ __forceinline__ bool ran_past_end() const { return p > end + 8; }
};

__device__ __forceinline__ uint32_t reverse_bits(uint32_t code, int len) { return __brev(code) >> (32 - len); }

// Builds table (primary bits `bits`), sorted symbols and counts from n code lengths.  Returns false for an over-subscribed code.
// (All arrays live in the lane's scratch in device memory and no loop is unrolled: registers buy nothing in code that waits
// for its own previous load, occupancy does.)
// `table` and `count` are this lane's columns of the LDS arrays: element i at [i * 64].
__device__ __noinline__ bool build_table(const uint8_t *lens, int n, uint16_t *table, int bits, uint16_t *sorted, uint16_t *count,
                                         uint16_t *offs, uint16_t *next)
{
#pragma nounroll
    for (int i = 0; i < 16; ++i) count[i * 64] = 0;
#pragma nounroll
    for (int i = 0; i < n; ++i) ++count[lens[i] * 64];
    count[0] = 0;
    int left = 1;
#pragma nounroll
    for (int len = 1; len < 16; ++len) {
        left = (left << 1) - count[len * 64];
        if (left < 0) return false;
    }
    offs[1] = 0;
    uint32_t code = 0;
    next[0] = 0;
#pragma nounroll
    for (int len = 1; len < 16; ++len) {
        if (len > 1) offs[len] = offs[len - 1] + count[(len - 1) * 64];
        code = (code + count[(len - 1) * 64]) << 1;
        next[len] = (uint16_t)code;
    }
    const int size = 1 << bits;
#pragma nounroll
    for (int i = 0; i < size; ++i) table[i * 64] = 0;
#pragma nounroll
    for (int s = 0; s < n; ++s) {
        const int len = lens[s];
        if (!len) continue;
        sorted[offs[len]++] = (uint16_t)s;
        const uint32_t c = next[len]++;
        if (len <= bits) {
            const uint32_t r = reverse_bits(c, len);
            const uint16_t e = (uint16_t)((len << 9) | s);
#pragma nounroll
            for (uint32_t i = r; i < (uint32_t)size; i += 1u << len) table[i * 64] = e;
        }
    }
    return true;
}

// The canonical walk (RFC 1951 3.2.2) for a code the table does not hold: (length << 9) | symbol, or 0 for an invalid code.
// (Inlined into the decode loop: a call makes the compiler wait for every load in flight, the input word requested ahead of
// its use among them.)
__device__ __forceinline__ uint32_t walk_symbol_inline(uint64_t b, const uint16_t *sorted, const uint16_t *count)
{
    int code = 0, first = 0, index = 0;
#pragma nounroll
    for (int len = 1; len < 16; ++len) {
        code |= (int)(b & 1);
        b >>= 1;
        const int c = count[len * 64];
        if (code - c < first) return ((uint32_t)len << 9) | sorted[index + (code - first)];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return 0;
}
__device__ __noinline__ uint32_t walk_symbol(uint64_t b, const uint16_t *sorted, const uint16_t *count)
{
    return walk_symbol_inline(b, sorted, count);
}

// Sixteen bytes at any alignment.
typedef uint32_t Bytes16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Bytes16 load16(const uint8_t *p)
{
    Bytes16 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
__device__ __forceinline__ void store16(uint8_t *p, Bytes16 v) { __builtin_memcpy(p, &v, 16); }
// One symbol where lockstep does not matter (the trees of a dynamic block).  -1 = invalid code.
__device__ __forceinline__ int decode_symbol(BitReader &br, const uint16_t *table, int bits, const uint16_t *sorted, const uint16_t *count)
{
    if (br.cnt < 15) br.refill();
    uint32_t e = table[br.peek(bits) * 64];
    if (!e) e = walk_symbol(br.buf, sorted, count);
    if (!e) return -1;
    br.drop((int)(e >> 9));
    return (int)(e & 511);
}

// Block header (and, for a dynamic block, its trees).  Returns the block type 0 / 1 / 2, or -1 for a corrupt header; for a
// stored block *stored = its length.
__device__ __noinline__ int read_block_header(BitReader &br, LaneTables &T, uint16_t *L, uint32_t *last, uint32_t *stored)
{
    uint16_t *const lit = L + LDS_LIT, *const dtab = L + LDS_DIST, *const lit_count = L + LDS_LIT_COUNT, *const dist_count = L + LDS_DIST_COUNT;
    br.refill();
    *last = br.take(1);
    const uint32_t type = br.take(2);
    if (type == 0) {
        br.drop(br.cnt & 7);
        const uint32_t len = br.take(16), nlen = br.take(16);
        if ((len ^ 0xFFFFu) != nlen) return -1;
        *stored = len;
        return 0;
    }
    if (type == 1) {
#pragma nounroll
        for (int i = 0; i < 288; ++i) T.lens[i] = i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8));
        if (!build_table(T.lens, 288, lit, LIT_BITS, T.lit_sorted, lit_count, T.offs, T.next)) return -1;
#pragma nounroll
        for (int i = 0; i < 30; ++i) T.lens[i] = 5;
        return build_table(T.lens, 30, dtab, DIST_BITS, T.dist_sorted, dist_count, T.offs, T.next) ? 1 : -1;
    }
    if (type != 2) return -1;
    const int hlit = (int)br.take(5) + 257, hdist = (int)br.take(5) + 1, hclen = (int)br.take(4) + 4;
    if (hlit > 286 || hdist > 30) return -1;
    uint8_t *cl = T.cl;
#pragma nounroll
    for (int i = 0; i < 19; ++i) cl[i] = 0;
#pragma nounroll
    for (int i = 0; i < hclen; ++i) cl[kClenOrder[i]] = (uint8_t)br.take(3);
    // the code-length code (at most 7 bits) goes through the literal table's storage (rebuilt right after)
    static_assert(LIT_BITS >= 7, "the code-length code needs a 7-bit table");
    if (!build_table(cl, 19, lit, 7, T.dist_sorted, dist_count, T.offs, T.next)) return -1;
    int i = 0;
    while (i < hlit + hdist) {
        const int sym = decode_symbol(br, lit, 7, T.dist_sorted, dist_count);
        if (sym < 0 || br.ran_past_end()) return -1;
        if (sym < 16) {
            T.lens[i++] = (uint8_t)sym;
            continue;
        }
        int rep, val = 0;
        if (sym == 16) {
            if (i == 0) return -1;
            val = T.lens[i - 1];
            rep = 3 + (int)br.take(2);
        } else if (sym == 17) {
            rep = 3 + (int)br.take(3);
        } else {
            rep = 11 + (int)br.take(7);
        }
        if (i + rep > hlit + hdist) return -1;
#pragma nounroll
        while (rep--) T.lens[i++] = (uint8_t)val;
    }
    if (T.lens[256] == 0) return -1;
    // distance lengths follow the literal/length lengths: move them out before the tables are built over `lens`
    uint8_t *dl = T.dl;
#pragma nounroll
    for (int d = 0; d < 30; ++d) dl[d] = d < hdist ? T.lens[hlit + d] : 0;
    if (!build_table(T.lens, hlit, lit, LIT_BITS, T.lit_sorted, lit_count, T.offs, T.next)) return -1;
    return build_table(dl, 30, dtab, DIST_BITS, T.dist_sorted, dist_count, T.offs, T.next) ? 2 : -1;
}

// One lane per BGZF block, the 64 lanes of a wavefront in LOCKSTEP: every trip of the loop emits one step per lane
// -- a literal, up to sixteen bytes of a copy, one byte of a stored block -- so lanes that are decoding symbols execute the same
// instructions together, and so do the lanes that are copying.  (The first version let every lane run its own
// symbol-by-symbol loop: 64 different instruction streams per wavefront, 5 GB/s.)  Only block headers -- a few per 64 KiB --
// leave the common path, and the lanes that reach one at the same trip build their tables together.
enum : int { ST_HEADER = 0, ST_SYMBOL = 1, ST_COPY = 2, ST_STORED = 3, ST_DONE = 4, ST_WALK_LIT = 5, ST_WALK_DIST = 6 };

#ifdef WGS_INFLATE_STATS
// (experiments: -DWGS_INFLATE_STATS adds up, per launch and over all wavefronts: [0] lockstep trips, trips in which some lane [1] read
// a block header, [2] [3] walked a long literal/length / distance code, [4] copied ([5] a far piece), [6] decoded a symbol; [7] the
// most trips of any wavefront; [8] lanes at work summed over trips; clock cycles [9] in all, [10] in block headers, [11] from
// the refill to the memory phase, [12] waiting for memory, [13] in the memory phase behind the wait.  Counted in registers, one
// atomic per wavefront and counter at the end.)
__device__ unsigned long long g_inflate_stats[16];
#define INFLATE_STAT(i, cond) do { if (__any(cond)) ++stat[i]; } while (0)
#define INFLATE_CLOCK(i) do { const unsigned long long now_ = clock64(); stat[i] += now_ - mark; mark = now_; } while (0)
#else
#define INFLATE_STAT(i, cond) do { } while (0)
#define INFLATE_CLOCK(i) do { } while (0)
#endif

__global__ __launch_bounds__(64) void inflate_kernel(InflateArgs A)
{
    __shared__ uint16_t lds[LDS_WORDS];
    uint16_t *const L = lds + threadIdx.x;                     // this lane's column of every table
    const int blk = (int)blockIdx.x * 64 + (int)threadIdx.x;
    const bool have = blk < A.nblocks;
    LaneTables &T = A.tables[have ? blk : 0];
    BitReader br;
    {
        const uint8_t *first = A.comp + (have ? A.in_off[blk] : 0);
        br.start(first, first + (have ? A.in_len[blk] : 0));
    }
    uint8_t *const out0 = A.out + (have ? A.out_off[blk] : 0);
    const uint32_t want = have ? A.isize[blk] : 0;
    uint32_t pos = 0, last = 0, left = 0, dist = 0, done = 0, trip = 0;
    int state = have ? ST_HEADER : ST_DONE;
    bool bad = false;
    // the piece of a match whose bytes were REQUESTED in the previous trip and are written in this one (see "memory phase")
    Bytes16 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0, v4 = v0;
    uint32_t one = 0, due_at = 0, due_bytes = 0;
    bool due_far = false, due_near = false, due_byte = false;
#ifdef WGS_INFLATE_STATS
    unsigned long long stat[16] = {0}, mark = clock64();
    const unsigned long long started = mark;
#endif
    while (__any(state != ST_DONE || due_far || due_near || due_byte)) {
        INFLATE_STAT(0, true);
        INFLATE_STAT(1, state == ST_HEADER);
#ifdef WGS_INFLATE_STATS
        stat[8] += (unsigned long long)__popcll(__ballot(state != ST_DONE));
        mark = clock64();
#endif
        if (state == ST_HEADER) {
            // (the header code is a call and takes the reader by reference: on a copy, so that the loop's own reader, `last` and
            // `stored` stay in registers -- handed over directly they lived in scratch, and every trip of the loop went through
            // a dozen dependent scratch loads and waited for the "prefetched" input word in order to store it there)
            uint32_t stored = 0, final_block = 0;
            BitReader hr = br;
            const int type = read_block_header(hr, T, L, &final_block, &stored);
            br = hr;
            last = final_block;
            // (everything read back from the copy has arrived before the paths join again: otherwise the compiler, which counts
            // outstanding memory operations per path and assumes the worst at a join, makes EVERY trip wait for most of the
            // previous trip's stores at its first use of one of these registers)
            __builtin_amdgcn_s_waitcnt(0);
            if (type < 0 || (type == 0 && pos + stored > want)) {
                bad = true;
                state = ST_DONE;
            } else if (type == 0) {
                left = stored;
                state = stored ? ST_STORED : (last ? ST_DONE : ST_HEADER);
            } else {
                state = ST_SYMBOL;
            }
        }
        INFLATE_CLOCK(10);
        br.refill_take(state != ST_DONE && br.cnt < 48);
        if (state != ST_DONE && br.ran_past_end()) {
            // input that belongs to the next member (or to nobody): a damaged stream -- e.g. empty stored blocks or end-of-block
            // codes that never set the final bit -- would otherwise decode on through the following members and past the chunk
            bad = true;
            state = ST_DONE;
        }
        // Codes longer than the tables' 8 / 7 bits go through the canonical walk: fifteen dependent LDS reads and a load from the
        // lane's scratch in device memory.  One lane in 140 needs it per trip -- but with 64 lanes in lockstep that was a walk in 37 %
        // (literal/length) and 34 % (distance) of all trips, for which the other lanes stood still: half of the kernel's time
        // (-DWGS_INFLATE_STATS counts them).  A lane that meets a long code now WAITS (ST_WALK_*), and the wavefront walks for all
        // waiting lanes at once: when a dozen have gathered, when they are a quarter of the lanes still decoding, or every sixteenth trip.
        const unsigned long long waiting = __ballot(state == ST_WALK_LIT || state == ST_WALK_DIST);
        const int n_wait = __popcll(waiting), n_active = __popcll(__ballot(state != ST_DONE));
        const bool do_walk = n_wait > 0 && (n_wait >= 12 || 4 * n_wait >= n_active || (trip & 15u) == 15u);
        ++trip;
        INFLATE_STAT(6, state == ST_SYMBOL);
        uint32_t e = 0, literal_at = 0, literal_byte = 0;
        bool have_e = false, literal = false;
        if (state == ST_SYMBOL) {
            e = L[LDS_LIT + br.peek(LIT_BITS) * 64];
            if (e) have_e = true;
            else state = ST_WALK_LIT;
        }
        INFLATE_STAT(2, do_walk && state == ST_WALK_LIT);
        if (do_walk && state == ST_WALK_LIT) {
            e = walk_symbol_inline(br.buf, T.lit_sorted, L + LDS_LIT_COUNT);
            have_e = true;
            state = ST_SYMBOL;
        }
        uint32_t d = 0;
        bool have_d = false;
        if (have_e) {
            const uint32_t sym = e & 511;
            br.drop((int)(e >> 9));
            if (!e || sym > 285) {
                bad = true;
                state = ST_DONE;
            } else if (sym < 256) {
                if (pos < want) literal = true, literal_at = pos++, literal_byte = sym;       // (written in the memory phase)
                else bad = true, state = ST_DONE;
            } else if (sym == 256) {
                state = last ? ST_DONE : ST_HEADER;
            } else {
                // base and extra bits of the length code by arithmetic (RFC 1951 3.2.5: four codes per number of extra bits);
                // a table in constant memory would be two more dependent loads per match
                const int li = (int)sym - 257;
                const int xb = li < 8 || li == 28 ? 0 : (li >> 2) - 1;
                const uint32_t lbase = li < 8 ? 3u + (uint32_t)li : li == 28 ? 258u : ((4u + ((uint32_t)li & 3u)) << xb) + 3u;
                left = lbase + br.peek(xb);
                br.drop(xb);
                d = L[LDS_DIST + br.peek(DIST_BITS) * 64];
                if (d) have_d = true;
                else state = ST_WALK_DIST;                 // (the match's length waits in `left`)
            }
        }
        INFLATE_STAT(3, do_walk && state == ST_WALK_DIST);
        if (do_walk && state == ST_WALK_DIST) {
            d = walk_symbol_inline(br.buf, T.dist_sorted, L + LDS_DIST_COUNT);
            have_d = true;
            state = ST_SYMBOL;
        }
        if (have_d) {
            const uint32_t ds = d & 511;
            br.drop((int)(d >> 9));
            if (!d || ds > 29) {
                bad = true;
                state = ST_DONE;
            } else {
                const int db = ds < 4 ? 0 : (int)(ds >> 1) - 1;           // two distance codes per number of extra bits
                dist = (ds < 4 ? 1u + ds : ((2u + (ds & 1u)) << db) + 1u) + br.peek(db);
                br.drop(db);
                done = 0;
                if (dist > pos || pos + left > want) bad = true, state = ST_DONE;
                else state = ST_COPY;
            }
        }
        // (a lane that has just decoded a match moves its first bytes in the same trip -- a match is one trip shorter, and the wavefront
        // runs this code in every trip anyway, for the lanes that are in the middle of a copy)
        INFLATE_STAT(4, state == ST_COPY);
        INFLATE_STAT(5, state == ST_COPY && dist >= 64 && left > 16);
        // matches in this kind of text are long (51 bytes on average) and come from far back (half of them from more than 2 KiB),
        // or repeat a short period ("0.333333\t" three times per unobserved genotype: distance 9).  Three ways to move a piece:
        //   far    a long match from far back: sixty-four bytes in one trip (four loads in flight, then four stores);
        //   near   sixteen bytes whatever the distance.  A source that overlaps the destination repeats with period dist: of the
        //          bytes behind `pos`, dist + done follow that period (done = what this match has written so far), so the source
        //          may lie any multiple of dist back within them -- the largest one below sixteen plus one more if it fits: once
        //          sixteen periodic bytes exist every trip moves sixteen (a distance of 9: 9, then 16 per trip instead of 9 every
        //          time).  What lies past the valid bytes is overwritten by what follows;
        //   byte   one byte, within sixteen bytes of the member's end.
        //
        // The memory phase.  A lane's decoding depends on its input alone, never on what it has written; and the bytes of a match
        // come from DRAM as often as not (64 lanes x 32 KiB of history per wavefront, 600 MB for a chunk: no cache holds that).
        // So a piece's bytes are requested at the end of one trip and written at the end of the NEXT, with that trip's decoding
        // in between instead of a wait: per trip one wait for memory, which by then has had a whole trip's time.  In program
        // order -- which is all a lane's own loads and stores need (a load sees the lane's earlier stores without waiting for
        // them) -- every piece is still written before anything that follows it is written or read:
        //   1. wait for what the previous trip requested (the piece and the reader's next word),
        //   2. write that piece, then this trip's literal (which may lie in the bytes a piece writes past its end),
        //   3. request the next piece.
        // The requests are inline assembly for one reason: the registers stay the variables' own.  A load in a conditional block
        // that the compiler knows about is followed, at the block's end, by a copy into the merged variable's register -- and by
        // a wait for the load at that point.  (The compiler's own waits elsewhere stay safe: they may count too few operations
        // in flight, never too many.)
        INFLATE_CLOCK(11);
        __builtin_amdgcn_s_waitcnt(0x0F70);                                  // vmcnt(0), nothing else
        INFLATE_CLOCK(12);
        br.refill_commit();
        if (due_far) {
            uint8_t *dst = out0 + due_at;
            store16(dst, v0), store16(dst + 16, v1);
            if (due_bytes > 32) store16(dst + 32, v2);
            if (due_bytes > 48) store16(dst + 48, v3);
        }
        if (due_near) store16(out0 + due_at, v4);
        if (due_byte) out0[due_at] = (uint8_t)one;
        if (literal) out0[literal_at] = (uint8_t)literal_byte;
        const bool copying = state == ST_COPY;
        const bool far = copying && dist >= 64 && left > 16 && pos + 64 <= want;
        const bool near = copying && !far && pos + 16 <= want;
        const bool bytewise = copying && !far && !near;
        uint32_t back = dist;
        if (near && dist < 16) {
            const uint32_t have_bytes = dist + done;
            const uint32_t k = (15u + dist) / dist;                 // periods that cover sixteen bytes
            back = k * dist <= have_bytes ? k * dist : (have_bytes / dist) * dist;
        }
        if (far) {
            // (the third and fourth sixteen bytes only in the lanes whose piece has them: what a divergent access costs the
            // compute unit's address unit -- the kernel's bottleneck once three wavefronts share one -- goes by the lanes in it)
            const uint8_t *src = out0 + pos - dist;
            asm volatile("global_load_dwordx4 %0, %2, off\n\t"
                         "global_load_dwordx4 %1, %2, off offset:16"
                         : "+v"(v0), "+v"(v1)
                         : "v"(src)
                         : "memory");
            if (left > 32) asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "+v"(v2) : "v"(src) : "memory");
            if (left > 48) asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "+v"(v3) : "v"(src) : "memory");
        }
        if (near) {
            const uint8_t *src = out0 + pos - back;
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(v4) : "v"(src) : "memory");
        }
        if (bytewise) {
            const uint8_t *src = out0 + pos - dist;
            asm volatile("global_load_ubyte %0, %1, off" : "+v"(one) : "v"(src) : "memory");
        }
        due_far = far, due_near = near, due_byte = bytewise, due_at = pos, due_bytes = left;
        if (copying) {
            uint32_t step = far ? 64u : near ? (back < 16 ? back : 16u) : 1u;
            step = left < step ? left : step;
            pos += step;
            left -= step;
            done += step;
            if (left == 0) state = ST_SYMBOL;
        } else if (state == ST_STORED) {
            out0[pos++] = (uint8_t)br.take(8);
            if (--left == 0) state = last ? ST_DONE : ST_HEADER;
        }
        INFLATE_CLOCK(13);
    }
#ifdef WGS_INFLATE_STATS
    stat[9] = clock64() - started;
    if (threadIdx.x == 0) {
        for (int i = 0; i < 16; ++i)
            if (i != 7 && stat[i]) atomicAdd(&g_inflate_stats[i], stat[i]);
        atomicMax(&g_inflate_stats[7], stat[0]);
    }
#endif
    if (have) A.status[blk] = (bad || pos != want || br.consumed_past_end()) ? 1 : 0;
}

}  // namespace

size_t inflate_table_bytes(void) { return sizeof(LaneTables); }

// Inflates nblocks BGZF deflate streams of d_comp into d_out (device pointers; tables: nblocks * inflate_table_bytes()).
int launch_inflate(wgs_ctx *ctx, const uint8_t *d_comp, const uint64_t *d_in_off, const uint32_t *d_in_len, const uint64_t *d_out_off,
                   const uint32_t *d_isize, uint8_t *d_out, uint8_t *d_status, void *d_tables, int32_t nblocks)
{
    if (nblocks <= 0) return 0;
    InflateArgs A;
    A.comp = d_comp;
    A.in_off = d_in_off;
    A.in_len = d_in_len;
    A.out_off = d_out_off;
    A.isize = d_isize;
    A.out = d_out;
    A.status = d_status;
    A.tables = reinterpret_cast<LaneTables *>(d_tables);
    A.nblocks = nblocks;
    hipLaunchKernelGGL(inflate_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, ctx->stream, A);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" {

/* Test hook and building block: inflate `nblocks` raw deflate streams (BGZF members without header and trailer) that lie in
 * one host buffer `comp` (in_off / in_len) into `out` (out_off / isize), on the device; status[i] != 0 marks a stream the
 * device did not accept.  *kernel_ms: the inflate kernel alone. */
int wgs_debug_inflate(wgs_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, const uint64_t *in_off, const uint32_t *in_len,
                      const uint64_t *out_off, const uint32_t *isize, int32_t nblocks, uint8_t *out, int64_t out_bytes, uint8_t *status,
                      float *kernel_ms)
{
    WGS_REQUIRE(ctx && comp && in_off && in_len && out_off && isize && out && status && nblocks >= 0, "bad argument");
    for (int32_t i = 0; i < nblocks; ++i)
        WGS_REQUIRE(in_off[i] + in_len[i] <= (uint64_t)comp_bytes && out_off[i] + isize[i] <= (uint64_t)out_bytes, "stream %d outside its buffer", i);
    if (nblocks == 0) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    uint8_t *d_comp = nullptr, *d_out = nullptr, *d_status = nullptr;
    uint64_t *d_io = nullptr, *d_oo = nullptr;
    uint32_t *d_il = nullptr, *d_is = nullptr;
    void *d_tab = nullptr;
    auto guard = on_failure([&] {
        for (void *p : {(void *)d_comp, (void *)d_out, (void *)d_status, (void *)d_io, (void *)d_oo, (void *)d_il, (void *)d_is, d_tab})
            if (p) (void)hipFree(p);
    });
    HIP_TRY(wgs_malloc(&d_comp, (size_t)comp_bytes + 128));
    HIP_TRY(hipMemset(d_comp + comp_bytes, 0, 128));
    HIP_TRY(wgs_malloc(&d_out, (size_t)std::max<int64_t>(out_bytes, 1)));
    HIP_TRY(wgs_malloc(&d_status, (size_t)nblocks));
    HIP_TRY(wgs_malloc(&d_io, sizeof(uint64_t) * nblocks));
    HIP_TRY(wgs_malloc(&d_oo, sizeof(uint64_t) * nblocks));
    HIP_TRY(wgs_malloc(&d_il, sizeof(uint32_t) * nblocks));
    HIP_TRY(wgs_malloc(&d_is, sizeof(uint32_t) * nblocks));
    HIP_TRY(wgs_malloc(&d_tab, inflate_table_bytes() * (size_t)nblocks));
    HIP_TRY(hipMemcpy(d_comp, comp, (size_t)comp_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_io, in_off, sizeof(uint64_t) * nblocks, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_oo, out_off, sizeof(uint64_t) * nblocks, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_il, in_len, sizeof(uint32_t) * nblocks, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_is, isize, sizeof(uint32_t) * nblocks, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)std::max<int64_t>(out_bytes, 1), ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    if (launch_inflate(ctx, d_comp, d_io, d_il, d_oo, d_is, d_out, d_status, d_tab, nblocks)) return 1;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(status, d_status, (size_t)nblocks, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1);
#ifdef WGS_INFLATE_STATS
    {
        unsigned long long st[16] = {0};
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_inflate_stats), sizeof st);
        const int waves = (nblocks + 63) / 64;
        fprintf(stderr, "[inflate stats] %d wavefronts, %llu trips (most in one: %llu), lanes at work per trip %.1f; trips with a header %llu, a long "
                "literal/length walk %llu, a long distance walk %llu, a copy %llu (far %llu), a symbol %llu\n", waves, st[0], st[7],
                st[0] ? (double)st[8] / (double)st[0] : 0.0, st[1], st[2], st[3], st[4], st[5], st[6]);
        fprintf(stderr, "[inflate stats] cycles per wavefront %.0f: headers %.0f, decode %.0f, waiting for memory %.0f, memory phase %.0f; per trip: decode "
                "%.0f, wait %.0f, memory phase %.0f\n", (double)st[9] / waves, (double)st[10] / waves, (double)st[11] / waves, (double)st[12] / waves,
                (double)st[13] / waves, (double)st[11] / st[0], (double)st[12] / st[0], (double)st[13] / st[0]);
        unsigned long long zero[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_inflate_stats), zero, sizeof zero);
    }
#endif
    guard.dismiss();
    for (void *p : {(void *)d_comp, (void *)d_out, (void *)d_status, (void *)d_io, (void *)d_oo, (void *)d_il, (void *)d_is, d_tab}) (void)hipFree(p);
    return 0;
}

}  // extern "C"
