// inflate_lane.h -- Deflate decode, one compressed stream per lane.
//
// Replaces the body of Deflate.decompress(_ bitReader:) (reference Sources/Deflate/Deflate.swift:30-249)
// together with Code.huffmanCodes (Sources/Common/CodingTree/Code.swift:15-39) and DecodingTree
// (Sources/Common/CodingTree/DecodingTree.swift:15-50).  Semantics -- including the reference's
// acceptance of incomplete and over-subscribed Huffman sets -- are preserved;
// the data structures are not: instead of a 2^(maxBits+1) heap walked one bit per step, each lane
// keeps a compact canonical description of its two alphabets in LDS (152 words per lane, interleaved
// at wave stride so that arbitrary per-lane indices never bank-conflict) and the 15 left-justified
// code limits of each alphabet in VGPRs:
//
//   slot[d] (d = 1..15)   = v0[d] << 9 | start[d]   v0 = counter value of the first code of length d
//                                                    (Code.swift's `symbol` after the shift), start =
//                                                    index of that code in the (length, symbol)-sorted
//                                                    symbol array; slot[16] = number of codes
//   lim[d]                = (first[d] + count[d]) << (15 - d), non-decreasing in d
//
// Fast path (set not over-subscribed): code length = 1 + #{d : c15 >= lim[d]} where c15 is the next 15
// stream bits, bit-reversed (first bit = MSB); sorted index = start[len] + (c15 >> (15-len)) - v0[len].
// Exact path (over-subscribed sets; also used for the 19-symbol code-length alphabet): for
// d = 1..15 the heap node at depth d on the path c15 is occupied iff
// k0 = ((c15 >> (15-d)) - v0[d]) mod 2^d < count[d]; the last writer among k0, k0+2^d, ... wins
// (DecodingTree.swift:22-32), the shallowest occupied node wins (:45).
//
// The dynamic header is decoded twice (count pass, then scatter pass) so that no per-symbol length
// array has to be kept: 152 words/lane = 38 KiB per wave => 4 waves (256 streams) resident per CU.
#ifndef SWC_INFLATE_LANE_H
#define SWC_INFLATE_LANE_H

#include "swc_common.h"
#include "lz_resolve.h"

namespace swc {
namespace inflate {

constexpr int W_LIT_SYM = 0;     // 96 words : 288 x 9-bit symbols, three per word
constexpr int W_DIST_SYM = 96;   //  8 words : 32 x 8-bit symbols
constexpr int W_LIT_LEN = 104;   // 17 words : slot[0..16]
constexpr int W_DIST_LEN = 121;  // 17 words
constexpr int W_CL_SYM = 138;    //  5 words : 19 x 8-bit symbols
constexpr int W_CL_LEN = 143;    //  9 words : slot[0..8]
constexpr int kWordsPerLane = 152;
constexpr int kLdsBytesPerWave = kWordsPerLane * 4 * kWave;  // 38,912 B

// LSB-first bit reader (BitByteData.LsbBitReader contract) with a 64-bit window
// and one dword of read-ahead so the HBM/L2 latency of the next refill is hidden behind decode work.
struct BitReader {
    gcptr in;
    uint32_t len;    // stream length in bytes
    uint32_t ppos;   // byte offset of the next dword to prefetch
    uint32_t nextw;  // prefetched dword (nextn valid bytes)
    uint32_t nextn;
    uint64_t bb;     // window, bit 0 = next stream bit; bits >= bc are zero
    uint32_t bc;     // REAL bits in the window (never counts padding past the end)

    SWC_HD void prefetch() {
        uint32_t rem = len - ppos;
        if (rem >= 4) {
            nextw = load_u32(in + ppos);
            nextn = 4;
        } else {
            uint32_t w = 0;
            for (uint32_t i = 0; i < rem; i++) w |= (uint32_t)in[ppos + i] << (8 * i);
            nextw = w;
            nextn = rem;
        }
        ppos += nextn;
    }
    SWC_HD void init(gcptr p, uint32_t n, uint32_t byte_off) {
        in = p; len = n; ppos = byte_off; bb = 0; bc = 0;
        prefetch();
        refill();
    }
    // after refill(): bc >= 33 unless the stream has fewer bits left
    SWC_HD void refill() {
        if (bc <= 32) {
            bb |= (uint64_t)nextw << bc;
            bc += nextn * 8;
            prefetch();
        }
    }
    // refill() for the interior of the stream: the caller guarantees ppos + 4 <= len, so nextn == 4
    SWC_HD void refill_fast() {
        if (bc <= 32) {
            bb |= (uint64_t)nextw << bc;
            bc += 32;
            nextw = load_u32(in + ppos);
            ppos += 4;
        }
    }
    SWC_HD uint32_t peek32() const { return (uint32_t)bb; }
    SWC_HD void consume(uint32_t n) { bb >>= n; bc -= n; }
    SWC_HD uint32_t bits(uint32_t n) {  // n <= 16, caller checked bc >= n
        uint32_t v = (uint32_t)bb & ((1u << n) - 1);
        consume(n);
        return v;
    }
    // bits consumed from the start of the stream
    SWC_HD uint64_t consumed_bits() const { return (uint64_t)(ppos - nextn) * 8 - bc; }
};

struct Limits {
    uint32_t lim[16];  // [1..15]; fully unrolled users keep this in VGPRs
    bool oversub;
};

SWC_HD uint32_t lit_sym(const LaneLds& l, uint32_t i) {
    uint32_t q = (i * 171u) >> 9;  // i / 3 for i < 512
    return (l.get(W_LIT_SYM + q) >> (9 * (i - 3 * q))) & 511u;
}
SWC_HD void set_lit_sym(const LaneLds& l, uint32_t i, uint32_t s) {
    uint32_t q = (i * 171u) >> 9, sh = 9 * (i - 3 * q);
    uint32_t w = l.get(W_LIT_SYM + q);
    l.set(W_LIT_SYM + q, (w & ~(511u << sh)) | (s << sh));
}
SWC_HD uint32_t byte_sym(const LaneLds& l, int base, uint32_t i) { return (l.get(base + (i >> 2)) >> (8 * (i & 3))) & 255u; }
SWC_HD void set_byte_sym(const LaneLds& l, int base, uint32_t i, uint32_t s) {
    uint32_t sh = 8 * (i & 3);
    uint32_t w = l.get(base + (i >> 2));
    l.set(base + (i >> 2), (w & ~(255u << sh)) | (s << sh));
}

// Exact heap-equivalent lookup.  Returns the index into the sorted symbol array or -1.
template <int MAXD>
SWC_HD int lookup_exact(const LaneLds& l, int wlen, uint32_t c15, uint32_t& len) {
    uint32_t w = l.get(wlen + 1);
    for (int d = 1; d <= MAXD; d++) {
        uint32_t wn = l.get(wlen + d + 1);
        uint32_t cnt = (wn & 511u) - (w & 511u);
        uint32_t k0 = ((c15 >> (15 - d)) - (w >> 9)) & ((1u << d) - 1u);
        if (k0 < cnt) {
            len = (uint32_t)d;
            return (int)((w & 511u) + k0 + (((cnt - 1u - k0) >> d) << d));
        }
        w = wn;
    }
    return -1;
}

// slot[1..MAXD] hold per-length COUNTS on entry; on exit slot[d] = v0 << 9 | running index (= start[d])
// ready for the scatter pass.  Code.swift:23-37 restated per length.
template <int MAXD>
SWC_HD void counts_to_slots(const LaneLds& l, int wlen, Limits* lm) {
    uint32_t v = 0, off = 0;
    bool over = false;
#pragma unroll
    for (int d = 1; d <= MAXD; d++) {
        uint32_t cnt = l.get(wlen + d);
        if (lm) lm->lim[d] = (v + cnt) << (15 - d);
        if (cnt != 0 && v + cnt > (1u << d)) over = true;
        l.set(wlen + d, ((v & 0x7FFFu) << 9) | off);
        off += cnt;
        v = (v + cnt) << 1;
    }
    if (lm) lm->oversub = over;
}
// after the scatter pass slot[d].low == start[d+1]; shift back so slot[d].low == start[d], slot[MAXD+1] = total
template <int MAXD>
SWC_HD void fixup_slots(const LaneLds& l, int wlen) {
    uint32_t total = l.get(wlen + MAXD) & 511u;
    for (int d = MAXD; d >= 1; d--) {
        uint32_t prev = d > 1 ? (l.get(wlen + d - 1) & 511u) : 0u;
        l.set(wlen + d, (l.get(wlen + d) & ~511u) | prev);
    }
    l.set(wlen + MAXD + 1, total);
}
template <int MAXD>
SWC_HD void clear_slots(const LaneLds& l, int wlen) {
    for (int d = 0; d <= MAXD + 1; d++) l.set(wlen + d, 0);
}

// Phase 1 of the two-phase Deflate path (see lz_resolve.h): literals go straight to their final position,
// every match becomes one record of the stream's record list; the output buffer is never read here.
struct Lane {
    LaneLds l;
    BitReader br;
    Limits lit, dist;
    gptr out;
    uint64_t cap;
    uint64_t pos;  // bytes produced (keeps counting past cap: size pass for SWC_E_CAPACITY)
    SWC_AS_GLOBAL uint32_t* recs;  // record list in the HBM workspace
    uint32_t nrec, max_rec;
    uint64_t last_end;             // position just past the previous record

    SWC_HD void push(uint32_t v) {
        if (nrec < max_rec) recs[nrec] = v;
        nrec++;
    }
    // Deflate.swift:216-232 deferred: record (literal run, length, distance); phase 2 executes the copy.
    SWC_HD void emit_match(uint32_t length, uint32_t distance) {
        if (pos < cap) {  // records exist only for matches that start below the capacity
            uint64_t run = pos - last_end;
            while (run >= 255) {
                uint32_t s = run > lzr::kMaxSkip ? lzr::kMaxSkip : (uint32_t)run;
                push(lzr::kSkipFlag | s);
                run -= s;
            }
            push(lzr::make_match((uint32_t)run, length, distance));
            last_end = pos + length;
        }
        pos += length;
    }

    // Decode one symbol of the lit/len (LIT=true) or distance alphabet.  Returns the symbol or
    // -1 (DeflateError.symbolNotFound: unassigned path, or the code runs past the end of input).
    template <bool LIT, bool CHECKED = true>
    SWC_HD int decode_sym() {
        const Limits& lm = LIT ? lit : dist;
        const int wlen = LIT ? W_LIT_LEN : W_DIST_LEN;
        uint32_t c15 = brev32(br.peek32()) >> 17;
        uint32_t len;
        int idx;
        if (!CHECKED || !lm.oversub) {
            len = 1;
#pragma unroll
            for (int d = 1; d <= 15; d++) len += (c15 >= lm.lim[d]) ? 1u : 0u;
            if (len > 15) return -1;
            uint32_t w = l.get(wlen + (int)len);
            idx = (int)((w & 511u) + (c15 >> (15 - len)) - (w >> 9));
        } else {
            idx = lookup_exact<15>(l, wlen, c15, len);
            if (idx < 0) return -1;
        }
        if (CHECKED && len > br.bc) return -1;  // DecodingTree.swift:39 -- ran out of bits before reaching a leaf
        br.consume(len);
        return LIT ? (int)lit_sym(l, (uint32_t)idx) : (int)byte_sym(l, W_DIST_SYM, (uint32_t)idx);
    }

    // Walk the code-length section of a dynamic header (Deflate.swift:117-162).  PASS2 = false counts
    // codes per length into slot[]; PASS2 = true scatters symbols into the sorted arrays.
    template <bool PASS2>
    SWC_HD int scan_lengths(int literals, int total) {
        int n = 0;
        uint32_t prev = 0;
        while (n < total) {
            br.refill();
            uint32_t c15 = brev32(br.peek32()) >> 17, len;
            int idx = lookup_exact<7>(l, W_CL_LEN, c15, len);
            if (idx < 0 || len > br.bc) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :122
            br.consume(len);
            uint32_t sym = byte_sym(l, W_CL_SYM, (uint32_t)idx);
            int rep;
            uint32_t val;
            if (sym <= 15) {
                rep = 1; val = sym;
            } else if (sym == 16 && n > 0) {
                if (br.bc < 2) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :132
                rep = (int)br.bits(2) + 3; val = prev;
                if (n + rep > total) return SWC_E_DEFLATE_WRONG_SYMBOL;  // :135
            } else if (sym == 17) {
                if (br.bc < 3) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :145
                n += (int)br.bits(3) + 3; prev = 0;
                continue;
            } else if (sym == 18) {
                if (br.bc < 7) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :152
                n += (int)br.bits(7) + 11; prev = 0;
                continue;
            } else {
                return SWC_E_DEFLATE_WRONG_SYMBOL;  // :155 (symbol 16 first)
            }
            prev = val;
            if (val != 0) {
                for (int i = 0; i < rep; i++) {
                    int s = n + i;
                    bool is_lit = s < literals;
                    int slot = (is_lit ? W_LIT_LEN : W_DIST_LEN) + (int)val;
                    uint32_t w = l.get(slot);
                    l.set(slot, w + 1);
                    if (PASS2) {
                        if (is_lit) set_lit_sym(l, w & 511u, (uint32_t)s);
                        else set_byte_sym(l, W_DIST_SYM, w & 511u, (uint32_t)(s - literals));
                    }
                }
            }
            n += rep;
        }
        if (n != total) return SWC_E_DEFLATE_WRONG_SYMBOL;  // :161
        return SWC_OK;
    }

    // Deflate.swift:86-167
    SWC_HD int build_dynamic() {
        br.refill();
        if (br.bc < 14) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :86
        int literals = (int)br.bits(5) + 257;
        if (literals > 286) return SWC_E_DEFLATE_WRONG_SYMBOL;  // :94
        int distances = (int)br.bits(5) + 1;
        int ncl = (int)br.bits(4) + 4;
        // 3*ncl <= 57 bits: gather them through two refills
        br.refill();
        uint64_t avail = br.bc;
        // bitsLeft covers the whole remaining stream, not just the window (:101)
        uint64_t total_left = (uint64_t)br.len * 8 - br.consumed_bits();
        if (total_left < (uint64_t)(3 * ncl)) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;
        (void)avail;
        // code-length alphabet: 19 x 3 bits in codeLengthOrders order (Deflate+Constants.swift:175)
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint64_t clens = 0;  // 3 bits per symbol, indexed by symbol
        for (int i = 0; i < ncl; i++) {
            br.refill();
            clens |= (uint64_t)br.bits(3) << (3 * order[i]);
        }
        clear_slots<7>(l, W_CL_LEN);
        for (int s = 0; s < 19; s++) {
            uint32_t len = (uint32_t)(clens >> (3 * s)) & 7u;
            if (len) l.set(W_CL_LEN + (int)len, l.get(W_CL_LEN + (int)len) + 1);
        }
        counts_to_slots<7>(l, W_CL_LEN, nullptr);
        for (int s = 0; s < 19; s++) {
            uint32_t len = (uint32_t)(clens >> (3 * s)) & 7u;
            if (len) {
                uint32_t w = l.get(W_CL_LEN + (int)len);
                l.set(W_CL_LEN + (int)len, w + 1);
                set_byte_sym(l, W_CL_SYM, w & 511u, (uint32_t)s);
            }
        }
        fixup_slots<7>(l, W_CL_LEN);

        clear_slots<15>(l, W_LIT_LEN);
        clear_slots<15>(l, W_DIST_LEN);
        BitReader save = br;
        int st = scan_lengths<false>(literals, literals + distances);
        if (st) return st;
        counts_to_slots<15>(l, W_LIT_LEN, &lit);
        counts_to_slots<15>(l, W_DIST_LEN, &dist);
        br = save;
        st = scan_lengths<true>(literals, literals + distances);
        if (st) return st;
        fixup_slots<15>(l, W_LIT_LEN);
        fixup_slots<15>(l, W_DIST_LEN);
        return SWC_OK;
    }

    // Deflate.swift:77-81 with the fixed code of Deflate+Constants.swift:11-173
    SWC_HD void build_static() {
        clear_slots<15>(l, W_LIT_LEN);
        clear_slots<15>(l, W_DIST_LEN);
        l.set(W_LIT_LEN + 7, 24);
        l.set(W_LIT_LEN + 8, 152);
        l.set(W_LIT_LEN + 9, 112);
        l.set(W_DIST_LEN + 5, 32);
        counts_to_slots<15>(l, W_LIT_LEN, &lit);
        counts_to_slots<15>(l, W_DIST_LEN, &dist);
        for (uint32_t s = 0; s < 288; s++) {
            int len = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            uint32_t w = l.get(W_LIT_LEN + len);
            l.set(W_LIT_LEN + len, w + 1);
            set_lit_sym(l, w & 511u, s);
        }
        for (uint32_t s = 0; s < 32; s++) set_byte_sym(l, W_DIST_SYM, s, s);
        l.set(W_DIST_LEN + 5, l.get(W_DIST_LEN + 5) + 32);
        fixup_slots<15>(l, W_LIT_LEN);
        fixup_slots<15>(l, W_DIST_LEN);
    }

    SWC_HD void put_byte(uint8_t b) {
        if (pos < cap) out[pos] = b;
        pos++;
    }

    // Deflate.swift:171-236
    SWC_HD int run_block() {
        // ---- interior fast loop: >= 8 input bytes beyond the read-ahead and >= 272 output bytes of
        // room, code sets not over-subscribed => no truncation / capacity / exact-lookup checks.
        if (!lit.oversub && !dist.oversub) {
            while ((uint64_t)br.ppos + 8 <= br.len && pos + 272 <= cap) {
                br.refill_fast();
                int sym = decode_sym<true, false>();
                if (sym < 0) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;
                if (sym < 256) {
                    out[pos] = (uint8_t)sym;
                    pos++;
                    continue;
                }
                if (sym == 256) return SWC_OK;
                if (sym > 285) return SWC_E_DEFLATE_WRONG_SYMBOL;
                uint32_t s = (uint32_t)sym - 257u;
                uint32_t e = s < 8 || s == 28 ? 0u : (s >> 2) - 1u;
                uint32_t length = (s < 8 ? 3u + s : s == 28 ? 258u : 3u + ((4u + (s & 3u)) << e)) + br.bits(e);
                br.refill_fast();
                int dc = decode_sym<false, false>();
                if (dc < 0) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;
                if (dc > 29) return SWC_E_DEFLATE_WRONG_SYMBOL;
                uint32_t de = dc < 4 ? 0u : ((uint32_t)dc >> 1) - 1u;
                uint32_t distance = (dc < 4 ? 1u + (uint32_t)dc : 1u + ((2u + ((uint32_t)dc & 1u)) << de)) + br.bits(de);
                if ((uint64_t)distance > pos) return SWC_E_REF_TRAP;
                emit_match(length, distance);
            }
        }
        // ---- careful loop: stream tail, output tail, over-subscribed sets, size-counting mode
        for (;;) {
            br.refill();
            int sym = decode_sym<true>();
            if (sym < 0) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :175
            if (sym < 256) {
                put_byte((uint8_t)sym);
                continue;
            }
            if (sym == 256) return SWC_OK;
            if (sym > 285) return SWC_E_DEFLATE_WRONG_SYMBOL;  // :233
            uint32_t s = (uint32_t)sym - 257u, length;
            if (s < 8) {
                length = 3 + s;
            } else if (s == 28) {
                length = 258;
            } else {
                uint32_t e = (s >> 2) - 1;  // :188
                if (br.bc < e) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :192
                length = 3 + ((4 + (s & 3)) << e) + br.bits(e);  // Constants.lengthBase
            }
            br.refill();
            int dc = decode_sym<false>();
            if (dc < 0) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :199
            if (dc > 29) return SWC_E_DEFLATE_WRONG_SYMBOL;     // :201
            uint32_t distance;
            if (dc < 4) {
                distance = 1 + (uint32_t)dc;
            } else {
                uint32_t e = ((uint32_t)dc >> 1) - 1;  // :206
                if (br.bc < e) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :208
                distance = 1 + ((2 + ((uint32_t)dc & 1)) << e) + br.bits(e);  // Constants.distanceBase
            }
            // :216-221 out[count - distance] with distance > count is a Swift trap (App. A6)
            if ((uint64_t)distance > pos) return SWC_E_REF_TRAP;
            emit_match(length, distance);
        }
    }

    // Deflate.swift:45-65
    SWC_HD int run_stored() {
        br.consume(br.bc & 7);  // align()
        uint32_t p = (uint32_t)(br.consumed_bits() >> 3);
        if (br.len - p < 4) return SWC_E_DEFLATE_WRONG_UNCOMPRESSED_BLOCK_LENGTHS;  // :48
        gcptr q = br.in + p;
        uint32_t length = (uint32_t)q[0] | ((uint32_t)q[1] << 8);
        uint32_t nlength = (uint32_t)q[2] | ((uint32_t)q[3] << 8);
        if ((length & nlength) != 0) return SWC_E_DEFLATE_WRONG_UNCOMPRESSED_BLOCK_LENGTHS;  // :56
        if (br.len - (p + 4) < length) return SWC_E_DEFLATE_WRONG_UNCOMPRESSED_BLOCK_LENGTHS;  // :59
        q += 4;
        uint32_t i = 0;
        if (pos + length <= cap) {
            for (; i + 8 <= length; i += 8) store_u64(out + pos + i, load_u64(q + i));
        }
        for (; i < length; i++)
            if (pos + i < cap) out[pos + i] = q[i];
        pos += length;
        br.init(br.in, br.len, p + 4 + length);
        return SWC_OK;
    }

    SWC_HD int run() {
        if ((uint64_t)br.len * 8 < 10) return SWC_E_DEFLATE_WRONG_BLOCK_TYPE;  // :36
        for (;;) {
            br.refill();
            // a second or later block header past the end: LsbBitReader.bit() traps
            if (br.bc < 3) return SWC_E_REF_TRAP;
            uint32_t is_last = br.bits(1);
            uint32_t type = br.bits(2);
            int st;
            if (type == 0) {
                st = run_stored();
            } else if (type == 1) {
                build_static();
                st = run_block();
            } else if (type == 2) {
                st = build_dynamic();
                if (st == SWC_OK) st = run_block();
            } else {
                st = SWC_E_DEFLATE_WRONG_BLOCK_TYPE;  // :239
            }
            if (st) return st;
            if (is_last) return SWC_OK;  // :243
        }
    }
};

// One lane = one job.  `lds` is this lane's view of the wave's table region; `ws` / `ws_bytes` the
// stream's area in the HBM workspace (lzr::StreamHeader + records).
SWC_HD void inflate_job(Job& job, LaneLds lds, uint8_t* ws, size_t ws_bytes) {
    Lane ln;
    ln.l = lds;
    ln.out = (gptr)job.out;
    ln.cap = job.out_cap;
    ln.pos = 0;
    ln.nrec = 0;
    ln.last_end = 0;
    ln.recs = (SWC_AS_GLOBAL uint32_t*)(ws + sizeof(lzr::StreamHeader));
    ln.max_rec = ws && ws_bytes > sizeof(lzr::StreamHeader) ? (uint32_t)((ws_bytes - sizeof(lzr::StreamHeader)) / 4) : 0u;
    int st;
    if (job.in_len > 0xFFFFFFF0ull) {
        st = SWC_E_INVALID_ARGUMENT;  // streams are addressed with 32-bit byte offsets on device
        ln.br.init((gcptr)job.in, 0, 0);
    } else {
        ln.br.init((gcptr)job.in, (uint32_t)job.in_len, 0);
        st = ln.run();
    }
    if (ln.nrec > ln.max_rec) {
        st = SWC_E_NEED_WORKSPACE;  // the record list outgrew the workspace (sized from out_cap)
        ln.nrec = ln.max_rec;
    }
    if (st == SWC_OK && ln.pos > ln.cap) st = SWC_E_CAPACITY;
    if (ws && ws_bytes >= sizeof(lzr::StreamHeader)) ((SWC_AS_GLOBAL lzr::StreamHeader*)ws)->nrec = ln.nrec;
    uint64_t bits = ln.br.consumed_bits();
    uint64_t consumed = (bits + 7) >> 3;  // callers align() right after (GzipArchive.swift:89)
    job.in_consumed = consumed > job.in_len ? job.in_len : consumed;
    job.out_len = ln.pos;
    job.status = st;
}

}  // namespace inflate
}  // namespace swc
#endif
