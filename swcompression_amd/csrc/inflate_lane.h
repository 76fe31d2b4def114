// inflate_lane.h -- Deflate entropy decode (phase 1 of the two-phase path), one compressed stream per lane.
//
// Replaces the body of Deflate.decompress(_ bitReader:) (reference Sources/Deflate/Deflate.swift:30-249)
// together with Code.huffmanCodes (Sources/Common/CodingTree/Code.swift:15-39) and DecodingTree
// (Sources/Common/CodingTree/DecodingTree.swift:15-50).  Semantics -- including the reference's
// acceptance of incomplete and over-subscribed Huffman sets -- are preserved;
// the data structures are not.  Instead of a 2^(maxBits+1) heap walked one bit per step, each lane keeps
//
//   in LDS (80 words per lane, interleaved at wave stride so that arbitrary per-lane indices never
//   bank-conflict; 20 KiB per wave => 8 waves = 2 per SIMD resident per CU):
//       288 + 32 bytes: the lit/len and distance symbols sorted by (code length, symbol); lit/len symbols
//       are stored modulo 256 -- within one length the symbols >= 256 come last, so bit 8 is an index compare;
//   in VGPRs, per alphabet (struct Table):
//       lim[d]  = (first[d] + count[d]) << (15 - d), non-decreasing in d: the code length of the next 15 stream
//                 bits c15 (bit-reversed, first bit = MSB) is 1 + #{d : c15 >= lim[d]}, found by a 4-step
//                 binary search over registers;
//       slot[d] = (start[d] - first[d]) & 0xFFFF | thr[d] << 16: sorted index = slot + (c15 >> (15 - len)),
//                 symbol >= 256 iff index >= thr.
//
// Exact path (over-subscribed sets, and always for the 19-symbol code-length alphabet): for d = 1..15 the heap
// node at depth d on the path c15 is occupied iff k0 = ((c15 >> (15-d)) - first[d]) mod 2^d < count[d]; the
// last writer among k0, k0+2^d, ... wins (DecodingTree.swift:22-32), the shallowest occupied node wins (:45).
//
// The dynamic header is decoded twice (count pass into LDS counters that alias the symbol area, then the
// scatter pass with the per-length running positions packed in registers) so that no per-symbol length array
// has to be kept.
#ifndef SWC_INFLATE_LANE_H
#define SWC_INFLATE_LANE_H

#include "swc_common.h"
#include "lz_resolve.h"

namespace swc {
namespace inflate {

constexpr int W_LIT_SYM = 0;     // 72 words : 288 x 8-bit symbols (symbol & 255)
constexpr int W_DIST_SYM = 72;   //  8 words : 32 x 8-bit symbols
constexpr int kWordsPerLane = 80;
// wave mode: direct-lookup tables for codes of up to kLut*Bits bits (longer codes take the checked step) and the staged input
constexpr int kLutLitBits = 10, kLutDistBits = 9, kStageBytes = 512;
constexpr int kWaveModeLdsBytes = kWordsPerLane * 4 + (4 << kLutLitBits) + (4 << kLutDistBits) + kStageBytes + 16;
constexpr int kLdsBytesPerWave = kWordsPerLane * 4 * kWave;  // 20,480 B
// count pass only (aliases the symbol area, which is written afterwards):
constexpr int W_CNT_LIT = 0;     // 16 words : codes per length, lit/len alphabet
constexpr int W_CNT_LO = 16;     // 16 words : of those, symbols < 256
constexpr int W_CNT_DIST = 32;   // 16 words : codes per length, distance alphabet

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
    // reposition at absolute bit `p` of the stream
    SWC_HD void seek(uint64_t p) {
        init(in, len, (uint32_t)(p >> 3));
        consume((uint32_t)p & 7u);
    }
    // bits consumed from the start of the stream
    SWC_HD uint64_t consumed_bits() const { return (uint64_t)(ppos - nextn) * 8 - bc; }
};

// One canonical Huffman alphabet as Code.swift:15-39 assigns it, in registers (every user is fully unrolled).
struct Table {
    uint32_t lim[16];   // [1..15]
    uint32_t slot[17];  // [1..15]; [16] = number of codes
    bool oversub;

    // derived per-length quantities (exact path only)
    SWC_HD uint32_t first(int d) const { return d == 1 ? 0u : lim[d - 1] >> (15 - d); }          // code counter of the first code of length d
    SWC_HD uint32_t start(int d) const { return d == 16 ? slot[16] : (slot[d] + first(d)) & 0xFFFFu; }  // its index in the sorted array
};

SWC_HD uint8_t* sym_ptr(const LaneLds& l, int base, uint32_t i) { return (uint8_t*)(l.p + (size_t)(base + (int)(i >> 2)) * l.stride) + (i & 3u); }

// t.slot[i] for i in 0..15 as a 4-level select tree over register values.  The table is taken BY VALUE so that it
// is scalarised inside this function before the selects are formed, and the selector bits are opaque so that the
// tree is not re-expressed as sixteen `i == k` compares.
SWC_HD uint32_t slot_of(Table t, uint32_t i) {
    uint32_t b8 = i & 8u, b4 = i & 4u, b2 = i & 2u, b1 = i & 1u;
    SWC_OPAQUE(b8); SWC_OPAQUE(b4); SWC_OPAQUE(b2); SWC_OPAQUE(b1);
    const bool c8 = b8 != 0, c4 = b4 != 0, c2 = b2 != 0, c1 = b1 != 0;
    const uint32_t a0 = c8 ? t.slot[8] : t.slot[0], a1 = c8 ? t.slot[9] : t.slot[1], a2 = c8 ? t.slot[10] : t.slot[2];
    const uint32_t a3 = c8 ? t.slot[11] : t.slot[3], a4 = c8 ? t.slot[12] : t.slot[4], a5 = c8 ? t.slot[13] : t.slot[5];
    const uint32_t a6 = c8 ? t.slot[14] : t.slot[6], a7 = c8 ? t.slot[15] : t.slot[7];
    const uint32_t e0 = c4 ? a4 : a0, e1 = c4 ? a5 : a1, e2 = c4 ? a6 : a2, e3 = c4 ? a7 : a3;
    const uint32_t f0 = c2 ? e2 : e0, f1 = c2 ? e3 : e1;
    return c1 ? f1 : f0;
}

// 1 + #{d in 1..15 : c15 >= lim[d]} by binary search (lim is non-decreasing)
SWC_HD uint32_t code_length(Table t, uint32_t c15) {
    const uint32_t* L = t.lim;
    bool b8 = c15 >= L[8];
    uint32_t m = b8 ? L[12] : L[4];
    bool b4 = c15 >= m;
    m = b8 ? (b4 ? L[14] : L[10]) : (b4 ? L[6] : L[2]);
    bool b2 = c15 >= m;
    {
        const uint32_t a = b2 ? L[3] : L[1], b = b2 ? L[7] : L[5];
        const uint32_t c = b2 ? L[11] : L[9], e = b2 ? L[15] : L[13];
        const uint32_t ab = b4 ? b : a, ce = b4 ? e : c;
        m = b8 ? ce : ab;
    }
    bool b1 = c15 >= m;
    return 1u + (b8 ? 8u : 0u) + (b4 ? 4u : 0u) + (b2 ? 2u : 0u) + (b1 ? 1u : 0u);
}

// Code length AND slot word of the code that starts c15, for a set that is not over-subscribed: the slot select
// tree shares the compare results of the binary search (slot index = length - 1).  length 16 = no such code.
SWC_HD uint32_t code_length_slot(Table t, uint32_t c15, uint32_t& sl) {
    const uint32_t* L = t.lim;
    const uint32_t* S = t.slot + 1;   // S[p] = slot of length p + 1; S[15] = slot[16] (never used: length 16 is rejected)
    const bool b8 = c15 >= L[8];
    uint32_t m = b8 ? L[12] : L[4];
    const uint32_t a0 = b8 ? S[8] : S[0], a1 = b8 ? S[9] : S[1], a2 = b8 ? S[10] : S[2], a3 = b8 ? S[11] : S[3];
    const uint32_t a4 = b8 ? S[12] : S[4], a5 = b8 ? S[13] : S[5], a6 = b8 ? S[14] : S[6], a7 = b8 ? S[15] : S[7];
    const bool b4 = c15 >= m;
    m = b8 ? (b4 ? L[14] : L[10]) : (b4 ? L[6] : L[2]);
    const uint32_t e0 = b4 ? a4 : a0, e1 = b4 ? a5 : a1, e2 = b4 ? a6 : a2, e3 = b4 ? a7 : a3;
    const bool b2 = c15 >= m;
    {
        const uint32_t a = b2 ? L[3] : L[1], b = b2 ? L[7] : L[5];
        const uint32_t c = b2 ? L[11] : L[9], e = b2 ? L[15] : L[13];
        const uint32_t ab = b4 ? b : a, ce = b4 ? e : c;
        m = b8 ? ce : ab;
    }
    const uint32_t f0 = b2 ? e2 : e0, f1 = b2 ? e3 : e1;
    const bool b1 = c15 >= m;
    sl = b1 ? f1 : f0;
    return 1u + (b8 ? 8u : 0u) + (b4 ? 4u : 0u) + (b2 ? 2u : 0u) + (b1 ? 1u : 0u);
}

// Exact heap-equivalent lookup.  Returns the index into the sorted symbol array or -1.
SWC_HD int lookup_exact(const Table& t, uint32_t c15, uint32_t& len) {
#pragma unroll
    for (int d = 1; d <= 15; d++) {
        const uint32_t st = t.start(d), cnt = (t.start(d + 1) - st) & 0xFFFFu;
        const uint32_t k0 = ((c15 >> (15 - d)) - t.first(d)) & ((1u << d) - 1u);
        if (k0 < cnt) {
            len = (uint32_t)d;
            return (int)(st + k0 + (((cnt - 1u - k0) >> d) << d));
        }
    }
    return -1;
}

// Fill lim[] / slot[] from the per-length counts (Code.swift:23-37 restated per length).  cnt(d), lo(d): number
// of codes of length d and, of those, symbols < 256.  Returns the start index of every length packed for the
// scatter pass: run[d / 6] holds six 10-bit fields.
template <typename CntFn, typename LoFn>
SWC_HD void build_table(Table& t, CntFn cnt_of, LoFn lo_of, uint64_t run[3]) {
    uint32_t v = 0, off = 0;
    bool over = false;
    run[0] = run[1] = run[2] = 0;
    t.lim[0] = 0;
    t.slot[0] = 0;
#pragma unroll
    for (int d = 1; d <= 15; d++) {
        const uint32_t cnt = cnt_of(d);
        t.lim[d] = (v + cnt) << (15 - d);
        if (cnt != 0 && v + cnt > (1u << d)) over = true;
        t.slot[d] = ((off - v) & 0xFFFFu) | ((off + lo_of(d)) << 16);
        run[d / 6] |= (uint64_t)off << (10 * (d % 6));
        off += cnt;
        v = (v + cnt) << 1;
    }
    t.slot[16] = off;
    t.oversub = over;
}
SWC_HD uint32_t run_take(uint64_t run[3], uint32_t d) {  // returns the running index of length d and advances it
    const uint32_t q = (d * 11u) >> 6, sh = 10u * (d - 6u * q);
    uint64_t r0 = run[0], r1 = run[1], r2 = run[2];
    SWC_OPAQUE(r0); SWC_OPAQUE(r1); SWC_OPAQUE(r2);
    const uint64_t r = q == 0 ? r0 : q == 1 ? r1 : r2;
    const uint64_t inc = 1ull << sh;
    run[0] += q == 0 ? inc : 0;
    run[1] += q == 1 ? inc : 0;
    run[2] += q == 2 ? inc : 0;
    return (uint32_t)(r >> sh) & 1023u;
}

// Phase 1 of the two-phase Deflate path (see lz_resolve.h): literals go straight to their final position,
// every match becomes one record of the stream's record list; the output buffer is never read here.
struct Lane {
    LaneLds l;
    BitReader br;
    Table lit, dist;
    gptr out;
    uint64_t cap;
    uint64_t pos;  // bytes produced (keeps counting past cap: size pass for SWC_E_CAPACITY)
    SWC_AS_GLOBAL uint32_t* recs;  // record list in the HBM workspace (the stream's StreamHeader sits 16 bytes before it)
    uint32_t nrec, max_rec;
    gptr lits;                     // dense literal stream in the HBM workspace (16-byte aligned, capacity cap + 16)
    uint64_t nlit;
    uint64_t last_end;             // position just past the previous record
    int dbg = 0;                   // timing experiments only (tools/exp_deflate.py): 1 no literal stores, 2 no record stores
    // ---- wave mode (one stream per WAVEFRONT, small batches; see wave_loop): all 64 lanes run this object redundantly
    // on one shared LDS column, `wlane` is the lane's number (-1: lane mode, one stream per lane)
    int wlane = -1;
    int wlanes = kWave;            // lanes that share the work of the parallel parts (1 in the host emulation)
    uint32_t* lut_lit = nullptr;   // LDS, 1 << kLutLitBits entries (wave_loop)
    uint32_t* lut_dist = nullptr;  // LDS, 1 << kLutDistBits entries
    uint8_t* stage = nullptr;      // LDS, kStageBytes + 8: a window of the input
    uint32_t stage_base = 0, stage_len = 0;
    bool luts_ready = false;

    SWC_HD void push(uint32_t v) {
        if (nrec < max_rec) recs[nrec] = v;
        nrec++;
    }
    // Deflate.swift:216-232 deferred: record (literal run, length, distance); phase 2 executes the copy.
    // literal-only records for `run` literal bytes that are already in the literal stream
    SWC_HD void push_lits(uint64_t run) {
        while (run > 0) {
            const uint32_t s = run > lzr::kMaxLitOnly ? lzr::kMaxLitOnly : (uint32_t)run;
            push(lzr::make_lits(s));
            run -= s;
        }
    }
    SWC_HD void emit_match(uint32_t length, uint32_t distance) {
        if (pos < cap) {  // records exist only for matches that start below the capacity
            uint64_t run = pos - last_end;
            if (run > lzr::kLitRunMax) { push_lits(run); run = 0; }
            push(lzr::make_match((uint32_t)run, length, distance));
            last_end = pos + length;
        }
        pos += length;
    }
    // the literals behind the last match (kept only below the capacity) become literal-only records
    SWC_HD void flush_tail() {
        const uint64_t kept = pos < cap ? pos : cap;
        if (kept > last_end) { push_lits(kept - last_end); last_end = kept; }
    }

    // Decode one symbol of the lit/len (LIT=true) or distance alphabet.  Returns the symbol or
    // -1 (DeflateError.symbolNotFound: unassigned path, or the code runs past the end of input).
    template <bool LIT, bool CHECKED = true>
    SWC_HD int decode_sym() {
        const Table& t = LIT ? lit : dist;
        const uint32_t c15 = brev32(br.peek32()) >> 17;
        uint32_t len, idx;
        if (!CHECKED || !t.oversub) {
            len = code_length(t, c15);
            if (len > 15) return -1;
        } else {
            int i = lookup_exact(t, c15, len);
            if (i < 0) return -1;
            idx = (uint32_t)i;
        }
        const uint32_t sl = slot_of(t, len);
        if (!CHECKED || !t.oversub) idx = (sl + (c15 >> (15 - len))) & 0xFFFFu;
        if (CHECKED && len > br.bc) return -1;  // DecodingTree.swift:39 -- ran out of bits before reaching a leaf
        br.consume(len);
        uint32_t sym = *sym_ptr(l, LIT ? W_LIT_SYM : W_DIST_SYM, idx);
        if (LIT) sym |= idx >= (sl >> 16) ? 256u : 0u;
        return (int)sym;
    }

    // The 19-symbol code-length alphabet, in registers: slot[d] = first << 9 | start (d = 1..7), slot[8] = total;
    // sorted symbols packed 5 bits each.
    struct ClTable {
        uint32_t slot[9];
        uint64_t sym_lo, sym_hi;
    };
    SWC_HD static int cl_lookup(const ClTable& t, uint32_t c15, uint32_t& len) {
#pragma unroll
        for (int d = 1; d <= 7; d++) {
            const uint32_t w = t.slot[d], wn = t.slot[d + 1];
            const uint32_t cnt = (wn & 511u) - (w & 511u);
            const uint32_t k0 = ((c15 >> (15 - d)) - (w >> 9)) & ((1u << d) - 1u);
            if (k0 < cnt) {
                len = (uint32_t)d;
                return (int)((w & 511u) + k0 + (((cnt - 1u - k0) >> d) << d));
            }
        }
        return -1;
    }

    // Walk the code-length section of a dynamic header (Deflate.swift:117-162).  PASS2 = false counts codes
    // per length into the LDS counters; PASS2 = true scatters symbols into the sorted arrays.
    template <bool PASS2>
    SWC_HD int scan_lengths(const ClTable& cl, int literals, int total, uint64_t run_lit[3], uint64_t run_dist[3]) {
        int n = 0;
        uint32_t prev = 0;
        while (n < total) {
            br.refill();
            uint32_t c15 = brev32(br.peek32()) >> 17, len;
            int idx = cl_lookup(cl, c15, len);
            if (idx < 0 || len > br.bc) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :122
            br.consume(len);
            uint32_t sym = (uint32_t)((idx < 12 ? cl.sym_lo >> (5 * idx) : cl.sym_hi >> (5 * (idx - 12))) & 31u);
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
                if (!PASS2) {
                    int n_lit = literals - n; n_lit = n_lit < 0 ? 0 : n_lit > rep ? rep : n_lit;
                    int n_lo = 256 - n;       n_lo = n_lo < 0 ? 0 : n_lo > rep ? rep : n_lo;
                    if (n_lit) l.set(W_CNT_LIT + (int)val, l.get(W_CNT_LIT + (int)val) + (uint32_t)n_lit);
                    if (n_lo) l.set(W_CNT_LO + (int)val, l.get(W_CNT_LO + (int)val) + (uint32_t)n_lo);
                    if (rep - n_lit) l.set(W_CNT_DIST + (int)val, l.get(W_CNT_DIST + (int)val) + (uint32_t)(rep - n_lit));
                } else {
                    for (int i = 0; i < rep; i++) {
                        int s = n + i;
                        if (s < literals) *sym_ptr(l, W_LIT_SYM, run_take(run_lit, val)) = (uint8_t)s;
                        else *sym_ptr(l, W_DIST_SYM, run_take(run_dist, val)) = (uint8_t)(s - literals);
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
        luts_ready = false;
        br.refill();
        if (br.bc < 14) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :86
        int literals = (int)br.bits(5) + 257;
        if (literals > 286) return SWC_E_DEFLATE_WRONG_SYMBOL;  // :94
        int distances = (int)br.bits(5) + 1;
        int ncl = (int)br.bits(4) + 4;
        br.refill();
        // bitsLeft covers the whole remaining stream, not just the window (:101)
        uint64_t total_left = (uint64_t)br.len * 8 - br.consumed_bits();
        if (total_left < (uint64_t)(3 * ncl)) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;
        // code-length alphabet: 19 x 3 bits in codeLengthOrders order (Deflate+Constants.swift:175)
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint64_t clens = 0;  // 3 bits per symbol, indexed by symbol
        for (int i = 0; i < ncl; i++) {
            br.refill();
            clens |= (uint64_t)br.bits(3) << (3 * order[i]);
        }
        ClTable cl;
        {
            uint64_t cnt8 = 0, runp = 0;  // eight 8-bit fields, indexed by length
            for (int s = 0; s < 19; s++) {
                uint32_t len = (uint32_t)(clens >> (3 * s)) & 7u;
                if (len) cnt8 += 1ull << (8 * len);
            }
            uint32_t v = 0, off = 0;
            cl.slot[0] = 0;
#pragma unroll
            for (int d = 1; d <= 7; d++) {
                const uint32_t c = (uint32_t)(cnt8 >> (8 * d)) & 255u;
                cl.slot[d] = ((v & 0x7FFFu) << 9) | off;
                runp |= (uint64_t)off << (8 * d);
                off += c;
                v = (v + c) << 1;
            }
            cl.slot[8] = off;
            cl.sym_lo = cl.sym_hi = 0;
            for (int s = 0; s < 19; s++) {
                uint32_t len = (uint32_t)(clens >> (3 * s)) & 7u;
                if (len) {
                    uint32_t pos = (uint32_t)(runp >> (8 * len)) & 255u;
                    runp += 1ull << (8 * len);
                    if (pos < 12) cl.sym_lo |= (uint64_t)s << (5 * pos);
                    else cl.sym_hi |= (uint64_t)s << (5 * (pos - 12));
                }
            }
        }
        for (int j = 0; j < 48; j++) l.set(j, 0);
        uint64_t run_lit[3], run_dist[3];
        BitReader save = br;
        int st = scan_lengths<false>(cl, literals, literals + distances, run_lit, run_dist);
        if (st) return st;
        build_table(lit, [&](int d) { return l.get(W_CNT_LIT + d); }, [&](int d) { return l.get(W_CNT_LO + d); }, run_lit);
        build_table(dist, [&](int d) { return l.get(W_CNT_DIST + d); }, [&](int) { return 0u; }, run_dist);
        br = save;
        return scan_lengths<true>(cl, literals, literals + distances, run_lit, run_dist);
    }

    // Deflate.swift:77-81 with the fixed code of Deflate+Constants.swift:11-173
    SWC_HD void build_static() {
        luts_ready = false;
        uint64_t run_lit[3], run_dist[3];
        build_table(lit, [](int d) { return d == 7 ? 24u : d == 8 ? 152u : d == 9 ? 112u : 0u; },
                    [](int d) { return d == 8 ? 144u : d == 9 ? 112u : 0u; }, run_lit);
        build_table(dist, [](int d) { return d == 5 ? 32u : 0u; }, [](int) { return 0u; }, run_dist);
        for (uint32_t s = 0; s < 288; s++) {
            uint32_t len = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            *sym_ptr(l, W_LIT_SYM, run_take(run_lit, len)) = (uint8_t)s;
        }
        for (uint32_t s = 0; s < 32; s++) *sym_ptr(l, W_DIST_SYM, s) = (uint8_t)s;
    }

    SWC_HD void put_byte(uint8_t b) {
        if (pos < cap) lits[nlit++] = b;
        pos++;
    }

    SWC_HD static uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh) {  // bits [sh, sh + 32) of hi:lo, sh < 32
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
    }

    // Interior fast loop of a block whose code sets are not over-subscribed.
    //
    // The body is ONE straight-line, fully predicated schedule -- every lane executes the lit/len decode, the
    // distance decode, one literal store, one record store and one 8-byte input load per iteration, whatever its
    // symbol is (a wave executes all of it anyway as soon as one lane holds a match).  With a fixed sequence of
    // memory instructions the compiler can wait for the input load with an exact s_waitcnt vmcnt(2) instead of
    // vmcnt(0); measured before: 81 % of wave time was spent at a loop-head vmcnt(0) that drained the stores of the
    // previous iteration (profiles/r01_pmc_deflate_two_phase_v1.txt).  Stores of lanes that do not own them are
    // harmless by construction: a match lane writes a garbage byte at its match start (phase 2 overwrites it), a
    // literal lane writes a garbage record into the NEXT FREE record slot (overwritten by the next real record or
    // never counted).
    //
    // Bit window: four valid dwords d0..d3 (128 bits) + two in flight; bp < 64 at the top of the decode, one
    // iteration consumes <= 63 bits (a literal, then a match: 15 + 20 + 28), so ONE 64-bit shift per iteration always
    // suffices and no code reaches past bit 127.  Positions are 32 bit.
    // ANYTHING unusual -- unassigned code, symbol > 285, distance symbol > 29, distance beyond the output, a literal
    // run of 128+ before a match -- leaves the loop BEFORE consuming the symbol; the caller then decodes that one
    // symbol with the fully checked step.  Returns true when the end-of-block symbol was consumed.
    SWC_HD bool fast_loop() {
        const uint64_t P = br.consumed_bits();
        uint32_t q = (uint32_t)(P >> 6) << 3, bp = (uint32_t)P & 63u;
        gcptr in = br.in;
        const uint32_t len = br.len;
        if ((uint64_t)q + 48 > len) return false;
        uint32_t d0 = load_u32(in + q), d1 = load_u32(in + q + 4), d2 = load_u32(in + q + 8), d3 = load_u32(in + q + 12);
        uint64_t nx = load_u64(in + q + 16);
        // drain the set-up loads here, once: otherwise the loop-head wait has to cover the entry path too and
        // degrades to vmcnt(0) for every iteration
        SWC_OPAQUE(d0); SWC_OPAQUE(d1); SWC_OPAQUE(d2); SWC_OPAQUE(d3); SWC_OPAQUE(nx);
        uint32_t p = (uint32_t)pos, le = (uint32_t)last_end, nr = nrec, nl = (uint32_t)nlit;
        // literals are collected eight at a time and leave as one aligned 8-byte store
        uint64_t lb = (nl & 7u) ? load_u64(lits + (nl & ~7u)) & ((1ull << (8 * (nl & 7u))) - 1ull) : 0ull;
        const uint32_t plimit = (uint32_t)(cap - 272);
        // Memory instructions of lanes that do not need them this iteration are pointed at ONE wave-wide dummy
        // location (the first lane's stream header, rewritten when its job ends / the first lane's input): the instruction stays
        // unconditional -- which is what lets the compiler count it -- but costs a single extra request.
        gptr dummy_st = (gptr)recs - 16;   // the 16-byte stream header (rewritten when the job ends)
        gcptr dummy_ld = in;
#if defined(__HIP_DEVICE_COMPILE__)
        {
            uint64_t a = (uint64_t)(uintptr_t)dummy_st, b = (uint64_t)(uintptr_t)dummy_ld;
            a = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
            b = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            dummy_st = (gptr)(uintptr_t)a;
            dummy_ld = (gcptr)(uintptr_t)b;
        }
#endif
        bool eob = false;
        for (;;) {
            // one 64-bit shift if due, then the unconditional prefetch of the granule after the next one
            const bool sh = bp >= 64;
            d0 = sh ? d2 : d0;
            d1 = sh ? d3 : d1;
            d2 = sh ? (uint32_t)nx : d2;
            d3 = sh ? (uint32_t)(nx >> 32) : d3;
            q += sh ? 8u : 0u;
            bp -= sh ? 64u : 0u;
            if (q + 48 > len || p > plimit) break;
            {
                const uint64_t t = load_u64(sh ? in + q + 16 : dummy_ld);
                nx = sh ? t : nx;
            }
            // ---- symbol A: lit/len code at bp (< 64)
            const bool k1 = bp >= 32;
            const uint32_t pkA = funnel(k1 ? d2 : d1, k1 ? d1 : d0, bp & 31u);
            const uint32_t c15A = brev32(pkA) >> 17;
            uint32_t slA;
            const uint32_t nA = code_length_slot(lit, c15A, slA);
            uint32_t idxA = (slA + (c15A >> (15 - (nA & 15u)))) & 0xFFFFu;
            const bool a_lit = nA <= 15 && idxA < (slA >> 16);
            idxA = idxA > 319u ? 319u : idxA;
            const uint32_t symA = *sym_ptr(l, W_LIT_SYM, idxA);
            // ---- symbol M: the lit/len code after A when A is a literal, A itself otherwise (decoded again at the same
            // position: the schedule stays fixed).  At most ONE match per iteration, and it is always M.
            const uint32_t bpm = bp + (a_lit ? nA : 0u);   // < 64 + 15
            const bool j1 = bpm >= 32, j2 = bpm >= 64;
            const uint32_t pk = funnel(j2 ? d3 : j1 ? d2 : d1, j2 ? d2 : j1 ? d1 : d0, bpm & 31u);
            const uint32_t c15 = brev32(pk) >> 17;
            uint32_t sl;
            const uint32_t n = code_length_slot(lit, c15, sl);
            uint32_t idx = (sl + (c15 >> (15 - (n & 15u)))) & 0xFFFFu;
            const bool code_ok = n <= 15;
            const bool m_lit = code_ok && idx < (sl >> 16);
            idx = idx > 319u ? 319u : idx;
            const uint32_t sym = *sym_ptr(l, W_LIT_SYM, idx);
            // ---- length + distance of M (computed by every lane)
            const uint32_t s = (sym - 1u) & 31u;
            const uint32_t e = s < 8 || s >= 28 ? 0u : (s >> 2) - 1u;
            const uint32_t length = (s < 8 ? 3u + s : s >= 28 ? 258u : 3u + ((4u + (s & 3u)) << e)) + ((pk >> n) & ((1u << e) - 1u));
            const uint32_t bp2 = bpm + n + e;  // < 79 + 20: the 28 bits of a distance code end below bit 128
            const bool m1 = bp2 >= 32, m2 = bp2 >= 64, m3 = bp2 >= 96;
            const uint32_t xlo = m3 ? d3 : m2 ? d2 : m1 ? d1 : d0, xhi = m2 ? d3 : m1 ? d2 : d1;
            const uint32_t pk2 = funnel(xhi, xlo, bp2 & 31u);
            const uint32_t c15d = brev32(pk2) >> 17;
            uint32_t sl2;
            const uint32_t n2 = code_length_slot(dist, c15d, sl2);
            uint32_t idx2 = (sl2 + (c15d >> (15 - (n2 & 15u)))) & 0xFFFFu;
            idx2 = idx2 > 31u ? 31u : idx2;
            const uint32_t dc = *sym_ptr(l, W_DIST_SYM, idx2);
            const uint32_t de = dc < 4 ? 0u : ((dc >> 1) - 1u) & 15u;
            const uint32_t distance = (dc < 4 ? 1u + dc : 1u + ((2u + (dc & 1u)) << de)) + ((pk2 >> n2) & ((1u << de) - 1u));
            const uint32_t pm = p + (a_lit ? 1u : 0u);   // output position of M
            const uint32_t run = pm - le;
            // ---- classify
            const bool is_eob = code_ok && !m_lit && sym == 0;
            const bool is_match = code_ok && !m_lit && sym != 0 && sym <= 29 && n2 <= 15 && dc <= 29 && distance <= pm && run <= lzr::kLitRunMax;
            const bool lit2 = a_lit && m_lit;
            // ---- unconditional stores (see above), BEFORE the exits so that every path from the input load to its use
            // in the next iteration passes exactly these two stores.  A lane that leaves below has written one garbage
            // record into the next free slot, which gets overwritten.  Up to two literals join the pending eight-byte
            // group; at most one of them completes it.
            const uint64_t lbA = lb | (a_lit ? (uint64_t)symA << (8 * (nl & 7u)) : 0ull);
            const bool fullA = a_lit && (nl & 7u) == 7u;
            const uint32_t nl1 = nl + (a_lit ? 1u : 0u);
            const uint64_t lbB = (fullA ? 0ull : lbA) | (lit2 ? (uint64_t)sym << (8 * (nl1 & 7u)) : 0ull);
            const bool fullB = lit2 && (nl1 & 7u) == 7u;
            *(SWC_AS_GLOBAL u64_unaligned*)(((fullA || fullB) && !(dbg & 1)) ? lits + ((fullA ? nl : nl1) & ~7u) : dummy_st) = fullA ? lbA : lbB;
            *((m_lit || (dbg & 2)) ? (SWC_AS_GLOBAL uint32_t*)(dummy_st + 8) : recs + nr) = lzr::make_match(run, length, distance);
            if (!a_lit) {
                if (is_eob) { bp += n; eob = true; break; }
                if (!is_match) break;   // leave BEFORE consuming: the checked step handles it
            }
            // A literal A is consumed whatever M is; M is consumed when it is a literal or a good match (an end-of-block
            // code or anything unusual behind a literal becomes A of the next iteration).
            lb = fullB ? 0ull : lbB;
            nl = nl1 + (lit2 ? 1u : 0u);
            p = pm + (lit2 ? 1u : is_match ? length : 0u);
            le = is_match ? p : le;
            nr += is_match ? 1u : 0u;
            bp = bpm + (lit2 ? n : is_match ? n + e + n2 + de : 0u);
        }
        if (nl & 7u) store_u64(lits + (nl & ~7u), lb);   // pending literals; the bytes above them are rewritten later
        pos = (pos & ~0xFFFFFFFFull) | p;
        last_end = (last_end & ~0xFFFFFFFFull) | le;
        nrec = nr;
        nlit = nl;
        br.seek((uint64_t)q * 8 + bp);
        return eob;
    }

    // ---- wave mode ---------------------------------------------------------------------------------------------------
    // One stream per lane needs thousands of streams to fill the chip and decodes each of them at ~8 MB/s (one 64 KiB
    // block: 8.8 ms).  For small batches one stream gets a whole wavefront instead.  The symbol chain itself stays
    // serial, but the expensive part of a step -- finding the code that starts at a given bit -- is done for 64 bit
    // positions at once: lane k looks up, in direct tables in LDS, the lit/len code AND the distance code that would
    // start at bit P + k; the chain then hops from lane to lane with v_readlane (a few cycles) instead of decoding.
    // Everything unusual (a code longer than the tables, symbols > 285 / 29, distance beyond the output, the end of the
    // input) stops the hop chain BEFORE that symbol, and the fully checked step decodes it.
    //
    // Table entries.  lit/len: [0:3] code length (0: not in the table) [4:6] extra bits [7:8] 1 literal, 2 length,
    // 3 end of block [9:17] byte / base length.  distance: [0:3] code length [4:7] extra bits [8] symbol <= 29
    // [9:24] base distance.
    SWC_HD void build_luts() {
        const int lanes = wlanes, me = wlane;
        for (int i = me; i < (1 << kLutLitBits); i += lanes) lut_lit[i] = 0;
        for (int i = me; i < (1 << kLutDistBits); i += lanes) lut_dist[i] = 0;
        // sorted index j -> (length d, code first(d) + j - start(d)); the stream carries codes LSB first, so the table
        // index is the bit-reversed code, replicated over the don't-care bits above it
        const uint32_t n_lit = lit.slot[16], n_dist = dist.slot[16];
        for (uint32_t j = (uint32_t)me; j < n_lit; j += (uint32_t)lanes) {
            uint32_t d = 1, fst = 0, stt = lit.start(1);
#pragma unroll
            for (int q = 2; q <= 15; q++)   // the length whose index range holds j (static register indices only)
                if (j >= lit.start(q)) { d = (uint32_t)q; fst = lit.first(q); stt = lit.start(q); }
            if (d > (uint32_t)kLutLitBits) continue;
            const uint32_t code = fst + (j - stt);
            const uint32_t rev = brev32(code) >> (32 - d);
            uint32_t sym = *sym_ptr(l, W_LIT_SYM, j);
            if (j >= (slot_of(lit, d) >> 16)) sym |= 256u;
            uint32_t entry;
            if (sym < 256) entry = d | (1u << 7) | (sym << 9);
            else if (sym == 256) entry = d | (3u << 7);
            else if (sym <= 285) {
                const uint32_t t = sym - 257u;
                const uint32_t e = t < 8 || t == 28 ? 0u : (t >> 2) - 1u;
                const uint32_t base = t < 8 ? 3u + t : t == 28 ? 258u : 3u + ((4u + (t & 3u)) << e);
                entry = d | (e << 4) | (2u << 7) | (base << 9);
            } else entry = 0;   // 286, 287: the checked step reports wrongSymbol
            if (entry) for (uint32_t m = rev; m < (1u << kLutLitBits); m += 1u << d) lut_lit[m] = entry;
        }
        for (uint32_t j = (uint32_t)me; j < n_dist; j += (uint32_t)lanes) {
            uint32_t d = 1, fst = 0, stt = dist.start(1);
#pragma unroll
            for (int q = 2; q <= 15; q++)
                if (j >= dist.start(q)) { d = (uint32_t)q; fst = dist.first(q); stt = dist.start(q); }
            if (d > (uint32_t)kLutDistBits) continue;
            const uint32_t code = fst + (j - stt);
            const uint32_t rev = brev32(code) >> (32 - d);
            const uint32_t dc = *sym_ptr(l, W_DIST_SYM, j);
            if (dc > 29) continue;
            const uint32_t e = dc < 4 ? 0u : (dc >> 1) - 1u;
            const uint32_t base = dc < 4 ? 1u + dc : 1u + ((2u + (dc & 1u)) << e);
            const uint32_t entry = d | (e << 4) | (1u << 8) | (base << 9);
            for (uint32_t m = rev; m < (1u << kLutDistBits); m += 1u << d) lut_dist[m] = entry;
        }
        luts_ready = true;
    }
    // (re)stage the input window so that it starts at byte `from`
    SWC_HD void stage_input(uint32_t from) {
        const int lanes = wlanes, me = wlane;
        stage_base = from;
        const uint32_t left = br.len - from;
        stage_len = left < (uint32_t)kStageBytes ? left : (uint32_t)kStageBytes;
        for (uint32_t i = (uint32_t)me * 8u; i < stage_len; i += (uint32_t)lanes * 8u) {
            uint64_t w = 0;
            if (stage_len - i >= 8) w = load_u64(br.in + from + i);
            else for (uint32_t k = 0; k < stage_len - i; k++) w |= (uint64_t)br.in[from + i + k] << (8 * k);
            *(u64_unaligned*)(stage + i) = w;
        }
    }
    // What lane k sees at bit position P + k: packed lit/len view [0:4] bits incl. extra [5:6] kind [7:15] byte / length
    // and distance view [0:4] bits incl. extra [5] valid [6:21] distance.
    SWC_HD void wave_views(uint64_t bitpos, uint32_t& vl, uint32_t& vd) const {
        const uint64_t w = *(const u64_unaligned*)(stage + ((uint32_t)(bitpos >> 3) - stage_base)) >> ((uint32_t)bitpos & 7u);   // >= 57 bits
        const uint32_t ll = lut_lit[(uint32_t)w & ((1u << kLutLitBits) - 1u)];
        const uint32_t dd = lut_dist[(uint32_t)w & ((1u << kLutDistBits) - 1u)];
        const uint32_t dl = ll & 15u, el = (ll >> 4) & 7u;
        const uint32_t valL = ((ll >> 9) & 511u) + ((uint32_t)(w >> dl) & ((1u << el) - 1u));
        vl = (dl + el) | (((ll >> 7) & 3u) << 5) | (valL << 7);
        const uint32_t dD = dd & 15u, eD = (dd >> 4) & 15u;
        const uint32_t valD = ((dd >> 9) & 0xFFFFu) + ((uint32_t)(w >> dD) & ((1u << eD) - 1u));
        vd = (dD + eD) | (((dd >> 8) & 1u) << 5) | (valD << 6);
    }
    // In wave mode every lane holds the same decoder state, but the compiler cannot know (the select trees of the
    // checked path go through opaque registers): readfirstlane marks values as wave-uniform, so that the hop chain below
    // runs on the scalar unit with scalar branches instead of EXEC-mask bookkeeping around every step.
    SWC_HD static uint32_t uni(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
#else
        return x;
#endif
    }
    SWC_HD static uint64_t uni(uint64_t x) { return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | uni((uint32_t)x); }
    SWC_HD static bool uni(bool x) { return uni((uint32_t)x) != 0; }

    // Returns true when the end-of-block symbol was consumed; false: the caller decodes one symbol with careful_step().
    // Inside the loop the positions are 32-bit scalars (the caller guarantees cap < 2^32) and everything that needs more
    // than one store -- a literal run of 255+ in front of a match, the last bytes below the capacity -- is left to the
    // checked step, so that a hop is a handful of scalar instructions.
    SWC_HD bool wave_loop() {
        if (!luts_ready) build_luts();
        uint64_t P = uni(br.consumed_bits());
        stage_base = uni(stage_base); stage_len = uni(stage_len);
        uint32_t p = uni((uint32_t)pos), nl = uni((uint32_t)nlit), le = uni((uint32_t)last_end), nr = uni(nrec);
        const uint32_t lim = uni((uint32_t)cap);   // symbols that start at or beyond cap - 258 go to the checked step
        const uint32_t safe = lim >= 258u ? lim - 258u : 0u;
        bool eob = false, stop = false;
        while (!eob && !stop) {
            // the 64 views need input up to bit P + 63 + 64
            const uint32_t first = (uint32_t)(P >> 3), last = (uint32_t)((P + 63) >> 3) + 8;
            if ((uint64_t)last > br.len) break;                     // tail of the stream: checked steps
            if (!(first >= stage_base && last <= stage_base + stage_len) || stage_len == 0) stage_input(first);
            if (last > stage_base + stage_len) break;
#if defined(__HIP_DEVICE_COMPILE__)
            uint32_t my_l, my_d;
            wave_views(P + (uint64_t)wlane, my_l, my_d);
            auto view_l = [&](uint32_t r) { return (uint32_t)__builtin_amdgcn_readlane((int)my_l, (int)r); };
            auto view_d = [&](uint32_t r) { return (uint32_t)__builtin_amdgcn_readlane((int)my_d, (int)r); };
#else
            uint32_t all_l[kWave], all_d[kWave];
            for (int k = 0; k < kWave; k++) wave_views(P + (uint64_t)k, all_l[k], all_d[k]);
            auto view_l = [&](uint32_t r) { return all_l[r]; };
            auto view_d = [&](uint32_t r) { return all_d[r]; };
#endif
            uint32_t rel = 0;
            for (;;) {
                rel = uni(rel);
                if (rel >= (uint32_t)kWave || p >= safe) { stop = p >= safe; break; }
                const uint32_t a = view_l(rel);
                const uint32_t kind = (a >> 5) & 3u, na = a & 31u, va = a >> 7;
                if (kind == 1) {                       // literal
                    lits[nl] = (uint8_t)va;
                    nl++;
                    p++;
                    rel += na;
                    continue;
                }
                if (kind == 2) {                       // length, then a distance code at rel + na
                    const uint32_t q = rel + na;
                    if (q >= (uint32_t)kWave) break;   // its distance code lies in the next window
                    const uint32_t b = view_d(q);
                    const uint32_t dist = b >> 6, run = p - le;
                    if (!((b >> 5) & 1u) || dist > p || run > lzr::kLitRunMax) { stop = true; break; }
                    if (nr < max_rec) recs[nr] = lzr::make_match(run, va, dist);
                    nr++;
                    p += va;
                    le = p;
                    rel = q + (b & 31u);
                    continue;
                }
                if (kind == 3) {                       // end of block
                    rel += na;
                    eob = true;
                } else {
                    stop = true;                       // not in the table
                }
                break;
            }
            P += rel;
        }
        pos = (pos & ~0xFFFFFFFFull) | p;
        nlit = (nlit & ~0xFFFFFFFFull) | nl;
        last_end = (last_end & ~0xFFFFFFFFull) | le;
        nrec = nr;
        br.seek(P);
        return eob;
    }

    // One symbol with every check of the reference (Deflate.swift:171-236).  Returns SWC_OK to continue, -1 at
    // the end-of-block symbol, or the error.
    SWC_HD int careful_step() {
        br.refill();
        int sym = decode_sym<true>();
        if (sym < 0) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :175
        if (sym < 256) {
            put_byte((uint8_t)sym);
            return SWC_OK;
        }
        if (sym == 256) return -1;
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
        return SWC_OK;
    }

    // Deflate.swift:171-236
    SWC_HD int run_block() {
        const bool fast_ok = !lit.oversub && !dist.oversub && cap >= 272 && cap <= 0xFFFFFFFFull &&
                             (size_t)max_rec >= lzr::max_records(cap);  // the fast loop appends records unchecked
        for (;;) {
            // the fast loop needs the upper halves of pos / last_end to be stable: both below 2^32 - 272 - 258
            if (wlane >= 0) {
                if (!lit.oversub && !dist.oversub && cap <= 0xFFFFFFFFull && pos + 258 < cap && wave_loop()) return SWC_OK;
            } else if (fast_ok && pos + 272 <= cap && (uint64_t)br.ppos + 56 <= br.len) {
                if (fast_loop()) return SWC_OK;
            }
            int st = careful_step();
            if (st == -1) return SWC_OK;
            if (st) return st;
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
        // stored bytes are literals: they go to the literal stream (as far as the capacity reaches)
        uint32_t keep = pos >= cap ? 0u : (cap - pos < length ? (uint32_t)(cap - pos) : length);
        uint32_t i = 0;
        for (; i + 8 <= keep; i += 8) store_u64(lits + nlit + i, load_u64(q + i));
        for (; i < keep; i++) lits[nlit + i] = q[i];
        nlit += keep;
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

// One lane = one job.  `lds` is this lane's view of the wave's table region; `ws` / `ws_bytes` the stream's area in
// the HBM workspace: lzr::StreamHeader | records | literal stream (lz_resolve.h).
// Wave mode: `wlane` = this lane's number and `wave_lds` = kWaveModeLdsBytes of LDS shared by the wave (tables, direct
// lookup tables, staged input); `lds` is then ignored.
SWC_HD void inflate_job(Job& job, LaneLds lds, uint8_t* ws, size_t ws_bytes, int dbg = 0, int wlane = -1, uint32_t* wave_lds = nullptr,
                        int wlanes = kWave) {
    Lane ln;
    ln.dbg = dbg;
    ln.l = lds;
    if (wlane >= 0) {
        ln.wlane = wlane;
        ln.wlanes = wlanes;
        ln.l = LaneLds{wave_lds, 1};
        ln.lut_lit = wave_lds + kWordsPerLane;
        ln.lut_dist = ln.lut_lit + (1 << kLutLitBits);
        ln.stage = (uint8_t*)(ln.lut_dist + (1 << kLutDistBits));
    }
    ln.out = (gptr)job.out;
    ln.cap = job.out_cap;
    ln.pos = 0;
    ln.nrec = 0;
    ln.nlit = 0;
    ln.last_end = 0;
    const size_t lo = ws ? lzr::lit_offset(ws_bytes, job.out_cap) : 0;
    ln.recs = (SWC_AS_GLOBAL uint32_t*)(ws + sizeof(lzr::StreamHeader));
    ln.max_rec = lo > sizeof(lzr::StreamHeader) ? (uint32_t)((lo - sizeof(lzr::StreamHeader)) / 4) : 0u;
    ln.lits = (gptr)(ws + lo);
    int st;
    if (lo == 0) {
        st = SWC_E_NEED_WORKSPACE;   // the workspace area cannot even hold the literal stream of this capacity
        ln.br.init((gcptr)job.in, 0, 0);
    } else if (job.in_len > 0xFFFFFFF0ull) {
        st = SWC_E_INVALID_ARGUMENT;  // streams are addressed with 32-bit byte offsets on device
        ln.br.init((gcptr)job.in, 0, 0);
    } else {
        ln.br.init((gcptr)job.in, (uint32_t)job.in_len, 0);
        st = ln.run();
        ln.flush_tail();
    }
    if (ln.nrec > ln.max_rec) {
        st = SWC_E_NEED_WORKSPACE;  // the record list outgrew the workspace (sized from out_cap)
        ln.nrec = ln.max_rec;
    }
    if (st == SWC_OK && ln.pos > ln.cap) st = SWC_E_CAPACITY;
    if (ws && ws_bytes >= sizeof(lzr::StreamHeader)) {
        SWC_AS_GLOBAL lzr::StreamHeader* h = (SWC_AS_GLOBAL lzr::StreamHeader*)ws;
        h->nrec = ln.nrec;
        h->pad0 = 0;
        h->nlit = ln.nlit;
    }
    uint64_t bits = ln.br.consumed_bits();
    uint64_t consumed = (bits + 7) >> 3;  // callers align() right after (GzipArchive.swift:89)
    job.in_consumed = consumed > job.in_len ? job.in_len : consumed;
    job.out_len = ln.pos;
    job.status = st;
}

}  // namespace inflate
}  // namespace swc
#endif
