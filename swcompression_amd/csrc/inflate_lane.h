// inflate_lane.h -- the serial parts of the Deflate entropy decode (phase 1): bit reader, block headers, stored blocks,
// record / literal bookkeeping.  The symbol loop, the code-length section of a dynamic header and the checked
// one-symbol step live in inflate_sync.h (one stream per wavefront, 64 sub-chunks decoded at once).
//
// Replaces parts of Deflate.decompress(_ bitReader:) (reference Sources/Deflate/Deflate.swift:30-249): the block loop
// (:33-44, 239-246), stored blocks (:45-65) and the head of a dynamic header (:86-116).  The code-length alphabet (19
// symbols, code lengths of up to 7 bits) is kept exactly as the reference's DecodingTree decodes it, over-subscribed sets
// included: for d = 1..7 the heap node at depth d on the path of the next bits is occupied iff
// k0 = ((c >> (15-d)) - first[d]) mod 2^d < count[d]; the last writer among k0, k0 + 2^d, ... wins
// (DecodingTree.swift:22-32), the shallowest occupied node wins (:45).
//
// The sorted symbol arrays (288 + 32 bytes: the lit/len and distance symbols sorted by (code length, symbol); lit/len
// symbols are stored modulo 256 -- within one length the symbols >= 256 come last, so bit 8 is an index compare) sit in
// LDS in the layout `LaneLds` describes (80 words; the wave kernel uses stride 1).
#ifndef SWC_INFLATE_LANE_H
#define SWC_INFLATE_LANE_H

#include "swc_common.h"
#include "lz_resolve.h"
#include "simt.h"

namespace swc {
namespace inflate {

constexpr int W_LIT_SYM = 0;     // 72 words : 288 x 8-bit symbols (symbol & 255)
constexpr int W_DIST_SYM = 72;   //  8 words : 32 x 8-bit symbols
constexpr int kWordsPerLane = 80;
// LSB-first bit reader (BitByteData.LsbBitReader contract) with a 64-bit window
// and one dword of read-ahead so the HBM/L2 latency of the next refill is hidden behind decode work.
// One reader per WAVEFRONT: every lane holds the same state.  The prefetched dword is declared wave-uniform where it
// enters the window (simt::uniform), so the window arithmetic of the serial parts (block headers, code-length section,
// checked steps) runs on the scalar unit and leaves the vector pipes -- the kernel's bottleneck -- to the other waves.
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
            bb |= (uint64_t)simt::uniform(nextw) << bc;
            bc += simt::uniform(nextn) * 8;
            prefetch();
        }
    }
    // refill() for the interior of the stream: the caller guarantees ppos + 4 <= len, so nextn == 4
    SWC_HD void refill_fast() {
        if (bc <= 32) {
            bb |= (uint64_t)simt::uniform(nextw) << bc;
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

SWC_HD uint8_t* sym_ptr(const LaneLds& l, int base, uint32_t i) { return (uint8_t*)(l.p + (size_t)(base + (int)(i >> 2)) * l.stride) + (i & 3u); }

// Phase 1 of the two-phase Deflate path (see lz_resolve.h): literals go straight to their final position,
// every match becomes one record of the stream's record list; the output buffer is never read here.
struct Lane {
    LaneLds l;
    BitReader br;
    gptr out;
    uint64_t cap;
    uint64_t pos;  // bytes produced (keeps counting past cap: size pass for SWC_E_CAPACITY)
    SWC_AS_GLOBAL uint32_t* recs;  // record list in the HBM workspace (the stream's StreamHeader sits 16 bytes before it)
    uint32_t nrec, max_rec;
    gptr lits;                     // dense literal stream in the HBM workspace (16-byte aligned, capacity cap + 16)
    uint64_t nlit;
    uint64_t last_end;             // position just past the previous record
    gptr prov = nullptr;           // scratch of the wave-parallel rounds (lzr::kProvBytes), or none
    int wlane = 0;                 // this lane's number in the wave (the uniform parts run on every lane)
    int wlanes = kWave;            // lanes that share the work of the parallel parts (1 in the host emulation)

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

    SWC_HD void put_byte(uint8_t b) {
        if (pos < cap) lits[nlit++] = b;
        pos++;
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

};

}  // namespace inflate
}  // namespace swc
#endif
