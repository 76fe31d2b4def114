// bzip2_comp.h -- BZip2 COMPRESSION on the device (the third piece of the encode side).
//
// Replaces BZip2.compress(data:blockSize:) (reference Sources/BZip2/BZip2+Compress.swift:40-74), process(_:_:) (:76-241) and
// their helpers -- initialRle :243-262, BurrowsWheeler.transform (BurrowsWheeler.swift:8-29, SuffixArray.swift), mtfRle
// :277-325 -- for ALL blocks of a stream at once (blocks are independent until their bits are joined):
//
//   rle1      one THREAD per input byte around two device scans.  Runs of 4..255 equal bytes become four bytes and a count
//             (:243-262): a maximum scan of the run starts gives every byte its position in its run, a sum scan of what the
//             bytes emit (the byte itself when its position in the 255-byte piece is below four, a count behind the last byte
//             of a piece of four and more) gives the places -- the blocks come out one behind the other;
//   sort      the Burrows-Wheeler transform of the result: the rotations of ALL blocks sorted together by prefix doubling.
//             The first key of a rotation is its block and its first seven bytes (one 62-bit radix sort settles most of a
//             text), then rounds over the rotations that still share a key with a neighbour ONLY: key = (rank of the
//             rotation, rank of the rotation h further on), h = 7, 14, 28 ...; a rotation that is alone in its group has
//             its final row and leaves the working set.  One thread per element, elementwise kernels around a device radix
//             sort and two device scans (rocPRIM: the only library calls of the engine; bzip2_compress.hip);
//   mtf       one WAVEFRONT per segment of 4,096 column bytes.  Move-to-front over the bytes that occur, zero runs as RUNA /
//             RUNB digits (bijective base 2), the end-of-block symbol (:277-325).  The list in front of a segment follows from
//             the last positions of the byte values before it; inside, a byte equal to its predecessor IS a zero, so a ballot
//             per 64 bytes finds the positions that change the list and only those are walked (section "mtf");
//   tables    up to six Huffman tables per block, a table per group of 50 symbols (:89-147), refined the way bzip2 refines them:
//             four times -- every group picks its cheapest table (a thread per group), the symbols of a table's groups are
//             counted, every table gets new code lengths from its counts (a wavefront per table; section "tables");
//   emit      one wavefront per segment: the bit every segment starts at from two small scans, then 64 codes per step -- a
//             wave scan of the code lengths for the places, ds_or_b32 into a staging area of the MSB-first bit stream in LDS,
//             byte-swapped dwords to HBM, straight into the stream.  The first segment of a block writes its header: magic, CRC,
//             origin pointer, the map of used bytes, the selectors after move-to-front in unary, the code lengths in delta form.
//
// The contract is that of the other two encoders: A valid bzip2 stream for the same bytes with the reference's block
// cutting (level x 80,000 raw bytes per block, :46), not the reference's bytes (the reference builds each of its up to six
// tables from ONE group of 50 symbols and keeps it if it beats the others, :95-139; here the tables are refined over all groups).
// Parity = decode(compress(x)) == x under the reference's decoder (the oracle), libbz2 and the engine's own decoder.
//
// The stages are written once, as functors over an EXECUTOR (compress_stream<X>): the device executor (bzip2_compress.hip)
// launches them as kernels, the host emulation (tests/host_emu) runs the same functors as loops with std::sort in place of the
// radix sort, so the CPU test tier covers everything except the three rocPRIM calls.
#ifndef SWC_BZIP2_COMP_H
#define SWC_BZIP2_COMP_H

#include "swc_common.h"
#include "simt.h"

namespace swc {
namespace bz2c {

constexpr uint32_t kMaxSyms = 258;           // 256 list positions shifted by the second run digit + end of block
constexpr uint32_t kStageDw = 256, kFlushDw = 128;
constexpr uint64_t kBlockMagic = 0x314159265359ull;
constexpr uint64_t kEosMagic = 0x177245385090ull;
constexpr uint32_t kBlocksPerLaunch = 64;    // <= 256 (the block index is the top byte of the first sort key)
#ifndef SWC_BZ2C_FIRST_BYTES
#define SWC_BZ2C_FIRST_BYTES 7
#endif
constexpr uint32_t kMaxTables = 6, kGroup = 50, kTableIters = 4, kSymStride = kMaxSyms + 2;
constexpr uint32_t kGroupsPerWave = 1024;    // table choice: groups a wavefront goes through (16 per lane)
constexpr uint32_t kFirstBytes = SWC_BZ2C_FIRST_BYTES;   // 5 / 6 / 7 measured: profiles/r05_experiments.txt

// what a block's stages hand to each other (HBM, one per block)
struct BlockInfo {
    uint32_t n_raw;        // bytes of the block in the input
    uint32_t n_rle;        // ... after rle1 (the length of the sorted column)
    uint32_t orig_ptr;     // row of the unrotated block in the sorted matrix
    uint32_t n_sym;        // symbols after mtf (the end-of-block symbol included)
    uint32_t n_used;       // distinct bytes of the column
    uint32_t crc;          // bzip2 CRC-32 of the block's raw bytes
    uint32_t out_bits;     // bits of the block in its output area
    uint32_t head_bits;    // from the host: bits in front of the first symbol
    uint32_t used[8];      // bit b of word w: byte 32 w + b occurs
    uint32_t freq[kMaxSyms + 2];
    uint32_t n_tables;     // from the host: Huffman tables of the block (2..6)
    uint32_t n_groups;     // ... and groups of 50 symbols (= selectors)
    uint32_t pad[2];
};
SWC_HD uint32_t align16(uint32_t v) { return (v + 15u) & ~15u; }
SWC_HD uint32_t rle1_bound(uint32_t n) { return align16(n + n / 4u + 16u); }
// every symbol at 20 bits; the header: 18002 selector bits, two tables of 258 x 39 bits at most
SWC_HD uint32_t out_bound(uint32_t n_rle) { return align16((n_rle + 16u) / 8u * 20u + 8192u); }

SWC_D void lds_or(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *p |= v;
#endif
}
SWC_D void lds_inc(uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *p += 1u;
#endif
}
SWC_D void global_or(SWC_AS_GLOBAL uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p |= v;
#endif
}
SWC_HD uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }

// ================================================================================================================ rle1
// BZip2+Compress.swift:243-262 over ALL blocks of a launch, one thread per input byte, three steps around two device scans:
//   flags    a byte that differs from its predecessor (or is the first of its block) starts a run: start[i] = i + 1, else 0;
//            (the maximum scan turns that into "first byte of my run + 1" for every byte)
//   counts   with q = my position in my run and p = q % 255 (runs are cut into pieces of 255): I write the byte itself when
//            p < 4, and a count behind me when I am the last of a piece of four and more;
//            (the sum scan turns the counts into places: the blocks end up one behind the other, ready for the sort)
//   write    ... and the first byte of a block leaves the block's offset.
struct Rle1 {
    const SWC_AS_GLOBAL uint8_t* raw;
    uint32_t n, raw_block, nb;
    SWC_AS_GLOBAL uint32_t* start;
    SWC_AS_GLOBAL uint32_t* cnt;
    SWC_AS_GLOBAL uint32_t* pos;
    SWC_AS_GLOBAL uint8_t* text;
    SWC_AS_GLOBAL uint8_t* blk;
    SWC_AS_GLOBAL uint32_t* off;
};
struct Rle1Flags {
    Rle1 c;
    SWC_HD void operator()(uint32_t i) const { c.start[i] = i % c.raw_block == 0u || c.raw[i] != c.raw[i - 1u] ? i + 1u : 0u; }
};
SWC_HD bool rle1_piece_ends(const Rle1& c, uint32_t i, uint32_t p) {
    return p == 254u || i + 1u == c.n || (i + 1u) % c.raw_block == 0u || c.raw[i + 1u] != c.raw[i];
}
struct Rle1Counts {
    Rle1 c;
    SWC_HD void operator()(uint32_t i) const {
        const uint32_t p = (i + 1u - c.start[i]) % 255u;
        c.cnt[i] = (p < 4u ? 1u : 0u) + (p >= 3u && rle1_piece_ends(c, i, p) ? 1u : 0u);
    }
};
struct Rle1Write {
    Rle1 c;
    SWC_HD void operator()(uint32_t i) const {
        const uint32_t p = (i + 1u - c.start[i]) % 255u, b = i / c.raw_block;
        uint32_t o = c.pos[i];
        if (i % c.raw_block == 0u) c.off[b] = o;
        if (i + 1u == c.n) c.off[c.nb] = o + c.cnt[i];
        if (p < 4u) { c.text[o] = c.raw[i]; c.blk[o] = (uint8_t)b; o++; }
        if (p >= 3u && rle1_piece_ends(c, i, p)) { c.text[o] = (uint8_t)(p - 3u); c.blk[o] = (uint8_t)b; }
    }
};

// ================================================================================================================ mtf
// BZip2+Compress.swift:277-325, a block's column cut into SEGMENTS of kSeg bytes that are coded by a wavefront each:
//   * the list in front of a segment is known without coding what precedes it: the bytes seen so far in order of their LAST
//     occurrence, then the bytes not seen yet in ascending order.  seg_last (a wavefront per segment) notes the last position
//     of every byte value inside the segment, seg_before (a thread per block and byte value) turns that into "last position
//     before the segment" by a running maximum over the block's segments -- and into the set of bytes the block uses;
//   * a zero run must not be cut: a segment BEGINS at the first byte at or behind its nominal start that differs from its
//     predecessor and ends where the next one begins, so runs lie inside segments and a segment's symbols depend on nothing else;
//   * inside a segment: a byte equal to its predecessor IS a zero, so a ballot per 64 bytes finds the positions that change the
//     list and only those are walked; the list's first 64 positions live one per lane in a register (finding a byte is a
//     compare and a ballot, moving it to the front one DPP shift), positions 64..255 in three more registers that are touched only
//     when the byte is found that deep.
// Symbols go to the segment's own area (kSegSyms entries), their number to seg_nsym, their frequencies to the block's record.
constexpr uint32_t kSeg = 4096;
constexpr uint32_t kSegSyms = kSeg + 64;     // the changes of the segment, the digits of the runs between them, of one long run behind, end of block

struct Segs {
    const SWC_AS_GLOBAL uint8_t* col;         // the last column, block after block
    const SWC_AS_GLOBAL uint32_t* off;        // first position of every block
    const SWC_AS_GLOBAL uint32_t* seg_off;    // first segment of every block, n_blocks + 1
    const SWC_AS_GLOBAL uint32_t* seg_blk;    // block of every segment
    SWC_AS_GLOBAL uint32_t* last;             // [segment][256]: last position + 1 of the byte value inside the segment, 0 = none
    SWC_AS_GLOBAL uint32_t* before;           // [segment][256]: ... anywhere in front of the segment
    SWC_AS_GLOBAL uint8_t* used;              // [block][256]
    SWC_AS_GLOBAL uint16_t* syms;             // [segment][kSegSyms]
    SWC_AS_GLOBAL uint32_t* seg_nsym;
    SWC_AS_GLOBAL uint32_t* seg_sym_at;       // symbols of the block in front of the segment
    SWC_AS_GLOBAL uint16_t* csyms;            // the blocks' symbols, one behind the other
    const SWC_AS_GLOBAL uint32_t* sym_off;    // first symbol of every block in csyms, n_blocks + 1
    const SWC_AS_GLOBAL uint32_t* grp_off;    // first group of every block, n_blocks + 1
    SWC_AS_GLOBAL uint8_t* sel;               // table of every group of 50 symbols
    SWC_AS_GLOBAL uint8_t* selmtf;            // ... after move-to-front
    SWC_AS_GLOBAL uint32_t* rfreq;            // [block][table][kSymStride]: how often a table's groups hold a symbol
    SWC_AS_GLOBAL uint8_t* len8;              // [block][table][kSymStride]: code lengths
    SWC_AS_GLOBAL uint32_t* codes;            // [block][table][kSymStride]: canonical code | length << 24
    SWC_AS_GLOBAL uint32_t* seg_bits;         // bits of the segment's symbols
    SWC_AS_GLOBAL uint32_t* seg_at;           // ... and where they start in the block's bit stream
    SWC_AS_GLOBAL uint8_t* stream;            // the bits of all blocks of the launch, one behind the other (zeroed)
    SWC_AS_GLOBAL uint32_t* blk_at;           // bit every block starts at in it, n_blocks + 1
    SWC_AS_GLOBAL BlockInfo* infos;
    uint32_t nb, stream_cap, lead;            // lead: the first block starts at this bit (the stream so far ends inside a dword)
    uint32_t cost_waves;                      // table choice: wavefronts per block in the grid
};
SWC_D void lds_max(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    if (v > *p) *p = v;
#endif
}
SWC_D void global_add(SWC_AS_GLOBAL uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p += v;
#endif
}
struct NoLds { uint32_t unused; };
struct SegLastLds { uint32_t tab[256]; };
struct SegLast {       // a wavefront per segment
    Segs c;
    typedef SegLastLds Lds;
    template <int N> SWC_D void run(uint32_t s, Lds* lds) const {
        const uint32_t b = c.seg_blk[s], base = c.off[b], n = c.off[b + 1u] - base;
        const uint32_t lo = (s - c.seg_off[b]) * kSeg, hi = lo + kSeg < n ? lo + kSeg : n;
        SIMT_BEGIN(t, N) for (uint32_t i = (uint32_t)t; i < 256u; i += (uint32_t)N) lds->tab[i] = 0u; SIMT_END_WAVE
        SIMT_BEGIN(t, N)
            for (uint32_t i = lo + (uint32_t)t; i < hi; i += (uint32_t)N) lds_max(&lds->tab[c.col[base + i]], i + 1u);
        SIMT_END_WAVE
        SIMT_BEGIN(t, N) for (uint32_t i = (uint32_t)t; i < 256u; i += (uint32_t)N) c.last[256u * s + i] = lds->tab[i]; SIMT_END
    }
};
struct SegBefore {     // a thread per (block, byte value)
    Segs c;
    SWC_HD void operator()(uint32_t i) const {
        const uint32_t b = i >> 8, v = i & 255u;
        uint32_t run = 0;
        for (uint32_t s = c.seg_off[b]; s < c.seg_off[b + 1u]; s++) {
            const uint32_t l = c.last[256u * s + v];
            c.before[256u * s + v] = run;
            run = l > run ? l : run;
        }
        c.used[i] = run != 0u ? 1 : 0;
    }
};

struct MtfLds {
    uint32_t freq[kMaxSyms + 6];
    uint32_t list[256];
    uint32_t key[256];
};
template <int N>
struct MtfOut {
    MtfLds* l;
    SWC_AS_GLOBAL uint16_t* out;
    simt::PT<uint32_t, N> stage;   // lane k: the k-th symbol not yet written
    uint32_t nsym, nstage;
    SWC_D void put(uint32_t s) {
        const uint32_t k = nstage;
        SIMT_BEGIN(t, N) stage[t] = (uint32_t)t == k ? s : stage[t]; SIMT_END
        nstage++;
        if (nstage == (uint32_t)N) flush();
    }
    SWC_D void flush() {
        const uint32_t k = nstage, o = nsym;
        SIMT_BEGIN(t, N)
            if ((uint32_t)t < k) { out[o + (uint32_t)t] = (uint16_t)stage[t]; lds_inc(&l->freq[stage[t]]); }
        SIMT_END_WAVE
        nsym += k;
        nstage = 0;
    }
    SWC_D void put_run(uint32_t run) {   // :293-307: bijective base 2, least significant digit first; RUNA = 0, RUNB = 1
        while (run != 0u) {
            put((run & 1u) ? 0u : 1u);
            run = (run - 1u) >> 1;
        }
    }
};
// first position >= from (< n) whose byte differs from the one before it; n if there is none
template <int N>
SWC_D uint32_t next_change(gcptr L, uint32_t from, uint32_t n) {
    for (uint32_t base = from; base < n; base += (uint32_t)N) {
        simt::PT<bool, N> ch;
        SIMT_BEGIN(t, N)
            const uint32_t i = base + (uint32_t)t;
            ch[t] = i < n && i > 0u && L[i] != L[i - 1u];
        SIMT_END
        const uint64_t m = simt::wave_ballot<N>(ch);
        if (m != 0ull) return base + (uint32_t)simt::ctz64(m);
    }
    return n;
}
struct MtfSeg {        // a wavefront per segment
    Segs c;
    typedef MtfLds Lds;
    template <int N> SWC_D void run(uint32_t s, Lds* lds) const {
        using simt::PT;
        static_assert(N == 64, "one list position per lane, four registers");
        const uint32_t b = c.seg_blk[s], cbase = c.off[b], n = c.off[b + 1u] - cbase;
        const uint32_t ls = s - c.seg_off[b], nseg = c.seg_off[b + 1u] - c.seg_off[b];
        gcptr L = c.col + cbase;
        SWC_AS_GLOBAL BlockInfo* info = c.infos + b;
        // ---- the list in front of the segment (:279-283 for the first one)
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < kMaxSyms + 6u; i += (uint32_t)N) lds->freq[i] = 0u;
            for (uint32_t v = (uint32_t)t; v < 256u; v += (uint32_t)N) {
                lds->list[v] = 0x1FFu;
                // seen: by last position, descending; then not seen yet: by value, ascending; bytes the block does not use: no place
                lds->key[v] = c.used[256u * b + v] ? ((c.before[256u * s + v] << 8) | (255u - v)) + 1u : 0u;
            }
        SIMT_END_WAVE
        uint32_t used[8];
        for (uint32_t r = 0; r < 4u; r++) {
            PT<bool, N> pr;
            SIMT_BEGIN(t, N) pr[t] = lds->key[64u * r + (uint32_t)t] != 0u; SIMT_END
            const uint64_t m = simt::wave_ballot<N>(pr);
            used[2u * r] = (uint32_t)m;
            used[2u * r + 1u] = (uint32_t)(m >> 32);
        }
        uint32_t n_used = 0;
        for (uint32_t w = 0; w < 8u; w++) n_used += (uint32_t)simt::popc32(used[w]);
        {
            PT<uint32_t, N> k0, k1, k2, k3, r0, r1, r2, r3;
            SIMT_BEGIN(t, N)
                k0[t] = lds->key[t]; k1[t] = lds->key[64 + t]; k2[t] = lds->key[128 + t]; k3[t] = lds->key[192 + t];
                r0[t] = 0u; r1[t] = 0u; r2[t] = 0u; r3[t] = 0u;
            SIMT_END
            for (uint32_t d = 0; d < 256u; d++) {
                SIMT_BEGIN(t, N)
                    const uint32_t kd = lds->key[d];
                    r0[t] += kd > k0[t] ? 1u : 0u; r1[t] += kd > k1[t] ? 1u : 0u; r2[t] += kd > k2[t] ? 1u : 0u; r3[t] += kd > k3[t] ? 1u : 0u;
                SIMT_END
            }
            SIMT_BEGIN(t, N)
                if (k0[t]) lds->list[r0[t]] = (uint32_t)t;
                if (k1[t]) lds->list[r1[t]] = 64u + (uint32_t)t;
                if (k2[t]) lds->list[r2[t]] = 128u + (uint32_t)t;
                if (k3[t]) lds->list[r3[t]] = 192u + (uint32_t)t;
            SIMT_END_WAVE
        }
        if (ls == 0u) {
            SIMT_BEGIN(t, N)
                if (t < 8) {
                    uint32_t uw = 0;
                    for (uint32_t k = 0; k < 8u; k++) uw = (uint32_t)t == k ? used[k] : uw;
                    info->used[t] = uw;
                }
                if (t == 0) info->n_used = n_used;
            SIMT_END
        }
        PT<uint32_t, N> l0, l1, l2, l3;
        SIMT_BEGIN(t, N)
            l0[t] = lds->list[t]; l1[t] = lds->list[64 + t]; l2[t] = lds->list[128 + t]; l3[t] = lds->list[192 + t];
        SIMT_END
        // ---- my part of the column
        const uint32_t start = ls == 0u ? 0u : next_change<N>(L, ls * kSeg, n);
        const uint32_t end = ls + 1u == nseg ? n : next_change<N>(L, (ls + 1u) * kSeg, n);
        MtfOut<N> m;
        m.l = lds; m.out = c.syms + (size_t)kSegSyms * s; m.nsym = 0; m.nstage = 0;
        SIMT_BEGIN(t, N) m.stage[t] = 0u; SIMT_END
        uint32_t run = 0;
        uint32_t carry_b = simt::wave_read<N>(l0, 0);             // the front of the list: a byte equal to it is a zero (:286-288)
        for (uint32_t base = start; base < end; base += (uint32_t)N) {
            PT<uint32_t, N> chunk, prev;
            PT<bool, N> ch;
            SIMT_BEGIN(t, N) chunk[t] = base + (uint32_t)t < end ? L[base + (uint32_t)t] : 0x200u; SIMT_END
            simt::wave_shift_up<N>(prev, chunk, carry_b);
            const uint32_t k1 = end - base < (uint32_t)N ? end - base : (uint32_t)N;
            SIMT_BEGIN(t, N) ch[t] = (uint32_t)t < k1 && chunk[t] != prev[t]; SIMT_END
            uint64_t cm = simt::wave_ballot<N>(ch);
            uint32_t pos = 0;
            while (cm != 0ull) {
                const uint32_t k = (uint32_t)simt::ctz64(cm);
                cm &= cm - 1ull;
                run += k - pos;
                pos = k + 1u;
                const uint32_t v = simt::wave_read<N>(chunk, (int)k);
                m.put_run(run);
                run = 0;
                // where is it?  (never at the front: that is the byte before it)
                PT<bool, N> e;
                SIMT_BEGIN(t, N) e[t] = l0[t] == v; SIMT_END
                uint64_t bal = simt::wave_ballot<N>(e);
                if (bal != 0ull) {
                    const uint32_t li = (uint32_t)simt::ctz64(bal);
                    m.put(li + 1u);                                               // :309-315
                    PT<uint32_t, N> s0;
                    simt::wave_shift_up_dpp<N>(s0, l0, v);
                    SIMT_BEGIN(t, N) l0[t] = (uint32_t)t <= li ? s0[t] : l0[t]; SIMT_END
                    continue;
                }
                uint32_t idx;
                SIMT_BEGIN(t, N) e[t] = l1[t] == v; SIMT_END
                bal = simt::wave_ballot<N>(e);
                if (bal != 0ull) idx = 64u + (uint32_t)simt::ctz64(bal);
                else {
                    SIMT_BEGIN(t, N) e[t] = l2[t] == v; SIMT_END
                    bal = simt::wave_ballot<N>(e);
                    if (bal != 0ull) idx = 128u + (uint32_t)simt::ctz64(bal);
                    else {
                        SIMT_BEGIN(t, N) e[t] = l3[t] == v; SIMT_END
                        bal = simt::wave_ballot<N>(e);
                        idx = 192u + (uint32_t)simt::ctz64(bal | (1ull << 63));
                    }
                }
                m.put(idx + 1u);
                // move to the front: the positions below idx move up by one
                const uint32_t r = idx >> 6, li = idx & 63u;
                const uint32_t e0 = simt::wave_read<N>(l0, N - 1), e1 = simt::wave_read<N>(l1, N - 1), e2 = simt::wave_read<N>(l2, N - 1);
                PT<uint32_t, N> s0, s1, s2, s3;
                simt::wave_shift_up_dpp<N>(s0, l0, v);
                simt::wave_shift_up_dpp<N>(s1, l1, e0);
                simt::wave_shift_up_dpp<N>(s2, l2, e1);
                simt::wave_shift_up_dpp<N>(s3, l3, e2);
                SIMT_BEGIN(t, N)
                    const uint32_t tt = (uint32_t)t;
                    l0[t] = s0[t];
                    l1[t] = r > 1u || tt <= li ? s1[t] : l1[t];
                    l2[t] = r > 2u || (r == 2u && tt <= li) ? s2[t] : l2[t];
                    l3[t] = r == 3u && tt <= li ? s3[t] : l3[t];
                SIMT_END
            }
            run += k1 - pos;
            carry_b = simt::wave_read<N>(chunk, (int)(k1 - 1u));
        }
        m.put_run(run);
        if (ls + 1u == nseg) m.put(n_used + 1u);                          // the end-of-block symbol (:322-323)
        if (m.nstage) m.flush();
        const uint32_t nsym = m.nsym;
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < kMaxSyms + 2u; i += (uint32_t)N) if (lds->freq[i]) global_add(&info->freq[i], lds->freq[i]);
            if (t == 0) { c.seg_nsym[s] = nsym; global_add(&info->n_sym, nsym); }
        SIMT_END_WAVE
    }
};

// ================================================================================================================ tables
// BZip2+Compress.swift:89-147 chooses among up to six Huffman tables per group of 50 symbols.  The reference builds a table from
// one group and keeps it if it beats the tables so far; here the tables are REFINED the way bzip2 itself does it: start from
// n tables that each favour a range of the alphabet (equal shares of the symbol counts), then four times -- every group picks
// the table that codes it in the fewest bits (a thread per group: the six lengths of a symbol sit packed in two LDS words,
// a group's six costs are two sums), the symbols of a table's groups are counted, and every table gets new code lengths
// from its counts (a wavefront per table: rank sort, the two-queue Huffman construction by one lane, depths and canonical
// codes by all).  The symbols of a block must lie one behind the other for that (groups do not respect segments): sym_scan /
// sym_compact.
template <int N>
SWC_D void wave_excl_scan_to(const SWC_AS_GLOBAL uint32_t* in, SWC_AS_GLOBAL uint32_t* out, uint32_t lo, uint32_t hi, uint32_t& carry) {
    for (uint32_t s0 = lo; s0 < hi; s0 += (uint32_t)N) {
        simt::PT<uint32_t, N> x, own;
        SIMT_BEGIN(t, N) own[t] = s0 + (uint32_t)t < hi ? in[s0 + (uint32_t)t] : 0u; x[t] = own[t]; SIMT_END
        simt::wave_scan_incl<N>(x);
        const uint32_t at = carry;
        SIMT_BEGIN(t, N) if (s0 + (uint32_t)t < hi) out[s0 + (uint32_t)t] = at + x[t] - own[t]; SIMT_END
        carry += simt::wave_read<N>(x, N - 1);
    }
}
struct SymScan {       // a wavefront per block
    Segs c;
    typedef NoLds Lds;
    template <int N> SWC_D void run(uint32_t b, Lds*) const {
        uint32_t at = 0;
        wave_excl_scan_to<N>(c.seg_nsym, c.seg_sym_at, c.seg_off[b], c.seg_off[b + 1u], at);
    }
};
struct SymCompact {    // a wavefront per segment
    Segs c;
    typedef NoLds Lds;
    template <int N> SWC_D void run(uint32_t s, Lds*) const {
        const uint32_t n = c.seg_nsym[s];
        const SWC_AS_GLOBAL uint16_t* src = c.syms + (size_t)kSegSyms * s;
        SWC_AS_GLOBAL uint16_t* dst = c.csyms + c.sym_off[c.seg_blk[s]] + c.seg_sym_at[s];
        SIMT_BEGIN(t, N) for (uint32_t i = (uint32_t)t; i < n; i += (uint32_t)N) dst[i] = src[i]; SIMT_END
    }
};
struct CostLds {
    uint32_t plen[kSymStride][2];             // lengths of tables 0-2 / 3-5, ten bits each (a group's sums stay below 1,024)
    uint32_t rfreq[kMaxTables][kSymStride];
};
struct GroupCost {     // a wavefront per kGroupsPerWave groups of a block
    Segs c;
    typedef CostLds Lds;
    template <int N> SWC_D void run(uint32_t idx, Lds* lds) const {
        const uint32_t b = idx / c.cost_waves, w = idx % c.cost_waves;
        const SWC_AS_GLOBAL BlockInfo* info = c.infos + b;
        const uint32_t n_groups = info->n_groups, n_tables = info->n_tables, n_sym = info->n_sym, alpha = info->n_used + 2u;
        const uint32_t g0 = w * kGroupsPerWave;
        if (g0 >= n_groups) return;
        const SWC_AS_GLOBAL uint8_t* len8 = c.len8 + (size_t)b * kMaxTables * kSymStride;
        SIMT_BEGIN(t, N)
            for (uint32_t sy = (uint32_t)t; sy < kSymStride; sy += (uint32_t)N) {
                uint32_t p0 = 0, p1 = 0;
                for (uint32_t k = 0; k < 3u; k++) {
                    p0 |= (sy < alpha && k < n_tables ? (uint32_t)len8[k * kSymStride + sy] : 20u) << (10u * k);
                    p1 |= (sy < alpha && k + 3u < n_tables ? (uint32_t)len8[(k + 3u) * kSymStride + sy] : 20u) << (10u * k);
                }
                lds->plen[sy][0] = p0; lds->plen[sy][1] = p1;
            }
            for (uint32_t i = (uint32_t)t; i < kMaxTables * kSymStride; i += (uint32_t)N) (&lds->rfreq[0][0])[i] = 0u;
        SIMT_END_WAVE
        const SWC_AS_GLOBAL uint16_t* sy = c.csyms + c.sym_off[b];
        SWC_AS_GLOBAL uint8_t* sel = c.sel + c.grp_off[b];
        SIMT_BEGIN(t, N)
            for (uint32_t g = g0 + (uint32_t)t; g < n_groups && g < g0 + kGroupsPerWave; g += (uint32_t)N) {
                const uint32_t lo = g * kGroup, hi = lo + kGroup < n_sym ? lo + kGroup : n_sym;
                uint32_t c0 = 0, c1 = 0;
                for (uint32_t i = lo; i < hi; i++) { const uint32_t v = sy[i]; c0 += lds->plen[v][0]; c1 += lds->plen[v][1]; }
                uint32_t best = 0, cost = c0 & 1023u;
                for (uint32_t k = 1; k < n_tables; k++) {
                    const uint32_t ck = ((k < 3u ? c0 : c1) >> (10u * (k % 3u))) & 1023u;
                    if (ck < cost) { cost = ck; best = k; }
                }
                sel[g] = (uint8_t)best;
                for (uint32_t i = lo; i < hi; i++) lds_inc(&lds->rfreq[best][sy[i]]);
            }
        SIMT_END_WAVE
        SWC_AS_GLOBAL uint32_t* rf = c.rfreq + (size_t)b * kMaxTables * kSymStride;
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < kMaxTables * kSymStride; i += (uint32_t)N) {
                const uint32_t v = (&lds->rfreq[0][0])[i];
                if (v) global_add(rf + i, v);
            }
        SIMT_END
    }
};
struct HuffLds {
    uint32_t w[kSymStride];          // weights
    uint32_t order[kSymStride];      // symbols by weight
    uint32_t wt[kSymStride];         // inner nodes, in order of creation (= of weight)
    uint32_t parent_leaf[kSymStride];
    uint32_t parent_inner[kSymStride];
    uint32_t depth_inner[kSymStride];
    uint32_t len[kSymStride];
    uint32_t count[24], base[24];
};
// Code lengths (no code longer than max_len: weights halved until that holds, as bzip2's hbMakeCodeLengths does) and canonical
// codes of one table from the counts of its symbols; a symbol that does not occur counts as one.
template <int N>
SWC_D void huffman_wave(HuffLds* l, uint32_t alpha, uint32_t max_len) {
    using simt::PT;
    for (;;) {
        // ---- the symbols in order of weight (ties: by symbol): every lane ranks its symbols against all
        SIMT_BEGIN(t, N)
            for (uint32_t s = (uint32_t)t; s < alpha; s += (uint32_t)N) {
                const uint32_t ws = l->w[s];
                uint32_t r = 0;
                for (uint32_t k = 0; k < alpha; k++) { const uint32_t wk = l->w[k]; r += wk < ws || (wk == ws && k < s) ? 1u : 0u; }
                l->order[r] = s;
            }
        SIMT_END_WAVE
        // ---- two queues: the lightest two of (next leaf, next inner node) are joined; inner nodes come into being in order of weight
        uint32_t made = 0;
        SIMT_BEGIN(t, N)
            if (t == 0) {
                uint32_t nl = 0, ni = 0, m = 0;
                while (alpha - nl + m - ni > 1u) {
                    uint32_t sum = 0;
                    for (int k = 0; k < 2; k++) {
                        if (nl < alpha && (ni >= m || l->w[l->order[nl]] <= l->wt[ni])) { const uint32_t s = l->order[nl++]; l->parent_leaf[s] = m; sum += l->w[s]; }
                        else { l->parent_inner[ni] = m; sum += l->wt[ni++]; }
                    }
                    l->wt[m] = sum;
                    l->parent_inner[m] = 0xFFFFFFFFu;
                    m++;
                }
                for (uint32_t k = m; k-- > 0u;) l->depth_inner[k] = l->parent_inner[k] == 0xFFFFFFFFu ? 0u : l->depth_inner[l->parent_inner[k]] + 1u;
                l->count[23] = m;
            }
        SIMT_END_WAVE
        made = l->count[23];
        PT<uint32_t, N> mx;
        SIMT_BEGIN(t, N)
            uint32_t m = 0;
            for (uint32_t s = (uint32_t)t; s < alpha; s += (uint32_t)N) {
                const uint32_t d = made ? l->depth_inner[l->parent_leaf[s]] + 1u : 1u;
                l->len[s] = d;
                m = d > m ? d : m;
            }
            mx[t] = m;
        SIMT_END_WAVE
        simt::wave_scan_max_incl<N>(mx);
        if (simt::wave_read<N>(mx, N - 1) <= max_len) break;
        SIMT_BEGIN(t, N) for (uint32_t s = (uint32_t)t; s < alpha; s += (uint32_t)N) l->w[s] = l->w[s] / 2u + 1u; SIMT_END_WAVE
    }
    // ---- canonical codes: in order of (length, symbol)
    SIMT_BEGIN(t, N) if (t < 24) l->count[t] = 0u; SIMT_END_WAVE
    SIMT_BEGIN(t, N) for (uint32_t s = (uint32_t)t; s < alpha; s += (uint32_t)N) lds_inc(&l->count[l->len[s]]); SIMT_END_WAVE
    SIMT_BEGIN(t, N)
        if (t == 0) {
            uint32_t next = 0;
            for (uint32_t k = 1; k <= max_len; k++) { l->base[k] = next; next = (next + l->count[k]) << 1; }
        }
    SIMT_END_WAVE
    SIMT_BEGIN(t, N)
        for (uint32_t s = (uint32_t)t; s < alpha; s += (uint32_t)N) {
            const uint32_t ls = l->len[s];
            uint32_t r = 0;
            for (uint32_t k = 0; k < s; k++) r += l->len[k] == ls ? 1u : 0u;
            l->wt[s] = (l->base[ls] + r) | (ls << 24);        // (the inner weights are not needed any more)
        }
    SIMT_END_WAVE
}
struct Huff {          // a wavefront per (block, table)
    Segs c;
    typedef HuffLds Lds;
    template <int N> SWC_D void run(uint32_t idx, Lds* lds) const {
        const uint32_t b = idx / kMaxTables, k = idx % kMaxTables;
        const SWC_AS_GLOBAL BlockInfo* info = c.infos + b;
        if (k >= info->n_tables) return;
        const uint32_t alpha = info->n_used + 2u;
        const size_t at = ((size_t)b * kMaxTables + k) * kSymStride;
        SIMT_BEGIN(t, N)
            for (uint32_t s = (uint32_t)t; s < alpha; s += (uint32_t)N) { const uint32_t f = c.rfreq[at + s]; lds->w[s] = f ? f : 1u; }
        SIMT_END_WAVE
        huffman_wave<N>(lds, alpha, 17u);
        SIMT_BEGIN(t, N)
            for (uint32_t s = (uint32_t)t; s < alpha; s += (uint32_t)N) { c.len8[at + s] = (uint8_t)lds->len[s]; c.codes[at + s] = lds->wt[s]; }
        SIMT_END
    }
};

// ================================================================================================================ emit
// BZip2+Compress.swift:149-240, a wavefront per SEGMENT again: block_head (a wavefront per block) moves the selectors to the
// front -- the position of a selector in the list is the number of tables used more recently than its own, six running
// maxima -- and adds up the bits in front of the block's first symbol; seg_bits (a wavefront per segment) adds up the code
// lengths of a segment's symbols, seg_scan / block_scan turn that into the bit every segment and block starts at in the
// STREAM, and emit_seg writes a segment's codes -- the first segment of a block the header in front of them -- 64 symbols per
// step: a wave scan of the code lengths for the places, ds_or_b32 into a staging area of the MSB-first bit stream in LDS,
// byte-swapped dwords to HBM, the dwords a segment shares with its neighbours by atomic OR (the stream is zeroed).
struct EmitLds {
    uint32_t stage[kStageDw + 4];
    uint32_t code[kMaxTables][kSymStride];
};
template <int N>
struct Emitter {
    EmitLds* l;
    gptr out;          // 16-byte aligned, `cap` a multiple of four
    uint32_t cap;
    uint32_t obits, odw, fill, first_dw;

    SWC_D void begin(EmitLds* lds, gptr o, uint32_t c, uint32_t at) {
        l = lds; out = o; cap = c; obits = 0; odw = at >> 5; fill = at & 31u; first_dw = at >> 5;
    }
    SWC_D void flush(bool all) {
        const uint32_t nd = all ? (fill + 31u) >> 5 : fill >> 5;
        const uint32_t o0 = odw, f = first_dw;
        simt::PT<uint32_t, N> carry;
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < nd; i += (uint32_t)N) {
                const uint32_t b = 4u * (o0 + i);
                const uint32_t v = bswap32(l->stage[i]);        // the stream's first bit is the top bit of byte 0
                if (b + 4u > cap) continue;
                if (o0 + i == f || (all && i + 1u == nd)) global_or((SWC_AS_GLOBAL uint32_t*)(out + b), v);   // shared with a neighbour
                else *(SWC_AS_GLOBAL uint32_t*)(out + b) = v;
            }
            carry[t] = l->stage[nd];
        SIMT_END_WAVE
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < kStageDw + 4u; i += (uint32_t)N) l->stage[i] = i == 0u && !all ? carry[t] : 0u;
        SIMT_END_WAVE
        odw += nd;
        fill = all ? 0u : fill & 31u;
    }
    // every lane adds the low `nb` bits (0..32) of `code`, most significant first, lane after lane
    SWC_D void emit(const simt::PT<uint32_t, N>& code, const simt::PT<uint32_t, N>& nb) {
        simt::PT<uint32_t, N> x;
        SIMT_BEGIN(t, N) x[t] = nb[t]; SIMT_END
        simt::wave_scan_incl<N>(x);
        const uint32_t f0 = fill;
        SIMT_BEGIN(t, N)
            const uint32_t n = nb[t];
            if (n != 0u) {
                const uint32_t b = f0 + x[t] - n, d = b >> 5, s = b & 31u;      // s bits of dword d are taken
                const uint32_t c = n == 32u ? code[t] : code[t] & ((1u << n) - 1u);
                if (s + n <= 32u) lds_or(&l->stage[d], c << (32u - s - n));
                else {
                    const uint32_t lo = s + n - 32u;                              // bits that go to the next dword
                    lds_or(&l->stage[d], c >> lo);
                    lds_or(&l->stage[d + 1u], c << (32u - lo));
                }
            }
        SIMT_END_WAVE
        const uint32_t total = simt::wave_read<N>(x, N - 1);
        fill += total;
        obits += total;
        if (fill >= 32u * kFlushDw) flush(false);
    }
};
// A wavefront per block: the selectors after move-to-front (:187-194, mtf :265-275), and the bits in front of the block's first symbol.
struct BlockHead {
    Segs c;
    typedef NoLds Lds;
    template <int N> SWC_D void run(uint32_t b, Lds*) const {
        using simt::PT;
        SWC_AS_GLOBAL BlockInfo* info = c.infos + b;
        const uint32_t n_groups = info->n_groups, n_tables = info->n_tables, alpha = info->n_used + 2u;
        const SWC_AS_GLOBAL uint8_t* sel = c.sel + c.grp_off[b];
        SWC_AS_GLOBAL uint8_t* out = c.selmtf + c.grp_off[b];
        // key of table v: where it was used last (+ 8), or -- not used yet -- its place from the back of the list 0 1 2 3 4 5
        uint32_t key[kMaxTables];
        for (uint32_t v = 0; v < kMaxTables; v++) key[v] = kMaxTables - v;
        uint32_t sel_bits = 0;
        for (uint32_t g0 = 0; g0 < n_groups; g0 += (uint32_t)N) {
            PT<uint32_t, N> mine, k0, k1, k2, k3, k4, k5, pos;
            SIMT_BEGIN(t, N) mine[t] = g0 + (uint32_t)t < n_groups ? sel[g0 + (uint32_t)t] : 0xFFu; SIMT_END
            PT<uint32_t, N>* ks[kMaxTables] = {&k0, &k1, &k2, &k3, &k4, &k5};
            for (uint32_t v = 0; v < kMaxTables; v++) {
                PT<uint32_t, N> x, ex;
                SIMT_BEGIN(t, N) x[t] = mine[t] == v ? g0 + (uint32_t)t + 8u : 0u; SIMT_END
                simt::wave_scan_max_incl<N>(x);
                simt::wave_shift_up<N>(ex, x, 0u);                                 // what lies in front of me
                const uint32_t kv = key[v];
                SIMT_BEGIN(t, N) (*ks[v])[t] = ex[t] > kv ? ex[t] : kv; SIMT_END
                const uint32_t last = simt::wave_read<N>(x, N - 1);
                key[v] = last > kv ? last : kv;
            }
            SIMT_BEGIN(t, N)
                const uint32_t m = mine[t];
                const uint32_t kk[kMaxTables] = {k0[t], k1[t], k2[t], k3[t], k4[t], k5[t]};
                uint32_t own = 0, p = 0;
                for (uint32_t v = 0; v < kMaxTables; v++) own = m == v ? kk[v] : own;
                for (uint32_t v = 0; v < kMaxTables; v++) p += kk[v] > own ? 1u : 0u;
                pos[t] = m < kMaxTables ? p : 0u;
                if (m < kMaxTables) out[g0 + (uint32_t)t] = (uint8_t)p;
            SIMT_END
            PT<uint32_t, N> x;
            SIMT_BEGIN(t, N) x[t] = mine[t] < kMaxTables ? pos[t] + 1u : 0u; SIMT_END
            simt::wave_scan_incl<N>(x);
            sel_bits += simt::wave_read<N>(x, N - 1);
        }
        // the tables in delta form (:196-221): 5 bits, then per symbol two bits per step and a closing zero
        const SWC_AS_GLOBAL uint8_t* len8 = c.len8 + (size_t)b * kMaxTables * kSymStride;
        PT<uint32_t, N> tb;
        SIMT_BEGIN(t, N)
            uint32_t a = 0;
            for (uint32_t k = 0; k < n_tables; k++)
                for (uint32_t s = (uint32_t)t; s < alpha; s += (uint32_t)N) {
                    const uint32_t len = len8[k * kSymStride + s], prev = s ? len8[k * kSymStride + s - 1u] : len;
                    a += 2u * (len > prev ? len - prev : prev - len) + 1u;
                }
            tb[t] = a;
        SIMT_END
        simt::wave_scan_incl<N>(tb);
        uint32_t ranges = 0;
        for (uint32_t r = 0; r < 16u; r++) if ((info->used[r >> 1] >> (16u * (r & 1u))) & 0xFFFFu) ranges++;
        const uint32_t bits = 48u + 32u + 1u + 24u + 16u + 16u * ranges + 3u + 15u + sel_bits + 5u * n_tables + simt::wave_read<N>(tb, N - 1);
        SIMT_BEGIN(t, N) if (t == 0) info->head_bits = bits; SIMT_END
    }
};
// The block from its magic to the last bit in front of its first symbol.
template <int N>
SWC_D void emit_header(Emitter<N>& e, const Segs& c, uint32_t b) {
    using simt::PT;
    const SWC_AS_GLOBAL BlockInfo* info = c.infos + b;
    const uint32_t alpha = info->n_used + 2u, n_tables = info->n_tables, n_groups = info->n_groups;
    PT<uint32_t, N> code, nb;
    // magic (48), block CRC (32), randomised = 0 (1), origin pointer (24), the 16 bits of the used ranges (:149-166)
    uint32_t ranges = 0;
    for (uint32_t r = 0; r < 16u; r++) if ((info->used[r >> 1] >> (16u * (r & 1u))) & 0xFFFFu) ranges |= 1u << (15u - r);
    const uint32_t crc = info->crc, optr = info->orig_ptr;
    SIMT_BEGIN(t, N)
        uint32_t cc = 0, n = 0;
        switch (t) {
            case 0: cc = (uint32_t)(kBlockMagic >> 24); n = 24; break;
            case 1: cc = (uint32_t)(kBlockMagic & 0xFFFFFFu); n = 24; break;
            case 2: cc = crc; n = 32; break;
            case 3: cc = 0; n = 1; break;
            case 4: cc = optr; n = 24; break;
            case 5: cc = ranges; n = 16; break;
            default: break;
        }
        code[t] = cc; nb[t] = n;
    SIMT_END
    e.emit(code, nb);
    // the 16 bits of every used range (:168-178)
    SIMT_BEGIN(t, N)
        uint32_t cc = 0, n = 0;
        if ((uint32_t)t < 16u && ((ranges >> (15u - (uint32_t)t)) & 1u)) {
            const uint32_t w = (info->used[t >> 1] >> (16u * ((uint32_t)t & 1u))) & 0xFFFFu;   // bit b: byte 16 t + b
            cc = brev32(w) >> 16;                                                               // byte 16 t first
            n = 16;
        }
        code[t] = cc; nb[t] = n;
    SIMT_END
    e.emit(code, nb);
    // the number of tables and of selectors, the selectors in unary (:180-194)
    SIMT_BEGIN(t, N)
        code[t] = t == 0 ? n_tables : t == 1 ? n_groups : 0u;
        nb[t] = t == 0 ? 3u : t == 1 ? 15u : 0u;
    SIMT_END
    e.emit(code, nb);
    const SWC_AS_GLOBAL uint8_t* sm = c.selmtf + c.grp_off[b];
    for (uint32_t g0 = 0; g0 < n_groups; g0 += (uint32_t)N) {
        SIMT_BEGIN(t, N)
            const uint32_t p = g0 + (uint32_t)t < n_groups ? sm[g0 + (uint32_t)t] : 0xFFu;
            code[t] = p < kMaxTables ? (2u << p) - 2u : 0u;        // p ones and a zero
            nb[t] = p < kMaxTables ? p + 1u : 0u;
        SIMT_END
        e.emit(code, nb);
    }
    // the code lengths of the tables in delta form (:196-221): 5 bits of the first length, then per symbol "10" / "11" steps and a 0
    const SWC_AS_GLOBAL uint8_t* len8 = c.len8 + (size_t)b * kMaxTables * kSymStride;
    for (uint32_t k = 0; k < n_tables; k++) {
        const SWC_AS_GLOBAL uint8_t* l8 = len8 + k * kSymStride;
        const uint32_t first = l8[0];
        SIMT_BEGIN(t, N) code[t] = t == 0 ? first : 0u; nb[t] = t == 0 ? 5u : 0u; SIMT_END
        e.emit(code, nb);
        for (uint32_t s0 = 0; s0 < alpha; s0 += 32u) {     // a symbol takes at most 2 * 19 + 1 = 39 bits: two lanes per symbol
            SIMT_BEGIN(t, N)
                const uint32_t s = s0 + ((uint32_t)t >> 1);
                uint32_t cc = 0, n = 0;
                if (s < alpha) {
                    const uint32_t len = l8[s], prev = s == 0u ? len : l8[s - 1u];
                    const uint32_t d = len > prev ? len - prev : prev - len;       // <= 19
                    const uint32_t pat = len > prev ? 0xAAAAAAAAu : 0xFFFFFFFFu;   // "10" steps up, "11" steps down
                    if (((uint32_t)t & 1u) == 0u) {        // the first min(d, 16) steps
                        const uint32_t q = d < 16u ? d : 16u;
                        n = 2u * q;
                        cc = q ? pat >> (32u - n) : 0u;
                    } else {                               // the remaining steps and the closing 0
                        const uint32_t q = d > 16u ? d - 16u : 0u;
                        n = 2u * q + 1u;
                        cc = q ? (pat >> (32u - 2u * q)) << 1 : 0u;
                    }
                }
                code[t] = cc; nb[t] = n;
            SIMT_END
            e.emit(code, nb);
        }
    }
}
template <int N>
SWC_D void load_codes(EmitLds* lds, const Segs& c, uint32_t b) {
    const SWC_AS_GLOBAL uint32_t* codes = c.codes + (size_t)b * kMaxTables * kSymStride;
    const uint32_t n = c.infos[b].n_tables * kSymStride;
    SIMT_BEGIN(t, N)
        for (uint32_t i = (uint32_t)t; i < kStageDw + 4u; i += (uint32_t)N) lds->stage[i] = 0u;
        for (uint32_t i = (uint32_t)t; i < n; i += (uint32_t)N) (&lds->code[0][0])[i] = codes[i];
    SIMT_END_WAVE
}
// canonical code | length << 24 of symbol i of segment s (the table: that of the symbol's group of 50 in the block)
SWC_D uint32_t code_of(const EmitLds* lds, const SWC_AS_GLOBAL uint16_t* sy, const SWC_AS_GLOBAL uint8_t* sel, uint32_t first, uint32_t i) {
    return lds->code[sel[(first + i) / kGroup]][sy[i]];
}
struct SegBits {       // a wavefront per segment
    Segs c;
    typedef EmitLds Lds;
    template <int N> SWC_D void run(uint32_t s, Lds* lds) const {
        const uint32_t b = c.seg_blk[s];
        load_codes<N>(lds, c, b);
        const uint32_t ns = c.seg_nsym[s], first = c.seg_sym_at[s];
        const SWC_AS_GLOBAL uint16_t* sy = c.syms + (size_t)kSegSyms * s;
        const SWC_AS_GLOBAL uint8_t* sel = c.sel + c.grp_off[b];
        simt::PT<uint32_t, N> x;
        SIMT_BEGIN(t, N)
            uint32_t a = 0;
            for (uint32_t i = (uint32_t)t; i < ns; i += (uint32_t)N) a += code_of(lds, sy, sel, first, i) >> 24;
            x[t] = a;
        SIMT_END
        simt::wave_scan_incl<N>(x);
        const uint32_t total = simt::wave_read<N>(x, N - 1);
        SIMT_BEGIN(t, N) if (t == 0) c.seg_bits[s] = total; SIMT_END
    }
};
struct SegScan {       // a wavefront per block
    Segs c;
    typedef NoLds Lds;
    template <int N> SWC_D void run(uint32_t b, Lds*) const {
        uint32_t at = c.infos[b].head_bits;
        wave_excl_scan_to<N>(c.seg_bits, c.seg_at, c.seg_off[b], c.seg_off[b + 1u], at);
        const uint32_t bits = at;
        SIMT_BEGIN(t, N) if (t == 0) c.infos[b].out_bits = bits; SIMT_END
    }
};
struct BlockScan {     // one wavefront
    Segs c;
    typedef NoLds Lds;
    template <int N> SWC_D void run(uint32_t, Lds*) const {
        static_assert(kBlocksPerLaunch <= (uint32_t)N, "one lane per block");
        simt::PT<uint32_t, N> x, own;
        SIMT_BEGIN(t, N) own[t] = (uint32_t)t < c.nb ? c.infos[t].out_bits : 0u; x[t] = own[t]; SIMT_END
        simt::wave_scan_incl<N>(x);
        SIMT_BEGIN(t, N)
            if ((uint32_t)t < c.nb) c.blk_at[t] = c.lead + x[t] - own[t];
            if ((uint32_t)t + 1u == c.nb) c.blk_at[c.nb] = c.lead + x[t];
        SIMT_END
    }
};
struct EmitSeg {       // a wavefront per segment
    Segs c;
    typedef EmitLds Lds;
    template <int N> SWC_D void run(uint32_t s, Lds* lds) const {
        using simt::PT;
        const uint32_t b = c.seg_blk[s];
        load_codes<N>(lds, c, b);
        Emitter<N> e;
        const bool first_seg = s == c.seg_off[b];
        e.begin(lds, c.stream, c.stream_cap, c.blk_at[b] + (first_seg ? 0u : c.seg_at[s]));
        if (first_seg) emit_header<N>(e, c, b);
        // the symbols (:223-240)
        const uint32_t ns = c.seg_nsym[s], first = c.seg_sym_at[s];
        const SWC_AS_GLOBAL uint16_t* sy = c.syms + (size_t)kSegSyms * s;
        const SWC_AS_GLOBAL uint8_t* sel = c.sel + c.grp_off[b];
        PT<uint32_t, N> code, nb;
        for (uint32_t base = 0; base < ns; base += (uint32_t)N) {
            SIMT_BEGIN(t, N)
                uint32_t cc = 0, n = 0;
                if (base + (uint32_t)t < ns) {
                    const uint32_t w = code_of(lds, sy, sel, first, base + (uint32_t)t);
                    cc = w & 0xFFFFFFu; n = w >> 24;
                }
                code[t] = cc; nb[t] = n;
            SIMT_END
            e.emit(code, nb);
        }
        e.flush(true);
    }
};

// ================================================================================================================ sort
// Everything the elementwise steps of the sort see (device pointers).
struct Bwt {
    const SWC_AS_GLOBAL uint8_t* text;    // the blocks after rle1, one behind the other
    const SWC_AS_GLOBAL uint8_t* blk;     // block of every position (and of every ROW: rows are grouped the same way)
    const SWC_AS_GLOBAL uint32_t* off;    // first position of every block, n_blocks + 1
    SWC_AS_GLOBAL uint32_t* rank;         // row (of the first rotation of its group) of every rotation
    SWC_AS_GLOBAL uint32_t* sa;           // rotation of every row
    SWC_AS_GLOBAL uint64_t* key_in;
    SWC_AS_GLOBAL uint64_t* key_out;
    SWC_AS_GLOBAL uint32_t* val_in;
    SWC_AS_GLOBAL uint32_t* val_out;
    SWC_AS_GLOBAL uint32_t* slot;         // rows of the working set, ascending (nullptr: all rows)
    SWC_AS_GLOBAL uint32_t* slot_next;
    SWC_AS_GLOBAL uint32_t* head;         // first element of my key group (after the max scan)
    SWC_AS_GLOBAL uint32_t* keep;         // 1: stays in the working set
    SWC_AS_GLOBAL uint32_t* kpos;         // exclusive prefix sum of keep
    SWC_AS_GLOBAL uint32_t* next_m;       // size of the next working set (one word the host reads each round)
    SWC_AS_GLOBAL uint8_t* col;           // the last column, block after block
    SWC_AS_GLOBAL BlockInfo* infos;
    uint32_t m, h, rank_bits;
};
SWC_HD uint32_t rot(const Bwt& c, uint32_t i, uint32_t h) {   // the rotation h further on, inside i's block
    const uint32_t b = c.blk[i], base = c.off[b], n = c.off[b + 1u] - base;
    uint32_t r = i - base + h;
    if (r >= n) { r -= n; if (r >= n) r %= n; }
    return base + r;
}
struct FirstKeys {     // key: block, the first seven bytes of the rotation
    Bwt c;
    SWC_HD void operator()(uint32_t i) const {
        const uint32_t b = c.blk[i], base = c.off[b], n = c.off[b + 1u] - base;
        uint64_t k = b;
        uint32_t r = i - base;
        for (uint32_t j = 0; j < kFirstBytes; j++) {
            k = (k << 8) | c.text[base + r];
            r++;
            if (r == n) r = 0;
        }
        c.key_in[i] = k;
        c.val_in[i] = i;
    }
};
struct NextKeys {      // key: my rank, the rank of the rotation h further on
    Bwt c;
    SWC_HD void operator()(uint32_t p) const {
        const uint32_t i = c.val_in[p];
        c.key_in[p] = ((uint64_t)c.rank[i] << c.rank_bits) | c.rank[rot(c, i, c.h)];
    }
};
struct Heads {
    Bwt c;
    SWC_HD void operator()(uint32_t p) const { c.head[p] = p != 0u && c.key_out[p] != c.key_out[p - 1u] ? p : 0u; }
};
struct Ranks {         // (after the max scan of head)
    Bwt c;
    SWC_HD void operator()(uint32_t p) const {
        const uint32_t v = c.val_out[p], g = c.head[p];
        c.rank[v] = c.slot ? c.slot[g] : g;
        c.sa[c.slot ? c.slot[p] : p] = v;
        const bool first = g == p, next_first = p + 1u == c.m || c.head[p + 1u] == p + 1u;
        c.keep[p] = first && next_first ? 0u : 1u;
    }
};
struct Compact {       // (after the sum scan of keep)
    Bwt c;
    SWC_HD void operator()(uint32_t p) const {
        if (p + 1u == c.m) *c.next_m = c.kpos[p] + c.keep[p];
        if (!c.keep[p]) return;
        const uint32_t q = c.kpos[p];
        c.val_in[q] = c.val_out[p];
        c.slot_next[q] = c.slot ? c.slot[p] : p;
    }
};
struct LastColumn {
    Bwt c;
    SWC_HD void operator()(uint32_t j) const {
        const uint32_t v = c.sa[j], b = c.blk[j], base = c.off[b], n = c.off[b + 1u] - base;
        c.col[j] = c.text[v == base ? base + n - 1u : v - 1u];
        if (v == base) c.infos[b].orig_ptr = j - base;
    }
};

}  // namespace bz2c
}  // namespace swc

// ================================================================================================================ host
#include <string.h>
#include <vector>
namespace swc {
namespace bz2c {

// How many tables a block of n_sym symbols gets, and the lengths the refinement starts from: table k favours (length 0 against
// 15) a range of the alphabet that holds about an equal share of the symbols -- bzip2's own starting point.
inline uint32_t tables_for(uint32_t n_sym) { return n_sym < 200u ? 2u : n_sym < 600u ? 3u : n_sym < 1200u ? 4u : n_sym < 2400u ? 5u : 6u; }
inline void initial_tables(const uint32_t* freq, uint32_t alpha, uint32_t n_sym, uint32_t n_tables, uint8_t* len8 /* [kMaxTables][kSymStride] */) {
    uint32_t part = n_tables, rem = n_sym;
    int gs = 0;
    while (part > 0) {
        const uint32_t target = rem / part;
        int ge = gs - 1;
        uint32_t acc = 0;
        while (acc < target && ge < (int)alpha - 1) { ge++; acc += freq[ge]; }
        if (ge > gs && part != n_tables && part != 1 && ((n_tables - part) % 2u) == 1u) { acc -= freq[ge]; ge--; }
        for (uint32_t v = 0; v < alpha; v++) len8[(part - 1) * kSymStride + v] = ((int)v >= gs && (int)v <= ge) ? 0 : 15;
        part--;
        gs = ge + 1;
        rem -= acc;
    }
}

// The stream being assembled, MSB first, in memory from the executor's result allocator (the C ABI hands exactly this buffer
// to the caller: no copy of the finished stream).
template <class X>
struct BitSink {
    X& x;
    uint8_t* p = nullptr;
    size_t cap = 0;
    uint64_t bits = 0;
    explicit BitSink(X& ex) : x(ex) {}
    ~BitSink() { if (p) x.result_free(p); }
    bool ensure(size_t n) {
        if (n <= cap) return true;
        const size_t want = n > 2 * cap ? n : 2 * cap;
        uint8_t* q = x.result_alloc(want);
        if (!q) return false;
        if (p) { memcpy(q, p, (size_t)((bits + 7u) >> 3)); x.result_free(p); }
        p = q; cap = want;
        return true;
    }
    bool put(uint64_t v, int n) {
        if (!ensure((size_t)((bits + (uint64_t)n + 7u) >> 3) + 8)) return false;
        for (int i = n - 1; i >= 0; i--) {
            if ((bits & 7u) == 0) p[bits >> 3] = 0;
            if ((v >> i) & 1u) p[bits >> 3] |= (uint8_t)(0x80u >> (bits & 7u));
            bits++;
        }
        return true;
    }
    uint8_t* release(size_t* n) { uint8_t* r = p; *n = (size_t)((bits + 7u) >> 3); p = nullptr; cap = 0; return r; }
};

// The whole stream.  X is the executor: begin_chunk / end_chunk (device memory of a launch is released there), alloc, upload,
// download, zero, each(m, f), per_block(nb, f), sort_pairs, scan_max, scan_sum, block_crcs, result_alloc / result_free.  Returns
// SWC_OK -- *out (from result_alloc, the caller's to free) holds the *out_len bytes of the stream -- or SWC_E_DEVICE (an
// allocation or a launch failed).
template <class X>
int compress_stream(X& x, const uint8_t* data, size_t len, int level, uint8_t** out, size_t* out_len) {
    const size_t raw_block = (size_t)level * 100u * 800u;                         // :46
    BitSink<X> sink(x);
    if (!sink.put(0x425a, 16) || !sink.put(0x68, 8) || !sink.put((uint64_t)(0x30 + level), 8)) return SWC_E_DEVICE;   // :48-50
    uint32_t total_crc = 0;
    const size_t n_blocks_all = (len + raw_block - 1) / raw_block;
    for (size_t b0 = 0; b0 < n_blocks_all; b0 += kBlocksPerLaunch) {
        const uint32_t nb = (uint32_t)(n_blocks_all - b0 < kBlocksPerLaunch ? n_blocks_all - b0 : kBlocksPerLaunch);
        const uint8_t* chunk = data + b0 * raw_block;
        const size_t chunk_len = (size_t)(b0 + nb == n_blocks_all ? len - b0 * raw_block : (size_t)nb * raw_block);
        x.begin_chunk();
        const uint32_t n_raw = (uint32_t)chunk_len;
        const size_t bound = chunk_len + chunk_len / 4 + 16 * (size_t)nb + 16;    // after rle1, at most
        const size_t t4 = 4 * bound + 16, t8 = 8 * bound + 16;
        uint8_t* d_raw = (uint8_t*)x.alloc(chunk_len + 16);
        uint32_t* d_offs = (uint32_t*)x.alloc(4 * (size_t)(nb + 1) * 4);            // off | blk_at | seg_off | the sort's word
        BlockInfo* d_infos = (BlockInfo*)x.alloc(sizeof(BlockInfo) * nb);
        uint8_t* d_text = (uint8_t*)x.alloc(bound + 16);
        uint8_t* d_blk = (uint8_t*)x.alloc(bound + 16);
        uint8_t* d_col = (uint8_t*)x.alloc(bound + 16);
        uint32_t* d_rank = (uint32_t*)x.alloc(t4);
        uint32_t* d_sa = (uint32_t*)x.alloc(t4);
        uint64_t* d_k0 = (uint64_t*)x.alloc(t8);
        uint64_t* d_k1 = (uint64_t*)x.alloc(t8);
        uint32_t* d_v0 = (uint32_t*)x.alloc(t4);
        uint32_t* d_v1 = (uint32_t*)x.alloc(t4);
        uint32_t* d_s0 = (uint32_t*)x.alloc(t4);
        uint32_t* d_s1 = (uint32_t*)x.alloc(t4);
        uint32_t* d_head = (uint32_t*)x.alloc(t4);
        uint32_t* d_keep = (uint32_t*)x.alloc(t4);
        uint32_t* d_kpos = (uint32_t*)x.alloc(t4);
        if (!d_raw || !d_offs || !d_infos || !d_text || !d_blk || !d_col || !d_rank || !d_sa || !d_k0 || !d_k1 || !d_v0 || !d_v1 ||
            !d_s0 || !d_s1 || !d_head || !d_keep || !d_kpos) return SWC_E_DEVICE;
        x.mark("alloc");
        x.upload(d_raw, chunk, chunk_len);
        x.mark("upload");
        x.zero(d_infos, sizeof(BlockInfo) * nb);
        std::vector<uint32_t> raw_off(nb + 1);
        for (uint32_t b = 0; b <= nb; b++) raw_off[b] = (uint32_t)(b * raw_block < chunk_len ? b * raw_block : chunk_len);
        std::vector<uint32_t> crcs(nb);
        if (x.block_crcs(d_raw, raw_off.data(), nb, crcs.data())) return SWC_E_DEVICE;
        x.mark("crc");
        // ---- rle1: the blocks one behind the other (the sort's buffers are free until then)
        {
            Rle1 r;
            memset(&r, 0, sizeof(r));
            r.raw = (const SWC_AS_GLOBAL uint8_t*)d_raw;
            r.n = n_raw; r.raw_block = (uint32_t)raw_block; r.nb = nb;
            r.start = (SWC_AS_GLOBAL uint32_t*)d_head;
            r.cnt = (SWC_AS_GLOBAL uint32_t*)d_keep;
            r.pos = (SWC_AS_GLOBAL uint32_t*)d_kpos;
            r.text = (SWC_AS_GLOBAL uint8_t*)d_text;
            r.blk = (SWC_AS_GLOBAL uint8_t*)d_blk;
            r.off = (SWC_AS_GLOBAL uint32_t*)d_offs;
            x.each(n_raw, Rle1Flags{r});
            if (x.scan_max(d_head, n_raw)) return SWC_E_DEVICE;
            x.each(n_raw, Rle1Counts{r});
            if (x.scan_sum(d_keep, d_kpos, n_raw)) return SWC_E_DEVICE;
            x.each(n_raw, Rle1Write{r});
        }
        std::vector<uint32_t> off(nb + 1), seg_off(nb + 1);
        x.download(off.data(), d_offs, 4 * (nb + 1));
        seg_off[0] = 0;
        uint32_t longest = 0;
        size_t stream_cap = 16;
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t n_rle = off[b + 1] - off[b];
            stream_cap += out_bound(n_rle);
            seg_off[b + 1] = seg_off[b] + (n_rle + kSeg - 1u) / kSeg;
            if (n_rle > longest) longest = n_rle;
        }
        const uint32_t total = off[nb], n_segs = seg_off[nb];
        if (total > bound) return SWC_E_DEVICE;
        std::vector<uint32_t> seg_blk(n_segs);
        for (uint32_t b = 0; b < nb; b++) for (uint32_t q = seg_off[b]; q < seg_off[b + 1]; q++) seg_blk[q] = b;
        x.upload(d_offs + 2 * (nb + 1), seg_off.data(), 4 * (nb + 1));
        x.mark("rle1");
        // ---- the sort
        {
            Bwt c;
            memset(&c, 0, sizeof(c));
            c.text = (const SWC_AS_GLOBAL uint8_t*)d_text;
            c.blk = (const SWC_AS_GLOBAL uint8_t*)d_blk;
            c.off = (const SWC_AS_GLOBAL uint32_t*)d_offs;
            c.infos = (SWC_AS_GLOBAL BlockInfo*)d_infos;
            c.rank = (SWC_AS_GLOBAL uint32_t*)d_rank;
            c.sa = (SWC_AS_GLOBAL uint32_t*)d_sa;
            c.key_in = (SWC_AS_GLOBAL uint64_t*)d_k0;
            c.key_out = (SWC_AS_GLOBAL uint64_t*)d_k1;
            c.val_in = (SWC_AS_GLOBAL uint32_t*)d_v0;
            c.val_out = (SWC_AS_GLOBAL uint32_t*)d_v1;
            c.head = (SWC_AS_GLOBAL uint32_t*)d_head;
            c.keep = (SWC_AS_GLOBAL uint32_t*)d_keep;
            c.kpos = (SWC_AS_GLOBAL uint32_t*)d_kpos;
            c.next_m = (SWC_AS_GLOBAL uint32_t*)(d_offs + 3 * (nb + 1));
            c.col = (SWC_AS_GLOBAL uint8_t*)d_col;
            c.slot = nullptr;
            c.slot_next = (SWC_AS_GLOBAL uint32_t*)d_s0;
            uint32_t* slot_bufs[2] = {d_s0, d_s1};
            int which = 0;
            uint32_t rank_bits = 1;
            while ((1ull << rank_bits) < (uint64_t)total) rank_bits++;
            uint32_t blk_bits = 1;
            while ((1u << blk_bits) < nb) blk_bits++;
            c.rank_bits = rank_bits;
            c.m = total; c.h = 0;
            x.each(total, FirstKeys{c});
            int key_bits = (int)(8u * kFirstBytes + blk_bits);
            while (c.m != 0u) {
                x.note("sort round: elements", c.m, (uint32_t)key_bits);
                if (x.sort_pairs(d_k0, d_k1, d_v0, d_v1, c.m, key_bits)) return SWC_E_DEVICE;
                x.each(c.m, Heads{c});
                if (x.scan_max(d_head, c.m)) return SWC_E_DEVICE;
                x.each(c.m, Ranks{c});
                if (x.scan_sum(d_keep, d_kpos, c.m)) return SWC_E_DEVICE;
                x.each(c.m, Compact{c});
                uint32_t next_m = 0;
                x.download(&next_m, d_offs + 3 * (nb + 1), 4);
                c.m = next_m;
                c.slot = c.slot_next;
                which ^= 1;
                c.slot_next = (SWC_AS_GLOBAL uint32_t*)slot_bufs[which];
                c.h = c.h ? 2u * c.h : kFirstBytes;
                if (c.h >= longest) break;                    // what is still equal is equal all the way round: any order
                if (c.m) x.each(c.m, NextKeys{c});
                key_bits = (int)(2u * rank_bits);
            }
            x.each(total, LastColumn{c});
        }
        x.mark("sort");
        // ---- symbols, tables, bits
        uint16_t* d_syms = (uint16_t*)x.alloc(2 * (size_t)kSegSyms * n_segs + 16);
        uint32_t* d_segs = (uint32_t*)x.alloc(4 * (size_t)n_segs * 5 + 16);          // seg_blk | seg_nsym | seg_bits | seg_at | seg_sym_at
        uint32_t* d_last = (uint32_t*)x.alloc(4 * 256 * (size_t)n_segs + 16);
        uint32_t* d_before = (uint32_t*)x.alloc(4 * 256 * (size_t)n_segs + 16);
        uint8_t* d_used = (uint8_t*)x.alloc(256 * (size_t)nb);
        uint8_t* d_stream = (uint8_t*)x.alloc(stream_cap);
        const size_t tab_cells = (size_t)nb * kMaxTables * kSymStride;
        uint32_t* d_rfreq = (uint32_t*)x.alloc(4 * tab_cells);
        uint32_t* d_codes = (uint32_t*)x.alloc(4 * tab_cells);
        uint8_t* d_len8 = (uint8_t*)x.alloc(tab_cells);
        if (!d_syms || !d_segs || !d_last || !d_before || !d_used || !d_stream || !d_rfreq || !d_codes || !d_len8) return SWC_E_DEVICE;
        x.upload(d_segs, seg_blk.data(), 4 * (size_t)n_segs);
        Segs g;
        {
            memset(&g, 0, sizeof(g));
            g.col = (const SWC_AS_GLOBAL uint8_t*)d_col;
            g.off = (const SWC_AS_GLOBAL uint32_t*)d_offs;
            g.seg_off = (const SWC_AS_GLOBAL uint32_t*)(d_offs + 2 * (nb + 1));
            g.seg_blk = (const SWC_AS_GLOBAL uint32_t*)d_segs;
            g.last = (SWC_AS_GLOBAL uint32_t*)d_last;
            g.before = (SWC_AS_GLOBAL uint32_t*)d_before;
            g.used = (SWC_AS_GLOBAL uint8_t*)d_used;
            g.syms = (SWC_AS_GLOBAL uint16_t*)d_syms;
            g.seg_nsym = (SWC_AS_GLOBAL uint32_t*)(d_segs + n_segs);
            g.seg_bits = (SWC_AS_GLOBAL uint32_t*)(d_segs + 2 * (size_t)n_segs);
            g.seg_at = (SWC_AS_GLOBAL uint32_t*)(d_segs + 3 * (size_t)n_segs);
            g.seg_sym_at = (SWC_AS_GLOBAL uint32_t*)(d_segs + 4 * (size_t)n_segs);
            g.rfreq = (SWC_AS_GLOBAL uint32_t*)d_rfreq;
            g.codes = (SWC_AS_GLOBAL uint32_t*)d_codes;
            g.len8 = (SWC_AS_GLOBAL uint8_t*)d_len8;
            g.stream = (SWC_AS_GLOBAL uint8_t*)d_stream;
            g.blk_at = (SWC_AS_GLOBAL uint32_t*)(d_offs + (nb + 1));
            g.infos = (SWC_AS_GLOBAL BlockInfo*)d_infos;
            g.nb = nb;
            g.stream_cap = (uint32_t)(stream_cap & ~(size_t)3);
            g.lead = (uint32_t)(sink.bits & 31u);
            x.per_block(n_segs, SegLast{g});
            x.each(256u * nb, SegBefore{g});
            x.per_block(n_segs, MtfSeg{g});
            x.per_block(nb, SymScan{g});
        }
        std::vector<BlockInfo> infos(nb);
        x.download(infos.data(), d_infos, sizeof(BlockInfo) * nb);
        x.mark("mtf");
        // the blocks' symbols one behind the other, the groups of 50, the tables the refinement starts from
        std::vector<uint32_t> sym_off(nb + 1), grp_off(nb + 1);
        std::vector<uint8_t> len8(tab_cells, 0);
        sym_off[0] = 0; grp_off[0] = 0;
        uint32_t most_groups = 0;
        for (uint32_t b = 0; b < nb; b++) {
            BlockInfo& in = infos[b];
            in.crc = crcs[b];
            in.n_tables = tables_for(in.n_sym);
            in.n_groups = (in.n_sym + kGroup - 1u) / kGroup;
            sym_off[b + 1] = sym_off[b] + ((in.n_sym + 1u) & ~1u);                 // (even: a group's symbols can be read as dwords)
            grp_off[b + 1] = grp_off[b] + in.n_groups;
            if (in.n_groups > most_groups) most_groups = in.n_groups;
            initial_tables(in.freq, in.n_used + 2u, in.n_sym, in.n_tables, len8.data() + (size_t)b * kMaxTables * kSymStride);
        }
        uint16_t* d_csyms = (uint16_t*)x.alloc(2 * (size_t)sym_off[nb] + 16);
        uint8_t* d_sel = (uint8_t*)x.alloc(2 * (size_t)grp_off[nb] + 16);             // sel | selmtf
        uint32_t* d_offs2 = (uint32_t*)x.alloc(4 * (size_t)(nb + 1) * 2);             // sym_off | grp_off
        if (!d_csyms || !d_sel || !d_offs2) return SWC_E_DEVICE;
        x.upload(d_infos, infos.data(), sizeof(BlockInfo) * nb);
        x.upload(d_offs2, sym_off.data(), 4 * (nb + 1));
        x.upload(d_offs2 + (nb + 1), grp_off.data(), 4 * (nb + 1));
        x.upload(d_len8, len8.data(), tab_cells);
        g.csyms = (SWC_AS_GLOBAL uint16_t*)d_csyms;
        g.sym_off = (const SWC_AS_GLOBAL uint32_t*)d_offs2;
        g.grp_off = (const SWC_AS_GLOBAL uint32_t*)(d_offs2 + (nb + 1));
        g.sel = (SWC_AS_GLOBAL uint8_t*)d_sel;
        g.selmtf = (SWC_AS_GLOBAL uint8_t*)(d_sel + grp_off[nb]);
        g.cost_waves = (most_groups + kGroupsPerWave - 1u) / kGroupsPerWave;
        x.per_block(n_segs, SymCompact{g});
        for (uint32_t it = 0; it < kTableIters; it++) {
            x.zero(d_rfreq, 4 * tab_cells);
            x.per_block(nb * g.cost_waves, GroupCost{g});
            x.per_block(nb * kMaxTables, Huff{g});
        }
        x.mark("tables");
        x.zero(d_stream, stream_cap);
        x.per_block(nb, BlockHead{g});
        x.per_block(n_segs, SegBits{g});
        x.per_block(nb, SegScan{g});
        x.per_block(1, BlockScan{g});
        x.per_block(n_segs, EmitSeg{g});
        // the blocks' bits go behind what the stream has: the device has put them from the bit the stream's last dword has
        // reached, that dword's bytes are kept across the download
        const uint64_t base = sink.bits & ~31ull;
        std::vector<uint32_t> at(nb + 1);
        x.download(at.data(), d_offs + (nb + 1), 4 * (nb + 1));
        x.mark("emit");
        const size_t got = (size_t)((at[nb] + 7u) >> 3), at_byte = (size_t)(base >> 3);
        const size_t launches_left = (n_blocks_all - b0 + kBlocksPerLaunch - 1) / kBlocksPerLaunch;
        if (!sink.ensure(at_byte + (launches_left > 1 ? got + got / 8 : got) * launches_left + 64)) return SWC_E_DEVICE;
        uint8_t keep[4] = {0, 0, 0, 0};
        const size_t kept = (size_t)((at[0] + 7u) >> 3);
        memcpy(keep, sink.p + at_byte, kept);
        x.download(sink.p + at_byte, d_stream, got);
        for (size_t k = 0; k < kept; k++) sink.p[at_byte + k] |= keep[k];
        sink.bits = base + at[nb];
        x.mark("download");
        for (uint32_t b = 0; b < nb; b++) {
            total_crc = (total_crc << 1) | (total_crc >> 31);                     // :56-57
            total_crc ^= crcs[b];
        }
        x.end_chunk();
    }
    if (!sink.put(kEosMagic, 48) || !sink.put(total_crc, 32)) return SWC_E_DEVICE;    // :67-71
    *out = sink.release(out_len);
    return SWC_OK;
}

}  // namespace bz2c
}  // namespace swc
#endif
