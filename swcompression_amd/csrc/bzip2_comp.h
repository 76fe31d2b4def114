// bzip2_comp.h -- BZip2 COMPRESSION on the device (the third piece of the encode side).
//
// Replaces BZip2.compress(data:blockSize:) (reference Sources/BZip2/BZip2+Compress.swift:40-74), process(_:_:) (:76-241) and
// their helpers -- initialRle :243-262, BurrowsWheeler.transform (BurrowsWheeler.swift:8-29, SuffixArray.swift), mtfRle
// :277-325 -- for ALL blocks of a stream at once (blocks are independent until their bits are joined):
//
//   rle1      one block per WAVEFRONT.  Runs of 4..255 equal bytes become four bytes and a count (:243-262): 64 bytes per step; a
//             max-scan of the run starts gives every lane its position in its run, a prefix sum of what the lanes emit (the
//             byte itself when its position in the 255-byte sub-run is below four, a count behind the last byte of a sub-run of
//             four and more) gives the places;
//   sort      the Burrows-Wheeler transform of the result: the rotations of ALL blocks sorted together by prefix doubling.
//             The first key of a rotation is its block and its first seven bytes (one 64-bit radix sort settles most of a
//             text), then rounds over the rotations that still share a key with a neighbour ONLY: key = (rank of the
//             rotation, rank of the rotation h further on), h = 7, 14, 28 ...; a rotation that is alone in its group has
//             its final row and leaves the working set.  One thread per element, elementwise kernels around a device radix
//             sort and two device scans (rocPRIM: the only library calls of the engine; bzip2_compress.hip);
//   mtf       one block per wavefront.  Move-to-front over the bytes that occur, zero runs as RUNA / RUNB digits (bijective base
//             2), the end-of-block symbol (:277-325).  A byte equal to its predecessor IS a zero, so a ballot per 64 bytes
//             finds the positions that change the list and only those are walked; the list's first 64 positions live one per
//             lane in a register (finding a byte is a compare and a ballot, moving it to the front one DPP shift), positions
//             64..255 in three more registers that are touched only when the byte is found that deep;
//   lengths   the HOST turns a block's frequencies into code lengths and canonical codes (huffman_lengths: a plain Huffman
//             tree, weights flattened until no code is longer than 17 bits, as bzip2 itself does) -- 258 numbers per block;
//   emit      one block per wavefront: magic, CRC, origin pointer, the map of used bytes, TWO identical tables (the format wants
//             two; the reference duplicates its only one the same way, :142-147) with all selectors zero, the code lengths in
//             delta form, then the symbols -- 64 per step, a wave scan of the code lengths for the places, ds_or_b32 into a
//             staging area of the MSB-first bit stream in LDS, byte-swapped dwords to HBM;
//   join      one block per wavefront: every block's bits are shifted to their place in the stream (blocks start at arbitrary
//             bit offsets), atomic OR into the zeroed result.
//
// The contract is that of the other two encoders: A valid bzip2 stream for the same bytes with the reference's block
// cutting (level x 80,000 raw bytes per block, :46), not the reference's bytes (the reference builds up to six tables with a
// greedy per-50-symbols rule, :95-139; one table built from the block's own frequencies is what this version has).
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
constexpr uint32_t kFirstBytes = 7;

// what a block's stages hand to each other (HBM, one per block)
struct BlockInfo {
    uint32_t n_raw;        // bytes of the block in the input
    uint32_t n_rle;        // ... after rle1 (the length of the sorted column)
    uint32_t orig_ptr;     // row of the unrotated block in the sorted matrix
    uint32_t n_sym;        // symbols after mtf (the end-of-block symbol included)
    uint32_t n_used;       // distinct bytes of the column
    uint32_t crc;          // bzip2 CRC-32 of the block's raw bytes
    uint32_t out_bits;     // bits of the block in its output area
    uint32_t pad;
    uint32_t used[8];      // bit b of word w: byte 32 w + b occurs
    uint32_t freq[kMaxSyms + 2];
    uint32_t code[kMaxSyms + 2];   // from the host: canonical code | length << 24
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
// BZip2+Compress.swift:243-262.  Returns the bytes written.
template <int N>
SWC_D uint32_t rle1_job(gcptr src, uint32_t n, gptr dst) {
    using simt::PT;
    uint32_t opos = 0, carry_q = 0, carry_b = 0x100;   // uncapped run position and value of the byte in front of the chunk
    for (uint32_t base = 0; base < n; base += (uint32_t)N) {
        PT<uint32_t, N> b, nb, prev, st, cnt, q, x;
        SIMT_BEGIN(t, N)
            const uint32_t i = base + (uint32_t)t;
            b[t] = i < n ? src[i] : 0x200u;
            nb[t] = i + 1u < n ? src[i + 1u] : 0x300u;
        SIMT_END
        simt::wave_shift_up<N>(prev, b, carry_b);
        SIMT_BEGIN(t, N) st[t] = b[t] != prev[t] ? (uint32_t)t + 1u : 0u; SIMT_END
        simt::wave_scan_max_incl<N>(st);
        SIMT_BEGIN(t, N)
            const uint32_t i = base + (uint32_t)t;
            q[t] = st[t] ? (uint32_t)t - (st[t] - 1u) : carry_q + 1u + (uint32_t)t;   // my position in my run
            const uint32_t p = q[t] % 255u;                                            // ... in its 255-byte sub-run
            const bool ends = nb[t] != b[t] || p == 254u;
            cnt[t] = i < n ? (p < 4u ? 1u : 0u) + (ends && p >= 3u ? 1u : 0u) : 0u;
            x[t] = cnt[t];
        SIMT_END
        simt::wave_scan_incl<N>(x);
        const uint32_t o0 = opos;
        SIMT_BEGIN(t, N)
            const uint32_t p = q[t] % 255u;
            uint32_t o = o0 + x[t] - cnt[t];
            if (cnt[t] != 0u && p < 4u) dst[o++] = (uint8_t)b[t];
            if (cnt[t] != 0u && (nb[t] != b[t] || p == 254u) && p >= 3u) dst[o] = (uint8_t)(p - 3u);
        SIMT_END
        opos += simt::wave_read<N>(x, N - 1);
        const uint32_t last = n - base < (uint32_t)N ? n - base - 1u : (uint32_t)N - 1u;
        carry_q = simt::wave_read<N>(q, (int)last);
        carry_b = simt::wave_read<N>(b, (int)last);
    }
    return opos;
}

// ================================================================================================================ mtf
struct MtfLds {
    uint32_t freq[kMaxSyms + 6];
    uint32_t list[256];
    uint8_t present[256];
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
// L (n >= 1 bytes) -> symbols; fills info.used / n_used / freq / n_sym
template <int N>
SWC_D void mtf_job(gcptr L, uint32_t n, SWC_AS_GLOBAL uint16_t* syms, MtfLds* lds, SWC_AS_GLOBAL BlockInfo* info) {
    using simt::PT;
    static_assert(N == 64, "one list position per lane, four registers");
    SIMT_BEGIN(t, N)
        for (uint32_t i = (uint32_t)t; i < kMaxSyms + 6u; i += (uint32_t)N) lds->freq[i] = 0u;
        for (uint32_t i = (uint32_t)t; i < 256u; i += (uint32_t)N) { lds->present[i] = 0; lds->list[i] = 0x1FFu; }
    SIMT_END_WAVE
    SIMT_BEGIN(t, N)
        for (uint32_t i = (uint32_t)t; i < n; i += (uint32_t)N) lds->present[L[i]] = 1;
    SIMT_END_WAVE
    // the bytes that occur, in order, are the list (:279-283)
    uint32_t used[8];
    for (uint32_t r = 0; r < 4u; r++) {
        PT<bool, N> pr;
        SIMT_BEGIN(t, N) pr[t] = lds->present[64u * r + (uint32_t)t] != 0; SIMT_END
        const uint64_t m = simt::wave_ballot<N>(pr);
        used[2u * r] = (uint32_t)m;
        used[2u * r + 1u] = (uint32_t)(m >> 32);
    }
    uint32_t n_used = 0, below[8];
    for (uint32_t w = 0; w < 8u; w++) { below[w] = n_used; n_used += (uint32_t)simt::popc32(used[w]); }
    SIMT_BEGIN(t, N)
        for (uint32_t r = 0; r < 4u; r++) {
            const uint32_t v = 64u * r + (uint32_t)t, w = v >> 5, bit = v & 31u;
            uint32_t uw = 0, bw = 0;     // (a select chain, not used[w]: the words live in scalar registers)
            for (uint32_t k = 0; k < 8u; k++) { uw = w == k ? used[k] : uw; bw = w == k ? below[k] : bw; }
            if ((uw >> bit) & 1u) lds->list[bw + (uint32_t)simt::popc32(uw & ((1u << bit) - 1u))] = v;
        }
        if (t < 8) {
            uint32_t uw = 0;
            for (uint32_t k = 0; k < 8u; k++) uw = (uint32_t)t == k ? used[k] : uw;
            info->used[t] = uw;
        }
    SIMT_END_WAVE
    PT<uint32_t, N> l0, l1, l2, l3;
    SIMT_BEGIN(t, N)
        l0[t] = lds->list[t]; l1[t] = lds->list[64 + t]; l2[t] = lds->list[128 + t]; l3[t] = lds->list[192 + t];
    SIMT_END
    MtfOut<N> m;
    m.l = lds; m.out = syms; m.nsym = 0; m.nstage = 0;
    SIMT_BEGIN(t, N) m.stage[t] = 0u; SIMT_END
    uint32_t run = 0;
    uint32_t carry_b = simt::wave_read<N>(l0, 0);             // the front of the list: a byte equal to it is a zero (:286-288)
    for (uint32_t base = 0; base < n; base += (uint32_t)N) {
        PT<uint32_t, N> chunk, prev;
        PT<bool, N> ch;
        SIMT_BEGIN(t, N) chunk[t] = base + (uint32_t)t < n ? L[base + (uint32_t)t] : 0x200u; SIMT_END
        simt::wave_shift_up<N>(prev, chunk, carry_b);
        const uint32_t k1 = n - base < (uint32_t)N ? n - base : (uint32_t)N;
        SIMT_BEGIN(t, N) ch[t] = (uint32_t)t < k1 && chunk[t] != prev[t]; SIMT_END
        uint64_t cm = simt::wave_ballot<N>(ch);
        uint32_t pos = 0;
        while (cm != 0ull) {
            const uint32_t k = (uint32_t)simt::ctz64(cm);
            cm &= cm - 1ull;
            run += k - pos;
            pos = k + 1u;
            const uint32_t b = simt::wave_read<N>(chunk, (int)k);
            m.put_run(run);
            run = 0;
            // where is it?  (never at the front: that is the byte before it)
            PT<bool, N> e;
            SIMT_BEGIN(t, N) e[t] = l0[t] == b; SIMT_END
            uint64_t bal = simt::wave_ballot<N>(e);
            if (bal != 0ull) {
                const uint32_t li = (uint32_t)simt::ctz64(bal);
                m.put(li + 1u);                                               // :309-315
                PT<uint32_t, N> s0;
                simt::wave_shift_up_dpp<N>(s0, l0, b);
                SIMT_BEGIN(t, N) l0[t] = (uint32_t)t <= li ? s0[t] : l0[t]; SIMT_END
                continue;
            }
            uint32_t idx;
            SIMT_BEGIN(t, N) e[t] = l1[t] == b; SIMT_END
            bal = simt::wave_ballot<N>(e);
            if (bal != 0ull) idx = 64u + (uint32_t)simt::ctz64(bal);
            else {
                SIMT_BEGIN(t, N) e[t] = l2[t] == b; SIMT_END
                bal = simt::wave_ballot<N>(e);
                if (bal != 0ull) idx = 128u + (uint32_t)simt::ctz64(bal);
                else {
                    SIMT_BEGIN(t, N) e[t] = l3[t] == b; SIMT_END
                    bal = simt::wave_ballot<N>(e);
                    idx = 192u + (uint32_t)simt::ctz64(bal | (1ull << 63));
                }
            }
            m.put(idx + 1u);
            // move to the front: the positions below idx move up by one
            const uint32_t r = idx >> 6, li = idx & 63u;
            const uint32_t e0 = simt::wave_read<N>(l0, N - 1), e1 = simt::wave_read<N>(l1, N - 1), e2 = simt::wave_read<N>(l2, N - 1);
            PT<uint32_t, N> s0, s1, s2, s3;
            simt::wave_shift_up_dpp<N>(s0, l0, b);
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
    m.put(n_used + 1u);                                                   // the end-of-block symbol (:322-323)
    if (m.nstage) m.flush();
    const uint32_t nsym = m.nsym;
    SIMT_BEGIN(t, N)
        for (uint32_t i = (uint32_t)t; i < kMaxSyms + 2u; i += (uint32_t)N) info->freq[i] = lds->freq[i];
        if (t == 0) { info->n_sym = nsym; info->n_used = n_used; }
    SIMT_END_WAVE
}

// ================================================================================================================ emit
struct EmitLds {
    uint32_t stage[kStageDw + 4];
    uint32_t code[kMaxSyms + 6];
};
template <int N>
struct Emitter {
    EmitLds* l;
    gptr out;          // 16-byte aligned, `cap` a multiple of four
    uint32_t cap;
    uint32_t obits, odw, fill;

    SWC_D void flush(bool all) {
        const uint32_t nd = all ? (fill + 31u) >> 5 : fill >> 5;
        const uint32_t o0 = odw;
        simt::PT<uint32_t, N> carry;
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < nd; i += (uint32_t)N) {
                const uint32_t b = 4u * (o0 + i);
                if (b + 4u <= cap) *(SWC_AS_GLOBAL uint32_t*)(out + b) = bswap32(l->stage[i]);   // the stream's first bit is the top bit of byte 0
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
    // `n` zero bits
    SWC_D void skip(uint32_t n) {
        while (n != 0u) {
            const uint32_t k = n < 2048u ? n : 2048u;
            fill += k;
            obits += k;
            n -= k;
            if (fill >= 32u * kFlushDw) flush(false);
        }
    }
};
// The block from its magic to its last symbol.  info.code[] = canonical code | length << 24 of every symbol 0 .. n_used + 1.
template <int N>
SWC_D void emit_job(const SWC_AS_GLOBAL uint16_t* syms, EmitLds* lds, SWC_AS_GLOBAL BlockInfo* info, gptr out, uint32_t cap) {
    using simt::PT;
    Emitter<N> e;
    e.l = lds; e.out = out; e.cap = cap; e.obits = 0; e.odw = 0; e.fill = 0;
    const uint32_t alpha = info->n_used + 2u;
    SIMT_BEGIN(t, N)
        for (uint32_t i = (uint32_t)t; i < kStageDw + 4u; i += (uint32_t)N) lds->stage[i] = 0u;
        for (uint32_t i = (uint32_t)t; i < kMaxSyms + 6u; i += (uint32_t)N) lds->code[i] = i < alpha ? info->code[i] : 0u;
    SIMT_END_WAVE
    PT<uint32_t, N> code, nb;
    // magic (48), block CRC (32), randomised = 0 (1), origin pointer (24), the 16 bits of the used ranges (:149-166)
    uint32_t ranges = 0;
    for (uint32_t r = 0; r < 16u; r++) if ((info->used[r >> 1] >> (16u * (r & 1u))) & 0xFFFFu) ranges |= 1u << (15u - r);
    const uint32_t crc = info->crc, optr = info->orig_ptr;
    SIMT_BEGIN(t, N)
        uint32_t c = 0, n = 0;
        switch (t) {
            case 0: c = (uint32_t)(kBlockMagic >> 24); n = 24; break;
            case 1: c = (uint32_t)(kBlockMagic & 0xFFFFFFu); n = 24; break;
            case 2: c = crc; n = 32; break;
            case 3: c = 0; n = 1; break;
            case 4: c = optr; n = 24; break;
            case 5: c = ranges; n = 16; break;
            default: break;
        }
        code[t] = c; nb[t] = n;
    SIMT_END
    e.emit(code, nb);
    // the 16 bits of every used range (:168-178)
    SIMT_BEGIN(t, N)
        uint32_t c = 0, n = 0;
        if ((uint32_t)t < 16u && ((ranges >> (15u - (uint32_t)t)) & 1u)) {
            const uint32_t w = (info->used[t >> 1] >> (16u * ((uint32_t)t & 1u))) & 0xFFFFu;   // bit b: byte 16 t + b
            c = brev32(w) >> 16;                                                                // byte 16 t first
            n = 16;
        }
        code[t] = c; nb[t] = n;
    SIMT_END
    e.emit(code, nb);
    // two tables, the selectors (:180-194): one per 50 symbols, all of them table 0 = a zero bit each after move-to-front
    const uint32_t nsym = info->n_sym;
    const uint32_t nsel = (nsym + 49u) / 50u;
    SIMT_BEGIN(t, N)
        code[t] = t == 0 ? 2u : t == 1 ? nsel : 0u;
        nb[t] = t == 0 ? 3u : t == 1 ? 15u : 0u;
    SIMT_END
    e.emit(code, nb);
    e.skip(nsel);
    // the code lengths of both tables in delta form (:196-221): 5 bits of the first length, then per symbol "10" / "11" steps and a 0
    for (uint32_t tab = 0; tab < 2u; tab++) {
        const uint32_t first = lds->code[0] >> 24;
        SIMT_BEGIN(t, N) code[t] = t == 0 ? first : 0u; nb[t] = t == 0 ? 5u : 0u; SIMT_END
        e.emit(code, nb);
        for (uint32_t s0 = 0; s0 < alpha; s0 += 32u) {     // a symbol takes at most 2 * 19 + 1 = 39 bits: two lanes per symbol
            SIMT_BEGIN(t, N)
                const uint32_t s = s0 + ((uint32_t)t >> 1);
                uint32_t c = 0, n = 0;
                if (s < alpha) {
                    const uint32_t len = lds->code[s] >> 24, prev = s == 0u ? len : lds->code[s - 1u] >> 24;
                    const uint32_t d = len > prev ? len - prev : prev - len;       // <= 19
                    const uint32_t pat = len > prev ? 0xAAAAAAAAu : 0xFFFFFFFFu;   // "10" steps up, "11" steps down
                    if (((uint32_t)t & 1u) == 0u) {        // the first min(d, 16) steps
                        const uint32_t k = d < 16u ? d : 16u;
                        n = 2u * k;
                        c = k ? pat >> (32u - n) : 0u;
                    } else {                               // the remaining steps and the closing 0
                        const uint32_t k = d > 16u ? d - 16u : 0u;
                        n = 2u * k + 1u;
                        c = k ? (pat >> (32u - 2u * k)) << 1 : 0u;
                    }
                }
                code[t] = c; nb[t] = n;
            SIMT_END
            e.emit(code, nb);
        }
    }
    // the symbols (:223-240)
    for (uint32_t base = 0; base < nsym; base += (uint32_t)N) {
        SIMT_BEGIN(t, N)
            uint32_t c = 0, n = 0;
            if (base + (uint32_t)t < nsym) {
                const uint32_t w = lds->code[syms[base + (uint32_t)t]];
                c = w & 0xFFFFFFu; n = w >> 24;
            }
            code[t] = c; nb[t] = n;
        SIMT_END
        e.emit(code, nb);
    }
    const uint32_t bits = e.obits;
    e.flush(true);
    SIMT_BEGIN(t, N) if (t == 0) info->out_bits = bits; SIMT_END_WAVE
}

// ================================================================================================================ join
// `bits` bits at src (big-endian dwords, 4-byte aligned) to bit offset `at` of dst (zeroed, 4-byte aligned)
template <int N>
SWC_D void join_job(gcptr src, uint32_t bits, gptr dst, uint64_t at) {
    const uint32_t nd = (bits + 31u) >> 5, sh = (uint32_t)(at & 31u);
    SWC_AS_GLOBAL uint32_t* d32 = (SWC_AS_GLOBAL uint32_t*)dst + (at >> 5);
    const SWC_AS_GLOBAL uint32_t* s32 = (const SWC_AS_GLOBAL uint32_t*)src;
    SIMT_BEGIN(t, N)
        for (uint32_t i = (uint32_t)t; i < nd; i += (uint32_t)N) {
            uint32_t v = bswap32(s32[i]);
            if (i == nd - 1u && (bits & 31u)) v &= ~(0xFFFFFFFFu >> (bits & 31u));      // nothing behind the last bit
            if (sh == 0u) global_or(d32 + i, bswap32(v));
            else {
                global_or(d32 + i, bswap32(v >> sh));
                const uint32_t lo = v << (32u - sh);
                if (lo) global_or(d32 + i + 1u, bswap32(lo));
            }
        }
    SIMT_END
}

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

// ================================================================================================================ per block
struct BlockPtrs {
    const SWC_AS_GLOBAL uint8_t* raw;           // the input of the launch
    const SWC_AS_GLOBAL uint32_t* raw_off;      // n_blocks + 1 offsets
    SWC_AS_GLOBAL uint8_t* area;                // rle1 results at rle1_bound() strides
    const SWC_AS_GLOBAL uint32_t* area_off;
    SWC_AS_GLOBAL uint8_t* text;
    SWC_AS_GLOBAL uint8_t* blk;
    const SWC_AS_GLOBAL uint32_t* off;
    const SWC_AS_GLOBAL uint8_t* col;
    SWC_AS_GLOBAL uint16_t* syms;               // block b at off[b] + 2 b
    SWC_AS_GLOBAL uint8_t* outs;                // block bit streams at out_bound() strides
    const SWC_AS_GLOBAL uint32_t* out_off;
    SWC_AS_GLOBAL uint8_t* stream;              // the joined result
    const SWC_AS_GLOBAL uint64_t* stream_at;    // bit offset of every block in it
    SWC_AS_GLOBAL BlockInfo* infos;
};
struct NoLds { uint32_t unused; };
struct Rle1Block {
    BlockPtrs p;
    typedef NoLds Lds;
    template <int N> SWC_D void run(uint32_t b, Lds*) const {
        const uint32_t n = p.raw_off[b + 1u] - p.raw_off[b];
        const uint32_t r = rle1_job<N>(p.raw + p.raw_off[b], n, p.area + p.area_off[b]);
        SIMT_BEGIN(t, N) if (t == 0) { p.infos[b].n_raw = n; p.infos[b].n_rle = r; } SIMT_END_WAVE
    }
};
struct GatherBlock {   // the rle1 results one behind the other, and the block of every position
    BlockPtrs p;
    typedef NoLds Lds;
    template <int N> SWC_D void run(uint32_t b, Lds*) const {
        const uint32_t n = p.off[b + 1u] - p.off[b];
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < n; i += (uint32_t)N) {
                p.text[p.off[b] + i] = p.area[p.area_off[b] + i];
                p.blk[p.off[b] + i] = (uint8_t)b;
            }
        SIMT_END
    }
};
struct MtfBlock {
    BlockPtrs p;
    typedef MtfLds Lds;
    template <int N> SWC_D void run(uint32_t b, Lds* lds) const {
        mtf_job<N>(p.col + p.off[b], p.off[b + 1u] - p.off[b], p.syms + p.off[b] + 2u * b, lds, p.infos + b);
    }
};
struct EmitBlock {
    BlockPtrs p;
    typedef EmitLds Lds;
    template <int N> SWC_D void run(uint32_t b, Lds* lds) const {
        emit_job<N>(p.syms + p.off[b] + 2u * b, lds, p.infos + b, p.outs + p.out_off[b], p.out_off[b + 1u] - p.out_off[b]);
    }
};
struct JoinBlock {
    BlockPtrs p;
    typedef NoLds Lds;
    template <int N> SWC_D void run(uint32_t b, Lds*) const {
        join_job<N>(p.outs + p.out_off[b], p.infos[b].out_bits, p.stream, p.stream_at[b]);
    }
};

}  // namespace bz2c
}  // namespace swc

// ================================================================================================================ host
#include <string.h>
#include <vector>
namespace swc {
namespace bz2c {

// freq[0 .. alpha): lengths of a Huffman code with no code longer than `max_len` (weights flattened until that holds, as
// bzip2's own hbMakeCodeLengths does), every symbol of the alphabet coded (a zero frequency counts as one).  code[s] =
// canonical code | length << 24: codes are handed out in order of (length, symbol), which is what decoders rebuild.
inline void huffman_lengths(const uint32_t* freq, uint32_t alpha, uint32_t max_len, uint32_t* code) {
    uint64_t w[kMaxSyms + 2];
    uint32_t len[kMaxSyms + 2];
    for (uint32_t i = 0; i < alpha; i++) w[i] = freq[i] ? freq[i] : 1u;
    for (;;) {
        // repeatedly join the two lightest nodes (alpha <= 258: quadratic is nothing next to a 900 kB block)
        uint64_t wt[2 * (kMaxSyms + 2)];
        int parent[2 * (kMaxSyms + 2)];
        bool live[2 * (kMaxSyms + 2)];
        uint32_t nn = alpha;
        for (uint32_t i = 0; i < alpha; i++) { wt[i] = w[i]; parent[i] = -1; live[i] = true; }
        for (uint32_t joined = 0; joined + 1 < alpha; joined++) {
            int a = -1, b = -1;
            for (uint32_t i = 0; i < nn; i++) {
                if (!live[i]) continue;
                if (a < 0 || wt[i] < wt[a]) { b = a; a = (int)i; }
                else if (b < 0 || wt[i] < wt[b]) b = (int)i;
            }
            wt[nn] = wt[a] + wt[b]; parent[nn] = -1; live[nn] = true;
            parent[a] = parent[b] = (int)nn; live[a] = live[b] = false;
            nn++;
        }
        uint32_t longest = 0;
        for (uint32_t i = 0; i < alpha; i++) {
            uint32_t d = 0;
            for (int q = parent[i]; q >= 0; q = parent[q]) d++;
            len[i] = d ? d : 1u;
            if (len[i] > longest) longest = len[i];
        }
        if (longest <= max_len) break;
        for (uint32_t i = 0; i < alpha; i++) w[i] = w[i] / 2 + 1;
    }
    uint32_t next = 0;
    for (uint32_t l = 1; l <= max_len; l++) {
        for (uint32_t s = 0; s < alpha; s++) if (len[s] == l) code[s] = next++ | (l << 24);
        next <<= 1;
    }
}

// MSB-first bits appended to a byte vector
struct BitSink {
    std::vector<uint8_t> bytes;
    uint64_t bits = 0;
    void put(uint64_t v, int n) {
        for (int i = n - 1; i >= 0; i--) {
            if ((bits & 7u) == 0) bytes.push_back(0);
            if ((v >> i) & 1u) bytes.back() |= (uint8_t)(0x80u >> (bits & 7u));
            bits++;
        }
    }
    // `nbits` bits of `src` whose first bit sits at bit (bits % 8) of src[0] -- the form the join kernel leaves them in
    void append_aligned(const uint8_t* src, uint64_t nbits) {
        if (nbits == 0) return;
        const uint32_t lead = (uint32_t)(bits & 7u);
        const uint64_t span = (lead + nbits + 7u) >> 3;
        const size_t at = bytes.size();
        if (lead) { bytes.back() |= (uint8_t)(src[0] & (0xFFu >> lead)); src++; }
        const uint64_t rest = span - (lead ? 1 : 0);
        bytes.resize(at + rest);
        if (rest) memcpy(bytes.data() + at, src, rest);
        bits += nbits;
    }
};

// The whole stream.  X is the executor: begin_chunk / end_chunk (device memory of a launch is released there), alloc, upload,
// download, zero, each(m, f), per_block(nb, f), sort_pairs, scan_max, scan_sum, block_crcs.  Returns SWC_OK or SWC_E_DEVICE
// (an allocation or a launch failed).
template <class X>
int compress_stream(X& x, const uint8_t* data, size_t len, int level, std::vector<uint8_t>& result) {
    const size_t raw_block = (size_t)level * 100u * 800u;                         // :46
    BitSink sink;
    sink.put(0x425a, 16); sink.put(0x68, 8); sink.put((uint64_t)(0x30 + level), 8);   // :48-50
    uint32_t total_crc = 0;
    const size_t n_blocks_all = (len + raw_block - 1) / raw_block;
    for (size_t b0 = 0; b0 < n_blocks_all; b0 += kBlocksPerLaunch) {
        const uint32_t nb = (uint32_t)(n_blocks_all - b0 < kBlocksPerLaunch ? n_blocks_all - b0 : kBlocksPerLaunch);
        const uint8_t* chunk = data + b0 * raw_block;
        const size_t chunk_len = (size_t)(b0 + nb == n_blocks_all ? len - b0 * raw_block : (size_t)nb * raw_block);
        x.begin_chunk();
        std::vector<uint32_t> raw_off(nb + 1), area_off(nb + 1);
        for (uint32_t b = 0; b <= nb; b++) raw_off[b] = (uint32_t)(b * raw_block < chunk_len ? b * raw_block : chunk_len);
        area_off[0] = 0;
        for (uint32_t b = 0; b < nb; b++) area_off[b + 1] = area_off[b] + rle1_bound(raw_off[b + 1] - raw_off[b]);
        BlockPtrs p;
        memset(&p, 0, sizeof(p));
        uint8_t* d_raw = (uint8_t*)x.alloc(chunk_len + 16);
        uint32_t* d_offs = (uint32_t*)x.alloc(4 * (size_t)(nb + 1) * 4);            // raw_off | area_off | off | out_off
        uint8_t* d_area = (uint8_t*)x.alloc(area_off[nb] + 16);
        BlockInfo* d_infos = (BlockInfo*)x.alloc(sizeof(BlockInfo) * nb);
        if (!d_raw || !d_offs || !d_area || !d_infos) return SWC_E_DEVICE;
        x.upload(d_raw, chunk, chunk_len);
        x.upload(d_offs, raw_off.data(), 4 * (nb + 1));
        x.upload(d_offs + (nb + 1), area_off.data(), 4 * (nb + 1));
        x.zero(d_infos, sizeof(BlockInfo) * nb);
        p.raw = (const SWC_AS_GLOBAL uint8_t*)d_raw;
        p.raw_off = (const SWC_AS_GLOBAL uint32_t*)d_offs;
        p.area = (SWC_AS_GLOBAL uint8_t*)d_area;
        p.area_off = (const SWC_AS_GLOBAL uint32_t*)(d_offs + (nb + 1));
        p.infos = (SWC_AS_GLOBAL BlockInfo*)d_infos;
        x.per_block(nb, Rle1Block{p});
        std::vector<uint32_t> crcs(nb);
        if (x.block_crcs(d_raw, raw_off.data(), nb, crcs.data())) return SWC_E_DEVICE;
        std::vector<BlockInfo> infos(nb);
        x.download(infos.data(), d_infos, sizeof(BlockInfo) * nb);
        // the blocks one behind the other
        std::vector<uint32_t> off(nb + 1), out_off(nb + 1);
        off[0] = 0; out_off[0] = 0;
        uint32_t longest = 0;
        for (uint32_t b = 0; b < nb; b++) {
            off[b + 1] = off[b] + infos[b].n_rle;
            out_off[b + 1] = out_off[b] + out_bound(infos[b].n_rle);
            if (infos[b].n_rle > longest) longest = infos[b].n_rle;
        }
        const uint32_t total = off[nb];
        x.upload(d_offs + 2 * (nb + 1), off.data(), 4 * (nb + 1));
        x.upload(d_offs + 3 * (nb + 1), out_off.data(), 4 * (nb + 1));
        uint8_t* d_text = (uint8_t*)x.alloc((size_t)total + 16);
        uint8_t* d_blk = (uint8_t*)x.alloc((size_t)total + 16);
        uint8_t* d_col = (uint8_t*)x.alloc((size_t)total + 16);
        if (!d_text || !d_blk || !d_col) return SWC_E_DEVICE;
        p.text = (SWC_AS_GLOBAL uint8_t*)d_text;
        p.blk = (SWC_AS_GLOBAL uint8_t*)d_blk;
        p.off = (const SWC_AS_GLOBAL uint32_t*)(d_offs + 2 * (nb + 1));
        p.col = (const SWC_AS_GLOBAL uint8_t*)d_col;
        p.out_off = (const SWC_AS_GLOBAL uint32_t*)(d_offs + 3 * (nb + 1));
        x.per_block(nb, GatherBlock{p});
        // ---- the sort
        {
            Bwt c;
            memset(&c, 0, sizeof(c));
            const size_t t4 = 4 * (size_t)total + 16, t8 = 8 * (size_t)total + 16;
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
            if (!d_rank || !d_sa || !d_k0 || !d_k1 || !d_v0 || !d_v1 || !d_s0 || !d_s1 || !d_head || !d_keep || !d_kpos) return SWC_E_DEVICE;
            c.text = p.text; c.blk = p.blk; c.off = p.off; c.infos = p.infos;
            c.rank = (SWC_AS_GLOBAL uint32_t*)d_rank;
            c.sa = (SWC_AS_GLOBAL uint32_t*)d_sa;
            c.key_in = (SWC_AS_GLOBAL uint64_t*)d_k0;
            c.key_out = (SWC_AS_GLOBAL uint64_t*)d_k1;
            c.val_in = (SWC_AS_GLOBAL uint32_t*)d_v0;
            c.val_out = (SWC_AS_GLOBAL uint32_t*)d_v1;
            c.head = (SWC_AS_GLOBAL uint32_t*)d_head;
            c.keep = (SWC_AS_GLOBAL uint32_t*)d_keep;
            c.kpos = (SWC_AS_GLOBAL uint32_t*)d_kpos;
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
                if (x.sort_pairs(d_k0, d_k1, d_v0, d_v1, c.m, key_bits)) return SWC_E_DEVICE;
                x.each(c.m, Heads{c});
                if (x.scan_max(d_head, c.m)) return SWC_E_DEVICE;
                x.each(c.m, Ranks{c});
                if (x.scan_sum(d_keep, d_kpos, c.m)) return SWC_E_DEVICE;
                x.each(c.m, Compact{c});
                uint32_t last[2] = {0, 0};
                x.download(&last[0], d_kpos + (c.m - 1u), 4);
                x.download(&last[1], d_keep + (c.m - 1u), 4);
                c.m = last[0] + last[1];
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
        // ---- symbols, lengths, bits
        uint16_t* d_syms = (uint16_t*)x.alloc(2 * ((size_t)total + 2u * nb) + 16);
        uint8_t* d_outs = (uint8_t*)x.alloc((size_t)out_off[nb] + 16);
        if (!d_syms || !d_outs) return SWC_E_DEVICE;
        p.syms = (SWC_AS_GLOBAL uint16_t*)d_syms;
        p.outs = (SWC_AS_GLOBAL uint8_t*)d_outs;
        x.per_block(nb, MtfBlock{p});
        x.download(infos.data(), d_infos, sizeof(BlockInfo) * nb);
        for (uint32_t b = 0; b < nb; b++) {
            infos[b].crc = crcs[b];
            huffman_lengths(infos[b].freq, infos[b].n_used + 2u, 17, infos[b].code);
        }
        x.upload(d_infos, infos.data(), sizeof(BlockInfo) * nb);
        x.per_block(nb, EmitBlock{p});
        x.download(infos.data(), d_infos, sizeof(BlockInfo) * nb);
        std::vector<uint64_t> at(nb + 1);
        at[0] = sink.bits & 7u;
        for (uint32_t b = 0; b < nb; b++) at[b + 1] = at[b] + infos[b].out_bits;
        const size_t stream_bytes = (size_t)((at[nb] + 31u) / 32u * 4u) + 16;
        uint8_t* d_stream = (uint8_t*)x.alloc(stream_bytes);
        uint64_t* d_at = (uint64_t*)x.alloc(8 * (size_t)(nb + 1));
        if (!d_stream || !d_at) return SWC_E_DEVICE;
        x.zero(d_stream, stream_bytes);
        x.upload(d_at, at.data(), 8 * (size_t)(nb + 1));
        p.stream = (SWC_AS_GLOBAL uint8_t*)d_stream;
        p.stream_at = (const SWC_AS_GLOBAL uint64_t*)d_at;
        x.per_block(nb, JoinBlock{p});
        std::vector<uint8_t> joined((size_t)((at[nb] + 7u) >> 3));
        x.download(joined.data(), d_stream, joined.size());
        sink.append_aligned(joined.data(), at[nb] - at[0]);
        for (uint32_t b = 0; b < nb; b++) {
            total_crc = (total_crc << 1) | (total_crc >> 31);                     // :56-57
            total_crc ^= crcs[b];
        }
        x.end_chunk();
    }
    sink.put(kEosMagic, 48);                                                      // :67-71
    sink.put(total_crc, 32);
    result.swap(sink.bytes);
    return SWC_OK;
}

}  // namespace bz2c
}  // namespace swc
#endif
