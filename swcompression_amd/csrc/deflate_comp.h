// deflate_comp.h -- Deflate COMPRESSION, one buffer per WAVEFRONT (the second piece of the encode side).
//
// Replaces Deflate.compress(data:) (reference Sources/Deflate/Deflate+Compress.swift:22-213): ONE block for the whole buffer --
// static Huffman (RFC 1951 3.2.6) over a greedy LZ77 parse (minimum match 3, maximum 258, distances up to 32,768, the last
// two bytes always literals, :146-213), or a stored block when that is not larger and the buffer fits its 16-bit length
// (:30-45, with the reference's own size formula :48-83 applied to THIS parse).
//
// The reference finds matches through an exact dictionary of the most recent position of every three-byte group it has looked
// up (a Swift Dictionary); a GPU wave keeps a HASH table of 4,096 positions in LDS (16-bit entries: a match reaches 32,768
// bytes back, so the low half of a position is enough to find the distance), looks 64 consecutive positions up at once --
// of the lanes of a window that share a hash the HIGHEST enters its position, found with one ballot per hash bit, so the
// result does not depend on the order of the lanes -- and takes the matches of the window greedily from the left; all 64
// lanes extend a match together.  The output therefore is A valid Deflate stream for the same bytes, not the reference's
// bytes: the contract of this path is decode(compress(x)) == x under the reference decoder (Deflate.swift:30-249), zlib and the
// engine's own decoder, and a size close to that of the reference's encoder restated (oracle/rc_deflatec.c) -- not byte
// parity of the compressed stream.
//
//   bits     the codes of a sequence -- up to 62 literals, the length code with its extra bits, the distance code with its
//            extra bits -- are computed one per lane (the static codes are arithmetic: no table), a wave scan of their bit
//            counts gives every lane its place, and the lanes OR their bits into a staging area of the bit stream in LDS
//            (ds_or_b32); whole dwords leave for HBM when half of the area is full.
#ifndef SWC_DEFLATE_COMP_H
#define SWC_DEFLATE_COMP_H

#include "swc_common.h"
#include "simt.h"

namespace swc {
namespace defc {

// Positions in the hash table: 13 / 12 / 11 bits = 17 / 9 / 5 KB of LDS per wave = 9 / 17 / 30 waves per CU.  100,000 x 64 KiB:
// 443 / 312 / 256 ms, 0.999 / 1.015 / 1.050 x the size of the reference encoder restated (profiles/r05_experiments.txt): 12.
#ifndef SWC_DEFC_HASH_BITS
#define SWC_DEFC_HASH_BITS 12
#endif
constexpr uint32_t kHashBits = SWC_DEFC_HASH_BITS, kHashSize = 1u << kHashBits;
constexpr uint32_t kStageDw = 256;                 // dwords of the bit stream staged in LDS (an emission adds at most 64)
constexpr uint32_t kFlushDw = 128;
constexpr uint32_t kMaxMatch = 258, kMaxDist = 32768;

struct Lds {
    alignas(16) uint16_t table[kHashSize];         // the low 16 bits of the most recent position of the hash
    alignas(16) uint32_t stage[kStageDw + 2];
};

SWC_HD uint32_t hash3(uint32_t w) { return ((w & 0xFFFFFFu) * 2654435761u) >> (32 - kHashBits); }
// the largest stream `n` bytes can turn into: a static block of nine-bit literals, or the stored block
SWC_HD uint64_t bound(uint64_t n) { return n + n / 8 + 16; }
SWC_HD uint32_t rev_bits(uint32_t v, uint32_t n) { return brev32(v) >> (32u - n); }
SWC_HD uint32_t log2u(uint32_t v) {   // floor(log2 v), v != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return 31u - (uint32_t)__clz((int)v);
#else
    return 31u - (uint32_t)__builtin_clz(v);
#endif
}
// static code of a literal byte: (bits, count), most significant code bit first in the stream (RFC 1951 3.2.6)
SWC_HD void literal_code(uint32_t v, uint32_t& code, uint32_t& nb) {
    if (v < 144u) { code = rev_bits(0x30u + v, 8); nb = 8; }
    else { code = rev_bits(0x190u + v - 144u, 9); nb = 9; }
}
SWC_HD void litlen_code(uint32_t sym, uint32_t& code, uint32_t& nb) {   // symbols 256..285
    if (sym < 280u) { code = rev_bits(sym - 256u, 7); nb = 7; }
    else { code = rev_bits(0xC0u + sym - 280u, 8); nb = 8; }
}
// length 3..258 -> its code followed by its extra bits (Deflate+Constants.swift lengthCode / lengthBase as arithmetic)
SWC_HD void length_bits(uint32_t len, uint32_t& code, uint32_t& nb) {
    const uint32_t l = len - 3u;
    uint32_t sym, e, extra;
    if (len == 258u) { sym = 285; e = 0; extra = 0; }
    else if (l < 8u) { sym = 257u + l; e = 0; extra = 0; }
    else { e = log2u(l) - 2u; sym = 261u + 4u * e + ((l >> e) & 3u); extra = l & ((1u << e) - 1u); }
    uint32_t c, n;
    litlen_code(sym, c, n);
    code = c | (extra << n);
    nb = n + e;
}
// distance 1..32768 -> its five-bit code followed by its extra bits (distanceBase as arithmetic)
SWC_HD void distance_bits(uint32_t dist, uint32_t& code, uint32_t& nb) {
    const uint32_t d = dist - 1u;
    uint32_t sym, e, extra;
    if (d < 4u) { sym = d; e = 0; extra = 0; }
    else { const uint32_t p = log2u(d); e = p - 1u; sym = 2u * p + ((d >> e) & 1u); extra = d & ((1u << e) - 1u); }
    code = rev_bits(sym, 5) | (extra << 5);
    nb = 5u + e;
}

template <int N>
struct Compressor {
    gcptr src;
    uint64_t n;          // bytes of input
    gptr out;            // 4-byte aligned
    uint64_t cap;
    Lds* l;
    uint64_t obits;      // bits of the stream so far (keeps counting past the capacity)
    uint64_t odw;        // dwords that have left the stage for `out`
    uint32_t fill;       // bits in the stage
    bool nonfinal = false;   // a segment of a larger buffer: BFINAL clear, an empty stored block behind the block (job.aux bit 0)

    SWC_D static void lds_or(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        *p |= v;
#endif
    }
    // whole dwords of the stage -> out; the incomplete one moves to the front (`all`: the incomplete one too, at the end)
    SWC_D void flush(bool all) {
        const uint32_t nd = all ? (fill + 31u) >> 5 : fill >> 5;
        const uint64_t o0 = odw;
        simt::PT<uint32_t, N> carry;
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < nd; i += (uint32_t)N) {
                const uint64_t b = 4ull * (o0 + i);
                const uint32_t w = l->stage[i];
                if (b + 4u <= cap) *(SWC_AS_GLOBAL uint32_t*)(out + b) = w;
                else for (uint32_t e = 0; e < 4u; e++) if (b + e < cap) out[b + e] = (uint8_t)(w >> (8u * e));
            }
            carry[t] = l->stage[nd];
        SIMT_END_WAVE
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < kStageDw + 2u; i += (uint32_t)N) l->stage[i] = i == 0u && !all ? carry[t] : 0u;
        SIMT_END_WAVE
        odw += nd;
        fill = all ? 0u : fill & 31u;
    }
    // every lane adds `nb` bits (0..32) of `code`, least significant first, lane after lane
    SWC_D void emit(const simt::PT<uint32_t, N>& code, const simt::PT<uint32_t, N>& nb) {
        simt::PT<uint32_t, N> x;
        SIMT_BEGIN(t, N) x[t] = nb[t]; SIMT_END
        simt::wave_scan_incl<N>(x);
        const uint32_t f0 = fill;
        SIMT_BEGIN(t, N)
            if (nb[t] != 0u) {
                const uint32_t b = f0 + x[t] - nb[t], d = b >> 5, s = b & 31u;
                lds_or(&l->stage[d], code[t] << s);
                if (s + nb[t] > 32u) lds_or(&l->stage[d + 1u], code[t] >> (32u - s));
            }
        SIMT_END_WAVE
        const uint32_t total = simt::wave_read<N>(x, N - 1);
        fill += total;
        obits += total;
        if (fill >= 32u * kFlushDw) flush(false);
    }
    // `lit` literals from src[from], then (mlen != 0) a match, then (eob) the end-of-block code
    SWC_D void sequence(uint64_t from, uint64_t lit, uint32_t dist, uint32_t mlen, bool eob) {
        simt::PT<uint32_t, N> code, nb;
        const uint64_t tail = (mlen != 0u ? 2u : 0u) + (eob ? 1u : 0u);
        for (;;) {
            const uint32_t k = lit > (uint64_t)N - tail ? (lit >= (uint64_t)N ? (uint32_t)N : (uint32_t)lit) : (uint32_t)lit;
            const bool last = (uint64_t)k == lit && (uint64_t)k + tail <= (uint64_t)N;
            SIMT_BEGIN(t, N)
                uint32_t c = 0, b = 0;
                if ((uint32_t)t < k) literal_code(src[from + (uint32_t)t], c, b);
                else if (last && mlen != 0u && (uint32_t)t == k) length_bits(mlen, c, b);
                else if (last && mlen != 0u && (uint32_t)t == k + 1u) distance_bits(dist, c, b);
                else if (last && eob && (uint32_t)t == k + (mlen != 0u ? 2u : 0u)) litlen_code(256u, c, b);
                code[t] = c; nb[t] = b;
            SIMT_END
            emit(code, nb);
            from += k;
            lit -= k;
            if (last) break;
        }
    }

    SWC_D void run() {
        using simt::PT;
        obits = 0; odw = 0; fill = 0;
        SIMT_BEGIN(t, N)
            for (uint32_t i = (uint32_t)t; i < kHashSize / 2; i += (uint32_t)N) ((uint32_t*)l->table)[i] = 0;
            for (uint32_t i = (uint32_t)t; i < kStageDw + 2u; i += (uint32_t)N) l->stage[i] = 0;
        SIMT_END_WAVE
        PT<uint32_t, N> cand, word, hsh, code, nb;
        PT<bool, N> pb, last;
        // of the lanes that hold the same hash, the highest: one ballot per bit of the hash narrows the set of equals
        auto highest_of_equals = [&](uint64_t valid) {
            PT<uint32_t, N> mlo, mhi;
            SIMT_BEGIN(t, N) mlo[t] = (uint32_t)valid; mhi[t] = (uint32_t)(valid >> 32); SIMT_END
            for (uint32_t b = 0; b < kHashBits; b++) {
                SIMT_BEGIN(t, N) pb[t] = ((hsh[t] >> b) & 1u) != 0u; SIMT_END
                const uint64_t bal = simt::wave_ballot<N>(pb);
                SIMT_BEGIN(t, N)
                    const uint64_t same = ((hsh[t] >> b) & 1u) ? bal : ~bal;
                    mlo[t] &= (uint32_t)same; mhi[t] &= (uint32_t)(same >> 32);
                SIMT_END
            }
            SIMT_BEGIN(t, N)
                const uint64_t m = ((uint64_t)mhi[t] << 32) | mlo[t];
                last[t] = ((valid >> t) & 1ull) != 0ull && (t == N - 1 || (m >> (t + 1)) == 0ull);
            SIMT_END
        };
        // block header: BFINAL = 1, BTYPE = 01 (Deflate+Compress.swift:103-104); a SEGMENT of a larger buffer (`nonfinal`,
        // framing_deflate.cpp) leaves BFINAL clear
        SIMT_BEGIN(t, N) code[t] = t == 0 ? (nonfinal ? 2u : 3u) : 0u; nb[t] = t == 0 ? 3u : 0u; SIMT_END
        emit(code, nb);
        uint64_t pos = 0, anchor = 0;
        const uint64_t plimit = n >= 3 ? n - 3 : 0;     // the last position a match may start at (:155: i < endIndex - 2)
        while (n >= 3 && pos <= plimit) {
            SIMT_BEGIN(t, N)
                const uint64_t p = pos + (uint32_t)t;
                const bool ok = p <= plimit;
                uint32_t w = 0;
                if (ok) w = p + 4 <= n ? load_u32(src + p) & 0xFFFFFFu : (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16);
                word[t] = w;
                hsh[t] = ok ? hash3(w) : 0u;
                cand[t] = ok ? (uint32_t)l->table[hsh[t]] : 0u;
                pb[t] = ok;
            SIMT_END_WAVE
            highest_of_equals(simt::wave_ballot<N>(pb));
            SIMT_BEGIN(t, N)
                const uint64_t p = pos + (uint32_t)t;
                if (last[t]) l->table[hsh[t]] = (uint16_t)p;
                bool v = false;
                const uint32_t d = ((uint32_t)p - cand[t]) & 0xFFFFu;      // the entry is the low half of a position: this is the distance
                if (p <= plimit && d != 0u && d <= kMaxDist && (uint64_t)d <= p)
                    v = (load_u32(src + p - d) & 0xFFFFFFu) == word[t];    // (p - d + 4 <= p + 3 <= n)
                pb[t] = v;
                cand[t] = d;                                                  // from here on: the distance
            SIMT_END_WAVE
            const uint64_t m = simt::wave_ballot<N>(pb);
            uint32_t cur = 0;
            while (cur < (uint32_t)N) {
                const uint64_t m2 = m & ~((cur == 0 ? 0ull : (1ull << cur) - 1ull));
                if (m2 == 0) break;
                const uint32_t f = (uint32_t)simt::ctz64(m2);
                const uint64_t mp = pos + f;
                if (mp < anchor) { cur = f + 1; continue; }     // (inside the match just written)
                const uint32_t dist = simt::uniform(simt::wave_read<N>(cand, (int)f));
                const uint64_t c = mp - dist;
                uint32_t len = 3;
                for (;;) {   // all lanes extend the match, 64 bytes per step (:185: up to 258, not past the end)
                    SIMT_BEGIN(t, N)
                        const uint64_t a = mp + len + (uint32_t)t;
                        pb[t] = !(a < n && len + (uint32_t)t < kMaxMatch && src[a] == src[c + len + (uint32_t)t]);
                    SIMT_END
                    const uint64_t mm = simt::wave_ballot<N>(pb);
                    if (mm) { len += (uint32_t)simt::ctz64(mm); break; }
                    len += N;
                }
                sequence(anchor, mp - anchor, dist, len, false);
                anchor = mp + len;
                cur = anchor - pos >= (uint64_t)N ? (uint32_t)N : (uint32_t)(anchor - pos);
            }
            pos = anchor > pos + N ? anchor : pos + N;
        }
        sequence(anchor, n - anchor, 0, 0, true);   // the rest as literals, the end-of-block code (:198-207, :135)
        if (nonfinal) {
            // an empty stored block behind it (000, the padding to the byte, LEN = 0, NLEN = 0xFFFF): the segment ends on a byte,
            // and the segments of a buffer are one Deflate stream when they are put one behind the other
            const uint32_t pad = (8u - (uint32_t)((obits + 3u) & 7u)) & 7u;
            SIMT_BEGIN(t, N)
                code[t] = t == 3 ? 0xFFFFu : 0u;
                nb[t] = t == 0 ? 3u : t == 1 ? pad : t == 2 || t == 3 ? 16u : 0u;
            SIMT_END
            emit(code, nb);
        }
        flush(true);
    }
};

// One wavefront = one job: job.in / in_len = the buffer, job.out (4-byte aligned) / out_cap = room for the stream.
// job.out_len = bytes of the stream (SWC_E_CAPACITY with the size needed if it does not fit out_cap).
template <int N>
SWC_D void deflate_compress_job(Job& job, Lds* lds) {
    Compressor<N> c;
    c.src = (gcptr)job.in;
    c.n = job.in_len;
    c.out = (gptr)job.out;
    c.cap = job.out_cap;
    c.l = lds;
    c.nonfinal = (job.aux & 1) != 0;
    c.run();
    uint64_t size = (c.obits + 7) >> 3;
    // Deflate+Compress.swift:30-45: stored if not larger than the static block and the length fits 16 bits
    if (5 + c.n <= size && 5 + c.n <= 65535) {
        const uint32_t nn = (uint32_t)c.n, nl = nn ^ 0xFFFFu;
        size = 5 + c.n;
        gptr o = c.out;
        gcptr s = c.src;
        const uint64_t cap = c.cap;
        SIMT_BEGIN(t, N)
            if (t == 0) {
                const uint8_t h[5] = {(uint8_t)(c.nonfinal ? 0 : 1), (uint8_t)(nn & 0xFFu), (uint8_t)(nn >> 8), (uint8_t)(nl & 0xFFu), (uint8_t)(nl >> 8)};
                for (uint32_t k = 0; k < 5u; k++) if (k < cap) o[k] = h[k];
            }
            for (uint32_t i = (uint32_t)t; i < nn; i += (uint32_t)N) if (5ull + i < cap) o[5u + i] = s[i];
        SIMT_END_WAVE
    }
    job.out_len = size;
    job.in_consumed = job.in_len;
    job.status = size > job.out_cap ? SWC_E_CAPACITY : SWC_OK;
}

}  // namespace defc
}  // namespace swc
#endif
