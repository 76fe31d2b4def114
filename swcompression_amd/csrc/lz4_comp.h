// lz4_comp.h -- LZ4 block COMPRESSION, one block per WAVEFRONT (the first piece of the encode side).
//
// Replaces LZ4.compress(block:_:) (reference Sources/LZ4/LZ4+Compress.swift:157-281): a greedy match search over a table of
// the most recent position of every four-byte group, minimum match 4, offsets up to 65,535, the last five bytes always
// literals and the last match starting at least twelve bytes before the end (:185-214), sequences written as token /
// length extensions / literals / offset / length extensions (:216-250), a final literals-only sequence (:253-277).
//
// The reference's table is an exact dictionary (Swift Dictionary keyed by the four bytes); a GPU wave keeps a HASH table of
// 8,192 positions in LDS instead (16-bit entries: the low half of the position -- a match reaches 65,535 bytes back, so the
// half is the distance; 16 KB per wave = two waves per SIMD, which is what hides the latency of this kernel's dependent
// global reads), looks 64 consecutive positions up at once, and takes the matches of such a window greedily from the left.  The output therefore is A valid LZ4 block for the same bytes, not the reference's bytes: the contract of
// this path is decode(compress(x)) == x under the reference decoder's rules (LZ4.swift:332-413, end-of-block rules
// included), checked against the oracle and liblz4 -- not byte parity of the compressed stream.
//
//   window   lane i hashes the four bytes at pos + i, reads the table's candidate, then enters its own position (of the lanes of
//            a window that share a hash the HIGHEST enters: found with one ballot per hash bit, so the result does not depend
//            on the order of the lanes); a candidate counts if it lies 1 .. 65,535 bytes back, not in front of the buffer,
//            and its four bytes are equal -- an entry that is stale or was never written only yields a candidate that fails;
//   greedy   the leftmost lane with a candidate: all 64 lanes extend its match together (64 bytes per step), the sequence is
//            written -- literals copied by all lanes -- and the search goes on behind the match, inside the window or beyond;
//   prefix   a dictionary / the tail of the previous block (dependent blocks) is handed over as bytes IN FRONT of the block in
//            the same buffer: its positions are entered into the table, matches may reach into it.
#ifndef SWC_LZ4_COMP_H
#define SWC_LZ4_COMP_H

#include "swc_common.h"
#include "simt.h"

namespace swc {
namespace lz4c {

#ifndef SWC_LZ4C_HASH_BITS
#define SWC_LZ4C_HASH_BITS 13
#endif
// 8,192 positions = 16 KB of LDS per wave.  The size of the table is the compression ratio: against the reference's exact
// dictionary (oracle/rc_lz4c.c) on text 2,048 entries lose 22 %, 4,096 12 %, 8,192 5 %, 16,384 1 % (tests/test_lz4_compress.py).
constexpr uint32_t kHashBits = SWC_LZ4C_HASH_BITS, kHashSize = 1u << kHashBits;
constexpr uint32_t kLdsBytes = kHashSize * 2;

SWC_HD uint32_t hash4(uint32_t w) { return (w * 2654435761u) >> (32 - kHashBits); }
// LZ4_compressBound: the largest block `n` bytes can turn into (all literals)
SWC_HD uint64_t bound(uint64_t n) { return n + n / 255 + 16; }

template <int N>
struct Compressor {
    gcptr src;          // prefix ++ block
    uint64_t start;     // first byte of the block in src
    uint64_t end;       // one past its last byte
    gptr out;
    uint64_t cap, opos; // output capacity, bytes written (keeps counting past the capacity)
    uint16_t* table;    // kHashSize entries: the low 16 bits of the most recent position of the hash
    uint64_t tbase;     // the buffer's first byte a match may start at

    // `cnt` bytes src[from ..] -> out, all lanes
    SWC_D void copy_out(uint64_t from, uint64_t cnt) {
        const uint64_t keep = opos >= cap ? 0 : (cap - opos < cnt ? cap - opos : cnt);
        SIMT_BEGIN(t, N)
            for (uint64_t i = 8ull * (uint32_t)t; i < keep; i += 8ull * N) {
                if (i + 8 <= keep) store_u64(out + opos + i, load_u64(src + from + i));
                else for (uint64_t j = i; j < keep; j++) out[opos + j] = src[from + j];
            }
        SIMT_END
        opos += cnt;
    }
    SWC_D void put(uint32_t b) {
        SIMT_BEGIN(t, N) if (t == 0 && opos < cap) out[opos] = (uint8_t)b; SIMT_END
        opos++;
    }
    // a length beyond the token's fifteen: 255, 255, ..., rest (LZ4+Compress.swift:220-228, 240-248)
    SWC_D void put_extension(uint64_t v) {
        const uint64_t full = v / 255;
        SIMT_BEGIN(t, N) for (uint64_t i = (uint32_t)t; i < full; i += N) if (opos + i < cap) out[opos + i] = 255; SIMT_END
        opos += full;
        put((uint32_t)(v - full * 255));
    }
    // token | literal-length extension | literals | offset | match-length extension (mlen == 0: the final, literals-only sequence)
    SWC_D void sequence(uint64_t lit_from, uint64_t lit, uint32_t offset, uint64_t mlen) {
        const uint32_t tl = lit < 15 ? (uint32_t)lit : 15u, tm = mlen == 0 ? 0u : (mlen - 4 < 15 ? (uint32_t)(mlen - 4) : 15u);
        put((tl << 4) | tm);
        if (lit >= 15) put_extension(lit - 15);
        copy_out(lit_from, lit);
        if (mlen != 0) {
            put(offset & 0xFFu);
            put(offset >> 8);
            if (mlen - 4 >= 15) put_extension(mlen - 19);
        }
    }

    SWC_D void run() {
        using simt::PT;
        SIMT_BEGIN(t, N) for (uint32_t i = (uint32_t)t; i < kHashSize / 2; i += N) ((uint32_t*)table)[i] = 0; SIMT_END_WAVE
        PT<uint32_t, N> cand, word, hsh;
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
        // the prefix enters the table (its last 65,535 bytes are all a match can reach)
        tbase = start > 65536 ? start - 65536 : 0;
        for (uint64_t p0 = tbase; p0 + 4 <= start; p0 += N) {
            SIMT_BEGIN(t, N)
                const uint64_t p = p0 + (uint32_t)t;
                pb[t] = p + 4 <= start;
                hsh[t] = pb[t] ? hash4(load_u32(src + p)) : 0u;
            SIMT_END
            highest_of_equals(simt::wave_ballot<N>(pb));
            SIMT_BEGIN(t, N) if (last[t]) table[hsh[t]] = (uint16_t)(p0 + (uint32_t)t); SIMT_END_WAVE
        }
        uint64_t pos = start, anchor = start;
        // a match starts at or before end - 12 and ends at or before end - 5 (LZ4+Compress.swift:185, 204-214)
        const uint64_t mflimit = end >= start + 12 ? end - 12 : 0, matchlimit = end >= 5 ? end - 5 : 0;
        while (end >= start + 13 && pos <= mflimit) {
            SIMT_BEGIN(t, N)
                const uint64_t p = pos + (uint32_t)t;
                const bool ok = p <= mflimit;
                const uint32_t w = ok ? load_u32(src + p) : 0u;
                word[t] = w;
                hsh[t] = ok ? hash4(w) : 0u;
                cand[t] = ok ? (uint32_t)table[hsh[t]] : 0u;
                pb[t] = ok;
            SIMT_END_WAVE
            highest_of_equals(simt::wave_ballot<N>(pb));
            SIMT_BEGIN(t, N)
                const uint64_t p = pos + (uint32_t)t;
                if (last[t]) table[hsh[t]] = (uint16_t)p;
                bool v = false;
                if (p <= mflimit) {
                    const uint32_t d = ((uint32_t)p - cand[t]) & 0xFFFFu;      // the entry is the low half of a position: this is the distance
                    v = d != 0u && (uint64_t)d <= p - tbase && load_u32(src + p - d) == word[t];
                }
                pb[t] = v;
                cand[t] = ((uint32_t)p - cand[t]) & 0xFFFFu;                 // from here on: the distance
            SIMT_END_WAVE
            const uint64_t m = simt::wave_ballot<N>(pb);
            uint32_t cur = 0;
            while (cur < (uint32_t)N) {
                const uint64_t m2 = m & ~((cur == 0 ? 0ull : (1ull << cur) - 1ull));
                if (m2 == 0) break;
                const uint32_t f = (uint32_t)simt::ctz64(m2);
                const uint64_t mp = pos + f;
                if (mp < anchor) { cur = f + 1; continue; }     // (inside the match just written)
                const uint64_t c = mp - simt::uniform(simt::wave_read<N>(cand, (int)f));
                uint64_t len = 4;
                for (;;) {   // all lanes extend the match, 64 bytes per step
                    SIMT_BEGIN(t, N)
                        const uint64_t a = mp + len + (uint32_t)t;
                        pb[t] = !(a < matchlimit && src[a] == src[c + len + (uint32_t)t]);
                    SIMT_END
                    const uint64_t mm = simt::wave_ballot<N>(pb);
                    if (mm) { len += (uint32_t)simt::ctz64(mm); break; }
                    len += N;
                }
                sequence(anchor, mp - anchor, (uint32_t)(mp - c), len);
                anchor = mp + len;
                cur = anchor - pos >= (uint64_t)N ? (uint32_t)N : (uint32_t)(anchor - pos);
            }
            pos = anchor > pos + N ? anchor : pos + N;
        }
        sequence(anchor, end - anchor, 0, 0);   // the rest as literals (:253-277; also a block of fewer than 13 bytes)
    }
};

// One wavefront = one job: job.in = prefix ++ block, job.dict_len = length of the prefix, job.in_len = both together.
// job.out_len = bytes of the compressed block (SWC_E_CAPACITY with the size needed if it does not fit out_cap).
template <int N>
SWC_D void lz4_compress_job(Job& job, uint16_t* table) {
    Compressor<N> c;
    c.src = (gcptr)job.in;
    c.start = job.dict_len <= job.in_len ? job.dict_len : job.in_len;
    c.end = job.in_len;
    c.out = (gptr)job.out;
    c.cap = job.out_cap;
    c.opos = 0;
    c.table = table;
    int st = SWC_OK;
    if (job.in_len - c.start > 0x7E000000ull) st = SWC_E_INVALID_ARGUMENT;   // (LZ4_MAX_INPUT_SIZE)
    else c.run();
    if (st == SWC_OK && c.opos > c.cap) st = SWC_E_CAPACITY;
    job.out_len = c.opos;
    job.in_consumed = job.in_len;
    job.status = st;
}

}  // namespace lz4c
}  // namespace swc
#endif
