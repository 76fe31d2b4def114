// lz_copy.h -- phase 2 of the Deflate and LZ4 paths, record-granular: LZ77 match resolution, one stream per WAVEFRONT,
// the recent output in an LDS window, older output read back from HBM.
//
// The reference executes every back-reference inline, one byte per `out.append`
// (Sources/Deflate/Deflate.swift:216-232, Sources/LZ4/LZ4.swift:398-410).  Phase 1 (inflate_sync.h, lz4_wave.h) leaves a
// record list per stream in the workspace (format: lz_resolve.h) and -- Deflate -- a dense literal stream; LZ4's literals stay
// where they are in the block.  Here the 64 lanes of a wave OWN RECORDS, not bytes:
//
//   group      64 consecutive records, one per lane (fewer when they cover more than kSpanMax output bytes).  One packed prefix
//              scan over (literals + length, literals) gives every lane the place of its literals and of its match in the output
//              (LZ4: the second field is the bytes of the record's SEQUENCE in the block -- the running sum says where its
//              literals lie, see Copier);
//   literals   a lane's literal run -- up to 16 bytes (Deflate) / 32 (LZ4) -- is loaded by front() from the literal stream (LZ4:
//              from the compressed block itself) in HBM straight into registers, one group AHEAD, as eight-byte pieces, the last
//              one shifted back so that it ends on the run's last byte; back() stores it into the window: one to three bytes as
//              a byte and / or a word, four to eight as two dwords, more as eight-byte pieces -- no byte tails.  (Round 5 staged
//              the literal stream in an LDS buffer; the buffer's KiB is window now, and its reads are gone);
//   matches    a lane copies its match -- up to 32 bytes as four eight-byte pieces -- when its source is final at the start of
//              the group: it ends in front of the group's first byte (in the window), or it lies in front of the window (FAR:
//              then the bytes come from the output buffer in HBM and were asked for by front() one group ahead, into
//              registers).  READS from the window at any byte alignment are two or three ALIGNED dwords and a byte funnel shift
//              (v_alignbyte_b32): the LDS serves an unaligned ds_read at one LANE per cycle, an aligned one at ten lanes
//              (tools/micro/lds_bench.hip); unaligned stores cost a quarter of a cycle per lane and are used as they are;
//   the rest   -- matches that reach into their own group (9 % on Deflate text, 28 % on LZ4 text), matches longer than 32 bytes
//              or overlapping themselves (distance < length), far matches nobody asked for.  Nothing long, overlapping or far
//              among them and at most 256 bytes in all (nine groups in ten): a BYTE PER LANE -- a running sum of the lengths
//              numbers the bytes, a marker per match in LDS and a running maximum tell a lane which match its byte belongs to
//              -- and all lanes copy their byte together, round after round, until a round reads what the round before it read
//              (sources lie strictly in front of their bytes: that state is the reference's sequence).  Otherwise in record
//              order, one after the other, by ALL lanes together (a byte per lane for the short ones; a dword per lane, 256
//              bytes per step, for the long ones; an overlapping match from its first period), after a bitmap of the group's
//              output has let those that depend on none of the others go at once, each by its own lane;
//   window     a LINEAR array (6 KiB for Deflate, 7 KiB for LZ4 -- CfgDeflate / CfgLz4 below), not a ring: when a group does
//              not fit behind the write position any more, the last kKeep bytes (3.25 KiB) move to the front of the array (16
//              bytes per lane and step).  The slide is the expensive thing, not the far matches a short history makes (a far
//              source is four loads that are issued anyway): kKeep is what asking ahead needs, no more.  Finished bytes are
//              flushed to HBM a KiB at a time (aligned 16-byte stores, the only time the output is written) at the TOP of an
//              iteration, so that the next full wait finds stores that have had a whole group's copies to arrive.  No index
//              masks anywhere in the copy code; every position, watermark and difference is 32 bits (streams under 4 GiB).
//
// One iteration = front(next group) + back(this group).  front() issues ALL global loads of the next group -- its far
// sources, its literals, the records of the group behind it -- and nothing waits for them before the drain at the top of
// the next iteration (s_waitcnt vmcnt(0)): a full group's copies lie between a load and its use.
//
// No workgroup barrier, no cross-wave traffic: a wave is alone with its stream.  Parity: the output is a function of the
// record list alone; tests/test_lane_emulation_*.py run this source on the host (lane order forward, reverse and shuffled).
#ifndef SWC_LZ_COPY_H
#define SWC_LZ_COPY_H

#include <type_traits>
#include "swc_common.h"
#include "simt.h"
#include "lz_resolve.h"   // record format, StreamHeader, workspace layout

#ifndef SWC_LZC_CUT
#define SWC_LZC_CUT 0
#endif

namespace swc {
namespace lzc {

using lzr::u128;

template <uint32_t WIN>
struct Lds {
    static constexpr uint32_t kWin = WIN;
    alignas(16) uint8_t win[WIN + 16];                    // byte at virtual position v lives at win[v - vbase] (+16: reads of short runs overshoot)
    uint32_t pmap[64 + 2];                                // one bit per output byte of the group: it belongs to a match that has not been copied yet
};

SWC_HD uint32_t ld32(const uint8_t* p) { return *(const u32_unaligned*)p; }
// LDS READS at any byte alignment are built from ALIGNED dwords and a byte funnel shift (v_alignbyte_b32 takes the shift from
// the low bits of the address): the LDS serves an unaligned ds_read_b32 at about one LANE per cycle (63 cycles for a full
// wave, tools/micro/lds_bench.hip), an aligned one in 6; unaligned STORES cost a quarter of a cycle per lane and stay.
// `a` is the byte offset from the (16-byte aligned) base b; up to 7 (rd32u) / 11 (rd64u) bytes behind the value are touched.
SWC_HD uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, a);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (a & 3u)));
#endif
}
SWC_HD uint32_t rd32u(const uint8_t* b, uint32_t a) {
    const uint32_t* p = (const uint32_t*)(b + (a & ~3u));
    return alignbyte(p[1], p[0], a);
}
SWC_HD uint64_t rd64u(const uint8_t* b, uint32_t a) {
    const uint32_t* p = (const uint32_t*)(b + (a & ~3u));
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
    return (uint64_t)alignbyte(w1, w0, a) | ((uint64_t)alignbyte(w2, w1, a) << 32);
}
SWC_HD void st32(uint8_t* p, uint32_t v) { *(u32_unaligned*)p = v; }
SWC_HD void st64(uint8_t* p, uint64_t v) { *(u64_unaligned*)p = v; }
SWC_HD void st16(uint8_t* p, uint32_t v) { *(u16_unaligned*)p = (uint16_t)v; }
// n bytes (1..3) of the little-endian dword w to p
SWC_HD void st_tail(uint8_t* p, uint32_t w, uint32_t n) {
    if (n & 1u) p[0] = (uint8_t)w;
    if (n & 2u) st16(p + (n & 1u), w >> (8u * (n & 1u)));
}
// n >= 1 bytes from s to d (both in LDS, any alignment, not overlapping): dwords, the last one shifted back to end on byte n - 1
SWC_HD void copy_run(uint8_t* d, const uint8_t* s, uint32_t n) {
    if (n >= 4u) {
        for (uint32_t k = 0; k < n; k += 4u) {
            const uint32_t o = k < n - 4u ? k : n - 4u;
            st32(d + o, ld32(s + o));
        }
    } else {
        st_tail(d, ld32(s), n);
    }
}

// CFG: kWin (bytes of the LDS window), kSpan (output bytes one group may cover), kKeep (history that survives a slide),
// kLitPieces (eight-byte pieces of a literal run a lane copies on its own).  P: the type of a position in the output or in the
// literal stream -- uint32_t for streams of less than 4 GiB (every position, watermark and difference is ONE scalar register and
// one scalar instruction: the copier is bound as much by the CU's scalar unit as by its vector ALU), uint64_t for the rest.
// R8: records of EIGHT bytes -- the 32-bit record and, in the upper dword, the offset of its literal run from `lits`, which then
// is not a dense literal stream but the compressed input itself: LZ4 keeps its literals byte-aligned in the block (LZ4.swift:
// 364-366), so the parse kernel writes no literal stream at all and the copier fetches a run where the encoder left it.
#ifndef SWC_LZC_FAKE
#define SWC_LZC_FAKE 0   // (timing experiments only, wrong output: 1 = the literal loads, 2 = the far loads, 3 = both read one hot place)
#endif
#ifndef SWC_LZC_FLUSH_EARLY
#define SWC_LZC_FLUSH_EARLY 1
#endif
// RM == 2 ("R4"): four-byte records again, `lits` the compressed block, and the place of a record's literals DERIVED: the start S
// of a sequence is a running sum over the records of seq_bytes(literals, length) -- what a sequence in LZ4's short form takes --
// and its literals lie lit_skip() behind S.  The parse (lz4_wave.h), which simulates this sum, leaves an ANCHOR (record index, S)
// wherever the rule would go wrong; a group never reaches across an anchor, and one that starts at an anchor takes its S from it.
template <typename CFG, typename P, int RM = 0>
struct Copier {
    static constexpr bool R8 = RM == 1, R4 = RM == 2, INP = RM != 0;   // INP: the literals lie in the compressed block (`lits`), a run at g.loff
    SWC_HD static uint32_t seq_bytes(uint32_t li, uint32_t le) { return le ? 3u + li + (li >= 15u ? 1u : 0u) + (le >= 19u ? 1u : 0u) : li; }
    SWC_HD static uint32_t lit_skip(uint32_t li, uint32_t le) { return le ? 1u + (li >= 15u ? 1u : 0u) : 0u; }
    // the second field of the packed scan: literal bytes (the offset into the dense literal stream), R4: bytes of the sequence
    SWC_HD static uint32_t lit_step(uint32_t li, uint32_t le) { return R4 ? seq_bytes(li, le) : li; }
    const SWC_AS_GLOBAL uint32_t* anc = nullptr;   // R4: the anchors, (record index, S) pairs
    uint32_t nanc = 0, anc_i = 0;                  // ... how many, the next one to load
    uint32_t anc_rec = 0xFFFFFFFFu, anc_S = 0;     // ... the next anchor (0xFFFFFFFF: none left)
    SWC_D void next_anchor() {
        if (anc_i < nanc) { anc_rec = simt::uniform(anc[2u * anc_i]); anc_S = simt::uniform(anc[2u * anc_i + 1u]); anc_i++; }
        else anc_rec = 0xFFFFFFFFu;
    }
    static constexpr uint32_t WIN = CFG::kWin, SPAN = CFG::kSpan;
    using L = Lds<WIN>;
    using SP = typename std::conditional<sizeof(P) == 4, int32_t, int64_t>::type;
    static constexpr int W = 64;
    static constexpr uint32_t kSpanMax = SPAN;                 // output bytes one group may cover
    static constexpr uint32_t kKeep = CFG::kKeep;              // history that survives a slide (a slide leaves kKeep .. kKeep + 15 bytes)
    static constexpr uint32_t kBigLit = 256;                   // a literal-only record of at least this many bytes is copied by all lanes together, on its own
#ifndef SWC_LZC_PIECES
#define SWC_LZC_PIECES 4
#endif
    static constexpr int kPieces = SWC_LZC_PIECES;             // eight-byte pieces of a match a lane copies on its own (copy_own is written for up to four)
    static constexpr uint32_t kLongLen = 8u * kPieces;         // a longer match is copied by all lanes together
    static constexpr int kLitPieces = CFG::kLitPieces;         // eight-byte pieces of a literal run a lane copies on its own
    static constexpr uint32_t kLongLit = 8u * kLitPieces;      // a longer literal run in front of a match is copied by all lanes together
#ifndef SWC_LZC_SEQMAX
#define SWC_LZC_SEQMAX 8
#endif
    static constexpr uint32_t kSeqMax = SWC_LZC_SEQMAX;        // more short matches left than this: those that do not depend on each other first, all at once
    static constexpr uint32_t kBack = 65536;                   // the furthest a source lies behind its match (record format)
    static_assert(kKeep % 16u == 0u && kKeep + kSpanMax + 16u <= WIN, "a group fits behind what a slide keeps");
    static_assert(lzr::kLitRunMax + lzr::kMaxLen <= kSpanMax && kBigLit - 1u + lzr::kMaxLen <= kSpanMax, "a record must fit a group");
    static_assert(64u * (kBigLit - 1u + lzr::kMaxLen + lzr::kLitRunMax) < 0x10000u, "the packed scan keeps 16 bits per sum");
    static_assert(kSpanMax <= 2048u && kLongLen <= 32u, "the bitmap of a group's output has 2,048 bits, a short match 32");
    // A far source lies in front of the window, i.e. at least kKeep - kMaxLen bytes behind the write position.  Finished bytes
    // are flushed when a KiB of them has gathered, and what was issued before the last drain() has arrived: `landed`.  front()
    // asks only for sources that end 128 bytes (a cache line: a line is never read while a part of it is still on its way)
    // in front of `landed`; anything else waits for its turn in back() and drains first.  With the figures below a far source
    // has as good as always landed when the group AHEAD of its match is being copied.
    // (The flush watermark lies less than 1,040 bytes behind the position after every group, `landed` one group behind that:
    // a source that is asked for -- kLongLen bytes at most, kKeep and more behind its match -- has always landed.)
    static_assert(kKeep >= kLongLen + 1040u + (SWC_LZC_FLUSH_EARLY ? 2u : 1u) * kSpanMax + 128u, "far sources that are asked for ahead must have landed");
    enum : uint32_t { kFlagCoop = 1u, kFlagFar = 2u };

    L* l;
    gptr out;
    gcptr lits;        // the stream's dense literal stream (16-byte aligned base); R8: the compressed block
    P limit;           // bytes of `out` that exist: min(bytes produced, capacity)
    uint32_t nin;      // R8: bytes of `lits` that exist (the block's compressed size: nothing is read beyond it)

    // wave state (the same in every lane)
    uint32_t A;        // out & 15: virtual position v = A + output position, so that 16-byte chunks of v are aligned in HBM
    P vbase;           // virtual position of win[0] (a multiple of 16)
    P fv;              // flush watermark (a multiple of 16): virtual positions below it have LEFT for `out`
    P landed;          // ... and below this one they have ARRIVED (the flush watermark at the last full wait)

    SWC_D static uint32_t mod_small(uint32_t m, uint32_t d) {   // m % d for m, d < 2^16, d != 0
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t q = (uint32_t)((float)m * __builtin_amdgcn_rcpf((float)d));   // v_rcp_f32: off by at most one, fixed up below
        uint32_t r = m - q * d;
        if ((int32_t)r < 0) r += d;
        if (r >= d) r -= d;
        return r;
#else
        return m % d;
#endif
    }
    SWC_D static void lds_or(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        *p |= v;
#endif
    }
    SWC_D static void unpack(uint32_t r, uint32_t& li, uint32_t& le, uint32_t& di) {
        li = r & 127u;
        le = (r >> 7) & 511u;
        if (le == 0u) li += (r >> 16) << 7;
        di = le ? (r >> 16) + 1u : 0u;
    }
    // all of this wave's loads have returned and all of its stores have arrived in memory
    SWC_D void drain() {
#if !defined(SWC_LZC_NODRAIN)   // (timing experiments only: the output may be wrong)
        simt::vmem_fence();
#endif
        landed = fv;
    }
    // The loads of a path that is rarely taken (all lanes on one long run, a far source nobody asked for ahead) are waited for
    // before the path rejoins the others: behind the join the compiler would otherwise wait -- for EVERYTHING, the loads front()
    // has just issued for the next group included -- wherever one of that path's destination registers is touched again.
    SWC_D static void rare_path_done() { simt::vmem_fence(); }

    // ---- flush: virtual positions [fv, vend) rounded down to whole 16-byte chunks go to HBM; `final`: the tail bytes too
    SWC_D void flush(P vend, bool final) {
        const P v16 = vend & ~(P)15;
        const P vlim = (P)A + limit;                            // virtual end of the output that exists
        gptr obase = out - A;                                   // HBM address of virtual position 0 (16-byte aligned)
        if (v16 > fv) {
            const P f0 = fv;
            const uint32_t w0 = (uint32_t)(f0 - vbase);
            const uint32_t nchunk = (uint32_t)((v16 - f0) >> 4);
            gptr ob = obase + f0;
            const bool whole = f0 >= A && v16 <= vlim;          // (all but the first and the last flush of a stream)
            if (whole) {
                SIMT_BEGIN(t, W)
#pragma unroll 1
                    for (uint32_t c = (uint32_t)t; c < nchunk; c += (uint32_t)W) lzr::store_16(ob + 16u * c, *(const u128*)(l->win + w0 + 16u * c));
                SIMT_END_WAVE
            } else {
                SIMT_BEGIN(t, W)
#pragma unroll 1
                    for (uint32_t c = (uint32_t)t; c < nchunk; c += (uint32_t)W) {
                        const P cv = f0 + (P)(16u * c);
                        if (cv >= A && cv + 16u <= vlim) {
                            lzr::store_16(ob + 16u * c, *(const u128*)(l->win + w0 + 16u * c));
                        } else {   // the first chunk of an unaligned output, the chunk the limit cuts
                            for (uint32_t e = 0; e < 16u; e++)
                                if (cv + e >= A && cv + e < vlim) ob[16u * c + e] = l->win[w0 + 16u * c + e];
                        }
                    }
                SIMT_END_WAVE
            }
            fv = v16;
        }
        if (final && vend > fv) {
            const uint32_t ntail = (uint32_t)(vend - fv);   // < 16
            const P f0 = fv;
            const uint32_t w0 = (uint32_t)(f0 - vbase);
            SIMT_BEGIN(t, W)
                if ((uint32_t)t < ntail && f0 + (uint32_t)t >= A && f0 + (uint32_t)t < vlim) obase[f0 + (uint32_t)t] = l->win[w0 + (uint32_t)t];
            SIMT_END_WAVE
        }
    }

    // ---- slide: make room behind the window index of output position rpos: flush what is finished, keep the last kKeep
    // bytes, move them to the front.  Returns the new write index.
    SWC_D uint32_t slide(P rpos) {
        using simt::PT;
        const P vcur = (P)A + rpos;
        flush(vcur, false);
        const P nb = (vcur - kKeep) & ~(P)15;                   // (the caller slides only when vcur - vbase > kKeep + 16)
        const uint32_t D = (uint32_t)(nb - vbase);
        const uint32_t nmove = ((uint32_t)(vcur - nb) + 15u) >> 4;   // chunks that stay
#pragma unroll 1
        for (uint32_t c0 = 0; c0 < nmove; c0 += (uint32_t)W) {
            PT<u128, W> v;
            SIMT_BEGIN(t, W)
                v[t] = c0 + (uint32_t)t < nmove ? *(const u128*)(l->win + D + 16u * (c0 + (uint32_t)t)) : u128{0, 0, 0, 0};
            SIMT_END_WAVE
            SIMT_BEGIN(t, W)
                if (c0 + (uint32_t)t < nmove) *(u128*)(l->win + 16u * (c0 + (uint32_t)t)) = v[t];
            SIMT_END_WAVE
        }
        vbase = nb;
        return (uint32_t)(vcur - nb);
    }

    // ---- all lanes copy n literal bytes from the literal stream in HBM (offset lo) to window index wd
    SWC_D void coop_literals(uint32_t wd, P lo, uint32_t n) {
        gcptr src = lits + lo;
        const uint32_t room = INP ? ((uint32_t)lo < nin ? nin - (uint32_t)lo : 0u) : 0xFFFFFFFFu;   // (INP: a run may end with the block)
        SIMT_BEGIN(t, W)
            for (uint32_t o = 4u * (uint32_t)t; o < n; o += 4u * (uint32_t)W) {
                // (whole dwords are read: the literal stream's allocation ends 32 bytes behind its last byte)
                uint32_t w;
                if (!INP || o + 4u <= room) w = load_u32(src + o);
                else { w = 0; for (uint32_t q = 0; o + q < room && q < 4u; q++) w |= (uint32_t)src[o + q] << (8u * q); }
                if (o + 4u <= n) st32(l->win + wd + o, w);
                else st_tail(l->win + wd + o, w, n - o);
            }
        SIMT_END_WAVE
        rare_path_done();
    }

    // ---- all lanes copy one match of any kind: `wm` window index of its first byte, `n` bytes, `dist` back; `gsrc`: HBM
    // address of the source's first byte (read where the source lies in front of the window)
    SWC_D void coop_match(uint32_t wm, uint32_t n, uint32_t dist, gcptr gsrc) {
        const int32_t ws = (int32_t)wm - (int32_t)dist;
        if (n < 4u) {
            // (a piece of a split match, or a far one of three bytes)
            SIMT_BEGIN(t, W)
                if ((uint32_t)t < n) l->win[wm + (uint32_t)t] = ws >= 0 ? l->win[ws + (int32_t)mod_small((uint32_t)t, dist)] : gsrc[t];
            SIMT_END_WAVE
        } else if (dist >= n || dist >= 4u * (uint32_t)W) {
            // every step's source was finished by earlier steps or earlier records
            for (uint32_t o0 = 0; o0 < n; o0 += 4u * (uint32_t)W) {
                SIMT_BEGIN(t, W)
                    const uint32_t o = o0 + 4u * (uint32_t)t;
                    if (o < n) {
                        const uint32_t oo = o + 4u <= n ? o : n - 4u;          // the last dword ends on the last byte
                        const int32_t si = ws + (int32_t)oo;
                        const uint32_t w = si >= 0 ? ld32(l->win + si) : load_u32(gsrc + oo);
                        st32(l->win + wm + oo, w);
                    }
                SIMT_END_WAVE
            }
        } else {
            // distance < length: byte k of the match is byte k % distance of its first period, which is final
            // (the period is inside the window: distance < 256)
            const uint32_t step = mod_small((uint32_t)W, dist);
            simt::PT<uint32_t, W> ph;
            SIMT_BEGIN(t, W) ph[t] = mod_small((uint32_t)t, dist); SIMT_END
            for (uint32_t o0 = 0; o0 < n; o0 += (uint32_t)W) {
                SIMT_BEGIN(t, W)
                    const uint32_t k = o0 + (uint32_t)t;
                    if (k < n) l->win[wm + k] = l->win[(uint32_t)ws + ph[t]];
                    uint32_t p = ph[t] + step;
                    if (p >= dist) p -= dist;
                    ph[t] = p;
                SIMT_END_WAVE
            }
        }
        if (ws < 0) rare_path_done();
    }

    // ---- a lane copies its own match of up to kLongLen bytes from window index s (or, far, from the pieces fw) to m:
    // eight-byte pieces, the last one shifted back to end on the last byte; four to seven bytes as two dwords
    SWC_D static void copy_own(uint8_t* B, uint32_t m, uint32_t s, uint32_t le, bool far, uint64_t f0, uint64_t f1, uint64_t f2, uint64_t f3) {
        if (le >= 8u) {
            const uint32_t last = le - 8u;
            st64(B + m, far ? f0 : rd64u(B, s));
            if (le > 8u) {
                const uint32_t o1 = 8u < last ? 8u : last;
                st64(B + m + o1, far ? f1 : rd64u(B, s + o1));
                if (le > 16u) {
                    const uint32_t o2 = 16u < last ? 16u : last;
                    st64(B + m + o2, far ? f2 : rd64u(B, s + o2));
                    if (le > 24u) st64(B + m + last, far ? f3 : rd64u(B, s + last));
                }
            }
        } else {
            const uint64_t v0 = far ? f0 : rd64u(B, s);
            if (le >= 4u) {
                st32(B + m, (uint32_t)v0);
                st32(B + m + (le & 3u), (uint32_t)(v0 >> (8u * (le & 3u))));
            } else {
                st_tail(B + m, (uint32_t)v0, le);
            }
        }
    }
    // ---- a lane copies its own literal run of up to kLongLit bytes from the pieces front() loaded (piece k: bytes
    // [min(8 k, n - 8), + 8) of the run; a run of fewer than eight bytes: its bytes at the bottom of piece 0) to window index d
    SWC_D static void copy_own_lits(uint8_t* B, uint32_t d, uint32_t n, const uint64_t* pc) {
        if (n >= 8u) {
            const uint32_t last = n - 8u;
            st64(B + d, pc[0]);
#pragma unroll
            for (int k = 1; k < kLitPieces; k++) {
                if (n > 8u * (uint32_t)k) st64(B + d + (8u * (uint32_t)k < last ? 8u * (uint32_t)k : last), pc[k]);
            }
        } else if (n >= 4u) {
            st32(B + d, (uint32_t)pc[0]);
            st32(B + d + (n & 3u), (uint32_t)(pc[0] >> (8u * (n & 3u))));
        } else {
            st_tail(B + d, (uint32_t)pc[0], n);
        }
    }

    // ---- a group of records between front() (records -> places, loads asked for) and back() (the copies)
    struct Group {
        simt::PT<uint32_t, W> rec;            // the records
        simt::PT<uint32_t, W> x;              // inclusive sums over the lanes: (literals + length) | literals << 16 (a big run: literal bytes)
        simt::PT<uint64_t, W> fw[kPieces];    // far sources asked for ahead, eight bytes each
        simt::PT<uint64_t, W> lw[kLitPieces]; // my literal run (up to kLongLit bytes) from the literal stream, eight bytes each
        simt::PT<uint32_t, W> loff;           // R8: where my literal run starts (offset from `lits`)
        uint32_t kind;                        // 0: records up to the first big literal-only one; 1: a run of big literal-only records
        uint32_t ntake, span, litspan;        // records, output bytes, literal bytes (R4: sequence bytes) of the group
        P S0;                                 // offset of the group's first literal from `lits` (R4: the start of its first sequence)
        P pf_vbase;                           // window base the far loads assumed (~0: none were wanted)
        int32_t pf_lim;                       // ... and how far (relative to the group's first byte) a source they asked for may reach
    };
    // how far -- relative to output position rpos -- a far source may reach that is read now
    SWC_D int32_t far_limit(P rpos) const {
        const SP d = (SP)(landed - ((P)A + rpos)) - 128;       // (landed never lies behind the position: d < 0)
        return d < -(SP)0x40000000 ? -(int32_t)0x40000000 : d > 0 ? 0 : (int32_t)d;
    }

    // ---- front: the records of a group (`left` records remain from it on) that starts at output position `rpos`, literal
    // offset `lbase`; `vb_pred`: the window base when the group in front of it has been copied (~0: unknown, no far loads
    // ahead).
    // EVERY call issues the same loads, wanted or not (a lane that wants nothing reads a place that certainly exists): a
    // register that is loaded on one path only reaches the next iteration through a copy, and the compiler waits for the
    // load in front of the copy -- at once, instead of an iteration later.
    SWC_D void front(Group& g, const simt::PT<uint32_t, W>& r_in, const simt::PT<uint32_t, W>& o_in, uint32_t left, P rpos, P lbase, P vb_pred, uint32_t base_rec) {
        using simt::PT;
        PT<bool, W> big;
        if (R4) {   // a group that starts at an anchor takes its S from it; no group reaches across the next one
            if (anc_rec == base_rec) { lbase = (P)anc_S; next_anchor(); }
            const uint32_t room = anc_rec - base_rec;     // (>= 1: the anchors' record indices grow)
            if (room < left) left = room;
        }
        g.S0 = lbase;
        SIMT_BEGIN(t, W)
            uint32_t li, le, di;
            g.rec[t] = (uint32_t)t < left ? r_in[t] : 0u;   // (the lanes past the last record loaded it again)
            g.loff[t] = R8 ? o_in[t] : 0u;
            unpack(g.rec[t], li, le, di);
            big[t] = le == 0u && li >= kBigLit;
        SIMT_END
        const uint64_t bigmask = simt::wave_ballot<W>(big);
        const uint32_t nb = bigmask ? (uint32_t)simt::ctz64(bigmask) : 64u;
        if (nb == 0u) {
            // the group starts with big literal-only records: the run of them is a group of its own
            uint32_t nrun = ~bigmask ? (uint32_t)simt::ctz64(~bigmask) : 64u;
            if (nrun > left) nrun = left;
            SIMT_BEGIN(t, W)
                uint32_t li, le, di;
                unpack(g.rec[t], li, le, di);
                g.x[t] = (uint32_t)t < nrun ? li : 0u;
            SIMT_END
            simt::wave_scan_incl<W>(g.x);
            g.kind = 1;
            g.ntake = nrun;
            g.span = g.litspan = simt::wave_read<W>(g.x, (int)nrun - 1);
        } else {
            SIMT_BEGIN(t, W)
                uint32_t li, le, di;
                unpack(g.rec[t], li, le, di);
                g.x[t] = (uint32_t)t < nb ? (li + le) | (lit_step(li, le) << 16) : 0u;
            SIMT_END
            simt::wave_scan_incl<W>(g.x);
            PT<bool, W> tk;
            SIMT_BEGIN(t, W)
                tk[t] = (uint32_t)t < left && (uint32_t)t < nb && (g.x[t] & 0xFFFFu) <= kSpanMax;
            SIMT_END
            const uint64_t tm = simt::wave_ballot<W>(tk);
            const uint32_t ntake = ~tm ? (uint32_t)simt::ctz64(~tm) : 64u;     // >= 1: any single record that is not big fits
            const uint32_t xl = simt::wave_read<W>(g.x, (int)ntake - 1);
            g.kind = 0;
            g.ntake = ntake;
            g.span = xl & 0xFFFFu;
            g.litspan = xl >> 16;
        }
        if (R4) {   // where my literals lie in the block: the start of my sequence by the running sum, the token and the length byte behind it
            const bool runs = g.kind == 0u;
            SIMT_BEGIN(t, W)
                uint32_t li, le, di;
                unpack(g.rec[t], li, le, di);
                const uint32_t o = (uint32_t)lbase + (runs ? (g.x[t] >> 16) - seq_bytes(li, le) + lit_skip(li, le) : g.x[t] - li);
                g.loff[t] = o < nin ? o : nin;   // (never beyond the block, whatever the records say: the guarded loads below then stay inside it)
            SIMT_END
        }
        // ---- my literal run, straight from the literal stream in HBM into registers: 32-bit offsets from the group's first literal
        {
            gcptr lb = INP ? lits : lits + lbase;
            const bool runs = g.kind == 0u;
            const uint32_t ntake = g.ntake;
            // R8: the last runs of a block end with the block -- an eight-byte piece that would reach past it is read further in
            // front and shifted down (the block's last groups only: the offsets grow with the records)
            const bool guard = INP && runs && nin >= 8u && simt::wave_read<W>(g.loff, (int)ntake - 1) + (lzr::kLitRunMax + 8u) > nin;
            if (INP && nin < 8u) {   // (a block of a few bytes that still claims output, i.e. one that ends in an error: back() copies its runs byte by byte)
                SIMT_BEGIN(t, W)
#pragma unroll
                    for (int k = 0; k < kLitPieces; k++) g.lw[k][t] = 0;
                SIMT_END
            } else if (guard) {
                const uint32_t top = nin - 8u;
                SIMT_BEGIN(t, W)
                    uint32_t li, le, di;
                    unpack(g.rec[t], li, le, di);
                    const bool ask = (uint32_t)t < ntake && li != 0u && li <= kLongLit;
                    const uint32_t o0 = ask ? g.loff[t] : 0u;
                    const uint32_t last = (li > 8u ? li : 8u) - 8u;
#pragma unroll
                    for (int k = 0; k < kLitPieces; k++) {
                        const uint32_t ok = 8u * (uint32_t)k < last ? 8u * (uint32_t)k : last;
                        const uint32_t a = ask && 8u * (uint32_t)k < li ? o0 + ok : 0u;
                        const uint32_t a1 = a < top ? a : top;
                        g.lw[k][t] = load_u64(lb + a1) >> (8u * (a - a1));   // (a - a1 < 8: a literal byte lies inside the block)
                    }
                SIMT_END
            } else {
            SIMT_BEGIN(t, W)
                uint32_t li, le, di;
                unpack(g.rec[t], li, le, di);
                const bool ask = !SWC_LZC_FAKE && runs && (uint32_t)t < ntake && li != 0u && li <= kLongLit;
                // (nothing wanted: the group's first literal, or the byte behind the last: it exists; R8: the block's first bytes)
                const uint32_t o0 = ask ? (INP ? g.loff[t] : (g.x[t] >> 16) - li) : 0u;
                const uint32_t last = (li > 8u ? li : 8u) - 8u;
#pragma unroll
                for (int k = 0; k < kLitPieces; k++) {
                    const uint32_t ok = 8u * (uint32_t)k < last ? 8u * (uint32_t)k : last;
                    g.lw[k][t] = load_u64(lb + (ask && 8u * (uint32_t)k < li ? o0 + ok : 0u));
                }
            SIMT_END
            }
        }
        // ---- far sources, one group ahead: 32-bit offsets from (the group's first byte - kBack)
        {
            const bool known = g.kind == 0u && vb_pred != ~(P)0;
            P vp = known ? vb_pred : 0u;
            const P vcur = (P)A + rpos;
            if (known && (uint32_t)(vcur - vp) + g.span > WIN) vp = (vcur - kKeep) & ~(P)15;   // back() will slide (the same arithmetic)
            const uint32_t wpn = (uint32_t)(vcur - vp);
            g.pf_vbase = known ? vp : ~(P)0;
            g.pf_lim = far_limit(rpos);
            const int32_t lim = g.pf_lim;
            const uint32_t ntake = g.ntake;
            gcptr fb = (gcptr)out + ((int64_t)rpos - (int64_t)kBack);
            const uint32_t idle = kBack - (rpos < kBack ? (uint32_t)rpos : kBack);      // a place that exists: the output's first bytes, or kBack bytes back
            SIMT_BEGIN(t, W)
                uint32_t li, le, di;
                unpack(g.rec[t], li, le, di);
                const int32_t srel = (int32_t)((g.x[t] & 0xFFFFu) - le - di);           // my source starts here, relative to the group's first byte
                const bool ask = !(SWC_LZC_FAKE & 2) && known && (uint32_t)t < ntake && le != 0u && (int32_t)wpn + srel < 0 && le <= kLongLen && di >= le && srel + (int32_t)le <= lim;
                const uint32_t o0 = kBack + (uint32_t)srel;
                const uint32_t last = (le > 8u ? le : 8u) - 8u;
#pragma unroll
                for (int k = 0; k < kPieces; k++) {
                    const uint32_t ok = 8u * (uint32_t)k < last ? 8u * (uint32_t)k : last;
                    g.fw[k][t] = load_u64(fb + (ask && 8u * (uint32_t)k < le ? o0 + ok : idle));
                }
            SIMT_END
        }
    }

    // ---- back: the copies of a group
    SWC_D void back(Group& g, P rpos, P lbase) {
        using simt::PT;
        if (g.kind == 1u) {
            // a run of big literal-only records: all lanes copy them, one record after the other
            P rp = rpos, lb = lbase;
            for (uint32_t i = 0; i < g.ntake; i++) {
                const uint32_t e1 = simt::wave_read<W>(g.x, (int)i);
                if (INP) lb = (P)simt::wave_read<W>(g.loff, (int)i);
                uint32_t n = (uint32_t)(rpos + e1 - rp);
                while (n != 0u) {   // (a record of up to kMaxLitOnly bytes in pieces of what a slide makes room for)
                    const uint32_t piece = n < kSpanMax ? n : kSpanMax;
                    uint32_t wp = (uint32_t)((P)A + rp - vbase);
                    if (wp + piece > WIN) wp = slide(rp);
                    coop_literals(wp, lb, piece);
                    rp += piece;
                    lb += piece;
                    n -= piece;
                }
            }
            return;
        }
        const uint32_t ntake = g.ntake;
        uint32_t wp = (uint32_t)((P)A + rpos - vbase);
        if (wp + g.span > WIN) wp = slide(rpos);
#if SWC_LZC_CUT == 1   // (instruction accounting builds, tools/attic/exp_copier_counts.sh: the output is wrong)
        return;
#endif
        uint8_t* const B = (uint8_t*)l;                                 // window indices are offsets from here
        const bool pfu = g.pf_vbase == vbase;                           // the far loads assumed the window base that came to be
        const int32_t pf_lim = g.pf_lim;
        PT<uint32_t, W> lit, len, dist, wm, flags, si;   // si: a signed value
        PT<bool, W> longlit, pend;
        // ---- my literals; my match, if its source was final when the group began
        SIMT_BEGIN(t, W)
            const bool mine = (uint32_t)t < ntake;
            uint32_t li, le, di;
            unpack(g.rec[t], li, le, di);
            lit[t] = li; len[t] = le; dist[t] = di;
            const uint32_t end = g.x[t] & 0xFFFFu;
            const uint32_t wd = wp + end - li - le;                   // window index of my literals
            const uint32_t m = wd + li;                               // ... of my match
            const uint32_t s = m - di;                                // ... of its source (negative: in front of the window)
            wm[t] = m;
            si[t] = s;
            // my literals (stores under the exec mask: the LDS takes as long over a store as lanes take part in it, and most
            // lanes have no literals at all)
            const bool own_lits = li <= kLongLit && !(INP && nin < 8u);
            longlit[t] = mine && li != 0u && !own_lits;
            if (mine && li != 0u && own_lits) {
                uint64_t pc[kLitPieces];
#pragma unroll
                for (int k = 0; k < kLitPieces; k++) pc[k] = g.lw[k][t];
                copy_own_lits(B, wd, li, pc);
            }
            const bool far = (int32_t)s < 0;
            const bool coop = le > kLongLen || di < le;
            flags[t] = (coop ? (uint32_t)kFlagCoop : 0u) | (far ? (uint32_t)kFlagFar : 0u);
            const int32_t srel = (int32_t)(end - le - di);
            const bool pf = far && pfu && srel + (int32_t)le <= pf_lim;            // front() asked for it
            const bool old = !far && s + le <= wp;                                  // it ends in front of the group
            const bool act = mine && le != 0u && !coop && (pf || old);
            pend[t] = mine && le != 0u && !act;
            // my match, if its source was final when the group began
            if (act) copy_own(B, m, s, le, far, g.fw[0][t], g.fw[kPieces > 1 ? 1 : 0][t], g.fw[kPieces > 2 ? 2 : 0][t], g.fw[kPieces > 3 ? 3 : 0][t]);
        SIMT_END_WAVE
#if SWC_LZC_CUT == 2
        return;
#endif
        // ---- long literal runs by all lanes
        for (uint64_t m = simt::wave_ballot<W>(longlit); m; m &= m - 1u) {
            const int h = simt::ctz64(m);
            const uint32_t li = simt::wave_read<W>(lit, h);
            const uint32_t lend = simt::wave_read<W>(g.x, h) >> 16;
            coop_literals(simt::wave_read<W>(wm, h) - li, INP ? (P)simt::wave_read<W>(g.loff, h) : lbase + (lend - li), li);
        }
#if SWC_LZC_CUT == 3
        return;
#endif
        // ---- the matches that are left reach into their own group (or are long, overlapping, far and not asked for: "odd").
        PT<bool, W> odd;
        SIMT_BEGIN(t, W) odd[t] = pend[t] && flags[t] != 0u; SIMT_END
        uint64_t pm = simt::wave_ballot<W>(pend);
        const uint64_t om = simt::wave_ballot<W>(odd);
#ifndef SWC_LZC_BYTELANES
#define SWC_LZC_BYTELANES 256
#endif
        if (SWC_LZC_BYTELANES != 0 && om == 0ull && pm != 0ull) {
            // ---- Nothing odd among them (nine groups in ten on text): the BYTES of the matches that are left, one per lane.  A
            // running sum of their lengths numbers the bytes; a lane finds the match its byte belongs to through a marker at
            // the match's first byte (LDS) and a running maximum over the lanes, fetches window index and source with two
            // cross-lane reads, and then all lanes copy their byte TOGETHER, again and again, until a round reads what the round
            // before it read: sources lie strictly in front of their bytes and everything else in the window is final, so that
            // state is the one and only fixed point -- the sequence the reference's append loop would have produced
            // (Deflate.swift:216-232).  A round is a byte read, a byte write and a ballot; a match that does not depend on another
            // one of them (five of six) is right after the first, the loop ends one round after the deepest chain.  64 bytes a pass.
            PT<uint32_t, W> incl, offs, pk;
            SIMT_BEGIN(t, W) incl[t] = pend[t] ? len[t] : 0u; SIMT_END
            simt::wave_scan_incl<W>(incl);
            const uint32_t nbytes = simt::wave_read<W>(incl, W - 1);
            if (nbytes <= (uint32_t)SWC_LZC_BYTELANES) {
                SIMT_BEGIN(t, W)
                    offs[t] = incl[t] - (pend[t] ? len[t] : 0u);
                    pk[t] = wm[t] | (si[t] << 16);          // (both are window indices below 64 KiB: nothing here is far)
                SIMT_END
                for (uint32_t c0 = 0; c0 < nbytes; c0 += (uint32_t)W) {
                    PT<uint32_t, W> mk, ga, gb, dsti, srci, prev, val;
                    PT<bool, W> act, chg;
                    SIMT_BEGIN(t, W) l->pmap[t] = 0u; SIMT_END_WAVE
                    SIMT_BEGIN(t, W)
                        if (pend[t] && offs[t] < c0 + (uint32_t)W && offs[t] + len[t] > c0) l->pmap[(offs[t] > c0 ? offs[t] : c0) - c0] = (uint32_t)t + 1u;
                    SIMT_END_WAVE
                    SIMT_BEGIN(t, W) mk[t] = l->pmap[t]; SIMT_END
                    simt::wave_scan_max_incl<W>(mk);
                    SIMT_BEGIN(t, W) mk[t] = mk[t] ? mk[t] - 1u : 0u; SIMT_END
                    simt::wave_gather<W>(ga, pk, mk);
                    simt::wave_gather<W>(gb, offs, mk);
                    SIMT_BEGIN(t, W)
                        const uint32_t j = c0 + (uint32_t)t, k = j - gb[t];
                        act[t] = j < nbytes;
                        dsti[t] = (ga[t] & 0xFFFFu) + k;
                        srci[t] = (ga[t] >> 16) + k;
                        prev[t] = 0x100u;
                    SIMT_END
                    for (;;) {
                        SIMT_BEGIN(t, W) val[t] = act[t] ? (uint32_t)B[srci[t]] : 0x100u; SIMT_END_WAVE
                        SIMT_BEGIN(t, W)
                            if (act[t]) B[dsti[t]] = (uint8_t)val[t];
                            chg[t] = val[t] != prev[t];
                            prev[t] = val[t];
                        SIMT_END_WAVE
                        if (simt::wave_ballot<W>(chg) == 0ull) break;
                    }
                }
                return;
            }
        }
        if ((uint32_t)simt::popc64(pm & ~om) > kSeqMax) {
            // Many short ones: most of them do NOT depend on each other -- the source of a match that reaches into its group
            // usually is made of literals and of matches that are final by now.  A bitmap of the group's output marks the bytes
            // of the matches still to be copied; a match whose source touches none of them (and lies in front of the first odd
            // one) is copied by its own lane at once, all of those together.  What is left goes in record order below.
            const uint32_t odd_m = om ? simt::wave_read<W>(wm, simt::ctz64(om)) : 0xFFFFFFFFu;
            const uint64_t pm0 = pm;
            SIMT_BEGIN(t, W) l->pmap[t] = 0u; if (t < 2) l->pmap[64 + t] = 0u; SIMT_END_WAVE
            SIMT_BEGIN(t, W)
                if (((pm0 >> t) & 1u) != 0u && flags[t] == 0u) {
                    const uint32_t a = wm[t] - wp;
                    const uint64_t mk = ((1ull << len[t]) - 1ull) << (a & 31u);       // (len <= kLongLen = 32)
                    lds_or(&l->pmap[a >> 5], (uint32_t)mk);
                    if ((uint32_t)(mk >> 32) != 0u) lds_or(&l->pmap[(a >> 5) + 1u], (uint32_t)(mk >> 32));
                }
            SIMT_END_WAVE
            PT<bool, W> free_;
            SIMT_BEGIN(t, W)
                bool f = ((pm0 >> t) & 1u) != 0u && flags[t] == 0u && si[t] + len[t] <= odd_m;
                if (f && si[t] + len[t] > wp) {                          // the part of my source inside the group
                    const uint32_t a = (si[t] > wp ? si[t] : wp) - wp, e = si[t] + len[t] - wp;
                    const uint64_t mk = ((1ull << (e - a)) - 1ull) << (a & 31u);
                    const uint64_t w = (uint64_t)l->pmap[a >> 5] | ((uint64_t)l->pmap[(a >> 5) + 1u] << 32);
                    f = (w & mk) == 0ull;
                }
                free_[t] = f;
            SIMT_END
            const uint64_t fm = simt::wave_ballot<W>(free_);
            SIMT_BEGIN(t, W)
                if (free_[t]) copy_own(B, wm[t], si[t], len[t], false, 0, 0, 0, 0);
            SIMT_END_WAVE
            pm &= ~fm;
        }
        // ---- in record order: the first record that has not been copied finds all bytes in front of it final
        if (om == 0ull) {
            // a few short matches (source in the window, not overlapping): one after the other, a byte per lane
            while (pm) {
                const int h = simt::ctz64(pm);
                pm &= pm - 1u;
                const uint32_t hm = simt::wave_read<W>(wm, h), hs = simt::wave_read<W>(si, h), hn = simt::wave_read<W>(len, h);
                SIMT_BEGIN(t, W)
                    if ((uint32_t)t < hn) B[hm + (uint32_t)t] = B[hs + (uint32_t)t];
                SIMT_END_WAVE
            }
            return;
        }
        while (pm) {
            const int h = simt::ctz64(pm);
            const uint32_t hm = simt::wave_read<W>(wm, h);
            if (!((om >> h) & 1u)) {
                const uint32_t hs = simt::wave_read<W>(si, h), hn = simt::wave_read<W>(len, h);
                SIMT_BEGIN(t, W)
                    if ((uint32_t)t < hn) B[hm + (uint32_t)t] = B[hs + (uint32_t)t];
                SIMT_END_WAVE
                pm &= pm - 1u;
                continue;
            }
            pm &= pm - 1u;
            const uint32_t hs = simt::wave_read<W>(si, h), hn = simt::wave_read<W>(len, h);
            const uint32_t fl = simt::wave_read<W>(flags, h);
            const int64_t sp = (int64_t)rpos + ((int64_t)(int32_t)hs - (int64_t)wp);      // output position of the source: >= 0, phase 1 rejects a distance beyond the output
            if ((fl & (uint32_t)kFlagFar) && (int32_t)((int32_t)hs - (int32_t)wp) + (int32_t)hn > far_limit(rpos)) drain();
            coop_match(hm, hn, simt::wave_read<W>(dist, h), (gcptr)out + (sp >= 0 ? sp : 0));
        }
    }

    // a stream of a few bytes: one lane builds it in the window, byte by byte (the prefetches of run() read eight bytes that exist)
    SWC_D void tiny(const SWC_AS_GLOBAL uint32_t* recs, uint32_t nrec) {
        const uint32_t lim = (uint32_t)limit;   // < 64
        SIMT_BEGIN(t, W)
            if (t == 0) {
                uint32_t pos = 0, ai = 0;
                uint64_t lp = 0, S = 0;
                for (uint32_t i = 0; i < nrec && pos < lim; i++) {
                    uint32_t li, le, di;
                    unpack(recs[R8 ? 2u * i : i], li, le, di);
                    if (R8) lp = recs[2u * i + 1u];
                    if (R4) {   // (the rule of front(), one record at a time)
                        if (ai < nanc && anc[2u * ai] == i) { S = anc[2u * ai + 1u]; ai++; }
                        lp = S + lit_skip(li, le);
                        S += seq_bytes(li, le);
                    }
                    for (uint32_t k = 0; k < li && pos < lim; k++, pos++, lp++) l->win[pos] = lits[lp];
                    for (uint32_t k = 0; k < le && pos < lim; k++, pos++) l->win[pos] = pos >= di ? l->win[pos - di] : (uint8_t)0;
                }
            }
        SIMT_END_WAVE
        SIMT_BEGIN(t, W)
            if ((uint32_t)t < lim) out[t] = l->win[t];
        SIMT_END_WAVE
    }

    SWC_D void run(const SWC_AS_GLOBAL uint32_t* recs, uint32_t nrec) {
        using simt::PT;
        A = (uint32_t)(uintptr_t)out & 15u;
        vbase = 0;
        fv = 0;
        landed = 0;
        if (nrec == 0) return;
        if (limit < 64u) { tiny(recs, nrec); return; }
        P rpos = 0;                      // output bytes finished by earlier groups
        uint32_t base = 0;               // first record of the group
        PT<uint32_t, W> r_nx, o_nx;      // the records of the group after `nxt` (R8: and their literal offsets), on their way
        Group cur, nxt;
        constexpr uint32_t kRecBytes = R8 ? 8u : 4u;
        const uint32_t rlast4 = kRecBytes * (nrec - 1u);
        gcptr rb = (gcptr)recs;
        // the 64 records from record `first` on (the lanes past the last record load it again)
        auto load_records = [&](uint32_t first) {
            const uint32_t b4 = kRecBytes * first;
            SIMT_BEGIN(t, W)
                const uint32_t o = b4 + kRecBytes * (uint32_t)t < rlast4 ? b4 + kRecBytes * (uint32_t)t : rlast4;
                if (R8) { const uint64_t v = load_u64(rb + o); r_nx[t] = (uint32_t)v; o_nx[t] = (uint32_t)(v >> 32); }
                else { r_nx[t] = load_u32(rb + o); o_nx[t] = 0u; }
            SIMT_END
        };
        load_records(0);
        if (R4) { anc_i = 0; next_anchor(); }
        front(nxt, r_nx, o_nx, nrec, 0, 0, ~(P)0, 0u);
        load_records(nxt.ntake);
        bool more = true;
        while (more) {
            // Everything asked for during the last iteration -- the next records, far sources, literals -- is here, and its
            // stores have arrived (an iteration's loads have the whole of the previous group's copies to come back).
            drain();
#if SWC_LZC_FLUSH_EARLY
            // finished bytes leave for HBM a KiB at a time (whole-wave stores), long before they leave the window -- and at the TOP of
            // the iteration: the drain of the next one then finds stores that have had a whole group's copies to arrive, not stores
            // issued a moment ago (a wave spent two fifths of its time waiting, most of it there)
            if ((P)A + rpos - fv >= 1024u) flush((P)A + rpos, false);
#endif
            cur = nxt;
            const uint32_t nbase = base + cur.ntake;
            more = nbase < nrec;
            if (more) {
                P vbp = ~(P)0;
                if (cur.kind == 0u) {   // the window base when `cur` has been copied (slide()'s arithmetic)
                    vbp = vbase;
                    const P vcur = (P)A + rpos;
                    if ((uint32_t)(vcur - vbp) + cur.span > WIN) vbp = (vcur - kKeep) & ~(P)15;
                }
                // (the literal offset of the next group: the literal bytes consumed so far; R4: where the running sum stands)
                front(nxt, r_nx, o_nx, nrec - nbase, rpos + cur.span, cur.S0 + cur.litspan, vbp, nbase);
                load_records(nbase + nxt.ntake);
            }
            back(cur, rpos, cur.S0);
            base = nbase;
            rpos += cur.span;
#if !SWC_LZC_FLUSH_EARLY
            // finished bytes leave for HBM a KiB at a time (whole-wave stores), long before they leave the window
            if ((P)A + rpos - fv >= 1024u) flush((P)A + rpos, false);
#endif
        }
        drain();
        flush((P)A + rpos, true);
    }
};

// The window is a trade between the waves a CU holds and the matches that are FAR; the history a slide keeps, between far
// matches and the LDS traffic of the slides (kernels.hip has the measurements).
template <uint32_t WINB, uint32_t SPANB, uint32_t KEEPB, int LITP>
struct Cfg {
    static constexpr uint32_t kWin = WINB, kSpan = SPANB, kKeep = KEEPB;
    static constexpr int kLitPieces = LITP;
};

// The configurations the library ships (kernels.hip: the waves per CU each is launched with; profiles/r05_experiments.txt and
// r06_experiments.txt: the sweeps).  Deflate: 6 KiB window of which a slide keeps 3.25 KiB (a slide costs more than the far
// matches it avoids: a far source is four loads that are issued anyway), groups of up to 1 KiB, literal runs of up to 16 bytes
// per lane -- 6,424 bytes of LDS per wave, 24 waves per CU.  LZ4 (offsets up to 65,535, 4 MiB blocks, longer
// literal runs): 7 KiB / 3.25 KiB kept / 1 KiB groups / runs of up to 32 bytes -- 7,448 bytes and 81 VGPRs, 20 waves per CU.
#ifndef SWC_LZC_WIN
#define SWC_LZC_WIN 6144
#define SWC_LZC_SPAN 1024
#define SWC_LZC_KEEP 3328
#endif
#ifndef SWC_LZC_LITP
#define SWC_LZC_LITP 2
#endif
using CfgDeflate = Cfg<SWC_LZC_WIN, SWC_LZC_SPAN, SWC_LZC_KEEP, SWC_LZC_LITP>;
#ifndef SWC_LZC4_WIN
#define SWC_LZC4_WIN 7168
#define SWC_LZC4_SPAN 1024
#define SWC_LZC4_KEEP 3328
#endif
#ifndef SWC_LZC4_LITP
#define SWC_LZC4_LITP 4
#endif
using CfgLz4 = Cfg<SWC_LZC4_WIN, SWC_LZC4_SPAN, SWC_LZC4_KEEP, SWC_LZC4_LITP>;
using CfgWide = Cfg<16384, 2048, 14320, 4>;   // (comparison runs: 8 waves per CU)

// One job: `ws` is the stream's workspace area of `area` bytes written by phase 1.  RM 1 (R8) / 2 (R4), LZ4: the literals are
// fetched from the job's input; R8: the area holds the header and eight-byte records; R4: four-byte records, and the anchors where
// the literal stream would be (their number in the header's pad0).
template <typename CFG, int RM = 0>
SWC_D void copy_job(const Job& job, const uint8_t* ws, size_t area, Lds<CFG::kWin>* lds) {
    const SWC_AS_GLOBAL lzr::StreamHeader* h = (const SWC_AS_GLOBAL lzr::StreamHeader*)ws;
    const size_t lo = lzr::lit_offset(area, job.out_cap);
    if (lo == 0) return;   // no literal stream: phase 1 reported SWC_E_NEED_WORKSPACE for this job
    const uint64_t limit = job.out_len < job.out_cap ? job.out_len : job.out_cap;
    const SWC_AS_GLOBAL uint32_t* recs = (const SWC_AS_GLOBAL uint32_t*)(ws + sizeof(lzr::StreamHeader));
    gcptr lits = RM != 0 ? (gcptr)job.in : (gcptr)ws + lo;
    if (limit < 0xFFF00000ull) {   // (positions, watermarks and their differences in 32 bits)
        Copier<CFG, uint32_t, RM> cp;
        cp.l = lds;
        cp.out = (gptr)job.out;
        cp.lits = lits;
        cp.nin = RM != 0 ? (uint32_t)job.in_len : 0xFFFFFFFFu;   // (blocks are addressed with 32-bit offsets: the parse rejects larger ones)
        if (RM == 2) { cp.anc = (const SWC_AS_GLOBAL uint32_t*)(ws + lo); cp.nanc = h->pad0; }
        cp.limit = (uint32_t)limit;
        cp.run(recs, h->nrec);
    } else {
        Copier<CFG, uint64_t, RM> cp;
        cp.l = lds;
        cp.out = (gptr)job.out;
        cp.lits = lits;
        cp.nin = RM != 0 ? (uint32_t)job.in_len : 0xFFFFFFFFu;
        if (RM == 2) { cp.anc = (const SWC_AS_GLOBAL uint32_t*)(ws + lo); cp.nanc = h->pad0; }
        cp.limit = limit;
        cp.run(recs, h->nrec);
    }
}

}  // namespace lzc
}  // namespace swc
#endif
