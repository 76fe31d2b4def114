// lz_resolve.h -- phase 2 of the Deflate path: LZ77 match resolution, one stream per WORKGROUP, with the
// sliding window held in LDS.
//
// The reference executes every back-reference inline, one byte per `out.append`
// (Sources/Deflate/Deflate.swift:216-232).  On the MI355X the entropy decode and the copy are split:
//
//   phase 1 (inflate_lane.h, one stream per lane) decodes the Huffman symbols and appends every LITERAL to the
//           stream's dense literal stream and one 32-bit record per MATCH to its record list, both in the HBM
//           workspace.  It never touches the output buffer, so the tens of thousands of streams that must be in
//           flight to hide the decode latency neither keep 32 KiB windows alive in L2 / Infinity Cache nor
//           sprinkle single bytes over every sector of the output (measured before: HBM traffic 7-28x the
//           algorithmic bytes, profiles/r01_pmc_deflate_lane_per_stream.txt, r01c_deflate64k_traffic.json);
//   phase 2 (this file, one stream per workgroup of T threads) walks the record list in batches of T records:
//           a workgroup prefix-scan turns (literal run, length) into output positions and literal offsets, every
//           thread drops its literal run into the 64 KiB LDS window and then executes its match inside the window
//           (matches whose source is the output of an earlier match of the batch take over that match's source by
//           pointer jumping; the rest wait on a done-bitmap), and the finished span leaves with coalesced
//           16-byte stores -- the only time the output is written.
//
// Record format (u32):  lit_run[0..7] < 255:  lit_run | (length - 3)[8..15] | (distance - 1)[16..31]
//                       lit_run[0..7] = 255:  skip[8..31]  -- `skip` literal bytes; always handled on its own
// lit_run = literal bytes between the end of the previous record and this match (< 255; longer runs
// are preceded by skip records); length 3..258 (longer matches are split), distance 1..65536.  Records exist
// only for matches that START below the output capacity.  The same format and kernel serve LZ4 (lz4_wave.h).
//
// The same source compiles for the host with T = 1 (tests/host_emu): batches of one record.
#ifndef SWC_LZ_RESOLVE_H
#define SWC_LZ_RESOLVE_H

#include "swc_common.h"

namespace swc {
namespace lzr {

constexpr uint32_t kSpan = 16384;          // output bytes one batch may cover
constexpr uint32_t kLitBuf = 4096;         // literal bytes staged in LDS ahead of the batch being resolved
constexpr uint32_t kBucket = 32;           // granularity of the position -> record index of a batch
constexpr uint32_t kSkipMark = 255u;       // lit_run value of a skip record
constexpr uint32_t kMaxSkip = 0x00FFFFFFu;
SWC_HD uint32_t make_skip(uint32_t n) { return kSkipMark | (n << 8); }

// Per-stream area in the workspace: 16-byte header | records | literal stream (at the END of the area).
struct StreamHeader {
    uint32_t nrec;
    uint32_t pad0;
    uint64_t nlit;    // bytes in the literal stream
};
// Records a stream of capacity `cap` can need: one per match piece (>= 3 output bytes each, started below cap)
// plus one skip per >= 255 literal bytes, plus slack.
SWC_HD size_t max_records(uint64_t cap) { return (size_t)(cap / 3 + cap / 255 + 8); }
SWC_HD size_t lit_bytes(uint64_t cap) { return (size_t)((cap + 16 + 15) & ~(uint64_t)15); }   // +16: 8-byte flushes and reads may overshoot
SWC_HD size_t ws_bytes_per_job(uint64_t cap) { return ((sizeof(StreamHeader) + max_records(cap) * 4 + 15) & ~(size_t)15) + lit_bytes(cap); }
// a job's literal stream inside its area of `stride` bytes (0 if the area is too small for it)
SWC_HD size_t lit_offset(size_t stride, uint64_t cap) { return stride >= lit_bytes(cap) + sizeof(StreamHeader) ? stride - lit_bytes(cap) : 0; }

SWC_HD uint32_t make_match(uint32_t lit_run, uint32_t length, uint32_t distance) { return lit_run | ((length - 3u) << 8) | ((distance - 1u) << 16); }

// T threads per stream; KEEP = bytes of history a match can reach (32 KiB for Deflate, 64 KiB for LZ4); WIN = window buffer
template <int T, uint32_t KEEP = 32768, uint32_t WIN = 65536>
struct Lds {
    uint8_t win[WIN + 32];              // +32: 8-byte accesses may run past the last valid byte
    uint8_t litbuf[kLitBuf + 16];        // window of the literal stream, indexed by (literal offset % kLitBuf); +16: unaligned tail reads
    uint32_t ends[T + 1];                // end of record i, relative to the batch start (0xFFFFFFFF: no record)
    uint32_t dsts[T];                    // start of match i, relative to the batch start (== ends[i] for skips)
    uint64_t link[T];                    // (source position relative to the batch start : i32) | (producer : i32) << 32
    uint32_t periods[T];                 // distance of match i
    uint32_t done[(T + 31) / 32];
    uint32_t wave_sum[T / 64 + 1];
    uint32_t wave_sum2[T / 64 + 1];
    uint32_t ntake, span, litspan;
    uint16_t first[kSpan / kBucket + 2];  // first[b] = first record of the batch that ends after byte kBucket * b of the span
};

// Workgroup-collective helpers.  Device: T threads, barriers.  Host emulation: T == 1.
template <int T, uint32_t KEEP = 32768, uint32_t WIN = 65536>
struct Group {
    int tid;
    Lds<T, KEEP, WIN>* l;

    // Workgroup barrier that orders LDS only.  __syncthreads() also drains vmcnt, which would expose the latency of
    // the record / literal read-ahead loads and of the write-back stores at every barrier; no thread of the group
    // ever reads HBM bytes another thread of the group wrote, so LDS ordering is all the resolver needs.
    SWC_D void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    }
    // inclusive prefix sums of two values over the workgroup (sums stay below 2^32: callers clamp their inputs).
    // Within a wave: DPP row shifts + row broadcasts (six VALU ops per value, no LDS round trips); across waves: one
    // LDS exchange and one barrier for both.
    SWC_D static uint32_t wave_scan(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
#endif
        return x;
    }
    SWC_D void scan2_incl(uint32_t& x, uint32_t& y) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const int lane = tid & 63, wave = tid >> 6;
        x = wave_scan(x);
        y = wave_scan(y);
        if (T > 64) {
            if (lane == 63) { l->wave_sum[wave] = x; l->wave_sum2[wave] = y; }
            sync();
            uint32_t ax = 0, ay = 0;
            if (T <= 512) {
#pragma unroll
                for (int w = 0; w < T / 64; w++) {
                    ax += w < wave ? l->wave_sum[w] : 0u;
                    ay += w < wave ? l->wave_sum2[w] : 0u;
                }
            } else {   // 16 waves: a rolled loop keeps the 64-VGPR budget of a 1024-thread group free of spills
#pragma unroll 1
                for (int w = 0; w < wave; w++) {
                    ax += l->wave_sum[w];
                    ay += l->wave_sum2[w];
                }
            }
            x += ax;
            y += ay;
        }
#endif
    }
    SWC_D void set_done(int i) const {
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_or(&l->done[i >> 5], 1u << (i & 31), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        l->done[i >> 5] |= 1u << (i & 31);
#endif
    }
    SWC_D uint32_t done_word(int w) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return __hip_atomic_load(&l->done[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        return l->done[w];
#endif
    }
    SWC_D uint64_t link_load(int i) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return __hip_atomic_load(&l->link[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        return l->link[i];
#endif
    }
    SWC_D void link_store(int i, uint64_t v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_store(&l->link[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        l->link[i] = v;
#endif
    }
    SWC_D void backoff() const {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_sleep(1);
#endif
    }
};

struct u128 {
    uint32_t x, y, z, w;
};
SWC_HD u128 load_16(gcptr p) {  // p is 16-byte aligned
    return *(const SWC_AS_GLOBAL u128*)p;
}
SWC_HD void store_16(gptr p, const u128& v) { *(SWC_AS_GLOBAL u128*)p = v; }
SWC_HD uint8_t byte_of(const u128& v, int j) {
    uint32_t w = j < 4 ? v.x : j < 8 ? v.y : j < 12 ? v.z : v.w;
    return (uint8_t)(w >> (8 * (j & 3)));
}

template <int T, uint32_t KEEP = 32768, uint32_t WIN = 65536>
struct Resolver {
    static constexpr uint32_t kKeep = KEEP, kWin = WIN;
    Group<T, KEEP, WIN> g;
    gptr out;
    gcptr lits;       // the stream's dense literal stream (16-byte aligned base)
    uint64_t nlit;    // bytes in it
    uint64_t lit_cap; // bytes that may be READ from it (allocation size)
    uint64_t limit;   // bytes of `out` that exist: min(bytes produced, capacity)
    uint8_t* win;     // LDS window: position p lives at win[(uint32_t)p + woff]; the mapping keeps 16-byte aligned HBM
    uint32_t woff;    // chunks 16-byte aligned in LDS and moves down by multiples of 16 when the window slides
    uint8_t* litbuf;
    int dbg;          // experiment switches (tools/exp_resolve.py), 0 in production: 1 no copies, 2 no write-back,
                      // 4 no literal placement, 8 no dependency search

    SWC_D uint32_t idx(uint64_t p) const { return (uint32_t)p + woff; }
    SWC_D static uint32_t uniform(uint32_t v) {  // v is the same in every lane: keep it in an SGPR
#if defined(__HIP_DEVICE_COMPILE__)
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
        return v;
#endif
    }
    SWC_D static uint64_t L8(const uint8_t* p) { return *(const u64_unaligned*)p; }
    SWC_D static void S8(uint8_t* p, uint64_t v) { *(u64_unaligned*)p = v; }
    // the low n (< 8) bytes of v
    SWC_D static void Stail(uint8_t* p, uint32_t n, uint64_t v) {
        if (n >= 4) { *(u32_unaligned*)p = (uint32_t)v; v >>= 32; p += 4; n -= 4; }
        if (n >= 2) { *(u16_unaligned*)p = (uint16_t)v; v >>= 16; p += 2; n -= 2; }
        if (n) *p = (uint8_t)v;
    }

    // window -> HBM for positions [lo, hi), hi <= limit: 16-byte chunks by absolute address, edge chunks bytewise
    SWC_D void flush_span(uint64_t lo, uint64_t hi) const {
        if (hi <= lo) return;
        const uint64_t a0 = (uint64_t)(uintptr_t)out;
        const uint64_t c0 = (a0 + lo) >> 4, c1 = (a0 + hi - 1) >> 4;
        for (uint64_t c = c0 + (uint64_t)g.tid; c <= c1; c += T) {
            const int64_t p0 = (int64_t)((c << 4) - a0);
            gptr dst = out + p0;
            const uint8_t* src = win + idx((uint64_t)p0);
            if (p0 >= (int64_t)lo && (uint64_t)p0 + 16 <= hi) {
                store_16(dst, *(const u128*)src);
            } else {
                for (int j = 0; j < 16; j++) {
                    int64_t p = p0 + j;
                    if (p >= (int64_t)lo && (uint64_t)p < hi) dst[j] = src[j];
                }
            }
        }
    }

    // Slide the window down so that the position `cur` lands just above the kKeep bytes a match can still reach.
    // Called (group-uniformly) when the next batch might not fit.  Moves by a multiple of 16 bytes.
    SWC_D void slide(uint64_t cur) {
        const uint32_t ci = idx(cur);
        const uint32_t D = (ci - kKeep) & ~15u;
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr int kMv = (int)((kKeep / 16 + 2 + T - 1) / T);
        const uint32_t chunks = (ci - D + 15u) >> 4;
        u128 t[kMv];
#pragma unroll
        for (int k = 0; k < kMv; k++) {
            const uint32_t c = (uint32_t)g.tid + (uint32_t)k * T;
            if (c < chunks) t[k] = *(const u128*)(win + D + 16u * c);
        }
        g.sync();
#pragma unroll
        for (int k = 0; k < kMv; k++) {
            const uint32_t c = (uint32_t)g.tid + (uint32_t)k * T;
            if (c < chunks) *(u128*)(win + 16u * c) = t[k];
        }
        woff -= D;
        g.sync();
#else
        for (uint32_t i = 0; i + D < ci; i++) win[i] = win[i + D];
        woff -= D;
#endif
    }

    // One match inside the window: `len` bytes at index d; the first min(len, period) bytes (the pattern) come from
    // index s (all producers done, s + pattern <= d), the rest repeats the pattern with `period`.  Reads may run up to
    // 15 bytes past their source; the surplus is discarded.
    SWC_D void copy_match(uint32_t d, uint32_t len, uint32_t s, uint32_t period) const {
        uint8_t* dp = win + d;
        const uint8_t* sp = win + s;
        const uint32_t plen = len < period ? len : period;
        uint32_t i = 0;
        for (; i + 16 <= plen; i += 16) {
            const uint64_t a = L8(sp + i), b = L8(sp + i + 8);
            S8(dp + i, a);
            S8(dp + i + 8, b);
        }
        uint32_t rem = plen - i;
        if (rem) {
            uint64_t a = L8(sp + i);
            const uint64_t b = L8(sp + i + 8);
            if (rem >= 8) { S8(dp + i, a); a = b; i += 8; rem -= 8; }
            Stail(dp + i, rem, a);
        }
        if (len <= period) return;
        // the match overlaps itself: extend the pattern (LDS executes a lane's accesses in order)
        uint32_t k = period, back = period;
        if (period < 8) {
            back = ((7u + period) / period) * period;   // copy distance: a multiple of the period, >= 8
            const uint32_t stop = len < back ? len : back;
            for (; k < stop; k++) dp[k] = dp[k - period];
        }
        for (; k + 8 <= len; k += 8) S8(dp + k, L8(dp + k - back));
        if (k < len) Stail(dp + k, len - k, L8(dp + k - back));
    }

    // `n` (< 255) literal bytes at literal offset `lo` -> window index d.  [lw_lo, lw_hi) of the literal stream is
    // staged in litbuf; anything else is read from HBM.
    SWC_D void place_literals(uint32_t d, uint64_t lo, uint32_t n, uint64_t lw_lo, uint64_t lw_hi) const {
        const bool staged = lo >= lw_lo && lo + n + 8 <= lw_hi;
        for (uint32_t i = 0; i < n; i += 8) {
            uint64_t v;
            if (staged) v = L8(litbuf + ((uint32_t)(lo + i) & (kLitBuf - 1)));   // litbuf carries 16 mirrored bytes past its end
            else v = lo + i + 8 <= lit_cap ? load_u64(lits + lo + i) : 0;
            const uint32_t m = n - i;
            if (m >= 8) S8(win + d + i, v);
            else Stail(win + d + i, m, v);
        }
    }
    // literal stream [lo, lo + n) -> HBM output at position pos (clamped to `limit`), all threads; used for literal
    // runs of 255+ bytes and for the literals after the last match.
    SWC_D void stream_literals(uint64_t pos, uint64_t lo, uint64_t n) const {
        if (pos >= limit) return;
        if (pos + n > limit) n = limit - pos;
        for (uint64_t i = (uint64_t)g.tid * 8; i < n; i += (uint64_t)T * 8) {
            const uint64_t m = n - i;
            if (m >= 8) store_u64(out + pos + i, load_u64(lits + lo + i));
            else for (uint64_t j = 0; j < m; j++) out[pos + i + j] = lits[lo + i + j];
        }
    }
    // After a long literal run: point the window at `cur` and refill the kKeep bytes below it from the output (they are
    // final there: streamed literals, or spans written back earlier by this group -- hence the full barrier).
    SWC_D void rebuild(uint64_t cur) {
#if defined(__HIP_DEVICE_COMPILE__)
        __syncthreads();   // includes s_waitcnt vmcnt(0): this group's output stores have reached L2
#endif
        const uint32_t a0l = (uint32_t)(uintptr_t)out;
        woff = kKeep + ((a0l + (uint32_t)cur) & 15u) - (uint32_t)cur;
        const uint64_t hi = cur < limit ? cur : limit;
        const uint64_t lo = hi > kKeep ? hi - kKeep : 0;
        for (uint64_t p = lo + (uint64_t)g.tid * 8; p < hi; p += (uint64_t)T * 8) {
            if (p + 8 <= hi) S8(win + idx(p), load_u64(out + p));
            else for (uint64_t j = p; j < hi; j++) win[idx(j)] = out[j];
        }
        g.sync();
    }

    // Producer range of a source [v1, v2) that starts inside the batch span: a = first record with ends[a] > v1 -- from
    // the bucket index (the first record that ends after the bucket boundary at or below v1), then a short walk -- and
    // b = last record with dsts[b] < v2, walking forward from a (a source rarely spans more than two records).
    // Arrays are non-decreasing; records that were not taken hold 0xFFFFFFFF.
    SWC_D static void producers(const uint32_t* ends, const uint32_t* dsts, const uint16_t* first, int n, uint32_t v1, uint32_t v2, int& a, int& b) {
        int k = first[v1 / kBucket];
        while (k < n && ends[k] <= v1) k++;
        a = k;
        while (k < n && dsts[k] < v2) k++;
        b = k - 1;
    }

#if defined(SWC_RESOLVE_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define SWC_RP(k) { const uint64_t t_ = __builtin_readcyclecounter(); prof[k] += t_ - tlast; tlast = t_; }
#else
#define SWC_RP(k)
#endif
    SWC_D void run(const SWC_AS_GLOBAL uint32_t* recs, uint32_t nrec) {
        Lds<T, KEEP, WIN>* l = g.l;
#if defined(SWC_RESOLVE_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        uint64_t prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t tlast = __builtin_readcyclecounter();
#endif
        const int tid = g.tid;
        uint64_t rpos = 0;      // everything below rpos is final in HBM
        uint64_t lbase = 0;     // literal bytes consumed so far
        uint64_t lw_lo = 0, lw_hi = 0;   // window of the literal stream staged in litbuf
        uint32_t base = 0;
        uint32_t r_next = (uint32_t)tid < nrec ? recs[tid] : 0u;   // record prefetch, one batch ahead
        SWC_OPAQUE(r_next);   // wait for it HERE: a load still pending at loop entry would put an s_waitcnt vmcnt(0) at the
                              // loop head, which then drains the write-back stores of the previous batch in every iteration
        woff = (uint32_t)(uintptr_t)out & 15u;
        if (tid < (int)((T + 31) / 32)) l->done[tid] = 0;
        while (base < nrec) {
            if (idx(rpos) + kSpan + 16 > kWin) slide(rpos);
            const uint32_t wcur = idx(rpos);                                  // window index of the batch start
            const uint32_t lim_rel = limit <= rpos ? 0u : limit - rpos > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(limit - rpos);
            const bool valid = base + (uint32_t)tid < nrec;
            const uint32_t r = r_next;                                        // always the record at base + tid, already landed
            const uint32_t next_base = base + T;
            r_next = next_base + (uint32_t)tid < nrec ? recs[next_base + tid] : 0u;   // in flight while this batch resolves
            const uint32_t lit_run = r & 255u, len = ((r >> 8) & 255u) + 3u, dist = (r >> 16) + 1u;
            const bool skip = lit_run == kSkipMark;
            // a skip record never joins a batch: it (and everything after it) is "not taken", and when it comes first
            // the literal run is streamed on its own below
            uint32_t end_rel = !valid ? 0u : skip ? kSpan + 1u : lit_run + len;
            uint32_t lit_end = !valid || skip ? 0u : lit_run;
            SWC_RP(0)   // top: slide, record fetch
            g.scan2_incl(end_rel, lit_end);                                  // (barrier A inside)
            SWC_RP(1)   // scan + barrier A
            const bool take = valid && end_rel <= kSpan;
            const uint32_t dst_rel = end_rel - len;
            l->ends[tid] = take ? end_rel : 0xFFFFFFFFu;
            l->dsts[tid] = take ? dst_rel : 0xFFFFFFFFu;
            l->periods[tid] = dist;
            if (tid == 0) { l->ends[T] = 0xFFFFFFFFu; l->ntake = 0; l->span = r >> 8; l->litspan = 0; }
            if (take) {   // position -> record index: I own every bucket boundary inside [my start, my end)
                for (uint32_t bk = (dst_rel - lit_run + kBucket - 1) / kBucket; bk * kBucket < end_rel; bk++) l->first[bk] = (uint16_t)tid;
            }
            // my literal run goes into the window now; barrier C orders it before every match copy
            if (take && lit_run != 0 && !(dbg & 4)) {
                const uint32_t ls = dst_rel - lit_run;
                if (ls < lim_rel) place_literals(wcur + ls, lbase + lit_end - lit_run, lim_rel - ls < lit_run ? lim_rel - ls : lit_run, lw_lo, lw_hi);
            }
            SWC_RP(2)   // publish + literal placement
            g.sync();                                                        // barrier B: batch geometry published
            SWC_RP(3)   // barrier B
            if (take && l->ends[tid + 1] == 0xFFFFFFFFu) { l->ntake = (uint32_t)tid + 1u; l->span = end_rel; l->litspan = lit_end; }  // read after barrier C
            // clamp to the bytes that exist (a match that starts below the capacity may end beyond it)
            bool pending = take && dst_rel < lim_rel;
            const uint32_t clen = pending && lim_rel - dst_rel < len ? lim_rel - dst_rel : len;
            // Producers of my pattern bytes: records a..b (inclusive) of this batch.  A match whose pattern lies inside
            // ONE earlier match does not wait for it: it takes over that match's source (pointer jumping), so the
            // chains that repeated words form (every occurrence copies the previous one) collapse to depth one.
            int dep_a = 0, dep_b = -1, prod = -1;
            int32_t s0 = (int32_t)dst_rel - (int32_t)dist;
            const uint32_t plen = clen < dist ? clen : dist;
            if (pending) {
                const int32_t s1 = s0 + (int32_t)plen;  // exclusive
                if (s1 > 0 && !(dbg & 8)) {
                    const uint32_t lo_rel = s0 > 0 ? (uint32_t)s0 : 0u;
                    // first record that ends after my first source byte .. last record that starts before my source end
                    // (records that were not taken hold 0xFFFFFFFF, so the whole array can be searched)
                    producers(l->ends, l->dsts, l->first, T, lo_rel, (uint32_t)s1, dep_a, dep_b);
                    if (dep_b >= tid) dep_b = tid - 1;
                    if (dep_a <= dep_b) {
                        if (dep_a == dep_b && s0 >= 0 && (uint32_t)s0 >= l->dsts[dep_a] && (uint32_t)s1 <= l->ends[dep_a]) prod = dep_a;
                        else prod = -2;
                    }
                }
            }
            SWC_RP(4)   // producer search
            g.link_store(tid, (uint64_t)(uint32_t)s0 | ((uint64_t)(uint32_t)(pending ? prod : -1) << 32));
            if (take && !pending) g.set_done(tid);
            g.sync();                                                        // barrier C: links and literals published
            SWC_RP(5)   // barrier C
            const int n_take = (int)uniform(l->ntake);
            const uint32_t span = uniform(l->span);
            const uint32_t litspan = uniform(l->litspan);
            if (n_take == 0) {
                // record `base` is a skip: a literal run of `span` bytes, streamed straight to the output; the window is
                // then rebuilt around the new position
                stream_literals(rpos, lbase, span);
                rpos += span;
                lbase += span;
                base += 1;
                rebuild(rpos);
                r_next = base + (uint32_t)tid < nrec ? recs[base + tid] : 0u;   // the prefetch was for base + T
                SWC_OPAQUE(r_next);
                continue;
            }
            const uint64_t batch_end = rpos + span;
            const uint64_t hi = batch_end < limit ? batch_end : limit;
            // literal-stream read-ahead for the next batch: aligned 8-byte loads, HBM -> registers now,
            // registers -> litbuf after the resolve
            const uint64_t nw_lo = (lbase + litspan) & ~(uint64_t)7;
            constexpr int kLitPf = (int)(kLitBuf / 8) > T ? (int)(kLitBuf / 8) / T : 1;   // 8-byte granules per thread
            uint64_t pfv[kLitPf];
#pragma unroll
            for (int k = 0; k < kLitPf; k++) {
                const uint64_t o = ((uint64_t)k * T + (uint64_t)tid) * 8;
                pfv[k] = o < kLitBuf && nw_lo + o + 8 <= lit_cap ? load_u64(lits + nw_lo + o) : 0;
            }
            while (prod >= 0) {
                const int i = prod;
                const uint64_t li = g.link_load(i);
                const int32_t si = (int32_t)(uint32_t)li, pi = (int32_t)(uint32_t)(li >> 32);
                const uint32_t per_i = l->periods[i], len_i = l->ends[i] - l->dsts[i];
                uint32_t o = (uint32_t)s0 - l->dsts[i];
                bool wait = pi == -2;
                if (per_i < len_i) {          // the producer repeats its own pattern: map into the pattern if I fit
                    o %= per_i;
                    if (o + plen > per_i) wait = true;
                }
                if (wait) {                   // the producer itself has to wait, or I straddle its period: wait for it
                    dep_a = dep_b = i;
                    prod = -2;
                    break;
                }
                s0 = si + (int32_t)o;
                prod = pi;
                g.link_store(tid, (uint64_t)(uint32_t)s0 | ((uint64_t)(uint32_t)prod << 32));
            }
            SWC_RP(6)   // prefetch issue + pointer jumping
            // a waiting match runs once every producer has run; producers are earlier records, so this cannot deadlock
            while (pending) {
                bool ready = true;
                if (prod == -2) {
                    for (int w = dep_a >> 5; w <= (dep_b >> 5); w++) {
                        uint32_t m = 0xFFFFFFFFu;
                        if (w == (dep_a >> 5)) m &= 0xFFFFFFFFu << (dep_a & 31);
                        if (w == (dep_b >> 5)) m &= 0xFFFFFFFFu >> (31 - (dep_b & 31));
                        if ((g.done_word(w) & m) != m) { ready = false; break; }
                    }
                }
                if (ready) {
                    if (!(dbg & 1)) copy_match(wcur + dst_rel, clen, wcur + (uint32_t)s0, dist);
                    g.set_done(tid);
                    pending = false;
                } else {
                    g.backoff();
                }
            }
            SWC_RP(7)   // copies (incl. waiting for producers)
            g.sync();                                                        // barrier D: every copy of the batch is in the window
            SWC_RP(8)   // barrier D
            // consume the loads issued before the resolve (literal read-ahead, next records) BEFORE the write-back
            // stores are issued: vmcnt retires in order, so a wait placed after the stores would wait for them too
#pragma unroll
            for (int k = 0; k < kLitPf; k++) {
                const uint64_t o = ((uint64_t)k * T + (uint64_t)tid) * 8;
                if (o < kLitBuf) {
                    const uint32_t x = (uint32_t)(nw_lo + o) & (kLitBuf - 1);
                    *(uint64_t*)(litbuf + x) = pfv[k];
                    if (x < 16) *(uint64_t*)(litbuf + kLitBuf + x) = pfv[k];   // mirror of the first 16 bytes past the end
                }
            }
            lw_lo = nw_lo;
            lw_hi = nw_lo + kLitBuf;
            if (lw_hi > lit_cap) lw_hi = lit_cap & ~(uint64_t)7;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(r_next));
#endif
            if (!(dbg & 2)) flush_span(rpos, hi);
            if (tid < (int)((T + 31) / 32)) l->done[tid] = 0;
            SWC_RP(9)   // literal staging + write-back
            rpos = batch_end;
            lbase += litspan;
            base += (uint32_t)n_take;
            if (n_take != T) {   // the batch was cut at the span limit or a skip record: the prefetch was for base + T
                r_next = base + (uint32_t)tid < nrec ? recs[base + tid] : 0u;
                SWC_OPAQUE(r_next);   // wait here, in the rare path, so that the loop head needs no s_waitcnt vmcnt
            }
            // no barrier here: the next batch passes barriers A and B (or the slide's) before it touches anything read above
        }
        // literals after the last match
        if (nlit > lbase) stream_literals(rpos, lbase, nlit - lbase);
#if defined(SWC_RESOLVE_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        if ((tid & 63) == 0 && (tid >> 6) < 2) {   // waves 0 and 1 report into the slack at the end of the literal stream
            SWC_AS_GLOBAL uint64_t* dbg = (SWC_AS_GLOBAL uint64_t*)(const_cast<uint8_t*>((const uint8_t*)lits) + lit_cap - 16) - 10 * (1 + (tid >> 6));
            for (int k = 0; k < 10; k++) dbg[k] = prof[k];
        }
#endif
    }
};

// One job: `ws` is the stream's workspace area of `stride` bytes written by phase 1.
template <int T, uint32_t KEEP = 32768, uint32_t WIN = 65536>
SWC_D void resolve_job(const Job& job, const uint8_t* ws, size_t stride, Lds<T, KEEP, WIN>* lds, int tid, int dbg = 0) {
    const SWC_AS_GLOBAL StreamHeader* h = (const SWC_AS_GLOBAL StreamHeader*)ws;
    const size_t lo = lit_offset(stride, job.out_cap);
    if (lo == 0) return;   // no literal stream: phase 1 reported SWC_E_NEED_WORKSPACE for this job
    Resolver<T, KEEP, WIN> rs;
    rs.g.tid = tid;
    rs.g.l = lds;
    rs.out = (gptr)job.out;
    rs.lits = (gcptr)ws + lo;
    rs.nlit = h->nlit;
    rs.lit_cap = lit_bytes(job.out_cap);
    rs.limit = job.out_len < job.out_cap ? job.out_len : job.out_cap;
    rs.win = lds->win;
    rs.woff = 0;
    rs.litbuf = lds->litbuf;
    rs.dbg = dbg;
    rs.run((const SWC_AS_GLOBAL uint32_t*)(ws + sizeof(StreamHeader)), h->nrec);
}

}  // namespace lzr
}  // namespace swc
#endif
