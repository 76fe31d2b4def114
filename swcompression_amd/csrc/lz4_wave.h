// lz4_wave.h -- LZ4 block parse (phase 1 of the two-phase LZ4 path), one block per WAVEFRONT.
//
// Replaces the sequence parser of LZ4.process(block:_:) (reference Sources/LZ4/LZ4.swift:332-413): token and
// length parsing, the end-of-block rules the reference enforces (:369-376) and the offset
// validation (:380-383).  Like the Deflate path it never touches the output: literals go to the block's dense
// literal stream, every match becomes one or more 32-bit records, and the LZ77 resolve kernel of lz_resolve.h
// (64 KiB history, one block per workgroup) builds the output.
//
// Finding the sequence boundaries is a pointer chase (the start of a sequence is known only after the previous
// one has been parsed).  A wavefront breaks the chase into stripes of 64 input bytes:
//   1. every lane assumes that a sequence starts at ITS byte of the stripe and decodes it from one unaligned
//      16-byte load (token, up to 13 literals, the offset) -- "simple" sequences; anything else (length extension
//      bytes, offset 0) marks the lane unusual;
//   2. the true starts are found by following next[] from lane 0 with v_readlane (a few scalar cycles per hop
//      instead of a memory round trip per sequence);
//   3. the lanes that are true starts take their output and literal positions from a wave prefix sum, validate
//      their offset against the bytes produced so far, and write their record and their literals in parallel.
// An unusual lane, an invalid offset, the tail of the block and the last 17 KiB of output capacity are handled by a
// fully checked one-sequence step (the reference's control flow line by line, executed wave-uniformly, long
// literal runs copied by all lanes); it also carries the error taxonomy.
//
// The same source compiles for the host with W = 1 (tests/host_emu): stripes of one byte.
#ifndef SWC_LZ4_WAVE_H
#define SWC_LZ4_WAVE_H

#include "swc_common.h"
#include "lz_resolve.h"

namespace swc {
namespace lz4w {

constexpr uint32_t kKeep = 65536;    // LZ4 offsets reach 65,535 bytes back
constexpr int kRingLog2 = 17;        // LDS ring of the resolve kernel: history + one batch span + its cells
constexpr int kResolveThreads = 1024;
constexpr uint32_t kRecBuf = 1024;   // records a parse wave stages in LDS between flushes
constexpr uint32_t kLitStage = 4096; // literal bytes likewise
constexpr uint32_t kInWin = 1024;    // input window of a parse wave in LDS

template <int W>
struct Wave {
    int lane;
    SWC_D uint64_t ballot(bool p) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return __ballot(p);
#else
        return p ? 1ull : 0ull;
#endif
    }
    // value of lane `i` (i wave-uniform)
    SWC_D uint32_t read(uint32_t v, uint32_t i) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)i);
#else
        (void)i;
        return v;
#endif
    }
    SWC_D uint32_t first(uint32_t v) const {
#if defined(__HIP_DEVICE_COMPILE__)
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
        return v;
#endif
    }
    SWC_D uint32_t scan_incl(uint32_t x) const {
#if defined(__HIP_DEVICE_COMPILE__)
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
        x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
#endif
        return x;
    }
};

SWC_HD int popc64(uint64_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(m);
#else
    return __builtin_popcountll(m);
#endif
}
SWC_HD int ctz64(uint64_t m) {   // m != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long)m) - 1;
#else
    return __builtin_ctzll(m);
#endif
}
SWC_HD int top64(uint64_t m) {   // index of the highest set bit, m != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return 63 - __clzll((long long)m);
#else
    return 63 - __builtin_clzll(m);
#endif
}

template <int W>
struct Parser {
    Wave<W> w;
    gcptr in;
    uint64_t n;          // compressed bytes
    uint64_t cap;
    gptr lits;           // dense literal stream (capacity cap + 16)
    SWC_AS_GLOBAL uint32_t* recs;
    uint32_t max_rec;
    uint8_t* iw;         // kInWin + 16 bytes of LDS: window of the input, byte a at iw[a % kInWin], first 16 bytes mirrored
    uint64_t iw_hi;      // the window holds [iw_hi - kInWin, iw_hi)
    uint64_t iw_next;    // 8 bytes per lane of the chunk [iw_hi, iw_hi + kInChunk), in flight or landed
    bool iw_pf;          // iw_next is valid
    // The stripe path stages its records and literals in LDS and writes them out with wide stores every few dozen
    // stripes: vmcnt retires in order, so a global store per stripe would make the next stripe's input load wait for
    // a full store round trip (measured: 4,400 cycles per stripe).
    uint32_t* rbuf;      // kRecBuf records
    uint8_t* lbuf;       // kLitStage + 32 bytes (+ 64 for `wnd`)
    uint32_t rb_n, lb_n; // staged, not yet in HBM (nrec / nlit count them already)

    // wave-uniform state
    uint64_t ip, pos, nlit, sequences;
    int64_t last_match_start;
    uint32_t nrec;

    SWC_D uint64_t chunk_load(uint64_t base) const {   // this lane's 8 bytes of the chunk at `base`
        const uint64_t a = base + 8 * (uint64_t)w.lane;
        if (a + 8 <= n) return load_u64(in + a);
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) if (a + j < n) v |= (uint64_t)in[a + j] << (8 * j);
        return v;
    }
    SWC_D void chunk_store(uint64_t base, uint64_t v) const {
        const uint32_t x = (uint32_t)(base + 8 * (uint64_t)w.lane) & (kInWin - 1);
        *(uint64_t*)(iw + x) = v;
        if (x < 16) *(uint64_t*)(iw + kInWin + x) = v;
    }
    // make [ip, ip + W + 16) readable from the window; keep one chunk load in flight
    SWC_D void fill_window() {
        constexpr uint32_t kChunk = 8 * W;
        if (ip + W + 16 > iw_hi + kInWin || ip < iw_hi - (iw_hi < kInWin ? iw_hi : kInWin)) {   // far away (first stripe, after a long literal run)
            iw_hi = ip & ~(uint64_t)(kChunk - 1);
            iw_pf = false;
        }
        while (ip + W + 16 > iw_hi) {
            const uint64_t v = iw_pf ? iw_next : chunk_load(iw_hi);
            chunk_store(iw_hi, v);
            iw_hi += kChunk;
            iw_pf = false;
        }
        if (!iw_pf) {
            iw_next = chunk_load(iw_hi);
            iw_pf = true;
        }
    }
    // staged records / literals -> HBM, all lanes
    SWC_D void flush() {
        if (rb_n) {
            SWC_AS_GLOBAL uint32_t* dst = recs + (nrec - rb_n);
            for (uint32_t i = (uint32_t)w.lane; i < rb_n; i += (uint32_t)W) dst[i] = rbuf[i];
            rb_n = 0;
        }
        if (lb_n) {
            gptr dst = lits + (nlit - lb_n);
            for (uint32_t i = (uint32_t)w.lane * 8; i < lb_n; i += (uint32_t)W * 8) {
                if (i + 8 <= lb_n) store_u64(dst + i, *(const u64_unaligned*)(lbuf + i));
                else for (uint32_t j = i; j < lb_n; j++) dst[j] = lbuf[j];
            }
            lb_n = 0;
        }
    }

    SWC_D void push(uint32_t v) {
        if (nrec < max_rec) {   // (beyond the workspace: counted only, the job ends with SWC_E_NEED_WORKSPACE)
            if (rb_n >= kRecBuf) flush();
            if (w.lane == 0) rbuf[rb_n] = v;
            rb_n++;
        }
        nrec++;
    }
    // `cnt` literal bytes in[from ..] -> literal stream, all lanes; only the part below the capacity is kept.  Short runs
    // that sit in the LDS input window are staged (LDS -> LDS); anything else is flushed around and copied in HBM.
    SWC_D void copy_literals(uint64_t from, uint64_t cnt) {
        uint64_t keep = pos >= cap ? 0 : (cap - pos < cnt ? cap - pos : cnt);
        if (keep == 0) return;
        if (keep + lb_n <= kLitStage && from + keep <= iw_hi && from + kInWin >= iw_hi) {
            for (uint32_t i = (uint32_t)w.lane; i < (uint32_t)keep; i += (uint32_t)W) lbuf[lb_n + i] = iw[(uint32_t)(from + i) & (kInWin - 1)];
            lb_n += (uint32_t)keep;
            nlit += keep;
            return;
        }
        flush();
        for (uint64_t i = (uint64_t)w.lane * 8; i < keep; i += (uint64_t)W * 8) {
            if (i + 8 <= keep) store_u64(lits + nlit + i, load_u64(in + from + i));
            else for (uint64_t j = i; j < keep; j++) lits[nlit + j] = in[from + j];
        }
        nlit += keep;
    }
    SWC_D void push_lits(uint64_t n) {
        while (n > 0) {
            const uint32_t s = n > lzr::kMaxLitOnly ? lzr::kMaxLitOnly : (uint32_t)n;
            push(lzr::make_lits(s));
            n -= s;
        }
    }
    // records of one sequence: `lit` literal bytes (already in the literal stream as far as they lie below the
    // capacity), then a match of `mlen` bytes (0: none).  pos = position BEFORE the literals.
    SWC_D void emit(uint64_t lit, uint64_t mlen, uint32_t offset) {
        uint64_t run = pos >= cap ? 0 : (cap - pos < lit ? cap - pos : lit);   // literal bytes that were kept
        uint64_t p = pos + lit;                                                // match start
        if (mlen == 0 || p >= cap) { push_lits(run); return; }
        if (run > lzr::kLitRunMax) { push_lits(run); run = 0; }
        uint64_t rem = mlen;
        while (rem > 0) {
            const uint32_t piece = rem > lzr::kMaxLen ? lzr::kMaxLen : (uint32_t)rem;
            if (p < cap) push(lzr::make_match((uint32_t)run, piece, offset));
            run = 0;
            p += piece;
            rem -= piece;
        }
    }

    // One sequence with every check of the reference (LZ4.swift:341-412).  Returns SWC_OK to continue, -1 when the
    // block ended normally, or the error.
    // byte `a` of the block (a < n): from the LDS input window when it is there (the reads of the checked step depend
    // on each other; each one served from HBM/L2 would cost a memory round trip)
    SWC_D uint32_t rd(uint64_t a) const {
        return a < iw_hi && a + kInWin >= iw_hi ? (uint32_t)iw[(uint32_t)a & (kInWin - 1)] : (uint32_t)in[a];
    }
    SWC_D int careful_step() {
        if (ip < n) fill_window();
        sequences++;
        if (n - ip < 1) return SWC_E_DATA_TRUNCATED;                               // :344
        const uint32_t token = rd(ip++);
        uint64_t lit = token >> 4;
        if (lit == 15) {
            for (;;) {
                if (n - ip < 1) return SWC_E_DATA_TRUNCATED;                       // :350
                const uint32_t b = rd(ip++);
                lit += b;   // Int overflow (:355 unsupportedFeature) needs > 2^55 input bytes: unreachable
                if (b != 255) break;
            }
        }
        if (n - ip < lit) return SWC_E_DATA_TRUNCATED;                             // :363
        copy_literals(ip, lit);
        ip += lit;
        const uint64_t produced = pos + lit;                                       // out.endIndex of the reference (no dictionary on this path)
        if (ip >= n) {                                                             // :368 last sequence: literals only
            emit(lit, 0, 0);
            pos += lit;
            if (!(lit >= 5 || sequences == 1)) return SWC_E_DATA_CORRUPTED;        // :370
            if (!((int64_t)produced - last_match_start >= 12 || last_match_start == -1)) return SWC_E_DATA_CORRUPTED;  // :372
            return -1;
        }
        if (n - ip < 2) { emit(lit, 0, 0); pos += lit; return SWC_E_DATA_TRUNCATED; }   // :378
        const uint32_t offset = rd(ip) | (rd(ip + 1) << 8);
        ip += 2;
        if (!(offset > 0 && offset <= produced)) { emit(lit, 0, 0); pos += lit; return SWC_E_DATA_CORRUPTED; }  // :382
        uint64_t mlen = 4 + (token & 0xF);
        if (mlen == 19) {
            for (;;) {
                if (n - ip < 1) { emit(lit, 0, 0); pos += lit; return SWC_E_DATA_TRUNCATED; }  // :388
                const uint32_t b = rd(ip++);
                mlen += b;
                if (b != 255) break;
            }
        }
        last_match_start = (int64_t)produced;
        emit(lit, mlen, offset);
        pos += lit + mlen;
        return SWC_OK;
    }

    // One stripe of W input bytes starting at ip.  Returns false when nothing could be taken (the caller then runs
    // the checked step on the sequence at ip).
    SWC_D bool stripe() {
        // The stripe's bytes come from a 1 KiB window of the input kept in LDS and refilled 512 bytes at a time, one
        // refill in flight ahead of the parse: a load straight from HBM would depend on the chase of the previous stripe
        // and cost a memory round trip per stripe (measured: 2 us per stripe).
        fill_window();
        const uint32_t qi = (uint32_t)(ip + (uint64_t)w.lane) & (kInWin - 1);
        const uint64_t lo = *(const u64_unaligned*)(iw + qi), hi = *(const u64_unaligned*)(iw + qi + 8);
        const uint32_t token = (uint32_t)lo & 0xFFu, lit = token >> 4, mln = token & 15u;
        // bytes 1.. of the sequence
        const uint64_t b1 = (lo >> 8) | (hi << 56), b9 = hi >> 8;
        const uint32_t osh = 8u * lit;   // bit offset of the 2-byte offset field inside (b1, b9)
        const uint32_t offset = (uint32_t)((osh < 64 ? (b1 >> osh) | (osh ? b9 << (64 - osh) : 0) : b9 >> (osh - 64)) & 0xFFFFu);
        // a match length of 19..258 carries ONE extension byte right after the offset; it must sit inside the 16 bytes
        const uint32_t esh = osh + 16u;
        const uint32_t ext = (uint32_t)((esh < 64 ? (b1 >> esh) | (b9 << (64 - esh)) : b9 >> (esh - 64)) & 0xFFu);
        const bool longm = mln == 15;
        const bool ok = offset != 0 && (longm ? lit <= 12 && ext <= 239 : lit <= 13);
        const uint32_t mlen = 4 + mln + (longm ? ext : 0u);
        const uint32_t nxt = (uint32_t)w.lane + 3u + lit + (longm ? 1u : 0u);
        // follow the chain from lane 0 through the lanes that are ok
        const uint64_t okmask = w.ballot(ok);
        uint64_t mask = 0;
        uint32_t c = 0;
        while (c < (uint32_t)W && ((okmask >> c) & 1)) {
            mask |= 1ull << c;
            c = w.read(nxt, c);
        }
        if (mask == 0) return false;
        bool mine = (mask >> w.lane) & 1;
        // output / literal positions of the true starts; offsets must not reach before the block
        const uint32_t incl_adv = w.scan_incl(mine ? lit + mlen : 0u), incl_lit = w.scan_incl(mine ? lit : 0u);
        uint32_t my_adv = incl_adv - (mine ? lit + mlen : 0u), my_lit = incl_lit - (mine ? lit : 0u);
        const uint64_t my_pos = pos + my_adv;
        const uint64_t bad = w.ballot(mine && (uint64_t)offset > my_pos + lit);
        uint32_t tot_adv, tot_lit, end_c = c;
        if (bad) {
            const int fb = ctz64(bad);
            if (fb == 0) return false;
            mask &= (1ull << fb) - 1ull;
            mine = (mask >> w.lane) & 1;
            tot_adv = w.read(my_adv, (uint32_t)fb);
            tot_lit = w.read(my_lit, (uint32_t)fb);
            end_c = (uint32_t)fb;
        } else {
            tot_adv = w.read(incl_adv, (uint32_t)(W - 1));
            tot_lit = w.read(incl_lit, (uint32_t)(W - 1));
        }
        const int last = top64(mask);
        const uint32_t lms_lo = w.read((uint32_t)(my_pos + lit), (uint32_t)last), lms_hi = w.read((uint32_t)((my_pos + lit) >> 32), (uint32_t)last);
        if (mine) {
            const int rank = popc64(mask & ((1ull << w.lane) - 1ull));
            rbuf[rb_n + (uint32_t)rank] = lzr::make_match(lit, mlen, offset);
            uint8_t* d = lbuf + lb_n + my_lit;
            uint64_t v = b1;
            uint32_t r = lit;
            if (r >= 8) { *(u64_unaligned*)d = v; v = b9; d += 8; r -= 8; }
            if (r >= 4) { *(u32_unaligned*)d = (uint32_t)v; v >>= 32; d += 4; r -= 4; }
            if (r >= 2) { *(u16_unaligned*)d = (uint16_t)v; v >>= 16; d += 2; r -= 2; }
            if (r) d[0] = (uint8_t)v;
        }
        const uint32_t cnt = (uint32_t)popc64(mask);
        rb_n += cnt;
        lb_n += tot_lit;
        nrec += cnt;
        sequences += cnt;
        last_match_start = (int64_t)(((uint64_t)lms_hi << 32) | lms_lo);
        pos += tot_adv;
        nlit += tot_lit;
        ip += end_c;
        return true;
    }

    SWC_D int run() {
        // the stripe path needs: 16 loadable bytes after the last stripe byte, no sequence of the stripe being the last
        // one of the block (guaranteed: it ends inside the loaded bytes), room for W * (13 + 258) output bytes below
        // the capacity, and room for W records
        const bool fast_ok = (size_t)max_rec >= lzr::max_records(cap);
#if defined(SWC_LZ4_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        uint64_t t_s = 0, t_c = 0, n_s = 0, n_c = 0, t_f = 0;
#define SWC_T0 const uint64_t t0_ = __builtin_readcyclecounter();
#define SWC_T1(acc, cnt) { acc += __builtin_readcyclecounter() - t0_; cnt++; }
#else
#define SWC_T0
#define SWC_T1(acc, cnt)
#endif
        int result = SWC_OK;
        for (;;) {
            if (fast_ok && ip + (uint64_t)W + 24 <= n && pos + (uint64_t)W * 272 + 16 <= cap) {
                if (rb_n + (uint32_t)W > kRecBuf || lb_n + (uint32_t)W * 13 > kLitStage) flush();
                SWC_T0
                const bool took = stripe();
                SWC_T1(t_s, n_s)
                if (took) continue;
            }
            SWC_T0
            const int st = careful_step();
            SWC_T1(t_c, n_c)
            if (st == -1) break;
            if (st) { result = st; break; }
        }
#if defined(SWC_LZ4_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        if (w.lane == 0) {
            SWC_AS_GLOBAL uint64_t* dbg = (SWC_AS_GLOBAL uint64_t*)(lits + lzr::lit_bytes(cap) - 32);
            dbg[0] = t_s; dbg[1] = n_s; dbg[2] = t_c; dbg[3] = n_c;
        }
        (void)t_f;
#endif
        return result;
    }
};

// One wavefront = one job (blocks WITHOUT a dictionary prefix; those with one stay on lz4_lane.h).
template <int W>
SWC_D void lz4_parse_job(Job& job, uint8_t* ws, size_t ws_bytes, int lane, uint32_t* rbuf, uint8_t* lbuf, uint8_t* iw) {
    Parser<W> ps;
    ps.w.lane = lane;
    ps.iw = iw;
    ps.iw_hi = 0;
    ps.iw_next = 0;
    ps.iw_pf = false;
    ps.rbuf = rbuf;
    ps.lbuf = lbuf;

    ps.rb_n = ps.lb_n = 0;
    ps.in = (gcptr)job.in;
    ps.n = job.in_len;
    ps.cap = job.out_cap;
    ps.ip = ps.pos = ps.nlit = ps.sequences = 0;
    ps.last_match_start = -1;
    ps.nrec = 0;
    const size_t lo = ws ? lzr::lit_offset(ws_bytes, job.out_cap) : 0;
    ps.recs = (SWC_AS_GLOBAL uint32_t*)(ws + sizeof(lzr::StreamHeader));
    ps.max_rec = lo > sizeof(lzr::StreamHeader) ? (uint32_t)((lo - sizeof(lzr::StreamHeader)) / 4) : 0u;
    ps.lits = (gptr)(ws + lo);
    int st = lo == 0 ? SWC_E_NEED_WORKSPACE : ps.run();
    ps.flush();
    if (ps.nrec > ps.max_rec) {
        st = SWC_E_NEED_WORKSPACE;
        ps.nrec = ps.max_rec;
    }
    if (st == SWC_OK && ps.pos > ps.cap) st = SWC_E_CAPACITY;
    if (lane == 0) {
        if (ws && ws_bytes >= sizeof(lzr::StreamHeader)) {
            SWC_AS_GLOBAL lzr::StreamHeader* h = (SWC_AS_GLOBAL lzr::StreamHeader*)ws;
            h->nrec = ps.nrec;
            h->pad0 = 0;
            h->nlit = ps.nlit;
        }
    }
    job.out_len = ps.pos;
    job.in_consumed = ps.ip;
    job.status = st;
}

}  // namespace lz4w
}  // namespace swc
#endif
