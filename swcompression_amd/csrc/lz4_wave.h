// lz4_wave.h -- LZ4 block parse (phase 1 of the two-phase LZ4 path), one block per WAVEFRONT.
//
// Replaces the sequence parser of LZ4.process(block:_:) (reference Sources/LZ4/LZ4.swift:332-413): token and
// length parsing, the end-of-block rules the reference enforces (:369-376) and the offset
// validation (:380-383).  Like the Deflate path it never touches the output: literals go to the block's dense
// literal stream, every match becomes one or more 32-bit records, and the LZ77 resolve kernel of lz_resolve.h
// (64 KiB history, one block per workgroup) builds the output.
//
// Finding the sequence boundaries is a pointer chase (the start of a sequence is known only after the previous one has
// been parsed), but a parser started at a WRONG byte falls into step with the true sequence chain after a few sequences
// (every token it lands on sends it forward to another position; true starts are frequent).  So a wavefront parses the
// block the way inflate_sync.h decodes a Deflate stream, in rounds of 64 sub-chunks of kChunk input bytes:
//   walk         lane 0 starts at the true position, every other lane kWalkBack bytes in front of its sub-chunk; each parses
//                "short" sequences (at most one extension byte per length), nothing but the lengths, until it crosses the end
//                of its sub-chunk, and notes where it ended -- a walk that meets something it cannot take (on a wrong chain:
//                garbage) steps one byte on and keeps looking, it never ends in front of its sub-chunk;
//   provisional  every lane parses once more from where its left neighbour's walk ended: counts, one record per sequence
//   parse        into the wave's row-major scratch, the literals appended four at a time to groups in the same scratch;
//                repeated for the lanes whose start turns out wrong, until the chain of (start == left neighbour's end) holds
//                up to the first lane that met something the fast path does not take;
//   copy         wave scans give every lane its offsets; records and literal groups move to the record list and the dense
//                literal stream with 16-byte stores.
// The input of a round is staged in LDS with coalesced loads (one pad dword behind every sub-chunk keeps the lanes, which
// work at similar offsets of their sub-chunks, on different banks).  Longer sequences (more than 127 literals, more than one
// extension byte), an invalid offset, the tail of the block and the last 17 KiB of output capacity are handled by a fully
// checked one-sequence step (the reference's control flow line by line, executed wave-uniformly, long literal runs copied by
// all lanes); it also carries the error taxonomy.
//
// The same source compiles for the host (tests/host_emu): the SIMT regions run their 64 lanes one after another.
#ifndef SWC_LZ4_WAVE_H
#define SWC_LZ4_WAVE_H

#include "swc_common.h"
#include "simt.h"
#include "lz_resolve.h"

namespace swc {
namespace lz4w {

// The resolve kernel runs LZ4 blocks in the configuration of the Deflate path -- 512 threads, a 64 KiB ring with 32 KiB of
// history, two workgroups per CU -- although LZ4 offsets reach 65,535 bytes back: the few match bytes whose source is older
// than the ring's history (3 % on text) are read from the output buffer (lz_resolve.h: FAR).  A ring with the whole 64 KiB
// of history needs 128 KiB of LDS: one workgroup per CU, nothing to run while it waits at a barrier (round 2: half the
// per-byte speed of the Deflate resolve).
constexpr uint32_t kKeep = 32768;
constexpr int kRingLog2 = 16;
constexpr int kResolveThreads = 512;
constexpr uint32_t kRecBuf = 256;    // records the checked step stages in LDS between flushes
constexpr uint32_t kLitStage = 1024; // literal bytes likewise
constexpr uint32_t kInWin = 1024;    // input window of the checked step in LDS
#ifndef SWC_LZ4_CHUNK
#define SWC_LZ4_CHUNK 128
#endif
constexpr uint32_t kChunk = SWC_LZ4_CHUNK;            // input bytes per lane and round of the sub-chunk-parallel parse (a multiple of 16)
#ifndef SWC_LZ4_WALK_BACK
#define SWC_LZ4_WALK_BACK 64
#endif
constexpr uint32_t kWalkBack = SWC_LZ4_WALK_BACK;     // bytes in front of its sub-chunk at which a lane's walk begins
constexpr uint32_t kStageBytes = 64u * kChunk + 64u;  // staged input of a round (+ what the last lane reads past its sub-chunk)
// The staged input carries ONE pad dword behind every sub-chunk: the lanes of a wave work at similar offsets of their
// sub-chunks, and with a power-of-two stride they would all hit the same LDS bank.
constexpr uint32_t kChunkDwords = kChunk / 4;
static_assert((kChunkDwords & (kChunkDwords - 1)) == 0 && kChunk % 16 == 0, "the pad arithmetic wants a power of two");
SWC_HD uint32_t stage_slot(uint32_t dword) { return dword + dword / kChunkDwords; }
SWC_HD uint32_t stage_byte(uint32_t a) { return a + 4u * (a / kChunk); }
constexpr uint32_t kStageLds = kStageBytes + 4u * (kStageBytes / kChunk + 2u);
// The checked step's three small LDS buffers (input window, staged records, staged literals) live INSIDE the stage: the
// rounds start with a flush and end before the next checked step, so the two never hold data at the same time, and the
// wave's LDS is the stage alone (8.5 KB: 19 waves per CU -- the parse is latency-bound, its speed follows the waves a CU holds).
constexpr uint32_t kScratchWin = 0, kScratchRec = kInWin + 16, kScratchLit = kScratchRec + 4 * kRecBuf, kScratchEnd = kScratchLit + kLitStage + 32 + 64;
static_assert(kScratchEnd <= kStageLds && kScratchRec % 16 == 0 && kScratchLit % 16 == 0, "the checked step's buffers must fit the stage");
constexpr uint32_t kTailKeep = 16;                    // the sequences that end in the last bytes of a block stay with the checked step
constexpr uint32_t kPosFail = 0xFFFFFFFFu;
enum { kLzStop = 1u, kLzFail = 2u, kLzTrap = 4u, kLzTail = 8u, kLzLong = 16u };
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define SWC_LP(k) { const uint64_t t_ = __builtin_readcyclecounter(); pacc[k] += t_ - tlast; tlast = t_; }
#else
#define SWC_LP(k)
#endif
#if defined(SWC_HOST_EMULATION)
inline uint64_t g_lz4_stats[8];   // emulated parser: rounds, lane parses, passes, sequences taken by rounds, checked steps
#define SWC_LZ4_STAT(i, n) (g_lz4_stats[i] += (n))
#else
#define SWC_LZ4_STAT(i, n) ((void)0)
#endif

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
SWC_HD int clz64(uint64_t m) {   // m != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)m);
#else
    return __builtin_clzll(m);
#endif
}
SWC_HD int top64(uint64_t m) {   // index of the highest set bit, m != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return 63 - __clzll((long long)m);
#else
    return 63 - __builtin_clzll(m);
#endif
}

// R8: records of EIGHT bytes -- the 32-bit record of lz_resolve.h and, in the upper dword, the offset of its literal run in the
// block (LZ4 literals are byte-aligned in the input, LZ4.swift:364-366): no literal is copied anywhere by the parse, the copy
// kernel (lz_copy.h) fetches a run from the block itself.  The area then holds header | records | scratch rows.
// RM (record mode): 0 = four-byte records + the dense literal stream (swc_lz4_resolve_kernel); 1 = R8; 2 = four-byte records whose
// literal offsets the copier DERIVES: in the short form the rounds take (at most one extension byte per length) a sequence is
// 3 + [lit >= 15] + [mlen >= 19] bytes and its literals, so the start S of every sequence is a running sum over the records and
// its literals lie at S + 1 + [lit >= 15] -- no second dword per record (3.1 G records for 34 GB of text: 12.5 GB written twice by
// the parse and read once by the copy).  Where the rule does not hold -- the records of a sequence in long form, a literal run cut
// into literal-only records (those continue at S, S += lit), what follows them -- the parse, which SIMULATES the copier's sum as it
// pushes records, writes an ANCHOR (record index, S) into the area the literal stream would have had; the copier starts a new
// group at an anchor.  An anchor stands for 128 bytes of output or more, so cap / 8 entries are room enough.
template <int W, int RM = 0>
struct Parser {
    static constexpr bool R8 = RM == 1, R4 = RM == 2;
    Wave<W> w;
    SWC_AS_GLOBAL uint32_t* anc = nullptr;   // R4: anchors (record index, S) in the order of the records
    uint32_t nanc = 0, max_anc = 0;
    uint64_t s_pred = 0;                     // R4: where the copier's running sum stands (the start of the next sequence by its rule)
    SWC_D void anchor(uint32_t rec, uint64_t S) {
        if (nanc < max_anc && w.lane == 0) { anc[2u * nanc] = rec; anc[2u * nanc + 1u] = (uint32_t)S; }
        nanc++;
        s_pred = S;
    }
    // the copier's rule (lz_copy.h: Copier::seq_bytes): bytes from the start of a record's sequence to the start of the next
    SWC_HD static uint32_t seq_bytes(uint32_t li, uint32_t le) { return le ? 3u + li + (li >= 15u ? 1u : 0u) + (le >= 19u ? 1u : 0u) : li; }
    SWC_HD static uint32_t lit_skip(uint32_t li, uint32_t le) { return le ? 1u + (li >= 15u ? 1u : 0u) : 0u; }
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
    gptr prov = nullptr; // the wave's scratch rows in the workspace (lzr::kProvBytes in front of the literal stream), or none

    // wave-uniform state
    uint64_t ip, pos, nlit, sequences;
    int64_t last_match_start;
    uint32_t nrec;
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    uint64_t pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // cycles: 0 staging, 1 count passes, 2 scans, 3 emit, 4 checked steps; 5 rounds, 6 passes
    uint64_t tlast = 0;
#endif

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
            constexpr uint32_t kD = R8 ? 2u : 1u;   // dwords per record
            SWC_AS_GLOBAL uint32_t* dst = recs + kD * (nrec - rb_n);
            for (uint32_t i = (uint32_t)w.lane; i < kD * rb_n; i += (uint32_t)W) dst[i] = rbuf[i];
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

    SWC_D void push(uint32_t v, uint64_t from) {   // `from`: where the record's literals lie in the block (R8)
        if (nrec < max_rec) {   // (beyond the workspace: counted only, the job ends with SWC_E_NEED_WORKSPACE)
            if (rb_n >= kRecBuf) flush();
            if (w.lane == 0) {
                if (R8) { rbuf[2u * rb_n] = v; rbuf[2u * rb_n + 1u] = (uint32_t)from; }
                else rbuf[rb_n] = v;
            }
            rb_n++;
            if (R4) {   // where would the copier look for this record's literals?  An anchor if not where they are.
                const uint32_t le = (v >> 7) & 511u;
                const uint32_t li = (v & 127u) + (le ? 0u : (v >> 16) << 7);
                const uint32_t skip = lit_skip(li, le);
                if (li != 0u && s_pred + skip != from) anchor(nrec, from - skip);
                s_pred += seq_bytes(li, le);
            }
        }
        nrec++;
    }
    // `cnt` literal bytes in[from ..] -> literal stream, all lanes; only the part below the capacity is kept.  Short runs
    // that sit in the LDS input window are staged (LDS -> LDS); anything else is flushed around and copied in HBM.
    SWC_D void copy_literals(uint64_t from, uint64_t cnt) {
        uint64_t keep = pos >= cap ? 0 : (cap - pos < cnt ? cap - pos : cnt);
        if (keep == 0) return;
        if (RM != 0) { nlit += keep; return; }   // (they stay where they are: the record says where, or the copier's sum does)
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
    SWC_D void push_lits(uint64_t n, uint64_t from) {
        while (n > 0) {
            const uint32_t s = n > lzr::kMaxLitOnly ? lzr::kMaxLitOnly : (uint32_t)n;
            push(lzr::make_lits(s), from);
            from += s;
            n -= s;
        }
    }
    // records of one sequence: `lit` literal bytes (already in the literal stream as far as they lie below the
    // capacity), then a match of `mlen` bytes (0: none).  pos = position BEFORE the literals.
    SWC_D void emit(uint64_t lit, uint64_t mlen, uint32_t offset, uint64_t from) {
        uint64_t run = pos >= cap ? 0 : (cap - pos < lit ? cap - pos : lit);   // literal bytes that were kept
        uint64_t p = pos + lit;                                                // match start
        if (mlen == 0 || p >= cap) { push_lits(run, from); return; }
        if (run > lzr::kLitRunMax) { push_lits(run, from); run = 0; }
        uint64_t rem = mlen;
        while (rem > 0) {
            const uint32_t piece = rem > lzr::kMaxLen ? lzr::kMaxLen : (uint32_t)rem;
            if (p < cap) push(lzr::make_match((uint32_t)run, piece, offset), from);
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
        SWC_LZ4_STAT(4, 1);
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
        const uint64_t lit_at = ip;
        copy_literals(ip, lit);
        ip += lit;
        const uint64_t produced = pos + lit;                                       // out.endIndex of the reference (no dictionary on this path)
        if (ip >= n) {                                                             // :368 last sequence: literals only
            emit(lit, 0, 0, lit_at);
            pos += lit;
            if (!(lit >= 5 || sequences == 1)) return SWC_E_DATA_CORRUPTED;        // :370
            if (!((int64_t)produced - last_match_start >= 12 || last_match_start == -1)) return SWC_E_DATA_CORRUPTED;  // :372
            return -1;
        }
        if (n - ip < 2) { emit(lit, 0, 0, lit_at); pos += lit; return SWC_E_DATA_TRUNCATED; }   // :378
        const uint32_t offset = rd(ip) | (rd(ip + 1) << 8);
        ip += 2;
        if (!(offset > 0 && offset <= produced)) { emit(lit, 0, 0, lit_at); pos += lit; return SWC_E_DATA_CORRUPTED; }  // :382
        uint64_t mlen = 4 + (token & 0xF);
        if (mlen == 19) {
            for (;;) {
                if (n - ip < 1) { emit(lit, 0, 0, lit_at); pos += lit; return SWC_E_DATA_TRUNCATED; }  // :388
                const uint32_t b = rd(ip++);
                mlen += b;
                if (b != 255) break;
            }
        }
        last_match_start = (int64_t)produced;
        emit(lit, mlen, offset, lit_at);
        pos += lit + mlen;
        return SWC_OK;
    }

    // ---- sub-chunk-parallel parse ----------------------------------------------------------------------------------------
    // Where a sequence starts is known only when the one before it has been parsed, but a parse started at a WRONG byte
    // falls into step with the true sequence chain after a few sequences (it lands on a new byte after every sequence;
    // about one byte in eleven is a true sequence start).  The wave stages the next 64 sub-chunks of kChunk bytes in LDS and
    // runs the round the way inflate_sync.h runs a Deflate round:
    //   walk         lane 0 starts at the true position, every other lane kWalkBack bytes IN FRONT of its sub-chunk; all parse
    //                -- nothing but the lengths -- until they cross the end of their sub-chunk and note where they ended;
    //   provisional  every lane parses ONCE more from where its left neighbour's walk ended: counts, and ONE record per
    //   parse        sequence into the wave's row-major scratch (lz_resolve.h: row k = the k-th record of all 64 lanes);
    //   chain check  lane k must end where lane k + 1 began; lanes for which it does not parse again;
    //   copy         wave scans give every lane its offsets; the records move to the record list with 16-byte stores, and the
    //                literals -- found again by walking the lane's OWN records, no token is read a third time -- from the stage
    //                to the dense literal stream.
    // A lane stops IN FRONT OF a sequence it does not take: one that ends in the last kTailKeep bytes of the block (the
    // end-of-block rules are the checked step's), one that needs bytes beyond the staged window or more than one extension byte
    // per length, more than 127 literals, offset 0; the round then ends with that lane.
    struct ProvOut {
        uint32_t end;      // byte (relative to the round base) of the first sequence NOT taken
        uint32_t nlit, nrec, nout;
        uint32_t lms;      // output offset (inside the sub-chunk) of the last match start
        uint32_t flags;
        int32_t need;      // the largest (offset - output bytes of the sub-chunk in front of the match)
    };
    SWC_D static uint32_t rd32(const uint8_t* stage, uint32_t a) {   // the four bytes at `a` (aligned LDS reads + a byte shift)
        const uint32_t* st32 = (const uint32_t*)stage;
        const uint32_t lo = st32[stage_slot(a >> 2)], hi = st32[stage_slot((a >> 2) + 1u)];
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_alignbyte(hi, lo, a & 3u);
#else
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (a & 3u)));
#endif
    }
    // One sequence at `ip`, SHORT form only (at most one extension byte per length): literal count, match length, offset,
    // where the literals are and where the next sequence starts; `stop`: the fast path does not take it.
    struct Seq {
        uint32_t lit, mlen, offset, lit_at, next;
        bool stop, tail, longer;   // longer: stopped for its lengths (more than one extension byte, more than kLitRunMax literals): no round takes it
    };
    SWC_D static Seq sequence_at(const uint8_t* stage, uint32_t ip, uint32_t safe, uint32_t tail_limit) {
        Seq q;
        const uint32_t tok32 = rd32(stage, ip);
        const uint32_t token = tok32 & 0xFFu, b1 = (tok32 >> 8) & 0xFFu;
        const uint32_t l0 = token >> 4, m0 = token & 15u;
        const uint32_t lx = l0 == 15 ? 1u : 0u;
        q.lit = l0 + (lx ? b1 : 0u);
        q.lit_at = ip + 1u + lx;
        const uint32_t p = q.lit_at + q.lit;                   // the offset field (<= ip + 272)
        const uint32_t pr = p > safe ? safe : p;               // (read something harmless when the lane is about to stop)
        const uint32_t off32 = rd32(stage, pr);
        q.offset = off32 & 0xFFFFu;
        const uint32_t b2 = (off32 >> 16) & 0xFFu;
        const uint32_t mx = m0 == 15 ? 1u : 0u;
        q.mlen = 4u + m0 + (mx ? b2 : 0u);
        q.next = p + 2u + mx;
        // long lengths, the staged window, the tail of the block: the lane stops in front of it
        q.stop = ((lx & (uint32_t)(b1 == 255)) | (mx & (uint32_t)(b2 == 255)) | (uint32_t)(p > safe) | (uint32_t)(q.next > tail_limit) | (uint32_t)(q.lit > lzr::kLitRunMax)) != 0u;
        q.tail = q.next > tail_limit && p <= safe;
        q.longer = ((lx & (uint32_t)(b1 == 255)) | (mx & (uint32_t)(b2 == 255)) | (uint32_t)(q.lit > lzr::kLitRunMax)) != 0u && p <= safe;
        return q;
    }
    // Where does a parse from `start` end?  Returns the first sequence start at or beyond `chunk_end` (or where the lane stopped).
    SWC_D static uint32_t walk_chunk(const uint8_t* stage, uint32_t stage_len, uint32_t start, uint32_t chunk_end, uint32_t tail_limit) {
        uint32_t ip = start;
        const uint32_t safe = stage_len >= 8 ? stage_len - 8u : 0u;   // a read of four bytes at or below this stays inside the staged window
        while (ip < chunk_end) {
            if (ip > safe) break;
            const Seq q = sequence_at(stage, ip, safe, tail_limit);
            // Something the fast path does not take: on the TRUE chain the provisional parse will stop there and end the round,
            // whatever this walk says; on a wrong chain (the usual case: garbage looks like long lengths one time in thirty) the
            // walk must not die -- a lane whose walk ends in front of its sub-chunk sends its neighbour off on a wrong start -- so
            // it steps one byte on and keeps looking for the true chain.
            ip = (q.stop || q.offset == 0) ? ip + 1u : q.next;
        }
        return ip;
    }
    // The provisional parse: from `start` until a sequence would begin at or beyond `chunk_end`.  Per step either the next
    // sequence is parsed -- its record goes to the row of the lane's next record, its counts are added -- or (a sequence with
    // more than four literals) the step only carries literals on; either way up to four literal bytes of the current sequence
    // are appended to the accumulator (oldest byte lowest), whose low dword is written to the row of the literal group being
    // filled when that group is full (the last, incomplete one behind the loop).
    SWC_D static void parse_chunk_prov(const uint8_t* stage, uint32_t stage_len, uint32_t start, uint32_t chunk_end, uint32_t tail_limit, gptr prov,
                                       uint32_t lane, ProvOut& r) {
        uint32_t ip = start, nlit = 0, nout = 0, lms = 0, flags = 0;
        uint32_t roff = 4u * lane + kProvRow;                  // byte offset of my next record in the scratch (row 1 is the first)
        uint32_t loff = (uint32_t)lzr::kProvRecBytes + 4u * lane + kProvRow;   // ... of the literal group being filled
        uint32_t acc = 0, nacc = 0;                             // literal bytes of the group being filled
        uint32_t rem = 0, src = 0;                              // literal bytes of the current sequence not yet taken, where they are
        int32_t need = -0x40000000;
        const uint32_t safe = stage_len >= 8 ? stage_len - 8u : 0u;
        bool go = ip < chunk_end;
        while (go) {
            if (rem == 0u) {
                bool bad = ip > safe;
                if (bad) flags |= kLzStop;
                else {
                    const Seq q = sequence_at(stage, ip, safe, tail_limit);
                    bad = q.stop || q.offset == 0;
                    if (bad) flags |= q.stop ? (q.tail ? kLzStop | kLzTail : q.longer ? kLzStop | kLzLong : kLzStop) : kLzFail;   // offset 0: LZ4.swift:382
                    else {
                        const int32_t nd = (int32_t)q.offset - (int32_t)(nout + q.lit);          // :382 offset <= bytes produced, checked after the scan
                        need = nd > need ? nd : need;
                        store_u32(prov + roff, lzr::make_match(q.lit, q.mlen, q.offset));
                        roff += kProvRow;
                        nlit += q.lit;
                        lms = nout + q.lit;
                        nout += q.lit + q.mlen;
                        ip = q.next;
                        rem = q.lit;
                        src = q.lit_at;
                    }
                }
                if (bad) go = false;
            }
            const uint32_t take = rem < 4u ? rem : 4u;
            const uint32_t w = rd32(stage, src) & (uint32_t)(((uint64_t)1 << (8u * take)) - 1u);
            const uint64_t a64 = (uint64_t)acc | ((uint64_t)w << (8u * nacc));
            const uint32_t tot = nacc + take, full = tot >> 2;
            if (full) store_u32(prov + loff, (uint32_t)a64);   // (masked per lane: the memory pipeline's work is per active lane, and a group rewritten every step reaches HBM every time)
            acc = full ? (uint32_t)(a64 >> 32) : (uint32_t)a64;
            nacc = tot & 3u;
            loff += full * kProvRow;
            rem -= take;
            src += take;
            go = go && (rem > 0u || ip < chunk_end);
        }
        if (nacc != 0u) store_u32(prov + loff, acc);   // (the bytes a full group left over in the last step)
        r.end = ip; r.nlit = nlit; r.nrec = (roff - 4u * lane) / kProvRow - 1u; r.nout = nout; r.lms = lms; r.flags = flags; r.need = need;
    }
    // A lane's piece of the round moves from its column of the scratch to its final place: `nrec` records to `rdst` (dword
    // aligned), `nlit` literal bytes to `ldst` (any alignment).  The loads of a step read one row: coalesced.  The last,
    // incomplete literal group holds its bytes at the bottom.
    SWC_D static void copy_prov(gcptr plit, gcptr prec, uint32_t nlit, uint32_t nrec, gptr ldst, SWC_AS_GLOBAL uint32_t* rdst) {
        const uint32_t ngrp = (nlit + 3u) >> 2;
        for (uint32_t i = 0, g = 0; i < nrec || g < ngrp; i += 16, g += 8) {
            uint32_t v[16], w[8];
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t row = i + k + 1u < (uint32_t)lzr::kProvRecRows ? i + k + 1u : (uint32_t)lzr::kProvRecRows - 1u;
                v[k] = load_u32(prec + (size_t)row * kProvRow);
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t row = g + k + 1u < (uint32_t)lzr::kProvLitRows ? g + k + 1u : (uint32_t)lzr::kProvLitRows - 1u;
                w[k] = load_u32(plit + (size_t)row * kProvRow);
            }
#pragma unroll
            for (uint32_t k = 0; k < 16; k += 4) {
                if (i + k + 4u <= nrec) store_u128_a4((gptr)(rdst + i + k), v[k], v[k + 1], v[k + 2], v[k + 3]);
                else {
#pragma unroll
                    for (uint32_t q = 0; q < 4; q++) if (i + k + q < nrec) rdst[i + k + q] = v[k + q];
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < 8; k += 4) {
                const uint32_t at = 4u * (g + k);
                if (at + 16u <= nlit) store_u128_a4(ldst + at, w[k], w[k + 1], w[k + 2], w[k + 3]);
                else {
#pragma unroll
                    for (uint32_t q = 0; q < 4; q++) {
                        const uint32_t aq = at + 4u * q;
                        if (aq + 4u <= nlit) store_u32(ldst + aq, w[k + q]);
                        else if (aq < nlit) { uint32_t x = w[k + q]; for (uint32_t z = aq; z < nlit; z++, x >>= 8) ldst[z] = (uint8_t)x; }
                    }
                }
            }
        }
    }
    static constexpr uint32_t kProvRow = 64u * 4u;

    // ---- R8: the same two steps without literals -- one sequence and one eight-byte record (record | literal offset << 32) per
    // step, rows of 512 bytes in the record part of the scratch (a sub-chunk holds at most (kChunk + 2) / 3 sequences)
    static constexpr uint32_t kProvRow8 = 64u * 8u;
    static_assert(((kChunk + 2u) / 3u + 3u) * kProvRow8 <= lzr::kProvRecBytes, "the scratch rows hold a sub-chunk's records");
    // (R4: the same with four-byte records in rows of 256 bytes)
    SWC_D static void parse_chunk_prov8(const uint8_t* stage, uint32_t stage_len, uint32_t start, uint32_t chunk_end, uint32_t tail_limit, gptr prov,
                                        uint32_t lane, uint32_t base32, ProvOut& r) {
        constexpr uint32_t kRecB = R4 ? 4u : 8u, kRow = 64u * kRecB;
        uint32_t ip = start, nlit = 0, nout = 0, lms = 0, flags = 0;
        uint32_t roff = kRecB * lane + kRow;                   // byte offset of my next record in the scratch (row 1 is the first)
        int32_t need = -0x40000000;
        const uint32_t safe = stage_len >= 8 ? stage_len - 8u : 0u;
        while (ip < chunk_end) {
            if (ip > safe) { flags |= kLzStop; break; }
            const Seq q = sequence_at(stage, ip, safe, tail_limit);
            if (q.stop || q.offset == 0) {
                flags |= q.stop ? (q.tail ? kLzStop | kLzTail : q.longer ? kLzStop | kLzLong : kLzStop) : kLzFail;   // offset 0: LZ4.swift:382
                break;
            }
            const int32_t nd = (int32_t)q.offset - (int32_t)(nout + q.lit);          // :382 offset <= bytes produced, checked after the scan
            need = nd > need ? nd : need;
            if (R4) store_u32(prov + roff, lzr::make_match(q.lit, q.mlen, q.offset));
            else store_u64(prov + roff, (uint64_t)lzr::make_match(q.lit, q.mlen, q.offset) | ((uint64_t)(base32 + q.lit_at) << 32));
            roff += kRow;
            nlit += q.lit;
            lms = nout + q.lit;
            nout += q.lit + q.mlen;
            ip = q.next;
        }
        r.end = ip; r.nlit = nlit; r.nrec = (roff - kRecB * lane) / kRow - 1u; r.nout = nout; r.lms = lms; r.flags = flags; r.need = need;
    }
    // `nrec` four-byte records from the lane's column of the scratch to `rdst` (R4)
    SWC_D static void copy_prov4(gcptr prec, uint32_t nrec, SWC_AS_GLOBAL uint32_t* rdst) {
        for (uint32_t i = 0; i < nrec; i += 16) {
            uint32_t v[16];
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t row = i + k + 1u < (uint32_t)lzr::kProvRecRows ? i + k + 1u : (uint32_t)lzr::kProvRecRows - 1u;
                v[k] = load_u32(prec + (size_t)row * kProvRow);
            }
#pragma unroll
            for (uint32_t k = 0; k < 16; k += 4) {
                if (i + k + 4u <= nrec) store_u128_a4((gptr)(rdst + i + k), v[k], v[k + 1], v[k + 2], v[k + 3]);
                else {
#pragma unroll
                    for (uint32_t q = 0; q < 4; q++) if (i + k + q < nrec) rdst[i + k + q] = v[k + q];
                }
            }
        }
    }
    // `nrec` eight-byte records from the lane's column of the scratch to `rdst` (8-byte aligned)
    SWC_D static void copy_prov8(gcptr prec, uint32_t nrec, SWC_AS_GLOBAL uint32_t* rdst) {
        constexpr uint32_t kRows = (uint32_t)(lzr::kProvRecBytes / kProvRow8);
        for (uint32_t i = 0; i < nrec; i += 16) {
            uint64_t v[16];
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t row = i + k + 1u < kRows ? i + k + 1u : kRows - 1u;
                v[k] = load_u64(prec + (size_t)row * kProvRow8);
            }
#pragma unroll
            for (uint32_t k = 0; k < 16; k += 2) {
                if (i + k + 2u <= nrec) store_u128_a4((gptr)(rdst + 2u * (i + k)), (uint32_t)v[k], (uint32_t)(v[k] >> 32), (uint32_t)v[k + 1], (uint32_t)(v[k + 1] >> 32));
                else if (i + k < nrec) store_u64((gptr)(rdst + 2u * (i + k)), v[k]);
            }
        }
    }

    // Rounds from `ip` on, as long as whole rounds can be committed.  Returns when the checked step has to take over
    // (the end of the block is near, a sequence that does not fit a round, anything invalid, the capacity).
    SWC_D void sync_rounds(uint8_t* stage) {
        using simt::PT;
        constexpr int N = kWave;
        flush();
        iw_hi = 0;        // the input window of the checked step shares the stage: nothing of it survives a round
        iw_pf = false;
        PT<uint32_t, N> start, endp, pe, c_lit, c_rec, c_out, c_lms, c_need, flg, x_lit, x_rec, x_out;
        PT<bool, N> pb, have;
        for (;;) {
            if (n - ip < 128) return;                              // the last bytes of a block are the checked step's
            SWC_LP(4)
            const uint64_t B = ip & ~(uint64_t)3;
            const uint32_t start0 = (uint32_t)(ip - B);
            const uint64_t avail = n - B;
            const uint32_t stage_len = avail < kStageBytes ? (uint32_t)avail : kStageBytes;
            const uint32_t tail_limit = avail - kTailKeep > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)(avail - kTailKeep);
            constexpr uint32_t kPartsAll = (kStageBytes + 16u * N - 1u) / (16u * N), kParts = (kPartsAll + 1u) / 2u;
#pragma unroll 1
            for (uint32_t half = 0; half < 2; half++) {   // the loads of HALF a round are issued before the first is used: two memory latencies per round, not nine
                PT<uint64_t, N> sa[kParts], sb[kParts];
                SIMT_BEGIN(t, N)
#pragma unroll
                    for (uint32_t k = 0; k < kParts; k++) {
                        const uint32_t o = 16u * (uint32_t)t + (half * kParts + k) * 16u * N;
                        uint64_t a = 0, b = 0;
                        if (o < kStageBytes) {
                            if (o + 16 <= stage_len) { a = load_u64(in + B + o); b = load_u64(in + B + o + 8); }
                            else {
                                for (uint32_t q = 0; q < 8; q++) if (o + q < stage_len) a |= (uint64_t)in[B + o + q] << (8 * q);
                                for (uint32_t q = 0; q < 8; q++) if (o + 8 + q < stage_len) b |= (uint64_t)in[B + o + 8 + q] << (8 * q);
                            }
                        }
                        sa[k][t] = a; sb[k][t] = b;
                    }
                SIMT_END
                SIMT_BEGIN(t, N)
#pragma unroll
                    for (uint32_t k = 0; k < kParts; k++) {
                        const uint32_t o = 16u * (uint32_t)t + (half * kParts + k) * 16u * N;
                        if (o < kStageBytes) {
                            uint32_t* st32 = (uint32_t*)stage;   // (a 16-byte piece never straddles a sub-chunk: the pad keeps its dwords together)
                            const uint32_t d = stage_slot(o >> 2);
                            const uint64_t a = sa[k][t], b = sb[k][t];
                            st32[d] = (uint32_t)a; st32[d + 1] = (uint32_t)(a >> 32); st32[d + 2] = (uint32_t)b; st32[d + 3] = (uint32_t)(b >> 32);
                        }
                    }
                SIMT_END_WAVE
            }
            SIMT_BEGIN(t, N)
                const uint32_t cs = (uint32_t)t * kChunk;
                start[t] = t == 0 ? start0 : (cs > kWalkBack + start0 ? cs - kWalkBack : start0);   // (never in front of the true position)
                have[t] = false;
                flg[t] = 0;
            SIMT_END
            SWC_LP(0)
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
            pacc[5]++;
            pacc[6]++;
#endif
            SWC_LZ4_STAT(0, 1);
            SWC_LZ4_STAT(2, 1);
            // the walk: where does a parse from my guess end?
            SIMT_BEGIN(t, N)
                const uint32_t ce = t == N - 1 ? stage_len : ((uint32_t)t + 1u) * kChunk;
                endp[t] = walk_chunk(stage, stage_len, start[t], ce, tail_limit);
            SIMT_END
            SWC_LP(1)
            uint32_t nv = 0;
            int E = 64;
            for (;;) {
                simt::wave_shift_up<N>(pe, endp, start0);
                SIMT_BEGIN(t, N) pb[t] = !(have[t] && (t == 0 || start[t] == pe[t])); SIMT_END
                const uint64_t m_bad = simt::wave_ballot<N>(pb);
                const int b = m_bad ? simt::ctz64(m_bad) : 64;           // lanes [0, b) are on the true chain
                SIMT_BEGIN(t, N) pb[t] = flg[t] != 0; SIMT_END
                const uint64_t m_stop = simt::wave_ballot<N>(pb) & (b == 64 ? ~0ull : (1ull << b) - 1ull);
                E = m_stop ? simt::ctz64(m_stop) : 64;                   // the lane that stopped in front of a sequence
                nv = (uint32_t)(E < 64 ? E + 1 : b);
                if (E < 64 || b == 64) break;
                SWC_LZ4_STAT(2, 1);
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
                pacc[6]++;
#endif
                SIMT_BEGIN(t, N)
                    const bool todo = t == 0 ? !have[t] : (start[t] != pe[t] || !have[t]);
                    if (todo) {
                        if (t != 0) start[t] = pe[t];
                        SWC_LZ4_STAT(1, 1);
                        ProvOut r;
                        const uint32_t ce = t == N - 1 ? stage_len : ((uint32_t)t + 1u) * kChunk;
                        if (RM != 0) parse_chunk_prov8(stage, stage_len, start[t], ce, tail_limit, prov, (uint32_t)t, (uint32_t)B, r);
                        else parse_chunk_prov(stage, stage_len, start[t], ce, tail_limit, prov, (uint32_t)t, r);
                        endp[t] = r.end; c_lit[t] = r.nlit; c_rec[t] = r.nrec; c_out[t] = r.nout; c_lms[t] = r.lms; flg[t] = r.flags;
                        c_need[t] = (uint32_t)r.need;
                        have[t] = true;
                    }
                SIMT_END
            }
            SWC_LP(3)
            SIMT_BEGIN(t, N)
                const bool v = (uint32_t)t < nv;
                x_lit[t] = v ? c_lit[t] : 0u; x_rec[t] = v ? c_rec[t] : 0u; x_out[t] = v ? c_out[t] : 0u;
            SIMT_END
            simt::wave_scan_incl<N>(x_lit);
            simt::wave_scan_incl<N>(x_rec);
            simt::wave_scan_incl<N>(x_out);
            const uint32_t tot_lit = simt::wave_read<N>(x_lit, N - 1), tot_rec = simt::wave_read<N>(x_rec, N - 1);
            const uint32_t tot_out = simt::wave_read<N>(x_out, N - 1);
            const uint32_t stop_flags = E < 64 ? simt::wave_read<N>(flg, E) : 0u;
            SWC_LP(2)
            if (tot_rec == 0) return;                                         // not even one sequence: the checked step
            if (pos + tot_out > cap || (uint64_t)nrec + tot_rec > max_rec) return;   // the capacity / the workspace: the checked step counts on
            // :382 every offset must reach back no further than the bytes produced in front of its match
            SIMT_BEGIN(t, N)
                const uint64_t p0 = pos + (x_out[t] - c_out[t]);
                const int32_t room = p0 > 0x40000000ull ? 0x40000000 : (int32_t)p0;
                pb[t] = (uint32_t)t < nv && c_rec[t] != 0 && (int32_t)c_need[t] > room;
            SIMT_END
            if (simt::wave_ballot<N>(pb)) return;                             // an offset beyond the output: the checked step reports it
            if (R4) {   // the sequences of a round follow the copier's rule among themselves: an anchor only if its running sum does not stand at the round's first sequence
                if (s_pred != ip) anchor(nrec, ip);
            }
            SIMT_BEGIN(t, N)
                if ((uint32_t)t < nv && c_rec[t] != 0) {
                    if (R4) copy_prov4(prov + 4u * (uint32_t)t, c_rec[t], recs + nrec + (x_rec[t] - c_rec[t]));
                    else if (R8) copy_prov8(prov + 8u * (uint32_t)t, c_rec[t], recs + 2u * (nrec + (x_rec[t] - c_rec[t])));
                    else copy_prov(prov + lzr::kProvRecBytes + 4u * (uint32_t)t, prov + 4u * (uint32_t)t, c_lit[t], c_rec[t], lits + nlit + (x_lit[t] - c_lit[t]), recs + nrec + (x_rec[t] - c_rec[t]));
                }
            SIMT_END
            SWC_LP(7)
            // the last match start of the round: in the last lane that took a sequence
            SIMT_BEGIN(t, N) pb[t] = (uint32_t)t < nv && c_rec[t] != 0; SIMT_END
            const uint64_t m_seq = simt::wave_ballot<N>(pb);
            const int last = 63 - (int)clz64(m_seq);
            SIMT_BEGIN(t, N) x_lit[t] = (x_out[t] - c_out[t]) + c_lms[t]; SIMT_END
            last_match_start = (int64_t)(pos + simt::wave_read<N>(x_lit, last));
            pos += tot_out;
            nlit += tot_lit;
            nrec += tot_rec;
            sequences += tot_rec;
            SWC_LZ4_STAT(3, tot_rec);
            ip = B + simt::wave_read<N>(endp, (int)nv - 1);
            if (R4) s_pred = ip;
            // anything invalid / the end of the block / a sequence no round takes (a long literal run: the next round would stage
            // 8 KiB to find out in its first lane): the checked step
            if (stop_flags & (kLzFail | kLzTail | kLzLong)) return;
        }
    }

    SWC_D int run(uint8_t* stage) {
        // the rounds append records unchecked: the workspace must hold what a block of this capacity can need
        const bool fast_ok = stage != nullptr && prov != nullptr && (size_t)max_rec >= (R8 ? lzr::max_records8(cap) : lzr::max_records(cap));
        int result = SWC_OK;
        for (;;) {
            if (fast_ok && pos < cap) sync_rounds(stage);
            const int st = careful_step();
            if (st == -1) break;
            if (st) { result = st; break; }
        }
        return result;
    }
};

// One wavefront = one job (blocks WITHOUT a dictionary prefix; those with one stay on lz4_lane.h).
template <int W, int RM = 0>
SWC_D void lz4_parse_job(Job& job, uint8_t* ws, size_t ws_bytes, int lane, uint8_t* stage, uint64_t* prof = nullptr) {
    constexpr bool R8 = RM == 1, R4 = RM == 2;
    Parser<W, RM> ps;
    ps.w.lane = lane;
    ps.iw = stage + kScratchWin;
    ps.iw_hi = 0;
    ps.iw_next = 0;
    ps.iw_pf = false;
    ps.rbuf = (uint32_t*)(stage + kScratchRec);
    ps.lbuf = stage + kScratchLit;

    ps.rb_n = ps.lb_n = 0;
    ps.in = (gcptr)job.in;
    ps.n = job.in_len;
    ps.cap = job.out_cap;
    ps.ip = ps.pos = ps.nlit = ps.sequences = 0;
    ps.last_match_start = -1;
    ps.nrec = 0;
    const size_t lo = ws ? lzr::lit_offset(ws_bytes, job.out_cap) : 0;
    ps.recs = (SWC_AS_GLOBAL uint32_t*)(ws + sizeof(lzr::StreamHeader));
    size_t rec_end = lo;   // the wave's scratch rows sit between the record list and the literal stream, if the area has room for them
    if (R8) {
        // no literal stream: header | eight-byte records | scratch rows at the END of the area (the records take what the literal
        // stream would have taken: a block needs up to cap / 4 of them)
        if (lo != 0 && ws_bytes >= sizeof(lzr::StreamHeader) + 256 + lzr::kProvBytes) {
            rec_end = (ws_bytes - lzr::kProvBytes) & ~(size_t)15;
            ps.prov = (gptr)(ws + rec_end);
        }
    } else if (lo >= sizeof(lzr::StreamHeader) + 256 + lzr::kProvBytes) {
        rec_end = (lo - lzr::kProvBytes) & ~(size_t)15;
        ps.prov = (gptr)(ws + rec_end);
    }
    ps.max_rec = rec_end > sizeof(lzr::StreamHeader) ? (uint32_t)((rec_end - sizeof(lzr::StreamHeader)) / (R8 ? 8 : 4)) : 0u;
    ps.lits = (gptr)(ws + lo);
    if (R4 && lo != 0) {   // the anchors take the place of the literal stream
        ps.anc = (SWC_AS_GLOBAL uint32_t*)(ws + lo);
        ps.max_anc = (uint32_t)(lzr::lit_bytes(job.out_cap) / 8);
    }
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    ps.tlast = __builtin_readcyclecounter();
#endif
    int st = lo == 0 ? SWC_E_NEED_WORKSPACE : (RM != 0 && job.in_len > 0xFFFFFFF0ull) ? SWC_E_INVALID_ARGUMENT /* literal offsets are 32 bits */ : ps.run(stage);
    ps.flush();
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    { const uint64_t t_ = __builtin_readcyclecounter(); ps.pacc[4] += t_ - ps.tlast; }
    if (prof && lane == 0) for (int k = 0; k < 8; k++) prof[k] = ps.pacc[k];
#else
    (void)prof;
#endif
    if (ps.nrec > ps.max_rec) {
        st = SWC_E_NEED_WORKSPACE;
        ps.nrec = ps.max_rec;
    }
    if (R4 && ps.nanc > ps.max_anc) {   // (cannot happen for an area sized by swc_batch_workspace_bytes: an anchor stands for 128 bytes of output)
        st = SWC_E_NEED_WORKSPACE;
        ps.nanc = ps.max_anc;
    }
    if (st == SWC_OK && ps.pos > ps.cap) st = SWC_E_CAPACITY;
    if (lane == 0) {
        if (ws && ws_bytes >= sizeof(lzr::StreamHeader)) {
            SWC_AS_GLOBAL lzr::StreamHeader* h = (SWC_AS_GLOBAL lzr::StreamHeader*)ws;
            h->nrec = ps.nrec;
            h->pad0 = R4 ? ps.nanc : 0u;   // (R4: the number of anchors)
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
