// inflate_sync.h -- Deflate entropy decode (phase 1), one compressed stream per WAVEFRONT, 64 lanes decoding 64
// sub-chunks of the stream at once.
//
// Replaces the symbol loop of Deflate.decompress(_ bitReader:) (reference Sources/Deflate/Deflate.swift:171-236) and the
// code-length section of a dynamic header (:86-167).  Huffman decoding is a serial chain -- where a code starts is known
// only when the previous one has been decoded -- but a decoder started at a WRONG bit offset falls into step with the
// true symbol sequence after a few symbols (it lands on a new position after every code; about one position in
// thirteen is a true symbol start).  A wavefront therefore cuts the next kSyncRound bytes of the stream into 64
// sub-chunks of kSyncChunk bytes, one per lane, and iterates:
//
//   pass 1      lane 0 starts at the true position, every other lane at its sub-chunk boundary; all decode (counting
//               only) until they cross the end of their sub-chunk and note where they ended;
//   pass 2..    a lane whose start differs from the end of its left neighbour decodes again from there.  After pass 2
//               nearly every lane is on the true sequence (its garbage decode had synchronised inside the neighbour's
//               sub-chunk); the loop runs until the chain of (start == left neighbour's end) reaches the end-of-block
//               symbol or lane 63 -- each pass makes at least one more lane final, so it terminates;
//   scan        exclusive prefix sums of the per-lane counts (literals, records, output bytes);
//   emit        every lane decodes its sub-chunk once more and writes its literals to the dense literal stream and one
//               record per match to the record list (lz_resolve.h) at exact offsets.  A sub-chunk closes its trailing
//               literals with a literal-only record, so that no lane needs the literal run of its neighbour.
//
// All 64 lanes share ONE set of tables in LDS: direct lookup tables (10 / 8 bits) whose entries carry code length,
// extra bits, kind and base value; a code longer than the tables finds its length by comparing the bit-reversed window
// against the canonical limits of the lengths 9..15 and its symbol in the sorted symbol array.  The input of a round is
// staged in LDS with coalesced loads; a lane walks its sub-chunk through a two-dword register window whose next dword is
// read from LDS every iteration, needed or not.  Sub-chunks are 17 dwords long: an odd stride keeps the 64 lanes on 64
// different LDS banks without padding, and everything in LDS together is exactly 10 KB (16 waves per CU).
//
// Anything the fast path does not want to decide -- an unassigned or over-subscribed code, symbols 286/287, distance
// symbols 30/31, a distance beyond the output, the end of the input inside a symbol, the capacity inside a round -- makes
// it return WITHOUT committing the round; the caller then runs the fully checked one-symbol step of inflate_lane.h,
// which carries the reference's error taxonomy.  Headers, stored blocks and static tables also stay with inflate_lane.h;
// the code-length section of a dynamic header is decoded here through a 128-entry table, and its histogram / counting
// sort run on all lanes.
#ifndef SWC_INFLATE_SYNC_H
#define SWC_INFLATE_SYNC_H

#include "inflate_lane.h"
#include "simt.h"

namespace swc {
namespace inflate {

constexpr int kSyncLitBits = 10, kSyncDistBits = 8;
// Input bytes per lane and round: a whole number of dwords and an ODD number of them (17), so that the lanes, which start
// a pass at the same offset of their sub-chunks, read 64 different LDS banks without any padding of the staged input.
// With 68 bytes the wave's LDS is EXACTLY 10,240 bytes = 16 waves per CU: the kernel's speed is proportional to the waves
// a CU holds (measured by padding the LDS: 12 waves 13.7 ms, 9 waves 17.6 ms, 7 waves 22.1 ms; 16 waves 12.05 ms), and
// shorter sub-chunks cost more in rounds and lane imbalance than further waves bring (60 bytes: 12.85 ms, 52: 13.75 ms).
constexpr uint32_t kSyncChunk = SWC_SYNC_CHUNK;
constexpr uint32_t kSyncRound = 64u * kSyncChunk;
constexpr uint32_t kSyncStage = (kSyncRound + 32u + 15u) & ~15u;    // + what the last lane may read past its sub-chunk (a code of <= 48 bits, then the window's two dwords and the next one: < 28 bytes)
constexpr uint32_t kSyncStageLds = kSyncStage;
constexpr uint32_t kEntInvalid = 0x80000000u;        // bit 31 = STOP: the decode loops end at this entry -- the end-of-block symbol (with kEntEob), or a code
                                                      // of the set that is not a symbol the fast path takes (without; such an entry counts no bits)
constexpr uint32_t kPosFail = 0xFFFFFFFFu;
static_assert(kSyncChunk % 4 == 0 && kSyncChunk >= 36, "sub-chunks are whole dwords");

// Table entry, laid out for a decode loop without branches per kind:
//   [0:4] bits the symbol takes (code + extra)   [5] end of block   [6:9] code length   [10] length symbol: the next code is
//   a distance (the bit IS the offset of the distance table in `lut`)   [11:14] extra bits   [15] literal
//   [16:30] base value   [31] stop (end of block, or invalid).   0: no entry (long code).
constexpr uint32_t kEntEob = 1u << 5, kEntLen = 1u << kSyncLitBits, kEntLit = 1u << 15;
constexpr uint32_t kEntClenShift = 6, kEntExtShift = 11;
static_assert(kSyncLitBits == 10, "the entry layout keeps bit 10 for the length flag");
// kind, lit/len table: 1 literal, 2 length, 3 end of block; distance table: 0 distance.
SWC_HD uint32_t make_entry(uint32_t clen, uint32_t ext, uint32_t kind, uint32_t value) {
    return (clen + ext) | (kind == 1 ? kEntLit : kind == 2 ? kEntLen : kind == 3 ? kEntEob | kEntInvalid : 0u) | (clen << kEntClenShift) | (ext << kEntExtShift) | (value << 16);
}

// The entry of lit/len symbol `sym` (0..287) / distance symbol `sym` (0..31) with a code of `d` bits
// (Deflate+Constants.swift: lengthBase / distanceBase as arithmetic).
SWC_HD uint32_t entry_of_symbol(bool dist, uint32_t sym, uint32_t d) {
    if (!dist) {
        if (sym < 256) return make_entry(d, 0, 1, sym);
        if (sym == 256) return make_entry(d, 0, 3, 0);
        if (sym > 285) return kEntInvalid;       // 286, 287: the checked step reports wrongSymbol
        const uint32_t s = sym - 257u;
        const uint32_t e = s < 8 || s == 28 ? 0u : (s >> 2) - 1u;
        const uint32_t base = s < 8 ? 3u + s : s == 28 ? 258u : 3u + ((4u + (s & 3u)) << e);
        return make_entry(d, e, 2, base);
    }
    if (sym > 29) return kEntInvalid;            // 30, 31: wrongSymbol
    const uint32_t e = sym < 4 ? 0u : (sym >> 1) - 1u;
    const uint32_t base = sym < 4 ? 1u + sym : 1u + ((2u + (sym & 1u)) << e);
    return make_entry(d, e, 0, base);
}

// The canonical tables of inflate_lane.h (struct Table) kept in LDS: per length d the left-justified code limit, the slot
// word (sorted index of the first code - first code | index of the first symbol >= 256 << 16) and the sorted index of the
// first code; the number of codes; the over-subscription flag.  Lit/len alphabet at kAuxLit, distance alphabet at kAuxDist.
constexpr int kAuxLim = 0, kAuxSlot = 16, kAuxStart = 32, kAuxCount = 48 /* == start[16] */, kAuxOver = 49, kAuxTable = 52;
constexpr int kAuxLit = 0, kAuxDist = kAuxTable, kAuxWords = 2 * kAuxTable;

struct SyncLds {   // 10,240 bytes: see kSyncChunk
    uint32_t syms[kWordsPerLane];                                   // the sorted symbol arrays of inflate_lane.h (LaneLds{syms, 1})
    uint32_t lut[(1 << kSyncLitBits) + (1 << kSyncDistBits)];      // direct tables: lit/len, then distance
    uint32_t aux[kAuxWords];                                        // the canonical tables (see kAux*)
#ifdef SWC_SYNC_LDS_PAD
    uint8_t occupancy_experiment_pad[SWC_SYNC_LDS_PAD];             // (tools/gpu_chunk_sweep.sh: fewer waves per CU, nothing else changed)
#endif
    alignas(16) uint8_t stage[kSyncStageLds];                       // staged input of a round; header build: code lengths, counters, code-length table
};
#if SWC_SYNC_CHUNK == 68
static_assert(sizeof(SyncLds) <= 10240, "16 waves per CU: the wave's LDS must stay within 160 KB / 16");
#endif
// header scratch inside `stage`
constexpr uint32_t kHdrLens = 0;       // 320 bytes: code length of symbol s
constexpr uint32_t kHdrCnt = 320;      // 48 words: codes per length (lit/len, of those symbols < 256, distance)
constexpr uint32_t kHdrClLut = 512;    // 128 bytes: code-length code, len | symbol << 3 (0xFF: no code)

enum { kSyncEob = 0, kSyncBail = 1, kSyncBailCap = 2 };
#if defined(SWC_HOST_EMULATION)
// statistics of the emulated decoder (tests, tools/sync_stats.py): rounds committed, bails, lane-passes, symbol iterations
inline uint64_t g_sync_stats[8];
#define SWC_SYNC_STAT(i, n) (g_sync_stats[i] += (n))
// wave-steps: a pass takes as long as its busiest lane.  g_sync_wave[k] adds up, per pass of kind k (0 walk, 1 count,
// 2 emit), the largest number of code iterations any of the 64 lanes ran (lanes are emulated in ascending order here).
inline uint64_t g_sync_iters = 0, g_sync_wave[4] = {0, 0, 0, 0}, g_sync_passmax = 0;
#define SWC_SYNC_ITER() (g_sync_iters++)
#define SWC_SYNC_LANE_BEGIN(t) if ((t) == 0) g_sync_passmax = 0; const uint64_t it0_ = g_sync_iters;
#define SWC_SYNC_LANE_END(t, n, k) { if (g_sync_iters - it0_ > g_sync_passmax) g_sync_passmax = g_sync_iters - it0_; if ((t) == (n) - 1) g_sync_wave[k] += g_sync_passmax; }
#else
#define SWC_SYNC_STAT(i, n) ((void)0)
#define SWC_SYNC_ITER() ((void)0)
#define SWC_SYNC_LANE_BEGIN(t)
#define SWC_SYNC_LANE_END(t, n, k)
#endif
enum { kFlagEob = 1u, kFlagFail = 2u, kFlagTrap = 4u };
// profile builds (-DSWC_PROFILE): cycles per part of one stream -- 0 header, 1 tables, 2 staging, 3 decode passes, 4 chain
// logic + scans, 5 copy / emit, 6 checked steps / rest; 7 rounds, 8 passes, 9 walk pass
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
struct SyncProf {
    uint64_t acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast = 0;
};
#define SWC_SP(pp, k) { const uint64_t t_ = __builtin_readcyclecounter(); (pp).acc[k] += t_ - (pp).tlast; (pp).tlast = t_; }
#define SWC_SPC(pp, k, n) ((pp).acc[k] += (n))
#else
struct SyncProf {};
#define SWC_SP(pp, k)
#define SWC_SPC(pp, k, n)
#endif

SWC_D void lds_atomic_inc(uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    (*p)++;
#endif
}

// ---- tables ------------------------------------------------------------------------------------------------------
// Code.swift:23-37 per length, from the per-length counts: cnt[d] codes of length d, of which lo[d] are symbols < 256
// (lo == nullptr: none).  Every lane computes and stores the same values.
SWC_D void table_from_counts(const uint32_t* cnt, const uint32_t* lo, uint32_t* tb) {
    uint32_t v = 0, off = 0, over = 0;
    tb[kAuxLim] = 0; tb[kAuxSlot] = 0; tb[kAuxStart] = 0;
#pragma unroll 1
    for (int d = 1; d <= 15; d++) {
        const uint32_t c = cnt[d];
        tb[kAuxLim + d] = (v + c) << (15 - d);
        if (c != 0 && v + c > (1u << d)) over = 1;
        tb[kAuxSlot + d] = ((off - v) & 0xFFFFu) | ((off + (lo ? lo[d] : 0u)) << 16);
        tb[kAuxStart + d] = off;
        off += c;
        v = (v + c) << 1;
    }
    tb[kAuxCount] = off;
    tb[kAuxOver] = over;
}
// One symbol of the lit/len (LIT) or distance alphabet from the LDS form of the tables, with the reference's
// semantics for every code set (Lane::decode_sym<LIT, true>: DecodingTree.swift:36-50 over the heap Code.swift:15-39
// builds -- for an over-subscribed set the shallowest occupied node wins, the last writer of a node wins).
// Returns the symbol or -1 (symbolNotFound: unassigned path, or the code runs past the end of the input).
template <bool LIT>
SWC_D int decode_sym_lds(BitReader& br, const SyncLds* sl) {
    const uint32_t* tb = sl->aux + (LIT ? kAuxLit : kAuxDist);
    const uint32_t c15 = brev32(br.peek32()) >> 17;
    uint32_t len = 16, idx = 0;
    if (tb[kAuxOver] == 0) {
        len = 1;
#pragma unroll 1
        for (int d = 1; d <= 15; d++) len += c15 >= tb[kAuxLim + d] ? 1u : 0u;
        if (len > 15) return -1;
        idx = (tb[kAuxSlot + len] + (c15 >> (15 - len))) & 0xFFFFu;
    } else {
#pragma unroll 1
        for (uint32_t d = 1; d <= 15; d++) {
            const uint32_t st = tb[kAuxStart + d], cnt = (tb[kAuxStart + d + 1] - st) & 0xFFFFu;
            const uint32_t fst = d == 1 ? 0u : tb[kAuxLim + d - 1] >> (15 - d);
            const uint32_t k0 = ((c15 >> (15 - d)) - fst) & ((1u << d) - 1u);
            if (k0 < cnt) {
                len = d;
                idx = st + k0 + (((cnt - 1u - k0) >> d) << d);
                break;
            }
        }
        if (len > 15) return -1;
    }
    if (len > br.bc) return -1;  // DecodingTree.swift:39 -- ran out of bits before reaching a leaf
    br.consume(len);
    const LaneLds l{const_cast<uint32_t*>(sl->syms), 1};
    uint32_t sym = *sym_ptr(l, LIT ? W_LIT_SYM : W_DIST_SYM, idx);
    if (LIT) sym |= idx >= (tb[kAuxSlot + len] >> 16) ? 256u : 0u;
    return (int)sym;
}

// One symbol with every check of the reference (Deflate.swift:171-236) -- Lane::careful_step over the LDS tables, so
// that the wave kernel never holds the register form of the tables.  Returns SWC_OK to continue, -1 at the
// end-of-block symbol, or the error.
SWC_D int careful_step_lds(Lane& ln, const SyncLds* sl) {
    BitReader& br = ln.br;
    br.refill();
    const int sym = decode_sym_lds<true>(br, sl);
    if (sym < 0) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :175
    if (sym < 256) {
        ln.put_byte((uint8_t)sym);
        return SWC_OK;
    }
    if (sym == 256) return -1;
    if (sym > 285) return SWC_E_DEFLATE_WRONG_SYMBOL;  // :233
    const uint32_t s = (uint32_t)sym - 257u;
    uint32_t length;
    if (s < 8) {
        length = 3 + s;
    } else if (s == 28) {
        length = 258;
    } else {
        const uint32_t e = (s >> 2) - 1;  // :188
        if (br.bc < e) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :192
        length = 3 + ((4 + (s & 3)) << e) + br.bits(e);  // Constants.lengthBase
    }
    br.refill();
    const int dc = decode_sym_lds<false>(br, sl);
    if (dc < 0) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :199
    if (dc > 29) return SWC_E_DEFLATE_WRONG_SYMBOL;     // :201
    uint32_t distance;
    if (dc < 4) {
        distance = 1 + (uint32_t)dc;
    } else {
        const uint32_t e = ((uint32_t)dc >> 1) - 1;  // :206
        if (br.bc < e) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :208
        distance = 1 + ((2 + ((uint32_t)dc & 1)) << e) + br.bits(e);  // Constants.distanceBase
    }
    // :216-221 out[count - distance] with distance > count is a Swift trap (App. A6)
    if ((uint64_t)distance > ln.pos) return SWC_E_REF_TRAP;
    ln.emit_match(length, distance);
    return SWC_OK;
}

// Direct tables from the canonical tables and the sorted symbol arrays in LDS.  All lanes.
SWC_D void sync_build_luts(SyncLds* sl) {
    constexpr int N = kWave;
    const LaneLds l{sl->syms, 1};
    SIMT_BEGIN(t, N)
        for (int i = t; i < (1 << kSyncLitBits) + (1 << kSyncDistBits); i += N) sl->lut[i] = 0;
    SIMT_END_WAVE
    SIMT_BEGIN(t, N)
#pragma unroll 1
        for (int tbl = 0; tbl < 2; tbl++) {
            const uint32_t* tb = sl->aux + (tbl ? kAuxDist : kAuxLit);
            const uint32_t n_sym = tb[kAuxCount], n_max = tbl ? 32u : 288u;
            const uint32_t lut_bits = tbl ? (uint32_t)kSyncDistBits : (uint32_t)kSyncLitBits;
            uint32_t* lut = sl->lut + (tbl ? (1 << kSyncLitBits) : 0);
            for (uint32_t j = (uint32_t)t; j < n_sym && j < n_max; j += (uint32_t)N) {
                uint32_t d = 1;
#pragma unroll 1
                for (uint32_t q = 2; q <= 15; q++) if (j >= tb[kAuxStart + q]) d = q;   // the length whose index range holds j
                const uint32_t fst = d == 1 ? 0u : tb[kAuxLim + d - 1] >> (15 - d);
                const uint32_t code = fst + (j - tb[kAuxStart + d]);
                const uint32_t rev = brev32(code) >> (32 - d);
                uint32_t sym;
                if (tbl == 0) {
                    sym = *sym_ptr(l, W_LIT_SYM, j);
                    if (j >= (tb[kAuxSlot + d] >> 16)) sym |= 256u;
                } else {
                    sym = *sym_ptr(l, W_DIST_SYM, j);
                }
                const uint32_t entry = entry_of_symbol(tbl != 0, sym, d);
                if (d <= lut_bits)
                    for (uint32_t m = rev; m < (1u << lut_bits); m += 1u << d) lut[m] = entry;
            }
        }
    SIMT_END_WAVE
}

// From the code lengths lens[0 .. literals + distances) in the header scratch: histogram, canonical tables, the
// (length, symbol)-sorted symbol arrays of inflate_lane.h, per-symbol entries and direct tables.  All lanes.
SWC_D void sync_tables_from_lengths(SyncLds* sl, int literals, int distances) {
    using simt::PT;
    constexpr int N = kWave;
    const uint8_t* lens = sl->stage + kHdrLens;
    uint32_t* cnt = (uint32_t*)(sl->stage + kHdrCnt);
    const LaneLds l{sl->syms, 1};
    const int total = literals + distances;
    // codes per length of the lit/len alphabet, of those symbols < 256, and of the distance alphabet
    SIMT_BEGIN(t, N)
        for (int s = t; s < total; s += N) {
            const uint32_t v = lens[s];
            if (v) {
                if (s < literals) { lds_atomic_inc(&cnt[v]); if (s < 256) lds_atomic_inc(&cnt[16 + v]); }
                else lds_atomic_inc(&cnt[32 + v]);
            }
        }
    SIMT_END_WAVE
    table_from_counts(cnt, cnt + 16, sl->aux + kAuxLit);
    table_from_counts(cnt + 32, nullptr, sl->aux + kAuxDist);
    simt::wave_fence();
    // counting sort by (length, symbol): 64 symbols at a time, rank within the length by ballot; cnt[d] / cnt[32 + d]
    // become the running positions
    SIMT_BEGIN(t, N)
        if (t < 16) { cnt[t] = sl->aux[kAuxLit + kAuxStart + t]; cnt[32 + t] = sl->aux[kAuxDist + kAuxStart + t]; }
    SIMT_END_WAVE
    PT<uint32_t, N> v;
    PT<bool, N> p;
#pragma unroll 1
    for (int g = 0; g * N < literals + N; g++) {   // the groups of the lit/len alphabet, then the one group of the distance alphabet
        const bool is_dist = g * N >= literals;
        const int s0 = is_dist ? literals : g * N, s_end = is_dist ? total : literals;
        SIMT_BEGIN(t, N) v[t] = s0 + t < s_end ? (uint32_t)lens[s0 + t] : 0u; SIMT_END
#pragma unroll 1
        for (uint32_t d = 1; d <= 15; d++) {
            SIMT_BEGIN(t, N) p[t] = v[t] == d; SIMT_END
            const uint64_t m = simt::wave_ballot<N>(p);
            if (m == 0) continue;
            uint32_t* rp = &cnt[(is_dist ? 32u : 0u) + d];
            const uint32_t r0 = *rp;
            SIMT_BEGIN(t, N)
                if (p[t]) *sym_ptr(l, is_dist ? W_DIST_SYM : W_LIT_SYM, r0 + (uint32_t)simt::popc64(m & ((1ull << t) - 1ull))) = (uint8_t)(s0 + t - (is_dist ? literals : 0));
            SIMT_END_WAVE
            *rp = r0 + (uint32_t)simt::popc64(m);
            simt::wave_fence();
        }
    }
    sync_build_luts(sl);
}

// ---- one sub-chunk -----------------------------------------------------------------------------------------------
struct ChunkOut {
    uint32_t end;     // bit position (relative to the round base) just past the last symbol taken; kPosFail: no valid decode
    uint32_t nlit, nrec, nout;
    uint32_t flags;
};

SWC_HD uint32_t funnel32(uint32_t hi, uint32_t lo, uint32_t sh) {  // bits [sh, sh + 32) of hi:lo, sh < 32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}
SWC_HD uint32_t bfe32(uint32_t v, uint32_t off, uint32_t width) {   // width 0..16, off + width <= 32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    return width == 0 ? 0u : (v >> off) & ((1u << width) - 1u);
#endif
}

// A code longer than the direct tables (lit/len: more than 10 bits, distance: more than 8): its length from the canonical
// limits of the lengths 9..15 in LDS (`aux`; reads the compiler pairs up) with compares, its symbol from the sorted symbol
// array, its entry by arithmetic -- no per-symbol entry table, which would cost the wave 1.3 KB of LDS.  The wave pays for
// this path whenever ANY lane meets such a code (about one iteration in seven on text), so the alphabet only selects a base
// address.
struct LongCodes {
    SWC_D void load(const SyncLds*) {}
    // the entry of the code that starts `bits` (state: 0 lit/len, anything else distance); kEntInvalid if there is none
    SWC_D uint32_t lookup(const SyncLds* sl, uint32_t bits, uint32_t state) const {
        const uint32_t c15 = brev32(bits) >> 17;
        const bool st = state != 0;
        const uint32_t* tb = sl->aux + (st ? kAuxDist : kAuxLit);
        // the limits do not decrease with the length: count how many the window reaches, keep the slot word of that length
        uint32_t len = 9u, slw = tb[kAuxSlot + 9];
#pragma unroll
        for (int d = 9; d < 15; d++) {
            const bool g = c15 >= tb[kAuxLim + d];
            len += g ? 1u : 0u;
            slw = g ? tb[kAuxSlot + d + 1] : slw;
        }
        len += c15 >= tb[kAuxLim + 15] ? 1u : 0u;
        const uint32_t lenc = len > 15 ? 15u : len;
        uint32_t j = (slw + (c15 >> (15 - lenc))) & 0xFFFFu;
        const uint32_t jmax = st ? 31u : 287u;
        j = j > jmax ? jmax : j;
        const LaneLds l{const_cast<uint32_t*>(sl->syms), 1};
        uint32_t sym = *sym_ptr(l, st ? W_DIST_SYM : W_LIT_SYM, j);
        if (!st && j >= (slw >> 16)) sym |= 256u;
        const uint32_t e = entry_of_symbol(st, sym, lenc);
        return len > 15 ? kEntInvalid : e;
    }
};

// MODE 0: count.  MODE 1: emit literals and records.  MODE 2: check distances only (output beyond the capacity).
// MODE 3: walk (where does the decode end?).  BIG: literal runs of more than lzr::kLitRunMax bytes in front of a match
// get a record of their own (only possible in a sub-chunk that holds more than that many literals: the caller picks).
// CHK: symbols may run past the end of the input.
// Decodes from bit `start` until a lit/len symbol would begin at or beyond `chunk_end`, or the end-of-block symbol.
template <int MODE, bool BIG, bool CHK>
SWC_D void decode_chunk(const SyncLds* sl, const LongCodes lc, uint32_t start, uint32_t chunk_end, uint32_t in_bits, gptr lit_dst,
                        SWC_AS_GLOBAL uint32_t* rec_dst, uint64_t out_pos0, ChunkOut& r) {
    const uint8_t* stg = sl->stage;
    uint32_t wa = (start >> 5) << 2, bp = start & 31u;    // the window: dwords at byte offset wa and wa + 4 of the staged input, bit bp of it is next
    uint32_t d0 = *(const uint32_t*)(stg + wa), d1 = *(const uint32_t*)(stg + wa + 4);
    uint32_t tsel = 0, tmsk = (1u << kSyncLitBits) - 1u;   // the table of the next code: dword offset into `lut` and index mask
    uint32_t plen = 0, run = 0, nlit = 0, nrec = 0, nout = 0, flags = 0;
    uint32_t lb = 0, lbn = 0;                              // literals on their way to the literal stream, four per store
    const uint32_t room = out_pos0 > 0x40000000ull ? 0x40000000u : (uint32_t)out_pos0;   // output in front of the sub-chunk, as far as a distance can reach
    // The loop is bound by the number of vector instructions it issues (measured: the SIMDs' VALU pipes are busy 80 % of
    // the kernel), so the body is one straight line of few selects -- no branch per kind of symbol, the window's next dword
    // is read whether the window moves on or not, everything rare (long code, end of block, anything invalid) sits behind
    // ONE test.
    for (;;) {
        const uint32_t posb = (wa << 3) + bp;
        if ((uint32_t)(tsel == 0) & (uint32_t)(posb >= chunk_end)) break;   // (bitwise: one compare pair, no nested mask region)
        SWC_SYNC_STAT(4 + (MODE == 3 ? 0 : MODE), 1);   // code iterations per mode
        SWC_SYNC_ITER();
        const uint32_t bits = funnel32(d1, d0, bp);
        uint32_t e = sl->lut[(bits & tmsk) | tsel];
        const uint32_t nx = *(const uint32_t*)(stg + wa + 8);
        if (e == 0) { SWC_SYNC_STAT(7, 1); e = lc.lookup(sl, bits, tsel); }   // a code longer than the direct table (or no code at all)
        const uint32_t n = e & 31u;
        if ((e & kEntInvalid) || (CHK && posb + n > in_bits)) {
            // not a symbol the fast path takes / the symbol runs past the end of the input / the end of the block
            if (!(e & kEntEob) || (CHK && posb + n > in_bits)) flags |= kFlagFail;
            else { bp += n; flags |= kFlagEob; }
            break;
        }
        const bool was_dist = tsel != 0;
        if (MODE != 3) {
            const uint32_t is_lit = (e >> 15) & 1u;
            const uint32_t val = ((e >> 16) & 0x7FFFu) + bfe32(bits, (e >> kEntClenShift) & 15u, (e >> kEntExtShift) & 15u);
            if (MODE != 0 && was_dist && val > room + nout) { flags |= kFlagTrap; break; }
            const bool big = BIG && was_dist && run > lzr::kLitRunMax;
            if (MODE == 1) {
                if (is_lit) {
                    lb = funnel32(val, lb, 8);            // the new byte enters at the top: after four the dword is in stream order
                    if (++lbn == 4) { store_u32(lit_dst, lb); lit_dst += 4; lbn = 0; }
                }
                if (was_dist) {
                    if (big) *rec_dst++ = lzr::make_lits(run);
                    *rec_dst++ = lzr::make_match(big ? 0u : run, plen, val);
                }
            }
            nlit += is_lit;
            nout += is_lit + (was_dist ? plen : 0u);
            nrec += was_dist ? (big ? 2u : 1u) : 0u;
            run = was_dist ? 0u : run + is_lit;
            plen = (e & kEntLen) ? val : plen;
        }
        tsel = e & kEntLen;                                                 // 0, or 1 << kSyncLitBits: the distance table follows the lit/len table
        tmsk = ((1u << kSyncLitBits) - 1u) >> ((tsel >> kSyncLitBits) * (uint32_t)(kSyncLitBits - kSyncDistBits));   // 10-bit index, or 8-bit
        bp += n;
        const bool sh = bp >= 32;
        wa += sh ? 4u : 0u;
        bp &= 31u;
        d0 = sh ? d1 : d0;
        d1 = sh ? nx : d1;
    }
    if (MODE != 3 && run > 0) {   // the sub-chunk closes its literal run itself
        nrec++;
        if (MODE == 1) *rec_dst++ = lzr::make_lits(run);
    }
    if (MODE == 1) {
        for (uint32_t i = 0; i < lbn; i++) lit_dst[i] = (uint8_t)(lb >> (8 * (4u - lbn + i)));   // the pending bytes sit at the top
    }
    r.end = (flags & kFlagFail) ? kPosFail : (wa << 3) + bp;
    r.nlit = nlit; r.nrec = nrec; r.nout = nout;
    r.flags = flags;
}

// ---- one sub-chunk, decoded ONCE into the round's scratch ------------------------------------------------------------------
// The count pass and the emit pass of a round in one: the lane decodes its sub-chunk from `start` and writes its literals
// and records to its column of the stream's scratch area (lz_resolve.h: rows across the lanes, sized for the worst case, so
// the loop needs no bounds test), counting as it goes; the round then scans the counts and COPIES every lane's piece to its
// final offset (copy_prov) instead of decoding a third time.  What the lane cannot know yet -- the output position of its
// sub-chunk -- enters only through `need`: the largest (distance - output bytes of the sub-chunk in front of the match),
// checked after the scan.  Literals gather in a 64-bit accumulator (the new byte enters at the top) and leave as a group of
// eight; the group store and the record store share ONE conditional region of the loop, each aimed at row 0 when it is not
// its turn.  The loop has no variant for the end of the input: it stops at the last bit, and a symbol that ran past it is
// caught by the final position.  `run0`: literals in front of the sub-chunk that no record covers yet (lane 0 of the first
// round of a block).  A sub-chunk that ends at the end-of-block symbol leaves its trailing literals uncovered (`tail`), every
// other one closes them with a literal-only record.
struct ProvOut {
    uint32_t end, nlit, nrec, nout, flags, tail;
    int32_t need;
};
enum { kFlagSlow = 8u };   // the sub-chunk needs the general path (a literal run that no single record can carry)
constexpr uint32_t kProvRecRow = 64u * 4u, kProvLitRow = 64u * 8u;   // bytes from one row of the scratch to the next

// The two loops below have ONE exit, at the top: the end-of-block symbol and anything the fast path does not take raise their
// flag and pull the end of the sub-chunk to zero, so that the next test at the top leaves (their side effects are neutral: an
// end-of-block entry is neither literal nor length, and a lane that raised the fail flag is discarded).  A loop with several
// exits costs the wavefront a dozen scalar instructions of mask bookkeeping per iteration.
// `seen`: the OR of all entries a loop took.  Only the entry that stopped it has bit 31, only the end-of-block entry bit 5.
SWC_HD uint32_t flags_of_seen(uint32_t seen) { return (seen & kEntEob) ? kFlagEob : (seen & kEntInvalid) ? kFlagFail : 0u; }
SWC_HD uint32_t sext_bit31(uint32_t e) { return (uint32_t)((int32_t)e >> 31); }   // all ones for a stop entry

// Where does a decode from `start` end?  (The walk pass: no counting.)  Returns the bit position just past the last symbol
// taken, or kPosFail.
SWC_D uint32_t walk_chunk(const SyncLds* sl, const LongCodes lc, uint32_t start, uint32_t chunk_end, uint32_t in_bits) {
    const uint8_t* stg = sl->stage;
    uint32_t wa = (start >> 5) << 2, bp = start & 31u;
    uint32_t d0 = *(const uint32_t*)(stg + wa), d1 = *(const uint32_t*)(stg + wa + 4);
    uint32_t tsel = 0, tmsk = (1u << kSyncLitBits) - 1u, seen = 0;
    if (chunk_end > in_bits) chunk_end = in_bits;          // (the zero fill behind the input is not worth decoding)
    if (start >= in_bits) { seen = kEntInvalid; chunk_end = 0; }
    if (start < chunk_end) {   // (tested at the bottom: one mask update and one branch per iteration)
        bool go;
        do {
            SWC_SYNC_STAT(4, 1);
            SWC_SYNC_ITER();
            const uint32_t bits = funnel32(d1, d0, bp);
            uint32_t e = sl->lut[(bits & tmsk) | tsel];
            const uint32_t nx = *(const uint32_t*)(stg + wa + 8);
            if (e == 0) { SWC_SYNC_STAT(7, 1); e = lc.lookup(sl, bits, tsel); }
            seen |= e;
            chunk_end &= ~sext_bit31(e);
            const bool is_len = (e & kEntLen) != 0u;
            tsel = e & kEntLen;
            tmsk = is_len ? (1u << kSyncDistBits) - 1u : (1u << kSyncLitBits) - 1u;
            bp += e & 31u;
            const bool sh = bp >= 32;
            wa += sh ? 4u : 0u;
            bp &= 31u;
            d0 = sh ? d1 : d0;
            d1 = sh ? nx : d1;
            go = is_len || (wa << 3) + bp < chunk_end;
        } while (go);
    }
    const uint32_t endb = (wa << 3) + bp;
    return flags_of_seen(seen) == kFlagFail || endb > in_bits ? kPosFail : endb;
}

// prov: the stream's scratch; lane: my column of its rows
SWC_D void decode_chunk_prov(const SyncLds* sl, const LongCodes lc, uint32_t start, uint32_t chunk_end, uint32_t in_bits, gptr prov,
                             uint32_t lane, uint32_t run0, ProvOut& r) {
    // (one base pointer for the wave and 32-bit offsets per lane: the stores take the base from scalar registers)
    const uint32_t rdummy = 4u * lane, ldummy = (uint32_t)lzr::kProvRecBytes + 8u * lane;
    const uint8_t* stg = sl->stage;
    uint32_t wa = (start >> 5) << 2, bp = start & 31u;
    uint32_t d0 = *(const uint32_t*)(stg + wa), d1 = *(const uint32_t*)(stg + wa + 4);
    uint32_t tsel = 0, tmsk = (1u << kSyncLitBits) - 1u, seen = 0;
    uint32_t plen = 0, run = run0, nlit = 0, nout = 0, pend = 0;   // pend: literals in the accumulator (nlit mod 8)
    uint32_t roff = rdummy + kProvRecRow, loff = ldummy + kProvLitRow;   // byte offsets of my next record / literal group in the scratch (row 1 is the first)
    uint32_t lb0 = 0, lb1 = 0;                             // pending literals: the newest at the top of lb1:lb0
    int32_t need = -0x40000000;
    if (chunk_end > in_bits) chunk_end = in_bits;          // (the zero fill behind the input is not worth decoding)
    if (start >= in_bits) { seen = kEntInvalid; chunk_end = 0; }   // nothing left for this sub-chunk: the checked step says what that means
    bool go = false;
    if (start < chunk_end) do {   // (tested at the bottom: one mask update and one branch per iteration)
        SWC_SYNC_STAT(5, 1);
        SWC_SYNC_ITER();
        const uint32_t bits = funnel32(d1, d0, bp);
        uint32_t e = sl->lut[(bits & tmsk) | tsel];
        const uint32_t nx = *(const uint32_t*)(stg + wa + 8);
        if (e == 0) { SWC_SYNC_STAT(7, 1); e = lc.lookup(sl, bits, tsel); }
        seen |= e;
        chunk_end &= ~sext_bit31(e);
        const bool was_dist = tsel != 0;
        const uint32_t is_lit = (e >> 15) & 1u;
        const uint32_t val = ((e >> 16) & 0x7FFFu) + bfe32(bits, (e >> kEntClenShift) & 15u, (e >> kEntExtShift) & 15u);
        // literal: into the accumulator (a shift by 0 leaves it alone)
        const uint32_t sh8 = is_lit << 3;
        lb0 = funnel32(lb1, lb0, sh8);
        lb1 = funnel32(val, lb1, sh8);
        nlit += is_lit;
        pend += is_lit;
        const uint32_t full8 = pend & 8u;                  // the eighth literal of a group has just entered
        pend &= 7u;
        const bool full = full8 != 0u;
        const int32_t nd = (int32_t)val - (int32_t)nout;   // a distance needs this much output in front of the sub-chunk
        need = was_dist && nd > need ? nd : need;
        if ((tsel | full8) != 0u) {
            store_u64(prov + (full ? loff : ldummy), ((uint64_t)lb1 << 32) | lb0);
            store_u32(prov + (was_dist ? roff : rdummy), lzr::make_match(run, plen, val));
            loff += full ? kProvLitRow : 0u;
            roff += was_dist ? kProvRecRow : 0u;
        }
        nout += is_lit + (was_dist ? plen : 0u);
        run = was_dist ? 0u : run + is_lit;
        const bool is_len = (e & kEntLen) != 0u;
        plen = is_len ? val : plen;
        tsel = e & kEntLen;
        tmsk = is_len ? (1u << kSyncDistBits) - 1u : (1u << kSyncLitBits) - 1u;
        bp += e & 31u;
        const bool sh = bp >= 32;
        wa += sh ? 4u : 0u;
        bp &= 31u;
        d0 = sh ? d1 : d0;
        d1 = sh ? nx : d1;
        go = is_len || (wa << 3) + bp < chunk_end;
    } while (go);
    if (nlit & 7u) {   // the last, incomplete group: its bytes sit at the top of the accumulator
        const uint32_t dn = (8u - (nlit & 7u)) << 3;
        store_u64(prov + loff, (((uint64_t)lb1 << 32) | lb0) >> dn);
    }
    uint32_t flags = flags_of_seen(seen);
    uint32_t tail = 0;
    if (run > 0) {
        if (flags & kFlagEob) tail = run;
        else { store_u32(prov + roff, lzr::make_lits(run)); roff += kProvRecRow; }
    }
    const uint32_t endb = (wa << 3) + bp;
    if (endb > in_bits) flags |= kFlagFail;                // a symbol ran past the end of the input
    if (nlit + run0 > lzr::kLitRunMax) flags |= kFlagSlow;  // (conservative: only then can a run exceed what a match record carries)
    r.end = (flags & kFlagFail) ? kPosFail : endb;
    r.nlit = nlit; r.nrec = (roff - rdummy) / kProvRecRow - 1u; r.nout = nout; r.flags = flags; r.tail = tail; r.need = need;
}

// A lane's piece of the round moves from its column of the scratch to its final place: `nrec` records to `rdst` (dword
// aligned), `nlit` literal bytes to `ldst` (any alignment).  The loads of a step read one row: coalesced.
SWC_D void copy_prov(gcptr plit, gcptr prec, uint32_t nlit, uint32_t nrec, gptr ldst, SWC_AS_GLOBAL uint32_t* rdst) {
    // Sixteen records and eight literal groups are loaded per step, all before the first store (the scratch of all resident
    // waves exceeds the L2, so a load takes its several hundred cycles: one load per step would expose that latency forty
    // times per round).  Rows past the lane's count hold something and exist (the scratch is sized for the worst case): they
    // are loaded and not stored.
    const uint32_t ngrp = (nlit + 7u) >> 3;
    for (uint32_t i = 0, g = 0; i < nrec || g < ngrp; i += 16, g += 8) {
        uint32_t v[16];
        uint64_t w[8];
#pragma unroll
        for (uint32_t k = 0; k < 16; k++) {
            const uint32_t row = i + k + 1u < (uint32_t)lzr::kProvRecRows ? i + k + 1u : (uint32_t)lzr::kProvRecRows - 1u;
            v[k] = load_u32(prec + (size_t)row * kProvRecRow);
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t row = g + k + 1u < (uint32_t)lzr::kProvLitRows ? g + k + 1u : (uint32_t)lzr::kProvLitRows - 1u;
            w[k] = load_u64(plit + (size_t)row * kProvLitRow);
        }
        // (wide stores: the lanes' destinations lie apart, so the memory pipeline takes a store lane by lane -- four records or
        // two groups per lane and instruction instead of one)
#pragma unroll
        for (uint32_t k = 0; k < 16; k += 4) {
            if (i + k + 4u <= nrec) store_u128_a4((gptr)(rdst + i + k), v[k], v[k + 1], v[k + 2], v[k + 3]);
            else {
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) if (i + k + q < nrec) rdst[i + k + q] = v[k + q];
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k += 2) {
            const uint32_t at = 8u * (g + k);
            if (at + 16u <= nlit) store_u128_a4(ldst + at, (uint32_t)w[k], (uint32_t)(w[k] >> 32), (uint32_t)w[k + 1], (uint32_t)(w[k + 1] >> 32));
            else {
#pragma unroll
                for (uint32_t q = 0; q < 2; q++) {
                    const uint32_t aq = at + 8u * q;
                    if (aq + 8u <= nlit) store_u64(ldst + aq, w[k + q]);
                    else if (aq < nlit) { uint64_t x = w[k + q]; for (uint32_t z = aq; z < nlit; z++, x >>= 8) ldst[z] = (uint8_t)x; }
                }
            }
        }
    }
}

// ---- the rounds of one block ---------------------------------------------------------------------------------------
// Decodes from the reader's position until the end-of-block symbol (kSyncEob) or until something the fast path leaves
// to the checked step (kSyncBail; kSyncBailCap: the capacity lies inside the next round).  Commits whole rounds only.
SWC_D int sync_block(Lane& ln, SyncLds* sl, SyncProf& pf) {
    using simt::PT;
    constexpr int N = kWave;
    // Literals in front of the block that no record covers yet (the tail of the previous block, stored bytes): the first
    // sub-chunk of the block takes a short run into its first record; a long one becomes a record of its own here.
    uint32_t pending = 0;
    {
        const uint64_t kept = ln.pos < ln.cap ? ln.pos : ln.cap;
        const uint64_t open = kept > ln.last_end ? kept - ln.last_end : 0;
        if (ln.prov != nullptr && open <= 64) pending = (uint32_t)open;
        else ln.flush_tail();
    }
    uint64_t P = simt::uniform(ln.br.consumed_bits());
    uint64_t pos = simt::uniform(ln.pos), nlit = simt::uniform(ln.nlit);
    uint32_t nrec = simt::uniform(ln.nrec);
    const uint32_t in_len = ln.br.len;
    gcptr in = ln.br.in;
    int result = kSyncBail;
    PT<uint32_t, N> start, endp, pe, c_lit, c_rec, c_out, flg, x_lit, x_rec, x_out, c_tail, c_need;
    PT<bool, N> todo, pb, have;
    LongCodes lc;
    lc.load(sl);
    for (;;) {
        const uint32_t B = (uint32_t)(P >> 3) & ~3u;
        const uint32_t q0 = (uint32_t)(P - 8ull * B);
        SWC_SP(pf, 6)
        SWC_SPC(pf, 7, 1);
        const uint64_t left = (uint64_t)(in_len - B) * 8;
        const uint32_t in_bits = left > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)left;
        // stage [B, B + kSyncStage), zero-filled beyond the input
        SIMT_BEGIN(t, N)
            for (uint32_t o = 16u * (uint32_t)t; o < kSyncStage; o += 16u * N) {
                uint64_t a = 0, b = 0;
                const uint64_t at = (uint64_t)B + o;
                if (at + 16 <= in_len) { a = load_u64(in + at); b = load_u64(in + at + 8); }
                else {
                    for (uint32_t k = 0; k < 8; k++) if (at + k < in_len) a |= (uint64_t)in[at + k] << (8 * k);
                    for (uint32_t k = 0; k < 8; k++) if (at + 8 + k < in_len) b |= (uint64_t)in[at + 8 + k] << (8 * k);
                }
                uint32_t* st32 = (uint32_t*)(sl->stage + o);
                st32[0] = (uint32_t)a; st32[1] = (uint32_t)(a >> 32); st32[2] = (uint32_t)b; st32[3] = (uint32_t)(b >> 32);
            }
            start[t] = t == 0 ? q0 : (uint32_t)t * kSyncChunk * 8u;
        SIMT_END_WAVE
        uint32_t nv = 0;
        bool eob = false, bail = false;
        const bool chk = (uint64_t)B + kSyncStage > in_len;   // only the last rounds of a stream can run out of input
        SWC_SP(pf, 2)
        // pass 1: where does a decode from my guess end?  (no counting)
        SWC_SYNC_STAT(2, 1);
        SWC_SPC(pf, 8, 1);
        SIMT_BEGIN(t, N)
            ChunkOut r;
            SWC_SYNC_LANE_BEGIN(t)
            const uint32_t ce = ((uint32_t)t + 1u) * kSyncChunk * 8u;
            r.end = walk_chunk(sl, lc, start[t], ce, in_bits);
            endp[t] = r.end; flg[t] = 0; have[t] = false;
            SWC_SYNC_LANE_END(t, N, 0)
        SIMT_END
        SWC_SP(pf, 9)
        // ---- the round in ONE more decode (the common case): every lane decodes its sub-chunk from the end of its left
        // neighbour into its scratch (decode_chunk_prov), the chain is checked as below, the pieces are copied to their
        // offsets.  Anything unusual -- a symbol for the checked step, a literal run too long for one record, the capacity
        // or the workspace inside the round, a distance beyond the output -- abandons the attempt BEFORE anything is
        // committed; the general passes below then take the round.
        if (ln.prov != nullptr && pos < ln.cap) {
            bool ok = true;
            for (;;) {
                simt::wave_shift_up<N>(pe, endp, q0);
                SIMT_BEGIN(t, N) pb[t] = !(have[t] && (t == 0 || start[t] == pe[t])); SIMT_END
                const uint64_t m_bad = simt::wave_ballot<N>(pb);
                const int b = m_bad ? simt::ctz64(m_bad) : 64;
                const uint64_t chain = b == 64 ? ~0ull : (1ull << b) - 1ull;
                SIMT_BEGIN(t, N) pb[t] = (flg[t] & kFlagEob) != 0; SIMT_END
                const uint64_t m_eob = simt::wave_ballot<N>(pb) & chain;
                const int E = m_eob ? simt::ctz64(m_eob) : 64;
                nv = (uint32_t)(E < 64 ? E + 1 : b);
                SIMT_BEGIN(t, N) pb[t] = (flg[t] & (kFlagFail | kFlagSlow)) != 0; SIMT_END
                if (simt::wave_ballot<N>(pb) & (nv == 64 ? ~0ull : (1ull << nv) - 1ull)) { ok = false; break; }
                if (E < 64) { eob = true; break; }
                if (b == 64) break;
                SWC_SYNC_STAT(2, 1);
                SWC_SPC(pf, 8, 1);
                SIMT_BEGIN(t, N)
                    todo[t] = t == 0 ? !have[t] : pe[t] != kPosFail && (start[t] != pe[t] || !have[t]);
                    SWC_SYNC_LANE_BEGIN(t)
                    if (todo[t]) {
                        SWC_SYNC_STAT(3, 1);
                        if (t != 0) start[t] = pe[t];
                        ProvOut r;
                        const uint32_t ce = ((uint32_t)t + 1u) * kSyncChunk * 8u;
                        if (start[t] + kSyncChunk * 8u < ce) {
                            // my left neighbour stopped in front of my sub-chunk: at the end-of-block symbol or at something for
                            // the checked step.  Nothing of mine belongs to the block (and a decode from there could run through
                            // several sub-chunks: the scratch is sized for one).
                            r.end = kPosFail; r.nlit = r.nrec = r.nout = r.tail = 0; r.flags = kFlagFail; r.need = 0;
                        } else
                        decode_chunk_prov(sl, lc, start[t], ce, in_bits, ln.prov, (uint32_t)t, t == 0 ? pending : 0u, r);
                        endp[t] = r.end; c_lit[t] = r.nlit; c_rec[t] = r.nrec; c_out[t] = r.nout; flg[t] = r.flags;
                        c_tail[t] = r.tail; c_need[t] = (uint32_t)r.need;
                        have[t] = true;
                    }
                    SWC_SYNC_LANE_END(t, N, 1)
                SIMT_END
                SWC_SP(pf, 3)
            }
            if (ok) {
                SIMT_BEGIN(t, N)
                    const bool v = (uint32_t)t < nv;
                    x_lit[t] = v ? c_lit[t] : 0u; x_rec[t] = v ? c_rec[t] : 0u; x_out[t] = v ? c_out[t] : 0u;
                SIMT_END
                simt::wave_scan_incl<N>(x_lit);
                simt::wave_scan_incl<N>(x_rec);
                simt::wave_scan_incl<N>(x_out);
                const uint32_t tot_lit = simt::wave_read<N>(x_lit, N - 1), tot_rec = simt::wave_read<N>(x_rec, N - 1), tot_out = simt::wave_read<N>(x_out, N - 1);
                SWC_SP(pf, 4)
                // every distance must reach back no further than the output in front of its match
                SIMT_BEGIN(t, N)
                    const uint64_t p0 = pos + (x_out[t] - c_out[t]);
                    const int32_t room = p0 > 0x40000000ull ? 0x40000000 : (int32_t)p0;
                    pb[t] = (uint32_t)t < nv && (int32_t)c_need[t] > room;
                SIMT_END
                if (simt::wave_ballot<N>(pb)) ok = false;
                if (pos + tot_out > ln.cap || (uint64_t)nrec + tot_rec > ln.max_rec) ok = false;
                if (ok) {
                    SIMT_BEGIN(t, N)
                        if ((uint32_t)t < nv) {
                            copy_prov(ln.prov + lzr::kProvRecBytes + 8u * (uint32_t)t, ln.prov + 4u * (uint32_t)t, c_lit[t], c_rec[t],
                                      ln.lits + nlit + (x_lit[t] - c_lit[t]), ln.recs + nrec + (x_rec[t] - c_rec[t]));
                        }
                    SIMT_END
                    SWC_SP(pf, 5)
                    SWC_SYNC_STAT(0, 1);
                    pos += tot_out;
                    nlit += tot_lit;
                    nrec += tot_rec;
                    P = 8ull * B + simt::wave_read<N>(endp, (int)nv - 1);
                    ln.last_end = pos - (eob ? simt::wave_read<N>(c_tail, (int)nv - 1) : 0u);
                    pending = 0;
                    if (eob) { result = kSyncEob; break; }
                    continue;
                }
            }
            // abandoned: the general passes start over
            SIMT_BEGIN(t, N) have[t] = false; flg[t] = 0; SIMT_END
            eob = false;
        }
        if (pending) {   // (first round only: nothing has been committed, so the lane's counters are the block's)
            ln.pos = pos; ln.nlit = nlit; ln.nrec = nrec;
            ln.flush_tail();
            nrec = ln.nrec;
            pending = 0;
        }
        for (;;) {
            simt::wave_shift_up<N>(pe, endp, q0);
            // a lane is final when it has been counted from the end of a final left neighbour
            SIMT_BEGIN(t, N) pb[t] = !(have[t] && (t == 0 || start[t] == pe[t])); SIMT_END
            const uint64_t m_bad = simt::wave_ballot<N>(pb);
            const int b = m_bad ? simt::ctz64(m_bad) : 64;               // lanes [0, b) are on the true sequence
            const uint64_t chain = b == 64 ? ~0ull : (1ull << b) - 1ull;
            SIMT_BEGIN(t, N) pb[t] = (flg[t] & kFlagEob) != 0; SIMT_END
            const uint64_t m_eob = simt::wave_ballot<N>(pb) & chain;
            const int E = m_eob ? simt::ctz64(m_eob) : 64;               // the lane that met the end of the block
            nv = (uint32_t)(E < 64 ? E + 1 : b);
            SIMT_BEGIN(t, N) pb[t] = (flg[t] & kFlagFail) != 0; SIMT_END
            const uint64_t m_fail = simt::wave_ballot<N>(pb) & (nv == 64 ? ~0ull : (1ull << nv) - 1ull);
            if (m_fail) { bail = true; break; }                          // the true sequence holds something for the checked step
            if (E < 64) { eob = true; break; }
            if (b == 64) break;
            SWC_SYNC_STAT(2, 1);   // passes
            SWC_SPC(pf, 8, 1);
            SIMT_BEGIN(t, N)
                todo[t] = t == 0 ? !have[t] : pe[t] != kPosFail && (start[t] != pe[t] || !have[t]);
                SWC_SYNC_LANE_BEGIN(t)
                if (todo[t]) {
                    SWC_SYNC_STAT(3, 1);   // lane decodes
                    if (t != 0) start[t] = pe[t];
                    ChunkOut r;
                    const uint32_t ce = ((uint32_t)t + 1u) * kSyncChunk * 8u;
                    if (start[t] + kSyncChunk * 8u < ce) { r.end = kPosFail; r.nlit = r.nrec = r.nout = 0; r.flags = kFlagFail; }   // (see the pass above)
                    else if (chk) decode_chunk<0, false, true>(sl, lc, start[t], ce, in_bits, nullptr, nullptr, 0, r);
                    else decode_chunk<0, false, false>(sl, lc, start[t], ce, in_bits, nullptr, nullptr, 0, r);
                    if (r.nlit > lzr::kLitRunMax) {   // a literal run may need a record of its own: count those too
                        if (chk) decode_chunk<0, true, true>(sl, lc, start[t], ce, in_bits, nullptr, nullptr, 0, r);
                        else decode_chunk<0, true, false>(sl, lc, start[t], ce, in_bits, nullptr, nullptr, 0, r);
                    }
                    endp[t] = r.end; c_lit[t] = r.nlit; c_rec[t] = r.nrec; c_out[t] = r.nout; flg[t] = r.flags;
                    have[t] = true;
                }
                SWC_SYNC_LANE_END(t, N, 1)
            SIMT_END
            SWC_SP(pf, 3)
        }
        if (bail) { SWC_SYNC_STAT(1, 1); break; }
        SWC_SYNC_STAT(0, 1);   // rounds that converged
        // exclusive offsets of the lanes [0, nv)
        SIMT_BEGIN(t, N)
            const bool v = (uint32_t)t < nv;
            x_lit[t] = v ? c_lit[t] : 0u; x_rec[t] = v ? c_rec[t] : 0u; x_out[t] = v ? c_out[t] : 0u;
        SIMT_END
        simt::wave_scan_incl<N>(x_lit);
        simt::wave_scan_incl<N>(x_rec);
        simt::wave_scan_incl<N>(x_out);
        const uint32_t tot_lit = simt::wave_read<N>(x_lit, N - 1), tot_rec = simt::wave_read<N>(x_rec, N - 1), tot_out = simt::wave_read<N>(x_out, N - 1);
        SWC_SP(pf, 4)
        const bool beyond = pos >= ln.cap;                               // size pass: nothing is kept, distances are still checked
        if (!beyond && pos + tot_out > ln.cap) { result = kSyncBailCap; break; }
        if (!beyond && (uint64_t)nrec + tot_rec > ln.max_rec) break;
        SIMT_BEGIN(t, N)
            SWC_SYNC_LANE_BEGIN(t)
            if ((uint32_t)t < nv) {
                ChunkOut r;
                const uint32_t ce = ((uint32_t)t + 1u) * kSyncChunk * 8u;
                const uint64_t p0 = pos + (x_out[t] - c_out[t]);
                const bool bigs = c_lit[t] > lzr::kLitRunMax;
                if (beyond) decode_chunk<2, false, true>(sl, lc, start[t], ce, in_bits, nullptr, nullptr, p0, r);
                else if (bigs || chk) decode_chunk<1, true, true>(sl, lc, start[t], ce, in_bits, ln.lits + nlit + (x_lit[t] - c_lit[t]), ln.recs + nrec + (x_rec[t] - c_rec[t]), p0, r);
                else decode_chunk<1, false, false>(sl, lc, start[t], ce, in_bits, ln.lits + nlit + (x_lit[t] - c_lit[t]), ln.recs + nrec + (x_rec[t] - c_rec[t]), p0, r);
                flg[t] = r.flags;
            }
            SWC_SYNC_LANE_END(t, N, 2)
        SIMT_END
        SWC_SP(pf, 5)
        SIMT_BEGIN(t, N) pb[t] = (uint32_t)t < nv && (flg[t] & kFlagTrap) != 0; SIMT_END
        if (simt::wave_ballot<N>(pb)) break;                             // a distance beyond the output: the checked step reports it
        pos += tot_out;
        if (!beyond) { nlit += tot_lit; nrec += tot_rec; }
        P = 8ull * B + simt::wave_read<N>(endp, (int)nv - 1);
        ln.last_end = pos < ln.cap ? pos : (ln.last_end > ln.cap ? ln.last_end : ln.cap);
        if (eob) { result = kSyncEob; break; }
    }
    ln.pos = pos;
    ln.nlit = nlit;
    ln.nrec = nrec;
    ln.br.seek(P);
    return result;
}

// ---- the code-length section of a dynamic header (Deflate.swift:86-167), wave-parallel where it can be -----------------
// Same results and errors as Lane::build_dynamic: the code lengths are decoded once, serially (through a 128-entry
// table of the code-length code), into an LDS array; the per-length histogram and the counting sort that produces the
// (length, symbol)-sorted arrays run on all lanes.
SWC_D int build_dynamic_par(Lane& ln, SyncLds* sl, SyncProf& pf) {
    using simt::PT;
    constexpr int N = kWave;
    BitReader& br = ln.br;
    br.refill();
    if (br.bc < 14) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :86
    const int literals = (int)br.bits(5) + 257;
    if (literals > 286) return SWC_E_DEFLATE_WRONG_SYMBOL;  // :94
    const int distances = (int)br.bits(5) + 1;
    const int ncl = (int)br.bits(4) + 4;
    br.refill();
    const uint64_t total_left = (uint64_t)br.len * 8 - br.consumed_bits();
    if (total_left < (uint64_t)(3 * ncl)) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :101
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint64_t clens = 0;  // 3 bits per symbol, indexed by symbol
    for (int i = 0; i < ncl; i++) {
        br.refill();
        clens |= (uint64_t)br.bits(3) << (3 * order[i]);
    }
    Lane::ClTable cl;
    {
        uint64_t cnt8 = 0, runp = 0;  // eight 8-bit fields, indexed by length
        for (int s = 0; s < 19; s++) {
            const uint32_t len = (uint32_t)(clens >> (3 * s)) & 7u;
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
            const uint32_t len = (uint32_t)(clens >> (3 * s)) & 7u;
            if (len) {
                const uint32_t p = (uint32_t)(runp >> (8 * len)) & 255u;
                runp += 1ull << (8 * len);
                if (p < 12) cl.sym_lo |= (uint64_t)s << (5 * p);
                else cl.sym_hi |= (uint64_t)s << (5 * (p - 12));
            }
        }
    }
    uint8_t* lens = sl->stage + kHdrLens;
    uint32_t* cnt = (uint32_t*)(sl->stage + kHdrCnt);
    uint8_t* cl_lut = sl->stage + kHdrClLut;
    const int total = literals + distances;
    SIMT_BEGIN(t, N)
        for (int i = t; i < 80; i += N) ((uint32_t*)lens)[i] = 0;
        if (t < 48) cnt[t] = 0;
        for (int x = t; x < 128; x += N) {   // the code the stream bits x (first bit = bit 0) begin with
            uint32_t len;
            const int idx = Lane::cl_lookup(cl, brev32((uint32_t)x) >> 17, len);
            const uint32_t sym = idx < 0 ? 0u : (uint32_t)((idx < 12 ? cl.sym_lo >> (5 * idx) : cl.sym_hi >> (5 * (idx - 12))) & 31u);
            cl_lut[x] = idx < 0 ? (uint8_t)0xFF : (uint8_t)(len | (sym << 3));
        }
    SIMT_END_WAVE
    {   // Deflate.swift:117-162, every lane the same
        int n = 0;
        uint32_t prev = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        // the 128-entry table of the code-length code sits in two registers spread over the lanes (lane i: entries i and
        // 64 + i): the serial loop below looks its entry up with a cross-lane read instead of an LDS round trip per symbol
        const uint32_t lut_lo = cl_lut[threadIdx.x & 63u], lut_hi = cl_lut[64u + (threadIdx.x & 63u)];
#endif
        while (n < total) {
            br.refill();
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t x7 = simt::uniform(br.peek32() & 127u);
            const uint32_t e = x7 < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)lut_lo, (int)x7) : (uint32_t)__builtin_amdgcn_readlane((int)lut_hi, (int)(x7 - 64u));
#else
            const uint32_t e = simt::uniform((uint32_t)cl_lut[br.peek32() & 127u]);   // (scalar from here on: see BitReader)
#endif
            if (e == 0xFFu || (e & 7u) > br.bc) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND;  // :122
            br.consume(e & 7u);
            const uint32_t sym = e >> 3;
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
#if defined(__HIP_DEVICE_COMPILE__)
            // every lane stores (lane i: entry min(i, rep - 1)): no exec-mask region in this serial loop; zeros land on zeros
            { const int i = ln.wlane < rep - 1 ? ln.wlane : rep - 1; lens[n + i] = (uint8_t)val; }
#else
            if (val != 0) for (int i = ln.wlane; i < rep; i += ln.wlanes) lens[n + i] = (uint8_t)val;
#endif
            n += rep;
        }
        if (n != total) return SWC_E_DEFLATE_WRONG_SYMBOL;  // :161
    }
    simt::wave_fence();
    SWC_SP(pf, 0)
    sync_tables_from_lengths(sl, literals, distances);
    return SWC_OK;
}

// Deflate.swift:77-81 with the fixed code of Deflate+Constants.swift:11-173: the same table build from the fixed lengths
SWC_D void build_static_par(SyncLds* sl, SyncProf& pf) {
    (void)pf;
    constexpr int N = kWave;
    uint8_t* lens = sl->stage + kHdrLens;
    uint32_t* cnt = (uint32_t*)(sl->stage + kHdrCnt);
    SIMT_BEGIN(t, N)
        if (t < 48) cnt[t] = 0;
        for (int s = t; s < 320; s += N) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
    SIMT_END_WAVE
    sync_tables_from_lengths(sl, 288, 32);
}

// ---- the job -------------------------------------------------------------------------------------------------------
// Deflate.swift:30-249 for one stream on one wavefront.  `ws` / `ws_bytes`: the stream's area in the HBM workspace
// (lzr::StreamHeader | records | literal stream).  On the device every lane of the wave calls this with its lane number;
// the host emulation calls it once (lane 0 of 1) and runs the 64 lanes of the parallel parts one after another.
SWC_D void inflate_sync_job(Job& job, SyncLds* sl, uint8_t* ws, size_t ws_bytes, int lane, int lanes, uint64_t* prof = nullptr) {
    Lane ln;
    SyncProf pf;
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    pf.tlast = __builtin_readcyclecounter();
#endif
    ln.wlane = lane;
    ln.wlanes = lanes;
    ln.l = LaneLds{sl->syms, 1};
    ln.out = (gptr)job.out;
    ln.cap = job.out_cap;
    ln.pos = 0;
    ln.nrec = 0;
    ln.nlit = 0;
    ln.last_end = 0;
    const size_t lo = ws ? lzr::lit_offset(ws_bytes, job.out_cap) : 0;
    ln.recs = (SWC_AS_GLOBAL uint32_t*)(ws + sizeof(lzr::StreamHeader));
    size_t rec_end = lo;   // the lanes' scratch (lzr::kProvBytes) sits between the record list and the literal stream, if the area has room for it
    ln.prov = nullptr;
    if (lo >= sizeof(lzr::StreamHeader) + 256 + lzr::kProvBytes) {
        rec_end = (lo - lzr::kProvBytes) & ~(size_t)15;
        ln.prov = (gptr)(ws + rec_end);
    }
    ln.max_rec = rec_end > sizeof(lzr::StreamHeader) ? (uint32_t)((rec_end - sizeof(lzr::StreamHeader)) / 4) : 0u;
    ln.lits = (gptr)(ws + lo);
    int st = SWC_OK;
    if (lo == 0) {
        st = SWC_E_NEED_WORKSPACE;
        ln.br.init((gcptr)job.in, 0, 0);
    } else if (job.in_len > 0xFFFFFFF0ull) {
        st = SWC_E_INVALID_ARGUMENT;  // streams are addressed with 32-bit byte offsets on device
        ln.br.init((gcptr)job.in, 0, 0);
    } else {
        ln.br.init((gcptr)job.in, (uint32_t)job.in_len, 0);
        if ((uint64_t)ln.br.len * 8 < 10) st = SWC_E_DEFLATE_WRONG_BLOCK_TYPE;  // :36
        while (st == SWC_OK) {
            ln.br.refill();
            if (ln.br.bc < 3) { st = SWC_E_REF_TRAP; break; }   // a second or later block header past the end: LsbBitReader.bit() traps
            const uint32_t is_last = ln.br.bits(1);
            const uint32_t type = ln.br.bits(2);
            if (type == 0) {
                st = ln.run_stored();
            } else if (type == 1 || type == 2) {
                SWC_SP(pf, 6)
                if (type == 1) build_static_par(sl, pf);
                else st = build_dynamic_par(ln, sl, pf);
                SWC_SP(pf, 1)
                if (st == SWC_OK) {
                    bool fast = sl->aux[kAuxLit + kAuxOver] == 0 && sl->aux[kAuxDist + kAuxOver] == 0;
                    for (;;) {   // Deflate.swift:171-236
                        if (fast) {
                            const int r = sync_block(ln, sl, pf);
                            if (r == kSyncEob) break;
                            if (r == kSyncBail) fast = false;   // the checked step takes the rest of the block
                            // kSyncBailCap: checked steps until the capacity is behind us
                        }
                        const uint64_t until = fast && ln.pos < ln.cap ? ln.cap : ~0ull;
                        int s2 = SWC_OK;
                        do { s2 = careful_step_lds(ln, sl); } while (s2 == SWC_OK && ln.pos < until && until != ~0ull);
                        if (s2 == -1) break;
                        if (s2) { st = s2; break; }
                    }
                }
            } else {
                st = SWC_E_DEFLATE_WRONG_BLOCK_TYPE;  // :239
            }
            if (st != SWC_OK || is_last) break;  // :243
        }
        ln.flush_tail();
    }
    if (ln.nrec > ln.max_rec) {
        st = SWC_E_NEED_WORKSPACE;  // the record list outgrew the workspace (sized from out_cap)
        ln.nrec = ln.max_rec;
    }
    if (st == SWC_OK && ln.pos > ln.cap) st = SWC_E_CAPACITY;
    if (ws && ws_bytes >= sizeof(lzr::StreamHeader) && (lane == 0)) {
        SWC_AS_GLOBAL lzr::StreamHeader* h = (SWC_AS_GLOBAL lzr::StreamHeader*)ws;
        h->nrec = ln.nrec;
        h->pad0 = 0;
        h->nlit = ln.nlit;
    }
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    SWC_SP(pf, 6)
    if (prof && lane == 0) for (int k = 0; k < 10; k++) prof[k] = pf.acc[k];
#else
    (void)prof; (void)pf;
#endif
    const uint64_t bits = ln.br.consumed_bits();
    const uint64_t consumed = (bits + 7) >> 3;  // callers align() right after (GzipArchive.swift:89)
    job.in_consumed = consumed > job.in_len ? job.in_len : consumed;
    job.out_len = ln.pos;
    job.status = st;
}

}  // namespace inflate
}  // namespace swc
#endif
