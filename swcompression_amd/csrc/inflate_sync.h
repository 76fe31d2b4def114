// inflate_sync.h -- Deflate entropy decode (phase 1), one compressed stream per WAVEFRONT, 64 lanes decoding 64
// sub-chunks of the stream at once.
//
// Replaces the symbol loop of Deflate.decompress(_ bitReader:) (reference Sources/Deflate/Deflate.swift:171-236), the
// code-length section of a dynamic header (:86-167) and the table construction of Code.huffmanCodes / DecodingTree.init
// (Sources/Common/CodingTree/Code.swift:15-39, DecodingTree.swift:15-34).  Huffman decoding is a serial chain -- where a
// code starts is known only when the previous one has been decoded -- but a decoder started at a WRONG bit offset falls
// into step with the true symbol sequence after a few symbols.  A wavefront therefore cuts the next kSyncRound bytes of the
// stream into 64 sub-chunks of kSyncChunk bytes, one per lane, and runs a ROUND:
//
//   walk        lane 0 starts at the true position, every other lane at its sub-chunk boundary; all decode -- nothing but
//               the code lengths -- until they cross the end of their sub-chunk and note where they ended (walk_chunk: ten
//               vector instructions per code);
//   provisional every lane decodes ONCE more from where its left neighbour's walk ended, and this pass counts and emits:
//   decode      records and literal groups go to the wave's row-major scratch in the workspace (decode_chunk_prov: about
//               thirty vector instructions per code, no conditional region -- both stores of a step are unconditional and
//               aim at the row that is current, so a row is simply rewritten until it is final);
//   chain check lane k's decode must end where lane k + 1's began; lanes for which it does not decode again (1.4 passes
//               per round on text); when the chain holds, three wave scans give every lane its offsets and copy_prov moves
//               the rows to their final places in the record list and the literal stream (lz_resolve.h);
//   fallback    anything unusual -- a symbol for the checked step, a literal run too long for one record, the capacity or the
//               workspace inside the round, a distance beyond the output -- abandons the attempt before anything is committed;
//               the general passes (count, scan, emit: decode_chunk) or the checked one-symbol step take over.
//
// All 64 lanes share ONE set of tables in LDS: direct lookup tables (10 / 8 bits, 32-bit entries that carry everything a
// step needs, laid out so that the position update, the table switch of a length code and the loop exit are one instruction
// each -- see the entry layout below) and two-level SUBTABLES for the codes longer than that (184 entries in LDS, which
// covers text and most binary data; larger sets overflow to the workspace).  The input of a round is staged in LDS shifted
// left by two bits, so that the funnel shift a lane reads its window with yields the table ADDRESS bits directly.
// Sub-chunks are 17 dwords long: an odd stride keeps the 64 lanes on different LDS banks, and everything in LDS together is
// exactly 10 KB (16 waves per CU).  What only the checked step needs -- canonical limits and the sorted symbol arrays with the
// reference's semantics for incomplete and over-subscribed sets -- lives in the workspace, not in LDS.
#ifndef SWC_INFLATE_SYNC_H
#define SWC_INFLATE_SYNC_H

#include "inflate_lane.h"
#include "simt.h"

namespace swc {
namespace inflate {

// Index bits of the first-level literal/length table.  9 bits make the wave's LDS exactly 8 KB -- 20 waves per CU instead of 16, and
// the kernel gains with its waves (12 / 14 / 16 per CU: 7.72 / 7.09 / 6.64 ms) -- but five waves per SIMD leave 96 VGPRs for a body that
// needs 127: 108 bytes of scratch, 6.55 ms against 6.50-6.59 (and 6.70 at four waves: the second-level lookups of the ten-bit codes).
// profiles/r06_experiments.txt, r07h-i.
#ifndef SWC_SYNC_LIT_BITS
#define SWC_SYNC_LIT_BITS 10
#endif
constexpr int kSyncLitBits = SWC_SYNC_LIT_BITS, kSyncDistBits = 8;
// Input bytes per lane and round: a whole number of dwords and an ODD number of them (17), so that the lanes, which start
// a pass at the same offset of their sub-chunks, read 64 different LDS banks without any padding of the staged input.
// With 68 bytes the wave's LDS is EXACTLY 10,240 bytes = 16 waves per CU (measured in round 2 by padding the LDS: 12 waves
// 13.7 ms, 9 waves 17.6 ms, 7 waves 22.1 ms against 12.05 ms; 60-byte sub-chunks 12.85 ms, 52 bytes 13.75 ms).
constexpr uint32_t kSyncChunk = SWC_SYNC_CHUNK;
constexpr uint32_t kSyncRound = 64u * kSyncChunk;
constexpr uint32_t kSyncStage = (kSyncRound + 32u + 15u) & ~15u;    // + what the last lane may read past its sub-chunk (a code of <= 48 bits, then the window's two dwords: < 24 bytes)
constexpr uint32_t kPosFail = 0xFFFFFFFFu;
#ifndef SWC_SYNC_WALK_BACK
#define SWC_SYNC_WALK_BACK 128
#endif
#ifndef SWC_COPY_REC
#define SWC_COPY_REC 16   // records / literal groups (both multiples of four) a lane moves per step of copy_prov
#define SWC_COPY_LIT 8
#endif
#ifndef SWC_WALK_WINDOW
#define SWC_WALK_WINDOW 0
#endif
constexpr uint32_t kSyncWalkBack = SWC_SYNC_WALK_BACK;   // bits
static_assert(kSyncChunk % 4 == 0 && kSyncChunk >= 36, "sub-chunks are whole dwords");

// ---- table entry ----------------------------------------------------------------------------------------------------
//   [0:4]   n2: bits the symbol takes (code + extra) + 2.  0: a LINK to a subtable (below)
//   [5:9]   c2: code length + 2
//   [10:11] both set UNLESS this is a length symbol: the two bits by which the index masks of the two tables differ
//   [12]    length symbol: the next code is a distance (the bit IS the byte offset of the distance table behind the lit/len table)
//   [13]    literal
//   [14:28] base value: the literal, the length base, the distance base - 1
//   [29]    distance symbol
//   [30]    stop: the end-of-block symbol (n2 > 2) or no symbol the fast path takes (n2 == 2: no bits)
//   [31]    set in length AND distance entries: added to a position it flips its top bit on at a length code and off again at the
//           distance code, so that ONE signed compare (position < end of the sub-chunk) is the whole loop condition; a stop
//           entry adds bit 30 and thereby ends the loop as well.  (Entries of the distance table that stop carry bit 31 too.)
// LINK (n2 == 0, bits 30 / 31 clear): [5:9] index width k of the subtable, [10:14] bits in front of it + 2 (12 / 10),
//   [18:29] index of the subtable's first entry.
// The window a lane looks at, `bits4`, is the stream from the symbol's first bit on, shifted LEFT by two (the stage is stored
// that way): (bits4 & mask) | table is the LDS byte address of the entry, and the extra bits of the symbol are
// bfe(bits4, 0, n2) >> c2.
constexpr uint32_t kEntLenBit = (uint32_t)kSyncLitBits + 2u;
constexpr uint32_t kEntNotLen = ((1u << (kSyncLitBits - kSyncDistBits)) - 1u) << (kSyncDistBits + 2), kEntLen = 1u << kEntLenBit, kEntLit = 1u << 13, kEntDist = 1u << 29, kEntStop = 1u << 30, kEntTog = 1u << 31;
static_assert(kSyncLitBits > kSyncDistBits && kEntLenBit <= 12u, "the entry keeps bits 10 .. 12 for the table switch");
constexpr uint32_t kEntBaseShift = 14, kEntPosMask = kEntTog | kEntStop | 31u;
constexpr uint32_t kLitMask4 = ((1u << kSyncLitBits) - 1u) << 2, kDistMask4 = ((1u << kSyncDistBits) - 1u) << 2;
static_assert(kEntLen == 4u << kSyncLitBits, "the length flag is the byte offset of the distance table");
static_assert((kLitMask4 & ~kEntNotLen) == kDistMask4, "the masks differ in the not-length bits");
constexpr uint32_t kInvLit = 2u | kEntNotLen | kEntStop, kInvDist = kInvLit | kEntTog;
// kind: 1 literal, 2 length, 3 end of block, 0 distance (value: the distance base itself)
SWC_HD uint32_t make_entry(uint32_t clen, uint32_t ext, uint32_t kind, uint32_t value) {
    const uint32_t c = (clen + ext + 2u) | ((clen + 2u) << 5);
    return kind == 1 ? c | kEntNotLen | kEntLit | (value << kEntBaseShift)
         : kind == 2 ? c | kEntLen | kEntTog | (value << kEntBaseShift)
         : kind == 3 ? c | kEntNotLen | kEntStop
                     : c | kEntNotLen | kEntDist | kEntTog | ((value - 1u) << kEntBaseShift);
}
SWC_HD uint32_t ent_bits(uint32_t e) { return (e & 31u) - 2u; }
SWC_HD bool ent_is_eob(uint32_t e) { return (e & kEntStop) != 0u && (e & 31u) > 2u; }
SWC_HD bool ent_is_link(uint32_t e) { return (e & kEntPosMask) == 0u; }

// The entry of lit/len symbol `sym` (0..287) / distance symbol `sym` (0..31) with a code of `d` bits
// (Deflate+Constants.swift: lengthBase / distanceBase as arithmetic).
SWC_HD uint32_t entry_of_symbol(bool dist, uint32_t sym, uint32_t d) {
    if (!dist) {
        if (sym < 256) return make_entry(d, 0, 1, sym);
        if (sym == 256) return make_entry(d, 0, 3, 0);
        if (sym > 285) return kInvLit;           // 286, 287: the checked step reports wrongSymbol
        const uint32_t s = sym - 257u;
        const uint32_t e = s < 8 || s == 28 ? 0u : (s >> 2) - 1u;
        const uint32_t base = s < 8 ? 3u + s : s == 28 ? 258u : 3u + ((4u + (s & 3u)) << e);
        return make_entry(d, e, 2, base);
    }
    if (sym > 29) return kInvDist;               // 30, 31: wrongSymbol
    const uint32_t e = sym < 4 ? 0u : (sym >> 1) - 1u;
    const uint32_t base = sym < 4 ? 1u + sym : 1u + ((2u + (sym & 1u)) << e);
    return make_entry(d, e, 0, base);
}

// The canonical tables the checked step decodes with (struct Table of round 1, now in the workspace): per length d the
// left-justified code limit, the slot word (sorted index of the first code - first code), the sorted index of the first code;
// the number of codes; the over-subscription flag.  Lit/len alphabet at kAuxLit, distance alphabet at kAuxDist.
constexpr int kAuxLim = 0, kAuxSlot = 16, kAuxStart = 32, kAuxCount = 48 /* == start[16] */, kAuxOver = 49, kAuxTable = 52;
constexpr int kAuxLit = 0, kAuxDist = kAuxTable, kAuxWords = 2 * kAuxTable;
// The spill in the workspace (lz_resolve.h: behind the scratch rows): aux | sorted symbols (16 bit each) | subtable overflow
constexpr uint32_t kSpillSyms = 512, kSpillSub = 1536, kSubLds = 184, kSubMax = (uint32_t)(lzr::kProvSpillBytes - kSpillSub) / 4u;
constexpr uint32_t kSymDist = 288;   // the distance symbols follow the lit/len symbols in the sorted array
static_assert(kAuxWords * 4 <= (int)kSpillSyms && kSpillSyms + 2 * (288 + 32) <= kSpillSub && kSubMax < 4096u, "spill layout");

struct SyncLds {   // 10,240 bytes: see kSyncChunk
    alignas(16) uint8_t stage[kSyncStage];                          // staged input of a round (shifted left by two bits); header build: scratch.  FIRST: the window reads (two dwords at a computed address) then need no base added
    uint32_t lut[(1 << kSyncLitBits) + (1 << kSyncDistBits)];      // direct tables: lit/len, then distance
    uint32_t sub[kSubLds];                                          // subtables of the long codes (if they fit)
#ifdef SWC_SYNC_LDS_PAD
    uint8_t occupancy_experiment_pad[SWC_SYNC_LDS_PAD];             // (tools/gpu_chunk_sweep.sh: fewer waves per CU, nothing else changed)
#endif
};
#if SWC_SYNC_CHUNK == 68 && !defined(SWC_SYNC_LDS_PAD) && SWC_SYNC_LIT_BITS == 10
static_assert(sizeof(SyncLds) == 10240, "16 waves per CU: the wave's LDS must stay within 160 KB / 16");
#endif
// header scratch inside `stage`
constexpr uint32_t kHdrLens = 0;       // 320 bytes: code length of symbol s (lit/len, then distance)
constexpr uint32_t kHdrRun = 320;      // 32 words: running / final number of codes per length (lit/len, distance)
constexpr uint32_t kHdrClLut = 512;    // 128 bytes: code-length code, len | symbol << 3 (0xFF: no code)
constexpr uint32_t kHdrSorted = 640;   // 320 x 16 bit: (symbol | length << 9) in canonical order
constexpr uint32_t kHdrRank = 1280;    // 320 x 16 bit: rank of symbol s among the codes of its length   (dead before kHdrPt is written)
constexpr uint32_t kHdrIn = 1280;      // 1,040 bytes: the staged code-length section of the header           (dead before the table build)
constexpr uint32_t kHdrPt = 1280;      // (1024 + 256) x 16 bit: per first-level prefix of a long code: subtable offset | width << 12
constexpr uint32_t kHdrTab = 3840;     // 2 x (first code[16] | sorted index of the first code[17]) words
static_assert(kHdrTab + 2 * 33 * 4 <= kSyncStage && kHdrPt + 2 * 1280 <= kHdrTab, "header scratch fits the stage");

enum { kSyncEob = 0, kSyncBail = 1, kSyncBailCap = 2 };
#ifndef SWC_PROV_STORES
#define SWC_PROV_STORES 2   // the provisional decode's stores: 0 unconditional, 1 masked per lane (a literal came in / a record is complete), 2 literal groups only when full
#endif
#if defined(SWC_HOST_EMULATION)
// statistics of the emulated decoder (tests, tools/sync_stats.py): rounds committed, bails, lane-passes, symbol iterations
inline uint64_t g_sync_stats[8];
#define SWC_SYNC_STAT(i, n) (g_sync_stats[i] += (n))
// wave-steps: a pass takes as long as its busiest lane.  g_sync_wave[k] adds up, per pass of kind k (0 walk, 1 provisional /
// count, 2 emit), the largest number of code iterations any of the 64 lanes ran; [3]: wave-steps in which some lane met a long code.
inline uint64_t g_sync_iters = 0, g_sync_wave[4] = {0, 0, 0, 0}, g_sync_passmax = 0, g_sync_it0 = 0;
inline uint8_t g_sync_long[4096];
#define SWC_SYNC_ITER() (g_sync_iters++)
#define SWC_SYNC_LONG() (g_sync_long[(g_sync_iters - 1 - g_sync_it0) & 4095] = 1)
#define SWC_SYNC_LANE_BEGIN(first) if (first) { g_sync_passmax = 0; for (auto& h_ : g_sync_long) h_ = 0; } g_sync_it0 = g_sync_iters;
#define SWC_SYNC_LANE_END(last, k) { if (g_sync_iters - g_sync_it0 > g_sync_passmax) g_sync_passmax = g_sync_iters - g_sync_it0; if (last) { g_sync_wave[k] += g_sync_passmax; for (auto h_ : g_sync_long) g_sync_wave[3] += h_; } }
#else
#define SWC_SYNC_STAT(i, n) ((void)0)
#define SWC_SYNC_ITER() ((void)0)
#define SWC_SYNC_LONG() ((void)0)
#define SWC_SYNC_LANE_BEGIN(first)
#define SWC_SYNC_LANE_END(last, k)
#endif
enum { kFlagEob = 1u, kFlagFail = 2u, kFlagTrap = 4u, kFlagSlow = 8u };
// profile builds (-DSWC_PROFILE): cycles per part of one stream -- 0 header, 1 tables, 2 staging, 3 decode passes, 4 chain
// logic + scans, 5 copy / emit, 6 checked steps / rest; 7 rounds, 8 passes, 9 walk pass
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
struct SyncProf {
    uint64_t acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tlast = 0;
};
#define SWC_SP(pp, k) { const uint64_t t_ = __builtin_readcyclecounter(); (pp).acc[k] += t_ - (pp).tlast; (pp).tlast = t_; }
#define SWC_SPC(pp, k, n) ((pp).acc[k] += (n))
#else
struct SyncProf {};
#define SWC_SP(pp, k)
#define SWC_SPC(pp, k, n)
#endif

SWC_HD uint32_t funnel32(uint32_t hi, uint32_t lo, uint32_t sh) {  // bits [sh, sh + 32) of hi:lo; only sh % 32 counts
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u));
#endif
}
SWC_HD uint32_t bfe32(uint32_t v, uint32_t off, uint32_t width) {   // only off % 32 and width % 32 count; width 0 -> 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);
#else
    off &= 31u; width &= 31u;
    return width == 0 ? 0u : (v >> off) & ((1u << width) - 1u);
#endif
}
SWC_HD uint32_t sbfe1(uint32_t v, uint32_t bit) {   // all ones if the bit is set
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_sbfe((int)v, bit, 1u);
#else
    return (v >> bit) & 1u ? 0xFFFFFFFFu : 0u;
#endif
}
SWC_HD uint32_t bfi32(uint32_t m, uint32_t a, uint32_t b) {   // (a & m) | (b & ~m) as v_bfi_b32 (the optimiser turns the C form into a compare and a select when m is a sign mask)
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(a), "v"(b));
    return d;
#else
    return (a & m) | (b & ~m);
#endif
}
SWC_HD uint32_t alignbyte32(uint32_t hi, uint32_t lo, uint32_t nbytes) {   // bytes [n, n + 4) of hi:lo, n = nbytes % 4
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, nbytes);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (nbytes & 3u)));
#endif
}
// (a << K) + b as ONE instruction (v_lshl_add_u32): the optimiser otherwise regroups sums of shifted terms into more of them
template <int K>
SWC_HD uint32_t lshl_add(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(K), "v"(b));
    return d;
#else
    return (a << K) + b;
#endif
}
// (a & m) | o as ONE instruction (v_and_or_b32) with both constants in registers (a VOP3 instruction of gfx9 takes no literal)
SWC_HD uint32_t and_or(uint32_t a, uint32_t m, uint32_t o) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(m), "v"(o));
    return d;
#else
    return (a & m) | o;
#endif
}
// m ? 0 : b for a mask m of all ones or all zeros (v_bfi_b32 with the constant 0: the optimiser turns the C form into a compare and a select)
SWC_HD uint32_t clear_if(uint32_t m, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(d) : "v"(m), "v"(b));
    return d;
#else
    return b & ~m;
#endif
}
// a * k + b, a signed 24-bit multiply-add (v_mad_i32_i24)
SWC_HD uint32_t mad24(uint32_t a, int32_t k, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(k), "v"(b));
    return d;
#else
    return (uint32_t)((int32_t)a * k) + b;
#endif
}

// Where the subtables of the stream are: in LDS (g == nullptr) or in the workspace.
struct SubTab {
    const SWC_AS_GLOBAL uint32_t* g;
};
// The spill of a stream in the workspace
struct Spill {
    SWC_AS_GLOBAL uint32_t* aux;
    SWC_AS_GLOBAL uint16_t* syms;
    SWC_AS_GLOBAL uint32_t* sub;
};

// the two dwords of the stage that hold LDS bit `p`, funnelled: LDS bits [p, p + 32)
SWC_D uint32_t stage_bits(const SyncLds* sl, uint32_t p) {
    const uint32_t* w = (const uint32_t*)(sl->stage + ((p >> 3) & 0x1FFCu));
    return funnel32(w[1], w[0], p);
}
// the entry a LINK leads to
SWC_D uint32_t long_lookup(const SyncLds* sl, const SubTab st, uint32_t bits4, uint32_t link) {
    const uint32_t idx = bfe32(bits4, link >> 10, link >> 5);
    const uint32_t at = (link >> 18) + idx;
    return st.g ? st.g[at] : sl->sub[at < kSubLds ? at : 0u];
}
// the entry of the code that starts at stream bit `pos` of the round (general passes: not the hot loops)
SWC_D uint32_t entry_at(const SyncLds* sl, const SubTab st, uint32_t pos, bool dist, uint32_t& bits4) {
    bits4 = stage_bits(sl, pos);
    uint32_t e = *(const uint32_t*)((const uint8_t*)sl->lut + ((bits4 & (dist ? kDistMask4 : kLitMask4)) | (dist ? kEntLen : 0u)));
    if (ent_is_link(e)) e = long_lookup(sl, st, bits4, e);
    return e;
}

// ---- the checked step (canonical tables in the workspace) --------------------------------------------------------------
// One symbol of the lit/len (LIT) or distance alphabet with the reference's semantics for every code set (DecodingTree.swift:
// 36-50 over the heap Code.swift:15-39 builds -- for an over-subscribed set the shallowest occupied node wins, the last writer
// of a node wins).  Returns the symbol or -1 (symbolNotFound: unassigned path, or the code runs past the end of the input).
template <bool LIT>
SWC_D int decode_sym_canon(BitReader& br, const Spill sp) {
    const SWC_AS_GLOBAL uint32_t* tb = sp.aux + (LIT ? kAuxLit : kAuxDist);
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
    if (idx >= (LIT ? 288u : 32u)) return -1;   // (cannot happen for a set the header accepted; keeps the read inside the array)
    return (int)sp.syms[(LIT ? 0u : kSymDist) + idx];
}

// One symbol with every check of the reference (Deflate.swift:171-236).  Returns SWC_OK to continue, -1 at the
// end-of-block symbol, or the error.
SWC_D int careful_step(Lane& ln, const Spill sp) {
    BitReader& br = ln.br;
    br.refill();
    const int sym = decode_sym_canon<true>(br, sp);
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
    const int dc = decode_sym_canon<false>(br, sp);
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

// ---- tables ------------------------------------------------------------------------------------------------------
// Code.swift:23-37 per length, from the per-length counts cnt[1..15]: into the workspace form (tb: kAux*) and into the two
// small arrays the table build reads (first code and sorted index of the first code per length).  One lane per table.
SWC_D void table_from_counts(const uint32_t* cnt, SWC_AS_GLOBAL uint32_t* tb, uint32_t* first, uint32_t* start) {
    uint32_t v = 0, off = 0, over = 0;
    tb[kAuxLim] = 0; tb[kAuxSlot] = 0; tb[kAuxStart] = 0;
    first[0] = 0; start[0] = 0;
#pragma unroll 1
    for (int d = 1; d <= 15; d++) {
        const uint32_t c = cnt[d];
        tb[kAuxLim + d] = (v + c) << (15 - d);
        if (c != 0 && v + c > (1u << d)) over = 1;
        tb[kAuxSlot + d] = (off - v) & 0xFFFFu;
        tb[kAuxStart + d] = off;
        first[d] = v;
        start[d] = off;
        off += c;
        v = (v + c) << 1;
    }
    tb[kAuxCount] = off;
    tb[kAuxOver] = over;
    start[16] = off;
}

// From the code lengths lens[0 .. literals + distances) in the header scratch: the canonical tables and the (length,
// symbol)-sorted symbol arrays in the workspace (what the checked step decodes with) and -- unless a set is over-subscribed,
// which only the checked step decodes -- the direct tables and the subtables of the long codes.  All lanes.  Returns false
// if the fast path cannot take the block (over-subscribed, or subtables beyond the overflow area).
//   A  rank of every symbol among the codes of its length, 64 symbols at a time: the lanes with the same length find each
//      other with four ballots (one per bit of the length), rank = the number of those in front of me + the running count;
//   B  first code / first sorted index per length (one lane per alphabet), scatter of the symbols into canonical order;
//   C  in canonical order: codes that fit the direct table fill their slots; a longer code belongs to the subtable of its
//      first 10 / 8 bits, whose width is set by the LAST (longest) code with that prefix -- prefixes do not decrease in
//      canonical order, so "last" is a look at the next symbol -- and whose offset is a running sum over those;
//   D  the long codes fill their subtable slots.
SWC_D bool sync_tables_from_lengths(SyncLds* sl, const Spill sp, int literals, int distances, SubTab& st, SyncProf& pf) {
    SWC_SP(pf, 0)
    using simt::PT;
    constexpr int N = kWave;
    const uint8_t* lens = sl->stage + kHdrLens;
    uint32_t* run = (uint32_t*)(sl->stage + kHdrRun);
    uint16_t* sorted = (uint16_t*)(sl->stage + kHdrSorted);
    uint16_t* rankb = (uint16_t*)(sl->stage + kHdrRank);
    uint16_t* pt = (uint16_t*)(sl->stage + kHdrPt);
    uint32_t* tab = (uint32_t*)(sl->stage + kHdrTab);   // per alphabet: first[16] | start[17]
    const int total = literals + distances;
    st.g = nullptr;
    // ---- A
    SIMT_BEGIN(t, N) if (t < 32) run[t] = 0; SIMT_END_WAVE
    PT<uint32_t, N> v, upd;
    PT<uint64_t, N> m;
    PT<bool, N> p;
#pragma unroll 1
    for (int g = 0; g * N < literals + N; g++) {   // the groups of the lit/len alphabet, then the one group of the distance alphabet
        const bool is_dist = g * N >= literals;
        const int s0 = is_dist ? literals : g * N, s_end = is_dist ? total : literals;
        SIMT_BEGIN(t, N) v[t] = s0 + t < s_end ? (uint32_t)lens[s0 + t] : 0u; m[t] = ~0ull; SIMT_END
#pragma unroll
        for (int k = 0; k < 4; k++) {
            SIMT_BEGIN(t, N) p[t] = ((v[t] >> k) & 1u) != 0u; SIMT_END
            const uint64_t b = simt::wave_ballot<N>(p);
            SIMT_BEGIN(t, N) m[t] &= p[t] ? b : ~b; SIMT_END
        }
        SIMT_BEGIN(t, N)
            upd[t] = 0;
            if (v[t] != 0u) {
                const uint32_t rank = (uint32_t)simt::popc64(m[t] & ((1ull << t) - 1ull)), cnt = (uint32_t)simt::popc64(m[t]);
                const uint32_t base = run[(is_dist ? 16u : 0u) + v[t]];
                rankb[s0 + t] = (uint16_t)(base + rank);
                if (rank == cnt - 1u) upd[t] = base + cnt;   // the last lane of the length moves the running count on
            }
        SIMT_END_WAVE
        SIMT_BEGIN(t, N) if (upd[t] != 0u) run[(is_dist ? 16u : 0u) + v[t]] = upd[t]; SIMT_END_WAVE
    }
    SWC_SP(pf, 10)
    // ---- B
    SIMT_BEGIN(t, N)
        if (t < 2) table_from_counts(run + 16 * t, sp.aux + (t ? kAuxDist : kAuxLit), tab + 33 * t, tab + 33 * t + 16);
    SIMT_END_WAVE
    SIMT_BEGIN(t, N)
        for (int s = t; s < total; s += N) {
            const uint32_t d = lens[s];
            if (d) {
                const bool dd = s >= literals;
                const uint32_t j = (dd ? kSymDist : 0u) + tab[(dd ? 33 : 0) + 16 + d] + rankb[s];
                const uint32_t sym = (uint32_t)(s - (dd ? literals : 0));
                if (j < kSymDist + 32u) { sorted[j] = (uint16_t)(sym | (d << 9)); sp.syms[j] = (uint16_t)sym; }
            }
        }
        // the direct tables start out as "no symbol the fast path takes"
        for (int i = t; i < (1 << kSyncLitBits) + (1 << kSyncDistBits); i += N) sl->lut[i] = i < (1 << kSyncLitBits) ? kInvLit : kInvDist;
    SIMT_END_WAVE
    simt::vmem_fence();   // (the wave reads the spill back -- here and in the checked step: its stores are done first)
    SWC_SP(pf, 11)
    if (sp.aux[kAuxLit + kAuxOver] != 0u || sp.aux[kAuxDist + kAuxOver] != 0u) return false;
    // ---- C
    uint32_t used = 0, used_lit = 0;
    PT<uint32_t, N> sz, rv, ln2;
#pragma unroll 1
    for (int tbl = 0; tbl < 2; tbl++) {
        const uint32_t root = tbl ? (uint32_t)kSyncDistBits : (uint32_t)kSyncLitBits, rmask = (1u << root) - 1u;
        const uint32_t* first = tab + 33 * tbl;
        const uint32_t* start = first + 16;
        const uint32_t n_sym = start[16] < (tbl ? 32u : 288u) ? start[16] : (tbl ? 32u : 288u);
        const uint16_t* srt = sorted + (tbl ? kSymDist : 0u);
        uint32_t* lut = sl->lut + (tbl ? (1 << kSyncLitBits) : 0);
        uint16_t* ptt = pt + (tbl ? (1 << kSyncLitBits) : 0);
#pragma unroll 1
        for (uint32_t j0 = 0; j0 < n_sym; j0 += (uint32_t)N) {
            SIMT_BEGIN(t, N)
                const uint32_t j = j0 + (uint32_t)t;
                sz[t] = 0; ln2[t] = 0; rv[t] = 0;
                if (j < n_sym) {
                    const uint32_t w = srt[j], sym = w & 511u, d = w >> 9;
                    const uint32_t code = first[d] + (j - start[d]);
                    const uint32_t rev = brev32(code) >> (32u - d);
                    rv[t] = rev; ln2[t] = d;
                    if (d <= root) {
                        const uint32_t entry = entry_of_symbol(tbl != 0, sym, d);
                        for (uint32_t q = rev; q <= rmask; q += 1u << d) lut[q] = entry;
                    } else {
                        bool last = j + 1u == n_sym;
                        if (!last) {
                            const uint32_t w2 = srt[j + 1u], d2 = w2 >> 9;
                            const uint32_t rev2 = brev32(first[d2] + (j + 1u - start[d2])) >> (32u - d2);
                            last = (rev2 & rmask) != (rev & rmask);
                        }
                        if (last) sz[t] = 1u << (d - root);
                    }
                }
            SIMT_END
            PT<uint32_t, N> x = sz;
            simt::wave_scan_incl<N>(x);
            SIMT_BEGIN(t, N)
                if (sz[t] != 0u) {
                    const uint32_t at = used + x[t] - sz[t], k = ln2[t] - root, pre = rv[t] & rmask;
                    if (at + sz[t] <= kSubMax) {
                        ptt[pre] = (uint16_t)(at | (k << 12));
                        lut[pre] = (k << 5) | ((root + 2u) << 10) | (at << 18);
                    }
                }
            SIMT_END
            used += simt::wave_read<N>(x, N - 1);
        }
        if (tbl == 0) used_lit = used;
    }
    SWC_SP(pf, 12)
    if (used > kSubMax) return false;
    SWC_AS_GLOBAL uint32_t* subg = used > kSubLds ? sp.sub : nullptr;
    st.g = subg;
    simt::wave_fence();
    // ---- D: the subtables start out as "no symbol", then every long code fills its slots
    SIMT_BEGIN(t, N)
        for (uint32_t i = (uint32_t)t; i < used; i += (uint32_t)N) {
            const uint32_t inv = i < used_lit ? kInvLit : kInvDist;
            if (subg) subg[i] = inv; else sl->sub[i] = inv;
        }
    SIMT_END_WAVE
    if (subg) simt::vmem_fence();
#pragma unroll 1
    for (int tbl = 0; tbl < 2; tbl++) {
        const uint32_t root = tbl ? (uint32_t)kSyncDistBits : (uint32_t)kSyncLitBits, rmask = (1u << root) - 1u;
        const uint32_t* first = tab + 33 * tbl;
        const uint32_t* start = first + 16;
        const uint32_t n_sym = start[16] < (tbl ? 32u : 288u) ? start[16] : (tbl ? 32u : 288u);
        if (start[root + 1] >= n_sym) continue;    // no code longer than the direct table
        const uint16_t* srt = sorted + (tbl ? kSymDist : 0u);
        const uint16_t* ptt = pt + (tbl ? (1 << kSyncLitBits) : 0);
        SIMT_BEGIN(t, N)
            for (uint32_t j = start[root + 1] + (uint32_t)t; j < n_sym; j += (uint32_t)N) {
                const uint32_t w = srt[j], sym = w & 511u, d = w >> 9;
                const uint32_t rev = brev32(first[d] + (j - start[d])) >> (32u - d);
                const uint32_t entry = entry_of_symbol(tbl != 0, sym, d);
                const uint32_t pw = ptt[rev & rmask], at = pw & 0xFFFu, k = pw >> 12;
                for (uint32_t q = rev >> root; q < (1u << k); q += 1u << (d - root)) {
                    if (subg) subg[at + q] = entry; else sl->sub[at + q] = entry;
                }
            }
        SIMT_END
    }
    simt::wave_fence();
    if (subg) simt::vmem_fence();
    SWC_SP(pf, 13)
    return true;
}

// ---- one sub-chunk: where does a decode from `start` end?  (The walk pass.) -----------------------------------------------
// Positions are stream bits relative to the round's base.  Returns the position just past the last symbol taken, or kPosFail.
// The loop: window address (2 instructions), funnel, table address, [LDS], position mask, link test, position, the two table
// selectors, compare -- ten vector instructions per code.
SWC_D uint32_t walk_chunk(const SyncLds* sl, const SubTab st, uint32_t start, uint32_t chunk_end, uint32_t in_bits) {
    if (chunk_end > in_bits) chunk_end = in_bits;          // (the zero fill behind the input is not worth decoding)
    if (start >= in_bits) return kPosFail;
    uint32_t pos = start, tm = kLitMask4, tb = 0, e = 0;
    uint32_t c_notlen = kEntNotLen, c_dmask = kDistMask4;
    SWC_OPAQUE(c_notlen); SWC_OPAQUE(c_dmask);
#if SWC_WALK_WINDOW
    // The walk is bound by the latency of its chain, not by its instruction count (four waves per SIMD, two dependent LDS reads
    // per code in the form below): here the window lives in two registers, the dword behind it is read every step -- needed or
    // not, off the chain -- and moves in when the position crosses a dword: ONE LDS read in the chain, fourteen instructions.
    uint32_t d0, d1, bp = pos & 31u;
    { const uint32_t* w = (const uint32_t*)(sl->stage + ((pos >> 3) & 0x1FFCu)); d0 = w[0]; d1 = w[1]; }
    if (start < chunk_end) do {
        SWC_SYNC_STAT(4, 1);
        SWC_SYNC_ITER();
        const uint32_t bits4 = funnel32(d1, d0, pos);
        e = *(const uint32_t*)((const uint8_t*)sl->lut + and_or(bits4, tm, tb));
        const uint32_t nx = *(const uint32_t*)(sl->stage + ((pos >> 3) & 0x1FFCu) + 8u);
        uint32_t mm = e & kEntPosMask;
        if (mm == 0u) { SWC_SYNC_STAT(7, 1); SWC_SYNC_LONG(); e = long_lookup(sl, st, bits4, e); mm = e & kEntPosMask; }
        pos = pos + mm + 0xFFFFFFFEu;
        const uint32_t bpn = pos & 31u;
        const bool crossed = bpn < bp;                      // (a symbol takes fewer than 32 bits)
        d0 = crossed ? d1 : d0;
        d1 = crossed ? nx : d1;
        bp = bpn;
        tm = and_or(e, c_notlen, c_dmask);
        tb = e & kEntLen;
    } while ((int32_t)pos < (int32_t)chunk_end);
#else
    if (start < chunk_end) do {   // (tested at the bottom: one mask update and one branch per iteration)
        SWC_SYNC_STAT(4, 1);
        SWC_SYNC_ITER();
        const uint32_t bits4 = stage_bits(sl, pos);
        e = *(const uint32_t*)((const uint8_t*)sl->lut + and_or(bits4, tm, tb));
        uint32_t mm = e & kEntPosMask;
        if (mm == 0u) { SWC_SYNC_STAT(7, 1); SWC_SYNC_LONG(); e = long_lookup(sl, st, bits4, e); mm = e & kEntPosMask; }
        pos = pos + mm + 0xFFFFFFFEu;
        tm = and_or(e, c_notlen, c_dmask);
        tb = e & kEntLen;
    } while ((int32_t)pos < (int32_t)chunk_end);
#endif
    const uint32_t endb = pos & 0x3FFFFFFFu;
    const bool fail = ((pos & kEntStop) != 0u && !ent_is_eob(e)) || endb > in_bits;
    return fail ? kPosFail : endb;
}

// ---- one sub-chunk, decoded ONCE into the round's scratch ------------------------------------------------------------------
// The count pass and the emit pass of a round in one: the lane decodes its sub-chunk from `start` and writes its literals
// and records to its column of the stream's scratch area (lz_resolve.h: rows across the lanes, sized for the worst case, so
// the loop needs no bounds test), counting as it goes; the round then scans the counts and COPIES every lane's piece to its
// final offset (copy_prov) instead of decoding a third time.  What the lane cannot know yet -- the output position of its
// sub-chunk -- enters only through `need`: the largest (distance - 1 - output bytes of the sub-chunk in front of the match),
// checked after the scan (NEED = false: the round starts 32 KiB or more into the output, where no distance can fail).
// Both stores of a step are UNCONDITIONAL: the literal accumulator (the newest byte at the top, four to a group) goes to the
// row of the group it belongs to, a record to the row of the next record -- a step that adds nothing rewrites what is there,
// and a row is final when its last writer has been.  `run0`: literals in front of the sub-chunk that no record covers yet
// (lane 0 of the first round of a block).  A sub-chunk that ends at the end-of-block symbol leaves its trailing literals
// uncovered (`tail`), every other one closes them with a literal-only record.
struct ProvOut {
    uint32_t end, nlit, nrec, nout, flags, tail;
    int32_t need;
};
constexpr uint32_t kProvRow = 64u * 4u;   // bytes from one row of the scratch to the next (records and literal groups alike)

template <bool NEED>
SWC_D void decode_chunk_prov(const SyncLds* sl, const SubTab st, uint32_t start, uint32_t chunk_end, uint32_t in_bits, gptr prov,
                             uint32_t lane, uint32_t run0, ProvOut& r) {
    // (one base pointer for the wave and 32-bit offsets per lane: the stores take the base from scalar registers)
    const uint32_t rbase = 4u * lane, lbase = (uint32_t)lzr::kProvRecBytes + 4u * lane;
    uint32_t pos = start, tm = kLitMask4, tb = 0, e = 0;
    uint32_t plen = 0, run = run0, nl3 = 3, nout = 0, lb = 0;
    uint32_t roff = rbase + kProvRow;                      // byte offset of my next record in the scratch (row 1 is the first)
    int32_t need = -0x40000000;
    uint32_t c_notlen = kEntNotLen, c_dmask = kDistMask4;
    int32_t c_m256 = -256;
    SWC_OPAQUE(c_notlen); SWC_OPAQUE(c_dmask); SWC_OPAQUE(c_m256);
    if (chunk_end > in_bits) chunk_end = in_bits;          // (the zero fill behind the input is not worth decoding)
    const bool dead = start >= in_bits;                    // nothing left for this sub-chunk: the checked step says what that means
    // (the window of the NEXT step is read as soon as the position is known: its LDS latency runs under the bookkeeping)
    uint32_t w0 = 0, w1 = 0;
    if (!dead && start < chunk_end) { const uint32_t* w = (const uint32_t*)(sl->stage + ((pos >> 3) & 0x1FFCu)); w0 = w[0]; w1 = w[1]; }
    if (!dead && start < chunk_end) do {                   // (tested at the bottom: one mask update and one branch per iteration)
        SWC_SYNC_STAT(5, 1);
        SWC_SYNC_ITER();
        const uint32_t bits4 = funnel32(w1, w0, pos);
        e = *(const uint32_t*)((const uint8_t*)sl->lut + and_or(bits4, tm, tb));
        uint32_t mm = e & kEntPosMask;
        if (mm == 0u) { SWC_SYNC_STAT(7, 1); SWC_SYNC_LONG(); e = long_lookup(sl, st, bits4, e); mm = e & kEntPosMask; }
        pos = pos + mm + 0xFFFFFFFEu;
        { const uint32_t* w = (const uint32_t*)(sl->stage + ((pos >> 3) & 0x1FFCu)); w0 = w[0]; w1 = w[1]; }
        tm = and_or(e, c_notlen, c_dmask);
        tb = e & kEntLen;
        const uint32_t val = (bfe32(bits4, 0, e) >> ((e >> 5) & 31u)) + bfe32(e, kEntBaseShift, 15);
        // a literal enters the accumulator at the top (a shift by 0 bytes leaves it alone); the group it belongs to is rewritten
        const uint32_t lit1 = bfe32(e, 13, 1);
        lb = alignbyte32(val, lb, lit1);
        nl3 += lit1;
        const uint32_t dm = sbfe1(e, 29);
#if SWC_PROV_STORES == 0
        store_u32(prov + lshl_add<6>(nl3 & ~3u, lbase), lb);
        // a distance completes a record; any other step writes a word that the next record of the lane overwrites
        store_u32(prov + roff, lshl_add<7>(plen, lshl_add<16>(val, run)));
#elif SWC_PROV_STORES == 1
        // (the stores of the lanes that have nothing new are masked off: the memory pipeline's work is per active lane)
        if (lit1 != 0u) store_u32(prov + lshl_add<6>(nl3 & ~3u, lbase), lb);
        if (dm != 0u) store_u32(prov + roff, lshl_add<7>(plen, lshl_add<16>(val, run)));
#else
        if (((nl3 & 3u) | (lit1 << 2)) == 7u) store_u32(prov + lshl_add<6>(nl3 & ~3u, lbase), lb);   // a literal came in and completed its group (nl3 % 4 == 3)
        if (dm != 0u) store_u32(prov + roff, lshl_add<7>(plen, lshl_add<16>(val, run)));
#endif
        roff = mad24(dm, c_m256, roff);
        if (NEED) {
            const int32_t nd = (int32_t)val - (int32_t)nout;   // this much output must exist in front of the sub-chunk
            need = (int32_t)bfi32(dm, (uint32_t)(nd > need ? nd : need), (uint32_t)need);
        }
        nout += lit1 + plen;                               // (plen is zero except in the step of the distance)
        run = clear_if(dm, run + lit1);
        plen = val & sbfe1(e, kEntLenBit);
    } while ((int32_t)pos < (int32_t)chunk_end);
    const uint32_t nlit = nl3 - 3u;
#if SWC_PROV_STORES == 2
    if (nlit & 3u) store_u32(prov + lshl_add<6>(nl3 & ~3u, lbase), lb);   // the last, incomplete group
#endif
    const uint32_t endb = pos & 0x3FFFFFFFu;
    uint32_t flags = 0;
    if (dead) flags = kFlagFail;
    else if (pos & kEntStop) flags = ent_is_eob(e) ? kFlagEob : kFlagFail;
    uint32_t tail = 0;
    if (run > 0) {
        if (flags & kFlagEob) tail = run;
        else { store_u32(prov + roff, lzr::make_lits(run)); roff += kProvRow; }
    }
    if (endb > in_bits) flags |= kFlagFail;                // a symbol ran past the end of the input
    if (nlit + run0 > lzr::kLitRunMax) flags |= kFlagSlow;  // (conservative: only then can a run exceed what a match record carries)
    r.end = (flags & kFlagFail) ? kPosFail : endb;
    r.nlit = nlit; r.nrec = (roff - rbase) / kProvRow - 1u; r.nout = nout; r.flags = flags; r.tail = tail; r.need = need;
}

// A lane's piece of the round moves from its column of the scratch to its final place: `nrec` records to `rdst` (dword
// aligned), `nlit` literal bytes to `ldst` (any alignment).  The loads of a step read one row: coalesced.  The last,
// incomplete literal group holds its bytes at the top.
SWC_D void copy_prov(gcptr plit, gcptr prec, uint32_t nlit, uint32_t nrec, gptr ldst, SWC_AS_GLOBAL uint32_t* rdst) {
    // kCopyRec records and kCopyLit literal groups are loaded per step, all before the first store (the scratch of all
    // resident waves exceeds the L2, so a load takes its several hundred cycles: one load per step would expose that latency
    // forty times per round; a step lasts as long as the slowest lane's, so the sizes aim at ONE step for a sub-chunk of text --
    // 23 records and 19 literals on average).  Rows past the lane's count hold something and exist (the scratch is sized for
    // the worst case): they are loaded and not stored.
    constexpr uint32_t kCopyRec = SWC_COPY_REC, kCopyLit = SWC_COPY_LIT;
    const uint32_t ngrp = (nlit + 3u) >> 2;
    for (uint32_t i = 0, g = 0; i < nrec || g < ngrp; i += kCopyRec, g += kCopyLit) {
        uint32_t v[kCopyRec], w[kCopyLit];
#pragma unroll
        for (uint32_t k = 0; k < kCopyRec; k++) {
            const uint32_t row = i + k + 1u < (uint32_t)lzr::kProvRecRows ? i + k + 1u : (uint32_t)lzr::kProvRecRows - 1u;
            v[k] = load_u32(prec + (size_t)row * kProvRow);
        }
#pragma unroll
        for (uint32_t k = 0; k < kCopyLit; k++) {
            const uint32_t row = g + k + 1u < (uint32_t)lzr::kProvLitRows ? g + k + 1u : (uint32_t)lzr::kProvLitRows - 1u;
            w[k] = load_u32(plit + (size_t)row * kProvRow);
        }
        // (wide stores: the lanes' destinations lie apart, so the memory pipeline takes a store lane by lane -- four records or
        // four groups per lane and instruction instead of one)
#pragma unroll
        for (uint32_t k = 0; k < kCopyRec; k += 4) {
            if (i + k + 4u <= nrec) store_u128_a4((gptr)(rdst + i + k), v[k], v[k + 1], v[k + 2], v[k + 3]);
            else {
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) if (i + k + q < nrec) rdst[i + k + q] = v[k + q];
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < kCopyLit; k += 4) {
            const uint32_t at = 4u * (g + k);
            if (at + 16u <= nlit) store_u128_a4(ldst + at, w[k], w[k + 1], w[k + 2], w[k + 3]);
            else {
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    const uint32_t aq = at + 4u * q;
                    if (aq + 4u <= nlit) store_u32(ldst + aq, w[k + q]);
                    else if (aq < nlit) { uint32_t x = w[k + q] >> (8u * (4u - (nlit - aq))); for (uint32_t z = aq; z < nlit; z++, x >>= 8) ldst[z] = (uint8_t)x; }
                }
            }
        }
    }
}

// ---- one sub-chunk, the general passes ------------------------------------------------------------------------------------
struct ChunkOut {
    uint32_t end;     // bit position (relative to the round base) just past the last symbol taken; kPosFail: no valid decode
    uint32_t nlit, nrec, nout;
    uint32_t flags;
};
// MODE 0: count.  MODE 1: emit literals and records.  MODE 2: check distances only (output beyond the capacity).
// BIG: literal runs of more than lzr::kLitRunMax bytes in front of a match get a record of their own (only possible in a
// sub-chunk that holds more than that many literals: the caller picks).  CHK: symbols may run past the end of the input.
// Decodes from bit `start` until a lit/len symbol would begin at or beyond `chunk_end`, or the end-of-block symbol.
// (Rounds the provisional decode abandoned, and lanes behind an early end of block: not the hot path.)
template <int MODE, bool BIG, bool CHK>
SWC_D void decode_chunk(const SyncLds* sl, const SubTab st, uint32_t start, uint32_t chunk_end, uint32_t in_bits, gptr lit_dst,
                        SWC_AS_GLOBAL uint32_t* rec_dst, uint64_t out_pos0, ChunkOut& r) {
    uint32_t pos = start;
    bool dist_next = false;
    uint32_t plen = 0, run = 0, nlit = 0, nrec = 0, nout = 0, flags = 0;
    uint32_t lb = 0, lbn = 0;                              // literals on their way to the literal stream, four per store
    const uint32_t room = out_pos0 > 0x40000000ull ? 0x40000000u : (uint32_t)out_pos0;   // output in front of the sub-chunk, as far as a distance can reach
    for (;;) {
        if (!dist_next && pos >= chunk_end) break;
        SWC_SYNC_STAT(4 + MODE, 1);
        SWC_SYNC_ITER();
        uint32_t bits4;
        const uint32_t e = entry_at(sl, st, pos, dist_next, bits4);
        const uint32_t n = ent_bits(e);
        if ((e & kEntStop) || (CHK && pos + n > in_bits)) {
            // not a symbol the fast path takes / the symbol runs past the end of the input / the end of the block
            if (!ent_is_eob(e) || (CHK && pos + n > in_bits)) flags |= kFlagFail;
            else { pos += n; flags |= kFlagEob; }
            break;
        }
        const bool was_dist = dist_next;
        const uint32_t is_lit = (e >> 13) & 1u;
        const uint32_t val = (bfe32(bits4, 0, e) >> ((e >> 5) & 31u)) + bfe32(e, kEntBaseShift, 15) + (was_dist ? 1u : 0u);   // literal, length or distance
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
        dist_next = (e & kEntLen) != 0u;
        pos += n;
    }
    if (run > 0) {   // the sub-chunk closes its literal run itself
        nrec++;
        if (MODE == 1) *rec_dst++ = lzr::make_lits(run);
    }
    if (MODE == 1) {
        for (uint32_t i = 0; i < lbn; i++) lit_dst[i] = (uint8_t)(lb >> (8 * (4u - lbn + i)));   // the pending bytes sit at the top
    }
    r.end = (flags & kFlagFail) ? kPosFail : pos;
    r.nlit = nlit; r.nrec = nrec; r.nout = nout;
    r.flags = flags;
}

// ---- a TEAM of wavefronts on one stream -----------------------------------------------------------------------------
// A launch of few streams is a matter of latency: one wave takes a round of a 64 KiB stream in 50 us whatever else the chip
// does, and the stream has five.  A team is a workgroup of kTeamWaves wavefronts on ONE stream: wave 0, the MASTER, runs the
// job as above; while it works on the round at byte B of the block, HELPER h stages the round at B + h * kTeamStride -- its
// sub-chunk 0 is the sub-chunk 63 of the round in front of it -- with the master's tables (copied into its own LDS once per
// block), walks all 64 sub-chunks, lane 0 from the first bit of its sub-chunk, and decodes sub-chunks 1 .. 63 into its own
// scratch rows exactly as the master's fast path does.  The helper cannot know whether its lane 0 was in step at the end of its
// sub-chunk (after 544 bits: 99.3 %), and it need not: the master, having committed the round in front, ADOPTS the helper's
// round if and only if the helper's lane 1 began at the bit where the committed round ended -- a chain of decodes that starts
// at the true position IS the true sequence -- and otherwise simply decodes on from there itself.  What the master needs for that is a
// dozen numbers per helper (the totals, the largest distance shortfall, where the round ended): the adoption is serial but scalar,
// and the helpers then copy their own rows to the offsets the master names, all at once.  Three workgroup barriers per super-round
// (rounds open / rounds done / copy); the helpers wait in the first while the master parses a header or takes checked steps.
#ifndef SWC_TEAM_WAVES
#define SWC_TEAM_WAVES 6
#endif
#if defined(SWC_HOST_EMULATION)
inline uint64_t g_team_adopted = 0;
#define SWC_TEAM_STAT() (g_team_adopted++)
#else
#define SWC_TEAM_STAT() ((void)0)
#endif
constexpr int kTeamWaves = SWC_TEAM_WAVES;
constexpr uint32_t kTeamStride = 63u * kSyncChunk;
constexpr size_t kTeamProvBytes = lzr::kProvRecBytes + lzr::kProvLitBytes;   // a helper's scratch rows (no table spill of its own)
struct TeamRound {   // what a helper hands back; lanes [1, nv) are its chain
    uint32_t c_lit[kWave], c_rec[kWave];   // per lane: literal bytes, records (the helper copies them itself when it is adopted)
    uint32_t nv, eob, ok;
    uint32_t start1;                       // where its lane 1 began (bits from its round's base)
    uint32_t end_last, tail_last;          // where lane nv - 1 ended; the literals behind its last record (a round that ends the block)
    uint32_t tot_lit, tot_rec, tot_out;
    uint32_t mneed;                        // the largest (distance - 1 - output in front of the match inside the round), biased by 2^31
    // the master's answer
    uint32_t adopt, nrec_base;
    uint64_t nlit_base;
};
struct TeamShared {
    uint32_t cmd;                    // 1: a round for the helpers, 0: the job is over
    uint32_t gen;                    // changes with every set of tables the master builds
    uint32_t B, in_len;              // the master's round; the stream
    uint64_t in, subg;               // the stream; the subtable overflow in the workspace (SubTab::g)
    uint64_t lits, recs;             // the stream's literal stream and record list (the helpers copy into them)
    uint32_t hgen[kTeamWaves];       // the tables helper h holds
    TeamRound r[kTeamWaves - 1];
};
struct Team {
    TeamShared* sh = nullptr;        // nullptr: the wave works alone
    SyncLds* lds = nullptr;          // kTeamWaves of them, [0] the master's
    gptr scratch = nullptr;          // helper h's rows at (h - 1) * kTeamProvBytes
    int helpers = 0;
    uint32_t gen = 0;
};

// one round of helper h (h = 1 .. helpers): see above
SWC_D void team_helper_round(const Team& tm, int h) {
    using simt::PT;
    constexpr int N = kWave;
    TeamShared* sh = tm.sh;
    SyncLds* sl = tm.lds + h;
    TeamRound& R = sh->r[h - 1];
    const uint32_t in_len = simt::uniform(sh->in_len);
    const uint64_t B64 = (uint64_t)simt::uniform(sh->B) + (uint64_t)h * kTeamStride;
    gcptr in = (gcptr)(uintptr_t)simt::uniform(sh->in);
    const SubTab st{(const SWC_AS_GLOBAL uint32_t*)(uintptr_t)simt::uniform(sh->subg)};
    gptr prov = tm.scratch + (size_t)(h - 1) * kTeamProvBytes;
    if (B64 + 2u * kSyncChunk >= (uint64_t)in_len) {   // nothing behind its sub-chunk 0
        SIMT_BEGIN(t, N) if (t == 0) { R.ok = 0; R.nv = 0; R.eob = 0; R.adopt = 0; } SIMT_END_WAVE
        return;
    }
    const uint32_t B = (uint32_t)B64;
    if (simt::uniform(sh->hgen[h]) != simt::uniform(sh->gen)) {   // the master's tables, once per block
        const SyncLds* msl = tm.lds;
        constexpr int kLutWords = (int)(sizeof(sl->lut) / 4), kSubWords = (int)(sizeof(sl->sub) / 4);
        SIMT_BEGIN(t, N)
            for (int i = t; i < kLutWords; i += N) sl->lut[i] = msl->lut[i];
            for (int i = t; i < kSubWords; i += N) sl->sub[i] = msl->sub[i];
            if (t == 0) sh->hgen[h] = sh->gen;
        SIMT_END_WAVE
    }
    const uint64_t left = (uint64_t)(in_len - B) * 8;
    const uint32_t in_bits = left > 0x3FFFFFFFull ? 0x3FFFFFFFu : (uint32_t)left;
    PT<uint32_t, N> start, endp, pe, c_lit, c_rec, c_out, flg, c_tail, c_need;
    PT<bool, N> todo, pb, have;
    {   // stage [B, B + kSyncStage) shifted left by two bits, zero-filled beyond the input (as sync_block does)
        constexpr uint32_t kParts = (kSyncStage + 16u * N - 1u) / (16u * N);
        PT<uint64_t, N> sa[kParts], sb[kParts];
        PT<uint32_t, N> spv[kParts];
        SIMT_BEGIN(t, N)
#pragma unroll
            for (uint32_t k = 0; k < kParts; k++) {
                const uint32_t o = 16u * (uint32_t)t + k * 16u * N;
                uint64_t a = 0, b = 0;
                uint32_t prev = 0;
                const uint64_t at = (uint64_t)B + o;
                if (o < kSyncStage) {
                    if (at + 16 <= in_len) {
                        a = load_u64(in + at); b = load_u64(in + at + 8);
                        if (o != 0) prev = load_u32(in + at - 4);
                    } else {
                        for (uint32_t q = 0; q < 8; q++) if (at + q < in_len) a |= (uint64_t)in[at + q] << (8 * q);
                        for (uint32_t q = 0; q < 8; q++) if (at + 8 + q < in_len) b |= (uint64_t)in[at + 8 + q] << (8 * q);
                        if (o != 0) for (uint32_t q = 0; q < 4; q++) if (at - 4 + q < in_len) prev |= (uint32_t)in[at - 4 + q] << (8 * q);
                    }
                }
                sa[k][t] = a; sb[k][t] = b; spv[k][t] = prev;
            }
        SIMT_END
        SIMT_BEGIN(t, N)
#pragma unroll
            for (uint32_t k = 0; k < kParts; k++) {
                const uint32_t o = 16u * (uint32_t)t + k * 16u * N;
                if (o < kSyncStage) {
                    uint32_t* st32 = (uint32_t*)(sl->stage + o);
                    const uint64_t a = sa[k][t], b = sb[k][t];
                    const uint32_t w0 = (uint32_t)a, w1 = (uint32_t)(a >> 32), w2 = (uint32_t)b, w3 = (uint32_t)(b >> 32);
                    st32[0] = funnel32(w0, spv[k][t], 30); st32[1] = funnel32(w1, w0, 30); st32[2] = funnel32(w2, w1, 30); st32[3] = funnel32(w3, w2, 30);
                }
            }
            start[t] = t == 0 ? 0u : (uint32_t)t * kSyncChunk * 8u - kSyncWalkBack;
        SIMT_END_WAVE
    }
    SIMT_BEGIN(t, N)
        const uint32_t ce = ((uint32_t)t + 1u) * kSyncChunk * 8u;
        endp[t] = walk_chunk(sl, st, start[t], ce, in_bits);
        flg[t] = 0; have[t] = t == 0;          // lane 0 only walks: its sub-chunk belongs to the round in front
        c_lit[t] = c_rec[t] = c_out[t] = c_tail[t] = 0; c_need[t] = 0;
    SIMT_END
    uint32_t nv = 0;
    bool ok = true, eob = false;
    for (;;) {
        simt::wave_shift_up<N>(pe, endp, 0u);
        SIMT_BEGIN(t, N) pb[t] = !(have[t] && (t == 0 || start[t] == pe[t])); SIMT_END
        const uint64_t m_bad = simt::wave_ballot<N>(pb);
        const int b = m_bad ? simt::ctz64(m_bad) : 64;
        const uint64_t chain = b == 64 ? ~0ull : (1ull << b) - 1ull;
        SIMT_BEGIN(t, N) pb[t] = (flg[t] & kFlagEob) != 0; SIMT_END
        const uint64_t m_eob = simt::wave_ballot<N>(pb) & chain;
        const int E = m_eob ? simt::ctz64(m_eob) : 64;
        nv = (uint32_t)(E < 64 ? E + 1 : b);
        SIMT_BEGIN(t, N) pb[t] = (flg[t] & (kFlagFail | kFlagSlow)) != 0 || (t == 0 && endp[t] == kPosFail); SIMT_END
        if (simt::wave_ballot<N>(pb) & (nv == 64 ? ~0ull : (1ull << nv) - 1ull)) { ok = false; break; }
        if (E < 64) { eob = true; break; }
        if (b == 64) break;
        SIMT_BEGIN(t, N)
            todo[t] = t != 0 && pe[t] != kPosFail && (start[t] != pe[t] || !have[t]);
            if (todo[t]) {
                start[t] = pe[t];
                ProvOut r;
                const uint32_t ce = ((uint32_t)t + 1u) * kSyncChunk * 8u;
                if (start[t] + kSyncChunk * 8u < ce) { r.end = kPosFail; r.nlit = r.nrec = r.nout = r.tail = 0; r.flags = kFlagFail; r.need = 0; }
                else decode_chunk_prov<true>(sl, st, start[t], ce, in_bits, prov, (uint32_t)t, 0u, r);
                endp[t] = r.end; c_lit[t] = r.nlit; c_rec[t] = r.nrec; c_out[t] = r.nout; flg[t] = r.flags;
                c_tail[t] = r.tail; c_need[t] = (uint32_t)r.need;
                have[t] = true;
            }
        SIMT_END
    }
    // totals over the chain, and the largest shortfall of a distance against the output in front of its match INSIDE the round:
    // lane t's matches need  need[t] <= (output in front of the round) + (output of lanes 1 .. t - 1)
    PT<uint32_t, N> x_out, xs;
    SIMT_BEGIN(t, N)
        const bool v = t >= 1 && (uint32_t)t < nv;
        c_lit[t] = v ? c_lit[t] : 0u; c_rec[t] = v ? c_rec[t] : 0u; c_out[t] = v ? c_out[t] : 0u;
        x_out[t] = c_out[t];
        pe[t] = c_lit[t]; xs[t] = c_rec[t];
    SIMT_END
    simt::wave_scan_incl<N>(x_out);
    simt::wave_scan_incl<N>(pe);
    simt::wave_scan_incl<N>(xs);
    const uint32_t tot_out = simt::wave_read<N>(x_out, N - 1), tot_lit = simt::wave_read<N>(pe, N - 1), tot_rec = simt::wave_read<N>(xs, N - 1);
    SIMT_BEGIN(t, N)
        const bool v = t >= 1 && (uint32_t)t < nv;
        const int32_t shortfall = (int32_t)c_need[t] - (int32_t)(x_out[t] - c_out[t]);
        xs[t] = v ? (uint32_t)shortfall + 0x80000000u : 0u;
    SIMT_END
    simt::wave_scan_max_incl<N>(xs);
    const uint32_t mneed = simt::wave_read<N>(xs, N - 1);
    const uint32_t last = nv >= 1u ? nv - 1u : 0u;
    const uint32_t end_last = simt::wave_read<N>(endp, (int)last), tail_last = simt::wave_read<N>(c_tail, (int)last), start1 = simt::wave_read<N>(start, 1);
    SIMT_BEGIN(t, N)
        R.c_lit[t] = c_lit[t]; R.c_rec[t] = c_rec[t];
        if (t == 0) {
            R.nv = nv; R.eob = eob ? 1u : 0u; R.ok = ok && nv >= 2u ? 1u : 0u;
            R.adopt = 0u;   // (the helper's own slot: the master writes it behind the barrier that follows, and not again before the next one)
            R.start1 = start1; R.end_last = end_last; R.tail_last = tail_last;
            R.tot_lit = tot_lit; R.tot_rec = tot_rec; R.tot_out = tot_out; R.mneed = mneed;
        }
    SIMT_END_WAVE
}
// an adopted helper moves its rows to the offsets the master named (all helpers at once)
SWC_D void team_helper_copy(const Team& tm, int h, gptr lits, SWC_AS_GLOBAL uint32_t* recs) {
    using simt::PT;
    constexpr int N = kWave;
    const TeamRound& R = tm.sh->r[h - 1];
    if (simt::uniform(R.adopt) == 0u) return;
    gptr hp = tm.scratch + (size_t)(h - 1) * kTeamProvBytes;
    const uint64_t nlit = ((uint64_t)simt::uniform((uint32_t)(R.nlit_base >> 32)) << 32) | simt::uniform((uint32_t)R.nlit_base);
    const uint32_t nrec = simt::uniform(R.nrec_base);
    PT<uint32_t, N> c_lit, c_rec, x_lit, x_rec;
    SIMT_BEGIN(t, N) c_lit[t] = R.c_lit[t]; c_rec[t] = R.c_rec[t]; x_lit[t] = c_lit[t]; x_rec[t] = c_rec[t]; SIMT_END
    simt::wave_scan_incl<N>(x_lit);
    simt::wave_scan_incl<N>(x_rec);
    SIMT_BEGIN(t, N)
        if (c_lit[t] | c_rec[t]) {
            copy_prov(hp + lzr::kProvRecBytes + 4u * (uint32_t)t, hp + 4u * (uint32_t)t, c_lit[t], c_rec[t],
                      lits + nlit + (x_lit[t] - c_lit[t]), recs + nrec + (x_rec[t] - c_rec[t]));
        }
    SIMT_END
}

// the master opens a super-round (the helpers leave their barrier) ...
SWC_D void team_begin(Team& tm, gcptr in, uint32_t in_len, uint32_t B, const SubTab st, gptr lits, SWC_AS_GLOBAL uint32_t* recs) {
    TeamShared* sh = tm.sh;
    SIMT_BEGIN(t, kWave)
        if (t == 0) { sh->cmd = 1u; sh->gen = tm.gen; sh->B = B; sh->in_len = in_len; sh->in = (uint64_t)(uintptr_t)in; sh->subg = (uint64_t)(uintptr_t)st.g; sh->lits = (uint64_t)(uintptr_t)lits; sh->recs = (uint64_t)(uintptr_t)recs; }
    SIMT_END_WAVE
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
}
// ... and closes it: behind this the helpers' results are in LDS and their rows in memory (the host emulation runs them here,
// one after the other)
SWC_D void team_end(Team& tm) {
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#else
    for (int h = 1; h <= tm.helpers; h++) team_helper_round(tm, h);
#endif
}
// ... and, when it has said which of them it adopts, lets the helpers copy (it does not wait for them: nothing of this job reads
// the records again, and the next barrier -- the next super-round or the end of the job -- is behind the copies)
SWC_D void team_go(Team& tm, gptr lits, SWC_AS_GLOBAL uint32_t* recs) {
#if defined(__HIP_DEVICE_COMPILE__)
    simt::wave_fence();
    __syncthreads();
    (void)lits; (void)recs;
#else
    for (int h = 1; h <= tm.helpers; h++) team_helper_copy(tm, h, lits, recs);
#endif
}
// the job is over: the helpers leave
SWC_D void team_dismiss(Team& tm) {
    SIMT_BEGIN(t, kWave) if (t == 0) tm.sh->cmd = 0u; SIMT_END_WAVE
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
}
// what the wavefronts 1 .. helpers of the workgroup run (device only: the host emulation runs the helpers' rounds in team_end)
SWC_D void team_helper_loop(const Team& tm, int h) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (;;) {
        __syncthreads();
        if (simt::uniform(tm.sh->cmd) == 0u) return;
        team_helper_round(tm, h);
        simt::wave_fence();
        __syncthreads();
        __syncthreads();   // (the master has said which rounds it adopts)
        team_helper_copy(tm, h, (gptr)(uintptr_t)(((uint64_t)simt::uniform((uint32_t)(tm.sh->lits >> 32)) << 32) | simt::uniform((uint32_t)tm.sh->lits)),
                         (SWC_AS_GLOBAL uint32_t*)(uintptr_t)(((uint64_t)simt::uniform((uint32_t)(tm.sh->recs >> 32)) << 32) | simt::uniform((uint32_t)tm.sh->recs)));
    }
#else
    (void)tm; (void)h;
#endif
}

// ---- the rounds of one block ---------------------------------------------------------------------------------------
// Decodes from the reader's position until the end-of-block symbol (kSyncEob) or until something the fast path leaves
// to the checked step (kSyncBail; kSyncBailCap: the capacity lies inside the next round).  Commits whole rounds only.
template <bool TEAM = false>
SWC_D int sync_block(Lane& ln, SyncLds* sl, const SubTab st, SyncProf& pf, Team* tm = nullptr) {
    using simt::PT;
    constexpr int N = kWave;
    // Literals in front of the block that no record covers yet (the tail of the previous block, stored bytes): the first
    // sub-chunk of the block takes a short run into its first record; a long one becomes a record of its own here.
    uint32_t pending = 0;
    {
        const uint64_t kept = ln.pos < ln.cap ? ln.pos : ln.cap;
        const uint64_t open = kept > ln.last_end ? kept - ln.last_end : 0;
        if (open <= 64) pending = (uint32_t)open;
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
    for (;;) {
        const uint32_t B = (uint32_t)(P >> 3) & ~3u;
        const uint32_t q0 = (uint32_t)(P - 8ull * B);
        SWC_SP(pf, 6)
        SWC_SPC(pf, 7, 1);
        bool helped = false;
        if constexpr (TEAM) if (tm != nullptr && tm->sh != nullptr && pos < ln.cap) {   // the helpers take the rounds behind this one, at once
            team_begin(*tm, in, in_len, B, st, ln.lits, ln.recs);
            helped = true;
        }
        const uint64_t left = (uint64_t)(in_len - B) * 8;
        const uint32_t in_bits = left > 0x3FFFFFFFull ? 0x3FFFFFFFu : (uint32_t)left;
        // stage [B, B + kSyncStage), zero-filled beyond the input, shifted left by two bits: LDS bit p + 2 is stream bit p.
        // All loads of the round are issued before the first is used: ONE memory latency per round, not one per kilobyte.
        {
            constexpr uint32_t kParts = (kSyncStage + 16u * N - 1u) / (16u * N);
            PT<uint64_t, N> sa[kParts], sb[kParts];
            PT<uint32_t, N> sp[kParts];
            SIMT_BEGIN(t, N)
#pragma unroll
                for (uint32_t k = 0; k < kParts; k++) {
                    const uint32_t o = 16u * (uint32_t)t + k * 16u * N;
                    uint64_t a = 0, b = 0;
                    uint32_t prev = 0;
                    const uint64_t at = (uint64_t)B + o;
                    if (o < kSyncStage) {
                        if (at + 16 <= in_len) {
                            a = load_u64(in + at); b = load_u64(in + at + 8);
                            if (o != 0) prev = load_u32(in + at - 4);
                        } else {
                            for (uint32_t q = 0; q < 8; q++) if (at + q < in_len) a |= (uint64_t)in[at + q] << (8 * q);
                            for (uint32_t q = 0; q < 8; q++) if (at + 8 + q < in_len) b |= (uint64_t)in[at + 8 + q] << (8 * q);
                            if (o != 0) for (uint32_t q = 0; q < 4; q++) if (at - 4 + q < in_len) prev |= (uint32_t)in[at - 4 + q] << (8 * q);
                        }
                    }
                    sa[k][t] = a; sb[k][t] = b; sp[k][t] = prev;
                }
            SIMT_END
            SIMT_BEGIN(t, N)
#pragma unroll
                for (uint32_t k = 0; k < kParts; k++) {
                    const uint32_t o = 16u * (uint32_t)t + k * 16u * N;
                    if (o < kSyncStage) {
                        uint32_t* st32 = (uint32_t*)(sl->stage + o);
                        const uint64_t a = sa[k][t], b = sb[k][t];
                        const uint32_t w0 = (uint32_t)a, w1 = (uint32_t)(a >> 32), w2 = (uint32_t)b, w3 = (uint32_t)(b >> 32);
                        st32[0] = funnel32(w0, sp[k][t], 30); st32[1] = funnel32(w1, w0, 30); st32[2] = funnel32(w2, w1, 30); st32[3] = funnel32(w3, w2, 30);
                    }
                }
                // (lanes 1..63 begin their walk kSyncWalkBack bits IN FRONT of their sub-chunk: the further a walk has come when it
                // crosses into the sub-chunk, the surer its end is the true one, and a wrong end costs the round another pass)
                start[t] = t == 0 ? q0 : (uint32_t)t * kSyncChunk * 8u - kSyncWalkBack;
            SIMT_END_WAVE
        }
        uint32_t nv = 0;
        bool eob = false, bail = false;
        const bool chk = (uint64_t)B + kSyncStage > in_len;   // only the last rounds of a stream can run out of input
        SWC_SP(pf, 2)
        // pass 1: where does a decode from my guess end?  (no counting)
        SWC_SYNC_STAT(2, 1);
        SWC_SPC(pf, 8, 1);
        SIMT_BEGIN(t, N)
            SWC_SYNC_LANE_BEGIN(t_i_ == 0)
            const uint32_t ce = ((uint32_t)t + 1u) * kSyncChunk * 8u;
            endp[t] = walk_chunk(sl, st, start[t], ce, in_bits);
            flg[t] = 0; have[t] = false;
            SWC_SYNC_LANE_END(t_i_ == N - 1, 0)
        SIMT_END
        SWC_SP(pf, 9)
        // ---- the round in ONE more decode (the common case): every lane decodes its sub-chunk from the end of its left
        // neighbour into its scratch (decode_chunk_prov), the chain is checked as below, the pieces are copied to their
        // offsets.  Anything unusual -- a symbol for the checked step, a literal run too long for one record, the capacity
        // or the workspace inside the round, a distance beyond the output -- abandons the attempt BEFORE anything is
        // committed; the general passes below then take the round.
        if (pos < ln.cap) {
            bool ok = true;
            const bool need_check = pos < 32768u;   // (further in, every distance the tables can produce has its source)
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
                    SWC_SYNC_LANE_BEGIN(t_i_ == 0)
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
                        } else if (need_check) decode_chunk_prov<true>(sl, st, start[t], ce, in_bits, ln.prov, (uint32_t)t, t == 0 ? pending : 0u, r);
                        else decode_chunk_prov<false>(sl, st, start[t], ce, in_bits, ln.prov, (uint32_t)t, t == 0 ? pending : 0u, r);
                        endp[t] = r.end; c_lit[t] = r.nlit; c_rec[t] = r.nrec; c_out[t] = r.nout; flg[t] = r.flags;
                        c_tail[t] = r.tail; c_need[t] = (uint32_t)r.need;
                        have[t] = true;
                    }
                    SWC_SYNC_LANE_END(t_i_ == N - 1, 1)
                SIMT_END
                SWC_SP(pf, 3)
            }
            if constexpr (TEAM) if (helped) team_end(*tm);   // (the helpers' rounds are in their rows and in LDS now, whatever becomes of this one)
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
                if (need_check) {
                    // every distance must reach back no further than the output in front of its match (need = distance - 1 - ...)
                    SIMT_BEGIN(t, N)
                        const uint64_t p0 = pos + (x_out[t] - c_out[t]);
                        const int32_t room = p0 > 0x40000000ull ? 0x40000000 : (int32_t)p0;
                        pb[t] = (uint32_t)t < nv && (int32_t)c_need[t] >= room;
                    SIMT_END
                    if (simt::wave_ballot<N>(pb)) ok = false;
                }
                if (pos + tot_out > ln.cap || (uint64_t)nrec + tot_rec > ln.max_rec) ok = false;
                if (ok) {
                    SIMT_BEGIN(t, N)
                        if ((uint32_t)t < nv) {
                            copy_prov(ln.prov + lzr::kProvRecBytes + 4u * (uint32_t)t, ln.prov + 4u * (uint32_t)t, c_lit[t], c_rec[t],
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
                    if constexpr (TEAM) if (helped) {
                        // The helpers' rounds, in order, as long as each began where the one in front of it ended.  Everything the
                        // master's own commit checks is checked here as well, on the numbers the helper reduced its round to; whatever
                        // fails ends the adoption, and the master decodes on from P itself (and meets the same thing in its own time).
                        for (int h = 1; h <= tm->helpers && !eob; h++) {
                            TeamRound& R = tm->sh->r[h - 1];
                            const uint64_t Bh8 = 8ull * ((uint64_t)B + (uint64_t)h * kTeamStride);
                            const uint32_t hnv = simt::uniform(R.nv);
                            if (simt::uniform(R.ok) == 0u || hnv < 2u || hnv > (uint32_t)N || P < Bh8 || P - Bh8 != (uint64_t)simt::uniform(R.start1)) break;
                            const bool heob = simt::uniform(R.eob) != 0u;
                            if (!heob && hnv != (uint32_t)N) break;
                            const uint32_t hl = simt::uniform(R.tot_lit), hr = simt::uniform(R.tot_rec), ho = simt::uniform(R.tot_out);
                            if (pos < 32768u && (int64_t)(int32_t)(simt::uniform(R.mneed) - 0x80000000u) >= (int64_t)pos) break;   // a distance beyond the output
                            if (pos + ho > ln.cap || (uint64_t)nrec + hr > ln.max_rec) break;
                            SIMT_BEGIN(t, N) if (t == 0) { R.adopt = 1u; R.nlit_base = nlit; R.nrec_base = nrec; } SIMT_END
                            SWC_TEAM_STAT();       // rounds adopted
                            pos += ho;
                            nlit += hl;
                            nrec += hr;
                            P = Bh8 + simt::uniform(R.end_last);
                            ln.last_end = pos - (heob ? simt::uniform(R.tail_last) : 0u);
                            if (heob) eob = true;
                        }
                        team_go(*tm, ln.lits, ln.recs);
                    }
                    if (eob) { result = kSyncEob; break; }
                    continue;
                }
            }
            // abandoned: the general passes start over
            if constexpr (TEAM) if (helped) team_go(*tm, ln.lits, ln.recs);   // (nobody is adopted)
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
                SWC_SYNC_LANE_BEGIN(t_i_ == 0)
                if (todo[t]) {
                    SWC_SYNC_STAT(3, 1);   // lane decodes
                    if (t != 0) start[t] = pe[t];
                    ChunkOut r;
                    const uint32_t ce = ((uint32_t)t + 1u) * kSyncChunk * 8u;
                    if (start[t] + kSyncChunk * 8u < ce) { r.end = kPosFail; r.nlit = r.nrec = r.nout = 0; r.flags = kFlagFail; }   // (see the pass above)
                    else if (chk) decode_chunk<0, false, true>(sl, st, start[t], ce, in_bits, nullptr, nullptr, 0, r);
                    else decode_chunk<0, false, false>(sl, st, start[t], ce, in_bits, nullptr, nullptr, 0, r);
                    if (r.nlit > lzr::kLitRunMax) {   // a literal run may need a record of its own: count those too
                        if (chk) decode_chunk<0, true, true>(sl, st, start[t], ce, in_bits, nullptr, nullptr, 0, r);
                        else decode_chunk<0, true, false>(sl, st, start[t], ce, in_bits, nullptr, nullptr, 0, r);
                    }
                    endp[t] = r.end; c_lit[t] = r.nlit; c_rec[t] = r.nrec; c_out[t] = r.nout; flg[t] = r.flags;
                    have[t] = true;
                }
                SWC_SYNC_LANE_END(t_i_ == N - 1, 1)
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
            SWC_SYNC_LANE_BEGIN(t_i_ == 0)
            if ((uint32_t)t < nv) {
                ChunkOut r;
                const uint32_t ce = ((uint32_t)t + 1u) * kSyncChunk * 8u;
                const uint64_t p0 = pos + (x_out[t] - c_out[t]);
                const bool bigs = c_lit[t] > lzr::kLitRunMax;
                if (beyond) decode_chunk<2, false, true>(sl, st, start[t], ce, in_bits, nullptr, nullptr, p0, r);
                else if (bigs || chk) decode_chunk<1, true, true>(sl, st, start[t], ce, in_bits, ln.lits + nlit + (x_lit[t] - c_lit[t]), ln.recs + nrec + (x_rec[t] - c_rec[t]), p0, r);
                else decode_chunk<1, false, false>(sl, st, start[t], ce, in_bits, ln.lits + nlit + (x_lit[t] - c_lit[t]), ln.recs + nrec + (x_rec[t] - c_rec[t]), p0, r);
                flg[t] = r.flags;
            }
            SWC_SYNC_LANE_END(t_i_ == N - 1, 2)
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

// ---- the code-length section of a dynamic header (Deflate.swift:86-167) ---------------------------------------------------
// Same results and errors as the reference's loop.  The section is staged in LDS once (round 3 read it dword by dword from
// the stream, one exposed memory latency per eight code lengths); then, 64 bit offsets at a time, every lane decodes the
// code-length symbol that WOULD start at its offset (table read, extra bits, repeat count), and a scalar chain walks from
// symbol to symbol over the lanes (one cross-lane read per symbol), checking what the reference checks in its order and
// storing the lengths.  Histogram, canonical codes and tables then run on all lanes (sync_tables_from_lengths).
SWC_D int build_dynamic_par(Lane& ln, SyncLds* sl, const Spill sp, SubTab& st, bool& fast, SyncProf& pf) {
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
    SWC_SP(pf, 14)
    uint8_t* lens = sl->stage + kHdrLens;
    uint8_t* cl_lut = sl->stage + kHdrClLut;
    uint32_t* hin = (uint32_t*)(sl->stage + kHdrIn);
    const int total = literals + distances;
    const uint64_t end_bits = (uint64_t)br.len * 8;
    uint64_t W = simt::uniform(br.consumed_bits());      // absolute bit position of the next code-length symbol
    gcptr in = br.in;
    const uint32_t in_len = br.len;
    SIMT_BEGIN(t, N)
        for (int i = t; i < 80; i += N) ((uint32_t*)lens)[i] = 0;
        for (int x = t; x < 128; x += N) {   // the code the stream bits x (first bit = bit 0) begin with
            uint32_t len;
            const int idx = Lane::cl_lookup(cl, brev32((uint32_t)x) >> 17, len);
            const uint32_t sym = idx < 0 ? 0u : (uint32_t)((idx < 12 ? cl.sym_lo >> (5 * idx) : cl.sym_hi >> (5 * (idx - 12))) & 31u);
            cl_lut[x] = idx < 0 ? (uint8_t)0xFF : (uint8_t)(len | (sym << 3));
        }
    SIMT_END_WAVE
    SWC_SP(pf, 15)
    int n = 0;
    uint32_t prev = 0;
    uint64_t HB = ~0ull;                                  // byte offset of the staged part of the header (none yet)
    PT<uint32_t, N> nxt, mlo, mhi, jmp, gl, gh, gj, src, symv, repv, errv, cbx, bef, key, kex;
    PT<bool, N> pb, actv;
    while (n < total) {   // Deflate.swift:117-162, 64 bit offsets per turn
        if (HB == ~0ull || W + 64 + 32 > 8 * (HB + 1024)) {
            HB = (W >> 3) & ~3ull;
            SIMT_BEGIN(t, N)
                uint64_t a = 0, b = 0;
                const uint64_t at = HB + 16u * (uint32_t)t;
                if (at + 16 <= in_len) { a = load_u64(in + at); b = load_u64(in + at + 8); }
                else {
                    for (uint32_t k = 0; k < 8; k++) if (at + k < in_len) a |= (uint64_t)in[at + k] << (8 * k);
                    for (uint32_t k = 0; k < 8; k++) if (at + 8 + k < in_len) b |= (uint64_t)in[at + 8 + k] << (8 * k);
                }
                uint32_t* h4 = hin + 4 * t;
                h4[0] = (uint32_t)a; h4[1] = (uint32_t)(a >> 32); h4[2] = (uint32_t)b; h4[3] = (uint32_t)(b >> 32);
                if (t < 4) hin[256 + t] = 0;
            SIMT_END_WAVE
        }
        // every lane: the symbol that WOULD start at bit W + t, and the lane the symbol behind it starts at
        const uint32_t rel0 = (uint32_t)(W - 8 * HB);
        const uint64_t left64 = end_bits - W;
        const uint32_t left = left64 > 0xFFFFu ? 0xFFFFu : (uint32_t)left64;    // stream bits from W on (as far as it matters)
        SIMT_BEGIN(t, N)
            const uint32_t rel = rel0 + (uint32_t)t;
            const uint32_t* w = hin + (rel >> 5);
            const uint32_t bits = funnel32(w[1], w[0], rel);
            const uint32_t e = cl_lut[bits & 127u];
            const uint32_t len = e & 7u, sym = e == 0xFFu ? 0u : e >> 3;
            const uint32_t xb = sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u;
            const uint32_t xv = bfe32(bits, len, xb);
            const uint32_t avail = left > (uint32_t)t ? left - (uint32_t)t : 0u;
            symv[t] = sym;
            repv[t] = sym < 16u ? 1u : sym == 18u ? 11u + xv : 3u + xv;
            // what the reference would throw HERE, short of what depends on the running count: 1 = no code / the code runs past
            // the end of the input (:122), 2 = the extra bits do (:132, :145, :152)
            errv[t] = e == 0xFFu || len > avail ? 1u : len + xb > avail ? 2u : 0u;
            nxt[t] = (uint32_t)t + (e == 0xFFu ? 1u : len + xb);
            jmp[t] = nxt[t];
            mlo[t] = t < 32 ? 1u << t : 0u;
            mhi[t] = t < 32 ? 0u : 1u << (t - 32);
        SIMT_END
        // which lanes does the chain from lane 0 visit?  Pointer doubling: after step k a lane knows the lanes within 2^k hops.
#pragma unroll 1
        for (int k = 0; k < 6; k++) {
            if (simt::uniform(simt::wave_read<N>(jmp, 0)) >= 64u) break;   // lane 0 has seen its whole chain (4 bits per symbol: 16 hops)
            SIMT_BEGIN(t, N) src[t] = jmp[t] < 64u ? jmp[t] : (uint32_t)t; SIMT_END
            simt::wave_gather<N>(gl, mlo, src);
            simt::wave_gather<N>(gh, mhi, src);
            simt::wave_gather<N>(gj, jmp, src);
            SIMT_BEGIN(t, N) if (jmp[t] < 64u) { mlo[t] |= gl[t]; mhi[t] |= gh[t]; jmp[t] = gj[t]; } SIMT_END
        }
        const uint64_t V = ((uint64_t)simt::uniform(simt::wave_read<N>(mhi, 0)) << 32) | simt::uniform(simt::wave_read<N>(mlo, 0));
        // the running count in front of every visited symbol
        SIMT_BEGIN(t, N) cbx[t] = (V >> t) & 1ull ? repv[t] : 0u; SIMT_END
        simt::wave_scan_incl<N>(cbx);
        // the symbols the reference's loop gets to (count < total), their errors in its order
        SIMT_BEGIN(t, N)
            const bool vis = ((V >> t) & 1ull) != 0ull;
            const uint32_t rep = repv[t], before = (uint32_t)n + cbx[t] - (vis ? rep : 0u), sym = symv[t];
            const bool act = vis && before < (uint32_t)total;
            uint32_t err = 0;
            if (act) {
                if (errv[t] == 1u) err = (uint32_t)SWC_E_DEFLATE_SYMBOL_NOT_FOUND;                       // :122
                else if (sym == 16u && before == 0u) err = (uint32_t)SWC_E_DEFLATE_WRONG_SYMBOL;          // :155 (symbol 16 first)
                else if (sym >= 16u && errv[t] == 2u) err = (uint32_t)SWC_E_DEFLATE_SYMBOL_NOT_FOUND;     // :132, :145, :152
                else if (sym == 16u && before + rep > (uint32_t)total) err = (uint32_t)SWC_E_DEFLATE_WRONG_SYMBOL;   // :135
            }
            errv[t] = err;
            pb[t] = err != 0u;
            actv[t] = act;
            bef[t] = before;
            // the value a symbol 16 repeats: that of the nearest symbol in front of it that is not a 16 (17 / 18 leave a zero)
            key[t] = act && sym != 16u ? (((uint32_t)t + 1u) << 8) | (sym < 16u ? sym : 0u) : 0u;
        SIMT_END
        {
            const uint64_t m_err = simt::wave_ballot<N>(pb);
            if (m_err) return (int)simt::wave_read<N>(errv, simt::ctz64(m_err));   // the first one in the reference's order
        }
        kex = key;
        simt::wave_scan_max_incl<N>(kex);
        const uint32_t klast = simt::uniform(simt::wave_read<N>(kex, N - 1));
        simt::wave_shift_up<N>(key, kex, 0u);
        SIMT_BEGIN(t, N)
            if (actv[t]) {
                const uint32_t sym = symv[t];
                const uint32_t val = sym < 16u ? sym : sym == 16u ? (key[t] ? key[t] & 255u : prev) : 0u;
                // (a non-zero value repeats at most six times, and -- checked above -- stays inside the table)
                if (val != 0u) for (uint32_t k = 0; k < repv[t]; k++) lens[bef[t] + k] = (uint8_t)val;
            }
        SIMT_END
        const uint64_t m_act = simt::wave_ballot<N>(actv);
        const int last = 63 - simt::clz64(m_act);           // (lane 0 is always one of them)
        n = (int)(simt::uniform(simt::wave_read<N>(bef, last)) + simt::uniform(simt::wave_read<N>(repv, last)));
        W += simt::uniform(simt::wave_read<N>(nxt, last));
        if (klast) prev = klast & 255u;
    }
    if (n != total) return SWC_E_DEFLATE_WRONG_SYMBOL;  // :161
    simt::wave_fence();
    br.seek(W);
    SWC_SP(pf, 0)
    fast = sync_tables_from_lengths(sl, sp, literals, distances, st, pf);
    return SWC_OK;
}

// Deflate.swift:77-81 with the fixed code of Deflate+Constants.swift:11-173: the same table build from the fixed lengths
SWC_D void build_static_par(SyncLds* sl, const Spill sp, SubTab& st, bool& fast, SyncProf& pf) {
    constexpr int N = kWave;
    uint8_t* lens = sl->stage + kHdrLens;
    SIMT_BEGIN(t, N)
        for (int s = t; s < 320; s += N) lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
    SIMT_END_WAVE
    fast = sync_tables_from_lengths(sl, sp, 288, 32, st, pf);
}

// ---- the job -------------------------------------------------------------------------------------------------------
// Deflate.swift:30-249 for one stream on one wavefront.  `ws` / `ws_bytes`: the stream's area in the HBM workspace
// (lzr::StreamHeader | records | scratch rows + table spill | literal stream).  On the device every lane of the wave calls
// this with its lane number; the host emulation calls it once (lane 0 of 1) and runs the 64 lanes of the parallel parts one
// after another.
template <bool TEAM = false>
SWC_D void inflate_sync_job(Job& job, SyncLds* sl, uint8_t* ws, size_t ws_bytes, int lane, int lanes, uint64_t* prof = nullptr, Team* tm = nullptr) {
    Lane ln;
    SyncProf pf;
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    pf.tlast = __builtin_readcyclecounter();
#endif
    ln.wlane = lane;
    ln.wlanes = lanes;
    ln.l = LaneLds{nullptr, 1};
    ln.out = (gptr)job.out;
    ln.cap = job.out_cap;
    ln.pos = 0;
    ln.nrec = 0;
    ln.nlit = 0;
    ln.last_end = 0;
    const size_t lo = ws ? lzr::lit_offset(ws_bytes, job.out_cap) : 0;
    ln.recs = (SWC_AS_GLOBAL uint32_t*)(ws + sizeof(lzr::StreamHeader));
    // the lanes' scratch and the table spill (lzr::kProvBytes) sit between the record list and the literal stream
    size_t rec_end = lo;
    ln.prov = nullptr;
    if (lo >= sizeof(lzr::StreamHeader) + 256 + lzr::kProvBytes) {
        rec_end = (lo - lzr::kProvBytes) & ~(size_t)15;
        ln.prov = (gptr)(ws + rec_end);
    }
    ln.max_rec = rec_end > sizeof(lzr::StreamHeader) ? (uint32_t)((rec_end - sizeof(lzr::StreamHeader)) / 4) : 0u;
    ln.lits = (gptr)(ws + lo);
    Spill sp{nullptr, nullptr, nullptr};
    if (ln.prov) {
        gptr s0 = ln.prov + lzr::kProvRecBytes + lzr::kProvLitBytes;
        sp.aux = (SWC_AS_GLOBAL uint32_t*)s0;
        sp.syms = (SWC_AS_GLOBAL uint16_t*)(s0 + kSpillSyms);
        sp.sub = (SWC_AS_GLOBAL uint32_t*)(s0 + kSpillSub);
    }
    int st = SWC_OK;
    if (lo == 0 || ln.prov == nullptr) {
        st = SWC_E_NEED_WORKSPACE;   // (an area smaller than swc_batch_workspace_bytes asks for)
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
                bool fast = false;
                SubTab stb{nullptr};
                if (type == 1) build_static_par(sl, sp, stb, fast, pf);
                else st = build_dynamic_par(ln, sl, sp, stb, fast, pf);
                SWC_SP(pf, 1)
                if constexpr (TEAM) if (tm != nullptr) tm->gen++;   // (new tables: the helpers copy them before their next round)
                if (st == SWC_OK) {
                    for (;;) {   // Deflate.swift:171-236
                        if (fast) {
                            const int r = sync_block<TEAM>(ln, sl, stb, pf, tm);
                            if (r == kSyncEob) break;
                            if (r == kSyncBail) fast = false;   // the checked step takes the rest of the block
                            // kSyncBailCap: checked steps until the capacity is behind us
                        }
                        const uint64_t until = fast && ln.pos < ln.cap ? ln.cap : ~0ull;
                        int s2 = SWC_OK;
                        do { s2 = careful_step(ln, sp); } while (s2 == SWC_OK && ln.pos < until && until != ~0ull);
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
    if constexpr (TEAM) if (tm != nullptr && tm->sh != nullptr) team_dismiss(*tm);
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
    if (prof && lane == 0) for (int k = 0; k < 16; k++) prof[k] = pf.acc[k];
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
