// lzma_wave.h -- LZMA / LZMA2 decode, one stream per WAVEFRONT.
//
// Replaces LZMADecoder.decode() (reference Sources/LZMA/LZMADecoder.swift:107-284) with its range
// decoder (LZMARangeDecoder.swift:20-80), bit-tree / length decoders (LZMABitTreeDecoder.swift:18-43,
// LZMALenDecoder.swift:30-38) and the LZMA2 chunk framing (Sources/LZMA2/LZMA2Decoder.swift:34-99).
//
// The range coder is a strictly serial chain, so a stream cannot be split; the adaptive probability model (up to 14,136
// 11-bit cells for lc + lp <= 4) is far too large to keep per LANE, so one wavefront owns one stream and all 64 lanes execute
// the decode chain redundantly; the LZ copy of every match is spread over the lanes.  Every branch on a value that is the same
// in all lanes is SAID to be so (same() = simt::wave_true: v_cmp + s_cmp_lg_u64 vcc + s_cbranch) -- the compiler cannot know
// it of a value read from LDS, and made each of them an exec-mask region with both sides issued; the decoded bit then is a
// constant per side, and tree indices, `state` and the tests on them live on the scalar unit (56 VGPRs).  The kernel is bound
// by the latency of its one chain, so its speed is the number of streams a CU holds -- i.e. the LDS a stream needs.  Two
// layouts (kernels.hip picks):
//   coder cache (the default when the launch has a workspace): the 1,336 non-literal cells without the two long-length trees
//       and FOUR LINES of literal coders (a line = 0x100 cells = a third of a coder: its plain tree or one of the two trees of
//       the matched mode) in LDS, all literal coders and the long-length trees in the workspace; a literal whose line is not
//       cached writes the oldest line back and loads its own (lines never used since the last reset are filled, not
//       loaded): 4,720 B => 8 streams per SIMD, the hardware's limit;
//   whole model in LDS (no workspace; round 2's layout): every literal coder of lc + lp <= 4 in LDS, 28,272 B => 5 streams
//       per CU; lc + lp > 4 (legal for .lzma, never produced by xz) reports SWC_E_NEED_WORKSPACE and is re-run with one.
// The next 256 input bytes sit in a register spread over the lanes and are refilled at one place per symbol.
//
// The same source is compiled for the host with WAVE = 1 (tests/host_emu) to check the serial logic
// against the oracle without a GPU; the wave-parallel copy is the only part that differs.
#ifndef SWC_LZMA_WAVE_H
#define SWC_LZMA_WAVE_H

#include "swc_common.h"
#include "simt.h"

namespace swc {
namespace lzma {

// probability model layout (u16 cells).  The reference keeps `probabilities[432]` with its own index
// arithmetic (LZMADecoder.swift:50-62); only cell independence matters, except that index 432 does
// not exist there (state 11, posState 15 => Swift trap) -- reproduced below.
constexpr int P_IS_MATCH = 0;        // [12 << 4]
constexpr int P_IS_REP = 192;        // [12]
constexpr int P_IS_REP_G0 = 204;     // [12]
constexpr int P_IS_REP_G1 = 216;     // [12]
constexpr int P_IS_REP_G2 = 228;     // [12]
constexpr int P_IS_REP0_LONG = 240;  // [12 << 4]
constexpr int P_POS_SLOT = 432;      // [4][64]
constexpr int P_ALIGN = 688;         // [16]
constexpr int P_POS_DEC = 704;       // [115]  (reference: 1 + 128 - 14 cells, LZMADecoder.swift:96-97)
// (the bit trees start at EVEN cells: tree() reads the two children of a node as one aligned dword)
constexpr int P_LEN = 820;           // choice, choice2, low[16][8], mid[16][8] = 258
constexpr int P_REP_LEN = 1078;
constexpr int P_LEN_HIGH = 1336;     // high[256] of the length coder, high[256] of the rep-length coder: lengths >= 18, rare --
                                     // the LAST non-literal cells, so that the coder cache can leave them out of LDS (kSlotBase)
constexpr int P_LITERAL = 1848;      // [0x300 << (lc+lp)]
constexpr int kMaxLdsLitBits = 4;    // largest lc + lp any build keeps in LDS (host emulation, tests)
constexpr int kProbCells = P_LITERAL + (0x300 << kMaxLdsLitBits);  // 14,135
// LDS of a wave when literal coders up to lc + lp = `bits` stay in LDS (more bits spill to the HBM workspace):
// 3 (what xz writes: lc 3, lp 0) -> 15,984 B = 10 streams per CU; 4 -> 28,272 B = 5 streams per CU
constexpr int lds_bytes_for(int bits) { return (((P_LITERAL + (0x300 << bits)) * 2 + 15) / 16) * 16; }
constexpr int kLdsBytesPerWave = lds_bytes_for(kMaxLdsLitBits);
// LDS as a CACHE of the literal coders (round 3).  The kernel is latency-bound on one serial chain per stream, so its speed
// is the number of streams a CU holds, and that is set by the model in LDS: 15,984 B with the eight literal coders of
// lc + lp = 3 -> 10 streams.  A literal coder (0x300 cells) is picked by the top bits of the previous byte; data use few of
// them at a time (text: four), so LDS keeps kCoderSlots of them and all coders live in the HBM workspace (round 3: four
// slots, 9,840 B -> 16 streams per CU).  A miss writes the victim back and loads the coder (1.5 KB each way, all lanes; a coder nobody has used
// since the last model reset is filled with the initial value instead); slots are replaced round-robin.  Any lc + lp works
// this way (the old path kept lc + lp <= 3 in LDS and decoded larger models cell by cell from HBM).
// Round 4: with every branch on a wave-identical value a scalar branch (simt::wave_true) the kernel needs 56 VGPRs and is
// bound by the latency of its chain again, i.e. by the streams per CU: ONE slot (5,232 B -> 31 streams per CU) beats two (24)
// and four (16) on every payload -- text 751 / 776 / 945 ms, binary records (all eight coders live) 2,462 / 2,612 / 2,818,
// P-mix 1,732 / 1,799 / 1,941 (profiles/r04_experiments.txt).
#ifndef SWC_LZMA_SLOTS
#define SWC_LZMA_SLOTS 1
#endif
constexpr int kCoderSlots = SWC_LZMA_SLOTS;
// Round 4, second step: the cache holds LINES of 0x100 cells -- a third of a literal coder: its plain tree, or one of the two
// trees of the matched mode -- not whole coders.  A plain literal (four of five on text) touches the plain tree only, so a
// miss moves 512 B each way instead of 1.5 KB, and the same LDS keeps the plain trees of several coders (text alternates
// between two: after a letter / after a space).  Line L = 3 x coder + third, at home in the workspace at cell L * 0x100.
#ifndef SWC_LZMA_LINES
#define SWC_LZMA_LINES 4
#endif
constexpr int kLines = SWC_LZMA_LINES;
static_assert(kLines == 4, "a matched literal can need three lines at once; four tags");
// (Round 4: the slots start where the two `high` length trees would be -- in cache mode those 512 cells live in the workspace
// behind the literal coders and are decoded through bit_spill(): 1 KB less LDS per stream, 4,208 B with one slot.)
constexpr int kSlotBase = P_LEN_HIGH;             // first cell of slot 0 (a multiple of four: dword-aligned copies)
static_assert(kSlotBase % 4 == 0, "slot alignment");
constexpr size_t kSpillHighCell = (size_t)0x300u << 12;   // workspace: all literal coders of lc + lp <= 12, then the 512 `high` cells
constexpr int lds_bytes_cached() { return (((kSlotBase + 0x100 * kLines) * 2 + 15) / 16) * 16; }
constexpr uint32_t kNoCoder = 0xFFFFFFFFu;
constexpr int LEN_CHOICE = 0, LEN_CHOICE2 = 1, LEN_LOW = 2, LEN_MID = 2 + 128;

// cycle accounting of profile builds (-DSWC_PROFILE, tools/exp_profile_lzma.py): a scope adds its cycles to one slot
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
struct ProfScope {
    uint64_t* slot; uint64_t t0;
    __device__ ProfScope(uint64_t* s) : slot(s), t0(__builtin_readcyclecounter()) {}
    __device__ ~ProfScope() { *slot += __builtin_readcyclecounter() - t0; }
};
#define SWC_LZMA_PROF(k) ProfScope prof_scope_##k(&pacc[k]);
#define SWC_LZMA_COUNT(k, n) (pacc[k] += (n))
#else
#define SWC_LZMA_PROF(k)
#define SWC_LZMA_COUNT(k, n) ((void)0)
#endif

template <int WAVE>
struct Decoder {
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    uint64_t pacc[16] = {};   // 0 decode() 1 copy_match 2 byte_at 3 literal symbols (incl. 2) 4 length + distance 5 matches 6 literals 7 short reps
#endif
    // ---- I/O -----------------------------------------------------------------------------------
    gcptr in;
    uint64_t n;        // input bytes
    // the next input byte is in[win_base + wk]: `wk` (32 bits, relative to the window of the input that sits in a register, see
    // below) is what the decision chain advances and tests; the 64-bit position exists only where a chunk or the job needs it
    uint32_t wk = 0;   // offset of the next input byte from win_base
    uint32_t wlim = 0; // n - win_base (saturated): wk > wlim <=> a byte past the end of the input has been read
    SWC_HD uint64_t ip() const { return win_base + wk; }
    gptr out;
    uint64_t cap;
    uint64_t pos;      // bytes produced == dictEnd == out.count of the reference
    int lane;
    // ---- model -----------------------------------------------------------------------------------
    uint16_t* probs;            // LDS (host: heap) -- P_LITERAL + (0x300 << lds_bits) cells
    int lds_bits;               // literal coders with lc + lp <= lds_bits live in LDS, larger ones in lit_spill
    SWC_AS_GLOBAL uint16_t* lit_spill;  // HBM literal coder when lc+lp > 4, else unused; cache mode: the home of all coders
    bool cached = false;        // LDS holds kCoderSlots literal coders (see kCoderSlots); lit_spill is required
    uint32_t tag0 = kNoCoder, tag1 = kNoCoder, tag2 = kNoCoder, tag3 = kNoCoder;   // the coder in each slot
    uint32_t victim = 0;        // the slot the next miss replaces
    uint32_t mru = 0;           // the slot that was used last
    uint64_t fresh = 0;         // bit c: coder c (c < 64) has not been used since the last reset -- its cells are all 1024, nothing to load
    uint64_t fresh1 = 0, fresh2 = 0;   // the same for the second and third line of coder c (`fresh`: its first)
    bool have_model;
    int lc, lp, pb;
    uint64_t dict_size;
    int64_t uncompressed_size;  // < 0: unknown
    uint64_t dict_start;
    uint32_t range, code;
    uint64_t rep0, rep1, rep2, rep3;
    int state;
    bool trap;                  // the reference would trap (reader past the end, index out of range)
    bool overflow;              // out_cap exhausted: LZMA needs the bytes it wrote, so decoding stops
    bool need_ws;               // lc+lp > 4 but no HBM workspace was supplied

    // The next 256 input bytes sit in a register spread over the lanes (lane i: the dword at win_base + 4 i): the range coder
    // takes a byte about once per four output bytes, and a global load in its serial chain costs the wave a memory round trip
    // each time.  The window is refilled (one coalesced load) at ONE place per symbol -- the top of the symbol loop, and of the
    // LZMA2 chunk loop -- whenever fewer than kSymbolBytes bytes of it are left: a symbol normalises at most 48 times (isMatch 1,
    // isRep 1, length 10, slot 6, 26 direct bits, 4 align bits), a chunk header has 6 bytes and a properties byte.  next_byte() is
    // then a cross-lane read and nothing else: it is inlined at every one of the ~50 decision sites of decode(), and with a
    // refill (a dword load, a byte loop for the tail of the input, the bounds checks) at each of them the kernel was 48 KB of
    // code -- more than the waves of two CUs, all at different places of it, can keep in the 64 KB instruction cache they share.
    // Bytes past the end of the input read as zero and `ip` keeps counting: trapped() (checked where the old `trap` flag was:
    // before anything of the symbol is written) is the reference's trap of LittleEndianByteReader.byte() (App. A L4).
    static constexpr uint32_t kSymbolBytes = 48;
    uint64_t win_base;          // input offset of lane 0's dword (a multiple of 4)
    uint32_t win;               // this lane's dword
    uint32_t prev_byte;         // out[pos - 1] (0 when the dictionary is empty): the literal coder's context, kept in a register
    SWC_HD void win_load(uint64_t at) {
        if (at > n) trap = true;   // (already past the end: stays so)
        win_base = at & ~(uint64_t)3;
        wk = (uint32_t)at & 3u;
        const uint64_t left = n > win_base ? n - win_base : 0;
        wlim = left > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)left;
        if (WAVE != 1) {
            const uint64_t o = win_base + 4ull * (uint32_t)lane;
            uint32_t w = 0;
            if (o + 4 <= n) w = load_u32(in + o);
            else for (uint32_t k = 0; k < 4; k++) if (o + k < n) w |= (uint32_t)in[o + k] << (8 * k);
            win = w;
        }
    }
    SWC_HD void ensure_window() {
        if (wk > 4u * 64u - kSymbolBytes) win_load(ip());   // (the host build keeps the same rhythm: one code path)
    }
    SWC_HD bool trapped() const { return trap || wk > wlim; }
    SWC_HD uint8_t next_byte() {
        const uint32_t k = wk++;   // < 256: ensure_window() ran within the last kSymbolBytes bytes
        if (WAVE == 1) return k < wlim ? in[win_base + k] : (uint8_t)0;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)win, __builtin_amdgcn_readfirstlane((int)(k >> 2)));   // (the index is the same in every lane: said so, or the compiler loops over the lanes)
#else
        const uint32_t w = win;
#endif
        return (uint8_t)(w >> (8 * (k & 3u)));
    }
#ifndef SWC_LZMA_UNIFORM_BRANCH
#define SWC_LZMA_UNIFORM_BRANCH 1
#endif
    SWC_HD static bool same(bool c) { return SWC_LZMA_UNIFORM_BRANCH ? simt::wave_true(c) : c; }
    SWC_HD void normalize() {     // LZMARangeDecoder.swift:38-43
        if (same(range < (1u << 24))) {
            range <<= 8;
            code = (code << 8) | next_byte();
        }
    }
#ifndef SWC_LZMA_BIT_SELECT
#define SWC_LZMA_BIT_SELECT 0
#endif
#ifndef SWC_LZMA_BIT_ASM
#define SWC_LZMA_BIT_ASM 1
#endif
    SWC_HD int bit(uint16_t* p) {  // LZMARangeDecoder.swift:65-80
#if defined(__HIP_DEVICE_COMPILE__) && SWC_LZMA_BIT_ASM
        // The decision as ONE block of gfx950 instructions: read, split the range, compare, ONE scalar branch, the side taken
        // (probability update, range / code), the store -- 13 instructions on the zero side, 12 on the one side.  From the C++
        // form below the compiler makes two conditional regions joined by a flag register (a move, a branch, an and-not and a
        // second branch more per decision), because it structures the if / else for lanes that might disagree; they cannot:
        // every value here is the same in all 64 lanes (vcc is all ones or zero).  The cell is in LDS: the low half of its
        // generic address is its LDS address.
        const uint32_t a = (uint32_t)(uintptr_t)p;
        uint32_t pr, bound;   // (bound doubles as the scratch register of the probability update once the range has taken it)
        int sym;
        asm volatile(
            "ds_read_u16 %[pr], %[a]\n\t"
            "v_lshrrev_b32 %[bound], 11, %[range]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mul_u32_u24 %[bound], %[bound], %[pr]\n\t"
            "v_cmp_lt_u32 vcc, %[code], %[bound]\n\t"
            "s_cbranch_vccz 1f\n\t"
            "v_mov_b32 %[range], %[bound]\n\t"
            "v_sub_u32 %[bound], 0x800, %[pr]\n\t"
            "v_lshrrev_b32 %[bound], 5, %[bound]\n\t"
            "v_add_u32 %[pr], %[pr], %[bound]\n\t"
            "s_mov_b32 %[sym], 0\n\t"
            "s_branch 2f\n"
            "1:\n\t"
            "v_sub_u32 %[code], %[code], %[bound]\n\t"
            "v_sub_u32 %[range], %[range], %[bound]\n\t"
            "v_lshrrev_b32 %[bound], 5, %[pr]\n\t"
            "v_sub_u32 %[pr], %[pr], %[bound]\n\t"
            "s_mov_b32 %[sym], 1\n"
            "2:\n\t"
            "ds_write_b16 %[a], %[pr]"
            : [pr] "=&v"(pr), [bound] "=&v"(bound), [sym] "=&s"(sym), [range] "+v"(range), [code] "+v"(code)
            : [a] "v"(a)
            : "vcc", "memory");
        normalize();
        return sym;
#else
        const uint32_t pr = *p;
        uint32_t bound = (range >> 11) * pr;
        if (SWC_LZMA_BIT_SELECT) {   // both sides computed, picked by selects on ONE scalar condition: no branch in the decision
            const bool zero = same(code < bound);
            const uint32_t pz = pr + ((2048u - pr) >> 5), po = pr - (pr >> 5);
            *p = (uint16_t)(zero ? pz : po);
            const uint32_t r1 = range - bound, c1 = code - bound;
            range = zero ? bound : r1;
            code = zero ? code : c1;
            normalize();
            return zero ? 0 : 1;
        }
        int sym;
        if (same(code < bound)) {
            *p = (uint16_t)(pr + ((2048u - pr) >> 5));
            range = bound;
            sym = 0;
        } else {
            *p = (uint16_t)(pr - (pr >> 5));
            code -= bound;
            range -= bound;
            sym = 1;
        }
        normalize();
        return sym;
#endif
    }
    SWC_HD int bit_spill(SWC_AS_GLOBAL uint16_t* p) {
        uint32_t pr = *p;
        uint32_t bound = (range >> 11) * pr;
        int sym;
        if (same(code < bound)) {
            if (lane == 0) *p = (uint16_t)(pr + ((2048u - pr) >> 5));
            range = bound;
            sym = 0;
        } else {
            if (lane == 0) *p = (uint16_t)(pr - (pr >> 5));
            code -= bound;
            range -= bound;
            sym = 1;
        }
        normalize();
        return sym;
    }
    SWC_HD uint32_t direct_bits(int count) {  // LZMARangeDecoder.swift:46-62
        uint32_t res = 0;
        do {
            range >>= 1;
            code -= range;
            uint32_t t = 0u - (code >> 31);
            code += range & t;
            normalize();
            res = (res << 1) + (t + 1);
            count--;
        } while (count > 0);
        return res;
    }
    // bit() for a cell whose value is already in a register
    SWC_HD int bit_known(uint16_t* p, uint32_t pr) {
        uint32_t bound = (range >> 11) * pr;
        int sym;
        if (same(code < bound)) {
            *p = (uint16_t)(pr + ((2048u - pr) >> 5));
            range = bound;
            sym = 0;
        } else {
            *p = (uint16_t)(pr - (pr >> 5));
            code -= bound;
            range -= bound;
            sym = 1;
        }
        normalize();
        return sym;
    }
#ifndef SWC_LZMA_PAIR_READ
#define SWC_LZMA_PAIR_READ 0
#endif
    // LZMABitTreeDecoder.swift:18-24.  SWC_LZMA_PAIR_READ (measured, off: 880 ms against 862, profiles/r04_experiments.txt): the
    // children of node m are the cells 2m and 2m + 1, ONE aligned dword (`p` starts at an even cell) that can be read while the
    // decision at m is still being taken, so that the LDS round trip leaves the serial chain -- but the two instructions it
    // adds per decision cost more than the latency it hides, as the chain microbenchmark of round 3 had said.
    SWC_HD int tree(uint16_t* p, int nbits) {
        int m = 1;
        if (!SWC_LZMA_PAIR_READ) {
            for (int i = 0; i < nbits; i++) m = (m << 1) + bit(&p[m]);
            return m - (1 << nbits);
        }
        uint32_t pr = p[1];
        for (int i = 0; i < nbits; i++) {
            uint32_t pair = 0;
            if (i + 1 < nbits) pair = *(const uint32_t*)__builtin_assume_aligned(p + 2 * m, 4);
            const int b = bit_known(&p[m], pr);
            pr = b ? pair >> 16 : pair & 0xFFFFu;
            m = (m << 1) + b;
        }
        return m - (1 << nbits);
    }
    SWC_HD int tree_reverse(uint16_t* p, int limit, int start, int bits) {  // :26-43
        int m = 1, sym = 0;
        bool bad = false;   // an index outside the table: the reference traps there; here the walk goes on over a clamped cell
        for (int i = 0; i < bits; i++) {   // (no test-and-leave per decision in the chain) and the trap is recorded behind the loop
            int idx = start + m;
            const bool out_of_range = idx < 0 || idx >= limit;
            bad = bad || out_of_range;
            idx = out_of_range ? 0 : idx;
            int b = bit(&p[idx]);
            m = (m << 1) + b;
            sym |= b << i;
        }
        trap = trap || bad;
        return bad ? 0 : sym;
    }
    SWC_HD int len_decode(uint16_t* p, int pos_state, int which) {  // LZMALenDecoder.swift:30-38; which: 0 length, 1 rep length
        if (bit(&p[LEN_CHOICE]) == 0) return tree(&p[LEN_LOW + pos_state * 8], 3);
        if (bit(&p[LEN_CHOICE2]) == 0) return 8 + tree(&p[LEN_MID + pos_state * 8], 3);
        if (cached) {   // the `high` trees are not in LDS (see kSlotBase)
            SWC_AS_GLOBAL uint16_t* hp = lit_spill + kSpillHighCell + (size_t)which * 256;
            int m = 1;
            for (int i = 0; i < 8; i++) m = (m << 1) + bit_spill(&hp[m]);
            return 16 + m - 256;
        }
        return 16 + tree(&probs[P_LEN_HIGH + which * 256], 8);
    }

    // The LDS copy of line `third` (0 plain tree, 1 / 2 the trees of the matched mode) of literal coder `c` (cache mode): a hit
    // is four compares and one branch; a miss writes the oldest line back and loads (or, if nobody has touched it since the
    // last reset, fills) the wanted one -- 512 B each way, two dwords per lane.
    SWC_HD uint16_t* literal_line(uint32_t c, uint32_t third) {
        // (Four lines, replaced round-robin.  Two ways for plain trees + two for the matched mode, so that matched literals cannot
        // evict plain trees, measured slower on every payload: 721 / 1,434 / 2,666 ms against 703 / 1,249 / 2,379.)
        const uint32_t L = c * 3u + third;
        {
            const uint32_t slot = (tag1 == L ? 1u : 0u) + (tag2 == L ? 2u : 0u) + (tag3 == L ? 3u : 0u);
            const bool hit = tag0 == L || slot != 0u;
#ifndef SWC_LZMA_KEEP_MRU
#define SWC_LZMA_KEEP_MRU 0   // (measured: text 612 against 621 ms, binary records 635 against 621, P-mix the same -- off)
#endif
            if (same(hit)) { if (SWC_LZMA_KEEP_MRU) mru = slot; return probs + kSlotBase + slot * 0x100; }
        }
        // (round-robin, but never the line that was used last: one scalar move on the hit path keeps the hot line -- on text the
        // plain tree of the lower-case coder -- out of the rotation)
        uint32_t v = victim;
        if (SWC_LZMA_KEEP_MRU && v == mru) v = v + 1 == (uint32_t)kLines ? 0u : v + 1;
        victim = v + 1 == (uint32_t)kLines ? 0u : v + 1;
        if (SWC_LZMA_KEEP_MRU) mru = v;
        const uint32_t old = v == 0 ? tag0 : v == 1 ? tag1 : v == 2 ? tag2 : tag3;
        uint16_t* sp = probs + kSlotBase + v * 0x100;
        uint32_t* sp32 = (uint32_t*)sp;
        if (same(old != kNoCoder)) {
            SWC_AS_GLOBAL uint32_t* home = (SWC_AS_GLOBAL uint32_t*)(lit_spill + (size_t)old * 0x100);
            for (int j = lane; j < 0x100 / 2; j += WAVE) home[j] = sp32[j];
        }
        const uint64_t fm = third == 0u ? fresh : third == 1u ? fresh1 : fresh2;
        if (same(c < 64 && ((fm >> (c & 63u)) & 1u))) {
            for (int j = lane; j < 0x100 / 2; j += WAVE) sp32[j] = 0x04000400u;
            const uint64_t bitc = 1ull << (c & 63u);
            fresh &= ~(third == 0u ? bitc : 0ull);
            fresh1 &= ~(third == 1u ? bitc : 0ull);
            fresh2 &= ~(third == 2u ? bitc : 0ull);
        } else {
            // (a line written back earlier comes back through the same lanes that stored it)
            const SWC_AS_GLOBAL uint32_t* home = (const SWC_AS_GLOBAL uint32_t*)(lit_spill + (size_t)L * 0x100);
            for (int j = lane; j < 0x100 / 2; j += WAVE) sp32[j] = home[j];
        }
        tag0 = v == 0 ? L : tag0; tag1 = v == 1 ? L : tag1; tag2 = v == 2 ? L : tag2; tag3 = v == 3 ? L : tag3;
        simt::wave_fence();
        return sp;
    }

    // LZMADecoder.swift:79-100.  All lanes initialise a slice of the model.
    SWC_HD void reset_state_and_decoders() {
        if (cached) {
            state = 0;
            rep0 = rep1 = rep2 = rep3 = 0;
            need_ws = lit_spill == nullptr;
            for (int i = lane; i < kSlotBase; i += WAVE) probs[i] = 1024;
            if (lit_spill) for (int i = lane; i < 512; i += WAVE) lit_spill[kSpillHighCell + (size_t)i] = 1024;   // the `high` length trees
            tag0 = tag1 = tag2 = tag3 = kNoCoder;   // (dropped, not written back: the model starts over)
            victim = 0;
            const int lit_bits = lc + lp;
            if (lit_bits <= 6) {
                fresh = lit_bits == 6 ? ~0ull : (1ull << (1u << lit_bits)) - 1ull;
                fresh1 = fresh2 = fresh;
            } else {
                fresh = fresh1 = fresh2 = 0;
                const uint32_t cells = 0x300u << lit_bits;
                if (lit_spill) for (uint32_t i = (uint32_t)lane; i < cells; i += WAVE) lit_spill[i] = 1024;
            }
            simt::wave_fence();
            have_model = true;
            return;
        }
        state = 0;
        rep0 = rep1 = rep2 = rep3 = 0;
        need_ws = false;
        const int lit_bits = lc + lp;
        const int lds_cells = P_LITERAL + (lit_bits <= lds_bits ? (0x300 << lit_bits) : 0);
        for (int i = lane; i < lds_cells; i += WAVE) probs[i] = 1024;
        if (lit_bits > lds_bits) {
            const uint32_t cells = 0x300u << lit_bits;
            if (lit_spill) for (uint32_t i = (uint32_t)lane; i < cells; i += WAVE) lit_spill[i] = 1024;
            else need_ws = true;
        }
        have_model = true;
    }
    SWC_HD void reset_dictionary() { dict_start = pos; }  // LZMADecoder.swift:102-104

    // put(): LZMADecoder.swift:288-294.  One byte, written by lane 0.
    SWC_HD void put(uint8_t b) {
        SWC_LZMA_PROF(9)
        if (same(pos < cap)) out[pos] = b;   // (every lane stores the same byte to the same address: one write, no second mask region)
        overflow = overflow || pos >= cap;
        prev_byte = b;
        pos++;
        dict_start += pos - dict_start == dict_size ? 1u : 0u;
    }
    // byte(at:): LZMADecoder.swift:296-298 -- out[distance <= dictEnd ? dictEnd - distance : dictSize - distance + dictEnd]
    SWC_HD uint8_t byte_at(uint64_t distance) {
        SWC_LZMA_PROF(2)
        uint64_t idx;
        if (same(distance <= pos)) idx = pos - distance;
        else { trap = true; return 0; }  // the wrap branch indexes at or past out.count (dictSize >= distance): Swift trap
        if (same(idx >= cap)) { overflow = true; return 0; }
        return out[idx];
    }
    // `len` bytes from `distance` back, spread over the wave (LZMADecoder.swift:278-282).
    SWC_HD void copy_match(uint64_t distance, uint32_t len) {
        SWC_LZMA_PROF(1)
        SWC_LZMA_COUNT(5, 1);
        const bool fits = pos + len <= cap;
        if (same(fits)) {
            gptr dst = out + pos;
            uint32_t last = 0;   // the byte this lane wrote last: the lane that wrote dst[len - 1] holds the new prev_byte
            if (same((len <= (uint32_t)WAVE) & (distance >= len))) {
                // the common case in one step: no loop, no overlap
                if ((uint32_t)lane < len) { last = dst[(int64_t)lane - (int64_t)distance]; dst[lane] = (uint8_t)last; }
            } else if (same(distance >= len)) {
                for (uint32_t i = (uint32_t)lane; i < len; i += WAVE) { last = dst[(int64_t)i - (int64_t)distance]; dst[i] = (uint8_t)last; }
            } else {
                // overlapping: every byte is a copy of one of the `distance` bytes before `pos`
                for (uint32_t i = (uint32_t)lane; i < len; i += WAVE) { last = dst[(int64_t)(i % (uint32_t)distance) - (int64_t)distance]; dst[i] = (uint8_t)last; }
            }
#if defined(__HIP_DEVICE_COMPILE__)
            prev_byte = (uint32_t)__builtin_amdgcn_readlane((int)last, __builtin_amdgcn_readfirstlane((int)((len - 1u) % (uint32_t)WAVE)));
#else
            prev_byte = last;
#endif
        }
        overflow = overflow || !fits;
        pos += len;
        const uint64_t span = pos - dict_start;
        dict_start = ((span >= dict_size) & (dict_size > 0)) ? pos - dict_size + 1 : dict_start;  // `len` put()s worth of window sliding
    }

    // LZMADecoder.swift:107-284.  Returns an swc_status.
    SWC_HD int decode() {
        SWC_LZMA_PROF(0)
        if (same(n - ip() < 5)) return SWC_E_LZMA_RANGE_DECODER_INIT_ERROR;  // LZMARangeDecoder.swift:21
        ensure_window();
        const uint8_t first = next_byte();
        code = 0;
        for (int i = 0; i < 4; i++) code = (code << 8) | next_byte();  // uint32().byteSwapped
        range = 0xFFFFFFFFu;
        if (same(first != 0)) return SWC_E_LZMA_RANGE_DECODER_INIT_ERROR;
        if (same(!have_model)) return SWC_E_REF_TRAP;  // `probabilities` is still empty: index trap at :119
        if (same(need_ws)) return SWC_E_NEED_WORKSPACE;
        const int lit_bits = lc + lp;
        const bool spill = !cached && lit_bits > lds_bits;

        // Where the input runs out is noticed lazily (reads past the end return zeros and harm nothing): ONE test per symbol,
        // in front of the write of its output, and inside every other error return, so that a stream that ends early reports the
        // reference's trap and not an error computed from the zeros behind it.  (On the device each such test is an exec-mask
        // region in the serial chain; there were six to eight per symbol.)
        for (;;) {
            if (same(overflow || wk > 4u * 64u - kSymbolBytes || uncompressed_size == 0)) {   // (one test in the common case)
                if (same(overflow)) return SWC_E_CAPACITY;
                ensure_window();
                if (same(uncompressed_size == 0 && code == 0)) break;  // :114
            }
            const int pos_state = (int)(pos & ((1u << pb) - 1));
            int is_match;
            { SWC_LZMA_PROF(8) is_match = bit(&probs[P_IS_MATCH + (state << 4) + pos_state]); }
            if (is_match == 0) {
                if (same(uncompressed_size == 0)) return trapped() ? SWC_E_REF_TRAP : SWC_E_LZMA_EXCEEDED_UNCOMPRESSED_SIZE;  // :121
                SWC_LZMA_PROF(3)
                SWC_LZMA_COUNT(6, 1);
                const uint32_t prev = pos == dict_start ? 0u : prev_byte;
                const uint32_t lit_state = (uint32_t)(((pos & ((1u << lp) - 1)) << lc) + (prev >> (8 - lc)));
                int symbol = 1;
                if (cached) {
                    if (same(state >= 7)) {
                        SWC_LZMA_PROF(10)
                        SWC_LZMA_COUNT(11, 1);
                        uint32_t match_byte = byte_at(rep0 + 1);
                        do {
                            const int match_bit = (match_byte >> 7) & 1;
                            match_byte = (match_byte << 1) & 0xFF;
                            uint16_t* lm = literal_line(lit_state, 1u + (uint32_t)match_bit);
                            const int b = bit(&lm[symbol]);
                            symbol = (symbol << 1) | b;
                            if (same(match_bit != b)) break;
                        } while (symbol < 0x100);
                    }
                    if (symbol < 0x100) {
                        uint16_t* lpb = literal_line(lit_state, 0u);
                        if (symbol == 1) {   // not in matched mode: exactly eight decisions (a constant trip count: no loop test in the chain)
#pragma unroll
                            for (int i = 0; i < 8; i++) symbol = (symbol << 1) | bit(&lpb[symbol]);
                        } else {
                            while (symbol < 0x100) symbol = (symbol << 1) | bit(&lpb[symbol]);
                        }
                    }
                } else if (!spill) {
                    uint16_t* lpb = &probs[P_LITERAL + lit_state * 0x300];
                    if (same(state >= 7)) {
                        SWC_LZMA_PROF(10)
                        SWC_LZMA_COUNT(11, 1);
                        uint32_t match_byte = byte_at(rep0 + 1);
                        do {
                            const int match_bit = (match_byte >> 7) & 1;
                            match_byte = (match_byte << 1) & 0xFF;
                            const int b = bit(&lpb[((1 + match_bit) << 8) + symbol]);
                            symbol = (symbol << 1) | b;
                            if (same(match_bit != b)) break;
                        } while (symbol < 0x100);
                    }
                    if (symbol == 1) {   // not in matched mode: exactly eight decisions (a constant trip count: no loop test in the chain)
                        if (SWC_LZMA_PAIR_READ) symbol = 0x100 + tree(lpb, 8);   // (every literal coder starts at an even cell)
                        else {
#pragma unroll
                            for (int i = 0; i < 8; i++) symbol = (symbol << 1) | bit(&lpb[symbol]);
                        }
                    } else {
                        while (symbol < 0x100) symbol = (symbol << 1) | bit(&lpb[symbol]);
                    }
                } else {
                    SWC_AS_GLOBAL uint16_t* lpb = lit_spill + (size_t)lit_state * 0x300;
                    if (state >= 7) {
                        uint32_t match_byte = byte_at(rep0 + 1);
                        do {
                            const int match_bit = (match_byte >> 7) & 1;
                            match_byte = (match_byte << 1) & 0xFF;
                            const int b = bit_spill(&lpb[((1 + match_bit) << 8) + symbol]);
                            symbol = (symbol << 1) | b;
                            if (same(match_bit != b)) break;
                        } while (symbol < 0x100);
                    }
                    while (symbol < 0x100) symbol = (symbol << 1) | bit_spill(&lpb[symbol]);
                }
                if (same(trapped())) return SWC_E_REF_TRAP;
                uncompressed_size -= 1;
                put((uint8_t)(symbol - 0x100));
                state = state < 4 ? 0 : state < 10 ? state - 3 : state - 6;
                continue;
            }

            uint32_t len;
            bool is_rep;
            {
            SWC_LZMA_PROF(4)
            is_rep = bit(&probs[P_IS_REP + state]) != 0;
            if (is_rep) {
                if (same(uncompressed_size == 0)) return trapped() ? SWC_E_REF_TRAP : SWC_E_LZMA_EXCEEDED_UNCOMPRESSED_SIZE;  // :178
                if (same(pos == dict_start)) return trapped() ? SWC_E_REF_TRAP : SWC_E_LZMA_WINDOW_IS_EMPTY;                  // :181
                if (bit(&probs[P_IS_REP_G0 + state]) == 0) {
                    if (same((state << 4) + pos_state >= 191)) return SWC_E_REF_TRAP;            // reference index 241+... == 432
                    if (bit(&probs[P_IS_REP0_LONG + (state << 4) + pos_state]) == 0) {
                        state = state < 7 ? 9 : 11;
                        SWC_LZMA_COUNT(7, 1);
                        const uint8_t b = byte_at(rep0 + 1);
                        if (same(trapped())) return SWC_E_REF_TRAP;
                        put(b);
                        uncompressed_size -= 1;
                        continue;
                    }
                } else {
                    uint64_t dist;
                    if (bit(&probs[P_IS_REP_G1 + state]) == 0) {
                        dist = rep1;
                    } else {
                        if (bit(&probs[P_IS_REP_G2 + state]) == 0) {
                            dist = rep2;
                        } else {
                            dist = rep3;
                            rep3 = rep2;
                        }
                        rep2 = rep1;
                    }
                    rep1 = rep0;
                    rep0 = dist;
                }
                len = (uint32_t)len_decode(&probs[P_REP_LEN], pos_state, 1);
                state = state < 7 ? 8 : 11;
            } else {
                rep3 = rep2; rep2 = rep1; rep1 = rep0;
                len = (uint32_t)len_decode(&probs[P_LEN], pos_state, 0);
                state = state < 7 ? 7 : 10;
                const int len_state = len > 3 ? 3 : (int)len;
                const int pos_slot = tree(&probs[P_POS_SLOT + len_state * 64], 6);
                if (pos_slot < 4) {
                    rep0 = (uint64_t)pos_slot;
                } else {
                    const int ndb = (pos_slot >> 1) - 1;
                    uint64_t dist = (uint64_t)(2 | (pos_slot & 1)) << ndb;
                    if (pos_slot < 14) {
                        dist += (uint64_t)tree_reverse(&probs[P_POS_DEC], 115, (int)(dist - (uint64_t)pos_slot), ndb);
                    } else {
                        dist += (uint64_t)direct_bits(ndb - 4) << 4;
                        dist += (uint64_t)tree_reverse(&probs[P_ALIGN], 16, 0, 4);
                    }
                    rep0 = dist;
                }
            }
            }
            // Everything that can stop a match -- the end marker, the declared size, the window, the end of the input -- in ONE
            // test (all input of the symbol has been read): the checks of the reference, in its order, sit behind it.
            len += 2;
            {
                const bool end_marker = !is_rep && rep0 == 0xFFFFFFFFull;
                const bool odd = end_marker | (!is_rep & (uncompressed_size == 0)) | (!is_rep & ((rep0 >= dict_size) | ((rep0 > pos) & (pos < dict_size))))
                               | trapped() | ((uncompressed_size > -1) & (uncompressed_size < (int64_t)len)) | (rep0 + 1 > pos);
                if (same(odd)) {
                    if (same(end_marker)) {                                                          // :260
                        if (same(code != 0)) return trapped() ? SWC_E_REF_TRAP : SWC_E_LZMA_RANGE_DECODER_FINISH_ERROR;   // :261
                        break;
                    }
                    if (same(!is_rep)) {
                        if (same(uncompressed_size == 0)) return trapped() ? SWC_E_REF_TRAP : SWC_E_LZMA_EXCEEDED_UNCOMPRESSED_SIZE;  // :266
                        if (same(rep0 >= dict_size || (rep0 > pos && pos < dict_size))) return trapped() ? SWC_E_REF_TRAP : SWC_E_LZMA_NOT_ENOUGH_TO_REPEAT;  // :269
                    }
                    if (same(trapped())) return SWC_E_REF_TRAP;
                    if (same(uncompressed_size > -1 && uncompressed_size < (int64_t)len)) return SWC_E_LZMA_REPEAT_WILL_EXCEED;  // :275
                    if (same(rep0 + 1 > pos)) return SWC_E_REF_TRAP;  // byte(at:) would index past out.count (App. A L2)
                }
            }
            copy_match(rep0 + 1, len);
            uncompressed_size -= len;
        }
        if (same(trapped())) return SWC_E_REF_TRAP;
        if (same(overflow)) return SWC_E_CAPACITY;
        return SWC_OK;
    }

    // LZMA2Decoder.init + decode(): Sources/LZMA2/LZMA2Decoder.swift:17-99
    SWC_HD int decode_lzma2(uint8_t dict_byte) {
        if (same(dict_byte & 0xC0)) return SWC_E_LZMA2_WRONG_DICTIONARY_SIZE;  // :21
        const int bits = dict_byte & 0x3F;
        if (same(bits >= 40)) return SWC_E_LZMA2_WRONG_DICTIONARY_SIZE;        // :24
        uint32_t ds = (uint32_t)(2 | (bits & 1)) << (bits / 2 + 11);
        dict_size = ds < 4096 ? 4096 : ds;                               // didSet clamp, LZMAProperties.swift:26-32
        for (;;) {
            ensure_window();
            const uint32_t control = next_byte();                        // :36
            if (same(trapped())) return SWC_E_REF_TRAP;
            if (same(control == 0)) return SWC_OK;
            if (same(control == 1 || control == 2)) {
                if (same(control == 1)) reset_dictionary();
                const uint32_t b1 = next_byte(), b2 = next_byte();       // decodeUncompressed :84-89
                if (same(trapped())) return SWC_E_REF_TRAP;
                const uint64_t size = ((uint64_t)b1 << 8) + b2 + 1;
                if (same(n - ip() < size)) return SWC_E_REF_TRAP;                // byte() past the end inside the copy loop
                if (same(pos + size <= cap)) {
                    for (uint64_t i = (uint64_t)lane; i < size; i += WAVE) out[pos + i] = in[ip() + i];
                    prev_byte = in[ip() + size - 1];
                } else {
                    overflow = true;
                }
                wk += (uint32_t)size;   // (<= 65,536; the loop top reloads the window)
                pos += size;
                if (pos - dict_start >= dict_size) dict_start = pos - dict_size + 1;
                if (same(overflow)) return SWC_E_CAPACITY;
                continue;
            }
            if (same(control <= 0x7F)) return SWC_E_LZMA2_WRONG_CONTROL_BYTE;  // :45
            const int reset = (control & 0x60) >> 5;                     // dispatch :56-82
            const uint32_t u1 = next_byte(), u2 = next_byte();
            const int64_t unpack = ((int64_t)(control & 0x1F) << 16) + ((int64_t)u1 << 8) + u2 + 1;
            const uint32_t c1 = next_byte(), c2 = next_byte();
            const int64_t comp = ((int64_t)c1 << 8) + c2 + 1;
            if (same(trapped())) return SWC_E_REF_TRAP;
            if (same(reset == 1)) {
                reset_state_and_decoders();
            } else if (same(reset >= 2)) {                                     // updateProperties :95-99
                const uint32_t pbyte = next_byte();
                if (same(trapped())) return SWC_E_REF_TRAP;
                if (same(pbyte >= 225)) return SWC_E_LZMA_WRONG_PROPERTIES;
                lc = pbyte % 9; pb = (pbyte / 9) / 5; lp = (pbyte / 9) % 5;
                reset_state_and_decoders();
                if (same(reset == 3)) reset_dictionary();
            }
            uncompressed_size = unpack;
            const uint64_t out_start = pos, in_start = ip();
            const int st = decode();
            if (same(st)) return st;
            if (same(!(unpack == (int64_t)(pos - out_start) && (int64_t)(ip() - in_start) == comp))) return SWC_E_LZMA2_WRONG_SIZES;  // :79-81
        }
    }
};

// job.aux: LZMA2 = dictionary-size byte; LZMA = lc | lp << 8 | pb << 16.
// job.dict_len: LZMA = declared uncompressed size (UINT64_MAX = unknown); job.dict (reinterpreted) = dictionary size.
template <int WAVE>
SWC_HD void lzma_job(Job& job, bool is_lzma2, uint16_t* probs, SWC_AS_GLOBAL uint16_t* lit_spill, int lane, int lds_bits = kMaxLdsLitBits, uint64_t* prof = nullptr, bool cached = false) {
    Decoder<WAVE> d;
    d.cached = cached;
    d.in = (gcptr)job.in; d.n = job.in_len;
    d.out = (gptr)job.out; d.cap = job.out_cap; d.pos = 0;
    d.lane = lane;
    d.probs = probs; d.lit_spill = lit_spill; d.lds_bits = lds_bits;
    d.have_model = false;
    d.lc = 3; d.lp = 0; d.pb = 2;             // LZMAProperties defaults, LZMAProperties.swift:12-18
    d.dict_size = 1u << 24;
    d.uncompressed_size = -1;
    d.dict_start = 0;
    d.range = 0; d.code = 0;
    d.rep0 = d.rep1 = d.rep2 = d.rep3 = 0;
    d.state = 0;
    d.trap = false; d.overflow = false; d.need_ws = false;
    d.win = 0; d.prev_byte = 0;
    d.win_load(0);
    int st;
    if (is_lzma2) {
        st = d.decode_lzma2((uint8_t)job.aux);
    } else {
        d.lc = job.aux & 0xFF; d.lp = (job.aux >> 8) & 0xFF; d.pb = (job.aux >> 16) & 0xFF;
        d.dict_size = (uint64_t)(uintptr_t)job.dict;
        d.uncompressed_size = (int64_t)job.dict_len;
        if (d.lc > 8 || d.lp > 4 || d.pb > 4) {
            st = SWC_E_REF_TRAP;  // LZMAProperties(lc:lp:pb:) is not validated; such values trap downstream
        } else {
            d.reset_state_and_decoders();
            st = d.decode();
        }
    }
    job.out_len = d.pos;
    job.in_consumed = d.ip() < d.n ? d.ip() : d.n;   // (a read past the end keeps counting, see next_byte())
    job.status = st;
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    if (prof && lane == 0) for (int k = 0; k < 16; k++) prof[k] = d.pacc[k];
#else
    (void)prof;
#endif
}

}  // namespace lzma
}  // namespace swc
#endif
