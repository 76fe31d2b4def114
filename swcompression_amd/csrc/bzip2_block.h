// bzip2_block.h -- BZip2 block decode as three device stages.
//
// Replaces BZip2.decode(_:_:) (reference Sources/BZip2/BZip2.swift:97-270), BurrowsWheeler.reverse
// (Sources/BZip2/BurrowsWheeler.swift:29-64) and CheckSums.bzip2crc32 (Sources/Common/CheckSums.swift:30-37):
//
//   stage 1  one block per WAVEFRONT  symbol map, selectors (MTF), 2-6 Huffman tables, symbol loop with
//            RUNA/RUNB runs and inverse MTF  ->  the BWT last column L[0..n) in the HBM workspace.  The decode chain
//            is serial and a wave follows it at the LATENCY of one symbol, so the symbol loop keeps everything it
//            touches per symbol in REGISTERS spread over the 64 lanes -- the code-length limits and index deltas of
//            the active table (lane d: length d), its symbols (two per lane and register), the MTF list (four bytes
//            per lane, shifted with one cross-lane move), the staged output (a byte per lane) -- and reads them with
//            ballots and v_readlane: no LDS round trip in the chain.  The tables of all 2-6 codes stay in LDS and are
//            loaded into the registers at a table switch (every 50 symbols).
//   stage 2  one block per WAVEFRONT  stable counting-sort scatter  P[base[c]++] = i << 8 | c  (c = L[i]): L is read 64
//            consecutive bytes at a time (one coalesced load), a lane finds its rank among the equal bytes of the group with
//            eight ballots (one per bit of the byte), the running counts of the 256 byte values sit in LDS.
//   (stage 3 of every launch but the tiny ones runs as kernels of its own, bzip2_team.h: the same cut of the cycle into segments,
//   16,384 of them, walked by ticket queues per XCD out of the L2; what follows describes the form inside this kernel)
//   stage 3a one block per WAVEFRONT  the n dependent gathers  v = P[end]; end = v >> 8; byte = v & 255  form ONE cycle
//            through the block, which a single walker can only follow at one HBM round trip per byte (measured: 17 G
//            gathers/s with a lane per block, 46-59 G/s -- the HBM row-activation limit -- with 64+ walkers per block,
//            tools/micro/gather_bench.hip).  So the cycle is CUT at marked indices (multiples of M, and origPtr): every
//            lane walks segments from a mark to the next mark, keeping the bytes it meets in the segment's buffer; the
//            <= 513 segments are then put in cycle order by following segment -> next segment from origPtr, which gives
//            every segment its offset in the block, and the buffers are copied to their places (segments longer than
//            their buffer are walked a second time).  A permutation that is not one n-cycle (damaged input) falls back
//            to the serial walk in stage 3b.
//   stage 3b one block per LANE       RLE1 undo (4 equal bytes + count, BZip2.swift:251-267) over the walked bytes, output
//            stores, capacity accounting.
//   stage 3c one block per WORKGROUP  MSB-first CRC-32 of the output (crc32_group.h) and the comparison with the stored
//            block CRC (BZip2.swift:81).
//
// Reference semantics kept: lengths 0..20 accepted and never validated (App. A B2), surplus selectors
// ignored / too few => wrongSelector (B3), origPtr unchecked => trap (B4), block size not enforced (B5),
// positional RLE1 rule (B6).  A FINAL code length > 20, which only the unchecked last delta can produce (BZip2.swift:185
// runs before the deltas of a symbol), is decoded like any other up to 26 bits; beyond that the reference's tree
// (DecodingTree.swift:19: 2^(maxBits+1) Ints) is an allocation of 2 GiB and more, classified trap-class.
#ifndef SWC_BZIP2_BLOCK_H
#define SWC_BZIP2_BLOCK_H

#include "swc_common.h"
#include "simt.h"
#include <type_traits>

namespace swc {
namespace bzip2 {

constexpr int kMaxSyms = 258;
constexpr int kMaxTables = 6;
constexpr int kMaxLen = 26;    // 20 is the checked limit (BZip2.swift:185); the last symbol's length is unchecked
constexpr int kCheckedLen = 20;

// ---- LDS layout of stage 1 (bytes) ---------------------------------------------------------------
struct Stage1Lds {
    uint32_t delta[kMaxTables][kMaxLen + 1]; // delta[d] = start[d] - first[d] (first[d] = the first code of length d): sym index = code + delta
    uint16_t start[kMaxTables][kMaxLen + 2]; // start[d] = index in sym[] of the first symbol of length d, start[kMaxLen + 1] = total
    uint32_t lim[kMaxTables][kMaxLen + 1];   // left-justified (kMaxLen-bit) limits, non-decreasing
    uint32_t oversub[kMaxTables];
    uint16_t sym[kMaxTables][kMaxSyms + 2];  // (length, symbol)-sorted symbols
    int8_t lengths[kMaxSyms + 6];            // scratch while building one table
    alignas(4) uint8_t mtf[256];             // usedSymbols as the header leaves it (BZip2.swift:128-137); the symbol loop keeps the list in registers
    uint8_t tmtf[8];                         // selector MTF
    uint32_t tcnt[kMaxLen + 2];              // scratch while building one table
    uint32_t tstart[kMaxLen + 2];
};
constexpr int kStage1LdsBytes = (sizeof(Stage1Lds) + 15) / 16 * 16;

constexpr uint32_t kSegs = 512;        // stage 3a: regular segments per block (+1 for origPtr)
#ifndef SWC_BZ_SEGCAP
#define SWC_BZ_SEGCAP 4
#endif
constexpr uint32_t kSegCapFactor = SWC_BZ_SEGCAP;  // a segment's buffer holds 4 x the mean segment length (2 % are longer, 9 % of the bytes)
constexpr int kParts = 64;             // stage 3a: the RLE1 undo of a block is cut into this many parts

// workspace per job (HBM): L[lcap] (stage 3: the walked bytes T) | selectors[32768] | P[lcap] (u32) | header | segment info | segment buffers
struct BlockHeader {   // stage 1 -> stage 2/3
    uint32_t n;        // length of L
    uint32_t orig_ptr;
    uint32_t status;
    uint32_t pad;
    uint64_t end_bit;  // absolute bit position just past the block's EOB symbol
};
// stage 3a -> 3b, kept in BlockHeader::pad
constexpr uint32_t kWalkNone = 0;     // nothing walked: stage 3b does the serial walk itself
constexpr uint32_t kWalkReady = 1;    // L[0..n) holds the bytes in cycle order (only while stage 3a runs)
constexpr uint32_t kWalkDone = 2;     // stage 3a also undid RLE1: output, out_len and status are final up to the CRC
// segment geometry for a block of n bytes: marks at multiples of 1 << mbits, at most kSegs of them
SWC_HD uint32_t seg_mbits(uint32_t n) {
    uint32_t m = 0;
    while (((n + (1u << m) - 1) >> m) > kSegs) m++;
    return m;
}
constexpr uint32_t kSegs2 = 16384, kTeamWords = 512;   // the team walk (bzip2_team.h): segments per block, words of its own per block
SWC_HD uint32_t seg_mbits2(uint32_t n) {
    uint32_t m = 5;
    while (((n + (1u << m) - 1) >> m) > kSegs2) m++;
    return m;
}
SWC_HD size_t segbuf_bytes(size_t lcap) {   // (regular segments + 1) x capacity, for the largest n = lcap; the larger of the two walks' needs
    const uint32_t m = seg_mbits((uint32_t)lcap), m2 = seg_mbits2((uint32_t)lcap);
    const size_t regs = (lcap + ((size_t)1 << m) - 1) >> m, regs2 = (lcap + ((size_t)1 << m2) - 1) >> m2;
    const size_t a = (regs + 1) * ((size_t)kSegCapFactor << m), b = (regs2 + 1) * ((size_t)kSegCapFactor << m2);
    return (a > b ? a : b) + 64;
}
// The team walk (bzip2_team.h) cuts a block into up to kSegs2 segments (of 32 bytes and more: a small block has few) and keeps
// four words per segment -- length and successor, offset, the index where a long segment's buffer was full -- and kTeamWords
// of its own (ticket counters, prefix of the segment counts) in the same area; the fused kernel uses the first 2 x (kSegs + 1) words.
SWC_HD uint32_t team_seg_slots(size_t lcap) {     // segments a block of at most lcap bytes can have (+ origPtr's, + one), a multiple of four
    const size_t by_size = (lcap + 31) / 32;
    return (uint32_t)((by_size < kSegs2 ? by_size : kSegs2) + 2 + 3) & ~3u;
}
SWC_HD size_t seg_info_bytes(size_t lcap) {        // length and successor of every segment (stage 3a)
    const size_t team = 4 * (size_t)team_seg_slots(lcap), fused = 2 * (size_t)(kSegs + 1);
    return ((((team > fused ? team : fused) + kTeamWords) * 4 + 15) / 16) * 16;
}
SWC_HD size_t ws_bytes_per_job(size_t lcap) {
    return ((lcap + 15) & ~(size_t)15) + 32768 + lcap * 4 + sizeof(BlockHeader) + 64 + seg_info_bytes(lcap) + ((segbuf_bytes(lcap) + 15) & ~(size_t)15);
}
struct Workspace {
    gptr L;
    gptr selectors;
    SWC_AS_GLOBAL uint32_t* P;
    SWC_AS_GLOBAL BlockHeader* hdr;
    SWC_AS_GLOBAL uint32_t* seg_len;    // [kSegs + 1] bytes of segment s
    SWC_AS_GLOBAL uint32_t* seg_next;   // [kSegs + 1] the segment that starts where s ends
    gptr segbuf;
    size_t lcap;
};
SWC_HD Workspace carve(uint8_t* base, size_t job, size_t lcap) {
    uint8_t* p = base + job * ws_bytes_per_job(lcap);
    Workspace w;
    size_t lpad = (lcap + 15) & ~(size_t)15;
    w.L = (gptr)p;
    w.selectors = (gptr)(p + lpad);
    w.P = (SWC_AS_GLOBAL uint32_t*)(p + lpad + 32768);
    w.hdr = (SWC_AS_GLOBAL BlockHeader*)(p + lpad + 32768 + lcap * 4);
    const size_t info = (lpad + 32768 + lcap * 4 + sizeof(BlockHeader) + 64 + 15) & ~(size_t)15;
    w.seg_len = (SWC_AS_GLOBAL uint32_t*)(p + info);
    w.seg_next = w.seg_len + (kSegs + 1);
    w.segbuf = (gptr)(p + info + seg_info_bytes(lcap));
    w.lcap = lcap;
    return w;
}

// MSB-first bit reader (BitByteData.MsbBitReader contract), wave-uniform, with one dword of read-ahead so that the
// memory latency of a refill is hidden behind the symbols decoded from the window.
struct MsbReader {
    gcptr in;
    uint64_t n;        // bytes
    uint64_t next;     // next byte to load
    uint64_t bb;       // next bit at bit 63
    uint32_t bc;
    uint32_t pw;       // the dword at `next`, already loaded (pw_ok)
    bool pw_ok;
    SWC_HD void init(gcptr p, uint64_t nbytes, uint64_t start_bit) {
        // (the whole state is declared wave-uniform once, here and where bytes come in: the symbol loop branches on it, and a
        // value the compiler cannot prove uniform turns every one of those branches into an exec-mask region)
        start_bit = simt::uniform(start_bit);
        in = p; n = simt::uniform(nbytes); next = start_bit >> 3; bb = 0; bc = 0; pw = 0; pw_ok = false;
        refill();
        uint32_t skip = (uint32_t)(start_bit & 7);
        if (skip > bc) skip = bc;
        bb <<= skip; bc -= skip;
        refill();
    }
    SWC_HD void refill() {
        if (bc <= 32) {
            if (next + 4 <= n) {
                uint32_t w = simt::uniform(pw_ok ? pw : load_u32(in + next));
                w = (w >> 24) | ((w >> 8) & 0xFF00u) | ((w << 8) & 0xFF0000u) | (w << 24);
                bb |= (uint64_t)w << (32 - bc);
                bc += 32;
                next += 4;
                pw_ok = next + 4 <= n;
                if (pw_ok) pw = load_u32(in + next);   // (made uniform where it is used: a readfirstlane here would wait for the load)
            } else {
                pw_ok = false;
                while (bc <= 56 && next < n) {
                    bb |= (uint64_t)simt::uniform((uint32_t)in[next++]) << (56 - bc);
                    bc += 8;
                }
            }
        }
    }
    SWC_HD int64_t bits_left() const { return (int64_t)(n - next) * 8 + bc; }
    SWC_HD uint64_t position() const { return next * 8 - bc; }
    SWC_HD uint32_t peek(uint32_t k) const { return k ? (uint32_t)(bb >> (64 - k)) : 0u; }
    SWC_HD void consume(uint32_t k) { bb <<= k; bc -= k; }
    SWC_HD uint32_t bits(uint32_t k) {  // k <= 24, caller checked bits_left() >= k
        refill();
        uint32_t v = peek(k);
        consume(k);
        return v;
    }
};

#if defined(__HIP_DEVICE_COMPILE__)
SWC_D uint32_t wave_count(bool pred) { return (uint32_t)__popcll(__ballot(pred)); }
#endif

// ---- stage 3a, first part: the walk of the segments, RESUMABLE ------------------------------------------------------------
// Every lane walks segments (v = P[cur]; byte = v & 255; cur = v >> 8) until none is left.  One call of walk_tick() takes
// every lane ONE step: it consumes the pointer loaded by the call before, and issues the next load without waiting for it --
// so whoever calls it can do other work while the gathers are in flight (the pipelined kernel runs the symbol loop of the
// NEXT block between two ticks; the fused kernel just calls it in a loop).  What the walk leaves per segment -- length and
// successor -- goes to the workspace, the ticket counter is the only LDS it needs (ctl[0]; ctl[1]: a walk ran away).
SWC_HD uint32_t take_ticket(uint32_t* t) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(t, 1u);
#else
    return (*t)++;
#endif
}
struct Walk {
    bool live = false;   // there is a block being walked (false: nothing to do, stage 3b reports the block)
    Workspace ws;
    uint32_t n, orig, mbits, mask, regs, segs, cap;
    bool extra;          // origPtr is a mark of its own
    uint32_t* ctl;
    // this lane's segment
    bool active, pending;
    uint32_t sg, cur, k, v;
    uint64_t acc;        // eight bytes per store: a byte per step would be a partial-line write to HBM each time
    SWC_HD bool is_mark(uint32_t i) const { return (i & mask) == 0 || i == orig; }
    SWC_HD uint32_t seg_of(uint32_t i) const { return (extra && i == orig) ? regs : i >> mbits; }
    SWC_HD uint32_t start_of(uint32_t s) const { return s < regs ? s << mbits : orig; }
};
SWC_HD void walk_begin(Walk& W, Workspace ws, uint32_t* ctl, int lane) {
    W.live = false;
    W.ws = ws;
    W.ctl = ctl;
    W.active = W.pending = false;
    if (lane == 0) ws.hdr->pad = kWalkNone;
    if (ws.hdr->status != SWC_OK) return;
    W.n = ws.hdr->n; W.orig = ws.hdr->orig_ptr;
    if (W.n == 0 || W.orig >= W.n) return;                   // empty block / trap: stage 3b reports it
    W.mbits = seg_mbits(W.n); W.mask = (1u << W.mbits) - 1u;
    W.regs = (W.n + W.mask) >> W.mbits;                      // marks 0, M, 2M, ...
    W.extra = (W.orig & W.mask) != 0;
    W.segs = W.regs + (W.extra ? 1u : 0u);
    W.cap = kSegCapFactor << W.mbits;
    if (lane == 0) { ctl[0] = 0; ctl[1] = 0; }
    W.live = true;
}
// one step of every lane; true when no segment is left and none is being walked
template <int WAVE>
SWC_HD bool walk_tick(Walk& W) {
    if (W.pending) {
        const uint32_t v = W.v;
        gptr buf = W.ws.segbuf + (size_t)W.sg * W.cap;       // the first `cap` bytes of a segment go to its buffer
        W.acc |= (uint64_t)(v & 0xFFu) << (8 * (W.k & 7u));
        if ((W.k & 7u) == 7u) {
            const uint32_t g0 = W.k - 7;
            if (g0 + 8 <= W.cap) store_u64(buf + g0, W.acc);
            else for (uint32_t j = 0; j < 8; j++) if (g0 + j < W.cap) buf[g0 + j] = (uint8_t)(W.acc >> (8 * j));
            W.acc = 0;
        }
        W.k++;
        W.cur = v >> 8;
        if (W.is_mark(W.cur) || W.k > W.n) {
            for (uint32_t j = W.k & ~7u; j < W.k; j++) if (j < W.cap) buf[j] = (uint8_t)(W.acc >> (8 * (j & 7u)));   // pending bytes
            if (W.k > W.n) W.ctl[1] = 1;                     // cannot happen for a permutation; guards the loops of walk_finish
            W.ws.seg_len[W.sg] = W.k;
            W.ws.seg_next[W.sg] = W.seg_of(W.cur);
            W.active = false;
        }
        W.pending = false;
    }
    if (!W.active) {
        const uint32_t s = take_ticket(&W.ctl[0]);
        if (s < W.segs) { W.active = true; W.sg = s; W.cur = W.start_of(s); W.k = 0; W.acc = 0; }
    }
    if (W.active) { W.v = W.ws.P[W.cur]; W.pending = true; }
#if defined(__HIP_DEVICE_COMPILE__)
    return WAVE > 1 ? __ballot(W.active) == 0ull : !W.active;
#else
    return !W.active;
#endif
}

// CXX: the plain-symbol loop from its C++ twin instead of the assembly (the host build always; on the device the tuning value
// "bzip2_hot_cxx" selects a second instantiation of the kernel, which the GPU tier compares with the default one)
template <int WAVE, bool CXX = false>
struct Stage1 {
    Stage1Lds* s;
    MsbReader br;
    Workspace ws;
    int lane;
    uint32_t n_out;     // bytes in L so far
    uint32_t lcap32;    // min(ws.lcap, 2^32 - 1)

    // Selectors were stored to HBM by lane 0; lane 0 reads them back (same-lane store -> load order is
    // architectural) and the value is broadcast, so no cross-lane memory visibility is assumed.
    SWC_HD int selector_at(int i) {
        int v = lane == 0 ? (int)ws.selectors[i] : 0;
#if defined(__HIP_DEVICE_COMPILE__)
        if (WAVE > 1) v = __builtin_amdgcn_readfirstlane(v);
#endif
        return v;
    }

    // Code.huffmanCodes + DecodingTree.init for table t from s->lengths[0..count)  (Code.swift:15-39)
    SWC_HD int build_table(int t, int count) {
        uint32_t* cnt = s->tcnt;
        uint32_t* start = s->tstart;
        for (int d = 0; d <= kMaxLen + 1; d++) cnt[d] = 0;
        for (int i = 0; i < count; i++) {
            int l = s->lengths[i];
            if (l > kMaxLen) return SWC_E_REF_TRAP;  // see header: only the unchecked final length can get here
            if (l > 0) cnt[l]++;
        }
        uint32_t v = 0, off = 0;
        bool over = false;
        for (int d = 1; d <= kMaxLen; d++) {
            s->lim[t][d] = (v + cnt[d]) << (kMaxLen - d);
            if (cnt[d] != 0 && v + cnt[d] > (1u << d)) over = true;
            s->delta[t][d] = off - v;
            s->start[t][d] = (uint16_t)off;
            start[d] = off;
            off += cnt[d];
            v = (v + cnt[d]) << 1;
        }
        s->start[t][kMaxLen + 1] = (uint16_t)off;
        s->oversub[t] = over ? 1u : 0u;
        for (int i = 0; i < count; i++) {
            int l = s->lengths[i];
            if (l > 0) s->sym[t][start[l]++] = (uint16_t)i;
        }
        return SWC_OK;
    }

    // DecodingTree.findNextSymbol for table t: returns the symbol or -1.
    // (The symbol loop is bound by the SCALAR instructions it issues -- one block per wave makes it wave-uniform, and all
    // waves of a CU share one scalar unit -- so everything that does not change between two table switches is kept out of
    // it: `over` is oversub[t]; the 64-bit "bits left" arithmetic runs only when the window itself is short.)
    SWC_HD int decode_symbol(int t, uint32_t my_lim, bool over) {
        br.refill();
        const uint32_t c = br.peek(kMaxLen);
        uint32_t len;
        int idx = -1;
        if (!over) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (WAVE > 1) {
                len = 1 + wave_count(lane >= 1 && lane <= kMaxLen && c >= my_lim);
            } else
#endif
            {
                len = 1;
                for (int d = 1; d <= kMaxLen; d++) len += c >= s->lim[t][d] ? 1u : 0u;
            }
            (void)my_lim;
            if (len > kMaxLen) return -1;
            idx = (int)((c >> (kMaxLen - len)) + s->delta[t][len]);
        } else {
            len = 0;
            for (int d = 1; d <= kMaxLen; d++) {
                const uint32_t st0 = s->start[t][d];
                const uint32_t cnt = (uint32_t)s->start[t][d + 1] - st0;
                const uint32_t k0 = ((c >> (kMaxLen - d)) - (st0 - s->delta[t][d])) & ((1u << d) - 1u);   // (first[d] = start[d] - delta[d])
                if (k0 < cnt) {
                    len = (uint32_t)d;
                    idx = (int)(st0 + k0 + (((cnt - 1u - k0) >> d) << d));
                    break;
                }
            }
            if (idx < 0) return -1;
        }
        if (len > br.bc && (int64_t)len > br.bits_left()) return -1;  // DecodingTree.swift:39 (bits_left() >= bc)
        br.consume(len);
        return (int)s->sym[t][idx];
    }

    // BZip2.swift:97-246.  Returns an swc_status; on success L/n/origPtr are complete.
    SWC_HD int run(uint32_t& orig_ptr) {
        if (br.bits_left() < 41) return SWC_E_BZIP2_WRONG_MAGIC;  // :103
        if (br.bits(1) != 0) return SWC_E_BZIP2_RANDOMIZED_BLOCK;  // :106
        orig_ptr = br.bits(24);
        const uint32_t used_map = br.bits(16);
        uint32_t pop = 0;
        for (int i = 0; i < 16; i++) pop += (used_map >> i) & 1u;
        if (br.bits_left() < (int64_t)(16 * pop + 3 + 15)) return SWC_E_BZIP2_WRONG_MAGIC;  // :118
        int n_used = 0;
        for (int blk = 0; blk < 16; blk++) {
            if (used_map & (0x8000u >> blk)) {
                const uint32_t m = br.bits(16);
                for (int k = 0; k < 16; k++)
                    if (m & (0x8000u >> k)) s->mtf[n_used++] = (uint8_t)(blk * 16 + k);
            }
        }
        const int used_count = 2 + n_used;  // :139
        const int n_tables = (int)br.bits(3);
        if (n_tables < 2 || n_tables > 6) return SWC_E_BZIP2_WRONG_HUFFMAN_GROUPS;  // :142
        const int n_selectors = (int)br.bits(15);
        for (int i = 0; i < n_tables; i++) s->tmtf[i] = (uint8_t)i;
        for (int i = 0; i < n_selectors; i++) {  // :155-173
            int c = 0;
            for (;;) {
                br.refill();
                if (br.bits_left() <= 0) break;
                const uint32_t b = br.peek(1);
                br.consume(1);
                if (b == 0) break;
                c++;
            }
            if (c >= n_tables) return SWC_E_BZIP2_WRONG_SELECTOR;
            const uint8_t el = s->tmtf[c];
            for (int k = c; k > 0; k--) s->tmtf[k] = s->tmtf[k - 1];
            s->tmtf[0] = el;
            if (lane == 0) ws.selectors[i] = el;
        }
        for (int t = 0; t < n_tables; t++) {  // :177-203
            if (br.bits_left() < 5) return SWC_E_BZIP2_WRONG_HUFFMAN_CODE_LENGTH;
            int length = (int)br.bits(5);
            for (int i = 0; i < used_count; i++) {
                if (!(length >= 0 && length <= kCheckedLen)) return SWC_E_BZIP2_WRONG_HUFFMAN_CODE_LENGTH;  // :185
                for (;;) {
                    br.refill();
                    if (br.bits_left() <= 0) break;
                    const uint32_t b = br.peek(1);
                    br.consume(1);
                    if (b == 0) break;
                    if (br.bits_left() <= 0) return SWC_E_BZIP2_WRONG_HUFFMAN_CODE_LENGTH;  // :193
                    br.refill();
                    const uint32_t d = br.peek(1);
                    br.consume(1);
                    length -= (int)d * 2 - 1;
                }
                // a non-final symbol outside 0...20 is rejected at the top of the next iteration (:185); the
                // FINAL one is never checked: <= 0 is skipped by Code.huffmanCodes, 21..26 is built, > 26 is trap-class
                s->lengths[i] = (int8_t)(length < -128 ? -128 : length > 127 ? 127 : length);
            }
            const int st = build_table(t, used_count);
            if (st) return st;
        }
        return symbol_loop(n_selectors, used_count, n_used);
    }

    // ---- symbol loop :205-246, state in registers (see the header comment) ---------------------------------------------
    // The loop is wave-uniform and bound by the NUMBER of instructions a symbol costs (round 3: 33 scalar + 31 vector), so it
    // is written per GROUP of 50 symbols (one selector): what can be decided once per group is -- enough input for 50 codes
    // of the maximal length (no end-of-input tests per symbol), room for 50 bytes in L, a prefix-free code (never anything
    // else from an encoder) -- and the groups that pass run the loop without those tests (phase<true>; the conditions only
    // get worse along a block, so the first group that fails hands the rest of the block to phase<false>).  Per symbol:
    //   decode   lane d holds limit, shift and index delta of code length d: ONE compare + ballot gives the length, every lane
    //            forms the index its length would give, one lane read picks it, one more reads the symbol (the 64 most
    //            frequent symbols -- the shortest codes -- sit one per lane);
    //   MTF      the list is kept one POSITION per lane (positions 0-63 in one register, the other 192 in three more): the
    //            element leaves through a lane read, the positions in front of it move up with one DPP shift of the whole
    //            wave and one select -- 5 vector instructions, nothing on the scalar unit; positions >= 64 take a general path;
    //   output   the staging register is a shift register over the lanes (one more DPP shift lets the byte in at lane 0); it is
    //            flushed at the end of the group (at most 50 bytes are staged: no "is it full" test per symbol) and in front
    //            of a run.
    static constexpr int N = 64;
    struct Loop {
        simt::PT<uint32_t, N> my_lim, my_delta, my_sh, sym_lo, sym_a, sym_b, sym_c;   // the active table
        simt::PT<uint32_t, N> sym_m;             // sym_lo as the assembly loop reads it: list position (symbol - 1); RUNA / RUNB: 2^31 | symbol
        simt::PT<uint32_t, N> l0, l1, l2, l3;    // the list: position 64 i + t in l<i>[t]
        simt::PT<uint32_t, N> stg;               // L[sbase .. sbase + k) staged: lane j holds byte sbase + j
        uint32_t sbase, k;
        uint64_t run_length, repeat_power;       // repeat_power != 1: RUNA / RUNB symbols since the last byte symbol
        int n_selectors, used_count, n_used, selector_index, table;
        uint32_t mfast;                          // list positions the short way takes (the end-of-block symbol is n_used + 1)
        bool over, have_table;
    };
    SWC_HD void load_table(Loop& L, int tb) {   // lane d: limit, shift and index delta of code length d; lane t: symbols t, t + 64, ... of the sorted array
        SIMT_BEGIN(t, N)
            const bool len_lane = t >= 1 && t <= kMaxLen;
            // lane 0 counts always (the number of limits <= the window IS the length), lane kMaxLen + 1 stands for "no
            // code": an index no table has
            L.my_lim[t] = len_lane ? s->lim[tb][t] : t == 0 ? 0u : 0xFFFFFFFFu;
            L.my_delta[t] = len_lane ? s->delta[tb][t] : 0x80000000u;
            L.my_sh[t] = len_lane ? (uint32_t)(kMaxLen - t) : 0u;
            const uint16_t* sy = s->sym[tb];
            L.sym_lo[t] = (uint32_t)sy[t];
            L.sym_m[t] = sy[t] < 2u ? 0x80000000u | (uint32_t)sy[t] : (uint32_t)sy[t] - 1u;
            L.sym_a[t] = (uint32_t)sy[t] | ((uint32_t)sy[t + 64] << 16);
            L.sym_b[t] = (uint32_t)sy[t + 128] | ((uint32_t)sy[t + 192] << 16);
            L.sym_c[t] = t < kMaxSyms + 2 - 256 ? (uint32_t)sy[256 + t] : 0u;
        SIMT_END
    }
    // one more staged byte (the C++ paths; the assembly loop: ONE v_writelane with the count, kept in M0, as the lane select)
    SWC_HD void stage_byte(Loop& L, uint32_t el) {
        const uint32_t k = L.k;
        SIMT_BEGIN(t, N) L.stg[t] = (uint32_t)t == k ? el : L.stg[t]; SIMT_END
        L.k = k + 1u;
    }
    SWC_HD void flush(Loop& L) {
        // (a fast group tests the room in L once, at its end: bytes staged beyond it are dropped here, never stored)
        SIMT_BEGIN(t, N)
            const uint32_t at = L.sbase + (uint32_t)t;
            if ((uint32_t)t < L.k && at < lcap32) ws.L[at] = (uint8_t)L.stg[t];
        SIMT_END
        // (said explicitly: the compiler merges these updates into the per-lane region above and then treats the counters --
        // and every branch on them -- as different from lane to lane)
        L.sbase = simt::uniform(L.sbase + L.k);
        L.k = 0;
    }
    // usedSymbols.remove(at: m) + insert(at: 0)  (BZip2.swift:243-244) for any m: positions 0 .. m - 1 move up by one
    SWC_HD uint32_t mtf_general(Loop& L, uint32_t m) {
        simt::PT<uint32_t, N> sh0, sh1, sh2, sh3;
        const uint32_t q = m >> 6, r = m & 63u;
        // (reads first, then a scalar select: a select between the REGISTERS would become a computed address into the
        // struct, which then stays in memory with everything in it)
        const uint32_t e0 = simt::wave_read<N>(L.l0, (int)r), e1 = simt::wave_read<N>(L.l1, (int)r), e2 = simt::wave_read<N>(L.l2, (int)r),
                       e3 = simt::wave_read<N>(L.l3, (int)r);
        const uint32_t el = q == 0u ? e0 : q == 1u ? e1 : q == 2u ? e2 : e3;
        const uint32_t c0 = simt::wave_read<N>(L.l0, 63), c1 = simt::wave_read<N>(L.l1, 63), c2 = simt::wave_read<N>(L.l2, 63);
        simt::wave_shift_up_dpp<N>(sh0, L.l0, el);
        simt::wave_shift_up_dpp<N>(sh1, L.l1, c0);
        simt::wave_shift_up_dpp<N>(sh2, L.l2, c1);
        simt::wave_shift_up_dpp<N>(sh3, L.l3, c2);
        SIMT_BEGIN(t, N)
            const bool in = (uint32_t)t <= r;
            L.l0[t] = (q > 0u || in) ? sh0[t] : L.l0[t];
            if (q >= 1u) L.l1[t] = (q > 1u || in) ? sh1[t] : L.l1[t];
            if (q >= 2u) L.l2[t] = (q > 2u || in) ? sh2[t] : L.l2[t];
            if (q >= 3u) L.l3[t] = in ? sh3[t] : L.l3[t];
        SIMT_END
        return el;
    }
    // The symbols of a group that need nothing special, one after the other (FAST groups only): RUNA / RUNB, a run short enough
    // for the staging register, a byte symbol whose code is among the 64 shortest and whose list position is below 64.
    // Anything else ends the loop with the symbol (kSymbol) or its table index (kIndex) decoded and counted but not applied.
    // ONE loop with ONE kind of exit and no stores: with the rare cases inside, the copies the compiler places at their merge
    // points cost more than the work (a dozen moves per symbol).
    // Staging: a group starts with an empty register; a byte symbol adds one byte, a run enters only if it leaves a lane for
    // every symbol the group still has -- so "is the register full" is never asked per symbol.
    enum { kGroupDone = 0, kSymbol = 1, kIndex = 2 };
    SWC_HD int hot_symbols(Loop& L, int& i, uint32_t& pending, uint32_t& pending_len) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SWC_BZ_HOT_CXX)
        if (!CXX) return hot_symbols_isa(L, i, pending, pending_len);
#endif
        return hot_symbols_cxx(L, i, pending, pending_len);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // The same loop as hot_symbols_cxx below, written in gfx950 assembly: 27 instructions per byte symbol and 25 per RUNA /
    // RUNB against the 58 / 41 the compiler makes of the C++ (it keeps the loop-carried values in different registers on
    // different paths and copies them at every merge point, and it will not shift a register in place).  All scalar state
    // lives in fixed registers inside the block (s80 - s99: the halves of the 64-bit values are needed on their own, and an
    // operand has no syntax for that); it is handed over at entry and exit.  What differs from the C++ in form only:
    //   * the list: `v_mov_b32_dpp` by one lane under a compare, lane 0 written afterwards with v_writelane; the staged bytes: ONE
    //     v_writelane per byte with the count as the lane select -- in M0, because a v_writelane whose value AND lane select are
    //     SGPRs violates the constant-bus rule of gfx9, and M0 as the lane select does not; a run: lanes k .. k + run - 1 under a compare;
    //   * the bit window is refilled inside too, by SCALAR loads (s_load_dwordx2 of the aligned pair that holds the next dword,
    //     asked for one refill ahead; the input is read-only for the kernel, and a FAST group has its 163 bytes): the word
    //     arrives in scalar registers, and the loop is left only at the end of a group or for a symbol it does not handle.
    // Wait states the assembler does not insert into inline code: a lane select that was written by v_readlane needs four
    // (the symbol's bookkeeping stands between the two lane reads); DPP sources are written at least two instructions earlier on every path.
    SWC_D int hot_symbols_isa(Loop& L, int& i, uint32_t& pending, uint32_t& pending_len) {
        const uint32_t lane = (uint32_t)threadIdx.x;
        uint32_t ev, pend, plen, t0, t1;
        // (every scalar operand is said to be wave-uniform once more: a value the compiler holds in a vector register cannot be
        // bound to an "s" operand)
        uint32_t ii = simt::uniform((uint32_t)i), bc = simt::uniform(br.bc), k = simt::uniform(L.k);
        uint64_t bb = simt::uniform(br.bb), rl = simt::uniform(L.run_length), rp = simt::uniform(L.repeat_power);
        const uint32_t mfast = simt::uniform(L.mfast);
        // the window is refilled inside (FAST groups have their input: phase<true>): the address of the next dword to enter it
        uint64_t addr = simt::uniform((uint64_t)(uintptr_t)br.in + br.next);
        br.pw_ok = false;
        {
            asm volatile(
                "s_mov_b64 s[80:81], %[bb]\n\t"
                "s_mov_b32 s82, %[bc]\n\t"
                "s_mov_b32 s83, %[i]\n\t"
                "s_mov_b32 m0, %[k]\n\t"                      // (the count of staged bytes lives in M0: see Lbyte)
                "s_mov_b64 s[86:87], %[rl]\n\t"
                "s_mov_b64 s[88:89], %[rp]\n\t"
                "s_mov_b32 s90, %[mfast]\n\t"
                "s_mov_b64 s[78:79], %[addr]\n\t"
                "s_mov_b32 s98, 0\n\t"
                "s_and_b32 s76, s78, -4\n\t"                 // the aligned pair that holds the dword at the read position, on its way
                "s_mov_b32 s77, s79\n\t"
                "s_load_dwordx2 s[74:75], s[76:77], 0x0\n"
                "Ltop%=:\n\t"
                "s_cmp_le_u32 s82, 32\n\t"
                "s_cbranch_scc1 Lrefill%=\n"
                "Lgo%=:\n\t"
                "s_lshr_b32 s91, s81, 6\n\t"
                "v_cmp_ge_u32_e32 vcc, s91, %[lim]\n\t"
                "v_lshrrev_b32_e64 %[t0], %[sh], s91\n\t"
                "v_add_u32_e32 %[t0], %[t0], %[delta]\n\t"
                "s_bcnt1_i32_b64 s92, vcc\n\t"
                "s_lshl_b64 s[80:81], s[80:81], s92\n\t"
                "v_readlane_b32 s93, %[t0], s92\n\t"
                "s_sub_u32 s82, s82, s92\n\t"                // (the bookkeeping of the symbol IS the four wait states of the lane select)
                "s_add_u32 s83, s83, 1\n\t"
                "s_cmp_gt_u32 s93, 63\n\t"
                "s_cbranch_scc1 Lindex%=\n\t"
                "v_readlane_b32 s94, %[symlo], s93\n\t"      // the list position of a byte symbol; RUNA / RUNB: 2^31 | symbol
                "s_cmp_ge_u32 s94, s90\n\t"                  // ONE test for everything but a byte symbol of the short way
                "s_cbranch_scc1 Lnotfast%=\n\t"
                "s_cmp_lg_u64 s[88:89], 1\n\t"
                "s_cbranch_scc1 Lpend%=\n"
                "Lbyte%=:\n\t"
                "v_readlane_b32 s95, %[l0], s94\n\t"
                "v_cmp_ge_u32_e32 vcc, s94, %[lane]\n\t"
                "v_mov_b32_dpp %[t0], %[l0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_cndmask_b32_e32 %[l0], %[l0], %[t0], vcc\n\t"
                "v_writelane_b32 %[l0], s95, 0\n\t"
                "v_writelane_b32 %[stg], s95, m0\n\t"         // staged: lane k.  Value AND lane select in scalar registers is one too many
                "s_add_u32 m0, m0, 1\n"                       // for the constant bus of gfx9 -- unless the lane select is M0
                "Lnext%=:\n\t"
                "s_cmp_lt_u32 s83, 50\n\t"
                "s_cbranch_scc1 Ltop%=\n\t"
                "s_branch Ldone%=\n"
                "Lnotfast%=:\n\t"
                "s_cmp_lt_i32 s94, 0\n\t"
                "s_cbranch_scc0 Lsym%=\n\t"
                "s_lshl_b64 s[96:97], s[88:89], s94\n\t"     // RUNA / RUNB (a 64-bit shift takes the low six bits of its count: 0 / 1)
                "s_add_u32 s86, s86, s96\n\t"
                "s_addc_u32 s87, s87, s97\n\t"
                "s_lshl_b64 s[88:89], s[88:89], 1\n\t"
                "s_cmp_lt_u32 s83, 50\n\t"                   // (every path closes the loop itself: a taken branch costs the wave its
                "s_cbranch_scc1 Ltop%=\n\t"                  // instruction buffer, and a run digit took three of them per symbol)
                "s_branch Ldone%=\n"
                "Lpend%=:\n\t"
                "s_add_u32 s96, s86, -1\n\t"
                "s_addc_u32 s97, s87, -1\n\t"
                "s_sub_u32 s99, s83, m0\n\t"
                "s_add_u32 s99, s99, 13\n\t"
                "s_cmp_lg_u32 s97, 0\n\t"
                "s_cbranch_scc1 Lsym0%=\n\t"
                "s_cmp_ge_u32 s96, s99\n\t"
                "s_cbranch_scc1 Lsym0%=\n\t"
                "v_readlane_b32 s95, %[l0], 0\n\t"
                "v_subrev_u32_e32 %[t0], m0, %[lane]\n\t"     // lane - k (huge below k)
                "s_and_b32 s95, s95, 0xff\n\t"
                "v_cmp_gt_u32_e32 vcc, s86, %[t0]\n\t"        // lanes k .. k + run - 1 take the run's byte
                "v_mov_b32_e32 %[t1], s95\n\t"
                "v_cndmask_b32_e32 %[stg], %[stg], %[t1], vcc\n\t"
                "s_add_u32 m0, m0, s86\n\t"
                "s_mov_b64 s[86:87], 0\n\t"
                "s_mov_b64 s[88:89], 1\n\t"
                "v_readlane_b32 s95, %[l0], s94\n\t"        // (Lbyte once more, instead of a branch there and one back)
                "v_cmp_ge_u32_e32 vcc, s94, %[lane]\n\t"
                "v_mov_b32_dpp %[t0], %[l0] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_cndmask_b32_e32 %[l0], %[l0], %[t0], vcc\n\t"
                "v_writelane_b32 %[l0], s95, 0\n\t"
                "v_writelane_b32 %[stg], s95, m0\n\t"
                "s_add_u32 m0, m0, 1\n\t"
                "s_cmp_lt_u32 s83, 50\n\t"
                "s_cbranch_scc1 Ltop%=\n\t"
                "s_branch Ldone%=\n"
                "Lsym0%=:\n\t"
                "s_add_u32 s93, s94, 1\n\t"
                "s_mov_b32 s98, 1\n\t"
                "s_branch Ldone%=\n"
                "Lsym%=:\n\t"
                "s_add_u32 s93, s94, 1\n\t"
                "s_mov_b32 s98, 1\n\t"
                "s_branch Ldone%=\n"
                "Lindex%=:\n\t"
                "s_mov_b32 s98, 2\n\t"
                "s_branch Ldone%=\n"
                "Lrefill%=:\n\t"                              // 32 more bits into the window; the pair behind them is asked for at once
                "s_and_b32 s99, s78, 3\n\t"
                "s_lshl_b32 s99, s99, 3\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "s_lshr_b64 s[96:97], s[74:75], s99\n\t"
                "s_add_u32 s78, s78, 4\n\t"
                "s_addc_u32 s79, s79, 0\n\t"
                "s_and_b32 s76, s78, -4\n\t"
                "s_mov_b32 s77, s79\n\t"
                "s_load_dwordx2 s[74:75], s[76:77], 0x0\n\t"
                "v_mov_b32_e32 %[t1], s96\n\t"               // the dword's bytes in stream order: there is no scalar byte swap, the vector
                "s_mov_b32 s97, 0x00010203\n\t"              // unit permutes bytes in one instruction (three here against nine scalar ones)
                "v_perm_b32 %[t1], %[t1], %[t1], s97\n\t"
                "s_nop 0\n\t"
                "v_readfirstlane_b32 s96, %[t1]\n\t"
                "s_mov_b32 s97, 0\n\t"
                "s_sub_u32 s99, 32, s82\n\t"
                "s_lshl_b64 s[96:97], s[96:97], s99\n\t"
                "s_or_b64 s[80:81], s[80:81], s[96:97]\n\t"
                "s_add_u32 s82, s82, 32\n\t"
                "s_branch Lgo%=\n"
                "Ldone%=:\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"                    // (the pair on its way must have landed before its registers are anybody else's)
                "s_mov_b64 %[addr], s[78:79]\n\t"
                "s_mov_b64 %[bb], s[80:81]\n\t"
                "s_mov_b32 %[bc], s82\n\t"
                "s_mov_b32 %[i], s83\n\t"
                "s_mov_b32 %[k], m0\n\t"
                "s_mov_b64 %[rl], s[86:87]\n\t"
                "s_mov_b64 %[rp], s[88:89]\n\t"
                "s_mov_b32 %[ev], s98\n\t"
                "s_mov_b32 %[pend], s93\n\t"
                "s_mov_b32 %[plen], s92"
                : [bb] "+s"(bb), [bc] "+s"(bc), [i] "+s"(ii), [k] "+s"(k), [rl] "+s"(rl), [rp] "+s"(rp), [addr] "+s"(addr),
                  [ev] "=s"(ev), [pend] "=s"(pend), [plen] "=s"(plen), [l0] "+v"(L.l0.v), [stg] "+v"(L.stg.v), [t0] "=&v"(t0), [t1] "=&v"(t1)
                : [mfast] "s"(mfast), [lim] "v"(L.my_lim.v), [sh] "v"(L.my_sh.v), [delta] "v"(L.my_delta.v), [symlo] "v"(L.sym_m.v), [lane] "v"(lane)
                : "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97",
                  "s98", "s99", "vcc", "scc", "m0");
        }
        br.bb = bb; br.bc = bc;
        br.next = addr - (uint64_t)(uintptr_t)br.in;
        L.k = k; L.run_length = rl; L.repeat_power = rp;
        i = (int)ii;
        pending = pend;
        pending_len = plen;
        return (int)ev;
    }
#endif
    SWC_HD int hot_symbols_cxx(Loop& L, int& i, uint32_t& pending, uint32_t& pending_len) {
        simt::PT<uint32_t, N> idxv, sh;
        simt::PT<bool, N> pb;
        // ONE exit test at the bottom and one way round: the backend gives a loop with several exits a guard variable and
        // a block of moves per exit (UnifyLoopExits), and a `continue` a second latch with its own copies.  `ev` is made opaque
        // in front of the test so that the exits are not threaded back into the paths that set it.
        uint32_t ev = kGroupDone;
        do {
            br.refill();
            const uint32_t c = br.peek(kMaxLen);
            SIMT_BEGIN(t, N)
                pb[t] = c >= L.my_lim[t];
                idxv[t] = (c >> L.my_sh[t]) + L.my_delta[t];
            SIMT_END
            const uint32_t len = (uint32_t)simt::popc64(simt::wave_ballot<N>(pb));
            const uint32_t idx = simt::wave_read<N>(idxv, (int)len);
            br.consume(len);
            i++;
            if (idx < 64u) {
                const uint32_t symbol = simt::wave_read<N>(L.sym_lo, (int)idx);
                if (symbol < 2u) {  // RUNA / RUNB :226-230 (wrapping, like &+ and smart shifts)
                    L.run_length += L.repeat_power << symbol;
                    L.repeat_power <<= 1;
                } else {
                    // k + run + (the 51 - i symbols from this one on) <= 64; a run that is not positive (wrapped) goes the long way
                    if (L.repeat_power != 1ull) {
                        if (L.run_length - 1ull < (uint64_t)(13u + (uint32_t)i - L.k)) {
                            const uint32_t run = (uint32_t)L.run_length;
                            const uint32_t b = simt::wave_read<N>(L.l0, 0) & 255u;
                            const uint32_t k0 = L.k;
                            SIMT_BEGIN(t, N) L.stg[t] = (uint32_t)t - k0 < run ? b : L.stg[t]; SIMT_END   // lanes k .. k + run - 1
                            L.k += run;
                            L.run_length = 0;
                            L.repeat_power = 1;
                        }
                    }
                    const uint32_t m = symbol - 1u;
                    if (L.repeat_power == 1ull && m < L.mfast) {
                        const uint32_t el = simt::wave_read<N>(L.l0, (int)m);
                        simt::wave_shift_up_dpp<N>(sh, L.l0, el);
                        SIMT_BEGIN(t, N) L.l0[t] = (uint32_t)t <= m ? sh[t] : L.l0[t]; SIMT_END
                        stage_byte(L, el);
                    } else { pending = symbol; ev = kSymbol; }
                }
            } else { pending = idx; pending_len = len; ev = kIndex; }
            SWC_OPAQUE_S(ev);
        } while (ev == kGroupDone && i < 50);
        return (int)ev;
    }
    // Groups until the end-of-block symbol (returns SWC_OK with everything flushed), an error (its status) or -- FAST only --
    // a group that does not qualify (returns -1 with that group's table loaded).
    template <bool FAST>
    SWC_HD int phase(Loop& L) {
        simt::PT<uint32_t, N> idxv, sh;
        simt::PT<bool, N> pb;
        for (;;) {
            if (!L.have_table) {
                if (!(L.selector_index < L.n_selectors)) return SWC_E_BZIP2_WRONG_SELECTOR;  // :214
                L.table = selector_at(L.selector_index++);
                load_table(L, L.table);
                L.over = s->oversub[L.table] != 0;
            }
            L.have_table = false;
            if (FAST) {
                // 50 codes of kMaxLen bits are 163 bytes; the reader looks 8 bytes ahead and the assembly loop asks for one more aligned
                // pair beyond that (up to 11 bytes): 184 (it was 176).  (n_used == 0: a run has no byte to repeat.)
                if (L.over || L.n_used == 0 || br.next + 184 > br.n || (uint64_t)L.sbase + 128 > (uint64_t)lcap32) { L.have_table = true; return -1; }
            }
            int i = 0;
            for (;;) {
                uint32_t symbol;
                if (FAST) {
                    uint32_t pend = 0, plen = 0;
                    if (i >= 50) break;
                    const int ev = hot_symbols(L, i, pend, plen);
                    if (ev == kGroupDone) break;
                    if (ev == kIndex) {
                        if (plen > (uint32_t)kMaxLen) return SWC_E_BZIP2_SYMBOL_NOT_FOUND;  // :222
                        const uint32_t wa = simt::wave_read<N>(L.sym_a, (int)(pend & 63u)), wb = simt::wave_read<N>(L.sym_b, (int)(pend & 63u)),
                                       wc = simt::wave_read<N>(L.sym_c, (int)(pend & 63u));
                        const uint32_t w = pend < 128u ? wa : pend < 256u ? wb : wc;
                        symbol = ((pend & 64u) != 0u && pend < 256u) ? w >> 16 : w & 0xFFFFu;
                    } else symbol = pend;
                } else {
                    if (i >= 50) break;
                    i++;
                    if (!L.over) {   // DecodingTree.findNextSymbol for a prefix-free set
                        br.refill();
                        const uint32_t c = br.peek(kMaxLen);
                        SIMT_BEGIN(t, N)
                            pb[t] = c >= L.my_lim[t];
                            idxv[t] = (c >> L.my_sh[t]) + L.my_delta[t];
                        SIMT_END
                        const uint32_t len = (uint32_t)simt::popc64(simt::wave_ballot<N>(pb));
                        if (len > (uint32_t)kMaxLen) return SWC_E_BZIP2_SYMBOL_NOT_FOUND;  // :222
                        const uint32_t idx = simt::wave_read<N>(idxv, (int)len);
                        if (len > br.bc && (int64_t)len > br.bits_left()) return SWC_E_BZIP2_SYMBOL_NOT_FOUND;  // DecodingTree.swift:39
                        br.consume(len);
                        const uint32_t wa = simt::wave_read<N>(L.sym_a, (int)(idx & 63u)), wb = simt::wave_read<N>(L.sym_b, (int)(idx & 63u)),
                                       wc = simt::wave_read<N>(L.sym_c, (int)(idx & 63u));
                        const uint32_t w = idx < 128u ? wa : idx < 256u ? wb : wc;
                        symbol = ((idx & 64u) != 0u && idx < 256u) ? w >> 16 : w & 0xFFFFu;
                    } else {       // over-subscribed set (never written by an encoder): the heap semantics, from LDS
                        const int sy = decode_symbol(L.table, 0u, true);
                        if (sy == -1) return SWC_E_BZIP2_SYMBOL_NOT_FOUND;  // :222
                        symbol = (uint32_t)sy;
                    }
                }
                // ---- one symbol the long way
                if (symbol < 2u) {  // RUNA / RUNB :226-230 (wrapping, like &+ and smart shifts)
                    L.run_length += L.repeat_power << symbol;
                    L.repeat_power <<= 1;
                    continue;
                }
                if (L.repeat_power != 1ull) {   // (a run that wrapped to something not positive is tested again at every byte symbol: same outcome)
                    if ((int64_t)L.run_length > 0) {
                        if (L.n_used == 0) return SWC_E_REF_TRAP;  // usedSymbols[0] on an empty array
                        if ((uint64_t)L.sbase + L.k + L.run_length > ws.lcap || L.run_length > 0xFFFFFFFFull) return SWC_E_NEED_WORKSPACE;
                        flush(L);
                        const uint32_t b = simt::wave_read<N>(L.l0, 0) & 255u;
                        const uint32_t run = (uint32_t)L.run_length;
                        SIMT_BEGIN(t, N) for (uint32_t j = (uint32_t)t; j < run; j += (uint32_t)N) ws.L[L.sbase + j] = (uint8_t)b; SIMT_END
                        L.sbase = simt::uniform(L.sbase + run);
                        L.run_length = 0;
                        L.repeat_power = 1;
                    }
                }
                if (symbol == (uint32_t)(L.used_count - 1)) {  // :239 end of block
                    if ((uint64_t)L.sbase + L.k > (uint64_t)lcap32) return SWC_E_NEED_WORKSPACE;
                    flush(L);
                    return SWC_OK;
                }
                const uint32_t el = mtf_general(L, symbol - 1u);
                // L[n_out] = el, staged a byte per lane
                if (!FAST && L.sbase + L.k >= lcap32) return SWC_E_NEED_WORKSPACE;
                stage_byte(L, el);
            }
            // FAST: runs staged inside the group may have used up the room in L.  (The byte that does not fit is reported here,
            // not where it was staged: an error in the rest of the group comes first -- only for a column that outgrows its
            // workspace, which run_units answers with a larger one.)
            if ((uint64_t)L.sbase + L.k > (uint64_t)lcap32) return SWC_E_NEED_WORKSPACE;
            flush(L);
        }
    }
    SWC_HD int symbol_loop(int n_selectors, int used_count, int n_used) {
        if (n_selectors == 0) return SWC_E_REF_TRAP;  // selectors[0] on an empty array (App. A B3)
        Loop L;
        SIMT_BEGIN(t, N)
            L.l0[t] = s->mtf[t]; L.l1[t] = s->mtf[64 + t]; L.l2[t] = s->mtf[128 + t]; L.l3[t] = s->mtf[192 + t];
            L.stg[t] = 0;
        SIMT_END
        L.sbase = 0; L.k = 0;
        L.run_length = 0; L.repeat_power = 1;
        L.n_selectors = n_selectors; L.used_count = used_count; L.n_used = n_used;
        L.selector_index = 0; L.table = 0;
        L.mfast = (uint32_t)(n_used < 64 ? n_used : 64);
        L.over = false; L.have_table = false;
        int st = phase<true>(L);
        if (st == -1) st = phase<false>(L);
        n_out = L.sbase + L.k;
        return st;
    }
};

// Stage 1 entry.  job.dict_len = bit offset of the block body (just past the 48-bit magic and the CRC).
template <int WAVE, bool CXX = false>
SWC_HD void stage1_job(const Job& job, Stage1Lds* lds, Workspace ws, int lane) {
    Stage1<WAVE, CXX> d;
    d.s = lds;
    d.ws = ws;
    d.lane = lane;
    d.n_out = 0;
    d.ws.lcap = (size_t)simt::uniform((uint64_t)ws.lcap);
    d.lcap32 = d.ws.lcap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)d.ws.lcap;
    uint32_t orig_ptr = 0;
    int st;
    if (job.dict_len > job.in_len * 8) {
        st = SWC_E_INVALID_ARGUMENT;
        d.br.init((gcptr)job.in, 0, 0);
    } else {
        d.br.init((gcptr)job.in, job.in_len, job.dict_len);
        st = d.run(orig_ptr);
    }
    if (lane == 0) {
        ws.hdr->n = d.n_out;
        ws.hdr->orig_ptr = orig_ptr;
        ws.hdr->status = (uint32_t)st;
        ws.hdr->end_bit = d.br.position();
    }
}

// Stage 2: P[base[c]++] = i << 8 | c in increasing i (BurrowsWheeler.swift:38-53) -- a STABLE counting sort by the 64
// lanes of a wave.  L is read 64 consecutive bytes at a time (one coalesced load); inside such a group a lane finds the
// lanes that hold the same byte value with eight ballots (one per bit of the value), its rank among them (the order of
// i is the order of the lanes) and their number; the first lane of every value adds the number to the value's counter.
// One pass counts, a wave scan turns the 256 counters into start positions, a second pass scatters.  `cnt`: 256 words
// of LDS (the earlier form -- a private slice of L and 256 private counters per lane -- needed 64 KiB and read L with
// 64 different lines per load).
struct Match {   // of one group of 64 bytes
    simt::PT<uint32_t, 64> rank, count;
};
SWC_HD void match_bytes(const simt::PT<uint32_t, 64>& sym, const simt::PT<bool, 64>& act, Match& m) {
    using simt::PT;
    constexpr int N = 64;
    PT<bool, N> pb;
    uint64_t bal[8];
    SIMT_BEGIN(t, N) pb[t] = act[t]; SIMT_END
    const uint64_t m_act = simt::wave_ballot<N>(pb);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        SIMT_BEGIN(t, N) pb[t] = act[t] && ((sym[t] >> b) & 1u) != 0u; SIMT_END
        bal[b] = simt::wave_ballot<N>(pb);
    }
    SIMT_BEGIN(t, N)
        uint64_t same = m_act;
#pragma unroll
        for (int b = 0; b < 8; b++) same &= ((sym[t] >> b) & 1u) ? bal[b] : ~bal[b];
        m.rank[t] = (uint32_t)simt::popc64(same & ((1ull << t) - 1ull));
        m.count[t] = (uint32_t)simt::popc64(same);
    SIMT_END
}
SWC_HD void stage2_job(Workspace ws, uint32_t* cnt) {
    using simt::PT;
    constexpr int N = 64;
    if (ws.hdr->status != SWC_OK) return;
    const uint32_t n = ws.hdr->n;
    if (n == 0) return;
    PT<uint32_t, N> sym, nsym, s4;
    PT<bool, N> act;
    Match m;
    SIMT_BEGIN(t, N)
        for (int c = t; c < 256; c += N) cnt[c] = 0;
        nsym[t] = (uint32_t)t < n ? (uint32_t)ws.L[t] : 0u;
    SIMT_END_WAVE
    // ---- count
    for (uint32_t i0 = 0; i0 < n; i0 += N) {
        SIMT_BEGIN(t, N)
            const uint32_t i = i0 + (uint32_t)t;
            act[t] = i < n;
            sym[t] = nsym[t];
            nsym[t] = i + N < n ? (uint32_t)ws.L[i + N] : 0u;   // the next group is on its way while this one is ranked
        SIMT_END
        match_bytes(sym, act, m);
        SIMT_BEGIN(t, N) if (act[t] && m.rank[t] == 0u) cnt[sym[t]] += m.count[t]; SIMT_END_WAVE
    }
    // ---- counters -> start positions (exclusive prefix sum over the byte values; lane t owns the values 4t .. 4t + 3)
    SIMT_BEGIN(t, N) s4[t] = cnt[4 * t] + cnt[4 * t + 1] + cnt[4 * t + 2] + cnt[4 * t + 3]; SIMT_END
    simt::wave_scan_incl<N>(s4);
    SIMT_BEGIN(t, N)
        uint32_t base = s4[t] - (cnt[4 * t] + cnt[4 * t + 1] + cnt[4 * t + 2] + cnt[4 * t + 3]);
        for (int k = 0; k < 4; k++) { const uint32_t c = cnt[4 * t + k]; cnt[4 * t + k] = base; base += c; }
        nsym[t] = (uint32_t)t < n ? (uint32_t)ws.L[t] : 0u;
    SIMT_END_WAVE
    // ---- scatter
    for (uint32_t i0 = 0; i0 < n; i0 += N) {
        SIMT_BEGIN(t, N)
            const uint32_t i = i0 + (uint32_t)t;
            act[t] = i < n;
            sym[t] = nsym[t];
            nsym[t] = i + N < n ? (uint32_t)ws.L[i + N] : 0u;
        SIMT_END
        match_bytes(sym, act, m);
        SIMT_BEGIN(t, N)
            if (act[t]) ws.P[cnt[sym[t]] + m.rank[t]] = ((i0 + (uint32_t)t) << 8) | sym[t];
        SIMT_END_WAVE
        SIMT_BEGIN(t, N) if (act[t] && m.rank[t] == 0u) cnt[sym[t]] += m.count[t]; SIMT_END_WAVE
    }
}

// ---- stage 3a: cut the cycle, walk the pieces, order them, lay the bytes out in L -------------------------------------
struct Stage3Lds {
    uint32_t len[kSegs + 1];    // bytes of segment s
    uint32_t nxoff[kSegs + 1];  // first the segment that starts where s ends, then (once the cycle order has passed s) the
                                // position of s in the block: one array, 2 KB less LDS = 29 instead of 23 waves per CU
    uint32_t ticket;            // next segment to hand out
    uint32_t bad;
    uint32_t part_at[kParts + 1];   // RLE1 undo: part j covers T[part_at[j] .. part_at[j + 1])
    uint64_t part_out[kParts + 1];  // ... and produces out[part_out[j] .. part_out[j + 1])
};
constexpr int kStage3LdsBytes = (sizeof(Stage3Lds) + 15) / 16 * 16;

// RLE1 undo (BZip2.swift:251-267) as a state machine over the walked bytes T: `run` = equal literals so far (1..4), 0 right
// after a count byte.  Feed one byte, get the number of output bytes it stands for (1 for a literal, its value for a count).
struct Rle1 {
    uint32_t run, prev;
    SWC_HD uint32_t feed(uint32_t b, bool& is_count) {
        if (run == 4) { run = 0; is_count = true; return b; }
        is_count = false;
        run = (run > 0 && b == prev) ? run + 1 : 1;
        prev = b;
        return 1;
    }
};
// Where may a part start?  At p with T[p-3] != T[p-2] != T[p-1]: whatever T[p-2] was (literal or count), T[p-1] is then a
// literal that starts a run, so the state in front of p is (run 1, prev T[p-1]) -- no history needed.
SWC_HD uint32_t first_safe_start(gcptr T, uint32_t n, uint32_t from) {
    uint32_t p = from < 3 ? 3 : from;
    for (; p < n; p++)
        if (T[p - 3] != T[p - 2] && T[p - 2] != T[p - 1]) return p;
    return n;
}

// T[lo .. hi) eight bytes per load (a byte per load is a memory round trip per byte and lane).
template <typename F>
SWC_HD void for_bytes(gcptr T, uint32_t lo, uint32_t hi, F f) {
    uint32_t i = lo;
    for (; i < hi && (i & 7u); i++) f((uint32_t)T[i]);
    // 32 bytes per round trip: the four loads are in flight together (every lane streams through its own part of T, so a step
    // is a memory latency; with one load per step the RLE1 undo of a launch was 43 ms of nothing but waiting)
    // (64 bytes per step, eight loads in flight: 50 ms for the finish of a launch against 54 with 32, 92 with 8)
    for (; i + 64 <= hi; i += 64) {
        uint64_t w[8];
#pragma unroll
        for (int q = 0; q < 8; q++) w[q] = load_u64(T + i + 8 * q);
#pragma unroll
        for (int q = 0; q < 8; q++) {
#pragma unroll
            for (int k = 0; k < 8; k++) f((uint32_t)(w[q] >> (8 * k)) & 0xFFu);
        }
    }
    for (; i + 32 <= hi; i += 32) {
        const uint64_t w0 = load_u64(T + i), w1 = load_u64(T + i + 8), w2 = load_u64(T + i + 16), w3 = load_u64(T + i + 24);
#pragma unroll
        for (int k = 0; k < 8; k++) f((uint32_t)(w0 >> (8 * k)) & 0xFFu);
#pragma unroll
        for (int k = 0; k < 8; k++) f((uint32_t)(w1 >> (8 * k)) & 0xFFu);
#pragma unroll
        for (int k = 0; k < 8; k++) f((uint32_t)(w2 >> (8 * k)) & 0xFFu);
#pragma unroll
        for (int k = 0; k < 8; k++) f((uint32_t)(w3 >> (8 * k)) & 0xFFu);
    }
    for (; i + 8 <= hi; i += 8) {
        const uint64_t w = load_u64(T + i);
#pragma unroll
        for (int k = 0; k < 8; k++) f((uint32_t)(w >> (8 * k)) & 0xFFu);
    }
    for (; i < hi; i++) f((uint32_t)T[i]);
}
// Output bytes out[pos ..) of ONE writer, eight per store once `pos` reaches a multiple of eight (the bytes below it in
// that group may belong to another writer); nothing is stored at or beyond `cap`, `pos` keeps counting.  (Sixteen per store
// -- half the partial-line writes -- was measured slower: the second 64-bit accumulator and its selects cost more than the
// stores save, profiles/r05_experiments.txt.)
struct OutPack {
    gptr out;
    uint64_t cap, pos, buf, gstart;   // gstart: first position this writer handles in groups
    bool grouped;
    SWC_HD void begin(gptr o, uint64_t c, uint64_t p) {
        out = o; cap = c; pos = p; buf = 0;
        gstart = (p + 7) & ~(uint64_t)7;
        grouped = p == gstart;
    }
    SWC_HD void put(uint32_t b) {
        if (pos < cap) {
            if (!grouped) {
                out[pos] = (uint8_t)b;
                grouped = (pos & 7u) == 7u;
            } else {
                buf |= (uint64_t)b << (8 * (pos & 7u));
                if ((pos & 7u) == 7u) { store_u64(out + (pos - 7), buf); buf = 0; }
            }
        }
        pos++;
    }
    SWC_HD void finish() {   // the bytes of the last, incomplete group -- if this writer put any below `cap`
        const uint64_t done = pos < cap ? pos : cap;
        uint64_t j = done & ~(uint64_t)7;
        if (j < gstart) j = gstart;
        for (; j < done; j++) out[j] = (uint8_t)(buf >> (8 * (j & 7u)));
    }
};

// RLE1 undo (BZip2.swift:251-267) of the walked bytes L[0 .. n) by the whole wave: kParts independent parts (see first_safe_start),
// sizes first, then the bytes; the job's result fields.  (L was written by lanes of this wave: stores and loads of one wave
// are performed in order.)  `part_at` / `part_out`: kParts + 1 words each in LDS.
template <int WAVE>
SWC_HD void rle1_undo_to_output(Job& job, const Workspace& ws, uint32_t n, uint32_t* part_at, uint64_t* part_out, int lane) {
    gcptr T = ws.L;
    for (int part = lane; part <= kParts; part += WAVE) {
        uint32_t at;
        if (part == 0) at = 0;
        else if (part == kParts) at = n;
        else at = first_safe_start(T, n, (uint32_t)(((uint64_t)n * (uint32_t)part) / kParts));
        part_at[part] = at;
    }
    for (int part = lane; part < kParts; part += WAVE) {
        const uint32_t lo = part_at[part], hi = part_at[part + 1];
        Rle1 f{part == 0 ? 0u : 1u, part == 0 ? 0u : (lo ? (uint32_t)T[lo - 1] : 0u)};
        uint64_t bytes = 0;
        for_bytes(T, lo, hi, [&](uint32_t b) { bool c; bytes += f.feed(b, c); });
        part_out[part + 1] = bytes;
    }
    if (lane == 0) {
        part_out[0] = 0;
        for (int j = 1; j <= kParts; j++) part_out[j] += part_out[j - 1];
    }
    gptr out = (gptr)job.out;
    const uint64_t ocap = job.out_cap;
    for (int part = lane; part < kParts; part += WAVE) {
        const uint32_t lo = part_at[part], hi = part_at[part + 1];
        Rle1 f{part == 0 ? 0u : 1u, part == 0 ? 0u : (lo ? (uint32_t)T[lo - 1] : 0u)};
        OutPack o;
        o.begin(out, ocap, part_out[part]);
        for_bytes(T, lo, hi, [&](uint32_t b) {
            bool c;
            const uint32_t prev = f.prev;
            const uint32_t cnt = f.feed(b, c);
            const uint32_t v = c ? prev : b;
            for (uint32_t k = 0; k < cnt; k++) o.put(v);
        });
        o.finish();
    }
    const uint64_t total = part_out[kParts];
    job.out_len = total;
    job.status = total > ocap ? SWC_E_CAPACITY : SWC_OK;
    job.in_consumed = ws.hdr->end_bit;
    job.aux = 0;
}

// Stage 3a, second part: the segments in cycle order, the bytes laid out in L, RLE1 undone (the walk is complete).
template <int WAVE>
SWC_HD void walk_finish(Job& job, Walk& W, Stage3Lds* l, int lane) {
    const Workspace ws = W.ws;
    const uint32_t n = W.n, orig = W.orig, segs = W.segs, cap = W.cap;
    if (W.ctl[1]) return;
    for (uint32_t q = (uint32_t)lane; q < segs; q += (uint32_t)WAVE) { l->len[q] = ws.seg_len[q]; l->nxoff[q] = ws.seg_next[q]; }
    simt::wave_fence();
    auto start_of = [&](uint32_t sg) { return W.start_of(sg); };
    // ---- cycle order: from origPtr's segment along `next` until the walk is back; one n-cycle <=> the lengths add up to n
    // exactly when the start comes round again.  (All lanes run this short chain redundantly: <= 513 LDS steps.)
    const uint32_t s0 = W.seg_of(orig);
    uint32_t sg = s0, off = 0, visited = 0;
    do {
        const uint32_t nx = l->nxoff[sg];   // (every lane reads it before any lane's store below: one instruction each)
        l->nxoff[sg] = off;
        off += l->len[sg];
        sg = nx;
        visited++;
    } while (sg != s0 && visited <= segs && off <= n);
    if (!(sg == s0 && off == n && visited == segs)) return;  // several cycles: the reference keeps circling the first one
    // ---- lay out: buffered prefixes by the whole wave, overlong tails by walking those segments again
    for (uint32_t q = 0; q < segs; q++) {
        const uint32_t len = l->len[q], have = len < cap ? len : cap;
        gcptr src = ws.segbuf + (size_t)q * cap;
        gptr dst = ws.L + l->nxoff[q];
        for (uint32_t i = (uint32_t)lane * 8u; i < have; i += (uint32_t)WAVE * 8u) {
            if (have - i >= 8) store_u64(dst + i, load_u64(src + i));
            else for (uint32_t j = i; j < have; j++) dst[j] = src[j];
        }
    }
    if (lane == 0) l->ticket = 0;
    for (;;) {
        const uint32_t q = take_ticket(&l->ticket);
        if (q >= segs) break;
        const uint32_t len = l->len[q];
        if (len <= cap) continue;
        gptr dst = ws.L + l->nxoff[q];
        uint32_t cur = start_of(q);
        uint64_t acc = 0;
        for (uint32_t k = 0; k < len; k++) {
            const uint32_t v = ws.P[cur];
            acc |= (uint64_t)(v & 0xFFu) << (8 * (k & 7u));
            if ((k & 7u) == 7u) {
                if (k - 7 >= cap) store_u64(dst + (k - 7), acc);
                else for (uint32_t j = 0; j < 8; j++) if (k - 7 + j >= cap) dst[k - 7 + j] = (uint8_t)(acc >> (8 * j));
                acc = 0;
            }
            cur = v >> 8;
        }
        for (uint32_t j = len & ~7u; j < len; j++) if (j >= cap) dst[j] = (uint8_t)(acc >> (8 * (j & 7u)));
    }
    rle1_undo_to_output<WAVE>(job, ws, n, l->part_at, l->part_out, lane);
    if (lane == 0) ws.hdr->pad = kWalkDone;
}

// The fused kernel's stage 3a: the whole walk at once.  `l` is the stage's LDS; its ticket / bad words serve as the walk's.
template <int WAVE>
SWC_HD void stage3_walk_job(Job& job, Workspace ws, Stage3Lds* l, int lane) {
    Walk W;
    walk_begin(W, ws, &l->ticket, lane);
    if (!W.live) return;
    while (!walk_tick<WAVE>(W)) {}
    walk_finish<WAVE>(job, W, l, lane);
}

// ---- stage 3b: RLE1 undo, one block per lane.  job.dict (as integer) = stored block CRC (checked by stage 3c). ----------
// ---- stage 3b: everything stage 3a left (failed blocks, empty blocks, origPtr out of range, permutations that are not one
// cycle): the serial walk of BurrowsWheeler.swift:58-62 fused with the RLE1 undo, one block per lane. ---------------------
SWC_HD bool stage3_expand_needed(Workspace ws) { return ws.hdr->pad != kWalkDone; }
SWC_HD void stage3_expand_job(Job& job, Workspace ws) {
    const uint32_t st1 = ws.hdr->status;
    job.in_consumed = ws.hdr->end_bit;  // bit position just past the block (BITS for this codec)
    job.aux = 0;
    if (st1 != SWC_OK) { job.status = (int32_t)st1; job.out_len = 0; return; }
    const uint32_t n = ws.hdr->n;
    gptr out = (gptr)job.out;
    const uint64_t cap = job.out_cap;
    int st = SWC_OK;
    Rle1 f{0u, 0u};
    OutPack o;
    o.begin(out, cap, 0);
    uint32_t end = ws.hdr->orig_ptr;
    for (uint32_t i = 0; i < n; i++) {  // n == 0: BurrowsWheeler.swift:30-31, empty input => []
        if (end >= n) { st = SWC_E_REF_TRAP; break; }  // pointers[end] out of range (App. A B4)
        const uint32_t v = ws.P[end];
        end = v >> 8;
        bool c;
        const uint32_t prev = f.prev;
        const uint32_t cnt = f.feed(v & 0xFFu, c);
        const uint32_t b = c ? prev : (v & 0xFFu);
        for (uint32_t k = 0; k < cnt; k++) o.put(b);
    }
    o.finish();
    if (st == SWC_OK && o.pos > cap) st = SWC_E_CAPACITY;
    job.out_len = o.pos;
    job.status = st;
}

// ---- stage 3c: the block CRC (BZip2.swift:81), `crc` = bzip2crc32 of out[0 .. out_len) ---------------------------------
SWC_HD void stage3_check_crc(Job& job, uint32_t crc) {
    job.aux = (int32_t)crc;
    if (job.status == SWC_OK && crc != (uint32_t)(uintptr_t)job.dict) job.status = SWC_E_BZIP2_WRONG_CRC;
}

}  // namespace bzip2
}  // namespace swc
#endif
