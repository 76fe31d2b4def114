// lz_resolve.h -- phase 2 of the Deflate and LZ4 paths: LZ77 match resolution, one stream per WORKGROUP, output
// window in LDS.
//
// The reference executes every back-reference inline, one byte per `out.append`
// (Sources/Deflate/Deflate.swift:216-232, Sources/LZ4/LZ4.swift:398-410).  On the MI355X the entropy decode / sequence
// parse and the copy are split:
//
//   phase 1 (inflate_sync.h / inflate_lane.h, lz4_wave.h) appends every LITERAL to the stream's dense literal stream and
//           one 32-bit RECORD per match (or per run of literals without a match) to its record list, both in the HBM
//           workspace.  It never touches the output buffer.
//   phase 2 (this file, one stream per workgroup of T threads) walks the record list in batches of up to T records
//           that cover up to 16 T - 16 output bytes (the batch SPAN):
//             R0/R1  a workgroup prefix scan turns (literal run, length) into output positions and literal offsets;
//                    every record publishes its geometry (8 bytes) and the 16-byte output slots it owns;
//             R2     EXPANSION, one thread per output byte and step: the thread finds the record that covers its byte
//                    (per 16-byte slot: the record of the slot's first byte + a popcount over the slot's record-start
//                    flags) and either knows the byte at once -- a literal (from the staged literal window), or a match
//                    byte whose source lies BEFORE the span (final bytes of the LDS ring) -- or notes the position inside
//                    the span it copies from (a match that overlaps itself is reduced to its first period).  The result
//                    is one 16-bit CELL per byte: 0x8000 | value, or the source cell index;
//             R3     CHASE, one thread per aligned output dword and step: the four cells follow their source indices
//                    together until each meets a value (every chain ends in one: sources point strictly backwards and
//                    everything that is not an in-span copy was resolved in R2).  Read-only pointer chasing over static
//                    cells -- no waiting on producers, no ordering between threads; resolved values are written back so
//                    that later readers stop early.  Then the dword goes to the LDS ring (history of later batches) and
//                    to HBM with one aligned store -- the only time the output is written.
//           Three LDS-only barriers per batch.  Every LDS access is naturally aligned.  The cells of a span live in the
//           stale part of the ring (behind the history and the span), so the ring is the only large LDS array.
//
// Record format (u32):  lit_run[0..6] | length[7..15] | (distance - 1)[16..31]
//   length 1..511: `lit_run` (0..127) literal bytes, then a match of `length` bytes `distance` (1..65536) back;
//   length 0:      literals only: lit_run + (bits[16..31] << 7) of them (1..kMaxLitOnly).
// Every record covers at least one output byte.  Records exist only for output below the capacity: a match is
// recorded when it STARTS below it (phase 2 clamps at `limit`), literals are kept only below it.
//
// Workspace area of a stream: StreamHeader (16 bytes) | records | literal stream (at the END of the area).
#ifndef SWC_LZ_RESOLVE_H
#define SWC_LZ_RESOLVE_H

#include "swc_common.h"
#include "simt.h"

namespace swc {
namespace lzr {

constexpr uint32_t kLitRunMax = 127;       // literal bytes a match record can carry in front of its match
constexpr uint32_t kMaxLen = 511;          // longest match piece of one record
constexpr uint32_t kMaxLitOnly = 2048;     // literal bytes of one literal-only record
SWC_HD uint32_t make_match(uint32_t lit_run, uint32_t length, uint32_t distance) { return lit_run | (length << 7) | ((distance - 1u) << 16); }
SWC_HD uint32_t make_lits(uint32_t n) { return (n & 127u) | ((n >> 7) << 16); }   // 1 <= n <= kMaxLitOnly

struct StreamHeader {
    uint32_t nrec;
    uint32_t pad0;
    uint64_t nlit;    // bytes in the literal stream
};
// Records a stream of capacity `cap` can need: one per match piece (>= 3 output bytes each for Deflate, >= 4 for LZ4)
// plus the literal-only records (one per kLitRunMax + 1 literals in front of a match, one per sub-chunk tail of the
// wave-parallel Deflate decode: a tail record closes >= 50 bits of input) plus slack.
SWC_HD size_t max_records(uint64_t cap) { return (size_t)(cap / 3 + cap / 32 + 64); }
// ... and an LZ4 block whose records carry the offset of their literals (lz4_wave.h: R8): a sequence with a match yields four
// output bytes or more, a literal-only record stands for up to kMaxLitOnly bytes
SWC_HD size_t max_records8(uint64_t cap) { return (size_t)(cap / 4 + cap / 1024 + 64); }
SWC_HD size_t lit_bytes(uint64_t cap) { return (size_t)((cap + 32 + 15) & ~(uint64_t)15); }   // +32: wide flushes and reads may overshoot
// Scratch of the wave-parallel Deflate decode (inflate_sync.h): the records and literals of the 64 sub-chunks of the current
// round before their final offsets are known, as ROWS across the lanes -- row k holds the k-th record (4 bytes) / the k-th
// group of FOUR literals of every lane, so that the 64 lanes, which advance at about the same rate, fill whole cache lines
// together and read them back with coalesced loads.  Sized for the worst case (a sub-chunk of 1-bit codes); row 0 of the
// literal part is where a lane without literals so far stores.  Behind the rows: the SPILL of the stream's code tables that
// only the checked one-symbol step needs (canonical limits, sorted symbols) and the overflow of the long-code subtables.
// Sits between the record list and the literal stream.
#ifndef SWC_SYNC_CHUNK
#define SWC_SYNC_CHUNK 68   // input bytes per lane and round of the wave-parallel decode (inflate_sync.h: kSyncChunk)
#endif
// a sub-chunk decodes at most 8 * SWC_SYNC_CHUNK + 48 bits: a match takes two bits or more, a literal one
constexpr size_t kProvRecRows = (8 * SWC_SYNC_CHUNK + 48) / 2 + 5, kProvLitRows = (8 * SWC_SYNC_CHUNK + 48) / 4 + 3;
constexpr size_t kProvRecBytes = kProvRecRows * 64 * 4, kProvLitBytes = kProvLitRows * 64 * 4, kProvSpillBytes = 8192;
constexpr size_t kProvBytes = kProvRecBytes + kProvLitBytes + kProvSpillBytes;
SWC_HD size_t ws_bytes_per_job(uint64_t cap) { return ((sizeof(StreamHeader) + max_records(cap) * 4 + 15) & ~(size_t)15) + kProvBytes + lit_bytes(cap); }
// a job's literal stream inside its area of `stride` bytes (0 if the area is too small for it)
SWC_HD size_t lit_offset(size_t stride, uint64_t cap) { return stride >= lit_bytes(cap) + sizeof(StreamHeader) ? (stride - lit_bytes(cap)) & ~(size_t)15 : 0; }

struct u128 {
    uint32_t x, y, z, w;
};
SWC_HD u128 load_16(gcptr p) { return *(const SWC_AS_GLOBAL u128*)p; }   // p is 16-byte aligned
SWC_HD void store_16(gptr p, const u128& v) { *(SWC_AS_GLOBAL u128*)p = v; }

// T threads per stream; RING_LOG2: log2 of the LDS ring (history a match can reach + one span + the cells of a span)
template <int T, int RING_LOG2>
struct Lds {
    static constexpr uint32_t kRing = 1u << RING_LOG2;
    static constexpr uint32_t kRpt = 2;                   // records per thread and batch
    static constexpr uint32_t kLitWin = 8u * T;           // bytes of the literal stream staged ahead (a batch takes less than that many literals)
    alignas(16) uint8_t ring[kRing];                       // byte at virtual position v lives at ring[v % kRing]
    alignas(16) uint8_t litbuf[kLitWin];                   // literal byte at stream offset o lives at litbuf[o % kLitWin]
    alignas(8) uint64_t rec8[kRpt * T + 18];                      // per record, in CELL indices (span-relative + off): mstart | thr << 16 | distance << 32 | litkey << 48 (see R1)
    uint32_t slotw[T + T / 4 + 2];                         // per 16-cell slot: record that covers its first in-span cell | record-start flags of its cells << 16 (a last batch may span 20 T cells)
    uint32_t wave_sum[2 * (T / 64) + 2];
    uint32_t ntake, span, litspan, overlap, first_rec, nbig, bigbytes;
};

template <int T, int RING_LOG2, uint32_t KEEP, bool FAR = false>
struct Resolver {
    using L = Lds<T, RING_LOG2>;
    static constexpr uint32_t kRing = L::kRing, kMask = kRing - 1u, kLitWin = L::kLitWin, kLitMask = kLitWin - 1u, kRpt = L::kRpt;
#ifndef SWC_RESOLVE_BIG_LIT
#define SWC_RESOLVE_BIG_LIT 1024
#endif
    static constexpr uint32_t kBigLit = SWC_RESOLVE_BIG_LIT;   // a literal-only record of at least this many bytes opens a run that is copied, not expanded
    static constexpr uint32_t kLitCap = kLitWin - 16u;    // literal bytes one batch may take: what the window holds whatever its granule alignment
    static constexpr uint32_t kSpanMax = 16u * T - 16u;   // output bytes one batch may cover: at most T slots whatever the alignment
    // The LAST batch of a stream may cover up to 20 T - 16 bytes (a fifth step of R2 / R3): a stream of 64 KiB is eight spans
    // of 8,176 bytes and 128 bytes over, and a ninth batch for those costs the scan, the geometry and three barriers like any
    // other (4 % of the kernel on BASELINE configs[1]).  Not with FAR (its expansion is unrolled to four steps; LZ4 blocks are
    // megabytes long, their last batch does not matter).
    static constexpr uint32_t kSpanBig = FAR ? kSpanMax : 20u * T - 16u;
    static constexpr uint32_t kCellOff = kSpanBig + 32u;  // cells of the span live in the stale part of the ring, past the span
    static_assert(3u * (kSpanBig + 16u) + 16u <= kRing - KEEP, "ring too small for history + span + cells");
    static_assert(kMaxLitOnly + kLitRunMax + kMaxLen <= kSpanMax, "a record must fit a span");
    static_assert(kMaxLitOnly + kLitRunMax <= kLitCap, "a record's literals must fit the literal window");
    static_assert(kRpt * T <= 0x10000u, "record indices are 16 bits");
    static_assert(16u * T <= 0x8000u, "cell indices are 15 bits");
    static_assert(offsetof(L, litbuf) == kRing, "the literal window sits right behind the ring");

#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    uint64_t* prof = nullptr;   // 8 counters per stream: cycles of R0+scan, R1, R2, R3, batches (profile builds only)
#define SWC_RP(k) { const uint64_t t_ = __builtin_readcyclecounter(); pacc[k] += t_ - tlast; tlast = t_; }
#else
#define SWC_RP(k)
#endif
    L* l;
    gptr out;
    gcptr lits;        // the stream's dense literal stream (16-byte aligned base)
    uint64_t lit_cap;  // bytes that may be READ from it (allocation size, a multiple of 16)
    uint64_t limit;    // bytes of `out` that exist: min(bytes produced, capacity)

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
    SWC_D static void lds_min(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        if (v < *p) *p = v;
#endif
    }
    SWC_D static void lds_or(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        *p |= v;
#endif
    }
    // Cells are read while other threads resolve theirs (R3): a reader may see the source index or the value that
    // replaced it -- both are right (16-bit LDS accesses are single accesses; every address is read at most once per hop,
    // so nothing can be cached).  Plain LDS accesses: `volatile` would turn them into flat accesses with a memory wait.
    SWC_D static uint32_t cell_load(const uint8_t* ring, uint32_t cbase, uint32_t ci) {
        return *(const uint16_t*)(ring + ((cbase + 2u * ci) & kMask));
    }
    SWC_D static void cell_store(uint8_t* ring, uint32_t cbase, uint32_t ci, uint32_t v) {
        *(uint16_t*)(ring + ((cbase + 2u * ci) & kMask)) = (uint16_t)v;
    }

    // ---- R2, one aligned DWORD (four cells) per thread and step.  Everything is a select over precomputed per-record geometry
    // (rec8): which record (the slot word, read once for the four cells, + a popcount each), literal or match (ci < mstart), the
    // ONE LDS byte read per cell (literal window or ring), value or source index (ci >= thr).  The trip count is the same for
    // every thread (cells past the span get harmless garbage), so the loop has no per-thread exit.  OVERLAP: the batch holds a
    // match that overlaps itself (length > distance): such bytes are reduced to their first period so that chains stay short
    // (one division per cell in this variant only).  FAR (LZ4: offsets reach 65,535 bytes back, the ring keeps KEEP = 32 KiB so
    // that two workgroups fit a CU): a match byte whose source lies in front of the ring's history -- 3 % of the match bytes of
    // text -- is read from the output buffer in HBM.  The loads are issued where the cell is worked out and consumed after the
    // last step (the loop is unrolled to its four steps for that: the values wait in registers), so their latency runs under
    // the remaining steps and is paid at most once per batch.  The bytes were stored by earlier batches of this workgroup at
    // least KEEP bytes = four batches ago, by any of its threads; run() makes every wave wait for its own stores once per
    // batch (two barriers before anybody's far loads of the next batch), which covers them.
    template <bool OVERLAP>
    SWC_D void expand_cells(int t, uint32_t ncell, uint32_t off, uint32_t v0, uint32_t cbase, gcptr obase) const {
        const uint8_t* lds0 = l->ring;                      // litbuf == ring + kRing (struct layout, asserted above)
        const uint32_t jb = 4u * ((uint32_t)t & 3u);
        uint32_t m_rest[4], m[4];
#pragma unroll
        for (uint32_t c = 0; c < 4; c++) {
            m_rest[c] = ((2u << (jb + c)) - 1u) & ~1u;      // record starts in cells (0, j] of my slot
            m[c] = (uint32_t)t < 4u ? ((2u << (jb + c)) - 1u) & ~((2u << off) - 1u) : m_rest[c];   // slot 0: starts in (off, j]
        }
        const uint32_t nq = (ncell + 3u) >> 2;
        const uint32_t iters = (nq + (uint32_t)T - 1u) / (uint32_t)T;   // <= 4: a span has at most 16 T - 16 + 15 cells
        uint32_t far_val[4][4];
        uint32_t far_mask = 0;
        auto step = [&](uint32_t q, int it) {
            const uint32_t sw = l->slotw[q >> 2];
            uint32_t cell[4];
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const uint32_t ci = 4u * q + c;
                const uint32_t r = (sw & 0xFFFFu) + (uint32_t)simt::popc32((sw >> 16) & m[c]);
                m[c] = m_rest[c];
                const uint64_t rc = l->rec8[r];
                const uint32_t w0 = (uint32_t)rc, w1 = (uint32_t)(rc >> 32);
                const uint32_t mstart = w0 & 0xFFFFu, thr = w0 >> 16, dist = w1 & 0xFFFFu, lkey = w1 >> 16;
                const bool is_lit = ci < mstart;
                uint32_t x = ci - dist;                      // the cell this byte copies (when it is a match byte)
                bool inspan = ci >= thr;
                if (OVERLAP) {
                    const uint32_t mo = ci - mstart;
                    if (!is_lit && mo >= dist) x = mstart - dist + mod_small(mo & 0xFFFFu, dist);   // repeats its first period
                    inspan = !is_lit && (int32_t)x >= (int32_t)off;
                }
                const uint32_t a_lit = kRing + ((ci + lkey) & kLitMask), a_ring = (v0 + x) & kMask;
                const uint32_t byte = lds0[is_lit ? a_lit : a_ring];
                if (FAR) {
                    if (!is_lit && (int32_t)x < -(int32_t)KEEP && ci < ncell) {   // the source is older than the ring's history
                        far_val[it][c] = obase[(int64_t)(int32_t)x];
                        far_mask |= 1u << (4 * it + (int)c);
                    }
                }
                cell[c] = inspan ? x : 0x8000u | byte;
            }
            *(uint64_t*)(l->ring + ((cbase + 8u * q) & kMask)) = (uint64_t)cell[0] | ((uint64_t)cell[1] << 16) | ((uint64_t)cell[2] << 32) | ((uint64_t)cell[3] << 48);
        };
        if (FAR) {
#pragma unroll
            for (int it = 0; it < 4; it++) {
#pragma unroll
                for (int c = 0; c < 4; c++) far_val[it][c] = 0;
            }
#pragma unroll
            for (int it = 0; it < 4; it++)
                if ((uint32_t)it < iters) step((uint32_t)t + (uint32_t)it * (uint32_t)T, it);
#pragma unroll
            for (int it = 0; it < 4; it++) {   // (keeps the compiler from waiting for each load where it is issued)
#pragma unroll
                for (int c = 0; c < 4; c++) SWC_OPAQUE(far_val[it][c]);
            }
            if (far_mask != 0u) {
#pragma unroll
                for (int it = 0; it < 4; it++) {
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if (far_mask & (1u << (4 * it + c))) cell_store(l->ring, cbase, 4u * ((uint32_t)t + (uint32_t)it * (uint32_t)T) + (uint32_t)c, 0x8000u | far_val[it][c]);
                }
            }
        } else {
            uint32_t q = (uint32_t)t;
            for (uint32_t it = 0; it < iters; it++, q += (uint32_t)T) step(q, 0);
        }
        // cells of slot 0 in front of the span belong to earlier batches: final bytes of the ring (written after the loop: the
        // loop left garbage there; by the thread that wrote the dword in the loop: its own LDS accesses stay in order)
        if ((uint32_t)t < 4u) {
#pragma unroll
            for (uint32_t c = 0; c < 4; c++) {
                const uint32_t ci = 4u * (uint32_t)t + c;
                if (ci < off) cell_store(l->ring, cbase, ci, 0x8000u | lds0[(v0 + ci) & kMask]);
            }
        }
    }
    SWC_D void run(const SWC_AS_GLOBAL uint32_t* recs, uint32_t nrec) {
        using simt::PT;
        if (nrec == 0) return;
        const uint32_t A = (uint32_t)(uintptr_t)out & 15u;
        uint64_t rpos = 0;    // output bytes finished by earlier batches
        uint64_t lbase = 0;   // literal bytes consumed by earlier batches
        uint64_t lfill = 0;   // the literal window holds stream bytes [lfill - kLitWin, lfill)
        uint32_t base = 0;    // first record of the batch
        PT<uint32_t, T> r_nx0, r_nx1;       // the records at base + 2 t and base + 2 t + 1, prefetched during the previous batch
        PT<u128, T> lit_pf;                 // a 16-byte granule of the literal stream on its way into the window
        PT<uint32_t, T> lit_pf_at;          // its stream offset (0xFFFFFFFF: none)
        PT<uint32_t, T> x, y, xb, yb;       // scan values (of the thread's pair of records; xb / yb: the second record alone)
        PT<uint32_t, T> far_pf;             // FAR: what the prefetch loads of R1 returned (kept so that they are real loads; see the end of run)
        // prologue: first records, first window of literals
        SIMT_BEGIN(t, T)
            r_nx0[t] = 2u * (uint32_t)t < nrec ? recs[2u * (uint32_t)t] : 0u;
            r_nx1[t] = 2u * (uint32_t)t + 1u < nrec ? recs[2u * (uint32_t)t + 1u] : 0u;
            const uint64_t o = 16ull * (uint32_t)t;
            lit_pf_at[t] = 0xFFFFFFFFu;
            far_pf[t] = 0;
            if (o + 16 <= kLitWin && o + 16 <= lit_cap) *(u128*)(l->litbuf + o) = load_16(lits + o);
            l->slotw[t] = 0;
            if (t < T / 4 + 2) l->slotw[T + t] = 0;
            if (t < 18) l->rec8[kRpt * T + t] = 0xFFFFull;
        SIMT_END
        lfill = kLitWin;
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        uint64_t pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t tlast = __builtin_readcyclecounter();
#endif
        while (base < nrec) {
            const uint32_t vcur = A + (uint32_t)rpos;           // virtual position of the batch start (mod 2^32; the ring mask applies)
            const uint32_t off = vcur & 15u;                    // cells of slot 0 that belong to earlier batches
            const uint32_t v0 = vcur - off;
            const uint32_t cbase = (v0 + kCellOff) & kMask;
            // ---- R0: the literal granule prefetched last batch lands in the window; my two records; scan inputs
            SIMT_BEGIN(t, T)
                if (lit_pf_at[t] != 0xFFFFFFFFu) *(u128*)(l->litbuf + (lit_pf_at[t] & kLitMask)) = lit_pf[t];
                uint32_t xs = 0, ys = 0;
#pragma unroll
                for (uint32_t k = 0; k < kRpt; k++) {
                    const uint32_t r = k ? r_nx1[t] : r_nx0[t];
                    uint32_t lit = r & 127u;
                    const uint32_t len = (r >> 7) & 511u;
                    if (len == 0) lit += (r >> 16) << 7;
                    xs += lit + len;
                    ys += lit;
                    if (k == kRpt - 1) { xb[t] = lit + len; yb[t] = lit; }
                }
                x[t] = xs;
                y[t] = ys;
                if (t == 0) { l->overlap = 0; l->first_rec = r_nx0[t]; l->nbig = kRpt * (uint32_t)T; }
            SIMT_END
            simt::group_scan2_incl<T>(x, y, l->wave_sum);       // (barrier A inside)
            SWC_RP(0)
            // ---- long runs of plain literals (stored blocks, incompressible data) do not go through cells: the run of
            // literal-only records the batch starts with is copied from the literal stream to the output and the ring, eight
            // bytes per thread and step.  (Through the cells such data ran at a third of the speed of text: the literal window
            // halves the batches, and every byte pays the full price of the expansion.)
            {
                const uint32_t fr = simt::uniform(l->first_rec);
                if (((fr >> 7) & 511u) == 0u && (fr & 127u) + ((fr >> 16) << 7) >= kBigLit) {
                    SIMT_BEGIN(t, T)
#pragma unroll
                        for (uint32_t k = 0; k < kRpt; k++) {
                            const uint32_t r = k ? r_nx1[t] : r_nx0[t];
                            const bool big = base + kRpt * (uint32_t)t + k < nrec && ((r >> 7) & 511u) == 0u && (r & 127u) + ((r >> 16) << 7) >= kBigLit;
                            if (!big) lds_min(&l->nbig, kRpt * (uint32_t)t + k);
                        }
                    SIMT_END_BARRIER
                    const uint32_t nb = simt::uniform(l->nbig);          // records of the run (>= 1: the first one is big)
                    SIMT_BEGIN(t, T)
                        if (kRpt * (uint32_t)t <= nb - 1u && nb - 1u < kRpt * (uint32_t)t + kRpt)
                            l->bigbytes = ((nb - 1u) & 1u) ? y[t] : y[t] - yb[t];   // literal bytes up to and including record nb - 1
                    SIMT_END_BARRIER
                    const uint32_t nbytes = simt::uniform(l->bigbytes);
                    const uint32_t V = A + (uint32_t)rpos;                 // virtual position of the first byte
                    const uint32_t head = (0u - V) & 7u, nhead = head < nbytes ? head : nbytes;
                    const uint32_t npiece = (nbytes - nhead) >> 3, ntail = (nbytes - nhead) & 7u;
                    gcptr src = lits + lbase;
                    SIMT_BEGIN(t, T)
                        l->slotw[t] = 0;
                        if (t < T / 4 + 2) l->slotw[T + t] = 0;
                        // (the batch before may have left the bytes of its last, incomplete dword to its successor: they are in the
                        // ring only)
                        if ((uint32_t)t < (V & 3u) && (uint64_t)((V & 3u) - (uint32_t)t) <= rpos) {
                            const uint32_t back = (V & 3u) - (uint32_t)t;
                            if (rpos - back < limit) out[rpos - back] = l->ring[(V - back) & kMask];
                        }
                        // the bytes up to the first 8-aligned position and behind the last one, one thread each
                        // (a run longer than the ring wraps around it: only its last kRing bytes are written there -- a byte that
                        // a later byte of the run replaces belongs to another thread, and the two would race)
                        if ((uint32_t)t < nhead) {
                            const uint32_t b = src[t];
                            if (nbytes - (uint32_t)t <= kRing) l->ring[(V + (uint32_t)t) & kMask] = (uint8_t)b;
                            if (rpos + (uint32_t)t < limit) out[rpos + (uint32_t)t] = (uint8_t)b;
                        }
                        if ((uint32_t)t < ntail) {
                            const uint32_t i = nhead + 8u * npiece + (uint32_t)t;
                            const uint32_t b = src[i];
                            l->ring[(V + i) & kMask] = (uint8_t)b;
                            if (rpos + i < limit) out[rpos + i] = (uint8_t)b;
                        }
                        for (uint32_t p = (uint32_t)t; p < npiece; p += (uint32_t)T) {
                            const uint32_t i = nhead + 8u * p;
                            const uint64_t v = load_u64(src + i);
                            if (nbytes - i <= kRing) *(uint64_t*)(l->ring + ((V + i) & kMask)) = v;
                            if (rpos + i + 8u <= limit) store_u64(out + rpos + i, v);
                            else for (uint32_t e = 0; e < 8; e++) if (rpos + i + e < limit) out[rpos + i + e] = (uint8_t)(v >> (8 * e));
                        }
                    SIMT_END
                    rpos += nbytes;
                    lbase += nbytes;
                    base += nb;
                    // the records and the literal window of what follows (their prefetches were aimed elsewhere)
                    lfill = (lbase & ~(uint64_t)15) + kLitWin;
                    SIMT_BEGIN(t, T)
                        // FAR: a run of more than KEEP bytes may be followed at once by a match whose source lies inside the run
                        // but in front of the ring's history -- the next batch would read it from `out` with nothing between
                        // this batch's stores and that load but an LDS-only barrier.  So: my stores done, THEN the barrier.
                        if (FAR) simt::vmem_fence();
                        const uint32_t nx = base + kRpt * (uint32_t)t;
                        r_nx0[t] = nx < nrec ? recs[nx] : 0u;
                        r_nx1[t] = nx + 1u < nrec ? recs[nx + 1u] : 0u;
                        lit_pf_at[t] = 0xFFFFFFFFu;
                        const uint64_t o = (lbase & ~(uint64_t)15) + 16ull * (uint32_t)t;
                        if (16u * (uint32_t)t < kLitWin && o + 16 <= lit_cap) *(u128*)(l->litbuf + ((uint32_t)o & kLitMask)) = load_16(lits + o);
                    SIMT_END_BARRIER
                    continue;
                }
            }
            // ---- R1: batch geometry, two records per thread, in CELL indices (span-relative position + off)
            const uint32_t lbk = (uint32_t)lbase;
            const uint32_t span_max = limit - rpos <= (uint64_t)kSpanBig ? kSpanBig : kSpanMax;   // (everything that is left, if it fits one long batch)
            gcptr obase_r1 = (gcptr)((const SWC_AS_GLOBAL uint8_t*)out + ((int64_t)rpos - (int64_t)(vcur & 15u)));   // output position of cell 0 (R2 computes the same)
            SIMT_BEGIN(t, T)
                // FAR: my stores of the batches before are done -- one barrier (B) in front of anybody's loads of old output
                // (expand_cells; the prefetches below only touch bytes of four batches ago and more)
                if (FAR) simt::vmem_fence();
#pragma unroll
                for (uint32_t k = 0; k < kRpt; k++) {
                    const uint32_t i = kRpt * (uint32_t)t + k;     // my record of the batch
                    const uint32_t r = k ? r_nx1[t] : r_nx0[t];
                    uint32_t lit = r & 127u;
                    const uint32_t len = (r >> 7) & 511u;
                    if (len == 0) lit += (r >> 16) << 7;
                    const uint32_t end = k ? x[t] : x[t] - xb[t], lit_end = k ? y[t] : y[t] - yb[t];
                    const uint32_t start = end - (lit + len), mstart = start + lit;
                    const bool take = r != 0u && end <= span_max && lit_end <= kLitCap;
                    // mstart: first match cell; thr: first cell that copies from INSIDE the span (a match cell at or past distance +
                    // off; 0xFFFF: none); litkey: (stream offset of the record's first literal) - (its first cell), modulo the window
                    const uint32_t dist = len ? (r >> 16) + 1u : 1u;
                    const uint32_t mstart_c = mstart + off;
                    uint32_t thr = dist + off > mstart_c ? dist + off : mstart_c;
                    if (len == 0 || thr > 0xFFFFu) thr = 0xFFFFu;
                    const uint32_t lkey = (lbk + (lit_end - lit) - (start + off)) & kLitMask;
                    if (FAR) {
                        // a match whose source lies in front of the ring's history is read from the output buffer by the cells of
                        // R2: the lines are asked for HERE, a barrier and the first steps of R2 earlier, so that those reads find
                        // them in the cache instead of ending the batch on a round trip to HBM
                        const int32_t x0 = (int32_t)mstart_c - (int32_t)dist;
                        uint32_t pv = 0;
                        if (take && len != 0u && x0 < -(int32_t)KEEP) {
                            pv = obase_r1[(int64_t)x0];                       // (one byte: the line is what is wanted)
                            const int32_t x1 = x0 + (int32_t)len - 1;
                            if ((x1 >> 6) != (x0 >> 6) && x1 < -(int32_t)KEEP) pv ^= (uint32_t)obase_r1[(int64_t)x1] << 8;
                        }
                        far_pf[t] ^= pv;
                    }
                    l->rec8[i] = take ? (uint64_t)mstart_c | ((uint64_t)thr << 16) | ((uint64_t)(dist & 0xFFFFu) << 32) | ((uint64_t)lkey << 48)
                                      : 0xFFFFull;
                    if (take) {
                        // the slots whose first in-span cell I cover, and the flag of my first cell
                        for (uint32_t q = start == 0 ? 0u : (start + off + 15u) >> 4; q == 0 ? start == 0 : 16u * q - off < end; q++) lds_or(&l->slotw[q], i);
                        const uint32_t c0 = start + off;
                        lds_or(&l->slotw[c0 >> 4], 0x10000u << (c0 & 15u));
                        if (len > dist) l->overlap = 1;      // (every writer stores the same value)
                        if (i == kRpt * T - 1) { l->ntake = kRpt * (uint32_t)T; l->span = end; l->litspan = lit_end; }
                    } else if (i != 0 && start <= span_max && lit_end - lit <= kLitCap && base + i - 1u < nrec) {
                        // the first record that is not taken (or the first one past the last record) closes the batch: `start` is
                        // the end of the record before it, which exists and fits span and window, i.e. was taken
                        l->ntake = i; l->span = start; l->litspan = lit_end - lit;
                    }
                }
            SIMT_END_BARRIER                                     // barrier B
            SWC_RP(1)
            const uint32_t ntake = simt::uniform(l->ntake), span = simt::uniform(l->span), litspan = simt::uniform(l->litspan);
            const bool overlap = simt::uniform(l->overlap) != 0u;
            const bool last_batch = base + ntake >= nrec;
            const uint32_t ncell = span + off;                   // cells [off, ncell) are this batch's bytes, [0, off) belong to earlier ones
            // the literal window after this batch: [lbase + litspan, .. + kLitWin) rounded down to granules
            const uint64_t lfill_next = (lbase + litspan + kLitWin) & ~(uint64_t)15;
            // Dwords [0, nfull) lie inside the span with all four cells; of those, [qmin, qlim) also lie inside the output and leave
            // with one store at a 32-bit offset from a wave-uniform base (no 64-bit address arithmetic per thread).
            const uint32_t nfull = ncell >> 2;
            const int64_t base64 = (int64_t)rpos - (int64_t)off;                 // output position of the span's cell 0 (negative only in the first batch)
            const uint32_t qmin = base64 < 0 ? (uint32_t)((3 - base64) >> 2) : 0u;
            const uint64_t room = (int64_t)limit > base64 ? (uint64_t)((int64_t)limit - base64) : 0u;
            const uint32_t qlim = room >= 0x100000000ull ? 0x40000000u : (uint32_t)room >> 2;
            gptr obase = (gptr)((SWC_AS_GLOBAL uint8_t*)out + base64);
            // ---- R2: prefetch for the next batch, then the cells
            SIMT_BEGIN(t, T)
                {
                    const uint32_t nx = base + ntake + kRpt * (uint32_t)t;
                    r_nx0[t] = nx < nrec ? recs[nx] : 0u;
                    r_nx1[t] = nx + 1u < nrec ? recs[nx + 1u] : 0u;
                    const uint64_t o = lfill + 16ull * (uint32_t)t;
                    const bool want = o < lfill_next && o + 16 <= lit_cap;
                    lit_pf_at[t] = want ? (uint32_t)o : 0xFFFFFFFFu;
                    if (want) lit_pf[t] = load_16(lits + o);
                }
                if (overlap) expand_cells<true>(t, ncell, off, v0, cbase, (gcptr)obase);
                else expand_cells<false>(t, ncell, off, v0, cbase, (gcptr)obase);
            SIMT_END_BARRIER                                     // barrier C: the cells are complete
            SWC_RP(2)
            // ---- R3: one aligned DWORD per thread and step: chase what is unresolved, then the dword leaves for the ring and for HBM
            SIMT_BEGIN(t, T)
                if ((uint32_t)t * 16u < ncell) l->slotw[t] = 0;   // (read in R2 only; the next batch sets it after its barrier A)
                if ((uint32_t)t < T / 4 + 2 && ((uint32_t)T + (uint32_t)t) * 16u < ncell + 16u) l->slotw[T + t] = 0;
                // (a last, incomplete dword -- ncell % 4 != 0 -- is the business of ONE thread after the loop)
                for (uint32_t q = (uint32_t)t; q < nfull; q += (uint32_t)T) {
                    const uint32_t ca = (cbase + 8u * q) & kMask;
                    const uint64_t c4 = *(const uint64_t*)(l->ring + ca);
                    uint32_t c[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) c[e] = (uint32_t)(c4 >> (16 * e)) & 0xFFFFu;
                    // the four chains advance together: one hop = four independent LDS reads in flight.  A cell that is already a
                    // value re-reads a harmless address and keeps its value (selects, no branch per cell).
                    const uint32_t was = (c[0] & c[1] & c[2] & c[3]) & 0x8000u;
                    while (!((c[0] & c[1] & c[2] & c[3]) & 0x8000u)) {
                        uint32_t n[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) n[e] = cell_load(l->ring, cbase, c[e] & 0x7FFFu);
#pragma unroll
                        for (int e = 0; e < 4; e++) c[e] = c[e] < 0x8000u ? n[e] : c[e];
                    }
                    if (!was)   // resolved values back into my cells: later readers stop here
                        *(uint64_t*)(l->ring + ca) = (uint64_t)c[0] | ((uint64_t)c[1] << 16) | ((uint64_t)c[2] << 32) | ((uint64_t)c[3] << 48);
                    const uint32_t word = (c[0] & 0xFFu) | ((c[1] & 0xFFu) << 8) | ((c[2] & 0xFFu) << 16) | ((c[3] & 0xFFu) << 24);
                    *(uint32_t*)(l->ring + ((v0 + 4u * q) & kMask)) = word;
                    if (q >= qmin && q < qlim) {
                        *(SWC_AS_GLOBAL uint32_t*)(obase + (size_t)(4u * q)) = word;
                    } else {   // the dword straddles the start of the output (first batch) or the limit
#pragma unroll 1
                        for (int e = 0; e < 4; e++) {
                            const int64_t a = base64 + 4 * (int64_t)q + e;
                            if (a >= 0 && (uint64_t)a < limit) out[a] = (uint8_t)(word >> (8 * e));
                        }
                    }
                }
                if ((ncell & 3u) != 0u && (uint32_t)t == (nfull & (uint32_t)(T - 1))) {
                    // the incomplete dword: its cells beyond the span belong to the next batch, which will finish and store
                    // it -- unless this is the last batch, whose bytes leave one by one
                    const uint32_t q = nfull;
                    const uint64_t c4 = *(const uint64_t*)(l->ring + ((cbase + 8u * q) & kMask));
                    uint32_t c[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        c[e] = (uint32_t)(c4 >> (16 * e)) & 0xFFFFu;
                        if (4u * q + (uint32_t)e >= ncell) c[e] = 0x8000u;
                    }
                    const uint32_t was = (c[0] & c[1] & c[2] & c[3]) & 0x8000u;
                    while (!((c[0] & c[1] & c[2] & c[3]) & 0x8000u)) {
#pragma unroll
                        for (int e = 0; e < 4; e++) if (!(c[e] & 0x8000u)) c[e] = cell_load(l->ring, cbase, c[e]);
                    }
                    if (!was)
                        for (int e = 0; e < 4; e++) if (4u * q + (uint32_t)e < ncell) cell_store(l->ring, cbase, 4u * q + (uint32_t)e, c[e]);
                    const uint32_t word = (c[0] & 0xFFu) | ((c[1] & 0xFFu) << 8) | ((c[2] & 0xFFu) << 16) | ((c[3] & 0xFFu) << 24);
                    *(uint32_t*)(l->ring + ((v0 + 4u * q) & kMask)) = word;
                    if (last_batch) {
#pragma unroll 1
                        for (int e = 0; e < 4; e++) {
                            const int64_t a = base64 + 4 * (int64_t)q + e;
                            if (4u * q + (uint32_t)e < ncell && a >= 0 && (uint64_t)a < limit) out[a] = (uint8_t)(word >> (8 * e));
                        }
                    }
                }
            SIMT_END
            SWC_RP(3)
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
            pacc[4] += 1; pacc[5] += span; pacc[6] += ntake;
#endif
            rpos += span;
            lbase += litspan;
            lfill = lfill_next > lfill ? lfill_next : lfill;
            base += ntake;
        }
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
        if (prof && threadIdx.x == 0) for (int k = 0; k < 8; k++) prof[k] = pacc[k];
#endif
        if (FAR) {   // (the prefetched words have a use, so the loads stay)
            SIMT_BEGIN(t, T) if (far_pf[t] == 0x9E3779B9u) l->wave_sum[0] = far_pf[t]; SIMT_END
        }
    }
};

// One job: `ws` is the stream's workspace area of `area` bytes written by phase 1.
template <int T, int RING_LOG2, uint32_t KEEP, bool FAR = false>
SWC_D void resolve_job(const Job& job, const uint8_t* ws, size_t area, Lds<T, RING_LOG2>* lds, uint64_t* prof = nullptr) {
    const SWC_AS_GLOBAL StreamHeader* h = (const SWC_AS_GLOBAL StreamHeader*)ws;
    const size_t lo = lit_offset(area, job.out_cap);
    if (lo == 0) return;   // no literal stream: phase 1 reported SWC_E_NEED_WORKSPACE for this job
    Resolver<T, RING_LOG2, KEEP, FAR> rs;
    rs.l = lds;
    rs.out = (gptr)job.out;
    rs.lits = (gcptr)ws + lo;
    rs.lit_cap = lit_bytes(job.out_cap);
    rs.limit = job.out_len < job.out_cap ? job.out_len : job.out_cap;
#if defined(SWC_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
    rs.prof = prof;
#else
    (void)prof;
#endif
    rs.run((const SWC_AS_GLOBAL uint32_t*)(ws + sizeof(StreamHeader)), h->nrec);
}

}  // namespace lzr
}  // namespace swc
#endif
