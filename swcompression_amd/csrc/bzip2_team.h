// bzip2_team.h -- BZip2 stage 3 as kernels of its own: the inverse Burrows-Wheeler walk out of the XCD's L2.
//
// The fused kernel (bzip2_block.h) keeps a block with ONE wavefront from its first code to its last output byte: 6,144 blocks
// are in flight, their pointer arrays P (3.6 MB each at level nine) are 22 GB, and every one of the walk's n dependent gathers
// v = P[end]; end = v >> 8 is a row activation in HBM (51 G gathers/s, tools/micro/gather_bench.hip).  An XCD's L2 (4 MiB) holds
// ONE block's array, and 32 CUs gathering from it reach 240 G/s -- but a block walked alone is as slow as its longest segment.
// So (profiles/r05e_gather_bench.txt, mode 3):
//   * the chip is eight TEAMS, one per XCD (HW_REG_XCC_ID); team x owns the blocks x, x + 8, x + 16 ... and takes them in order;
//   * a block's cycle is cut at the indices that are multiples of M = 2^mbits (and at origPtr) into up to kSegs2 = 16,384
//     segments instead of 513 -- at level nine M = 64, a segment is 64 steps on average, the longest a few hundred;
//   * a team's walkers (1,024 per CU) draw (block, segment) TICKETS from the team's counter -- the tickets of a block, then
//     those of the next: no barrier between blocks, the stragglers of a block finish while the others are on the next one,
//     and the team's L2 holds the one or two blocks its walkers are on;
//   * a walker keeps the bytes it meets in the segment's buffer (non-temporal stores: they must not push P out of the L2)
//     and leaves the segment's length and successor as the fused kernel's walk does -- and, for the one segment in 55 that is
//     longer than its buffer, the index at which the buffer was full.
// Then a wavefront per block (team_finish) puts the segments in cycle order -- 14,000 segments are themselves a chain too long
// to follow, so it is cut the same way once more: at the segments whose number is a multiple of 64 -- lays the bytes out in L
// (buffered prefixes by eight lanes per segment, the long segments finished together from where their buffers ended), and
// undoes RLE1 with the code of the fused kernel (rle1_undo_to_output).  Whatever does not check out (a permutation that is
// not one cycle, a damaged block) is left to the serial fallback of stage 3b, as before.
//
// Kernels: team_prep (segment counts of the blocks, their prefix per team, the counters), team_walk, team_finish; the fused
// kernel runs with its walk switched off in front of them.  launch_bzip2 takes this path for every launch but the tiny ones
// (one block of 100 kB is already faster this way: DESIGN.md 5.2).
#ifndef SWC_BZIP2_TEAM_H
#define SWC_BZIP2_TEAM_H

#include "bzip2_block.h"

namespace swc {
namespace bzip2 {

constexpr uint32_t kTeams = 8;
constexpr uint32_t kSuper = 64;                       // team_finish: the chain of segments is cut at every 64th segment
constexpr uint32_t kSupers = kSegs2 / kSuper + 2;     // ... into at most this many pieces (+ origPtr's segment)
constexpr uint32_t kOver = 1024;                      // team_finish: segments longer than their buffers it can take (e^-4 of them are: 260 of 14,000)
// words of a block's own (behind the per-segment arrays): [0] segments of the block, [1] segments of the team's blocks in front
// of it, [2] the walk ran away (not a permutation); block 0 also holds, for team x, at [16 + 32 x]: the ticket counter, at
// [16 + 32 x + 1]: the segments of all of the team's blocks
constexpr uint32_t kTwSegs = 0, kTwBefore = 1, kTwBad = 2, kTwTeam = 16;
static_assert(kTwTeam + 32 * kTeams <= kTeamWords, "the team counters fit block 0's words");

struct TeamSegs {
    SWC_AS_GLOBAL uint64_t* ln;      // [team_seg_slots(lcap)] length of the segment | the segment that begins where it ends << 32 (one load per step of the ordering)
    SWC_AS_GLOBAL uint32_t* off;     // [slots] team_finish: the segment's offset in the block (an array of its own: written by
                                     // some lanes and read by others, it must not sit in lines the wave has read before)
    SWC_AS_GLOBAL uint32_t* resume;  // [slots] a segment longer than its buffer: the index its walk had reached when the buffer was full
    SWC_AS_GLOBAL uint32_t* words;   // [kTeamWords]
};
SWC_HD TeamSegs team_segs(const Workspace& w) {
    TeamSegs t;
    const uint32_t slots = team_seg_slots(w.lcap);
    t.ln = (SWC_AS_GLOBAL uint64_t*)w.seg_len;    // (the area is 16-byte aligned, the slots a multiple of four)
    t.off = w.seg_len + 2 * slots;
    t.resume = w.seg_len + 3 * slots;
    t.words = w.seg_len + 4 * slots;
    return t;
}
// how a block of n bytes with origin pointer `orig` is cut
struct Cut {
    uint32_t n, orig, mbits, mask, regs, segs, cap;
    bool extra;
    SWC_HD void set(uint32_t n_, uint32_t orig_) {
        n = n_; orig = orig_;
        mbits = seg_mbits2(n); mask = (1u << mbits) - 1u;
        regs = (n + mask) >> mbits;
        extra = (orig & mask) != 0u;
        segs = regs + (extra ? 1u : 0u);
        cap = kSegCapFactor << mbits;
    }
    SWC_HD bool is_mark(uint32_t i) const { return (i & mask) == 0u || i == orig; }
    SWC_HD uint32_t seg_of(uint32_t i) const { return (extra && i == orig) ? regs : i >> mbits; }
    SWC_HD uint32_t start_of(uint32_t s) const { return s < regs ? s << mbits : orig; }
};
SWC_HD bool team_walkable(const Workspace& w) {
    return w.hdr->status == SWC_OK && w.hdr->n != 0u && w.hdr->orig_ptr < w.hdr->n;
}

// ---- team_prep: one wavefront per team (WAVE lanes; the host build: one) ---------------------------------------------------
template <int WAVE>
SWC_HD void team_prep(uint8_t* ws_base, size_t lcap, uint32_t n_blocks, uint32_t team, int lane) {
    uint32_t before = 0;
    for (uint32_t k0 = 0; team + kTeams * k0 < n_blocks; k0 += (uint32_t)WAVE) {
        const uint32_t b = team + kTeams * (k0 + (uint32_t)lane);
        uint32_t segs = 0;
        Workspace w;
        if (b < n_blocks) {
            w = carve(ws_base, b, lcap);
            if (team_walkable(w)) { Cut c; c.set(w.hdr->n, w.hdr->orig_ptr); segs = c.segs; }
        }
        uint32_t incl = segs;
#if defined(__HIP_DEVICE_COMPILE__)
        incl = simt::wave_scan_incl_dev(segs);
#endif
        if (b < n_blocks) {
            const TeamSegs t = team_segs(w);
            t.words[kTwSegs] = segs;
            t.words[kTwBefore] = before + incl - segs;
            t.words[kTwBad] = 0;
            w.hdr->pad = kWalkNone;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        before += (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
#else
        before += incl;
#endif
    }
    if (lane == 0) {
        const TeamSegs t0 = team_segs(carve(ws_base, 0, lcap));
        t0.words[kTwTeam + 32u * team] = 0;
        t0.words[kTwTeam + 32u * team + 1u] = before;
    }
}

// ---- team_walk: every thread of the team draws tickets until there are none -------------------------------------------------
SWC_HD uint32_t team_ticket(SWC_AS_GLOBAL uint32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return (*p)++;
#endif
}
SWC_HD void store_u64_stream(gptr p, uint64_t v) {   // past the caches: the segment buffers are read once, much later
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v, (SWC_AS_GLOBAL uint64_t*)p);
#else
    store_u64(p, v);
#endif
}
// (`home`: the team of the XCD the thread runs on.  A thread whose team has no tickets left goes on with the next team's: the
// walk is complete whatever the workgroups' spread over the XCDs was, and the last blocks of a launch are shared.)
SWC_HD void team_walk(uint8_t* ws_base, size_t lcap, uint32_t n_blocks, uint32_t home) {
    const TeamSegs t0 = team_segs(carve(ws_base, 0, lcap));
    for (uint32_t d = 0; d < kTeams; d++) {
        const uint32_t team = (home + d) % kTeams;
        SWC_AS_GLOBAL uint32_t* counter = t0.words + kTwTeam + 32u * team;
        const uint32_t total = counter[1];
        uint32_t k = 0;                                 // the team's k-th block ...
        uint32_t first = 0, segs = 0;                   // ... holds the tickets [first, first + segs)
        bool have = false;
        Workspace w;
        TeamSegs ts;
        Cut c;
        for (;;) {
            const uint32_t t = team_ticket(counter);
            if (t >= total) break;
            // the block of the ticket: tickets only grow, so the search goes on from the last block
            bool lost = false;
            while (!have || t >= first + segs) {
                if (have) k++;
                const uint32_t b = team + kTeams * k;
                if (b >= n_blocks) { lost = true; break; }   // (cannot happen: the tickets are the blocks' segments)
                w = carve(ws_base, b, lcap);
                ts = team_segs(w);
                first = ts.words[kTwBefore];
                segs = ts.words[kTwSegs];
                have = true;
                if (t < first + segs) c.set(w.hdr->n, w.hdr->orig_ptr);
            }
            if (lost) break;
            const uint32_t sg = t - first;
            gptr buf = w.segbuf + (size_t)sg * c.cap;
            uint32_t cur = c.start_of(sg), steps = 0;
            uint64_t acc = 0;
            do {
                const uint32_t v = w.P[cur];
                acc |= (uint64_t)(v & 0xFFu) << (8u * (steps & 7u));
                if ((steps & 7u) == 7u) {
                    if (steps < c.cap) store_u64_stream(buf + (steps - 7u), acc);      // (cap is a multiple of eight)
                    acc = 0;
                }
                steps++;
                cur = v >> 8;
                if (steps == c.cap) ts.resume[sg] = cur;
            } while (!c.is_mark(cur) && steps <= c.n);
            if ((steps & 7u) != 0u && (steps & ~7u) < c.cap) store_u64_stream(buf + (steps & ~7u), acc);
            if (steps > c.n) ts.words[kTwBad] = 1;      // cannot happen for a permutation
            ts.ln[sg] = (uint64_t)steps | ((uint64_t)c.seg_of(cur) << 32);
        }
    }
}

// ---- team_finish: a wavefront per block ------------------------------------------------------------------------------------------
// (3.1 KB per wavefront -- 8 wavefronts per SIMD: everything in this kernel waits for memory, the blocks in flight are what
// hides it -- so what the lay-out and the RLE1 undo need lies over the arrays of the ordering, which are dead by then)
struct FinishLds {
    union {
        struct {
            uint32_t sum[kSupers];       // bytes of the piece that begins at marked segment j
            uint32_t cnt[kSupers];       // its segments
            uint32_t nx[kSupers];        // the piece that begins where it ends; then: the piece's offset in the block
        } o;
        struct {
            uint16_t over[kOver];        // the segments longer than their buffers
            uint32_t part_at[kParts + 1];
            uint64_t part_out[kParts + 1];
        } a;
    };
    uint32_t n_over, bad;
};
static_assert(sizeof(FinishLds) <= 3 * kSupers * 4 + 16, "the lay-out's arrays fit over the ordering's");
template <int WAVE>
SWC_HD void team_finish(Job& job, Workspace ws, FinishLds* l, int lane) {
    if (!team_walkable(ws)) return;
    const TeamSegs ts = team_segs(ws);
    if (ts.words[kTwBad]) return;
    Cut c;
    c.set(ws.hdr->n, ws.hdr->orig_ptr);
    const uint32_t n = c.n, segs = c.segs;
    // pieces of the chain of segments: they begin at the segments 0, 64, 128 ... and at origPtr's segment
    const uint32_t s0 = c.seg_of(c.orig);
    const uint32_t regular = (segs + kSuper - 1u) / kSuper;
    const bool s0_extra = (s0 % kSuper) != 0u;
    const uint32_t pieces = regular + (s0_extra ? 1u : 0u);
    auto piece_start = [&](uint32_t j) { return j < regular ? j * kSuper : s0; };
    auto is_start = [&](uint32_t sg) { return sg % kSuper == 0u || sg == s0; };
    auto piece_of = [&](uint32_t sg) { return (s0_extra && sg == s0) ? regular : sg / kSuper; };
    if (lane == 0) { l->bad = 0; }
    simt::wave_fence();
    for (uint32_t j = (uint32_t)lane; j < pieces; j += (uint32_t)WAVE) {
        uint32_t sg = piece_start(j), bytes = 0, count = 0;
        do {
            const uint64_t v = ts.ln[sg];
            bytes += (uint32_t)v;
            sg = (uint32_t)(v >> 32);
            count++;
        } while (sg < segs && !is_start(sg) && count <= segs);
        if (sg >= segs || count > segs) { l->bad = 1; sg = 0; }
        l->o.sum[j] = bytes; l->o.cnt[j] = count; l->o.nx[j] = piece_of(sg);
    }
    simt::wave_fence();
    if (l->bad) return;
    // cycle order of the pieces from origPtr's: one n-cycle <=> bytes and segments add up exactly when the start comes round again
    // (all lanes run this short chain redundantly: <= kSupers LDS steps)
    {
        const uint32_t j0 = piece_of(s0);
        uint32_t j = j0, off = 0, visited = 0, seen = 0;
        do {
            const uint32_t nxt = l->o.nx[j];   // (every lane reads it before any lane's store below: one instruction each)
            l->o.nx[j] = off;
            off += l->o.sum[j];
            seen += l->o.cnt[j];
            j = nxt;
            visited++;
        } while (j != j0 && visited <= pieces && off <= n);
        if (!(j == j0 && off == n && visited == pieces && seen == segs)) return;   // several cycles: the reference keeps circling the first one
    }
    simt::wave_fence();
#if defined(SWC_TF_CUT) && SWC_TF_CUT == 1   // (timing experiments only: wrong results)
    return;
#endif
    // every segment's offset: the pieces once more
    for (uint32_t j = (uint32_t)lane; j < pieces; j += (uint32_t)WAVE) {
        uint32_t sg = piece_start(j), off = l->o.nx[j];
        for (uint32_t q = 0; q < l->o.cnt[j]; q++) {
            const uint64_t v = ts.ln[sg];
            ts.off[sg] = off;
            off += (uint32_t)v;
            sg = (uint32_t)(v >> 32);
        }
    }
    simt::vmem_fence();
#if defined(SWC_TF_CUT) && SWC_TF_CUT == 2
    return;
#endif
    // lay out.  The buffered prefixes: eight lanes per segment, eight bytes each per step (a lane per segment would read a line
    // per eight bytes and wait for the longest of 64 segments)
#ifndef SWC_TF_LPS
#define SWC_TF_LPS 8   // lanes per segment of the prefix copy (the finish with 2 / 4 / 8 / 16: 49.2 / 48.0 / 48.8 / 60.2 ms)
#endif
    constexpr int kLps = WAVE >= 8 ? SWC_TF_LPS : 1;
    const uint32_t sub = (uint32_t)lane % kLps, grp = (uint32_t)lane / kLps;
    if (lane == 0) l->n_over = 0;
    simt::wave_fence();
    for (uint32_t q0 = 0; q0 < segs; q0 += (uint32_t)(WAVE / kLps)) {
        const uint32_t q = q0 + grp;
        if (q >= segs) continue;
        const uint32_t len = (uint32_t)ts.ln[q], have = len < c.cap ? len : c.cap;
        gcptr src = ws.segbuf + (size_t)q * c.cap;
        gptr dst = ws.L + ts.off[q];
        for (uint32_t i0 = 8u * sub; i0 < have; i0 += 32u * kLps) {          // four pieces per lane in flight together
            const uint32_t i1 = i0 + 8u * kLps, i2 = i1 + 8u * kLps, i3 = i2 + 8u * kLps;
            const uint64_t w0 = load_u64(src + i0);                            // (the buffers are read past `have`, never past their end: i < cap)
            const uint64_t w1 = i1 < have ? load_u64(src + i1) : 0, w2 = i2 < have ? load_u64(src + i2) : 0, w3 = i3 < have ? load_u64(src + i3) : 0;
            const uint32_t at[4] = {i0, i1, i2, i3};
            const uint64_t ww[4] = {w0, w1, w2, w3};
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint32_t i = at[p];
                if (i >= have) continue;
                if (i + 8u <= have) store_u64(dst + i, ww[p]);
                else for (uint32_t k = i; k < have; k++) dst[k] = (uint8_t)(ww[p] >> (8u * (k - i)));
            }
        }
        if (len > c.cap && sub == 0u) {
#if defined(__HIP_DEVICE_COMPILE__)
            const uint32_t e = __hip_atomic_fetch_add(&l->n_over, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
            const uint32_t e = l->n_over++;
#endif
            if (e < kOver) l->a.over[e] = (uint16_t)q;
        }
    }
    simt::wave_fence();
    const uint32_t n_over = l->n_over;
    if (n_over > kOver) return;                       // (a distribution of segment lengths no data has shown: the serial walk)
#if defined(SWC_TF_CUT) && SWC_TF_CUT == 3
    return;
#endif
    // ... and what lies behind the buffers: a lane per such segment goes on from where the segment's walk had filled its buffer
    for (uint32_t e = (uint32_t)lane; e < n_over; e += (uint32_t)WAVE) {
        const uint32_t q = l->a.over[e], len = (uint32_t)ts.ln[q];
        gptr dst = ws.L + ts.off[q];
        uint32_t cur = ts.resume[q];
        for (uint32_t k = c.cap; k < len; k++) {
            const uint32_t v = ws.P[cur];
            dst[k] = (uint8_t)v;
            cur = v >> 8;
        }
    }
    simt::vmem_fence();
#if defined(SWC_TF_CUT) && SWC_TF_CUT == 4
    return;
#endif
    rle1_undo_to_output<WAVE>(job, ws, n, l->a.part_at, l->a.part_out, lane);
    if (lane == 0) ws.hdr->pad = kWalkDone;
}

}  // namespace bzip2
}  // namespace swc
#endif
