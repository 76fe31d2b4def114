// emu.cpp -- TEST INFRASTRUCTURE.  Builds the device decoders of swcompression_amd/csrc for the HOST
// (g++ -DSWC_HOST_EMULATION): the uniform parts run once, the threads of every SIMT region (csrc/simt.h) one
// after another in a selectable order.  Lets the CPU-only test tier exercise
// the exact source the gfx950 kernels are compiled from against the oracle.  Never shipped, never
// linked into libswc_hip.so.
#include <vector>
#include <cstring>
#include <algorithm>
#include <memory>
#include "../../swcompression_amd/csrc/inflate_lane.h"
#include "../../swcompression_amd/csrc/inflate_sync.h"
#include "../../swcompression_amd/csrc/lz4_lane.h"
#include "../../swcompression_amd/csrc/lz4_wave.h"
#include "../../swcompression_amd/csrc/lz4_comp.h"
#include "../../swcompression_amd/csrc/deflate_comp.h"
#include "../../swcompression_amd/csrc/lz_copy.h"
#include "../../swcompression_amd/csrc/lzma_wave.h"
#include "../../swcompression_amd/csrc/bzip2_block.h"
#include "../../swcompression_amd/csrc/bzip2_team.h"
#include "../../swcompression_amd/csrc/bzip2_comp.h"
#include "../../swcompression_amd/csrc/crc32_group.h"

extern "C" void emu_set_order(int o) { swc::simt::g_order = o; }
// phase 2: 1 = the record-granular copier of lz_copy.h as the library ships it (Deflate: 5 KiB window, LZ4: 8 KiB), 3 = the 5 KiB
// window for both, 2 = the 8 KiB window for both, 0 = the byte-cell resolver of lz_resolve.h
static int g_copier = 1;
extern "C" void emu_set_copier(int on) { g_copier = on; }
// (the two window configurations the library ships: kernels.hip)
template <typename CFG, int RM = 0>
static void emu_copy(swc::Job& job, const uint8_t* ws, size_t wsb) {
    alignas(16) static swc::lzc::Lds<CFG::kWin> cl;
    std::memset(&cl, 0xEE, sizeof cl);
    swc::lzc::copy_job<CFG, RM>(job, ws, wsb, &cl);
}
// how the LZ4 parse tells the copier where the literals lie (kernels.hip: SWC_LZ4_RECORD_MODE): 1 = eight-byte records, 2 = derived + anchors
static int g_lz4_mode = 2;
extern "C" void emu_set_lz4_record_mode(int m) { g_lz4_mode = m; }
// (Deflate: four-byte records + the dense literal stream; LZ4: eight-byte records whose literals stay in the block -- R8)
static void emu_copy_any(int deflate, swc::Job& job, const uint8_t* ws, size_t wsb) {
    if (deflate) {
        if (g_copier == 2) emu_copy<swc::lzc::CfgLz4>(job, ws, wsb);
        else emu_copy<swc::lzc::CfgDeflate>(job, ws, wsb);
    } else {
        if (g_lz4_mode == 1) {
            if (g_copier == 3) emu_copy<swc::lzc::CfgDeflate, 1>(job, ws, wsb);
            else emu_copy<swc::lzc::CfgLz4, 1>(job, ws, wsb);
        } else {
            if (g_copier == 3) emu_copy<swc::lzc::CfgDeflate, 2>(job, ws, wsb);
            else emu_copy<swc::lzc::CfgLz4, 2>(job, ws, wsb);
        }
    }
}

// Deflate, one stream per wavefront with 64 sub-chunks decoded at once (inflate_sync.h): the uniform parts run once, the
// 64 lanes of every parallel region one after another (csrc/simt.h).
// phase 1 with a team of wavefronts per stream (inflate_sync.h; kernels.hip: launches of few streams): 0 = one wavefront
static int g_team = 0;
extern "C" void emu_set_deflate_team(int on) { g_team = on; }
extern "C" uint64_t emu_team_adopted(int reset) { const uint64_t v = swc::inflate::g_team_adopted; if (reset) swc::inflate::g_team_adopted = 0; return v; }
extern "C" void emu_inflate_sync(swc::Job* jobs, size_t n) {
    alignas(16) static swc::inflate::SyncLds sl;
    alignas(16) static swc::inflate::SyncLds tl[swc::inflate::kTeamWaves];
    alignas(16) static swc::inflate::TeamShared tsh;
    alignas(16) static swc::lzr::Lds<512, 16> rl;
    for (size_t g = 0; g < n; g++) {
        std::memset(&sl, 0xEE, sizeof sl);
        size_t wsb = swc::lzr::ws_bytes_per_job(jobs[g].out_cap);
        std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
        if (g_team) {
            std::memset(tl, 0xEE, sizeof tl);
            std::memset(&tsh, 0xEE, sizeof tsh);
            for (auto& h : tsh.hgen) h = 0;
            tsh.cmd = 0;
            std::vector<uint8_t> rows((swc::inflate::kTeamWaves - 1) * swc::inflate::kTeamProvBytes + 16, (uint8_t)0xCD);
            swc::inflate::Team tm;
            tm.sh = &tsh; tm.lds = tl; tm.scratch = rows.data(); tm.helpers = swc::inflate::kTeamWaves - 1; tm.gen = 0;
            swc::inflate::inflate_sync_job<true>(jobs[g], &tl[0], ws.data(), wsb, 0, 1, nullptr, &tm);
        } else
        swc::inflate::inflate_sync_job(jobs[g], &sl, ws.data(), wsb, 0, 1);
        if (g_copier) {
            emu_copy_any(1, jobs[g], ws.data(), wsb);
            continue;
        }
        std::memset(&rl, 0xEE, sizeof rl);
        swc::lzr::resolve_job<512, 16, 32768>(jobs[g], ws.data(), wsb, &rl);
    }
}

extern "C" void emu_lz4_stats(uint64_t* out, int reset) {
    for (int i = 0; i < 8; i++) { out[i] = swc::lz4w::g_lz4_stats[i]; if (reset) swc::lz4w::g_lz4_stats[i] = 0; }
}
extern "C" void emu_sync_wave(uint64_t* out) { for (int i = 0; i < 4; i++) { out[i] = swc::inflate::g_sync_wave[i]; swc::inflate::g_sync_wave[i] = 0; } }
extern "C" void emu_sync_stats(uint64_t* out, int reset) {
    for (int i = 0; i < 8; i++) { out[i] = swc::inflate::g_sync_stats[i]; if (reset) swc::inflate::g_sync_stats[i] = 0; }
}

// LZ4: blocks with a dictionary prefix on the lane decoder, the others through the two-phase path (parse with a
// one-lane "wavefront", resolve with a one-thread "workgroup").
extern "C" void emu_lz4_block(swc::Job* jobs, size_t n) {
    alignas(16) static swc::lzr::Lds<swc::lz4w::kResolveThreads, swc::lz4w::kRingLog2> rl;
    for (size_t g = 0; g < n; g++) {
        if (jobs[g].dict) { swc::lz4::lz4_block_job(jobs[g]); continue; }
        size_t wsb = swc::lzr::ws_bytes_per_job(jobs[g].out_cap);
        std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
        alignas(16) static uint8_t stage[swc::lz4w::kStageLds];
        std::memset(stage, 0xEE, sizeof stage);
        if (g_copier) {
            if (g_lz4_mode == 1) swc::lz4w::lz4_parse_job<1, 1>(jobs[g], ws.data(), wsb, 0, stage);
            else swc::lz4w::lz4_parse_job<1, 2>(jobs[g], ws.data(), wsb, 0, stage);
            emu_copy_any(0, jobs[g], ws.data(), wsb);
            continue;
        }
        swc::lz4w::lz4_parse_job<1>(jobs[g], ws.data(), wsb, 0, stage);
        std::memset(&rl, 0xEE, sizeof rl);
        swc::lzr::resolve_job<swc::lz4w::kResolveThreads, swc::lz4w::kRingLog2, swc::lz4w::kKeep, true>(jobs[g], ws.data(), wsb, &rl);
    }
}

// LZ4 block compression (lz4_comp.h): job.in = prefix ++ block, job.dict_len = length of the prefix
extern "C" void emu_lz4_compress(swc::Job* jobs, size_t n) {
    alignas(16) static uint16_t table[swc::lz4c::kHashSize];
    for (size_t g = 0; g < n; g++) {
        std::memset(table, 0xEE, sizeof table);
        swc::lz4c::lz4_compress_job<64>(jobs[g], table);
    }
}

// Deflate compression (deflate_comp.h): job.in = the buffer
extern "C" void emu_deflate_compress(swc::Job* jobs, size_t n) {
    alignas(16) static swc::defc::Lds lds;
    for (size_t g = 0; g < n; g++) {
        std::memset(&lds, 0xEE, sizeof lds);
        swc::defc::deflate_compress_job<64>(jobs[g], &lds);
    }
}

// LZMA: the wave-uniform decode chain is run as a single logical lane (WAVE = 1); the literal-coder
// spill (lc+lp > 4) is always available, as the single-shot C ABI guarantees on the device.
// mode 0: every literal coder in LDS up to lc + lp = 4, larger models cell by cell from the spill (the kernel without a
// workspace, and the old spill path); mode 1: LDS as a cache of kCoderSlots literal coders, long-length trees in the spill (the kernel with a workspace).
extern "C" void emu_lzma_mode(swc::Job* jobs, size_t n, int is_lzma2, int mode) {
    std::vector<uint16_t> probs(swc::lzma::kProbCells + 8);
    std::vector<uint16_t> spill(((size_t)0x300 << 12) + 512);   // (+ the two `high` length trees of the cache mode)
    for (size_t g = 0; g < n; g++) {
        std::fill(probs.begin(), probs.end(), (uint16_t)0xBEEF);
        std::fill(spill.begin(), spill.end(), (uint16_t)0xDEAD);
        swc::lzma::lzma_job<1>(jobs[g], is_lzma2 != 0, probs.data(), spill.data(), 0, swc::lzma::kMaxLdsLitBits, nullptr, mode == 1);
    }
}
extern "C" void emu_lzma(swc::Job* jobs, size_t n, int is_lzma2) { emu_lzma_mode(jobs, n, is_lzma2, 0); }

// BZip2: the three stages run back to back for each job, single logical lane (WAVE = 1).
extern "C" void emu_bzip2_block(swc::Job* jobs, size_t n, size_t lcap) {
    using namespace swc::bzip2;
    std::vector<uint8_t> ws(ws_bytes_per_job(lcap) + 64);
    std::vector<uint32_t> cnt(256 + 256);
    Stage1Lds lds;
    Stage3Lds lds3;
    static swc::crc::Lds<1, uint32_t> crc_lds;
    for (size_t g = 0; g < n; g++) {
        std::memset(&lds, 0xEE, sizeof lds);
        std::fill(ws.begin(), ws.end(), (uint8_t)0xCD);
        Workspace w = carve(ws.data(), 0, lcap);
        stage1_job<1>(jobs[g], &lds, w, 0);
        stage2_job(w, cnt.data());
        std::memset(&lds3, 0xEE, sizeof lds3);
        stage3_walk_job<1>(jobs[g], w, &lds3, 0);
        if (stage3_expand_needed(w)) stage3_expand_job(jobs[g], w);
        if (jobs[g].status == SWC_OK)
            stage3_check_crc(jobs[g], swc::crc::crc_group<1, uint32_t, true>(jobs[g].out, jobs[g].out_len, &crc_lds, 0));
    }
}

// BZip2 with stage 3a as kernels of its own (bzip2_team.h): stage 1 + 2 of ALL jobs, then the segment counts and their prefixes
// per team, the walk (every team's tickets drawn by one thread; `start_team`: the team whose thread goes first and, with the
// work stealing, through the others' tickets too when `one_thread` is set), the finish per job, the serial fallback, the CRC.
// The workspace geometry of stage 3 for a block of n bytes inside a workspace cut for lcap: what the two walks need against what
// the layout gives them (tests/test_lane_emulation_bzip2.py walks through many (lcap, n) pairs).
extern "C" void emu_bzip2_layout(size_t lcap, uint32_t n, uint64_t* out) {
    using namespace swc::bzip2;
    Cut c;
    c.set(n, n / 3u + 1u < n ? n / 3u + 1u : 0u);                       // an origin pointer off the marks where there is room
    const uint32_t m = seg_mbits(n), regs = (n + (1u << m) - 1u) >> m;
    out[0] = team_seg_slots(lcap);                                      // slots of the per-segment arrays
    out[1] = c.segs;                                                    // segments the team walk cuts the block into
    out[2] = segbuf_bytes(lcap);                                        // bytes of segment buffers
    out[3] = (uint64_t)(c.regs + 1u) * c.cap;                           // ... the team walk needs
    out[4] = (uint64_t)(regs + 1u) * ((uint64_t)kSegCapFactor << m);    // ... the walk inside the block's wavefront needs
    out[5] = seg_info_bytes(lcap);
    out[6] = (2ull * (kSegs + 1) > 4ull * out[0] ? 2ull * (kSegs + 1) : 4ull * out[0]) * 4 + kTeamWords * 4;   // ... and what must fit in it
    out[7] = ws_bytes_per_job(lcap);
}
static uint64_t g_team_finished = 0;   // blocks whose output came from team_finish, not from the serial fallback
extern "C" uint64_t emu_bzip2_team_finished(int reset) { const uint64_t v = g_team_finished; if (reset) g_team_finished = 0; return v; }
extern "C" void emu_bzip2_block_team(swc::Job* all_jobs, size_t n_all, size_t lcap, int start_team, int one_thread) {
    using namespace swc::bzip2;
    const size_t per = ws_bytes_per_job(lcap), kAtOnce = 24;          // (the workspaces of a launch exist side by side: 24 jobs at a time)
    std::vector<uint8_t> ws(per * std::min(n_all, kAtOnce) + 64);
    std::vector<uint32_t> cnt(256 + 256);
    Stage1Lds lds;
    static FinishLds fl;
    static swc::crc::Lds<1, uint32_t> crc_lds;
    for (size_t j0 = 0; j0 < n_all; j0 += kAtOnce) {
        swc::Job* jobs = all_jobs + j0;
        const size_t n = std::min(kAtOnce, n_all - j0);
        std::fill(ws.begin(), ws.end(), (uint8_t)0xCD);
        for (size_t g = 0; g < n; g++) {
            std::memset(&lds, 0xEE, sizeof lds);
            Workspace w = carve(ws.data(), g, lcap);
            stage1_job<1>(jobs[g], &lds, w, 0);
            stage2_job(w, cnt.data());
        }
        for (uint32_t t = 0; t < kTeams; t++) team_prep<1>(ws.data(), lcap, (uint32_t)n, t, 0);
        for (uint32_t d = 0; d < (one_thread ? 1u : kTeams); d++) team_walk(ws.data(), lcap, (uint32_t)n, ((uint32_t)start_team + d) % kTeams);
        for (size_t g = 0; g < n; g++) {
            Workspace w = carve(ws.data(), g, lcap);
            std::memset(&fl, 0xEE, sizeof fl);
            team_finish<1>(jobs[g], w, &fl, 0);
            if (stage3_expand_needed(w)) stage3_expand_job(jobs[g], w);
            else g_team_finished++;
            if (jobs[g].status == SWC_OK)
                stage3_check_crc(jobs[g], swc::crc::crc_group<1, uint32_t, true>(jobs[g].out, jobs[g].out_len, &crc_lds, 0));
        }
    }
}

// Phase 2 alone (tests/test_lz_copy_records.py): a record list and a literal stream made by the test, laid out as phase 1 lays
// them out in a workspace area, through lz_copy.h (copier != 0) or lz_resolve.h; `out` may sit at any alignment.
extern "C" void emu_copy_records(const uint32_t* recs, uint32_t nrec, const uint8_t* lits, size_t nlit, uint8_t* out, size_t cap, size_t out_len, int copier) {
    const size_t wsb = swc::lzr::ws_bytes_per_job(cap);
    std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
    swc::lzr::StreamHeader* h = (swc::lzr::StreamHeader*)ws.data();
    h->nrec = nrec;
    h->nlit = nlit;
    std::memcpy(ws.data() + sizeof(swc::lzr::StreamHeader), recs, 4 * (size_t)nrec);
    std::memcpy(ws.data() + swc::lzr::lit_offset(wsb, cap), lits, nlit);
    swc::Job j{};
    j.out = out; j.out_cap = cap; j.out_len = out_len;
    if (copier == 3) emu_copy<swc::lzc::CfgDeflate>(j, ws.data(), wsb);
    else if (copier) emu_copy<swc::lzc::CfgLz4>(j, ws.data(), wsb);
    else {
        alignas(16) static swc::lzr::Lds<512, 16> rl;
        std::memset(&rl, 0xEE, sizeof rl);
        swc::lzr::resolve_job<512, 16, 32768, true>(j, ws.data(), wsb, &rl);
    }
}

// The same with EIGHT-byte records (record | literal offset << 32) whose literals lie in `in` (the LZ4 path): `in` is copied
// into a buffer of exactly in_len bytes, so that a read past the block shows under AddressSanitizer.
extern "C" void emu_copy_records8(const uint32_t* recs2, uint32_t nrec, const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, size_t out_len, int copier) {
    const size_t wsb = swc::lzr::ws_bytes_per_job(cap);
    std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
    swc::lzr::StreamHeader* h = (swc::lzr::StreamHeader*)ws.data();
    h->nrec = nrec;
    h->nlit = 0;
    std::memcpy(ws.data() + sizeof(swc::lzr::StreamHeader), recs2, 8 * (size_t)nrec);
    std::unique_ptr<uint8_t[]> exact(new uint8_t[in_len ? in_len : 1]);
    std::memcpy(exact.get(), in, in_len);
    swc::Job j{};
    j.in = exact.get(); j.in_len = in_len;
    j.out = out; j.out_cap = cap; j.out_len = out_len;
    if (copier == 3) emu_copy<swc::lzc::CfgDeflate, 1>(j, ws.data(), wsb);
    else emu_copy<swc::lzc::CfgLz4, 1>(j, ws.data(), wsb);
}
// ... and with FOUR-byte records whose literal offsets are derived (lz_copy.h RM == 2): `anchors` = nanc (record index, S) pairs
extern "C" void emu_copy_records4(const uint32_t* recs, uint32_t nrec, const uint32_t* anchors, uint32_t nanc, const uint8_t* in, size_t in_len,
                                  uint8_t* out, size_t cap, size_t out_len, int copier) {
    const size_t wsb = swc::lzr::ws_bytes_per_job(cap);
    std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
    swc::lzr::StreamHeader* h = (swc::lzr::StreamHeader*)ws.data();
    h->nrec = nrec;
    h->pad0 = nanc;
    h->nlit = 0;
    std::memcpy(ws.data() + sizeof(swc::lzr::StreamHeader), recs, 4 * (size_t)nrec);
    std::memcpy(ws.data() + swc::lzr::lit_offset(wsb, cap), anchors, 8 * (size_t)nanc);
    std::unique_ptr<uint8_t[]> exact(new uint8_t[in_len ? in_len : 1]);
    std::memcpy(exact.get(), in, in_len);
    swc::Job j{};
    j.in = exact.get(); j.in_len = in_len;
    j.out = out; j.out_cap = cap; j.out_len = out_len;
    if (copier == 3) emu_copy<swc::lzc::CfgDeflate, 2>(j, ws.data(), wsb);
    else emu_copy<swc::lzc::CfgLz4, 2>(j, ws.data(), wsb);
}

// The LZ4 parse alone in the derived-offset form (record mode 2): how many records and how many anchors a block needs
extern "C" void emu_lz4_parse_counts(const uint8_t* in, size_t in_len, size_t cap, uint32_t* nrec, uint32_t* nanc) {
    size_t wsb = swc::lzr::ws_bytes_per_job(cap);
    std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD), out(cap + 16);
    alignas(16) static uint8_t stage[swc::lz4w::kStageLds];
    swc::Job j{};
    j.in = in; j.in_len = in_len; j.out = out.data(); j.out_cap = cap;
    swc::lz4w::lz4_parse_job<1, 2>(j, ws.data(), wsb, 0, stage);
    const swc::lzr::StreamHeader* h = (const swc::lzr::StreamHeader*)ws.data();
    *nrec = h->nrec;
    *nanc = h->pad0;
}

// Debug/analysis helper (tools/analyze_records.py): phase 1 only, returns the record list of one stream.
extern "C" size_t emu_inflate_records(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, uint32_t* recs_out, size_t max_out) {
    alignas(16) static swc::inflate::SyncLds sl;
    size_t wsb = swc::lzr::ws_bytes_per_job(cap);
    std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
    swc::Job j{};
    j.in = in; j.in_len = in_len; j.out = out; j.out_cap = cap;
    swc::inflate::inflate_sync_job(j, &sl, ws.data(), wsb, 0, 1);
    uint32_t n = ((swc::lzr::StreamHeader*)ws.data())->nrec;
    const uint32_t* r = (const uint32_t*)(ws.data() + sizeof(swc::lzr::StreamHeader));
    for (size_t i = 0; i < n && i < max_out; i++) recs_out[i] = r[i];
    return n;
}

// The same for an LZ4 block (tools/analyze_records.py lz4)
extern "C" size_t emu_lz4_records(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, uint32_t* recs_out, size_t max_out) {
    size_t wsb = swc::lzr::ws_bytes_per_job(cap);
    std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
    alignas(16) static uint8_t stage[swc::lz4w::kStageLds];
    swc::Job j{};
    j.in = in; j.in_len = in_len; j.out = out; j.out_cap = cap;
    swc::lz4w::lz4_parse_job<1>(j, ws.data(), wsb, 0, stage);
    uint32_t n = ((swc::lzr::StreamHeader*)ws.data())->nrec;
    const uint32_t* r = (const uint32_t*)(ws.data() + sizeof(swc::lzr::StreamHeader));
    for (size_t i = 0; i < n && i < max_out; i++) recs_out[i] = r[i];
    return n;
}

// ---- checksums: the group kernels run with T real host threads and a pthread barrier behind swc::group_sync ----------
#include <thread>
#include <pthread.h>
#include "../../swcompression_amd/csrc/crc32_group.h"
#include "../../swcompression_amd/csrc/checksum_group.h"

namespace {
pthread_barrier_t* g_barrier;
void barrier_wait() { pthread_barrier_wait(g_barrier); }

template <int T, typename F>
uint64_t run_group(F body) {
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, T);
    g_barrier = &bar;
    uint64_t result = 0;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
            swc::emu_group_sync = barrier_wait;
            uint64_t r = body(t);
            if (t == 0) result = r;
        });
    for (auto& x : th) x.join();
    pthread_barrier_destroy(&bar);
    return result;
}
}  // namespace

// the wave-per-stream CRC-32 (crc32_wave.h): constants built once, the 64 lanes run one after the other
#include "../../swcompression_amd/csrc/crc32_wave.h"
extern "C" uint32_t emu_crc32_wave(const uint8_t* p, size_t n) {
    static swc::crcw::WaveConsts consts;
    static bool built = false;
    if (!built) { swc::crcw::build_consts<1>(&consts, 0); built = true; }
    return swc::crcw::crc32_wave(p, n, &consts);
}

// kind as swc_checksum (include/swc_hip.h); T = 64 emulated threads per group
extern "C" uint64_t emu_checksum(int kind, const uint8_t* p, size_t n) {
    using namespace swc;
    constexpr int T = 64;
    switch (kind) {
        case 1: { static crc::Lds<T, uint32_t> l; return run_group<T>([&](int t) { return (uint64_t)crc::crc_group<T, uint32_t, false>(p, n, &l, t); }); }
        case 2: { static sums::AdlerLds<T> l; return run_group<T>([&](int t) { return (uint64_t)sums::adler32_group<T>(p, n, &l, t); }); }
        case 3: { static crc::Lds<T, uint64_t> l; return run_group<T>([&](int t) { return (uint64_t)crc::crc_group<T, uint64_t, false>(p, n, &l, t); }); }
        case 4: { static crc::Lds<T, uint32_t> l; return run_group<T>([&](int t) { return (uint64_t)crc::crc_group<T, uint32_t, true>(p, n, &l, t); }); }
        case 5: {
            uint32_t acc[4];
            const uint32_t seed = 0;
            for (int j = 0; j < 4; j++) {
                const uint32_t init = j == 0 ? seed + sums::kP1 + sums::kP2 : j == 1 ? seed + sums::kP2 : j == 2 ? seed : seed - sums::kP1;
                acc[j] = sums::xxh32_lane(p, n / 16, j, init);
            }
            return sums::xxh32_quad(p, n, seed, 0, [&](uint32_t, int k) { return acc[k]; });
        }
    }
    return 0;
}

// Delta filter: the group kernel with T real host threads (same barrier plumbing as the checksums)
#include "../../swcompression_amd/csrc/delta_group.h"
extern "C" void emu_delta(const uint8_t* in, uint8_t* out, size_t n, unsigned distance) {
    constexpr int T = 256;
    static swc::delta::Lds<T> l;
    run_group<T>([&](int t) { swc::delta::delta_group<T>(in, out, n, distance, &l, t); return (uint64_t)0; });
}

// BZip2 compression (bzip2_comp.h): the executor of the emulation -- "device memory" is host memory, a kernel is a loop over
// its elements / blocks, std::sort and two serial scans stand in for the three rocPRIM calls of the device executor.
namespace {
struct EmuBz2Exec {
    std::vector<std::unique_ptr<uint8_t[]>> mem;
    void begin_chunk() {}
    void mark(const char*) {}
    uint8_t* result_alloc(size_t n) { uint8_t* q = (uint8_t*)malloc(n ? n : 1); if (q) std::memset(q, 0xEE, n); return q; }
    void result_free(uint8_t* q) { free(q); }
    void note(const char*, uint32_t, uint32_t) {}
    void end_chunk() { mem.clear(); }
    void* alloc(size_t n) { mem.emplace_back(new uint8_t[n ? n : 1]); std::memset(mem.back().get(), 0xEE, n); return mem.back().get(); }   // exact: the ASAN build sees every overrun
    void upload(void* d, const void* h, size_t n) { std::memcpy(d, h, n); }
    void download(void* h, const void* d, size_t n) { std::memcpy(h, d, n); }
    void zero(void* d, size_t n) { std::memset(d, 0, n); }
    template <class F> void each(uint32_t m, const F& f) {
        if (swc::simt::g_order == 1) for (uint32_t i = m; i-- > 0;) f(i);
        else for (uint32_t i = 0; i < m; i++) f(i);
    }
    template <class F> void per_block(uint32_t nb, const F& f) {
        for (uint32_t b = 0; b < nb; b++) {
            alignas(16) static typename F::Lds lds;
            std::memset(&lds, 0xEE, sizeof lds);
            f.template run<64>(b, &lds);
        }
    }
    int sort_pairs(uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, uint32_t m, int bits) {
        const uint64_t mask = bits >= 64 ? ~0ull : (1ull << bits) - 1ull;
        std::vector<uint32_t> idx(m);
        for (uint32_t i = 0; i < m; i++) { idx[i] = i; if (kin[i] & ~mask) return 1; }   // a key wider than announced: the radix sort would drop bits
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return kin[a] < kin[b]; });
        for (uint32_t i = 0; i < m; i++) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
        return 0;
    }
    int scan_max(uint32_t* x, uint32_t m) { uint32_t a = 0; for (uint32_t i = 0; i < m; i++) { a = x[i] > a ? x[i] : a; x[i] = a; } return 0; }
    int scan_sum(const uint32_t* in, uint32_t* out, uint32_t m) { uint32_t a = 0; for (uint32_t i = 0; i < m; i++) { out[i] = a; a += in[i]; } return 0; }
    int block_crcs(const uint8_t* raw, const uint32_t* off, uint32_t nb, uint32_t* crcs) {
        for (uint32_t b = 0; b < nb; b++) {   // CheckSums.bzip2crc32: polynomial 0x04C11DB7, MSB first
            uint32_t c = 0xFFFFFFFFu;
            for (uint32_t i = off[b]; i < off[b + 1]; i++) {
                c ^= (uint32_t)raw[i] << 24;
                for (int k = 0; k < 8; k++) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : c << 1;
            }
            crcs[b] = ~c;
        }
        return 0;
    }
};
}  // namespace
// returns the status; *out_len = bytes needed (the stream is copied only if it fits `cap`)
extern "C" int emu_bzip2_compress(const uint8_t* data, size_t len, int level, uint8_t* out, size_t cap, size_t* out_len) {
    EmuBz2Exec x;
    uint8_t* res = nullptr;
    size_t n = 0;
    const int st = swc::bz2c::compress_stream(x, data, len, level, &res, &n);
    *out_len = n;
    if (st == 0 && n <= cap) std::memcpy(out, res, n);
    free(res);
    return st;
}
