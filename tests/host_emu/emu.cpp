// emu.cpp -- TEST INFRASTRUCTURE.  Builds the per-lane device decoders of swcompression_amd/csrc for
// the HOST (g++ -DSWC_HOST_EMULATION) and runs the lanes of each wave one after another, with the
// same wave-interleaved table layout the kernels use in LDS.  Lets the CPU-only test tier exercise
// the exact source the gfx950 kernels are compiled from against the oracle.  Never shipped, never
// linked into libswc_hip.so.
#include <vector>
#include <cstring>
#include "../../swcompression_amd/csrc/inflate_lane.h"
#include "../../swcompression_amd/csrc/lz4_lane.h"
#include "../../swcompression_amd/csrc/lz4_wave.h"
#include "../../swcompression_amd/csrc/lzma_wave.h"
#include "../../swcompression_amd/csrc/bzip2_block.h"

// Deflate: phase 1 lane by lane (wave-interleaved tables), then phase 2 with a one-thread "workgroup".
extern "C" void emu_inflate(swc::Job* jobs, size_t n) {
    std::vector<uint32_t> lds(swc::inflate::kWordsPerLane * swc::kWave);
    alignas(16) static swc::lzr::Lds<1> rl;
    for (size_t g = 0; g < n; g++) {
        int lane = (int)(g % swc::kWave);
        if (lane == 0) std::fill(lds.begin(), lds.end(), 0xDEADBEEFu);  // LDS is uninitialised on device
        swc::LaneLds l{lds.data() + lane, swc::kWave};
        size_t wsb = swc::lzr::ws_bytes_per_job(jobs[g].out_cap);
        std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
        swc::inflate::inflate_job(jobs[g], l, ws.data(), wsb);
        std::memset(&rl, 0xEE, sizeof rl);
        swc::lzr::resolve_job<1>(jobs[g], ws.data(), wsb, &rl, 0);
    }
}

// LZ4: blocks with a dictionary prefix on the lane decoder, the others through the two-phase path (parse with a
// one-lane "wavefront", resolve with a one-thread "workgroup").
extern "C" void emu_lz4_block(swc::Job* jobs, size_t n) {
    static swc::lzr::Lds<1, swc::lz4w::kKeep, swc::lz4w::kWin> rl;
    for (size_t g = 0; g < n; g++) {
        if (jobs[g].dict) { swc::lz4::lz4_block_job(jobs[g]); continue; }
        size_t wsb = swc::lzr::ws_bytes_per_job(jobs[g].out_cap);
        std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
        static uint32_t rbuf[swc::lz4w::kRecBuf];
        static uint8_t lbuf[swc::lz4w::kLitStage + 32 + 64];
        alignas(16) static uint8_t iw[swc::lz4w::kInWin + 16];
        swc::lz4w::lz4_parse_job<1>(jobs[g], ws.data(), wsb, 0, rbuf, lbuf, iw);
        std::memset(&rl, 0xEE, sizeof rl);
        swc::lzr::resolve_job<1, swc::lz4w::kKeep, swc::lz4w::kWin>(jobs[g], ws.data(), wsb, &rl, 0);
    }
}

// LZMA: the wave-uniform decode chain is run as a single logical lane (WAVE = 1); the literal-coder
// spill (lc+lp > 4) is always available, as the single-shot C ABI guarantees on the device.
extern "C" void emu_lzma(swc::Job* jobs, size_t n, int is_lzma2) {
    std::vector<uint16_t> probs(swc::lzma::kProbCells + 8);
    std::vector<uint16_t> spill((size_t)0x300 << 12);
    for (size_t g = 0; g < n; g++) {
        std::fill(probs.begin(), probs.end(), (uint16_t)0xBEEF);
        swc::lzma::lzma_job<1>(jobs[g], is_lzma2 != 0, probs.data(), spill.data(), 0);
    }
}

// BZip2: the three stages run back to back for each job, single logical lane (WAVE = 1).
extern "C" void emu_bzip2_block(swc::Job* jobs, size_t n, size_t lcap) {
    using namespace swc::bzip2;
    std::vector<uint8_t> ws(ws_bytes_per_job(lcap) + 64);
    std::vector<uint32_t> cnt(256), crc(256);
    for (uint32_t i = 0; i < 256; i++) crc[i] = crc_table_entry(i);
    Stage1Lds lds;
    for (size_t g = 0; g < n; g++) {
        std::memset(&lds, 0xEE, sizeof lds);
        std::fill(ws.begin(), ws.end(), (uint8_t)0xCD);
        Workspace w = carve(ws.data(), 0, lcap);
        stage1_job<1>(jobs[g], &lds, w, 0);
        stage2_job<1>(w, cnt.data(), 0);
        stage3_job(jobs[g], w, crc.data());
    }
}

// Debug/analysis helper (tools/analyze_records.py): phase 1 only, returns the record list of one stream.
extern "C" size_t emu_inflate_records(const uint8_t* in, size_t in_len, uint8_t* out, size_t cap, uint32_t* recs_out, size_t max_out) {
    std::vector<uint32_t> lds(swc::inflate::kWordsPerLane * swc::kWave, 0xDEADBEEFu);
    swc::LaneLds l{lds.data(), swc::kWave};
    size_t wsb = swc::lzr::ws_bytes_per_job(cap);
    std::vector<uint8_t> ws(wsb + 16, (uint8_t)0xCD);
    swc::Job j{};
    j.in = in; j.in_len = in_len; j.out = out; j.out_cap = cap;
    swc::inflate::inflate_job(j, l, ws.data(), wsb);
    uint32_t n = ((swc::lzr::StreamHeader*)ws.data())->nrec;
    const uint32_t* r = (const uint32_t*)(ws.data() + sizeof(swc::lzr::StreamHeader));
    for (size_t i = 0; i < n && i < max_out; i++) recs_out[i] = r[i];
    return n;
}
