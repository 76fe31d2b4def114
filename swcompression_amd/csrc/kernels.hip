// kernels.hip -- gfx950 kernels of the many-stream decode engine and their launchers.
//
// Every unit (a Deflate stream, an LZ4 block, a bzip2 block, an LZMA stream) is independent; the chip is filled by units, and
// each codec maps a unit to the piece of the machine its serial part needs:
//   Deflate   phase 1 one stream per WAVEFRONT, 64 sub-chunks decoded at once (self-synchronising Huffman decode);
//             phase 2 (LZ77 resolve) one stream per WORKGROUP, byte cells in LDS   inflate_sync.h, inflate_lane.h, lz_resolve.h
//   LZ4       parse one block per wavefront (64 sub-chunks at once), then the same resolve kernel; dictionary blocks per lane
//                                                                                                lz4_wave.h, lz4_lane.h
//   LZMA      one stream per wavefront, probability model in LDS                                               lzma_wave.h
//   BZip2     Huffman + MTF per wavefront, counting-sort scatter per wavefront, inverse BWT by all lanes of a wavefront over
//             a cut cycle, block CRC per workgroup                                                           bzip2_block.h
//   checksums / Delta filter   one stream per workgroup                     crc32_group.h, checksum_group.h, delta_group.h
// No MFMA (no dense contraction anywhere on this path), no inter-workgroup communication.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <mutex>
#include <atomic>
#include <vector>
#include "swc_common.h"
#include "inflate_lane.h"
#include "inflate_sync.h"
#include "lz_resolve.h"
#include "lz_copy.h"
#include "lz4_lane.h"
#include "lz4_wave.h"
#include "lz4_comp.h"
#include "deflate_comp.h"
#include "lzma_wave.h"
#include "bzip2_block.h"
#include "bzip2_team.h"
#include "crc32_group.h"
#include "crc32_wave.h"
#include "checksum_group.h"
#include "delta_group.h"
#include "launch.h"

namespace swc {

// A job's area of the workspace: equal strides, or -- `ws_off` given -- prefix-summed per-job sizes (ws_off[n] = total),
// so that one large unit among many small ones does not size everybody's area.
struct WsMap {
    uint8_t* base;
    size_t stride;
    const uint64_t* off;
    __device__ uint8_t* area(uint32_t g) const { return base ? base + (off ? (size_t)off[g] : (size_t)g * stride) : nullptr; }
    __device__ size_t bytes(uint32_t g) const { return off ? (size_t)(off[g + 1] - off[g]) : stride; }
};

// Which job does workgroup `b` of an n-job launch take?  The hardware hands consecutive workgroups to the eight XCDs in turn
// (workgroup b runs on XCD b % 8), so a job list whose cost has a period that divides 8 -- every fourth unit an incompressible
// one, say -- would put all the expensive jobs on two XCDs and the launch would last as long as if every job were expensive
// (measured: 192 text + 64 P-mix LZ4 blocks interleaved 3 : 1 took exactly the time of 256 P-mix blocks).  Here XCD x works
// through the contiguous range [x n/8, (x + 1) n/8) of the list instead, in order: any eighth of the list costs about the same.
__device__ __forceinline__ uint32_t xcd_job(uint32_t b, uint32_t n) {
    const uint32_t per = n >> 3;
    return b < (per << 3) ? (b & 7u) * per + (b >> 3) : b;
}

// ---- launch order: the longest jobs first ------------------------------------------------------------------------------
// A job is one workgroup and lasts as long as its unit is big: 8,192 LZ4 blocks of 4 MiB are 1.7 rounds of the parse kernel's
// resident waves, and when the incompressible blocks of the batch (twice the time of a text block) happen to start in the
// second round, the launch ends one long block after everybody else has finished (measured: 73 ms for 34 ms of work).  The
// hardware hands out workgroups in index order, so the jobs are ORDERED by size class -- compressed bytes, in steps of an
// eighth of an octave -- largest first: a counting sort in three small kernels (histogram, prefix, scatter; the order inside a
// class is whatever the atomics make it).  perm[b] is the job of workgroup b.  Batches of fewer than kOrderMin jobs keep the
// plain XCD-aware order.
constexpr int kOrderClasses = 256;
constexpr size_t kOrderMin = 2048;
__device__ __forceinline__ uint32_t order_class(uint64_t len) {
    const uint32_t v = len > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len | 8u;
    const uint32_t e = 31u - (uint32_t)__clz((int)v);
    return (kOrderClasses - 1) - (e * 8u + ((v >> (e - 3u)) & 7u));   // class 0 = the largest
}
// (one LDS histogram per workgroup and one global atomic per class it holds: 100,000 members of ONE size class otherwise meet at
// one counter -- 2 ms for the two kernels, a tenth of the launch they were meant to shorten)
__global__ __launch_bounds__(256) void swc_order_hist_kernel(const Job* __restrict__ jobs, uint32_t n, uint32_t* __restrict__ hist) {
    __shared__ uint32_t cnt[kOrderClasses];
    const uint32_t t = threadIdx.x, g = blockIdx.x * 256u + t;
    cnt[t] = 0;
    __syncthreads();
    if (g < n) atomicAdd(&cnt[order_class(jobs[g].in_len)], 1u);
    __syncthreads();
    if (cnt[t]) atomicAdd(&hist[t], cnt[t]);
}
__global__ __launch_bounds__(kOrderClasses) void swc_order_prefix_kernel(uint32_t* __restrict__ hist) {   // hist[c] -> first slot of class c
    __shared__ uint32_t s[kOrderClasses];
    const uint32_t t = threadIdx.x;
    s[t] = hist[t];
    __syncthreads();
    uint32_t a = 0;
    for (uint32_t i = 0; i < t; i++) a += s[i];
    hist[t] = a;
}
__global__ __launch_bounds__(256) void swc_order_scatter_kernel(const Job* __restrict__ jobs, uint32_t n, uint32_t* __restrict__ cursor, uint32_t* __restrict__ perm) {
    __shared__ uint32_t cnt[kOrderClasses], base[kOrderClasses];
    const uint32_t t = threadIdx.x, g = blockIdx.x * 256u + t;
    cnt[t] = 0;
    __syncthreads();
    const uint32_t c = g < n ? order_class(jobs[g].in_len) : 0u;
    const uint32_t r = g < n ? atomicAdd(&cnt[c], 1u) : 0u;      // my rank among the group's members of the class
    __syncthreads();
    if (cnt[t]) base[t] = atomicAdd(&cursor[t], cnt[t]);          // the group's slots of class t
    __syncthreads();
    if (g < n) perm[base[c] + r] = g;
}
// device buffer of the calling thread for the order of one launch on `stream` (perm[n] | cursors), or nullptr
static const uint32_t* job_order(const Job* jobs, size_t n, hipStream_t stream) {
    if (n < kOrderMin) return nullptr;
    struct Buf { hipStream_t s; int dev; uint32_t* p; size_t cap; };
    static thread_local std::vector<Buf> bufs;   // (never freed: a few hundred KB per launching thread and stream)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    Buf* b = nullptr;
    for (auto& x : bufs) if (x.s == stream && x.dev == dev) b = &x;
    if (!b) { bufs.push_back(Buf{stream, dev, nullptr, 0}); b = &bufs.back(); }
    const size_t need = n + kOrderClasses;
    if (b->cap < need) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr; b->cap = 0;
        void* q = nullptr;
        if (hipMalloc(&q, need * 2 * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        b->p = (uint32_t*)q; b->cap = need * 2;
    }
    uint32_t* cursor = b->p + n;
    if (hipMemsetAsync(cursor, 0, kOrderClasses * sizeof(uint32_t), stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(swc_order_hist_kernel, grid, dim3(256), 0, stream, jobs, (uint32_t)n, cursor);
    hipLaunchKernelGGL(swc_order_prefix_kernel, dim3(1), dim3(kOrderClasses), 0, stream, cursor);
    hipLaunchKernelGGL(swc_order_scatter_kernel, grid, dim3(256), 0, stream, jobs, (uint32_t)n, cursor, b->p);
    return b->p;
}
__device__ __forceinline__ uint32_t job_of(const uint32_t* order, uint32_t b, uint32_t n) { return order ? order[b] : xcd_job(b, n); }

// ---- Deflate: two phases (inflate_sync.h + inflate_lane.h, lz_resolve.h) -------------------------------------------
// Phase 1: one stream per WAVEFRONT, 64 sub-chunks of the stream decoded at once (inflate_sync.h).
// LDS: the shared tables + the staged input of a round, exactly 10 KiB per wave -> 16 waves per CU.
#ifndef SWC_INFLATE_ORDER
#define SWC_INFLATE_ORDER 1
#endif
#ifndef SWC_SYNC_WAVES_PER_SIMD
#define SWC_SYNC_WAVES_PER_SIMD 4
#endif
// Measurement state belongs to the CALLING THREAD, like the launch stream and the staging buffers (api.cpp): two threads
// that bench at once do not see each other's events.
static thread_local uint64_t* g_prof = nullptr;   // profile builds (-DSWC_PROFILE): 32 counters per job, [0..16) phase 1, [16..32) phase 2
void set_profile_buffer(void* p) { g_prof = static_cast<uint64_t*>(p); }
__global__ __launch_bounds__(64, SWC_SYNC_WAVES_PER_SIMD) void swc_inflate_sync_kernel(Job* __restrict__ jobs, uint32_t n, WsMap wm, uint64_t* prof, const uint32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) inflate::SyncLds sync_lds;
    uint32_t g = job_of(order, blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    inflate::inflate_sync_job(job, &sync_lds, wm.area(g), wm.bytes(g), (int)threadIdx.x, kWave, prof ? prof + 32 * (size_t)g : nullptr);
    if (threadIdx.x == 0) {
        jobs[g].out_len = job.out_len;
        jobs[g].in_consumed = job.in_consumed;
        jobs[g].status = job.status;
    }
}

// Phase 2: one stream per workgroup of 512 threads, 64 KiB LDS ring (32 KiB of history + span + cells) -> 2 workgroups per CU.
constexpr int kInflateResolveThreads = 512, kInflateRingLog2 = 16;
constexpr uint32_t kInflateKeep = 32768;
__global__ __launch_bounds__(kInflateResolveThreads) void swc_lz_resolve_kernel(const Job* __restrict__ jobs, uint32_t n, WsMap wm, uint64_t* prof, const uint32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) lzr::Lds<kInflateResolveThreads, kInflateRingLog2> lzr_lds;  // static: > 64 KiB needs no opt-in this way
    uint32_t g = job_of(order, blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    lzr::resolve_job<kInflateResolveThreads, kInflateRingLog2, kInflateKeep>(job, wm.area(g), wm.bytes(g), &lzr_lds, prof ? prof + 32 * (size_t)g + 16 : nullptr);
}

// Phase 2, record-granular (lz_copy.h): one stream per WAVEFRONT, the last few KiB of its output in an LDS window, older
// sources read back from the output buffer.  The configurations (lz_copy.h: CfgDeflate, CfgLz4; profiles/r05_experiments.txt r05n-p,
// profiles/r06_experiments.txt r06b-c, r06t, r07a):
//   Deflate  6 KiB window of which a slide keeps 3.25 KiB, groups of up to 1 KiB, 6.4 KB of LDS per wave -> 24 waves per CU: 6.2 ms on
//            BASELINE configs[1] (round 5: 5 KiB window, 4,080 B kept, the literal stream staged in LDS: 7.65).  What a slide keeps is
//            what asking for far sources one group ahead needs, no more: the slide -- two LDS passes over everything kept -- costs
//            more than the far matches a short history makes (6 KiB / 5,104 kept: 7.30; / 4,096: 6.58; / 2,560: 6.41 before the
//            flush moved to the top of the iteration, which needs 3,328);
//   LZ4      7 KiB window, 3.25 KiB kept, groups of up to 1 KiB, runs of up to 32 literal bytes per lane, 81 VGPRs -> 20 waves per CU
//            (9 KiB / 2 KiB groups / 16 waves: 57.9 against 56.6 ms).
// The 16 KiB variant (8 waves per CU) is kept for comparison runs.
template <typename CFG, int RM = 0>
__device__ __forceinline__ void lz_copy_body(const Job* __restrict__ jobs, uint32_t n, const WsMap& wm, const uint32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) lzc::Lds<CFG::kWin> lds;
    uint32_t g = job_of(order, blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    if (job.dict != nullptr) return;   // (LZ4 blocks with a dictionary prefix were decoded by the lane kernel)
    lzc::copy_job<CFG, RM>(job, wm.area(g), wm.bytes(g), &lds);
}
#ifndef SWC_LZC_WAVES
#define SWC_LZC_WAVES 6
#endif
// how the LZ4 parse tells the copy kernel where a record's literals lie in the block (lz4_wave.h): 1 = eight-byte records with the
// offset in the upper dword, 2 = four-byte records, the offset derived by a running sum, anchors where the rule breaks
#ifndef SWC_LZ4_RECORD_MODE
#define SWC_LZ4_RECORD_MODE 2
#endif
#ifndef SWC_LZC4_WAVES
#define SWC_LZC4_WAVES 5
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SWC_LZC_WAVES, SWC_LZC_WAVES))) void swc_lz_copy_kernel(const Job* __restrict__ jobs, uint32_t n, WsMap wm, const uint32_t* __restrict__ order) {
    lz_copy_body<lzc::CfgDeflate>(jobs, n, wm, order);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SWC_LZC4_WAVES, SWC_LZC4_WAVES))) void swc_lz4_copy_kernel(const Job* __restrict__ jobs, uint32_t n, WsMap wm, const uint32_t* __restrict__ order) {
    lz_copy_body<lzc::CfgLz4, SWC_LZ4_RECORD_MODE>(jobs, n, wm, order);   // (the literals come from the block itself)
}
__global__ __launch_bounds__(64) void swc_lz_copy16_kernel(const Job* __restrict__ jobs, uint32_t n, WsMap wm, const uint32_t* __restrict__ order) {
    lz_copy_body<lzc::CfgWide>(jobs, n, wm, order);
}
// "lz_copier" (swc_set_tuning): 1 = lz_copy.h with the windows above (default), 2 = with a 16 KiB window, 0 = the byte-cell resolver
// of lz_resolve.h (rounds 2-4) -- kept for A/B measurements; all produce the same bytes.
static std::atomic<int> g_lz_copier{1};
void set_lz_copier(int v) { g_lz_copier = v; }
// A launch of FEW streams is a matter of latency, not throughput: one wave walks a 64 KiB stream group by group in ~0.3 ms,
// the 512 threads of the byte-cell resolver in 0.06 ms -- but only 512 of those fit the chip at a time against 4,096 waves.
// Below kCopierMin streams the launch takes the workgroup kernel (the single-shot calls, small containers), above it the
// wave kernel (the batches this engine is built for); both write the same bytes (tests/test_lz_copy_records.py).
constexpr size_t kCopierMin = 2560;
static int copier_for(size_t n) {
    const int c = g_lz_copier;
    return c < 0 ? -c : (c == 1 && n < kCopierMin ? 0 : c);   // (tuning value -1 / -2: the wave kernel whatever the batch size)
}
static void launch_lz_copy(int mode, bool lz4, const Job* jobs, size_t n, const WsMap& wm, const uint32_t* order, hipStream_t stream) {
    if (mode == 2 && !lz4) hipLaunchKernelGGL(swc_lz_copy16_kernel, dim3((unsigned)n), dim3(kWave), 0, stream, jobs, (uint32_t)n, wm, order);
    else if (lz4) hipLaunchKernelGGL(swc_lz4_copy_kernel, dim3((unsigned)n), dim3(kWave), 0, stream, jobs, (uint32_t)n, wm, order);
    else hipLaunchKernelGGL(swc_lz_copy_kernel, dim3((unsigned)n), dim3(kWave), 0, stream, jobs, (uint32_t)n, wm, order);
}

// Optional per-kernel timing of the calling thread's last batch launch (bench.py: roofline per kernel).  HIP events on the
// launch stream between the kernels; off by default so that the production path issues nothing but the kernels.
constexpr int kMaxPhases = 8;
struct PhaseTimer {
    int on = 0, n = 0;
    bool ok = false, valid = false;
    hipEvent_t ev[kMaxPhases + 1];
    void begin(hipStream_t s) { n = 0; valid = false; mark(s); }
    void mark(hipStream_t s) {   // after every kernel of the launch
        if (on && ok && n <= kMaxPhases) { (void)hipEventRecord(ev[n], s); n++; valid = n >= 2; }
    }
};
static thread_local PhaseTimer g_pt;
void set_phase_timing(int on) {
    g_pt.on = on;
    if (on && !g_pt.ok) {
        g_pt.ok = true;
        for (auto& e : g_pt.ev) if (hipEventCreate(&e) != hipSuccess) g_pt.ok = false;
    }
}
int last_phase_ms(float* ms, int cap) {
    if (!g_pt.valid || cap < g_pt.n - 1) return 0;
    if (hipEventSynchronize(g_pt.ev[g_pt.n - 1]) != hipSuccess) return 0;
    for (int i = 0; i + 1 < g_pt.n; i++) if (hipEventElapsedTime(&ms[i], g_pt.ev[i], g_pt.ev[i + 1]) != hipSuccess) return 0;
    return g_pt.n - 1;
}
size_t inflate_ws_bytes_per_job(uint64_t cap) { return lzr::ws_bytes_per_job(cap); }

// "deflate_team" (swc_set_tuning): 1 = launches of up to kTeamMaxStreams streams give every stream a team of wavefronts (default),
// 0 = one wavefront per stream whatever the launch, -1 = a team for launches of up to kTeamForceStreams (tests)
static std::atomic<int> g_deflate_team{1};
void set_deflate_team(int v) { g_deflate_team = v; }
constexpr size_t kTeamMaxStreams = 256, kTeamForceStreams = 4096;

hipError_t launch_inflate(Job* jobs, size_t n, void* ws, size_t ws_bytes, hipStream_t stream, const uint64_t* ws_off) {
    if (n == 0) return hipSuccess;
    size_t stride = ws ? (ws_bytes / n) & ~(size_t)15 : 0;
    if (!ws_off && stride < sizeof(lzr::StreamHeader)) return hipErrorInvalidValue;
    const WsMap wm{(uint8_t*)ws, stride, ws_off};
    const dim3 block(kWave);
    const int team = g_deflate_team;
    if (!g_prof && ((team == 1 && n <= kTeamMaxStreams) || (team < 0 && n <= kTeamForceStreams))) {
        // a team per stream (inflate_team.hip): the helpers' rows come from the stream-ordered pool for the length of the launch; no
        // ordering pass (a workgroup per stream and at most two per CU: the launch is one residency round)
        void* scratch = nullptr;
        if (hipMallocAsync(&scratch, inflate_team_scratch_bytes(n), stream) == hipSuccess) {
            g_pt.begin(stream);
            (void)launch_inflate_team(jobs, n, (uint8_t*)ws, stride, ws_off, (uint8_t*)scratch, stream);   // (hipGetLastError below)
            g_pt.mark(stream);
            (void)hipFreeAsync(scratch, stream);
            const int copier = copier_for(n);
            if (copier) launch_lz_copy(copier, false, jobs, n, wm, nullptr, stream);
            else hipLaunchKernelGGL(swc_lz_resolve_kernel, dim3((unsigned)n), dim3(kInflateResolveThreads), 0, stream, jobs, (uint32_t)n, wm, g_prof, (const uint32_t*)nullptr);
            g_pt.mark(stream);
            return hipGetLastError();
        }
        (void)hipGetLastError();   // no room for the rows: one wavefront per stream
    }
    // (streams of unequal cost -- stored blocks, incompressible stretches -- are launched longest first, like the LZ4 blocks)
    g_pt.begin(stream);   // (the ordering kernels are part of the first phase's figure: ADVICE r4)
    const uint32_t* order = SWC_INFLATE_ORDER ? job_order(jobs, n, stream) : nullptr;
    hipLaunchKernelGGL(swc_inflate_sync_kernel, dim3((unsigned)n), block, 0, stream, jobs, (uint32_t)n, wm, g_prof, order);
    g_pt.mark(stream);
    const int copier = copier_for(n);
    if (copier) launch_lz_copy(copier, false, jobs, n, wm, order, stream);
    else hipLaunchKernelGGL(swc_lz_resolve_kernel, dim3((unsigned)n), dim3(kInflateResolveThreads), 0, stream, jobs, (uint32_t)n, wm, g_prof, order);
    g_pt.mark(stream);
    return hipGetLastError();
}

// ---- LZ4 -------------------------------------------------------------------------------------------------
// Blocks without a dictionary prefix take the two-phase path: one block per wavefront for the sequence parse
// (lz4_wave.h), then the LZ77 resolve kernel of lz_resolve.h in the Deflate configuration (one block per 512-thread
// workgroup, 64 KiB ring with 32 KiB of history, two workgroups per CU; match bytes from further back come from the
// output buffer).  Blocks with a dictionary prefix (dependent frames, external dictionaries) stay on the
// one-block-per-lane decoder (lz4_lane.h).  Each kernel skips the jobs of the other kind.
__global__ __launch_bounds__(64) void swc_lz4_lane_kernel(Job* __restrict__ jobs, uint32_t n, int only_dict) {
    uint32_t g = blockIdx.x * kWave + threadIdx.x;
    if (g >= n) return;
    Job job = jobs[g];
    if (only_dict && job.dict == nullptr) return;
    lz4::lz4_block_job(job);
    jobs[g].out_len = job.out_len;
    jobs[g].in_consumed = job.in_consumed;
    jobs[g].status = job.status;
}

#ifndef SWC_LZ4_PARSE_WAVES
#define SWC_LZ4_PARSE_WAVES 4
#endif
// R8: eight-byte records that say where their literals lie in the block, no literal stream (for the wave copy kernel);
// otherwise the records and the dense literal stream swc_lz4_resolve_kernel reads
template <int RM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SWC_LZ4_PARSE_WAVES))) void swc_lz4_parse_kernel(Job* __restrict__ jobs, uint32_t n, WsMap wm, uint64_t* prof, const uint32_t* __restrict__ order) {
    uint32_t g = job_of(order, blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    if (job.dict != nullptr) return;
    __shared__ __attribute__((aligned(16))) uint8_t stage[lz4w::kStageLds];
    lz4w::lz4_parse_job<kWave, RM>(job, wm.area(g), wm.bytes(g), (int)threadIdx.x, stage, prof ? prof + 32 * (size_t)g : nullptr);
    if (threadIdx.x == 0) {
        jobs[g].out_len = job.out_len;
        jobs[g].in_consumed = job.in_consumed;
        jobs[g].status = job.status;
    }
}

__global__ __launch_bounds__(lz4w::kResolveThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void swc_lz4_resolve_kernel(const Job* __restrict__ jobs, uint32_t n, WsMap wm, uint64_t* prof, const uint32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) lzr::Lds<lz4w::kResolveThreads, lz4w::kRingLog2> lds;
    uint32_t g = job_of(order, blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    if (job.dict != nullptr) return;
    lzr::resolve_job<lz4w::kResolveThreads, lz4w::kRingLog2, lz4w::kKeep, true>(job, wm.area(g), wm.bytes(g), &lds, prof ? prof + 32 * (size_t)g + 16 : nullptr);
}

size_t lz4_ws_bytes_per_job(uint64_t cap) { return lzr::ws_bytes_per_job(cap); }

hipError_t launch_lz4(Job* jobs, size_t n, void* ws, size_t ws_bytes, hipStream_t stream, const uint64_t* ws_off) {
    if (n == 0) return hipSuccess;
    dim3 grid((unsigned)((n + kWave - 1) / kWave)), block(kWave);
    size_t stride = ws ? (ws_bytes / n) & ~(size_t)15 : 0;
    if (!ws_off && stride < sizeof(lzr::StreamHeader)) {
        // no workspace: every block on the lane decoder
        hipLaunchKernelGGL(swc_lz4_lane_kernel, grid, block, 0, stream, jobs, (uint32_t)n, 0);
        return hipGetLastError();
    }
    // The parse is latency bound per block (a serial chase), so all blocks are parsed in ONE launch: the more waves
    // in flight, the better the latency hides (8,192 blocks = 8 waves per SIMD).
    g_pt.begin(stream);
    const uint32_t* order = job_order(jobs, n, stream);
    hipLaunchKernelGGL(swc_lz4_lane_kernel, grid, block, 0, stream, jobs, (uint32_t)n, 1);
    g_pt.mark(stream);
    const WsMap wm{(uint8_t*)ws, stride, ws_off};
    const int copier = copier_for(n);
    if (copier) hipLaunchKernelGGL(swc_lz4_parse_kernel<SWC_LZ4_RECORD_MODE>, dim3((unsigned)n), block, 0, stream, jobs, (uint32_t)n, wm, g_prof, order);
    else hipLaunchKernelGGL(swc_lz4_parse_kernel<0>, dim3((unsigned)n), block, 0, stream, jobs, (uint32_t)n, wm, g_prof, order);
    g_pt.mark(stream);
    if (copier) launch_lz_copy(copier, true, jobs, n, wm, order, stream);
    else hipLaunchKernelGGL(swc_lz4_resolve_kernel, dim3((unsigned)n), dim3(lz4w::kResolveThreads), 0, stream, jobs, (uint32_t)n, wm, g_prof, order);
    g_pt.mark(stream);
    return hipGetLastError();
}

// ---- LZ4 block compression (lz4_comp.h), one block per wavefront, the hash table in LDS -------------------------------------
__global__ __launch_bounds__(64) void swc_lz4_compress_kernel(Job* __restrict__ jobs, uint32_t n, const uint32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) uint16_t table[lz4c::kHashSize];
    uint32_t g = job_of(order, blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    lz4c::lz4_compress_job<kWave>(job, table);
    if (threadIdx.x == 0) {
        jobs[g].out_len = job.out_len;
        jobs[g].in_consumed = job.in_consumed;
        jobs[g].status = job.status;
    }
}
hipError_t launch_lz4_compress(Job* jobs, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    g_pt.begin(stream);
    const uint32_t* order = job_order(jobs, n, stream);
    hipLaunchKernelGGL(swc_lz4_compress_kernel, dim3((unsigned)n), dim3(kWave), 0, stream, jobs, (uint32_t)n, order);
    g_pt.mark(stream);
    return hipGetLastError();
}

// ---- Deflate compression (deflate_comp.h), one buffer per wavefront: hash table + a staging area of the bit stream in LDS ------
__global__ __launch_bounds__(64) void swc_deflate_compress_kernel(Job* __restrict__ jobs, uint32_t n, const uint32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) defc::Lds lds;
    uint32_t g = job_of(order, blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    defc::deflate_compress_job<kWave>(job, &lds);
    if (threadIdx.x == 0) {
        jobs[g].out_len = job.out_len;
        jobs[g].in_consumed = job.in_consumed;
        jobs[g].status = job.status;
    }
}
hipError_t launch_deflate_compress(Job* jobs, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    g_pt.begin(stream);
    const uint32_t* order = job_order(jobs, n, stream);
    hipLaunchKernelGGL(swc_deflate_compress_kernel, dim3((unsigned)n), dim3(kWave), 0, stream, jobs, (uint32_t)n, order);
    g_pt.mark(stream);
    return hipGetLastError();
}

// ---- LZMA / LZMA2, one stream per wavefront ---------------------------------------------------------
// LDS: the u16 probability cells of one stream.  With the workspace (spill area for big literal coders) at hand the
// kernel keeps literal coders up to lc + lp = 3 in LDS -- what xz writes -- which is 15,984 B per wave = 10 streams per
// CU; streams with lc + lp > 3 then run their literal coder out of HBM.  Without a workspace lc + lp = 4 still fits in
// LDS (28,272 B, 5 streams per CU) and only lc + lp > 4 (legal for .lzma, never produced by xz) reports
// SWC_E_NEED_WORKSPACE.  `spill` holds kLzmaSpillBytes per job.
constexpr size_t kLzmaSpillBytes = (size_t)(0x300u << 12) * 2 + 1024;   // every literal coder of lc + lp <= 12, then the two `high` length trees

// LDSBITS >= 0: the literal coders of lc + lp <= LDSBITS in LDS (no workspace: 4 -> 28 KB, 5 streams per CU);
// LDSBITS < 0: LDS as a cache of kCoderSlots literal coders (one: 5,232 B, 31 streams per CU), all of them in the workspace (lzma_wave.h)
#ifndef SWC_LZMA_WAVES
#define SWC_LZMA_WAVES 8
#endif
template <bool LZMA2, int LDSBITS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SWC_LZMA_WAVES))) void swc_lzma_kernel(Job* __restrict__ jobs, uint32_t n, uint8_t* spill, uint64_t* prof) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lzma_lds[];
    uint32_t g = xcd_job(blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    SWC_AS_GLOBAL uint16_t* sp = spill ? (SWC_AS_GLOBAL uint16_t*)(spill + (size_t)g * kLzmaSpillBytes) : nullptr;
    lzma::lzma_job<kWave>(job, LZMA2, lzma_lds, sp, (int)threadIdx.x, LDSBITS < 0 ? 0 : LDSBITS, prof ? prof + 32 * (size_t)g : nullptr, LDSBITS < 0);
    if (threadIdx.x == 0) {
        jobs[g].out_len = job.out_len;
        jobs[g].in_consumed = job.in_consumed;
        jobs[g].status = job.status;
    }
}

size_t lzma_spill_bytes_per_job() { return kLzmaSpillBytes; }

// "lzma_coder_cache" (swc_set_tuning): 1 = LDS as a cache of the literal coders (kCoderSlots of them; default), 0 = round 2's layout (all eight
// coders of lc + lp <= 3 in LDS, 10 streams per CU) -- kept for A/B measurements; both produce the same bytes.
static std::atomic<int> g_lzma_cache{1};
void set_lzma_coder_cache(int on) { g_lzma_cache = on; }

hipError_t launch_lzma(bool lzma2, Job* jobs, size_t n, void* spill, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 grid((unsigned)n), block(kWave);
    g_pt.begin(stream);
    if (spill && !g_lzma_cache) {
        if (lzma2) hipLaunchKernelGGL((swc_lzma_kernel<true, 3>), grid, block, lzma::lds_bytes_for(3), stream, jobs, (uint32_t)n, (uint8_t*)spill, g_prof);
        else hipLaunchKernelGGL((swc_lzma_kernel<false, 3>), grid, block, lzma::lds_bytes_for(3), stream, jobs, (uint32_t)n, (uint8_t*)spill, g_prof);
    } else if (spill) {
        if (lzma2) hipLaunchKernelGGL((swc_lzma_kernel<true, -1>), grid, block, lzma::lds_bytes_cached(), stream, jobs, (uint32_t)n, (uint8_t*)spill, g_prof);
        else hipLaunchKernelGGL((swc_lzma_kernel<false, -1>), grid, block, lzma::lds_bytes_cached(), stream, jobs, (uint32_t)n, (uint8_t*)spill, g_prof);
    } else {
        if (lzma2) hipLaunchKernelGGL((swc_lzma_kernel<true, 4>), grid, block, lzma::lds_bytes_for(4), stream, jobs, (uint32_t)n, (uint8_t*)nullptr, g_prof);
        else hipLaunchKernelGGL((swc_lzma_kernel<false, 4>), grid, block, lzma::lds_bytes_for(4), stream, jobs, (uint32_t)n, (uint8_t*)nullptr, g_prof);
    }
    g_pt.mark(stream);
    return hipGetLastError();
}

// ---- BZip2: three stages per block (see bzip2_block.h), ONE kernel ----------------------------------------------
// A wavefront takes its block through stage 1 (Huffman + MTF: bound by the CU's scalar unit, which all resident waves
// share), stage 2 (counting-sort scatter) and stage 3a (the segmented walk of the BWT cycle: bound by the latency of
// random HBM accesses) back to back.  As separate launches the three ran one after the other, each limited by its own
// resource while the others idled; in one kernel the waves of a CU are at different stages at any time, so the scalar
// work of some overlaps the memory waits of the others.  LDS: the three stages' areas share one allocation.
#ifndef SWC_BZ_LDS_PAD
#define SWC_BZ_LDS_PAD 0   // (occupancy experiments: more LDS per wave = fewer waves per CU, nothing else changed)
#endif
constexpr size_t kBzLdsBytes = (bzip2::kStage1LdsBytes > sizeof(bzip2::Stage3Lds) ? (size_t)bzip2::kStage1LdsBytes : sizeof(bzip2::Stage3Lds)) + SWC_BZ_LDS_PAD;
static_assert(kBzLdsBytes >= 256 * sizeof(uint32_t), "stage 2 counters");
#ifndef SWC_BZ_WAVES
#define SWC_BZ_WAVES 6   // (round 4, stage 3 inside: 7 -> 72 VGPRs + 8 bytes of scratch, 495 ms against 479; round 6, stages 1 + 2 only: 6 / 7 / 8 -> 198.4 / 200.3 / 199.4 ms)
#endif
// CXX: the plain-symbol loop of stage 1 compiled from its C++ twin instead of the hand-written assembly ("bzip2_hot_cxx",
// swc_set_tuning: the differential GPU test of the two, tests/test_gpu_bzip2.py)
// team != 0: stage 1 and 2 only -- the walk, the lay-out and the RLE1 undo follow as kernels of their own (bzip2_team.h)
template <bool CXX>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SWC_BZ_WAVES))) void swc_bzip2_block_kernel(Job* __restrict__ jobs, uint32_t n, uint8_t* ws, size_t lcap, int team) {
    __shared__ __attribute__((aligned(16))) uint8_t bz_lds[kBzLdsBytes];
    uint32_t g = xcd_job(blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    const bzip2::Workspace w = bzip2::carve(ws, g, lcap);
    bzip2::stage1_job<kWave, CXX>(job, reinterpret_cast<bzip2::Stage1Lds*>(bz_lds), w, (int)threadIdx.x);
#if defined(SWC_BZ_STOP_AFTER) && SWC_BZ_STOP_AFTER == 1   // (timing experiments only, tools/attic/exp_bz_stages.py: wrong results)
    if (threadIdx.x == 0) { w.hdr->pad = bzip2::kWalkDone; jobs[g].status = SWC_E_DEVICE; }
    return;
#endif
    __threadfence_block();   // L and the block header, written by some lanes, are read by all of them from here on
    bzip2::stage2_job(w, reinterpret_cast<uint32_t*>(bz_lds));
#if defined(SWC_BZ_STOP_AFTER) && SWC_BZ_STOP_AFTER == 2
    if (threadIdx.x == 0) { w.hdr->pad = bzip2::kWalkDone; jobs[g].status = SWC_E_DEVICE; }
    return;
#endif
    if (team) return;
    __threadfence_block();   // likewise the pointer array P
    bzip2::stage3_walk_job<kWave>(job, w, reinterpret_cast<bzip2::Stage3Lds*>(bz_lds), (int)threadIdx.x);
    if (threadIdx.x == 0 && !bzip2::stage3_expand_needed(w)) {
        jobs[g].out_len = job.out_len;
        jobs[g].in_consumed = job.in_consumed;
        jobs[g].status = job.status;
        jobs[g].aux = job.aux;
    }
}

// ---- stage 3a as kernels of its own (bzip2_team.h) ------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void swc_bzip2_team_prep_kernel(uint8_t* ws, size_t lcap, uint32_t n) {
    bzip2::team_prep<kWave>(ws, lcap, n, blockIdx.x, (int)threadIdx.x);
}
#ifndef SWC_BZ_TEAM_THREADS
#define SWC_BZ_TEAM_THREADS 1024
#endif
__global__ __launch_bounds__(SWC_BZ_TEAM_THREADS) void swc_bzip2_team_walk_kernel(uint8_t* ws, size_t lcap, uint32_t n) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    bzip2::team_walk(ws, lcap, n, xcc & (bzip2::kTeams - 1u));
}
__global__ __launch_bounds__(64) void swc_bzip2_team_finish_kernel(Job* __restrict__ jobs, uint32_t n, uint8_t* ws, size_t lcap) {
    __shared__ bzip2::FinishLds lds;
    uint32_t g = xcd_job(blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    const bzip2::Workspace w = bzip2::carve(ws, g, lcap);
    bzip2::team_finish<kWave>(job, w, &lds, (int)threadIdx.x);
    if (threadIdx.x == 0 && !bzip2::stage3_expand_needed(w)) {
        jobs[g].out_len = job.out_len;
        jobs[g].in_consumed = job.in_consumed;
        jobs[g].status = job.status;
        jobs[g].aux = job.aux;
    }
}

// stage 3b: one block per lane, only what stage 3a could not finish (serial walk + RLE1 undo)
__global__ __launch_bounds__(64) void swc_bzip2_expand_kernel(Job* __restrict__ jobs, uint32_t n, uint8_t* ws, size_t lcap) {
    uint32_t g = blockIdx.x * kWave + threadIdx.x;
    if (g >= n) return;
    const bzip2::Workspace w = bzip2::carve(ws, g, lcap);
    if (!bzip2::stage3_expand_needed(w)) return;
    Job job = jobs[g];
    bzip2::stage3_expand_job(job, w);
    jobs[g].out_len = job.out_len;
    jobs[g].in_consumed = job.in_consumed;
    jobs[g].status = job.status;
    jobs[g].aux = job.aux;
}

// stage 3c: one block per workgroup (block CRC, BZip2.swift:81)
__global__ __launch_bounds__(256) void swc_bzip2_crc_kernel(Job* __restrict__ jobs, uint32_t n) {
    __shared__ crc::Lds<256, uint32_t> lds;
    uint32_t g = xcd_job(blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    if (job.status != SWC_OK) return;   // decode errors and SWC_E_CAPACITY stand
    const uint32_t c = crc::crc_group<256, uint32_t, true>((gcptr)job.out, job.out_len, &lds, (int)threadIdx.x);
    if (threadIdx.x == 0) {
        bzip2::stage3_check_crc(job, c);
        jobs[g].status = job.status;
        jobs[g].aux = job.aux;
    }
}

size_t bzip2_ws_bytes_per_job(size_t lcap) { return bzip2::ws_bytes_per_job(lcap); }
static std::atomic<int> g_bzip2_hot_cxx{0};
void set_bzip2_hot_cxx(int v) { g_bzip2_hot_cxx = v; }
static std::atomic<int> g_bzip2_team_walk{1}, g_bzip2_team_per_cu{1};
constexpr size_t kTeamMinBytes = 65536;
void set_bzip2_team_walk(int v) { g_bzip2_team_walk = v; }
void set_bzip2_team_per_cu(int v) { g_bzip2_team_per_cu = v; }

hipError_t launch_bzip2(Job* jobs, size_t n, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (!ws) return hipErrorInvalidValue;
    const size_t per_job = ws_bytes / n;
    // the largest L capacity whose workspace fits (ws_bytes_per_job is monotonic in lcap)
    if (per_job < bzip2::ws_bytes_per_job(16)) return hipErrorInvalidValue;
    size_t lo = 16, hi = 16000000;               // upper limit: i << 8 | c packing of stage 2
    if (bzip2::ws_bytes_per_job(hi) <= per_job) lo = hi;
    while (hi - lo > 1) {
        const size_t mid = lo + (hi - lo) / 2;
        if (bzip2::ws_bytes_per_job(mid) <= per_job) lo = mid; else hi = mid;
    }
    const size_t lcap = lo;
    dim3 block(kWave);
    g_pt.begin(stream);
    // "bzip2_team_walk" (swc_set_tuning): 1 = launches with kTeamMinBytes of column and more walk out of the XCDs' L2 (bzip2_team.h:
    // faster from ONE block of 100 kB on -- 900 kB: 71 against 85 ms -- its four extra launches cost a 500-byte stream 0.3 ms),
    // 0 = never, 2 = always
    const int tw = g_bzip2_team_walk;
    const int team = tw == 2 || (tw == 1 && n * lcap >= kTeamMinBytes) ? 1 : 0;
    if (g_bzip2_hot_cxx) hipLaunchKernelGGL(swc_bzip2_block_kernel<true>, dim3((unsigned)n), block, 0, stream, jobs, (uint32_t)n, (uint8_t*)ws, lcap, team);
    else hipLaunchKernelGGL(swc_bzip2_block_kernel<false>, dim3((unsigned)n), block, 0, stream, jobs, (uint32_t)n, (uint8_t*)ws, lcap, team);
    g_pt.mark(stream);
    if (team) {   // (phase timing: block | team prep + walk | team finish | expand | crc -- five figures with the team walk, three without)
        hipLaunchKernelGGL(swc_bzip2_team_prep_kernel, dim3(bzip2::kTeams), block, 0, stream, (uint8_t*)ws, lcap, (uint32_t)n);
        hipLaunchKernelGGL(swc_bzip2_team_walk_kernel, dim3(256 * g_bzip2_team_per_cu), dim3(SWC_BZ_TEAM_THREADS), 0, stream, (uint8_t*)ws, lcap, (uint32_t)n);
        g_pt.mark(stream);
        hipLaunchKernelGGL(swc_bzip2_team_finish_kernel, dim3((unsigned)n), block, 0, stream, jobs, (uint32_t)n, (uint8_t*)ws, lcap);
        g_pt.mark(stream);
    }
    hipLaunchKernelGGL(swc_bzip2_expand_kernel, dim3((unsigned)((n + kWave - 1) / kWave)), block, 0, stream, jobs, (uint32_t)n, (uint8_t*)ws, lcap);
    g_pt.mark(stream);
    hipLaunchKernelGGL(swc_bzip2_crc_kernel, dim3((unsigned)n), dim3(256), 0, stream, jobs, (uint32_t)n);
    g_pt.mark(stream);
    return hipGetLastError();
}

// ---- CRC-32 of every job's output -----------------------------------------------------------------
// Two kernels, both launched over all n jobs, each taking the jobs of its size class (the sizes are on the device; a wave or
// group whose job belongs to the other kernel ends at once): one stream per WAVE below 1 MB (crc32_wave.h: no per-stream
// set-up, no barrier after the constants are in LDS), one stream per 256-thread group above (crc32_group.h).
constexpr uint64_t kCrcGroupLen = 1u << 20;
__device__ crcw::WaveConsts g_crc_consts;
__global__ __launch_bounds__(256) void swc_crc32_consts_kernel() { crcw::build_consts<256>(&g_crc_consts, (int)threadIdx.x); }

__global__ __launch_bounds__(256) void swc_crc32_kernel(const Job* __restrict__ jobs, uint32_t n, uint32_t* __restrict__ crcs) {
    __shared__ crcw::WaveConsts lds;
    {   // tables, G tables and fold matrices: 2240 words
        const uint32_t* src = (const uint32_t*)&g_crc_consts;
        uint32_t* dst = (uint32_t*)&lds;
        constexpr int kWords = (int)(offsetof(crcw::WaveConsts, sq) / 4);
        for (int i = (int)threadIdx.x; i < kWords; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n) return;
    const uint64_t len = jobs[g].out_len < jobs[g].out_cap ? jobs[g].out_len : jobs[g].out_cap;
    if (len >= kCrcGroupLen) return;
    const uint32_t c = crcw::crc32_wave((gcptr)jobs[g].out, simt::uniform(len), &lds);
    if ((threadIdx.x & 63) == 0) crcs[g] = c;
}

// A fixed number of groups, each looking after a contiguous range of jobs: one parallel read of the lengths, then the streams
// of a megabyte and more one after the other (none in a batch of small members: the groups end after that one read).
constexpr int kCrcGroupGrid = 2048;
__global__ __launch_bounds__(256) void swc_crc32_group_kernel(const Job* __restrict__ jobs, uint32_t n, uint32_t* __restrict__ crcs) {
    __shared__ crc::Lds<256> lds;
    __shared__ uint32_t big[256];
    __shared__ uint32_t nbig;
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t base = lo; base < hi; base += 256) {
        if (threadIdx.x == 0) nbig = 0;
        __syncthreads();
        const uint32_t g = base + threadIdx.x;
        if (g < hi) {
            const uint64_t len = jobs[g].out_len < jobs[g].out_cap ? jobs[g].out_len : jobs[g].out_cap;
            if (len >= kCrcGroupLen) big[atomicAdd(&nbig, 1u)] = g;
        }
        __syncthreads();
        const uint32_t cnt = nbig;
        for (uint32_t i = 0; i < cnt; i++) {
            const uint32_t j = big[i];
            const uint64_t len = jobs[j].out_len < jobs[j].out_cap ? jobs[j].out_len : jobs[j].out_cap;
            const uint32_t c = crc::crc32_group<256>((gcptr)jobs[j].out, len, &lds, (int)threadIdx.x);
            if (threadIdx.x == 0) crcs[j] = c;
            __syncthreads();
        }
    }
}

hipError_t launch_crc32(const Job* jobs, size_t n, uint32_t* crcs, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    {   // the constants, once per device (the first call on a device builds them on the caller's stream; later launches on any
        // stream of the device come after it in host time and, through the wait below, in device time)
        static std::mutex mu;
        static bool built[64] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> lk(mu);
        if (!built[dev]) {
            hipLaunchKernelGGL(swc_crc32_consts_kernel, dim3(1), dim3(256), 0, stream);
            if (hipStreamSynchronize(stream) != hipSuccess) return hipErrorUnknown;
            built[dev] = true;
        }
    }
    hipLaunchKernelGGL(swc_crc32_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, jobs, (uint32_t)n, crcs);
    hipLaunchKernelGGL(swc_crc32_group_kernel, dim3((unsigned)(n < (size_t)kCrcGroupGrid ? n : (size_t)kCrcGroupGrid)), dim3(256), 0, stream, jobs, (uint32_t)n, crcs);
    return hipGetLastError();
}

// ---- the other checksums of the archive layer; sums[g] is zero-extended to 64 bits --------------
template <typename W, bool MSB>
__global__ __launch_bounds__(256) void swc_crc_kernel(const Job* __restrict__ jobs, uint32_t n, uint64_t* __restrict__ sums) {
    __shared__ crc::Lds<256, W> lds;
    uint32_t g = xcd_job(blockIdx.x, n);
    if (g >= n) return;
    const uint64_t len = jobs[g].out_len < jobs[g].out_cap ? jobs[g].out_len : jobs[g].out_cap;
    W c = crc::crc_group<256, W, MSB>((gcptr)jobs[g].out, len, &lds, (int)threadIdx.x);
    if (threadIdx.x == 0) sums[g] = (uint64_t)c;
}

__global__ __launch_bounds__(256) void swc_adler32_kernel(const Job* __restrict__ jobs, uint32_t n, uint64_t* __restrict__ sums) {
    __shared__ sums::AdlerLds<256> lds;
    uint32_t g = xcd_job(blockIdx.x, n);
    if (g >= n) return;
    const uint64_t len = jobs[g].out_len < jobs[g].out_cap ? jobs[g].out_len : jobs[g].out_cap;
    uint32_t c = sums::adler32_group<256>((gcptr)jobs[g].out, len, &lds, (int)threadIdx.x);
    if (threadIdx.x == 0) sums[g] = c;
}

// 16 streams per wave, four lanes each
__global__ __launch_bounds__(64) void swc_xxh32_kernel(const Job* __restrict__ jobs, uint32_t n, uint64_t* __restrict__ sums) {
    const uint32_t g = blockIdx.x * 16 + (threadIdx.x >> 2);
    const int j = (int)(threadIdx.x & 3);
    const bool live = g < n;
    gcptr out = live ? (gcptr)jobs[g].out : nullptr;
    const uint64_t len = live ? (jobs[g].out_len < jobs[g].out_cap ? jobs[g].out_len : jobs[g].out_cap) : 0;
    auto quad_get = [](uint32_t v, int k) -> uint32_t {
        switch (k) {   // quad_perm broadcasts
            case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xf, 0xf, false);
            case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xf, 0xf, false);
            case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xf, 0xf, false);
            default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xf, 0xf, false);
        }
    };
    const uint32_t h = sums::xxh32_quad(out, len, 0u, j, quad_get);
    if (live && j == 0) sums[g] = h;
}

// ---- Delta filter, one stream per 256-thread workgroup ------------------------------------------
__global__ __launch_bounds__(256) void swc_delta_kernel(Job* __restrict__ jobs, uint32_t n) {
    __shared__ delta::Lds<256> lds;
    uint32_t g = xcd_job(blockIdx.x, n);
    if (g >= n) return;
    Job job = jobs[g];
    const bool fits = job.in_len <= job.out_cap;
    if (fits) delta::delta_group<256>((gcptr)job.in, (gptr)job.out, job.in_len, (uint32_t)job.aux, &lds, (int)threadIdx.x);
    if (threadIdx.x == 0) {
        jobs[g].out_len = job.in_len;
        jobs[g].in_consumed = job.in_len;
        jobs[g].status = fits ? SWC_OK : SWC_E_CAPACITY;
    }
}

hipError_t launch_delta(Job* jobs, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(swc_delta_kernel, dim3((unsigned)n), dim3(256), 0, stream, jobs, (uint32_t)n);
    return hipGetLastError();
}

hipError_t launch_checksum(int kind, const Job* jobs, size_t n, uint64_t* sums, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const dim3 grid((unsigned)n), block(256);
    switch (kind) {
        case 1: hipLaunchKernelGGL((swc_crc_kernel<uint32_t, false>), grid, block, 0, stream, jobs, (uint32_t)n, sums); break;
        case 2: hipLaunchKernelGGL(swc_adler32_kernel, grid, block, 0, stream, jobs, (uint32_t)n, sums); break;
        case 3: hipLaunchKernelGGL((swc_crc_kernel<uint64_t, false>), grid, block, 0, stream, jobs, (uint32_t)n, sums); break;
        case 4: hipLaunchKernelGGL((swc_crc_kernel<uint32_t, true>), grid, block, 0, stream, jobs, (uint32_t)n, sums); break;
        case 5: hipLaunchKernelGGL(swc_xxh32_kernel, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, stream, jobs, (uint32_t)n, sums); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace swc
