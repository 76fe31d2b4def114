// kernels.hip -- gfx950 kernels of the many-stream decode engine and their launchers.
//
// Execution model: ONE COMPRESSED STREAM PER LANE.  A workgroup is a single 64-lane wavefront; each
// lane owns one independent unit (deflate stream / LZ4 block / ...) and runs the sequential decoder
// of <codec>_lane.h with all of its per-stream tables in LDS, interleaved at wave stride
// (word j of lane l at lds[j*64+l]) so arbitrary per-lane table indices are bank-conflict free.
// Entropy decoding is inherently serial per stream; the chip is filled by streams, not by
// splitting a stream: 256 CUs x 4 waves x 64 lanes = 65,536 streams in flight for Deflate.
// No MFMA (no dense contraction anywhere on this path), no inter-workgroup communication.
#include <hip/hip_runtime.h>
#include "swc_common.h"
#include "inflate_lane.h"
#include "lz4_lane.h"
#include "lzma_wave.h"
#include "bzip2_block.h"
#include "launch.h"

namespace swc {

// ---- Deflate --------------------------------------------------------------------------------
// LDS: 152 words/lane -> 38,912 B per wave -> 4 resident waves per CU (160 KiB LDS).
// G lanes execute each stream redundantly (identical registers, shared LDS tables): G x fewer streams per
// wave, G x less LDS per wave, so G x more resident waves per SIMD to hide latency.
template <int G, int DBG>
__global__ __launch_bounds__(64, 1) void swc_inflate_kernel(Job* __restrict__ jobs, uint32_t n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int kStreams = kWave / G;
    uint32_t sl = threadIdx.x / G;
    uint32_t g = blockIdx.x * kStreams + sl;
    if (g >= n) return;
    Job job = jobs[g];
    inflate::inflate_job<DBG>(job, LaneLds{lds + sl, kStreams});
    if (threadIdx.x % G == 0) {
        jobs[g].out_len = job.out_len;
        jobs[g].in_consumed = job.in_consumed;
        jobs[g].status = job.status;
    }
}

static int g_inflate_g = 1, g_inflate_dbg = 0;
void set_inflate_group(int g) { g_inflate_g = g; }
void set_inflate_debug(int m) { g_inflate_dbg = m; }

hipError_t launch_inflate(Job* jobs, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const int G = g_inflate_g;
    const unsigned streams = kWave / G;
    dim3 grid((unsigned)((n + streams - 1) / streams)), block(kWave);
    size_t lds = inflate::kLdsBytesPerWave;
    if (g_inflate_dbg == 1) hipLaunchKernelGGL((swc_inflate_kernel<1, 1>), grid, block, lds, stream, jobs, (uint32_t)n);
    else if (g_inflate_dbg == 2) hipLaunchKernelGGL((swc_inflate_kernel<1, 2>), grid, block, lds, stream, jobs, (uint32_t)n);
    else if (G == 1) hipLaunchKernelGGL((swc_inflate_kernel<1, 0>), grid, block, lds, stream, jobs, (uint32_t)n);
    else if (G == 2) hipLaunchKernelGGL((swc_inflate_kernel<2, 0>), grid, block, lds / 2, stream, jobs, (uint32_t)n);
    else hipLaunchKernelGGL((swc_inflate_kernel<4, 0>), grid, block, lds / 4, stream, jobs, (uint32_t)n);
    return hipGetLastError();
}

// ---- LZ4, one block per lane (many small blocks) -------------------------------------------------
__global__ __launch_bounds__(64) void swc_lz4_lane_kernel(Job* __restrict__ jobs, uint32_t n) {
    uint32_t g = blockIdx.x * kWave + threadIdx.x;
    if (g >= n) return;
    Job job = jobs[g];
    lz4::lz4_block_job(job);
    jobs[g].out_len = job.out_len;
    jobs[g].in_consumed = job.in_consumed;
    jobs[g].status = job.status;
}

hipError_t launch_lz4(Job* jobs, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 grid((unsigned)((n + kWave - 1) / kWave)), block(kWave);
    hipLaunchKernelGGL(swc_lz4_lane_kernel, grid, block, 0, stream, jobs, (uint32_t)n);
    return hipGetLastError();
}

// ---- LZMA / LZMA2, one stream per wavefront ---------------------------------------------------------
// LDS: 28,272 B of u16 probability cells per wave -> 5 streams per CU.  `spill` (may be null) holds
// kLzmaSpillBytes per job for streams with lc+lp > 4.
constexpr size_t kLzmaSpillBytes = (size_t)(0x300u << 12) * 2;

template <bool LZMA2>
__global__ __launch_bounds__(64) void swc_lzma_kernel(Job* __restrict__ jobs, uint32_t n, uint8_t* spill) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lzma_lds[];
    uint32_t g = blockIdx.x;
    if (g >= n) return;
    Job job = jobs[g];
    SWC_AS_GLOBAL uint16_t* sp = spill ? (SWC_AS_GLOBAL uint16_t*)(spill + (size_t)g * kLzmaSpillBytes) : nullptr;
    lzma::lzma_job<kWave>(job, LZMA2, lzma_lds, sp, (int)threadIdx.x);
    if (threadIdx.x == 0) {
        jobs[g].out_len = job.out_len;
        jobs[g].in_consumed = job.in_consumed;
        jobs[g].status = job.status;
    }
}

size_t lzma_spill_bytes_per_job() { return kLzmaSpillBytes; }

hipError_t launch_lzma(bool lzma2, Job* jobs, size_t n, void* spill, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 grid((unsigned)n), block(kWave);
    if (lzma2) hipLaunchKernelGGL(swc_lzma_kernel<true>, grid, block, lzma::kLdsBytesPerWave, stream, jobs, (uint32_t)n, (uint8_t*)spill);
    else hipLaunchKernelGGL(swc_lzma_kernel<false>, grid, block, lzma::kLdsBytesPerWave, stream, jobs, (uint32_t)n, (uint8_t*)spill);
    return hipGetLastError();
}

// ---- BZip2: three stages per block (see bzip2_block.h) ---------------------------------------------
__global__ __launch_bounds__(64) void swc_bzip2_stage1_kernel(const Job* __restrict__ jobs, uint32_t n, uint8_t* ws, size_t lcap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t bz_lds[];
    uint32_t g = blockIdx.x;
    if (g >= n) return;
    Job job = jobs[g];
    bzip2::stage1_job<kWave>(job, reinterpret_cast<bzip2::Stage1Lds*>(bz_lds), bzip2::carve(ws, g, lcap), (int)threadIdx.x);
}

__global__ __launch_bounds__(64) void swc_bzip2_stage2_kernel(uint32_t n, uint8_t* ws, size_t lcap) {
    extern __shared__ __attribute__((aligned(16))) uint8_t bz_lds[];
    uint32_t g = blockIdx.x;
    if (g >= n) return;
    bzip2::stage2_job<kWave>(bzip2::carve(ws, g, lcap), reinterpret_cast<uint32_t*>(bz_lds), (int)threadIdx.x);
}

__global__ __launch_bounds__(64) void swc_bzip2_stage3_kernel(Job* __restrict__ jobs, uint32_t n, uint8_t* ws, size_t lcap) {
    __shared__ uint32_t crc_tab[256];
    for (uint32_t i = threadIdx.x; i < 256; i += kWave) crc_tab[i] = bzip2::crc_table_entry(i);
    __syncthreads();
    uint32_t g = blockIdx.x * kWave + threadIdx.x;
    if (g >= n) return;
    Job job = jobs[g];
    bzip2::stage3_job(job, bzip2::carve(ws, g, lcap), crc_tab);
    jobs[g].out_len = job.out_len;
    jobs[g].in_consumed = job.in_consumed;
    jobs[g].status = job.status;
    jobs[g].aux = job.aux;
}

size_t bzip2_ws_bytes_per_job(size_t lcap) { return bzip2::ws_bytes_per_job(lcap); }

hipError_t launch_bzip2(Job* jobs, size_t n, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (!ws) return hipErrorInvalidValue;
    const size_t per_job = ws_bytes / n;
    const size_t fixed = 32768 + sizeof(bzip2::BlockHeader) + 64 + 16;
    if (per_job <= fixed + 5 * 16) return hipErrorInvalidValue;
    size_t lcap = (per_job - fixed) / 5;
    if (lcap > 16000000) lcap = 16000000;  // i << 8 | c packing of stage 2
    dim3 block(kWave);
    hipLaunchKernelGGL(swc_bzip2_stage1_kernel, dim3((unsigned)n), block, bzip2::kStage1LdsBytes, stream, jobs, (uint32_t)n, (uint8_t*)ws, lcap);
    hipLaunchKernelGGL(swc_bzip2_stage2_kernel, dim3((unsigned)n), block, 256 * kWave * 4, stream, (uint32_t)n, (uint8_t*)ws, lcap);
    hipLaunchKernelGGL(swc_bzip2_stage3_kernel, dim3((unsigned)((n + kWave - 1) / kWave)), block, 0, stream, jobs, (uint32_t)n, (uint8_t*)ws, lcap);
    return hipGetLastError();
}

}  // namespace swc
