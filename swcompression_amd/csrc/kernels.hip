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
#include "launch.h"

namespace swc {

// ---- Deflate --------------------------------------------------------------------------------
// LDS: 152 words/lane -> 38,912 B per wave -> 4 resident waves per CU (160 KiB LDS).
__global__ __launch_bounds__(64, 1) void swc_inflate_kernel(Job* __restrict__ jobs, uint32_t n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t g = blockIdx.x * kWave + threadIdx.x;
    if (g >= n) return;
    Job job = jobs[g];
    inflate::inflate_job(job, LaneLds{lds + threadIdx.x});
    jobs[g].out_len = job.out_len;
    jobs[g].in_consumed = job.in_consumed;
    jobs[g].status = job.status;
}

hipError_t launch_inflate(Job* jobs, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    dim3 grid((unsigned)((n + kWave - 1) / kWave)), block(kWave);
    hipLaunchKernelGGL(swc_inflate_kernel, grid, block, inflate::kLdsBytesPerWave, stream, jobs, (uint32_t)n);
    return hipGetLastError();
}

}  // namespace swc
