// inflate_team.hip -- Deflate phase 1 for launches of FEW streams: a workgroup of kTeamWaves wavefronts per stream (inflate_sync.h,
// "a team of wavefronts on one stream").  Block (64, kTeamWaves): threadIdx.x is the lane, threadIdx.y the wavefront -- the master on
// the job, the helpers on the rounds behind the master's.
//
// A translation unit of its own ON PURPOSE: in one module with swc_inflate_sync_kernel the second caller of the job's helper functions
// changes what the inliner does with them in the FIRST, and the throughput kernel -- 128 registers, none spilled -- came out with three
// spills (measured on the assembly; profiles/r06_experiments.txt).  Here nothing the other kernels are made of changes.
#include <hip/hip_runtime.h>
#include <cstddef>
#include "swc_common.h"
#include "inflate_lane.h"
#include "inflate_sync.h"
#include "launch.h"

namespace swc {

// (the workspace map of kernels.hip: equal strides, or prefix-summed per-job sizes)
struct TeamWsMap {
    uint8_t* base;
    size_t stride;
    const uint64_t* off;
    __device__ uint8_t* area(uint32_t g) const { return base ? base + (off ? (size_t)off[g] : (size_t)g * stride) : nullptr; }
    __device__ size_t bytes(uint32_t g) const { return off ? (size_t)(off[g + 1] - off[g]) : stride; }
};

// `scratch`: (kTeamWaves - 1) * kTeamProvBytes per stream, the helpers' rows
__global__ __launch_bounds__(64 * inflate::kTeamWaves) void swc_inflate_team_kernel(Job* __restrict__ jobs, uint32_t n, TeamWsMap wm, uint8_t* __restrict__ scratch) {
    __shared__ __attribute__((aligned(16))) inflate::SyncLds team_lds[inflate::kTeamWaves];
    __shared__ __attribute__((aligned(16))) inflate::TeamShared team_shared;
    const uint32_t g = blockIdx.x;
    if (g >= n) return;
    inflate::Team tm;
    tm.sh = &team_shared;
    tm.lds = team_lds;
    tm.scratch = (gptr)(scratch + (size_t)g * (inflate::kTeamWaves - 1) * inflate::kTeamProvBytes);
    tm.helpers = inflate::kTeamWaves - 1;
    tm.gen = 0;
    if (threadIdx.y == 0) {
        if (threadIdx.x < (unsigned)inflate::kTeamWaves) team_shared.hgen[threadIdx.x] = 0u;   // (the master's first barrier comes later)
        if (threadIdx.x == 0) team_shared.cmd = 0u;
        Job job = jobs[g];
        inflate::inflate_sync_job<true>(job, &team_lds[0], wm.area(g), wm.bytes(g), (int)threadIdx.x, kWave, nullptr, &tm);
        if (threadIdx.x == 0) {
            jobs[g].out_len = job.out_len;
            jobs[g].in_consumed = job.in_consumed;
            jobs[g].status = job.status;
        }
    } else {
        inflate::team_helper_loop(tm, (int)threadIdx.y);
    }
}

size_t inflate_team_scratch_bytes(size_t n) { return n * (size_t)(inflate::kTeamWaves - 1) * inflate::kTeamProvBytes; }

hipError_t launch_inflate_team(Job* jobs, size_t n, uint8_t* ws, size_t stride, const uint64_t* ws_off, uint8_t* scratch, hipStream_t stream) {
    const TeamWsMap wm{ws, stride, ws_off};
    hipLaunchKernelGGL(swc_inflate_team_kernel, dim3((unsigned)n), dim3(kWave, inflate::kTeamWaves), 0, stream, jobs, (uint32_t)n, wm, scratch);
    return hipGetLastError();
}

}  // namespace swc
