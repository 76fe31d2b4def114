// launch.h -- host-side launch wrappers implemented in kernels.hip.
#ifndef SWC_LAUNCH_H
#define SWC_LAUNCH_H
#include <hip/hip_runtime.h>
#include "swc_common.h"
namespace swc {
// ws_off != nullptr: device array of n + 1 prefix-summed per-job workspace offsets (then ws_bytes is ignored)
hipError_t launch_inflate(Job* jobs, size_t n, void* ws, size_t ws_bytes, hipStream_t stream, const uint64_t* ws_off = nullptr);
// inflate_team.hip: phase 1 with a team of wavefronts per stream (launches of few streams); `scratch`: inflate_team_scratch_bytes(n)
size_t inflate_team_scratch_bytes(size_t n);
hipError_t launch_inflate_team(Job* jobs, size_t n, uint8_t* ws, size_t stride, const uint64_t* ws_off, uint8_t* scratch, hipStream_t stream);
size_t inflate_ws_bytes_per_job(uint64_t cap);
hipError_t launch_crc32(const Job* jobs, size_t n, uint32_t* crcs, hipStream_t stream);
hipError_t launch_delta(Job* jobs, size_t n, hipStream_t stream);
hipError_t launch_checksum(int kind, const Job* jobs, size_t n, uint64_t* sums, hipStream_t stream);
void set_profile_buffer(void* p);
void set_phase_timing(int on);
void set_lzma_coder_cache(int on);
void set_lz_copier(int v);
void set_deflate_team(int v);
void set_bzip2_hot_cxx(int v);
void set_bzip2_team_walk(int v);
void set_bzip2_team_per_cu(int v);
int last_phase_ms(float* ms, int cap);
hipError_t launch_lz4(Job* jobs, size_t n, void* ws, size_t ws_bytes, hipStream_t stream, const uint64_t* ws_off = nullptr);
size_t lz4_ws_bytes_per_job(uint64_t cap);
hipError_t launch_lz4_compress(Job* jobs, size_t n, hipStream_t stream);
hipError_t launch_deflate_compress(Job* jobs, size_t n, hipStream_t stream);
hipError_t launch_lzma(bool lzma2, Job* jobs, size_t n, void* spill, hipStream_t stream);
size_t lzma_spill_bytes_per_job();
hipError_t launch_bzip2(Job* jobs, size_t n, void* ws, size_t ws_bytes, hipStream_t stream);
size_t bzip2_ws_bytes_per_job(size_t lcap);
}
#endif
