// bzip2_compress.hip -- the device executor of BZip2 compression (bzip2_comp.h has the stages and the driver).
//
// Every stage of bzip2_comp.h is a functor; here they become kernels:
//   each(m, f)        one THREAD per element (the steps of the prefix-doubling sort), 256 per workgroup;
//   per_block(nb, f)  one WAVEFRONT per bzip2 block (rle1, gather, mtf, emit, join);
//   sort_pairs        rocprim::radix_sort_pairs over the key bits that are in use (the block index and seven bytes in the first
//                     round, two ranks of ceil(log2(total)) bits after it);
//   scan_max/scan_sum rocprim::inclusive_scan (maximum) / exclusive_scan (plus).
// rocPRIM is header-only: these are device kernels compiled into this library for gfx950, not a dependency at run time.
// Memory comes from the stream-ordered pool on the calling thread's stream and goes back at the end of every launch group.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <thread>
#include <system_error>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "bzip2_comp.h"
#include "host_util.h"
#include "launch.h"

namespace swc {
namespace {

template <class F>
__global__ __launch_bounds__(256) void swc_bz2c_each_kernel(F f, uint32_t m) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < m) f(i);
}
template <class F>
__global__ __launch_bounds__(64) void swc_bz2c_block_kernel(F f) {
    __shared__ typename F::Lds lds;
    f.template run<64>(blockIdx.x, &lds);
}

struct DeviceExec {
    hipStream_t s = hipStreamPerThread;
    std::vector<void*> mem;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    bool failed = false;

    // SWC_BZ2C_TRACE=1: wall time of every stage of a launch group on stderr (each mark waits for the stream: a diagnosis aid)
    bool trace = getenv("SWC_BZ2C_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0;
    void mark(const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(s);
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[bz2c] %-14s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
    void note(const char* what, uint32_t a, uint32_t b) { if (trace) { mark("..."); fprintf(stderr, "[bz2c] %s %u %u\n", what, a, b); } }
    ~DeviceExec() { end_chunk(); }
    void begin_chunk() { t0 = std::chrono::steady_clock::now(); }
    void end_chunk() {
        for (void* p : mem) (void)hipFreeAsync(p, s);
        mem.clear();
        temp = nullptr; temp_bytes = 0;
    }
    uint8_t* result_alloc(size_t n) { return host_result(n); }
    void result_free(uint8_t* q) { host_result_free(q); }
    bool ok(hipError_t e) { if (e != hipSuccess) { failed = true; (void)hipGetLastError(); } return e == hipSuccess; }
    void* alloc(size_t n) {
        void* p = nullptr;
        if (!ok(hipMallocAsync(&p, n ? n : 16, s))) return nullptr;
        mem.push_back(p);
        return p;
    }
    // Large transfers go through the calling thread's page-locked staging buffers (host_util.h): a pageable source made the
    // same 32 MiB upload take anything between 1 and 24 ms.  Several threads fill the buffer (one core copies about 10 GB/s).
    static void copy_threads(uint8_t* dst, const uint8_t* src, size_t n) {
        const size_t nt = std::min<size_t>(8, n >> 22);
        if (nt < 2) { memcpy(dst, src, n); return; }
        std::vector<std::thread> th;
        const size_t per = (n / nt + 63) & ~(size_t)63;
        size_t done = 0;   // bytes whose copy a thread has taken
        try {
            for (size_t t = 0; t < nt; t++) {
                const size_t lo = std::min(n, t * per), hi = t + 1 == nt ? n : std::min(n, (t + 1) * per);
                th.emplace_back([=] { if (hi > lo) memcpy(dst + lo, src + lo, hi - lo); });
                done = hi;
            }
        } catch (const std::system_error&) {}   // no more threads to be had: this one copies the rest (never std::terminate)
        if (done < n) memcpy(dst + done, src + done, n - done);
        for (auto& t : th) t.join();
    }
    void upload(void* d, const void* h, size_t n) {
        if (!n) return;
        uint8_t* stage = n >= ((size_t)1 << 20) ? pinned_stage(0, n) : nullptr;
        if (stage) {
            copy_threads(stage, (const uint8_t*)h, n);
            ok(hipMemcpyAsync(d, stage, n, hipMemcpyHostToDevice, s));
        } else {
            ok(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
        }
        ok(hipStreamSynchronize(s));          // (the source is the caller's to change, the staging buffer the next upload's)
    }
    void download(void* h, const void* d, size_t n) {
        if (!n) return;
        uint8_t* stage = n >= ((size_t)1 << 20) ? pinned_stage(1, n) : nullptr;
        ok(hipMemcpyAsync(stage ? (void*)stage : h, d, n, hipMemcpyDeviceToHost, s));
        ok(hipStreamSynchronize(s));
        if (stage) copy_threads((uint8_t*)h, stage, n);
    }
    void zero(void* d, size_t n) { if (n) ok(hipMemsetAsync(d, 0, n, s)); }
    template <class F> void each(uint32_t m, const F& f) {
        if (!m) return;
        hipLaunchKernelGGL((swc_bz2c_each_kernel<F>), dim3((m + 255u) / 256u), dim3(256), 0, s, f, m);
        ok(hipGetLastError());
    }
    template <class F> void per_block(uint32_t nb, const F& f) {
        if (!nb) return;
        hipLaunchKernelGGL((swc_bz2c_block_kernel<F>), dim3(nb), dim3(64), 0, s, f);
        ok(hipGetLastError());
    }
    bool need_temp(size_t n) {
        if (n <= temp_bytes) return true;
        temp = alloc(n);
        temp_bytes = temp ? n : 0;
        return temp != nullptr;
    }
    int sort_pairs(uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, uint32_t m, int bits) {
        size_t need = 0;
        if (!ok(rocprim::radix_sort_pairs(nullptr, need, kin, kout, vin, vout, m, 0u, (unsigned)bits, s))) return 1;
        if (!need_temp(need)) return 1;
        size_t have = temp_bytes;
        return ok(rocprim::radix_sort_pairs(temp, have, kin, kout, vin, vout, m, 0u, (unsigned)bits, s)) ? 0 : 1;
    }
    int scan_max(uint32_t* x, uint32_t m) {
        size_t need = 0;
        if (!ok(rocprim::inclusive_scan(nullptr, need, x, x, (size_t)m, rocprim::maximum<uint32_t>(), s))) return 1;
        if (!need_temp(need)) return 1;
        size_t have = temp_bytes;
        return ok(rocprim::inclusive_scan(temp, have, x, x, (size_t)m, rocprim::maximum<uint32_t>(), s)) ? 0 : 1;
    }
    int scan_sum(const uint32_t* in, uint32_t* out, uint32_t m) {
        size_t need = 0;
        if (!ok(rocprim::exclusive_scan(nullptr, need, in, out, 0u, (size_t)m, rocprim::plus<uint32_t>(), s))) return 1;
        if (!need_temp(need)) return 1;
        size_t have = temp_bytes;
        return ok(rocprim::exclusive_scan(temp, have, in, out, 0u, (size_t)m, rocprim::plus<uint32_t>(), s)) ? 0 : 1;
    }
    // CheckSums.bzip2crc32 of every block's raw bytes (BZip2+Compress.swift:54): the checksum kernel of the decoders
    int block_crcs(const uint8_t* d_raw, const uint32_t* off, uint32_t nb, uint32_t* crcs) {
        std::vector<Job> jobs(nb);
        memset(jobs.data(), 0, sizeof(Job) * nb);
        for (uint32_t b = 0; b < nb; b++) {
            jobs[b].out = const_cast<uint8_t*>(d_raw) + off[b];
            jobs[b].out_len = jobs[b].out_cap = off[b + 1] - off[b];
        }
        Job* d_jobs = (Job*)alloc(sizeof(Job) * nb);
        uint64_t* d_sums = (uint64_t*)alloc(8 * (size_t)nb);
        if (!d_jobs || !d_sums) return 1;
        upload(d_jobs, jobs.data(), sizeof(Job) * nb);
        if (!ok(launch_checksum(4, d_jobs, nb, d_sums, s))) return 1;
        std::vector<uint64_t> sums(nb);
        download(sums.data(), d_sums, 8 * (size_t)nb);
        for (uint32_t b = 0; b < nb; b++) crcs[b] = (uint32_t)sums[b];
        return failed ? 1 : 0;
    }
};

}  // namespace

// *out: from host_result (swc_free releases it)
int bzip2_compress_device(const uint8_t* data, size_t len, int level, uint8_t** out, size_t* out_len) {
    DeviceExec x;
    *out = nullptr; *out_len = 0;
    const int st = bz2c::compress_stream(x, data, len, level, out, out_len);
    if (st == SWC_OK && x.failed) { host_result_free(*out); *out = nullptr; *out_len = 0; }
    return st != SWC_OK || x.failed ? SWC_E_DEVICE : SWC_OK;
}

}  // namespace swc
