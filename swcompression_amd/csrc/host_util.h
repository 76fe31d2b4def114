// host_util.h -- small host-side helpers shared by the C-ABI translation units.
#ifndef SWC_HOST_UTIL_H
#define SWC_HOST_UTIL_H

#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "swc_common.h"
#include "../../include/swc_hip.h"

namespace swc {

// RAII device allocation from the stream-ordered pool, on the calling thread's own stream (hipStreamPerThread: concurrent
// callers do not serialise on the default stream) (hipMalloc / hipFree cost milliseconds
// each, which dominated the single-shot calls: 9.4 ms for one 64 KiB block; the pool keeps up to 1 GiB cached, see
// device_ready()).  ok() is false when the allocation failed (=> SWC_E_DEVICE, never a CPU fallback).
class DevBuf {
public:
    DevBuf() : p_(nullptr), n_(0) {}
    explicit DevBuf(size_t n) : p_(nullptr), n_(0) { alloc(n); }
    ~DevBuf() { release(); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    bool alloc(size_t n) {
        release();
        if (n == 0) n = 1;
        if (hipMallocAsync(&p_, n, hipStreamPerThread) != hipSuccess) { p_ = nullptr; (void)hipGetLastError(); return false; }
        n_ = n;
        return true;
    }
    void release() {
        if (p_) (void)hipFreeAsync(p_, hipStreamPerThread);
        p_ = nullptr; n_ = 0;
    }
    bool ok() const { return p_ != nullptr; }
    uint8_t* u8() const { return static_cast<uint8_t*>(p_); }
    void* ptr() const { return p_; }
    size_t size() const { return n_; }
private:
    void* p_;
    size_t n_;
};

// Page-locked host staging memory of the calling thread, kept between calls (pageable hipMemcpy was most of a small
// single-shot call).  nullptr if the allocation fails (the caller falls back to its own pageable buffer).
uint8_t* pinned_stage(int which, size_t n);

// Host result handed to the caller (released with swc_free).  Large results come from -- and go back to -- a small cache
// of buffers whose pages are already mapped (api.cpp): a fresh 268 MB allocation costs more in page faults (24 ms) than the
// PCIe transfer that fills it (5 ms).
uint8_t* host_result(size_t n);
void host_result_free(void* p);

bool device_is_gfx950(int dev);  // api.cpp: device `dev` exists and is a gfx950 (also prepares its memory pool, once)
bool device_ready();  // api.cpp: true when a gfx950 device is present and selected

// One unit of work for the host-side batch runner (host pointers; the runner stages to HBM).
struct HostUnit {
    const uint8_t* in = nullptr;
    size_t in_len = 0;
    // Optional: the host buffer `in` points into (a container, a multi-member file).  Units with the same base are staged
    // ONCE and address their sub-range of it -- not each its own copy of everything behind its start.
    const uint8_t* base = nullptr;
    size_t base_len = 0;
    size_t cap_hint = 0;         // 0 = use the codec's default policy
    bool cap_exact = false;      // cap_hint is authoritative (declared size): do not grow
    int32_t aux = 0;
    const uint8_t* dict = nullptr;
    size_t dict_len = 0;
    uint64_t extra = 0;          // codec specific (goes to Job::dict_len when dict == nullptr)
    uint64_t dict_value = 0;     // codec specific integer carried in Job::dict when dict == nullptr (LZMA: dictionary size)
    // Optional: where the output is wanted (a place inside the caller's final buffer, `dst_cap` bytes of room).  If the
    // unit's output fits it is copied THERE from the pinned staging buffer -- once -- and `out` stays empty (in_dst, out_size).
    uint8_t* dst = nullptr;
    size_t dst_cap = 0;
    // Optional: a checksum of the output computed ON THE DEVICE in the same launch sequence (swc_checksum kinds of
    // include/swc_hip.h: 1 CRC-32, 2 Adler-32, 3 CRC-64, 4 bzip2 CRC-32, 5 XXH32); one kind per run_units call.
    int sum_kind = 0;
    // results
    std::vector<uint8_t> out;
    bool in_dst = false;         // the output is at `dst`, not in `out`
    size_t out_size = 0;         // bytes of output (wherever they are)
    uint64_t sum = 0;            // the checksum, if sum_valid
    bool sum_valid = false;
    const uint8_t* data() const { return in_dst ? dst : out.data(); }
    size_t size() const { return in_dst ? out_size : out.size(); }
    size_t in_consumed = 0;
    int32_t status = SWC_OK;
    int32_t aux_out = 0;         // codec specific result (bzip2: computed block CRC)
};

// Stage units to HBM, run ONE batched launch of `codec` (re-launching only the units that reported
// SWC_E_CAPACITY with a larger buffer), fetch the results.  Returns SWC_OK or SWC_E_DEVICE.
int run_units(int codec, std::vector<HostUnit>& units);

// SWC_TRACE=1 in the environment: the host-side stages of a call with their wall-clock times, on stderr
struct Trace {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    Trace() { static const bool e = getenv("SWC_TRACE") != nullptr; on = e; t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char* what, size_t bytes = 0) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[swc] %-28s %8.3f ms  (%zu bytes)\n", what, std::chrono::duration<double, std::milli>(now - last).count(), bytes);
        last = now;
    }
};

// swc_stat counters (api.cpp)
void stat_add(int which, long long v);   // 0 launches, 1 units, 2 xz_cache_hits

}  // namespace swc
#endif
