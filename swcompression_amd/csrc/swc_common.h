// swc_common.h -- shared declarations for the MI355X decode engine (device code + host launcher).
//
// The per-lane decoders in this directory are written as plain sequential C++ ("one lane = one
// compressed stream") so that the same source compiles for gfx950 (hipcc) and, with
// -DSWC_HOST_EMULATION, for the host (g++) where tests/ run every lane of a wave sequentially and
// compare against the oracle without a GPU.  The host build is test infrastructure only; the
// shipped library contains the device build and nothing else.
#ifndef SWC_COMMON_H
#define SWC_COMMON_H

#include <stdint.h>
#include <stddef.h>
#include "../../include/swc_status.h"

#if defined(__HIPCC__) && !defined(SWC_HOST_EMULATION)
#include <hip/hip_runtime.h>
#define SWC_HD __host__ __device__ __forceinline__
#define SWC_D __device__ __forceinline__
#else
#define SWC_HD inline
#define SWC_D inline
#endif

// Makes a value opaque to the optimiser at no run-time cost.  Used where register-resident tables are read through
// select trees: without it LLVM folds `c ? a[i] : a[j]` into a load at a DYNAMIC offset, which pins the whole
// table (and the struct around it) in scratch memory instead of VGPRs.
#if defined(__HIP_DEVICE_COMPILE__)
#define SWC_OPAQUE(x) asm("" : "+v"(x))
#define SWC_OPAQUE_S(x) asm("" : "+s"(x))   // the same for a wave-uniform value (stays in a scalar register)
#else
#define SWC_OPAQUE(x) ((void)0)
#define SWC_OPAQUE_S(x) ((void)0)
#endif

namespace swc {

constexpr int kWave = 64;  // CDNA wavefront width; LDS tables are interleaved at this stride

// One decode job.  Layout == `swc_job` of include/swc_hip.h (checked by static_assert in api.cpp).
struct Job {
    const uint8_t* in;
    uint64_t in_len;
    uint8_t* out;
    uint64_t out_cap;
    uint64_t out_len;      // result: bytes produced (or required, for SWC_E_CAPACITY on size-countable codecs)
    uint64_t in_consumed;  // result: bytes of `in` consumed (after the caller-side align())
    int32_t status;        // result: swc_status
    int32_t aux;           // codec specific input (LZMA2: dictionary-size byte; LZMA: packed lc/lp/pb; LZ4: unused)
    const uint8_t* dict;   // optional prefix dictionary (LZ4) / reserved
    uint64_t dict_len;     // LZ4: dictionary length; LZMA: declared uncompressed size (or ~0 = unknown)
};

SWC_HD uint32_t brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
#endif
}

// Pointers into HBM.  On the device they carry the global address space so that the compiler emits
// global_load/global_store instead of FLAT instructions (pointers loaded from a job record are generic
// otherwise; FLAT ops tie up both vmcnt and lgkmcnt and serialise against the LDS table reads).
#if defined(__HIP_DEVICE_COMPILE__)
#define SWC_AS_GLOBAL __attribute__((address_space(1)))
#else
#define SWC_AS_GLOBAL
#endif
typedef const SWC_AS_GLOBAL uint8_t* gcptr;
typedef SWC_AS_GLOBAL uint8_t* gptr;
typedef uint32_t __attribute__((aligned(1), may_alias)) u32_unaligned;
typedef uint64_t __attribute__((aligned(1), may_alias)) u64_unaligned;
typedef uint16_t __attribute__((aligned(1), may_alias)) u16_unaligned;

// gfx950 global loads/stores are unaligned-capable (hipcc emits plain global_load_dword for these)
SWC_HD uint32_t load_u32(gcptr p) { return *(const SWC_AS_GLOBAL u32_unaligned*)p; }
SWC_HD uint64_t load_u64(gcptr p) { return *(const SWC_AS_GLOBAL u64_unaligned*)p; }
SWC_HD void store_u32(gptr p, uint32_t v) { *(SWC_AS_GLOBAL u32_unaligned*)p = v; }
SWC_HD void store_u64(gptr p, uint64_t v) { *(SWC_AS_GLOBAL u64_unaligned*)p = v; }
struct __attribute__((packed, may_alias)) u128_any { uint32_t x, y, z, w; };
SWC_HD void store_u128_a4(gptr p, uint32_t x, uint32_t y, uint32_t z, uint32_t w) { *(SWC_AS_GLOBAL u128_any*)p = u128_any{x, y, z, w}; }   // 16 bytes at any alignment

// Workgroup barrier of the group-per-stream checksum kernels.  The host emulation build runs the T "threads" of a group
// as real host threads and plugs its own barrier in here (tests/host_emu/emu.cpp).
#if defined(__HIP_DEVICE_COMPILE__)
SWC_D void group_sync() { __syncthreads(); }
#elif defined(SWC_HOST_EMULATION)
inline thread_local void (*emu_group_sync)() = nullptr;
SWC_D void group_sync() { if (emu_group_sync) emu_group_sync(); }
#else
SWC_D void group_sync() {}
#endif

// Per-lane view of an LDS region interleaved at wave stride: word j of this lane lives at
// base[j * 64 + lane], so any per-lane index pattern is bank-conflict free for 32-bit accesses
// (bank = lane % 32, the two 32-lane halves are serviced separately).
struct LaneLds {
    uint32_t* p;  // base + stream column
    int stride;   // streams per wave (64 when every lane owns a stream)
    SWC_HD uint32_t get(int j) const { return p[j * stride]; }
    SWC_HD void set(int j, uint32_t v) const { p[j * stride] = v; }
};

}  // namespace swc
#endif
