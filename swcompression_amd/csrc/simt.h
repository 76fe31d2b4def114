// simt.h -- the small vocabulary the group-per-stream and wave-per-stream kernels are written in.
//
// A kernel body is a sequence of REGIONS.  Inside a region every thread of the group runs the same code on its own
// per-thread values (PT<V, N>) and may read shared (LDS) data that was written BEFORE the region's opening barrier;
// it may write shared data nobody else reads inside the same region.  Cross-lane steps (scans, ballots, shuffles)
// sit BETWEEN regions and take whole PT values.
//
//   device (gfx950):  a region is a plain block, `t` is threadIdx.x, PT<V, N> is one register value, the closing
//                     SIMT_END_BARRIER is an LDS-only workgroup barrier (s_waitcnt lgkmcnt(0); s_barrier -- global
//                     loads and stores stay in flight across it);
//   host emulation:   a region is a loop over the N threads of the group (in forward, reverse or shuffled order --
//                     simt::g_order -- so that a region that depends on the order of its threads shows up as a test
//                     difference), PT<V, N> is an array.  Single-threaded, deterministic, debuggable.
//
// The host form is TEST INFRASTRUCTURE (tests/host_emu); the shipped library contains the device form only.
#ifndef SWC_SIMT_H
#define SWC_SIMT_H

#include "swc_common.h"

namespace swc {
namespace simt {

// (hipcc compiles this header twice; the host pass sees the device form of the declarations with empty intrinsics)
#if defined(__HIPCC__) && !defined(SWC_HOST_EMULATION)
#define SWC_SIMT_DEVICE_FORM 1
#endif

#if defined(SWC_SIMT_DEVICE_FORM)

template <typename V, int N>
struct PT {
    V v;
    SWC_D V& operator[](int) { return v; }
    SWC_D const V& operator[](int) const { return v; }
};
SWC_D void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
// orders this wave's LDS accesses for the compiler; the LDS executes one wave's instructions in order
SWC_D void wave_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
// all of this wave's vector-memory operations (loads and stores) have completed
// (the builtin, not an asm string: the compiler's own wait insertion then KNOWS that nothing is outstanding behind it and
// does not wait again -- for everything, loads issued later included -- at the first use of a register loaded before it)
SWC_D void vmem_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt and lgkmcnt not waited for (gfx9 encoding)
    asm volatile("" ::: "memory");
#endif
}
#define SIMT_BEGIN(t, N) { const int t = (int)threadIdx.x; (void)t;
#define SIMT_END }
#define SIMT_END_BARRIER } ::swc::simt::lds_barrier();
#define SIMT_END_WAVE } ::swc::simt::wave_fence();
// closes a region after which the threads of the group read GLOBAL memory that others of them wrote: all of the thread's
// memory operations done, then the workgroup barrier (the LDS-only barrier above leaves global stores in flight on purpose)
#define SIMT_END_SYNC } __syncthreads();

#else

template <typename V, int N>
struct PT {
    V v[N];
    V& operator[](int t) { return v[t]; }
    const V& operator[](int t) const { return v[t]; }
};
inline void lds_barrier() {}
inline void wave_fence() {}
inline void vmem_fence() {}
inline int g_order = 0;   // 0 forward, 1 reverse, 2 shuffled
inline int order(int i, int n) {
    if (g_order == 1) return n - 1 - i;
    if (g_order == 2 && (n & (n - 1)) == 0) return (i * 37 + 11) & (n - 1);   // odd multiplier: a permutation of 0..n-1
    return i;
}
#define SIMT_BEGIN(t, N) for (int t##_i_ = 0; t##_i_ < (N); t##_i_++) { const int t = ::swc::simt::order(t##_i_, (N)); (void)t;
#define SIMT_END }
#define SIMT_END_BARRIER }
#define SIMT_END_WAVE }
#define SIMT_END_SYNC }

#endif

// ---- cross-lane steps (between regions) ----------------------------------------------------------------------

// inclusive prefix sum within each 64-lane wave: DPP row shifts + row broadcasts, no LDS
SWC_D uint32_t wave_scan_incl_dev(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
#endif
    return x;
}

// Inclusive prefix sums of two values over a group of N threads (N a multiple of 64).  `ws` = 2 * (N / 64) words of
// LDS.  Contains one barrier when N > 64; on return every thread may read its sums.
template <int N>
SWC_D void group_scan2_incl(PT<uint32_t, N>& x, PT<uint32_t, N>& y, uint32_t* ws) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t a = wave_scan_incl_dev(x.v), b = wave_scan_incl_dev(y.v);
    if (N > 64) {
        if (lane == 63) { ws[wave] = a; ws[N / 64 + wave] = b; }
        lds_barrier();
        // the totals of the waves in front of mine: all partial sums are read at once (wave-uniform addresses: one broadcast read
        // each, no chain of dependent reads) and added under a compare with my wave number
        uint32_t sa = 0, sb = 0;
#pragma unroll
        for (int w = 0; w < N / 64 - 1; w++) {
            const uint32_t va = ws[w], vb = ws[N / 64 + w];
            sa += w < wave ? va : 0u;
            sb += w < wave ? vb : 0u;
        }
        a += sa;
        b += sb;
    }
    x.v = a;
    y.v = b;
#elif defined(SWC_SIMT_DEVICE_FORM)
    (void)x; (void)y; (void)ws;
#else
    (void)ws;
    uint32_t a = 0, b = 0;
    for (int t = 0; t < N; t++) {
        a += x.v[t]; x.v[t] = a;
        b += y.v[t]; y.v[t] = b;
    }
#endif
}

// ---- one wave (N = 64) ---------------------------------------------------------------------------------------
template <int N>
SWC_D uint64_t wave_ballot(const PT<bool, N>& p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(p.v);
#elif defined(SWC_SIMT_DEVICE_FORM)
    (void)p; return 0;
#else
    uint64_t m = 0;
    for (int t = 0; t < N; t++) if (p.v[t]) m |= 1ull << t;
    return m;
#endif
}
template <int N>
SWC_D void wave_scan_incl(PT<uint32_t, N>& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x.v = wave_scan_incl_dev(x.v);
#elif defined(SWC_SIMT_DEVICE_FORM)
    (void)x;
#else
    uint32_t a = 0;
    for (int t = 0; t < N; t++) { a += x.v[t]; x.v[t] = a; }
#endif
}
// inclusive prefix MAXIMUM (unsigned) within the 64-lane wave: the DPP scheme of wave_scan_incl_dev with max in place of add
// (a lane without a source reads the identity 0)
SWC_D uint32_t wave_scan_max_dev(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false));  // row_shr:1
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false));  // row_shr:2
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false));  // row_shr:4
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false));  // row_shr:8
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
#endif
    return x;
}
template <int N>
SWC_D void wave_scan_max_incl(PT<uint32_t, N>& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x.v = wave_scan_max_dev(x.v);
#elif defined(SWC_SIMT_DEVICE_FORM)
    (void)x;
#else
    uint32_t a = 0;
    for (int t = 0; t < N; t++) { a = x.v[t] > a ? x.v[t] : a; x.v[t] = a; }
#endif
}
// y[t] = x[idx[t] % 64]: every lane reads the value of a lane of its choice
template <int N>
SWC_D void wave_gather(PT<uint32_t, N>& y, const PT<uint32_t, N>& x, const PT<uint32_t, N>& idx) {
#if defined(__HIP_DEVICE_COMPILE__)
    y.v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx.v << 2), (int)x.v);
#elif defined(SWC_SIMT_DEVICE_FORM)
    (void)y; (void)x; (void)idx;
#else
    PT<uint32_t, N> r;
    for (int t = 0; t < N; t++) r.v[t] = x.v[idx.v[t] & (N - 1)];
    y = r;
#endif
}
// y[t] = x[t - 1] (lane 0: `fill`)
template <int N>
SWC_D void wave_shift_up(PT<uint32_t, N>& y, const PT<uint32_t, N>& x, uint32_t fill) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = (int)threadIdx.x & 63;
    const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - 1) << 2, (int)x.v);
    y.v = lane == 0 ? fill : v;
#elif defined(SWC_SIMT_DEVICE_FORM)
    (void)y; (void)x; (void)fill;
#else
    uint32_t prev = fill;
    for (int t = 0; t < N; t++) { const uint32_t cur = x.v[t]; y.v[t] = prev; prev = cur; }
#endif
}
// the same through the data-parallel-primitive path of the vector ALU (v_mov_b32_dpp wave_shr:1, lane 0 keeps `fill`): no
// LDS instruction and no select -- for serial chains that shift a register once per step
template <int N>
SWC_D void wave_shift_up_dpp(PT<uint32_t, N>& y, const PT<uint32_t, N>& x, uint32_t fill) {
#if defined(__HIP_DEVICE_COMPILE__)
    y.v = (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)x.v, 0x138, 0xf, 0xf, false);   // wave_shr:1
#elif defined(SWC_SIMT_DEVICE_FORM)
    (void)y; (void)x; (void)fill;
#else
    uint32_t prev = fill;
    for (int t = 0; t < N; t++) { const uint32_t cur = x.v[t]; y.v[t] = prev; prev = cur; }
#endif
}
// y[t] = x[t + s] (the last s lanes: unspecified on the device, zero in the emulation)
template <int N>
SWC_D void wave_shift_down(PT<uint32_t, N>& y, const PT<uint32_t, N>& x, int s) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = (int)threadIdx.x & 63;
    y.v = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + s) & 63) << 2, (int)x.v);
#elif defined(SWC_SIMT_DEVICE_FORM)
    (void)y; (void)x; (void)s;
#else
    for (int t = 0; t < N; t++) y.v[t] = t + s < N ? x.v[t + s] : 0u;
#endif
}
// value of lane `i` (i the same in every lane)
template <int N>
SWC_D uint32_t wave_read(const PT<uint32_t, N>& x, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readlane((int)x.v, i);
#elif defined(SWC_SIMT_DEVICE_FORM)
    (void)x; (void)i; return 0;
#else
    return x.v[i];
#endif
}
SWC_HD int popc64(uint64_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(m);
#else
    return __builtin_popcountll(m);
#endif
}
SWC_HD int popc32(uint32_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(m);
#else
    return __builtin_popcount(m);
#endif
}
SWC_HD int clz64(uint64_t m) {   // m != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)m);
#else
    return __builtin_clzll(m);
#endif
}
SWC_HD int ctz64(uint64_t m) {   // m != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((long long)m) - 1;
#else
    return __builtin_ctzll(m);
#endif
}
// a condition that is the same in every lane, said so: the branch on it becomes a scalar branch (v_cmp, s_cmp_lg_u64 vcc,
// s_cbranch) instead of an exec-mask region with both sides issued -- for code that runs redundantly on all lanes of a wave
SWC_D bool wave_true(bool c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(c) != 0ull;
#else
    return c;
#endif
}
// make a value that is the same in every lane live in a scalar register
SWC_D uint32_t uniform(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}
SWC_D uint64_t uniform(uint64_t v) { return ((uint64_t)uniform((uint32_t)(v >> 32)) << 32) | uniform((uint32_t)v); }

}  // namespace simt
}  // namespace swc
#endif
