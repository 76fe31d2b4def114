// issue_bench.hip -- what does a wave64 instruction cost on gfx950?  (DESIGN's "issue roofline" assumed
// one wave64 VALU instruction per cycle per CU; MI355X_MICROARCH.md quotes v_fma_f32 at 2 cycles per SIMD.)
//
// Every test is a loop of 64 copies of one instruction (or a short group), run by W waves per SIMD on every CU (blocks of
// 256 threads = one wave per SIMD; dynamic LDS sized so that exactly W blocks fit a CU).  Each wave times itself with
// s_memtime (shader cycles); reported per test and W:
//     cyc/inst/wave   = cycles one wave needs per instruction (latency-bound when W = 1 and the chain is dependent)
//     inst/cyc/CU     = 4 SIMDs x W waves x instructions / cycles  (the issue rate the CU sustains)
// "ind" = 8 independent chains, "dep" = one dependent chain.
// Usage: issue_bench [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define REP8(s) s s s s s s s s
#define REP64(s) REP8(REP8(s))

// 8 VGPRs v0..v7 (as %0..%7) + two sources %8 %9; ind: each instruction writes its own register
#define IND8(op, tail) \
    op " %0, %0" tail "\n\t" op " %1, %1" tail "\n\t" op " %2, %2" tail "\n\t" op " %3, %3" tail "\n\t" \
    op " %4, %4" tail "\n\t" op " %5, %5" tail "\n\t" op " %6, %6" tail "\n\t" op " %7, %7" tail "\n\t"
#define DEP8(op, tail) \
    op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" \
    op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t"

struct Res { unsigned long long cycles; };

#define KERNEL_V(name, BODY64)                                                                                            \
    __global__ __launch_bounds__(256) void name(Res* res, int iters, unsigned seed) {                                     \
        extern __shared__ unsigned pad_lds[];                                                                             \
        unsigned v0 = threadIdx.x + seed, v1 = v0 * 3u, v2 = v0 * 5u, v3 = v0 * 7u, v4 = v0 + 11u, v5 = v0 + 13u, v6 = v0 ^ 17u, \
                 v7 = v0 + 19u, a = seed | 1u, b = (seed >> 3) | 5u;                                                      \
        if (seed == 0xFFFFFFFFu) pad_lds[threadIdx.x] = v0;                                                               \
        unsigned long long t0 = __builtin_readcyclecounter();                                                              \
        for (int i = 0; i < iters; i++) {                                                                                 \
            asm volatile(BODY64 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b)); \
        }                                                                                                                 \
        unsigned long long t1 = __builtin_readcyclecounter();                                                              \
        unsigned s = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;                                                               \
        if (s == 0x12345678u) pad_lds[0] = s;                                                                             \
        if ((threadIdx.x & 63) == 0) res[blockIdx.x * 4 + (threadIdx.x >> 6)].cycles = t1 - t0;                           \
    }

// s_memtime based (readcyclecounter lowers to s_memtime / s_memrealtime depending on target; we also time on the host)

KERNEL_V(k_add_ind, REP8(IND8("v_add_u32", ", %8")))
KERNEL_V(k_add_dep, REP8(DEP8("v_add_u32", ", %8")))
KERNEL_V(k_and_ind, REP8(IND8("v_and_b32", ", %8")))
KERNEL_V(k_xor_dep, REP8(DEP8("v_xor_b32", ", %8")))
KERNEL_V(k_alignbit_ind, REP8(IND8("v_alignbit_b32", ", %8, %9")))
KERNEL_V(k_alignbit_dep, REP8(DEP8("v_alignbit_b32", ", %8, %9")))
KERNEL_V(k_bfe_ind, REP8(IND8("v_bfe_u32", ", %8, %9")))
KERNEL_V(k_bfe_dep, REP8(DEP8("v_bfe_u32", ", %8, %9")))
KERNEL_V(k_andor_ind, REP8(IND8("v_and_or_b32", ", %8, %9")))
KERNEL_V(k_lshladd_ind, REP8(IND8("v_lshl_add_u32", ", 3, %9")))
KERNEL_V(k_add3_ind, REP8(IND8("v_add3_u32", ", %8, %9")))
KERNEL_V(k_perm_ind, REP8(IND8("v_perm_b32", ", %8, %9")))
KERNEL_V(k_mul24_ind, REP8(IND8("v_mul_u32_u24", ", %8")))
KERNEL_V(k_mad24_ind, REP8(IND8("v_mad_u32_u24", ", %8, %9")))
KERNEL_V(k_mullo_ind, REP8(IND8("v_mul_lo_u32", ", %8")))
KERNEL_V(k_fma_ind, REP8(IND8("v_fma_f32", ", %8, %9")))
KERNEL_V(k_fma_dep, REP8(DEP8("v_fma_f32", ", %8, %9")))
KERNEL_V(k_pkadd16_ind, REP8(IND8("v_pk_add_u16", ", %8")))
KERNEL_V(k_lshlrev_ind, REP8(IND8("v_lshlrev_b32", ", %8")))   // note: operand order (shift, value): shifts %0 by... see isa; timing only
KERNEL_V(k_bfrev_ind, REP8("v_bfrev_b32 %0, %0\n\tv_bfrev_b32 %1, %1\n\tv_bfrev_b32 %2, %2\n\tv_bfrev_b32 %3, %3\n\tv_bfrev_b32 %4, %4\n\tv_bfrev_b32 %5, %5\n\tv_bfrev_b32 %6, %6\n\tv_bfrev_b32 %7, %7\n\t"))
KERNEL_V(k_bcnt_ind, REP8(IND8("v_bcnt_u32_b32", ", %8")))
KERNEL_V(k_mbcnt_ind, REP8(IND8("v_mbcnt_lo_u32_b32", ", %8")))
// v_cmp + v_cndmask pair (the select idiom of the decode loops): 4 pairs per group
KERNEL_V(k_cmpsel_ind, REP8("v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
                            "v_cmp_lt_u32 vcc, %4, %8\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cndmask_b32 %7, %7, %9, vcc\n\t"))
// DPP move, readlane, readfirstlane, writelane, permlane
KERNEL_V(k_dpp_ind, REP8("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"))

// scalar tests: 8 SGPR chains
#define KERNEL_S(name, BODY64)                                                                                            \
    __global__ __launch_bounds__(256) void name(Res* res, int iters, unsigned seed) {                                     \
        extern __shared__ unsigned pad_lds[];                                                                             \
        unsigned s0 = seed, s1 = seed * 3u, s2 = seed * 5u, s3 = seed * 7u, s4 = seed + 11u, s5 = seed + 13u, s6 = seed ^ 17u, s7 = seed + 19u; \
        unsigned v0 = threadIdx.x, v1 = threadIdx.x * 3u;                                                                 \
        if (seed == 0xFFFFFFFFu) pad_lds[threadIdx.x] = s0;                                                               \
        unsigned long long t0 = __builtin_readcyclecounter();                                                              \
        for (int i = 0; i < iters; i++) {                                                                                 \
            asm volatile(BODY64 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7), "+v"(v0), "+v"(v1) : : "scc", "vcc", "s20", "s21"); \
        }                                                                                                                 \
        unsigned long long t1 = __builtin_readcyclecounter();                                                              \
        unsigned s = s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7 ^ v0 ^ v1;                                                     \
        if (s == 0x12345678u) pad_lds[0] = s;                                                                             \
        if ((threadIdx.x & 63) == 0) res[blockIdx.x * 4 + (threadIdx.x >> 6)].cycles = t1 - t0;                           \
    }
#define SIND8(op, tail) \
    op " %0, %0" tail "\n\t" op " %1, %1" tail "\n\t" op " %2, %2" tail "\n\t" op " %3, %3" tail "\n\t" \
    op " %4, %4" tail "\n\t" op " %5, %5" tail "\n\t" op " %6, %6" tail "\n\t" op " %7, %7" tail "\n\t"
#define SDEP8(op, tail) \
    op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" \
    op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t" op " %0, %0" tail "\n\t"
KERNEL_S(k_sadd_ind, REP8(SIND8("s_add_u32", ", 7")))
KERNEL_S(k_sadd_dep, REP8(SDEP8("s_add_u32", ", 7")))
KERNEL_S(k_sand_ind, REP8(SIND8("s_and_b32", ", 0x7fffffff")))
KERNEL_S(k_slshr_dep, REP8(SDEP8("s_lshr_b32", ", 1")))
// alternate VALU and SALU in ONE wave: does the wave issue both in one slot?
KERNEL_S(k_mix_vs, REP8("v_add_u32 %8, %8, %9\n\ts_add_u32 %0, %0, 7\n\tv_add_u32 %9, %9, %8\n\ts_add_u32 %1, %1, 7\n\t"
                        "v_add_u32 %8, %8, %9\n\ts_add_u32 %2, %2, 7\n\tv_add_u32 %9, %9, %8\n\ts_add_u32 %3, %3, 7\n\t"))
// v_readlane / v_writelane / v_readfirstlane round trips
KERNEL_S(k_readlane, REP8("v_readlane_b32 %0, %8, 5\n\tv_readlane_b32 %1, %9, 6\n\tv_readlane_b32 %2, %8, 7\n\tv_readlane_b32 %3, %9, 8\n\t"
                          "v_readlane_b32 %4, %8, 9\n\tv_readlane_b32 %5, %9, 10\n\tv_readlane_b32 %6, %8, 11\n\tv_readlane_b32 %7, %9, 12\n\t"))
// a dependent VALU -> readfirstlane -> SALU -> VALU loop (the LZMA decision chain crosses the units like this)
KERNEL_S(k_v2s2v_dep, REP8("v_add_u32 %8, %8, %9\n\ts_nop 0\n\tv_readfirstlane_b32 %0, %8\n\ts_add_u32 %0, %0, 7\n\tv_add_u32 %8, %0, %8\n\t"
                           "v_add_u32 %8, %8, %9\n\ts_nop 0\n\tv_readfirstlane_b32 %0, %8\n\ts_add_u32 %0, %0, 7\n\tv_add_u32 %8, %0, %8\n\t"))
// taken / not-taken scalar branches
KERNEL_S(k_branch_nt, REP8("s_cmp_eq_u32 %0, 0x7fffffff\n\ts_cbranch_scc1 1f\n\ts_add_u32 %1, %1, 1\n\t1:\n\ts_cmp_eq_u32 %0, 0x7ffffffe\n\ts_cbranch_scc1 2f\n\ts_add_u32 %2, %2, 1\n\t2:\n\t"
                           "s_cmp_eq_u32 %0, 0x7ffffffd\n\ts_cbranch_scc1 3f\n\ts_add_u32 %3, %3, 1\n\t3:\n\ts_cmp_eq_u32 %0, 0x7ffffffc\n\ts_cbranch_scc1 4f\n\ts_add_u32 %4, %4, 1\n\t4:\n\t"))
KERNEL_S(k_branch_tk, REP8("s_cmp_lg_u32 %0, 0x7fffffff\n\ts_cbranch_scc1 1f\n\ts_add_u32 %1, %1, 1\n\t1:\n\ts_cmp_lg_u32 %0, 0x7ffffffe\n\ts_cbranch_scc1 2f\n\ts_add_u32 %2, %2, 1\n\t2:\n\t"
                           "s_cmp_lg_u32 %0, 0x7ffffffd\n\ts_cbranch_scc1 3f\n\ts_add_u32 %3, %3, 1\n\t3:\n\ts_cmp_lg_u32 %0, 0x7ffffffc\n\ts_cbranch_scc1 4f\n\ts_add_u32 %4, %4, 1\n\t4:\n\t"))
// exec-mask region (the compiler's if): saveexec + branch-over-if-empty + restore
KERNEL_S(k_saveexec, REP8("v_cmp_ne_u32 vcc, 0, %8\n\ts_and_saveexec_b64 s[20:21], vcc\n\ts_cbranch_execz 1f\n\tv_add_u32 %9, %9, %8\n\t1:\n\ts_or_b64 exec, exec, s[20:21]\n\t"
                          "v_cmp_ne_u32 vcc, 0, %9\n\ts_and_saveexec_b64 s[20:21], vcc\n\ts_cbranch_execz 2f\n\tv_add_u32 %8, %9, %8\n\t2:\n\ts_or_b64 exec, exec, s[20:21]\n\t"))

// LDS: dependent pointer chase (latency) and independent random reads (throughput), 4-byte and 1-byte / 2-byte
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(Res* res, int iters, unsigned seed, unsigned words) {
    extern __shared__ unsigned lds[];
    for (unsigned i = threadIdx.x; i < words; i += 256) lds[i] = ((i * 2654435761u + seed) % words) * 4u;   // byte offsets of a pseudo-random successor
    __syncthreads();
    unsigned p0 = (threadIdx.x * 4u) % (words * 4u), p1 = ((threadIdx.x + 64) * 4u) % (words * 4u), p2 = ((threadIdx.x + 128) * 4u) % (words * 4u), p3 = ((threadIdx.x + 192) * 4u) % (words * 4u);
    unsigned acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters * 16; i++) {
        if (MODE == 0) {   // one dependent chain of ds_read_b32
            p0 = *(volatile unsigned*)((char*)lds + p0);
        } else if (MODE == 1) {   // four independent chains
            unsigned a = *(unsigned*)((char*)lds + p0), b = *(unsigned*)((char*)lds + p1), c = *(unsigned*)((char*)lds + p2), d = *(unsigned*)((char*)lds + p3);
            p0 = a; p1 = b; p2 = c; p3 = d;
        } else if (MODE == 2) {   // dependent 16-bit chase (the resolve kernel's cells)
            p0 = (unsigned)*(volatile unsigned short*)((char*)lds + (p0 & ~1u)) & ((words * 4u) - 2u);
        } else {   // four independent 16-bit chains
            unsigned a = *(unsigned short*)((char*)lds + (p0 & ~1u)), b = *(unsigned short*)((char*)lds + (p1 & ~1u)), c = *(unsigned short*)((char*)lds + (p2 & ~1u)), d = *(unsigned short*)((char*)lds + (p3 & ~1u));
            p0 = a & ((words * 4u) - 2u); p1 = b & ((words * 4u) - 2u); p2 = c & ((words * 4u) - 2u); p3 = d & ((words * 4u) - 2u);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    acc = p0 ^ p1 ^ p2 ^ p3;
    if (acc == 0x12345678u) lds[0] = acc;
    if ((threadIdx.x & 63) == 0) res[blockIdx.x * 4 + (threadIdx.x >> 6)].cycles = t1 - t0;
}

typedef void (*KernV)(Res*, int, unsigned);

struct Test { const char* name; KernV k; int insts_per_iter; };

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    printf("device %s, %d CUs, clock %d kHz, iters %d\n", prop.name, cus, clk_khz, iters);
    Res* dres;
    CK(hipMalloc(&dres, sizeof(Res) * 4 * cus * 8));
    std::vector<Res> h(4 * cus * 8);
    std::vector<Test> tests = {
        {"v_add_u32 ind", k_add_ind, 64}, {"v_add_u32 dep", k_add_dep, 64}, {"v_and_b32 ind", k_and_ind, 64}, {"v_xor_b32 dep", k_xor_dep, 64},
        {"v_alignbit_b32 ind", k_alignbit_ind, 64}, {"v_alignbit_b32 dep", k_alignbit_dep, 64}, {"v_bfe_u32 ind", k_bfe_ind, 64}, {"v_bfe_u32 dep", k_bfe_dep, 64},
        {"v_and_or_b32 ind", k_andor_ind, 64}, {"v_lshl_add_u32 ind", k_lshladd_ind, 64}, {"v_add3_u32 ind", k_add3_ind, 64}, {"v_perm_b32 ind", k_perm_ind, 64},
        {"v_mul_u32_u24 ind", k_mul24_ind, 64}, {"v_mad_u32_u24 ind", k_mad24_ind, 64}, {"v_mul_lo_u32 ind", k_mullo_ind, 64},
        {"v_fma_f32 ind", k_fma_ind, 64}, {"v_fma_f32 dep", k_fma_dep, 64}, {"v_pk_add_u16 ind", k_pkadd16_ind, 64},
        {"v_lshlrev_b32 ind", k_lshlrev_ind, 64}, {"v_bfrev_b32 ind", k_bfrev_ind, 64}, {"v_bcnt_u32_b32 ind", k_bcnt_ind, 64}, {"v_mbcnt_lo ind", k_mbcnt_ind, 64},
        {"v_cmp+v_cndmask pairs", k_cmpsel_ind, 64}, {"v_mov_b32_dpp row_shr", k_dpp_ind, 64},
        {"s_add_u32 ind", k_sadd_ind, 64}, {"s_add_u32 dep", k_sadd_dep, 64}, {"s_and_b32 ind", k_sand_ind, 64}, {"s_lshr_b32 dep", k_slshr_dep, 64},
        {"v_add/s_add alternating (one wave)", k_mix_vs, 64}, {"v_readlane_b32", k_readlane, 64}, {"valu->readfirstlane->salu->valu dep (per 5-inst hop)", k_v2s2v_dep, 16},
        {"s_cmp+s_cbranch not taken (+1 salu) per group of 3", k_branch_nt, 32}, {"s_cmp+s_cbranch taken per group of 2", k_branch_tk, 32},
        {"v_cmp+saveexec+cbranch_execz+valu+restore per region", k_saveexec, 16},
    };
    const int Ws[] = {1, 2, 4, 8};
    printf("%-56s", "test (cycles per instruction per wave | instructions per cycle per CU)");
    for (int W : Ws) printf("   W=%d            ", W);
    printf("\n");
    auto run = [&](const char* name, auto launch, double insts) {
        printf("%-56s", name);
        for (int W : Ws) {
            const size_t lds = W == 8 ? 16 * 1024 : (160 * 1024) / W - 512;   // exactly W blocks of 256 threads per CU
            const int grid = cus * W;
            launch(grid, lds);   // warm-up
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            launch(grid, lds);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), dres, sizeof(Res) * 4 * grid, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> c;
            for (int i = 0; i < 4 * grid; i++) c.push_back(h[i].cycles);
            std::sort(c.begin(), c.end());
            const double med = (double)c[c.size() / 2];
            const double cpi = med / insts;
            printf("  %7.2f | %6.3f", cpi, 4.0 * W / cpi);
            (void)ms;
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
        printf("\n");
        fflush(stdout);
    };
    // calibrate the cycle counter against wall time: a long dependent v_add chain
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_add_dep, dim3(cus), dim3(256), 1024, 0, dres, iters * 4, 1u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_add_dep, dim3(cus), dim3(256), 1024, 0, dres, iters * 4, 1u);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), dres, sizeof(Res) * 4 * cus, hipMemcpyDeviceToHost));
        printf("counter calibration: %llu counter ticks in %.3f ms of kernel => %.1f MHz tick (>= : launch overhead is inside the ms)\n", h[0].cycles, ms, h[0].cycles / (ms * 1e3));
    }
    for (auto& t : tests) {
        run(t.name, [&](int grid, size_t lds) { hipLaunchKernelGGL(t.k, dim3(grid), dim3(256), lds, 0, dres, iters, 12345u); }, (double)t.insts_per_iter * iters);
    }
    const unsigned words = 2048;
    auto lds_run = [&](const char* name, auto kern, double per_iter) {
        printf("%-56s", name);
        for (int W : Ws) {
            const size_t lds = W == 8 ? 16 * 1024 : (160 * 1024) / W - 512;
            const int grid = cus * W;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dres, iters / 8, 12345u, words);
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dres, iters / 8, 12345u, words);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), dres, sizeof(Res) * 4 * grid, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> c;
            for (int i = 0; i < 4 * grid; i++) c.push_back(h[i].cycles);
            std::sort(c.begin(), c.end());
            const double med = (double)c[c.size() / 2];
            const double cpi = med / (per_iter * (iters / 8) * 16);
            printf("  %7.2f | %6.3f", cpi, 4.0 * W / cpi);
        }
        printf("\n");
    };
    printf("LDS (cycles per read per wave | reads per cycle per CU); 8 KB table, pseudo-random successors\n");
    lds_run("ds_read_b32 dependent chase (1 chain)", k_lds<0>, 1);
    lds_run("ds_read_b32 4 independent chains", k_lds<1>, 4);
    lds_run("ds_read_u16 dependent chase (1 chain)", k_lds<2>, 1);
    lds_run("ds_read_u16 4 independent chains", k_lds<3>, 4);
    return 0;
}
