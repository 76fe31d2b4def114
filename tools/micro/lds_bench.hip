// lds_bench.hip -- what an LDS instruction costs on gfx950 when its 64 addresses are what an LZ77 copy makes of them:
// scattered, at any byte alignment, with part of the lanes switched off.  Throughput per CU (16 waves per CU, all issuing
// the same kind of access back to back) in cycles per instruction.    hipcc --offload-arch=gfx950 -O2 lds_bench.hip -o lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t __attribute__((aligned(1), may_alias)) u32u;
typedef uint64_t __attribute__((aligned(1), may_alias)) u64u;
typedef uint16_t __attribute__((aligned(1), may_alias)) u16u;

constexpr int kIters = 512;
// mode: 0 rd32 1 rd64 2 wr32 3 wr64 4 rd8 5 wr8 6 rd128(aligned only) 7 wr16
template <int MODE>
__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ addr, uint32_t nactive, uint64_t* __restrict__ cycles, uint32_t* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[9216];
    const int t = threadIdx.x;
    for (int i = t; i < 9216 / 4; i += 64) ((uint32_t*)lds)[i] = i;
    __syncthreads();
    uint32_t a = addr[t];
    uint32_t acc = 0;
    const bool on = (uint32_t)t < nactive;
    const uint64_t t0 = __builtin_readcyclecounter();
    if (on) {
#pragma unroll 8
        for (int i = 0; i < kIters; i++) {
            if (MODE == 0) acc += *(const u32u*)(lds + a);
            if (MODE == 1) acc += (uint32_t)*(const u64u*)(lds + a);
            if (MODE == 2) *(u32u*)(lds + a) = acc + i;
            if (MODE == 3) *(u64u*)(lds + a) = acc + i;
            if (MODE == 4) acc += lds[a];
            if (MODE == 5) lds[a] = (uint8_t)(acc + i);
            if (MODE == 6) { uint4 v = *(const uint4*)(lds + (a & ~15u)); acc += v.x + v.w; }
            if (MODE == 7) *(u16u*)(lds + a) = (uint16_t)(acc + i);
            if (MODE == 8) { const uint32_t* p = (const uint32_t*)(lds + (a & ~3u)); acc += __builtin_amdgcn_alignbyte(p[1], p[0], a); }          // rd32 from two aligned dwords
            if (MODE == 9) { const uint64_t v = *(const u64u*)(lds + a); acc += (uint32_t)v ^ (uint32_t)(v >> 32); }                               // a real 64-bit read
            if (MODE == 10) { const uint32_t* p = (const uint32_t*)(lds + (a & ~3u)); acc += __builtin_amdgcn_alignbyte(p[1], p[0], a) ^ __builtin_amdgcn_alignbyte(p[2], p[1], a); }
            if (MODE >= 12) {
                const uint32_t aa = (a & 0x80000000u) ? 8192u + 8u * (uint32_t)t : a;
                if (MODE == 12) *(u32u*)(lds + aa) = acc + i;
                if (MODE == 13) *(u64u*)(lds + aa) = acc + i;
                if (MODE == 15) lds[aa] = (uint8_t)(acc + i);
                if (MODE == 17) *(u16u*)(lds + aa) = (uint16_t)(acc + i);
                a = (a & 0x80000000u) | ((a + 260u) & 8191u);
                asm volatile("" : "+v"(a), "+v"(acc));
                continue;
            }
            a = (a + 260u) & 8191u;   // (keeps the alignment class of the address: 260 = 4 * 65)
            asm volatile("" : "+v"(a), "+v"(acc));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter();
    if (t == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc + lds[t];
}
template <int MODE>
static double run(const std::vector<uint32_t>& h, uint32_t nactive, int blocks) {
    uint32_t *d, *sink; uint64_t* cyc;
    hipMalloc(&d, 256); hipMalloc(&sink, 4); hipMalloc(&cyc, blocks * 8);
    hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, nactive, cyc, sink);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, nactive, cyc, sink);
    hipDeviceSynchronize();
    std::vector<uint64_t> c(blocks);
    hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : c) s += (double)x;
    hipFree(d); hipFree(sink); hipFree(cyc);
    return s / blocks / kIters;   // cycles per instruction as ONE wave sees it (16 waves share the CU's LDS)
}
int main() {
    // ---- stores where only a part of the lanes store for real (scattered, any alignment) and the others hit their trash slot
    {
        printf("stores with a part of the lanes on their trash slot (8 KiB + 8 * lane), 16 waves per CU: cycles of the LDS per instruction\n");
        for (int real : {64, 48, 32, 16, 8, 0}) {
            std::vector<uint32_t> h(64);
            uint32_t seed2 = 777;
            for (int t = 0; t < 64; t++) { seed2 = seed2 * 1664525u + 1013904223u; h[t] = t < real ? ((seed2 >> 8) % 8000) : 0x80000000u; }
            printf("  real lanes %2d:", real);
            printf("  wr32 %.1f", run<12>(h, 64, 256 * 16) / 16.0);
            printf("  wr64 %.1f", run<13>(h, 64, 256 * 16) / 16.0);
            printf("  wr8 %.1f", run<15>(h, 64, 256 * 16) / 16.0);
            printf("  wr16 %.1f\n", run<17>(h, 64, 256 * 16) / 16.0);
        }
    }
    const char* names[8] = {"ds_read_b32", "ds_read_b64", "ds_write_b32", "ds_write_b64", "ds_read_u8", "ds_write_b8", "ds_read_b128 (aligned)", "ds_write_b16"};
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int pat = 0; pat < 4; pat++) {
        std::vector<uint32_t> h(64);
        for (int t = 0; t < 64; t++) {
            if (pat == 0) h[t] = 8 * t;                                // lane-linear, aligned
            if (pat == 1) h[t] = (rnd() % 2000) * 4;                    // scattered, dword aligned
            if (pat == 2) h[t] = (rnd() % 8000);                        // scattered, any alignment
            if (pat == 3) h[t] = 600 + 9 * t + (rnd() % 3);             // an LZ77 group: consecutive records, ~9 bytes apart
        }
        const char* pn[4] = {"lane-linear aligned", "scattered aligned", "scattered any alignment", "records 9 bytes apart"};
        for (int na : {64, 16, 4}) {
            printf("%-26s lanes %2d :", pn[pat], na);
            for (int m = 0; m < 11; m++) {
                double one, many;
                switch (m) {
#define C(M) case M: one = run<M>(h, na, 256); many = run<M>(h, na, 256 * 16); break;
                    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10)
                }
                printf("  %s %.1f/%.1f", m == 0 ? "rd32" : m == 1 ? "rd64" : m == 2 ? "wr32" : m == 3 ? "wr64" : m == 4 ? "rd8" : m == 5 ? "wr8" : m == 6 ? "rd128" : m == 7 ? "wr16" : m == 8 ? "rd32via2" : m == 9 ? "rd64real" : "rd64via3", one, many / 16.0);
            }
            printf("\n");
        }
    }
    (void)names;
    printf("(first figure: cycles per instruction of a wave alone on its CU = latency-bound chain of independent accesses;\n second: 16 waves per CU, cycles per instruction per wave / 16 = cycles of the CU's LDS per instruction)\n");
    return 0;
}
