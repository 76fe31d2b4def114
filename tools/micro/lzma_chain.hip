// lzma_chain.hip -- what does ONE binary decision of the LZMA range decoder cost a wavefront?  (VERDICT r2, next-round item 5:
// "a chain-latency microbenchmark: cycles per binary decision, measured vs the sum of its instruction latencies".)
//
// The decode of a stream is one serial chain of decisions (LZMARangeDecoder.swift:65-80): read an 11-bit probability, split the
// range, compare the code, update the probability, renormalise.  A wavefront owns a stream, so the chain's latency per decision
// times the decisions per byte (2.7 on the P-text units of BASELINE configs[4]) is the kernel's time per byte; occupancy is
// capped by the model in LDS (16 KB per stream: 10 waves per CU).  Three forms of the chain, each decoding 8-level bit trees
// (the shape of a literal) over a model in LDS and random input bytes, per wave W = 1 / 2 / 3 waves per SIMD:
//   0  as the library's Decoder::bit() is written: wave-uniform C++, branch per decision, model cell read inside the chain
//   1  the same with both children of a node read BEFORE the decision that picks one (the library's tree(), round 3)
//   3  form 0 with every value read from LDS declared wave-uniform (v_readfirstlane): range, code, the tree index and all
//      branches on them move to the scalar unit (s_cmp / s_cbranch instead of exec-mask regions)
//   4  form 0 with the cells of TWO levels ahead in flight: the four grandchildren of a node are one aligned 8-byte read
//      issued as soon as the node is known; the pair of children comes out of the previous such read by a select, the
//      cell of the next decision out of that pair -- no LDS read is left in the decision chain
//   2  branch-free and pinned to the vector pipes: every value a VGPR, selects instead of branches, children and the next input
//      byte prefetched -- the chain is then ~9 dependent VALU instructions per decision
// Usage: lzma_chain [trees]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kCells = 7992;          // the model of an lc=3 stream (lzma_wave.h): 15,984 bytes
constexpr int kIn = 256;

struct State { unsigned range, code, ip; };

__device__ __forceinline__ unsigned next_byte(const unsigned char* in, State& s) { return in[(s.ip++) & (kIn - 1)]; }

template <int MODE>
__global__ __launch_bounds__(64) void chain(unsigned long long* cycles, unsigned* sink, int trees) {
    __shared__ __attribute__((aligned(16))) unsigned short probs[kCells + 8];
    __shared__ unsigned char in[kIn];
    for (int i = threadIdx.x; i < kCells; i += 64) probs[i] = 1024;
    for (int i = threadIdx.x; i < kIn; i += 64) in[i] = (unsigned char)(i * 197 + 13 + blockIdx.x);
    __syncthreads();
    State s{0xFFFFFFFFu, 0x12345678u ^ blockIdx.x, 0};
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 4) {
        for (int t = 0; t < trees; t++) {
            const unsigned base = 1848 + (acc & 7) * 0x300;   // a multiple of four cells: the 8-byte reads are aligned
            unsigned short* p = probs + base;
            const uint2* p4 = (const uint2*)p;               // p4[m] = cells 4m .. 4m + 3
            unsigned m = 1;
            const uint2 r0 = p4[0];
            uint2 g = p4[1];                                  // grandchildren of the root: cells 4..7
            unsigned pr = r0.x >> 16;                         // cell 1
            unsigned pair = r0.y;                             // cells 2, 3: the root's children
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned bound = (s.range >> 11) * pr;
                unsigned b;
                if (s.code < bound) { p[m] = (unsigned short)(pr + ((2048u - pr) >> 5)); s.range = bound; b = 0; }
                else { p[m] = (unsigned short)(pr - (pr >> 5)); s.code -= bound; s.range -= bound; b = 1; }
                if (s.range < (1u << 24)) { s.range <<= 8; s.code = (s.code << 8) | next_byte(in, s); }
                pr = b ? pair >> 16 : pair & 0xFFFFu;         // the child taken (level i + 1)
                pair = b ? g.y : g.x;                         // its children (level i + 2), from the read issued one decision ago
                m = 2 * m + b;
                if (i + 3 < 8) g = p4[m];                     // its grandchildren (level i + 3): needed at the end of the NEXT decision
            }
            acc = acc * 31 + m;
        }
    } else if (MODE == 3) {
        unsigned range = __builtin_amdgcn_readfirstlane(s.range), code = __builtin_amdgcn_readfirstlane(s.code), ip = 0;
        for (int t = 0; t < trees; t++) {
            unsigned short* p = probs + 1847 + (acc & 7) * 0x300;
            unsigned m = 1;
            for (int i = 0; i < 8; i++) {
                const unsigned pr = __builtin_amdgcn_readfirstlane((unsigned)p[m]);
                const unsigned bound = (range >> 11) * pr;
                unsigned b;
                if (code < bound) { p[m] = (unsigned short)(pr + ((2048u - pr) >> 5)); range = bound; b = 0; }
                else { p[m] = (unsigned short)(pr - (pr >> 5)); code -= bound; range -= bound; b = 1; }
                if (range < (1u << 24)) { range <<= 8; code = (code << 8) | __builtin_amdgcn_readfirstlane((unsigned)in[(ip++) & (kIn - 1)]); }
                m = 2 * m + b;
            }
            acc = __builtin_amdgcn_readfirstlane(acc * 31 + m);
        }
        s.range = range; s.code = code; s.ip = ip;
    } else if (MODE == 0 || MODE == 1) {
        for (int t = 0; t < trees; t++) {
            unsigned short* p = probs + 1847 + (acc & 7) * 0x300;     // a literal coder picked by what came before
            unsigned m = 1;
            unsigned pr = p[1];
            for (int i = 0; i < 8; i++) {
                unsigned p0 = 0, p1 = 0;
                if (MODE == 1) { p0 = p[(2 * m) & 0x1FF]; p1 = p[(2 * m + 1) & 0x1FF]; }
                else pr = p[m];
                const unsigned bound = (s.range >> 11) * pr;
                unsigned b;
                if (s.code < bound) { p[m] = (unsigned short)(pr + ((2048u - pr) >> 5)); s.range = bound; b = 0; }
                else { p[m] = (unsigned short)(pr - (pr >> 5)); s.code -= bound; s.range -= bound; b = 1; }
                if (s.range < (1u << 24)) { s.range <<= 8; s.code = (s.code << 8) | next_byte(in, s); }
                if (MODE == 1) pr = b ? p1 : p0;
                m = 2 * m + b;
            }
            acc = acc * 31 + m;
        }
    } else {
        // every value pinned to a VGPR (the empty asm makes it opaque: no scalarisation, no branch on it)
        unsigned range = s.range, code = s.code, ip = 0, nb = in[0];
        asm volatile("" : "+v"(range), "+v"(code), "+v"(ip), "+v"(nb));
        for (int t = 0; t < trees; t++) {
            unsigned base = (1847 + (acc & 7) * 0x300) * 2;           // byte address of the coder's cell 0
            asm volatile("" : "+v"(base));
            unsigned m = 1;
            unsigned pr = *(const unsigned short*)((const char*)probs + base + 2);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned a = base + 4 * m;                        // cells 2m and 2m + 1
                const unsigned p0 = *(const unsigned short*)((const char*)probs + (a & 0x7FFF)), p1 = *(const unsigned short*)((const char*)probs + ((a + 2) & 0x7FFF));
                const unsigned bound = (range >> 11) * pr;
                const bool zero = code < bound;
                const unsigned pn = zero ? pr + ((2048u - pr) >> 5) : pr - (pr >> 5);
                *(unsigned short*)((char*)probs + base + 2 * m) = (unsigned short)pn;
                range = zero ? bound : range - bound;
                code = zero ? code : code - bound;
                const bool norm = range < (1u << 24);
                range = norm ? range << 8 : range;
                code = norm ? (code << 8) | nb : code;
                ip += norm ? 1u : 0u;
                nb = in[ip & (kIn - 1)];                               // (the same byte again when nothing was consumed)
                pr = zero ? p0 : p1;
                m = 2 * m + (zero ? 0u : 1u);
            }
            acc = acc * 31 + m;
        }
        s.range = range; s.code = code; s.ip = ip;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; sink[blockIdx.x] = acc ^ s.range ^ s.code ^ s.ip; }
}

int main(int argc, char** argv) {
    const int trees = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* dc; unsigned* ds;
    CK(hipMalloc(&dc, sizeof(unsigned long long) * cus * 16));
    CK(hipMalloc(&ds, sizeof(unsigned) * cus * 16));
    std::vector<unsigned long long> h(cus * 16);
    const char* names[5] = {"0 library form (uniform C++, cell read in the chain)", "1 + children read before the decision", "2 branch-free, all VGPR, children + input prefetched", "3 form 0 on the scalar unit (readfirstlane per cell)", "4 form 0, cells of two levels ahead in flight"};
    printf("%d trees of 8 decisions per wave; cycles per DECISION (median over waves)\n%-58s %10s %10s %10s\n", trees, "form", "4 waves/CU", "8 waves/CU", "10 waves/CU");
    for (int mode = 0; mode < 5; mode++) {
        printf("%-58s", names[mode]);
        for (int wpc : {4, 8, 10}) {
            const int grid = cus * wpc;
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(grid), dim3(64), 0, 0, dc, ds, trees);
                if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(grid), dim3(64), 0, 0, dc, ds, trees);
                if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(grid), dim3(64), 0, 0, dc, ds, trees);
                if (mode == 3) hipLaunchKernelGGL(chain<3>, dim3(grid), dim3(64), 0, 0, dc, ds, trees);
                if (mode == 4) hipLaunchKernelGGL(chain<4>, dim3(grid), dim3(64), 0, 0, dc, ds, trees);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), dc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.begin() + grid);
            printf(" %10.1f", (double)h[grid / 2] / (8.0 * trees));
        }
        printf("\n");
    }
    return 0;
}
