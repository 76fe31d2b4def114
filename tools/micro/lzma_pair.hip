// lzma_pair.hip -- one LZMA stream per wavefront against TWO streams per wavefront.
//
// The shipped kernel (lzma_wave.h) gives a stream a whole wavefront: 64 lanes execute its decision chain redundantly, 16 waves
// (= streams) per CU, vector ALU busy 98 % of the time.  The proposal: lanes 0-31 decode one stream and lanes 32-63 another,
// every value per lane (nothing on the scalar unit: it is shared by the two halves), so that a vector instruction serves two
// streams; with the model cut to 5.2 KB per stream (one literal-coder slot) 30 streams would fit a CU as 15 such waves.
//
// Measured here on the chain itself -- 8-level bit trees (a literal) or an 11-decision path (a match: 1 + 4 + 6) over a model in
// LDS, random input, the kind of the next symbol taken from the stream's own decoded bits -- in three forms:
//   A  one stream per wave, the library's form (wave-uniform C++, branches), 9,840 B of LDS per stream: 16 waves per CU
//   B  two streams per wave, all per lane and branch-free inside a symbol (selects), 2 x 9,840 B per wave: 8 waves per CU
//   C  form B with 2 x 5,200 B per wave: 15 waves = 30 streams per CU
//   A' form A with 5,200 B per stream and 30 waves per CU (possible here: this chain needs few registers; the kernel's 126
//      VGPRs allow 16 waves)
// each with every symbol a literal (the two halves never diverge) and with the kind of a symbol data-dependent (the halves of
// a wave then want different code about half of the time, and the wave executes both).
// Output: stream-decisions per 1,000 cycles per CU (higher is better), cycles per decision of one stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kIn = 256;

// ---- form A ------------------------------------------------------------------------------------------------------------
template <int CELLS, int SLOTS, bool MIX>
__global__ __launch_bounds__(64) void one_stream(unsigned long long* cycles, unsigned long long* decisions, unsigned* sink, int symbols) {
    __shared__ __attribute__((aligned(16))) unsigned short probs[CELLS + 8];
    __shared__ unsigned char in[kIn];
    for (int i = threadIdx.x; i < CELLS; i += 64) probs[i] = 1024;
    for (int i = threadIdx.x; i < kIn; i += 64) in[i] = (unsigned char)(i * 197 + 13 + blockIdx.x);
    __syncthreads();
    unsigned range = 0xFFFFFFFFu, code = 0x12345678u ^ (blockIdx.x * 2654435761u), ip = 0, acc = blockIdx.x, nd = 0;
    auto bit = [&](unsigned short* p) -> unsigned {
        const unsigned pr = *p;
        const unsigned bound = (range >> 11) * pr;
        unsigned b;
        if (code < bound) { *p = (unsigned short)(pr + ((2048u - pr) >> 5)); range = bound; b = 0; }
        else { *p = (unsigned short)(pr - (pr >> 5)); code -= bound; range -= bound; b = 1; }
        if (range < (1u << 24)) { range <<= 8; code = (code << 8) | in[(ip++) & (kIn - 1)]; }
        nd++;
        return b;
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < symbols; s++) {
        if (!MIX || (acc & 1u) == 0u) {   // a literal: 8 decisions in the coder its predecessor picks
            unsigned short* p = probs + 1848 + (acc % SLOTS) * 0x300;
            unsigned m = 1;
            for (int i = 0; i < 8; i++) m = 2 * m + bit(p + m);
            acc = acc * 31 + m;
        } else {                          // a match: isMatch, 4 length decisions, 6 distance-slot decisions
            unsigned m = bit(probs + (acc & 15));
            unsigned short* pl = probs + 64 + (acc & 3) * 16;
            unsigned l = 1;
            for (int i = 0; i < 4; i++) l = 2 * l + bit(pl + (l & 15));
            unsigned short* pd = probs + 256 + (l & 3) * 64;
            unsigned d = 1;
            for (int i = 0; i < 6; i++) d = 2 * d + bit(pd + (d & 63));
            acc = acc * 31 + d + l + m;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; decisions[blockIdx.x] = nd; sink[blockIdx.x] = acc ^ range ^ code ^ ip; }
}

// ---- forms B, C --------------------------------------------------------------------------------------------------------
template <int CELLS, int SLOTS, bool MIX>
__global__ __launch_bounds__(64) void two_streams(unsigned long long* cycles, unsigned long long* decisions, unsigned* sink, int symbols) {
    __shared__ __attribute__((aligned(16))) unsigned short probs[2 * (CELLS + 8)];
    __shared__ unsigned char in[2 * kIn];
    for (int i = threadIdx.x; i < 2 * (CELLS + 8); i += 64) probs[i] = 1024;
    for (int i = threadIdx.x; i < 2 * kIn; i += 64) in[i] = (unsigned char)(i * 197 + 13 + blockIdx.x);
    __syncthreads();
    const unsigned half = threadIdx.x >> 5;
    const unsigned mbase = half * (CELLS + 8) * 2;        // byte address of my stream's model
    const unsigned ibase = half * kIn;
    unsigned range = 0xFFFFFFFFu, code = 0x12345678u ^ ((blockIdx.x * 2 + half) * 2654435761u), ip = 0, acc = blockIdx.x * 2 + half, nd = 0;
    asm volatile("" : "+v"(range), "+v"(code), "+v"(ip), "+v"(acc));
    // one decision, per lane, no branch: cell at byte address a of my model
    auto bit = [&](unsigned a) -> unsigned {
        unsigned short* p = (unsigned short*)((char*)probs + mbase + a);
        const unsigned pr = *p;
        const unsigned bound = (range >> 11) * pr;
        const bool zero = code < bound;
        *p = (unsigned short)(zero ? pr + ((2048u - pr) >> 5) : pr - (pr >> 5));
        range = zero ? bound : range - bound;
        code = zero ? code : code - bound;
        const bool norm = range < (1u << 24);
        const unsigned nb = in[ibase + (ip & (kIn - 1))];
        range = norm ? range << 8 : range;
        code = norm ? (code << 8) | nb : code;
        ip += norm ? 1u : 0u;
        nd++;
        return zero ? 0u : 1u;
    };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < symbols; s++) {
        if (!MIX || (acc & 1u) == 0u) {   // (per lane: the halves of the wave may disagree, the wave then runs both sides)
            const unsigned base = (1848 + (acc % SLOTS) * 0x300) * 2;
            unsigned m = 1;
#pragma unroll
            for (int i = 0; i < 8; i++) m = 2 * m + bit(base + 2 * m);
            acc = acc * 31 + m;
        } else {
            unsigned m = bit((acc & 15) * 2);
            const unsigned bl = (64 + (acc & 3) * 16) * 2;
            unsigned l = 1;
#pragma unroll
            for (int i = 0; i < 4; i++) l = 2 * l + bit(bl + 2 * (l & 15));
            const unsigned bd = (256 + (l & 3) * 64) * 2;
            unsigned d = 1;
#pragma unroll
            for (int i = 0; i < 6; i++) d = 2 * d + bit(bd + 2 * (d & 63));
            acc = acc * 31 + d + l + m;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 31) == 0) { cycles[blockIdx.x * 2 + half] = t1 - t0; decisions[blockIdx.x * 2 + half] = nd; sink[blockIdx.x * 2 + half] = acc ^ range ^ code ^ ip; }
}

template <typename K>
static void run(const char* name, K kernel, int cus, int waves_per_cu, int streams_per_wave, int symbols, unsigned long long* dc, unsigned long long* dd, unsigned* ds) {
    const int grid = cus * waves_per_cu, n = grid * streams_per_wave;
    std::vector<unsigned long long> hc(n), hd(n);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), 0, 0, dc, dd, ds, symbols);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(hc.data(), dc, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hd.data(), dd, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    double per = 0, dec = 0;
    std::vector<double> v(n);
    for (int i = 0; i < n; i++) { v[i] = (double)hc[i] / (double)hd[i]; dec += (double)hd[i]; }
    std::sort(v.begin(), v.end());
    per = v[n / 2];
    // all waves of a CU run at once (grid = what fits): stream-decisions per cycle per CU = streams per CU / cycles per decision
    printf("%-78s %3d x %d  %8.1f  %10.2f\n", name, waves_per_cu, streams_per_wave, per, 1000.0 * waves_per_cu * streams_per_wave / per);
    (void)dec;
}

int main(int argc, char** argv) {
    const int symbols = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long *dc, *dd; unsigned* ds;
    CK(hipMalloc(&dc, sizeof(unsigned long long) * cus * 64));
    CK(hipMalloc(&dd, sizeof(unsigned long long) * cus * 64));
    CK(hipMalloc(&ds, sizeof(unsigned) * cus * 64));
    printf("%d symbols per stream; waves per CU x streams per wave, cycles per decision of a stream (median), stream-decisions per 1,000 cycles per CU\n", symbols);
    run("A  one stream per wave, library form, 9,840 B; literals only", one_stream<4920, 4, false>, cus, 16, 1, symbols, dc, dd, ds);
    run("A' one stream per wave, library form, 5,200 B, 30 waves per CU; literals only", one_stream<2600, 1, false>, cus, 30, 1, symbols, dc, dd, ds);
    run("B  two streams per wave, per lane, 2 x 9,840 B; literals only", two_streams<4920, 4, false>, cus, 8, 2, symbols, dc, dd, ds);
    run("C  two streams per wave, per lane, 2 x 5,200 B (one literal slot); literals only", two_streams<2600, 1, false>, cus, 15, 2, symbols, dc, dd, ds);
    run("A  one stream per wave; literals and matches by the data", one_stream<4920, 4, true>, cus, 16, 1, symbols, dc, dd, ds);
    run("A' one stream per wave, 5,200 B, 30 waves per CU; literals and matches by the data", one_stream<2600, 1, true>, cus, 30, 1, symbols, dc, dd, ds);
    run("B  two streams per wave; literals and matches by the data (halves diverge)", two_streams<4920, 4, true>, cus, 8, 2, symbols, dc, dd, ds);
    run("C  two streams per wave, 2 x 5,200 B; literals and matches by the data", two_streams<2600, 1, true>, cus, 15, 2, symbols, dc, dd, ds);
    return 0;
}
