// gather_bench.hip -- how fast can the chip chase pointers through per-block permutations?  (BZip2 inverse BWT shape:
// 10,240 blocks x 900,000 u32 entries.)  Each walker does `steps` dependent 4-byte gathers inside its block's array.
//   mode 0: one walker (lane) per block          -- the current stage 3
//   mode 1: W walkers per block, one workgroup of W threads per block
// Usage: gather_bench <blocks> <n> <W> [lds_kb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void replicate(const uint32_t* src, uint32_t* dst, size_t n, size_t blocks) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = n * blocks;
    for (; i < total; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i % n];
}

__global__ void walk_lane_per_block(const uint32_t* P, size_t n, uint32_t blocks, uint32_t steps, uint32_t* sink) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= blocks) return;
    const uint32_t* p = P + (size_t)b * n;
    uint32_t end = b % n, acc = 0;
    for (uint32_t i = 0; i < steps; i++) { uint32_t v = p[end]; end = v >> 8; acc += v & 255; }
    sink[b] = acc + end;
}

__global__ void walk_group_per_block(const uint32_t* P, size_t n, uint32_t steps, uint32_t* sink) {
    extern __shared__ uint32_t pad[];
    const uint32_t* p = P + (size_t)blockIdx.x * n;
    uint32_t end = (uint32_t)(((uint64_t)threadIdx.x * n) / blockDim.x), acc = 0;
    for (uint32_t i = 0; i < steps; i++) { uint32_t v = p[end]; end = v >> 8; acc += v & 255; }
    if (acc + end == 0xFFFFFFFFu) pad[0] = 1;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + end;
}

// mode 2: 32 workgroups of W threads per block, placed on ONE XCD (workgroup w runs on XCD w % 8, observed): the block's
// pointer array (3.6 MB) stays inside that XCD's 4 MiB L2 while 32 x W walkers chase through it.
__global__ void walk_xcd_per_block(const uint32_t* P, size_t n, uint32_t steps, uint32_t blocks, uint32_t* sink) {
    extern __shared__ uint32_t pad[];
    const uint32_t w = blockIdx.x, x = w & 7, q = w >> 3;
    const uint32_t b = (q / 32) * 8 + x, sub = q % 32;
    if (b >= blocks) return;
    const uint32_t* p = P + (size_t)b * n;
    const uint32_t walker = sub * blockDim.x + threadIdx.x, walkers = 32 * blockDim.x;
    uint32_t end = (uint32_t)(((uint64_t)walker * n) / walkers), acc = 0;
    for (uint32_t i = 0; i < steps; i++) { uint32_t v = p[end]; end = v >> 8; acc += v & 255; }
    if (acc + end == 0xFFFFFFFFu) pad[0] = 1;
    sink[(w * blockDim.x + threadIdx.x) & 0xFFFFF] = acc + end;
}

// mode 3: what a decoder could do with the L2-resident rate.  The cycle of every block is cut at the indices that are multiples
// of M (segments of geometrically distributed length, mean M).  The chip is eight TEAMS (one per XCD, told apart by the XCC_ID
// hardware register); a team takes the blocks x, x + 8, x + 16 ... one after the other, its walkers draw (block, segment) tickets
// from the team's counter -- no barrier between blocks: the stragglers of a block finish while the others are on the next one.
__global__ void walk_tickets(const uint32_t* P, size_t n, uint32_t M, uint32_t blocks, uint32_t* counters, uint32_t* sink, uint32_t* xcd_seen, uint8_t* segbuf, uint32_t* seginfo) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if (threadIdx.x == 0) atomicAdd(&xcd_seen[xcc], 1u);
    const uint32_t segs = (uint32_t)((n + M - 1) / M);
    const uint32_t team_blocks = (blocks + 7u - xcc) / 8u;            // blocks xcc, xcc + 8, ...
    const uint64_t total = (uint64_t)team_blocks * segs;
    uint32_t acc = 0;
    for (;;) {
        const uint32_t t = atomicAdd(&counters[xcc * 32], 1u);       // (a cache line per team)
        if (t >= total) break;
        const uint32_t b = xcc + 8u * (t / segs), sg = t % segs;
        const uint32_t* p = P + (size_t)b * n;
        uint32_t end = sg * M, steps = 0;
        if (segbuf) {   // as the decoder would: the bytes of a segment into its buffer (4 M bytes each), eight per store; length and successor
            uint8_t* buf = segbuf + ((size_t)b * segs + sg) * (4u * M);
            uint64_t w = 0;
            do {
                const uint32_t v = p[end]; end = v >> 8;
                w |= (uint64_t)(v & 255u) << (8u * (steps & 7u));
                if ((steps & 7u) == 7u) { if (steps < 4u * M) __builtin_nontemporal_store(w, (uint64_t*)(buf + (steps - 7u))); w = 0; }
                steps++;
            } while (end % M != 0);
            if ((steps & 7u) && steps < 4u * M) __builtin_nontemporal_store(w, (uint64_t*)(buf + (steps & ~7u)));
            seginfo[((size_t)b * segs + sg) * 2] = steps;
            seginfo[((size_t)b * segs + sg) * 2 + 1] = end / M;
        } else {
            do { const uint32_t v = p[end]; end = v >> 8; acc += v & 255; steps++; } while (end % M != 0);
        }
        acc += steps;
    }
    sink[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = acc;
}

int main(int argc, char** argv) {
    size_t blocks = argc > 1 ? atol(argv[1]) : 10240, n = argc > 2 ? atol(argv[2]) : 900000;
    int W = argc > 3 ? atoi(argv[3]) : 64, lds_kb = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<uint32_t> perm(n), P(n);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(12345);
    std::shuffle(perm.begin(), perm.end(), rng);
    for (size_t i = 0; i < n; i++) P[perm[i]] = (perm[(i + 1) % n] << 8) | (uint32_t)(i & 255);   // one n-cycle
    uint32_t *d_src, *d_P, *d_sink;
    CK(hipMalloc(&d_src, n * 4));
    CK(hipMalloc(&d_P, n * blocks * 4));
    CK(hipMalloc(&d_sink, blocks * 1024 * 4));
    CK(hipMemcpy(d_src, P.data(), n * 4, hipMemcpyHostToDevice));
    replicate<<<8192, 256>>>(d_src, d_P, n, blocks);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        uint64_t gathers;
        if (argc > 6) {   // mode 3: tickets, argv[5] = M, argv[6] = workgroups per CU
            uint32_t M = (uint32_t)atoi(argv[5]);
            int per_cu = atoi(argv[6]);
            uint32_t *d_cnt, *d_seen;
            CK(hipMalloc(&d_cnt, 8 * 32 * 4)); CK(hipMemset(d_cnt, 0, 8 * 32 * 4));
            CK(hipMalloc(&d_seen, 8 * 4)); CK(hipMemset(d_seen, 0, 8 * 4));
            CK(hipEventRecord(e0));
            static uint8_t* d_segbuf = nullptr; static uint32_t* d_seginfo = nullptr;
            if (argc > 7 && !d_segbuf) { CK(hipMalloc(&d_segbuf, blocks * (n + M) * 4 + 4096)); CK(hipMalloc(&d_seginfo, blocks * ((n + M - 1) / M) * 8 + 64)); }
            walk_tickets<<<256 * per_cu, W>>>(d_P, n, M, (uint32_t)blocks, d_cnt, d_sink, d_seen, d_segbuf, d_seginfo);
            gathers = (uint64_t)n * blocks;
            uint32_t seen[8];
            CK(hipMemcpy(seen, d_seen, 32, hipMemcpyDeviceToHost));
            if (rep == 0) printf("workgroups per XCD: %u %u %u %u %u %u %u %u\n", seen[0], seen[1], seen[2], seen[3], seen[4], seen[5], seen[6], seen[7]);
            CK(hipFree(d_cnt)); CK(hipFree(d_seen));
        } else if (argc > 5) {   // mode 2: XCD-affine, argv[5] = passes over the block (steps = passes * n / walkers)
            int passes = atoi(argv[5]);
            uint32_t walkers = 32 * W;
            uint32_t steps = (uint32_t)((uint64_t)passes * n / walkers);
            CK(hipFuncSetAttribute((const void*)walk_xcd_per_block, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
            uint32_t groups = (uint32_t)((blocks + 7) / 8) * 32 * 8;
            walk_xcd_per_block<<<groups, W, lds_kb * 1024>>>(d_P, n, steps, (uint32_t)blocks, d_sink);
            gathers = (uint64_t)steps * walkers * blocks;
        } else if (W == 1) {
            uint32_t steps = (uint32_t)n / 8;   // an eighth of the walk is enough to time it
            walk_lane_per_block<<<(unsigned)((blocks + 63) / 64), 64>>>(d_P, n, (uint32_t)blocks, steps, d_sink);
            gathers = (uint64_t)steps * blocks;
        } else {
            uint32_t steps = (uint32_t)(n / W);
            if (lds_kb) CK(hipFuncSetAttribute((const void*)walk_group_per_block, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024));
            walk_group_per_block<<<(unsigned)blocks, W, lds_kb * 1024>>>(d_P, n, steps, d_sink);
            gathers = (uint64_t)steps * W * blocks;
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("blocks=%zu n=%zu W=%d lds=%dKB: %.1f ms, %.2f G gathers/s\n", blocks, n, W, lds_kb, ms, gathers / (ms * 1e-3) / 1e9);
    }
    return 0;
}
