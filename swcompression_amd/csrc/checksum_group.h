// checksum_group.h -- Adler-32 and XXH32 of every job's output on the device.
//
//   adler32_group  CheckSums.adler32 (reference Sources/Common/CheckSums.swift:48-57; ZlibArchive.swift:38 compares it
//                  with the stream trailer).  One stream per WORKGROUP.  With s1 = 1 + sum b_i and
//                  s2 = n + sum (n - i) b_i (mod 65521) a slice [lo, hi) contributes  A = sum b_i  to s1 and
//                  B + (n - hi) * A  to s2, where B = sum (hi - i) b_i is the slice-local weighted sum -- so every
//                  thread reduces its own slice and the group adds the contributions (no ordering between slices).
//   xxh32_quad     XxHash32.hash (reference Sources/LZ4/XxHash32.swift:24-83; LZ4.swift:300,326 compare it with the
//                  frame's content / block checksums).  The four accumulators of XXH32 are independent serial chains
//                  (rotate + multiply: not linear, a stream cannot be cut into slices), so one stream takes FOUR lanes
//                  -- lane j of a quad owns accumulator j and reads dword j of every 16-byte stripe -- and a wave holds
//                  16 streams.  Lane 0 of the quad merges the accumulators and runs the (< 16 byte) tail.
#ifndef SWC_CHECKSUM_GROUP_H
#define SWC_CHECKSUM_GROUP_H

#include "swc_common.h"

namespace swc {
namespace sums {

constexpr uint32_t kAdlerBase = 65521u;

struct q128 { uint32_t x, y, z, w; };

// (A, B) of p[0..n): A = sum b_i, B = sum (n - i) b_i, both mod 65521.
SWC_D void adler_slice(gcptr p, uint64_t n, uint32_t& A, uint32_t& B) {
    uint32_t a = 0, b = 0;   // running s1 (without the leading 1) and s2 of the slice
    uint64_t i = 0;
    auto byte = [&](uint32_t v) { a += v; b += a; };
    auto word = [&](uint32_t w) { byte(w & 0xFF); byte((w >> 8) & 0xFF); byte((w >> 16) & 0xFF); byte(w >> 24); };
    // a < 65521 + 255 k, b grows by at most a per byte: 2048 bytes between reductions keep both below 2^32
    while (i < n && ((uintptr_t)(p + i) & 15)) { byte(p[i]); i++; }
    a %= kAdlerBase; b %= kAdlerBase;
    while (i + 128 <= n) {
        const uint64_t stop = i + 2048 < n ? i + 2048 : n;
        for (; i + 128 <= stop; i += 128) {
            q128 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = *(const SWC_AS_GLOBAL q128*)(p + i + 16 * k);
#pragma unroll
            for (int k = 0; k < 8; k++) { word(v[k].x); word(v[k].y); word(v[k].z); word(v[k].w); }
        }
        a %= kAdlerBase; b %= kAdlerBase;
    }
    for (; i < n; i++) byte(p[i]);   // < 128 bytes
    A = a % kAdlerBase;
    B = b % kAdlerBase;
}

template <int T>
struct AdlerLds {
    uint32_t a[T];
    uint32_t b[T];
};

// (s2 << 16) + s1 of out[0..len); returned to every thread.  T == 1: host emulation.
template <int T>
SWC_D uint32_t adler32_group(gcptr out, uint64_t len, AdlerLds<T>* l, int tid) {
    const uint64_t n = T > 1 ? (len / T) & ~(uint64_t)15 : 0;
    const uint64_t head = len - n * (uint64_t)(T - 1);
    const uint64_t lo = tid == 0 ? 0 : head + n * (uint64_t)(tid - 1);
    const uint64_t cnt = tid == 0 ? head : n;
    uint32_t A, B;
    adler_slice(out + lo, cnt, A, B);
    const uint64_t after = (len - (lo + cnt)) % kAdlerBase;   // bytes behind the slice: each adds A once more to s2
    uint32_t s1 = A;
    uint32_t s2 = (uint32_t)((B + after * A) % kAdlerBase);
    if (T > 1) {
        l->a[tid] = s1;
        l->b[tid] = s2;
        group_sync();
        for (int s = T / 2; s > 0; s >>= 1) {
            if (tid < s) {
                l->a[tid] = (l->a[tid] + l->a[tid + s]) % kAdlerBase;
                l->b[tid] = (l->b[tid] + l->b[tid + s]) % kAdlerBase;
            }
            group_sync();
        }
        s1 = l->a[0];
        s2 = l->b[0];
    }
    s1 = (s1 + 1u) % kAdlerBase;                                   // the initial s1 = 1 ...
    s2 = (uint32_t)((s2 + len % kAdlerBase) % kAdlerBase);         // ... is added to s2 once per byte
    return (s2 << 16) + s1;
}

constexpr uint32_t kP1 = 0x9E3779B1u, kP2 = 0x85EBCA77u, kP3 = 0xC2B2AE3Du, kP4 = 0x27D4EB2Fu, kP5 = 0x165667B1u;
SWC_HD uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

SWC_HD uint32_t xxh32_finish(gcptr p, uint64_t len, uint64_t i, uint32_t acc) {   // XxHash32.swift:58-81
    acc += (uint32_t)len;
    for (; len - i >= 4; i += 4) acc = rotl(acc + load_u32(p + i) * kP3, 17) * kP4;
    for (; i < len; i++) acc = rotl(acc + (uint32_t)p[i] * kP5, 11) * kP1;
    acc ^= acc >> 15; acc *= kP2;
    acc ^= acc >> 13; acc *= kP3;
    acc ^= acc >> 16;
    return acc;
}

// One accumulator chain: dword `j` of every stripe of p[0 .. 16 * stripes).
SWC_D uint32_t xxh32_lane(gcptr p, uint64_t stripes, int j, uint32_t acc) {
    gcptr q = p + 4 * j;
    uint64_t s = 0;
    for (; s + 8 <= stripes; s += 8) {   // eight stripes = one 128-byte line across the quad
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = load_u32(q + (s + k) * 16);
#pragma unroll
        for (int k = 0; k < 8; k++) acc = rotl(acc + v[k] * kP2, 13) * kP1;
    }
    for (; s < stripes; s++) acc = rotl(acc + load_u32(q + s * 16) * kP2, 13) * kP1;
    return acc;
}

// XXH32 of out[0..len) by the quad of lanes (lane & 3 == j); valid in lane j == 0.  `quad_get(v, k)` returns lane k of
// the quad's value of v and is called by all lanes together (outside divergent control flow).  Host emulation calls
// the four chains in turn.
template <typename QuadGet>
SWC_D uint32_t xxh32_quad(gcptr out, uint64_t len, uint32_t seed, int j, QuadGet quad_get) {
    const uint32_t init = j == 0 ? seed + kP1 + kP2 : j == 1 ? seed + kP2 : j == 2 ? seed : seed - kP1;
    const uint64_t stripes = len / 16;
    const uint32_t acc = xxh32_lane(out, stripes, j, init);
    uint32_t h = rotl(quad_get(acc, 0), 1) + rotl(quad_get(acc, 1), 7) + rotl(quad_get(acc, 2), 12) + rotl(quad_get(acc, 3), 18);
    if (len < 16) h = seed + kP5;   // hashSmall, XxHash32.swift:33-36
    if (j != 0) return 0;
    return xxh32_finish(out, len, stripes * 16, h);
}

}  // namespace sums
}  // namespace swc
#endif
