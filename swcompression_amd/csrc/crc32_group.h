// crc32_group.h -- CRC-32 (IEEE 802.3, reflected) of every job's output, one stream per WORKGROUP.
//
// once decode runs at 100+ GB/s the byte-table checks of the archive layer --
// CheckSums.crc32 (reference Sources/Common/CheckSums.swift:12-28, called from GzipArchive.swift:99, XZArchive.swift:
// 109-120) -- dominate end-to-end time, so the check moves next to the data.
//
// A CRC is a linear map over GF(2):  state(x, A || B) = shift_|B|(state(x, A)) xor state(0, B), where shift_n is the
// 32 x 32 bit matrix "append n zero bytes".  Each of the T threads of a group takes one slice of the output (the first
// thread takes the odd-sized head so that all others have the same length n), runs the table-driven byte loop on it
// (slice-by-4 tables in LDS), and the T partial states are folded in a log2(T)-level tree with the matrices
// shift_n, shift_2n, shift_4n, ... which the group builds by repeated squaring (one matrix column per lane).
#ifndef SWC_CRC32_GROUP_H
#define SWC_CRC32_GROUP_H

#include "swc_common.h"

namespace swc {
namespace crc {

constexpr uint32_t kPoly = 0xEDB88320u;

template <int T>
struct Lds {
    uint32_t tab[4][256];   // slice-by-4
    uint32_t mat[32];       // shift_n for the current tree level: column j = image of bit j
    uint32_t sq[32];
    uint32_t part[T];
};

SWC_HD uint32_t tab0(uint32_t i) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? kPoly : 0u);
    return c;
}
// y = M x over GF(2), M given by its 32 columns
SWC_HD uint32_t mat_vec(const uint32_t* m, uint32_t x) {
    uint32_t y = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) y ^= (x >> j) & 1u ? m[j] : 0u;
    return y;
}

SWC_D uint32_t step4(const uint32_t (*tab)[256], uint32_t c, uint32_t data) {
    const uint32_t w = c ^ data;
    return tab[3][w & 0xFF] ^ tab[2][(w >> 8) & 0xFF] ^ tab[1][(w >> 16) & 0xFF] ^ tab[0][w >> 24];
}
struct q128 { uint32_t x, y, z, w; };
// state after the bytes p[0..n) starting from state c (no pre/post inversion).  The slices of the threads of a wave are
// n bytes apart, so every load of a wave touches 64 different lines; the main loop therefore pulls a whole 128-byte line
// per thread with eight 16-byte loads issued together (the line is fetched once) before it runs the table steps on it.
SWC_D uint32_t run_bytes(const uint32_t (*tab)[256], gcptr p, uint64_t n, uint32_t c) {
    uint64_t i = 0;
    while (i < n && ((uintptr_t)(p + i) & 15)) { c = tab[0][(c ^ p[i]) & 0xFF] ^ (c >> 8); i++; }
    for (; i + 128 <= n; i += 128) {
        q128 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = *(const SWC_AS_GLOBAL q128*)(p + i + 16 * k);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            c = step4(tab, c, v[k].x);
            c = step4(tab, c, v[k].y);
            c = step4(tab, c, v[k].z);
            c = step4(tab, c, v[k].w);
        }
    }
    for (; i + 4 <= n; i += 4) c = step4(tab, c, *(const SWC_AS_GLOBAL uint32_t*)(p + i));
    for (; i < n; i++) c = tab[0][(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c;
}

// CRC-32 of out[0..len), all T threads of the group; the result is returned to every thread.
// Host emulation (T == 1): the same code without barriers.
template <int T>
SWC_D uint32_t crc32_group(gcptr out, uint64_t len, Lds<T>* l, int tid) {
    auto sync = [] {
#if defined(__HIP_DEVICE_COMPILE__)
        __syncthreads();
#endif
    };
    for (int i = tid; i < 256; i += T) l->tab[0][i] = tab0((uint32_t)i);
    sync();
    for (int i = tid; i < 256; i += T) {
        uint32_t c = l->tab[0][i];
        for (int t = 1; t < 4; t++) { c = l->tab[0][c & 0xFF] ^ (c >> 8); l->tab[t][i] = c; }
    }
    sync();
    // slices: thread 0 takes the head of len - (T - 1) * n bytes, threads 1.. take n bytes each
    const uint64_t n = T > 1 ? (len / T) & ~(uint64_t)15 : 0;   // multiple of 16: every slice starts as aligned as the head ends
    const uint64_t head = len - n * (uint64_t)(T - 1);
    uint32_t x;
    if (tid == 0) x = run_bytes(l->tab, out, head, 0xFFFFFFFFu);
    else x = run_bytes(l->tab, out + head + n * (uint64_t)(tid - 1), n, 0u);
    if (T == 1) return ~x;
    l->part[tid] = x;
    // shift_n by square-and-multiply over the bits of n, starting from shift_1 (one zero byte): column j of shift_1 is
    // the state after one zero byte from state 1 << j
    if (tid < 32) {
        uint32_t c = 1u << tid;
        c = l->tab[0][c & 0xFF] ^ (c >> 8);
        l->sq[tid] = c;                       // shift_1
        l->mat[tid] = 1u << tid;              // identity
    }
    sync();
    for (uint64_t k = n; k != 0; k >>= 1) {
        uint32_t nm = 0, ns = 0;
        if (tid < 32) {
            if (k & 1) nm = mat_vec(l->sq, l->mat[tid]);   // mat = sq * mat
            ns = mat_vec(l->sq, l->sq[tid]);                // sq = sq * sq
        }
        sync();
        if (tid < 32) {
            if (k & 1) l->mat[tid] = nm;
            l->sq[tid] = ns;
        }
        sync();
    }
    // tree fold: at level s (stride), part[i] (i multiple of 2s) absorbs part[i + s]; mat = shift_(s * n)
    for (int s = 1; s < T; s <<= 1) {
        uint32_t v = 0;
        const bool act = (tid % (2 * s)) == 0 && tid + s < T;
        if (act) v = mat_vec(l->mat, l->part[tid]) ^ l->part[tid + s];
        uint32_t nm = 0;
        if (tid < 32) nm = mat_vec(l->mat, l->mat[tid]);   // next level: shift_(2 s n) = mat * mat
        sync();
        if (act) l->part[tid] = v;
        if (tid < 32) l->mat[tid] = nm;
        sync();
    }
    return ~l->part[0];
}

}  // namespace crc
}  // namespace swc
#endif
