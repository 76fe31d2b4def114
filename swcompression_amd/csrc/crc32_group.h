// crc32_group.h -- CRC-32 (IEEE 802.3, reflected), CRC-64/XZ and bzip2's MSB-first CRC-32 of every job's output,
// one stream per WORKGROUP.
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

// Reflected polynomials: CRC-32 (CheckSums.swift:59-93 table), CRC-64/XZ (CheckSums.swift:129-195 table).
template <typename W> struct Poly;
template <> struct Poly<uint32_t> { static constexpr uint32_t value = 0xEDB88320u; };
template <> struct Poly<uint64_t> { static constexpr uint64_t value = 0xC96C5795D7870F42ull; };

template <int T, typename W = uint32_t>
struct Lds {
    static constexpr int kBits = (int)sizeof(W) * 8;
    W tab[4][256];    // slice-by-4
    W mat[kBits];     // shift_n for the current tree level: column j = image of bit j
    W sq[kBits];
    W part[T];
};

template <typename W>
SWC_HD W tab0(uint32_t i) {
    W c = i;
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? Poly<W>::value : (W)0);
    return c;
}
// y = M x over GF(2), M given by its columns
template <typename W>
SWC_HD W mat_vec(const W* m, W x) {
    W y = 0;
#pragma unroll
    for (int j = 0; j < (int)sizeof(W) * 8; j++) y ^= (x >> j) & 1u ? m[j] : (W)0;
    return y;
}

// The MSB-first CRC-32 of bzip2 (CheckSums.bzip2crc32, CheckSums.swift:30-37; polynomial 0x04C11DB7) is the reflected
// CRC-32 of the bit-reversed bytes, bit-reversed: with R' = brev32(R) the update R = (R << 8) ^ tab[(R >> 24) ^ b]
// becomes R' = (R' >> 8) ^ tab'[(R' & 0xFF) ^ brev8(b)], and both the initial value and the final inversion are
// symmetric.  MSB = true therefore only reverses the bits of every byte on load and of the result.
template <bool MSB> SWC_HD uint32_t fix_byte(uint32_t b) { return MSB ? brev32(b) >> 24 : b; }
template <bool MSB> SWC_HD uint32_t fix_word(uint32_t w) {
    if (!MSB) return w;
    w = brev32(w);
    return (w >> 24) | ((w >> 8) & 0xFF00u) | ((w << 8) & 0xFF0000u) | (w << 24);
}

template <typename W>
SWC_D W step4(const W (*tab)[256], W c, uint32_t data) {
    const uint32_t w = (uint32_t)c ^ data;
    W r = tab[3][w & 0xFF] ^ tab[2][(w >> 8) & 0xFF] ^ tab[1][(w >> 16) & 0xFF] ^ tab[0][w >> 24];
    if (sizeof(W) > 4) r ^= (W)((uint64_t)c >> 32);
    return r;
}
struct q128 { uint32_t x, y, z, w; };
// state after the bytes p[0..n) starting from state c (no pre/post inversion).  The slices of the threads of a wave are
// n bytes apart, so every load of a wave touches 64 different lines; the main loop therefore pulls a whole 128-byte line
// per thread with eight 16-byte loads issued together (the line is fetched once) before it runs the table steps on it.
template <typename W, bool MSB>
SWC_D W run_bytes(const W (*tab)[256], gcptr p, uint64_t n, W c) {
    uint64_t i = 0;
    while (i < n && ((uintptr_t)(p + i) & 15)) { c = tab[0][((uint32_t)c ^ fix_byte<MSB>(p[i])) & 0xFF] ^ (c >> 8); i++; }
    for (; i + 128 <= n; i += 128) {
        q128 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = *(const SWC_AS_GLOBAL q128*)(p + i + 16 * k);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            c = step4(tab, c, fix_word<MSB>(v[k].x));
            c = step4(tab, c, fix_word<MSB>(v[k].y));
            c = step4(tab, c, fix_word<MSB>(v[k].z));
            c = step4(tab, c, fix_word<MSB>(v[k].w));
        }
    }
    for (; i + 4 <= n; i += 4) c = step4(tab, c, fix_word<MSB>(*(const SWC_AS_GLOBAL uint32_t*)(p + i)));
    for (; i < n; i++) c = tab[0][((uint32_t)c ^ fix_byte<MSB>(p[i])) & 0xFF] ^ (c >> 8);
    return c;
}

// CRC of out[0..len), all T threads of the group; the result is returned to every thread.
// Host emulation: T == 1 without barriers, or T host threads with the barrier of swc_common.h.
template <int T, typename W = uint32_t, bool MSB = false>
SWC_D W crc_group(gcptr out, uint64_t len, Lds<T, W>* l, int tid) {
    constexpr int kBits = (int)sizeof(W) * 8;
    static_assert(T == 1 || T >= kBits, "one lane per matrix column");
    auto sync = [] { group_sync(); };
    auto fin = [](W x) -> W {
        x = ~x;
        return MSB ? (W)brev32((uint32_t)x) : x;
    };
    for (int i = tid; i < 256; i += T) l->tab[0][i] = tab0<W>((uint32_t)i);
    sync();
    for (int i = tid; i < 256; i += T) {
        W c = l->tab[0][i];
        for (int t = 1; t < 4; t++) { c = l->tab[0][c & 0xFF] ^ (c >> 8); l->tab[t][i] = c; }
    }
    sync();
    // slices: thread 0 takes the head of len - (T - 1) * n bytes, threads 1.. take n bytes each
    const uint64_t n = T > 1 ? (len / T) & ~(uint64_t)15 : 0;   // multiple of 16: every slice starts as aligned as the head ends
    const uint64_t head = len - n * (uint64_t)(T - 1);
    W x;
    if (tid == 0) x = run_bytes<W, MSB>(l->tab, out, head, ~(W)0);
    else x = run_bytes<W, MSB>(l->tab, out + head + n * (uint64_t)(tid - 1), n, (W)0);
    if (T == 1) return fin(x);
    l->part[tid] = x;
    // shift_n by square-and-multiply over the bits of n, starting from shift_1 (one zero byte): column j of shift_1 is
    // the state after one zero byte from state 1 << j
    if (tid < kBits) {
        W c = (W)1 << tid;
        c = l->tab[0][c & 0xFF] ^ (c >> 8);
        l->sq[tid] = c;                       // shift_1
        l->mat[tid] = (W)1 << tid;            // identity
    }
    sync();
    for (uint64_t k = n; k != 0; k >>= 1) {
        W nm = 0, ns = 0;
        if (tid < kBits) {
            if (k & 1) nm = mat_vec(l->sq, l->mat[tid]);   // mat = sq * mat
            ns = mat_vec(l->sq, l->sq[tid]);                // sq = sq * sq
        }
        sync();
        if (tid < kBits) {
            if (k & 1) l->mat[tid] = nm;
            l->sq[tid] = ns;
        }
        sync();
    }
    // tree fold: at level s (stride), part[i] (i multiple of 2s) absorbs part[i + s]; mat = shift_(s * n)
    for (int s = 1; s < T; s <<= 1) {
        W v = 0;
        const bool act = (tid % (2 * s)) == 0 && tid + s < T;
        if (act) v = mat_vec(l->mat, l->part[tid]) ^ l->part[tid + s];
        W nm = 0;
        if (tid < kBits) nm = mat_vec(l->mat, l->mat[tid]);   // next level: shift_(2 s n) = mat * mat
        sync();
        if (act) l->part[tid] = v;
        if (tid < kBits) l->mat[tid] = nm;
        sync();
    }
    return fin(l->part[0]);
}

template <int T>
SWC_D uint32_t crc32_group(gcptr out, uint64_t len, Lds<T, uint32_t>* l, int tid) {
    return crc_group<T, uint32_t, false>(out, len, l, tid);
}

}  // namespace crc
}  // namespace swc
#endif
