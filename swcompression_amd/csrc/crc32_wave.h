// crc32_wave.h -- CRC-32 (IEEE 802.3, reflected) of every job's output, one stream per WAVEFRONT: the shape for batches of
// many members of up to a megabyte (the gzip members of BASELINE configs[1] are 64 KiB each), where the group kernel of
// crc32_group.h spends most of its time in the per-stream set-up (tables, the shift matrix of its slice length by repeated
// squaring, eight fold levels with two barriers each) on one wave while the others wait.
//
// CheckSums.crc32 (reference Sources/Common/CheckSums.swift:12-28; checked by GzipArchive.swift:99, ZipContainer's entry CRC,
// XZArchive.swift:109-120).
//
// A CRC is linear over GF(2): state(0, A || B) = shift_|B|(state(0, A)) xor state(0, B), shift_n = "append n zero bytes".
//   * the stream is cut into PIECES of 32 bytes; piece 64 j + t belongs to lane t, so one load instruction of the wave covers
//     2 KB of consecutive memory.  A lane runs the table loop on its piece from state zero (eight slice-by-4 steps) and folds
//     it into its running state: x = G x xor c with G = shift_2048, applied through four 256-entry tables like a CRC step
//     (the two are independent chains: the G step of piece j overlaps the table loop of piece j + 1);
//   * everything that depends on the stream's length is avoided: the stream is padded IN FRONT with zero bytes to a multiple of
//     2 KB (leading zeros do not change a zero state), and the initial value 0xFFFFFFFF is xor-ed into the first four data
//     bytes instead of being shifted by the length (the textbook identity; streams shorter than four bytes take a byte loop);
//   * at the end the 64 lane states are folded in six levels, x_t = shift_(32 * 2^k) x_t xor x_(t + 2^k), lane shuffles, no
//     barrier; the matrices (columns) are constants.
// All constants (tables, G tables, fold matrices: WaveConsts, 9 KB) are built once per device by a one-group kernel and copied
// into LDS by every group.
#ifndef SWC_CRC32_WAVE_H
#define SWC_CRC32_WAVE_H

#include "swc_common.h"
#include "simt.h"

namespace swc {
namespace crcw {

constexpr uint32_t kPoly = 0xEDB88320u;   // CheckSums.swift:59-93 is the table of this polynomial
constexpr int kPiece = 32;                // bytes per lane per step
constexpr int kRow = kPiece * kWave;      // 2048: bytes per wave per step
constexpr int kLevels = 6;

struct WaveConsts {
    uint32_t tab[4][256];        // slice-by-4
    uint32_t tabg[4][256];       // tabg[k][b] = shift_2048 (b << 8 k)
    uint32_t fold[kLevels][32];  // columns of shift_(32 * 2^k)
    uint32_t sq[32], tmp[32];    // scratch of the construction
};

SWC_HD uint32_t mat_vec(const uint32_t* m, uint32_t x) {
    uint32_t y = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) y ^= (x >> j) & 1u ? m[j] : 0u;
    return y;
}

// Called by the T threads of ONE group (host: T = 1).  shift_1 squared eleven times: shift_(2^m), m = 5..10 are the fold
// levels, m = 11 is G.
template <int T>
SWC_D void build_consts(WaveConsts* c, int tid) {
    for (int i = tid; i < 256; i += T) {
        uint32_t v = (uint32_t)i;
        for (int k = 0; k < 8; k++) v = (v >> 1) ^ ((v & 1u) ? kPoly : 0u);
        c->tab[0][i] = v;
    }
    group_sync();
    for (int i = tid; i < 256; i += T) {
        uint32_t v = c->tab[0][i];
        for (int t = 1; t < 4; t++) { v = c->tab[0][v & 0xFF] ^ (v >> 8); c->tab[t][i] = v; }
    }
    for (int j = tid; j < 32; j += T) {
        const uint32_t v = 1u << j;
        c->sq[j] = c->tab[0][v & 0xFF] ^ (v >> 8);   // shift_1
    }
    group_sync();
    for (int m = 1; m <= 11; m++) {
        for (int j = tid; j < 32; j += T) c->tmp[j] = mat_vec(c->sq, c->sq[j]);
        group_sync();
        for (int j = tid; j < 32; j += T) {
            c->sq[j] = c->tmp[j];
            if (m >= 5 && m <= 10) c->fold[m - 5][j] = c->tmp[j];
        }
        group_sync();
    }
    for (int i = tid; i < 1024; i += T) c->tabg[i >> 8][i & 255] = mat_vec(c->sq, (uint32_t)(i & 255) << (8 * (i >> 8)));
    group_sync();
}

SWC_D uint32_t step4(const WaveConsts* c, uint32_t s, uint32_t data) {
    const uint32_t w = s ^ data;
    return c->tab[3][w & 0xFF] ^ c->tab[2][(w >> 8) & 0xFF] ^ c->tab[1][(w >> 16) & 0xFF] ^ c->tab[0][w >> 24];
}
SWC_D uint32_t stepg(const WaveConsts* c, uint32_t x) {
    return c->tabg[0][x & 0xFF] ^ c->tabg[1][(x >> 8) & 0xFF] ^ c->tabg[2][(x >> 16) & 0xFF] ^ c->tabg[3][x >> 24];
}

struct __attribute__((packed, aligned(1), may_alias)) q128u { uint32_t x, y, z, w; };

// word at real offset r of the padded stream (r < 0: the zero padding), the initial value folded into data bytes 0..3
SWC_D uint32_t head_word(gcptr out, int64_t r) {
    uint32_t w = 0;
    if (r >= 0) w = load_u32(out + r);
    else if (r > -4) for (int b = (int)-r; b < 4; b++) w |= (uint32_t)out[r + b] << (8 * b);
    if (r > -4 && r < 4) w ^= r >= 0 ? 0xFFFFFFFFu >> (8 * r) : 0xFFFFFFFFu << (8 * -r);
    return w;
}

// CRC-32 of out[0..len) by the 64 lanes of one wave; `c` in LDS (or plain memory on the host).  The same value in every lane.
SWC_D uint32_t crc32_wave(gcptr out, uint64_t len, const WaveConsts* c) {
    using namespace simt;
    constexpr int N = kWave;
    if (len < 4) {   // the same serial loop in every lane
        uint32_t s = 0xFFFFFFFFu;
        for (uint64_t i = 0; i < len; i++) s = c->tab[0][(s ^ out[i]) & 0xFF] ^ (s >> 8);
        return ~s;
    }
    const uint64_t pad = (uint64_t)(kRow - (len & (kRow - 1))) & (kRow - 1);
    const uint64_t rows = (len + pad) / kRow;
    PT<uint32_t, N> x, y;
    SIMT_BEGIN(t, N)
        const int ln = t & (N - 1);
        // row 0 of a stream that needs padding: begins in the padding, holds (most of) the four bytes that carry the initial
        // value.  A stream that is a whole number of rows starts in the main loop.
        const uint64_t first = pad != 0 ? 1 : 0;
        uint32_t s = 0;
        const int64_t r0 = (int64_t)ln * kPiece - (int64_t)pad;
        if (pad != 0 && r0 + kPiece > 0) {
#pragma unroll
            for (int k = 0; k < kPiece / 4; k++) s = step4(c, s, head_word(out, r0 + 4 * k));
        }
        uint32_t acc = s;
        // whole pieces, two rows per step: the two table chains (eight dependent steps each) are independent of each other and of
        // the two G steps; the loads of the next two rows are issued before the table steps of these.
        const int64_t r1 = r0 + (int64_t)first * kRow;   // >= 0
        gcptr p = out + r1;
        uint64_t j = first;
        auto ld = [](gcptr q) { return *(const SWC_AS_GLOBAL q128u*)q; };
        auto chain = [c](const q128u& u, const q128u& v) {
            uint32_t z = step4(c, 0, u.x);
            z = step4(c, z, u.y);
            z = step4(c, z, u.z);
            z = step4(c, z, u.w);
            z = step4(c, z, v.x);
            z = step4(c, z, v.y);
            z = step4(c, z, v.z);
            return step4(c, z, v.w);
        };
        const uint32_t init = r1 < 4 ? 0xFFFFFFFFu >> (8 * r1) : 0u;   // the initial value on data bytes r1..3 (0..r1-1 were in row 0)
        if (((rows - first) & 1) != 0) {   // an odd row first
            q128u a0 = ld(p), a1 = ld(p + 16);
            a0.x ^= init;
            acc = stepg(c, acc) ^ chain(a0, a1);
            p += kRow;
            j++;
        }
        q128u a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        if (j < rows) {
            a0 = ld(p); a1 = ld(p + 16); a2 = ld(p + kRow); a3 = ld(p + kRow + 16);
            if (j == first) a0.x ^= init;
        }
        for (; j < rows; j += 2) {
            const q128u v0 = a0, v1 = a1, v2 = a2, v3 = a3;
            p += 2 * kRow;
            if (j + 2 < rows) { a0 = ld(p); a1 = ld(p + 16); a2 = ld(p + kRow); a3 = ld(p + kRow + 16); }
            const uint32_t sa = chain(v0, v1), sb = chain(v2, v3);
            acc = stepg(c, stepg(c, acc) ^ sa) ^ sb;
        }
        x[t] = acc;
    SIMT_END
    for (int k = 0; k < kLevels; k++) {
        wave_shift_down(y, x, 1 << k);
        SIMT_BEGIN(t, N)
            x[t] = mat_vec(c->fold[k], x[t]) ^ y[t];   // (only the lanes that are multiples of 2^(k+1) are used further on)
        SIMT_END
    }
    return ~wave_read(x, 0);
}

}  // namespace crcw
}  // namespace swc
#endif
