// checksums.cpp -- host-side checksums used by the framing layer (gzip / zlib / xz / lz4-frame / bzip2).
// Same functions as the reference's CheckSums.swift:12-57 (CRC-32, bzip2 CRC-32, CRC-64/XZ, Adler-32),
// XxHash32.swift:24-83 and Sha256.swift:28-142, restructured for throughput: slicing-by-8 CRC tables
// instead of one lookup per byte, Adler-32 with deferred modulo (the reference takes two `%` per byte).
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <mutex>
#include "../../include/swc_hip.h"

namespace {

uint32_t g_crc32[8][256];
uint32_t g_bz[256];
uint64_t g_crc64[8][256];
std::once_flag g_once;

void build_tables() {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        uint64_t d = i;
        uint32_t b = i << 24;
        for (int k = 0; k < 8; k++) {
            c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
            d = (d >> 1) ^ (0xC96C5795D7870F42ull & (0ull - (d & 1ull)));
            b = (b << 1) ^ (0x04C11DB7u & (0u - (b >> 31)));
        }
        g_crc32[0][i] = c;
        g_crc64[0][i] = d;
        g_bz[i] = b;
    }
    for (int t = 1; t < 8; t++)
        for (uint32_t i = 0; i < 256; i++) {
            g_crc32[t][i] = (g_crc32[t - 1][i] >> 8) ^ g_crc32[0][g_crc32[t - 1][i] & 0xFF];
            g_crc64[t][i] = (g_crc64[t - 1][i] >> 8) ^ g_crc64[0][g_crc64[t - 1][i] & 0xFF];
        }
}
inline void ensure_tables() { std::call_once(g_once, build_tables); }

inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t rol(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline uint32_t ror(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

}  // namespace

extern "C" {

uint32_t swc_crc32(const uint8_t* p, size_t n, uint32_t prev) {
    ensure_tables();
    uint32_t c = ~prev;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c = g_crc32[0][(c ^ *p++) & 0xFF] ^ (c >> 8); n--; }
    while (n >= 8) {
        uint64_t w = rd64(p) ^ c;
        c = g_crc32[7][w & 0xFF] ^ g_crc32[6][(w >> 8) & 0xFF] ^ g_crc32[5][(w >> 16) & 0xFF] ^ g_crc32[4][(w >> 24) & 0xFF] ^
            g_crc32[3][(w >> 32) & 0xFF] ^ g_crc32[2][(w >> 40) & 0xFF] ^ g_crc32[1][(w >> 48) & 0xFF] ^ g_crc32[0][w >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = g_crc32[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}

uint64_t swc_crc64(const uint8_t* p, size_t n) {
    ensure_tables();
    uint64_t c = ~0ull;
    while (n >= 8) {
        uint64_t w = rd64(p) ^ c;
        c = g_crc64[7][w & 0xFF] ^ g_crc64[6][(w >> 8) & 0xFF] ^ g_crc64[5][(w >> 16) & 0xFF] ^ g_crc64[4][(w >> 24) & 0xFF] ^
            g_crc64[3][(w >> 32) & 0xFF] ^ g_crc64[2][(w >> 40) & 0xFF] ^ g_crc64[1][(w >> 48) & 0xFF] ^ g_crc64[0][w >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = g_crc64[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}

uint32_t swc_bzip2_crc32(const uint8_t* p, size_t n) {
    ensure_tables();
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = (c << 8) ^ g_bz[(c >> 24) ^ p[i]];
    return ~c;
}

uint32_t swc_adler32(const uint8_t* p, size_t n) {
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552;  // largest run for which b cannot overflow 32 bits
        n -= k;
        while (k--) { a += *p++; b += a; }
        a %= 65521u;
        b %= 65521u;
    }
    return (b << 16) | a;
}

uint32_t swc_xxh32(const uint8_t* p, size_t n, uint32_t seed) {
    const uint32_t P1 = 0x9E3779B1u, P2 = 0x85EBCA77u, P3 = 0xC2B2AE3Du, P4 = 0x27D4EB2Fu, P5 = 0x165667B1u;
    const uint8_t* e = p + n;
    uint32_t h;
    if (n >= 16) {
        uint32_t a = seed + P1 + P2, b = seed + P2, c = seed, d = seed - P1;
        do {
            a = rol(a + rd32(p) * P2, 13) * P1;
            b = rol(b + rd32(p + 4) * P2, 13) * P1;
            c = rol(c + rd32(p + 8) * P2, 13) * P1;
            d = rol(d + rd32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (e - p >= 16);
        h = rol(a, 1) + rol(b, 7) + rol(c, 12) + rol(d, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)n;
    for (; e - p >= 4; p += 4) h = rol(h + rd32(p) * P3, 17) * P4;
    for (; p < e; p++) h = rol(h + *p * P5, 11) * P1;
    h ^= h >> 15; h *= P2;
    h ^= h >> 13; h *= P3;
    h ^= h >> 16;
    return h;
}

void swc_sha256(const uint8_t* p, size_t n, uint8_t digest[32]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t H[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t block[64];
    const uint64_t bitlen = (uint64_t)n * 8;
    size_t off = 0;
    bool pad_written = false, done = false;
    while (!done) {
        size_t take = n - off < 64 ? n - off : 64;
        memcpy(block, p + off, take);
        off += take;
        if (take < 64) {
            size_t i = take;
            if (!pad_written) { block[i++] = 0x80; pad_written = true; }
            if (i <= 56) {
                memset(block + i, 0, 56 - i);
                for (int k = 0; k < 8; k++) block[56 + k] = (uint8_t)(bitlen >> (56 - 8 * k));
                done = true;
            } else {
                memset(block + i, 0, 64 - i);
            }
        }
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)block[4 * i] << 24 | (uint32_t)block[4 * i + 1] << 16 | (uint32_t)block[4 * i + 2] << 8 | block[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t v[8];
        memcpy(v, H, sizeof v);
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = v[7] + (ror(v[4], 6) ^ ror(v[4], 11) ^ ror(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + K[i] + w[i];
            uint32_t t2 = (ror(v[0], 2) ^ ror(v[0], 13) ^ ror(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
            v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
        }
        for (int i = 0; i < 8; i++) H[i] += v[i];
    }
    for (int i = 0; i < 8; i++) { digest[4 * i] = (uint8_t)(H[i] >> 24); digest[4 * i + 1] = (uint8_t)(H[i] >> 16); digest[4 * i + 2] = (uint8_t)(H[i] >> 8); digest[4 * i + 3] = (uint8_t)H[i]; }
}

}  // extern "C"
