/*
 * rc_checksums.c -- ORACLE (test infrastructure).  Restates
 *   CheckSums.crc32      Sources/Common/CheckSums.swift:12-28  (reflected, poly 0xEDB88320)
 *   CheckSums.bzip2crc32 Sources/Common/CheckSums.swift:30-37  (MSB-first, poly 0x04C11DB7)
 *   CheckSums.crc64      Sources/Common/CheckSums.swift:39-46  (CRC-64/XZ, reflected poly 0xC96C5795D7870F42)
 *   CheckSums.adler32    Sources/Common/CheckSums.swift:48-57
 *   XxHash32.hash        Sources/LZ4/XxHash32.swift:24-83
 *   Sha256.hash          Sources/XZ/Sha256.swift:28-142 (FIPS 180-4)
 * The reference stores the CRC tables as literals (CheckSums.swift:61-181); they are the standard
 * tables of these polynomials, generated here.
 */
#include "rc_common.h"

size_t rc_max_output = (size_t)1 << 30;
void refcpu_set_max_output(size_t bytes) { rc_max_output = bytes; }
void refcpu_free(void* p) { free(p); }

static uint32_t t_crc32[256], t_bz[256];
static uint64_t t_crc64[256];
static int tables_ready = 0;

static void init_tables(void) {
    if (tables_ready) return;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        t_crc32[i] = c;
        uint32_t b = i << 24;
        for (int k = 0; k < 8; k++) b = (b & 0x80000000u) ? (b << 1) ^ 0x04C11DB7u : b << 1;
        t_bz[i] = b;
        uint64_t d = i;
        for (int k = 0; k < 8; k++) d = (d & 1) ? 0xC96C5795D7870F42ull ^ (d >> 1) : d >> 1;
        t_crc64[i] = d;
    }
    tables_ready = 1;
}

uint32_t refcpu_crc32(const uint8_t* p, size_t n, uint32_t prev) {
    init_tables();
    uint32_t crc = ~prev;
    for (size_t i = 0; i < n; i++) crc = t_crc32[(crc & 0xFF) ^ p[i]] ^ (crc >> 8);
    return ~crc;
}

uint32_t refcpu_bzip2crc32(const uint8_t* p, size_t n) {
    init_tables();
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) crc = (crc << 8) ^ t_bz[(crc >> 24) ^ p[i]];
    return ~crc;
}

uint64_t refcpu_crc64(const uint8_t* p, size_t n) {
    init_tables();
    uint64_t crc = ~(uint64_t)0;
    for (size_t i = 0; i < n; i++) crc = t_crc64[(crc & 0xFF) ^ p[i]] ^ (crc >> 8);
    return ~crc;
}

uint32_t refcpu_adler32(const uint8_t* p, size_t n) {
    uint32_t s1 = 1, s2 = 0;
    for (size_t i = 0; i < n; i++) {
        s1 = (s1 + p[i]) % 65521u;
        s2 = (s2 + s1) % 65521u;
    }
    return (s2 << 16) + s1;
}

/* ---- XXH32 (XxHash32.swift:18-83) ---- */
#define P1 0x9E3779B1u
#define P2 0x85EBCA77u
#define P3 0xC2B2AE3Du
#define P4 0x27D4EB2Fu
#define P5 0x165667B1u
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

uint32_t refcpu_xxh32(const uint8_t* p, size_t n, uint32_t seed) {
    const uint8_t* end = p + n;
    uint32_t h;
    if (n >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        while (end - p >= 16) {
            v1 = rotl32(v1 + rd32(p) * P2, 13) * P1;
            v2 = rotl32(v2 + rd32(p + 4) * P2, 13) * P1;
            v3 = rotl32(v3 + rd32(p + 8) * P2, 13) * P1;
            v4 = rotl32(v4 + rd32(p + 12) * P2, 13) * P1;
            p += 16;
        }
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)n;
    while (end - p >= 4) { h = rotl32(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl32(h + (*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

/* ---- SHA-256 ---- */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t rotr32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

static void sha256_block(uint32_t h[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void refcpu_sha256(const uint8_t* p, size_t n, uint8_t digest[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t i = 0;
    for (; i + 64 <= n; i += 64) sha256_block(h, p + i);
    uint8_t tail[128];
    size_t rem = n - i;
    memset(tail, 0, sizeof tail);
    if (rem) memcpy(tail, p + i, rem);
    tail[rem] = 0x80;
    size_t tl = rem + 1 + 8 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)n * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int k = 0; k < 8; k++) { digest[4 * k] = (uint8_t)(h[k] >> 24); digest[4 * k + 1] = (uint8_t)(h[k] >> 16); digest[4 * k + 2] = (uint8_t)(h[k] >> 8); digest[4 * k + 3] = (uint8_t)h[k]; }
}
