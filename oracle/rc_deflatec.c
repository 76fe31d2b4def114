/*
 * rc_deflatec.c -- ORACLE (test infrastructure).  Restates the Deflate compressor of Sources/Deflate/Deflate+Compress.swift:
 *   compress(data:) :22-46          staticHuffmanBitSize :48-83      createUncompressedBlock :85-95
 *   encodeHuffmanBlock :97-139      lengthEncode :146-213
 * The reference writes ONE block -- stored if that is not larger than the static-Huffman block and fits 16 bits of length, else
 * static Huffman -- over a greedy match search that keeps, in a Swift Dictionary [UInt32: Int], the most recent position of
 * every three-byte group it has LOOKED UP (positions inside a match are not entered).  The dictionary is exact; here it is a
 * direct table over the 2^24 groups with a generation stamp.  Line numbers in comments refer to Deflate+Compress.swift;
 * the code tables are those of Deflate+Constants.swift:11-196 as arithmetic (RFC 1951 3.2.5 / 3.2.6).
 * The engine's compressor (csrc/deflate_comp.h) does not reproduce these bytes; this file is the CPU path timed beside
 * it, the yardstick for its compression ratio, and -- through its own decoder and zlib -- a check of the format.
 */
#include <stdlib.h>
#include <string.h>
#include "rc_common.h"

static const int len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const int dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static int length_symbol(int length) {   /* Constants.lengthCode[length - 3] */
    int s = 28;
    while (len_base[s] > length) s--;
    return 257 + s;
}
static int distance_symbol(int distance) {   /* (distanceBase.firstIndex { $0 > distance } ?? 30) - 1 */
    int s = 0;
    while (s < 30 && dist_base[s] <= distance) s++;
    return s - 1;
}
static int length_extra_bits(int sym) { return (sym <= 260 || sym == 285) ? 0 : ((sym - 257) >> 2) - 1; }   /* :113-114 */
static int distance_extra_bits(int sym) { return sym <= 1 ? 0 : (sym >> 1) - 1; }                             /* :121-122 */

typedef struct { uint8_t* out; size_t cap, n; uint64_t acc; int nacc; } bw_t;   /* LsbBitWriter */
static void bw_bits(bw_t* w, uint32_t v, int n) {   /* write(number:bitsCount:): n bits of v, least significant first */
    w->acc |= (uint64_t)v << w->nacc;
    w->nacc += n;
    while (w->nacc >= 8) {
        if (w->n < w->cap) w->out[w->n] = (uint8_t)w->acc;
        w->n++;
        w->acc >>= 8;
        w->nacc -= 8;
    }
}
static uint32_t rev(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i); return r; }
/* the static literal / length code of `sym` (RFC 1951 3.2.6), written most significant code bit first */
static void put_litlen(bw_t* w, int sym) {
    if (sym < 144) bw_bits(w, rev(0x30 + sym, 8), 8);
    else if (sym < 256) bw_bits(w, rev(0x190 + sym - 144, 9), 9);
    else if (sym < 280) bw_bits(w, rev(sym - 256, 7), 7);
    else bw_bits(w, rev(0xC0 + sym - 280, 8), 8);
}

/* Deflate.compress(data:).  Returns SWC_OK, or SWC_E_CAPACITY with *out_len = the size needed. */
int refcpu_deflate_compress(const uint8_t* data, size_t n, uint8_t* out, size_t cap, size_t* out_len) {
    /* ---- lengthEncode :146-213 ---- */
    typedef struct { uint16_t len, dist; } code_t;    /* len 0: a byte (in dist) */
    code_t* codes = (code_t*)malloc((n + 1) * sizeof(code_t));
    static int32_t* last = NULL;                      /* most recent position of a three-byte group, + 1 (0: none in this call) */
    static uint32_t* gen = NULL;
    static uint32_t cur_gen = 0;
    if (!last) { last = (int32_t*)calloc((size_t)1 << 24, sizeof(int32_t)); gen = (uint32_t*)calloc((size_t)1 << 24, sizeof(uint32_t)); }
    if (!codes || !last || !gen || n > 0x7FFFFFF0u) { free(codes); return SWC_E_CAPACITY; }
    cur_gen++;
    size_t ncodes = 0;
    int64_t stats[316];
    memset(stats, 0, sizeof stats);
    size_t i = 0;
    while ((int64_t)i < (int64_t)n - 2) {                                    /* :155 */
        const uint8_t byte = data[i];
        const uint32_t id = (uint32_t)data[i] << 16 | (uint32_t)data[i + 1] << 8 | data[i + 2];
        if (gen[id] != cur_gen) {                                           /* :160 no match found */
            gen[id] = cur_gen; last[id] = (int32_t)i;
            codes[ncodes].len = 0; codes[ncodes++].dist = byte;
            stats[byte]++;
            i++;
            continue;
        }
        const size_t match_start = (size_t)last[id];
        last[id] = (int32_t)i;                                               /* :170 */
        size_t match_len = 3, match_index = match_start + 3;
        const size_t distance = i - match_start;
        if (distance > 32768) {                                              /* :178 */
            codes[ncodes].len = 0; codes[ncodes++].dist = byte;
            stats[byte]++;
            i++;
            continue;
        }
        while (i + match_len < n && data[i + match_len] == data[match_index] && match_len < 258) { match_len++; match_index++; }   /* :185 */
        codes[ncodes].len = (uint16_t)match_len; codes[ncodes++].dist = (uint16_t)distance;
        stats[length_symbol((int)match_len)]++;
        stats[286 + distance_symbol((int)distance)]++;
        i += match_len;
    }
    while (i < n) {                                                          /* :198 the last two bytes */
        codes[ncodes].len = 0; codes[ncodes++].dist = data[i];
        stats[data[i]]++;
        i++;
    }
    stats[256]++;                                                            /* :207 */
    /* ---- compress :22-46 ---- */
    const uint64_t uncomp_size = 1 + 2 + 2 + (uint64_t)n;
    uint64_t bits = 3;                                                       /* staticHuffmanBitSize :48-83 */
    for (int s = 0; s < 316; s++) {
        int cs, eb;
        if (s <= 143) { cs = 8; eb = 0; }
        else if (s <= 255) { cs = 9; eb = 0; }
        else if (s <= 279) { cs = 7; eb = (s <= 260) ? 0 : ((s - 257) >> 2) - 1; }
        else if (s <= 285) { cs = 8; eb = s == 285 ? 0 : ((s - 257) >> 2) - 1; }
        else { cs = 5; eb = (s == 286 || s == 287) ? 0 : ((s - 286) >> 1) - 1; }
        bits += (uint64_t)stats[s] * (uint64_t)(cs + eb);
    }
    const uint64_t static_size = bits % 8 == 0 ? bits / 8 : bits / 8 + 1;
    size_t o = 0;
    if (uncomp_size <= static_size && uncomp_size <= 65535) {                /* createUncompressedBlock :85-95 */
        const uint32_t nl = (uint32_t)n ^ 0xFFFFu;
        const uint8_t hdr[5] = {1, (uint8_t)(n & 0xFF), (uint8_t)((n >> 8) & 0xFF), (uint8_t)(nl & 0xFF), (uint8_t)((nl >> 8) & 0xFF)};
        for (int k = 0; k < 5; k++, o++) if (o < cap) out[o] = hdr[k];
        for (size_t k = 0; k < n; k++, o++) if (o < cap) out[o] = data[k];
    } else {                                                                 /* encodeHuffmanBlock :97-139 */
        bw_t w = {out, cap, 0, 0, 0};
        bw_bits(&w, 1, 1);                                                   /* :103 BFINAL */
        bw_bits(&w, 1, 1); bw_bits(&w, 0, 1);                                /* :104 write(bits: [1, 0]): BTYPE = 01 */
        for (size_t k = 0; k < ncodes; k++) {
            if (codes[k].len == 0) { put_litlen(&w, codes[k].dist); continue; }
            const int ls = length_symbol(codes[k].len);
            put_litlen(&w, ls);
            bw_bits(&w, (uint32_t)(codes[k].len - len_base[ls - 257]), length_extra_bits(ls));
            const int ds = distance_symbol(codes[k].dist);
            bw_bits(&w, rev((uint32_t)ds, 5), 5);
            bw_bits(&w, (uint32_t)(codes[k].dist - dist_base[ds]), distance_extra_bits(ds));
        }
        put_litlen(&w, 256);                                                 /* :135 */
        if (w.nacc) bw_bits(&w, 0, 8 - w.nacc);                              /* align() */
        o = w.n;
    }
    free(codes);
    *out_len = o;
    return o > cap ? SWC_E_CAPACITY : SWC_OK;
}
