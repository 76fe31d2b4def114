/*
 * rc_lz4c.c -- ORACLE (test infrastructure).  Restates the block compressor of Sources/LZ4/LZ4+Compress.swift:
 *   compress(block:_:) :157-281        populateMatchStorage :283-292        combine :294-302
 * The reference keeps a Swift Dictionary [UInt32: Int] from the four bytes at a position to the most recent position at
 * which they were seen: an EXACT map, restated here as an open-addressing table keyed by the full 32-bit group (no two
 * different groups share a slot).  Line numbers in comments refer to LZ4+Compress.swift.
 * The engine's compressor (csrc/lz4_comp.h) does not reproduce these bytes; this file is the CPU path timed
 * beside it, and the yardstick for its compression ratio.
 */
#include <stdlib.h>
#include <string.h>
#include "rc_common.h"

typedef struct { uint32_t key; int64_t pos; } slot_t;
typedef struct { slot_t* s; size_t mask; } map_t;
static int map_init(map_t* m, size_t n_keys) {
    size_t cap = 16;
    while (cap < 2 * n_keys + 16) cap <<= 1;
    m->s = (slot_t*)malloc(cap * sizeof(slot_t));
    if (!m->s) return 0;
    for (size_t i = 0; i < cap; i++) m->s[i].pos = -1;
    m->mask = cap - 1;
    return 1;
}
static slot_t* map_find(map_t* m, uint32_t key) {   /* the slot of `key`, or the empty slot where it belongs */
    size_t i = (size_t)(key * 2654435761u) & m->mask;
    while (m->s[i].pos >= 0 && m->s[i].key != key) i = (i + 1) & m->mask;
    return &m->s[i];
}
static uint32_t combine(const uint8_t* b, size_t i) {   /* :294-302 big-endian group of four */
    return (uint32_t)b[i] << 24 | (uint32_t)b[i + 1] << 16 | (uint32_t)b[i + 2] << 8 | b[i + 3];
}
static int put(uint8_t* out, size_t cap, size_t* o, unsigned v) {
    if (*o < cap) out[*o] = (uint8_t)v;
    (*o)++;
    return 1;
}
static void put_len(uint8_t* out, size_t cap, size_t* o, int64_t v) {   /* :220-228 / :240-248: while v >= 0 */
    while (v >= 0) {
        put(out, cap, o, v > 255 ? 255u : (unsigned)v);
        v -= 255;
    }
}

/* compress(block:_:): `bytes` = dict ++ block (the reference appends the block to the dictionary's bytes, :160-163),
 * the block starts at `start`.  Returns SWC_OK (or SWC_E_CAPACITY with *out_len = the size needed). */
int refcpu_lz4_compress_block(const uint8_t* bytes, size_t total, size_t start, uint8_t* out, size_t cap, size_t* out_len) {
    map_t m;
    size_t o = 0;
    if (!map_init(&m, total)) return SWC_E_CAPACITY;
    if (start > 0 && start >= 4) {   /* populateMatchStorage :283-292: i in 0 ..< dict.count - 4 */
        for (size_t i = 0; i + 4 < start; i++) {
            slot_t* s = map_find(&m, combine(bytes, i));
            s->key = combine(bytes, i); s->pos = (int64_t)i;
        }
    }
    size_t i = start, lit0 = start;   /* currentLiterals = bytes[lit0 ..< i] */
    const int64_t end = (int64_t)total;
    while ((int64_t)i < end - 9) {    /* :185 */
        const uint32_t id = combine(bytes, i);
        slot_t* s = map_find(&m, id);
        if (s->pos < 0) {             /* :187-193 no match found */
            s->key = id; s->pos = (int64_t)i;
            i++;
            continue;
        }
        const int64_t ms = s->pos;
        s->pos = (int64_t)i;          /* :195 */
        int64_t len = 4, mi = ms + 4; /* :198-200 */
        const int64_t distance = (int64_t)i - ms;
        if (distance > 65535) { i++; continue; }   /* :203-207 */
        while ((int64_t)i + len < end - 5 && bytes[i + len] == bytes[mi]) { len++; mi++; }   /* :213-216 */
        if (end - (int64_t)i < 12) break;   /* :218-222 */
        const size_t nlit = i - lit0;
        put(out, cap, &o, (unsigned)((nlit < 15 ? nlit : 15) << 4 | (len - 4 < 15 ? len - 4 : 15)));   /* :226-228 */
        put_len(out, cap, &o, (int64_t)nlit - 15);
        for (size_t k = lit0; k < i; k++) put(out, cap, &o, bytes[k]);
        put(out, cap, &o, (unsigned)(distance & 0xFF));
        put(out, cap, &o, (unsigned)((distance >> 8) & 0xFF));
        i += (size_t)len;
        put_len(out, cap, &o, len - 19);
        lit0 = i;
    }
    i = total;                        /* :253-256 the remaining bytes are literals */
    {
        const size_t nlit = i - lit0;
        put(out, cap, &o, (unsigned)((nlit < 15 ? nlit : 15) << 4));
        put_len(out, cap, &o, (int64_t)nlit - 15);
        for (size_t k = lit0; k < i; k++) put(out, cap, &o, bytes[k]);
    }
    free(m.s);
    *out_len = o;
    return o > cap ? SWC_E_CAPACITY : SWC_OK;
}
