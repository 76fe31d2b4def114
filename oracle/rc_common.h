/*
 * rc_common.h -- ORACLE internals (test infrastructure): growable output buffer and restatements of
 * the three BitByteData readers whose contract the reference's call sites pin.
 * Reading past the end of a reader is a *trap* in the reference (precondition failure), never a
 * thrown error; the readers below latch `trap` so callers return SWC_E_REF_TRAP.
 */
#ifndef RC_COMMON_H
#define RC_COMMON_H

#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include "refcpu.h"

extern size_t rc_max_output;

typedef struct rc_buf {
    uint8_t* p;
    size_t len, cap;
    int overflow; /* rc_max_output exceeded or malloc failed */
} rc_buf;

static inline void rc_buf_init(rc_buf* b) { b->p = NULL; b->len = b->cap = 0; b->overflow = 0; }

static inline int rc_buf_reserve(rc_buf* b, size_t extra) {
    if (b->overflow) return 0;
    size_t need = b->len + extra;
    if (need < b->len || need > rc_max_output) { b->overflow = 1; return 0; }
    if (need <= b->cap) return 1;
    size_t ncap = b->cap ? b->cap : 4096;
    while (ncap < need) ncap *= 2;
    uint8_t* np = (uint8_t*)realloc(b->p, ncap);
    if (!np) { b->overflow = 1; return 0; }
    b->p = np; b->cap = ncap;
    return 1;
}
static inline int rc_buf_put(rc_buf* b, uint8_t v) {
    if (b->len == b->cap && !rc_buf_reserve(b, 1)) return 0;
    b->p[b->len++] = v;
    return 1;
}
static inline int rc_buf_append(rc_buf* b, const uint8_t* src, size_t n) {
    if (!rc_buf_reserve(b, n)) return 0;
    if (n) memcpy(b->p + b->len, src, n);
    b->len += n;
    return 1;
}
/* hand the buffer to the caller (never NULL so that ctypes can always free it) */
static inline void rc_buf_release(rc_buf* b, uint8_t** out, size_t* out_len) {
    if (!b->p) b->p = (uint8_t*)malloc(1);
    *out = b->p; *out_len = b->len;
    b->p = NULL; b->len = b->cap = 0;
}

/* ---------------------------------------------------------------------------------------------
 * Bit reader over [data, data+size), position in bits.  `msb` selects MsbBitReader semantics
 * (bit 7 first, multi-bit integers MSB-first) vs LsbBitReader (bit 0 first, LSB-first integers).
 * Aligned whole-byte reads are little-endian for BOTH readers (BZip2.swift:59-60 pins this).
 * ------------------------------------------------------------------------------------------- */
typedef struct rc_bits {
    const uint8_t* data;
    size_t size;     /* bytes */
    uint64_t pos;    /* bits consumed */
    int msb;
    int trap;
} rc_bits;

static inline void rc_bits_init(rc_bits* r, const uint8_t* d, size_t n, int msb) {
    r->data = d; r->size = n; r->pos = 0; r->msb = msb; r->trap = 0;
}
static inline int64_t rc_bits_left(const rc_bits* r) { return (int64_t)r->size * 8 - (int64_t)r->pos; }
static inline int rc_is_aligned(const rc_bits* r) { return (r->pos & 7) == 0; }
/* bytesLeft counts from the current byte (a partially consumed byte still counts) */
static inline int64_t rc_bytes_left(const rc_bits* r) { return (int64_t)r->size - (int64_t)(r->pos >> 3); }
static inline size_t rc_offset(const rc_bits* r) { return (size_t)(r->pos >> 3); }
static inline int rc_is_finished(const rc_bits* r) { return (r->pos >> 3) >= r->size; }
static inline void rc_align(rc_bits* r) { r->pos = (r->pos + 7) & ~(uint64_t)7; }

static inline int rc_bit(rc_bits* r) {
    if (r->pos >= (uint64_t)r->size * 8) { r->trap = 1; return 0; }
    uint8_t b = r->data[r->pos >> 3];
    int k = (int)(r->pos & 7);
    r->pos++;
    return r->msb ? (b >> (7 - k)) & 1 : (b >> k) & 1;
}
static inline uint64_t rc_int_bits(rc_bits* r, int n) {
    uint64_t v = 0;
    if (n <= 0) return 0;
    if (rc_bits_left(r) < n) { r->trap = 1; r->pos = (uint64_t)r->size * 8; return 0; }
    if (r->msb) { for (int i = 0; i < n; i++) v = (v << 1) | (uint64_t)rc_bit(r); }
    else        { for (int i = 0; i < n; i++) v |= (uint64_t)rc_bit(r) << i; }
    return v;
}
/* aligned whole-byte little-endian reads (precondition in BitByteData: reader is aligned) */
static inline uint64_t rc_le_bytes(rc_bits* r, int n) {
    if (!rc_is_aligned(r) || rc_bytes_left(r) < n) { r->trap = 1; return 0; }
    uint64_t v = 0;
    size_t o = rc_offset(r);
    for (int i = 0; i < n; i++) v |= (uint64_t)r->data[o + i] << (8 * i);
    r->pos += (uint64_t)n * 8;
    return v;
}
static inline uint8_t rc_byte(rc_bits* r) { return (uint8_t)rc_le_bytes(r, 1); }

/* ---------------------------------------------------------------------------------------------
 * LittleEndianByteReader
 * ------------------------------------------------------------------------------------------- */
typedef struct rc_bytes {
    const uint8_t* data;
    size_t size;
    int64_t off; /* may be moved backwards by callers (XZ) */
    int trap;
} rc_bytes;

static inline void rc_bytes_init(rc_bytes* r, const uint8_t* d, size_t n) { r->data = d; r->size = n; r->off = 0; r->trap = 0; }
static inline int64_t rc_b_left(const rc_bytes* r) { return (int64_t)r->size - r->off; }
static inline int rc_b_finished(const rc_bytes* r) { return r->off >= (int64_t)r->size; }
static inline uint8_t rc_b_byte(rc_bytes* r) {
    if (r->off < 0 || r->off >= (int64_t)r->size) { r->trap = 1; return 0; }
    return r->data[r->off++];
}
static inline uint64_t rc_b_le(rc_bytes* r, int n) {
    if (r->off < 0 || rc_b_left(r) < n) { r->trap = 1; r->off = (int64_t)r->size; return 0; }
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint64_t)r->data[r->off + i] << (8 * i);
    r->off += n;
    return v;
}

/* ---- Huffman primitives shared by Deflate and BZip2 (rc_huffman.c) ------------------------- */
typedef struct rc_tree {
    int32_t* nodes;   /* implicit binary heap, -1 = empty; DecodingTree.swift:15-34 */
    int64_t leaf_count;
} rc_tree;
/* lengths[i] is the code length of symbol i (0 = unused).  Code.swift:15-39 + DecodingTree.swift:15-34 */
int rc_tree_build(rc_tree* t, const int* lengths, int n);
void rc_tree_free(rc_tree* t);
/* DecodingTree.swift:36-50 */
int rc_tree_next(const rc_tree* t, rc_bits* r);

#endif
