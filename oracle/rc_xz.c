/*
 * rc_xz.c -- ORACLE (test infrastructure).  Restates
 *   XZArchive.unarchive / splitUnarchive / processStream / processIndex / processFooter /
 *   processPadding                      Sources/XZ/XZArchive.swift:27-218
 *   XZBlock.init                        Sources/XZ/XZBlock.swift:18-97
 *   XZStreamHeader.init                 Sources/XZ/XZStreamHeader.swift:33-57
 *   multiByteDecode                     Sources/XZ/LittleEndianByteReader+XZ.swift:10-30
 *   DeltaFilter.decode                  Sources/Common/DeltaFilter.swift:11-33
 */
#include "rc_common.h"

#define TRAPCHK(r) do { if ((r)->trap) return SWC_E_REF_TRAP; } while (0)

/* LittleEndianByteReader+XZ.swift:10-30 */
static int multibyte(rc_bytes* r, int64_t* v) {
    int i = 1;
    int64_t result = rc_b_byte(r);
    TRAPCHK(r);
    if (result <= 127) { *v = result; return SWC_OK; }
    result &= 0x7F;
    for (;;) {
        unsigned b = rc_b_byte(r);
        TRAPCHK(r);
        if (i >= 9 || b == 0) return SWC_E_XZ_MULTI_BYTE_INTEGER_ERROR;
        result += (int64_t)(b & 0x7F) << (7 * i);
        i++;
        if ((b & 0x80) == 0) break;
    }
    *v = result;
    return SWC_OK;
}

/* DeltaFilter.swift:11-33 */
void refcpu_delta_decode(const uint8_t* in, size_t n, int distance, uint8_t* out) {
    uint8_t delta[256];
    memset(delta, 0, sizeof delta);
    int pos = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t tmp = delta[(distance + pos) % 256];
        tmp = (uint8_t)(in[i] + tmp);
        delta[pos] = tmp;
        out[i] = tmp;
        pos = pos == 0 ? 255 : pos - 1;
    }
}

typedef struct xz_filter { int id; int prop; } xz_filter;

/* XZBlock.init :18-97.  Appends the block's data to `out`. */
static int xz_block(unsigned header_size_byte, rc_bytes* r, int check_size, rc_buf* out,
                    int64_t* unpadded_size, int64_t* uncomp_size) {
    int st;
    int64_t header_start = r->off - 1;
    int64_t real_header_size = ((int64_t)header_size_byte + 1) * 4;
    unsigned flags = rc_b_byte(r);
    TRAPCHK(r);
    int filters_count = (int)(flags & 0x03) + 1; /* `&` binds tighter than `+` in Swift */
    if (flags & 0x3C) return SWC_E_XZ_WRONG_FIELD;
    int64_t compressed_size = -1, uncompressed_size = -1;
    if (flags & 0x40) { if ((st = multibyte(r, &compressed_size))) return st; }
    if (flags & 0x80) { if ((st = multibyte(r, &uncompressed_size))) return st; }
    xz_filter filters[4];
    for (int i = 0; i < filters_count; i++) {
        int64_t id;
        if ((st = multibyte(r, &id))) return st;
        if ((uint64_t)id >= 0x4000000000000000ull) return SWC_E_XZ_WRONG_FILTER_ID;
        if (id == 0x21) {
            int64_t ps;
            if ((st = multibyte(r, &ps))) return st;
            if (ps != 1) return SWC_E_LZMA2_WRONG_DICTIONARY_SIZE; /* :47 */
            filters[i].id = 0x21;
            filters[i].prop = rc_b_byte(r);
        } else if (id == 0x03) {
            int64_t ps;
            if ((st = multibyte(r, &ps))) return st;
            if (ps != 1) return SWC_E_XZ_WRONG_FIELD; /* :55 */
            filters[i].id = 0x03;
            filters[i].prop = (int)(uint8_t)(rc_b_byte(r) + 1); /* &+ 1 wraps: 255 -> 0 */
        } else {
            return SWC_E_XZ_WRONG_FILTER_ID;
        }
        TRAPCHK(r);
    }
    while (r->off - header_start < real_header_size - 4) { /* :64-68 */
        unsigned b = rc_b_byte(r);
        TRAPCHK(r);
        if (b != 0) return SWC_E_XZ_WRONG_PADDING;
    }
    uint32_t hcrc = (uint32_t)rc_b_le(r, 4);
    TRAPCHK(r);
    /* :72-75 rewind and CRC the header */
    if (header_start < 0 || header_start + real_header_size - 4 > (int64_t)r->size) return SWC_E_REF_TRAP;
    if (refcpu_crc32(r->data + header_start, (size_t)(real_header_size - 4), 0) != hcrc) return SWC_E_XZ_WRONG_INFO_CRC;
    r->off = header_start + real_header_size - 4 + 4;

    int64_t data_start = r->off;
    /* :78 filters.reversed().reduce(byteReader): the LAST filter reads from the archive reader, each
     * earlier filter reads the previous filter's whole output. */
    uint8_t* cur = NULL; size_t cur_len = 0; int have_cur = 0;
    for (int i = filters_count - 1; i >= 0; i--) {
        rc_bytes tmp_reader;
        rc_bytes* src = r;
        if (have_cur) { rc_bytes_init(&tmp_reader, cur, cur_len); src = &tmp_reader; }
        uint8_t* next = NULL; size_t next_len = 0;
        if (filters[i].id == 0x21) {
            size_t consumed = 0;
            const uint8_t* p = src->data + src->off;
            st = refcpu_lzma2_decompress(p, (size_t)(src->size - src->off), (uint8_t)filters[i].prop, &next, &next_len, &consumed);
            src->off += (int64_t)consumed;
            if (st) { free(next); free(cur); return st; }
        } else {
            /* Delta reads until its reader is finished */
            size_t n = (size_t)(src->size - src->off);
            next = (uint8_t*)malloc(n ? n : 1);
            int distance = filters[i].prop;
            if (distance == 0) { /* (0 + pos) % 256 is fine; distance 0 => tmp = delta[pos] */ }
            refcpu_delta_decode(src->data + src->off, n, distance, next);
            next_len = n;
            src->off = (int64_t)src->size;
        }
        free(cur);
        cur = next; cur_len = next_len; have_cur = 1;
    }
    if (!((compressed_size < 0 || compressed_size == r->off - data_start) &&
          (uncompressed_size < 0 || uncompressed_size == (int64_t)cur_len))) { free(cur); return SWC_E_XZ_WRONG_DATA_SIZE; } /* :80-82 */
    int64_t unpadded = r->off - header_start;
    if (unpadded % 4 != 0) {
        int pad = (int)(4 - unpadded % 4);
        for (int i = 0; i < pad; i++) {
            unsigned b = rc_b_byte(r);
            if (r->trap) { free(cur); return SWC_E_REF_TRAP; }
            if (b != 0) { free(cur); return SWC_E_XZ_WRONG_PADDING; }
        }
    }
    if (!rc_buf_append(out, cur, cur_len)) { free(cur); return SWC_E_CAPACITY; }
    free(cur);
    *unpadded_size = unpadded + check_size;
    *uncomp_size = (int64_t)cur_len;
    return SWC_OK;
}

/* processStream :90-130.  *check_error as the reference's tuple member. */
static int xz_stream(rc_bytes* r, rc_buf* out, int* check_error) {
    static const uint8_t magic[6] = {0xFD, 0x37, 0x7A, 0x58, 0x5A, 0x00};
    *check_error = 0;
    /* XZStreamHeader.init :33-57 (caller guaranteed >= 32 bytes) */
    if (rc_b_left(r) < 12) return SWC_E_REF_TRAP;
    if (memcmp(r->data + r->off, magic, 6) != 0) return SWC_E_XZ_WRONG_MAGIC;
    r->off += 6;
    uint8_t f0 = rc_b_byte(r), f1 = rc_b_byte(r);
    uint32_t fcrc = (uint32_t)rc_b_le(r, 4);
    uint8_t fb[2] = {f0, f1};
    if (refcpu_crc32(fb, 2, 0) != fcrc) return SWC_E_XZ_WRONG_INFO_CRC;
    if (!(f0 == 0 && (f1 & 0xF0) == 0)) return SWC_E_XZ_WRONG_FIELD;
    int check_type = f1 & 0xF, check_size;
    switch (check_type) {
        case 0x00: check_size = 0; break;
        case 0x01: check_size = 4; break;
        case 0x04: check_size = 8; break;
        case 0x0A: check_size = 32; break;
        default: return SWC_E_XZ_WRONG_FIELD;
    }

    size_t cap = 16, nb = 0;
    int64_t* infos = (int64_t*)malloc(cap * 2 * sizeof(int64_t));
    int64_t index_size = -1;
    int st = SWC_OK;
    for (;;) {
        unsigned hs = rc_b_byte(r);
        if (r->trap) { st = SWC_E_REF_TRAP; break; }
        if (hs == 0) {
            /* processIndex :132-167 */
            int64_t index_start = r->off - 1;
            int64_t records;
            if ((st = multibyte(r, &records))) break;
            if (records != (int64_t)nb) { st = SWC_E_XZ_WRONG_FIELD; break; }
            for (size_t i = 0; i < nb && !st; i++) {
                int64_t a, b;
                if ((st = multibyte(r, &a))) break;
                if (a != infos[2 * i]) { st = SWC_E_XZ_WRONG_FIELD; break; }
                if ((st = multibyte(r, &b))) break;
                if (b != infos[2 * i + 1]) { st = SWC_E_XZ_WRONG_DATA_SIZE; break; }
            }
            if (st) break;
            index_size = r->off - index_start;
            if (index_size % 4 != 0) {
                int pad = (int)(4 - index_size % 4);
                for (int i = 0; i < pad; i++) {
                    unsigned b = rc_b_byte(r);
                    if (r->trap) { st = SWC_E_REF_TRAP; break; }
                    if (b != 0) { st = SWC_E_XZ_WRONG_PADDING; break; }
                    index_size++;
                }
                if (st) break;
            }
            uint32_t icrc = (uint32_t)rc_b_le(r, 4);
            if (r->trap) { st = SWC_E_REF_TRAP; break; }
            if (refcpu_crc32(r->data + index_start, (size_t)index_size, 0) != icrc) { st = SWC_E_XZ_WRONG_INFO_CRC; break; }
            index_size += 4;
            break;
        } else {
            int64_t unpadded, uncomp;
            size_t bstart = out->len;
            st = xz_block(hs, r, check_size, out, &unpadded, &uncomp);
            if (st) break;
            const uint8_t* bd = out->p + bstart;
            size_t bl = out->len - bstart;
            if (check_type == 0x01) {
                uint32_t c = (uint32_t)rc_b_le(r, 4);
                if (r->trap) { st = SWC_E_REF_TRAP; break; }
                if (refcpu_crc32(bd, bl, 0) != c) { *check_error = 1; break; }
            } else if (check_type == 0x04) {
                uint64_t c = rc_b_le(r, 8);
                if (r->trap) { st = SWC_E_REF_TRAP; break; }
                if (refcpu_crc64(bd, bl) != c) { *check_error = 1; break; }
            } else if (check_type == 0x0A) {
                if (rc_b_left(r) < 32) { st = SWC_E_REF_TRAP; break; }
                uint8_t dg[32];
                refcpu_sha256(bd, bl, dg);
                int ok = memcmp(dg, r->data + r->off, 32) == 0;
                r->off += 32;
                if (!ok) { *check_error = 1; break; }
            }
            if (nb == cap) { cap *= 2; infos = (int64_t*)realloc(infos, cap * 2 * sizeof(int64_t)); }
            infos[2 * nb] = unpadded; infos[2 * nb + 1] = uncomp; nb++;
        }
    }
    free(infos);
    if (st || *check_error) return st;

    /* processFooter :169-192 */
    uint32_t footer_crc = (uint32_t)rc_b_le(r, 4);
    int64_t backward = ((int64_t)rc_b_le(r, 4) + 1) * 4;
    unsigned fflags = (unsigned)rc_b_le(r, 2);
    TRAPCHK(r);
    if (refcpu_crc32(r->data + r->off - 6, 6, 0) != footer_crc) return SWC_E_XZ_WRONG_INFO_CRC;
    if (backward != index_size) return SWC_E_XZ_WRONG_FIELD;
    if (!((fflags & 0xFF) == 0 && ((fflags & 0xF00) >> 8) == (unsigned)check_type && (fflags & 0xF000) == 0)) return SWC_E_XZ_WRONG_FIELD;
    if (rc_b_left(r) < 2) return SWC_E_REF_TRAP;
    if (!(r->data[r->off] == 0x59 && r->data[r->off + 1] == 0x5A)) return SWC_E_XZ_WRONG_MAGIC;
    r->off += 2;
    return SWC_OK;
}

/* processPadding :194-218 */
static int xz_padding(rc_bytes* r) {
    if (rc_b_finished(r)) return SWC_OK;
    int64_t padding = 0;
    for (;;) {
        unsigned b = rc_b_byte(r);
        TRAPCHK(r);
        if (b != 0) {
            if (padding % 4 != 0) return SWC_E_XZ_WRONG_PADDING;
            break;
        }
        if (rc_b_finished(r)) {
            if (b != 0 || padding % 4 != 3) return SWC_E_XZ_WRONG_PADDING;
            return SWC_OK;
        }
        padding++;
    }
    r->off -= 1;
    return SWC_OK;
}

static int xz_run(const uint8_t* in, size_t in_len, rc_buf* b, size_t** sizes_out, size_t* n_out) {
    rc_bytes r;
    rc_bytes_init(&r, in, in_len);
    size_t cap = 8, n = 0;
    size_t* sizes = (size_t*)malloc(cap * sizeof(size_t));
    int st = SWC_OK;
    while (!rc_b_finished(&r)) {
        if (rc_b_left(&r) < 32) { st = SWC_E_XZ_WRONG_MAGIC; break; } /* :37 */
        int check_error;
        size_t start = b->len;
        st = xz_stream(&r, b, &check_error);
        if (st) { b->len = 0; n = 0; break; } /* thrown errors other than wrongCheck carry nothing */
        if (n == cap) { cap *= 2; sizes = (size_t*)realloc(sizes, cap * sizeof(size_t)); }
        sizes[n++] = b->len - start;
        if (check_error) { st = SWC_E_XZ_WRONG_CHECK; break; } /* :44 carries result so far */
        st = xz_padding(&r);
        if (st) { b->len = 0; n = 0; break; }
    }
    *sizes_out = sizes;
    *n_out = n;
    return st;
}

int refcpu_xz_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) {
    rc_buf b; size_t* sizes; size_t n;
    rc_buf_init(&b);
    int st = xz_run(in, in_len, &b, &sizes, &n);
    free(sizes);
    rc_buf_release(&b, out, out_len);
    return st;
}

int refcpu_xz_split_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                              size_t** stream_sizes, size_t* n_streams) {
    rc_buf b;
    rc_buf_init(&b);
    int st = xz_run(in, in_len, &b, stream_sizes, n_streams);
    rc_buf_release(&b, out, out_len);
    return st;
}
