/*
 * rc_zip.c -- ORACLE (test infrastructure).  Restates ZipContainer.getEntryData
 *   Sources/ZIP/ZipContainer.swift:61-118
 * on top of the oracle's own Deflate / BZip2 / LZMA restatements.  The central-directory walk that produces the
 * arguments (ZipEntryInfoHelper.swift:22-44) is the caller's side of the boundary and is not restated.
 */
#include <stdlib.h>
#include <string.h>
#include "refcpu.h"

static uint32_t z32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static uint64_t z64(const uint8_t* p) { return (uint64_t)z32(p) | (uint64_t)z32(p + 4) << 32; }

/* Returns the status; *crc_error as in the tuple of :61; *out (malloc, refcpu_free) is empty on error. */
int refcpu_zip_get_entry_data(const uint8_t* d, size_t len, uint64_t data_offset, uint64_t comp_size, uint64_t uncomp_size,
                              uint32_t crc32, int method, int has_data_descriptor, int zip64, uint8_t** out, size_t* out_len,
                              int* crc_error) {
    uint8_t* data = NULL;
    size_t n = 0, real = 0;
    int st = SWC_OK;
    *crc_error = 0;
    *out = NULL; *out_len = 0;
    if (data_offset > len) st = SWC_E_REF_TRAP;                   /* byteReader.offset = helper.dataOffset :68 */
    const uint8_t* p = d + (st ? 0 : data_offset);
    const size_t avail = st ? 0 : len - (size_t)data_offset;
    if (st == SWC_OK) switch (method) {
        case 0:                                                   /* .copy :70-71 */
            if (uncomp_size > avail) { st = SWC_E_REF_TRAP; break; }
            data = (uint8_t*)malloc(uncomp_size ? uncomp_size : 1);
            memcpy(data, p, (size_t)uncomp_size);
            n = (size_t)uncomp_size; real = n;
            break;
        case 8:                                                   /* .deflate :72-78 (in_consumed already includes align()) */
            st = refcpu_deflate_decompress(p, avail, &data, &n, &real);
            break;
        case 12:                                                  /* .bzip2 :79-86 */
            st = refcpu_bzip2_decompress(p, avail, &data, &n, &real);
            break;
        case 14: {                                                /* .lzma :87-89 */
            if (avail < 9) { st = SWC_E_REF_TRAP; break; }
            unsigned b = p[4];
            if (b >= 225) { st = SWC_E_LZMA_WRONG_PROPERTIES; break; }   /* LZMAProperties.swift:51 */
            if (uncomp_size > (uint64_t)INT64_MAX) { st = SWC_E_REF_TRAP; break; }
            size_t used = 0;
            st = refcpu_lzma_decompress(p + 9, avail - 9, (int)(b % 9), (int)((b / 9) % 5), (int)((b / 9) / 5), (int64_t)z32(p + 5),
                                        (int64_t)uncomp_size, &data, &n, &used);
            real = 9 + used;
            break;
        }
        default: st = SWC_E_ZIP_COMPRESSION_NOT_SUPPORTED;         /* :90-91 */
    }
    if (st == SWC_OK && has_data_descriptor) {                    /* :96-110 */
        size_t q = (size_t)data_offset + real;
        if (len - q < 4) st = SWC_E_REF_TRAP;
        else {
            if (z32(d + q) == 0x08074b50u) q += 4;
            if (len < q || len - q < (size_t)(4 + (zip64 ? 16 : 8))) st = SWC_E_REF_TRAP;
            else {
                crc32 = z32(d + q);
                if (zip64) { comp_size = z64(d + q + 4); uncomp_size = z64(d + q + 12); }
                else { comp_size = z32(d + q + 4); uncomp_size = z32(d + q + 8); }
            }
        }
    }
    if (st == SWC_OK && !(comp_size == (uint64_t)real && uncomp_size == (uint64_t)n)) st = SWC_E_ZIP_WRONG_SIZE;  /* :112-113 */
    if (st == SWC_OK) *crc_error = crc32 != refcpu_crc32(data, n, 0);   /* :114 */
    if (st != SWC_OK) { refcpu_free(data); data = (uint8_t*)malloc(1); n = 0; }
    *out = data; *out_len = n;
    return st;
}

/*
 * SevenZipFolder.unpack(data:)   Sources/7-Zip/7zFolder.swift:138-194
 * The archive header (coders, bind pairs, unpack sizes) is the caller's; it hands over the ORDERED coder chain.
 * method: 0 copy, 1 deflate, 2 bzip2, 3 LZMA2, 4 LZMA, 5 Delta, 6 LZ4, 7 encryption, 8 other (as swc_7z_coder).
 */

int refcpu_7z_unpack_folder(const uint8_t* data, size_t len, const refcpu_7z_coder* coders, size_t n_coders, uint8_t** out, size_t* out_len) {
    uint8_t* cur = (uint8_t*)malloc(len ? len : 1);
    size_t cur_len = len;
    memcpy(cur, data, len);
    int st = SWC_OK;
    for (size_t k = 0; k < n_coders && st == SWC_OK; k++) {
        const refcpu_7z_coder* c = &coders[k];
        if (c->multi_stream) { st = SWC_E_7Z_MULTI_STREAM_NOT_SUPPORTED; break; }      /* :141-142 */
        uint8_t* next = NULL;
        size_t next_len = 0, used = 0;
        switch (c->method) {
            case 0: continue;                                                          /* .copy :147-148 (no size check) */
            case 1: st = refcpu_deflate_decompress(cur, cur_len, &next, &next_len, &used); break;      /* :149-150 */
            case 2: st = refcpu_bzip2_decompress(cur, cur_len, &next, &next_len, &used); break;        /* :151-152 */
            case 3:                                                                                    /* :153-159 */
                if (c->props_len != 1) { st = SWC_E_LZMA2_WRONG_DICTIONARY_SIZE; break; }
                st = refcpu_lzma2_decompress(cur, cur_len, c->props[0], &next, &next_len, &used);
                break;
            case 4: {                                                                                  /* :160-173 */
                if (c->props_len != 5) { st = SWC_E_LZMA_WRONG_PROPERTIES; break; }
                unsigned b = c->props[0];
                if (b >= 225) { st = SWC_E_LZMA_WRONG_PROPERTIES; break; }
                int64_t dict = (int64_t)((uint32_t)c->props[1] | (uint32_t)c->props[2] << 8 | (uint32_t)c->props[3] << 16 | (uint32_t)c->props[4] << 24);
                st = refcpu_lzma_decompress(cur, cur_len, (int)(b % 9), (int)((b / 9) % 5), (int)((b / 9) / 5), dict, (int64_t)c->unpack_size,
                                            &next, &next_len, &used);
                break;
            }
            case 5:                                                                                    /* :175-181 */
                if (c->props_len != 1) { st = SWC_E_7Z_INTERNAL_STRUCTURE_ERROR; break; }
                next = (uint8_t*)malloc(cur_len ? cur_len : 1);
                next_len = cur_len;
                refcpu_delta_decode(cur, cur_len, (int)(uint8_t)(c->props[0] + 1), next);
                break;
            case 6: st = refcpu_lz4_decompress(cur, cur_len, NULL, 0, -1, &next, &next_len, &used); break;  /* :182-183 */
            case 7: st = SWC_E_7Z_ENCRYPTION_NOT_SUPPORTED; break;                                     /* :185 */
            default: st = SWC_E_7Z_COMPRESSION_NOT_SUPPORTED;                                          /* :187 */
        }
        if (st == SWC_OK && next_len != c->unpack_size) st = SWC_E_7Z_WRONG_SIZE;                       /* :190-191 */
        if (st != SWC_OK) { refcpu_free(next); break; }
        refcpu_free(cur);
        cur = next;
        cur_len = next_len;
    }
    if (st != SWC_OK) { refcpu_free(cur); cur = (uint8_t*)malloc(1); cur_len = 0; }
    *out = cur;
    *out_len = cur_len;
    return st;
}
