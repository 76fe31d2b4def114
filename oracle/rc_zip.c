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
