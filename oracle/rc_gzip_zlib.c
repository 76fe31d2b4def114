/*
 * rc_gzip_zlib.c -- ORACLE (test infrastructure).  Restates the decode side of
 *   GzipArchive.unarchive / multiUnarchive / processMember   Sources/GZip/GzipArchive.swift:38-100
 *   GzipHeader.init(_:)                                       Sources/GZip/GzipHeader.swift:68-199
 *   ZlibArchive.unarchive                                     Sources/Zlib/ZlibArchive.swift:25-42
 *   ZlibHeader.init(_:)                                       Sources/Zlib/ZlibHeader.swift:47-92
 */
#include "rc_common.h"

int rc_deflate_stream(rc_bits* r, rc_buf* out);

/* GzipHeader.swift:68-199.  Only validation matters for the decode path (metadata is not returned). */
static int gzip_header(rc_bits* r) {
    if (rc_bytes_left(r) < 10) return SWC_E_GZIP_WRONG_MAGIC; /* :70 */
    size_t hstart = rc_offset(r);
    unsigned magic = (unsigned)rc_le_bytes(r, 2);
    if (magic != 0x8b1f) return SWC_E_GZIP_WRONG_MAGIC; /* :75 */
    unsigned method = rc_byte(r);
    if (method != 8) return SWC_E_GZIP_WRONG_COMPRESSION_METHOD; /* :81 */
    unsigned flags = rc_byte(r);
    if (flags & 0xE0) return SWC_E_GZIP_WRONG_FLAGS; /* :87 */
    rc_le_bytes(r, 4); /* mtime */
    rc_byte(r);        /* xfl */
    rc_byte(r);        /* os */
    if (flags & 0x04) { /* FEXTRA :110-156 */
        if (rc_bytes_left(r) < 2) return SWC_E_GZIP_WRONG_MAGIC;
        int xlen = (int)rc_le_bytes(r, 2);
        if (!(rc_bytes_left(r) >= xlen && xlen >= 4)) return SWC_E_GZIP_WRONG_MAGIC; /* :123 */
        while (xlen > 0) {
            rc_byte(r); /* si1 */
            unsigned si2 = rc_byte(r);
            if (r->trap) return SWC_E_REF_TRAP;
            if (si2 == 0) return SWC_E_GZIP_WRONG_FLAGS; /* :131 */
            int len = (int)rc_le_bytes(r, 2);
            if (r->trap) return SWC_E_REF_TRAP;
            xlen -= 4;
            if (xlen < len) return SWC_E_GZIP_WRONG_MAGIC; /* :145 */
            for (int i = 0; i < len; i++) rc_byte(r);
            if (r->trap) return SWC_E_REF_TRAP;
            xlen -= len;
        }
    }
    if (flags & 0x08) { /* FNAME :158-172 */
        for (;;) {
            if (rc_is_finished(r)) return SWC_E_GZIP_WRONG_MAGIC;
            if (rc_byte(r) == 0) break;
        }
    }
    if (flags & 0x10) { /* FCOMMENT :174-188 */
        for (;;) {
            if (rc_is_finished(r)) return SWC_E_GZIP_WRONG_MAGIC;
            if (rc_byte(r) == 0) break;
        }
    }
    if (flags & 0x02) { /* FHCRC :190-198 */
        if (rc_bytes_left(r) < 2) return SWC_E_GZIP_WRONG_MAGIC;
        size_t hend = rc_offset(r);
        unsigned crc16 = (unsigned)rc_le_bytes(r, 2);
        if ((refcpu_crc32(r->data + hstart, hend - hstart, 0) & 0xFFFF) != crc16) return SWC_E_GZIP_WRONG_HEADER_CRC;
    }
    if (r->trap) return SWC_E_REF_TRAP;
    return SWC_OK;
}

/* GzipArchive.swift:79-100.  Appends the member's data to `out`; *crc_error as Member.crcError. */
static int gzip_member(rc_bits* r, rc_buf* out, int* crc_error) {
    *crc_error = 0;
    if (!(rc_is_aligned(r) && rc_bytes_left(r) >= 20)) return SWC_E_GZIP_WRONG_MAGIC; /* :83 */
    int st = gzip_header(r);
    if (st) return st;
    size_t start = out->len;
    st = rc_deflate_stream(r, out);
    if (st == SWC_OK && r->trap) st = SWC_E_REF_TRAP;
    if (st) { out->len = start; return st; } /* Deflate errors carry no data */
    rc_align(r);
    if (rc_bytes_left(r) < 8) { out->len = start; return SWC_E_GZIP_WRONG_MAGIC; } /* :91 */
    uint32_t crc = (uint32_t)rc_le_bytes(r, 4);
    uint64_t isize = rc_le_bytes(r, 4);
    size_t n = out->len - start;
    if (((uint64_t)n & 0xFFFFFFFFull) != isize) { out->len = start; return SWC_E_GZIP_WRONG_ISIZE; } /* :95 */
    *crc_error = refcpu_crc32(out->p + start, n, 0) != crc;
    return SWC_OK;
}

int refcpu_gzip_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) {
    rc_bits r; rc_buf b; int crc_error;
    rc_bits_init(&r, in, in_len, 0);
    rc_buf_init(&b);
    int st = gzip_member(&r, &b, &crc_error);
    if (st == SWC_OK && crc_error) st = SWC_E_GZIP_WRONG_CRC; /* :44 carries [member] */
    rc_buf_release(&b, out, out_len);
    return st;
}

int refcpu_gzip_multi_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                                size_t** member_sizes, size_t* n_members) {
    rc_bits r; rc_buf b;
    rc_bits_init(&r, in, in_len, 0);
    rc_buf_init(&b);
    size_t cap = 16, n = 0;
    size_t* sizes = (size_t*)malloc(cap * sizeof(size_t));
    int st = SWC_OK;
    while (!rc_is_finished(&r)) { /* :66 */
        int crc_error;
        size_t start = b.len;
        st = gzip_member(&r, &b, &crc_error);
        if (st) break;
        if (n == cap) { cap *= 2; sizes = (size_t*)realloc(sizes, cap * sizeof(size_t)); }
        sizes[n++] = b.len - start;
        if (crc_error) { st = SWC_E_GZIP_WRONG_CRC; break; } /* :71-72 carries members so far incl. this one */
    }
    rc_buf_release(&b, out, out_len);
    *member_sizes = sizes;
    *n_members = n;
    return st;
}

int refcpu_zlib_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) {
    rc_bits r; rc_buf b;
    rc_bits_init(&r, in, in_len, 0);
    rc_buf_init(&b);
    int st = SWC_OK;
    do {
        /* ZlibHeader.swift:47-92 */
        if (rc_bytes_left(&r) < 2) { st = SWC_E_ZLIB_WRONG_COMPRESSION_METHOD; break; } /* :49 */
        unsigned cmf = rc_byte(&r);
        if ((cmf & 0xF) != 8) { st = SWC_E_ZLIB_WRONG_COMPRESSION_METHOD; break; } /* :57 */
        if (((cmf & 0xF0) >> 4) > 7) { st = SWC_E_ZLIB_WRONG_COMPRESSION_INFO; break; } /* :63 */
        unsigned flags = rc_byte(&r);
        /* compression level (flags>>6) is 0...3: every value is a valid enum case (:78) */
        if (((cmf << 8) + flags) % 31 != 0) { st = SWC_E_ZLIB_WRONG_FCHECK; break; } /* :83 */
        if ((flags & 0x20) >> 5) {
            if (rc_bytes_left(&r) < 4) { st = SWC_E_ZLIB_WRONG_FCHECK; break; } /* :88 */
            r.pos += 32;
        }
        st = rc_deflate_stream(&r, &b); /* ZlibArchive.swift:31 */
        if (st == SWC_OK && r.trap) st = SWC_E_REF_TRAP;
        if (st) { b.len = 0; break; }
        rc_align(&r);
        if (rc_bytes_left(&r) < 4) { st = SWC_E_ZLIB_WRONG_ADLER32; break; } /* :34 carries out */
        uint32_t v = (uint32_t)rc_le_bytes(&r, 4);
        uint32_t adler = (v >> 24) | ((v >> 8) & 0xFF00) | ((v << 8) & 0xFF0000) | (v << 24); /* byteSwapped :37 */
        if (refcpu_adler32(b.p, b.len) != adler) { st = SWC_E_ZLIB_WRONG_ADLER32; break; }
    } while (0);
    rc_buf_release(&b, out, out_len);
    return st;
}
