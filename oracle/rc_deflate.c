/*
 * rc_deflate.c -- ORACLE (test infrastructure).  Restates Deflate.decompress(_ bitReader:)
 * Sources/Deflate/Deflate.swift:30-249 with the tables of Deflate+Constants.swift:175-186.
 * Line numbers in comments refer to Deflate.swift unless stated otherwise.
 */
#include "rc_common.h"

static const int k_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15}; /* Constants:175 */
static const int k_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35,
                                   43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258}; /* Constants:179 */
static const int k_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
                                    257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                    8193, 12289, 16385, 24577}; /* Constants:183 */

/* body of decompress(_:); reader shared with the caller */
int rc_deflate_stream(rc_bits* r, rc_buf* out) {
    int st = SWC_OK;
    if (rc_bits_left(r) < 10) return SWC_E_DEFLATE_WRONG_BLOCK_TYPE; /* :36 */

    for (;;) {
        int is_last = rc_bit(r);                 /* :41 */
        int block_type = (int)rc_int_bits(r, 2); /* :43 */
        if (r->trap) return SWC_E_REF_TRAP;      /* 2nd+ block header past the end: bit() traps */

        if (block_type == 0) {
            rc_align(r); /* :46 */
            if (rc_bytes_left(r) < 4) return SWC_E_DEFLATE_WRONG_UNCOMPRESSED_BLOCK_LENGTHS; /* :48 */
            unsigned length = (unsigned)rc_le_bytes(r, 2);
            unsigned nlength = (unsigned)rc_le_bytes(r, 2);
            if ((length & nlength) != 0) return SWC_E_DEFLATE_WRONG_UNCOMPRESSED_BLOCK_LENGTHS; /* :56 */
            if (rc_bytes_left(r) < (int64_t)length) return SWC_E_DEFLATE_WRONG_UNCOMPRESSED_BLOCK_LENGTHS; /* :59 */
            if (!rc_buf_append(out, r->data + rc_offset(r), length)) return SWC_E_CAPACITY;
            r->pos += (uint64_t)length * 8;
        } else if (block_type == 1 || block_type == 2) {
            rc_tree lit, dist;
            lit.nodes = dist.nodes = NULL;
            if (block_type == 1) {
                int ll[288], dl[32]; /* Constants:11-173: canonical codes of these lengths */
                for (int i = 0; i < 288; i++) ll[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                for (int i = 0; i < 32; i++) dl[i] = 5;
                if ((st = rc_tree_build(&lit, ll, 288))) return st;
                if ((st = rc_tree_build(&dist, dl, 32))) { rc_tree_free(&lit); return st; }
            } else {
                if (rc_bits_left(r) < 14) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND; /* :86 */
                int literals = (int)rc_int_bits(r, 5) + 257;
                if (literals > 286) return SWC_E_DEFLATE_WRONG_SYMBOL; /* :94 */
                int distances = (int)rc_int_bits(r, 5) + 1;
                int cl_count = (int)rc_int_bits(r, 4) + 4;
                if (rc_bits_left(r) < 3 * cl_count) return SWC_E_DEFLATE_SYMBOL_NOT_FOUND; /* :101 */
                int ordered[19];
                memset(ordered, 0, sizeof ordered);
                for (int i = 0; i < cl_count; i++) ordered[k_cl_order[i]] = (int)rc_int_bits(r, 3);
                rc_tree cl;
                if ((st = rc_tree_build(&cl, ordered, 19))) return st;

                int code_lengths[286 + 32];
                int total = literals + distances;
                memset(code_lengths, 0, sizeof code_lengths);
                int n = 0;
                st = SWC_OK;
                while (n < total) {
                    int symbol = rc_tree_next(&cl, r);
                    if (symbol == -1) { st = SWC_E_DEFLATE_SYMBOL_NOT_FOUND; break; } /* :122 */
                    if (symbol >= 0 && symbol <= 15) {
                        code_lengths[n++] = symbol;
                    } else if (symbol == 16 && n > 0) {
                        if (rc_bits_left(r) < 2) { st = SWC_E_DEFLATE_SYMBOL_NOT_FOUND; break; } /* :132 */
                        int copy = (int)rc_int_bits(r, 2) + 3;
                        if (n + copy > total) { st = SWC_E_DEFLATE_WRONG_SYMBOL; break; } /* :135 */
                        for (int i = 0; i < copy; i++) code_lengths[n + i] = code_lengths[n - 1];
                        n += copy;
                    } else if (symbol == 17) {
                        if (rc_bits_left(r) < 3) { st = SWC_E_DEFLATE_SYMBOL_NOT_FOUND; break; } /* :145 */
                        n += (int)rc_int_bits(r, 3) + 3;
                    } else if (symbol == 18) {
                        if (rc_bits_left(r) < 7) { st = SWC_E_DEFLATE_SYMBOL_NOT_FOUND; break; } /* :152 */
                        n += (int)rc_int_bits(r, 7) + 11;
                    } else {
                        st = SWC_E_DEFLATE_WRONG_SYMBOL; /* :155 (also symbol 16 first) */
                        break;
                    }
                }
                rc_tree_free(&cl);
                if (st) return st;
                if (n != total) return SWC_E_DEFLATE_WRONG_SYMBOL; /* :161 */
                if ((st = rc_tree_build(&lit, code_lengths, literals))) return st;
                if ((st = rc_tree_build(&dist, code_lengths + literals, distances))) { rc_tree_free(&lit); return st; }
            }

            /* main loop :171-236 */
            st = SWC_OK;
            for (;;) {
                int sym = rc_tree_next(&lit, r);
                if (sym == -1) { st = SWC_E_DEFLATE_SYMBOL_NOT_FOUND; break; } /* :175 */
                if (sym <= 255) {
                    if (!rc_buf_put(out, (uint8_t)sym)) { st = SWC_E_CAPACITY; break; }
                } else if (sym == 256) {
                    break;
                } else if (sym <= 285) {
                    int extra_len = (sym <= 260 || sym == 285) ? 0 : (((sym - 257) >> 2) - 1); /* :188 */
                    if (rc_bits_left(r) < extra_len) { st = SWC_E_DEFLATE_SYMBOL_NOT_FOUND; break; } /* :192 */
                    int length = k_len_base[sym - 257] + (int)rc_int_bits(r, extra_len);
                    int dcode = rc_tree_next(&dist, r);
                    if (dcode == -1) { st = SWC_E_DEFLATE_SYMBOL_NOT_FOUND; break; } /* :199 */
                    if (dcode > 29) { st = SWC_E_DEFLATE_WRONG_SYMBOL; break; }     /* :201 */
                    int extra_dist = dcode <= 1 ? 0 : ((dcode >> 1) - 1); /* :206 */
                    if (rc_bits_left(r) < extra_dist) { st = SWC_E_DEFLATE_SYMBOL_NOT_FOUND; break; } /* :208 */
                    int distance = k_dist_base[dcode] + (int)rc_int_bits(r, extra_dist);
                    /* :216-232 -- `length/distance` whole repeats of the last `distance` bytes plus a
                     * remainder == the usual overlapping copy; out[count - distance] with
                     * distance > count is a negative index => Swift trap (App. A6). */
                    if ((size_t)distance > out->len) { st = SWC_E_REF_TRAP; break; }
                    if (!rc_buf_reserve(out, (size_t)length)) { st = SWC_E_CAPACITY; break; }
                    uint8_t* p = out->p + out->len;
                    for (int i = 0; i < length; i++) p[i] = p[i - distance];
                    out->len += (size_t)length;
                } else {
                    st = SWC_E_DEFLATE_WRONG_SYMBOL; /* :233 (286, 287) */
                    break;
                }
            }
            rc_tree_free(&lit);
            rc_tree_free(&dist);
            if (st) return st;
        } else {
            return SWC_E_DEFLATE_WRONG_BLOCK_TYPE; /* :239 */
        }
        if (is_last == 1) break; /* :243 */
    }
    return SWC_OK;
}

int refcpu_deflate_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                              size_t* in_consumed) {
    rc_bits r;
    rc_buf b;
    rc_bits_init(&r, in, in_len, 0);
    rc_buf_init(&b);
    int st = rc_deflate_stream(&r, &b);
    if (st == SWC_OK && r.trap) st = SWC_E_REF_TRAP;
    rc_align(&r); /* every caller aligns after the call (GzipArchive.swift:89, ZlibArchive.swift:32) */
    if (in_consumed) *in_consumed = rc_offset(&r) > in_len ? in_len : rc_offset(&r);
    rc_buf_release(&b, out, out_len);
    return st;
}
