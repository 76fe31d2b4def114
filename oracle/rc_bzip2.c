/*
 * rc_bzip2.c -- ORACLE (test infrastructure).  Restates
 *   BZip2.decompress(_ bitReader:)      Sources/BZip2/BZip2.swift:50-95
 *   BZip2.decode(_:_:)                  Sources/BZip2/BZip2.swift:97-270
 *   BZip2.multiDecompress               Sources/BZip2/BZip2.swift:40-48
 *   BurrowsWheeler.reverse              Sources/BZip2/BurrowsWheeler.swift:29-64
 *   BlockSize.init?                     Sources/BZip2/BZip2+BlockSize.swift:29-52
 * Line numbers in comments refer to BZip2.swift unless stated otherwise.
 */
#include "rc_common.h"

/* decode(_:_:) :97-270 -- appends the decoded block to `out` */
static int bz_block(rc_bits* r, rc_buf* out) {
    int st = SWC_OK;
    if (rc_bits_left(r) < 41) return SWC_E_BZIP2_WRONG_MAGIC; /* :103 */
    if (rc_bit(r) != 0) return SWC_E_BZIP2_RANDOMIZED_BLOCK;  /* :106 */
    int64_t pointer = (int64_t)rc_int_bits(r, 24);
    unsigned used_map = (unsigned)rc_int_bits(r, 16);
    if (rc_bits_left(r) < 16 * __builtin_popcount(used_map) + 3 + 15) return SWC_E_BZIP2_WRONG_MAGIC; /* :118 */

    uint8_t used[256];
    int n_used = 0;
    for (int blk = 0; blk < 16; blk++) {
        if (used_map & (0x8000u >> blk)) {
            unsigned m = (unsigned)rc_int_bits(r, 16);
            for (int s = 0; s < 16; s++)
                if (m & (0x8000u >> s)) used[n_used++] = (uint8_t)(blk * 16 + s);
        }
    }
    int used_count = 2 + n_used; /* :139 */
    int n_tables = (int)rc_int_bits(r, 3);
    if (n_tables < 2 || n_tables > 6) return SWC_E_BZIP2_WRONG_HUFFMAN_GROUPS; /* :142 */
    int n_selectors = (int)rc_int_bits(r, 15);

    int mtf[6];
    for (int i = 0; i < n_tables; i++) mtf[i] = i;
    int* selectors = (int*)malloc((size_t)(n_selectors ? n_selectors : 1) * sizeof(int));
    int64_t bits_left = rc_bits_left(r);
    for (int i = 0; i < n_selectors; i++) { /* :155-173 */
        int c = 0;
        while (bits_left > 0) {
            int bit = rc_bit(r);
            bits_left--;
            if (bit == 0) break;
            c++;
        }
        if (c >= n_tables) { free(selectors); return SWC_E_BZIP2_WRONG_SELECTOR; }
        int el = mtf[c];
        for (int k = c; k > 0; k--) mtf[k] = mtf[k - 1];
        mtf[0] = el;
        selectors[i] = el;
    }

    rc_tree tables[6];
    int built = 0;
    for (int t = 0; t < n_tables && !st; t++) { /* :177-203 */
        if (bits_left < 5) { st = SWC_E_BZIP2_WRONG_HUFFMAN_CODE_LENGTH; break; }
        int length = (int)rc_int_bits(r, 5);
        bits_left -= 5;
        int lengths[258];
        for (int i = 0; i < used_count; i++) {
            if (!(length >= 0 && length <= 20)) { st = SWC_E_BZIP2_WRONG_HUFFMAN_CODE_LENGTH; break; } /* :185 */
            while (bits_left > 0) {
                int bit = rc_bit(r);
                bits_left--;
                if (bit == 0) break;
                if (!(bits_left > 0)) { st = SWC_E_BZIP2_WRONG_HUFFMAN_CODE_LENGTH; break; } /* :193 */
                length -= rc_bit(r) * 2 - 1;
                bits_left--;
            }
            if (st) break;
            lengths[i] = length; /* the final symbol's length is never range-checked (:185 runs before the deltas) */
        }
        if (st) break;
        st = rc_tree_build(&tables[t], lengths, used_count);
        if (st) break;
        built++;
    }
    if (st) { for (int t = 0; t < built; t++) rc_tree_free(&tables[t]); free(selectors); return st; }

    /* :205-246 symbol loop */
    rc_buf buffer;
    rc_buf_init(&buffer);
    do {
        if (n_selectors == 0) { st = SWC_E_REF_TRAP; break; } /* selectors[0] on an empty array (App. A B3) */
        int decoded = 0;
        const rc_tree* table = &tables[selectors[0]];
        int selector_index = 1;
        int64_t run_length = 0, repeat_power = 1;
        for (;;) {
            if (decoded >= 50) {
                if (!(selector_index < n_selectors)) { st = SWC_E_BZIP2_WRONG_SELECTOR; break; } /* :214 */
                table = &tables[selectors[selector_index]];
                selector_index++;
                decoded = 0;
            }
            int symbol = rc_tree_next(table, r);
            if (symbol == -1) { st = SWC_E_BZIP2_SYMBOL_NOT_FOUND; break; } /* :222 */
            decoded++;
            if (symbol == 0 || symbol == 1) {
                run_length += (int64_t)((uint64_t)repeat_power << symbol); /* &+ and smart shifts wrap */
                repeat_power = (int64_t)((uint64_t)repeat_power << 1);
                continue;
            }
            if (run_length > 0) {
                if (n_used == 0) { st = SWC_E_REF_TRAP; break; } /* usedSymbols[0] on empty array */
                if (!rc_buf_reserve(&buffer, (size_t)run_length)) { st = SWC_E_CAPACITY; break; }
                memset(buffer.p + buffer.len, used[0], (size_t)run_length);
                buffer.len += (size_t)run_length;
                run_length = 0;
                repeat_power = 1;
            }
            if (symbol == used_count - 1) break; /* :239 */
            /* :243-245 inverse MTF on usedSymbols */
            int idx = symbol - 1;
            if (idx >= n_used) { st = SWC_E_REF_TRAP; break; } /* cannot happen: symbol < usedSymbolsCount-1 */
            uint8_t el = used[idx];
            memmove(used + 1, used, (size_t)idx);
            used[0] = el;
            if (!rc_buf_put(&buffer, el)) { st = SWC_E_CAPACITY; break; }
        }
    } while (0);
    for (int t = 0; t < built; t++) rc_tree_free(&tables[t]);
    free(selectors);
    if (st) { free(buffer.p); return st; }

    /* BurrowsWheeler.reverse BurrowsWheeler.swift:29-64 */
    size_t n = buffer.len;
    uint8_t* nt = NULL;
    if (n > 0) {
        int64_t counts[256], base[256];
        memset(counts, 0, sizeof counts);
        for (size_t i = 0; i < n; i++) counts[buffer.p[i]]++;
        int64_t sum = 0;
        for (int c = 0; c < 256; c++) { base[c] = counts[c] ? sum : -1; sum += counts[c]; }
        int64_t* pointers = (int64_t*)malloc(n * sizeof(int64_t));
        for (size_t i = 0; i < n; i++) pointers[base[buffer.p[i]]++] = (int64_t)i;
        nt = (uint8_t*)malloc(n);
        int64_t end = pointer;
        for (size_t i = 0; i < n; i++) {
            if (end < 0 || end >= (int64_t)n) { st = SWC_E_REF_TRAP; break; } /* App. A B4 */
            end = pointers[end];
            nt[i] = buffer.p[end];
        }
        free(pointers);
    }
    free(buffer.p);
    if (st) { free(nt); return st; }

    /* RLE1 undo :251-267: 4 equal bytes AND i < n - 4, then a count byte */
    int64_t i = 0, cnt = (int64_t)n;
    while (i < cnt) {
        if (i < cnt - 4 && nt[i] == nt[i + 1] && nt[i] == nt[i + 2] && nt[i] == nt[i + 3]) {
            size_t run = (size_t)nt[i + 4] + 4;
            if (!rc_buf_reserve(out, run)) { st = SWC_E_CAPACITY; break; }
            memset(out->p + out->len, nt[i], run);
            out->len += run;
            i += 5;
        } else {
            if (!rc_buf_put(out, nt[i])) { st = SWC_E_CAPACITY; break; }
            i += 1;
        }
    }
    free(nt);
    return st;
}

/* decompress(_:) :50-95 */
static int bz_stream(rc_bits* r, rc_buf* out) {
    if (rc_bits_left(r) < 32) return SWC_E_BZIP2_WRONG_MAGIC; /* :53 */
    if (!rc_is_aligned(r)) return SWC_E_REF_TRAP;              /* uint16() precondition */
    unsigned magic = (unsigned)rc_le_bytes(r, 2);
    if (magic != 0x5a42) return SWC_E_BZIP2_WRONG_MAGIC;
    if (rc_byte(r) != 104) return SWC_E_BZIP2_WRONG_VERSION;
    unsigned bs = rc_byte(r);
    if (bs < 0x31 || bs > 0x39) return SWC_E_BZIP2_WRONG_BLOCK_SIZE;

    uint32_t total_crc = 0;
    for (;;) {
        if (rc_bits_left(r) < 80) return SWC_E_BZIP2_WRONG_MAGIC; /* :71 */
        uint64_t block_type = rc_int_bits(r, 48);
        uint32_t block_crc = (uint32_t)rc_int_bits(r, 32);
        if (block_type == 0x314159265359ull) {
            size_t start = out->len;
            int st = bz_block(r, out);
            if (st == SWC_OK && r->trap) st = SWC_E_REF_TRAP;
            if (st) return st;
            if (refcpu_bzip2crc32(out->p + start, out->len - start) != block_crc) return SWC_E_BZIP2_WRONG_CRC; /* :81 */
            total_crc = (total_crc << 1) | (total_crc >> 31);
            total_crc ^= block_crc;
        } else if (block_type == 0x177245385090ull) {
            if (total_crc != block_crc) return SWC_E_BZIP2_WRONG_CRC; /* :86 */
            break;
        } else {
            return SWC_E_BZIP2_WRONG_BLOCK_TYPE;
        }
    }
    return SWC_OK;
}

int refcpu_bzip2_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                            size_t* in_consumed) {
    rc_bits r; rc_buf b;
    rc_bits_init(&r, in, in_len, 1);
    rc_buf_init(&b);
    int st = bz_stream(&r, &b);
    if (st != SWC_OK && st != SWC_E_BZIP2_WRONG_CRC) b.len = 0; /* only wrongCRC carries data (:80-87) */
    rc_align(&r);
    if (in_consumed) *in_consumed = rc_offset(&r) > in_len ? in_len : rc_offset(&r);
    rc_buf_release(&b, out, out_len);
    return st;
}

/* multiDecompress :40-48 */
int refcpu_bzip2_multi_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                                  size_t** stream_sizes, size_t* n_streams) {
    rc_bits r; rc_buf b;
    rc_bits_init(&r, in, in_len, 1);
    rc_buf_init(&b);
    size_t cap = 8, n = 0;
    size_t* sizes = (size_t*)malloc(cap * sizeof(size_t));
    int st = SWC_OK;
    while (!rc_is_finished(&r)) {
        size_t start = b.len;
        st = bz_stream(&r, &b);
        if (st == SWC_E_BZIP2_WRONG_CRC) { /* carries only the failing archive's bytes */
            memmove(b.p, b.p + start, b.len - start);
            b.len -= start;
            n = 0;
            sizes[n++] = b.len;
            break;
        }
        if (st) { b.len = 0; n = 0; break; }
        if (n == cap) { cap *= 2; sizes = (size_t*)realloc(sizes, cap * sizeof(size_t)); }
        sizes[n++] = b.len - start;
        rc_align(&r);
    }
    rc_buf_release(&b, out, out_len);
    *stream_sizes = sizes;
    *n_streams = n;
    return st;
}
