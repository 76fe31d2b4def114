/*
 * rc_bzip2c.c -- ORACLE (test infrastructure).  Restates the BZip2 compressor of Sources/BZip2/BZip2+Compress.swift:
 *   compress(data:blockSize:) :41-74    process :76-241    initialRle :243-262    mtf :265-275    mtfRle :277-325
 * with BurrowsWheeler.transform (BurrowsWheeler.swift:8-29), BZip2.lengths(from:) (BZip2+Lengths.swift:14-117, the JPEG
 * Annex K procedure) and Code.huffmanCodes (Common/CodingTree/Code.swift:15-39).  Line numbers in comments refer to
 * BZip2+Compress.swift unless a file is named.
 *
 * What the reference does inside a block, and what this file therefore does: a Huffman table is built from the FIRST 50 symbols
 * (every symbol's count starting at one), and for each further group of 50 symbols a table built from that group alone is
 * added if it codes the group in fewer bits than the best table so far -- until there are six; a group's selector is the
 * cheapest table at the time the group is reached (:92-139).  Code lengths come out of the Annex K procedure as COUNTS per
 * length and are handed to the symbols in symbol order (BZip2+Lengths.swift:104-116): symbol 0 gets the shortest code.
 *
 * The one liberty: the reference sorts the suffixes of the doubled block (SuffixArray.swift); here the rotations are sorted by
 * prefix doubling.  The column is the same; among equal rotations (periodic blocks) the reference's pointer is the first of
 * them (the suffix that starts at the second copy is a prefix of, and so sorts in front of, every equal rotation), which is
 * what rank[0] is here.
 *
 * The engine's compressor (csrc/bzip2_comp.h) does not reproduce these bytes; this file is the yardstick for its
 * compression ratio and -- through the oracle's own decoder and libbz2 -- a check of the format.
 */
#include <stdlib.h>
#include <string.h>
#include "rc_common.h"

typedef struct { uint8_t* out; size_t cap, n; uint32_t acc; int nacc; } mbw_t;   /* MsbBitWriter */
static void mbw_bit(mbw_t* w, int b) {
    w->acc = (w->acc << 1) | (uint32_t)(b & 1);
    if (++w->nacc == 8) {
        if (w->n < w->cap) w->out[w->n] = (uint8_t)w->acc;
        w->n++;
        w->acc = 0; w->nacc = 0;
    }
}
static void mbw_bits(mbw_t* w, uint64_t v, int n) { for (int i = n - 1; i >= 0; i--) mbw_bit(w, (int)((v >> i) & 1)); }
static void mbw_align(mbw_t* w) { while (w->nacc) mbw_bit(w, 0); }

/* ---- BurrowsWheeler.transform ------------------------------------------------------------------------------------------ */
static __thread const int* s_rank;
static __thread int s_h, s_n;
static int cmp_rot(const void* a, const void* b) {
    const int i = *(const int*)a, j = *(const int*)b;
    if (s_rank[i] != s_rank[j]) return s_rank[i] < s_rank[j] ? -1 : 1;
    const int ri = s_rank[(i + s_h) % s_n], rj = s_rank[(j + s_h) % s_n];
    return ri < rj ? -1 : ri > rj;
}
static int bwt(const int* bytes, int n, int* col) {   /* returns the pointer */
    int* sa = malloc(sizeof(int) * (size_t)n);
    int* rank = malloc(sizeof(int) * (size_t)n);
    int* next = malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) { sa[i] = i; rank[i] = bytes[i]; }
    for (int h = 0;; h = h ? 2 * h : 1) {
        s_rank = rank; s_h = h; s_n = n;
        qsort(sa, (size_t)n, sizeof(int), cmp_rot);
        int distinct = 1;
        next[sa[0]] = 0;
        for (int k = 1; k < n; k++) {
            if (cmp_rot(&sa[k - 1], &sa[k]) != 0) { next[sa[k]] = k; distinct++; }
            else next[sa[k]] = next[sa[k - 1]];
        }
        memcpy(rank, next, sizeof(int) * (size_t)n);
        if (distinct == n || (h ? 2 * h : 1) >= n) break;
    }
    for (int k = 0; k < n; k++) col[k] = bytes[(sa[k] + n - 1) % n];
    const int pointer = rank[0];
    free(sa); free(rank); free(next);
    return pointer;
}

/* ---- BZip2.lengths(from:) BZip2+Lengths.swift:14-117 ---------------------------------------------------------------------- */
static void lengths_from(const int* stats_in, int count, int* len_of_symbol) {
    if (count == 1) { len_of_symbol[0] = 1; return; }
    long* stats = malloc(sizeof(long) * (size_t)count);
    int* cl = calloc((size_t)count, sizeof(int));
    int* others = malloc(sizeof(int) * (size_t)count);
    int* bits = calloc((size_t)count + 1, sizeof(int));
    for (int i = 0; i < count; i++) { stats[i] = stats_in[i]; others[i] = -1; }
    for (;;) {                                         /* calculateCodeLengths :33-78 */
        int c1 = -1, c2 = -1;
        long min = 0x7FFFFFFFFFFFFFFFL;
        for (int i = 0; i < count; i++) if (stats[i] > 0 && stats[i] <= min) { min = stats[i]; c1 = i; }
        min = 0x7FFFFFFFFFFFFFFFL;
        for (int i = 0; i < count; i++) if (stats[i] > 0 && stats[i] <= min && i != c1) { min = stats[i]; c2 = i; }
        if (c2 < 0) break;
        stats[c1] += stats[c2];
        stats[c2] = 0;
        cl[c1]++;
        while (others[c1] >= 0) { c1 = others[c1]; cl[c1]++; }
        others[c1] = c2;
        cl[c2]++;
        while (others[c2] >= 0) { c2 = others[c2]; cl[c2]++; }
    }
    for (int i = 0; i < count; i++) bits[cl[i]]++;     /* count :80-87 (a length never reaches `count`) */
    for (int i = count - 1; i > 20; i--) {             /* adjust :89-102 */
        while (bits[i] > 0) {
            int j = i - 2;
            while (bits[j] == 0) j--;
            bits[i] -= 2;
            bits[i - 1] += 1;
            bits[j + 1] += 2;
            bits[j] -= 1;
        }
    }
    int symbol = 0;                                     /* generateSizeTable :104-116: in SYMBOL order */
    const int top = count - 1 < 20 ? count - 1 : 20;
    for (int i = 1; i <= top; i++) for (int j = 1; j <= bits[i]; j++) if (symbol < count) len_of_symbol[symbol++] = i;
    while (symbol < count) len_of_symbol[symbol++] = top;   /* (not reached: the counts add up to `count`) */
    free(stats); free(cl); free(others); free(bits);
}
/* Code.huffmanCodes Code.swift:15-39 (MSB-first use: the codes as numbers, not reversed) */
static void codes_from(const int* len, int count, uint32_t* code) {
    uint32_t next = 0;
    for (int l = 1; l <= 20; l++) {
        for (int s = 0; s < count; s++) if (len[s] == l) code[s] = next++;
        next <<= 1;
    }
}
static long bit_size(const int* len, const int* stats, int count) {   /* EncodingTree.bitSize(for:) */
    long t = 0;
    for (int s = 0; s < count; s++) t += (long)stats[s] * len[s];
    return t;
}

/* ---- process :76-241 ---------------------------------------------------------------------------------------------------- */
static void process_block(const uint8_t* block, int n, mbw_t* w) {
    /* initialRle :243-262 */
    int* rle = malloc(sizeof(int) * ((size_t)n + (size_t)n / 4 + 8));
    int nr = 0;
    for (int i = 0; i < n;) {
        int run = 1;
        while (i + 1 < n && block[i] == block[i + 1] && run < 255) { run++; i++; }
        for (int k = 0; k < (run < 4 ? run : 4); k++) rle[nr++] = block[i];
        if (run >= 4) rle[nr++] = run - 4;
        i++;
    }
    int* col = malloc(sizeof(int) * (size_t)nr);
    const int pointer = bwt(rle, nr, col);                                               /* :79-80 */
    int used[256], n_used = 0, seen[256] = {0};
    for (int i = 0; i < nr; i++) seen[col[i]] = 1;
    for (int v = 0; v < 256; v++) if (seen[v]) used[n_used++] = v;                       /* :82 */
    /* mtfRle :277-325 */
    int* sym = malloc(sizeof(int) * ((size_t)nr + 2));
    int ns = 0, dict[256], run = 0, max_symbol = 1;
    memcpy(dict, used, sizeof(int) * (size_t)n_used);
    for (int i = 0; i < nr; i++) {
        int idx = 0;
        while (dict[idx] != col[i]) idx++;
        if (idx == 0) run++;
        if ((idx == 0 && i == nr - 1) || idx != 0) {
            while (run > 0) {                     /* digits of the bijective base 2, least significant first (:293-307) */
                sym[ns++] = (run & 1) ? 0 : 1;
                run = (run - 1) >> 1;
            }
        }
        if (idx != 0) {
            sym[ns++] = idx + 1;
            if (idx + 1 > max_symbol) max_symbol = idx + 1;
        }
        const int old = dict[idx];
        memmove(dict + 1, dict, sizeof(int) * (size_t)idx);
        dict[0] = old;
    }
    sym[ns++] = max_symbol + 1;
    const int alpha = max_symbol + 2;
    /* tables and selectors :89-139 */
    int tab_len[6][258], n_tab = 0;
    int* selectors = malloc(sizeof(int) * ((size_t)ns / 50 + 2));
    int n_sel = 0, stats[258], processed = 50;
    for (int s = 0; s < alpha; s++) stats[s] = 1;
    for (int i = 0; i < ns; i++) {
        stats[sym[i]]++;
        processed--;
        if (processed <= 0 || i == ns - 1) {
            processed = 50;
            long best = 0x7FFFFFFFFFFFFFFFL;
            int best_sel = -1;
            for (int t = 0; t < n_tab; t++) {
                const long b = bit_size(tab_len[t], stats, alpha);
                if (b < best) { best = b; best_sel = t; }
            }
            if (n_tab == 6) selectors[n_sel++] = best_sel;
            else {
                int len[258];
                lengths_from(stats, alpha, len);
                if (bit_size(len, stats, alpha) < best) {
                    memcpy(tab_len[n_tab], len, sizeof(int) * (size_t)alpha);
                    selectors[n_sel++] = n_tab++;
                } else selectors[n_sel++] = best_sel;
            }
            for (int s = 0; s < alpha; s++) stats[s] = 1;
        }
    }
    if (n_tab == 1) { memcpy(tab_len[1], tab_len[0], sizeof(int) * (size_t)alpha); n_tab = 2; }   /* :142-147 */
    /* header :151-221 */
    mbw_bit(w, 0);
    mbw_bits(w, (uint64_t)pointer, 24);
    int map16[16] = {0};
    for (int k = 0; k < n_used; k++) map16[used[k] / 16] = 1;
    for (int k = 0; k < 16; k++) mbw_bit(w, map16[k]);
    for (int k = 0; k < 16; k++) if (map16[k]) for (int j = 0; j < 16; j++) mbw_bit(w, seen[16 * k + j]);
    mbw_bits(w, (uint64_t)n_tab, 3);
    mbw_bits(w, (uint64_t)n_sel, 15);
    int order[6] = {0, 1, 2, 3, 4, 5};               /* mtf(selectors, maxValue:) :265-275 */
    for (int k = 0; k < n_sel; k++) {
        int idx = 0;
        while (order[idx] != selectors[k]) idx++;
        for (int j = 0; j < idx; j++) mbw_bit(w, 1);
        mbw_bit(w, 0);
        const int old = order[idx];
        memmove(order + 1, order, sizeof(int) * (size_t)idx);
        order[0] = old;
    }
    for (int t = 0; t < n_tab; t++) {
        int prev = tab_len[t][0];
        mbw_bits(w, (uint64_t)prev, 5);
        for (int s = 0; s < alpha; s++) {
            const int len = tab_len[t][s];
            for (int d = prev; d > len; d--) { mbw_bit(w, 1); mbw_bit(w, 1); }
            for (int d = prev; d < len; d++) { mbw_bit(w, 1); mbw_bit(w, 0); }
            prev = len;
            mbw_bit(w, 0);
        }
    }
    /* contents :223-240 */
    uint32_t code[6][258];
    for (int t = 0; t < n_tab; t++) codes_from(tab_len[t], alpha, code[t]);
    for (int i = 0; i < ns; i++) {
        const int t = selectors[i / 50];
        mbw_bits(w, code[t][sym[i]], tab_len[t][sym[i]]);
    }
    free(rle); free(col); free(sym); free(selectors);
}

/* compress(data:blockSize:) :41-74.  *out_len = bytes of the stream (written as far as `cap` reaches). */
int refcpu_bzip2_compress(const uint8_t* data, size_t n, int level, uint8_t* out, size_t cap, size_t* out_len) {
    if (level < 1 || level > 9) return SWC_E_INVALID_ARGUMENT;
    mbw_t w = {out, cap, 0, 0, 0};
    const size_t raw = (size_t)level * 100 * 800;                                        /* :46 */
    mbw_bits(&w, 0x425a, 16); mbw_bits(&w, 0x68, 8); mbw_bits(&w, (uint64_t)(0x30 + level), 8);
    uint32_t total = 0;
    for (size_t i = 0; i < n; i += raw) {
        const size_t len = n - i < raw ? n - i : raw;
        const uint32_t crc = refcpu_bzip2crc32(data + i, len);
        total = ((total << 1) | (total >> 31)) ^ crc;                                    /* :56-57 */
        mbw_bits(&w, 0x314159265359ull, 48);
        mbw_bits(&w, crc, 32);
        process_block(data + i, (int)len, &w);
    }
    mbw_bits(&w, 0x177245385090ull, 48);
    mbw_bits(&w, total, 32);
    mbw_align(&w);
    *out_len = w.n;
    return w.n <= cap ? SWC_OK : SWC_E_CAPACITY;
}
