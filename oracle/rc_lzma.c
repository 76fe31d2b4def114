/*
 * rc_lzma.c -- ORACLE (test infrastructure).  Restates
 *   LZMADecoder            Sources/LZMA/LZMADecoder.swift:79-298
 *   LZMARangeDecoder       Sources/LZMA/LZMARangeDecoder.swift:20-80
 *   LZMABitTreeDecoder     Sources/LZMA/LZMABitTreeDecoder.swift:18-43
 *   LZMALenDecoder         Sources/LZMA/LZMALenDecoder.swift:24-38
 *   LZMAProperties         Sources/LZMA/LZMAProperties.swift:50-65
 *   LZMA.decompress        Sources/LZMA/LZMA.swift:25-73
 *   LZMA2Decoder           Sources/LZMA2/LZMA2Decoder.swift:17-99,  LZMA2.decompress LZMA2.swift:25-36
 * "D:" line numbers refer to LZMADecoder.swift.
 *
 * Probability cells are kept in the reference's own index layout (D:50-62): probabilities[432]
 * (index 432 is out of range => trap, App. A L1), literalProbs[1<<(lc+lp)][0x300], 4 posSlot trees,
 * align tree, posDecoders[115], two length decoders.
 */
#include "rc_common.h"

#define PROB_INIT 1024

typedef struct len_dec { /* LZMALenDecoder.swift:6-22 */
    int choice, choice2;
    int low[16][8], mid[16][8], high[256];
} len_dec;

typedef struct lzma_dec {
    rc_bytes* rd;
    int lc, lp, pb;
    int64_t dict_size;
    int64_t uncompressed_size;
    rc_buf out;
    int64_t dict_start, dict_end;
    /* range decoder */
    uint32_t range, code;
    /* model */
    int have_model;            /* resetStateAndDecoders() has run at least once */
    int probabilities[432];
    int* literal_probs;        /* [(1<<(lc+lp)) * 0x300] */
    int n_lit_tables;
    int pos_slot[4][64];
    int align[16];
    int pos_decoders[115];
    len_dec len, rep_len;
    int64_t rep0, rep1, rep2, rep3;
    int state;
    int trap;                  /* reference would trap */
} lzma_dec;

static void len_init(len_dec* l) {
    l->choice = l->choice2 = PROB_INIT;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 8; j++) l->low[i][j] = l->mid[i][j] = PROB_INIT;
    for (int i = 0; i < 256; i++) l->high[i] = PROB_INIT;
}

/* D:79-100 */
static void reset_state_and_decoders(lzma_dec* d) {
    d->state = 0;
    d->rep0 = d->rep1 = d->rep2 = d->rep3 = 0;
    for (int i = 0; i < 432; i++) d->probabilities[i] = PROB_INIT;
    free(d->literal_probs);
    int sh = d->lc + d->lp;
    if (sh > 12) { d->trap = 1; sh = 12; } /* lc<=8, lp<=4 from the props byte */
    d->n_lit_tables = 1 << sh;
    d->literal_probs = (int*)malloc((size_t)d->n_lit_tables * 0x300 * sizeof(int));
    for (int i = 0; i < d->n_lit_tables * 0x300; i++) d->literal_probs[i] = PROB_INIT;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 64; j++) d->pos_slot[i][j] = PROB_INIT;
    for (int i = 0; i < 16; i++) d->align[i] = PROB_INIT;
    for (int i = 0; i < 115; i++) d->pos_decoders[i] = PROB_INIT;
    len_init(&d->len);
    len_init(&d->rep_len);
    d->have_model = 1;
}

/* LZMARangeDecoder.swift:38-43 */
static inline void rd_normalize(lzma_dec* d) {
    if (d->range < (1u << 24)) {
        d->range <<= 8;
        uint8_t b = rc_b_byte(d->rd); /* unchecked read past the end => trap (App. A L4) */
        d->code = (d->code << 8) | b;
    }
}
/* LZMARangeDecoder.swift:65-80 */
static inline int rd_bit(lzma_dec* d, int* prob) {
    uint32_t bound = (d->range >> 11) * (uint32_t)*prob;
    int sym;
    if (d->code < bound) {
        *prob += ((1 << 11) - *prob) >> 5;
        d->range = bound;
        sym = 0;
    } else {
        *prob -= *prob >> 5;
        d->code -= bound;
        d->range -= bound;
        sym = 1;
    }
    rd_normalize(d);
    return sym;
}
/* LZMARangeDecoder.swift:46-62 (repeat-while: executes once even for 0 bits -- only reachable with
 * numDirectBits-4 >= 2, so never with 0) */
static inline uint32_t rd_direct(lzma_dec* d, int count) {
    uint32_t res = 0;
    do {
        d->range >>= 1;
        d->code -= d->range;
        uint32_t t = 0u - (d->code >> 31);
        d->code += d->range & t;
        rd_normalize(d);
        res = (res << 1) + (t + 1);
        count--;
    } while (count > 0);
    return res;
}
static inline int bt_decode(lzma_dec* d, int* probs, int nbits) { /* LZMABitTreeDecoder.swift:18-24 */
    int m = 1;
    for (int i = 0; i < nbits; i++) m = (m << 1) + rd_bit(d, &probs[m]);
    return m - (1 << nbits);
}
static inline int bt_reverse(lzma_dec* d, int* probs, int n_probs, int start, int bits) { /* :26-43 */
    int m = 1, sym = 0;
    for (int i = 0; i < bits; i++) {
        int idx = start + m;
        if (idx < 0 || idx >= n_probs) { d->trap = 1; return 0; }
        int bit = rd_bit(d, &probs[idx]);
        m = (m << 1) + bit;
        sym |= bit << i;
    }
    return sym;
}
static inline int len_decode(lzma_dec* d, len_dec* l, int pos_state) { /* LZMALenDecoder.swift:30-38 */
    if (rd_bit(d, &l->choice) == 0) return bt_decode(d, l->low[pos_state], 3);
    if (rd_bit(d, &l->choice2) == 0) return 8 + bt_decode(d, l->mid[pos_state], 3);
    return 16 + bt_decode(d, l->high, 8);
}

/* D:288-294 */
static inline int dec_put(lzma_dec* d, uint8_t b) {
    if (!rc_buf_put(&d->out, b)) return 0;
    d->dict_end += 1;
    if (d->dict_end - d->dict_start == d->dict_size) d->dict_start += 1;
    return 1;
}
/* D:296-298: out[distance <= dictEnd ? dictEnd - distance : dictSize - distance + dictEnd] */
static inline uint8_t dec_byte_at(lzma_dec* d, int64_t distance) {
    int64_t idx = distance <= d->dict_end ? d->dict_end - distance : d->dict_size - distance + d->dict_end;
    if (idx < 0 || idx >= (int64_t)d->out.len) { d->trap = 1; return 0; }
    return d->out.p[idx];
}

#define FAIL(x) do { CHECK_TRAP(); return (x); } while (0)
#define CHECK_TRAP() do { if (d->trap || d->rd->trap) return SWC_E_REF_TRAP; if (d->out.overflow) return SWC_E_CAPACITY; } while (0)

/* D:107-284 */
static int lzma_decode(lzma_dec* d) {
    /* LZMARangeDecoder.init :20-31 */
    if (rc_b_left(d->rd) < 5) return SWC_E_LZMA_RANGE_DECODER_INIT_ERROR;
    uint8_t first = rc_b_byte(d->rd);
    uint32_t le = (uint32_t)rc_b_le(d->rd, 4);
    d->code = (le >> 24) | ((le >> 8) & 0xFF00) | ((le << 8) & 0xFF0000) | (le << 24);
    d->range = 0xFFFFFFFFu;
    if (first != 0) return SWC_E_LZMA_RANGE_DECODER_INIT_ERROR;
    if (!d->have_model) return SWC_E_REF_TRAP; /* probabilities == [] => index trap at D:119 */

    for (;;) {
        if (d->uncompressed_size == 0 && d->code == 0) break; /* D:114 */
        int pos_state = (int)((int64_t)d->out.len & ((1 << d->pb) - 1));
        if (rd_bit(d, &d->probabilities[(d->state << 4) + pos_state]) == 0) {
            CHECK_TRAP();
            if (d->uncompressed_size == 0) FAIL(SWC_E_LZMA_EXCEEDED_UNCOMPRESSED_SIZE); /* D:121 */
            int prev = d->dict_end == d->dict_start ? 0 : dec_byte_at(d, 1);
            int symbol = 1;
            int lit_state = (int)((((int64_t)d->out.len & ((1 << d->lp) - 1)) << d->lc) + (prev >> (8 - d->lc)));
            if (lit_state < 0 || lit_state >= d->n_lit_tables) return SWC_E_REF_TRAP;
            int* lp = d->literal_probs + (size_t)lit_state * 0x300;
            if (d->state >= 7) {
                uint8_t match_byte = dec_byte_at(d, d->rep0 + 1);
                CHECK_TRAP();
                do {
                    int match_bit = (match_byte >> 7) & 1;
                    match_byte = (uint8_t)(match_byte << 1);
                    int bit = rd_bit(d, &lp[((1 + match_bit) << 8) + symbol]);
                    symbol = (symbol << 1) | bit;
                    if (match_bit != bit) break;
                } while (symbol < 0x100);
            }
            while (symbol < 0x100) symbol = (symbol << 1) | rd_bit(d, &lp[symbol]);
            CHECK_TRAP();
            d->uncompressed_size -= 1;
            dec_put(d, (uint8_t)(symbol - 0x100));
            CHECK_TRAP();
            if (d->state < 4) d->state = 0; else if (d->state < 10) d->state -= 3; else d->state -= 6;
            continue;
        }
        CHECK_TRAP();

        int len;
        if (rd_bit(d, &d->probabilities[193 + d->state]) != 0) {
            if (d->uncompressed_size == 0) FAIL(SWC_E_LZMA_EXCEEDED_UNCOMPRESSED_SIZE); /* D:178 */
            if (d->dict_end == d->dict_start) FAIL(SWC_E_LZMA_WINDOW_IS_EMPTY);        /* D:181 */
            if (rd_bit(d, &d->probabilities[205 + d->state]) == 0) {
                int idx = 241 + (d->state << 4) + pos_state;
                if (idx >= 432) return SWC_E_REF_TRAP; /* App. A L1 */
                if (rd_bit(d, &d->probabilities[idx]) == 0) {
                    CHECK_TRAP();
                    d->state = d->state < 7 ? 9 : 11;
                    uint8_t b = dec_byte_at(d, d->rep0 + 1);
                    CHECK_TRAP();
                    dec_put(d, b);
                    d->uncompressed_size -= 1;
                    CHECK_TRAP();
                    continue;
                }
            } else {
                int64_t dist;
                if (rd_bit(d, &d->probabilities[217 + d->state]) == 0) {
                    dist = d->rep1;
                } else {
                    if (rd_bit(d, &d->probabilities[229 + d->state]) == 0) {
                        dist = d->rep2;
                    } else {
                        dist = d->rep3;
                        d->rep3 = d->rep2;
                    }
                    d->rep2 = d->rep1;
                }
                d->rep1 = d->rep0;
                d->rep0 = dist;
            }
            len = len_decode(d, &d->rep_len, pos_state);
            d->state = d->state < 7 ? 8 : 11;
        } else {
            d->rep3 = d->rep2; d->rep2 = d->rep1; d->rep1 = d->rep0;
            len = len_decode(d, &d->len, pos_state);
            d->state = d->state < 7 ? 7 : 10;
            int len_state = len > 3 ? 3 : len;
            int pos_slot = bt_decode(d, d->pos_slot[len_state], 6);
            if (pos_slot < 4) {
                d->rep0 = pos_slot;
            } else {
                int ndb = (pos_slot >> 1) - 1;
                int64_t dist = (int64_t)(2 | (pos_slot & 1)) << ndb;
                if (pos_slot < 14) {
                    dist += bt_reverse(d, d->pos_decoders, 115, (int)(dist - pos_slot), ndb);
                } else {
                    dist += (int64_t)rd_direct(d, ndb - 4) << 4;
                    dist += bt_reverse(d, d->align, 16, 0, 4);
                }
                d->rep0 = dist;
            }
            CHECK_TRAP();
            /* D:260 UInt32(rep0): rep0 > UInt32.max traps.  Max reachable value: posSlot 63 =>
             * (3<<30) + (2^26-1)<<4 + 15 = 0xFFFFFFFF, so it never exceeds UInt32.max. */
            if ((uint64_t)d->rep0 == 0xFFFFFFFFull) {
                if (d->code != 0) FAIL(SWC_E_LZMA_RANGE_DECODER_FINISH_ERROR); /* D:261 */
                break;
            }
            if (d->uncompressed_size == 0) FAIL(SWC_E_LZMA_EXCEEDED_UNCOMPRESSED_SIZE); /* D:266 */
            if (d->rep0 >= d->dict_size || (d->rep0 > d->dict_end && d->dict_end < d->dict_size))
                FAIL(SWC_E_LZMA_NOT_ENOUGH_TO_REPEAT); /* D:269 */
        }
        CHECK_TRAP();
        len += 2;
        if (d->uncompressed_size > -1 && d->uncompressed_size < len) FAIL(SWC_E_LZMA_REPEAT_WILL_EXCEED); /* D:275 */
        for (int i = 0; i < len; i++) {
            uint8_t b = dec_byte_at(d, d->rep0 + 1);
            CHECK_TRAP();
            dec_put(d, b);
            d->uncompressed_size -= 1;
        }
        CHECK_TRAP();
    }
    CHECK_TRAP();
    return SWC_OK;
}

static void dec_init(lzma_dec* d, rc_bytes* rd) {
    memset(d, 0, sizeof *d);
    d->rd = rd;
    d->lc = 3; d->lp = 0; d->pb = 2;       /* LZMAProperties defaults :12-18 */
    d->dict_size = 1 << 24;
    d->uncompressed_size = -1;
    rc_buf_init(&d->out);
}
static void dec_free(lzma_dec* d) { free(d->literal_probs); d->literal_probs = NULL; }

int refcpu_lzma_decompress(const uint8_t* in, size_t in_len, int lc, int lp, int pb, int64_t dict_size,
                           int64_t uncompressed_size, uint8_t** out, size_t* out_len, size_t* in_consumed) {
    rc_bytes rd; lzma_dec d;
    rc_bytes_init(&rd, in, in_len);
    dec_init(&d, &rd);
    d.lc = lc; d.lp = lp; d.pb = pb; d.dict_size = dict_size; /* no clamp: didSet does not run inside init */
    int st;
    if (lc < 0 || lc > 8 || lp < 0 || lp > 4 || pb < 0 || pb > 4) {
        st = SWC_E_REF_TRAP; /* the public initializer does not validate; such values trap downstream */
    } else {
        reset_state_and_decoders(&d);
        d.uncompressed_size = uncompressed_size;
        st = lzma_decode(&d);
    }
    if (in_consumed) *in_consumed = rd.off < 0 ? 0 : (size_t)rd.off > in_len ? in_len : (size_t)rd.off;
    rc_buf_release(&d.out, out, out_len);
    dec_free(&d);
    return st;
}

/* LZMA.decompress(data:) LZMA.swift:25-34 + LZMAProperties.init(lzmaByte:_:) :50-58 */
int refcpu_lzma_alone_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) {
    if (in_len < 13) { *out = (uint8_t*)malloc(1); *out_len = 0; return SWC_E_LZMA_WRONG_PROPERTIES; }
    unsigned b = in[0];
    if (b >= 225) { *out = (uint8_t*)malloc(1); *out_len = 0; return SWC_E_LZMA_WRONG_PROPERTIES; }
    int lc = b % 9, pb = (b / 9) / 5, lp = (b / 9) % 5;
    int64_t dict = (int64_t)((uint32_t)in[1] | (uint32_t)in[2] << 8 | (uint32_t)in[3] << 16 | (uint32_t)in[4] << 24);
    uint64_t us = 0;
    for (int i = 0; i < 8; i++) us |= (uint64_t)in[5 + i] << (8 * i);
    size_t consumed;
    return refcpu_lzma_decompress(in + 13, in_len - 13, lc, lp, pb, dict, (int64_t)us, out, out_len, &consumed);
}

/* ---- LZMA2 ---- */
int refcpu_lzma2_decompress(const uint8_t* in, size_t in_len, uint8_t dict_byte, uint8_t** out,
                            size_t* out_len, size_t* in_consumed) {
    rc_bytes rd; lzma_dec d;
    rc_bytes_init(&rd, in, in_len);
    dec_init(&d, &rd);
    int st = SWC_OK;
    do {
        /* LZMA2Decoder.init :17-31 */
        if (dict_byte & 0xC0) { st = SWC_E_LZMA2_WRONG_DICTIONARY_SIZE; break; }
        int bits = dict_byte & 0x3F;
        if (bits >= 40) { st = SWC_E_LZMA2_WRONG_DICTIONARY_SIZE; break; }
        /* UInt32 << n is a smart shift: n >= 32 yields 0 (bits/2+11 <= 30 here, so no overflow to 0
         * except the high bit cases which still fit in UInt32 for bits <= 39) */
        uint32_t ds = (uint32_t)(2 | (bits & 1)) << (bits / 2 + 11);
        d.dict_size = ds < 4096 ? 4096 : ds; /* didSet clamp LZMAProperties.swift:26-32 */

        for (;;) { /* decode() :34-53 */
            unsigned control = rc_b_byte(&rd);
            if (rd.trap) { st = SWC_E_REF_TRAP; break; }
            if (control == 0) break;
            if (control == 1 || control == 2) {
                if (control == 1) d.dict_start = d.dict_end; /* resetDictionary D:102-104 */
                /* decodeUncompressed :84-89 */
                unsigned b1 = rc_b_byte(&rd), b2 = rc_b_byte(&rd);
                size_t sz = ((size_t)b1 << 8) + b2 + 1;
                for (size_t i = 0; i < sz && !rd.trap && !d.out.overflow; i++) dec_put(&d, rc_b_byte(&rd));
                if (rd.trap) { st = SWC_E_REF_TRAP; break; }
                if (d.out.overflow) { st = SWC_E_CAPACITY; break; }
                continue;
            }
            if (control <= 0x7F) { st = SWC_E_LZMA2_WRONG_CONTROL_BYTE; break; }
            /* dispatch :56-82 */
            int reset = (control & 0x60) >> 5;
            unsigned u1 = rc_b_byte(&rd), u2 = rc_b_byte(&rd);
            int64_t unpack = ((int64_t)(control & 0x1F) << 16) + ((int64_t)u1 << 8) + u2 + 1;
            unsigned c1 = rc_b_byte(&rd), c2 = rc_b_byte(&rd);
            int64_t comp = ((int64_t)c1 << 8) + c2 + 1;
            if (rd.trap) { st = SWC_E_REF_TRAP; break; }
            if (reset == 1) {
                reset_state_and_decoders(&d);
            } else if (reset >= 2) { /* updateProperties :95-99 */
                unsigned pbyte = rc_b_byte(&rd);
                if (rd.trap) { st = SWC_E_REF_TRAP; break; }
                if (pbyte >= 225) { st = SWC_E_LZMA_WRONG_PROPERTIES; break; }
                d.lc = pbyte % 9; d.pb = (pbyte / 9) / 5; d.lp = (pbyte / 9) % 5;
                reset_state_and_decoders(&d);
                if (reset == 3) d.dict_start = d.dict_end;
            }
            d.uncompressed_size = unpack;
            size_t out_start = d.out.len;
            int64_t in_start = rd.off;
            st = lzma_decode(&d);
            if (st) break;
            if (!(unpack == (int64_t)(d.out.len - out_start) && rd.off - in_start == comp)) { st = SWC_E_LZMA2_WRONG_SIZES; break; }
        }
    } while (0);
    if (in_consumed) *in_consumed = rd.off < 0 ? 0 : (size_t)rd.off > in_len ? in_len : (size_t)rd.off;
    rc_buf_release(&d.out, out, out_len);
    dec_free(&d);
    return st;
}

int refcpu_lzma2_decompress_data(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) {
    if (in_len < 1) { *out = (uint8_t*)malloc(1); *out_len = 0; return SWC_E_LZMA_RANGE_DECODER_INIT_ERROR; } /* LZMA2.swift:27 */
    size_t consumed;
    return refcpu_lzma2_decompress(in + 1, in_len - 1, in[0], out, out_len, &consumed);
}
