/*
 * rc_lz4.c -- ORACLE (test infrastructure).  Restates Sources/LZ4/LZ4.swift:
 *   decompress(data:dictionary:dictionaryID:) :73-91     multiDecompress :116-146
 *   process(skippableFrame:) :148-155                    process(legacyFrame:) :160-186
 *   process(frame:_:_:) :188-330                         process(block:_:) :332-413
 * Line numbers in comments refer to LZ4.swift.
 */
#include "rc_common.h"

/* process(block:_:) :332-413.  `dict` (may be NULL) is the prefix the block may reference; the
 * decoded bytes are appended to `out`.  Offsets are validated against dict_len + produced. */
static int lz4_block(const uint8_t* in, size_t n, const uint8_t* dict, size_t dict_len, rc_buf* out) {
    size_t ip = 0;
    size_t base = out->len;          /* block output starts here */
    int64_t sequence_count = 0;
    int64_t last_match_start = -1;   /* index in (dict ++ block-output) space */

    for (;;) {
        sequence_count++;
        if (n - ip < 1) return SWC_E_DATA_TRUNCATED; /* :344 */
        unsigned token = in[ip++];
        uint64_t literal_count = token >> 4;
        if (literal_count == 15) {
            for (;;) {
                if (n - ip < 1) return SWC_E_DATA_TRUNCATED; /* :350 */
                unsigned b = in[ip++];
                literal_count += b; /* Int overflow (:355) is unreachable with < 2^63 input bytes */
                if (b != 255) break;
            }
        }
        if ((uint64_t)(n - ip) < literal_count) return SWC_E_DATA_TRUNCATED; /* :363 */
        if (!rc_buf_append(out, in + ip, (size_t)literal_count)) return SWC_E_CAPACITY;
        ip += (size_t)literal_count;

        size_t produced = dict_len + (out->len - base); /* out.endIndex in the reference */
        if (ip >= n) { /* reader.isFinished :368 */
            if (!(literal_count >= 5 || sequence_count == 1)) return SWC_E_DATA_CORRUPTED; /* :370 */
            if (!((int64_t)produced - last_match_start >= 12 || last_match_start == -1)) return SWC_E_DATA_CORRUPTED; /* :372 */
            break;
        }
        if (n - ip < 2) return SWC_E_DATA_TRUNCATED; /* :378 */
        size_t offset = (size_t)in[ip] | (size_t)in[ip + 1] << 8;
        ip += 2;
        if (!(offset > 0 && offset <= produced)) return SWC_E_DATA_CORRUPTED; /* :382 */

        uint64_t match_length = 4 + (token & 0xF);
        if (match_length == 19) {
            for (;;) {
                if (n - ip < 1) return SWC_E_DATA_TRUNCATED; /* :388 */
                unsigned b = in[ip++];
                match_length += b;
                if (b != 255) break;
            }
        }
        last_match_start = (int64_t)produced;
        if (!rc_buf_reserve(out, (size_t)match_length)) return SWC_E_CAPACITY;
        /* :404-409 byte-wise copy from (dict ++ out) */
        size_t cur = out->len - base; /* bytes of this block produced so far */
        for (uint64_t i = 0; i < match_length; i++) {
            int64_t src = (int64_t)(dict_len + cur + i) - (int64_t)offset; /* index in dict++out space */
            uint8_t v = src < (int64_t)dict_len ? dict[src] : out->p[base + (size_t)(src - (int64_t)dict_len)];
            out->p[out->len++] = v;
        }
    }
    return SWC_OK;
}

int refcpu_lz4_block(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                     uint8_t** out, size_t* out_len) {
    rc_buf b;
    rc_buf_init(&b);
    int st = lz4_block(in, in_len, dict, dict_len, &b);
    rc_buf_release(&b, out, out_len);
    return st;
}

/* process(skippableFrame:) :148-155 -- data starts after the magic; returns size+4 */
static int lz4_skippable(const uint8_t* p, size_t n, size_t* adv) {
    if (n < 4) return SWC_E_DATA_TRUNCATED;
    size_t size = (size_t)p[0] | (size_t)p[1] << 8 | (size_t)p[2] << 16 | (size_t)p[3] << 24;
    if (n < size + 4) return SWC_E_DATA_TRUNCATED;
    *adv = size + 4;
    return SWC_OK;
}

static int is_lz4_magic(uint32_t m) {
    return m == 0x184D2204u || m == 0x184C2102u || (m >= 0x184D2A50u && m <= 0x184D2A5Fu);
}

/* process(legacyFrame:) :160-186 -- p points after the magic */
static int lz4_legacy(const uint8_t* p, size_t n, rc_buf* out, size_t* adv) {
    size_t off = 0;
    while (off < n) {
        if (n - off < 4) return SWC_E_DATA_TRUNCATED; /* :165 */
        uint32_t raw = (uint32_t)p[off] | (uint32_t)p[off + 1] << 8 | (uint32_t)p[off + 2] << 16 | (uint32_t)p[off + 3] << 24;
        off += 4;
        if (is_lz4_magic(raw)) { off -= 4; break; } /* :168-171 */
        size_t bs = raw;
        if (n - off < bs) return SWC_E_DATA_TRUNCATED; /* :177 */
        int st = lz4_block(p + off, bs, NULL, 0, out);
        if (st) return st;
        off += bs;
    }
    *adv = off;
    return SWC_OK;
}

/* process(frame:_:_:) :188-330 -- p points after the magic */
static int lz4_frame(const uint8_t* p, size_t n, const uint8_t* dict, size_t dict_len, int have_dict,
                     int64_t ext_dict_id, rc_buf* out, size_t* adv) {
    size_t start = out->len; /* this frame's `out` */
    if (n < 7) return SWC_E_DATA_TRUNCATED; /* :191 */
    size_t off = 0;
    unsigned flg = p[off++];
    if (!(((flg & 0xC0) >> 6) == 1 && (flg & 0x2) == 0)) return SWC_E_DATA_CORRUPTED; /* :198 */
    int independent = (flg & 0x20) != 0, block_checksum = (flg & 0x10) != 0;
    int content_size_present = (flg & 0x8) != 0, content_checksum = (flg & 0x4) != 0, dict_id_present = (flg & 1) != 0;
    unsigned bd = p[off++];
    size_t max_block;
    switch (bd) { /* :216-228 */
        case 0x40: max_block = 64 * 1024; break;
        case 0x50: max_block = 256 * 1024; break;
        case 0x60: max_block = 1024 * 1024; break;
        case 0x70: max_block = 4 * 1024 * 1024; break;
        default: return SWC_E_DATA_CORRUPTED;
    }
    uint64_t content_size = 0;
    if (content_size_present) {
        if (n - off < 13) return SWC_E_DATA_TRUNCATED; /* :234 */
        for (int i = 0; i < 8; i++) content_size |= (uint64_t)p[off + i] << (8 * i);
        off += 8;
        if (content_size > (uint64_t)INT64_MAX) return SWC_E_DATA_UNSUPPORTED_FEATURE; /* :240 */
    }
    int64_t dict_id = -1;
    if (dict_id_present) {
        if (!have_dict) return SWC_E_DATA_CORRUPTED; /* :250 */
        if (n - off < 9) return SWC_E_DATA_TRUNCATED; /* :254 */
        dict_id = (int64_t)((uint32_t)p[off] | (uint32_t)p[off + 1] << 8 | (uint32_t)p[off + 2] << 16 | (uint32_t)p[off + 3] << 24);
        off += 4;
    }
    if (ext_dict_id >= 0 && dict_id >= 0 && ext_dict_id != dict_id) return SWC_E_DATA_CORRUPTED; /* :266-270 */
    /* header checksum over the descriptor :272-275 (n >= 7 guarantees the byte exists) */
    uint32_t hc = refcpu_xxh32(p, off, 0);
    if ((uint8_t)((hc >> 8) & 0xFF) != p[off]) return SWC_E_DATA_CORRUPTED;
    off++;

    for (;;) {
        if (n - off < 4) return SWC_E_DATA_TRUNCATED; /* :279 */
        uint32_t mark = (uint32_t)p[off] | (uint32_t)p[off + 1] << 8 | (uint32_t)p[off + 2] << 16 | (uint32_t)p[off + 3] << 24;
        off += 4;
        if (mark == 0) break; /* EndMark :284 */
        int compressed = (mark & 0x80000000u) == 0;
        size_t bs = mark & 0x7FFFFFFFu;
        if (bs > max_block) return SWC_E_DATA_CORRUPTED; /* :292 */
        if (n - off < bs + (block_checksum ? 4 : 0) + 4) return SWC_E_DATA_TRUNCATED; /* :295 */
        const uint8_t* bdata = p + off;
        off += bs;
        if (block_checksum) {
            uint32_t c = (uint32_t)p[off] | (uint32_t)p[off + 1] << 8 | (uint32_t)p[off + 2] << 16 | (uint32_t)p[off + 3] << 24;
            off += 4;
            if (refcpu_xxh32(bdata, bs, 0) != c) return SWC_E_DATA_CORRUPTED; /* :300 */
        }
        if (compressed) {
            int st;
            size_t produced = out->len - start;
            if (independent) {
                st = lz4_block(bdata, bs, have_dict ? dict : NULL, have_dict ? dict_len : 0, out); /* :305 */
            } else if (produced == 0 && have_dict) { /* :307-309 last 64 KiB of the dictionary */
                size_t dl = dict_len > 65536 ? 65536 : dict_len;
                st = lz4_block(bdata, bs, dict + (dict_len - dl), dl, out);
            } else { /* :311-312 last 64 KiB of this frame's output; copy because `out` may realloc */
                size_t dl = produced > 65536 ? 65536 : produced;
                uint8_t* tmp = (uint8_t*)malloc(dl ? dl : 1);
                if (dl) memcpy(tmp, out->p + out->len - dl, dl);
                st = lz4_block(bdata, bs, tmp, dl, out);
                free(tmp);
            }
            if (st) return st;
        } else {
            if (!rc_buf_append(out, bdata, bs)) return SWC_E_CAPACITY; /* :315 */
        }
    }
    if (content_size_present && (uint64_t)(out->len - start) != content_size) return SWC_E_DATA_CORRUPTED; /* :320 */
    if (content_checksum) {
        if (n - off < 4) return SWC_E_DATA_TRUNCATED; /* :324 */
        uint32_t c = (uint32_t)p[off] | (uint32_t)p[off + 1] << 8 | (uint32_t)p[off + 2] << 16 | (uint32_t)p[off + 3] << 24;
        off += 4;
        if (refcpu_xxh32(out->p + start, out->len - start, 0) != c) { *adv = off; return SWC_E_DATA_CHECKSUM_MISMATCH; } /* :326 */
    }
    *adv = off;
    return SWC_OK;
}

/* decompress(data:dictionary:dictionaryID:) :73-91.  Errors other than checksumMismatch carry no data. */
int refcpu_lz4_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                          int64_t dict_id, uint8_t** out, size_t* out_len, size_t* in_consumed) {
    rc_buf b;
    rc_buf_init(&b);
    int st = SWC_OK;
    size_t pos = 0, adv = 0;
    int have_dict = dict != NULL;
    for (;;) {
        if (in_len - pos < 4) { st = SWC_E_DATA_TRUNCATED; break; } /* :75 */
        uint32_t magic = (uint32_t)in[pos] | (uint32_t)in[pos + 1] << 8 | (uint32_t)in[pos + 2] << 16 | (uint32_t)in[pos + 3] << 24;
        pos += 4;
        if (magic == 0x184D2204u) {
            st = lz4_frame(in + pos, in_len - pos, dict, dict_len, have_dict, dict_id, &b, &adv);
            if (st == SWC_OK || st == SWC_E_DATA_CHECKSUM_MISMATCH) pos += adv;
            break;
        } else if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {
            st = lz4_skippable(in + pos, in_len - pos, &adv);
            if (st) break;
            pos += adv;
            /* :85 recursion WITHOUT the dictionary: LZ4.decompress(data:) */
            have_dict = 0; dict = NULL; dict_len = 0; dict_id = -1;
            continue;
        } else if (magic == 0x184C2102u) {
            st = lz4_legacy(in + pos, in_len - pos, &b, &adv);
            if (st == SWC_OK) pos += adv;
            break;
        } else {
            st = SWC_E_DATA_CORRUPTED;
            break;
        }
    }
    if (st != SWC_OK && st != SWC_E_DATA_CHECKSUM_MISMATCH) b.len = 0;
    if (in_consumed) *in_consumed = pos;
    rc_buf_release(&b, out, out_len);
    return st;
}

/* multiDecompress :116-146 */
int refcpu_lz4_multi_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                                int64_t dict_id, uint8_t** out, size_t* out_len, size_t** frame_sizes,
                                size_t* n_frames) {
    rc_buf b;
    rc_buf_init(&b);
    size_t cap = 16, nf = 0;
    size_t* sizes = (size_t*)malloc(cap * sizeof(size_t));
    int st = SWC_OK;
    size_t pos = 0;
    int have_dict = dict != NULL;
    do {
        if (pos + 4 > in_len) { st = SWC_E_DATA_TRUNCATED; break; } /* :123 */
        uint32_t magic = (uint32_t)in[pos] | (uint32_t)in[pos + 1] << 8 | (uint32_t)in[pos + 2] << 16 | (uint32_t)in[pos + 3] << 24;
        pos += 4;
        size_t adv = 0, start = b.len;
        int produced = 0;
        if (magic == 0x184D2204u) {
            st = lz4_frame(in + pos, in_len - pos, dict, dict_len, have_dict, dict_id, &b, &adv);
            produced = 1;
        } else if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {
            st = lz4_skippable(in + pos, in_len - pos, &adv);
        } else if (magic == 0x184C2102u) {
            st = lz4_legacy(in + pos, in_len - pos, &b, &adv);
            produced = 1;
        } else {
            st = SWC_E_DATA_CORRUPTED;
        }
        if (st == SWC_E_DATA_CHECKSUM_MISMATCH) {
            /* process(frame:) throws checksumMismatch([out]) with only the failing frame; the doc
             * comment promises "all frames up to and including" but the code (:326) carries one. */
            memmove(b.p, b.p + start, b.len - start);
            b.len -= start;
            nf = 0;
            sizes[nf++] = b.len;
            break;
        }
        if (st) { b.len = 0; nf = 0; break; }
        pos += adv;
        if (produced) {
            if (nf == cap) { cap *= 2; sizes = (size_t*)realloc(sizes, cap * sizeof(size_t)); }
            sizes[nf++] = b.len - start;
        }
    } while (pos < in_len);
    rc_buf_release(&b, out, out_len);
    *frame_sizes = sizes;
    *n_frames = nf;
    return st;
}
