// lz4_lane.h -- LZ4 block decode, one block per lane.
//
// Replaces the body of LZ4.process(block:_:) (reference Sources/LZ4/LZ4.swift:332-413): token /
// length parsing, the end-of-block rules the reference enforces (:369-376), offset
// validation against dictionary + bytes produced (:380-383), and the overlapping match copy
// (:403-409).  The reference copies one byte per `append`; here literals and matches move 8 bytes per
// load/store, overlapping matches replicate their `offset`-byte pattern in a register first.
//
// Used for batches of many small blocks (one lane each, 64 blocks per wave).  Large blocks go through
// the wave-cooperative kernel in lz4_wave.h; both produce identical bytes and statuses.
#ifndef SWC_LZ4_LANE_H
#define SWC_LZ4_LANE_H

#include "swc_common.h"

namespace swc {
namespace lz4 {

// `dict` (may be null) is the prefix the block may reference: the whole external dictionary for
// independent blocks, the last 64 KiB of output for dependent ones (LZ4.swift:304-313).
SWC_HD void lz4_block_job(Job& job) {
    gcptr in = (gcptr)job.in;
    const uint64_t n = job.in_len;
    gptr out = (gptr)job.out;
    const uint64_t cap = job.out_cap;
    gcptr dict = (gcptr)job.dict;
    const uint64_t dlen = dict ? job.dict_len : 0;

    uint64_t ip = 0, pos = 0;           // pos keeps counting past cap (size pass for SWC_E_CAPACITY)
    uint64_t sequences = 0;
    int64_t last_match_start = -1;      // in (dict ++ out) index space
    int st = SWC_OK;

    for (;;) {
        sequences++;
        if (n - ip < 1) { st = SWC_E_DATA_TRUNCATED; break; }                      // :344
        const uint32_t token = in[ip++];
        uint64_t lit = token >> 4;
        if (lit == 15) {
            for (;;) {
                if (n - ip < 1) { st = SWC_E_DATA_TRUNCATED; break; }              // :350
                const uint32_t b = in[ip++];
                lit += b;   // Int overflow (:355 unsupportedFeature) needs > 2^55 input bytes: unreachable
                if (b != 255) break;
            }
            if (st) break;
        }
        if (n - ip < lit) { st = SWC_E_DATA_TRUNCATED; break; }                    // :363
        {   // literals: input -> output
            uint64_t i = 0;
            if (pos + lit + 8 <= cap && ip + lit + 8 <= n) {
                for (; i < lit; i += 8) store_u64(out + pos + i, load_u64(in + ip + i));  // may overrun < 8 B inside both buffers
            } else {
                for (; i < lit; i++)
                    if (pos + i < cap) out[pos + i] = in[ip + i];
            }
        }
        ip += lit;
        pos += lit;
        const uint64_t produced = dlen + pos;                                      // out.endIndex of the reference
        if (ip >= n) {                                                             // :368 last sequence: literals only
            if (!(lit >= 5 || sequences == 1)) st = SWC_E_DATA_CORRUPTED;          // :370
            else if (!((int64_t)produced - last_match_start >= 12 || last_match_start == -1)) st = SWC_E_DATA_CORRUPTED;  // :372
            break;
        }
        if (n - ip < 2) { st = SWC_E_DATA_TRUNCATED; break; }                      // :378
        const uint64_t offset = (uint64_t)in[ip] | ((uint64_t)in[ip + 1] << 8);
        ip += 2;
        if (!(offset > 0 && offset <= produced)) { st = SWC_E_DATA_CORRUPTED; break; }  // :382
        uint64_t mlen = 4 + (token & 0xF);
        if (mlen == 19) {
            for (;;) {
                if (n - ip < 1) { st = SWC_E_DATA_TRUNCATED; break; }              // :388
                const uint32_t b = in[ip++];
                mlen += b;
                if (b != 255) break;
            }
            if (st) break;
        }
        last_match_start = (int64_t)produced;

        // match copy :403-409
        uint64_t done = 0;
        if (offset > pos) {
            // source starts inside the dictionary prefix: byte-wise until it crosses into `out`
            const uint64_t in_dict = offset - pos;             // bytes available before the source reaches out[0]
            const uint64_t k = in_dict < mlen ? in_dict : mlen;
            for (uint64_t i = 0; i < k; i++)
                if (pos + i < cap) out[pos + i] = dict[dlen - in_dict + i];
            done = k;
        }
        if (done < mlen) {
            uint64_t p = pos + done, rem = mlen - done;
            if (p + rem + 8 <= cap) {
                uint64_t d = offset;
                if (d < 8) {
                    // replicate the d-byte pattern to 8 bytes, then continue at the smallest multiple of d >= 8
                    uint32_t sh = 8 * (uint32_t)d;
                    uint64_t w = load_u64(out + p - d) & ((1ull << sh) - 1ull);
                    w |= w << sh;
                    sh *= 2;
                    if (sh < 64) { w |= w << sh; sh *= 2; }
                    if (sh < 64) w |= w << sh;
                    store_u64(out + p, w);
                    const uint64_t k = rem < 8 ? rem : 8;
                    p += k;
                    rem -= k;
                    d = ((7 + d) / d) * d;
                }
                for (uint64_t i = 0; i < rem; i += 8) store_u64(out + p + i, load_u64(out + p + i - d));
            } else {
                for (uint64_t i = 0; i < rem; i++)
                    if (p + i < cap) out[p + i] = out[p + i - offset];
            }
        }
        pos += mlen;
    }
    if (st == SWC_OK && pos > cap) st = SWC_E_CAPACITY;
    job.out_len = pos;
    job.in_consumed = ip;
    job.status = st;
}

}  // namespace lz4
}  // namespace swc
#endif
