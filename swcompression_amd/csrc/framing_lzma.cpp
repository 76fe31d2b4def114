// framing_lzma.cpp -- host side of LZMA / LZMA2 / XZ.
//   LZMA.decompress(data:)  reference Sources/LZMA/LZMA.swift:25-34 (.lzma header) and :56-73 (raw + properties)
//   LZMA2.decompress        reference Sources/LZMA2/LZMA2.swift:25-36
//   XZArchive.unarchive / splitUnarchive / processStream / processIndex / processFooter / processPadding
//                           reference Sources/XZ/XZArchive.swift:27-218
//   XZBlock.init            reference Sources/XZ/XZBlock.swift:18-97
//   XZStreamHeader.init     reference Sources/XZ/XZStreamHeader.swift:33-57
//   multiByteDecode         reference Sources/XZ/LittleEndianByteReader+XZ.swift:10-30
//   DeltaFilter.decode      reference Sources/Common/DeltaFilter.swift:11-33
// Headers, index, footer, padding and checks are verified on the host; every LZMA2 block body is a
// unit of the batched device launch.
#include <algorithm>
#include <map>
#include <vector>
#include "framing.h"

namespace swc {
namespace {

// A cursor with LittleEndianByteReader's contract: reads past the end latch `trap`.
struct Reader {
    const uint8_t* d;
    size_t n;
    int64_t off;
    bool trap;
    int64_t left() const { return (int64_t)n - off; }
    bool finished() const { return off >= (int64_t)n; }
    uint8_t u8() {
        if (off < 0 || off >= (int64_t)n) { trap = true; return 0; }
        return d[off++];
    }
    uint64_t le(int k) {
        if (off < 0 || left() < k) { trap = true; off = (int64_t)n; return 0; }
        uint64_t v = 0;
        for (int i = 0; i < k; i++) v |= (uint64_t)d[off + i] << (8 * i);
        off += k;
        return v;
    }
};

int multibyte(Reader& r, int64_t& v) {  // LittleEndianByteReader+XZ.swift:10-30
    int i = 1;
    int64_t result = r.u8();
    if (r.trap) return SWC_E_REF_TRAP;
    if (result <= 127) { v = result; return SWC_OK; }
    result &= 0x7F;
    for (;;) {
        const uint32_t b = r.u8();
        if (r.trap) return SWC_E_REF_TRAP;
        if (i >= 9 || b == 0) return SWC_E_XZ_MULTI_BYTE_INTEGER_ERROR;
        result += (int64_t)(b & 0x7F) << (7 * i);
        i++;
        if (!(b & 0x80)) break;
    }
    v = result;
    return SWC_OK;
}

}  // namespace

// Sum of the unpacked sizes announced by the chunk headers (LZMA2Decoder.swift:56-61,84-86) -- an exact
// output capacity for well-formed streams, a hint otherwise.
size_t lzma2_announced_size(const uint8_t* p, size_t n) {
    size_t off = 0, total = 0;
    while (off < n) {
        const uint32_t c = p[off];
        if (c == 0) break;
        if (c == 1 || c == 2) {
            if (n - off < 3) break;
            const size_t sz = ((size_t)p[off + 1] << 8) + p[off + 2] + 1;
            total += sz;
            off += 3 + sz;
        } else if (c >= 0x80) {
            if (n - off < 5) break;
            total += ((size_t)(c & 0x1F) << 16) + ((size_t)p[off + 1] << 8) + p[off + 2] + 1;
            const size_t comp = ((size_t)p[off + 3] << 8) + p[off + 4] + 1;
            off += 5 + (((c >> 5) & 3) >= 2 ? 1 : 0) + comp;
        } else {
            break;
        }
        if (total > ((size_t)1 << 34)) break;
    }
    return total;
}

namespace {

int run_lzma_unit(int codec, HostUnit& u) {
    int st = run_one(codec, u);
    if (st) return st;
    return SWC_OK;
}

void delta_decode(const std::vector<uint8_t>& in, int distance, std::vector<uint8_t>& out) {  // DeltaFilter.swift:11-33
    uint8_t delta[256] = {0};
    int pos = 0;
    out.resize(in.size());
    for (size_t i = 0; i < in.size(); i++) {
        uint8_t tmp = delta[(distance + pos) % 256];
        tmp = (uint8_t)(in[i] + tmp);
        delta[pos] = tmp;
        out[i] = tmp;
        pos = pos == 0 ? 255 : pos - 1;
    }
}

struct Filter { int id; int prop; };

// Blocks decoded ahead of the sequential walk (xz_predecode below): key = offset of the LZMA2 data in the archive.
struct Predecoded {
    int prop;
    std::vector<uint8_t> out;
    size_t in_consumed;
    int sum_kind = 0;        // swc_checksum kind of `sum` (1 CRC-32, 3 CRC-64), 0 = none: computed on the device behind the decode
    uint64_t sum = 0;
};
// what xz_block knows about the data it appended beyond the bytes: a checksum the device computed (a block whose only filter
// is LZMA2 and whose data were decoded ahead)
struct BlockSum { int kind = 0; uint64_t sum = 0; };
typedef std::map<int64_t, Predecoded> BlockCache;

// XZBlock.init (XZBlock.swift:18-97).  Appends the block's data to `out`.
int xz_block(uint32_t header_size_byte, Reader& r, int check_size, std::vector<uint8_t>& out, int64_t& unpadded_size,
             int64_t& uncomp_size, BlockCache* cache, BlockSum* bsum = nullptr) {
    int st;
    const int64_t header_start = r.off - 1;
    const int64_t real_header_size = ((int64_t)header_size_byte + 1) * 4;
    const uint32_t flags = r.u8();
    if (r.trap) return SWC_E_REF_TRAP;
    const int filters_count = (int)(flags & 0x03) + 1;
    if (flags & 0x3C) return SWC_E_XZ_WRONG_FIELD;                                // :27
    int64_t compressed_size = -1, uncompressed_size = -1;
    if (flags & 0x40) { if ((st = multibyte(r, compressed_size))) return st; }
    if (flags & 0x80) { if ((st = multibyte(r, uncompressed_size))) return st; }
    Filter filters[4];
    for (int i = 0; i < filters_count; i++) {
        int64_t id, ps;
        if ((st = multibyte(r, id))) return st;
        if ((uint64_t)id >= 0x4000000000000000ull) return SWC_E_XZ_WRONG_FILTER_ID;
        if (id == 0x21) {
            if ((st = multibyte(r, ps))) return st;
            if (ps != 1) return SWC_E_LZMA2_WRONG_DICTIONARY_SIZE;                // :47
            filters[i] = {0x21, r.u8()};
        } else if (id == 0x03) {
            if ((st = multibyte(r, ps))) return st;
            if (ps != 1) return SWC_E_XZ_WRONG_FIELD;                             // :55
            filters[i] = {0x03, (int)(uint8_t)(r.u8() + 1)};
        } else {
            return SWC_E_XZ_WRONG_FILTER_ID;
        }
        if (r.trap) return SWC_E_REF_TRAP;
    }
    while (r.off - header_start < real_header_size - 4) {                         // :64-68
        const uint8_t b = r.u8();
        if (r.trap) return SWC_E_REF_TRAP;
        if (b != 0) return SWC_E_XZ_WRONG_PADDING;
    }
    const uint32_t hcrc = (uint32_t)r.le(4);
    if (r.trap) return SWC_E_REF_TRAP;
    if (header_start < 0 || header_start + real_header_size - 4 > (int64_t)r.n) return SWC_E_REF_TRAP;
    if (swc_crc32(r.d + header_start, (size_t)(real_header_size - 4), 0) != hcrc) return SWC_E_XZ_WRONG_INFO_CRC;
    r.off = header_start + real_header_size;

    const int64_t data_start = r.off;
    // :78 filters.reversed().reduce(byteReader): the last filter reads the archive, earlier ones its output
    std::vector<uint8_t> cur;
    bool have_cur = false;
    for (int i = filters_count - 1; i >= 0; i--) {
        std::vector<uint8_t> next;
        BlockCache::iterator hit;
        if (filters[i].id == 0x21 && !have_cur && cache && (hit = cache->find(r.off)) != cache->end() && hit->second.prop == filters[i].prop) {
            // decoded ahead from exactly these bytes, cleanly: the same result the launch below would produce
            r.off += (int64_t)hit->second.in_consumed;
            next = std::move(hit->second.out);
            if (bsum && filters_count == 1) { bsum->kind = hit->second.sum_kind; bsum->sum = hit->second.sum; }
            cache->erase(hit);
            stat_add(2, 1);
        } else if (filters[i].id == 0x21) {
            HostUnit u;
            u.in = have_cur ? cur.data() : r.d + r.off;
            u.in_len = have_cur ? cur.size() : (size_t)(r.n - r.off);
            u.aux = filters[i].prop;
            u.cap_hint = lzma2_announced_size(u.in, u.in_len);
            if (u.cap_hint == 0) u.cap_hint = 16;
            if ((st = run_lzma_unit(SWC_CODEC_LZMA2, u))) return st;
            if (!have_cur) r.off += (int64_t)u.in_consumed;
            if (u.status) return u.status;
            next = std::move(u.out);
        } else {
            std::vector<uint8_t> src;
            if (have_cur) src = std::move(cur);
            else { src.assign(r.d + r.off, r.d + r.n); r.off = (int64_t)r.n; }   // Delta reads until its reader is finished
            delta_decode(src, filters[i].prop, next);
        }
        cur = std::move(next);
        have_cur = true;
    }
    if (!((compressed_size < 0 || compressed_size == r.off - data_start) &&
          (uncompressed_size < 0 || uncompressed_size == (int64_t)cur.size()))) return SWC_E_XZ_WRONG_DATA_SIZE;  // :80-82
    const int64_t unpadded = r.off - header_start;
    if (unpadded % 4 != 0) {
        for (int k = (int)(4 - unpadded % 4); k > 0; k--) {
            const uint8_t b = r.u8();
            if (r.trap) return SWC_E_REF_TRAP;
            if (b != 0) return SWC_E_XZ_WRONG_PADDING;
        }
    }
    out.insert(out.end(), cur.begin(), cur.end());
    unpadded_size = unpadded + check_size;
    uncomp_size = (int64_t)cur.size();
    return SWC_OK;
}

// processStream (XZArchive.swift:90-130)
int xz_stream(Reader& r, std::vector<uint8_t>& out, bool& check_error, BlockCache* cache) {
    static const uint8_t magic[6] = {0xFD, 0x37, 0x7A, 0x58, 0x5A, 0x00};
    check_error = false;
    if (r.left() < 12) return SWC_E_REF_TRAP;
    if (memcmp(r.d + r.off, magic, 6) != 0) return SWC_E_XZ_WRONG_MAGIC;         // XZStreamHeader.swift:35
    r.off += 6;
    const uint8_t f0 = r.u8(), f1 = r.u8();
    const uint32_t fcrc = (uint32_t)r.le(4);
    const uint8_t fb[2] = {f0, f1};
    if (swc_crc32(fb, 2, 0) != fcrc) return SWC_E_XZ_WRONG_INFO_CRC;
    if (!(f0 == 0 && (f1 & 0xF0) == 0)) return SWC_E_XZ_WRONG_FIELD;
    const int check_type = f1 & 0x0F;
    int check_size;
    switch (check_type) {
        case 0x00: check_size = 0; break;
        case 0x01: check_size = 4; break;
        case 0x04: check_size = 8; break;
        case 0x0A: check_size = 32; break;
        default: return SWC_E_XZ_WRONG_FIELD;
    }
    std::vector<std::pair<int64_t, int64_t>> infos;
    int64_t index_size = -1;
    int st;
    for (;;) {
        const uint32_t hs = r.u8();
        if (r.trap) return SWC_E_REF_TRAP;
        if (hs == 0) {                                                            // processIndex :132-167
            const int64_t index_start = r.off - 1;
            int64_t records;
            if ((st = multibyte(r, records))) return st;
            if (records != (int64_t)infos.size()) return SWC_E_XZ_WRONG_FIELD;
            for (auto& bi : infos) {
                int64_t a, b;
                if ((st = multibyte(r, a))) return st;
                if (a != bi.first) return SWC_E_XZ_WRONG_FIELD;
                if ((st = multibyte(r, b))) return st;
                if (b != bi.second) return SWC_E_XZ_WRONG_DATA_SIZE;
            }
            index_size = r.off - index_start;
            if (index_size % 4 != 0) {
                for (int k = (int)(4 - index_size % 4); k > 0; k--) {
                    const uint8_t b = r.u8();
                    if (r.trap) return SWC_E_REF_TRAP;
                    if (b != 0) return SWC_E_XZ_WRONG_PADDING;
                    index_size++;
                }
            }
            const uint32_t icrc = (uint32_t)r.le(4);
            if (r.trap) return SWC_E_REF_TRAP;
            if (swc_crc32(r.d + index_start, (size_t)index_size, 0) != icrc) return SWC_E_XZ_WRONG_INFO_CRC;
            index_size += 4;
            break;
        }
        int64_t unpadded, uncomp;
        const size_t bstart = out.size();
        BlockSum bsum;
        if ((st = xz_block(hs, r, check_size, out, unpadded, uncomp, cache, &bsum))) return st;
        const uint8_t* bd = out.data() + bstart;
        const size_t bl = out.size() - bstart;
        if (check_type == 0x01) {
            const uint32_t c = (uint32_t)r.le(4);
            if (r.trap) return SWC_E_REF_TRAP;
            if ((bsum.kind == 1 ? (uint32_t)bsum.sum : swc_crc32(bd, bl, 0)) != c) { check_error = true; return SWC_OK; }
        } else if (check_type == 0x04) {
            const uint64_t c = r.le(8);
            if (r.trap) return SWC_E_REF_TRAP;
            if ((bsum.kind == 3 ? bsum.sum : swc_crc64(bd, bl)) != c) { check_error = true; return SWC_OK; }
        } else if (check_type == 0x0A) {
            if (r.left() < 32) return SWC_E_REF_TRAP;
            uint8_t dg[32];
            swc_sha256(bd, bl, dg);
            const bool ok = memcmp(dg, r.d + r.off, 32) == 0;
            r.off += 32;
            if (!ok) { check_error = true; return SWC_OK; }
        }
        infos.emplace_back(unpadded, uncomp);
    }
    // processFooter :169-192
    const uint32_t footer_crc = (uint32_t)r.le(4);
    const int64_t backward = ((int64_t)r.le(4) + 1) * 4;
    const uint32_t fflags = (uint32_t)r.le(2);
    if (r.trap) return SWC_E_REF_TRAP;
    if (swc_crc32(r.d + r.off - 6, 6, 0) != footer_crc) return SWC_E_XZ_WRONG_INFO_CRC;
    if (backward != index_size) return SWC_E_XZ_WRONG_FIELD;
    if (!((fflags & 0xFF) == 0 && ((fflags & 0xF00) >> 8) == (uint32_t)check_type && (fflags & 0xF000) == 0)) return SWC_E_XZ_WRONG_FIELD;
    if (r.left() < 2) return SWC_E_REF_TRAP;
    if (!(r.d[r.off] == 0x59 && r.d[r.off + 1] == 0x5A)) return SWC_E_XZ_WRONG_MAGIC;
    r.off += 2;
    return SWC_OK;
}

int xz_padding(Reader& r) {                                                       // processPadding :194-218
    if (r.finished()) return SWC_OK;
    int64_t padding = 0;
    for (;;) {
        const uint8_t b = r.u8();
        if (r.trap) return SWC_E_REF_TRAP;
        if (b != 0) {
            if (padding % 4 != 0) return SWC_E_XZ_WRONG_PADDING;
            break;
        }
        if (r.finished()) {
            if (b != 0 || padding % 4 != 3) return SWC_E_XZ_WRONG_PADDING;
            return SWC_OK;
        }
        padding++;
    }
    r.off -= 1;
    return SWC_OK;
}

// Index-driven block discovery (XZArchive.swift:132-192 read backwards): an .xz stream ends with
// footer <- index, and the index lists every block's unpadded and uncompressed size, so all blocks of all streams of the
// archive can be located without decoding anything.  Every block whose only filter is LZMA2 is decoded in ONE batched
// launch from exactly the bytes the index assigns to it; a result is kept only if it decoded cleanly (then it cannot
// depend on the bytes behind the block), and the sequential walk below -- which alone decides what the archive means --
// picks it up when it arrives at the same offset with the same dictionary byte.  Anything inconsistent simply yields
// no cache entry.
struct Cand { int64_t data_off; size_t len; int prop; int64_t uncomp; int check_type; };
static void xz_candidates(const uint8_t* d, size_t n, std::vector<Cand>& cands) {
    int64_t end = (int64_t)n;
    for (int streams = 0; streams < 4096 && end >= 32; streams++) {
        while (end >= 4 && d[end - 1] == 0 && d[end - 2] == 0 && d[end - 3] == 0 && d[end - 4] == 0) end -= 4;   // stream padding
        if (end < 32 || d[end - 2] != 0x59 || d[end - 1] != 0x5A) break;                      // "YZ"
        const int64_t backward = ((int64_t)((uint32_t)d[end - 8] | (uint32_t)d[end - 7] << 8 | (uint32_t)d[end - 6] << 16 | (uint32_t)d[end - 5] << 24) + 1) * 4;
        const int check_type = d[end - 3] & 0x0F;
        const int check_size = check_type == 0 ? 0 : check_type == 1 ? 4 : check_type == 4 ? 8 : check_type == 10 ? 32 : -1;
        const int64_t index_start = end - 12 - backward;
        if (check_size < 0 || index_start < 12 || d[index_start] != 0) break;
        Reader r{d, (size_t)(end - 12), index_start + 1, false};
        int64_t records;
        if (multibyte(r, records) || records < 0 || records > (1 << 20)) break;
        std::vector<std::pair<int64_t, int64_t>> recs;
        bool ok = true;
        int64_t total = 0;
        for (int64_t k = 0; k < records && ok; k++) {
            int64_t a, b;
            ok = !multibyte(r, a) && !multibyte(r, b) && a > 0 && a < ((int64_t)1 << 40);
            if (ok) { recs.emplace_back(a, b); total += (a + 3) & ~(int64_t)3; }
        }
        const int64_t stream_start = index_start - total - 12;
        if (!ok || stream_start < 0) break;
        int64_t pos = stream_start + 12;
        for (auto& rc : recs) {
            const int64_t hsize = ((int64_t)d[pos] + 1) * 4;
            const int64_t comp = rc.first - hsize - check_size;
            // a header this code understands: one filter, LZMA2 with one property byte, optional size fields
            if (d[pos] != 0 && comp > 0 && pos + rc.first <= index_start && (d[pos + 1] & 0x3F) == 0) {
                Reader h{d, (size_t)(pos + hsize), pos + 2, false};
                int64_t tmp, id, ps;
                bool good = true;
                if (d[pos + 1] & 0x40) good = good && !multibyte(h, tmp);
                if (d[pos + 1] & 0x80) good = good && !multibyte(h, tmp);
                good = good && !multibyte(h, id) && id == 0x21 && !multibyte(h, ps) && ps == 1;
                const int prop = h.u8();
                if (good && !h.trap) cands.push_back({pos + hsize, (size_t)comp, prop, rc.second, check_type});
            }
            pos += (rc.first + 3) & ~(int64_t)3;
        }
        end = stream_start;
    }
}

void xz_predecode(const uint8_t* d, size_t n, BlockCache& cache) {
    std::vector<Cand> cands;
    xz_candidates(d, n, cands);
    if (cands.size() < 2) return;            // a single block gains nothing from being decoded ahead
    // the blocks' check, if the streams of the archive agree on CRC-32 or CRC-64: on the device, behind the decode (a launch
    // computes one kind; XZArchive.swift:106-121 on one host thread was as long as the launch itself)
    int kind = cands[0].check_type == 0x01 ? 1 : cands[0].check_type == 0x04 ? 3 : 0;
    for (const Cand& c : cands) if (c.check_type != cands[0].check_type) kind = 0;
    std::vector<HostUnit> units(cands.size());
    for (size_t k = 0; k < cands.size(); k++) {
        HostUnit& u = units[k];
        u.in = d + cands[k].data_off;
        u.in_len = cands[k].len;
        u.base = d;
        u.base_len = n;
        u.aux = cands[k].prop;
        u.cap_hint = std::max<size_t>(lzma2_announced_size(u.in, u.in_len), 16);
        u.sum_kind = kind;
    }
    if (run_units(SWC_CODEC_LZMA2, units) != SWC_OK) return;
    for (size_t k = 0; k < cands.size(); k++) {
        if (units[k].status != SWC_OK) continue;
        Predecoded p{cands[k].prop, std::move(units[k].out), units[k].in_consumed};
        if (units[k].sum_valid) { p.sum_kind = kind; p.sum = units[k].sum; }
        cache[cands[k].data_off] = std::move(p);
    }
}

int xz_run(const uint8_t* in, size_t in_len, std::vector<uint8_t>& all, std::vector<size_t>& sizes) {
    Trace tr;
    BlockCache cache_store;
    xz_predecode(in, in_len, cache_store);
    tr.mark("xz: blocks decoded ahead", in_len);
    BlockCache* cache = &cache_store;
    size_t ahead = 0;
    for (const auto& kv : cache_store) ahead += kv.second.out.size();
    all.reserve(ahead + 64);                 // (the result grows block by block: without this, by doubling and copying)
    Reader r{in, in_len, 0, false};
    while (!r.finished()) {
        if (r.left() < 32) { all.clear(); sizes.clear(); return SWC_E_XZ_WRONG_MAGIC; }  // :37
        bool check_error;
        const size_t start = all.size();
        int st = xz_stream(r, all, check_error, cache);
        if (st) { all.clear(); sizes.clear(); return st; }
        sizes.push_back(all.size() - start);
        if (check_error) return SWC_E_XZ_WRONG_CHECK;                             // :44 carries the result so far
        st = xz_padding(r);
        if (st) { all.clear(); sizes.clear(); return st; }
    }
    tr.mark("xz: the walk of the streams", all.size());
    return SWC_OK;
}

}  // namespace

// The LZMA2-only blocks the index of every stream of the archive lists: offset / length of the LZMA2 data, uncompressed
// size, aux = dictionary-size byte.  (Blocks with other filter chains are left out.)
void xz_block_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out) {
    std::vector<Cand> cands;
    xz_candidates(in, in_len, cands);
    for (const Cand& c : cands) out.push_back({(uint64_t)c.data_off, (uint64_t)c.len, (uint64_t)c.uncomp, (uint32_t)c.prop});
    std::sort(out.begin(), out.end(), [](const BlockRef64& a, const BlockRef64& b) { return a.offset < b.offset; });
}

// The chunk walk of LZMA2Decoder.decode() / dispatch() (reference Sources/LZMA2/LZMA2Decoder.swift:36-74) without the
// LZMA decode: `in` = the dictionary-size byte followed by the chunks (what LZMA2.decompress(data:) takes,
// LZMA2.swift:26-33).  One ref per chunk: offset of its control byte, comp_len = header + payload, uncomp_len,
// aux = control byte; flags bit 0 = the chunk resets the dictionary (control 1, or reset 3: an independently decodable
// run of chunks starts here), bit 1 = it carries a properties byte (reset 2, 3).  Stops at the end marker (control 0),
// which is not listed.  Errors as the reference raises them: wrongControlByte for 3...0x7F; a chunk that runs past the
// buffer (the reference would read out of range) is SWC_E_REF_TRAP.  Refs found before the error stay in `out`.
int lzma2_chunk_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out) {
    if (in_len < 1) return SWC_E_REF_TRAP;
    size_t p = 1;
    for (;;) {
        if (p >= in_len) return SWC_E_REF_TRAP;
        const uint8_t control = in[p];
        if (control == 0) return SWC_OK;
        if (control == 1 || control == 2) {                   // decodeUncompressed(): 2 size bytes + data (:76-84)
            if (p + 3 > in_len) return SWC_E_REF_TRAP;
            const size_t size = ((size_t)in[p + 1] << 8) + in[p + 2] + 1;
            if (size > in_len - (p + 3)) return SWC_E_REF_TRAP;
            out.push_back(BlockRef64{p, 3 + size, size, control, control == 1 ? 1u : 0u});
            p += 3 + size;
            continue;
        }
        if (control < 0x80) return SWC_E_LZMA2_WRONG_CONTROL_BYTE;                                   // :47-48
        const uint32_t reset = (control & 0x60u) >> 5;                                               // :60
        const size_t hdr = 5 + (reset >= 2 ? 1 : 0);
        if (p + hdr > in_len) return SWC_E_REF_TRAP;
        const size_t unpack = (((size_t)(control & 0x1Fu)) << 16) + ((size_t)in[p + 1] << 8) + in[p + 2] + 1;   // :61-62
        const size_t comp = ((size_t)in[p + 3] << 8) + in[p + 4] + 1;                                // :63
        if (comp > in_len - (p + hdr)) return SWC_E_REF_TRAP;
        out.push_back(BlockRef64{p, hdr + comp, unpack, control, (reset == 3 ? 1u : 0u) | (reset >= 2 ? 2u : 0u)});
        p += hdr + comp;
    }
}

}  // namespace swc

using namespace swc;

extern "C" {

int swc_lzma_decompress(const uint8_t* in, size_t in_len, int lc, int lp, int pb, int64_t dict_size,
                        int64_t uncompressed_size, uint8_t** out, size_t* out_len, size_t* in_consumed) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    HostUnit u;
    u.in = in; u.in_len = in_len;
    if (lc < 0 || lc > 255 || lp < 0 || lp > 255 || pb < 0 || pb > 255 || dict_size < 0) { give_empty(out, out_len); return SWC_E_REF_TRAP; }
    u.aux = lc | (lp << 8) | (pb << 16);
    u.extra = uncompressed_size < 0 ? ~0ull : (uint64_t)uncompressed_size;
    u.dict_value = (uint64_t)dict_size;
    if (uncompressed_size >= 0) u.cap_hint = (size_t)uncompressed_size + 16;      // +16: a stream may overrun its declared size before the error is detected
    int st = run_one(SWC_CODEC_LZMA, u);
    if (st) { give_empty(out, out_len); return st; }
    if (in_consumed) *in_consumed = u.in_consumed;
    if (u.status) { give_empty(out, out_len); return u.status; }                  // LZMAError cases carry no data
    give(u.out, out, out_len);
    return SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

int swc_lzma_alone_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    if (in_len < 13) { give_empty(out, out_len); return SWC_E_LZMA_WRONG_PROPERTIES; }   // LZMA.swift:27
    const uint32_t b = in[0];
    if (b >= 225) { give_empty(out, out_len); return SWC_E_LZMA_WRONG_PROPERTIES; }      // LZMAProperties.swift:51
    const int lc = b % 9, pb = (b / 9) / 5, lp = (b / 9) % 5;
    const int64_t dict = (int64_t)((uint32_t)in[1] | (uint32_t)in[2] << 8 | (uint32_t)in[3] << 16 | (uint32_t)in[4] << 24);
    uint64_t us = 0;
    for (int i = 0; i < 8; i++) us |= (uint64_t)in[5 + i] << (8 * i);
    size_t consumed;
    return swc_lzma_decompress(in + 13, in_len - 13, lc, lp, pb, dict, (int64_t)us, out, out_len, &consumed);
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

int swc_lzma2_decompress(const uint8_t* in, size_t in_len, uint8_t dict_byte, uint8_t** out, size_t* out_len, size_t* in_consumed) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    HostUnit u;
    u.in = in; u.in_len = in_len;
    u.aux = dict_byte;
    u.cap_hint = lzma2_announced_size(in, in_len);
    if (u.cap_hint == 0) u.cap_hint = 16;
    int st = run_one(SWC_CODEC_LZMA2, u);
    if (st) { give_empty(out, out_len); return st; }
    if (in_consumed) *in_consumed = u.in_consumed;
    if (u.status) { give_empty(out, out_len); return u.status; }
    give(u.out, out, out_len);
    return SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

int swc_lzma2_decompress_data(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    if (in_len < 1) { give_empty(out, out_len); return SWC_E_LZMA_RANGE_DECODER_INIT_ERROR; }  // LZMA2.swift:27
    size_t consumed;
    return swc_lzma2_decompress(in + 1, in_len - 1, in[0], out, out_len, &consumed);
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

int swc_xz_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    std::vector<uint8_t> all;
    std::vector<size_t> sizes;
    int st = xz_run(in, in_len, all, sizes);
    give(all, out, out_len);
    return st;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

int swc_xz_split_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t** sizes, size_t* n_streams) try {
    if (!out || !out_len || !sizes || !n_streams || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    std::vector<uint8_t> all;
    std::vector<size_t> sz;
    int st = xz_run(in, in_len, all, sz);
    give(all, out, out_len);
    *sizes = give_sizes(sz);
    *n_streams = sz.size();
    return st;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    if (sizes) *sizes = nullptr;
    if (n_streams) *n_streams = 0;
    return SWC_E_DEVICE;
}

}  // extern "C"
