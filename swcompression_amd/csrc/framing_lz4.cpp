// framing_lz4.cpp -- host side of LZ4: frame / legacy-frame / skippable-frame parsing and block
// discovery feeding ONE batched launch per frame (independent blocks) -- the reference decodes the
// blocks one after another on the CPU.
//   LZ4.decompress(data:dictionary:dictionaryID:)   reference Sources/LZ4/LZ4.swift:73-91
//   LZ4.multiDecompress                              :116-146
//   process(skippableFrame:) :148-155   process(legacyFrame:) :160-186   process(frame:_:_:) :188-330
// Errors are reported in STREAM ORDER exactly as the sequential reference would meet them: framing
// checks of block k only count once every block before k has decoded cleanly.
#include <vector>
#include "framing.h"

namespace swc {

namespace {

inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

struct BlockRef {
    size_t off, len;
    bool compressed;
};

// Units for the compressed blocks of `blocks` (all independent of each other), in order.
void independent_units(const uint8_t* base, const std::vector<BlockRef>& blocks, const uint8_t* dict, size_t dict_len,
                       bool have_dict, size_t max_block, std::vector<HostUnit>& units) {
    for (const BlockRef& b : blocks) {
        if (!b.compressed) continue;
        HostUnit u;
        u.in = base + b.off;
        u.in_len = b.len;
        u.cap_hint = max_block;  // a block normally decodes to <= the frame's maximum block size (not enforced by the reference)
        if (have_dict) { u.dict = dict ? dict : reinterpret_cast<const uint8_t*>(""); u.dict_len = dict_len; }
        units.push_back(std::move(u));
    }
}
// Appends the blocks to `out` in stream order; `units` = the decoded units of independent_units().  Returns the
// first decode error in stream order.
int independent_append(const uint8_t* base, const std::vector<BlockRef>& blocks, const HostUnit* units, std::vector<uint8_t>& out) {
    size_t k = 0;
    for (const BlockRef& b : blocks) {
        if (b.compressed) {
            const HostUnit& u = units[k++];
            if (u.status) return u.status;
            out.insert(out.end(), u.out.begin(), u.out.end());
        } else {
            out.insert(out.end(), base + b.off, base + b.off + b.len);
        }
    }
    return SWC_OK;
}
// Decode `blocks` on the device in one batch and append to `out` in order.
// Returns the first decode error in stream order, or SWC_OK / SWC_E_DEVICE.
int decode_independent(const uint8_t* base, const std::vector<BlockRef>& blocks, const uint8_t* dict, size_t dict_len,
                       bool have_dict, size_t max_block, std::vector<uint8_t>& out) {
    std::vector<HostUnit> units;
    units.reserve(blocks.size());
    independent_units(base, blocks, dict, dict_len, have_dict, max_block, units);
    if (!units.empty()) {
        int st = run_units(SWC_CODEC_LZ4_BLOCK, units);
        if (st) return st;
    }
    return independent_append(base, blocks, units.data(), out);
}

// process(legacyFrame:) :160-186; p points just after the magic.
int legacy_frame(const uint8_t* p, size_t n, std::vector<uint8_t>& out, size_t& adv) {
    std::vector<BlockRef> blocks;
    size_t off = 0;
    int framing = SWC_OK;
    while (off < n) {
        if (n - off < 4) { framing = SWC_E_DATA_TRUNCATED; break; }               // :165
        uint32_t raw = le32(p + off);
        off += 4;
        if (raw == 0x184D2204u || raw == 0x184C2102u || (raw >= 0x184D2A50u && raw <= 0x184D2A5Fu)) { off -= 4; break; }  // :168-171
        if (n - off < raw) { framing = SWC_E_DATA_TRUNCATED; break; }             // :177
        blocks.push_back({off, raw, true});
        off += raw;
    }
    int st = decode_independent(p, blocks, nullptr, 0, false, 8u << 20, out);
    if (st) return st;
    if (framing) return framing;
    adv = off;
    return SWC_OK;
}

// process(frame:_:_:) :188-330 in three steps: header + block discovery (host only), block decode, trailer checks.
struct FrameInfo {
    bool independent = false, content_size_present = false, content_checksum = false;
    size_t max_block = 0;
    uint64_t content_size = 0;
    std::vector<BlockRef> blocks;
    int framing = SWC_OK;   // framing error met during discovery: counts only once every block before it decoded cleanly
    size_t off = 0;         // just past the last block / EndMark
};

// p points just after the magic.  Returns the header error, or SWC_OK with `fi` filled.
int frame_parse(const uint8_t* p, size_t n, bool have_dict, int64_t ext_dict_id, FrameInfo& fi) {
    if (n < 7) return SWC_E_DATA_TRUNCATED;                                       // :191
    size_t off = 0;
    const uint32_t flg = p[off++];
    if (!(((flg & 0xC0) >> 6) == 1 && (flg & 0x02) == 0)) return SWC_E_DATA_CORRUPTED;  // :198
    const bool block_checksum = flg & 0x10, dict_id_present = flg & 0x01;
    fi.independent = flg & 0x20;
    fi.content_size_present = flg & 0x08;
    fi.content_checksum = flg & 0x04;
    switch (p[off++]) {                                                           // :216-228
        case 0x40: fi.max_block = 64u << 10; break;
        case 0x50: fi.max_block = 256u << 10; break;
        case 0x60: fi.max_block = 1u << 20; break;
        case 0x70: fi.max_block = 4u << 20; break;
        default: return SWC_E_DATA_CORRUPTED;
    }
    if (fi.content_size_present) {
        if (n - off < 13) return SWC_E_DATA_TRUNCATED;                            // :234
        for (int i = 0; i < 8; i++) fi.content_size |= (uint64_t)p[off + i] << (8 * i);
        off += 8;
        if (fi.content_size > (uint64_t)INT64_MAX) return SWC_E_DATA_UNSUPPORTED_FEATURE;  // :240
    }
    int64_t dict_id = -1;
    if (dict_id_present) {
        if (!have_dict) return SWC_E_DATA_CORRUPTED;                              // :250
        if (n - off < 9) return SWC_E_DATA_TRUNCATED;                             // :254
        dict_id = le32(p + off);
        off += 4;
    }
    if (ext_dict_id >= 0 && dict_id >= 0 && ext_dict_id != dict_id) return SWC_E_DATA_CORRUPTED;  // :266-270
    if ((uint8_t)((swc_xxh32(p, off, 0) >> 8) & 0xFF) != p[off]) return SWC_E_DATA_CORRUPTED;     // :272-275
    off++;

    // block discovery
    for (;;) {
        if (n - off < 4) { fi.framing = SWC_E_DATA_TRUNCATED; break; }            // :279
        const uint32_t mark = le32(p + off);
        off += 4;
        if (mark == 0) break;                                                     // EndMark :284
        const size_t bs = mark & 0x7FFFFFFFu;
        if (bs > fi.max_block) { fi.framing = SWC_E_DATA_CORRUPTED; break; }      // :292
        if (n - off < bs + (block_checksum ? 4 : 0) + 4) { fi.framing = SWC_E_DATA_TRUNCATED; break; }  // :295
        const size_t boff = off;
        off += bs;
        if (block_checksum) {
            const uint32_t c = le32(p + off);
            off += 4;
            if (swc_xxh32(p + boff, bs, 0) != c) { fi.framing = SWC_E_DATA_CORRUPTED; break; }  // :300
        }
        fi.blocks.push_back({boff, bs, (mark & 0x80000000u) == 0});
    }
    fi.off = off;
    return SWC_OK;
}

// After the blocks were appended to out[start..): deferred framing error, content size and content checksum.
int frame_tail(const uint8_t* p, size_t n, const FrameInfo& fi, const std::vector<uint8_t>& out, size_t start, size_t& adv) {
    size_t off = fi.off;
    if (fi.framing) return fi.framing;
    if (fi.content_size_present && (uint64_t)(out.size() - start) != fi.content_size) return SWC_E_DATA_CORRUPTED;  // :320
    if (fi.content_checksum) {
        if (n - off < 4) return SWC_E_DATA_TRUNCATED;                             // :324
        const uint32_t c = le32(p + off);
        off += 4;
        adv = off;
        if (swc_xxh32(out.data() + start, out.size() - start, 0) != c) return SWC_E_DATA_CHECKSUM_MISMATCH;  // :326
    }
    adv = off;
    return SWC_OK;
}

int frame(const uint8_t* p, size_t n, const uint8_t* dict, size_t dict_len, bool have_dict, int64_t ext_dict_id,
          std::vector<uint8_t>& out, size_t& adv) {
    FrameInfo fi;
    int st = frame_parse(p, n, have_dict, ext_dict_id, fi);
    if (st) return st;
    const size_t start = out.size();
    if (fi.independent) {
        st = decode_independent(p, fi.blocks, dict, dict_len, have_dict, fi.max_block, out);  // :305
        if (st) return st;
    } else {
        // Dependent blocks: block k references the last 64 KiB produced so far (:306-313) -- a serial chain.
        for (const BlockRef& b : fi.blocks) {
            if (!b.compressed) { out.insert(out.end(), p + b.off, p + b.off + b.len); continue; }
            HostUnit u;
            u.in = p + b.off;
            u.in_len = b.len;
            u.cap_hint = fi.max_block;
            const size_t produced = out.size() - start;
            std::vector<uint8_t> window;
            if (produced == 0 && have_dict) {
                const size_t dl = dict_len > 65536 ? 65536 : dict_len;
                u.dict = dict_len ? dict + (dict_len - dl) : reinterpret_cast<const uint8_t*>("");
                u.dict_len = dl;
            } else {
                const size_t dl = produced > 65536 ? 65536 : produced;
                window.assign(out.end() - dl, out.end());
                u.dict = dl ? window.data() : reinterpret_cast<const uint8_t*>("");
                u.dict_len = dl;
            }
            st = run_one(SWC_CODEC_LZ4_BLOCK, u);
            if (st) return st;
            if (u.status) return u.status;
            out.insert(out.end(), u.out.begin(), u.out.end());
        }
    }
    return frame_tail(p, n, fi, out, start, adv);
}

int skippable(const uint8_t* p, size_t n, size_t& adv) {                          // :148-155
    if (n < 4) return SWC_E_DATA_TRUNCATED;
    const size_t size = le32(p);
    if (n < size + 4) return SWC_E_DATA_TRUNCATED;
    adv = size + 4;
    return SWC_OK;
}

}  // namespace

// ---- many-archive batching (swc_unarchive_many): an archive that opens with a standard frame of independent blocks
// contributes its blocks to a shared launch; everything else (skippable / legacy / dependent frames) is not batchable.
struct Lz4Plan::Impl { FrameInfo fi; };
Lz4Plan::Lz4Plan() : impl(new Impl) {}
Lz4Plan::~Lz4Plan() { delete impl; }
bool lz4_plan_prepare(const uint8_t* in, size_t n, Lz4Plan& plan, std::vector<HostUnit>& units) {
    if (n < 4 || le32(in) != 0x184D2204u) return false;
    plan.early_status = frame_parse(in + 4, n - 4, false, -1, plan.impl->fi);
    if (plan.early_status) return true;                       // header error: final, no units
    if (!plan.impl->fi.independent) return false;
    plan.first_unit = units.size();
    independent_units(in + 4, plan.impl->fi.blocks, nullptr, 0, false, plan.impl->fi.max_block, units);
    return true;
}
int lz4_plan_finish(const uint8_t* in, size_t n, const Lz4Plan& plan, const std::vector<HostUnit>& units, std::vector<uint8_t>& res) {
    if (plan.early_status) return plan.early_status;
    int st = independent_append(in + 4, plan.impl->fi.blocks, units.data() + plan.first_unit, res);
    if (st) { res.clear(); return st; }
    size_t adv = 0;
    st = frame_tail(in + 4, n - 4, plan.impl->fi, res, 0, adv);
    if (st != SWC_OK && st != SWC_E_DATA_CHECKSUM_MISMATCH) res.clear();          // only checksumMismatch carries data
    return st;
}

// The blocks of the frame `in` opens with (offsets relative to `in`); aux = 1 for stored blocks.  False: not a standard frame.
bool lz4_frame_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out) {
    if (in_len < 4 || le32(in) != 0x184D2204u) return false;
    FrameInfo fi;
    if (frame_parse(in + 4, in_len - 4, false, -1, fi) != SWC_OK) return false;
    for (const BlockRef& b : fi.blocks) out.push_back({(uint64_t)b.off + 4, (uint64_t)b.len, 0, b.compressed ? 0u : 1u});
    return true;
}

}  // namespace swc

using namespace swc;

extern "C" {

int swc_lz4_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len, int64_t dict_id,
                       uint8_t** out, size_t* out_len, size_t* in_consumed) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    std::vector<uint8_t> res;
    bool have_dict = dict != nullptr;
    size_t pos = 0, adv = 0;
    int st;
    for (;;) {
        if (in_len - pos < 4) { st = SWC_E_DATA_TRUNCATED; break; }               // :75
        const uint32_t magic = le32(in + pos);
        pos += 4;
        if (magic == 0x184D2204u) {
            st = frame(in + pos, in_len - pos, dict, dict_len, have_dict, dict_id, res, adv);
            if (st == SWC_OK || st == SWC_E_DATA_CHECKSUM_MISMATCH) pos += adv;
            break;
        } else if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {
            st = skippable(in + pos, in_len - pos, adv);
            if (st) break;
            pos += adv;
            have_dict = false; dict = nullptr; dict_len = 0; dict_id = -1;        // :85 recursion drops the dictionary
        } else if (magic == 0x184C2102u) {
            st = legacy_frame(in + pos, in_len - pos, res, adv);
            if (st == SWC_OK) pos += adv;
            break;
        } else {
            st = SWC_E_DATA_CORRUPTED;
            break;
        }
    }
    if (st != SWC_OK && st != SWC_E_DATA_CHECKSUM_MISMATCH) res.clear();          // only checksumMismatch carries data
    if (in_consumed) *in_consumed = pos;
    give(res, out, out_len);
    return st;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

// LZ4.compress(data:independentBlocks:...) LZ4+Compress.swift:47-155: the frame as the reference writes it -- magic, FLG, BD,
// content size, dictionary ID, header checksum (:55-95), then per block the size word, the block (stored when compression
// does not shrink it, :119-127), its checksum, the end mark and the content checksum (:142-151).  The blocks are compressed
// on the device: independent blocks all in ONE launch; dependent blocks one launch too -- what a block may reference
// (the previous block's last 64 KiB, the dictionary in front of the first, :109-116) is uncompressed INPUT, known up front.
int swc_lz4_compress(const uint8_t* data, size_t len, int independent_blocks, int block_checksums, int content_checksum,
                     int content_size, size_t block_size, const uint8_t* dict, size_t dict_len, int64_t dict_id,
                     uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (len && !data) || (dict_len && !dict)) return SWC_E_INVALID_ARGUMENT;
    if (block_size == 0 || block_size > ((size_t)4 << 20)) { give_empty(out, out_len); return SWC_E_INVALID_ARGUMENT; }   // :50 precondition
    std::vector<uint8_t> res;
    const uint8_t magic[4] = {0x04, 0x22, 0x4D, 0x18};
    res.insert(res.end(), magic, magic + 4);
    res.push_back((uint8_t)(0x40 | (independent_blocks ? 0x20 : 0) | (block_checksums ? 0x10 : 0) | (content_size ? 0x08 : 0) |
                            (content_checksum ? 0x04 : 0) | (dict_id >= 0 ? 0x01 : 0)));                                        // :58-63
    res.push_back(block_size <= (64u << 10) ? 0x40 : block_size <= (256u << 10) ? 0x50 : block_size <= (1u << 20) ? 0x60 : 0x70);   // :66-76
    if (content_size) for (int i = 0; i < 8; i++) res.push_back((uint8_t)((uint64_t)len >> (8 * i)));                          // :78-83
    if (dict_id >= 0) for (int i = 0; i < 4; i++) res.push_back((uint8_t)((uint32_t)dict_id >> (8 * i)));                      // :85-89
    res.push_back((uint8_t)((swc_xxh32(res.data() + 4, res.size() - 4, 0) >> 8) & 0xFF));                                       // :92-93
    // units: the kernel wants prefix ++ block in one piece (the prefix: at most the last 64 KiB of the dictionary / previous block).
    // A dependent block's prefix is the input right in front of it: the unit addresses `data` in place, and the units share
    // ONE staged copy of it (HostUnit::base).  Only a DICTIONARY prefix has to be joined with its block on the host; those
    // units go in rounds of at most kJoinBudget bytes (small independent blocks with a dictionary: 64 KiB of prefix each).
    const size_t nblk = (len + block_size - 1) / block_size;
    constexpr size_t kJoinBudget = (size_t)256 << 20;
    std::vector<std::vector<uint8_t>> comp(nblk);
    const uint8_t* d0 = dict ? dict + (dict_len > 65536 ? dict_len - 65536 : 0) : nullptr;                                      // :98-99
    const size_t d0n = dict ? std::min<size_t>(dict_len, 65536) : 0;
    for (size_t b0 = 0; b0 < nblk;) {
        std::vector<HostUnit> units;
        std::vector<std::vector<uint8_t>> joined;
        size_t b1 = b0, joined_bytes = 0;
        for (; b1 < nblk; b1++) {
            const size_t at = b1 * block_size, n = std::min(block_size, len - at);
            const bool in_place = !independent_blocks && b1 > 0;                   // :112-116 the previous block's data, its last 64 KiB
            const size_t pren = in_place ? std::min(block_size, (size_t)65536) : d0n;
            if (!in_place && pren && b1 > b0 && joined_bytes + pren + n > kJoinBudget) break;
            units.emplace_back();
            HostUnit& u = units.back();
            if (in_place) { u.in = data + at - pren; u.in_len = pren + n; u.base = data; u.base_len = len; }
            else if (pren == 0) { u.in = data + at; u.in_len = n; u.base = data; u.base_len = len; }
            else {
                joined.emplace_back(pren + n);
                memcpy(joined.back().data(), d0, pren);
                memcpy(joined.back().data() + pren, data + at, n);
                joined_bytes += pren + n;
                u.in_len = pren + n;      // (u.in below: the vector of vectors may still grow)
            }
            u.extra = pren;
            u.cap_hint = n + n / 255 + 16;
            u.cap_exact = true;
        }
        for (size_t k = 0, jn = 0; k < units.size(); k++) if (!units[k].in) units[k].in = joined[jn++].data();
        const int st = run_units(SWC_CODEC_LZ4_COMPRESS, units);
        if (st) { give_empty(out, out_len); return st; }
        for (size_t k = 0; k < units.size(); k++) {
            if (units[k].status != SWC_OK) { give_empty(out, out_len); return SWC_E_DEVICE; }
            comp[b0 + k] = std::move(units[k].out);
        }
        b0 = b1;
    }
    auto put32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) res.push_back((uint8_t)(v >> (8 * i))); };
    for (size_t b = 0; b < nblk; b++) {
        const size_t at = b * block_size, n = std::min(block_size, len - at);
        const std::vector<uint8_t>& c = comp[b];
        if (c.size() > n) {   // :119 not compressible: stored
            put32(0x80000000u | (uint32_t)n);
            res.insert(res.end(), data + at, data + at + n);
            if (block_checksums) put32(swc_xxh32(data + at, n, 0));
        } else {
            put32((uint32_t)c.size());
            res.insert(res.end(), c.begin(), c.end());
            if (block_checksums) put32(swc_xxh32(c.data(), c.size(), 0));
        }
    }
    put32(0);                                                   // :143 EndMark
    if (content_checksum) put32(swc_xxh32(data, len, 0));        // :146-151
    give(res, out, out_len);
    return SWC_OK;
} catch (...) {   // std::bad_alloc: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

// All frames of a multi-frame buffer in one launch: block sizes are in the block headers, so the frames (standard frames
// with independent blocks, skippable frames) can be walked without decoding anything.  Used only if every frame then passes
// its own checks; anything else -- dependent blocks, legacy frames, a dictionary, any error -- goes to the sequential loop
// below, which alone defines errors and partial results (LZ4.swift:116-146).
static bool lz4_multi_batched(const uint8_t* in, size_t in_len, std::vector<uint8_t>& all, std::vector<size_t>& sz) {
    struct Fr { size_t at; FrameInfo fi; size_t first_unit; };
    std::vector<Fr> frames;
    std::vector<HostUnit> units;
    size_t pos = 0;
    while (pos < in_len) {
        if (in_len - pos < 4) return false;
        const uint32_t magic = le32(in + pos);
        if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {
            size_t adv = 0;
            if (skippable(in + pos + 4, in_len - pos - 4, adv) != SWC_OK) return false;
            pos += 4 + adv;
            continue;
        }
        if (magic != 0x184D2204u) return false;
        frames.emplace_back();
        Fr& f = frames.back();
        f.at = pos + 4;
        if (frame_parse(in + f.at, in_len - f.at, false, -1, f.fi) != SWC_OK || !f.fi.independent || f.fi.framing != SWC_OK) return false;
        f.first_unit = units.size();
        independent_units(in + f.at, f.fi.blocks, nullptr, 0, false, f.fi.max_block, units);
        pos = f.at + f.fi.off + (f.fi.content_checksum ? 4 : 0);
        if (pos > in_len) return false;
    }
    if (frames.size() < 2) return false;
    if (!units.empty() && run_units(SWC_CODEC_LZ4_BLOCK, units) != SWC_OK) return false;
    for (const Fr& f : frames) {
        const size_t start = all.size();
        if (independent_append(in + f.at, f.fi.blocks, units.data() + f.first_unit, all) != SWC_OK) return false;
        size_t adv = 0;
        std::vector<uint8_t> one(all.begin() + (std::ptrdiff_t)start, all.end());
        if (frame_tail(in + f.at, in_len - f.at, f.fi, one, 0, adv) != SWC_OK) return false;
        sz.push_back(all.size() - start);
    }
    return true;
}

int swc_lz4_multi_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len, int64_t dict_id,
                             uint8_t** out, size_t* out_len, size_t** sizes, size_t* n_frames) try {
    if (!out || !out_len || !sizes || !n_frames || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    if (dict == nullptr) {
        std::vector<uint8_t> fast;
        std::vector<size_t> fsz;
        if (lz4_multi_batched(in, in_len, fast, fsz)) {
            give(fast, out, out_len);
            *sizes = give_sizes(fsz);
            *n_frames = fsz.size();
            return SWC_OK;
        }
    }
    std::vector<uint8_t> all;
    std::vector<size_t> sz;
    const bool have_dict = dict != nullptr;
    size_t pos = 0;
    int st = SWC_OK;
    do {
        if (pos + 4 > in_len) { st = SWC_E_DATA_TRUNCATED; break; }               // :123
        const uint32_t magic = le32(in + pos);
        pos += 4;
        size_t adv = 0;
        std::vector<uint8_t> one;
        bool produced = false;
        if (magic == 0x184D2204u) { st = frame(in + pos, in_len - pos, dict, dict_len, have_dict, dict_id, one, adv); produced = true; }
        else if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) st = skippable(in + pos, in_len - pos, adv);
        else if (magic == 0x184C2102u) { st = legacy_frame(in + pos, in_len - pos, one, adv); produced = true; }
        else st = SWC_E_DATA_CORRUPTED;
        if (st == SWC_E_DATA_CHECKSUM_MISMATCH) {                                 // :326 carries [out] of the failing frame only
            all = std::move(one);
            sz.assign(1, all.size());
            break;
        }
        if (st) { all.clear(); sz.clear(); break; }
        pos += adv;
        if (produced) { sz.push_back(one.size()); all.insert(all.end(), one.begin(), one.end()); }
    } while (pos < in_len);
    give(all, out, out_len);
    *sizes = give_sizes(sz);
    *n_frames = sz.size();
    return st;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    if (sizes) *sizes = nullptr;
    if (n_frames) *n_frames = 0;
    return SWC_E_DEVICE;
}

}  // extern "C"
