// framing_bzip2.cpp -- host side of BZip2: stream header, block discovery, CRC chaining.
//   BZip2.decompress(data:) / decompress(_ bitReader:)   reference Sources/BZip2/BZip2.swift:22-26, 50-95
//   BZip2.multiDecompress(data:)                          :40-48
//   BlockSize.init?                                       Sources/BZip2/BZip2+BlockSize.swift:29-52
// bzip2 blocks start at arbitrary BIT offsets and carry no length, so the sequential reference only
// learns where block k+1 starts after decoding block k.  Here the host scans the input for every
// occurrence of the 48-bit block magic, hands ALL candidates to one batched device launch, and then
// replays the reference's sequential walk over the results: a candidate is used only if the walk
// arrives exactly at its bit offset, so false positives inside block data are simply never visited
// and the outcome (bytes, error, error order) is identical to the sequential decoder.
#include <algorithm>
#include <map>
#include <thread>
#include <system_error>
#include <vector>
#include "framing.h"

namespace swc {
namespace {

constexpr uint64_t kBlockMagic = 0x314159265359ull;
constexpr uint64_t kEosMagic = 0x177245385090ull;

inline uint64_t read_bits(const uint8_t* d, uint64_t bit, int count) {  // MSB-first, count <= 56, caller checked bounds
    uint64_t v = 0;
    for (int i = 0; i < count; i++) {
        const uint64_t p = bit + i;
        v = (v << 1) | ((d[p >> 3] >> (7 - (p & 7))) & 1u);
    }
    return v;
}

// every bit offset (>= from_bit) at which the 48-bit block magic occurs, in increasing order.
// Discovery is the host's share of a bzip2 decode and must not be slower than the device: per BYTE position one big-endian
// 64-bit window is compared against the magic at its eight bit alignments (48 + 7 bits fit), and large inputs are cut into
// ranges scanned by several threads (a magic is found by the range that holds its first bit).
void scan_range(const uint8_t* d, size_t n, size_t lo, size_t hi, uint64_t from_bit, std::vector<uint64_t>& out) {
    for (size_t i = lo; i < hi; i++) {
        if (i + 6 > n) break;                        // fewer than 48 bits left
        uint64_t v = 0;
        const size_t take = n - i < 8 ? n - i : 8;
        for (size_t k = 0; k < take; k++) v |= (uint64_t)d[i + k] << (56 - 8 * k);
        for (int s = 0; s < 8; s++) {
            if (((v >> (16 - s)) & 0xFFFFFFFFFFFFull) != kBlockMagic) continue;
            if (s > 0 && i + 7 > n) continue;        // the last bits would lie past the end
            const uint64_t start = (uint64_t)i * 8 + (uint64_t)s;
            if (start >= from_bit) out.push_back(start);
        }
    }
}
void scan_block_magics(const uint8_t* d, size_t n, uint64_t from_bit, std::vector<uint64_t>& out) {
    if (n < 6) return;
    const size_t kMinPerThread = (size_t)4 << 20;
    size_t threads = n / kMinPerThread;
    const size_t hw = std::thread::hardware_concurrency();
    if (threads > 16) threads = 16;
    if (hw && threads > hw) threads = hw;
    if (threads < 2) { scan_range(d, n, 0, n, from_bit, out); return; }
    std::vector<std::vector<uint64_t>> found(threads);
    std::vector<std::thread> pool;
    const size_t per = (n + threads - 1) / threads;
    size_t started = 0;
    try {
        for (size_t t = 0; t < threads; t++) {
            pool.emplace_back([&, t] { scan_range(d, n, t * per, std::min(n, (t + 1) * per), from_bit, found[t]); });
            started = t + 1;
        }
    } catch (const std::system_error&) {}   // no more threads to be had: this one scans the ranges nobody took
    for (size_t t = started; t < threads; t++) scan_range(d, n, t * per, std::min(n, (t + 1) * per), from_bit, found[t]);
    for (auto& th : pool) th.join();
    for (auto& f : found) out.insert(out.end(), f.begin(), f.end());
}

struct Decoded {
    int status;
    std::vector<uint8_t> out;
    uint64_t end_bit;
};

// decompress(_:) BZip2.swift:50-95 replayed over pre-decoded blocks.  `byte_pos` is the aligned start of
// the stream; on return it is the aligned position just past the stream (callers align(), :45).
int walk_stream(const uint8_t* d, size_t n, size_t& byte_pos, const std::map<uint64_t, Decoded>& blocks,
                std::vector<uint8_t>& out) {
    const uint64_t total = (uint64_t)n * 8;
    uint64_t cur = (uint64_t)byte_pos * 8;
    if (total - cur < 32) return SWC_E_BZIP2_WRONG_MAGIC;                         // :53
    if (!(d[byte_pos] == 0x42 && d[byte_pos + 1] == 0x5A)) return SWC_E_BZIP2_WRONG_MAGIC;   // "BZ" :59-60
    if (d[byte_pos + 2] != 104) return SWC_E_BZIP2_WRONG_VERSION;                  // 'h' :62-63
    if (d[byte_pos + 3] < 0x31 || d[byte_pos + 3] > 0x39) return SWC_E_BZIP2_WRONG_BLOCK_SIZE;  // :65-66
    cur += 32;
    uint32_t total_crc = 0;
    for (;;) {
        if (total - cur < 80) return SWC_E_BZIP2_WRONG_MAGIC;                     // :71
        const uint64_t type = read_bits(d, cur, 48);
        const uint32_t crc = (uint32_t)read_bits(d, cur + 48, 32);
        if (type == kBlockMagic) {
            auto it = blocks.find(cur);
            if (it == blocks.end()) return SWC_E_DEVICE;                           // cannot happen: every magic was scanned
            const Decoded& b = it->second;
            if (b.status != SWC_OK && b.status != SWC_E_BZIP2_WRONG_CRC) return b.status;
            out.insert(out.end(), b.out.begin(), b.out.end());
            if (b.status == SWC_E_BZIP2_WRONG_CRC) return SWC_E_BZIP2_WRONG_CRC;   // :81 carries everything so far
            total_crc = (total_crc << 1) | (total_crc >> 31);                      // :83-84
            total_crc ^= crc;
            cur = b.end_bit;
        } else if (type == kEosMagic) {
            if (total_crc != crc) return SWC_E_BZIP2_WRONG_CRC;                    // :86
            cur += 80;
            break;
        } else {
            return SWC_E_BZIP2_WRONG_BLOCK_TYPE;                                   // :89
        }
    }
    byte_pos = (size_t)((cur + 7) >> 3);
    return SWC_OK;
}

}  // namespace

// Every bit offset at which the block magic occurs (candidates: a magic inside block data is possible).
void bzip2_magic_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out) {
    std::vector<uint64_t> cand;
    scan_block_magics(in, in_len, 32, cand);
    for (uint64_t c : cand) out.push_back({c, 0, 0, 0});
}

// Units for every candidate block of `d` (appended to `units`; `used` receives their bit offsets).
void bzip2_collect_candidates(const uint8_t* d, size_t n, std::vector<HostUnit>& units, std::vector<uint64_t>& used) {
    std::vector<uint64_t> cand;
    scan_block_magics(d, n, 32, cand);
    // first capacity of a block: what the stream header's level allows for the BWT column (level x 100,000 bytes,
    // BZip2+BlockSize.swift:11-33) -- the output is larger only where RLE1 runs expand, and such blocks are relaunched with
    // the size they report
    const size_t level = (n >= 4 && d[0] == 0x42 && d[1] == 0x5A && d[2] == 104 && d[3] >= 0x31 && d[3] <= 0x39) ? (size_t)(d[3] - 0x30) : 9;
    for (uint64_t c : cand) {
        if ((uint64_t)n * 8 - c < 80) continue;                                    // the walk reports wrongMagic there
        HostUnit u;
        u.in = d;
        u.in_len = n;
        u.extra = c + 80;                                                         // bit offset of the block body
        u.dict_value = read_bits(d, c + 48, 32);                                  // stored block CRC
        u.cap_hint = std::min<size_t>(level * 100000 + 64, std::max<size_t>(4096, n * 64));
        units.push_back(std::move(u));
        used.push_back(c);
    }
}

namespace {
void build_blocks(std::vector<HostUnit>& units, size_t first_unit, const std::vector<uint64_t>& used, std::map<uint64_t, Decoded>& blocks) {
    for (size_t i = 0; i < used.size(); i++) {
        HostUnit& u = units[first_unit + i];
        Decoded dec;
        dec.status = u.status;
        dec.out = std::move(u.out);
        dec.end_bit = u.in_consumed;                                              // bits for this codec
        blocks.emplace(used[i], std::move(dec));
    }
}
}  // namespace

// The sequential walk of one stream over its decoded candidates units[first_unit .. first_unit + used.size()).
int bzip2_finish_stream(const uint8_t* d, size_t n, std::vector<HostUnit>& units, size_t first_unit, const std::vector<uint64_t>& used,
                        std::vector<uint8_t>& res, size_t& byte_pos) {
    std::map<uint64_t, Decoded> blocks;
    build_blocks(units, first_unit, used, blocks);
    return walk_stream(d, n, byte_pos, blocks, res);
}

namespace {

// Decode every candidate block of `d` on the device in one batch.
int decode_candidates(const uint8_t* d, size_t n, std::map<uint64_t, Decoded>& blocks) {
    std::vector<HostUnit> units;
    std::vector<uint64_t> used;
    bzip2_collect_candidates(d, n, units, used);
    if (!units.empty()) {
        int st = run_units(SWC_CODEC_BZIP2_BLOCK, units);
        if (st) return st;
    }
    build_blocks(units, 0, used, blocks);
    return SWC_OK;
}

}  // namespace
}  // namespace swc

namespace swc { int bzip2_compress_device(const uint8_t* data, size_t len, int level, uint8_t** out, size_t* out_len); }   // bzip2_compress.hip
using namespace swc;

extern "C" {

int swc_bzip2_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t* in_consumed) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    std::map<uint64_t, Decoded> blocks;
    std::vector<uint8_t> res;
    int st = decode_candidates(in, in_len, blocks);
    if (st) { give_empty(out, out_len); return st; }
    size_t pos = 0;
    st = walk_stream(in, in_len, pos, blocks, res);
    if (st != SWC_OK && st != SWC_E_BZIP2_WRONG_CRC) res.clear();                  // only wrongCRC carries data
    if (in_consumed) *in_consumed = pos;
    give(res, out, out_len);
    return st;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

int swc_bzip2_multi_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t** sizes, size_t* n_streams) try {
    if (!out || !out_len || !sizes || !n_streams || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    std::map<uint64_t, Decoded> blocks;
    std::vector<uint8_t> all;
    std::vector<size_t> sz;
    int st = decode_candidates(in, in_len, blocks);
    if (st) { give_empty(out, out_len); *sizes = give_sizes(sz); *n_streams = 0; return st; }
    size_t pos = 0;
    while (pos < in_len) {                                                         // :43 !reader.isFinished
        std::vector<uint8_t> one;
        st = walk_stream(in, in_len, pos, blocks, one);
        if (st == SWC_E_BZIP2_WRONG_CRC) { all = std::move(one); sz.assign(1, all.size()); break; }  // carries the failing archive only
        if (st) { all.clear(); sz.clear(); break; }
        sz.push_back(one.size());
        all.insert(all.end(), one.begin(), one.end());
    }
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

// BZip2.compress(data:blockSize:) BZip2+Compress.swift:40-74 (BZip2.compress(data:) = block size one, :19-21)
int swc_bzip2_compress(const uint8_t* data, size_t len, int block_size, uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (len && !data) || block_size < 1 || block_size > 9) return SWC_E_INVALID_ARGUMENT;
    if (!device_ready()) { give_empty(out, out_len); return SWC_E_DEVICE; }
    const int st = bzip2_compress_device(data, len, block_size, out, out_len);   // assembled in the caller's buffer
    if (st) { give_empty(out, out_len); return st; }
    stat_add(0, 1);
    return SWC_OK;
} catch (...) {
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

}  // extern "C"
