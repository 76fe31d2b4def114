// framing_deflate.cpp -- host side of the Deflate family: single-shot Deflate, and the GZip / Zlib
// archive framing that stays on the host and feeds the batched HIP launch.
//   Deflate.decompress(data:)            reference Sources/Deflate/Deflate.swift:24-28
//   GzipArchive.unarchive/multiUnarchive reference Sources/GZip/GzipArchive.swift:38-100
//   GzipHeader.init(_:)                  reference Sources/GZip/GzipHeader.swift:68-199
//   ZlibArchive.unarchive                reference Sources/Zlib/ZlibArchive.swift:25-42
//   ZlibHeader.init(_:)                  reference Sources/Zlib/ZlibHeader.swift:47-92
#include <thread>
#include <system_error>
#include <new>
#include <algorithm>
#include <vector>
#include "host_util.h"
#include "framing.h"

namespace swc {

int run_one(int codec, HostUnit& u) {
    std::vector<HostUnit> v(1);
    v[0] = std::move(u);
    int st = run_units(codec, v);
    u = std::move(v[0]);
    return st;
}

// run_one for a unit that is followed by unrelated data (the next members of a file, the next entries of a container):
// first with only `bound` bytes of input staged; the result stands if the unit decoded cleanly and ended inside them,
// anything else is decided by a second run over the whole tail -- so that a walk over M units stages O(file) bytes, not
// M times the rest of the file, and the outcome is the one the reference (which reads on from one reader) would have.
int run_one_bounded(int codec, HostUnit& u, size_t bound) {
    if (u.in_len <= bound) return run_one(codec, u);
    HostUnit t = u;
    t.in_len = bound;
    int st = run_one(codec, t);
    if (st != SWC_OK) return st;
    if (t.status == SWC_OK && t.in_consumed + 64 <= bound) { u = std::move(t); return SWC_OK; }
    return run_one(codec, u);
}

void give(const std::vector<uint8_t>& src, uint8_t** out, size_t* out_len) {
    uint8_t* p = host_result(src.size());
    if (!p) throw std::bad_alloc();   // (host_result is malloc: the entry points' function-try-blocks turn this into SWC_E_DEVICE)
    const size_t n = src.size(), nt = std::min<size_t>(8, n >> 24);         // a thread per 16 MB: one core copies 10 GB/s
    if (nt >= 2) {
        std::vector<std::thread> th;
        const size_t per = (n / nt + 63) & ~(size_t)63;
        size_t done = 0;   // bytes whose copy a thread has taken
        try {
            for (size_t t = 0; t < nt; t++) {
                const size_t lo = std::min(n, t * per), hi = t + 1 == nt ? n : std::min(n, (t + 1) * per);
                th.emplace_back([=, &src] { if (hi > lo) memcpy(p + lo, src.data() + lo, hi - lo); });
                done = hi;
            }
        } catch (const std::system_error&) {}   // no more threads to be had: this one copies the rest
        if (done < n) memcpy(p + done, src.data() + done, n - done);
        for (auto& t : th) t.join();
    } else if (n) memcpy(p, src.data(), n);
    *out = p;
    *out_len = src.size();
}
void give_empty(uint8_t** out, size_t* out_len) {
    *out = host_result(0);
    *out_len = 0;
}
size_t* give_sizes(const std::vector<size_t>& v) {
    size_t* s = static_cast<size_t*>(malloc((v.size() ? v.size() : 1) * sizeof(size_t)));
    for (size_t i = 0; i < v.size(); i++) s[i] = v[i];
    return s;
}

// ---- GZip header (host) ------------------------------------------------------------------------
// Walks one member header starting at `pos`; on success `pos` is the first byte of the Deflate
// stream.  Error taxonomy follows GzipHeader.swift line by line; `trap` marks inputs on which the
// reference would read past the end of its reader (a Swift precondition failure).
int gzip_parse_header(const uint8_t* d, size_t n, size_t& pos, GzipHeaderInfo* info) {
    struct Cursor {
        const uint8_t* d; size_t n; size_t p; bool trap;
        size_t left() const { return n - p; }
        uint8_t u8() { if (p >= n) { trap = true; return 0; } return d[p++]; }
        uint32_t le16() { uint32_t a = u8(); uint32_t b = u8(); return a | (b << 8); }
    } c{d, n, pos, false};
    const size_t start = pos;
    if (c.left() < 10) return SWC_E_GZIP_WRONG_MAGIC;                  // GzipHeader.swift:70
    if (c.le16() != 0x8b1f) return SWC_E_GZIP_WRONG_MAGIC;             // :75
    if (c.u8() != 8) return SWC_E_GZIP_WRONG_COMPRESSION_METHOD;       // :81
    const uint32_t flags = c.u8();
    if (flags & 0xE0) return SWC_E_GZIP_WRONG_FLAGS;                   // :87
    c.p += 6;                                                          // MTIME, XFL, OS
    if (info) { info->bgzf_bsize = 0; }
    if (flags & 0x04) {                                                // FEXTRA :110-156
        if (c.left() < 2) return SWC_E_GZIP_WRONG_MAGIC;
        int64_t xlen = c.le16();
        if (!((int64_t)c.left() >= xlen && xlen >= 4)) return SWC_E_GZIP_WRONG_MAGIC;  // :123
        while (xlen > 0) {
            uint8_t si1 = c.u8();
            uint8_t si2 = c.u8();
            if (c.trap) return SWC_E_REF_TRAP;
            if (si2 == 0) return SWC_E_GZIP_WRONG_FLAGS;               // :131
            int64_t len = c.le16();
            if (c.trap) return SWC_E_REF_TRAP;
            xlen -= 4;
            if (xlen < len) return SWC_E_GZIP_WRONG_MAGIC;             // :145
            if ((int64_t)c.left() < len) return SWC_E_REF_TRAP;
            if (info && si1 == 'B' && si2 == 'C' && len == 2)           // BGZF: BSIZE = member size - 1
                info->bgzf_bsize = (uint32_t)c.d[c.p] | ((uint32_t)c.d[c.p + 1] << 8);
            c.p += (size_t)len;
            xlen -= len;
        }
    }
    for (uint32_t bit : {0x08u, 0x10u}) {                              // FNAME :158-172, FCOMMENT :174-188
        if (flags & bit) {
            for (;;) {
                if (c.p >= c.n) return SWC_E_GZIP_WRONG_MAGIC;
                if (c.d[c.p++] == 0) break;
            }
        }
    }
    if (flags & 0x02) {                                                // FHCRC :190-198
        if (c.left() < 2) return SWC_E_GZIP_WRONG_MAGIC;
        const size_t hend = c.p;
        uint32_t crc16 = c.le16();
        if ((swc_crc32(d + start, hend - start, 0) & 0xFFFF) != crc16) return SWC_E_GZIP_WRONG_HEADER_CRC;
    }
    if (c.trap) return SWC_E_REF_TRAP;
    pos = c.p;
    return SWC_OK;
}

// processMember (GzipArchive.swift:79-100) split in two around the device call.
int gzip_member_prepare(const uint8_t* d, size_t n, size_t pos, HostUnit& u) {
    if (n - pos < 20) return SWC_E_GZIP_WRONG_MAGIC;                   // :83 (members are byte aligned by construction)
    size_t p = pos;
    int st = gzip_parse_header(d, n, p, nullptr);
    if (st) return st;
    u = HostUnit();
    u.in = d + p;
    u.in_len = n - p;
    // ISIZE of a single-member archive is its last four bytes: a good capacity guess, never trusted.
    uint32_t isize = (uint32_t)d[n - 4] | (uint32_t)d[n - 3] << 8 | (uint32_t)d[n - 2] << 16 | (uint32_t)d[n - 1] << 24;
    if ((uint64_t)isize <= (uint64_t)u.in_len * 1100 + 4096) u.cap_hint = std::max<size_t>(isize, 64);
    return SWC_OK;
}
// `u` holds the decoded Deflate stream; `data_pos` = offset of the Deflate stream in `d`.
int gzip_member_finish(const uint8_t* d, size_t n, size_t data_pos, const HostUnit& u, size_t& next_pos, bool& crc_error) {
    crc_error = false;
    if (u.status) return u.status;                                     // DeflateError propagates
    size_t p = data_pos + u.in_consumed;                               // align() already applied by the engine
    if (n - p < 8) return SWC_E_GZIP_WRONG_MAGIC;                      // :91
    uint32_t crc = (uint32_t)d[p] | (uint32_t)d[p + 1] << 8 | (uint32_t)d[p + 2] << 16 | (uint32_t)d[p + 3] << 24;
    uint32_t isize = (uint32_t)d[p + 4] | (uint32_t)d[p + 5] << 8 | (uint32_t)d[p + 6] << 16 | (uint32_t)d[p + 7] << 24;
    if ((uint32_t)(u.size() & 0xFFFFFFFFu) != isize) return SWC_E_GZIP_WRONG_ISIZE;  // :95
    // :99 -- the CRC-32 the device computed behind the decode (HostUnit::sum_kind = 1), else here
    crc_error = (u.sum_valid && u.sum_kind == 1 ? (uint32_t)u.sum : swc_crc32(u.data(), u.size(), 0)) != crc;
    next_pos = p + 8;
    return SWC_OK;
}

// ZlibHeader.init (ZlibHeader.swift:47-92): returns offset of the Deflate stream in `pos`.
int zlib_parse_header(const uint8_t* d, size_t n, size_t& pos) {
    if (n < 2) return SWC_E_ZLIB_WRONG_COMPRESSION_METHOD;             // :49
    const uint32_t cmf = d[0], flg = d[1];
    if ((cmf & 0x0F) != 8) return SWC_E_ZLIB_WRONG_COMPRESSION_METHOD; // :57
    if ((cmf >> 4) > 7) return SWC_E_ZLIB_WRONG_COMPRESSION_INFO;      // :63
    if (((cmf << 8) + flg) % 31 != 0) return SWC_E_ZLIB_WRONG_FCHECK;  // :83
    size_t p = 2;
    if (flg & 0x20) {                                                  // FDICT: skip DICTID :87-90
        if (n - p < 4) return SWC_E_ZLIB_WRONG_FCHECK;
        p += 4;
    }
    pos = p;
    return SWC_OK;
}

}  // namespace swc

using namespace swc;

extern "C" {

int swc_deflate_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t* in_consumed) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    HostUnit u;
    u.in = in; u.in_len = in_len;
    int st = run_one(SWC_CODEC_DEFLATE, u);
    if (st) { give_empty(out, out_len); return st; }
    if (in_consumed) *in_consumed = u.in_consumed;
    if (u.status) { give_empty(out, out_len); return u.status; }       // DeflateError cases carry no data
    give(u.out, out, out_len);
    return SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

int swc_gzip_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    HostUnit u;
    int st = gzip_member_prepare(in, in_len, 0, u);
    if (st) { give_empty(out, out_len); return st; }
    const size_t data_pos = (size_t)(u.in - in);
    u.sum_kind = 1;                                                    // CRC-32 on the device, behind the decode
    st = run_one(SWC_CODEC_DEFLATE, u);
    if (st) { give_empty(out, out_len); return st; }
    size_t next; bool crc_error;
    st = gzip_member_finish(in, in_len, data_pos, u, next, crc_error);
    if (st) { give_empty(out, out_len); return st; }
    give(u.out, out, out_len);                                         // wrongCRC carries the member (:44)
    return crc_error ? SWC_E_GZIP_WRONG_CRC : SWC_OK;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

// BGZF (and any multi-member gzip whose members carry the 'BC' extra field): BSIZE = member size - 1 locates every member
// without decoding, so ALL members go to the device in one batch.  The result is used only if every member decoded cleanly
// from exactly the bytes its BSIZE assigns to it -- then the sequential walk of GzipArchive.multiUnarchive (:62-77) would have
// met the same members at the same offsets; anything else (a member without the field, a decode error, a stream that ends
// before or after its BSIZE, a CRC / ISIZE mismatch) falls back to that walk, which alone defines errors and partial results.
}  // extern "C"
namespace swc {
// Every member of a BGZF file: offset / length of its Deflate stream and its ISIZE.  False if a member lacks the field.
bool bgzf_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out) {
    size_t pos = 0;
    while (pos < in_len) {
        if (in_len - pos < 20) return false;
        size_t p = pos;
        GzipHeaderInfo info;
        if (gzip_parse_header(in, in_len, p, &info) != SWC_OK || info.bgzf_bsize == 0) return false;
        const size_t total = (size_t)info.bgzf_bsize + 1;
        if (total > in_len - pos || p - pos + 8 > total) return false;
        const uint8_t* t = in + pos + total - 8;
        const uint32_t isize = (uint32_t)t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
        out.push_back({p, total - (p - pos) - 8, isize, 0});
        pos += total;
    }
    return true;
}
}  // namespace swc
extern "C" {
// On success *all_out is the malloc()ed concatenation of the members (the caller's result buffer: every member is copied
// from the staging buffer straight to its place in it, and its CRC-32 comes from the device).
static bool bgzf_multi(const uint8_t* in, size_t in_len, uint8_t** all_out, size_t* all_len, std::vector<size_t>& sz) {
    struct Member { size_t data_pos, data_len, trailer; };
    std::vector<Member> members;
    {
        std::vector<BlockRef64> refs;
        if (!bgzf_index(in, in_len, refs)) return false;
        for (const BlockRef64& r : refs) members.push_back({(size_t)r.offset, (size_t)r.comp_len, (size_t)(r.offset + r.comp_len)});
    }
    if (members.size() < 2) return false;
    std::vector<HostUnit> units(members.size());
    size_t total = 0;
    for (size_t k = 0; k < members.size(); k++) {
        const uint8_t* t = in + members[k].trailer;
        const uint32_t isize = (uint32_t)t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
        units[k].in = in + members[k].data_pos;
        units[k].in_len = members[k].data_len;
        units[k].base = in;                 // one staged copy of the file, every member a sub-range of it
        units[k].base_len = in_len;
        units[k].sum_kind = 1;              // CRC-32 on the device
        if ((uint64_t)isize > (uint64_t)members[k].data_len * 1100 + 4096) return false;   // (not a size this member can have: the walk decides)
        units[k].cap_hint = std::max<size_t>(isize, 64);
        units[k].dst_cap = isize;
        total += isize;
    }
    // (ISIZE is a field of the file: the sum of them sizes ONE host buffer before a byte is decoded.  A file whose trailers claim
    // more than 64 x its own size -- or 64 MiB -- goes the sequential walk, which grows its result with the bytes it decodes.)
    if (total > std::max<size_t>((size_t)64 << 20, in_len * 64)) return false;
    uint8_t* all = host_result(total);
    if (!all) return false;
    {
        size_t o = 0;
        for (HostUnit& u : units) { u.dst = all + o; o += u.dst_cap; }
    }
    bool ok = run_units(SWC_CODEC_DEFLATE, units) == SWC_OK;
    for (size_t k = 0; ok && k < members.size(); k++) {
        const HostUnit& u = units[k];
        size_t next;
        bool crc_error;
        ok = u.status == SWC_OK && u.in_consumed == members[k].data_len && u.in_dst && u.out_size == u.dst_cap &&
             gzip_member_finish(in, in_len, members[k].data_pos, u, next, crc_error) == SWC_OK && !crc_error;
    }
    if (!ok) { host_result_free(all); return false; }
    for (const HostUnit& u : units) sz.push_back(u.out_size);
    *all_out = all;
    *all_len = total;
    return true;
}

int swc_gzip_multi_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t** sizes, size_t* n_members) try {
    if (!out || !out_len || !sizes || !n_members || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    {
        std::vector<size_t> fsz;
        if (bgzf_multi(in, in_len, out, out_len, fsz)) {
            *sizes = give_sizes(fsz);
            *n_members = fsz.size();
            return SWC_OK;
        }
    }
    // Plain multi-member gzip has no compressed-size field: member k+1 can only be located after
    // member k has been inflated, so members are discovered sequentially here.
    // (BGZF and many-archive batches go through swc_unarchive_many, which launches them together.)
    std::vector<uint8_t> all;
    std::vector<size_t> sz;
    size_t pos = 0, walk_bound = (size_t)4 << 20;
    int st = SWC_OK;
    while (pos < in_len) {                                             // :66
        HostUnit u;
        st = gzip_member_prepare(in, in_len, pos, u);
        if (st) break;
        u.cap_hint = 0;                                                // ISIZE guess only valid for the last member
        const size_t data_pos = (size_t)(u.in - in);
        st = run_one_bounded(SWC_CODEC_DEFLATE, u, walk_bound);
        if (st) break;
        walk_bound = std::max<size_t>(walk_bound, 4 * u.in_consumed + 65536);   // the next member is probably of this size
        bool crc_error;
        st = gzip_member_finish(in, in_len, data_pos, u, pos, crc_error);
        if (st) break;
        all.insert(all.end(), u.out.begin(), u.out.end());
        sz.push_back(u.out.size());
        if (crc_error) { st = SWC_E_GZIP_WRONG_CRC; break; }           // :71-72 carries the members so far
    }
    if (st != SWC_OK && st != SWC_E_GZIP_WRONG_CRC) { all.clear(); sz.clear(); }
    give(all, out, out_len);
    *sizes = give_sizes(sz);
    *n_members = sz.size();
    return st;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    if (sizes) *sizes = nullptr;
    if (n_members) *n_members = 0;
    return SWC_E_DEVICE;
}

// Deflate.compress(data:) (Deflate+Compress.swift:22-46): a buffer of up to 1 MiB is one unit of SWC_CODEC_DEFLATE_COMPRESS -- one
// wavefront parses it, as the reference's one loop does (a wavefront's pace is about 5 MB/s: the engine's throughput is in the
// number of units of a launch).  The checksums of the archive writers (Adler-32, CRC-32 of the INPUT) are computed on the host by
// their callers.
// A LARGE buffer is cut into segments of kCompressSegment bytes, one wavefront each, all in one launch: every segment but the
// last comes back as a non-final static block followed by an empty stored block (so that it ends on a byte: job.aux bit 0,
// deflate_comp.h), and the segments one behind the other are one Deflate stream that the reference's decoder reads block by
// block (Deflate.swift:30-249).  What is given up: matches across a segment's start (under 0.1 % of the size at 256 KiB), and
// the reference's "one block" shape -- which SURVEY 8f-4 does not ask to keep.  Buffers of up to kCompressWhole bytes stay ONE
// block, as the reference writes them.
constexpr size_t kCompressWhole = (size_t)1 << 20, kCompressSegment = (size_t)256 << 10;
static int deflate_compress_unit(const uint8_t* data, size_t len, HostUnit& u) {
    if (len > kCompressWhole) {
        const size_t nseg = (len + kCompressSegment - 1) / kCompressSegment;
        std::vector<HostUnit> segs(nseg);
        for (size_t k = 0; k < nseg; k++) {
            HostUnit& s = segs[k];
            const size_t lo = k * kCompressSegment, n = std::min(kCompressSegment, len - lo);
            s.in = data + lo; s.in_len = n;
            s.base = data; s.base_len = len;            // one staged copy of the buffer, every segment a sub-range of it
            s.cap_hint = n + n / 8 + 32;
            s.cap_exact = true;
            s.aux = k + 1 < nseg ? 1 : 0;
        }
        const int st = run_units(SWC_CODEC_DEFLATE_COMPRESS, segs);
        if (st != SWC_OK) return st;
        size_t total = 0;
        for (const HostUnit& s : segs) { if (s.status != SWC_OK) { u.status = s.status; return SWC_OK; } total += s.size(); }
        u.out.clear();
        u.out.reserve(total);
        for (const HostUnit& s : segs) u.out.insert(u.out.end(), s.data(), s.data() + s.size());
        u.out_size = u.out.size();
        u.in_consumed = len;
        u.status = SWC_OK;
        return SWC_OK;
    }
    u.in = data; u.in_len = len;
    u.cap_hint = len + len / 8 + 16;
    u.cap_exact = true;
    return run_one(SWC_CODEC_DEFLATE_COMPRESS, u);
}
int swc_deflate_compress(const uint8_t* data, size_t len, uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (len && !data)) return SWC_E_INVALID_ARGUMENT;
    HostUnit u;
    int st = deflate_compress_unit(data, len, u);
    if (st == SWC_OK) st = u.status;
    if (st) { give_empty(out, out_len); return st; }
    give(u.out, out, out_len);
    return SWC_OK;
} catch (...) {
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}
// GzipArchive.archive(data:comment:fileName:writeHeaderCRC:isTextFile:osType:modificationTime:extraFields:) (GzipArchive.swift:126-240)
int swc_gzip_archive(const uint8_t* data, size_t len, const uint8_t* comment, size_t comment_len, const uint8_t* file_name,
                     size_t file_name_len, int write_header_crc, int is_text_file, int os_type, int has_mtime, int64_t mtime,
                     const swc_gzip_extra_field* extra, size_t n_extra, uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (len && !data) || (comment_len && !comment) || (file_name_len && !file_name) || (n_extra && !extra))
        return SWC_E_INVALID_ARGUMENT;
    uint8_t flags = 0;
    if (comment) flags |= 1 << 4;                                      // :131-132
    if (file_name) flags |= 1 << 3;                                    // :144-145
    if (n_extra) flags |= 1 << 2;                                      // :157-159
    if (write_header_crc) flags |= 1 << 1;                             // :161-163
    if (is_text_file) flags |= 1 << 0;                                 // :165-167
    std::vector<uint8_t> z = {0x1f, 0x8b, 8, flags};                   // :180-184
    for (int i = 0; i < 4; i++) z.push_back(has_mtime ? (uint8_t)((uint64_t)mtime >> (8 * i)) : 0);   // :171-178, :185-187
    z.push_back(2);                                                    // :188 XFL: the slowest algorithm
    z.push_back((uint8_t)os_type);                                     // :169, :189
    if (n_extra) {                                                     // :191-209
        size_t xlen = 0;
        for (size_t k = 0; k < n_extra; k++) {
            if (extra[k].len && !extra[k].bytes) { give_empty(out, out_len); return SWC_E_INVALID_ARGUMENT; }
            xlen += 4 + extra[k].len;
        }
        if (xlen > 65535) { give_empty(out, out_len); return SWC_E_GZIP_CANNOT_ENCODE_ISO_LATIN1; }
        z.push_back((uint8_t)(xlen & 0xFF));
        z.push_back((uint8_t)(xlen >> 8));
        for (size_t k = 0; k < n_extra; k++) {
            z.push_back(extra[k].si1);
            z.push_back(extra[k].si2);
            z.push_back((uint8_t)(extra[k].len & 0xFF));
            z.push_back((uint8_t)((extra[k].len >> 8) & 0xFF));
            z.insert(z.end(), extra[k].bytes, extra[k].bytes + extra[k].len);
        }
    }
    auto terminated = [&](const uint8_t* p, size_t n) {                // :134-136, :147-149: a zero behind it unless it ends in one
        z.insert(z.end(), p, p + n);
        if (n == 0 || p[n - 1] != 0) z.push_back(0);
    };
    if (file_name) terminated(file_name, file_name_len);               // :213
    if (comment) terminated(comment, comment_len);                     // :214
    if (write_header_crc) {                                            // :216-221
        const uint32_t h = swc_crc32(z.data(), z.size(), 0);
        z.push_back((uint8_t)h);
        z.push_back((uint8_t)(h >> 8));
    }
    HostUnit u;
    int st = deflate_compress_unit(data, len, u);                      // :223
    if (st == SWC_OK) st = u.status;
    if (st) { give_empty(out, out_len); return st; }
    z.insert(z.end(), u.out.begin(), u.out.end());
    const uint32_t c = swc_crc32(data, len, 0);                        // :225-230
    for (int i = 0; i < 4; i++) z.push_back((uint8_t)(c >> (8 * i)));
    const uint64_t isize = (uint64_t)len % ((uint64_t)1 << 32);       // :232-237
    for (int i = 0; i < 4; i++) z.push_back((uint8_t)(isize >> (8 * i)));
    give(z, out, out_len);
    return SWC_OK;
} catch (...) {
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}
// ZlibArchive.archive(data:) (ZlibArchive.swift:54-70)
int swc_zlib_archive(const uint8_t* data, size_t len, uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (len && !data)) return SWC_E_INVALID_ARGUMENT;
    HostUnit u;
    int st = deflate_compress_unit(data, len, u);
    if (st == SWC_OK) st = u.status;
    if (st) { give_empty(out, out_len); return st; }
    std::vector<uint8_t> z;
    z.reserve(u.out.size() + 6);
    z.push_back(120);                                                  // :56 CM = 8, CINFO = 7
    z.push_back(218);                                                  // :57 slowest algorithm, no preset dictionary
    z.insert(z.end(), u.out.begin(), u.out.end());
    const uint32_t a = swc_adler32(data, len);                         // :62-66 big endian
    for (int i = 3; i >= 0; i--) z.push_back((uint8_t)(a >> (8 * i)));
    give(z, out, out_len);
    return SWC_OK;
} catch (...) {
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

int swc_zlib_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) try {
    if (!out || !out_len || (in_len && !in)) return SWC_E_INVALID_ARGUMENT;
    size_t p = 0;
    int st = zlib_parse_header(in, in_len, p);
    if (st) { give_empty(out, out_len); return st; }
    HostUnit u;
    u.in = in + p; u.in_len = in_len - p;
    u.sum_kind = 2;                                                    // Adler-32 on the device, behind the decode
    st = run_one(SWC_CODEC_DEFLATE, u);
    if (st) { give_empty(out, out_len); return st; }
    if (u.status) { give_empty(out, out_len); return u.status; }
    size_t q = p + u.in_consumed;
    give(u.out, out, out_len);                                         // wrongAdler32 carries the data (:34,39)
    if (in_len - q < 4) return SWC_E_ZLIB_WRONG_ADLER32;
    uint32_t stored = (uint32_t)in[q] << 24 | (uint32_t)in[q + 1] << 16 | (uint32_t)in[q + 2] << 8 | in[q + 3];
    return (u.sum_valid ? (uint32_t)u.sum : swc_adler32(u.out.data(), u.out.size())) == stored ? SWC_OK : SWC_E_ZLIB_WRONG_ADLER32;
} catch (...) {   // std::bad_alloc / length_error from a size taken from the input: never through the C boundary
    if (out && out_len) give_empty(out, out_len);
    return SWC_E_DEVICE;
}

}  // extern "C"
