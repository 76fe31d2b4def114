// framing.h -- host-side framing helpers shared between the C-ABI translation units.
#ifndef SWC_FRAMING_H
#define SWC_FRAMING_H
#include <vector>
#include "host_util.h"
namespace swc {
struct GzipHeaderInfo { uint32_t bgzf_bsize; };
int run_one(int codec, HostUnit& u);
int run_one_bounded(int codec, HostUnit& u, size_t bound);
void give(const std::vector<uint8_t>& src, uint8_t** out, size_t* out_len);
void give_empty(uint8_t** out, size_t* out_len);
size_t* give_sizes(const std::vector<size_t>& v);
int gzip_parse_header(const uint8_t* d, size_t n, size_t& pos, GzipHeaderInfo* info);
int gzip_member_prepare(const uint8_t* d, size_t n, size_t pos, HostUnit& u);
int gzip_member_finish(const uint8_t* d, size_t n, size_t data_pos, const HostUnit& u, size_t& next_pos, bool& crc_error);
int zlib_parse_header(const uint8_t* d, size_t n, size_t& pos);
// many-archive batching helpers (framing_many.cpp)
struct Lz4Plan {
    struct Impl;
    Impl* impl;
    int early_status = SWC_OK;
    size_t first_unit = 0;
    Lz4Plan();
    ~Lz4Plan();
    Lz4Plan(const Lz4Plan&) = delete;
    Lz4Plan& operator=(const Lz4Plan&) = delete;
};
bool lz4_plan_prepare(const uint8_t* in, size_t n, Lz4Plan& plan, std::vector<HostUnit>& units);
int lz4_plan_finish(const uint8_t* in, size_t n, const Lz4Plan& plan, const std::vector<HostUnit>& units, std::vector<uint8_t>& res);
void bzip2_collect_candidates(const uint8_t* d, size_t n, std::vector<HostUnit>& units, std::vector<uint64_t>& used);
int bzip2_finish_stream(const uint8_t* d, size_t n, std::vector<HostUnit>& units, size_t first_unit, const std::vector<uint64_t>& used,
                        std::vector<uint8_t>& res, size_t& byte_pos);
size_t lzma2_announced_size(const uint8_t* p, size_t n);
// block discovery for swc_index_blocks (framing_many.cpp)
struct BlockRef64 { uint64_t offset, comp_len, uncomp_len; uint32_t aux; uint32_t flags = 0; };
bool bgzf_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out);
bool lz4_frame_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out);
void bzip2_magic_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out);
void xz_block_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out);
int lzma2_chunk_index(const uint8_t* in, size_t in_len, std::vector<BlockRef64>& out);
}
#endif
