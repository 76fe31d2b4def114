// framing.h -- host-side framing helpers shared between the C-ABI translation units.
#ifndef SWC_FRAMING_H
#define SWC_FRAMING_H
#include <vector>
#include "host_util.h"
namespace swc {
struct GzipHeaderInfo { uint32_t bgzf_bsize; };
int run_one(int codec, HostUnit& u);
void give(const std::vector<uint8_t>& src, uint8_t** out, size_t* out_len);
void give_empty(uint8_t** out, size_t* out_len);
size_t* give_sizes(const std::vector<size_t>& v);
int gzip_parse_header(const uint8_t* d, size_t n, size_t& pos, GzipHeaderInfo* info);
int gzip_member_prepare(const uint8_t* d, size_t n, size_t pos, HostUnit& u);
int gzip_member_finish(const uint8_t* d, size_t n, size_t data_pos, const HostUnit& u, size_t& next_pos, bool& crc_error);
int zlib_parse_header(const uint8_t* d, size_t n, size_t& pos);
}
#endif
