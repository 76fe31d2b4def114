/*
 * refcpu.h -- CPU ORACLE for the SWCompression decode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This directory is a plain-C restatement of the reference's (tsolomko/SWCompression 4.9.0, Swift)
 * decode algorithms, function by function, including its deviations from "standard" decoders
 * (SURVEY.md Appendix A).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load librefcpu.so; the product (libswc_hip.so / swcompression_amd) never links or calls it.
 *
 * Pinning: the genuine reference is Swift and cannot be built here (no swiftc, BitByteData not
 * vendored), and its fixture submodule is empty.  The oracle is pinned against (i) every inline
 * golden vector that survives in the reference's Tests/ Swift sources (tests/golden/ref_inline_vectors.json,
 * SURVEY.md App. B) and (ii) three-way agreement with zlib / libbz2 / liblzma / liblz4 on valid
 * streams (tests/test_oracle_*.py).
 *
 * Conventions
 *  - every function returns an swc_status (include/swc_status.h);
 *  - *out is malloc()ed by the callee (free with refcpu_free) and is valid ALSO on error: it holds
 *    the bytes decoded so far (the reference carries them inside wrongCRC/wrongAdler32/... errors);
 *  - *in_consumed is the number of input bytes the reference's shared reader would have advanced
 *    (bit readers: after the align() every caller performs);
 *  - SWC_E_REF_TRAP = the Swift reference would hit a runtime trap on this input.
 */
#ifndef REFCPU_H
#define REFCPU_H

#include <stddef.h>
#include <stdint.h>
#include "../include/swc_status.h"

#ifdef __cplusplus
extern "C" {
#endif

void refcpu_free(void* p);
/* Upper bound for any single output buffer (default 1 GiB); exceeding it yields SWC_E_CAPACITY. */
void refcpu_set_max_output(size_t bytes);

/* ---- checksums (Sources/Common/CheckSums.swift, Sources/LZ4/XxHash32.swift, Sources/XZ/Sha256.swift) */
uint32_t refcpu_crc32(const uint8_t* p, size_t n, uint32_t prev);
uint32_t refcpu_bzip2crc32(const uint8_t* p, size_t n);
uint64_t refcpu_crc64(const uint8_t* p, size_t n);
uint32_t refcpu_adler32(const uint8_t* p, size_t n);
uint32_t refcpu_xxh32(const uint8_t* p, size_t n, uint32_t seed);
void refcpu_sha256(const uint8_t* p, size_t n, uint8_t digest[32]);

/* ---- Deflate (Sources/Deflate/Deflate.swift:30-249) */
int refcpu_deflate_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                              size_t* in_consumed);

/* ---- GZip / Zlib framing (Sources/GZip/GzipArchive.swift:38-100, Sources/Zlib/ZlibArchive.swift:25-42) */
int refcpu_gzip_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);
/* members are concatenated into *out; member_sizes (malloc'ed, n_members entries) gives the split. */
int refcpu_gzip_multi_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                                size_t** member_sizes, size_t* n_members);
int refcpu_zlib_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);

/* ---- BZip2 (Sources/BZip2/BZip2.swift:22-95) */
int refcpu_bzip2_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                            size_t* in_consumed);
int refcpu_bzip2_multi_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                                  size_t** stream_sizes, size_t* n_streams);

/* ---- LZMA / LZMA2 (Sources/LZMA/LZMA.swift:25-73, Sources/LZMA2/LZMA2.swift:25-36) */
int refcpu_lzma_decompress(const uint8_t* in, size_t in_len, int lc, int lp, int pb, int64_t dict_size,
                           int64_t uncompressed_size /* <0: unknown, end marker required */,
                           uint8_t** out, size_t* out_len, size_t* in_consumed);
int refcpu_lzma_alone_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);
int refcpu_lzma2_decompress(const uint8_t* in, size_t in_len, uint8_t dict_byte, uint8_t** out,
                            size_t* out_len, size_t* in_consumed);
/* LZMA2.decompress(data:) -- first byte is the dictionary-size byte */
int refcpu_lzma2_decompress_data(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);

/* ---- XZ (Sources/XZ/XZArchive.swift:27-218) */
int refcpu_xz_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);
int refcpu_xz_split_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len,
                              size_t** stream_sizes, size_t* n_streams);
void refcpu_delta_decode(const uint8_t* in, size_t n, int distance, uint8_t* out);

/* ---- LZ4 (Sources/LZ4/LZ4.swift:49-413) */
/* dict may be NULL (dict_len 0 with a non-NULL pointer is an EMPTY dictionary, which differs);
 * dict_id < 0 means "no dictionary id passed". */
int refcpu_lz4_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                          int64_t dict_id, uint8_t** out, size_t* out_len, size_t* in_consumed);
int refcpu_lz4_multi_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                                int64_t dict_id, uint8_t** out, size_t* out_len, size_t** frame_sizes,
                                size_t* n_frames);
/* LZ4.process(block:_:) -- one raw block with an optional prefix dictionary */
int refcpu_lz4_block(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len,
                     uint8_t** out, size_t* out_len);
/* Deflate.compress(data:) Deflate+Compress.swift:22-213: one stored or static-Huffman block */
int refcpu_deflate_compress(const uint8_t* data, size_t n, uint8_t* out, size_t cap, size_t* out_len);
/* BZip2.compress(data:blockSize:) BZip2+Compress.swift:41-325 (level 1..9 = BlockSize.one ... .nine) */
int refcpu_bzip2_compress(const uint8_t* data, size_t n, int level, uint8_t* out, size_t cap, size_t* out_len);
/* LZ4.compress(block:_:) LZ4+Compress.swift:157-281: bytes = dict ++ block, the block starts at `start` */
int refcpu_lz4_compress_block(const uint8_t* bytes, size_t total, size_t start, uint8_t* out, size_t cap, size_t* out_len);

/* ---- ZIP entry data (Sources/ZIP/ZipContainer.swift:61-118); the central-directory walk is the caller's */
int refcpu_zip_get_entry_data(const uint8_t* d, size_t len, uint64_t data_offset, uint64_t comp_size, uint64_t uncomp_size,
                              uint32_t crc32, int method, int has_data_descriptor, int zip64, uint8_t** out, size_t* out_len,
                              int* crc_error);

/* ---- 7-Zip folder (Sources/7-Zip/7zFolder.swift:138-194): the ORDERED coder chain with each coder's unpack size.
 * method: 0 copy, 1 deflate, 2 bzip2, 3 LZMA2, 4 LZMA, 5 Delta, 6 LZ4, 7 encryption, 8 other (layout == swc_7z_coder) */
typedef struct refcpu_7z_coder {
    uint32_t method;
    uint8_t props[5];
    uint8_t props_len;    /* 0xFF = absent */
    uint8_t multi_stream;
    uint8_t pad;
    uint64_t unpack_size;
} refcpu_7z_coder;
int refcpu_7z_unpack_folder(const uint8_t* data, size_t len, const refcpu_7z_coder* coders, size_t n_coders, uint8_t** out, size_t* out_len);

/* ---- timing harness (rc_pool.c): one independent unit per task over `threads` threads for `seconds` */
double refcpu_timed_pool(int codec, int aux, const uint8_t* const* ins, const size_t* lens, size_t n, int threads,
                         double seconds, uint64_t* out_bytes, uint64_t* in_bytes, uint64_t* units);

#ifdef __cplusplus
}
#endif
#endif
