/*
 * swc_hip.h -- C ABI of libswc_hip.so, the MI355X (gfx950) many-stream decode engine that sits behind
 * SWCompression's `decompress(data:)` entry points.
 *
 * The reference (tsolomko/SWCompression 4.9.0) is pure Swift and has no FFI layer; its boundary for
 * this path is the protocol `DecompressionAlgorithm { static func decompress(data: Data) throws -> Data }`
 * (Sources/Common/DecompressionAlgorithm.swift:9-14) plus the internal reader-taking overloads that
 * the archive/container layers call.  Each entry point below names the Swift function whose BODY it
 * replaces (file:line under /root/reference); INTEGRATION.md shows the Swift-side binding.
 *
 * Rules common to all entry points
 *  - return value / swc_job.status is an swc_status (swc_status.h): 0 = OK, otherwise 1:1 with the Swift
 *    error enum case the reference would throw, SWC_E_REF_TRAP where the reference would trap.
 *  - single-shot calls take HOST buffers, run the HIP kernels (there is no CPU fallback: without a
 *    usable gfx950 device they return SWC_E_DEVICE), and hand back a malloc()ed buffer the caller
 *    releases with swc_free().  *out is valid also for the error codes that carry data in the reference
 *    (wrongCRC / wrongAdler32 / wrongCheck / checksumMismatch).
 *  - *in_consumed reports how far the reference's shared reader would have advanced, so callers can
 *    keep parsing trailers exactly like GzipArchive.swift:88-94 / ZlibArchive.swift:31-37 /
 *    ZipContainer.swift:73-86 / XZBlock.swift:78-82 do.
 *  - thread-safe; never aborts the process.
 */
#ifndef SWC_HIP_H
#define SWC_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "swc_status.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Batched many-buffer launch (the engine proper).  All pointers inside swc_job are DEVICE pointers
 * (HBM-resident input and output); the job array itself is in device memory too.
 * ---------------------------------------------------------------------------------------------- */
typedef enum swc_codec {
    SWC_CODEC_DEFLATE = 1, /* raw RFC 1951 stream      -- Deflate.decompress(_:)   Deflate.swift:30-249     */
    SWC_CODEC_LZ4_BLOCK = 2, /* one LZ4 block           -- LZ4.process(block:_:)    LZ4.swift:332-413        */
    SWC_CODEC_LZMA2 = 3,   /* raw LZMA2 chunk stream   -- LZMA2Decoder.decode()    LZMA2Decoder.swift:34-99 */
    SWC_CODEC_LZMA = 4,    /* raw LZMA1 stream         -- LZMADecoder.decode()     LZMADecoder.swift:107-284*/
    SWC_CODEC_BZIP2_BLOCK = 5, /* one bzip2 block body -- BZip2.decode(_:_:)       BZip2.swift:97-270       */
    SWC_CODEC_DELTA = 6,   /* XZ / 7-Zip Delta filter  -- DeltaFilter.decode(_:_:) DeltaFilter.swift:11-33; aux = distance as the
                              reference passes it (XZBlock.swift:57: property + 1), out may equal in            */
    SWC_CODEC_LZ4_COMPRESS = 7, /* ENCODE, one LZ4 block -- LZ4.compress(block:_:) LZ4+Compress.swift:157-281: in = prefix ++ block,
                              dict_len = length of the prefix (dictionary / previous block), out_cap >= n + n / 255 + 16;
                              A valid block for the same bytes, not the reference's bytes            */
    SWC_CODEC_DEFLATE_COMPRESS = 8 /* ENCODE, one raw Deflate stream -- Deflate.compress(data:) Deflate+Compress.swift:22-213: one
                              stored or static-Huffman block over a greedy LZ77 parse; out_cap >= n + n / 8 + 32, out 4-byte
                              aligned; A valid stream for the same bytes, not the reference's bytes.  aux bit 0:
                              the unit is a SEGMENT of a longer stream -- BFINAL clear, an empty stored block behind the block, so
                              that the outputs of consecutive segments, the last one with aux = 0, are one stream           */
} swc_codec;

typedef struct swc_job {
    const uint8_t* in;    /* device pointer to the unit's compressed bytes                           */
    uint64_t in_len;
    uint8_t* out;         /* device pointer, caller allocated                                        */
    uint64_t out_cap;
    uint64_t out_len;     /* OUT: bytes produced; for SWC_E_CAPACITY on Deflate/LZ4: bytes required   */
    uint64_t in_consumed; /* OUT                                                                     */
    int32_t status;       /* OUT: swc_status                                                         */
    int32_t aux;          /* IN : LZMA2 dictionary-size byte | LZMA props (lc | lp<<8 | pb<<16) | BZIP2: start bit (0..7) */
    const uint8_t* dict;  /* IN : LZ4 prefix dictionary (device) or NULL                             */
    uint64_t dict_len;    /* IN : LZ4 dictionary length | LZMA: uncompressed size (UINT64_MAX = unknown) | LZMA dict size in the high half, see swc_hip.h notes */
} swc_job;

typedef struct swc_batch_opts {
    int32_t device;        /* HIP device ordinal, -1 = current                                       */
    void* stream;          /* hipStream_t, NULL = default stream                                     */
    int32_t synchronize;   /* non-zero: wait for completion before returning                         */
    int32_t reserved;
} swc_batch_opts;

/* One launch for all n jobs.  `jobs` is a device pointer to n swc_job records.  Deflate needs an HBM workspace:
 * this entry point allocates it from the stream-ordered pool; swc_batch_decompress_ws takes it from the caller
 * (required for BZip2, recommended for LZ4). */
int swc_batch_decompress(int codec, swc_job* jobs, size_t n, const swc_batch_opts* opts);

/* Scratch bytes the codec needs in HBM (Deflate / LZ4: match records + literal stream of the two-phase path,
 * LZMA: the home of the literal coders (LDS caches four lines of them) and of the long-length trees -- 32 streams per CU and any lc + lp; without it the model
 * of lc + lp <= 4 sits in LDS whole, 5 streams per CU, larger models are refused with SWC_E_NEED_WORKSPACE; BZip2: tt[]).
 * LZ4 also runs without it (one block per lane, much slower on large blocks).  Allocated internally by the single-shot
 * calls; batch callers pass it via swc_batch_decompress_ws. */
size_t swc_batch_workspace_bytes(int codec, size_t n_jobs, uint64_t max_out_cap);
int swc_batch_decompress_ws(int codec, swc_job* jobs, size_t n, void* workspace, size_t workspace_bytes,
                            const swc_batch_opts* opts);

/* CRC-32 (CheckSums.crc32, reference Sources/Common/CheckSums.swift:12-28, as checked by GzipArchive.swift:99)
 * of every job's output, computed on the device after a batch has been decoded: crcs[i] covers
 * out[0 .. min(out_len, out_cap)).  `jobs` and `crcs` are device pointers. */
int swc_batch_crc32(const swc_job* jobs, size_t n, uint32_t* crcs, const swc_batch_opts* opts);

/* The other checksums of the archive layer over every job's output (same coverage rule), zero-extended to 64 bits:
 *   SWC_SUM_CRC32        CheckSums.crc32       CheckSums.swift:12-28  (GzipArchive.swift:99, XZArchive.swift:109-120)
 *   SWC_SUM_ADLER32      CheckSums.adler32     CheckSums.swift:48-57  (ZlibArchive.swift:38)
 *   SWC_SUM_CRC64        CheckSums.crc64       CheckSums.swift:39-46  (XZArchive.swift:109-120)
 *   SWC_SUM_BZIP2_CRC32  CheckSums.bzip2crc32  CheckSums.swift:30-37  (BZip2.swift:81)
 *   SWC_SUM_XXH32        XxHash32.hash(seed 0) XxHash32.swift:24-83   (LZ4.swift:300,326)
 * `jobs` and `sums` are device pointers. */
typedef enum swc_checksum {
    SWC_SUM_CRC32 = 1,
    SWC_SUM_ADLER32 = 2,
    SWC_SUM_CRC64 = 3,
    SWC_SUM_BZIP2_CRC32 = 4,
    SWC_SUM_XXH32 = 5
} swc_checksum;
int swc_batch_checksum(int kind, const swc_job* jobs, size_t n, uint64_t* sums, const swc_batch_opts* opts);

/* ------------------------------------------------------------------------------------------------
 * Single-shot calls (host buffers in, malloc()ed host buffer out).
 * ---------------------------------------------------------------------------------------------- */
/* Deflate.decompress(data:) Deflate.swift:24-28 and decompress(_ bitReader:) :30-249
 * (callers GzipArchive.swift:88, ZlibArchive.swift:31, ZipContainer.swift:74) */
int swc_deflate_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t* in_consumed);

/* BZip2.decompress(data:) BZip2.swift:22-26 and decompress(_ bitReader:) :50-95 (caller ZipContainer.swift:82);
 * one stream; *out valid on SWC_E_BZIP2_WRONG_CRC */
int swc_bzip2_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t* in_consumed);
/* BZip2.multiDecompress(data:) BZip2.swift:40-48; sizes[] (swc_free) splits *out per stream */
int swc_bzip2_multi_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t** sizes, size_t* n_streams);

/* LZMA.decompress(data:properties:uncompressedSize:) LZMA.swift:56-73 (caller ZipContainer.swift:89);
 * uncompressed_size < 0 = unknown (end marker required) */
int swc_lzma_decompress(const uint8_t* in, size_t in_len, int lc, int lp, int pb, int64_t dict_size,
                        int64_t uncompressed_size, uint8_t** out, size_t* out_len, size_t* in_consumed);
/* LZMA.decompress(data:) LZMA.swift:25-34 (13-byte .lzma header) */
int swc_lzma_alone_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);
/* LZMA2.decompress(_:_:) LZMA2.swift:32-36 (callers XZBlock.swift:51, 7zFolder.swift:159) */
int swc_lzma2_decompress(const uint8_t* in, size_t in_len, uint8_t dict_byte, uint8_t** out, size_t* out_len, size_t* in_consumed);
/* LZMA2.decompress(data:) LZMA2.swift:25-30 (first byte = dictionary size) */
int swc_lzma2_decompress_data(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);

/* LZ4.decompress(data:dictionary:dictionaryID:) LZ4.swift:73-91; dict NULL = no dictionary,
 * dict_id < 0 = no id passed; *out valid on SWC_E_DATA_CHECKSUM_MISMATCH */
int swc_lz4_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len, int64_t dict_id,
                       uint8_t** out, size_t* out_len, size_t* in_consumed);
/* LZ4.compress(data:independentBlocks:blockChecksums:contentChecksum:contentSize:blockSize:dictionary:dictionaryID:)
 * LZ4+Compress.swift:47-155 (LZ4.compress(data:) = independent 1, block checksums 0, content checksum 1, content size 0,
 * 4 MiB blocks, no dictionary).  Frame layout and fields as the reference writes them; the blocks are compressed on the
 * device, all of them in one launch when they are independent (SWC_CODEC_LZ4_COMPRESS).  dict NULL = none, dict_id < 0 =
 * none.  block_size 1..4194304 (the reference's precondition), else SWC_E_INVALID_ARGUMENT. */
int swc_lz4_compress(const uint8_t* data, size_t len, int independent_blocks, int block_checksums, int content_checksum,
                     int content_size, size_t block_size, const uint8_t* dict, size_t dict_len, int64_t dict_id,
                     uint8_t** out, size_t* out_len);
/* Deflate.compress(data:) Deflate+Compress.swift:22-46: one stored or static-Huffman block, compressed on the device
 * (SWC_CODEC_DEFLATE_COMPRESS; one wavefront per buffer of up to 1 MiB; a larger one is cut into segments of 256 KiB that
 * are compressed in one launch and come out as non-final static blocks joined by empty stored blocks -- one Deflate stream).  ZlibArchive.archive(data:) ZlibArchive.swift:54-70: 0x78 0xDA, that stream, Adler-32 of the
 * input (big endian; computed on the host). */
int swc_deflate_compress(const uint8_t* data, size_t len, uint8_t** out, size_t* out_len);
int swc_zlib_archive(const uint8_t* data, size_t len, uint8_t** out, size_t* out_len);
/* GzipArchive.archive(data:comment:fileName:writeHeaderCRC:isTextFile:osType:modificationTime:extraFields:)
 * GzipArchive.swift:126-240: the header as the reference writes it (magic, CM 8, flags, MTIME, XFL 2, OS, FEXTRA, FNAME,
 * FCOMMENT, FHCRC), Deflate.compress(data) on the device, CRC-32 and ISIZE.  comment / file_name: ISO Latin-1 bytes (NULL =
 * none; the terminating zero is added unless the last byte is one, :134-136 / :147-149 -- the String conversion and its
 * cannotEncodeISOLatin1 are the shim's); os_type: the header byte (FileSystemType+Gzip.swift:23-36, 255 = unknown);
 * has_mtime 0 = four zero bytes, else the low four bytes of `mtime` seconds (:172-178).  Extra fields that sum up (4 + length
 * each) to more than 65,535 bytes: SWC_E_GZIP_CANNOT_ENCODE_ISO_LATIN1, as the reference throws (:190-191). */
typedef struct swc_gzip_extra_field {
    uint8_t si1, si2;
    const uint8_t* bytes;
    size_t len;
} swc_gzip_extra_field;
int swc_gzip_archive(const uint8_t* data, size_t len, const uint8_t* comment, size_t comment_len, const uint8_t* file_name,
                     size_t file_name_len, int write_header_crc, int is_text_file, int os_type, int has_mtime, int64_t mtime,
                     const swc_gzip_extra_field* extra, size_t n_extra, uint8_t** out, size_t* out_len);
/* BZip2.compress(data:blockSize:) BZip2+Compress.swift:40-74 (BZip2.compress(data:) :19-21 = block_size 1).  block_size 1..9 =
 * BlockSize.one ... .nine (else SWC_E_INVALID_ARGUMENT); the input is cut into blocks of block_size x 80,000 bytes as the
 * reference cuts it (:46), all blocks are compressed on the device together: initial run-length coding, Burrows-Wheeler
 * transform (prefix doubling over all blocks at once), move-to-front and zero-run coding, Huffman coding with up to six
 * tables per block, one per group of 50 symbols, refined in four passes over all groups (bzip2's scheme).  The stream decodes
 * to `data` with the reference's decoder and with libbz2; its bytes are not the reference encoder's (which builds each of its
 * tables from a single group of 50 symbols). */
int swc_bzip2_compress(const uint8_t* data, size_t len, int block_size, uint8_t** out, size_t* out_len);
/* LZ4.multiDecompress(data:dictionary:dictionaryID:) LZ4.swift:116-146 */
int swc_lz4_multi_decompress(const uint8_t* in, size_t in_len, const uint8_t* dict, size_t dict_len, int64_t dict_id,
                             uint8_t** out, size_t* out_len, size_t** sizes, size_t* n_frames);

/* Host-side framing around the batch API (reference L3 "archives") */
/* GzipArchive.unarchive(archive:) GzipArchive.swift:38-48 */
int swc_gzip_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);
/* GzipArchive.multiUnarchive(archive:) GzipArchive.swift:62-77; sizes[] splits *out per member */
int swc_gzip_multi_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t** sizes, size_t* n_members);
/* ZlibArchive.unarchive(archive:) ZlibArchive.swift:25-42 */
int swc_zlib_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);
/* XZArchive.unarchive(archive:) XZArchive.swift:27-51 / splitUnarchive :69-88 */
int swc_xz_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len);
int swc_xz_split_unarchive(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len, size_t** sizes, size_t* n_streams);

/* Many independent archives in one call: host-side block discovery + ONE batched launch.
 * archives[i]/lens[i] are host buffers; outs[i]/out_lens[i]/statuses[i] are filled per archive
 * (outs[i] released with swc_free).  kind: 1 = gzip member (incl. BGZF), 2 = zlib stream,
 * 3 = raw deflate, 4 = LZ4 frame, 5 = bzip2 stream, 6 = xz stream, 7 = raw LZMA2 (first byte = dict byte). */
int swc_unarchive_many(int kind, const uint8_t* const* archives, const size_t* lens, size_t n,
                       uint8_t** outs, size_t* out_lens, int32_t* statuses);
/* The same over several GPUs of one node (the reference has no counterpart: its callers loop over archives on one core).
 * devices[0..n_devices) are HIP device ordinals (a device may be listed more than once); the archive list is cut into one
 * contiguous range per entry, balanced by compressed + uncompressed bytes (the uncompressed size where the framing declares
 * it -- gzip ISIZE, LZ4 content size -- else the compressed size again), and every range is decoded on its device by that
 * device's worker thread (one per device for the life of the process: its stream and page-locked staging buffers are set up
 * once).  The archives are independent, so there is no exchange between devices; results land at their archive's index
 * exactly as swc_unarchive_many returns them.  SWC_E_DEVICE if an ordinal does not name a gfx950 device (the calling
 * thread's own current device does not matter). */
int swc_unarchive_many_devices(int kind, const uint8_t* const* archives, const size_t* lens, size_t n,
                               const int* devices, size_t n_devices, uint8_t** outs, size_t* out_lens, int32_t* statuses);

/* Block discovery on the host, as a library call (no device needed) -- what swc_unarchive_many / the multi-member entry
 * points use internally, for callers that stage their data on the device themselves:
 *   kind 1  BGZF: every gzip member that carries the 'BC' extra field (GzipHeader.swift:110-156): offset / comp_len of its
 *           Deflate stream, uncomp_len = ISIZE.  SWC_E_INVALID_ARGUMENT if a member lacks the field.
 *   kind 4  LZ4 frame (LZ4.swift:278-299): the blocks of the frame the buffer opens with; aux = 1 for stored blocks.
 *   kind 5  bzip2: every BIT offset of the 48-bit block magic (BZip2.swift:74-88) -- candidates, a magic may occur in data.
 *   kind 6  xz: the LZMA2-only blocks listed by the index of every stream (XZArchive.swift:132-192 read backwards):
 *           offset / comp_len of the LZMA2 data, uncomp_len, aux = dictionary-size byte.
 *   kind 7  raw LZMA2 (first byte = dictionary-size byte): the chunk walk of LZMA2Decoder.decode() / dispatch()
 *           (LZMA2Decoder.swift:36-74) without the LZMA decode: one ref per chunk, offset of its control byte, comp_len =
 *           header + payload, uncomp_len, aux = control byte, flags bit 0 = the chunk resets the dictionary (a run of
 *           chunks that decodes on its own starts here), bit 1 = it carries a properties byte.  The end marker is not
 *           listed.  Returns SWC_E_LZMA2_WRONG_CONTROL_BYTE (:47-48) or SWC_E_REF_TRAP (a chunk runs past the buffer)
 *           with the chunks before the error in refs / *n.
 * *n receives the number found; up to `cap` are written. */
typedef struct swc_block_ref {
    uint64_t offset;      /* bytes from `in` (kind 5: bits) */
    uint64_t comp_len;
    uint64_t uncomp_len;  /* 0 = not known from the framing */
    uint32_t aux;
    uint32_t flags;       /* kind 7 only, else 0 */
} swc_block_ref;
int swc_index_blocks(int kind, const uint8_t* in, size_t len, swc_block_ref* refs, size_t cap, size_t* n);

/* ZipContainer.getEntryData (reference Sources/ZIP/ZipContainer.swift:61-118) for all entries of one container: every
 * entry is an independent stream whose location and sizes the caller already has from the central directory
 * (ZipEntryInfoHelper.swift:22-44), so the Deflate (8) and LZMA (14) entries go to the device as ONE batch each; stored (0)
 * entries are copied, BZip2 (12) entries run their own block discovery.  Per entry: status (DeflateError / BZip2Error /
 * LZMAError / SWC_E_ZIP_WRONG_SIZE / SWC_E_ZIP_COMPRESSION_NOT_SUPPORTED), crc_error (the caller raises
 * ZipError.wrongCRC(entries so far), ZipContainer.swift:52-53) and the data (malloc()ed, swc_free). */
typedef struct swc_zip_entry {
    uint64_t data_offset;        /* IN : helper.dataOffset                                             */
    uint64_t comp_size;          /* IN : helper.compSize                                               */
    uint64_t uncomp_size;        /* IN : helper.uncompSize                                             */
    uint32_t crc32;              /* IN : helper.entryInfo.crc                                          */
    uint16_t method;             /* IN : 0 copy, 8 deflate, 12 bzip2, 14 lzma (CompressionMethod+Zip.swift:9-21) */
    uint8_t has_data_descriptor; /* IN : helper.hasDataDescriptor                                      */
    uint8_t zip64;               /* IN : helper.zip64FieldsArePresent                                  */
    int32_t status;              /* OUT                                                                */
    uint8_t crc_error;           /* OUT                                                                */
    uint8_t pad[3];
    uint8_t* data;               /* OUT: entry data (empty on error)                                   */
    size_t data_len;             /* OUT                                                                */
} swc_zip_entry;
int swc_zip_get_entries_data(const uint8_t* container, size_t len, swc_zip_entry* entries, size_t n);

/* SevenZipFolder.unpack(data:) (reference Sources/7-Zip/7zFolder.swift:138-194) for many folders at once.  The caller has
 * parsed the archive header and hands over, per folder, its packed stream and the ORDERED coder chain (orderedCoders(),
 * :88-97) with each coder's unpack size (unpackSize(for:), :129-136).  Folders are independent: stage k of all chains
 * runs as one batched launch per codec.  method: 0 copy, 1 deflate, 2 bzip2, 3 LZMA2, 4 LZMA, 5 Delta (id 03),
 * 6 LZ4 (id 04 F7 11 04), 7 an encryption method, 8 anything else.  Per folder: status (the codec's error,
 * SWC_E_7Z_*), data (malloc()ed, swc_free; empty on error). */
typedef struct swc_7z_coder {
    uint32_t method;
    uint8_t props[5];        /* coder.properties (LZMA2: 1 byte, LZMA: 5, Delta: 1)                          */
    uint8_t props_len;       /* 0xFF = properties absent                                                     */
    uint8_t multi_stream;    /* !(numInStreams == 1 || numOutStreams == 1), 7zFolder.swift:141               */
    uint8_t pad;
    uint64_t unpack_size;
} swc_7z_coder;
typedef struct swc_7z_folder {
    const uint8_t* data;     /* IN : packed stream of the folder                                             */
    size_t len;
    const swc_7z_coder* coders;
    size_t n_coders;
    int32_t status;          /* OUT                                                                          */
    int32_t pad;
    uint8_t* out;            /* OUT                                                                          */
    size_t out_len;
} swc_7z_folder;
int swc_7z_unpack_folders(swc_7z_folder* folders, size_t n);

/* checksums used by the framing layer (CheckSums.swift:12-57, XxHash32.swift:24-83, Sha256.swift:28-142) */
uint32_t swc_crc32(const uint8_t* p, size_t n, uint32_t prev);
uint32_t swc_adler32(const uint8_t* p, size_t n);
uint64_t swc_crc64(const uint8_t* p, size_t n);
uint32_t swc_bzip2_crc32(const uint8_t* p, size_t n);
uint32_t swc_xxh32(const uint8_t* p, size_t n, uint32_t seed);
void swc_sha256(const uint8_t* p, size_t n, uint8_t digest[32]);

void swc_free(void* p);
/* What the library keeps between calls -- freed device memory in the device's pool (4 GiB), the calling thread's two page-locked
 * staging buffers (512 MiB each), large host results handed back through swc_free() (512 MiB in all) -- goes back to the
 * driver / the system: the parked results, the CALLING thread's staging buffers, the current device's pool.  The limits:
 * swc_set_tuning "pool_keep_mib" / "pinned_keep_mib" / "result_cache_mib" (or the environment variables SWC_POOL_KEEP_MIB /
 * SWC_PINNED_KEEP_MIB / SWC_RESULT_CACHE_MIB, read when the library is loaded).  Returns SWC_OK. */
int swc_trim(void);
/* 1 if a gfx950 device is usable, 0 otherwise (then every decode entry point returns SWC_E_DEVICE) */
int swc_device_available(void);
const char* swc_version(void);
/* Knobs that never change results.  Memory kept between calls (see swc_trim): "pool_keep_mib", "pinned_keep_mib",
 * "result_cache_mib" = MiB (>= 0).  Measurement knobs, meant for benchmarking:
 *   "phase_timing" = 0 | 1      HIP events between the kernels of the batch launches of the CALLING THREAD (like the launch
 *                               stream and the staging buffers, measurement state is per thread);
 *   "lzma_coder_cache" = 1 | 0  process-wide: LZMA / LZMA2 launches with a workspace keep four LINES (a third of a literal
 *                               coder each) in LDS as a cache of the coders in the workspace (32 streams per CU, default) or
 *                               all coders of lc + lp <= 3 in LDS (10 streams per CU);
 *   "lz_copier" = 1 | 0 | 2 | -1 | -2   process-wide: the LZ77 copy phase of Deflate / LZ4 launches -- 1 (default): one stream
 *                               per wave (csrc/lz_copy.h: Deflate a 6 KiB LDS window in groups of up to 1 KiB, LZ4 9 KiB / 2 KiB)
 *                               for launches of 2,560 streams and more, one stream per 512-thread workgroup
 *                               (csrc/lz_resolve.h) below; 0: the workgroup kernel always; 2: Deflate launches take the wave
 *                               kernel with a 16 KiB window; -1 / -2: the wave kernel whatever the launch size;
 *   "deflate_team" = 1 | 0 | -1 process-wide: Deflate launches of up to 256 streams give every stream a workgroup of six wavefronts
 *                               (csrc/inflate_sync.h: the master on the job, the helpers on the rounds behind the master's --
 *                               the latency of ONE stream is a wave's, and a team cuts it by three) (1, default), one wavefront
 *                               per stream whatever the launch (0), or a team up to 4,096 streams (-1: tests);
 *   "bzip2_hot_cxx" = 0 | 1     process-wide: BZip2 launches run the instantiation of the block kernel whose plain-symbol loop is
 *                               compiled from C++ (1) instead of the hand-written assembly (0, default) -- the two are compared
 *                               by the GPU tests;
 *   "bzip2_team_walk" = 1 | 0 | 2   process-wide: BZip2 launches run stage 3 -- the inverse Burrows-Wheeler walk, the lay-out,
 *                               the RLE1 undo -- as kernels of their own that walk out of the XCDs' L2 (csrc/bzip2_team.h)
 *                               unless the launch is tiny (1, default), never (0: one wavefront takes a block through all
 *                               stages), or always (2);
 *   "bzip2_team_per_cu" = 1 | 2 process-wide: workgroups of the team walk per CU (1, default; 2 for comparison runs). */
int swc_set_tuning(const char* key, int value);
/* Profile builds of the library (-DSWC_PROFILE) only: a device buffer of 32 x uint64 per job of the next Deflate
 * launches that the kernels fill with cycle counts per stage (tools/exp_profile.py).  NULL switches it off.  A no-op in
 * the shipped build. */
int swc_set_profile_buffer(void* device_ptr);
/* With "phase_timing" on: durations (ms) of the kernels of the calling thread's last batch launch, in launch order --
 * Deflate: entropy decode, LZ77 resolve; LZ4: dictionary-block kernel, parse, resolve; BZip2: block kernel (Huffman + MTF,
 * counting sort, walk), serial fallback, block CRC; LZMA / LZMA2: the one kernel.  Returns the number of values written
 * (0 if none or if `cap` is too small; at most 4). */
int swc_last_phase_ms(float* ms, int cap);

/* Process-wide launch statistics of the host framing layer (monotonic, for tests and tuning): "launches" = batched
 * launches issued by the single-shot / many-archive entry points, "units" = units decoded by them, "xz_cache_hits" =
 * .xz blocks the sequential walk took from the index-driven ahead-of-time batch.  Unknown key: -1. */
long long swc_stat(const char* key);

#ifdef __cplusplus
}
#endif
#endif /* SWC_HIP_H */
