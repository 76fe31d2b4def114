/*
 * swc_status.h -- status codes shared by the MI355X engine (libswc_hip.so) and the CPU oracle.
 *
 * Every non-zero code maps 1:1 onto one case of a Swift error enum of the reference
 * (tsolomko/SWCompression 4.9.0); file:line of the enum case is cited next to each code.
 * Three codes have no Swift counterpart:
 *   SWC_E_REF_TRAP  - input on which the reference hits a Swift runtime trap (array index out of
 *                     range, reading past the end of a BitByteData reader, ...). The reference
 *                     would abort the process; we report it. Parity tests treat it as a class.
 *   SWC_E_CAPACITY  - caller-provided output capacity too small (batch API); from the single-shot calls: a unit beyond the
 *                     engine's limits -- output above 16 GiB (bzip2: 1 GiB per block), or a bzip2 block whose BWT column
 *                     exceeds 16,000,000 bytes (17 x the largest block an encoder writes; the reference enforces no
 *                     block size).
 *   SWC_E_DEVICE    - HIP runtime failure (no device, launch error). Never a CPU fallback.
 */
#ifndef SWC_STATUS_H
#define SWC_STATUS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum swc_status {
    SWC_OK = 0,

    /* DeflateError -- Sources/Deflate/DeflateError.swift:10-19 */
    SWC_E_DEFLATE_WRONG_UNCOMPRESSED_BLOCK_LENGTHS = 101, /* :12 */
    SWC_E_DEFLATE_WRONG_BLOCK_TYPE = 102,                 /* :14 */
    SWC_E_DEFLATE_WRONG_SYMBOL = 103,                     /* :16 */
    SWC_E_DEFLATE_SYMBOL_NOT_FOUND = 104,                 /* :18 */

    /* BZip2Error -- Sources/BZip2/BZip2Error.swift:12-44 */
    SWC_E_BZIP2_WRONG_MAGIC = 201,               /* :14 */
    SWC_E_BZIP2_WRONG_VERSION = 202,             /* :16 */
    SWC_E_BZIP2_WRONG_BLOCK_SIZE = 203,          /* :18 */
    SWC_E_BZIP2_WRONG_BLOCK_TYPE = 204,          /* :20 */
    SWC_E_BZIP2_RANDOMIZED_BLOCK = 205,          /* :22 */
    SWC_E_BZIP2_WRONG_HUFFMAN_GROUPS = 206,      /* :24 */
    SWC_E_BZIP2_WRONG_SELECTOR = 207,            /* :26 */
    SWC_E_BZIP2_WRONG_HUFFMAN_CODE_LENGTH = 208, /* :28 */
    SWC_E_BZIP2_SYMBOL_NOT_FOUND = 209,          /* :30 */
    SWC_E_BZIP2_WRONG_CRC = 210,                 /* :43 carries output decoded so far */

    /* LZMAError -- Sources/LZMA/LZMAError.swift:10-25 */
    SWC_E_LZMA_WRONG_PROPERTIES = 301,           /* :12 */
    SWC_E_LZMA_RANGE_DECODER_INIT_ERROR = 302,   /* :14 */
    SWC_E_LZMA_EXCEEDED_UNCOMPRESSED_SIZE = 303, /* :16 */
    SWC_E_LZMA_WINDOW_IS_EMPTY = 304,            /* :18 */
    SWC_E_LZMA_RANGE_DECODER_FINISH_ERROR = 305, /* :20 */
    SWC_E_LZMA_REPEAT_WILL_EXCEED = 306,         /* :22 */
    SWC_E_LZMA_NOT_ENOUGH_TO_REPEAT = 307,       /* :24 */

    /* LZMA2Error -- Sources/LZMA2/LZMA2Error.swift:10-22 */
    SWC_E_LZMA2_WRONG_DICTIONARY_SIZE = 401, /* :12 */
    SWC_E_LZMA2_WRONG_CONTROL_BYTE = 402,    /* :14 */
    SWC_E_LZMA2_WRONG_RESET = 403,           /* :16 */
    SWC_E_LZMA2_WRONG_SIZES = 404,           /* :21 */

    /* DataError (LZ4) -- Sources/Common/DataError.swift:9-25 */
    SWC_E_DATA_TRUNCATED = 501,           /* :11 */
    SWC_E_DATA_CORRUPTED = 502,           /* :16 */
    SWC_E_DATA_CHECKSUM_MISMATCH = 503,   /* :22 carries output */
    SWC_E_DATA_UNSUPPORTED_FEATURE = 504, /* :24 */

    /* GzipError -- Sources/GZip/GzipError.swift:10-35 */
    SWC_E_GZIP_WRONG_MAGIC = 601,              /* :12 */
    SWC_E_GZIP_WRONG_COMPRESSION_METHOD = 602, /* :14 */
    SWC_E_GZIP_WRONG_FLAGS = 603,              /* :19 */
    SWC_E_GZIP_WRONG_HEADER_CRC = 604,         /* :21 */
    SWC_E_GZIP_WRONG_CRC = 605,                /* :30 carries members decoded so far */
    SWC_E_GZIP_WRONG_ISIZE = 606,              /* :32 */
    SWC_E_GZIP_CANNOT_ENCODE_ISO_LATIN1 = 607, /* :34 (GzipArchive.archive: a name or comment outside ISO Latin-1, extra fields beyond 65,535 bytes) */

    /* ZlibError -- Sources/Zlib/ZlibError.swift:12-26 */
    SWC_E_ZLIB_WRONG_COMPRESSION_METHOD = 701, /* :14 */
    SWC_E_ZLIB_WRONG_COMPRESSION_INFO = 702,   /* :16 */
    SWC_E_ZLIB_WRONG_FCHECK = 703,             /* :18 */
    SWC_E_ZLIB_WRONG_COMPRESSION_LEVEL = 704,  /* :20 */
    SWC_E_ZLIB_WRONG_ADLER32 = 705,            /* :25 carries output */

    /* XZError -- Sources/XZ/XZError.swift:12-48 */
    SWC_E_XZ_WRONG_MAGIC = 801,              /* :14 */
    SWC_E_XZ_WRONG_FIELD = 802,              /* :19 */
    SWC_E_XZ_WRONG_INFO_CRC = 803,           /* :21 */
    SWC_E_XZ_WRONG_FILTER_ID = 804,          /* :23 */
    SWC_E_XZ_CHECK_TYPE_SHA256 = 805,        /* :29 (unused by the decoder; kept for the enum) */
    SWC_E_XZ_WRONG_DATA_SIZE = 806,          /* :34 */
    SWC_E_XZ_WRONG_CHECK = 807,              /* :43 carries output */
    SWC_E_XZ_WRONG_PADDING = 808,            /* :45 */
    SWC_E_XZ_MULTI_BYTE_INTEGER_ERROR = 809, /* :47 */

    /* ZipError -- Sources/ZIP/ZipError.swift:10-36 (only the cases ZipContainer.getEntryData can produce) */
    SWC_E_ZIP_WRONG_SIZE = 851,                /* :16 */
    SWC_E_ZIP_COMPRESSION_NOT_SUPPORTED = 852, /* :26 */
    SWC_E_ZIP_WRONG_CRC = 853,                 /* :34 carries the entries processed so far (raised by the caller from crc_error) */

    /* SevenZipError -- Sources/7-Zip/7zError.swift:10-33 (only the cases SevenZipFolder.unpack can produce) */
    SWC_E_7Z_WRONG_SIZE = 861,                /* :18 */
    SWC_E_7Z_MULTI_STREAM_NOT_SUPPORTED = 862, /* :24 */
    SWC_E_7Z_COMPRESSION_NOT_SUPPORTED = 863, /* :28 */
    SWC_E_7Z_ENCRYPTION_NOT_SUPPORTED = 864,  /* :30 */
    SWC_E_7Z_INTERNAL_STRUCTURE_ERROR = 865,  /* :32 */

    SWC_E_REF_TRAP = 900,
    SWC_E_CAPACITY = 901,
    SWC_E_DEVICE = 902,
    SWC_E_INVALID_ARGUMENT = 903,
    SWC_E_NEED_WORKSPACE = 904 /* batch API: the unit needs the HBM workspace of swc_batch_workspace_bytes(); single-shot calls retry internally */
} swc_status;

#ifdef __cplusplus
}
#endif
#endif /* SWC_STATUS_H */
