"""ctypes binding of the CPU oracle (oracle/librefcpu.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package `swcompression_amd` never does.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "librefcpu.so")


def build():
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR], check=True)


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(_LIB_PATH)
    u8p = C.POINTER(C.c_uint8)
    szp = C.POINTER(C.c_size_t)
    lib.refcpu_free.argtypes = [C.c_void_p]
    lib.refcpu_set_max_output.argtypes = [C.c_size_t]
    lib.refcpu_crc32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
    lib.refcpu_crc32.restype = C.c_uint32
    lib.refcpu_bzip2crc32.argtypes = [C.c_char_p, C.c_size_t]
    lib.refcpu_bzip2crc32.restype = C.c_uint32
    lib.refcpu_crc64.argtypes = [C.c_char_p, C.c_size_t]
    lib.refcpu_crc64.restype = C.c_uint64
    lib.refcpu_adler32.argtypes = [C.c_char_p, C.c_size_t]
    lib.refcpu_adler32.restype = C.c_uint32
    lib.refcpu_xxh32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
    lib.refcpu_xxh32.restype = C.c_uint32
    lib.refcpu_sha256.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    for name in ("deflate_decompress", "bzip2_decompress"):
        getattr(lib, "refcpu_" + name).argtypes = [C.c_char_p, C.c_size_t, C.POINTER(u8p), szp, szp]
    for name in ("gzip_unarchive", "zlib_unarchive", "lzma_alone_decompress", "lzma2_decompress_data", "xz_unarchive"):
        getattr(lib, "refcpu_" + name).argtypes = [C.c_char_p, C.c_size_t, C.POINTER(u8p), szp]
    for name in ("gzip_multi_unarchive", "bzip2_multi_decompress", "xz_split_unarchive"):
        getattr(lib, "refcpu_" + name).argtypes = [C.c_char_p, C.c_size_t, C.POINTER(u8p), szp, C.POINTER(szp), szp]
    lib.refcpu_lzma_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                           C.POINTER(u8p), szp, szp]
    lib.refcpu_lzma2_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_uint8, C.POINTER(u8p), szp, szp]
    lib.refcpu_lz4_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int64,
                                          C.POINTER(u8p), szp, szp]
    lib.refcpu_lz4_multi_decompress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int64,
                                                C.POINTER(u8p), szp, C.POINTER(szp), szp]
    lib.refcpu_lz4_block.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(u8p), szp]
    return lib


lib = _load()


def _take(ptr, n):
    data = C.string_at(ptr, n) if n else b""
    lib.refcpu_free(ptr)
    return data


def _take_sizes(ptr, n):
    sizes = [ptr[i] for i in range(n)]
    lib.refcpu_free(ptr)
    return sizes


def _split(data, sizes):
    out, o = [], 0
    for s in sizes:
        out.append(data[o:o + s])
        o += s
    return out


def _simple(fn, data):
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    st = fn(data, len(data), C.byref(out), C.byref(n))
    return st, _take(out, n.value)


def _consuming(fn, data, *mid):
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    consumed = C.c_size_t()
    st = fn(data, len(data), *mid, C.byref(out), C.byref(n), C.byref(consumed))
    return st, _take(out, n.value), consumed.value


def _multi(fn, data, *mid):
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    sizes = C.POINTER(C.c_size_t)()
    cnt = C.c_size_t()
    st = fn(data, len(data), *mid, C.byref(out), C.byref(n), C.byref(sizes), C.byref(cnt))
    blob = _take(out, n.value)
    return st, _split(blob, _take_sizes(sizes, cnt.value))


def deflate(data):
    """-> (status, output, in_consumed)"""
    return _consuming(lib.refcpu_deflate_decompress, bytes(data))


def gzip_unarchive(data):
    return _simple(lib.refcpu_gzip_unarchive, bytes(data))


def gzip_multi_unarchive(data):
    return _multi(lib.refcpu_gzip_multi_unarchive, bytes(data))


def zlib_unarchive(data):
    return _simple(lib.refcpu_zlib_unarchive, bytes(data))


def bzip2(data):
    return _consuming(lib.refcpu_bzip2_decompress, bytes(data))


def bzip2_multi(data):
    return _multi(lib.refcpu_bzip2_multi_decompress, bytes(data))


def lzma_raw(data, lc=3, lp=0, pb=2, dict_size=1 << 24, uncompressed_size=-1):
    return _consuming(lib.refcpu_lzma_decompress, bytes(data), lc, lp, pb, dict_size, uncompressed_size)


def lzma_alone(data):
    return _simple(lib.refcpu_lzma_alone_decompress, bytes(data))


def lzma2(data, dict_byte):
    return _consuming(lib.refcpu_lzma2_decompress, bytes(data), dict_byte)


def lzma2_data(data):
    return _simple(lib.refcpu_lzma2_decompress_data, bytes(data))


def xz_unarchive(data):
    return _simple(lib.refcpu_xz_unarchive, bytes(data))


def xz_split_unarchive(data):
    return _multi(lib.refcpu_xz_split_unarchive, bytes(data))


def lz4(data, dictionary=None, dict_id=-1):
    d = None if dictionary is None else bytes(dictionary)
    if d is not None and len(d) == 0:
        d = C.create_string_buffer(1).raw  # non-NULL pointer with length 0 == empty dictionary
        return _consuming(lib.refcpu_lz4_decompress, bytes(data), d, 0, dict_id)
    return _consuming(lib.refcpu_lz4_decompress, bytes(data), d, 0 if d is None else len(d), dict_id)


def lz4_multi(data, dictionary=None, dict_id=-1):
    d = None if dictionary is None else bytes(dictionary)
    return _multi(lib.refcpu_lz4_multi_decompress, bytes(data), d, 0 if d is None else len(d), dict_id)


def lz4_block(data, dictionary=None):
    d = None if dictionary is None else bytes(dictionary)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    st = lib.refcpu_lz4_block(bytes(data), len(data), d, 0 if d is None else len(d), C.byref(out), C.byref(n))
    return st, _take(out, n.value)


def lz4_compress_block(block, prefix=b""):
    """LZ4.compress(block:_:) (LZ4+Compress.swift:157-281), restated: (status, compressed bytes)."""
    buf = bytes(prefix) + bytes(block)
    cap = len(block) + len(block) // 255 + 16
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_size_t()
    lib.refcpu_lz4_compress_block.restype = C.c_int
    lib.refcpu_lz4_compress_block.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    st = lib.refcpu_lz4_compress_block(buf, len(buf), len(prefix), out, cap, C.byref(n))
    return st, out.raw[:min(n.value, cap)]


def deflate_compress(data):
    """Deflate.compress(data:) (Deflate+Compress.swift:22-213), restated: the compressed bytes."""
    data = bytes(data)
    cap = len(data) + len(data) // 8 + 64
    out = C.create_string_buffer(cap)
    n = C.c_size_t()
    lib.refcpu_deflate_compress.restype = C.c_int
    lib.refcpu_deflate_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    st = lib.refcpu_deflate_compress(data, len(data), out, cap, C.byref(n))
    assert st == 0, st
    return out.raw[:n.value]


def bzip2_compress(data, level=1):
    """BZip2.compress(data:blockSize:) (BZip2+Compress.swift:41-325), restated: the compressed bytes."""
    data = bytes(data)
    cap = len(data) + len(data) // 2 + 4096
    out = C.create_string_buffer(cap)
    n = C.c_size_t()
    lib.refcpu_bzip2_compress.restype = C.c_int
    lib.refcpu_bzip2_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    st = lib.refcpu_bzip2_compress(data, len(data), level, out, cap, C.byref(n))
    assert st == 0, st
    return out.raw[:n.value]


def crc32(data, prev=0):
    return lib.refcpu_crc32(bytes(data), len(data), prev)


def bzip2crc32(data):
    return lib.refcpu_bzip2crc32(bytes(data), len(data))


def crc64(data):
    return lib.refcpu_crc64(bytes(data), len(data))


def adler32(data):
    return lib.refcpu_adler32(bytes(data), len(data))


def xxh32(data, seed=0):
    return lib.refcpu_xxh32(bytes(data), len(data), seed)


def sha256(data):
    buf = C.create_string_buffer(32)
    lib.refcpu_sha256(bytes(data), len(data), buf)
    return buf.raw


lib.refcpu_zip_get_entry_data.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int,
                                          C.c_int, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
lib.refcpu_zip_get_entry_data.restype = C.c_int


def zip_entry(container, h):
    """ZipContainer.getEntryData for one entry helper (dict as produced by swcompression_amd.zipcontainer.ZipContainer.helpers).
    Returns (status, crc_error, data)."""
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    ce = C.c_int()
    st = lib.refcpu_zip_get_entry_data(bytes(container), len(container), h["data_offset"], h["comp_size"], h["uncomp_size"], h["crc32"],
                                       h["method"], int(h["has_data_descriptor"]), int(h["zip64"]), C.byref(out), C.byref(n), C.byref(ce))
    return st, bool(ce.value), _take(out, n.value)


def sevenzip_folder(data, chain):
    """SevenZipFolder.unpack(data:) for one folder; chain as for swcompression_amd.sevenzip.coder_array.  -> (status, bytes)"""
    from swcompression_amd.sevenzip import coder_array
    ca = coder_array(chain)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    lib.refcpu_7z_unpack_folder.restype = C.c_int
    lib.refcpu_7z_unpack_folder.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    st = lib.refcpu_7z_unpack_folder(bytes(data), len(data), ca, len(chain), C.byref(out), C.byref(n))
    return st, _take(out, n.value)
