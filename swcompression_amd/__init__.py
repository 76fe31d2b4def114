"""swcompression_amd -- Python binding of the MI355X-native decode engine (libswc_hip.so).

The names mirror the decode side of tsolomko/SWCompression's public API so that parity tests read like
the reference's own tests:

    Deflate.decompress(data)            Sources/Deflate/Deflate.swift:24
    GzipArchive.unarchive(archive)      Sources/GZip/GzipArchive.swift:38      (.multi_unarchive :62)
    ZlibArchive.unarchive(archive)      Sources/Zlib/ZlibArchive.swift:25
    BZip2.decompress(data)              Sources/BZip2/BZip2.swift:22           (.multi_decompress :40)
    LZMA.decompress(data[, properties]) Sources/LZMA/LZMA.swift:25,56
    LZMA2.decompress(data)              Sources/LZMA2/LZMA2.swift:25
    XZArchive.unarchive(archive)        Sources/XZ/XZArchive.swift:27          (.split_unarchive :69)
    LZ4.decompress(data[, dictionary])  Sources/LZ4/LZ4.swift:49,73            (.multi_decompress :116)

Everything decodes on the GPU through the C ABI in include/swc_hip.h.  There is no CPU fallback: on a
machine without a gfx950 device every call raises DeviceError.  Swift error enums map to exception
classes whose `.case` is the Swift case name; errors that carry decoded bytes in the reference
(`wrongCRC`, `wrongAdler32`, `wrongCheck`, `checksumMismatch`) expose them as `.data`.
"""
import ctypes as C

from . import _lib
from ._lib import SwcJob, SwcBatchOpts

__all__ = ["Deflate", "GzipArchive", "ZlibArchive", "BZip2", "LZMA", "LZMA2", "LZMAProperties", "XZArchive", "LZ4",
           "SWCError", "DeflateError", "GzipError", "ZlibError", "BZip2Error", "LZMAError", "LZMA2Error", "XZError",
           "DataError", "ZipError", "SevenZipError", "ReferenceTrap", "DeviceError", "device_available", "STATUS", "unarchive_many", "index_blocks"]

# status code -> (exception family, Swift case name); numeric values from include/swc_status.h
STATUS = {
    101: ("DeflateError", "wrongUncompressedBlockLengths"), 102: ("DeflateError", "wrongBlockType"),
    103: ("DeflateError", "wrongSymbol"), 104: ("DeflateError", "symbolNotFound"),
    201: ("BZip2Error", "wrongMagic"), 202: ("BZip2Error", "wrongVersion"), 203: ("BZip2Error", "wrongBlockSize"),
    204: ("BZip2Error", "wrongBlockType"), 205: ("BZip2Error", "randomizedBlock"), 206: ("BZip2Error", "wrongHuffmanGroups"),
    207: ("BZip2Error", "wrongSelector"), 208: ("BZip2Error", "wrongHuffmanCodeLength"), 209: ("BZip2Error", "symbolNotFound"),
    210: ("BZip2Error", "wrongCRC"),
    301: ("LZMAError", "wrongProperties"), 302: ("LZMAError", "rangeDecoderInitError"),
    303: ("LZMAError", "exceededUncompressedSize"), 304: ("LZMAError", "windowIsEmpty"),
    305: ("LZMAError", "rangeDecoderFinishError"), 306: ("LZMAError", "repeatWillExceed"), 307: ("LZMAError", "notEnoughToRepeat"),
    401: ("LZMA2Error", "wrongDictionarySize"), 402: ("LZMA2Error", "wrongControlByte"), 403: ("LZMA2Error", "wrongReset"),
    404: ("LZMA2Error", "wrongSizes"),
    501: ("DataError", "truncated"), 502: ("DataError", "corrupted"), 503: ("DataError", "checksumMismatch"),
    504: ("DataError", "unsupportedFeature"),
    601: ("GzipError", "wrongMagic"), 602: ("GzipError", "wrongCompressionMethod"), 603: ("GzipError", "wrongFlags"),
    604: ("GzipError", "wrongHeaderCRC"), 605: ("GzipError", "wrongCRC"), 606: ("GzipError", "wrongISize"), 607: ("GzipError", "cannotEncodeISOLatin1"),
    701: ("ZlibError", "wrongCompressionMethod"), 702: ("ZlibError", "wrongCompressionInfo"), 703: ("ZlibError", "wrongFcheck"),
    704: ("ZlibError", "wrongCompressionLevel"), 705: ("ZlibError", "wrongAdler32"),
    801: ("XZError", "wrongMagic"), 802: ("XZError", "wrongField"), 803: ("XZError", "wrongInfoCRC"), 804: ("XZError", "wrongFilterID"),
    805: ("XZError", "checkTypeSHA256"), 806: ("XZError", "wrongDataSize"), 807: ("XZError", "wrongCheck"),
    808: ("XZError", "wrongPadding"), 809: ("XZError", "multiByteIntegerError"),
    851: ("ZipError", "wrongSize"), 852: ("ZipError", "compressionNotSupported"), 853: ("ZipError", "wrongCRC"),
    861: ("SevenZipError", "wrongSize"), 862: ("SevenZipError", "multiStreamNotSupported"),
    863: ("SevenZipError", "compressionNotSupported"), 864: ("SevenZipError", "encryptionNotSupported"),
    865: ("SevenZipError", "internalStructureError"),
    900: ("ReferenceTrap", "trap"), 901: ("SWCError", "capacity"), 902: ("DeviceError", "device"),
    903: ("SWCError", "invalidArgument"), 904: ("SWCError", "needWorkspace"),
}


class SWCError(Exception):
    def __init__(self, status, data=None):
        family, case = STATUS.get(status, ("SWCError", "unknown"))
        super().__init__("%s.%s (status %d)" % (family, case, status))
        self.status = status
        self.case = case
        self.data = data  # bytes / list[bytes] carried by the Swift error, else None


class DeflateError(SWCError): pass
class GzipError(SWCError): pass
class ZlibError(SWCError): pass
class BZip2Error(SWCError): pass
class LZMAError(SWCError): pass
class LZMA2Error(SWCError): pass
class XZError(SWCError): pass
class DataError(SWCError): pass
class ZipError(SWCError): pass
class SevenZipError(SWCError): pass
class ReferenceTrap(SWCError):
    """Input on which the Swift reference would hit a runtime trap (abort)."""
class DeviceError(SWCError):
    """No usable gfx950 device / HIP failure.  There is no CPU fallback."""


_FAMILIES = {c.__name__: c for c in (SWCError, DeflateError, GzipError, ZlibError, BZip2Error, LZMAError, LZMA2Error,
                                     XZError, DataError, ZipError, SevenZipError, ReferenceTrap, DeviceError)}
_CARRIES_DATA = {210, 503, 605, 705, 807, 853}


def _raise(status, data=None):
    family = STATUS.get(status, ("SWCError", ""))[0]
    raise _FAMILIES[family](status, data if status in _CARRIES_DATA else None)


def device_available():
    return bool(_lib.load().swc_device_available())


def trim():
    """swc_trim(): what the library keeps between calls (freed device memory in the pool, this thread's page-locked staging
    buffers, parked host results) goes back to the driver / the system."""
    return _lib.load().swc_trim()


def _take(ptr, n):
    lib = _lib.load()
    data = C.string_at(ptr, n) if n else b""
    lib.swc_free(ptr)
    return data


def _take_sizes(ptr, n):
    lib = _lib.load()
    v = [ptr[i] for i in range(n)]
    lib.swc_free(ptr)
    return v


def _split(blob, sizes):
    out, o = [], 0
    for s in sizes:
        out.append(blob[o:o + s])
        o += s
    return out


def _call_simple(name, data, *mid, consumed=False, view=False):
    """view=True: the result is a read-only memoryview OF the C result instead of a bytes object copied from it (what a Swift
    shim does with Data(bytesNoCopy:count:deallocator:)); the buffer is released when the view dies."""
    lib = _lib.load()
    data = bytes(data)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    cons = C.c_size_t()
    args = [data, len(data)] + list(mid) + [C.byref(out), C.byref(n)]
    if consumed:
        args.append(C.byref(cons))
    st = getattr(lib, name)(*args)
    if view and not st and n.value:
        arr = (C.c_ubyte * n.value).from_address(C.cast(out, C.c_void_p).value)
        arr._owner = _Owner(out)
        mv = memoryview(arr).cast("B").toreadonly()
        return (mv, cons.value) if consumed else mv
    blob = _take(out, n.value)
    if st:
        _raise(st, blob)
    return (blob, cons.value) if consumed else blob


class _Owner:
    """Keeps a C result alive for the memoryviews cut from it; swc_free when the last of them is gone."""
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            _lib.load().swc_free(self.ptr)
        except Exception:
            pass


def _call_multi(name, data, *mid, views=False):
    """views=True: the parts are read-only memoryviews INTO the C result instead of bytes objects copied from it -- what a
    Swift shim does with Data(bytesNoCopy:count:deallocator:); the buffer is released when the last view dies."""
    lib = _lib.load()
    data = bytes(data)
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    sizes = C.POINTER(C.c_size_t)()
    cnt = C.c_size_t()
    st = getattr(lib, name)(data, len(data), *mid, C.byref(out), C.byref(n), C.byref(sizes), C.byref(cnt))
    # every part is copied ONCE, from the C result to its bytes object (a Swift shim wraps the buffer without a copy:
    # Data(bytesNoCopy:count:deallocator:))
    szs = _take_sizes(sizes, cnt.value)
    base = C.cast(out, C.c_void_p).value or 0
    parts, o = [], 0
    if views and not st and n.value:
        arr = (C.c_ubyte * n.value).from_address(base)
        arr._owner = _Owner(out)
        mv = memoryview(arr).cast("B").toreadonly()
        for sz in szs:
            parts.append(mv[o:o + sz])
            o += sz
        return parts
    for sz in szs:
        parts.append(C.string_at(base + o, sz) if sz else b"")
        o += sz
    lib.swc_free(out)
    if st:
        _raise(st, parts)
    return parts


class Deflate:
    @staticmethod
    def decompress(data):
        return _call_simple("swc_deflate_decompress", data, consumed=True)[0]

    @staticmethod
    def decompress_consumed(data):
        """(output, bytes consumed) -- the reader-taking overload Deflate.swift:30."""
        return _call_simple("swc_deflate_decompress", data, consumed=True)

    @staticmethod
    def compress(data):
        """Deflate.compress(data:) (Deflate+Compress.swift:22-46): one stored or static-Huffman block, compressed on the device.
        The stream decodes to `data` with the reference's decoder; its bytes are not the reference encoder's."""
        return _call_simple("swc_deflate_compress", data)


class GzipArchive:
    @staticmethod
    def unarchive(archive):
        return _call_simple("swc_gzip_unarchive", archive)

    @staticmethod
    def multi_unarchive(archive, views=False):
        return _call_multi("swc_gzip_multi_unarchive", archive, views=views)

    @staticmethod
    def archive(data, comment=None, file_name=None, write_header_crc=False, is_text_file=False, os_type=None,
                modification_time=None, extra_fields=()):
        """GzipArchive.archive(data:comment:fileName:writeHeaderCRC:isTextFile:osType:modificationTime:extraFields:)
        (GzipArchive.swift:126-240).  os_type: the header byte (0 FAT, 3 Unix, 7 Macintosh, 11 NTFS, None = 255 unknown:
        FileSystemType+Gzip.swift:23-36); modification_time: seconds since 1970; extra_fields: (si1, si2, bytes) triples.
        The body is Deflate.compress(data) on the device."""
        def latin1(text):
            if text is None:
                return None, 0
            try:
                b = text.encode("latin-1") if isinstance(text, str) else bytes(text)
            except UnicodeEncodeError:
                _raise(607)                                             # GzipError.cannotEncodeISOLatin1 (:137-141, :150-154)
            return b, len(b)
        lib = _lib.load()
        data = bytes(data)
        cb, cn = latin1(comment)
        fb, fn = latin1(file_name)
        keep = [bytes(e[2]) for e in extra_fields]
        arr = (_lib.SwcGzipExtraField * max(len(keep), 1))()
        for k, e in enumerate(extra_fields):
            arr[k].si1, arr[k].si2 = int(e[0]), int(e[1])
            arr[k].bytes = C.cast(C.c_char_p(keep[k]), C.c_void_p)
            arr[k].len = len(keep[k])
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        if cb is not None and cn == 0:
            cb = C.create_string_buffer(1).raw                          # a non-NULL pointer: an empty comment is a comment
        if fb is not None and fn == 0:
            fb = C.create_string_buffer(1).raw
        st = lib.swc_gzip_archive(data, len(data), cb, cn, fb, fn, int(bool(write_header_crc)), int(bool(is_text_file)),
                                  255 if os_type is None else int(os_type), 0 if modification_time is None else 1,
                                  0 if modification_time is None else int(modification_time), C.cast(arr, C.c_void_p) if keep else None,
                                  len(keep), C.byref(out), C.byref(n))
        res = _take(out, n.value)
        if st:
            _raise(st, res)
        return res


class ZlibArchive:
    @staticmethod
    def unarchive(archive):
        return _call_simple("swc_zlib_unarchive", archive)

    @staticmethod
    def archive(data):
        """ZlibArchive.archive(data:) (ZlibArchive.swift:54-70)."""
        return _call_simple("swc_zlib_archive", data)


class BZip2:
    @staticmethod
    def decompress(data):
        return _call_simple("swc_bzip2_decompress", data, consumed=True)[0]

    @staticmethod
    def multi_decompress(data):
        return _call_multi("swc_bzip2_multi_decompress", data)

    @staticmethod
    def compress(data, block_size=1):
        """BZip2.compress(data:blockSize:) (BZip2+Compress.swift:40-74; block_size 1..9 = BlockSize.one ... .nine, the default
        is that of BZip2.compress(data:), :19-21).  All blocks are compressed on the device together.  The stream decodes to
        `data` with the reference's decoder and libbz2; its bytes are not the reference encoder's."""
        if not 1 <= int(block_size) <= 9:
            raise ValueError("block_size must be 1..9")
        return _call_simple("swc_bzip2_compress", data, int(block_size))


class LZMAProperties:
    """LZMAProperties.swift:9-48 (no validation, as in the reference)."""
    def __init__(self, lc=3, lp=0, pb=2, dictionary_size=1 << 24):
        self.lc, self.lp, self.pb, self.dictionary_size = lc, lp, pb, dictionary_size


class LZMA:
    @staticmethod
    def decompress(data, properties=None, uncompressed_size=None):
        if properties is None:
            return _call_simple("swc_lzma_alone_decompress", data)
        us = -1 if uncompressed_size is None else int(uncompressed_size)
        return _call_simple("swc_lzma_decompress", data, properties.lc, properties.lp, properties.pb,
                            properties.dictionary_size, us, consumed=True)[0]


class LZMA2:
    @staticmethod
    def decompress(data):
        return _call_simple("swc_lzma2_decompress_data", data)

    @staticmethod
    def decompress_raw(data, dict_byte):
        """(output, consumed) -- LZMA2.decompress(_:_:) LZMA2.swift:32."""
        return _call_simple("swc_lzma2_decompress", data, dict_byte, consumed=True)


class XZArchive:
    @staticmethod
    def unarchive(archive, view=False):
        return _call_simple("swc_xz_unarchive", archive, view=view)

    @staticmethod
    def split_unarchive(archive):
        return _call_multi("swc_xz_split_unarchive", archive)


class LZ4:
    @staticmethod
    def decompress(data, dictionary=None, dictionary_id=None):
        d = None if dictionary is None else bytes(dictionary)
        if d is not None and len(d) == 0:
            d = C.create_string_buffer(1).raw  # non-NULL pointer + length 0 = empty dictionary
            dl = 0
        else:
            dl = 0 if d is None else len(d)
        did = -1 if dictionary_id is None else int(dictionary_id)
        return _call_simple("swc_lz4_decompress", data, d, dl, did, consumed=True)[0]

    @staticmethod
    def compress(data, independent_blocks=True, block_checksums=False, content_checksum=True, content_size=False,
                 block_size=4 * 1024 * 1024, dictionary=None, dictionary_id=None):
        """LZ4.compress(data:independentBlocks:blockChecksums:contentChecksum:contentSize:blockSize:dictionary:dictionaryID:)
        (LZ4+Compress.swift:47-155; the defaults are those of LZ4.compress(data:), :16-19).  The blocks are compressed on the
        device; the frame decodes to `data` with the reference's decoder, the block bytes are not the reference encoder's."""
        lib = _lib.load()
        data = bytes(data)
        d = None if dictionary is None else bytes(dictionary)
        out = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        rc = lib.swc_lz4_compress(data, len(data), int(bool(independent_blocks)), int(bool(block_checksums)), int(bool(content_checksum)),
                                  int(bool(content_size)), int(block_size), d, 0 if d is None else len(d),
                                  -1 if dictionary_id is None else int(dictionary_id), C.byref(out), C.byref(n))
        res = _take(out, n.value)
        if rc:
            _raise(rc)
        return res

    @staticmethod
    def multi_decompress(data, dictionary=None, dictionary_id=None):
        d = None if dictionary is None else bytes(dictionary)
        did = -1 if dictionary_id is None else int(dictionary_id)
        return _call_multi("swc_lz4_multi_decompress", data, d, 0 if d is None else len(d), did)


def unarchive_many(kind, archives, devices=None):
    """Host-side discovery + ONE batched launch for many independent archives.
    kind: 'gzip' | 'zlib' | 'deflate' | 'lz4' | 'bzip2' | 'xz' | 'lzma2'.  Returns list of (status, bytes).
    devices: a list of device ordinals to spread the archives over (swc_unarchive_many_devices: one contiguous range of
    the list per device, balanced by compressed bytes, each on its own host thread); default: the current device."""
    kinds = {"gzip": 1, "zlib": 2, "deflate": 3, "lz4": 4, "bzip2": 5, "xz": 6, "lzma2": 7}
    lib = _lib.load()
    n = len(archives)
    bufs = [bytes(a) for a in archives]
    arr = (C.c_char_p * n)(*bufs)
    lens = (C.c_size_t * n)(*[len(b) for b in bufs])
    outs = (C.POINTER(C.c_uint8) * n)()
    out_lens = (C.c_size_t * n)()
    sts = (C.c_int32 * n)()
    if devices is None:
        rc = lib.swc_unarchive_many(kinds[kind], arr, lens, n, outs, out_lens, sts)
    else:
        devs = (C.c_int * len(devices))(*devices)
        rc = lib.swc_unarchive_many_devices(kinds[kind], arr, lens, n, devs, len(devices), outs, out_lens, sts)
    if rc:
        _raise(rc)
    return [(sts[i], _take(outs[i], out_lens[i])) for i in range(n)]


class SwcBlockRef(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("comp_len", C.c_uint64), ("uncomp_len", C.c_uint64), ("aux", C.c_uint32), ("flags", C.c_uint32)]


def index_blocks(kind, data):
    """Host block discovery (swc_index_blocks; no device needed).  kind: 'bgzf' | 'lz4' | 'bzip2' | 'xz' | 'lzma2'.
    Returns [(offset, comp_len, uncomp_len, aux)] -- for 'lzma2' (offset, comp_len, uncomp_len, control byte, flags);
    offsets are bytes from the start (bzip2: bits)."""
    kinds = {"bgzf": 1, "lz4": 4, "bzip2": 5, "xz": 6, "lzma2": 7}
    lib = _lib.load()
    data = bytes(data)
    n = C.c_size_t()
    st = lib.swc_index_blocks(kinds[kind], data, len(data), None, 0, C.byref(n))
    if st:
        _raise(st)
    refs = (SwcBlockRef * max(n.value, 1))()
    st = lib.swc_index_blocks(kinds[kind], data, len(data), refs, n.value, C.byref(n))
    if st:
        _raise(st)
    if kind == "lzma2":
        return [(r.offset, r.comp_len, r.uncomp_len, r.aux, r.flags) for r in refs[:n.value]]
    return [(r.offset, r.comp_len, r.uncomp_len, r.aux) for r in refs[:n.value]]
