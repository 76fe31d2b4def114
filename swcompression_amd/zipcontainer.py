"""ZipContainer.open for device-batched entry data.

The reference (Sources/ZIP/ZipContainer.swift:43-59) walks the central directory and then decodes entry after entry on
the CPU (getEntryData, :61-118).  Here the directory walk stays with the caller -- this mirror borrows the stdlib
`zipfile` reader for it, the Swift shim keeps its own ZipEntryInfoHelper -- and ALL entries go through
swc_zip_get_entries_data, which batches the Deflate and LZMA streams of the archive into one launch each."""
import ctypes as C
import io
import struct
import zipfile

from . import _lib, SWCError, ZipError, _raise

__all__ = ["ZipContainer", "SwcZipEntry"]


class SwcZipEntry(C.Structure):
    _fields_ = [("data_offset", C.c_uint64), ("comp_size", C.c_uint64), ("uncomp_size", C.c_uint64), ("crc32", C.c_uint32),
                ("method", C.c_uint16), ("has_data_descriptor", C.c_uint8), ("zip64", C.c_uint8), ("status", C.c_int32),
                ("crc_error", C.c_uint8), ("pad", C.c_uint8 * 3), ("data", C.POINTER(C.c_uint8)), ("data_len", C.c_size_t)]


class ZipContainer:
    @staticmethod
    def helpers(container):
        """What ZipEntryInfoHelper.init (ZipEntryInfoHelper.swift:22-44) hands to getEntryData, per entry."""
        container = bytes(container)
        out = []
        with zipfile.ZipFile(io.BytesIO(container)) as z:
            for info in z.infolist():
                ho = info.header_offset
                flags, method, _, _, lcrc, lcomp, luncomp, fn, ex = struct.unpack_from("<HHHHIIIHH", container, ho + 6)
                dd = bool(flags & 0x08)
                extra = container[ho + 30 + fn: ho + 30 + fn + ex]
                zip64 = False
                p = 0
                while p + 4 <= len(extra):
                    hid, hl = struct.unpack_from("<HH", extra, p)
                    zip64 |= hid == 0x0001
                    p += 4 + hl
                out.append(dict(name=info.filename, is_dir=info.is_dir(), data_offset=ho + 30 + fn + ex, method=method,
                                comp_size=info.compress_size if dd else lcomp, uncomp_size=info.file_size if dd else luncomp,
                                crc32=info.CRC, has_data_descriptor=dd, zip64=zip64))
        return out

    @staticmethod
    def entries_data(container, helpers):
        """swc_zip_get_entries_data for the given helpers.  Returns [(status, crc_error, bytes)]."""
        lib = _lib.load()
        container = bytes(container)
        n = len(helpers)
        arr = (SwcZipEntry * n)()
        for e, h in zip(arr, helpers):
            e.data_offset, e.comp_size, e.uncomp_size, e.crc32 = h["data_offset"], h["comp_size"], h["uncomp_size"], h["crc32"]
            e.method, e.has_data_descriptor, e.zip64 = h["method"], int(h["has_data_descriptor"]), int(h["zip64"])
        rc = lib.swc_zip_get_entries_data(container, len(container), arr, n)
        if rc:
            _raise(rc)
        res = []
        for e in arr:
            res.append((e.status, bool(e.crc_error), C.string_at(e.data, e.data_len) if e.data_len else b""))
            lib.swc_free(e.data)
        return res

    @staticmethod
    def open(container):
        """ZipContainer.open(container:) (ZipContainer.swift:43-59): [(name, data | None for directories)].  The first
        failing entry raises its error; a CRC mismatch raises ZipError.wrongCRC carrying the entries so far."""
        helpers = ZipContainer.helpers(container)
        files = [h for h in helpers if not h["is_dir"]]
        got = iter(ZipContainer.entries_data(container, files))
        entries = []
        for h in helpers:
            if h["is_dir"]:
                entries.append((h["name"], None))
                continue
            status, crc_error, data = next(got)
            if status:
                _raise(status)
            entries.append((h["name"], data))
            if crc_error:
                raise ZipError(853, entries)
        return entries
