"""SevenZipFolder.unpack for many folders in one call.

The reference (Sources/7-Zip/7zFolder.swift:138-194) applies a folder's coder chain on the CPU, folder after folder.  The
archive header (coders, bind pairs, unpack sizes) stays with the caller; swc_7z_unpack_folders takes the ordered chains and
runs stage k of all of them as one batched launch per codec."""
import ctypes as C

from . import _lib, _raise

__all__ = ["SevenZipFolder", "Swc7zCoder", "Swc7zFolder", "METHODS"]

METHODS = {"copy": 0, "deflate": 1, "bzip2": 2, "lzma2": 3, "lzma": 4, "delta": 5, "lz4": 6, "encryption": 7, "other": 8}


class Swc7zCoder(C.Structure):
    _fields_ = [("method", C.c_uint32), ("props", C.c_uint8 * 5), ("props_len", C.c_uint8), ("multi_stream", C.c_uint8),
                ("pad", C.c_uint8), ("unpack_size", C.c_uint64)]


class Swc7zFolder(C.Structure):
    _fields_ = [("data", C.c_char_p), ("len", C.c_size_t), ("coders", C.POINTER(Swc7zCoder)), ("n_coders", C.c_size_t),
                ("status", C.c_int32), ("pad", C.c_int32), ("out", C.POINTER(C.c_uint8)), ("out_len", C.c_size_t)]


def coder_array(chain):
    """chain: [(method name, properties bytes | None, unpack size[, multi_stream])] in orderedCoders() order."""
    arr = (Swc7zCoder * max(len(chain), 1))()
    for c, item in zip(arr, chain):
        name, props, size = item[:3]
        c.method = METHODS[name]
        c.props_len = 0xFF if props is None else len(props)
        for k, b in enumerate((props or b"")[:5]):
            c.props[k] = b
        c.multi_stream = 1 if len(item) > 3 and item[3] else 0
        c.unpack_size = size
    return arr


class SevenZipFolder:
    @staticmethod
    def unpack_many(folders):
        """folders: [(packed bytes, chain)].  Returns [(status, bytes)]."""
        lib = _lib.load()
        n = len(folders)
        arr = (Swc7zFolder * max(n, 1))()
        keep = []
        for f, (data, chain) in zip(arr, folders):
            data = bytes(data)
            ca = coder_array(chain)
            keep.append((data, ca))
            f.data, f.len, f.coders, f.n_coders = data, len(data), ca, len(chain)
        rc = lib.swc_7z_unpack_folders(arr, n)
        if rc:
            _raise(rc)
        res = []
        for f in arr[:n]:
            res.append((f.status, C.string_at(f.out, f.out_len) if f.out_len else b""))
            lib.swc_free(f.out)
        return res

    @staticmethod
    def unpack(data, chain):
        """SevenZipFolder.unpack(data:) for one folder; raises the folder's error."""
        status, out = SevenZipFolder.unpack_many([(data, chain)])[0]
        if status:
            _raise(status)
        return out
