"""Loader for libswc_hip.so.  Fails loudly: there is no fallback implementation in this package."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SWC_LIB (DEVELOPMENT ONLY): a variant build of the same library (tools/build_variant.sh; A/B runs of compile-time knobs on the GPU
# box).  Whatever the variable names is loaded as it is -- nothing a deployment should set.
LIB_PATH = os.environ.get("SWC_LIB") or os.path.join(_HERE, "libswc_hip.so")


class SwcJob(C.Structure):
    """Mirror of `swc_job` (include/swc_hip.h)."""
    _fields_ = [("in_", C.c_void_p), ("in_len", C.c_uint64), ("out", C.c_void_p), ("out_cap", C.c_uint64),
                ("out_len", C.c_uint64), ("in_consumed", C.c_uint64), ("status", C.c_int32), ("aux", C.c_int32),
                ("dict", C.c_void_p), ("dict_len", C.c_uint64)]


class SwcGzipExtraField(C.Structure):
    _fields_ = [("si1", C.c_uint8), ("si2", C.c_uint8), ("bytes", C.c_void_p), ("len", C.c_size_t)]


class SwcBatchOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("stream", C.c_void_p), ("synchronize", C.c_int32), ("reserved", C.c_int32)]


_lib = None


def load():
    """Return the ctypes handle of libswc_hip.so, building it first if the sources are newer."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64; it must be the first HIP runtime mapped into the process,
    # otherwise torch later reports "No HIP GPUs are available".  torch is only plumbing (HBM tensors,
    # streams, torch.distributed) -- a C/Swift host links libswc_hip.so against /opt/rocm directly.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build()
    if not os.path.exists(LIB_PATH):
        raise ImportError("libswc_hip.so is missing and could not be built: the MI355X engine is required")
    lib = C.CDLL(LIB_PATH)
    u8pp = C.POINTER(C.POINTER(C.c_uint8))
    szp = C.POINTER(C.c_size_t)
    szpp = C.POINTER(szp)
    I = C.c_int

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig("swc_free", None, C.c_void_p)
    sig("swc_trim", I)
    sig("swc_device_available", I)
    sig("swc_version", C.c_char_p)
    sig("swc_set_tuning", I, C.c_char_p, I)
    sig("swc_last_phase_ms", I, C.POINTER(C.c_float), I)
    sig("swc_batch_decompress", I, I, C.c_void_p, C.c_size_t, C.POINTER(SwcBatchOpts))
    sig("swc_batch_crc32", I, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(SwcBatchOpts))
    sig("swc_batch_checksum", I, I, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(SwcBatchOpts))
    sig("swc_batch_workspace_bytes", C.c_size_t, I, C.c_size_t, C.c_uint64)
    sig("swc_batch_decompress_ws", I, I, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(SwcBatchOpts))
    for n in ("swc_deflate_decompress", "swc_bzip2_decompress"):
        sig(n, I, C.c_char_p, C.c_size_t, u8pp, szp, szp)
    for n in ("swc_gzip_unarchive", "swc_zlib_unarchive", "swc_xz_unarchive", "swc_lzma_alone_decompress", "swc_lzma2_decompress_data"):
        sig(n, I, C.c_char_p, C.c_size_t, u8pp, szp)
    for n in ("swc_gzip_multi_unarchive", "swc_xz_split_unarchive", "swc_bzip2_multi_decompress"):
        sig(n, I, C.c_char_p, C.c_size_t, u8pp, szp, szpp, szp)
    sig("swc_lzma_decompress", I, C.c_char_p, C.c_size_t, I, I, I, C.c_int64, C.c_int64, u8pp, szp, szp)
    sig("swc_lzma2_decompress", I, C.c_char_p, C.c_size_t, C.c_uint8, u8pp, szp, szp)
    sig("swc_lz4_decompress", I, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int64, u8pp, szp, szp)
    sig("swc_lz4_multi_decompress", I, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int64, u8pp, szp, szpp, szp)
    sig("swc_deflate_compress", I, C.c_char_p, C.c_size_t, u8pp, szp)
    sig("swc_bzip2_compress", I, C.c_char_p, C.c_size_t, I, u8pp, szp)
    sig("swc_zlib_archive", I, C.c_char_p, C.c_size_t, u8pp, szp)
    sig("swc_gzip_archive", I, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, I, I, I, I, C.c_int64,
        C.c_void_p, C.c_size_t, u8pp, szp)
    sig("swc_lz4_compress", I, C.c_char_p, C.c_size_t, I, I, I, I, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int64, u8pp, szp)
    sig("swc_zip_get_entries_data", I, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t)
    sig("swc_stat", C.c_longlong, C.c_char_p)
    sig("swc_set_profile_buffer", C.c_int, C.c_void_p)
    sig("swc_7z_unpack_folders", I, C.c_void_p, C.c_size_t)
    sig("swc_index_blocks", I, I, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, szp)
    sig("swc_unarchive_many", I, I, C.POINTER(C.c_char_p), szp, C.c_size_t, u8pp, szp, C.POINTER(C.c_int32))
    sig("swc_unarchive_many_devices", I, I, C.POINTER(C.c_char_p), szp, C.c_size_t, C.POINTER(C.c_int), C.c_size_t, u8pp, szp,
        C.POINTER(C.c_int32))
    sig("swc_crc32", C.c_uint32, C.c_char_p, C.c_size_t, C.c_uint32)
    sig("swc_adler32", C.c_uint32, C.c_char_p, C.c_size_t)
    sig("swc_crc64", C.c_uint64, C.c_char_p, C.c_size_t)
    sig("swc_bzip2_crc32", C.c_uint32, C.c_char_p, C.c_size_t)
    sig("swc_xxh32", C.c_uint32, C.c_char_p, C.c_size_t, C.c_uint32)
    sig("swc_sha256", None, C.c_char_p, C.c_size_t, C.c_char_p)
    # SWC_TUNING="key=value,key=value": performance knobs for experiments (swc_set_tuning; results never change)
    for item in filter(None, os.environ.get("SWC_TUNING", "").split(",")):
        key, _, val = item.partition("=")
        if lib.swc_set_tuning(key.strip().encode(), int(val)) != 0:
            raise ValueError("SWC_TUNING: bad entry %r" % item)
    _lib = lib
    return lib
