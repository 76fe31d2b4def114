"""ctypes binding of tests/host_emu/libswc_emu.so -- the device decoders compiled for the host.
TEST INFRASTRUCTURE ONLY (see tests/host_emu/emu.cpp)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "host_emu")
_LIB = os.path.join(_DIR, "libswc_emu.so")
_CSRC = os.path.join(os.path.dirname(_HERE), "swcompression_amd", "csrc")


class Job(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_len", C.c_uint64), ("out", C.c_void_p), ("out_cap", C.c_uint64),
                ("out_len", C.c_uint64), ("in_consumed", C.c_uint64), ("status", C.c_int32), ("aux", C.c_int32),
                ("dict", C.c_void_p), ("dict_len", C.c_uint64)]


def build(force=False):
    srcs = [os.path.join(_DIR, "emu.cpp")] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    if not force and os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-DSWC_HOST_EMULATION", "-fPIC", "-shared",
                    "-Wno-unknown-pragmas", "-pthread", "-o", _LIB, os.path.join(_DIR, "emu.cpp")], check=True)


build()
lib = C.CDLL(_LIB)


def set_order(order):
    """Thread order of the emulated SIMT regions (csrc/simt.h): 0 forward, 1 reverse, 2 shuffled."""
    lib.emu_set_order(C.c_int(order))


def run_batch(fn_name, inputs, caps, aux=None, dicts=None, extra=None, fn_args=(), dict_ptr_values=None, misalign=0):
    """inputs: list[bytes]; caps: list[int].  Returns list of (status, out_bytes, in_consumed, out_len).
    misalign: the output buffers start that many bytes past a 16-byte boundary."""
    n = len(inputs)
    jobs = (Job * n)()
    keep = []
    for i, (data, cap) in enumerate(zip(inputs, caps)):
        ib = C.create_string_buffer(bytes(data), max(len(data), 1))
        ob = C.create_string_buffer(max(cap, 1) + 16 + 48)  # 16 guard bytes on either side + alignment slack
        o0 = (-C.addressof(ob)) % 16 + 16 + misalign
        C.memset(C.addressof(ob), 0xA5, len(ob))
        keep.append((ib, ob, o0))
        jobs[i].in_ = C.addressof(ib)
        jobs[i].in_len = len(data)
        jobs[i].out = C.addressof(ob) + o0
        jobs[i].out_cap = cap
        jobs[i].aux = 0 if aux is None else aux[i]
        if dicts is not None and dicts[i] is not None:
            db = C.create_string_buffer(bytes(dicts[i]), max(len(dicts[i]), 1))
            keep.append(db)
            jobs[i].dict = C.addressof(db)
            jobs[i].dict_len = len(dicts[i])
        if extra is not None:
            jobs[i].dict_len = extra[i]
        if dict_ptr_values is not None:
            jobs[i].dict = dict_ptr_values[i]
    getattr(lib, fn_name)(jobs, C.c_size_t(n), *fn_args)
    res = []
    for i in range(n):
        ib, ob, o0 = keep[i] if dicts is None else [k for k in keep if isinstance(k, tuple)][i]
        cap = caps[i]
        raw = ob.raw
        assert raw[o0 + cap:o0 + cap + 16] == b"\xA5" * 16 and raw[:o0] == b"\xA5" * o0, "guard bytes overwritten (job %d)" % i
        nout = min(jobs[i].out_len, cap)
        res.append((jobs[i].status, raw[o0:o0 + nout], jobs[i].in_consumed, jobs[i].out_len))
    return res


def inflate(inputs, caps, misalign=0):
    """Deflate: inflate_sync.h (one stream per wavefront, 64 sub-chunks at once) + lz_resolve.h."""
    return run_batch("emu_inflate_sync", inputs, caps, misalign=misalign)


def lz4_block(inputs, caps, dicts=None, misalign=0):
    return run_batch("emu_lz4_block", inputs, caps, dicts=dicts, misalign=misalign)


LZMA_MODE = 0   # 0: literal coders in LDS / cell-by-cell spill (kernel without a workspace); 1: LDS as a cache of four coders


def lzma2(inputs, caps, dict_bytes, mode=None):
    return run_batch("emu_lzma_mode", inputs, caps, aux=dict_bytes, fn_args=(C.c_int(1), C.c_int(LZMA_MODE if mode is None else mode)))


def lzma(inputs, caps, props, dict_sizes, sizes, mode=None):
    """props: list of (lc, lp, pb); sizes: declared uncompressed size or -1."""
    aux = [lc | (lp << 8) | (pb << 16) for lc, lp, pb in props]
    extra = [s & 0xFFFFFFFFFFFFFFFF for s in sizes]
    return run_batch("emu_lzma_mode", inputs, caps, aux=aux, extra=extra, fn_args=(C.c_int(0), C.c_int(LZMA_MODE if mode is None else mode)),
                     dict_ptr_values=dict_sizes)


_bzip2_team = None


def set_bzip2_team(mode):
    """None: stage 3a inside the block's own wavefront (bzip2_block.h); (start_team, one_thread): stage 3a as kernels of its own
    (bzip2_team.h) -- the team whose thread goes first, and whether that thread alone draws all teams' tickets."""
    global _bzip2_team
    _bzip2_team = mode


def bzip2_block(streams, body_bits, crcs, caps, lcap=1000000):
    """One bzip2 block per job: `streams[i]` is the whole stream, body_bits[i] the bit offset of the block body."""
    if _bzip2_team is not None:
        return run_batch("emu_bzip2_block_team", streams, caps, extra=body_bits, dict_ptr_values=crcs,
                         fn_args=(C.c_size_t(lcap), C.c_int(_bzip2_team[0]), C.c_int(_bzip2_team[1])))
    return run_batch("emu_bzip2_block", streams, caps, extra=body_bits, dict_ptr_values=crcs, fn_args=(C.c_size_t(lcap),))


lib.emu_checksum.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
lib.emu_checksum.restype = C.c_uint64


def checksum(kind, data, misalign=0):
    """kind: 1 crc32, 2 adler32, 3 crc64, 4 bzip2crc32, 5 xxh32 (swc_checksum of include/swc_hip.h).  The data are
    placed `misalign` bytes past a 64-byte boundary."""
    data = bytes(data)
    buf = C.create_string_buffer(len(data) + 128)
    base = (C.addressof(buf) + 63) // 64 * 64 + misalign
    C.memmove(base, data, len(data))
    return lib.emu_checksum(kind, base, len(data))


lib.emu_crc32_wave.argtypes = [C.c_void_p, C.c_size_t]
lib.emu_crc32_wave.restype = C.c_uint32


def crc32_wave(data, misalign=0):
    """CRC-32 by the wave-per-stream code of the device (crc32_wave.h), the data `misalign` bytes past a 64-byte boundary."""
    data = bytes(data)
    buf = C.create_string_buffer(len(data) + 128)
    base = (C.addressof(buf) + 63) // 64 * 64 + misalign
    C.memmove(base, data, len(data))
    return lib.emu_crc32_wave(base, len(data))


lib.emu_delta.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_uint]
lib.emu_delta.restype = None


def delta(data, distance, in_place=False):
    """DeltaFilter.decode by the device group code (256 host threads).  distance as the reference passes it (0..255)."""
    data = bytes(data)
    if in_place:
        buf = C.create_string_buffer(data, max(len(data), 1))
        lib.emu_delta(C.cast(buf, C.c_char_p), C.cast(buf, C.c_void_p), len(data), distance)
        return buf.raw[:len(data)]
    out = C.create_string_buffer(max(len(data), 1))
    lib.emu_delta(data, C.cast(out, C.c_void_p), len(data), distance)
    return out.raw[:len(data)]


def lz4_compress(blocks, prefixes=None, caps=None):
    """LZ4 block compression (lz4_comp.h): returns list of (status, compressed bytes, in_consumed, out_len)."""
    prefixes = prefixes or [b""] * len(blocks)
    ins = [bytes(p) + bytes(b) for p, b in zip(prefixes, blocks)]
    caps = caps or [len(b) + len(b) // 255 + 16 for b in blocks]
    return run_batch("emu_lz4_compress", ins, caps, extra=[len(p) for p in prefixes])


def deflate_compress(bufs, caps=None, aux=None):
    """Deflate compression (deflate_comp.h): returns list of (status, compressed bytes, in_consumed, out_len).  aux[i] & 1: unit i
    is a segment of a longer stream (BFINAL clear, an empty stored block behind its block)."""
    caps = caps or [len(b) + len(b) // 8 + 32 for b in bufs]
    return run_batch("emu_deflate_compress", [bytes(b) for b in bufs], caps, aux=aux)


def bzip2_compress(data, block_size=1):
    """BZip2 compression (bzip2_comp.h) with the emulation's executor: returns (status, stream bytes)."""
    data = bytes(data)
    cap = len(data) + len(data) // 2 + 4096
    ob = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    lib.emu_bzip2_compress.restype = C.c_int
    st = lib.emu_bzip2_compress(C.c_char_p(data), C.c_size_t(len(data)), C.c_int(block_size), ob, C.c_size_t(cap), C.byref(n))
    assert n.value <= cap
    return st, ob.raw[:n.value]
