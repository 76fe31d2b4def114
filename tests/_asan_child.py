"""Child process of test_emulation_asan.py: runs emulated kernels (tests/host_emu built with -fsanitize=address) on
exact-size heap buffers, so that any read or write past an input, an output or the workspace aborts."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swcompression_amd import corpus  # noqa: E402

lib = C.CDLL(sys.argv[1])


class Job(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("in_len", C.c_uint64), ("out", C.c_void_p), ("out_cap", C.c_uint64),
                ("out_len", C.c_uint64), ("in_consumed", C.c_uint64), ("status", C.c_int32), ("aux", C.c_int32),
                ("dict", C.c_void_p), ("dict_len", C.c_uint64)]


libc = C.CDLL("libc.so.6")
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
libc.free.argtypes = [C.c_void_p]


def run(fn, data, plain, cap=None):
    cap = len(plain) if cap is None else cap
    j = (Job * 1)()
    ib = libc.malloc(max(len(data), 1))
    C.memmove(ib, data, len(data))
    ob = libc.malloc(max(cap, 1))
    j[0].in_, j[0].in_len, j[0].out, j[0].out_cap = ib, len(data), ob, cap
    getattr(lib, fn)(j, C.c_size_t(1))
    ok = j[0].status == 0 and C.string_at(ob, cap) == plain[:cap]
    libc.free(ib)
    libc.free(ob)
    return ok, j[0].status


n = 0
units, plains = corpus.build_units("lz4_block", 2048, 65536, payload="mix")
for mode in (1, 2):   # both forms of the LZ4 records (eight-byte / derived offsets + anchors): the copier reads literals out of the INPUT
    lib.emu_set_lz4_record_mode(mode)
    for i in list(range(0, 2048, 97)) + [1137]:   # 1137: a true sequence chain that reaches the staged window's last bytes
        ok, st = run("emu_lz4_block", units[i], plains[i])
        assert ok, ("lz4 mix", i, st, mode)
        n += 1
    for kind, size in (("text", 200000), ("mix", 150000), ("bin", 100000)):
        p = corpus.PAYLOADS[kind](size, 5)
        z = corpus.lz4_block(p)
        assert run("emu_lz4_block", z, p)[0], ("lz4", kind, mode)
        for cut in (1, 2, 5, 9, len(z) // 3):
            assert not run("emu_lz4_block", z[:len(z) - cut], p)[0]
        n += 6
lib.emu_set_lz4_record_mode(2)
for kind, size in (("text", 300000), ("mix", 200000), ("rep", 100000), ("zero", 70000), ("rand", 50000)):
    p = corpus.PAYLOADS[kind](size, 3)
    z = corpus.lz4_block(p)
    assert run("emu_lz4_block", z, p)[0], ("lz4", kind)
    assert not run("emu_lz4_block", z[:len(z) * 2 // 3], p)[0]      # truncated: an error status, no stray access
    z = corpus.deflate_raw(p, 6)
    for team in (0, 1):   # one wavefront per stream / a team of wavefronts (the helpers stage rounds up to 21 KB beyond the master's)
        lib.emu_set_deflate_team(team)
        for fn in ("emu_inflate_sync",):
            assert run(fn, z, p)[0], (fn, kind, team)
            assert not run(fn, z[:len(z) * 2 // 3], p)[0]
            n += 2
    lib.emu_set_deflate_team(0)
    n += 2
# LZMA2 in both model layouts (all literal coders in LDS / LDS as a cache of four), whole and truncated
import lzma as _lzma
lib.emu_lzma_mode.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]


def run_lzma2(data, plain, mode, cap=None):
    cap = len(plain) if cap is None else cap
    j = (Job * 1)()
    ib = libc.malloc(max(len(data), 1))
    C.memmove(ib, data, len(data))
    ob = libc.malloc(max(cap, 1))
    j[0].in_, j[0].in_len, j[0].out, j[0].out_cap, j[0].aux = ib, len(data), ob, cap, 16   # 64 KiB dictionary
    lib.emu_lzma_mode(j, 1, 1, mode)
    ok = j[0].status == 0 and C.string_at(ob, cap) == plain[:cap]
    libc.free(ib)
    libc.free(ob)
    return ok, j[0].status


for kind, size in (("text", 120000), ("bin", 90000), ("mix", 80000), ("rand", 20000)):
    p = corpus.PAYLOADS[kind](size, 5)
    for lc, lp in ((3, 0), (0, 4), (4, 0)):
        z = _lzma.compress(p, format=_lzma.FORMAT_RAW, filters=[{"id": _lzma.FILTER_LZMA2, "lc": lc, "lp": lp, "dict_size": 1 << 16}])
        for mode in (0, 1):
            assert run_lzma2(z, p, mode)[0], ("lzma2", kind, lc, lp, mode)
            assert not run_lzma2(z[:len(z) * 2 // 3], p, mode)[0]
            assert run_lzma2(z, p, mode, cap=len(p) // 2)[1] == 901
            n += 3

# the wave-per-stream CRC-32: 16-byte loads at every alignment must stay inside an exact-size buffer
import zlib
lib.emu_crc32_wave.argtypes = [C.c_void_p, C.c_size_t]
lib.emu_crc32_wave.restype = C.c_uint32
base = corpus.p_mix(70000, 9)
for length in list(range(0, 70)) + [2044, 2047, 2048, 2049, 4096, 6143, 6145, 65536, 69999]:
    for mis in (0, 1, 3, 8, 13):
        buf = libc.malloc(max(length + mis, 1))
        C.memmove(buf + mis, base[:length], length)          # the data end exactly where the allocation ends
        assert lib.emu_crc32_wave(buf + mis, length) == zlib.crc32(base[:length]), (length, mis)
        libc.free(buf)
        n += 1
# BZip2 compression: every stage's buffers are allocated at exactly the size the driver asks for
import bz2
lib.emu_bzip2_compress.restype = C.c_int
for x in (b"", b"a", b"ab" * 3000, bytes(range(256)) * 20, corpus.p_text(100000, 3), corpus.p_mix(170000, 4), bytes(90000), corpus.p_rand(30000, 6)):
    for level in (1, 2):
        cap = len(x) + len(x) // 2 + 4096
        ob = libc.malloc(cap)
        got = C.c_size_t(0)
        ib = libc.malloc(max(len(x), 1))
        C.memmove(ib, x, len(x))
        assert lib.emu_bzip2_compress(C.c_void_p(ib), C.c_size_t(len(x)), C.c_int(level), C.c_void_p(ob), C.c_size_t(cap), C.byref(got)) == 0
        assert bz2.decompress(C.string_at(ob, got.value)) == x
        libc.free(ib)
        libc.free(ob)
        n += 1
print("asan-clean", n)
