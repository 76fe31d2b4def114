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
for i in list(range(0, 2048, 97)) + [1137]:   # 1137: a true sequence chain that reaches the staged window's last bytes
    ok, st = run("emu_lz4_block", units[i], plains[i])
    assert ok, ("lz4 mix", i, st)
    n += 1
for kind, size in (("text", 300000), ("mix", 200000), ("rep", 100000), ("zero", 70000), ("rand", 50000)):
    p = corpus.PAYLOADS[kind](size, 3)
    z = corpus.lz4_block(p)
    assert run("emu_lz4_block", z, p)[0], ("lz4", kind)
    assert not run("emu_lz4_block", z[:len(z) * 2 // 3], p)[0]      # truncated: an error status, no stray access
    z = corpus.deflate_raw(p, 6)
    for fn in ("emu_inflate_sync",):
        assert run(fn, z, p)[0], (fn, kind)
        assert not run(fn, z[:len(z) * 2 // 3], p)[0]
        n += 2
    n += 2
print("asan-clean", n)
