import ctypes as C, sys, time, os
sys.path.insert(0, "/root/repo")
from swcompression_amd import _lib, corpus
lib = _lib.load()
p = corpus.p_text(65536, 1)
z = corpus.deflate_raw(p, 6)
out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); used = C.c_size_t()
for k in range(12):
    st = lib.swc_deflate_decompress(z, len(z), C.byref(out), C.byref(n), C.byref(used))
    lib.swc_free(out)
