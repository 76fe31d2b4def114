"""Latency of BASELINE configs[0] -- ONE 64 KiB dynamic-Huffman block through the single-shot C ABI -- with phase 1 as a team of
wavefronts (default for launches of up to 256 streams) and as one wavefront; and of small batches.  SWC_TRACE=1 prints the stages."""
import ctypes as C, sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from swcompression_amd import _lib, corpus
from swcompression_amd.batch import DeviceBatch
lib = _lib.load()
p = corpus.p_text(65536, 1)
z = corpus.deflate_raw(p, 6)
out = C.POINTER(C.c_uint8)(); n = C.c_size_t(); used = C.c_size_t()
for team in (1, 0, 1, 0):
    lib.swc_set_tuning(b"deflate_team", team)
    ts = []
    for k in range(60):
        t0 = time.perf_counter()
        st = lib.swc_deflate_decompress(z, len(z), C.byref(out), C.byref(n), C.byref(used))
        dt = time.perf_counter() - t0
        assert st == 0 and C.string_at(out, n.value) == p
        lib.swc_free(out)
        ts.append(dt)
    ts = sorted(ts[10:])
    print("single shot, team %d: median %.3f ms, min %.3f ms" % (team, ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
lib.swc_set_tuning(b"phase_timing", 1)
for nstreams in (1, 8, 64, 256):
    for team in (1, 0):
        lib.swc_set_tuning(b"deflate_team", team)
        b = DeviceBatch("deflate", [z] * nstreams, [len(p)] * nstreams)
        best = None
        for k in range(8):
            b.launch(sync=True)
            ms = (C.c_float * 8)()
            m = lib.swc_last_phase_ms(ms, 8)
            v = [ms[i] for i in range(m)]
            if best is None or sum(v) < sum(best):
                best = v
        assert all(int(s) == 0 for s in b.results()["status"]) and b.output(nstreams - 1, len(p)) == p
        print("batch of %d, team %d: phases (ms) %s" % (nstreams, team, ["%.3f" % x for x in best]))
lib.swc_set_tuning(b"deflate_team", 1)
