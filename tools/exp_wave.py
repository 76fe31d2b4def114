"""Deflate phase 1: one stream per wavefront vs one per lane, by batch size (64 KiB members)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swcompression_amd import _lib, corpus
from swcompression_amd.batch import DeviceBatch

lib = _lib.load()
units, plains = corpus.build_units("gzip", 1024, 65536)
raw = [u[10:-8] for u in units]
lib.swc_set_tuning(b"phase_timing", 1)
import ctypes as C
for n in (1, 16, 256, 1024, 4096, 8192, 16384, 32768):
    rs = (raw * (n // len(raw) + 1))[:n]
    b = DeviceBatch("deflate", rs, [65536] * n)
    for mode, lim in (("wave", 1 << 30), ("lane", 0)):
        lib.swc_set_tuning(b"inflate_wave_max_jobs", lim)
        b.launch(sync=True)
        ts, ph = [], None
        for _ in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); b.launch(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
            buf = (C.c_float * 4)()
            if lib.swc_last_phase_ms(buf, 4) == 2: ph = (buf[0], buf[1])
        r = b.results()
        ok = bool((r["status"] == 0).all()) and b.output(n - 1, 65536) == plains[(n - 1) % len(plains)]
        print("n=%6d %s  total %.3f ms  phase1 %.3f ms  resolve %.3f ms  ok=%s" % (n, mode, min(ts), ph[0], ph[1], ok), flush=True)
    del b
