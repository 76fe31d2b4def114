"""(needs a library built with SWC_EXTRA_HIPCC_FLAGS=-DSWC_ENABLE_ABLATION_KNOBS python -m swcompression_amd.build --force)
Ablation of the Deflate entropy kernel (timing only; results are wrong when dbg != 0)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swcompression_amd import corpus, _lib
from swcompression_amd.batch import DeviceBatch
lib = _lib.load()
units, plains = corpus.build_units("gzip", 2048, 65536)
raw = [u[10:-8] for u in units]
lib.swc_set_tuning(b"phase_timing", 1)
lib.swc_set_tuning(b"resolve_debug", 15)
for tile in (48, 32):
    b = DeviceBatch("deflate", raw, [65536] * len(raw), tile=tile)
    for dbg in (0, 1, 2, 3):
        lib.swc_set_tuning(b"inflate_debug", dbg)
        b.launch(sync=True)
        acc = 0.0
        for _ in range(3):
            b.launch(sync=True)
            buf = (C.c_float * 4)()
            lib.swc_last_phase_ms(buf, 4)
            acc += buf[0] / 3
        print("jobs=%d inflate_debug=%d  phase1 %.2f ms" % (b.n, dbg, acc), flush=True)
    del b
