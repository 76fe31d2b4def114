"""(needs a library built with SWC_EXTRA_HIPCC_FLAGS=-DSWC_ENABLE_ABLATION_KNOBS python -m swcompression_amd.build --force)
Ablation of the LZ resolve kernel (timing only; results are wrong when dbg != 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swcompression_amd import corpus, _lib
from swcompression_amd.batch import DeviceBatch
lib = _lib.load()
units, plains = corpus.build_units("gzip", 2048, 65536)
raw = [u[10:-8] for u in units]
def timeit(b, reps=3):
    b.launch(sync=True); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); b.launch(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return min(ts)
b = DeviceBatch("deflate", raw, [65536] * len(raw), tile=32)
for T in (512,):
    lib.swc_set_tuning(b"resolve_threads", T)
    for dbg in (0, 1, 2, 4, 8, 9, 15):
        lib.swc_set_tuning(b"resolve_debug", dbg)
        print("T=%d dbg=%2d  total %.2f ms" % (T, dbg, timeit(b)), flush=True)
lib.swc_set_tuning(b"resolve_debug", 0)
