"""A/B of the LZ77 copy phase on the device: the byte-cell resolver (lz_copier=0) against the record-granular copier with an
8 KiB / 16 KiB window (1 / 2), same batch, every unit verified after every mode.
    python tools/exp_copier.py deflate64k|lz4_4m|deflate64k_mix [scale] [modes]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from swcompression_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "deflate64k"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
modes = [int(m) for m in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2]
lib = _lib.load()
w = bench.WORKLOADS[name]


class A:
    pass


args = A()
args.scale = scale
args.parts = None
args.workload = name
parts = bench.scaled_parts(w, scale)
n_distinct = sum(n for _, n in parts)
n_total = w["n_units"] if scale == 1.0 else max(n_distinct, int(w["n_units"] * scale))
batch, raw, plains, trailers = bench.make_batch(name, w, parts, 2, "cuda:0", (0, n_total))
names = w["kernels"]
for mode in modes:
    assert lib.swc_set_tuning(b"lz_copier", mode) == 0
    batch.launch(sync=True)
    batch.wipe_results()
    torch.cuda.synchronize()
    lib.swc_set_tuning(b"phase_timing", 1)
    acc = [0.0] * len(names)
    reps = 4
    t0 = time.perf_counter()
    for _ in range(reps):
        batch.launch(sync=True)
        buf = (C.c_float * 8)()
        assert lib.swc_last_phase_ms(buf, 8) == len(names)
        for i in range(len(names)):
            acc[i] += buf[i]
    wall = (time.perf_counter() - t0) / reps
    lib.swc_set_tuning(b"phase_timing", 0)
    v = {"units_verified": 0} if os.environ.get("NOVERIFY") else bench.verify_all_units(name, batch, raw, plains, trailers)
    print("lz_copier=%d  %s  wall %.2f ms  verified %d" % (mode, {n: round(a / reps, 2) for n, a in zip(names, acc)}, wall * 1e3, v["units_verified"]), flush=True)
