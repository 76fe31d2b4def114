"""BZip2 launches of n blocks of 900 kB: one wavefront per block through all stages (bzip2_team_walk = 0) against stage 3 as kernels
of its own (= 2).  python tools/exp_bzteam_sweep.py"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import bench
from swcompression_amd import _lib

lib = _lib.load()
w = dict(bench.WORKLOADS["bzip2_900k"])
for n in (1, 8, 32, 128, 512, 2048):
    row = []
    for mode in (0, 2):
        assert lib.swc_set_tuning(b"bzip2_team_walk", mode) == 0
        batch, plains, raw, trailers = bench.make_batch("bzip2_900k", w, [("text", min(n, 64))], 0x5C0DE, torch.device("cuda:0"), (0, n))
        batch.launch(sync=True)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t = time.perf_counter()
            batch.launch(sync=True)
            ts.append((time.perf_counter() - t) * 1e3)
        r = batch.results()
        if not (r["status"] == 0).all():
            print("n", n, "mode", mode, "statuses", r["status"].tolist()[:16], "out_len", r["out_len"].tolist()[:16])
        row.append(min(ts))
        del batch
    print("n = %5d blocks: fused %.2f ms, team %.2f ms" % (n, row[0], row[1]))
