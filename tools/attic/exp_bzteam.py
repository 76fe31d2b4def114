"""bzip2_900k launches with the team walk forced, no verification (variant builds cut short produce wrong results).  For kernel traces."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch

import bench
from swcompression_amd import _lib

lib = _lib.load()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
assert lib.swc_set_tuning(b"bzip2_team_walk", mode) == 0
w = bench.WORKLOADS["bzip2_900k"]
batch, plains, raw, trailers = bench.make_batch("bzip2_900k", w, w["parts"], 0x5C0DE, torch.device("cuda:0"), (0, w["n_units"]))
for _ in range(3):
    batch.launch(sync=True)
r = batch.results()
print("done; status ok:", int((r["status"] == 0).sum()), "of", batch.n)
