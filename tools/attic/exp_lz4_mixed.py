"""Per-block cycles of the LZ4 kernels in the MIXED bench workload (profile build): text vs P-mix blocks, and the start / end
times a block's wave reports -- is the launch as long as its slowest blocks, or as its busiest CUs?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from swcompression_amd import corpus, _lib
from swcompression_amd.batch import DeviceBatch
lib = _lib.load()
parts = [("text", 192), ("mix", 64)] if len(sys.argv) < 2 else [(c, int(k)) for c, k in (x.split(":") for x in sys.argv[1].split(","))]
units, plains = corpus.build_units_mixed("lz4_block", parts, 4 << 20, seed=2)
kinds = []
taken = [0] * len(parts)
# (the interleave of build_units_mixed, restated: which class is unit i)
total = sum(n for _, n in parts)
for i in range(total):
    k = max(range(len(parts)), key=lambda j: (parts[j][1] * (i + 1) / total - taken[j]) if taken[j] < parts[j][1] else -1e9)
    kinds.append(k); taken[k] += 1
b = DeviceBatch("lz4_block", units, [4 << 20] * len(units), tile=8192 // len(units))
prof = torch.zeros(b.n * 32, dtype=torch.int64, device="cuda")
lib.swc_set_profile_buffer(prof.data_ptr())
b.launch(sync=True)
b.launch(sync=True)
p = prof.cpu().numpy().reshape(b.n, 32).astype(np.float64)
kind_of = np.array([kinds[i] for i in b.unit_index])
tp = p[:, :5].sum(axis=1) + p[:, 7]
tr = p[:, 16:20].sum(axis=1)
for k, (c, _) in enumerate(parts):
    m = kind_of == k
    print("%s blocks: parse %.1f Mcycles (max %.1f), resolve %.1f Mcycles (max %.1f)" % (c, tp[m].mean() / 1e6, tp[m].max() / 1e6, tr[m].mean() / 1e6, tr[m].max() / 1e6))
print("sum over blocks / (256 CUs x slots): parse %.1f Mcycles at 19 slots, resolve %.1f Mcycles at 2 slots" % (tp.sum() / 256 / 19 / 1e6, tr.sum() / 256 / 2 / 1e6))
