"""Cycle split of the LZ4 parse and resolve kernels per stage (library built with SWC_EXTRA_HIPCC_FLAGS=-DSWC_PROFILE).
Usage: exp_profile_lz4.py [tile] [payload]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swcompression_amd import corpus, _lib
from swcompression_amd.batch import DeviceBatch
lib = _lib.load()
units, plains = corpus.build_units("lz4_block", 32, 4 << 20, payload=sys.argv[2] if len(sys.argv) > 2 else "text")
b = DeviceBatch("lz4_block", units, [4 << 20] * len(units), tile=int(sys.argv[1]) if len(sys.argv) > 1 else 128)
prof = torch.zeros(b.n * 32, dtype=torch.int64, device="cuda")
lib.swc_set_profile_buffer(prof.data_ptr())
b.launch(sync=True)
b.launch(sync=True)
p = prof.cpu().numpy().reshape(b.n, 32).astype(np.float64)
names = ["staging", "walk", "scans", "provisional parse", "checked steps + rest"]
t = p[:, :5].sum(axis=1).mean() + p[:, 7].mean()
print("lz4 parse: %.0f kcycles per block; rounds %.0f passes %.0f; %%: " % (t / 1e3, p[:, 5].mean(), p[:, 6].mean())
      + ", ".join("%s %.1f" % (n, 100 * p[:, k].mean() / t) for k, n in enumerate(names)) + ", copy %.1f" % (100 * p[:, 7].mean() / t))
n2 = ["R0+scan", "R1", "R2 expand", "R3 chase+out"]
t2 = p[:, 16:20].sum(axis=1).mean()
print("lz4 resolve: %.0f kcycles per block; batches %.0f span/batch %.0f records/batch %.0f; %%: "
      % (t2 / 1e3, p[:, 20].mean(), (p[:, 21] / p[:, 20]).mean(), (p[:, 22] / p[:, 20]).mean())
      + ", ".join("%s %.1f" % (n, 100 * p[:, 16 + k].mean() / t2) for k, n in enumerate(n2)))
