"""Cycle split of the two Deflate kernels per stage (library built with SWC_EXTRA_HIPCC_FLAGS=-DSWC_PROFILE)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swcompression_amd import corpus, _lib
from swcompression_amd.batch import DeviceBatch
lib = _lib.load()
units, plains = corpus.build_units("gzip", 2048, 65536)
raw = [u[10:-8] for u in units]
b = DeviceBatch("deflate", raw, [65536] * len(raw), tile=int(sys.argv[1]) if len(sys.argv) > 1 else 48)
prof = torch.zeros(b.n * 32, dtype=torch.int64, device="cuda")
lib.swc_set_profile_buffer(prof.data_ptr())
b.launch(sync=True)
b.launch(sync=True)
p = prof.cpu().numpy().reshape(b.n, 32).astype(np.float64)
n1 = ["header", "tables", "staging", "decode passes", "chain+scans", "copy/emit", "rest"]
t1 = p[:, :7].sum(axis=1).mean() + p[:, 9:16].sum(axis=1).mean()
print("phase 1 (inflate_sync): %.0f kcycles per stream; rounds %.2f passes %.2f; %%: " % (t1 / 1e3, p[:, 7].mean(), p[:, 8].mean())
      + ", ".join("%s %.1f" % (n, 100 * (p[:, k].mean() + (p[:, 14:16].sum(axis=1).mean() if k == 0 else p[:, 10:14].sum(axis=1).mean() if k == 1 else 0)) / t1) for k, n in enumerate(n1))
      + ", walk %.1f" % (100 * p[:, 9].mean() / t1))
fine = ["tables A ranks", "tables B codes+scatter", "tables C direct+links", "tables D subtables", "header: reader, 19 lengths, their code", "header: its table"]
print("   of which (kcycles): " + ", ".join("%s %.1f" % (n, p[:, 10 + k].mean() / 1e3) for k, n in enumerate(fine))
      + ", header windows + reader seek %.1f, tables rest %.1f" % (p[:, 0].mean() / 1e3, p[:, 1].mean() / 1e3))
n2 = ["R0+scan", "R1", "R2 expand", "R3 chase+out"]
t2 = p[:, 16:20].sum(axis=1).mean()
print("phase 2 (lz_resolve): %.0f kcycles per stream; batches %.2f span/batch %.0f records/batch %.0f; %%: "
      % (t2 / 1e3, p[:, 20].mean(), (p[:, 21] / p[:, 20]).mean(), (p[:, 22] / p[:, 20]).mean())
      + ", ".join("%s %.1f" % (n, 100 * p[:, 16 + k].mean() / t2) for k, n in enumerate(n2)))
