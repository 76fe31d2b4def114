"""Where the cycles of the LZMA2 kernel go (library built with SWC_EXTRA_HIPCC_FLAGS=-DSWC_PROFILE): per unit of BASELINE
configs[4] (256 KiB of text), the share of the match copies, of the dictionary reads in the decision chain, of the literal
symbols and of the length / distance decode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from swcompression_amd import corpus, _lib
from swcompression_amd.batch import DeviceBatch
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
units, plains = corpus.build_units("lzma2", 32, 256 << 10)
tile = max(1, n // len(units))
b = DeviceBatch("lzma2", units, [256 << 10] * len(units), aux=[corpus.lzma2_dict_byte(1 << 20)] * len(units), tile=tile)
n = b.n
prof = torch.zeros(b.n * 32, dtype=torch.int64, device="cuda")
lib.swc_set_profile_buffer(prof.data_ptr())
b.launch(sync=True)
p = prof.cpu().numpy().reshape(b.n, 32).astype(np.float64).mean(axis=0)
t = p[0]
print("lzma2: %.1f Mcycles per unit (%d units in flight); matches %.0f, literals %.0f, short reps %.0f" % (t / 1e6, n, p[5], p[6], p[7]))
print("  match copies        %5.1f %%  (%.0f cycles each)" % (100 * p[1] / t, p[1] / max(p[5], 1)))
print("  dictionary byte reads %5.1f %%  (%.0f cycles each; inside the literal share)" % (100 * p[2] / t, p[2] / max(p[6] + p[7], 1)))
print("  literal symbols     %5.1f %%  (%.0f cycles each)" % (100 * p[3] / t, p[3] / max(p[6], 1)))
print("  length + distance   %5.1f %%  (%.0f cycles per match)" % (100 * p[4] / t, p[4] / max(p[5], 1)))
print("  rest (isMatch, loop) %5.1f %%" % (100 * (t - p[1] - p[3] - p[4]) / t))
print("  isMatch decision    %5.1f %%  (%.0f cycles each)" % (100 * p[8] / t, p[8] / max(p[5] + p[6] + p[7], 1)))
print("  put()               %5.1f %%  (%.0f cycles each)" % (100 * p[9] / t, p[9] / max(p[6] + p[7], 1)))
print("  matched part of literals %5.1f %%  (%.0f cycles each, %.0f of them)" % (100 * p[10] / t, p[10] / max(p[11], 1), p[11]))
