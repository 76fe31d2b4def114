"""Statistics of the match records of the bench corpus (host emulation of phase 1; guides the LZ resolve kernel):
bytes per record, literal share, matches that overlap themselves (distance < length), how far matches reach."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _emu
from swcompression_amd import corpus

units, plains = corpus.build_units("gzip", 32, 65536)
lib = _emu.lib
lib.emu_inflate_records.restype = C.c_size_t
lib.emu_inflate_records.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t]
tot = dict(records=0, lit_only=0, out=0, lit=0, match_bytes=0, overlap_bytes=0, overlap2_bytes=0, far=0)
span = 8176
for u in units:
    raw = u[10:-8]
    out = C.create_string_buffer(65536 + 16)
    recs = (C.c_uint32 * 40000)()
    n = lib.emu_inflate_records(raw, len(raw), out, 65536, recs, 40000)
    r = np.frombuffer(recs, dtype=np.uint32, count=n).astype(np.int64)
    lit = r & 127
    ln = (r >> 7) & 511
    lit = np.where(ln == 0, lit + ((r >> 16) << 7), lit)
    dist = np.where(ln == 0, 1 << 30, (r >> 16) + 1)
    tot["records"] += n
    tot["lit_only"] += int((ln == 0).sum())
    tot["out"] += int((lit + ln).sum())
    tot["lit"] += int(lit.sum())
    tot["match_bytes"] += int(ln.sum())
    tot["overlap_bytes"] += int(np.maximum(ln - dist, 0).sum())          # bytes beyond the first period (m >= dist)
    tot["overlap2_bytes"] += int(np.maximum(ln - 2 * dist, 0).sum())     # bytes beyond the second period
    tot["far"] += int(ln[dist > span].sum())
print("records %d (literal-only %d), %.1f output bytes per record, literals %.1f %% of the output"
      % (tot["records"], tot["lit_only"], tot["out"] / tot["records"], 100 * tot["lit"] / tot["out"]))
print("match bytes beyond the first period (distance < length): %.2f %% of the output; beyond the second: %.2f %%"
      % (100 * tot["overlap_bytes"] / tot["out"], 100 * tot["overlap2_bytes"] / tot["out"]))
print("match bytes whose source is certainly before the batch span (distance > %d): %.1f %% of the match bytes"
      % (span, 100 * tot["far"] / tot["match_bytes"]))
