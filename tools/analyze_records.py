"""Dependency structure of the match records of the bench corpus (guides the LZ resolve kernel design)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import _emu
from swcompression_amd import corpus
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
units, plains = corpus.build_units("gzip", 64, 65536)
lib = _emu.lib
lib.emu_inflate_records.restype = C.c_size_t
lib.emu_inflate_records.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t]
tot = dict(rec=0, inbatch=0, single=0, multi=0, depth_sum=0, batches=0, depth_max=0, lens=[], resid_depth_sum=0, resid_max=0)
for u in units[:32]:
    raw = u[10:-8]
    out = C.create_string_buffer(65536 + 16)
    recs = (C.c_uint32 * 40000)()
    n = lib.emu_inflate_records(raw, len(raw), out, 65536, recs, 40000)
    r = np.frombuffer(recs, dtype=np.uint32, count=n)
    lit = r & 255; ln = ((r >> 8) & 255) + 3; dist = ((r >> 16) & 0x7FFF) + 1
    end = np.cumsum(lit + ln); dst = end - ln
    tot["lens"].append(ln)
    for b0 in range(0, n, T):
        b1 = min(n, b0 + T)
        base = dst[b0] - lit[b0]
        e = end[b0:b1] - base; d = dst[b0:b1] - base; di = dist[b0:b1]; le = ln[b0:b1]
        m = b1 - b0
        depth = np.zeros(m, dtype=np.int32)     # naive chain depth (no redirect)
        rdepth = np.zeros(m, dtype=np.int32)    # residual depth with single-producer redirect
        for j in range(m):
            s0 = d[j] - di[j]; s1 = s0 + min(le[j], di[j])
            if s1 <= 0: continue
            a = np.searchsorted(e[:j], max(s0, 0), side="right"); bb = np.searchsorted(d[:j], s1, side="left") - 1
            if a > bb: continue
            tot["inbatch"] += 1
            depth[j] = 1 + depth[a:bb + 1].max()
            if a == bb and s0 >= d[a] and s1 <= e[a] and di[a] >= le[a]:
                tot["single"] += 1
                rdepth[j] = rdepth[a]            # inherits the producer's wait level
            else:
                tot["multi"] += 1
                rdepth[j] = 1 + rdepth[a:bb + 1].max()
        tot["rec"] += m; tot["batches"] += 1
        tot["depth_sum"] += depth.max(); tot["depth_max"] = max(tot["depth_max"], depth.max())
        tot["resid_depth_sum"] += rdepth.max(); tot["resid_max"] = max(tot["resid_max"], rdepth.max())
L = np.concatenate(tot["lens"])
print("T=%d records=%d per-block=%.0f in-batch-dep=%.1f%% single(redirectable)=%.1f%% multi=%.1f%%" % (T, tot["rec"], tot["rec"] / 32, 100 * tot["inbatch"] / tot["rec"], 100 * tot["single"] / tot["rec"], 100 * tot["multi"] / tot["rec"]))
print("naive chain depth per batch: mean %.1f max %d ; residual depth after redirect: mean %.1f max %d" % (tot["depth_sum"] / tot["batches"], tot["depth_max"], tot["resid_depth_sum"] / tot["batches"], tot["resid_max"]))
print("match length: mean %.1f  p50 %d p90 %d p99 %d  >16: %.1f%% >32: %.1f%%" % (L.mean(), np.percentile(L, 50), np.percentile(L, 90), np.percentile(L, 99), 100 * (L > 16).mean(), 100 * (L > 32).mean()))
