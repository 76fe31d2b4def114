"""One Deflate launch pair at a chosen size (for rocprofv3 --pmc runs): python tools/exp_small.py [n_distinct] [tile]"""
import sys, os, pickle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cache = "/tmp/swc_corpus_%d.pkl" % n
if os.path.exists(cache):
    raw = pickle.load(open(cache, "rb"))
else:
    units, plains = corpus.build_units("gzip", n, 65536)
    raw = [u[10:-8] for u in units]
    pickle.dump(raw, open(cache, "wb"))
b = DeviceBatch("deflate", raw, [65536] * len(raw), tile=tile)
b.launch(sync=True)
b.launch(sync=True)
r = b.results()
assert (r["status"] == 0).all()
