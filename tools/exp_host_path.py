"""The drop-in path as a caller of the archive layer sees it: bench.py's `archive_paths` and `config1_latency` alone."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from swcompression_amd import _lib, corpus
lib = _lib.load()
r = bench.archive_paths(lib)
p = corpus.p_text(65536, 0x5C0DE + 1)
r["config1_latency"] = bench.config1_latency(lib, [corpus.deflate_raw(p)], [p])
print(json.dumps(r, indent=1))
