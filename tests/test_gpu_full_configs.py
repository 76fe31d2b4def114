"""GPU tier: BASELINE.json's configurations at their FULL unit counts (configs[2] 8192 x 4 MiB LZ4 blocks, configs[3]
10240 x 900 kB bzip2 blocks, configs[4] 32768 x 256 KiB LZMA2 units) through the C ABI.  A batch of this size cannot be
compared byte by byte on the host in test time, so every unit is checked through two independent device checksums of its
output (swc_batch_checksum: CRC-32 and XXH32) against the host's checksums of the plain text -- a checksum of checksums,
the size-independent property SURVEY.md section 8c names -- plus status / out_len / in_consumed of every job, and a sample
of units byte by byte."""
import zlib

import numpy as np
import pytest

import _oracle as O
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch

pytestmark = pytest.mark.gpu


def _check(b, raw, plains, consumed_bits=False):
    b.launch(sync=True)
    r = b.results()
    k = b.unit_index
    assert (r["status"] == 0).all(), np.unique(r["status"])
    assert (r["out_len"] == np.array([len(plains[i]) for i in k], dtype=np.uint64)).all()
    if not consumed_bits:
        assert (r["in_consumed"] == np.array([len(raw[i]) for i in k], dtype=np.uint64)).all()
    want_crc = np.array([zlib.crc32(p) for p in plains], dtype=np.uint64)[k]
    want_xxh = np.array([O.xxh32(p) for p in plains], dtype=np.uint64)[k]
    assert (b.checksum("crc32") == want_crc).all()
    assert (b.checksum("xxh32") == want_xxh).all()
    for i in (0, 1, b.n // 2, b.n - 1):
        assert b.output(i, len(plains[k[i]])) == plains[k[i]]


# 256 distinct units per configuration, of mixed payload classes (text, text / repeated phrase / random pieces spliced every
# 4 KiB, and a few all-random and all-zero units: stored blocks, literal-only sequences, uncompressed LZMA2 chunks, runs), tiled
# to the full unit count at distinct device addresses.
MIX = [("text", 160), ("mix", 64), ("rand", 16), ("rep", 12), ("zero", 4)]


def test_config3_8192_lz4_blocks_of_4MiB():
    units, plains = corpus.build_units_mixed("lz4_block", [(c, n // 2) for c, n in MIX], 4 << 20, seed=31)   # 128 distinct (0.5 GiB of payload on the host)
    nd = len(units)
    b = DeviceBatch("lz4_block", units, [4 << 20] * nd, tile=8192 // nd)
    assert b.n == 8192
    _check(b, units, plains)


def test_config4_10240_bzip2_blocks_of_900kB():
    units, plains = corpus.build_units_mixed("bzip2", [(c, n) for c, n in MIX if c != "zero"] + [("text", 4)], 899000, seed=32)
    nd = len(units)
    assert nd == 256 and all(u.count(bytes.fromhex("314159265359")) >= 1 for u in units)
    b = DeviceBatch("bzip2_block", units, [899000 + 64] * nd, extra=[112] * nd,
                    dict_values=[int.from_bytes(s[10:14], "big") for s in units], tile=10240 // nd)
    assert b.n == 10240
    _check(b, units, plains, consumed_bits=True)


def test_config5_32768_lzma2_units_of_256KiB():
    units, plains = corpus.build_units_mixed("lzma2", MIX, 262144, seed=33)
    nd = len(units)
    assert nd == 256
    b = DeviceBatch("lzma2", units, [262144] * nd, aux=[corpus.lzma2_dict_byte(1 << 20)] * nd, tile=32768 // nd)
    assert b.n == 32768
    _check(b, units, plains)


def test_config2_100000_gzip_members_of_64KiB():
    """BASELINE configs[1] at its full count with 4,096 distinct members of mixed classes (the bench line uses P-text only):
    raw Deflate streams, every member's CRC-32 / XXH32 / length on the device."""
    units, plains = corpus.build_units_mixed("deflate", [("text", 2560), ("mix", 1024), ("rand", 256), ("rep", 192), ("zero", 64)], 65536, seed=34)
    nd = len(units)
    b = DeviceBatch("deflate", units, [65536] * nd, tile=100000 // nd + 1, select=(0, 100000))
    assert b.n == 100000
    _check(b, units, plains)
