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


def test_config3_8192_lz4_blocks_of_4MiB():
    nd = 16
    units, plains = corpus.build_units("lz4_block", nd, 4 << 20, seed=31)
    b = DeviceBatch("lz4_block", units, [4 << 20] * nd, tile=8192 // nd)
    assert b.n == 8192
    _check(b, units, plains)


def test_config4_10240_bzip2_blocks_of_900kB():
    nd = 16
    units, plains = corpus.build_units("bzip2", nd, 899000, seed=32)
    b = DeviceBatch("bzip2_block", units, [899000 + 64] * nd, extra=[112] * nd,
                    dict_values=[int.from_bytes(s[10:14], "big") for s in units], tile=10240 // nd)
    assert b.n == 10240
    _check(b, units, plains, consumed_bits=True)


def test_config5_32768_lzma2_units_of_256KiB():
    nd = 64
    units, plains = corpus.build_units("lzma2", nd, 262144, seed=33)
    b = DeviceBatch("lzma2", units, [262144] * nd, aux=[corpus.lzma2_dict_byte(1 << 20)] * nd, tile=32768 // nd)
    assert b.n == 32768
    _check(b, units, plains)
