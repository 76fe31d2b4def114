"""GPU tier: LZMA / LZMA2 / XZ through the C ABI on the MI355X vs the oracle
(reference Sources/LZMA, Sources/LZMA2, Sources/XZ)."""
import lzma
import random
import struct

import numpy as np
import pytest

import _oracle as O
import _streams as S
import swcompression_amd as swc
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch

pytestmark = pytest.mark.gpu


def test_lzma2_batch_valid_and_fuzz():
    O.lib.refcpu_set_max_output(1 << 24)
    valid = S.lzma2_valid()
    cases = [(z, db) for z, db, _ in valid] + S.lzma2_fuzz()
    exp = [O.lzma2(z, db) for z, db in cases]
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    b = DeviceBatch("lzma2", [cases[i][0] for i in keep], [max(len(exp[i][1]), 1) + 64 for i in keep],
                    aux=[cases[i][1] for i in keep])
    b.launch(sync=True)
    r = b.results()
    for k, i in enumerate(keep):
        st = int(r["status"][k])
        assert st != 904, "DeviceBatch always supplies the workspace: SWC_E_NEED_WORKSPACE must not come back"
        assert st == exp[i][0], (cases[i][0][:16].hex(), cases[i][1])
        if exp[i][0] == 0:
            assert int(r["out_len"][k]) == len(exp[i][1]) and int(r["in_consumed"][k]) == exp[i][2]
            assert b.output(k, len(exp[i][1])) == exp[i][1]
    O.lib.refcpu_set_max_output(1 << 30)


def test_lzma_alone_and_raw_single_shot():
    O.lib.refcpu_set_max_output(1 << 24)
    for z, x in S.lzma_alone_valid():
        st, out = O.lzma_alone(z)
        if st == 0:
            assert swc.LZMA.decompress(z) == out == x
        else:
            with pytest.raises(swc.SWCError) as ei:
                swc.LZMA.decompress(z)
            assert ei.value.status == st
    for body, p, ds, size in S.lzma_raw_fuzz()[:160]:
        st, out, cons = O.lzma_raw(body, p[0], p[1], p[2], ds, size)
        if st == 901:
            continue
        props = swc.LZMAProperties(p[0], p[1], p[2], ds)
        if st == 0:
            assert swc.LZMA.decompress(body, props, None if size < 0 else size) == out
        else:
            with pytest.raises(swc.SWCError) as ei:
                swc.LZMA.decompress(body, props, None if size < 0 else size)
            assert ei.value.status == st, (p, ds, size, body[:12].hex())
    for data in (b"", b"\x00", bytes(12)):
        with pytest.raises(swc.LZMAError) as ei:
            swc.LZMA.decompress(data)
        assert ei.value.case == "wrongProperties"                      # LzmaTests.swift:42-57
    O.lib.refcpu_set_max_output(1 << 30)


def test_lzma2_data_entry_point():
    x = corpus.p_text(100000, 2)
    raw = corpus.lzma2_raw(x)
    assert swc.LZMA2.decompress(bytes([corpus.lzma2_dict_byte(1 << 20)]) + raw) == x
    assert swc.LZMA2.decompress_raw(raw, corpus.lzma2_dict_byte(1 << 20)) == (x, len(raw))
    with pytest.raises(swc.LZMAError) as ei:
        swc.LZMA2.decompress(b"")
    assert ei.value.case == "rangeDecoderInitError"
    with pytest.raises(swc.LZMA2Error) as ei:
        swc.LZMA2.decompress(bytes([40]) + raw)
    assert ei.value.case == "wrongDictionarySize"


def test_xz_archives():
    x = corpus.p_text(200000, 3)
    for chk in (lzma.CHECK_NONE, lzma.CHECK_CRC32, lzma.CHECK_CRC64, lzma.CHECK_SHA256):
        assert swc.XZArchive.unarchive(lzma.compress(x, check=chk)) == x
    for kind in ("rep", "zero", "rand", "mix"):
        y = corpus.PAYLOADS[kind](70000, 5)
        assert swc.XZArchive.unarchive(lzma.compress(y)) == y
    assert swc.XZArchive.unarchive(lzma.compress(b"")) == b""
    delta = lzma.compress(x, format=lzma.FORMAT_XZ, filters=[{"id": lzma.FILTER_DELTA, "dist": 4}, {"id": lzma.FILTER_LZMA2, "preset": 6}])
    assert swc.XZArchive.unarchive(delta) == x                          # XzTests.swift delta filter
    two = lzma.compress(x[:1000]) + b"\0" * 8 + lzma.compress(x[1000:])
    assert swc.XZArchive.split_unarchive(two) == [x[:1000], x[1000:]]    # multi-stream + padding
    assert swc.XZArchive.unarchive(two) == x
    with pytest.raises(swc.XZError) as ei:
        swc.XZArchive.unarchive(lzma.compress(x) + b"\0" * 3)
    assert ei.value.case == "wrongPadding"
    bad = bytearray(lzma.compress(x, check=lzma.CHECK_CRC32))
    bw = (struct.unpack("<I", bad[-8:-4])[0] + 1) * 4
    bad[len(bad) - 12 - bw - 1] ^= 1
    with pytest.raises(swc.XZError) as ei:
        swc.XZArchive.unarchive(bytes(bad))
    assert ei.value.case == "wrongCheck" and ei.value.data == x         # XzTests.swift:122-139
    for data in (b"\x00", bytes(1 << 16)):
        with pytest.raises(swc.SWCError) as ei:
            swc.XZArchive.unarchive(data)
        assert ei.value.status == O.xz_unarchive(data)[0]
    assert swc.XZArchive.unarchive(b"") == b"" and O.xz_unarchive(b"") == (0, b"")
    rnd = random.Random(4)
    good = lzma.compress(x[:20000], check=lzma.CHECK_CRC64)
    for _ in range(24):                                                  # truncation + bit-flip fuzz: same status
        b2 = bytearray(good)
        if rnd.random() < 0.5:
            b2 = b2[:rnd.randrange(1, len(b2))]
        else:
            b2[rnd.randrange(len(b2))] ^= 1 << rnd.randrange(8)
        st, out = O.xz_unarchive(bytes(b2))
        if st == 0:
            assert swc.XZArchive.unarchive(bytes(b2)) == out
        else:
            with pytest.raises(swc.SWCError) as ei:
                swc.XZArchive.unarchive(bytes(b2))
            assert ei.value.status == st


def _xz_multiblock(payload, block_size=65536, check="crc64"):
    """`xz --block-size` output: ONE stream with many independent blocks, located through the index."""
    import shutil
    import subprocess
    if shutil.which("xz") is None:
        pytest.skip("xz command not available")
    return subprocess.run(["xz", "-z", "-c", "-T1", "--block-size=%d" % block_size, "--check=" + check], input=payload,
                          stdout=subprocess.PIPE, check=True).stdout


def test_xz_multi_block_index_driven_discovery():
    """SURVEY 8d config 5 variant / 8f row 2: the blocks of an `xz --block-size` stream are found through the index and
    decoded in one launch; the sequential walk then consumes them.  Damaged archives must behave exactly like the oracle's
    strictly sequential restatement of XZArchive.swift:90-192."""
    x = corpus.p_text(900000, 8)
    from swcompression_amd import _lib
    lib = _lib.load()
    for chk in ("none", "crc32", "crc64", "sha256"):
        a = _xz_multiblock(x, 65536, chk)
        assert O.xz_unarchive(a) == (0, x)
        hits, launches = lib.swc_stat(b"xz_cache_hits"), lib.swc_stat(b"launches")
        assert swc.XZArchive.unarchive(a) == x
        assert lib.swc_stat(b"xz_cache_hits") - hits == 14          # ceil(900000 / 65536) blocks, all taken from the batch
        assert lib.swc_stat(b"launches") - launches <= 2            # one launch (plus at most one capacity retry), not 14
    a = _xz_multiblock(x, 131072)
    two = a + b"\0" * 4 + _xz_multiblock(x[:300000], 32768, "crc32") + lzma.compress(x[:10])
    assert swc.XZArchive.split_unarchive(two) == [x, x[:300000], x[:10]]
    rnd = random.Random(11)
    for k in range(40):
        b2 = bytearray(a)
        r = rnd.random()
        if r < 0.25:
            b2 = b2[:rnd.randrange(1, len(b2))]
        elif r < 0.5:
            b2[len(b2) - 1 - rnd.randrange(60)] ^= 1 << rnd.randrange(8)      # footer / index
        else:
            b2[rnd.randrange(len(b2))] ^= 1 << rnd.randrange(8)
        st, out = O.xz_unarchive(bytes(b2))
        if st == 0:
            assert swc.XZArchive.unarchive(bytes(b2)) == out
        else:
            with pytest.raises(swc.SWCError) as ei:
                swc.XZArchive.unarchive(bytes(b2))
            assert ei.value.status == st, "variant %d" % k
            if st == 807:
                assert ei.value.data == out


def test_config5_shape_many_256k_units():
    """BASELINE.json config 5 shape at reduced count: independent raw-LZMA2 units of 256 KiB (one 0xE0 chunk each)."""
    units, plains = corpus.build_units("lzma2", 256, 262144)
    db = corpus.lzma2_dict_byte(1 << 20)
    assert all((u[0] & 0xE0) == 0xE0 for u in units[:8])  # first chunk: LZMA with state+props+dictionary reset
    b = DeviceBatch("lzma2", units, [262144] * len(units), aux=[db] * len(units), tile=2)
    b.launch(sync=True)
    r = b.results()
    assert (r["status"] == 0).all() and (r["out_len"] == 262144).all()
    assert (r["in_consumed"] == np.tile(np.array([len(u) for u in units]), 2)).all()
    blob = b.d_out.cpu().numpy()
    for i in range(b.n):
        o = int(b._out_off[i])
        assert blob[o:o + 262144].tobytes() == plains[i % len(units)]


def test_literal_coder_cache_against_whole_model():
    """The two model layouts of the device kernel (lzma_wave.h: LDS as a cache of the literal coders / all coders in LDS) on
    payloads that use few coders (text), all eight of lc = 3 (binary records) and sixteen (lc = 4; lc = 0, lp = 4): the same
    bytes, sizes and consumed input, and both equal to the oracle."""
    from swcompression_amd import _lib
    lib = _lib.load()
    plains = [corpus.p_text(150000, 11), corpus.PAYLOADS["bin"](150000, 12), corpus.p_mix(150000, 13), corpus.p_rand(30000, 14)]
    units, want = [], []
    for p in plains:
        for lc, lp in ((3, 0), (4, 0), (0, 4), (2, 2), (0, 0)):
            f = [{"id": lzma.FILTER_LZMA2, "preset": 6, "lc": lc, "lp": lp, "dict_size": 1 << 20}]
            units.append(lzma.compress(p, format=lzma.FORMAT_RAW, filters=f))
            want.append(p)
    db = corpus.lzma2_dict_byte(1 << 20)
    assert O.lzma2(units[7], db)[:2] == (0, want[7])
    outs = {}
    try:
        for mode in (1, 0):
            assert lib.swc_set_tuning(b"lzma_coder_cache", mode) == 0
            b = DeviceBatch("lzma2", units, [len(p) for p in want], aux=[db] * len(units))
            b.launch(sync=True)
            r = b.results()
            assert (r["status"] == 0).all(), (mode, r["status"])
            outs[mode] = [b.output(k, len(want[k])) for k in range(len(units))]
            for k in range(len(units)):
                assert int(r["out_len"][k]) == len(want[k]) and int(r["in_consumed"][k]) == len(units[k]), (mode, k)
    finally:
        lib.swc_set_tuning(b"lzma_coder_cache", 1)
    for k in range(len(units)):
        assert outs[1][k] == want[k], ("cache", k)
        assert outs[0][k] == want[k], ("whole model", k)
