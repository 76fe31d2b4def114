"""LZ4 compression (reference Sources/LZ4/LZ4+Compress.swift).  The device compressor does not reproduce
the reference encoder's bytes (a hash table in LDS instead of an exact dictionary): the contract is that what it writes
decodes to the input under the REFERENCE DECODER's rules -- the oracle's restatement of LZ4.process(block:_:), end-of-block
rules included -- and under liblz4.  CPU tier: the kernel source built for the host; GPU tier: the C ABI."""
import ctypes as C
import json
import os
import random
import struct

import pytest

import _emu as E
import _oracle as O
from swcompression_amd import corpus


GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_inline_vectors.json")))


def _payloads():
    rnd = random.Random(11)
    out = [b"", b"a", b"abcd" * 3, b"x" * 12, b"x" * 13, b"x" * 14, b"abcdefgh" * 40, bytes(range(256)) * 3]
    # the inputs of the reference's own compression tests (tests/golden: written inline in its XCTest sources)
    out += [s.encode("latin1") for s in GOLD["roundtrip_strings"]] + [bytes.fromhex(h) for h in GOLD["roundtrip_bytes"]]
    for kind in ("text", "mix", "rep", "zero", "rand"):
        for n in (1, 5, 12, 13, 64, 65, 300, 4096, 65536, 70001, 300000):
            out.append(corpus.PAYLOADS[kind](n, rnd.randrange(1000)))
    return out


def _liblz4_decode(z, n, prefix=b""):
    l4 = corpus._liblz4()
    l4.LZ4_decompress_safe_usingDict.restype = C.c_int
    l4.LZ4_decompress_safe_usingDict.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
    dst = C.create_string_buffer(max(n, 1))
    k = l4.LZ4_decompress_safe_usingDict(z, dst, len(z), n, prefix, len(prefix))
    assert k == n, k
    return dst.raw[:n]


def test_blocks_decode_with_the_reference_rules_and_liblz4():
    O.lib.refcpu_set_max_output(1 << 22)
    plains = [p for p in _payloads() if len(p) > 0]
    res = E.lz4_compress(plains)
    for p, r in zip(plains, res):
        st, z, _, zl = r
        assert st == 0 and zl == len(z) <= len(p) + len(p) // 255 + 16
        assert O.lz4_block(z)[:2] == (0, p), len(p)          # LZ4.swift:332-413 incl. the end-of-block rules (:369-376)
        assert _liblz4_decode(z, len(p)) == p
    # compressible data must actually shrink
    for kind in ("text", "rep", "zero"):
        p = corpus.PAYLOADS[kind](200000, 3)
        z = E.lz4_compress([p])[0][1]
        assert len(z) < len(p) * (0.75 if kind == "text" else 0.05), (kind, len(z))
    O.lib.refcpu_set_max_output(1 << 30)


def test_prefix_blocks_and_capacity():
    rnd = random.Random(5)
    text = corpus.p_text(400000, 9)
    blocks, prefixes = [], []
    for _ in range(12):
        a = rnd.randrange(0, 300000)
        n = rnd.choice([100, 5000, 65536, 90000])
        pre = rnd.choice([0, 17, 4096, 65536, 70000])
        pre = min(pre, a)
        blocks.append(text[a:a + n]); prefixes.append(text[a - pre:a])
    res = E.lz4_compress(blocks, prefixes)
    O.lib.refcpu_set_max_output(1 << 22)
    for b, pre, r in zip(blocks, prefixes, res):
        assert r[0] == 0
        assert O.lz4_block(r[1], pre if pre else None)[:2] == (0, b)     # dictionary = what lies in front (LZ4.swift:334)
        assert _liblz4_decode(r[1], len(b), pre) == b
    # with a prefix that holds the same text the block must come out much smaller than without
    with_pre = E.lz4_compress([text[100000:150000]], [text[60000:100000] + text[100000:110000]])[0]
    without = E.lz4_compress([text[100000:150000]])[0]
    assert len(with_pre[1]) < len(without[1])
    # a capacity that is too small: SWC_E_CAPACITY with the size needed, nothing written past it
    need = without[3]
    r = E.lz4_compress([text[100000:150000]], caps=[need - 1])[0]
    assert r[0] == 901 and r[3] == need
    O.lib.refcpu_set_max_output(1 << 30)


def test_oracle_restatement_of_the_reference_compressor():
    """oracle/rc_lz4c.c restates LZ4.compress(block:_:): its blocks must decode under the reference decoder's rules and liblz4
    (pinning of the restatement), and the engine's hash-table compressor must stay within a few per cent of its ratio."""
    O.lib.refcpu_set_max_output(1 << 22)
    for p in [x for x in _payloads() if 0 < len(x) <= 70001]:
        st, z = O.lz4_compress_block(p)
        assert st == 0 and O.lz4_block(z)[:2] == (0, p) and _liblz4_decode(z, len(p)) == p
    pre, blk = corpus.p_text(70000, 5), corpus.p_text(200000, 6)
    st, z = O.lz4_compress_block(blk, pre[-65536:])
    assert st == 0 and O.lz4_block(z, pre[-65536:])[:2] == (0, blk)
    for kind in ("text", "mix"):
        p = corpus.PAYLOADS[kind](300000, 12)
        ref, eng = O.lz4_compress_block(p)[1], E.lz4_compress([p])[0][1]
        assert len(eng) <= len(ref) * 1.15, (kind, len(eng), len(ref))
    O.lib.refcpu_set_max_output(1 << 30)


@pytest.mark.parametrize("order", [1, 2])
def test_lane_order_does_not_matter(order):
    p = corpus.p_mix(150000, 4)
    want = E.lz4_compress([p])[0][1]
    E.set_order(order)
    try:
        assert E.lz4_compress([p])[0][1] == want
    finally:
        E.set_order(0)


@pytest.mark.gpu
def test_gpu_frames_decode_with_oracle_liblz4_and_own_decoder():
    import swcompression_amd as swc
    rnd = random.Random(21)
    O.lib.refcpu_set_max_output(1 << 24)
    for p in [b""] + [x for x in _payloads() if len(x) in (1, 13, 300, 70001, 300000)] + [corpus.p_text(5 << 20, 3), corpus.p_mix(3 << 20, 4)]:
        for _ in range(2):
            kw = dict(independent_blocks=rnd.random() < 0.6, block_checksums=rnd.random() < 0.5, content_checksum=rnd.random() < 0.7,
                      content_size=rnd.random() < 0.5, block_size=rnd.choice([1024, 65536, 200000, 1 << 20, 4 << 20]))
            use_dict = rnd.random() < 0.3
            d = corpus.p_text(rnd.choice([100, 70000]), 77) if use_dict else None
            did = rnd.randrange(1 << 32) if use_dict and rnd.random() < 0.5 else None
            f = swc.LZ4.compress(p, dictionary=d, dictionary_id=did, **kw)
            assert f[:4] == b"\x04\x22\x4d\x18" and f[4] & 0xC0 == 0x40
            assert O.lz4(f, d, -1 if did is None else did)[:2] == (0, p), (len(p), kw)          # the reference's frame walk
            assert swc.LZ4.decompress(f, d, did) == p                                           # and the engine's own decoder
    # the default call: LZ4.compress(data:) -- independent, content checksum, 4 MiB blocks (LZ4+Compress.swift:16-19)
    x = corpus.p_text(9 << 20, 8)
    f = swc.LZ4.compress(x)
    assert f[4] == 0x64 and f[5] == 0x70 and len(f) < len(x) * 0.75
    assert O.lz4(f)[:2] == (0, x)
    # incompressible data is stored (LZ4+Compress.swift:119-127)
    r = corpus.p_rand(100000, 2)
    f = swc.LZ4.compress(r, block_size=65536)
    assert struct.unpack("<I", f[7:11])[0] == 0x80000000 | 65536 and O.lz4(f)[:2] == (0, r)
    with pytest.raises(swc.SWCError):
        swc.LZ4.compress(b"abc", block_size=(4 << 20) + 1)                                      # :50 precondition
    O.lib.refcpu_set_max_output(1 << 30)


def test_tricky_sequence_block_on_the_host_build():
    """Tests/LZ4CompressionTests.swift:162-172 (a match index once used as a cyclical index): the 21 bytes through the block
    compressor built for the host, decoded by the oracle's restatement of the reference decoder and by liblz4; and through the
    oracle's restatement of the reference ENCODER."""
    t = GOLD["lz4_tricky_sequence"]
    p = bytes.fromhex(t["input"])
    assert len(p) == 21
    st, z, _, zl = E.lz4_compress([p])[0]
    assert st == 0 and O.lz4_block(z)[:2] == (0, p) and _liblz4_decode(z, len(p)) == p
    zr = O.lz4_compress_block(p)[1]
    assert O.lz4_block(zr)[:2] == (0, p) and _liblz4_decode(zr, len(p)) == p


@pytest.mark.gpu
def test_gpu_tricky_sequence_frame():
    """The same vector through LZ4.compress with the options the reference's test passes (dependent blocks, block checksums,
    content checksum, content size), then the oracle's frame walk and the engine's own decoder."""
    import swcompression_amd as swc
    t = GOLD["lz4_tricky_sequence"]
    p = bytes.fromhex(t["input"])
    f = swc.LZ4.compress(p, **t["options"])
    assert f[:4] == b"\x04\x22\x4d\x18" and f[4] == 0x40 | 0x10 | 0x08 | 0x04     # version 01, block checksums, content size, content checksum
    assert struct.unpack("<Q", f[6:14])[0] == len(p)
    assert O.lz4(f)[:2] == (0, p)
    assert swc.LZ4.decompress(f) == p


@pytest.mark.gpu
def test_gpu_prefixes_in_place_and_in_rounds():
    """Dependent blocks address the input in front of them in place (odd block sizes: the units start anywhere); small
    independent blocks with a 64 KiB dictionary -- 64 KiB of joined prefix per KiB of data -- go in bounded rounds."""
    import swcompression_amd as swc
    O.lib.refcpu_set_max_output(1 << 24)
    x = corpus.p_text(700000, 31) + corpus.p_mix(300001, 32)
    d = corpus.p_text(70000, 33)
    for kw in (dict(independent_blocks=False, block_size=1000), dict(independent_blocks=False, block_size=65537, dictionary=d),
               dict(independent_blocks=False, block_size=99999), dict(independent_blocks=True, block_size=173)):
        f = swc.LZ4.compress(x, **kw)
        assert O.lz4(f, kw.get("dictionary"))[:2] == (0, x), kw
        assert swc.LZ4.decompress(f, kw.get("dictionary")) == x
    y = corpus.p_text(6 << 20, 34)                                     # 6,144 blocks of 1 KiB x (64 KiB + 1 KiB) = 400 MB joined: two rounds
    f = swc.LZ4.compress(y, independent_blocks=True, block_size=1024, dictionary=d)
    assert O.lz4(f, d)[:2] == (0, y)
    O.lib.refcpu_set_max_output(1 << 30)


@pytest.mark.gpu
def test_gpu_batch_codec_many_blocks():
    from swcompression_amd.batch import DeviceBatch
    plains = [corpus.PAYLOADS[k](n, 5 + i) for i, (k, n) in enumerate([("text", 65536), ("mix", 65536), ("rand", 4000), ("zero", 65536), ("rep", 30000)] * 40)]
    b = DeviceBatch("lz4_compress", plains, [len(p) + len(p) // 255 + 16 for p in plains])
    b.launch(sync=True)
    r = b.results()
    O.lib.refcpu_set_max_output(1 << 22)
    for i, p in enumerate(plains):
        assert int(r["status"][i]) == 0
        z = b.output(i, int(r["out_len"][i]))
        if i % 7 == 0:
            assert O.lz4_block(z)[:2] == (0, p)
        assert _liblz4_decode(z, len(p)) == p
    O.lib.refcpu_set_max_output(1 << 30)
