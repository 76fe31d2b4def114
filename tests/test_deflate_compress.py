"""Deflate.compress (reference Sources/Deflate/Deflate+Compress.swift:22-213) on the device.

The contract differs from the decoders': the engine's stream is A valid Deflate stream for the same bytes, not the
reference encoder's bytes (a hash table with collisions, every position of a window entered, finds other matches than the
reference's exact dictionary of looked-up positions).  So parity is
  * decode(compress(x)) == x under the REFERENCE decoder (the oracle's restatement of Deflate.swift:30-249, which must also
    consume the whole stream), under zlib, and -- GPU tier -- under the engine's own decoder;
  * the block-type rule of Deflate+Compress.swift:30-45 (stored when not larger and at most 65,535 bytes);
  * a size within a few per cent of the reference encoder restated (oracle/rc_deflatec.c).
CPU tier: the kernel source on the host emulation; GPU tier: the C ABI (swc_deflate_compress, swc_zlib_archive, batches)."""
import json
import os
import random
import zlib

import pytest

import _emu as E
import _oracle as O
from swcompression_amd import corpus

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_inline_vectors.json")))


def payloads():
    rnd = random.Random(5)
    ps = [b"", b"a", b"ab", b"abc", b"aaaa", b"ban", b"banana", b"abaaba", b"abracadabra", b"cabbage", b"baabaabac", b"AAAAAAABBBBCCCD", b"AAAAAAA",
          bytes(range(256)), bytes([0x2E, 0x20, 0x2E, 0x20, 0x2E, 0x20, 0x20]),      # DeflateCompressionTests.swift:29-39,77-85
          b"\0" * 70000, b"ab" * 40000, corpus.p_text(65536, 11), corpus.p_text(300000, 12), corpus.p_mix(100000, 13),
          corpus.p_rand(5000, 14), corpus.p_rand(70000, 15), corpus.p_text(65530, 16), corpus.p_rand(65530, 17), corpus.p_rand(65531, 18)]
    for _ in range(12):
        n = rnd.choice([1, 2, 3, 4, 5, 63, 64, 65, 257, 258, 259, 300, 1000, 4097])
        ps.append(bytes(rnd.choice(b"abcd") for _ in range(n)))
    return ps


def check_stream(x, z):
    assert zlib.decompress(z, -15) == x
    st, y, cons = O.deflate(z)
    assert (st, y, cons) == (0, x, len(z)), (st, cons, len(z))
    # one block, the last one, stored or static (Deflate+Compress.swift:12-20)
    assert z[0] & 1 == 1 and (z[0] >> 1) & 3 in (0, 1)
    if (z[0] >> 1) & 3 == 0:
        assert len(z) == 5 + len(x) <= 65535
    return (z[0] >> 1) & 3


def test_oracle_compressor_round_trips_and_block_rule():
    """oracle/rc_deflatec.c -- the restatement of the reference encoder -- against its own decoder, zlib and the block rule."""
    for x in payloads():
        z = O.deflate_compress(x)
        kind = check_stream(x, z)
        if len(x) > 65530:
            assert kind == 1                      # "if data size is greater than 65535 ... static Huffman" (:18-20)
    for x in [s.encode("latin1") for s in GOLD["roundtrip_strings"]] + [bytes.fromhex(h) for h in GOLD["roundtrip_bytes"]]:
        check_stream(x, O.deflate_compress(x))            # the inputs of the reference's own compression tests (tests/golden)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_emulated_kernel_round_trips(order):
    E.set_order(order)
    try:
        ps = payloads() + [s.encode("latin1") for s in GOLD["roundtrip_strings"]] + [bytes.fromhex(h) for h in GOLD["roundtrip_bytes"]]
        for x, (st, z, cons, n) in zip(ps, E.deflate_compress(ps)):
            assert st == 0 and n == len(z) and cons == len(x)
            check_stream(x, z)
    finally:
        E.set_order(0)


def test_emulated_kernel_does_not_depend_on_lane_order():
    ps = [corpus.p_text(50000, 21), corpus.p_mix(30000, 22), b"abcabcabc" * 500]
    ref = [r[1] for r in E.deflate_compress(ps)]
    for order in (1, 2):
        E.set_order(order)
        try:
            assert [r[1] for r in E.deflate_compress(ps)] == ref
        finally:
            E.set_order(0)


def test_size_against_the_reference_encoder_restated():
    """The hash table of 8,192 positions against the reference's exact dictionary."""
    for x, limit in ((corpus.p_text(262144, 31), 1.05), (corpus.p_mix(262144, 32), 1.05), (corpus.p_text(65536, 33), 1.05)):
        ours = len(E.deflate_compress([x])[0][1])
        ref = len(O.deflate_compress(x))
        assert ours <= ref * limit, (ours, ref, ours / ref)


def test_capacity_reports_required_size():
    x = corpus.p_text(20000, 41)
    full = E.deflate_compress([x])[0]
    st, z, cons, n = E.deflate_compress([x], caps=[len(full[1]) - 7])[0]
    assert st == 901 and n == len(full[1])


def test_segments_of_a_longer_stream_join_into_one():
    """job.aux bit 0 (what swc_deflate_compress does with a buffer of more than 1 MiB): every segment but the last is a NON-FINAL
    block followed by an empty stored block, so it ends on a byte, and the segments one behind the other are one Deflate stream --
    for the reference's decoder restated (Deflate.swift:30-249: block after block until BFINAL) and for zlib.  Segments of every
    kind: text, incompressible (a stored block when short enough), empty, a few bytes (every padding 0 .. 7 comes up)."""
    rnd = random.Random(3)
    segs = [corpus.p_text(70000, 1), corpus.p_rand(3000, 2), b"", b"a", b"ab", corpus.p_text(333, 3), corpus.p_rep(5000, 4), corpus.p_rand(70000, 5)]
    segs += [corpus.p_text(rnd.randint(1, 400), 10 + i) for i in range(24)] + [corpus.p_text(4097, 6)]
    res = E.deflate_compress(segs, aux=[1] * (len(segs) - 1) + [0])
    stream = b""
    for (st, z, _, zl), s in zip(res[:-1], segs[:-1]):
        assert st == 0 and zl == len(z) and z[0] & 1 == 0                                        # non-final
        if (z[0] >> 1) & 3 == 1:
            assert z[-4:] == b"\x00\x00\xff\xff"                                                 # static: the empty stored block behind it (LEN = 0, NLEN = 0xFFFF)
        else:
            assert (z[0] >> 1) & 3 == 0 and len(z) == 5 + len(s)                                   # stored: ends on a byte as it is
        stream += z
    assert res[-1][0] == 0 and res[-1][1][0] & 1 == 1
    stream += res[-1][1]
    plain = b"".join(segs)
    assert zlib.decompress(stream, -15) == plain
    st, out, used = O.deflate(stream)
    assert (st, out) == (0, plain) and used == len(stream)


# ---------------------------------------------------------------------------------------------------------------------- GPU tier
@pytest.mark.gpu
def test_gpu_single_shot_and_zlib_archive():
    import swcompression_amd as swc
    for x in payloads():
        z = swc.Deflate.compress(x)
        check_stream(x, z)
        assert swc.Deflate.decompress(z) == x
        a = swc.ZlibArchive.archive(x)
        assert a[:2] == bytes([120, 218]) and zlib.decompress(a) == x and swc.ZlibArchive.unarchive(a) == x
        assert O.zlib_unarchive(a)[:2] == (0, x)


@pytest.mark.gpu
def test_gpu_large_buffer_in_segments():
    """swc_deflate_compress on more than 1 MiB: segments of 256 KiB in one launch, one stream out (one wavefront took a
    100 MB buffer tens of seconds)."""
    import time
    import swcompression_amd as swc
    x = corpus.p_text(5 * (1 << 20) + 12345, 77) + corpus.p_rand(300000, 78) + corpus.p_text(1 << 20, 79)
    t0 = time.perf_counter()
    z = swc.Deflate.compress(x)
    dt = time.perf_counter() - t0
    assert z[0] & 1 == 0 and len(z) < len(x) * 0.62                      # several blocks; text still halves
    assert zlib.decompress(z, -15) == x
    st, out, used = O.deflate(z)
    assert (st, out) == (0, x) and used == len(z)
    assert swc.Deflate.decompress(z) == x
    assert dt < 1.0, dt                                                   # (one wavefront over 6.5 MB would take more than a second)
    a = swc.ZlibArchive.archive(x)
    assert zlib.decompress(a) == x


@pytest.mark.gpu
def test_gpu_gzip_archive_header_and_trailer():
    """GzipArchive.archive (GzipArchive.swift:126-240): the header fields as the reference writes them, the body a Deflate stream
    of the device, CRC-32 and ISIZE; read back by Python's gzip, the oracle's GzipArchive.unarchive and the engine's."""
    import gzip
    import struct
    import swcompression_amd as swc
    x = corpus.p_text(100000, 51)
    a = swc.GzipArchive.archive(x)
    assert a[:10] == bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 2, 255])                       # no flags, no MTIME, XFL 2, OS unknown
    assert a[-8:] == struct.pack("<II", zlib.crc32(x) & 0xFFFFFFFF, len(x))
    assert gzip.decompress(a) == x and O.gzip_unarchive(a) == (0, x) and swc.GzipArchive.unarchive(a) == x
    b = swc.GzipArchive.archive(x, comment="a comment \xe9", file_name="name.txt", write_header_crc=True, is_text_file=True, os_type=3,
                                modification_time=1700000000.9, extra_fields=[(ord("B"), ord("C"), b"\x34\x12"), (1, 2, b"")])
    assert b[3] == 0x1F and b[4:8] == struct.pack("<I", 1700000000) and b[8] == 2 and b[9] == 3
    assert b[10:12] == struct.pack("<H", 10) and b[12:18] == b"BC\x02\x00\x34\x12" and b[18:22] == bytes([1, 2, 0, 0])
    assert b[22:31] == b"name.txt\0" and b[31:43] == "a comment \xe9".encode("latin-1") + b"\0"
    assert b[43:45] == struct.pack("<H", zlib.crc32(b[:43]) & 0xFFFF)                     # FHCRC: the low half of the header's CRC-32
    assert gzip.decompress(b) == x and O.gzip_unarchive(b) == (0, x) and swc.GzipArchive.unarchive(b) == x
    e = swc.GzipArchive.archive(b"", comment="", file_name="x\0")
    assert e[3] == 0x18 and e[10:13] == b"x\0\0" and gzip.decompress(e) == b""          # a name that ends in zero keeps its one; an empty comment is a zero
    with pytest.raises(swc.GzipError) as ei:
        swc.GzipArchive.archive(x, comment="\u20ac")
    assert ei.value.case == "cannotEncodeISOLatin1"
    with pytest.raises(swc.GzipError):
        swc.GzipArchive.archive(x, extra_fields=[(1, 1, bytes(65532))])                    # 4 + 65,532 > 65,535 (:190-191)


@pytest.mark.gpu
def test_gpu_batch_decodes_on_the_device():
    """4,096 distinct 64 KiB buffers compressed in one launch, decoded again by the engine's own decoder in one launch."""
    import numpy as np
    from swcompression_amd.batch import DeviceBatch
    plains = [corpus.p_text(65536, 700 + i) if i % 4 else corpus.p_mix(65536, 700 + i) for i in range(4096)]
    enc = DeviceBatch("deflate_compress", plains, [65536 + 65536 // 8 + 16] * len(plains))
    enc.launch(sync=True)
    r = enc.results()
    assert (r["status"] == 0).all()
    streams = [enc.output(i, int(r["out_len"][i])) for i in range(len(plains))]
    for i in (0, 1, 2, 3, 1000, 4095):
        check_stream(plains[i], streams[i])
    dec = DeviceBatch("deflate", streams, [65536] * len(plains))
    dec.launch(sync=True)
    d = dec.results()
    assert (d["status"] == 0).all() and (d["out_len"] == 65536).all()
    want = np.array([zlib.crc32(p) & 0xFFFFFFFF for p in plains], dtype=np.uint32)
    assert (dec.crc32() == want).all()
    ref = sum(len(O.deflate_compress(p)) for p in plains[:64])
    assert sum(len(s) for s in streams[:64]) <= ref * 1.05
