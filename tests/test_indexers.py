"""CPU tier: host block discovery as a library call (swc_index_blocks).  No device is needed for
discovery; every discovered unit is then decoded by the ORACLE from exactly the bytes the index assigns to it."""
import bz2
import lzma
import shutil
import subprocess

import numpy as np
import pytest

import _oracle as O
import swcompression_amd as swc
from swcompression_amd import corpus


def test_bgzf_members():
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 97))
    parts = [corpus.p_mix(int(rng.integers(0, 40000)), 8000 + i) for i in range(40)]
    data = b"".join(corpus.gzip_member(p, bgzf=True) for p in parts)
    refs = swc.index_blocks("bgzf", data)
    assert len(refs) == len(parts)
    for (off, clen, ulen, _), p in zip(refs, parts):
        assert ulen == len(p)
        st, out, used = O.deflate(data[off:off + clen])
        assert (st, out, used) == (0, p, clen)
    with pytest.raises(swc.SWCError):                       # a member without the 'BC' field: not indexable
        swc.index_blocks("bgzf", data + corpus.gzip_member(b"plain member"))


def test_lz4_frame_blocks():
    x = corpus.p_text(700000, 4)
    frame = corpus.lz4f_frame(x, 4, False, True, True)       # 64 KiB independent blocks, with checksums
    refs = swc.index_blocks("lz4", frame)
    assert len(refs) == (len(x) + 65535) // 65536
    O.lib.refcpu_set_max_output(1 << 22)
    out = b""
    for off, clen, _, stored in refs:
        blk = frame[off:off + clen]
        out += blk if stored else O.lz4_block(blk)[1]
    assert out == x
    stored = corpus.lz4f_frame(corpus.p_rand(100000, 2), 4, False)
    assert any(r[3] == 1 for r in swc.index_blocks("lz4", stored))


def test_bzip2_block_magics():
    x = corpus.p_text(450000, 6)
    z = bz2.compress(x, 1)                                   # 100 kB blocks
    refs = swc.index_blocks("bzip2", z)
    bits = [r[0] for r in refs]
    assert bits[0] == 32 and bits == sorted(bits) and len(bits) >= 5
    assert swc.index_blocks("bzip2", b"BZh9") == []


def test_xz_blocks_through_the_index():
    if shutil.which("xz") is None:
        pytest.skip("xz command not available")
    x = corpus.p_text(500000, 9)
    a = subprocess.run(["xz", "-z", "-c", "-T1", "--block-size=65536"], input=x, stdout=subprocess.PIPE, check=True).stdout
    two = a + lzma.compress(x[:1000])                        # two streams
    refs = swc.index_blocks("xz", two)
    assert len(refs) == (len(x) + 65535) // 65536 + 1
    out = b""
    for off, clen, ulen, dict_byte in refs:
        st, part, used = O.lzma2(two[off:off + clen], dict_byte)
        assert st == 0 and len(part) == ulen and used == clen
        out += part
    assert out == x + x[:1000]
    assert swc.index_blocks("xz", b"\\xfd7zXZ\\x00garbage") == []


def test_bzip2_magic_scan_against_a_bit_string_search():
    """The byte-window / multi-thread scan against a plain search over the bit string: magics planted at every bit
    alignment, at the very end, across the thread-range boundaries of a large buffer, and random data (no false hits)."""
    magic = "%048d" % int(bin(0x314159265359)[2:])
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 101))

    def plant(buf, bit):
        bits = np.unpackbits(np.frombuffer(bytes(buf[bit // 8:bit // 8 + 7]), dtype=np.uint8))
        m = np.array([int(c) for c in magic], dtype=np.uint8)
        bits[bit % 8:bit % 8 + 48] = m
        buf[bit // 8:bit // 8 + 7] = np.packbits(bits).tobytes()

    def reference(buf):
        s = "".join("%d" % b for b in np.unpackbits(np.frombuffer(bytes(buf), dtype=np.uint8)))
        out, at = [], s.find(magic, 32)
        while at >= 0:
            out.append(at)
            at = s.find(magic, at + 1)
        return out

    for n in (6, 7, 13, 64, 1000):
        for trial in range(12):
            buf = bytearray(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
            if n >= 13:
                plant(buf, int(rng.integers(32, (n - 7) * 8)))
            if n >= 64:
                plant(buf, (n - 6) * 8)          # the last possible position
                plant(buf, 32 + trial)           # every alignment near the start
            assert [r[0] for r in swc.index_blocks("bzip2", bytes(buf))] == reference(buf), (n, trial)
    big = bytearray(rng.integers(0, 256, 9 << 20, dtype=np.uint8).tobytes())   # two scan threads (one per 4 MiB, at least 2)
    per = (len(big) + 1) // 2
    for delta_bits in (-47, -8, -1, 0, 3):                                     # magics that straddle / touch the range boundary
        plant(big, per * 8 + delta_bits + 640 * (delta_bits + 47))
    plant(big, per * 8 - 20 + 100000)
    want = reference(big)
    assert len(want) >= 6
    assert [r[0] for r in swc.index_blocks("bzip2", bytes(big))] == want


def test_lzma2_chunk_walk():
    """kind 'lzma2': LZMA2Decoder.decode()/dispatch() (LZMA2Decoder.swift:36-74) as an index: chunk boundaries sum up to the
    whole stream, unpack sizes to the plain length, and every run of chunks that begins at a dictionary reset decodes on its
    own (checked with the oracle)."""
    import _oracle as O
    x = corpus.p_text(700000, 3) + corpus.p_rand(200000, 4) + corpus.p_text(400000, 5)
    db = corpus.lzma2_dict_byte(1 << 16)
    raw = corpus.lzma2_raw(x, dict_size=1 << 16)
    stream = bytes([db]) + raw
    refs = swc.index_blocks("lzma2", stream)
    assert len(refs) > 5
    assert refs[0][0] == 1 and refs[0][4] & 1                       # the first chunk must reset the dictionary
    pos = 1
    for off, comp, unc, control, flags in refs:
        assert off == pos and stream[off] == control
        pos += comp
    assert stream[pos] == 0 and pos == len(stream) - 1              # end marker right behind the last chunk
    assert sum(r[2] for r in refs) == len(x)
    assert any(r[3] in (1, 2) for r in refs)                        # the random part is stored uncompressed
    # independent runs: cut at dictionary resets; each run + end marker is a stream of its own
    starts = [i for i, r in enumerate(refs) if r[4] & 1]
    done = 0
    for a, b in zip(starts, starts[1:] + [len(refs)]):
        lo, hi = refs[a][0], refs[b - 1][0] + refs[b - 1][1]
        n = sum(r[2] for r in refs[a:b])
        st, out, _ = O.lzma2(stream[lo:hi] + b"\x00", db)
        assert st == 0 and out == x[done:done + n]
        done += n
    assert done == len(x)
    # errors: a control byte in 3...0x7F is LZMA2Error.wrongControlByte; a chunk past the end is trap-class
    bad = bytearray(stream); bad[refs[1][0]] = 0x40
    with pytest.raises(swc.SWCError) as ei:
        swc.index_blocks("lzma2", bytes(bad))
    assert ei.value.status == O.lzma2(bytes(bad[1:]), db)[0]
    with pytest.raises(swc.SWCError) as ei:
        swc.index_blocks("lzma2", stream[:refs[2][0] + 4])
    assert ei.value.status == 900
    assert swc.index_blocks("lzma2", bytes([db, 0])) == []
