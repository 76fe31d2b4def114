"""GPU tier: a soak of the two LZ77 decode paths (Deflate, LZ4 block) and of LZMA2 / BZip2 on randomly BUILT streams, against the oracle; the three encoders on the same kind of text.

The other GPU tests take their streams from encoders, which never produce most of what the formats allow.  Here the LZ4
blocks are assembled sequence by sequence (LZ4.swift:341-412) from random literal-run lengths, match lengths and offsets
that sit on and around every boundary of the two kernels' record formats (literal runs of 14 / 15 / 16 / 269 / 270 and of
kilobytes, match lengths of 18 / 19 / 20 / 273 / 274 and of tens of kilobytes, offsets of 1 .. 65,535 including the ones that
reach exactly to the start of the output), and the Deflate streams come from zlib over plain text that is itself a random
splice of payload classes and copies of earlier stretches at every distance up to the window.  Every stream is decoded by
the oracle; the engine must return the oracle's status and -- for status 0 -- its bytes, consumed input and length; damaged
copies (bit flips, truncations, garbage tails) must return the oracle's status.  Each round runs once with the library's
choice of phase-2 kernel and once with the wave kernel forced.

SWC_SOAK_ROUNDS (default 2) sets the number of seeded rounds; profiles/r06z_soak_gpu.log holds runs of 300 - 1,500 rounds per test.
"""
import os
import random

import pytest

import _oracle as O
import _soak as K
from swcompression_amd.batch import DeviceBatch

pytestmark = pytest.mark.gpu

ROUNDS = int(os.environ.get("SWC_SOAK_ROUNDS", "2"))


def _both_kernels(fn):
    from swcompression_amd import _lib
    lib = _lib.load()
    try:
        for mode in (1, -1):
            assert lib.swc_set_tuning(b"lz_copier", mode) == 0
            fn("auto" if mode == 1 else "wave")
    finally:
        lib.swc_set_tuning(b"lz_copier", 1)


def _check(codec, streams, exp, caps, label, aux=None):
    b = DeviceBatch(codec, streams, caps, aux=aux)
    b.launch(sync=True)
    r = b.results()
    for i, e in enumerate(exp):
        what = "%s, stream %d (%d bytes in, oracle status %d)" % (label, i, len(streams[i]), e[0])
        assert int(r["status"][i]) == e[0], "status %d: %s" % (int(r["status"][i]), what)
        if e[0] == 0:
            assert int(r["out_len"][i]) == len(e[1]) and int(r["in_consumed"][i]) == e[2], what
            assert b.output(i, len(e[1])) == e[1], "bytes differ: " + what


@pytest.mark.parametrize("seed", range(ROUNDS))
def test_lz4_blocks_built_sequence_by_sequence(seed):
    rnd = random.Random(0x4C5A34 + seed)
    O.lib.refcpu_set_max_output(1 << 24)
    try:
        streams = []
        for i in range(220):
            z, p = K.random_lz4_block(rnd, rnd.choice([1, 40, 700, 5000, 66000, 140000, 600000, 2500000]))
            st, out = O.lz4_block(z)[:2]
            assert out == p[:len(out)] and (st != 0 or out == p), "the builder and the oracle disagree on a block it built (seed %d, block %d)" % (seed, i)
            streams.append(z)
        streams += [K.damage(rnd, streams[rnd.randrange(len(streams))]) for _ in range(160)]
        exp = [O.lz4_block(z) + (len(z),) for z in streams]   # (a block that decodes is consumed whole)
        keep = [i for i, e in enumerate(exp) if e[0] != 901]
        caps = [max(len(exp[i][1]), 1) + rnd.choice([0, 0, 0, 3, 64]) for i in keep]
        _both_kernels(lambda label: _check("lz4_block", [streams[i] for i in keep], [exp[i] for i in keep], caps, "seed %d, %s" % (seed, label)))
    finally:
        O.lib.refcpu_set_max_output(1 << 30)


@pytest.mark.parametrize("seed", range(ROUNDS))
def test_deflate_streams_over_spliced_text(seed):
    rnd = random.Random(0xDEF1A7E + seed)
    O.lib.refcpu_set_max_output(1 << 24)
    try:
        streams = []
        for i in range(260):
            z = K.random_deflate_stream(rnd, rnd.choice([0, 1, 9, 300, 4000, 65536, 70000, 200000, 700000]), 7000 * seed + i)
            streams.append(z)
        streams += [K.damage(rnd, streams[rnd.randrange(len(streams))]) for _ in range(200)]
        exp = [O.deflate(z) for z in streams]
        keep = [i for i, e in enumerate(exp) if e[0] != 901]
        caps = [max(len(exp[i][1]), 1) + rnd.choice([0, 0, 0, 5]) for i in keep]
        from swcompression_amd import _lib
        lib = _lib.load()
        try:
            for team in (1, 0, -1):   # phase 1: the library's choice (launches of up to 256 streams: a team of wavefronts per stream), one wavefront per stream, teams forced
                assert lib.swc_set_tuning(b"deflate_team", team) == 0
                _both_kernels(lambda label: _check("deflate", [streams[i] for i in keep], [exp[i] for i in keep], caps, "seed %d, team %d, %s" % (seed, team, label)))
        finally:
            lib.swc_set_tuning(b"deflate_team", 1)
    finally:
        O.lib.refcpu_set_max_output(1 << 30)


@pytest.mark.parametrize("seed", range(ROUNDS))
def test_lzma2_units_of_random_encoder_settings(seed):
    """The range decoder's decision is one block of gfx950 instructions (lzma_wave.h: SWC_LZMA_BIT_ASM): every lc / lp / pb,
    dictionaries from 4 KiB, both layouts of the literal coders."""
    from swcompression_amd import _lib
    lib = _lib.load()
    rnd = random.Random(0x7A4C + seed)
    O.lib.refcpu_set_max_output(1 << 24)
    try:
        units = [K.random_lzma2_unit(rnd, rnd.choice([0, 1, 9, 300, 4000, 70000, 300000]), 5000 * seed + i) for i in range(96)]
        units += [(K.damage(rnd, z), db) for z, db in (units[rnd.randrange(len(units))] for _ in range(64))]
        exp = [O.lzma2(z, db) for z, db in units]
        keep = [i for i, e in enumerate(exp) if e[0] != 901]
        caps = [max(len(exp[i][1]), 1) + rnd.choice([0, 0, 64]) for i in keep]
        try:
            for mode in (1, 0):
                assert lib.swc_set_tuning(b"lzma_coder_cache", mode) == 0
                _check("lzma2", [units[i][0] for i in keep], [exp[i] for i in keep], caps, "seed %d, coder cache %d" % (seed, mode), aux=[units[i][1] for i in keep])
        finally:
            lib.swc_set_tuning(b"lzma_coder_cache", 1)
    finally:
        O.lib.refcpu_set_max_output(1 << 30)


@pytest.mark.parametrize("seed", range(ROUNDS))
def test_bzip2_streams_over_spliced_text(seed):
    """libbz2 at every block size over spliced text (zero runs and distance-1 copies: runs of every length for the RLE1 undo, the
    ones of 4 .. 259 bytes and the megabyte ones), damaged copies; all archives of a round in one launch, the inverse BWT once as
    the team kernels and once inside the block's wavefront."""
    import bz2
    import swcompression_amd as swc
    from swcompression_amd import _lib
    lib = _lib.load()
    rnd = random.Random(0xB2 + 977 * seed)
    O.lib.refcpu_set_max_output(1 << 24)
    try:
        streams = [bz2.compress(K.spliced_plain(rnd, rnd.choice([0, 1, 9, 300, 4000, 70000, 250000, 1200000]), 3000 * seed + i), rnd.randrange(1, 10)) for i in range(48)]
        streams += [K.damage(rnd, streams[rnd.randrange(len(streams))]) for _ in range(48)]
        exp = [O.bzip2(z) for z in streams]
        keep = [i for i, e in enumerate(exp) if e[0] != 901]
        try:
            for mode in (2, 0):
                assert lib.swc_set_tuning(b"bzip2_team_walk", mode) == 0
                got = swc.unarchive_many("bzip2", [streams[i] for i in keep])
                for k, i in enumerate(keep):
                    what = "seed %d, team walk %d, stream %d (%d bytes in)" % (seed, mode, i, len(streams[i]))
                    assert got[k][0] == exp[i][0], "status %d, oracle %d: %s" % (got[k][0], exp[i][0], what)
                    if exp[i][0] == 0:
                        assert got[k][1] == exp[i][1], "bytes differ: " + what
        finally:
            lib.swc_set_tuning(b"bzip2_team_walk", 1)
    finally:
        O.lib.refcpu_set_max_output(1 << 30)


@pytest.mark.parametrize("seed", range(ROUNDS))
def test_encoders_on_spliced_text(seed):
    """The three encoders on spliced text of random sizes (Deflate.compress beyond the megabyte above which it works in segments,
    LZ4 frames of every block size with and without linked blocks, BZip2 at every block size): the ORACLE's decoder, which must
    consume the whole stream, and the system's codec return the input."""
    import bz2
    import zlib
    import swcompression_amd as swc
    rnd = random.Random(0xE2C + 31 * seed)
    O.lib.refcpu_set_max_output(1 << 24)
    try:
        for i in range(10):
            p = K.spliced_plain(rnd, rnd.choice([0, 1, 2, 3, 9, 300, 4000, 65536, 300000, 1048576, 1048577, 2500000]), 800 * seed + i)
            z = swc.Deflate.compress(p)
            assert O.deflate(z) == (0, p, len(z)), "Deflate.compress, seed %d, buffer %d (%d bytes)" % (seed, i, len(p))
            assert zlib.decompress(z, -15) == p
        for i in range(10):
            p = K.spliced_plain(rnd, rnd.choice([0, 1, 12, 13, 300, 4000, 65536, 65537, 300000, 2500000]), 900 * seed + i)
            kw = dict(independent_blocks=rnd.random() < 0.5, block_checksums=rnd.random() < 0.5, content_checksum=rnd.random() < 0.5,
                      content_size=rnd.random() < 0.5, block_size=rnd.choice([173, 1024, 65536, 65537, 200000, 1 << 20, 4 << 20]))
            z = swc.LZ4.compress(p, **kw)
            assert O.lz4(z) == (0, p, len(z)), "LZ4.compress, seed %d, buffer %d (%d bytes, %r)" % (seed, i, len(p), kw)
        for i in range(6):
            p = K.spliced_plain(rnd, rnd.choice([0, 1, 9, 300, 4000, 99999, 100000, 100001, 250000, 1000000]), 700 * seed + i)
            bs = rnd.randrange(1, 10)
            z = swc.BZip2.compress(p, bs)
            assert O.bzip2(z) == (0, p, len(z)), "BZip2.compress, seed %d, buffer %d (%d bytes, block size %d)" % (seed, i, len(p), bs)
            assert bz2.decompress(z) == p
    finally:
        O.lib.refcpu_set_max_output(1 << 30)
