"""GPU tier: a soak of the two LZ77 decode paths (Deflate, LZ4 block) on randomly BUILT streams, against the oracle.

The other GPU tests take their streams from encoders, which never produce most of what the formats allow.  Here the LZ4
blocks are assembled sequence by sequence (LZ4.swift:341-412) from random literal-run lengths, match lengths and offsets
that sit on and around every boundary of the two kernels' record formats (literal runs of 14 / 15 / 16 / 269 / 270 and of
kilobytes, match lengths of 18 / 19 / 20 / 273 / 274 and of tens of kilobytes, offsets of 1 .. 65,535 including the ones that
reach exactly to the start of the output), and the Deflate streams come from zlib over plain text that is itself a random
splice of payload classes and copies of earlier stretches at every distance up to the window.  Every stream is decoded by
the oracle; the engine must return the oracle's status and -- for status 0 -- its bytes, consumed input and length; damaged
copies (bit flips, truncations, garbage tails) must return the oracle's status.  Each round runs once with the library's
choice of phase-2 kernel and once with the wave kernel forced.

SWC_SOAK_ROUNDS (default 2) sets the number of seeded rounds; profiles/r06z_soak_gpu.log is a run of 400 (800 tests, 19 minutes).
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


def _check(codec, streams, exp, caps, label):
    b = DeviceBatch(codec, streams, caps)
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
        _both_kernels(lambda label: _check("deflate", [streams[i] for i in keep], [exp[i] for i in keep], caps, "seed %d, %s" % (seed, label)))
    finally:
        O.lib.refcpu_set_max_output(1 << 30)
