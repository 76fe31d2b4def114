"""GPU tier: BZip2 through the C ABI on the MI355X vs the oracle (reference Sources/BZip2)."""
import bz2
import random

import numpy as np
import pytest

import _oracle as O
import _streams as S
import swcompression_amd as swc
from swcompression_amd import corpus
from swcompression_amd.batch import DeviceBatch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fused", "team"])
def stage3(request):
    """Every test with stage 3a inside the block's wavefront and with stage 3a as kernels of its own (bzip2_team.h), whatever
    the number of blocks of the launch."""
    from swcompression_amd import _lib
    lib = _lib.load()
    assert lib.swc_set_tuning(b"bzip2_team_walk", 2 if request.param == "team" else 0) == 0
    yield
    lib.swc_set_tuning(b"bzip2_team_walk", 1)


def _same(data):
    st, out, cons = O.bzip2(data)
    if st == 0:
        assert swc.BZip2.decompress(data) == out
    else:
        with pytest.raises(swc.SWCError) as ei:
            swc.BZip2.decompress(data)
        assert ei.value.status == st, data[:24].hex()
        if st == 210:
            assert ei.value.data == out                                   # wrongCRC carries the output so far


def test_valid_streams_single_and_multi_block():
    for z, x in S.bzip2_valid():
        assert swc.BZip2.decompress(z) == x


def test_fuzz_same_status_as_reference():
    O.lib.refcpu_set_max_output(1 << 24)
    for z in S.bzip2_fuzz():
        if O.bzip2(z)[0] != 901:
            _same(z)
    O.lib.refcpu_set_max_output(1 << 30)


def test_trivial_inputs_and_crc_errors():
    for data in (b"", b"\x00", bytes(1 << 16)):
        _same(data)                                                       # BZip2Tests.swift:61-79
    x = corpus.p_text(250000, 2)
    z = bytearray(bz2.compress(x, 1))                                     # 3 blocks
    z[10] ^= 1                                                            # first block's stored CRC
    with pytest.raises(swc.BZip2Error) as ei:
        swc.BZip2.decompress(bytes(z))
    assert ei.value.case == "wrongCRC" and ei.value.data == O.bzip2(bytes(z))[1] and len(ei.value.data) > 0
    z = bytearray(bz2.compress(x, 1)); z[-1] ^= 0x10                      # combined CRC at the end
    _same(bytes(z))


def test_unchecked_final_code_length():
    """The last symbol's code length is never range-checked (BZip2.swift:185): 21..26 bits decode, more is trap-class."""
    d = b"banana bandana cabana " * 3
    for eob_len in (20, 21, 22, 26, 27, 0, -3):
        _same(S.bzip2_crafted(d, [1, 2, 3, 4, 5, 6, 7, eob_len]))
    _same(S.bzip2_crafted(d, [2, 2, 3, 3, 3, 4, 4, 23]))
    assert swc.BZip2.decompress(S.bzip2_crafted(d, [1, 2, 3, 4, 5, 6, 7, 24])) == d


def test_multi_stream():
    x = corpus.p_text(5000, 4)
    two = bz2.compress(x[:10]) + bz2.compress(x[10:])
    assert swc.BZip2.multi_decompress(two) == [x[:10], x[10:]] == O.bzip2_multi(two)[1]
    assert swc.BZip2.decompress(two) == x[:10]                            # decompress(data:) stops after the first stream (App. A B8)


def test_config4_shape_900k_blocks():
    """BASELINE.json config 4 shape at reduced count: level-9 streams of one ~900 kB block each, decoded as ONE batch."""
    n = 48
    plains = [corpus.p_text(899000, 500 + i) for i in range(n)]
    streams = [bz2.compress(p, 9) for p in plains]
    assert all(s.count(bytes.fromhex("314159265359")) >= 1 for s in streams)
    b = DeviceBatch("bzip2_block", streams, [899000 + 64] * n, extra=[112] * n,
                    dict_values=[int.from_bytes(s[10:14], "big") for s in streams])
    b.launch(sync=True)
    r = b.results()
    assert (r["status"] == 0).all() and (r["out_len"] == 899000).all()
    for i in range(n):
        assert b.output(i, 899000) == plains[i]


def test_symbol_loop_boundaries():
    """Inputs built to walk the stage-1 symbol loop through its special cases (bzip2_block.h: hot_symbols / phase): alphabets
    below and above 64 symbols (list positions >= 64, table indices >= 64), zero-run lengths around what the staging register
    takes (a run enters it only if it leaves a lane for every symbol the group of 50 still has), runs across group boundaries,
    blocks that end inside a group, exact capacities (the last groups run with the per-symbol tests)."""
    rnd = random.Random(2024)
    cases = []
    for trial in range(120):
        alpha = rnd.choice([1, 2, 3, 17, 63, 64, 65, 130, 256])
        syms = bytes(rnd.sample(range(256), alpha))
        parts = []
        total = rnd.choice([1, 49, 50, 51, 99, 500, 5000, 40000])
        while sum(len(p) for p in parts) < total:
            k = rnd.random()
            if k < 0.4:
                parts.append(bytes([rnd.choice(syms)]) * rnd.choice([1, 2, 3, 4, 5, 12, 13, 14, 15, 16, 49, 50, 51, 63, 64, 65, 200, 3000]))
            elif k < 0.7:
                w = bytes(rnd.choice(syms) for _ in range(rnd.randrange(1, 9)))
                parts.append(w * rnd.randrange(1, 60))
            else:
                parts.append(bytes(rnd.choice(syms) for _ in range(rnd.randrange(1, 300))))
        x = b"".join(parts)
        cases.append((bz2.compress(x, rnd.choice([1, 9])), x))
    for z, x in cases:
        assert swc.BZip2.decompress(z) == x
    # the same blocks as ONE batch at exact capacities (single-block streams only)
    one = [(z, x) for z, x in cases if z.count(bytes.fromhex("314159265359")) == 1 and len(x) > 0]
    b = DeviceBatch("bzip2_block", [z for z, _ in one], [len(x) for _, x in one], extra=[112] * len(one),
                    dict_values=[int.from_bytes(z[10:14], "big") for z, _ in one])
    b.launch(sync=True)
    r = b.results()
    for i, (z, x) in enumerate(one):
        assert int(r["status"][i]) == 0 and int(r["out_len"][i]) == len(x), (i, int(r["status"][i]))
        assert b.output(i, len(x)) == x


def _decode_both(z):
    """(status, data) of BZip2.decompress under the assembly loop and under its C++ twin (tuning "bzip2_hot_cxx")."""
    from swcompression_amd import _lib
    lib = _lib.load()
    res = []
    for cxx in (0, 1):
        assert lib.swc_set_tuning(b"bzip2_hot_cxx", cxx) == 0
        try:
            res.append((0, swc.BZip2.decompress(z)))
        except swc.SWCError as e:
            res.append((e.status, e.data))
        finally:
            lib.swc_set_tuning(b"bzip2_hot_cxx", 0)
    return res


def test_assembly_loop_against_its_cxx_twin():
    """hot_symbols_isa (hand-written gfx950 assembly, the default) and hot_symbols_cxx (what the CPU tier runs) are two
    instantiations of the block kernel: same status and same bytes on streams of three encoders -- libbz2 (six refined tables),
    the reference encoder restated (tables from 50 symbols, lengths up to 20 bits for the deep symbols) and the engine's own (one
    table, up to 17 bits) -- on long zero runs (the staging limit of the run path), and on streams cut short at every byte of
    the last 200 (a group that ends inside the window the look-ahead loads guard, bzip2_block.h: br.next + 184 > br.n)."""
    rnd = random.Random(17)
    skew = bytes(min(255, int(rnd.expovariate(0.05))) for _ in range(200000))            # geometric byte values: long codes
    runs = b"".join(bytes([rnd.randrange(3)]) * rnd.choice([1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 1000, 70000]) for _ in range(300))
    plains = [corpus.p_text(300000, 61), corpus.p_mix(200000, 62), corpus.p_rand(50000, 63), skew, runs, bytes(range(256)) * 300,
              b"ab" * 50000, corpus.p_text(1000, 64), b"a"]
    streams = []
    for x in plains:
        streams.append((bz2.compress(x, 1), x))
        streams.append((bz2.compress(x, 9), x))
        streams.append((swc.BZip2.compress(x, 1), x))
        if len(x) <= 300000:
            streams.append((O.bzip2_compress(x, 1), x))
    for z, x in streams:
        a, b = _decode_both(z)
        assert a == b == (0, x)
    O.lib.refcpu_set_max_output(1 << 24)
    for z, x in streams[:12:3] + streams[-3:]:
        for cut in list(range(1, 200, 1)):
            if cut >= len(z):
                break
            a, b = _decode_both(z[:-cut])
            assert a == b, (len(z), cut, a[0], b[0])
    O.lib.refcpu_set_max_output(1 << 30)
