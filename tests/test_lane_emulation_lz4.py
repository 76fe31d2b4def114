"""CPU tier: the per-lane LZ4 block decoder (swcompression_amd/csrc/lz4_lane.h) built for the host vs
the oracle's restatement of LZ4.process(block:_:) (reference Sources/LZ4/LZ4.swift:332-413)."""
import random

import _emu as E
import _oracle as O
import _streams as S


def test_valid_blocks_and_dictionaries():
    cases = S.lz4_blocks_valid()
    exp = [O.lz4_block(z, d) for z, d in cases]
    assert all(e[0] == 0 for e in exp)
    res = E.lz4_block([z for z, _ in cases], [len(e[1]) for e in exp], [d for _, d in cases])
    for r, e, (z, _) in zip(res, exp, cases):
        assert r[:2] == e and r[2] == len(z)


def test_fuzz_status_and_bytes():
    O.lib.refcpu_set_max_output(1 << 22)
    cases = S.lz4_blocks_fuzz()
    exp = [O.lz4_block(z, d) for z, d in cases]
    rnd = random.Random(1)
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    res = E.lz4_block([cases[i][0] for i in keep], [max(len(exp[i][1]), 1) + rnd.choice([0, 0, 3, 64]) for i in keep],
                      [cases[i][1] for i in keep])
    for r, i in zip(res, keep):
        assert r[0] == exp[i][0], cases[i][0].hex()
        if exp[i][0] == 0:
            assert r[1] == exp[i][1]
    O.lib.refcpu_set_max_output(1 << 30)


def test_capacity_reports_required_size():
    cases = [c for c in S.lz4_blocks_valid() if c[1] is None]
    exp = [O.lz4_block(z) for z, _ in cases]
    res = E.lz4_block([z for z, _ in cases], [len(e[1]) // 2 for e in exp])
    for r, e in zip(res, exp):
        assert r[0] == 901 and r[3] == len(e[1]) and r[1] == e[1][:len(e[1]) // 2]


import pytest


@pytest.mark.parametrize("order", [0, 1, 2])
def test_large_blocks_sub_chunk_rounds(order):
    """Blocks large enough for the sub-chunk-parallel rounds of lz4_wave.h (64 lanes x 256 bytes per round): every payload
    class, then damaged copies (byte flips -- offset 0, offsets beyond the output, broken length extensions -- and
    truncations) whose status and output must still be the oracle's.  The thread order of the emulated SIMT regions must
    not matter."""
    from swcompression_amd import corpus
    O.lib.refcpu_set_max_output(1 << 23)
    rnd = random.Random(5 + order)
    cases = []
    for kind, size in (("text", 300000), ("mix", 400000), ("rep", 200000), ("zero", 100000), ("rand", 70000), ("text", 1 << 20)):
        z = corpus.lz4_block(corpus.PAYLOADS[kind](size, 31))
        cases.append(z)
        if kind in ("text", "mix") and size <= 400000:
            for _ in range(6):
                b = bytearray(z)
                for _ in range(rnd.choice([1, 1, 3])):
                    b[rnd.randrange(len(b))] = rnd.choice([0, 0, 255, rnd.randrange(256)])
                cases.append(bytes(b))
            cases.append(z[:rnd.randrange(len(z) // 2, len(z))])
    exp = [O.lz4_block(z) for z in cases]
    E.set_order(order)
    try:
        res = E.lz4_block(cases, [max(len(e[1]), 1) + 64 for e in exp], misalign=order)
    finally:
        E.set_order(0)
    for i, (r, e) in enumerate(zip(res, exp)):
        if e[0] == 901:
            continue
        assert r[0] == e[0], "case %d" % i
        if e[0] == 0:
            assert r[1] == e[1] and r[2] == len(cases[i]), "case %d" % i
    O.lib.refcpu_set_max_output(1 << 30)
