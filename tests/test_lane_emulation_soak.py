"""CPU tier: the streams test_gpu_soak.py builds, through the host emulation of the same device code (tests/host_emu): LZ4 blocks
assembled sequence by sequence around the boundaries of the record formats (both record modes of the parse), Deflate streams
over spliced text, and damaged copies of both -- the oracle's status, bytes, consumed input and length; the lanes of every
parallel region in forward, reverse and shuffled order, the copier in each of its window configurations."""
import random

import pytest

import _emu as E
import _oracle as O
import _soak as K


def _compare(res, exp, streams, what):
    for k, (r, e) in enumerate(zip(res, exp)):
        label = "%s, stream %d (%d bytes in, oracle status %d)" % (what, k, len(streams[k]), e[0])
        assert r[0] == e[0], "status %d: %s" % (r[0], label)
        if e[0] == 0:
            assert r[1] == e[1] and r[2] == e[2] and r[3] == len(e[1]), label


@pytest.fixture
def bounded_oracle():
    O.lib.refcpu_set_max_output(1 << 24)
    yield
    O.lib.refcpu_set_max_output(1 << 30)
    E.set_order(0)
    E.lib.emu_set_copier(1)
    E.lib.emu_set_lz4_record_mode(2)


@pytest.mark.parametrize("seed", range(6))
def test_lz4_blocks_built_sequence_by_sequence(seed, bounded_oracle):
    rnd = random.Random(0xE4C5A34 + seed)
    streams = []
    for i in range(100):
        z, p = K.random_lz4_block(rnd, rnd.choice([1, 40, 700, 5000, 66000, 140000, 300000]))
        st, out = O.lz4_block(z)
        assert out == p[:len(out)] and (st != 0 or out == p), "the builder and the oracle disagree (seed %d, block %d)" % (seed, i)
        streams.append(z)
    streams += [K.damage(rnd, streams[rnd.randrange(len(streams))]) for _ in range(100)]
    exp = [O.lz4_block(z) + (len(z),) for z in streams]
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    ins, want = [streams[i] for i in keep], [exp[i] for i in keep]
    caps = [max(len(e[1]), 1) + rnd.choice([0, 0, 0, 3, 64]) for e in want]
    E.set_order(seed % 3)
    for mode in (2, 1):
        E.lib.emu_set_lz4_record_mode(mode)
        for copier in (1, 3):   # the LZ4 window / the Deflate window
            E.lib.emu_set_copier(copier)
            _compare(E.lz4_block(ins, caps, misalign=seed % 16), want, ins, "seed %d, record mode %d, copier %d" % (seed, mode, copier))


@pytest.mark.parametrize("seed", range(6))
def test_deflate_streams_over_spliced_text(seed, bounded_oracle):
    rnd = random.Random(0xEDEF1A7E + seed)
    streams = [K.random_deflate_stream(rnd, rnd.choice([0, 1, 9, 300, 4000, 65536, 70000, 200000]), 9000 * seed + i) for i in range(100)]
    streams += [K.damage(rnd, streams[rnd.randrange(len(streams))]) for _ in range(100)]
    exp = [O.deflate(z) for z in streams]
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    ins, want = [streams[i] for i in keep], [exp[i] for i in keep]
    caps = [max(len(e[1]), 1) + rnd.choice([0, 0, 0, 5]) for e in want]
    E.set_order(seed % 3)
    try:
        for team in (0, 1):         # one wavefront per stream / a team of wavefronts
            E.lib.emu_set_deflate_team(team)
            for copier in (1, 2, 0):    # the Deflate window, the LZ4 window, the workgroup resolver
                E.lib.emu_set_copier(copier)
                _compare(E.inflate(ins, caps, misalign=seed % 16), want, ins, "seed %d, team %d, copier %d" % (seed, team, copier))
    finally:
        E.lib.emu_set_deflate_team(0)


@pytest.mark.parametrize("seed", range(4))
def test_lzma2_units_of_random_encoder_settings(seed, bounded_oracle):
    rnd = random.Random(0xE7A4C + seed)
    units = [K.random_lzma2_unit(rnd, rnd.choice([0, 1, 9, 300, 4000, 70000, 200000]), 5000 * seed + i) for i in range(64)]
    units += [(K.damage(rnd, z), db) for z, db in (units[rnd.randrange(len(units))] for _ in range(48))]
    exp = [O.lzma2(z, db) for z, db in units]
    keep = [i for i, e in enumerate(exp) if e[0] != 901]
    ins, want = [units[i][0] for i in keep], [exp[i] for i in keep]
    caps = [max(len(e[1]), 1) + rnd.choice([0, 0, 64]) for e in want]
    E.set_order(seed % 3)
    for mode in (1, 0):         # LDS as a cache of the literal coders / all coders in LDS
        _compare(E.lzma2(ins, caps, [units[i][1] for i in keep], mode=mode), want, ins, "seed %d, model layout %d" % (seed, mode))
