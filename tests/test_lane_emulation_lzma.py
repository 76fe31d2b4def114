"""CPU tier: the LZMA / LZMA2 wave decoder (swcompression_amd/csrc/lzma_wave.h) built for the host as a
single logical lane vs the oracle (reference Sources/LZMA/*.swift, Sources/LZMA2/*.swift)."""
import _emu as E
import _oracle as O
import _streams as S


def test_lzma2_valid():
    cases = S.lzma2_valid()
    res = E.lzma2([z for z, _, _ in cases], [max(len(x), 1) for _, _, x in cases], [db for _, db, _ in cases])
    for r, (z, db, x) in zip(res, cases):
        assert r[:3] == (0, x, len(z))
        assert O.lzma2(z, db) == (0, x, len(z))


def test_lzma2_fuzz():
    O.lib.refcpu_set_max_output(1 << 24)
    cases = S.lzma2_fuzz()
    exp = [O.lzma2(z, db) for z, db in cases]
    res = E.lzma2([z for z, _ in cases], [max(len(e[1]), 1) + 64 for e in exp], [db for _, db in cases])
    for r, e, (z, db) in zip(res, exp, cases):
        if e[0] == 901:
            continue
        assert r[0] == e[0], (z[:16].hex(), db)
        if e[0] == 0:
            assert r[1] == e[1] and r[2] == e[2]
    O.lib.refcpu_set_max_output(1 << 30)


def test_lzma_alone_valid_and_raw_fuzz():
    O.lib.refcpu_set_max_output(1 << 24)
    ins, props, dss, szs, exp = [], [], [], [], []
    for z, x in S.lzma_alone_valid():
        b = z[0]
        p = (b % 9, (b // 9) % 5, (b // 9) // 5)
        ds = int.from_bytes(z[1:5], "little")
        for size in (-1, len(x)):
            ins.append(z[13:]); props.append(p); dss.append(ds); szs.append(size)
            exp.append(O.lzma_raw(z[13:], p[0], p[1], p[2], ds, size))
            if size == -1 and p[2] < 4:
                assert exp[-1][:2] == (0, x)
            # pb == 4: a VALID stream can reach state 11 / posState 15, where the reference indexes
            # probabilities[432] (LZMADecoder.swift:186-187) and traps
    for body, p, ds, size in S.lzma_raw_fuzz():
        ins.append(body); props.append(p); dss.append(ds); szs.append(size)
        exp.append(O.lzma_raw(body, p[0], p[1], p[2], ds, size))
    res = E.lzma(ins, [max(len(e[1]), 1) + 300 for e in exp], props, dss, szs)
    for i, (r, e) in enumerate(zip(res, exp)):
        if e[0] == 901:
            continue
        assert r[0] == e[0], (i, props[i], dss[i], szs[i], ins[i][:16].hex())
        if e[0] == 0:
            assert r[1] == e[1] and r[2] == e[2]
    O.lib.refcpu_set_max_output(1 << 30)


def test_capacity_is_reported():
    cases = S.lzma2_valid(sizes=(5000, 70000))[:6]
    res = E.lzma2([z for z, _, _ in cases], [len(x) // 2 for _, _, x in cases], [db for _, db, _ in cases])
    assert all(r[0] == 901 for r in res)
