"""GPU tier: the container callers that batch through the engine -- swc_unarchive_many for every
archive kind and swc_zip_get_entries_data -- against the oracle's single-archive restatements of the reference."""
import struct
import zipfile

import numpy as np
import pytest

import _oracle as O
import _streams as S
import _zips as Z
import swcompression_amd as swc
from swcompression_amd import _lib, corpus
from swcompression_amd.zipcontainer import ZipContainer

pytestmark = pytest.mark.gpu


def _payloads(seed, k=12):
    sizes = [0, 1, 100, 5000, 65536, 70001, 300000, 17, 4096, 1 << 20, 33333, 250000]
    return [corpus.p_mix(sizes[i % len(sizes)], seed + i) for i in range(k)]


def _damage(b, at=None):
    d = bytearray(b)
    d[len(d) // 2 if at is None else at] ^= 0x20
    return bytes(d)


def _check(kind, archives, oracle_fn, carries, max_launches=3):
    lib = _lib.load()
    before = lib.swc_stat(b"launches")
    got = swc.unarchive_many(kind, archives)
    if max_launches:
        assert lib.swc_stat(b"launches") - before <= max_launches, "%s: the set must share its launches" % kind
    assert len(got) == len(archives)
    for i, (a, (st, data)) in enumerate(zip(archives, got)):
        est, edata = oracle_fn(a)[:2]
        assert st == est, "%s archive %d: status %d, oracle %d" % (kind, i, st, est)
        if est == 0 or est in carries:
            assert data == edata, "%s archive %d differs" % (kind, i)
        else:
            assert data == b""


def test_many_gzip_members():
    ps = _payloads(100)
    arch = [corpus.gzip_member(p, bgzf=(i % 3 == 0 and len(p) < 60000)) for i, p in enumerate(ps)]
    arch += [_damage(arch[4]), arch[5][:-3], _damage(arch[6], at=len(arch[6]) - 6), b"", b"\x1f\x8b\x08"]
    _check("gzip", arch, O.gzip_unarchive, {605})


def test_many_zlib_and_raw_deflate():
    ps = _payloads(200)
    z = [corpus.zlib_stream(p) for p in ps]
    z += [_damage(z[4]), z[5][:-2], _damage(z[6], at=len(z[6]) - 2), b"\x78"]
    _check("zlib", z, O.zlib_unarchive, {705})
    r = [corpus.deflate_raw(p) for p in ps]
    r += [_damage(r[4]), r[6][:len(r[6]) // 2], b""]
    _check("deflate", r, O.deflate, set())


def test_many_lzma2():
    ps = _payloads(300, 8)
    a = [bytes([corpus.lzma2_dict_byte(1 << 20)]) + corpus.lzma2_raw(p) for p in ps]
    a += [_damage(a[4]), a[5][:len(a[5]) // 2], b"", b"\x29"]
    _check("lzma2", a, O.lzma2_data, set())


def test_many_lz4_frames():
    O.lib.refcpu_set_max_output(1 << 24)
    a = [f for _, f, d, _ in S.lz4_frames() if d is None]
    a += [corpus.lz4f_frame(p, 4 + i % 4, False, i % 2 == 0, i % 3 == 0, i % 5 == 0) for i, p in enumerate(_payloads(400, 10))]
    a += [_damage(a[-1]), a[-2][:len(a[-2]) - 5], _damage(a[-3], at=len(a[-3]) - 2), b"", b"\x04\x22\x4d\x18"]
    _check("lz4", a, O.lz4, {503}, max_launches=0)


def test_many_bzip2_streams():
    ps = _payloads(500, 8)
    a = [corpus.bzip2_stream(p, level=1 + i % 9) for i, p in enumerate(ps)]
    a.append(corpus.bzip2_stream(corpus.p_text(450000, 77), level=1))   # several blocks
    a += [_damage(a[4]), a[5][:len(a[5]) - 7], b"BZh9", b""]
    _check("bzip2", a, O.bzip2, {210})


def test_many_xz():
    ps = _payloads(600, 6)
    a = [corpus.xz_stream(p) for p in ps]
    a += [_damage(a[3]), a[4][:-5]]
    _check("xz", a, O.xz_unarchive, {807}, max_launches=0)


@pytest.mark.parametrize("method", [zipfile.ZIP_STORED, zipfile.ZIP_DEFLATED, zipfile.ZIP_BZIP2, zipfile.ZIP_LZMA])
@pytest.mark.parametrize("streamed", [False, True])
def test_zip_open_matches_zipfile_and_oracle(method, streamed):
    c = Z.make_zip(method, streamed=streamed, seed=21)
    assert ZipContainer.open(c) == Z.reference_extract(c)
    files = [h for h in ZipContainer.helpers(c) if not h["is_dir"]]
    assert ZipContainer.entries_data(c, files) == [O.zip_entry(c, h) for h in files]


def test_zip_mixed_methods_one_call():
    c = Z.mixed_zip(seed=31)
    assert ZipContainer.open(c) == Z.reference_extract(c)


def test_zip_damaged_entries_match_oracle():
    c = Z.make_zip(zipfile.ZIP_DEFLATED, seed=22, with_dir=False)
    hs = ZipContainer.helpers(c)
    h = hs[2]
    variants = [dict(h, crc32=h["crc32"] ^ 1), dict(h, uncomp_size=h["uncomp_size"] + 1), dict(h, comp_size=h["comp_size"] - 1),
                dict(h, method=9), dict(h, data_offset=len(c) + 5), dict(h, uncomp_size=5), dict(hs[4], uncomp_size=1 << 40)]
    assert ZipContainer.entries_data(c, variants) == [O.zip_entry(c, v) for v in variants]
    dmg = bytearray(c)
    dmg[h["data_offset"] + 40] ^= 0x10
    assert ZipContainer.entries_data(bytes(dmg), hs) == [O.zip_entry(bytes(dmg), x) for x in hs]
    lz = Z.make_zip(zipfile.ZIP_LZMA, seed=23, with_dir=False)
    hl = ZipContainer.helpers(lz)
    dl = bytearray(lz)
    dl[hl[2]["data_offset"] + 4] = 230      # invalid LZMA properties byte
    dl[hl[3]["data_offset"] + 60] ^= 0x08
    assert ZipContainer.entries_data(bytes(dl), hl) == [O.zip_entry(bytes(dl), x) for x in hl]
    # a damaged size field must not become a terabyte allocation: the declared size is only a starting capacity
    huge = [dict(hl[1], uncomp_size=1 << 40), dict(hl[4], uncomp_size=(1 << 31) + 5)]
    assert ZipContainer.entries_data(lz, huge) == [O.zip_entry(lz, x) for x in huge]
    # ZipContainer.open raises wrongCRC carrying the entries so far (ZipContainer.swift:52-53)
    bad = bytearray(c)
    cd = bad.rfind(b"PK\x01\x02")
    struct.pack_into("<I", bad, cd + 16, 0x12345678)     # CRC-32 of the LAST entry in its central-directory record ...
    lh = hs[-1]["data_offset"] - 30 - len(hs[-1]["name"])
    struct.pack_into("<I", bad, lh + 14, 0x12345678)     # ... and in its local header
    with pytest.raises(swc.ZipError) as ei:
        ZipContainer.open(bytes(bad))
    assert ei.value.status == 853 and len(ei.value.data) == len(hs)


def test_7z_folders_one_call():
    """swc_7z_unpack_folders (SevenZipFolder.unpack, 7zFolder.swift:138-194): every supported coder, two-coder chains and
    the error taxonomy, all folders in ONE call, against the oracle."""
    import _sevenzip as Z7
    from swcompression_amd.sevenzip import SevenZipFolder
    good = Z7.folders(seed=3)
    bad = Z7.damaged(seed=4)
    lib = _lib.load()
    before = lib.swc_stat(b"launches")
    got = SevenZipFolder.unpack_many([(p, c) for _, p, c, _ in good] + [(p, c) for _, p, c in bad])
    assert lib.swc_stat(b"launches") - before <= 12, "folders must share their launches (one per codec and stage, plus retries)"
    for (name, packed, chain, plain), r in zip(good, got):
        assert r == (0, plain), name
    for (name, packed, chain), r in zip(bad, got[len(good):]):
        assert r == O.sevenzip_folder(packed, chain), name
    assert SevenZipFolder.unpack(good[1][1], good[1][2]) == good[1][3]
    with pytest.raises(swc.SevenZipError) as ei:
        SevenZipFolder.unpack(bad[0][1], bad[0][2])
    assert ei.value.case == "wrongSize"


def test_bgzf_members_share_one_launch():
    """GzipArchive.multiUnarchive (GzipArchive.swift:62-77) on BGZF data: the 'BC' extra field locates every member, so
    they decode in one launch; damaged files fall back to the sequential walk and must match the oracle exactly."""
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 95))
    parts = [corpus.p_mix(int(rng.integers(1, 50000)), 7000 + i) for i in range(300)] + [b""]
    members = [corpus.gzip_member(p, bgzf=True) for p in parts]
    data = b"".join(members)
    lib = _lib.load()
    before = lib.swc_stat(b"launches")
    assert swc.GzipArchive.multi_unarchive(data) == parts
    assert lib.swc_stat(b"launches") - before <= 2, "BGZF members must share their launch"
    assert O.gzip_multi_unarchive(data) == (0, parts)
    # damaged variants: identical outcome to the oracle's strictly sequential restatement
    off = [0]
    for m in members:
        off.append(off[-1] + len(m))
    variants = []
    v = bytearray(data); v[off[7] + 30] ^= 0x08; variants.append(bytes(v))                 # inside member 7's Deflate stream
    v = bytearray(data); v[off[9] + 16] ^= 0x01; variants.append(bytes(v))                 # BSIZE of member 9
    v = bytearray(data); v[off[11] - 6] ^= 0x01; variants.append(bytes(v))                 # CRC-32 of member 10
    v = bytearray(data); v[off[12] - 2] ^= 0x01; variants.append(bytes(v))                 # ISIZE of member 11
    variants.append(data[:off[20]] + corpus.gzip_member(parts[20]) + data[off[21]:])        # one member without the field
    variants.append(data[:-5])                                                             # truncated
    for k, bad in enumerate(variants):
        st, exp = O.gzip_multi_unarchive(bad)
        if st == 0:
            assert swc.GzipArchive.multi_unarchive(bad) == exp, k
        else:
            with pytest.raises(swc.SWCError) as ei:
                swc.GzipArchive.multi_unarchive(bad)
            assert ei.value.status == st, k
            if st == 605:
                assert ei.value.data == exp, k


def test_single_result_as_a_view():
    import lzma as pylzma
    x = corpus.p_text(3 << 20, 77)
    a = pylzma.compress(x, format=pylzma.FORMAT_XZ, check=pylzma.CHECK_CRC64, preset=1)
    v = swc.XZArchive.unarchive(a, view=True)
    assert isinstance(v, memoryview) and v.readonly and v == x
    w = swc.XZArchive.unarchive(a, view=True)                   # the first result is still alive and untouched
    assert v == x and w == x
    del v, w
    assert swc.XZArchive.unarchive(a) == x


def test_bgzf_views_and_recycled_results():
    """The members of a BGZF file as VIEWS into the C result (what a Swift shim gets with Data(bytesNoCopy:)): same bytes as the
    copies, alive as long as one view is; results of 4 MiB and more are parked by swc_free and taken over by the next call of
    about that size -- a stale byte from the call before must never show."""
    import gc
    a = [corpus.p_text(65536, 9100 + i) for i in range(100)]
    b = [corpus.p_mix(65536, 9300 + i) for i in range(100)]
    da = b"".join(corpus.gzip_member(p, bgzf=True) for p in a)
    db = b"".join(corpus.gzip_member(p, bgzf=True) for p in b)
    for _ in range(3):
        va = swc.GzipArchive.multi_unarchive(da, views=True)
        assert [bytes(v) for v in va] == a and all(v.readonly for v in va)
        keep = va[37]
        del va
        gc.collect()
        assert swc.GzipArchive.multi_unarchive(db) == b            # a different result of the same size in between
        assert bytes(keep) == a[37]                                  # the kept view still owns its buffer
        del keep
        gc.collect()
        assert swc.GzipArchive.multi_unarchive(db, views=True)[99] == b[99]
        assert swc.GzipArchive.multi_unarchive(da) == a


def test_lz4_multi_frame_one_launch():
    """LZ4.multiDecompress (LZ4.swift:116-146) on a buffer of many frames: block sizes are in the headers, so all frames
    share one launch; anything unusual falls back to the sequential loop and must match the oracle."""
    rng = np.random.Generator(np.random.PCG64(0x5C0DE + 99))
    parts = [corpus.p_mix(int(rng.integers(0, 300000)), 9000 + i) for i in range(40)]
    frames = [corpus.lz4f_frame(p, 4 + i % 4, False, i % 2 == 0, i % 3 == 0, i % 5 == 0) for i, p in enumerate(parts)]
    skip = struct.pack("<II", 0x184D2A51, 3) + b"abc"
    data = b"".join(f + (skip if i % 7 == 0 else b"") for i, f in enumerate(frames))
    O.lib.refcpu_set_max_output(1 << 24)
    lib = _lib.load()
    before = lib.swc_stat(b"launches")
    assert swc.LZ4.multi_decompress(data) == parts
    assert lib.swc_stat(b"launches") - before <= 2, "the frames must share their launch"
    assert O.lz4_multi(data) == (0, parts)
    variants = []
    v = bytearray(data); v[len(frames[0]) + len(skip) + 40] ^= 0x10; variants.append(bytes(v))     # inside frame 1
    v = bytearray(data); v[len(data) - 2] ^= 0x01; variants.append(bytes(v))                      # last frame's tail
    variants.append(data + corpus.lz4f_frame(parts[3], 4, True))                                   # a dependent-block frame at the end
    variants.append(data[:-3])
    for k, bad in enumerate(variants):
        st, exp = O.lz4_multi(bad)
        if st == 0:
            assert swc.LZ4.multi_decompress(bad) == exp, k
        else:
            with pytest.raises(swc.SWCError) as ei:
                swc.LZ4.multi_decompress(bad)
            assert ei.value.status == st, k


def test_many_over_a_device_list():
    """swc_unarchive_many_devices: the archive list cut into one range per listed device, each decoded by its own host
    thread.  One GPU here, so the list names device 0 several times -- same code path, same results as the one-device call
    and as the oracle, in the archives' order; ranges may be empty (more devices than archives)."""
    ps = _payloads(900, k=40)
    for kind, enc, ora in (("gzip", corpus.gzip_member, O.gzip_unarchive), ("bzip2", corpus.bzip2_stream, O.bzip2),
                           ("lz4", corpus.lz4_frame, O.lz4), ("xz", corpus.xz_stream, O.xz_unarchive)):
        arch = [enc(p) for p in ps]
        arch[7] = _damage(arch[7])
        one = swc.unarchive_many(kind, arch)
        for devs in ([0], [0, 0], [0, 0, 0, 0, 0]):
            assert swc.unarchive_many(kind, arch, devices=devs) == one
        for (st, data), a in zip(one, arch):
            est, edata = ora(a)[:2]
            assert st == est and (est != 0 or data == edata)
    few = [corpus.gzip_member(ps[3]), corpus.gzip_member(ps[4])]
    assert [d for _, d in swc.unarchive_many("gzip", few, devices=[0] * 8)] == [ps[3], ps[4]]
    assert swc.unarchive_many("gzip", [], devices=[0, 0]) == []
    with pytest.raises(swc.DeviceError):
        swc.unarchive_many("gzip", few, devices=[0, 99])       # no such gfx950 device: loud, nothing decoded elsewhere
    with pytest.raises(swc.SWCError):
        swc.unarchive_many("gzip", few, devices=[])
