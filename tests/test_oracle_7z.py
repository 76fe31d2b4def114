"""Oracle pinning for SevenZipFolder.unpack (reference Sources/7-Zip/7zFolder.swift:138-194): oracle/rc_zip.c against the
payloads the system encoders were given, for every coder the reference supports and chains of two."""
import _oracle as O
import _sevenzip as Z


def test_supported_coders_and_chains():
    for name, packed, chain, plain in Z.folders(seed=3):
        st, out = O.sevenzip_folder(packed, chain)
        assert (st, out) == (0, plain), name


def test_error_taxonomy():
    got = {name: O.sevenzip_folder(packed, chain) for name, packed, chain in Z.damaged(seed=4)}
    assert got["wrong-size"] == (861, b"") and got["wrong-size-after-delta"] == (861, b"")
    assert got["multi-stream"][0] == 862 and got["unsupported"][0] == 863 and got["encryption"][0] == 864
    assert got["delta-no-props"][0] == 865
    assert got["lzma2-no-props"][0] == 401 and got["lzma2-two-props"][0] == 401      # LZMA2Error.wrongDictionarySize
    assert got["lzma-short-props"][0] == 301 and got["lzma-bad-props-byte"][0] == 301  # LZMAError.wrongProperties
    # (a flipped bit inside a Deflate stream may still decode to the declared size: unpack() itself checks no CRC)
    assert all(st != 0 and out == b"" for name, (st, out) in got.items() if name != "deflate-flipped-bit")
