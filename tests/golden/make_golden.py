"""Regenerates tests/golden/ref_inline_vectors.json.

The reference's fixture submodule (Tests/Test Files) is empty in this checkout, so the only golden
vectors that survive are the ones written INLINE in its XCTest sources.  This script restates them:
byte literals are copied as data; the two vectors the reference builds with BitByteData's
`LsbBitWriter` (Tests/DeflateTests.swift:104-146 and :148-193) are rebuilt here bit by bit from the
same sequence of writer calls.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os


class LsbBitWriter:
    """LSB-first bit writer with BitByteData.LsbBitWriter's call shapes."""
    def __init__(self):
        self.bits = []

    def write_bits(self, bits):
        self.bits.extend(bits)

    def write(self, number, bits_count):
        self.bits.extend((number >> i) & 1 for i in range(bits_count))

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    @property
    def data(self):
        assert len(self.bits) % 8 == 0
        return bytes(sum(b << i for i, b in enumerate(self.bits[k:k + 8])) for k in range(0, len(self.bits), 8))


def _dynamic_header(w):
    w.write_bits([1, 0, 1])            # last block, dynamic Huffman
    w.write(29, 5)                      # 286 literal/length codes
    w.write(1, 5)                       # 2 distance codes
    w.write(14, 4)                      # 18 code length codes
    w.write(0, 3); w.write(3, 3); w.write(2, 3)
    for _ in range(10):
        w.write(0, 3)
    w.write(2, 3); w.write(0, 3); w.write(3, 3); w.write(0, 3); w.write(2, 3)


def deflate_all_zero_litlen():         # DeflateTests.swift:104-146  -> throws
    w = LsbBitWriter()
    _dynamic_header(w)
    w.write(1, 2); w.write(127, 7)      # symbol 18, repeat 138
    w.write(1, 2); w.write(127, 7)      # symbol 18, repeat 138
    w.write(7, 3); w.write(7, 3)        # symbol 17, repeat 10
    w.write(0, 2); w.write(0, 2)        # two distance code lengths of 1
    w.align()
    return w.data


def deflate_empty_distance_tree():     # DeflateTests.swift:148-193  -> Data([0])
    w = LsbBitWriter()
    _dynamic_header(w)
    w.write(3, 3)                       # code length 2 for symbol 0
    w.write(1, 2); w.write(127, 7)      # symbol 18, repeat 138
    w.write(1, 2); w.write(106, 7)      # symbol 18, repeat 117
    w.write(2, 2)                       # code length 3 for symbol 256
    w.write(1, 2); w.write(20, 7)       # symbol 18, repeat 31
    w.write(0, 2)                       # literal 0
    w.write(2, 3)                       # end of block
    w.align()
    return w.data


def main():
    vectors = {
        "source": "tsolomko/SWCompression 4.9.0 Tests/*.swift inline vectors (fixtures submodule is empty)",
        "deflate": [
            {"name": "testSymbol16First", "ref": "Tests/DeflateTests.swift:35-48",
             "input": bytes([0b00000101, 0b00000000, 0b10100010, 0b00001101]).hex(), "expect": "throws"},
            {"name": "testCodeLengthsOverCopy/zero-repeat", "ref": "Tests/DeflateTests.swift:55-66",
             "input": bytes([0b00000101, 0b00000000, 0b10100010, 0b11101101, 0b11111111, 0b11111111, 0b00000001]).hex(), "expect": "throws"},
            {"name": "testCodeLengthsOverCopy/copy-previous", "ref": "Tests/DeflateTests.swift:68-81",
             "input": bytes([0b00000101, 0b00000000, 0b10100010, 0b11101101, 0b11111111, 0b10110011, 0b00000101]).hex(), "expect": "throws"},
            {"name": "testCodeLengthsAllZero/code-length-tree", "ref": "Tests/DeflateTests.swift:84-98",
             "input": bytes([0b00000101, 0b00000000, 0b00000000, 0b00000]).hex(), "expect": "throws"},
            {"name": "testCodeLengthsAllZero/literal-tree", "ref": "Tests/DeflateTests.swift:104-146",
             "input": deflate_all_zero_litlen().hex(), "expect": "throws"},
            {"name": "testCodeLengthsAllZero/distance-tree", "ref": "Tests/DeflateTests.swift:148-193",
             "input": deflate_empty_distance_tree().hex(), "expect": "00"},
        ],
        "xxh32": [  # Tests/XxHash32Tests.swift:12-59, seed 0
            {"input": "", "hash": "02cc5d05"}, {"input": "a", "hash": "550d7456"}, {"input": "abc", "hash": "32d153ff"},
            {"input": "message digest", "hash": "7c948494"}, {"input": "abcdefghijklmnopqrstuvwxyz", "hash": "63a14d5f"},
            {"input": "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", "hash": "9c285e64"},
            {"input": "1234567890" * 8, "hash": "9c05f475"},
        ],
        "sha256": [  # Tests/Sha256Tests.swift:12-64 (Sources/Common/Sha256.swift:28-142, the XZ check type 0x0A)
            {"input": "", "hash": "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"},
            {"input": "a", "hash": "ca978112ca1bbdcafac231b39a23dc4da786eff8147c4e72b9807785afee48bb"},
            {"input": "abc", "hash": "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"},
            {"input": "message digest", "hash": "f7846f55cf23e14eebeab5b4e1550cad5b509e3348fbc4efa3a1413d393cb650"},
            {"input": "abcdefghijklmnopqrstuvwxyz", "hash": "71c480df93d6ae2f1efad1447c66c9525e316218cf51fc8d9ed832f2daf18b73"},
            {"input": "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789", "hash": "db4bfcbd4da0cd85a60c3c37d3fbd8805c77f15fc6b1fdfe614ee0a7c8fdb4c0"},
            {"input": "1234567890" * 8, "hash": "f371bc4a311f2b009eef952dd83ca80e2b60026c8e935592d0f9c308453c813e"},
        ],
        # inputs every codec must reject (Tests/BZip2Tests.swift:61-79, LzmaTests.swift:42-57, LZ4Tests.swift:87-108,
        # GzipTests.swift:174-188, XzTests.swift:112-120, ZlibTests.swift:45-57)
        "must_throw": {"empty": "", "single_zero": "00"},
        "lz4_specific": {"empty": "truncated", "single_zero": "truncated", "zeros_1mb": "corrupted"},
        # round-trip payloads used by the reference's compression tests (DeflateCompressionTests.swift:29-39,77-85;
        # BZip2CompressionTests.swift:31-49,87-95; LZ4CompressionTests.swift:29-47)
        "roundtrip_strings": ["ban", "banana", "abaaba", "abracadabra", "cabbage", "baabaabac", "AAAAAAABBBBCCCD", "AAAAAAA",
                              "qwertyuiopasdfghjklzxcvbnm1234567890"],
        # ... and LZ4CompressionTests.swift:162-172 (testTrickySequence: "match index was wrongly used as cyclical index"; the
        # last ten bytes only allow a sequence with a match)
        "roundtrip_bytes": ["2e202e202e2020", "000100010000010001", "616c202d43202d43202d2d01020304050607080900"],
        "lz4_tricky_sequence": {"ref": "Tests/LZ4CompressionTests.swift:162-172", "input": "616c202d43202d43202d2d01020304050607080900",
                                "options": {"independent_blocks": False, "block_checksums": True, "content_checksum": True, "content_size": True}},
        "magic": {"gzip": "1f8b", "bzip2_block": "314159265359", "bzip2_eos": "177245385090", "xz_header": "fd377a585a00",
                  "xz_footer": "595a", "lz4_frame": "04224d18", "lz4_legacy": "02214c18", "lz4_skippable_first": "502a4d18"},
    }
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_inline_vectors.json")
    with open(out, "w") as f:
        json.dump(vectors, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
