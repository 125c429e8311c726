"""The container restatement (oracle/pna_format.py) against the reference's own known answers and fixtures."""
import hashlib
import struct
import zlib

import pytest

from conftest import golden


def test_empty_archive_is_golden(pf):
    # lib/src/archive/write.rs:792-798: write_header + finalize == resources/test/empty.pna (40 bytes)
    assert pf.write_archive_header() + pf.finalize_archive() == golden("empty.pna")
    assert golden("empty.pna").hex() == ("89504e410d0a1a0a" "00000008" "41484544" "0000000000000000" "47755bb5"
                                         "00000000" "41454e44" "6bf6486d")


def test_crc_known_answers(pf):
    assert pf.chunk_crc(b"FDAT", bytes([0xAA, 0xBB, 0xCC, 0xDD])) == 0x47F32B10   # lib/src/format/chunk.rs:30-32
    assert pf.chunk_crc(b"FDAT", bytes([1, 2, 3])) == 2776590148                   # lib/src/chunk/traits.rs:20-24
    assert pf.chunk_crc(b"AEND", b"") == 0x6BF6486D                                # lib/src/io.rs:176-179
    assert pf.chunk_crc(b"FEND", b"") == 0xF62170D4
    assert pf.chunk_crc(b"SEND", b"") == 0x91E6D779


def test_chunk_layout(pf):
    # lib/src/chunk/write.rs:55-63: "hello world" as FDAT is 23 bytes, length prefix 0x0000000B
    c = pf.write_chunk(b"FDAT", b"hello world")
    assert len(c) == 23 and c[:4] == bytes([0, 0, 0, 0x0B]) and c[4:8] == b"FDAT"
    # lib/src/chunk/write.rs:76-92: max chunk size 4 -> chunks of 4, 4, 3 ... at offsets 0, 16, 32
    s = pf.chunk_stream_writer(b"FDAT", [b"hello world"], 4)
    sizes, pos = [], 0
    while pos < len(s):
        (n,) = struct.unpack_from(">I", s, pos); sizes.append(n); pos += 12 + n
    assert sizes == [4, 4, 3] and len(s) == 3 * 12 + 11


def test_header_layouts(pf):
    assert pf.archive_header_bytes(1, 2, 3) == bytes([1, 2, 0, 0, 0, 0, 0, 3])     # lib/src/archive/header.rs:64-74
    assert pf.file_entry_header(pf.COMPRESSION_ZSTD, "name") == bytes([0, 0, 0, 2, 0, 1]) + b"name"
    assert pf.file_entry_header(pf.COMPRESSION_DEFLATE, "n") == bytes([0, 0, 0, 1, 0, 1]) + b"n"
    assert pf.solid_header_bytes(pf.COMPRESSION_ZSTD) == bytes([0, 0, 2, 0, 1])
    assert pf.entry_header_bytes(0, 0, 0, 1, "test.txt") == bytes([0, 0, 0, 0, 0, 1]) + b"test.txt"  # lib/src/entry.rs:1366-1375
    assert pf.dir_entry_header("d") == bytes([0, 0, 1, 0, 0, 0]) + b"d"
    assert pf.fsiz_bytes(0) == b"" and pf.fsiz_bytes(10) == b"\x0a" and pf.fsiz_bytes(51475) == bytes([0xC9, 0x13])


def test_flatten_writer(pf):
    # lib/src/util/io.rs:209-233: writes coalesce up to max_chunk_size, then split
    assert pf.flatten_writer([b"abc", b"def"]) == [b"abcdef"]
    assert pf.flatten_writer([b"abc", b"defgh"], 4) == [b"abcd", b"efgh"]
    assert pf.flatten_writer([b"", b"a" * 9], 4) == [b"aaaa", b"aaaa", b"a"]
    assert pf.flatten_writer([]) == []


# EntryName::sanitize = normalize_utf8path, then only Normal components: the reference's own vectors
# (lib/src/entry/name.rs:19-23,52-55,143-145,483-490,533-544,593-617; lib/src/util/utf8path.rs:38-58 through the filter)
SANITIZE_VECTORS = [
    ("uer/bin", "uer/bin"), ("/user/bin", "user/bin"), ("/user/bin/", "user/bin"), ("../user/bin/", "user/bin"), ("/", ""),
    ("foo.txt", "foo.txt"), ("/foo.txt", "foo.txt"), ("./foo.txt", "foo.txt"), ("../foo.txt", "foo.txt"),
    ("/var/../tmp/./log", "tmp/log"), ("/test/test.txt", "test/test.txt"), ("test/", "test"), ("test/test/", "test/test"),
    ("./test/test.txt", "test/test.txt"), ("../test/test.txt", "test/test.txt"), ("test/../test.txt", "test.txt"),
    ("test//test.txt", "test/test.txt"), ("test///test.txt", "test/test.txt"), ("///test///test.txt", "test/test.txt"),
    ("", ""), ("..", ""), (".", ""), ("../../..", ""), ("/../foo", "foo"), ("./foo", "foo"),
    ("a/b/../../a.txt", "a.txt"), ("a/../../a.txt", "a.txt"), ("a/b/./../a.txt", "a/a.txt"), ("/a//b///", "a/b"), ("a/.", "a"), ("/..", ""),
    ("日本語/テスト.txt", "日本語/テスト.txt"), ("dir\\file.txt", "dir\\file.txt"),      # '\\' separates on Windows only
    ("a/../..//b/./c/../d", "b/d"),
]


def test_name_sanitize(pf):
    for raw, want in SANITIZE_VECTORS:
        assert pf.sanitize_name(raw) == want, raw


@pytest.mark.parametrize("name", ["deflate.pna", "zstd.pna", "zstd_with_raw_file_size.pna"])
def test_reserialise_fixture_byte_exact(pf, name):
    # lib/tests/copy_entries.rs:15-21: read -> re-add reproduces the archive byte for byte
    raw = golden(name)
    n, items = pf.read_archive(raw)
    out = pf.write_archive_header(n)
    for it in items:
        out += pf.write_normal_entry(pf.entry_header_bytes(it.kind, it.compression, it.encryption, it.cipher_mode, it.name),
                                     [d for t, d in it.chunks if t == b"FDAT"], it.raw_file_size)
    out += pf.finalize_archive()
    assert out == raw


def test_reader_rejects_bad_crc_and_unknown_critical_chunk(pf):
    raw = bytearray(golden("zstd.pna"))
    raw[60] ^= 1
    with pytest.raises(ValueError):
        pf.read_archive(bytes(raw))
    # lib/src/entry.rs:1651-1688: unknown critical chunk inside an entry is an error, ancillary is kept
    bad = pf.write_archive_header() + pf.write_chunk(b"FHED", pf.file_entry_header(0, "x")) + pf.write_chunk(b"XXXX", b"") \
        + pf.write_chunk(b"FEND") + pf.finalize_archive()
    with pytest.raises(ValueError):
        pf.read_archive(bad)
    ok = pf.write_archive_header() + pf.write_chunk(b"FHED", pf.file_entry_header(0, "x")) + pf.write_chunk(b"xxXx", b"") \
        + pf.write_chunk(b"FEND") + pf.finalize_archive()
    assert len(pf.read_archive(ok)[1]) == 1


def test_fixture_payload_hashes(pf, codec):
    # sha256[:16] of the decoded payloads of the reference's golden archives
    want = {"raw/text.txt": "f4796bff42910365", "raw/images/icon.png": "3edb62033736e171", "raw/images/icon.svg": "2379e0d74449ad06",
            "raw/first/second/third/pna.txt": "471bc8030879f3f0", "raw/pna/nest.pna": "84c6c86c8d7de359", "raw/pna/empty.pna": "d379f9324878088c"}
    for arc in ("zstd.pna", "deflate.pna"):
        _, items = pf.read_archive(golden(arc))
        assert len(items) == 9                                      # lib/tests/extract_compatibility.rs asserts 9 entries
        got = {it.name: hashlib.sha256(codec.decode_payload(it.compression, it.data, 8 << 20)).hexdigest()[:16] for it in items}
        for k, v in want.items():
            assert got[k] == v
    assert hashlib.sha256(golden("zstd.pna")).hexdigest()[:16] == "5185fd6444089203"
    assert hashlib.sha256(golden("deflate.pna")).hexdigest()[:16] == "fbb4240ebad45d1c"


def test_split_parts_reproduces_the_reference_multipart_fixture(pf):
    """resources/test/multipart.part{1,2}.pna: joining the parts, merging the FDAT fragments back into the chunk the writer was handed
    and splitting again at the fixture's part size gives the very same parts (SplitParts, lib/src/archive/split_parts.rs:140-173)."""
    p1, p2 = golden("multipart.part1.pna"), golden("multipart.part2.pna")
    body = pf.join_parts([p1, p2])
    assert [(t, len(d)) for t, d in body] == [(b"FHED", 24), (b"FDAT", 464), (b"FDAT", 357), (b"FEND", 0)]
    merged = []
    for t, d in body:
        if merged and t == merged[-1][0] == b"FDAT":
            merged[-1] = (t, merged[-1][1] + d)
        else:
            merged.append((t, d))
    assert pf.split_parts(merged, len(p1)) == [p1, p2]
    with pytest.raises(ValueError):
        pf.split_parts(merged, pf.MIN_SPLIT_PART_BYTES - 1)                  # split_parts.rs:88-94
    with pytest.raises(ValueError):
        pf.join_parts([p2, p1])
    # budget arithmetic of split_parts.rs:14-23
    assert (pf.PART_HEADER_BYTES, pf.SPLIT_ARCHIVE_OVERHEAD_BYTES, pf.MIN_SPLIT_PART_BYTES) == (28, 52, 64)
