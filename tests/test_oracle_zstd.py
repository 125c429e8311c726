"""The RFC 8878 decoder and the deterministic encoder model (oracle/zstd_dec.c, oracle/zstd_model.c)."""
import os

import pytest

from conftest import GOLDEN, golden


def _raw(rel):
    with open(os.path.join(GOLDEN, rel), "rb") as f:
        return f.read()


def test_decoder_on_reference_fixtures(pf, codec):
    """Every zstd payload of the reference's golden archives decodes to the reference's raw files."""
    for arc in ("zstd.pna", "zstd_with_raw_file_size.pna", "zstd_keep_all.pna"):
        _, items = pf.read_archive(golden(arc))
        for it in items:
            if it.kind != 0:
                continue
            plain = codec.zstd_decompress(it.data, 8 << 20)
            if it.raw_file_size is not None:
                assert len(plain) == it.raw_file_size
            p = os.path.join(GOLDEN, it.name)
            if os.path.exists(p):
                assert plain == _raw(it.name), it.name
    _, items = pf.read_archive(golden("solid_zstd.pna"))
    plain = codec.zstd_decompress(items[0].data, 8 << 20)
    inner = pf.read_solid_inner(plain)
    assert len(plain) == 4305625 and len(inner) == 9
    for e in inner:
        if os.path.exists(os.path.join(GOLDEN, e.name)):
            assert e.data == _raw(e.name)


def test_codec_known_answers(codec):
    # frames found in the reference's fixtures
    assert codec.zstd_decompress(bytes.fromhex("28B52FFD2000010000"), 16) == b""
    assert codec.zstd_decompress(bytes.fromhex("28B52FFD0080510000746578742066696C650A"), 64) == b"text file\n"
    assert codec.zlib_decompress(bytes.fromhex("789C030000000001")) == b""
    assert codec.model_compress(b"") == bytes.fromhex("28B52FFD2000010000")


def test_decoder_agrees_with_libzstd(codec):
    if codec.system_libzstd() is None:
        pytest.skip("no system libzstd on this host")
    cases = [b"", b"a", b"a" * 1000, os.urandom(5000), codec.corpus_file(0, 1, 70000), codec.corpus_file(1, 2, 300000),
             bytes(200000), _raw("raw/images/icon.png"), _raw("raw/images/icon.svg")]
    for d in cases:
        for lvl in (1, 3, 19):
            c = codec.libzstd_compress(d, lvl)
            assert codec.zstd_decompress(c, len(d)) == d
    multi = codec.libzstd_compress(cases[4], 3) + codec.libzstd_compress(cases[3], 3) + codec.libzstd_compress(b"", 3)
    assert codec.zstd_decompress(multi, 100000) == cases[4] + cases[3]
    assert codec.zstd_frame_count(multi, 100000) == 3


def test_decoder_rejects_garbage(codec):
    with pytest.raises(ValueError):
        codec.zstd_decompress(b"\x00\x01\x02\x03\x04", 64)
    good = codec.model_compress(codec.corpus_file(0, 3, 5000))
    with pytest.raises(ValueError):
        codec.zstd_decompress(good[:-3], 8192)


MODEL_CASES = {
    "empty": lambda c: b"", "one": lambda c: b"a", "abc": lambda c: b"abc", "a1000": lambda c: b"a" * 1000,
    "zeros4k": lambda c: bytes(4096), "rnd5000": lambda c: c.corpus_file(2, 1, 5000), "txt4k": lambda c: c.corpus_file(1, 1, 4096),
    "txt64k": lambda c: c.corpus_file(1, 2, 65536), "txt300k": lambda c: c.corpus_file(0, 3, 300000),
    "txt1m+": lambda c: c.corpus_file(0, 4, (1 << 20) + 12345), "rnd200k": lambda c: c.corpus_file(2, 0, 200000),
    "zero1m": lambda c: bytes(1 << 20), "ab": lambda c: b"ab" * 70000, "abc_long": lambda c: (b"abcdefghij" * 20000)[:131072 + 77],
    "t2047": lambda c: c.corpus_file(0, 9, 2047), "t2049": lambda c: c.corpus_file(0, 9, 2049),
    "png": lambda c: _raw("raw/images/icon.png"), "nest": lambda c: _raw("raw/pna/nest.pna"),
    "mixed": lambda c: c.corpus_file(0, 7, 200000) + c.corpus_file(2, 7, 100000) + bytes(150000) + c.corpus_file(1, 7, 300000),
}


@pytest.mark.parametrize("name", sorted(MODEL_CASES))
@pytest.mark.parametrize("flags", [0, 1, 2, 7, 15])
def test_model_round_trip(codec, name, flags):
    """Model output is a valid zstd stream for both independent decoders, for every feature subset."""
    d = MODEL_CASES[name](codec)
    p = codec.default_params()
    p.flags = flags
    c = codec.model_compress(d, p)
    assert codec.zstd_decompress(c, len(d)) == d
    if codec.system_libzstd() is not None:
        assert codec.libzstd_decompress_stream(c, len(d)) == d
    assert len(c) <= len(d) + 6 * (len(d) // (1 << 20) + 1) + 3 * (len(d) // (1 << 17) + 2) + 16
    assert c == codec.model_compress(d, p)                      # deterministic


def test_model_frames_and_ratio(codec):
    d = codec.corpus_file(0, 5, 3 << 20)
    c = codec.model_compress(d)
    assert codec.zstd_frame_count(c, len(d)) == 3               # one frame per 1 MiB segment
    assert c[:6] == bytes.fromhex("28B52FFD0050")               # FHD 0x00, window descriptor 0x50 (1 MiB)
    assert len(d) / len(c) > 2.3                                # enwik-style corpus; libzstd-3 gets ~2.83


@pytest.mark.parametrize("name", sorted(MODEL_CASES))
def test_deflate_model_inflates_with_zlib(codec, name):
    """The zlib/deflate encoder model against an independent RFC 1950/1951 decoder (stdlib zlib)."""
    import zlib
    d = MODEL_CASES[name](codec)
    c = codec.deflate_model_compress(d)
    assert c[:2] == b"\x78\x9c"
    assert codec.zlib_decompress(c) == d
    assert len(c) <= len(d) + 23 * (len(d) // (1 << 17) + 1) + 16
    assert c == codec.deflate_model_compress(d)
    if name == "empty":
        assert c == bytes.fromhex("789C030000000001")          # the reference's empty zlib stream (tests/golden/deflate.pna)
    assert c[-4:] == zlib.adler32(d).to_bytes(4, "big")


def test_deflate_model_ratio(codec):
    import zlib
    d = codec.corpus_file(1, 3, 65536)                          # BASELINE.json configs[0] shape: 64 KiB random-text
    assert len(codec.deflate_model_compress(d)) < 1.05 * len(zlib.compress(d, 6))
