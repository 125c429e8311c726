"""C-ABI surface and host logic (no GPU needed): the library loads, exports every declared symbol, the container
writer is byte-exact against the oracle and the reference fixtures, and there is no CPU compression fallback."""
import io
import os
import re

import pytest

from conftest import ROOT, golden


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pna_[a-z0-9_]+)\s*\(", txt)) - {"pna_sink_fn"})


def test_library_exports_every_declared_symbol(pna):
    lib = pna.load_library()
    names = _declared("pna_gpu.h") + _declared("pna_archive.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(set(names)) == sorted(set(pna.EXPORTS))


def test_no_cpu_fallback(pna):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pna.PnaGpuError) as e:
        pna.Context(0)
    assert e.value.code == -1                                   # PNA_E_NODEVICE
    with pytest.raises(pna.PnaGpuError):
        pna.create_archive(None, ["a"], [b"abc"], algo=pna.ALGO_ZSTD)


def test_bound_and_levels(pna):
    assert pna.bound(pna.ALGO_ZSTD, 0) >= 9
    for n in (1, 4096, (1 << 20), (1 << 20) + 1, 5 << 20):
        assert pna.bound(pna.ALGO_ZSTD, n) >= n + 6 * ((n + (1 << 20) - 1) >> 20) + 3 * ((n + (1 << 17) - 1) >> 17)
    # lib/src/compress/zstandard.rs:91-146 and lib/src/compress/deflate.rs:128-187
    assert pna.clamp_level(pna.ALGO_ZSTD) == 3 and pna.clamp_level(pna.ALGO_ZSTD, 100) == 22 and pna.clamp_level(pna.ALGO_ZSTD, 7) == 7
    assert pna.clamp_level(pna.ALGO_DEFLATE) == 6 and pna.clamp_level(pna.ALGO_DEFLATE, 100) == 9
    # a negative custom level goes through `value as u32` and clamps to NINE (lib/src/compress/deflate.rs:89-101), 0 stays 0
    assert pna.clamp_level(pna.ALGO_DEFLATE, -5) == 9 and pna.clamp_level(pna.ALGO_DEFLATE, -1) == 9 and pna.clamp_level(pna.ALGO_DEFLATE, 0) == 0
    assert pna.clamp_level(pna.ALGO_ZSTD, -5) == -5 and pna.clamp_level(pna.ALGO_ZSTD, -(1 << 20)) == -131072      # zstd min_c_level (zstandard.rs:43-57)


def test_crc_and_empty_archive(pna):
    assert pna.crc32(b"FDAT" + bytes([0xAA, 0xBB, 0xCC, 0xDD])) == 0x47F32B10
    assert pna.crc32(b"FDAT\x01\x02\x03") == 2776590148
    assert pna.crc32(b"AEND") == 0x6BF6486D
    data = os.urandom(100003)
    import zlib
    assert pna.crc32(data) == zlib.crc32(data)
    assert pna.crc32(data[5000:], pna.crc32(data[:5000])) == zlib.crc32(data)
    buf = io.BytesIO()
    pna.Archive(buf).finalize()
    assert buf.getvalue() == golden("empty.pna")


@pytest.mark.parametrize("name", ["deflate.pna", "zstd.pna"])
def test_writer_reproduces_reference_fixture(pna, pf, name):
    # lib/tests/copy_entries.rs:15-21 with the C++ writer
    raw = golden(name)
    n, items = pf.read_archive(raw)
    buf = io.BytesIO()
    a = pna.Archive(buf, n)
    for it in items:
        a.add_file(it.name, it.compression, it.raw_file_size, it.data)
    a.finalize()
    assert buf.getvalue() == raw


def test_store_create_matches_oracle_writer(pna, pf, codec):
    # BASELINE.json configs[0] plumbing: 100 x 64 KiB random-text entries through the archive writer (no GPU)
    names = [f"corpus/f{i:05d}.txt" for i in range(100)]
    ents = [codec.corpus_file(1, i, 65536) for i in range(100)]
    arc = pna.create_archive(None, names, ents, algo=pna.ALGO_STORE)
    exp = pf.write_archive_header() + b"".join(
        pf.write_normal_entry(pf.file_entry_header(0, nm), pf.flatten_writer([e]), len(e)) for nm, e in zip(names, ents)) + pf.finalize_archive()
    assert arc == exp
    _, items = pf.read_archive(arc)
    assert [it.name for it in items] == names and all(it.data == e for it, e in zip(items, ents))   # create/entry_order.rs:9-67


def test_deflate_payloads_through_writer(pna, pf, codec):
    """Container path with the reference's other codec: payloads from stdlib zlib (level 6, header 78 9C)."""
    import zlib
    ents = [codec.corpus_file(1, i, 65536) for i in range(8)] + [b""]
    buf = io.BytesIO()
    a = pna.Archive(buf)
    for i, e in enumerate(ents):
        a.add_file(f"d/{i}", pna.ALGO_DEFLATE, len(e), zlib.compress(e, 6))
    a.finalize()
    _, items = pf.read_archive(buf.getvalue())
    assert [codec.decode_payload(it.compression, it.data, 1 << 20) for it in items] == ents
    assert items[0].data[:2] == b"\x78\x9c" and items[-1].data == bytes.fromhex("789C030000000001")


def test_fdat_splitting_and_dirs(pna, pf):
    buf = io.BytesIO()
    a = pna.Archive(buf)
    a.add_dir("some/dir/")
    a.add_file("f", 0, 11, b"hello world", max_chunk_size=4)
    a.finalize()
    _, items = pf.read_archive(buf.getvalue())
    assert items[0].kind == 1 and items[0].name == "some/dir" and [t for t, _ in items[0].chunks] == [b"FHED", b"FEND"]
    assert [d for t, d in items[1].chunks if t == b"FDAT"] == [b"hell", b"o wo", b"rld"]


def test_solid_store(pna, pf):
    names, ents = ["a.txt", "b/c.bin", "empty"], [b"hello" * 50, bytes(range(256)), b""]
    arc = pna.create_archive(None, names, ents, algo=pna.ALGO_STORE, solid=True)
    _, items = pf.read_archive(arc)
    assert isinstance(items[0], pf.ParsedSolid) and items[0].compression == 0
    inner = pf.read_solid_inner(items[0].data)
    assert [(e.name, e.data, e.raw_file_size) for e in inner] == [(n, d, len(d)) for n, d in zip(names, ents)]
    assert pna.inner_entry_bytes("a.txt", ents[0]) == pf.write_normal_entry(pf.file_entry_header(0, "a.txt"), [ents[0]], len(ents[0]))


def test_device_crc_schedule_matches_crc32(pna):
    """The lane schedule of k_frame (front padding, init folded into the type bytes, Z_16320 between tiles, fold tree)
    walked on the host with the kernel's own tables must equal chunk_crc (lib/src/format/chunk.rs:7-12)."""
    import random
    import zlib
    rnd = random.Random(7)
    for n in [0, 1, 3, 4, 5, 59, 60, 61, 64, 16379, 16380, 16381, 16384, 32764, 32765, 50001, (1 << 20) + 7]:
        b = rnd.randbytes(n)
        assert pna.crc_schedule(b) == zlib.crc32(b"FDAT" + b) == pna.crc32(b, pna.crc32(b"FDAT")), n


def test_archive_bound_covers_framing(pna):
    names = ["a/b.txt", "x" * 200]
    lens = [0, 3 << 20]
    b = pna.archive_bound(pna.ALGO_ZSTD, names, lens)
    assert b >= 40 + sum(pna.bound(pna.ALGO_ZSTD, n) + 12 + 6 + len(nm) + 12 + 8 + 12 + 12 for nm, n in zip(names, lens))


def test_host_kdf_matches_hashlib(pna):
    """pna_kdf_pbkdf2_sha256 (the C++ host's hash::pbkdf2_with_salt, lib/src/hash.rs:35-45) against hashlib, and its PHC string."""
    import base64
    import hashlib
    for pw, salt, rounds, kl in [(b"password", bytes(range(16)), 1000, 32), (b"", b"s", 1, 32), (b"p" * 100, b"salt" * 5, 7, 70)]:
        key, phsf = pna.kdf_pbkdf2_sha256(pw, salt, rounds, kl)
        assert key == hashlib.pbkdf2_hmac("sha256", pw, salt, rounds, kl)
        assert phsf == f"$pbkdf2-sha256$i={rounds},l=32$" + base64.b64encode(salt).decode().rstrip("=")


def test_host_kdf_phsf_is_read_by_the_oracle(pna, codec):
    key, phsf = pna.kdf_pbkdf2_sha256(b"password", bytes(range(16)), 1000)
    assert codec.derive_key_from_phsf(phsf, b"password") == key


def test_host_argon2_matches_the_oracle(pna, codec):
    """pna_kdf_argon2 (the C++ host's hash::argon2_with_salt) against the oracle's Argon2, which is pinned by the reference's encrypted
    fixtures: all three kinds, one and several lanes, several passes, odd output lengths."""
    for kind, pw, salt, t, m, p, kl in [(2, b"password", bytes(range(16)), 3, 4096, 1, 32), (2, b"password", b"saltsalt", 1, 50, 1, 32),
                                        (2, b"pw", b"0123456789abcdef", 2, 64, 4, 32), (1, b"pw", b"0123456789abcdef", 2, 64, 2, 24),
                                        (0, b"", b"0123456789abcdef", 1, 32, 2, 70), (2, b"x" * 100, b"y" * 33, 4, 19, 2, 16)]:
        assert pna.kdf_argon2(kind, pw, salt, t, m, p, kl) == codec.argon2(kind, pw, salt, t, m, p, kl), (kind, t, m, p)
    with pytest.raises(pna.PnaGpuError):
        pna.kdf_argon2(2, b"pw", b"salt", 1, 4, 1)             # m < 8 p


def test_split_archive_equals_the_oracle_and_joins_back(pna, pf, codec):
    """pna_split_archive / pna_join_parts (host only): equal to the oracle's SplitParts for a range of part sizes, including sizes
    that cut FDAT chunks several times, leave no room behind a chunk, or fit everything into one part; the reference's own multipart
    fixture is reproduced from its joined image."""
    import os
    from conftest import golden
    payloads = [os.urandom(n) for n in (0, 5, 1000, 3000, 70)]
    arc = pf.write_archive_header()
    for i, pl in enumerate(payloads):
        arc += pf.write_normal_entry(pf.file_entry_header(0, f"d/f{i}.bin"), [pl] if pl else [], len(pl))
    arc += pf.write_solid_entry(0, [os.urandom(500), os.urandom(40)]) + pf.finalize_archive()
    body = pf.archive_body_chunks(arc)
    for size in (64, 65, 77, 100, 128, 200, 564, 1000, 1024, 4096, 1 << 20):
        try:
            want = pf.split_parts(body, size)
        except ValueError:
            with pytest.raises(pna.PnaGpuError):
                pna.split_archive(arc, size)
            continue
        got = pna.split_archive(arc, size)
        assert got == want and all(len(p) <= size for p in got), size
        joined = pna.join_parts(got)
        assert pf.join_parts(got) == pf.archive_body_chunks(joined)
        # the entries read back identically from the joined image
        a, b = pf.read_archive(arc)[1], pf.read_archive(joined)[1]
        assert [(x.data, getattr(x, "name", None)) for x in a] == [(y.data, getattr(y, "name", None)) for y in b]
    with pytest.raises(pna.PnaGpuError):
        pna.split_archive(arc, 63)
    p1, p2 = golden("multipart.part1.pna"), golden("multipart.part2.pna")
    one = pna.join_parts([p1, p2])
    (it,) = pf.read_archive(one)[1]
    whole = pf.write_archive_header() + pf.write_normal_entry(it.chunks[0][1], [it.data], None) + pf.finalize_archive()
    assert pna.split_archive(whole, len(p1)) == [p1, p2]
    with pytest.raises(pna.PnaGpuError):
        pna.join_parts([p2, p1])


def test_entry_names_are_sanitised_like_the_reference(pna, pf):
    """EntryName::sanitize (normalise, then keep the normal components) on the writer side: the FHED of an entry written by the product
    carries the name the reference would write; same vectors as the oracle's test (the reference's name.rs / utf8path.rs)."""
    from test_oracle_container import SANITIZE_VECTORS
    for raw, want in SANITIZE_VECTORS:
        rec = pna.inner_entry_bytes(raw, b"x")
        body = [d for t, d, _ in pf.read_chunks(rec) if t == b"FHED"][0]
        assert body[6:].decode() == want, raw
        assert rec == pf.write_normal_entry(pf.file_entry_header(0, pf.sanitize_name(raw)), [b"x"], 1), raw


def test_seek_to_end_and_entry_listing(pna, pf):
    """Archive::seek_to_end (lib/src/archive/read.rs:412-424, tests :576-604) and the raw-entry walk `pna append` / `pna update` stand on."""
    arc = golden("zstd.pna")
    at, nxt = pna.seek_to_end(arc)
    assert (at, nxt) == pf.seek_to_end(arc) == (len(arc) - 12, False) and arc[at + 4:at + 8] == b"AEND"
    # an archive that continues in another part: ANXT is seen on the way (seek_to_end_detects_next_archive_marker)
    p1 = golden("multipart.part1.pna")
    assert pna.seek_to_end(p1) == pf.seek_to_end(p1) and pna.seek_to_end(p1)[1] is True
    # truncated inside the tail chunk: UnexpectedEof in the reference (seek_to_end_rejects_archives_truncated_inside_the_tail_chunk)
    empty = golden("empty.pna")
    for cut in range(1, 9):
        with pytest.raises(pna.PnaGpuError):
            pna.seek_to_end(empty[:-cut])
        with pytest.raises(ValueError):
            pf.seek_to_end(empty[:-cut])
    with pytest.raises(pna.PnaGpuError):
        pna.seek_to_end(b"not an archive at all, just forty bytes..")
    # the record list: names, kinds and byte ranges that tile the space between AHED and AEND
    ents = pna.list_entries(arc)
    _, items = pf.read_archive(arc)
    assert [n.decode() for n, _, _, _ in ents] == [it.name for it in items]
    pos = 28
    for _, kind, off, ln in ents:
        assert off == pos and arc[off + 4:off + 8] == b"FHED" and arc[off + ln - 8:off + ln - 4] == b"FEND" and kind in (0, 1, 2, 3)
        pos += ln
    assert pos == at
    solid = golden("solid_zstd.pna")
    (s,) = pna.list_entries(solid)
    assert s[1] == -1 and solid[s[2] + 4:s[2] + 8] == b"SHED" and s[2] + s[3] == pna.seek_to_end(solid)[0]


def test_gather_offsets(pna):
    """The ordered gather places rank r's part at the prefix sum of the earlier parts' sizes (the order of ReorderByIndex,
    cli/src/command/core/iter.rs:21-80): pinned here on the CPU, used by pna_gpu_gather_ordered and by bench.py's direct-D2H comparison path."""
    assert pna.gather_offsets([5]) == [0, 5]
    assert pna.gather_offsets([3, 0, 7, 1 << 40, 2]) == [0, 3, 3, 10, 10 + (1 << 40), 12 + (1 << 40)]
    with pytest.raises(pna.PnaGpuError):
        pna.gather_offsets([1 << 63, 1 << 63])
