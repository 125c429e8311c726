"""The cipher layer of the oracle (oracle/cipher_model.c) against published vectors, the reference's own known-answer
test and the reference's encrypted golden archives (Argon2id key derivation included)."""
import hashlib
import os

import pytest

from conftest import GOLDEN, golden


def test_fips197_vectors(codec):
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    assert codec.aes_block(bytes(range(32)), pt).hex() == "8ea2b7ca516745bfeafc49904b496089"      # FIPS-197 C.3
    assert codec.aes_block(bytes(range(16)), pt).hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"      # FIPS-197 C.1
    assert codec.aes_block(bytes(range(32)), bytes.fromhex("8ea2b7ca516745bfeafc49904b496089"), decrypt=True) == pt


def test_sp800_38a_ctr_and_cbc(codec):
    # NIST SP 800-38A F.5.5 (CTR-AES256) and F.2.5 (CBC-AES256), first two blocks
    key = bytes.fromhex("603deb1015ca71be2b73aef0857d77811f352c073b6108d72d9810a30914dff4")
    pt = bytes.fromhex("6bc1bee22e409f96e93d7e117393172aae2d8a571e03ac9c9eb76fac45af8e51")
    ct = codec.aes_ctr(key, bytes.fromhex("f0f1f2f3f4f5f6f7f8f9fafbfcfdfeff"), pt)
    assert ct.hex() == "601ec313775789a5b7a7f504bbf3d228f443e3ca4d62b59aca84e990cacaf5c5"
    cbc = codec.aes_cbc_encrypt(key, bytes.fromhex("000102030405060708090a0b0c0d0e0f"), pt)
    assert cbc[:32].hex() == "f58c4c04d6e5f1ba779eabfb5f7bfbd69cfc4e967edb808d679f777bc6702c7d" and len(cbc) == 48
    assert codec.aes_cbc_decrypt(key, bytes.fromhex("000102030405060708090a0b0c0d0e0f"), cbc) == pt


def test_reference_ctr_known_answer(codec):
    # lib/src/cipher/stream/write.rs:78-98 (Aes128Ctr64LEWriter)
    ct = bytes([51, 87, 18, 30, 187, 90, 41, 70, 139, 216, 97, 70, 117, 150, 206, 61, 165, 155, 222, 228, 45, 204, 6, 20, 222, 169,
                85, 54, 141, 138, 93, 192, 202, 212])
    pt = b"hello world! this is my plaintext."
    assert codec.aes_ctr(bytes([0x42] * 16), bytes([0x24] * 16), pt, flavor=1) == ct
    # "CTR mode can be used with streaming messages": pieces at arbitrary stream positions
    out = b"".join(codec.aes_ctr(bytes([0x42] * 16), bytes([0x24] * 16), ct[i:i + 3], pos=i, flavor=1) for i in range(0, 34, 3))
    assert out == pt


def test_ctr128be_carry_and_positions(codec):
    key, iv = bytes(range(32)), bytes([0xFF] * 16)          # counter wraps through all 16 bytes
    data = os.urandom(100)
    whole = codec.aes_ctr(key, iv, data)
    assert whole[:16] == bytes(a ^ b for a, b in zip(data[:16], codec.aes_block(key, iv)))
    assert whole[16:32] == bytes(a ^ b for a, b in zip(data[16:32], codec.aes_block(key, bytes(16))))
    assert b"".join(codec.aes_ctr(key, iv, data[i:i + 7], pos=i) for i in range(0, 100, 7)) == whole


def test_hashes_against_hashlib(codec):
    d = bytes(range(256)) * 5
    for n in (0, 1, 127, 128, 129, 1000):
        assert codec.blake2b(d[:n]) == hashlib.blake2b(d[:n]).digest()
        assert codec.blake2b(d[:n], 20) == hashlib.blake2b(d[:n], digest_size=20).digest()
        assert codec.sha256(d[:n]) == hashlib.sha256(d[:n]).digest()
    assert codec.pbkdf2_sha256(b"pass", b"saltsalt", 1000) == hashlib.pbkdf2_hmac("sha256", b"pass", b"saltsalt", 1000, 32)
    assert codec.pbkdf2_sha256(b"p" * 100, b"s", 3, 70) == hashlib.pbkdf2_hmac("sha256", b"p" * 100, b"s", 3, 70)


def test_cbc_roundtrip_all_tail_lengths(codec):
    key, iv = os.urandom(32), os.urandom(16)
    for n in range(0, 50):
        d = os.urandom(n)
        c = codec.aes_cbc_encrypt(key, iv, d)
        assert len(c) == (n // 16 + 1) * 16 and codec.aes_cbc_decrypt(key, iv, c) == d
    with pytest.raises(ValueError):
        codec.aes_cbc_decrypt(key, iv, bytes(15))


RAW = os.path.join(GOLDEN, "raw")
# sha256[:16] of the one payload whose raw file the reference keeps only inside its fixtures
ICON_BMP = ("raw/images/icon.bmp", 4194442)


def _check_entry(name, plain):
    path = os.path.join(GOLDEN, name)
    if os.path.exists(path):
        with open(path, "rb") as f:
            assert plain == f.read(), name
    else:
        assert (name, len(plain)) == ICON_BMP


@pytest.mark.parametrize("fixture", ["zstd_aes_ctr.pna", "zstd_aes_cbc.pna"])
def test_reference_encrypted_archives_decrypt(codec, pf, fixture):
    """lib/tests/extract_compatibility.rs:120-141: password "password"; Argon2id (m=4096,t=3,p=1) -> AES-256 key, IV = first
    16 bytes of the data stream, then zstd."""
    _, items = pf.read_archive(golden(fixture))
    assert len(items) == 9
    for it in items:
        phsf = [d for ty, d in it.chunks if ty == b"PHSF"][0].decode()
        key = codec.derive_key_from_phsf(phsf, b"password")
        comp = codec.decrypt_payload(it.encryption, it.cipher_mode, key, it.data)
        _check_entry(it.name, codec.decode_payload(it.compression, comp, 8 << 20))
    # a wrong password yields a different key: CTR decrypts to garbage that is not a zstd stream
    it = items[1]
    phsf = [d for ty, d in it.chunks if ty == b"PHSF"][0].decode()
    bad = codec.derive_key_from_phsf(phsf, b"passw0rd")
    with pytest.raises(ValueError):
        codec.decode_payload(it.compression, codec.decrypt_payload(it.encryption, it.cipher_mode, bad, it.data), 8 << 20)


@pytest.mark.parametrize("fixture", ["solid_zstd_aes_ctr.pna", "solid_zstd_aes_cbc.pna"])
def test_reference_encrypted_solid_archives_decrypt(codec, pf, fixture):
    """lib/tests/extract_solid_compatibility.rs: SHED | PHSF | SDAT* (IV first) | SEND."""
    _, items = pf.read_archive(golden(fixture))
    assert len(items) == 1 and isinstance(items[0], pf.ParsedSolid)
    so = items[0]
    phsf = [d for ty, d in so.chunks if ty == b"PHSF"][0].decode()
    key = codec.derive_key_from_phsf(phsf, b"password")
    plain = codec.decode_payload(so.compression, codec.decrypt_payload(so.encryption, so.cipher_mode, key, so.data), 16 << 20)
    inner = pf.read_solid_inner(plain)
    assert len(inner) == 9
    for e in inner:
        _check_entry(e.name, e.data)


def test_encrypted_entry_writer_reads_back(codec, pf):
    key, iv = bytes(range(32)), bytes(range(16, 32))
    body = codec.model_compress(b"hello " * 1000)
    phsf = "$pbkdf2-sha256$i=1000,l=32$c2FsdHNhbHRzYWx0"
    rec = pf.write_encrypted_file_entry(pf.COMPRESSION_ZSTD, 1, pf.CIPHER_MODE_CTR, "a/b.txt", phsf, iv, codec.aes_ctr(key, iv, body), 6000)
    arc = pf.write_archive_header() + rec + pf.finalize_archive()
    (it,) = pf.read_archive(arc)[1]
    assert [ty for ty, _ in it.chunks] == [b"FHED", b"fSIZ", b"PHSF", b"FDAT", b"FDAT", b"FEND"]
    assert (it.encryption, it.cipher_mode, it.raw_file_size) == (1, 1, 6000) and it.chunks[3][1] == iv
    assert codec.derive_key_from_phsf(phsf, b"pw") == hashlib.pbkdf2_hmac("sha256", b"pw", b"saltsaltsalt", 1000, 32)
    assert codec.decode_payload(2, codec.decrypt_payload(1, 1, key, it.data), 1 << 20) == b"hello " * 1000


# ----------------------------------------------------------------------------- cipher mode 2: GCM STREAM

def test_hkdf_rfc5869_and_reference_stream_key_vector(codec):
    # lib/src/cipher/aead.rs:321-330 (RFC 5869 test case 1) and :399-405 (K_STREAM_FHED, "produced by an RFC 5869 implementation
    # outside this crate")
    okm = codec.hkdf_sha256(bytes([0x0b] * 22), bytes(range(13)), bytes(range(0xf0, 0xfa)), 42)
    assert okm.hex() == "3cb25f25faacd57a90434f64d0362f2a2d2d0a90cf1a5a4c5db02d56ecc4c5bf34007208d5b887185865"
    ks = codec.derive_stream_key(b"master_key", bytes([0x42] * 32), bytes([0x5A] * 7), 0x01020304, b"FHED", b"header", b"phsf")
    assert ks.hex() == "b88e2edc07538bdd2b9afff57fb0d3433a1f4498d22a5911507e6827590fadb5"
    # entry_context layout, aead.rs:370-381
    ctx = codec.entry_context(bytes([0x5A] * 7), 0x01020304, b"FHED", b"test_header", b"test_phsf")
    assert ctx == b"PNA-STREAM-v1" + hashlib.sha256(b"FHEDtest_header").digest() + hashlib.sha256(b"test_phsf").digest() + bytes([0x5A] * 7) + bytes([1, 2, 3, 4])
    assert codec.segment_nonce(bytes([3] * 7), 0x01020304, True) == bytes([3] * 7) + bytes([1, 2, 3, 4, 1])


def test_gcm_nist_vectors(codec):
    # NIST GCM test cases 13 and 14 (AES-256, 96-bit IV)
    c, t = codec.aes_gcm(bytes(32), bytes(12), b"")
    assert c == b"" and t.hex() == "530f8afbc74536b9a963b4f1c4cb738b"
    c, t = codec.aes_gcm(bytes(32), bytes(12), bytes(16))
    assert c.hex() == "cea7403d4d606b6e074ec5d3baf39d18" and t.hex() == "d0d1c8a799996bf0265b98b5d48ab919"
    assert codec.aes_gcm(bytes(32), bytes(12), c, tag=t) == bytes(16)
    with pytest.raises(ValueError):
        codec.aes_gcm(bytes(32), bytes(12), c, tag=bytes(16))


def test_gcm_stream_segmentation(codec):
    # lib/src/cipher/gcm.rs:417-462: KEY = [7; 32], PREFIX = [3; 7], segment size 4
    key, prefix = bytes([7] * 32), bytes([3] * 7)
    for plain, nseg in [(b"", 1), (b"abc", 1), (b"abcd", 1), (b"abcdefgh", 2), (b"abcdefghi", 3)]:
        ct = codec.gcm_stream_encrypt(key, prefix, 4, plain)
        assert len(ct) == len(plain) + 16 * nseg
        assert codec.gcm_stream_decrypt(key, prefix, 4, ct) == plain
    ct = codec.gcm_stream_encrypt(key, prefix, 4, b"abcdefgh")
    with pytest.raises(ValueError):                            # swapped segments: authentication failure (gcm.rs:762-770)
        codec.gcm_stream_decrypt(key, prefix, 4, ct[20:40] + ct[:20])
    with pytest.raises(ValueError):                            # removed final segment (gcm.rs:782-788)
        codec.gcm_stream_decrypt(key, prefix, 4, ct[:20])


@pytest.mark.parametrize("fixture", ["zstd_aes_gcm.pna", "solid_zstd_aes_gcm.pna"])
def test_reference_gcm_archives_decrypt(codec, pf, fixture):
    """The reference's GCM STREAM fixtures: stream header (key confirmation), per-entry HKDF stream key bound to the FHED / SHED
    chunk and the PHSF string, segment tags -- then zstd."""
    _, items = pf.read_archive(golden(fixture))
    for it in items:
        assert it.cipher_mode == 2
        phsf = [d for ty, d in it.chunks if ty == b"PHSF"][0]
        km = codec.derive_key_from_phsf(phsf.decode(), b"password")
        hty, hdat = it.chunks[0]
        comp = codec.decrypt_payload_gcm(km, it.data, hty, hdat, phsf)
        plain = codec.decode_payload(it.compression, comp, 16 << 20)
        if isinstance(it, pf.ParsedSolid):
            inner = pf.read_solid_inner(plain)
            assert len(inner) == 9
            for e in inner:
                _check_entry(e.name, e.data)
        else:
            _check_entry(it.name, plain)
    it = items[0]
    phsf = [d for ty, d in it.chunks if ty == b"PHSF"][0]
    with pytest.raises(ValueError, match="wrong password"):
        codec.decrypt_payload_gcm(codec.derive_key_from_phsf(phsf.decode(), b"passw0rd"), it.data, it.chunks[0][0], it.chunks[0][1], phsf)
