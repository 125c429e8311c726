"""The cipher stage on the device (k_cipher.hip) against the oracle's cipher layer (oracle/cipher_model.c), through the C ABI.
Integer work: bit-exact."""
import hashlib
import zlib
import os

import pytest

pytestmark = pytest.mark.gpu

KEY = bytes(range(32))
PHSF = "$pbkdf2-sha256$i=1000,l=32$c2FsdHNhbHRzYWx0"


def test_fips197_block_on_device(gpu_ctx, pna):
    """CTR over 16 zero bytes with IV = the FIPS-197 C.3 plaintext yields AES-256(key, IV): the appendix ciphertext."""
    import torch
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    ci = pna.Cipher(KEY, PHSF, pna.MODE_CTR, ivs=bytes.fromhex("00112233445566778899aabbccddeeff"))
    gpu_ctx.cipher_apply_device(ci, buf.data_ptr(), [0], [16])
    out = bytes(buf.cpu().numpy())
    assert out[:16].hex() == "8ea2b7ca516745bfeafc49904b496089" and out[16:] == bytes(48)


def test_sp800_38a_ctr_on_device(gpu_ctx, pna):
    import torch
    key = bytes.fromhex("603deb1015ca71be2b73aef0857d77811f352c073b6108d72d9810a30914dff4")
    pt = bytes.fromhex("6bc1bee22e409f96e93d7e117393172aae2d8a571e03ac9c9eb76fac45af8e51"
                       "30c81c46a35ce411e5fbc1191a0a52eff69f2445df4f9b17ad2b417be66c3710")
    buf = torch.frombuffer(bytearray(pt), dtype=torch.uint8).cuda()
    gpu_ctx.cipher_apply_device(pna.Cipher(key, PHSF, pna.MODE_CTR, ivs=bytes.fromhex("f0f1f2f3f4f5f6f7f8f9fafbfcfdfeff")), buf.data_ptr(), [0], [64])
    assert bytes(buf.cpu().numpy()).hex() == ("601ec313775789a5b7a7f504bbf3d228f443e3ca4d62b59aca84e990cacaf5c5"
                                              "2b0930daa23de94ce87017ba2d84988ddfc9c58db67aada613c2dd08457941a6")


def test_ctr_ranges_equal_oracle(gpu_ctx, pna, codec):
    """Ragged ranges at odd byte offsets, lengths around the 16-byte block and the 256 KiB unit size, a counter that carries through
    all 128 bits; bytes between the ranges stay untouched; applying it twice restores the input."""
    import torch
    lens = [0, 1, 15, 16, 17, 31, 33, 4096, 65537, (256 << 10) - 1, 256 << 10, (256 << 10) + 1, 700001, 3]
    blobs = [os.urandom(n) for n in lens]
    gaps = [os.urandom(1 + (i * 7) % 13) for i in range(len(lens))]
    flat, offs = bytearray(), []
    for b, g in zip(blobs, gaps):
        flat += g; offs.append(len(flat)); flat += b
    flat += bytes(64)
    ivs = bytearray(os.urandom(16 * len(lens)))
    ivs[16 * 8:16 * 9] = bytes([0xFF] * 16)                   # entry 8: IV + 1 wraps to zero
    ivs[16 * 9:16 * 10] = bytes([0] * 7 + [0xFF] * 9)         # carry across the low 64 bits
    buf = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    ci = pna.Cipher(KEY, PHSF, pna.MODE_CTR, ivs=bytes(ivs))
    gpu_ctx.cipher_apply_device(ci, buf.data_ptr(), offs, lens)
    got = bytes(buf.cpu().numpy())
    want = bytearray(flat)
    for i, (o, b) in enumerate(zip(offs, blobs)):
        want[o:o + len(b)] = codec.aes_ctr(KEY, bytes(ivs[16 * i:16 * i + 16]), b)
    assert got == bytes(want)
    gpu_ctx.cipher_apply_device(ci, buf.data_ptr(), offs, lens, decrypt=True)
    assert bytes(buf.cpu().numpy()) == bytes(flat)
    assert gpu_ctx.timing().ms_cipher > 0


def test_cbc_encrypt_equals_oracle(gpu_ctx, pna, codec):
    import torch
    lens = [0, 1, 15, 16, 17, 32, 1000, 65536, 100003] + list(range(40, 72))
    blobs = [os.urandom(n) for n in lens]
    flat, offs = bytearray(), []
    for i, b in enumerate(blobs):
        flat += os.urandom(1 + i % 5); offs.append(len(flat)); flat += b + bytes(16 - len(b) % 16)     # room for the padding block
    flat += bytes(64)
    ivs = os.urandom(16 * len(lens))
    buf = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    gpu_ctx.cipher_apply_device(pna.Cipher(KEY, PHSF, pna.MODE_CBC, ivs=ivs), buf.data_ptr(), offs, lens)
    got = bytes(buf.cpu().numpy())
    want = bytearray(flat)
    for i, (o, b) in enumerate(zip(offs, blobs)):
        c = codec.aes_cbc_encrypt(KEY, ivs[16 * i:16 * i + 16], b)
        want[o:o + len(c)] = c
        assert codec.aes_cbc_decrypt(KEY, ivs[16 * i:16 * i + 16], got[o:o + len(c)]) == b
    assert got == bytes(want)
    with pytest.raises(pna.PnaGpuError) as ei:
        gpu_ctx.cipher_apply_device(pna.Cipher(KEY, PHSF, pna.MODE_CBC, ivs=ivs), buf.data_ptr(), offs, lens, decrypt=True)
    assert ei.value.code == -7


@pytest.mark.parametrize("mode_name", ["ctr", "cbc"])
@pytest.mark.parametrize("algo_name", ["zstd", "deflate"])
def test_encrypted_archive_in_hbm_equals_oracle_writer(gpu_ctx, pna, pf, codec, algo_name, mode_name):
    """pna_gpu_create_archive_enc_device: compress -> cipher -> chunk CRC, all in HBM.  Expected bytes: the plain batch API's payloads,
    encrypted by the oracle, framed by the oracle's container writer (FHED | fSIZ | PHSF | FDAT(iv) | FDAT(ciphertext) | FEND)."""
    import torch
    algo = pna.ALGO_ZSTD if algo_name == "zstd" else pna.ALGO_DEFLATE
    mode = pna.MODE_CTR if mode_name == "ctr" else pna.MODE_CBC
    lens = [0, 1, 5, 4095, 16373, 70001, 131073, 300000, (1 << 20) + 1, 2500000, 12, 65536]
    ents = [codec.corpus_file(i % 2, 150 + i, n) if n else b"" for i, n in enumerate(lens)]
    names = [f"dir{i % 3}/f{i:03d}.txt" for i in range(len(lens))]
    offs, pos = [], 0
    for e in ents:
        offs.append(pos); pos = (pos + len(e) + 15) & ~15
    src = torch.zeros(pos + 8192, dtype=torch.uint8, device="cuda")
    for o, e in zip(offs, ents):
        if e:
            src[o:o + len(e)] = torch.frombuffer(bytearray(e), dtype=torch.uint8).cuda()
    ivs = os.urandom(16 * len(lens))
    ci = pna.Cipher(KEY, PHSF, mode, ivs=ivs)
    cap = pna.archive_enc_bound(algo, names, lens, ci)
    dst = torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")
    total, eoff = gpu_ctx.create_archive_device(names, src.data_ptr(), offs, lens, dst.data_ptr(), cap, algo=algo, cipher=ci)
    got = dst[:total].cpu().numpy().tobytes()
    assert gpu_ctx.timing().ms_cipher > 0
    payloads = gpu_ctx.compress_batch(ents, algo=algo)
    enc = (lambda iv, p: codec.aes_ctr(KEY, iv, p)) if mode == pna.MODE_CTR else (lambda iv, p: codec.aes_cbc_encrypt(KEY, iv, p))
    want = pf.write_archive_header() + b"".join(
        pf.write_encrypted_file_entry(algo, 1, mode, pf.sanitize_name(nm), PHSF, ivs[16 * i:16 * i + 16], enc(ivs[16 * i:16 * i + 16], pl), len(e))
        for i, (nm, pl, e) in enumerate(zip(names, payloads, ents))) + pf.finalize_archive()
    assert len(got) == len(want) and got == want
    assert bytes(dst[total:total + 16].cpu().numpy()) == b"\xA5" * 16
    # read back the way the reference does: CRC of every chunk, key from the PHSF string, IV = first 16 bytes, then decompress
    _, items = pf.read_archive(got)
    for it, e in zip(items, ents):
        assert (it.encryption, it.cipher_mode, it.raw_file_size) == (1, mode, len(e))
        comp = codec.decrypt_payload(it.encryption, it.cipher_mode, KEY, it.data)
        assert codec.decode_payload(algo, comp, len(e) + 64) == e


def test_encrypted_archive_library_ivs_and_device_read_back(gpu_ctx, pna, pf, codec):
    """IVs drawn by the library (ivs = NULL): distinct per entry; the archive is read back on the device: CTR decrypt in place
    (pna_gpu_cipher_apply_device) then the device zstd decoder."""
    import torch
    n, L = 64, 1 << 20
    src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
    gpu_ctx.corpus_fill_device(0, 900, n, L, L, src.data_ptr())
    names = [f"e/{i:04d}" for i in range(n)]
    key = hashlib.pbkdf2_hmac("sha256", b"password", b"saltsaltsalt", 1000, 32)
    ci = pna.Cipher(key, PHSF, pna.MODE_CTR)
    cap = pna.archive_enc_bound(pna.ALGO_ZSTD, names, [L] * n, ci)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    total, eoff = gpu_ctx.create_archive_device(names, src.data_ptr(), [i * L for i in range(n)], [L] * n, dst.data_ptr(), cap, cipher=ci)
    arc = dst[:total].cpu().numpy().tobytes()
    _, items = pf.read_archive(arc)
    assert len(items) == n
    ivs = [it.chunks[3][1] for it in items]
    assert all(ty == b"FDAT" and len(iv) == 16 for (ty, iv) in (it.chunks[3] for it in items)) and len(set(ivs)) == n
    assert codec.derive_key_from_phsf(PHSF, b"password") == key
    # device read-back: locate each ciphertext chunk in the archive buffer, decrypt in place, decode
    p_off, p_len = [], []
    for it, o in zip(items, eoff):
        rel = arc.index(it.chunks[4][1][:32], o)                        # body of the second FDAT chunk
        p_off.append(rel); p_len.append(len(it.chunks[4][1]))
    gpu_ctx.cipher_apply_device(pna.Cipher(key, PHSF, pna.MODE_CTR, ivs=b"".join(ivs)), dst.data_ptr(), p_off, p_len, decrypt=True)
    out = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    gpu_ctx.decompress_batch_device(dst.data_ptr(), p_off, p_len, out.data_ptr(), [i * L for i in range(n)], [L] * n)
    assert torch.equal(out, src[:n * L])


def test_cipher_error_codes(gpu_ctx, pna):
    import torch
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(pna.PnaGpuError) as ei:
        gpu_ctx.cipher_apply_device(pna.Cipher(KEY, PHSF, pna.MODE_CTR, encryption=pna.ENC_CAMELLIA, ivs=bytes(16)), buf.data_ptr(), [0], [16])
    assert ei.value.code == -7
    with pytest.raises(pna.PnaGpuError) as ei:
        gpu_ctx.cipher_apply_device(pna.Cipher(KEY, PHSF, pna.MODE_GCM, ivs=bytes(39)), buf.data_ptr(), [0], [16])   # GCM: archive entry points only
    assert ei.value.code == -7
    with pytest.raises(pna.PnaGpuError) as ei:
        gpu_ctx.cipher_apply_device(pna.Cipher(KEY, PHSF, 3, ivs=bytes(16)), buf.data_ptr(), [0], [16])              # unknown mode
    assert ei.value.code == -7


@pytest.mark.parametrize("mode_name", ["ctr", "cbc"])
def test_create_archive_encrypted_from_host_memory(gpu_ctx, pna, pf, codec, mode_name):
    """pna_create_archive_encrypted: PBKDF2 on the host, pipelined device path (two sub-batches), random salt and IVs.  Read back the
    way the reference reads: key from PHSF + password, IV = first data chunk, decrypt, decompress."""
    mode = pna.MODE_CTR if mode_name == "ctr" else pna.MODE_CBC
    ents = [codec.corpus_file(0, 300 + i, n) for i, n in enumerate([1 << 20, 70000, 0, 3, (1 << 20) + 17, 65536, 250000, 999])] * 3
    names = [f"h/{i:03d}.txt" for i in range(len(ents))]
    with gpu_ctx.options(sub_mib=(16, 1024)):
        arc = pna.create_archive_encrypted(gpu_ctx, names, ents, b"password", mode=mode, rounds=1000)
    _, items = pf.read_archive(arc)
    assert [it.name for it in items] == names
    phsfs = {[d for ty, d in it.chunks if ty == b"PHSF"][0] for it in items}
    assert len(phsfs) == 1                                     # one key derivation per archive
    phsf = phsfs.pop().decode()
    assert phsf.startswith("$pbkdf2-sha256$i=1000,l=32$")
    key = codec.derive_key_from_phsf(phsf, b"password")
    assert len({it.data[:16] for it in items}) == len(items)   # a fresh IV per entry
    for it, e in zip(items, ents):
        assert (it.encryption, it.cipher_mode, it.raw_file_size) == (1, mode, len(e))
        assert codec.decode_payload(2, codec.decrypt_payload(1, mode, key, it.data), len(e) + 64) == e
    # the same entries with caller-supplied key / IVs give the device-path bytes
    ivs = os.urandom(16 * len(ents))
    ci = pna.Cipher(key, phsf, mode, ivs=ivs)
    a1 = pna.create_archive_encrypted(gpu_ctx, names, ents, b"", cipher=ci)
    payloads = gpu_ctx.compress_batch(ents)
    enc = (lambda iv, p: codec.aes_ctr(key, iv, p)) if mode == pna.MODE_CTR else (lambda iv, p: codec.aes_cbc_encrypt(key, iv, p))
    want = pf.write_archive_header() + b"".join(
        pf.write_encrypted_file_entry(2, 1, mode, nm, phsf, ivs[16 * i:16 * i + 16], enc(ivs[16 * i:16 * i + 16], pl), len(e))
        for i, (nm, pl, e) in enumerate(zip(names, payloads, ents))) + pf.finalize_archive()
    assert a1 == want


@pytest.mark.parametrize("algo_name", ["zstd", "deflate"])
def test_encrypted_solid_archive_in_hbm(gpu_ctx, pna, pf, codec, algo_name):
    """pna create --solid --aes ctr in HBM: SHED(enc) | PHSF | SDAT(iv) | SDAT(ciphertext)* | SEND with ONE keystream over all SDAT
    bodies.  Byte-exact against the plain solid archive with the oracle's cipher applied to the concatenated SDAT bodies."""
    import torch
    algo = pna.ALGO_ZSTD if algo_name == "zstd" else pna.ALGO_DEFLATE
    lens = [300000, 0, 5, (1 << 20) + 3, 70001, 2500000, 12]
    ents = [codec.corpus_file(i % 2, 400 + i, n) if n else b"" for i, n in enumerate(lens)]
    names = [f"s/{i}.txt" for i in range(len(lens))]
    offs, pos = [], 0
    for e in ents:
        offs.append(pos); pos = (pos + len(e) + 15) & ~15
    src = torch.zeros(pos + 8192, dtype=torch.uint8, device="cuda")
    for o, e in zip(offs, ents):
        if e:
            src[o:o + len(e)] = torch.frombuffer(bytearray(e), dtype=torch.uint8).cuda()
    iv = os.urandom(16)
    ci = pna.Cipher(KEY, PHSF, pna.MODE_CTR, ivs=iv)
    cap = pna.solid_archive_bound(algo, names, lens) + 256
    dst = torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")
    total = gpu_ctx.create_solid_archive_device(names, src.data_ptr(), offs, lens, dst.data_ptr(), cap, algo=algo, cipher=ci)
    got = dst[:total].cpu().numpy().tobytes()
    assert gpu_ctx.timing().ms_cipher > 0
    dst2 = torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")
    total2 = gpu_ctx.create_solid_archive_device(names, src.data_ptr(), offs, lens, dst2.data_ptr(), cap, algo=algo)
    (plain_solid,) = pf.read_archive(dst2[:total2].cpu().numpy().tobytes())[1]
    bodies = [d for ty, d in plain_solid.chunks if ty == b"SDAT"]
    ct = codec.aes_ctr(KEY, iv, b"".join(bodies))
    want = pf.write_archive_header() + pf.write_chunk(b"SHED", pf.solid_header_bytes(algo, 1, pf.CIPHER_MODE_CTR)) + pf.write_chunk(b"PHSF", PHSF.encode())
    want += pf.write_chunk(b"SDAT", iv)
    p = 0
    for b in bodies:
        want += pf.write_chunk(b"SDAT", ct[p:p + len(b)]); p += len(b)
    want += pf.write_chunk(b"SEND") + pf.finalize_archive()
    assert got == want
    assert bytes(dst[total:total + 16].cpu().numpy()) == b"\xA5" * 16
    # the reference's read path: decrypt the concatenated SDAT bodies (IV first), decompress, walk the inner entries
    (so,) = pf.read_archive(got)[1]
    assert (so.encryption, so.cipher_mode) == (1, 1)
    inner = pf.read_solid_inner(codec.decode_payload(algo, codec.decrypt_payload(1, 1, KEY, so.data), sum(lens) + 4096))
    assert [e.name for e in inner] == names and [e.data for e in inner] == ents
    with pytest.raises(pna.PnaGpuError) as ei:
        gpu_ctx.create_solid_archive_device(names, src.data_ptr(), offs, lens, dst.data_ptr(), cap, algo=algo, cipher=pna.Cipher(KEY, PHSF, pna.MODE_CBC, ivs=iv))
    assert ei.value.code == -7
    if algo == pna.ALGO_ZSTD:
        # and through the read-side driver: PHSF -> key, SDAT CRCs, CTR decrypt, open-size decode, inner CRCs, all but the walk on the device
        key, phsf = pna.kdf_pbkdf2_sha256(b"password", bytes(range(16)), 1000)
        ci2 = pna.Cipher(key, phsf, pna.MODE_CTR, ivs=iv)
        total = gpu_ctx.create_solid_archive_device(names, src.data_ptr(), offs, lens, dst.data_ptr(), cap, algo=algo, cipher=ci2)
        back = pna.extract_archive(gpu_ctx, dst[:total].cpu().numpy().tobytes(), b"password")
        assert [n for n, _, _ in back] == names and [d for _, _, d in back] == ents


@pytest.mark.parametrize("seg_arg,seg", [(0, 1 << 20), (65536, 65536), (1 << 18, 1 << 18)])
def test_gcm_solid_archive_in_hbm_equals_oracle_writer(gpu_ctx, pna, pf, codec, seg_arg, seg):
    """pna create --solid --aes gcm in HBM: SHED | PHSF | SDAT(stream header) | SDAT(segment ciphertext || tag)* | SEND -- ONE GCM STREAM over the
    compressed solid stream, its key bound to the SHED chunk (entry_context, lib/src/cipher/aead.rs:167-190).  Byte-exact against the plain solid
    archive's SDAT bodies run through the oracle's GcmEncryptWriter; read back the reference's way and by the extract driver."""
    import torch
    algo = pna.ALGO_ZSTD
    lens = [300000, 0, 5, (1 << 20) + 3, 70001, 2500000, 12, 3 << 20]
    ents = [codec.corpus_file(i % 2, 500 + i, n) if n else b"" for i, n in enumerate(lens)]
    ents[7] = codec.corpus_file(2, 9, lens[7])                 # incompressible: the compressed stream spans several 1 MiB GCM segments
    names = [f"s/{i}.txt" for i in range(len(lens))]
    offs, pos = [], 0
    for e in ents:
        offs.append(pos); pos = (pos + len(e) + 15) & ~15
    src = torch.zeros(pos + 8192, dtype=torch.uint8, device="cuda")
    for o, e in zip(offs, ents):
        if e:
            src[o:o + len(e)] = torch.frombuffer(bytearray(e), dtype=torch.uint8).cuda()
    k_master, phsf = pna.kdf_pbkdf2_sha256(b"password", bytes(range(16)), 1000)
    sp = os.urandom(39)
    ci = pna.Cipher(k_master, phsf, pna.MODE_GCM, ivs=sp, gcm_segment_size=seg_arg)
    cap = pna.solid_archive_enc_bound(algo, names, lens, ci)
    dst = torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")
    total = gpu_ctx.create_solid_archive_device(names, src.data_ptr(), offs, lens, dst.data_ptr(), cap, algo=algo, cipher=ci)
    got = dst[:total].cpu().numpy().tobytes()
    assert gpu_ctx.timing().ms_cipher > 0 and bytes(dst[total:total + 16].cpu().numpy()) == b"\xA5" * 16
    dst2 = torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")
    total2 = gpu_ctx.create_solid_archive_device(names, src.data_ptr(), offs, lens, dst2.data_ptr(), cap, algo=algo)
    (plain_solid,) = pf.read_archive(dst2[:total2].cpu().numpy().tobytes())[1]
    stream = b"".join(d for ty, d in plain_solid.chunks if ty == b"SDAT")
    salt, prefix = sp[:32], sp[32:]
    shed = pf.solid_header_bytes(algo, 1, 2)
    ks = codec.derive_stream_key(k_master, salt, prefix, seg, b"SHED", shed, phsf.encode())
    ct = codec.gcm_stream_encrypt(ks, prefix, seg, stream)
    want = pf.write_archive_header() + pf.write_chunk(b"SHED", shed) + pf.write_chunk(b"PHSF", phsf.encode())
    want += pf.write_chunk(b"SDAT", codec.stream_header_bytes(salt, prefix, seg, k_master))
    for p in range(0, len(ct), seg + 16):
        want += pf.write_chunk(b"SDAT", ct[p:p + seg + 16])
    want += pf.write_chunk(b"SEND") + pf.finalize_archive()
    assert len(stream) > 3 * seg and got == want
    # the reference's read path, then the extract driver
    (so,) = pf.read_archive(got)[1]
    assert (so.encryption, so.cipher_mode) == (1, 2)
    inner = pf.read_solid_inner(codec.decode_payload(algo, codec.decrypt_payload_gcm(k_master, so.data, b"SHED", shed, phsf.encode()), sum(lens) + 4096))
    assert [e.name for e in inner] == names and [e.data for e in inner] == ents
    back = pna.extract_archive(gpu_ctx, got, b"password")
    assert [n for n, _, _ in back] == names and [d for _, _, d in back] == ents


@pytest.mark.parametrize("algo_name", ["zstd", "deflate"])
def test_gcm_stream_archive_in_hbm_equals_oracle_writer(gpu_ctx, pna, pf, codec, algo_name):
    """Cipher mode 2 (GCM STREAM) in HBM: per-entry stream header + HKDF stream key on the host, CTR keystream and GHASH + tag on the
    device.  Byte-exact against the oracle (stream header, one final segment: ciphertext || tag) and read back the reference's way
    (key confirmation, stream key bound to FHED + PHSF, tag verification)."""
    import torch
    algo = pna.ALGO_ZSTD if algo_name == "zstd" else pna.ALGO_DEFLATE
    lens = [0, 1, 5, 15, 16, 17, 4095, 16373, 70001, 131073, 300000, (1 << 20) + 1, 2500000, 12, 65536, 4081 * 16]
    ents = [codec.corpus_file(i % 2, 250 + i, n) if n else b"" for i, n in enumerate(lens)]
    ents[9] = codec.corpus_file(2, 5, lens[9])                # incompressible: ciphertext longer than 256 lanes x 16 B x several rounds
    names = [f"g{i % 3}/f{i:03d}.bin" for i in range(len(lens))]
    offs, pos = [], 0
    for e in ents:
        offs.append(pos); pos = (pos + len(e) + 15) & ~15
    src = torch.zeros(pos + 8192, dtype=torch.uint8, device="cuda")
    for o, e in zip(offs, ents):
        if e:
            src[o:o + len(e)] = torch.frombuffer(bytearray(e), dtype=torch.uint8).cuda()
    k_master = hashlib.pbkdf2_hmac("sha256", b"password", b"saltsaltsalt", 1000, 32)
    sp = os.urandom(39 * len(lens))
    payloads = gpu_ctx.compress_batch(ents, algo=algo)
    # segment sizes: the default (1 MiB = the reference's DEFAULT_SEGMENT_SIZE: every payload here is one final segment), 64 KiB and 4 KiB
    # (payloads of up to several hundred segments, full last segments, tags between the segments: GcmEncryptWriter, gcm.rs:48-100)
    for seg_arg, seg in ((0, 1 << 20), (65536, 65536), (4096, 4096)):
        ci = pna.Cipher(k_master, PHSF, pna.MODE_GCM, ivs=sp, gcm_segment_size=seg_arg)
        cap = pna.archive_enc_bound(algo, names, lens, ci)
        dst = torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")
        total, eoff = gpu_ctx.create_archive_device(names, src.data_ptr(), offs, lens, dst.data_ptr(), cap, algo=algo, cipher=ci)
        got = dst[:total].cpu().numpy().tobytes()
        assert gpu_ctx.timing().ms_cipher > 0
        want = bytearray(pf.write_archive_header())
        for i, (nm, pl, e) in enumerate(zip(names, payloads, ents)):
            salt, prefix = sp[39 * i:39 * i + 32], sp[39 * i + 32:39 * i + 39]
            fhed = pf.entry_header_bytes(pf.KIND_FILE, algo, 1, 2, pf.sanitize_name(nm))
            ks = codec.derive_stream_key(k_master, salt, prefix, seg, b"FHED", fhed, PHSF.encode())
            want += pf.write_encrypted_file_entry(algo, 1, 2, pf.sanitize_name(nm), PHSF, codec.stream_header_bytes(salt, prefix, seg, k_master),
                                                  codec.gcm_stream_encrypt(ks, prefix, seg, pl), len(e))
        want += pf.finalize_archive()
        assert len(got) == len(want) and got == bytes(want), seg
        assert bytes(dst[total:total + 16].cpu().numpy()) == b"\xA5" * 16
        if seg != 1 << 20:
            assert max(len(pl) for pl in payloads) > 3 * seg                          # several segments were written
    _, items = pf.read_archive(got)                                                  # (the 4 KiB-segment archive of the loop's last round)
    for it, e in zip(items, ents):
        assert (it.encryption, it.cipher_mode, it.raw_file_size) == (1, 2, len(e))
        phsf = [d for ty, d in it.chunks if ty == b"PHSF"][0]
        comp = codec.decrypt_payload_gcm(k_master, it.data, it.chunks[0][0], it.chunks[0][1], phsf)
        assert codec.decode_payload(algo, comp, len(e) + 64) == e
    # library-drawn salts / prefixes through the host pipeline; a tampered byte is an authentication failure
    arc = pna.create_archive_encrypted(gpu_ctx, names, ents, b"password", algo=algo, mode=pna.MODE_GCM, rounds=1000)
    _, items = pf.read_archive(arc)
    assert len({it.data[:39] for it in items}) == len(items)
    phsf = [d for ty, d in items[0].chunks if ty == b"PHSF"][0]
    km = codec.derive_key_from_phsf(phsf.decode(), b"password")
    for it, e in zip(items, ents):
        assert codec.decode_payload(algo, codec.decrypt_payload_gcm(km, it.data, it.chunks[0][0], it.chunks[0][1], phsf), len(e) + 64) == e
    bad = bytearray(items[8].data); bad[100] ^= 1
    with pytest.raises(ValueError, match="authentication"):
        codec.decrypt_payload_gcm(km, bytes(bad), items[8].chunks[0][0], items[8].chunks[0][1], phsf)
    # multi-segment entries through the device read side as well (segment size 64 KiB: tags verified per segment, then decrypt + decode)
    ci = pna.Cipher(km, phsf.decode(), pna.MODE_GCM, gcm_segment_size=65536)
    cap = pna.archive_enc_bound(algo, names, lens, ci)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    total, _ = gpu_ctx.create_archive_device(names, src.data_ptr(), offs, lens, dst.data_ptr(), cap, algo=algo, cipher=ci)
    back = pna.extract_archive(gpu_ctx, dst[:total].cpu().numpy().tobytes(), b"password")
    assert [d for _, _, d in back] == ents


def test_extract_driver_decrypts_ctr_archives(gpu_ctx, pna, codec):
    """create (PBKDF2 + AES-CTR on the device) -> extract (key from the PHSF string + password, CTR decrypt and decode on the device)."""
    ents = [codec.corpus_file(0, 700 + i, n) for i, n in enumerate([1 << 20, 70000, 0, 3, (1 << 20) + 17, 65536])]
    names = [f"e/{i}.txt" for i in range(len(ents))]
    arc = pna.create_archive_encrypted(gpu_ctx, names, ents, b"password", rounds=1000)
    got = pna.extract_archive(gpu_ctx, arc, b"password")
    assert [n for n, _, _ in got] == names and [d for _, _, d in got] == ents
    with pytest.raises(pna.PnaGpuError):                        # wrong key: garbage that is not a zstd stream
        pna.extract_archive(gpu_ctx, arc, b"passw0rd")
    with pytest.raises(pna.PnaGpuError) as ei:
        pna.extract_archive(gpu_ctx, arc)
    assert ei.value.code == -2


@pytest.mark.parametrize("fixture", ["zstd_aes_ctr.pna", "zstd_aes_cbc.pna", "zstd_aes_gcm.pna"])
def test_extract_driver_opens_the_reference_encrypted_archives(gpu_ctx, pna, pf, codec, fixture):
    """The reference's encrypted golden archives through the read-side driver: Argon2id on the C++ host, CTR / CBC / GCM-STREAM on
    the device (GCM: key confirmation, segment tags), zstd entries without fSIZ sized by the decoder.  Expected: the oracle's reading."""
    arc = open(os.path.join(os.path.dirname(__file__), "golden", fixture), "rb").read()
    got = pna.extract_archive(gpu_ctx, arc, b"password")
    _, items = pf.read_archive(arc)
    assert [n for n, _, _ in got] == [it.name for it in items] and len(got) == 9
    for (n, _, d), it in zip(got, items):
        phsf = [x for ty, x in it.chunks if ty == b"PHSF"][0]
        km = codec.derive_key_from_phsf(phsf.decode(), b"password")
        comp = (codec.decrypt_payload_gcm(km, it.data, it.chunks[0][0], it.chunks[0][1], phsf) if it.cipher_mode == 2
                else codec.decrypt_payload(it.encryption, it.cipher_mode, km, it.data))
        assert d == codec.decode_payload(it.compression, comp, 8 << 20), n
    with pytest.raises(pna.PnaGpuError) as ei:
        pna.extract_archive(gpu_ctx, arc, b"passw0rd")
    assert ei.value.code == -2
    if fixture.endswith("gcm.pna"):
        assert "key confirmation" in str(ei.value)
        bad = bytearray(arc); bad[arc.index(b"FDAT", 2000) + 4 + 75 + 40] ^= 1      # inside a ciphertext: CRC first ...
        with pytest.raises(pna.PnaGpuError) as ei:
            pna.extract_archive(gpu_ctx, bytes(bad), b"password")
        assert ei.value.code == -2


@pytest.mark.parametrize("mode_name", ["cbc", "gcm"])
def test_extract_driver_round_trips_cbc_and_gcm(gpu_ctx, pna, pf, codec, mode_name):
    """create (device cipher stage) -> extract (device decrypt): CBC with its padding, GCM STREAM with tag verification; a GCM archive
    whose ciphertext is altered but whose chunk CRC is repaired fails on the tag."""
    import zlib
    mode = pna.MODE_CBC if mode_name == "cbc" else pna.MODE_GCM
    ents = [codec.corpus_file(0, 900 + i, n) for i, n in enumerate([1 << 20, 70000, 0, 3, (1 << 20) + 17, 65536, 15, 16, 17])]
    names = [f"m/{i}.txt" for i in range(len(ents))]
    arc = pna.create_archive_encrypted(gpu_ctx, names, ents, b"password", mode=mode, rounds=1000)
    got = pna.extract_archive(gpu_ctx, arc, b"password")
    assert [n for n, _, _ in got] == names and [d for _, _, d in got] == ents
    with pytest.raises(pna.PnaGpuError):
        pna.extract_archive(gpu_ctx, arc, b"passw0rd")
    # an Argon2id PHSF string on an archive of this library: the C++ host's Argon2 derives the key
    salt = bytes(range(16))
    key = codec.argon2(2, b"password", salt, 2, 64, 1)
    import base64
    phsf = "$argon2id$v=19$m=64,t=2,p=1$" + base64.b64encode(salt).decode().rstrip("=")
    a2 = pna.create_archive_encrypted(gpu_ctx, names, ents, b"", cipher=pna.Cipher(key, phsf, mode))
    assert [d for _, _, d in pna.extract_archive(gpu_ctx, a2, b"password")] == ents
    if mode == pna.MODE_GCM:
        _, items = pf.read_archive(arc)
        body = items[0].chunks[4][1]                          # FDAT(ciphertext || tag) of the first entry
        at = arc.index(body[:64])
        bad = bytearray(arc); bad[at + 100] ^= 1
        crc_at = at + len(body)
        bad[crc_at:crc_at + 4] = (zlib.crc32(bytes(bad[at:at + len(body)]), zlib.crc32(b"FDAT")) & 0xFFFFFFFF).to_bytes(4, "big")
        with pytest.raises(pna.PnaGpuError) as ei:
            pna.extract_archive(gpu_ctx, bytes(bad), b"password")
        assert ei.value.code == -2 and "authentication" in str(ei.value)


@pytest.mark.parametrize("fixture", ["solid_zstd_aes_ctr.pna", "solid_zstd_aes_cbc.pna", "solid_zstd_aes_gcm.pna"])
def test_extract_driver_opens_the_reference_encrypted_solid_archives(gpu_ctx, pna, pf, codec, fixture):
    """The reference's encrypted solid fixtures (lib/tests/extract_solid_compatibility.rs): SHED | PHSF (Argon2id) | SDAT* | SEND, one cipher
    stream over the SDAT bodies -- CTR / CBC: IV first; GCM STREAM: header, segments with tags, the stream key bound to the SHED chunk
    (entry_context, lib/src/cipher/aead.rs:167-190) --, one zstd frame, inner entries stored."""
    arc = open(os.path.join(os.path.dirname(__file__), "golden", fixture), "rb").read()
    (so,) = pf.read_archive(arc)[1]
    phsf = [d for ty, d in so.chunks if ty == b"PHSF"][0]
    km = codec.derive_key_from_phsf(phsf.decode(), b"password")
    comp = (codec.decrypt_payload_gcm(km, so.data, so.chunks[0][0], so.chunks[0][1], phsf) if so.cipher_mode == 2
            else codec.decrypt_payload(so.encryption, so.cipher_mode, km, so.data))
    inner = pf.read_solid_inner(codec.decode_payload(so.compression, comp, 16 << 20))
    got = pna.extract_archive(gpu_ctx, arc, b"password")
    assert [(n, d) for n, _, d in got] == [(e.name, e.data) for e in inner] and len(got) == 9
    with pytest.raises(pna.PnaGpuError) as ei:
        pna.extract_archive(gpu_ctx, arc, b"passw0rd")
    assert ei.value.code == -2
    if so.cipher_mode == 2:                                       # a flipped ciphertext bit (chunk CRC repaired): the segment's tag says so
        body = [d for ty, d in so.chunks if ty == b"SDAT"][-1]
        at = arc.index(body)
        bad = bytearray(arc); bad[at + len(body) // 2] ^= 1
        bad[at + len(body):at + len(body) + 4] = (zlib.crc32(bytes(bad[at:at + len(body)]), zlib.crc32(b"SDAT")) & 0xFFFFFFFF).to_bytes(4, "big")
        with pytest.raises(pna.PnaGpuError) as ei:
            pna.extract_archive(gpu_ctx, bytes(bad), b"password")
        assert ei.value.code == -2 and "authentication" in str(ei.value)


def test_cbc_stream_in_units(gpu_ctx, pna, pf, codec):
    """A CBC stream longer than the decryption's unit (16 MiB: a unit's IV is the ciphertext block in front of it) -- an entry of incompressible
    bytes, written by this library, read back by the extract driver."""
    import torch  # noqa: F401
    data = [codec.corpus_file(2, 5, (40 << 20) + 12345), codec.corpus_file(2, 6, (16 << 20) - 7), codec.corpus_file(0, 7, 100000)]
    names = ["big.bin", "unit.bin", "small.txt"]
    arc = pna.create_archive_encrypted(gpu_ctx, names, data, b"password", mode=pna.MODE_CBC, rounds=1000)
    got = pna.extract_archive(gpu_ctx, arc, b"password")
    assert [n for n, _, _ in got] == names and [d for _, _, d in got] == data


@pytest.mark.parametrize("mode_name", ["ctr", "cbc", "gcm"])
def test_encrypted_round_trip_512_mib_through_the_driver(gpu_ctx, pna, mode_name):
    """512 x 1 MiB created with the cipher stage in HBM, read back by the extract driver (device CRC, decrypt, GCM tags, decode): every
    byte equal to the generated corpus."""
    import torch
    mode = {"ctr": pna.MODE_CTR, "cbc": pna.MODE_CBC, "gcm": pna.MODE_GCM}[mode_name]
    n, L = 512, 1 << 20
    src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
    gpu_ctx.corpus_fill_device(0, 5000, n, L, L, src.data_ptr())
    names = [f"big/{i:04d}" for i in range(n)]
    key, phsf = pna.kdf_pbkdf2_sha256(b"password", bytes(range(16)), 1000)
    ci = pna.Cipher(key, phsf, mode)
    cap = pna.archive_enc_bound(pna.ALGO_ZSTD, names, [L] * n, ci)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    total, _ = gpu_ctx.create_archive_device(names, src.data_ptr(), [i * L for i in range(n)], [L] * n, dst.data_ptr(), cap, cipher=ci)
    arc = dst[:total].cpu().numpy().tobytes()
    host = src[:n * L].cpu().numpy().tobytes()
    got = pna.extract_archive(gpu_ctx, arc, b"password")
    assert [nm for nm, _, _ in got] == names
    assert all(d == host[i * L:(i + 1) * L] for i, (_, _, d) in enumerate(got))
