"""BASELINE.json's configurations at their FULL sizes on one MI355X (run with -m gpu), through size-independent properties plus
sampled comparisons with the oracle:

  configs[3]  `pna create --solid`, one 8 GiB stream (8 192 x 1 MiB inner entries): the archive is read back through the extract driver
              (every entry compared with the source), SDAT chunks on both sides of the 4 GiB offset go through an independent decoder;
  configs[4]  1 000 000 x 4 KiB, Compression::Deflate: offsets partition the output, every entry round-trips on the device, >= 1 000
              sampled entries equal the oracle's encoder model and inflate with zlib.

Reference behaviour matched: cli/src/command/create.rs:594-623 (solid create), tests/bats/large_file.bats (sizes beyond 4 GiB).
"""
import ctypes
import os
import struct
import zlib

import pytest

pytestmark = pytest.mark.gpu


def _need_hbm(torch, gib):
    free, _ = torch.cuda.mem_get_info()
    assert free >= gib * (1 << 30), f"the full-size case needs {gib} GiB of free HBM, found {free >> 30} GiB: an MI355X has 288 GB"


def _chunks(buf):
    """Walk a .pna image (numpy uint8 array / bytes): yields (type, payload offset, payload length, stored crc)."""
    pos = 8
    n = len(buf)
    while pos < n:
        ln, = struct.unpack(">I", bytes(buf[pos:pos + 4]))
        ty = bytes(buf[pos + 4:pos + 8])
        crc, = struct.unpack(">I", bytes(buf[pos + 8 + ln:pos + 12 + ln]))
        yield ty, pos + 8, ln, crc
        pos += 12 + ln


def test_solid_8gib_archive_round_trips(big_ctx, pna, pf, codec):
    gpu_ctx = big_ctx
    import numpy as np
    import torch
    n, L = 8192, 1 << 20
    _need_hbm(torch, 120)
    src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
    gpu_ctx.corpus_fill_device(0, 0, n, L, L, src.data_ptr())
    names = [f"solid/f{i:05d}.txt" for i in range(n)]
    so, sl = [i * L for i in range(n + 1)], [L] * n
    cap = pna.solid_archive_bound(pna.ALGO_ZSTD, names, sl)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    total = gpu_ctx.create_solid_archive_device(names, src.data_ptr(), so, sl, dst.data_ptr(), cap)
    arc = dst[:total].cpu().numpy()
    del dst
    host_src = src[:n * L].cpu().numpy()
    del src
    torch.cuda.empty_cache()

    # ---- structure: signature, AHED, SHED, SDAT*, SEND, AEND; every chunk CRC (the oracle's rule: crc32(type || data))
    assert bytes(arc[:8]) == bytes.fromhex("89504e410d0a1a0a")
    kinds, sdat = [], []
    for ty, off, ln, crc in _chunks(arc):
        assert zlib.crc32(arc[off:off + ln].tobytes(), zlib.crc32(ty)) == crc, (ty, off)
        if ty == b"SDAT":
            sdat.append((off, ln))
        else:
            kinds.append(ty)
    assert kinds == [b"AHED", b"SHED", b"SEND", b"AEND"]
    rec = len(pf.write_normal_entry(pf.file_entry_header(0, names[0]), [bytes(L)], L))   # bytes of one inner record (names have one length)
    inner_len = rec * n
    assert len(sdat) == (inner_len + L - 1) // L                   # one SDAT chunk = one frame = 1 MiB of the inner stream
    ratio = n * L / sum(ln for _, ln in sdat)
    assert 2.4 < ratio < 3.2

    # ---- independent decoder on both sides of the 4 GiB offset of the inner stream (u32 wrap), and at the ends
    def inner_slice(a, b):                                         # bytes [a, b) of the serialised inner stream, from the ORACLE's writer
        out = bytearray()
        for e in range(a // rec, min(n, (b - 1) // rec + 1)):
            data = codec.corpus_file(0, e, L)
            r = pf.write_normal_entry(pf.file_entry_header(0, names[e]), [data], len(data))    # a STORE entry, lib/src/entry.rs:888-913
            assert len(r) == rec
            lo, hi = max(a, e * rec), min(b, (e + 1) * rec)
            out += r[lo - e * rec:hi - e * rec]
        return bytes(out)
    dec = codec.libzstd_decompress_stream if codec.system_libzstd() is not None else codec.zstd_decompress
    for k in (0, 4095, 4096, 4097, len(sdat) - 1):
        off, ln = sdat[k]
        a, b = k * L, min((k + 1) * L, inner_len)
        assert dec(arc[off:off + ln].tobytes(), b - a) == inner_slice(a, b), k
        assert codec.zstd_decompress(arc[off:off + ln].tobytes(), b - a) == inner_slice(a, b), k

    # ---- the same archive from PAGEABLE host memory, streamed through windows (pna_gpu_create_solid_archive_host: SolidArchive::add_entry streams its entries,
    # lib/src/archive/write.rs:575-580): byte for byte the one assembled in HBM in one piece, with at most 1.5 GiB of page-locked memory for the 8 GiB
    pos = [0]
    same = [True]

    def _sink(_u, buf, k):
        piece = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_ubyte)), shape=(k,))
        same[0] = same[0] and pos[0] + k <= len(arc) and bool(np.array_equal(piece, arc[pos[0]:pos[0] + k]))
        pos[0] += k
        return 0
    scb = pna.SINK_FN(_sink)
    base = host_src.ctypes.data
    a_names = (ctypes.c_char_p * n)(*[s.encode() for s in names])
    a_src = (ctypes.c_void_p * n)(*[base + i * L for i in range(n)])
    a_len = (ctypes.c_size_t * n)(*[L] * n)
    gpu_ctx._check(gpu_ctx._L.pna_gpu_create_solid_archive_host(gpu_ctx._h, pna.ALGO_ZSTD, pna.LEVEL_DEFAULT, n, a_names, a_src, a_len, scb, None))
    assert same[0] and pos[0] == len(arc)
    gpu_ctx._L.pna_gpu_debug_pinned_bytes.restype = ctypes.c_uint64
    gpu_ctx._L.pna_gpu_debug_pinned_bytes.argtypes = [ctypes.c_void_p]
    assert gpu_ctx._L.pna_gpu_debug_pinned_bytes(gpu_ctx._h) <= (3 << 29), gpu_ctx._L.pna_gpu_debug_pinned_bytes(gpu_ctx._h)

    # ---- the whole archive through the extract driver (device CRCs, open-size decode, inner FDAT CRCs): every entry == its source
    seen = []

    def _cb(_u, idx, name, kind, data, ln):
        ok = (name.decode() == names[idx] and kind == 0 and ln == L and
              bool(np.array_equal(np.ctypeslib.as_array(ctypes.cast(data, ctypes.POINTER(ctypes.c_ubyte)), shape=(ln,)), host_src[idx * L:(idx + 1) * L])))
        seen.append(idx if ok else -1)
        return 0
    cb = pna.ENTRY_FN(_cb)
    buf = arc.tobytes()
    gpu_ctx._check(gpu_ctx._L.pna_gpu_extract_archive_host(gpu_ctx._h, buf, len(buf), None, 0, cb, None))
    assert seen == list(range(n))


def test_deflate_one_million_small_entries(big_ctx, pna, codec):
    gpu_ctx = big_ctx
    import torch
    n, L = 1_000_000, 4096
    _need_hbm(torch, 40)
    src = torch.empty(n * L + 8192, dtype=torch.uint8, device="cuda")
    gpu_ctx.corpus_fill_device(1, 0, n, L, L, src.data_ptr())
    cap = n * pna.bound(pna.ALGO_DEFLATE, L) + 64
    comp = torch.empty(cap, dtype=torch.uint8, device="cuda")
    so = [i * L for i in range(n + 1)]
    offs = gpu_ctx.compress_batch_device(src.data_ptr(), so, [L] * n, comp.data_ptr(), cap, algo=pna.ALGO_DEFLATE)
    assert len(offs) == n + 1 and offs[0] == 0 and all(offs[i] < offs[i + 1] for i in range(n)) and offs[-1] <= cap
    assert 1.6 < n * L / offs[-1] < 2.2
    # every entry, every byte: inflated on the device, compared in HBM
    back = torch.zeros(n * L + 64, dtype=torch.uint8, device="cuda")
    gpu_ctx.decompress_batch_device(comp.data_ptr(), offs[:n], [offs[i + 1] - offs[i] for i in range(n)], back.data_ptr(), so[:n], [L] * n,
                                    algo=pna.ALGO_DEFLATE)
    assert torch.equal(back[:n * L], src[:n * L])
    del back
    # sampled entries (1 001 of them, first and last included): the oracle's encoder model bit for bit, stdlib zlib reads them
    host = comp[:offs[-1]].cpu().numpy()
    for i in list(range(0, n, 1000)) + [n - 1]:
        want = codec.corpus_file(1, i, L)
        got = host[offs[i]:offs[i + 1]].tobytes()
        assert zlib.decompress(got) == want, i
        assert got == codec.deflate_model_compress(want), i


def test_entry_beyond_4gib_round_trips(big_ctx, pna, pf, codec):
    gpu_ctx = big_ctx
    """tests/bats/large_file.bats (a 5 GiB file through create + extract): ONE entry of 5 GiB (+ a small one behind it) -- 64-bit offsets
    through segment planning, the write kernels, the FDAT cut (max_chunk_size 1 GiB: FlattenWriter chunks of exactly that size, lib/src/util/io.rs:60-77)
    and the device decoder.  The archive is checked structurally (fSIZ of five bytes, FDAT chunks of exactly 1 GiB but the last, every chunk CRC), sampled frames go through an independent decoder, and every byte
    is decoded on the device and compared in HBM."""
    import numpy as np
    import torch
    n1, L = 5 * 1024, 1 << 20
    big = n1 * L
    _need_hbm(torch, 80)
    src = torch.empty(big + 4096 + 8192, dtype=torch.uint8, device="cuda")
    gpu_ctx.corpus_fill_device(0, 7000, n1, L, L, src.data_ptr())                # the big entry = 5 120 corpus files back to back
    gpu_ctx.corpus_fill_device(1, 1, 1, 3000, 3000, src.data_ptr() + big)
    so, sl = [0, big, big + 3000], [big, 3000]
    names = ["large/five_gib.bin", "large/small.txt"]
    cap = pna.archive_chunked_bound(pna.ALGO_ZSTD, names, sl, 1 << 30)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    total, eoff = gpu_ctx.create_archive_chunked_device(names, src.data_ptr(), so, sl, dst.data_ptr(), cap, 1 << 30)
    assert eoff[0] == 28 and eoff[2] == total - 12 and total > (1 << 30)
    # ---- structure on the host: chunk walk with CRCs, fSIZ, FDAT sizes
    arc = dst[:total].cpu().numpy()
    kinds, fdat = [], []
    for ty, off, ln, crc in _chunks(arc):
        assert zlib.crc32(arc[off:off + ln].tobytes(), zlib.crc32(ty)) == crc, (ty, off)
        kinds.append(ty)
        if ty == b"FDAT":
            fdat.append((off, ln))
        if ty == b"fSIZ" and len(kinds) == 3:
            assert int.from_bytes(arc[off:off + ln].tobytes(), "big") == big and ln == 5
    assert kinds[:3] == [b"AHED", b"FHED", b"fSIZ"] and kinds[-1] == b"AEND" and kinds.count(b"FHED") == 2
    big_fdat = fdat[:-1]
    assert len(big_fdat) >= 2 and all(ln == (1 << 30) for _, ln in big_fdat[:-1]) and 0 < big_fdat[-1][1] <= (1 << 30) and 2.4 < big / sum(ln for _, ln in big_fdat) < 3.2
    # ---- an independent decoder on frames around the 4 GiB mark of the entry (frame k = input bytes [k MiB, (k + 1) MiB))
    stream = np.concatenate([arc[o:o + ln] for o, ln in big_fdat])
    pos, k, want_frames = 0, 0, {0, 4095, 4096, 4097, n1 - 1}
    dec = codec.libzstd_decompress_stream if codec.system_libzstd() is not None else codec.zstd_decompress
    while pos < len(stream):                                                     # frame walk: header 6 bytes, blocks until the last-block flag
        assert bytes(stream[pos:pos + 4]) == bytes.fromhex("28b52ffd")
        q = pos + 6
        while True:
            h = int(stream[q]) | (int(stream[q + 1]) << 8) | (int(stream[q + 2]) << 16)
            q += 3 + (1 if (h >> 1) & 3 == 1 else h >> 3)
            if h & 1:
                break
        if k in want_frames:
            assert dec(stream[pos:q].tobytes(), L) == codec.corpus_file(0, 7000 + k, L), k
        pos, k = q, k + 1
    assert k == n1
    del stream
    # ---- every byte: payload offsets from the chunk walk -> device decoder -> compare in HBM
    back = torch.zeros(big + 4096 + 64, dtype=torch.uint8, device="cuda")
    # (the device decoder takes one contiguous payload per entry: gather the big entry's FDAT bodies on the device first)
    packed = torch.cat([dst[o:o + ln] for o, ln in big_fdat])
    gpu_ctx.decompress_batch_device(packed.data_ptr(), [0], [packed.numel()], back.data_ptr(), [0], [big])
    assert torch.equal(back[:big], src[:big])
    o2, l2 = fdat[-1]
    gpu_ctx.decompress_batch_device(dst.data_ptr(), [o2], [l2], back.data_ptr(), [big], [3000])
    assert torch.equal(back[big:big + 3000], src[big:big + 3000])
    # ---- and through the extract driver (`pna extract` of the 5 GiB entry: tests/bats/large_file.bats)
    del back, packed
    host_src = src.cpu().numpy()
    seen = []

    def _cb(_u, idx, name, kind, data, ln):
        lo = so[idx]
        got = np.ctypeslib.as_array(ctypes.cast(data, ctypes.POINTER(ctypes.c_ubyte)), shape=(ln,))
        seen.append((name.decode(), ln, bool(np.array_equal(got, host_src[lo:lo + ln]))))
        return 0
    cb = pna.ENTRY_FN(_cb)
    buf = arc.tobytes()
    gpu_ctx._check(gpu_ctx._L.pna_gpu_extract_archive_host(gpu_ctx._h, buf, len(buf), None, 0, cb, None))
    assert seen == [(names[0], big, True), (names[1], 3000, True)]


@pytest.mark.gpu
def test_deflate_entry_beyond_4gib_round_trips(big_ctx, pna, pf, codec):
    """The same file as Compression::Deflate (tests/bats/large_file.bats; flate2::read::ZlibDecoder, lib/src/entry/read.rs:178-179): ONE zlib
    stream that decodes to more than 4 GiB.  The device decoder takes it by its sync-flush delimited pieces (64-bit stream positions and sizes,
    the execution pass's 32-bit positions re-based as it goes); Adler-32 over all of it on the device; zlib (the C library) decodes the same
    stream on the host piece by piece as the independent reader."""
    gpu_ctx = big_ctx
    import numpy as np
    import torch
    n1, L = 4 * 1024 + 160, 1 << 20
    big = n1 * L
    assert big > 1 << 32
    _need_hbm(torch, 80)
    src = torch.empty(big + 4096 + 8192, dtype=torch.uint8, device="cuda")
    gpu_ctx.corpus_fill_device(0, 9000, n1, L, L, src.data_ptr())
    gpu_ctx.corpus_fill_device(1, 1, 1, 3000, 3000, src.data_ptr() + big)
    so, sl = [0, big, big + 3000], [big, 3000]
    names = ["large/four_gib_and_more.bin", "large/small.txt"]
    cap = pna.archive_chunked_bound(pna.ALGO_DEFLATE, names, sl, 1 << 30)
    dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
    total, eoff = gpu_ctx.create_archive_chunked_device(names, src.data_ptr(), so, sl, dst.data_ptr(), cap, 1 << 30, algo=pna.ALGO_DEFLATE)
    arc = dst[:total].cpu().numpy()
    kinds, fdat = [], []
    for ty, off, ln, crc in _chunks(arc):
        assert zlib.crc32(arc[off:off + ln].tobytes(), zlib.crc32(ty)) == crc, (ty, off)
        kinds.append(ty)
        if ty == b"FDAT":
            fdat.append((off, ln))
        if ty == b"fSIZ" and len(kinds) == 3:
            assert int.from_bytes(arc[off:off + ln].tobytes(), "big") == big and ln == 5
    assert kinds[:3] == [b"AHED", b"FHED", b"fSIZ"] and kinds[-1] == b"AEND" and kinds.count(b"FHED") == 2
    big_fdat = fdat[:-1]
    assert len(big_fdat) >= 2 and all(ln == (1 << 30) for _, ln in big_fdat[:-1]) and 2.2 < big / sum(ln for _, ln in big_fdat) < 3.0
    # ---- the independent reader: zlib over the whole stream, compared as it comes
    host_src = src.cpu().numpy()
    d, pos = zlib.decompressobj(), 0
    for o, ln in big_fdat:
        for a in range(o, o + ln, 64 << 20):
            out = d.decompress(arc[a:min(a + (64 << 20), o + ln)].tobytes())
            assert np.array_equal(np.frombuffer(out, dtype=np.uint8), host_src[pos:pos + len(out)]), pos
            pos += len(out)
    out = d.flush()
    assert np.array_equal(np.frombuffer(out, dtype=np.uint8), host_src[pos:pos + len(out)]) and pos + len(out) == big and d.eof
    # ---- the device decoder: every byte, compared in HBM
    back = torch.zeros(big + 4096 + 64, dtype=torch.uint8, device="cuda")
    packed = torch.cat([dst[o:o + ln] for o, ln in big_fdat])
    gpu_ctx.decompress_batch_device(packed.data_ptr(), [0], [packed.numel()], back.data_ptr(), [0], [big], algo=pna.ALGO_DEFLATE)
    assert torch.equal(back[:big], src[:big])
    # a damaged byte far behind the 4 GiB mark is found (Adler-32 / the piece's own checks)
    bad = packed.clone()
    bad[packed.numel() - 5000] ^= 0x10
    with pytest.raises(pna.PnaGpuError):
        gpu_ctx.decompress_batch_device(bad.data_ptr(), [0], [bad.numel()], back.data_ptr(), [0], [big], algo=pna.ALGO_DEFLATE)
    del back, packed, bad
    # ---- and through the extract driver
    seen = []

    def _cb(_u, idx, name, kind, data, ln):
        lo = so[idx]
        got = np.ctypeslib.as_array(ctypes.cast(data, ctypes.POINTER(ctypes.c_ubyte)), shape=(ln,))
        seen.append((name.decode(), ln, bool(np.array_equal(got, host_src[lo:lo + ln]))))
        return 0
    cb = pna.ENTRY_FN(_cb)
    buf = arc.tobytes()
    gpu_ctx._check(gpu_ctx._L.pna_gpu_extract_archive_host(gpu_ctx._h, buf, len(buf), None, 0, cb, None))
    assert seen == [(names[0], big, True), (names[1], 3000, True)]


@pytest.mark.gpu
def test_one_foreign_zstd_frame_beyond_4gib(big_ctx, pna, pf, codec):
    """What the reference writes for a file of more than 4 GiB (tests/bats/large_file.bats; zstd::stream::write::Encoder, lib/src/compress.rs:32-41):
    ONE zstd frame, however large the entry.  The writer here is the system libzstd (one ZSTD_compress call: a single frame with an 8-byte
    Frame_Content_Size).  The device decoder parses its 33 000 blocks side by side and executes them by pointer jumping in WINDOWS of 1 GiB (k_zexec_par.hip: a word
    counts 31 bits from its window's start, sources in front of the window are bytes of the output) -- round 4; until then one workgroup walked such a frame at
    ~11 MiB/s, which the option zdec_fallback_max_mib refuses here so that a regression fails at once --; every byte is compared in HBM; a flipped bit behind the
    4 GiB mark of the content is found."""
    if codec.system_libzstd() is None:
        pytest.skip("system libzstd (the writer of the test frame) is absent")
    gpu_ctx = big_ctx
    import numpy as np
    import torch
    n1, L = 4 * 1024 + 48, 1 << 20
    big = n1 * L
    assert big > 1 << 32
    _need_hbm(torch, 40)
    src = torch.empty(big + 4096, dtype=torch.uint8, device="cuda")
    gpu_ctx.corpus_fill_device(0, 9500, n1, L, L, src.data_ptr())
    Z = codec.system_libzstd()
    host = src[:big].cpu().numpy()
    cap = Z.ZSTD_compressBound(big)
    buf = np.empty(cap, dtype=np.uint8)
    n = Z.ZSTD_compress(buf.ctypes.data, cap, ctypes.c_char_p(host.ctypes.data), big, 1)
    assert not Z.ZSTD_isError(n) and n < big // 2
    fhd = int(buf[4])
    assert (fhd >> 6) == 3                                                       # Frame_Content_Size of 8 bytes (behind the window descriptor, if any)
    fcs_at = 5 + (0 if (fhd >> 5) & 1 else 1)
    assert int.from_bytes(buf[fcs_at:fcs_at + 8].tobytes(), "little") == big
    del host
    comp = torch.from_numpy(buf[:n]).cuda()
    back = torch.zeros(big + 64, dtype=torch.uint8, device="cuda")
    gpu_ctx.set_option("zdec_fallback_max_mib", 64)
    gpu_ctx.decompress_batch_device(comp.data_ptr(), [0], [n], back.data_ptr(), [0], [big])
    assert torch.equal(back[:big], src[:big])
    bad = comp.clone()
    bad[n - 70000] ^= 0x04
    back.zero_()
    try:
        gpu_ctx.decompress_batch_device(bad.data_ptr(), [0], [n], back.data_ptr(), [0], [big])
        same = torch.equal(back[:big], src[:big])                                # (a flipped literal bit may still decode: then the content differs)
    except pna.PnaGpuError:
        same = False
    assert not same


@pytest.mark.gpu
def test_one_foreign_zlib_stream_beyond_4gib(big_ctx, pna, pf, codec):
    """The same file as Compression::Deflate from the reference's writer (flate2's ZlibEncoder: ONE zlib stream without sync flushes, lib/src/compress.rs:32-41), here from
    the stdlib's zlib at level 1: more than 4 GiB of content.  The chunk decoder (block starts by trial, a wave per chunk, 64-bit output positions) hands its records to the
    pointer-jumping executor, which runs them in windows of 1 GiB; until round 4's second half such a stream was PNA_E_UNSUPPORTED (and 2 - 4 GiB took the wave-per-stream
    walk: 107 s per GiB).  Every byte is compared in HBM, the Adler-32 trailer is checked on the device, damage is refused or decodes differently."""
    gpu_ctx = big_ctx
    import numpy as np
    import torch
    n1, L = 4 * 1024 + 24, 1 << 20
    big = n1 * L
    assert big > 1 << 32
    _need_hbm(torch, 60)
    src = torch.empty(big + 4096, dtype=torch.uint8, device="cuda")
    gpu_ctx.corpus_fill_device(0, 9900, n1, L, L, src.data_ptr())
    host = src[:big].cpu().numpy()
    co, parts = zlib.compressobj(1), []
    for a in range(0, big, 256 << 20):
        parts.append(co.compress(host[a:a + (256 << 20)].tobytes()))
    parts.append(co.flush())
    del host
    buf = np.frombuffer(b"".join(parts), dtype=np.uint8)
    del parts
    n = buf.size
    assert n < big // 2
    comp = torch.from_numpy(buf.copy()).cuda()
    back = torch.zeros(big + 64, dtype=torch.uint8, device="cuda")
    gpu_ctx.decompress_batch_device(comp.data_ptr(), [0], [n], back.data_ptr(), [0], [big], algo=pna.ALGO_DEFLATE)
    assert gpu_ctx.timing().lz_match_launches == 1                                # (decode calls: the streams that went through the chunk decoder)
    assert torch.equal(back[:big], src[:big])
    bad = comp.clone()
    bad[n - 90000] ^= 0x08
    back.zero_()
    try:
        gpu_ctx.decompress_batch_device(bad.data_ptr(), [0], [n], back.data_ptr(), [0], [big], algo=pna.ALGO_DEFLATE)
        same = torch.equal(back[:big], src[:big])
    except pna.PnaGpuError:
        same = False
    assert not same
