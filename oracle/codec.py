"""oracle/codec.py -- ctypes access to the plain-C oracle (liboracle.so).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess
import zlib

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    """Compile oracle/*.c into oracle/liboracle.so (gcc, no dependencies)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "liboracle.so")


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.pna_oracle_zstd_decompress.restype = ctypes.c_long
        L.pna_oracle_zstd_decompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p,
                                                 ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
        _LIB = L
    return _LIB


def zstd_decompress(data: bytes, max_out: int) -> bytes:
    """RFC 8878 decode of all concatenated frames (oracle/zstd_dec.c)."""
    out = ctypes.create_string_buffer(max_out + 64)
    frames = ctypes.c_int()
    r = lib().pna_oracle_zstd_decompress(bytes(data), len(data), out, max_out + 64, ctypes.byref(frames))
    if r < 0:
        raise ValueError(f"zstd oracle decode error {r}")
    return out.raw[:r]


def zstd_frame_count(data: bytes, max_out: int) -> int:
    out = ctypes.create_string_buffer(max_out + 64)
    frames = ctypes.c_int()
    r = lib().pna_oracle_zstd_decompress(bytes(data), len(data), out, max_out + 64, ctypes.byref(frames))
    if r < 0:
        raise ValueError(f"zstd oracle decode error {r}")
    return frames.value


def zlib_decompress(data: bytes) -> bytes:
    """RFC 1950 decode through the stdlib (independent of the product)."""
    d = zlib.decompressobj()
    out = d.decompress(data)
    if not d.eof or d.unused_data:
        raise ValueError("zlib stream not terminated exactly")
    return out


def decode_payload(compression: int, data: bytes, max_out: int) -> bytes:
    """decompress_reader dispatch -- lib/src/entry/read.rs:171-190."""
    if compression == 0:
        return bytes(data)
    if compression == 1:
        return zlib_decompress(data)
    if compression == 2:
        return zstd_decompress(data, max_out)
    raise ValueError(f"unsupported compression {compression}")


_SYS_ZSTD = None


def system_libzstd():
    """System libzstd (independent decoder / CPU baseline codec), or None when absent."""
    global _SYS_ZSTD
    if _SYS_ZSTD is None:
        for name in ("libzstd.so.1", "/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/opt/conda/lib/libzstd.so.1"):
            try:
                Z = ctypes.CDLL(name)
            except OSError:
                continue
            Z.ZSTD_compressBound.restype = ctypes.c_size_t
            Z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
            Z.ZSTD_compress.restype = ctypes.c_size_t
            Z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
            Z.ZSTD_decompress.restype = ctypes.c_size_t
            Z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
            Z.ZSTD_isError.restype = ctypes.c_uint
            Z.ZSTD_isError.argtypes = [ctypes.c_size_t]
            Z.ZSTD_versionNumber.restype = ctypes.c_uint
            Z.ZSTD_createDStream.restype = ctypes.c_void_p
            Z.ZSTD_freeDStream.argtypes = [ctypes.c_void_p]
            Z.ZSTD_decompressStream.restype = ctypes.c_size_t
            Z.ZSTD_decompressStream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            _SYS_ZSTD = Z
            break
        else:
            _SYS_ZSTD = False
    return _SYS_ZSTD or None


class _Buf(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t)]


def libzstd_decompress_stream(data: bytes, max_out: int) -> bytes:
    """Multi-frame streaming decode with the system libzstd (what zstd-rs Decoder does)."""
    Z = system_libzstd()
    if Z is None:
        raise RuntimeError("system libzstd not available")
    ds = Z.ZSTD_createDStream()
    src = ctypes.create_string_buffer(bytes(data), len(data))
    dst = ctypes.create_string_buffer(max_out + 64)
    ib = _Buf(ctypes.cast(src, ctypes.c_void_p), len(data), 0)
    ob = _Buf(ctypes.cast(dst, ctypes.c_void_p), max_out + 64, 0)
    try:
        while ib.pos < ib.size:
            r = Z.ZSTD_decompressStream(ds, ctypes.byref(ob), ctypes.byref(ib))
            if Z.ZSTD_isError(r):
                raise ValueError("libzstd decode error")
            if r == 0 and ib.pos >= ib.size:
                break
            if ob.pos >= ob.size:
                raise ValueError("libzstd output overflow")
        if ib.pos >= ib.size and len(data) and r != 0:
            raise ValueError("libzstd: truncated frame")
    finally:
        Z.ZSTD_freeDStream(ds)
    return dst.raw[:ob.pos]


def libzstd_compress(data: bytes, level: int = 3) -> bytes:
    Z = system_libzstd()
    cap = Z.ZSTD_compressBound(len(data))
    buf = ctypes.create_string_buffer(cap)
    n = Z.ZSTD_compress(buf, cap, bytes(data), len(data), level)
    if Z.ZSTD_isError(n):
        raise ValueError("libzstd compress error")
    return buf.raw[:n]


def libzstd_compress_checksum(data: bytes, level: int = 3, extra=()) -> bytes:
    """One frame WITH Content_Checksum (XXH64 low 32 bits behind the last block) from the system libzstd: test input for the device decoder's check.
    `extra`: further (ZSTD_cParameter, value) pairs, e.g. (101, 26) = ZSTD_c_windowLog, (160, 1) = ZSTD_c_enableLongDistanceMatching."""
    Z = system_libzstd()
    Z.ZSTD_createCCtx.restype = ctypes.c_void_p
    Z.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
    Z.ZSTD_CCtx_setParameter.restype = ctypes.c_size_t
    Z.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    Z.ZSTD_compress2.restype = ctypes.c_size_t
    Z.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    cctx = Z.ZSTD_createCCtx()
    try:
        for prm, v in ((100, level), (201, 1)) + tuple(extra):  # ZSTD_c_compressionLevel, ZSTD_c_checksumFlag
            if Z.ZSTD_isError(Z.ZSTD_CCtx_setParameter(cctx, prm, v)):
                raise ValueError("libzstd: parameter refused")
        cap = Z.ZSTD_compressBound(len(data))
        buf = ctypes.create_string_buffer(cap)
        n = Z.ZSTD_compress2(cctx, buf, cap, bytes(data), len(data))
        if Z.ZSTD_isError(n):
            raise ValueError("libzstd compress error")
        return buf.raw[:n]
    finally:
        Z.ZSTD_freeCCtx(cctx)


class ZstdParams(ctypes.Structure):
    """Mirror of pna_zstd_params (oracle/zstd_model.h)."""
    _fields_ = [("hash_log", ctypes.c_uint32), ("min_match", ctypes.c_uint32), ("tile", ctypes.c_uint32),
                ("max_off", ctypes.c_uint32), ("cap1", ctypes.c_uint32), ("lookahead", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("max_len", ctypes.c_uint32), ("region", ctypes.c_uint32),
                ("ins_mod", ctypes.c_uint32), ("back_cap", ctypes.c_uint32), ("rounds", ctypes.c_uint32),
                ("near_off", ctypes.c_uint32), ("cap_far", ctypes.c_uint32), ("blk_log", ctypes.c_uint32), ("len_word_max", ctypes.c_uint32),
                ("tab3", ctypes.c_uint32), ("mtile", ctypes.c_uint32), ("small_seg", ctypes.c_uint32), ("small_slots", ctypes.c_uint32),
                ("small_tile", ctypes.c_uint32), ("mid_seg", ctypes.c_uint32), ("mid_slots", ctypes.c_uint32), ("cut_min", ctypes.c_uint32), ("fixup", ctypes.c_uint32), ("far_slots", ctypes.c_uint32), ("far_from", ctypes.c_uint32)]


F_HUF, F_FSE, F_LAZY = 1, 2, 4
F_FAR, F_ADOPT, F_INS2, F_STRONG = 0x10, 0x20, 0x40, 0x80       # the product's level-set bits (include/pna_gpu.h); the model takes explicit parameters


def default_params() -> ZstdParams:
    p = ZstdParams()
    lib().pna_zstd_default_params(ctypes.byref(p))
    return p


def product_level_flags(level, deflate: bool = False, ctx_flags: int | None = None):
    """(flags, gtab): the level sets behind the reference's level scale as the product maps them (pna_host.cpp level_flags / set_call_level):
    zstd < 0, 1 fast; 2 light; 0, 3 default (+ F_STRONG, round 4); 4..9 high (the 16 KiB window: level_win32k; a fourth adoption round); 10..22 max (+ the hash table in global memory) -- deflate 1..3, 4..8, 9 (deflate 0: stored blocks only, params_for_level).  ctx_flags: the context's own flag bits where a test creates it with explicit ones (default: the library's choice)."""
    base = ctx_flags if ctx_flags is not None else ((F_ADOPT | F_INS2 | F_LAZY) if deflate else (F_HUF | F_FSE | F_LAZY | F_FAR | F_ADOPT | F_INS2))
    if deflate:
        lv = 6 if level is None or level == -1000 else (9 if level < 0 or level > 9 else level)      # (PNA_LEVEL_DEFAULT; a negative Custom(n) wraps and clamps to 9)
        fast, balanced, strong = lv <= 3, False, lv >= 9                    # (deflate 4..5 = the default set)
    else:
        lv = 3 if level is None or level == -1000 else min(level, 22)
        fast, balanced, strong = lv < 0 or lv == 1, False, (lv >= 3 or lv == 0)          # (zstd 2 = the light set: two adoption rounds; 0 = the default = 3)
    if fast:
        fl = base & ~(F_FAR | F_ADOPT | F_INS2 | F_STRONG)                    # (lazy deferral stays)
    elif balanced:
        fl = base & ~(F_LAZY | F_STRONG)
    elif strong and base & F_ADOPT and base & F_LAZY:
        fl = base | F_STRONG
    else:
        fl = base
    return fl, bool(not deflate and lv >= 10 and fl & F_STRONG and fl & F_ADOPT)


HIGH_FROM = 4          # the first zstd level of the high set (pna_host.cpp)


def level_win32k(level, deflate: bool = False, win32k: int = 1) -> int:
    """The window geometry a zstd level runs with the context's option win32k (1 by default): levels 4..9 take the 16 KiB window (2), the others the option's."""
    lv = 3 if level is None or level == -1000 else min(level, 22)
    return 2 if (not deflate and win32k and lv >= HIGH_FROM) else win32k


def params_for_level(level, deflate: bool = False, blk_log: int = 0, ctx_flags: int | None = None, win32k: int = 1, tab3: int = 1, far1: int = 1, strong2: int = 1) -> "ZstdParams":
    fl, gtab = product_level_flags(level, deflate, ctx_flags)
    high = bool(strong2) and not deflate and level is not None and level != -1000 and min(level, 22) >= HIGH_FROM        # zstd 4 .. 22 (the product's option strong2, default on)
    p = params_for_flags(fl, deflate=deflate, blk_log=blk_log, gtab=gtab, win32k=level_win32k(level, deflate, win32k), tab3=tab3, far1=far1, strong2=high)
    if deflate and level == 0:
        p.flags |= 0x200               # PNA_F_STORED: deflate level 0 = Compression::none(), stored blocks only (lib/src/compress/deflate.rs:89-101)
    return p


def params_for_flags(flags: int, deflate: bool = False, blk_log: int = 0, gtab: bool = False, win32k: int = 1, lazy2: int = 2, tab3: int = 1, far1: int = 1, strong2: bool = False) -> ZstdParams:
    """The model parameters that correspond to the product's flag bits: without F_FAR the look-back ends with the LDS window, without
    F_ADOPT there is no backward adoption, without F_INS2 every position enters the table.  blk_log: the block size the device chose
    (pna_gpu_timing.blk_log: 13..16 in its latency mode for small batches, else 17 = 128 KiB).  gtab: the match kernel's table lies in global
    memory (zstd levels 10..22).  The zstd sets with far candidates and lazy deferral run the 32 KiB-window geometry of the match finder (32 704
    table slots) unless the table is global or the context's option win32k is 0; everything else the 64 KiB one (24 512)."""
    p = deflate_default_params() if deflate else default_params()
    p.blk_log = blk_log
    # the packed table (three 21-bit entries per 64-bit LDS word; the product's option tab3, default on): the sets on the 32 / 16 KiB geometries with
    # even-position inserts and backward adoption -- 49 062 / 55 206 slots instead of 32 704 / 36 800
    packed = bool(tab3 and flags & F_INS2 and flags & F_ADOPT)
    p.tab3 = 0
    p.far_slots = 0
    if not deflate and not (flags & F_FAR and flags & F_LAZY and not gtab and win32k):
        p.hash_log, p.near_off = 24512, 56064
    elif not deflate and win32k >= 2:
        p.hash_log, p.near_off, p.tab3 = (55206 if packed else 36800), 6912, int(packed)       # the 16 KiB window: the high set (and, with the product's option win32k = 2, the default set)
    elif not deflate:
        p.hash_log, p.near_off, p.tab3 = (49062 if packed else 32704), 23296, int(packed)
        if packed and far1:            # FLAG_FAR1 (the product's option far1, default on): the light / default sets verify at most 63 far candidates (offset >= 28 368: beyond k_lzm's window) per wave of 256 positions
            p.far_slots, p.far_from = 63, 28368
    p.flags = (p.flags & ~(F_HUF | F_FSE | F_LAZY)) | (flags & (F_HUF | F_FSE | F_LAZY)) if not deflate else ((p.flags & ~F_LAZY) | (flags & F_LAZY))
    p.flags &= ~0x180                      # two- and three-step lazy deferral go with F_LAZY (the product's option lazy2 = 2 / 1 / 0: both, the first, none -- the high sets keep the first)
    if flags & F_LAZY and (lazy2 or flags & F_STRONG):
        p.flags |= 0x80
        if lazy2 >= 2:
            p.flags |= 0x100
    if not deflate:
        p.flags &= ~8                     # no repeat codes on the device
        if not flags & F_FAR:
            p.max_off = p.near_off
    if flags & F_STRONG and flags & F_ADOPT:   # third adoption round (4 lanes, 7 back bytes) + two-step lazy deferral
        p.rounds = 0x214                       # round 5: the round over four positions FIRST, then 1, then 2 (+ 0.03 % of ratio on the corpus for nothing)
        p.back_cap = 7
        p.flags |= 0x80
    if strong2 and flags & F_STRONG and flags & F_ADOPT and not deflate and (gtab or (p.tab3 and p.hash_log == 55206)):
        # the high and max sets (zstd 4 .. 22 on their standard geometries; round 5): a FOURTH adoption round over eight positions, first, and up to 15 back bytes -- levels 6 .. 9 2.864 -> 2.880
        p.rounds = 0x2148
        p.back_cap = 15
    if gtab:
        p.hash_log = 19                # zstd levels 10 .. 22: the match kernel's table lies in global memory, 2^19 slots per segment (index = the hash's top bits)
        p.max_off = 1 << 20            # ... and its words keep 4 bytes per position: the whole segment as look-back, no clamp of the adopted lengths
        p.len_word_max = 0
    if not flags & F_ADOPT:
        p.rounds = 0
        p.back_cap = 0
    if not flags & F_INS2:
        p.ins_mod = 1
    return p


def model_compress(data: bytes, params: ZstdParams | None = None) -> bytes:
    """The deterministic zstd-format encoder model (oracle/zstd_model.c)."""
    L = lib()
    L.pna_zstd_bound.restype = ctypes.c_size_t
    L.pna_zstd_bound.argtypes = [ctypes.c_size_t]
    L.pna_zstd_model_compress.restype = ctypes.c_size_t
    L.pna_zstd_model_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                          ctypes.POINTER(ZstdParams)]
    if params is None:
        params = default_params()
    cap = L.pna_zstd_bound(len(data))
    buf = ctypes.create_string_buffer(cap)
    n = L.pna_zstd_model_compress(bytes(data), len(data), buf, cap, ctypes.byref(params))
    if n == 0:
        raise ValueError("model compress failed")
    return buf.raw[:n]


_CORPUS = None


def corpus_tables():
    global _CORPUS
    if _CORPUS is None:
        vocab = ctypes.create_string_buffer(50000 * 16)
        cum = (ctypes.c_uint64 * 50000)()
        ph = (ctypes.c_uint32 * (8192 * 4))()
        lib().pna_corpus_tables(vocab, cum, ph)
        _CORPUS = (vocab, cum, ph)
    return _CORPUS


def corpus_file(kind: int, idx: int, n: int) -> bytes:
    """Synthetic corpus file (oracle/corpus_model.c): kind 0 enwik-style, 1 random-text, 2 random bytes, 3 zeros, 4 'x'."""
    vocab, cum, ph = corpus_tables()
    out = ctypes.create_string_buffer(max(n, 1))
    lib().pna_corpus_file(ctypes.c_int(kind), ctypes.c_uint64(idx), out, ctypes.c_size_t(n), vocab, cum, ph)
    return out.raw[:n]


class _Seq(ctypes.Structure):
    _fields_ = [("ll", ctypes.c_uint32), ("ml", ctypes.c_uint32), ("off", ctypes.c_uint32)]


def model_lz_segment(seg: bytes, params: ZstdParams | None = None):
    """LZ stage of the model for one segment (<= 1 MiB): list over blocks of ([(ll, ml, off)...], literal bytes)."""
    L = lib()
    if params is None:
        params = default_params()
    assert len(seg) <= (1 << 20)
    L.pna_lz_block.restype = ctypes.c_uint32
    L.pna_lz_block.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                               ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ZstdParams),
                               ctypes.POINTER(_Seq), ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
    table = (ctypes.c_uint32 * ((1 << params.hash_log) if params.hash_log <= 31 else params.hash_log))()
    out = []
    BLK = (1 << params.blk_log) if 13 <= params.blk_log < 17 else (1 << 17)
    seqs = (_Seq * (BLK // 4))()
    lits = ctypes.create_string_buffer(BLK + 8)
    for b0 in range(0, len(seg), BLK):
        bl = min(BLK, len(seg) - b0)
        nlit = ctypes.c_uint32()
        ns = L.pna_lz_block(bytes(seg), len(seg), b0, bl, table, ctypes.byref(params), seqs, lits, ctypes.byref(nlit))
        out.append(([(seqs[i].ll, seqs[i].ml, seqs[i].off) for i in range(ns)], lits.raw[:nlit.value]))
    return out


def deflate_default_params() -> ZstdParams:
    p = ZstdParams()
    lib().pna_deflate_default_params(ctypes.byref(p))
    return p


def deflate_model_compress(data: bytes, params: ZstdParams | None = None) -> bytes:
    """The deterministic zlib/deflate encoder model (oracle/deflate_model.c)."""
    L = lib()
    L.pna_deflate_bound.restype = ctypes.c_size_t
    L.pna_deflate_bound.argtypes = [ctypes.c_size_t]
    L.pna_deflate_model_compress.restype = ctypes.c_size_t
    L.pna_deflate_model_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                             ctypes.POINTER(ZstdParams)]
    if params is None:
        params = deflate_default_params()
    cap = L.pna_deflate_bound(len(data))
    buf = ctypes.create_string_buffer(cap)
    n = L.pna_deflate_model_compress(bytes(data), len(data), buf, cap, ctypes.byref(params))
    if n == 0:
        raise ValueError("deflate model compress failed")
    return buf.raw[:n]


# ----------------------------------------------------------------------------- cipher layer (oracle/cipher_model.c)

def aes_block(key: bytes, block: bytes, decrypt: bool = False) -> bytes:
    out = ctypes.create_string_buffer(16)
    lib().pna_oracle_aes_block(bytes(key), len(key), bytes(block), out, int(decrypt))
    return out.raw


def aes_ctr(key: bytes, iv: bytes, data: bytes, pos: int = 0, flavor: int = 0) -> bytes:
    """CTR keystream XOR (encrypt == decrypt); flavor 0 = Ctr128BE (the reference's writer), 1 = Ctr64LE (its unit test)."""
    buf = ctypes.create_string_buffer(bytes(data), len(data))
    lib().pna_oracle_aes_ctr(bytes(key), len(key), bytes(iv), flavor, ctypes.c_uint64(pos), buf, ctypes.c_size_t(len(data)))
    return buf.raw[:len(data)]


def aes_cbc_encrypt(key: bytes, iv: bytes, data: bytes) -> bytes:
    L = lib()
    L.pna_oracle_aes_cbc_encrypt.restype = ctypes.c_size_t
    out = ctypes.create_string_buffer((len(data) // 16 + 1) * 16)
    n = L.pna_oracle_aes_cbc_encrypt(bytes(key), len(key), bytes(iv), bytes(data), ctypes.c_size_t(len(data)), out)
    return out.raw[:n]


def aes_cbc_decrypt(key: bytes, iv: bytes, data: bytes) -> bytes:
    L = lib()
    L.pna_oracle_aes_cbc_decrypt.restype = ctypes.c_long
    out = ctypes.create_string_buffer(max(len(data), 16))
    n = L.pna_oracle_aes_cbc_decrypt(bytes(key), len(key), bytes(iv), bytes(data), ctypes.c_size_t(len(data)), out)
    if n < 0:
        raise ValueError("CBC: bad length or padding")
    return out.raw[:n]


def blake2b(data: bytes, outlen: int = 64) -> bytes:
    out = ctypes.create_string_buffer(outlen)
    lib().pna_oracle_blake2b(bytes(data), ctypes.c_size_t(len(data)), out, ctypes.c_size_t(outlen))
    return out.raw


def sha256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().pna_oracle_sha256(bytes(data), ctypes.c_size_t(len(data)), out)
    return out.raw


def pbkdf2_sha256(password: bytes, salt: bytes, rounds: int, outlen: int = 32) -> bytes:
    out = ctypes.create_string_buffer(outlen)
    lib().pna_oracle_pbkdf2_sha256(bytes(password), ctypes.c_size_t(len(password)), bytes(salt), ctypes.c_size_t(len(salt)),
                                   ctypes.c_uint32(rounds), out, ctypes.c_size_t(outlen))
    return out.raw


def argon2(kind: int, password: bytes, salt: bytes, t_cost: int, m_cost: int, lanes: int, outlen: int = 32) -> bytes:
    """kind: 0 Argon2d, 1 Argon2i, 2 Argon2id (version 0x13)."""
    out = ctypes.create_string_buffer(outlen)
    r = lib().pna_oracle_argon2(kind, bytes(password), len(password), bytes(salt), len(salt), t_cost, m_cost, lanes, out, outlen)
    if r:
        raise ValueError("argon2 parameters rejected")
    return out.raw


def _b64_nopad(s: str) -> bytes:
    import base64
    return base64.b64decode(s + "=" * (-len(s) % 4))


def derive_key_from_phsf(phsf: str, password: bytes, key_len: int = 32) -> bytes:
    """derive_password_hash -- lib/src/hash.rs:47-88: the PHC string (hash part removed, lib/src/entry/write.rs:181-186)
    names the algorithm, its parameters and the B64 salt; the key is the hash output of key_size() bytes (write.rs:146-151)."""
    f = phsf.split("$")
    if len(f) < 4 or f[0] != "":
        raise ValueError("not a PHC string")
    alg = f[1]
    if alg in ("argon2id", "argon2i", "argon2d"):
        if not f[2].startswith("v="):
            raise ValueError("argon2 version missing")
        if int(f[2][2:]) != 19:
            raise ValueError("argon2 version")
        prm = dict(kv.split("=") for kv in f[3].split(","))
        salt = _b64_nopad(f[4])
        return argon2({"argon2d": 0, "argon2i": 1, "argon2id": 2}[alg], password, salt, int(prm["t"]), int(prm["m"]), int(prm["p"]), key_len)
    if alg == "pbkdf2-sha256":
        prm = dict(kv.split("=") for kv in f[2].split(","))
        salt = _b64_nopad(f[3])
        return pbkdf2_sha256(password, salt, int(prm["i"]), int(prm.get("l", key_len)))
    raise ValueError(f"unsupported password hash {alg}")


def decrypt_payload(encryption: int, cipher_mode: int, key: bytes, data: bytes) -> bytes:
    """decrypt_reader -- lib/src/entry/read.rs:59-104: the first 16 bytes of the data stream are the IV (CBC/CTR)."""
    if encryption == 0:
        return bytes(data)
    if encryption != 1:
        raise ValueError("only AES in the oracle")
    iv, body = bytes(data[:16]), bytes(data[16:])
    if len(iv) != 16:
        raise ValueError("missing IV")
    if cipher_mode == 1:
        return aes_ctr(key, iv, body)
    if cipher_mode == 0:
        return aes_cbc_decrypt(key, iv, body)
    raise ValueError("cipher mode not in the oracle")


# ----------------------------------------------------------------------------- cipher mode 2: GCM STREAM (lib/src/cipher/aead.rs, gcm.rs)

STREAM_HEADER_LEN, GCM_TAG_LEN = 75, 16      # aead.rs:15-16
DEFAULT_SEGMENT_SIZE, MAX_SEGMENT_SIZE = 1 << 20, 64 << 20   # aead.rs:17-18


def hkdf_sha256(ikm: bytes, salt: bytes, info: bytes, outlen: int = 32) -> bytes:
    """hkdf_sha256 -- lib/src/cipher/aead.rs:151-157."""
    out = ctypes.create_string_buffer(outlen)
    lib().pna_oracle_hkdf_sha256(bytes(ikm), ctypes.c_size_t(len(ikm)), bytes(salt), ctypes.c_size_t(len(salt)), bytes(info),
                                 ctypes.c_size_t(len(info)), out, ctypes.c_size_t(outlen))
    return out.raw


def aes_gcm(key: bytes, nonce: bytes, data: bytes, tag: bytes | None = None, aad: bytes = b""):
    """Encrypt (tag is None): returns (ciphertext, tag).  Decrypt: returns the plaintext or raises on a tag mismatch."""
    assert len(nonce) == 12
    buf = ctypes.create_string_buffer(bytes(data), max(len(data), 1))
    t = ctypes.create_string_buffer(bytes(tag) if tag is not None else bytes(16), 16)
    r = lib().pna_oracle_aes_gcm(bytes(key), len(key), bytes(nonce), bytes(aad), ctypes.c_size_t(len(aad)), buf, ctypes.c_size_t(len(data)),
                                 t, 0 if tag is None else 1)
    if r:
        raise ValueError("GCM authentication failure")
    return (buf.raw[:len(data)], t.raw) if tag is None else buf.raw[:len(data)]


def key_confirmation(k_master: bytes) -> bytes:
    """aead.rs:161-163: HKDF with an empty salt and the info "PNA-KC-v1"."""
    return hkdf_sha256(k_master, b"", b"PNA-KC-v1")


def stream_header_bytes(salt: bytes, nonce_prefix: bytes, segment_size: int, k_master: bytes) -> bytes:
    """StreamHeader::to_bytes -- aead.rs:130-137: salt(32) || nonce_prefix(7) || segment_size(u32 BE) || key_confirmation(32)."""
    assert len(salt) == 32 and len(nonce_prefix) == 7 and 1 <= segment_size <= MAX_SEGMENT_SIZE
    return bytes(salt) + bytes(nonce_prefix) + segment_size.to_bytes(4, "big") + key_confirmation(k_master)


def entry_context(nonce_prefix: bytes, segment_size: int, header_chunk_type: bytes, header_chunk_data: bytes, phsf: bytes) -> bytes:
    """aead.rs:165-182: "PNA-STREAM-v1" || sha256(type || header data) || sha256(phsf) || nonce_prefix || segment_size BE."""
    return b"PNA-STREAM-v1" + sha256(header_chunk_type + header_chunk_data) + sha256(phsf) + bytes(nonce_prefix) + segment_size.to_bytes(4, "big")


def derive_stream_key(k_master: bytes, salt: bytes, nonce_prefix: bytes, segment_size: int, header_chunk_type: bytes,
                      header_chunk_data: bytes, phsf: bytes) -> bytes:
    """aead.rs:184-199: HKDF(ikm = K_master, salt = header.salt, info = entry_context)."""
    return hkdf_sha256(k_master, salt, entry_context(nonce_prefix, segment_size, header_chunk_type, header_chunk_data, phsf))


def segment_nonce(nonce_prefix: bytes, counter: int, is_final: bool) -> bytes:
    """aead.rs:201-207."""
    return bytes(nonce_prefix) + counter.to_bytes(4, "big") + (b"\x01" if is_final else b"\x00")


def gcm_stream_encrypt(k_stream: bytes, nonce_prefix: bytes, segment_size: int, plain: bytes) -> bytes:
    """GcmEncryptWriter -- gcm.rs:45-90: segments of segment_size bytes, each followed by its tag; a segment is flushed as non-final
    only when more data follows, finish() flushes the last one (possibly empty, possibly full) with the final flag."""
    out, counter, pos = bytearray(), 0, 0
    while len(plain) - pos > segment_size:
        c, t = aes_gcm(k_stream, segment_nonce(nonce_prefix, counter, False), plain[pos:pos + segment_size])
        out += c + t; pos += segment_size; counter += 1
    c, t = aes_gcm(k_stream, segment_nonce(nonce_prefix, counter, True), plain[pos:])
    return bytes(out + c + t)


def gcm_stream_decrypt(k_stream: bytes, nonce_prefix: bytes, segment_size: int, data: bytes) -> bytes:
    """GcmDecryptReader -- gcm.rs:206-290: a segment is final when the stream ends behind it."""
    out, counter, pos = bytearray(), 0, 0
    if len(data) < GCM_TAG_LEN:
        raise ValueError("datastream shorter than a tag")
    while True:
        seg = data[pos:pos + segment_size + GCM_TAG_LEN]
        pos += len(seg)
        final = pos >= len(data)
        if not final and len(seg) < segment_size + GCM_TAG_LEN:
            raise ValueError("short non-final segment")
        if len(seg) < GCM_TAG_LEN:
            raise ValueError("truncated segment")
        out += aes_gcm(k_stream, segment_nonce(nonce_prefix, counter, final), seg[:-GCM_TAG_LEN], tag=seg[-GCM_TAG_LEN:])
        if final:
            return bytes(out)
        counter += 1


def decrypt_payload_gcm(k_master: bytes, data: bytes, header_chunk_type: bytes, header_chunk_data: bytes, phsf: bytes) -> bytes:
    """decrypt_reader, (_, CipherMode::GCM) -- lib/src/entry/read.rs:105-140: 75-byte stream header first, key confirmation checked
    before any segment is processed, K_stream bound to the entry's header chunk and PHSF."""
    if len(data) < STREAM_HEADER_LEN:
        raise ValueError("datastream shorter than the stream header")
    hd = data[:STREAM_HEADER_LEN]
    salt, prefix, seg = hd[:32], hd[32:39], int.from_bytes(hd[39:43], "big")
    if not 1 <= seg <= MAX_SEGMENT_SIZE:
        raise ValueError("segment size out of range")
    if hd[43:75] != key_confirmation(k_master):
        raise ValueError("wrong password (key confirmation)")
    ks = derive_stream_key(k_master, salt, prefix, seg, header_chunk_type, header_chunk_data, phsf)
    return gcm_stream_decrypt(ks, prefix, seg, data[STREAM_HEADER_LEN:])
