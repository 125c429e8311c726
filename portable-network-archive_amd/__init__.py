"""portable-network-archive_amd -- MI355X-native per-entry compression path of PNA (host-side Python binding).

Thin ctypes layer over the C ABI of ``libpna_gpu.so`` (include/pna_gpu.h).  It mirrors the names of the reference's
seam -- ``Compression`` (lib/src/entry/options.rs:241-247), ``CompressionLevel`` defaults
(lib/src/compress/zstandard.rs:13, lib/src/compress/deflate.rs:33-38) and a ``CompressionWriter`` with
``write`` / ``flush`` / ``try_into_inner`` (lib/src/compress.rs:21-76) -- so tests read like the reference's own.

There is no CPU fallback: if the HIP extension is missing or no GPU is usable, construction raises.
(The package directory name contains a hyphen; import it with ``importlib.import_module("portable-network-archive_amd")``.)
"""
from __future__ import annotations

import ctypes
import os
from typing import Iterable, List, Optional, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNA_GPU_LIB") or os.path.join(_HERE, "libpna_gpu.so")

PNA_OK = 0
ALGO_STORE, ALGO_DEFLATE, ALGO_ZSTD = 0, 1, 2
LEVEL_DEFAULT = -1000
# error codes of include/pna_gpu.h
E_NODEVICE, E_INVAL, E_NOMEM, E_DSTSIZE, E_HIP, E_SINK, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6, -7
F_HUF, F_FSE, F_LAZY, F_REP, F_DEFAULT = 1, 2, 4, 8, 0x80000000
F_FAR, F_ADOPT, F_INS2, F_STRONG = 0x10, 0x20, 0x40, 0x80
F_LZ_WAVEPARSE, F_LZ_FUSED = 0x4000, 0x8000                      # forms of the LZ stage (default: split, lane-per-region parse); same bytes
F_STD = F_HUF | F_FSE | F_LAZY | F_FAR | F_ADOPT | F_INS2          # what F_DEFAULT selects

SEG_SIZE = 1 << 20
PART_HEAD, PART_TAIL = 1, 2
BLK_SIZE = 1 << 17
SEQ_CAP = 22528


class Compression:
    """Compression::to_byte() values -- lib/src/entry/options.rs:241-247."""
    No = ALGO_STORE
    Deflate = ALGO_DEFLATE
    ZStandard = ALGO_ZSTD


class PnaGpuError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"pna_gpu error {code}: {text}")
        self.code = code


class Timing(ctypes.Structure):
    _fields_ = [("ms_lz", ctypes.c_double), ("ms_stats", ctypes.c_double), ("ms_lit", ctypes.c_double),
                ("ms_seq", ctypes.c_double), ("ms_pack", ctypes.c_double), ("in_bytes", ctypes.c_uint64),
                ("out_bytes", ctypes.c_uint64), ("n_segments", ctypes.c_uint64), ("n_blocks", ctypes.c_uint64),
                ("ms_frame", ctypes.c_double), ("ms_cipher", ctypes.c_double), ("ms_lz_match", ctypes.c_double),
                ("lz_match_launches", ctypes.c_uint64), ("blk_log", ctypes.c_uint32), ("lz_units", ctypes.c_uint32)]


ENC_NONE, ENC_AES, ENC_CAMELLIA = 0, 1, 2          # Encryption::to_byte()
MODE_CBC, MODE_CTR, MODE_GCM = 0, 1, 2             # CipherMode::to_byte()


class CipherStruct(ctypes.Structure):
    """pna_gpu_cipher (include/pna_gpu.h)."""
    _fields_ = [("encryption", ctypes.c_int), ("cipher_mode", ctypes.c_int), ("key", ctypes.c_uint8 * 32),
                ("phsf", ctypes.c_char_p), ("ivs", ctypes.c_void_p), ("gcm_segment_size", ctypes.c_uint32)]


class MetaStruct(ctypes.Structure):
    """pna_gpu_entry_meta (include/pna_gpu.h)."""
    _fields_ = [("extra", ctypes.POINTER(ctypes.c_void_p)), ("extra_len", ctypes.POINTER(ctypes.c_size_t)),
                ("facets", ctypes.POINTER(ctypes.c_void_p)), ("facets_len", ctypes.POINTER(ctypes.c_size_t))]


class Cipher:
    """What WriteCipher carries (lib/src/entry/write.rs:54-57): algorithm, mode, the derived key, the PHSF string; plus the
    per-entry IVs (None = drawn by the library like random::random_vec)."""

    def __init__(self, key: bytes, phsf: str, mode: int = MODE_CTR, encryption: int = ENC_AES, ivs: Optional[bytes] = None,
                 gcm_segment_size: int = 0):
        assert len(key) == 32
        self.key, self.phsf, self.mode, self.encryption, self.ivs = bytes(key), phsf, mode, encryption, ivs
        self.gcm_segment_size = gcm_segment_size
        self._keep = None

    def struct(self, n: int) -> CipherStruct:
        c = CipherStruct()
        c.encryption, c.cipher_mode = self.encryption, self.mode
        c.key = (ctypes.c_uint8 * 32)(*self.key)
        c.phsf = self.phsf.encode()
        c.gcm_segment_size = self.gcm_segment_size
        if self.ivs is not None:
            assert len(self.ivs) >= (39 if self.mode == MODE_GCM else 16) * n
            self._keep = ctypes.create_string_buffer(bytes(self.ivs), max(len(self.ivs), 1))
            c.ivs = ctypes.cast(self._keep, ctypes.c_void_p)
        else:
            c.ivs = None
        return c


SINK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)
PART_SINK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t)
RAW_ENTRY_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64)
ENTRY_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t)

_lib = None

EXPORTS = [
    "pna_gpu_init", "pna_gpu_set_option", "pna_gpu_shutdown", "pna_gpu_archive_chunked_bound", "pna_gpu_create_archive_chunked_device",
    "pna_gpu_create_archive_chunked_host", "pna_gpu_comm_unique_id", "pna_gpu_comm_init", "pna_gpu_comm_destroy", "pna_gpu_comm_last_error",
    "pna_gpu_host_alloc", "pna_gpu_host_free", "pna_gpu_gather_ordered", "pna_gpu_gather_ordered_start", "pna_gpu_gather_wait", "pna_gpu_gather_ticket", "pna_gpu_gather_wait_for", "pna_gather_verdict", "pna_gather_offsets", "pna_gpu_last_error", "pna_gpu_strerror", "pna_gpu_bound", "pna_gpu_clamp_level",
    "pna_gpu_compress_batch", "pna_gpu_compress_batch_device", "pna_gpu_stream_new", "pna_gpu_stream_write",
    "pna_gpu_stream_flush", "pna_gpu_stream_finish", "pna_gpu_stream_abort", "pna_gpu_compress_solid",
    "pna_gpu_last_timing", "pna_gpu_debug_block", "pna_gpu_debug_lz_stamps", "pna_bench_corpus_fill_device",
    "pna_gpu_archive_bound", "pna_gpu_create_archive_device", "pna_gpu_create_archive_host", "pna_gpu_debug_crc_schedule", "pna_gpu_debug_pinned_bytes",
    "pna_gpu_solid_archive_bound", "pna_gpu_solid_archive_enc_bound", "pna_gpu_create_solid_archive_device", "pna_gpu_create_solid_archive_host",
    "pna_gpu_create_archive_part_device", "pna_gpu_decompress_batch", "pna_gpu_decompress_batch_device",
    "pna_gpu_archive_enc_bound", "pna_gpu_create_archive_enc_device", "pna_gpu_cipher_apply_device", "pna_gpu_create_archive_enc_host",
    "pna_gpu_create_solid_archive_enc_device", "pna_gpu_extract_archive_host", "pna_gpu_zstd_stream_frames_device",
    "pna_gpu_zstd_decompress_open_device", "pna_gpu_inflate_open_device", "pna_gpu_create_archive_meta_device",
    "pna_gpu_create_archive_meta_host", "pna_gpu_stream_stats", "pna_bench_stream_threads",
    # include/pna_archive.h
    "pna_crc32", "pna_archive_new", "pna_archive_add_file", "pna_archive_add_dir", "pna_archive_add_solid",
    "pna_archive_inner_entry_bytes", "pna_archive_finalize", "pna_archive_abort", "pna_create_archive",
    "pna_kdf_pbkdf2_sha256", "pna_create_archive_encrypted", "pna_kdf_argon2", "pna_split_archive", "pna_join_parts",
    "pna_archive_seek_to_end", "pna_archive_list_entries",
    # streaming entries, parts, append (include/pna_gpu.h)
    "pna_gpu_create_archive_part_host", "pna_gpu_create_archive_multi_host", "pna_gpu_append_archive_host", "pna_gpu_stream_entry_begin", "pna_gpu_stream_entry_write",
    "pna_gpu_stream_entry_finish", "pna_gpu_stream_entry_abort",
]


def load_library() -> ctypes.CDLL:
    """dlopen libpna_gpu.so (built in-tree by __graft_entry__.build()).  Fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the compression path)")
    try:  # share PyTorch's HIP runtime when torch is used in the same process
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, u32, u64p = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)
    L.pna_gpu_init.restype = ctypes.c_int
    L.pna_gpu_init.argtypes = [ctypes.POINTER(vp), ctypes.c_int, u32]
    L.pna_gpu_shutdown.restype = None
    L.pna_gpu_shutdown.argtypes = [vp]
    L.pna_gpu_last_error.restype = ctypes.c_char_p
    L.pna_gpu_last_error.argtypes = [vp]
    L.pna_gpu_strerror.restype = ctypes.c_char_p
    L.pna_gpu_strerror.argtypes = [ctypes.c_int]
    L.pna_gpu_bound.restype = sz
    L.pna_gpu_bound.argtypes = [ctypes.c_int, sz]
    L.pna_gpu_clamp_level.restype = ctypes.c_int
    L.pna_gpu_clamp_level.argtypes = [ctypes.c_int, ctypes.c_int]
    L.pna_gpu_compress_batch.restype = ctypes.c_int
    L.pna_gpu_compress_batch.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(vp), ctypes.POINTER(sz),
                                         ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(sz)]
    L.pna_gpu_compress_batch_device.restype = ctypes.c_int
    L.pna_gpu_compress_batch_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, vp, u64p, u64p, vp, sz, u64p, vp]
    L.pna_gpu_archive_bound.restype = sz
    L.pna_gpu_archive_bound.argtypes = [ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), u64p]
    L.pna_gpu_create_archive_device.restype = ctypes.c_int
    L.pna_gpu_create_archive_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), vp, u64p, u64p,
                                                vp, sz, u64p, u64p, vp]
    L.pna_gpu_create_archive_part_device.restype = ctypes.c_int
    L.pna_gpu_create_archive_part_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), vp, u64p, u64p,
                                                     vp, sz, u64p, u64p, u32, vp]
    L.pna_gpu_archive_enc_bound.restype = sz
    L.pna_gpu_archive_enc_bound.argtypes = [ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), u64p, ctypes.POINTER(CipherStruct)]
    L.pna_gpu_create_archive_enc_device.restype = ctypes.c_int
    L.pna_gpu_create_archive_enc_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), vp, u64p, u64p,
                                                    ctypes.POINTER(CipherStruct), vp, sz, u64p, u64p, u32, vp]
    L.pna_gpu_cipher_apply_device.restype = ctypes.c_int
    L.pna_gpu_cipher_apply_device.argtypes = [vp, ctypes.POINTER(CipherStruct), ctypes.c_int, sz, vp, u64p, u64p, vp]
    L.pna_gpu_decompress_batch.restype = ctypes.c_int
    L.pna_gpu_decompress_batch.argtypes = [vp, ctypes.c_int, sz, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp), ctypes.POINTER(sz)]
    L.pna_gpu_decompress_batch_device.restype = ctypes.c_int
    L.pna_gpu_decompress_batch_device.argtypes = [vp, ctypes.c_int, sz, vp, u64p, u64p, vp, u64p, u64p, vp]
    L.pna_gpu_solid_archive_bound.restype = sz
    L.pna_gpu_solid_archive_bound.argtypes = [ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), u64p]
    L.pna_gpu_solid_archive_enc_bound.restype = sz
    L.pna_gpu_solid_archive_enc_bound.argtypes = [ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), u64p, ctypes.POINTER(CipherStruct)]
    L.pna_gpu_create_solid_archive_device.restype = ctypes.c_int
    L.pna_gpu_create_solid_archive_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), vp, u64p, u64p,
                                                      vp, sz, u64p, vp]
    L.pna_gpu_create_solid_archive_enc_device.restype = ctypes.c_int
    L.pna_gpu_create_solid_archive_enc_device.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), vp, u64p, u64p,
                                                          ctypes.POINTER(CipherStruct), vp, sz, u64p, vp]
    L.pna_gpu_debug_crc_schedule.restype = ctypes.c_uint32
    L.pna_gpu_debug_crc_schedule.argtypes = [ctypes.c_char_p, sz]
    L.pna_gpu_stream_new.restype = ctypes.c_int
    L.pna_gpu_stream_new.argtypes = [vp, ctypes.c_int, ctypes.c_int, SINK_FN, vp, ctypes.POINTER(vp)]
    L.pna_gpu_stream_write.restype = ctypes.c_int
    L.pna_gpu_stream_write.argtypes = [vp, ctypes.c_char_p, sz]
    L.pna_gpu_stream_flush.restype = ctypes.c_int
    L.pna_gpu_stream_flush.argtypes = [vp]
    L.pna_gpu_stream_finish.restype = ctypes.c_int
    L.pna_gpu_stream_finish.argtypes = [vp]
    L.pna_gpu_stream_abort.restype = None
    L.pna_gpu_stream_abort.argtypes = [vp]
    L.pna_gpu_stream_stats.restype = ctypes.c_int
    L.pna_gpu_stream_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.pna_bench_stream_threads.restype = ctypes.c_double
    L.pna_bench_stream_threads.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint, sz, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(sz),
                                           ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
    L.pna_gpu_compress_solid.restype = ctypes.c_int
    L.pna_gpu_compress_solid.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, sz, SINK_FN, vp]
    L.pna_gpu_last_timing.restype = ctypes.c_int
    L.pna_gpu_last_timing.argtypes = [vp, ctypes.POINTER(Timing)]
    L.pna_gpu_debug_block.restype = ctypes.c_int
    L.pna_gpu_debug_block.argtypes = [vp, u32, u64p, u32, ctypes.POINTER(u32), ctypes.c_char_p, u32, ctypes.POINTER(u32)]
    L.pna_gpu_debug_lz_stamps.restype = ctypes.c_int
    L.pna_gpu_debug_lz_stamps.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong)]
    L.pna_bench_corpus_fill_device.restype = ctypes.c_int
    L.pna_bench_corpus_fill_device.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                               ctypes.c_uint64, vp, vp]
    L.pna_crc32.restype = ctypes.c_uint32
    L.pna_crc32.argtypes = [ctypes.c_uint32, ctypes.c_char_p, sz]
    L.pna_archive_new.restype = ctypes.c_int
    L.pna_archive_new.argtypes = [SINK_FN, vp, u32, ctypes.POINTER(vp)]
    L.pna_archive_add_file.restype = ctypes.c_int
    L.pna_archive_add_file.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int64, ctypes.c_char_p, sz, u32]
    L.pna_archive_add_dir.restype = ctypes.c_int
    L.pna_archive_add_dir.argtypes = [vp, ctypes.c_char_p]
    L.pna_archive_add_solid.restype = ctypes.c_int
    L.pna_archive_add_solid.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(sz), sz]
    L.pna_archive_inner_entry_bytes.restype = sz
    L.pna_archive_inner_entry_bytes.argtypes = [ctypes.c_char_p, ctypes.c_char_p, sz, vp, sz]
    L.pna_archive_finalize.restype = ctypes.c_int
    L.pna_archive_finalize.argtypes = [vp]
    L.pna_archive_abort.restype = None
    L.pna_archive_abort.argtypes = [vp]
    L.pna_create_archive.restype = ctypes.c_int
    L.pna_create_archive.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p),
                                     ctypes.POINTER(vp), ctypes.POINTER(sz), SINK_FN, vp]
    L.pna_gpu_create_archive_enc_host.restype = ctypes.c_int
    L.pna_gpu_create_archive_enc_host.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(vp),
                                                  ctypes.POINTER(sz), ctypes.POINTER(CipherStruct), SINK_FN, vp]
    L.pna_gpu_create_archive_meta_host.restype = ctypes.c_int
    L.pna_gpu_create_archive_meta_host.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(vp),
                                                   ctypes.POINTER(sz), ctypes.POINTER(CipherStruct), ctypes.POINTER(MetaStruct), SINK_FN, vp]
    L.pna_gpu_extract_archive_host.restype = ctypes.c_int
    L.pna_gpu_extract_archive_host.argtypes = [vp, ctypes.c_char_p, sz, ctypes.c_char_p, sz, ENTRY_FN, vp]
    L.pna_kdf_pbkdf2_sha256.restype = ctypes.c_int
    L.pna_kdf_pbkdf2_sha256.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p, sz, u32, ctypes.c_char_p, sz, ctypes.c_char_p, sz]
    L.pna_split_archive.restype = ctypes.c_int
    L.pna_split_archive.argtypes = [ctypes.c_char_p, sz, sz, PART_SINK_FN, vp, ctypes.POINTER(u32)]
    L.pna_join_parts.restype = ctypes.c_int
    L.pna_join_parts.argtypes = [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(sz), sz, SINK_FN, vp]
    L.pna_kdf_argon2.restype = ctypes.c_int
    L.pna_kdf_argon2.argtypes = [ctypes.c_int, ctypes.c_char_p, sz, ctypes.c_char_p, sz, u32, u32, u32, ctypes.c_char_p, sz]
    L.pna_create_archive_encrypted.restype = ctypes.c_int
    L.pna_create_archive_encrypted.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(vp),
                                               ctypes.POINTER(sz), ctypes.c_char_p, sz, ctypes.c_int, u32, SINK_FN, vp]
    L.pna_archive_seek_to_end.restype = ctypes.c_int
    L.pna_archive_seek_to_end.argtypes = [ctypes.c_char_p, sz, u64p, ctypes.POINTER(ctypes.c_int)]
    L.pna_archive_list_entries.restype = ctypes.c_int
    L.pna_archive_list_entries.argtypes = [ctypes.c_char_p, sz, RAW_ENTRY_FN, vp]
    L.pna_gpu_create_archive_part_host.restype = ctypes.c_int
    L.pna_gpu_create_archive_part_host.argtypes = [vp, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(vp),
                                                   ctypes.POINTER(sz), u32, SINK_FN, vp]
    L.pna_gpu_create_archive_multi_host.restype = ctypes.c_int
    L.pna_gpu_create_archive_multi_host.argtypes = [ctypes.POINTER(vp), sz, ctypes.c_int, ctypes.c_int, sz, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(vp),
                                                    ctypes.POINTER(sz), SINK_FN, vp]
    L.pna_gpu_append_archive_host.restype = ctypes.c_int
    L.pna_gpu_append_archive_host.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, sz, sz, ctypes.POINTER(ctypes.c_char_p),
                                              ctypes.POINTER(vp), ctypes.POINTER(sz), u64p, SINK_FN, vp]
    L.pna_gpu_stream_entry_begin.restype = ctypes.c_int
    L.pna_gpu_stream_entry_begin.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, sz, u32, SINK_FN, vp, ctypes.POINTER(vp)]
    L.pna_gpu_stream_entry_write.restype = ctypes.c_int
    L.pna_gpu_stream_entry_write.argtypes = [vp, ctypes.c_char_p, sz]
    L.pna_gpu_stream_entry_finish.restype = ctypes.c_int
    L.pna_gpu_stream_entry_finish.argtypes = [vp]
    L.pna_gpu_stream_entry_abort.restype = None
    L.pna_gpu_stream_entry_abort.argtypes = [vp]
    _lib = L
    return L


def bound(algo: int, n: int) -> int:
    return load_library().pna_gpu_bound(algo, n)


def clamp_level(algo: int, level: int = LEVEL_DEFAULT) -> int:
    """From<CompressionLevel> mappings -- compress/zstandard.rs:43-57, compress/deflate.rs:89-101."""
    return load_library().pna_gpu_clamp_level(algo, level)


class Context:
    """One GPU context (pna_gpu_init).  Raises PnaGpuError(PNA_E_NODEVICE) without a usable MI355X."""

    def __init__(self, device: int = 0, flags: int = F_DEFAULT):
        self._L = load_library()
        h = ctypes.c_void_p()
        rc = self._L.pna_gpu_init(ctypes.byref(h), device, flags)
        if rc != PNA_OK:
            raise PnaGpuError(rc, self._L.pna_gpu_strerror(rc).decode())
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.pna_gpu_shutdown(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int) -> None:
        if rc != PNA_OK:
            raise PnaGpuError(rc, self._L.pna_gpu_last_error(self._h).decode() or self._L.pna_gpu_strerror(rc).decode())

    # ---- batch of independent entries (host buffers)
    def compress_batch(self, entries: Sequence[bytes], algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT) -> List[bytes]:
        n = len(entries)
        if n == 0:
            return []
        srcs = [e if isinstance(e, bytes) else bytes(e) for e in entries]       # passed by pointer, not copied
        caps = [self._L.pna_gpu_bound(algo, len(e)) for e in srcs]
        arena = bytearray(sum(caps) + 1)                                        # one output arena, entry i at offs[i]
        base = ctypes.addressof((ctypes.c_char * len(arena)).from_buffer(arena))
        offs, pos = [], 0
        for c in caps:
            offs.append(pos); pos += c
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        a_src = (vp * n)(*[ctypes.cast(ctypes.c_char_p(b), vp) for b in srcs])
        a_len = (sz * n)(*[len(e) for e in srcs])
        a_dst = (vp * n)(*[base + o for o in offs])
        a_cap = (sz * n)(*caps)
        a_out = (sz * n)()
        self._check(self._L.pna_gpu_compress_batch(self._h, algo, level, n, a_src, a_len, a_dst, a_cap, a_out))
        mv = memoryview(arena)
        return [bytes(mv[offs[i]:offs[i] + a_out[i]]) for i in range(n)]

    # ---- batch resident in HBM (raw device pointers; torch tensors via .data_ptr())
    def compress_batch_device(self, d_src: int, src_off: Sequence[int], src_len: Sequence[int], d_dst: int, dst_cap: int,
                              algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT, stream: int = 0) -> List[int]:
        n = len(src_len)
        a_off = (ctypes.c_uint64 * (n + 1))(*src_off) if len(src_off) == n + 1 else (ctypes.c_uint64 * (n + 1))(*list(src_off), 0)
        a_len = (ctypes.c_uint64 * max(n, 1))(*src_len)
        a_out = (ctypes.c_uint64 * (n + 1))()
        self._check(self._L.pna_gpu_compress_batch_device(self._h, algo, level, n, ctypes.c_void_p(d_src), a_off, a_len,
                                                          ctypes.c_void_p(d_dst), dst_cap, a_out,
                                                          ctypes.c_void_p(stream) if stream else None))
        return list(a_out)

    def create_archive_device(self, names: Sequence[str], d_src: int, src_off: Sequence[int], src_len: Sequence[int], d_dst: int,
                              dst_cap: int, algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT, stream: int = 0,
                              _cache: Optional[dict] = None, part: int = PART_HEAD | PART_TAIL, cipher: Optional[Cipher] = None,
                              want_offsets: bool = True):
        """Whole non-solid archive assembled in HBM (pna_gpu_create_archive_enc_device; `part` selects whether this shard
        carries the archive header / AEND; `cipher` adds the AES stage between compression and chunk CRC).
        Returns (archive_len, entry_off)."""
        n = len(src_len)
        if _cache is not None and "a" in _cache:
            a_names, a_off, a_len = _cache["a"]
        else:
            a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
            a_off = (ctypes.c_uint64 * (n + 1))(*(list(src_off)[:n] + [0]))
            a_len = (ctypes.c_uint64 * max(n, 1))(*src_len)
            if _cache is not None:
                _cache["a"] = (a_names, a_off, a_len)
        if _cache is not None and "out" in _cache:
            a_out = _cache["out"]
        else:
            a_out = (ctypes.c_uint64 * (n + 1))()
            if _cache is not None:
                _cache["out"] = a_out
        total = ctypes.c_uint64()
        if cipher is not None and _cache is not None:
            cs = _cache.get("cs") or _cache.setdefault("cs", cipher.struct(n))
        else:
            cs = cipher.struct(n) if cipher is not None else None
        self._check(self._L.pna_gpu_create_archive_enc_device(self._h, algo, level, n, a_names, ctypes.c_void_p(d_src), a_off, a_len,
                                                              ctypes.byref(cs) if cs is not None else None,
                                                              ctypes.c_void_p(d_dst), dst_cap, a_out, ctypes.byref(total), part,
                                                              ctypes.c_void_p(stream) if stream else None))
        return total.value, (list(a_out) if want_offsets else None)      # the list conversion costs milliseconds for 10^5 entries

    def cipher_apply_device(self, cipher: Cipher, d_buf: int, off: Sequence[int], length: Sequence[int], decrypt: bool = False,
                            stream: int = 0) -> None:
        """The cipher stage alone, in place, over byte ranges of a device buffer (range i uses cipher.ivs[16 i ..))."""
        n = len(length)
        cs = cipher.struct(n)
        mk = lambda xs: (ctypes.c_uint64 * max(n, 1))(*list(xs)[:n])
        self._check(self._L.pna_gpu_cipher_apply_device(self._h, ctypes.byref(cs), 1 if decrypt else 0, n, ctypes.c_void_p(d_buf),
                                                        mk(off), mk(length), ctypes.c_void_p(stream) if stream else None))

    def create_solid_archive_device(self, names: Sequence[str], d_src: int, src_off: Sequence[int], src_len: Sequence[int], d_dst: int,
                                    dst_cap: int, algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT, stream: int = 0,
                                    _cache: Optional[dict] = None, cipher: Optional[Cipher] = None) -> int:
        """`pna create --solid` assembled in HBM (pna_gpu_create_solid_archive_enc_device; `cipher`: CTR with one IV, or GCM with one salt || nonce prefix).
        Returns the archive length."""
        n = len(src_len)
        if _cache is not None and "a" in _cache:
            a_names, a_off, a_len = _cache["a"]
        else:
            a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
            a_off = (ctypes.c_uint64 * (n + 1))(*(list(src_off)[:n] + [0]))
            a_len = (ctypes.c_uint64 * max(n, 1))(*src_len)
            if _cache is not None:
                _cache["a"] = (a_names, a_off, a_len)
        total = ctypes.c_uint64()
        cs = cipher.struct(1) if cipher is not None else None
        self._check(self._L.pna_gpu_create_solid_archive_enc_device(self._h, algo, level, n, a_names, ctypes.c_void_p(d_src), a_off, a_len,
                                                                    ctypes.byref(cs) if cs is not None else None,
                                                                    ctypes.c_void_p(d_dst), dst_cap, ctypes.byref(total),
                                                                    ctypes.c_void_p(stream) if stream else None))
        return total.value

    def decompress_batch(self, payloads: Sequence[bytes], raw_sizes: Sequence[int], algo: int = ALGO_ZSTD) -> List[bytes]:
        """decompress_reader for a batch of entries: payload i (concatenated FDAT bodies) -> raw_sizes[i] bytes."""
        n = len(payloads)
        keep = [p if isinstance(p, bytes) else bytes(p) for p in payloads]
        outs = [ctypes.create_string_buffer(max(r, 1)) for r in raw_sizes]
        a_src = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in keep])
        a_sl = (ctypes.c_size_t * max(n, 1))(*[len(b) for b in keep])
        a_dst = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(o, ctypes.c_void_p) for o in outs])
        a_rl = (ctypes.c_size_t * max(n, 1))(*raw_sizes)
        self._check(self._L.pna_gpu_decompress_batch(self._h, algo, n, a_src, a_sl, a_dst, a_rl))
        return [o.raw[:r] for o, r in zip(outs, raw_sizes)]

    def decompress_batch_device(self, d_src: int, src_off: Sequence[int], src_len: Sequence[int], d_dst: int, dst_off: Sequence[int],
                                raw_len: Sequence[int], algo: int = ALGO_ZSTD, stream: int = 0) -> None:
        n = len(src_len)
        mk = lambda xs: (ctypes.c_uint64 * max(n, 1))(*list(xs)[:n])
        self._check(self._L.pna_gpu_decompress_batch_device(self._h, algo, n, ctypes.c_void_p(d_src), mk(src_off), mk(src_len),
                                                            ctypes.c_void_p(d_dst), mk(dst_off), mk(raw_len),
                                                            ctypes.c_void_p(stream) if stream else None))

    def create_archive_chunked_device(self, names: Sequence[str], d_src: int, src_off: Sequence[int], src_len: Sequence[int], d_dst: int, dst_cap: int,
                                      max_chunk_size: int, algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT, cipher: Optional[Cipher] = None,
                                      part: int = PART_HEAD | PART_TAIL):
        """pna_gpu_create_archive_chunked_device: the archive in HBM with FDAT chunks of at most max_chunk_size bytes.  Returns (archive_len, entry_off)."""
        n = len(src_len)
        a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
        a_off = (ctypes.c_uint64 * (n + 1))(*(list(src_off)[:n] + [0]))
        a_len = (ctypes.c_uint64 * max(n, 1))(*src_len)
        a_out = (ctypes.c_uint64 * (n + 1))()
        total = ctypes.c_uint64()
        cs = cipher.struct(n) if cipher is not None else None
        f = self._L.pna_gpu_create_archive_chunked_device
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        self._check(f(self._h, algo, level, n, a_names, ctypes.c_void_p(d_src), a_off, a_len, ctypes.byref(cs) if cs is not None else None, None,
                      max_chunk_size, ctypes.c_void_p(d_dst), dst_cap, a_out, ctypes.byref(total), part, None))
        return total.value, list(a_out)

    def set_option(self, name: str, value: int) -> None:
        """pna_gpu_set_option: tuning knobs of the context (include/pna_gpu.h lists them)."""
        self._L.pna_gpu_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_long]
        self._check(self._L.pna_gpu_set_option(self._h, name.encode(), value))

    def options(self, **kw):
        """Context manager: set options for the duration of a block, then put the given previous values back (tests: `with ctx.options(sub_mib=(128, 1024))`
        -- value, value to restore)."""
        ctx = self

        class _O:
            def __enter__(self_):
                for k, (v, _) in kw.items():
                    ctx.set_option(k, v)

            def __exit__(self_, *a):
                for k, (_, r) in kw.items():
                    ctx.set_option(k, r)
        return _O()

    def timing(self) -> Timing:
        t = Timing()
        self._check(self._L.pna_gpu_last_timing(self._h, ctypes.byref(t)))
        return t

    def debug_block(self, block: int):
        """LZ-stage output of one block of the last device batch: (list of (ll, ml, off), literal bytes)."""
        seqs = (ctypes.c_uint64 * SEQ_CAP)()
        lits = ctypes.create_string_buffer(BLK_SIZE)
        ns, nl = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self._L.pna_gpu_debug_block(self._h, block, seqs, SEQ_CAP, ctypes.byref(ns), lits, BLK_SIZE, ctypes.byref(nl)))
        out = [((s >> 38) & 0x3FFFF, (s >> 20) & 0x3FFFF, s & 0xFFFFF) for s in seqs[:ns.value]]
        return out, lits.raw[:nl.value]

    def lz_stamps(self):
        a = (ctypes.c_ulonglong * 8)()
        self._check(self._L.pna_gpu_debug_lz_stamps(self._h, a))
        return list(a)

    def corpus_fill_device(self, kind: int, first_file: int, n_files: int, file_len: int, stride: int, d_dst: int) -> None:
        self._check(self._L.pna_bench_corpus_fill_device(self._h, kind, first_file, n_files, file_len, stride,
                                                         ctypes.c_void_p(d_dst), None))

    # ---- streaming facade
    def writer(self, sink, algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT) -> "CompressionWriter":
        """One writer per entry; writers of one context may be driven from many threads at once (their finishes are batched)."""
        return CompressionWriter(self, sink, algo, level)

    def stream_stats(self):
        """(device batches, entries, largest batch) run so far on behalf of CompressionWriter.try_into_inner()."""
        b, e, m = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._L.pna_gpu_stream_stats(self._h, ctypes.byref(b), ctypes.byref(e), ctypes.byref(m)))
        return b.value, e.value, m.value

    def bench_stream_threads(self, entries: Sequence[bytes], threads: int, algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT):
        """The reference's one-entry-per-task fan-out on `threads` native host threads over the streaming facade; (seconds, compressed bytes)."""
        n = len(entries)
        keep = [ctypes.create_string_buffer(e, len(e)) for e in entries]
        src = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(k, ctypes.c_void_p) for k in keep])
        sl = (ctypes.c_size_t * max(n, 1))(*[len(e) for e in entries])
        out, rc = ctypes.c_uint64(), ctypes.c_int()
        secs = self._L.pna_bench_stream_threads(self._h, algo, level, threads, n, src, sl, ctypes.byref(out), ctypes.byref(rc))
        self._check(rc.value)
        return secs, out.value

    def compress_solid(self, data: bytes, algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT) -> List[bytes]:
        """SolidArchive payload: returns the pieces pushed to the sink (one SDAT chunk each in the reference)."""
        pieces: List[bytes] = []

        def _sink(_u, buf, n):
            pieces.append(ctypes.string_at(buf, n))
            return 0
        cb = SINK_FN(_sink)
        self._check(self._L.pna_gpu_compress_solid(self._h, algo, level, bytes(data), len(data), cb, None))
        return pieces


class CompressionWriter:
    """Shape of CompressionWriter<W> -- lib/src/compress.rs:21-76: write(), flush(), try_into_inner() (= finish)."""

    def __init__(self, ctx: Context, sink, algo: int, level: int):
        self._ctx, self._sink_obj = ctx, sink

        def _sink(_u, buf, n):
            try:
                sink.write(ctypes.string_at(buf, n))
                return 0
            except Exception:
                return 1
        self._cb = SINK_FN(_sink)
        h = ctypes.c_void_p()
        ctx._check(ctx._L.pna_gpu_stream_new(ctx._h, algo, level, self._cb, None, ctypes.byref(h)))
        self._h = h

    def write(self, data: bytes) -> int:
        self._ctx._check(self._ctx._L.pna_gpu_stream_write(self._h, bytes(data), len(data)))
        return len(data)

    def flush(self) -> None:
        self._ctx._check(self._ctx._L.pna_gpu_stream_flush(self._h))

    def try_into_inner(self):
        h, self._h = self._h, None
        self._ctx._check(self._ctx._L.pna_gpu_stream_finish(h))
        return self._sink_obj


class Archive:
    """Archive<W> writer -- lib/src/archive/write.rs: write_header (constructor), add_entry, finalize.

    The sink is any object with .write(bytes).  Payloads handed to add_file() are already-compressed streams."""

    def __init__(self, sink, archive_number: int = 0):
        self._L = load_library()
        self._sink_obj = sink

        def _sink(_u, buf, n):
            try:
                sink.write(ctypes.string_at(buf, n))
                return 0
            except Exception:
                return 1
        self._cb = SINK_FN(_sink)
        h = ctypes.c_void_p()
        rc = self._L.pna_archive_new(self._cb, None, archive_number, ctypes.byref(h))
        if rc:
            raise PnaGpuError(rc, self._L.pna_gpu_strerror(rc).decode())
        self._h = h

    def _check(self, rc):
        if rc:
            raise PnaGpuError(rc, self._L.pna_gpu_strerror(rc).decode())

    def add_file(self, name: str, compression: int, raw_size: Optional[int], payload: bytes, max_chunk_size: int = 0):
        self._check(self._L.pna_archive_add_file(self._h, name.encode(), compression, -1 if raw_size is None else raw_size,
                                                 bytes(payload), len(payload), max_chunk_size))

    def add_dir(self, name: str):
        self._check(self._L.pna_archive_add_dir(self._h, name.encode()))

    def add_solid(self, compression: int, pieces: Sequence[bytes]):
        n = len(pieces)
        bufs = [ctypes.create_string_buffer(bytes(p), max(len(p), 1)) for p in pieces]
        pp = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(b, ctypes.c_void_p) for b in bufs])
        pl = (ctypes.c_size_t * max(n, 1))(*[len(p) for p in pieces])
        self._check(self._L.pna_archive_add_solid(self._h, compression, pp, pl, n))

    def finalize(self):
        h, self._h = self._h, None
        self._check(self._L.pna_archive_finalize(h))
        return self._sink_obj


def inner_entry_bytes(name: str, data: bytes) -> bytes:
    """One STORE entry serialised as chunk bytes (what SolidArchive::add_entry feeds the compressor)."""
    L = load_library()
    need = L.pna_archive_inner_entry_bytes(name.encode(), bytes(data), len(data), None, 0)
    buf = ctypes.create_string_buffer(need)
    got = L.pna_archive_inner_entry_bytes(name.encode(), bytes(data), len(data), buf, need)
    return buf.raw[:got]


def archive_bound(algo: int, names: Sequence[str], src_len: Sequence[int]) -> int:
    n = len(src_len)
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_len = (ctypes.c_uint64 * max(n, 1))(*src_len)
    return load_library().pna_gpu_archive_bound(algo, n, a_names, a_len)


def archive_enc_bound(algo: int, names: Sequence[str], src_len: Sequence[int], cipher: Optional[Cipher]) -> int:
    n = len(src_len)
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_len = (ctypes.c_uint64 * max(n, 1))(*src_len)
    cs = cipher.struct(0) if cipher is not None and cipher.ivs is None else (cipher.struct(n) if cipher is not None else None)
    return load_library().pna_gpu_archive_enc_bound(algo, n, a_names, a_len, ctypes.byref(cs) if cs is not None else None)


def archive_chunked_bound(algo: int, names: Sequence[str], src_len: Sequence[int], max_chunk_size: int, cipher: Optional[Cipher] = None) -> int:
    n = len(src_len)
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_len = (ctypes.c_uint64 * max(n, 1))(*src_len)
    cs = cipher.struct(0) if cipher is not None and cipher.ivs is None else (cipher.struct(n) if cipher is not None else None)
    L = load_library()
    L.pna_gpu_archive_chunked_bound.restype = ctypes.c_size_t
    L.pna_gpu_archive_chunked_bound.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    return L.pna_gpu_archive_chunked_bound(algo, n, a_names, a_len, ctypes.byref(cs) if cs is not None else None, max_chunk_size)


def create_archive_chunked(ctx: "Context", names: Sequence[str], entries: Sequence[bytes], max_chunk_size: int, algo: int = ALGO_ZSTD,
                           level: int = LEVEL_DEFAULT, cipher: Optional[Cipher] = None, part: int = PART_HEAD | PART_TAIL) -> bytes:
    """pna_gpu_create_archive_chunked_host: the bounded host pipeline with FlattenWriter::max_chunk_size (lib/src/util/io.rs:60-77)."""
    L = load_library()
    n = len(entries)
    out = bytearray()

    def _sink(_u, buf, k):
        out.extend((ctypes.c_char * k).from_address(buf))
        return 0
    cb = SINK_FN(_sink)
    bufs = [e if isinstance(e, bytes) else bytes(e) for e in entries]
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_src = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in bufs])
    a_len = (ctypes.c_size_t * max(n, 1))(*[len(e) for e in entries])
    cs = cipher.struct(n) if cipher is not None else None
    L.pna_gpu_create_archive_chunked_host.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, SINK_FN, ctypes.c_void_p]
    rc = L.pna_gpu_create_archive_chunked_host(ctx._h, algo, level, n, a_names, a_src, a_len, ctypes.byref(cs) if cs is not None else None, None,
                                               max_chunk_size, part, cb, None)
    if rc:
        raise PnaGpuError(rc, L.pna_gpu_last_error(ctx._h).decode() or L.pna_gpu_strerror(rc).decode())
    return bytes(out)


def gather_verdict(sizes: Sequence[int], caps: Sequence[int], root: int = 0):
    """pna_gather_verdict: what every rank decides from the all-gathered (size, capacity) pairs -- (rc, offsets); rc = PNA_E_DSTSIZE when the parts
    do not fit the root's capacity (the same answer on every rank, so an overflow is an error everywhere and never a hang)."""
    n = len(sizes)
    pairs = (ctypes.c_uint64 * (2 * n))(*[v for sc in zip(sizes, caps) for v in sc])
    o = (ctypes.c_uint64 * (n + 1))()
    L = load_library()
    L.pna_gather_verdict.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rc = L.pna_gather_verdict(pairs, n, root, None, o)
    return rc, list(o)


def gather_offsets(sizes: Sequence[int]) -> List[int]:
    """pna_gather_offsets: where every rank's part starts in the gathered stream (+ the total)."""
    n = len(sizes)
    a = (ctypes.c_uint64 * max(n, 1))(*sizes)
    o = (ctypes.c_uint64 * (n + 1))()
    rc = load_library().pna_gather_offsets(a, n, o)
    if rc:
        raise PnaGpuError(rc, "pna_gather_offsets")
    return list(o)


class Comm:
    """pna_gpu_comm: the RCCL communicator of the ordered gather (one process per GPU).  `Comm.unique_id()` on rank 0, the 128 bytes to all ranks."""

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        rc = load_library().pna_gpu_comm_unique_id(buf)
        if rc:
            raise PnaGpuError(rc, "pna_gpu_comm_unique_id (is RCCL installed?)")
        return buf.raw

    def __init__(self, device: int, uid: bytes, nranks: int, rank: int):
        self._L = load_library()
        self._L.pna_gpu_comm_last_error.restype = ctypes.c_char_p
        self._L.pna_gpu_comm_last_error.argtypes = [ctypes.c_void_p]
        h = ctypes.c_void_p()
        rc = self._L.pna_gpu_comm_init(device, bytes(uid), nranks, rank, ctypes.byref(h))
        if rc:
            raise PnaGpuError(rc, "pna_gpu_comm_init")
        self._h, self.nranks, self.rank = h, nranks, rank

    def gather_ordered(self, d_local: int, local_len: int, d_out: int = 0, out_cap: int = 0, root: int = 0, stream: int = 0):
        """Returns (sizes of all ranks' parts, total); on `root` the parts lie in d_out in rank order."""
        sizes = (ctypes.c_uint64 * self.nranks)()
        total = ctypes.c_uint64()
        f = self._L.pna_gpu_gather_ordered
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        rc = f(self._h, ctypes.c_void_p(d_local), local_len, root, ctypes.c_void_p(d_out), out_cap, sizes, ctypes.byref(total), ctypes.c_void_p(stream) if stream else None)
        if rc:
            raise PnaGpuError(rc, self._L.pna_gpu_comm_last_error(self._h).decode())
        return list(sizes), total.value

    def gather_ordered_start(self, d_local: int, local_len: int, d_out: int = 0, out_cap: int = 0, root: int = 0, stream: int = 0):
        """Posts the gather behind the work queued on `stream` and returns (sizes, total) as soon as the size exchange is through; the transfers run on the
        communicator's own stream until gather_wait().  PnaGpuError(PNA_E_DSTSIZE) is raised on EVERY rank when the parts do not fit the root's d_out."""
        sizes = (ctypes.c_uint64 * self.nranks)()
        total = ctypes.c_uint64()
        f = self._L.pna_gpu_gather_ordered_start
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        rc = f(self._h, ctypes.c_void_p(d_local), local_len, root, ctypes.c_void_p(d_out), out_cap, sizes, ctypes.byref(total), ctypes.c_void_p(stream) if stream else None)
        if rc:
            raise PnaGpuError(rc, self._L.pna_gpu_comm_last_error(self._h).decode())
        return list(sizes), total.value

    def ticket(self) -> int:
        """The latest posted gather's ticket (= how many this communicator has posted)."""
        self._L.pna_gpu_gather_ticket.argtypes = [ctypes.c_void_p]
        self._L.pna_gpu_gather_ticket.restype = ctypes.c_uint64
        return int(self._L.pna_gpu_gather_ticket(self._h))

    def gather_wait(self, ticket: int = 0):
        """Blocks until the gathers up to `ticket` are done (0: all posted ones)."""
        self._L.pna_gpu_gather_wait_for.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        rc = self._L.pna_gpu_gather_wait_for(self._h, ticket)
        if rc:
            raise PnaGpuError(rc, self._L.pna_gpu_comm_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.pna_gpu_comm_destroy.argtypes = [ctypes.c_void_p]
            self._L.pna_gpu_comm_destroy(self._h)
            self._h = None


def solid_archive_bound(algo: int, names: Sequence[str], src_len: Sequence[int]) -> int:
    n = len(src_len)
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_len = (ctypes.c_uint64 * max(n, 1))(*src_len)
    return load_library().pna_gpu_solid_archive_bound(algo, n, a_names, a_len)


def solid_archive_enc_bound(algo: int, names: Sequence[str], src_len: Sequence[int], cipher: Optional[Cipher]) -> int:
    n = len(src_len)
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_len = (ctypes.c_uint64 * max(n, 1))(*src_len)
    cs = cipher.struct(1) if cipher is not None else None
    return load_library().pna_gpu_solid_archive_enc_bound(algo, n, a_names, a_len, ctypes.byref(cs) if cs is not None else None)


def crc_schedule(payload: bytes) -> int:
    """crc32(b"FDAT" + payload) computed by the host walk through k_frame's lane schedule (table self-check)."""
    return load_library().pna_gpu_debug_crc_schedule(payload, len(payload))


def crc32(data: bytes, crc: int = 0) -> int:
    return load_library().pna_crc32(crc, bytes(data), len(data))


class HostSlot:
    """pna_gpu_host_alloc: a page-locked buffer the host reads its files into (read_exact into the slot instead of fs::read into a Vec,
    cli/src/command/core.rs:889-913); entries that lie in one go to the device straight from there -- no staging copy.  `view` is a writable
    memoryview of the buffer, `ptr` its address."""

    def __init__(self, ctx: "Context", nbytes: int):
        self._ctx, self.nbytes = ctx, nbytes
        p = ctypes.c_void_p()
        L = ctx._L
        L.pna_gpu_host_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        ctx._check(L.pna_gpu_host_alloc(ctx._h, nbytes, ctypes.byref(p)))
        self.ptr = p.value
        self.view = memoryview((ctypes.c_ubyte * nbytes).from_address(self.ptr)).cast("B")

    def free(self):
        if self.ptr:
            self.view.release()
            self._ctx._L.pna_gpu_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            self._ctx._check(self._ctx._L.pna_gpu_host_free(self._ctx._h, ctypes.c_void_p(self.ptr)))
            self.ptr = 0


def create_archive_from_slot(ctx: "Context", names: Sequence[str], slot: HostSlot, offsets: Sequence[int], lengths: Sequence[int], algo: int = ALGO_ZSTD,
                             level: int = LEVEL_DEFAULT) -> bytes:
    """pna_gpu_create_archive_host over entries that lie in a HostSlot (entry i = slot bytes [offsets[i], offsets[i] + lengths[i])): the zero-staging path."""
    L = load_library()
    n = len(lengths)
    out = bytearray()

    def _sink(_u, buf, k):
        out.extend((ctypes.c_char * k).from_address(buf))
        return 0
    cb = SINK_FN(_sink)
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_src = (ctypes.c_void_p * max(n, 1))(*[slot.ptr + o for o in offsets])
    a_len = (ctypes.c_size_t * max(n, 1))(*lengths)
    ctx._check(L.pna_gpu_create_archive_host(ctx._h, algo, level, n, a_names, a_src, a_len, cb, None))
    return bytes(out)


def create_archive(ctx: Optional[Context], names: Sequence[str], entries: Sequence[bytes], algo: int = ALGO_ZSTD,
                   level: int = LEVEL_DEFAULT, solid: bool = False) -> bytes:
    """`pna create` (cli/src/command/create.rs:575-635): entry-parallel compression on the GPU, ordered write."""
    L = load_library()
    n = len(entries)
    out = bytearray()

    def _sink(_u, buf, k):
        out.extend((ctypes.c_char * k).from_address(buf))      # one copy out of the (page-locked) staging buffer
        return 0
    cb = SINK_FN(_sink)
    bufs = [e if isinstance(e, bytes) else bytes(e) for e in entries]      # kept alive; passed by pointer, not copied
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_src = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in bufs])
    a_len = (ctypes.c_size_t * max(n, 1))(*[len(e) for e in entries])
    rc = L.pna_create_archive(ctx._h if ctx is not None else None, algo, level, 1 if solid else 0, n, a_names, a_src, a_len, cb, None)
    if rc:
        msg = L.pna_gpu_last_error(ctx._h).decode() if ctx is not None else ""
        raise PnaGpuError(rc, msg or L.pna_gpu_strerror(rc).decode())
    return bytes(out)


def kdf_pbkdf2_sha256(password: bytes, salt: bytes, rounds: int, key_len: int = 32):
    """hash::pbkdf2_with_salt on the C++ host (include/pna_archive.h): returns (key, PHSF string)."""
    key = ctypes.create_string_buffer(key_len)
    phsf = ctypes.create_string_buffer(256)
    rc = load_library().pna_kdf_pbkdf2_sha256(bytes(password), len(password), bytes(salt), len(salt), rounds, key, key_len, phsf, 256)
    if rc:
        raise PnaGpuError(rc, load_library().pna_gpu_strerror(rc).decode())
    return key.raw, phsf.value.decode()


def create_archive_encrypted(ctx: Context, names: Sequence[str], entries: Sequence[bytes], password: bytes, algo: int = ALGO_ZSTD,
                             level: int = LEVEL_DEFAULT, mode: int = MODE_CTR, rounds: int = 0, cipher: Optional[Cipher] = None) -> bytes:
    """`pna create --aes [ctr|cbc] --pbkdf2`: key derivation on the host, compression + cipher + framing on the device.
    With `cipher` the caller supplies key / PHSF / IVs itself (pna_gpu_create_archive_enc_host)."""
    L = load_library()
    n = len(entries)
    out = bytearray()

    def _sink(_u, buf, k):
        out.extend((ctypes.c_char * k).from_address(buf))
        return 0
    cb = SINK_FN(_sink)
    bufs = [e if isinstance(e, bytes) else bytes(e) for e in entries]
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_src = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in bufs])
    a_len = (ctypes.c_size_t * max(n, 1))(*[len(e) for e in entries])
    if cipher is not None:
        cs = cipher.struct(n)
        rc = L.pna_gpu_create_archive_enc_host(ctx._h, algo, level, n, a_names, a_src, a_len, ctypes.byref(cs), cb, None)
    else:
        rc = L.pna_create_archive_encrypted(ctx._h, algo, level, n, a_names, a_src, a_len, bytes(password), len(password), mode, rounds, cb, None)
    if rc:
        raise PnaGpuError(rc, L.pna_gpu_last_error(ctx._h).decode() or L.pna_gpu_strerror(rc).decode())
    return bytes(out)


def extract_archive(ctx: Context, archive: bytes, password: Optional[bytes] = None):
    """`pna extract` for a non-solid archive (pna_gpu_extract_archive_host): returns [(name, kind, data)] in archive order; chunk CRCs
    are verified (data chunks on the device), entries and solid streams are decrypted (AES CTR / CBC / GCM STREAM) and decoded on the device."""
    out = []

    def _cb(_u, idx, name, kind, data, n):
        out.append((name.decode("utf-8"), kind, ctypes.string_at(data, n) if n else b""))
        return 0
    cb = ENTRY_FN(_cb)
    buf = archive if isinstance(archive, bytes) else bytes(archive)
    rc = ctx._L.pna_gpu_extract_archive_host(ctx._h, buf, len(buf), password, len(password) if password else 0, cb, None)
    ctx._check(rc)
    return out


def kdf_argon2(kind: int, password: bytes, salt: bytes, t_cost: int, m_cost_kib: int, lanes: int, key_len: int = 32) -> bytes:
    """hash::argon2_with_salt on the C++ host (include/pna_archive.h); kind 0 = Argon2d, 1 = Argon2i, 2 = Argon2id."""
    key = ctypes.create_string_buffer(key_len)
    rc = load_library().pna_kdf_argon2(kind, bytes(password), len(password), bytes(salt), len(salt), t_cost, m_cost_kib, lanes, key, key_len)
    if rc:
        raise PnaGpuError(rc, load_library().pna_gpu_strerror(rc).decode())
    return key.raw


def split_archive(archive: bytes, max_part_bytes: int) -> List[bytes]:
    """`pna create --split` (pna_split_archive): one archive image -> the bytes of every part."""
    parts: List[bytearray] = []

    def _sink(_u, idx, buf, n):
        while len(parts) <= idx:
            parts.append(bytearray())
        parts[idx] += ctypes.string_at(buf, n)
        return 0
    cb = PART_SINK_FN(_sink)
    cnt = ctypes.c_uint32()
    rc = load_library().pna_split_archive(bytes(archive), len(archive), max_part_bytes, cb, None, ctypes.byref(cnt))
    if rc:
        raise PnaGpuError(rc, load_library().pna_gpu_strerror(rc).decode())
    assert cnt.value == len(parts)
    return [bytes(p) for p in parts]


def join_parts(parts: Sequence[bytes]) -> bytes:
    """Parts of a multipart archive, in order -> one archive image (pna_join_parts)."""
    out = bytearray()

    def _sink(_u, buf, n):
        out.extend(ctypes.string_at(buf, n))
        return 0
    cb = SINK_FN(_sink)
    n = len(parts)
    keep = [bytes(p) for p in parts]
    a = (ctypes.c_char_p * max(n, 1))(*keep)
    l = (ctypes.c_size_t * max(n, 1))(*[len(p) for p in keep])
    rc = load_library().pna_join_parts(a, l, n, cb, None)
    if rc:
        raise PnaGpuError(rc, load_library().pna_gpu_strerror(rc).decode())
    return bytes(out)


def create_archive_with_metadata(ctx: Context, names: Sequence[str], entries: Sequence[bytes], facets: Optional[Sequence[bytes]] = None,
                                 extra: Optional[Sequence[bytes]] = None, algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT,
                                 cipher: Optional[Cipher] = None) -> bytes:
    """`pna create --keep-timestamp --keep-permission ...` (pna_gpu_create_archive_meta_host): facets[i] / extra[i] are chunks the
    caller has already framed (length | type | data | crc); they land around fSIZ exactly where NormalEntry::write_chunks_to puts them."""
    L = load_library()
    n = len(entries)
    out = bytearray()

    def _sink(_u, buf, k):
        out.extend((ctypes.c_char * k).from_address(buf))
        return 0
    cb = SINK_FN(_sink)
    bufs = [e if isinstance(e, bytes) else bytes(e) for e in entries]
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_src = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in bufs])
    a_len = (ctypes.c_size_t * max(n, 1))(*[len(e) for e in entries])
    ms = MetaStruct()
    keep = []
    for field, blobs in (("extra", extra), ("facets", facets)):
        if blobs is None:
            continue
        bb = [bytes(b) for b in blobs]
        keep.append(bb)
        arr = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) if b else None for b in bb])
        lens = (ctypes.c_size_t * max(n, 1))(*[len(b) for b in bb])
        keep += [arr, lens]
        setattr(ms, field, ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p)))
        setattr(ms, field + "_len", ctypes.cast(lens, ctypes.POINTER(ctypes.c_size_t)))
    cs = cipher.struct(n) if cipher is not None else None
    rc = L.pna_gpu_create_archive_meta_host(ctx._h, algo, level, n, a_names, a_src, a_len, ctypes.byref(cs) if cs is not None else None,
                                            ctypes.byref(ms), cb, None)
    if rc:
        raise PnaGpuError(rc, L.pna_gpu_last_error(ctx._h).decode() or L.pna_gpu_strerror(rc).decode())
    return bytes(out)


def write_file(ctx: Context, sink, name: str, writes: Sequence[bytes], algo: int = ALGO_ZSTD, level: int = LEVEL_DEFAULT,
               meta: bytes = b"", max_chunk_size: int = 0) -> None:
    """Archive::write_file (lib/src/archive/write.rs:276-299 -> write_stream_entry, :730-777): an entry whose data arrives in `writes`;
    the record carries no fSIZ and one FDAT chunk per encoder burst.  `sink(bytes)` receives the entry's chunk bytes; `meta` = already
    framed extra + metadata chunks."""
    L = load_library()

    def _sink(_u, buf, n):
        sink(ctypes.string_at(buf, n))
        return 0
    cb = SINK_FN(_sink)
    w = ctypes.c_void_p()
    ctx._check(L.pna_gpu_stream_entry_begin(ctx._h, algo, level, name.encode(), meta, len(meta), max_chunk_size, cb, None, ctypes.byref(w)))
    for d in writes:
        rc = L.pna_gpu_stream_entry_write(w, bytes(d), len(d))
        if rc:
            L.pna_gpu_stream_entry_abort(w)
            ctx._check(rc)
    ctx._check(L.pna_gpu_stream_entry_finish(w))


def seek_to_end(archive: bytes):
    """Archive::seek_to_end (lib/src/archive/read.rs:412-424): (offset of the AEND chunk, has_next_archive)."""
    off, nxt = ctypes.c_uint64(), ctypes.c_int()
    rc = load_library().pna_archive_seek_to_end(bytes(archive), len(archive), ctypes.byref(off), ctypes.byref(nxt))
    if rc:
        raise PnaGpuError(rc, load_library().pna_gpu_strerror(rc).decode())
    return off.value, bool(nxt.value)


def list_entries(archive: bytes):
    """Top-level records of an archive image: [(name bytes, kind (-1 = solid), offset, length)] (pna_archive_list_entries)."""
    out = []

    def _cb(_u, idx, name, name_len, kind, off, ln):
        out.append((ctypes.string_at(name, name_len) if name_len else b"", kind, off, ln))
        return 0
    cb = RAW_ENTRY_FN(_cb)
    rc = load_library().pna_archive_list_entries(bytes(archive), len(archive), cb, None)
    if rc:
        raise PnaGpuError(rc, load_library().pna_gpu_strerror(rc).decode())
    return out


def append_archive(ctx: Context, archive: bytes, names: Sequence[str], entries: Sequence[bytes], algo: int = ALGO_ZSTD,
                   level: int = LEVEL_DEFAULT) -> bytes:
    """`pna append` (cli/src/command/append.rs:504-560): the archive with the new entries behind the old ones."""
    L = load_library()
    n = len(entries)
    out = bytearray()

    def _sink(_u, buf, k):
        out.extend(ctypes.string_at(buf, k))
        return 0
    cb = SINK_FN(_sink)
    keep = [bytes(e) for e in entries]
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_src = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in keep])
    a_len = (ctypes.c_size_t * max(n, 1))(*[len(b) for b in keep])
    at = ctypes.c_uint64()
    ctx._check(L.pna_gpu_append_archive_host(ctx._h, algo, level, bytes(archive), len(archive), n, a_names, a_src, a_len, ctypes.byref(at), cb, None))
    return bytes(archive[:at.value]) + bytes(out)


def create_archive_multi(ctxs: Sequence[Context], names: Sequence[str], entries: Sequence[bytes], algo: int = ALGO_ZSTD,
                         level: int = LEVEL_DEFAULT) -> bytes:
    """`pna create` from ONE process over several contexts / GPUs (pna_gpu_create_archive_multi_host): contiguous index ranges per
    context, parts written in index order."""
    L = load_library()
    n = len(entries)
    out = bytearray()

    def _sink(_u, buf, k):
        out.extend(ctypes.string_at(buf, k))
        return 0
    cb = SINK_FN(_sink)
    keep = [bytes(e) for e in entries]
    a_ctx = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    a_names = (ctypes.c_char_p * max(n, 1))(*[s.encode() for s in names])
    a_src = (ctypes.c_void_p * max(n, 1))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p) for b in keep])
    a_len = (ctypes.c_size_t * max(n, 1))(*[len(b) for b in keep])
    rc = L.pna_gpu_create_archive_multi_host(a_ctx, len(ctxs), algo, level, n, a_names, a_src, a_len, cb, None)
    if rc:
        raise PnaGpuError(rc, L.pna_gpu_last_error(ctxs[0]._h).decode() or L.pna_gpu_strerror(rc).decode())
    return bytes(out)
