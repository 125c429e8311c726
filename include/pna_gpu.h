/*
 * include/pna_gpu.h -- C ABI of libpna_gpu.so, the MI355X-native replacement for the per-entry compression
 * path of ChanTsune/Portable-Network-Archive (reference paths relative to /root/reference).
 *
 * The reference has no FFI for this path; the seam is the crate-private Rust type
 *     enum CompressionWriter<W: Write> { No, Deflate(ZlibEncoder<W>), ZStd(ZstdEncoder<W>), Xz(..) }
 *     lib/src/compress.rs:21-76, constructed only by compression_writer() lib/src/entry/write.rs:251-265.
 * Each entry point below names the reference interface it replaces; INTEGRATION.md shows the Rust `extern "C"`
 * stub a maintainer would add behind the Compression::{ZStandard,Deflate} arms.
 *
 * Conventions: every function returns 0 (PNA_OK) or a negative PNA_E_* code; no global state; a ctx may be used
 * from one thread at a time (the reference creates one encoder per rayon task, cli/src/command/core.rs:505-517;
 * here one ctx owns one GPU and batches those tasks).  Pointers are host pointers unless the name says `_device`.
 * There is NO CPU fallback: without a usable HIP device pna_gpu_init fails with PNA_E_NODEVICE.
 */
#ifndef PNA_GPU_H
#define PNA_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNA_OK            0
#define PNA_E_NODEVICE   (-1)   /* no HIP device / HIP runtime error at init                        */
#define PNA_E_INVAL      (-2)   /* bad argument (io::ErrorKind::InvalidInput in the reference)       */
#define PNA_E_NOMEM      (-3)   /* device or host allocation failed                                  */
#define PNA_E_DSTSIZE    (-4)   /* destination capacity smaller than pna_gpu_bound()                 */
#define PNA_E_HIP        (-5)   /* a HIP call or kernel failed; pna_gpu_last_error() has the text    */
#define PNA_E_SINK       (-6)   /* the sink callback (== W::write) returned non-zero                 */
#define PNA_E_UNSUPPORTED (-7)  /* algorithm not offered by this build                               */

/* == Compression::to_byte(), lib/src/entry/options.rs:241-247 */
#define PNA_ALGO_STORE    0
#define PNA_ALGO_DEFLATE  1
#define PNA_ALGO_ZSTD     2

/* encoder feature bits (pna_gpu_init flags, low byte); default = all of them */
#define PNA_F_HUF   1u
#define PNA_F_FSE   2u
#define PNA_F_LAZY  4u
#define PNA_F_REP   8u
#define PNA_F_FAR   0x10u       /* zstd: look-back over the whole 1 MiB segment (candidates beyond the LDS window verified in HBM / L2) */
#define PNA_F_ADOPT 0x20u       /* backward adoption: a match found late is moved back to its true start */
#define PNA_F_INS2  0x40u       /* only even positions enter the hash table (with PNA_F_ADOPT) */
#define PNA_F_STRONG 0x80u      /* third adoption round (up to 7 positions back) + two-step lazy deferral: the set of the high levels */
#define PNA_F_LZ_WAVEPARSE 0x4000u  /* testing: the split LZ stage with the wave-per-region parse kernel as its second half */
#define PNA_F_LZ_FUSED 0x8000u     /* the LZ stage as ONE kernel (match + parse in k_lz) instead of the default split form (match kernel ->
                                     * one word per input byte in a workspace of up to 16 GiB -> parse kernel): same bytes, ~20 % slower,
                                     * no workspace; the library falls back to it by itself when the workspace cannot be allocated */
#define PNA_F_DEFAULT 0x80000000u   /* let the library choose */

typedef struct pna_gpu_ctx pna_gpu_ctx;
/* == W::write of the reference's sink: non-zero return aborts with PNA_E_SINK */
typedef int (*pna_sink_fn)(void *user, const void *buf, size_t len);

/* One context per (process, GPU).  Replaces the per-entry encoder construction
 * ZstdEncoder::new(writer, level) / ZlibEncoder::new(writer, level) (lib/src/entry/write.rs:257-262). */
int  pna_gpu_init(pna_gpu_ctx **out, int device_id, uint32_t flags);
void pna_gpu_shutdown(pna_gpu_ctx *ctx);
const char *pna_gpu_last_error(const pna_gpu_ctx *ctx);
/* Tuning knobs of a context.  Each starts from an environment variable that pna_gpu_init reads ONCE (named in brackets); no entry point
 * consults the environment afterwards.  PNA_E_INVAL for unknown names and values out of range.
 *   "blk_log" [PNA_BLK_LOG]               block size of every batch = 1 << value (13..17); 0 = chosen by batch size (default)
 *   "unit_log" [PNA_LZ_UNIT_LOG]          LZ-stage units of 1 << value bytes (block size..20); 0 = chosen by batch size (default)
 *   "latency_max_mib" [PNA_LATENCY_MAX_MIB]  batches of at most this many MiB of input run in LATENCY MODE (default 128; 0 = never): smaller
 *                                         blocks inside the same frames and one LZ workgroup per unit instead of per segment, so that a handful
 *                                         of entries -- the CompressionWriter seam under the reference's thread pool -- fills the chip.  Same
 *                                         format, same decoders; ratio -0.1 .. -0.3 % (block headers).  pna_gpu_last_timing reports what was chosen.
 *                                         zstd calls beyond the mode still take smaller blocks by their input (32 KiB up to 384 MiB, 64 KiB up to 1 GiB, 128 KiB
 *                                         beyond: a batch's time below ~1 GiB is its blocks' chains); 0 = never AND 128 KiB blocks whatever the batch.
 *   "hist_by_block" [PNA_HIST_BY_BLOCK]   zstd entropy stage with statistics per block and three-lane state chains (1) or per segment and the one-kernel
 *                                         sequence coder (0); -1 (default): the first up to 40 960 blocks per sub-batch.  Same bytes either way.
 *   "lz_split" [PNA_LZ_SPLIT]             0 one-kernel LZ stage, 1 split form for long runs (default), 2 split form with the wave-per-region parse
 *   "lz_split_blocks" [PNA_LZ_SPLIT_BLOCKS]  blocks per run of the split form (default 131 072 = 16 GiB of input, 48 GiB of workspace; halved while the workspace cannot be had)
 *   "lz_split_min" [PNA_LZ_SPLIT_MIN]     shortest run, in segments, that takes the split form (default 0: every run)
 *   "lz_pbuf_fail" [PNA_LZ_PBUF_FAIL]     testing: behave as if the split form's workspace could not be allocated
 *   "win32k" [PNA_WIN32K]                 LDS geometry of the zstd match finder: 1 (default) the light and default level sets (zstd 2, 3) on a 32 KiB window, the high set (4 .. 9) on
 *                                         a 16 KiB window; 0: both on 64 KiB; 2: both on 16 KiB.  Other bytes, same format
 *   "tab3" [PNA_TAB3]                     the hash table of the zstd light / default / high sets: 1 (default) PACKED, three 21-bit entries (even position + 2-bit tag) per
 *                                         64-bit LDS word -- 49 062 slots next to the 32 KiB window, 55 206 next to the 16 KiB one; 0: one 32-bit entry per slot
 *                                         (32 704 / 36 800: round 3's table).  Other bytes (ratio 2.847 against 2.759 on text at the default level), same format
 *   "seq_hist" [PNA_SEQ_HIST]             1 (default): large zstd batches -- the parse kernel of the split LZ stage counts every block's sequence codes, the statistics kernel walks the literals only; 0: it
 *                                         reads the sequences once more.  Same bytes
 *   "far1" [PNA_FAR1]                     1 (default): the zstd light / default sets (levels 2, 3: packed table, 32 KiB window) verify at most 63 candidates beyond the match kernel's LDS window
 *                                         per wave of 256 positions -- one compacted round of far candidates -- and drop the rest (round 5: - 0.16 % of ratio, - 8.5 % of the match kernel);
 *                                         0: every far candidate, in as many rounds as it takes.  Other bytes, same format
 *   "strong2" [PNA_STRONG2]               1 (default): zstd levels 4 .. 22 (the high and max sets) on their standard geometries run a fourth backward-adoption round over eight positions and count up to
 *                                         15 back bytes (round 5: the high set 2.860 -> 2.883 on text, 10 - 22 2.977 -> 3.002); 0: the default level's three rounds.  Other bytes, same format
 *   "small_geometry" [PNA_SMALL_GEOMETRY] 1 (default): segments of at most 16 KiB -- small entries, the tail of longer ones -- are matched by one wave each with look-ups and inserts
 *                                         alternating per 256 positions (k_lzms) instead of by a workgroup per 4 096: such a segment's first tile finds nothing in the large
 *                                         geometry (4 KiB text entries: ratio 1.77 -> 2.04, libzstd -3: 2.05).  0: the large geometry for every segment.  Other bytes, same format
 *   "single_frame" [PNA_SINGLE_FRAME]     zstd: 1 = an entry's payload is ONE frame (one frame header, the 1 MiB segments' blocks behind each other, matches never
 *                                         cross a segment start) as the reference's encoder writes (lib/src/compress/zstandard.rs: one Encoder per entry);
 *                                         0 (default) = a frame per 1 MiB segment, which this library's decoder takes in parallel.  3 + 0..2 bytes per segment apart
 *   "zexec_par_min_mib" [PNA_ZEXEC_PAR_MIN_MIB]  decoder: single zstd frames of at least this many MiB (default 8; any size) take the header walk +
 *                                         wave-per-block parse + pointer-jumping execution; 0 = never
 *   "zdec_fallback_max_mib" [PNA_ZDEC_FALLBACK_MAX_MIB]  decoder: a zstd frame of more content than this that the parallel paths cannot take is refused (PNA_E_UNSUPPORTED)
 *                                         instead of decoded by one workgroup at ~11 MiB/s; 0 (default): no limit
 *   "zexec_win_mib" [PNA_ZEXEC_WIN_MIB]   decoder: the pointer-jumping execution runs a frame / stream in windows of whole blocks of at most this many MiB of output,
 *                                         one after the other (default and maximum 1 024: a word counts 31 bits from its window's start; scratch = 4 bytes per byte of a window)
 *   "stream_batch_mib" [PNA_STREAM_BATCH_MIB] (256), "stream_overlap_mib" [PNA_STREAM_OVERLAP_MIB] (24), "stream_gather_wgs" [PNA_STREAM_GATHER_WGS] (48): the streaming
 *                                         facade's pipeline -- largest device batch, how much may queue before a second batch is cut while one is on the device,
 *                                         workgroups of the copy-in kernel
 *   "lazy2" [PNA_LAZY2]                   how far a start looks ahead before it is taken, beyond the next position: 2 (default), 1, 0
 *   "strong_gtab" [PNA_STRONG_GTAB]       zstd levels 10..22 with the hash table in global memory (1, default) or in LDS (0)
 *   "lit_beside_seq" [PNA_LIT_BESIDE_SEQ] large zstd batches: the literal coder on a second stream next to the sequence coder (1, default)
 *   "pipeline_chunks" [PNA_PIPELINE_CHUNKS], "max_chunk_size" [PNA_MAX_CHUNK_SIZE] (FDAT chunk size of the entry points without such a parameter), "sub_mib" [PNA_SUB_MIB], "stage_threads" [PNA_STAGE_THREADS],
 *   "extract_win_mib" [PNA_EXTRACT_WIN_MIB], "batch_piece_mib" [PNA_BATCH_PIECE_MIB], "inflate_serial" [PNA_INFLATE_SERIAL],
 *   "zdec_serial" [PNA_ZDEC_SERIAL], "zdec_dbg" [PNA_ZDEC_DBG] (diagnostics of the one-workgroup zstd decoder: bit 8 = small re-base distances, for its test), "stream_pool_mib" [PNA_STREAM_POOL_MIB], "stream_linger_us" [PNA_STREAM_LINGER_US] (-1 = adaptive): DESIGN.md. */
int  pna_gpu_set_option(pna_gpu_ctx *ctx, const char *name, long value);
const char *pna_gpu_strerror(int code);

/* Worst-case compressed size of one entry (zstd: ZSTD_compressBound-like; zlib: deflateBound-like). */
size_t pna_gpu_bound(int algo, size_t src_len);

/* Level mapping of the reference, restated so callers can pass CompressionLevel values through unchanged:
 * lib/src/compress/zstandard.rs:43-57 (Default -> 3, clamp to min..max) and lib/src/compress/deflate.rs:89-101
 * (Default -> 6, clamp 0..9).  `level` < 0 with level == PNA_LEVEL_DEFAULT means default. */
/* The encoder has these parameter sets behind that scale ("Level sets"):
 *   stored   deflate 0 = Compression::none(): stored blocks only (zlib header 78 01)
 *   fast     zstd < 0 and 1, deflate 1..3: every position in the table, lazy deferral, look-back = the LDS window
 *   light    zstd 2: even-position PACKED table next to a 32 KiB window, backward adoption (two rounds), 1 MiB look-back, lazy deferral over three positions
 *   default  zstd 0 and 3..5: light + a third adoption round (up to 7 positions back) and two-step lazy deferral;  deflate 4..8: even-position table,
 *            adoption, lazy deferral over three positions
 *   high     zstd 6..9: default on a 16 KiB window with 55 206 slots;  deflate 9: + the third adoption round and two-step lazy deferral
 *   max      zstd 10..22: + the hash table in global memory, 2^19 slots per segment */
#define PNA_LEVEL_DEFAULT  (-1000)
int  pna_gpu_clamp_level(int algo, int level);

/* ---- batch of independent entries: the data-parallel fan-out of spawn_entry_results()/create_entry()
 * (cli/src/command/core.rs:496-537,915-977).  Entry i becomes one independent stream in dst[i]
 * (zstd: concatenated frames; deflate: one zlib stream), exactly what FileEntryBuilder::build() hands to
 * NormalEntry as FDAT payload (lib/src/entry/builder/file.rs:131-135). */
int  pna_gpu_compress_batch(pna_gpu_ctx *ctx, int algo, int level, size_t n,
                            const void *const *src, const size_t *src_len,
                            void *const *dst, const size_t *dst_cap, size_t *dst_len);

/* Same, with the inputs already resident in HBM (bench / pipelined callers).
 *   d_src      device buffer holding all entries; entry i = [src_off[i], src_off[i] + src_len[i]);
 *              every src_off[i] must be a multiple of 16 and the buffer must extend 4 KiB past the last entry.
 *   d_dst      device buffer of dst_cap bytes receiving the compressed entries back to back.
 *   dst_off    host array of n+1 offsets (out): entry i's stream = d_dst[dst_off[i] .. dst_off[i+1]).
 * Work is enqueued on `hip_stream` (a hipStream_t, or NULL for the ctx's own stream) and the call returns after
 * the stream has finished (dst_off needs the sizes). */
int  pna_gpu_compress_batch_device(pna_gpu_ctx *ctx, int algo, int level, size_t n,
                                   const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                   void *d_dst, size_t dst_cap, uint64_t *dst_off, void *hip_stream);

/* ---- non-solid archive assembled in HBM: what create_archive_file() (cli/src/command/create.rs:575-635) writes through
 * Archive::write_header / add_entry / finalize (lib/src/archive/write.rs:92-101,368-370,438-441) for file entries whose
 * records are FHED | fSIZ | FDAT | FEND (NormalEntry::write_chunks_to, lib/src/entry.rs:895-911; write_chunk and its
 * CRC-32, lib/src/io.rs:183-197, lib/src/format/chunk.rs:7-12).  The compressed payloads are written once, straight at
 * their archive offsets; the FDAT CRCs are computed on the device.  Same inputs as pna_gpu_compress_batch_device plus
 * names[i] (host strings, sanitised like EntryName::sanitize).  d_dst: 16-byte aligned device buffer of at least
 * pna_gpu_archive_bound() bytes; on return d_dst[0 .. *archive_len) is the complete .pna file and, if entry_off is not
 * NULL, entry_off[i] (n + 1 values) is the offset of entry i's FHED chunk.  Byte-identical to pna_create_archive()
 * (include/pna_archive.h) over the same entries. */
size_t pna_gpu_archive_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len);
/* One shard of an archive whose entries are split over several producers (one rank per GPU): only the shard with
 * PNA_PART_HEAD carries the signature + AHED, only the one with PNA_PART_TAIL the AEND; the shards' outputs concatenated
 * in entry order are the archive (the ordered drain of drain_entry_results, cli/src/command/core.rs:471-493). */
#define PNA_PART_HEAD 1u
#define PNA_PART_TAIL 2u
int  pna_gpu_create_archive_part_device(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                        const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                        void *d_dst, size_t dst_cap, uint64_t *entry_off, uint64_t *archive_len,
                                        uint32_t part_flags, void *hip_stream);
int  pna_gpu_create_archive_device(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                   const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                   void *d_dst, size_t dst_cap, uint64_t *entry_off, uint64_t *archive_len,
                                   void *hip_stream);

/* ---- cipher stage: the reference stacks compress -> cipher -> sink (get_writer, lib/src/entry/write.rs:268-274); the cipher
 * writers CipherWriter::{CtrAes, CbcAes} (encryption_writer, lib/src/entry/write.rs:189-248: Ctr128BE<Aes256>, CBC + PKCS#7) run
 * here as kernels over the compressed payload where it already sits in the archive buffer, before the chunk CRC-32.
 * The key is the password hash the Rust host derives once per WriteOptions (derive_key_material, lib/src/entry/write.rs:64-72;
 * pna_kdf_pbkdf2_sha256 in pna_archive.h is the C++ host's equivalent); `phsf` is the PHC string that goes into each entry's PHSF
 * chunk.  Every entry gets its own IV (random::random_vec in to_hashed, lib/src/entry/write.rs:108-121): ivs[16 * i ..) for
 * entry i, or NULL to draw them from the OS.  Entry record: FHED(encryption, cipher_mode) | fSIZ | PHSF | FDAT(iv) |
 * FDAT(ciphertext) | FEND (lib/src/entry.rs:895-911; the IV is its own data piece: prepend_data_prefix, lib/src/entry/builder.rs:62-69). */
#define PNA_ENC_NONE      0     /* == Encryption::to_byte(), lib/src/entry/options.rs */
#define PNA_ENC_AES       1
#define PNA_ENC_CAMELLIA  2     /* not offered: PNA_E_UNSUPPORTED */
#define PNA_MODE_CBC      0     /* == CipherMode::to_byte() */
#define PNA_MODE_CTR      1
#define PNA_MODE_GCM      2     /* "GCM STREAM": lib/src/cipher/aead.rs, lib/src/cipher/gcm.rs */
typedef struct {
    int encryption;             /* PNA_ENC_* */
    int cipher_mode;            /* PNA_MODE_* */
    uint8_t key[32];            /* AES-256 key; GCM: K_master, the per-entry stream keys are derived from it (HKDF) */
    const char *phsf;           /* body of the PHSF chunk */
    const uint8_t *ivs;         /* CBC / CTR: n x 16 bytes; GCM: n x 39 bytes (salt[32] || nonce_prefix[7] per entry); or NULL */
    uint32_t gcm_segment_size;  /* GCM: segment size written into the stream header (0 = 1 MiB, the reference's DEFAULT_SEGMENT_SIZE; at most 64 MiB) */
} pna_gpu_cipher;
/* PNA_MODE_GCM (archive entry points, non-solid): entry record FHED(enc, 2) | fSIZ | PHSF | FDAT(stream header, 75 bytes) |
 * FDAT(ciphertext || tag) | FEND.  The stream key is HKDF(K_master, salt, entry context) with the context bound to the entry's FHED
 * chunk and the PHSF string (derive_stream_key, aead.rs:184-199).  The payload is cut into segments of gcm_segment_size bytes, each
 * followed by its 16-byte tag (GcmEncryptWriter, lib/src/cipher/gcm.rs:48-100): nonce = prefix || counter || flag, flag 1 on the last
 * segment (an empty payload is one empty final segment).  An entry whose payload spans several FDAT chunks (beyond max_chunk_size) is
 * PNA_E_UNSUPPORTED with GCM. */
size_t pna_gpu_archive_enc_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len, const pna_gpu_cipher *cipher);
/* pna_gpu_create_archive_part_device with a cipher (cipher == NULL or encryption == PNA_ENC_NONE: identical to it). */
int  pna_gpu_create_archive_enc_device(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                       const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                       const pna_gpu_cipher *cipher, void *d_dst, size_t dst_cap, uint64_t *entry_off,
                                       uint64_t *archive_len, uint32_t part_flags, void *hip_stream);
/* Per-entry metadata: chunks the host has ALREADY framed (length | type | data | CRC-32) and that NormalEntry::write_chunks_to places
 * around fSIZ (lib/src/entry.rs:895-911): `extra[i]` between FHED and fSIZ (user-defined chunks), `facets[i]` between fSIZ and PHSF / FDAT
 * (cTIM mTIM aTIM fPRM fUId fGId fONm fGNm xATR ...: try_for_each_metadata_facet, lib/src/entry.rs:124-180) -- what `pna create
 * --keep-timestamp --keep-permission --keep-xattr` adds.  NULL arrays or zero lengths mean none.  Every blob is checked (chunk lengths,
 * CRCs, none of the chunk types this library writes itself): PNA_E_INVAL otherwise.  The caller's bound grows by the blobs' lengths. */
typedef struct {
    const void *const *extra;  const size_t *extra_len;
    const void *const *facets; const size_t *facets_len;
} pna_gpu_entry_meta;
int  pna_gpu_create_archive_meta_device(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                        const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                        const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, void *d_dst, size_t dst_cap,
                                        uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream);
int  pna_gpu_create_archive_meta_host(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                      const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                      const pna_gpu_entry_meta *meta, pna_sink_fn sink, void *user);
/* FlattenWriter::max_chunk_size (lib/src/util/io.rs:60-77; FileEntryBuilder::max_chunk_size, lib/src/entry/builder/file.rs:105-112; default u32::MAX,
 * lib/src/chunk.rs:28): an entry's stream -- after the cipher -- is cut into FDAT chunks of exactly max_chunk_size bytes, the last one holding the rest
 * (an IV / stream header stays the data piece of its own that prepend_data_prefix makes it).  The general forms of the archive entry points take it as a
 * parameter (0 = u32::MAX); the entry points above and below without one use the context's option "max_chunk_size" (pna_gpu_set_option, default 0).
 * Byte-identical to the reference's chunking up to chunks of 2^32 - 5 bytes (the device CRC takes type + data as one message below 2^32 bytes: an entry
 * whose compressed payload exceeds 4 GiB - 5 bytes is cut 4 bytes earlier than the reference would).  CBC and GCM entries must fit one chunk
 * (PNA_E_UNSUPPORTED otherwise; CTR -- the CLI's default -- has no such limit).  Bound: pna_gpu_archive_chunked_bound (+ the meta blobs' lengths). */
size_t pna_gpu_archive_chunked_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len, const pna_gpu_cipher *cipher, uint32_t max_chunk_size);
int  pna_gpu_create_archive_chunked_device(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                           const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                           const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, uint32_t max_chunk_size, void *d_dst, size_t dst_cap,
                                           uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream);
int  pna_gpu_create_archive_chunked_host(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                         const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                         const pna_gpu_entry_meta *meta, uint32_t max_chunk_size, uint32_t part_flags, pna_sink_fn sink, void *user);
/* pna_gpu_create_archive_host (bounded in-flight window from host memory) with the cipher stage. */
int  pna_gpu_create_archive_enc_host(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                     const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                     pna_sink_fn sink, void *user);
/* The cipher alone, in place, over n byte ranges of a device buffer: range i = d_buf[off[i] .. off[i] + len[i]) is one cipher
 * stream with IV ivs[16 * i ..).  CTR: encrypt == decrypt (DecryptReader::CtrAes, lib/src/entry/read.rs:83-88).  CBC: encryption
 * only (decrypt != 0 is PNA_E_UNSUPPORTED); the ciphertext is (len / 16 + 1) * 16 bytes long, the caller leaves that room. */
int  pna_gpu_cipher_apply_device(pna_gpu_ctx *ctx, const pna_gpu_cipher *cipher, int decrypt, size_t n, void *d_buf,
                                 const uint64_t *off, const uint64_t *len, void *hip_stream);

/* `pna create --solid` assembled in HBM: the inner entries are serialised as STORE records (FHED | fSIZ | FDAT | FEND, their
 * CRC-32 on the device) into one stream, the stream is compressed as ONE entry and every segment's output becomes one SDAT
 * chunk between SHED and SEND (create_archive_file's solid branch, cli/src/command/create.rs:594-598,603-617;
 * lib/src/archive/write.rs:443-470,575-580,716-727; SolidHeader::to_bytes, lib/src/entry/header.rs:274-282).
 * Arguments as pna_gpu_create_archive_device; inner entries >= 2 GiB are rejected (one FDAT chunk each). */
size_t pna_gpu_solid_archive_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len);
int  pna_gpu_create_solid_archive_device(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                         const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                         void *d_dst, size_t dst_cap, uint64_t *archive_len, void *hip_stream);

/* With a cipher: one cipher stream over all SDAT bodies (into_solid_archive takes any cipher, lib/src/archive/write.rs:443-470).
 *   CTR: SHED(encryption, cipher_mode) | PHSF | SDAT(iv) | SDAT(ciphertext)* | SEND; cipher->ivs is ONE 16-byte IV or NULL.
 *   GCM: SHED | PHSF | SDAT(stream header, 75 bytes) | SDAT(segment ciphertext || tag)* | SEND -- the GCM STREAM layout of
 *        pna_gpu_create_archive_enc_device over the compressed solid stream, the stream key bound to the SHED chunk (entry_context,
 *        lib/src/cipher/aead.rs:167-190); cipher->ivs is ONE salt(32) || nonce_prefix(7) or NULL.
 *   CBC encryption would be one serial chain over the whole stream: PNA_E_UNSUPPORTED (the extract driver decrypts CBC solid streams,
 *   which is parallel).
 * pna_gpu_solid_archive_enc_bound: the destination capacity to offer (cipher == NULL: the plain bound). */
size_t pna_gpu_solid_archive_enc_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len, const pna_gpu_cipher *cipher);
int  pna_gpu_create_solid_archive_enc_device(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                             const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                             const pna_gpu_cipher *cipher, void *d_dst, size_t dst_cap, uint64_t *archive_len,
                                             void *hip_stream);

/* The same from host memory.  zstd: STREAMING, as SolidArchive::add_entry feeds its one encoder (lib/src/archive/write.rs:575-580): the serialised inner
 * entries reach the device in windows of `solid_win_mib` MiB (256) through two page-locked slots each way -- about four windows of page-locked memory
 * whatever the archive's size --, the sink receives the head, one piece per window, the tail; inner entries of ANY size (FDAT chunks of at most
 * 2^32 - 5 bytes, FlattenWriter's cut, lib/src/util/io.rs:60-77).  The bytes equal pna_gpu_create_solid_archive_device's.  deflate (one zlib stream
 * with one Adler-32) and option single_frame: the whole stream in flight at once, inner entries below 2 GiB. */
int  pna_gpu_create_solid_archive_host(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                       const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user);

/* Same archive from host memory with a bounded in-flight window: entries stream through two page-locked staging slots
 * (<= ~1 GiB of input each); staging of sub-batch k+1, the H2D copy, the kernels of sub-batch k and the D2H copy of
 * sub-batch k-1 overlap.  The sink receives the signature + AHED, then one piece per sub-batch, then AEND.  Replaces the
 * reference's fan-out that keeps every compressed entry in RAM until the rayon scope ends
 * (cli/src/command/core.rs:496-537, cli/src/command/create.rs:575-635).  pna_create_archive() uses it for non-solid
 * zstd / deflate archives. */
int  pna_gpu_create_archive_host(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                 const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user);

/* ---- read side (extract / verify): replaces decompress_reader() -> zstd::stream::read::Decoder / flate2::read::ZlibDecoder
 * (lib/src/entry/read.rs:171-190; callers cli/src/command/extract.rs:594-640, verify.rs:140-188).  Entry i's payload (the
 * concatenated FDAT bodies) is decoded to raw_len[i] bytes (the entry's fSIZ).
 *   PNA_ALGO_ZSTD: one or more RFC 8878 frames without dictionary, offsets up to 2^28 - 4 (windows up to 128 MiB: every libzstd level without --long); a multi-frame payload must follow this library's
 *     segmentation (every frame but the last holds 1 MiB) because frames carry no content size -- single-frame payloads, which
 *     is what the reference writes, always work.
 *   PNA_ALGO_DEFLATE: one RFC 1950 zlib stream (any block types, sync-flush markers, window <= 32 KiB); Adler-32 is verified.
 *     Streams of 4 GiB and more (compressed or decoded) are decoded by their sync-flush delimited pieces -- what this library
 *     writes --, or, a foreign encoder's stream without such markers (one flate2 / zlib stream per entry: what the reference writes), in chunks between
 *     block starts found by trial, as long as its COMPRESSED bytes stay below 4 GiB and it has dynamic blocks to find (up to ~5.9 GiB of content);
 *     what fits neither is PNA_E_UNSUPPORTED (the wave-per-stream walk counts in 32 bits).
 *   Single streams of any size are executed in parallel (zstd frames of 2 GiB and more since the second half of round 4: the executor's windows,
 *     option "zexec_win_mib"); a zstd frame whose compressed bytes exceed 4 GiB, or which does not fit its pooled resources, is left to one workgroup
 *     (~11 MiB/s) unless the option "zdec_fallback_max_mib" refuses it.
 * Errors: PNA_E_INVAL for corrupt / mismatching streams (pna_gpu_last_error names the entry), PNA_E_UNSUPPORTED for
 * dictionaries and other algorithms. */
int  pna_gpu_decompress_batch(pna_gpu_ctx *ctx, int algo, size_t n, const void *const *src, const size_t *src_len,
                              void *const *dst, const size_t *raw_len);
int  pna_gpu_decompress_batch_device(pna_gpu_ctx *ctx, int algo, size_t n, const void *d_src, const uint64_t *src_off,
                                     const uint64_t *src_len, void *d_dst, const uint64_t *dst_off, const uint64_t *raw_len,
                                     void *hip_stream);

/* A zstd stream whose decoded size is recorded nowhere (the SDAT stream of a solid entry).  Step 1 counts its frames; the caller
 * provides frames x 1 MiB of room (this library's segmentation) -- for ONE frame, as the reference writes it, any capacity it sees fit;
 * step 2 decodes and reports the size found (PNA_E_INVAL when the stream does not fit). */
int  pna_gpu_zstd_stream_frames_device(pna_gpu_ctx *ctx, const void *d_src, uint64_t src_off, uint64_t src_len, uint32_t *n_frames, void *hip_stream);
int  pna_gpu_zstd_decompress_open_device(pna_gpu_ctx *ctx, const void *d_src, uint64_t src_off, uint64_t src_len, void *d_dst, uint64_t dst_off,
                                         uint64_t dst_cap, uint64_t *raw_len, void *hip_stream);

/* The same for one zlib stream (deflate entries without fSIZ, deflate solid streams). */
int  pna_gpu_inflate_open_device(pna_gpu_ctx *ctx, const void *d_src, uint64_t src_off, uint64_t src_len, void *d_dst, uint64_t dst_off,
                                 uint64_t dst_cap, uint64_t *raw_len, void *hip_stream);

/* Read-side driver for archives in host memory (normal and solid entries): `pna extract` / `pna verify` (cli/src/command/extract.rs:594-640,
 * verify.rs:140-188; Archive::read_header + entries, lib/src/archive/read.rs:22-66; read_chunk's mandatory CRC check, lib/src/io.rs:117-149;
 * decrypt_reader / decompress_reader, lib/src/entry/read.rs:59-104,171-190).  The chunk walk and the small chunks' CRCs are host work;
 * the FDAT CRC-32s, the gather of every entry's data pieces, AES decryption -- CTR, CBC (PKCS#7 checked), GCM STREAM (key confirmation,
 * every segment tag verified) -- with the key derived from the PHSF string ("$argon2{d,i,id}$..." or "$pbkdf2-sha256$...") and `password`,
 * and zstd / deflate / store decoding run on the device; entries without fSIZ are sized by the decoder.  cb is called once per entry in archive order (kind =
 * DataKind::to_byte(): 0 file, 1 directory, ...); `data` is valid during the call.  PNA_E_INVAL: structural damage, CRC mismatch, corrupt
 * stream, wrong password (GCM key confirmation, CBC padding), authentication failure; PNA_E_UNSUPPORTED: multipart archives, xz,
 * Camellia, solid streams with inner entries that are not stored.  Solid entries (SHED [PHSF] SDAT* SEND; plain or AES CTR / CBC / GCM):
 * SDAT CRCs and the inner FDAT CRCs on the device, the stream is decoded without a recorded size (frames counted first). */
/* `name` is the entry's PATH as the reference's reader exposes it (EntryHeader::path(), lib/src/entry/header.rs:91-94): the FHED name
 * normalised and reduced to its normal components (EntryName::sanitize, lib/src/entry/name.rs:148-156 -- no root, no "." / ".."), so a
 * callback that writes files below an output directory cannot be led outside it by a crafted archive.  FHED names that are not valid
 * UTF-8 (InvalidData in the reference, header.rs:143-146) or that contain a NUL byte (not representable here) fail with PNA_E_INVAL. */
typedef int (*pna_entry_fn)(void *user, size_t index, const char *name, int kind, const void *data, size_t len);
int  pna_gpu_extract_archive_host(pna_gpu_ctx *ctx, const void *archive, size_t archive_len, const void *password, size_t password_len,
                                  pna_entry_fn cb, void *user);

/* pna_gpu_create_archive_host for ONE PART of an archive (PNA_PART_HEAD: signature + AHED first, PNA_PART_TAIL: AEND last): what
 * `pna append` writes behind the existing entries (PNA_PART_TAIL only) and `pna update` for the entries it re-creates (neither flag). */
int  pna_gpu_create_archive_part_host(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                      const void *const *src, const size_t *src_len, uint32_t part_flags, pna_sink_fn sink, void *user);
/* One process driving several GPUs (SURVEY 8(b) `device_ids, n_devices`): n_ctx contexts (pna_gpu_init per device; they may share a
 * device), the entries cut into contiguous index ranges balanced by bytes, one range per context on a thread of its own through the bounded
 * host pipeline, the parts handed to the sink in index order -- the reference's fan-out + ordered drain (cli/src/command/core.rs:496-537,
 * 471-493) with devices in place of worker threads; no device-to-device traffic.  The archive equals pna_gpu_create_archive_host's (every range runs with
 * the block size of the whole call's input bytes). */
int  pna_gpu_create_archive_multi_host(pna_gpu_ctx *const *ctxs, size_t n_ctx, int algo, int level, size_t n, const char *const *names,
                                       const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user);
/* Zero-staging input.  The reference reads every file into memory of its own (fs::read, cli/src/command/core.rs:889-913 write_from_path) before the
 * encoder sees it; a device needs the bytes in PAGE-LOCKED memory to copy them at the link's rate.  pna_gpu_host_alloc hands the host such a buffer to
 * read its files into (read_exact into the slot instead of fs::read into a Vec): the host-memory create entry points (pna_gpu_create_archive_host and its
 * _enc / _meta / _chunked / _part forms, pna_gpu_append_archive_host) send a batch whose entries all lie in buffers of this call to the device straight
 * from there -- no pageable -> page-locked copy, one host thread instead of eight --; runs of entries that are contiguous at a 16-byte stride travel as
 * one copy.  A batch with an entry elsewhere is staged as before.  A buffer must stay untouched until the create call that reads it has returned;
 * pna_gpu_host_free gives it back (pna_gpu_destroy frees what is left). */
int  pna_gpu_host_alloc(pna_gpu_ctx *ctx, size_t bytes, void **out);
int  pna_gpu_host_free(pna_gpu_ctx *ctx, void *buf);

/* `pna append` (cli/src/command/append.rs:504-560 run_append_archive: open_archive_then_seek_to_end, add the new entries in order,
 * finalize): `archive` is the existing image (or its last part); *write_at receives the offset of its AEND chunk, and the sink receives
 * the bytes that belong there -- the n new entries, compressed on the device, then AEND.  The result, archive[0 .. write_at) followed by
 * the sink's bytes, is the archive `pna create` writes from all entries at once -- the same chunks in the same order; byte for byte where both calls
 * choose the same block size (option `latency_max_mib` = 0 pins it: by default the block size of a call follows the CALL's input bytes, section "levels",
 * so the old entries' payloads are those of the call that made them). */
int  pna_gpu_append_archive_host(pna_gpu_ctx *ctx, int algo, int level, const void *archive, size_t archive_len, size_t n,
                                 const char *const *names, const void *const *src, const size_t *src_len, uint64_t *write_at,
                                 pna_sink_fn sink, void *user);

/* ---- several GPUs, one process per GPU: rank r compresses the contiguous index range r of the entries into an archive PART in its HBM
 * (pna_gpu_create_archive_part_device: PNA_PART_HEAD on the first rank, PNA_PART_TAIL on the last), and the parts in rank order are the archive -- the
 * fan-out and ordered drain of cli/src/command/core.rs:496-537,471-493 with GPUs in place of worker threads.  The one exchange step is the ordered
 * gather: an all-gather of the parts' sizes (8 bytes per rank), then every rank sends its part to `root` (ncclSend / ncclRecv in one group: a link per
 * peer over xGMI), which receives them at the prefix sums of the sizes.  RCCL is loaded with dlopen at the first call (PNA_E_UNSUPPORTED when the
 * host has none).  Bootstrap as with NCCL: rank 0 calls pna_gpu_comm_unique_id and hands the 128 bytes to the other ranks by whatever channel the
 * host has (the reference has none: it is a single process), every rank calls pna_gpu_comm_init.  All calls are collective.
 *   sizes  (host, nranks values, may be NULL) receives every part's length on every rank; *total their sum;
 *   d_out / out_cap matter on `root` only.  The root's capacity travels with the sizes, so the verdict is COLLECTIVE: when the parts do not fit, every
 *   rank returns PNA_E_DSTSIZE and no rank has sent anything (the communicator stays usable; sizes / total are filled in, so the host can retry with
 *   a larger destination).
 * pna_gpu_gather_ordered_start posts the gather and returns: what it waits for is the 16-bytes-per-rank size exchange on the communicator's own
 * stream -- which runs BEHIND the transfers of the gathers posted before it (one stream per communicator: a second start therefore blocks until the first
 * gather is done; the overlap that matters, a gather beside the NEXT piece's compression, comes from calling start after that compression was launched) --;
 * the transfers are ordered behind the work already queued on `hip_stream` (the producer of d_local) and run on the communicator's stream, so
 * they overlap whatever the host launches next (the next piece's compression into another buffer).  pna_gpu_gather_wait blocks until the posted
 * gathers are done; d_local and d_out belong to the gather until then.  pna_gpu_gather_ticket = how many gathers the communicator has posted (the
 * latest one's ticket), pna_gpu_gather_wait_for(ticket) waits for the gathers up to that one only -- the double-buffering host waits for the gather
 * that used a buffer two pieces ago while the latest still travels.  pna_gpu_gather_ordered = start + wait.
 * pna_gather_verdict is the host arithmetic every rank runs on the gathered (size, capacity) pairs (exported: the CPU tests pin it). */
#define PNA_COMM_ID_BYTES 128
typedef struct pna_gpu_comm pna_gpu_comm;
int  pna_gpu_comm_unique_id(void *id128);
int  pna_gpu_comm_init(int device_id, const void *id128, int nranks, int rank, pna_gpu_comm **out);
void pna_gpu_comm_destroy(pna_gpu_comm *comm);
const char *pna_gpu_comm_last_error(const pna_gpu_comm *comm);
int  pna_gpu_gather_ordered(pna_gpu_comm *comm, const void *d_local, uint64_t local_len, int root, void *d_out, uint64_t out_cap,
                            uint64_t *sizes, uint64_t *total, void *hip_stream);
int  pna_gpu_gather_ordered_start(pna_gpu_comm *comm, const void *d_local, uint64_t local_len, int root, void *d_out, uint64_t out_cap,
                                  uint64_t *sizes, uint64_t *total, void *hip_stream);
int  pna_gpu_gather_wait(pna_gpu_comm *comm);
uint64_t pna_gpu_gather_ticket(const pna_gpu_comm *comm);
int  pna_gpu_gather_wait_for(pna_gpu_comm *comm, uint64_t ticket);
int  pna_gather_verdict(const uint64_t *pairs, int nranks, int root, uint64_t *sizes, uint64_t *offs);
/* offs[r] = where rank r's part starts in the gathered stream, offs[nranks] = the total (host arithmetic; what the gather above uses) */
int  pna_gather_offsets(const uint64_t *sizes, int nranks, uint64_t *offs);

/* ---- Archive::write_file / write_stream_entry (lib/src/archive/write.rs:276-299,730-777): an entry written WHILE its data arrives.
 * The record has no fSIZ (the size is not known when FHED goes out): FHED, the caller's already framed extra + metadata chunks
 * (`meta`, may be NULL), then the compressed stream as FDAT chunks -- one per burst the encoder hands to the ChunkStreamWriter
 * (lib/src/chunk/write.rs:32-47: every write() becomes a chunk of at most max_chunk_size bytes; the encoders flush in bursts of at most
 * 32 KiB, the size of the FDAT / SDAT pieces in the reference's stream-written fixtures) --, then FEND.  begin() emits FHED + meta,
 * write() buffers like CompressionWriter::write, finish() compresses (group commit with the context's other writers), emits the FDAT
 * chunks and FEND and frees the writer; abort() frees it without output (the archive is then unusable, as in the reference).
 * max_chunk_size 0 = u32::MAX.  Thread rules: those of pna_gpu_stream_*. */
typedef struct pna_gpu_entry_writer pna_gpu_entry_writer;
int  pna_gpu_stream_entry_begin(pna_gpu_ctx *ctx, int algo, int level, const char *name, const void *meta, size_t meta_len,
                                uint32_t max_chunk_size, pna_sink_fn sink, void *user, pna_gpu_entry_writer **out);
int  pna_gpu_stream_entry_write(pna_gpu_entry_writer *w, const void *buf, size_t len);
int  pna_gpu_stream_entry_finish(pna_gpu_entry_writer *w);
void pna_gpu_stream_entry_abort(pna_gpu_entry_writer *w);

/* ---- streaming facade with the shape of CompressionWriter<W> (lib/src/compress.rs:32-41,66-75):
 * write() buffers, finish() == try_into_inner(): compresses and pushes the stream into the sink (== W::write).
 * THREADS: unlike the rest of this header, the stream functions may be called from any number of host threads on ONE context at
 * the same time -- that is how the reference drives its encoders (one writer per rayon task, cli/src/command/core.rs:505-517).
 * finish() is a group commit: concurrent finishes are collected into one device batch (the thread that finds no batch in flight
 * leads it; finishes arriving meanwhile form the next batch) and every caller receives its own stream through its own sink on
 * its own thread.  Each stream object belongs to one thread; do not mix stream calls with the context's other entry points without
 * external synchronisation.  PNA_STREAM_LINGER_US (environment) fixes how long a leader waits for stragglers before it submits; unset: 200 us once more than one
 * writer has been seen, nothing for a lone writer. */
typedef struct pna_gpu_stream pna_gpu_stream;
int  pna_gpu_stream_new(pna_gpu_ctx *ctx, int algo, int level, pna_sink_fn sink, void *user, pna_gpu_stream **out);
int  pna_gpu_stream_write(pna_gpu_stream *s, const void *buf, size_t len);
int  pna_gpu_stream_flush(pna_gpu_stream *s);           /* no-op like the buffered encoders' flush        */
int  pna_gpu_stream_finish(pna_gpu_stream *s);          /* consumes s                                      */
void pna_gpu_stream_abort(pna_gpu_stream *s);           /* drop without output (failed builder is discarded) */
/* device batches run / entries carried / largest batch so far on behalf of pna_gpu_stream_finish (any pointer may be NULL) */
int  pna_gpu_stream_stats(pna_gpu_ctx *ctx, uint64_t *batches, uint64_t *entries, uint64_t *largest_batch);

/* ---- solid mode: one logical stream, split into independent 1 MiB frames inside the kernels
 * (replaces the single serial encoder of SolidArchive, lib/src/archive/write.rs:443-470,575-580,716-727). */
int  pna_gpu_compress_solid(pna_gpu_ctx *ctx, int algo, int level, const void *src, size_t src_len,
                            pna_sink_fn sink, void *user);

/* ---- introspection used by tests and the benchmark */
typedef struct {
    double   ms_lz, ms_stats, ms_lit, ms_seq, ms_pack;   /* HIP-event time of each stage of the last batch (large zstd batches run the literal coder
                                                          * beside the sequence coder on a second stream: ms_lit is then ~0 and ms_seq covers both) */
    uint64_t in_bytes, out_bytes, n_segments, n_blocks;
    double   ms_frame;                                   /* k_frame (pna_gpu_create_archive_device only)       */
    double   ms_cipher;                                  /* k_aes_* (archives written with a cipher)           */
    double   ms_lz_match;                                /* of ms_lz: the match kernel (k_lzm) launches of the split LZ stage, summed */
    uint64_t lz_match_launches;                          /* ... and how many there were (0: the batch went through the one-kernel form).  After a DEFLATE DECODE call: the large foreign
                                                          * streams that were decoded in chunks between block starts found by trial (the others of that size took one wave's walk) */
    uint32_t blk_log;                                    /* block size of the last (sub-)batch = 1 << blk_log: 17, or 13..16 in latency mode   */
    uint32_t lz_units;                                   /* latency mode: workgroups (units) of the LZ stage's launch; 0: one per segment       */
} pna_gpu_timing;
int  pna_gpu_last_timing(const pna_gpu_ctx *ctx, pna_gpu_timing *out);

/* LZ-stage outputs of the last device batch for one block (tests compare them with the oracle's pna_lz_block).
 * seqs: packed u64 (off:20 | ml:18<<20 | ll:18<<38). Copies at most cap_* items; returns counts. */
int  pna_gpu_debug_block(pna_gpu_ctx *ctx, uint32_t block, uint64_t *seqs, uint32_t cap_seqs, uint32_t *nseq,
                         uint8_t *lits, uint32_t cap_lits, uint32_t *nlit);

/* Host walk through the device CRC schedule of k_frame with the same tables; returns crc32("FDAT" || payload).
 * CPU-only tests use it to check the table construction; it is not on any product path. */
uint32_t pna_gpu_debug_crc_schedule(const void *payload, size_t len);

/* Page-locked staging memory the context holds (the host pipelines' slots): tests pin the bounded-memory claims with it; not on any product path. */
uint64_t pna_gpu_debug_pinned_bytes(pna_gpu_ctx *ctx);

/* Diagnostic build of the LZ kernel (ctx created with flag 0x100): per-phase s_memtime sums over all waves, cleared on read. */
int  pna_gpu_debug_lz_stamps(pna_gpu_ctx *ctx, unsigned long long *out8);

/* ---- benchmark support (not part of the reference's surface): fills d_dst with `n_files` synthetic files of
 * `file_len` bytes each (file i at i * stride), bit-identical to oracle/corpus_model.c. kind: 0 enwik-style text,
 * 1 random-text, 2 random bytes, 3 zeros, 4 repeated byte. */
int  pna_bench_corpus_fill_device(pna_gpu_ctx *ctx, int kind, uint64_t first_file, uint64_t n_files,
                                  uint64_t file_len, uint64_t stride, void *d_dst, void *hip_stream);
/* The reference's fan-out (one entry per task, FIFO, cli/src/command/core.rs:496-537) on `threads` host threads over the streaming
 * facade: stream_new / one write of the whole entry / finish into a counting sink.  Returns seconds; *out_bytes = compressed bytes
 * the sinks received, *rc = first error. */
double pna_bench_stream_threads(pna_gpu_ctx *ctx, int algo, int level, unsigned threads, size_t n, const void *const *src,
                                const size_t *src_len, uint64_t *out_bytes, int *rc);

#ifdef __cplusplus
}
#endif
#endif
