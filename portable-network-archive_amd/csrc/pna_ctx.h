// pna_host.cpp -- host side of libpna_gpu.so: context, workspace, batch planning and the C ABI of
// include/pna_gpu.h.  Mirrors the construction/finish protocol of the reference's CompressionWriter
// (lib/src/compress.rs:21-76, lib/src/entry/write.rs:251-265) and the per-entry fan-out of
// cli/src/command/core.rs:496-537.  No CPU compression path exists here: everything goes through the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <chrono>
#include <new>
#include <functional>
#include <memory>
#include <sys/random.h>
#include "pna_dev.h"
#include "../../include/pna_gpu.h"
#include "../../include/pna_archive.h"

namespace pna {
void launch_lz_small(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
                     uint32_t flags, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, const LzParseGrid *pg, bool w3);
void launch_lz(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
               uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, hipEvent_t ev_match, uint32_t *gtab, const LzParseGrid *pg);
uint32_t lz_gtab_log();
void launch_deflate_stage1(const uint8_t *src, const SegDesc *segs, uint32_t nseg, const uint32_t *blk_seg, uint32_t nblk,
                           const uint64_t *seqs, const uint8_t *lits, BlkInfo *blk, const uint4 *ctab, DeflTables *tabs, uint8_t *outc,
                           uint64_t *seg_size, uint64_t *seg_off, hipStream_t st, hipEvent_t *ev, uint32_t dbg, bool stored_only, bool wave_per_seg);
void launch_deflate_write(const uint8_t *src, const SegDesc *segs, const uint32_t *blk_seg, uint32_t nblk, const BlkInfo *blk,
                          const uint64_t *seg_off, const uint64_t *seg_size, const uint8_t *outc, const uint32_t *entry_seg, uint32_t nentry,
                          uint8_t *dst, hipStream_t st, bool stored_only, bool small_blocks);
void launch_entropy_chunk(const SegDesc *segs, uint32_t s0, uint32_t ns, const uint32_t *blk_seg, uint32_t g0, uint32_t nb,
                          const uint64_t *seqs, const uint8_t *lits, BlkInfo *blk, SegTables *tabs, uint8_t *litc, uint8_t *seqc, uint32_t *seqw,
                          uint32_t flags, uint32_t blk_log, uint32_t *hist, hipStream_t st, hipEvent_t *ev, hipStream_t side, hipEvent_t fork, hipEvent_t join, bool single_block, const uint32_t *seq_hist);
void launch_default_tables(hipStream_t st);   // k_entropy.hip: the predefined sequence tables, once per device
void launch_plan(const SegDesc *segs, uint32_t nseg, BlkInfo *blk, const SegTables *tabs, uint64_t *seg_size, uint64_t *seg_off,
                 uint32_t flags, hipStream_t st);
void launch_write(const uint8_t *src, const SegDesc *segs, uint32_t nseg, const uint32_t *blk_seg, uint32_t nblk, const BlkInfo *blk,
                  const SegTables *tabs, const uint64_t *seg_off, const uint8_t *lits, const uint8_t *litc,
                  const uint8_t *seqc, uint8_t *dst, bool any_empty, hipStream_t st, bool small_blocks);
void lz_read_stamps(unsigned long long *out);
void launch_frame(const FrameDesc *fd, uint32_t nentry, const uint8_t *blob, const CrcTabs *ct, uint8_t *dst, uint64_t cap16,
                  uint32_t fend_crc, const char ty[4], bool with_fend, hipStream_t st, uint32_t max_payload);
void launch_place(const void *pd, uint32_t n, const uint8_t *src, uint8_t *dst, hipStream_t st);
void launch_gather(const void *pd, uint32_t n, const uint8_t *src, uint8_t *dst, hipStream_t st);
void launch_link_copy(const uint8_t *src, uint8_t *dst, size_t n, uint32_t wgs, hipStream_t st);
void launch_link_gather(const void *segs, uint32_t nseg, uint32_t wgs, hipStream_t st);   // k_frame.hip: {src, dst, len} x nseg, page-locked sources
void launch_layout(FrameDesc *fd, uint8_t *blob, const uint32_t *entry_seg, const uint64_t *seg_off, uint32_t nentry, uint32_t nseg, uint64_t out_base,
                   uint64_t *segdst, uint64_t *ent_off, uint64_t *total, hipStream_t st);
void launch_frame_pieces(const FrameDesc *fd, uint32_t n, const CrcTabs *ct, const uint8_t *buf, uint64_t cap16, const char ty[4], uint32_t *states, hipStream_t st);
struct CrcPatchH { int64_t off; uint32_t crc, mask; };   // = CrcPatch of k_frame.hip
void launch_crc_patch(const void *patches, uint32_t n, uint8_t *dst, hipStream_t st);
void launch_frame_verify(const FrameDesc *fd, uint32_t n, const CrcTabs *ct, const uint8_t *buf, uint64_t cap16, const char ty[4], uint32_t *verify, hipStream_t st, uint32_t max_payload);
void launch_zdec(ZFrame *frames, uint32_t n, const uint8_t *src, uint8_t *dst, uint8_t *lit_scratch, uint32_t dbg, hipStream_t st);
void launch_zxxh(ZFrame *frames, uint32_t n, const uint8_t *src, const uint8_t *dst, hipStream_t st);
void launch_zscan(const ZEntry *ents, uint32_t n, const uint8_t *src, ZFrame *frames, ZFrameX *fx, hipStream_t st);
void launch_zcount(const ZEntry *ents, uint32_t n, const uint8_t *src, uint32_t *counts, hipStream_t st);
void launch_zlist(const uint8_t *src, uint64_t base, uint64_t len, uint64_t ip0, void *items, uint32_t cap, uint64_t *hdr, hipStream_t st);   // items: {off, len, fcs} x cap (24 B each)
void launch_zparse(ZFrame *frames, ZFrameX *fx, uint32_t n, const uint8_t *src, ZBlock *blocks, ZTables *tabs, uint32_t *huf_list, uint32_t *seq_list,
                   void *work, hipStream_t st);
void launch_zparse_big_a(ZFrame *frames, ZFrameX *fx, const uint32_t *big_list, uint32_t nbig, const uint8_t *src, ZBlock *blocks, uint32_t *one_list, void *work, hipStream_t st);
void launch_zparse_big_b(ZFrame *frames, ZFrameX *fx, uint32_t nblocks, const uint8_t *src, ZBlock *blocks, ZTables *tabs, uint32_t *huf_list, uint32_t *seq_list,
                         void *work, const uint32_t *one_list, hipStream_t st);
void launch_zstreams(uint32_t n_huf, uint32_t n_seq, const uint32_t *huf_list, const uint32_t *seq_list, const void *work, ZBlock *blocks,
                     const ZFrame *frames, const ZTables *tabs, const uint8_t *src, uint8_t *lit_scratch, uint64_t *seqs, hipStream_t st);
struct ISChunkH { uint64_t start_bit, end_bit, lit_base, out_base, rec_base, end_found, mtot; uint32_t nlit, nrec, status, adler; };   // = ISChunk of k_inflate.hip
static_assert(sizeof(ISChunkH) == 72, "ISChunk layout");
void launch_ispec(const uint8_t *src, uint64_t src_off, uint64_t src_len, uint32_t cbytes, uint32_t nchunks, uint64_t *start, hipStream_t st);
void launch_inflate_chunks(ZFrame *frames, ZFrameX *fx, uint32_t frame, void *chunks, uint32_t nchunks, uint32_t emit, const uint8_t *src, ZBlock *blocks,
                           uint8_t *lit_scratch, uint64_t *seqs, hipStream_t st);
void launch_inflate(ZFrame *frames, ZFrameX *fx, uint32_t n, const uint8_t *src, ZBlock *blocks, uint8_t *lit_scratch, uint64_t *seqs, const uint32_t *mode, hipStream_t st);
void launch_icount(const uint8_t *src, const uint64_t *off, const uint64_t *len, uint32_t n, uint32_t *count, uint32_t G, hipStream_t st);
void launch_vinflate(ZFrame *frames, ZFrameX *fx, uint32_t n, const void *pieces, uint32_t npieces, uint64_t *pb, uint32_t *mode, uint32_t *cntg, uint32_t G, const uint8_t *src,
                     ZBlock *blocks, uint8_t *lit_scratch, uint64_t *seqs, hipStream_t st);
void launch_iadler(ZFrame *frames, const ZFrameX *fx, const ZBlock *blocks, uint32_t n, const uint32_t *cbase, uint32_t npieces, const uint8_t *dst,
                   void *part, hipStream_t st);
void launch_zexec_groups(ZFrame *frames, const ZFrameX *fx, uint32_t n, ZBlock *blocks, const void *pieces, uint32_t npieces, const uint8_t *src, const uint8_t *lit_scratch,
                         const uint64_t *seqs, uint8_t *dst, hipStream_t st);
struct ZxFrame { uint64_t dst_off, dst_len; uint32_t blk_base, nblk, status, unresolved; };      // k_zexec_par.hip
int launch_zexec_par(ZxFrame *zf, const ZxFrame &h, const ZBlock *blocks, const uint8_t *src, const uint8_t *lit_scratch, uint64_t *seqs, uint32_t *rep_scratch,
                     uint32_t *words, uint8_t *dst, uint32_t *status_out, uint32_t *rounds_out, hipStream_t st,
                     uint32_t nwin, const uint32_t *win_blk, const uint64_t *win_off);
void launch_zexec(ZFrame *frames, const ZFrameX *fx, uint32_t n, ZBlock *blocks, const uint8_t *src, const uint8_t *lit_scratch,
                  const uint64_t *seqs, uint8_t *dst, hipStream_t st);
std::string pna_sanitize_name(const char *name, size_t n);
void frame_inner_entry_empty(std::vector<uint8_t> &o, const char *name);
void frame_solid_head(std::vector<uint8_t> &o, int compression);
void frame_solid_head_enc(std::vector<uint8_t> &o, int compression, int encryption, int cipher_mode, const char *phsf, const uint8_t *prefix, size_t prefix_len);
void frame_solid_tail(std::vector<uint8_t> &o);
void frame_archive_head(std::vector<uint8_t> &o, uint32_t archive_number);
void frame_archive_tail(std::vector<uint8_t> &o);
void frame_entry_prefix(std::vector<uint8_t> &o, const char *name, int compression, uint64_t raw_size, uint32_t payload_len);
size_t frame_entry_prefix_bound(const char *name);
size_t frame_entry_prefix_into(uint8_t *out, const char *name, int compression, uint64_t raw_size);
uint32_t frame_fend_crc();
void frame_entry_prefix_enc(std::vector<uint8_t> &o, const char *name, int compression, uint64_t raw_size, int encryption, int cipher_mode,
                            const char *phsf, const uint8_t *prefix, size_t prefix_len);
std::vector<uint8_t> frame_fhed_bytes(const char *name, int compression, int encryption, int cipher_mode);
void sha256_bytes(const void *a, size_t an, const void *b, size_t bn, uint8_t out[32]);
void hkdf_sha256_32(const void *ikm, size_t ikm_len, const void *salt, size_t salt_len, const void *info, size_t info_len, uint8_t okm[32]);
void launch_gcm_tag(const GcmEntry *ents, uint32_t n, uint8_t *buf, hipStream_t st);
void launch_gcm_verify(const GcmEntry *ents, uint32_t n, const uint8_t *buf, const uint8_t *expect, uint32_t *bad, hipStream_t st);
void launch_aes_cbc_dec(const CipherUnit *units, uint32_t n, const uint8_t *ivs, const AesDecTabs *tabs, uint8_t *buf, const AesKey &dkey, uint32_t *plain_len, hipStream_t st);
size_t frame_entry_prefix_enc_bound(const char *name, const char *phsf);
void launch_aes_ctr(const CipherUnit *units, uint32_t n, const uint8_t *ivs, const AesTabs *tabs, uint8_t *buf, const AesKey &key, const AesKey *keys, hipStream_t st);
void launch_aes_cbc_enc(const CipherUnit *units, uint32_t n, const uint8_t *ivs, const AesTabs *tabs, uint8_t *buf, const AesKey &key, hipStream_t st);
void launch_corpus(int kind, uint64_t first_file, uint64_t n_files, uint64_t file_len, uint64_t stride,
                   const uint8_t *vocab, const uint64_t *cum, const uint32_t *phrases, uint8_t *dst, hipStream_t st);
}
using namespace pna;

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = n + (n >> 3) + 4096;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; if (hipMalloc(&p, n) != hipSuccess) { p = nullptr; return -1; } want = n; }
        cap = want; return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct PinBuf {                                  // page-locked host staging: async copies that really are asynchronous
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        size_t want = n + (n >> 2) + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return -1; }
        cap = want; return 0;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// Tuning knobs of a context (pna_gpu_set_option; include/pna_gpu.h lists them).  Each starts from the environment variable of the same
// purpose, read ONCE in pna_gpu_init -- no entry point consults the environment afterwards.
struct Tuning {
    long lz_split = 1;               // PNA_LZ_SPLIT: 0 one-kernel LZ stage, 1 split form for long runs (default), 2 split form with the wave-per-region parse kernel
    long lz_split_blocks = 131072;   // PNA_LZ_SPLIT_BLOCKS: blocks per run of the split form (the words workspace holds one run: 3 bytes per input byte -- 16 GiB of input, 48 GiB of words at most; round 5:
                                     // until then 32 768 = 4 GiB per run, and the 10 GiB headline paid three kernel tails per stage -- one run: 74.7 -> 73.5 ms; a workspace that cannot be had is halved, as before)
    long lz_split_min = 0;           // PNA_LZ_SPLIT_MIN: shortest run (segments) that takes the split form (0: every run; shorter ones take the one-kernel form)
    long lz_pbuf_fail = 0;           // PNA_LZ_PBUF_FAIL: testing -- behave as if the words workspace could not be allocated
    long pipeline_chunks = 1;        // PNA_PIPELINE_CHUNKS: zstd entropy stage of chunk k next to the LZ stage of chunk k + 1 (measured: slower)
    long max_chunk_size = 0;         // PNA_MAX_CHUNK_SIZE: FlattenWriter::max_chunk_size of the archive entry points without such a parameter (0 = the reference's default, u32::MAX)
    long sub_mib = 256;              // PNA_SUB_MIB: largest sub-batch (input bytes) of the bounded host pipeline
    long sub_ramp_down = 0;          // PNA_SUB_RAMP_DOWN: sub-batches shrink towards the end of the input (measured: no gain)
    long stage_threads = 0;          // PNA_STAGE_THREADS: host threads that stage entries into page-locked memory (0: min(8, cores / 2))
    long extract_win_mib = 1024;     // PNA_EXTRACT_WIN_MIB: archive bytes per window of the extract driver
    long solid_win_mib = 256;        // PNA_SOLID_WIN_MIB: pna_gpu_create_solid_archive_host takes the serialised inner entries through in windows of this many MiB (page-locked memory: ~4 windows)
    long batch_piece_mib = 256;      // PNA_BATCH_PIECE_MIB: pna_gpu_compress_batch takes a large batch through in pieces of this size (0: one piece)
    long inflate_serial = 0;         // PNA_INFLATE_SERIAL: deflate decoding on the wave-per-stream walk only
    long zdec_serial = 0;            // PNA_ZDEC_SERIAL: zstd decoding with one workgroup per frame only
    long zdec_dbg = 0;               // PNA_ZDEC_DBG: the one-workgroup kernel's diagnostics (1 skip execution, 2 skip sequences, 4 skip Huffman streams, 8 small re-base distances)
    long blk_log = 0;                // PNA_BLK_LOG: block size of every batch = 1 << blk_log (13..17); 0 = by batch size (latency mode)
    long unit_log = 0;               // PNA_LZ_UNIT_LOG: LZ units of 1 << unit_log bytes (>= the block size, <= 20); 0 = by batch size
    long latency_max_mib = 128;      // PNA_LATENCY_MAX_MIB: batches of at most this many MiB of input run in latency mode (0: never)
    long tail_units = 1;             // PNA_TAIL_UNITS: the segments behind a run's last full round of the CUs go through the match kernel in units of one block
    long lazy2 = 2;                  // PNA_LAZY2: how far the lazy level sets look ahead beyond the next position: 2 = two more positions (default), 1 = one more, 0 = none (the high sets: one)
    long single_frame = 0;           // PNA_SINGLE_FRAME: 1: a zstd entry is ONE frame whatever its size (header once, last-block bit once; SURVEY 8 a14's fallback for a reader that would
                                     // not take concatenated frames -- zstd's own Decoder, which the reference uses, does); 0 (default): one frame per 1 MiB segment
    long stream_gather_wgs = 48;     // PNA_STREAM_GATHER_WGS: workgroups of the kernel that copies a batch's page-locked slabs to the device (0: one hipMemcpyAsync per slab, ~30 us each)
    long stream_overlap_mib = 24;    // PNA_STREAM_OVERLAP_MIB: while a batch runs on the device the next one is taken (and copied in beside it) only once the queue holds this much
    long stream_batch_mib = 256;     // PNA_STREAM_BATCH_MIB: input bytes one batch of the streaming facade takes at most (the queue's rest is the next batch, which is copied in meanwhile)
    long zexec_par_min_mib = 8;      // PNA_ZEXEC_PAR_MIN_MIB: zstd frames whose content takes this many MiB and more are executed in parallel by pointer jumping (0: never)
    long zdec_fallback_max_mib = 0;  // PNA_ZDEC_FALLBACK_MAX_MIB: zstd frames of more content than this that the parallel paths cannot take are REFUSED (PNA_E_UNSUPPORTED) instead of decoded by one workgroup at ~11 MiB/s (0: no limit) -- a host may prefer its CPU decoder
    long zexec_win_mib = 1024;       // PNA_ZEXEC_WIN_MIB: output bytes of one window of the parallel executor (its words count 31 bits from the window's start: at most 1 024; tests take a few MiB)
    long small_geometry = 1;         // PNA_SMALL_GEOMETRY: 1 (default): segments of at most 4 096 bytes run the small geometry of the match finder (pna_dev.h SMALL_SEG: one wave per segment, sub-tiles of 256 positions); 0: the large one like every segment (they then find no match: one tile)
    long tab3 = 1;                   // PNA_TAB3: 1 (default): the zstd sets on the 32 / 16 KiB geometries keep their table PACKED (three 21-bit entries per 64-bit LDS word: 49 062 / 55 206 slots, lz_common.h); 0: 32-bit entries (32 704 / 36 800)
    long strong2 = 1;                // PNA_STRONG2: 1 (default): zstd levels 4 .. 22 on their standard geometries adopt over eight positions as well and count up to 15 back bytes (levels 6 - 9: 2.864 -> 2.880); 0: the default set's three rounds
    long seq_hist = 1;               // PNA_SEQ_HIST: 1 (default): large zstd batches behind the split LZ stage -- the parse kernel counts every block's sequence codes, k_stats walks the literals only (2.6 -> 1.1 ms per 10 000 segments, + 0.5 in the parse kernel); 0: k_stats reads the sequences once more
    long far1 = 1;                   // PNA_FAR1: 1 (default): the zstd light / default sets (packed table, 32 KiB window) verify at most 63 far candidates per wave of 256 positions -- one compacted round of k_lzm -- and drop the rest (FLAG_FAR1: - 0.16 % of ratio, - 6 % of the match kernel); 0: every far candidate, in as many rounds as it takes
    long win32k = 1;                 // PNA_WIN32K: 1 (default): the zstd default set on the 32 KiB-window geometry of the match finder (32 704 table slots), the high set on the 16 KiB one (36 800); 0: both on 64 KiB / 24 512; 2: both on 16 KiB
    long lit_beside_seq = 1;         // PNA_LIT_BESIDE_SEQ: large zstd batches: the literal coder on a second stream next to the sequence coder
    long strong_gtab = 1;            // PNA_STRONG_GTAB: zstd levels 10 .. 22 with the match kernel's hash tables in global memory (2^19 slots per segment); 0: the LDS table
    long dev_layout = 1;             // PNA_DEV_LAYOUT: archive layout of plain one-chunk entries on the device (k_layout); 0: on the host, after a wait for the sizes
    long trace = 0;                  // PNA_TRACE: phase times of the host pipelines on stderr
    long d2h_wgs = 6;                // PNA_D2H_WGS: workgroups of the kernel that carries a sub-batch's archive bytes to the host (0: the copy engine / runtime's choice)
    long hist_by_block = -1;         // PNA_HIST_BY_BLOCK: zstd entropy stage in its per-block form (1: k_hist, k_seqa, k_seqb) or its per-segment form (0: k_stats, k_seq); -1: by batch size
};
struct TuningName { const char *name, *env; long Tuning::*field; long lo, hi; };

struct pna_gpu_stream;
struct pna_gpu_ctx {
    Tuning tun;
    // the plan of a sub-batch -- segment descriptors, LZ units (latency mode: pieces of segments, one workgroup each), block -> segment, entry -> first
    // segment -- is staged in ONE page-locked blob and travels in one copy; the per-segment histograms of k_hist lie behind the BlkInfo array (one memset)
    DevBuf plan; PinBuf h_plan;
    DevBuf d_tail; PinBuf h_tail; size_t tail_used = 0;     // unit descriptors of the segments behind a run's last full round of workgroups (lz_stage)
    SegDesc *d_segs = nullptr, *d_units = nullptr; uint32_t *d_blk_seg = nullptr, *d_entry_seg = nullptr, *d_hist = nullptr;
    uint32_t last_blk_log = PNA_BLK_LOG, last_units = 0;
    int device = 0;
    uint32_t flags = 0;
    uint32_t call_flags = 0;                        // flags of the current call: the level picks the parse (level_flags)
    bool call_lazy3 = false;
    bool call_lazy2 = false;                         // two-step lazy deferral (FLAG_LAZY2 of the LZ kernels)
    uint32_t n_cus = 256;                           // compute units of the device (hipDeviceProp_t::multiProcessorCount): a full round of one-workgroup-per-CU kernels
    bool call_stored = false;                       // deflate level 0: stored blocks only (Compression::none())
    bool call_strong2 = false;                      // ... the high / max sets' fourth adoption round and 15 back bytes (FLAG_STRONG2: zstd 4 .. 22 on the packed 16 KiB geometry or the global table)
    uint32_t *lzp_hist = nullptr; bool lzp_hist_all = false;   // the current sub-batch: the counters the parse kernel adds the sequence codes to (null: none), and whether EVERY segment's parse did
    bool call_tab3 = false;                         // ... its table packed (lz_common.h TAB3)
    bool call_w16 = false;                          // ... the 16 KiB window (zstd 6..9)
    bool call_gtab = false, call_w32 = false;       // ... and where the match finder's table lies / its LDS geometry (set_call_level)
    std::vector<hipEvent_t> lzm_ev; size_t lzm_used = 0;   // event pairs around the match kernel launches of the current sub-batch (timed calls)
    std::vector<uint8_t> lzm_nl;                            // launches inside each pair (2 where a run's last segments go in units)
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    DevBuf blk, tabs, seqs, lits, litc, seqc, seqw, seg_size, seg_off, stage_in, stage_out, ctab, pbuf;
    DevBuf z_list;                                  // foreign multi-frame payloads: the frame list (k_zlist)
    DevBuf z_big, z_one;                            // large zstd frames: their numbers, their blocks (k_zparse_a -> k_zparse<true>)
    DevBuf z_spec;                                  // large foreign zlib streams: chunk starts + chunk descriptors (k_ispec, k_inflate's chunk mode)
    DevBuf z_words, z_rep, z_zxf;                   // the parallel executor of large zstd frames (k_zexec_par.hip): a word per output byte, histories per block
    uint32_t zexec_par_rounds = 0;                  // pointer-jumping rounds of the latest large frame (diagnostics)
    std::vector<uint64_t> pl_off, pl_len, pl_cap, pl_eoff;   // the host pipeline's per-entry plan (staging offsets, lengths, capacity terms), kept between calls
    uint32_t inflate_spec_streams = 0;              // streams of the latest inflate call that went through the speculative chunk decoder (diagnostics)
    DevBuf c_vocab, c_cum, c_phr;
    DevBuf gtab;                                    // hash tables of the strong level set's match kernel (global memory)
    DevBuf fr_desc, fr_blob, fr_segdst, fr_entoff, crc_tabs;
    PinBuf h_entoff;
    DevBuf x_arc, x_pk, x_raw[2], x_desc, x_place, x_flag, x_tags, x_plen, aes_dtabs;
    hipStream_t x_cp = nullptr; hipEvent_t x_ev[2] = {}, x_done = nullptr;   // extract driver: D2H of window k on x_cp next to window k+1's work
    bool aes_dec_ready = false;        // read side (pna_gpu_extract_archive_host): archive image, packed payloads, decoded entries
    DevBuf z_vp, z_pb, z_mode;                                 // lane-per-piece inflate: piece list, piece boundaries, per-stream mode
    DevBuf ci_spread, ci_spread_desc;                          // GCM entries of several segments: their compact payloads, the pieces to move
    DevBuf aes_tabs, ci_units, ci_ivs, ci_keys, ci_gcm;        // cipher stage: round tables, unit descriptors, IVs; GCM: per-entry round keys, segment descriptors
    bool aes_ready = false;
    hipEvent_t ev_ci[2] = {};
    DevBuf solid_plain, solid_desc, solid_blob, solid_place;   // serialised inner entries of a solid archive
    DevBuf z_ents, z_frames, z_lit;                            // decoder descriptors, literal scratch
    DevBuf z_fx, z_blocks, z_tabs, z_seqs, z_hlist, z_slist, z_work, z_fb, z_cbase, z_apart;   // lane-parallel decoder workspace
    PinBuf h_desc, h_blob, h_segdst, h_segoff;
    // pipelined host path (pna_gpu_create_archive_host): two slots of staging
    PinBuf hp_in[4], hp_out[2];
    // page-locked buffers handed to the host (pna_gpu_host_alloc): entries that live in one go to the device straight from there (no staging copy)
    std::mutex lent_mu; std::vector<std::pair<const uint8_t *, size_t>> lent;
    DevBuf dp_in[4], dp_out[2];
    hipStream_t cp_in = nullptr, cp_out = nullptr;
    uint64_t call_total = 0;                          // input bytes of the whole call that run_subbatch's sub-batch belongs to (0: the sub-batch is the call): decides the block size
    hipStream_t aux = nullptr;                        // entropy stage of chunk c runs here while k_lz works on chunk c+1
    static constexpr int MAXCH = 8;
    hipEvent_t ev_lz[MAXCH + 1] = {}, ev_en[MAXCH][4] = {}, ev_join = nullptr, ev_fork = nullptr;
    hipEvent_t ev_in[4] = {}, ev_out[2] = {};
    bool crc_ready = false;
    bool corpus_ready = false;
    std::string err;
    pna_gpu_timing timing = {};
    uint32_t last_nblk = 0;
    uint32_t plan_log = PNA_BLK_LOG;                // block size the current call's sub-batches are planned with (plan_call)
    size_t max_blocks = (size_t)1 << 17;            // blocks per sub-batch: what ~96 GiB of per-block workspace hold at that block size (16 GiB of input at 128 KiB)
    // group commit of the streaming facade (pna_gpu_stream_finish from many host threads -> one device batch)
    std::mutex comb_mu, run_mu;            // comb_mu: queue + leader flag; run_mu: the device batch itself and ctx->err
    std::condition_variable comb_cv, gate_cv;   // gate_cv: the ONE leader waiting for its slot / the device / a larger queue (every push signals it: not the hundreds of writers on comb_cv)
    std::vector<pna_gpu_stream *> comb_queue;
    bool comb_leader = false;
    uint64_t comb_batches = 0, comb_entries = 0, comb_max = 0, comb_seq = 0;
    uint32_t comb_linger_us = 0xFFFFFFFFu; // PNA_STREAM_LINGER_US: the leader waits this long for more finishes before it submits (unset: adaptive)
    size_t comb_last = 0;                  // entries of the previous batch
    size_t comb_peak = 0;                  // the cohort the linger waits for: the largest recent batch (max(batch, peak - 1): a straggler's batch of one does not shrink it)
    // page-locked memory of the streaming facade: write() copies straight into 1 MiB slabs of a pool (no staging copy before the H2D
    // copy), the compressed streams come back into one of two page-locked output slots and the owners drain them from there
    std::mutex pool_mu;
    std::vector<void *> pool_arenas; std::vector<uint8_t *> pool_free; size_t pool_bytes = 0, pool_cap = 4096ull << 20;
    // Round 4: the batches of the facade form a PIPELINE of three stages over three slots -- (1) the H2D copies of batch k + 1 from the writers' slabs, (2) the
    // device batch k, (3) the D2H copy of batch k - 1's streams -- each on a stream of its own; a leader holds `comb_leader` only while it takes its batch
    // and copies it in, `run_mu` only for the device batch.  A batch takes at most stream_batch_mib of input (what fills the chip), the rest of the queue
    // is the next leader's.
    static constexpr int S_SLOTS = 3;
    PinBuf s_out[S_SLOTS];
    DevBuf st_in[S_SLOTS], st_out[S_SLOTS];
    PinBuf s_segs[S_SLOTS];                 // the copy-in kernel's segment list of the slot's batch
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t s_ev[S_SLOTS] = {nullptr, nullptr, nullptr};
    uint64_t slot_pending[S_SLOTS] = {0, 0, 0};     // streams of the slot's last batch that have not been drained yet (under comb_mu)
    bool device_busy = false;              // a batch of the facade holds the device (under comb_mu).  While it does, a new leader keeps collecting until the queue holds
                                           // stream_overlap_mib -- enough to be worth copying in beside the running batch --; few writers therefore still form ONE batch per
                                           // device turn (the fixed ~0.7 ms of a batch is shared by all of them), many writers fill the pipeline
    uint32_t staged_waiting = 0;           // batches copied in and waiting for the device (under comb_mu): a new leader takes its batch only when there is none --
                                           // while the device is busy the queue keeps growing, so few writers still share batches (4 writers: batches of 2 - 3, not 1)
    std::mutex err_mu;                     // ctx->err from the pipeline's copy stages (the device batch sets it under run_mu as every entry point does)
};

inline int fail(pna_gpu_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
    if (c) { c->err = what; if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); } }
    return code;
}
#define HIPCHK(c, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail((c), PNA_E_HIP, #call, e__); } while (0)

// ---- what the translation units of the host code share (defined in pna_host.cpp unless noted)
constexpr uint64_t CTR_UNIT = 256u << 10;                    // bytes of one CTR work unit (one workgroup)
struct PlaceDescH { uint64_t src_off, dst_off; uint32_t len, pad; };   // = PlaceDesc of k_frame.hip (k_place / k_gather)
struct FrameJob { const char *const *names; int solid; const pna_gpu_cipher *cipher = nullptr; const uint8_t *ivs = nullptr; const pna_gpu_entry_meta *meta = nullptr;
                  uint32_t max_chunk = 0; bool want_offsets = true; };   // FDAT chunks of at most this many bytes (FlattenWriter::max_chunk_size; 0 = the reference's default u32::MAX)
struct GcmMaterial { uint8_t header[75]; AesKey rk; uint32_t h[4], ej0[4]; uint8_t ctr_iv[16]; };

void set_call_level(pna_gpu_ctx *c, int algo, int level);
inline size_t plan_blocks(const pna_gpu_ctx *c, uint64_t len) {
    return (size_t)((len + ((uint64_t)1 << c->plan_log) - 1) >> c->plan_log);
}
// Block size of a batch whose entries are all small: the per-block arrays have the block size as their stride, so a batch of 4 KiB entries on 128 KiB
// blocks would spend 32 times the memory (and a sub-batch per 131 072 entries) that 8 KiB blocks need.  Entries of up to 64 KiB: the power of two that
// holds the largest (>= 8 KiB); anything larger: 128 KiB.  (Latency mode, for small batches of large entries, chooses on top of this in run_subbatch.)
// The input bytes of the whole call for the duration of a scope (pna_gpu_ctx::call_total: run_subbatch picks the block size by the call, not by its sub-batches)
// A scope opened while another is open on the context (an entry point that cuts its call into parts or pieces and runs each through another entry point: the
// multi-context create, the host form of compress_batch) leaves the outer total in place -- every part then picks the block size of the WHOLE call, so the
// bytes do not depend on how the call was cut.
struct CallTotalScope {
    pna_gpu_ctx *c; bool own;
    template <class L> CallTotalScope(pna_gpu_ctx *ctx, const L *len, size_t n) : c(ctx), own(ctx->call_total == 0) {
        if (own) { uint64_t t = 0; for (size_t i = 0; i < n; i++) t += len[i]; c->call_total = t; } }
    CallTotalScope(pna_gpu_ctx *ctx, uint64_t total) : c(ctx), own(ctx->call_total == 0) { if (own) c->call_total = total; }
    ~CallTotalScope() { if (own) c->call_total = 0; }
    CallTotalScope(const CallTotalScope &) = delete; CallTotalScope &operator=(const CallTotalScope &) = delete;
};
inline uint32_t blk_log_for_longest(const pna_gpu_ctx *c, uint64_t mx) {
    if (c->tun.blk_log) return (uint32_t)c->tun.blk_log;
    if (mx > 65536) return PNA_BLK_LOG;
    uint32_t lg = BLK_LOG_MIN;
    while (((uint64_t)1 << lg) < mx) lg++;
    return lg;
}
template <class L>
inline uint32_t small_entry_blk_log(const pna_gpu_ctx *c, const L *src_len, size_t e0, size_t e1) {
    if (c->tun.blk_log) return (uint32_t)c->tun.blk_log;
    uint64_t mx = 0;
    for (size_t e = e0; e < e1; e++) mx = std::max<uint64_t>(mx, src_len[e]);
    return blk_log_for_longest(c, mx);
}
// The per-entry host loops of a sub-batch of 10^5 .. 10^6 small entries (plan, bounds, record prefixes) run on several threads over CONTIGUOUS index
// ranges: fn(t, a, b) for thread t and its range [a, b) of [0, n); the caller's thread takes the first range.
inline unsigned host_loop_threads(size_t n) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::min<size_t>(std::min<unsigned>(16u, hw), std::max<size_t>(1, n / 16384));
}
// The helper threads are a process-wide POOL (round 5): a sub-batch of 10^6 entries runs half a dozen of these loops, and starting + joining fifteen threads for each
// was ~0.5 ms a time -- more than the loops themselves.  One job at a time (a second caller -- another context's thread -- starts threads of its own, as before).
struct HostPool {
    std::mutex mu, busy; std::condition_variable cv, done_cv;
    std::vector<std::thread> th;
    std::function<void(unsigned)> job; unsigned want = 0, gen = 0, left = 0; bool stop = false;
    void worker(unsigned t) {
        unsigned seen = 0;
        for (;;) {
            std::function<void(unsigned)> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || (gen != seen && t < want); });
                if (stop) return;
                seen = gen; j = job;
            }
            j(t);
            { std::lock_guard<std::mutex> lk(mu); if (--left == 0) done_cv.notify_all(); }
        }
    }
    // runs fn(1 .. nt - 1) on the pool's threads and fn(0) on the caller's; false: the pool is taken (the caller falls back)
    bool run(unsigned nt, const std::function<void(unsigned)> &fn) {
        if (!busy.try_lock()) return false;
        {
            std::lock_guard<std::mutex> lk(mu);
            while (th.size() + 1 < nt) { const unsigned t = (unsigned)th.size() + 1; th.emplace_back([this, t]() { worker(t); }); }
            job = fn; want = nt; left = nt - 1; gen++;
        }
        cv.notify_all();
        fn(0);
        { std::unique_lock<std::mutex> lk(mu); done_cv.wait(lk, [&] { return left == 0; }); want = 0; }
        busy.unlock();
        return true;
    }
    ~HostPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &x : th) x.join(); }
};
inline HostPool &host_pool() { static HostPool p; return p; }
template <class F>
inline void par_ranges(size_t n, unsigned nt, F &&fn) {
    if (nt <= 1) { fn(0u, (size_t)0, n); return; }
    auto part = [&fn, n, nt](unsigned t) { if (t == 0) fn(0u, (size_t)0, n / nt); else fn(t, n * t / nt, n * (t + 1) / nt); };
    if (host_pool().run(nt, part)) return;
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back([&part, t]() { part(t); });
    part(0);
    for (auto &x : th) x.join();
}
// per call: the block size the sub-batches are cut with and how many blocks fit the workspace budget
inline void plan_call_longest(pna_gpu_ctx *c, uint64_t longest) {           // plan_call for a caller that knows the longest entry already
    c->plan_log = blk_log_for_longest(c, longest);
    const uint64_t per_block = (uint64_t)seq_cap_of(c->plan_log) * 16 + ((uint64_t)3 << c->plan_log) + 64;
    c->max_blocks = (size_t)std::max<uint64_t>(1024, (96ull << 30) / per_block);
}
template <class L>
inline void plan_call(pna_gpu_ctx *c, const L *src_len, size_t n) {
    c->plan_log = small_entry_blk_log(c, src_len, 0, n);
    const uint64_t per_block = (uint64_t)seq_cap_of(c->plan_log) * 16 + ((uint64_t)3 << c->plan_log) + 64;
    c->max_blocks = (size_t)std::max<uint64_t>(1024, (96ull << 30) / per_block);
}


int  ensure_crc(pna_gpu_ctx *c);
uint32_t crc_gf2_mulmod(uint32_t a, uint32_t b);       // a * b mod P of the CRC-32 (reflected bit order), x^e mod P: what chains the raw registers of a chunk's pieces
uint32_t crc_gf2_xpow(uint64_t e);
int  ensure_aes(pna_gpu_ctx *c);
int  ensure_aes_dec(pna_gpu_ctx *c);
int  check_cipher(pna_gpu_ctx *c, const pna_gpu_cipher *ci);
void aes256_expand(const uint8_t key[32], AesKey &k);
void aes256_dec_key(const AesKey &k, AesKey &d);
void aes256_block_host(const AesKey &k, const uint8_t in[16], uint8_t out[16]);
int  resolve_ivs(pna_gpu_ctx *c, const pna_gpu_cipher *cipher, size_t n, std::vector<uint8_t> &own, const uint8_t **ivs);
uint64_t chunk_limit(uint32_t max_chunk);
size_t meta_len(const pna_gpu_entry_meta *m, size_t e);
bool meta_blob_ok(const uint8_t *p, size_t n);
int  check_meta(pna_gpu_ctx *c, const pna_gpu_entry_meta *meta, size_t n);
int  run_subbatch(pna_gpu_ctx *c, int algo, const uint8_t *d_src, const uint64_t *src_off, const uint64_t *src_len,
                  size_t e0, size_t e1, uint8_t *d_dst, size_t dst_cap, uint64_t out_base, uint64_t *dst_off,
                  hipStream_t st, bool timed, const FrameJob *fj = nullptr);
