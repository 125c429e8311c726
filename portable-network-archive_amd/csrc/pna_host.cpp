// pna_host.cpp -- host side of libpna_gpu.so: context, workspace, batch planning and the C ABI of
// include/pna_gpu.h.  Mirrors the construction/finish protocol of the reference's CompressionWriter
// (lib/src/compress.rs:21-76, lib/src/entry/write.rs:251-265) and the per-entry fan-out of
// cli/src/command/core.rs:496-537.  No CPU compression path exists here: everything goes through the HIP kernels.
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
void launch_lz(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
               uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, hipEvent_t ev_match, uint32_t *gtab, const LzParseGrid *pg);
uint32_t lz_gtab_log();
void launch_deflate_stage1(const uint8_t *src, const SegDesc *segs, uint32_t nseg, const uint32_t *blk_seg, uint32_t nblk,
                           const uint64_t *seqs, const uint8_t *lits, BlkInfo *blk, const uint4 *ctab, DeflTables *tabs, uint8_t *outc,
                           uint64_t *seg_size, uint64_t *seg_off, hipStream_t st, hipEvent_t *ev, uint32_t dbg, bool stored_only);
void launch_deflate_write(const uint8_t *src, const SegDesc *segs, const uint32_t *blk_seg, uint32_t nblk, const BlkInfo *blk,
                          const uint64_t *seg_off, const uint64_t *seg_size, const uint8_t *outc, const uint32_t *entry_seg, uint32_t nentry,
                          uint8_t *dst, hipStream_t st, bool stored_only);
void launch_entropy_chunk(const SegDesc *segs, uint32_t s0, uint32_t ns, const uint32_t *blk_seg, uint32_t g0, uint32_t nb,
                          const uint64_t *seqs, const uint8_t *lits, BlkInfo *blk, SegTables *tabs, uint8_t *litc, uint8_t *seqc, uint32_t *seqw,
                          uint32_t flags, uint32_t blk_log, uint32_t *hist, hipStream_t st, hipEvent_t *ev, hipStream_t side, hipEvent_t fork, hipEvent_t join);
void launch_plan(const SegDesc *segs, uint32_t nseg, BlkInfo *blk, const SegTables *tabs, uint64_t *seg_size, uint64_t *seg_off,
                 uint32_t flags, hipStream_t st);
void launch_write(const uint8_t *src, const SegDesc *segs, uint32_t nseg, const uint32_t *blk_seg, uint32_t nblk, const BlkInfo *blk,
                  const SegTables *tabs, const uint64_t *seg_off, const uint8_t *lits, const uint8_t *litc,
                  const uint8_t *seqc, uint8_t *dst, bool any_empty, hipStream_t st);
void lz_read_stamps(unsigned long long *out);
void launch_frame(const FrameDesc *fd, uint32_t nentry, const uint8_t *blob, const CrcTabs *ct, uint8_t *dst, uint64_t cap16,
                  uint32_t fend_crc, const char ty[4], bool with_fend, hipStream_t st, uint32_t max_payload);
void launch_place(const void *pd, uint32_t n, const uint8_t *src, uint8_t *dst, hipStream_t st);
void launch_gather(const void *pd, uint32_t n, const uint8_t *src, uint8_t *dst, hipStream_t st);
void launch_link_copy(const uint8_t *src, uint8_t *dst, size_t n, uint32_t wgs, hipStream_t st);
void launch_link_gather(const void *segs, uint32_t nseg, uint32_t wgs, hipStream_t st);   // k_frame.hip: {src, dst, len} x nseg, page-locked sources
void launch_layout(FrameDesc *fd, uint8_t *blob, const uint32_t *entry_seg, const uint64_t *seg_off, uint32_t nentry, uint32_t nseg, uint64_t out_base,
                   uint64_t *segdst, uint64_t *ent_off, uint64_t *total, hipStream_t st);
void launch_frame_verify(const FrameDesc *fd, uint32_t n, const CrcTabs *ct, const uint8_t *buf, uint64_t cap16, const char ty[4], uint32_t *verify, hipStream_t st, uint32_t max_payload);
void launch_zdec(ZFrame *frames, uint32_t n, const uint8_t *src, uint8_t *dst, uint8_t *lit_scratch, uint32_t dbg, hipStream_t st);
void launch_zxxh(ZFrame *frames, uint32_t n, const uint8_t *src, const uint8_t *dst, hipStream_t st);
void launch_zscan(const ZEntry *ents, uint32_t n, const uint8_t *src, ZFrame *frames, ZFrameX *fx, hipStream_t st);
void launch_zcount(const ZEntry *ents, uint32_t n, const uint8_t *src, uint32_t *counts, hipStream_t st);
void launch_zparse(ZFrame *frames, ZFrameX *fx, uint32_t n, const uint8_t *src, ZBlock *blocks, ZTables *tabs, uint32_t *huf_list, uint32_t *seq_list,
                   void *work, hipStream_t st);
void launch_zparse_big_a(ZFrame *frames, ZFrameX *fx, const uint32_t *big_list, uint32_t nbig, const uint8_t *src, ZBlock *blocks, uint32_t *one_list, void *work, hipStream_t st);
void launch_zparse_big_b(ZFrame *frames, ZFrameX *fx, uint32_t nblocks, const uint8_t *src, ZBlock *blocks, ZTables *tabs, uint32_t *huf_list, uint32_t *seq_list,
                         void *work, const uint32_t *one_list, hipStream_t st);
void launch_zstreams(uint32_t n_huf, uint32_t n_seq, const uint32_t *huf_list, const uint32_t *seq_list, const void *work, ZBlock *blocks,
                     const ZFrame *frames, const ZTables *tabs, const uint8_t *src, uint8_t *lit_scratch, uint64_t *seqs, hipStream_t st);
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
                     uint32_t *words, uint8_t *dst, uint32_t *status_out, uint32_t *rounds_out, hipStream_t st);
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
    long lz_split_blocks = 32768;    // PNA_LZ_SPLIT_BLOCKS: blocks per run of the split form (the words workspace holds one run)
    long lz_split_min = 0;           // PNA_LZ_SPLIT_MIN: shortest run (segments) that takes the split form (0: every run; shorter ones take the one-kernel form)
    long lz_pbuf_fail = 0;           // PNA_LZ_PBUF_FAIL: testing -- behave as if the words workspace could not be allocated
    long pipeline_chunks = 1;        // PNA_PIPELINE_CHUNKS: zstd entropy stage of chunk k next to the LZ stage of chunk k + 1 (measured: slower)
    long max_chunk_size = 0;         // PNA_MAX_CHUNK_SIZE: FlattenWriter::max_chunk_size of the archive entry points without such a parameter (0 = the reference's default, u32::MAX)
    long sub_mib = 256;              // PNA_SUB_MIB: largest sub-batch (input bytes) of the bounded host pipeline
    long sub_ramp_down = 0;          // PNA_SUB_RAMP_DOWN: sub-batches shrink towards the end of the input (measured: no gain)
    long stage_threads = 0;          // PNA_STAGE_THREADS: host threads that stage entries into page-locked memory (0: min(8, cores / 2))
    long extract_win_mib = 1024;     // PNA_EXTRACT_WIN_MIB: archive bytes per window of the extract driver
    long batch_piece_mib = 256;      // PNA_BATCH_PIECE_MIB: pna_gpu_compress_batch takes a large batch through in pieces of this size (0: one piece)
    long inflate_serial = 0;         // PNA_INFLATE_SERIAL: deflate decoding on the wave-per-stream walk only
    long zdec_serial = 0;            // PNA_ZDEC_SERIAL: zstd decoding with one workgroup per frame only
    long zdec_dbg = 0;               // PNA_ZDEC_DBG: the one-workgroup kernel's diagnostics (1 skip execution, 2 skip sequences, 4 skip Huffman streams, 8 small re-base distances)
    long blk_log = 0;                // PNA_BLK_LOG: block size of every batch = 1 << blk_log (13..17); 0 = by batch size (latency mode)
    long unit_log = 0;               // PNA_LZ_UNIT_LOG: LZ units of 1 << unit_log bytes (>= the block size, <= 20); 0 = by batch size
    long latency_max_mib = 192;      // PNA_LATENCY_MAX_MIB: batches of at most this many MiB of input run in latency mode (0: never)
    long tail_units = 1;             // PNA_TAIL_UNITS: the segments behind a run's last full round of the CUs go through the match kernel in units of one block
    long lazy2 = 2;                  // PNA_LAZY2: how far the lazy level sets look ahead beyond the next position: 2 = two more positions (default), 1 = one more, 0 = none (the high sets: one)
    long single_frame = 0;           // PNA_SINGLE_FRAME: 1: a zstd entry is ONE frame whatever its size (header once, last-block bit once; SURVEY 8 a14's fallback for a reader that would
                                     // not take concatenated frames -- zstd's own Decoder, which the reference uses, does); 0 (default): one frame per 1 MiB segment
    long stream_gather_wgs = 48;     // PNA_STREAM_GATHER_WGS: workgroups of the kernel that copies a batch's page-locked slabs to the device (0: one hipMemcpyAsync per slab, ~30 us each)
    long stream_overlap_mib = 64;    // PNA_STREAM_OVERLAP_MIB: while a batch runs on the device the next one is taken (and copied in beside it) only once the queue holds this much
    long stream_batch_mib = 256;     // PNA_STREAM_BATCH_MIB: input bytes one batch of the streaming facade takes at most (the queue's rest is the next batch, which is copied in meanwhile)
    long zexec_par_min_mib = 8;      // PNA_ZEXEC_PAR_MIN_MIB: zstd frames whose content takes this many MiB and more are executed in parallel by pointer jumping (0: never)
    long tab3 = 1;                   // PNA_TAB3: 1 (default): the zstd sets on the 32 / 16 KiB geometries keep their table PACKED (three 21-bit entries per 64-bit LDS word: 49 062 / 55 206 slots, lz_common.h); 0: 32-bit entries (32 704 / 36 800)
    long win32k = 1;                 // PNA_WIN32K: 1 (default): the zstd default set on the 32 KiB-window geometry of the match finder (32 704 table slots), the high set on the 16 KiB one (36 800); 0: both on 64 KiB / 24 512; 2: both on 16 KiB
    long lit_beside_seq = 1;         // PNA_LIT_BESIDE_SEQ: large zstd batches: the literal coder on a second stream next to the sequence coder
    long strong_gtab = 1;            // PNA_STRONG_GTAB: zstd levels 10 .. 22 with the match kernel's hash tables in global memory (2^19 slots per segment); 0: the LDS table
    long dev_layout = 1;             // PNA_DEV_LAYOUT: archive layout of plain one-chunk entries on the device (k_layout); 0: on the host, after a wait for the sizes
    long trace = 0;                  // PNA_TRACE: phase times of the host pipelines on stderr
    long d2h_wgs = 6;                // PNA_D2H_WGS: workgroups of the kernel that carries a sub-batch's archive bytes to the host (0: the copy engine / runtime's choice)
    long hist_by_block = -1;         // PNA_HIST_BY_BLOCK: zstd entropy stage in its per-block form (1: k_hist, k_seqa, k_seqb) or its per-segment form (0: k_stats, k_seq); -1: by batch size
};
struct TuningName { const char *name, *env; long Tuning::*field; long lo, hi; };
static const TuningName TUNING_NAMES[] = {
    {"lz_split", "PNA_LZ_SPLIT", &Tuning::lz_split, 0, 2}, {"lz_split_blocks", "PNA_LZ_SPLIT_BLOCKS", &Tuning::lz_split_blocks, 8, 1 << 17},
    {"lz_split_min", "PNA_LZ_SPLIT_MIN", &Tuning::lz_split_min, 0, 1 << 30}, {"lz_pbuf_fail", "PNA_LZ_PBUF_FAIL", &Tuning::lz_pbuf_fail, 0, 1},
    {"pipeline_chunks", "PNA_PIPELINE_CHUNKS", &Tuning::pipeline_chunks, 1, 8}, {"max_chunk_size", "PNA_MAX_CHUNK_SIZE", &Tuning::max_chunk_size, 0, 0xFFFFFFFFl},
    {"sub_mib", "PNA_SUB_MIB", &Tuning::sub_mib, 16, 16384}, {"stage_threads", "PNA_STAGE_THREADS", &Tuning::stage_threads, 0, 64},
    {"extract_win_mib", "PNA_EXTRACT_WIN_MIB", &Tuning::extract_win_mib, 1, 1 << 20}, {"batch_piece_mib", "PNA_BATCH_PIECE_MIB", &Tuning::batch_piece_mib, 0, 1 << 20},
    {"inflate_serial", "PNA_INFLATE_SERIAL", &Tuning::inflate_serial, 0, 1}, {"zdec_serial", "PNA_ZDEC_SERIAL", &Tuning::zdec_serial, 0, 1}, {"zdec_dbg", "PNA_ZDEC_DBG", &Tuning::zdec_dbg, 0, 15},
    {"blk_log", "PNA_BLK_LOG", &Tuning::blk_log, 0, PNA_BLK_LOG}, {"unit_log", "PNA_LZ_UNIT_LOG", &Tuning::unit_log, 0, 20},
    {"latency_max_mib", "PNA_LATENCY_MAX_MIB", &Tuning::latency_max_mib, 0, 1 << 20}, {"hist_by_block", "PNA_HIST_BY_BLOCK", &Tuning::hist_by_block, -1, 1},
    {"d2h_wgs", "PNA_D2H_WGS", &Tuning::d2h_wgs, 0, 4096}, {"trace", "PNA_TRACE", &Tuning::trace, 0, 1}, {"dev_layout", "PNA_DEV_LAYOUT", &Tuning::dev_layout, 0, 1}, {"strong_gtab", "PNA_STRONG_GTAB", &Tuning::strong_gtab, 0, 1}, {"win32k", "PNA_WIN32K", &Tuning::win32k, 0, 2}, {"tab3", "PNA_TAB3", &Tuning::tab3, 0, 1}, {"zexec_par_min_mib", "PNA_ZEXEC_PAR_MIN_MIB", &Tuning::zexec_par_min_mib, 0, 1 << 20}, {"stream_batch_mib", "PNA_STREAM_BATCH_MIB", &Tuning::stream_batch_mib, 1, 1 << 16}, {"stream_gather_wgs", "PNA_STREAM_GATHER_WGS", &Tuning::stream_gather_wgs, 0, 4096}, {"stream_overlap_mib", "PNA_STREAM_OVERLAP_MIB", &Tuning::stream_overlap_mib, 0, 1 << 16}, {"single_frame", "PNA_SINGLE_FRAME", &Tuning::single_frame, 0, 1}, {"lazy2", "PNA_LAZY2", &Tuning::lazy2, 0, 2}, {"tail_units", "PNA_TAIL_UNITS", &Tuning::tail_units, 0, 1}, {"lit_beside_seq", "PNA_LIT_BESIDE_SEQ", &Tuning::lit_beside_seq, 0, 1}, {"sub_ramp_down", "PNA_SUB_RAMP_DOWN", &Tuning::sub_ramp_down, 0, 1},
};

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
    bool call_tab3 = false;                         // ... its table packed (lz_common.h TAB3)
    bool call_w16 = false;                          // ... the 16 KiB window (zstd 6..9)
    bool call_gtab = false, call_w32 = false;       // ... and where the match finder's table lies / its LDS geometry (set_call_level)
    std::vector<hipEvent_t> lzm_ev; size_t lzm_used = 0;   // event pairs around the match kernel launches of the current sub-batch (timed calls)
    std::vector<uint8_t> lzm_nl;                            // launches inside each pair (2 where a run's last segments go in units)
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    DevBuf blk, tabs, seqs, lits, litc, seqc, seqw, seg_size, seg_off, stage_in, stage_out, ctab, pbuf;
    DevBuf z_big, z_one;                            // large zstd frames: their numbers, their blocks (k_zparse_a -> k_zparse<true>)
    DevBuf z_words, z_rep, z_zxf;                   // the parallel executor of large zstd frames (k_zexec_par.hip): a word per output byte, histories per block
    uint32_t zexec_par_rounds = 0;                  // pointer-jumping rounds of the latest large frame (diagnostics)
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

static int fail(pna_gpu_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
    if (c) { c->err = what; if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); } }
    return code;
}
#define HIPCHK(c, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail((c), PNA_E_HIP, #call, e__); } while (0)

extern "C" const char *pna_gpu_strerror(int code) {
    switch (code) {
        case PNA_OK: return "ok";
        case PNA_E_NODEVICE: return "no usable HIP device";
        case PNA_E_INVAL: return "invalid argument";
        case PNA_E_NOMEM: return "out of memory";
        case PNA_E_DSTSIZE: return "destination too small";
        case PNA_E_HIP: return "HIP error";
        case PNA_E_SINK: return "sink callback failed";
        case PNA_E_UNSUPPORTED: return "algorithm not supported by this build";
        default: return "unknown error";
    }
}
extern "C" const char *pna_gpu_last_error(const pna_gpu_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int pna_gpu_init(pna_gpu_ctx **out, int device_id, uint32_t flags) {
    if (!out) return PNA_E_INVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PNA_E_NODEVICE;
    if (device_id < 0 || device_id >= n) return PNA_E_INVAL;
    if (hipSetDevice(device_id) != hipSuccess) return PNA_E_NODEVICE;
    pna_gpu_ctx *c = new pna_gpu_ctx();
    c->device = device_id;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->n_cus = (uint32_t)cus; }
    c->flags = (flags & PNA_F_DEFAULT) ? (F_HUF | F_FSE | F_LAZY | F_FAR | F_ADOPT | F_INS2) : (flags & 0xFF);
    c->flags &= ~F_REP;                    // repeat-offset codes are not produced by this build
    for (const TuningName &t : TUNING_NAMES)
        if (const char *e = getenv(t.env)) { const long v = atol(e); if (v >= t.lo && v <= t.hi) c->tun.*(t.field) = v; }
    if (const char *pm = getenv("PNA_STREAM_POOL_MIB")) c->pool_cap = (size_t)std::min<unsigned long>(strtoul(pm, nullptr, 10), 1ul << 20) << 20;
    if (const char *lg = getenv("PNA_STREAM_LINGER_US")) c->comb_linger_us = (uint32_t)std::min<unsigned long>(strtoul(lg, nullptr, 10), 100000ul);
    if (!(flags & PNA_F_DEFAULT)) c->flags |= flags & 0x3F00u;  // diagnostics: 0x100 phase stamps, 0x200 force the serial fallback in k_lz, 0x1000 / 0x2000 force the one-kernel / two-phase sequence coder
    c->flags |= flags & (PNA_F_LZ_FUSED | PNA_F_LZ_WAVEPARSE);  // the form of the LZ stage: honoured next to PNA_F_DEFAULT as well
    c->call_flags = c->flags;
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return PNA_E_NODEVICE; }
    for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return PNA_E_NODEVICE; }
    *out = c;
    return PNA_OK;
}

extern "C" int pna_gpu_set_option(pna_gpu_ctx *c, const char *name, long value) {
    if (!c || !name) return PNA_E_INVAL;
    if (!strcmp(name, "stream_pool_mib")) { if (value < 0) return PNA_E_INVAL; c->pool_cap = (size_t)std::min<long>(value, 1l << 20) << 20; return PNA_OK; }
    if (!strcmp(name, "stream_linger_us")) { c->comb_linger_us = value < 0 ? 0xFFFFFFFFu : (uint32_t)std::min<long>(value, 100000); return PNA_OK; }
    for (const TuningName &t : TUNING_NAMES)
        if (!strcmp(name, t.name)) {
            if (value < t.lo || value > t.hi) return fail(c, PNA_E_INVAL, "option value out of range");
            if (t.field == &Tuning::blk_log && value != 0 && value < (long)BLK_LOG_MIN) return fail(c, PNA_E_INVAL, "blk_log: 0 or 13..17");
            c->tun.*(t.field) = value; return PNA_OK;
        }
    return fail(c, PNA_E_INVAL, "unknown option");
}

extern "C" void pna_gpu_shutdown(pna_gpu_ctx *c) {
    if (c) { for (void *a : c->pool_arenas) (void)hipHostFree(a); c->pool_arenas.clear(); c->pool_free.clear(); for (auto &b : c->s_out) b.release(); for (auto &b : c->st_in) b.release(); for (auto &b : c->st_out) b.release(); for (auto &b : c->s_segs) b.release();
             if (c->s_h2d) (void)hipStreamDestroy(c->s_h2d); if (c->s_d2h) (void)hipStreamDestroy(c->s_d2h); for (auto &e : c->s_ev) if (e) (void)hipEventDestroy(e); }
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->plan, &c->d_tail, &c->blk, &c->tabs, &c->seqs, &c->lits, &c->litc, &c->seqc, &c->seqw, &c->pbuf, &c->seg_size,
                      &c->seg_off, &c->stage_in, &c->stage_out, &c->z_words, &c->z_rep, &c->z_zxf, &c->z_big, &c->z_one, &c->ctab, &c->c_vocab, &c->c_cum, &c->c_phr,
                      &c->gtab, &c->fr_desc, &c->fr_blob, &c->fr_segdst, &c->fr_entoff, &c->crc_tabs, &c->aes_tabs, &c->ci_units, &c->ci_ivs, &c->ci_keys, &c->ci_gcm, &c->ci_spread, &c->ci_spread_desc, &c->z_vp, &c->z_pb, &c->z_mode, &c->x_arc, &c->x_pk, &c->x_raw[0], &c->x_raw[1], &c->x_desc, &c->x_place, &c->x_flag, &c->x_tags, &c->x_plen, &c->aes_dtabs, &c->solid_plain, &c->solid_desc, &c->solid_blob, &c->solid_place, &c->z_ents, &c->z_frames, &c->z_lit, &c->z_fx, &c->z_blocks, &c->z_tabs, &c->z_seqs, &c->z_hlist, &c->z_slist, &c->z_work, &c->z_fb, &c->z_cbase, &c->z_apart}) b->release();
    for (auto &b : c->lent) (void)hipHostFree((void *)b.first);          // (buffers the host never gave back)
    c->lent.clear();
    for (PinBuf *b : {&c->h_entoff, &c->h_plan, &c->h_tail, &c->h_desc, &c->h_blob, &c->h_segdst, &c->h_segoff, &c->hp_in[0], &c->hp_in[1], &c->hp_in[2], &c->hp_in[3], &c->hp_out[0], &c->hp_out[1]}) b->release();
    for (DevBuf *b : {&c->dp_in[0], &c->dp_in[1], &c->dp_in[2], &c->dp_in[3], &c->dp_out[0], &c->dp_out[1]}) b->release();
    for (auto &e : c->ev_in) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_out) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_lz) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->lzm_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_ci) if (e) (void)hipEventDestroy(e);
    for (auto &r : c->ev_en) for (auto &e : r) if (e) (void)hipEventDestroy(e);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->cp_in) (void)hipStreamDestroy(c->cp_in);
    if (c->cp_out) (void)hipStreamDestroy(c->cp_out);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" size_t pna_gpu_bound(int algo, size_t n) {
    if (algo == PNA_ALGO_STORE) return n;
    // (per block of the SMALLEST size a batch may be cut into -- latency mode, BLK_LOG_MIN --: deflate 5 bytes of stored-block header + 5 of sync flush,
    // zstd 3 bytes of block header; per segment / 128 KiB the older, coarser allowances)
    constexpr size_t BMIN = (size_t)1 << BLK_LOG_MIN;
    if (algo == PNA_ALGO_DEFLATE) { size_t b = (n + BMIN - 1) / BMIN; if (!b) b = 1; return n + b * 10 + (n >> PNA_BLK_LOG) * 16 + 64; }
    size_t segs = (n + SEG_SIZE - 1) / SEG_SIZE; if (segs == 0) segs = 1;
    size_t blks = (n + BMIN - 1) / BMIN + segs;
    return n + segs * 6 + blks * 3 + 16;
}

extern "C" int pna_gpu_clamp_level(int algo, int level) {
    if (algo == PNA_ALGO_ZSTD) {            // lib/src/compress/zstandard.rs:13,43-57 (min_c_level .. 22, default 3)
        if (level == PNA_LEVEL_DEFAULT) return 3;
        if (level < -131072) return -131072;
        return level > 22 ? 22 : level;
    }
    if (algo == PNA_ALGO_DEFLATE) {         // lib/src/compress/deflate.rs:33-38,89-101
        if (level == PNA_LEVEL_DEFAULT) return 6;
        // Custom(n) => Compression::new((n as u32).clamp(0, 9)): a negative n wraps to a large u32 and clamps to 9 (deflate.rs:89-101)
        return (uint32_t)level > 9u ? 9 : level;
    }
    return 0;
}

// Parameter sets behind the reference's level scale (CompressionLevel -> ZstdCompressionLevel / flate2::Compression,
// lib/src/compress/zstandard.rs:43-57, deflate.rs:89-101; PNA_LEVEL_DEFAULT and zstd level 0 = the default):
//   stored    deflate 0                      Compression::none(): stored blocks only (no match finder, header 78 01)
//   fast      zstd < 0 and 1, deflate 1..3   every position in the table, look-back = the LDS window, no backward adoption; lazy deferral as below
//   light     zstd 2,         deflate 4..8   + even-position table, backward adoption (two rounds), 1 MiB look-back (zstd), lazy deferral over three positions
//   default   zstd 0, 3..5                   + a third adoption round (matches move back by up to 7 positions): ratio at the reference's default level (round 4)
//   high      zstd 6..9,      deflate 9      the default set on the 16 KiB-window geometry (more table slots, more candidates verified in HBM / L2); deflate: + third round
//   max       zstd 10..22                    + the match kernel's hash table in global memory: 2^19 slots per segment instead of what LDS holds
// zstd light / default run the match finder's 32 KiB-window geometry, high the 16 KiB one (lz_common.h LzGeo), both with the PACKED table (three 21-bit
// entries per 64-bit LDS word: 49 062 / 55 206 slots; option tab3 = 0: 32-bit entries, 32 704 / 36 800); the others and deflate the 64 KiB geometry (24 512).
static uint32_t level_flags(const pna_gpu_ctx *c, int algo, int level) {
    const int lv = pna_gpu_clamp_level(algo, level);
    const bool fast = algo == PNA_ALGO_DEFLATE ? lv <= 3 : (lv < 0 || lv == 1);
    const bool balanced = false;   // (the set without lazy deferral -- zstd 2, deflate 4..5 until round 3 -- is as fast as the default set since the parse kernel looks ahead for free; the levels take the default set, the bits remain)
    const bool strong = algo == PNA_ALGO_DEFLATE ? lv >= 9 : (lv >= 3 || lv == 0);      // (zstd 0 = the default = 3; round 4: the third adoption round from the default level on)
    if (fast) return c->flags & ~(F_FAR | F_ADOPT | F_INS2 | F_STRONG);      // (lazy deferral stays: the parse kernel does it for free -- zstd-1 2.510 -> 2.54 at the same speed)
    if (balanced) return c->flags & ~(F_LAZY | F_STRONG);
    if (strong && (c->flags & F_ADOPT) && (c->flags & F_LAZY)) return c->flags | F_STRONG;
    return c->flags;
}
static void set_call_level(pna_gpu_ctx *c, int algo, int level) {
    c->call_flags = level_flags(c, algo, level);
    const bool zstd = algo != PNA_ALGO_DEFLATE;
    c->call_gtab = zstd && pna_gpu_clamp_level(algo, level) >= 10 && (c->call_flags & F_STRONG) && (c->call_flags & F_ADOPT) && c->tun.strong_gtab != 0;
    c->call_w32 = zstd && !c->call_gtab && (c->call_flags & F_FAR) && (c->call_flags & F_LAZY) && c->tun.win32k != 0;
    c->call_w16 = c->call_w32 && (c->tun.win32k >= 2 || pna_gpu_clamp_level(algo, level) >= 6);          // the high set's geometry (round 4: chosen by the level, no longer by F_STRONG)
    c->call_stored = !zstd && pna_gpu_clamp_level(algo, level) == 0;
    c->call_tab3 = c->call_w32 && (c->call_flags & F_INS2) && (c->call_flags & F_ADOPT) && c->tun.tab3 != 0;
    c->call_lazy2 = (c->call_flags & F_LAZY) && (c->tun.lazy2 != 0 || (c->call_flags & F_STRONG));   // every lazy set defers over two positions (the high sets always did)
    c->call_lazy3 = c->call_lazy2 && c->tun.lazy2 >= 2;                                               // ... and over three (option lazy2 = 2, the default)
}

// blocks an entry of `len` bytes takes in the per-block workspace (sub-batches are cut by block count); a forced block size counts as such
static size_t plan_blocks(const pna_gpu_ctx *c, uint64_t len) {
    return (size_t)((len + ((uint64_t)1 << c->plan_log) - 1) >> c->plan_log);
}
// Block size of a batch whose entries are all small: the per-block arrays have the block size as their stride, so a batch of 4 KiB entries on 128 KiB
// blocks would spend 32 times the memory (and a sub-batch per 131 072 entries) that 8 KiB blocks need.  Entries of up to 64 KiB: the power of two that
// holds the largest (>= 8 KiB); anything larger: 128 KiB.  (Latency mode, for small batches of large entries, chooses on top of this in run_subbatch.)
template <class L>
static uint32_t small_entry_blk_log(const pna_gpu_ctx *c, const L *src_len, size_t e0, size_t e1) {
    if (c->tun.blk_log) return (uint32_t)c->tun.blk_log;
    uint64_t mx = 0;
    for (size_t e = e0; e < e1; e++) mx = std::max<uint64_t>(mx, src_len[e]);
    if (mx > 65536) return PNA_BLK_LOG;
    uint32_t lg = BLK_LOG_MIN;
    while (((uint64_t)1 << lg) < mx) lg++;
    return lg;
}
// per call: the block size the sub-batches are cut with and how many blocks fit the workspace budget
template <class L>
static void plan_call(pna_gpu_ctx *c, const L *src_len, size_t n) {
    c->plan_log = small_entry_blk_log(c, src_len, 0, n);
    const uint64_t per_block = (uint64_t)seq_cap_of(c->plan_log) * 16 + ((uint64_t)3 << c->plan_log) + 64;
    c->max_blocks = (size_t)std::max<uint64_t>(1024, (96ull << 30) / per_block);
}

extern "C" int pna_gpu_last_timing(const pna_gpu_ctx *c, pna_gpu_timing *out) {
    if (!c || !out) return PNA_E_INVAL;
    *out = c->timing; out->blk_log = c->last_blk_log; out->lz_units = c->last_units; return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// One sub-batch: entries [e0, e1) -> segments -> kernels; output appended at d_dst + out_base.
// ---- CRC-32 tables of the framing kernel (k_frame.hip explains the algebra)
static uint32_t gf2_mulmod(uint32_t a, uint32_t b) {          // a * b mod P, reflected bit order (bit 31 = x^0)
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) { if (a & 0x80000000u) p ^= b; a <<= 1; b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
    return p;
}
static uint32_t gf2_xpow(uint64_t e) {                        // x^e mod P
    uint32_t r = 0x80000000u, base = 0x40000000u;
    while (e) { if (e & 1) r = gf2_mulmod(base, r); base = gf2_mulmod(base, base); e >>= 1; }
    return r;
}
static void build_crc_tabs(CrcTabs &t) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t v = i; for (int k = 0; k < 8; k++) v = (v & 1) ? (v >> 1) ^ 0xEDB88320u : v >> 1; t.T[0][i] = v; }
    for (int k = 1; k < 4; k++) for (uint32_t i = 0; i < 256; i++) t.T[k][i] = (t.T[k - 1][i] >> 8) ^ t.T[0][t.T[k - 1][i] & 0xFF];
    const uint32_t z = gf2_xpow(8ull * 16320);
    for (int j = 0; j < 4; j++) for (uint32_t b = 0; b < 256; b++) t.Z[j][b] = gf2_mulmod(z, b << (8 * j));
    for (int j = 0; j < 8; j++) t.sh[j] = gf2_xpow(8ull * 64 << j);
    for (int m = 1; m <= 4; m++) for (uint32_t k = 0; k < 64; k++) t.pw[m - 1][k] = gf2_xpow(8ull * 64 * m * k);
}
static int ensure_crc(pna_gpu_ctx *c) {
    if (c->crc_ready) return PNA_OK;
    CrcTabs t; build_crc_tabs(t);
    if (c->crc_tabs.ensure(sizeof(t))) return fail(c, PNA_E_NOMEM, "crc tables");
    HIPCHK(c, hipMemcpy(c->crc_tabs.p, &t, sizeof(t), hipMemcpyHostToDevice));
    c->crc_ready = true;
    return PNA_OK;
}

// Host walk through k_frame's CRC schedule with the same tables (lane states, Z_16320 between tiles, fold tree): lets the
// CPU-only test suite check the algebra and the table construction against pna_crc32 without a GPU.  Not a product path.
extern "C" uint32_t pna_gpu_debug_crc_schedule(const void *payload, size_t len) {
    static const CrcTabs t = [] { CrcTabs x; build_crc_tabs(x); return x; }();      // initialised once, thread-safe (C++11 static)
    const uint8_t *pl = (const uint8_t *)payload;
    const uint64_t n = 4 + (uint64_t)len, ntile = (n + 16383) / 16384, pad = ntile * 16384 - n;
    static const uint8_t ty[4] = {0x46 ^ 0xFF, 0x44 ^ 0xFF, 0x41 ^ 0xFF, 0x54 ^ 0xFF};
    uint32_t lane[256] = {0};
    for (uint64_t k = 0; k < ntile; k++)
        for (uint32_t l = 0; l < 256; l++) {
            uint32_t s = lane[l];
            s = t.Z[0][s & 0xFF] ^ t.Z[1][(s >> 8) & 0xFF] ^ t.Z[2][(s >> 16) & 0xFF] ^ t.Z[3][s >> 24];
            for (uint32_t j = 0; j < 16; j++) {
                uint32_t w = 0;
                for (uint32_t b = 0; b < 4; b++) {
                    const uint64_t p = k * 16384 + l * 64 + 4 * j + b;
                    const uint8_t v = p < pad ? 0 : (p < pad + 4 ? ty[p - pad] : pl[p - pad - 4]);
                    w |= (uint32_t)v << (8 * b);
                }
                const uint32_t x = s ^ w;
                s = t.T[3][x & 0xFF] ^ t.T[2][(x >> 8) & 0xFF] ^ t.T[1][(x >> 16) & 0xFF] ^ t.T[0][x >> 24];
            }
            lane[l] = s;
        }
    for (uint32_t j = 0; j < 8; j++) {
        const uint32_t st = 1u << j;
        for (uint32_t l = 0; l < 256; l += 2 * st) lane[l] = gf2_mulmod(t.sh[j], lane[l]) ^ lane[l + st];
    }
    return ~lane[0];
}

// ---- cipher stage: AES-256 key schedule and round tables (FIPS-197), built on the host once per context / per call
static void aes_sbox(uint8_t sb[256]) {
    uint8_t p = 1, q = 1;                                      // p walks the multiplicative group of GF(2^8), q is its inverse
    do {
        p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1B : 0));
        q ^= (uint8_t)(q << 1); q ^= (uint8_t)(q << 2); q ^= (uint8_t)(q << 4); if (q & 0x80) q ^= 0x09;
        const uint8_t r1 = (uint8_t)((q << 1) | (q >> 7)), r2 = (uint8_t)((q << 2) | (q >> 6)), r3 = (uint8_t)((q << 3) | (q >> 5)), r4 = (uint8_t)((q << 4) | (q >> 4));
        sb[p] = (uint8_t)(q ^ r1 ^ r2 ^ r3 ^ r4 ^ 0x63);
    } while (p != 1);
    sb[0] = 0x63;
}
static void build_aes_tabs(AesTabs &t) {
    uint8_t sb[256]; aes_sbox(sb);
    for (uint32_t x = 0; x < 256; x++) {
        const uint32_t s1 = sb[x], s2 = ((s1 << 1) ^ ((s1 & 0x80) ? 0x1B : 0)) & 0xFF, s3 = s2 ^ s1;
        const uint32_t w = s2 | (s1 << 8) | (s1 << 16) | (s3 << 24);
        t.Te[0][x] = w; t.Te[1][x] = (w << 8) | (w >> 24); t.Te[2][x] = (w << 16) | (w >> 16); t.Te[3][x] = (w << 24) | (w >> 8);
    }
}
static void aes256_expand(const uint8_t key[32], AesKey &k) {
    uint8_t sb[256]; aes_sbox(sb);
    uint8_t w[60][4]; uint8_t rc = 1;
    memcpy(w, key, 32);
    for (int i = 8; i < 60; i++) {
        uint8_t t[4]; memcpy(t, w[i - 1], 4);
        if (i % 8 == 0) { const uint8_t t0 = t[0]; t[0] = (uint8_t)(sb[t[1]] ^ rc); t[1] = sb[t[2]]; t[2] = sb[t[3]]; t[3] = sb[t0]; rc = (uint8_t)((rc << 1) ^ ((rc & 0x80) ? 0x1B : 0)); }
        else if (i % 8 == 4) for (int j = 0; j < 4; j++) t[j] = sb[t[j]];
        for (int j = 0; j < 4; j++) w[i][j] = (uint8_t)(w[i - 8][j] ^ t[j]);
    }
    for (int i = 0; i < 60; i++) k.rk[i] = (uint32_t)w[i][0] | ((uint32_t)w[i][1] << 8) | ((uint32_t)w[i][2] << 16) | ((uint32_t)w[i][3] << 24);
}
static uint8_t gf_mul8(uint8_t a, uint8_t b) { uint8_t r = 0; while (b) { if (b & 1) r ^= a; a = (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1B : 0)); b >>= 1; } return r; }
// equivalent inverse cipher (FIPS-197 5.3.5): tables of InvSubBytes + InvMixColumns, round keys reversed with InvMixColumns applied to
// the middle ones
static void build_aes_dec_tabs(AesDecTabs &t) {
    uint8_t sb[256], si[256]; aes_sbox(sb);
    for (int x = 0; x < 256; x++) si[sb[x]] = (uint8_t)x;
    for (uint32_t x = 0; x < 256; x++) {
        const uint8_t v = si[x];
        const uint32_t w = (uint32_t)gf_mul8(v, 14) | ((uint32_t)gf_mul8(v, 9) << 8) | ((uint32_t)gf_mul8(v, 13) << 16) | ((uint32_t)gf_mul8(v, 11) << 24);
        t.Td[0][x] = w; t.Td[1][x] = (w << 8) | (w >> 24); t.Td[2][x] = (w << 16) | (w >> 16); t.Td[3][x] = (w << 24) | (w >> 8);
        t.Sd[x] = v;
    }
}
static void aes256_dec_key(const AesKey &k, AesKey &d) {
    for (int c4 = 0; c4 < 4; c4++) { d.rk[c4] = k.rk[56 + c4]; d.rk[56 + c4] = k.rk[c4]; }
    for (int r = 1; r < 14; r++)
        for (int c4 = 0; c4 < 4; c4++) {
            const uint32_t w = k.rk[4 * (14 - r) + c4];
            const uint8_t a0 = (uint8_t)w, a1 = (uint8_t)(w >> 8), a2 = (uint8_t)(w >> 16), a3 = (uint8_t)(w >> 24);
            const uint8_t r0 = gf_mul8(a0, 14) ^ gf_mul8(a1, 11) ^ gf_mul8(a2, 13) ^ gf_mul8(a3, 9), r1 = gf_mul8(a0, 9) ^ gf_mul8(a1, 14) ^ gf_mul8(a2, 11) ^ gf_mul8(a3, 13);
            const uint8_t r2 = gf_mul8(a0, 13) ^ gf_mul8(a1, 9) ^ gf_mul8(a2, 14) ^ gf_mul8(a3, 11), r3 = gf_mul8(a0, 11) ^ gf_mul8(a1, 13) ^ gf_mul8(a2, 9) ^ gf_mul8(a3, 14);
            d.rk[4 * r + c4] = (uint32_t)r0 | ((uint32_t)r1 << 8) | ((uint32_t)r2 << 16) | ((uint32_t)r3 << 24);
        }
}
static int ensure_aes_dec(pna_gpu_ctx *c) {
    if (c->aes_dec_ready) return PNA_OK;
    static AesDecTabs t; build_aes_dec_tabs(t);
    if (c->aes_dtabs.ensure(sizeof(t))) return fail(c, PNA_E_NOMEM, "aes tables");
    HIPCHK(c, hipMemcpy(c->aes_dtabs.p, &t, sizeof(t), hipMemcpyHostToDevice));
    c->aes_dec_ready = true;
    return PNA_OK;
}
static int ensure_aes(pna_gpu_ctx *c) {
    if (c->aes_ready) return PNA_OK;
    AesTabs t; build_aes_tabs(t);
    if (c->aes_tabs.ensure(sizeof(t))) return fail(c, PNA_E_NOMEM, "aes tables");
    HIPCHK(c, hipMemcpy(c->aes_tabs.p, &t, sizeof(t), hipMemcpyHostToDevice));
    for (auto &e : c->ev_ci) HIPCHK(c, hipEventCreate(&e));
    c->aes_ready = true;
    return PNA_OK;
}
static int check_cipher(pna_gpu_ctx *c, const pna_gpu_cipher *ci) {
    if (ci->encryption == PNA_ENC_CAMELLIA) return fail(c, PNA_E_UNSUPPORTED, "Camellia is not offered on the device path");
    if (ci->encryption != PNA_ENC_AES) return fail(c, PNA_E_INVAL, "unknown encryption");
    if (ci->cipher_mode != PNA_MODE_CTR && ci->cipher_mode != PNA_MODE_CBC && ci->cipher_mode != PNA_MODE_GCM) return fail(c, PNA_E_UNSUPPORTED, "cipher mode not offered on the device path");
    if (ci->cipher_mode == PNA_MODE_GCM && ci->gcm_segment_size > (64u << 20)) return fail(c, PNA_E_INVAL, "GCM segment size beyond 64 MiB");
    return PNA_OK;
}
// one AES-256 block on the host (FIPS-197 with the round tables of the kernels): hash subkey and E(K, J0) of a GCM segment
static void aes256_block_host(const AesKey &k, const uint8_t in[16], uint8_t out[16]) {
    static const AesTabs T = [] { AesTabs x; build_aes_tabs(x); return x; }();      // called from several host threads: C++11 static initialisation is thread-safe
    uint32_t s[4], t[4];
    for (int i = 0; i < 4; i++) s[i] = ((uint32_t)in[4 * i] | ((uint32_t)in[4 * i + 1] << 8) | ((uint32_t)in[4 * i + 2] << 16) | ((uint32_t)in[4 * i + 3] << 24)) ^ k.rk[i];
    for (int r = 1; r < 14; r++) {
        for (int i = 0; i < 4; i++)
            t[i] = T.Te[0][s[i] & 0xFF] ^ T.Te[1][(s[(i + 1) & 3] >> 8) & 0xFF] ^ T.Te[2][(s[(i + 2) & 3] >> 16) & 0xFF] ^ T.Te[3][s[(i + 3) & 3] >> 24] ^ k.rk[4 * r + i];
        memcpy(s, t, sizeof s);
    }
    auto sb = [&](uint32_t v) { return (T.Te[0][v & 0xFF] >> 8) & 0xFF; };
    for (int i = 0; i < 4; i++)
        t[i] = (sb(s[i]) | (sb(s[(i + 1) & 3] >> 8) << 8) | (sb(s[(i + 2) & 3] >> 16) << 16) | (sb(s[(i + 3) & 3] >> 24) << 24)) ^ k.rk[56 + i];
    for (int i = 0; i < 4; i++) { out[4 * i] = (uint8_t)t[i]; out[4 * i + 1] = (uint8_t)(t[i] >> 8); out[4 * i + 2] = (uint8_t)(t[i] >> 16); out[4 * i + 3] = (uint8_t)(t[i] >> 24); }
}
// GCM STREAM material of one entry (lib/src/entry/write.rs:81-107 to_hashed; lib/src/cipher/aead.rs): stream header, stream key bound to
// the FHED chunk and the PHSF string, round keys, hash subkey, E(K, J0) of the (single, final) segment 0 and its first counter block.
struct GcmMaterial { uint8_t header[75]; AesKey rk; uint32_t h[4], ej0[4]; uint8_t ctr_iv[16]; };
// (the stream key is bound to the header chunk of the entry that carries the stream: FHED of a normal entry, SHED of a solid one -- entry_context,
// lib/src/cipher/aead.rs:167-190; name == nullptr: the solid entry's SHED)
static void gcm_entry_material(const pna_gpu_cipher *ci, const uint8_t kc[32], const uint8_t phsf_hash[32], const uint8_t salt_prefix[39],
                               uint32_t seg_size, const char *name, int compression, GcmMaterial &m) {
    memcpy(m.header, salt_prefix, 39);
    m.header[39] = (uint8_t)(seg_size >> 24); m.header[40] = (uint8_t)(seg_size >> 16); m.header[41] = (uint8_t)(seg_size >> 8); m.header[42] = (uint8_t)seg_size;
    memcpy(m.header + 43, kc, 32);
    const std::vector<uint8_t> fh = name ? frame_fhed_bytes(name, compression, ci->encryption, PNA_MODE_GCM)
                                         : std::vector<uint8_t>{0, 0, (uint8_t)compression, (uint8_t)ci->encryption, (uint8_t)PNA_MODE_GCM};
    uint8_t info[88];
    memcpy(info, "PNA-STREAM-v1", 13);
    sha256_bytes(name ? "FHED" : "SHED", 4, fh.data(), fh.size(), info + 13);
    memcpy(info + 45, phsf_hash, 32);
    memcpy(info + 77, salt_prefix + 32, 7);
    memcpy(info + 84, m.header + 39, 4);
    uint8_t ks[32];
    hkdf_sha256_32(ci->key, 32, salt_prefix, 32, info, 88, ks);
    aes256_expand(ks, m.rk);
    uint8_t zero[16] = {0}, hb[16], j0[16], eb[16];
    aes256_block_host(m.rk, zero, hb);
    memcpy(j0, salt_prefix + 32, 7); j0[7] = j0[8] = j0[9] = j0[10] = 0; j0[11] = 1;      // segment_nonce(prefix, 0, final)
    j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
    aes256_block_host(m.rk, j0, eb);
    for (int i = 0; i < 4; i++) {
        m.h[i] = ((uint32_t)hb[4 * i] << 24) | ((uint32_t)hb[4 * i + 1] << 16) | ((uint32_t)hb[4 * i + 2] << 8) | hb[4 * i + 3];
        m.ej0[i] = ((uint32_t)eb[4 * i] << 24) | ((uint32_t)eb[4 * i + 1] << 16) | ((uint32_t)eb[4 * i + 2] << 8) | eb[4 * i + 3];
    }
    memcpy(m.ctr_iv, j0, 16); m.ctr_iv[15] = 2;                                           // first data block: counter 2
}
// the per-entry IVs of a cipher job: the caller's, or random ones (random::random_vec(block_size) per entry, lib/src/entry/write.rs:108-112)
static int resolve_ivs(pna_gpu_ctx *c, const pna_gpu_cipher *cipher, size_t n, std::vector<uint8_t> &own, const uint8_t **ivs) {
    int rc = check_cipher(c, cipher); if (rc) return rc;
    if (!cipher->phsf) return fail(c, PNA_E_INVAL, "cipher without a PHSF string");
    *ivs = cipher->ivs;
    if (*ivs) return PNA_OK;
    const size_t per = cipher->cipher_mode == PNA_MODE_GCM ? 39 : 16;  // GCM: salt(32) || nonce_prefix(7) per stream (to_hashed, write.rs:83-86)
    own.resize(n * per + 16);
    for (size_t o = 0; o < n * per;) {
        const ssize_t got = getrandom(own.data() + o, std::min<size_t>(n * per - o, 1u << 20), 0);
        if (got <= 0) return fail(c, PNA_E_INVAL, "getrandom failed");
        o += (size_t)got;
    }
    *ivs = own.data();
    return PNA_OK;
}
constexpr uint64_t CTR_UNIT = 256u << 10;                    // bytes of one CTR work unit (one workgroup)

// names[e] for the batch's global entry index e; solid: one SDAT chunk per segment of the (single) entry; cipher + ivs (16 bytes per
// global entry index): the payloads are encrypted in place before their CRC-32 is taken
struct PlaceDescH { uint64_t src_off, dst_off; uint32_t len, pad; };   // = PlaceDesc of k_frame.hip (k_place / k_gather)
struct FrameJob { const char *const *names; int solid; const pna_gpu_cipher *cipher = nullptr; const uint8_t *ivs = nullptr; const pna_gpu_entry_meta *meta = nullptr;
                  uint32_t max_chunk = 0; bool want_offsets = true; };   // FDAT chunks of at most this many bytes (FlattenWriter::max_chunk_size; 0 = the reference's default u32::MAX)
// largest FDAT chunk the device paths write: the CRC kernel takes "FDAT" || data as one message of at most 2^32 - 1 bytes (the reference's default cuts
// at u32::MAX: the same chunks unless an entry's compressed payload exceeds 4 GiB - 5 bytes)
static uint64_t chunk_limit(uint32_t max_chunk) { return max_chunk ? std::min<uint64_t>(max_chunk, 0xFFFFFFFBull) : 0xFFFFFFFBull; }
static size_t meta_len(const pna_gpu_entry_meta *m, size_t e) {
    if (!m) return 0;
    return (m->extra && m->extra_len ? m->extra_len[e] : 0) + (m->facets && m->facets_len ? m->facets_len[e] : 0);
}
// a blob of already framed chunks: lengths consistent, CRCs right, none of the chunk types this library writes itself
static bool meta_blob_ok(const uint8_t *p, size_t n) {
    size_t pos = 0;
    while (pos < n) {
        if (n - pos < 12) return false;
        const uint32_t dl = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3];
        if (n - pos - 12 < dl) return false;
        const uint8_t *ty = p + pos + 4;
        for (const char *own : {"FHED", "FDAT", "FEND", "fSIZ", "PHSF", "SHED", "SDAT", "SEND", "AHED", "AEND", "ANXT"}) if (memcmp(ty, own, 4) == 0) return false;
        const uint8_t *cp = p + pos + 8 + dl;
        if (pna_crc32(0, ty, 4 + (size_t)dl) != (((uint32_t)cp[0] << 24) | ((uint32_t)cp[1] << 16) | ((uint32_t)cp[2] << 8) | cp[3])) return false;
        pos += 12 + (size_t)dl;
    }
    return true;
}
// prefix = FHED | fSIZ | rest  ->  FHED | extra | fSIZ | facets | rest   (NormalEntry::write_chunks_to, lib/src/entry.rs:895-911)
static void splice_meta(std::vector<uint8_t> &pre, const pna_gpu_entry_meta *m, size_t e) {
    if (!meta_len(m, e)) return;
    const size_t c0 = 12 + ((size_t)pre[0] << 24 | (size_t)pre[1] << 16 | (size_t)pre[2] << 8 | pre[3]);
    const size_t c1 = 12 + ((size_t)pre[c0] << 24 | (size_t)pre[c0 + 1] << 16 | (size_t)pre[c0 + 2] << 8 | pre[c0 + 3]);
    std::vector<uint8_t> out(pre.begin(), pre.begin() + c0);
    if (m->extra && m->extra_len && m->extra_len[e]) out.insert(out.end(), (const uint8_t *)m->extra[e], (const uint8_t *)m->extra[e] + m->extra_len[e]);
    out.insert(out.end(), pre.begin() + c0, pre.begin() + c0 + c1);
    if (m->facets && m->facets_len && m->facets_len[e]) out.insert(out.end(), (const uint8_t *)m->facets[e], (const uint8_t *)m->facets[e] + m->facets_len[e]);
    out.insert(out.end(), pre.begin() + c0 + c1, pre.end());
    pre.swap(out);
}

// The LZ stage over segments [s0, s1) of a sub-batch.  Runs of more than 1 024 segments take the split form -- match kernel (k_lzm) + parse
// kernel (k_lzp) per run of at most `split_blocks` blocks, which meet in c->pbuf (4 bytes per input byte of the run; one run at a time on the
// stream, so runs share it) --, shorter ones and PNA_F_LZ_FUSED / PNA_LZ_SPLIT=0 the one-kernel form (k_lz<MODE 0>, no pbuf);
// PNA_F_LZ_WAVEPARSE / PNA_LZ_SPLIT=2: the split form as k_lz<MODE 1> + k_lz<MODE 2> (testing).  All forms give the same bytes.  If pbuf
// cannot be had, the run is halved down to 1 024 blocks, then fused.
static int lz_stage(pna_gpu_ctx *c, const uint8_t *d_src, const SegDesc *segs, uint32_t nseg_all, uint32_t s0, uint32_t s1, uint32_t nblk, uint4 *ctab,
                    uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, bool timed) {
    const int env_split = (int)c->tun.lz_split;
    // (the option counts blocks of 128 KiB: a run is that many BYTES of input whatever the batch's block size)
    const uint32_t env_blocks = (uint32_t)std::min<uint64_t>((uint64_t)c->tun.lz_split_blocks << (PNA_BLK_LOG - segs[s0].blk_log), 1u << 30);
    const uint32_t bps = 1u << (20 - segs[s0].blk_log);          // blocks of a full segment
    // zstd levels 10 .. 22 (the strong set): the match kernel's hash tables lie in global memory (2^19 slots per segment instead of the 24 512 LDS
    // holds; k_lz_split.hip) -- only k_lzm has that form, so those levels always take the split form, whatever the run's length
    const bool gt = !ctab && c->call_gtab;
    const bool fused = !gt && ((c->call_flags & PNA_F_LZ_FUSED) || env_split == 0 || (flags & 0x100u));   // (0x100: the phase stamps live in the fused kernel)
    const bool waveparse = !gt && ((c->call_flags & PNA_F_LZ_WAVEPARSE) || env_split == 2);
    uint32_t split_blocks = env_blocks;
    if (s1 > s0) {
        // several runs: of about equal size (whole rounds of 256 one-MiB segments) instead of full ones and a short tail -- a tail under the
        // split form's threshold would fall back to the slower one-kernel form (5 000 segments: 2 560 + 2 440 instead of 4 096 + 904)
        const uint32_t total = (s1 < nseg_all ? segs[s1].blk_base : nblk) - segs[s0].blk_base;
        const uint32_t nruns = (total + split_blocks - 1) / split_blocks;
        if (nruns > 1) {
            const uint32_t round = c->n_cus * bps, even = ((total + nruns - 1) / nruns + round - 1) / round * round;
            if (even < split_blocks) split_blocks = even;
        }
    }
    const uint32_t s1_all = s1; bool fused_tail = false;
    // The split form at every size: its parse kernel runs one wave per BLOCK (a block's parse depends on nothing outside the block), so a short run no
    // longer waits for one wave's walk over a whole segment (N x 1 MiB, one-kernel / split: 32: 2.3 / 1.6 ms, 256: 2.4 / 1.8, 1 024: 9.3 / 6.3,
    // 2 048: 18.6 / 12.3; scripts/lz_forms.py).  The one-kernel form remains for PNA_F_LZ_FUSED, for a workspace that cannot be allocated and for the
    // units of the latency mode; lz_split_min restores a threshold.
    const uint32_t min_segs = gt ? 0u : (uint32_t)c->tun.lz_split_min;
    for (uint32_t a = s0; a < s1 && !fused;) {
        const uint32_t b0 = segs[a].blk_base;
        uint32_t b = a + 1;
        while (b < s1 && (b < nseg_all ? segs[b].blk_base : nblk) - b0 + bps <= split_blocks) b++;
        const uint32_t b1 = b < nseg_all ? segs[b].blk_base : nblk;
        if (b - a < min_segs && !waveparse) { s0 = a; s1 = b; fused_tail = b < s1_all; break; }
        if ((c->tun.lz_pbuf_fail && !gt) /* testing: as if the allocation failed */ || c->pbuf.ensure(((size_t)std::max<uint32_t>(b1 - b0, 1) << segs[a].blk_log) * 4) ||
            (gt && c->gtab.ensure((size_t)(b - a) << (lz_gtab_log() + 2)))) {
            (void)hipGetLastError();                                   // (the failed allocation's sticky code)
            if (split_blocks > 1024 && b - a > 1) { split_blocks /= 2; continue; }                    // (for this call only: the next one tries the full run again)
            if (gt) return fail(c, PNA_E_NOMEM, "no room for the strong level set's hash tables");    // (the one-kernel form has LDS tables: other bytes)
            s0 = a; break;                                             // no room for the words: the rest goes through the fused kernel
        }
        hipEvent_t e1 = nullptr;
        if (timed) {
            while (c->lzm_ev.size() < c->lzm_used + 2) { hipEvent_t e = nullptr; HIPCHK(c, hipEventCreate(&e)); c->lzm_ev.push_back(e); }
            HIPCHK(c, hipEventRecord(c->lzm_ev[c->lzm_used], st)); e1 = c->lzm_ev[c->lzm_used + 1]; c->lzm_used += 2; c->lzm_nl.push_back(1);
        }
        const LzParseGrid pg{c->d_segs, c->d_blk_seg, b1 - b0};            // the parse kernel: one wave per block of the run
        // The match kernel runs one workgroup per segment and CU, all of equal length: a run of 3 334 segments is 13 full rounds of the 256 CUs and a 14th for
        // 6 of them.  The segments behind the last full round are therefore cut into UNITS of one block each (table pre-warmed: the same words, SS4a), a launch
        // of their own behind the full rounds: R x 8 short workgroups instead of R long ones next to 256 - R idle CUs.
        uint32_t R = (!gt && !waveparse && c->tun.tail_units && b - a >= 2 * c->n_cus) ? (b - a) % c->n_cus : 0;     // (n_cus: the device's compute units, 256 on an MI355X)
        if (R > 3 * c->n_cus / 8) R = 0;
        if (R) {
            const uint32_t bl = segs[a].blk_log, bs = 1u << bl;
            size_t nu = 0;
            for (uint32_t sgi = b - R; sgi < b; sgi++) nu += std::max<uint32_t>(1, (segs[sgi].len + bs - 1) >> bl);
            constexpr size_t TAIL_CAP = 16 * 96 * 128;                 // (allocated once at this size: a buffer that grew would move under the launches already queued)
            if (c->tail_used + nu > TAIL_CAP || c->h_tail.ensure(TAIL_CAP * sizeof(SegDesc)) || c->d_tail.ensure(TAIL_CAP * sizeof(SegDesc))) R = 0;
            else {
                SegDesc *hu = (SegDesc *)c->h_tail.p + c->tail_used; SegDesc *du = (SegDesc *)c->d_tail.p + c->tail_used;
                size_t k = 0;
                for (uint32_t sgi = b - R; sgi < b; sgi++)
                    for (uint32_t u = 0; u < std::max<uint32_t>(segs[sgi].len, 1); u += bs) { SegDesc us = segs[sgi]; us.u0 = u; us.u1 = std::min<uint32_t>(segs[sgi].len, u + bs); hu[k++] = us; }
                HIPCHK(c, hipMemcpyAsync(du, hu, nu * sizeof(SegDesc), hipMemcpyHostToDevice, st));
                launch_lz(d_src, c->d_segs + a, b - a - R, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, ctab, flags, max_off, max_len, st,
                          (uint32_t *)c->pbuf.p, b0, nullptr, nullptr, nullptr);                        // the full rounds: match kernel only
                launch_lz(d_src, du, (uint32_t)nu, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, ctab, flags, max_off, max_len, st,
                          (uint32_t *)c->pbuf.p, b0, e1, nullptr, &pg);                                 // the rest in units, then the parse kernel over the whole run
                c->tail_used += nu;
                if (timed) c->lzm_nl.back() = 2;                               // (the pair of events spans both launches of the match kernel)
            }
        }
        if (!R)
        launch_lz(d_src, c->d_segs + a, b - a, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, ctab, flags | (waveparse ? 0x1000u : 0u), max_off, max_len, st,
                  (uint32_t *)c->pbuf.p, b0, e1, gt ? (uint32_t *)c->gtab.p : nullptr, &pg);
        a = b;
        if (a >= s1) return PNA_OK;
    }
    launch_lz(d_src, c->d_segs + s0, s1 - s0, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, ctab, flags, max_off, max_len, st, nullptr, 0, nullptr, nullptr, nullptr);
    if (fused_tail) return lz_stage(c, d_src, segs, nseg_all, s1, s1_all, nblk, ctab, flags, max_off, max_len, st, timed);   // (a short run in the middle: only with tiny PNA_LZ_SPLIT_BLOCKS)
    return PNA_OK;
}

// stage times of a finished sub-batch from its events (the stream has been waited for)
static int collect_timing(pna_gpu_ctx *c, bool defl, int nch, uint32_t nseg, uint32_t nblk, bool with_cipher) {
    float ms[6] = {0, 0, 0, 0, 0, 0}, msf = 0;
    (void)hipEventElapsedTime(&msf, c->ev[6], c->ev[7]);
    c->timing.ms_frame += msf;
    float mc = 0;                                             // the cipher kernels run inside the "pack" interval: report them apart
    if (with_cipher) { (void)hipEventElapsedTime(&mc, c->ev_ci[0], c->ev_ci[1]); c->timing.ms_cipher += mc; c->timing.ms_pack -= mc; }
    if (defl) {
        (void)hipEventElapsedTime(&ms[0], c->ev[0], c->ev[1]);
        (void)hipEventElapsedTime(&ms[1], c->ev[1], c->ev[2]);
        (void)hipEventElapsedTime(&ms[2], c->ev[2], c->ev[3]);
        (void)hipEventElapsedTime(&ms[3], c->ev[3], c->ev[4]);
        (void)hipEventElapsedTime(&ms[4], c->ev[4], c->ev[5]);
        (void)hipEventElapsedTime(&ms[5], c->ev[5], c->ev[6]);
    } else {
        // k_lz: first launch to last completion on the main stream; the entropy stages are summed over the chunks on the
        // auxiliary stream (with more than one chunk they overlap k_lz and add up to more than the wall time); "pack" = from the
        // end of the last chunk's entropy stage to the end of the write kernels (plan + scan + layout + write)
        (void)hipEventElapsedTime(&ms[0], c->ev_lz[0], c->ev_lz[nch]);
        for (int k = 0; k < nch; k++) {
            float a = 0, b2 = 0, d = 0;
            (void)hipEventElapsedTime(&a, c->ev_en[k][0], c->ev_en[k][1]);
            (void)hipEventElapsedTime(&b2, c->ev_en[k][1], c->ev_en[k][2]);
            (void)hipEventElapsedTime(&d, c->ev_en[k][2], c->ev_en[k][3]);
            ms[1] += a; ms[2] += b2; ms[3] += d;
        }
        (void)hipEventElapsedTime(&ms[4], c->ev_en[nch - 1][3], c->ev[6]);
    }
    c->timing.ms_lz += ms[0]; c->timing.ms_stats += ms[1]; c->timing.ms_lit += ms[2]; c->timing.ms_seq += ms[3];
    c->timing.ms_pack += ms[4] + ms[5];
    for (size_t i = 0; i + 1 < c->lzm_used; i += 2) { float m = 0; (void)hipEventElapsedTime(&m, c->lzm_ev[i], c->lzm_ev[i + 1]); c->timing.ms_lz_match += m; c->timing.lz_match_launches += i / 2 < c->lzm_nl.size() ? c->lzm_nl[i / 2] : 1u; }
    c->timing.n_segments += nseg; c->timing.n_blocks += nblk;
    return PNA_OK;
}

static int run_subbatch(pna_gpu_ctx *c, int algo, const uint8_t *d_src, const uint64_t *src_off, const uint64_t *src_len,
                        size_t e0, size_t e1, uint8_t *d_dst, size_t dst_cap, uint64_t out_base, uint64_t *dst_off,
                        hipStream_t st, bool timed, const FrameJob *fj = nullptr) {
    // LATENCY MODE: a small batch -- the CompressionWriter seam with a handful of writers in flight, one entry of
    // `pna_gpu_compress_batch` -- has fewer segments than the chip has CUs, and its time is the length of the per-segment and per-block serial
    // chains (one workgroup walks a segment's 256 tiles; one lane codes a block's sequences).  Such a batch is cut finer: blocks of 8 .. 64 KiB
    // inside the same frames (blk_log), and the LZ stage runs one workgroup per UNIT of 1 << unit_log bytes whose table is pre-warmed with
    // everything before it (lz_prewarm), which gives the very matches of the segment-long walk.  Both follow from the batch's size alone
    // (and pna_gpu_set_option), are reported by pna_gpu_last_timing, and are parameters of the oracle's model.
    uint64_t in_total = 0, nseg_est = 0, max_len = 0;
    for (size_t e = e0; e < e1; e++) { in_total += src_len[e]; nseg_est += src_len[e] ? (src_len[e] + SEG_SIZE - 1) / SEG_SIZE : 1; max_len = std::max<uint64_t>(max_len, src_len[e]); }
    // an upper bound of every payload of the sub-batch when the entries are small and plain (k_frame's wave-per-entry form takes those; 0: no such bound)
    const uint32_t frame_max_payload = (fj && !fj->solid && !fj->cipher && max_len <= 16384) ? (uint32_t)std::min<size_t>(pna_gpu_bound(algo, (size_t)max_len), 0xFFFFFFFFu) : 0u;
    const bool latency = c->tun.latency_max_mib > 0 && in_total <= ((uint64_t)c->tun.latency_max_mib << 20) && nseg_est <= 1024 && !(c->call_flags & 0x100u);
    uint32_t blk_log = small_entry_blk_log(c, src_len, e0, e1), unit_log = 20;
    if (latency && blk_log == PNA_BLK_LOG) {
        // blocks: 16 KiB up to 16 MiB of input (a block's sequence chain then is ~1 000 steps), then growing with the batch so that the
        // block count -- per-block fixed costs of the entropy kernels -- stays near 1 024 .. 2 048
        blk_log = 14;
        while (blk_log < PNA_BLK_LOG && (in_total >> blk_log) > 2048) blk_log++;
        // units: about one per CU (256), never smaller than a block
        unit_log = blk_log;
        while (unit_log < 20 && (in_total >> unit_log) > 384) unit_log++;
    }
    if (c->tun.unit_log) unit_log = (uint32_t)std::max<long>(c->tun.unit_log, blk_log);
    if (unit_log < blk_log) unit_log = blk_log;
    const uint32_t bsz = 1u << blk_log;
    // The plan: counted first, then written straight into the page-locked blob that travels to the device in one copy (several threads for the
    // batches of 10^5 .. 10^6 small entries, where this loop is a tenth of the call).  Layout: [segs | units | blk_seg | entry_first_seg].
    const size_t ne_all = e1 - e0;
    std::vector<uint32_t> efs_v(ne_all + 1), efb_v(ne_all + 1), efu_v(ne_all + 1);     // first segment / block / unit of every entry
    bool any_empty = false;
    {
        uint32_t sg = 0, bk = 0, un = 0;
        for (size_t e = e0; e < e1; e++) {
            efs_v[e - e0] = sg; efb_v[e - e0] = bk; efu_v[e - e0] = un;
            const uint64_t len = src_len[e];
            if (src_off[e] & 15) return fail(c, PNA_E_INVAL, "entry offset not 16-byte aligned");
            if (len == 0) { sg++; any_empty = true; continue; }
            const uint64_t full = len >> 20, rest = len & (SEG_SIZE - 1);
            sg += (uint32_t)(full + (rest ? 1 : 0));
            bk += (uint32_t)((full << (20 - blk_log)) + ((rest + bsz - 1) >> blk_log));
            if (unit_log < 20) un += (uint32_t)((full << (20 - unit_log)) + ((rest + (1u << unit_log) - 1) >> unit_log));
        }
        efs_v[ne_all] = sg; efb_v[ne_all] = bk; efu_v[ne_all] = un;
    }
    const uint32_t nseg = efs_v[ne_all], nblk = efb_v[ne_all], nunits = efu_v[ne_all];
    if (nseg == 0) return PNA_OK;
    const size_t o_units = ((size_t)nseg * sizeof(SegDesc) + 15) & ~(size_t)15, o_blkseg = (o_units + (size_t)nunits * sizeof(SegDesc) + 15) & ~(size_t)15,
                 o_entry = (o_blkseg + (size_t)(nblk + 1) * 4 + 15) & ~(size_t)15, plan_bytes = o_entry + (ne_all + 2) * 4;
    if (c->plan.ensure(plan_bytes) || c->h_plan.ensure(plan_bytes)) return fail(c, PNA_E_NOMEM, "workspace allocation failed");
    SegDesc *segs = (SegDesc *)c->h_plan.p, *units = (SegDesc *)((uint8_t *)c->h_plan.p + o_units);   // (the previous sub-batch has been waited for: the staging is free)
    c->tail_used = 0;
    uint32_t *blk_seg = (uint32_t *)((uint8_t *)c->h_plan.p + o_blkseg), *entry_first_seg = (uint32_t *)((uint8_t *)c->h_plan.p + o_entry);
    // option single_frame (zstd): an entry's segments form ONE frame -- the frame header in front of the first segment only, the last-block bit on the entry's
    // last block only (SegDesc::first bit 2 tells k_plan / k_write); the blocks are what they are in the frame-per-segment form
    const uint32_t sf_bit = (algo == PNA_ALGO_ZSTD && c->tun.single_frame) ? 4u : 0u;
    {
        auto fill = [&](size_t a, size_t b) {
            for (size_t e = a; e < b; e++) {
                uint32_t sg = efs_v[e - e0], bk = efb_v[e - e0], un = efu_v[e - e0];
                entry_first_seg[e - e0] = sg;
                const uint64_t len = src_len[e], off = src_off[e];
                if (len == 0) { segs[sg] = SegDesc{off, 0, bk, (uint32_t)e, 3, 0, 0, blk_log, 0}; continue; }
                for (uint64_t p = 0; p < len; p += SEG_SIZE) {
                    const uint32_t sl = (uint32_t)std::min<uint64_t>(SEG_SIZE, len - p);
                    const SegDesc s{off + p, sl, bk, (uint32_t)e, (p == 0 ? 1u : 0u) | (p + SEG_SIZE >= len ? 2u : 0u) | sf_bit, 0, sl, blk_log, 0};
                    const uint32_t nb = (sl + bsz - 1) >> blk_log;
                    for (uint32_t b2 = 0; b2 < nb; b2++) blk_seg[bk + b2] = sg;
                    bk += nb; segs[sg++] = s;
                    if (unit_log < 20)
                        for (uint32_t u = 0; u < sl; u += 1u << unit_log) { SegDesc us = s; us.u0 = u; us.u1 = std::min<uint32_t>(sl, u + (1u << unit_log)); units[un++] = us; }
                }
            }
        };
        const unsigned nt = (unsigned)std::min<size_t>(8, std::max<size_t>(1, ne_all / 32768));
        if (nt > 1) { std::vector<std::thread> th; for (unsigned t = 0; t < nt; t++) th.emplace_back(fill, e0 + ne_all * t / nt, e0 + ne_all * (t + 1) / nt); for (auto &x : th) x.join(); }
        else fill(e0, e1);
        entry_first_seg[ne_all] = nseg;
    }
    const bool unit_mode = unit_log < 20 && nunits > 0;
    c->last_blk_log = blk_log; c->last_units = unit_mode ? nunits : 0;
    // zstd entropy stage, two forms: statistics per block (k_hist) + tables + three-lane state chains (k_seqa) + token-parallel packing (k_seqb) while
    // the chain waves fit the chip's SIMDs (<= 40 960 blocks); beyond, statistics per segment inside k_stats and the one-kernel coder k_seq.  Flags
    // 0x1000 / 0x2000 and option hist_by_block force one.
    const bool hist_on = algo == PNA_ALGO_ZSTD && !(c->call_flags & 0x1000u) &&
                         ((c->call_flags & 0x2000u) || (c->tun.hist_by_block < 0 ? nblk <= 40960u : c->tun.hist_by_block != 0));
    const size_t o_hist = ((size_t)(nblk + 1) * sizeof(BlkInfo) + 15) & ~(size_t)15, blk_bytes = o_hist + (hist_on ? (size_t)nseg * 448 * 4 : 0);
    if (c->blk.ensure(blk_bytes) || c->tabs.ensure((size_t)nseg * std::max(sizeof(SegTables), sizeof(DeflTables))) ||
        (algo == PNA_ALGO_DEFLATE && c->ctab.ensure(((size_t)(nblk + 1) << (blk_log - 11)) * 16)) ||
        c->seqs.ensure((size_t)(nblk + 1) * seq_cap_of(blk_log) * 8) || c->lits.ensure((size_t)(nblk + 1) << blk_log) ||
        c->litc.ensure((size_t)(nblk + 1) << blk_log) || (algo == PNA_ALGO_ZSTD && c->seqc.ensure((size_t)(nblk + 1) << blk_log)) ||
        (algo == PNA_ALGO_ZSTD && c->seqw.ensure(hist_on ? (size_t)(nblk + 1) * seq_cap_of(blk_log) * 8 : 64)) ||
        c->seg_size.ensure((size_t)nseg * 8) || c->seg_off.ensure(((size_t)nseg + 1 + 2 * ((size_t)nseg / 4096 + 2)) * 8))       // (+ the scratch of the hierarchical scan)
        return fail(c, PNA_E_NOMEM, "workspace allocation failed");
    {
        uint8_t *hp = (uint8_t *)c->h_plan.p, *dp = (uint8_t *)c->plan.p;
        c->d_segs = (SegDesc *)dp; c->d_units = (SegDesc *)(dp + o_units); c->d_blk_seg = (uint32_t *)(dp + o_blkseg); c->d_entry_seg = (uint32_t *)(dp + o_entry);
        c->d_hist = (uint32_t *)((uint8_t *)c->blk.p + o_hist);
        HIPCHK(c, hipMemcpyAsync(dp, hp, plan_bytes, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemsetAsync(c->blk.p, 0, blk_bytes, st));                     // BlkInfo of every block and, behind them, the segments' histogram counters
    }
    c->lzm_used = 0; c->lzm_nl.clear();
    const bool defl = algo == PNA_ALGO_DEFLATE;
    if (timed) HIPCHK(c, hipEventRecord(c->ev[0], st));
    int nch = 1;
    if (defl) {
        const uint32_t dfl = (c->call_flags & (F_LAZY | F_ADOPT | F_INS2 | F_STRONG | 0x300u)) | (c->call_lazy2 ? FLAG_LAZY2 : 0u) | (c->call_lazy3 ? FLAG_LAZY3 : 0u) | FLAG_LEN36;
        if (c->call_stored) { }                                // deflate level 0 = Compression::none(): stored blocks only, no match finder, no codes
        else if (unit_mode) launch_lz(d_src, c->d_units, nunits, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, (uint4 *)c->ctab.p, dfl, 32768u, 258u, st, nullptr, 0, nullptr, nullptr, nullptr);
        else { const int rc = lz_stage(c, d_src, segs, nseg, 0, nseg, nblk, (uint4 *)c->ctab.p, dfl, 32768u, 258u, st, timed); if (rc) return rc; }
        if (timed) HIPCHK(c, hipEventRecord(c->ev[1], st));
        launch_deflate_stage1(d_src, c->d_segs, nseg, c->d_blk_seg, nblk, (const uint64_t *)c->seqs.p,
                              (const uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, (const uint4 *)c->ctab.p, (DeflTables *)c->tabs.p, (uint8_t *)c->litc.p,
                              (uint64_t *)c->seg_size.p, (uint64_t *)c->seg_off.p, st, timed ? &c->ev[2] : nullptr, c->call_flags, c->call_stored);
    } else {
        // zstd: the segments go through k_lz in chunks on `st`; the entropy stage of a finished chunk runs on the auxiliary
        // stream next to the following chunk's k_lz (latency-bound kernels hide in the issue slots k_lz leaves free)
        if (!c->aux) {
            { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // (lowest priority: what runs here fills in beside the main stream's kernels)
              HIPCHK(c, hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, lo)); }
            for (auto &e : c->ev_lz) HIPCHK(c, hipEventCreate(&e));
            for (auto &r : c->ev_en) for (auto &e : r) HIPCHK(c, hipEventCreate(&e));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        }
        // Measured (10 000 x 1 MiB): 4 chunks 111.8 ms vs 108.9 ms unchunked -- k_seq's duration is set by the length of one
        // block's tANS chain, not by the number of blocks, so every chunk pays it in full and the co-resident waves slow k_lz
        // by 12 %.  The chunked form therefore stays off unless PNA_PIPELINE_CHUNKS asks for it.
        nch = (int)c->tun.pipeline_chunks; if (nch < 1) nch = 1; if (nch > pna_gpu_ctx::MAXCH) nch = pna_gpu_ctx::MAXCH; if ((uint32_t)nch > nseg || latency) nch = 1;
        HIPCHK(c, hipEventRecord(c->ev_lz[0], st));
        for (int k = 0; k < nch; k++) {
            const uint32_t s0 = (uint32_t)((uint64_t)nseg * k / nch), s1 = (uint32_t)((uint64_t)nseg * (k + 1) / nch);
            const uint32_t g0 = segs[s0].blk_base, g1 = s1 < nseg ? segs[s1].blk_base : nblk;
            const uint32_t zfl = (c->call_flags & 0x3FFu) | (c->call_w32 ? (c->call_w16 ? FLAG_W16 : FLAG_W32) : 0u) | (c->call_lazy2 ? FLAG_LAZY2 : 0u) | (c->call_lazy3 ? FLAG_LAZY3 : 0u) | (c->call_gtab ? 0u : FLAG_LEN36) | (c->call_tab3 ? FLAG_TAB3 : 0u);
            const uint32_t zmax = (c->call_flags & F_FAR) ? (c->call_gtab ? MAX_OFF : MAX_OFF_W3) : NEAR_OFF;   // (3-byte words keep 19 bits of offset)
            if (unit_mode) {
                // (nch == 1: one launch over all units; the strong set: split form over the units, tables in global memory)
                const bool gt = c->call_gtab;
                const LzParseGrid pgu{c->d_segs, c->d_blk_seg, nblk};
                if (gt && (c->pbuf.ensure(((size_t)nblk << blk_log) * 4) || c->gtab.ensure((size_t)nunits << (lz_gtab_log() + 2)))) return fail(c, PNA_E_NOMEM, "no room for the strong level set's hash tables");
                launch_lz(d_src, c->d_units, nunits, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, nullptr, zfl,
                          zmax, 0xFFFFFFFFu, st, gt ? (uint32_t *)c->pbuf.p : nullptr, 0, nullptr, gt ? (uint32_t *)c->gtab.p : nullptr, gt ? &pgu : nullptr);
            }
            else { const int rc = lz_stage(c, d_src, segs, nseg, s0, s1, nblk, nullptr, zfl, zmax, 0xFFFFFFFFu, st, timed); if (rc) return rc; }
            // (one chunk: everything stays on `st` -- a hand-over to the auxiliary stream and back costs ~45 us of idle device, a tenth of a small batch)
            hipStream_t est = nch > 1 ? c->aux : st;
            HIPCHK(c, hipEventRecord(c->ev_lz[k + 1], st));
            if (nch > 1) HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_lz[k + 1], 0));
            HIPCHK(c, hipEventRecord(c->ev_en[k][0], est));
            launch_entropy_chunk(c->d_segs, s0, s1 - s0, c->d_blk_seg, g0, g1 - g0, (const uint64_t *)c->seqs.p,
                                 (const uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, (SegTables *)c->tabs.p, (uint8_t *)c->litc.p, (uint8_t *)c->seqc.p,
                                 (uint32_t *)c->seqw.p, c->call_flags, blk_log, hist_on ? c->d_hist : nullptr, est, &c->ev_en[k][1],
                                 (nch == 1 && c->tun.lit_beside_seq) ? c->aux : nullptr, c->ev_fork, c->ev_join);
        }
        if (nch > 1) { HIPCHK(c, hipEventRecord(c->ev_join, c->aux)); HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0)); }
        if (timed) HIPCHK(c, hipEventRecord(c->ev[4], st));
        launch_plan(c->d_segs, nseg, (BlkInfo *)c->blk.p, (const SegTables *)c->tabs.p, (uint64_t *)c->seg_size.p,
                    (uint64_t *)c->seg_off.p, c->call_flags, st);
        if (timed) HIPCHK(c, hipEventRecord(c->ev[5], st));
    }
    HIPCHK(c, hipGetLastError());
    // while the kernels run: the name-dependent part of every entry record (FHED and fSIZ chunks with their CRCs)
    FrameDesc *fds = nullptr; uint8_t *blob = nullptr; uint64_t *segdst = nullptr; size_t blob_len = 0;
    const bool solid = fj && fj->solid;
    const bool gcm = fj && fj->cipher && fj->cipher->cipher_mode == PNA_MODE_GCM;
    const uint32_t gcm_seg = gcm ? (fj->cipher->gcm_segment_size ? fj->cipher->gcm_segment_size : (1u << 20)) : 0u;   // default = the reference's DEFAULT_SEGMENT_SIZE (1 MiB)
    std::vector<GcmMaterial> gmat; std::vector<GcmEntry> gents;
    // GCM STREAM segments of this sub-batch, in order (an entry has ceil(payload / segment_size) of them, at least one): counter-mode IV
    // (nonce || 2) and the entry they belong to; `spread`: pieces of the compact payload that move to their place between the tags
    struct GcmSeg { uint8_t ctr_iv[16]; uint32_t entry; };
    std::vector<GcmSeg> gsegs;
    struct SpreadPiece { uint64_t src, dst; uint32_t len; };
    std::vector<SpreadPiece> spread; uint64_t spread_bytes = 0;
    std::vector<std::pair<uint64_t, uint64_t>> spread_copy;     // (archive offset, length) of the compact payloads to save first
    size_t nunit = e1 - e0;                                    // framed units: entries, or the segments of the solid stream
    if (solid) {
        if (e1 - e0 != 1) return fail(c, PNA_E_INVAL, "a solid stream is one entry");
        nunit = nseg;
        // (GCM: one SDAT chunk per GCM segment of the compressed stream -- at most the worst-case output / segment size + 1 of them)
        const size_t ucap = gcm ? (size_t)(pna_gpu_bound(algo, (size_t)src_len[e0]) / gcm_seg) + 2 : nunit;
        if (c->h_desc.ensure(ucap * sizeof(FrameDesc)) || c->h_blob.ensure(ucap * 8 + 16) || c->h_segdst.ensure((size_t)(nseg + 1) * 8))
            return fail(c, PNA_E_NOMEM, "framing staging");
        fds = (FrameDesc *)c->h_desc.p; blob = (uint8_t *)c->h_blob.p; segdst = (uint64_t *)c->h_segdst.p;
        if (gcm) {
            gmat.resize(1);
            uint8_t kc[32], ph[32];
            hkdf_sha256_32(fj->cipher->key, 32, nullptr, 0, "PNA-KC-v1", 9, kc);
            sha256_bytes(fj->cipher->phsf, strlen(fj->cipher->phsf), nullptr, 0, ph);
            gcm_entry_material(fj->cipher, kc, ph, fj->ivs, gcm_seg, nullptr, algo, gmat[0]);
        }
    } else if (fj) {
        std::vector<uint8_t> tmp;
        size_t bound = 0;
        for (size_t e = e0; e < e1; e++) bound += (fj->cipher ? frame_entry_prefix_enc_bound(fj->names[e], fj->cipher->phsf) : frame_entry_prefix_bound(fj->names[e])) + meta_len(fj->meta, e);
        // (every FDAT chunk behind an entry's first needs a descriptor and 8 prefix bytes: at most one per max_chunk_size bytes of the worst-case output)
        size_t extra_chunks = 0;
        { const uint64_t CHb = chunk_limit(fj->max_chunk); for (size_t e = e0; e < e1; e++) extra_chunks += (size_t)((pna_gpu_bound(algo, (size_t)src_len[e]) + 64 + 16 * (src_len[e] >> 12)) / CHb); }
        if (c->h_desc.ensure(((e1 - e0) + extra_chunks + 1) * sizeof(FrameDesc)) || c->h_blob.ensure(bound + 8 * extra_chunks + 16) || c->h_segdst.ensure((size_t)(nseg + 1) * 8))
            return fail(c, PNA_E_NOMEM, "framing staging");
        fds = (FrameDesc *)c->h_desc.p; blob = (uint8_t *)c->h_blob.p; segdst = (uint64_t *)c->h_segdst.p;
        if (gcm) {
            // GCM STREAM: per entry a stream header, an HKDF stream key bound to its FHED chunk, round keys, hash subkey, E(K, J0);
            // a few host threads share the entries (SHA-256 / HKDF / key schedule: a few microseconds each) while k_lz runs
            gmat.resize(e1 - e0);
            uint8_t kc[32], ph[32];
            hkdf_sha256_32(fj->cipher->key, 32, nullptr, 0, "PNA-KC-v1", 9, kc);               // key_confirmation, aead.rs:161-163
            sha256_bytes(fj->cipher->phsf, strlen(fj->cipher->phsf), nullptr, 0, ph);
            const unsigned nt = (unsigned)std::min<size_t>(8, std::max<size_t>(1, (e1 - e0) / 256));
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    for (size_t e = e0 + t; e < e1; e += nt)
                        gcm_entry_material(fj->cipher, kc, ph, fj->ivs + 39 * e, gcm_seg, fj->names[e], algo, gmat[e - e0]);
                });
            for (auto &x : th) x.join();
        }
        if (!fj->cipher && !fj->meta) {
            // plain entries: the prefixes are written straight into the staging blob; with many entries several threads share them
            // (each thread fills the slots of its range at the bound offsets, a compaction pass closes the gaps)
            const size_t ne = e1 - e0;
            const unsigned nt = (unsigned)std::min<size_t>(8, std::max<size_t>(1, ne / 8192));
            std::vector<uint32_t> plen(ne); std::vector<size_t> slot(ne + 1); slot[0] = 0;
            for (size_t i = 0; i < ne; i++) slot[i + 1] = slot[i] + frame_entry_prefix_bound(fj->names[e0 + i]);
            auto work = [&](unsigned t) { for (size_t i = t; i < ne; i += nt) plen[i] = (uint32_t)frame_entry_prefix_into(blob + slot[i], fj->names[e0 + i], algo, src_len[e0 + i]); };
            if (nt > 1) { std::vector<std::thread> th; for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); } else work(0);
            for (size_t i = 0; i < ne; i++) {
                if (blob_len != slot[i]) memmove(blob + blob_len, blob + slot[i], plen[i]);
                fds[i] = FrameDesc{0, 0, (uint32_t)blob_len, plen[i], 0};
                blob_len += plen[i];
            }
        } else
        for (size_t e = e0; e < e1; e++) {
            tmp.clear();
            if (gcm) frame_entry_prefix_enc(tmp, fj->names[e], algo, src_len[e], fj->cipher->encryption, PNA_MODE_GCM, fj->cipher->phsf, gmat[e - e0].header, 75);
            else if (fj->cipher) frame_entry_prefix_enc(tmp, fj->names[e], algo, src_len[e], fj->cipher->encryption, fj->cipher->cipher_mode, fj->cipher->phsf, fj->ivs + 16 * e, 16);
            else frame_entry_prefix(tmp, fj->names[e], algo, src_len[e], 0);
            splice_meta(tmp, fj->meta, e);
            memcpy(blob + blob_len, tmp.data(), tmp.size());
            fds[e - e0] = FrameDesc{0, 0, (uint32_t)blob_len, (uint32_t)tmp.size(), 0};
            blob_len += tmp.size();
        }
    }
    // Plain file entries of one FDAT chunk each (no cipher; the worst case of every payload below the chunk limit and of the whole sub-batch below the
    // destination's capacity): the archive layout is computed on the device (k_layout) and the host never waits in the middle of the sub-batch.
    bool dev_layout = fj && !solid && !fj->cipher && c->tun.dev_layout != 0;
    if (dev_layout) {
        const uint64_t CH = chunk_limit(fj->max_chunk);
        uint64_t need = out_base;
        for (size_t e = e0; e < e1 && dev_layout; e++) {
            const uint64_t b = pna_gpu_bound(algo, (size_t)src_len[e]);
            if (b > CH) dev_layout = false;
            need += fds[e - e0].prefix_len + b + 16;
        }
        if (need + 16 > dst_cap) dev_layout = false;
    }
    if (dev_layout) {
        const size_t ne = e1 - e0;
        if (c->fr_desc.ensure(ne * sizeof(FrameDesc)) || c->fr_blob.ensure(blob_len + 16) || c->fr_segdst.ensure((size_t)(nseg + 1) * 8) ||
            c->fr_entoff.ensure((ne + 2 + 2 * (ne / 1024 + 2)) * 8) || c->h_entoff.ensure((ne + 2) * 8)) return fail(c, PNA_E_NOMEM, "framing workspace");
        int rcc = ensure_crc(c); if (rcc) return rcc;
        HIPCHK(c, hipMemcpyAsync(c->fr_desc.p, fds, ne * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->fr_blob.p, blob, blob_len, hipMemcpyHostToDevice, st));
        uint64_t *d_ent = (uint64_t *)c->fr_entoff.p;
        launch_layout((FrameDesc *)c->fr_desc.p, (uint8_t *)c->fr_blob.p, c->d_entry_seg, (const uint64_t *)c->seg_off.p, (uint32_t)ne, nseg, out_base,
                      (uint64_t *)c->fr_segdst.p, d_ent, d_ent + ne + 1, st);
        if (defl) launch_deflate_write(d_src, c->d_segs, c->d_blk_seg, nblk, (const BlkInfo *)c->blk.p, (const uint64_t *)c->fr_segdst.p, (const uint64_t *)c->seg_size.p,
                                       (const uint8_t *)c->litc.p, c->d_entry_seg, (uint32_t)ne, d_dst, st, c->call_stored);
        else launch_write(d_src, c->d_segs, nseg, c->d_blk_seg, nblk, (const BlkInfo *)c->blk.p, (const SegTables *)c->tabs.p, (const uint64_t *)c->fr_segdst.p,
                          (const uint8_t *)c->lits.p, (const uint8_t *)c->litc.p, (const uint8_t *)c->seqc.p, d_dst, any_empty, st);
        if (timed) HIPCHK(c, hipEventRecord(c->ev[6], st));
        launch_frame((const FrameDesc *)c->fr_desc.p, (uint32_t)ne, (const uint8_t *)c->fr_blob.p, (const CrcTabs *)c->crc_tabs.p,
                     d_dst, (uint64_t)dst_cap & ~(uint64_t)15, frame_fend_crc(), "FDAT", true, st, frame_max_payload);
        if (timed) HIPCHK(c, hipEventRecord(c->ev[7], st));
        // the entry offsets (when the caller wants them) and the sub-batch's length travel back behind the kernels: the call's one wait
        uint64_t *h_ent = (uint64_t *)c->h_entoff.p;
        if (fj->want_offsets) HIPCHK(c, hipMemcpyAsync(h_ent, d_ent, (ne + 2) * 8, hipMemcpyDeviceToHost, st));
        else HIPCHK(c, hipMemcpyAsync(h_ent + ne, d_ent + ne, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        HIPCHK(c, hipGetLastError());
        if (fj->want_offsets) memcpy(dst_off + e0, h_ent, ne * 8);
        dst_off[e1] = h_ent[ne];
        c->last_nblk = nblk;
        if (timed) {
            int rct = collect_timing(c, defl, nch, nseg, nblk, false);
            if (rct) return rct;
        }
        return PNA_OK;
    }
    // the output offsets are needed on the host before the write pass can be bounds-checked
    if (c->h_segoff.ensure((size_t)(nseg + 1) * 8)) return fail(c, PNA_E_NOMEM, "offset staging");
    const uint64_t *seg_off = (const uint64_t *)c->h_segoff.p;
    // (plain batches whose destination holds the worst case of every entry need no check against the sizes found: the write kernels go out
    // at once, the offsets travel behind them and the one wait is the call's last -- a wait in the middle of a small batch is a tenth of it)
    bool early_write = !fj;
    if (early_write) { uint64_t need = out_base; for (size_t e = e0; e < e1; e++) need += pna_gpu_bound(algo, (size_t)src_len[e]); early_write = need <= dst_cap; }
    if (!early_write) {
        HIPCHK(c, hipMemcpyAsync(c->h_segoff.p, c->seg_off.p, (size_t)(nseg + 1) * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    uint64_t total = early_write ? 0 : seg_off[nseg];
    std::vector<CipherUnit> cunits;
    const uint64_t *d_segdst = (const uint64_t *)c->seg_off.p;
    uint8_t *wbase = d_dst + out_base;
    if (fj) {
        // archive layout of this sub-batch: [prefix | payload | crc | FEND] per entry; the write kernels put every
        // segment straight at its final place, k_frame adds the rest (no second copy of the payload)
        uint64_t pos = out_base;
        if (solid) {
            // solid stream: one SDAT chunk per segment (= per zstd frame / per run of deflate blocks): [len "SDAT" | payload | crc]
            dst_off[e0] = pos;
            if (gcm) {
                // GCM STREAM over the solid stream (into_solid_archive takes any cipher, lib/src/archive/write.rs:443-470; GcmEncryptWriter, lib/src/cipher/
                // gcm.rs:48-100): the head carries the stream header as its first SDAT chunk; here one SDAT chunk per GCM segment, ciphertext || tag.  The
                // write kernels put the compressed stream down in one piece behind the first chunk header, segments k >= 1 then move forward by 28 k
                // bytes (tag and CRC of the chunk before + their own chunk header), as a GCM entry of several segments does.
                const uint64_t P = seg_off[nseg] - seg_off[0], G = gcm_seg;
                const uint64_t K = P ? (P + G - 1) / G : 1;
                if (K > 0xFFFFFFFFull) return fail(c, PNA_E_INVAL, "GCM segment counter overflow");
                const uint64_t B = pos + 8;
                for (uint32_t sg = 0; sg < nseg; sg++) segdst[sg] = B + (seg_off[sg] - seg_off[0]);
                if (K > 1) spread_copy.emplace_back(B, P);
                const GcmMaterial &gm = gmat[0];
                for (uint64_t k = 0; k < K; k++) {
                    const uint64_t sl = std::min<uint64_t>(G, P - k * G), hk = pos + k * (G + 28), so_ = hk + 8;
                    uint8_t *pf = blob + 8 * (size_t)k;
                    const uint64_t cl = sl + 16;
                    pf[0] = (uint8_t)(cl >> 24); pf[1] = (uint8_t)(cl >> 16); pf[2] = (uint8_t)(cl >> 8); pf[3] = (uint8_t)cl;
                    memcpy(pf + 4, "SDAT", 4);
                    fds[k] = FrameDesc{hk, (uint32_t)cl, (uint32_t)(8 * k), 8u, 0};
                    const uint32_t si = (uint32_t)gsegs.size();
                    uint8_t j0[16], eb[16];
                    memcpy(j0, gm.ctr_iv, 7);
                    j0[7] = (uint8_t)(k >> 24); j0[8] = (uint8_t)(k >> 16); j0[9] = (uint8_t)(k >> 8); j0[10] = (uint8_t)k; j0[11] = k + 1 == K ? 1 : 0;
                    j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
                    aes256_block_host(gm.rk, j0, eb);
                    GcmSeg gs; memcpy(gs.ctr_iv, j0, 16); gs.ctr_iv[15] = 2; gs.entry = 0;
                    gsegs.push_back(gs);
                    for (uint64_t o = 0; o < sl; o += CTR_UNIT) cunits.push_back(CipherUnit{so_ + o, o, (uint32_t)std::min<uint64_t>(CTR_UNIT, sl - o), si});
                    GcmEntry ge{so_, (uint32_t)sl, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
                    memcpy(ge.h, gm.h, 16);
                    for (int w = 0; w < 4; w++) ge.ej0[w] = ((uint32_t)eb[4 * w] << 24) | ((uint32_t)eb[4 * w + 1] << 16) | ((uint32_t)eb[4 * w + 2] << 8) | eb[4 * w + 3];
                    gents.push_back(ge);
                    if (k >= 1)
                        for (uint64_t o = 0; o < sl; o += (1u << 20))
                            spread.push_back(SpreadPiece{spread_bytes + k * G + o, so_ + o, (uint32_t)std::min<uint64_t>(1u << 20, sl - o)});
                }
                if (K > 1) spread_bytes += (P + 15) & ~(uint64_t)15;
                pos += P + 28 * K;
                nunit = (size_t)K; blob_len = 8 * (size_t)K;
            } else {
            for (uint32_t sg = 0; sg < nseg; sg++) {
                const uint64_t plen = seg_off[sg + 1] - seg_off[sg];
                uint8_t *pf = blob + 8 * (size_t)sg;
                pf[0] = (uint8_t)(plen >> 24); pf[1] = (uint8_t)(plen >> 16); pf[2] = (uint8_t)(plen >> 8); pf[3] = (uint8_t)plen;
                memcpy(pf + 4, "SDAT", 4);
                fds[sg] = FrameDesc{pos, (uint32_t)plen, 8u * sg, 8u, 0};
                segdst[sg] = pos + 8;
                if (fj->cipher)                                    // one cipher stream over all SDAT bodies: the keystream position runs on
                    for (uint64_t o = 0; o < plen; o += CTR_UNIT)
                        cunits.push_back(CipherUnit{pos + 8 + o, (seg_off[sg] - seg_off[0]) + o, (uint32_t)std::min<uint64_t>(CTR_UNIT, plen - o), 0u});
                pos += 8 + plen + 4;
            }
            blob_len = 8 * (size_t)nseg;
            }
        } else {
            // FlattenWriter cuts an entry's stream into FDAT chunks of max_chunk_size bytes, the last one holding the rest (lib/src/util/io.rs:60-77:
            // the open chunk is topped up before a new one starts; FileEntryBuilder::max_chunk_size, lib/src/entry/builder/file.rs:105-112; default
            // u32::MAX, lib/src/chunk.rs:28).  The write kernels put an entry's payload down in one piece behind the first FDAT header; for an entry of
            // K > 1 chunks the payload is saved to a scratch buffer and chunks 1 .. K - 1 move forward by 12 k bytes (CRC of the chunk before +
            // their own length / type), k_frame then takes one descriptor per chunk.  The same pass serves the GCM STREAM layout below.
            const uint64_t CH = chunk_limit(fj->max_chunk);
            std::vector<FrameDesc> units; units.reserve(e1 - e0);
            const bool cbc = fj->cipher && fj->cipher->cipher_mode == PNA_MODE_CBC;
            const bool ctr = fj->cipher && !cbc && !gcm;
            for (size_t e = e0; e < e1; e++) {
                const uint32_t s0 = entry_first_seg[e - e0], s1 = entry_first_seg[e - e0 + 1];
                const FrameDesc f0 = fds[e - e0];                  // prefix of the entry: FHED | fSIZ | ... | first FDAT header
                dst_off[e] = pos;
                const uint64_t p0 = pos + f0.prefix_len;           // where the payload starts
                uint64_t plen = seg_off[s1] - seg_off[s0];         // the entry's compressed stream, then what the cipher makes of it
                for (uint32_t sg = s0; sg < s1; sg++) segdst[sg] = p0 + (seg_off[sg] - seg_off[s0]);
                if (cbc) {
                    // CBC chains the whole entry (one lane) and appends the PKCS#7 padding block, in place
                    cunits.push_back(CipherUnit{p0, 0, (uint32_t)plen, (uint32_t)(e - e0)});
                    plen = (plen / 16 + 1) * 16;
                    if (plen > CH) return fail(c, PNA_E_UNSUPPORTED, "CBC entry beyond one FDAT chunk");
                } else if (gcm) {
                    // GCM STREAM (GcmEncryptWriter, lib/src/cipher/gcm.rs:48-100): the payload in segments of segment_size bytes, every
                    // segment followed by its 16-byte tag; all but the last carry nonce flag 0, the last one (possibly full, possibly
                    // empty) flag 1; counters 0, 1, ...  Segments k >= 1 move forward by 16 k bytes before the cipher runs.
                    const uint64_t K = plen ? (plen + gcm_seg - 1) / gcm_seg : 1;
                    if (K > 0xFFFFFFFFull) return fail(c, PNA_E_INVAL, "GCM segment counter overflow");
                    if (plen + 16 * K > CH) return fail(c, PNA_E_UNSUPPORTED, "GCM entry beyond one FDAT chunk");
                    const GcmMaterial &gm = gmat[e - e0];
                    if (K > 1) { spread_copy.emplace_back(p0, plen); }
                    for (uint64_t k = 0; k < K; k++) {
                        const uint64_t sl = std::min<uint64_t>(gcm_seg, plen - k * gcm_seg), so_ = p0 + k * ((uint64_t)gcm_seg + 16);
                        const uint32_t si = (uint32_t)gsegs.size();
                        uint8_t j0[16], eb[16];
                        memcpy(j0, gm.ctr_iv, 7);                                  // nonce prefix
                        j0[7] = (uint8_t)(k >> 24); j0[8] = (uint8_t)(k >> 16); j0[9] = (uint8_t)(k >> 8); j0[10] = (uint8_t)k; j0[11] = k + 1 == K ? 1 : 0;
                        j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
                        aes256_block_host(gm.rk, j0, eb);
                        GcmSeg gs; memcpy(gs.ctr_iv, j0, 16); gs.ctr_iv[15] = 2; gs.entry = (uint32_t)(e - e0);
                        gsegs.push_back(gs);
                        for (uint64_t o = 0; o < sl; o += CTR_UNIT)
                            cunits.push_back(CipherUnit{so_ + o, o, (uint32_t)std::min<uint64_t>(CTR_UNIT, sl - o), si});
                        GcmEntry ge{so_, (uint32_t)sl, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
                        memcpy(ge.h, gm.h, 16);
                        for (int w = 0; w < 4; w++) ge.ej0[w] = ((uint32_t)eb[4 * w] << 24) | ((uint32_t)eb[4 * w + 1] << 16) | ((uint32_t)eb[4 * w + 2] << 8) | eb[4 * w + 3];
                        gents.push_back(ge);
                        if (k >= 1)
                            for (uint64_t o = 0; o < sl; o += (1u << 20))
                                spread.push_back(SpreadPiece{spread_bytes + k * gcm_seg + o, so_ + o, (uint32_t)std::min<uint64_t>(1u << 20, sl - o)});
                    }
                    if (K > 1) spread_bytes += (plen + 15) & ~(uint64_t)15;
                    plen += 16 * K;
                }
                // the FDAT chunks: CH bytes each, the last one the rest (an empty payload is one empty chunk)
                const uint64_t K = plen ? (plen + CH - 1) / CH : 1;
                if (K > 1) spread_copy.emplace_back(p0, plen);
                for (uint64_t k = 0; k < K; k++) {
                    const uint64_t cl = std::min<uint64_t>(CH, plen - k * CH), cstart = p0 + k * (CH + 12);   // the chunk's data in the archive
                    FrameDesc u;
                    if (k == 0) { u = f0; u.arc_off = pos; }
                    else { u.prefix_off = (uint32_t)blob_len; u.prefix_len = 8; u.arc_off = cstart - 8; memcpy(blob + blob_len + 4, "FDAT", 4); blob_len += 8; }
                    uint8_t *lenf = &blob[u.prefix_off + u.prefix_len - 8];  // FDAT chunk length, big-endian
                    lenf[0] = (uint8_t)(cl >> 24); lenf[1] = (uint8_t)(cl >> 16); lenf[2] = (uint8_t)(cl >> 8); lenf[3] = (uint8_t)cl;
                    u.payload_len = (uint32_t)cl; u.pad = k + 1 < K ? 2u : 0u;
                    units.push_back(u);
                    if (k >= 1)
                        for (uint64_t o = 0; o < cl; o += (1u << 20))
                            spread.push_back(SpreadPiece{spread_bytes + k * CH + o, cstart + o, (uint32_t)std::min<uint64_t>(1u << 20, cl - o)});
                    if (ctr)                                       // CTR keeps the length and may be cut anywhere: the keystream position runs on over the chunks
                        for (uint64_t o = 0; o < cl; o += CTR_UNIT)
                            cunits.push_back(CipherUnit{cstart + o, k * CH + o, (uint32_t)std::min<uint64_t>(CTR_UNIT, cl - o), (uint32_t)(e - e0)});
                }
                if (K > 1) spread_bytes += (plen + 15) & ~(uint64_t)15;
                pos += f0.prefix_len + plen + 12 * (K - 1) + 4 + 12;
            }
            nunit = units.size();
            memcpy(fds, units.data(), nunit * sizeof(FrameDesc));
        }
        segdst[nseg] = pos;
        total = pos - out_base;
        if (pos + 16 > dst_cap) return fail(c, PNA_E_DSTSIZE, "device destination too small");
        if (c->fr_desc.ensure(nunit * sizeof(FrameDesc)) || c->fr_blob.ensure(blob_len + 16) || c->fr_segdst.ensure((size_t)(nseg + 1) * 8))
            return fail(c, PNA_E_NOMEM, "framing workspace");
        int rc = ensure_crc(c); if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->fr_desc.p, fds, nunit * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->fr_blob.p, blob, blob_len, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->fr_segdst.p, segdst, (size_t)(nseg + 1) * 8, hipMemcpyHostToDevice, st));
        d_segdst = (const uint64_t *)c->fr_segdst.p; wbase = d_dst;
    } else if (!early_write && out_base + total > dst_cap) return fail(c, PNA_E_DSTSIZE, "device destination too small");
    if (defl) launch_deflate_write(d_src, c->d_segs, c->d_blk_seg, nblk, (const BlkInfo *)c->blk.p,
                                   d_segdst, (const uint64_t *)c->seg_size.p, (const uint8_t *)c->litc.p, c->d_entry_seg,
                                   (uint32_t)(e1 - e0), wbase, st, c->call_stored);
    else launch_write(d_src, c->d_segs, nseg, c->d_blk_seg, nblk, (const BlkInfo *)c->blk.p,
                 (const SegTables *)c->tabs.p, d_segdst, (const uint8_t *)c->lits.p,
                 (const uint8_t *)c->litc.p, (const uint8_t *)c->seqc.p, wbase, any_empty, st);
    if (!spread.empty()) {
        // entries of several FDAT chunks / GCM segments: save the compact payloads, then put the pieces behind the first one at their places
        std::vector<PlaceDescH> pd(spread.size());
        for (size_t i = 0; i < spread.size(); i++) pd[i] = PlaceDescH{spread[i].src, spread[i].dst, spread[i].len, 0};
        if (c->ci_spread.ensure(spread_bytes + 64) || c->ci_spread_desc.ensure(pd.size() * sizeof(PlaceDescH) + 16)) return fail(c, PNA_E_NOMEM, "chunk workspace");
        uint64_t sp = 0;
        for (auto &cp : spread_copy) { HIPCHK(c, hipMemcpyAsync((uint8_t *)c->ci_spread.p + sp, d_dst + cp.first, cp.second, hipMemcpyDeviceToDevice, st)); sp += (cp.second + 15) & ~(uint64_t)15; }
        HIPCHK(c, hipMemcpyAsync(c->ci_spread_desc.p, pd.data(), pd.size() * sizeof(PlaceDescH), hipMemcpyHostToDevice, st));
        launch_gather(c->ci_spread_desc.p, (uint32_t)pd.size(), (const uint8_t *)c->ci_spread.p, d_dst, st);
        HIPCHK(c, hipStreamSynchronize(st));                      // pd goes out of scope
    }
    if (fj && fj->cipher) {
        if (solid && fj->cipher->cipher_mode == PNA_MODE_CBC) return fail(c, PNA_E_UNSUPPORTED, "solid archives: CTR and GCM on the device path (CBC encryption is one serial chain over the whole stream)");
        int rc = ensure_aes(c); if (rc) return rc;
        if (c->ci_units.ensure(cunits.size() * sizeof(CipherUnit) + 16) || c->ci_ivs.ensure((e1 - e0) * 16 + 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
        AesKey key; aes256_expand(fj->cipher->key, key);
        HIPCHK(c, hipMemcpyAsync(c->ci_units.p, cunits.data(), cunits.size() * sizeof(CipherUnit), hipMemcpyHostToDevice, st));
        std::vector<uint8_t> giv; std::vector<AesKey> gkeys;
        if (gcm) {
            giv.resize(gsegs.size() * 16); gkeys.resize(gsegs.size());
            for (size_t i = 0; i < gsegs.size(); i++) { memcpy(&giv[16 * i], gsegs[i].ctr_iv, 16); gkeys[i] = gmat[gsegs[i].entry].rk; }
            if (c->ci_keys.ensure(gkeys.size() * sizeof(AesKey) + 16) || c->ci_gcm.ensure(gents.size() * sizeof(GcmEntry) + 16) || c->ci_ivs.ensure(giv.size() + 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
            HIPCHK(c, hipMemcpyAsync(c->ci_ivs.p, giv.data(), giv.size(), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->ci_keys.p, gkeys.data(), gkeys.size() * sizeof(AesKey), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->ci_gcm.p, gents.data(), gents.size() * sizeof(GcmEntry), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));                  // the host vectors above go out of scope
        } else HIPCHK(c, hipMemcpyAsync(c->ci_ivs.p, fj->ivs + 16 * e0, (e1 - e0) * 16, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipEventRecord(c->ev_ci[0], st));
        if (gcm) {
            launch_aes_ctr((const CipherUnit *)c->ci_units.p, (uint32_t)cunits.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, d_dst, key, (const AesKey *)c->ci_keys.p, st);
            launch_gcm_tag((const GcmEntry *)c->ci_gcm.p, (uint32_t)gents.size(), d_dst, st);
        } else if (fj->cipher->cipher_mode == PNA_MODE_CTR)
            launch_aes_ctr((const CipherUnit *)c->ci_units.p, (uint32_t)cunits.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, d_dst, key, nullptr, st);
        else
            launch_aes_cbc_enc((const CipherUnit *)c->ci_units.p, (uint32_t)cunits.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, d_dst, key, st);
        HIPCHK(c, hipEventRecord(c->ev_ci[1], st));
    }
    if (timed) HIPCHK(c, hipEventRecord(c->ev[6], st));
    if (fj) launch_frame((const FrameDesc *)c->fr_desc.p, (uint32_t)nunit, (const uint8_t *)c->fr_blob.p, (const CrcTabs *)c->crc_tabs.p,
                         d_dst, (uint64_t)dst_cap & ~(uint64_t)15, frame_fend_crc(), solid ? "SDAT" : "FDAT", !solid, st, frame_max_payload);
    if (timed) HIPCHK(c, hipEventRecord(c->ev[7], st));
    HIPCHK(c, hipGetLastError());
    if (early_write) {
        HIPCHK(c, hipMemcpyAsync(c->h_segoff.p, c->seg_off.p, (size_t)(nseg + 1) * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        total = seg_off[nseg];
    }
    if (!fj) for (size_t e = e0; e < e1; e++) dst_off[e] = out_base + seg_off[entry_first_seg[e - e0]];
    dst_off[e1] = out_base + total;
    c->last_nblk = nblk;
    if (fj && !timed) HIPCHK(c, hipStreamSynchronize(st));        // the staging buffers are reused by the next sub-batch
    if (timed) {
        HIPCHK(c, hipStreamSynchronize(st));
        int rct = collect_timing(c, defl, nch, nseg, nblk, fj && fj->cipher);
        if (rct) return rct;
    }
    return PNA_OK;
}

extern "C" int pna_gpu_compress_batch_device(pna_gpu_ctx *c, int algo, int level, size_t n, const void *d_src,
                                             const uint64_t *src_off, const uint64_t *src_len, void *d_dst, size_t dst_cap,
                                             uint64_t *dst_off, void *hip_stream) {
    if (!c || !src_off || !src_len || !dst_off || (!d_src && n) || (!d_dst && n)) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    set_call_level(c, algo, level);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->timing = pna_gpu_timing{};
    dst_off[0] = 0;
    uint64_t out_base = 0, in_total = 0;
    size_t e = 0;
    plan_call(c, src_len, n);
    while (e < n) {
        size_t e1 = e; size_t blocks = 0;
        while (e1 < n) {
            size_t nb = plan_blocks(c, src_len[e1]);
            if (e1 > e && blocks + nb > c->max_blocks) break;
            blocks += nb; in_total += src_len[e1]; e1++;
        }
        int rc = run_subbatch(c, algo, (const uint8_t *)d_src, src_off, src_len, e, e1, (uint8_t *)d_dst, dst_cap, out_base, dst_off, st, true);
        if (rc) return rc;
        out_base = dst_off[e1];
        e = e1;
    }
    HIPCHK(c, hipStreamSynchronize(st));
    c->timing.in_bytes = in_total; c->timing.out_bytes = out_base;
    return PNA_OK;
}

extern "C" size_t pna_gpu_archive_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len) {
    size_t b = 28 + 12 + 64;                                    // signature + AHED, AEND, alignment slack of the CRC reads
    for (size_t i = 0; i < n; i++) {
        const size_t pb = pna_gpu_bound(algo, (size_t)src_len[i]);
        b += frame_entry_prefix_bound(names ? names[i] : nullptr) + pb + 16 + 12 * (src_len[i] >> 20);    // + one FDAT header / CRC per possible cut
    }
    return b;
}

// Non-solid `pna create` with the archive assembled in HBM: create_archive_file (cli/src/command/create.rs:575-635) +
// Archive::write_header / add_entry / finalize (lib/src/archive/write.rs) for file entries carrying FHED, fSIZ, FDAT, FEND.
extern "C" int pna_gpu_create_archive_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                             const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                             void *d_dst, size_t dst_cap, uint64_t *entry_off, uint64_t *archive_len,
                                             void *hip_stream) {
    return pna_gpu_create_archive_part_device(c, algo, level, n, names, d_src, src_off, src_len, d_dst, dst_cap, entry_off, archive_len,
                                              PNA_PART_HEAD | PNA_PART_TAIL, hip_stream);
}

// One shard of an archive whose entries are split over several producers (ranks): only the first shard carries the signature +
// AHED, only the last one AEND; the shards' outputs concatenated in entry order are the archive.
extern "C" int pna_gpu_create_archive_part_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                  const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                  void *d_dst, size_t dst_cap, uint64_t *entry_off, uint64_t *archive_len,
                                                  uint32_t part_flags, void *hip_stream) {
    return pna_gpu_create_archive_enc_device(c, algo, level, n, names, d_src, src_off, src_len, nullptr, d_dst, dst_cap, entry_off, archive_len,
                                             part_flags, hip_stream);
}

extern "C" size_t pna_gpu_archive_enc_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len, const pna_gpu_cipher *cipher) {
    size_t b = pna_gpu_archive_bound(algo, n, names, src_len);
    if (cipher && cipher->encryption != PNA_ENC_NONE && cipher->phsf) {
        b += n * (12 + strlen(cipher->phsf) + 12 + 75 + 16);   // PHSF, FDAT(iv | stream header), CBC padding / the final GCM tag
        if (cipher->cipher_mode == PNA_MODE_GCM) {               // one more tag per full stream segment of the (bounded) payload
            const uint64_t seg = cipher->gcm_segment_size ? cipher->gcm_segment_size : (1u << 20);
            for (size_t i = 0; i < n; i++) b += 16 * (pna_gpu_bound(algo, (size_t)src_len[i]) / seg);
        }
    }
    return b;
}

// The same with the cipher stage between the write kernels and the chunk CRC (get_writer: compress -> cipher -> sink,
// lib/src/entry/write.rs:268-274): entry record FHED | fSIZ | PHSF | FDAT(iv) | FDAT(ciphertext) | FEND.
extern "C" int pna_gpu_create_archive_enc_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                 const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                 const pna_gpu_cipher *cipher, void *d_dst, size_t dst_cap, uint64_t *entry_off,
                                                 uint64_t *archive_len, uint32_t part_flags, void *hip_stream) {
    return pna_gpu_create_archive_meta_device(c, algo, level, n, names, d_src, src_off, src_len, cipher, nullptr, d_dst, dst_cap, entry_off, archive_len,
                                              part_flags, hip_stream);
}

static int check_meta(pna_gpu_ctx *c, const pna_gpu_entry_meta *meta, size_t n) {
    if (!meta) return PNA_OK;
    if ((meta->extra && !meta->extra_len) || (meta->facets && !meta->facets_len)) return fail(c, PNA_E_INVAL, "metadata blobs without lengths");
    for (size_t e = 0; e < n; e++) {
        if (meta->extra && meta->extra_len[e] && (!meta->extra[e] || !meta_blob_ok((const uint8_t *)meta->extra[e], meta->extra_len[e]))) return fail(c, PNA_E_INVAL, "extra chunks of an entry are not well-formed chunks");
        if (meta->facets && meta->facets_len[e] && (!meta->facets[e] || !meta_blob_ok((const uint8_t *)meta->facets[e], meta->facets_len[e]))) return fail(c, PNA_E_INVAL, "metadata chunks of an entry are not well-formed chunks");
    }
    return PNA_OK;
}

// ... and with per-entry metadata: chunks the host has already framed (timestamps, permissions, owner, xattr: try_for_each_metadata_facet,
// lib/src/entry.rs:124-180; user-defined extra chunks) are placed where NormalEntry::write_chunks_to puts them.
static int create_archive_device_impl(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                      const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                      const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, uint32_t max_chunk, void *d_dst, size_t dst_cap,
                                      uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream);
extern "C" int pna_gpu_create_archive_meta_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                  const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                  const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, void *d_dst, size_t dst_cap,
                                                  uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream) {
    return create_archive_device_impl(c, algo, level, n, names, d_src, src_off, src_len, cipher, meta, c ? (uint32_t)c->tun.max_chunk_size : 0u, d_dst, dst_cap, entry_off, archive_len, part_flags, hip_stream);
}
extern "C" int pna_gpu_create_archive_chunked_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                     const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                     const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, uint32_t max_chunk_size, void *d_dst, size_t dst_cap,
                                                     uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream) {
    return create_archive_device_impl(c, algo, level, n, names, d_src, src_off, src_len, cipher, meta, max_chunk_size, d_dst, dst_cap, entry_off, archive_len, part_flags, hip_stream);
}
extern "C" size_t pna_gpu_archive_chunked_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len, const pna_gpu_cipher *cipher, uint32_t max_chunk_size) {
    size_t b = cipher && cipher->encryption != PNA_ENC_NONE ? pna_gpu_archive_enc_bound(algo, n, names, src_len, cipher) : pna_gpu_archive_bound(algo, n, names, src_len);
    const uint64_t CH = chunk_limit(max_chunk_size);
    for (size_t i = 0; i < n; i++) b += 12 * (size_t)((pna_gpu_bound(algo, (size_t)src_len[i]) + 64 + 16 * (src_len[i] >> 12)) / CH + 1);   // a CRC + a header per further chunk
    return b;
}
static int create_archive_device_impl(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                      const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                      const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, uint32_t max_chunk, void *d_dst, size_t dst_cap,
                                      uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream) {
    { int rcm = check_meta(c, meta, n); if (rcm) return rcm; }
    if (!c || !archive_len || (n && (!names || !src_off || !src_len || !d_src)) || !d_dst) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    if ((uintptr_t)d_dst & 15) return fail(c, PNA_E_INVAL, "archive buffer must be 16-byte aligned");
    if (cipher && cipher->encryption == PNA_ENC_NONE) cipher = nullptr;
    std::vector<uint8_t> own_ivs;
    const uint8_t *ivs = nullptr;
    if (cipher) { int rc = resolve_ivs(c, cipher, n, own_ivs, &ivs); if (rc) return rc; }
    set_call_level(c, algo, level);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->timing = pna_gpu_timing{};
    std::vector<uint8_t> head, tail;
    if (part_flags & PNA_PART_HEAD) frame_archive_head(head, 0);
    if (part_flags & PNA_PART_TAIL) frame_archive_tail(tail);
    if (head.size() + tail.size() + 16 > dst_cap) return fail(c, PNA_E_DSTSIZE, "device destination too small");
    if (!head.empty()) HIPCHK(c, hipMemcpyAsync(d_dst, head.data(), head.size(), hipMemcpyHostToDevice, st));
    std::vector<uint64_t> offs(n + 1);
    uint64_t pos = head.size(), in_total = 0;
    FrameJob fj{names, 0, cipher, ivs, meta, max_chunk, entry_off != nullptr};
    size_t e = 0;
    plan_call(c, src_len, n);
    while (e < n) {
        size_t e1 = e, blocks = 0;
        while (e1 < n) {
            size_t nb = plan_blocks(c, src_len[e1]);
            if (e1 > e && blocks + nb > c->max_blocks) break;
            blocks += nb; in_total += src_len[e1]; e1++;
        }
        int rc = run_subbatch(c, algo, (const uint8_t *)d_src, src_off, src_len, e, e1, (uint8_t *)d_dst, dst_cap - tail.size(), pos, offs.data(), st, true, &fj);
        if (rc) return rc;
        pos = offs[e1];
        e = e1;
    }
    offs[n] = pos;
    if (!tail.empty()) HIPCHK(c, hipMemcpyAsync((uint8_t *)d_dst + pos, tail.data(), tail.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    pos += tail.size();
    if (entry_off) memcpy(entry_off, offs.data(), (n + 1) * 8);
    *archive_len = pos;
    c->timing.in_bytes = in_total; c->timing.out_bytes = pos;
    return PNA_OK;
}

// The cipher stage alone over byte ranges of a device buffer (read side: DecryptReader::CtrAes, lib/src/entry/read.rs:83-88).
extern "C" int pna_gpu_cipher_apply_device(pna_gpu_ctx *c, const pna_gpu_cipher *cipher, int decrypt, size_t n, void *d_buf,
                                           const uint64_t *off, const uint64_t *len, void *hip_stream) {
    if (!c || !cipher || (n && (!d_buf || !off || !len || !cipher->ivs))) return fail(c, PNA_E_INVAL, "null argument");
    int rc = check_cipher(c, cipher); if (rc) return rc;
    const bool cbc = cipher->cipher_mode == PNA_MODE_CBC;
    if (cipher->cipher_mode == PNA_MODE_GCM) return fail(c, PNA_E_UNSUPPORTED, "GCM STREAM is offered by the archive entry points only");
    if (cbc && decrypt) return fail(c, PNA_E_UNSUPPORTED, "CBC decryption is not offered on the device path");
    if (n == 0) return PNA_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    rc = ensure_aes(c); if (rc) return rc;
    std::vector<CipherUnit> units;
    for (size_t i = 0; i < n; i++) {
        if (len[i] >= 0xFFFFFFF0ull) return fail(c, PNA_E_INVAL, "cipher range too long");
        if (cbc) units.push_back(CipherUnit{off[i], 0, (uint32_t)len[i], (uint32_t)i});
        else for (uint64_t o = 0; o < len[i]; o += CTR_UNIT) units.push_back(CipherUnit{off[i] + o, o, (uint32_t)std::min<uint64_t>(CTR_UNIT, len[i] - o), (uint32_t)i});
    }
    if (c->ci_units.ensure(units.size() * sizeof(CipherUnit) + 16) || c->ci_ivs.ensure(n * 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
    AesKey key; aes256_expand(cipher->key, key);
    c->timing = pna_gpu_timing{};
    HIPCHK(c, hipMemcpyAsync(c->ci_units.p, units.data(), units.size() * sizeof(CipherUnit), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->ci_ivs.p, cipher->ivs, n * 16, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev_ci[0], st));
    if (cbc) launch_aes_cbc_enc((const CipherUnit *)c->ci_units.p, (uint32_t)units.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, (uint8_t *)d_buf, key, st);
    else launch_aes_ctr((const CipherUnit *)c->ci_units.p, (uint32_t)units.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, (uint8_t *)d_buf, key, nullptr, st);
    HIPCHK(c, hipEventRecord(c->ev_ci[1], st));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    float mc = 0; (void)hipEventElapsedTime(&mc, c->ev_ci[0], c->ev_ci[1]);
    c->timing.ms_cipher = mc;
    for (size_t i = 0; i < n; i++) c->timing.in_bytes += len[i];
    return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// `pna create --solid` with the archive assembled in HBM (create_archive_file's solid branch, cli/src/command/create.rs:594-598,
// 603-617; SolidArchive / SolidEntryBuilder, lib/src/archive/write.rs:443-470,575-580,716-727):
//   1. the inner entries are serialised as STORE records FHED | fSIZ | FDAT | FEND (their FDAT CRC-32 computed by k_frame) into
//      one stream in HBM,
//   2. that stream is compressed as ONE entry (independent 1 MiB frames / zlib blocks inside the kernels),
//   3. every segment's output becomes one SDAT chunk between SHED and SEND.

extern "C" size_t pna_gpu_solid_archive_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len) {
    uint64_t plain = 0;
    for (size_t i = 0; i < n; i++) plain += frame_entry_prefix_bound(names ? names[i] : nullptr) + src_len[i] + 16;
    const uint64_t segs = (plain + SEG_SIZE - 1) / SEG_SIZE + 1;
    return 28 + 17 + pna_gpu_bound(algo, (size_t)plain) + 12 * segs + 12 + 12 + 64;
}

extern "C" size_t pna_gpu_solid_archive_enc_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len, const pna_gpu_cipher *cipher) {
    size_t b = pna_gpu_solid_archive_bound(algo, n, names, src_len);
    if (!cipher || cipher->encryption == PNA_ENC_NONE) return b;
    b += 12 + (cipher->phsf ? strlen(cipher->phsf) : 0) + 12 + 75 + 64;          // PHSF chunk, the chunk of the IV / stream header
    if (cipher->cipher_mode == PNA_MODE_GCM) {
        const uint64_t seg = cipher->gcm_segment_size ? cipher->gcm_segment_size : (1u << 20);
        b += 28 * (size_t)(b / seg + 2);                                            // tag + chunk framing per GCM segment
    }
    return b;
}

extern "C" int pna_gpu_create_solid_archive_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                   const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                   void *d_dst, size_t dst_cap, uint64_t *archive_len, void *hip_stream) {
    return pna_gpu_create_solid_archive_enc_device(c, algo, level, n, names, d_src, src_off, src_len, nullptr, d_dst, dst_cap, archive_len, hip_stream);
}

// With a cipher (CTR): SHED(encryption, cipher_mode) | PHSF | SDAT(iv) | SDAT(ciphertext)* | SEND -- into_solid_archive writes the PHSF
// chunk behind SHED and the IV as the first write of the SDAT stream (lib/src/archive/write.rs:443-470); one cipher stream runs over
// all SDAT bodies.  cipher->ivs: ONE 16-byte IV (or NULL).
extern "C" int pna_gpu_create_solid_archive_enc_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                       const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                       const pna_gpu_cipher *cipher, void *d_dst, size_t dst_cap, uint64_t *archive_len,
                                                       void *hip_stream) {
    if (cipher && cipher->encryption == PNA_ENC_NONE) cipher = nullptr;
    std::vector<uint8_t> own_ivs;
    const uint8_t *ivs = nullptr;
    if (cipher) { int rc0 = resolve_ivs(c, cipher, 1, own_ivs, &ivs); if (rc0) return rc0; }
    if (cipher && cipher->cipher_mode == PNA_MODE_CBC) return fail(c, PNA_E_UNSUPPORTED, "solid archives: CTR and GCM on the device path (CBC encryption is one serial chain over the whole stream)");
    if (!c || !archive_len || (n && (!names || !src_off || !src_len || !d_src)) || !d_dst) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    if ((uintptr_t)d_dst & 15) return fail(c, PNA_E_INVAL, "archive buffer must be 16-byte aligned");
    set_call_level(c, algo, level);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    // ---- 1. layout of the serialised inner entries (all sizes are known up front)
    std::vector<FrameDesc> fds(n); std::vector<uint8_t> blob; std::vector<PlaceDescH> places;
    uint64_t pos = 0, max_inner = 0;
    for (size_t i = 0; i < n; i++) max_inner = std::max<uint64_t>(max_inner, src_len[i]);
    const uint32_t solid_max_inner = max_inner <= 16380 ? (uint32_t)std::max<uint64_t>(max_inner, 1) : 0u;      // (stored inner entries: the FDAT payload is the entry; small ones take k_frame's wave-per-entry form)
    for (size_t i = 0; i < n; i++) {
        if (src_off[i] & 15) return fail(c, PNA_E_INVAL, "entry offset not 16-byte aligned");
        if (src_len[i] >= 0x7FFF0000ull) return fail(c, PNA_E_INVAL, "inner entry too large for one FDAT chunk");
        const size_t po = blob.size();
        if (src_len[i] == 0) {
            frame_inner_entry_empty(blob, names[i]);
            fds[i] = FrameDesc{pos, 0, (uint32_t)po, (uint32_t)(blob.size() - po), 1};
            pos += blob.size() - po;
            continue;
        }
        frame_entry_prefix(blob, names[i], PNA_ALGO_STORE, src_len[i], (uint32_t)src_len[i]);
        const uint32_t pl = (uint32_t)(blob.size() - po);
        fds[i] = FrameDesc{pos, (uint32_t)src_len[i], (uint32_t)po, pl, 0};
        for (uint64_t k = 0; k < src_len[i]; k += SEG_SIZE)
            places.push_back(PlaceDescH{src_off[i] + k, pos + pl + k, (uint32_t)std::min<uint64_t>(SEG_SIZE, src_len[i] - k), 0});
        pos += pl + src_len[i] + 16;
    }
    const uint64_t plain_len = pos;
    if (c->solid_plain.ensure(plain_len + 8192) || c->solid_desc.ensure(n * sizeof(FrameDesc) + 16) || c->solid_blob.ensure(blob.size() + 16) ||
        c->solid_place.ensure(places.size() * sizeof(PlaceDescH) + 16)) return fail(c, PNA_E_NOMEM, "solid workspace");
    int rc = ensure_crc(c); if (rc) return rc;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->solid_desc.p, fds.data(), n * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->solid_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
        if (!places.empty()) HIPCHK(c, hipMemcpyAsync(c->solid_place.p, places.data(), places.size() * sizeof(PlaceDescH), hipMemcpyHostToDevice, st));
        launch_place(c->solid_place.p, (uint32_t)places.size(), (const uint8_t *)d_src, (uint8_t *)c->solid_plain.p, st);
        launch_frame((const FrameDesc *)c->solid_desc.p, (uint32_t)n, (const uint8_t *)c->solid_blob.p, (const CrcTabs *)c->crc_tabs.p,
                     (uint8_t *)c->solid_plain.p, (uint64_t)c->solid_plain.cap & ~(uint64_t)15, frame_fend_crc(), "FDAT", true, st, solid_max_inner);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));                     // the host vectors above are read by the async copies
    }
    // ---- 2 + 3. one entry -> SDAT chunks, between the fixed chunks
    c->timing = pna_gpu_timing{};
    std::vector<uint8_t> head, tail;
    frame_archive_head(head, 0);
    if (cipher && cipher->cipher_mode == PNA_MODE_GCM) {
        // the stream header (salt || nonce prefix || segment size || key confirmation: 75 bytes) is the stream's first write, a chunk of its own
        GcmMaterial gm; uint8_t kc[32], ph[32];
        hkdf_sha256_32(cipher->key, 32, nullptr, 0, "PNA-KC-v1", 9, kc);
        sha256_bytes(cipher->phsf, strlen(cipher->phsf), nullptr, 0, ph);
        gcm_entry_material(cipher, kc, ph, ivs, cipher->gcm_segment_size ? cipher->gcm_segment_size : (1u << 20), nullptr, algo, gm);
        frame_solid_head_enc(head, algo, cipher->encryption, cipher->cipher_mode, cipher->phsf, gm.header, 75);
    } else if (cipher) frame_solid_head_enc(head, algo, cipher->encryption, cipher->cipher_mode, cipher->phsf, ivs, 16); else frame_solid_head(head, algo);
    frame_solid_tail(tail); frame_archive_tail(tail);
    if (head.size() + tail.size() + 64 > dst_cap) return fail(c, PNA_E_DSTSIZE, "device destination too small");
    HIPCHK(c, hipMemcpyAsync(d_dst, head.data(), head.size(), hipMemcpyHostToDevice, st));
    const uint64_t off0 = 0, len0 = plain_len; uint64_t offs[2] = {0, 0};
    FrameJob fj{nullptr, 1, cipher, ivs};
    rc = run_subbatch(c, algo, (const uint8_t *)c->solid_plain.p, &off0, &len0, 0, 1, (uint8_t *)d_dst, dst_cap - tail.size(), head.size(), offs, st, true, &fj);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync((uint8_t *)d_dst + offs[1], tail.data(), tail.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *archive_len = offs[1] + tail.size();
    uint64_t in_total = 0; for (size_t i = 0; i < n; i++) in_total += src_len[i];
    c->timing.in_bytes = in_total; c->timing.out_bytes = *archive_len;
    return PNA_OK;
}

static void parallel_stage(uint8_t *dst, const void *const *src, const size_t *src_len, const uint64_t *off, size_t e0, size_t e1, unsigned threads);

// The same from host memory: one H2D of the entries, the device path above, one D2H of the archive, handed to the sink in
// pieces of at most 16 MiB.  (The whole solid stream is in flight at once: a solid entry is one compression unit.)
extern "C" int pna_gpu_create_solid_archive_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                 const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user) {
    if (!c || !sink || (n && (!names || !src || !src_len))) return fail(c, PNA_E_INVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> off(n + 1), len(n);
    uint64_t pos = 0;
    for (size_t i = 0; i < n; i++) { off[i] = pos; len[i] = src_len[i]; pos = (pos + src_len[i] + 15) & ~(uint64_t)15; }
    off[n] = pos;
    const size_t cap = pna_gpu_solid_archive_bound(algo, n, names, len.data());
    if (c->stage_in.ensure(pos + 8192) || c->stage_out.ensure(cap + 64) || c->hp_in[0].ensure(pos + 64) || c->hp_out[0].ensure(cap + 64))
        return fail(c, PNA_E_NOMEM, "staging allocation failed");
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    parallel_stage((uint8_t *)c->hp_in[0].p, src, src_len, off.data(), 0, n, std::min(8u, std::max(1u, hw / 2)));
    if (pos) HIPCHK(c, hipMemcpyAsync(c->stage_in.p, c->hp_in[0].p, pos, hipMemcpyHostToDevice, c->stream));
    uint64_t total = 0;
    int rc = pna_gpu_create_solid_archive_device(c, algo, level, n, names, c->stage_in.p, off.data(), len.data(), c->stage_out.p, cap + 64, &total, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->hp_out[0].p, c->stage_out.p, total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint64_t p = 0; p < total; p += (16u << 20)) {
        const size_t k = (size_t)std::min<uint64_t>(16u << 20, total - p);
        if (sink(user, (const uint8_t *)c->hp_out[0].p + p, k) != 0) return fail(c, PNA_E_SINK, "sink failed");
    }
    return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Host-memory `pna create` (non-solid), bounded memory: the entries stream through two staging slots of at most
// ~1 GiB of input each.  While the GPU compresses and frames sub-batch k, helper threads stage sub-batch k+1 into
// page-locked memory and its H2D copy runs on a second stream; the archive bytes of sub-batch k-1 travel back on a
// third stream and are handed to the sink in one piece.  Replaces the reference's "every compressed entry in RAM until
// the scope ends" (cli/src/command/core.rs:496-537, create.rs:575-635) with a fixed in-flight window.
// ---- zero-staging input (round 4).  cli/src/command/core.rs:889-913 write_from_path reads every file into memory the library could own: with
// pna_gpu_host_alloc the host gets PAGE-LOCKED buffers to read its files into (read_exact into the slot instead of fs::read into a Vec), and the create
// entry points send entries that lie in such a buffer to the device straight from there -- the pageable -> page-locked copy on eight host threads is gone,
// one thread issues the copies.  Any mix works: a batch with an entry elsewhere is staged as before.
extern "C" int pna_gpu_host_alloc(pna_gpu_ctx *c, size_t bytes, void **out) {
    if (!c || !out || !bytes) return fail(c, PNA_E_INVAL, "null argument");
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return fail(c, PNA_E_NOMEM, "page-locked allocation failed");
    { std::lock_guard<std::mutex> lk(c->lent_mu); c->lent.emplace_back((const uint8_t *)p, bytes); }
    *out = p;
    return PNA_OK;
}
extern "C" int pna_gpu_host_free(pna_gpu_ctx *c, void *p) {
    if (!c || !p) return fail(c, PNA_E_INVAL, "null argument");
    {
        std::lock_guard<std::mutex> lk(c->lent_mu);
        auto it = std::find_if(c->lent.begin(), c->lent.end(), [&](const std::pair<const uint8_t *, size_t> &b) { return b.first == (const uint8_t *)p; });
        if (it == c->lent.end()) return fail(c, PNA_E_INVAL, "not a buffer of pna_gpu_host_alloc");
        c->lent.erase(it);
    }
    (void)hipSetDevice(c->device);
    (void)hipHostFree(p);
    return PNA_OK;
}
static bool entries_all_lent(pna_gpu_ctx *c, const void *const *src, const size_t *src_len, size_t n) {
    std::lock_guard<std::mutex> lk(c->lent_mu);
    if (c->lent.empty() || !n) return false;
    size_t hint = 0;
    for (size_t i = 0; i < n; i++) {
        if (!src_len[i]) continue;
        const uint8_t *p = (const uint8_t *)src[i];
        bool in = false;
        for (size_t k = 0; k < c->lent.size() && !in; k++) {                            // (entries of one call mostly share a buffer: start with the last hit)
            const auto &b = c->lent[(hint + k) % c->lent.size()];
            if (p >= b.first && p + src_len[i] <= b.first + b.second) { in = true; hint = (hint + k) % c->lent.size(); }
        }
        if (!in) return false;
    }
    return true;
}

static void parallel_stage(uint8_t *dst, const void *const *src, const size_t *src_len, const uint64_t *off, size_t e0, size_t e1, unsigned threads) {
    uint64_t total = 0;
    for (size_t e = e0; e < e1; e++) total += src_len[e];
    if (threads <= 1 || total < (8u << 20)) { for (size_t e = e0; e < e1; e++) if (src_len[e]) memcpy(dst + off[e], src[e], src_len[e]); return; }
    std::vector<std::thread> th;
    const uint64_t per = (total + threads - 1) / threads;
    size_t e = e0;
    for (unsigned t = 0; t < threads && e < e1; t++) {
        size_t b = e; uint64_t acc = 0;
        while (e < e1 && (acc < per || t + 1 == threads)) acc += src_len[e++];
        th.emplace_back([=]() { for (size_t i = b; i < e; i++) if (src_len[i]) memcpy(dst + off[i], src[i], src_len[i]); });
    }
    for (auto &x : th) x.join();
}

extern "C" int pna_gpu_create_archive_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                           const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user) {
    return pna_gpu_create_archive_enc_host(c, algo, level, n, names, src, src_len, nullptr, sink, user);
}

extern "C" int pna_gpu_create_archive_enc_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                               const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                               pna_sink_fn sink, void *user) {
    return pna_gpu_create_archive_meta_host(c, algo, level, n, names, src, src_len, cipher, nullptr, sink, user);
}

static int create_archive_host_impl(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                    const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                    const pna_gpu_entry_meta *meta, uint32_t part_flags, pna_sink_fn sink, void *user, uint32_t max_chunk);
extern "C" int pna_gpu_create_archive_meta_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                                const pna_gpu_entry_meta *meta, pna_sink_fn sink, void *user) {
    return create_archive_host_impl(c, algo, level, n, names, src, src_len, cipher, meta, PNA_PART_HEAD | PNA_PART_TAIL, sink, user, c ? (uint32_t)c->tun.max_chunk_size : 0u);
}
extern "C" int pna_gpu_create_archive_chunked_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                   const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                                   const pna_gpu_entry_meta *meta, uint32_t max_chunk_size, uint32_t part_flags, pna_sink_fn sink, void *user) {
    return create_archive_host_impl(c, algo, level, n, names, src, src_len, cipher, meta, part_flags, sink, user, max_chunk_size);
}
extern "C" int pna_gpu_create_archive_part_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                const void *const *src, const size_t *src_len, uint32_t part_flags, pna_sink_fn sink, void *user) {
    return create_archive_host_impl(c, algo, level, n, names, src, src_len, nullptr, nullptr, part_flags, sink, user, c ? (uint32_t)c->tun.max_chunk_size : 0u);
}
// `pna append`: Archive::seek_to_end, then the new entries and AEND where the old AEND stood (cli/src/command/append.rs:504-560)
extern "C" int pna_gpu_append_archive_host(pna_gpu_ctx *c, int algo, int level, const void *archive, size_t archive_len, size_t n,
                                           const char *const *names, const void *const *src, const size_t *src_len, uint64_t *write_at,
                                           pna_sink_fn sink, void *user) {
    if (!c || !archive || !write_at || !sink) return fail(c, PNA_E_INVAL, "null argument");
    int has_next = 0;
    if (pna_archive_seek_to_end(archive, archive_len, write_at, &has_next) != PNA_OK) return fail(c, PNA_E_INVAL, "not a PNA archive, or truncated before its AEND chunk");
    if (has_next) return fail(c, PNA_E_INVAL, "the archive continues in another part (ANXT): append to its last part");
    return create_archive_host_impl(c, algo, level, n, names, src, src_len, nullptr, nullptr, PNA_PART_TAIL, sink, user, (uint32_t)c->tun.max_chunk_size);
}
// One process, several GPUs (SURVEY §8(b)'s `device_ids, n_devices`; §8(e)'s comparison path in C): the entries are cut into contiguous
// index ranges balanced by bytes, one per context (= per device), every context runs the bounded host pipeline on its range on a thread of
// its own (part flags: the first range carries the archive header, the last AEND) into host memory, and the parts reach the sink in index
// order -- the reference's fan-out + ordered drain (cli/src/command/core.rs:496-537,471-493) with devices in place of rayon workers and no
// device-to-device traffic at all.  Contexts may share a device (that is how the one-GPU boxes test it).
extern "C" int pna_gpu_create_archive_multi_host(pna_gpu_ctx *const *ctxs, size_t n_ctx, int algo, int level, size_t n, const char *const *names,
                                                 const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user) {
    if (!ctxs || !n_ctx || !sink || (n && (!names || !src || !src_len))) return PNA_E_INVAL;
    for (size_t r = 0; r < n_ctx; r++) if (!ctxs[r]) return PNA_E_INVAL;
    if (n_ctx == 1) return pna_gpu_create_archive_host(ctxs[0], algo, level, n, names, src, src_len, sink, user);
    // contiguous ranges balanced by input bytes (shard.partition_entries)
    uint64_t total = 0; for (size_t i = 0; i < n; i++) total += src_len[i];
    std::vector<size_t> lo(n_ctx + 1, n);
    {   // range r ends where the running byte count passes r + 1 shares of the total (a range may be empty when there are few entries)
        size_t i = 0; uint64_t acc = 0;
        for (size_t r = 0; r < n_ctx; r++) {
            lo[r] = i;
            const uint64_t target = (uint64_t)((__uint128_t)total * (r + 1) / n_ctx);
            while (i < n && (r + 1 == n_ctx || acc + src_len[i] <= target)) { acc += src_len[i]; i++; }
        }
        lo[n_ctx] = n;
    }
    // Every range runs the bounded pipeline on a thread of its own; range 0 hands its pieces to the caller's sink as they come, the ranges behind it keep
    // theirs in memory until every range before them has finished (the sink sees the archive in index order, on the calling thread only).  An exception
    // inside a worker (allocation) is that range's PNA_E_NOMEM, not a terminate; the first failing range's code is returned and its message copied to
    // ctxs[0] (what pna_gpu_last_error of the first context reports).
    struct Part { std::vector<uint8_t> buf; int rc = PNA_OK; bool done = false; };
    std::vector<Part> parts(n_ctx);
    std::mutex mu; std::condition_variable cv;
    auto vec_sink = [](void *u, const void *b, size_t k) -> int { auto *v = (std::vector<uint8_t> *)u; try { v->insert(v->end(), (const uint8_t *)b, (const uint8_t *)b + k); } catch (...) { return 1; } return 0; };
    std::vector<std::thread> th;
    for (size_t r = 1; r < n_ctx; r++)
        th.emplace_back([&, r]() {
            int rc;
            try {
                const uint32_t pf = r + 1 == n_ctx ? PNA_PART_TAIL : 0u;
                rc = pna_gpu_create_archive_part_host(ctxs[r], algo, level, lo[r + 1] - lo[r], names + lo[r], src + lo[r], src_len + lo[r], pf, vec_sink, &parts[r].buf);
            } catch (...) { rc = PNA_E_NOMEM; }
            std::lock_guard<std::mutex> lk(mu);
            parts[r].rc = rc; parts[r].done = true; cv.notify_all();
        });
    int rc0;
    try { rc0 = pna_gpu_create_archive_part_host(ctxs[0], algo, level, lo[1] - lo[0], names + lo[0], src + lo[0], src_len + lo[0], PNA_PART_HEAD, sink, user); }
    catch (...) { rc0 = fail(ctxs[0], PNA_E_NOMEM, "out of memory"); }
    int rc = rc0;
    for (size_t r = 1; r < n_ctx; r++) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return parts[r].done; }); }
        if (rc == PNA_OK && parts[r].rc != PNA_OK) { rc = parts[r].rc; const std::string msg = std::string("range ") + std::to_string(r) + ": " + pna_gpu_last_error(ctxs[r]); (void)fail(ctxs[0], rc, msg.c_str()); }
        if (rc == PNA_OK && !parts[r].buf.empty() && sink(user, parts[r].buf.data(), parts[r].buf.size()) != 0) rc = fail(ctxs[0], PNA_E_SINK, "sink failed");
        std::vector<uint8_t>().swap(parts[r].buf);                // handed on (or abandoned): the memory goes back at once
    }
    for (auto &t : th) t.join();
    return rc;
}
static int create_archive_host_impl(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                    const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                    const pna_gpu_entry_meta *meta, uint32_t part_flags, pna_sink_fn sink, void *user, uint32_t max_chunk) {
    { int rcm = check_meta(c, meta, n); if (rcm) return rcm; }
    if (!c || !sink || (n && (!names || !src || !src_len))) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    set_call_level(c, algo, level);
    if (cipher && cipher->encryption == PNA_ENC_NONE) cipher = nullptr;
    std::vector<uint8_t> own_ivs;
    const uint8_t *ivs = nullptr;
    if (cipher) { int rc0 = resolve_ivs(c, cipher, n, own_ivs, &ivs); if (rc0) return rc0; }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->cp_in) {
        HIPCHK(c, hipStreamCreate(&c->cp_in)); HIPCHK(c, hipStreamCreate(&c->cp_out));
        for (auto &e : c->ev_in) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : c->ev_out) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    c->timing = pna_gpu_timing{};
    std::vector<uint8_t> head, tail;
    if (part_flags & PNA_PART_HEAD) frame_archive_head(head, 0);
    if (part_flags & PNA_PART_TAIL) frame_archive_tail(tail);
    if (!head.empty() && sink(user, head.data(), head.size()) != 0) return fail(c, PNA_E_SINK, "sink failed");
    // Sub-batches.  What bounds this path is the host link in the H2D direction (page-locked memory -> HBM: 56.8 GB/s on the MI355X boxes;
    // experiments/link_duplex.hip, profiles/r03_c_link_duplex.txt) -- the kernels take a sixth of that time, the archive bytes going back a third of the
    // volume and the link is full duplex.  So the pipeline is built around ONE rule: the H2D copy engine never waits.
    //   * a stager thread runs ahead through all sub-batches over a ring of four input slots (page-locked staging + device buffer): it copies
    //     the entries of a sub-batch into the slot's page-locked buffer (several threads, groups of ~128 MiB) and issues each group's H2D copy
    //     right behind it; it blocks only while all four slots are in use (a slot is free again when its sub-batch's kernels are done);
    //   * the main thread takes the sub-batches in order: kernels on the context's stream, then the archive bytes of the sub-batch travel to
    //     the host next to the following sub-batch's kernels and copies -- by a small copy KERNEL that stores into the page-locked buffer
    //     (d2h_wgs workgroups: ~30 GB/s, which leaves the H2D engine its full rate; the runtime's own D2H copy ran as a blit kernel at 51 GB/s
    //     and took 30 % off the H2D copies next to it) --, and are handed to the sink one sub-batch later;
    //   * sub-batch sizes grow from 64 MiB to `sub_mib` (default 256 MiB) at the start: the first kernels start after 2 ms instead of 20, and
    //     what is left to do when the last input byte has arrived is the work of one sub-batch.  256 MiB is the smallest size whose kernels
    //     (1.2 ms of fixed costs + 1 ms per 85 MiB) keep up with its H2D copy (1 ms per 53 MiB); shrinking sizes at the end only makes the
    //     kernels fall behind the copies (measured: option sub_ramp_down).
    const uint64_t SUBMAX = (uint64_t)c->tun.sub_mib << 20, SUBMIN = std::min<uint64_t>(64ull << 20, SUBMAX);
    struct Sub { size_t e0, e1; uint64_t in_bytes, out_cap; };
    std::vector<Sub> subs; std::vector<uint64_t> off(n + 1), len64(n);
    uint64_t in_total = 0;
    for (size_t e = 0; e < n; e++) in_total += src_len[e];
    plan_call(c, src_len, n);
    {
        uint64_t done = 0, target = SUBMIN;
        for (size_t e = 0; e < n;) {
            const uint64_t rest = in_total - done;
            uint64_t want = std::min(target, SUBMAX);
            if (c->tun.sub_ramp_down && rest < 2 * want) want = std::max(SUBMIN, rest / 2);   // (option) ramp down: half of what is left, not below the minimum
            (void)rest;
            Sub sb{e, e, 0, 64}; uint64_t pos = 0; size_t blocks = 0;
            while (sb.e1 < n) {
                const size_t i = sb.e1; const uint64_t l = src_len[i];
                const size_t nb = plan_blocks(c, l);
                if (i > sb.e0 && (pos + l > want || blocks + nb > c->max_blocks)) break;
                off[i] = pos; len64[i] = l; pos = (pos + l + 15) & ~(uint64_t)15; blocks += nb;
                sb.out_cap += (cipher ? frame_entry_prefix_enc_bound(names[i], cipher->phsf) + 16 : frame_entry_prefix_bound(names[i])) + meta_len(meta, i) + pna_gpu_bound(algo, (size_t)l) + 16;
                if (cipher && cipher->cipher_mode == PNA_MODE_GCM) sb.out_cap += 16 * (pna_gpu_bound(algo, (size_t)l) / (cipher->gcm_segment_size ? cipher->gcm_segment_size : (1u << 20)));   // a tag per full stream segment
                // a CRC + a header per further FDAT chunk once max_chunk_size cuts the payload (the term of pna_gpu_archive_chunked_bound: without it data
                // that does not compress overran the sub-batch's device buffer by 12 bytes per chunk -- PNA_E_DSTSIZE for a 2 MiB random entry at mcs = 1000)
                sb.out_cap += 12 * (uint64_t)((pna_gpu_bound(algo, (size_t)l) + 64 + 16 * (l >> 12)) / chunk_limit(max_chunk) + 1);
                done += l; sb.e1++;
            }
            sb.in_bytes = pos; subs.push_back(sb); e = sb.e1;
            target = std::min(SUBMAX, target * 2);
        }
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned threads = std::min(8u, std::max(1u, hw / 2));
    if (c->tun.stage_threads) threads = (unsigned)c->tun.stage_threads;
    FrameJob fj{names, 0, cipher, ivs, meta, max_chunk, false};
    std::vector<uint64_t> eoff(n + 1);
    uint64_t out_len[2] = {0, 0}, out_total = head.size();
    constexpr int NS = 4;
    {   // slots sized once for the largest sub-batch (allocation of page-locked memory is slow: not inside the pipeline)
        uint64_t max_in = 0, max_out = 0;
        for (const Sub &sb : subs) { max_in = std::max(max_in, sb.in_bytes); max_out = std::max(max_out, sb.out_cap); }
        const bool lent0 = entries_all_lent(c, src, src_len, n);                    // (then no page-locked staging of the library's own is needed)
        for (int s = 0; s < NS && s < (int)subs.size(); s++)
            if ((!lent0 && c->hp_in[s].ensure(max_in + 8192)) || c->dp_in[s].ensure(max_in + 8192)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
        for (int s = 0; s < 2 && s < (int)subs.size(); s++)
            if (c->dp_out[s].ensure(max_out + 64) || c->hp_out[s].ensure(max_out + 64)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    }
    int rc = PNA_OK;
    const auto tr0 = std::chrono::steady_clock::now();
    auto trace = [&](const char *what, size_t k) { if (c->tun.trace) fprintf(stderr, "[pna create] %8.3f ms  %s %zu\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(), what, k); };
    trace("planned, slots ready; sub-batches:", subs.size());
    // the stager: sub-batch k into slot k % NS as soon as sub-batch k - NS has left the device
    std::mutex mu; std::condition_variable cv;
    size_t staged = 0, freed = 0; int stager_rc = PNA_OK; bool stop = false;      // sub-batches staged (copies issued) / sub-batches whose kernels are done
    const int dev_id = c->device;
    hipStream_t cp_in = c->cp_in;
    const bool all_lent = entries_all_lent(c, src, src_len, n);
    std::thread stager([&]() {
        try {
            if (hipSetDevice(dev_id) != hipSuccess) { std::lock_guard<std::mutex> lk(mu); stager_rc = PNA_E_HIP; staged = subs.size(); cv.notify_all(); return; }
            for (size_t k = 0; k < subs.size(); k++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || k < freed + NS; });
                    if (stop) return;
                }
                const Sub &nx = subs[k]; const int sl = (int)(k % NS);
                uint8_t *hb = (uint8_t *)c->hp_in[sl].p, *db = (uint8_t *)c->dp_in[sl].p;
                int r = PNA_OK;
                size_t g0 = nx.e0;
                if (all_lent) {
                    // every entry lies in page-locked memory the library lent out (pna_gpu_host_alloc): no staging copy, the copy engine reads the host's
                    // buffers themselves -- runs of entries that are contiguous there and here (16-byte stride) travel as ONE copy
                    while (g0 < nx.e1 && r == PNA_OK) {
                        size_t g1 = g0 + 1;
                        while (g1 < nx.e1 && (const uint8_t *)src[g1] == (const uint8_t *)src[g0] + (off[g1] - off[g0])) g1++;
                        const uint64_t bytes = (g1 < nx.e1 ? off[g1] : off[g1 - 1] + src_len[g1 - 1]) - off[g0];
                        const uint64_t k = std::min<uint64_t>(bytes, (const uint8_t *)src[g1 - 1] + src_len[g1 - 1] - (const uint8_t *)src[g0]);
                        if (k && hipMemcpyAsync(db + off[g0], src[g0], k, hipMemcpyHostToDevice, cp_in) != hipSuccess) r = PNA_E_HIP;
                        g0 = g1;
                    }
                }
                while (g0 < nx.e1 && r == PNA_OK) {
                    size_t g1 = g0; uint64_t acc = 0;
                    while (g1 < nx.e1 && acc < (128ull << 20)) acc += src_len[g1++];
                    parallel_stage(hb, src, src_len, off.data(), g0, g1, threads);
                    const uint64_t b0 = off[g0], b1 = g1 < nx.e1 ? off[g1] : nx.in_bytes;
                    if (b1 > b0 && hipMemcpyAsync(db + b0, hb + b0, b1 - b0, hipMemcpyHostToDevice, cp_in) != hipSuccess) r = PNA_E_HIP;
                    g0 = g1;
                }
                if (r == PNA_OK && hipEventRecord(c->ev_in[sl], cp_in) != hipSuccess) r = PNA_E_HIP;
                trace("staged + H2D issued", k);
                std::lock_guard<std::mutex> lk(mu);
                if (r != PNA_OK) { stager_rc = r; staged = subs.size(); cv.notify_all(); return; }
                staged = k + 1; cv.notify_all();
            }
        } catch (...) { std::lock_guard<std::mutex> lk(mu); stager_rc = PNA_E_NOMEM; staged = subs.size(); cv.notify_all(); }
    });
    uint8_t *hp_out_dev[2] = {nullptr, nullptr};                 // device views of the page-locked output slots (the copy kernel's destination)
    const uint32_t d2h_wgs = (uint32_t)c->tun.d2h_wgs;
    for (size_t k = 0; k < subs.size() && rc == PNA_OK; k++) {
        const Sub &sb = subs[k]; const int sl = (int)(k % NS), so = (int)(k & 1);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return staged > k; });
            if (stager_rc != PNA_OK) { rc = fail(c, stager_rc, "staging / H2D copy failed"); break; }
        }
        if (hipEventSynchronize(c->ev_in[sl]) != hipSuccess) { rc = fail(c, PNA_E_HIP, "H2D copy failed"); break; }
        trace("H2D done, kernels start", k);
        rc = run_subbatch(c, algo, (const uint8_t *)c->dp_in[sl].p, off.data(), len64.data(), sb.e0, sb.e1, (uint8_t *)c->dp_out[so].p,
                          sb.out_cap + 64, 0, eoff.data(), c->stream, true, &fj);
        { std::lock_guard<std::mutex> lk(mu); freed = k + 1; cv.notify_all(); }       // (run_subbatch has waited for its kernels: the input slot is free)
        trace("kernels done", k);
        if (rc == PNA_OK) {
            out_len[so] = eoff[sb.e1];
            bool ok = true;
            if (d2h_wgs && !hp_out_dev[so]) ok = hipHostGetDevicePointer((void **)&hp_out_dev[so], c->hp_out[so].p, 0) == hipSuccess;
            if (ok && d2h_wgs) { launch_link_copy((const uint8_t *)c->dp_out[so].p, hp_out_dev[so], out_len[so], d2h_wgs, c->cp_out); ok = hipGetLastError() == hipSuccess; }
            else if (ok) ok = hipMemcpyAsync(c->hp_out[so].p, c->dp_out[so].p, out_len[so], hipMemcpyDeviceToHost, c->cp_out) == hipSuccess;
            if (!ok || hipEventRecord(c->ev_out[so], c->cp_out) != hipSuccess) rc = fail(c, PNA_E_HIP, "D2H copy failed");
        }
        if (rc == PNA_OK && k > 0) {                             // archive bytes of the previous sub-batch -> sink
            if (hipEventSynchronize(c->ev_out[so ^ 1]) != hipSuccess) rc = fail(c, PNA_E_HIP, "D2H copy failed");
            else if (out_len[so ^ 1] && sink(user, c->hp_out[so ^ 1].p, out_len[so ^ 1]) != 0) rc = fail(c, PNA_E_SINK, "sink failed");
            out_total += out_len[so ^ 1];
        }
    }
    { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); }
    stager.join();
    trace("loop done", 0);
    if (rc != PNA_OK) { (void)hipDeviceSynchronize(); return rc; }
    if (!subs.empty()) {
        const int so = (int)((subs.size() - 1) & 1);
        HIPCHK(c, hipEventSynchronize(c->ev_out[so]));
        if (out_len[so] && sink(user, c->hp_out[so].p, out_len[so]) != 0) return fail(c, PNA_E_SINK, "sink failed");
        out_total += out_len[so];
    }
    if (!tail.empty() && sink(user, tail.data(), tail.size()) != 0) return fail(c, PNA_E_SINK, "sink failed");
    out_total += tail.size();
    trace("all bytes handed to the sink", 0);
    c->timing.in_bytes = in_total; c->timing.out_bytes = out_total;
    return PNA_OK;
}


// ---------------------------------------------------------------------------------------------------------
// Read side driver: `pna extract` / `pna verify` for non-solid archives (cli/src/command/extract.rs:594-640, verify.rs:140-188;
// Archive::read_header + next_raw_item, lib/src/archive/read.rs:22-66; TryFrom<RawEntry>, lib/src/entry.rs:757-885; read_chunk with its
// mandatory CRC check, lib/src/io.rs:117-149; decrypt_reader / decompress_reader, lib/src/entry/read.rs:59-104,171-190).
// The chunk walk and the small chunks' CRCs are host work; the data chunks' CRC-32 (k_frame in verify mode), the gather of every
// entry's data pieces into one stream (k_gather), AES-CTR decryption and the zstd / deflate decoding run on the device.
// The name an entry is handed out under: EntryHeader::path() (lib/src/entry/header.rs:91-94,143-147) -- the FHED bytes must be UTF-8
// (InvalidData otherwise), and what callers see is the SANITISED form (EntryName::sanitize: no root, no "." / "..", so a crafted
// "../../etc/x" or "/abs" cannot leave the extraction directory).  The callback takes a C string, so an embedded NUL is rejected too.
static bool utf8_ok(const uint8_t *p, size_t n) {
    for (size_t i = 0; i < n;) {
        const uint8_t b = p[i];
        size_t k; uint32_t cp;
        if (b < 0x80) { i++; continue; }
        else if ((b & 0xE0) == 0xC0) { k = 1; cp = b & 0x1F; }
        else if ((b & 0xF0) == 0xE0) { k = 2; cp = b & 0x0F; }
        else if ((b & 0xF8) == 0xF0) { k = 3; cp = b & 0x07; }
        else return false;
        for (size_t j = 1; j <= k; j++) { if (i + j >= n || (p[i + j] & 0xC0) != 0x80) return false; cp = (cp << 6) | (p[i + j] & 0x3F); }
        if ((k == 1 && cp < 0x80) || (k == 2 && cp < 0x800) || (k == 3 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += k + 1;
    }
    return true;
}
static int entry_path(pna_gpu_ctx *c, const std::string &raw, std::string &out) {
    if (memchr(raw.data(), 0, raw.size())) return fail(c, PNA_E_INVAL, "entry name contains a NUL byte");
    if (!utf8_ok((const uint8_t *)raw.data(), raw.size())) return fail(c, PNA_E_INVAL, "entry name is not valid UTF-8");
    out = pna::pna_sanitize_name(raw.data(), raw.size());
    return PNA_OK;
}
namespace {
struct XPiece { uint64_t off; uint32_t len; };
struct XEntry {
    std::string name; int kind = 0, compression = 0, encryption = 0, cipher_mode = 0;
    bool has_size = false; uint64_t raw_size = 0; std::string phsf;
    std::vector<XPiece> pieces; uint64_t stream_len = 0;
    uint64_t pk_off = 0, pay_len = 0, raw_off = 0;            // payload (prefix stripped) in the packed buffer; decoded bytes in the raw buffer
    std::vector<uint8_t> fhed;                                 // FHED body: the GCM stream key is bound to it
    uint32_t gcm_seg = 0;                                      // GCM STREAM: segment size of the stream header
    size_t d0 = 0, d1 = 0;                                     // its FDAT chunks in the descriptor list
    uint64_t lo = 0, hi = 0;                                   // archive bytes [lo, hi) that hold its data chunks
};
typedef std::vector<std::pair<std::string, std::vector<uint8_t>>> XKeys;
static uint32_t max_chunk_len(const std::vector<FrameDesc> &v) {       // the longest data chunk of a list (0: none below 16 380 bytes, the wave-per-chunk CRC kernel's limit)
    uint32_t m = 1;
    for (const FrameDesc &d : v) { if (d.payload_len > 16380u) return 0u; m = std::max(m, d.payload_len); }
    return m;
}
struct XSolid {                                                // SHED [PHSF] SDAT* SEND -- lib/src/entry.rs:465-484,567-583
    int compression = 0, encryption = 0, cipher_mode = 0; std::string phsf;
    std::vector<XPiece> pieces; uint64_t stream_len = 0;
    size_t order = 0;                                          // number of normal entries in front of it
    uint64_t pk_off = 0, pay_len = 0;
    std::vector<uint8_t> shed;                                 // SHED body: the GCM stream key is bound to it (entry_context, lib/src/cipher/aead.rs:167-190)
    uint32_t gcm_seg = 0;
    size_t s0 = 0, s1 = 0;                                     // its SDAT chunks in the descriptor list
    uint64_t lo = 0, hi = 0;
};
// an encrypted data stream of the archive, a normal entry's or a solid entry's: what the cipher stage needs of either
struct XCipherStream {
    const std::string *phsf; int mode; const std::vector<XPiece> *pieces; uint64_t stream_len, pk_off; uint64_t *pay_len; uint32_t gcm_seg;
    const char *htype; const std::vector<uint8_t> *hdr; uint8_t iv[16];
};
uint32_t rd_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int b64_val(char ch) {
    if (ch >= 'A' && ch <= 'Z') return ch - 'A'; if (ch >= 'a' && ch <= 'z') return ch - 'a' + 26;
    if (ch >= '0' && ch <= '9') return ch - '0' + 52; if (ch == '+') return 62; if (ch == '/') return 63; return -1;
}
bool b64_decode_nopad(const std::string &s, std::vector<uint8_t> &out) {
    uint32_t acc = 0; int bits = 0;
    for (char ch : s) { const int v = b64_val(ch); if (v < 0) return false; acc = (acc << 6) | (uint32_t)v; bits += 6; if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); } }
    return true;
}
}

// what a window leaves behind for later: `issue` starts the D2H copy of its decoded entries (called by the NEXT window once its own bytes
// are on the device, so the copy runs next to that window's kernels), `deliver` waits for it and hands the entries out
struct XDeferred { std::function<int()> issue, deliver; bool issued = false; explicit operator bool() const { return (bool)deliver; } };
static int extract_window(pna_gpu_ctx *c, const uint8_t *a, size_t archive_len, const void *password, size_t password_len, pna_entry_fn cb, void *user,
                          std::vector<XEntry> &ents, std::vector<FrameDesc> &dchunks, std::vector<FrameDesc> &schunks, std::vector<XSolid> &solids,
                          XKeys &keys, size_t &index, int slot, XDeferred *later, XDeferred *prev);

extern "C" int pna_gpu_extract_archive_host(pna_gpu_ctx *c, const void *archive, size_t archive_len, const void *password, size_t password_len,
                                            pna_entry_fn cb, void *user) {
    if (!c || !archive || !cb || (!password && password_len)) return fail(c, PNA_E_INVAL, "null argument");
    const uint8_t *a = (const uint8_t *)archive;
    static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x41, 0x0D, 0x0A, 0x1A, 0x0A};
    if (archive_len < 8 + 20 + 12 || memcmp(a, sig, 8) != 0) return fail(c, PNA_E_INVAL, "not a PNA archive");
    // ---- 1. chunk walk (host): structure, small-chunk CRCs, data-chunk descriptors
    std::vector<XEntry> ents; std::vector<FrameDesc> dchunks, schunks;
    std::vector<XSolid> solids; XSolid scur; bool in_solid = false;
    XEntry cur; bool in_entry = false, seen_ahed = false, ended = false;
    size_t pos = 8;
    while (pos < archive_len) {
        if (archive_len - pos < 12) return fail(c, PNA_E_INVAL, "truncated chunk header");
        const uint32_t len = rd_be32(a + pos); const uint8_t *ty = a + pos + 4, *data = a + pos + 8;
        if (archive_len - pos - 12 < len) return fail(c, PNA_E_INVAL, "truncated chunk body");
        const bool is_fdat = memcmp(ty, "FDAT", 4) == 0, is_sdat = memcmp(ty, "SDAT", 4) == 0;
        if (is_fdat || is_sdat) { if (len >= 0xFFFFFFF0u) return fail(c, PNA_E_INVAL, "data chunk too long"); (is_fdat ? dchunks : schunks).push_back(FrameDesc{pos, len, 0, 8, 0}); }
        else if (pna_crc32(pna_crc32(0, ty, 4), data, len) != rd_be32(data + len)) return fail(c, PNA_E_INVAL, "chunk CRC mismatch");
        if (!seen_ahed) {
            if (memcmp(ty, "AHED", 4) != 0 || len != 8 || data[0] != 0) return fail(c, PNA_E_INVAL, "first chunk must be AHED (major version 0)");
            seen_ahed = true;
        } else if (memcmp(ty, "AEND", 4) == 0) { ended = true; break; }
        else if (memcmp(ty, "ANXT", 4) == 0) return fail(c, PNA_E_UNSUPPORTED, "multipart archives are not read by this driver");
        else if (memcmp(ty, "SHED", 4) == 0) {
            if (in_entry || in_solid || len != 5 || data[0] != 0 || data[1] != 0) return fail(c, PNA_E_INVAL, "bad solid header");
            scur = XSolid(); in_solid = true; scur.order = ents.size(); scur.s0 = schunks.size(); scur.lo = pos;
            scur.compression = data[2]; scur.encryption = data[3]; scur.cipher_mode = data[4]; scur.shed.assign(data, data + len);
        } else if (in_solid) {
            if (memcmp(ty, "SDAT", 4) == 0) { scur.pieces.push_back(XPiece{pos + 8, len}); scur.stream_len += len; }
            else if (memcmp(ty, "PHSF", 4) == 0) scur.phsf.assign((const char *)data, len);
            else if (memcmp(ty, "SEND", 4) == 0) { scur.s1 = schunks.size(); scur.hi = pos + 12; solids.push_back(std::move(scur)); in_solid = false; }
            else if (!(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "unknown critical chunk in a solid entry");
        }
        else if (memcmp(ty, "FHED", 4) == 0) {
            if (in_entry || len < 6 || data[0] != 0 || data[1] != 0) return fail(c, PNA_E_INVAL, "bad entry header");
            cur = XEntry(); in_entry = true; cur.d0 = dchunks.size(); cur.lo = pos;
            cur.kind = data[2]; cur.compression = data[3]; cur.encryption = data[4]; cur.cipher_mode = data[5];
            cur.name.assign((const char *)data + 6, len - 6); cur.fhed.assign(data, data + len);
        } else if (!in_entry) { if (!(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "unknown critical chunk between entries"); }
        else if (is_fdat) { cur.pieces.push_back(XPiece{pos + 8, len}); cur.stream_len += len; }
        else if (memcmp(ty, "fSIZ", 4) == 0) { if (len > 8) return fail(c, PNA_E_UNSUPPORTED, "entry beyond 2^64 bytes"); cur.has_size = true; cur.raw_size = 0; for (uint32_t i = 0; i < len; i++) cur.raw_size = (cur.raw_size << 8) | data[i]; }
        else if (memcmp(ty, "PHSF", 4) == 0) cur.phsf.assign((const char *)data, len);
        else if (memcmp(ty, "FEND", 4) == 0) { cur.d1 = dchunks.size(); cur.hi = pos + 12; ents.push_back(std::move(cur)); in_entry = false; }
        else if (!(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "unknown critical chunk");      // chunk/types.rs: bit 5 of byte 0 clear = critical
        pos += 12 + (size_t)len;
    }
    if (!ended || in_entry || in_solid) return fail(c, PNA_E_INVAL, "archive not terminated by AEND");
    // ---- 2. windows: a run of entries whose archive bytes, packed payloads and decoded bytes stay within a few GiB each goes through the
    // device at a time (an archive of any size in host memory against a bounded footprint in HBM); a solid entry is a window of its own
    XKeys keys; size_t index = 0, si = 0, w0 = 0;
    const size_t n_all = ents.size();
    const uint64_t WIN = (uint64_t)c->tun.extract_win_mib << 20;  // 1 GiB of archive (and at most 3 GiB decoded) per window by default: small enough to pipeline, large enough for the kernels
    auto rebase_run = [&](size_t e0, size_t e1, std::vector<XEntry> &we, std::vector<FrameDesc> &wd, uint64_t base) {
        we.assign(std::make_move_iterator(ents.begin() + e0), std::make_move_iterator(ents.begin() + e1));
        wd.assign(dchunks.begin() + we.front().d0, dchunks.begin() + we.back().d1);
        for (auto &f : wd) f.arc_off -= base;
        for (auto &e : we) for (auto &p : e.pieces) p.off -= base;
    };
    // Windows are pipelined against each other: the decoded entries of window k travel to the host (their own stream, their own pair of
    // buffers) while window k + 1 is copied in and decoded; window k's entries are handed out once k + 1 has been launched, before k + 1's.
    XDeferred pending; int slot = 0;
    auto finish_pending = [&]() -> int {
        if (!pending) return PNA_OK;
        XDeferred f = std::move(pending); pending = XDeferred();
        if (!f.issued) { const int r = f.issue(); if (r) return r; }
        return f.deliver();
    };
    auto run_window = [&](const uint8_t *wa, size_t wlen, std::vector<XEntry> &we, std::vector<FrameDesc> &wd, std::vector<FrameDesc> &ws, std::vector<XSolid> &wso) -> int {
        XDeferred cur;
        int rc = extract_window(c, wa, wlen, password, password_len, cb, user, we, wd, ws, wso, keys, index, slot, &cur, pending ? &pending : nullptr);
        const int rc2 = finish_pending();
        if (rc == PNA_OK) rc = rc2;
        if (rc != PNA_OK) { (void)hipDeviceSynchronize(); return rc; }
        pending = std::move(cur); slot ^= 1;
        return PNA_OK;
    };
    while (w0 < n_all || si < solids.size()) {
        std::vector<XEntry> we; std::vector<FrameDesc> wd, ws; std::vector<XSolid> wso;
        if (si < solids.size() && solids[si].order <= w0) {
            XSolid so = std::move(solids[si]);
            const uint64_t base = so.lo;
            ws.assign(schunks.begin() + so.s0, schunks.begin() + so.s1);
            for (auto &f : ws) f.arc_off -= base;
            for (auto &p : so.pieces) p.off -= base;
            const uint64_t span = so.hi - base;
            so.order = 0; wso.push_back(std::move(so)); si++;
            int rc = run_window(a + base, (size_t)span, we, wd, ws, wso);
            if (rc) return rc;
            continue;
        }
        size_t w1 = w0; uint64_t raw = 0, pk = 0;
        const size_t stop = si < solids.size() ? std::min(n_all, solids[si].order) : n_all;
        while (w1 < stop) {
            const XEntry &e = ents[w1];
            const uint64_t r = e.has_size ? e.raw_size : 0;
            if (w1 > w0 && (e.hi - ents[w0].lo > WIN || raw + r > 3 * WIN || pk + e.stream_len > WIN)) break;
            raw += r; pk += e.stream_len; w1++;
        }
        const uint64_t base = ents[w0].lo, span = ents[w1 - 1].hi - base;
        rebase_run(w0, w1, we, wd, base);
        int rc = run_window(a + base, (size_t)span, we, wd, ws, wso);
        if (rc) return rc;
        w0 = w1;
    }
    return finish_pending();
}

// One window of the driver above: `a` / archive_len are the window's bytes, every offset in ents / dchunks / schunks / solids is relative to it.
static int extract_window(pna_gpu_ctx *c, const uint8_t *a, size_t archive_len, const void *password, size_t password_len, pna_entry_fn cb, void *user,
                          std::vector<XEntry> &ents, std::vector<FrameDesc> &dchunks, std::vector<FrameDesc> &schunks, std::vector<XSolid> &solids,
                          XKeys &keys, size_t &index, int slot, XDeferred *later, XDeferred *prev) {
    const size_t n = ents.size();
    // keys (one derivation per distinct PHSF string), layout of the packed payloads and of the decoded entries
    auto key_for = [&](const std::string &phsf, const uint8_t **out) -> int {
        for (auto &k : keys) if (k.first == phsf) { *out = k.second.data(); return PNA_OK; }
        // "$pbkdf2-sha256$i=<rounds>,l=<len>$<salt>" (derive_password_hash, lib/src/hash.rs:47-88); Argon2 strings need the Rust host
        if (phsf.rfind("$argon2", 0) == 0) {
            // "$argon2id$v=19$m=<KiB>,t=<passes>,p=<lanes>$<salt>" (argon2 0.5: Params::try_from(&PasswordHash), lib/src/hash.rs:56-70)
            int kind = -1; size_t p1 = 0;
            if (phsf.rfind("$argon2id$", 0) == 0) { kind = 2; p1 = 10; } else if (phsf.rfind("$argon2i$", 0) == 0) { kind = 1; p1 = 9; } else if (phsf.rfind("$argon2d$", 0) == 0) { kind = 0; p1 = 9; }
            if (kind < 0) return fail(c, PNA_E_INVAL, "malformed PHSF");
            if (phsf.compare(p1, 2, "v=") == 0) { const size_t q = phsf.find('$', p1); if (q == std::string::npos || strtoul(phsf.c_str() + p1 + 2, nullptr, 10) != 19) return fail(c, PNA_E_UNSUPPORTED, "argon2 version other than 0x13"); p1 = q + 1; }
            const size_t p2 = phsf.find('$', p1);
            if (p2 == std::string::npos) return fail(c, PNA_E_INVAL, "malformed PHSF");
            uint32_t m = 19456, t = 2, lanes = 1;                 // argon2 0.5 defaults
            const std::string prm = phsf.substr(p1, p2 - p1);
            for (size_t q = 0; q < prm.size();) {
                const size_t e2 = prm.find(',', q); const std::string kv = prm.substr(q, e2 == std::string::npos ? std::string::npos : e2 - q);
                if (kv.size() > 2 && kv[1] == '=') {
                    char *endp = nullptr; const unsigned long long v = strtoull(kv.c_str() + 2, &endp, 10);
                    if (!endp || *endp || endp == kv.c_str() + 2) return fail(c, PNA_E_INVAL, "malformed argon2 parameter in PHSF");
                    // the parameters come from an untrusted archive: refuse costs that only serve to stall / exhaust the host
                    if ((kv[0] == 'm' && v > (4ull << 20)) || (kv[0] == 't' && v > 64) || (kv[0] == 'p' && v > 256)) return fail(c, PNA_E_UNSUPPORTED, "argon2 cost beyond the accepted maximum (m <= 4 GiB, t <= 64, p <= 256)");
                    if (kv[0] == 'm') m = (uint32_t)v; else if (kv[0] == 't') t = (uint32_t)v; else if (kv[0] == 'p') lanes = (uint32_t)v;
                }
                if (e2 == std::string::npos) break; q = e2 + 1;
            }
            std::vector<uint8_t> salt;
            std::string sb = phsf.substr(p2 + 1); const size_t p3 = sb.find('$'); if (p3 != std::string::npos) sb.resize(p3);
            if (!b64_decode_nopad(sb, salt)) return fail(c, PNA_E_INVAL, "malformed PHSF");
            std::vector<uint8_t> key(32);
            int rc = pna_kdf_argon2(kind, password, password_len, salt.data(), salt.size(), t, m, lanes, key.data(), 32);
            if (rc) return fail(c, rc, "key derivation failed (argon2 parameters)");
            keys.emplace_back(phsf, std::move(key)); *out = keys.back().second.data();
            return PNA_OK;
        }
        if (phsf.rfind("$pbkdf2-sha256$", 0) != 0) return fail(c, PNA_E_UNSUPPORTED, "password hash other than argon2 / pbkdf2-sha256");
        const size_t p1 = 15, p2 = phsf.find('$', p1);
        if (p2 == std::string::npos) return fail(c, PNA_E_INVAL, "malformed PHSF");
        uint32_t rounds = 600000;
        const std::string prm = phsf.substr(p1, p2 - p1);
        const size_t ip = prm.find("i=");
        if (ip != std::string::npos) {
            char *endp = nullptr; const unsigned long long v = strtoull(prm.c_str() + ip + 2, &endp, 10);
            if (!endp || (*endp && *endp != ',') || v == 0) return fail(c, PNA_E_INVAL, "malformed pbkdf2 round count in PHSF");
            if (v > 10000000ull) return fail(c, PNA_E_UNSUPPORTED, "pbkdf2 round count beyond the accepted maximum (10 000 000)");
            rounds = (uint32_t)v;
        }
        std::vector<uint8_t> salt;
        std::string sb = phsf.substr(p2 + 1); const size_t p3 = sb.find('$'); if (p3 != std::string::npos) sb.resize(p3);
        if (!b64_decode_nopad(sb, salt) || rounds == 0) return fail(c, PNA_E_INVAL, "malformed PHSF");
        std::vector<uint8_t> key(32);
        int rc = pna_kdf_pbkdf2_sha256(password, password_len, salt.data(), salt.size(), rounds, key.data(), 32, nullptr, 0);
        if (rc) return fail(c, rc, "key derivation failed");
        keys.emplace_back(phsf, std::move(key)); *out = keys.back().second.data();
        return PNA_OK;
    };
    uint64_t pk_total = 0, raw_total = 0;
    std::vector<PlaceDescH> places; std::vector<XCipherStream> enc_list, gcm_list; std::vector<size_t> nosize_idx;
    std::vector<std::vector<uint8_t>> nosize_data;
    // A data stream (the concatenated FDAT / SDAT bodies) is laid into the packed buffer at pk_off with its cipher prefix stripped: CTR / CBC lose the
    // IV, a GCM STREAM its header and the segments' tags (only the ciphertext is gathered).  Sets pay_len (and gcm_seg) and registers the stream
    // with the cipher stage.
    auto plan_stream = [&](const std::vector<XPiece> &pieces, uint64_t stream_len, int encryption, int cipher_mode, const std::string &phsf,
                           const char *htype, const std::vector<uint8_t> &hdr, uint64_t pk_off, uint64_t &pay_len, uint32_t &gcm_seg) -> int {
        auto stream_read = [&](uint64_t lo, uint64_t n2, uint8_t *out) {        // (the prefix may span data pieces: prepend_data_prefix makes it a piece of its own)
            uint64_t at2 = 0, got = 0;
            for (const XPiece &p : pieces) { for (uint32_t k = 0; k < p.len && got < n2; k++) if (at2 + k >= lo) out[got++] = a[p.off + k]; at2 += p.len; if (got >= n2) break; }
        };
        auto stream_place = [&](uint64_t lo, uint64_t hi, uint64_t dst) {
            uint64_t at2 = 0;
            for (const XPiece &p : pieces) {
                const uint64_t s0 = std::max<uint64_t>(lo, at2), s1 = std::min<uint64_t>(hi, at2 + p.len);
                for (uint64_t k = s0; k < s1; k += (1u << 20)) places.push_back(PlaceDescH{p.off + (k - at2), dst + (k - lo), (uint32_t)std::min<uint64_t>(1u << 20, s1 - k), 0});
                at2 += p.len;
            }
        };
        if (encryption == PNA_ENC_NONE) { pay_len = stream_len; stream_place(0, stream_len, pk_off); return PNA_OK; }
        if (encryption != PNA_ENC_AES) return fail(c, PNA_E_UNSUPPORTED, "only AES entries are decrypted by this driver");
        if (!password) return fail(c, PNA_E_INVAL, "encrypted entry and no password");
        if (phsf.empty()) return fail(c, PNA_E_INVAL, "`PHSF` chunk not found");
        XCipherStream cs{&phsf, cipher_mode, &pieces, stream_len, pk_off, &pay_len, 0, htype, &hdr, {0}};
        if (cipher_mode == PNA_MODE_CTR || cipher_mode == PNA_MODE_CBC) {
            if (stream_len < 16) return fail(c, PNA_E_INVAL, "data stream shorter than the IV");
            stream_read(0, 16, cs.iv);
            pay_len = stream_len - 16;
            stream_place(16, stream_len, pk_off);
            enc_list.push_back(cs);
        } else if (cipher_mode == PNA_MODE_GCM) {
            // stream header, then segments of (segment size + 16-byte tag), the last one shorter: only the ciphertext is gathered
            if (stream_len < 75 + 16) return fail(c, PNA_E_INVAL, "datastream shorter than the stream header");
            uint8_t hd[75]; stream_read(0, 75, hd);
            gcm_seg = rd_be32(hd + 39);
            if (gcm_seg == 0 || gcm_seg > (64u << 20)) return fail(c, PNA_E_INVAL, "GCM segment size out of range");
            cs.gcm_seg = gcm_seg;
            uint64_t rest = stream_len - 75, at2 = 75, outp = pk_off;
            while (rest) {
                const uint64_t segl = std::min<uint64_t>(rest, (uint64_t)gcm_seg + 16);
                if (segl < 16) return fail(c, PNA_E_INVAL, "GCM segment shorter than a tag");
                stream_place(at2, at2 + segl - 16, outp);
                outp += segl - 16; at2 += segl; rest -= segl;
            }
            pay_len = outp - pk_off;
            gcm_list.push_back(cs);
        } else return fail(c, PNA_E_UNSUPPORTED, "unknown cipher mode");
        return PNA_OK;
    };
    for (size_t i = 0; i < n; i++) {
        XEntry &e = ents[i];
        if (e.compression != PNA_ALGO_STORE && e.compression != PNA_ALGO_ZSTD && e.compression != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "compression method not decoded on the device (xz)");
        e.pk_off = pk_total;
        { const int r = plan_stream(e.pieces, e.stream_len, e.encryption, e.cipher_mode, e.phsf, "FHED", e.fhed, e.pk_off, e.pay_len, e.gcm_seg); if (r) return r; }
        pk_total = (pk_total + e.pay_len + 15) & ~(uint64_t)15;
        if (e.compression != PNA_ALGO_STORE) {
            // fSIZ is optional (older writers omit it): the payload is then decoded like a solid stream, its size found by the decoder
            if (!e.has_size) nosize_idx.push_back(i);
            else {
                // fSIZ comes from the archive: a size no payload of this length can decode to (deflate tops out at 1032 : 1, zstd at a few
                // thousand : 1 through RLE blocks) is damage, not a reason to ask the device for exabytes
                if (e.raw_size > (1ull << 40) || e.raw_size / 65536 > e.pay_len + 1) return fail(c, PNA_E_INVAL, "fSIZ is out of proportion to the entry's data");
                e.raw_off = raw_total; raw_total = (raw_total + e.raw_size + 15) & ~(uint64_t)15;
                if (raw_total > (1ull << 42)) return fail(c, PNA_E_NOMEM, "archive decodes to more than this driver takes in one call");
            }
        }
    }
    for (XSolid &so : solids) {
        if (so.compression != PNA_ALGO_STORE && so.compression != PNA_ALGO_ZSTD && so.compression != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "solid stream: compression method not decoded on the device (xz)");
        so.pk_off = pk_total;
        { const int r = plan_stream(so.pieces, so.stream_len, so.encryption, so.cipher_mode, so.phsf, "SHED", so.shed, so.pk_off, so.pay_len, so.gcm_seg); if (r) return r; }
        pk_total = (pk_total + so.pay_len + 15) & ~(uint64_t)15;
    }
    // ---- 3. device: upload, data-chunk CRCs, gather, decrypt, decode
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    static const bool xtrace = getenv("PNA_EXTRACT_TRACE") != nullptr;   // per-window phase times on stderr
    const auto xt0 = std::chrono::steady_clock::now();
    auto xms = [&](std::chrono::steady_clock::time_point a2) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a2).count(); };
    int rc = ensure_crc(c); if (rc) return rc;
    if (c->x_arc.ensure(archive_len + 64) || c->x_pk.ensure(pk_total + 8192) || c->x_raw[slot].ensure(raw_total + 64) || c->x_flag.ensure(64) ||
        c->x_desc.ensure(dchunks.size() * sizeof(FrameDesc) + 16) || c->x_place.ensure(places.size() * sizeof(PlaceDescH) + 16)) return fail(c, PNA_E_NOMEM, "extract workspace");
    HIPCHK(c, hipMemcpyAsync(c->x_arc.p, a, archive_len, hipMemcpyHostToDevice, st));
    const uint32_t flag0[2] = {0u, 0xFFFFFFFFu};
    HIPCHK(c, hipMemcpyAsync(c->x_flag.p, flag0, 8, hipMemcpyHostToDevice, st));
    if (!dchunks.empty()) {
        HIPCHK(c, hipMemcpyAsync(c->x_desc.p, dchunks.data(), dchunks.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        launch_frame_verify((const FrameDesc *)c->x_desc.p, (uint32_t)dchunks.size(), (const CrcTabs *)c->crc_tabs.p, (const uint8_t *)c->x_arc.p,
                            (uint64_t)c->x_arc.cap & ~(uint64_t)15, "FDAT", (uint32_t *)c->x_flag.p, st, max_chunk_len(dchunks));
    }
    if (!schunks.empty()) {
        if (c->solid_desc.ensure(schunks.size() * sizeof(FrameDesc) + 16)) return fail(c, PNA_E_NOMEM, "extract workspace");
        HIPCHK(c, hipMemcpyAsync(c->solid_desc.p, schunks.data(), schunks.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        launch_frame_verify((const FrameDesc *)c->solid_desc.p, (uint32_t)schunks.size(), (const CrcTabs *)c->crc_tabs.p, (const uint8_t *)c->x_arc.p,
                            (uint64_t)c->x_arc.cap & ~(uint64_t)15, "SDAT", (uint32_t *)c->x_flag.p, st, max_chunk_len(schunks));
    }
    if (!places.empty()) {
        HIPCHK(c, hipMemcpyAsync(c->x_place.p, places.data(), places.size() * sizeof(PlaceDescH), hipMemcpyHostToDevice, st));
        launch_gather(c->x_place.p, (uint32_t)places.size(), (const uint8_t *)c->x_arc.p, (uint8_t *)c->x_pk.p, st);
    }
    uint32_t flag[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(flag, c->x_flag.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    const double x_in = xms(xt0);
    // This window's bytes are on the device: now the previous window's decoded entries start their way back, next to this window's
    // decryption and decoding.  (Issued earlier, the two copies share the link -- H2D of 1 GiB next to D2H of 2.5 GiB took 67 ms, as long
    // as one after the other -- and the kernels would again run with the link idle.)
    if (prev && *prev && !prev->issued) { prev->issued = true; const int r = prev->issue(); if (r) return r; }
    if (flag[0]) { c->err = "data chunk CRC mismatch (" + std::to_string(flag[0]) + " FDAT / SDAT chunks)"; return PNA_E_INVAL; }
    if (!enc_list.empty()) {
        // streams sharing a PHSF string and a mode share the key: one cipher call per group
        std::vector<bool> done(enc_list.size(), false);
        for (size_t j = 0; j < enc_list.size(); j++) {
            if (done[j]) continue;
            const XCipherStream &e0 = enc_list[j];
            const uint8_t *key = nullptr;
            rc = key_for(*e0.phsf, &key); if (rc) return rc;
            std::vector<uint64_t> off, len; std::vector<uint8_t> iv2; std::vector<size_t> who;
            for (size_t k = j; k < enc_list.size(); k++)
                if (!done[k] && *enc_list[k].phsf == *e0.phsf && enc_list[k].mode == e0.mode) {
                    done[k] = true; off.push_back(enc_list[k].pk_off); len.push_back(*enc_list[k].pay_len); who.push_back(k);
                    iv2.insert(iv2.end(), enc_list[k].iv, enc_list[k].iv + 16);
                }
            if (e0.mode == PNA_MODE_CTR) {
                pna_gpu_cipher ci{}; ci.encryption = PNA_ENC_AES; ci.cipher_mode = PNA_MODE_CTR; memcpy(ci.key, key, 32); ci.phsf = ""; ci.ivs = iv2.data();
                rc = pna_gpu_cipher_apply_device(c, &ci, 1, off.size(), c->x_pk.p, off.data(), len.data(), st);
                if (rc) return rc;
            } else {                                              // CBC: DecryptCbcAes256Reader, lib/src/entry/read.rs:77-82
                rc = ensure_aes_dec(c); if (rc) return rc;
                // (a block's plaintext needs its own and the previous ciphertext block only: a long stream -- a solid one -- is cut into units of 16 MiB whose
                // IV is the ciphertext block in front; the padding is read at the end of the stream's last unit)
                constexpr uint64_t CBC_UNIT = 16u << 20;
                std::vector<CipherUnit> units; std::vector<uint8_t> uiv; std::vector<size_t> last_unit(off.size());
                for (size_t q = 0; q < off.size(); q++) {
                    if (len[q] == 0 || (len[q] & 15)) return fail(c, PNA_E_INVAL, "CBC: bad length or padding (wrong password or damaged data)");
                    for (uint64_t o = 0; o < len[q]; o += CBC_UNIT) {
                        units.push_back(CipherUnit{off[q] + o, 0, (uint32_t)std::min<uint64_t>(CBC_UNIT, len[q] - o), (uint32_t)units.size()});
                        last_unit[q] = units.size() - 1;
                    }
                }
                if (c->ci_units.ensure(units.size() * sizeof(CipherUnit) + 16) || c->ci_ivs.ensure(units.size() * 16 + 16) || c->x_plen.ensure(units.size() * 4 + 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
                // the units' IVs: the stream's own for its first unit, else the 16 ciphertext bytes in front of the unit (copied on the device BEFORE the
                // kernel overwrites them: the decryption is in place)
                { size_t u = 0;
                  for (size_t q = 0; q < off.size(); q++)
                      for (uint64_t o = 0; o < len[q]; o += CBC_UNIT, u++) {
                          if (o == 0) HIPCHK(c, hipMemcpyAsync((uint8_t *)c->ci_ivs.p + 16 * u, iv2.data() + 16 * q, 16, hipMemcpyHostToDevice, st));
                          else HIPCHK(c, hipMemcpyAsync((uint8_t *)c->ci_ivs.p + 16 * u, (const uint8_t *)c->x_pk.p + off[q] + o - 16, 16, hipMemcpyDeviceToDevice, st));
                      } }
                AesKey ek, dk; aes256_expand(key, ek); aes256_dec_key(ek, dk);
                std::vector<uint32_t> plen(units.size());
                HIPCHK(c, hipMemcpyAsync(c->ci_units.p, units.data(), units.size() * sizeof(CipherUnit), hipMemcpyHostToDevice, st));
                launch_aes_cbc_dec((const CipherUnit *)c->ci_units.p, (uint32_t)units.size(), (const uint8_t *)c->ci_ivs.p, (const AesDecTabs *)c->aes_dtabs.p,
                                   (uint8_t *)c->x_pk.p, dk, (uint32_t *)c->x_plen.p, st);
                HIPCHK(c, hipMemcpyAsync(plen.data(), c->x_plen.p, units.size() * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, hipStreamSynchronize(st));
                for (size_t q = 0; q < who.size(); q++) {
                    const uint32_t pl = plen[last_unit[q]];
                    if (pl == 0xFFFFFFFFu) return fail(c, PNA_E_INVAL, "CBC: bad length or padding (wrong password or damaged data)");
                    *enc_list[who[q]].pay_len = (len[q] - 1) / CBC_UNIT * CBC_UNIT + pl;
                }
            }
        }
    }
    if (!gcm_list.empty()) {
        // cipher mode 2 (decrypt_reader, (_, CipherMode::GCM): lib/src/entry/read.rs:105-140): key confirmation first -- a wrong password is
        // told apart from tampering --, then every segment's tag (k_gcm_tag in verify mode), then the CTR keystream with the stream keys
        rc = ensure_aes(c); if (rc) return rc;
        std::vector<GcmEntry> gents; std::vector<uint8_t> tags, giv; std::vector<AesKey> gkeys; std::vector<CipherUnit> units;
        for (const XCipherStream &e : gcm_list) {
            const uint8_t *km = nullptr;
            rc = key_for(*e.phsf, &km); if (rc) return rc;
            uint8_t hd[75]; { uint64_t got = 0; for (const XPiece &p : *e.pieces) { for (uint32_t k = 0; k < p.len && got < 75; k++) hd[got++] = a[p.off + k]; if (got >= 75) break; } }
            uint8_t kc[32]; hkdf_sha256_32(km, 32, nullptr, 0, "PNA-KC-v1", 9, kc);
            { uint8_t diff = 0; for (int b = 0; b < 32; b++) diff |= (uint8_t)(kc[b] ^ hd[43 + b]);      // constant time, like the reference's ct_eq
              if (diff) return fail(c, PNA_E_INVAL, "GCM STREAM: key confirmation failed (wrong password)"); }
            uint8_t info[88], ph[32], ks[32];
            memcpy(info, "PNA-STREAM-v1", 13);
            sha256_bytes(e.htype, 4, e.hdr->data(), e.hdr->size(), info + 13);      // (entry_context: the header chunk's type and body, FHED or SHED -- lib/src/cipher/aead.rs:167-190)
            sha256_bytes(e.phsf->data(), e.phsf->size(), nullptr, 0, ph); memcpy(info + 45, ph, 32);
            memcpy(info + 77, hd + 32, 7); memcpy(info + 84, hd + 39, 4);
            hkdf_sha256_32(km, 32, hd, 32, info, 88, ks);
            AesKey rk; aes256_expand(ks, rk);
            uint8_t zero[16] = {0}, hb[16]; aes256_block_host(rk, zero, hb);
            uint64_t rest = e.stream_len - 75, at2 = 75, outp = e.pk_off; uint32_t counter = 0;
            while (rest) {
                const uint64_t segl = std::min<uint64_t>(rest, (uint64_t)e.gcm_seg + 16), ctl = segl - 16;
                const bool fin = segl == rest;
                uint8_t j0[16], eb[16], tag[16];
                memcpy(j0, hd + 32, 7); j0[7] = (uint8_t)(counter >> 24); j0[8] = (uint8_t)(counter >> 16); j0[9] = (uint8_t)(counter >> 8); j0[10] = (uint8_t)counter; j0[11] = fin ? 1 : 0;
                j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
                aes256_block_host(rk, j0, eb);
                GcmEntry ge{outp, (uint32_t)ctl, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
                for (int w = 0; w < 4; w++) {
                    ge.h[w] = ((uint32_t)hb[4 * w] << 24) | ((uint32_t)hb[4 * w + 1] << 16) | ((uint32_t)hb[4 * w + 2] << 8) | hb[4 * w + 3];
                    ge.ej0[w] = ((uint32_t)eb[4 * w] << 24) | ((uint32_t)eb[4 * w + 1] << 16) | ((uint32_t)eb[4 * w + 2] << 8) | eb[4 * w + 3];
                }
                { uint64_t p2 = 0, got = 0; const uint64_t lo = at2 + ctl;      // the stored tag, wherever the chunk boundaries fall
                  for (const XPiece &p : *e.pieces) { for (uint32_t k = 0; k < p.len && got < 16; k++) if (p2 + k >= lo) tag[got++] = a[p.off + k]; p2 += p.len; if (got >= 16) break; } }
                const uint32_t idx = (uint32_t)gents.size();
                gents.push_back(ge); tags.insert(tags.end(), tag, tag + 16); gkeys.push_back(rk);
                j0[15] = 2; giv.insert(giv.end(), j0, j0 + 16);
                for (uint64_t o = 0; o < ctl; o += CTR_UNIT) units.push_back(CipherUnit{outp + o, o, (uint32_t)std::min<uint64_t>(CTR_UNIT, ctl - o), idx});
                outp += ctl; at2 += segl; rest -= segl; counter++;
                if (!fin && segl != (uint64_t)e.gcm_seg + 16) return fail(c, PNA_E_INVAL, "GCM STREAM: short non-final segment");
            }
        }
        if (c->ci_gcm.ensure(gents.size() * sizeof(GcmEntry) + 16) || c->x_tags.ensure(tags.size() + 16) || c->ci_keys.ensure(gkeys.size() * sizeof(AesKey) + 16) ||
            c->ci_ivs.ensure(giv.size() + 16) || c->ci_units.ensure(units.size() * sizeof(CipherUnit) + 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
        HIPCHK(c, hipMemcpyAsync(c->x_flag.p, flag0, 8, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ci_gcm.p, gents.data(), gents.size() * sizeof(GcmEntry), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->x_tags.p, tags.data(), tags.size(), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ci_keys.p, gkeys.data(), gkeys.size() * sizeof(AesKey), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ci_ivs.p, giv.data(), giv.size(), hipMemcpyHostToDevice, st));
        if (!units.empty()) HIPCHK(c, hipMemcpyAsync(c->ci_units.p, units.data(), units.size() * sizeof(CipherUnit), hipMemcpyHostToDevice, st));
        launch_gcm_verify((const GcmEntry *)c->ci_gcm.p, (uint32_t)gents.size(), (const uint8_t *)c->x_pk.p, (const uint8_t *)c->x_tags.p, (uint32_t *)c->x_flag.p, st);
        HIPCHK(c, hipMemcpyAsync(flag, c->x_flag.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));
        if (flag[0]) return fail(c, PNA_E_INVAL, "GCM STREAM: authentication failure (a segment tag does not match)");
        AesKey k0{};
        launch_aes_ctr((const CipherUnit *)c->ci_units.p, (uint32_t)units.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, (uint8_t *)c->x_pk.p, k0, (const AesKey *)c->ci_keys.p, st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));
    }
    for (int algo : {PNA_ALGO_ZSTD, PNA_ALGO_DEFLATE}) {
        std::vector<uint64_t> so, sl, dof, rl;
        for (const XEntry &e : ents) if (e.compression == algo && e.has_size) { so.push_back(e.pk_off); sl.push_back(e.pay_len); dof.push_back(e.raw_off); rl.push_back(e.raw_size); }
        if (so.empty()) continue;
        rc = pna_gpu_decompress_batch_device(c, algo, so.size(), c->x_pk.p, so.data(), sl.data(), c->x_raw[slot].p, dof.data(), rl.data(), st);
        if (rc) return rc;
    }
    for (size_t i : nosize_idx) {                                 // compatibility path, one decode call per entry
        XEntry &e = ents[i];
        uint32_t nfr = 1; uint64_t got = 0;
        if (e.compression == PNA_ALGO_ZSTD) { rc = pna_gpu_zstd_stream_frames_device(c, c->x_pk.p, e.pk_off, e.pay_len, &nfr, st); if (rc) return rc; }
        const uint64_t cap = nfr > 1 ? (uint64_t)nfr * SEG_SIZE : std::min<uint64_t>(1ull << 30, std::max<uint64_t>(64ull << 20, 64 * e.pay_len));
        if (c->solid_plain.ensure(cap + 8192)) return fail(c, PNA_E_NOMEM, "entry buffer");
        rc = e.compression == PNA_ALGO_ZSTD ? pna_gpu_zstd_decompress_open_device(c, c->x_pk.p, e.pk_off, e.pay_len, c->solid_plain.p, 0, cap, &got, st)
                                            : pna_gpu_inflate_open_device(c, c->x_pk.p, e.pk_off, e.pay_len, c->solid_plain.p, 0, cap, &got, st);
        if (rc) return rc;
        nosize_data.emplace_back((size_t)got);
        if (got) HIPCHK(c, hipMemcpy(nosize_data.back().data(), c->solid_plain.p, got, hipMemcpyDeviceToHost));
        e.raw_size = got; e.raw_off = nosize_data.size() - 1;      // index into nosize_data
    }
    // ---- solid entries: decode a stream of unknown size, walk the inner records
    struct Inner { std::string name; int kind; std::vector<XPiece> pieces; uint64_t len; };
    std::vector<std::vector<Inner>> inner(solids.size());
    std::vector<std::vector<uint8_t>> plain(solids.size());
    for (size_t si = 0; si < solids.size(); si++) {
        XSolid &so = solids[si];
        // (an encrypted stream has been decrypted in place by the cipher stage above, with the normal entries' streams)
        uint64_t plen = so.pay_len; const void *d_plain = (const uint8_t *)c->x_pk.p + so.pk_off;
        if (so.compression != PNA_ALGO_STORE) {
            uint32_t nfr = 1;
            if (so.compression == PNA_ALGO_ZSTD) { rc = pna_gpu_zstd_stream_frames_device(c, c->x_pk.p, so.pk_off, so.pay_len, &nfr, st); if (rc) return rc; }
            // this library's zstd solid streams: frames of 1 MiB; one frame / one zlib stream: a bounded guess of its size
            const uint64_t cap = nfr > 1 ? (uint64_t)nfr * SEG_SIZE : std::min<uint64_t>(1ull << 30, std::max<uint64_t>(64ull << 20, 64 * so.pay_len));
            if (c->solid_plain.ensure(cap + 8192)) return fail(c, PNA_E_NOMEM, "solid stream buffer");
            rc = so.compression == PNA_ALGO_ZSTD ? pna_gpu_zstd_decompress_open_device(c, c->x_pk.p, so.pk_off, so.pay_len, c->solid_plain.p, 0, cap, &plen, st)
                                                 : pna_gpu_inflate_open_device(c, c->x_pk.p, so.pk_off, so.pay_len, c->solid_plain.p, 0, cap, &plen, st);
            if (rc) return rc;
            d_plain = c->solid_plain.p;
        }
        plain[si].resize(plen);
        if (plen) HIPCHK(c, hipMemcpy(plain[si].data(), d_plain, plen, hipMemcpyDeviceToHost));
        // read_next_normal_entry_from_stream over the decoded stream (lib/src/entry.rs:401-424): small chunks checked here, the
        // inner FDAT CRCs on the device over the decoded stream where it stands
        const uint8_t *b = plain[si].data();
        std::vector<FrameDesc> ichunks; Inner ic; bool in_i = false;
        for (size_t q = 0; q < plen;) {
            if (plen - q < 12) return fail(c, PNA_E_INVAL, "solid stream: truncated chunk header");
            const uint32_t len = rd_be32(b + q); const uint8_t *ty = b + q + 4, *data = b + q + 8;
            if (plen - q - 12 < len) return fail(c, PNA_E_INVAL, "solid stream: truncated chunk body");
            const bool fd = memcmp(ty, "FDAT", 4) == 0;
            if (fd) { if (len >= 0xFFFFFFF0u) return fail(c, PNA_E_INVAL, "data chunk too long"); ichunks.push_back(FrameDesc{q, len, 0, 8, 0}); }
            else if (pna_crc32(pna_crc32(0, ty, 4), data, len) != rd_be32(data + len)) return fail(c, PNA_E_INVAL, "solid stream: chunk CRC mismatch");
            if (memcmp(ty, "FHED", 4) == 0) {
                if (in_i || len < 6 || data[0] != 0 || data[1] != 0) return fail(c, PNA_E_INVAL, "solid stream: bad entry header");
                if (data[3] != PNA_ALGO_STORE || data[4] != PNA_ENC_NONE) return fail(c, PNA_E_UNSUPPORTED, "solid stream: inner entry that is not stored");
                ic = Inner(); in_i = true; ic.kind = data[2]; ic.len = 0; ic.name.assign((const char *)data + 6, len - 6);
            } else if (!in_i) { if (!(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "solid stream: unknown critical chunk"); }
            else if (fd) { ic.pieces.push_back(XPiece{q + 8, len}); ic.len += len; }
            else if (memcmp(ty, "FEND", 4) == 0) { inner[si].push_back(std::move(ic)); in_i = false; }
            else if (memcmp(ty, "fSIZ", 4) != 0 && !(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "solid stream: unknown critical chunk");
            q += 12 + (size_t)len;
        }
        if (in_i) return fail(c, PNA_E_INVAL, "solid stream: dangling chunks");
        if (!ichunks.empty()) {
            if (c->solid_desc.ensure(ichunks.size() * sizeof(FrameDesc) + 16)) return fail(c, PNA_E_NOMEM, "extract workspace");
            HIPCHK(c, hipMemcpyAsync(c->x_flag.p, flag0, 8, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->solid_desc.p, ichunks.data(), ichunks.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
            const DevBuf &pb = so.compression != PNA_ALGO_STORE ? c->solid_plain : c->x_pk;
            std::vector<FrameDesc> adj;
            if (so.compression == PNA_ALGO_STORE) {                // descriptors are relative to the stream's start inside the packed buffer
                adj = ichunks; for (auto &f : adj) f.arc_off += so.pk_off;
                HIPCHK(c, hipMemcpyAsync(c->solid_desc.p, adj.data(), adj.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
            }
            launch_frame_verify((const FrameDesc *)c->solid_desc.p, (uint32_t)ichunks.size(), (const CrcTabs *)c->crc_tabs.p, (const uint8_t *)pb.p,
                                (uint64_t)pb.cap & ~(uint64_t)15, "FDAT", (uint32_t *)c->x_flag.p, st, max_chunk_len(ichunks));
            HIPCHK(c, hipMemcpyAsync(flag, c->x_flag.p, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(st));
            if (flag[0]) return fail(c, PNA_E_INVAL, "solid stream: inner FDAT CRC mismatch");
        }
    }
    // ---- 4. back to the host, entries in archive order.  Deferred form (no stored entries in the window): the D2H copy runs on its own
    // stream behind the window's kernels and the hand-out happens later (see the driver); everything it needs moves into `D`.
    bool any_store = false; for (const XEntry &e : ents) any_store |= e.compression == PNA_ALGO_STORE && e.pay_len;
    const bool defer = later != nullptr && !any_store;
    if (xtrace) { (void)hipStreamSynchronize(st); fprintf(stderr, "[pna extract window] %zu entries, %.0f MiB in -> %.0f MiB out: H2D + CRC + gather %.1f ms, decrypt + decode %.1f ms (slot %d, %s)\n", n, archive_len / 1048576.0, raw_total / 1048576.0, x_in, xms(xt0) - x_in, slot, defer ? "deferred hand-out" : "immediate"); }
    if (c->hp_out[slot].ensure(raw_total + 64) || (any_store && c->hp_in[0].ensure(pk_total + 64))) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    const uint64_t raw_bytes = raw_total;
    auto issue = [c, slot, raw_bytes]() -> int {                      // the window's kernels are complete on c->stream when this runs or are ordered before it by x_done
        if (raw_bytes && hipMemcpyAsync(c->hp_out[slot].p, c->x_raw[slot].p, raw_bytes, hipMemcpyDeviceToHost, c->x_cp) != hipSuccess) return fail(c, PNA_E_HIP, "D2H copy failed");
        return hipEventRecord(c->x_ev[slot], c->x_cp) == hipSuccess ? PNA_OK : fail(c, PNA_E_HIP, "D2H copy failed");
    };
    if (defer) {
        if (!c->x_cp) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->x_cp, hipStreamNonBlocking));
            for (auto &e : c->x_ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->x_done, hipEventDisableTiming));
        }
        HIPCHK(c, hipEventRecord(c->x_done, st));
        HIPCHK(c, hipStreamWaitEvent(c->x_cp, c->x_done, 0));
    } else {
        if (raw_total) HIPCHK(c, hipMemcpyAsync(c->hp_out[slot].p, c->x_raw[slot].p, raw_total, hipMemcpyDeviceToHost, st));
        if (any_store) HIPCHK(c, hipMemcpyAsync(c->hp_in[0].p, c->x_pk.p, pk_total, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    struct Deliver {
        std::vector<XEntry> ents; std::vector<size_t> solid_order; std::vector<std::vector<Inner>> inner; std::vector<std::vector<uint8_t>> plain, nosize_data;
        size_t index0 = 0;
    };
    auto D = std::make_shared<Deliver>();
    D->index0 = index;
    for (const XSolid &so : solids) D->solid_order.push_back(so.order);
    index += n; for (const auto &v : inner) index += v.size();
    D->ents = std::move(ents); D->inner = std::move(inner); D->plain = std::move(plain); D->nosize_data = std::move(nosize_data);
    const uint8_t *raw_host = (const uint8_t *)c->hp_out[slot].p, *pk_host = (const uint8_t *)c->hp_in[0].p;
    hipEvent_t wait_ev = defer ? c->x_ev[slot] : nullptr;
    auto deliver = [c, cb, user, D, raw_host, pk_host, wait_ev]() -> int {
        const auto dt0 = std::chrono::steady_clock::now();
        if (wait_ev && hipEventSynchronize(wait_ev) != hipSuccess) return fail(c, PNA_E_HIP, "D2H copy failed");
        const double dwait = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dt0).count();
        struct Tr { double w; std::chrono::steady_clock::time_point t; ~Tr() { if (getenv("PNA_EXTRACT_TRACE")) fprintf(stderr, "[pna extract hand-out] waited %.1f ms for the D2H copy, callbacks %.1f ms\n", w, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count()); } } tr{dwait, std::chrono::steady_clock::now()};
        size_t idx = D->index0, si = 0;
        const size_t n = D->ents.size();
        std::vector<uint8_t> joined;
        auto deliver_solids = [&](size_t upto) -> int {
            for (; si < D->solid_order.size() && D->solid_order[si] <= upto; si++)
                for (const Inner &ie : D->inner[si]) {
                    const uint8_t *d = D->plain[si].data();
                    if (ie.pieces.size() == 1) d += ie.pieces[0].off;
                    else { joined.clear(); for (const XPiece &p : ie.pieces) joined.insert(joined.end(), d + p.off, d + p.off + p.len); d = joined.data(); }
                    std::string path; { const int rp = entry_path(c, ie.name, path); if (rp) return rp; }
                    if (cb(user, idx++, path.c_str(), ie.kind, ie.len ? d : nullptr, (size_t)ie.len) != 0) return fail(c, PNA_E_SINK, "entry callback failed");
                }
            return PNA_OK;
        };
        for (size_t i = 0; i < n; i++) {
            int rc = deliver_solids(i); if (rc) return rc;
            const XEntry &e = D->ents[i];
            const uint8_t *d = e.compression == PNA_ALGO_STORE ? pk_host + e.pk_off
                             : (e.has_size ? raw_host + e.raw_off : D->nosize_data[(size_t)e.raw_off].data());
            const size_t l = e.compression == PNA_ALGO_STORE ? (size_t)e.pay_len : (size_t)e.raw_size;
            if (e.compression == PNA_ALGO_STORE && e.has_size && e.raw_size != e.pay_len) return fail(c, PNA_E_INVAL, "stored entry: fSIZ differs from the data length");
            std::string path; { const int rp = entry_path(c, e.name, path); if (rp) return rp; }
            if (cb(user, idx++, path.c_str(), e.kind, d, l) != 0) return fail(c, PNA_E_SINK, "entry callback failed");
        }
        return deliver_solids(n);
    };
    if (defer) { later->issue = issue; later->deliver = deliver; later->issued = false; return PNA_OK; }
    return deliver();
}

extern "C" int pna_gpu_compress_batch(pna_gpu_ctx *c, int algo, int level, size_t n, const void *const *src,
                                      const size_t *src_len, void *const *dst, const size_t *dst_cap, size_t *dst_len) {
    if (!c || (n && (!src || !src_len || !dst || !dst_cap || !dst_len))) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> off(n + 1), len(n), doff(n + 1), obase(n + 1);       // obase: running sum of the entries' bounds
    uint64_t pos = 0, bound = 0;
    for (size_t i = 0; i < n; i++) {
        if (dst_cap[i] < pna_gpu_bound(algo, src_len[i])) return fail(c, PNA_E_DSTSIZE, "dst_cap below pna_gpu_bound");
        off[i] = pos; len[i] = src_len[i]; pos = (pos + src_len[i] + 15) & ~(uint64_t)15;
        obase[i] = bound; bound += pna_gpu_bound(algo, src_len[i]);
    }
    off[n] = pos; obase[n] = bound;
    // Inputs are staged into page-locked memory by several threads and copied H2D, outputs copied D2H and scattered by several threads
    // (per-entry copies from pageable memory ran at ~1 GiB/s).  A large batch goes through in PIECES of >= 256 MiB (a round of the CUs:
    // the kernels' fixed latencies stay amortised): piece k + 1 is staged and copied while piece k is on the device, piece k - 1's results
    // travel back meanwhile -- 512 x 1 MiB: 27.5 -> see profiles/ (PNA_BATCH_PIECE_MIB; 0 = one piece).
    const uint64_t piece_bytes = c->tun.batch_piece_mib <= 0 ? ~0ull >> 1 : (uint64_t)c->tun.batch_piece_mib << 20;
    std::vector<size_t> pe{0};                                   // piece k = entries [pe[k], pe[k + 1])
    for (size_t i = 0; i < n;) {
        size_t j = i; uint64_t acc = 0;
        while (j < n && (j == i || acc + src_len[j] <= piece_bytes)) acc += src_len[j++];
        pe.push_back(j); i = j;
    }
    if (pe.size() > 2 && off[n] - off[pe[pe.size() - 2]] < piece_bytes / 2) pe.erase(pe.end() - 2);   // a short last piece joins its neighbour
    const size_t K = pe.size() - 1;
    uint64_t max_in = 0, max_out = 0;
    std::vector<uint64_t> pbase(K + 1, 0);                       // where piece k's output starts in stage_out (256-byte aligned)
    for (size_t k = 0; k < K; k++) {
        max_in = std::max(max_in, off[pe[k + 1]] - off[pe[k]]); max_out = std::max(max_out, obase[pe[k + 1]] - obase[pe[k]]);
        pbase[k + 1] = (pbase[k] + (obase[pe[k + 1]] - obase[pe[k]]) + 64 + 255) & ~(uint64_t)255;
    }
    if (c->stage_in.ensure(pos + 8192) || c->stage_out.ensure(pbase[K] + 64)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    for (int sl = 0; sl < (K > 1 ? 2 : 1); sl++) if (c->hp_in[sl].ensure(max_in + 64) || c->hp_out[sl].ensure(max_out + 64)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    if (K > 1 && !c->cp_in) {
        HIPCHK(c, hipStreamCreate(&c->cp_in)); HIPCHK(c, hipStreamCreate(&c->cp_out));
        for (auto &e : c->ev_in) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : c->ev_out) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned threads = std::min(8u, std::max(1u, hw / 2));
    hipStream_t s_in = K > 1 ? c->cp_in : c->stream, s_out = K > 1 ? c->cp_out : c->stream;
    auto stage = [&](size_t k) -> int {                          // entries of piece k -> pinned slot -> their place in stage_in
        const int sl = (int)(k & 1);
        const uint64_t b0 = off[pe[k]], nb = off[pe[k + 1]] - b0;
        std::vector<uint64_t> rel(pe[k + 1] - pe[k] + 1);
        for (size_t i = pe[k]; i <= pe[k + 1]; i++) rel[i - pe[k]] = off[i] - b0;
        parallel_stage((uint8_t *)c->hp_in[sl].p, src + pe[k], src_len + pe[k], rel.data(), 0, pe[k + 1] - pe[k], threads);
        if (nb) HIPCHK(c, hipMemcpyAsync((uint8_t *)c->stage_in.p + b0, c->hp_in[sl].p, nb, hipMemcpyHostToDevice, s_in));
        if (K > 1) HIPCHK(c, hipEventRecord(c->ev_in[sl], s_in));
        return PNA_OK;
    };
    std::vector<uint64_t> ptotal(K);
    auto scatter = [&](size_t k) -> int {                        // piece k's compressed entries: pinned slot -> the caller's buffers
        const int sl = (int)(k & 1);
        if (K > 1) HIPCHK(c, hipEventSynchronize(c->ev_out[sl])); else HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint8_t *hb = (const uint8_t *)c->hp_out[sl].p;
        const size_t e0 = pe[k], e1 = pe[k + 1];
        const unsigned T = ptotal[k] < (8u << 20) ? 1u : threads;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([=, &doff]() { for (size_t i = e0 + t; i < e1; i += T) if (dst_len[i]) memcpy(dst[i], hb + (doff[i] - doff[e0]), dst_len[i]); });
        for (auto &x : th) x.join();
        return PNA_OK;
    };
    pna_gpu_timing tsum{};
    int rc = K ? stage(0) : PNA_OK;
    if (rc) return rc;
    doff[0] = 0;
    for (size_t k = 0; k < K; k++) {
        const int sl = (int)(k & 1);
        if (k + 1 < K && (rc = stage(k + 1))) return rc;         // (its slot's previous copy, piece k - 1, was waited for by that piece's kernels)
        if (K > 1) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_in[sl], 0));
        const size_t e0 = pe[k], nk = pe[k + 1] - e0;
        std::vector<uint64_t> d(nk + 1);
        uint8_t *ob = (uint8_t *)c->stage_out.p + pbase[k];
        rc = pna_gpu_compress_batch_device(c, algo, level, nk, c->stage_in.p, off.data() + e0, len.data() + e0, ob, obase[pe[k + 1]] - obase[e0] + 64, d.data(), nullptr);
        if (rc) return rc;
        { const pna_gpu_timing &t = c->timing; tsum.ms_lz += t.ms_lz; tsum.ms_stats += t.ms_stats; tsum.ms_lit += t.ms_lit; tsum.ms_seq += t.ms_seq; tsum.ms_pack += t.ms_pack;
          tsum.in_bytes += t.in_bytes; tsum.out_bytes += t.out_bytes; tsum.n_segments += t.n_segments; tsum.n_blocks += t.n_blocks; tsum.ms_lz_match += t.ms_lz_match; tsum.lz_match_launches += t.lz_match_launches; }
        for (size_t i = 0; i < nk; i++) { doff[e0 + i + 1] = doff[e0] + d[i + 1]; dst_len[e0 + i] = (size_t)(d[i + 1] - d[i]); }
        ptotal[k] = d[nk];
        if (ptotal[k]) HIPCHK(c, hipMemcpyAsync(c->hp_out[sl].p, ob, ptotal[k], hipMemcpyDeviceToHost, s_out));   // (the kernels are done: compress_batch_device returns synchronised)
        if (K > 1) HIPCHK(c, hipEventRecord(c->ev_out[sl], s_out));
        if (k >= 1 && (rc = scatter(k - 1))) return rc;
    }
    if (K && (rc = scatter(K - 1))) return rc;
    c->timing = tsum;
    return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Read side, Compression::Deflate: one zlib stream per entry (flate2::read::ZlibDecoder, lib/src/entry/read.rs:178-179).
// k_inflate turns each stream into literals + (run, length, distance) records, k_zoff / k_zexec execute them, k_iadler_* check
// the Adler-32 trailer.
static int inflate_batch_device(pna_gpu_ctx *c, size_t n, const void *d_src, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                                const uint64_t *dst_off, const uint64_t *raw_len, hipStream_t st, bool open = false, uint64_t *raw_out = nullptr) {
    if (n > 0x3FFFFFFFull) return fail(c, PNA_E_INVAL, "batch too large for one decode call");
    std::vector<ZFrame> frs(n);
    std::vector<ZFrameX> fxs(n);
    std::vector<uint32_t> cbase(n + 1);
    uint64_t nseq_cap = 0, out_span = 0, pieces = 0, nblk = 0;
    // Streams of known size go lane-per-piece (k_vinflate): a stream of at most BLK_SIZE decoded bytes is one piece, a larger one is taken
    // to consist of ceil(raw_len / BLK_SIZE) sync-flush delimited pieces of BLK_SIZE bytes each (what this library's encoder writes) --
    // k_imark / k_vinflate / k_vfin check that and leave every stream that does not fit to the wave-per-stream kernel.  Streams of
    // unknown size (`open`) take the wave-per-stream kernel directly.
    struct VPieceH { uint32_t frame, j; };
    std::vector<VPieceH> vp;
    // pieces per stream: from the size when it is known; for streams of unknown size (solid streams, entries without fSIZ) from a count of
    // the sync-flush markers (one pass + one small read-back): markers + 1 pieces, all but the last holding BLK_SIZE bytes
    std::vector<uint64_t> npc(n);
    uint64_t tot_pieces = 0;
    bool lanes = !c->tun.inflate_serial;
    // workgroups per stream for the marker scans: one per 256 KiB of the batch's longest stream (n x G bounded)
    uint64_t max_src = 0;
    for (size_t i = 0; i < n; i++) max_src = std::max<uint64_t>(max_src, src_len[i]);
    const uint32_t scan_g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(max_src >> 18, 1024), (1ull << 24) / std::max<size_t>(n, 1)));
    if (lanes && open) {
        std::vector<uint32_t> cnt(n);
        if (c->z_pb.ensure(n * 4 + 8) || c->z_vp.ensure(n * 16 + 16)) return fail(c, PNA_E_NOMEM, "decoder workspace");
        HIPCHK(c, hipMemcpyAsync(c->z_vp.p, src_off, n * 8, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync((uint8_t *)c->z_vp.p + n * 8, src_len, n * 8, hipMemcpyHostToDevice, st));
        launch_icount((const uint8_t *)d_src, (const uint64_t *)c->z_vp.p, (const uint64_t *)((uint8_t *)c->z_vp.p + n * 8), (uint32_t)n, (uint32_t *)c->z_pb.p, scan_g, st);
        HIPCHK(c, hipMemcpyAsync(cnt.data(), c->z_pb.p, n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        for (size_t i = 0; i < n; i++) { npc[i] = (uint64_t)cnt[i] + 1; if ((npc[i] - 1) * BLK_SIZE > raw_len[i]) npc[i] = 1; }   // more pieces than the room allows: not this library's layout
    } else
        for (size_t i = 0; i < n; i++) npc[i] = std::max<uint64_t>(1, (raw_len[i] + BLK_SIZE - 1) / BLK_SIZE);
    for (size_t i = 0; i < n; i++) tot_pieces += npc[i];
    // a handful of pieces is served better by the wave-per-stream walk (a lane needs ~110 ms for a 128 KiB piece, however few there are)
    if (tot_pieces < 1024) lanes = false;
    for (size_t i = 0; i < n; i++) {
        // streams of 4 GiB and more: decoded by pieces (this library's layout: a sync flush behind every 128 KiB); the wave-per-stream walk counts in 32 bits
        if ((raw_len[i] > 0xFFFFFFFFull || src_len[i] > 0xFFFFFFFFull) && !lanes) return fail(c, PNA_E_UNSUPPORTED, "zlib streams of 4 GiB and more are decoded by sync-flush delimited pieces only");
        frs[i] = ZFrame{src_off[i], dst_off[i], src_len[i], raw_len[i], 0, open ? ZF_OPEN : 0u};   // open: raw_len is a capacity
        ZFrameX &x = fxs[i];
        const uint64_t P = lanes ? npc[i] : 1;
        const uint64_t pcap = std::min<uint64_t>(raw_len[i], lanes ? BLK_SIZE : raw_len[i]) / 3 + (raw_len[i] >> 16) / P + 16;   // matches are >= 3 bytes; + literal-run splits (serial walk)
        if (nblk + P > 0x7FFFFFFFull) return fail(c, PNA_E_INVAL, "batch too large for one decode call");
        x.blk_base = (uint32_t)nblk; x.blk_cap = (uint32_t)P; x.slot_base = 0; x.slot_cap = 0; x.nblk = 0;
        x.seq_base = nseq_cap; x.seq_cap = (uint32_t)std::min<uint64_t>(P * pcap, 0x7FFFFFFFu); x.pcap = (uint32_t)std::min<uint64_t>(pcap, 0x7FFFFFFFu); x.pad = 0;
        nseq_cap += P * pcap;
        if (lanes) for (uint64_t j = 0; j < P; j++) vp.push_back(VPieceH{(uint32_t)i, (uint32_t)j});
        nblk += P;
        out_span = std::max<uint64_t>(out_span, dst_off[i] + raw_len[i]);
        cbase[i] = (uint32_t)pieces;
        pieces += (raw_len[i] + 65535) >> 16;
        if (pieces > 0xFFFFFFF0ull) return fail(c, PNA_E_INVAL, "batch too large for one decode call");
    }
    cbase[n] = (uint32_t)pieces;
    if (c->z_frames.ensure(n * sizeof(ZFrame)) || c->z_fx.ensure(n * sizeof(ZFrameX)) || c->z_blocks.ensure(nblk * sizeof(ZBlock)) ||
        c->z_lit.ensure(out_span + 64) || c->z_seqs.ensure(nseq_cap * 8 + 64) || c->z_cbase.ensure((n + 1) * 4) || c->z_apart.ensure(pieces * 8 + 8) ||
        (lanes && (c->z_vp.ensure(vp.size() * 8 + 8) || c->z_pb.ensure((nblk + n) * 8 + 8) || c->z_mode.ensure(n * 4 + 8 + (size_t)n * scan_g * 4))))
        return fail(c, PNA_E_NOMEM, "decoder workspace");
    HIPCHK(c, hipMemcpyAsync(c->z_frames.p, frs.data(), n * sizeof(ZFrame), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->z_fx.p, fxs.data(), n * sizeof(ZFrameX), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->z_cbase.p, cbase.data(), (n + 1) * 4, hipMemcpyHostToDevice, st));
    if (lanes) HIPCHK(c, hipMemcpyAsync(c->z_vp.p, vp.data(), vp.size() * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    if (lanes) launch_vinflate((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, (uint32_t)n, c->z_vp.p, (uint32_t)vp.size(), (uint64_t *)c->z_pb.p, (uint32_t *)c->z_mode.p,
                               (uint32_t *)c->z_mode.p + n + 2, scan_g, (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p, (uint8_t *)c->z_lit.p, (uint64_t *)c->z_seqs.p, st);
    launch_inflate((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, (uint32_t)n, (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p, (uint8_t *)c->z_lit.p,
                   (uint64_t *)c->z_seqs.p, lanes ? (const uint32_t *)c->z_mode.p : nullptr, st);
    HIPCHK(c, hipEventRecord(c->ev[2], st));
    if (lanes) launch_zexec_groups((ZFrame *)c->z_frames.p, (const ZFrameX *)c->z_fx.p, (uint32_t)n, (ZBlock *)c->z_blocks.p, c->z_vp.p, (uint32_t)vp.size(), (const uint8_t *)d_src,
                                   (const uint8_t *)c->z_lit.p, (const uint64_t *)c->z_seqs.p, (uint8_t *)d_dst, st);   // execution groups side by side (k_vfin)
    else launch_zexec((ZFrame *)c->z_frames.p, (const ZFrameX *)c->z_fx.p, (uint32_t)n, (ZBlock *)c->z_blocks.p, (const uint8_t *)d_src,
                      (const uint8_t *)c->z_lit.p, (const uint64_t *)c->z_seqs.p, (uint8_t *)d_dst, st);
    HIPCHK(c, hipEventRecord(c->ev[3], st));
    launch_iadler((ZFrame *)c->z_frames.p, (const ZFrameX *)c->z_fx.p, (const ZBlock *)c->z_blocks.p, (uint32_t)n, (const uint32_t *)c->z_cbase.p,
                  (uint32_t)pieces, (const uint8_t *)d_dst, c->z_apart.p, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, n * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
    if (lanes) HIPCHK(c, hipStreamSynchronize(st));          // (vp is read by the copy above until then)
    HIPCHK(c, hipStreamSynchronize(st));
    float ms = 0, ms_h = 0, ms_x = 0;
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); (void)hipEventElapsedTime(&ms_h, c->ev[0], c->ev[2]); (void)hipEventElapsedTime(&ms_x, c->ev[2], c->ev[3]);
    c->timing = pna_gpu_timing{}; c->timing.ms_lz = ms; c->timing.ms_stats = ms_h; c->timing.ms_lit = ms_x;   // total, Huffman walk, execution
    for (size_t i = 0; i < n; i++)
        if (frs[i].status) {
            char msg[160];
            snprintf(msg, sizeof msg, "entry %zu: %s (produced %u of %llu bytes)", i,
                     frs[i].status == 2 ? "unsupported stream" : (frs[i].status == 3 ? "size mismatch" : "corrupt stream"), frs[i].out_len, (unsigned long long)frs[i].dst_len);
            return fail(c, frs[i].status == 2 ? PNA_E_UNSUPPORTED : PNA_E_INVAL, msg);
        }
    if (open && raw_out) for (size_t i = 0; i < n; i++) raw_out[i] = frs[i].dst_len;
    return PNA_OK;
}

// A zlib stream whose decoded size is recorded nowhere (deflate entries without fSIZ, deflate solid streams): decoded into dst_cap
// bytes of room, the size found is reported (PNA_E_INVAL when it does not fit).
extern "C" int pna_gpu_inflate_open_device(pna_gpu_ctx *c, const void *d_src, uint64_t src_off, uint64_t src_len, void *d_dst, uint64_t dst_off,
                                           uint64_t dst_cap, uint64_t *raw_len, void *hip_stream) {
    if (!c || !d_src || !d_dst || !raw_len) return fail(c, PNA_E_INVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return inflate_batch_device(c, 1, d_src, &src_off, &src_len, d_dst, &dst_off, &dst_cap, st, true, raw_len);
}

// ---------------------------------------------------------------------------------------------------------
// Read side: decompress_reader (lib/src/entry/read.rs:171-190); entries already in device memory.
static int zstd_decode_device(pna_gpu_ctx *c, size_t n, const void *d_src, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                              const uint64_t *dst_off, const uint64_t *raw_len, bool open, uint64_t *raw_out, hipStream_t st);

extern "C" int pna_gpu_decompress_batch_device(pna_gpu_ctx *c, int algo, size_t n, const void *d_src, const uint64_t *src_off,
                                               const uint64_t *src_len, void *d_dst, const uint64_t *dst_off, const uint64_t *raw_len,
                                               void *hip_stream) {
    if (!c || (n && (!d_src || !src_off || !src_len || !d_dst || !dst_off || !raw_len))) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "only zstd and deflate streams are decoded on the device");
    if (!n) return PNA_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (algo == PNA_ALGO_DEFLATE) return inflate_batch_device(c, n, d_src, src_off, src_len, d_dst, dst_off, raw_len, st);
    return zstd_decode_device(c, n, d_src, src_off, src_len, d_dst, dst_off, raw_len, false, nullptr, st);
}

// A zstd stream whose decoded size is not recorded anywhere (the SDAT stream of a solid entry: SHED carries no size): step 1 counts
// its frames, the caller provides frames x 1 MiB (this library's segmentation; one frame of any size: `cap` bytes), step 2 decodes
// and reports the size found.
extern "C" int pna_gpu_zstd_stream_frames_device(pna_gpu_ctx *c, const void *d_src, uint64_t src_off, uint64_t src_len, uint32_t *n_frames, void *hip_stream) {
    if (!c || !d_src || !n_frames) return fail(c, PNA_E_INVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (c->z_ents.ensure(sizeof(ZEntry)) || c->z_work.ensure(64)) return fail(c, PNA_E_NOMEM, "decoder workspace");
    const ZEntry en{src_off, src_len, 0, 0, 0, 0, 1, 0};
    HIPCHK(c, hipMemcpyAsync(c->z_ents.p, &en, sizeof en, hipMemcpyHostToDevice, st));
    launch_zcount((const ZEntry *)c->z_ents.p, 1, (const uint8_t *)d_src, (uint32_t *)c->z_work.p, st);
    HIPCHK(c, hipMemcpyAsync(n_frames, c->z_work.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (*n_frames == 0 && src_len) return fail(c, PNA_E_INVAL, "not a sequence of zstd frames");
    return PNA_OK;
}
extern "C" int pna_gpu_zstd_decompress_open_device(pna_gpu_ctx *c, const void *d_src, uint64_t src_off, uint64_t src_len, void *d_dst, uint64_t dst_off,
                                                   uint64_t dst_cap, uint64_t *raw_len, void *hip_stream) {
    if (!c || !d_src || !d_dst || !raw_len) return fail(c, PNA_E_INVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return zstd_decode_device(c, 1, d_src, &src_off, &src_len, d_dst, &dst_off, &dst_cap, true, raw_len, st);
}

static int zstd_decode_device(pna_gpu_ctx *c, size_t n, const void *d_src, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                              const uint64_t *dst_off, const uint64_t *raw_len, bool open, uint64_t *raw_out, hipStream_t st) {
    std::vector<ZEntry> ents(n);
    uint64_t nfr = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t k = raw_len[i] ? (raw_len[i] + SEG_SIZE - 1) / SEG_SIZE : 1;
        if (nfr + k > 0x7FFFFFFFull) return fail(c, PNA_E_INVAL, "too many frames");
        ents[i] = ZEntry{src_off[i], src_len[i], dst_off[i], raw_len[i], (uint32_t)nfr, (uint32_t)k, open ? 1u : 0u, 0u};
        nfr += k;
    }
    // per-frame bounds of the lane-parallel pipeline (frames that exceed them fall back to the one-workgroup-per-frame kernel)
    std::vector<ZFrameX> fxs(nfr);
    uint64_t nblk_cap = 0, nslot = 0, nseq_cap = 0, out_span = 0;
    for (size_t i = 0; i < n; i++) {
        out_span = std::max<uint64_t>(out_span, dst_off[i] + raw_len[i]);
        for (uint32_t f = 0; f < ents[i].n_frames; f++) {
            const uint64_t done = (uint64_t)f * SEG_SIZE;
            const uint64_t dl = (f + 1 == ents[i].n_frames) ? (raw_len[i] > done ? raw_len[i] - done : 0) : SEG_SIZE;
            ZFrameX &x = fxs[ents[i].first_frame + f];
            x.blk_base = (uint32_t)nblk_cap; x.blk_cap = (uint32_t)std::min<uint64_t>((dl >> 12) + 4, 1u << 20);
            x.slot_base = (uint32_t)nslot; x.slot_cap = (uint32_t)std::min<uint64_t>((dl >> 17) + 2, 1u << 16);
            x.seq_base = nseq_cap; x.seq_cap = (uint32_t)std::min<uint64_t>(dl / 4 + 16, 0x7FFFFFFFu); x.nblk = 0;
            nblk_cap += x.blk_cap; nslot += x.slot_cap; nseq_cap += x.seq_cap;
            if (nblk_cap > 0x3FFFFFFFull) return fail(c, PNA_E_INVAL, "batch too large for one decode call");
        }
    }
    const bool serial_only = c->tun.zdec_serial != 0;                  // diagnostics: one workgroup per frame for everything
    if (c->z_ents.ensure(n * sizeof(ZEntry)) || c->z_frames.ensure(nfr * sizeof(ZFrame)) || c->z_lit.ensure(out_span + 64) ||
        c->z_fx.ensure(nfr * sizeof(ZFrameX)) || c->z_blocks.ensure(nblk_cap * sizeof(ZBlock)) || c->z_tabs.ensure(nslot * sizeof(ZTables)) ||
        c->z_seqs.ensure(nseq_cap * 8 + 64) || c->z_hlist.ensure(nblk_cap * 16 + 16) || c->z_slist.ensure(nblk_cap * 4 + 16) || c->z_work.ensure(64))
        return fail(c, PNA_E_NOMEM, "decoder workspace");
    HIPCHK(c, hipMemcpyAsync(c->z_ents.p, ents.data(), n * sizeof(ZEntry), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->z_fx.p, fxs.data(), nfr * sizeof(ZFrameX), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(c->z_work.p, 0, 64, st));
    launch_zscan((const ZEntry *)c->z_ents.p, (uint32_t)n, (const uint8_t *)d_src, (ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, st);
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    std::vector<ZFrame> frs(nfr);
    if (!serial_only) {
        // Large frames (the reference writes ONE frame per entry whatever its size): k_zscan has found them -- a frame whose content takes zexec_par_min_mib
        // and more (below 2 GiB: the parallel executor's words hold 31-bit positions).  Their blocks are PARSED side by side (k_zparse_a: the header walk,
        // k_zparse<true>: a wave per block for the tables) and their sequences EXECUTED in parallel by pointer jumping (k_zexec_par.hip) instead of by one
        // wave each; the per-frame kernels skip them (ZFrameX::pad).
        std::vector<uint32_t> big;
        std::vector<ZFrameX> fxd;
        const uint64_t big_min = (uint64_t)c->tun.zexec_par_min_mib << 20;
        if (c->tun.zexec_par_min_mib > 0) {
            bool any = false;
            for (size_t i = 0; i < n && !any; i++) any = raw_len[i] >= big_min;
            if (any) {
                HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, nfr * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                for (uint64_t f = 0; f < nfr; f++)
                    if (frs[f].status == 0 && frs[f].dst_len >= big_min && frs[f].dst_len < (1ull << 31) - 4096) big.push_back((uint32_t)f);
                if (!big.empty()) {
                    if (c->z_big.ensure(big.size() * 4 + 64) || c->z_one.ensure(nblk_cap * 4 + 64)) return fail(c, PNA_E_NOMEM, "decoder workspace");
                    const uint32_t one = 1;
                    for (uint32_t f : big) HIPCHK(c, hipMemcpyAsync((uint8_t *)c->z_fx.p + (size_t)f * sizeof(ZFrameX) + offsetof(ZFrameX, pad), &one, 4, hipMemcpyHostToDevice, st));
                    HIPCHK(c, hipMemcpyAsync(c->z_big.p, big.data(), big.size() * 4, hipMemcpyHostToDevice, st));
                }
            }
        }
        // sequence records of frame f start at seq_base: k_zparse adds it to the block's running count
        launch_zparse((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, (uint32_t)nfr, (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p, (ZTables *)c->z_tabs.p,
                      (uint32_t *)c->z_hlist.p, (uint32_t *)c->z_slist.p, c->z_work.p, st);
        uint32_t work[4] = {0, 0, 0, 0};
        if (!big.empty()) {
            launch_zparse_big_a((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, (const uint32_t *)c->z_big.p, (uint32_t)big.size(), (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p,
                                (uint32_t *)c->z_one.p, c->z_work.p, st);
            fxd.resize(nfr);
            HIPCHK(c, hipMemcpyAsync(work, c->z_work.p, 16, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(fxd.data(), c->z_fx.p, nfr * sizeof(ZFrameX), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            launch_zparse_big_b((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, work[2], (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p, (ZTables *)c->z_tabs.p,
                                (uint32_t *)c->z_hlist.p, (uint32_t *)c->z_slist.p, c->z_work.p, (const uint32_t *)c->z_one.p, st);
        }
        HIPCHK(c, hipMemcpyAsync(work, c->z_work.p, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        launch_zstreams(work[0], work[1], (const uint32_t *)c->z_hlist.p, (const uint32_t *)c->z_slist.p, c->z_work.p, (ZBlock *)c->z_blocks.p,
                        (const ZFrame *)c->z_frames.p, (const ZTables *)c->z_tabs.p, (const uint8_t *)d_src, (uint8_t *)c->z_lit.p, (uint64_t *)c->z_seqs.p, st);
        launch_zexec((ZFrame *)c->z_frames.p, (const ZFrameX *)c->z_fx.p, (uint32_t)nfr, (ZBlock *)c->z_blocks.p, (const uint8_t *)d_src,
                     (const uint8_t *)c->z_lit.p, (const uint64_t *)c->z_seqs.p, (uint8_t *)d_dst, st);
        if (!big.empty()) {
            HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, nfr * sizeof(ZFrame), hipMemcpyDeviceToHost, st));      // (k_zoff has fixed the sizes of open frames)
            HIPCHK(c, hipStreamSynchronize(st));
            for (uint32_t f : big) {
                if (frs[f].status) continue;
                ZxFrame h{frs[f].dst_off, frs[f].dst_len, fxd[f].blk_base, fxd[f].nblk, 0, 0};
                if (c->z_words.ensure(h.dst_len * 4 + 4096) || c->z_rep.ensure((size_t)h.nblk * 24 + 64) || c->z_zxf.ensure(64)) return fail(c, PNA_E_NOMEM, "decoder workspace");
                HIPCHK(c, hipMemcpyAsync(c->z_zxf.p, &h, sizeof h, hipMemcpyHostToDevice, st));
                uint32_t zst = 0, rounds = 0;
                if (launch_zexec_par((ZxFrame *)c->z_zxf.p, h, (const ZBlock *)c->z_blocks.p, (const uint8_t *)d_src, (const uint8_t *)c->z_lit.p, (uint64_t *)c->z_seqs.p,
                                     (uint32_t *)c->z_rep.p, (uint32_t *)c->z_words.p, (uint8_t *)d_dst, &zst, &rounds, st) != 0) return fail(c, PNA_E_HIP, "parallel frame execution failed");
                c->zexec_par_rounds = rounds;
                if (zst) {                                            // 2: the serial kernel takes the frame (it decodes from the source again); 3: corrupt
                    const uint32_t code = zst == 2 ? 2u : 1u;
                    HIPCHK(c, hipMemcpyAsync((uint8_t *)c->z_frames.p + (size_t)f * sizeof(ZFrame) + offsetof(ZFrame, status), &code, 4, hipMemcpyHostToDevice, st));
                }
            }
        }
        launch_zxxh((ZFrame *)c->z_frames.p, (uint32_t)nfr, (const uint8_t *)d_src, (const uint8_t *)d_dst, st);   // frames that carry a content checksum
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, nfr * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    } else {
        HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, nfr * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        for (auto &fr : frs) if (fr.status == 0) fr.status = 2;        // route every well-formed frame through the fallback below
    }
    // ---- frames the bounded pipeline could not take: one workgroup per frame
    std::vector<uint32_t> fb;
    for (uint64_t f = 0; f < nfr; f++) if (frs[f].status == 2) fb.push_back((uint32_t)f);
    if (!fb.empty()) {
        std::vector<ZFrame> sub(fb.size());
        for (size_t k = 0; k < fb.size(); k++) { sub[k] = frs[fb[k]]; sub[k].status = 0; sub[k].out_len = 0; }
        if (open)                                                 // the frame that closes a stream of unknown size keeps its flag
            for (size_t k = 0; k < fb.size(); k++)
                for (size_t i = 0; i < n; i++) {
                    const uint32_t f0 = ents[i].first_frame, f1 = f0 + ents[i].n_frames;
                    if (fb[k] >= f0 && fb[k] < f1 && (fb[k] + 1 == f1 || (fb[k] == f0 && f1 - f0 > 1 && frs[f0 + 1].status == 4))) sub[k].out_len = ZF_OPEN;
                }
        if (c->z_fb.ensure(sub.size() * sizeof(ZFrame)) || c->z_lit.ensure(std::max<uint64_t>(out_span + 64, sub.size() * (uint64_t)(128u << 10) + 64)))
            return fail(c, PNA_E_NOMEM, "decoder workspace");
        HIPCHK(c, hipMemcpyAsync(c->z_fb.p, sub.data(), sub.size() * sizeof(ZFrame), hipMemcpyHostToDevice, st));
        launch_zdec((ZFrame *)c->z_fb.p, (uint32_t)sub.size(), (const uint8_t *)d_src, (uint8_t *)d_dst, (uint8_t *)c->z_lit.p, (uint32_t)c->tun.zdec_dbg, st);
        launch_zxxh((ZFrame *)c->z_fb.p, (uint32_t)sub.size(), (const uint8_t *)d_src, (const uint8_t *)d_dst, st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(sub.data(), c->z_fb.p, sub.size() * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        for (size_t k = 0; k < fb.size(); k++) frs[fb[k]] = sub[k];
    }
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    HIPCHK(c, hipStreamSynchronize(st));
    float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    c->timing = pna_gpu_timing{}; c->timing.ms_lz = ms;            // decoder time reported in the first stage slot
    for (size_t i = 0; i < n; i++)
        for (uint32_t f = 0; f < ents[i].n_frames; f++) {
            const ZFrame &fr = frs[ents[i].first_frame + f];
            if (fr.status && fr.status != 4) {                    // 4: void slot behind a single frame that holds the whole entry
                char msg[160];
                snprintf(msg, sizeof msg, "entry %zu frame %u: %s (produced %u of %llu bytes)", i, f,
                         fr.status == 2 ? "unsupported stream" : (fr.status == 3 ? "size mismatch (foreign multi-frame stream?)" : "corrupt stream"), fr.out_len, (unsigned long long)fr.dst_len);
                return fail(c, fr.status == 2 ? PNA_E_UNSUPPORTED : PNA_E_INVAL, msg);
            }
        }
    if (open && raw_out)
        for (size_t i = 0; i < n; i++) {                          // sizes found by the decoder: frames in front hold SEG_SIZE each
            uint64_t total = 0;
            for (uint32_t f = 0; f < ents[i].n_frames; f++) { const ZFrame &fr = frs[ents[i].first_frame + f]; if (fr.status != 4) total += fr.dst_len; }
            raw_out[i] = total;
        }
    return PNA_OK;
}

// The same for payloads in host memory (extract / verify of an archive read from disk).
extern "C" int pna_gpu_decompress_batch(pna_gpu_ctx *c, int algo, size_t n, const void *const *src, const size_t *src_len,
                                        void *const *dst, const size_t *raw_len) {
    if (!c || (n && (!src || !src_len || !dst || !raw_len))) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "only zstd and deflate streams are decoded on the device");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> so(n), sl(n), dof(n), rl(n);
    uint64_t sp = 0, dp = 0;
    for (size_t i = 0; i < n; i++) { so[i] = sp; sl[i] = src_len[i]; sp = (sp + src_len[i] + 15) & ~(uint64_t)15; dof[i] = dp; rl[i] = raw_len[i]; dp = (dp + raw_len[i] + 15) & ~(uint64_t)15; }
    if (c->stage_in.ensure(sp + 64) || c->stage_out.ensure(dp + 64)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    for (size_t i = 0; i < n; i++) if (src_len[i]) HIPCHK(c, hipMemcpyAsync((uint8_t *)c->stage_in.p + so[i], src[i], src_len[i], hipMemcpyHostToDevice, c->stream));
    int rc = pna_gpu_decompress_batch_device(c, algo, n, c->stage_in.p, so.data(), sl.data(), c->stage_out.p, dof.data(), rl.data(), nullptr);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) if (raw_len[i]) HIPCHK(c, hipMemcpyAsync(dst[i], (uint8_t *)c->stage_out.p + dof[i], raw_len[i], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// The seam is used the way the reference uses its encoders: one writer per rayon task, many tasks in flight on many host threads
// (cli/src/command/core.rs:505-517).  One entry per device batch would leave the GPU idle, so finish() is a GROUP COMMIT: the
// stream joins the context's queue; the first thread to find no leader becomes the leader, takes everything queued so far, runs
// ONE pna_gpu_compress_batch for it and wakes the owners, each of which drains its own stream into its own sink on its own thread
// (W::write is never called from a foreign thread).  While a batch runs, the finishes that arrive pile up and form the next,
// larger batch -- no timer needed under load (PNA_STREAM_LINGER_US adds an optional wait for stragglers).
// one large copy on several threads (a pageable stream of GiBs, e.g. pna_gpu_compress_solid over a whole solid archive)
static void big_memcpy(uint8_t *dst, const uint8_t *src, size_t n) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = n < (64u << 20) ? 1u : std::min(8u, std::max(1u, hw / 2));
    if (T == 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = ((n + T - 1) / T + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < T; t++) {
        const size_t a = std::min(n, (size_t)t * per), b = std::min(n, a + per);
        if (b > a) th.emplace_back([=]() { memcpy(dst + a, src + a, b - a); });
    }
    for (auto &x : th) x.join();
}
constexpr size_t S_SLAB = 1u << 20, S_ARENA = 64u << 20, S_MAX_SLABS = 256;   // a stream beyond 256 MiB continues in pageable memory
struct pna_gpu_stream {
    pna_gpu_ctx *ctx; int algo, level; pna_sink_fn sink; void *user;
    std::vector<uint8_t *> slabs; size_t slab_len = 0;      // page-locked mode: bytes [k * S_SLAB, ...) live in slabs[k]
    bool pageable = false; std::vector<uint8_t> buf;        // pageable mode (pool exhausted / very large stream): everything in buf
    const uint8_t *out = nullptr; size_t out_len = 0; int rc = PNA_OK, slot = 0; bool done = false, queued = false;
    size_t total() const { return pageable ? buf.size() : slab_len; }
};

static uint8_t *pool_get(pna_gpu_ctx *c) {
    std::lock_guard<std::mutex> lk(c->pool_mu);
    if (c->pool_free.empty()) {
        if (c->pool_bytes + S_ARENA > c->pool_cap) return nullptr;
        void *p = nullptr;
        if (hipSetDevice(c->device) != hipSuccess || hipHostMalloc(&p, S_ARENA, hipHostMallocDefault) != hipSuccess) return nullptr;
        c->pool_arenas.push_back(p); c->pool_bytes += S_ARENA;
        for (size_t k = 0; k < S_ARENA / S_SLAB; k++) c->pool_free.push_back((uint8_t *)p + k * S_SLAB);
    }
    uint8_t *r = c->pool_free.back(); c->pool_free.pop_back();
    return r;
}
static void pool_put(pna_gpu_ctx *c, std::vector<uint8_t *> &slabs) {
    if (slabs.empty()) return;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    for (uint8_t *p : slabs) c->pool_free.push_back(p);
    slabs.clear();
}

extern "C" int pna_gpu_stream_new(pna_gpu_ctx *c, int algo, int level, pna_sink_fn sink, void *user, pna_gpu_stream **out) {
    if (!c || !sink || !out) return PNA_E_INVAL;
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return PNA_E_UNSUPPORTED;   // no fail(): other threads may be inside the context
    pna_gpu_stream *s = new (std::nothrow) pna_gpu_stream();
    if (!s) return PNA_E_NOMEM;
    s->ctx = c; s->algo = algo; s->level = level; s->sink = sink; s->user = user;
    *out = s;
    return PNA_OK;
}
extern "C" int pna_gpu_stream_write(pna_gpu_stream *s, const void *buf, size_t len) {
    if (!s || (!buf && len)) return PNA_E_INVAL;
    const uint8_t *p = (const uint8_t *)buf; size_t left = len;
    try {
        while (left && !s->pageable) {
            const size_t in_slab = s->slab_len % S_SLAB;
            if (s->slab_len == s->slabs.size() * S_SLAB) {                  // the last slab is full (or there is none yet)
                uint8_t *sl = s->slabs.size() < S_MAX_SLABS ? pool_get(s->ctx) : nullptr;
                if (!sl) {                                                  // continue in pageable memory
                    s->buf.reserve(s->slab_len + left);
                    for (size_t k = 0; k < s->slabs.size(); k++) s->buf.insert(s->buf.end(), s->slabs[k], s->slabs[k] + std::min(S_SLAB, s->slab_len - k * S_SLAB));
                    pool_put(s->ctx, s->slabs); s->slab_len = 0; s->pageable = true;
                    break;
                }
                s->slabs.push_back(sl);
            }
            const size_t k = std::min(left, S_SLAB - in_slab);
            memcpy(s->slabs.back() + in_slab, p, k); p += k; left -= k; s->slab_len += k;
        }
        if (left) s->buf.insert(s->buf.end(), p, p + left);
    } catch (const std::bad_alloc &) { return PNA_E_NOMEM; }
    return PNA_OK;
}
extern "C" int pna_gpu_stream_flush(pna_gpu_stream *s) { return s ? PNA_OK : PNA_E_INVAL; }
extern "C" void pna_gpu_stream_abort(pna_gpu_stream *s) { if (s) { pool_put(s->ctx, s->slabs); delete s; } }

// One batch of the facade = the streams a leader took, per (algo, level) group: (1) H2D copies straight from the streams' page-locked slabs (pageable
// streams are staged first) on the copy-in stream; `in_done()` then hands the leader's role on, and the next batch is copied in while this one runs;
// (2) the device batch under run_mu; (3) ONE D2H copy of the group's streams into the slot's page-locked output on the copy-out stream.
static void stream_run_batch(pna_gpu_ctx *c, const std::vector<pna_gpu_stream *> &batch, int slot, const std::function<void()> &in_done, const std::function<void()> &on_device,
                             const std::function<void()> &off_device) {
    auto set_err = [&](int code, const char *what) { std::lock_guard<std::mutex> lk(c->err_mu); return fail(c, code, what); };
    auto fail_all = [&](int rc) { for (pna_gpu_stream *x : batch) if (x->rc == PNA_OK && !x->out) { x->rc = rc; x->out_len = 0; } };
    struct Grp { std::vector<pna_gpu_stream *> st; std::vector<uint64_t> off, len, doff; uint64_t in_base = 0, in_bytes = 0, bound = 0, out_base = 0; int rc = PNA_OK; };
    std::vector<Grp> groups;
    static const bool trace = getenv("PNA_STREAM_TRACE") != nullptr;         // per-batch phase times on stderr
    const auto t0 = std::chrono::steady_clock::now();
    // ---- stage 1: plan + copy in (the leader still holds comb_leader: one batch at a time in this stage)
    int rc0 = PNA_OK;
    {
        if (hipSetDevice(c->device) != hipSuccess) rc0 = set_err(PNA_E_HIP, "hipSetDevice failed");
        if (rc0 == PNA_OK && !c->s_h2d) {
            if (hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking) != hipSuccess) rc0 = set_err(PNA_E_HIP, "stream creation failed");
            for (auto &e : c->s_ev) if (rc0 == PNA_OK && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc0 = set_err(PNA_E_HIP, "event creation failed");
        }
        std::vector<char> taken(batch.size(), 0);
        uint64_t in_total = 0, bound_total = 0, out_cap = 64, page_bytes = 0;
        for (size_t i = 0; i < batch.size(); i++) {
            if (taken[i]) continue;
            Grp g;
            for (size_t q = i; q < batch.size(); q++)
                if (!taken[q] && batch[q]->algo == batch[i]->algo && batch[q]->level == batch[i]->level) { taken[q] = 1; g.st.push_back(batch[q]); }
            const size_t n = g.st.size();
            g.off.resize(n + 1); g.len.resize(n); g.doff.resize(n + 1);
            uint64_t pos = 0;
            for (size_t k = 0; k < n; k++) { g.off[k] = pos; g.len[k] = g.st[k]->total(); pos = (pos + g.len[k] + 15) & ~(uint64_t)15; g.bound += pna_gpu_bound(g.st[k]->algo, (size_t)g.len[k]) + 16; }
            g.off[n] = pos; g.in_bytes = pos;
            g.in_base = in_total; in_total += (pos + 8192 + 255) & ~(uint64_t)255;
            g.out_base = bound_total; bound_total += (g.bound + 64 + 255) & ~(uint64_t)255;
            groups.push_back(std::move(g));
        }
        for (pna_gpu_stream *x : batch) { out_cap += pna_gpu_bound(x->algo, x->total()) + 32; if (x->pageable) page_bytes += (x->buf.size() + 15) & ~(size_t)15; }
        // (buffers of a slot are sized for a full batch at once: growing them batch by batch cost the first seconds of a run 10 - 20 ms of page-locking each)
        const uint64_t capb = ((uint64_t)c->tun.stream_batch_mib << 20), cap_out = pna_gpu_bound(PNA_ALGO_DEFLATE, (size_t)capb) + (capb >> 12) + (1u << 20);
        size_t nsl = 0;
        for (pna_gpu_stream *x : batch) nsl += x->slabs.size() + 1;
        if (rc0 == PNA_OK && (c->s_out[slot].ensure(std::max<uint64_t>(out_cap + 256 * groups.size(), cap_out)) || c->st_in[slot].ensure(std::max<uint64_t>(in_total + 64, capb + (1u << 20))) ||
                              c->st_out[slot].ensure(std::max<uint64_t>(bound_total + 64, cap_out)) || c->s_segs[slot].ensure(std::max<size_t>(nsl, 4096) * 24) ||
                              (page_bytes && c->hp_in[0].ensure(page_bytes + 64)))) rc0 = set_err(PNA_E_NOMEM, "staging allocation failed");
        struct LinkSegH { const uint8_t *src; uint8_t *dst; uint64_t len; };
        LinkSegH *lsg = (LinkSegH *)c->s_segs[slot].p; uint32_t nlsg = 0;
        uint64_t ppos = 0;
        for (Grp &g : groups) {
            for (size_t k = 0; k < g.st.size() && rc0 == PNA_OK; k++) {
                pna_gpu_stream *x = g.st[k];
                uint8_t *d = (uint8_t *)c->st_in[slot].p + g.in_base + g.off[k];
                if (x->pageable) {
                    if (!x->buf.empty()) {
                        big_memcpy((uint8_t *)c->hp_in[0].p + ppos, x->buf.data(), x->buf.size());
                        if (hipMemcpyAsync(d, (uint8_t *)c->hp_in[0].p + ppos, x->buf.size(), hipMemcpyHostToDevice, c->s_h2d) != hipSuccess) rc0 = set_err(PNA_E_HIP, "H2D copy failed");
                        ppos += (x->buf.size() + 15) & ~(size_t)15;
                    }
                } else {
                    for (size_t b = 0; b < x->slabs.size() && rc0 == PNA_OK; b++) {
                        const size_t nb = std::min(S_SLAB, x->slab_len - b * S_SLAB);
                        lsg[nlsg++] = LinkSegH{x->slabs[b], d + b * S_SLAB, nb};       // (the slabs are page-locked and device-mapped: one kernel reads them all)
                    }
                }
            }
        }
        if (rc0 == PNA_OK && nlsg) {
            if (c->tun.stream_gather_wgs) { launch_link_gather(lsg, nlsg, (uint32_t)c->tun.stream_gather_wgs, c->s_h2d); if (hipGetLastError() != hipSuccess) rc0 = set_err(PNA_E_HIP, "copy-in kernel failed"); }
            else for (uint32_t q = 0; q < nlsg && rc0 == PNA_OK; q++)
                if (hipMemcpyAsync(lsg[q].dst, lsg[q].src, lsg[q].len, hipMemcpyHostToDevice, c->s_h2d) != hipSuccess) rc0 = set_err(PNA_E_HIP, "H2D copy failed");
        }
        if (rc0 == PNA_OK && hipStreamSynchronize(c->s_h2d) != hipSuccess) rc0 = set_err(PNA_E_HIP, "H2D copy failed");
    }
    in_done();                                                           // the leader's role is free (the next batch is taken once this one is on the device)
    if (rc0 != PNA_OK) { on_device(); off_device(); fail_all(rc0); return; }
    const auto t1 = std::chrono::steady_clock::now();
    // ---- stage 2: the device batch (the context's kernels and workspaces: one at a time)
    auto t2 = t1;
    {
        std::lock_guard<std::mutex> run(c->run_mu);
        on_device();                                                     // the next leader may take its batch and copy it in beside this one's kernels
        t2 = std::chrono::steady_clock::now();
        (void)hipSetDevice(c->device);
        for (Grp &g : groups)
            g.rc = pna_gpu_compress_batch_device(c, g.st[0]->algo, g.st[0]->level, g.st.size(), (uint8_t *)c->st_in[slot].p + g.in_base, g.off.data(), g.len.data(),
                                                 (uint8_t *)c->st_out[slot].p + g.out_base, g.bound + 64, g.doff.data(), nullptr);
        off_device();
    }
    const auto t3 = std::chrono::steady_clock::now();
    // ---- stage 3: the streams travel back (the next batch's kernels are running by now)
    uint64_t hpos = 0;
    for (Grp &g : groups) {
        const size_t n = g.st.size();
        if (g.rc == PNA_OK && g.doff[n] && hipMemcpyAsync((uint8_t *)c->s_out[slot].p + hpos, (uint8_t *)c->st_out[slot].p + g.out_base, g.doff[n], hipMemcpyDeviceToHost, c->s_d2h) != hipSuccess)
            g.rc = set_err(PNA_E_HIP, "D2H copy failed");
        for (size_t k = 0; k < n; k++) {
            g.st[k]->out = (const uint8_t *)c->s_out[slot].p + hpos + (g.rc == PNA_OK ? g.doff[k] : 0);
            g.st[k]->out_len = g.rc == PNA_OK ? (size_t)(g.doff[k + 1] - g.doff[k]) : 0;
        }
        if (g.rc == PNA_OK) hpos += (g.doff[n] + 255) & ~(uint64_t)255;
    }
    bool ok = hipEventRecord(c->s_ev[slot], c->s_d2h) == hipSuccess && hipEventSynchronize(c->s_ev[slot]) == hipSuccess;
    for (Grp &g : groups) { if (!ok && g.rc == PNA_OK) g.rc = set_err(PNA_E_HIP, "device batch failed"); for (pna_gpu_stream *x : g.st) { x->rc = g.rc; if (g.rc != PNA_OK) x->out_len = 0; } }
    if (trace) {
        const auto t4 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        uint64_t inb = 0; for (Grp &g : groups) inb += g.in_bytes;
        fprintf(stderr, "[pna stream batch] slot %d, %zu entries, %.1f MiB in: copy in %.2f ms, wait for the device %.2f ms, device batch %.2f ms, copy out %.2f ms\n",
                slot, batch.size(), inb / 1048576.0, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
    }
}

extern "C" int pna_gpu_stream_finish(pna_gpu_stream *s) {
    if (!s) return PNA_E_INVAL;
    pna_gpu_ctx *c = s->ctx;
    {
        std::unique_lock<std::mutex> lk(c->comb_mu);
        c->comb_queue.push_back(s); s->queued = true;
        c->gate_cv.notify_one();                                     // (the leader may be waiting for the queue to grow)
        while (!s->done) {
            // (a stream the current leader left in the queue -- its batch was full -- waits for the leader's role like a new one)
            if (c->comb_leader || !s->queued) { c->comb_cv.wait(lk); continue; }
            c->comb_leader = true;                                   // s is still queued, so the batch taken below contains it
            const int slot = (int)(c->comb_seq++ % pna_gpu_ctx::S_SLOTS);
            // the batch three before this one is still being drained from this slot; a batch is copied in and waits for the device; the device is busy and
            // the queue is still small
            for (;;) {
                uint64_t qb = 0;
                for (pna_gpu_stream *x : c->comb_queue) qb += x->total();
                if (!c->slot_pending[slot] && !c->staged_waiting && (!c->device_busy || qb >= ((uint64_t)c->tun.stream_overlap_mib << 20))) break;
                c->gate_cv.wait(lk);
            }
            {   // a short linger lets the other writers of the pool reach their finish(): with T writers in flight the batches then hold ~T
                // entries instead of T / 2 (two alternating cohorts) -- 16 threads: 1.5 -> 2.7 GiB/s, 4: 0.40 -> 0.73, 64: 4.8 -> 5.4.
                // Adaptive default: 200 us (a few % of a batch's latency) once more than one writer has been seen, none for a lone writer
                const uint32_t lg = c->comb_linger_us != 0xFFFFFFFFu ? c->comb_linger_us : ((c->comb_last > 1 || c->comb_queue.size() > 1) ? 200u : 0u);
                if (lg) { lk.unlock(); std::this_thread::sleep_for(std::chrono::microseconds(lg)); lk.lock(); }
            }
            // the batch: the queue's streams in arrival order up to stream_batch_mib of input -- s itself always (it may be anywhere in the queue)
            std::vector<pna_gpu_stream *> batch, rest;
            {
                const uint64_t cap = (uint64_t)c->tun.stream_batch_mib << 20;
                uint64_t bytes = s->total();
                batch.push_back(s);
                for (pna_gpu_stream *x : c->comb_queue) {
                    if (x == s) continue;
                    if (bytes + x->total() <= cap) { batch.push_back(x); bytes += x->total(); } else rest.push_back(x);
                }
                c->comb_queue.swap(rest);
            }
            c->slot_pending[slot] = batch.size(); c->comb_last = batch.size();
            for (pna_gpu_stream *x : batch) { x->slot = slot; x->queued = false; }
            lk.unlock();
            stream_run_batch(c, batch, slot, [&]() { std::lock_guard<std::mutex> g(c->comb_mu); c->comb_leader = false; c->staged_waiting++; c->comb_cv.notify_all(); },
                             [&]() { std::lock_guard<std::mutex> g(c->comb_mu); c->staged_waiting--; c->device_busy = true; c->gate_cv.notify_one(); },
                             [&]() { std::lock_guard<std::mutex> g(c->comb_mu); c->device_busy = false; c->gate_cv.notify_one(); });
            lk.lock();
            for (pna_gpu_stream *x : batch) x->done = true;          // owners may free their streams as soon as the lock is released
            c->comb_batches++; c->comb_entries += batch.size(); c->comb_max = std::max<uint64_t>(c->comb_max, batch.size());
            c->comb_cv.notify_all();
        }
    }
    pool_put(c, s->slabs);                                           // the input has been copied to the device
    int rc = s->rc;
    if (rc == PNA_OK) {
        // the reference's zstd writer drains in bursts of at most 32 KiB (zio::Writer); keep that shape
        for (size_t p = 0; p < s->out_len && rc == PNA_OK; p += 32768) {
            const size_t k = std::min<size_t>(32768, s->out_len - p);
            if (s->sink(s->user, s->out + p, k) != 0) { std::lock_guard<std::mutex> run(c->err_mu); rc = fail(c, PNA_E_SINK, "sink failed"); }
        }
    }
    {
        std::lock_guard<std::mutex> lk(c->comb_mu);
        if (--c->slot_pending[s->slot] == 0) c->gate_cv.notify_one();
    }
    delete s;
    return rc;
}
extern "C" int pna_gpu_stream_stats(pna_gpu_ctx *c, uint64_t *batches, uint64_t *entries, uint64_t *largest_batch) {
    if (!c) return PNA_E_INVAL;
    std::lock_guard<std::mutex> lk(c->comb_mu);
    if (batches) *batches = c->comb_batches;
    if (entries) *entries = c->comb_entries;
    if (largest_batch) *largest_batch = c->comb_max;
    return PNA_OK;
}

// Benchmark support: the reference's fan-out restated on host threads (cli/src/command/core.rs:496-537) over the streaming facade --
// `threads` workers take entries FIFO, each entry = stream_new / write (whole entry in one call, core.rs:900-902) / finish into a
// counting sink.  Returns the seconds spent; *out_bytes = compressed bytes seen by the sinks.
// ---- Archive::write_file / write_stream_entry (lib/src/archive/write.rs:276-299,730-777): FHED, extra + metadata chunks, the compressed
// stream as one FDAT chunk per encoder burst (ChunkStreamWriter::write, lib/src/chunk/write.rs:32-47), FEND; no fSIZ.
struct pna_gpu_entry_writer { pna_gpu_ctx *ctx; pna_gpu_stream *st; pna_sink_fn sink; void *user; uint32_t max_chunk; };
static int entry_writer_chunk(pna_gpu_entry_writer *w, const char ty[4], const uint8_t *data, size_t len) {
    uint8_t head[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, (uint8_t)ty[0], (uint8_t)ty[1], (uint8_t)ty[2], (uint8_t)ty[3]};
    const uint32_t crc = pna_crc32(pna_crc32(0, ty, 4), data, len);
    const uint8_t tail[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    if (w->sink(w->user, head, 8) != 0 || (len && w->sink(w->user, data, len) != 0) || w->sink(w->user, tail, 4) != 0) return 1;
    return 0;
}
static int entry_writer_burst(void *u, const void *buf, size_t len) {       // one encoder burst -> FDAT chunk(s) of at most max_chunk bytes
    pna_gpu_entry_writer *w = (pna_gpu_entry_writer *)u;
    const uint8_t *p = (const uint8_t *)buf;
    while (len) {
        const size_t k = std::min<size_t>(len, w->max_chunk);
        if (entry_writer_chunk(w, "FDAT", p, k)) return 1;
        p += k; len -= k;
    }
    return 0;
}
extern "C" int pna_gpu_stream_entry_begin(pna_gpu_ctx *c, int algo, int level, const char *name, const void *meta, size_t meta_len,
                                          uint32_t max_chunk_size, pna_sink_fn sink, void *user, pna_gpu_entry_writer **out) {
    if (!c || !name || !sink || !out || (meta_len && !meta)) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    if (meta_len && !meta_blob_ok((const uint8_t *)meta, meta_len)) return fail(c, PNA_E_INVAL, "extra / metadata chunks are not well-formed chunks");
    pna_gpu_entry_writer *w = new (std::nothrow) pna_gpu_entry_writer{c, nullptr, sink, user, max_chunk_size ? max_chunk_size : 0xFFFFFFFFu};
    if (!w) return fail(c, PNA_E_NOMEM, "out of memory");
    int rc = pna_gpu_stream_new(c, algo, level, entry_writer_burst, w, &w->st);
    if (rc) { delete w; return rc; }
    const std::vector<uint8_t> fh = frame_fhed_bytes(name, algo, 0, 1);           // cipher_mode CTR (1) when unencrypted, lib/src/entry/options.rs:156-159
    if (entry_writer_chunk(w, "FHED", fh.data(), fh.size()) || (meta_len && sink(user, meta, meta_len) != 0)) {
        pna_gpu_stream_abort(w->st); delete w; return fail(c, PNA_E_SINK, "sink failed");
    }
    *out = w;
    return PNA_OK;
}
extern "C" int pna_gpu_stream_entry_write(pna_gpu_entry_writer *w, const void *buf, size_t len) { return w ? pna_gpu_stream_write(w->st, buf, len) : PNA_E_INVAL; }
extern "C" int pna_gpu_stream_entry_finish(pna_gpu_entry_writer *w) {
    if (!w) return PNA_E_INVAL;
    int rc = pna_gpu_stream_finish(w->st);                                        // consumes the stream; the bursts went through entry_writer_burst
    if (rc == PNA_OK && entry_writer_chunk(w, "FEND", nullptr, 0)) rc = fail(w->ctx, PNA_E_SINK, "sink failed");
    delete w;
    return rc;
}
extern "C" void pna_gpu_stream_entry_abort(pna_gpu_entry_writer *w) { if (w) { pna_gpu_stream_abort(w->st); delete w; } }

static int counting_sink(void *user, const void *, size_t len) { ((std::atomic<uint64_t> *)user)->fetch_add(len, std::memory_order_relaxed); return 0; }
extern "C" double pna_bench_stream_threads(pna_gpu_ctx *c, int algo, int level, unsigned threads, size_t n, const void *const *src,
                                           const size_t *src_len, uint64_t *out_bytes, int *rc_out) {
    if (!c || !threads || (n && (!src || !src_len))) { if (rc_out) *rc_out = PNA_E_INVAL; return 0.0; }
    std::atomic<size_t> next{0}; std::atomic<uint64_t> total{0}; std::atomic<int> rc_all{PNA_OK};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back([&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) break;
                pna_gpu_stream *s = nullptr;
                int rc = pna_gpu_stream_new(c, algo, level, counting_sink, &total, &s);
                if (rc == PNA_OK) { rc = pna_gpu_stream_write(s, src[i], src_len[i]); if (rc != PNA_OK) pna_gpu_stream_abort(s); }
                if (rc == PNA_OK) rc = pna_gpu_stream_finish(s);
                if (rc != PNA_OK) { rc_all.store(rc); break; }
            }
        });
    for (auto &x : th) x.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out_bytes) *out_bytes = total.load();
    if (rc_out) *rc_out = rc_all.load();
    return secs;
}

extern "C" int pna_gpu_compress_solid(pna_gpu_ctx *c, int algo, int level, const void *src, size_t src_len,
                                      pna_sink_fn sink, void *user) {
    if (!c || !sink || (!src && src_len)) return PNA_E_INVAL;
    pna_gpu_stream *s = nullptr;
    int rc = pna_gpu_stream_new(c, algo, level, sink, user, &s);
    if (rc) return rc;
    rc = pna_gpu_stream_write(s, src, src_len);
    if (rc) { pna_gpu_stream_abort(s); return rc; }
    return pna_gpu_stream_finish(s);
}

extern "C" int pna_gpu_debug_block(pna_gpu_ctx *c, uint32_t block, uint64_t *seqs, uint32_t cap_seqs, uint32_t *nseq,
                                   uint8_t *lits, uint32_t cap_lits, uint32_t *nlit) {
    if (!c || block >= c->last_nblk) return PNA_E_INVAL;
    HIPCHK(c, hipSetDevice(c->device));
    BlkInfo bi;
    HIPCHK(c, hipMemcpy(&bi, (BlkInfo *)c->blk.p + block, sizeof(bi), hipMemcpyDeviceToHost));
    if (nseq) *nseq = bi.nseq;
    if (nlit) *nlit = bi.nlit;
    if (seqs) HIPCHK(c, hipMemcpy(seqs, (uint64_t *)c->seqs.p + (size_t)block * seq_cap_of(c->last_blk_log), (size_t)std::min(cap_seqs, bi.nseq) * 8, hipMemcpyDeviceToHost));
    if (lits) HIPCHK(c, hipMemcpy(lits, (uint8_t *)c->lits.p + ((size_t)block << c->last_blk_log), std::min(cap_lits, bi.nlit), hipMemcpyDeviceToHost));
    return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// corpus tables (integer-only; same construction as the checker's corpus model, written independently here)
static uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int ensure_corpus(pna_gpu_ctx *c) {
    if (c->corpus_ready) return PNA_OK;
    const int VOCAB = 50000, SLOT = 16, NPHRASE = 8192;
    static const uint16_t LCUM[26] = {817, 966, 1244, 1669, 2939, 3162, 3364, 3973, 4670, 4685, 4762, 5165, 5406,
                                      6081, 6832, 7025, 7035, 7634, 8267, 9173, 9449, 9547, 9783, 9798, 9995, 10000};
    std::vector<uint8_t> vocab((size_t)VOCAB * SLOT, 0); std::vector<uint64_t> cum(VOCAB); std::vector<uint32_t> phr((size_t)NPHRASE * 4, 0);
    uint64_t s = 0x504E41ull;
    for (int w = 0; w < VOCAB; w++) {
        uint64_t r = splitmix(s);
        int len = 2 + (int)((r & 0xFFFF) * 11 >> 16);
        if (w < 64) len = 2 + (int)((r & 0xFFFF) * 3 >> 16);
        else if (w < 1024) len = 3 + (int)((r & 0xFFFF) * 5 >> 16);
        uint8_t *slot = &vocab[(size_t)w * SLOT];
        slot[0] = (uint8_t)len;
        for (int i = 0; i < len; i++) { uint32_t x = (uint32_t)(splitmix(s) >> 33) % 10000u; int ch = 0; while (LCUM[ch] <= x) ch++; slot[1 + i] = (uint8_t)('a' + ch); }
    }
    uint64_t acc = 0;
    for (int k = 0; k < VOCAB; k++) { acc += (1ull << 40) / (uint64_t)(k + 1); cum[k] = acc; }
    auto draw = [&](uint64_t r, int n) {
        unsigned __int128 m = (unsigned __int128)r * cum[n - 1]; uint64_t x = (uint64_t)(m >> 64);
        int lo = 0, hi = n - 1; while (lo < hi) { int mid = (lo + hi) >> 1; if (cum[mid] > x) hi = mid; else lo = mid + 1; } return lo;
    };
    for (int p = 0; p < NPHRASE; p++) {
        uint64_t r = splitmix(s);
        phr[4 * p] = (uint32_t)(2 + (int)(r & 1));
        for (int i = 0; i < 3; i++) phr[4 * p + 1 + i] = (uint32_t)draw(splitmix(s), VOCAB);
    }
    if (c->c_vocab.ensure(vocab.size()) || c->c_cum.ensure(cum.size() * 8) || c->c_phr.ensure(phr.size() * 4)) return fail(c, PNA_E_NOMEM, "corpus tables");
    HIPCHK(c, hipMemcpy(c->c_vocab.p, vocab.data(), vocab.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->c_cum.p, cum.data(), cum.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->c_phr.p, phr.data(), phr.size() * 4, hipMemcpyHostToDevice));
    c->corpus_ready = true;
    return PNA_OK;
}

extern "C" int pna_bench_corpus_fill_device(pna_gpu_ctx *c, int kind, uint64_t first_file, uint64_t n_files, uint64_t file_len,
                                            uint64_t stride, void *d_dst, void *hip_stream) {
    if (!c || !d_dst || kind < 0 || kind > 4 || stride < file_len) return fail(c, PNA_E_INVAL, "bad corpus argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_corpus(c); if (rc) return rc;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    launch_corpus(kind, first_file, n_files, file_len, stride, (const uint8_t *)c->c_vocab.p, (const uint64_t *)c->c_cum.p,
                  (const uint32_t *)c->c_phr.p, (uint8_t *)d_dst, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    return PNA_OK;
}

extern "C" int pna_gpu_debug_lz_stamps(pna_gpu_ctx *c, unsigned long long *out8) {
    if (!c || !out8) return PNA_E_INVAL;
    HIPCHK(c, hipSetDevice(c->device));
    lz_read_stamps(out8);
    return PNA_OK;
}
