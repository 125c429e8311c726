// tests/san/device_stub.cpp -- TEST INFRASTRUCTURE: the launch layer of libpna_gpu.so on the CPU for the sanitizer builds.
// The zstd write path is stubbed with a trivially valid encoder (every segment = one frame of RAW blocks; the empty entry = the
// reference's 9-byte frame), the framing kernel (prefix + CRC-32 + FEND) is restated bytewise; everything else aborts: the sanitizer
// driver exercises the HOST code around the kernels, not the codecs (those are checked against the oracle on the GPU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "pna_dev.h"
#include "../../include/pna_archive.h"

namespace pna {
[[noreturn]] static void nostub(const char *what) { fprintf(stderr, "device_stub: %s is not stubbed\n", what); abort(); }

void launch_lz(const uint8_t *, const SegDesc *, uint32_t, uint64_t *, uint8_t *, BlkInfo *, uint4 *, uint32_t, uint32_t, uint32_t, hipStream_t, uint32_t *, uint32_t, hipEvent_t, uint32_t *, const LzParseGrid *) {}
void launch_lz_small(const uint8_t *, const SegDesc *, uint32_t, uint64_t *, uint8_t *, BlkInfo *, uint4 *, uint32_t, uint32_t, hipStream_t, uint32_t *, uint32_t, const LzParseGrid *, bool) {}
void launch_default_tables(hipStream_t) {}
uint32_t lz_gtab_log() { return 19; }
void launch_entropy_chunk(const SegDesc *, uint32_t, uint32_t, const uint32_t *, uint32_t, uint32_t, const uint64_t *, const uint8_t *, BlkInfo *, SegTables *,
                          uint8_t *, uint8_t *, uint32_t *, uint32_t, uint32_t, uint32_t *, hipStream_t, hipEvent_t *, hipStream_t, hipEvent_t, hipEvent_t, bool, const uint32_t *) {}
static uint64_t seg_bytes(const SegDesc &sd) {
    if (sd.len == 0) return 9;
    return 6 + 3ull * seg_nblk(sd) + sd.len;
}
void launch_plan(const SegDesc *segs, uint32_t nseg, BlkInfo *, const SegTables *, uint64_t *seg_size, uint64_t *seg_off, uint32_t, hipStream_t) {
    uint64_t pos = 0;
    for (uint32_t s = 0; s < nseg; s++) { seg_size[s] = seg_bytes(segs[s]); seg_off[s] = pos; pos += seg_size[s]; }
    seg_off[nseg] = pos;
}
void launch_write(const uint8_t *src, const SegDesc *segs, uint32_t nseg, const uint32_t *, uint32_t, const BlkInfo *, const SegTables *,
                  const uint64_t *seg_off, const uint8_t *, const uint8_t *, const uint8_t *, uint8_t *dst, bool, hipStream_t, bool) {
    for (uint32_t s = 0; s < nseg; s++) {
        const SegDesc &sd = segs[s];
        uint8_t *o = dst + seg_off[s];
        if (sd.len == 0) { static const uint8_t e[9] = {0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00}; memcpy(o, e, 9); continue; }
        static const uint8_t h[6] = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x50};
        memcpy(o, h, 6); o += 6;
        const uint32_t bsz = 1u << sd.blk_log;                                      // (small batches run on smaller blocks: latency mode)
        for (uint32_t b0 = 0; b0 < sd.len; b0 += bsz) {
            const uint32_t bl = sd.len - b0 < bsz ? sd.len - b0 : bsz;
            const uint32_t hd = (b0 + bl == sd.len ? 1u : 0u) | (bl << 3);          // last | raw (0) << 1 | size << 3
            o[0] = (uint8_t)hd; o[1] = (uint8_t)(hd >> 8); o[2] = (uint8_t)(hd >> 16);
            memcpy(o + 3, src + sd.src_off + b0, bl); o += 3 + bl;
        }
    }
}
void launch_frame(const FrameDesc *fd, uint32_t n, const uint8_t *blob, const CrcTabs *, uint8_t *dst, uint64_t, uint32_t fend_crc, const char ty[4], bool with_fend, hipStream_t, uint32_t) {
    for (uint32_t i = 0; i < n; i++) {
        const FrameDesc &d = fd[i];
        memcpy(dst + d.arc_off, blob + d.prefix_off, d.prefix_len);
        uint8_t *pay = dst + d.arc_off + d.prefix_len;
        const uint32_t crc = pna_crc32(pna_crc32(0, ty, 4), pay, d.payload_len);
        uint8_t *q = pay + d.payload_len;
        q[0] = (uint8_t)(crc >> 24); q[1] = (uint8_t)(crc >> 16); q[2] = (uint8_t)(crc >> 8); q[3] = (uint8_t)crc;
        if (with_fend && !(d.pad & 2)) {
            const uint8_t fe[12] = {0, 0, 0, 0, 'F', 'E', 'N', 'D', (uint8_t)(fend_crc >> 24), (uint8_t)(fend_crc >> 16), (uint8_t)(fend_crc >> 8), (uint8_t)fend_crc};
            memcpy(q + 4, fe, 12);
        }
    }
}
struct PlaceDesc { uint64_t src_off, dst_off; uint32_t len, pad; };
void launch_place(const void *pd, uint32_t n, const uint8_t *src, uint8_t *dst, hipStream_t) {
    for (uint32_t i = 0; i < n; i++) { const PlaceDesc &d = ((const PlaceDesc *)pd)[i]; memcpy(dst + d.dst_off, src + d.src_off, d.len); }
}
void launch_gather(const void *pd, uint32_t n, const uint8_t *src, uint8_t *dst, hipStream_t st) { launch_place(pd, n, src, dst, st); }
void launch_layout(FrameDesc *fd, uint8_t *blob, const uint32_t *entry_seg, const uint64_t *seg_off, uint32_t nentry, uint32_t nseg, uint64_t out_base,
                   uint64_t *segdst, uint64_t *ent_off, uint64_t *total, hipStream_t) {
    uint64_t pos = out_base;
    for (uint32_t i = 0; i < nentry; i++) {
        const uint32_t s0 = entry_seg[i], s1 = entry_seg[i + 1];
        const uint64_t base = seg_off[s0], plen = seg_off[s1] - base;
        fd[i].arc_off = pos; fd[i].payload_len = (uint32_t)plen;
        uint8_t *lenf = blob + fd[i].prefix_off + fd[i].prefix_len - 8;
        lenf[0] = (uint8_t)(plen >> 24); lenf[1] = (uint8_t)(plen >> 16); lenf[2] = (uint8_t)(plen >> 8); lenf[3] = (uint8_t)plen;
        for (uint32_t sg = s0; sg < s1; sg++) segdst[sg] = pos + fd[i].prefix_len + (seg_off[sg] - base);
        ent_off[i] = pos;
        pos += fd[i].prefix_len + plen + 16;
    }
    segdst[nseg] = pos; ent_off[nentry] = pos; *total = pos - out_base;
}
void launch_link_copy(const uint8_t *src, uint8_t *dst, size_t n, uint32_t, hipStream_t) { memcpy(dst, src, n); }
void launch_link_gather(const void *segs, uint32_t nseg, uint32_t, hipStream_t) {
    struct Seg { const uint8_t *src; uint8_t *dst; uint64_t len; };
    for (uint32_t i = 0; i < nseg; i++) memcpy(((const Seg *)segs)[i].dst, ((const Seg *)segs)[i].src, ((const Seg *)segs)[i].len);
}
void lz_read_stamps(unsigned long long *out) { memset(out, 0, 8 * sizeof *out); }

void launch_deflate_stage1(const uint8_t *, const SegDesc *, uint32_t, const uint32_t *, uint32_t, const uint64_t *, const uint8_t *, BlkInfo *, const uint4 *, DeflTables *,
                           uint8_t *, uint64_t *, uint64_t *, hipStream_t, hipEvent_t *, uint32_t, bool, bool) { nostub("deflate"); }
void launch_deflate_write(const uint8_t *, const SegDesc *, const uint32_t *, uint32_t, const BlkInfo *, const uint64_t *, const uint64_t *, const uint8_t *, const uint32_t *,
                          uint32_t, uint8_t *, hipStream_t, bool, bool) { nostub("deflate"); }
void launch_frame_verify(const FrameDesc *, uint32_t, const CrcTabs *, const uint8_t *, uint64_t, const char[4], uint32_t *, hipStream_t, uint32_t) { nostub("frame_verify"); }
void launch_zdec(ZFrame *, uint32_t, const uint8_t *, uint8_t *, uint8_t *, uint32_t, hipStream_t) { nostub("zdec"); }
void launch_zparse_big_a(ZFrame *, ZFrameX *, const uint32_t *, uint32_t, const uint8_t *, ZBlock *, uint32_t *, void *, hipStream_t) { nostub("zparse"); }
void launch_zparse_big_b(ZFrame *, ZFrameX *, uint32_t, const uint8_t *, ZBlock *, ZTables *, uint32_t *, uint32_t *, void *, const uint32_t *, hipStream_t) { nostub("zparse"); }
struct ZxFrame;
int launch_zexec_par(ZxFrame *, const ZxFrame &, const ZBlock *, const uint8_t *, const uint8_t *, uint64_t *, uint32_t *, uint32_t *, uint8_t *, uint32_t *, uint32_t *, hipStream_t, uint32_t, const uint32_t *, const uint64_t *) { nostub("zexec_par"); return -1; }
void launch_zxxh(ZFrame *, uint32_t, const uint8_t *, const uint8_t *, hipStream_t) { nostub("zxxh"); }
void launch_zscan(const ZEntry *, uint32_t, const uint8_t *, ZFrame *, ZFrameX *, hipStream_t) { nostub("zscan"); }
void launch_zcount(const ZEntry *, uint32_t, const uint8_t *, uint32_t *, hipStream_t) { nostub("zcount"); }
void launch_zlist(const uint8_t *, uint64_t, uint64_t, uint64_t, void *, uint32_t, uint64_t *, hipStream_t) { nostub("zlist"); }
void launch_frame_pieces(const FrameDesc *, uint32_t, const CrcTabs *, const uint8_t *, uint64_t, const char[4], uint32_t *, hipStream_t) { nostub("frame_pieces"); }
void launch_crc_patch(const void *, uint32_t, uint8_t *, hipStream_t) { nostub("crc_patch"); }
void launch_zparse(ZFrame *, ZFrameX *, uint32_t, const uint8_t *, ZBlock *, ZTables *, uint32_t *, uint32_t *, void *, hipStream_t) { nostub("zparse"); }
void launch_zstreams(uint32_t, uint32_t, const uint32_t *, const uint32_t *, const void *, ZBlock *, const ZFrame *, const ZTables *, const uint8_t *, uint8_t *, uint64_t *, hipStream_t) { nostub("zstreams"); }
void launch_inflate(ZFrame *, ZFrameX *, uint32_t, const uint8_t *, ZBlock *, uint8_t *, uint64_t *, const uint32_t *, hipStream_t) { nostub("inflate"); }
void launch_ispec(const uint8_t *, uint64_t, uint64_t, uint32_t, uint32_t, uint64_t *, hipStream_t) { nostub("inflate (chunk starts)"); }
void launch_inflate_chunks(ZFrame *, ZFrameX *, uint32_t, void *, uint32_t, uint32_t, const uint8_t *, ZBlock *, uint8_t *, uint64_t *, hipStream_t) { nostub("inflate (chunks)"); }
void launch_icount(const uint8_t *, const uint64_t *, const uint64_t *, uint32_t, uint32_t *, uint32_t, hipStream_t) { nostub("inflate"); }
void launch_vinflate(ZFrame *, ZFrameX *, uint32_t, const void *, uint32_t, uint64_t *, uint32_t *, uint32_t *, uint32_t, const uint8_t *, ZBlock *, uint8_t *, uint64_t *, hipStream_t) { nostub("inflate"); }
void launch_iadler(ZFrame *, const ZFrameX *, const ZBlock *, uint32_t, const uint32_t *, uint32_t, const uint8_t *, void *, hipStream_t) { nostub("iadler"); }
void launch_zexec(ZFrame *, const ZFrameX *, uint32_t, ZBlock *, const uint8_t *, const uint8_t *, const uint64_t *, uint8_t *, hipStream_t) { nostub("zexec"); }
void launch_zexec_groups(ZFrame *, const ZFrameX *, uint32_t, ZBlock *, const void *, uint32_t, const uint8_t *, const uint8_t *, const uint64_t *, uint8_t *, hipStream_t) { nostub("zexec"); }
void launch_gcm_tag(const GcmEntry *, uint32_t, uint8_t *, hipStream_t) { nostub("gcm"); }
void launch_gcm_verify(const GcmEntry *, uint32_t, const uint8_t *, const uint8_t *, uint32_t *, hipStream_t) { nostub("gcm"); }
void launch_aes_cbc_dec(const CipherUnit *, uint32_t, const uint8_t *, const AesDecTabs *, uint8_t *, const AesKey &, uint32_t *, hipStream_t) { nostub("aes"); }
void launch_aes_ctr(const CipherUnit *, uint32_t, const uint8_t *, const AesTabs *, uint8_t *, const AesKey &, const AesKey *, hipStream_t) { nostub("aes"); }
void launch_aes_cbc_enc(const CipherUnit *, uint32_t, const uint8_t *, const AesTabs *, uint8_t *, const AesKey &, hipStream_t) { nostub("aes"); }
void launch_corpus(int, uint64_t, uint64_t, uint64_t, uint64_t, const uint8_t *, const uint64_t *, const uint32_t *, uint8_t *, hipStream_t) { nostub("corpus"); }
} // namespace pna
