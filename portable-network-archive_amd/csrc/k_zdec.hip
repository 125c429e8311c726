// k_zdec.hip -- zstd frame decoder for gfx950: the read side of the compression seam,
// decompress_reader() -> zstd::stream::read::Decoder (lib/src/entry/read.rs:171-190), used by extract / verify
// (cli/src/command/extract.rs:594-640, verify.rs:140-188).  General RFC 8878 frames without dictionary (what the reference
// and this repository's encoder write): raw / RLE / compressed blocks, raw / RLE / Huffman / treeless literals with direct
// or FSE-compressed weights, predefined / RLE / FSE / repeat sequence tables, repeat offsets.
//
// One 256-thread workgroup per frame, blocks in order (a block may copy from everything before it in the frame):
//   thread 0 parses the block and section headers and builds the Huffman weights / FSE distributions (short serial codes),
//   all threads fill the Huffman decoding table, 1 or 4 lanes decode the literal streams into the frame's 128 KiB scratch slot
//   (bit windows in registers, next word prefetched), lane 0 of wave 0 runs the serial FSE chain in batches of 64 sequences and
//   the wave executes each batch one sequence per lane: positions by a wave scan, literal runs in parallel, matches in
//   dependency passes with a workgroup-scope fence between them.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "pna_dev.h"

namespace pna {

constexpr uint32_t ZD_THREADS = 256;
constexpr int ZD_HUF_MAX = 11;
constexpr uint32_t ZD_STAGE_W = 6144;                 // 48 KiB
constexpr uint32_t ZD_BATCH = 64;                     // sequences decoded by lane 0, then executed one per lane
enum { ZD_OK = 0, ZD_CORRUPT = 1, ZD_UNSUPPORTED = 2, ZD_DSTSIZE = 3 };

typedef unsigned long long zd_u64u __attribute__((aligned(1)));

__constant__ int16_t ZD_LL_DEF[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
__constant__ int16_t ZD_ML_DEF[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
__constant__ int16_t ZD_OF_DEF[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
__constant__ uint32_t ZD_LL_BASE[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
__constant__ uint8_t  ZD_LL_BITS[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
__constant__ uint32_t ZD_ML_BASE[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
__constant__ uint8_t  ZD_ML_BITS[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};

__device__ __forceinline__ int zd_hb(uint32_t v) { return 31 - (int)__builtin_clz(v); }           // v != 0

// n bits (<= 56) at bit offset off >= 0 of the little-endian bit string p; reads 8 bytes at p + off / 8 (the source buffer
// carries 8 bytes of slack behind its last stream)
__device__ __forceinline__ uint64_t zd_bits(const uint8_t *p, int64_t off, uint32_t n) {
    const uint64_t v = *(const zd_u64u *)(p + (off >> 3)) >> (off & 7);
    return v & (((uint64_t)1 << n) - 1);
}
// backward reader: `off` = number of unread bits below the cursor; bits below the start of the string read as zero
struct ZdBits { const uint8_t *p; int64_t off; };
__device__ __forceinline__ bool zd_binit(ZdBits &b, const uint8_t *p, uint32_t len) {
    if (len == 0) return false;
    const uint32_t last = p[len - 1];
    if (last == 0) return false;
    b.p = p; b.off = (int64_t)len * 8 - (8 - zd_hb(last));
    return true;
}
__device__ __forceinline__ uint64_t zd_bread(ZdBits &b, uint32_t n) {
    if (n == 0) return 0;
    b.off -= n;
    if (b.off >= 0) return zd_bits(b.p, b.off, n);
    const int64_t have = (int64_t)n + b.off;                       // bits that exist
    if (have <= 0) return 0;
    return zd_bits(b.p, 0, (uint32_t)have) << (uint32_t)(-b.off);
}

// The same reader with a 128-bit window in registers (words k, k+1 of the string, 8 bytes each) and word k-1 already requested.
// `fetch(j)` returns word j of the string: from the LDS copy of the stream when it fits (the serial Huffman / FSE chains then
// never wait for HBM), else straight from global memory.
struct ZdWin { int64_t off; int64_t k; uint64_t lo, hi, pre; };
template <class F> __device__ __forceinline__ bool zd_winit(ZdWin &w, F &&fetch, uint32_t last_byte, uint32_t len) {
    if (len == 0 || last_byte == 0) return false;
    w.off = (int64_t)len * 8 - (8 - zd_hb(last_byte));
    w.k = ((w.off + 63) >> 6) - 2; if (w.k < 0) w.k = 0;
    w.lo = fetch(w.k); w.hi = fetch(w.k + 1); w.pre = w.k > 0 ? fetch(w.k - 1) : 0;
    return true;
}
// the n (<= 57) bits below the cursor, not consumed
template <class F> __device__ __forceinline__ uint64_t zd_wpeek(ZdWin &w, F &&fetch, uint32_t n) {
    const int64_t lo_bit = w.off - (int64_t)n;
    const int64_t need = lo_bit < 0 ? 0 : lo_bit;
    while (need < 64 * w.k) { w.hi = w.lo; w.lo = w.pre; w.k--; w.pre = w.k > 0 ? fetch(w.k - 1) : 0; }
    const uint32_t sft = (uint32_t)(need - 64 * w.k);              // 0..127
    uint64_t v = sft < 64 ? (w.lo >> sft) | (sft ? w.hi << (64 - sft) : 0) : w.hi >> (sft - 64);
    if (lo_bit < 0) {                                              // part of the field lies below the start: zero there
        if (w.off <= 0) return 0;
        v = (v & (((uint64_t)1 << (uint32_t)w.off) - 1)) << (uint32_t)(-lo_bit);
    }
    return v & (((uint64_t)1 << n) - 1);
}
template <class F> __device__ __forceinline__ uint64_t zd_wread(ZdWin &w, F &&fetch, uint32_t n) {
    if (n == 0) return 0;
    const uint64_t v = zd_wpeek(w, fetch, n);
    w.off -= n;
    return v;
}

// FSE decoding table (sym | nbits << 8 | base << 16) from a normalised distribution; serial (one thread)
__device__ bool zd_fse_build(uint32_t *tab, const int16_t *norm, int nsym, int alog, uint16_t *next /* [256] scratch */) {
    if (alog > 9) return false;
    const int size = 1 << alog;
    int high = size - 1;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) { tab[high--] = (uint32_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int i = 0; i < norm[s]; i++) { tab[pos] = (uint32_t)s; do { pos = (pos + step) & mask; } while (pos > high); }
    if (pos != 0) return false;
    for (int u = 0; u < size; u++) {
        const uint32_t s = tab[u] & 0xFF;
        const uint32_t ns = next[s]++;
        const uint32_t nb = (uint32_t)(alog - zd_hb(ns));
        tab[u] = s | (nb << 8) | ((((ns << nb) - (uint32_t)size) & 0xFFFF) << 16);
    }
    return true;
}

// FSE table description (forward bits); returns bytes consumed, 0 on error
__device__ uint32_t zd_fse_desc(const uint8_t *src, uint32_t len, int16_t *norm, int *nsym_out, int *alog_out, int max_sym, int max_alog) {
    if (len < 1) return 0;
    uint64_t bitpos = 0;
    auto peek = [&](uint32_t n) -> uint32_t {
        uint64_t v = 0; const uint64_t byte = bitpos >> 3;
        for (int i = 0; i < 5; i++) if (byte + i < len) v |= (uint64_t)src[byte + i] << (8 * i);
        return (uint32_t)((v >> (bitpos & 7)) & (((uint64_t)1 << n) - 1));
    };
    const int alog = (int)peek(4) + 5; bitpos += 4;
    if (alog > max_alog) return 0;
    int remaining = (1 << alog) + 1, threshold = 1 << alog, nbits = alog + 1, sym = 0;
    for (int i = 0; i <= max_sym; i++) norm[i] = 0;
    while (remaining > 1 && sym <= max_sym) {
        const int mx = (2 * threshold - 1) - remaining;
        int count;
        const uint32_t v = peek((uint32_t)nbits);
        if ((int)(v & (uint32_t)(threshold - 1)) < mx) { count = (int)(v & (uint32_t)(threshold - 1)); bitpos += (uint32_t)(nbits - 1); }
        else { count = (int)(v & (uint32_t)(2 * threshold - 1)); if (count >= threshold) count -= mx; bitpos += (uint32_t)nbits; }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        if (count == 0) {
            for (;;) {
                const uint32_t rep = peek(2); bitpos += 2;
                for (uint32_t i = 0; i < rep && sym <= max_sym; i++) norm[sym++] = 0;
                if (rep != 3) break;
            }
        }
        while (remaining < threshold) { nbits--; threshold >>= 1; }
        if ((bitpos + 7) / 8 > len) return 0;
    }
    if (remaining != 1 || sym > max_sym + 1) return 0;
    *nsym_out = sym; *alog_out = alog;
    return (uint32_t)((bitpos + 7) / 8);
}

struct ZdBlk {                 // what thread 0 tells the workgroup about the current block
    uint32_t status;           // ZD_*
    uint32_t last, type, size; // block header
    uint32_t ltype, regen, streams, lit_off, lit_csize;   // literals: payload offset (block-relative) and compressed size behind the tree
    uint32_t new_tree, nweights, maxbits;
    uint32_t nseq, seq_off, seq_len;                      // sequence bitstream (block-relative)
    uint32_t alog[3];
};

__global__ __launch_bounds__(ZD_THREADS)
void k_zdec(ZFrame *__restrict__ frames, const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint8_t *__restrict__ lit_scratch, uint32_t dbg) {
    __shared__ uint16_t huf_tab[1 << ZD_HUF_MAX];      // sym | nbits << 8
    __shared__ uint32_t fse_tab[3][512];               // LL, OF, ML
    __shared__ uint8_t  weights[256];
    __shared__ int16_t  norm[256];
    __shared__ uint16_t nexts[256];
    __shared__ ZdBlk    B;
    __shared__ uint64_t sq[ZD_BATCH];                  // zrec_pack(ll, ml, resolved offset)
    __shared__ uint32_t wtab[64];                      // weight-decoding table (alog <= 6)
    __shared__ uint64_t stage64[ZD_STAGE_W];           // LDS copy of the stream(s) being decoded (8-byte aligned starts)
    __shared__ uint32_t s_nb, s_ok[3], s_alog[3], s_huf_ok, s_hufbits, s_err, s_op, s_rep[3];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ZFrame fr = frames[blockIdx.x];
    const uint8_t *in = src + fr.src_off;
    uint8_t *out = dst + fr.dst_off;
    // Positions (ip, op) count in 32 bits from `in` / `out`; for a frame of 4 GiB and more (one frame per entry is what the reference writes,
    // whatever the entry's size: tests/bats/large_file.bats) both bases move along at block boundaries -- the input's to the block, the output's
    // to 2 GiB behind it (offsets are below 2^28) --, in_len and cap are what is left from there, saturated.
    uint64_t in_left = fr.src_len, cap_left = fr.dst_len, out_base = 0;
    uint32_t in_len = (uint32_t)(in_left < 0xFFFFFFFFull ? in_left : 0xFFFFFFFFull), cap = (uint32_t)(cap_left < 0xFFFFFFFFull ? cap_left : 0xFFFFFFFFull);
    const bool open = (fr.out_len & ZF_OPEN) != 0;                     // cap is only a capacity
    uint32_t ip = 0, op = 0;
    // (dbg & 8: the bases move every few MiB instead of every few GiB, for frames with windows of at most 8 MiB -- the test of this logic)
    const uint32_t rb_in = (dbg & 8) ? (1u << 16) : (1u << 30), rb_hi = (dbg & 8) ? (3u << 22) : (3u << 30), rb_keep = (dbg & 8) ? (1u << 23) : (1u << 31);
    uint32_t status = ZD_OK;
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
    if (tid == 0) { s_ok[0] = s_ok[1] = s_ok[2] = 0; s_huf_ok = 0; s_err = 0; }

    // ---- frame header (uniform)
    {
        if (in_len < 6) status = ZD_CORRUPT;
        else {
            const uint32_t magic = in[0] | (in[1] << 8) | (in[2] << 16) | ((uint32_t)in[3] << 24);
            if (magic != 0xFD2FB528u) status = ZD_CORRUPT;
            else {
                const uint32_t fhd = in[4];
                const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, dict = fhd & 3;
                ip = 5;
                if (fhd & 0x08) status = ZD_CORRUPT;
                if (!single) ip += 1;                                    // window descriptor: the whole frame is addressable here
                if (dict) status = ZD_UNSUPPORTED;
                const uint32_t fsz = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2u : (fcs_flag == 2 ? 4u : 8u));
                if (ip + fsz > in_len) status = ZD_CORRUPT;
                else {
                    uint64_t fcs = 0;
                    for (uint32_t i = 0; i < fsz; i++) fcs |= (uint64_t)in[ip + i] << (8 * i);
                    if (fsz == 2) fcs += 256;
                    if (fsz && (open ? fcs > fr.dst_len : fcs != fr.dst_len)) status = ZD_DSTSIZE;
                    ip += fsz;
                }
                if ((fhd >> 2) & 1) { /* content checksum: 4 bytes after the last block, verified by k_zxxh once the content is there */ }
            }
        }
    }
    __syncthreads();

    if (fr.status) status = fr.status;                                 // set by the scan
    bool done = status != ZD_OK;
    while (!done) {
        if (ip > rb_in) { in += ip; in_left -= ip; ip = 0; in_len = (uint32_t)(in_left < 0xFFFFFFFFull ? in_left : 0xFFFFFFFFull); }
        if (op > rb_hi) {
            const uint32_t shift = op - rb_keep;
            out += shift; out_base += shift; cap_left -= shift; op -= shift; cap = (uint32_t)(cap_left < 0xFFFFFFFFull ? cap_left : 0xFFFFFFFFull);
        }
        // ================= thread 0: headers, weights, distributions
        if (tid == 0) {
            ZdBlk b; b.status = ZD_OK; b.new_tree = 0; b.nseq = 0; b.regen = 0; b.ltype = 0; b.streams = 1; b.lit_off = 0; b.lit_csize = 0;
            b.seq_off = 0; b.seq_len = 0; b.nweights = 0; b.maxbits = 0; b.alog[0] = b.alog[1] = b.alog[2] = 0;
            b.last = 1; b.type = 0; b.size = 0;
            do {
                if (ip + 3 > in_len) { b.status = ZD_CORRUPT; break; }
                const uint32_t bh = in[ip] | (in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16);
                b.last = bh & 1; b.type = (bh >> 1) & 3; b.size = bh >> 3;
                if (b.type == 3 || b.size > (128u << 10)) { b.status = ZD_CORRUPT; break; }
                const uint32_t body = ip + 3;
                if (b.type == 1) { if (body + 1 > in_len) b.status = ZD_CORRUPT; break; }
                if (body + b.size > in_len) { b.status = ZD_CORRUPT; break; }
                if (b.type == 0) break;
                // ---- compressed block: literals section
                const uint8_t *p = in + body; const uint32_t len = b.size;
                if (len < 1) { b.status = ZD_CORRUPT; break; }
                const uint32_t ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
                uint32_t regen, comp = 0, hdr, streams = 1;
                if (ltype < 2) {
                    if (sf == 0 || sf == 2) { regen = p[0] >> 3; hdr = 1; }
                    else if (sf == 1) { if (len < 2) { b.status = ZD_CORRUPT; break; } regen = (p[0] >> 4) + ((uint32_t)p[1] << 4); hdr = 2; }
                    else { if (len < 3) { b.status = ZD_CORRUPT; break; } regen = (p[0] >> 4) + ((uint32_t)p[1] << 4) + ((uint32_t)p[2] << 12); hdr = 3; }
                } else {
                    if (len < 5) { b.status = ZD_CORRUPT; break; }
                    uint64_t v = 0; for (int i = 0; i < 5; i++) v |= (uint64_t)p[i] << (8 * i);
                    if (sf == 0) { streams = 1; hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; }
                    else if (sf == 1) { streams = 4; hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; }
                    else if (sf == 2) { streams = 4; hdr = 4; regen = (v >> 4) & 0x3FFF; comp = (v >> 18) & 0x3FFF; }
                    else { streams = 4; hdr = 5; regen = (v >> 4) & 0x3FFFF; comp = (v >> 22) & 0x3FFFF; }
                }
                if (regen > (128u << 10) || regen > cap) { b.status = ZD_CORRUPT; break; }
                b.ltype = ltype; b.regen = regen; b.streams = streams;
                uint32_t pos = hdr;
                if (ltype == 0) { if (pos + regen > len) { b.status = ZD_CORRUPT; break; } b.lit_off = pos; pos += regen; }
                else if (ltype == 1) { if (pos + 1 > len) { b.status = ZD_CORRUPT; break; } b.lit_off = pos; pos += 1; }
                else {
                    if (pos + comp > len) { b.status = ZD_CORRUPT; break; }
                    const uint8_t *cs = p + pos; uint32_t cl = comp, used = 0;
                    if (ltype == 2) {
                        // ---- Huffman tree description -> weights[0 .. nw), the last weight is implied
                        if (cl < 1) { b.status = ZD_CORRUPT; break; }
                        const uint32_t hbyte = cs[0];
                        uint32_t nw = 0;
                        if (hbyte >= 128) {
                            nw = hbyte - 127;
                            const uint32_t bytes = (nw + 1) / 2;
                            if (1 + bytes > cl) { b.status = ZD_CORRUPT; break; }
                            for (uint32_t i = 0; i < nw; i++) { const uint32_t by = cs[1 + i / 2]; weights[i] = (uint8_t)((i & 1) ? (by & 15) : (by >> 4)); }
                            used = 1 + bytes;
                        } else {
                            const uint32_t csize = hbyte;
                            if (csize == 0 || 1 + csize > cl) { b.status = ZD_CORRUPT; break; }
                            int nsym, alog;
                            const uint32_t dl = zd_fse_desc(cs + 1, csize, norm, &nsym, &alog, 255, 6);
                            if (!dl || dl >= csize || !zd_fse_build(wtab, norm, nsym, alog, nexts)) { b.status = ZD_CORRUPT; break; }
                            ZdBits bb;
                            if (!zd_binit(bb, cs + 1 + dl, csize - dl)) { b.status = ZD_CORRUPT; break; }
                            uint32_t s1 = (uint32_t)zd_bread(bb, (uint32_t)alog), s2 = (uint32_t)zd_bread(bb, (uint32_t)alog);
                            bool bad = false;
                            for (;;) {
                                if (nw >= 255) { bad = true; break; }
                                weights[nw++] = (uint8_t)(wtab[s1] & 0xFF);
                                s1 = (wtab[s1] >> 16) + (uint32_t)zd_bread(bb, (wtab[s1] >> 8) & 0xFF);
                                if (bb.off < 0) { if (nw >= 255) { bad = true; break; } weights[nw++] = (uint8_t)(wtab[s2] & 0xFF); break; }
                                if (nw >= 255) { bad = true; break; }
                                weights[nw++] = (uint8_t)(wtab[s2] & 0xFF);
                                s2 = (wtab[s2] >> 16) + (uint32_t)zd_bread(bb, (wtab[s2] >> 8) & 0xFF);
                                if (bb.off < 0) { if (nw >= 255) { bad = true; break; } weights[nw++] = (uint8_t)(wtab[s1] & 0xFF); break; }
                            }
                            if (bad) { b.status = ZD_CORRUPT; break; }
                            used = 1 + csize;
                        }
                        uint32_t total = 0; bool badw = false;
                        for (uint32_t i = 0; i < nw; i++) { if (weights[i] > ZD_HUF_MAX) badw = true; else if (weights[i]) total += 1u << (weights[i] - 1); }
                        if (badw || total == 0 || nw < 1) { b.status = ZD_CORRUPT; break; }
                        const uint32_t maxbits = (uint32_t)zd_hb(total) + 1;
                        const uint32_t rest = (1u << maxbits) - total;
                        if (maxbits > (uint32_t)ZD_HUF_MAX || rest == 0 || (rest & (rest - 1))) { b.status = ZD_CORRUPT; break; }
                        weights[nw] = (uint8_t)(zd_hb(rest) + 1);
                        b.new_tree = 1; b.nweights = nw + 1; b.maxbits = maxbits; s_huf_ok = 1; s_hufbits = maxbits;
                    } else if (!s_huf_ok) { b.status = ZD_CORRUPT; break; }
                    if (used > cl) { b.status = ZD_CORRUPT; break; }
                    b.lit_off = pos + used; b.lit_csize = cl - used;
                    pos += comp;
                }
                // ---- sequences section
                if (pos >= len) { b.status = ZD_CORRUPT; break; }
                uint32_t nseq; const uint32_t b0 = p[pos++];
                if (b0 < 128) nseq = b0;
                else if (b0 < 255) { if (pos >= len) { b.status = ZD_CORRUPT; break; } nseq = ((b0 - 128) << 8) + p[pos++]; }
                else { if (pos + 2 > len) { b.status = ZD_CORRUPT; break; } nseq = p[pos] + ((uint32_t)p[pos + 1] << 8) + 0x7F00; pos += 2; }
                b.nseq = nseq;
                if (nseq == 0) { if (pos != len) b.status = ZD_CORRUPT; break; }
                if (pos >= len) { b.status = ZD_CORRUPT; break; }
                const uint32_t modes = p[pos++];
                if (modes & 3) { b.status = ZD_CORRUPT; break; }
                bool ok = true;
                for (int k = 0; k < 3 && ok; k++) {                         // LL, OF, ML in stream order
                    const uint32_t mode = (modes >> (6 - 2 * k)) & 3;
                    const int16_t *def = k == 0 ? ZD_LL_DEF : (k == 1 ? ZD_OF_DEF : ZD_ML_DEF);
                    const int def_n = k == 0 ? 36 : (k == 1 ? 29 : 53), def_log = k == 1 ? 5 : 6;
                    const int max_sym = k == 0 ? 35 : (k == 1 ? 31 : 52), max_log = k == 1 ? 8 : 9;
                    if (mode == 0) {
                        for (int i = 0; i < def_n; i++) norm[i] = def[i];
                        ok = zd_fse_build(fse_tab[k], norm, def_n, def_log, nexts); s_ok[k] = 1; s_alog[k] = (uint32_t)def_log;
                    } else if (mode == 1) {
                        if (pos >= len || p[pos] > max_sym) { ok = false; break; }
                        fse_tab[k][0] = p[pos]; pos++; s_ok[k] = 1; s_alog[k] = 0;
                    } else if (mode == 2) {
                        int nsym, alog;
                        const uint32_t used = zd_fse_desc(p + pos, len - pos, norm, &nsym, &alog, max_sym, max_log);
                        if (!used) { ok = false; break; }
                        ok = zd_fse_build(fse_tab[k], norm, nsym, alog, nexts); pos += used; s_ok[k] = 1; s_alog[k] = (uint32_t)alog;
                    } else if (!s_ok[k]) ok = false;
                    b.alog[k] = s_alog[k];
                }
                if (!ok || pos >= len) { b.status = ZD_CORRUPT; break; }
                b.seq_off = pos; b.seq_len = len - pos;
            } while (false);
            B = b;
        }
        __syncthreads();
        const ZdBlk b = B;
        if (b.status != ZD_OK) { status = b.status; break; }
        const uint32_t body = ip + 3;
        if (b.type == 0 || b.type == 1) {                               // raw / RLE block
            if (op + b.size > cap) { status = ZD_DSTSIZE; break; }
            const uint32_t rle = in[body];
            for (uint32_t i = tid; i < b.size; i += ZD_THREADS) out[op + i] = b.type == 0 ? in[body + i] : (uint8_t)rle;
            op += b.size; ip = body + (b.type == 0 ? b.size : 1);
            __threadfence_block(); __syncthreads();
            if (b.last) break;
            continue;
        }
        const uint8_t *p = in + body;
        // ================= Huffman table (all threads): symbol s of weight w owns 2^(w-1) cells behind all symbols of smaller
        // weight and the lower-numbered symbols of its own weight
        if (b.new_tree) {
            for (uint32_t s = tid; s < b.nweights; s += ZD_THREADS) {
                const uint32_t w = weights[s];
                if (!w) continue;
                uint32_t pos = 0;
                for (uint32_t o = 0; o < b.nweights; o++) { const uint32_t wo = weights[o]; if (wo && (wo < w || (wo == w && o < s))) pos += 1u << (wo - 1); }
                const uint32_t n = 1u << (w - 1), cell = s | ((b.maxbits + 1 - w) << 8);
                for (uint32_t i = 0; i < n; i++) huf_tab[pos + i] = (uint16_t)cell;
            }
        }
        __syncthreads();
        // ================= literals -> tail of the frame's output region
        uint8_t *lit_stage = lit_scratch + (size_t)blockIdx.x * (128u << 10);   // this frame's slot for decoded literals
        if (b.ltype >= 2 && !(dbg & 4)) {
            const uint32_t mb = s_hufbits;
            const uint8_t *cs = p + b.lit_off; const uint32_t cl = b.lit_csize;
            // stream s: bytes [s_off, s_off + s_len) of the section, regenerates o_len bytes at o_off
            uint32_t sof4[4] = {0, 0, 0, 0}, sln4[4] = {cl, 0, 0, 0}, oof4[4] = {0, 0, 0, 0}, oln4[4] = {b.regen, 0, 0, 0};
            bool okh = true;
            if (b.streams == 4) {
                if (cl < 6) okh = false;
                else {
                    const uint32_t l1 = cs[0] | (cs[1] << 8), l2 = cs[2] | (cs[3] << 8), l3 = cs[4] | (cs[5] << 8);
                    const uint32_t seg = (b.regen + 3) / 4;
                    if (6 + l1 + l2 + l3 > cl || seg * 3 > b.regen) okh = false;
                    else {
                        sof4[0] = 6; sln4[0] = l1; sof4[1] = 6 + l1; sln4[1] = l2; sof4[2] = 6 + l1 + l2; sln4[2] = l3;
                        sof4[3] = 6 + l1 + l2 + l3; sln4[3] = cl - sof4[3];
                        for (int q = 0; q < 4; q++) { oof4[q] = q * seg; oln4[q] = q < 3 ? seg : b.regen - 3 * seg; }
                    }
                }
            }
            // LDS copies of the streams (each starts on an 8-byte boundary, 16 bytes of slack behind the last)
            uint32_t lw4[4] = {0, 0, 0, 0}, words = 0;
            for (uint32_t q = 0; q < b.streams; q++) { lw4[q] = words; words += (sln4[q] + 7) / 8 + 1; }
            const bool staged = okh && words + 2 <= ZD_STAGE_W;
            if (staged) {
                uint8_t *st8 = (uint8_t *)stage64;
                for (uint32_t q = 0; q < b.streams; q++)
                    for (uint32_t i = tid; i < sln4[q]; i += ZD_THREADS) st8[8 * lw4[q] + i] = cs[sof4[q] + i];
                __syncthreads();
            }
            if (okh && lane == 0 && wave < b.streams) {
                const uint32_t s_off = sof4[wave], s_len = sln4[wave], o_off = oof4[wave], o_len = oln4[wave];
                const uint32_t last = s_len ? cs[s_off + s_len - 1] : 0u;
                auto run = [&](auto fetch) {
                    ZdWin bw;
                    if (!zd_winit(bw, fetch, last, s_len)) { okh = false; return; }
                    uint32_t i = 0;
                    for (; i + 4 <= o_len; i += 4) {                       // four symbols per store
                        uint32_t word = 0;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint32_t cell = huf_tab[(uint32_t)zd_wpeek(bw, fetch, mb)];
                            word |= (cell & 0xFF) << (8 * q);
                            bw.off -= cell >> 8;
                        }
                        if (bw.off < 0) { okh = false; return; }
                        if (((o_off + i) & 3) == 0) *(uint32_t *)(lit_stage + o_off + i) = word;
                        else { lit_stage[o_off + i] = (uint8_t)word; lit_stage[o_off + i + 1] = (uint8_t)(word >> 8); lit_stage[o_off + i + 2] = (uint8_t)(word >> 16); lit_stage[o_off + i + 3] = (uint8_t)(word >> 24); }
                    }
                    for (; i < o_len; i++) {
                        const uint32_t cell = huf_tab[(uint32_t)zd_wpeek(bw, fetch, mb)];
                        lit_stage[o_off + i] = (uint8_t)cell;
                        bw.off -= cell >> 8;
                        if (bw.off < 0) { okh = false; return; }
                    }
                    if (bw.off != 0) okh = false;
                };
                const uint32_t lw = lw4[wave];
                const uint8_t *gs = cs + s_off;
                if (staged) run([&](int64_t j) -> uint64_t { return stage64[lw + j]; });
                else run([&](int64_t j) -> uint64_t { return *(const zd_u64u *)(gs + 8 * j); });
            }
            if (!okh) atomicOr(&s_err, 1u);
        }
        __threadfence_block();
        __syncthreads();
        if (s_err) { status = ZD_CORRUPT; break; }
        if (b.nseq && (b.seq_len + 7) / 8 + 3 <= ZD_STAGE_W) {                   // LDS copy of the sequence bitstream
            uint8_t *st8 = (uint8_t *)stage64; const uint8_t *gq = p + b.seq_off;
            for (uint32_t i = tid; i < b.seq_len; i += ZD_THREADS) st8[i] = gq[i];
            __syncthreads();
        }
        // ================= sequences: wave 0 alternates "lane 0 decodes 64 sequences (FSE chain, repeat offsets)" and "every lane
        // executes one of them": output positions by a wave scan, literal runs in parallel, then the matches in passes -- a match
        // is ready when its source ends at or below the output start of the first match still pending (usually 1-3 passes).
        if (wave == 0) {
            const uint8_t *lit_raw = p + b.lit_off;
            auto LIT = [&](uint32_t i) -> uint8_t { return b.ltype == 0 ? lit_raw[i] : (b.ltype == 1 ? lit_raw[0] : lit_stage[i]); };
            uint32_t litpos = 0;
            ZdWin bw; uint32_t sll = 0, sof = 0, sml = 0;
            bool okq = true;
            const uint8_t *gq = p + b.seq_off;
            const bool qstaged = (b.seq_len + 7) / 8 + 3 <= ZD_STAGE_W;        // staged by the whole workgroup before this branch
            auto fq = [&](int64_t j) -> uint64_t { return qstaged ? stage64[j] : *(const zd_u64u *)(gq + 8 * j); };
            if (b.nseq) {
                if (!zd_winit(bw, fq, b.seq_len ? gq[b.seq_len - 1] : 0u, b.seq_len)) okq = false;
                else { sll = (uint32_t)zd_wread(bw, fq, b.alog[0]); sof = (uint32_t)zd_wread(bw, fq, b.alog[1]); sml = (uint32_t)zd_wread(bw, fq, b.alog[2]); }
            }
            for (uint32_t base = 0; base < b.nseq && okq && !(dbg & 2); base += ZD_BATCH) {
                const uint32_t nb = b.nseq - base < ZD_BATCH ? b.nseq - base : ZD_BATCH;
                if (lane == 0) {
                    for (uint32_t i = 0; i < nb; i++) {
                        const uint32_t cl = fse_tab[0][sll], co = fse_tab[1][sof], cm = fse_tab[2][sml];
                        const uint32_t llc = cl & 0xFF, ofc = co & 0xFF, mlc = cm & 0xFF;
                        if (ofc > 31 || mlc > 52 || llc > 35) { okq = false; break; }
                        const uint64_t ofv = ((uint64_t)1 << ofc) + zd_wread(bw, fq, ofc);
                        const uint32_t ml = ZD_ML_BASE[mlc] + (uint32_t)zd_wread(bw, fq, ZD_ML_BITS[mlc]);
                        const uint32_t ll = ZD_LL_BASE[llc] + (uint32_t)zd_wread(bw, fq, ZD_LL_BITS[llc]);
                        if (bw.off < 0 || ofv > ZREC_OF_MAX) { okq = false; break; }
                        if (base + i + 1 < b.nseq) {
                            sll = (cl >> 16) + (uint32_t)zd_wread(bw, fq, (cl >> 8) & 0xFF);
                            sml = (cm >> 16) + (uint32_t)zd_wread(bw, fq, (cm >> 8) & 0xFF);
                            sof = (co >> 16) + (uint32_t)zd_wread(bw, fq, (co >> 8) & 0xFF);
                            if (bw.off < 0) { okq = false; break; }
                        }
                        uint32_t offset;
                        if (ofv > 3) { offset = (uint32_t)ofv - 3; rep2 = rep1; rep1 = rep0; rep0 = offset; }
                        else {
                            const uint32_t idx = (uint32_t)ofv - 1 + (ll == 0 ? 1u : 0u);
                            if (idx == 0) offset = rep0;
                            else {
                                offset = idx == 1 ? rep1 : (idx == 2 ? rep2 : rep0 - 1);
                                if (offset == 0) { okq = false; break; }
                                if (idx > 1) rep2 = rep1;
                                rep1 = rep0; rep0 = offset;
                            }
                        }
                        sq[i] = zrec_pack(ll, ml, offset);
                    }
                    if (okq && base + nb == b.nseq && bw.off != 0) okq = false;
                    s_nb = okq ? 1u : 0u;
                }
                __builtin_amdgcn_wave_barrier();
                okq = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_nb) != 0;
                if (!okq) break;
                if (dbg & 1) continue;
                const bool act = lane < nb;
                const uint64_t sv = act ? sq[lane] : 0;
                const uint32_t ll = zrec_ll(sv), ml = zrec_ml(sv), offset = zrec_of(sv);
                uint32_t incl = ll + ml, lincl = ll;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t1 = (uint32_t)__shfl_up((int)incl, d), t2 = (uint32_t)__shfl_up((int)lincl, d);
                    if (lane >= (uint32_t)d) { incl += t1; lincl += t2; }
                }
                const uint32_t o0 = op + incl - (ll + ml), l0 = litpos + lincl - ll;       // this sequence's output / literal start
                const uint32_t T = (uint32_t)__shfl((int)incl, 63), TL = (uint32_t)__shfl((int)lincl, 63);
                const bool bad = act && (l0 + ll > b.regen || (uint64_t)o0 + ll + ml > cap || offset == 0 || offset > o0 + ll);
                if (__ballot(bad)) { okq = false; break; }
                // ---- literal runs: up to 32 bytes per lane in lock step, the longer ones by the whole wave
                {
                    const uint32_t ls = ll < 32 ? ll : 32u;
                    for (uint32_t k = 0; __ballot(k < ls); k++) if (k < ls) out[o0 + k] = LIT(l0 + k);
                    uint64_t longm = __ballot(ll > 32);
                    while (longm) {
                        const uint32_t j = (uint32_t)__builtin_ctzll(longm); longm &= longm - 1;
                        const uint32_t oj = (uint32_t)__shfl((int)o0, (int)j), lj = (uint32_t)__shfl((int)l0, (int)j), nj = (uint32_t)__shfl((int)ll, (int)j);
                        for (uint32_t k = 32 + lane; k < nj; k += 64) out[oj + k] = LIT(lj + k);
                    }
                }
                // ---- matches in passes
                const uint32_t dstp = o0 + ll, m0 = dstp - offset;
                const uint32_t src_end = m0 + (ml < offset ? ml : offset);
                bool pending = act && ml != 0;
                for (;;) {
                    const uint64_t pm = __ballot(pending);
                    if (!pm) break;
                    __threadfence_block();                                       // everything written so far is visible to every lane
                    const uint32_t frontier = (uint32_t)__shfl((int)dstp, (int)__builtin_ctzll(pm));
                    const bool ready = pending && src_end <= frontier;
                    // short ready matches: one per lane, byte-serial inside the lane (an overlapping match reads what it wrote)
                    const bool shortr = ready && ml <= 32;
                    for (uint32_t k = 0; __ballot(shortr && k < ml); k++) if (shortr && k < ml) out[dstp + k] = out[m0 + k];
                    uint64_t longm = __ballot(ready && ml > 32);
                    while (longm) {
                        const uint32_t j = (uint32_t)__builtin_ctzll(longm); longm &= longm - 1;
                        const uint32_t dj = (uint32_t)__shfl((int)dstp, (int)j), mj = (uint32_t)__shfl((int)m0, (int)j);
                        const uint32_t nj = (uint32_t)__shfl((int)ml, (int)j), fj = (uint32_t)__shfl((int)offset, (int)j);
                        // periodic with period fj: every byte comes from the fj bytes in front of the match
                        if (fj >= nj) { for (uint32_t k = lane; k < nj; k += 64) out[dj + k] = out[mj + k]; }
                        else { for (uint32_t k = lane; k < nj; k += 64) out[dj + k] = out[mj + k % fj]; }
                    }
                    pending = pending && !ready;
                }
                op += T; litpos += TL;
            }
            if (okq) {
                const uint32_t rest = b.regen - litpos;
                if ((uint64_t)op + rest > cap) okq = false;
                else { for (uint32_t k = lane; k < rest; k += 64) out[op + k] = LIT(litpos + k); op += rest; }
            }
            if (lane == 0) { s_op = op; s_rep[0] = rep0; s_rep[1] = rep1; s_rep[2] = rep2; if (!okq) s_err = 1; }
        }
        __threadfence_block();
        __syncthreads();
        if (s_err) { status = ZD_CORRUPT; break; }
        op = s_op; rep0 = s_rep[0]; rep1 = s_rep[1]; rep2 = s_rep[2];
        ip = body + b.size;
        __syncthreads();
        if (b.last) break;
    }
    const uint64_t produced = out_base + op;
    if (status == ZD_OK && produced != fr.dst_len && !(dbg & 7) && !open) status = ZD_DSTSIZE;
    if (tid == 0) {
        frames[blockIdx.x].status = status; frames[blockIdx.x].out_len = (uint32_t)(produced < 0xFFFFFFFFull ? produced : 0xFFFFFFFFull);
        if (open && status == ZD_OK) frames[blockIdx.x].dst_len = produced;
    }
}

// =====================================================================================================================
// Lane-parallel pipeline (the default): the serial chains of MANY blocks run side by side, one lane each.
//   k_zparse  one wave per frame: lane 0 walks the blocks (headers, Huffman weights, FSE distributions), tables are built in
//             LDS and copied to the frame's table slots; every block gets a descriptor, every Huffman stream and every block
//             with sequences an entry in a work list
//   k_zhuf    one lane per Huffman stream   (tables in global memory / L2, bit windows in registers)
//   k_zfse    one lane per block: the FSE chain -> sequence records (ll, ml, offset value), block output size
//   k_zoff    one thread per frame: output offsets of its blocks
//   k_zexec   one wave per frame, blocks in order: 64 sequences at a time, one per lane (as in k_zdec)
// Frames that do not fit the bounded per-frame resources (table slots, block descriptors, sequence records) are flagged and
// go through k_zdec.
struct ZWork { uint32_t n_huf, n_seq, pad0, pad1; };

// ONE (round 4): the frame's blocks were laid out by k_zparse_a (a serial walk of the headers only: positions, sizes, table slots) and every wave of this
// launch parses ONE of them -- blockIdx.x = index into `one_list` of block numbers --: what took the single wave of a 2 048-block frame 0.6 s (0.29 ms of
// weight decoding and table building per block on lane 0) runs side by side.  Same code, the running state comes from the block's descriptor.
template <bool ONE>
__global__ __launch_bounds__(64)
void k_zparse(ZFrame *__restrict__ frames, ZFrameX *__restrict__ fx, const uint8_t *__restrict__ src, ZBlock *__restrict__ blocks,
              ZTables *__restrict__ tabs, uint32_t *__restrict__ huf_list, uint32_t *__restrict__ seq_list, ZWork *__restrict__ work, const uint32_t *__restrict__ one_list) {
    __shared__ uint16_t huf_tab[1 << ZD_HUF_MAX];
    __shared__ uint32_t fse_tab[512];
    __shared__ uint8_t  weights[256];
    __shared__ int16_t  norm[256];
    __shared__ uint16_t nexts[256];
    __shared__ uint32_t wtab[64];
    __shared__ uint32_t s_cmd[8];          // lane 0 -> wave: [0] what to flush (1 huffman, 2 fse), [1] slot, [2] which, [3] maxbits / alog, [4] nweights
    const uint32_t lane = threadIdx.x;
    ZBlock pre;                                                         // ONE: what k_zparse_a put down for this wave's block
    if (ONE) pre = blocks[one_list[blockIdx.x]];
    const uint32_t f = ONE ? pre.frame : blockIdx.x;
    ZFrame fr = frames[f];
    ZFrameX x = fx[f];
    if (!ONE && x.pad) return;                                          // a large frame: k_zparse_a + k_zparse<true> take it
    const uint8_t *in = src + fr.src_off;
    const uint32_t in_len = (uint32_t)fr.src_len;
    const uint64_t cap = fr.dst_len < 0xFFFFFFFFull ? fr.dst_len : 0xFFFFFFFFull;   // (literal positions count in 32 bits: a frame with 4 GiB of literals and more in compressed blocks is left to the one-workgroup kernel)
    const uint64_t fcs_want = fr.dst_len;
    const bool open = (fr.out_len & ZF_OPEN) != 0;
    uint32_t status = fr.status, ip = 0;
    // ---- frame header
    if (!ONE && status == ZD_OK) {
        if (in_len < 6) status = ZD_CORRUPT;
        else {
            const uint32_t magic = in[0] | (in[1] << 8) | (in[2] << 16) | ((uint32_t)in[3] << 24);
            const uint32_t fhd = in[4];
            const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, dict = fhd & 3;
            if (magic != 0xFD2FB528u || (fhd & 0x08)) status = ZD_CORRUPT;
            else if (dict) status = ZD_UNSUPPORTED;
            else {
                ip = 5 + (single ? 0 : 1);
                const uint32_t fsz = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2u : (fcs_flag == 2 ? 4u : 8u));
                if (ip + fsz > in_len) status = ZD_CORRUPT;
                else {
                    uint64_t fcs = 0;
                    for (uint32_t i = 0; i < fsz; i++) fcs |= (uint64_t)in[ip + i] << (8 * i);
                    if (fsz == 2) fcs += 256;
                    if (fsz && (open ? fcs > fcs_want : fcs != fcs_want)) status = ZD_DSTSIZE;
                    ip += fsz;
                }
            }
        }
    }
    uint32_t nblk = 0, next_slot = 0, huf_slot = 0xFFFFFFFFu, slot3[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t lit_pos = 0; uint64_t seq_pos = 0;
    if (ONE) {                                                          // the walk's state in front of this block
        nblk = one_list[blockIdx.x] - x.blk_base; next_slot = pre.pad[3]; huf_slot = pre.huf_slot; slot3[0] = pre.slot[0]; slot3[1] = pre.slot[1]; slot3[2] = pre.slot[2];
        lit_pos = pre.lit_pos; seq_pos = pre.seq_pos - x.seq_base; ip = (uint32_t)(pre.body - 3 - fr.src_off);
        if (status != ZD_OK) return;
    }
    bool last = status != ZD_OK;
    while (!last) {
        // lane 0 parses one block; whenever a table is complete in LDS the whole wave copies it out
        ZBlock b;
        uint32_t bstat = ZD_OK;
        if (lane == 0) {
            b.body = 0; b.out_off = 0; b.seq_pos = x.seq_base + seq_pos; b.size = 0; b.type = 0; b.ltype = 0; b.regen = 0; b.streams = 1; b.lit_off = 0; b.lit_csize = 0;
            b.lit_pos = lit_pos; b.nseq = 0; b.seq_off = 0; b.seq_len = 0; b.frame = f; b.out_len = 0; b.status = 0; b.uses_rep = 0;
            for (int i = 0; i < 7; i++) b.pad[i] = 0;
        }
        // ---- phase 1 (lane 0): headers + literals section + Huffman weights
        uint32_t p1_tree = 0;
        if (lane == 0) {
            do {
                if (nblk >= x.blk_cap) { bstat = ZD_UNSUPPORTED; break; }
                if (ip + 3 > in_len) { bstat = ZD_CORRUPT; break; }
                const uint32_t bh = in[ip] | (in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16);
                b.type = (bh >> 1) & 3; b.size = bh >> 3; b.pad[0] = bh & 1;
                b.body = fr.src_off + ip + 3;
                if (b.type == 3 || b.size > (128u << 10)) { bstat = ZD_CORRUPT; break; }
                if (b.type == 1) { if (ip + 4 > in_len) bstat = ZD_CORRUPT; b.out_len = b.size; break; }
                if (ip + 3 + b.size > in_len) { bstat = ZD_CORRUPT; break; }
                if (b.type == 0) { b.out_len = b.size; break; }
                const uint8_t *p = in + ip + 3; const uint32_t len = b.size;
                if (len < 1) { bstat = ZD_CORRUPT; break; }
                const uint32_t ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
                uint32_t regen, comp = 0, hdr, streams = 1;
                if (ltype < 2) {
                    if (sf == 0 || sf == 2) { regen = p[0] >> 3; hdr = 1; }
                    else if (sf == 1) { if (len < 2) { bstat = ZD_CORRUPT; break; } regen = (p[0] >> 4) + ((uint32_t)p[1] << 4); hdr = 2; }
                    else { if (len < 3) { bstat = ZD_CORRUPT; break; } regen = (p[0] >> 4) + ((uint32_t)p[1] << 4) + ((uint32_t)p[2] << 12); hdr = 3; }
                } else {
                    if (len < 5) { bstat = ZD_CORRUPT; break; }
                    uint64_t v = 0; for (int i = 0; i < 5; i++) v |= (uint64_t)p[i] << (8 * i);
                    if (sf == 0) { streams = 1; hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; }
                    else if (sf == 1) { streams = 4; hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; }
                    else if (sf == 2) { streams = 4; hdr = 4; regen = (v >> 4) & 0x3FFF; comp = (v >> 18) & 0x3FFF; }
                    else { streams = 4; hdr = 5; regen = (v >> 4) & 0x3FFFF; comp = (v >> 22) & 0x3FFFF; }
                }
                if (regen > (128u << 10) || (uint64_t)lit_pos + regen > cap) { bstat = regen > (128u << 10) ? ZD_CORRUPT : (fcs_want > cap ? ZD_UNSUPPORTED : ZD_DSTSIZE); break; }
                b.ltype = ltype; b.regen = regen; b.streams = streams;
                uint32_t pos = hdr;
                if (ltype == 0) { if (pos + regen > len) { bstat = ZD_CORRUPT; break; } b.lit_off = pos; pos += regen; }
                else if (ltype == 1) { if (pos + 1 > len) { bstat = ZD_CORRUPT; break; } b.lit_off = pos; pos += 1; }
                else {
                    if (pos + comp > len) { bstat = ZD_CORRUPT; break; }
                    const uint8_t *cs = p + pos; const uint32_t cl = comp; uint32_t used = 0;
                    if (ltype == 2) {
                        if (cl < 1) { bstat = ZD_CORRUPT; break; }
                        const uint32_t hbyte = cs[0];
                        uint32_t nw = 0;
                        if (hbyte >= 128) {
                            nw = hbyte - 127;
                            const uint32_t bytes = (nw + 1) / 2;
                            if (1 + bytes > cl) { bstat = ZD_CORRUPT; break; }
                            for (uint32_t i = 0; i < nw; i++) { const uint32_t by = cs[1 + i / 2]; weights[i] = (uint8_t)((i & 1) ? (by & 15) : (by >> 4)); }
                            used = 1 + bytes;
                        } else {
                            const uint32_t csize = hbyte;
                            if (csize == 0 || 1 + csize > cl) { bstat = ZD_CORRUPT; break; }
                            int nsym, alog;
                            const uint32_t dl = zd_fse_desc(cs + 1, csize, norm, &nsym, &alog, 255, 6);
                            if (!dl || dl >= csize || !zd_fse_build(wtab, norm, nsym, alog, nexts)) { bstat = ZD_CORRUPT; break; }
                            ZdBits bb;
                            if (!zd_binit(bb, cs + 1 + dl, csize - dl)) { bstat = ZD_CORRUPT; break; }
                            uint32_t s1 = (uint32_t)zd_bread(bb, (uint32_t)alog), s2 = (uint32_t)zd_bread(bb, (uint32_t)alog);
                            bool bad = false;
                            for (;;) {
                                if (nw >= 255) { bad = true; break; }
                                weights[nw++] = (uint8_t)(wtab[s1] & 0xFF);
                                s1 = (wtab[s1] >> 16) + (uint32_t)zd_bread(bb, (wtab[s1] >> 8) & 0xFF);
                                if (bb.off < 0) { if (nw >= 255) { bad = true; break; } weights[nw++] = (uint8_t)(wtab[s2] & 0xFF); break; }
                                if (nw >= 255) { bad = true; break; }
                                weights[nw++] = (uint8_t)(wtab[s2] & 0xFF);
                                s2 = (wtab[s2] >> 16) + (uint32_t)zd_bread(bb, (wtab[s2] >> 8) & 0xFF);
                                if (bb.off < 0) { if (nw >= 255) { bad = true; break; } weights[nw++] = (uint8_t)(wtab[s1] & 0xFF); break; }
                            }
                            if (bad) { bstat = ZD_CORRUPT; break; }
                            used = 1 + csize;
                        }
                        uint32_t total = 0; bool badw = false;
                        for (uint32_t i = 0; i < nw; i++) { if (weights[i] > ZD_HUF_MAX) badw = true; else if (weights[i]) total += 1u << (weights[i] - 1); }
                        if (badw || total == 0 || nw < 1) { bstat = ZD_CORRUPT; break; }
                        const uint32_t maxbits = (uint32_t)zd_hb(total) + 1;
                        const uint32_t rest = (1u << maxbits) - total;
                        if (maxbits > (uint32_t)ZD_HUF_MAX || rest == 0 || (rest & (rest - 1))) { bstat = ZD_CORRUPT; break; }
                        weights[nw] = (uint8_t)(zd_hb(rest) + 1);
                        if (next_slot >= x.slot_cap) { bstat = ZD_UNSUPPORTED; break; }
                        huf_slot = x.slot_base + next_slot;             // this block's own slot (shared with its FSE tables below)
                        p1_tree = 1; s_cmd[3] = maxbits; s_cmd[4] = nw + 1;
                    } else if (huf_slot == 0xFFFFFFFFu) { bstat = ZD_CORRUPT; break; }
                    if (used > cl) { bstat = ZD_CORRUPT; break; }
                    b.lit_off = pos + used; b.lit_csize = cl - used;
                    pos += comp;
                }
                b.seq_off = pos;                                          // provisional: start of the sequences section
            } while (false);
            s_cmd[0] = p1_tree; s_cmd[5] = bstat;
        }
        __builtin_amdgcn_wave_barrier();
        bstat = s_cmd[5];
        if (s_cmd[0]) {
            // ---- Huffman table: all lanes (same construction as k_zdec)
            const uint32_t nwt = s_cmd[4], mbits = s_cmd[3];
            for (uint32_t sy = lane; sy < nwt; sy += 64) {
                const uint32_t w = weights[sy];
                if (!w) continue;
                uint32_t pos = 0;
                for (uint32_t o = 0; o < nwt; o++) { const uint32_t wo = weights[o]; if (wo && (wo < w || (wo == w && o < sy))) pos += 1u << (wo - 1); }
                const uint32_t n = 1u << (w - 1), cell = sy | ((mbits + 1 - w) << 8);
                for (uint32_t i = 0; i < n; i++) huf_tab[pos + i] = (uint16_t)cell;
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)huf_slot);
            ZTables *T = tabs + sl;
            for (uint32_t i = lane; i < (1u << mbits); i += 64) T->huf[i] = huf_tab[i];
            if (lane == 0) T->hufbits = mbits;
            __builtin_amdgcn_wave_barrier();
        }
        // ---- phase 2 (lane 0 with the wave flushing tables): sequences section
        bool own_slot = s_cmd[0] != 0;                                    // does this block already occupy next_slot?
        if (bstat == ZD_OK && (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane == 0 ? b.type : 0)) == 2) {
            uint32_t nseq = 0, modes = 0, pos = 0, len = 0;
            const uint8_t *p = in + ip + 3;
            if (lane == 0) {
                len = b.size; pos = b.seq_off;
                do {
                    if (pos >= len) { bstat = ZD_CORRUPT; break; }
                    const uint32_t b0 = p[pos++];
                    if (b0 < 128) nseq = b0;
                    else if (b0 < 255) { if (pos >= len) { bstat = ZD_CORRUPT; break; } nseq = ((b0 - 128) << 8) + p[pos++]; }
                    else { if (pos + 2 > len) { bstat = ZD_CORRUPT; break; } nseq = p[pos] + ((uint32_t)p[pos + 1] << 8) + 0x7F00; pos += 2; }
                    if (nseq == 0) { if (pos != len) bstat = ZD_CORRUPT; break; }
                    if (pos >= len) { bstat = ZD_CORRUPT; break; }
                    modes = p[pos++];
                    if (modes & 3) bstat = ZD_CORRUPT;
                } while (false);
                s_cmd[5] = bstat; s_cmd[6] = nseq; s_cmd[7] = modes;
            }
            __builtin_amdgcn_wave_barrier();
            bstat = s_cmd[5]; nseq = s_cmd[6]; modes = s_cmd[7];
            for (int k = 0; k < 3 && bstat == ZD_OK && nseq; k++) {        // LL, OF, ML in stream order
                const uint32_t mode = (modes >> (6 - 2 * k)) & 3;
                if (mode == 3) { if (slot3[k] == 0xFFFFFFFFu) bstat = ZD_CORRUPT; continue; }
                if (lane == 0) {
                    uint32_t st2 = ZD_OK, alog_out = 0;
                    const int16_t *def = k == 0 ? ZD_LL_DEF : (k == 1 ? ZD_OF_DEF : ZD_ML_DEF);
                    const int def_n = k == 0 ? 36 : (k == 1 ? 29 : 53), def_log = k == 1 ? 5 : 6;
                    const int max_sym = k == 0 ? 35 : (k == 1 ? 31 : 52), max_log = k == 1 ? 8 : 9;
                    if (!own_slot && next_slot >= x.slot_cap) st2 = ZD_UNSUPPORTED;
                    else if (mode == 0) {
                        for (int i = 0; i < def_n; i++) norm[i] = def[i];
                        if (!zd_fse_build(fse_tab, norm, def_n, def_log, nexts)) st2 = ZD_CORRUPT;
                        alog_out = (uint32_t)def_log;
                    } else if (mode == 1) {
                        if (pos >= len || p[pos] > max_sym) st2 = ZD_CORRUPT; else { fse_tab[0] = p[pos]; pos++; alog_out = 0; }
                    } else {
                        int nsym, alog;
                        const uint32_t used = zd_fse_desc(p + pos, len - pos, norm, &nsym, &alog, max_sym, max_log);
                        if (!used || !zd_fse_build(fse_tab, norm, nsym, alog, nexts)) st2 = ZD_CORRUPT; else { pos += used; alog_out = (uint32_t)alog; }
                    }
                    s_cmd[5] = st2; s_cmd[3] = alog_out;
                }
                __builtin_amdgcn_wave_barrier();
                bstat = s_cmd[5];
                if (bstat != ZD_OK) break;
                own_slot = true;
                slot3[k] = x.slot_base + next_slot;
                ZTables *T = tabs + slot3[k];
                const uint32_t alog = s_cmd[3];
                for (uint32_t i = lane; i < (1u << alog); i += 64) T->fse[k][i] = fse_tab[i];
                if (lane == 0) T->alog[k] = alog;
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0 && bstat == ZD_OK) {
                b.nseq = nseq;
                if (nseq) {
                    if (pos >= len) bstat = ZD_CORRUPT;
                    else if (seq_pos + nseq > x.seq_cap) bstat = ZD_UNSUPPORTED;
                    else { b.seq_off = pos; b.seq_len = len - pos; }
                }
                s_cmd[5] = bstat;
            }
            __builtin_amdgcn_wave_barrier();
            bstat = s_cmd[5];
        }
        if (own_slot) next_slot++;
        // ---- record the block, enqueue its work
        if (bstat != ZD_OK) { status = bstat; break; }
        if (lane == 0) {
            b.huf_slot = huf_slot; b.slot[0] = slot3[0]; b.slot[1] = slot3[1]; b.slot[2] = slot3[2];
            const uint32_t bi = x.blk_base + nblk;
            blocks[bi] = b;
            if (b.type == 2 && b.ltype >= 2) {
                const uint32_t q = atomicAdd(&work->n_huf, b.streams);
                for (uint32_t sq2 = 0; sq2 < b.streams; sq2++) huf_list[q + sq2] = bi * 4 + sq2;
            }
            if (b.type == 2) { const uint32_t q = atomicAdd(&work->n_seq, 1u); seq_list[q] = bi; }
            s_cmd[1] = (uint32_t)b.pad[0]; s_cmd[2] = b.size; s_cmd[3] = b.type; s_cmd[4] = b.regen; s_cmd[6] = b.nseq;
        }
        __builtin_amdgcn_wave_barrier();
        last = s_cmd[1] != 0;
        ip += 3 + (s_cmd[3] == 1 ? 1 : s_cmd[2]);
        if (s_cmd[3] == 2) { lit_pos += s_cmd[4]; seq_pos += s_cmd[6]; }
        nblk++;
        __builtin_amdgcn_wave_barrier();
        if (ONE) break;
    }
    if (ONE) { if (status != ZD_OK && lane == 0) atomicMax(&frames[f].status, status); return; }
    if (lane == 0) { fx[f].nblk = nblk; frames[f].status = status; }
}

// ------------------------------------------------------------------ k_zparse_a : the header walk of a LARGE frame, one lane per frame of the list
// Per block only what the walk itself needs: the 3-byte block header, the literals section's header (<= 5 bytes: sizes) and the sequences section's
// (count + modes) -- which tables the block DEFINES follows from those, so the slots can be handed out without decoding anything.  Every block gets its
// descriptor with the walk's state in front of it (k_zparse<true> starts from there); a few dependent loads per block: ~10 ms for 2 048 blocks.
__global__ void k_zparse_a(ZFrame *__restrict__ frames, ZFrameX *__restrict__ fx, const uint32_t *__restrict__ big_list, uint32_t nbig, const uint8_t *__restrict__ src,
                           ZBlock *__restrict__ blocks, uint32_t *__restrict__ one_list, uint32_t *__restrict__ one_count) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbig) return;
    const uint32_t f = big_list[t];
    ZFrame fr = frames[f];
    const ZFrameX x = fx[f];
    const uint8_t *in = src + fr.src_off;
    const uint32_t in_len = (uint32_t)fr.src_len;
    const uint64_t cap = fr.dst_len < 0xFFFFFFFFull ? fr.dst_len : 0xFFFFFFFFull;   // (literal positions count in 32 bits: a frame with 4 GiB of literals and more in compressed blocks is left to the one-workgroup kernel)
    const uint64_t fcs_want = fr.dst_len;
    const bool open = (fr.out_len & ZF_OPEN) != 0;
    uint32_t status = fr.status, ip = 0;
    if (status == ZD_OK) {
        if (in_len < 6) status = ZD_CORRUPT;
        else {
            const uint32_t magic = in[0] | (in[1] << 8) | (in[2] << 16) | ((uint32_t)in[3] << 24);
            const uint32_t fhd = in[4];
            const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, dict = fhd & 3;
            if (magic != 0xFD2FB528u || (fhd & 0x08)) status = ZD_CORRUPT;
            else if (dict) status = ZD_UNSUPPORTED;
            else {
                ip = 5 + (single ? 0 : 1);
                const uint32_t fsz = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2u : (fcs_flag == 2 ? 4u : 8u));
                if (ip + fsz > in_len) status = ZD_CORRUPT;
                else {
                    uint64_t fcs = 0;
                    for (uint32_t i = 0; i < fsz; i++) fcs |= (uint64_t)in[ip + i] << (8 * i);
                    if (fsz == 2) fcs += 256;
                    if (fsz && (open ? fcs > fcs_want : fcs != fcs_want)) status = ZD_DSTSIZE;
                    ip += fsz;
                }
            }
        }
    }
    uint32_t nblk = 0, next_slot = 0, huf_slot = 0xFFFFFFFFu, slot3[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t lit_pos = 0; uint64_t seq_pos = 0;
    bool last = status != ZD_OK;
    while (!last) {
        if (nblk >= x.blk_cap) { status = ZD_UNSUPPORTED; break; }
        if (ip + 3 > in_len) { status = ZD_CORRUPT; break; }
        const uint32_t bh = in[ip] | (in[ip + 1] << 8) | ((uint32_t)in[ip + 2] << 16);
        const uint32_t type = (bh >> 1) & 3, size = bh >> 3;
        if (type == 3 || size > (128u << 10)) { status = ZD_CORRUPT; break; }
        if (type == 1 ? ip + 4 > in_len : ip + 3 + size > in_len) { status = ZD_CORRUPT; break; }
        ZBlock b;
        b.body = fr.src_off + ip + 3; b.out_off = 0; b.seq_pos = x.seq_base + seq_pos; b.size = size; b.type = type; b.ltype = 0; b.regen = 0; b.streams = 1; b.lit_off = 0; b.lit_csize = 0;
        b.lit_pos = lit_pos; b.huf_slot = huf_slot; b.slot[0] = slot3[0]; b.slot[1] = slot3[1]; b.slot[2] = slot3[2];
        b.nseq = 0; b.seq_off = 0; b.seq_len = 0; b.frame = f; b.out_len = type < 2 ? size : 0; b.status = 0; b.uses_rep = 0;
        for (int i = 0; i < 7; i++) b.pad[i] = 0;
        b.pad[0] = bh & 1; b.pad[3] = next_slot;
        uint32_t regen = 0, nseq = 0;
        if (type == 2) {
            const uint8_t *p = in + ip + 3; const uint32_t len = size;
            if (len < 1) { status = ZD_CORRUPT; break; }
            const uint32_t ltype = p[0] & 3, sf = (p[0] >> 2) & 3;
            uint32_t comp = 0, hdr;
            if (ltype < 2) {
                if (sf == 0 || sf == 2) { regen = p[0] >> 3; hdr = 1; }
                else if (sf == 1) { if (len < 2) { status = ZD_CORRUPT; break; } regen = (p[0] >> 4) + ((uint32_t)p[1] << 4); hdr = 2; }
                else { if (len < 3) { status = ZD_CORRUPT; break; } regen = (p[0] >> 4) + ((uint32_t)p[1] << 4) + ((uint32_t)p[2] << 12); hdr = 3; }
            } else {
                if (len < 5) { status = ZD_CORRUPT; break; }
                uint64_t v = 0; for (int i = 0; i < 5; i++) v |= (uint64_t)p[i] << (8 * i);
                if (sf == 0) { hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; }
                else if (sf == 1) { hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; }
                else if (sf == 2) { hdr = 4; regen = (v >> 4) & 0x3FFF; comp = (v >> 18) & 0x3FFF; }
                else { hdr = 5; regen = (v >> 4) & 0x3FFFF; comp = (v >> 22) & 0x3FFFF; }
            }
            if (regen > (128u << 10) || (uint64_t)lit_pos + regen > cap) { status = regen > (128u << 10) ? ZD_CORRUPT : (fcs_want > cap ? ZD_UNSUPPORTED : ZD_DSTSIZE); break; }
            uint32_t pos = hdr + (ltype == 0 ? regen : (ltype == 1 ? 1u : comp));
            if (pos >= len) { status = ZD_CORRUPT; break; }                 // (the sequences section holds at least its count)
            const uint32_t b0 = p[pos++];
            if (b0 < 128) nseq = b0;
            else if (b0 < 255) { if (pos >= len) { status = ZD_CORRUPT; break; } nseq = ((b0 - 128) << 8) + p[pos++]; }
            else { if (pos + 2 > len) { status = ZD_CORRUPT; break; } nseq = p[pos] + ((uint32_t)p[pos + 1] << 8) + 0x7F00; pos += 2; }
            uint32_t modes = 0xFF;                                          // (no sequences: nothing is defined)
            if (nseq) { if (pos >= len) { status = ZD_CORRUPT; break; } modes = p[pos]; }
            if (seq_pos + nseq > x.seq_cap) { status = ZD_UNSUPPORTED; break; }
            const bool def_tree = ltype == 2;
            bool own = def_tree;
            for (int k = 0; k < 3; k++) if (nseq && ((modes >> (6 - 2 * k)) & 3) != 3) own = true;
            if (own && next_slot >= x.slot_cap) { status = ZD_UNSUPPORTED; break; }
            if (ltype == 3 && huf_slot == 0xFFFFFFFFu) { status = ZD_CORRUPT; break; }
            if (def_tree) huf_slot = x.slot_base + next_slot;
            for (int k = 0; k < 3; k++) if (nseq && ((modes >> (6 - 2 * k)) & 3) != 3) slot3[k] = x.slot_base + next_slot;
            if (own) next_slot++;
        }
        const uint32_t bi = x.blk_base + nblk;
        blocks[bi] = b;
        one_list[atomicAdd(one_count, 1u)] = bi;
        last = (bh & 1) != 0;
        ip += 3 + (type == 1 ? 1 : size);
        if (type == 2) { lit_pos += regen; seq_pos += nseq; }
        nblk++;
    }
    fx[f].nblk = nblk; frames[f].status = status;
}

// ------------------------------------------------------------------ k_zhuf : one lane per Huffman stream
// The 64 streams of a wave belong to ~16 consecutive blocks, i.e. to a handful of frames: up to four distinct decoding tables
// are copied into LDS once (a lane whose table did not get a place reads it from global memory).
constexpr uint32_t ZH_CACHE = 4;
__global__ __launch_bounds__(64)
void k_zhuf(const uint32_t *__restrict__ huf_list, const ZWork *__restrict__ work, ZBlock *__restrict__ blocks, const ZFrame *__restrict__ frames,
            const ZTables *__restrict__ tabs, const uint8_t *__restrict__ src, uint8_t *__restrict__ lit_scratch) {
    __shared__ uint16_t ctab[ZH_CACHE][1 << ZD_HUF_MAX];
    const uint32_t lane = threadIdx.x, i = blockIdx.x * 64 + lane;
    const bool valid = i < work->n_huf;
    const uint32_t item = valid ? huf_list[i] : 0u, bi = item >> 2, sidx = item & 3;
    ZBlock b; if (valid) b = blocks[bi]; else { b.huf_slot = 0xFFFFFFFFu; b.streams = 1; b.regen = 0; b.lit_csize = 0; b.body = 0; b.lit_off = 0; b.frame = 0; b.lit_pos = 0; }
    // ---- table cache
    uint32_t mine = ZH_CACHE;
    {
        uint64_t rem = __ballot(valid);
        for (uint32_t c = 0; c < ZH_CACHE && rem; c++) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(rem);
            const uint32_t sl = (uint32_t)__shfl((int)b.huf_slot, (int)leader);
            const uint32_t *g = (const uint32_t *)tabs[sl].huf;
            uint32_t *l = (uint32_t *)ctab[c];
            for (uint32_t k = lane; k < (1u << ZD_HUF_MAX) / 2; k += 64) l[k] = g[k];
            const bool same = valid && b.huf_slot == sl;
            if (same) mine = c;
            rem &= ~__ballot(same);
        }
        __syncthreads();
    }
    if (!valid) return;
    const uint8_t *cs = src + b.body + b.lit_off;
    const uint32_t cl = b.lit_csize;
    uint32_t s_off = 0, s_len = cl, o_off = 0, o_len = b.regen;
    bool ok = true;
    if (b.streams == 4) {
        if (cl < 6) ok = false;
        else {
            const uint32_t l1 = cs[0] | (cs[1] << 8), l2 = cs[2] | (cs[3] << 8), l3 = cs[4] | (cs[5] << 8);
            const uint32_t seg = (b.regen + 3) / 4;
            if (6 + l1 + l2 + l3 > cl || seg * 3 > b.regen) ok = false;
            else {
                s_off = 6 + (sidx > 0 ? l1 : 0) + (sidx > 1 ? l2 : 0) + (sidx > 2 ? l3 : 0);
                s_len = sidx == 0 ? l1 : (sidx == 1 ? l2 : (sidx == 2 ? l3 : cl - 6 - l1 - l2 - l3));
                o_off = sidx * seg; o_len = sidx < 3 ? seg : b.regen - 3 * seg;
            }
        }
    }
    if (ok) {
        const ZTables *T = tabs + b.huf_slot;
        const uint32_t mb = T->hufbits;
        const uint8_t *gs = cs + s_off;
        uint8_t *lit = lit_scratch + frames[b.frame].dst_off + b.lit_pos + o_off;
        auto fetch = [&](int64_t j) -> uint64_t { return *(const zd_u64u *)(gs + 8 * j); };
        auto cellof = [&](uint32_t idx) -> uint32_t { return mine < ZH_CACHE ? (uint32_t)ctab[mine][idx] : (uint32_t)T->huf[idx]; };
        ZdWin bw;
        if (!zd_winit(bw, fetch, s_len ? gs[s_len - 1] : 0u, s_len)) ok = false;
        else {
            uint32_t k = 0;
            // head: single bytes up to a 4-byte boundary of the destination, then four symbols per store
            for (; k < o_len && ((uintptr_t)(lit + k) & 3); k++) {
                const uint32_t cell = cellof((uint32_t)zd_wpeek(bw, fetch, mb));
                lit[k] = (uint8_t)cell; bw.off -= cell >> 8;
            }
            // sixteen symbols per store once the destination is 16-byte aligned (fewer stores for the window loads to wait behind)
            for (; ok && k + 4 <= o_len && ((uintptr_t)(lit + k) & 15); k += 4) {
                uint32_t word = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t cell = cellof((uint32_t)zd_wpeek(bw, fetch, mb));
                    word |= (cell & 0xFF) << (8 * q); bw.off -= cell >> 8;
                }
                if (bw.off < 0) ok = false;
                else *(uint32_t *)(lit + k) = word;
            }
            for (; ok && k + 16 <= o_len; k += 16) {
                uint32_t w4[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const uint32_t cell = cellof((uint32_t)zd_wpeek(bw, fetch, mb));
                    w4[q >> 2] |= (cell & 0xFF) << (8 * (q & 3)); bw.off -= cell >> 8;
                }
                if (bw.off < 0) ok = false;
                else *(uint4 *)(lit + k) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            for (; ok && k + 4 <= o_len; k += 4) {
                uint32_t word = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t cell = cellof((uint32_t)zd_wpeek(bw, fetch, mb));
                    word |= (cell & 0xFF) << (8 * q); bw.off -= cell >> 8;
                }
                if (bw.off < 0) ok = false;
                else *(uint32_t *)(lit + k) = word;
            }
            for (; ok && k < o_len; k++) {
                const uint32_t cell = cellof((uint32_t)zd_wpeek(bw, fetch, mb));
                lit[k] = (uint8_t)cell; bw.off -= cell >> 8;
            }
            if (bw.off != 0) ok = false;
        }
    }
    if (!ok) atomicOr(&blocks[bi].status, (uint32_t)ZD_CORRUPT);
}

// ------------------------------------------------------------------ k_zfse : one lane per block with sequences
// The 64 blocks of a wave belong to a few frames: up to eight distinct (LL, OF, ML) table sets are copied into LDS.
constexpr uint32_t ZF_CACHE = 8;
__global__ __launch_bounds__(64)
void k_zfse(const uint32_t *__restrict__ seq_list, const ZWork *__restrict__ work, ZBlock *__restrict__ blocks, const ZTables *__restrict__ tabs,
            const uint8_t *__restrict__ src, uint64_t *__restrict__ seqs) {
    __shared__ uint32_t cfse[ZF_CACHE][3][512];
    __shared__ uint64_t rbuf[64][9];                   // eight records per lane (+1: conflict-free pitch), flushed together
    __shared__ uint32_t t_llb[36], t_mlb[53];
    __shared__ uint8_t  t_lln[36], t_mln[53];
    const uint32_t lane = threadIdx.x;
    if (lane < 36) { t_llb[lane] = ZD_LL_BASE[lane]; t_lln[lane] = ZD_LL_BITS[lane]; }
    if (lane < 53) { t_mlb[lane] = ZD_ML_BASE[lane]; t_mln[lane] = ZD_ML_BITS[lane]; }
    const uint32_t i = blockIdx.x * 64 + lane;
    const bool valid = i < work->n_seq;
    const uint32_t bi = valid ? seq_list[i] : 0u;
    ZBlock b; if (valid) b = blocks[bi]; else { b.nseq = 0; b.slot[0] = b.slot[1] = b.slot[2] = 0xFFFFFFFFu; b.regen = 0; }
    const bool want = valid && b.nseq != 0;
    uint32_t mine = ZF_CACHE;
    {
        uint64_t rem = __ballot(want);
        for (uint32_t c = 0; c < ZF_CACHE && rem; c++) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(rem);
            const uint32_t s0 = (uint32_t)__shfl((int)b.slot[0], (int)leader), s1 = (uint32_t)__shfl((int)b.slot[1], (int)leader), s2 = (uint32_t)__shfl((int)b.slot[2], (int)leader);
            for (uint32_t k = lane; k < 512; k += 64) { cfse[c][0][k] = tabs[s0].fse[0][k]; cfse[c][1][k] = tabs[s1].fse[1][k]; cfse[c][2][k] = tabs[s2].fse[2][k]; }
            const bool same = want && b.slot[0] == s0 && b.slot[1] == s1 && b.slot[2] == s2;
            if (same) mine = c;
            rem &= ~__ballot(same);
        }
    }
    __syncthreads();
    if (!valid) return;
    uint32_t out_len = b.regen, uses_rep = 0;
    bool ok = true;
    if (b.nseq) {
        const uint32_t *gl = tabs[b.slot[0]].fse[0], *go = tabs[b.slot[1]].fse[1], *gm = tabs[b.slot[2]].fse[2];
        const uint32_t al = tabs[b.slot[0]].alog[0], ao = tabs[b.slot[1]].alog[1], am = tabs[b.slot[2]].alog[2];
        auto TL = [&](uint32_t st) -> uint32_t { return mine < ZF_CACHE ? cfse[mine][0][st] : gl[st]; };
        auto TO = [&](uint32_t st) -> uint32_t { return mine < ZF_CACHE ? cfse[mine][1][st] : go[st]; };
        auto TM = [&](uint32_t st) -> uint32_t { return mine < ZF_CACHE ? cfse[mine][2][st] : gm[st]; };
        const uint8_t *gq = src + b.body + b.seq_off;
        auto fetch = [&](int64_t j) -> uint64_t { return *(const zd_u64u *)(gq + 8 * j); };
        ZdWin bw;
        if (!zd_winit(bw, fetch, b.seq_len ? gq[b.seq_len - 1] : 0u, b.seq_len)) ok = false;
        else {
            uint32_t sll = (uint32_t)zd_wread(bw, fetch, al), sof = (uint32_t)zd_wread(bw, fetch, ao), sml = (uint32_t)zd_wread(bw, fetch, am);
            uint64_t *rec = seqs + b.seq_pos;
            uint64_t total = b.regen;
            for (uint32_t k = 0; k < b.nseq; k++) {
                const uint32_t cl = TL(sll & 511), co = TO(sof & 511), cm = TM(sml & 511);
                const uint32_t llc = cl & 0xFF, ofc = co & 0xFF, mlc = cm & 0xFF;
                if (ofc > 31 || mlc > 52 || llc > 35) { ok = false; break; }
                // three window reads per sequence: offset bits | match-length + literal-length bits | the three state updates
                // (a field read earlier sits above the later ones in a combined read)
                const uint64_t ofv = ((uint64_t)1 << ofc) + zd_wread(bw, fetch, ofc);
                const uint32_t nm = t_mln[mlc], nl = t_lln[llc];
                const uint32_t e2 = (uint32_t)zd_wread(bw, fetch, nm + nl);
                const uint32_t ml = t_mlb[mlc] + (e2 >> nl);
                const uint32_t ll = t_llb[llc] + (e2 & ((1u << nl) - 1));
                if (bw.off < 0 || ofv > ZREC_OF_MAX) { ok = false; break; }
                if (k + 1 < b.nseq) {
                    const uint32_t bl = (cl >> 8) & 0xFF, bm = (cm >> 8) & 0xFF, bo = (co >> 8) & 0xFF;
                    const uint32_t e3 = (uint32_t)zd_wread(bw, fetch, bl + bm + bo);
                    sll = (cl >> 16) + (e3 >> (bm + bo));
                    sml = (cm >> 16) + ((e3 >> bo) & ((1u << bm) - 1));
                    sof = (co >> 16) + (e3 & ((1u << bo) - 1));
                    if (bw.off < 0) { ok = false; break; }
                }
                if (ofv <= 3) uses_rep = 1;
                rbuf[lane][k & 7] = zrec_pack(ll, ml, ofv);
                total += ml;
                if ((k & 7) == 7 || k + 1 == b.nseq) {
                    const uint32_t k0 = k & ~7u;
                    for (uint32_t q = 0; q <= (k & 7); q++) rec[k0 + q] = rbuf[lane][q];
                }
            }
            if (ok && bw.off != 0) ok = false;
            if (total > (128u << 10)) ok = false;
            out_len = (uint32_t)total;
        }
    }
    blocks[bi].out_len = out_len; blocks[bi].uses_rep = uses_rep;
    if (!ok) atomicOr(&blocks[bi].status, (uint32_t)ZD_CORRUPT);
}

// ------------------------------------------------------------------ k_zoff : one thread per frame
__global__ void k_zoff(ZFrame *__restrict__ frames, const ZFrameX *__restrict__ fx, uint32_t n, ZBlock *__restrict__ blocks) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    if (frames[f].status) return;
    const ZFrameX x = fx[f];
    uint64_t off = frames[f].dst_off, total = 0; uint32_t st = 0;
    for (uint32_t k = 0; k < x.nblk; k++) {
        ZBlock *b = blocks + x.blk_base + k;
        b->out_off = off + total; total += b->out_len; st |= b->status;
    }
    if (st) frames[f].status = ZD_CORRUPT;
    else if (frames[f].out_len & ZF_OPEN) { if (total > frames[f].dst_len) frames[f].status = ZD_DSTSIZE; else frames[f].dst_len = total; }
    else if (total != frames[f].dst_len) frames[f].status = ZD_DSTSIZE;
}

// the low n (0 .. 8 and more = 8) bytes of v to p: one 8-byte store, or 4 + 2 + 1 (unaligned accesses)
typedef uint32_t __attribute__((aligned(1))) zd_u32u;
typedef uint16_t __attribute__((aligned(1))) zd_u16u;
__device__ __forceinline__ void zx_store_upto8(uint8_t *p, uint64_t v, uint32_t n) {
    if (n >= 8) { *(zd_u64u *)p = v; return; }
    if (n & 4) { *(zd_u32u *)p = (uint32_t)v; p += 4; v >>= 32; }
    if (n & 2) { *(zd_u16u *)p = (uint16_t)v; p += 2; v >>= 16; }
    if (n & 1) *p = (uint8_t)v;
}
// ------------------------------------------------------------------ k_zexec : one wave per frame, blocks in order
// (zexec_blocks: blocks [k0, k1) of a frame on the calling wave; k_zexec_groups runs the execution groups of inflated streams side by side)
__device__ __forceinline__ bool zexec_blocks(const ZFrame &fr, const ZFrameX &x, uint32_t k0, uint32_t k1, const ZBlock *__restrict__ blocks, const uint8_t *__restrict__ src,
                                             const uint8_t *__restrict__ lit_scratch, const uint64_t *__restrict__ seqs, uint8_t *__restrict__ dst, uint32_t lane) {
    // Positions below count in 32 bits from `out`.  For frames below 2 GiB that is the frame's start; further on the base follows the blocks
    // 2 GiB behind (a block is at most 128 KiB, a reference reaches back less than 2^28: both stay in range, and the test `offset > o0 + ll`
    // cannot fail there, as it must not).  Frames of 4 GiB and more: zlib streams decoded by pieces (k_vinflate).
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;
    bool okq = true;
    for (uint32_t k = k0; k < k1 && okq; k++) {
        const ZBlock b = blocks[x.blk_base + k];
        const uint64_t bpos = b.out_off - fr.dst_off, rebase = bpos > (1ull << 31) ? bpos - (1ull << 31) : 0;
        uint8_t *out = dst + fr.dst_off + rebase;
        const uint64_t cap = fr.dst_len - rebase;
        uint32_t op = (uint32_t)(bpos - rebase);
        const uint8_t *body = src + b.body;
        if (b.type < 2) {
            const uint32_t rle = body[0];
            for (uint32_t i = lane; i < b.size; i += 64) out[op + i] = b.type == 0 ? body[i] : (uint8_t)rle;
            continue;
        }
        const uint8_t *lit_raw = body + b.lit_off;
        const uint8_t *lit_dec = lit_scratch + fr.dst_off + ((uint64_t)b.lit_pos | ((uint64_t)b.pad[2] << 32));
        auto LIT = [&](uint32_t i) -> uint8_t { return b.ltype == 0 ? lit_raw[i] : (b.ltype == 1 ? lit_raw[0] : lit_dec[i]); };
        const uint64_t *rec = seqs + b.seq_pos;
        uint32_t litpos = 0;
        for (uint32_t base = 0; base < b.nseq && okq; base += 64) {
            const uint32_t nb = b.nseq - base < 64 ? b.nseq - base : 64u;
            const bool act = lane < nb;
            const uint64_t sv = act ? rec[base + lane] : 0;
            const uint32_t ll = zrec_ll(sv), ml = zrec_ml(sv), ofv = zrec_of(sv);
            // ---- offsets: without repeat codes in the batch every offset is its own value - 3 and the history is simply
            // the last three offsets; otherwise lane order has to be walked
            uint32_t offset = ofv - 3;
            const uint64_t repm = __ballot(act && ofv <= 3);
            if (!repm) {
                if (nb >= 3) { rep2 = (uint32_t)__shfl((int)offset, (int)nb - 3); rep1 = (uint32_t)__shfl((int)offset, (int)nb - 2); rep0 = (uint32_t)__shfl((int)offset, (int)nb - 1); }
                else if (nb == 2) { rep2 = rep0; rep1 = (uint32_t)__shfl((int)offset, 0); rep0 = (uint32_t)__shfl((int)offset, 1); }
                else { rep2 = rep1; rep1 = rep0; rep0 = (uint32_t)__shfl((int)offset, 0); }
            } else {
                // Only the repeat codes are walked (libzstd's frames hold some in nearly every batch; walking all 64 lanes was most of the kernel's
                // instructions for them): the history in front of a repeat code = the history behind the previous one, pushed down by the up to three
                // plain offsets in between -- read from their lanes (uniform lane numbers: v_readlane).
                auto lane_of = [&](uint32_t v, uint32_t j) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)j)); };
                auto advance = [&](uint32_t cur, uint32_t j) {             // plain offsets of lanes [cur, j) enter the history
                    const uint32_t c = j - cur;
                    if (c >= 3) { rep0 = lane_of(offset, j - 1); rep1 = lane_of(offset, j - 2); rep2 = lane_of(offset, j - 3); }
                    else if (c == 2) { rep2 = rep0; rep0 = lane_of(offset, j - 1); rep1 = lane_of(offset, j - 2); }
                    else if (c == 1) { rep2 = rep1; rep1 = rep0; rep0 = lane_of(offset, j - 1); }
                };
                uint32_t cur = 0;
                for (uint64_t m = repm; m; m &= m - 1) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(m);
                    advance(cur, j);
                    const uint32_t vj = lane_of(ofv, j), lj = lane_of(ll, j);
                    const uint32_t idx = vj - 1 + (lj == 0 ? 1u : 0u);
                    uint32_t o;
                    if (idx == 0) o = rep0;
                    else { o = idx == 1 ? rep1 : (idx == 2 ? rep2 : rep0 - 1); if (idx > 1) rep2 = rep1; rep1 = rep0; rep0 = o; }
                    if (lane == j) offset = o;
                    cur = j + 1;
                }
                advance(cur, nb);
            }
            uint32_t incl = ll + ml, lincl = ll;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t1 = (uint32_t)__shfl_up((int)incl, d), t2 = (uint32_t)__shfl_up((int)lincl, d);
                if (lane >= (uint32_t)d) { incl += t1; lincl += t2; }
            }
            const uint32_t o0 = op + incl - (ll + ml), l0 = litpos + lincl - ll;
            const uint32_t T = (uint32_t)__shfl((int)incl, 63), TL = (uint32_t)__shfl((int)lincl, 63);
            const bool bad = act && (l0 + ll > b.regen || (uint64_t)o0 + ll + ml > cap || offset == 0 || offset > o0 + ll);
            if (__ballot(bad)) { okq = false; break; }
            {
                const uint32_t ls = ll < 32 ? ll : 32u;
                if (b.ltype == 1) { for (uint32_t i = 0; __ballot(i < ls); i++) if (i < ls) out[o0 + i] = lit_raw[0]; }
                else {
                    // All the loads, then the stores: up to four 8-byte pieces (unaligned 64-bit accesses), the last, partial one read whole where the
                    // block's literals reach that far (byte by byte at their very end) and stored as 4 + 2 + 1 bytes.  A load waited for per piece and per
                    // tail byte, as before, was what a batch of 64 sequences took its ~10 us for: one wave sees the memory latency undiluted.
                    const uint8_t *ls_src = (b.ltype == 0 ? lit_raw : lit_dec) + l0;
                    uint64_t q[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) {
                        q[k] = 0;
                        if (8 * k < ls) {
                            if (l0 + 8 * k + 8 <= b.regen) q[k] = *(const zd_u64u *)(ls_src + 8 * k);
                            else for (uint32_t t = 0; t < 8 && 8 * k + t < ls; t++) q[k] |= (uint64_t)ls_src[8 * k + t] << (8 * t);
                        }
                    }
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) zx_store_upto8(out + o0 + 8 * k, q[k], ls > 8 * k ? ls - 8 * k : 0u);
                }
                uint64_t longm = __ballot(ll > 32);
                while (longm) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(longm); longm &= longm - 1;
                    const uint32_t oj = (uint32_t)__shfl((int)o0, (int)j), lj = (uint32_t)__shfl((int)l0, (int)j), nj = (uint32_t)__shfl((int)ll, (int)j);
                    for (uint32_t i = 32 + lane; i < nj; i += 64) out[oj + i] = LIT(lj + i);
                }
            }
            const uint32_t dstp = o0 + ll, m0 = dstp - offset;
            const uint32_t src_end = m0 + (ml < offset ? ml : offset);
            bool pending = act && ml != 0;
            for (;;) {
                const uint64_t pm = __ballot(pending);
                if (!pm) break;
                __threadfence_block();
                const uint32_t frontier = (uint32_t)__shfl((int)dstp, (int)__builtin_ctzll(pm));
                const bool ready = pending && src_end <= frontier;
                const bool shortr = ready && ml <= 32;
                {
                    // source and destination apart (offset >= length: most matches): all the loads, then the stores, as for the literals (bytes behind the
                    // source's end may be in the making: they are read and dropped; the read stays inside the frame)
                    const bool apart = shortr && offset >= ml;
                    if (__ballot(apart)) {
                        uint64_t q[4];
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) {
                            q[k] = 0;
                            if (apart && 8 * k < ml) {
                                if ((uint64_t)m0 + 8 * k + 8 <= cap) q[k] = *(const zd_u64u *)(out + m0 + 8 * k);
                                else for (uint32_t t = 0; t < 8 && 8 * k + t < ml; t++) q[k] |= (uint64_t)out[m0 + 8 * k + t] << (8 * t);
                            }
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) zx_store_upto8(out + dstp + 8 * k, q[k], apart && ml > 8 * k ? ml - 8 * k : 0u);
                    }
                    // overlapping: distance >= 8: eight bytes per step; closer matches copy byte by byte (they read what they just wrote)
                    const bool lap = shortr && !apart, wide = lap && offset >= 8;
                    uint32_t i = 0;
                    for (; __ballot(wide && i + 8 <= ml); i += 8) if (wide && i + 8 <= ml) *(zd_u64u *)(out + dstp + i) = *(const zd_u64u *)(out + m0 + i);
                    i = wide ? (ml & ~7u) : 0u;
                    for (; __ballot(lap && i < ml); i++) if (lap && i < ml) out[dstp + i] = out[m0 + i];
                }
                uint64_t longm = __ballot(ready && ml > 32);
                while (longm) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(longm); longm &= longm - 1;
                    const uint32_t dj = (uint32_t)__shfl((int)dstp, (int)j), mj = (uint32_t)__shfl((int)m0, (int)j);
                    const uint32_t nj = (uint32_t)__shfl((int)ml, (int)j), fj = (uint32_t)__shfl((int)offset, (int)j);
                    if (fj >= nj) { for (uint32_t i = lane; i < nj; i += 64) out[dj + i] = out[mj + i]; }
                    else { for (uint32_t i = lane; i < nj; i += 64) out[dj + i] = out[mj + i % fj]; }
                }
                pending = pending && !ready;
            }
            op += T; litpos += TL;
        }
        if (okq) {
            const uint32_t rest = b.regen - litpos;
            if ((uint64_t)op + rest > cap) okq = false;
            else for (uint32_t i = lane; i < rest; i += 64) out[op + i] = LIT(litpos + i);
        }
        __threadfence_block();
    }
    return okq;
}

__global__ __launch_bounds__(64)
void k_zexec(ZFrame *__restrict__ frames, const ZFrameX *__restrict__ fx, const ZBlock *__restrict__ blocks, const uint8_t *__restrict__ src,
             const uint8_t *__restrict__ lit_scratch, const uint64_t *__restrict__ seqs, uint8_t *__restrict__ dst) {
    const uint32_t lane = threadIdx.x, f = blockIdx.x;
    const ZFrame fr = frames[f];
    if (fr.status) return;
    const ZFrameX x = fx[f];
    if (x.pad) return;                                                  // a large frame: executed in parallel (k_zexec_par.hip)
    const bool okq = zexec_blocks(fr, x, 0, x.nblk, blocks, src, lit_scratch, seqs, dst, lane);
    if (!okq && lane == 0) frames[f].status = ZD_CORRUPT;
}

// One wave per PIECE slot of the inflated streams (the list k_vinflate ran on): the wave of a piece that starts an execution group (pad[5], k_vfin;
// block 0 always does) executes the group's blocks in order, the others leave at once.  A stream the wave-per-stream walk took has one block.
__global__ __launch_bounds__(64)
void k_zexec_groups(ZFrame *__restrict__ frames, const ZFrameX *__restrict__ fx, const ZBlock *__restrict__ blocks, const uint2 *__restrict__ pieces, uint32_t npieces,
                    const uint8_t *__restrict__ src, const uint8_t *__restrict__ lit_scratch, const uint64_t *__restrict__ seqs, uint8_t *__restrict__ dst) {
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= npieces) return;
    const uint2 pc = pieces[blockIdx.x];                               // (frame, piece)
    const uint32_t f = pc.x, j = pc.y;
    const ZFrame fr = frames[f];
    if (fr.status) return;
    const ZFrameX x = fx[f];
    if (x.pad) return;                                                  // a large foreign stream: its chunks are executed in parallel (k_zexec_par.hip)
    if (j >= x.nblk || (j && !blocks[x.blk_base + j].pad[5])) return;
    uint32_t k1 = j + 1;
    while (k1 < x.nblk && !blocks[x.blk_base + k1].pad[5]) k1++;
    const bool okq = zexec_blocks(fr, x, j, k1, blocks, src, lit_scratch, seqs, dst, lane);
    if (!okq && lane == 0) frames[f].status = ZD_CORRUPT;
}

void launch_zparse(ZFrame *frames, ZFrameX *fx, uint32_t n, const uint8_t *src, ZBlock *blocks, ZTables *tabs, uint32_t *huf_list, uint32_t *seq_list,
                   void *work, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_zparse<false>, dim3(n), dim3(64), 0, st, frames, fx, src, blocks, tabs, huf_list, seq_list, (ZWork *)work, (const uint32_t *)nullptr);
}
// large frames (ZFrameX::pad set, listed in big_list): the header walk; one_list receives their blocks, work[2] their number
void launch_zparse_big_a(ZFrame *frames, ZFrameX *fx, const uint32_t *big_list, uint32_t nbig, const uint8_t *src, ZBlock *blocks, uint32_t *one_list, void *work, hipStream_t st) {
    if (nbig) hipLaunchKernelGGL(k_zparse_a, dim3((nbig + 63) / 64), dim3(64), 0, st, frames, fx, big_list, nbig, src, blocks, one_list, &((ZWork *)work)->pad0);
}
// ... and the blocks' tables, one wave per block
void launch_zparse_big_b(ZFrame *frames, ZFrameX *fx, uint32_t nblocks, const uint8_t *src, ZBlock *blocks, ZTables *tabs, uint32_t *huf_list, uint32_t *seq_list,
                         void *work, const uint32_t *one_list, hipStream_t st) {
    if (nblocks) hipLaunchKernelGGL(k_zparse<true>, dim3(nblocks), dim3(64), 0, st, frames, fx, src, blocks, tabs, huf_list, seq_list, (ZWork *)work, one_list);
}
void launch_zstreams(uint32_t n_huf, uint32_t n_seq, const uint32_t *huf_list, const uint32_t *seq_list, const void *work, ZBlock *blocks,
                     const ZFrame *frames, const ZTables *tabs, const uint8_t *src, uint8_t *lit_scratch, uint64_t *seqs, hipStream_t st) {
    if (n_huf) hipLaunchKernelGGL(k_zhuf, dim3((n_huf + 63) / 64), dim3(64), 0, st, huf_list, (const ZWork *)work, blocks, frames, tabs, src, lit_scratch);
    if (n_seq) hipLaunchKernelGGL(k_zfse, dim3((n_seq + 63) / 64), dim3(64), 0, st, seq_list, (const ZWork *)work, blocks, tabs, src, seqs);
}
void launch_zexec(ZFrame *frames, const ZFrameX *fx, uint32_t n, ZBlock *blocks, const uint8_t *src, const uint8_t *lit_scratch,
                  const uint64_t *seqs, uint8_t *dst, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_zoff, dim3((n + 63) / 64), dim3(64), 0, st, frames, fx, n, blocks);
    hipLaunchKernelGGL(k_zexec, dim3(n), dim3(64), 0, st, frames, fx, blocks, src, lit_scratch, seqs, dst);
}
void launch_zexec_groups(ZFrame *frames, const ZFrameX *fx, uint32_t n, ZBlock *blocks, const void *pieces, uint32_t npieces, const uint8_t *src, const uint8_t *lit_scratch,
                         const uint64_t *seqs, uint8_t *dst, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_zoff, dim3((n + 63) / 64), dim3(64), 0, st, frames, fx, n, blocks);
    if (npieces) hipLaunchKernelGGL(k_zexec_groups, dim3(npieces), dim3(64), 0, st, frames, fx, (const ZBlock *)blocks, (const uint2 *)pieces, npieces, src, lit_scratch, seqs, dst);
}

// ------------------------------------------------------------------ k_zscan : one thread per entry
// An entry's payload is one or more concatenated frames (zstd-rs' Decoder reads them all).  Frames carry no content size here,
// so the split of the entry's raw size over its frames follows this repository's encoder: every frame but the last holds
// SEG_SIZE bytes (a single-frame entry -- what the reference writes -- holds all of it); k_zdec verifies the sizes.
// walk one frame starting at p[ip]: header, blocks, optional checksum; returns its end or 0 when it is malformed / truncated
__device__ uint64_t zscan_frame_end(const uint8_t *p, uint64_t ip, uint64_t len) {
    uint64_t q = ip;
    if (q + 6 > len) return 0;
    const uint32_t magic = p[q] | (p[q + 1] << 8) | (p[q + 2] << 16) | ((uint32_t)p[q + 3] << 24);
    if (magic != 0xFD2FB528u) return 0;
    const uint32_t fhd = p[q + 4];
    const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, dict = fhd & 3;
    q += 5 + (single ? 0 : 1) + (dict == 0 ? 0 : (dict == 1 ? 1 : (dict == 2 ? 2 : 4)));
    q += fcs_flag == 0 ? single : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    for (;;) {
        if (q + 3 > len) return 0;
        const uint32_t bh = p[q] | (p[q + 1] << 8) | ((uint32_t)p[q + 2] << 16);
        const uint32_t type = (bh >> 1) & 3, size = bh >> 3;
        q += 3 + (type == 1 ? 1 : size);
        if (type == 3 || q > len) return 0;
        if (bh & 1) break;
    }
    if ((fhd >> 2) & 1) q += 4;
    return q > len ? 0 : q;
}

__global__ void k_zscan(const ZEntry *__restrict__ ents, uint32_t n, const uint8_t *__restrict__ src, ZFrame *__restrict__ frames, ZFrameX *__restrict__ fx) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const ZEntry en = ents[e];
    const uint8_t *p = src + en.src_off;
    const uint64_t len = en.src_len;
    const uint32_t nfr = en.n_frames;
    // ONE frame that holds the whole entry, however large (what the reference writes: libzstd streaming, one frame per entry): it gets all of
    // raw_len and -- its compressed bytes below 4 GiB -- all the block descriptors, table slots and sequence records planned for the entry's frames (their regions are
    // contiguous), so that its blocks are decoded side by side like those of many small frames (k_zparse .. k_zexec: a 64 MiB frame 10.7 MiB/s on
    // the one-workgroup kernel).  A frame of 4 GiB of COMPRESSED bytes and more, or one that still does not fit (k_zparse finds out), goes to that kernel.  The other
    // frame slots planned for the entry are void.
    if (nfr > 1 && zscan_frame_end(p, 0, len) == len) {
        ZFrame fr; fr.src_off = en.src_off; fr.dst_off = en.dst_off; fr.src_len = len; fr.out_len = en.open ? ZF_OPEN : 0u;
        fr.dst_len = en.raw_len; fr.status = 2u;
        if (len <= 0xFFFFFFFFull && fx) {                            // (content of any size: the parse counts output in 64 bits, the executor works in windows; a compressed size beyond 32 bits stays with the one-workgroup kernel)
            ZFrameX x = fx[en.first_frame];
            uint64_t nb = 0, ns = 0, nq = 0;
            for (uint32_t g = 0; g < nfr; g++) { const ZFrameX y = fx[en.first_frame + g]; nb += y.blk_cap; ns += y.slot_cap; nq += y.seq_cap; }
            x.blk_cap = (uint32_t)(nb < 0x7FFFFFFFull ? nb : 0x7FFFFFFFull); x.slot_cap = (uint32_t)(ns < 0x7FFFFFFFull ? ns : 0x7FFFFFFFull);
            x.seq_cap = (uint32_t)(nq < 0x7FFFFFFFull ? nq : 0x7FFFFFFFull);
            fx[en.first_frame] = x;
            fr.status = 0u;
        }
        frames[en.first_frame] = fr;
        fr.src_len = 0; fr.dst_len = 0; fr.status = 4;              // ZD_VOID
        for (uint32_t g = 1; g < nfr; g++) frames[en.first_frame + g] = fr;
        return;
    }
    uint64_t ip = 0;
    for (uint32_t f = 0; f < nfr; f++) {
        ZFrame fr; fr.src_off = en.src_off + ip; fr.dst_off = en.dst_off + (uint64_t)f * SEG_SIZE; fr.status = 0; fr.out_len = 0;
        const uint64_t done = (uint64_t)f * SEG_SIZE;
        fr.dst_len = (uint32_t)(en.raw_len - done < SEG_SIZE || f + 1 == nfr ? (en.raw_len > done ? en.raw_len - done : 0) : SEG_SIZE);
        if (en.raw_len - done > 0xFFFFFFFFull && f + 1 == nfr) fr.status = 2;
        const uint64_t q = zscan_frame_end(p, ip, len);
        if (!q) { fr.status = 1; fr.src_len = 0; frames[en.first_frame + f] = fr; for (uint32_t g = f + 1; g < nfr; g++) { fr.src_off = 0; frames[en.first_frame + g] = fr; } return; }
        fr.src_len = (uint32_t)(q - ip);
        if (en.open && f + 1 == nfr) fr.out_len = ZF_OPEN;             // the last frame of a stream of unknown size
        frames[en.first_frame + f] = fr;
        ip = q;
    }
    if (ip != len && nfr) frames[en.first_frame + nfr - 1].status = 1;      // bytes left over behind the last frame
}

// number of frames of every entry's payload (0: malformed) -- the planning step for streams that carry no size (solid)
__global__ void k_zcount(const ZEntry *__restrict__ ents, uint32_t n, const uint8_t *__restrict__ src, uint32_t *__restrict__ counts) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const ZEntry en = ents[e];
    const uint8_t *p = src + en.src_off;
    uint64_t ip = 0; uint32_t cnt = 0;
    while (ip < en.src_len) {
        if (ip + 8 <= en.src_len && ((p[ip] | (p[ip + 1] << 8) | (p[ip + 2] << 16) | ((uint32_t)p[ip + 3] << 24)) & 0xFFFFFFF0u) == 0x184D2A50u) {   // a skippable frame: not counted
            const uint64_t sz = (uint64_t)p[ip + 4] | ((uint64_t)p[ip + 5] << 8) | ((uint64_t)p[ip + 6] << 16) | ((uint64_t)p[ip + 7] << 24);
            if (ip + 8 + sz > en.src_len) { cnt = 0; break; }
            ip += 8 + sz;
            continue;
        }
        const uint64_t q = zscan_frame_end(p, ip, en.src_len);
        if (!q) { cnt = 0; break; }
        ip = q; cnt++;
        if (cnt == 0x7FFFFFFFu) { cnt = 0; break; }
    }
    counts[e] = cnt;
}
void launch_zcount(const ZEntry *ents, uint32_t n, const uint8_t *src, uint32_t *counts, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_zcount, dim3((n + 63) / 64), dim3(64), 0, st, ents, n, src, counts);
}

// ------------------------------------------------------------------ k_zlist : the frames of ONE payload, whatever their sizes
// zstd::stream::read::Decoder (decompress_reader, lib/src/entry/read.rs:171-190) reads ANY concatenation of frames, skippable frames (magic 0x184D2A5?, RFC 8878
// 3.1.2: 4 bytes of length, then that many bytes to ignore) included.  k_zscan only knows the two shapes this library and the reference write -- one frame per
// entry, or this library's 1 MiB grid --; a payload it cannot place goes through this walk: one thread lists up to `cap` zstd frames from `ip0` on (offset,
// length, Frame_Content_Size or ~0 where the header carries none), skipping the skippable ones; hdr[0] = frames listed, hdr[1] = where the walk stopped
// (== len: the payload is through), hdr[2] = 1 when the bytes at the stop are neither kind of frame / truncated.
struct ZListItem { uint64_t off, len, fcs; };
__global__ void k_zlist(const uint8_t *__restrict__ src, uint64_t base, uint64_t len, uint64_t ip0, ZListItem *__restrict__ items, uint32_t cap, uint64_t *__restrict__ hdr) {
    if (blockIdx.x || threadIdx.x) return;
    const uint8_t *p = src + base;
    uint64_t ip = ip0; uint32_t cnt = 0; uint64_t bad = 0;
    while (ip < len && cnt < cap) {
        if (ip + 4 > len) { bad = 1; break; }
        const uint32_t magic = p[ip] | (p[ip + 1] << 8) | (p[ip + 2] << 16) | ((uint32_t)p[ip + 3] << 24);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (ip + 8 > len) { bad = 1; break; }
            const uint64_t sz = (uint64_t)p[ip + 4] | ((uint64_t)p[ip + 5] << 8) | ((uint64_t)p[ip + 6] << 16) | ((uint64_t)p[ip + 7] << 24);
            if (ip + 8 + sz > len) { bad = 1; break; }
            ip += 8 + sz;
            continue;
        }
        const uint64_t q = zscan_frame_end(p, ip, len);
        if (!q) { bad = 1; break; }
        const uint32_t fhd = p[ip + 4];
        const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, dict = fhd & 3;
        const uint64_t fp = ip + 5 + (single ? 0 : 1) + (dict == 0 ? 0 : (dict == 1 ? 1 : (dict == 2 ? 2 : 4)));
        const uint32_t fb = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2u : (fcs_flag == 2 ? 4u : 8u));
        uint64_t fcs = ~0ull;
        if (fb) { fcs = 0; for (uint32_t i = 0; i < fb; i++) fcs |= (uint64_t)p[fp + i] << (8 * i); if (fb == 2) fcs += 256; }
        items[cnt].off = base + ip; items[cnt].len = q - ip; items[cnt].fcs = fcs;
        cnt++; ip = q;
    }
    hdr[0] = cnt; hdr[1] = ip; hdr[2] = bad;
}
void launch_zlist(const uint8_t *src, uint64_t base, uint64_t len, uint64_t ip0, void *items, uint32_t cap, uint64_t *hdr, hipStream_t st) {
    hipLaunchKernelGGL(k_zlist, dim3(1), dim3(64), 0, st, src, base, len, ip0, (ZListItem *)items, cap, hdr);
}

void launch_zscan(const ZEntry *ents, uint32_t n, const uint8_t *src, ZFrame *frames, ZFrameX *fx, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_zscan, dim3((n + 63) / 64), dim3(64), 0, st, ents, n, src, frames, fx);
}

// ------------------------------------------------------------------ k_zxxh : Content_Checksum (RFC 8878 3.1.1: the low 32 bits of XXH64(content, seed 0))
// Frames written with a checksum -- this library and the reference's writer set none (lib/src/entry/write.rs:260-262: the encoder's defaults), other
// zstd writers may -- are verified after decoding: a frame whose decoded bytes do not hash to the stored value is corrupt, as zstd::stream::read::Decoder
// reports it (decompress_reader, lib/src/entry/read.rs:171-190).  XXH64 runs four independent accumulator chains over 32-byte stripes: four lanes per
// frame, sixteen frames per wave; frames without the flag cost one byte read.
__device__ __forceinline__ uint64_t xxh_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xxh_rd64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i); return v; }
__global__ __launch_bounds__(64)
void k_zxxh(ZFrame *__restrict__ frames, uint32_t n, const uint8_t *__restrict__ src, const uint8_t *__restrict__ dst) {
    constexpr uint64_t P1 = 11400714785074694791ull, P2 = 14029467366897019727ull, P3 = 1609587929392839161ull, P4 = 9650029242287828579ull, P5 = 2870177450012600261ull;
    const uint32_t lane = threadIdx.x, f = blockIdx.x * 16 + (lane >> 2), k = lane & 3;
    bool act = f < n;
    ZFrame fr; fr.status = 1; fr.src_len = 0; fr.dst_len = 0; fr.src_off = 0; fr.dst_off = 0;
    if (act) fr = frames[f];
    act = act && fr.status == 0 && fr.src_len >= 10 && ((src[fr.src_off + 4] >> 2) & 1);
    const uint8_t *p = dst + fr.dst_off;
    const uint64_t len = act ? fr.dst_len : 0;
    uint64_t acc = k == 0 ? P1 + P2 : (k == 1 ? P2 : (k == 2 ? 0ull : 0ull - P1));
    const uint64_t nstripe = len >> 5;
    for (uint64_t s = 0; s < nstripe; s++) {
        uint64_t w;
        const uint8_t *q = p + 32 * s + 8 * k;
        if ((((uintptr_t)q) & 7) == 0) w = *(const uint64_t *)q; else w = xxh_rd64(q);
        acc = xxh_rotl(acc + w * P2, 31) * P1;
    }
    // the four accumulators meet in lane 0 of the frame's group
    const uint32_t g0 = lane & ~3u;
    uint64_t v[4];
    for (int j = 0; j < 4; j++) { const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)acc, (int)(g0 + j)), hi = (uint32_t)__shfl((int)(uint32_t)(acc >> 32), (int)(g0 + j)); v[j] = ((uint64_t)hi << 32) | lo; }
    if (!act || k != 0) return;
    uint64_t h;
    if (len >= 32) {
        h = xxh_rotl(v[0], 1) + xxh_rotl(v[1], 7) + xxh_rotl(v[2], 12) + xxh_rotl(v[3], 18);
        for (int j = 0; j < 4; j++) { const uint64_t r = xxh_rotl(v[j] * P2, 31) * P1; h = (h ^ r) * P1 + P4; }
    } else h = P5;
    h += len;
    const uint8_t *q = p + (nstripe << 5), *e = p + len;
    for (; q + 8 <= e; q += 8) { const uint64_t r = xxh_rotl(xxh_rd64(q) * P2, 31) * P1; h = xxh_rotl(h ^ r, 27) * P1 + P4; }
    if (q + 4 <= e) { const uint64_t w = (uint64_t)q[0] | ((uint64_t)q[1] << 8) | ((uint64_t)q[2] << 16) | ((uint64_t)q[3] << 24); h = xxh_rotl(h ^ (w * P1), 23) * P2 + P3; q += 4; }
    for (; q < e; q++) h = xxh_rotl(h ^ ((uint64_t)*q * P5), 11) * P1;
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    const uint8_t *c4 = src + fr.src_off + fr.src_len - 4;
    const uint32_t stored = (uint32_t)c4[0] | ((uint32_t)c4[1] << 8) | ((uint32_t)c4[2] << 16) | ((uint32_t)c4[3] << 24);
    if (stored != (uint32_t)h) frames[f].status = 1;                   // corrupt: the content does not match its checksum
}
void launch_zxxh(ZFrame *frames, uint32_t n, const uint8_t *src, const uint8_t *dst, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_zxxh, dim3((n + 15) / 16), dim3(64), 0, st, frames, n, src, dst);
}

// dbg (option "zdec_dbg"): diagnostics: 1 skip execution, 2 skip sequences, 4 skip Huffman streams, 8 small re-base distances
void launch_zdec(ZFrame *frames, uint32_t n, const uint8_t *src, uint8_t *dst, uint8_t *lit_scratch, uint32_t dbg, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_zdec, dim3(n), dim3(ZD_THREADS), 0, st, frames, src, dst, lit_scratch, dbg);
}

} // namespace pna
