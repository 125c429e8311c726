// k_zexec_par.hip -- execution of ONE LARGE zstd frame in parallel (gfx950).  The reference writes one frame per entry whatever its size
// (lib/src/entry/write.rs:260-262: one zstd::stream::write::Encoder per entry; read back by decompress_reader, lib/src/entry/read.rs:171-190, from
// cli/src/command/extract.rs:594-640), so a large file of a reference-written archive is one frame of thousands of blocks.  Its entropy decoding is
// parallel over the blocks already (k_zhuf / k_zfse, k_zdec.hip); what stayed serial was the EXECUTION of the sequences: one wave walked the frame's
// blocks in order (k_zexec: ~10 us per 64 sequences, 114 MiB/s) -- a match may copy what the match before it wrote, so blocks cannot simply run side by
// side (a block's first matches reach into the end of the block before).  Here the copy problem is solved by pointer jumping instead:
//   k_zrep_block  one wave per block: repeat-offset codes resolved inside the block from a SYMBOLIC start history (tags U0..U2 and U0-1..U2-1 flow
//                 through the history's permutations like numbers; after three plain offsets a block's history is its own), end history per block
//   k_zrep_scan   one lane: the blocks' start histories from the end histories in order (a few instructions per block)
//   k_zx_expand   one workgroup per block: for every output byte of the block ONE 32-bit word -- FLAG | byte for a literal, the frame position of
//                 its source byte for a match byte -- written position-parallel (the sequence of a position by binary search in the chunk's prefix sums)
//   k_zx_jump     rounds of word[p] = word[word[p]] over all unresolved words until every word holds a byte (chains halve per round: log2 of the longest
//                 copy chain; racing reads see an older or a newer ancestor, both valid)
//   k_zx_emit     the bytes
// Scratch: 4 bytes per output byte of a WINDOW.  A frame is executed in windows of whole blocks, at most 1 GiB of output each (the host cuts them: pna_decode.cpp zx_windows, option zexec_win_mib), one after the other:
// a word counts from its window's first byte (31 bits), and a match byte whose source lies in FRONT of the window -- in a window that has been emitted --
// is that byte, read from the output.  So a frame of any size takes this path (round 4, second half: until then frames of 2 GiB and more kept the
// one-workgroup executor, ~11 MiB/s).  Offsets beyond 2^28 - 16 and histories that chain "rep0 - 1" twice through a block start keep the serial executor.
// Integer / byte work, no MFMA.
#include <hip/hip_runtime.h>
#include "pna_dev.h"

namespace pna {

constexpr uint32_t ZX_FLAG = 0x80000000u;                  // word = FLAG | byte: resolved
constexpr uint32_t ZX_TAG0 = 0xFFFFFF0u;                   // record offset values from here on are tags: TAG0 + k = start history slot k, TAG0 + 3 + k = that - 1
constexpr uint32_t ZX_SYM0 = 0xFFFFFFF0u;                  // the same tags as 32-bit history values during the walk
enum { ZX_OK = 0, ZX_FALLBACK = 2, ZX_CORRUPT = 3 };      // (frame status: the larger code wins)

struct ZxFrame {               // one large frame: what the kernels below share
    uint64_t dst_off;          // where the frame's content starts in dst
    uint64_t dst_len;          // bytes of content (k_zoff has checked them against the blocks' sum)
    uint32_t blk_base, nblk;   // its blocks in the block array
    uint32_t status;           // out: ZX_*
    uint32_t unresolved;       // k_zx_jump: words still pointing somewhere after the latest round
};

__device__ __forceinline__ uint32_t zx_rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)l)); }
// offset - 1 of a history value: a number stays a number (0 = corrupt, the caller checks), a tag Uk becomes the tag Uk - 1, that one has no successor
__device__ __forceinline__ uint32_t zx_minus1(uint32_t v, bool &unsupported) {
    if (v < ZX_SYM0) return v - 1;
    const uint32_t t = v - ZX_SYM0;
    if (t < 3) return ZX_SYM0 + 3 + t;
    unsupported = true; return v;
}

// ------------------------------------------------------------------ k_zrep_block : one wave per block
// Rewrites the block's records: the offset value of every sequence becomes offset + 3 (never a repeat code) or, where the offset comes from the history the
// block started with, a tag.  rep_end[3 b ..]: the history behind the block (numbers or symbolic values).
__global__ __launch_bounds__(64)
void k_zrep_block(ZxFrame *__restrict__ zf, const ZBlock *__restrict__ blocks, uint64_t *__restrict__ seqs, uint32_t *__restrict__ rep_end) {
    const uint32_t lane = threadIdx.x, k = blockIdx.x;
    const ZBlock b = blocks[zf->blk_base + k];
    uint32_t rep0 = ZX_SYM0, rep1 = ZX_SYM0 + 1, rep2 = ZX_SYM0 + 2;
    bool unsupported = false, corrupt = false;
    if (b.type == 2 && b.nseq) {
        uint64_t *rec = seqs + b.seq_pos;
        for (uint32_t base = 0; base < b.nseq; base += 64) {
            const uint32_t nb = b.nseq - base < 64 ? b.nseq - base : 64u;
            const bool act = lane < nb;
            const uint64_t sv = act ? rec[base + lane] : 0;
            const uint32_t ll = zrec_ll(sv), ofv = zrec_of(sv);
            uint32_t offset = ofv - 3;                                      // (a number for plain offsets; repeat codes are set below)
            if (act && ofv > 3 && ofv >= ZX_TAG0) unsupported = true;       // an offset as large as the tags: the serial executor takes the frame
            const uint64_t repm = __ballot(act && ofv <= 3);
            auto advance = [&](uint32_t cur, uint32_t j) {                  // plain offsets of lanes [cur, j) enter the history
                const uint32_t c = j - cur;
                if (c >= 3) { rep0 = zx_rdlane(offset, j - 1); rep1 = zx_rdlane(offset, j - 2); rep2 = zx_rdlane(offset, j - 3); }
                else if (c == 2) { rep2 = rep0; rep0 = zx_rdlane(offset, j - 1); rep1 = zx_rdlane(offset, j - 2); }
                else if (c == 1) { rep2 = rep1; rep1 = rep0; rep0 = zx_rdlane(offset, j - 1); }
            };
            uint32_t cur = 0;
            for (uint64_t m = repm; m; m &= m - 1) {
                const uint32_t j = (uint32_t)__builtin_ctzll(m);
                advance(cur, j);
                const uint32_t vj = zx_rdlane(ofv, j), lj = zx_rdlane(ll, j);
                if (vj == 0) { corrupt = true; cur = j + 1; continue; }
                const uint32_t idx = vj - 1 + (lj == 0 ? 1u : 0u);
                uint32_t o;
                if (idx == 0) o = rep0;
                else { o = idx == 1 ? rep1 : (idx == 2 ? rep2 : zx_minus1(rep0, unsupported)); if (idx > 1) rep2 = rep1; rep1 = rep0; rep0 = o; }
                if (o == 0) corrupt = true;
                if (lane == j) offset = o;
                cur = j + 1;
            }
            advance(cur, nb);
            if (act && repm) {                                              // (batches without repeat codes keep their records)
                const uint32_t nv = offset >= ZX_SYM0 ? ZX_TAG0 + (offset - ZX_SYM0) : offset + 3;
                rec[base + lane] = zrec_pack(ll, zrec_ml(sv), nv);
            }
        }
    }
    if (__ballot(unsupported) && lane == 0) atomicMax(&zf->status, (uint32_t)ZX_FALLBACK);
    if (__ballot(corrupt) && lane == 0) atomicMax(&zf->status, (uint32_t)ZX_CORRUPT);
    if (lane == 0) { rep_end[3 * k] = rep0; rep_end[3 * k + 1] = rep1; rep_end[3 * k + 2] = rep2; }
}

// ------------------------------------------------------------------ k_zrep_scan : one lane
// rep_start[3 b ..] = the history block b starts with (numbers): the frame starts with 1, 4, 8 (RFC 8878 3.1.1.5)
__global__ void k_zrep_scan(ZxFrame *__restrict__ zf, const uint32_t *__restrict__ rep_end, uint32_t *__restrict__ rep_start) {
    if (threadIdx.x || blockIdx.x) return;
    uint32_t s[3] = {1, 4, 8};
    bool bad = false;
    for (uint32_t k = 0; k < zf->nblk; k++) {
        rep_start[3 * k] = s[0]; rep_start[3 * k + 1] = s[1]; rep_start[3 * k + 2] = s[2];
        uint32_t e[3];
        for (int i = 0; i < 3; i++) {
            const uint32_t v = rep_end[3 * k + i];
            if (v < ZX_SYM0) e[i] = v;
            else { const uint32_t t = v - ZX_SYM0; e[i] = t < 3 ? s[t] : s[t - 3] - 1; if (t >= 3 && s[t - 3] <= 1) bad = true; }
        }
        s[0] = e[0]; s[1] = e[1]; s[2] = e[2];
    }
    if (bad) atomicMax(&zf->status, (uint32_t)ZX_CORRUPT);
}

// ------------------------------------------------------------------ k_zx_expand : one workgroup per block
constexpr uint32_t ZXE_THREADS = 256;
__global__ __launch_bounds__(ZXE_THREADS)
void k_zx_expand(ZxFrame *__restrict__ zf, const ZBlock *__restrict__ blocks, const uint8_t *__restrict__ src, const uint8_t *__restrict__ lit_scratch,
                 const uint64_t *__restrict__ seqs, const uint32_t *__restrict__ rep_start, uint32_t *__restrict__ words,
                 const uint8_t *__restrict__ dst, uint32_t wb0, uint64_t win_off) {                 // the window: blocks from wb0 on (one workgroup each), its first byte frame-relative
    __shared__ uint32_t s_end[ZXE_THREADS];        // inclusive end (block-relative output position) of each sequence of the chunk
    __shared__ uint32_t s_lit[ZXE_THREADS];        // literal index behind each sequence's literals (inclusive prefix of ll)
    __shared__ uint32_t s_ll[ZXE_THREADS], s_off[ZXE_THREADS];
    __shared__ uint32_t s_scan[ZXE_THREADS];
    __shared__ uint32_t s_bad, s_status;
    const uint32_t tid = threadIdx.x, k = wb0 + blockIdx.x;
    // (other workgroups raise the status with atomicMax while this one runs: ONE thread reads it, so that every thread of the workgroup takes the same way past the barriers below)
    if (tid == 0) s_status = zf->status;
    __syncthreads();
    if (s_status) return;
    const ZBlock b = blocks[zf->blk_base + k];
    const uint64_t bpos64 = b.out_off - zf->dst_off;                          // block's first byte, frame-relative
    const uint32_t bpos = (uint32_t)(bpos64 - win_off);                       // ... and window-relative (< 2^30: the host's windows)
    const uint8_t *done = dst + zf->dst_off;                                  // the frame's output: everything in front of the window is there
    uint32_t *w = words + bpos;
    const uint8_t *body = src + b.body;
    if (tid == 0) s_bad = 0;
    if (b.type < 2) {                                                           // raw / RLE block: every byte a literal
        const uint32_t rle = body[0];
        for (uint32_t i = tid; i < b.size; i += ZXE_THREADS) w[i] = ZX_FLAG | (b.type == 0 ? body[i] : rle);
        return;
    }
    const uint8_t *lit_raw = body + b.lit_off;
    const uint8_t *lit_dec = lit_scratch + zf->dst_off + ((uint64_t)b.lit_pos | ((uint64_t)b.pad[2] << 32));
    auto LIT = [&](uint32_t i) -> uint32_t { return b.ltype == 0 ? lit_raw[i] : (b.ltype == 1 ? lit_raw[0] : lit_dec[i]); };
    const uint64_t *rec = seqs + b.seq_pos;
    const uint32_t r0 = rep_start[3 * k], r1 = rep_start[3 * k + 1], r2 = rep_start[3 * k + 2];
    uint32_t op = 0, litpos = 0;                                                // running output position (block-relative) and literal index (uniform)
    __syncthreads();
    for (uint32_t base = 0; base < b.nseq; base += ZXE_THREADS) {
        const uint32_t nb = b.nseq - base < ZXE_THREADS ? b.nseq - base : ZXE_THREADS;
        const bool act = tid < nb;
        const uint64_t sv = act ? rec[base + tid] : 0;
        const uint32_t ll = zrec_ll(sv), ml = zrec_ml(sv), ofv = zrec_of(sv);
        uint32_t offset;
        if (ofv >= ZX_TAG0) { const uint32_t t = ofv - ZX_TAG0, v = t % 3 == 0 ? r0 : (t % 3 == 1 ? r1 : r2); offset = t < 3 ? v : v - 1; }
        else offset = ofv - 3;
        // workgroup scan of (ll + ml, ll) packed: the sums of a chunk stay below 2^16 x 256... no: a block holds <= 128 KiB, both sums fit 18 bits -> two scans
        uint32_t incl = ll + ml, lincl = ll;
        s_scan[tid] = incl; __syncthreads();
        for (uint32_t d = 1; d < ZXE_THREADS; d <<= 1) { const uint32_t t = tid >= d ? s_scan[tid - d] : 0; __syncthreads(); s_scan[tid] += t; __syncthreads(); }
        incl = s_scan[tid]; __syncthreads();
        s_scan[tid] = lincl; __syncthreads();
        for (uint32_t d = 1; d < ZXE_THREADS; d <<= 1) { const uint32_t t = tid >= d ? s_scan[tid - d] : 0; __syncthreads(); s_scan[tid] += t; __syncthreads(); }
        lincl = s_scan[tid];
        const uint32_t o0 = op + incl - (ll + ml), l0 = litpos + lincl - ll;    // where the sequence's literals go / come from
        // (offset > bpos + o0 + ll: a match may reach back to the frame's first byte, not further)
        if (act && (offset == 0 || (uint64_t)offset > bpos64 + o0 + ll || l0 + ll > b.regen || (uint64_t)o0 + ll + ml > b.out_len)) s_bad = 1;
        s_end[tid] = act ? op + incl : 0xFFFFFFFFu; s_lit[tid] = l0; s_ll[tid] = ll; s_off[tid] = offset;
        __syncthreads();
        const uint32_t span = s_end[nb - 1] - op;                               // output bytes of this chunk
        if (s_bad) break;
        for (uint32_t p = tid; p < span; p += ZXE_THREADS) {
            const uint32_t pos = op + p;
            uint32_t lo = 0, hi = nb - 1;                                       // first sequence whose end lies behind pos
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_end[mid] > pos) hi = mid; else lo = mid + 1; }
            const uint32_t start = lo ? s_end[lo - 1] : op, rel = pos - start;
            uint32_t v;
            if (rel < s_ll[lo]) v = ZX_FLAG | LIT(s_lit[lo] + rel);
            else if (s_off[lo] <= bpos + pos) v = bpos + pos - s_off[lo];      // the source inside the window: a pointer
            else v = ZX_FLAG | done[bpos64 + pos - s_off[lo]];                  // ... in front of it: the byte (checked above: not in front of the frame)
            w[pos] = v;
        }
        const uint32_t T = s_end[nb - 1] - op, TL = s_lit[nb - 1] + s_ll[nb - 1] - litpos;
        __syncthreads();
        op += T; litpos += TL;
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) atomicMax(&zf->status, (uint32_t)ZX_CORRUPT); return; }
    const uint32_t rest = b.regen - litpos;                                     // the block's last literals
    if ((uint64_t)op + rest != b.out_len) { if (tid == 0) atomicMax(&zf->status, (uint32_t)ZX_CORRUPT); return; }
    for (uint32_t i = tid; i < rest; i += ZXE_THREADS) w[op + i] = ZX_FLAG | LIT(litpos + i);
}

// ------------------------------------------------------------------ k_zx_jump : one round of pointer jumping over the frame's words
__global__ __launch_bounds__(256)
void k_zx_jump(ZxFrame *__restrict__ zf, uint32_t *__restrict__ words, uint64_t n) {
    uint32_t left = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (uint64_t)gridDim.x * 1024) {
        uint32_t v[4];
        if (i + 4 <= n) { const uint4 q = *(const uint4 *)(words + i); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else { for (int t = 0; t < 4; t++) v[t] = i + t < n ? words[i + t] : ZX_FLAG; }
        if ((v[0] & v[1] & v[2] & v[3]) & ZX_FLAG) continue;                  // all four resolved
        uint32_t u[4];
#pragma unroll
        for (int t = 0; t < 4; t++) u[t] = (v[t] & ZX_FLAG) ? v[t] : words[v[t]];      // (a source always lies in front of its position: inside the frame)
#pragma unroll
        for (int t = 0; t < 4; t++) if (!(v[t] & ZX_FLAG)) { if (i + t < n) words[i + t] = u[t]; left += (u[t] & ZX_FLAG) ? 0u : 1u; }
    }
    for (int d = 32; d; d >>= 1) left += (uint32_t)__shfl_xor((int)left, d);
    if ((threadIdx.x & 63) == 0 && left) atomicAdd(&zf->unresolved, left);
}

// ------------------------------------------------------------------ k_zx_emit : the bytes
__global__ __launch_bounds__(256)
void k_zx_emit(const ZxFrame *__restrict__ zf, const uint32_t *__restrict__ words, uint8_t *__restrict__ dst, uint64_t n, uint64_t win_off) {
    uint8_t *out = dst + zf->dst_off + win_off;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (uint64_t)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const uint4 q = *(const uint4 *)(words + i);
            const uint32_t pk = (q.x & 0xFF) | ((q.y & 0xFF) << 8) | ((q.z & 0xFF) << 16) | ((q.w & 0xFF) << 24);
            if ((((uintptr_t)(out + i)) & 3) == 0) *(uint32_t *)(out + i) = pk;
            else { out[i] = (uint8_t)pk; out[i + 1] = (uint8_t)(pk >> 8); out[i + 2] = (uint8_t)(pk >> 16); out[i + 3] = (uint8_t)(pk >> 24); }
        } else for (int t = 0; t < 4 && i + t < n; t++) out[i + t] = (uint8_t)words[i + t];
    }
}

// host side: the steps for ONE frame; `zf` (device) holds its description, `words` >= 4 * min(dst_len, ZX_WIN_MAX + a block) + 64 bytes of scratch, rep_scratch 24 * nblk bytes;
// win_blk[0 .. nwin]: the windows' first blocks (win_blk[nwin] = nblk), win_off[0 .. nwin]: their first bytes, frame-relative (win_off[nwin] = dst_len).
// Returns after the last emit has been queued; *rounds = jump rounds run (all windows).  The unresolved counter is read back between rounds (a few microseconds each).
int launch_zexec_par(ZxFrame *zf, const ZxFrame &h, const ZBlock *blocks, const uint8_t *src, const uint8_t *lit_scratch, uint64_t *seqs, uint32_t *rep_scratch,
                     uint32_t *words, uint8_t *dst, uint32_t *status_out, uint32_t *rounds_out, hipStream_t st,
                     uint32_t nwin, const uint32_t *win_blk, const uint64_t *win_off) {
    uint32_t *rep_end = rep_scratch, *rep_start = rep_scratch + 3 * (size_t)h.nblk;
    hipLaunchKernelGGL(k_zrep_block, dim3(h.nblk), dim3(64), 0, st, zf, blocks, seqs, rep_end);
    hipLaunchKernelGGL(k_zrep_scan, dim3(1), dim3(64), 0, st, zf, (const uint32_t *)rep_end, rep_start);
    ZxFrame cur; cur.status = 0;
    uint32_t rounds = 0;
    for (uint32_t wi = 0; wi < nwin && cur.status == 0; wi++) {
        const uint32_t b0 = win_blk[wi], b1 = win_blk[wi + 1];
        const uint64_t a0 = win_off[wi], wlen = win_off[wi + 1] - a0;
        if (b1 == b0) continue;
        hipLaunchKernelGGL(k_zx_expand, dim3(b1 - b0), dim3(ZXE_THREADS), 0, st, zf, blocks, src, lit_scratch, (const uint64_t *)seqs, (const uint32_t *)rep_start, words,
                           (const uint8_t *)dst, b0, a0);
        if (hipMemcpyAsync(&cur, zf, sizeof cur, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
        if (cur.status) break;
        const uint32_t wgs = (uint32_t)((wlen / 4096 + 1) < 16384 ? (wlen / 4096 + 1) : 16384);
        for (uint32_t r = 0;; r++) {
            if (r >= 64) { cur.status = ZX_CORRUPT; break; }                              // (2^64 positions: cannot happen for a well-formed chain)
            if (hipMemsetAsync(&zf->unresolved, 0, 4, st) != hipSuccess) return -1;
            hipLaunchKernelGGL(k_zx_jump, dim3(wgs), dim3(256), 0, st, zf, words, wlen);
            if (hipMemcpyAsync(&cur, zf, sizeof cur, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
            rounds++;
            if (cur.unresolved == 0) break;
        }
        if (cur.status == 0) hipLaunchKernelGGL(k_zx_emit, dim3(wgs), dim3(256), 0, st, (const ZxFrame *)zf, (const uint32_t *)words, dst, wlen, a0);
    }
    if (status_out) *status_out = cur.status;
    if (rounds_out) *rounds_out = rounds;
    return 0;
}

} // namespace pna
