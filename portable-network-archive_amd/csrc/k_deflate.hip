// k_deflate.hip -- zlib/deflate emitter on top of the k_lz sequences (gfx950).
//   k_dstats  one workgroup per segment: lit/len (286) + distance (30) histograms over all the segment's blocks, then
//             thread 0: Huffman lengths (<= 15), canonical bit-reversed codes, dynamic-block table description.
//   k_adler   one workgroup per block: Adler-32 halves of the block's input (combined per entry in k_dfinal).
//   k_dblock  one workgroup per block, token-parallel inside k_lz's 2 KiB tiles: header + literals + (length, distance)
//             pairs + EOB (+ the sync-flush header bits) placed by two packed prefix scans per tile, staged in LDS.
//   k_dplan   one thread per segment: dynamic vs stored per block, sizes.
//   k_dwrite  one workgroup per block: payload (or stored blocks) + sync flush into the packed output.
//   k_dfinal  one thread per entry: 78 9C header, Adler-32 trailer (combine over blocks), empty entries.
// Replaces miniz_oxide behind flate2::write::ZlibEncoder (lib/src/entry/write.rs:257-259).  Integer/bit work only.
#include <hip/hip_runtime.h>
#include "pna_dev.h"

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ v4u ld16u(const uint8_t *p) { v4u v; __builtin_memcpy(&v, p, 16); return v; }   // 16 bytes at any byte address

namespace pna {

__constant__ uint16_t D_LEN_BASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
__constant__ uint8_t  D_LEN_EXTRA[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__constant__ uint8_t  D_CL_ORDER[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
constexpr uint32_t ADLER_P = 65521;

__device__ __forceinline__ uint32_t dhb(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
// length 3..258 -> (code index 0..28, extra bits, extra value); distance 1..32768 -> (code 0..29, extra bits, value)
__device__ __forceinline__ void len_sym(uint32_t l, uint32_t &c, uint32_t &eb, uint32_t &ev) {
    const uint32_t d = l - 3;
    if (d < 8) { c = d; eb = 0; ev = 0; }
    else if (d == 255) { c = 28; eb = 0; ev = 0; }
    else { eb = dhb(d) - 2; c = 4 + 4 * eb + ((d >> eb) & 3); ev = d & ((1u << eb) - 1); }
}
__device__ __forceinline__ void dist_sym(uint32_t dist, uint32_t &c, uint32_t &eb, uint32_t &ev) {
    const uint32_t dd = dist - 1;
    if (dd < 4) { c = dd; eb = 0; ev = 0; }
    else { const uint32_t h = dhb(dd); eb = h - 1; c = 2 * h + ((dd >> eb) & 1); ev = dd & ((1u << eb) - 1); }
}

struct DBitW { uint8_t *p; uint32_t pos; uint64_t acc; uint32_t nb; };
__device__ __forceinline__ void dw_add(DBitW &w, uint32_t v, uint32_t n) {
    if (!n) return;
    w.acc |= (uint64_t)(v & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u))) << w.nb; w.nb += n;
    while (w.nb >= 8) { w.p[w.pos++] = (uint8_t)w.acc; w.acc >>= 8; w.nb -= 8; }
}

// Code construction of one segment on T threads: T = 256, a workgroup per segment, or T = 64, ONE WAVE per segment -- the form for batches of many small
// entries, where four waves per entry spent most of their instructions and three quarters of their time in barriers next to the one-lane sections
// (10^6 x 4 KiB: 14.8 ms of a 56 ms step).  Same procedure, same picks, same bytes for either T.
//
// code lengths <= maxlen (two-queue Huffman + Kraft repair; identical procedure to the zstd literal code): the stable rank sort by count and the leaf
// depths are parallel, the two-queue merge (n - 1 dependent steps) and the rare Kraft repair stay on thread 0.  Every thread must call it; returns the
// number of used symbols.  wt: [580] words, 16-byte aligned -- the sort keys of the symbols live in its upper part (wt + 288: nsym + 4 words) until
// the sorted weights are in its lower part; the merge then overwrites them with the inner nodes.
__device__ __forceinline__ uint32_t d_rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <uint32_t T>
__device__ int d_build_lens(const uint32_t *count, int nsym, int maxlen, uint8_t *lens, uint16_t *order, uint32_t *wt, uint16_t *parent,
                            uint32_t *sh /* [2] scratch */, uint32_t tid) {
    uint32_t *keys = wt + 288;
    if (tid == 0) { sh[0] = 0; sh[1] = 0; }
    for (int s = (int)tid; s < nsym + 4; s += (int)T) keys[s] = 0xFFFFFFFFu;
    __syncthreads();
    // the symbols that occur (a 4 KiB entry uses ~70 of the 286), as keys count << 9 | symbol in no particular order: a symbol's place in the order
    // by (count, symbol) is the number of smaller keys (counts are < 2^21: a segment has at most 2^20 literals)
    for (int s = (int)tid; s < nsym; s += (int)T) {
        lens[s] = 0;
        const uint32_t c = count[s];
        if (c) keys[atomicAdd(&sh[0], 1u)] = (c << 9) | (uint32_t)s;
    }
    __syncthreads();
    const int n = (int)sh[0];
    for (int k = (int)tid; k < n; k += (int)T) {
        const uint32_t key = keys[k];
        uint32_t rank = 0;
        for (int j = 0; j < n; j += 4) {                                  // the list is padded with 0xFFFFFFFF
            const uint4 v = *(const uint4 *)(keys + j);
            rank += (v.x < key ? 1u : 0u) + (v.y < key ? 1u : 0u) + (v.z < key ? 1u : 0u) + (v.w < key ? 1u : 0u);
        }
        order[rank] = (uint16_t)(key & 511u); wt[rank] = key >> 9;
    }
    __syncthreads();
    if (n == 0) return 0;
    if (n == 1) { if (tid == 0) lens[order[0]] = 1; __syncthreads(); return 1; }
    const int nn = 2 * n - 1;
    if (tid < 64) {
        // two-queue merge, n - 1 dependent steps, executed by the first wave IN SCALAR REGISTERS: every value of the loop is wave-uniform (what comes
        // from LDS through readfirstlane), so compares, selects and counters are SALU work and only the LDS traffic -- the queues' next elements, the
        // new node's weight, two parent links -- goes through the vector unit.  Both queues keep two heads in scalar registers and a third element in
        // flight in a vector register (requested when the one before it moved up, read a take later), so no step waits for an LDS round trip; a new
        // node enters whichever of the three places of the internal queue it belongs to directly.  Weights are < 2^31, INF marks an exhausted / not
        // yet filled place.  Same picks, same order as `if (lq < n && (iq >= m || wt[lq] <= wt[iq])) leaf else internal`.
        constexpr uint32_t INF = 0xFFFFFFFFu;
        int lq = 0, iq = n, m = n;
        uint32_t l0 = d_rfl(wt[0]), l1 = d_rfl(wt[1]), i0 = INF, i1 = INF;
        uint32_t l2v = n > 2 ? wt[2] : INF, i2v = INF;
        auto take = [&](uint32_t &w) -> int {
            if (l0 != INF && l0 <= i0) {
                w = l0; const int a = lq++; l0 = l1; l1 = d_rfl(l2v); l2v = (lq + 2 < n) ? wt[lq + 2] : INF; return a;
            }
            w = i0; const int a = iq++; i0 = i1; i1 = d_rfl(i2v); i2v = (iq + 2 < m) ? wt[iq + 2] : INF; return a;
        };
        while (m < nn) {
            uint32_t wa, wb;
            const int a = take(wa), b = take(wb);
            const uint32_t sum = wa + wb;
            if (tid == 0) { wt[m] = sum; parent[a] = (uint16_t)m; parent[b] = (uint16_t)m; }
            if (iq == m) i0 = sum; else if (iq + 1 == m) i1 = sum; else if (iq + 2 == m) i2v = sum;   // the new node's place among the three heads
            m++;
        }
    }
    __syncthreads();
    for (int i = (int)tid; i < n; i += (int)T) {
        int d = 0, q = i;
        while (q != nn - 1) { q = parent[q]; d++; }
        if (d > maxlen) { d = maxlen; sh[1] = 1; }
        lens[order[i]] = (uint8_t)d;
    }
    __syncthreads();
    if (sh[1] && tid == 0) {
        int K = 0;
        for (int i = 0; i < n; i++) K += 1 << (maxlen - lens[order[i]]);
        int debt = K - (1 << maxlen);
        while (debt > 0) {
            int pick = -1, bl = 0;
            for (int i = 0; i < n; i++) { int l = lens[order[i]]; if (l < maxlen && l > bl) { bl = l; pick = i; } }
            lens[order[pick]]++; debt -= 1 << (maxlen - 1 - bl);
        }
        while (debt < 0) {
            int pick = -1, bl = 99, slack = -debt;
            for (int i = n - 1; i >= 0; i--) { int l = lens[order[i]]; if (l > 1 && (1 << (maxlen - l)) <= slack && l < bl) { bl = l; pick = i; } }
            if (pick < 0) break;
            lens[order[pick]]--; debt += 1 << (maxlen - bl);
        }
    }
    __syncthreads();
    return n;
}
// canonical codes, bit-reversed; out[s] = code | len << 16 (out may be global memory).  Symbol s sits on thread s mod T of round s / T:
// code of s = first code of its length + number of lower-numbered symbols of the same length.  That rank comes from one ballot per code
// length THAT OCCURS in the wave's 64 symbols -- lanes below in the wave, plus the counts of the earlier waves / rounds from LDS -- instead of every
// symbol looping over all others.  cntw: 9 x 16 words of scratch (a slot of 16 per wave and round: at most 8).
template <uint32_t T>
__device__ void d_assign(const uint8_t *lens, int nsym, uint32_t *out, uint32_t *first /* [16] scratch */, uint32_t *cntw, uint32_t tid) {
    constexpr uint32_t NW = T / 64, MAXR = (288 + T - 1) / T;
    static_assert(NW * MAXR <= 8, "cntw holds eight slots");
    const uint32_t lane = tid & 63, wv = tid >> 6;
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    const uint32_t nr = ((uint32_t)nsym + T - 1) / T, nslot = nr * NW;
    for (uint32_t i = tid; i < nslot * 16; i += T) cntw[i] = 0;
    __syncthreads();
    uint32_t myl[MAXR], myrank[MAXR];
#pragma unroll
    for (uint32_t r = 0; r < MAXR; r++) {
        myl[r] = 0; myrank[r] = 0;
        if (r < nr) {
            const int sy = (int)(r * T + tid);
            const uint32_t l = sy < nsym ? (uint32_t)lens[sy] : 0u;
            uint32_t rk = 0;
            uint64_t rem = __ballot(l != 0);
            while (rem) {
                const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)l, (int)__builtin_ctzll(rem));
                const uint64_t m = __ballot(l == L);
                if (l == L) rk = (uint32_t)__popcll(m & lt);
                if (lane == 0) cntw[(r * NW + wv) * 16 + L] = (uint32_t)__popcll(m);
                rem &= ~m;
            }
            myl[r] = l; myrank[r] = rk;
        }
    }
    __syncthreads();
    if (tid >= 1 && tid < 16) { uint32_t c = 0; for (uint32_t k = 0; k < nslot; k++) c += cntw[k * 16 + tid]; cntw[8 * 16 + tid] = c; }   // symbols per length
    __syncthreads();
    if (tid == 0) {
        uint32_t code = 0; first[0] = 0;
        for (int b = 1; b <= 15; b++) { code = (code + (b > 1 ? cntw[8 * 16 + b - 1] : 0u)) << 1; first[b] = code; }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < MAXR; r++) {
        const int sy = (int)(r * T + tid);
        if (r < nr && sy < nsym) {
            const uint32_t l = myl[r];
            uint32_t v = 0;
            if (l) {
                uint32_t c = first[l] + myrank[r];
                for (uint32_t k = 0; k < r * NW + wv; k++) c += cntw[k * 16 + l];
                v = (__builtin_bitreverse32(c) >> (32 - l)) | (l << 16);
            }
            out[sy] = v;
        }
    }
    __syncthreads();
}

template <uint32_t T>
__global__ __launch_bounds__(T)
void k_dstats(const SegDesc *__restrict__ segs, const uint64_t *__restrict__ seqs, const uint8_t *__restrict__ lits,
              const BlkInfo *__restrict__ blk, DeflTables *__restrict__ tabs) {
    constexpr uint32_t NW = T / 64, HL = T == 256 ? 8 : 2;              // waves; copies of the literal histogram (32 lanes share one)
    constexpr uint32_t RND = (320 + T - 1) / T, NSLOT = RND * NW;       // rounds of T over the <= 316 code lengths of the table description
    // LDS by phase (the wave-per-segment form lives on occupancy: 6.7 KiB instead of 13): region 1 holds the histograms, then the scratch of the code
    // builds (sorted weights + inner nodes + sort keys, parent links, symbol order); region 2 the symbol counts, then the table description's arrays
    constexpr uint32_t HC = T == 256 ? 4 : 1;                           // copies of the length / distance histograms
    constexpr uint32_t A_BYTES = HL * 1024 + 2 * HC * 128, C_BYTES = 580 * 4 + 576 * 2 + 288 * 2;
    __shared__ __attribute__((aligned(16))) uint8_t r1[A_BYTES > C_BYTES ? A_BYTES : C_BYTES];
    __shared__ __attribute__((aligned(16))) uint32_t r2[344];
    uint32_t (*h_lit)[256] = (uint32_t (*)[256])r1;
    uint32_t *h_len = (uint32_t *)(r1 + HL * 1024), *h_dist = h_len + HC * 32;
    uint32_t *wt = (uint32_t *)r1; uint16_t *parent = (uint16_t *)(r1 + 580 * 4), *order = parent + 576;
    uint32_t *llc = r2, *dc = r2 + 288;
    uint8_t *seq = (uint8_t *)r2, *sym = seq + 320, *ext = sym + 320; uint32_t *hbuf = r2 + 240;
    __shared__ uint8_t ll_len[288], d_len[32], cl_len[20];
    __shared__ uint32_t clc[19], cl_code[19];
    __shared__ uint32_t first_s[16], sh2[2], hsh[3];
    __shared__ uint32_t wsum[8], wsum2[8], cntw_s[9 * 16];
    const uint32_t tid = threadIdx.x;
    const SegDesc sd = segs[blockIdx.x];
    DeflTables *T_ = tabs + blockIdx.x;
    const uint32_t nblk = seg_nblk(sd);
    for (uint32_t i = tid; i < HL * 256; i += T) (&h_lit[0][0])[i] = 0;
    for (uint32_t i = tid; i < 2 * HC * 32; i += T) h_len[i] = 0;       // both histograms, all copies
    __syncthreads();
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t g = sd.blk_base + b;
        const uint32_t nlit = blk[g].nlit, nseq = blk[g].nseq;
        const uint8_t *bl = lits + ((size_t)g << sd.blk_log);
        uint32_t *hl = h_lit[tid & (HL - 1)];
        const uint32_t n16 = nlit >> 4;
        for (uint32_t i = tid; i < n16; i += T) {
            uint4 v = ((const uint4 *)bl)[i];
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&hl[w[k] & 0xFF], 1u); atomicAdd(&hl[(w[k] >> 8) & 0xFF], 1u);
                atomicAdd(&hl[(w[k] >> 16) & 0xFF], 1u); atomicAdd(&hl[w[k] >> 24], 1u);
            }
        }
        for (uint32_t i = (n16 << 4) + tid; i < nlit; i += T) atomicAdd(&hl[bl[i]], 1u);
        const uint64_t *bs = seqs + (size_t)g * seq_cap_of(sd.blk_log);
        for (uint32_t i = tid; i < nseq; i += T) {
            const uint64_t s = bs[i];
            uint32_t c, eb, ev;
            len_sym(seq_ml(s), c, eb, ev); atomicAdd(&h_len[(tid & (HC - 1)) * 32 + c], 1u);
            dist_sym(seq_off(s), c, eb, ev); atomicAdd(&h_dist[(tid & (HC - 1)) * 32 + c], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < 256; i += T) { uint32_t c = 0; for (uint32_t k = 0; k < HL; k++) c += h_lit[k][i]; llc[i] = c; }
    if (tid < 32) {
        uint32_t cl = 0, cd = 0;
        for (uint32_t k = 0; k < HC; k++) { cl += tid >= 1 && tid <= 29 ? h_len[k * 32 + tid - 1] : 0u; cd += tid < 30 ? h_dist[k * 32 + tid] : 0u; }
        llc[256 + tid] = tid == 0 ? nblk : cl;
        dc[tid] = cd;
    }
    __syncthreads();
    (void)d_build_lens<T>(llc, 286, 15, ll_len, order, wt, parent, sh2, tid);
    (void)d_build_lens<T>(dc, 30, 15, d_len, order, wt, parent, sh2, tid);
    d_assign<T>(ll_len, 286, T_->ll_code, first_s, cntw_s, tid);       // straight to the tables in global memory
    d_assign<T>(d_len, 30, T_->d_code, first_s, cntw_s, tid);
    if (tid < 2) { T_->ll_code[286 + tid] = 0; T_->d_code[30 + tid] = 0; }
    // table description: run-length tokens of the code lengths, their 19-symbol code, the header bits -- all of it on all threads
    // (for 4 KiB entries this used to be one lane walking ~300 lengths and ~150 tokens: the largest per-entry cost)
    const uint32_t lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { hsh[0] = 257; hsh[1] = 1; }
    if (tid < 19) clc[tid] = 0;
    for (uint32_t i = tid; i < 104; i += T) hbuf[i] = 0;
    __syncthreads();
    for (uint32_t s2 = 257 + tid; s2 < 286; s2 += T) if (ll_len[s2]) atomicMax(&hsh[0], s2 + 1);
    if (tid >= 1 && tid < 30 && d_len[tid]) atomicMax(&hsh[1], tid + 1);
    __syncthreads();
    const uint32_t nll = hsh[0], nd = hsh[1], n = nll + nd;
    for (uint32_t i = tid; i < n; i += T) seq[i] = i < nll ? ll_len[i] : d_len[i - nll];
    __syncthreads();
    // token starts: a non-zero length is its own token; a run of R zeros is R / 138 tokens "18 x 138", then one "18" (rest >= 11), one
    // "17" (rest >= 3) or the rest as literal zeros -- what the greedy left-to-right scan produces
    uint32_t tsym[RND], text[RND], tidx[RND]; bool tst[RND];
    // the zero runs' ends come from ballots instead of walks along the run (a 4 KiB entry leaves runs of a hundred zeros and more, and every
    // lane of a run walked all of it): per (round, wave) the first / last position that is non-zero -- or beyond n, which ends a run too
    __shared__ int16_t nz_first[8], nz_last[8];
    uint32_t vv[RND]; uint64_t nzm[RND];
#pragma unroll
    for (uint32_t c2 = 0; c2 < RND; c2++) {
        const uint32_t i = c2 * T + tid;
        tsym[c2] = 0; text[c2] = 0; tidx[c2] = 0; tst[c2] = false;
        vv[c2] = i < n ? (uint32_t)seq[i] : 1u;
        nzm[c2] = __ballot(vv[c2] != 0);
        if (lane == 0) {
            const int base = (int)(c2 * T + wv * 64);
            nz_first[c2 * NW + wv] = nzm[c2] ? (int16_t)(base + __builtin_ctzll(nzm[c2])) : (int16_t)-1;
            nz_last[c2 * NW + wv] = nzm[c2] ? (int16_t)(base + 63 - __builtin_clzll(nzm[c2])) : (int16_t)-1;
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t c2 = 0; c2 < RND; c2++) {
        const uint32_t i = c2 * T + tid;
        if (i < n) {
            const uint32_t v = vv[c2];
            if (v) { tst[c2] = true; tsym[c2] = v; }
            else {
                const int slot = (int)(c2 * NW + wv), base = (int)(c2 * T + wv * 64);
                const uint64_t lt = ((uint64_t)1 << lane) - 1;
                const uint64_t below = nzm[c2] & lt, above = nzm[c2] & ~(lt | ((uint64_t)1 << lane));
                int pb = -1, pa = (int)n;                                  // nearest run-ending position below / above (position n always ends a run)
                if (below) pb = base + 63 - __builtin_clzll(below);
                else for (int s2 = slot - 1; s2 >= 0; s2--) if (nz_last[s2] >= 0) { pb = nz_last[s2]; break; }
                if (above) pa = base + __builtin_ctzll(above);
                else for (int s2 = slot + 1; s2 < (int)NSLOT; s2++) if (nz_first[s2] >= 0) { pa = nz_first[s2]; break; }
                const uint32_t a = (uint32_t)(pb + 1), e = (uint32_t)(pa - 1);
                const uint32_t R = e - a + 1, o = i - a, q = R / 138, rem = R - q * 138;
                if (o < q * 138) { tst[c2] = (o % 138) == 0; tsym[c2] = 18; text[c2] = 138 - 11; }
                else if (rem >= 11) { tst[c2] = o == q * 138; tsym[c2] = 18; text[c2] = rem - 11; }
                else if (rem >= 3) { tst[c2] = o == q * 138; tsym[c2] = 17; text[c2] = rem - 3; }
                else { tst[c2] = true; tsym[c2] = 0; }
            }
        }
        const uint64_t m = __ballot(tst[c2]);
        tidx[c2] = (uint32_t)__popcll(m & (((uint64_t)1 << lane) - 1));
        if (lane == 0) wsum[c2 * NW + wv] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint32_t ns = 0;
    for (uint32_t k = 0; k < NSLOT; k++) ns += wsum[k];
#pragma unroll
    for (uint32_t c2 = 0; c2 < RND; c2++)
        if (tst[c2]) {
            uint32_t base = 0;
            for (uint32_t k = 0; k < c2 * NW + wv; k++) base += wsum[k];
            sym[base + tidx[c2]] = (uint8_t)tsym[c2]; ext[base + tidx[c2]] = (uint8_t)text[c2];
            atomicAdd(&clc[tsym[c2]], 1u);
        }
    __syncthreads();
    int n_cl = d_build_lens<T>(clc, 19, 7, cl_len, order, wt, parent, sh2, tid);
    if (n_cl == 1) {
        if (tid == 0) for (int k = 0; k < 19; k++) if (!cl_len[k]) { cl_len[k] = 1; break; }
        n_cl = 2;
        __syncthreads();
    }
    d_assign<T>(cl_len, 19, cl_code, first_s, cntw_s, tid);
    int ncl = 19; while (ncl > 4 && cl_len[D_CL_ORDER[ncl - 1]] == 0) ncl--;
    // bit lengths of the tokens -> positions (packed scans over the waves) -> ORed into the header image in LDS
    auto put_bits = [&](uint32_t pos, uint32_t v, uint32_t nb2) {       // nb2 <= 14: at most two words
        if (!nb2) return;
        const uint32_t w2 = pos >> 5, sh = pos & 31;
        atomicOr(&hbuf[w2], v << sh);
        if (sh + nb2 > 32) atomicOr(&hbuf[w2 + 1], v >> (32 - sh));
    };
    uint32_t tl[RND], tv[RND], tp[RND];
    __syncthreads();
#pragma unroll
    for (uint32_t c2 = 0; c2 < RND; c2++) {
        const uint32_t k = c2 * T + tid;
        tl[c2] = 0; tv[c2] = 0;
        if (k < ns) {
            const uint32_t sy = sym[k], cc = cl_code[sy], cl = cc >> 16, eb = sy == 17 ? 3u : (sy == 18 ? 7u : 0u);
            tl[c2] = cl + eb; tv[c2] = (cc & 0xFFFF) | ((uint32_t)ext[k] << cl);
        }
        uint32_t x = tl[c2];                                            // inclusive prefix over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d); if ((int)lane >= d) x += y; }
        tp[c2] = x - tl[c2];
        if (lane == 63) wsum2[c2 * NW + wv] = x;
    }
    __syncthreads();
    const uint32_t hbits = 14 + 3 * (uint32_t)ncl;
    uint32_t total = hbits;
    for (uint32_t k = 0; k < NSLOT; k++) total += wsum2[k];
#pragma unroll
    for (uint32_t c2 = 0; c2 < RND; c2++)
        if (tl[c2]) {
            uint32_t base = hbits;
            for (uint32_t k = 0; k < c2 * NW + wv; k++) base += wsum2[k];
            put_bits(base + tp[c2], tv[c2], tl[c2]);
        }
    if (tid == 0) { put_bits(0, nll - 257, 5); put_bits(5, nd - 1, 5); put_bits(10, (uint32_t)(ncl - 4), 4); T_->hdr_bits = total; }
    if (tid < (uint32_t)ncl) put_bits(14 + 3 * tid, cl_len[D_CL_ORDER[tid]], 3);
    __syncthreads();
    for (uint32_t i = tid; i < 100; i += T) ((uint32_t *)T_->hdr)[i] = hbuf[i];
}

// ------------------------------------------------------------------ Adler-32 halves of one block's input
// T = 256: a workgroup per block; T = 64 (batches of many small entries: blocks of a few KiB): a wave per block, the sums reduced by lane shuffles -- no LDS,
// no barriers (the workgroup form spends eight barriers on a 4 KiB block's 256 partial sums)
template <uint32_t T>
__global__ __launch_bounds__(T)
void k_adler(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, const uint32_t *__restrict__ blk_seg, BlkInfo *__restrict__ blk) {
    __shared__ unsigned long long r1[T == 64 ? 1 : T], r2[T == 64 ? 1 : T];
    const uint32_t tid = threadIdx.x, g = blockIdx.x;
    const SegDesc sd = segs[blk_seg[g]];
    const uint32_t bsz = 1u << sd.blk_log, b0 = (g - sd.blk_base) * bsz, n = sd.len - b0 < bsz ? sd.len - b0 : bsz;
    const uint8_t *p = src + sd.src_off + b0;
    unsigned long long s1 = 0, s2 = 0;
    // S1 = sum d_i ; S2 = sum (n - i) d_i   (16-byte vector loads; src offsets are 16-byte aligned)
    for (uint32_t i = tid * 16; i < n; i += T * 16) {
        if (i + 16 <= n) {
            const uint4 v = *(const uint4 *)(p + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) { const uint32_t d = (w[k >> 2] >> (8 * (k & 3))) & 0xFF; s1 += d; s2 += (unsigned long long)(n - i - k) * d; }
        } else for (uint32_t k = i; k < n; k++) { const uint32_t d = p[k]; s1 += d; s2 += (unsigned long long)(n - k) * d; }
    }
    if (T == 64) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(s1 >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)s1, o);
            s2 += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(s2 >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)s2, o);
        }
        if (tid == 0) { blk[g].adler_a = (uint32_t)((1 + s1) % ADLER_P); blk[g].adler_b = (uint32_t)((n + s2) % ADLER_P); }
        return;
    }
    r1[tid] = s1; r2[tid] = s2;
    __syncthreads();
    for (uint32_t s = T / 2; s > 0; s >>= 1) { if (tid < s) { r1[tid] += r1[tid + s]; r2[tid] += r2[tid + s]; } __syncthreads(); }
    if (tid == 0) { blk[g].adler_a = (uint32_t)((1 + r1[0]) % ADLER_P); blk[g].adler_b = (uint32_t)((n + r2[0]) % ADLER_P); }
}

// ------------------------------------------------------------------ k_dblock : one workgroup per block, token-parallel
// The block's tiles (k_lz's 2 KiB parse tiles) are packed one after the other by all 256 threads; inside a tile every
// literal and every match is placed independently.  k_lz's chunk table gives the sequence / literal stream positions at the
// tile start and the literal index of the tile's first match.  With literal SLOT s = local literal index, a match sits in
// the slot of the literal that follows it (slot nl = after the tile's last literal), so in stream order a slot is
// [its matches..., its literal].  One packed scan over the sequences yields (slot, Mx = match bits before this match),
// one packed scan over the slots yields (Lx = literal bits before the slot, match bits in earlier slots):
//     literal s at  base + Lx[s] + Mb[<s] + wm[s]          match k at  base + Lx[slot(k)] + Mx[k]
// Tokens are ORed into an LDS staging area (64-bit units) and flushed with coalesced stores; the partial unit at the tile
// end is carried into the next tile, so the global stream is written exactly once and never read.
constexpr uint32_t DB_THREADS = 256;
constexpr uint32_t DB_STAGE = 768;                           // u64 units: 2048 x 15 + 342 x 48 bits per tile + carry
static_assert(TILE == 2048 && TILE / MIN_MATCH <= 2 * DB_THREADS, "k_dblock handles two sequences and eight literal slots per thread and tile");

#define DB_ROW_SHR(v, k) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x110 + (k), 0xF, 0xF, true))
__device__ __forceinline__ uint32_t db_wave_scan(uint32_t v, uint32_t lane) {     // inclusive, 64 lanes, VALU only
    v += DB_ROW_SHR(v, 1); v += DB_ROW_SHR(v, 2); v += DB_ROW_SHR(v, 4); v += DB_ROW_SHR(v, 8);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const uint32_t row = lane >> 4;
    return v + (row > 0 ? r0 : 0u) + (row > 1 ? r1 : 0u) + (row > 2 ? r2 : 0u);
}
// inclusive scan over the workgroup's 256 threads (one barrier); total = sum over all threads
__device__ __forceinline__ uint32_t db_wg_scan(uint32_t v, uint32_t *wsum, uint32_t tid, uint32_t &total) {
    const uint32_t lane = tid & 63, wave = tid >> 6;
    uint32_t sc = db_wave_scan(v, lane);
    if (lane == 63) wsum[wave] = sc;
    __syncthreads();
    const uint32_t w0 = wsum[0], w1 = wsum[1], w2 = wsum[2], w3 = wsum[3];
    total = w0 + w1 + w2 + w3;
    return sc + (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
}
__device__ __forceinline__ void db_put(unsigned long long *st, uint32_t off, unsigned long long v, uint32_t n) {
    if (!n) return;
    const uint32_t u = off >> 6, sh = off & 63;
    atomicOr(&st[u], v << sh);
    if (sh + n > 64) atomicOr(&st[u + 1], v >> (64 - sh));
}

__global__ __launch_bounds__(DB_THREADS)
void k_dblock(const SegDesc *__restrict__ segs, const uint32_t *__restrict__ blk_seg, const uint64_t *__restrict__ seqs,
              const uint8_t *__restrict__ lits, BlkInfo *__restrict__ blk, const uint4 *__restrict__ ctab,
              const DeflTables *__restrict__ tabs, uint8_t *__restrict__ outc, uint32_t dbg) {
    // Round 5: 25.5 -> 15.5 KiB of LDS per workgroup -- the kernel's time for small blocks is inversely proportional to the workgroups a CU holds (10^6 x 4 KiB: 8.6 ms at six,
    // 11.8 at four, 22 at two): match bits and literal-bit prefixes per slot as 16-bit halves of words (both stay below 2^15 per tile: 2 048 x 15 + 342 x 48 bits), the
    // literal bytes straight from memory (eight per thread, one unaligned 8-byte load) instead of through a staging array
    __shared__ uint32_t t_ll[288];
    __shared__ uint32_t t_d[32];
    __shared__ __attribute__((aligned(16))) uint32_t wm2[TILE / 2 + 8];                   // match bits per literal slot, two slots per word (atomic adds of a shifted value: no half overflows)
    __shared__ __attribute__((aligned(16))) uint32_t lx2[TILE / 2 + 8];                   // literal bits in front of each slot (exclusive scan inside the tile), two slots per word
    __shared__ unsigned long long stage[DB_STAGE];
    __shared__ uint32_t wsum1[4], wsum2[4];
    (void)dbg;
    const uint32_t tid = threadIdx.x, g = blockIdx.x;
    const uint32_t sidx = blk_seg[g];
    const SegDesc sd = segs[sidx];
    const DeflTables *T = tabs + sidx;
    for (uint32_t i = tid; i < 288; i += DB_THREADS) t_ll[i] = T->ll_code[i];
    if (tid < 32) t_d[tid] = T->d_code[tid];
    for (uint32_t i = tid; i < DB_STAGE; i += DB_THREADS) stage[i] = 0;
    for (uint32_t i = tid; i < TILE / 2 + 8; i += DB_THREADS) wm2[i] = 0;
    const uint32_t b = g - sd.blk_base, nblk = seg_nblk(sd), bsz = 1u << sd.blk_log;
    const uint32_t bl_len = sd.len - b * bsz < bsz ? sd.len - b * bsz : bsz;
    const bool last = (sd.first & 2) && (b + 1 == nblk);
    const uint32_t ntile = (bl_len + TILE - 1) / TILE;
    const uint32_t nseq = blk[g].nseq, nlit = blk[g].nlit;
    const uint8_t *bl = lits + ((size_t)g << sd.blk_log);
    const uint64_t *bs = seqs + (size_t)g * seq_cap_of(sd.blk_log);
    unsigned long long *out64 = (unsigned long long *)(outc + ((size_t)g << sd.blk_log));
    const uint32_t hdr_bits = T->hdr_bits;
    __syncthreads();

    // block header: BFINAL, BTYPE = 2, then the table description (one byte per thread)
    if (tid == 0) db_put(stage, 0, (last ? 1u : 0u) | (2u << 1), 3);
    for (uint32_t i = tid; i * 8 < hdr_bits; i += DB_THREADS) {
        const uint32_t n = hdr_bits - 8 * i < 8 ? hdr_bits - 8 * i : 8;
        db_put(stage, 3 + 8 * i, (unsigned long long)(T->hdr[i] & ((1u << n) - 1)), n);
    }
    uint32_t bitpos = 3 + hdr_bits;                          // bits produced so far (uniform)
    uint32_t flushed = 0;                                    // 64-bit units already stored to global

    // flush the complete units of the staging area, carry the partial one to its front, clear the rest and the slots
    auto flush = [&](uint32_t nslots) {
        __syncthreads();
        const uint32_t nfull = (bitpos >> 6) - flushed;
        for (uint32_t i = tid; i < nfull; i += DB_THREADS) if (flushed + i < bsz / 8) out64[flushed + i] = stage[i];
        const unsigned long long carry = stage[nfull];
        __syncthreads();
        for (uint32_t i = tid; i <= nfull + 1 && i < DB_STAGE; i += DB_THREADS) stage[i] = i == 0 ? carry : 0ull;
        for (uint32_t i = tid; i <= nslots / 2; i += DB_THREADS) wm2[i] = 0;
        flushed += nfull;
        __syncthreads();
    };
    flush(0);

    for (uint32_t t = 0; t < ntile; t++) {
        const uint4 c = ctab[((size_t)g << (sd.blk_log - 11)) + t];
        const uint32_t s0 = c.x, l0 = c.y;
        uint32_t s1 = nseq, l1 = nlit;
        if (t + 1 < ntile) { const uint4 cn = ctab[((size_t)g << (sd.blk_log - 11)) + t + 1]; s1 = cn.x; l1 = cn.y; }
        const uint32_t ns = s1 - s0, nl = l1 - l0;
        const uint32_t gf = (ns ? c.z : l1) - l0;            // slot of the tile's first match
        const uint32_t base = bitpos - (flushed << 6);      // staging bit offset of the tile's first token

        // this thread's eight literal bytes (slots 8 tid ..), requested now: one unaligned 8-byte load (the literal array of a block is followed by the next block's / the
        // buffer's slack, so the load may run past nl)
        typedef unsigned long long u64u __attribute__((aligned(1)));
        const unsigned long long lit8 = tid * 8 < nl ? *(const u64u *)(bl + l0 + tid * 8) : 0ull;

        // ---- sequences: two per thread
        uint32_t mb[2] = {0, 0}, inc[2] = {0, 0};
        unsigned long long tok[2] = {0, 0};
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const uint32_t k = 2 * tid + r;
            if (k < ns) {
                const uint64_t s = bs[s0 + k];
                uint32_t lc_i, leb, lev, dc_i, deb, dev;
                len_sym(seq_ml(s), lc_i, leb, lev);
                dist_sym(seq_off(s), dc_i, deb, dev);
                const uint32_t lc = t_ll[257 + lc_i], dc = t_d[dc_i];
                const uint32_t ln = lc >> 16, dn = dc >> 16;
                unsigned long long v = lc & 0xFFFF; uint32_t n = ln;
                v |= (unsigned long long)lev << n; n += leb;
                v |= (unsigned long long)(dc & 0xFFFF) << n; n += dn;
                v |= (unsigned long long)dev << n; n += deb;
                tok[r] = v; mb[r] = n;
                inc[r] = k == 0 ? gf : seq_ll(s);
            }
        }
        uint32_t tot1;
        const uint32_t p1 = (inc[0] + inc[1]) | ((mb[0] + mb[1]) << 16);
        const uint32_t ex1 = db_wg_scan(p1, wsum1, tid, tot1) - p1;
        const uint32_t slot0 = (ex1 & 0xFFFF) + inc[0], slot1 = slot0 + inc[1];
        const uint32_t mx0 = ex1 >> 16, mx1 = mx0 + mb[0];
        if (mb[0]) atomicAdd(&wm2[slot0 >> 1], mb[0] << (16 * (slot0 & 1)));
        if (mb[1]) atomicAdd(&wm2[slot1 >> 1], mb[1] << (16 * (slot1 & 1)));
        __syncthreads();

        // ---- literal slots: eight per thread
        uint32_t code[8], wmv[8];
        uint32_t lsum = 0, msum = 0;
        {
            const uint32_t x0 = (uint32_t)lit8, x1 = (uint32_t)(lit8 >> 32);
            const uint4 wv = *(const uint4 *)&wm2[tid * 4];                         // the eight slots' match bits
            const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t sl = tid * 8 + i;
                const uint32_t byte = ((i < 4 ? x0 : x1) >> (8 * (i & 3))) & 0xFF;
                code[i] = sl < nl ? t_ll[byte] : 0u;
                wmv[i] = (ww[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                lsum += code[i] >> 16; msum += wmv[i];
            }
        }
        uint32_t tot2;
        const uint32_t p2 = lsum | (msum << 16);
        uint32_t run = db_wg_scan(p2, wsum2, tid, tot2) - p2;
        {
            unsigned long long acc = 0; uint32_t nacc = 0, aoff = 0;
            uint32_t lxw[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                lxw[i >> 1] |= (run & 0xFFFFu) << (16 * (i & 1));
                const uint32_t ln = code[i] >> 16;
                if (wmv[i] && nacc) { db_put(stage, aoff, acc, nacc); acc = 0; nacc = 0; }
                if (ln) {
                    if (!nacc) aoff = base + (run & 0xFFFF) + (run >> 16) + wmv[i];
                    acc |= (unsigned long long)(code[i] & 0xFFFF) << nacc; nacc += ln;
                    if (nacc > 48) { db_put(stage, aoff, acc, nacc); acc = 0; nacc = 0; }
                }
                run += ln | (wmv[i] << 16);
            }
            if (nacc) db_put(stage, aoff, acc, nacc);
            *(uint4 *)&lx2[tid * 4] = make_uint4(lxw[0], lxw[1], lxw[2], lxw[3]);
            if (tid == DB_THREADS - 1) lx2[TILE / 2] = run & 0xFFFFu;                 // (slot TILE: a match behind the tile's last literal)
        }
        __syncthreads();

        // ---- matches
        if (mb[0]) db_put(stage, base + ((lx2[slot0 >> 1] >> (16 * (slot0 & 1))) & 0xFFFFu) + mx0, tok[0], mb[0]);
        if (mb[1]) db_put(stage, base + ((lx2[slot1 >> 1] >> (16 * (slot1 & 1))) & 0xFFFFu) + mx1, tok[1], mb[1]);
        bitpos += (tot2 & 0xFFFF) + (tot1 >> 16);
        flush(nl);
    }
    // end of block, and the empty stored block header of the sync flush when more blocks follow
    if (tid == 0) { const uint32_t eob = t_ll[256]; db_put(stage, bitpos - (flushed << 6), eob & 0xFFFF, eob >> 16); }
    bitpos += (t_ll[256] >> 16) + (last ? 0u : 3u);
    __syncthreads();
    const uint32_t bytes = (bitpos + 7) / 8;
    const uint32_t nun = (bitpos + 63) / 64 - flushed;
    for (uint32_t i = tid; i < nun; i += DB_THREADS) if (flushed + i < bsz / 8) out64[flushed + i] = stage[i];
    if (tid == 0) blk[g].lit_body = bytes;
}

// ------------------------------------------------------------------ k_dplan : one thread per segment
// stored != 0: deflate level 0 (Compression::none(), lib/src/compress/deflate.rs:89-101): every block a stored block, no dynamic block was built
__global__ void k_dplan(const SegDesc *__restrict__ segs, uint32_t nseg, BlkInfo *__restrict__ blk, uint64_t *__restrict__ seg_size, uint32_t stored_only) {
    const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nseg) return;
    const SegDesc sd = segs[sidx];
    if (sd.len == 0) { seg_size[sidx] = stored_only ? 11 : 8; return; }        // empty entry: 78 9C 03 00 00 00 00 01 (level 0: 78 01 + an empty final stored block + Adler-32)
    const uint32_t nblk = seg_nblk(sd), bsz = 1u << sd.blk_log;
    uint64_t off = (sd.first & 1) ? 2 : 0;                  // zlib header in front of the entry's first segment
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t g = sd.blk_base + b;
        const uint32_t b0 = b * bsz, bl_len = sd.len - b0 < bsz ? sd.len - b0 : bsz;
        const bool last = (sd.first & 2) && (b + 1 == nblk);
        const uint32_t dyn = blk[g].lit_body, stored = bl_len + 5 * ((bl_len + 65534) / 65535);
        uint32_t sz, plan;
        if (stored_only || dyn >= stored || dyn > bsz) { plan = 0; sz = stored + (last ? 0 : 5); }   // same rule in the model
        else { plan = 1; sz = dyn + (last ? 0 : 4); }
        blk[g].plan = plan; blk[g].out_size = sz; blk[g].out_off = off;
        off += sz;
    }
    if (sd.first & 2) off += 4;                             // Adler-32 trailer
    seg_size[sidx] = off;
}

// ------------------------------------------------------------------ k_dwrite : one workgroup per block
__global__ __launch_bounds__(256)
void k_dwrite(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, const uint32_t *__restrict__ blk_seg,
              const BlkInfo *__restrict__ blk, const uint64_t *__restrict__ seg_off, const uint8_t *__restrict__ outc,
              uint8_t *__restrict__ dst) {
    const uint32_t tid = threadIdx.x, g = blockIdx.x;
    const uint32_t sidx = blk_seg[g];
    const SegDesc sd = segs[sidx];
    const BlkInfo bi = blk[g];
    const uint32_t b = g - sd.blk_base, nblk = seg_nblk(sd), bsz = 1u << sd.blk_log;
    const uint32_t b0 = b * bsz, bl_len = sd.len - b0 < bsz ? sd.len - b0 : bsz;
    const bool last = (sd.first & 2) && (b + 1 == nblk);
    uint8_t *out = dst + seg_off[sidx] + bi.out_off;
    const uint32_t T = blockDim.x;                               // 256, or 64 for the batches of many small blocks (a wave per block: four times the blocks in flight)
    // n bytes from p to o: the destination's unaligned head and tail byte by byte, its aligned middle as 16-byte stores of unaligned 16-byte loads
    auto copy = [&](uint8_t *o, const uint8_t *p, uint32_t n) {
        const uint32_t head = (uint32_t)((16u - ((uintptr_t)o & 15u)) & 15u), h = head < n ? head : n, mid = (n - h) >> 4;
        if (tid < h) o[tid] = p[tid];
        for (uint32_t i = tid; i < mid; i += T) *(v4u *)(o + h + 16 * i) = ld16u(p + h + 16 * i);
        for (uint32_t i = h + 16 * mid + tid; i < n; i += T) o[i] = p[i];
    };
    if (bi.plan & 1) {
        const uint8_t *p = outc + ((size_t)g << sd.blk_log);
        copy(out, p, bi.lit_body);
        if (!last && tid < 4) out[bi.lit_body + tid] = (tid < 2) ? 0x00 : 0xFF;
    } else {
        const uint8_t *p = src + sd.src_off + b0;
        uint32_t pos = 0;
        for (uint32_t o = 0; o < bl_len; o += 65535) {
            const uint32_t k = bl_len - o < 65535 ? bl_len - o : 65535;
            if (tid == 0) {
                out[pos] = (last && o + k >= bl_len) ? 1 : 0;
                out[pos + 1] = (uint8_t)k; out[pos + 2] = (uint8_t)(k >> 8); out[pos + 3] = (uint8_t)~k; out[pos + 4] = (uint8_t)(~k >> 8);
            }
            copy(out + pos + 5, p + o, k);
            pos += 5 + k;
        }
        if (!last && tid < 5) out[pos + tid] = (tid < 3) ? 0x00 : 0xFF;
    }
}

// ------------------------------------------------------------------ k_dfinal : one thread per entry
__global__ void k_dfinal(const SegDesc *__restrict__ segs, const uint32_t *__restrict__ entry_seg, uint32_t nentry,
                         const BlkInfo *__restrict__ blk, const uint64_t *__restrict__ seg_off, const uint64_t *__restrict__ seg_size,
                         uint8_t *__restrict__ dst, uint32_t stored_only) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nentry) return;
    const uint32_t s0 = entry_seg[e], s1 = entry_seg[e + 1];
    uint8_t *o = dst + seg_off[s0];
    if (segs[s0].len == 0) {
        if (stored_only) { const uint8_t z[11] = {0x78, 0x01, 0x01, 0x00, 0x00, 0xFF, 0xFF, 0x00, 0x00, 0x00, 0x01}; for (int i = 0; i < 11; i++) o[i] = z[i]; return; }
        const uint8_t z[8] = {0x78, 0x9C, 0x03, 0x00, 0x00, 0x00, 0x00, 0x01}; for (int i = 0; i < 8; i++) o[i] = z[i]; return;
    }
    o[0] = 0x78; o[1] = stored_only ? 0x01 : 0x9C;          // FLEVEL 0 = "fastest", what a level-0 zlib stream says (FCHECK makes 0x7801 a multiple of 31)
    // adler(X || Y): A = A_x + A_y - 1, B = B_x + B_y + len_y * (A_x - 1)   (mod 65521)
    unsigned long long A = 1, B = 0;
    for (uint32_t s = s0; s < s1; s++) {
        const SegDesc sd = segs[s];
        const uint32_t nblk = seg_nblk(sd), bsz = 1u << sd.blk_log;
        for (uint32_t b = 0; b < nblk; b++) {
            const BlkInfo bi = blk[sd.blk_base + b];
            const uint32_t n = sd.len - b * bsz < bsz ? sd.len - b * bsz : bsz;
            B = (B + bi.adler_b + (unsigned long long)(n % ADLER_P) * ((A + ADLER_P - 1) % ADLER_P)) % ADLER_P;
            A = (A + bi.adler_a + ADLER_P - 1) % ADLER_P;
        }
    }
    uint8_t *t = dst + seg_off[s1 - 1] + seg_size[s1 - 1] - 4;     // segment offsets need not be contiguous (in-HBM framing)
    t[0] = (uint8_t)(B >> 8); t[1] = (uint8_t)B; t[2] = (uint8_t)(A >> 8); t[3] = (uint8_t)A;
}

void k_scan_launch_big(const uint64_t *in, uint64_t *out, uint32_t n, hipStream_t st);

void launch_deflate_stage1(const uint8_t *src, const SegDesc *segs, uint32_t nseg, const uint32_t *blk_seg, uint32_t nblk,
                           const uint64_t *seqs, const uint8_t *lits, BlkInfo *blk, const uint4 *ctab, DeflTables *tabs, uint8_t *outc,
                           uint64_t *seg_size, uint64_t *seg_off, hipStream_t st, hipEvent_t *ev, uint32_t dbg, bool stored_only, bool wave_per_seg) {
    if (!stored_only) {
        if (wave_per_seg) hipLaunchKernelGGL(k_dstats<64>, dim3(nseg), dim3(64), 0, st, segs, seqs, lits, blk, tabs);
        else hipLaunchKernelGGL(k_dstats<256>, dim3(nseg), dim3(256), 0, st, segs, seqs, lits, blk, tabs);
    }
    if (nblk && wave_per_seg) hipLaunchKernelGGL(k_adler<64>, dim3(nblk), dim3(64), 0, st, src, segs, blk_seg, blk);
    else if (nblk) hipLaunchKernelGGL(k_adler<256>, dim3(nblk), dim3(256), 0, st, src, segs, blk_seg, blk);
    if (ev) (void)hipEventRecord(ev[0], st);
    if (ev) (void)hipEventRecord(ev[1], st);
    if (nblk && !stored_only) hipLaunchKernelGGL(k_dblock, dim3(nblk), dim3(DB_THREADS), 0, st, segs, blk_seg, seqs, lits, blk, ctab, tabs, outc, dbg);
    if (ev) (void)hipEventRecord(ev[2], st);
    hipLaunchKernelGGL(k_dplan, dim3((nseg + 255) / 256), dim3(256), 0, st, segs, nseg, blk, seg_size, stored_only ? 1u : 0u);
    k_scan_launch_big(seg_size, seg_off, nseg, st);
    if (ev) (void)hipEventRecord(ev[3], st);
}

void launch_deflate_write(const uint8_t *src, const SegDesc *segs, const uint32_t *blk_seg, uint32_t nblk, const BlkInfo *blk,
                          const uint64_t *seg_off, const uint64_t *seg_size, const uint8_t *outc, const uint32_t *entry_seg, uint32_t nentry,
                          uint8_t *dst, hipStream_t st, bool stored_only, bool small_blocks) {
    if (nblk) hipLaunchKernelGGL(k_dwrite, dim3(nblk), dim3(small_blocks ? 64 : 256), 0, st, src, segs, blk_seg, blk, seg_off, outc, dst);
    hipLaunchKernelGGL(k_dfinal, dim3((nentry + 255) / 256), dim3(256), 0, st, segs, entry_seg, nentry, blk, seg_off, seg_size, dst, stored_only ? 1u : 0u);
}

} // namespace pna
