// k_lz.hip -- LZ77 match finding + greedy parse for gfx950 (CDNA4), one 1024-thread workgroup per segment.
//
// Replaces the match finder inside the third-party encoder the reference drives at
// lib/src/compress.rs:32-41 (CompressionWriter::write -> ZstdEncoder::write).  Integer/byte work, no MFMA.
//
// LDS (one workgroup per CU, ~146 KiB of the 160 KiB):
//   win   [65536 B]  circular copy of the segment's most recent 64 KiB (look-back + 1 KiB look-ahead)
//   table [16384 x u32] hash table: (position+1) << 11 | 11-bit tag of the latest occurrence (0 = empty);
//         inserts are ds_max_u32, the tag lets a lookup skip candidates whose 6 bytes cannot match
//   len/off/fixlen [2048 x u16 each]  written and read by the serial fallback only
//   per-wave records
// Per tile of 2048 positions (2 per lane):
//   (next window chunk requested into registers) lookup -> B2 -> insert + match +
//   per-wave SPECULATIVE parse (as if the parse entered the wave at its first position; scalar loops that also
//   build the coverage bit masks) -> B3 -> every wave resolves its TRUE entry in parallel (carry chained through
//   the speculative exits of the earlier, not fully covered waves) and walks from there until it lands on a
//   position its speculative parse also stood on -> masks of selected matches / literals -> B4 -> 16-lane DPP
//   scans of the waves' counts -> emission of sequences and literals straight to HBM.
//   Cross-lane traffic on this path: ballots, readlanes, DPP and one ds_bpermute per 64 positions.
//   The parallel resolution is exact when every not-fully-covered wave re-synchronises (checked); otherwise the
//   tile falls back to a serial resolution by wave 0 (rare; forced with flag 0x200 for testing).
#include <hip/hip_runtime.h>
#include "pna_dev.h"

namespace pna {

constexpr uint32_t TAG_BITS = 11, TAG_MASK = (1u << TAG_BITS) - 1;

// LDS layout (byte offsets into the dynamic shared array)
constexpr uint32_t L_WIN    = 0;
constexpr uint32_t WIN_MIRROR = 16;                        // the window's first 16 bytes again behind its end: unaligned reads never wrap
constexpr uint32_t L_TABLE  = L_WIN + WIN_BYTES + WIN_MIRROR;
constexpr uint32_t L_LEN    = L_TABLE + (4u << HASH_LOG);
constexpr uint32_t L_OFF    = L_LEN + 2 * TILE;
constexpr uint32_t L_FIXLEN = L_OFF + 2 * TILE;
constexpr uint32_t L_WMETA  = L_FIXLEN + 2 * TILE;          // 16 x 64 B
constexpr uint32_t L_WRES   = L_WMETA + 64 * LZ_WAVES;      // 16 x 48 B
constexpr uint32_t L_WPUB   = L_WRES + 48 * LZ_WAVES;       // 16 x 16 B
constexpr uint32_t L_STATE  = L_WPUB + 16 * LZ_WAVES;       // 16 B
constexpr uint32_t L_TOTAL  = L_STATE + 16;

struct WMeta { uint64_t sel[2]; uint64_t vis[2]; uint64_t eff[2]; uint32_t exit0; uint32_t pad[3]; };
struct WRes  { uint64_t fix[2]; uint64_t fcov[2]; uint32_t carry; uint32_t sync; uint32_t exit; uint32_t pad; };
struct WPub  { uint32_t cnt; uint32_t gl; uint32_t bad; uint32_t exit; };   // cnt = nsel | nlit << 16; gl = (local literal index of the LAST match + 1) | (same for the FIRST match) << 16, 0 = no match
static_assert(sizeof(WMeta) == 64 && sizeof(WRes) == 48 && sizeof(WPub) == 16, "LDS record sizes");

constexpr uint32_t FLAG_STAMP = 0x100u, FLAG_FORCE_FALLBACK = 0x200u;
static_assert(CAP1 == 16, "the match step compares 8 + 8 bytes");

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, uint32_t l) {
    return (uint64_t)rdlane((uint32_t)v, l) | ((uint64_t)rdlane((uint32_t)(v >> 32), l) << 32);
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t ctz64(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
__device__ __forceinline__ uint32_t clz64(uint64_t v) { return (uint32_t)__builtin_clzll(v); }
__device__ __forceinline__ uint64_t mlow(uint32_t n) { return n >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << n) - 1); }   // bits [0, n)

// DPP helpers (VALU only): value of lane i-k inside each row of 16 lanes (0 outside), and of lane i+1 of the wave
#define DPP_ROW_SHR(v, k) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x110 + (k), 0xF, 0xF, true))
__device__ __forceinline__ uint32_t dpp_next_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t row_scan_add(uint32_t v) {          // inclusive prefix sum inside a row of 16 lanes
    v += DPP_ROW_SHR(v, 1); v += DPP_ROW_SHR(v, 2); v += DPP_ROW_SHR(v, 4); v += DPP_ROW_SHR(v, 8); return v;
}
__device__ __forceinline__ uint32_t row_scan_max(uint32_t v) {
    uint32_t t;
    t = DPP_ROW_SHR(v, 1); v = v > t ? v : t; t = DPP_ROW_SHR(v, 2); v = v > t ? v : t;
    t = DPP_ROW_SHR(v, 4); v = v > t ? v : t; t = DPP_ROW_SHR(v, 8); v = v > t ? v : t; return v;
}

// 8 / 4 bytes at an arbitrary segment position from the circular window
__device__ __forceinline__ void fetch8(const uint32_t *win32, uint32_t pos, uint32_t &lo, uint32_t &hi) {
    const uint32_t *p = win32 + ((pos & (WIN_BYTES - 1)) >> 2);                    // p[1], p[2] may lie in the mirror
    const uint32_t sh = (pos & 3) * 8;
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    lo = __builtin_amdgcn_alignbit(d1, d0, sh);
    hi = __builtin_amdgcn_alignbit(d2, d1, sh);
}
__device__ __forceinline__ uint32_t fetch4(const uint32_t *win32, uint32_t pos) {
    const uint32_t *p = win32 + ((pos & (WIN_BYTES - 1)) >> 2);
    return __builtin_amdgcn_alignbit(p[1], p[0], (pos & 3) * 8);
}

// Wave-cooperative extension of a match that reached CAP1: q, c, lim are wave-uniform; returns the full length
// (<= lim).  64 lanes x 4 bytes per step.
__device__ __forceinline__ uint32_t lz_extend(const uint32_t *win32, uint32_t q, uint32_t c, uint32_t lim, uint32_t lane) {
    uint32_t L = CAP1;
    for (;;) {
        uint32_t pos = L + lane * 4;
        uint32_t x = fetch4(win32, q + pos) ^ fetch4(win32, c + pos);
        uint32_t nb = x ? ((uint32_t)__builtin_ctz(x) >> 3) : 4u;
        uint32_t room = lim > pos ? lim - pos : 0u;
        nb = nb < room ? nb : room;
        uint64_t bad = __ballot(nb < 4u);
        if (bad) { uint32_t f = ctz64(bad); L += 4 * f + rdlane(nb, f); break; }
        L += 256;
    }
    return L;
}

__device__ __forceinline__ uint4 load_chunk(const uint8_t *seg, uint32_t i, uint32_t seg_len) {
    if (i + 16 <= seg_len) return *(const uint4 *)(seg + i);
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16; k++) if (i + k < seg_len) w[k >> 2] |= (uint32_t)seg[i + k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// coverage of a match at wave-relative position p (0..127) with length L, as bit masks of the wave's two groups
__device__ __forceinline__ void cover(uint64_t &c0, uint64_t &c1, uint32_t p, uint32_t L) {
    const uint32_t end = p + L;
    if (p < 64) c0 |= mlow(end) & ~mlow(p);
    if (end > 64) c1 |= mlow(end - 64) & ~mlow(p > 64 ? p - 64 : 0u);
}

// diagnostic build only (STAMP = true): wave 0 / lane 0 accumulates s_memtime deltas per phase
__device__ unsigned long long g_lz_stamps[8];

template <bool STAMP>
__global__ __launch_bounds__(LZ_THREADS)
void k_lz(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, uint64_t *__restrict__ seqs,
          uint8_t *__restrict__ lits, BlkInfo *__restrict__ blk, uint4 *__restrict__ ctab, uint32_t flags, uint32_t max_off, uint32_t max_len) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *win32   = (uint32_t *)(lds + L_WIN);
    uint32_t *table   = (uint32_t *)(lds + L_TABLE);
    uint16_t *len_arr = (uint16_t *)(lds + L_LEN);
    uint16_t *off_arr = (uint16_t *)(lds + L_OFF);
    uint16_t *fix_arr = (uint16_t *)(lds + L_FIXLEN);
    WMeta    *wmeta   = (WMeta *)(lds + L_WMETA);
    WRes     *wres    = (WRes *)(lds + L_WRES);
    WPub     *wpub    = (WPub *)(lds + L_WPUB);

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = uni(tid >> 6);                  // tell the compiler it is wave-uniform: keeps the parse walks on the scalar unit
    const SegDesc sd = segs[blockIdx.x];
    const uint8_t *seg = src + sd.src_off;
    const uint32_t seg_len = sd.len;
    const uint32_t lazy = flags & F_LAZY;
    const bool force_fb = (flags & FLAG_FORCE_FALLBACK) != 0;
    const uint64_t lane_lt = ((uint64_t)1 << lane) - 1;   // lanes below this one
    const uint32_t wbase = wave * (64 * GROUPS_PER_WAVE); // tile-relative first position of this wave

    for (uint32_t i = tid; i < (1u << HASH_LOG); i += LZ_THREADS) table[i] = 0;
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    if (STAMP && tid == 0) st_prev = __builtin_amdgcn_s_memtime();
#define LZ_STAMP(k) do { if (STAMP && tid == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_prev; st_prev = t_; } } while (0)

    // initial window fill [0, TILE + LOOKAHEAD + 16); afterwards one TILE-sized chunk per tile: requested at the top of
    // tile t, stored into LDS before tile t's B3, first read after B4 (tile t+1's lookups).  The slots it overwrites hold
    // positions below t0 - 60400 < t0 - MAX_OFF, which no match of tile t can reference.
    uint32_t loaded_end = TILE + LOOKAHEAD + 16;
    for (uint32_t i = tid * 16; i < loaded_end; i += LZ_THREADS * 16) {
        const uint4 v = load_chunk(seg, i, seg_len);
        *(uint4 *)(lds + L_WIN + i) = v;
        if (i == 0) *(uint4 *)(lds + L_WIN + WIN_BYTES) = v;
    }
    __syncthreads();
    uint4 pf = make_uint4(0, 0, 0, 0);

    const uint32_t nblk = (seg_len + BLK_SIZE - 1) / BLK_SIZE;
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t blk_start = b * BLK_SIZE;
        const uint32_t blk_end = (seg_len - blk_start < BLK_SIZE) ? seg_len : blk_start + BLK_SIZE;
        const uint32_t gblk = sd.blk_base + b;
        uint64_t *bseq = seqs + (size_t)gblk * SEQ_CAP;
        uint8_t  *blit = lits + (size_t)gblk * BLK_SIZE;
        // block-level parse state, uniform across the workgroup
        uint32_t next_free = blk_start;      // first position not covered by an emitted match
        uint32_t seq_run = 0, lit_run = 0;   // sequences / literals emitted so far
        uint32_t g_last1 = 1;                // 1 + literal index at the most recent match (0 literals before the block start)

        for (uint32_t t0 = blk_start; t0 < blk_end; t0 += TILE) {
            const uint32_t t1 = (blk_end - t0 < TILE) ? blk_end : t0 + TILE;
            const uint32_t ext_lim = (t1 + LOOKAHEAD < blk_end) ? t1 + LOOKAHEAD : blk_end;

            // ---- request the next tile's window chunk (consumed before B3)
            if (tid < TILE / 16) { pf = (loaded_end + tid * 16 < seg_len) ? load_chunk(seg, loaded_end + tid * 16, seg_len) : make_uint4(0, 0, 0, 0); }
            LZ_STAMP(0);

#ifdef LZ_EXP_PAD
            // issue-model experiment: 128 extra independent SALU (flag 0x400) or VALU (flag 0x800) instructions per wave and tile
            if (flags & 0x400u) {
                uint32_t a0 = tid, a1 = 1, a2 = 2, a3 = 3;
                a0 = uni(a0);
#pragma unroll
                for (int k = 0; k < 32; k++) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1" : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3) :: "scc");
                if (a0 + a1 + a2 + a3 == 0x7FFFFFF0u) lits[0] = 1;
            }
            if (flags & 0x800u) {
                uint32_t a0 = tid, a1 = 1, a2 = 2, a3 = 3;
#pragma unroll
                for (int k = 0; k < 32; k++) asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                if (a0 + a1 + a2 + a3 == 0x7FFFFFF0u) lits[0] = 1;
            }
#endif
            // ---- lookup
            uint32_t q[2], lo[2], hi[2], hsh[2], tag[2], ent[2];
            bool hv[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                q[r] = t0 + wbase + 64 * r + lane;
                hv[r] = (q[r] < t1) && (q[r] + 8 <= seg_len);
                fetch8(win32, q[r], lo[r], hi[r]);
                const uint32_t h32 = lo[r] * 0x9E3779B1u + (hi[r] & 0xFFFFu) * 0x85EBCA6Bu;
                hsh[r] = h32 >> (32 - HASH_LOG);
                tag[r] = (h32 >> (32 - HASH_LOG - TAG_BITS)) & TAG_MASK;
                ent[r] = hv[r] ? table[hsh[r]] : 0u;
            }
            __syncthreads();                                                        // B2
            LZ_STAMP(1);

            // ---- insert + match
            uint32_t len[2], off[2], flen[2];
            uint64_t effm[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (hv[r]) atomicMax(&table[hsh[r]], ((q[r] + 1) << TAG_BITS) | tag[r]);
                uint32_t l = 0, o = 0;
                const uint32_t c1 = ent[r] >> TAG_BITS;
                // a candidate whose tag differs hashed differently, so its first 6 bytes differ: no match possible
                if (c1 != 0 && (ent[r] & TAG_MASK) == tag[r]) {
                    uint32_t c = c1 - 1; o = q[r] - c;
                    if (o <= max_off) {
                        uint32_t lim = blk_end - q[r]; lim = lim < CAP1 ? lim : CAP1;
                        uint32_t clo, chi;
                        fetch8(win32, c, clo, chi);
                        uint64_t x = (uint64_t)(lo[r] ^ clo) | ((uint64_t)(hi[r] ^ chi) << 32);
                        if (x) l = ctz64(x) >> 3;
                        else {                                                      // bytes 8..15 (CAP1 == 16: one more step at most)
                            uint32_t alo, ahi;
                            fetch8(win32, q[r] + 8, alo, ahi);
                            fetch8(win32, c + 8, clo, chi);
                            x = (uint64_t)(alo ^ clo) | ((uint64_t)(ahi ^ chi) << 32);
                            l = x ? 8 + (ctz64(x) >> 3) : 16;
                        }
                        l = l < lim ? l : lim;
                        if (l < MIN_MATCH) l = 0;
                    }
                }
                len[r] = l; off[r] = o; flen[r] = l;
                const uint32_t nl = dpp_next_lane(l);                               // len of the next position (lane 63: 0)
                const bool eff = l >= MIN_MATCH && !(lazy && lane != 63 && (q[r] + 1 < t1) && nl > l);
                effm[r] = __ballot(eff);
            }
            LZ_STAMP(2);

            // ---- speculative parse of this wave's 128 positions (entry at its first position).  The scalar loop only
            // picks the match starts; coverage masks are rebuilt afterwards with one cross-lane gather per group
            // (the scalar unit is the bottleneck of this kernel, the LDS crossbar is not).
            uint64_t sel[2] = {0, 0}, cov[2];
            uint32_t cur = 0;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint32_t e0 = cur > 64u * r ? cur - 64u * r : 0u;             // positions covered by the previous group's last match
                uint64_t rem = e0 < 64 ? effm[r] & (~(uint64_t)0 << e0) : 0;
                uint32_t e_last = e0;
                const uint32_t endp = lane + len[r];                                // group-relative end of this position's match
                const uint64_t capm = __ballot(len[r] == CAP1);
                while (rem) {
                    const uint32_t s = ctz64(rem);
                    uint32_t e = rdlane(endp, s);
                    if ((capm >> s) & 1) {
                        const uint32_t qs = t0 + wbase + 64 * r + s;
                        const uint32_t L = lz_extend(win32, qs, qs - rdlane(off[r], s), (ext_lim - qs < max_len ? ext_lim - qs : max_len), lane);
                        if (lane == s) flen[r] = L;
                        e = s + L;
                    }
                    sel[r] |= (uint64_t)1 << s;
                    e_last = e;
                    const uint32_t ec = e < 64 ? e : 64u;                           // e >= s + MIN_MATCH, so ec - 1 is a valid shift
                    rem &= (~(uint64_t)0 << 1) << (ec - 1);
                }
                cur = 64 * r + (e_last > 64 ? e_last : 64u);
                // coverage (starts included): nearest selected start at or below the lane, its length via bpermute
                const uint64_t m_le = sel[r] & (lane_lt | ((uint64_t)1 << lane));
                const uint32_t sl = m_le ? 63 - clz64(m_le) : 0u;
                const uint32_t fs = (uint32_t)__shfl((int)flen[r], (int)sl);
                cov[r] = __ballot((m_le != 0 && lane < sl + fs) || lane < e0);
            }
            // positions the speculative parse stands on: everything except match interiors (starts are stood on;
            // positions past the tile end count as stood on so a walk stops there)
            const uint64_t vis[2] = {~cov[0] | sel[0], ~cov[1] | sel[1]};
            LZ_STAMP(7);
            const uint32_t flen_spec[2] = {flen[0], flen[1]};
            if (tid < TILE / 16) {
                const uint32_t wo = (loaded_end + tid * 16) & (WIN_BYTES - 1);
                *(uint4 *)(lds + L_WIN + wo) = pf;
                if (wo == 0) *(uint4 *)(lds + L_WIN + WIN_BYTES) = pf;
            }
            loaded_end += TILE;
            if (lane == 0) {
                WMeta m;
                m.sel[0] = sel[0]; m.sel[1] = sel[1]; m.vis[0] = vis[0]; m.vis[1] = vis[1];
                m.eff[0] = effm[0]; m.eff[1] = effm[1]; m.exit0 = cur; m.pad[0] = m.pad[1] = m.pad[2] = 0;
                wmeta[wave] = m;
            }
            __syncthreads();                                                        // B3
            LZ_STAMP(3);

            // ---- parallel resolution.  Carry chain over the speculative exits: a wave the carry already covers
            // passes it through, any other wave is assumed to re-synchronise and to leave at its speculative exit.
            const uint32_t c_in = next_free > t0 ? next_free - t0 : 0u;             // tile-relative
            const uint32_t ex_l = lane < LZ_WAVES ? lane * 128 + wmeta[lane & (LZ_WAVES - 1)].exit0 : 0u;
            uint32_t A = c_in;
            for (uint32_t j = 0; j < wave; j++) if (A < j * 128 + 128) A = rdlane(ex_l, j);
            uint32_t cw = A > wbase ? A - wbase : 0u;                               // wave-relative entry
            uint64_t fix[2] = {0, 0}, fcov[2] = {0, 0};
            uint32_t e_sync = 128;
            bool synced = false;
            {
                // walk from the true entry until the parse stands on a position the speculative parse also stood on;
                // one copy of the loop per 64-position group keeps the masks and the readlane sources fixed inside it
                uint32_t e = cw;
#pragma unroll
                for (int r1 = 0; r1 < 2; r1++) {
                    const uint64_t V = ~cov[r1] | sel[r1], E = effm[r1];            // positions the speculative parse stood on; match starts
                    while (!synced && e < 64u * (r1 + 1)) {
                        const uint32_t bp = e - 64u * r1;
                        const uint64_t vs = V >> bp, es = E >> bp;
                        const uint32_t nv = vs ? ctz64(vs) : 64u, ne = es ? ctz64(es) : 64u;
                        if (nv <= ne) {
                            if (!vs) { e = 64u * (r1 + 1); break; }
                            e += nv; synced = true; e_sync = e; break;
                        }
                        const uint32_t b2 = bp + ne;
                        e += ne;
                        uint32_t L = rdlane(len[r1], b2);
                        if (L == CAP1) {
                            const uint32_t qs = t0 + wbase + e;
                            L = lz_extend(win32, qs, qs - rdlane(off[r1], b2), (ext_lim - qs < max_len ? ext_lim - qs : max_len), lane);
                            if (lane == b2) flen[r1] = L;
                        }
                        fix[r1] |= (uint64_t)1 << b2;
                        cover(fcov[0], fcov[1], e, L);
                        e += L;
                    }
                }
                // true exit of this wave (tile-relative) under the assumption; published for the next tile's carry
                uint32_t my_exit = cw >= 128 ? A : (synced ? wbase + cur : wbase + e);
                // a wave that neither is covered nor re-synchronises breaks the assumption made by its successors
                const bool bad_w = !synced && cw < 128;
                // ---- masks of the final selection
                uint64_t fsel[2], litm[2];
                uint32_t nsel0, nlit0;
                const uint32_t in0 = t1 > t0 + wbase ? t1 - (t0 + wbase) : 0u;           // in-range positions of the wave
                auto masks_and_publish = [&](const uint64_t s0m, const uint64_t s1m, const uint64_t c0m, const uint64_t c1m, uint32_t badflag) {
                    const uint64_t keep0 = e_sync >= 64 ? 0 : ~mlow(e_sync), keep1 = e_sync >= 128 ? 0 : (e_sync <= 64 ? ~(uint64_t)0 : ~mlow(e_sync - 64));
                    fsel[0] = (s0m & keep0) | fix[0]; fsel[1] = (s1m & keep1) | fix[1];
                    litm[0] = mlow(in0) & ~((c0m & keep0) | fcov[0] | mlow(cw));
                    litm[1] = mlow(in0 > 64 ? in0 - 64 : 0u) & ~((c1m & keep1) | fcov[1] | mlow(cw > 64 ? cw - 64 : 0u));
                    nsel0 = (uint32_t)__popcll(fsel[0]);
                    const uint32_t nsel = nsel0 + (uint32_t)__popcll(fsel[1]);
                    nlit0 = (uint32_t)__popcll(litm[0]);
                    const uint32_t nlit = nlit0 + (uint32_t)__popcll(litm[1]);
                    // local literal index of the wave's last match (+1), 0 when it has none; the same for its first match
                    // (needed by the chunk table only)
                    uint32_t gl = 0, gf = 0;
                    if (fsel[1]) { const uint32_t sp = 63 - clz64(fsel[1]); gl = 1 + nlit0 + (uint32_t)__popcll(litm[1] & mlow(sp)); }
                    else if (fsel[0]) { const uint32_t sp = 63 - clz64(fsel[0]); gl = 1 + (uint32_t)__popcll(litm[0] & mlow(sp)); }
                    if (ctab) {
                        if (fsel[0]) { const uint32_t sp = ctz64(fsel[0]); gf = 1 + (uint32_t)__popcll(litm[0] & mlow(sp)); }
                        else if (fsel[1]) { const uint32_t sp = ctz64(fsel[1]); gf = 1 + nlit0 + (uint32_t)__popcll(litm[1] & mlow(sp)); }
                    }
                    if (lane == 0) { WPub p; p.cnt = nsel | (nlit << 16); p.gl = gl | (gf << 16); p.bad = badflag; p.exit = my_exit; wpub[wave] = p; }
                };
                masks_and_publish(sel[0], sel[1], cov[0], cov[1], bad_w ? 1u : 0u);
                LZ_STAMP(5);
                __syncthreads();                                                    // B4
                LZ_STAMP(4);
                WPub pl = wpub[lane & (LZ_WAVES - 1)];
                const bool lv = lane < LZ_WAVES;
                if (__ballot(lv && pl.bad) != 0 || force_fb) {
                    // ---- rare: some wave broke the assumption.  Wave 0 resolves the true entry of every wave serially
                    // (scalar code); everything is re-derived from the LDS records so that the common path above keeps no
                    // state alive for it.
                    len_arr[wbase + lane] = (uint16_t)len[0]; len_arr[wbase + 64 + lane] = (uint16_t)len[1];
                    off_arr[wbase + lane] = (uint16_t)off[0]; off_arr[wbase + 64 + lane] = (uint16_t)off[1];
                    __syncthreads();
                    if (wave == 0) {
                        const WMeta m = wmeta[lane & (LZ_WAVES - 1)];
                        uint32_t c = c_in;
                        for (uint32_t w = 0; w < LZ_WAVES; w++) {
                            const uint32_t base = w * 128;
                            const uint32_t cw2 = c > base ? c - base : 0u;
                            const uint64_t vis0 = rdlane64(m.vis[0], w), vis1 = rdlane64(m.vis[1], w);
                            const uint64_t eff0 = rdlane64(m.eff[0], w), eff1 = rdlane64(m.eff[1], w);
                            const uint32_t exit0 = rdlane(m.exit0, w);
                            uint64_t fx0 = 0, fx1 = 0, fc0 = 0, fc1 = 0;
                            uint32_t e2 = cw2;
                            bool sy = false;
                            while (e2 < 128) {
                                const uint32_t bp = e2 & 63;
                                const uint64_t visr = e2 < 64 ? vis0 : vis1, effr = e2 < 64 ? eff0 : eff1;
                                if ((visr >> bp) & 1) { sy = true; break; }
                                if ((effr >> bp) & 1) {
                                    const uint32_t pq = base + e2;
                                    uint32_t L = uni(len_arr[pq]);
                                    const uint32_t o = uni(off_arr[pq]);
                                    const uint32_t qs = t0 + pq;
                                    if (L == CAP1) L = lz_extend(win32, qs, qs - o, (ext_lim - qs < max_len ? ext_lim - qs : max_len), lane);
                                    if (lane == 0) fix_arr[pq] = (uint16_t)L;
                                    if (e2 < 64) fx0 |= (uint64_t)1 << bp; else fx1 |= (uint64_t)1 << bp;
                                    cover(fc0, fc1, e2, L);
                                    e2 += L;
                                } else e2 += 1;
                            }
                            c = sy ? base + exit0 : base + e2;
                            if (lane == 0) {
                                WRes rr; rr.fix[0] = fx0; rr.fix[1] = fx1; rr.fcov[0] = fc0; rr.fcov[1] = fc1;
                                rr.carry = cw2; rr.sync = sy ? e2 : 128u; rr.exit = c; rr.pad = 0;
                                wres[w] = rr;
                            }
                        }
                    }
                    __syncthreads();
                    {
                        const WRes rr = wres[wave];
                        const WMeta mm = wmeta[wave];
                        fix[0] = rr.fix[0]; fix[1] = rr.fix[1]; fcov[0] = rr.fcov[0]; fcov[1] = rr.fcov[1];
                        cw = rr.carry; e_sync = rr.sync;
                        my_exit = wres[LZ_WAVES - 1].exit;                           // every wave publishes the tile exit
                        flen[0] = ((fix[0] >> lane) & 1) ? (uint32_t)fix_arr[wbase + lane] : flen_spec[0];
                        flen[1] = ((fix[1] >> lane) & 1) ? (uint32_t)fix_arr[wbase + 64 + lane] : flen_spec[1];
                        // speculative selection and coverage back from the record: vis = ~cov | sel and sel is a subset of cov
                        masks_and_publish(mm.sel[0], mm.sel[1], ~mm.vis[0] | mm.sel[0], ~mm.vis[1] | mm.sel[1], 0u);
                    }
                    __syncthreads();
                    pl = wpub[lane & (LZ_WAVES - 1)];
                }
                // ---- 16-lane DPP scans over the waves' records
                uint32_t seq_base, lit_base, glast1_before;
                {
                    const uint32_t incl = row_scan_add(lv ? pl.cnt : 0u), excl = incl - (lv ? pl.cnt : 0u);
                    const uint32_t gabs = (lv && pl.gl) ? lit_run + (excl >> 16) + (pl.gl & 0xFFFF) : 0u;   // 1 + literal index of wave j's last match
                    const uint32_t gmax = row_scan_max(gabs);
                    const uint32_t ex_w = rdlane(excl, wave), tot = rdlane(incl, LZ_WAVES - 1);
                    seq_base = seq_run + (ex_w & 0xFFFF); lit_base = lit_run + (ex_w >> 16);
                    const uint32_t gb = wave ? rdlane(gmax, wave - 1) : 0u;
                    glast1_before = gb > g_last1 ? gb : g_last1;
                    const uint32_t ga = rdlane(gmax, LZ_WAVES - 1);
                    g_last1 = ga > g_last1 ? ga : g_last1;
                    if (ctab) {
                        // chunk table: state of the block's sequence / literal streams at this tile's start and the
                        // literal index of the tile's first match (lets later stages split a block by tiles)
                        const uint64_t hm = __ballot(lv && pl.gl);
                        uint32_t g_first = lit_run + (tot >> 16);
                        if (hm) { const uint32_t j0 = ctz64(hm); g_first = lit_run + (rdlane(excl, j0) >> 16) + (rdlane(pl.gl, j0) >> 16) - 1; }
                        if (tid == 0) ctab[(size_t)gblk * (BLK_SIZE / TILE) + (t0 - blk_start) / TILE] = make_uint4(seq_run, lit_run, g_first, 0u);
                    }
                    seq_run += tot & 0xFFFF; lit_run += tot >> 16;
                    next_free = t0 + rdlane(pl.exit, LZ_WAVES - 1);
                }

                // ---- emission: all indices come from popcounts of the masks (no cross-lane data movement)
                const uint32_t sp0 = fsel[0] ? 63 - clz64(fsel[0]) : 0u;
                const uint32_t tail0 = fsel[0] ? (uint32_t)__popcll(litm[0] & ~mlow(sp0 + 1)) : 0u;   // literals of group 0 after its last match
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const uint32_t lb = (r ? nlit0 : 0u) + (uint32_t)__popcll(litm[r] & lane_lt);     // literals of the wave before this lane
                    if ((fsel[r] >> lane) & 1) {
                        const uint64_t pm = fsel[r] & lane_lt;
                        uint32_t ll;
                        if (pm) { const uint32_t sp = 63 - clz64(pm); ll = (uint32_t)__popcll(litm[r] & lane_lt & ~mlow(sp + 1)); }
                        else if (r == 1 && fsel[0]) ll = tail0 + (uint32_t)__popcll(litm[1] & lane_lt);
                        else ll = lit_base + lb - (glast1_before - 1);
                        const uint32_t idx = seq_base + (uint32_t)__popcll(pm) + (r ? nsel0 : 0u);
                        if (idx < SEQ_CAP) bseq[idx] = seq_pack(ll, flen[r], off[r]);
                    }
                    if ((litm[r] >> lane) & 1) { const uint32_t li = lit_base + lb; if (li < BLK_SIZE) blit[li] = (uint8_t)lo[r]; }
                }
            }
            LZ_STAMP(6);
        } // tiles
        if (tid == 0) { blk[gblk].nseq = seq_run; blk[gblk].nlit = lit_run; }
    } // blocks
    if (STAMP && tid == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_lz_stamps[k], st_acc[k]);
}

void launch_lz(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
               uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_lz<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_TOTAL);
        (void)hipFuncSetAttribute((const void *)k_lz<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_TOTAL);
        attr_set = true;
    }
    if (flags & FLAG_STAMP) hipLaunchKernelGGL(k_lz<true>, dim3(nseg), dim3(LZ_THREADS), L_TOTAL, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len);
    else hipLaunchKernelGGL(k_lz<false>, dim3(nseg), dim3(LZ_THREADS), L_TOTAL, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len);
}

// diagnostic: read and clear the phase stamps (cycles summed over workgroups)
void lz_read_stamps(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lz_stamps), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz_stamps), z, sizeof(z));
}

} // namespace pna
