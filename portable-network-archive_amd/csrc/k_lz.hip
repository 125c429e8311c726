// k_lz.hip -- LZ77 match finding + greedy parse for gfx950 (CDNA4), one 1024-thread workgroup per segment.
//
// Replaces the match finder inside the third-party encoder the reference drives at
// lib/src/compress.rs:32-41 (CompressionWriter::write -> ZstdEncoder::write).  Integer/byte work, no MFMA.
//
// LDS (one workgroup per CU, 163 792 of the 163 840 bytes):
//   win   [65536 + 16 B]  circular copy of the segment's most recent 64 KiB (look-back + 1 KiB look-ahead); the first
//         16 bytes are mirrored behind the end so that unaligned reads never wrap
//   table [24512 x u32] hash table (what fits next to the window): (position+1) << 11 | 11-bit tag of the latest occurrence (0 = empty);
//         inserts are ds_max_u32, the tag lets a lookup skip candidates whose 6 bytes cannot match
//   per-wave records (end of the wave's last match; counts)
// Per tile of 1024 G positions (G per lane; both codecs run G = 4: tiles of 4 096):
//   (next window chunk requested into registers) lookup -> match (the tile's inserts wait until every wave has looked up: they
//   go behind B3, so lookups and inserts need no barrier of their own) +
//   REGION-LOCAL parse: every wave parses its own 64 G positions greedily from max(its first position, the tile's carry)
//   with scalar loops on ballot masks, and publishes the end of its last match -> B3 -> inserts (ds_max_u32) -> MERGE: the running end E of the
//   earlier waves' matches is a 16-lane prefix maximum (exact unless an end falls 1-2 bytes behind E: then a short serial
//   scan); a wave entirely below E emits nothing, matches that end before E are dropped, the one straddling E is cut from
//   the front (>= 3 bytes must remain),
//   everything else stands -> selection / literal masks -> counts -> B4 -> 16-lane DPP scans of the waves' counts ->
//   emission of sequences and literals straight to HBM.
//   Cross-lane traffic on this path: ballots, readlanes, DPP and two ds_bpermute per 64 positions.
// Look-back beyond the LDS window: the table keeps positions of the whole 1 MiB segment (20 bits + 1), a candidate more than
//   NEAR_OFF bytes back ("far") is verified against the segment in HBM / L2 -- its 16 + 4 bytes are requested right behind the
//   table look-ups and consumed after the near candidates of the tile went through the LDS path.
// Table load: only EVEN positions are inserted (half the pressure on 24 512 slots); a match that is therefore found one or
//   two positions late is moved back to its true start by BACKWARD ADOPTION: every match knows how many bytes (<= 3) before it
//   also agree with its candidate, and two DPP rounds (lane + 1, then lane + 2) let a position take over its right
//   neighbours' matches, one / two bytes longer.
#include <hip/hip_runtime.h>
#include "pna_dev.h"

#ifdef LZ_EXP_ALLINS
#define LZ_INS_COND true
#else
#define LZ_INS_COND (ins_all || !(lane & 1))
#endif

namespace pna {

constexpr uint32_t TAG_BITS = 11, TAG_MASK = (1u << TAG_BITS) - 1;

// LDS layout (byte offsets into the dynamic shared array)
constexpr uint32_t L_WIN    = 0;
constexpr uint32_t WIN_MIRROR = 48;                        // the window's first 48 bytes again behind its end: unaligned reads never wrap (k_lz reads 16 past a position, k_lzm 36 past a lane's first)
constexpr uint32_t L_TABLE  = L_WIN + WIN_BYTES + WIN_MIRROR;
constexpr uint32_t L_WEND   = L_TABLE + 4u * HASH_ENTRIES;   // 16 x u32: tile-relative end of each wave's last match (0 = none)
constexpr uint32_t L_WPUB   = L_WEND + 4 * LZ_WAVES;        // 16 x 8 B
constexpr uint32_t L_TOTAL  = L_WPUB + 8 * LZ_WAVES;

struct WPub  { uint32_t cnt; uint32_t gl; };   // cnt = nsel | nlit << 16; gl = (local literal index of the LAST match + 1) | (same for the FIRST match) << 16, 0 = no match
static_assert(sizeof(WPub) == 8, "LDS record size");
static_assert(L_TOTAL <= 160 * 1024 && HASH_ENTRIES % 4 == 0 && L_TABLE % 16 == 0, "k_lz's LDS: window + table + records within one CU's 160 KiB");

constexpr uint32_t FLAG_SPLIT_WAVEPARSE = 0x1000u;   // split form: the parse half as k_lz<MODE = 2> (a wave per region) instead of k_lzp (testing)
constexpr uint32_t FLAG_STAMP = 0x100u, FLAG_FORCE_SERIAL = 0x200u;   // 0x200: always take the serial form of the end scan (testing)
static_assert(CAP1 >= 16 && CAP1 % 16 == 0 && CAP1 <= 32 && BACK_CAP == 3 && MIN_MATCH > 3, "the match step compares 16 bytes at a time, the next 16 only where all before matched");
static_assert(GROUPS_PER_WAVE == 2 && TILE == 2048, "TILE / GROUPS_PER_WAVE describe the G = 2 (deflate) form; k_lz itself is generic in G");

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
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

struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };   // 16 bytes at any byte address
typedef uint32_t u32u __attribute__((aligned(1)));

// Wave-cooperative extension of a match whose first L0 bytes are known to agree: q, c, L0, lim are wave-uniform; returns the
// full length (<= lim).  64 lanes x 4 bytes per step.  FARC: the candidate lies outside the LDS window, its bytes come from the
// segment in HBM / L2 (c + lim < q, so every address is inside the segment).
template <bool FARC>
__device__ __forceinline__ uint32_t lz_extend(const uint32_t *win32, const uint8_t *seg, uint32_t q, uint32_t c, uint32_t L0, uint32_t lim, uint32_t lane) {
    uint32_t L = L0;
    for (;;) {
        uint32_t pos = L + lane * 4;
        const uint32_t cw = FARC ? *(const u32u *)(seg + c + pos) : fetch4(win32, c + pos);
        uint32_t x = fetch4(win32, q + pos) ^ cw;
        uint32_t nb = x ? ((uint32_t)__builtin_ctz(x) >> 3) : 4u;
        uint32_t room = lim > pos ? lim - pos : 0u;
        nb = nb < room ? nb : room;
        uint64_t bad = __ballot(nb < 4u);
        if (bad) { uint32_t f = ctz64(bad); L += 4 * f + rdlane(nb, f); break; }
        L += 256;
    }
    return L;
}

// the same with both sides read from the segment in memory (the parse half of the split form has no window)
__device__ __forceinline__ uint32_t lz_extend_mem(const uint8_t *seg, uint32_t seg_len, uint32_t q, uint32_t c, uint32_t L0, uint32_t lim, uint32_t lane) {
    uint32_t L = L0;
    for (;;) {
        const uint32_t pos = L + lane * 4;
        uint32_t nb = 0;
        const uint32_t room = lim > pos ? lim - pos : 0u;
        if (room) {                                                                 // q + pos < q + lim <= the block's end: inside the segment
            if (q + pos + 4 <= seg_len) {
                const uint32_t x = *(const u32u *)(seg + q + pos) ^ *(const u32u *)(seg + c + pos);
                nb = x ? ((uint32_t)__builtin_ctz(x) >> 3) : 4u;
            } else {
                while (nb < room && seg[q + pos + nb] == seg[c + pos + nb]) nb++;
            }
            nb = nb < room ? nb : room;
        }
        const uint64_t bad = __ballot(nb < 4u);
        if (bad) { const uint32_t f = ctz64(bad); L += 4 * f + rdlane(nb, f); break; }
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

// diagnostic build only (STAMP = true): lane 0 of every wave accumulates s_memtime deltas per phase (sums over the 16 waves)
__device__ unsigned long long g_lz_stamps[8];

// G = positions per lane and tile (groups of 64 positions per wave): the wave's G groups are ONE parse region of 64 G positions, a tile is
// 1024 G positions (zstd: LZ_G_ZSTD, deflate: LZ_G_DEFLATE; the deflate chunk table keeps one entry per 2 KiB).
// CT: the launch fills the deflate chunk table (a template parameter so that the zstd instance carries none of that code).
// STRONG: the parameter set of the high levels (third adoption round over 7 back bytes, two-step lazy deferral) as an instance of its own,
// so that the default instance carries none of its loads and branches (as run-time switches they cost it 3.3 %).
// MODE: 0 = the whole stage in one kernel; 1 / 2 = its two halves as kernels of their own (launch_lz_split): 1 = look-up, match and
// inserts only -- one word per position (length | offset << 6, after adoption) goes to `pbuf` --, 2 = parse, merge and emission from those
// words (no window, no table: 192 bytes of LDS, two workgroups per CU).  Same code, same results: the halves only meet in `pbuf`.
template <bool STAMP, int G, bool CT, bool STRONG, int MODE>
__global__ __launch_bounds__(LZ_THREADS, MODE == 2 ? 8 : 4)   // (second figure: waves per SIMD the compiler must leave room for)
void k_lz(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, uint64_t *__restrict__ seqs,
          uint8_t *__restrict__ lits, BlkInfo *__restrict__ blk, uint4 *__restrict__ ctab, uint32_t flags, uint32_t max_off, uint32_t max_len,
          uint32_t *__restrict__ pbuf, uint32_t blk0) {
    constexpr uint32_t RW = 64u * G;                       // positions one wave owns = the parse region
    constexpr uint32_t TILE_G = RW * LZ_WAVES;             // positions per synchronous step
    constexpr uint32_t NONE = 0xFFFFFFFFu;
#ifdef LZ_EXP_NOFAR
    constexpr bool FAR = false; max_off = max_off < NEAR_OFF ? max_off : NEAR_OFF;   // timing experiment: no look-back beyond the LDS window
#else
    constexpr bool FAR = !CT;                               // deflate offsets (<= 32 KiB) never leave the LDS window
#endif
    constexpr uint32_t NEAR = G == 2 ? MAX_OFF_G2 : NEAR_OFF;
    static_assert(TILE_G % TILE == 0 && WIN_BYTES >= 2 * TILE_G + LOOKAHEAD + 16 + (G == 2 ? MAX_OFF_G2 : NEAR_OFF), "window: look-back + this tile + look-ahead + the chunk in flight");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *win32   = (uint32_t *)(lds + L_WIN);
    uint32_t *table   = (uint32_t *)(lds + L_TABLE);
    uint32_t *wend    = (uint32_t *)(lds + (MODE == 2 ? 0u : L_WEND));
    WPub     *wpub    = (WPub *)(lds + (MODE == 2 ? 4u * LZ_WAVES : L_WPUB));

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = uni(tid >> 6);                  // tell the compiler it is wave-uniform: keeps the parse walks on the scalar unit
    const SegDesc sd = segs[blockIdx.x];
    const uint8_t *seg = src + sd.src_off;
    const uint32_t seg_len = sd.len;
    uint32_t *pb = MODE ? pbuf + (size_t)(sd.blk_base - blk0) * BLK_SIZE : nullptr;   // the segment's words (split form)
    const uint32_t lazy = flags & F_LAZY;
    const bool adopt = (flags & F_ADOPT) != 0, ins_all = !(flags & F_INS2);
    constexpr bool strong = STRONG;   // (wave-uniform) level sets: pna_host.cpp level_flags()
    const bool force_serial = (flags & FLAG_FORCE_SERIAL) != 0;
    const uint64_t lane_lt = ((uint64_t)1 << lane) - 1;   // lanes below this one
    const uint32_t wbase = wave * RW;                     // tile-relative first position of this wave

    if (MODE != 2) for (uint32_t i = tid; i < HASH_ENTRIES / 4; i += LZ_THREADS) ((uint4 *)table)[i] = make_uint4(0, 0, 0, 0);   // 16 bytes per store
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    if (STAMP && lane == 0) st_prev = __builtin_amdgcn_s_memtime();
#define LZ_STAMP(k) do { if (STAMP && lane == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_prev; st_prev = t_; } } while (0)

    // initial window fill [0, TILE_G + LOOKAHEAD + 16); afterwards one tile-sized chunk per tile: requested at the top of
    // tile t, stored into LDS before tile t's B3, first read after B4 (tile t+1's lookups).  The slots it overwrites hold
    // positions below t0 + 2 TILE_G + LOOKAHEAD + 16 - 64 Ki <= t0 - max_off, which no match of tile t can reference.
    uint32_t loaded_end = TILE_G + LOOKAHEAD + 16;
    if (MODE != 2) for (uint32_t i = tid * 16; i < loaded_end; i += LZ_THREADS * 16) {
        const uint4 v = load_chunk(seg, i, seg_len);
        *(uint4 *)(lds + L_WIN + i) = v;
        if (i < WIN_MIRROR) *(uint4 *)(lds + L_WIN + WIN_BYTES + i) = v;
    }
    if (MODE != 2) __syncthreads();
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

        for (uint32_t t0 = blk_start; t0 < blk_end; t0 += TILE_G) {
            const uint32_t t1 = (blk_end - t0 < TILE_G) ? blk_end : t0 + TILE_G;
            const uint32_t ext_lim = (t1 + LOOKAHEAD < blk_end) ? t1 + LOOKAHEAD : blk_end;

            // ---- request the next tile's window chunk (consumed before B3)
            if (MODE != 2 && tid < TILE_G / 16) { pf = (loaded_end + tid * 16 < seg_len) ? load_chunk(seg, loaded_end + tid * 16, seg_len) : make_uint4(0, 0, 0, 0); }
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
            uint32_t q[G], lo[G], hi[G], hsh[G], tag[G], ent[G], bq[G], bq2[G];
            bool hv[G];
            // (uniform) a tile that lies wholly inside the block and at least 8 bytes before the segment end needs no per-lane range checks
            const bool tile_full = (t1 - t0 == TILE_G) && (t0 + TILE_G + 8 <= seg_len);
            uint32_t pv[G];                                                         // MODE 2: the positions' words
#pragma unroll
            for (int r = 0; r < G; r++) {
                q[r] = t0 + wbase + 64 * r + lane;
                hv[r] = tile_full || ((q[r] < t1) && (q[r] + 8 <= seg_len));
                if constexpr (MODE == 2) {
                    const bool in = q[r] < t1;
                    pv[r] = in ? pb[q[r]] : 0u;
                    lo[r] = in ? seg[q[r]] : 0u;                                    // the literal byte
                    hi[r] = hsh[r] = tag[r] = ent[r] = bq[r] = bq2[r] = 0;
                    continue;
                }
                {
                    const uint32_t *p = win32 + ((q[r] & (WIN_BYTES - 1)) >> 2);    // p[1], p[2] may lie in the mirror
                    const uint32_t sh = (q[r] & 3) * 8;
                    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], dm = win32[((q[r] - 4) & (WIN_BYTES - 1)) >> 2];
                    lo[r] = __builtin_amdgcn_alignbit(d1, d0, sh);
                    hi[r] = __builtin_amdgcn_alignbit(d2, d1, sh);
                    bq[r] = __builtin_amdgcn_alignbit(d0, dm, sh);                  // the 4 bytes before q (q - 1 in the top byte)
                    bq2[r] = 0;
                    if (strong) bq2[r] = __builtin_amdgcn_alignbit(dm, win32[((q[r] - 8) & (WIN_BYTES - 1)) >> 2], sh);   // (uniform) and the 4 before those
                }
                const uint32_t h32 = lo[r] * 0x9E3779B1u + (hi[r] & 0xFFFFu) * 0x85EBCA6Bu;
                hsh[r] = __umulhi(h32, HASH_ENTRIES);                               // floor(h32 * entries / 2^32): any table size
                tag[r] = (h32 >> 6) & TAG_MASK;                                     // a filter only: any function of the hash will do
                ent[r] = hv[r] ? table[hsh[r]] : 0u;
            }
            // ---- candidates: offset (0 = none: empty slot, foreign tag -- a candidate whose tag differs hashed differently, so its first
            // 6 bytes differ --, position below 8, beyond max_off).  Far candidates (beyond the LDS window) get the 4 bytes before and the
            // 16 bytes at the candidate requested from the segment now; the other lanes read the segment's first bytes (one line, no
            // exec masking), which nobody looks at.  A usable candidate lies at position >= 8, so the loads never reach below the segment.
            uint32_t off[G];
            U4u fa[G]; uint32_t fb[G], fc[G];
#pragma unroll
            for (int r = 0; r < G; r++) {
                if constexpr (MODE == 2) { off[r] = pv[r] >> 6; continue; }
                fa[r].x = fa[r].y = fa[r].z = fa[r].w = fb[r] = fc[r] = 0;
                const uint32_t c1 = ent[r] >> TAG_BITS, o = q[r] + 1 - c1;
                off[r] = (c1 > 8 && (ent[r] & TAG_MASK) == tag[r] && o <= max_off) ? o : 0u;
                if (FAR && seg_len > NEAR && max_off > NEAR) {                      // (uniform) shorter segments / near-only levels have no far candidates
                    const uint32_t fo = off[r] > NEAR ? c1 - 5 : 0u;                // byte offset of c - 4 in the segment
                    fa[r] = *(const U4u *)(seg + fo);
                    fb[r] = *(const u32u *)(seg + fo + 16);
                    if (strong) fc[r] = *(const u32u *)(seg + (off[r] > NEAR ? fo - 4 : 0u));   // (uniform) bytes c - 8 .. c - 5
                }
            }
            LZ_STAMP(1);

            // ---- match (the inserts of this tile wait behind B3)
            uint32_t len[G], flen[G];
            uint64_t effm[G];
            auto do_match = [&](const int r) __attribute__((always_inline)) {
                uint32_t l = 0, bk = 0;                                             // bk = bytes before q and c that agree as well (<= 3; strong set: <= 7)
                const uint32_t o = off[r];
                if constexpr (MODE == 2) l = pv[r] & 63u;
                else {
                if (o != 0) {
                    const uint32_t c = q[r] - o;
                    const bool isfar = FAR && o > NEAR;
                    const bool edge = blk_end - (t0 + wbase + 64u * r) < 64u + CAP1;       // (uniform) only the block's last groups can run into its end
                    // all 16 bytes at once: in a wave of 64 candidates some lane nearly always needs bytes 8..15, so a
                    // two-step form pays for both steps plus the exec-mask juggling between them (-0.7 %)
                    const uint32_t *pq = win32 + ((q[r] & (WIN_BYTES - 1)) >> 2);
                    const uint32_t shc = (c & 3) * 8, shq = (q[r] & 3) * 8;
                    uint32_t w0, w1, w2, w3, bc, bc2 = 0;                           // 16 bytes at c, the 4 (strong: 8) bytes before c
                    if (isfar) { bc = fa[r].x; w0 = fa[r].y; w1 = fa[r].z; w2 = fa[r].w; w3 = fb[r]; bc2 = fc[r]; }
                    else {
                        const uint32_t *pc = win32 + ((c & (WIN_BYTES - 1)) >> 2);
                        const uint32_t d0 = pc[0], d1 = pc[1], d2 = pc[2], d3 = pc[3], d4 = pc[4], dm = win32[((c - 4) & (WIN_BYTES - 1)) >> 2];
                        w0 = __builtin_amdgcn_alignbit(d1, d0, shc); w1 = __builtin_amdgcn_alignbit(d2, d1, shc);
                        w2 = __builtin_amdgcn_alignbit(d3, d2, shc); w3 = __builtin_amdgcn_alignbit(d4, d3, shc);
                        bc = __builtin_amdgcn_alignbit(d0, dm, shc);
                        if (strong) bc2 = __builtin_amdgcn_alignbit(dm, win32[((c - 8) & (WIN_BYTES - 1)) >> 2], shc);
                    }
                    const uint32_t e2 = pq[2], e3 = pq[3], e4 = pq[4];
                    const uint32_t x0 = lo[r] ^ w0, x1 = hi[r] ^ w1;
                    const uint32_t x2 = __builtin_amdgcn_alignbit(e3, e2, shq) ^ w2;
                    const uint32_t x3 = __builtin_amdgcn_alignbit(e4, e3, shq) ^ w3;
                    const uint64_t xa = (uint64_t)x0 | ((uint64_t)x1 << 32), xb = (uint64_t)x2 | ((uint64_t)x3 << 32);
                    l = xa ? ctz64(xa) >> 3 : (xb ? 8 + (ctz64(xb) >> 3) : 16);
#pragma unroll
                    for (uint32_t k16 = 16; k16 < CAP1; k16 += 16) {
                        if (l == k16) {
                            // the next 16 bytes, only for the lanes where everything before matched (same alignment as above): most capped
                            // matches end here, which keeps them off the wave-cooperative extension in the parse loop.  Far candidates
                            // fetch theirs from the segment now (rare: a few lanes per tile, and the line is usually still in L1 / L2)
                            const uint32_t *pq2 = win32 + (((q[r] + k16) & (WIN_BYTES - 1)) >> 2);
                            const uint32_t g0 = pq2[0], g1 = pq2[1], g2 = pq2[2], g3 = pq2[3], g4 = pq2[4];
                            uint32_t v0, v1, v2, v3;
                            if (isfar) { const U4u t = *(const U4u *)(seg + c + k16); v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w; }
                            else {
                                const uint32_t *pc2 = win32 + (((c + k16) & (WIN_BYTES - 1)) >> 2);
                                const uint32_t f0 = pc2[0], f1 = pc2[1], f2 = pc2[2], f3 = pc2[3], f4 = pc2[4];
                                v0 = __builtin_amdgcn_alignbit(f1, f0, shc); v1 = __builtin_amdgcn_alignbit(f2, f1, shc);
                                v2 = __builtin_amdgcn_alignbit(f3, f2, shc); v3 = __builtin_amdgcn_alignbit(f4, f3, shc);
                            }
                            const uint32_t y0 = __builtin_amdgcn_alignbit(g1, g0, shq) ^ v0, y1 = __builtin_amdgcn_alignbit(g2, g1, shq) ^ v1;
                            const uint32_t y2 = __builtin_amdgcn_alignbit(g3, g2, shq) ^ v2, y3 = __builtin_amdgcn_alignbit(g4, g3, shq) ^ v3;
                            const uint64_t ya = (uint64_t)y0 | ((uint64_t)y1 << 32), yb = (uint64_t)y2 | ((uint64_t)y3 << 32);
                            l = k16 + (ya ? ctz64(ya) >> 3 : (yb ? 8 + (ctz64(yb) >> 3) : 16));
                        }
                    }
                    if (edge) { const uint32_t lim = blk_end - q[r]; l = l < lim ? l : lim; }
                    if (l < MIN_MATCH) l = 0;
                    // bytes before q and c that agree as well, nearest first: the low byte forced to differ caps the count at BACK_CAP = 3
                    // (strong set: when all four agree, the four before them are counted the same way: at most 7)
                    const uint32_t xk = bq[r] ^ bc;
                    bk = (uint32_t)__builtin_clz(xk | 0xFFu) >> 3;
                    if (strong && xk == 0) bk = 4 + ((uint32_t)__builtin_clz((bq2[r] ^ bc2) | 0xFFu) >> 3);
                }
                // ---- backward adoption.  K = len << 6 | back << 3 | lanes the match was moved by.  A lane without a match may carry a stray
                // back count and adopt "lengths" of 1..3 from such neighbours: they stay below MIN_MATCH and nobody reads them as a match.
                uint32_t K = (l << 6) | (bk << 3);
                if (STRONG && !l) K = 0;                                            // (three rounds could lift a stray back count to a "length" of 7 >= MIN_MATCH; with two it stays below)
#ifndef LZ_EXP_NOADOPT
                if (adopt) {
                {   // round 1: the right neighbour's match, one byte longer (lane 63 sees 0)
                    const uint32_t K1 = dpp_next_lane(K), T = K1 + 57u;
                    K = ((K1 & 0x38u) != 0 && T > (K | 63u)) ? T : K;
                }
                {   // round 2: the match two lanes to the right (after round 1), two bytes longer
                    const uint32_t K2 = dpp_next_lane(dpp_next_lane(K)), T = K2 + 114u;
                    K = ((K2 & 0x30u) != 0 && T > (K | 63u)) ? T : K;
                }
                if (strong) {   // (uniform) round 3: four lanes to the right, four bytes longer
                    const uint32_t K4 = dpp_next_lane(dpp_next_lane(dpp_next_lane(dpp_next_lane(K)))), T = K4 + 228u;
                    K = ((K4 & 0x20u) != 0 && T > (K | 63u)) ? T : K;
                }
                l = K >> 6;
                off[r] = (uint32_t)__shfl((int)o, (int)(lane + (K & 7u)));          // the offset travels with the match
                }
#endif
                }
                len[r] = l; flen[r] = l;
                if constexpr (MODE == 1) { effm[r] = 0; return; }
                const uint32_t nl = dpp_next_lane(l);                               // len of the next position (lane 63: 0, so lane 63 never defers)
                // lazy deferral: position q waits iff q + 1 is still in the tile and holds a longer match.  nl > l with l < MIN_MATCH is harmless
                // (the position is no start anyway); q + 1 >= t1 only happens in a block's last, partial tile (nl is 0 there: lanes >= t1 hold no match)
                uint64_t longer = lazy ? __ballot(nl > l) : 0;
                if (lazy && strong) longer |= __ballot(dpp_next_lane(nl) > l + 1);     // (uniform) two-step deferral: q + 2 holds a match longer by two or more
                effm[r] = __ballot(l >= MIN_MATCH) & ~longer;
            };
            LZ_STAMP(2);

            // ---- region-local parse of this wave's 64 G positions, from the tile's carry if that reaches into them.  The
            // scalar loop only picks the match starts; coverage masks are rebuilt afterwards with one cross-lane gather
            // per group (the scalar unit is the bottleneck of this kernel, the LDS crossbar is not).
            const uint32_t c_in = next_free > t0 ? next_free - t0 : 0u;             // tile-relative carry
            uint64_t sel[G], cov[G];
#pragma unroll
            for (int r = 0; r < G; r++) sel[r] = 0;
            uint32_t cur = c_in > wbase ? (c_in - wbase < RW ? c_in - wbase : RW) : 0u;
            uint32_t el = 0;                                                        // wave-relative end of the last selected match
            auto do_parse = [&](const int r) __attribute__((always_inline)) {
                const uint32_t e0 = cur > 64u * r ? cur - 64u * r : 0u;             // positions covered by the carry / the previous group's last match
                uint64_t rem = e0 < 64 ? effm[r] & (~(uint64_t)0 << e0) : 0;
                uint32_t e_last = e0;
                const uint32_t endp = lane + len[r];                                // group-relative end of this position's match
                const uint64_t capm = __ballot(len[r] >= CAP1);
#ifdef LZ_EXP_VECTORPARSE     // measured: bit-exact, 128 SALU fewer but 132 VALU more per wave and tile, B3 wait 20.6 -> 16.2 %, and 4.8 % SLOWER: the VALU port binds
                if (rem != 0 && (capm & rem) == 0) {
                    // No match of the group needs the extension: the greedy walk s -> first usable start at or behind the end of s's match is a
                    // fixed jump function J, and the chosen starts are the orbit of the first usable start under J.  The orbit is marked by
                    // pointer doubling -- members push a mark along J (ds_permute), then J := J o J (ds_bpermute) -- in four rounds (a match is
                    // >= 6 long, so a group holds at most 11 starts <= 1 + 2 + 4 + 8): constant time per group instead of a scalar loop per chosen
                    // match, which is what made the waves of a tile finish their parse at different times.  J is kept times four (the permute
                    // address); "no further start" = 256 wraps to lane 0, which no jump can target (J >= 6), and lane 0 ignores what it receives.
                    const uint64_t shf = endp < 64 ? (effm[r] >> endp) : 0;
                    uint32_t J4 = shf ? (endp + ctz64(shf)) << 2 : 256u;
                    const uint32_t s0 = ctz64(rem);
                    uint32_t m = lane == s0 ? 1u : 0u;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t recv = (uint32_t)__builtin_amdgcn_ds_permute((int)(m ? J4 : 0u), 1);
                        m |= lane ? recv : 0u;
                        if (k < 3) { const uint32_t J2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)J4, (int)J4); J4 = J4 >= 256u ? 256u : J2; }
                    }
                    const uint64_t M = __ballot(m != 0);
                    sel[r] = M;
                    e_last = rdlane(endp, 63 - clz64(M));
                    rem = 0;
                }
#endif
                while (rem) {
                    const uint32_t s = ctz64(rem);
                    uint32_t e = rdlane(endp, s);
                    if ((capm >> s) & 1) {
                        const uint32_t qs = t0 + wbase + 64 * r + s, os = rdlane(off[r], s), L0 = e - s;
                        const uint32_t xl = ext_lim - qs < max_len ? ext_lim - qs : max_len;
                        #ifdef LZ_EXP_NOFAREXT
                        const uint32_t L = (FAR && os > NEAR) ? L0
#else
                        const uint32_t L = MODE == 2 ? lz_extend_mem(seg, seg_len, qs, qs - os, L0, xl, lane)
                                         : (FAR && os > NEAR) ? lz_extend<true>(win32, seg, qs, qs - os, L0, xl, lane)
#endif
                                                              : lz_extend<false>(win32, seg, qs, qs - os, L0, xl, lane);
                        if (lane == s) flen[r] = L;
                        e = s + L;
                    }
                    sel[r] |= (uint64_t)1 << s;
                    e_last = e;
                    const uint32_t ec = e < 64 ? e : 64u;                           // e >= s + MIN_MATCH, so ec - 1 is a valid shift
                    rem &= (~(uint64_t)0 << 1) << (ec - 1);
                }
                if (sel[r]) el = 64 * r + e_last;
                cur = 64 * r + (e_last > 64 ? e_last : 64u);
                // coverage (starts included): nearest selected start at or below the lane, its length via bpermute
                const uint64_t m_le = sel[r] & (lane_lt | ((uint64_t)1 << lane));
                const uint32_t sl = m_le ? 63 - clz64(m_le) : 0u;
                const uint32_t fs = (uint32_t)__shfl((int)flen[r], (int)sl);
                cov[r] = __ballot((m_le != 0 && lane < sl + fs) || lane < e0);
            };
            if constexpr (MODE == 1) {
#pragma unroll
                for (int r = 0; r < G; r++) do_match(r);
#pragma unroll
                for (int r = 0; r < G; r++) if (q[r] < t1) pb[q[r]] = len[r] | (off[r] << 6);
                if (tid < TILE_G / 16) {
                    const uint32_t wo = (loaded_end + tid * 16) & (WIN_BYTES - 1);
                    *(uint4 *)(lds + L_WIN + wo) = pf;
                    if (wo < WIN_MIRROR) *(uint4 *)(lds + L_WIN + WIN_BYTES + wo) = pf;
                }
                loaded_end += TILE_G;
                __syncthreads();                                                    // every wave has looked up
#pragma unroll
                for (int r = 0; r < G; r++) if (hv[r] && LZ_INS_COND) atomicMax(&table[hsh[r]], ((q[r] + 1) << TAG_BITS) | tag[r]);
                __syncthreads();                                                    // inserts + window chunk in place
                continue;
            }
            // near candidates of all groups first (LDS), then the far ones + adoption: the far bytes had that long to arrive
            // half of the waves of a SIMD run all matches, then all parses, the other half match / parse group by group:
            // vector-heavy and scalar-heavy stretches of different waves then overlap at the issue port (-3 %)
            if ((wave >> 2) & 1) {
#pragma unroll
                for (int r = 0; r < G; r++) do_match(r);
#pragma unroll
                for (int r = 0; r < G; r++) do_parse(r);
            } else {
#pragma unroll
                for (int r = 0; r < G; r++) { do_match(r); do_parse(r); }
            }
            LZ_STAMP(7);
            if (MODE != 2 && tid < TILE_G / 16) {
                const uint32_t wo = (loaded_end + tid * 16) & (WIN_BYTES - 1);
                *(uint4 *)(lds + L_WIN + wo) = pf;
                if (wo < WIN_MIRROR) *(uint4 *)(lds + L_WIN + WIN_BYTES + wo) = pf;
            }
            loaded_end += TILE_G;
            if (lane == 0) wend[wave] = el ? wbase + el : 0u;
            __syncthreads();                                                        // B3
            // every wave has finished its lookups: the tile's inserts go here (all of them land before B4, i.e. before the next lookups)
            if constexpr (MODE != 2) {
#pragma unroll
            for (int r = 0; r < G; r++) if (hv[r] && LZ_INS_COND) atomicMax(&table[hsh[r]], ((q[r] + 1) << TAG_BITS) | tag[r]);   // even positions only
            }
            LZ_STAMP(3);

            // ---- merge.  E = running end of the matches of the earlier waves (and the carry): a wave's last match moves E
            // to its end iff E lies inside the wave's region and that end is at least 3 bytes behind E (a wave entirely below
            // E contributes nothing, a shorter remainder is dropped by its owner).  Without such a case this is a prefix maximum.
            uint32_t E, tile_exit;
            {
                const bool lv = lane < LZ_WAVES;
                const uint32_t el_l = lv ? wend[lane & (LZ_WAVES - 1)] : 0u;
                const uint32_t incl = row_scan_max(el_l);
                const uint32_t before = DPP_ROW_SHR(incl, 1);                       // maximum over the earlier waves (0 for wave 0)
                const uint32_t prev = before > c_in ? before : c_in;
                const bool near = lv && el_l > prev && (el_l < prev + 3 || prev >= lane * RW + RW);
                if (__ballot(near) == 0 && !force_serial) {
                    const uint32_t m = wave ? rdlane(incl, wave - 1) : 0u, ma = rdlane(incl, LZ_WAVES - 1);
                    E = m > c_in ? m : c_in;
                    tile_exit = ma > c_in ? ma : c_in;
                } else {
                    uint32_t x = c_in; E = c_in;
                    for (uint32_t j = 0; j < LZ_WAVES; j++) {
                        if (j == wave) E = x;
                        const uint32_t ej = rdlane(el_l, j);
                        if (x < j * RW + RW && ej >= x + 3) x = ej;
                    }
                    tile_exit = x;
                }
            }
            // ---- masks of the final selection
            uint64_t fsel[G], litm[G];
            uint32_t nselp[G + 1], nlitp[G + 1];                                    // counts of the groups before group r
            {
                const uint32_t Ew = E > wbase ? (E - wbase < RW ? E - wbase : RW) : 0u;         // wave-relative, 0..RW
                const uint32_t in0 = t1 > t0 + wbase ? t1 - (t0 + wbase) : 0u;       // in-range positions of the wave
                uint64_t K[G], cv[G];                                               // positions below E; coverage
#pragma unroll
                for (int r = 0; r < G; r++) {
                    const uint32_t e = Ew > 64u * r ? Ew - 64u * r : 0u;
                    K[r] = mlow(e < 64 ? e : 64u); fsel[r] = sel[r] & ~K[r]; cv[r] = cov[r];
                }
                if (Ew > 0 && Ew < RW) {
                    const uint32_t grp = Ew >> 6, b = Ew & 63;
                    uint64_t cg = cov[0], sg = sel[0];
#pragma unroll
                    for (int r = 1; r < G; r++) if (grp == (uint32_t)r) { cg = cov[r]; sg = sel[r]; }
                    if (((cg >> b) & 1) && !((sg >> b) & 1)) {
                        // position E lies inside a match that starts below it (the nearest selected start): cut that match from the front
                        uint64_t below = sg & mlow(b);
                        uint32_t g2 = grp;
#pragma unroll
                        for (int r = G - 2; r >= 0; r--) if (!below && (uint32_t)r < grp && sel[r]) { below = sel[r]; g2 = (uint32_t)r; }
                        const uint32_t s2 = 63 - clz64(below);
                        uint32_t l2 = rdlane(flen[0], s2), o2 = rdlane(off[0], s2);
#pragma unroll
                        for (int r = 1; r < G; r++) if (g2 == (uint32_t)r) { l2 = rdlane(flen[r], s2); o2 = rdlane(off[r], s2); }
                        const uint32_t end2 = 64 * g2 + s2 + l2;                    // wave-relative end of the straddling match
                        const uint32_t rmn = end2 - Ew;
                        if (rmn >= 3) {
#pragma unroll
                            for (int r = 0; r < G; r++) if (grp == (uint32_t)r) { fsel[r] |= (uint64_t)1 << b; if (lane == b) { flen[r] = rmn; off[r] = o2; } }
                        } else {
                            // its last 1-2 bytes stay literals
#pragma unroll
                            for (int r = 0; r < G; r++) {
                                const uint32_t a0 = Ew > 64u * r ? (Ew - 64u * r < 64 ? Ew - 64u * r : 64u) : 0u;
                                const uint32_t z0 = end2 > 64u * r ? (end2 - 64u * r < 64 ? end2 - 64u * r : 64u) : 0u;
                                cv[r] &= ~(mlow(z0) & ~mlow(a0));
                            }
                        }
                    }
                }
                nselp[0] = 0; nlitp[0] = 0;
#pragma unroll
                for (int r = 0; r < G; r++) {
                    const uint32_t ir = in0 > 64u * r ? in0 - 64u * r : 0u;
                    litm[r] = mlow(ir < 64 ? ir : 64u) & ~(cv[r] | K[r]);
                    nselp[r + 1] = nselp[r] + (uint32_t)__popcll(fsel[r]);
                    nlitp[r + 1] = nlitp[r] + (uint32_t)__popcll(litm[r]);
                }
                // local literal index of the wave's last match (+1), 0 when it has none; the same for its first match
                // (needed by the chunk table only)
                uint32_t gl = 0, gf = 0;
#pragma unroll
                for (int r = G - 1; r >= 0; r--) if (!gl && fsel[r]) { const uint32_t sp = 63 - clz64(fsel[r]); gl = 1 + nlitp[r] + (uint32_t)__popcll(litm[r] & mlow(sp)); }
                if (CT) {
#pragma unroll
                    for (int r = 0; r < G; r++) if (!gf && fsel[r]) { const uint32_t sp = ctz64(fsel[r]); gf = 1 + nlitp[r] + (uint32_t)__popcll(litm[r] & mlow(sp)); }
                }
                if (lane == 0) { WPub p; p.cnt = nselp[G] | (nlitp[G] << 16); p.gl = gl | (gf << 16); wpub[wave] = p; }
            }
            LZ_STAMP(5);
            __syncthreads();                                                        // B4
            LZ_STAMP(4);
            // ---- 16-lane DPP scans over the waves' records
            uint32_t seq_base, lit_base, glast1_before;
            {
                const bool lv = lane < LZ_WAVES;
                const WPub pl = wpub[lane & (LZ_WAVES - 1)];
                const uint32_t incl = row_scan_add(lv ? pl.cnt : 0u), excl = incl - (lv ? pl.cnt : 0u);
                const uint32_t gabs = (lv && pl.gl) ? lit_run + (excl >> 16) + (pl.gl & 0xFFFF) : 0u;   // 1 + literal index of wave j's last match
                const uint32_t gmax = row_scan_max(gabs);
                const uint32_t ex_w = rdlane(excl, wave), tot = rdlane(incl, LZ_WAVES - 1);
                seq_base = seq_run + (ex_w & 0xFFFF); lit_base = lit_run + (ex_w >> 16);
                const uint32_t gb = wave ? rdlane(gmax, wave - 1) : 0u;
                glast1_before = gb > g_last1 ? gb : g_last1;
                const uint32_t ga = rdlane(gmax, LZ_WAVES - 1);
                g_last1 = ga > g_last1 ? ga : g_last1;
                if (CT) {
                    // chunk table (deflate): one entry per 2 KiB of positions (a tile holds CH = G / 2 of them, each the share of
                    // 16 / CH consecutive waves): state of the block's sequence / literal streams at the chunk's start and the
                    // literal index of the chunk's first match -- k_dblock packs a block chunk by chunk
                    constexpr uint32_t CH = TILE_G / TILE, WPC = LZ_WAVES / CH;
                    const uint64_t hm = __ballot(lv && pl.gl);
#pragma unroll
                    for (uint32_t h = 0; h < CH; h++) {
                        const uint32_t ex_h = rdlane(excl, h * WPC);
                        const uint64_t hm_h = hm & (mlow(WPC) << (h * WPC));
                        uint32_t g_first = lit_run + (tot >> 16);
                        if (hm_h) { const uint32_t j0 = ctz64(hm_h); g_first = lit_run + (rdlane(excl, j0) >> 16) + (rdlane(pl.gl, j0) >> 16) - 1; }
                        if (tid == 0) ctab[(size_t)gblk * (BLK_SIZE / TILE) + (t0 - blk_start) / TILE + h] = make_uint4(seq_run + (ex_h & 0xFFFF), lit_run + (ex_h >> 16), g_first, 0u);
                    }
                }
                seq_run += tot & 0xFFFF; lit_run += tot >> 16;
                next_free = t0 + tile_exit;
            }
            // ---- emission: all indices come from popcounts of the masks (no cross-lane data movement)
            {
                // literals between the wave's last match before group r and the start of group r (NONE: no match before it in this wave)
                uint32_t since[G];
                {
                    uint32_t acc = NONE;
#pragma unroll
                    for (int r = 0; r < G; r++) {
                        since[r] = acc;
                        if (fsel[r]) { const uint32_t sp = 63 - clz64(fsel[r]); acc = (uint32_t)__popcll(litm[r] & ~mlow(sp + 1)); }
                        else if (acc != NONE) acc += nlitp[r + 1] - nlitp[r];
                    }
                }
#pragma unroll
                for (int r = 0; r < G; r++) {
                    const uint32_t lq = (uint32_t)__popcll(litm[r] & lane_lt);       // literals of the group before this lane
                    const uint32_t lb = nlitp[r] + lq;                               // ... of the wave
                    if ((fsel[r] >> lane) & 1) {
                        const uint64_t pm = fsel[r] & lane_lt;
                        uint32_t ll;
                        if (pm) { const uint32_t sp = 63 - clz64(pm); ll = (uint32_t)__popcll(litm[r] & lane_lt & ~mlow(sp + 1)); }
                        else if (since[r] != NONE) ll = since[r] + lq;
                        else ll = lit_base + lb - (glast1_before - 1);
                        const uint32_t idx = seq_base + nselp[r] + (uint32_t)__popcll(pm);
                        if (idx < SEQ_CAP) bseq[idx] = seq_pack(ll, flen[r], off[r]);
                    }
                    if ((litm[r] >> lane) & 1) { const uint32_t li = lit_base + lb; if (li < BLK_SIZE) blit[li] = (uint8_t)lo[r]; }
                }
            }
            LZ_STAMP(6);
        } // tiles
        if (MODE != 1 && tid == 0) { blk[gblk].nseq = seq_run; blk[gblk].nlit = lit_run; }
    } // blocks
    if (STAMP && lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_lz_stamps[k], st_acc[k]);
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_lzm -- the match half of the split form: look-up, match, backward adoption and the tile's inserts, as in k_lz<MODE 1>, but with
// FOUR CONSECUTIVE POSITIONS PER LANE (lane i of wave w: positions t0 + 256 w + 4 i + j, j = 0..3) instead of one position per lane and
// group.  Nothing in this half needs a ballot over a group's positions, and with consecutive positions
//   * the 36 + 4 bytes around a lane's positions are ten aligned dwords, loaded once; the 8 / 16 / 32 bytes at position j are
//     v_alignbyte with a constant (j = 0: the registers themselves) -- k_lz loads and aligns them per position;
//   * the right neighbours of the adoption rounds sit in the same lane, except across the lane border (DPP row_shl: a row of
//     16 lanes is a group of 64 positions, and the zero fill at the row's end is the rule "adoption stops at the group border");
//     the offset moves along with every adoption instead of one ds_bpermute at the end;
//   * the four words of a lane go out as one 16-byte store; with even positions only, the inserts are those of j = 0 and 2.
// Same table, window, tile order and barriers as k_lz, hence the same words (tests/test_gpu_parity.py: forms of the LZ stage).
#define DPP_ROW_SHL1(v) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x101, 0xF, 0xF, true))   // value of lane i + 1 inside the row of 16 (0 at its end)
template <bool DEFL, bool STRONG>
__global__ __launch_bounds__(LZ_THREADS)
void k_lzm(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, uint32_t flags, uint32_t max_off, uint32_t *__restrict__ pbuf, uint32_t blk0) {
    constexpr uint32_t RW = 256, TILE_G = RW * LZ_WAVES;
    constexpr bool FAR = !DEFL;                             // deflate offsets (<= 32 KiB) never leave the LDS window
    constexpr uint32_t NEAR = NEAR_OFF;
    static_assert(LZ_G_ZSTD == 4 && LZ_G_DEFLATE == 4 && WIN_MIRROR >= 40, "k_lzm: four positions per lane, 36 bytes read behind a lane's first position");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *win32 = (uint32_t *)(lds + L_WIN);
    uint32_t *table = (uint32_t *)(lds + L_TABLE);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = uni(tid >> 6);
    const SegDesc sd = segs[blockIdx.x];
    const uint8_t *seg = src + sd.src_off;
    const uint32_t seg_len = sd.len;
    uint32_t *pb = pbuf + (size_t)(sd.blk_base - blk0) * BLK_SIZE;
    const bool adopt = (flags & F_ADOPT) != 0, ins_all = !(flags & F_INS2);

    for (uint32_t i = tid; i < HASH_ENTRIES / 4; i += LZ_THREADS) ((uint4 *)table)[i] = make_uint4(0, 0, 0, 0);
    uint32_t loaded_end = TILE_G + LOOKAHEAD + 16;
    for (uint32_t i = tid * 16; i < loaded_end; i += LZ_THREADS * 16) {
        const uint4 v = load_chunk(seg, i, seg_len);
        *(uint4 *)(lds + L_WIN + i) = v;
        if (i < WIN_MIRROR) *(uint4 *)(lds + L_WIN + WIN_BYTES + i) = v;
    }
    __syncthreads();
    uint4 pf = make_uint4(0, 0, 0, 0);

    const uint32_t nblk = (seg_len + BLK_SIZE - 1) / BLK_SIZE;
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t blk_start = b * BLK_SIZE;
        const uint32_t blk_end = (seg_len - blk_start < BLK_SIZE) ? seg_len : blk_start + BLK_SIZE;
        for (uint32_t t0 = blk_start; t0 < blk_end; t0 += TILE_G) {
            const uint32_t t1 = (blk_end - t0 < TILE_G) ? blk_end : t0 + TILE_G;
            if (tid < TILE_G / 16) { pf = (loaded_end + tid * 16 < seg_len) ? load_chunk(seg, loaded_end + tid * 16, seg_len) : make_uint4(0, 0, 0, 0); }
            const bool tile_full = (t1 - t0 == TILE_G) && (t0 + TILE_G + 8 <= seg_len);
            const uint32_t q0 = t0 + wave * RW + 4 * lane;
            // ---- the bytes around the lane's positions: D[k] = bytes q0 + 4 k .. + 3, Dm = the 4 (8) before q0
            uint32_t D[9], Dm1, Dm2 = 0;
            {
                const uint32_t *pq = win32 + ((q0 & (WIN_BYTES - 1)) >> 2);            // pq[1..8] may lie in the mirror
#pragma unroll
                for (int k = 0; k < 9; k++) D[k] = pq[k];
                Dm1 = win32[((q0 - 4) & (WIN_BYTES - 1)) >> 2];
                if (STRONG) Dm2 = win32[((q0 - 8) & (WIN_BYTES - 1)) >> 2];
            }
#define QW(k, j) ((j) ? __builtin_amdgcn_alignbyte(D[(k) + 1], D[k], (j)) : D[k])          /* 4 bytes at position j, + 4 k */
            // ---- look-up
            uint32_t hsh[4], tag[4], ent[4];
            bool hv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t q = q0 + j;
                hv[j] = tile_full || ((q < t1) && (q + 8 <= seg_len));
                const uint32_t h32 = QW(0, j) * 0x9E3779B1u + (QW(1, j) & 0xFFFFu) * 0x85EBCA6Bu;
                hsh[j] = __umulhi(h32, HASH_ENTRIES);
                tag[j] = (h32 >> 6) & TAG_MASK;
                ent[j] = hv[j] ? table[hsh[j]] : 0u;
            }
            // ---- candidates (rules as in k_lz); far ones get their bytes requested from the segment now
            uint32_t off[4];
            U4u fa[4]; uint32_t fb[4], fc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                fa[j].x = fa[j].y = fa[j].z = fa[j].w = fb[j] = fc[j] = 0;
                const uint32_t c1 = ent[j] >> TAG_BITS, o = q0 + j + 1 - c1;
                off[j] = (c1 > 8 && (ent[j] & TAG_MASK) == tag[j] && o <= max_off) ? o : 0u;
                if (FAR && seg_len > NEAR && max_off > NEAR) {
                    const uint32_t fo = off[j] > NEAR ? c1 - 5 : 0u;
                    fa[j] = *(const U4u *)(seg + fo);
                    fb[j] = *(const u32u *)(seg + fo + 16);
                    if (STRONG) fc[j] = *(const u32u *)(seg + (off[j] > NEAR ? fo - 4 : 0u));
                }
            }
            // ---- match
            uint32_t K[4];
            const bool edge = blk_end - (t0 + wave * RW) < RW + CAP1;                   // (uniform) only the block's last waves can run into its end
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t l = 0, bk = 0;
                const uint32_t o = off[j], q = q0 + j;
                if (o != 0) {
                    const uint32_t c = q - o;
                    const bool isfar = FAR && o > NEAR;
                    const uint32_t shc = (c & 3) * 8;
                    uint32_t w0, w1, w2, w3, bc, bc2 = 0;
                    if (isfar) { bc = fa[j].x; w0 = fa[j].y; w1 = fa[j].z; w2 = fa[j].w; w3 = fb[j]; bc2 = fc[j]; }
                    else {
                        const uint32_t *pc = win32 + ((c & (WIN_BYTES - 1)) >> 2);
                        const uint32_t d0 = pc[0], d1 = pc[1], d2 = pc[2], d3 = pc[3], d4 = pc[4], dm = win32[((c - 4) & (WIN_BYTES - 1)) >> 2];
                        w0 = __builtin_amdgcn_alignbit(d1, d0, shc); w1 = __builtin_amdgcn_alignbit(d2, d1, shc);
                        w2 = __builtin_amdgcn_alignbit(d3, d2, shc); w3 = __builtin_amdgcn_alignbit(d4, d3, shc);
                        bc = __builtin_amdgcn_alignbit(d0, dm, shc);
                        if (STRONG) bc2 = __builtin_amdgcn_alignbit(dm, win32[((c - 8) & (WIN_BYTES - 1)) >> 2], shc);
                    }
                    const uint32_t x0 = QW(0, j) ^ w0, x1 = QW(1, j) ^ w1, x2 = QW(2, j) ^ w2, x3 = QW(3, j) ^ w3;
                    const uint64_t xa = (uint64_t)x0 | ((uint64_t)x1 << 32), xb = (uint64_t)x2 | ((uint64_t)x3 << 32);
                    l = xa ? ctz64(xa) >> 3 : (xb ? 8 + (ctz64(xb) >> 3) : 16);
                    if (l == 16) {
                        uint32_t v0, v1, v2, v3;
                        if (isfar) { const U4u t = *(const U4u *)(seg + c + 16); v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w; }
                        else {
                            const uint32_t *pc2 = win32 + (((c + 16) & (WIN_BYTES - 1)) >> 2);
                            const uint32_t f0 = pc2[0], f1 = pc2[1], f2 = pc2[2], f3 = pc2[3], f4 = pc2[4];
                            v0 = __builtin_amdgcn_alignbit(f1, f0, shc); v1 = __builtin_amdgcn_alignbit(f2, f1, shc);
                            v2 = __builtin_amdgcn_alignbit(f3, f2, shc); v3 = __builtin_amdgcn_alignbit(f4, f3, shc);
                        }
                        const uint32_t y0 = QW(4, j) ^ v0, y1 = QW(5, j) ^ v1, y2 = QW(6, j) ^ v2, y3 = QW(7, j) ^ v3;
                        const uint64_t ya = (uint64_t)y0 | ((uint64_t)y1 << 32), yb = (uint64_t)y2 | ((uint64_t)y3 << 32);
                        l = 16 + (ya ? ctz64(ya) >> 3 : (yb ? 8 + (ctz64(yb) >> 3) : 16));
                    }
                    if (edge) { const uint32_t lim = blk_end - q; l = l < lim ? l : lim; }
                    if (l < MIN_MATCH) l = 0;
                    const uint32_t bqj = j ? __builtin_amdgcn_alignbyte(D[0], Dm1, j) : Dm1;       // the 4 bytes before q (q - 1 in the top byte)
                    const uint32_t xk = bqj ^ bc;
                    bk = (uint32_t)__builtin_clz(xk | 0xFFu) >> 3;
                    if (STRONG && xk == 0) { const uint32_t bq2 = j ? __builtin_amdgcn_alignbyte(Dm1, Dm2, j) : Dm2; bk = 4 + ((uint32_t)__builtin_clz((bq2 ^ bc2) | 0xFFu) >> 3); }
                }
                K[j] = (l << 6) | (bk << 3);
                if (STRONG && !l) K[j] = 0;
            }
            // ---- backward adoption: K = len << 6 | back << 3 | positions moved; the offset goes along
            if (adopt) {
                {   // round 1: the right neighbour's match, one byte longer
                    const uint32_t Kn = DPP_ROW_SHL1(K[0]), on = DPP_ROW_SHL1(off[0]);
                    uint32_t K1[4] = {K[1], K[2], K[3], Kn}, o1[4] = {off[1], off[2], off[3], on};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t T = K1[j] + 57u;
                        const bool a = (K1[j] & 0x38u) != 0 && T > (K[j] | 63u);
                        K[j] = a ? T : K[j]; off[j] = a ? o1[j] : off[j];
                    }
                }
                {   // round 2: the match two positions to the right (after round 1), two bytes longer
                    const uint32_t Ka = DPP_ROW_SHL1(K[0]), Kb = DPP_ROW_SHL1(K[1]), oa = DPP_ROW_SHL1(off[0]), ob = DPP_ROW_SHL1(off[1]);
                    uint32_t K2[4] = {K[2], K[3], Ka, Kb}, o2[4] = {off[2], off[3], oa, ob};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t T = K2[j] + 114u;
                        const bool a = (K2[j] & 0x30u) != 0 && T > (K[j] | 63u);
                        K[j] = a ? T : K[j]; off[j] = a ? o2[j] : off[j];
                    }
                }
                if (STRONG) {   // round 3: four positions to the right = the same position of the next lane, four bytes longer
                    uint32_t K4[4], o4[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { K4[j] = DPP_ROW_SHL1(K[j]); o4[j] = DPP_ROW_SHL1(off[j]); }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t T = K4[j] + 228u;
                        const bool a = (K4[j] & 0x20u) != 0 && T > (K[j] | 63u);
                        K[j] = a ? T : K[j]; off[j] = a ? o4[j] : off[j];
                    }
                }
            }
            if (q0 < t1) *(uint4 *)(pb + q0) = make_uint4((K[0] >> 6) | (off[0] << 6), (K[1] >> 6) | (off[1] << 6), (K[2] >> 6) | (off[2] << 6), (K[3] >> 6) | (off[3] << 6));
#undef QW
            if (tid < TILE_G / 16) {
                const uint32_t wo = (loaded_end + tid * 16) & (WIN_BYTES - 1);
                *(uint4 *)(lds + L_WIN + wo) = pf;
                if (wo < WIN_MIRROR) *(uint4 *)(lds + L_WIN + WIN_BYTES + wo) = pf;
            }
            loaded_end += TILE_G;
            __syncthreads();                                                        // every wave has looked up
#pragma unroll
            for (int j = 0; j < 4; j++) if (hv[j] && (ins_all || !(j & 1))) atomicMax(&table[hsh[j]], ((q0 + j + 1) << TAG_BITS) | tag[j]);
            __syncthreads();                                                        // inserts + window chunk in place
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_lzp -- the parse half of the split form with one LANE per parse region.  What a whole wave does in k_lz with scalar loops on
// ballot masks (an SALU instruction takes an issue slot like a vector one), sixteen lanes do here with vector arithmetic for the
// sixteen regions of a tile at once.  One wave (= one workgroup: no barriers, 8.4 KiB of LDS) per segment; per tile of 4 096 positions:
//   1. the tile's words (k_lz<MODE 1>'s output), 4 consecutive positions per lane and 16-byte load -> their lengths, a byte each, to
//      LDS; then one lane per GROUP of 64 positions: start / cap masks of the group
//      from its 64 length bytes, four at a time inside a register (byte-wise compares by carry-free subtraction, the four
//      results gathered into a nibble by one multiplication)
//   2. lanes 0..15: greedy walk over the region's eight half-groups on 32-bit masks (length of a chosen start from LDS; a capped
//      match is extended by the whole wave, the lengths are kept in LDS), merge across the regions = across the lanes (serial
//      form of the scan, DPP row scans for the counts), selection / literal masks, one record per group to LDS
//   3. one lane per group again: the group's sequences (offsets from the words in memory, four requested at a time)
//   4. the literals, 4 consecutive positions per lane (their input bytes were requested from memory before step 3): the lane's literal bytes are packed by v_perm (selector from a 16-entry
//      table) and stored behind the literals of the positions before it.
// Same results as k_lz<MODE = 2> (and so as the fused kernel): tests/test_gpu_parity.py runs all three.
constexpr uint32_t LZP_THREADS = 64;
template <bool CT, bool STRONG>
__global__ __launch_bounds__(LZP_THREADS)
void k_lzp(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, uint64_t *__restrict__ seqs, uint8_t *__restrict__ lits,
           BlkInfo *__restrict__ blk, uint4 *__restrict__ ctab, uint32_t flags, uint32_t max_len, const uint32_t *__restrict__ pbuf, uint32_t blk0) {
    constexpr uint32_t RW = 256, TG = 4096;
    static_assert(LZ_G_ZSTD == 4 && LZ_G_DEFLATE == 4 && BLK_SIZE % TG == 0 && CAP1 == 32, "k_lzp: regions of 4 groups, tiles of 16 regions");
    __shared__ uint32_t l32[TG / 4];                        // the tile's match lengths, one byte per position
    __shared__ uint4 lmask[TG / 64];                        // per group: start mask, cap mask
    __shared__ uint32_t plut[16];                           // v_perm selectors that pack the bytes named by a nibble
    __shared__ uint4 rec[3][TG / 64];                       // per group: [0] literal mask, first literal index, first sequence index; [1] chosen starts, the capped ones among all chosen;
                                                            // [2] literal-run base of its first sequence, xlen base, cut position | length << 8, cut offset
    __shared__ uint16_t xlen[16 * 8];                       // lengths of a region's extended matches, in the order the walk met them (<= 256 / 32)
    const uint32_t lane = threadIdx.x, w = lane & 15;
    const uint8_t *len8 = (const uint8_t *)l32;
    const SegDesc sd = segs[blockIdx.x];
    const uint32_t seg_len = sd.len;
    const uint8_t *seg = src + sd.src_off;
    const uint32_t *pb = pbuf + (size_t)(sd.blk_base - blk0) * BLK_SIZE;
    const uint32_t lazy = flags & F_LAZY;
    const uint32_t wbase = w * RW;
    const uint32_t ntile = (seg_len + TG - 1) / TG;
    const bool lv = lane < 16;
    if (lv) {                                               // selector of nibble n: the bytes whose bits are set, lowest first
        uint32_t sel = 0, j = 0;
        for (uint32_t bit = 0; bit < 4; bit++) if ((lane >> bit) & 1) { sel |= bit << (8 * j); j++; }
        plut[lane] = sel;
    }
    // literals: bits of the group's literal mask below this lane's four positions (lane i of a region: group i / 16, bits 4 (i % 16) ..)
    const uint64_t lit_below = ((uint64_t)1 << (4 * (lane & 15))) - 1;
    uint32_t next_free = 0, seq_run = 0, lit_run = 0, g_last1 = 1;     // block-level parse state (uniform)

#ifdef LZP_PROF   // diagnostic build (scripts/lzp_stamps.py): s_memtime deltas per phase, summed over all waves
    unsigned long long pa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0 = __builtin_amdgcn_s_memtime();
#define LZP_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pa[k] += t_ - pt0; pt0 = t_; } while (0)
#else
#define LZP_STAMP(k) do { } while (0)
#endif
    for (uint32_t T = 0; T < ntile; T++) {
        const uint32_t t0 = T * TG, blk_start = t0 & ~(BLK_SIZE - 1);
        const uint32_t blk_end = (seg_len - blk_start < BLK_SIZE) ? seg_len : blk_start + BLK_SIZE;
        const uint32_t t1 = (blk_end - t0 < TG) ? blk_end : t0 + TG;
        const uint32_t npos = t1 - t0;
        const uint32_t ext_lim = (t1 + LOOKAHEAD < blk_end) ? t1 + LOOKAHEAD : blk_end;
        const uint32_t gblk = sd.blk_base + (t0 >> PNA_BLK_LOG);
        const uint32_t ng = (npos + 63) >> 6;                                           // groups with positions in them
        // ---- 1. lengths to LDS (positions behind the block's end count as "no match"; the words of a whole tile lie inside the segment's
        // share of pbuf, whole blocks, so the loads need no bounds of their own)
        {
            const uint4 *pt = (const uint4 *)(pb + t0);
            for (uint32_t wq = 0; wq < 4 && wq * 1024 < npos; wq++) {                   // four regions at a time: their loads go out together
                uint4 v[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) v[i] = pt[(wq * 4 + i) * 64 + lane];
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const uint32_t p = (wq * 4 + i) * RW + lane * 4;
                    uint32_t d = (v[i].x & 63u) | ((v[i].y & 63u) << 8) | ((v[i].z & 63u) << 16) | ((v[i].w & 63u) << 24);
                    if (npos < TG && p + 4 > npos) d = p >= npos ? 0u : d & (0xFFFFFFFFu >> (8 * (p + 4 - npos)));
                    l32[p >> 2] = d;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // masks of group `lane` from its 64 length bytes.  Per register of 4 lengths l (all < 64): x = l | 0x80 per byte; x - 6 keeps the top bit iff
        // l >= 6; x - l' (l' = the next position's length, 0 behind the group) keeps it iff l' <= l, i.e. the position does not defer; no byte
        // borrows from its neighbour.  The four top bits become a nibble by (y >> 7) * 0x00204081 >> 21.
        if (lane < ng) {
            uint32_t d[17];
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) { const uint4 t = ((const uint4 *)l32)[lane * 4 + i]; d[4 * i] = t.x; d[4 * i + 1] = t.y; d[4 * i + 2] = t.z; d[4 * i + 3] = t.w; }
            d[16] = 0;
            uint32_t em2[2] = {0, 0}, cm2[2] = {0, 0};
#pragma unroll
            for (uint32_t i = 0; i < 16; i++) {
                const uint32_t x = d[i] | 0x80808080u;
                uint32_t e = x - 0x06060606u;
                if (lazy) {
                    e &= x - __builtin_amdgcn_alignbyte(d[i + 1], d[i], 1);
                    if (STRONG) e &= x + 0x01010101u - __builtin_amdgcn_alignbyte(d[i + 1], d[i], 2);   // ... nor to the position after the next (longer by two or more)
                }
                const uint32_t en = ((((e >> 7) & 0x01010101u) * 0x00204081u) >> 21) & 15u;
                const uint32_t cn = ((((d[i] >> 5) & 0x01010101u) * 0x00204081u) >> 21) & 15u;
                em2[i >> 3] |= en << (4 * (i & 7)); cm2[i >> 3] |= cn << (4 * (i & 7));
            }
            lmask[lane] = make_uint4(em2[0], em2[1], cm2[0], cm2[1]);
        }
        __builtin_amdgcn_wave_barrier();
        LZP_STAMP(0);
        if (t0 == blk_start) { next_free = blk_start; seq_run = 0; lit_run = 0; g_last1 = 1; }
        // ---- 2. the region's greedy walk, from the tile's carry if that reaches into it; half-groups of 32 positions: one-register masks
        const uint32_t c_in = next_free > t0 ? next_free - t0 : 0u;
        uint64_t sel[4], cov[4], cm[4];
        uint32_t el = 0, nx = 0;
        uint32_t em_lo[4], em_hi[4], cm_lo[4], cm_hi[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint4 m = make_uint4(0, 0, 0, 0);
            if (lv && w * 4 + r < ng) m = lmask[w * 4 + r];
            em_lo[r] = m.x; em_hi[r] = m.y; cm_lo[r] = m.z; cm_hi[r] = m.w;
        }
        {
            uint32_t cur = c_in > wbase ? (c_in - wbase < RW ? c_in - wbase : RW) : 0u;
            uint32_t s32[8], c32[8];
#pragma unroll
            for (int hb = 0; hb < 8; hb++) {
                const uint32_t em32 = (hb & 1) ? em_hi[hb >> 1] : em_lo[hb >> 1], cm32 = (hb & 1) ? cm_hi[hb >> 1] : cm_lo[hb >> 1];
                const uint32_t e0 = cur > 32u * hb ? cur - 32u * hb : 0u;
                uint32_t rem = e0 < 32 ? em32 & (0xFFFFFFFFu << e0) : 0u;
                uint32_t e_last = e0, selr = 0, covr = e0 < 32 ? (1u << e0) - 1 : 0xFFFFFFFFu;
                while (__ballot(rem != 0)) {
                    const bool a = rem != 0;
                    const uint32_t s = a ? (uint32_t)__builtin_ctz(rem) : 0u;
                    const uint32_t ps = wbase + 32u * hb + s;                       // tile-relative
                    uint32_t L = a ? len8[ps] : 0u;
                    const bool cap = a && ((cm32 >> s) & 1);
                    uint64_t need = __ballot(cap);
                    if (need) {
                        const uint32_t qs = t0 + ps;
                        const uint32_t pwv = cap ? pb[qs] : 0u;                      // (its offset)
                        const uint32_t xl = ext_lim - qs < max_len ? ext_lim - qs : max_len;
                        while (need) {
                            const uint32_t k = ctz64(need); need &= need - 1;
                            const uint32_t qk = rdlane(qs, k), ok = rdlane(pwv >> 6, k);
                            const uint32_t Lk = lz_extend_mem(seg, seg_len, qk, qk - ok, rdlane(L, k), rdlane(xl, k), lane);
                            if (lane == k) L = Lk;
                        }
                        if (cap) { xlen[w * 8 + (nx & 7)] = (uint16_t)L; nx++; }
                    }
                    if (a) {
                        const uint32_t e = s + L;
                        const uint32_t me = e < 32 ? (1u << e) - 1 : 0xFFFFFFFFu;   // positions below the match's end
                        selr |= 1u << s; e_last = e;
                        covr |= me & ~((1u << s) - 1);
                        rem &= ~me;
                    }
                }
                s32[hb] = selr; c32[hb] = covr;
                if (selr) el = 32u * hb + e_last;
                cur = 32u * hb + (e_last > 32 ? e_last : 32u);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                sel[r] = (uint64_t)s32[2 * r] | ((uint64_t)s32[2 * r + 1] << 32); cov[r] = (uint64_t)c32[2 * r] | ((uint64_t)c32[2 * r + 1] << 32);
                cm[r] = (uint64_t)cm_lo[r] | ((uint64_t)cm_hi[r] << 32);
            }
        }
        LZP_STAMP(1);
        uint32_t pc[4];                                     // capped chosen starts in the groups before r = index base into xlen
        pc[0] = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) pc[r + 1] = pc[r] + (uint32_t)__popcll(sel[r] & cm[r]);

        // ---- merge across the lanes: the serial form of the scan (k_lz takes it only when an end falls 1-2 bytes behind E; it is the definition)
        uint32_t E = c_in, tile_exit;
        {
            const uint32_t wend = el ? wbase + el : 0u;
            uint32_t x = c_in;
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t ek = rdlane(wend, k);
                if (w == k) E = x;
                if (x < k * RW + RW && ek >= x + 3) x = ek;
            }
            tile_exit = x;
        }
        uint64_t fsel[4], litm[4];
        uint32_t nselp[5], nlitp[5];
        uint32_t cut_r = 4, cut_b = 0, cut_len = 0, cut_off = 0;   // the match cut from the front at E, if any
        {
            const uint32_t Ew = E > wbase ? (E - wbase < RW ? E - wbase : RW) : 0u;
            const uint32_t in0 = t1 > t0 + wbase ? t1 - (t0 + wbase) : 0u;
            uint64_t K[4], cv[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t e = Ew > 64u * r ? Ew - 64u * r : 0u;
                K[r] = mlow(e < 64 ? e : 64u); fsel[r] = sel[r] & ~K[r]; cv[r] = cov[r];
            }
            if (lv && Ew > 0 && Ew < RW) {
                const uint32_t grp = Ew >> 6, b = Ew & 63;
                uint64_t cg = cov[0], sg = sel[0];
#pragma unroll
                for (int r = 1; r < 4; r++) if (grp == (uint32_t)r) { cg = cov[r]; sg = sel[r]; }
                if (((cg >> b) & 1) && !((sg >> b) & 1)) {
                    uint64_t below = sg & mlow(b);
                    uint32_t g2 = grp;
#pragma unroll
                    for (int r = 2; r >= 0; r--) if (!below && (uint32_t)r < grp && sel[r]) { below = sel[r]; g2 = (uint32_t)r; }
                    const uint32_t s2 = 63 - clz64(below);
                    const uint32_t pw2 = pb[t0 + wbase + 64 * g2 + s2];
                    uint64_t sc2 = sel[0] & cm[0]; uint32_t pc2 = pc[0];
#pragma unroll
                    for (int r = 1; r < 4; r++) if (g2 == (uint32_t)r) { sc2 = sel[r] & cm[r]; pc2 = pc[r]; }
                    const uint32_t l2 = ((sc2 >> s2) & 1) ? xlen[w * 8 + ((pc2 + (uint32_t)__popcll(sc2 & mlow(s2))) & 7)] : (pw2 & 63u);
                    const uint32_t end2 = 64 * g2 + s2 + l2, rmn = end2 - Ew;
                    if (rmn >= 3) {
#pragma unroll
                        for (int r = 0; r < 4; r++) if (grp == (uint32_t)r) fsel[r] |= (uint64_t)1 << b;
                        cut_r = grp; cut_b = b; cut_len = rmn; cut_off = pw2 >> 6;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const uint32_t a0 = Ew > 64u * r ? (Ew - 64u * r < 64 ? Ew - 64u * r : 64u) : 0u;
                            const uint32_t z0 = end2 > 64u * r ? (end2 - 64u * r < 64 ? end2 - 64u * r : 64u) : 0u;
                            cv[r] &= ~(mlow(z0) & ~mlow(a0));
                        }
                    }
                }
            }
            nselp[0] = 0; nlitp[0] = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t ir = in0 > 64u * r ? in0 - 64u * r : 0u;
                litm[r] = lv ? mlow(ir < 64 ? ir : 64u) & ~(cv[r] | K[r]) : 0;
                if (!lv) fsel[r] = 0;
                nselp[r + 1] = nselp[r] + (uint32_t)__popcll(fsel[r]);
                nlitp[r + 1] = nlitp[r] + (uint32_t)__popcll(litm[r]);
            }
        }
        uint32_t gl = 0, gf = 0;                            // 1 + the region's literal index at its last / first match, 0 = it has none
#pragma unroll
        for (int r = 3; r >= 0; r--) if (!gl && fsel[r]) { const uint32_t sp = 63 - clz64(fsel[r]); gl = 1 + nlitp[r] + (uint32_t)__popcll(litm[r] & mlow(sp)); }
        if (CT) {
#pragma unroll
            for (int r = 0; r < 4; r++) if (!gf && fsel[r]) { const uint32_t sp = ctz64(fsel[r]); gf = 1 + nlitp[r] + (uint32_t)__popcll(litm[r] & mlow(sp)); }
        }
        {
            const uint32_t cnt = nselp[4] | (nlitp[4] << 16);
            const uint32_t incl = row_scan_add(cnt), excl = incl - cnt;
            const uint32_t gabs = gl ? lit_run + (excl >> 16) + gl : 0u;
            const uint32_t gmax = row_scan_max(gabs);
            const uint32_t tot = rdlane(incl, 15);
            const uint32_t seq_base = seq_run + (excl & 0xFFFF), lit_base = lit_run + (excl >> 16);
            const uint32_t gb = DPP_ROW_SHR(gmax, 1);
            const uint32_t glast1_before = gb > g_last1 ? gb : g_last1;
            const uint32_t ga = rdlane(gmax, 15);
            if (CT) {
                constexpr uint32_t CH = TG / TILE, WPC = 16 / CH;
                const uint32_t hrow = (uint32_t)__ballot(gl != 0) & 0xFFFFu;
#pragma unroll
                for (uint32_t h = 0; h < CH; h++) {
                    const uint32_t ex_h = rdlane(excl, h * WPC);
                    const uint32_t hm_h = hrow & (((1u << WPC) - 1) << (h * WPC));
                    uint32_t g_first = lit_run + (tot >> 16);
                    if (hm_h) { const uint32_t j0 = (uint32_t)__builtin_ctz(hm_h); g_first = lit_run + (rdlane(excl, j0) >> 16) + rdlane(gf, j0) - 1; }
                    if (lane == 0) ctab[(size_t)gblk * (BLK_SIZE / TILE) + (t0 - blk_start) / TILE + h] = make_uint4(seq_run + (ex_h & 0xFFFF), lit_run + (ex_h >> 16), g_first, 0u);
                }
            }
            g_last1 = ga > g_last1 ? ga : g_last1;
            seq_run += tot & 0xFFFF; lit_run += tot >> 16;
            next_free = t0 + rdlane(tile_exit, 0);
            // one record per group for the lanes that write its sequences and literals
            if (lv) {
                uint32_t prevl = 0; bool any = false;       // literal index (region-local) at the region's latest chosen start so far
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t llbase = any ? nlitp[r] - prevl : lit_base + nlitp[r] - (glast1_before - 1);
                    if (fsel[r]) { const uint32_t sp = 63 - clz64(fsel[r]); prevl = nlitp[r] + (uint32_t)__popcll(litm[r] & mlow(sp)); any = true; }
                    const uint64_t sc = sel[r] & cm[r];
                    const uint32_t cutw = cut_r == (uint32_t)r ? cut_b | (cut_len << 8) : 64u;
                    rec[0][w * 4 + r] = make_uint4((uint32_t)litm[r], (uint32_t)(litm[r] >> 32), lit_base + nlitp[r], seq_base + nselp[r]);
                    rec[1][w * 4 + r] = make_uint4((uint32_t)fsel[r], (uint32_t)(fsel[r] >> 32), (uint32_t)sc, (uint32_t)(sc >> 32));
                    rec[2][w * 4 + r] = make_uint4(llbase, w * 8 + pc[r], cutw, cut_off);
                }
            }
            if (t1 == blk_end && lane == 0) { blk[gblk].nseq = seq_run; blk[gblk].nlit = lit_run; }
        }
        __builtin_amdgcn_wave_barrier();
        LZP_STAMP(2);
        // the input bytes of the lane's four positions in every region, for the literals: requested now, used behind the sequences
        uint32_t lw[16];
#pragma unroll
        for (uint32_t wr = 0; wr < 16; wr++) {
            const uint32_t q = t0 + wr * RW + 4 * lane;
            lw[wr] = 0;
            if (q + 4 <= seg_len) lw[wr] = *(const uint32_t *)(seg + q);                 // (segments start at multiples of 16)
            else { for (uint32_t i = 0; i < 3; i++) if (q + i < seg_len) lw[wr] |= (uint32_t)seg[q + i] << (8 * i); }
        }
        // ---- 3. the sequences, one lane per group
        {
            uint64_t *bseq = seqs + (size_t)gblk * SEQ_CAP;
            const uint4 ra = rec[0][lane], rb = rec[1][lane], rc = rec[2][lane];
            const uint64_t lm = (uint64_t)ra.x | ((uint64_t)ra.y << 32), sc = (uint64_t)rb.z | ((uint64_t)rb.w << 32);
            uint64_t rem = lane < ng ? (uint64_t)rb.x | ((uint64_t)rb.y << 32) : 0;
            uint32_t idx = ra.w, prev = 0;
            const uint32_t cut_b2 = rc.z & 0xFFu, cut_l2 = rc.z >> 8;
            bool first = true;
            const uint32_t *pg = pb + t0 + lane * 64;
            while (rem) {
                // four starts at a time: their words (the offsets) are requested together
                uint32_t sq[4], pw[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    sq[u] = rem ? ctz64(rem) : 64u;
                    pw[u] = rem ? pg[sq[u]] : 0u;
                    rem &= rem - 1;                                                 // (0 stays 0)
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t s = sq[u];
                    if (s < 64) {
                        uint32_t ml = pw[u] & 63u, of = pw[u] >> 6;
                        if ((sc >> s) & 1) ml = xlen[(rc.y + (uint32_t)__popcll(sc & mlow(s))) & 127];
                        if (cut_b2 == s) { ml = cut_l2; of = rc.w; }
                        const uint32_t lq = (uint32_t)__popcll(lm & mlow(s));
                        const uint32_t ll = first ? rc.x + lq : lq - prev;
                        if (idx < SEQ_CAP) bseq[idx] = seq_pack(ll, ml, of);
                        idx++; prev = lq; first = false;
                    }
                }
            }
        }
        LZP_STAMP(3);
        // ---- 4. literals, region by region, 4 consecutive positions per lane
        {
            uint8_t *blit = lits + (size_t)gblk * BLK_SIZE;
            const uint32_t sh = 4 * (lane & 15);
#pragma unroll
            for (uint32_t wr = 0; wr < 16; wr++) {
                if (wr * RW >= npos) continue;                                      // (uniform)
                const uint4 m = rec[0][wr * 4 + (lane >> 4)];
                const uint32_t wd = lw[wr];
                const uint64_t lm = (uint64_t)m.x | ((uint64_t)m.y << 32);
                const uint32_t nib = (uint32_t)(lm >> sh) & 15u;
                const uint32_t pk = __builtin_amdgcn_perm(wd, wd, plut[nib]), cnt = (uint32_t)__popc(nib);
                uint8_t *o = blit + m.z + (uint32_t)__popcll(lm & lit_below);
                if (cnt > 0) o[0] = (uint8_t)pk;
                if (cnt > 1) o[1] = (uint8_t)(pk >> 8);
                if (cnt > 2) o[2] = (uint8_t)(pk >> 16);
                if (cnt > 3) o[3] = (uint8_t)(pk >> 24);
            }
        }
        __builtin_amdgcn_wave_barrier();
        LZP_STAMP(4);
    }
#ifdef LZP_PROF
    if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_lz_stamps[k], pa[k]);
#endif
}

template <int G, bool CT, bool STRONG>
static void launch_lz_g(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
                        uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, hipEvent_t ev_match) {
    static const hipError_t attr_set = [] {                    // once per process, thread-safe (contexts may be created on several threads)
        (void)hipFuncSetAttribute((const void *)k_lz<false, G, CT, STRONG, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_TOTAL);
        (void)hipFuncSetAttribute((const void *)k_lz<false, G, CT, STRONG, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_TOTAL);
        (void)hipFuncSetAttribute((const void *)k_lzm<CT, STRONG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_TOTAL);
        return hipFuncSetAttribute((const void *)k_lz<true, G, CT, STRONG, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_TOTAL);
    }();
    (void)attr_set;
    if (pbuf) {
        if (flags & FLAG_SPLIT_WAVEPARSE) {
            hipLaunchKernelGGL((k_lz<false, G, CT, STRONG, 1>), dim3(nseg), dim3(LZ_THREADS), L_TOTAL, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len, pbuf, blk0);
            if (ev_match) (void)hipEventRecord(ev_match, st);
            hipLaunchKernelGGL((k_lz<false, G, CT, STRONG, 2>), dim3(nseg), dim3(LZ_THREADS), 4 * LZ_WAVES + 8 * LZ_WAVES, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len, pbuf, blk0);
            return;
        }
        hipLaunchKernelGGL((k_lzm<CT, STRONG>), dim3(nseg), dim3(LZ_THREADS), L_TOTAL, st, src, segs, flags, max_off, pbuf, blk0);
        if (ev_match) (void)hipEventRecord(ev_match, st);
        hipLaunchKernelGGL((k_lzp<CT, STRONG>), dim3(nseg), dim3(LZP_THREADS), 0, st, src, segs, seqs, lits, blk, ctab, flags, max_len, pbuf, blk0);
    }
    else if (flags & FLAG_STAMP) hipLaunchKernelGGL((k_lz<true, G, CT, STRONG, 0>), dim3(nseg), dim3(LZ_THREADS), L_TOTAL, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len, pbuf, blk0);
    else hipLaunchKernelGGL((k_lz<false, G, CT, STRONG, 0>), dim3(nseg), dim3(LZ_THREADS), L_TOTAL, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len, pbuf, blk0);
}
// zstd launches (no chunk table) run LZ_G_ZSTD positions per lane and tile, deflate launches LZ_G_DEFLATE (k_dblock walks the 2 KiB chunks of the table)
// pbuf != nullptr: the split form (two kernels; pbuf holds one word per position of the launch's blocks, blk0 = the first of them;
// ev_match, if given, is recorded between the two)
void launch_lz(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
               uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, hipEvent_t ev_match) {
    const bool strong = (flags & F_STRONG) && (flags & F_ADOPT);
    if (ctab) { if (strong) launch_lz_g<LZ_G_DEFLATE, true, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match);
                else launch_lz_g<LZ_G_DEFLATE, true, false>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match); }
    else { if (strong) launch_lz_g<LZ_G_ZSTD, false, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match);
           else launch_lz_g<LZ_G_ZSTD, false, false>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match); }
}

// diagnostic: read and clear the phase stamps (cycles summed over workgroups)
void lz_read_stamps(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lz_stamps), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz_stamps), z, sizeof(z));
}

} // namespace pna
