// lz_common.h -- what the kernels of the LZ stage share: LDS layout of the match finder (window + hash table), flags, cross-lane helpers,
// the wave-cooperative match extension.  k_lz.hip: the one-kernel form and its two halves as kernels; k_lz_split.hip: k_lzm + k_lzp, the
// default split form.
#pragma once
#include <hip/hip_runtime.h>
#include "pna_dev.h"

#ifndef PNA_EXP
#define PNA_EXP 0          /* timing experiments (scripts/exp_variants.sh builds variants of the library with -DPNA_EXP=bits) */
#endif
namespace pna {

constexpr uint32_t TAG_BITS = 11, TAG_MASK = (1u << TAG_BITS) - 1;

// LDS layout (byte offsets into the dynamic shared array)
constexpr uint32_t L_WIN    = 0;
constexpr uint32_t WIN_MIRROR = 48;                        // the window's first 48 bytes again behind its end: unaligned reads never wrap (k_lz reads 16 past a position, k_lzm 36 past a lane's first)
// Two geometries share the CU's 160 KiB (WLOG = log2 of the window): 64 KiB window + 24 512-slot table (deflate, whose 32 KiB look-back must lie in
// the window; the zstd sets without far candidates, whose look-back IS the window), and 32 KiB window + 32 704-slot table (the zstd sets with far
// candidates: a third more slots are worth +1.6 % of ratio on text, and what the window no longer holds -- candidates more than NEAR bytes back -- is
// verified against the segment in HBM / L2 like everything beyond the window before)
// TAB3 (round 4): the PACKED table of the zstd default / high sets.  What the table remembers sets the ratio on text (LAB_LOG.md 3.3: 32 704 slots 2.759,
// 49 152: 2.83, 65 536: 2.86), and LDS holds no more 32-bit entries; an entry needs 19 bits of (even) position and tolerates a 2-bit tag -- a false
// candidate only costs a compare that fails --, so THREE entries of 21 bits share a 64-bit LDS word: 49 062 slots next to the 32 KiB window, 55 206 next to the
// 16 KiB one.  slot -> word = mulhi(hash, words), field = ((hash & 0xFFFF) * 3) >> 16.  A tile's inserts are ONE ds_max_u64 each of "the word as my look-up
// saw it, my field replaced": every contender's value exceeds the old word (positions only grow), among a tile's contenders for one word the highest
// (field, position) wins and the others are LOST (4 - 8 % of the inserts, -0.1 % of ratio in the model -- oracle/zstd_model.c does exactly this), no CAS loop.
template <uint32_t WLOG, bool TAB3 = false> struct LzGeo {
    static_assert(WLOG >= 14 && WLOG <= 16, "window of 16, 32 or 64 KiB");
    static_assert(!TAB3 || WLOG < 16, "the packed table belongs to the zstd sets with far candidates");
    static constexpr uint32_t WIN     = 1u << WLOG;
    static constexpr uint32_t L_TABLE = L_WIN + WIN + WIN_MIRROR;
    static constexpr uint32_t WORDS3  = (160u * 1024 - 12 * LZ_WAVES - L_TABLE) / 8 & ~1u;           // 64-bit words of the packed table: 16 354 (32 KiB window), 18 402 (16 KiB)
    static constexpr uint32_t ENTRIES = TAB3 ? 3 * WORDS3 : (WLOG == 16 ? HASH_ENTRIES : ((160u * 1024 - 12 * LZ_WAVES - L_TABLE) / 4 & ~63u));   // 32 704 (32 KiB window), 36 800 (16 KiB); packed: 49 062, 55 206
    static constexpr uint32_t TABLE_BYTES = TAB3 ? 8u * WORDS3 : 4u * ENTRIES;
    static constexpr uint32_t L_WEND  = L_TABLE + TABLE_BYTES;    // 16 x u32: tile-relative end of each wave's last match (0 = none)
    static constexpr uint32_t L_WPUB  = L_WEND + 4 * LZ_WAVES;    // 16 x 8 B
    static constexpr uint32_t L_TOTAL = L_WPUB + 8 * LZ_WAVES;
    static constexpr uint32_t NEAR    = WLOG == 16 ? NEAR_OFF : WIN - 2 * 1024 * LZ_G_ZSTD - LOOKAHEAD - 16 - 240;   // candidates at most this far back are verified in the window (32 KiB: 23 296; 16 KiB: 6 912)
    static constexpr uint32_t NEARM   = NEAR + 1024 * LZ_G_ZSTD + (LOOKAHEAD + 16 - 64);   // the split form's match kernel (k_lzm) keeps a tile more of look-back in its window and runs only 64 bytes ahead: what it verifies in LDS (32 KiB: 28 368)
    static_assert(L_TOTAL <= 160 * 1024 && TABLE_BYTES % 16 == 0 && L_TABLE % 16 == 0, "k_lz's LDS: window + table + records within one CU's 160 KiB");
};
// the packed table's arithmetic: (word, bit position of the field) of a hash; the 21-bit entry of an even position; the word with a field replaced
constexpr uint32_t T3_MASK = 0x1FFFFFu;
template <uint32_t NW> __device__ __forceinline__ void t3_slot(uint32_t h32, uint32_t &widx, uint32_t &sh) { widx = __umulhi(h32, NW); sh = (((h32 & 0xFFFFu) * 3u) >> 16) * 21u; }
__device__ __forceinline__ uint32_t t3_tag(uint32_t h32) { return (h32 >> 16) & 3u; }
// field at bit 0, 21 or 42 of the word WITHOUT a 64-bit shift (a quarter-rate instruction): v_alignbit_b32 takes the shift modulo 32 -- 0, 21, 10 -- over
// {hi, lo} for the first two fields and over {hi, hi} for the third
__device__ __forceinline__ uint32_t t3_field(uint64_t w, uint32_t sh) {
    if (PNA_EXP & 64) { const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32); return __builtin_amdgcn_alignbit(hi, sh == 42u ? hi : lo, sh) & T3_MASK; }
    return (uint32_t)(w >> sh) & T3_MASK;
}
__device__ __forceinline__ uint32_t t3_entry(uint32_t q_even, uint32_t tag2) { return (q_even << 1) | tag2; }     // (q / 2) << 2 | tag
__device__ __forceinline__ uint32_t t3_pos(uint32_t fld) { return (fld >> 1) & ~1u; }                             // the position an entry names
__device__ __forceinline__ uint64_t t3_put(uint64_t w, uint32_t sh, uint32_t fld, uint32_t v) { return w ^ ((uint64_t)(fld ^ v) << sh); }
__device__ __forceinline__ void t3_max(uint64_t *p, uint64_t v) { (void)__hip_atomic_fetch_max((unsigned long long *)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct WPub  { uint32_t cnt; uint32_t gl; };   // cnt = nsel | nlit << 16; gl = (local literal index of the LAST match + 1) | (same for the FIRST match) << 16, 0 = no match
static_assert(sizeof(WPub) == 8, "LDS record size");

constexpr uint32_t FLAG_SPLIT_WAVEPARSE = 0x1000u;   // split form: the parse half as k_lz<MODE = 2> (a wave per region) instead of k_lzp (testing)
constexpr uint32_t FLAG_STAMP = 0x100u, FLAG_FORCE_SERIAL = 0x200u;   // 0x200: always take the serial form of the end scan (testing)
static_assert(CAP1 >= 16 && CAP1 % 16 == 0 && CAP1 <= 32 && BACK_CAP == 3 && MIN_MATCH > 3, "the match step compares 16 bytes at a time, the next 16 only where all before matched");
static_assert(GROUPS_PER_WAVE == 2 && TILE == 2048, "TILE / GROUPS_PER_WAVE describe the G = 2 (deflate) form; k_lz itself is generic in G");

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t ctz64(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
__device__ __forceinline__ uint32_t clz64(uint64_t v) { return (uint32_t)__builtin_clzll(v); }
// Index of the first nonzero byte among the 16 of four XOR words (16: none), branch-free: v_ffbl_b32 yields 0xFFFFFFFF for a zero word, the saturating
// adds keep it there, two min3 / min pick the lowest bit index.  (Written as `xa ? ctz64(xa) >> 3 : (xb ? ... : 16)` the compiler built two nested
// exec-masked branches per compare: ~22 vector + 10 scalar instructions instead of 11.)
__device__ __forceinline__ uint32_t ffbl_hw(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t first_diff16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    const uint32_t f0 = ffbl_hw(x0), f1 = __builtin_elementwise_add_sat(ffbl_hw(x1), 32u), f2 = __builtin_elementwise_add_sat(ffbl_hw(x2), 64u),
                   f3 = __builtin_elementwise_add_sat(ffbl_hw(x3), 96u);
    if (PNA_EXP & 32) { uint32_t m = f0 < f1 ? f0 : f1; const uint32_t n = f2 < f3 ? f2 : f3; m = m < n ? m : n; m >>= 3; return m < 16u ? m : 16u; }
    uint32_t n, m;                                                                  // (two v_min3_u32 -- the cap of 16 bytes rides in the first; left to the compiler: three mins)
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(n) : "v"(f2), "v"(f3), "s"(128u));
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(f0), "v"(f1), "v"(n));
    return m >> 3;
}
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

// A window word by its LDS BYTE ADDRESS.  The match kernels (k_lz, k_lzm, k_lzms) have no static __shared__ variables, so their dynamic LDS starts at address 0 and the window (L_WIN = 0)
// with it (checked once per workgroup: lds_base_is_zero); through the `lds` symbol every address computation ends in an add of the symbol's (zero) address that the
// compiler cannot fold -- three v_add_u32 v, 0, v per match step.
typedef const __attribute__((address_space(3))) uint32_t lds_cu32;
__device__ __forceinline__ lds_cu32 *lds_word(uint32_t byte_addr) { return (lds_cu32 *)(uintptr_t)byte_addr; }
__device__ __forceinline__ bool lds_base_is_zero(const uint8_t *dyn) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)dyn == 0u; }
// 8 / 4 bytes at an arbitrary segment position from the circular window
template <uint32_t WB>
__device__ __forceinline__ void fetch8(const uint32_t *win32, uint32_t pos, uint32_t &lo, uint32_t &hi) {
    const uint32_t *p = win32 + ((pos & (WB - 1)) >> 2);                    // p[1], p[2] may lie in the mirror
    const uint32_t sh = (pos & 3) * 8;
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    lo = __builtin_amdgcn_alignbit(d1, d0, sh);
    hi = __builtin_amdgcn_alignbit(d2, d1, sh);
}
template <uint32_t WB>
__device__ __forceinline__ uint32_t fetch4(const uint32_t *win32, uint32_t pos) {
    const uint32_t *p = win32 + ((pos & (WB - 1)) >> 2);
    return __builtin_amdgcn_alignbit(p[1], p[0], (pos & 3) * 8);
}

struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };   // 16 bytes at any byte address
typedef uint32_t u32u __attribute__((aligned(1)));
// 16 bytes at any byte address as ONE register tuple: the load writes it in place (the components of the struct above get moved behind the load,
// which puts a wait for the load right there)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) W12 { uint32_t x, y, z; };   // four 3-byte words = 12 bytes at a dword-aligned address (a 3-element vector type would be 16 bytes wide)
__device__ __forceinline__ v4u ld16u(const uint8_t *p) { v4u v; __builtin_memcpy(&v, p, 16); return v; }

__device__ __forceinline__ uint4 load_chunk_tail(const uint8_t *seg, uint32_t i, uint32_t seg_len) {    // 16 bytes at i, zeros behind the segment's end
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16; k++) if (i + k < seg_len) w[k >> 2] |= (uint32_t)seg[i + k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Wave-cooperative extension of a match whose first L0 bytes are known to agree: q, c, L0, lim are wave-uniform; returns the
// full length (<= lim).  64 lanes x 4 bytes per step.  FARC: the candidate lies outside the LDS window, its bytes come from the
// segment in HBM / L2 (c + lim < q, so every address is inside the segment).
template <bool FARC, uint32_t WB>
__device__ __forceinline__ uint32_t lz_extend(const uint32_t *win32, const uint8_t *seg, uint32_t q, uint32_t c, uint32_t L0, uint32_t lim, uint32_t lane) {
    uint32_t L = L0;
    for (;;) {
        uint32_t pos = L + lane * 4;
        const uint32_t cw = FARC ? *(const u32u *)(seg + c + pos) : fetch4<WB>(win32, c + pos);
        uint32_t x = fetch4<WB>(win32, q + pos) ^ cw;
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

// Pre-warm of a UNIT's hash table (latency mode: a segment is cut into units, one workgroup each, so that a small batch fills the chip): the table
// gets the state the segment-long walk has when it reaches position `end` -- every position q < end with q + 8 <= seg_len (with even_only: every
// second one) entered, the latest position of a slot wins.  Inserts are ds_max_u32, so their order does not matter and the bytes can come straight
// from memory, eight consecutive positions per lane and 16-byte load, without the window and without barriers.  The unit's matches are therefore
// those of the segment-long walk, bit for bit.
// GLOG: 0 = the LDS table of ENT slots (slot = mulhi(hash, entries)); otherwise the table lies in global memory and has 1 << GLOG slots
// (slot = the hash's top bits): the strong level set of the zstd encoder (k_lz_split.hip)
template <uint32_t GLOG, uint32_t ENT>
__device__ __forceinline__ uint32_t lz_slot(uint32_t h32) { return GLOG ? h32 >> (32 - (GLOG ? GLOG : 1)) : __umulhi(h32, ENT); }
template <uint32_t GLOG, uint32_t ENT>
__device__ __forceinline__ void lz_prewarm(uint32_t *table, const uint8_t *seg, uint32_t seg_len, uint32_t end, bool ins_all, uint32_t tid) {
#pragma unroll 2
    for (uint32_t p = tid * 8; p < end; p += LZ_THREADS * 8) {
        uint32_t w[4];
        if (p + 16 <= seg_len) { const uint2 a = *(const uint2 *)(seg + p), b = *(const uint2 *)(seg + p + 8); w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y; }
        else { const uint4 v = load_chunk_tail(seg, p, seg_len); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (!ins_all && (j & 1)) continue;
            const uint32_t q = p + j;
            const uint32_t lo = (j & 3) ? __builtin_amdgcn_alignbyte(w[(j >> 2) + 1], w[j >> 2], j & 3) : w[j >> 2];
            const uint32_t hi = (j & 3) ? __builtin_amdgcn_alignbyte(w[(j >> 2) + 2 < 4 ? (j >> 2) + 2 : 3], w[(j >> 2) + 1], j & 3) : w[(j >> 2) + 1];
            const uint32_t h32 = lo * 0x9E3779B1u + (hi & 0xFFFFu) * 0x85EBCA6Bu;
            if (q < end && q + 8 <= seg_len) atomicMax(&table[lz_slot<GLOG, ENT>(h32)], ((q + 1) << TAG_BITS) | ((h32 >> 6) & TAG_MASK));
        }
    }
}

// The same for the PACKED table.  Its inserts are not order-free -- of a tile's contenders for one word only one is stored --, so the pre-warm replays
// the walk's tiles (aligned 4 096 positions: blocks are multiples of that) one by one: every thread reads the words of its two even positions, barrier,
// one ds_max_u64 each, barrier.  The next tile's bytes are requested before the current tile is stored (two tiles in flight): ~0.3 us per tile.
template <uint32_t NW>
__device__ __forceinline__ void lz_prewarm3(uint64_t *table64, const uint8_t *seg, uint32_t seg_len, uint32_t end, uint32_t tid) {
    auto fetch = [&](uint32_t t0, uint32_t (&w)[3]) {
        const uint32_t p = t0 + 4 * tid;                                              // positions p and p + 2: bytes p .. p + 9
        w[0] = w[1] = w[2] = 0;
        if (p + 12 <= seg_len) { w[0] = *(const uint32_t *)(seg + p); w[1] = *(const uint32_t *)(seg + p + 4); w[2] = *(const uint32_t *)(seg + p + 8); }
        else if (p < seg_len) { const uint4 v = load_chunk_tail(seg, p, seg_len); w[0] = v.x; w[1] = v.y; w[2] = v.z; }
    };
    uint32_t wa[3], wb[3];
    fetch(0, wa);
    for (uint32_t t0 = 0; t0 < end; t0 += 4 * LZ_THREADS) {
        if (t0 + 4 * LZ_THREADS < end) fetch(t0 + 4 * LZ_THREADS, wb);
        uint32_t widx[2], sh[2], v[2]; uint64_t w64[2]; bool ok[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t q = t0 + 4 * tid + 2 * k;
            const uint32_t lo = k ? __builtin_amdgcn_alignbyte(wa[1], wa[0], 2) : wa[0], hi = k ? __builtin_amdgcn_alignbyte(wa[2], wa[1], 2) : wa[1];
            const uint32_t h32 = lo * 0x9E3779B1u + (hi & 0xFFFFu) * 0x85EBCA6Bu;
            t3_slot<NW>(h32, widx[k], sh[k]);
            ok[k] = q < end && q + 8 <= seg_len && q != 0;
            v[k] = t3_entry(q, t3_tag(h32));
            w64[k] = table64[widx[k]];
        }
        __syncthreads();                                                                // every thread has read the words as the tile's look-ups see them
#pragma unroll
        for (int k = 0; k < 2; k++) if (ok[k]) t3_max(&table64[widx[k]], t3_put(w64[k], sh[k], t3_field(w64[k], sh[k]), v[k]));
        __syncthreads();
        wa[0] = wb[0]; wa[1] = wb[1]; wa[2] = wb[2];
    }
}

__device__ __forceinline__ uint4 load_chunk(const uint8_t *seg, uint32_t i, uint32_t seg_len) {
    if (i + 16 <= seg_len) return *(const uint4 *)(seg + i);
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16; k++) if (i + k < seg_len) w[k >> 2] |= (uint32_t)seg[i + k] << (8 * (k & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

} // namespace pna
