// k_lz_split.hip -- the default form of the LZ stage for gfx950 (CDNA4): two kernels that meet in a workspace of one word per position.
//   k_lzm: look-up, match, backward adoption, inserts (the match finder of k_lz.hip with four consecutive positions per lane)
//   k_lzp: greedy parse, merge, emission of sequences and literals with one LANE per parse region
// Replaces, together, the match finder + parser inside the third-party encoder the reference drives at lib/src/compress.rs:32-41
// (CompressionWriter::write -> ZstdEncoder::write / ZlibEncoder::write).  Same results as k_lz (k_lz.hip), which stays as the one-kernel
// form for short runs; DESIGN.md section 5 "Round 2, second half" has the measurements.  Integer / byte work, no MFMA.
#include <type_traits>
#include "lz_common.h"

namespace pna {


// ---------------------------------------------------------------------------------------------------------------------------
// k_lzm -- the match half of the split form: look-up, match, backward adoption and the tile's inserts, as in k_lz<MODE 1>, but with
// FOUR CONSECUTIVE POSITIONS PER LANE (lane i of wave w: positions t0 + 256 w + 4 i + j, j = 0..3) instead of one position per lane and
// group.  Nothing in this half needs a ballot over a group's positions, and with consecutive positions
//   * the 36 + 4 bytes around a lane's positions are ten aligned dwords, loaded once; the 8 / 16 / 32 bytes at position j are
//     v_alignbyte with a constant (j = 0: the registers themselves) -- k_lz loads and aligns them per position;
//   * the right neighbours of the adoption rounds sit in the same lane, except across the lane border (DPP row_shl: a row of
//     16 lanes is a group of 64 positions, and the zero fill at the row's end is the rule "adoption stops at the group border");
//     the offset moves along with every adoption instead of one ds_bpermute at the end;
//   * the four words of a lane go out as one 16-byte store; with even positions only, the inserts are those of j = 0 and 2;
//   * a far candidate's bytes (all 36 the match step can ask for) are requested only on the lanes that hold one, into register tuples.
// Same table, window, tile order and barriers as k_lz, hence the same words (tests/test_gpu_parity.py: forms of the LZ stage).
#define DPP_ROW_SHL1(v) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x101, 0xF, 0xF, true))   // value of lane i + 1 inside the row of 16 (0 at its end)
// The match key of k_lzm: length << 26 | offset << 6 | back << 3 in ONE register (offsets < 2^20, lengths <= 36 + 7, back <= 7), so that an adoption round
// moves one value per position (one DPP, one select) instead of key and offset: taking over the match of position q + s is + (s << 26) - (s << 3), and
// "strictly longer" is a compare against the own key with everything below the length set.
// STRONG = 2 (the high and max zstd sets, round 5: a fourth adoption round over EIGHT positions and up to 15 back bytes): the back count takes four bits, << 2 (PKB).
constexpr uint32_t PK_LEN = 26, PK_OFF = 6, PK_LOW = (1u << PK_LEN) - 1;
__host__ __device__ constexpr uint32_t pkb_of(int strong) { return strong == 2 ? 2u : 3u; }
__device__ __forceinline__ uint32_t pk_make(uint32_t l, uint32_t off, uint32_t bkf) { return (l << PK_LEN) | ((off << PK_OFF) | bkf); }   // bkf: back << PKB
template <uint32_t S, uint32_t PKB = 3>
__device__ __forceinline__ uint32_t pk_adopt(uint32_t P, uint32_t Pn) {            // Pn: the key of position q + S
    constexpr uint32_t BM = PKB == 2 ? 0x3Cu : 0x38u;
    const uint32_t T = Pn + ((S << PK_LEN) - (S << PKB));
    const bool a = (Pn & (BM & ~((S - 1) << PKB))) != 0 && T > (P | PK_LOW);        // back >= S (S = 1, 2, 4, 8: a test of the upper back bits)
    return a ? T : P;
}
#define DPP_ROW_SHL2(v) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x102, 0xF, 0xF, true))   // value of lane i + 2 inside the row of 16 (0 at its end)
// back << 2 for up to 15 back bytes: x1 .. x4 = position XOR candidate of the four dwords before them, nearest first (the byte right before in x1's top byte); the equal bytes
// from the top of x1 down, the sixteenth never counted -- the leading zero bits of the 128, the lowest byte of x4 forced to differ (first_diff16 upside down)
__device__ __forceinline__ uint32_t ffbh_hw(uint32_t x) { uint32_t r; asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x)); return r; }    // 0xFFFFFFFF for 0
__device__ __forceinline__ uint32_t back15_field(uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4) {
    const uint32_t f1 = ffbh_hw(x1), f2 = __builtin_elementwise_add_sat(ffbh_hw(x2), 32u), f3 = __builtin_elementwise_add_sat(ffbh_hw(x3), 64u), f4 = ffbh_hw(x4 | 0xFFu) + 96u;
    uint32_t n, m;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(n) : "v"(f2), "v"(f3), "v"(f4));
    asm("v_min_u32 %0, %1, %2" : "=v"(m) : "v"(f1), "v"(n));
    return (m >> 1) & 0x3Cu;
}
// GLOG != 0: the hash table of this workgroup lies in GLOBAL memory, 1 << GLOG slots (gtab + blockIdx.x << GLOG): the zstd levels 10 .. 22.  What a
// 1 MiB segment's match finder can remember is what sets the ratio on text (24 512 slots in LDS 2.70, 2^19 slots 2.96), and LDS
// cannot hold more; the high levels trade speed for it, as the reference's do.  Look-ups are loads that bypass the CU's L1 (agent-scope atomic loads:
// the table is only ever modified by atomics, which execute in L2), inserts are global atomics, and a tile's inserts are waited for (vmcnt) before
// the barrier that lets the next tile's look-ups go.
constexpr uint32_t GTAB_LOG = 19;
// ---- k_lzm's far candidates (see the kernel): pair s of the wave's far (lane, j) pairs hands position and offset to lane s - r0 (lane 63: the others' pushes)
// (one word: owner lane | j << 6 | offset << 8 -- offsets are at most 2^20; the position follows from lane and j, q0w = the wave's first position of the tile)
__device__ __forceinline__ void far_push(const bool (&farj)[4], const uint32_t (&idx)[4], uint32_t r0, uint32_t q0w, uint32_t lane, const uint32_t (&off)[4], uint32_t &sq, uint32_t &so) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t d = idx[j] - r0;
        const int dst = (int)((farj[j] && d < 63u) ? d : 63u) * 4;
        w |= (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)(lane | (uint32_t)j << 6 | off[j] << 8));   // (a lane is the target of at most one pair; lanes nobody writes to get 0)
    }
    sq = q0w + 4 * (w & 63u) + ((w >> 6) & 3u); so = w >> 8;
}
// the candidate's bytes c - 4 .. c + 32 (STRONG: from c - 8) from the segment in HBM / L2
template <bool ON, int STRONG>
__device__ __forceinline__ void far_load(const uint8_t *seg, uint32_t c, v4u &fa, uint32_t &fb, v4u &fd, uint32_t &fc, v2u &fe) {
    fe = 0;
    if (!ON) { fa = 0; fd = 0; fb = fc = 0; return; }
    const uint8_t *pc = seg + c - 4;
    fa = ld16u(pc); fb = *(const u32u *)(pc + 16); fd = ld16u(pc + 20);
    fc = 0;
    if (STRONG) fc = *(const u32u *)(pc - 4);
    if (STRONG == 2) __builtin_memcpy(&fe, pc - 12, 8);                            // bytes c - 16 .. c - 9 (a usable candidate lies at 16 or beyond)
}
// the owners take their results from the lanes that computed them
__device__ __forceinline__ void far_pull(const bool (&farj)[4], const uint32_t (&idx)[4], uint32_t r0, uint32_t Kf, uint32_t (&K)[4]) {   // K: packed keys (pk_make)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t d = idx[j] - r0;
        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((d < 63u ? d : 63u) * 4), (int)Kf);
        if (farj[j] && d < 63u) K[j] = got;
    }
}
// the match step of position q against those bytes: the packed key (pk_make) with the offset so, exactly as for a candidate inside the window
template <int STRONG, uint32_t WB>
__device__ __forceinline__ uint32_t far_match(const uint32_t *win32, uint32_t q, uint32_t so, v4u fa, uint32_t fb, v4u fd, uint32_t fc, v2u fe, bool edge, uint32_t blk_end) {
    // the position's bytes q - 8 .. q + 36 from ONE base address (the dword that holds q - 8; what lies behind the window's end is its mirror):
    // pq[0] = q - 8 .., pq[1] = q - 4 .., pq[2 ..] = q ..
    lds_cu32 *pq = lds_word(L_WIN + ((q - 8) & (WB - 4)));
    const uint32_t shq = q << 3;                                                    // (v_alignbit takes the shift modulo 32)
    uint32_t E[10];
#pragma unroll
    for (int k = 0; k < 10; k++) E[k] = pq[k + 1];
#define EW(k) __builtin_amdgcn_alignbit(E[(k) + 1], E[k], shq)                     /* bytes q - 4 + 4 k .. + 3 */
    const uint32_t x0 = EW(1) ^ fa.y, x1 = EW(2) ^ fa.z, x2 = EW(3) ^ fa.w, x3 = EW(4) ^ fb;
    uint32_t l = first_diff16(x0, x1, x2, x3);
    if (l == 16) {
        const uint32_t y0 = EW(5) ^ fd.x, y1 = EW(6) ^ fd.y, y2 = EW(7) ^ fd.z, y3 = EW(8) ^ fd.w;
        l = 16 + first_diff16(y0, y1, y2, y3);
    }
    if (edge) { const uint32_t lim = blk_end - q; l = l < lim ? l : lim; }
    const uint32_t xk = EW(0) ^ fa.x;
    uint32_t bk3 = (uint32_t)__builtin_clz(xk | 0xFFu) & 24u;                      // back << 3
    if (STRONG == 1 && xk == 0) bk3 = 32u + ((uint32_t)__builtin_clz((__builtin_amdgcn_alignbit(E[0], pq[0], shq) ^ fc) | 0xFFu) & 24u);
    if (STRONG == 2) {
        lds_cu32 *pe = lds_word(L_WIN + ((q - 16) & (WB - 4)));                      // the dwords of q - 16 and q - 12 (pq[0] holds q - 8)
        const uint32_t e0 = pe[0], e1 = pe[1], p0 = pq[0];
        bk3 = back15_field(xk, __builtin_amdgcn_alignbit(E[0], p0, shq) ^ fc, __builtin_amdgcn_alignbit(p0, e1, shq) ^ fe.y, __builtin_amdgcn_alignbit(e1, e0, shq) ^ fe.x);
    }
#undef EW
    return STRONG ? (l >= MIN_MATCH ? pk_make(l, so, bk3) : 0u) : pk_make(l >= MIN_MATCH ? l : 0u, so, bk3);
}

template <bool DEFL, int STRONG, uint32_t GLOG, uint32_t WLOG, bool FARP, bool TAB3>   // STRONG: 0 = two adoption rounds / 3 back bytes, 1 = + the round over four positions / 7 back bytes, 2 = + the round over eight / 15
__global__ __launch_bounds__(LZ_THREADS)
void k_lzm(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, uint32_t flags, uint32_t max_off, uint32_t *__restrict__ pbuf, uint32_t blk0,
           uint32_t *__restrict__ gtab) {
    constexpr uint32_t RW = 256, TILE_G = RW * LZ_WAVES, PKB = pkb_of(STRONG), MIN_C = STRONG == 2 ? 16u : 8u;   // MIN_C: a usable candidate lies at this position or beyond (its back bytes exist)
    constexpr bool FAR = !DEFL && FARP;                     // deflate offsets (<= 32 KiB) never leave the LDS window; FARP = false: a zstd launch whose look-back ends with the window (the fast set)
    using GEO = LzGeo<WLOG, TAB3>;                          // (lz_common.h) these names hide the 64 KiB geometry's constants of pna_dev.h
    constexpr uint32_t WIN_BYTES = GEO::WIN, HASH_ENTRIES = GEO::ENTRIES, L_TABLE = GEO::L_TABLE, NW3 = GEO::WORDS3;
    // candidates at most NEAR back are verified in the window: the window gives way to the next chunk only BEHIND the tile's first barrier (round 4), so its look-back is a tile
    // longer than the one-kernel form's (GEO::NEAR): 27 392 (+ 976, below) bytes with the 32 KiB window (far pairs per wave and tile 56.9 -> 52.6; same results -- NEAR only says where the bytes come from)
    // ... and this kernel reads at most 40 bytes past a position (the parse kernel extends long matches from memory), so its window runs LA = 64 bytes ahead of the tile instead of
    // the one-kernel form's LOOKAHEAD + 16: another 976 bytes of look-back
    constexpr uint32_t LA = 64;
    constexpr uint32_t NEAR = GEO::NEAR + TILE_G + (LOOKAHEAD + 16 - LA);
    static_assert(NEAR == GEO::NEARM, "lz_common.h: the one-kernel form numbers this kernel's far candidates (FLAG_FAR1)");
    static_assert(WIN_BYTES >= TILE_G + LA + NEAR + 8 + 200 && (!DEFL || NEAR >= 32768), "window: look-back (+ 8 back bytes) + this tile + look-ahead");
    static_assert(!TAB3 || (GLOG == 0 && !DEFL && FAR), "the packed table: zstd sets with the table in LDS and far candidates");
    static_assert(LZ_G_ZSTD == 4 && LZ_G_DEFLATE == 4 && WIN_MIRROR >= 40, "k_lzm: four positions per lane, 36 bytes read behind a lane's first position");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *win32 = (uint32_t *)(lds + L_WIN);
    uint32_t *table = GLOG ? gtab + ((size_t)blockIdx.x << GLOG) : (uint32_t *)(lds + L_TABLE);
    uint64_t *table64 = (uint64_t *)(lds + L_TABLE);        // TAB3: the packed table, three 21-bit entries per word (lz_common.h)
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = uni(tid >> 6);
    const SegDesc sd = segs[blockIdx.x];
    const uint8_t *seg = src + sd.src_off;
    const uint32_t seg_len = sd.len;
    if ((flags & FLAG_HAS_SMALL) && seg_len <= MID_SEG) return;       // (uniform) a short segment: k_lzms's
    if (!lds_base_is_zero(lds)) __builtin_trap();                     // (lds_word: the window's words are addressed from 0)
    const uint32_t blk_log = sd.blk_log, bsz = 1u << blk_log;
    // The words: one per position, length | offset.  With the table in LDS (GLOG = 0) they take THREE bytes -- 5 bits for the length (0, or length - 5
    // for 6 .. 36: adopted lengths beyond 36 are clamped, FLAG_LEN36 tells the one-kernel form to do the same) and 19 for the offset (MAX_OFF_W3) --:
    // the parse kernel's time is the time to stream them (42 GB per step at 4 bytes).  With the table in global memory: 4 bytes, 6 + 20 bits.
    constexpr bool W3 = GLOG == 0;
    uint32_t *pb = pbuf + ((size_t)(sd.blk_base - blk0) << blk_log);
    uint8_t *pb8 = (uint8_t *)pbuf + 3 * ((size_t)(sd.blk_base - blk0) << blk_log);
    const bool adopt = (flags & F_ADOPT) != 0, ins_all = !(flags & F_INS2);

    for (uint32_t i = tid; i < (GLOG ? (4u << GLOG) : GEO::TABLE_BYTES) / 16; i += LZ_THREADS) ((uint4 *)table)[i] = make_uint4(0, 0, 0, 0);
    if (GLOG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (the zeros are in L2 before the pre-warm's atomics and the first look-ups)
    // a unit that starts inside the segment (latency mode): window and table as the segment-long walk has them there (k_lz.hip, lz_common.h)
    uint32_t loaded_end = sd.u0 + TILE_G + LA;
    __syncthreads();
    if (sd.u0) {
        if constexpr (TAB3) lz_prewarm3<NW3>(table64, seg, seg_len, sd.u0, tid);
        else { lz_prewarm<GLOG, HASH_ENTRIES>(table, seg, seg_len, sd.u0, ins_all, tid); if (GLOG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    for (uint32_t i = (loaded_end > WIN_BYTES ? loaded_end - WIN_BYTES : 0u) + tid * 16; i < loaded_end; i += LZ_THREADS * 16) {
        const uint4 v = load_chunk(seg, i, seg_len);
        const uint32_t wo = i & (WIN_BYTES - 1);
        *(uint4 *)(lds + L_WIN + wo) = v;
        if (wo < WIN_MIRROR) *(uint4 *)(lds + L_WIN + WIN_BYTES + wo) = v;
    }
    __syncthreads();
    uint4 pf = make_uint4(0, 0, 0, 0);

    for (uint32_t blk_start = sd.u0; blk_start < sd.u1; blk_start += bsz) {
        const uint32_t blk_end = (seg_len - blk_start < bsz) ? seg_len : blk_start + bsz;
        for (uint32_t t0 = blk_start; t0 < blk_end; t0 += TILE_G) {
            const uint32_t t1 = (blk_end - t0 < TILE_G) ? blk_end : t0 + TILE_G;
            if (tid < TILE_G / 16) { pf = (loaded_end + tid * 16 < seg_len) ? load_chunk(seg, loaded_end + tid * 16, seg_len) : make_uint4(0, 0, 0, 0); }
            const bool tile_full = (t1 - t0 == TILE_G) && (t0 + TILE_G + 8 <= seg_len) && (blk_end - t0 >= TILE_G + CAP1);
            // The tile's work is instantiated twice: FULL = every position of the tile is inside the block, at least 8 bytes before the segment's end and far enough
            // from the block's end that no match step can reach it (all tiles but a block's / segment's last, and never tile 0 with the packed table: position 0 is
            // not stored) -- no validity masks, no selects, no bounds on the stores, no clamp at the block's end --, and the general form.  (uniform branch)
            auto tile_body = [&](auto full_t) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_t)::value;
            const uint32_t q0 = t0 + wave * RW + 4 * lane;
            // ---- the bytes around the lane's positions: D[k] = bytes q0 + 4 k .. + 3, Dm = the 4 (8) before q0.  ONE base address (the dword of q0 - 8): what
            // lies behind the window's end is its mirror (48 bytes: the base + 44)
            uint32_t D[9], Dm1, Dm2 = 0, Dm3 = 0, Dm4 = 0;
            {
                lds_cu32 *pq = lds_word(L_WIN + ((q0 - 8) & (WIN_BYTES - 1)));
#pragma unroll
                for (int k = 0; k < 9; k++) D[k] = pq[k + 2];
                Dm1 = pq[1];
                if (STRONG) Dm2 = pq[0];
                if (STRONG == 2) { lds_cu32 *pe = lds_word(L_WIN + ((q0 - 16) & (WIN_BYTES - 1))); Dm4 = pe[0]; Dm3 = pe[1]; }
            }
#define QW(k, j) ((j) ? __builtin_amdgcn_alignbyte(D[(k) + 1], D[k], (j)) : D[k])          /* 4 bytes at position j, + 4 k */
            // ---- look-up
            uint32_t hsh[4], tag[4], ent[4];
            uint32_t sh3[4] = {0, 0, 0, 0}; uint64_t w3a = 0, w3b = 0;                  // TAB3: the fields' bit positions; the words of positions 0 and 2 as the look-ups saw them
            bool hv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t q = q0 + j;
                hv[j] = FULL || ((q < t1) && (q + 8 <= seg_len));
                const uint32_t h32 = QW(0, j) * 0x9E3779B1u + (QW(1, j) & 0xFFFFu) * 0x85EBCA6Bu;
                if constexpr (TAB3) {
                    t3_slot<NW3>(h32, hsh[j], sh3[j]);
                    tag[j] = t3_tag(h32);
                    const uint64_t w = table64[hsh[j]];
                    if (j == 0) w3a = w;
                    if (j == 2) w3b = w;
                    const uint32_t e = t3_field(w, sh3[j]);
                    ent[j] = hv[j] ? e : 0u;
                } else {
                hsh[j] = lz_slot<GLOG, HASH_ENTRIES>(h32);
                tag[j] = (h32 >> 6) & TAG_MASK;
                if (GLOG) ent[j] = hv[j] ? __hip_atomic_load(&table[hsh[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                else { const uint32_t e = table[hsh[j]]; ent[j] = hv[j] ? e : 0u; }     // (the LDS read goes out for every lane -- any slot is readable --, a select instead of an exec-masked region)
                }
            }
            // ---- candidates (rules as in k_lz)
            uint32_t off[4];
            bool farj[4], nearj[4];
            uint64_t fmj[4] = {0, 0, 0, 0};                                          // TAB3: the far lanes' masks
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if constexpr (TAB3) {
                    // an entry = (position / 2) << 2 | tag: usable iff its position is >= 8 (entry >= 16: the empty entry 0 included), the tag agrees, the offset fits.
                    // Tag and offset in ONE subtraction (round 5): (q / 2 - 1) << 2 | my tag, less the entry, is 4 r1 with r1 = (offset - 2 - (j & 1)) / 2 where the tags agree
                    // and has a low bit set where they do not; rotated right by two, r1 or something >= 2^30.  off[j] is only read under nearj / farj.
                    const uint32_t dd = ((2u * q0 + 4u * (uint32_t)(j >> 1) - 4u) | tag[j]) - ent[j];
                    const uint32_t r1 = __builtin_amdgcn_alignbit(dd, dd, 2);
                    // (the three compares' lane masks are combined as SCALARS: a ballot of `a & b` makes the compiler build the boolean on the lanes first)
                    const uint64_t mv = __builtin_amdgcn_ballot_w64(r1 < ((max_off - (uint32_t)(j & 1)) >> 1)) & __builtin_amdgcn_ballot_w64(ent[j] >= 2u * MIN_C);
                    const uint64_t mn = __builtin_amdgcn_ballot_w64(r1 < ((NEAR - 1u) >> 1));   // (offset <= NEAR - 1 + (j & 1): inside the window)
                    fmj[j] = mv & ~mn;
                    nearj[j] = __builtin_amdgcn_inverse_ballot_w64(mv & mn);
                    farj[j] = __builtin_amdgcn_inverse_ballot_w64(fmj[j]);
                    off[j] = 2u * r1 + 2u + (uint32_t)(j & 1);
                } else {
                const uint32_t c1 = ent[j] >> TAG_BITS, o = q0 + j + 1 - c1;
                off[j] = ((c1 > MIN_C) & ((ent[j] & TAG_MASK) == tag[j]) & (o <= max_off)) ? o : 0u;  // (& not &&: no short-circuit branches)
                farj[j] = FAR && off[j] > NEAR;
                nearj[j] = (off[j] != 0) & !farj[j];
                }
            }
            // ---- the far candidates (more than NEAR bytes back: outside the window) are handled COMPACTED.  A vector load costs the CU's address unit 16
            // cycles per instruction whatever the number of active lanes, and nearly every wave and position j holds a far lane or two (3 - 6 % of the
            // positions): loaded in place, 12 loads per wave and tile were 22 % of the kernel.  So the wave's far pairs (lane, j) are numbered (ballot +
            // mbcnt), pair s hands its position and offset to lane s (ds_permute), lane s requests the candidate's 4 (8) + 32 bytes from the segment --
            // three loads per wave and tile, in flight during the match step of the near candidates --, then reads the position's bytes from the window,
            // runs the same match step on them, and the owner pulls the result (ds_bpermute).  63 pairs per round (lane 63 takes the pushes of the
            // lanes without a pair); more than 63: further rounds behind the first.
            // (The first round's loads are issued whether the wave holds a far pair or not -- nearly every wave does --, lanes without a pair reading the
            // segment's first bytes: a load under a branch makes the compiler copy the loaded tuple behind the branch, with a wait for it right there.)
            uint32_t idx[4] = {0, 0, 0, 0}, npair = 0, sq = 0, so = 0;
            if (FAR) {
                uint64_t fm[4];
#pragma unroll
                for (int j = 0; j < 4; j++) fm[j] = TAB3 ? fmj[j] : __builtin_amdgcn_ballot_w64(farj[j]);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    idx[j] = npair + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm[j], 0u));
                    npair += (uint32_t)__builtin_popcountll(fm[j]);
                }
                npair = uni(npair);
                far_push(farj, idx, 0u, t0 + wave * RW, lane, off, sq, so);
            }
            const bool slot0 = FAR && lane < 63u && lane < npair;
            v4u ffa, ffd; uint32_t ffb, ffc; v2u ffe;
            far_load<FAR, STRONG>(seg, slot0 ? sq - so : 16u, ffa, ffb, ffd, ffc, ffe);
            // ---- match: the candidates inside the window.  P[j] = the packed key (pk_make): length, offset, back bytes
            // (Round 4, measured and dropped: the bytes 16 .. 35 compared ONCE per lane and offset in a wave-uniform loop -- a lane's positions inside a long
            // match share the offset, position j's length is position A's less j - A --: bit-exact, and 11 % SLOWER than the four in-place compares below;
            // the loop's dynamic selects and its second and third turns cost more than the 80 instructions it saves.)
            uint32_t P[4];
            const uint32_t q8[4] = {q0 - 8, q0 - 7, q0 - 6, q0 - 5};
            const bool edge = !FULL && blk_end - (t0 + wave * RW) < RW + CAP1;          // (uniform) only the block's last waves can run into its end
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t o = off[j], q = q0 + j;
                uint32_t Pj = 0;
                if (nearj[j]) {
                    uint32_t l, bk3;
                    // the candidate's bytes c - 8 .. c + 36 from ONE base address (the dword of c - 8; the mirror behind the window's end covers the base + 44):
                    // pc[0] = c - 8 .., pc[1] = c - 4 .., pc[2 .. 6] the first 16 (+ 4) bytes, pc[6 .. 10] the second
                    const uint32_t c8 = q8[j] - o;                                      // c - 8
                    const uint32_t shc = c8 << 3;                                       // (v_alignbit takes the shift modulo 32: that of c)
                    lds_cu32 *pc = lds_word(L_WIN + (c8 & (WIN_BYTES - 4)));
                    const uint32_t dm = pc[1], d0 = pc[2], d1 = pc[3], d2 = pc[4], d3 = pc[5], d4 = pc[6];
                    const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, shc), w1 = __builtin_amdgcn_alignbit(d2, d1, shc);
                    const uint32_t w2 = __builtin_amdgcn_alignbit(d3, d2, shc), w3 = __builtin_amdgcn_alignbit(d4, d3, shc);
                    const uint32_t bc = __builtin_amdgcn_alignbit(d0, dm, shc);
                    uint32_t bc2 = 0;
                    if (STRONG) bc2 = __builtin_amdgcn_alignbit(dm, pc[0], shc);
                    const uint32_t x0 = QW(0, j) ^ w0, x1 = QW(1, j) ^ w1, x2 = QW(2, j) ^ w2, x3 = QW(3, j) ^ w3;
                    l = first_diff16(x0, x1, x2, x3);
                    if (l == 16) {
                        const uint32_t f1 = pc[7], f2 = pc[8], f3 = pc[9], f4 = pc[10];
                        const uint32_t v0 = __builtin_amdgcn_alignbit(f1, d4, shc), v1 = __builtin_amdgcn_alignbit(f2, f1, shc);
                        const uint32_t v2 = __builtin_amdgcn_alignbit(f3, f2, shc), v3 = __builtin_amdgcn_alignbit(f4, f3, shc);
                        const uint32_t y0 = QW(4, j) ^ v0, y1 = QW(5, j) ^ v1, y2 = QW(6, j) ^ v2, y3 = QW(7, j) ^ v3;
                        l = 16 + first_diff16(y0, y1, y2, y3);
                    }
                    if (edge) { const uint32_t lim = blk_end - q; l = l < lim ? l : lim; }
                    const uint32_t bqj = j ? __builtin_amdgcn_alignbyte(D[0], Dm1, j) : Dm1;       // the 4 bytes before q (q - 1 in the top byte)
                    const uint32_t xk = bqj ^ bc;
                    bk3 = (uint32_t)__builtin_clz(xk | 0xFFu) & 24u;                               // back << 3
                    if (STRONG == 1 && xk == 0) { const uint32_t bq2 = j ? __builtin_amdgcn_alignbyte(Dm1, Dm2, j) : Dm2; bk3 = 32u + ((uint32_t)__builtin_clz((bq2 ^ bc2) | 0xFFu) & 24u); }
                    if (STRONG == 2) {                                                             // up to 15 back bytes: the dwords of c - 16 and c - 12 from a base of their own (the mirror covers base + 44)
                        lds_cu32 *pe = lds_word(L_WIN + ((c8 - 8) & (WIN_BYTES - 4)));
                        const uint32_t e0 = pe[0], e1 = pe[1], p0 = pc[0];
                        const uint32_t bq2 = j ? __builtin_amdgcn_alignbyte(Dm1, Dm2, j) : Dm2, bq3 = j ? __builtin_amdgcn_alignbyte(Dm2, Dm3, j) : Dm3, bq4 = j ? __builtin_amdgcn_alignbyte(Dm3, Dm4, j) : Dm4;
                        bk3 = back15_field(xk, bq2 ^ bc2, bq3 ^ __builtin_amdgcn_alignbit(p0, e1, shc), bq4 ^ __builtin_amdgcn_alignbit(e1, e0, shc));      // back << 2
                    }
                    // (a length below MIN_MATCH counts as 0; the sets with the third adoption round then drop the whole key -- 0 + 1 + 2 + 4 adopted bytes would be a match)
                    Pj = STRONG ? (l >= MIN_MATCH ? pk_make(l, o, bk3) : 0u) : pk_make(l >= MIN_MATCH ? l : 0u, o, bk3);
                }
                P[j] = Pj;
            }
            // ---- the far pairs' match step (first round: the bytes requested above have had the near candidates' match step to arrive)
            if (FAR && npair) {
                {
                    uint32_t Kf = 0;
                    if (slot0) Kf = far_match<STRONG, WIN_BYTES>(win32, sq, so, ffa, ffb, ffd, ffc, ffe, edge, blk_end);
                    far_pull(farj, idx, 0u, Kf, P);
                }
                if (!(flags & FLAG_FAR1))                                               // (uniform; FLAG_FAR1, the default and light sets since round 5: the pairs beyond the first 63 are dropped -- - 0.16 % of ratio, - 6 % of the kernel's time)
                for (uint32_t r0 = 63; r0 < npair; r0 += 63) {                         // (more than 63 far pairs in 256 positions: a quarter of the wave-tiles of the default set)
                    uint32_t sq2, so2, Kf = 0;
                    far_push(farj, idx, r0, t0 + wave * RW, lane, off, sq2, so2);
                    if (lane < 63u && lane < npair - r0) {
                        v4u ga, gd; uint32_t gb, gc; v2u ge;
                        far_load<true, STRONG>(seg, sq2 - so2, ga, gb, gd, gc, ge);
                        Kf = far_match<STRONG, WIN_BYTES>(win32, sq2, so2, ga, gb, gd, gc, ge, edge, blk_end);
                    }
                    far_pull(farj, idx, r0, Kf, P);
                }
            }
            // ---- backward adoption on the packed keys (the offset goes along inside the key)
            if (adopt) {
                if (STRONG == 2) {   // the high / max sets (round 5): a round over EIGHT positions first (the model's 0x2148) -- the same position two lanes on, eight bytes longer
                    uint32_t P8[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) P8[j] = DPP_ROW_SHL2(P[j]);
#pragma unroll
                    for (int j = 0; j < 4; j++) P[j] = pk_adopt<8, PKB>(P[j], P8[j]);
                }
                if (STRONG) {   // the strong sets' round over four positions comes FIRST (round 5: rounds 4, 1, 2 -- the model's 0x214 --: + 0.03 % of ratio for nothing): the same position of the next lane, four bytes longer
                    uint32_t P4[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) P4[j] = DPP_ROW_SHL1(P[j]);
#pragma unroll
                    for (int j = 0; j < 4; j++) P[j] = pk_adopt<4, PKB>(P[j], P4[j]);
                }
                {   // the right neighbour's match, one byte longer
                    const uint32_t Pn = DPP_ROW_SHL1(P[0]);
                    const uint32_t P1[4] = {P[1], P[2], P[3], Pn};
#pragma unroll
                    for (int j = 0; j < 4; j++) P[j] = pk_adopt<1, PKB>(P[j], P1[j]);
                }
                {   // the match two positions to the right (after the round before), two bytes longer
                    const uint32_t Pa = DPP_ROW_SHL1(P[0]), Pb = DPP_ROW_SHL1(P[1]);
                    const uint32_t P2[4] = {P[2], P[3], Pa, Pb};
#pragma unroll
                    for (int j = 0; j < 4; j++) P[j] = pk_adopt<2, PKB>(P[j], P2[j]);
                }
            }
            __syncthreads();                                                        // every wave has looked up and matched: the window's oldest chunk is free
            // (the window chunk goes to LDS BEHIND this barrier -- until round 4 before it, which cost the window a tile of look-back: a candidate up to NEAR back must
            // still be there while the slowest wave matches --, and the words are stored behind it: the vector memory counter is in order, and a wait for the chunk's
            // load behind this tile's stores would wait for their completion as well)
            if (tid < TILE_G / 16) {
                const uint32_t wo = (loaded_end + tid * 16) & (WIN_BYTES - 1);
                *(uint4 *)(lds + L_WIN + wo) = pf;
                if (wo < WIN_MIRROR) *(uint4 *)(lds + L_WIN + WIN_BYTES + wo) = pf;
            }
            if constexpr (TAB3) {
                // one 64-bit maximum per even position: the word as the look-up saw it with this position's field replaced (position 0 is never stored:
                // its entry could be the empty one)
                if (FULL || (hv[0] && q0 != 0)) t3_max(&table64[hsh[0]], t3_put(w3a, sh3[0], t3_field(w3a, sh3[0]), t3_entry(q0, tag[0])));
                if (FULL || hv[2]) t3_max(&table64[hsh[2]], t3_put(w3b, sh3[2], t3_field(w3b, sh3[2]), t3_entry(q0 + 2, tag[2])));
            } else if (FULL) {                                                      // (all but a block's last tile -- every position valid, no exec masks)
#pragma unroll
                for (int j = 0; j < 4; j++) if (ins_all || !(j & 1)) atomicMax(&table[hsh[j]], ((q0 + j + 1) << TAG_BITS) | tag[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) if (hv[j] && (ins_all || !(j & 1))) atomicMax(&table[hsh[j]], ((q0 + j + 1) << TAG_BITS) | tag[j]);
            }
            if (W3) {
                if (FULL || q0 < t1) {
                    uint32_t ww[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t l = P[j] >> PK_LEN, lc = l < 5u ? 5u : (l > 36u ? 36u : l);   // (v_med3_u32: lengths below MIN_MATCH are strays of the adoption, nobody's match)
                        ww[j] = (lc - 5u) | ((P[j] >> (PK_OFF - 5)) & (0xFFFFFu << 5));
                    }
                    W12 *o = (W12 *)(pb8 + 3 * (size_t)q0);                              // (q0 is a multiple of 4: the 12 bytes are dword-aligned)
                    __builtin_nontemporal_store(__builtin_amdgcn_perm(ww[1], ww[0], 0x04020100u), &o->x);   // bytes 0 1 2 of word 0, byte 0 of word 1
                    __builtin_nontemporal_store(__builtin_amdgcn_perm(ww[2], ww[1], 0x05040201u), &o->y);   // bytes 1 2 of word 1, bytes 0 1 of word 2
                    __builtin_nontemporal_store(__builtin_amdgcn_perm(ww[3], ww[2], 0x06050402u), &o->z);   // byte 2 of word 2, bytes 0 1 2 of word 3
                }
            } else if (FULL || q0 < t1) {
                v4u wv;
                wv.x = (P[0] >> PK_LEN) | (P[0] & (0xFFFFFu << PK_OFF)); wv.y = (P[1] >> PK_LEN) | (P[1] & (0xFFFFFu << PK_OFF));
                wv.z = (P[2] >> PK_LEN) | (P[2] & (0xFFFFFu << PK_OFF)); wv.w = (P[3] >> PK_LEN) | (P[3] & (0xFFFFFu << PK_OFF));
                __builtin_nontemporal_store(wv, (v4u *)(pb + q0));       // (streamed: the parse kernel reads the words, this one never; without the hint they push the
                                                                         // segment's recent bytes -- where most far candidates lie -- out of L2: k_lzm + 1.5 %)
            }
#undef QW
            loaded_end += TILE_G;
            if (GLOG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the atomics have reached L2)
            __syncthreads();                                                        // inserts + window chunk in place
            };
            if (tile_full && !(TAB3 && t0 == 0)) tile_body(std::true_type{}); else tile_body(std::false_type{});
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_lzms -- the match half for SHORT segments (pna_dev.h: at most SMALL_SEG = 4 096 bytes).  k_lzm's tile is 4 096 positions whose look-ups all come before
// its inserts, so a segment of one tile finds nothing, and its workgroup of 16 waves with 160 KiB of LDS per CU spends its time being launched and
// clearing the table (10^6 entries of 4 KiB: 13.7 ms for no match at all).  Here ONE WAVE owns the segment: the whole of it in LDS (no circular window, no far
// candidates: every offset is inside), a table of SMALL_SLOTS 32-bit entries, and the walk in sub-tiles of 256 positions -- four per lane, as in k_lzm --
// whose look-ups see the inserts of the sub-tiles before them: 12.4 KiB of LDS, twelve segments per CU.  Same hash, entries, match step, adoption rounds
// and words as k_lzm (the parse kernel does not know the difference); oracle/zstd_model.c: mtile / small_seg.  4 KiB text entries: ratio 1.77 -> 2.07.
// Two tiers (pna_dev.h), two launches: segments of at most SMALL_SEG bytes, and those of SMALL_SEG + 1 .. MID_SEG bytes; a launch takes the segments of
// (seg_min, seg_max] and has LDS for the table and a window of seg_max bytes (dynamic: 12.4 KiB per wave for the first tier, 16.5 .. 24.5 for the second).
constexpr uint32_t LZMS_PAD = 64;
__host__ __device__ constexpr uint32_t lzms_lds(uint32_t seg_max) { return SMALL_SLOTS * 4 + LZMS_PAD + ((seg_max + 255u) & ~255u) + 64; }
template <int STRONG, bool W3>
__global__ __launch_bounds__(64)
void k_lzms(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, uint32_t flags, uint32_t *__restrict__ pbuf, uint32_t blk0, uint32_t seg_min, uint32_t seg_max) {
    constexpr uint32_t RW = 256, PAD = LZMS_PAD, SLOTS = SMALL_SLOTS, PKB = pkb_of(STRONG), MIN_C = STRONG == 2 ? 16u : 8u;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *table = (uint32_t *)lds;
    uint8_t *winb = lds + SLOTS * 4;                            // 64 bytes in front of the segment (read, never used), the segment, zeros behind it
    const uint32_t lane = threadIdx.x;
    const SegDesc sd = segs[blockIdx.x];
    const uint32_t seg_len = sd.len;
    if (seg_len <= seg_min || seg_len > seg_max) return;
    const uint8_t *seg = src + sd.src_off;
    const uint32_t blk_log = sd.blk_log;
    uint32_t *pb = pbuf + ((size_t)(sd.blk_base - blk0) << blk_log);
    uint8_t *pb8 = (uint8_t *)pbuf + 3 * ((size_t)(sd.blk_base - blk0) << blk_log);
    const bool adopt = (flags & F_ADOPT) != 0, ins_all = !(flags & F_INS2);
    if (!lds_base_is_zero(lds)) __builtin_trap();                                   // (lds_word: the segment's words are addressed from 0)
    for (uint32_t i = lane; i < SLOTS / 4; i += 64) ((uint4 *)table)[i] = make_uint4(0, 0, 0, 0);
    if (lane < PAD / 16) ((uint4 *)winb)[lane] = make_uint4(0, 0, 0, 0);
    for (uint32_t i = lane * 16; i < ((seg_len + 15u) & ~15u) + 64; i += 64 * 16) *(uint4 *)(winb + PAD + i) = i < seg_len ? load_chunk(seg, i, seg_len) : make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t bsz = 1u << blk_log;                                             // (a segment of the second tier may span two blocks of the smallest size: matches end with their block)
    for (uint32_t t0 = 0; t0 < seg_len; t0 += RW) {
        const uint32_t blk_end = ((t0 & ~(bsz - 1)) + bsz < seg_len) ? (t0 & ~(bsz - 1)) + bsz : seg_len;
        const uint32_t t1 = blk_end - t0 < RW ? blk_end : t0 + RW;
        const uint32_t q0 = t0 + 4 * lane;
        // (as in k_lzm: ONE base address per run of words -- the dword of q0 - 8 / c - 8, in front of the segment lie PAD bytes --, words by LDS byte address, packed keys)
        constexpr uint32_t WB8 = SLOTS * 4 + PAD - 8;                              // LDS byte address of the segment's byte -8
        uint32_t D[9], Dm1, Dm2 = 0, Dm3 = 0, Dm4 = 0;
        {
            lds_cu32 *pq = lds_word(WB8 + q0);
#pragma unroll
            for (int k = 0; k < 9; k++) D[k] = pq[k + 2];
            Dm1 = pq[1];
            if (STRONG) Dm2 = pq[0];
            if (STRONG == 2) { lds_cu32 *pe = lds_word(WB8 - 8 + q0); Dm4 = pe[0]; Dm3 = pe[1]; }      // (PAD = 64 bytes lie in front of the segment)
        }
#define QW(k, j) ((j) ? __builtin_amdgcn_alignbyte(D[(k) + 1], D[k], (j)) : D[k])          /* 4 bytes at position j, + 4 k */
        uint32_t hsh[4], tag[4], off[4], P[4];
        bool hv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t q = q0 + j;
            hv[j] = (q < t1) && (q + 8 <= seg_len);
            const uint32_t h32 = QW(0, j) * 0x9E3779B1u + (QW(1, j) & 0xFFFFu) * 0x85EBCA6Bu;
            hsh[j] = __umulhi(h32, SLOTS);
            tag[j] = (h32 >> 6) & TAG_MASK;
            const uint32_t e = table[hsh[j]], ent = hv[j] ? e : 0u;
            const uint32_t c1 = ent >> TAG_BITS;
            off[j] = ((c1 > MIN_C) & ((ent & TAG_MASK) == tag[j])) ? q + 1 - c1 : 0u;
        }
        const bool edge = blk_end - t0 < RW + CAP1;                                 // (uniform) only a block's last sub-tiles can run into its end
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t o = off[j], q = q0 + j;
            uint32_t Pj = 0;
            if (o != 0) {
                const uint32_t c8 = q - 8 - o;                                      // c - 8 (c >= 8)
                const uint32_t shc = c8 << 3;                                       // (v_alignbit takes the shift modulo 32: that of c)
                lds_cu32 *pc = lds_word(WB8 + 8 + (c8 & ~3u));
                const uint32_t dm = pc[1], d0 = pc[2], d1 = pc[3], d2 = pc[4], d3 = pc[5], d4 = pc[6];
                const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, shc), w1 = __builtin_amdgcn_alignbit(d2, d1, shc);
                const uint32_t w2 = __builtin_amdgcn_alignbit(d3, d2, shc), w3 = __builtin_amdgcn_alignbit(d4, d3, shc);
                const uint32_t bc = __builtin_amdgcn_alignbit(d0, dm, shc);
                uint32_t bc2 = 0;
                if (STRONG) bc2 = __builtin_amdgcn_alignbit(dm, pc[0], shc);
                const uint32_t x0 = QW(0, j) ^ w0, x1 = QW(1, j) ^ w1, x2 = QW(2, j) ^ w2, x3 = QW(3, j) ^ w3;
                uint32_t l = first_diff16(x0, x1, x2, x3);
                if (l == 16) {
                    const uint32_t f1 = pc[7], f2 = pc[8], f3 = pc[9], f4 = pc[10];
                    const uint32_t v0 = __builtin_amdgcn_alignbit(f1, d4, shc), v1 = __builtin_amdgcn_alignbit(f2, f1, shc);
                    const uint32_t v2 = __builtin_amdgcn_alignbit(f3, f2, shc), v3 = __builtin_amdgcn_alignbit(f4, f3, shc);
                    const uint32_t y0 = QW(4, j) ^ v0, y1 = QW(5, j) ^ v1, y2 = QW(6, j) ^ v2, y3 = QW(7, j) ^ v3;
                    l = 16 + first_diff16(y0, y1, y2, y3);
                }
                if (edge) { const uint32_t lim = blk_end - q; l = l < lim ? l : lim; }
                const uint32_t bqj = j ? __builtin_amdgcn_alignbyte(D[0], Dm1, j) : Dm1;           // the 4 bytes before q (q - 1 in the top byte)
                const uint32_t xk = bqj ^ bc;
                uint32_t bk3 = (uint32_t)__builtin_clz(xk | 0xFFu) & 24u;                          // back << 3
                if (STRONG == 1 && xk == 0) { const uint32_t bq2 = j ? __builtin_amdgcn_alignbyte(Dm1, Dm2, j) : Dm2; bk3 = 32u + ((uint32_t)__builtin_clz((bq2 ^ bc2) | 0xFFu) & 24u); }
                if (STRONG == 2) {                                                                 // up to 15 back bytes (k_lzm)
                    lds_cu32 *pe = lds_word(WB8 + 8 + ((c8 - 8) & ~3u));                              // the dword of c - 16 (c >= 16)
                    const uint32_t e0 = pe[0], e1 = pe[1], p0 = pc[0];
                    const uint32_t bq2 = j ? __builtin_amdgcn_alignbyte(Dm1, Dm2, j) : Dm2, bq3 = j ? __builtin_amdgcn_alignbyte(Dm2, Dm3, j) : Dm3, bq4 = j ? __builtin_amdgcn_alignbyte(Dm3, Dm4, j) : Dm4;
                    bk3 = back15_field(xk, bq2 ^ bc2, bq3 ^ __builtin_amdgcn_alignbit(p0, e1, shc), bq4 ^ __builtin_amdgcn_alignbit(e1, e0, shc));          // back << 2
                }
                Pj = STRONG ? (l >= MIN_MATCH ? pk_make(l, o, bk3) : 0u) : pk_make(l >= MIN_MATCH ? l : 0u, o, bk3);
            }
            P[j] = Pj;
        }
        if (adopt) {        // backward adoption, the rounds of k_lzm (strong sets: 4, 1, 2; the high / max sets: 8, 4, 1, 2)
            if (STRONG == 2) {
                uint32_t P8[4];
#pragma unroll
                for (int j = 0; j < 4; j++) P8[j] = DPP_ROW_SHL2(P[j]);
#pragma unroll
                for (int j = 0; j < 4; j++) P[j] = pk_adopt<8, PKB>(P[j], P8[j]);
            }
            if (STRONG) {
                uint32_t P4[4];
#pragma unroll
                for (int j = 0; j < 4; j++) P4[j] = DPP_ROW_SHL1(P[j]);
#pragma unroll
                for (int j = 0; j < 4; j++) P[j] = pk_adopt<4, PKB>(P[j], P4[j]);
            }
            {
                const uint32_t Pn = DPP_ROW_SHL1(P[0]);
                const uint32_t P1[4] = {P[1], P[2], P[3], Pn};
#pragma unroll
                for (int j = 0; j < 4; j++) P[j] = pk_adopt<1, PKB>(P[j], P1[j]);
            }
            {
                const uint32_t Pa = DPP_ROW_SHL1(P[0]), Pb = DPP_ROW_SHL1(P[1]);
                const uint32_t P2[4] = {P[2], P[3], Pa, Pb};
#pragma unroll
                for (int j = 0; j < 4; j++) P[j] = pk_adopt<2, PKB>(P[j], P2[j]);
            }
        }
        if (q0 < t1) {
            if (W3) {
                uint32_t ww[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t l = P[j] >> PK_LEN, lc = l < 5u ? 5u : (l > 36u ? 36u : l);
                    ww[j] = (lc - 5u) | ((P[j] >> (PK_OFF - 5)) & (0xFFFFFu << 5));
                }
                W12 *o = (W12 *)(pb8 + 3 * (size_t)q0);
                __builtin_nontemporal_store(__builtin_amdgcn_perm(ww[1], ww[0], 0x04020100u), &o->x);
                __builtin_nontemporal_store(__builtin_amdgcn_perm(ww[2], ww[1], 0x05040201u), &o->y);
                __builtin_nontemporal_store(__builtin_amdgcn_perm(ww[3], ww[2], 0x06050402u), &o->z);
            } else {
                const uint32_t lmax = (flags & FLAG_LEN36) ? 36u : 63u;            // (four-byte words behind the one-kernel form's match half keep the clamp of the three-byte ones)
                uint32_t l4[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { const uint32_t l = P[j] >> PK_LEN; l4[j] = l < lmax ? l : lmax; }
                v4u wv;
                wv.x = l4[0] | (P[0] & (0xFFFFFu << PK_OFF)); wv.y = l4[1] | (P[1] & (0xFFFFFu << PK_OFF));
                wv.z = l4[2] | (P[2] & (0xFFFFFu << PK_OFF)); wv.w = l4[3] | (P[3] & (0xFFFFFu << PK_OFF));
                __builtin_nontemporal_store(wv, (v4u *)(pb + q0));
            }
        }
#undef QW
#pragma unroll
        for (int j = 0; j < 4; j++) if (hv[j] && (ins_all || !(j & 1))) atomicMax(&table[hsh[j]], ((q0 + j + 1) << TAG_BITS) | tag[j]);
        __syncthreads();                                                            // (one wave: the next sub-tile's look-ups come behind these inserts)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_lzp -- the parse half of the split form with one LANE per parse region.  What a whole wave does in k_lz with scalar loops on
// ballot masks (an SALU instruction takes an issue slot like a vector one), sixteen lanes do here with vector arithmetic for the
// sixteen regions of a tile at once.  One wave (= one workgroup: no barriers, 8.4 KiB of LDS) per segment; per tile of 4 096 positions:
//   1. the tile's words (k_lzm's output), 4 consecutive positions per lane and 16-byte load -> their lengths, a byte each, to LDS;
//      then one lane per GROUP of 64 positions: start / cap masks of the group from its 64 length bytes, four at a time inside a
//      register (byte-wise compares by carry-free subtraction, the four results gathered into a nibble by one multiplication)
//   2. lanes 0..15: greedy walk over the region's eight half-groups on 32-bit masks (length of a chosen start from LDS; a capped
//      match is extended by the whole wave, the lengths are kept in LDS), merge across the regions = across the lanes (serial
//      form of the scan, DPP row scans for the counts), selection / literal masks, one record per group to LDS
//   3. one lane per group again: the group's sequences (offsets from the words in memory, four requested at a time)
//   4. the literals, 4 consecutive positions per lane (their input bytes were requested from memory before step 3): the lane's
//      literal bytes are packed by v_perm (selector from a 16-entry table) and stored behind the literals of the positions before it.
// Same results as k_lz<MODE = 2> (and so as the fused kernel): tests/test_gpu_parity.py runs all three.
constexpr uint32_t LZP_THREADS = 64;
template <bool CT, int LZD, bool W3>   // W3: the words take three bytes (k_lzm); LZD: how many positions ahead a start looks before it is taken (lazy deferral: 1, 2 or 3; without F_LAZY none)
__global__ __launch_bounds__(LZP_THREADS) __attribute__((amdgpu_waves_per_eu(5, 5)))   // (88 registers: 5 waves per SIMD; asked for 6 / 7 the allocator spills 4 / 9 registers and the kernel is no faster / 2 % slower)
void k_lzp(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, const uint32_t *__restrict__ blk_seg, uint64_t *__restrict__ seqs, uint8_t *__restrict__ lits,
           BlkInfo *__restrict__ blk, uint4 *__restrict__ ctab, uint32_t flags, uint32_t max_len, const uint32_t *__restrict__ pbuf, uint32_t blk0, uint32_t *__restrict__ hist) {
    constexpr uint32_t RW = 256, TG = 4096;
    static_assert(LZ_G_ZSTD == 4 && LZ_G_DEFLATE == 4 && (1u << BLK_LOG_MIN) % TG == 0 && CAP1 == 32, "k_lzp: regions of 4 groups, tiles of 16 regions");
    __shared__ __attribute__((aligned(16))) uint32_t l32[TG / 4];   // the tile's match lengths, one byte per position; from the end of step 2 on: the records (below)
    __shared__ uint4 lmask[TG / 64];                        // per group: start mask, cap mask
    __shared__ uint32_t plut[16];                           // v_perm selectors that pack the bytes named by a nibble
    // per group: [0] literal mask, first literal index, first sequence index; [1] chosen starts, the capped ones among all chosen; [2] literal-run base of
    // its first sequence, xlen base, cut position | length << 8, cut offset.  The records are written when the walk is through with the lengths and live
    // until the tile's literals are out: they take the lengths' place (3 of its 4 KiB) -- 5.4 KiB of LDS per wave instead of 8.4, so that the registers,
    // not LDS, bound the waves per CU (the kernel hides its memory latency by occupancy: 13 instead of 19 waves per CU cost it 12 %, 9 waves 36 %; above 20 nothing more is gained).
    uint4 (*rec)[TG / 64] = (uint4 (*)[TG / 64])l32;
    __shared__ uint16_t xlen[16 * 8];                       // lengths of a region's extended matches, in the order the walk met them (<= 256 / 32)
    // hist != nullptr (zstd, large batches; round 5): the block's sequence codes are counted HERE -- 3 x 64 counters (one copy: a second one takes the wave's LDS over 8 KiB and the CU from 20 waves to 19), added to the segment's
    // counters in memory when the block is through -- instead of by k_stats, which read all 8-byte sequences once more for them (1.5 of its 2.6 ms per 10 000 segments)
    __shared__ uint32_t shist[192];
    __shared__ uint8_t s_llc[64], s_mlc[128];               // codes of the small literal / match lengths (RFC 8878 3.1.1.3.2.1.1)
    __shared__ uint32_t lstage[264];                        // step 4: the literals of four regions (<= 1 024 bytes behind <= 3 carried ones), stored from here as dwords
    const uint32_t lane = threadIdx.x, w = lane & 15;
    const uint8_t *len8 = (const uint8_t *)l32;
    // one wave per block: block blk0 + blockIdx.x of the sub-batch, its segment from the block -> segment map (segs: all segments of the sub-batch)
    const uint32_t gb = blk0 + blockIdx.x;
    const SegDesc sd = segs[blk_seg[gb]];
    const uint32_t seg_len = sd.len;
    if ((flags & FLAG_SMALL_ONLY) && seg_len > MID_SEG) return;       // (uniform) the pass over the short segments behind a one-kernel launch
    const uint8_t *seg = src + sd.src_off;
    const uint32_t blk_log = sd.blk_log, bsz = 1u << blk_log, SC = seq_cap_of(blk_log);
    const uint32_t *pb = pbuf + ((size_t)(sd.blk_base - blk0) << blk_log);
    const uint8_t *pb8 = (const uint8_t *)pbuf + 3 * ((size_t)(sd.blk_base - blk0) << blk_log);
    // the word of position `pos` of the segment in the 4-byte form (length | offset << 6), whichever way it is stored
    auto word_at = [&](uint32_t pos) -> uint32_t {
        if (!W3) return pb[pos];
        const uint32_t x = *(const u32u *)(pb8 + 3 * (size_t)pos) & 0xFFFFFFu, e = x & 31u;
        return (e ? e + 5u : 0u) | ((x >> 5) << 6);
    };
    const uint32_t lazy = flags & F_LAZY;
    const uint32_t wbase = w * RW;
    const uint32_t bs0 = (gb - sd.blk_base) << blk_log, bs1 = seg_len - bs0 < bsz ? seg_len : bs0 + bsz;    // (an empty segment's only block: no tiles)
    const uint32_t tile0 = bs0 / TG, ntile = bs0 < seg_len ? (bs1 + TG - 1) / TG : tile0;
    const bool lv = lane < 16;
    if (hist) {                                             // (uniform)
        const uint32_t i = lane, m0 = lane, m1 = lane + 64;
        s_llc[i] = (uint8_t)(i < 16 ? i : (i < 24 ? 16 + ((i - 16) >> 1) : (i < 32 ? 20 + ((i - 24) >> 2) : (i < 48 ? 22 + ((i - 32) >> 3) : 24))));
        s_mlc[m0] = (uint8_t)(m0 < 32 ? m0 : (m0 < 40 ? 32 + ((m0 - 32) >> 1) : (m0 < 48 ? 36 + ((m0 - 40) >> 2) : 38 + ((m0 - 48) >> 3))));
        s_mlc[m1] = (uint8_t)(m1 < 96 ? 40 + ((m1 - 64) >> 4) : 42);
    }
    if (lv) {                                               // selector of nibble n: the bytes whose bits are set, lowest first
        uint32_t sel = 0, j = 0;
        for (uint32_t bit = 0; bit < 4; bit++) if ((lane >> bit) & 1) { sel |= bit << (8 * j); j++; }
        plut[lane] = sel;
    }
    // literals: bits of the group's literal mask below this lane's four positions (lane i of a region: group i / 16, bits 4 (i % 16) ..)
    const uint64_t lit_below = ((uint64_t)1 << (4 * (lane & 15))) - 1;
    uint32_t next_free = 0, seq_run = 0, lit_run = 0, g_last1 = 1;     // block-level parse state (uniform)
    uint32_t lcarry = 0;                                    // the < 4 literal bytes behind the block's last complete dword of literals (step 4)

    for (uint32_t T = tile0; T < ntile; T++) {
        const uint32_t t0 = T * TG, blk_start = t0 & ~(bsz - 1);
        const uint32_t blk_end = (seg_len - blk_start < bsz) ? seg_len : blk_start + bsz;
        const uint32_t t1 = (blk_end - t0 < TG) ? blk_end : t0 + TG;
        const uint32_t npos = t1 - t0;
        const uint32_t ext_lim = (t1 + LOOKAHEAD < blk_end) ? t1 + LOOKAHEAD : blk_end;
        const uint32_t gblk = sd.blk_base + (t0 >> blk_log);
        const uint32_t ng = (npos + 63) >> 6;                                           // groups with positions in them
        // ---- 1. lengths to LDS (positions behind the block's end count as "no match"; the words of a whole tile lie inside the segment's
        // share of pbuf, whole blocks, so the loads need no bounds of their own)
        {
            const uint4 *pt = (const uint4 *)(pb + t0);
            const W12 *pt3 = (const W12 *)(pb8 + 3 * (size_t)t0);
            for (uint32_t wq = 0; wq < 4 && wq * 1024 < npos; wq++) {                   // four regions at a time: their loads go out together
                uint4 v[4]; W12 v3[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) { if (W3) v3[i] = pt3[(wq * 4 + i) * 64 + lane]; else v[i] = pt[(wq * 4 + i) * 64 + lane]; }
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const uint32_t p = (wq * 4 + i) * RW + lane * 4;
                    uint32_t d;
                    if (W3) {
                        // the four 5-bit fields (bytes 0, 3, 6, 9 of the lane's 12) as bytes; length = field + 5 where the field is not 0
                        const uint32_t E = (v3[i].x & 31u) | (((v3[i].x >> 24) & 31u) << 8) | (((v3[i].y >> 16) & 31u) << 16) | (((v3[i].z >> 8) & 31u) << 24);
                        const uint32_t nz = ((E + 0x1F1F1F1Fu) >> 5) & 0x01010101u;
                        d = E + nz * 5u;
                    } else d = (v[i].x & 63u) | ((v[i].y & 63u) << 8) | ((v[i].z & 63u) << 16) | ((v[i].w & 63u) << 24);
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
                    if (LZD >= 2) e &= x + 0x01010101u - __builtin_amdgcn_alignbyte(d[i + 1], d[i], 2);   // ... nor to the position after the next (longer by two or more)
                    if (LZD >= 3) e &= x + 0x02020202u - __builtin_amdgcn_alignbyte(d[i + 1], d[i], 3);   // ... nor to the one after that (longer by three or more)
                }
                const uint32_t en = ((((e >> 7) & 0x01010101u) * 0x00204081u) >> 21) & 15u;
                const uint32_t cn = ((((d[i] >> 5) & 0x01010101u) * 0x00204081u) >> 21) & 15u;
                em2[i >> 3] |= en << (4 * (i & 7)); cm2[i >> 3] |= cn << (4 * (i & 7));
            }
            lmask[lane] = make_uint4(em2[0], em2[1], cm2[0], cm2[1]);
        }
        __builtin_amdgcn_wave_barrier();
        if (t0 == blk_start) { next_free = blk_start; seq_run = 0; lit_run = 0; g_last1 = 1; lcarry = 0;
                               if (hist) { for (uint32_t i = lane; i < 192; i += 64) shist[i] = 0; __builtin_amdgcn_wave_barrier(); } }
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
                        const uint32_t pwv = cap ? word_at(qs) : 0u;                 // (its offset)
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
                if (x < k * RW + RW && ek >= x + CUT_MIN) x = ek;
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
                    const uint32_t pw2 = word_at(t0 + wbase + 64 * g2 + s2);
                    uint64_t sc2 = sel[0] & cm[0]; uint32_t pc2 = pc[0];
#pragma unroll
                    for (int r = 1; r < 4; r++) if (g2 == (uint32_t)r) { sc2 = sel[r] & cm[r]; pc2 = pc[r]; }
                    const uint32_t l2 = ((sc2 >> s2) & 1) ? xlen[w * 8 + ((pc2 + (uint32_t)__popcll(sc2 & mlow(s2))) & 7)] : (pw2 & 63u);
                    const uint32_t end2 = 64 * g2 + s2 + l2, rmn = end2 - Ew;
                    if (rmn >= CUT_MIN) {
#pragma unroll
                        for (int r = 0; r < 4; r++) if (grp == (uint32_t)r) fsel[r] |= (uint64_t)1 << b;
                        cut_r = grp; cut_b = b; cut_len = rmn; cut_off = pw2 >> 6;
                    } else {
                        // The remainder is dropped.  Round 5 (the model's `fixup`): ONE match from inside it -- the first start in [E, end of the straddling match) the walk's own
                        // rule would take, if it is not a capped one, ends inside the region and the tile, and ends on a position the region's walk stood on (uncovered, or a
                        // chosen start): from there on the region's parse is the one that entered at E.  Only where the straddling match ends inside the region.
                        uint32_t fx_s = RW, fx_x = 0;                                   // the fix-up match's start and end (region-relative); RW = none
                        if (end2 < RW) {
                            uint64_t eg = 0, eg1 = 0, cg2 = 0, cg3 = 0;                 // start and cap masks of group grp and of the one behind it
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const uint64_t e64 = (uint64_t)em_lo[r] | ((uint64_t)em_hi[r] << 32);
                                if (grp == (uint32_t)r) { eg = e64; cg2 = cm[r]; }
                                if (grp + 1 == (uint32_t)r) { eg1 = e64; cg3 = cm[r]; }
                            }
                            const uint32_t want = (1u << rmn) - 1u;                     // (rmn < CUT_MIN: the window [E, end2) is at most five positions)
                            const uint32_t w8 = (uint32_t)((eg >> b) | (b ? eg1 << (64 - b) : 0)) & want, c8 = (uint32_t)((cg2 >> b) | (b ? cg3 << (64 - b) : 0));
                            if (w8) {
                                const uint32_t ds = (uint32_t)__builtin_ctz(w8), sp = Ew + ds;
                                const uint32_t L = len8[wbase + sp], x = sp + L;
                                if (!((c8 >> ds) & 1) && x < RW && x < in0) {
                                    const uint32_t gx = x >> 6, bx = x & 63;
                                    uint64_t cvx = cov[0], slx = sel[0];
#pragma unroll
                                    for (int r = 1; r < 4; r++) if (gx == (uint32_t)r) { cvx = cov[r]; slx = sel[r]; }
                                    if (!((cvx >> bx) & 1) || ((slx >> bx) & 1)) { fx_s = sp; fx_x = x; }
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const uint32_t a0 = Ew > 64u * r ? (Ew - 64u * r < 64 ? Ew - 64u * r : 64u) : 0u;
                            const uint32_t zend = fx_s < RW ? fx_x : end2;              // coverage is cleared up to the straddling match's end, or up to the fix-up match's (>= that end)
                            const uint32_t z0 = zend > 64u * r ? (zend - 64u * r < 64 ? zend - 64u * r : 64u) : 0u;
                            cv[r] &= ~(mlow(z0) & ~mlow(a0));
                            if (fx_s < RW) {
                                const uint32_t s0 = fx_s > 64u * r ? (fx_s - 64u * r < 64 ? fx_s - 64u * r : 64u) : 0u;
                                cv[r] |= mlow(z0) & ~mlow(s0);                          // the fix-up match's positions
                                fsel[r] &= ~mlow(z0);                                   // the region's starts before its end go with the straddling match
                                if ((fx_s >> 6) == (uint32_t)r) fsel[r] |= (uint64_t)1 << (fx_s & 63);
                            }
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
                    if (lane == 0) ctab[((size_t)gblk << (blk_log - 11)) + (t0 - blk_start) / TILE + h] = make_uint4(seq_run + (ex_h & 0xFFFF), lit_run + (ex_h >> 16), g_first, 0u);
                }
            }
            g_last1 = ga > g_last1 ? ga : g_last1;
            seq_run += tot & 0xFFFF; lit_run += tot >> 16;
            next_free = t0 + rdlane(tile_exit, 0);
            // one record per group for the lanes that write its sequences and literals
            asm volatile("" ::: "memory");                 // (the records overwrite the lengths: every read of those stays in front)
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
        // the input bytes of the lane's four positions in every region, for the literals: requested now, used behind the sequences
        uint32_t lw[16];
        if (t0 + TG <= seg_len) {                           // (uniform) all but a segment's last tile: no end to look out for
#pragma unroll
            for (uint32_t wr = 0; wr < 16; wr++) lw[wr] = *(const uint32_t *)(seg + t0 + wr * RW + 4 * lane);      // (segments start at multiples of 16)
        } else {
#pragma unroll
            for (uint32_t wr = 0; wr < 16; wr++) {
                const uint32_t q = t0 + wr * RW + 4 * lane;
                lw[wr] = 0;
                if (q + 4 <= seg_len) lw[wr] = *(const uint32_t *)(seg + q);
                else { for (uint32_t i = 0; i < 3; i++) if (q + i < seg_len) lw[wr] |= (uint32_t)seg[q + i] << (8 * i); }
            }
        }
        // ---- 3. the sequences, one lane per group
        {
            uint64_t *bseq = seqs + (size_t)gblk * SC;
            const uint4 ra = rec[0][lane], rb = rec[1][lane], rc = rec[2][lane];
            const uint64_t lm = (uint64_t)ra.x | ((uint64_t)ra.y << 32), sc = (uint64_t)rb.z | ((uint64_t)rb.w << 32);
            uint64_t rem = lane < ng ? (uint64_t)rb.x | ((uint64_t)rb.y << 32) : 0;
            uint32_t idx = ra.w, prev = 0;
            const uint32_t cut_b2 = rc.z & 0xFFu, cut_l2 = rc.z >> 8;
            bool first = true;
            const uint32_t pg0 = t0 + lane * 64;
            while (rem) {
                // four starts at a time: their words (the offsets) are requested together
                uint32_t sq[4], pw[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    sq[u] = rem ? ctz64(rem) : 64u;
                    pw[u] = rem ? word_at(pg0 + sq[u]) : 0u;
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
                        if (idx < SC) bseq[idx] = seq_pack(ll, ml, of);
                        if (hist) {
                            const uint32_t mb = ml - 3u, of3 = of + 3u;
                            uint32_t *hh = shist;
                            atomicAdd(&hh[ll < 64 ? (uint32_t)s_llc[ll] : 50u - (uint32_t)__builtin_clz(ll)], 1u);                 // hb(ll) + 19
                            atomicAdd(&hh[64 + 31u - (uint32_t)__builtin_clz(of3)], 1u);                                           // hb(offset + 3)
                            atomicAdd(&hh[128 + (mb < 128 ? (uint32_t)s_mlc[mb] : 67u - (uint32_t)__builtin_clz(mb))], 1u);       // hb(ml - 3) + 36
                        }
                        idx++; prev = lq; first = false;
                    }
                }
            }
        }
        if (hist && t1 == blk_end) {                            // (uniform) the block's counters go to its segment's
            __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
            uint32_t *hs = hist + (size_t)blk_seg[gb] * 448u + 256u;
            for (uint32_t i = lane; i < 192; i += 64) { const uint32_t v = shist[i]; if (v) atomicAdd(&hs[i], v); }
        }
        // ---- 4. literals, region by region, 4 consecutive positions per lane.  Round 5: the lanes' one to four bytes go to an LDS stage (byte writes) and leave
        // four regions at a time as aligned DWORDS, the < 4 bytes behind the last complete dword carried to the next chunk / tile (until then: four predicated byte
        // stores per region to memory, 64 store instructions per tile -- a fifth of the kernel's time)
        {
            uint8_t *blit = lits + ((size_t)gblk << blk_log);
            const uint32_t sh = 4 * (lane & 15);
            uint8_t *stage8 = (uint8_t *)lstage;
#pragma unroll
            for (uint32_t wq = 0; wq < 4; wq++) {
                if (wq * 1024 >= npos) continue;                                    // (uniform)
                const uint32_t L0 = rec[0][wq * 16].z, L1 = wq < 3 ? rec[0][wq * 16 + 16].z : lit_run;   // literal indices (block-relative) the chunk's regions cover
                const uint32_t A0 = L0 & ~3u;
                if (lane == 0) lstage[0] = lcarry;                                  // the bytes [A0, L0)
                __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const uint32_t wr = wq * 4 + i;
                    if (wr * RW >= npos) continue;                                  // (uniform)
                    const uint4 m = rec[0][wr * 4 + (lane >> 4)];
                    const uint32_t wd = lw[wr];
                    const uint64_t lm = (uint64_t)m.x | ((uint64_t)m.y << 32);
                    const uint32_t nib = (uint32_t)(lm >> sh) & 15u;
                    const uint32_t pk = __builtin_amdgcn_perm(wd, wd, plut[nib]), cnt = (uint32_t)__popc(nib);
                    uint8_t *o = stage8 + (m.z + (uint32_t)__popcll(lm & lit_below) - A0);
                    if (cnt > 0) o[0] = (uint8_t)pk;
                    if (cnt > 1) o[1] = (uint8_t)(pk >> 8);
                    if (cnt > 2) o[2] = (uint8_t)(pk >> 16);
                    if (cnt > 3) o[3] = (uint8_t)(pk >> 24);
                }
                __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
                const uint32_t nfull = (L1 - A0) >> 2;                              // <= 256
                uint32_t *g32 = (uint32_t *)(blit + A0);
                for (uint32_t j = lane; j < nfull; j += 64) g32[j] = lstage[j];
                lcarry = uni(lstage[nfull]);
                __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
            }
            if (t1 == blk_end) {                                                    // the block's last tile: the bytes behind its last complete dword
                const uint32_t rem = lit_run & 3u;
                if (lane < rem) blit[(lit_run & ~3u) + lane] = (uint8_t)(lcarry >> (8 * lane));
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");                     // (the next tile's lengths overwrite the records)
    }
}

template <bool CT, int STRONG, uint32_t GLOG, uint32_t WLOG, bool FARP = !CT, bool TAB3 = false>
static void launch_split_g(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
                           uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, hipEvent_t ev_match, uint32_t *gtab, const LzParseGrid *pg) {
    constexpr uint32_t LT = GLOG ? LzGeo<WLOG>::L_TABLE : LzGeo<WLOG, TAB3>::L_TOTAL;
    static const hipError_t attr_set = hipFuncSetAttribute((const void *)k_lzm<CT, STRONG, GLOG, WLOG, FARP, TAB3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LzGeo<WLOG, TAB3>::L_TOTAL);   // once per process, thread-safe
    (void)attr_set;
    constexpr bool W3 = GLOG == 0;
    if (!(flags & FLAG_ALL_SMALL)) hipLaunchKernelGGL((k_lzm<CT, STRONG, GLOG, WLOG, FARP, TAB3>), dim3(nseg), dim3(LZ_THREADS), LT, st, src, segs, flags, max_off, pbuf, blk0, gtab);
    if (flags & FLAG_TIER1) hipLaunchKernelGGL((k_lzms<STRONG, W3>), dim3(nseg), dim3(64), lzms_lds(SMALL_SEG), st, src, segs, flags, pbuf, blk0, 0u, SMALL_SEG);     // the short segments, which k_lzm skipped
    if (flags & FLAG_TIER2) { const uint32_t mx = 8192u + 4096u * ((flags >> FLAG_T2_SHIFT) & 3u);
                              hipLaunchKernelGGL((k_lzms<STRONG, W3>), dim3(nseg), dim3(64), lzms_lds(mx), st, src, segs, flags, pbuf, blk0, SMALL_SEG, mx); }
    if (ev_match) (void)hipEventRecord(ev_match, st);
    if (!pg || pg->nb == 0) return;                            // (no grid: the caller wants the match kernel alone; a run of empty entries has segments and no blocks)
    if (flags & FLAG_LAZY3) hipLaunchKernelGGL((k_lzp<CT, 3, W3>), dim3(pg->nb), dim3(LZP_THREADS), 0, st, src, pg->segs_all, pg->blk_seg, seqs, lits, blk, ctab, flags, max_len, pbuf, blk0, pg->hist);
    else if (flags & FLAG_LAZY2) hipLaunchKernelGGL((k_lzp<CT, 2, W3>), dim3(pg->nb), dim3(LZP_THREADS), 0, st, src, pg->segs_all, pg->blk_seg, seqs, lits, blk, ctab, flags, max_len, pbuf, blk0, pg->hist);
    else hipLaunchKernelGGL((k_lzp<CT, 1, W3>), dim3(pg->nb), dim3(LZP_THREADS), 0, st, src, pg->segs_all, pg->blk_seg, seqs, lits, blk, ctab, flags, max_len, pbuf, blk0, pg->hist);
}
// match kernel + parse kernel over `nseg` segments; pbuf holds one word per position of the launch's blocks, blk0 = the first of them; ctab != nullptr:
// a deflate launch (chunk table, look-back inside the LDS window); ev_match, if given, is recorded between the two kernels
// gtab != nullptr (zstd, strong set): the match kernel's hash tables in global memory, nseg << GTAB_LOG words
void launch_lz_split(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
                     uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, hipEvent_t ev_match, uint32_t *gtab, const LzParseGrid *pg) {
    const bool strong = (flags & F_STRONG) && (flags & F_ADOPT), strong2 = strong && (flags & FLAG_STRONG2);     // (FLAG_STRONG2: only with the global table or the packed 16 KiB geometry)
    if (ctab) { if (strong) launch_split_g<true, true, 0, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg);
                else launch_split_g<true, false, 0, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg); }
    else if (strong && gtab) { if (strong2) launch_split_g<false, 2, GTAB_LOG, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg);
                               else launch_split_g<false, true, GTAB_LOG, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg); }
    else if ((flags & FLAG_TAB3) && (flags & FLAG_W16)) {
           if (strong2) launch_split_g<false, 2, 0, 14, true, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg);
           else if (strong) launch_split_g<false, true, 0, 14, true, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg);
           else launch_split_g<false, false, 0, 14, true, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg); }
    else if ((flags & FLAG_TAB3) && (flags & FLAG_W32)) {
           if (strong) launch_split_g<false, true, 0, 15, true, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg);
           else launch_split_g<false, false, 0, 15, true, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg); }
    else if (flags & FLAG_W16) {
           if (strong) launch_split_g<false, true, 0, 14>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg);
           else launch_split_g<false, false, 0, 14>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg); }
    else if (flags & FLAG_W32) {
           if (strong) launch_split_g<false, true, 0, 15>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg);
           else launch_split_g<false, false, 0, 15>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg); }
    else { if (strong) launch_split_g<false, true, 0, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg);
           else if (max_off <= NEAR_OFF) launch_split_g<false, false, 0, 16, false>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg);
           else launch_split_g<false, false, 0, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, nullptr, pg); }
}
// The short segments of a launch that took the one-kernel form (which skips them: FLAG_HAS_SMALL): k_lzms + the parse kernel over their blocks only.  w3: words of
// three bytes (the caller's choice: every launch but the zstd levels with the table in global memory takes them, as in the split form).
void launch_lz_small(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
                     uint32_t flags, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, const LzParseGrid *pg, bool w3) {
    const bool strong = (flags & F_STRONG) && (flags & F_ADOPT), strong2 = strong && (flags & FLAG_STRONG2);
#define LZMS(ST_, W3_) do { \
        if (flags & FLAG_TIER1) hipLaunchKernelGGL((k_lzms<ST_, W3_>), dim3(nseg), dim3(64), lzms_lds(SMALL_SEG), st, src, segs, flags, pbuf, blk0, 0u, SMALL_SEG); \
        if (flags & FLAG_TIER2) { const uint32_t mx = 8192u + 4096u * ((flags >> FLAG_T2_SHIFT) & 3u); \
                                  hipLaunchKernelGGL((k_lzms<ST_, W3_>), dim3(nseg), dim3(64), lzms_lds(mx), st, src, segs, flags, pbuf, blk0, SMALL_SEG, mx); } } while (0)
    if (strong2) { if (w3) LZMS(2, true); else LZMS(2, false); }
    else if (strong) { if (w3) LZMS(true, true); else LZMS(true, false); }
    else { if (w3) LZMS(false, true); else LZMS(false, false); }
#undef LZMS
    if (!pg || pg->nb == 0) return;
    const uint32_t pf = flags | FLAG_SMALL_ONLY;
#define LZP_SMALL(CT_, LZD_) do { if (w3) hipLaunchKernelGGL((k_lzp<CT_, LZD_, true>), dim3(pg->nb), dim3(LZP_THREADS), 0, st, src, pg->segs_all, pg->blk_seg, seqs, lits, blk, ctab, pf, max_len, pbuf, blk0, pg->hist); \
                                  else hipLaunchKernelGGL((k_lzp<CT_, LZD_, false>), dim3(pg->nb), dim3(LZP_THREADS), 0, st, src, pg->segs_all, pg->blk_seg, seqs, lits, blk, ctab, pf, max_len, pbuf, blk0, pg->hist); } while (0)
    if (ctab) { if (flags & FLAG_LAZY3) LZP_SMALL(true, 3); else if (flags & FLAG_LAZY2) LZP_SMALL(true, 2); else LZP_SMALL(true, 1); }
    else { if (flags & FLAG_LAZY3) LZP_SMALL(false, 3); else if (flags & FLAG_LAZY2) LZP_SMALL(false, 2); else LZP_SMALL(false, 1); }
#undef LZP_SMALL
}
uint32_t lz_gtab_log() { return GTAB_LOG; }
void lzp_read_stamps(unsigned long long *out) {
    for (int k = 0; k < 8; k++) out[k] = 0;
}

} // namespace pna
