// k_lz.hip -- LZ77 match finding + greedy parse for gfx950 (CDNA4), one 1024-thread workgroup per segment: the ONE-kernel form of the LZ stage
// (short runs, PNA_F_LZ_FUSED, fallback without workspace) and, as MODE 1 / 2, its two halves as kernels.  The default for large batches is the
// split form in k_lz_split.hip (k_lzm + k_lzp); shared pieces live in lz_common.h.
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
//   the front (>= CUT_MIN = 6 bytes must remain),
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
#include "lz_common.h"

#define LZ_INS_COND (ins_all || !(lane & 1))

namespace pna {

void launch_lz_split(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
                     uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, hipEvent_t ev_match, uint32_t *gtab, const LzParseGrid *pg);   // k_lz_split.hip
void lzp_read_stamps(unsigned long long *out);
void launch_lz_small(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
                     uint32_t flags, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, const LzParseGrid *pg, bool w3);   // k_lz_split.hip

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
template <bool STAMP, int G, bool CT, int STRONG, int MODE, uint32_t WLOG, bool TAB3 = false>   // TAB3: the packed table (lz_common.h); STRONG: 0 / 1 / 2 as in k_lzm (2: a fourth adoption round over eight positions, 15 back bytes)
__global__ __launch_bounds__(LZ_THREADS, MODE == 2 ? 8 : 4)   // (second figure: waves per SIMD the compiler must leave room for)
void k_lz(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, uint64_t *__restrict__ seqs,
          uint8_t *__restrict__ lits, BlkInfo *__restrict__ blk, uint4 *__restrict__ ctab, uint32_t flags, uint32_t max_off, uint32_t max_len,
          uint32_t *__restrict__ pbuf, uint32_t blk0) {
    constexpr uint32_t RW = 64u * G;                       // positions one wave owns = the parse region
    constexpr uint32_t TILE_G = RW * LZ_WAVES;             // positions per synchronous step
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    constexpr bool FAR = !CT;                               // deflate offsets (<= 32 KiB) never leave the LDS window
    using GEO = LzGeo<WLOG, TAB3>;                          // (lz_common.h) these names hide the 64 KiB geometry's constants of pna_dev.h
    constexpr uint32_t WIN_BYTES = GEO::WIN, HASH_ENTRIES = GEO::ENTRIES, L_TABLE = GEO::L_TABLE, L_WEND = GEO::L_WEND, L_WPUB = GEO::L_WPUB, NW3 = GEO::WORDS3;
    static_assert(!TAB3 || (!CT && G == 4), "the packed table: zstd launches");
    constexpr uint32_t NEAR = G == 2 ? MAX_OFF_G2 : GEO::NEAR;
    static_assert(TILE_G % TILE == 0 && WIN_BYTES >= 2 * TILE_G + LOOKAHEAD + 16 + NEAR && (!CT || NEAR >= 32768), "window: look-back + this tile + look-ahead + the chunk in flight");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *win32   = (uint32_t *)(lds + L_WIN);
    uint32_t *table   = (uint32_t *)(lds + L_TABLE);
    uint64_t *table64 = (uint64_t *)(lds + L_TABLE);      // TAB3
    uint32_t *wend    = (uint32_t *)(lds + (MODE == 2 ? 0u : L_WEND));
    WPub     *wpub    = (WPub *)(lds + (MODE == 2 ? 4u * LZ_WAVES : L_WPUB));

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = uni(tid >> 6);                  // tell the compiler it is wave-uniform: keeps the parse walks on the scalar unit
    const SegDesc sd = segs[blockIdx.x];
    const uint8_t *seg = src + sd.src_off;
    const uint32_t seg_len = sd.len;
    if (MODE != 2 && !lds_base_is_zero(lds)) __builtin_trap();       // (lds_word: the window's words are addressed from 0)
    const uint32_t blk_log = sd.blk_log, bsz = 1u << blk_log, SC = seq_cap_of(blk_log);   // block size of the batch = stride of the per-block arrays
    if (MODE != 2 && (flags & FLAG_HAS_SMALL) && seg_len <= MID_SEG) return;         // (uniform) a short segment: k_lzms's (pna_dev.h; MODE 2 parses its words like any)
    uint32_t *pb = MODE ? pbuf + ((size_t)(sd.blk_base - blk0) << blk_log) : nullptr;   // the segment's words (split form)
    const uint32_t lazy = flags & F_LAZY;
    const bool adopt = (flags & F_ADOPT) != 0, ins_all = !(flags & F_INS2);
    constexpr bool strong = STRONG != 0, strong2 = STRONG == 2;   // (wave-uniform) level sets: pna_host.cpp level_flags()
    constexpr uint32_t KL = strong2 ? 8u : 6u, KB = strong2 ? 4u : 3u, KLOW = (1u << KL) - 1u, MIN_C = strong2 ? 16u : 8u;   // the adoption key: len << KL | back << KB | lanes moved; a usable candidate lies at MIN_C or beyond
    const bool force_serial = (flags & FLAG_FORCE_SERIAL) != 0;
    const uint64_t lane_lt = ((uint64_t)1 << lane) - 1;   // lanes below this one
    const uint32_t wbase = wave * RW;                     // tile-relative first position of this wave

    if (MODE != 2) for (uint32_t i = tid; i < GEO::TABLE_BYTES / 16; i += LZ_THREADS) ((uint4 *)table)[i] = make_uint4(0, 0, 0, 0);   // 16 bytes per store
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    if (STAMP && lane == 0) st_prev = __builtin_amdgcn_s_memtime();
#define LZ_STAMP(k) do { if (STAMP && lane == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_prev; st_prev = t_; } } while (0)

    // initial window fill [0, TILE_G + LOOKAHEAD + 16); afterwards one tile-sized chunk per tile: requested at the top of
    // tile t, stored into LDS before tile t's B3, first read after B4 (tile t+1's lookups).  The slots it overwrites hold
    // positions below t0 + 2 TILE_G + LOOKAHEAD + 16 - 64 Ki <= t0 - max_off, which no match of tile t can reference.
    // A unit that starts inside the segment (latency mode): the window holds the 64 KiB that end where the first tile's chunk ends, the table is
    // pre-warmed with the positions before the unit (lz_common.h) -- the state the segment-long walk would have here.
    uint32_t loaded_end = sd.u0 + TILE_G + LOOKAHEAD + 16;
    if (MODE != 2) {
        __syncthreads();                                                            // (the table is zero before the pre-warm's inserts)
        if (sd.u0) { if constexpr (TAB3) lz_prewarm3<NW3>(table64, seg, seg_len, sd.u0, tid); else lz_prewarm<0, HASH_ENTRIES>(table, seg, seg_len, sd.u0, ins_all, tid); }
        for (uint32_t i = (loaded_end > WIN_BYTES ? loaded_end - WIN_BYTES : 0u) + tid * 16; i < loaded_end; i += LZ_THREADS * 16) {
            const uint4 v = load_chunk(seg, i, seg_len);
            const uint32_t wo = i & (WIN_BYTES - 1);
            *(uint4 *)(lds + L_WIN + wo) = v;
            if (wo < WIN_MIRROR) *(uint4 *)(lds + L_WIN + WIN_BYTES + wo) = v;
        }
        __syncthreads();
    }
    uint4 pf = make_uint4(0, 0, 0, 0);

    for (uint32_t blk_start = sd.u0; blk_start < sd.u1; blk_start += bsz) {
        const uint32_t b = blk_start >> blk_log;
        const uint32_t blk_end = (seg_len - blk_start < bsz) ? seg_len : blk_start + bsz;
        const uint32_t gblk = sd.blk_base + b;
        uint64_t *bseq = seqs + (size_t)gblk * SC;
        uint8_t  *blit = lits + ((size_t)gblk << blk_log);
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

            // ---- lookup
            uint32_t q[G], lo[G], hi[G], hsh[G], tag[G], ent[G], bq[G], bq2[G], bq3[G], bq4[G];
            uint32_t sh3[G]; uint64_t w3[G];                                         // TAB3: the field's bit position and the word as the look-up saw it
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
                    hi[r] = hsh[r] = tag[r] = ent[r] = bq[r] = bq2[r] = bq3[r] = bq4[r] = 0;
                    continue;
                }
                {
                    // ONE base address per run of window words (round 4, as in k_lzm): the dword that holds q - 8; what lies behind the window's end is its mirror
                    // (48 bytes: the base + 44 at most), and the words are read by their LDS byte address (lz_common.h lds_word)
                    lds_cu32 *p = lds_word(L_WIN + ((q[r] - 8) & (WIN_BYTES - 4)));
                    const uint32_t sh = q[r] << 3;                                  // (v_alignbit takes the shift modulo 32)
                    const uint32_t d0 = p[2], d1 = p[3], d2 = p[4], dm = p[1];
                    lo[r] = __builtin_amdgcn_alignbit(d1, d0, sh);
                    hi[r] = __builtin_amdgcn_alignbit(d2, d1, sh);
                    bq[r] = __builtin_amdgcn_alignbit(d0, dm, sh);                  // the 4 bytes before q (q - 1 in the top byte)
                    bq2[r] = bq3[r] = bq4[r] = 0;
                    if (strong) bq2[r] = __builtin_amdgcn_alignbit(dm, p[0], sh);   // (uniform) and the 4 before those
                    if (strong2) { lds_cu32 *pe = lds_word(L_WIN + ((q[r] - 16) & (WIN_BYTES - 4))); const uint32_t e0 = pe[0], e1 = pe[1];      // (uniform) ... and the 8 before those
                                   bq3[r] = __builtin_amdgcn_alignbit(p[0], e1, sh); bq4[r] = __builtin_amdgcn_alignbit(e1, e0, sh); }
                }
                const uint32_t h32 = lo[r] * 0x9E3779B1u + (hi[r] & 0xFFFFu) * 0x85EBCA6Bu;
                sh3[r] = 0; w3[r] = 0;
                if constexpr (TAB3) {
                    t3_slot<NW3>(h32, hsh[r], sh3[r]);
                    tag[r] = t3_tag(h32);
                    w3[r] = table64[hsh[r]];
                    ent[r] = hv[r] ? t3_field(w3[r], sh3[r]) : 0u;
                } else {
                hsh[r] = __umulhi(h32, HASH_ENTRIES);                               // floor(h32 * entries / 2^32): any table size
                tag[r] = (h32 >> 6) & TAG_MASK;                                     // a filter only: any function of the hash will do
                ent[r] = hv[r] ? table[hsh[r]] : 0u;
                }
            }
            // ---- candidates: offset (0 = none: empty slot, foreign tag -- a candidate whose tag differs hashed differently, so its first
            // 6 bytes differ --, position below 8, beyond max_off).  Far candidates (beyond the LDS window) get the 4 bytes before and the
            // 16 bytes at the candidate requested from the segment now; the other lanes read the segment's first bytes (one line, no
            // exec masking), which nobody looks at.  A usable candidate lies at position >= 8, so the loads never reach below the segment.
            uint32_t off[G];
            v4u fa[G]; uint32_t fb[G], fc[G]; v2u fe[G];
#pragma unroll
            for (int r = 0; r < G; r++) {
                if constexpr (MODE == 2) { off[r] = pv[r] >> 6; continue; }
                fa[r] = 0; fb[r] = fc[r] = 0; fe[r] = 0;
                // (TAB3: an entry = (position / 2) << 2 | tag, usable from position 8 on: entry >= 16)
                const uint32_t c1 = TAB3 ? t3_pos(ent[r]) + 1 : ent[r] >> TAG_BITS, o = q[r] + 1 - c1;
                off[r] = TAB3 ? ((ent[r] >= 2u * MIN_C && (ent[r] & 3u) == tag[r] && o <= max_off) ? o : 0u)
                              : ((c1 > MIN_C && (ent[r] & TAG_MASK) == tag[r] && o <= max_off) ? o : 0u);
            }
            // FLAG_FAR1 (the default / light zstd sets, round 5): the split form's match kernel verifies at most 63 candidates beyond ITS window (offset >= GEO::NEARM) per
            // wave of 256 positions -- one compacted round -- and drops the rest; its numbering is j-major over four consecutive positions per lane (all positions
            // = 0 mod 4 of the wave in ascending order, then = 1 mod 4, ...).  The same candidates are dropped here: position 64 r + lane has class j = lane & 3, and its
            // number is the count of the classes below, of its class in the groups before r, and of its class below the lane.
            if constexpr (TAB3 && MODE != 2) {
                if (flags & FLAG_FAR1) {                                            // (uniform)
                    uint64_t fm[G]; uint32_t cls[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int r = 0; r < G; r++) {
                        fm[r] = __ballot(off[r] >= GEO::NEARM);
#pragma unroll
                        for (int j = 0; j < 4; j++) cls[j] += (uint32_t)__builtin_popcountll(fm[r] & (0x1111111111111111ull << j));
                    }
                    const uint32_t j = lane & 3u;
                    uint32_t run = j == 0 ? 0u : (j == 1 ? cls[0] : (j == 2 ? cls[0] + cls[1] : cls[0] + cls[1] + cls[2]));      // far candidates of the classes below mine
                    const uint64_t mine = (0x1111111111111111ull << j) & lane_lt;                                                 // my class, below my lane
#pragma unroll
                    for (int r = 0; r < G; r++) {
                        const uint32_t idx = run + (uint32_t)__builtin_popcountll(fm[r] & mine);
                        if (off[r] >= GEO::NEARM && idx >= 63u) off[r] = 0;
                        uint32_t c0 = (uint32_t)__builtin_popcountll(fm[r] & 0x1111111111111111ull), c1 = (uint32_t)__builtin_popcountll(fm[r] & 0x2222222222222222ull);
                        uint32_t c2 = (uint32_t)__builtin_popcountll(fm[r] & 0x4444444444444444ull), c3 = (uint32_t)__builtin_popcountll(fm[r] & 0x8888888888888888ull);
                        run += j == 0 ? c0 : (j == 1 ? c1 : (j == 2 ? c2 : c3));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < G; r++) {
                if constexpr (MODE == 2) continue;
                const uint32_t c1 = q[r] + 1 - off[r];                              // (only read where off[r] != 0)
                if (FAR && seg_len > NEAR && max_off > NEAR) {                      // (uniform) shorter segments / near-only levels have no far candidates
                    const uint32_t fo = off[r] > NEAR ? c1 - 5 : 0u;                // byte offset of c - 4 in the segment
                    fa[r] = ld16u(seg + fo);
                    fb[r] = *(const u32u *)(seg + fo + 16);
                    if (strong) fc[r] = *(const u32u *)(seg + (off[r] > NEAR ? fo - 4 : 0u));   // (uniform) bytes c - 8 .. c - 5
                    if (strong2) __builtin_memcpy(&fe[r], seg + (off[r] > NEAR ? fo - 12 : 0u), 8);   // (uniform) bytes c - 16 .. c - 9
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
                    lds_cu32 *pq = lds_word(L_WIN + ((q[r] - 8) & (WIN_BYTES - 4)));   // pq[2 ..] = the words at q, pc[2 ..] those at c (one base each: see the look-up)
                    lds_cu32 *pc = lds_word(L_WIN + ((c - 8) & (WIN_BYTES - 4)));
                    const uint32_t shc = c << 3, shq = q[r] << 3;
                    uint32_t w0, w1, w2, w3, bc, bc2 = 0, bc3 = 0, bc4 = 0;                // 16 bytes at c, the 4 (strong: 8, 16) bytes before c
                    if (isfar) { bc = fa[r].x; w0 = fa[r].y; w1 = fa[r].z; w2 = fa[r].w; w3 = fb[r]; bc2 = fc[r]; bc3 = fe[r].y; bc4 = fe[r].x; }
                    else {
                        const uint32_t d0 = pc[2], d1 = pc[3], d2 = pc[4], d3 = pc[5], d4 = pc[6], dm = pc[1];
                        w0 = __builtin_amdgcn_alignbit(d1, d0, shc); w1 = __builtin_amdgcn_alignbit(d2, d1, shc);
                        w2 = __builtin_amdgcn_alignbit(d3, d2, shc); w3 = __builtin_amdgcn_alignbit(d4, d3, shc);
                        bc = __builtin_amdgcn_alignbit(d0, dm, shc);
                        if (strong) bc2 = __builtin_amdgcn_alignbit(dm, pc[0], shc);
                        if (strong2) { lds_cu32 *pe = lds_word(L_WIN + ((c - 16) & (WIN_BYTES - 4))); const uint32_t e0 = pe[0], e1 = pe[1];
                                       bc3 = __builtin_amdgcn_alignbit(pc[0], e1, shc); bc4 = __builtin_amdgcn_alignbit(e1, e0, shc); }
                    }
                    const uint32_t e2 = pq[4], e3 = pq[5], e4 = pq[6];
                    const uint32_t x0 = lo[r] ^ w0, x1 = hi[r] ^ w1;
                    const uint32_t x2 = __builtin_amdgcn_alignbit(e3, e2, shq) ^ w2;
                    const uint32_t x3 = __builtin_amdgcn_alignbit(e4, e3, shq) ^ w3;
                    l = first_diff16(x0, x1, x2, x3);
#pragma unroll
                    for (uint32_t k16 = 16; k16 < CAP1; k16 += 16) {
                        if (l == k16) {
                            // the next 16 bytes, only for the lanes where everything before matched (same alignment as above): most capped
                            // matches end here, which keeps them off the wave-cooperative extension in the parse loop.  Far candidates
                            // fetch theirs from the segment now (rare: a few lanes per tile, and the line is usually still in L1 / L2)
                            static_assert(CAP1 <= 32, "the bases' reach: the second 16 bytes end at base + 44, inside the window's 48-byte mirror");
                            const uint32_t g0 = pq[2 + k16 / 4], g1 = pq[3 + k16 / 4], g2 = pq[4 + k16 / 4], g3 = pq[5 + k16 / 4], g4 = pq[6 + k16 / 4];
                            uint32_t v0, v1, v2, v3;
                            if (isfar) { const U4u t = *(const U4u *)(seg + c + k16); v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w; }
                            else {
                                const uint32_t f0 = pc[2 + k16 / 4], f1 = pc[3 + k16 / 4], f2 = pc[4 + k16 / 4], f3 = pc[5 + k16 / 4], f4 = pc[6 + k16 / 4];
                                v0 = __builtin_amdgcn_alignbit(f1, f0, shc); v1 = __builtin_amdgcn_alignbit(f2, f1, shc);
                                v2 = __builtin_amdgcn_alignbit(f3, f2, shc); v3 = __builtin_amdgcn_alignbit(f4, f3, shc);
                            }
                            const uint32_t y0 = __builtin_amdgcn_alignbit(g1, g0, shq) ^ v0, y1 = __builtin_amdgcn_alignbit(g2, g1, shq) ^ v1;
                            const uint32_t y2 = __builtin_amdgcn_alignbit(g3, g2, shq) ^ v2, y3 = __builtin_amdgcn_alignbit(g4, g3, shq) ^ v3;
                            l = k16 + first_diff16(y0, y1, y2, y3);
                        }
                    }
                    if (edge) { const uint32_t lim = blk_end - q[r]; l = l < lim ? l : lim; }
                    if (l < MIN_MATCH) l = 0;
                    // bytes before q and c that agree as well, nearest first: the low byte forced to differ caps the count at BACK_CAP = 3
                    // (strong set: when all four agree, the four before them are counted the same way: at most 7)
                    const uint32_t xk = bq[r] ^ bc;
                    bk = (uint32_t)__builtin_clz(xk | 0xFFu) >> 3;
                    if (strong && xk == 0) bk = 4 + ((uint32_t)__builtin_clz((bq2[r] ^ bc2) | 0xFFu) >> 3);
                    if (strong2 && bk == 7 && bq2[r] == bc2) {                      // all eight agree: the eight before them, the sixteenth never counted
                        const uint32_t x3 = bq3[r] ^ bc3;
                        bk = x3 ? 8 + ((uint32_t)__builtin_clz(x3) >> 3) : 12 + ((uint32_t)__builtin_clz((bq4[r] ^ bc4) | 0xFFu) >> 3);
                    }
                }
                // ---- backward adoption.  K = len << 6 | back << 3 | lanes the match was moved by.  A lane without a match may carry a stray
                // back count and adopt "lengths" of 1..3 from such neighbours: they stay below MIN_MATCH and nobody reads them as a match.
                uint32_t K = (l << KL) | (bk << KB);
                if (STRONG && !l) K = 0;                                            // (three rounds could lift a stray back count to a "length" of 7 >= MIN_MATCH; with two it stays below)
                if (adopt) {
                // taking over the match S lanes to the right: + S bytes, back - S, moved + S; allowed iff its back count is at least S -- a test of the count's upper bits
                constexpr uint32_t BM = ((strong2 ? 15u : 7u) << KB);
#define K_STEP(S) ((uint32_t)(S) * ((1u << KL) - (1u << KB) + 1u))
#define K_OK(S) (BM & ~(((uint32_t)(S) - 1u) << KB))
                if (strong2) {  // (uniform) the high / max sets' round over eight lanes comes first (rounds 8, 4, 1, 2)
                    const uint32_t K8r = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 8u) & 63u) << 2), (int)K);      // (every lane takes part: a lane switched off would hand out 0)
                    const uint32_t K8 = lane < 56u ? K8r : 0u, T = K8 + K_STEP(8);
                    K = ((K8 & K_OK(8)) != 0 && T > (K | KLOW)) ? T : K;
                }
                if (strong) {   // (uniform) the strong sets' round comes first (rounds 4, 1, 2): four lanes to the right, four bytes longer
                    const uint32_t K4 = dpp_next_lane(dpp_next_lane(dpp_next_lane(dpp_next_lane(K)))), T = K4 + K_STEP(4);
                    K = ((K4 & K_OK(4)) != 0 && T > (K | KLOW)) ? T : K;
                }
                {   // the right neighbour's match, one byte longer (lane 63 sees 0)
                    const uint32_t K1 = dpp_next_lane(K), T = K1 + K_STEP(1);
                    K = ((K1 & K_OK(1)) != 0 && T > (K | KLOW)) ? T : K;
                }
                {   // the match two lanes to the right (after the round before), two bytes longer
                    const uint32_t K2 = dpp_next_lane(dpp_next_lane(K)), T = K2 + K_STEP(2);
                    K = ((K2 & K_OK(2)) != 0 && T > (K | KLOW)) ? T : K;
                }
#undef K_STEP
#undef K_OK
                l = K >> KL;
                if (flags & FLAG_LEN36) l = l < 36u ? l : 36u;                      // (uniform) what the split form's 3-byte words keep
                off[r] = (uint32_t)__shfl((int)o, (int)(lane + (K & ((1u << KB) - 1u))));   // the offset travels with the match
                }
                }
                len[r] = l; flen[r] = l;
                if constexpr (MODE == 1) { effm[r] = 0; return; }
                const uint32_t nl = dpp_next_lane(l);                               // len of the next position (lane 63: 0, so lane 63 never defers)
                // lazy deferral: position q waits iff q + 1 is still in the tile and holds a longer match.  nl > l with l < MIN_MATCH is harmless
                // (the position is no start anyway); q + 1 >= t1 only happens in a block's last, partial tile (nl is 0 there: lanes >= t1 hold no match)
                uint64_t longer = lazy ? __ballot(nl > l) : 0;
                if (lazy && (flags & FLAG_LAZY2)) longer |= __ballot(dpp_next_lane(nl) > l + 1);     // (uniform) two-step deferral: q + 2 holds a match longer by two or more
                if (lazy && (flags & FLAG_LAZY3)) longer |= __ballot(dpp_next_lane(dpp_next_lane(nl)) > l + 2);   // (uniform) three-step: q + 3, longer by three or more
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
                while (rem) {
                    const uint32_t s = ctz64(rem);
                    uint32_t e = rdlane(endp, s);
                    if ((capm >> s) & 1) {
                        const uint32_t qs = t0 + wbase + 64 * r + s, os = rdlane(off[r], s), L0 = e - s;
                        const uint32_t xl = ext_lim - qs < max_len ? ext_lim - qs : max_len;
                        const uint32_t L = MODE == 2 ? lz_extend_mem(seg, seg_len, qs, qs - os, L0, xl, lane)
                                         : (FAR && os > NEAR) ? lz_extend<true, WIN_BYTES>(win32, seg, qs, qs - os, L0, xl, lane)
                                                              : lz_extend<false, WIN_BYTES>(win32, seg, qs, qs - os, L0, xl, lane);
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
                for (int r = 0; r < G; r++) {
                    if constexpr (TAB3) { if (hv[r] && !(lane & 1) && q[r] != 0) t3_max(&table64[hsh[r]], t3_put(w3[r], sh3[r], t3_field(w3[r], sh3[r]), t3_entry(q[r], tag[r]))); }
                    else if (hv[r] && LZ_INS_COND) atomicMax(&table[hsh[r]], ((q[r] + 1) << TAG_BITS) | tag[r]);
                }
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
            for (int r = 0; r < G; r++) {                                            // even positions only
                if constexpr (TAB3) { if (hv[r] && !(lane & 1) && q[r] != 0) t3_max(&table64[hsh[r]], t3_put(w3[r], sh3[r], t3_field(w3[r], sh3[r]), t3_entry(q[r], tag[r]))); }
                else if (hv[r] && LZ_INS_COND) atomicMax(&table[hsh[r]], ((q[r] + 1) << TAG_BITS) | tag[r]);
            }
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
                const bool near = lv && el_l > prev && (el_l < prev + CUT_MIN || prev >= lane * RW + RW);
                if (__ballot(near) == 0 && !force_serial) {
                    const uint32_t m = wave ? rdlane(incl, wave - 1) : 0u, ma = rdlane(incl, LZ_WAVES - 1);
                    E = m > c_in ? m : c_in;
                    tile_exit = ma > c_in ? ma : c_in;
                } else {
                    uint32_t x = c_in; E = c_in;
                    for (uint32_t j = 0; j < LZ_WAVES; j++) {
                        if (j == wave) E = x;
                        const uint32_t ej = rdlane(el_l, j);
                        if (x < j * RW + RW && ej >= x + CUT_MIN) x = ej;
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
                        if (rmn >= CUT_MIN) {
#pragma unroll
                            for (int r = 0; r < G; r++) if (grp == (uint32_t)r) { fsel[r] |= (uint64_t)1 << b; if (lane == b) { flen[r] = rmn; off[r] = o2; } }
                        } else {
                            // its last bytes (fewer than CUT_MIN) stay literals -- or, round 5 (the model's `fixup`, as in k_lzp): ONE match from inside them, the first start in
                            // [E, end2) the walk's rule would take, if it is not a capped one, ends inside the region and the tile and on a position the walk stood on
                            uint32_t fx_s = RW, fx_x = 0;
                            if (end2 < RW) {
                                uint64_t eg = effm[0], eg1 = G > 1 ? effm[1] : 0;
#pragma unroll
                                for (int r = 1; r < G; r++) if (grp == (uint32_t)r) { eg = effm[r]; eg1 = r + 1 < G ? effm[r + 1 < G ? r + 1 : r] : 0; }
                                const uint32_t w8 = (uint32_t)((eg >> b) | (b ? eg1 << (64 - b) : 0)) & ((1u << rmn) - 1u);
                                if (w8) {
                                    const uint32_t sp = Ew + (uint32_t)__builtin_ctz(w8), gs = sp >> 6, ls = sp & 63;
                                    uint32_t L = rdlane(len[0], ls);
#pragma unroll
                                    for (int r = 1; r < G; r++) if (gs == (uint32_t)r) L = rdlane(len[r], ls);
                                    const uint32_t x = sp + L;
                                    if (L < CAP1 && x < RW && x < in0) {
                                        const uint32_t gx = x >> 6, bx = x & 63;
                                        uint64_t cvx = cov[0], slx = sel[0];
#pragma unroll
                                        for (int r = 1; r < G; r++) if (gx == (uint32_t)r) { cvx = cov[r]; slx = sel[r]; }
                                        if (!((cvx >> bx) & 1) || ((slx >> bx) & 1)) { fx_s = sp; fx_x = x; }
                                    }
                                }
                            }
#pragma unroll
                            for (int r = 0; r < G; r++) {
                                const uint32_t a0 = Ew > 64u * r ? (Ew - 64u * r < 64 ? Ew - 64u * r : 64u) : 0u;
                                const uint32_t zend = fx_s < RW ? fx_x : end2;
                                const uint32_t z0 = zend > 64u * r ? (zend - 64u * r < 64 ? zend - 64u * r : 64u) : 0u;
                                cv[r] &= ~(mlow(z0) & ~mlow(a0));
                                if (fx_s < RW) {
                                    const uint32_t s0 = fx_s > 64u * r ? (fx_s - 64u * r < 64 ? fx_s - 64u * r : 64u) : 0u;
                                    cv[r] |= mlow(z0) & ~mlow(s0);
                                    fsel[r] &= ~mlow(z0);
                                    if ((fx_s >> 6) == (uint32_t)r) fsel[r] |= (uint64_t)1 << (fx_s & 63);
                                }
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
                        if (tid == 0) ctab[((size_t)gblk << (blk_log - 11)) + (t0 - blk_start) / TILE + h] = make_uint4(seq_run + (ex_h & 0xFFFF), lit_run + (ex_h >> 16), g_first, 0u);
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
                        if (idx < SC) bseq[idx] = seq_pack(ll, flen[r], off[r]);
                    }
                    if ((litm[r] >> lane) & 1) { const uint32_t li = lit_base + lb; if (li < bsz) blit[li] = (uint8_t)lo[r]; }
                }
            }
            LZ_STAMP(6);
        } // tiles
        if (MODE != 1 && tid == 0) { blk[gblk].nseq = seq_run; blk[gblk].nlit = lit_run; }
    } // blocks
    if (STAMP && lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_lz_stamps[k], st_acc[k]);
}

template <int G, bool CT, int STRONG, uint32_t WLOG, bool TAB3 = false>
static void launch_lz_g(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
                        uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, hipEvent_t ev_match, uint32_t *gtab, const LzParseGrid *pg) {
    static constexpr uint32_t LT = LzGeo<WLOG, TAB3>::L_TOTAL;
    static const hipError_t attr_set = [] {                    // once per process, thread-safe (contexts may be created on several threads)
        (void)hipFuncSetAttribute((const void *)k_lz<false, G, CT, STRONG, 0, WLOG, TAB3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LT);
        (void)hipFuncSetAttribute((const void *)k_lz<false, G, CT, STRONG, 1, WLOG, TAB3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LT);
        return hipFuncSetAttribute((const void *)k_lz<true, G, CT, STRONG, 0, WLOG, TAB3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LT);
    }();
    (void)attr_set;
    if (pbuf) {
        if (flags & FLAG_SPLIT_WAVEPARSE) {
            hipLaunchKernelGGL((k_lz<false, G, CT, STRONG, 1, WLOG, TAB3>), dim3(nseg), dim3(LZ_THREADS), LT, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len, pbuf, blk0);
            if (flags & FLAG_HAS_SMALL) launch_lz_small(src, segs, nseg, seqs, lits, blk, ctab, flags, max_len, st, pbuf, blk0, nullptr, false);   // (match kernel only; this form's words take four bytes)
            if (ev_match) (void)hipEventRecord(ev_match, st);
            hipLaunchKernelGGL((k_lz<false, G, CT, STRONG, 2, WLOG, TAB3>), dim3(nseg), dim3(LZ_THREADS), 4 * LZ_WAVES + 8 * LZ_WAVES, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len, pbuf, blk0);
            return;
        }
        launch_lz_split(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg);   // k_lzm + k_lzp (k_lz_split.hip)
    }
    else if (flags & FLAG_STAMP) hipLaunchKernelGGL((k_lz<true, G, CT, STRONG, 0, WLOG, TAB3>), dim3(nseg), dim3(LZ_THREADS), LT, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len, pbuf, blk0);
    else hipLaunchKernelGGL((k_lz<false, G, CT, STRONG, 0, WLOG, TAB3>), dim3(nseg), dim3(LZ_THREADS), LT, st, src, segs, seqs, lits, blk, ctab, flags, max_off, max_len, pbuf, blk0);
}
// zstd launches (no chunk table) run LZ_G_ZSTD positions per lane and tile, deflate launches LZ_G_DEFLATE (k_dblock walks the 2 KiB chunks of the table)
// pbuf != nullptr: the split form (two kernels; pbuf holds one word per position of the launch's blocks, blk0 = the first of them;
// ev_match, if given, is recorded between the two)
void launch_lz(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk, uint4 *ctab,
               uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, uint32_t *pbuf, uint32_t blk0, hipEvent_t ev_match, uint32_t *gtab, const LzParseGrid *pg) {
    const bool strong = (flags & F_STRONG) && (flags & F_ADOPT), strong2 = strong && (flags & FLAG_STRONG2);
    if (ctab) { if (strong) launch_lz_g<LZ_G_DEFLATE, true, true, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg);
                else launch_lz_g<LZ_G_DEFLATE, true, false, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg); }
    else if ((flags & FLAG_TAB3) && (flags & FLAG_W16)) {
           if (strong2) launch_lz_g<LZ_G_ZSTD, false, 2, 14, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg);
           else if (strong) launch_lz_g<LZ_G_ZSTD, false, true, 14, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg);
           else launch_lz_g<LZ_G_ZSTD, false, false, 14, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg); }
    else if ((flags & FLAG_TAB3) && (flags & FLAG_W32)) {
           if (strong) launch_lz_g<LZ_G_ZSTD, false, true, 15, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg);
           else launch_lz_g<LZ_G_ZSTD, false, false, 15, true>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg); }
    else if (flags & FLAG_W16) {
           if (strong) launch_lz_g<LZ_G_ZSTD, false, true, 14>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg);
           else launch_lz_g<LZ_G_ZSTD, false, false, 14>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg); }
    else if (flags & FLAG_W32) {
           if (strong) launch_lz_g<LZ_G_ZSTD, false, true, 15>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg);
           else launch_lz_g<LZ_G_ZSTD, false, false, 15>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg); }
    else { if (strong) launch_lz_g<LZ_G_ZSTD, false, true, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg);
           else launch_lz_g<LZ_G_ZSTD, false, false, 16>(src, segs, nseg, seqs, lits, blk, ctab, flags, max_off, max_len, st, pbuf, blk0, ev_match, gtab, pg); }
}

// diagnostic: read and clear the phase stamps (cycles summed over workgroups); a -DLZP_PROF build hands out k_lzp's instead
void lz_read_stamps(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lz_stamps), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz_stamps), z, sizeof(z));
}

} // namespace pna
