// k_lz.hip -- LZ77 match finding + greedy parse for gfx950 (CDNA4), one 1024-thread workgroup per segment.
//
// Replaces the match finder inside the third-party encoder the reference drives at
// lib/src/compress.rs:32-41 (CompressionWriter::write -> ZstdEncoder::write).  Integer/byte work, no MFMA.
//
// LDS (one workgroup per CU, ~146 KiB of the 160 KiB):
//   win   [65536 B]  circular copy of the segment's most recent 64 KiB (look-back + 1 KiB look-ahead)
//   table [16384 x u32] hash table: position+1 of the latest occurrence (0 = empty); inserts are ds_max_u32
//   len/off/fixlen [2048 x u16 each]  per-position results of the current tile (fallback path only reads them)
//   per-wave records
// Per tile of 2048 positions (2 per lane):
//   window chunk (register-prefetched during the previous tile) -> B1 -> lookup -> B2 -> insert + match +
//   per-wave SPECULATIVE parse (as if the parse entered the wave at its first position) -> B3 ->
//   every wave resolves its TRUE entry in parallel (carry chained through the speculative exits of the earlier,
//   not fully covered waves); it walks from there until it lands on a position its speculative parse also stood on -> finalise ->
//   B4 -> prefix sums over the waves' counts -> emission of sequences and literals straight to HBM.
//   The parallel resolution is exact when every not-fully-covered wave re-synchronises (checked); otherwise the
//   tile falls back to a serial resolution by wave 0 (rare; forced with flag 0x200 for testing).
#include <hip/hip_runtime.h>
#include "pna_dev.h"

namespace pna {

constexpr uint32_t WMASK32 = WIN_BYTES / 4 - 1;

// LDS layout (byte offsets into the dynamic shared array)
constexpr uint32_t L_WIN    = 0;
constexpr uint32_t L_TABLE  = L_WIN + WIN_BYTES;
constexpr uint32_t L_LEN    = L_TABLE + (4u << HASH_LOG);
constexpr uint32_t L_OFF    = L_LEN + 2 * TILE;
constexpr uint32_t L_FIXLEN = L_OFF + 2 * TILE;
constexpr uint32_t L_WMETA  = L_FIXLEN + 2 * TILE;          // 16 x 64 B
constexpr uint32_t L_WRES   = L_WMETA + 64 * LZ_WAVES;      // 16 x 32 B
constexpr uint32_t L_WPUB   = L_WRES + 32 * LZ_WAVES;       // 16 x 16 B
constexpr uint32_t L_STATE  = L_WPUB + 16 * LZ_WAVES;       // 16 B
constexpr uint32_t L_TOTAL  = L_STATE + 16;

struct WMeta { uint64_t sel[2]; uint64_t vis[2]; uint64_t eff[2]; uint32_t exit0; uint32_t last_end0; uint32_t pad[2]; };
struct WRes  { uint64_t fix[2]; uint32_t carry; uint32_t sync; uint32_t lit_start; uint32_t pad; };
struct WPub  { uint32_t nsel, sumlen, last_end, bad; };
static_assert(sizeof(WMeta) == 64 && sizeof(WRes) == 32 && sizeof(WPub) == 16, "LDS record sizes");

constexpr uint32_t FLAG_STAMP = 0x100u, FLAG_FORCE_FALLBACK = 0x200u;

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, uint32_t l) {
    return (uint64_t)rdlane((uint32_t)v, l) | ((uint64_t)rdlane((uint32_t)(v >> 32), l) << 32);
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t ctz64(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
__device__ __forceinline__ uint32_t clz64(uint64_t v) { return (uint32_t)__builtin_clzll(v); }
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t t = (uint32_t)__shfl_xor((int)v, d); v = v > t ? v : t; }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    return v;
}

// 8 / 4 bytes at an arbitrary segment position from the circular window
__device__ __forceinline__ void fetch8(const uint32_t *win32, uint32_t pos, uint32_t &lo, uint32_t &hi) {
    uint32_t w = (pos & (WIN_BYTES - 1)) >> 2, sh = (pos & 3) * 8;
    uint32_t d0 = win32[w], d1 = win32[(w + 1) & WMASK32], d2 = win32[(w + 2) & WMASK32];
    lo = __builtin_amdgcn_alignbit(d1, d0, sh);
    hi = __builtin_amdgcn_alignbit(d2, d1, sh);
}
__device__ __forceinline__ uint32_t fetch4(const uint32_t *win32, uint32_t pos) {
    uint32_t w = (pos & (WIN_BYTES - 1)) >> 2, sh = (pos & 3) * 8;
    return __builtin_amdgcn_alignbit(win32[(w + 1) & WMASK32], win32[w], sh);
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

// diagnostic build only (STAMP = true): wave 0 / lane 0 accumulates s_memtime deltas per phase
__device__ unsigned long long g_lz_stamps[8];

template <bool STAMP>
__global__ __launch_bounds__(LZ_THREADS)
void k_lz(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, uint64_t *__restrict__ seqs,
          uint8_t *__restrict__ lits, BlkInfo *__restrict__ blk, uint32_t flags) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *win32   = (uint32_t *)(lds + L_WIN);
    uint32_t *table   = (uint32_t *)(lds + L_TABLE);
    uint16_t *len_arr = (uint16_t *)(lds + L_LEN);
    uint16_t *off_arr = (uint16_t *)(lds + L_OFF);
    uint16_t *fix_arr = (uint16_t *)(lds + L_FIXLEN);
    WMeta    *wmeta   = (WMeta *)(lds + L_WMETA);
    WRes     *wres    = (WRes *)(lds + L_WRES);
    WPub     *wpub    = (WPub *)(lds + L_WPUB);
    uint32_t *state   = (uint32_t *)(lds + L_STATE);      // fallback: [0] next_free (abs), [1] lit_start (abs)

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // initial window fill [0, TILE + LOOKAHEAD + 16); afterwards one TILE-sized chunk per tile, prefetched in registers
    uint32_t loaded_end = TILE + LOOKAHEAD + 16;
    for (uint32_t i = tid * 16; i < loaded_end; i += LZ_THREADS * 16)
        *(uint4 *)(lds + L_WIN + i) = load_chunk(seg, i, seg_len);
    uint4 pf = make_uint4(0, 0, 0, 0);
    bool have_pf = false;

    const uint32_t nblk = (seg_len + BLK_SIZE - 1) / BLK_SIZE;
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t blk_start = b * BLK_SIZE;
        const uint32_t blk_end = (seg_len - blk_start < BLK_SIZE) ? seg_len : blk_start + BLK_SIZE;
        const uint32_t gblk = sd.blk_base + b;
        uint64_t *bseq = seqs + (size_t)gblk * SEQ_CAP;
        uint8_t  *blit = lits + (size_t)gblk * BLK_SIZE;
        uint32_t next_free = blk_start, lit_start = blk_start;     // uniform across the workgroup
        uint32_t seq_run = 0, len_run = 0;

        for (uint32_t t0 = blk_start; t0 < blk_end; t0 += TILE) {
            const uint32_t t1 = (blk_end - t0 < TILE) ? blk_end : t0 + TILE;
            const uint32_t ext_lim = (t1 + LOOKAHEAD < blk_end) ? t1 + LOOKAHEAD : blk_end;

            // ---- window: store the chunk prefetched during the previous tile, then prefetch the next one.
            // Invariant at B1: window holds [t0 + TILE + LOOKAHEAD + 16 - 65536, t0 + TILE + LOOKAHEAD + 16).
            if (have_pf && tid < TILE / 16) *(uint4 *)(lds + L_WIN + ((loaded_end - TILE + tid * 16) & (WIN_BYTES - 1))) = pf;
            __syncthreads();                                                        // B1
            if (tid < TILE / 16) { pf = (loaded_end + tid * 16 < seg_len) ? load_chunk(seg, loaded_end + tid * 16, seg_len) : make_uint4(0, 0, 0, 0); }
            loaded_end += TILE; have_pf = true;
            LZ_STAMP(0);

            // ---- lookup
            uint32_t q[2], lo[2], hi[2], hsh[2], c1[2];
            bool inb[2], hv[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                q[r] = t0 + wbase + 64 * r + lane;
                inb[r] = q[r] < t1;
                hv[r] = inb[r] && (q[r] + 8 <= seg_len);
                fetch8(win32, q[r], lo[r], hi[r]);
                hsh[r] = (lo[r] * 0x9E3779B1u + (hi[r] & 0xFFFFu) * 0x85EBCA6Bu) >> (32 - HASH_LOG);
                c1[r] = hv[r] ? table[hsh[r]] : 0u;
            }
            __syncthreads();                                                        // B2
            LZ_STAMP(1);

            // ---- insert + match
            uint32_t len[2], off[2], flen[2];
            uint64_t effm[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if (hv[r]) atomicMax(&table[hsh[r]], q[r] + 1);
                uint32_t l = 0, o = 0;
                if (c1[r] != 0) {
                    uint32_t c = c1[r] - 1; o = q[r] - c;
                    if (o <= MAX_OFF) {
                        uint32_t lim = blk_end - q[r]; lim = lim < CAP1 ? lim : CAP1;
                        uint32_t clo, chi;
                        fetch8(win32, c, clo, chi);
                        uint64_t x = (uint64_t)(lo[r] ^ clo) | ((uint64_t)(hi[r] ^ chi) << 32);
                        if (x) l = ctz64(x) >> 3;
                        else {
                            l = 8;
                            while (l < lim) {
                                uint32_t alo, ahi;
                                fetch8(win32, q[r] + l, alo, ahi);
                                fetch8(win32, c + l, clo, chi);
                                x = (uint64_t)(alo ^ clo) | ((uint64_t)(ahi ^ chi) << 32);
                                if (x) { l += ctz64(x) >> 3; break; }
                                l += 8;
                            }
                        }
                        l = l < lim ? l : lim;
                        if (l < MIN_MATCH) l = 0;
                    }
                }
                len[r] = l; off[r] = o; flen[r] = l;
                len_arr[wbase + 64 * r + lane] = (uint16_t)l;
                off_arr[wbase + 64 * r + lane] = (uint16_t)o;
                uint32_t nl = (uint32_t)__shfl_down((int)l, 1);
                bool eff = l >= MIN_MATCH && !(lazy && lane != 63 && (q[r] + 1 < t1) && nl > l);
                effm[r] = __ballot(eff);
            }
            LZ_STAMP(2);

            // ---- speculative parse of this wave's 128 positions (entry at its first position)
            uint64_t sel[2] = {0, 0};
            uint32_t entry[2];
            uint32_t cur = 0, last_end0 = 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                uint32_t e = cur > 64u * r ? cur - 64u * r : 0u;
                entry[r] = e;
                while (e < 64) {
                    uint64_t mm = (effm[r] >> e) << e;
                    if (!mm) { e = 64; break; }
                    uint32_t s = ctz64(mm);
                    uint32_t L = rdlane(len[r], s);
                    if (L == CAP1) {
                        uint32_t qs = t0 + wbase + 64 * r + s;
                        L = lz_extend(win32, qs, qs - rdlane(off[r], s), ext_lim - qs, lane);
                    }
                    if (lane == s) flen[r] = L;
                    sel[r] |= (uint64_t)1 << s;
                    e = s + L;
                    last_end0 = 64 * r + e;
                }
                cur = 64 * r + e;
            }
            // positions the speculative parse stands on (not covered by a speculative match)
            uint64_t vis[2];
            {
                uint32_t ce_prev = 0;   // wave-relative end of the last match of the previous group
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    uint64_t m_le = sel[r] & (lane_lt | ((uint64_t)1 << lane));
                    uint32_t sl = m_le ? 63 - clz64(m_le) : 0;
                    uint32_t fl = (uint32_t)__shfl((int)flen[r], (int)sl);
                    // a match START is a position the parse stands on; only its interior is "covered"
                    bool cov = (m_le != 0 && lane > sl && lane < sl + fl) || (64u * r + lane < ce_prev) || (lane < entry[r]);
                    vis[r] = __ballot(!cov);   // positions past the tile end count as stood-on
                    if (sel[r]) { uint32_t s = 63 - clz64(sel[r]); ce_prev = 64 * r + s + rdlane(flen[r], s); }
                }
            }
            const uint32_t flen_spec[2] = {flen[0], flen[1]};
            if (lane == 0) {
                WMeta m;
                m.sel[0] = sel[0]; m.sel[1] = sel[1]; m.vis[0] = vis[0]; m.vis[1] = vis[1];
                m.eff[0] = effm[0]; m.eff[1] = effm[1]; m.exit0 = cur; m.last_end0 = last_end0; m.pad[0] = m.pad[1] = 0;
                wmeta[wave] = m;
            }
            __syncthreads();                                                        // B3
            LZ_STAMP(3);

            // ---- parallel resolution: assumed carry = max(tile carry, speculative exits of all earlier waves)
            const uint32_t c_in = next_free > t0 ? next_free - t0 : 0u;             // tile-relative
            const uint32_t ex_l = lane < LZ_WAVES ? lane * 128 + wmeta[lane & (LZ_WAVES - 1)].exit0 : 0u;
            // carry chain over the speculative exits: a wave the carry already covers passes it through, any other
            // wave is assumed to re-synchronise and therefore to leave at its speculative exit (checked below)
            uint32_t A = c_in, c_chain = c_in;
            for (uint32_t j = 0; j < LZ_WAVES; j++) {
                if (j == wave) A = c_chain;
                if (c_chain < j * 128 + 128) c_chain = rdlane(ex_l, j);
            }
            const uint32_t cw = A > wbase ? A - wbase : 0u;                         // wave-relative entry
            uint64_t fix[2] = {0, 0};
            uint32_t e_sync = 128, last_fix_end = 0;
            bool synced = false;
            {
                uint32_t e = cw;
                while (e < 128) {
                    const uint32_t r1 = e >> 6, bp = e & 63;
                    const uint64_t vs = (r1 ? vis[1] : vis[0]) >> bp, es = (r1 ? effm[1] : effm[0]) >> bp;
                    const uint32_t nv = vs ? ctz64(vs) : 64u, ne = es ? ctz64(es) : 64u;
                    if (nv <= ne) {
                        if (!vs) { e = 64 * (r1 + 1); continue; }
                        e += nv; synced = true; e_sync = e; break;
                    }
                    e += ne;
                    const uint32_t b2 = e & 63;
                    uint32_t L = r1 ? rdlane(len[1], b2) : rdlane(len[0], b2);
                    if (L == CAP1) {
                        const uint32_t o = r1 ? rdlane(off[1], b2) : rdlane(off[0], b2);
                        const uint32_t qs = t0 + wbase + e;
                        L = lz_extend(win32, qs, qs - o, ext_lim - qs, lane);
                    }
                    if (r1) { fix[1] |= (uint64_t)1 << b2; if (lane == b2) flen[1] = L; }
                    else    { fix[0] |= (uint64_t)1 << b2; if (lane == b2) flen[0] = L; }
                    e += L; last_fix_end = e;
                }
            }
            const bool bad_w = !synced && cw < 128;                                 // walked off the wave without re-synchronising

            // ---- finalisation (shared by both paths): selection, coverage, literal mask, in-wave prefix of lengths
            uint64_t fsel[2]; uint32_t fl[2]; bool islit[2]; uint32_t pre[2]; uint32_t tot_len = 0;
            uint32_t carry_w = cw, sync_w = e_sync;
            uint64_t fixm[2] = {fix[0], fix[1]};
            auto finalize = [&](bool from_lds) {
                tot_len = 0;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    uint64_t keep;
                    if (sync_w >= 64u * (r + 1)) keep = 0;
                    else if (sync_w <= 64u * r) keep = ~(uint64_t)0;
                    else keep = (~(uint64_t)0) << (sync_w - 64u * r);
                    fsel[r] = (sel[r] & keep) | fixm[r];
                    if (from_lds) fl[r] = ((fixm[r] >> lane) & 1) ? (uint32_t)fix_arr[wbase + 64 * r + lane] : flen_spec[r];
                    else fl[r] = flen[r];
                }
                uint32_t ce_prev = 0;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    uint64_t m_le = fsel[r] & (lane_lt | ((uint64_t)1 << lane));
                    uint32_t sl = m_le ? 63 - clz64(m_le) : 0;
                    uint32_t fs = (uint32_t)__shfl((int)fl[r], (int)sl);
                    bool cov = (m_le != 0 && lane < sl + fs) || (64u * r + lane < ce_prev) || (64u * r + lane < carry_w);
                    islit[r] = inb[r] && !cov;
                    if (fsel[r]) { uint32_t s = 63 - clz64(fsel[r]); ce_prev = 64 * r + s + rdlane(fl[r], s); }
                    uint32_t v = ((fsel[r] >> lane) & 1) ? fl[r] : 0u, sc = v;
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { uint32_t t = (uint32_t)__shfl_up((int)sc, d); if (lane >= (uint32_t)d) sc += t; }
                    pre[r] = tot_len + sc - v;
                    tot_len += rdlane(sc, 63);
                }
            };
            finalize(false);
            uint32_t nsel0 = (uint32_t)__popcll(fsel[0]), nsel = nsel0 + (uint32_t)__popcll(fsel[1]);
            if (lane == 0) {
                uint32_t le = 0;                                                    // abs end of this wave's last true match
                if (synced) {
                    bool later = e_sync < 64 ? (((sel[0] >> e_sync) != 0) || sel[1] != 0) : ((sel[1] >> (e_sync - 64)) != 0);
                    if (later) le = t0 + wbase + last_end0;
                }
                if (le == 0 && last_fix_end) le = t0 + wbase + last_fix_end;
                WPub p; p.nsel = nsel; p.sumlen = tot_len; p.last_end = le; p.bad = bad_w ? 1u : 0u;
                wpub[wave] = p;
            }
            __syncthreads();                                                        // B4
            LZ_STAMP(4);

            WPub pl = wpub[lane & (LZ_WAVES - 1)];
            const bool lv = lane < LZ_WAVES;
            const bool any_bad = __ballot(lv && pl.bad) != 0;
            uint32_t seq_base, len_base, lit_start_w;
            if (!any_bad && !force_fb) {
                seq_base = seq_run + wave_sum(lv && lane < wave ? pl.nsel : 0u);
                len_base = len_run + wave_sum(lv && lane < wave ? pl.sumlen : 0u);
                seq_run += wave_sum(lv ? pl.nsel : 0u);
                len_run += wave_sum(lv ? pl.sumlen : 0u);
                uint32_t le_before = wave_max(lv && lane < wave ? pl.last_end : 0u);
                lit_start_w = le_before > lit_start ? le_before : lit_start;
                uint32_t le_all = wave_max(lv ? pl.last_end : 0u);
                lit_start = le_all > lit_start ? le_all : lit_start;
                next_free = t0 + c_chain;
            } else {
                // ---- fallback: wave 0 resolves the true entry of every wave serially (scalar code)
                if (wave == 0) {
                    WMeta m = wmeta[lane & (LZ_WAVES - 1)];
                    uint32_t c = c_in;
                    uint32_t ls = lit_start;
                    for (uint32_t w = 0; w < LZ_WAVES; w++) {
                        const uint32_t base = w * 128;
                        uint32_t cw2 = c > base ? c - base : 0u;
                        const uint64_t vis0 = rdlane64(m.vis[0], w), vis1 = rdlane64(m.vis[1], w);
                        const uint64_t eff0 = rdlane64(m.eff[0], w), eff1 = rdlane64(m.eff[1], w);
                        const uint64_t sel0 = rdlane64(m.sel[0], w), sel1 = rdlane64(m.sel[1], w);
                        const uint32_t exit0 = rdlane(m.exit0, w), le0 = rdlane(m.last_end0, w);
                        uint64_t fix0 = 0, fix1 = 0;
                        uint32_t e = cw2, ls_in = ls;
                        bool sy = false;
                        while (e < 128) {
                            const uint32_t bp = e & 63;
                            const uint64_t visr = e < 64 ? vis0 : vis1, effr = e < 64 ? eff0 : eff1;
                            if ((visr >> bp) & 1) { sy = true; break; }
                            if ((effr >> bp) & 1) {
                                const uint32_t pq = base + e;
                                uint32_t L = uni(len_arr[pq]);
                                const uint32_t o = uni(off_arr[pq]);
                                const uint32_t qs = t0 + pq;
                                if (L == CAP1) L = lz_extend(win32, qs, qs - o, ext_lim - qs, lane);
                                if (lane == 0) fix_arr[pq] = (uint16_t)L;
                                if (e < 64) fix0 |= (uint64_t)1 << bp; else fix1 |= (uint64_t)1 << bp;
                                e += L; ls = t0 + base + e;
                            } else e += 1;
                        }
                        if (sy) {
                            bool later = e < 64 ? (((sel0 >> e) != 0) || sel1 != 0) : ((sel1 >> (e - 64)) != 0);
                            if (later) ls = t0 + base + le0;
                            c = base + exit0;
                        } else c = base + e;
                        if (lane == 0) {
                            WRes rr; rr.fix[0] = fix0; rr.fix[1] = fix1; rr.carry = cw2; rr.sync = sy ? e : 128u;
                            rr.lit_start = ls_in; rr.pad = 0;
                            wres[w] = rr;
                        }
                    }
                    if (lane == 0) { state[0] = t0 + c; state[1] = ls; }
                }
                __syncthreads();
                const WRes rr = wres[wave];
                next_free = state[0]; lit_start = state[1];
                carry_w = rr.carry; sync_w = rr.sync; fixm[0] = rr.fix[0]; fixm[1] = rr.fix[1];
                lit_start_w = rr.lit_start;
                finalize(true);
                nsel0 = (uint32_t)__popcll(fsel[0]); nsel = nsel0 + (uint32_t)__popcll(fsel[1]);
                __syncthreads();                                                    // everyone has read wpub / state
                if (lane == 0) { WPub p; p.nsel = nsel; p.sumlen = tot_len; p.last_end = 0; p.bad = 0; wpub[wave] = p; }
                __syncthreads();
                pl = wpub[lane & (LZ_WAVES - 1)];
                seq_base = seq_run + wave_sum(lv && lane < wave ? pl.nsel : 0u);
                len_base = len_run + wave_sum(lv && lane < wave ? pl.sumlen : 0u);
                seq_run += wave_sum(lv ? pl.nsel : 0u);
                len_run += wave_sum(lv ? pl.sumlen : 0u);
                LZ_STAMP(5);
            }

            // ---- emission (shuffles stay outside divergent code: an inactive source lane would read as 0)
            uint32_t end_g0 = 0;                                    // end of the last selected match of group 0
            if (fsel[0]) { uint32_t sp = 63 - clz64(fsel[0]); end_g0 = t0 + wbase + sp + rdlane(fl[0], sp); }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const uint64_t pm = fsel[r] & lane_lt;
                const uint32_t sp = pm ? 63 - clz64(pm) : 0u;
                const uint32_t pfl = (uint32_t)__shfl((int)fl[r], (int)sp);
                uint32_t prev_end;
                if (pm) prev_end = t0 + wbase + 64 * r + sp + pfl;
                else if (r == 1 && fsel[0]) prev_end = end_g0;
                else prev_end = lit_start_w;
                if ((fsel[r] >> lane) & 1) {
                    uint32_t rank = (uint32_t)__popcll(pm) + (r ? nsel0 : 0u);
                    uint32_t idx = seq_base + rank;
                    if (idx < SEQ_CAP) bseq[idx] = seq_pack(q[r] - prev_end, fl[r], off[r]);
                }
                if (islit[r]) { uint32_t li = (q[r] - blk_start) - (len_base + pre[r]); if (li < BLK_SIZE) blit[li] = (uint8_t)lo[r]; }
            }
            LZ_STAMP(6);
        } // tiles
        if (tid == 0) { blk[gblk].nseq = seq_run; blk[gblk].nlit = (blk_end - blk_start) - len_run; }
    } // blocks
    if (STAMP && tid == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_lz_stamps[k], st_acc[k]);
}

void launch_lz(const uint8_t *src, const SegDesc *segs, uint32_t nseg, uint64_t *seqs, uint8_t *lits, BlkInfo *blk,
               uint32_t flags, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_lz<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_TOTAL);
        (void)hipFuncSetAttribute((const void *)k_lz<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L_TOTAL);
        attr_set = true;
    }
    if (flags & FLAG_STAMP) hipLaunchKernelGGL(k_lz<true>, dim3(nseg), dim3(LZ_THREADS), L_TOTAL, st, src, segs, seqs, lits, blk, flags);
    else hipLaunchKernelGGL(k_lz<false>, dim3(nseg), dim3(LZ_THREADS), L_TOTAL, st, src, segs, seqs, lits, blk, flags);
}

// diagnostic: read and clear the phase stamps (cycles summed over workgroups)
void lz_read_stamps(unsigned long long *out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lz_stamps), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lz_stamps), z, sizeof(z));
}

} // namespace pna
