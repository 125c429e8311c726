// k_inflate.hip -- zlib / deflate decoder front end (gfx950), RFC 1950 / RFC 1951.
//   k_inflate      one wave per zlib stream (= one entry's FDAT payload).  Lane 0 walks the Huffman-coded bit stream -- the only
//                  part that is serial by construction -- and turns it into the SAME intermediate form the zstd decoder uses:
//                  a literal byte string plus (literal run, match length, distance) records.  Everything around that walk is
//                  done by all 64 lanes: input ring top-up, code table construction, stored-block copies, flushing the staged
//                  literals / records with coalesced stores.
//   k_zoff/k_zexec (k_zdec.hip) then execute the records, 64 sequences in flight per wave.
//   k_iadler_part / k_iadler_fin   Adler-32 of the produced bytes (64 KiB pieces, then one thread per stream) against the trailer.
// Replaces flate2::read::ZlibDecoder behind decompress_reader (lib/src/entry/read.rs:171-190).  Integer / bit work only.
#include <hip/hip_runtime.h>
#include "pna_dev.h"

namespace pna {

enum { IF_OK = 0, IF_CORRUPT = 1, IF_UNSUPPORTED = 2, IF_DSTSIZE = 3 };          // = ZD_* of k_zdec.hip (ZFrame::status)
typedef unsigned long long if_u64u __attribute__((aligned(1)));

constexpr uint32_t IF_LROOT = 11, IF_DROOT = 10;          // bits resolved by the first-level tables
constexpr uint32_t IF_RING = 512;                          // input ring, dwords (two halves of 256)
constexpr uint32_t IF_HALF = 256;
constexpr uint32_t IF_PHASE_TOKENS = 128;                  // tokens per phase: at most 48 bits each, 768 bytes < one ring half
constexpr uint32_t IF_SEQ_STAGE = IF_PHASE_TOKENS, IF_LIT_STAGE = IF_PHASE_TOKENS;
constexpr uint32_t IF_LSUB = 143, IF_DSUB = 15;            // second-level tables (16 / 32 cells each): a complete code has at most this many long prefixes
constexpr uint32_t IF_LL_SPLIT = 0x80000;                  // literal runs are cut into records of at most this many bytes (20-bit field)
constexpr uint32_t IF_ADLER_PIECE = 65536, IF_ADLER_P = 65521;

enum { IST_ZHEAD = 0, IST_BLOCK = 1, IST_CODES = 2, IST_TRAILER = 3, IST_DONE = 4 };
enum { IACT_NONE = 0, IACT_STORED = 1, IACT_FIXED = 2, IACT_DYN = 3 };

__device__ __forceinline__ uint32_t if_mbcnt(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Canonical Huffman decoding table for `nsym` code lengths (<= 15), built by the whole wave.
//   tab[i], i = next `root` stream bits (LSB first): sym | len << 9, or 0x8000 | k = longer code, resolved by the k-th
//   second-level table tab2[k << (15 - root) | following 15 - root bits] (same cell format), or 0 = no code.
//   Long codes sit at the top of the canonical code space, so their `root`-bit prefixes are consecutive: k = prefix - first prefix.
// Same acceptance as zlib's inflate_table: over-subscribed sets and incomplete sets (other than a single one-bit code or no code
// at all) are rejected.  Returns 0 when the table is usable.
template <int NCH>
__device__ uint32_t if_build(const uint8_t *lens, uint32_t nsym, uint32_t root, uint16_t *tab, uint16_t *tab2, uint32_t nsub, uint16_t *fst,
                             uint32_t lane) {
    for (uint32_t i = lane; i < (1u << root) / 2; i += 64) ((uint32_t *)tab)[i] = 0;
    uint32_t run[16];
#pragma unroll
    for (int L = 0; L < 16; L++) run[L] = 0;
    uint32_t rk[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint32_t s = (uint32_t)c * 64 + lane;
        const uint32_t len = s < nsym ? lens[s] : 0u;
        uint32_t r = 0;
#pragma unroll
        for (int L = 1; L < 16; L++) {
            const uint64_t m = __ballot(len == (uint32_t)L);
            if (len == (uint32_t)L) r = run[L] + if_mbcnt(m);
            run[L] += (uint32_t)__builtin_popcountll(m);
        }
        rk[c] = r;
    }
    int left = 1; uint32_t maxl = 0, p0 = 0;
#pragma unroll
    for (int L = 1; L < 16; L++) { left = (left << 1) - (int)run[L]; if (left < 0) return 1; if (run[L]) maxl = (uint32_t)L; }
    if (left > 0 && maxl > 1) return 1;
    {
        uint32_t code = 0;
#pragma unroll
        for (int L = 1; L < 16; L++) {
            if (lane == 0) fst[L] = (uint16_t)code;
            code = (code + run[L]) << 1;
            if ((uint32_t)L == root) p0 = code >> 1;                                // prefix of the first code longer than `root`
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t bad = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint32_t s = (uint32_t)c * 64 + lane;
        const uint32_t len = s < nsym ? lens[s] : 0u;
        if (len) {
            const uint32_t code = (uint32_t)fst[len] + rk[c];
            const uint32_t rc = __builtin_bitreverse32(code) >> (32 - len);
            const uint16_t ent = (uint16_t)(s | (len << 9));
            if (len <= root) {
                for (uint32_t i = rc; i < (1u << root); i += 1u << len) tab[i] = ent;
            } else {
                const uint32_t k = (code >> (len - root)) - p0, sb = 15 - root;
                if (k >= nsub) bad = 1;
                else {
                    tab[rc & ((1u << root) - 1)] = (uint16_t)(0x8000u | k);
                    for (uint32_t i = rc >> root; i < (1u << sb); i += 1u << (len - root)) tab2[(k << sb) + i] = ent;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return __ballot(bad != 0) ? 1u : 0u;
}

// The walk below is written as UNIFORM code: every lane carries the same reader state and executes the same scalar
// instruction stream (values read from LDS go through readfirstlane), so the compiler keeps it on the scalar unit -- one
// instruction per step instead of a 64-lane vector operation with one live lane.  Only LDS / global stores are lane-guarded.
#define IF_U(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

__global__ __launch_bounds__(64)
void k_inflate(ZFrame *__restrict__ frames, ZFrameX *__restrict__ fx, const uint8_t *__restrict__ src, ZBlock *__restrict__ blocks,
               uint8_t *__restrict__ lit_scratch, uint64_t *__restrict__ seqs) {
    __shared__ uint16_t lt[1u << IF_LROOT], dt[1u << IF_DROOT];
    __shared__ uint32_t ring[IF_RING];
    __shared__ uint64_t sstage[IF_SEQ_STAGE];
    __shared__ uint8_t lstage[IF_LIT_STAGE];
    __shared__ uint8_t lens[320 + 8];
    __shared__ uint8_t cltab[128];
    __shared__ uint16_t lt2[IF_LSUB << (15 - IF_LROOT)], dt2[IF_DSUB << (15 - IF_DROOT)], fst[16];
    const uint32_t lane = threadIdx.x, f = blockIdx.x;
    const bool l0 = lane == 0;
    if (IF_U(frames[f].status)) return;
    const uint64_t src_off = (uint64_t)IF_U((uint32_t)frames[f].src_off) | ((uint64_t)IF_U((uint32_t)(frames[f].src_off >> 32)) << 32);
    const uint64_t dst_off = (uint64_t)IF_U((uint32_t)frames[f].dst_off) | ((uint64_t)IF_U((uint32_t)(frames[f].dst_off >> 32)) << 32);
    const uint32_t src_len = IF_U(frames[f].src_len), dst_len = IF_U(frames[f].dst_len);
    const bool open = (IF_U(frames[f].out_len) & ZF_OPEN) != 0;        // dst_len is a capacity: the stream's size is reported back
    const uint64_t seq_base = (uint64_t)IF_U((uint32_t)fx[f].seq_base) | ((uint64_t)IF_U((uint32_t)(fx[f].seq_base >> 32)) << 32);
    const uint32_t seq_cap = IF_U(fx[f].seq_cap), blk_base = IF_U(fx[f].blk_base), blk_cap = IF_U(fx[f].blk_cap);
    const uint64_t a0 = src_off & ~(uint64_t)3;
    const uint32_t mis = (uint32_t)(src_off & 3);
    const uint64_t end_bytes = (uint64_t)mis + src_len;                    // stream end, relative to a0
    const uint32_t nwords = (uint32_t)((end_bytes + 3) >> 2);
    const uint32_t *gsrc = (const uint32_t *)(src + a0);
    auto gload = [&](uint32_t w) -> uint32_t { return w < nwords ? gsrc[w] : 0u; };
    uint8_t *lit_out = lit_scratch + dst_off;
    uint64_t *rec_out = seqs + seq_base;

    uint32_t rbase = 0;
    uint32_t pend[4];
#pragma unroll
    for (int k = 0; k < 8; k++) ring[lane + 64 * k] = gload(lane + 64 * (uint32_t)k);
#pragma unroll
    for (int k = 0; k < 4; k++) pend[k] = gload(IF_RING + lane + 64 * (uint32_t)k);
    uint32_t wi = 1, bitcnt = 32 - 8 * mis;
    uint64_t bitbuf = (uint64_t)(IF_U(gload(0)) >> (8 * mis));
    uint32_t state = IST_ZHEAD, last = 0, ll = 0, status = IF_OK, adler = 0;
    uint32_t nseq_tot = 0, nlit_tot = 0;
    uint64_t mtot = 0;                                                      // bytes produced by matches
    __builtin_amdgcn_wave_barrier();

    auto refill = [&]() { if (bitcnt <= 32) { bitbuf |= (uint64_t)IF_U(ring[wi & (IF_RING - 1)]) << bitcnt; bitcnt += 32; wi++; } };
    auto take = [&](uint32_t n) -> uint32_t { const uint32_t v = (uint32_t)bitbuf & ((1u << n) - 1u); bitbuf >>= n; bitcnt -= n; return v; };
    while (state != IST_DONE && status == IF_OK) {
        // ---- keep [wi, wi + 256) resident in the ring; the half after it is already on its way in `pend`
        if (wi >= rbase + IF_HALF) {
#pragma unroll
            for (int k = 0; k < 4; k++) ring[(rbase + lane + 64 * (uint32_t)k) & (IF_RING - 1)] = pend[k];
            rbase += IF_HALF;
#pragma unroll
            for (int k = 0; k < 4; k++) pend[k] = gload(rbase + IF_RING + lane + 64 * (uint32_t)k);
            __builtin_amdgcn_wave_barrier();
        }
        // (re-assert uniformity of the carried state: the values are identical in all lanes by construction)
        wi = IF_U(wi); bitcnt = IF_U(bitcnt); bitbuf = (uint64_t)IF_U((uint32_t)bitbuf) | ((uint64_t)IF_U((uint32_t)(bitbuf >> 32)) << 32);
        state = IF_U(state); last = IF_U(last); ll = IF_U(ll); status = IF_U(status); nseq_tot = IF_U(nseq_tot); nlit_tot = IF_U(nlit_tot); rbase = IF_U(rbase);
        mtot = (uint64_t)IF_U((uint32_t)mtot) | ((uint64_t)IF_U((uint32_t)(mtot >> 32)) << 32);
        uint32_t nq = 0, nl = 0, act = IACT_NONE, p0 = 0, p1 = 0;
        if (state == IST_ZHEAD) {
            // RFC 1950: CMF, FLG.  Deflate with a window of at most 32 KiB, header check, no preset dictionary.
            refill();
            const uint32_t cmf = take(8), flg = take(8);
            if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0) status = IF_CORRUPT;
            else if (flg & 0x20) status = IF_UNSUPPORTED;
            state = IST_BLOCK;
        } else if (state == IST_BLOCK) {
            refill();
            last = take(1);
            const uint32_t btype = take(2);
            if (btype == 0) {
                (void)take(bitcnt & 7);
                refill();
                const uint32_t len = take(16), nlen = take(16);
                if ((len ^ nlen) != 0xFFFFu) status = IF_CORRUPT;
                else { act = IACT_STORED; p0 = len; p1 = wi * 4 - (bitcnt >> 3); }          // p1: byte position of the data (bitcnt is a multiple of 8)
                state = last ? IST_TRAILER : IST_BLOCK;
            } else if (btype == 1) { act = IACT_FIXED; state = IST_CODES; }
            else if (btype == 2) {
                refill();
                const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
                if (hlit > 286 || hdist > 30) status = IF_CORRUPT;
                else {
                    uint64_t clp = 0;                                           // 19 three-bit lengths, by symbol
#pragma unroll
                    for (uint32_t i = 0; i < 19; i++) {
                        constexpr uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                        if (i < hclen) { refill(); const uint32_t v = take(3); clp |= (uint64_t)v << (3 * ORDER[i]); }
                    }
                    // code length code: lane s owns symbol s; complete sets only
                    const uint32_t myl = lane < 19 ? (uint32_t)(clp >> (3 * lane)) & 7 : 0u;
                    uint32_t code = 0, mycode = 0; int left = 1;
#pragma unroll
                    for (uint32_t L = 1; L < 8; L++) {
                        const uint64_t m = __ballot(myl == L);
                        if (myl == L) mycode = code + if_mbcnt(m);
                        const uint32_t k = (uint32_t)__builtin_popcountll(m);
                        left = (left << 1) - (int)k; code = (code + k) << 1;
                    }
                    if (left != 0) status = IF_CORRUPT;
                    else {
                        if (myl) {
                            const uint32_t rc = __builtin_bitreverse32(mycode) >> (32 - myl);
                            for (uint32_t i = rc; i < 128; i += 1u << myl) cltab[i] = (uint8_t)(lane | (myl << 5));
                        }
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t total = hlit + hdist;
                        uint32_t n = 0, prev = 0;
                        while (n < total) {
                            refill();
                            const uint32_t e = IF_U(cltab[(uint32_t)bitbuf & 127]);
                            const uint32_t sym = e & 31;
                            (void)take(e >> 5);
                            uint32_t rep = 1, val = sym;
                            if (sym < 16) prev = sym;
                            else if (sym == 16) { if (n == 0) { status = IF_CORRUPT; break; } rep = 3 + take(2); val = prev; }
                            else if (sym == 17) { rep = 3 + take(3); val = 0; prev = 0; }
                            else { rep = 11 + take(7); val = 0; prev = 0; }
                            if (n + rep > total) { status = IF_CORRUPT; break; }
                            for (uint32_t k = lane; k < rep; k += 64) lens[n + k] = (uint8_t)val;
                            n += rep;
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (status == IF_OK && IF_U(lens[256]) == 0) status = IF_CORRUPT;      // no end-of-block code
                        act = IACT_DYN; p0 = hlit; p1 = hdist; state = IST_CODES;
                    }
                }
            } else status = IF_CORRUPT;
        } else if (state == IST_CODES) {
            // One LDS round trip per literal, two per match: the table cell and the next ring word are requested together, and
            // the word tops the bit buffer up after the symbol has been consumed (bitcnt >= 33 at every loop head).
            refill();
            for (uint32_t t = 0; t < IF_PHASE_TOKENS; t++) {
                const uint32_t ev = lt[(uint32_t)bitbuf & ((1u << IF_LROOT) - 1)], wv = ring[wi & (IF_RING - 1)];
                uint32_t e = IF_U(ev);
                const uint32_t w = IF_U(wv);
                if (e & 0x8000u) e = IF_U(lt2[((e & 0x7FFu) << (15 - IF_LROOT)) + ((uint32_t)(bitbuf >> IF_LROOT) & ((1u << (15 - IF_LROOT)) - 1))]);
                uint32_t len = (e >> 9) & 15;
                const uint32_t sym = e & 511;
                if (!len) { status = IF_CORRUPT; break; }
                bitbuf >>= len; bitcnt -= len;
                if (sym < 256) {
                    if (l0) lstage[nl] = (uint8_t)sym;
                    nl++;
                    if (++ll == IF_LL_SPLIT) { if (l0) sstage[nq] = (uint64_t)IF_LL_SPLIT | (4ull << 40); nq++; ll = 0; }
                    if (bitcnt <= 32) { bitbuf |= (uint64_t)w << bitcnt; bitcnt += 32; wi++; }
                    continue;
                }
                if (sym == 256) { state = last ? IST_TRAILER : IST_BLOCK; break; }
                const uint32_t li = sym - 257;
                if (li > 28) { status = IF_CORRUPT; break; }
                uint32_t ml;
                if (li < 8) ml = 3 + li;
                else if (li == 28) ml = 258;
                else { const uint32_t eb = (li >> 2) - 1; ml = 3 + ((4 + (li & 3)) << eb) + take(eb); }
                if (bitcnt <= 32) { bitbuf |= (uint64_t)w << bitcnt; bitcnt += 32; wi++; }
                const uint32_t dv = dt[(uint32_t)bitbuf & ((1u << IF_DROOT) - 1)], wv2 = ring[wi & (IF_RING - 1)];
                e = IF_U(dv);
                const uint32_t w2 = IF_U(wv2);
                if (e & 0x8000u) e = IF_U(dt2[((e & 0x7FFu) << (15 - IF_DROOT)) + ((uint32_t)(bitbuf >> IF_DROOT) & ((1u << (15 - IF_DROOT)) - 1))]);
                len = (e >> 9) & 15;
                const uint32_t ds = e & 511;
                if (!len || ds > 29) { status = IF_CORRUPT; break; }
                bitbuf >>= len; bitcnt -= len;
                uint32_t dist;
                if (ds < 4) dist = 1 + ds;
                else { const uint32_t eb = (ds >> 1) - 1; dist = 1 + ((2 + (ds & 1)) << eb) + take(eb); }
                if (l0) sstage[nq] = (uint64_t)ll | ((uint64_t)ml << 20) | ((uint64_t)(dist + 3) << 40);
                nq++; mtot += ml; ll = 0;
                if (bitcnt <= 32) { bitbuf |= (uint64_t)w2 << bitcnt; bitcnt += 32; wi++; }
            }
        } else {                                                                // IST_TRAILER: Adler-32, big endian, at the next byte boundary
            (void)take(bitcnt & 7);
            refill();
            const uint32_t b0 = take(8), b1 = take(8), b2 = take(8), b3 = take(8);
            adler = (b0 << 24) | (b1 << 16) | (b2 << 8) | b3;
            state = IST_DONE;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- flush what the phase staged
        if ((uint64_t)nlit_tot + nl + mtot > dst_len) { if (status == IF_OK) status = IF_DSTSIZE; }
        else if (nseq_tot + nq > seq_cap) { if (status == IF_OK) status = IF_UNSUPPORTED; }
        else {
            for (uint32_t k = lane; k < nq; k += 64) rec_out[nseq_tot + k] = sstage[k];
            for (uint32_t k = lane; k < nl; k += 64) lit_out[nlit_tot + k] = lstage[k];
            nseq_tot += nq; nlit_tot += nl;
        }
        // every consumed bit must lie inside the stream
        if (status == IF_OK && (uint64_t)wi * 32 - bitcnt > end_bytes * 8) status = IF_CORRUPT;
        if (status != IF_OK) break;
        if (act == IACT_STORED) {
            const uint32_t len = p0;
            if ((uint64_t)p1 + len > end_bytes) { status = IF_CORRUPT; break; }
            if ((uint64_t)nlit_tot + len + mtot > dst_len) { status = IF_DSTSIZE; break; }
            const uint8_t *sp = src + a0 + p1;
            uint8_t *dp = lit_out + nlit_tot;
            for (uint32_t i = lane * 8; i < len; i += 512) {
                if (i + 8 <= len) *(if_u64u *)(dp + i) = *(const if_u64u *)(sp + i);
                else for (uint32_t k = i; k < len; k++) dp[k] = sp[k];
            }
            nlit_tot += len; ll += len;
            if (ll >= IF_LL_SPLIT) {
                if (nseq_tot + 1 > seq_cap) { status = IF_UNSUPPORTED; break; }
                if (l0) rec_out[nseq_tot] = (uint64_t)IF_LL_SPLIT | (4ull << 40);
                nseq_tot++; ll -= IF_LL_SPLIT;
            }
            // re-seat the reader behind the stored bytes
            const uint32_t p = p1 + len;
            wi = p >> 2;
            rbase = wi & ~(IF_HALF - 1);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 8; k++) { const uint32_t w = rbase + lane + 64 * (uint32_t)k; ring[w & (IF_RING - 1)] = gload(w); }
#pragma unroll
            for (int k = 0; k < 4; k++) pend[k] = gload(rbase + IF_RING + lane + 64 * (uint32_t)k);
            bitbuf = (uint64_t)(IF_U(gload(wi)) >> (8 * (p & 3))); bitcnt = 32 - 8 * (p & 3); wi++;
            __builtin_amdgcn_wave_barrier();
        } else if (act == IACT_FIXED || act == IACT_DYN) {
            uint32_t hlit = p0, hdist = p1;
            if (act == IACT_FIXED) {
                hlit = 288; hdist = 32;
                for (uint32_t s = lane; s < 288; s += 64) lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
                if (lane < 32) lens[288 + lane] = 5;
                __builtin_amdgcn_wave_barrier();
            }
            const uint32_t b1 = if_build<5>(lens, hlit, IF_LROOT, lt, lt2, IF_LSUB, fst, lane);
            const uint32_t b2 = if_build<1>(lens + hlit, hdist, IF_DROOT, dt, dt2, IF_DSUB, fst, lane);
            if (b1 | b2) { status = IF_CORRUPT; break; }
        }
    }
    // ---- one block record for k_zoff / k_zexec
    const uint64_t total = (uint64_t)nlit_tot + mtot;
    if (status == IF_OK && (open ? total > dst_len : total != dst_len)) status = IF_DSTSIZE;
    if (l0) {
        ZBlock b;
        b.body = 0; b.out_off = dst_off; b.seq_pos = seq_base; b.size = 0; b.type = 2;
        b.ltype = 2; b.regen = nlit_tot; b.streams = 1; b.lit_off = 0; b.lit_csize = 0; b.lit_pos = 0;
        b.huf_slot = 0xFFFFFFFFu; b.slot[0] = b.slot[1] = b.slot[2] = 0xFFFFFFFFu;
        b.nseq = nseq_tot; b.seq_off = 0; b.seq_len = 0; b.frame = f; b.out_len = (uint32_t)total; b.status = 0; b.uses_rep = 0;
        for (int k = 0; k < 7; k++) b.pad[k] = 0;
        b.pad[1] = adler;
        if (blk_cap) blocks[blk_base] = b;
        fx[f].nblk = status == IF_OK ? 1u : 0u;
        frames[f].status = status;
        frames[f].out_len = (uint32_t)total;
        if (open && status == IF_OK) frames[f].dst_len = (uint32_t)total;
    }
}


// ------------------------------------------------------------------ Adler-32 of the produced bytes
// piece c of stream f = bytes [j * 64 KiB, ...) of its output: (S1, S2) with S1 = sum d_i, S2 = sum (n - i) d_i, both mod 65521
__global__ __launch_bounds__(256)
void k_iadler_part(const ZFrame *__restrict__ frames, const uint32_t *__restrict__ cbase, uint32_t n, const uint8_t *__restrict__ dst,
                   uint2 *__restrict__ part) {
    __shared__ unsigned long long r1[256], r2[256];
    const uint32_t tid = threadIdx.x, c = blockIdx.x;
    uint32_t lo = 0, hi = n;                                   // largest f with cbase[f] <= c
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cbase[mid] <= c) lo = mid; else hi = mid; }
    const ZFrame fr = frames[lo];
    const uint64_t start = (uint64_t)(c - cbase[lo]) * IF_ADLER_PIECE;
    const uint32_t len = (fr.status || start >= fr.dst_len) ? 0u : (uint32_t)(fr.dst_len - start < IF_ADLER_PIECE ? fr.dst_len - start : IF_ADLER_PIECE);
    const uint8_t *p = dst + fr.dst_off + start;
    unsigned long long s1 = 0, s2 = 0;
    for (uint32_t i = tid * 8; i < len; i += 256 * 8) {
        if (i + 8 <= len) {
            const unsigned long long v = *(const if_u64u *)(p + i);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) { const uint32_t d = (uint32_t)(v >> (8 * k)) & 0xFF; s1 += d; s2 += (unsigned long long)(len - i - k) * d; }
        } else for (uint32_t k = i; k < len; k++) { const uint32_t d = p[k]; s1 += d; s2 += (unsigned long long)(len - k) * d; }
    }
    r1[tid] = s1; r2[tid] = s2;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) { if (tid < s) { r1[tid] += r1[tid + s]; r2[tid] += r2[tid + s]; } __syncthreads(); }
    if (tid == 0) part[c] = make_uint2((uint32_t)(r1[0] % IF_ADLER_P), (uint32_t)(r2[0] % IF_ADLER_P));
}
__global__ void k_iadler_fin(ZFrame *__restrict__ frames, const ZFrameX *__restrict__ fx, const ZBlock *__restrict__ blocks,
                             const uint32_t *__restrict__ cbase, const uint2 *__restrict__ part, uint32_t n) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n || frames[f].status) return;
    uint64_t a = 1, b = 0, left = frames[f].dst_len;
    for (uint32_t c = cbase[f]; c < cbase[f + 1]; c++) {
        const uint64_t len = left < IF_ADLER_PIECE ? left : IF_ADLER_PIECE;
        const uint2 pr = part[c];
        b = (b + (len % IF_ADLER_P) * a + pr.y) % IF_ADLER_P;
        a = (a + pr.x) % IF_ADLER_P;
        left -= len;
    }
    if ((uint32_t)((b << 16) | a) != blocks[fx[f].blk_base].pad[1]) frames[f].status = IF_CORRUPT;
}

void launch_inflate(ZFrame *frames, ZFrameX *fx, uint32_t n, const uint8_t *src, ZBlock *blocks, uint8_t *lit_scratch, uint64_t *seqs, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_inflate, dim3(n), dim3(64), 0, st, frames, fx, src, blocks, lit_scratch, seqs);
}
void launch_iadler(ZFrame *frames, const ZFrameX *fx, const ZBlock *blocks, uint32_t n, const uint32_t *cbase, uint32_t npieces, const uint8_t *dst,
                   void *part, hipStream_t st) {
    if (!n) return;
    if (npieces) hipLaunchKernelGGL(k_iadler_part, dim3(npieces), dim3(256), 0, st, (const ZFrame *)frames, cbase, n, dst, (uint2 *)part);
    hipLaunchKernelGGL(k_iadler_fin, dim3((n + 255) / 256), dim3(256), 0, st, frames, fx, blocks, cbase, (const uint2 *)part, n);
}

} // namespace pna
