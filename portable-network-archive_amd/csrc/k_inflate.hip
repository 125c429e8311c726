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
typedef uint32_t u32u_i __attribute__((aligned(1)));

constexpr uint32_t IF_LROOT = 11, IF_DROOT = 10;          // bits resolved by the first-level tables
constexpr uint32_t IF_RING = 512;                          // input ring, dwords (two halves of 256)
constexpr uint32_t IF_HALF = 256;
constexpr uint32_t IF_PHASE_TOKENS = 128;                  // tokens per phase: at most 48 bits each, 768 bytes < one ring half
constexpr uint32_t IF_SEQ_STAGE = IF_PHASE_TOKENS, IF_LIT_STAGE = IF_PHASE_TOKENS;
constexpr uint32_t IF_LSUB = 143, IF_DSUB = 15;            // second-level tables (16 / 32 cells each): a complete code has at most this many long prefixes
constexpr uint32_t IF_LL_SPLIT = 0x20000;                  // literal runs are cut into records of at most this many bytes (18-bit field: zrec_pack)
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

// CHUNK MODE (chunks != nullptr; round 4): one LARGE stream of a foreign encoder -- no sync-flush markers to cut it at, one wave's walk is a few MiB/s -- is walked by
// one wave per CHUNK: [start_bit, end_bit) of the stream's bits, block starts that k_ispec found by testing every bit position for a well-formed dynamic block header
// (complete code-length code, complete literal / length and distance codes, an end-of-block code: the chance of a false one is negligible, and a false one breaks the
// chain -- a chunk must end exactly where the next begins -- which sends the stream to the serial walk).  The records need no window (matches are only recorded), so a
// chunk starts from nothing.  Two passes: COUNT (sizes of every chunk's literals, records and output), then -- prefix sums on the host -- EMIT into the stream's regions,
// one ZBlock per chunk; k_zexec_par executes them.
struct ISChunk {
    uint64_t start_bit, end_bit;                          // end_bit = ~0: up to the stream's trailer
    uint64_t lit_base, out_base, rec_base;                // EMIT: the chunk's first literal / output byte / record, relative to the stream's
    uint64_t end_found, mtot;                             // out: the bit position behind the chunk's last block (behind the trailer for the last chunk); bytes produced by matches
    uint32_t nlit, nrec, status, adler;                   // out
};
static_assert(sizeof(ISChunk) == 72, "ISChunk layout");
enum { IF_CHAIN = 4 };                                    // chunk mode: the walk passed its end_bit (the next chunk's start was not a block start)

__global__ __launch_bounds__(64)
void k_inflate(ZFrame *__restrict__ frames, ZFrameX *__restrict__ fx, const uint8_t *__restrict__ src, ZBlock *__restrict__ blocks,
               uint8_t *__restrict__ lit_scratch, uint64_t *__restrict__ seqs, const uint32_t *__restrict__ mode,
               ISChunk *__restrict__ chunks, uint32_t chunk_frame, uint32_t emit) {
    __shared__ uint16_t lt[1u << IF_LROOT], dt[1u << IF_DROOT];
    __shared__ uint32_t ring[IF_RING];
    __shared__ uint64_t sstage[IF_SEQ_STAGE];
    __shared__ uint8_t lstage[IF_LIT_STAGE];
    __shared__ uint8_t lens[320 + 8];
    __shared__ uint8_t cltab[128];
    __shared__ uint16_t lt2[IF_LSUB << (15 - IF_LROOT)], dt2[IF_DSUB << (15 - IF_DROOT)], fst[16];
    const uint32_t lane = threadIdx.x, f = chunks ? chunk_frame : blockIdx.x, ck = blockIdx.x;
    const bool l0 = lane == 0, emit_on = !chunks || emit != 0;
    if (IF_U(frames[f].status)) return;
    if (!chunks && mode && IF_U(mode[f]) != 1u) return;                    // lane-per-piece decode (or the chunk mode) took this stream (VM_SERIAL = 1)
    const uint64_t src_off = (uint64_t)IF_U((uint32_t)frames[f].src_off) | ((uint64_t)IF_U((uint32_t)(frames[f].src_off >> 32)) << 32);
    const uint64_t dst_off = (uint64_t)IF_U((uint32_t)frames[f].dst_off) | ((uint64_t)IF_U((uint32_t)(frames[f].dst_off >> 32)) << 32);
    if (IF_U((uint32_t)(frames[f].src_len >> 32)) | (chunks ? 0u : IF_U((uint32_t)(frames[f].dst_len >> 32)))) {      // this walk counts in 32 bits: streams of 4 GiB and more are decoded by pieces, in chunks
                                                                                                                           // (whose output positions are 64-bit: only the COMPRESSED bytes must fit 32 bits there) or not at all
        if (l0) { frames[f].status = IF_UNSUPPORTED; fx[f].nblk = 0; }
        return;
    }
    const uint32_t src_len = IF_U((uint32_t)frames[f].src_len);
    const uint64_t dst_len = (uint64_t)IF_U((uint32_t)frames[f].dst_len) | ((uint64_t)IF_U((uint32_t)(frames[f].dst_len >> 32)) << 32);
    const bool open = (IF_U(frames[f].out_len) & ZF_OPEN) != 0;        // dst_len is a capacity: the stream's size is reported back
    const uint64_t seq_base = (uint64_t)IF_U((uint32_t)fx[f].seq_base) | ((uint64_t)IF_U((uint32_t)(fx[f].seq_base >> 32)) << 32);
    const uint32_t seq_cap = IF_U(fx[f].seq_cap), blk_base = IF_U(fx[f].blk_base), blk_cap = IF_U(fx[f].blk_cap);
    const uint64_t a0 = src_off & ~(uint64_t)3;
    const uint32_t mis = (uint32_t)(src_off & 3);
    const uint64_t end_bytes = (uint64_t)mis + src_len;                    // stream end, relative to a0
    const uint32_t nwords = (uint32_t)((end_bytes + 3) >> 2);
    const uint32_t *gsrc = (const uint32_t *)(src + a0);
    auto gload = [&](uint32_t w) -> uint32_t { return w < nwords ? gsrc[w] : 0u; };
    auto u64u = [&](const uint64_t &v) -> uint64_t { return (uint64_t)IF_U((uint32_t)v) | ((uint64_t)IF_U((uint32_t)(v >> 32)) << 32); };
    const uint64_t sbit = chunks ? u64u(chunks[ck].start_bit) : 0, ebit = chunks ? u64u(chunks[ck].end_bit) : ~0ull;
    const uint64_t lit_base = chunks ? u64u(chunks[ck].lit_base) : 0, rec_base = chunks ? u64u(chunks[ck].rec_base) : 0, out_base = chunks ? u64u(chunks[ck].out_base) : 0;
    uint8_t *lit_out = lit_scratch + dst_off + lit_base;
    uint64_t *rec_out = seqs + seq_base + rec_base;

    // the reader starts at bit `sbit` of the stream (0 but for the chunks of a large stream)
    const uint64_t abit = (uint64_t)mis * 8 + sbit;
    const uint32_t wi0 = (uint32_t)(abit >> 5), boff = (uint32_t)abit & 31u;
    uint32_t rbase = wi0 & ~(IF_HALF - 1);
    uint32_t pend[4];
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint32_t w = rbase + lane + 64 * (uint32_t)k; ring[w & (IF_RING - 1)] = gload(w); }
#pragma unroll
    for (int k = 0; k < 4; k++) pend[k] = gload(rbase + IF_RING + lane + 64 * (uint32_t)k);
    uint32_t wi = wi0 + 1, bitcnt = 32 - boff;
    uint64_t bitbuf = (uint64_t)(IF_U(gload(wi0)) >> boff);
    uint32_t state = sbit ? IST_BLOCK : IST_ZHEAD, last = 0, ll = 0, status = IF_OK, adler = 0;
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
        if (chunks && state == IST_BLOCK) {                                 // a chunk ends at the block start that is the next chunk's first bit
            const uint64_t bp = (uint64_t)wi * 32 - bitcnt - (uint64_t)mis * 8;
            if (bp > ebit) status = IF_CHAIN;
            if (bp >= ebit) break;
        }
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
                    if (++ll == IF_LL_SPLIT) { if (l0) sstage[nq] = zrec_pack(IF_LL_SPLIT, 0, 4); nq++; ll = 0; }
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
                if (l0) sstage[nq] = zrec_pack(ll, ml, dist + 3);
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
        if (out_base + nlit_tot + nl + mtot > dst_len) { if (status == IF_OK) status = IF_DSTSIZE; }
        else if (rec_base + nseq_tot + nq > seq_cap) { if (status == IF_OK) status = IF_UNSUPPORTED; }
        else {
            if (emit_on) {
                for (uint32_t k = lane; k < nq; k += 64) rec_out[nseq_tot + k] = sstage[k];
                for (uint32_t k = lane; k < nl; k += 64) lit_out[nlit_tot + k] = lstage[k];
            }
            nseq_tot += nq; nlit_tot += nl;
        }
        // every consumed bit must lie inside the stream
        if (status == IF_OK && (uint64_t)wi * 32 - bitcnt > end_bytes * 8) status = IF_CORRUPT;
        if (status != IF_OK) break;
        if (act == IACT_STORED) {
            const uint32_t len = p0;
            if ((uint64_t)p1 + len > end_bytes) { status = IF_CORRUPT; break; }
            if (out_base + nlit_tot + len + mtot > dst_len) { status = IF_DSTSIZE; break; }
            const uint8_t *sp = src + a0 + p1;
            uint8_t *dp = lit_out + nlit_tot;
            if (emit_on) for (uint32_t i = lane * 8; i < len; i += 512) {
                if (i + 8 <= len) *(if_u64u *)(dp + i) = *(const if_u64u *)(sp + i);
                else for (uint32_t k = i; k < len; k++) dp[k] = sp[k];
            }
            nlit_tot += len; ll += len;
            if (ll >= IF_LL_SPLIT) {
                if (rec_base + nseq_tot + 1 > seq_cap) { status = IF_UNSUPPORTED; break; }
                if (l0 && emit_on) rec_out[nseq_tot] = zrec_pack(IF_LL_SPLIT, 0, 4);
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
    if (chunks) {
        if (status == IF_OK && state != IST_DONE && state != IST_BLOCK) status = IF_CORRUPT;
        if (status == IF_OK && total > 0xFFFFFFFFull) status = IF_UNSUPPORTED;
        if (l0) {
            ISChunk &cr = chunks[ck];
            cr.end_found = (uint64_t)wi * 32 - bitcnt - (uint64_t)mis * 8; cr.mtot = mtot; cr.nlit = nlit_tot; cr.nrec = nseq_tot; cr.status = status; cr.adler = adler;
            if (emit && status == IF_OK && ck < blk_cap) {
                ZBlock b;
                b.body = 0; b.out_off = dst_off + out_base; b.seq_pos = seq_base + rec_base; b.size = 0; b.type = 2;
                b.ltype = 2; b.regen = nlit_tot; b.streams = 1; b.lit_off = 0; b.lit_csize = 0; b.lit_pos = (uint32_t)lit_base;
                b.huf_slot = 0xFFFFFFFFu; b.slot[0] = b.slot[1] = b.slot[2] = 0xFFFFFFFFu;
                b.nseq = nseq_tot; b.seq_off = 0; b.seq_len = 0; b.frame = f; b.out_len = (uint32_t)total; b.status = 0; b.uses_rep = 0;
                for (int k = 0; k < 7; k++) b.pad[k] = 0;
                b.pad[1] = adler; b.pad[2] = (uint32_t)(lit_base >> 32);
                blocks[blk_base + ck] = b;
            }
        }
        return;
    }
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
        if (open && status == IF_OK) frames[f].dst_len = total;
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
    if ((uint32_t)((b << 16) | a) != blocks[fx[f].blk_base + fx[f].nblk - 1].pad[1]) frames[f].status = IF_CORRUPT;   // the trailer was read with the last block
}

void launch_inflate(ZFrame *frames, ZFrameX *fx, uint32_t n, const uint8_t *src, ZBlock *blocks, uint8_t *lit_scratch, uint64_t *seqs, const uint32_t *mode, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_inflate, dim3(n), dim3(64), 0, st, frames, fx, src, blocks, lit_scratch, seqs, mode, (ISChunk *)nullptr, 0u, 0u);
}
// chunk mode: the `nchunks` chunks (device array of ISChunk, 72 bytes each) of stream `frame`; emit = 0: count only
void launch_inflate_chunks(ZFrame *frames, ZFrameX *fx, uint32_t frame, void *chunks, uint32_t nchunks, uint32_t emit, const uint8_t *src, ZBlock *blocks,
                           uint8_t *lit_scratch, uint64_t *seqs, hipStream_t st) {
    if (nchunks) hipLaunchKernelGGL(k_inflate, dim3(nchunks), dim3(64), 0, st, frames, fx, src, blocks, lit_scratch, seqs, (const uint32_t *)nullptr, (ISChunk *)chunks, frame, emit);
}
void launch_iadler(ZFrame *frames, const ZFrameX *fx, const ZBlock *blocks, uint32_t n, const uint32_t *cbase, uint32_t npieces, const uint8_t *dst,
                   void *part, hipStream_t st) {
    if (!n) return;
    if (npieces) hipLaunchKernelGGL(k_iadler_part, dim3(npieces), dim3(256), 0, st, (const ZFrame *)frames, cbase, n, dst, (uint2 *)part);
    hipLaunchKernelGGL(k_iadler_fin, dim3((n + 255) / 256), dim3(256), 0, st, frames, fx, blocks, cbase, (const uint2 *)part, n);
}

// ------------------------------------------------------------------ k_ispec: block starts of a large foreign stream, by trial
// One workgroup per chunk of `cbytes` compressed bytes: the first bit position in the chunk at which a DYNAMIC block with BFINAL = 0 can start -- BTYPE 2, HLIT / HDIST
// in range, a complete code-length code, and the HLIT + HDIST lengths it codes forming complete literal / length and distance codes (or a single one-bit code, as
// zlib's inflate_table accepts) with an end-of-block code.  256 bit positions per round, one per thread: the three-bit lengths' Kraft sum is the cheap filter (a few in
// a thousand positions pass), the lengths are then decoded canonically on the lane (puff's scheme: counts per length and the symbols in canonical order, held in
// registers).  start[c] = the bit position relative to the stream, or ~0 (none: the chunk before simply runs on through this one).
__device__ __forceinline__ uint64_t isp_bits(const uint8_t *s, uint64_t bit) { return *(const if_u64u *)(s + (bit >> 3)) >> (bit & 7); }     // >= 57 valid bits
__device__ bool isp_header_ok(const uint8_t *s, uint64_t p, uint64_t nbits) {
    if (p + 17 + 57 + 64 > nbits) return false;                                  // (headers in the stream's last bytes are not needed as chunk starts)
    const uint64_t h = isp_bits(s, p);
    if ((h & 7u) != 4u) return false;                                            // BFINAL = 0, BTYPE = 2 (LSB first: 0, then 01b -> value 2)
    const uint32_t hlit = (uint32_t)(h >> 3) & 31u, hdist = (uint32_t)(h >> 8) & 31u, hclen = ((uint32_t)(h >> 13) & 15u) + 4u;
    if (hlit > 29u || hdist > 29u) return false;
    // the code-length code's lengths, by symbol (3 bits each)
    constexpr uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const uint64_t c0 = isp_bits(s, p + 17);                                     // 19 x 3 = 57 bits: all of them
    uint64_t cl = 0, cntp = 0;                                                   // lengths: 3 bits per symbol; symbols per length: 5 bits per length (registers, no indexed arrays)
    uint32_t kraft = 0;
#pragma unroll
    for (uint32_t i = 0; i < 19; i++) {
        const uint32_t v = i < hclen ? (uint32_t)(c0 >> (3 * i)) & 7u : 0u;
        cl |= (uint64_t)v << (3 * ORDER[i]);
        if (v) { kraft += 128u >> v; cntp += 1ull << (5 * v); }
    }
    if (kraft != 128u) return false;
    // symbols in canonical order (by length, then value): 19 x 5 bits
    uint64_t so0 = 0, so1 = 0; uint32_t ns = 0;
    for (uint32_t l = 1; l < 8; l++)
        for (uint32_t sy = 0; sy < 19; sy++)
            if (((uint32_t)(cl >> (3 * sy)) & 7u) == l) { if (ns < 12) so0 |= (uint64_t)sy << (5 * ns); else so1 |= (uint64_t)sy << (5 * (ns - 12)); ns++; }
    // the HLIT + HDIST code lengths
    const uint32_t nll = hlit + 257u, total = nll + hdist + 1u;
    uint64_t bp = p + 17 + 3ull * hclen;
    uint32_t n = 0, prev = 0, kll = 0, kd = 0, maxll = 0, maxd = 0, eob = 0;
    auto add = [&](uint32_t len, uint32_t times) {                                // `times` symbols of length `len` from position n on
        for (uint32_t t = 0; t < times; t++, n++) {
            if (!len) continue;
            if (n < nll) { kll += 32768u >> len; if (len > maxll) maxll = len; if (n == 256) eob = len; }
            else { kd += 32768u >> len; if (len > maxd) maxd = len; }
        }
    };
    while (n < total) {
        if (bp + 64 > nbits) return false;
        uint64_t bb = isp_bits(s, bp);
        uint32_t code = 0, first = 0, index = 0, sym = 99, used = 0;
        for (uint32_t l = 1; l < 8; l++) {
            code |= (uint32_t)bb & 1u; bb >>= 1; used++;
            const uint32_t c = (uint32_t)(cntp >> (5 * l)) & 31u;
            if (code - first < c) { const uint32_t k = index + (code - first); sym = k < 12 ? (uint32_t)(so0 >> (5 * k)) & 31u : (uint32_t)(so1 >> (5 * (k - 12))) & 31u; break; }
            index += c; first = (first + c) << 1; code <<= 1;
        }
        if (sym == 99) return false;
        uint32_t rep = 1, val = sym;
        if (sym < 16) prev = sym;
        else if (sym == 16) { if (n == 0) return false; rep = 3 + ((uint32_t)bb & 3u); used += 2; val = prev; }
        else if (sym == 17) { rep = 3 + ((uint32_t)bb & 7u); used += 3; val = 0; prev = 0; }
        else { rep = 11 + ((uint32_t)bb & 127u); used += 7; val = 0; prev = 0; }
        if (n + rep > total) return false;
        add(val, rep);
        bp += used;
    }
    if (!eob) return false;
    if (kll != 32768u && !(kll < 32768u && maxll <= 1)) return false;
    if (kd != 32768u && !(kd < 32768u && maxd <= 1)) return false;
    return true;
}
__global__ __launch_bounds__(256)
void k_ispec(const uint8_t *__restrict__ src, uint64_t src_off, uint64_t src_len, uint32_t cbytes, uint32_t nchunks, uint64_t *__restrict__ start) {
    __shared__ unsigned long long best;
    const uint32_t tid = threadIdx.x, c = blockIdx.x;
    if (c >= nchunks) return;
    if (c == 0) { if (tid == 0) start[0] = 0; return; }                          // the stream's first chunk starts with the zlib header
    if (tid == 0) best = ~0ull;
    __syncthreads();
    const uint8_t *s = src + src_off;
    const uint64_t nbits = src_len * 8, b0 = (uint64_t)c * cbytes * 8, b1 = (uint64_t)(c + 1) * cbytes * 8 < nbits ? (uint64_t)(c + 1) * cbytes * 8 : nbits;
    for (uint64_t base = b0; base < b1; base += 256) {
        const uint64_t p = base + tid;
        if (p < b1 && isp_header_ok(s, p, nbits)) atomicMin(&best, (unsigned long long)p);
        __syncthreads();
        if (best != ~0ull) break;
        __syncthreads();
    }
    if (tid == 0) start[c] = best;
}
void launch_ispec(const uint8_t *src, uint64_t src_off, uint64_t src_len, uint32_t cbytes, uint32_t nchunks, uint64_t *start, hipStream_t st) {
    if (nchunks) hipLaunchKernelGGL(k_ispec, dim3(nchunks), dim3(256), 0, st, src, src_off, src_len, cbytes, nchunks, start);
}

// =====================================================================================================================
// Lane-per-piece inflate (round 2).  The wave-per-stream walk above keeps ONE scalar unit busy per stream and saturates the chip's 256
// scalar units at a few thousand streams (11 GiB/s); here every LANE walks a "piece" on the vector units:
//   * a whole zlib stream whose decoded size is at most BLK_SIZE (small entries: BASELINE configs[4]'s 4 KiB files), or
//   * one sync-flush delimited piece of a larger stream.  This library's deflate encoder closes every 128 KiB block with a sync flush
//     (empty stored block: ... 00 00 FF FF), so a stream of raw_len bytes holds ceil(raw_len / BLK_SIZE) byte-aligned pieces of exactly
//     BLK_SIZE decoded bytes (the last: the rest).  k_imark finds the markers; a stream whose marker count does not fit that pattern, or
//     whose pieces do not decode to exactly those sizes (a foreign encoder, a chance 00 00 FF FF inside compressed data), is handed to
//     the wave-per-stream kernel afterwards -- nothing is assumed that is not checked.
// Decoding is CANONICAL: per lane the symbols of both codes in canonical order in LDS (320 x u16, lane-interleaved: element e of lane l at
// e * 64 + l), the first code and list position of every code length in registers; a symbol = fifteen compare steps of plain ALU work
// plus ONE LDS read.  Lookup tables per lane were measured first (9 + 7 bits, 126 KiB of LDS per wave, one wave per CU): with 64
// independent streams in step some lane misses the table in most rounds, so every round paid the miss path anyway, and the one resident
// wave left three SIMDs of the CU idle (90 ms per 2 048 x 1 MiB against 171 ms for the scalar walk); 40 KiB per wave puts four waves on
// a CU.  The code lengths of a dynamic block are decoded TWICE (count, then place) instead of being stored.  Output: the same
// intermediate form as k_inflate -- literal bytes + (literal run, match length, distance + 3) records, one ZBlock per piece; k_zoff /
// k_zexec / the Adler kernels run unchanged.
constexpr uint32_t VI_LDS = 320 * 64 * 2;                               // 40 KiB per wave: four waves (one per SIMD) share a CU
enum { VM_PIECES = 0, VM_SERIAL = 1 };

struct VPiece { uint32_t frame, j; };

// ---- k_imark_*: piece boundaries.  pb[blk_base + f + j] = start of piece j (relative to the stream, 64 bits), pb[... + P] = end.  G workgroups per
// stream (G from the batch's longest stream: one workgroup scanned a 1 GiB stream's 400 MB for 1.27 s), each over its share of the bytes: a counting pass,
// then -- every workgroup sums the counts in front of it -- the placing pass.
__device__ __forceinline__ bool if_marker(const uint8_t *p, uint64_t i, uint64_t n) { return i + 4 <= n && p[i] == 0 && p[i + 1] == 0 && p[i + 2] == 0xFF && p[i + 3] == 0xFF; }
__global__ __launch_bounds__(256)
void k_imark_count(const ZFrame *__restrict__ frames, const ZFrameX *__restrict__ fx, const uint8_t *__restrict__ src, uint32_t *__restrict__ cntg, uint32_t G) {
    __shared__ uint32_t cnt[256];
    const uint32_t f = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    if (fx[f].blk_cap <= 1) return;
    const ZFrame fr = frames[f];
    const uint8_t *p = src + fr.src_off;
    const uint64_t n = fr.src_len, per_g = (n + G - 1) / G, g0 = per_g * g < n ? per_g * g : n, g1 = g0 + per_g < n ? g0 + per_g : n;
    const uint64_t per = (g1 - g0 + 255) / 256, a = g0 + tid * per < g1 ? g0 + tid * per : g1, e = a + per < g1 ? a + per : g1;
    uint32_t c = 0;
    for (uint64_t i = a; i < e; i++) if (if_marker(p, i, n)) c++;
    cnt[tid] = c;
    __syncthreads();
    for (uint32_t s2 = 128; s2 > 0; s2 >>= 1) { if (tid < s2) cnt[tid] += cnt[tid + s2]; __syncthreads(); }
    if (tid == 0) cntg[(size_t)f * G + g] = cnt[0];
}
__global__ __launch_bounds__(256)
void k_imark_place(const ZFrame *__restrict__ frames, const ZFrameX *__restrict__ fx, const uint8_t *__restrict__ src, uint64_t *__restrict__ pb,
                   uint32_t *__restrict__ mode, const uint32_t *__restrict__ cntg, uint32_t G) {
    __shared__ uint32_t cnt[256];
    __shared__ uint32_t s_before, s_total;
    const uint32_t f = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const ZFrame fr = frames[f];
    const uint32_t P = fx[f].blk_cap, base = fx[f].blk_base + f;          // P + 1 boundary slots per stream
    if (P <= 1) { if (g == 0 && tid == 0) { pb[base] = 0; pb[base + 1] = fr.src_len; mode[f] = VM_PIECES; } return; }
    if (tid == 0) { uint32_t b = 0, t = 0; for (uint32_t h = 0; h < G; h++) { const uint32_t c = cntg[(size_t)f * G + h]; if (h < g) b += c; t += c; } s_before = b; s_total = t; }
    __syncthreads();
    const uint64_t n = fr.src_len;
    const bool pieces = s_total == P - 1;
    if (g == 0 && tid == 0) { mode[f] = pieces ? VM_PIECES : VM_SERIAL; pb[base] = 0; pb[base + P] = n; }
    if (!pieces) return;
    const uint8_t *p = src + fr.src_off;
    const uint64_t per_g = (n + G - 1) / G, g0 = per_g * g < n ? per_g * g : n, g1 = g0 + per_g < n ? g0 + per_g : n;
    const uint64_t per = (g1 - g0 + 255) / 256, a = g0 + tid * per < g1 ? g0 + tid * per : g1, e = a + per < g1 ? a + per : g1;
    uint32_t c = 0;
    for (uint64_t i = a; i < e; i++) if (if_marker(p, i, n)) c++;
    cnt[tid] = c;
    __syncthreads();
    if (tid == 0) { uint32_t r = s_before; for (uint32_t i = 0; i < 256; i++) { const uint32_t t = cnt[i]; cnt[i] = r; r += t; } }
    __syncthreads();
    uint32_t k = cnt[tid];
    for (uint64_t i = a; i < e; i++) if (if_marker(p, i, n)) { pb[base + 1 + k] = i + 4; k++; }
}

// ---- k_icount: sync-flush markers per stream (streams of unknown size: the host sizes the piece list from it); G workgroups per stream add up in count[f] (zeroed before)
__global__ __launch_bounds__(256)
void k_icount(const uint8_t *__restrict__ src, const uint64_t *__restrict__ off, const uint64_t *__restrict__ len, uint32_t *__restrict__ count, uint32_t G) {
    __shared__ uint32_t cnt[256];
    const uint32_t f = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const uint8_t *p = src + off[f];
    const uint64_t n = len[f], per_g = (n + G - 1) / G, g0 = per_g * g < n ? per_g * g : n, g1 = g0 + per_g < n ? g0 + per_g : n;
    const uint64_t per = (g1 - g0 + 255) / 256, a = g0 + tid * per < g1 ? g0 + tid * per : g1, e = a + per < g1 ? a + per : g1;
    uint32_t c = 0;
    for (uint64_t i = a; i < e; i++) if (if_marker(p, i, n)) c++;
    cnt[tid] = c;
    __syncthreads();
    for (uint32_t s2 = 128; s2 > 0; s2 >>= 1) { if (tid < s2) cnt[tid] += cnt[tid + s2]; __syncthreads(); }
    if (tid == 0 && cnt[0]) atomicAdd(&count[f], cnt[0]);
}
void launch_icount(const uint8_t *src, const uint64_t *off, const uint64_t *len, uint32_t n, uint32_t *count, uint32_t G, hipStream_t st) {
    if (!n) return;
    (void)hipMemsetAsync(count, 0, (size_t)n * 4, st);
    hipLaunchKernelGGL(k_icount, dim3(n, G), dim3(256), 0, st, src, off, len, count, G);
}

__global__ __launch_bounds__(64)
void k_vinflate(const ZFrame *__restrict__ frames, const ZFrameX *__restrict__ fx, const VPiece *__restrict__ pieces, uint32_t npieces,
                const uint64_t *__restrict__ pb, const uint32_t *__restrict__ mode, const uint8_t *__restrict__ src, ZBlock *__restrict__ blocks,
                uint8_t *__restrict__ lit_scratch, uint64_t *__restrict__ seqs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t vlds[];
    uint16_t *lsy = (uint16_t *)vlds;                                        // [320][64]: 0..287 literal / length symbols in canonical order, 288.. distance symbols
    const uint32_t lane = threadIdx.x, pi = blockIdx.x * 64 + lane;
    bool live = pi < npieces;
    VPiece pc = live ? pieces[pi] : VPiece{0, 0};
    const ZFrame fr = frames[pc.frame];
    const ZFrameX x = fx[pc.frame];
    if (live && (fr.status || mode[pc.frame] != VM_PIECES)) live = false;
    const uint32_t P = x.blk_cap;
    const uint32_t bslot = x.blk_base + pc.frame;
    // the piece's bytes: positions below are relative to its start (the stream may be longer than 32 bits count, a piece is not)
    const uint64_t p_start = live ? pb[bslot + pc.j] : 0u, p_stop = live ? pb[bslot + pc.j + 1] : 0u;
    const uint32_t p_end = (uint32_t)(p_stop - p_start < 0xFFFFFFF0ull ? p_stop - p_start : 0xFFFFFFF0ull);
    const bool first = pc.j == 0, lastp = pc.j + 1 == P;
    const bool open = (fr.out_len & ZF_OPEN) != 0;                           // dst_len is a capacity: the last piece's size is found by decoding it
    const uint64_t room = (uint64_t)fr.dst_len > (uint64_t)pc.j * BLK_SIZE ? (uint64_t)fr.dst_len - (uint64_t)pc.j * BLK_SIZE : 0;
    const uint32_t expect = lastp ? (uint32_t)(open && room > BLK_SIZE ? BLK_SIZE : room) : BLK_SIZE;
    const uint8_t *sp = src + fr.src_off + p_start;
    uint8_t *lit_out = lit_scratch + fr.dst_off + (size_t)pc.j * BLK_SIZE;
    const uint32_t pcap = x.pcap;
    uint64_t *rec_out = seqs + x.seq_base + (size_t)pc.j * pcap;
#define LS(e)  lsy[(e) * 64 + lane]
    // bit reader: bytes [0, p_end) of the piece, zeros beyond; the three words after the buffer are always on their way
    uint32_t pos = 0;                                                        // first byte not yet in the buffer
    uint64_t bitbuf = 0; uint32_t bitcnt = 0;
    auto load4 = [&](uint32_t at) -> uint32_t {
        if (at + 4 <= p_end) return *(const u32u_i *)(sp + at);
        uint32_t v = 0; for (uint32_t k = 0; k < 4; k++) if (at + k < p_end) v |= (uint32_t)sp[at + k] << (8 * k); return v;
    };
    uint32_t ahead = live ? load4(pos) : 0u, ahead1 = live ? load4(pos + 4) : 0u, ahead2 = live ? load4(pos + 8) : 0u;
    auto refill = [&]() { if (bitcnt <= 32) { bitbuf |= (uint64_t)ahead << bitcnt; bitcnt += 32; pos += 4; ahead = ahead1; ahead1 = ahead2; ahead2 = load4(pos + 8); } };
    auto take = [&](uint32_t nb) -> uint32_t { const uint32_t v = (uint32_t)bitbuf & ((1u << nb) - 1u); bitbuf >>= nb; bitcnt -= nb; return v; };
    auto bytepos = [&]() -> uint32_t { return pos - (bitcnt >> 3); };          // byte position of the next unread bit (when byte-aligned)
    enum { S_ZHEAD, S_BLOCK, S_TOKENS, S_TRAILER, S_DONE };
    uint32_t state = live ? (first ? S_ZHEAD : S_BLOCK) : S_DONE, status = IF_OK, lastblk = 0;
    uint32_t nlit = 0, nseq = 0, ll = 0, adler = 0; uint64_t mtot = 0;
    int64_t minsrc = 0;                                                      // the earliest byte a match of this piece copies from
    uint32_t litw = 0;                                                       // up to 3 literals waiting for a dword store
    auto put_lit = [&](uint32_t b) {
        litw |= b << (8 * (nlit & 3)); nlit++;
        if (!(nlit & 3)) { *(u32u_i *)(lit_out + nlit - 4) = litw; litw = 0; }
    };
    // canonical codes, per length L = 1..15: first code (lfc / dfc), position of its first symbol in LS (lbs / dbs; entry 16 = end)
    uint32_t lfc[16], lbs[17], dfc[16], dbs[17];
#pragma unroll
    for (int q = 0; q < 16; q++) { lfc[q] = 0; lbs[q] = 0; dfc[q] = 0; dbs[q] = 288; }
    lbs[16] = 0; dbs[16] = 288;
    // counts per length -> first codes and list positions; false for an invalid code set (zlib's inflate_table rules)
    auto layout = [&](const uint32_t (&cnt)[16], uint32_t (&fc)[16], uint32_t (&bs)[17], uint32_t base) -> bool {
        int left = 1; uint32_t maxl = 0, code = 0, run = base;
        fc[0] = 0; bs[0] = base;
#pragma unroll
        for (int L = 1; L < 16; L++) { left = (left << 1) - (int)cnt[L]; if (cnt[L]) maxl = (uint32_t)L; fc[L] = code; bs[L] = run; code = (code + cnt[L]) << 1; run += cnt[L]; }
        bs[16] = run;
        return !(left < 0 || (left > 0 && maxl > 1));
    };
    auto decode = [&](const uint32_t (&fc)[16], const uint32_t (&bs)[17]) -> uint32_t {   // symbol | length << 12, 0 = no code
        const uint32_t rv = __builtin_bitreverse32((uint32_t)bitbuf);         // the stream's next bits, first bit in the top position
        uint32_t idx = 0, len = 0;
#pragma unroll
        for (int L = 15; L >= 1; L--) { const uint32_t d = (rv >> (32 - L)) - fc[L]; if (d < bs[L + 1] - bs[L]) { idx = bs[L] + d; len = (uint32_t)L; } }
        return len ? (uint32_t)LS(idx) | (len << 12) : 0u;
    };
    auto bump = [&](uint32_t (&a)[16], uint32_t L, uint32_t by) {
#pragma unroll
        for (int q = 1; q < 16; q++) a[q] += L == (uint32_t)q ? by : 0u;
    };
    for (uint32_t guard = 0; __ballot(state != S_DONE) != 0; guard++) {
        if (state == S_ZHEAD) {
            refill();
            const uint32_t cmf = take(8), flg = take(8);
            if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0) status = IF_CORRUPT;
            else if (flg & 0x20) status = IF_UNSUPPORTED;
            state = status ? S_DONE : S_BLOCK;
        } else if (state == S_BLOCK) {
            refill();
            lastblk = take(1);
            const uint32_t btype = take(2);
            if (btype == 0) {
                (void)take(bitcnt & 7);
                refill();
                const uint32_t len = take(16); refill(); const uint32_t nlen = take(16);
                if ((len ^ nlen) != 0xFFFFu) { status = IF_CORRUPT; state = S_DONE; }
                else {
                    uint32_t at = bytepos();
                    if (at + len > p_end || (uint64_t)nlit + len + mtot > expect) { status = at + len > p_end ? IF_CORRUPT : IF_DSTSIZE; state = S_DONE; }
                    else {
                        for (uint32_t k = 0; k < len; k++) put_lit(sp[at + k]);
                        ll += len; at += len;
                        pos = at; bitbuf = 0; bitcnt = 0; ahead = load4(pos); ahead1 = load4(pos + 4); ahead2 = load4(pos + 8);
                        // the empty stored block that closes a piece: the reader stands at the piece's end
                        if (lastblk) state = S_TRAILER;
                        else if (!lastp && at == p_end) state = S_DONE;
                    }
                }
            } else if (btype == 1) {
                // fixed code: lengths 7 (256..279), 8 (0..143, 280..287), 9 (144..255); 32 distance codes of 5 bits
                uint32_t cl[16], cd[16];
#pragma unroll
                for (int q = 0; q < 16; q++) { cl[q] = 0; cd[q] = 0; }
                cl[7] = 24; cl[8] = 152; cl[9] = 112; cd[5] = 32;
                (void)layout(cl, lfc, lbs, 0); (void)layout(cd, dfc, dbs, 288);
                for (uint32_t i = 0; i < 24; i++) LS(i) = (uint16_t)(256 + i);
                for (uint32_t i = 0; i < 144; i++) LS(24 + i) = (uint16_t)i;
                for (uint32_t i = 0; i < 8; i++) LS(168 + i) = (uint16_t)(280 + i);
                for (uint32_t i = 0; i < 112; i++) LS(176 + i) = (uint16_t)(144 + i);
                for (uint32_t i = 0; i < 32; i++) LS(288 + i) = (uint16_t)i;
                state = S_TOKENS;
            } else if (btype == 2) {
                refill();
                const uint32_t hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
                if (hlit > 286 || hdist > 30) { status = IF_CORRUPT; state = S_DONE; }
                else {
                    constexpr uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                    uint64_t clp = 0;
                    for (uint32_t i = 0; i < hclen; i++) { refill(); clp |= (uint64_t)take(3) << (3 * ORDER[i]); }
                    // the code length code: 19 symbols, lengths <= 7, decoded by matching canonical codes in symbol order
                    uint32_t ccnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (uint32_t s2 = 0; s2 < 19; s2++) { const uint32_t L = (uint32_t)(clp >> (3 * s2)) & 7;
#pragma unroll
                        for (int q = 1; q < 8; q++) ccnt[q] += L == (uint32_t)q ? 1u : 0u; }
                    int left = 1; uint32_t cfirst[8]; { uint32_t code = 0;
#pragma unroll
                        for (int L = 1; L < 8; L++) { left = (left << 1) - (int)ccnt[L]; cfirst[L] = code; code = (code + ccnt[L]) << 1; } }
                    if (left != 0) { status = IF_CORRUPT; state = S_DONE; }
                    else {
                        const uint32_t total = hlit + hdist;
                        // the code lengths are walked twice from here: pass 0 counts the lengths, pass 1 places the symbols
                        const uint32_t s_pos = pos, s_cnt = bitcnt, s_a = ahead, s_a1 = ahead1, s_a2 = ahead2; const uint64_t s_buf = bitbuf;
                        uint32_t cl[16], cd[16], rl[16], rd[16];
#pragma unroll
                        for (int q = 0; q < 16; q++) { cl[q] = 0; cd[q] = 0; rl[q] = 0; rd[q] = 0; }
                        bool eob = false;
                        for (uint32_t pass = 0; pass < 2 && status == IF_OK; pass++) {
                            if (pass == 1) {
                                pos = s_pos; bitcnt = s_cnt; ahead = s_a; ahead1 = s_a1; ahead2 = s_a2; bitbuf = s_buf;
                                if (!eob || !layout(cl, lfc, lbs, 0) || !layout(cd, dfc, dbs, 288)) { status = IF_CORRUPT; break; }
#pragma unroll
                                for (int q = 0; q < 16; q++) { rl[q] = lbs[q]; rd[q] = dbs[q]; }
                            }
                            uint32_t n = 0, prev = 0;
                            while (n < total && status == IF_OK) {
                                refill();
                                const uint32_t rv = __builtin_bitreverse32((uint32_t)bitbuf);
                                uint32_t sym = 19, sl = 0, nx[8];
#pragma unroll
                                for (int q = 1; q < 8; q++) nx[q] = cfirst[q];
                                for (uint32_t s2 = 0; s2 < 19 && sym == 19; s2++) {
                                    const uint32_t L = (uint32_t)(clp >> (3 * s2)) & 7;
                                    uint32_t cdv = 0;
#pragma unroll
                                    for (int q = 1; q < 8; q++) if (L == (uint32_t)q) { cdv = nx[q]; nx[q]++; }
                                    if (L && (rv >> (32 - L)) == cdv) { sym = s2; sl = L; }
                                }
                                if (sym == 19) { status = IF_CORRUPT; break; }
                                (void)take(sl);
                                uint32_t rep = 1, val = sym;
                                if (sym < 16) prev = sym;
                                else if (sym == 16) { if (n == 0) { status = IF_CORRUPT; break; } rep = 3 + take(2); val = prev; }
                                else if (sym == 17) { rep = 3 + take(3); val = 0; prev = 0; }
                                else { rep = 11 + take(7); val = 0; prev = 0; }
                                if (n + rep > total) { status = IF_CORRUPT; break; }
                                if (val) {
                                    const uint32_t a = n < hlit ? (rep < hlit - n ? rep : hlit - n) : 0u;   // how many of the run belong to the literal / length code
                                    if (pass == 0) {
                                        bump(cl, val, a); bump(cd, val, rep - a);
                                        if (n <= 256 && 256 < n + rep) eob = true;
                                    } else {
                                        for (uint32_t k = 0; k < rep; k++) {
                                            const uint32_t sy = n + k; const bool isl = sy < hlit;
                                            uint32_t at = 0;
#pragma unroll
                                            for (int q = 1; q < 16; q++) if (val == (uint32_t)q) { at = isl ? rl[q] : rd[q]; if (isl) rl[q]++; else rd[q]++; }
                                            LS(at) = (uint16_t)(isl ? sy : sy - hlit);
                                        }
                                    }
                                }
                                n += rep;
                            }
                        }
                        state = status ? S_DONE : S_TOKENS;
                    }
                }
            } else { status = IF_CORRUPT; state = S_DONE; }
        } else if (state == S_TOKENS) {
            // a handful of tokens per visit, so that lanes in other states are not starved and the loop overhead is shared
            for (uint32_t t = 0; t < 8 && state == S_TOKENS; t++) {
                refill();
                const uint32_t e = decode(lfc, lbs);
                const uint32_t L = e >> 12, sym = e & 0xFFF;
                if (!L) { status = IF_CORRUPT; state = S_DONE; break; }
                (void)take(L);
                if (sym < 256) {
                    if ((uint64_t)nlit + 1 + mtot > expect) { status = IF_DSTSIZE; state = S_DONE; break; }
                    put_lit(sym); ll++;
                    continue;
                }
                if (sym == 256) { state = lastblk ? S_TRAILER : S_BLOCK; break; }
                const uint32_t li = sym - 257;
                if (li > 28) { status = IF_CORRUPT; state = S_DONE; break; }
                uint32_t ml;
                if (li < 8) ml = 3 + li;
                else if (li == 28) ml = 258;
                else { const uint32_t eb = (li >> 2) - 1; ml = 3 + ((4 + (li & 3)) << eb) + take(eb); }
                refill();
                const uint32_t d = decode(dfc, dbs);
                const uint32_t dl = d >> 12, ds = d & 0xFFF;
                if (!dl || ds > 29) { status = IF_CORRUPT; state = S_DONE; break; }
                (void)take(dl);
                uint32_t dist;
                if (ds < 4) dist = 1 + ds;
                else { const uint32_t eb = (ds >> 1) - 1; refill(); dist = 1 + ((2 + (ds & 1)) << eb) + take(eb); }
                if (nseq >= pcap || (uint64_t)nlit + mtot + ml > expect) { status = nseq >= pcap ? IF_UNSUPPORTED : IF_DSTSIZE; state = S_DONE; break; }
                rec_out[nseq++] = zrec_pack(ll, ml, dist + 3);
                { const int64_t sp0 = (int64_t)((uint64_t)nlit + mtot) - (int64_t)dist; minsrc = sp0 < minsrc ? sp0 : minsrc; }   // (piece-relative; negative: an earlier piece)
                mtot += ml; ll = 0;
            }
        } else if (state == S_TRAILER) {
            (void)take(bitcnt & 7);
            refill();
            const uint32_t b0 = take(8), b1 = take(8), b2 = take(8), b3 = take(8);
            adler = (b0 << 24) | (b1 << 16) | (b2 << 8) | b3;
            if (!lastp || bytepos() != p_end) status = IF_CORRUPT;            // the trailer belongs to the last piece and ends the stream
            state = S_DONE;
        }
        if (state != S_DONE && ((uint64_t)pos * 8 - bitcnt > (uint64_t)p_end * 8 + 64 || guard > (1u << 24))) { status = IF_CORRUPT; state = S_DONE; }
    }
    if (!live) return;
    if (nlit & 3) { for (uint32_t k = 0; k < (nlit & 3); k++) lit_out[(nlit & ~3u) + k] = (uint8_t)(litw >> (8 * k)); }
    const uint64_t total = (uint64_t)nlit + mtot;
    if (status == IF_OK && (open && lastp ? total > expect : total != expect)) status = IF_DSTSIZE;
    ZBlock b;
    b.body = 0; b.out_off = 0; b.seq_pos = x.seq_base + (uint64_t)pc.j * pcap; b.size = 0; b.type = 2;
    b.ltype = 2; b.regen = nlit; b.streams = 1; b.lit_off = 0; b.lit_csize = 0; b.lit_pos = (uint32_t)((uint64_t)pc.j * BLK_SIZE);
    b.huf_slot = 0xFFFFFFFFu; b.slot[0] = b.slot[1] = b.slot[2] = 0xFFFFFFFFu;
    b.nseq = nseq; b.seq_off = 0; b.seq_len = 0; b.frame = pc.frame; b.out_len = (uint32_t)total; b.status = status; b.uses_rep = 0;
    for (int k = 0; k < 7; k++) b.pad[k] = 0;
    b.pad[1] = adler; b.pad[2] = (uint32_t)(((uint64_t)pc.j * BLK_SIZE) >> 32);
    { const int64_t r = (int64_t)((uint64_t)pc.j * BLK_SIZE) + minsrc; const uint64_t ru = r < 0 ? 0ull : (uint64_t)r; b.pad[3] = (uint32_t)ru; b.pad[4] = (uint32_t)(ru >> 32); }   // how far back the piece reaches (stream position): k_vfin
    blocks[x.blk_base + pc.j] = b;
#undef LS
}

// ---- k_vfin: one thread per stream: all pieces fine -> the frame has P blocks; otherwise the wave-per-stream kernel takes the stream.
// Execution groups: piece j starts one iff no piece k >= j copies from in front of piece j's start (pad[5] = 1) -- the groups of a stream are
// executed side by side (k_zexec_groups).  What this library writes: a group per 1 MiB segment (its matches never leave the segment), so an entry of
// N MiB is executed by N waves instead of one (4 GiB: 40 s on one wave).
__global__ void k_vfin(ZFrame *__restrict__ frames, ZFrameX *__restrict__ fx, uint32_t n, ZBlock *__restrict__ blocks, uint32_t *__restrict__ mode) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n || frames[f].status) return;
    if (mode[f] == VM_PIECES) {
        const uint32_t P = fx[f].blk_cap; uint32_t bad = 0; uint64_t tot = 0;
        for (uint32_t j = 0; j < P; j++) { bad |= blocks[fx[f].blk_base + j].status; tot += blocks[fx[f].blk_base + j].out_len; }
        const bool open = (frames[f].out_len & ZF_OPEN) != 0;
        if (!bad && (open ? tot <= frames[f].dst_len : tot == frames[f].dst_len)) {
            uint64_t sufmin = ~0ull;
            for (uint32_t j = P; j-- > 0;) {
                ZBlock *b = blocks + fx[f].blk_base + j;
                const uint64_t reach = (uint64_t)b->pad[3] | ((uint64_t)b->pad[4] << 32);
                sufmin = reach < sufmin ? reach : sufmin;
                b->pad[5] = sufmin >= (uint64_t)j * BLK_SIZE ? 1u : 0u;
            }
            fx[f].nblk = P; if (!open) frames[f].out_len = (uint32_t)(tot < 0xFFFFFFFFull ? tot : 0xFFFFFFFFull); return;   // (open: k_zoff reports the size)
        }
        mode[f] = VM_SERIAL;
    }
}

void launch_vinflate(ZFrame *frames, ZFrameX *fx, uint32_t n, const void *pieces, uint32_t npieces, uint64_t *pb, uint32_t *mode, uint32_t *cntg, uint32_t G, const uint8_t *src,
                     ZBlock *blocks, uint8_t *lit_scratch, uint64_t *seqs, hipStream_t st) {
    if (!n) return;
    static const hipError_t attr_set = hipFuncSetAttribute((const void *)k_vinflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VI_LDS);
    (void)attr_set;
    hipLaunchKernelGGL(k_imark_count, dim3(n, G), dim3(256), 0, st, (const ZFrame *)frames, (const ZFrameX *)fx, src, cntg, G);
    hipLaunchKernelGGL(k_imark_place, dim3(n, G), dim3(256), 0, st, (const ZFrame *)frames, (const ZFrameX *)fx, src, pb, mode, (const uint32_t *)cntg, G);
    hipLaunchKernelGGL(k_vinflate, dim3((npieces + 63) / 64), dim3(64), VI_LDS, st, (const ZFrame *)frames, (const ZFrameX *)fx, (const VPiece *)pieces, npieces,
                       (const uint64_t *)pb, (const uint32_t *)mode, src, blocks, lit_scratch, seqs);
    hipLaunchKernelGGL(k_vfin, dim3((n + 255) / 256), dim3(256), 0, st, frames, fx, n, blocks, mode);
}

} // namespace pna
