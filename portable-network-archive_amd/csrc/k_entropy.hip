// k_entropy.hip -- entropy stage of the gfx950 zstd-format encoder:
//   k_stats  one workgroup per segment: literal / LL / OF / ML histograms (LDS atomics), then the segment's
//            Huffman code (<= 11 bits) + tree description and the three FSE tables + descriptions.
//   k_lit    one workgroup per block: RLE test and 1- or 4-stream Huffman bit packing (one wave per stream,
//            wave scan of bit lengths, 32-bit atomic OR into the zeroed body).
//   k_hist   one workgroup per block (batches of up to 40 960 blocks): the histograms of k_stats gathered per block into per-segment
//            counters, and every sequence's three codes for k_seqa.
//   k_seqa   three LANES per block, one per FSE stream: the serial tANS state chain -- states only: per sequence and stream the
//            state flush.
//   k_seqb   one workgroup per block: the sequences bitstream assembled token-parallel from those fields and the
//            extra bits (prefix sums of the field lengths, 64-bit ORs into an LDS stage, coalesced stores).
//   k_plan   one wave per segment, a lane per block: block types, which block carries the table descriptions, sizes.
//   k_scan   exclusive scan of segment sizes -> output offsets.
//   k_write  one workgroup per block: frame/block/section headers + payload into the packed output.
// Replaces libzstd's HUF_compress4X / ZSTD_encodeSequences / block+frame assembly behind
// lib/src/entry/write.rs:260-262.  Integer/bit work only.
#include <hip/hip_runtime.h>
#include "pna_dev.h"

namespace pna {

// ------------------------------------------------------------------ code tables (RFC 8878 3.1.1.3.2.1)
__constant__ uint8_t C_LL_CODE[64] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,
                                      22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24};
__constant__ uint8_t C_ML_CODE[128] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
                                       32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
                                       40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
                                       42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42};
__constant__ uint32_t C_LL_BASE[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
__constant__ uint8_t  C_LL_BITS[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
__constant__ uint32_t C_ML_BASE[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
__constant__ uint8_t  C_ML_BITS[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
__constant__ int16_t C_LL_DEF[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
__constant__ int16_t C_ML_DEF[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
__constant__ int16_t C_OF_DEF[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};

__device__ __forceinline__ uint32_t hb(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }
__device__ __forceinline__ uint32_t ll_code(uint32_t v) { return v < 64 ? C_LL_CODE[v] : hb(v) + 19; }
__device__ __forceinline__ uint32_t ml_code(uint32_t ml) { uint32_t b = ml - 3; return b < 128 ? C_ML_CODE[b] : hb(b) + 36; }

// ------------------------------------------------------------------ serial bit writer (thread-private)
struct BitW { uint8_t *p; uint32_t pos; uint64_t acc; uint32_t nb; };
__device__ __forceinline__ void bw_init(BitW &w, uint8_t *p) { w.p = p; w.pos = 0; w.acc = 0; w.nb = 0; }
__device__ __forceinline__ void bw_add(BitW &w, uint32_t v, uint32_t n) {
    if (n == 0) return;
    w.acc |= (uint64_t)(v & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u))) << w.nb; w.nb += n;
    while (w.nb >= 8) { w.p[w.pos++] = (uint8_t)w.acc; w.acc >>= 8; w.nb -= 8; }
}
__device__ __forceinline__ uint32_t bw_close(BitW &w, bool marker) {
    if (marker) bw_add(w, 1, 1);
    if (w.nb > 0) { w.p[w.pos++] = (uint8_t)w.acc; w.acc = 0; w.nb = 0; }
    return w.pos;
}

// ------------------------------------------------------------------ FSE helpers (thread 0 of k_stats)
// counts -> normalised counts summing to 1 << tlog (every present symbol >= 1, no "-1" entries)
__device__ void fse_normalize(const uint32_t *count, int nsym, uint32_t total, int tlog, int16_t *norm) {
    int size = 1 << tlog, sum = 0, best = 0;
    for (int s = 0; s < nsym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        // (counts are below 2^20 -- a segment holds at most 2^20 / 3 sequences, a Huffman tree 256 weights -- and tlog <= 9: the product fits 32 bits, and a 32-bit
        // division is a fifth of the instructions of a 64-bit one, which was a third of this kernel's work for 4 KiB entries)
        uint32_t q = (count[s] << tlog) / total;
        if (q == 0) q = 1;
        norm[s] = (int16_t)q; sum += (int)q;
        if (count[s] > count[best]) best = s;
    }
    while (sum > size) {
        int m = 0;
        for (int s = 1; s < nsym; s++) if (norm[s] > norm[m]) m = s;
        norm[m]--; sum--;
    }
    if (sum < size) norm[best] = (int16_t)(norm[best] + (size - sum));
}

__device__ uint32_t fse_write_ncount(uint8_t *dst, const int16_t *norm, int nsym, int tlog) {
    BitW w; bw_init(w, dst);
    bw_add(w, (uint32_t)(tlog - 5), 4);
    int remaining = (1 << tlog) + 1, threshold = 1 << tlog, nbits = tlog + 1, s = 0;
    while (remaining > 1 && s < nsym) {
        int count = norm[s++];
        int mx = (2 * threshold - 1) - remaining;
        remaining -= count < 0 ? -count : count;
        int v = count + 1;
        if (v >= threshold) v += mx;
        bw_add(w, (uint32_t)v, (uint32_t)(nbits - (v < mx ? 1 : 0)));
        if (count == 0) {
            int z = 0;
            while (s + z < nsym && norm[s + z] == 0) z++;
            s += z;
            while (z >= 3) { bw_add(w, 3, 2); z -= 3; }
            bw_add(w, (uint32_t)z, 2);
        }
        while (remaining < threshold) { nbits--; threshold >>= 1; }
    }
    return bw_close(w, false);
}

// The same table built by a whole WAVE (all 64 lanes call it; norm, cell, tmp192 in LDS -- tmp192[0 .. 127] is scratch here --; nsym <= 64, tlog <= 8).
// Same cells, same states as the serial form above, which walks the table three times on one lane:
//   * the low-probability symbols (norm = -1) take the top cells in symbol order: a ballot and a rank;
//   * spread: position i of the walk is p_i = (i * step) & mask -- a permutation of the cells --, the walk skips the cells above `high`, so the k-th
//     cell it fills is the k-th i with p_i <= high, and it belongs to the symbol whose run of norm[s] cells contains k: a prefix count over the lanes
//     per round of 64 i's, a binary search in the symbols' cumulative counts (the LAST symbol whose count of cells before it is <= k: symbols
//     without walk cells share that count with the next symbol that has some);
//   * the state table lists a symbol's cells in ascending order behind those of the lower symbols: per round of 64 cells one ballot per distinct
//     symbol of the round gives a cell's rank among its symbol's cells, a running count per symbol carries it over the rounds; a symbol's first
//     cell (first_state = table size + cell) is the lowest lane of its first ballot.
__device__ void fse_build_table_wave(SeqTable *t, const int16_t *norm, int nsym, int tlog, uint8_t *cell, uint16_t *tmp192, uint32_t lane) {
    const int size = 1 << tlog, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint16_t *cum = tmp192, *run = tmp192 + 64;
    const uint64_t lane_lt = ((uint64_t)1 << lane) - 1;
    const int nrm = (int)lane < nsym ? (int)norm[lane] : 0;
    const int n_walk = nrm > 0 ? nrm : 0, n_all = nrm == -1 ? 1 : n_walk;
    // low-probability symbols: the top cells, in symbol order
    const uint64_t lowm = __ballot(nrm == -1);
    const int high = size - 1 - (int)__popcll(lowm);
    if (nrm == -1) cell[size - 1 - (int)__popcll(lowm & lane_lt)] = (uint8_t)lane;
    // inclusive scans over the symbols: the walk's cells, all cells
    int pc = n_walk, ac = n_all;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int a2 = __shfl_up(pc, o), b2 = __shfl_up(ac, o); if ((int)lane >= o) { pc += a2; ac += b2; } }
    cum[lane] = (uint16_t)(pc - n_walk);
    __builtin_amdgcn_wave_barrier();
    // spread
    int kbase = 0;
    for (int i0 = 0; i0 < size; i0 += 64) {
        const int i = i0 + (int)lane, pi = (i * step) & mask;
        const bool valid = i < size && pi <= high;
        const uint64_t vm = __ballot(valid);
        if (valid) {
            const int k = kbase + (int)__popcll(vm & lane_lt);
            int lo = 0, hi = nsym - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)cum[mid] <= k) lo = mid; else hi = mid - 1; }
            cell[pi] = (uint8_t)lo;
        }
        kbase += (int)__popcll(vm);
    }
    __builtin_amdgcn_wave_barrier();
    // state table, and every symbol's first cell on the way
    cum[lane] = (uint16_t)(ac - n_all);                          // from here on: the symbol's first state slot
    run[lane] = 0;
    uint32_t first_cell = 0;                                    // (kept by the symbol's own lane)
    __builtin_amdgcn_wave_barrier();
    for (int u0 = 0; u0 < size; u0 += 64) {
        const int u = u0 + (int)lane;
        const bool in = u < size;
        const int sy = in ? (int)cell[u] : -1;
        uint64_t todo = __ballot(in);
        while (todo) {
            const int l0 = (int)__builtin_ctzll(todo);         // the lowest lane not yet served: the lowest lane of its symbol
            const int sx = __shfl(sy, l0);
            const uint64_t m = __ballot(sy == sx);
            const int before = (int)run[sx];
            if (sy == sx) t->state[(int)cum[sx] + before + (int)__popcll(m & lane_lt)] = (uint16_t)(size + u);
            if ((int)lane == sx && before == 0) first_cell = (uint32_t)(size + u0 + l0);
            __builtin_amdgcn_wave_barrier();
            if ((int)lane == l0) run[sx] = (uint16_t)(before + (int)__popcll(m));
            __builtin_amdgcn_wave_barrier();
            todo &= ~m;
        }
    }
    // symbol entries
    if ((int)lane < nsym) {
        SeqSym y; y.delta_nb = 0; y.delta_find = 0; y.first_state = 0;
        if (n_all > 0) {
            const int maxbits = (n_all == 1) ? tlog : tlog - (int)hb((uint32_t)(n_all - 1));
            y.delta_nb = (uint32_t)((maxbits << 16) - (n_all << maxbits));
            y.delta_find = (int16_t)((ac - n_all) - n_all);
            y.first_state = (uint16_t)first_cell;
        }
        t->sym[lane] = y;
    }
}

// ------------------------------------------------------------------ k_stats
constexpr uint32_t ST_THREADS = 256;
constexpr int HUF_MAX = 11;

// Executed by ONE WAVE (all 64 lanes call it; LDS arrays): stable rank sort by count and the leaf depths run on the lanes, the
// two-queue merge (n - 1 dependent steps) and the rare Kraft repair on lane 0.  Returns the number of used symbols, -1 when
// the length limit cannot be met.  Same procedure, ties and results as the serial form in oracle/zstd_model.c.
__device__ int huf_build_lens(const uint32_t *count, uint8_t *lens, uint16_t *order, uint32_t *wt, uint16_t *parent, uint32_t *sh, uint32_t lane) {
    if (lane == 0) { sh[0] = 0; sh[1] = 0; sh[2] = 0; }
    __builtin_amdgcn_wave_barrier();
    // the symbols that occur (text: ~100 of the 256): the rank sort only walks these.  (`parent` is free until the merge: it holds the list.)
    uint16_t *used = parent;
    for (int s = (int)lane; s < 256; s += 64) {
        lens[s] = 0;
        if (count[s]) used[atomicAdd(&sh[0], 1u)] = (uint16_t)s;
    }
    __builtin_amdgcn_wave_barrier();
    const int n = (int)sh[0];
    if (n < 2) return n;
    for (int k = (int)lane; k < n; k += 64) {
        const int s = used[k];
        const uint32_t c = count[s];
        uint32_t rank = 0;
        for (int j = 0; j < n; j++) { const int o = used[j]; const uint32_t co = count[o]; rank += (co < c || (co == c && o < s)) ? 1u : 0u; }   // sort by (count asc, symbol asc)
        order[rank] = (uint16_t)s;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = (int)lane; i < n; i += 64) wt[i] = count[order[i]];
    __builtin_amdgcn_wave_barrier();
    const int nn = 2 * n - 1;
    {
        // two-queue Huffman, leaves win ties; n - 1 dependent steps, executed by the whole wave IN SCALAR REGISTERS (k_deflate.hip's d_build_lens has the
        // same form): every value of the loop is wave-uniform (what comes from LDS through readfirstlane), so compares, selects and counters are SALU work and
        // only the LDS traffic goes through the vector unit.  Both queues keep two heads in scalar registers and a third element in flight in a vector
        // register, so no step waits for an LDS round trip; a new node enters whichever of the internal queue's three places it belongs to directly.
        // Weights are < 2^31, INF marks an exhausted / not yet filled place.  Same picks, same order as
        // `if (lq < n && (iq >= m || wt[lq] <= wt[iq])) leaf else internal`.
        constexpr uint32_t INF = 0xFFFFFFFFu;
        auto rfl = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        int lq = 0, iq = n, m = n;
        uint32_t l0 = rfl(wt[0]), l1 = rfl(wt[1]), i0 = INF, i1 = INF;
        uint32_t l2v = n > 2 ? wt[2] : INF, i2v = INF;
        auto take = [&](uint32_t &w) -> int {
            if (l0 != INF && l0 <= i0) { w = l0; const int a = lq++; l0 = l1; l1 = rfl(l2v); l2v = (lq + 2 < n) ? wt[lq + 2] : INF; return a; }
            w = i0; const int a = iq++; i0 = i1; i1 = rfl(i2v); i2v = (iq + 2 < m) ? wt[iq + 2] : INF; return a;
        };
        while (m < nn) {
            uint32_t wa, wb;
            const int a = take(wa), b = take(wb);
            const uint32_t sum = wa + wb;
            if (lane == 0) { wt[m] = sum; parent[a] = (uint16_t)m; parent[b] = (uint16_t)m; }
            if (iq == m) i0 = sum; else if (iq + 1 == m) i1 = sum; else if (iq + 2 == m) i2v = sum;   // the new node's place among the three heads
            m++;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = (int)lane; i < n; i += 64) {
        int d = 0, q = i;
        while (q != nn - 1) { q = parent[q]; d++; }
        if (d > HUF_MAX) { d = HUF_MAX; sh[1] = 1; }
        lens[order[i]] = (uint8_t)d;
    }
    __builtin_amdgcn_wave_barrier();
    if (sh[1] && lane == 0) {
        int K = 0;
        for (int i = 0; i < n; i++) K += 1 << (HUF_MAX - lens[order[i]]);
        int debt = K - (1 << HUF_MAX);
        while (debt > 0) {
            int pick = -1, bl = 0;
            for (int i = 0; i < n; i++) { int l = lens[order[i]]; if (l < HUF_MAX && l > bl) { bl = l; pick = i; } }
            lens[order[pick]]++; debt -= 1 << (HUF_MAX - 1 - bl);
        }
        while (debt < 0) {
            int pick = -1, bl = 99, slack = -debt;
            for (int i = n - 1; i >= 0; i--) { int l = lens[order[i]]; if (l > 1 && (1 << (HUF_MAX - l)) <= slack && l < bl) { bl = l; pick = i; } }
            if (pick < 0) { sh[2] = 1; break; }
            lens[order[pick]]--; debt += 1 << (HUF_MAX - bl);
        }
    }
    __builtin_amdgcn_wave_barrier();
    return sh[2] ? -1 : n;
}

// Huffman tree description (direct 4-bit weights or FSE-compressed weights) BY A WHOLE WAVE; returns its length (0 = not representable) on every lane.
// On one lane this was the longest serial chain of k_stats (4 KiB entries: 5 of the kernel's 8 ms): the weights' table built cell by cell and ~120 tANS
// steps of two dependent LDS reads each.  Here the lanes compute the weights and count them; lane 0 normalises and writes the table's description
// (<= 13 symbols); fse_build_table_wave builds the table; then the TWO interleaved tANS chains (even / odd weight indices, from the last weight down) run in
// scalar registers -- the symbols' entries and the state table (<= 64 states) sit on the lanes, one value each, and a step is a handful of v_readlane /
// SALU instructions instead of LDS round trips; every step leaves (bits, count) on the lane of its weight; a suffix sum over the counts places them
// and the lanes OR them into the stream.  Same bytes as the serial form (oracle/zstd_model.c huf_write_tree).
__device__ __forceinline__ uint32_t huf_write_tree_wave(uint8_t *dst, const uint8_t *lens, int max_sym_v, int maxbits_v, uint8_t *wts, uint8_t *tmp /* 320 bytes, 4-byte aligned */,
                                        SeqTable *tab, uint8_t *cell, uint16_t *tmp192, uint32_t *sh /* 4 words */, uint32_t lane) {
    // (wave-uniform values are told to be: the chains below then run on the scalar unit -- as vector loops under exec masks they were a third of the kernel)
    const int nw = __builtin_amdgcn_readfirstlane(max_sym_v), maxbits = __builtin_amdgcn_readfirstlane(maxbits_v);
    // (cnt and norm live in LDS, behind the 128 halfwords fse_build_table_wave uses of tmp192)
    uint32_t *cnt = (uint32_t *)(tmp192 + 144);
    int16_t *norm = (int16_t *)(tmp192 + 128);
    uint32_t *tmp32 = (uint32_t *)tmp;
    if (lane < 16) cnt[lane] = 0;
    for (uint32_t i = lane; i < 80; i += 64) tmp32[i] = 0;
    __builtin_amdgcn_wave_barrier();
    uint32_t wv[4];                                             // the weight of symbol 64 r + lane (0 from nw on)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int sy = 64 * r + (int)lane;
        const uint32_t l = sy < nw ? (uint32_t)lens[sy] : 0u;
        wv[r] = l ? (uint32_t)maxbits + 1u - l : 0u;
        if (sy < nw) { wts[sy] = (uint8_t)wv[r]; atomicAdd(&cnt[wv[r]], 1u); }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        int maxw = 0, distinct = 0; uint32_t maxc = 0, kind = 0, hs = 0; int tlog = 0;
        for (int v = 0; v < 16; v++) if (cnt[v]) { maxw = v; distinct++; if (cnt[v] > maxc) maxc = cnt[v]; }
        if (distinct >= 2 && nw >= 2 && maxc > 1) {
            int minlog = 5;
            tlog = (int)hb((uint32_t)(nw - 1)) - 2;
            while ((1 << minlog) < distinct) minlog++;
            if (tlog < minlog) tlog = minlog;
            if (tlog > 6) tlog = 6;
            fse_normalize(cnt, maxw + 1, (uint32_t)nw, tlog, norm);
            hs = fse_write_ncount(tmp + 1, norm, maxw + 1, tlog);
            kind = 1;
        }
        sh[0] = kind; sh[1] = (uint32_t)tlog; sh[2] = hs; sh[3] = (uint32_t)(maxw + 1);
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[0]), tlog = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[1]),
                   hs = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[2]);
    const int nsym = __builtin_amdgcn_readfirstlane((int)sh[3]);
    uint32_t fse_size = 0;
    if (kind) {
        fse_build_table_wave(tab, norm, nsym, (int)tlog, cell, tmp192, lane);
        __builtin_amdgcn_wave_barrier();
        const SeqSym y = tab->sym[(int)lane < nsym ? lane : 0u];
        const uint32_t Ydnb = y.delta_nb, Ydf = (uint32_t)(uint16_t)y.delta_find | ((uint32_t)y.first_state << 16);
        const uint32_t ST = tab->state[lane & ((1u << tlog) - 1u)];
        // The two chains, from the last weight down: odd indices carry state s2, even ones s1 (whatever nw's parity: the serial form's start-up amounts to
        // that); each chain's first symbol only sets its state.  Everything is wave-uniform: the weight, its symbol's entry and the next state come through
        // v_readlane, the bits are shifted into a 64-bit scalar accumulator and leave 32 at a time for the stream (lane 0's store) -- ~25 instructions a
        // step, against ~80 with the bits parked on the weights' lanes and placed by a scan afterwards.
        uint32_t s1 = 0, s2 = 0;
        uint64_t acc = 0; uint32_t nacc = 0, wout = (8u * (1u + hs)) >> 5;                 // (the stream starts at byte 1 + hs of tmp: accumulate from that byte's word on)
        { const uint32_t b0 = (8u * (1u + hs)) & 31u; acc = (uint64_t)(tmp32[wout] & ((1u << b0) - 1u)); nacc = b0; }      // the header bytes sharing the first word
        auto emit = [&](uint32_t v, uint32_t n) {
            acc |= (uint64_t)v << nacc; nacc += n;
            if (nacc >= 32u) { if (lane == 0) tmp32[wout] = (uint32_t)acc; wout++; acc >>= 32; nacc -= 32u; }
        };
        auto wgt = [&](int i) -> uint32_t {                          // (i is uniform: one of the four reads executes)
            if (i < 64) return (uint32_t)__builtin_amdgcn_readlane((int)wv[0], i);
            if (i < 128) return (uint32_t)__builtin_amdgcn_readlane((int)wv[1], i - 64);
            if (i < 192) return (uint32_t)__builtin_amdgcn_readlane((int)wv[2], i - 128);
            return (uint32_t)__builtin_amdgcn_readlane((int)wv[3], i - 192);
        };
        auto step = [&](int i, uint32_t &sc) {
            const uint32_t w = wgt(i);
            const uint32_t dnb = (uint32_t)__builtin_amdgcn_readlane((int)Ydnb, (int)w), df = (uint32_t)__builtin_amdgcn_readlane((int)Ydf, (int)w);
            const uint32_t nb = (sc + dnb) >> 16;
            emit(sc & ((1u << nb) - 1u), nb);
            sc = (uint32_t)__builtin_amdgcn_readlane((int)ST, (int)(sc >> nb) + (int)(int16_t)(df & 0xFFFFu));
        };
        auto first = [&](int i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)Ydf, (int)wgt(i)) >> 16; };
        int i = nw - 1;
        if (i & 1) { s2 = first(i); s1 = first(i - 1); } else { s1 = first(i); s2 = first(i - 1); }
        i -= 2;
        if (i >= 0 && !(i & 1)) { step(i, s1); i--; }              // (down to an odd index: pairs from here)
        for (; i >= 1; i -= 2) { step(i, s2); step(i - 1, s1); }
        const uint32_t m = (1u << tlog) - 1u;
        emit(s2 & m, tlog); emit(s1 & m, tlog); emit(1u, 1u);
        if (lane == 0 && nacc) tmp32[wout] = (uint32_t)acc;
        const uint32_t tbits = (wout << 5) + nacc - 8u * (1u + hs);                         // bits of the stream, the closing one included
        const uint32_t bs = (tbits + 7u) >> 3;
        __builtin_amdgcn_wave_barrier();
        if (hs + bs < 128) { fse_size = hs + bs; if (lane == 0) tmp[0] = (uint8_t)fse_size; }
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t direct = (nw <= 128) ? (uint32_t)(nw + 1) / 2 : 0;
    if (fse_size && (!direct || fse_size < direct)) { for (uint32_t i = lane; i < 1 + fse_size; i += 64) dst[i] = tmp[i]; return 1 + fse_size; }
    if (!direct) return 0;
    if (lane == 0) dst[0] = (uint8_t)(127 + nw);
    for (int i = 2 * (int)lane; i < nw; i += 128) dst[1 + i / 2] = (uint8_t)((wts[i] << 4) | (i + 1 < nw ? wts[i + 1] : 0));
    return 1 + direct;
}

// The three PREDEFINED tables (RFC 8878 3.1.1.3.2.2) are the same for every segment that takes them -- with fewer than 64 sequences, i.e. nearly every small
// entry --: built once per device (launch_default_tables, from pna_gpu_init) and COPIED into a segment's tables (240 words by the wave) instead of built there
// again (10^5 .. 10^6 small entries: three table builds per entry were most of what k_stats did besides the literal code).
__device__ SeqTable g_def_tab[3];
__global__ __launch_bounds__(192)
void k_deftab() {
    __shared__ uint8_t cell[3][256];
    __shared__ __attribute__((aligned(16))) uint16_t tmp192[3][192];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int16_t *def = wave == 0 ? C_LL_DEF : (wave == 1 ? C_OF_DEF : C_ML_DEF);
    const int n = wave == 0 ? 36 : (wave == 1 ? 29 : 53), lg = wave == 1 ? 5 : 6;
    int16_t *norm = (int16_t *)(tmp192[wave] + 128);
    if ((int)lane < n) norm[lane] = def[lane];
    for (uint32_t i = lane; i < sizeof(SeqTable) / 4; i += 64) ((uint32_t *)&g_def_tab[wave])[i] = 0;
    __builtin_amdgcn_wave_barrier();
    fse_build_table_wave(&g_def_tab[wave], norm, n, lg, cell[wave], tmp192[wave], lane);
}
void launch_default_tables(hipStream_t st) { hipLaunchKernelGGL(k_deftab, dim3(1), dim3(192), 0, st); }

// seq_build by a whole wave: lane 0 decides the mode, normalises and writes the description (short loops over <= 53 symbols), the encoder table is
// then built by all lanes (fse_build_table_wave).  sh: four LDS words of the wave.
__device__ bool seq_build_wave(SegTables *T, int which, const uint32_t *count, uint32_t nseq, int alphabet,
                               const int16_t *def, int def_n, int def_log, uint32_t flags, uint8_t *cell, uint16_t *tmp192, uint32_t *sh, uint32_t lane) {
    int16_t *norm = (int16_t *)(tmp192 + 128);
    if (lane == 0) {
        int kind = 0, nsym = 0, tl = 0, ok = 1;                 // kind: 0 no table to build, 1 the predefined distribution, 2 norm
        int maxs = 0, distinct = 0;
        for (int s = 0; s < alphabet; s++) if (count[s]) { maxs = s; distinct++; }
        T->desc_len[which] = 0;
        if (distinct == 1 && nseq > 2) { T->desc[which][0] = (uint8_t)maxs; T->desc_len[which] = 1; T->tlog[which] = 0; T->mode[which] = 1; }
        else {
            const bool def_ok = maxs < def_n;
            if (!(flags & F_FSE) || (nseq < 64 && def_ok)) {
                if (!def_ok) ok = 0;
                else { kind = 1; nsym = def_n; tl = def_log; T->tlog[which] = (uint32_t)def_log; T->mode[which] = 0; }
            } else {
                int tlog = (int)hb(nseq - 1) - 2, minlog = 5;
                while ((1 << minlog) < distinct) minlog++;
                if (tlog < minlog) tlog = minlog;
                if (tlog > (int)SEQ_MAX_LOG) tlog = (int)SEQ_MAX_LOG;
                fse_normalize(count, maxs + 1, nseq, tlog, norm);
                T->desc_len[which] = fse_write_ncount(T->desc[which], norm, maxs + 1, tlog);
                kind = 2; nsym = maxs + 1; tl = tlog; T->tlog[which] = (uint32_t)tlog; T->mode[which] = 2;
            }
        }
        sh[0] = (uint32_t)kind; sh[1] = (uint32_t)nsym; sh[2] = (uint32_t)tl; sh[3] = (uint32_t)ok;
    }
    __builtin_amdgcn_wave_barrier();
    const int kind = (int)sh[0], nsym = (int)sh[1], tl = (int)sh[2];
    if (kind == 1) {                                            // the predefined table: a copy of the device's (which = LL, OF, ML in g_def_tab's order)
        const uint32_t *srcw = (const uint32_t *)&g_def_tab[which];
        uint32_t *dstw = (uint32_t *)&T->tab[which];
        for (uint32_t i = lane; i < sizeof(SeqTable) / 4; i += 64) dstw[i] = srcw[i];
    } else if (kind) fse_build_table_wave(&T->tab[which], norm, nsym, tl, cell, tmp192, lane);
    (void)def;
    return sh[3] != 0;
}

// k_hist: the histograms of k_stats with one workgroup per BLOCK, added into the segment's 448 counters in memory (256 literal bytes, 3 x 64
// codes) -- for batches whose segments are few and cut into many blocks (latency mode) the statistics of a segment are then gathered by up to
// 128 workgroups instead of one; k_stats<true> builds the tables from the counters.
constexpr uint32_t HIST_WORDS = 256 + 3 * 64;
__global__ __launch_bounds__(ST_THREADS)
void k_hist(const uint32_t *__restrict__ blk_seg, const uint64_t *__restrict__ seqs, const uint8_t *__restrict__ lits,
            const BlkInfo *__restrict__ blk, uint32_t *__restrict__ hist, uint16_t *__restrict__ seqw, uint32_t g0, uint32_t blk_log) {
    const uint32_t SC = seq_cap_of(blk_log);
    __shared__ uint32_t h_lit[8][256];
    __shared__ uint32_t h_seq[3][4][64];
    __shared__ uint8_t s_llc[64], s_mlc[128], s_llb[64], s_mlb[128];
    const uint32_t tid = threadIdx.x, g = blockIdx.x + g0;
    const uint32_t nlit = blk[g].nlit, nseq = blk[g].nseq;
    for (uint32_t i = tid; i < 8 * 256; i += ST_THREADS) (&h_lit[0][0])[i] = 0;
    for (uint32_t i = tid; i < 3 * 4 * 64; i += ST_THREADS) (&h_seq[0][0][0])[i] = 0;
    if (tid < 64) { s_llc[tid] = C_LL_CODE[tid]; s_llb[tid] = C_LL_BITS[C_LL_CODE[tid]]; }
    if (tid < 128) { s_mlc[tid] = C_ML_CODE[tid]; s_mlb[tid] = C_ML_BITS[C_ML_CODE[tid]]; }
    __syncthreads();
    const uint8_t *bl = lits + ((size_t)g << blk_log);
    uint32_t *hl = h_lit[tid & 7];
    const uint32_t n16 = nlit >> 4;
    for (uint32_t i = tid; i < n16; i += ST_THREADS) {
        uint4 v = ((const uint4 *)bl)[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            atomicAdd(&hl[w[k] & 0xFF], 1u); atomicAdd(&hl[(w[k] >> 8) & 0xFF], 1u);
            atomicAdd(&hl[(w[k] >> 16) & 0xFF], 1u); atomicAdd(&hl[w[k] >> 24], 1u);
        }
    }
    for (uint32_t j = (n16 << 4) + tid; j < nlit; j += ST_THREADS) atomicAdd(&hl[bl[j]], 1u);
    const uint64_t *bs = seqs + (size_t)g * SC;
    uint16_t *bw = seqw + (size_t)g * SC * 4;                           // three arrays of SC u16: LL, OF, ML
    for (uint32_t i = tid; i < nseq; i += ST_THREADS) {
        const uint64_t s = bs[i];
        const uint32_t llv = seq_ll(s), mb = seq_ml(s) - 3;
        const uint32_t lc = llv < 64 ? (uint32_t)s_llc[llv] : hb(llv) + 19, lx = llv < 64 ? (uint32_t)s_llb[llv] : hb(llv);
        const uint32_t oc = hb(seq_off(s) + 3);
        const uint32_t mc = mb < 128 ? (uint32_t)s_mlc[mb] : hb(mb) + 36, mx = mb < 128 ? (uint32_t)s_mlb[mb] : hb(mb);
        atomicAdd(&h_seq[0][tid & 3][lc], 1u);
        atomicAdd(&h_seq[1][tid & 3][oc], 1u);
        atomicAdd(&h_seq[2][tid & 3][mc], 1u);
        // for the chain kernel (k_seqa): per stream the sequence's code and its number of extra bits, a u16; k_seqa puts its state flushes in their place
        bw[i] = (uint16_t)(lc | (lx << 8)); bw[SC + i] = (uint16_t)(oc | (oc << 8)); bw[2 * SC + i] = (uint16_t)(mc | (mx << 8));
    }
    __syncthreads();
    uint32_t *hs = hist + (size_t)blk_seg[g] * HIST_WORDS;
    { uint32_t c = 0; for (int k = 0; k < 8; k++) c += h_lit[k][tid]; if (c) atomicAdd(&hs[tid], c); }
    if (tid < 192) { const uint32_t w = tid >> 6, s = tid & 63, c = h_seq[w][0][s] + h_seq[w][1][s] + h_seq[w][2][s] + h_seq[w][3][s]; if (c) atomicAdd(&hs[256 + tid], c); }
}

// PRE: the histograms were gathered by k_hist (`hist`, 448 counters per segment); otherwise this workgroup walks the segment's blocks itself
// LEAN (batches of single-block segments, i.e. many small entries): two copies of the literal histogram and one of the code histograms instead of eight and
// four -- 13 KiB of LDS instead of 22, eleven workgroups per CU instead of seven; the kernel's time there is wave 0's chain of dependent steps x the entries in flight.
// PRE = 2 (round 5; large batches behind the split LZ stage): the LITERALS are walked here, the sequence codes were counted by the parse kernel (k_lzp: hist words 256 .. 447)
template <int PRE, bool LEAN>
__global__ __launch_bounds__(ST_THREADS)
void k_stats(const SegDesc *__restrict__ segs, const uint64_t *__restrict__ seqs, const uint8_t *__restrict__ lits,
             const BlkInfo *__restrict__ blk, SegTables *__restrict__ tabs, uint32_t flags, const uint32_t *__restrict__ hist) {
    constexpr uint32_t HL = LEAN ? 2 : 8, HS = LEAN ? 1 : 4;
    __shared__ uint32_t h_lit[HL][256];
    __shared__ uint32_t h_seq[3][HS][64];
    __shared__ uint32_t count[256];
    __shared__ uint32_t scount[3][64];
    __shared__ uint16_t order[256];
    __shared__ uint32_t wt[512];
    __shared__ uint16_t parent[512];
    __shared__ uint8_t  lens[256];
    __shared__ uint8_t  wts[256];
    __shared__ __attribute__((aligned(4))) uint8_t tmp[320];
    __shared__ uint8_t  cell[4][256];            // one scratch set per table-building task (waves 0..3)
    __shared__ __attribute__((aligned(16))) uint16_t tmp192[4][192];
    __shared__ uint32_t hsh[8], wbase_s[HUF_MAX + 2], cntw_s[HUF_MAX + 2], crun_s[HUF_MAX + 2], sq_sh[4][4];
    __shared__ SeqTable wtab;
    __shared__ uint8_t s_llc[64], s_mlc[128];
    const uint32_t tid = threadIdx.x;
    const SegDesc sd = segs[blockIdx.x];
    SegTables *T = tabs + blockIdx.x;
    const uint32_t nblk = seg_nblk(sd);

    for (uint32_t i = tid; i < HL * 256; i += ST_THREADS) (&h_lit[0][0])[i] = 0;
    for (uint32_t i = tid; i < 3 * HS * 64; i += ST_THREADS) (&h_seq[0][0][0])[i] = 0;
    // code look-ups of the small literal / match lengths from LDS (a per-lane index into __constant__ memory is a global load)
    if (tid < 64) s_llc[tid] = C_LL_CODE[tid];
    if (tid < 128) s_mlc[tid] = C_ML_CODE[tid];
    if (tid == 0) {
        T->huf_ok = 0; T->tree_len = 0; T->max_sym = 0; T->maxbits = 0; T->seq_ok = 1;
        for (int k = 0; k < 3; k++) { T->mode[k] = 0; T->tlog[k] = 0; T->desc_len[k] = 0; }
    }
    __syncthreads();
    uint32_t nseq_seg = 0;
    for (uint32_t b = 0; b < (PRE == 1 ? 0u : nblk); b++) {
        const uint32_t g = sd.blk_base + b;
        const uint32_t nlit = blk[g].nlit, nseq = blk[g].nseq;
        nseq_seg += nseq;
        const uint8_t *bl = lits + ((size_t)g << sd.blk_log);
        uint32_t *hl = h_lit[tid & (HL - 1)];
        const uint32_t n16 = nlit >> 4;
        for (uint32_t i = tid; i < n16; i += ST_THREADS) {
            uint4 v = ((const uint4 *)bl)[i];
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&hl[w[k] & 0xFF], 1u); atomicAdd(&hl[(w[k] >> 8) & 0xFF], 1u);
                atomicAdd(&hl[(w[k] >> 16) & 0xFF], 1u); atomicAdd(&hl[w[k] >> 24], 1u);
            }
        }
        for (uint32_t j = (n16 << 4) + tid; j < nlit; j += ST_THREADS) atomicAdd(&hl[bl[j]], 1u);
        const uint64_t *bs = seqs + (size_t)g * seq_cap_of(sd.blk_log);
        for (uint32_t i = tid; i < (PRE ? 0u : nseq); i += ST_THREADS) {
            const uint64_t s = bs[i];
            const uint32_t llv = seq_ll(s), mb = seq_ml(s) - 3;
            atomicAdd(&h_seq[0][tid & (HS - 1)][llv < 64 ? (uint32_t)s_llc[llv] : hb(llv) + 19], 1u);
            atomicAdd(&h_seq[1][tid & (HS - 1)][hb(seq_off(s) + 3)], 1u);
            atomicAdd(&h_seq[2][tid & (HS - 1)][mb < 128 ? (uint32_t)s_mlc[mb] : hb(mb) + 36], 1u);
        }
    }
    __syncthreads();
    if (PRE == 1) {
        const uint32_t *hs = hist + (size_t)blockIdx.x * HIST_WORDS;
        count[tid] = hs[tid];
        if (tid < 192) scount[tid >> 6][tid & 63] = hs[256 + tid];
        __syncthreads();
        for (uint32_t s = 0; s < 36; s++) nseq_seg += scount[0][s];             // every sequence has one literal-length code
    } else if (PRE == 2) {
        const uint32_t *hs = hist + (size_t)blockIdx.x * HIST_WORDS;
        { uint32_t c = 0; for (uint32_t k = 0; k < HL; k++) c += h_lit[k][tid]; count[tid] = c; }
        if (tid < 192) scount[tid >> 6][tid & 63] = hs[256 + tid];
    } else {
        { uint32_t c = 0; for (uint32_t k = 0; k < HL; k++) c += h_lit[k][tid]; count[tid] = c; }
        if (tid < 192) { uint32_t w = tid >> 6, s = tid & 63, c = 0; for (uint32_t k = 0; k < HS; k++) c += h_seq[w][k][s]; scount[w][s] = c; }
    }
    __syncthreads();
    // ---- tables: wave 0 builds the literal code, lane 0 of waves 1..3 one sequence table each (LL, OF, ML), concurrently
    const uint32_t wave = tid >> 6, lane = tid & 63;
    if (wave != 0) {
        if (!nseq_seg) return;
        bool ok;
        if (wave == 1) ok = seq_build_wave(T, 0, scount[0], nseq_seg, 36, C_LL_DEF, 36, 6, flags, cell[1], tmp192[1], sq_sh[1], lane);
        else if (wave == 2) ok = seq_build_wave(T, 1, scount[1], nseq_seg, 32, C_OF_DEF, 29, 5, flags, cell[2], tmp192[2], sq_sh[2], lane);
        else ok = seq_build_wave(T, 2, scount[2], nseq_seg, 53, C_ML_DEF, 53, 6, flags, cell[3], tmp192[3], sq_sh[3], lane);
        if (!ok && lane == 0) atomicAnd(&T->seq_ok, 0u);
        return;
    }
    if (!(flags & F_HUF)) return;
    const int np = huf_build_lens(count, lens, order, wt, parent, hsh, lane);
    if (np < 2) return;
    // canonical codes: weight-1 symbols first (ascending symbol), code = cell index >> (weight-1); symbol s sits behind all
    // symbols of smaller weight and the lower-numbered ones of its own weight
    // (all of this on the wave's lanes: the highest symbol and the longest code by a reduction, the symbols per weight by LDS atomics, a symbol's rank
    // among the lower-numbered symbols of its length by one ballot per length and round -- each was a loop on one lane before)
    uint32_t ms = 0, mb = 0;
    for (int s2 = (int)lane; s2 < 256; s2 += 64) { if (count[s2]) ms = (uint32_t)s2; const uint32_t l = lens[s2]; mb = l > mb ? l : mb; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t a2 = (uint32_t)__shfl_xor((int)ms, o), b2 = (uint32_t)__shfl_xor((int)mb, o); ms = a2 > ms ? a2 : ms; mb = b2 > mb ? b2 : mb; }
    const int max_sym = (int)ms, maxbits = (int)mb;
    if (lane < HUF_MAX + 2) { cntw_s[lane] = 0; crun_s[lane] = 0; }
    __builtin_amdgcn_wave_barrier();
    for (int s2 = (int)lane; s2 < 256; s2 += 64) { const uint32_t l = lens[s2]; if (l && s2 <= max_sym) atomicAdd(&cntw_s[(uint32_t)maxbits + 1 - l], 1u); }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        uint32_t pos = 0;
        for (int w = 1; w <= maxbits; w++) { wbase_s[w] = pos; pos += cntw_s[w] << (w - 1); }
        hsh[3] = (uint32_t)max_sym; hsh[4] = (uint32_t)maxbits;
    }
    __builtin_amdgcn_wave_barrier();
    const uint64_t lane_lt = ((uint64_t)1 << lane) - 1;
    for (int r = 0; r < 4; r++) {
        const int s2 = 64 * r + (int)lane;
        const uint32_t l = (s2 <= max_sym) ? (uint32_t)lens[s2] : 0u;
        uint32_t before = 0;
        for (uint32_t L = 1; L <= (uint32_t)maxbits; L++) {
            const uint64_t m = __ballot(l == L);
            if (l == L) before = crun_s[L] + (uint32_t)__popcll(m & lane_lt);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0 && m) crun_s[L] += (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t v = 0;
        if (l) {
            const uint32_t w = (uint32_t)maxbits + 1 - l;
            v = ((wbase_s[w] + (before << (w - 1))) >> (w - 1)) | (l << 16);
        }
        T->huf_code[s2] = v;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t tl = huf_write_tree_wave(T->tree, lens, max_sym, maxbits, wts, tmp, &wtab, cell[0], tmp192[0], sq_sh[0], lane);
    if (lane == 0) { T->tree_len = tl; T->max_sym = (uint32_t)max_sym; T->maxbits = (uint32_t)maxbits; T->huf_ok = tl > 0; }
}

// ------------------------------------------------------------------ k_lit
// One workgroup per block, one wave per Huffman stream.  Both passes read the literals with coalesced 16-byte loads
// (lane i <- granule base+i): pass 1 sums the code lengths of each stream, pass 2 walks each stream from its END in
// rounds of 64 granules (later symbols sit at lower bit positions), places the lanes' 16-symbol pieces with a wave
// suffix scan, ORs them into a per-wave LDS staging buffer (64-bit LDS atomics) and flushes completed qwords to the
// zeroed body with coalesced 64-bit atomic ORs (neighbouring streams share their boundary qwords).
constexpr uint32_t LIT_THREADS = 256;
constexpr uint32_t LIT_STAGE_Q = 184;            // qwords of staging per wave: 64 lanes x 176 bits + carry

__global__ __launch_bounds__(LIT_THREADS)
void k_lit(const SegDesc *__restrict__ segs, const uint32_t *__restrict__ blk_seg, const uint8_t *__restrict__ lits,
           BlkInfo *__restrict__ blk, const SegTables *__restrict__ tabs, uint8_t *__restrict__ litc, uint32_t flags, uint32_t g0, uint32_t blk_log) {
    __shared__ uint32_t code[256];
    __shared__ uint32_t sbits[4];
    __shared__ unsigned long long stage[4][LIT_STAGE_Q];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = blockIdx.x + g0;                          // g0: first block of the chunk of segments this launch covers
    const SegTables *T = tabs + blk_seg[g];
    const uint32_t nlit = blk[g].nlit;
    const uint8_t *bl = lits + ((size_t)g << blk_log);
    const uint4 *bl16 = (const uint4 *)bl;
    unsigned long long *out64 = (unsigned long long *)(litc + ((size_t)g << blk_log));
    if (!(flags & F_HUF) || nlit < 64) { if (tid == 0) { blk[g].lit_body = 0; blk[g].lit_rle = 0; } return; }
    if (!T->huf_ok) {
        // no Huffman code for this segment: only the RLE test is left (coalesced)
        const uint32_t b0 = bl[0] * 0x01010101u; int same = 1;
        const uint32_t n16 = nlit >> 4;
        for (uint32_t i = tid; i < n16; i += LIT_THREADS) { const uint4 v = bl16[i]; same &= (v.x == b0) & (v.y == b0) & (v.z == b0) & (v.w == b0); }
        for (uint32_t i = (n16 << 4) + tid; i < nlit; i += LIT_THREADS) same &= (bl[i] == (uint8_t)b0);
        same = __syncthreads_and(same);
        if (tid == 0) { blk[g].lit_rle = same ? 1u : 0u; blk[g].lit_body = 0; }
        return;
    }
    code[tid] = T->huf_code[tid];
    __syncthreads();
    const uint32_t nstreams = nlit >= 256 ? 4u : 1u;
    const uint32_t segsz = nstreams == 4 ? (nlit + 3) / 4 : nlit;
    // stream `wave`: symbols [a, a+m)
    const uint32_t a = wave * segsz;
    uint32_t m = 0;
    if (wave < nstreams) m = (nstreams == 4 && wave == 3) ? nlit - 3 * segsz : segsz;
    const uint32_t e = a + m;
    const uint32_t gfirst = a >> 4, gend = m ? (e + 15) >> 4 : gfirst;
    // ---- pass 1: bits of the stream, and the RLE test (all literals equal) in the same read: the waves' ranges cover every literal
    {
        const uint32_t b0 = bl[0];
        uint32_t bits = 0; int same = 1;
        for (uint32_t gi = gfirst + lane; gi < gend; gi += 64) {
            const uint4 v = bl16[gi];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w}, base = gi << 4;
#pragma unroll
            for (uint32_t k = 0; k < 16; k++) {
                const uint32_t i = base + k, sym = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
                if (i >= a && i < e) { bits += code[sym] >> 16; same &= (sym == b0); }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) bits += (uint32_t)__shfl_xor((int)bits, d);
        if (lane == 0) sbits[wave] = wave < nstreams ? bits : 0;
        same = __syncthreads_and(same);
        if (same) { if (tid == 0) { blk[g].lit_rle = 1; blk[g].lit_body = 0; } return; }
    }
    // stream byte sizes and offsets
    uint32_t sz[4], off[4], body = nstreams == 4 ? 6u : 0u;
    for (uint32_t k = 0; k < 4; k++) { sz[k] = k < nstreams ? (sbits[k] >> 3) + 1 : 0; off[k] = body; body += sz[k]; }
    if (tid == 0) { blk[g].lit_body = body; blk[g].lit_rle = 0; }
    if (body >= nlit || body > (1u << blk_log)) return;                   // Huffman cannot win: k_plan picks raw literals
    for (uint32_t i = tid; i < (body + 7) / 8; i += LIT_THREADS) out64[i] = 0;
    for (uint32_t i = lane; i < LIT_STAGE_Q; i += 64) stage[wave][i] = 0;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (wave < nstreams) {
        unsigned long long *st = stage[wave];
        uint32_t pos = off[wave] * 8;                              // next free bit of the body (global bit address)
        uint32_t qfl = pos >> 6;                                   // st[0] holds body qword qfl
        const uint32_t nround = (gend - gfirst + 63) / 64;
        for (uint32_t r = 0; r < nround; r++) {
            const int32_t gi = (int32_t)gend - 64 * (int32_t)(r + 1) + (int32_t)lane;
            const bool act = gi >= (int32_t)gfirst;
            unsigned long long gv[4] = {0, 0, 0, 0}; uint32_t gl[4] = {0, 0, 0, 0};
            if (act) {
                const uint4 v = bl16[gi];
                const uint32_t w[4] = {v.x, v.y, v.z, v.w}, base = (uint32_t)gi << 4;
#pragma unroll
                for (int k = 15; k >= 0; k--) {                    // later symbols first (lower bit positions)
                    const uint32_t i = base + (uint32_t)k;
                    if (i >= a && i < e) {
                        const uint32_t cv = code[(w[k >> 2] >> (8 * (k & 3))) & 0xFF];
                        const int grp = 3 - (k >> 2);
                        gv[grp] |= (unsigned long long)(cv & 0xFFFF) << gl[grp]; gl[grp] += cv >> 16;
                    }
                }
            }
            const uint32_t ltot = gl[0] + gl[1] + gl[2] + gl[3];
            uint32_t sc = ltot;                                    // inclusive prefix over lanes
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { uint32_t t = (uint32_t)__shfl_up((int)sc, d); if (lane >= (uint32_t)d) sc += t; }
            const uint32_t rtot = (uint32_t)__shfl((int)sc, 63);
            uint32_t p = pos + (rtot - sc) - (qfl << 6);           // staging bit of this lane's first group (lane 63 lowest)
#pragma unroll
            for (int grp = 0; grp < 4; grp++) {
                if (gl[grp]) {
                    const uint32_t qi = p >> 6, sh = p & 63;
                    atomicOr(&st[qi], gv[grp] << sh);
                    if (sh + gl[grp] > 64) atomicOr(&st[qi + 1], gv[grp] >> (64 - sh));
                    p += gl[grp];
                }
            }
            pos += rtot;
            // flush completed qwords, keep the partial one at st[0]
            const uint32_t nq = (pos >> 6) - qfl;
            const unsigned long long part = st[nq];
            for (uint32_t j = lane; j < nq; j += 64) { const unsigned long long x = st[j]; if (x) atomicOr(&out64[qfl + j], x); }
            for (uint32_t j = lane; j <= nq; j += 64) st[j] = (j == 0) ? part : 0ull;
            qfl += nq;
        }
        // closing 1-bit, then flush what is left (at most two qwords)
        if (lane == 0) atomicOr(&st[(pos - (qfl << 6)) >> 6], 1ull << ((pos - (qfl << 6)) & 63));
        pos += 1;
        const uint32_t nq = ((pos - (qfl << 6)) + 63) >> 6;
        for (uint32_t j = lane; j < nq; j += 64) { const unsigned long long x = st[j]; if (x) atomicOr(&out64[qfl + j], x); }
    }
    if (tid == 0 && nstreams == 4) atomicOr(&out64[0], (unsigned long long)sz[0] | ((unsigned long long)sz[1] << 16) | ((unsigned long long)sz[2] << 32));
}

// ------------------------------------------------------------------ k_seqa / k_seqb : sequence bitstreams in two phases
// The FSE states of a block form serial chains (each sequence's state depends on the next one's), but only the STATES do: the extra
// bits of the literal length / match length / offset and the position of every field follow from prefix sums, and the three streams'
// states do not depend on each other.  So the chain kernel k_seqa (three lanes per block, one per stream) walks the sequences last to
// first and records per sequence and stream just the state flush (<= 8 bits + their count, a u16); k_seqb (one workgroup per block)
// then assembles the bitstream token-parallel: 256 sequences per round, field lengths -> workgroup scan -> 64-bit ORs into an LDS
// stage -> coalesced stores.  History: one lane per block with the bit packing inside the chain ~350 instructions per sequence (k_seq,
// still the form for the largest batches: it touches the sequences once), states only ~95, one stream per lane with the codes prepared
// by k_hist ~20 -- and the chain's length times its instructions is what bounds this stage, for one entry as for ten thousand.
constexpr uint32_t SEQ_SEGS_PER_WG = 64 / BLK_PER_SEG;   // k_seq: 8 segments x 8 blocks = 64 lanes (blocks of BLK_SIZE; in latency mode a segment has 16 .. 128 smaller
                                                          // blocks, a wave then carries 4, 2, 1 or half a segment: bps_log = log2(blocks per full segment))
static_assert(SEQ_MAX_LOG <= 8, "k_seqa: a state flush is at most 8 bits (12 are kept), the final states go into 3 x 8 bits");

// k_seqa: THREE lanes per block, one per FSE stream (literal lengths, offsets, match lengths: their state chains are independent of each other), 21
// blocks per wave.  A lane's step per sequence is what is serial about the stream -- nb = (state + delta_nb) >> 16, the flushed bits, the next state
// from the table in LDS -- and nothing else: the code and the count of extra bits of every sequence were put down by k_hist (token-parallel), and the
// lane writes its flush (bits | count << 12, a u16) in their place for k_seqb.  A wave issues one instruction every few cycles whatever its lanes do, so
// the time of this kernel is (instructions per step) x (sequences per block): ~20 x 1 000 in latency mode, where it was ~95 x 1 000 with one lane
// doing all three streams and the code look-ups.
constexpr uint32_t SEQA_BLKS = 21, SEQA_MAXSEG = 5;         // blocks per wave; segments those can span (>= 8 block slots per segment: 21 slots touch at most 4)
__global__ __launch_bounds__(64)
void k_seqa(const SegDesc *__restrict__ segs, uint32_t nseg, BlkInfo *__restrict__ blk, const SegTables *__restrict__ tabs,
            uint16_t *__restrict__ seqw, uint32_t bps_log) {
    __shared__ SeqTable tab[SEQA_MAXSEG][3];
    __shared__ SeqTable ztab;                           // all zero: what an RLE-mode table amounts to (0 bits per step, the state stays 0)
    const uint32_t lane = threadIdx.x;
    const uint32_t vb0 = blockIdx.x * SEQA_BLKS;
    const uint32_t seg0 = vb0 >> bps_log, seg1 = (vb0 + SEQA_BLKS - 1) >> bps_log;
    for (uint32_t i = lane; i < sizeof(SeqTable) / 4; i += 64) ((uint32_t *)&ztab)[i] = 0;
    for (uint32_t s = seg0; s <= seg1 && s < nseg && s - seg0 < SEQA_MAXSEG; s++) {
        const uint32_t *srcw = (const uint32_t *)&tabs[s].tab[0];
        uint32_t *dstw = (uint32_t *)&tab[s - seg0][0];
        for (uint32_t i = lane; i < 3 * sizeof(SeqTable) / 4; i += 64) dstw[i] = srcw[i];
    }
    __syncthreads();
    const uint32_t slot = lane / 3, st = lane - 3 * slot;              // st: 0 literal lengths, 1 offsets, 2 match lengths (the order of SegTables::tab)
    const uint32_t vb = vb0 + slot;
    const uint32_t sidx = vb >> bps_log, b = vb & ((1u << bps_log) - 1);
    if (lane >= 3 * SEQA_BLKS || sidx >= nseg) return;
    const SegDesc sd = segs[sidx];
    if (b >= seg_nblk(sd)) return;
    const uint32_t g = sd.blk_base + b;
    const SegTables *T = tabs + sidx;
    const uint32_t nseq = blk[g].nseq;
    if (nseq == 0 || !T->seq_ok) { if (st == 0) blk[g].seq_bits = 0; return; }
    const uint32_t mode = T->mode[st], tlog = T->tlog[st];
    const SeqTable *tb = mode == 1 ? &ztab : &tab[sidx - seg0][st];
    const uint32_t SC = seq_cap_of(sd.blk_log);
    uint16_t *w = seqw + (size_t)g * SC * 4 + (size_t)st * SC;   // this stream's u16 per sequence (k_hist: code | extra bits << 8)
    uint32_t state = 0, bits = 0;
    // Sequences last to first in groups of eight (one 16-byte load, one 16-byte store); the next group is requested before the current one is walked.
    // The block's last sequence only sets the initial state.
    uint4 *w8 = (uint4 *)w;
    int32_t j = (int32_t)((nseq - 1) >> 3);
    // (four groups in flight: a group's eight steps are ~0.3 us of the lane's chain, a load under this kernel's traffic takes ~2 us -- with one group
    // ahead the kernel waited for memory 80 % of its cycles, PMC round 4)
    uint4 cur = w8[j], r1 = cur, r2 = cur, r3 = cur;
    if (j > 0) r1 = w8[j - 1];
    if (j > 1) r2 = w8[j - 2];
    if (j > 2) r3 = w8[j - 3];
    bool first = true;
    for (; j >= 0; j--) {
        uint4 r4 = r3;
        if (j > 3) r4 = w8[j - 4];
        const uint32_t cw32[4] = {cur.x, cur.y, cur.z, cur.w};
        uint32_t out[4] = {0, 0, 0, 0};
        const int32_t top = (int32_t)(nseq - 1) - 8 * j;                // highest sequence of the group that exists (7 but in the top group)
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            if (k > top) continue;
            const uint32_t cw = (cw32[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            const SeqSym y = tb->sym[cw & 0xFF];
            uint32_t rec = 0;
            if (first) { state = y.first_state; first = false; }
            else {
                const uint32_t nb = (state + y.delta_nb) >> 16;
                rec = (state & ((1u << nb) - 1)) | (nb << 12);
                state = tb->state[(int)(state >> nb) + y.delta_find];
                bits += nb;
            }
            bits += cw >> 8;
            out[k >> 1] |= rec << (16 * (k & 1));
        }
        w8[j] = make_uint4(out[0], out[1], out[2], out[3]);
        cur = r1; r1 = r2; r2 = r3; r3 = r4;
    }
    // the block's bit count: the three streams' sums + the final states + the closing bit; the final states go to BlkInfo::pad (zero so far), ML | OF << 8 | LL << 16
    bits += mode != 1 ? tlog : 0u;
    const uint32_t b1 = (uint32_t)__shfl((int)bits, (int)(3 * slot + 1)), b2 = (uint32_t)__shfl((int)bits, (int)(3 * slot + 2));
    atomicOr(&blk[g].pad, (state & ((1u << tlog) - 1)) << (st == 2 ? 0 : (st == 1 ? 8 : 16)));
    if (st == 0) blk[g].seq_bits = (bits + b1 + b2 + 1 + 7) >> 3;
}

// The one-kernel form: the same chain with the bit packing inside it (~350 instructions per sequence instead of ~95).  Slower per
// chain, but it touches the sequences once and writes nothing in between; it wins when the launch has more chain waves than the
// chip has SIMDs, where k_seqa's time is set by 4-cycle VALU issue on the SIMDs that hold two waves and k_seqb's by its 12 bytes of
// traffic per sequence (10 000 x 1 MiB: 6.4 ms against 4.7 + 2.4 ms; 1 GiB sub-batches and single entries: 3.4 ms against 1.9 + 0.2 ms).
// GT (batches whose segments hold ONE block each: entries of at most a block -- the many-small-files case): a lane per SEGMENT (bps_log = 0) with the tables
// read where k_stats left them, in global memory.  With the block slots of a full segment per segment, a 4 KiB entry kept one lane of two waves busy, and
// both waves copied its three tables to LDS first: 26 ns per entry, the stage's largest cost; its chain is short, and 64 of them per wave hide the loads.
template <bool GT>
__global__ __launch_bounds__(64)
void k_seq(const SegDesc *__restrict__ segs, uint32_t nseg, const uint64_t *__restrict__ seqs, BlkInfo *__restrict__ blk,
           const SegTables *__restrict__ tabs, uint8_t *__restrict__ seqc, uint32_t bps_log) {
    __shared__ SeqTable tab[GT ? 1 : SEQ_SEGS_PER_WG][3];
    __shared__ SeqTable ztab;                           // all zero: what an RLE-mode table amounts to
    __shared__ uint32_t lut_ll[64], lut_ml[128];       // code | extra bits << 8 | base << 16 (small values only)
    __shared__ uint32_t ring[8][64];                   // per lane (column) the bitstream's last dwords: see put()
    const uint32_t lane = threadIdx.x;
    const uint32_t seg0 = (blockIdx.x * 64u) >> bps_log;
    const uint32_t segs_wg = bps_log >= 6 ? 1u : 64u >> bps_log;
    for (uint32_t i = lane; i < sizeof(SeqTable) / 4; i += 64) ((uint32_t *)&ztab)[i] = 0;
    { uint32_t c = C_LL_CODE[lane]; lut_ll[lane] = c | ((uint32_t)C_LL_BITS[c] << 8) | (C_LL_BASE[c] << 16); }
    for (uint32_t i = lane; i < 128; i += 64) { uint32_t c = C_ML_CODE[i]; lut_ml[i] = c | ((uint32_t)C_ML_BITS[c] << 8) | (C_ML_BASE[c] << 16); }
    for (uint32_t s = 0; !GT && s < segs_wg && seg0 + s < nseg; s++) {
        const uint32_t *srcw = (const uint32_t *)&tabs[seg0 + s].tab[0];
        uint32_t *dstw = (uint32_t *)&tab[s][0];
        for (uint32_t i = lane; i < 3 * sizeof(SeqTable) / 4; i += 64) dstw[i] = srcw[i];
    }
    __syncthreads();
    const uint32_t vb = blockIdx.x * 64u + lane;                       // lane -> (segment, block): block slot vb of the launch, 1 << bps_log slots per segment
    const uint32_t sidx = vb >> bps_log, sl = sidx - seg0, b = vb & ((1u << bps_log) - 1);
    if (sidx >= nseg) return;
    const SegDesc sd = segs[sidx];
    const uint32_t nblk = seg_nblk(sd);
    if (b >= nblk) return;
    const uint32_t g = sd.blk_base + b;
    const SegTables *T = tabs + sidx;
    const uint32_t nseq = blk[g].nseq;
    if (nseq == 0 || !T->seq_ok) { blk[g].seq_bits = 0; return; }
    const uint32_t mll = T->mode[0], mof = T->mode[1], mml = T->mode[2];
    const uint32_t tl_ll = T->tlog[0], tl_of = T->tlog[1], tl_ml = T->tlog[2];
    // RLE mode (one symbol, no state bits) runs through the same code on an all-zero table: delta_nb = 0 gives 0 bits, state[0] = 0
    // keeps the state at 0 -- no per-sequence branch on the mode
    const SeqTable *tll = mll == 1 ? &ztab : (GT ? &T->tab[0] : &tab[sl][0]), *tof = mof == 1 ? &ztab : (GT ? &T->tab[1] : &tab[sl][1]),
                   *tml = mml == 1 ? &ztab : (GT ? &T->tab[2] : &tab[sl][2]);
    const uint64_t *bs = seqs + (size_t)g * seq_cap_of(sd.blk_log);
    uint32_t *out32 = (uint32_t *)(seqc + ((size_t)g << sd.blk_log));
    const uint32_t cap_words = (1u << sd.blk_log) / 4;
    uint32_t acc = 0, nb = 0, widx = 0, gdone = 0;
    // Finished dwords leave four at a time as ONE 16-byte store: every lane writes a stream of its own, and a 4-byte store per lane reached HBM as a
    // masked 32-byte write each (WRITE_SIZE 12.5 GB per step for 1.5 GB of bitstreams -- the kernel's time was that traffic).  The group is collected
    // in the lane's column of an LDS ring of eight dwords: put() writes the accumulator's low dword to slot widx & 7 WHETHER OR NOT it is complete (a
    // later put overwrites it with more bits; the slots ahead belong to the group that left before), so a put is a shift, an OR, a store and three
    // selects -- with the group held in registers it was twenty instructions, a third of the kernel's, and the kernel is bound by issue (PMC: VALU
    // active half of the wave cycles at one wave per SIMD).  A sequence appends < 3 dwords, so one look per sequence finds a finished group.
    auto put = [&](uint32_t v, uint32_t n) {               // n <= 32, v < 2^n (the accumulator holds < 32 bits before)
        const uint64_t y = (uint64_t)acc | ((uint64_t)v << nb);
        ring[widx & 7u][lane] = (uint32_t)y;
        nb += n;
        const bool full = nb >= 32;
        widx += full ? 1u : 0u; acc = full ? (uint32_t)(y >> 32) : (uint32_t)y; nb &= 31u;
    };
    auto flush = [&]() {
        if ((widx >> 2) != gdone) {                      // group gdone is complete (cap_words is a multiple of 4: a group lies inside or outside as a whole)
            const uint32_t h = (gdone & 1u) * 4u;
            if (4 * gdone < cap_words) *(uint4 *)(out32 + 4 * gdone) = make_uint4(ring[h][lane], ring[h + 1][lane], ring[h + 2][lane], ring[h + 3][lane]);
            gdone++;
        }
    };
    // code / extra-bit count / base of a literal length and a match length: LDS LUT for small values, arithmetic above (code = highbit +
    // 19 / 36), both computed and selected (no divergent branch)
    auto codes = [&](uint64_t s, uint32_t &llv, uint32_t &mlv, uint32_t &ofb, uint32_t &lc, uint32_t &lbits, uint32_t &lbase,
                     uint32_t &mc, uint32_t &mbits, uint32_t &mbase, uint32_t &oc) {
        llv = seq_ll(s); mlv = seq_ml(s); ofb = seq_off(s) + 3;
        const uint32_t tl = lut_ll[llv < 64 ? llv : 63u], hl = hb(llv | 1u);
        const bool ls = llv < 64;
        lc = ls ? (tl & 0xFF) : hl + 19; lbits = ls ? ((tl >> 8) & 0xFF) : hl; lbase = ls ? (tl >> 16) : (1u << hl);
        const uint32_t mb = mlv - 3;
        const uint32_t tm = lut_ml[mb < 128 ? mb : 127u], hm = hb(mb | 1u);
        const bool ms = mb < 128;
        mc = ms ? (tm & 0xFF) : hm + 36; mbits = ms ? ((tm >> 8) & 0xFF) : hm; mbase = ms ? (tm >> 16) : ((1u << hm) + 3);
        oc = hb(ofb);
    };
    // Sequences are consumed last-to-first in 32-byte chunks (4 sequences, two 16-byte loads per lane); the next
    // chunk is requested before the current one is encoded so the HBM/L2 latency overlaps the serial tANS chain.
    // Round 5: what a sequence needs BESIDES the states -- its codes, extra bits, the three symbols' table entries (two dependent LDS round trips) -- is
    // prepared for the four sequences of a chunk at once, ahead of the chain; on the chain itself a step is the three state look-ups (one round trip) and the
    // puts.  (Until then every sequence walked LUT -> symbol entry -> state one after the other: ~1 400 cycles per sequence at one wave per SIMD, three
    // quarters of them LDS latency.)  The block's last chunk (partial; its last sequence only sets the states) is peeled off the loop.
    uint32_t st_ml = 0, st_of = 0, st_ll = 0;
    const uint32_t top = nseq - 1;
    uint32_t k = top >> 2;
    const uint4 *bs4 = (const uint4 *)bs;
    struct Prep { uint32_t x_lm, n_lm, x_of, n_of; SeqSym yo, ym, yl; };
    auto prep = [&](uint64_t sq, Prep &P) {
        uint32_t llv, mlv, ofb, lc, lbits, lbase, mc, mbits, mbase, oc;
        codes(sq, llv, mlv, ofb, lc, lbits, lbase, mc, mbits, mbase, oc);
        P.yo = tof->sym[oc]; P.ym = tml->sym[mc]; P.yl = tll->sym[lc];
        P.x_lm = (llv - lbase) | ((mlv - mbase) << lbits); P.n_lm = lbits + mbits;     // literal-length and match-length extra bits (<= 16 + 16) as one field
        P.x_of = ofb - (1u << oc); P.n_of = oc;                                        // then the offset's
    };
    auto step = [&](const Prep &P) {
        // the three state flushes (<= 9 bits each) go out as one field
        const uint32_t no = (st_of + P.yo.delta_nb) >> 16, nm = (st_ml + P.ym.delta_nb) >> 16, nl = (st_ll + P.yl.delta_nb) >> 16;
        const uint32_t fv = (st_of & ((1u << no) - 1)) | ((st_ml & ((1u << nm) - 1)) << no) | ((st_ll & ((1u << nl) - 1)) << (no + nm));
        st_of = tof->state[(int)(st_of >> no) + P.yo.delta_find];
        st_ml = tml->state[(int)(st_ml >> nm) + P.ym.delta_find];
        st_ll = tll->state[(int)(st_ll >> nl) + P.yl.delta_find];
        put(fv, no + nm + nl);
        put(P.x_lm, P.n_lm);
        put(P.x_of, P.n_of);
        flush();
    };
    // (two chunks ahead: a chunk's four sequences are ~800 instructions = 1.6 us of one wave, about what a load takes under the stage's traffic -- with one
    // chunk in flight a third of the kernel's cycles were waits for it)
    uint4 a0 = bs4[2 * k], a1 = bs4[2 * k + 1];
    uint4 n0 = a0, n1 = a1, m0 = a0, m1 = a1;
    if (k > 0) { n0 = bs4[2 * (k - 1)]; n1 = bs4[2 * (k - 1) + 1]; }
    if (k > 1) { m0 = bs4[2 * (k - 2)]; m1 = bs4[2 * (k - 2) + 1]; }
    {   // the last chunk: sequences top & 3 .. 0 of it; the block's last sequence only sets the initial states
        const uint64_t sq[4] = {(uint64_t)a0.x | ((uint64_t)a0.y << 32), (uint64_t)a0.z | ((uint64_t)a0.w << 32),
                                (uint64_t)a1.x | ((uint64_t)a1.y << 32), (uint64_t)a1.z | ((uint64_t)a1.w << 32)};
        const uint32_t jt = top & 3u;
        Prep P;
        prep(jt == 3 ? sq[3] : (jt == 2 ? sq[2] : (jt == 1 ? sq[1] : sq[0])), P);
        st_ml = P.ym.first_state; st_of = P.yo.first_state; st_ll = P.yl.first_state;
        put(P.x_lm, P.n_lm); put(P.x_of, P.n_of); flush();
        for (uint32_t j = jt; j-- > 0;) { prep(j == 2 ? sq[2] : (j == 1 ? sq[1] : sq[0]), P); step(P); }
    }
    while (k > 0) {
        k--; a0 = n0; a1 = n1; n0 = m0; n1 = m1;
        if (k > 1) { m0 = bs4[2 * (k - 2)]; m1 = bs4[2 * (k - 2) + 1]; }
        const uint64_t sq[4] = {(uint64_t)a0.x | ((uint64_t)a0.y << 32), (uint64_t)a0.z | ((uint64_t)a0.w << 32),
                                (uint64_t)a1.x | ((uint64_t)a1.y << 32), (uint64_t)a1.z | ((uint64_t)a1.w << 32)};
        Prep P[4];
#pragma unroll
        for (int j = 0; j < 4; j++) prep(sq[j], P[j]);
#pragma unroll
        for (int j = 3; j >= 0; j--) step(P[j]);
    }
    if (mml != 1) put(st_ml & ((1u << tl_ml) - 1), tl_ml);
    if (mof != 1) put(st_of & ((1u << tl_of) - 1), tl_of);
    if (mll != 1) put(st_ll & ((1u << tl_ll) - 1), tl_ll);
    put(1, 1);
    flush();
    uint32_t bytes = widx * 4 + (nb + 7) / 8;
    {   // the dwords of the last, incomplete group, then the accumulator's rest
        const uint32_t k = widx & 3u, gb = widx - k, h = (gdone & 1u) * 4u;
        if (gb < cap_words) {
            if (k > 0) out32[gb] = ring[h][lane];
            if (k > 1) out32[gb + 1] = ring[h + 1][lane];
            if (k > 2) out32[gb + 2] = ring[h + 2][lane];
            if (nb) out32[widx] = acc;
        }
    }
    blk[g].seq_bits = bytes;
}

constexpr uint32_t SB_THREADS = 256;
constexpr uint32_t SB_STAGE_Q = 320;                    // 256 sequences x <= 76 bits = 304 qwords, + the carried partial one

__global__ __launch_bounds__(SB_THREADS)
void k_seqb(const uint32_t *__restrict__ blk_seg, const uint64_t *__restrict__ seqs, const uint16_t *__restrict__ seqw,
            const BlkInfo *__restrict__ blk, const SegTables *__restrict__ tabs, uint8_t *__restrict__ seqc, uint32_t g0, uint32_t blk_log) {
    const uint32_t SC = seq_cap_of(blk_log);
    __shared__ unsigned long long st[SB_STAGE_Q];
    __shared__ uint32_t wtot[SB_THREADS / 64];
    __shared__ uint32_t lut_ll[64], lut_ml[128];       // extra bits | base << 8 (small values only)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t g = blockIdx.x + g0;
    const BlkInfo bi = blk[g];
    const SegTables *T = tabs + blk_seg[g];
    const uint32_t n = bi.nseq;
    if (n == 0 || !T->seq_ok || bi.seq_bits >= (1u << blk_log)) return;       // nothing to encode / cannot beat a raw block (k_plan)
    if (tid < 64) { const uint32_t c = C_LL_CODE[tid]; lut_ll[tid] = (uint32_t)C_LL_BITS[c] | (C_LL_BASE[c] << 8); }
    if (tid < 128) { const uint32_t c = C_ML_CODE[tid]; lut_ml[tid] = (uint32_t)C_ML_BITS[c] | (C_ML_BASE[c] << 8); }
    for (uint32_t i = tid; i < SB_STAGE_Q; i += SB_THREADS) st[i] = 0;
    __syncthreads();
    unsigned long long *out64 = (unsigned long long *)(seqc + ((size_t)g << blk_log));
    const uint64_t *bs = seqs + (size_t)g * SC;
    const uint16_t *w = seqw + (size_t)g * SC * 4;              // k_seqa's flushes (bits | count << 12): arrays LL, OF, ML
    uint32_t pos = 0, qfl = 0;                                          // next free bit of the stream; st[0] holds stream qword qfl
    for (uint32_t hi = n; hi > 0; hi = hi > SB_THREADS ? hi - SB_THREADS : 0u) {
        // thread t takes sequence hi - 1 - t: later sequences lie at lower bit positions
        unsigned long long lo64 = 0; uint32_t nlo = 0, oval = 0, oc = 0;
        if (tid < hi) {
            const uint32_t i = hi - 1 - tid;
            const uint64_t s = bs[i];
            const uint32_t rl = w[i], ro = w[SC + i], rm = w[2 * SC + i];
            const uint32_t llv = seq_ll(s), mlv = seq_ml(s), mb = mlv - 3, ofb = seq_off(s) + 3;
            const uint32_t tl = lut_ll[llv < 64 ? llv : 63u], hl = hb(llv | 1u);
            const uint32_t lbits = llv < 64 ? (tl & 0xFF) : hl, lbase = llv < 64 ? (tl >> 8) : (1u << hl);
            const uint32_t tm = lut_ml[mb < 128 ? mb : 127u], hm = hb(mb | 1u);
            const uint32_t mbits = mb < 128 ? (tm & 0xFF) : hm, mbase = mb < 128 ? (tm >> 8) : ((1u << hm) + 3);
            oc = hb(ofb); oval = ofb - (1u << oc);
            const uint32_t no = ro >> 12, nm = rm >> 12, nl = rl >> 12, nst = no + nm + nl;
            // the three state flushes as one field: offset state lowest, then match length, then literal length
            const uint32_t fv = (ro & 0xFFFu) | ((rm & 0xFFFu) << no) | ((rl & 0xFFFu) << (no + nm));
            lo64 = (unsigned long long)fv | ((unsigned long long)(llv - lbase) << nst) | ((unsigned long long)(mlv - mbase) << (nst + lbits));
            nlo = nst + lbits + mbits;                                  // <= 24 + 16 + 16
        }
        const uint32_t tot = nlo + oc;
        uint32_t incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl += y; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        uint32_t base = 0, ctot = 0;
#pragma unroll
        for (uint32_t k = 0; k < SB_THREADS / 64; k++) { const uint32_t x = wtot[k]; ctot += x; if (k < wave) base += x; }
        uint32_t p = pos + base + incl - tot - (qfl << 6);              // stage-relative bit of this sequence's first field
        if (nlo) {
            const uint32_t qi = p >> 6, sh = p & 63;
            atomicOr(&st[qi], lo64 << sh);
            if (sh + nlo > 64) atomicOr(&st[qi + 1], lo64 >> (64 - sh));
        }
        p += nlo;
        if (oc) {
            const uint32_t qi = p >> 6, sh = p & 63;
            atomicOr(&st[qi], (unsigned long long)oval << sh);
            if (sh + oc > 64) atomicOr(&st[qi + 1], (unsigned long long)oval >> (64 - sh));
        }
        pos += ctot;
        __syncthreads();
        // flush the completed qwords, carry the partial one to st[0]
        const uint32_t nq = (pos >> 6) - qfl;
        const unsigned long long part = st[nq];
        for (uint32_t j = tid; j < nq; j += SB_THREADS) out64[qfl + j] = st[j];
        __syncthreads();
        for (uint32_t j = tid; j <= nq; j += SB_THREADS) st[j] = (j == 0) ? part : 0ull;
        qfl += nq;
        __syncthreads();
    }
    if (tid == 0) {
        // final states (match length, offset, literal length; none for an RLE-mode table), then the closing 1-bit
        unsigned long long acc = st[0]; uint32_t nb = pos - (qfl << 6);
        auto put = [&](uint32_t v, uint32_t k) {                        // k <= 8, nb < 64
            acc |= (unsigned long long)v << nb;
            if (nb + k >= 64) { out64[qfl++] = acc; acc = (nb + k > 64) ? (unsigned long long)v >> (64 - nb) : 0ull; nb = nb + k - 64; }
            else nb += k;
        };
        if (T->mode[2] != 1) put(bi.pad & 0xFF, T->tlog[2]);
        if (T->mode[1] != 1) put((bi.pad >> 8) & 0xFF, T->tlog[1]);
        if (T->mode[0] != 1) put((bi.pad >> 16) & 0xFF, T->tlog[0]);
        put(1, 1);
        if (nb) out64[qfl] = acc;
    }
}

// ------------------------------------------------------------------ k_plan
__device__ __forceinline__ uint32_t raw_lit_hdr(uint32_t nlit) { return nlit < 32 ? 1u : (nlit < 4096 ? 2u : 3u); }
__device__ __forceinline__ uint32_t nseq_hdr(uint32_t nseq) { return nseq < 128 ? 1u : (nseq < 0x7F00 ? 2u : 3u); }

// One WAVE per segment, a lane per block (64 at a time).  Which block carries the segment's Huffman tree / FSE table descriptions is a serial rule --
// the first block that uses them --, but the state (tree seen, tables seen) changes at most twice along a segment: every lane evaluates its block
// under the current state, the first lane whose block changes the state settles everything up to itself, and the rest is evaluated again; sizes by a
// wave scan.  (One thread per segment walked up to 128 blocks with dependent loads: 40 us for a single segment in latency mode.)
constexpr uint32_t PLAN_THREADS = 256;
__global__ __launch_bounds__(PLAN_THREADS)
void k_plan(const SegDesc *__restrict__ segs, uint32_t nseg, BlkInfo *__restrict__ blk,
            const SegTables *__restrict__ tabs, uint64_t *__restrict__ seg_size, uint32_t flags) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t sidx = blockIdx.x * (PLAN_THREADS / 64) + (threadIdx.x >> 6);
    if (sidx >= nseg) return;
    const SegDesc sd = segs[sidx];
    const SegTables *T = tabs + sidx;
    const uint32_t nblk = seg_nblk(sd), bsz = 1u << sd.blk_log;
    if (sd.len == 0) { if (lane == 0) seg_size[sidx] = 9; return; }     // empty entry: the reference's 9-byte empty frame
    const uint32_t desc_total = T->desc_len[0] + T->desc_len[1] + T->desc_len[2];
    const uint32_t seq_ok = T->seq_ok, huf_ok = T->huf_ok, tree_len = T->tree_len;
    bool have_huf = false, have_seq = false;                            // (uniform)
    uint64_t off = ((sd.first & 4) && !(sd.first & 1)) ? 0 : 6;          // (single-frame entries: the frame header stands in front of the entry's first segment only)
    for (uint32_t c0 = 0; c0 < nblk; c0 += 64) {
        const uint32_t b = c0 + lane, n = nblk - c0 < 64 ? nblk - c0 : 64u;
        const bool in = lane < n;
        const uint32_t g = sd.blk_base + (in ? b : c0);
        const uint32_t b0 = (in ? b : c0) * bsz, bl_len = sd.len - b0 < bsz ? sd.len - b0 : bsz;
        const uint32_t nlit = blk[g].nlit, nseq = blk[g].nseq, lit_rle = blk[g].lit_rle, lit_body = blk[g].lit_body, seq_bits = blk[g].seq_bits;
        uint32_t plan = 0, csz = 0;
        uint32_t pos = 0;                                               // lanes below it are settled
        for (;;) {
            const bool ok = seq_ok || nseq == 0;
            uint32_t p = 0, cs = 0;
            if (ok) {
                const uint32_t raw_h = raw_lit_hdr(nlit);
                const bool rle = (flags & F_HUF) && nlit >= 64 && lit_rle;
                if (rle) { cs = raw_h + 1; p |= 16; }
                else {
                    const uint32_t hs = (huf_ok && nlit >= 64) ? lit_body : 0;
                    const uint32_t lh = 3 + (nlit >= 1024) + (nlit >= 16384), ts = have_huf ? 0 : tree_len;
                    if (hs && lh + ts + hs < raw_h + nlit) { cs = lh + ts + hs; p |= 2; if (!have_huf) p |= 4; }
                    else cs = raw_h + nlit;
                }
                cs += nseq_hdr(nseq);
                if (nseq) { cs += 1 + (have_seq ? 0 : desc_total) + seq_bits; if (!have_seq) p |= 8; }
            }
            bool comp = true;
            if (!ok || cs >= bl_len) { p = 0; cs = bl_len; comp = false; } else p |= 1;
            const bool s_huf = comp && (p & 2) && !have_huf, s_seq = comp && nseq && !have_seq;
            const uint64_t m = __ballot((s_huf || s_seq) && in && lane >= pos);
            const uint32_t f = m ? (uint32_t)__builtin_ctzll(m) : 63u;
            if (lane >= pos && lane <= f) { plan = p; csz = cs; }
            if (!m) break;
            have_huf = have_huf || ((__ballot(s_huf) >> f) & 1);
            have_seq = have_seq || ((__ballot(s_seq) >> f) & 1);
            pos = f + 1;
            if (pos >= n) break;
        }
        const uint32_t sz = in ? 3 + csz : 0u;
        uint32_t incl = sz;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl += t; }
        if (in) { blk[g].plan = plan; blk[g].out_size = sz; blk[g].out_off = off + (incl - sz); }
        off += (uint32_t)__shfl((int)incl, 63);
    }
    if (lane == 0) seg_size[sidx] = off;
}

// ------------------------------------------------------------------ k_scan : exclusive scan (single workgroup)
__global__ __launch_bounds__(1024)
void k_scan(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t n) {
    __shared__ uint64_t part[2][1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t a = tid * per, e = a + per < n ? a + per : n;
    uint64_t s = 0;
    for (uint32_t i = a; i < e; i++) s += in[i];
    // inclusive scan of the 1 024 partial sums in LDS (log steps, two buffers)
    uint32_t cur = 0;
    part[0][tid] = s;
    __syncthreads();
    const uint32_t nact = per ? (n + per - 1) / per : 0;                // threads that hold anything
    for (uint32_t d = 1; d < 1024 && d < nact; d <<= 1) {
        const uint64_t v = part[cur][tid] + (tid >= d ? part[cur][tid - d] : 0ull);
        part[cur ^ 1][tid] = v;
        cur ^= 1;
        __syncthreads();
    }
    uint64_t r = part[cur][tid] - s;
    if (tid == 0) out[n] = nact ? part[cur][nact - 1] : 0ull;           // (threads from nact on hold nothing and were not scanned to the end)
    for (uint32_t i = a; i < e; i++) { out[i] = r; r += in[i]; }
}

// ------------------------------------------------------------------ k_write : one workgroup per block
constexpr uint32_t WR_THREADS = 256;
// n bytes by the workgroup's T threads: the destination's unaligned head and tail byte by byte, its aligned middle as 16-byte stores of unaligned 16-byte loads
// (byte by byte a wave moved 64 bytes per instruction; the streams of a 128 KiB block are 10 .. 40 KiB)
typedef uint32_t wr_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void copy_bytes(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t tid, uint32_t T) {
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u), h = head < n ? head : n, mid = (n - h) >> 4;
    if (tid < h) dst[tid] = src[tid];
    for (uint32_t i = tid; i < mid; i += T) { wr_v4u v; __builtin_memcpy(&v, src + h + 16 * i, 16); *(wr_v4u *)(dst + h + 16 * i) = v; }
    for (uint32_t i = h + 16 * mid + tid; i < n; i += T) dst[i] = src[i];
}

__global__ __launch_bounds__(WR_THREADS)
void k_write(const uint8_t *__restrict__ src, const SegDesc *__restrict__ segs, const uint32_t *__restrict__ blk_seg,
             const BlkInfo *__restrict__ blk, const SegTables *__restrict__ tabs, const uint64_t *__restrict__ seg_off,
             const uint8_t *__restrict__ lits, const uint8_t *__restrict__ litc, const uint8_t *__restrict__ seqc,
             uint8_t *__restrict__ dst) {
    const uint32_t tid = threadIdx.x, g = blockIdx.x, NT = blockDim.x;   // 256 threads, or 64 for batches of many small blocks (a wave per block: four times the blocks in flight)
    const uint32_t sidx = blk_seg[g];
    const SegDesc sd = segs[sidx];
    const SegTables *T = tabs + sidx;
    const BlkInfo bi = blk[g];
    const uint32_t b = g - sd.blk_base;
    const uint32_t nblk = seg_nblk(sd), bsz = 1u << sd.blk_log;
    const uint32_t b0 = b * bsz, bl_len = sd.len - b0 < bsz ? sd.len - b0 : bsz;
    const bool single = (sd.first & 4) != 0;                            // the entry is ONE frame: header at its first segment, last-block bit at its last block
    const uint32_t last = (b + 1 == nblk && (!single || (sd.first & 2))) ? 1u : 0u;
    uint8_t *fr = dst + seg_off[sidx];
    uint8_t *out = fr + bi.out_off;
    if (b == 0 && tid < 6 && (!single || (sd.first & 1))) { const uint8_t fh[6] = {0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x50}; fr[tid] = fh[tid]; }
    const uint32_t csz = bi.out_size - 3;
    if (tid < 3) { uint32_t hdr = last | ((bi.plan & 1 ? 2u : 0u) << 1) | (csz << 3); out[tid] = (uint8_t)(hdr >> (8 * tid)); }
    out += 3;
    if (!(bi.plan & 1)) { copy_bytes(out, src + sd.src_off + b0, bl_len, tid, NT); return; }
    const uint32_t nlit = bi.nlit, nseq = bi.nseq;
    const uint8_t *bl = lits + ((size_t)g << sd.blk_log);
    uint32_t pos = 0;
    // literals section
    if (bi.plan & 16) {
        uint32_t h = raw_lit_hdr(nlit);
        if (tid == 0) {
            if (h == 1) out[0] = (uint8_t)(1 | (nlit << 3));
            else if (h == 2) { out[0] = (uint8_t)(1 | (1 << 2) | ((nlit & 15) << 4)); out[1] = (uint8_t)(nlit >> 4); }
            else { out[0] = (uint8_t)(1 | (3 << 2) | ((nlit & 15) << 4)); out[1] = (uint8_t)(nlit >> 4); out[2] = (uint8_t)(nlit >> 12); }
            out[h] = bl[0];
        }
        pos = h + 1;
    } else if (bi.plan & 2) {
        const uint32_t lh = 3 + (nlit >= 1024) + (nlit >= 16384), ts = (bi.plan & 4) ? T->tree_len : 0, hs = bi.lit_body;
        if (tid == 0) {
            uint64_t type = (bi.plan & 4) ? 2u : 3u, comp = ts + hs, h;
            if (lh == 3) h = type | ((uint64_t)(nlit >= 256 ? 1 : 0) << 2) | ((uint64_t)nlit << 4) | (comp << 14);
            else if (lh == 4) h = type | (2u << 2) | ((uint64_t)nlit << 4) | (comp << 18);
            else h = type | (3u << 2) | ((uint64_t)nlit << 4) | (comp << 22);
            for (uint32_t i = 0; i < lh; i++) out[i] = (uint8_t)(h >> (8 * i));
        }
        copy_bytes(out + lh, T->tree, ts, tid, NT);
        copy_bytes(out + lh + ts, litc + ((size_t)g << sd.blk_log), hs, tid, NT);
        pos = lh + ts + hs;
    } else {
        uint32_t h = raw_lit_hdr(nlit);
        if (tid == 0) {
            if (h == 1) out[0] = (uint8_t)(nlit << 3);
            else if (h == 2) { out[0] = (uint8_t)((1 << 2) | ((nlit & 15) << 4)); out[1] = (uint8_t)(nlit >> 4); }
            else { out[0] = (uint8_t)((3 << 2) | ((nlit & 15) << 4)); out[1] = (uint8_t)(nlit >> 4); out[2] = (uint8_t)(nlit >> 12); }
        }
        copy_bytes(out + h, bl, nlit, tid, NT);
        pos = h + nlit;
    }
    // sequences section
    const uint32_t nh = nseq_hdr(nseq);
    if (tid == 0) {
        if (nh == 1) out[pos] = (uint8_t)nseq;
        else if (nh == 2) { out[pos] = (uint8_t)((nseq >> 8) + 128); out[pos + 1] = (uint8_t)nseq; }
        else { out[pos] = 255; out[pos + 1] = (uint8_t)(nseq - 0x7F00); out[pos + 2] = (uint8_t)((nseq - 0x7F00) >> 8); }
    }
    pos += nh;
    if (nseq) {
        const bool carry = (bi.plan & 8) != 0;
        if (tid == 0) {
            uint32_t m[3];
            for (int k = 0; k < 3; k++) m[k] = T->mode[k] == 0 ? 0u : (carry ? T->mode[k] : 3u);
            out[pos] = (uint8_t)((m[0] << 6) | (m[1] << 4) | (m[2] << 2));
        }
        pos += 1;
        if (carry) for (int k = 0; k < 3; k++) { copy_bytes(out + pos, T->desc[k], T->desc_len[k], tid, NT); pos += T->desc_len[k]; }
        copy_bytes(out + pos, seqc + ((size_t)g << sd.blk_log), bi.seq_bits, tid, NT);
    }
}

// empty entries: 28 B5 2F FD 20 00 01 00 00 (what the reference emits, tests/golden/zstd.pna raw/empty.txt)
__global__ void k_empty(const SegDesc *__restrict__ segs, uint32_t nseg, const uint64_t *__restrict__ seg_off, uint8_t *__restrict__ dst) {
    const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nseg || segs[sidx].len != 0) return;
    const uint8_t e[9] = {0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00};
    uint8_t *o = dst + seg_off[sidx];
    for (int i = 0; i < 9; i++) o[i] = e[i];
}

// ------------------------------------------------------------------ launchers
// large n (10^5 .. 10^6 small entries): chunks of 4 096 values scanned by a workgroup each, the chunk sums by k_scan, then added back
__global__ __launch_bounds__(1024)
void k_scan_a(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t n, uint64_t *__restrict__ chunk_sum) {
    __shared__ uint64_t part[2][1024];
    const uint32_t tid = threadIdx.x, i0 = blockIdx.x * 4096 + tid * 4;
    uint64_t v[4], s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { v[k] = i0 + k < n ? in[i0 + k] : 0ull; s += v[k]; }
    uint32_t cur = 0;
    part[0][tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        part[cur ^ 1][tid] = part[cur][tid] + (tid >= d ? part[cur][tid - d] : 0ull);
        cur ^= 1;
        __syncthreads();
    }
    uint64_t r = part[cur][tid] - s;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { if (i0 + k < n) out[i0 + k] = r; r += v[k]; }
    if (tid == 1023) chunk_sum[blockIdx.x] = part[cur][1023];
}
__global__ __launch_bounds__(1024)
void k_scan_c(uint64_t *__restrict__ out, uint32_t n, const uint64_t *__restrict__ chunk_off, uint32_t nchunk) {
    const uint32_t i0 = blockIdx.x * 4096 + threadIdx.x * 4;
    const uint64_t o = chunk_off[blockIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) if (i0 + k < n) out[i0 + k] += o;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = chunk_off[nchunk];
}
// (scratch of the hierarchical form: the caller's `out` must have room for n + 1 + 2 * (n / 4096 + 2) values)
void k_scan_launch(const uint64_t *in, uint64_t *out, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, in, out, n);
}
void k_scan_launch_big(const uint64_t *in, uint64_t *out, uint32_t n, hipStream_t st) {
    if (n <= 65536) { k_scan_launch(in, out, n, st); return; }
    const uint32_t nchunk = (n + 4095) / 4096;
    uint64_t *chunk_sum = out + n + 1, *chunk_off = chunk_sum + nchunk + 1;
    hipLaunchKernelGGL(k_scan_a, dim3(nchunk), dim3(1024), 0, st, in, out, n, chunk_sum);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, chunk_sum, chunk_off, nchunk);
    hipLaunchKernelGGL(k_scan_c, dim3(nchunk), dim3(1024), 0, st, out, n, chunk_off, nchunk);
}

// Entropy stage of the segments [s0, s0 + ns) whose blocks are [g0, g0 + nb): statistics + tables, literal streams,
// sequence streams.  `tabs`, `segs` are the arrays of the whole batch.
void launch_entropy_chunk(const SegDesc *segs, uint32_t s0, uint32_t ns, const uint32_t *blk_seg, uint32_t g0, uint32_t nb,
                          const uint64_t *seqs, const uint8_t *lits, BlkInfo *blk, SegTables *tabs, uint8_t *litc, uint8_t *seqc, uint32_t *seqw,
                          uint32_t flags, uint32_t blk_log, uint32_t *hist, hipStream_t st, hipEvent_t *ev /* 3 events: after stats, lit, seq; may be null */,
                          hipStream_t side, hipEvent_t fork, hipEvent_t join, bool single_block /* no segment of the chunk holds more than one block */, const uint32_t *seq_hist) {
    // side != null (large batches, the one-kernel sequence coder): the literal coder runs on a second stream NEXT TO the sequence coder -- both only
    // need the tables; k_seq is a few long chains per SIMD (1 250 waves of 64 chains on 1 024 SIMDs: issue slots to spare), k_lit streams memory
    const uint32_t bps_log = single_block ? 0u : 20u - blk_log;             // block slots per segment: the blocks of a full one (SEG_SIZE = 1 MiB), or the one block every segment has
    const uint32_t seq_wgs = (uint32_t)((((uint64_t)ns << bps_log) + 63) / 64);
    if (hist) {                                                             // histograms per block, tables from the counters (the caller zeroed them)
        if (nb) hipLaunchKernelGGL(k_hist, dim3(nb), dim3(ST_THREADS), 0, st, blk_seg, seqs, lits, blk, hist, (uint16_t *)seqw, g0, blk_log);
        hipLaunchKernelGGL((k_stats<1, false>), dim3(ns), dim3(ST_THREADS), 0, st, segs + s0, seqs, lits, blk, tabs + s0, flags, hist + (size_t)s0 * HIST_WORDS);
    } else if (single_block)
        hipLaunchKernelGGL((k_stats<0, true>), dim3(ns), dim3(ST_THREADS), 0, st, segs + s0, seqs, lits, blk, tabs + s0, flags, (const uint32_t *)nullptr);
    else if (seq_hist)                                                        // the parse kernel counted the sequence codes (seq_hist: the segments' counters)
        hipLaunchKernelGGL((k_stats<2, false>), dim3(ns), dim3(ST_THREADS), 0, st, segs + s0, seqs, lits, blk, tabs + s0, flags, seq_hist + (size_t)s0 * HIST_WORDS);
    else
        hipLaunchKernelGGL((k_stats<0, false>), dim3(ns), dim3(ST_THREADS), 0, st, segs + s0, seqs, lits, blk, tabs + s0, flags, (const uint32_t *)nullptr);
    if (ev) (void)hipEventRecord(ev[0], st);
    const bool forked = side && !hist && nb;
    if (forked) {
        // The sequence coder goes FIRST: its 1 250 long-running waves must hold their SIMD slots before the literal coder's 80 000 workgroups flood the
        // dispatcher (with k_lit ahead -- even by the few microseconds of an event marker between the two launches -- the pair took 11 ms instead of 7).
        // The "literals" interval of the timing is empty then, the "sequences" interval covers both kernels.
        if (ev) (void)hipEventRecord(ev[1], st);
        (void)hipEventRecord(fork, st);
        if (single_block) hipLaunchKernelGGL(k_seq<true>, dim3(seq_wgs), dim3(64), 0, st, segs + s0, ns, seqs, blk, tabs + s0, seqc, bps_log);
        else hipLaunchKernelGGL(k_seq<false>, dim3(seq_wgs), dim3(64), 0, st, segs + s0, ns, seqs, blk, tabs + s0, seqc, bps_log);
        (void)hipStreamWaitEvent(side, fork, 0);
        hipLaunchKernelGGL(k_lit, dim3(nb), dim3(LIT_THREADS), 0, side, segs, blk_seg, lits, blk, tabs, litc, flags, g0, blk_log);
        (void)hipEventRecord(join, side);
        (void)hipStreamWaitEvent(st, join, 0);
        if (ev) (void)hipEventRecord(ev[2], st);
        return;
    }
    if (nb) hipLaunchKernelGGL(k_lit, dim3(nb), dim3(LIT_THREADS), 0, st, segs, blk_seg, lits, blk, tabs, litc, flags, g0, blk_log);
    if (ev) (void)hipEventRecord(ev[1], st);
    // two phases (short chains on three lanes per block, parallel packing) for the batches whose statistics were gathered per block (the host
    // picks them: few enough blocks that the chain waves fit the SIMDs), the one-kernel form otherwise (see k_seq)
    if (hist) {
        const uint32_t wgs = (uint32_t)((((uint64_t)ns << bps_log) + SEQA_BLKS - 1) / SEQA_BLKS);
        hipLaunchKernelGGL(k_seqa, dim3(wgs), dim3(64), 0, st, segs + s0, ns, blk, tabs + s0, (uint16_t *)seqw, bps_log);
        if (nb) hipLaunchKernelGGL(k_seqb, dim3(nb), dim3(SB_THREADS), 0, st, blk_seg, seqs, (const uint16_t *)seqw, blk, tabs, seqc, g0, blk_log);
    } else if (single_block) {
        hipLaunchKernelGGL(k_seq<true>, dim3(seq_wgs), dim3(64), 0, st, segs + s0, ns, seqs, blk, tabs + s0, seqc, bps_log);
    } else {
        hipLaunchKernelGGL(k_seq<false>, dim3(seq_wgs), dim3(64), 0, st, segs + s0, ns, seqs, blk, tabs + s0, seqc, bps_log);
    }
    if (ev) (void)hipEventRecord(ev[2], st);
}
// sizes of all segments -> offsets
void launch_plan(const SegDesc *segs, uint32_t nseg, BlkInfo *blk, const SegTables *tabs, uint64_t *seg_size, uint64_t *seg_off,
                 uint32_t flags, hipStream_t st) {
    hipLaunchKernelGGL(k_plan, dim3((nseg + PLAN_THREADS / 64 - 1) / (PLAN_THREADS / 64)), dim3(PLAN_THREADS), 0, st, segs, nseg, blk, tabs, seg_size, flags);
    k_scan_launch_big(seg_size, seg_off, nseg, st);
}
void launch_write(const uint8_t *src, const SegDesc *segs, uint32_t nseg, const uint32_t *blk_seg, uint32_t nblk, const BlkInfo *blk,
                  const SegTables *tabs, const uint64_t *seg_off, const uint8_t *lits, const uint8_t *litc,
                  const uint8_t *seqc, uint8_t *dst, bool any_empty, hipStream_t st, bool small_blocks) {
    if (any_empty) hipLaunchKernelGGL(k_empty, dim3((nseg + 255) / 256), dim3(256), 0, st, segs, nseg, seg_off, dst);
    if (nblk) hipLaunchKernelGGL(k_write, dim3(nblk), dim3(small_blocks ? 64 : WR_THREADS), 0, st, src, segs, blk_seg, blk, tabs, seg_off, lits, litc, seqc, dst);
}

} // namespace pna
