/*
 * oracle/zstd_model.c -- TEST INFRASTRUCTURE ONLY.  See zstd_model.h.
 *
 * The algorithm (every rule below is normative for the HIP kernels):
 *
 * ENTRY -> FRAMES   an entry of n bytes is cut into 1 MiB segments; each segment is one zstd frame
 *                   (magic, FHD 0x00, window descriptor 0x50 = 1 MiB, no content size, no checksum, like the
 *                   reference's streaming frames `28 B5 2F FD 00 58` except for the smaller window).  n == 0
 *                   gives the reference's empty frame 28B52FFD 20 00 01 00 00 (tests/golden/zstd.pna,
 *                   raw/empty.txt).  Frames are concatenated (zstd-rs Decoder reads all of them).
 * SEGMENT -> BLOCKS 128 KiB blocks; the hash table persists across the blocks of a segment, so matches reach
 *                   back into earlier blocks (<= max_off bytes).
 * LZ STAGE          tile-synchronous hash matching, see pna_lz_block().
 * ENTROPY STAGE     literals: raw / RLE / Huffman (<= 11 bits, 1 or 4 streams); sequences: predefined / RLE /
 *                   FSE_Compressed tables; raw-block fallback when a block does not shrink.  Repeat-offset codes
 *                   use only history established inside the same block (see encode_sequences).
 */
#include "zstd_model.h"
#include <string.h>
#include <stdlib.h>

static int hb32(uint32_t v) { int r = -1; while (v) { v >>= 1; r++; } return r; }

void pna_zstd_default_params(pna_zstd_params *p) {
    p->hash_log = 14; p->min_match = 6; p->tile = 2048; p->max_off = 61440; p->cap1 = 32;
    p->lookahead = 1024; p->flags = PNA_F_HUF | PNA_F_FSE | PNA_F_LAZY | PNA_F_REP;
}

size_t pna_zstd_bound(size_t n) {
    size_t segs = (n + PNA_SEG_SIZE - 1) / PNA_SEG_SIZE; if (segs == 0) segs = 1;
    size_t blks = (n + PNA_BLK_SIZE - 1) / PNA_BLK_SIZE + segs;
    return n + segs * 6 + blks * 3 + 16;
}

/* ======================================================================== LZ stage */

static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

/* hash of min_match (4..6) bytes at p; bytes past `avail` read as 0 (never happens for hashed positions) */
static uint32_t lz_hash(const uint8_t *p, uint32_t min_match, uint32_t hash_log) {
    uint32_t lo = rd32(p);
    uint32_t hi = 0;
    if (min_match >= 5) hi = p[4];
    if (min_match >= 6) hi |= (uint32_t)p[5] << 8;
    uint32_t h = lo * 0x9E3779B1u + hi * 0x85EBCA6Bu;
    return h >> (32 - hash_log);
}

/*
 * One 128 KiB block.  For each tile of p->tile positions, in this order:
 *   L  every position q with q + 8 <= seg_len looks up cand[q] = table[hash(q)] (value = position+1, 0 = empty);
 *   I  every such position stores table[hash(q)] = max(old, q+1)   (ascending q here == atomic max on the GPU);
 *   M  len[q] = length of the common prefix of seg[q..] and seg[c..] (c = cand-1), capped to cap1 and to the
 *      block end; a candidate is usable iff cand != 0 and q - c <= max_off; len < min_match counts as 0;
 *   P  greedy parse in ascending q from `next_free`: position q starts a match iff len[q] >= min_match and not
 *      (LAZY and (q & 63) != 63 and q+1 < tile end and len[q+1] > len[q]); a chosen match whose len == cap1 is
 *      extended byte-wise up to min(block end, tile end + lookahead); the parse then continues at q + len.
 * Literals are the bytes not covered by matches, in order; the block's last literals follow the last sequence.
 */
uint32_t pna_lz_block(const uint8_t *seg, uint32_t seg_len, uint32_t blk_start, uint32_t blk_len,
                      uint32_t *table, const pna_zstd_params *p,
                      pna_seq *seqs, uint8_t *lits, uint32_t *nlit_out) {
    uint32_t blk_end = blk_start + blk_len;
    uint32_t nseq = 0, nlit = 0;
    uint32_t next_free = blk_start, lit_start = blk_start;
    uint32_t T = p->tile;
    uint32_t *cand = (uint32_t *)malloc(sizeof(uint32_t) * T);
    uint16_t *len = (uint16_t *)malloc(sizeof(uint16_t) * (T + 1));
    for (uint32_t t0 = blk_start; t0 < blk_end; t0 += T) {
        uint32_t t1 = t0 + T < blk_end ? t0 + T : blk_end;
        /* L */
        for (uint32_t q = t0; q < t1; q++)
            cand[q - t0] = (q + 8 <= seg_len) ? table[lz_hash(seg + q, p->min_match, p->hash_log)] : 0;
        /* I */
        for (uint32_t q = t0; q < t1; q++)
            if (q + 8 <= seg_len) {
                uint32_t h = lz_hash(seg + q, p->min_match, p->hash_log);
                if (table[h] < q + 1) table[h] = q + 1;
            }
        /* M */
        for (uint32_t q = t0; q < t1; q++) {
            uint32_t c1 = cand[q - t0], l = 0;
            if (c1 != 0 && q - (c1 - 1) <= p->max_off) {
                uint32_t c = c1 - 1, lim = blk_end - q;
                if (lim > p->cap1) lim = p->cap1;
                while (l < lim && seg[q + l] == seg[c + l]) l++;
                if (l < p->min_match) l = 0;
            }
            len[q - t0] = (uint16_t)l;
        }
        len[t1 - t0] = 0;
        /* P */
        uint32_t ext_lim = t1 + p->lookahead < blk_end ? t1 + p->lookahead : blk_end;
        for (uint32_t q = (next_free > t0 ? next_free : t0); q < t1; ) {
            uint32_t l = len[q - t0];
            int take = l >= p->min_match;
            if (take && (p->flags & PNA_F_LAZY) && (q & 63) != 63 && q + 1 < t1 && len[q + 1 - t0] > l) take = 0;
            if (!take) { q++; continue; }
            uint32_t c = cand[q - t0] - 1;
            if (l == p->cap1) while (q + l < ext_lim && seg[q + l] == seg[c + l]) l++;
            seqs[nseq].ll = q - lit_start; seqs[nseq].ml = l; seqs[nseq].off = q - c; nseq++;
            memcpy(lits + nlit, seg + lit_start, q - lit_start); nlit += q - lit_start;
            q += l; lit_start = q; next_free = q;
        }
    }
    memcpy(lits + nlit, seg + lit_start, blk_end - lit_start); nlit += blk_end - lit_start;
    free(cand); free(len);
    *nlit_out = nlit;
    return nseq;
}

/* ======================================================================== bit writer (forward, LSB first) */

typedef struct { uint8_t *p; size_t pos; uint64_t acc; int nb; } bitw;
static void bw_init(bitw *w, uint8_t *p) { w->p = p; w->pos = 0; w->acc = 0; w->nb = 0; }
static void bw_add(bitw *w, uint32_t v, int n) {
    if (n == 0) return;
    w->acc |= (uint64_t)(v & ((n == 32) ? 0xFFFFFFFFu : ((1u << n) - 1))) << w->nb; w->nb += n;
    while (w->nb >= 8) { w->p[w->pos++] = (uint8_t)w->acc; w->acc >>= 8; w->nb -= 8; }
}
static size_t bw_close_marker(bitw *w) { /* closing 1-bit then pad to a byte */
    bw_add(w, 1, 1);
    if (w->nb > 0) { w->p[w->pos++] = (uint8_t)w->acc; w->acc = 0; w->nb = 0; }
    return w->pos;
}
static size_t bw_flush_plain(bitw *w) { /* pad to a byte without marker (FSE table descriptions) */
    if (w->nb > 0) { w->p[w->pos++] = (uint8_t)w->acc; w->acc = 0; w->nb = 0; }
    return w->pos;
}

/* ======================================================================== FSE encoding tables */

typedef struct {
    uint16_t state_table[512];
    int32_t  delta_nb[64];      /* per symbol */
    int32_t  delta_find[64];
    uint16_t first_state[64];   /* encoder state (tableSize + u) of the lowest cell holding the symbol */
    int      tlog;
} fse_ctable;

/* norm[s]: >0 count, 0 absent, -1 "less than one" (one cell at the high end).  nsym <= 64, tlog <= 9. */
static void fse_build_ctable(fse_ctable *ct, const int16_t *norm, int nsym, int tlog) {
    int size = 1 << tlog, high = size - 1;
    uint8_t cell[512];
    int cum[65];
    for (int s = 0; s < nsym; s++) if (norm[s] == -1) cell[high--] = (uint8_t)s;
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s < nsym; s++)
        for (int i = 0; i < norm[s]; i++) { cell[pos] = (uint8_t)s; do { pos = (pos + step) & mask; } while (pos > high); }
    cum[0] = 0;
    for (int s = 0; s < nsym; s++) { int n = norm[s] == -1 ? 1 : norm[s]; cum[s + 1] = cum[s] + n; }
    int fill[64];
    for (int s = 0; s < nsym; s++) { fill[s] = cum[s]; ct->first_state[s] = 0; }
    for (int u = 0; u < size; u++) {
        int s = cell[u];
        if (fill[s] == cum[s]) ct->first_state[s] = (uint16_t)(size + u);
        ct->state_table[fill[s]++] = (uint16_t)(size + u);
    }
    for (int s = 0; s < nsym; s++) {
        int n = norm[s] == -1 ? 1 : norm[s];
        if (n == 0) { ct->delta_nb[s] = 0; ct->delta_find[s] = 0; continue; }
        int maxbits = (n == 1) ? tlog : tlog - hb32((uint32_t)(n - 1));
        ct->delta_nb[s] = (maxbits << 16) - (n << maxbits);
        ct->delta_find[s] = cum[s] - n;
    }
    ct->tlog = tlog;
}
static uint32_t fse_encode(const fse_ctable *ct, bitw *w, uint32_t state, int s) {
    int nb = (int)((state + (uint32_t)ct->delta_nb[s]) >> 16);
    bw_add(w, state, nb);
    return ct->state_table[(state >> nb) + ct->delta_find[s]];
}

/* count[] -> norm[] summing to 1<<tlog; every present symbol gets >= 1; no -1 entries are produced. */
static void fse_normalize(const uint32_t *count, int nsym, uint32_t total, int tlog, int16_t *norm) {
    int size = 1 << tlog, sum = 0, best = 0;
    for (int s = 0; s < nsym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        uint32_t q = (uint32_t)(((uint64_t)count[s] << tlog) / total);
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

/* FSE table description (RFC 8878 4.1.1); returns bytes written */
static size_t fse_write_ncount(uint8_t *dst, const int16_t *norm, int nsym, int tlog) {
    bitw w; bw_init(&w, dst);
    bw_add(&w, (uint32_t)(tlog - 5), 4);
    int remaining = (1 << tlog) + 1, threshold = 1 << tlog, nbits = tlog + 1, s = 0;
    while (remaining > 1 && s < nsym) {
        int count = norm[s++];
        int max = (2 * threshold - 1) - remaining;
        remaining -= count < 0 ? -count : count;
        int v = count + 1;
        if (v >= threshold) v += max;
        bw_add(&w, (uint32_t)v, nbits - (v < max ? 1 : 0));
        if (count == 0) {
            int z = 0;
            while (s + z < nsym && norm[s + z] == 0) z++;
            s += z;
            while (z >= 3) { bw_add(&w, 3, 2); z -= 3; }
            bw_add(&w, (uint32_t)z, 2);
        }
        while (remaining < threshold) { nbits--; threshold >>= 1; }
    }
    return bw_flush_plain(&w);
}

/* ======================================================================== Huffman */

#define HUF_MAX 11

/* code lengths (<= 11) for symbols with count > 0; returns number of present symbols */
static int huf_build_lens(const uint32_t *count, int nsym, uint8_t *lens) {
    int order[256], n = 0;
    for (int s = 0; s < nsym; s++) { lens[s] = 0; if (count[s]) order[n++] = s; }
    if (n < 2) return n;
    /* sort by (count asc, symbol asc) -- insertion sort */
    for (int i = 1; i < n; i++) {
        int x = order[i], j = i - 1;
        while (j >= 0 && count[order[j]] > count[x]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = x;
    }
    /* two-queue Huffman: nodes 0..n-1 leaves (sorted), n.. internal in creation order */
    uint64_t wt[512]; int parent[512];
    for (int i = 0; i < n; i++) wt[i] = count[order[i]];
    int lq = 0, iq = n, nn = n;
    while (nn < 2 * n - 1) {
        int a, b;
        if (lq < n && (iq >= nn || wt[lq] <= wt[iq])) a = lq++; else a = iq++;
        if (lq < n && (iq >= nn || wt[lq] <= wt[iq])) b = lq++; else b = iq++;
        wt[nn] = wt[a] + wt[b]; parent[a] = nn; parent[b] = nn; nn++;
    }
    int depth[512];
    depth[nn - 1] = 0;
    for (int i = nn - 2; i >= 0; i--) depth[i] = depth[parent[i]] + 1;
    int over = 0;
    for (int i = 0; i < n; i++) { int d = depth[i]; if (d > HUF_MAX) { d = HUF_MAX; over = 1; } lens[order[i]] = (uint8_t)d; }
    if (!over) return n;
    /* length limiting: Kraft repair in units of 2^-11.  order[] is rarest-first. */
    int32_t K = 0;
    for (int i = 0; i < n; i++) K += 1 << (HUF_MAX - lens[order[i]]);
    int32_t debt = K - (1 << HUF_MAX);
    while (debt > 0) {
        /* lengthen the rarest symbol among those with the largest length < 11 */
        int pick = -1, bl = 0;
        for (int i = 0; i < n; i++) { int l = lens[order[i]]; if (l < HUF_MAX && l > bl) { bl = l; pick = i; } }
        lens[order[pick]]++; debt -= 1 << (HUF_MAX - 1 - bl);
    }
    while (debt < 0) {
        /* shorten the most frequent symbol whose gain 2^(11-len) fits in the slack; prefer the largest gain */
        int pick = -1, bl = 99; int32_t slack = -debt;
        for (int i = n - 1; i >= 0; i--) { int l = lens[order[i]]; if (l > 1 && (1 << (HUF_MAX - l)) <= slack && l < bl) { bl = l; pick = i; } }
        if (pick < 0) return -1; /* cannot complete the code: caller falls back to raw literals */
        lens[order[pick]]--; debt += 1 << (HUF_MAX - bl);
    }
    return n;
}

/* canonical zstd codes from lengths: within the decoding table, weight-1 symbols first (ascending symbol),
 * then weight 2, ...; code = cell index >> (weight-1), i.e. numerically increasing within one length. */
static void huf_assign_codes(const uint8_t *lens, int nsym, int maxbits, uint16_t *codes) {
    uint32_t pos = 0;
    for (int w = 1; w <= maxbits; w++) {
        int l = maxbits + 1 - w;
        for (int s = 0; s < nsym; s++) if (lens[s] == l) { codes[s] = (uint16_t)(pos >> (w - 1)); pos += 1u << (w - 1); }
    }
}

/* Huffman tree description; returns bytes written, 0 when not representable */
static size_t huf_write_tree(uint8_t *dst, const uint8_t *lens, int max_sym, int maxbits) {
    uint8_t wts[256];
    int nw = max_sym;                                     /* explicit weights for symbols 0..max_sym-1 */
    for (int s = 0; s < nw; s++) wts[s] = lens[s] ? (uint8_t)(maxbits + 1 - lens[s]) : 0;
    /* FSE-compressed weights */
    size_t fse_size = 0; uint8_t tmp[300];
    {
        uint32_t cnt[16] = {0}; int maxw = 0, distinct = 0; uint32_t maxc = 0;
        for (int i = 0; i < nw; i++) { cnt[wts[i]]++; if (wts[i] > maxw) maxw = wts[i]; }
        for (int v = 0; v <= maxw; v++) { if (cnt[v]) distinct++; if (cnt[v] > maxc) maxc = cnt[v]; }
        if (distinct >= 2 && nw >= 2 && maxc > 1) {
            int tlog = hb32((uint32_t)(nw - 1)) - 2; int minlog = 5;
            while ((1 << minlog) < distinct) minlog++;
            if (tlog < minlog) tlog = minlog;
            if (tlog > 6) tlog = 6;
            int16_t norm[16];
            fse_normalize(cnt, maxw + 1, (uint32_t)nw, tlog, norm);
            size_t hs = fse_write_ncount(tmp + 1, norm, maxw + 1, tlog);
            fse_ctable ct; fse_build_ctable(&ct, norm, maxw + 1, tlog);
            bitw w; bw_init(&w, tmp + 1 + hs);
            /* two interleaved states; even symbol indices belong to state 1, odd to state 2 */
            int i = nw; uint32_t s1, s2;
            if (nw & 1) { s1 = ct.first_state[wts[--i]]; s2 = ct.first_state[wts[--i]]; s1 = fse_encode(&ct, &w, s1, wts[--i]); }
            else { s2 = ct.first_state[wts[--i]]; s1 = ct.first_state[wts[--i]]; }
            while (i > 0) { s2 = fse_encode(&ct, &w, s2, wts[--i]); s1 = fse_encode(&ct, &w, s1, wts[--i]); }
            bw_add(&w, s2, tlog); bw_add(&w, s1, tlog);
            size_t bs = bw_close_marker(&w);
            if (hs + bs < 128) { fse_size = hs + bs; tmp[0] = (uint8_t)fse_size; }
        }
    }
    size_t direct_size = (nw <= 128) ? (size_t)(nw + 1) / 2 : 0;
    if (fse_size && (!direct_size || fse_size < direct_size)) { memcpy(dst, tmp, 1 + fse_size); return 1 + fse_size; }
    if (!direct_size) return 0;
    dst[0] = (uint8_t)(127 + nw);
    for (int i = 0; i < nw; i += 2) dst[1 + i / 2] = (uint8_t)((wts[i] << 4) | (i + 1 < nw ? wts[i + 1] : 0));
    return 1 + direct_size;
}

/* one backward-read Huffman stream: symbol m-1 at bit 0 upward, then the closing 1 */
static size_t huf_encode_stream(uint8_t *dst, const uint8_t *sym, uint32_t m, const uint8_t *lens, const uint16_t *codes) {
    bitw w; bw_init(&w, dst);
    for (uint32_t j = m; j-- > 0;) bw_add(&w, codes[sym[j]], lens[sym[j]]);
    return bw_close_marker(&w);
}

/* literals section; returns bytes written */
static size_t encode_literals(uint8_t *dst, const uint8_t *lits, uint32_t nlit, uint32_t flags) {
    /* raw header size */
    size_t raw_h = nlit < 32 ? 1 : (nlit < 4096 ? 2 : 3);
    if ((flags & PNA_F_HUF) && nlit >= 64) {
        uint32_t count[256] = {0}; int max_sym = 0; uint32_t maxc = 0;
        for (uint32_t i = 0; i < nlit; i++) count[lits[i]]++;
        for (int s = 0; s < 256; s++) if (count[s]) { max_sym = s; if (count[s] > maxc) maxc = count[s]; }
        if (maxc == nlit) { /* RLE literals */
            if (raw_h == 1) dst[0] = (uint8_t)(1 | (nlit << 3));
            else if (raw_h == 2) { dst[0] = (uint8_t)(1 | (1 << 2) | ((nlit & 15) << 4)); dst[1] = (uint8_t)(nlit >> 4); }
            else { dst[0] = (uint8_t)(1 | (3 << 2) | ((nlit & 15) << 4)); dst[1] = (uint8_t)(nlit >> 4); dst[2] = (uint8_t)(nlit >> 12); }
            dst[raw_h] = lits[0];
            return raw_h + 1;
        }
        uint8_t lens[256]; uint16_t codes[256];
        int np = huf_build_lens(count, 256, lens);
        if (np >= 2) {
            int maxbits = 0;
            for (int s = 0; s <= max_sym; s++) if (lens[s] > maxbits) maxbits = lens[s];
            huf_assign_codes(lens, max_sym + 1, maxbits, codes);
            int streams4 = nlit >= 256;
            size_t lh = 3 + (nlit >= 1024) + (nlit >= 16384);
            uint8_t *body = dst + lh;
            size_t ts = huf_write_tree(body, lens, max_sym, maxbits);
            if (ts) {
                size_t csz = ts;
                if (!streams4) csz += huf_encode_stream(body + csz, lits, nlit, lens, codes);
                else {
                    uint32_t seg = (nlit + 3) / 4;
                    uint8_t *jt = body + csz; csz += 6;
                    for (int k = 0; k < 4; k++) {
                        uint32_t a = (uint32_t)k * seg, m = k < 3 ? seg : nlit - 3 * seg;
                        size_t ss = huf_encode_stream(body + csz, lits + a, m, lens, codes);
                        if (k < 3) { jt[2 * k] = (uint8_t)ss; jt[2 * k + 1] = (uint8_t)(ss >> 8); }
                        csz += ss;
                    }
                }
                if (lh + csz < raw_h + nlit) {
                    uint64_t h;
                    if (lh == 3) h = 2u | ((uint64_t)(streams4 ? 1 : 0) << 2) | ((uint64_t)nlit << 4) | ((uint64_t)csz << 14);
                    else if (lh == 4) h = 2u | (2u << 2) | ((uint64_t)nlit << 4) | ((uint64_t)csz << 18);
                    else h = 2u | (3u << 2) | ((uint64_t)nlit << 4) | ((uint64_t)csz << 22);
                    for (size_t i = 0; i < lh; i++) dst[i] = (uint8_t)(h >> (8 * i));
                    return lh + csz;
                }
            }
        }
    }
    if (raw_h == 1) dst[0] = (uint8_t)(nlit << 3);
    else if (raw_h == 2) { dst[0] = (uint8_t)((1 << 2) | ((nlit & 15) << 4)); dst[1] = (uint8_t)(nlit >> 4); }
    else { dst[0] = (uint8_t)((3 << 2) | ((nlit & 15) << 4)); dst[1] = (uint8_t)(nlit >> 4); dst[2] = (uint8_t)(nlit >> 12); }
    memcpy(dst + raw_h, lits, nlit);
    return raw_h + nlit;
}

/* ======================================================================== sequences */

static const int16_t LL_DEF[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const int16_t ML_DEF[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
static const int16_t OF_DEF[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
static const uint32_t LL_BASE[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
static const uint8_t  LL_BITS[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const uint32_t ML_BASE[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
static const uint8_t  ML_BITS[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};

static int ll_code(uint32_t v) { int c = 35; while (LL_BASE[c] > v) c--; return c; }
static int ml_code(uint32_t v) { int c = 52; while (ML_BASE[c] > v) c--; return c; }

/* table for one of LL/OF/ML: chooses mode, writes its description, builds the encoder table.
 * mode: 0 predefined, 1 RLE, 2 FSE_Compressed. */
static int seq_make_table(fse_ctable *ct, const uint8_t *codes, uint32_t nseq, int alphabet, int max_log,
                          const int16_t *def, int def_n, int def_log, uint32_t flags, uint8_t *desc, size_t *desc_len) {
    uint32_t count[64] = {0}; int maxs = 0, distinct = 0;
    for (uint32_t i = 0; i < nseq; i++) count[codes[i]]++;
    for (int s = 0; s < alphabet; s++) if (count[s]) { maxs = s; distinct++; }
    *desc_len = 0;
    if (distinct == 1 && nseq > 2) { desc[0] = (uint8_t)maxs; *desc_len = 1; ct->tlog = 0; return 1; }
    int def_ok = maxs < def_n;
    if (def_ok) for (int s = 0; s <= maxs; s++) if (count[s] && def[s] == 0) def_ok = 0;
    if (!(flags & PNA_F_FSE) || (nseq < 64 && def_ok)) {
        if (!def_ok) return -1;
        fse_build_ctable(ct, def, def_n, def_log); return 0;
    }
    int tlog = hb32(nseq - 1) - 2, minlog = 5;
    while ((1 << minlog) < distinct) minlog++;
    if (tlog < minlog) tlog = minlog;
    if (tlog > max_log) tlog = max_log;
    int16_t norm[64];
    fse_normalize(count, maxs + 1, nseq, tlog, norm);
    *desc_len = fse_write_ncount(desc, norm, maxs + 1, tlog);
    fse_build_ctable(ct, norm, maxs + 1, tlog);
    return 2;
}

/* sequences section; returns bytes written (0 => cannot encode, caller emits a raw block) */
static size_t encode_sequences(uint8_t *dst, const pna_seq *seqs, uint32_t nseq, uint32_t flags) {
    size_t pos = 0;
    if (nseq < 128) dst[pos++] = (uint8_t)nseq;
    else if (nseq < 0x7F00) { dst[pos++] = (uint8_t)((nseq >> 8) + 128); dst[pos++] = (uint8_t)nseq; }
    else { dst[pos++] = 255; dst[pos++] = (uint8_t)(nseq - 0x7F00); dst[pos++] = (uint8_t)((nseq - 0x7F00) >> 8); }
    if (nseq == 0) return pos;
    uint8_t *llc = (uint8_t *)malloc(nseq * 3), *ofc = llc + nseq, *mlc = ofc + nseq;
    uint32_t *ofb = (uint32_t *)malloc(nseq * sizeof(uint32_t));
    /* offBase: offset+3, or a repeat code 1..3 when PNA_F_REP.  Only history established INSIDE this block is
     * used (rep[k] == 0 means "unknown"), so a block stays decodable whatever the previous blocks were
     * (raw fallback drops their sequences and with them their history updates). */
    uint32_t rep[3] = {0, 0, 0};
    for (uint32_t i = 0; i < nseq; i++) {
        uint32_t off = seqs[i].off, ob = off + 3;
        if (flags & PNA_F_REP) {
            if (seqs[i].ll != 0) {
                if (off == rep[0]) ob = 1;
                else if (off == rep[1]) { ob = 2; rep[1] = rep[0]; rep[0] = off; }
                else if (off == rep[2]) { ob = 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
            } else {
                if (off == rep[1]) { ob = 1; rep[1] = rep[0]; rep[0] = off; }
                else if (off == rep[2]) { ob = 2; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
                else if (rep[0] > 1 && off == rep[0] - 1) { ob = 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
            }
        }
        if (ob > 3) { rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
        ofb[i] = ob;
        llc[i] = (uint8_t)ll_code(seqs[i].ll); mlc[i] = (uint8_t)ml_code(seqs[i].ml);
        ofc[i] = (uint8_t)hb32(ob);
    }
    fse_ctable *ctl = (fse_ctable *)malloc(3 * sizeof(fse_ctable)), *cto = ctl + 1, *ctm = ctl + 2;
    uint8_t *modes = dst + pos++; size_t dl;
    int ml_ = seq_make_table(ctl, llc, nseq, 36, 9, LL_DEF, 36, 6, flags, dst + pos, &dl); pos += dl;
    int mo_ = seq_make_table(cto, ofc, nseq, 32, 8, OF_DEF, 29, 5, flags, dst + pos, &dl); pos += dl;
    int mm_ = seq_make_table(ctm, mlc, nseq, 53, 9, ML_DEF, 53, 6, flags, dst + pos, &dl); pos += dl;
    if (ml_ < 0 || mo_ < 0 || mm_ < 0) { free(llc); free(ctl); free(ofb); return 0; }
    *modes = (uint8_t)((ml_ << 6) | (mo_ << 4) | (mm_ << 2));
    bitw w; bw_init(&w, dst + pos);
    uint32_t i = nseq - 1;
    uint32_t sm = mm_ == 1 ? 0 : ctm->first_state[mlc[i]];
    uint32_t so = mo_ == 1 ? 0 : cto->first_state[ofc[i]];
    uint32_t sl = ml_ == 1 ? 0 : ctl->first_state[llc[i]];
    bw_add(&w, seqs[i].ll - LL_BASE[llc[i]], LL_BITS[llc[i]]);
    bw_add(&w, seqs[i].ml - ML_BASE[mlc[i]], ML_BITS[mlc[i]]);
    bw_add(&w, ofb[i] - (1u << ofc[i]), ofc[i]);
    while (i-- > 0) {
        if (mo_ != 1) so = fse_encode(cto, &w, so, ofc[i]);
        if (mm_ != 1) sm = fse_encode(ctm, &w, sm, mlc[i]);
        if (ml_ != 1) sl = fse_encode(ctl, &w, sl, llc[i]);
        bw_add(&w, seqs[i].ll - LL_BASE[llc[i]], LL_BITS[llc[i]]);
        bw_add(&w, seqs[i].ml - ML_BASE[mlc[i]], ML_BITS[mlc[i]]);
        bw_add(&w, ofb[i] - (1u << ofc[i]), ofc[i]);
    }
    if (mm_ != 1) bw_add(&w, sm, ctm->tlog);
    if (mo_ != 1) bw_add(&w, so, cto->tlog);
    if (ml_ != 1) bw_add(&w, sl, ctl->tlog);
    pos += bw_close_marker(&w);
    free(llc); free(ctl); free(ofb);
    return pos;
}

/* ======================================================================== block / frame */

size_t pna_zstd_encode_block(const uint8_t *blk, uint32_t blk_len, const pna_seq *seqs, uint32_t nseq,
                             const uint8_t *lits, uint32_t nlit, int last, uint32_t flags, uint8_t *dst) {
    uint8_t *tmp = (uint8_t *)malloc((size_t)blk_len * 2 + (size_t)nseq * 12 + 1024);
    size_t csz = 0;
    if (nseq > 0 || nlit > 0) {
        size_t ls = encode_literals(tmp, lits, nlit, flags);
        size_t ss = encode_sequences(tmp + ls, seqs, nseq, flags);
        csz = ss ? ls + ss : 0;
    }
    uint32_t hdr;
    if (csz == 0 || csz >= blk_len) {
        hdr = (uint32_t)(last ? 1 : 0) | (0u << 1) | (blk_len << 3);
        memcpy(dst + 3, blk, blk_len); csz = blk_len;
    } else {
        hdr = (uint32_t)(last ? 1 : 0) | (2u << 1) | ((uint32_t)csz << 3);
        memcpy(dst + 3, tmp, csz);
    }
    dst[0] = (uint8_t)hdr; dst[1] = (uint8_t)(hdr >> 8); dst[2] = (uint8_t)(hdr >> 16);
    free(tmp);
    return 3 + csz;
}

size_t pna_zstd_model_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, const pna_zstd_params *p) {
    if (cap < pna_zstd_bound(n)) return 0;
    size_t op = 0;
    if (n == 0) { static const uint8_t e[9] = {0x28,0xB5,0x2F,0xFD,0x20,0x00,0x01,0x00,0x00}; memcpy(dst, e, 9); return 9; }
    uint32_t *table = (uint32_t *)malloc(sizeof(uint32_t) << p->hash_log);
    pna_seq *seqs = (pna_seq *)malloc(sizeof(pna_seq) * (PNA_BLK_SIZE / 3 + 8));
    uint8_t *lits = (uint8_t *)malloc(PNA_BLK_SIZE + 8);
    for (size_t s0 = 0; s0 < n; s0 += PNA_SEG_SIZE) {
        uint32_t seg_len = (uint32_t)(n - s0 < PNA_SEG_SIZE ? n - s0 : PNA_SEG_SIZE);
        const uint8_t *seg = src + s0;
        static const uint8_t fh[6] = {0x28,0xB5,0x2F,0xFD,0x00,0x50};
        memcpy(dst + op, fh, 6); op += 6;
        memset(table, 0, sizeof(uint32_t) << p->hash_log);
        for (uint32_t b0 = 0; b0 < seg_len; b0 += PNA_BLK_SIZE) {
            uint32_t bl = seg_len - b0 < PNA_BLK_SIZE ? seg_len - b0 : PNA_BLK_SIZE;
            uint32_t nlit, nseq = pna_lz_block(seg, seg_len, b0, bl, table, p, seqs, lits, &nlit);
            op += pna_zstd_encode_block(seg + b0, bl, seqs, nseq, lits, nlit, b0 + bl >= seg_len, p->flags, dst + op);
        }
    }
    free(table); free(seqs); free(lits);
    return op;
}
