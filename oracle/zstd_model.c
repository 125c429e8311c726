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
 * SEGMENT -> BLOCKS 128 KiB blocks (p->blk_log: 8 - 64 KiB in the device's latency mode); the hash table persists across the blocks of a segment, so matches reach
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

/* The default level set of the product: its match finder's 32 KiB-window geometry with the PACKED table (49 062 slots in words of three, round 4; with
 * 32-bit entries 32 704; candidates more than 23 296 bytes back count as "far").  The fast / balanced sets, the sets with the table in global memory and deflate run the 64 KiB geometry: 24 512 slots, far beyond 56 064
 * (oracle/codec.py params_for_flags sets those). */
void pna_zstd_default_params(pna_zstd_params *p) {
    p->hash_log = 49062; p->min_match = 6; p->tile = 4096; p->max_off = (1u << 19) - 1; p->cap1 = 32;
    p->lookahead = 1024; p->flags = PNA_F_HUF | PNA_F_FSE | PNA_F_LAZY | PNA_F_LAZY2 | PNA_F_LAZY3 | PNA_F_REP; p->max_len = 0; p->region = 256;
    p->ins_mod = 2; p->back_cap = 3; p->rounds = 0x21; p->near_off = 23296; p->cap_far = 32;
    p->blk_log = 0; p->len_word_max = 36; p->tab3 = 1;
    p->mtile = 0; p->small_seg = 4096; p->small_slots = 2048; p->small_tile = 256; p->mid_seg = 16384; p->mid_slots = 2048;
    p->cut_min = 6; p->far_slots = 63; p->far_from = 28368; p->fixup = 1;
}
const pna_zstd_params *pna_seg_params(const pna_zstd_params *p, uint32_t seg_len, pna_zstd_params *tmp) {
    uint32_t slots = 0;
    if (p->small_seg && seg_len <= p->small_seg) slots = p->small_slots;
    else if (p->small_seg && p->mid_seg && seg_len <= p->mid_seg) slots = p->mid_slots;
    if (!slots) return p;
    *tmp = *p; tmp->hash_log = slots; tmp->tab3 = 0; tmp->mtile = p->small_tile;
    return tmp;
}

/* block size of a parameter set: 128 KiB unless blk_log names a smaller power of two (the device's latency mode: small batches are cut
 * into 8 - 64 KiB blocks so that the per-block serial chains are short) */
uint32_t pna_blk_size(const pna_zstd_params *p) {
    return (p->blk_log >= PNA_BLK_LOG_MIN && p->blk_log < 17) ? 1u << p->blk_log : PNA_BLK_SIZE;
}

size_t pna_zstd_bound(size_t n) {
    size_t segs = (n + PNA_SEG_SIZE - 1) / PNA_SEG_SIZE; if (segs == 0) segs = 1;
    size_t blks = (n + PNA_BLK_MIN - 1) / PNA_BLK_MIN + segs;      /* (the smallest block size a parameter set may name) */
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
    /* hash_log <= 31: a table of 2^hash_log entries, index = the top bits; larger values ARE the entry count (any size that fits the
     * LDS): index = floor(h * count / 2^32) */
    return hash_log <= 31 ? h >> (32 - hash_log) : (uint32_t)(((uint64_t)h * hash_log) >> 32);
}
/* the 32-bit hash itself (the packed table's tag = its bits 16..17) */
static uint32_t lz_h32(const uint8_t *p, uint32_t min_match) {
    uint32_t lo = rd32(p), hi = 0;
    if (min_match >= 5) hi = p[4];
    if (min_match >= 6) hi |= (uint32_t)p[5] << 8;
    return lo * 0x9E3779B1u + hi * 0x85EBCA6Bu;
}
/* the packed table (tab3): slot = 3 * word + field; the word from the hash's top bits, the field from its low 16 */
static uint32_t lz_hash3(const uint8_t *p, uint32_t min_match, uint32_t slots) {
    uint32_t lo = rd32(p);
    uint32_t hi = 0;
    if (min_match >= 5) hi = p[4];
    if (min_match >= 6) hi |= (uint32_t)p[5] << 8;
    uint32_t h = lo * 0x9E3779B1u + hi * 0x85EBCA6Bu;
    return 3u * (uint32_t)(((uint64_t)h * (slots / 3)) >> 32) + (((h & 0xFFFFu) * 3u) >> 16);
}

/*
 * One 128 KiB block.  For each tile of p->tile positions, in this order:
 *   L  every position q with q + 8 <= seg_len looks up cand[q] = table[hash(q)] (value = position+1, 0 = empty);
 *   M  len[q] = length of the common prefix of seg[q..] and seg[c..] (c = cand-1), capped to cap(q) and to the block end;
 *      a candidate is usable iff c >= 8 and q - c <= max_off; len < min_match counts as 0.  cap(q) = cap1 when the offset
 *      q - c <= near_off (the candidate lies in the GPU's LDS window) and cap_far otherwise (the candidate is read from
 *      HBM/L2).  For a usable match back[q] = number of equal bytes immediately before q and c, at most back_cap (<= 4);
 *   A  backward adoption, rounds of shift s (the nibbles of p->rounds, lowest first; the GPU's DPP lane shifts): all positions
 *      at once, from the values of the previous round: position q with (q & 63) + s <= 63 adopts the match of j = q + s --
 *      len[q] = len[j] + s (not capped again), cand[q] = cand[j] - s, back[q] = back[j] - s, same cap kind -- iff
 *      len[j] >= min_match, back[j] >= s and len[j] + s > len[q].  (A match found one or two positions late -- the table
 *      holds every ins_mod-th position only -- is thereby moved to its true start.)
 *   I  every position q with q + 8 <= seg_len and q % ins_mod == 0 stores table[hash(q)] = max(old, q+1)   (ascending q here
 *      == atomic max on the GPU; after L for the whole tile, so the positions of one tile do not see each other);
 *   P  region-local greedy parse.  The tile is cut into regions of p->region positions (what one GPU wave owns; region 0
 *      = the whole tile).  Every region is parsed on its own, in ascending q from max(region start, next_free): position q
 *      starts a match iff len[q] >= min_match and not (LAZY and (q & 63) != 63 and q+1 < tile end and len[q+1] > len[q]) and not
 *      (LAZY2 and (q & 63) < 62 and q+2 < tile end and len[q+2] > len[q] + 1);
 *      a chosen match whose len >= its cap is extended byte-wise up to min(block end, tile end + lookahead) (and max_len);
 *      the region's parse continues at q + len and stops at the region end (its last match may reach beyond it).
 *   F  merge of the regions in ascending order against the running end E of the emitted matches (E = next_free at the
 *      tile start): a region that lies entirely below E contributes nothing; otherwise a match that ends at or before E
 *      is dropped, a match that starts before E and ends r bytes after it is cut from the front (start E, length r, same
 *      offset) when r >= cut_min (3 for deflate, min_match for zstd) and dropped otherwise, and any other match is emitted as parsed.  E becomes the end of every
 *      emitted match; next_free = E after the tile.
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
    uint8_t *back = (uint8_t *)calloc(2 * (T + 1), 1), *far = back + T + 1;      /* back[q]; far[q] = 1: cap kind cap_far */
    uint16_t *len0 = (uint16_t *)malloc(sizeof(uint16_t) * (T + 1)); uint32_t *cand0 = (uint32_t *)malloc(sizeof(uint32_t) * T);
    uint8_t *back0 = (uint8_t *)malloc(2 * (T + 1)), *far0 = back0 + T + 1;
    const uint32_t ins_mod = p->ins_mod ? p->ins_mod : 1;
    const uint32_t min_c = p->back_cap > 7 ? 16u : 8u;            /* a usable candidate's position + 1 exceeds it */
    uint32_t *mq = (uint32_t *)malloc(sizeof(uint32_t) * (T + 1) * 4), *ml = mq + T + 1, *mc = ml + T + 1, *mr = mc + T + 1;
    uint8_t *vis = (uint8_t *)calloc(T + 1, 1);                   /* F (fixup): positions a region's own walk stood on */
    uint32_t *wbest = p->tab3 ? (uint32_t *)calloc(p->hash_log / 3 + 1, sizeof(uint32_t)) : NULL, *wlist = p->tab3 ? (uint32_t *)malloc(sizeof(uint32_t) * (T + 1)) : NULL;
    for (uint32_t t0 = blk_start; t0 < blk_end; t0 += T) {
        uint32_t t1 = t0 + T < blk_end ? t0 + T : blk_end;
        /* L, I: per sub-tile of MT positions (the whole tile unless p->mtile says otherwise) */
        const uint32_t MT = p->mtile ? p->mtile : T;
        for (uint32_t m0 = t0; m0 < t1; m0 += MT) {
        const uint32_t m1 = m0 + MT < t1 ? m0 + MT : t1;
        for (uint32_t q = m0; q < m1; q++)
            cand[q - t0] = (q + 8 > seg_len) ? 0 : table[p->tab3 ? lz_hash3(seg + q, p->min_match, p->hash_log) : lz_hash(seg + q, p->min_match, p->hash_log)];
        /* I */
        if (p->tab3) {
            /* per word of three slots, ONE of the tile's inserts is stored: the contender with the highest (field, position) -- the device's single
             * 64-bit maximum of "the word as the look-ups saw it, my field replaced".  (Position 0 is never stored: its entry is the empty one.) */
            uint32_t nw = 0;
            for (uint32_t q = m0; q < m1; q++) {
                if (q + 8 > seg_len || q % ins_mod || q == 0) continue;
                const uint32_t s3 = lz_hash3(seg + q, p->min_match, p->hash_log), w = s3 / 3;
                const uint32_t key = ((s3 % 3 + 1) << 24) | (q - t0);               /* (field + 1, position): 0 = no contender yet */
                if (!wbest[w]) wlist[nw++] = w;
                if (key > wbest[w]) wbest[w] = key;
            }
            for (uint32_t i = 0; i < nw; i++) {
                const uint32_t w = wlist[i], key = wbest[w];
                wbest[w] = 0;
                table[3 * w + (key >> 24) - 1] = t0 + (key & 0xFFFFFFu) + 1;
            }
        } else
        for (uint32_t q = m0; q < m1; q++)
            if (q + 8 <= seg_len) {
                uint32_t h = lz_hash(seg + q, p->min_match, p->hash_log);
                if (q % ins_mod == 0 && table[h] < q + 1) table[h] = q + 1;
            }
        }
        /* far candidates beyond the wave's slots (far_slots): dropped before M */
        if (p->far_slots && p->tab3) {
            for (uint32_t w0 = t0; w0 < t1; w0 += 256) {
                uint32_t idx = 0;
                for (uint32_t j = 0; j < 4; j++)
                    for (uint32_t q = w0 + j; q < w0 + 256 && q < t1; q += 4) {
                        const uint32_t c1 = cand[q - t0];
                        if (c1 <= min_c || q - (c1 - 1) > p->max_off || q - (c1 - 1) < p->far_from) continue;
                        if (((lz_h32(seg + q, p->min_match) ^ lz_h32(seg + c1 - 1, p->min_match)) >> 16) & 3u) continue;      /* foreign tag: no candidate for the device either */
                        if (idx++ >= p->far_slots) cand[q - t0] = 0;
                    }
            }
        }
        /* M */
        for (uint32_t q = t0; q < t1; q++) {
            uint32_t c1 = cand[q - t0], l = 0, bk = 0, fr = 0;
            if (c1 > min_c && q - (c1 - 1) <= p->max_off) {              /* candidates at positions 0..7 (0..15 with more than 7 back bytes) are not used: 8 (16) bytes before a candidate always exist */
                uint32_t c = c1 - 1, lim = blk_end - q;
                fr = p->near_off && q - c > p->near_off;
                uint32_t cap = fr ? p->cap_far : p->cap1;
                if (lim > cap) lim = cap;
                while (l < lim && seg[q + l] == seg[c + l]) l++;
                if (l < p->min_match) l = 0;
                if (l) while (bk < p->back_cap && seg[q - 1 - bk] == seg[c - 1 - bk]) bk++;
            }
            len[q - t0] = (uint16_t)l; back[q - t0] = (uint8_t)bk; far[q - t0] = (uint8_t)(l ? fr : 0);
        }
        len[t1 - t0] = 0;
        /* A */
        for (uint32_t rr = p->rounds; rr; rr >>= 4) {
            const uint32_t sft = rr & 15;
            memcpy(len0, len, sizeof(uint16_t) * (T + 1)); memcpy(cand0, cand, sizeof(uint32_t) * T); memcpy(back0, back, 2 * (T + 1));
            for (uint32_t q = t0; q + sft < t1; q++) {
                const uint32_t j = q + sft - t0;
                if ((q & 63) + sft > 63 || len0[j] < p->min_match || back0[j] < sft || len0[j] + sft <= len0[q - t0]) continue;
                len[q - t0] = (uint16_t)(len0[j] + sft); cand[q - t0] = cand0[j] - sft;
                back[q - t0] = (uint8_t)(back0[j] - sft); far[q - t0] = far0[j];
            }
        }
        if (p->len_word_max) for (uint32_t i = 0; i < t1 - t0; i++) if (len[i] > p->len_word_max) len[i] = (uint16_t)p->len_word_max;
        /* P */
#define PNA_TAKE(q_, l_) ((l_) >= p->min_match && \
            !((p->flags & PNA_F_LAZY) && ((q_) & 63) != 63 && (q_) + 1 < t1 && len[(q_) + 1 - t0] > (l_)) && \
            !((p->flags & PNA_F_LAZY) && (p->flags & PNA_F_LAZY2) && ((q_) & 63) < 62 && (q_) + 2 < t1 && len[(q_) + 2 - t0] > (l_) + 1) && \
            !((p->flags & PNA_F_LAZY) && (p->flags & PNA_F_LAZY3) && ((q_) & 63) < 61 && (q_) + 3 < t1 && len[(q_) + 3 - t0] > (l_) + 2))
        uint32_t ext_lim = t1 + p->lookahead < blk_end ? t1 + p->lookahead : blk_end;
        uint32_t R = p->region ? p->region : T;
        uint32_t nm = 0;                                  /* matches of this tile: mq (start), ml (length), mc (candidate) */
        memset(vis, 0, T + 1);
        for (uint32_t r0 = t0; r0 < t1; r0 += R) {
            uint32_t r1 = r0 + R < t1 ? r0 + R : t1;
            for (uint32_t q = (next_free > r0 ? next_free : r0); q < r1; ) {
                uint32_t l = len[q - t0];
                vis[q - t0] = 1;                                     /* the region's walk stands on q */
                if (!PNA_TAKE(q, l)) { q++; continue; }
                uint32_t c = cand[q - t0] - 1;
                if (l >= (far[q - t0] ? p->cap_far : p->cap1)) {
                    uint32_t el = ext_lim;
                    if (p->max_len && q + p->max_len < el) el = q + p->max_len;
                    while (q + l < el && seg[q + l] == seg[c + l]) l++;
                }
                mq[nm] = q; ml[nm] = l; mc[nm] = c; mr[nm] = r1; nm++;
                q += l;
            }
        }
        /* F */
        for (uint32_t i = 0; i < nm; i++) {
            uint32_t q = mq[i], l = ml[i], c = mc[i];
            if (q + l <= next_free || mr[i] <= next_free) continue;      /* mr = end of the match's region */
            if (q < next_free) {
                uint32_t r = q + l - next_free;
                if (r < (p->cut_min ? p->cut_min : 3u)) {
                    /* fixup (round 5): instead of only dropping the remainder, ONE match from inside it -- the first position s in [E, end of the straddling match) that the
                     * walk's own rule would take, if its match is not a capped one (no extension here), ends inside the region and the tile, and ends on a position the
                     * region's walk stood on: from there on the region's parse is the one that entered at E.  The region's matches that start before that end are dropped
                     * with the straddling one.  Only where the straddling match itself ends inside its region (its end is then a position the walk stood on). */
                    if (p->fixup && q + l < mr[i]) {
                        uint32_t s = next_free;
                        while (s < q + l && !PNA_TAKE(s, len[s - t0])) s++;
                        if (s < q + l) {
                            const uint32_t l2 = len[s - t0], x = s + l2;
                            if (l2 < (far[s - t0] ? p->cap_far : p->cap1) && x < mr[i] && x < t1 && vis[x - t0]) {
                                seqs[nseq].ll = s - lit_start; seqs[nseq].ml = l2; seqs[nseq].off = s - (cand[s - t0] - 1); nseq++;
                                memcpy(lits + nlit, seg + lit_start, s - lit_start); nlit += s - lit_start;
                                lit_start = x; next_free = x;
                                while (i + 1 < nm && mr[i + 1] == mr[i] && mq[i + 1] < x) i++;
                            }
                        }
                    }
                    continue;
                }
                c += next_free - q; q = next_free; l = r;
            }
            seqs[nseq].ll = q - lit_start; seqs[nseq].ml = l; seqs[nseq].off = q - c; nseq++;
            memcpy(lits + nlit, seg + lit_start, q - lit_start); nlit += q - lit_start;
            lit_start = q + l; next_free = lit_start;
        }
    }
    memcpy(lits + nlit, seg + lit_start, blk_end - lit_start); nlit += blk_end - lit_start;
    free(vis); free(cand); free(len); free(mq); free(back); free(len0); free(cand0); free(back0); free(wbest); free(wlist);
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

/* ======================================================================== per-segment entropy tables
 *
 * ONE Huffman table and ONE set of LL/OF/ML tables per segment (1 MiB frame), built from the statistics of all
 * the segment's blocks.  The first block that uses a table carries its description (Compressed literals /
 * FSE_Compressed or RLE modes); later blocks of the segment reuse it (Treeless literals / Repeat_Mode).
 * This is what lets the GPU encode sequences with one lane per block while the lanes of a segment share one
 * table image in LDS.
 */
typedef struct {
    int      huf_ok, max_sym, maxbits;
    uint8_t  lens[256]; uint16_t codes[256];
    uint8_t  tree[160]; size_t tree_len;
    int      mode[3];                 /* 0 predefined, 1 RLE, 2 FSE_Compressed; order LL, OF, ML */
    fse_ctable ct[3];
    uint8_t  desc[3][96]; size_t desc_len[3];
} seg_tables;

static void seg_build_huf(seg_tables *t, const uint32_t *count) {
    t->huf_ok = 0; t->tree_len = 0; t->max_sym = 0; t->maxbits = 0;
    int np = huf_build_lens(count, 256, t->lens);
    if (np < 2) return;
    for (int s = 0; s < 256; s++) if (count[s]) t->max_sym = s;
    for (int s = 0; s <= t->max_sym; s++) if (t->lens[s] > t->maxbits) t->maxbits = t->lens[s];
    huf_assign_codes(t->lens, t->max_sym + 1, t->maxbits, t->codes);
    t->tree_len = huf_write_tree(t->tree, t->lens, t->max_sym, t->maxbits);
    t->huf_ok = t->tree_len > 0;
}

/* returns 0 on success, -1 when no valid table exists (caller emits raw blocks) */
static int seg_build_seq_table(seg_tables *t, int which, const uint32_t *count, uint32_t nseq, int alphabet, int max_log,
                               const int16_t *def, int def_n, int def_log, uint32_t flags) {
    int maxs = 0, distinct = 0;
    for (int s = 0; s < alphabet; s++) if (count[s]) { maxs = s; distinct++; }
    t->desc_len[which] = 0;
    if (distinct == 1 && nseq > 2) { t->desc[which][0] = (uint8_t)maxs; t->desc_len[which] = 1; t->ct[which].tlog = 0; t->mode[which] = 1; return 0; }
    int def_ok = maxs < def_n;
    if (!(flags & PNA_F_FSE) || (nseq < 64 && def_ok)) {
        if (!def_ok) return -1;
        fse_build_ctable(&t->ct[which], def, def_n, def_log); t->mode[which] = 0; return 0;
    }
    int tlog = hb32(nseq - 1) - 2, minlog = 5;
    while ((1 << minlog) < distinct) minlog++;
    if (tlog < minlog) tlog = minlog;
    if (tlog > max_log) tlog = max_log;
    int16_t norm[64];
    fse_normalize(count, maxs + 1, nseq, tlog, norm);
    t->desc_len[which] = fse_write_ncount(t->desc[which], norm, maxs + 1, tlog);
    fse_build_ctable(&t->ct[which], norm, maxs + 1, tlog);
    t->mode[which] = 2;
    return 0;
}

/* offBase for every sequence of ONE block: offset+3, or a repeat code 1..3 when PNA_F_REP.  Only history
 * established INSIDE this block is used (rep[k] == 0 means "unknown"), so a block stays decodable whatever
 * happened to the previous blocks (a raw fallback drops their sequences and with them their history updates). */
static void block_offbase(const pna_seq *seqs, uint32_t nseq, uint32_t flags, uint32_t *ofb) {
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
    }
}

/* Huffman body of one block (jump table + streams, no tree); returns bytes */
static size_t block_huf_body(const seg_tables *t, const uint8_t *lits, uint32_t nlit, uint8_t *dst) {
    if (nlit < 256) return huf_encode_stream(dst, lits, nlit, t->lens, t->codes);
    uint32_t seg = (nlit + 3) / 4; size_t csz = 6;
    for (int k = 0; k < 4; k++) {
        uint32_t a = (uint32_t)k * seg, m = k < 3 ? seg : nlit - 3 * seg;
        size_t ss = huf_encode_stream(dst + csz, lits + a, m, t->lens, t->codes);
        if (k < 3) { dst[2 * k] = (uint8_t)ss; dst[2 * k + 1] = (uint8_t)(ss >> 8); }
        csz += ss;
    }
    return csz;
}

/* sequence bitstream of one block (no headers); returns bytes */
static size_t block_seq_bits(const seg_tables *t, const pna_seq *seqs, const uint32_t *ofb, uint32_t nseq, uint8_t *dst) {
    bitw w; bw_init(&w, dst);
    const fse_ctable *ctl = &t->ct[0], *cto = &t->ct[1], *ctm = &t->ct[2];
    uint32_t i = nseq - 1;
    int llc = ll_code(seqs[i].ll), mlc = ml_code(seqs[i].ml), ofc = hb32(ofb[i]);
    uint32_t sm = t->mode[2] == 1 ? 0 : ctm->first_state[mlc];
    uint32_t so = t->mode[1] == 1 ? 0 : cto->first_state[ofc];
    uint32_t sl = t->mode[0] == 1 ? 0 : ctl->first_state[llc];
    bw_add(&w, seqs[i].ll - LL_BASE[llc], LL_BITS[llc]);
    bw_add(&w, seqs[i].ml - ML_BASE[mlc], ML_BITS[mlc]);
    bw_add(&w, ofb[i] - (1u << ofc), ofc);
    while (i-- > 0) {
        llc = ll_code(seqs[i].ll); mlc = ml_code(seqs[i].ml); ofc = hb32(ofb[i]);
        if (t->mode[1] != 1) so = fse_encode(cto, &w, so, ofc);
        if (t->mode[2] != 1) sm = fse_encode(ctm, &w, sm, mlc);
        if (t->mode[0] != 1) sl = fse_encode(ctl, &w, sl, llc);
        bw_add(&w, seqs[i].ll - LL_BASE[llc], LL_BITS[llc]);
        bw_add(&w, seqs[i].ml - ML_BASE[mlc], ML_BITS[mlc]);
        bw_add(&w, ofb[i] - (1u << ofc), ofc);
    }
    if (t->mode[2] != 1) bw_add(&w, sm, ctm->tlog);
    if (t->mode[1] != 1) bw_add(&w, so, cto->tlog);
    if (t->mode[0] != 1) bw_add(&w, sl, ctl->tlog);
    return bw_close_marker(&w);
}

static size_t put_raw_lit_header(uint8_t *dst, int type, uint32_t nlit) {
    if (nlit < 32) { dst[0] = (uint8_t)(type | (nlit << 3)); return 1; }
    if (nlit < 4096) { dst[0] = (uint8_t)(type | (1 << 2) | ((nlit & 15) << 4)); dst[1] = (uint8_t)(nlit >> 4); return 2; }
    dst[0] = (uint8_t)(type | (3 << 2) | ((nlit & 15) << 4)); dst[1] = (uint8_t)(nlit >> 4); dst[2] = (uint8_t)(nlit >> 12); return 3;
}

/* ======================================================================== segment = one frame
 *
 * blk_nseq/blk_nlit/seqs/lits are the LZ stage's outputs for the segment's blocks (seqs and lits are stored at
 * the block's input offset scaled: seqs at index b*(BLK/4), lits at byte b*BLK).  Returns frame bytes.
 */
size_t pna_zstd_encode_segment(const uint8_t *seg, uint32_t seg_len, const pna_seq *seqs, const uint8_t *lits,
                               const uint32_t *blk_nseq, const uint32_t *blk_nlit, uint32_t flags, uint32_t blk_size, uint8_t *dst) {
    const uint32_t BS = blk_size;
    uint32_t nblk = (seg_len + BS - 1) / BS;
    static const uint8_t fh[6] = {0x28,0xB5,0x2F,0xFD,0x00,0x50};
    size_t op = 0;
    memcpy(dst, fh, 6); op = 6;
    /* segment statistics */
    uint32_t lcount[256] = {0}, scount[3][64]; memset(scount, 0, sizeof(scount));
    uint32_t nseq_seg = 0;
    uint32_t *ofb = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nblk * (BS / 4));
    for (uint32_t b = 0; b < nblk; b++) {
        const uint8_t *bl = lits + (size_t)b * BS;
        const pna_seq *bs = seqs + (size_t)b * (BS / 4);
        uint32_t *bo = ofb + (size_t)b * (BS / 4);
        for (uint32_t i = 0; i < blk_nlit[b]; i++) lcount[bl[i]]++;
        block_offbase(bs, blk_nseq[b], flags, bo);
        for (uint32_t i = 0; i < blk_nseq[b]; i++) {
            scount[0][ll_code(bs[i].ll)]++; scount[1][hb32(bo[i])]++; scount[2][ml_code(bs[i].ml)]++;
        }
        nseq_seg += blk_nseq[b];
    }
    seg_tables *t = (seg_tables *)calloc(1, sizeof(seg_tables));
    if (flags & PNA_F_HUF) seg_build_huf(t, lcount);
    int seq_ok = 1;
    if (nseq_seg) {
        if (seg_build_seq_table(t, 0, scount[0], nseq_seg, 36, 8, LL_DEF, 36, 6, flags)) seq_ok = 0;
        if (seg_build_seq_table(t, 1, scount[1], nseq_seg, 32, 8, OF_DEF, 29, 5, flags)) seq_ok = 0;
        if (seg_build_seq_table(t, 2, scount[2], nseq_seg, 53, 8, ML_DEF, 53, 6, flags)) seq_ok = 0;
    }
    /* blocks, in order; have_huf / have_seq record whether a previous block of the frame carried the tables */
    int have_huf = 0, have_seq = 0;
    uint8_t *tmpblk = (uint8_t *)malloc(BS * 2 + 1024);
    uint8_t *hbody = (uint8_t *)malloc(BS * 2 + 64), *sbits = (uint8_t *)malloc((size_t)(BS / 4) * 12 + 64);
    for (uint32_t b = 0; b < nblk; b++) {
        uint32_t b0 = b * BS, bl_len = seg_len - b0 < BS ? seg_len - b0 : BS;
        const uint8_t *bl = lits + (size_t)b * BS;
        const pna_seq *bs = seqs + (size_t)b * (BS / 4);
        uint32_t nlit = blk_nlit[b], nseq = blk_nseq[b];
        int last = (b + 1 == nblk);
        uint8_t *out = tmpblk; size_t csz = 0, sb = 0; int ok = seq_ok || nseq == 0;
        int used_huf = 0;
        if (ok) {
            /* literals section */
            int rle = 0;
            if ((flags & PNA_F_HUF) && nlit >= 64) { rle = 1; for (uint32_t i = 1; i < nlit; i++) if (bl[i] != bl[0]) { rle = 0; break; } }
            size_t raw_h = nlit < 32 ? 1 : (nlit < 4096 ? 2 : 3);
            if (rle) { csz = put_raw_lit_header(out, 1, nlit); out[csz++] = bl[0]; }
            else {
                size_t hs = 0, lh = 3 + (nlit >= 1024) + (nlit >= 16384), ts = have_huf ? 0 : t->tree_len;
                if (t->huf_ok && nlit >= 64) hs = block_huf_body(t, bl, nlit, hbody);
                if (hs && lh + ts + hs < raw_h + nlit) {
                    uint64_t type = have_huf ? 3u : 2u, comp = ts + hs, h;
                    if (lh == 3) h = type | ((uint64_t)(nlit >= 256 ? 1 : 0) << 2) | ((uint64_t)nlit << 4) | (comp << 14);
                    else if (lh == 4) h = type | (2u << 2) | ((uint64_t)nlit << 4) | (comp << 18);
                    else h = type | (3u << 2) | ((uint64_t)nlit << 4) | (comp << 22);
                    for (size_t i = 0; i < lh; i++) out[csz++] = (uint8_t)(h >> (8 * i));
                    memcpy(out + csz, t->tree, ts); csz += ts;
                    memcpy(out + csz, hbody, hs); csz += hs;
                    used_huf = 1;
                } else { csz = put_raw_lit_header(out, 0, nlit); memcpy(out + csz, bl, nlit); csz += nlit; }
            }
            /* sequences section */
            if (nseq < 128) out[csz++] = (uint8_t)nseq;
            else if (nseq < 0x7F00) { out[csz++] = (uint8_t)((nseq >> 8) + 128); out[csz++] = (uint8_t)nseq; }
            else { out[csz++] = 255; out[csz++] = (uint8_t)(nseq - 0x7F00); out[csz++] = (uint8_t)((nseq - 0x7F00) >> 8); }
            if (nseq) {
                int m[3];
                for (int k = 0; k < 3; k++) m[k] = t->mode[k] == 0 ? 0 : (have_seq ? 3 : t->mode[k]);
                out[csz++] = (uint8_t)((m[0] << 6) | (m[1] << 4) | (m[2] << 2));
                if (!have_seq) for (int k = 0; k < 3; k++) { memcpy(out + csz, t->desc[k], t->desc_len[k]); csz += t->desc_len[k]; }
                sb = block_seq_bits(t, bs, ofb + (size_t)b * (BS / 4), nseq, sbits);
            }
        }
        uint32_t hdr;
        if (!ok || csz + sb >= bl_len) {
            hdr = (uint32_t)last | (0u << 1) | (bl_len << 3);
            memcpy(dst + op + 3, seg + b0, bl_len); csz = bl_len;
        } else {
            memcpy(dst + op + 3, out, csz); memcpy(dst + op + 3 + csz, sbits, sb); csz += sb;
            hdr = (uint32_t)last | (2u << 1) | ((uint32_t)csz << 3);
            if (used_huf) have_huf = 1;
            if (nseq) have_seq = 1;
        }
        dst[op] = (uint8_t)hdr; dst[op + 1] = (uint8_t)(hdr >> 8); dst[op + 2] = (uint8_t)(hdr >> 16);
        op += 3 + csz;
    }
    free(tmpblk); free(hbody); free(sbits); free(ofb); free(t);
    return op;
}

size_t pna_zstd_model_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, const pna_zstd_params *p) {
    if (cap < pna_zstd_bound(n)) return 0;
    size_t op = 0;
    if (n == 0) { static const uint8_t e[9] = {0x28,0xB5,0x2F,0xFD,0x20,0x00,0x01,0x00,0x00}; memcpy(dst, e, 9); return 9; }
    size_t table_entries = p->hash_log <= 31 ? (size_t)1 << p->hash_log : p->hash_log;
    if (table_entries < p->small_slots) table_entries = p->small_slots;
    if (table_entries < p->mid_slots) table_entries = p->mid_slots;
    uint32_t *table = (uint32_t *)malloc(sizeof(uint32_t) * table_entries);
    const uint32_t BS = pna_blk_size(p);
    uint32_t maxblk = PNA_SEG_SIZE / BS;
    pna_seq *seqs = (pna_seq *)malloc(sizeof(pna_seq) * (size_t)maxblk * (BS / 4));
    uint8_t *lits = (uint8_t *)malloc((size_t)PNA_SEG_SIZE + 8);
    uint32_t blk_nseq[PNA_SEG_SIZE / PNA_BLK_MIN], blk_nlit[PNA_SEG_SIZE / PNA_BLK_MIN];
    for (size_t s0 = 0; s0 < n; s0 += PNA_SEG_SIZE) {
        uint32_t seg_len = (uint32_t)(n - s0 < PNA_SEG_SIZE ? n - s0 : PNA_SEG_SIZE);
        const uint8_t *seg = src + s0;
        pna_zstd_params small;
        const pna_zstd_params *ps = pna_seg_params(p, seg_len, &small);           /* a short segment: the small geometry */
        memset(table, 0, sizeof(uint32_t) * table_entries);
        uint32_t b = 0;
        for (uint32_t b0 = 0; b0 < seg_len; b0 += BS, b++) {
            uint32_t bl = seg_len - b0 < BS ? seg_len - b0 : BS;
            blk_nseq[b] = pna_lz_block(seg, seg_len, b0, bl, table, ps, seqs + (size_t)b * (BS / 4),
                                       lits + (size_t)b * BS, &blk_nlit[b]);
        }
        size_t fs = pna_zstd_encode_segment(seg, seg_len, seqs, lits, blk_nseq, blk_nlit, p->flags, BS, dst + op);
        if (p->flags & PNA_F_SINGLE_FRAME) {
            /* the entry as ONE frame: the segments' blocks as they are, the 6-byte frame header in front of the first segment only, the last-block bit on
             * the entry's last block only (the product's option single_frame; SURVEY 8 a14's fallback for a reader that refuses concatenated frames) */
            if (s0 + seg_len < n) {                                      /* not the entry's last segment: clear its last block's bit */
                size_t q = 6;
                for (;;) {
                    uint32_t h = dst[op + q] | (dst[op + q + 1] << 8) | ((uint32_t)dst[op + q + 2] << 16);
                    if (h & 1) { dst[op + q] &= (uint8_t)~1u; break; }
                    q += 3 + (((h >> 1) & 3) == 1 ? 1 : (h >> 3));
                }
            }
            if (s0 > 0) { memmove(dst + op, dst + op + 6, fs - 6); fs -= 6; }
        }
        op += fs;
    }
    free(table); free(seqs); free(lits);
    return op;
}
