/* experiments/ratio/lzexp.c -- scratch copy of the LZ stage of oracle/zstd_model.c with extra knobs, to explore ratio (round 3).
 * Output metric: estimated compressed size from an entropy estimate (sum of -log2 p over literal bytes and LL/ML/OF codes + extra bits), per segment. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef struct {
    uint32_t entries, min_match, tile, max_off, cap1, lookahead, lazy, lazy2, region, ins_mod, back_cap, rounds, look_mod, ways, hash_bytes, blk;
    uint32_t rep;      /* estimate repeat-offset codes */
    uint32_t prefer_near; uint32_t gran; uint32_t lazy3, far_min, far_thr, min2, thr2; /* gran: entries hold 1 << gran byte granules, the position inside is found by comparing the hashed bytes */
} P;
static uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t hashf(const uint8_t *p, const P *pr) {
    uint32_t lo = rd32(p), hi = 0;
    if (pr->hash_bytes >= 5) hi = p[4];
    if (pr->hash_bytes >= 6) hi |= (uint32_t)p[5] << 8;
    if (pr->hash_bytes >= 7) hi |= (uint32_t)p[6] << 16;
    if (pr->hash_bytes >= 8) hi |= (uint32_t)p[7] << 24;
    uint32_t h = lo * 0x9E3779B1u + hi * 0x85EBCA6Bu;
    return (uint32_t)(((uint64_t)h * pr->entries) >> 32);
}
typedef struct { uint32_t ll, ml, off; } seq;
static int hb32(uint32_t v) { int r = -1; while (v) { v >>= 1; r++; } return r; }

/* one block; table: ways x entries */
static uint64_t g_cand[8];   /* candidates looked at: total, offset > 8K, > 16K, > 23K, > 39K, > 56K */
static uint32_t lz_block(const uint8_t *seg, uint32_t seg_len, uint32_t blk_start, uint32_t blk_len, uint32_t *table, const P *p, seq *seqs, uint8_t *lits, uint32_t *nlit_out) {
    uint32_t blk_end = blk_start + blk_len, nseq = 0, nlit = 0, next_free = blk_start, lit_start = blk_start, T = p->tile;
    uint32_t *cand = malloc(4 * T), *cand0 = malloc(4 * T); uint16_t *len = malloc(2 * (T + 1)), *len0 = malloc(2 * (T + 1));
    uint8_t *back = calloc(T + 1, 1), *back0 = malloc(T + 1);
    uint32_t *mq = malloc(16 * (T + 1)), *ml = mq + T + 1, *mc = ml + T + 1, *mr = mc + T + 1;
    const uint32_t W = p->ways ? p->ways : 1, E = p->entries;
    for (uint32_t t0 = blk_start; t0 < blk_end; t0 += T) {
        uint32_t t1 = t0 + T < blk_end ? t0 + T : blk_end;
        /* L + M: candidates from each way, best kept */
        for (uint32_t q = t0; q < t1; q++) {
            uint32_t bl = 0, bc = 0, bbk = 0;
            if (q + 8 <= seg_len && (p->look_mod <= 1 || q % p->look_mod == 0)) {
                uint32_t h = hashf(seg + q, p);
                for (uint32_t w = 0; w < W; w++) {
                    uint32_t c1 = table[w * E + h], l = 0, bk = 0;
                    if (p->gran && c1) {                                  /* granule -> the highest (even) position in it whose hashed bytes equal those at q */
                        uint32_t g0 = (c1 - 1) << p->gran, found = 0;
                        for (int32_t pp = (int32_t)(g0 + (1u << p->gran) - 1); pp >= (int32_t)g0; pp--) {
                            if (p->ins_mod > 1 && (pp % p->ins_mod)) continue;
                            if ((uint32_t)pp >= t0 || (uint32_t)pp + 8 > seg_len) continue;
                            if (!memcmp(seg + pp, seg + q, p->hash_bytes)) { found = (uint32_t)pp + 1; break; }
                        }
                        c1 = found;
                    }
                    if (c1 > 8 && q - (c1 - 1) <= p->max_off) {
                        uint32_t c = c1 - 1, lim = blk_end - q;
                        { uint32_t o = q - c; g_cand[0]++; if (o > 8192) g_cand[1]++; if (o > 16384) g_cand[2]++; if (o > 23000) g_cand[3]++; if (o > 39000) g_cand[4]++; if (o > 56064) g_cand[5]++; }
                        if (lim > p->cap1) lim = p->cap1;
                        while (l < lim && seg[q + l] == seg[c + l]) l++;
                        if (l < p->min_match) l = 0;
                        if (l) while (bk < p->back_cap && seg[q - 1 - bk] == seg[c - 1 - bk]) bk++;
                    }
                    if (l > bl || (l == bl && l && !p->prefer_near && 0)) { bl = l; bc = c1; bbk = bk; }
                }
            }
            cand[q - t0] = bc; len[q - t0] = (uint16_t)bl; back[q - t0] = (uint8_t)bbk;
        }
        len[t1 - t0] = 0;
        /* I: way 0 = newest of this tile (max), way 1 = what way 0 held before this tile */
        for (uint32_t q = t0; q < t1; q++)
            if (q + 8 <= seg_len && q % (p->ins_mod ? p->ins_mod : 1) == 0) {
                uint32_t h = hashf(seg + q, p);
                uint32_t old = table[h];
                uint32_t nv = p->gran ? (q >> p->gran) + 1 : q + 1, tl = p->gran ? (t0 >> p->gran) + 1 : t0 + 1;
                if (old < nv) { if (W > 1 && old < tl && old) table[E + h] = old; table[h] = nv; }
            }
        /* A */
        for (uint32_t rr = p->rounds; rr; rr >>= 4) {
            const uint32_t sft = rr & 15;
            memcpy(len0, len, 2 * (T + 1)); memcpy(cand0, cand, 4 * T); memcpy(back0, back, T + 1);
            for (uint32_t q = t0; q + sft < t1; q++) {
                const uint32_t j = q + sft - t0;
                if ((q & 63) + sft > 63 || len0[j] < p->min_match || back0[j] < sft || len0[j] + sft <= len0[q - t0]) continue;
                len[q - t0] = (uint16_t)(len0[j] + sft); cand[q - t0] = cand0[j] - sft; back[q - t0] = (uint8_t)(back0[j] - sft);
            }
        }
        /* P */
        uint32_t ext_lim = t1 + p->lookahead < blk_end ? t1 + p->lookahead : blk_end, R = p->region ? p->region : T, nm = 0;
        for (uint32_t r0 = t0; r0 < t1; r0 += R) {
            uint32_t r1 = r0 + R < t1 ? r0 + R : t1;
            for (uint32_t q = (next_free > r0 ? next_free : r0); q < r1;) {
                uint32_t l = len[q - t0];
                int take = l >= p->min_match;
                if (take && p->lazy && (q & 63) != 63 && q + 1 < t1 && len[q + 1 - t0] > l) take = 0;
                if (take && p->lazy && p->lazy2 && (q & 63) < 62 && q + 2 < t1 && len[q + 2 - t0] > l + 1) take = 0;
                if (take && p->lazy && p->lazy3 && (q & 63) < 61 && q + 3 < t1 && len[q + 3 - t0] > l + 2) take = 0;
                if (take && p->far_min && (q - (cand[q - t0] - 1)) > p->far_thr && l < p->far_min) take = 0;
                if (take && p->min2 && (q - (cand[q - t0] - 1)) > p->thr2 && l < p->min2) take = 0;
                if (!take) { q++; continue; }
                uint32_t c = cand[q - t0] - 1;
                if (l >= p->cap1) { uint32_t el = ext_lim; while (q + l < el && seg[q + l] == seg[c + l]) l++; }
                mq[nm] = q; ml[nm] = l; mc[nm] = c; mr[nm] = r1; nm++;
                q += l;
            }
        }
        /* F */
        for (uint32_t i = 0; i < nm; i++) {
            uint32_t q = mq[i], l = ml[i], c = mc[i];
            if (q + l <= next_free || mr[i] <= next_free) continue;
            if (q < next_free) { uint32_t r = q + l - next_free; if (r < 3) continue; c += next_free - q; q = next_free; l = r; }
            seqs[nseq].ll = q - lit_start; seqs[nseq].ml = l; seqs[nseq].off = q - c; nseq++;
            memcpy(lits + nlit, seg + lit_start, q - lit_start); nlit += q - lit_start;
            lit_start = q + l; next_free = lit_start;
        }
    }
    memcpy(lits + nlit, seg + lit_start, blk_end - lit_start); nlit += blk_end - lit_start;
    free(cand); free(cand0); free(len); free(len0); free(back); free(back0); free(mq);
    *nlit_out = nlit; return nseq;
}
static const uint8_t LLC[64] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24};
static const uint8_t LLB[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static int ll_code(uint32_t v) { return v < 64 ? LLC[v] : hb32(v) + 19; }
static int ml_code(uint32_t ml) { uint32_t b = ml - 3; static const uint8_t MLC[128] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42}; return b < 128 ? MLC[b] : hb32(b) + 36; }
static const uint8_t MLB[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
static double ent(const uint32_t *c, int n) { double t = 0, e = 0; for (int i = 0; i < n; i++) t += c[i]; for (int i = 0; i < n; i++) if (c[i]) e -= c[i] * log2(c[i] / t); return e; }

int main(int argc, char **argv) {
    P p = {24512, 6, 4096, 1u << 20, 32, 1024, 1, 0, 256, 2, 3, 0x21, 1, 1, 6, 1u << 17, 0, 1, 0, 0, 0, 0, 0, 0};
    const char *files = NULL;
    for (int i = 1; i < argc; i++) {
        char *eq = strchr(argv[i], '=');
        if (!eq) { files = argv[i]; continue; }
        *eq = 0; uint32_t v = (uint32_t)strtoul(eq + 1, NULL, 0);
#define K(n) if (!strcmp(argv[i], #n)) p.n = v;
        K(entries) K(min_match) K(tile) K(max_off) K(cap1) K(lookahead) K(lazy) K(lazy2) K(region) K(ins_mod) K(back_cap) K(rounds) K(look_mod) K(ways) K(hash_bytes) K(blk) K(rep) K(gran) K(lazy3) K(far_min) K(far_thr) K(min2) K(thr2)
    }
    FILE *f = fopen(files, "rb"); if (!f) { perror("open"); return 1; }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *src = malloc(n); if (fread(src, 1, n, f) != (size_t)n) return 1; fclose(f);
    uint32_t *table = malloc(4 * (size_t)p.entries * (p.ways ? p.ways : 1));
    seq *seqs = malloc(sizeof(seq) * (1 << 18)); uint8_t *lits = malloc((1 << 20) + 8);
    double bits_total = 0; uint64_t nseq_total = 0, nlit_total = 0, mlsum = 0;
    for (long s0 = 0; s0 < n; s0 += 1 << 20) {
        uint32_t seg_len = (uint32_t)(n - s0 < (1 << 20) ? n - s0 : (1 << 20));
        const uint8_t *seg = src + s0;
        memset(table, 0, 4 * (size_t)p.entries * (p.ways ? p.ways : 1));
        uint32_t lc[256] = {0}, llc[36] = {0}, mlc[53] = {0}, ofc[32] = {0}; double extra = 0; uint32_t ns_seg = 0;
        for (uint32_t b0 = 0; b0 < seg_len; b0 += p.blk) {
            uint32_t bl = seg_len - b0 < p.blk ? seg_len - b0 : p.blk, nl;
            uint32_t ns = lz_block(seg, seg_len, b0, bl, table, &p, seqs, lits, &nl);
            for (uint32_t i = 0; i < nl; i++) lc[lits[i]]++;
            uint32_t rep[3] = {1, 4, 8};
            for (uint32_t i = 0; i < ns; i++) {
                int a = ll_code(seqs[i].ll), m = ml_code(seqs[i].ml);
                llc[a]++; mlc[m]++; extra += LLB[a] + MLB[m]; mlsum += seqs[i].ml;
                uint32_t ofv = seqs[i].off + 3;
                if (p.rep) {
                    uint32_t o = seqs[i].off; int ll0 = seqs[i].ll == 0;
                    if (!ll0 && o == rep[0]) ofv = 1; else if (o == rep[1]) { ofv = ll0 ? 1 : 2; } else if (o == rep[2]) { ofv = ll0 ? 2 : 3; }
                    else if (ll0 && o == rep[0] - 1 && rep[0] > 1) ofv = 3;
                    if (ofv > 3) { rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = o; }
                    else if (o == rep[1]) { rep[1] = rep[0]; rep[0] = o; } else if (o == rep[2] || ofv == 3) { rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = o; }
                }
                int oc = hb32(ofv); ofc[oc]++; extra += oc;
            }
            nseq_total += ns; nlit_total += nl; ns_seg += ns;
        }
        (void)ns_seg;
        bits_total += ent(lc, 256) + ent(llc, 36) + ent(mlc, 53) + ent(ofc, 32) + extra + 8 * 200;
    }
    printf("candidates %.1f%% of positions; of them beyond 8K %.1f%%, 16K %.1f%%, 23K %.1f%%, 39K %.1f%%, 56K %.1f%%\n", 100.0 * g_cand[0] / n, 100.0 * g_cand[1] / g_cand[0], 100.0 * g_cand[2] / g_cand[0], 100.0 * g_cand[3] / g_cand[0], 100.0 * g_cand[4] / g_cand[0], 100.0 * g_cand[5] / g_cand[0]);
    printf("est ratio %.4f  (est bytes %.0f of %ld; %llu seqs, %llu lits, mean ml %.2f)\n", n / (bits_total / 8), bits_total / 8, n, (unsigned long long)nseq_total, (unsigned long long)nlit_total, nseq_total ? (double)mlsum / nseq_total : 0.0);
    return 0;
}
