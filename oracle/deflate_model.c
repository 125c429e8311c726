/*
 * oracle/deflate_model.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C model ("the spec") of the MI355X zlib/deflate encoder that replaces, behind Compression::Deflate, the
 * third-party encoder the reference calls at
 *   lib/src/entry/write.rs:257-259  flate2::write::ZlibEncoder::new(writer, level)  (miniz_oxide 0.8.5, un-vendored)
 * The reference pins only DECOMPRESSED bytes here; every output of this model must inflate with an
 * independent RFC 1950/1951 decoder (Python's zlib in tests) to the input, and the HIP kernels must equal it bit for bit.
 *
 * Stream layout (one zlib stream per entry):
 *   78 9C | per 128 KiB block: one DYNAMIC-Huffman block (or stored blocks when that is not smaller), followed --
 *   except after the entry's last block -- by an empty stored block (sync flush `00 00 FF FF`) so that every block
 *   starts byte-aligned and can be produced independently | Adler-32 big-endian.   Empty entry: 78 9C 03 00 00 00 00 01
 *   (the reference's bytes, tests/golden/deflate.pna raw/empty.txt).
 * LZ stage: pna_lz_block() of zstd_model.c with max_off 32 768 and max_len 258; the hash table is reset per 1 MiB
 * segment.  Entropy stage: ONE lit/len + distance table pair per segment (statistics of all its blocks, lengths <= 15,
 * two-queue Huffman + Kraft repair as in zstd_model.c); each block repeats the table description (zero runs coded with
 * 17/18, no symbol 16).
 */
#include "zstd_model.h"
#include <string.h>
#include <stdlib.h>

typedef struct { uint8_t *p; size_t pos; uint64_t acc; int nb; } dbw;
static void dw_init(dbw *w, uint8_t *p) { w->p = p; w->pos = 0; w->acc = 0; w->nb = 0; }
static void dw_add(dbw *w, uint32_t v, int n) {
    if (!n) return;
    w->acc |= (uint64_t)(v & ((n >= 32) ? 0xFFFFFFFFu : ((1u << n) - 1))) << w->nb; w->nb += n;
    while (w->nb >= 8) { w->p[w->pos++] = (uint8_t)w->acc; w->acc >>= 8; w->nb -= 8; }
}
static size_t dw_align(dbw *w) { if (w->nb > 0) { w->p[w->pos++] = (uint8_t)w->acc; w->acc = 0; w->nb = 0; } return w->pos; }

static const uint16_t LEN_BASE[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t  LEN_EXTRA[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t DIST_BASE[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t  DIST_EXTRA[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const uint8_t  CL_ORDER[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

static int len_code(uint32_t l) { int c = 28; while (LEN_BASE[c] > l) c--; return c; }      /* 0..28 -> symbol 257+c */
static int dist_code(uint32_t d) { int c = 29; while (DIST_BASE[c] > d) c--; return c; }

/* code lengths (<= maxlen) for symbols with count > 0: two-queue Huffman, leaves win ties, symbols sorted by
 * (count asc, symbol asc); Kraft repair exactly as huf_build_lens() in zstd_model.c.  A single present symbol gets
 * length 1.  Returns the number of present symbols. */
static int build_lens(const uint32_t *count, int nsym, int maxlen, uint8_t *lens) {
    int order[288], n = 0;
    for (int s = 0; s < nsym; s++) { lens[s] = 0; if (count[s]) order[n++] = s; }
    if (n == 0) return 0;
    if (n == 1) { lens[order[0]] = 1; return 1; }
    for (int i = 1; i < n; i++) { int x = order[i], j = i - 1; while (j >= 0 && count[order[j]] > count[x]) { order[j + 1] = order[j]; j--; } order[j + 1] = x; }
    uint64_t wt[576]; int parent[576], depth[576];
    for (int i = 0; i < n; i++) wt[i] = count[order[i]];
    int lq = 0, iq = n, nn = n;
    while (nn < 2 * n - 1) {
        int a, b;
        if (lq < n && (iq >= nn || wt[lq] <= wt[iq])) a = lq++; else a = iq++;
        if (lq < n && (iq >= nn || wt[lq] <= wt[iq])) b = lq++; else b = iq++;
        wt[nn] = wt[a] + wt[b]; parent[a] = nn; parent[b] = nn; nn++;
    }
    depth[nn - 1] = 0;
    for (int i = nn - 2; i >= 0; i--) depth[i] = depth[parent[i]] + 1;
    int over = 0;
    for (int i = 0; i < n; i++) { int d = depth[i]; if (d > maxlen) { d = maxlen; over = 1; } lens[order[i]] = (uint8_t)d; }
    if (!over) return n;
    int32_t K = 0;
    for (int i = 0; i < n; i++) K += 1 << (maxlen - lens[order[i]]);
    int32_t debt = K - (1 << maxlen);
    while (debt > 0) {
        int pick = -1, bl = 0;
        for (int i = 0; i < n; i++) { int l = lens[order[i]]; if (l < maxlen && l > bl) { bl = l; pick = i; } }
        lens[order[pick]]++; debt -= 1 << (maxlen - 1 - bl);
    }
    while (debt < 0) {
        int pick = -1, bl = 99; int32_t slack = -debt;
        for (int i = n - 1; i >= 0; i--) { int l = lens[order[i]]; if (l > 1 && (1 << (maxlen - l)) <= slack && l < bl) { bl = l; pick = i; } }
        if (pick < 0) break;   /* an incomplete code is still a valid prefix code for deflate decoders only when complete; callers check */
        lens[order[pick]]--; debt += 1 << (maxlen - bl);
    }
    return n;
}

/* canonical deflate codes (RFC 1951 3.2.2), returned bit-reversed so they can be added LSB first */
static void assign_codes(const uint8_t *lens, int nsym, uint16_t *codes) {
    int bl_count[16] = {0}, next[16];
    for (int s = 0; s < nsym; s++) bl_count[lens[s]]++;
    bl_count[0] = 0;
    int code = 0;
    for (int b = 1; b <= 15; b++) { code = (code + bl_count[b - 1]) << 1; next[b] = code; }
    for (int s = 0; s < nsym; s++) {
        int l = lens[s]; codes[s] = 0;
        if (!l) continue;
        int c = next[l]++, r = 0;
        for (int i = 0; i < l; i++) r |= ((c >> i) & 1) << (l - 1 - i);
        codes[s] = (uint16_t)r;
    }
}

typedef struct {
    uint8_t  ll_len[288], d_len[32];
    uint16_t ll_code[288], d_code[32];
    uint8_t  hdr[400]; uint32_t hdr_bits;     /* HLIT HDIST HCLEN + code-length code + coded lengths */
} dtables;

static void build_tables(dtables *t, const uint32_t *llc, const uint32_t *dc) {
    build_lens(llc, 286, 15, t->ll_len);
    build_lens(dc, 30, 15, t->d_len);
    assign_codes(t->ll_len, 286, t->ll_code);
    assign_codes(t->d_len, 30, t->d_code);
    int nll = 286; while (nll > 257 && t->ll_len[nll - 1] == 0) nll--;
    int nd = 30; while (nd > 1 && t->d_len[nd - 1] == 0) nd--;
    /* code-length sequence with zero runs (17: 3..10, 18: 11..138) */
    uint8_t seq[320]; uint8_t sym[320], ext[320]; int ns = 0, n = 0;
    for (int i = 0; i < nll; i++) seq[n++] = t->ll_len[i];
    for (int i = 0; i < nd; i++) seq[n++] = t->d_len[i];
    for (int i = 0; i < n;) {
        if (seq[i] == 0) {
            int z = 1; while (i + z < n && seq[i + z] == 0 && z < 138) z++;
            if (z >= 11) { sym[ns] = 18; ext[ns++] = (uint8_t)(z - 11); i += z; continue; }
            if (z >= 3) { sym[ns] = 17; ext[ns++] = (uint8_t)(z - 3); i += z; continue; }
        }
        sym[ns] = seq[i]; ext[ns++] = 0; i++;
    }
    uint32_t clc[19] = {0}; uint8_t cl_len[19]; uint16_t cl_code[19];
    for (int i = 0; i < ns; i++) clc[sym[i]]++;
    if (build_lens(clc, 19, 7, cl_len) == 1) {            /* the code-length code must be complete: add a dummy second code */
        for (int k = 0; k < 19; k++) if (!cl_len[k]) { cl_len[k] = 1; break; }
    }
    assign_codes(cl_len, 19, cl_code);
    int ncl = 19; while (ncl > 4 && cl_len[CL_ORDER[ncl - 1]] == 0) ncl--;
    dbw w; dw_init(&w, t->hdr);
    dw_add(&w, (uint32_t)(nll - 257), 5); dw_add(&w, (uint32_t)(nd - 1), 5); dw_add(&w, (uint32_t)(ncl - 4), 4);
    for (int i = 0; i < ncl; i++) dw_add(&w, cl_len[CL_ORDER[i]], 3);
    for (int i = 0; i < ns; i++) {
        dw_add(&w, cl_code[sym[i]], cl_len[sym[i]]);
        if (sym[i] == 17) dw_add(&w, ext[i], 3);
        if (sym[i] == 18) dw_add(&w, ext[i], 7);
    }
    t->hdr_bits = (uint32_t)(w.pos * 8 + (size_t)w.nb);
    dw_align(&w);
}

static uint32_t adler32_update(uint32_t adler, const uint8_t *p, size_t n) {
    uint32_t a = adler & 0xFFFF, b = adler >> 16;
    while (n) { size_t k = n < 5552 ? n : 5552; n -= k; while (k--) { a += *p++; b += a; } a %= 65521; b %= 65521; }
    return (b << 16) | a;
}

size_t pna_deflate_bound(size_t n) {
    size_t blks = (n + PNA_BLK_MIN - 1) / PNA_BLK_MIN; if (!blks) blks = 1;      /* (the smallest block size a parameter set may name) */
    return n + blks * 10 + (n >> 17) * 16 + 64;
}

void pna_deflate_default_params(pna_zstd_params *p) {
    pna_zstd_default_params(p);
    p->max_off = 32768; p->max_len = 258; p->flags = PNA_F_LAZY | PNA_F_LAZY2 | PNA_F_LAZY3;
    p->hash_log = 24512; p->near_off = 56064; p->tab3 = 0; p->far_slots = 0;      /* the 64 KiB-window geometry: the whole look-back lies in the window; 32-bit table entries */
}

size_t pna_deflate_model_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, const pna_zstd_params *p) {
    if (cap < pna_deflate_bound(n)) return 0;
    /* level 0 = Compression::none() (lib/src/compress/deflate.rs:89-101): stored blocks only -- header 78 01 (FLEVEL 0), per block of the block size
     * the stored pieces (<= 65 535 bytes each), the sync flush between blocks as everywhere, an empty entry = one empty final stored block */
    const int stored_only = (p->flags & PNA_F_STORED) != 0;
    if (n == 0 && stored_only) { static const uint8_t e[11] = {0x78,0x01,0x01,0x00,0x00,0xFF,0xFF,0x00,0x00,0x00,0x01}; memcpy(dst, e, 11); return 11; }
    if (n == 0) { static const uint8_t e[8] = {0x78,0x9C,0x03,0x00,0x00,0x00,0x00,0x01}; memcpy(dst, e, 8); return 8; }
    size_t op = 0;
    dst[op++] = 0x78; dst[op++] = stored_only ? 0x01 : 0x9C;
    size_t table_entries = p->hash_log <= 31 ? (size_t)1 << p->hash_log : p->hash_log;
    if (table_entries < p->small_slots) table_entries = p->small_slots;
    if (table_entries < p->mid_slots) table_entries = p->mid_slots;
    uint32_t *table = (uint32_t *)malloc(sizeof(uint32_t) * table_entries);
    const uint32_t BS = pna_blk_size(p);
    uint32_t maxblk = PNA_SEG_SIZE / BS;
    pna_seq *seqs = (pna_seq *)malloc(sizeof(pna_seq) * (size_t)maxblk * (BS / 4));
    uint8_t *lits = (uint8_t *)malloc((size_t)PNA_SEG_SIZE + 8);
    uint8_t *tmp = (uint8_t *)malloc(BS * 2 + 4096);
    dtables *t = (dtables *)malloc(sizeof(dtables));
    uint32_t blk_nseq[PNA_SEG_SIZE / PNA_BLK_MIN], blk_nlit[PNA_SEG_SIZE / PNA_BLK_MIN];
    for (size_t s0 = 0; s0 < n; s0 += PNA_SEG_SIZE) {
        uint32_t seg_len = (uint32_t)(n - s0 < PNA_SEG_SIZE ? n - s0 : PNA_SEG_SIZE);
        const uint8_t *seg = src + s0;
        pna_zstd_params small;
        const pna_zstd_params *ps = pna_seg_params(p, seg_len, &small);           /* a short segment: the small geometry (zstd_model.h) */
        memset(table, 0, sizeof(uint32_t) * table_entries);
        uint32_t nb = 0;
        for (uint32_t b0 = 0; b0 < seg_len; b0 += BS, nb++) {
            uint32_t bl = seg_len - b0 < BS ? seg_len - b0 : BS;
            blk_nseq[nb] = pna_lz_block(seg, seg_len, b0, bl, table, ps, seqs + (size_t)nb * (BS / 4),
                                        lits + (size_t)nb * BS, &blk_nlit[nb]);
        }
        /* segment statistics */
        uint32_t llc[288] = {0}, dc[32] = {0};
        for (uint32_t b = 0; b < nb; b++) {
            const uint8_t *bl = lits + (size_t)b * BS; const pna_seq *bs = seqs + (size_t)b * (BS / 4);
            for (uint32_t i = 0; i < blk_nlit[b]; i++) llc[bl[i]]++;
            for (uint32_t i = 0; i < blk_nseq[b]; i++) { llc[257 + len_code(bs[i].ml)]++; dc[dist_code(bs[i].off)]++; }
            llc[256]++;
        }
        build_tables(t, llc, dc);
        for (uint32_t b = 0; b < nb; b++) {
            uint32_t b0 = b * BS, bl_len = seg_len - b0 < BS ? seg_len - b0 : BS;
            const uint8_t *bl = lits + (size_t)b * BS; const pna_seq *bs = seqs + (size_t)b * (BS / 4);
            int last = (s0 + b0 + bl_len >= n);
            dbw w; dw_init(&w, tmp);
            dw_add(&w, (uint32_t)last, 1); dw_add(&w, 2, 2);
            { uint32_t hb = t->hdr_bits, i = 0; while (hb >= 8) { dw_add(&w, t->hdr[i++], 8); hb -= 8; } if (hb) dw_add(&w, t->hdr[i], (int)hb); }
            uint32_t li = 0;
            for (uint32_t i = 0; i < blk_nseq[b]; i++) {
                for (uint32_t k = 0; k < bs[i].ll; k++, li++) dw_add(&w, t->ll_code[bl[li]], t->ll_len[bl[li]]);
                int lc = len_code(bs[i].ml), dcd = dist_code(bs[i].off);
                dw_add(&w, t->ll_code[257 + lc], t->ll_len[257 + lc]); dw_add(&w, bs[i].ml - LEN_BASE[lc], LEN_EXTRA[lc]);
                dw_add(&w, t->d_code[dcd], t->d_len[dcd]); dw_add(&w, bs[i].off - DIST_BASE[dcd], DIST_EXTRA[dcd]);
            }
            for (; li < blk_nlit[b]; li++) dw_add(&w, t->ll_code[bl[li]], t->ll_len[bl[li]]);
            dw_add(&w, t->ll_code[256], t->ll_len[256]);
            if (!last) dw_add(&w, 0, 3);                   /* header of the empty stored block (sync flush), then align */
            size_t dyn = dw_align(&w);
            size_t stored = (size_t)bl_len + 5 * (((size_t)bl_len + 65534) / 65535);
            if (stored_only || dyn >= stored || dyn > BS) {       /* second clause: the device's per-block scratch is one block */
                for (uint32_t o = 0; o < bl_len; o += 65535) {
                    uint32_t k = bl_len - o < 65535 ? bl_len - o : 65535;
                    dst[op++] = (uint8_t)((last && o + k >= bl_len) ? 1 : 0);
                    dst[op++] = (uint8_t)k; dst[op++] = (uint8_t)(k >> 8); dst[op++] = (uint8_t)~k; dst[op++] = (uint8_t)(~k >> 8);
                    memcpy(dst + op, seg + b0 + o, k); op += k;
                }
                if (!last) dst[op++] = 0x00;               /* stored data ends byte-aligned: 000 + padding */
            } else { memcpy(dst + op, tmp, dyn); op += dyn; }
            if (!last) { static const uint8_t sync[4] = {0x00, 0x00, 0xFF, 0xFF}; memcpy(dst + op, sync, 4); op += 4; }
        }
    }
    uint32_t ad = adler32_update(1, src, n);
    dst[op++] = (uint8_t)(ad >> 24); dst[op++] = (uint8_t)(ad >> 16); dst[op++] = (uint8_t)(ad >> 8); dst[op++] = (uint8_t)ad;
    free(table); free(seqs); free(lits); free(tmp); free(t);
    return op;
}
