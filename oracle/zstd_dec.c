/*
 * oracle/zstd_dec.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Independent Zstandard decoder written from RFC 8878 (the published format of
 * the `zstd` 0.13.3 -> zstd-sys 2.0.14+zstd.1.5.7 dependency, Cargo.lock:3547-3572,
 * which is NOT vendored under /root/reference).  It plays the role of the
 * reference's read side for Compression::ZStandard:
 *   lib/src/entry/read.rs:171-190  decompress_reader() -> zstd::Decoder (multi-frame)
 * Pinned by: decoding every zstd FDAT/SDAT payload of the reference's golden
 * fixtures (tests/golden/, copied from resources/test) and comparing with
 * resources/test/raw, and by agreeing with the system libzstd on random inputs
 * (tests/test_oracle_zstd_dec.py).
 *
 * Plain C, no dependencies.  All frames of a concatenation are decoded
 * (zstd-rs `Decoder` is multi-frame by default).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define ZD_ERR_TRUNC      (-1)
#define ZD_ERR_MAGIC      (-2)
#define ZD_ERR_RESERVED   (-3)
#define ZD_ERR_DSTFULL    (-4)
#define ZD_ERR_CORRUPT    (-5)
#define ZD_ERR_DICT       (-6)
#define ZD_ERR_WINDOW     (-7)

#define MAX_LL 35
#define MAX_ML 52
#define MAX_OF 31
#define HUF_MAXBITS 11

typedef struct {
    uint8_t  sym[512];
    uint8_t  nbits[512];
    uint16_t base[512];
    int      alog;
} fse_dtable;

typedef struct {
    uint8_t sym[1 << HUF_MAXBITS];
    uint8_t nbits[1 << HUF_MAXBITS];
    int     maxbits;
    int     valid;
} huf_dtable;

typedef struct {
    fse_dtable ll, of, ml;
    int ll_ok, of_ok, ml_ok;
    huf_dtable huf;
    uint64_t rep[3];
} frame_ctx;

static int highbit32(uint32_t v) { int r = -1; while (v) { v >>= 1; r++; } return r; }

/* ---------- forward (LSB-first) bit reader for FSE table descriptions ---------- */
typedef struct { const uint8_t *p; size_t len; size_t bitpos; } fbits;
static uint32_t fb_peek(const fbits *b, int n) {
    uint64_t v = 0; size_t byte = b->bitpos >> 3; int sh = (int)(b->bitpos & 7);
    for (int i = 0; i < 8; i++) if (byte + i < b->len) v |= (uint64_t)b->p[byte + i] << (8 * i);
    return (uint32_t)((v >> sh) & ((1ull << n) - 1));
}

/* ---------- backward bit reader (sequences, huffman streams, fse weights) ---------- */
typedef struct { const uint8_t *p; int64_t off; /* bits remaining below the cursor */ } bbits;
static int bb_init(bbits *b, const uint8_t *p, size_t len) {
    if (len == 0) return ZD_ERR_CORRUPT;
    uint8_t last = p[len - 1];
    if (last == 0) return ZD_ERR_CORRUPT;
    b->p = p; b->off = (int64_t)len * 8 - (8 - highbit32(last));
    return 0;
}
static uint64_t read_bits_le(const uint8_t *p, int nbits, int64_t off) {
    /* nbits <= 57 */
    uint64_t v = 0; int64_t byte = off >> 3; int sh = (int)(off & 7);
    int need = (nbits + sh + 7) >> 3;
    for (int i = 0; i < need; i++) v |= (uint64_t)p[byte + i] << (8 * i);
    return (v >> sh) & ((nbits == 64) ? ~0ull : ((1ull << nbits) - 1));
}
static uint64_t bb_read(bbits *b, int n) {
    if (n == 0) return 0;
    b->off -= n;
    int64_t aoff = b->off; int abits = n;
    if (b->off < 0) { abits += (int)b->off; aoff = 0; }
    uint64_t r = abits > 0 ? read_bits_le(b->p, abits, aoff) : 0;
    if (b->off < 0) r = (-b->off >= 64) ? 0 : (r << (-b->off));
    return r;
}

/* ---------- FSE ---------- */
static int fse_build(fse_dtable *dt, const int16_t *norm, int nsym, int alog) {
    int size = 1 << alog, high = size - 1;
    uint16_t next[256];
    if (alog > 9) return ZD_ERR_CORRUPT;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] == -1) { dt->sym[high--] = (uint8_t)s; next[s] = 1; }
        else next[s] = (uint16_t)norm[s];
    }
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s < nsym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            dt->sym[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos > high);
        }
    }
    if (pos != 0) return ZD_ERR_CORRUPT;
    for (int u = 0; u < size; u++) {
        int s = dt->sym[u];
        uint32_t ns = next[s]++;
        int nb = alog - highbit32(ns);
        dt->nbits[u] = (uint8_t)nb;
        dt->base[u] = (uint16_t)((ns << nb) - size);
    }
    dt->alog = alog;
    return 0;
}

/* parse an FSE table description; returns bytes consumed or <0 */
static long fse_read_desc(const uint8_t *src, size_t len, int16_t *norm, int *nsym_out,
                          int *alog_out, int max_sym, int max_alog) {
    if (len < 1) return ZD_ERR_TRUNC;
    fbits b = { src, len, 0 };
    int alog = (int)fb_peek(&b, 4) + 5; b.bitpos += 4;
    if (alog > max_alog) return ZD_ERR_CORRUPT;
    int remaining = (1 << alog) + 1, threshold = 1 << alog, nbits = alog + 1, sym = 0;
    memset(norm, 0, sizeof(int16_t) * (size_t)(max_sym + 1));
    while (remaining > 1 && sym <= max_sym) {
        int max = (2 * threshold - 1) - remaining;
        int count;
        uint32_t v = fb_peek(&b, nbits);
        if ((int)(v & (uint32_t)(threshold - 1)) < max) {
            count = (int)(v & (uint32_t)(threshold - 1)); b.bitpos += (size_t)(nbits - 1);
        } else {
            count = (int)(v & (uint32_t)(2 * threshold - 1));
            if (count >= threshold) count -= max;
            b.bitpos += (size_t)nbits;
        }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        if (count == 0) {
            for (;;) {
                uint32_t rep = fb_peek(&b, 2); b.bitpos += 2;
                for (uint32_t i = 0; i < rep && sym <= max_sym; i++) norm[sym++] = 0;
                if (rep != 3) break;
            }
        }
        while (remaining < threshold) { nbits--; threshold >>= 1; }
        if ((b.bitpos + 7) / 8 > len) return ZD_ERR_TRUNC;
    }
    if (remaining != 1 || sym > max_sym + 1) return ZD_ERR_CORRUPT;
    *nsym_out = sym; *alog_out = alog;
    return (long)((b.bitpos + 7) / 8);
}

/* ---------- Huffman ---------- */
static int huf_build(huf_dtable *h, const uint8_t *weights, int nweights) {
    /* weights[0..nweights) explicit; the last one is implied */
    uint32_t total = 0;
    uint8_t w[256];
    if (nweights < 1 || nweights > 255) return ZD_ERR_CORRUPT;
    for (int i = 0; i < nweights; i++) {
        if (weights[i] > HUF_MAXBITS) return ZD_ERR_CORRUPT;
        w[i] = weights[i];
        if (w[i]) total += 1u << (w[i] - 1);
    }
    if (total == 0) return ZD_ERR_CORRUPT;
    int maxbits = highbit32(total) + 1;
    if (maxbits > HUF_MAXBITS) return ZD_ERR_CORRUPT;
    uint32_t rest = (1u << maxbits) - total;
    if (rest == 0 || (rest & (rest - 1))) return ZD_ERR_CORRUPT;
    w[nweights] = (uint8_t)(highbit32(rest) + 1);
    int nsym = nweights + 1;
    uint32_t pos = 0;
    for (int wt = 1; wt <= maxbits; wt++) {
        for (int s = 0; s < nsym; s++) {
            if (w[s] != wt) continue;
            uint32_t n = 1u << (wt - 1);
            for (uint32_t i = 0; i < n; i++) {
                h->sym[pos + i] = (uint8_t)s;
                h->nbits[pos + i] = (uint8_t)(maxbits + 1 - wt);
            }
            pos += n;
        }
    }
    if (pos != (1u << maxbits)) return ZD_ERR_CORRUPT;
    h->maxbits = maxbits; h->valid = 1;
    return 0;
}

static long huf_read_tree(huf_dtable *h, const uint8_t *src, size_t len) {
    if (len < 1) return ZD_ERR_TRUNC;
    uint8_t weights[256];
    int hb = src[0], nw;
    long used;
    if (hb >= 128) {
        nw = hb - 127;
        size_t bytes = (size_t)(nw + 1) / 2;
        if (1 + bytes > len) return ZD_ERR_TRUNC;
        for (int i = 0; i < nw; i++) {
            uint8_t b = src[1 + i / 2];
            weights[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        used = 1 + (long)bytes;
    } else {
        size_t csize = (size_t)hb;
        if (csize == 0 || 1 + csize > len) return ZD_ERR_TRUNC;
        int16_t norm[256]; int nsym, alog;
        long dl = fse_read_desc(src + 1, csize, norm, &nsym, &alog, 255, 6);
        if (dl < 0) return dl;
        fse_dtable dt;
        int rc = fse_build(&dt, norm, nsym, alog);
        if (rc) return rc;
        if ((size_t)dl >= csize) return ZD_ERR_CORRUPT;
        bbits b; rc = bb_init(&b, src + 1 + dl, csize - (size_t)dl);
        if (rc) return rc;
        uint32_t s1 = (uint32_t)bb_read(&b, alog), s2 = (uint32_t)bb_read(&b, alog);
        nw = 0;
        for (;;) {
            if (nw >= 255) return ZD_ERR_CORRUPT;
            weights[nw++] = dt.sym[s1];
            s1 = dt.base[s1] + (uint32_t)bb_read(&b, dt.nbits[s1]);
            if (b.off < 0) { if (nw >= 255) return ZD_ERR_CORRUPT; weights[nw++] = dt.sym[s2]; break; }
            if (nw >= 255) return ZD_ERR_CORRUPT;
            weights[nw++] = dt.sym[s2];
            s2 = dt.base[s2] + (uint32_t)bb_read(&b, dt.nbits[s2]);
            if (b.off < 0) { if (nw >= 255) return ZD_ERR_CORRUPT; weights[nw++] = dt.sym[s1]; break; }
        }
        used = 1 + (long)csize;
    }
    int rc = huf_build(h, weights, nw);
    if (rc) return rc;
    return used;
}

static int huf_decode_stream(const huf_dtable *h, const uint8_t *src, size_t len,
                             uint8_t *dst, size_t n) {
    bbits b; int rc = bb_init(&b, src, len);
    if (rc) return rc;
    int mb = h->maxbits;
    /* state = next mb bits */
    for (size_t i = 0; i < n; i++) {
        /* peek mb bits (zero-padded past the start) */
        bbits t = b;
        uint32_t idx = (uint32_t)bb_read(&t, mb);
        dst[i] = h->sym[idx];
        b.off -= h->nbits[idx];
        if (b.off < 0) return ZD_ERR_CORRUPT;
    }
    if (b.off != 0) return ZD_ERR_CORRUPT;
    return 0;
}

/* ---------- sequences tables ---------- */
static const int16_t LL_DEF[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const int16_t ML_DEF[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
static const int16_t OF_DEF[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
static const uint32_t LL_BASE[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
static const uint8_t  LL_BITS[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const uint32_t ML_BASE[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
static const uint8_t  ML_BITS[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};

static long seq_table(fse_dtable *dt, int *ok, int mode, const uint8_t *src, size_t len,
                      const int16_t *def, int def_n, int def_alog, int max_sym, int max_alog) {
    if (mode == 0) { int rc = fse_build(dt, def, def_n, def_alog); if (rc) return rc; *ok = 1; return 0; }
    if (mode == 1) {
        if (len < 1) return ZD_ERR_TRUNC;
        if (src[0] > max_sym) return ZD_ERR_CORRUPT;
        dt->alog = 0; dt->sym[0] = src[0]; dt->nbits[0] = 0; dt->base[0] = 0; *ok = 1; return 1;
    }
    if (mode == 2) {
        int16_t norm[64]; int nsym, alog;
        long used = fse_read_desc(src, len, norm, &nsym, &alog, max_sym, max_alog);
        if (used < 0) return used;
        int rc = fse_build(dt, norm, nsym, alog); if (rc) return rc;
        *ok = 1; return used;
    }
    if (!*ok) return ZD_ERR_CORRUPT; /* repeat without a previous table */
    return 0;
}

/* ---------- block ---------- */
static long decode_block(frame_ctx *fc, const uint8_t *src, size_t len,
                         uint8_t *dst_base, size_t dst_pos, size_t dst_cap, size_t frame_start,
                         uint8_t *litbuf /* >= 128 KiB + 32 */) {
    if (len < 1) return ZD_ERR_TRUNC;
    /* literals section */
    int ltype = src[0] & 3, sf = (src[0] >> 2) & 3;
    size_t regen, comp = 0, hdr;
    int streams = 1;
    if (ltype < 2) {
        if (sf == 0 || sf == 2) { regen = src[0] >> 3; hdr = 1; }
        else if (sf == 1) { if (len < 2) return ZD_ERR_TRUNC; regen = (src[0] >> 4) + ((size_t)src[1] << 4); hdr = 2; }
        else { if (len < 3) return ZD_ERR_TRUNC; regen = (src[0] >> 4) + ((size_t)src[1] << 4) + ((size_t)src[2] << 12); hdr = 3; }
    } else {
        if (len < 5) return ZD_ERR_TRUNC;
        uint64_t v = 0; for (int i = 0; i < 5; i++) v |= (uint64_t)src[i] << (8 * i);
        if (sf == 0) { streams = 1; hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; }
        else if (sf == 1) { streams = 4; hdr = 3; regen = (v >> 4) & 0x3FF; comp = (v >> 14) & 0x3FF; }
        else if (sf == 2) { streams = 4; hdr = 4; regen = (v >> 4) & 0x3FFF; comp = (v >> 18) & 0x3FFF; }
        else { streams = 4; hdr = 5; regen = (v >> 4) & 0x3FFFF; comp = (v >> 22) & 0x3FFFF; }
    }
    if (regen > 128 * 1024) return ZD_ERR_CORRUPT;
    const uint8_t *lit; size_t pos = hdr;
    if (ltype == 0) {
        if (pos + regen > len) return ZD_ERR_TRUNC;
        lit = src + pos; pos += regen;
    } else if (ltype == 1) {
        if (pos + 1 > len) return ZD_ERR_TRUNC;
        memset(litbuf, src[pos], regen); lit = litbuf; pos += 1;
    } else {
        if (pos + comp > len) return ZD_ERR_TRUNC;
        const uint8_t *cs = src + pos; size_t cl = comp;
        if (ltype == 2) {
            long used = huf_read_tree(&fc->huf, cs, cl);
            if (used < 0) return used;
            cs += used; cl -= (size_t)used;
        } else if (!fc->huf.valid) return ZD_ERR_CORRUPT;
        if (streams == 1) {
            int rc = huf_decode_stream(&fc->huf, cs, cl, litbuf, regen); if (rc) return rc;
        } else {
            if (cl < 6) return ZD_ERR_CORRUPT;
            size_t s1 = cs[0] | (cs[1] << 8), s2 = cs[2] | (cs[3] << 8), s3 = cs[4] | (cs[5] << 8);
            if (6 + s1 + s2 + s3 > cl) return ZD_ERR_CORRUPT;
            size_t s4 = cl - 6 - s1 - s2 - s3;
            size_t seg = (regen + 3) / 4;
            if (seg * 3 > regen) return ZD_ERR_CORRUPT;
            const uint8_t *p = cs + 6;
            int rc;
            rc = huf_decode_stream(&fc->huf, p, s1, litbuf, seg); if (rc) return rc; p += s1;
            rc = huf_decode_stream(&fc->huf, p, s2, litbuf + seg, seg); if (rc) return rc; p += s2;
            rc = huf_decode_stream(&fc->huf, p, s3, litbuf + 2 * seg, seg); if (rc) return rc; p += s3;
            rc = huf_decode_stream(&fc->huf, p, s4, litbuf + 3 * seg, regen - 3 * seg); if (rc) return rc;
        }
        lit = litbuf; pos += comp;
    }
    /* sequences section */
    if (pos >= len) return ZD_ERR_TRUNC;
    size_t nseq; uint8_t b0 = src[pos++];
    if (b0 < 128) nseq = b0;
    else if (b0 < 255) { if (pos >= len) return ZD_ERR_TRUNC; nseq = ((size_t)(b0 - 128) << 8) + src[pos++]; }
    else { if (pos + 2 > len) return ZD_ERR_TRUNC; nseq = src[pos] + ((size_t)src[pos + 1] << 8) + 0x7F00; pos += 2; }
    size_t out = dst_pos;
    if (nseq == 0) {
        if (pos != len) return ZD_ERR_CORRUPT;
        if (out + regen > dst_cap) return ZD_ERR_DSTFULL;
        memcpy(dst_base + out, lit, regen);
        return (long)regen;
    }
    if (pos >= len) return ZD_ERR_TRUNC;
    uint8_t modes = src[pos++];
    if (modes & 3) return ZD_ERR_RESERVED;
    long u;
    u = seq_table(&fc->ll, &fc->ll_ok, modes >> 6, src + pos, len - pos, LL_DEF, 36, 6, MAX_LL, 9); if (u < 0) return u; pos += (size_t)u;
    u = seq_table(&fc->of, &fc->of_ok, (modes >> 4) & 3, src + pos, len - pos, OF_DEF, 29, 5, MAX_OF, 8); if (u < 0) return u; pos += (size_t)u;
    u = seq_table(&fc->ml, &fc->ml_ok, (modes >> 2) & 3, src + pos, len - pos, ML_DEF, 53, 6, MAX_ML, 9); if (u < 0) return u; pos += (size_t)u;
    if (pos >= len) return ZD_ERR_TRUNC;
    bbits b; int rc = bb_init(&b, src + pos, len - pos); if (rc) return rc;
    uint32_t sll = (uint32_t)bb_read(&b, fc->ll.alog);
    uint32_t sof = (uint32_t)bb_read(&b, fc->of.alog);
    uint32_t sml = (uint32_t)bb_read(&b, fc->ml.alog);
    size_t litpos = 0;
    for (size_t i = 0; i < nseq; i++) {
        int ofc = fc->of.sym[sof], mlc = fc->ml.sym[sml], llc = fc->ll.sym[sll];
        if (ofc > MAX_OF || mlc > MAX_ML || llc > MAX_LL) return ZD_ERR_CORRUPT;
        uint64_t ofv = ((uint64_t)1 << ofc) + bb_read(&b, ofc);
        uint64_t ml = ML_BASE[mlc] + bb_read(&b, ML_BITS[mlc]);
        uint64_t ll = LL_BASE[llc] + bb_read(&b, LL_BITS[llc]);
        if (b.off < 0) return ZD_ERR_CORRUPT;
        uint64_t offset;
        if (ofv > 3) { offset = ofv - 3; fc->rep[2] = fc->rep[1]; fc->rep[1] = fc->rep[0]; fc->rep[0] = offset; }
        else {
            uint64_t idx = ofv - 1 + (ll == 0 ? 1 : 0);
            if (idx == 0) offset = fc->rep[0];
            else {
                offset = idx < 3 ? fc->rep[idx] : fc->rep[0] - 1;
                if (offset == 0) return ZD_ERR_CORRUPT;
                if (idx > 1) fc->rep[2] = fc->rep[1];
                fc->rep[1] = fc->rep[0]; fc->rep[0] = offset;
            }
        }
        if (i + 1 < nseq) {
            sll = fc->ll.base[sll] + (uint32_t)bb_read(&b, fc->ll.nbits[sll]);
            sml = fc->ml.base[sml] + (uint32_t)bb_read(&b, fc->ml.nbits[sml]);
            sof = fc->of.base[sof] + (uint32_t)bb_read(&b, fc->of.nbits[sof]);
            if (b.off < 0) return ZD_ERR_CORRUPT;
        }
        if (litpos + ll > regen) return ZD_ERR_CORRUPT;
        if (out + ll + ml > dst_cap) return ZD_ERR_DSTFULL;
        memcpy(dst_base + out, lit + litpos, ll); out += ll; litpos += ll;
        if (offset > out - frame_start) return ZD_ERR_CORRUPT;
        for (uint64_t k = 0; k < ml; k++) { dst_base[out] = dst_base[out - offset]; out++; }
    }
    if (b.off != 0) return ZD_ERR_CORRUPT;
    size_t rest = regen - litpos;
    if (out + rest > dst_cap) return ZD_ERR_DSTFULL;
    memcpy(dst_base + out, lit + litpos, rest); out += rest;
    if (out - dst_pos > 128 * 1024) return ZD_ERR_CORRUPT;
    return (long)(out - dst_pos);
}

/* Decode all concatenated frames in src. Returns total bytes or <0.
 * If frames_out != NULL it receives the number of frames decoded. */
long pna_oracle_zstd_decompress(const uint8_t *src, size_t len, uint8_t *dst, size_t cap, int *frames_out) {
    size_t ip = 0, op = 0; int frames = 0;
    uint8_t *litbuf = (uint8_t *)malloc(128 * 1024 + 64);
    frame_ctx *fc = (frame_ctx *)malloc(sizeof(frame_ctx));
    long ret = 0;
    if (!litbuf || !fc) { ret = ZD_ERR_DSTFULL; goto done; }
    while (ip < len) {
        if (len - ip < 4) { ret = ZD_ERR_TRUNC; goto done; }
        uint32_t magic = src[ip] | (src[ip + 1] << 8) | (src[ip + 2] << 16) | ((uint32_t)src[ip + 3] << 24);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (len - ip < 8) { ret = ZD_ERR_TRUNC; goto done; }
            uint32_t sz = src[ip + 4] | (src[ip + 5] << 8) | (src[ip + 6] << 16) | ((uint32_t)src[ip + 7] << 24);
            if (len - ip - 8 < sz) { ret = ZD_ERR_TRUNC; goto done; }
            ip += 8 + sz; continue;
        }
        if (magic != 0xFD2FB528u) { ret = ZD_ERR_MAGIC; goto done; }
        ip += 4;
        if (ip >= len) { ret = ZD_ERR_TRUNC; goto done; }
        uint8_t fhd = src[ip++];
        int fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, checksum = (fhd >> 2) & 1, dict = fhd & 3;
        if (fhd & 0x08) { ret = ZD_ERR_RESERVED; goto done; }
        uint64_t window = 0;
        if (!single) {
            if (ip >= len) { ret = ZD_ERR_TRUNC; goto done; }
            uint8_t wd = src[ip++]; int wl = 10 + (wd >> 3);
            window = (1ull << wl) + ((1ull << wl) >> 3) * (wd & 7);
            if (wl > 31) { ret = ZD_ERR_WINDOW; goto done; }
        }
        static const int dsz[4] = {0, 1, 2, 4};
        if (dict) {
            uint32_t id = 0;
            if (len - ip < (size_t)dsz[dict]) { ret = ZD_ERR_TRUNC; goto done; }
            for (int i = 0; i < dsz[dict]; i++) id |= (uint32_t)src[ip + i] << (8 * i);
            ip += (size_t)dsz[dict];
            if (id) { ret = ZD_ERR_DICT; goto done; }
        }
        int fsz = fcs_flag == 0 ? single : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
        uint64_t fcs = 0;
        if (len - ip < (size_t)fsz) { ret = ZD_ERR_TRUNC; goto done; }
        for (int i = 0; i < fsz; i++) fcs |= (uint64_t)src[ip + i] << (8 * i);
        if (fsz == 2) fcs += 256;
        ip += (size_t)fsz;
        (void)window;
        memset(fc, 0, sizeof(*fc));
        fc->rep[0] = 1; fc->rep[1] = 4; fc->rep[2] = 8;
        size_t frame_start = op;
        for (;;) {
            if (len - ip < 3) { ret = ZD_ERR_TRUNC; goto done; }
            uint32_t bh = src[ip] | (src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16);
            ip += 3;
            int last = bh & 1, type = (bh >> 1) & 3; size_t bsz = bh >> 3;
            if (type == 3) { ret = ZD_ERR_RESERVED; goto done; }
            if (type == 0) {
                if (bsz > 128 * 1024) { ret = ZD_ERR_CORRUPT; goto done; }
                if (len - ip < bsz) { ret = ZD_ERR_TRUNC; goto done; }
                if (cap - op < bsz) { ret = ZD_ERR_DSTFULL; goto done; }
                memcpy(dst + op, src + ip, bsz); ip += bsz; op += bsz;
            } else if (type == 1) {
                if (bsz > 128 * 1024) { ret = ZD_ERR_CORRUPT; goto done; }
                if (len - ip < 1) { ret = ZD_ERR_TRUNC; goto done; }
                if (cap - op < bsz) { ret = ZD_ERR_DSTFULL; goto done; }
                memset(dst + op, src[ip], bsz); ip += 1; op += bsz;
            } else {
                if (bsz > 128 * 1024) { ret = ZD_ERR_CORRUPT; goto done; }
                if (len - ip < bsz) { ret = ZD_ERR_TRUNC; goto done; }
                long n = decode_block(fc, src + ip, bsz, dst, op, cap, frame_start, litbuf);
                if (n < 0) { ret = n; goto done; }
                ip += bsz; op += (size_t)n;
            }
            if (last) break;
        }
        if (fsz && fcs != (uint64_t)(op - frame_start)) { ret = ZD_ERR_CORRUPT; goto done; }
        if (checksum) { if (len - ip < 4) { ret = ZD_ERR_TRUNC; goto done; } ip += 4; /* XXH64 not verified */ }
        frames++;
    }
    ret = (long)op;
done:
    if (frames_out) *frames_out = frames;
    free(litbuf); free(fc);
    return ret;
}
