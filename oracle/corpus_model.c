/*
 * oracle/corpus_model.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Integer-only model of the synthetic corpora named by BASELINE.json / SURVEY.md §8(d):
 *   kind 0  "enwik-style text": Zipf(1.0) draws from a 50 000 pseudo-word vocabulary, sentences of
 *           5..25 words, capitalised, ". "-terminated, newline every ~40 sentences, 2 % markup tokens.
 *   kind 1  "random-text": uniform draws from the first 4 096 vocabulary words, newline every 64..96 chars.
 *   kind 2  incompressible bytes (xoshiro-style stream), kind 3 all-zero, kind 4 single repeated byte.
 * A file is a concatenation of independently generated 4 KiB PIECES (seeded by (kind, file index, piece
 * index)) so the HIP generator in the product's bench helper can produce the identical bytes with one
 * thread per piece; tests compare the two bit-for-bit.
 * There is no reference counterpart (the reference ships no corpus generator; cli/benches/create.rs:24-60
 * uses resources/test/raw).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define VOCAB      50000
#define WORD_SLOT  16          /* byte 0 = length (2..12), bytes 1.. = letters */
#define PIECE      4096
#define NPHRASE    8192        /* phrase table: 4 x u16-ish word indices per phrase (u32 each), count in [0] */
#define PHRASE_P   20000       /* of 65536: probability that the next token is a whole phrase */

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* cumulative English letter frequencies, per 10000 */
static const uint16_t LETTER_CUM[26] = {
    /* a     b     c     d     e     f     g     h     i     j     k     l     m */
    817, 966, 1244, 1669, 2939, 3162, 3364, 3973, 4670, 4685, 4762, 5165, 5406,
    /* n     o     p     q     r     s     t     u     v     w     x     y     z */
    6081, 6832, 7025, 7035, 7634, 8267, 9173, 9449, 9547, 9783, 9798, 9995, 10000};

static uint64_t mul64hi(uint64_t a, uint64_t b) {
    uint64_t a0 = a & 0xFFFFFFFFu, a1 = a >> 32, b0 = b & 0xFFFFFFFFu, b1 = b >> 32;
    uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    uint64_t mid = (p00 >> 32) + (p01 & 0xFFFFFFFFu) + (p10 & 0xFFFFFFFFu);
    return p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}

static int zipf_draw(const uint64_t *cum, uint64_t r);

/* vocab: VOCAB*WORD_SLOT bytes; zipf_cum: VOCAB u64 cumulative weights floor(2^40/(k+1));
 * phrases: NPHRASE*4 u32 = {nwords(2..3) | w0<<8.. } stored as 4 words: [0]=n, [1..3]=word ids */
void pna_corpus_tables(uint8_t *vocab, uint64_t *zipf_cum, uint32_t *phrases) {
    uint64_t s = 0x504E41ull;
    for (int w = 0; w < VOCAB; w++) {
        uint64_t r = splitmix64(&s);
        /* short words are more frequent among the top ranks: length 2..12 */
        int len = 2 + (int)((r & 0xFFFF) * 11 >> 16);
        if (w < 64) len = 2 + (int)((r & 0xFFFF) * 3 >> 16);          /* 2..4 */
        else if (w < 1024) len = 3 + (int)((r & 0xFFFF) * 5 >> 16);   /* 3..7 */
        uint8_t *slot = vocab + (size_t)w * WORD_SLOT;
        memset(slot, 0, WORD_SLOT);
        slot[0] = (uint8_t)len;
        for (int i = 0; i < len; i++) {
            uint32_t x = (uint32_t)(splitmix64(&s) >> 33) % 10000u;
            int c = 0;
            while (LETTER_CUM[c] <= x) c++;
            slot[1 + i] = (uint8_t)('a' + c);
        }
    }
    uint64_t acc = 0;
    for (int k = 0; k < VOCAB; k++) { acc += (1ull << 40) / (uint64_t)(k + 1); zipf_cum[k] = acc; }
    for (int p = 0; p < NPHRASE; p++) {
        uint64_t r = splitmix64(&s);
        int n = 2 + (int)(r & 1);
        phrases[4 * p] = (uint32_t)n;
        for (int i = 0; i < 3; i++) phrases[4 * p + 1 + i] = (uint32_t)zipf_draw(zipf_cum, splitmix64(&s));
    }
}

static int zipf_draw(const uint64_t *cum, uint64_t r) {
    uint64_t x = mul64hi(r, cum[VOCAB - 1]);
    int lo = 0, hi = VOCAB - 1;             /* first k with cum[k] > x */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cum[mid] > x) hi = mid; else lo = mid + 1; }
    return lo;
}

/* generate one 4 KiB piece (always writes exactly PIECE bytes into out) */
void pna_corpus_piece(int kind, uint64_t file_idx, uint64_t piece_idx,
                      const uint8_t *vocab, const uint64_t *zipf_cum, const uint32_t *phrases, uint8_t *out) {
    uint64_t s = 0x504E410000000000ull ^ ((uint64_t)kind << 56) ^ (file_idx * 0x9E3779B97F4A7C15ull) ^ (piece_idx * 0xD1B54A32D192ED03ull);
    if (kind == 3) { memset(out, 0, PIECE); return; }
    if (kind == 4) { memset(out, 'x', PIECE); return; }
    if (kind == 2) {
        for (int i = 0; i < PIECE; i += 8) { uint64_t r = splitmix64(&s); memcpy(out + i, &r, 8); }
        return;
    }
    int pos = 0;
    if (kind == 1) {
        int line = 0, line_max = 64 + (int)(splitmix64(&s) & 31);
        while (pos < PIECE) {
            uint64_t r = splitmix64(&s);
            const uint8_t *slot = vocab + (size_t)(r & 4095) * WORD_SLOT;
            int len = slot[0];
            for (int i = 0; i < len && pos < PIECE; i++) out[pos++] = slot[1 + i];
            line += len + 1;
            if (pos < PIECE) {
                if (line >= line_max) { out[pos++] = '\n'; line = 0; line_max = 64 + (int)((r >> 40) & 31); }
                else out[pos++] = ' ';
            }
        }
        return;
    }
    /* kind 0: enwik-style */
    int sent_left = 0, sent_count = 0, first = 1, ph_left = 0;
    const uint32_t *ph = phrases;
    while (pos < PIECE) {
        uint64_t r = splitmix64(&s);
        if (sent_left == 0) { sent_left = 5 + (int)((r >> 48) % 21); first = 1; }
        uint64_t r2 = splitmix64(&s);
        int w;
        if (ph_left > 0) { w = (int)ph[1 + (int)ph[0] - ph_left]; ph_left--; }
        else if (((r >> 24) & 0xFFFF) < PHRASE_P) {
            /* phrase index: Zipf over the first NPHRASE ranks of the same cumulative table */
            uint64_t x = mul64hi(r2, zipf_cum[NPHRASE - 1]);
            int lo = 0, hi = NPHRASE - 1;
            while (lo < hi) { int mid = (lo + hi) >> 1; if (zipf_cum[mid] > x) hi = mid; else lo = mid + 1; }
            ph = phrases + 4 * lo; w = (int)ph[1]; ph_left = (int)ph[0] - 1;
        } else w = zipf_draw(zipf_cum, r2);
        const uint8_t *slot = vocab + (size_t)w * WORD_SLOT;
        int len = slot[0];
        uint32_t mk = (uint32_t)(r & 0xFFFF);
        int markup = mk < 1311 ? 1 + (int)(mk % 3) : 0;          /* 2 % */
        if (markup == 1) { if (pos < PIECE) out[pos++] = '['; if (pos < PIECE) out[pos++] = '['; }
        if (markup == 2) { static const char t[] = "<title>"; for (int i = 0; i < 7 && pos < PIECE; i++) out[pos++] = (uint8_t)t[i]; }
        for (int i = 0; i < len && pos < PIECE; i++) {
            uint8_t c = slot[1 + i];
            if (i == 0 && first) c = (uint8_t)(c - 32);
            out[pos++] = c;
        }
        first = 0;
        if (markup == 1) { if (pos < PIECE) out[pos++] = ']'; if (pos < PIECE) out[pos++] = ']'; }
        if (markup == 2) { static const char t[] = "</title>"; for (int i = 0; i < 8 && pos < PIECE; i++) out[pos++] = (uint8_t)t[i]; }
        if (markup == 3) { static const char t[] = " &amp;"; for (int i = 0; i < 6 && pos < PIECE; i++) out[pos++] = (uint8_t)t[i]; }
        sent_left--;
        if (sent_left == 0) {
            if (pos < PIECE) out[pos++] = '.';
            sent_count++;
            if (sent_count % 40 == 0) { if (pos < PIECE) out[pos++] = '\n'; }
            else if (pos < PIECE) out[pos++] = ' ';
        } else {
            uint32_t pc = (uint32_t)((r >> 16) & 0xFF);
            if (pc < 20 && pos < PIECE) out[pos++] = ',';
            if (pos < PIECE) out[pos++] = ' ';
        }
    }
}

/* fill `len` bytes of file `file_idx` (len need not be a multiple of PIECE) */
void pna_corpus_file(int kind, uint64_t file_idx, uint8_t *out, size_t len,
                     const uint8_t *vocab, const uint64_t *zipf_cum, const uint32_t *phrases) {
    uint8_t tmp[PIECE];
    for (size_t off = 0, pi = 0; off < len; off += PIECE, pi++) {
        size_t n = len - off < PIECE ? len - off : PIECE;
        if (n == PIECE) pna_corpus_piece(kind, file_idx, pi, vocab, zipf_cum, phrases, out + off);
        else { pna_corpus_piece(kind, file_idx, pi, vocab, zipf_cum, phrases, tmp); memcpy(out + off, tmp, n); }
    }
}
