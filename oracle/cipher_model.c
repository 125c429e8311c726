/*
 * oracle/cipher_model.c -- TEST INFRASTRUCTURE ONLY (checker, never on the product path).
 *
 * Plain-C restatement of the cipher layer that the reference stacks between the compressor and the chunk sink
 * (get_writer: compress -> cipher -> sink, lib/src/entry/write.rs:268-274; encryption_writer, :189-248) and of the password
 * hashing that produces its key (hash(), lib/src/entry/write.rs:155-186; lib/src/hash.rs:6-45,47-88).  The arithmetic lives
 * in un-vendored crates pinned by Cargo.lock: aes 0.9.2, ctr 0.10.1 (Ctr128BE), cbc 0.2.1 (+ Pkcs7), argon2 0.5.3 (Argon2id
 * v=0x13), pbkdf2 0.12.2 (HMAC-SHA-256), password-hash 0.5.0 (the salt is the B64-decoded SaltString).  Restated from the
 * published algorithms: FIPS-197, NIST SP 800-38A, RFC 7693 (BLAKE2b), RFC 9106 (Argon2), FIPS 180-4, RFC 2104, RFC 8018.
 *
 * Pinned by tests/test_oracle_cipher.py: the FIPS-197 appendix vectors, the reference's own CTR known-answer test
 * (lib/src/cipher/stream/write.rs:78-98, AES-128 / Ctr64LE), hashlib (BLAKE2b, SHA-256, PBKDF2) and -- end to end, Argon2id
 * included -- by decrypting the reference's encrypted golden archives (resources/test/zstd_aes_{ctr,cbc}.pna,
 * solid_zstd_aes_{ctr,cbc}.pna, password "password", lib/tests/extract_compatibility.rs:120-141) to resources/test/raw/ *.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ AES (FIPS-197) */
static uint8_t SB[256], ISB[256];
static int sb_ready = 0;
static uint8_t rotl8(uint8_t x, int s) { return (uint8_t)((x << s) | (x >> (8 - s))); }
static void aes_init(void) {
    if (sb_ready) return;
    uint8_t p = 1, q = 1;
    do {                                         /* p runs over the multiplicative group (generator 3), q = 1/p */
        p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1B : 0));
        q ^= (uint8_t)(q << 1); q ^= (uint8_t)(q << 2); q ^= (uint8_t)(q << 4);
        if (q & 0x80) q ^= 0x09;
        uint8_t x = (uint8_t)(q ^ rotl8(q, 1) ^ rotl8(q, 2) ^ rotl8(q, 3) ^ rotl8(q, 4) ^ 0x63);
        SB[p] = x; ISB[x] = p;
    } while (p != 1);
    SB[0] = 0x63; ISB[0x63] = 0;
    sb_ready = 1;
}
static uint8_t xt(uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1B : 0)); }
static uint8_t gmul(uint8_t a, uint8_t b) { uint8_t r = 0; while (b) { if (b & 1) r ^= a; a = xt(a); b >>= 1; } return r; }

typedef struct { uint8_t rk[15][16]; int nr; } aes_key;

/* key_len 16 or 32 -- FIPS-197 5.2 */
void pna_aes_expand(const uint8_t *key, int key_len, aes_key *k) {
    aes_init();
    const int nk = key_len / 4, nr = nk + 6;
    uint8_t w[60][4];
    uint8_t rc = 1;
    for (int i = 0; i < nk; i++) memcpy(w[i], key + 4 * i, 4);
    for (int i = nk; i < 4 * (nr + 1); i++) {
        uint8_t t[4]; memcpy(t, w[i - 1], 4);
        if (i % nk == 0) {
            uint8_t t0 = t[0];
            t[0] = (uint8_t)(SB[t[1]] ^ rc); t[1] = SB[t[2]]; t[2] = SB[t[3]]; t[3] = SB[t0];
            rc = xt(rc);
        } else if (nk > 6 && i % nk == 4) { for (int j = 0; j < 4; j++) t[j] = SB[t[j]]; }
        for (int j = 0; j < 4; j++) w[i][j] = (uint8_t)(w[i - nk][j] ^ t[j]);
    }
    k->nr = nr;
    for (int r = 0; r <= nr; r++) for (int c = 0; c < 4; c++) memcpy(&k->rk[r][4 * c], w[4 * r + c], 4);
}

void pna_aes_encrypt_block(const aes_key *k, const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = (uint8_t)(in[i] ^ k->rk[0][i]);
    for (int r = 1; r <= k->nr; r++) {
        for (int c = 0; c < 4; c++) for (int row = 0; row < 4; row++) t[4 * c + row] = SB[s[4 * ((c + row) & 3) + row]];   /* SubBytes + ShiftRows */
        if (r < k->nr) {
            for (int c = 0; c < 4; c++) {
                const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c]     = (uint8_t)(xt(a0) ^ xt(a1) ^ a1 ^ a2 ^ a3);
                s[4 * c + 1] = (uint8_t)(a0 ^ xt(a1) ^ xt(a2) ^ a2 ^ a3);
                s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ xt(a2) ^ xt(a3) ^ a3);
                s[4 * c + 3] = (uint8_t)(xt(a0) ^ a0 ^ a1 ^ a2 ^ xt(a3));
            }
        } else memcpy(s, t, 16);
        for (int i = 0; i < 16; i++) s[i] ^= k->rk[r][i];
    }
    memcpy(out, s, 16);
}

void pna_aes_decrypt_block(const aes_key *k, const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = (uint8_t)(in[i] ^ k->rk[k->nr][i]);
    for (int r = k->nr - 1; r >= 0; r--) {
        for (int c = 0; c < 4; c++) for (int row = 0; row < 4; row++) t[4 * ((c + row) & 3) + row] = ISB[s[4 * c + row]];    /* InvShiftRows + InvSubBytes */
        for (int i = 0; i < 16; i++) t[i] ^= k->rk[r][i];
        if (r > 0) {
            for (int c = 0; c < 4; c++) {
                const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c]     = (uint8_t)(gmul(a0, 14) ^ gmul(a1, 11) ^ gmul(a2, 13) ^ gmul(a3, 9));
                s[4 * c + 1] = (uint8_t)(gmul(a0, 9) ^ gmul(a1, 14) ^ gmul(a2, 11) ^ gmul(a3, 13));
                s[4 * c + 2] = (uint8_t)(gmul(a0, 13) ^ gmul(a1, 9) ^ gmul(a2, 14) ^ gmul(a3, 11));
                s[4 * c + 3] = (uint8_t)(gmul(a0, 11) ^ gmul(a1, 13) ^ gmul(a2, 9) ^ gmul(a3, 14));
            }
        } else memcpy(s, t, 16);
    }
    memcpy(out, s, 16);
}

/* one-shot block helpers for the Python side */
void pna_oracle_aes_block(const uint8_t *key, int key_len, const uint8_t in[16], uint8_t out[16], int decrypt) {
    aes_key k; pna_aes_expand(key, key_len, &k);
    if (decrypt) pna_aes_decrypt_block(&k, in, out); else pna_aes_encrypt_block(&k, in, out);
}

/* CTR keystream applied to buf[0..n) starting at stream byte `pos` (so pieces of one stream can be processed apart).
 * flavor 0: Ctr128BE -- the whole IV is a 128-bit big-endian counter (what the reference uses: Ctr128BEWriter,
 * lib/src/entry/write.rs:214,232); flavor 1: Ctr64LE -- the first 8 bytes of the IV are a little-endian 64-bit counter (only
 * in the reference's unit test, lib/src/cipher/stream/write.rs:74-76). */
void pna_oracle_aes_ctr(const uint8_t *key, int key_len, const uint8_t iv[16], int flavor, uint64_t pos, uint8_t *buf, size_t n) {
    aes_key k; pna_aes_expand(key, key_len, &k);
    uint64_t blk = pos / 16; unsigned o = (unsigned)(pos % 16);
    size_t i = 0;
    while (i < n) {
        uint8_t ctr[16], ks[16];
        memcpy(ctr, iv, 16);
        if (flavor == 0) {
            uint64_t add = blk; unsigned carry = 0;
            for (int b = 15; b >= 0; b--) { unsigned v = ctr[b] + (unsigned)(add & 0xFF) + carry; ctr[b] = (uint8_t)v; carry = v >> 8; add >>= 8; }
        } else {
            /* ctr crate, Ctr64LE: the FIRST 8 bytes are the counter, read little-endian; the rest is the nonce */
            uint64_t c = 0; for (int b = 0; b < 8; b++) c |= (uint64_t)iv[b] << (8 * b);
            c += blk; for (int b = 0; b < 8; b++) ctr[b] = (uint8_t)(c >> (8 * b));
        }
        pna_aes_encrypt_block(&k, ctr, ks);
        for (; o < 16 && i < n; o++, i++) buf[i] ^= ks[o];
        o = 0; blk++;
    }
}

/* CBC with PKCS#7 padding (EncryptCbcAes256Writer = CbcBlockCipherEncryptWriter<_, Aes256, Pkcs7>,
 * lib/src/cipher/block/write.rs:44-57,67-107).  out must hold (n / 16 + 1) * 16 bytes; returns that size. */
size_t pna_oracle_aes_cbc_encrypt(const uint8_t *key, int key_len, const uint8_t iv[16], const uint8_t *in, size_t n, uint8_t *out) {
    aes_key k; pna_aes_expand(key, key_len, &k);
    uint8_t prev[16]; memcpy(prev, iv, 16);
    const size_t nb = n / 16 + 1;
    for (size_t b = 0; b < nb; b++) {
        uint8_t x[16];
        const size_t have = (b + 1 < nb) ? 16 : n - 16 * b;
        memcpy(x, in + 16 * b, have);
        for (size_t j = have; j < 16; j++) x[j] = (uint8_t)(16 - have);
        for (int j = 0; j < 16; j++) x[j] ^= prev[j];
        pna_aes_encrypt_block(&k, x, prev);
        memcpy(out + 16 * b, prev, 16);
    }
    return nb * 16;
}
/* returns the plaintext length or -1 (bad length / padding) */
long pna_oracle_aes_cbc_decrypt(const uint8_t *key, int key_len, const uint8_t iv[16], const uint8_t *in, size_t n, uint8_t *out) {
    if (n == 0 || n % 16) return -1;
    aes_key k; pna_aes_expand(key, key_len, &k);
    const uint8_t *prev = iv;
    for (size_t b = 0; b < n / 16; b++) {
        uint8_t x[16];
        pna_aes_decrypt_block(&k, in + 16 * b, x);
        for (int j = 0; j < 16; j++) out[16 * b + j] = (uint8_t)(x[j] ^ prev[j]);
        prev = in + 16 * b;
    }
    const unsigned pad = out[n - 1];
    if (pad < 1 || pad > 16) return -1;
    for (unsigned j = 0; j < pad; j++) if (out[n - 1 - j] != pad) return -1;
    return (long)(n - pad);
}

/* ------------------------------------------------------------------ BLAKE2b (RFC 7693), unkeyed */
static const uint64_t B2_IV[8] = {
    0x6A09E667F3BCC908ull, 0xBB67AE8584CAA73Bull, 0x3C6EF372FE94F82Bull, 0xA54FF53A5F1D36F1ull,
    0x510E527FADE682D1ull, 0x9B05688C2B3E6C1Full, 0x1F83D9ABFB41BD6Bull, 0x5BE0CD19137E2179ull };
static const uint8_t B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3} };
static uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static uint64_t ld64(const uint8_t *p) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i); return v; }
static void st64(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static void st32(uint8_t *p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i)); }

typedef struct { uint64_t h[8]; uint64_t t; uint8_t buf[128]; size_t fill; size_t outlen; } b2_state;
static void b2_compress(b2_state *S, const uint8_t blk[128], int last) {
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; i++) m[i] = ld64(blk + 8 * i);
    for (int i = 0; i < 8; i++) { v[i] = S->h[i]; v[i + 8] = B2_IV[i]; }
    v[12] ^= S->t;                               /* the high counter word stays 0 for the sizes used here */
    if (last) v[14] = ~v[14];
#define B2G(a, b, c, d, x, y) do { v[a] += v[b] + (x); v[d] = rotr64(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = rotr64(v[b] ^ v[c], 24); \
                                   v[a] += v[b] + (y); v[d] = rotr64(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = rotr64(v[b] ^ v[c], 63); } while (0)
    for (int r = 0; r < 12; r++) {
        const uint8_t *s = B2_SIGMA[r];
        B2G(0, 4, 8, 12, m[s[0]], m[s[1]]); B2G(1, 5, 9, 13, m[s[2]], m[s[3]]); B2G(2, 6, 10, 14, m[s[4]], m[s[5]]); B2G(3, 7, 11, 15, m[s[6]], m[s[7]]);
        B2G(0, 5, 10, 15, m[s[8]], m[s[9]]); B2G(1, 6, 11, 12, m[s[10]], m[s[11]]); B2G(2, 7, 8, 13, m[s[12]], m[s[13]]); B2G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2G
    for (int i = 0; i < 8; i++) S->h[i] ^= v[i] ^ v[i + 8];
}
static void b2_init(b2_state *S, size_t outlen) {
    memcpy(S->h, B2_IV, sizeof S->h);
    S->h[0] ^= 0x01010000ull ^ (uint64_t)outlen;
    S->t = 0; S->fill = 0; S->outlen = outlen;
}
static void b2_update(b2_state *S, const uint8_t *p, size_t n) {
    while (n) {
        if (S->fill == 128) { S->t += 128; b2_compress(S, S->buf, 0); S->fill = 0; }
        size_t k = 128 - S->fill; if (k > n) k = n;
        memcpy(S->buf + S->fill, p, k); S->fill += k; p += k; n -= k;
    }
}
static void b2_final(b2_state *S, uint8_t *out) {
    S->t += S->fill;
    memset(S->buf + S->fill, 0, 128 - S->fill);
    b2_compress(S, S->buf, 1);
    uint8_t full[64];
    for (int i = 0; i < 8; i++) st64(full + 8 * i, S->h[i]);
    memcpy(out, full, S->outlen);
}
void pna_oracle_blake2b(const uint8_t *in, size_t n, uint8_t *out, size_t outlen) {
    b2_state S; b2_init(&S, outlen); b2_update(&S, in, n); b2_final(&S, out);
}

/* ------------------------------------------------------------------ Argon2 (RFC 9106), version 0x13, no secret / associated data */
/* variable-length hash H' (RFC 9106 3.3) */
static void a2_hprime(const uint8_t *in, size_t n, uint8_t *out, uint32_t T) {
    uint8_t le[4]; st32(le, T);
    b2_state S;
    if (T <= 64) { b2_init(&S, T); b2_update(&S, le, 4); b2_update(&S, in, n); b2_final(&S, out); return; }
    uint8_t v[64];
    b2_init(&S, 64); b2_update(&S, le, 4); b2_update(&S, in, n); b2_final(&S, v);
    memcpy(out, v, 32); out += 32;
    uint32_t left = T - 32;
    while (left > 64) {
        uint8_t nx[64];
        pna_oracle_blake2b(v, 64, nx, 64); memcpy(v, nx, 64);
        memcpy(out, v, 32); out += 32; left -= 32;
    }
    uint8_t last[64];
    pna_oracle_blake2b(v, 64, last, left);
    memcpy(out, last, left);
}
typedef struct { uint64_t v[128]; } a2_block;
static uint64_t blamka(uint64_t x, uint64_t y) { return x + y + 2 * (x & 0xFFFFFFFFull) * (y & 0xFFFFFFFFull); }
#define A2G(a, b, c, d) do { a = blamka(a, b); d = rotr64(d ^ a, 32); c = blamka(c, d); b = rotr64(b ^ c, 24); \
                             a = blamka(a, b); d = rotr64(d ^ a, 16); c = blamka(c, d); b = rotr64(b ^ c, 63); } while (0)
static void a2_round(uint64_t *v0, uint64_t *v1, uint64_t *v2, uint64_t *v3, uint64_t *v4, uint64_t *v5, uint64_t *v6, uint64_t *v7,
                     uint64_t *v8, uint64_t *v9, uint64_t *v10, uint64_t *v11, uint64_t *v12, uint64_t *v13, uint64_t *v14, uint64_t *v15) {
    A2G(*v0, *v4, *v8, *v12); A2G(*v1, *v5, *v9, *v13); A2G(*v2, *v6, *v10, *v14); A2G(*v3, *v7, *v11, *v15);
    A2G(*v0, *v5, *v10, *v15); A2G(*v1, *v6, *v11, *v12); A2G(*v2, *v7, *v8, *v13); A2G(*v3, *v4, *v9, *v14);
}
/* next = G(prev, ref) (xor-ed onto the old next from the second pass on) -- RFC 9106 3.5 */
static void a2_fill(const a2_block *prev, const a2_block *ref, a2_block *next, int with_xor) {
    a2_block W, keep;
    for (int i = 0; i < 128; i++) W.v[i] = prev->v[i] ^ ref->v[i];
    keep = W;
    if (with_xor) for (int i = 0; i < 128; i++) keep.v[i] ^= next->v[i];
    for (int i = 0; i < 8; i++) {
        uint64_t *r = W.v + 16 * i;
        a2_round(r, r + 1, r + 2, r + 3, r + 4, r + 5, r + 6, r + 7, r + 8, r + 9, r + 10, r + 11, r + 12, r + 13, r + 14, r + 15);
    }
    for (int i = 0; i < 8; i++) {
        uint64_t *c = W.v + 2 * i;
        a2_round(c, c + 1, c + 16, c + 17, c + 32, c + 33, c + 48, c + 49, c + 64, c + 65, c + 80, c + 81, c + 96, c + 97, c + 112, c + 113);
    }
    for (int i = 0; i < 128; i++) next->v[i] = W.v[i] ^ keep.v[i];
}
/* type: 0 Argon2d, 1 Argon2i, 2 Argon2id.  Returns 0 or -1. */
int pna_oracle_argon2(int type, const uint8_t *pwd, uint32_t pwd_len, const uint8_t *salt, uint32_t salt_len,
                      uint32_t t_cost, uint32_t m_cost, uint32_t lanes, uint8_t *out, uint32_t out_len) {
    if (lanes < 1 || m_cost < 8 * lanes || t_cost < 1 || out_len < 4) return -1;
    const uint32_t mprime = 4 * lanes * (m_cost / (4 * lanes));
    const uint32_t lane_len = mprime / lanes, seg_len = lane_len / 4;
    a2_block *mem = (a2_block *)malloc((size_t)mprime * sizeof(a2_block));
    if (!mem) return -1;
    uint8_t h0[64 + 8];
    {
        b2_state S; uint8_t le[4];
        b2_init(&S, 64);
        const uint32_t hdr[6] = {lanes, out_len, m_cost, t_cost, 0x13u, (uint32_t)type};
        for (int i = 0; i < 6; i++) { st32(le, hdr[i]); b2_update(&S, le, 4); }
        st32(le, pwd_len); b2_update(&S, le, 4); b2_update(&S, pwd, pwd_len);
        st32(le, salt_len); b2_update(&S, le, 4); b2_update(&S, salt, salt_len);
        st32(le, 0); b2_update(&S, le, 4);       /* no secret */
        st32(le, 0); b2_update(&S, le, 4);       /* no associated data */
        b2_final(&S, h0);
    }
    for (uint32_t l = 0; l < lanes; l++) for (uint32_t j = 0; j < 2; j++) {
        uint8_t bytes[1024];
        st32(h0 + 64, j); st32(h0 + 68, l);
        a2_hprime(h0, 72, bytes, 1024);
        for (int i = 0; i < 128; i++) mem[(size_t)l * lane_len + j].v[i] = ld64(bytes + 8 * i);
    }
    a2_block zero; memset(&zero, 0, sizeof zero);
    for (uint32_t pass = 0; pass < t_cost; pass++) for (uint32_t slice = 0; slice < 4; slice++) for (uint32_t lane = 0; lane < lanes; lane++) {
        const int indep = type == 1 || (type == 2 && pass == 0 && slice < 2);
        a2_block input, addr;
        memset(&input, 0, sizeof input); memset(&addr, 0, sizeof addr);
        if (indep) { input.v[0] = pass; input.v[1] = lane; input.v[2] = slice; input.v[3] = mprime; input.v[4] = t_cost; input.v[5] = (uint64_t)type; }
        uint32_t start = 0;
        if (pass == 0 && slice == 0) {
            start = 2;
            if (indep) { input.v[6]++; a2_fill(&zero, &input, &addr, 0); a2_fill(&zero, &addr, &addr, 0); }
        }
        uint32_t cur = lane * lane_len + slice * seg_len + start;
        uint32_t prev = (cur % lane_len == 0) ? cur + lane_len - 1 : cur - 1;
        for (uint32_t i = start; i < seg_len; i++, cur++, prev++) {
            if (cur % lane_len == 1) prev = cur - 1;
            uint64_t rnd;
            if (indep) {
                if (i % 128 == 0) { input.v[6]++; a2_fill(&zero, &input, &addr, 0); a2_fill(&zero, &addr, &addr, 0); }
                rnd = addr.v[i % 128];
            } else rnd = mem[prev].v[0];
            uint32_t ref_lane = (uint32_t)((rnd >> 32) % lanes);
            if (pass == 0 && slice == 0) ref_lane = lane;
            const int same = ref_lane == lane;
            uint32_t area;
            if (pass == 0) {
                if (slice == 0) area = i - 1;
                else if (same) area = slice * seg_len + i - 1;
                else area = slice * seg_len - (i == 0 ? 1u : 0u);
            } else {
                if (same) area = lane_len - seg_len + i - 1;
                else area = lane_len - seg_len - (i == 0 ? 1u : 0u);
            }
            uint64_t rel = rnd & 0xFFFFFFFFull;
            rel = (rel * rel) >> 32;
            rel = (uint64_t)area - 1 - (((uint64_t)area * rel) >> 32);
            uint32_t sp = 0;
            if (pass != 0) sp = (slice == 3) ? 0 : (slice + 1) * seg_len;
            const uint32_t ref_index = (uint32_t)((sp + rel) % lane_len);
            const a2_block *ref = &mem[(size_t)ref_lane * lane_len + ref_index];
            a2_fill(&mem[prev], ref, &mem[cur], pass != 0);
        }
    }
    a2_block fin = mem[lane_len - 1];
    for (uint32_t l = 1; l < lanes; l++) for (int i = 0; i < 128; i++) fin.v[i] ^= mem[(size_t)l * lane_len + lane_len - 1].v[i];
    uint8_t bytes[1024];
    for (int i = 0; i < 128; i++) st64(bytes + 8 * i, fin.v[i]);
    a2_hprime(bytes, 1024, out, out_len);
    free(mem);
    return 0;
}

/* ------------------------------------------------------------------ SHA-256, HMAC, PBKDF2 (FIPS 180-4, RFC 2104, RFC 8018) */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2 };
typedef struct { uint32_t h[8]; uint64_t len; uint8_t buf[64]; size_t fill; } sha_state;
static uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha_block(sha_state *S, const uint8_t *p) {
    uint32_t w[64], a[8];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        const uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    memcpy(a, S->h, sizeof a);
    for (int i = 0; i < 64; i++) {
        const uint32_t S1 = rotr32(a[4], 6) ^ rotr32(a[4], 11) ^ rotr32(a[4], 25), ch = (a[4] & a[5]) ^ (~a[4] & a[6]);
        const uint32_t t1 = a[7] + S1 + ch + K256[i] + w[i];
        const uint32_t S0 = rotr32(a[0], 2) ^ rotr32(a[0], 13) ^ rotr32(a[0], 22), mj = (a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]);
        const uint32_t t2 = S0 + mj;
        a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) S->h[i] += a[i];
}
static void sha_init(sha_state *S) {
    static const uint32_t h0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(S->h, h0, sizeof h0); S->len = 0; S->fill = 0;
}
static void sha_update(sha_state *S, const uint8_t *p, size_t n) {
    S->len += n;
    while (n) {
        size_t k = 64 - S->fill; if (k > n) k = n;
        memcpy(S->buf + S->fill, p, k); S->fill += k; p += k; n -= k;
        if (S->fill == 64) { sha_block(S, S->buf); S->fill = 0; }
    }
}
static void sha_final(sha_state *S, uint8_t out[32]) {
    const uint64_t bits = S->len * 8;
    uint8_t pad = 0x80; sha_update(S, &pad, 1);
    pad = 0; while (S->fill != 56) sha_update(S, &pad, 1);
    uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha_update(S, lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(S->h[i] >> 24); out[4 * i + 1] = (uint8_t)(S->h[i] >> 16); out[4 * i + 2] = (uint8_t)(S->h[i] >> 8); out[4 * i + 3] = (uint8_t)S->h[i]; }
}
void pna_oracle_sha256(const uint8_t *in, size_t n, uint8_t out[32]) { sha_state S; sha_init(&S); sha_update(&S, in, n); sha_final(&S, out); }
static void hmac_sha256(const uint8_t *key, size_t klen, const uint8_t *a, size_t alen, const uint8_t *b, size_t blen, uint8_t out[32]) {
    uint8_t k[64] = {0}, pad[64], inner[32];
    if (klen > 64) pna_oracle_sha256(key, klen, k); else memcpy(k, key, klen);
    sha_state S;
    for (int i = 0; i < 64; i++) pad[i] = k[i] ^ 0x36;
    sha_init(&S); sha_update(&S, pad, 64); sha_update(&S, a, alen); sha_update(&S, b, blen); sha_final(&S, inner);
    for (int i = 0; i < 64; i++) pad[i] = k[i] ^ 0x5c;
    sha_init(&S); sha_update(&S, pad, 64); sha_update(&S, inner, 32); sha_final(&S, out);
}
void pna_oracle_pbkdf2_sha256(const uint8_t *pwd, size_t pwd_len, const uint8_t *salt, size_t salt_len, uint32_t rounds, uint8_t *out, size_t out_len) {
    for (uint32_t blk = 1; out_len; blk++) {
        uint8_t be[4] = {(uint8_t)(blk >> 24), (uint8_t)(blk >> 16), (uint8_t)(blk >> 8), (uint8_t)blk}, u[32], t[32];
        hmac_sha256(pwd, pwd_len, salt, salt_len, be, 4, u);
        memcpy(t, u, 32);
        for (uint32_t r = 1; r < rounds; r++) { hmac_sha256(pwd, pwd_len, u, 32, NULL, 0, u); for (int i = 0; i < 32; i++) t[i] ^= u[i]; }
        const size_t k = out_len < 32 ? out_len : 32;
        memcpy(out, t, k); out += k; out_len -= k;
    }
}

/* ------------------------------------------------------------------ HKDF-SHA-256 (RFC 5869), AES-GCM (NIST SP 800-38D)
 * Cipher mode 2 of the reference ("GCM STREAM", lib/src/cipher/aead.rs, lib/src/cipher/gcm.rs): third-party crates hkdf 0.13 / sha2 /
 * aes-gcm; restated from the published algorithms. */
void pna_oracle_hkdf_sha256(const uint8_t *ikm, size_t ikm_len, const uint8_t *salt, size_t salt_len, const uint8_t *info, size_t info_len,
                            uint8_t *okm, size_t okm_len) {
    uint8_t zero[32] = {0}, prk[32], t[32];
    if (salt_len == 0) { salt = zero; salt_len = 32; }                 /* RFC 5869 2.2: absent salt = HashLen zeros */
    hmac_sha256(salt, salt_len, ikm, ikm_len, NULL, 0, prk);
    size_t tl = 0;
    for (uint8_t ctr = 1; okm_len; ctr++) {
        /* T(n) = HMAC(PRK, T(n-1) || info || n) */
        uint8_t buf[32 + 1024 + 1]; size_t n = 0;
        if (info_len > 1024) return;
        memcpy(buf, t, tl); n += tl; memcpy(buf + n, info, info_len); n += info_len; buf[n++] = ctr;
        hmac_sha256(prk, 32, buf, n, NULL, 0, t); tl = 32;
        const size_t k = okm_len < 32 ? okm_len : 32;
        memcpy(okm, t, k); okm += k; okm_len -= k;
    }
}

/* GF(2^128) multiply, GCM bit order (bit 0 = most significant bit of byte 0) */
static void gf128_mul(const uint8_t x[16], const uint8_t y[16], uint8_t out[16]) {
    uint8_t z[16] = {0}, v[16];
    memcpy(v, y, 16);
    for (int i = 0; i < 128; i++) {
        if ((x[i >> 3] >> (7 - (i & 7))) & 1) for (int j = 0; j < 16; j++) z[j] ^= v[j];
        const int lsb = v[15] & 1;
        for (int j = 15; j > 0; j--) v[j] = (uint8_t)((v[j] >> 1) | (v[j - 1] << 7));
        v[0] >>= 1;
        if (lsb) v[0] ^= 0xE1;
    }
    memcpy(out, z, 16);
}
static void ghash(const uint8_t h[16], const uint8_t *aad, size_t aad_len, const uint8_t *c, size_t c_len, uint8_t out[16]) {
    uint8_t y[16] = {0};
    const uint8_t *parts[2] = {aad, c}; const size_t lens[2] = {aad_len, c_len};
    for (int p = 0; p < 2; p++)
        for (size_t o = 0; o < lens[p]; o += 16) {
            const size_t k = lens[p] - o < 16 ? lens[p] - o : 16;
            for (size_t j = 0; j < k; j++) y[j] ^= parts[p][o + j];
            gf128_mul(y, h, y);
        }
    uint8_t lb[16];
    const uint64_t ab = (uint64_t)aad_len * 8, cb = (uint64_t)c_len * 8;
    for (int i = 0; i < 8; i++) { lb[i] = (uint8_t)(ab >> (56 - 8 * i)); lb[8 + i] = (uint8_t)(cb >> (56 - 8 * i)); }
    for (int j = 0; j < 16; j++) y[j] ^= lb[j];
    gf128_mul(y, h, y);
    memcpy(out, y, 16);
}
/* AES-GCM with a 96-bit nonce: buf is encrypted / decrypted in place; tag (16 bytes) is written (encrypt) or compared (decrypt:
 * returns -1 on mismatch, the buffer then holds garbage). */
int pna_oracle_aes_gcm(const uint8_t *key, int key_len, const uint8_t nonce[12], const uint8_t *aad, size_t aad_len,
                       uint8_t *buf, size_t n, uint8_t tag[16], int decrypt) {
    aes_key k; pna_aes_expand(key, key_len, &k);
    uint8_t h[16], zero[16] = {0}, j0[16], ej0[16], s[16];
    pna_aes_encrypt_block(&k, zero, h);
    memcpy(j0, nonce, 12); j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
    pna_aes_encrypt_block(&k, j0, ej0);
    if (decrypt) {
        ghash(h, aad, aad_len, buf, n, s);
        uint8_t diff = 0; for (int j = 0; j < 16; j++) diff |= (uint8_t)(s[j] ^ ej0[j] ^ tag[j]);
        if (diff) return -1;
    }
    uint32_t ctr = 2;
    for (size_t o = 0; o < n; o += 16, ctr++) {
        uint8_t cb[16], ks[16];
        memcpy(cb, nonce, 12); cb[12] = (uint8_t)(ctr >> 24); cb[13] = (uint8_t)(ctr >> 16); cb[14] = (uint8_t)(ctr >> 8); cb[15] = (uint8_t)ctr;
        pna_aes_encrypt_block(&k, cb, ks);
        for (size_t j = 0; j < 16 && o + j < n; j++) buf[o + j] ^= ks[j];
    }
    if (!decrypt) { ghash(h, aad, aad_len, buf, n, s); for (int j = 0; j < 16; j++) tag[j] = (uint8_t)(s[j] ^ ej0[j]); }
    return 0;
}
