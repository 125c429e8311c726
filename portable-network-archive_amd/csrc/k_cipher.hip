// k_cipher.hip -- the cipher stage between the compressor and the chunk sink, on gfx950: AES-256 in CTR (Ctr128BE) and CBC
// (PKCS#7) mode over payloads that already sit at their archive offsets in HBM.
//
// Replaces CipherWriter::{CtrAes, CbcAes} (lib/src/entry/write.rs:189-248 encryption_writer; lib/src/cipher/stream/write.rs:52-58
// apply_keystream; lib/src/cipher/block/write.rs:44-57,67-107) in the stack compress -> cipher -> sink (get_writer,
// lib/src/entry/write.rs:268-274) and, on the read side, DecryptReader::CtrAes / CbcAes (lib/src/entry/read.rs:77-88).
//
// Integer/table work, no MFMA.  One 16-byte AES block per lane and step; the round tables live in LDS (CTR: one table replicated
// per bank, CBC: four shared ones), the 15 round keys arrive as kernel arguments (scalar registers).  CTR: block j of a stream is AES(IV + j) with the IV read as one 128-bit
// big-endian counter; every block is independent, so a stream is cut into units of <= 256 KiB (any byte position: the unit
// carries its stream offset) and each unit is one workgroup.  CBC encryption chains its blocks, so there the parallelism is
// across entries only: one lane per entry.
#include <hip/hip_runtime.h>
#include "pna_dev.h"

namespace pna {

struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };   // 16 bytes at any byte address (unaligned dwordx4 access)

__device__ __forceinline__ void aes_load_tables(uint32_t (*sT)[256], const AesTabs *__restrict__ tabs, uint32_t tid, uint32_t nthr) {
    for (uint32_t i = tid; i < 1024; i += nthr) (&sT[0][0])[i] = (&tabs->Te[0][0])[i];
}

// state words are the columns, little-endian (byte 0 = row 0)
__device__ __forceinline__ void aes256_encrypt(const uint32_t (*sT)[256], const AesKey &k, uint32_t &s0, uint32_t &s1, uint32_t &s2, uint32_t &s3) {
    s0 ^= k.rk[0]; s1 ^= k.rk[1]; s2 ^= k.rk[2]; s3 ^= k.rk[3];
#pragma unroll
    for (int r = 1; r < 14; r++) {
        const uint32_t t0 = sT[0][s0 & 0xFF] ^ sT[1][(s1 >> 8) & 0xFF] ^ sT[2][(s2 >> 16) & 0xFF] ^ sT[3][s3 >> 24] ^ k.rk[4 * r];
        const uint32_t t1 = sT[0][s1 & 0xFF] ^ sT[1][(s2 >> 8) & 0xFF] ^ sT[2][(s3 >> 16) & 0xFF] ^ sT[3][s0 >> 24] ^ k.rk[4 * r + 1];
        const uint32_t t2 = sT[0][s2 & 0xFF] ^ sT[1][(s3 >> 8) & 0xFF] ^ sT[2][(s0 >> 16) & 0xFF] ^ sT[3][s1 >> 24] ^ k.rk[4 * r + 2];
        const uint32_t t3 = sT[0][s3 & 0xFF] ^ sT[1][(s0 >> 8) & 0xFF] ^ sT[2][(s1 >> 16) & 0xFF] ^ sT[3][s2 >> 24] ^ k.rk[4 * r + 3];
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
    }
    // last round: SubBytes + ShiftRows only; S[x] is byte 1 of Te0[x]
#define SBX(v) ((sT[0][(v) & 0xFF] >> 8) & 0xFF)
    const uint32_t u0 = SBX(s0) | (SBX(s1 >> 8) << 8) | (SBX(s2 >> 16) << 16) | (SBX(s3 >> 24) << 24);
    const uint32_t u1 = SBX(s1) | (SBX(s2 >> 8) << 8) | (SBX(s3 >> 16) << 16) | (SBX(s0 >> 24) << 24);
    const uint32_t u2 = SBX(s2) | (SBX(s3 >> 8) << 8) | (SBX(s0 >> 16) << 16) | (SBX(s1 >> 24) << 24);
    const uint32_t u3 = SBX(s3) | (SBX(s0 >> 8) << 8) | (SBX(s1 >> 16) << 16) | (SBX(s2 >> 24) << 24);
#undef SBX
    s0 = u0 ^ k.rk[56]; s1 = u1 ^ k.rk[57]; s2 = u2 ^ k.rk[58]; s3 = u3 ^ k.rk[59];
}

// ------------------------------------------------------------------ CTR (Ctr128BE), in place; encrypt == decrypt
// Unit u covers buf[off .. off + len) = stream bytes [pos, pos + len) of the stream whose IV is ivs[iv_idx].
//
// Round-table layout of this kernel: the four round tables replicated over the 32 LDS banks -- entry x of bank b is word 32 x + b,
// and lane l only ever reads bank l & 31.  A wave's 64 look-ups are then conflict-free by construction (the two halves of a wave go
// through the LDS in different cycles), where four shared 1 KiB tables serialise random look-ups ~3.6-fold (measured on 10 000 x
// 1 MiB: 11.1 ms shared tables -> 8.0 ms one replicated table + rotations -> 7.6 ms four replicated tables, workgroups striding
// over the units so that the 128 KiB table image is built once per workgroup).  What is left is instruction issue: 882 VALU + 224
// LDS instructions per block and lane.
constexpr uint32_t CTR_THREADS = 1024;
constexpr uint32_t CTR_LDS = 4 * 256 * 32 * 4;                        // four bank-replicated tables: 128 KiB
__device__ __forceinline__ uint32_t rotl8(uint32_t w) { return __builtin_amdgcn_alignbit(w, w, 24); }
__device__ __forceinline__ uint32_t rotl16(uint32_t w) { return __builtin_amdgcn_alignbit(w, w, 16); }
__device__ __forceinline__ uint32_t rotl24(uint32_t w) { return __builtin_amdgcn_alignbit(w, w, 8); }

__global__ __launch_bounds__(CTR_THREADS)
void k_aes_ctr(const CipherUnit *__restrict__ units, uint32_t nunits, const uint8_t *__restrict__ ivs, const AesTabs *__restrict__ tabs,
               uint8_t *__restrict__ buf, const AesKey key0, const AesKey *__restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sR[];
    const uint32_t tid = threadIdx.x;
    {   // thread t owns table entry (t >> 8, t & 255): one global read, 32 bank copies
        const uint32_t v = tabs->Te[tid >> 8][tid & 0xFF];
        uint32_t *row = sR + (tid << 5);
#pragma unroll
        for (int b = 0; b < 32; b += 4) *(uint4 *)(row + b) = make_uint4(v, v, v, v);
    }
    const uint8_t *my = (const uint8_t *)sR + ((tid & 31) << 2);      // this lane's bank
#define TE(x) (*(const uint32_t *)(my + ((x) << 7)))
#define TE1(x) (*(const uint32_t *)(my + 32768 + ((x) << 7)))
#define TE2(x) (*(const uint32_t *)(my + 65536 + ((x) << 7)))
#define TE3(x) (*(const uint32_t *)(my + 98304 + ((x) << 7)))
    __syncthreads();
    // the table image is built once per workgroup; the workgroups stride over the units
    for (uint32_t ui = blockIdx.x; ui < nunits; ui += gridDim.x) {
    const CipherUnit u = units[ui];
    AesKey key = key0;
    if (keys) key = keys[u.iv_idx];                                   // GCM STREAM: every entry has its own stream key (wave-uniform load)
    const uint8_t *ivp = ivs + (size_t)u.iv_idx * 16;
    // the IV as a 128-bit big-endian number: hi = bytes 0..7, lo = bytes 8..15
    uint64_t iv_hi = 0, iv_lo = 0;
    for (int b = 0; b < 8; b++) { iv_hi = (iv_hi << 8) | ivp[b]; iv_lo = (iv_lo << 8) | ivp[8 + b]; }
    const uint64_t b0 = u.pos >> 4;                                   // first keystream block of the unit
    const uint64_t end = u.pos + u.len;
    const uint32_t nblk = (uint32_t)(((end + 15) >> 4) - b0);
    const int64_t base = (int64_t)u.off - (int64_t)(u.pos & 15);      // buf offset of keystream block b0's byte 0
    for (uint32_t j = tid; j < nblk; j += CTR_THREADS) {
        const uint64_t ctr = b0 + j;
        const uint64_t lo = iv_lo + ctr, hi = iv_hi + (lo < iv_lo ? 1u : 0u);
        uint32_t s0 = __builtin_bswap32((uint32_t)(hi >> 32)) ^ key.rk[0], s1 = __builtin_bswap32((uint32_t)hi) ^ key.rk[1];
        uint32_t s2 = __builtin_bswap32((uint32_t)(lo >> 32)) ^ key.rk[2], s3 = __builtin_bswap32((uint32_t)lo) ^ key.rk[3];
#pragma unroll
        for (int r = 1; r < 14; r++) {
            const uint32_t t0 = TE(s0 & 0xFF) ^ TE1((s1 >> 8) & 0xFF) ^ TE2((s2 >> 16) & 0xFF) ^ TE3(s3 >> 24) ^ key.rk[4 * r];
            const uint32_t t1 = TE(s1 & 0xFF) ^ TE1((s2 >> 8) & 0xFF) ^ TE2((s3 >> 16) & 0xFF) ^ TE3(s0 >> 24) ^ key.rk[4 * r + 1];
            const uint32_t t2 = TE(s2 & 0xFF) ^ TE1((s3 >> 8) & 0xFF) ^ TE2((s0 >> 16) & 0xFF) ^ TE3(s1 >> 24) ^ key.rk[4 * r + 2];
            const uint32_t t3 = TE(s3 & 0xFF) ^ TE1((s0 >> 8) & 0xFF) ^ TE2((s1 >> 16) & 0xFF) ^ TE3(s2 >> 24) ^ key.rk[4 * r + 3];
            s0 = t0; s1 = t1; s2 = t2; s3 = t3;
        }
        // last round: SubBytes + ShiftRows only; S[x] is byte 1 of Te0[x]
#define SBX(v) ((TE((v) & 0xFF) >> 8) & 0xFF)
        const uint32_t u0 = SBX(s0) | (SBX(s1 >> 8) << 8) | (SBX(s2 >> 16) << 16) | (SBX(s3 >> 24) << 24);
        const uint32_t u1 = SBX(s1) | (SBX(s2 >> 8) << 8) | (SBX(s3 >> 16) << 16) | (SBX(s0 >> 24) << 24);
        const uint32_t u2 = SBX(s2) | (SBX(s3 >> 8) << 8) | (SBX(s0 >> 16) << 16) | (SBX(s1 >> 24) << 24);
        const uint32_t u3 = SBX(s3) | (SBX(s0 >> 8) << 8) | (SBX(s1 >> 16) << 16) | (SBX(s2 >> 24) << 24);
#undef SBX
        s0 = u0 ^ key.rk[56]; s1 = u1 ^ key.rk[57]; s2 = u2 ^ key.rk[58]; s3 = u3 ^ key.rk[59];
        const int64_t a = base + (int64_t)j * 16;
        const uint64_t sp = (b0 + j) << 4;                            // stream position of this block
        if (sp >= u.pos && sp + 16 <= end) {
            U4u *q = (U4u *)(buf + a);
            U4u v = *q;
            v.x ^= s0; v.y ^= s1; v.z ^= s2; v.w ^= s3;
            *q = v;
        } else {                                                      // first / last block of the unit: only the bytes inside it
            const uint32_t ks[4] = {s0, s1, s2, s3};
            for (uint32_t b = 0; b < 16; b++) {
                const uint64_t p = sp + b;
                if (p >= u.pos && p < end) buf[a + b] ^= (uint8_t)(ks[b >> 2] >> (8 * (b & 3)));
            }
        }
    }
    } // units
#undef TE
#undef TE1
#undef TE2
#undef TE3
}

// ------------------------------------------------------------------ CBC encryption with PKCS#7 padding, in place, one lane per entry
// Entry e: plaintext buf[off .. off + len) becomes (len / 16 + 1) * 16 bytes of ciphertext at the same offset (the layout
// leaves room for the padding block).
__global__ __launch_bounds__(64)
void k_aes_cbc_enc(const CipherUnit *__restrict__ units, uint32_t n, const uint8_t *__restrict__ ivs, const AesTabs *__restrict__ tabs,
                   uint8_t *__restrict__ buf, const AesKey key) {
    __shared__ uint32_t sT[4][256];
    aes_load_tables(sT, tabs, threadIdx.x, 64);
    __syncthreads();
    const uint32_t e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n) return;
    const CipherUnit u = units[e];
    const U4u iv = *(const U4u *)(ivs + (size_t)u.iv_idx * 16);
    uint32_t c0 = iv.x, c1 = iv.y, c2 = iv.z, c3 = iv.w;
    const uint32_t nb = u.len / 16 + 1;
    uint8_t *p = buf + u.off;
    // the chain is serial, the loads are not: block b + 1 is requested before block b goes through the 14 rounds
    U4u nx = {0, 0, 0, 0};
    if (nb > 1) nx = *(const U4u *)p;
    for (uint32_t b = 0; b < nb; b++) {
        uint32_t x0, x1, x2, x3;
        if (b + 1 < nb) {
            x0 = nx.x; x1 = nx.y; x2 = nx.z; x3 = nx.w;
            if (b + 2 < nb) nx = *(const U4u *)(p + 16 * (size_t)(b + 1));
        } else {
            const uint32_t have = u.len - 16 * b, padv = 16 - have;
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t k2 = 0; k2 < 16; k2++) { const uint32_t byte = k2 < have ? p[16 * (size_t)b + k2] : padv; w[k2 >> 2] |= byte << (8 * (k2 & 3)); }
            x0 = w[0]; x1 = w[1]; x2 = w[2]; x3 = w[3];
        }
        c0 ^= x0; c1 ^= x1; c2 ^= x2; c3 ^= x3;
        aes256_encrypt(sT, key, c0, c1, c2, c3);
        U4u o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
        *(U4u *)(p + 16 * (size_t)b) = o;
    }
}


// ------------------------------------------------------------------ GHASH + tag of AES-GCM (NIST SP 800-38D), one workgroup per segment
// Cipher mode 2 of the reference ("GCM STREAM": GcmEncryptWriter::flush_segment, lib/src/cipher/gcm.rs:45-60 -- AES-256-GCM, 96-bit
// nonce, no associated data, detached 16-byte tag behind the segment).  The CTR part runs in k_aes_ctr (counter block nonce || 2 ...);
// this kernel computes  tag = GHASH_H(ciphertext) ^ E(K, nonce || 1)  over the ciphertext where it stands.
//
// GHASH = sum_i Y_i * H^(N - i) over the blocks Y_0 .. Y_(N-1) = [zero blocks in front (free), ciphertext blocks, length block],
// N a multiple of the 256 lanes.  Lane j runs Horner over Y_j, Y_(j+256), ... with the multiplier H^256 (4-bit table method, the
// table of the 16 nibble multiples of H^256 shared in LDS), then a log-step tree folds the lanes with H, H^2, H^4 .. H^128 (bitwise
// multiplies) and a last multiply by H finishes the sum.  128-bit values are four big-endian words (word 0 = bytes 0..3); "times x"
// is a right shift of that number with 0xE1 << 120 folded in when a bit drops out.
constexpr uint32_t GH_THREADS = 256;
struct G128 { uint32_t a, b, c, d; };
__device__ __forceinline__ G128 gx(const G128 &p, const G128 &q) { return G128{p.a ^ q.a, p.b ^ q.b, p.c ^ q.c, p.d ^ q.d}; }
__device__ __forceinline__ G128 g_mulx(const G128 &v) {
    const uint32_t carry = (0u - (v.d & 1u)) & 0xE1000000u;
    return G128{(v.a >> 1) ^ carry, (v.b >> 1) | (v.a << 31), (v.c >> 1) | (v.b << 31), (v.d >> 1) | (v.c << 31)};
}
__device__ G128 g_mul_bitwise(const G128 &x, G128 v) {                 // x * v, 128 conditional adds
    G128 z{0, 0, 0, 0};
    const uint32_t xw[4] = {x.a, x.b, x.c, x.d};
    for (int w = 0; w < 4; w++)
        for (int b = 31; b >= 0; b--) {
            const uint32_t m = 0u - ((xw[w] >> b) & 1u);
            z.a ^= v.a & m; z.b ^= v.b & m; z.c ^= v.c & m; z.d ^= v.d & m;
            v = g_mulx(v);
        }
    return z;
}
__device__ __forceinline__ uint32_t be32(uint32_t v) { return __builtin_bswap32(v); }

__global__ __launch_bounds__(GH_THREADS)
void k_gcm_tag(const GcmEntry *__restrict__ ents, uint8_t *__restrict__ buf, const uint8_t *__restrict__ expect, uint32_t *__restrict__ bad) {
    __shared__ G128 sM[16];                  // nibble multiples of H^256: sM[8] = H^256, sM[4] = H^256 x, sM[2], sM[1], the rest by addition
    __shared__ uint32_t sR[16];              // what the four bits dropped by a 4-bit shift fold back into the top 16 bits
    __shared__ G128 sHp[9];                  // H^(2^k), k = 0..8
    __shared__ G128 sAcc[GH_THREADS];
    const uint32_t tid = threadIdx.x;
    const GcmEntry e = ents[blockIdx.x];
    if (tid == 0) {
        G128 h{e.h[0], e.h[1], e.h[2], e.h[3]};
        sHp[0] = h;
        for (int k = 1; k <= 8; k++) { h = g_mul_bitwise(h, h); sHp[k] = h; }
        G128 m = h;                                                    // H^256
        sM[0] = G128{0, 0, 0, 0};
        sM[8] = m; m = g_mulx(m); sM[4] = m; m = g_mulx(m); sM[2] = m; m = g_mulx(m); sM[1] = m;
        for (int i = 2; i <= 8; i <<= 1) for (int j = 1; j < i; j++) sM[i + j] = gx(sM[i], sM[j]);
    }
    if (tid < 16) {
        uint32_t r = 0;
        for (int p = 0; p < 4; p++) if ((tid >> p) & 1) r ^= 0xE100u >> (3 - p);
        sR[tid] = r << 16;
    }
    __syncthreads();
    const uint32_t m = (e.len + 15) / 16;                              // ciphertext blocks; block m is the length block
    const uint32_t N = (m + 1 + GH_THREADS - 1) / GH_THREADS * GH_THREADS;
    const uint32_t pad = N - (m + 1);                                  // zero blocks in front
    const uint8_t *ct = buf + e.off;
    G128 acc{0, 0, 0, 0};
    for (uint32_t i = tid; i < N; i += GH_THREADS) {
        // acc = acc * H^256 (table method: 32 nibbles of acc, last nibble first), then + Y_i
        {
            const uint32_t xw[4] = {acc.a, acc.b, acc.c, acc.d};
            G128 z{0, 0, 0, 0};
#pragma unroll
            for (int n = 0; n < 32; n++) {                             // nibble n: n = 0 is the low nibble of byte 15
                const uint32_t nib = (xw[3 - (n >> 3)] >> (4 * (n & 7))) & 0xF;
                if (n) {
                    const uint32_t rem = z.d & 0xF;
                    z.d = (z.d >> 4) | (z.c << 28); z.c = (z.c >> 4) | (z.b << 28); z.b = (z.b >> 4) | (z.a << 28); z.a = (z.a >> 4) ^ sR[rem];
                }
                const G128 t = sM[nib];
                z.a ^= t.a; z.b ^= t.b; z.c ^= t.c; z.d ^= t.d;
            }
            acc = z;
        }
        if (i >= pad) {
            const uint32_t bi = i - pad;
            if (bi < m) {
                const uint64_t o = (uint64_t)bi * 16;
                if (o + 16 <= e.len) { const U4u v = *(const U4u *)(ct + o); acc.a ^= be32(v.x); acc.b ^= be32(v.y); acc.c ^= be32(v.z); acc.d ^= be32(v.w); }
                else {
                    uint32_t w[4] = {0, 0, 0, 0};
                    for (uint32_t k = 0; o + k < e.len; k++) w[k >> 2] |= (uint32_t)ct[o + k] << (24 - 8 * (k & 3));
                    acc.a ^= w[0]; acc.b ^= w[1]; acc.c ^= w[2]; acc.d ^= w[3];
                }
            } else {                                                   // len(A) = 0 || len(C) in bits
                const uint64_t bits = (uint64_t)e.len * 8;
                acc.c ^= (uint32_t)(bits >> 32); acc.d ^= (uint32_t)bits;
            }
        }
    }
    // fold the lanes: P = sum_j acc_j * H^(255 - j)
    sAcc[tid] = acc;
    __syncthreads();
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t st = 1u << k;
        if ((tid & (2 * st - 1)) == 0) sAcc[tid] = gx(g_mul_bitwise(sAcc[tid], sHp[k]), sAcc[tid + st]);
        __syncthreads();
    }
    if (tid == 0) {
        const G128 s = g_mul_bitwise(sAcc[0], sHp[0]);
        U4u t; t.x = be32(s.a ^ e.ej0[0]); t.y = be32(s.b ^ e.ej0[1]); t.z = be32(s.c ^ e.ej0[2]); t.w = be32(s.d ^ e.ej0[3]);
        if (expect) {                                                  // read side: the tag is compared, nothing is written
            const U4u x = *(const U4u *)(expect + 16 * (size_t)blockIdx.x);
            if (((t.x ^ x.x) | (t.y ^ x.y) | (t.z ^ x.z) | (t.w ^ x.w)) != 0) atomicAdd(bad, 1u);
        } else *(U4u *)(buf + e.off + e.len) = t;
    }
}

void launch_gcm_tag(const GcmEntry *ents, uint32_t n, uint8_t *buf, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_gcm_tag, dim3(n), dim3(GH_THREADS), 0, st, ents, buf, (const uint8_t *)nullptr, (uint32_t *)nullptr);
}
void launch_gcm_verify(const GcmEntry *ents, uint32_t n, const uint8_t *buf, const uint8_t *expect, uint32_t *bad, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_gcm_tag, dim3(n), dim3(GH_THREADS), 0, st, ents, const_cast<uint8_t *>(buf), expect, bad);
}


// ------------------------------------------------------------------ CBC decryption (read side: DecryptCbcAes256Reader, lib/src/cipher/block/read.rs)
// P_j = D(C_j) ^ C_(j-1), C_(-1) = IV: every block is independent once its predecessor's ciphertext is at hand.  One workgroup per
// entry, 256 blocks per step, in place: a step loads its ciphertext (and the block in front of it) before anything is written, the
// last ciphertext block of a step is carried to the next one through LDS.  The PKCS#7 padding is checked at the end and the
// plaintext length reported (0xFFFFFFFF: bad length or padding -- wrong key or damage).  `key` holds the round keys of the
// equivalent inverse cipher (FIPS-197 5.3.5).
__global__ __launch_bounds__(256)
void k_aes_cbc_dec(const CipherUnit *__restrict__ units, const uint8_t *__restrict__ ivs, const AesDecTabs *__restrict__ tabs,
                   uint8_t *__restrict__ buf, const AesKey key, uint32_t *__restrict__ plain_len) {
    __shared__ uint32_t sD[4][256], sS[256];
    __shared__ uint32_t carry[4];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 1024; i += 256) (&sD[0][0])[i] = (&tabs->Td[0][0])[i];
    sS[tid] = tabs->Sd[tid];
    const CipherUnit u = units[blockIdx.x];
    if (tid == 0) { const U4u iv = *(const U4u *)(ivs + (size_t)u.iv_idx * 16); carry[0] = iv.x; carry[1] = iv.y; carry[2] = iv.z; carry[3] = iv.w; }
    __syncthreads();
    if (u.len == 0 || (u.len & 15)) { if (tid == 0) plain_len[blockIdx.x] = 0xFFFFFFFFu; return; }
    const uint32_t nb = u.len >> 4;
    uint8_t *p = buf + u.off;
    for (uint32_t base = 0; base < nb; base += 256) {
        const uint32_t j = base + tid;
        U4u cj = {0, 0, 0, 0}, pv = {0, 0, 0, 0};
        if (j < nb) {
            cj = *(const U4u *)(p + 16 * (size_t)j);
            if (tid) pv = *(const U4u *)(p + 16 * (size_t)(j - 1)); else { pv.x = carry[0]; pv.y = carry[1]; pv.z = carry[2]; pv.w = carry[3]; }
        }
        __syncthreads();                                             // everyone holds its ciphertext: the step may overwrite it now
        if (j < nb) {
            uint32_t s0 = cj.x ^ key.rk[0], s1 = cj.y ^ key.rk[1], s2 = cj.z ^ key.rk[2], s3 = cj.w ^ key.rk[3];
#pragma unroll
            for (int r = 1; r < 14; r++) {
                const uint32_t t0 = sD[0][s0 & 0xFF] ^ sD[1][(s3 >> 8) & 0xFF] ^ sD[2][(s2 >> 16) & 0xFF] ^ sD[3][s1 >> 24] ^ key.rk[4 * r];
                const uint32_t t1 = sD[0][s1 & 0xFF] ^ sD[1][(s0 >> 8) & 0xFF] ^ sD[2][(s3 >> 16) & 0xFF] ^ sD[3][s2 >> 24] ^ key.rk[4 * r + 1];
                const uint32_t t2 = sD[0][s2 & 0xFF] ^ sD[1][(s1 >> 8) & 0xFF] ^ sD[2][(s0 >> 16) & 0xFF] ^ sD[3][s3 >> 24] ^ key.rk[4 * r + 2];
                const uint32_t t3 = sD[0][s3 & 0xFF] ^ sD[1][(s2 >> 8) & 0xFF] ^ sD[2][(s1 >> 16) & 0xFF] ^ sD[3][s0 >> 24] ^ key.rk[4 * r + 3];
                s0 = t0; s1 = t1; s2 = t2; s3 = t3;
            }
            const uint32_t u0 = sS[s0 & 0xFF] | (sS[(s3 >> 8) & 0xFF] << 8) | (sS[(s2 >> 16) & 0xFF] << 16) | (sS[s1 >> 24] << 24);
            const uint32_t u1 = sS[s1 & 0xFF] | (sS[(s0 >> 8) & 0xFF] << 8) | (sS[(s3 >> 16) & 0xFF] << 16) | (sS[s2 >> 24] << 24);
            const uint32_t u2 = sS[s2 & 0xFF] | (sS[(s1 >> 8) & 0xFF] << 8) | (sS[(s0 >> 16) & 0xFF] << 16) | (sS[s3 >> 24] << 24);
            const uint32_t u3 = sS[s3 & 0xFF] | (sS[(s2 >> 8) & 0xFF] << 8) | (sS[(s1 >> 16) & 0xFF] << 16) | (sS[s0 >> 24] << 24);
            U4u o; o.x = u0 ^ key.rk[56] ^ pv.x; o.y = u1 ^ key.rk[57] ^ pv.y; o.z = u2 ^ key.rk[58] ^ pv.z; o.w = u3 ^ key.rk[59] ^ pv.w;
            *(U4u *)(p + 16 * (size_t)j) = o;
            if (tid == 255) { carry[0] = cj.x; carry[1] = cj.y; carry[2] = cj.z; carry[3] = cj.w; }
            if (j + 1 == nb) {                                       // PKCS#7: the last byte names the padding length, all padding bytes repeat it
                const uint32_t pad = o.w >> 24;
                bool ok = pad >= 1 && pad <= 16;
                const uint32_t w[4] = {o.x, o.y, o.z, o.w};
                for (uint32_t k = 0; ok && k < pad; k++) { const uint32_t bi = 15 - k; ok = ((w[bi >> 2] >> (8 * (bi & 3))) & 0xFF) == pad; }
                plain_len[blockIdx.x] = ok ? u.len - pad : 0xFFFFFFFFu;
            }
        }
        __syncthreads();                                             // the carry is in place before the next step's lane 0 reads it
    }
}
void launch_aes_cbc_dec(const CipherUnit *units, uint32_t n, const uint8_t *ivs, const AesDecTabs *tabs, uint8_t *buf, const AesKey &dkey, uint32_t *plain_len, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_aes_cbc_dec, dim3(n), dim3(256), 0, st, units, ivs, tabs, buf, dkey, plain_len);
}

void launch_aes_ctr(const CipherUnit *units, uint32_t n, const uint8_t *ivs, const AesTabs *tabs, uint8_t *buf, const AesKey &key, const AesKey *keys, hipStream_t st) {
    static const hipError_t attr_set = hipFuncSetAttribute((const void *)k_aes_ctr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CTR_LDS);   // once, thread-safe
    (void)attr_set;
    // one workgroup per CU at a time (128 KiB of LDS); a few per CU in the grid even out the ragged units
    if (n) hipLaunchKernelGGL(k_aes_ctr, dim3(n < 1024 ? n : 1024), dim3(CTR_THREADS), CTR_LDS, st, units, n, ivs, tabs, buf, key, keys);
}
void launch_aes_cbc_enc(const CipherUnit *units, uint32_t n, const uint8_t *ivs, const AesTabs *tabs, uint8_t *buf, const AesKey &key, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_aes_cbc_enc, dim3((n + 63) / 64), dim3(64), 0, st, units, n, ivs, tabs, buf, key);
}

} // namespace pna
