// k_frame.hip -- PNA entry framing in HBM: chunk headers, the FDAT CRC-32 and FEND around a payload that the write
// kernels have already placed at its final archive offset.  One 256-thread workgroup per entry.
//
// Mirrors write_chunk (lib/src/io.rs:183-197: length BE | type | data | crc32(type || data) BE, crc = chunk_crc,
// lib/src/format/chunk.rs:7-12) for the chunks of NormalEntry::write_chunks_to (lib/src/entry.rs:895-911).
//
// CRC-32 (reflected 0xEDB88320) of a long message (chunk type || payload) on 256 lanes.  With R(s, M) the raw register update (no init / final
// xor) the code relies on three identities of the linear map R:
//   (1) R(0xFFFFFFFF, M) = R(0, M ^ FF FF FF FF 00 00 ...)        |M| >= 4: the init value folds into the first 4 bytes
//   (2) R(0, 0^k || M)   = R(0, M)                                zero bytes in front of the message are free
//   (3) R(0, A || B)     = Z_|B|(R(0, A)) ^ R(0, B)               Z_k = "append k zero bytes" = multiply by x^(8k) mod P
// The message "FDAT" || payload is front-padded (2) to a whole number of 16 KiB tiles.  Lane t owns the 64-byte piece t of
// every tile; between tiles its state is advanced by Z_16320 (four table look-ups), so after the last tile lane t holds
// the contribution of its pieces as if piece t of the LAST tile were the end of the message.  A log-step tree applies (3)
// with the fixed shifts 64 * 2^j to fold the 256 lane states into R(0, M').
#include <hip/hip_runtime.h>
#include "pna_dev.h"

namespace pna {

constexpr uint32_t FR_THREADS = 256;
constexpr uint32_t FR_TILE = FR_THREADS * 64;              // 16 KiB of message per step
constexpr uint32_t FR_TILE_DW = FR_TILE / 4 + 4;           // + one 16-byte word for the misalignment of the source

__device__ __forceinline__ uint32_t lds_pad(uint32_t d) { return d + (d >> 4); }   // 17-dword row pitch: lane stride 16 dwords -> conflict-free

// a * b mod P, reflected bit order (bit 31 = x^0)
__device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}

__global__ __launch_bounds__(FR_THREADS)
void k_frame(const FrameDesc *__restrict__ fd, const uint8_t *__restrict__ blob, const CrcTabs *__restrict__ ct,
             uint8_t *__restrict__ dst, uint64_t cap16, uint32_t fend_crc, uint32_t ty_x, uint32_t with_fend, uint32_t *__restrict__ verify,
             uint32_t nent, uint32_t epw) {
    __shared__ uint32_t sT[4][256], sZ[4][256];
    __shared__ uint32_t tile[FR_TILE_DW + FR_TILE_DW / 16 + 8];
    __shared__ uint32_t part[FR_THREADS];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 1024; i += FR_THREADS) { (&sT[0][0])[i] = (&ct->T[0][0])[i]; (&sZ[0][0])[i] = (&ct->Z[0][0])[i]; }
    // `epw` consecutive descriptors per workgroup: with 10^5 .. 10^6 small entries the 8 KiB of tables are then loaded once per epw entries
    for (uint32_t ent = blockIdx.x * epw; ent < nent && ent < (blockIdx.x + 1) * epw; ent++) {
    __syncthreads();                                                 // (the previous entry's `part` and `tile` are done with)
    const FrameDesc d = fd[ent];
    // verify != NULL: read side (read_chunk, lib/src/io.rs:117-149) -- nothing is written, the CRC is compared with the stored one
    if (!verify) for (uint32_t i = tid; i < d.prefix_len; i += FR_THREADS) dst[d.arc_off + i] = blob[d.prefix_off + i];
    if (d.pad & 1) continue;                                         // record without a data chunk: the prefix is all of it

    const uint64_t pay = d.arc_off + d.prefix_len;                   // payload offset in dst
    // pad bit 4 (with `verify`): a PIECE of a chunk -- the raw register R(0, M) of its bytes goes to verify[ent], the host chains the pieces by identity (3); the
    // first piece carries the type bytes and the folded init value like a whole chunk, pad bit 8 = a later piece: its bytes alone (the streaming solid create:
    // an inner entry's data chunk reaches the device window by window)
    const uint32_t tlen = (d.pad & 8) ? 0u : 4u;
    const uint32_t n = tlen + d.payload_len;                         // "FDAT" || payload  (payload_len <= 2^32 - 5 checked by the host)
    const uint32_t ntile = (n + FR_TILE - 1) / FR_TILE;
    const uint32_t pad = ntile * FR_TILE - n;                        // zero bytes put in front, < FR_TILE
    const int64_t base = (int64_t)pay - (int64_t)tlen - (int64_t)pad; // dst offset of message position 0 (may be negative)
    const uint32_t a = (uint32_t)(base & 15);                        // two's complement: correct for negative base too
    uint32_t state = 0;

    for (uint32_t k = 0; k < ntile; k++) {
        const int64_t A = base + (int64_t)k * FR_TILE - a;           // multiple of 16
        __syncthreads();
        for (uint32_t i = tid; i < FR_TILE / 16 + 1; i += FR_THREADS) {
            const int64_t o = A + (int64_t)i * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            // (16-byte pieces that lie wholly in the zero padding in front of a short message are not fetched: for a 2 KiB payload that is
            // 7/8 of the tile)
            const bool in_pad = k == 0 && (uint64_t)i * 16 + 16 <= (uint64_t)pad + a;
            if (!in_pad && o >= 0 && (uint64_t)o + 16 <= cap16) v = *reinterpret_cast<const uint4 *>(dst + o);
            tile[lds_pad(4 * i)] = v.x; tile[lds_pad(4 * i + 1)] = v.y; tile[lds_pad(4 * i + 2)] = v.z; tile[lds_pad(4 * i + 3)] = v.w;
        }
        __syncthreads();
        uint32_t w[17];
        const uint32_t d0 = (a >> 2) + tid * 16;
#pragma unroll
        for (int j = 0; j < 17; j++) w[j] = tile[lds_pad(d0 + j)];
        const uint32_t sh = (a & 3) * 8;
#pragma unroll
        for (int j = 0; j < 16; j++) w[j] = (uint32_t)((((uint64_t)w[j + 1] << 32) | w[j]) >> sh);
        const uint32_t p0 = k * FR_TILE + tid * 64;                  // message position of this piece
        if (p0 < pad + tlen) {                                       // first tile only: zero padding and the type bytes
#pragma unroll
            for (int j = 0; j < 16; j++) {
                uint32_t m = 0, x = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t p = p0 + 4 * j + b;
                    if (p >= pad + tlen) m |= 0xFFu << (8 * b);
                    else if (p >= pad) x |= ((ty_x >> (8 * (p - pad))) & 0xFFu) << (8 * b);
                }
                w[j] = (w[j] & m) | x;
            }
        }
        state = sZ[0][state & 0xFF] ^ sZ[1][(state >> 8) & 0xFF] ^ sZ[2][(state >> 16) & 0xFF] ^ sZ[3][state >> 24];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t c = state ^ w[j];
            state = sT[3][c & 0xFF] ^ sT[2][(c >> 8) & 0xFF] ^ sT[1][(c >> 16) & 0xFF] ^ sT[0][c >> 24];
        }
    }
    part[tid] = state;
    __syncthreads();
    for (uint32_t j = 0; j < 8; j++) {
        const uint32_t st = 1u << j;
        // (a zero state stays zero: the lanes in front of a short message -- most of them for a 4 KiB entry -- skip the multiply)
        if ((tid & (2 * st - 1)) == 0) { const uint32_t pv = part[tid]; part[tid] = (pv ? gf2_mulmod(ct->sh[j], pv) : 0u) ^ part[tid + st]; }
        __syncthreads();
    }
    if (verify && (d.pad & 4)) { if (tid == 0) verify[ent] = part[0]; continue; }
    if (verify) {
        if (tid == 0) {
            const uint8_t *q = dst + pay + d.payload_len;
            const uint32_t stored = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
            if (stored != ~part[0]) { atomicAdd(&verify[0], 1u); atomicMin(&verify[1], ent); }
        }
        continue;
    }
    if (tid < ((with_fend && !(d.pad & 2)) ? 16u : 4u)) {                // pad bit 1: another data chunk of the same entry follows, no FEND yet
        const uint32_t crc = ~part[0];
        // crc BE | 00 00 00 00 | "FEND" | crc("FEND") BE
        const uint32_t fe = 0x444E4546u;                              // "FEND" little-endian
        uint8_t v;
        if (tid < 4) v = (uint8_t)(crc >> (24 - 8 * tid));
        else if (tid < 8) v = 0;
        else if (tid < 12) v = (uint8_t)(fe >> (8 * (tid - 8)));
        else v = (uint8_t)(fend_crc >> (24 - 8 * (tid - 12)));
        dst[pay + d.payload_len + tid] = v;
    }
    }
}

// ------------------------------------------------------------------ k_frame_wave : the same for SMALL payloads, one WAVE per entry
// With 10^5 .. 10^6 entries of a few KiB the workgroup form above spends its time on what does not depend on the payload: 8 KiB of tables per
// workgroup, a 16 KiB tile of which 7/8 are padding, eight fold steps with a barrier and a 32-step multiply each.  Here a workgroup's four waves
// take an entry each (the slice-by-4 table is loaded once per workgroup).  The message "type || payload" (n bytes) is anchored at its END: lane t
// owns the 64 m bytes that end 64 m (63 - t) bytes before the message's end (m = ceil(n / 4096) pieces of 64 bytes per lane; positions in front of
// the message are zero bytes, which are free -- identity (2) above), reads them straight from memory (16-byte loads at any byte address), and its
// state is folded in by ONE multiplication with x^(8 * 64 m (63 - t)) from a table (m <= 4: payloads up to 16 380 bytes; beyond that the power
// is computed on the spot -- correct, slow, and not what the host sends here) and an XOR across the wave.
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u ld16u(const uint8_t *p) { v4u v; __builtin_memcpy(&v, p, 16); return v; }   // 16 bytes at any byte address
__device__ __forceinline__ uint32_t gf2_xpow_dev(uint64_t e) {
    uint32_t r = 0x80000000u, base = 0x40000000u;
    while (e) { if (e & 1) r = gf2_mulmod(base, r); base = gf2_mulmod(base, base); e >>= 1; }
    return r;
}
__global__ __launch_bounds__(256)
void k_frame_wave(const FrameDesc *__restrict__ fd, const uint8_t *__restrict__ blob, const CrcTabs *__restrict__ ct, uint8_t *__restrict__ dst,
                  uint32_t fend_crc, uint32_t ty_x, uint32_t with_fend, uint32_t nent, uint32_t *__restrict__ verify) {
    __shared__ uint32_t sT[4][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < 1024; i += 256) (&sT[0][0])[i] = (&ct->T[0][0])[i];
    __syncthreads();
    const uint32_t ent = blockIdx.x * 4 + (tid >> 6);
    if (ent >= nent) return;
    const FrameDesc d = fd[ent];
    // verify != NULL: read side (read_chunk, lib/src/io.rs:117-149) -- nothing is written, the CRC is compared with the stored one
    if (!verify) for (uint32_t i = lane; i < d.prefix_len; i += 64) dst[d.arc_off + i] = blob[d.prefix_off + i];
    if (d.pad & 1) return;                                           // record without a data chunk: the prefix is all of it
    const uint64_t pay = d.arc_off + d.prefix_len;                   // payload offset in dst
    const uint32_t n = 4 + d.payload_len;                            // "FDAT" || payload
    const uint32_t m = (n + 4095) >> 12;                             // pieces of 64 bytes per lane (uniform)
    const int64_t s0 = (int64_t)n - (int64_t)64 * m * (64 - lane);   // message position of the lane's first byte (negative: in front of the message)
    const uint8_t *msg = dst + pay - 4;                              // address of message position 0 (the type bytes come from ty_x, not from memory)
    uint32_t state = 0;
    for (uint32_t i = 0; i < m; i++) {
        const int64_t P0 = s0 + (int64_t)64 * i;
        if (P0 + 64 <= 0) continue;                                  // all zero, and the state still is
        uint32_t w[16];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            v4u v = 0;
            if (P0 + 16 * g + 16 > 0) v = ld16u(msg + P0 + 16 * g);     // (a group that straddles position 0 reads up to 15 bytes in front of the type bytes: inside the entry's prefix / the chunk before)
            w[4 * g] = v.x; w[4 * g + 1] = v.y; w[4 * g + 2] = v.z; w[4 * g + 3] = v.w;
        }
        if (P0 < 4) {                                                // the dwords that hold positions below 4: zero in front of the message, the type bytes (already XORed with the CRC's initial value) at 0 .. 3
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int32_t P = (int32_t)P0 + 4 * j;               // (P0 > -64 here)
                if (P >= 4) continue;
                const uint32_t k = P > 0 ? (uint32_t)P : 0u;          // payload bytes in the dword: its top k
                const uint32_t keep = k ? 0xFFFFFFFFu << (8 * (4 - k)) : 0u;
                const uint32_t tyw = P >= 0 ? ty_x >> (8 * P) : (P > -4 ? ty_x << (8 * -P) : 0u);
                w[j] = (w[j] & keep) | tyw;
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t c = state ^ w[j];
            state = sT[3][c & 0xFF] ^ sT[2][(c >> 8) & 0xFF] ^ sT[1][(c >> 16) & 0xFF] ^ sT[0][c >> 24];
        }
    }
    uint32_t x = 0;
    if (state) x = gf2_mulmod(m <= 4 ? ct->pw[m - 1][63 - lane] : gf2_xpow_dev((uint64_t)8 * 64 * m * (63 - lane)), state);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x ^= (uint32_t)__shfl_xor((int)x, o);
    if (verify) {
        if (lane == 0) {
            const uint8_t *q = dst + pay + d.payload_len;
            const uint32_t stored = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
            if (stored != ~x) { atomicAdd(&verify[0], 1u); atomicMin(&verify[1], ent); }
        }
        return;
    }
    if (lane < ((with_fend && !(d.pad & 2)) ? 16u : 4u)) {             // pad bit 1: another data chunk of the same entry follows, no FEND yet
        const uint32_t crc = ~x;
        const uint32_t fe = 0x444E4546u;                              // "FEND" little-endian
        uint8_t v;
        if (lane < 4) v = (uint8_t)(crc >> (24 - 8 * lane));
        else if (lane < 8) v = 0;
        else if (lane < 12) v = (uint8_t)(fe >> (8 * (lane - 8)));
        else v = (uint8_t)(fend_crc >> (24 - 8 * (lane - 12)));
        dst[pay + d.payload_len + lane] = v;
    }
}

// ------------------------------------------------------------------ k_place : byte ranges to arbitrary offsets
// One workgroup per piece (<= 1 MiB) of an entry: dst[dst_off .. +len) = src[src_off .. +len).  src_off is 16-byte aligned,
// dst_off is not: whole destination dwords are assembled from two aligned source dwords, the <= 3 bytes at either end are
// written singly.
struct PlaceDesc { uint64_t src_off, dst_off; uint32_t len, pad; };
__global__ __launch_bounds__(256)
void k_place(const PlaceDesc *__restrict__ pd, const uint8_t *__restrict__ src, uint8_t *__restrict__ dst) {
    const PlaceDesc d = pd[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const uint8_t *s = src + d.src_off;
    uint8_t *o = dst + d.dst_off;
    const uint32_t head = (uint32_t)((4 - (d.dst_off & 3)) & 3) < d.len ? (uint32_t)((4 - (d.dst_off & 3)) & 3) : d.len;
    if (tid < head) o[tid] = s[tid];
    const uint32_t body = (d.len - head) >> 2;                       // whole destination dwords
    const uint32_t *s32 = (const uint32_t *)s;                       // aligned (src_off % 16 == 0)
    uint32_t *o32 = (uint32_t *)(o + head);
    const uint32_t sh = head * 8;                                    // source byte head + 4 i = dword i, shifted by head bytes
    for (uint32_t i = tid; i < body; i += 256) {
        const uint32_t lo = s32[i], hi = head ? s32[i + 1] : 0u;     // i + 1 stays inside the entry's 16-byte padded slot
        o32[i] = head ? __builtin_amdgcn_alignbit(hi, lo, sh) : lo;
    }
    const uint32_t done = head + 4 * body;
    if (tid < d.len - done) o[done + tid] = s[done + tid];
}

void launch_place(const void *pd, uint32_t n, const uint8_t *src, uint8_t *dst, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_place, dim3(n), dim3(256), 0, st, (const PlaceDesc *)pd, src, dst);
}

// ty: chunk type as it stands in the file (e.g. "FDAT"); with_fend: append an FEND chunk behind the data chunk's CRC
// max_payload: an upper bound of the launch's payload lengths if the host knows one that is small (0: none) -- up to 16 380 bytes the wave-per-entry form runs
void launch_frame(const FrameDesc *fd, uint32_t nentry, const uint8_t *blob, const CrcTabs *ct, uint8_t *dst, uint64_t cap16,
                  uint32_t fend_crc, const char ty[4], bool with_fend, hipStream_t st, uint32_t max_payload) {
    const uint32_t ty_le = (uint32_t)(uint8_t)ty[0] | ((uint32_t)(uint8_t)ty[1] << 8) | ((uint32_t)(uint8_t)ty[2] << 16) | ((uint32_t)(uint8_t)ty[3] << 24);
    if (nentry && max_payload && max_payload <= 16380u) {
        hipLaunchKernelGGL(k_frame_wave, dim3((nentry + 3) / 4), dim3(256), 0, st, fd, blob, ct, dst, fend_crc, ~ty_le, with_fend ? 1u : 0u, nentry, (uint32_t *)nullptr);
        return;
    }
    const uint32_t epw = 1u;                                    // (4 entries per workgroup measured SLOWER for 10^6 small entries: 9.1 vs 8.0 ms -- more workgroups in flight hide the per-entry chain better)
    if (nentry) hipLaunchKernelGGL(k_frame, dim3((nentry + epw - 1) / epw), dim3(FR_THREADS), 0, st, fd, blob, ct, dst, cap16, fend_crc, ~ty_le, with_fend ? 1u : 0u, (uint32_t *)nullptr, nentry, epw);
}
// Read side: CRC-32 of n data chunks of type `ty` where they stand in buf (FrameDesc: arc_off = chunk start, prefix_len = 8,
// payload_len = chunk length); verify[0] counts mismatches, verify[1] keeps the lowest failing descriptor index.
void launch_frame_verify(const FrameDesc *fd, uint32_t n, const CrcTabs *ct, const uint8_t *buf, uint64_t cap16, const char ty[4], uint32_t *verify, hipStream_t st, uint32_t max_payload) {
    const uint32_t ty_le = (uint32_t)(uint8_t)ty[0] | ((uint32_t)(uint8_t)ty[1] << 8) | ((uint32_t)(uint8_t)ty[2] << 16) | ((uint32_t)(uint8_t)ty[3] << 24);
    if (n && max_payload && max_payload <= 16380u) {               // many small chunks: a wave per chunk
        hipLaunchKernelGGL(k_frame_wave, dim3((n + 3) / 4), dim3(256), 0, st, fd, (const uint8_t *)nullptr, ct, const_cast<uint8_t *>(buf), 0u, ~ty_le, 0u, n, verify);
        return;
    }
    const uint32_t epw = 1u;
    if (n) hipLaunchKernelGGL(k_frame, dim3((n + epw - 1) / epw), dim3(FR_THREADS), 0, st, fd, (const uint8_t *)nullptr, ct, const_cast<uint8_t *>(buf), cap16, 0u, ~ty_le, 0u, verify, n, epw);
}

// Pieces of chunks (FrameDesc::pad bits 4 / 8, above): raw CRC registers into states[0 .. n)
void launch_frame_pieces(const FrameDesc *fd, uint32_t n, const CrcTabs *ct, const uint8_t *buf, uint64_t cap16, const char ty[4], uint32_t *states, hipStream_t st) {
    const uint32_t ty_le = (uint32_t)(uint8_t)ty[0] | ((uint32_t)(uint8_t)ty[1] << 8) | ((uint32_t)(uint8_t)ty[2] << 16) | ((uint32_t)(uint8_t)ty[3] << 24);
    if (n) hipLaunchKernelGGL(k_frame, dim3(n), dim3(FR_THREADS), 0, st, fd, (const uint8_t *)nullptr, ct, const_cast<uint8_t *>(buf), cap16, 0u, ~ty_le, 0u, states, n, 1u);
}
// Big-endian CRC fields written where the host says: patch i puts the bytes j of crc[i] with bit j of mask[i] set at dst[off[i] + j] (a field may straddle the
// buffer's end: the host then hands the other bytes to the next window)
struct CrcPatch { int64_t off; uint32_t crc, mask; };
__global__ void k_crc_patch(const CrcPatch *__restrict__ p, uint32_t n, uint8_t *__restrict__ dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CrcPatch q = p[i];
    for (int j = 0; j < 4; j++) if ((q.mask >> j) & 1) dst[q.off + j] = (uint8_t)(q.crc >> (24 - 8 * j));
}
void launch_crc_patch(const void *patches, uint32_t n, uint8_t *dst, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_crc_patch, dim3((n + 255) / 256), dim3(256), 0, st, (const CrcPatch *)patches, n, dst);
}

// ------------------------------------------------------------------ k_gather : byte ranges from arbitrary offsets to 16-byte aligned ones
// The reverse of k_place (read side): piece p = src[src_off .. +len) goes to dst[dst_off .. +len); dst_off is a multiple of 16 only for
// the first piece of a stream, so both sides are treated as unaligned 16-byte accesses, the tail bytewise.
__global__ __launch_bounds__(256)
void k_gather(const PlaceDesc *__restrict__ pd, const uint8_t *__restrict__ src, uint8_t *__restrict__ dst) {
    struct __attribute__((packed, aligned(1))) U4 { uint32_t x, y, z, w; };
    const PlaceDesc d = pd[blockIdx.x];
    const uint8_t *s = src + d.src_off;
    uint8_t *o = dst + d.dst_off;
    const uint32_t n16 = d.len >> 4;
    for (uint32_t i = threadIdx.x; i < n16; i += 256) *(U4 *)(o + 16 * (size_t)i) = *(const U4 *)(s + 16 * (size_t)i);
    const uint32_t done = n16 << 4;
    if (threadIdx.x < d.len - done) o[done + threadIdx.x] = s[done + threadIdx.x];
}
// ------------------------------------------------------------------ k_layout : the archive layout of a sub-batch, on the device
// For plain file entries (one FDAT chunk each) the record of entry e is  prefix_e | payload_e | crc | FEND : its place follows from the prefix
// lengths (host: they depend on names and sizes only) and the compressed sizes (device: the segments' offsets in the packed stream, k_scan).  One
// entry per thread: sizes and a scan inside each workgroup of 1 024 entries (k_layout_a), a scan over the workgroups' sums (k_scan), then every
// thread fills in what the write kernels and k_frame need (k_layout_c) -- FrameDesc::arc_off / payload_len, the FDAT length inside the prefix bytes, every segment's destination.  The host
// reads back one number (the sub-batch's length) at the END of the call instead of all segment sizes in the middle of it; for 125 000 entries of
// 4 KiB that wait and the layout loop behind it were a third of the sub-batch's time.
// workgroup-wide inclusive scan of one u64 per thread (1 024 threads, log steps over two LDS buffers); returns the inclusive value, *wg_total the sum
__device__ __forceinline__ uint64_t wg_scan_1024(uint64_t v, uint64_t (*part)[1024], uint64_t *wg_total) {
    const uint32_t tid = threadIdx.x;
    uint32_t cur = 0;
    part[0][tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        part[cur ^ 1][tid] = part[cur][tid] + (tid >= d ? part[cur][tid - d] : 0ull);
        cur ^= 1;
        __syncthreads();
    }
    *wg_total = part[cur][1023];
    return part[cur][tid];
}
// phase A: one entry per thread; its record's size -> offset inside the workgroup's 1 024 entries (ent_off), the workgroup's total (wg_sum)
__global__ __launch_bounds__(1024)
void k_layout_a(const FrameDesc *__restrict__ fd, const uint32_t *__restrict__ entry_seg, const uint64_t *__restrict__ seg_off, uint32_t nentry,
                uint64_t *__restrict__ ent_off, uint64_t *__restrict__ wg_sum) {
    __shared__ uint64_t part[2][1024];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    uint64_t s = 0;
    if (i < nentry) s = (uint64_t)fd[i].prefix_len + (seg_off[entry_seg[i + 1]] - seg_off[entry_seg[i]]) + 16;
    uint64_t tot;
    const uint64_t incl = wg_scan_1024(s, part, &tot);
    if (i < nentry) ent_off[i] = incl - s;
    if (threadIdx.x == 0) wg_sum[blockIdx.x] = tot;
}
// phase C (behind a scan of wg_sum -> wg_off): the final places
__global__ __launch_bounds__(1024)
void k_layout_c(FrameDesc *__restrict__ fd, uint8_t *__restrict__ blob, const uint32_t *__restrict__ entry_seg, const uint64_t *__restrict__ seg_off,
                uint32_t nentry, uint32_t nseg, uint64_t out_base, const uint64_t *__restrict__ wg_off, uint32_t nwg,
                uint64_t *__restrict__ segdst, uint64_t *__restrict__ ent_off, uint64_t *__restrict__ total) {
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (i < nentry) {
        const uint64_t pos = out_base + wg_off[blockIdx.x] + ent_off[i];
        const uint32_t s0 = entry_seg[i], s1 = entry_seg[i + 1];
        const uint64_t base = seg_off[s0], plen = seg_off[s1] - base;
        FrameDesc d = fd[i];
        d.arc_off = pos; d.payload_len = (uint32_t)plen;
        fd[i] = d;
        uint8_t *lenf = blob + d.prefix_off + d.prefix_len - 8;                     // the FDAT chunk's length, big-endian
        lenf[0] = (uint8_t)(plen >> 24); lenf[1] = (uint8_t)(plen >> 16); lenf[2] = (uint8_t)(plen >> 8); lenf[3] = (uint8_t)plen;
        for (uint32_t sg = s0; sg < s1; sg++) segdst[sg] = pos + d.prefix_len + (seg_off[sg] - base);
        ent_off[i] = pos;
    }
    if (i == 0) { const uint64_t end = out_base + wg_off[nwg]; segdst[nseg] = end; ent_off[nentry] = end; *total = end - out_base; }
}
void k_scan_launch(const uint64_t *in, uint64_t *out, uint32_t n, hipStream_t st);   // k_entropy.hip
// scratch: 2 x (ceil(nentry / 1024) + 1) u64 behind the entry offsets (the caller's ent_off has nentry + 2 + that many slots)
void launch_layout(FrameDesc *fd, uint8_t *blob, const uint32_t *entry_seg, const uint64_t *seg_off, uint32_t nentry, uint32_t nseg, uint64_t out_base,
                   uint64_t *segdst, uint64_t *ent_off, uint64_t *total, hipStream_t st) {
    const uint32_t nwg = (nentry + 1023) / 1024;
    uint64_t *wg_sum = ent_off + nentry + 2, *wg_off = wg_sum + nwg + 1;
    hipLaunchKernelGGL(k_layout_a, dim3(nwg), dim3(1024), 0, st, fd, entry_seg, seg_off, nentry, ent_off, wg_sum);
    k_scan_launch(wg_sum, wg_off, nwg, st);
    hipLaunchKernelGGL(k_layout_c, dim3(nwg), dim3(1024), 0, st, fd, blob, entry_seg, seg_off, nentry, nseg, out_base, wg_off, nwg, segdst, ent_off, total);
}

// ------------------------------------------------------------------ k_link_copy : a sub-batch's archive bytes from HBM into page-locked host memory
// A plain 16-byte copy on FEW workgroups: the number of workgroups sets the rate (4: 26 GB/s, 8: 36 GB/s of stores over the link), and that is the
// point -- next to it the H2D copy engine keeps its 57 GB/s, where the runtime's D2H copy (a blit kernel at full tilt, 51 GB/s) took 30 % off it
// (experiments/link_duplex.hip).  src, dst: 16-byte aligned.
__global__ __launch_bounds__(256)
void k_link_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, const uint8_t *__restrict__ src8, uint8_t *__restrict__ dst8, uint32_t tail) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < tail) dst8[(n16 << 4) + threadIdx.x] = src8[(n16 << 4) + threadIdx.x];
}
void launch_link_copy(const uint8_t *src, uint8_t *dst, size_t n, uint32_t wgs, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_link_copy, dim3(wgs ? wgs : 1), dim3(256), 0, st, (const uint4 *)src, (uint4 *)dst, n >> 4, src, dst, (uint32_t)(n & 15));
}

// ------------------------------------------------------------------ k_link_gather : many small page-locked host buffers -> one device buffer
// The streaming facade's writers fill 1 MiB page-locked slabs; as one hipMemcpyAsync each they took ~30 us apiece on the copy engine (33 GB/s for a
// batch of 256: the copy-in stage bound the facade's pipeline).  One kernel reads them through their device mapping instead -- workgroup w takes the
// segments w, w + W, ..., 16 bytes per lane and step --; the list of segments lies in page-locked memory as well.  len: multiples of 16 but for a
// stream's last slab (its tail bytes go one by one).
struct LinkSeg { const uint8_t *src; uint8_t *dst; uint64_t len; };
__global__ __launch_bounds__(256)
void k_link_gather(const LinkSeg *__restrict__ segs, uint32_t nseg) {
    for (uint32_t s = blockIdx.x; s < nseg; s += gridDim.x) {
        const LinkSeg g = segs[s];
        const uint4 *src = (const uint4 *)g.src; uint4 *dst = (uint4 *)g.dst;
        const size_t n16 = g.len >> 4;
        size_t i = threadIdx.x;
        for (; i + 768 < n16; i += 1024) {                                   // four loads in flight per lane: the link's latency is what a lane waits for
            const uint4 a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
            dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d;
        }
        for (; i < n16; i += 256) dst[i] = src[i];
        if (threadIdx.x < (g.len & 15)) g.dst[(n16 << 4) + threadIdx.x] = g.src[(n16 << 4) + threadIdx.x];
    }
}
void launch_link_gather(const void *segs, uint32_t nseg, uint32_t wgs, hipStream_t st) {
    if (nseg) hipLaunchKernelGGL(k_link_gather, dim3(wgs < nseg ? wgs : nseg), dim3(256), 0, st, (const LinkSeg *)segs, nseg);
}

void launch_gather(const void *pd, uint32_t n, const uint8_t *src, uint8_t *dst, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_gather, dim3(n), dim3(256), 0, st, (const PlaceDesc *)pd, src, dst);
}

} // namespace pna
