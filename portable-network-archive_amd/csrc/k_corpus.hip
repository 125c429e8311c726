// k_corpus.hip -- benchmark support: synthetic corpus generator on the GPU (one thread per 4 KiB piece).
// Not part of the reference's surface (the reference ships no corpus generator; its benches read
// resources/test/raw, cli/benches/create.rs:24-60).  Byte-identical to oracle/corpus_model.c, which tests verify.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pna {

constexpr int VOCAB = 50000, WORD_SLOT = 16, PIECE = 4096, NPHRASE = 8192;
constexpr uint32_t PHRASE_P = 20000;

__device__ __forceinline__ uint64_t sm64(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ int cum_search(const uint64_t *cum, int n, uint64_t r) {
    uint64_t x = __umul64hi(r, cum[n - 1]);
    int lo = 0, hi = n - 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cum[mid] > x) hi = mid; else lo = mid + 1; }
    return lo;
}

struct PieceOut {
    uint8_t *p; int pos;
    __device__ __forceinline__ void put(uint8_t c) { if (pos < PIECE) p[pos++] = c; }
};

__global__ void k_corpus(int kind, uint64_t first_file, uint64_t n_files, uint64_t file_len, uint64_t stride,
                         const uint8_t *__restrict__ vocab, const uint64_t *__restrict__ cum,
                         const uint32_t *__restrict__ phrases, uint8_t *__restrict__ dst) {
    const uint64_t ppf = (file_len + PIECE - 1) / PIECE;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= ppf * n_files) return;
    const uint64_t f = gid / ppf, pi = gid % ppf, file_idx = first_file + f;
    uint8_t buf[PIECE];                       // private scratch: pieces may be cut short at the file end
    PieceOut o{buf, 0};
    uint64_t s = 0x504E410000000000ull ^ ((uint64_t)kind << 56) ^ (file_idx * 0x9E3779B97F4A7C15ull) ^ (pi * 0xD1B54A32D192ED03ull);
    if (kind == 3) { for (int i = 0; i < PIECE; i++) buf[i] = 0; }
    else if (kind == 4) { for (int i = 0; i < PIECE; i++) buf[i] = 'x'; }
    else if (kind == 2) { for (int i = 0; i < PIECE; i += 8) { uint64_t r = sm64(s); for (int k = 0; k < 8; k++) buf[i + k] = (uint8_t)(r >> (8 * k)); } }
    else if (kind == 1) {
        int line = 0, line_max = 64 + (int)(sm64(s) & 31);
        while (o.pos < PIECE) {
            uint64_t r = sm64(s);
            const uint8_t *slot = vocab + (size_t)(r & 4095) * WORD_SLOT;
            int len = slot[0];
            for (int i = 0; i < len; i++) o.put(slot[1 + i]);
            line += len + 1;
            if (o.pos < PIECE) {
                if (line >= line_max) { o.put('\n'); line = 0; line_max = 64 + (int)((r >> 40) & 31); }
                else o.put(' ');
            }
        }
    } else {
        int sent_left = 0, sent_count = 0, first = 1, ph_left = 0;
        const uint32_t *ph = phrases;
        while (o.pos < PIECE) {
            uint64_t r = sm64(s);
            if (sent_left == 0) { sent_left = 5 + (int)((r >> 48) % 21); first = 1; }
            uint64_t r2 = sm64(s);
            int w;
            if (ph_left > 0) { w = (int)ph[1 + (int)ph[0] - ph_left]; ph_left--; }
            else if (((r >> 24) & 0xFFFF) < PHRASE_P) { int k = cum_search(cum, NPHRASE, r2); ph = phrases + 4 * k; w = (int)ph[1]; ph_left = (int)ph[0] - 1; }
            else w = cum_search(cum, VOCAB, r2);
            const uint8_t *slot = vocab + (size_t)w * WORD_SLOT;
            int len = slot[0];
            uint32_t mk = (uint32_t)(r & 0xFFFF);
            int markup = mk < 1311 ? 1 + (int)(mk % 3) : 0;
            if (markup == 1) { o.put('['); o.put('['); }
            if (markup == 2) { const char t[] = "<title>"; for (int i = 0; i < 7; i++) o.put((uint8_t)t[i]); }
            for (int i = 0; i < len; i++) { uint8_t c = slot[1 + i]; if (i == 0 && first) c = (uint8_t)(c - 32); o.put(c); }
            first = 0;
            if (markup == 1) { o.put(']'); o.put(']'); }
            if (markup == 2) { const char t[] = "</title>"; for (int i = 0; i < 8; i++) o.put((uint8_t)t[i]); }
            if (markup == 3) { const char t[] = " &amp;"; for (int i = 0; i < 6; i++) o.put((uint8_t)t[i]); }
            sent_left--;
            if (sent_left == 0) {
                o.put('.'); sent_count++;
                if (sent_count % 40 == 0) o.put('\n'); else o.put(' ');
            } else {
                uint32_t pc = (uint32_t)((r >> 16) & 0xFF);
                if (pc < 20) o.put(',');
                o.put(' ');
            }
        }
    }
    const uint64_t off = pi * PIECE;
    const uint64_t n = file_len - off < PIECE ? file_len - off : PIECE;
    uint8_t *out = dst + f * stride + off;
    for (uint64_t i = 0; i < n; i++) out[i] = buf[i];
}

void launch_corpus(int kind, uint64_t first_file, uint64_t n_files, uint64_t file_len, uint64_t stride,
                   const uint8_t *vocab, const uint64_t *cum, const uint32_t *phrases, uint8_t *dst, hipStream_t st) {
    const uint64_t ppf = (file_len + PIECE - 1) / PIECE, total = ppf * n_files;
    if (total == 0) return;
    hipLaunchKernelGGL(k_corpus, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st, kind, first_file, n_files, file_len, stride, vocab, cum, phrases, dst);
}

} // namespace pna
