/*
 * oracle/cpu_baseline.c -- TEST / BENCH INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg).
 *
 * CPU restatement of the reference's create pipeline for timing purposes:
 *   cli/src/command/core.rs:496-537  rayon scope_fifo, one entry per task on all logical CPUs
 *   lib/src/entry/write.rs:260-262   zstd::stream::write::Encoder::new(w, 3): streaming libzstd level 3,
 *                                    no pledged size, no checksum (frames start 28 B5 2F FD 00 58)
 * libzstd is the reference's own codec (Cargo.lock:3547-3572 pins 1.5.7; the host's libzstd.so.1 is used here
 * and its version is reported).  It is loaded with dlopen so the oracle has no link-time dependency; when it is
 * absent the caller falls back to timing the model encoder and says so.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { const void *src; size_t size; size_t pos; } zin;
typedef struct { void *dst; size_t size; size_t pos; } zout;
typedef void *(*fn_create)(void);
typedef size_t (*fn_free)(void *);
typedef size_t (*fn_setp)(void *, int, int);
typedef size_t (*fn_stream2)(void *, zout *, zin *, int);
typedef unsigned (*fn_iserr)(size_t);
typedef unsigned (*fn_ver)(void);
typedef size_t (*fn_reset)(void *, int);

static struct { void *h; fn_create create; fn_free freec; fn_setp setp; fn_stream2 stream2; fn_iserr iserr; fn_ver ver; fn_reset reset; } Z;

static int load_zstd(void) {
    if (Z.h) return 0;
    const char *names[] = {"libzstd.so.1", "/usr/lib/x86_64-linux-gnu/libzstd.so.1", "/opt/conda/lib/libzstd.so.1", 0};
    for (int i = 0; names[i] && !Z.h; i++) Z.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!Z.h) return -1;
    Z.create = (fn_create)dlsym(Z.h, "ZSTD_createCCtx"); Z.freec = (fn_free)dlsym(Z.h, "ZSTD_freeCCtx");
    Z.setp = (fn_setp)dlsym(Z.h, "ZSTD_CCtx_setParameter"); Z.stream2 = (fn_stream2)dlsym(Z.h, "ZSTD_compressStream2");
    Z.iserr = (fn_iserr)dlsym(Z.h, "ZSTD_isError"); Z.ver = (fn_ver)dlsym(Z.h, "ZSTD_versionNumber");
    Z.reset = (fn_reset)dlsym(Z.h, "ZSTD_CCtx_reset");
    if (!Z.create || !Z.freec || !Z.setp || !Z.stream2 || !Z.iserr || !Z.ver || !Z.reset) { dlclose(Z.h); Z.h = 0; return -1; }
    return 0;
}

unsigned pna_cpu_zstd_version(void) { return load_zstd() ? 0 : Z.ver(); }

typedef struct {
    const uint8_t *data; size_t n_files, file_len, stride; int level;
    volatile long *next; uint64_t out_bytes; int err;
} job;

static void *worker(void *arg) {
    job *j = (job *)arg;
    void *cctx = Z.create();
    size_t cap = j->file_len + (j->file_len >> 7) + 1024;
    uint8_t *buf = (uint8_t *)malloc(cap);
    for (;;) {
        long i = __sync_fetch_and_add(j->next, 1);
        if ((size_t)i >= j->n_files) break;
        /* a fresh encoder per entry, as the reference builds one per FileEntryBuilder */
        Z.reset(cctx, 1 /* ZSTD_reset_session_only */);
        Z.setp(cctx, 100 /* ZSTD_c_compressionLevel */, j->level);
        zin in = {j->data + (size_t)i * j->stride, j->file_len, 0};
        zout out = {buf, cap, 0};
        size_t r = Z.stream2(cctx, &out, &in, 0 /* ZSTD_e_continue */);
        if (Z.iserr(r)) { j->err = 1; break; }
        do { r = Z.stream2(cctx, &out, &in, 2 /* ZSTD_e_end */); if (Z.iserr(r)) { j->err = 1; break; } } while (r != 0);
        j->out_bytes += out.pos;
    }
    free(buf); Z.freec(cctx);
    return 0;
}

/* Compress n_files files of file_len bytes (file i at data + i*stride) with `threads` workers.
 * Returns seconds (wall) or a negative value on failure; *out_total receives the compressed bytes. */
double pna_cpu_baseline_zstd(const uint8_t *data, size_t n_files, size_t file_len, size_t stride, int threads, int level,
                             uint64_t *out_total) {
    if (load_zstd()) return -1.0;
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_t th[1024]; job jobs[1024];
    volatile long next = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        job jj = {data, n_files, file_len, stride, level, &next, 0, 0}; jobs[t] = jj;
        pthread_create(&th[t], 0, worker, &jobs[t]);
    }
    uint64_t total = 0; int err = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], 0); total += jobs[t].out_bytes; err |= jobs[t].err; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (out_total) *out_total = total;
    if (err) return -2.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- Compression::Deflate: flate2's ZlibEncoder::new(w, level 6) (lib/src/entry/write.rs:257-259) == a streaming zlib deflate per entry.
 * libz is loaded with dlopen like libzstd; the z_stream layout below is zlib's stable ABI (zlib.h, 1.2.x). */
typedef struct {
    const unsigned char *next_in; unsigned avail_in; unsigned long total_in;
    unsigned char *next_out; unsigned avail_out; unsigned long total_out;
    const char *msg; void *state; void *zalloc; void *zfree; void *opaque; int data_type; unsigned long adler; unsigned long reserved;
} zstrm;
typedef int (*fn_dinit)(zstrm *, int, const char *, int);
typedef int (*fn_deflate)(zstrm *, int);
typedef int (*fn_dend)(zstrm *);
typedef const char *(*fn_zver)(void);
static struct { void *h; fn_dinit init; fn_deflate deflate; fn_dend end; fn_zver ver; } ZL;

static int load_zlib(void) {
    if (ZL.h) return 0;
    const char *names[] = {"libz.so.1", "/usr/lib/x86_64-linux-gnu/libz.so.1", "/lib/x86_64-linux-gnu/libz.so.1", 0};
    for (int i = 0; names[i] && !ZL.h; i++) ZL.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!ZL.h) return -1;
    ZL.init = (fn_dinit)dlsym(ZL.h, "deflateInit_"); ZL.deflate = (fn_deflate)dlsym(ZL.h, "deflate");
    ZL.end = (fn_dend)dlsym(ZL.h, "deflateEnd"); ZL.ver = (fn_zver)dlsym(ZL.h, "zlibVersion");
    if (!ZL.init || !ZL.deflate || !ZL.end || !ZL.ver) { dlclose(ZL.h); ZL.h = 0; return -1; }
    return 0;
}
const char *pna_cpu_zlib_version(void) { return load_zlib() ? "" : ZL.ver(); }

static void *worker_deflate(void *arg) {
    job *j = (job *)arg;
    size_t cap = j->file_len + (j->file_len >> 9) + 1024;
    uint8_t *buf = (uint8_t *)malloc(cap);
    for (;;) {
        long i = __sync_fetch_and_add(j->next, 1);
        if ((size_t)i >= j->n_files) break;
        zstrm s; memset(&s, 0, sizeof s);                       /* a fresh encoder per entry */
        if (ZL.init(&s, j->level, ZL.ver(), (int)sizeof s) != 0) { j->err = 1; break; }
        s.next_in = j->data + (size_t)i * j->stride; s.avail_in = (unsigned)j->file_len;
        s.next_out = buf; s.avail_out = (unsigned)cap;
        int r = ZL.deflate(&s, 4 /* Z_FINISH */);
        if (r != 1 /* Z_STREAM_END */) { ZL.end(&s); j->err = 1; break; }
        j->out_bytes += s.total_out;
        ZL.end(&s);
    }
    free(buf);
    return 0;
}

double pna_cpu_baseline_deflate(const uint8_t *data, size_t n_files, size_t file_len, size_t stride, int threads, int level,
                                uint64_t *out_total) {
    if (load_zlib()) return -1.0;
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_t th[1024]; job jobs[1024];
    volatile long next = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        job jj = {data, n_files, file_len, stride, level, &next, 0, 0}; jobs[t] = jj;
        pthread_create(&th[t], 0, worker_deflate, &jobs[t]);
    }
    uint64_t total = 0; int err = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], 0); total += jobs[t].out_bytes; err |= jobs[t].err; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (out_total) *out_total = total;
    if (err) return -2.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- `pna create --solid`: ONE streaming encoder over the serialised inner entries, on ONE thread (lib/src/archive/write.rs:459-463:
 * into_solid_archive wraps the sink in one compression_writer; cli/src/command/create.rs:594-623 feeds it entry by entry).  The inner
 * records are `n_files` stored entries of `file_len` bytes; `rec_overhead` bytes of chunk framing per record are fed as zeros-free
 * filler (taken from the data itself) so that the byte count matches.  algo 2 = zstd, 1 = zlib. */
double pna_cpu_baseline_solid(const uint8_t *data, size_t n_files, size_t file_len, size_t stride, size_t rec_overhead, int algo, int level,
                              uint64_t *out_total) {
    struct timespec t0, t1;
    uint64_t total = 0;
    size_t cap = 1u << 20;
    uint8_t *buf = (uint8_t *)malloc(cap);
    if (algo == 2) {
        if (load_zstd()) { free(buf); return -1.0; }
        void *cctx = Z.create();
        Z.setp(cctx, 100, level);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (size_t i = 0; i < n_files; i++) {
            for (int part = 0; part < 2; part++) {
                zin in = {part ? data + i * stride : data, part ? file_len : rec_overhead, 0};
                while (in.pos < in.size) {
                    zout out = {buf, cap, 0};
                    size_t r = Z.stream2(cctx, &out, &in, 0);
                    if (Z.iserr(r)) { Z.freec(cctx); free(buf); return -2.0; }
                    total += out.pos;
                }
            }
        }
        for (;;) { zin in = {0, 0, 0}; zout out = {buf, cap, 0}; size_t r = Z.stream2(cctx, &out, &in, 2); total += out.pos; if (Z.iserr(r)) { Z.freec(cctx); free(buf); return -2.0; } if (r == 0) break; }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        Z.freec(cctx);
    } else {
        if (load_zlib()) { free(buf); return -1.0; }
        zstrm s; memset(&s, 0, sizeof s);
        if (ZL.init(&s, level, ZL.ver(), (int)sizeof s) != 0) { free(buf); return -2.0; }
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (size_t i = 0; i < n_files; i++) {
            for (int part = 0; part < 2; part++) {
                s.next_in = part ? data + i * stride : data; s.avail_in = (unsigned)(part ? file_len : rec_overhead);
                while (s.avail_in) { s.next_out = buf; s.avail_out = (unsigned)cap; if (ZL.deflate(&s, 0) != 0) { ZL.end(&s); free(buf); return -2.0; } }
            }
        }
        for (;;) { s.next_out = buf; s.avail_out = (unsigned)cap; int r = ZL.deflate(&s, 4); if (r == 1) break; if (r != 0) { ZL.end(&s); free(buf); return -2.0; } }
        total = s.total_out;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        ZL.end(&s);
    }
    free(buf);
    if (out_total) *out_total = total;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

#include <immintrin.h>
/* CRC-32 (IEEE, reflected) by carry-less multiplication: the folding scheme of Gopal et al. (Intel, "Fast CRC Computation for Generic Polynomials Using
 * PCLMULQDQ"), constants for the reflected polynomial 0xEDB88320.  len >= 64 and a multiple of 16; the caller handles the rest with a table. */
__attribute__((target("pclmul,sse4.1")))
static uint32_t crc32_pclmul(const uint8_t *buf, size_t len, uint32_t crc) {
    static const uint64_t __attribute__((aligned(16))) k1k2[] = {0x0154442bd4, 0x01c6e41596};
    static const uint64_t __attribute__((aligned(16))) k3k4[] = {0x01751997d0, 0x00ccaa009e};
    static const uint64_t __attribute__((aligned(16))) k5k0[] = {0x0163cd6124, 0x0000000000};
    static const uint64_t __attribute__((aligned(16))) poly[] = {0x01db710641, 0x01f7011641};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, x3); x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);
    x2 = _mm_and_si128(x1, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x10); x2 = _mm_and_si128(x2, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

/* The reference's serial tail behind the parallel phase (cli/src/command/core.rs:471-493 drain_entry_results -> Archive::add_entry ->
 * lib/src/chunk/write.rs: CRC-32 over every chunk, then the bytes go to the writer): ONE thread, once all entries are compressed (the rayon
 * scope at core.rs:505-537 ends before the drain starts).  Timed here as a carry-less-multiplication CRC-32 (what the reference's crc32fast runs on
 * x86-64 with PCLMULQDQ; zlib's table crc32 where the CPU has none) plus one copy of `bytes` compressed bytes into a sink buffer, in chunks of `chunk`
 * bytes.  Returns seconds, < 0 on failure; *simd = 1 when the carry-less form ran. */
typedef unsigned long (*fn_crc32)(unsigned long, const unsigned char *, unsigned);
double pna_cpu_baseline_tail(size_t bytes, size_t chunk, int *simd) {
    if (load_zlib()) return -1.0;
    fn_crc32 crc = (fn_crc32)dlsym(ZL.h, "crc32");
    if (!crc || !bytes || !chunk) return -1.0;
    const int fast = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (simd) *simd = fast;
    const size_t win = bytes < ((size_t)256 << 20) ? bytes : ((size_t)256 << 20);    /* a 256 MiB window, walked as often as needed */
    unsigned char *src = (unsigned char *)malloc(win), *dst = (unsigned char *)malloc(win);
    if (!src || !dst) { free(src); free(dst); return -1.0; }
    uint32_t x = 12345u;
    for (size_t i = 0; i < win; i++) { x = x * 1664525u + 1013904223u; src[i] = (unsigned char)(x >> 24); }
    memset(dst, 0, win);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    unsigned long acc = 0;
    for (size_t done = 0; done < bytes; ) {
        const size_t off = done % win;
        size_t n = chunk < bytes - done ? chunk : bytes - done;
        if (n > win - off) n = win - off;
        if (fast && n >= 64) {
            const size_t body = n & ~(size_t)15;
            uint32_t c = crc32_pclmul(src + off, body, 0xFFFFFFFFu);
            acc ^= crc(~c & 0xFFFFFFFFu, src + off + body, (unsigned)(n - body));     /* (the last < 16 bytes by the table) */
        } else acc ^= crc(0, src + off, (unsigned)n);
        memcpy(dst + off, src + off, n);
        done += n;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    volatile unsigned long sinkv = acc ^ dst[win / 2]; (void)sinkv;
    free(src); free(dst);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
