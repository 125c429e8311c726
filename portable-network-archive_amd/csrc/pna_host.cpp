// pna_host.cpp -- context, options, level sets and the ENCODE core of libpna_gpu.so (plans, the LZ / entropy / framing launches of one sub-batch, the
// device-resident entry points); the host pipelines, the read side and the streaming facade live in pna_pipeline.cpp, pna_extract.cpp, pna_decode.cpp and
// pna_stream.cpp, what they share in pna_ctx.h.
#include "pna_ctx.h"
static const TuningName TUNING_NAMES[] = {
    {"lz_split", "PNA_LZ_SPLIT", &Tuning::lz_split, 0, 2}, {"lz_split_blocks", "PNA_LZ_SPLIT_BLOCKS", &Tuning::lz_split_blocks, 8, 1 << 17},
    {"lz_split_min", "PNA_LZ_SPLIT_MIN", &Tuning::lz_split_min, 0, 1 << 30}, {"lz_pbuf_fail", "PNA_LZ_PBUF_FAIL", &Tuning::lz_pbuf_fail, 0, 1},
    {"pipeline_chunks", "PNA_PIPELINE_CHUNKS", &Tuning::pipeline_chunks, 1, 8}, {"max_chunk_size", "PNA_MAX_CHUNK_SIZE", &Tuning::max_chunk_size, 0, 0xFFFFFFFFl},
    {"sub_mib", "PNA_SUB_MIB", &Tuning::sub_mib, 16, 16384}, {"stage_threads", "PNA_STAGE_THREADS", &Tuning::stage_threads, 0, 64},
    {"extract_win_mib", "PNA_EXTRACT_WIN_MIB", &Tuning::extract_win_mib, 1, 1 << 20}, {"batch_piece_mib", "PNA_BATCH_PIECE_MIB", &Tuning::batch_piece_mib, 0, 1 << 20}, {"solid_win_mib", "PNA_SOLID_WIN_MIB", &Tuning::solid_win_mib, 1, 1 << 16},
    {"inflate_serial", "PNA_INFLATE_SERIAL", &Tuning::inflate_serial, 0, 1}, {"zdec_serial", "PNA_ZDEC_SERIAL", &Tuning::zdec_serial, 0, 1}, {"zdec_dbg", "PNA_ZDEC_DBG", &Tuning::zdec_dbg, 0, 15},
    {"blk_log", "PNA_BLK_LOG", &Tuning::blk_log, 0, PNA_BLK_LOG}, {"unit_log", "PNA_LZ_UNIT_LOG", &Tuning::unit_log, 0, 20},
    {"latency_max_mib", "PNA_LATENCY_MAX_MIB", &Tuning::latency_max_mib, 0, 1 << 20}, {"hist_by_block", "PNA_HIST_BY_BLOCK", &Tuning::hist_by_block, -1, 1},
    {"d2h_wgs", "PNA_D2H_WGS", &Tuning::d2h_wgs, 0, 4096}, {"trace", "PNA_TRACE", &Tuning::trace, 0, 1}, {"dev_layout", "PNA_DEV_LAYOUT", &Tuning::dev_layout, 0, 1}, {"strong_gtab", "PNA_STRONG_GTAB", &Tuning::strong_gtab, 0, 1}, {"win32k", "PNA_WIN32K", &Tuning::win32k, 0, 2}, {"tab3", "PNA_TAB3", &Tuning::tab3, 0, 1}, {"far1", "PNA_FAR1", &Tuning::far1, 0, 1}, {"seq_hist", "PNA_SEQ_HIST", &Tuning::seq_hist, 0, 1}, {"strong2", "PNA_STRONG2", &Tuning::strong2, 0, 1}, {"small_geometry", "PNA_SMALL_GEOMETRY", &Tuning::small_geometry, 0, 1}, {"zexec_par_min_mib", "PNA_ZEXEC_PAR_MIN_MIB", &Tuning::zexec_par_min_mib, 0, 1 << 20}, {"zexec_win_mib", "PNA_ZEXEC_WIN_MIB", &Tuning::zexec_win_mib, 1, 1024}, {"zdec_fallback_max_mib", "PNA_ZDEC_FALLBACK_MAX_MIB", &Tuning::zdec_fallback_max_mib, 0, 1 << 30}, {"stream_batch_mib", "PNA_STREAM_BATCH_MIB", &Tuning::stream_batch_mib, 1, 1 << 16}, {"stream_gather_wgs", "PNA_STREAM_GATHER_WGS", &Tuning::stream_gather_wgs, 0, 4096}, {"stream_overlap_mib", "PNA_STREAM_OVERLAP_MIB", &Tuning::stream_overlap_mib, 0, 1 << 16}, {"single_frame", "PNA_SINGLE_FRAME", &Tuning::single_frame, 0, 1}, {"lazy2", "PNA_LAZY2", &Tuning::lazy2, 0, 2}, {"tail_units", "PNA_TAIL_UNITS", &Tuning::tail_units, 0, 1}, {"lit_beside_seq", "PNA_LIT_BESIDE_SEQ", &Tuning::lit_beside_seq, 0, 1}, {"sub_ramp_down", "PNA_SUB_RAMP_DOWN", &Tuning::sub_ramp_down, 0, 1},
};

extern "C" const char *pna_gpu_strerror(int code) {
    switch (code) {
        case PNA_OK: return "ok";
        case PNA_E_NODEVICE: return "no usable HIP device";
        case PNA_E_INVAL: return "invalid argument";
        case PNA_E_NOMEM: return "out of memory";
        case PNA_E_DSTSIZE: return "destination too small";
        case PNA_E_HIP: return "HIP error";
        case PNA_E_SINK: return "sink callback failed";
        case PNA_E_UNSUPPORTED: return "algorithm not supported by this build";
        default: return "unknown error";
    }
}
extern "C" const char *pna_gpu_last_error(const pna_gpu_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int pna_gpu_init(pna_gpu_ctx **out, int device_id, uint32_t flags) {
    if (!out) return PNA_E_INVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PNA_E_NODEVICE;
    if (device_id < 0 || device_id >= n) return PNA_E_INVAL;
    if (hipSetDevice(device_id) != hipSuccess) return PNA_E_NODEVICE;
    pna_gpu_ctx *c = new pna_gpu_ctx();
    c->device = device_id;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->n_cus = (uint32_t)cus; }
    c->flags = (flags & PNA_F_DEFAULT) ? (F_HUF | F_FSE | F_LAZY | F_FAR | F_ADOPT | F_INS2) : (flags & 0xFF);
    c->flags &= ~F_REP;                    // repeat-offset codes are not produced by this build
    for (const TuningName &t : TUNING_NAMES)
        if (const char *e = getenv(t.env)) { const long v = atol(e); if (v >= t.lo && v <= t.hi) c->tun.*(t.field) = v; }
    if (const char *pm = getenv("PNA_STREAM_POOL_MIB")) c->pool_cap = (size_t)std::min<unsigned long>(strtoul(pm, nullptr, 10), 1ul << 20) << 20;
    if (const char *lg = getenv("PNA_STREAM_LINGER_US")) c->comb_linger_us = (uint32_t)std::min<unsigned long>(strtoul(lg, nullptr, 10), 100000ul);
    if (!(flags & PNA_F_DEFAULT)) c->flags |= flags & 0x3F00u;  // diagnostics: 0x100 phase stamps, 0x200 force the serial fallback in k_lz, 0x1000 / 0x2000 force the one-kernel / two-phase sequence coder
    c->flags |= flags & (PNA_F_LZ_FUSED | PNA_F_LZ_WAVEPARSE);  // the form of the LZ stage: honoured next to PNA_F_DEFAULT as well
    c->call_flags = c->flags;
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return PNA_E_NODEVICE; }
    for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return PNA_E_NODEVICE; }
    {   // the predefined sequence tables of this device (k_entropy.hip g_def_tab): built by the first context on it
        static std::mutex mu; static bool built[64] = {};
        std::lock_guard<std::mutex> lk(mu);
        if (device_id >= 0 && device_id < 64 && !built[device_id]) {
            launch_default_tables(c->stream);
            if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) { delete c; return PNA_E_HIP; }
            built[device_id] = true;
        }
    }
    *out = c;
    return PNA_OK;
}

extern "C" int pna_gpu_set_option(pna_gpu_ctx *c, const char *name, long value) {
    if (!c || !name) return PNA_E_INVAL;
    if (!strcmp(name, "stream_pool_mib")) { if (value < 0) return PNA_E_INVAL; c->pool_cap = (size_t)std::min<long>(value, 1l << 20) << 20; return PNA_OK; }
    if (!strcmp(name, "stream_linger_us")) { c->comb_linger_us = value < 0 ? 0xFFFFFFFFu : (uint32_t)std::min<long>(value, 100000); return PNA_OK; }
    for (const TuningName &t : TUNING_NAMES)
        if (!strcmp(name, t.name)) {
            if (value < t.lo || value > t.hi) return fail(c, PNA_E_INVAL, "option value out of range");
            if (t.field == &Tuning::blk_log && value != 0 && value < (long)BLK_LOG_MIN) return fail(c, PNA_E_INVAL, "blk_log: 0 or 13..17");
            c->tun.*(t.field) = value; return PNA_OK;
        }
    return fail(c, PNA_E_INVAL, "unknown option");
}

extern "C" void pna_gpu_shutdown(pna_gpu_ctx *c) {
    if (c) { for (void *a : c->pool_arenas) (void)hipHostFree(a); c->pool_arenas.clear(); c->pool_free.clear(); for (auto &b : c->s_out) b.release(); for (auto &b : c->st_in) b.release(); for (auto &b : c->st_out) b.release(); for (auto &b : c->s_segs) b.release();
             if (c->s_h2d) (void)hipStreamDestroy(c->s_h2d); if (c->s_d2h) (void)hipStreamDestroy(c->s_d2h); for (auto &e : c->s_ev) if (e) (void)hipEventDestroy(e); }
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->plan, &c->d_tail, &c->blk, &c->tabs, &c->seqs, &c->lits, &c->litc, &c->seqc, &c->seqw, &c->pbuf, &c->seg_size,
                      &c->seg_off, &c->stage_in, &c->stage_out, &c->z_words, &c->z_rep, &c->z_zxf, &c->z_spec, &c->z_big, &c->z_one, &c->ctab, &c->c_vocab, &c->c_cum, &c->c_phr,
                      &c->gtab, &c->fr_desc, &c->fr_blob, &c->fr_segdst, &c->fr_entoff, &c->crc_tabs, &c->aes_tabs, &c->ci_units, &c->ci_ivs, &c->ci_keys, &c->ci_gcm, &c->ci_spread, &c->ci_spread_desc, &c->z_vp, &c->z_pb, &c->z_mode, &c->x_arc, &c->x_pk, &c->x_raw[0], &c->x_raw[1], &c->x_desc, &c->x_place, &c->x_flag, &c->x_tags, &c->x_plen, &c->aes_dtabs, &c->solid_plain, &c->solid_desc, &c->solid_blob, &c->solid_place, &c->z_ents, &c->z_frames, &c->z_lit, &c->z_fx, &c->z_blocks, &c->z_tabs, &c->z_seqs, &c->z_hlist, &c->z_slist, &c->z_work, &c->z_fb, &c->z_cbase, &c->z_apart}) b->release();
    for (auto &b : c->lent) (void)hipHostFree((void *)b.first);          // (buffers the host never gave back)
    c->lent.clear();
    for (PinBuf *b : {&c->h_entoff, &c->h_plan, &c->h_tail, &c->h_desc, &c->h_blob, &c->h_segdst, &c->h_segoff, &c->hp_in[0], &c->hp_in[1], &c->hp_in[2], &c->hp_in[3], &c->hp_out[0], &c->hp_out[1]}) b->release();
    for (DevBuf *b : {&c->dp_in[0], &c->dp_in[1], &c->dp_in[2], &c->dp_in[3], &c->dp_out[0], &c->dp_out[1]}) b->release();
    for (auto &e : c->ev_in) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_out) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_lz) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->lzm_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_ci) if (e) (void)hipEventDestroy(e);
    for (auto &r : c->ev_en) for (auto &e : r) if (e) (void)hipEventDestroy(e);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->cp_in) (void)hipStreamDestroy(c->cp_in);
    if (c->cp_out) (void)hipStreamDestroy(c->cp_out);
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" size_t pna_gpu_bound(int algo, size_t n) {
    if (algo == PNA_ALGO_STORE) return n;
    // (per block of the SMALLEST size a batch may be cut into -- latency mode, BLK_LOG_MIN --: deflate 5 bytes of stored-block header + 5 of sync flush,
    // zstd 3 bytes of block header; per segment / 128 KiB the older, coarser allowances)
    constexpr size_t BMIN = (size_t)1 << BLK_LOG_MIN;
    if (algo == PNA_ALGO_DEFLATE) { size_t b = (n + BMIN - 1) / BMIN; if (!b) b = 1; return n + b * 10 + (n >> PNA_BLK_LOG) * 16 + 64; }
    size_t segs = (n + SEG_SIZE - 1) / SEG_SIZE; if (segs == 0) segs = 1;
    size_t blks = (n + BMIN - 1) / BMIN + segs;
    return n + segs * 6 + blks * 3 + 16;
}

extern "C" int pna_gpu_clamp_level(int algo, int level) {
    if (algo == PNA_ALGO_ZSTD) {            // lib/src/compress/zstandard.rs:13,43-57 (min_c_level .. 22, default 3)
        if (level == PNA_LEVEL_DEFAULT) return 3;
        if (level < -131072) return -131072;
        return level > 22 ? 22 : level;
    }
    if (algo == PNA_ALGO_DEFLATE) {         // lib/src/compress/deflate.rs:33-38,89-101
        if (level == PNA_LEVEL_DEFAULT) return 6;
        // Custom(n) => Compression::new((n as u32).clamp(0, 9)): a negative n wraps to a large u32 and clamps to 9 (deflate.rs:89-101)
        return (uint32_t)level > 9u ? 9 : level;
    }
    return 0;
}

// Parameter sets behind the reference's level scale (CompressionLevel -> ZstdCompressionLevel / flate2::Compression,
// lib/src/compress/zstandard.rs:43-57, deflate.rs:89-101; PNA_LEVEL_DEFAULT and zstd level 0 = the default):
//   stored    deflate 0                      Compression::none(): stored blocks only (no match finder, header 78 01)
//   fast      zstd < 0 and 1, deflate 1..3   every position in the table, look-back = the LDS window, no backward adoption; lazy deferral as below
//   light     zstd 2,         deflate 4..8   + even-position table, backward adoption (two rounds), 1 MiB look-back (zstd), lazy deferral over three positions
//   default   zstd 0, 3                      + a third adoption round (matches move back by up to 7 positions): ratio at the reference's default level (round 4)
//   high      zstd 4..9,      deflate 9      the default set on the 16 KiB-window geometry (more table slots, more candidates verified in HBM / L2) + a fourth adoption round over eight positions, 15 back bytes; deflate: + third round
//   max       zstd 10..22                    + the match kernel's hash table in global memory: 2^19 slots per segment instead of what LDS holds
// zstd light / default run the match finder's 32 KiB-window geometry, high the 16 KiB one (lz_common.h LzGeo), both with the PACKED table (three 21-bit
// entries per 64-bit LDS word: 49 062 / 55 206 slots; option tab3 = 0: 32-bit entries, 32 704 / 36 800); the others and deflate the 64 KiB geometry (24 512).
constexpr int HIGH_FROM = 4;        // the first zstd level of the high set
static uint32_t level_flags(const pna_gpu_ctx *c, int algo, int level) {
    const int lv = pna_gpu_clamp_level(algo, level);
    const bool fast = algo == PNA_ALGO_DEFLATE ? lv <= 3 : (lv < 0 || lv == 1);
    const bool balanced = false;   // (the set without lazy deferral -- zstd 2, deflate 4..5 until round 3 -- is as fast as the default set since the parse kernel looks ahead for free; the levels take the default set, the bits remain)
    const bool strong = algo == PNA_ALGO_DEFLATE ? lv >= 9 : (lv >= 3 || lv == 0);      // (zstd 0 = the default = 3; round 4: the third adoption round from the default level on)
    if (fast) return c->flags & ~(F_FAR | F_ADOPT | F_INS2 | F_STRONG);      // (lazy deferral stays: the parse kernel does it for free -- zstd-1 2.510 -> 2.54 at the same speed)
    if (balanced) return c->flags & ~(F_LAZY | F_STRONG);
    if (strong && (c->flags & F_ADOPT) && (c->flags & F_LAZY)) return c->flags | F_STRONG;
    return c->flags;
}
void set_call_level(pna_gpu_ctx *c, int algo, int level) {
    c->call_flags = level_flags(c, algo, level);
    const bool zstd = algo != PNA_ALGO_DEFLATE;
    c->call_gtab = zstd && pna_gpu_clamp_level(algo, level) >= 10 && (c->call_flags & F_STRONG) && (c->call_flags & F_ADOPT) && c->tun.strong_gtab != 0;
    c->call_w32 = zstd && !c->call_gtab && (c->call_flags & F_FAR) && (c->call_flags & F_LAZY) && c->tun.win32k != 0;
    c->call_w16 = c->call_w32 && (c->tun.win32k >= 2 || pna_gpu_clamp_level(algo, level) >= HIGH_FROM);   // the high set's geometry (round 4: chosen by the level, no longer by F_STRONG; round 5: from level 4 on -- libzstd 1.5.7's own 4 / 5 give 2.863 / 2.906 on the corpus, the default set 2.850, the high set 2.884)
    c->call_stored = !zstd && pna_gpu_clamp_level(algo, level) == 0;
    c->call_tab3 = c->call_w32 && (c->call_flags & F_INS2) && (c->call_flags & F_ADOPT) && c->tun.tab3 != 0;
    c->call_strong2 = zstd && pna_gpu_clamp_level(algo, level) >= HIGH_FROM && (c->call_flags & F_STRONG) && (c->call_flags & F_ADOPT) && c->tun.strong2 != 0 && (c->call_gtab || (c->call_tab3 && c->call_w16));
    c->call_lazy2 = (c->call_flags & F_LAZY) && (c->tun.lazy2 != 0 || (c->call_flags & F_STRONG));   // every lazy set defers over two positions (the high sets always did)
    c->call_lazy3 = c->call_lazy2 && c->tun.lazy2 >= 2;                                               // ... and over three (option lazy2 = 2, the default)
}

// blocks an entry of `len` bytes takes in the per-block workspace (sub-batches are cut by block count); a forced block size counts as such
extern "C" int pna_gpu_last_timing(const pna_gpu_ctx *c, pna_gpu_timing *out) {
    if (!c || !out) return PNA_E_INVAL;
    *out = c->timing; out->blk_log = c->last_blk_log; out->lz_units = c->last_units; return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// One sub-batch: entries [e0, e1) -> segments -> kernels; output appended at d_dst + out_base.
// ---- CRC-32 tables of the framing kernel (k_frame.hip explains the algebra)
uint32_t crc_gf2_mulmod(uint32_t a, uint32_t b);
uint32_t crc_gf2_xpow(uint64_t e);
static uint32_t gf2_mulmod(uint32_t a, uint32_t b) {          // a * b mod P, reflected bit order (bit 31 = x^0)
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) { if (a & 0x80000000u) p ^= b; a <<= 1; b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1; }
    return p;
}
static uint32_t gf2_xpow(uint64_t e) {                        // x^e mod P
    uint32_t r = 0x80000000u, base = 0x40000000u;
    while (e) { if (e & 1) r = gf2_mulmod(base, r); base = gf2_mulmod(base, base); e >>= 1; }
    return r;
}
static void build_crc_tabs(CrcTabs &t) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t v = i; for (int k = 0; k < 8; k++) v = (v & 1) ? (v >> 1) ^ 0xEDB88320u : v >> 1; t.T[0][i] = v; }
    for (int k = 1; k < 4; k++) for (uint32_t i = 0; i < 256; i++) t.T[k][i] = (t.T[k - 1][i] >> 8) ^ t.T[0][t.T[k - 1][i] & 0xFF];
    const uint32_t z = gf2_xpow(8ull * 16320);
    for (int j = 0; j < 4; j++) for (uint32_t b = 0; b < 256; b++) t.Z[j][b] = gf2_mulmod(z, b << (8 * j));
    for (int j = 0; j < 8; j++) t.sh[j] = gf2_xpow(8ull * 64 << j);
    for (int m = 1; m <= 4; m++) for (uint32_t k = 0; k < 64; k++) t.pw[m - 1][k] = gf2_xpow(8ull * 64 * m * k);
}
uint32_t crc_gf2_mulmod(uint32_t a, uint32_t b) { return gf2_mulmod(a, b); }
uint32_t crc_gf2_xpow(uint64_t e) { return gf2_xpow(e); }
int ensure_crc(pna_gpu_ctx *c) {
    if (c->crc_ready) return PNA_OK;
    CrcTabs t; build_crc_tabs(t);
    if (c->crc_tabs.ensure(sizeof(t))) return fail(c, PNA_E_NOMEM, "crc tables");
    HIPCHK(c, hipMemcpy(c->crc_tabs.p, &t, sizeof(t), hipMemcpyHostToDevice));
    c->crc_ready = true;
    return PNA_OK;
}

// Host walk through k_frame's CRC schedule with the same tables (lane states, Z_16320 between tiles, fold tree): lets the
// CPU-only test suite check the algebra and the table construction against pna_crc32 without a GPU.  Not a product path.
extern "C" uint32_t pna_gpu_debug_crc_schedule(const void *payload, size_t len) {
    static const CrcTabs t = [] { CrcTabs x; build_crc_tabs(x); return x; }();      // initialised once, thread-safe (C++11 static)
    const uint8_t *pl = (const uint8_t *)payload;
    const uint64_t n = 4 + (uint64_t)len, ntile = (n + 16383) / 16384, pad = ntile * 16384 - n;
    static const uint8_t ty[4] = {0x46 ^ 0xFF, 0x44 ^ 0xFF, 0x41 ^ 0xFF, 0x54 ^ 0xFF};
    uint32_t lane[256] = {0};
    for (uint64_t k = 0; k < ntile; k++)
        for (uint32_t l = 0; l < 256; l++) {
            uint32_t s = lane[l];
            s = t.Z[0][s & 0xFF] ^ t.Z[1][(s >> 8) & 0xFF] ^ t.Z[2][(s >> 16) & 0xFF] ^ t.Z[3][s >> 24];
            for (uint32_t j = 0; j < 16; j++) {
                uint32_t w = 0;
                for (uint32_t b = 0; b < 4; b++) {
                    const uint64_t p = k * 16384 + l * 64 + 4 * j + b;
                    const uint8_t v = p < pad ? 0 : (p < pad + 4 ? ty[p - pad] : pl[p - pad - 4]);
                    w |= (uint32_t)v << (8 * b);
                }
                const uint32_t x = s ^ w;
                s = t.T[3][x & 0xFF] ^ t.T[2][(x >> 8) & 0xFF] ^ t.T[1][(x >> 16) & 0xFF] ^ t.T[0][x >> 24];
            }
            lane[l] = s;
        }
    for (uint32_t j = 0; j < 8; j++) {
        const uint32_t st = 1u << j;
        for (uint32_t l = 0; l < 256; l += 2 * st) lane[l] = gf2_mulmod(t.sh[j], lane[l]) ^ lane[l + st];
    }
    return ~lane[0];
}

// ---- cipher stage: AES-256 key schedule and round tables (FIPS-197), built on the host once per context / per call
static void aes_sbox(uint8_t sb[256]) {
    uint8_t p = 1, q = 1;                                      // p walks the multiplicative group of GF(2^8), q is its inverse
    do {
        p = (uint8_t)(p ^ (p << 1) ^ ((p & 0x80) ? 0x1B : 0));
        q ^= (uint8_t)(q << 1); q ^= (uint8_t)(q << 2); q ^= (uint8_t)(q << 4); if (q & 0x80) q ^= 0x09;
        const uint8_t r1 = (uint8_t)((q << 1) | (q >> 7)), r2 = (uint8_t)((q << 2) | (q >> 6)), r3 = (uint8_t)((q << 3) | (q >> 5)), r4 = (uint8_t)((q << 4) | (q >> 4));
        sb[p] = (uint8_t)(q ^ r1 ^ r2 ^ r3 ^ r4 ^ 0x63);
    } while (p != 1);
    sb[0] = 0x63;
}
static void build_aes_tabs(AesTabs &t) {
    uint8_t sb[256]; aes_sbox(sb);
    for (uint32_t x = 0; x < 256; x++) {
        const uint32_t s1 = sb[x], s2 = ((s1 << 1) ^ ((s1 & 0x80) ? 0x1B : 0)) & 0xFF, s3 = s2 ^ s1;
        const uint32_t w = s2 | (s1 << 8) | (s1 << 16) | (s3 << 24);
        t.Te[0][x] = w; t.Te[1][x] = (w << 8) | (w >> 24); t.Te[2][x] = (w << 16) | (w >> 16); t.Te[3][x] = (w << 24) | (w >> 8);
    }
}
void aes256_expand(const uint8_t key[32], AesKey &k) {
    uint8_t sb[256]; aes_sbox(sb);
    uint8_t w[60][4]; uint8_t rc = 1;
    memcpy(w, key, 32);
    for (int i = 8; i < 60; i++) {
        uint8_t t[4]; memcpy(t, w[i - 1], 4);
        if (i % 8 == 0) { const uint8_t t0 = t[0]; t[0] = (uint8_t)(sb[t[1]] ^ rc); t[1] = sb[t[2]]; t[2] = sb[t[3]]; t[3] = sb[t0]; rc = (uint8_t)((rc << 1) ^ ((rc & 0x80) ? 0x1B : 0)); }
        else if (i % 8 == 4) for (int j = 0; j < 4; j++) t[j] = sb[t[j]];
        for (int j = 0; j < 4; j++) w[i][j] = (uint8_t)(w[i - 8][j] ^ t[j]);
    }
    for (int i = 0; i < 60; i++) k.rk[i] = (uint32_t)w[i][0] | ((uint32_t)w[i][1] << 8) | ((uint32_t)w[i][2] << 16) | ((uint32_t)w[i][3] << 24);
}
static uint8_t gf_mul8(uint8_t a, uint8_t b) { uint8_t r = 0; while (b) { if (b & 1) r ^= a; a = (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1B : 0)); b >>= 1; } return r; }
// equivalent inverse cipher (FIPS-197 5.3.5): tables of InvSubBytes + InvMixColumns, round keys reversed with InvMixColumns applied to
// the middle ones
static void build_aes_dec_tabs(AesDecTabs &t) {
    uint8_t sb[256], si[256]; aes_sbox(sb);
    for (int x = 0; x < 256; x++) si[sb[x]] = (uint8_t)x;
    for (uint32_t x = 0; x < 256; x++) {
        const uint8_t v = si[x];
        const uint32_t w = (uint32_t)gf_mul8(v, 14) | ((uint32_t)gf_mul8(v, 9) << 8) | ((uint32_t)gf_mul8(v, 13) << 16) | ((uint32_t)gf_mul8(v, 11) << 24);
        t.Td[0][x] = w; t.Td[1][x] = (w << 8) | (w >> 24); t.Td[2][x] = (w << 16) | (w >> 16); t.Td[3][x] = (w << 24) | (w >> 8);
        t.Sd[x] = v;
    }
}
void aes256_dec_key(const AesKey &k, AesKey &d) {
    for (int c4 = 0; c4 < 4; c4++) { d.rk[c4] = k.rk[56 + c4]; d.rk[56 + c4] = k.rk[c4]; }
    for (int r = 1; r < 14; r++)
        for (int c4 = 0; c4 < 4; c4++) {
            const uint32_t w = k.rk[4 * (14 - r) + c4];
            const uint8_t a0 = (uint8_t)w, a1 = (uint8_t)(w >> 8), a2 = (uint8_t)(w >> 16), a3 = (uint8_t)(w >> 24);
            const uint8_t r0 = gf_mul8(a0, 14) ^ gf_mul8(a1, 11) ^ gf_mul8(a2, 13) ^ gf_mul8(a3, 9), r1 = gf_mul8(a0, 9) ^ gf_mul8(a1, 14) ^ gf_mul8(a2, 11) ^ gf_mul8(a3, 13);
            const uint8_t r2 = gf_mul8(a0, 13) ^ gf_mul8(a1, 9) ^ gf_mul8(a2, 14) ^ gf_mul8(a3, 11), r3 = gf_mul8(a0, 11) ^ gf_mul8(a1, 13) ^ gf_mul8(a2, 9) ^ gf_mul8(a3, 14);
            d.rk[4 * r + c4] = (uint32_t)r0 | ((uint32_t)r1 << 8) | ((uint32_t)r2 << 16) | ((uint32_t)r3 << 24);
        }
}
int ensure_aes_dec(pna_gpu_ctx *c) {
    if (c->aes_dec_ready) return PNA_OK;
    static AesDecTabs t; build_aes_dec_tabs(t);
    if (c->aes_dtabs.ensure(sizeof(t))) return fail(c, PNA_E_NOMEM, "aes tables");
    HIPCHK(c, hipMemcpy(c->aes_dtabs.p, &t, sizeof(t), hipMemcpyHostToDevice));
    c->aes_dec_ready = true;
    return PNA_OK;
}
int ensure_aes(pna_gpu_ctx *c) {
    if (c->aes_ready) return PNA_OK;
    AesTabs t; build_aes_tabs(t);
    if (c->aes_tabs.ensure(sizeof(t))) return fail(c, PNA_E_NOMEM, "aes tables");
    HIPCHK(c, hipMemcpy(c->aes_tabs.p, &t, sizeof(t), hipMemcpyHostToDevice));
    for (auto &e : c->ev_ci) HIPCHK(c, hipEventCreate(&e));
    c->aes_ready = true;
    return PNA_OK;
}
int check_cipher(pna_gpu_ctx *c, const pna_gpu_cipher *ci) {
    if (ci->encryption == PNA_ENC_CAMELLIA) return fail(c, PNA_E_UNSUPPORTED, "Camellia is not offered on the device path");
    if (ci->encryption != PNA_ENC_AES) return fail(c, PNA_E_INVAL, "unknown encryption");
    if (ci->cipher_mode != PNA_MODE_CTR && ci->cipher_mode != PNA_MODE_CBC && ci->cipher_mode != PNA_MODE_GCM) return fail(c, PNA_E_UNSUPPORTED, "cipher mode not offered on the device path");
    if (ci->cipher_mode == PNA_MODE_GCM && ci->gcm_segment_size > (64u << 20)) return fail(c, PNA_E_INVAL, "GCM segment size beyond 64 MiB");
    return PNA_OK;
}
// one AES-256 block on the host (FIPS-197 with the round tables of the kernels): hash subkey and E(K, J0) of a GCM segment
void aes256_block_host(const AesKey &k, const uint8_t in[16], uint8_t out[16]) {
    static const AesTabs T = [] { AesTabs x; build_aes_tabs(x); return x; }();      // called from several host threads: C++11 static initialisation is thread-safe
    uint32_t s[4], t[4];
    for (int i = 0; i < 4; i++) s[i] = ((uint32_t)in[4 * i] | ((uint32_t)in[4 * i + 1] << 8) | ((uint32_t)in[4 * i + 2] << 16) | ((uint32_t)in[4 * i + 3] << 24)) ^ k.rk[i];
    for (int r = 1; r < 14; r++) {
        for (int i = 0; i < 4; i++)
            t[i] = T.Te[0][s[i] & 0xFF] ^ T.Te[1][(s[(i + 1) & 3] >> 8) & 0xFF] ^ T.Te[2][(s[(i + 2) & 3] >> 16) & 0xFF] ^ T.Te[3][s[(i + 3) & 3] >> 24] ^ k.rk[4 * r + i];
        memcpy(s, t, sizeof s);
    }
    auto sb = [&](uint32_t v) { return (T.Te[0][v & 0xFF] >> 8) & 0xFF; };
    for (int i = 0; i < 4; i++)
        t[i] = (sb(s[i]) | (sb(s[(i + 1) & 3] >> 8) << 8) | (sb(s[(i + 2) & 3] >> 16) << 16) | (sb(s[(i + 3) & 3] >> 24) << 24)) ^ k.rk[56 + i];
    for (int i = 0; i < 4; i++) { out[4 * i] = (uint8_t)t[i]; out[4 * i + 1] = (uint8_t)(t[i] >> 8); out[4 * i + 2] = (uint8_t)(t[i] >> 16); out[4 * i + 3] = (uint8_t)(t[i] >> 24); }
}
// GCM STREAM material of one entry (lib/src/entry/write.rs:81-107 to_hashed; lib/src/cipher/aead.rs): stream header, stream key bound to
// the FHED chunk and the PHSF string, round keys, hash subkey, E(K, J0) of the (single, final) segment 0 and its first counter block.
// (the stream key is bound to the header chunk of the entry that carries the stream: FHED of a normal entry, SHED of a solid one -- entry_context,
// lib/src/cipher/aead.rs:167-190; name == nullptr: the solid entry's SHED)
void gcm_entry_material(const pna_gpu_cipher *ci, const uint8_t kc[32], const uint8_t phsf_hash[32], const uint8_t salt_prefix[39],
                               uint32_t seg_size, const char *name, int compression, GcmMaterial &m) {
    memcpy(m.header, salt_prefix, 39);
    m.header[39] = (uint8_t)(seg_size >> 24); m.header[40] = (uint8_t)(seg_size >> 16); m.header[41] = (uint8_t)(seg_size >> 8); m.header[42] = (uint8_t)seg_size;
    memcpy(m.header + 43, kc, 32);
    const std::vector<uint8_t> fh = name ? frame_fhed_bytes(name, compression, ci->encryption, PNA_MODE_GCM)
                                         : std::vector<uint8_t>{0, 0, (uint8_t)compression, (uint8_t)ci->encryption, (uint8_t)PNA_MODE_GCM};
    uint8_t info[88];
    memcpy(info, "PNA-STREAM-v1", 13);
    sha256_bytes(name ? "FHED" : "SHED", 4, fh.data(), fh.size(), info + 13);
    memcpy(info + 45, phsf_hash, 32);
    memcpy(info + 77, salt_prefix + 32, 7);
    memcpy(info + 84, m.header + 39, 4);
    uint8_t ks[32];
    hkdf_sha256_32(ci->key, 32, salt_prefix, 32, info, 88, ks);
    aes256_expand(ks, m.rk);
    uint8_t zero[16] = {0}, hb[16], j0[16], eb[16];
    aes256_block_host(m.rk, zero, hb);
    memcpy(j0, salt_prefix + 32, 7); j0[7] = j0[8] = j0[9] = j0[10] = 0; j0[11] = 1;      // segment_nonce(prefix, 0, final)
    j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
    aes256_block_host(m.rk, j0, eb);
    for (int i = 0; i < 4; i++) {
        m.h[i] = ((uint32_t)hb[4 * i] << 24) | ((uint32_t)hb[4 * i + 1] << 16) | ((uint32_t)hb[4 * i + 2] << 8) | hb[4 * i + 3];
        m.ej0[i] = ((uint32_t)eb[4 * i] << 24) | ((uint32_t)eb[4 * i + 1] << 16) | ((uint32_t)eb[4 * i + 2] << 8) | eb[4 * i + 3];
    }
    memcpy(m.ctr_iv, j0, 16); m.ctr_iv[15] = 2;                                           // first data block: counter 2
}
// the per-entry IVs of a cipher job: the caller's, or random ones (random::random_vec(block_size) per entry, lib/src/entry/write.rs:108-112)
int resolve_ivs(pna_gpu_ctx *c, const pna_gpu_cipher *cipher, size_t n, std::vector<uint8_t> &own, const uint8_t **ivs) {
    int rc = check_cipher(c, cipher); if (rc) return rc;
    if (!cipher->phsf) return fail(c, PNA_E_INVAL, "cipher without a PHSF string");
    *ivs = cipher->ivs;
    if (*ivs) return PNA_OK;
    const size_t per = cipher->cipher_mode == PNA_MODE_GCM ? 39 : 16;  // GCM: salt(32) || nonce_prefix(7) per stream (to_hashed, write.rs:83-86)
    own.resize(n * per + 16);
    for (size_t o = 0; o < n * per;) {
        const ssize_t got = getrandom(own.data() + o, std::min<size_t>(n * per - o, 1u << 20), 0);
        if (got <= 0) return fail(c, PNA_E_INVAL, "getrandom failed");
        o += (size_t)got;
    }
    *ivs = own.data();
    return PNA_OK;
}

// names[e] for the batch's global entry index e; solid: one SDAT chunk per segment of the (single) entry; cipher + ivs (16 bytes per
// global entry index): the payloads are encrypted in place before their CRC-32 is taken
// largest FDAT chunk the device paths write: the CRC kernel takes "FDAT" || data as one message of at most 2^32 - 1 bytes (the reference's default cuts
// at u32::MAX: the same chunks unless an entry's compressed payload exceeds 4 GiB - 5 bytes)
uint64_t chunk_limit(uint32_t max_chunk) { return max_chunk ? std::min<uint64_t>(max_chunk, 0xFFFFFFFBull) : 0xFFFFFFFBull; }
size_t meta_len(const pna_gpu_entry_meta *m, size_t e) {
    if (!m) return 0;
    return (m->extra && m->extra_len ? m->extra_len[e] : 0) + (m->facets && m->facets_len ? m->facets_len[e] : 0);
}
// a blob of already framed chunks: lengths consistent, CRCs right, none of the chunk types this library writes itself
bool meta_blob_ok(const uint8_t *p, size_t n) {
    size_t pos = 0;
    while (pos < n) {
        if (n - pos < 12) return false;
        const uint32_t dl = ((uint32_t)p[pos] << 24) | ((uint32_t)p[pos + 1] << 16) | ((uint32_t)p[pos + 2] << 8) | p[pos + 3];
        if (n - pos - 12 < dl) return false;
        const uint8_t *ty = p + pos + 4;
        for (const char *own : {"FHED", "FDAT", "FEND", "fSIZ", "PHSF", "SHED", "SDAT", "SEND", "AHED", "AEND", "ANXT"}) if (memcmp(ty, own, 4) == 0) return false;
        const uint8_t *cp = p + pos + 8 + dl;
        if (pna_crc32(0, ty, 4 + (size_t)dl) != (((uint32_t)cp[0] << 24) | ((uint32_t)cp[1] << 16) | ((uint32_t)cp[2] << 8) | cp[3])) return false;
        pos += 12 + (size_t)dl;
    }
    return true;
}
// prefix = FHED | fSIZ | rest  ->  FHED | extra | fSIZ | facets | rest   (NormalEntry::write_chunks_to, lib/src/entry.rs:895-911)
static void splice_meta(std::vector<uint8_t> &pre, const pna_gpu_entry_meta *m, size_t e) {
    if (!meta_len(m, e)) return;
    const size_t c0 = 12 + ((size_t)pre[0] << 24 | (size_t)pre[1] << 16 | (size_t)pre[2] << 8 | pre[3]);
    const size_t c1 = 12 + ((size_t)pre[c0] << 24 | (size_t)pre[c0 + 1] << 16 | (size_t)pre[c0 + 2] << 8 | pre[c0 + 3]);
    std::vector<uint8_t> out(pre.begin(), pre.begin() + c0);
    if (m->extra && m->extra_len && m->extra_len[e]) out.insert(out.end(), (const uint8_t *)m->extra[e], (const uint8_t *)m->extra[e] + m->extra_len[e]);
    out.insert(out.end(), pre.begin() + c0, pre.begin() + c0 + c1);
    if (m->facets && m->facets_len && m->facets_len[e]) out.insert(out.end(), (const uint8_t *)m->facets[e], (const uint8_t *)m->facets[e] + m->facets_len[e]);
    out.insert(out.end(), pre.begin() + c0 + c1, pre.end());
    pre.swap(out);
}

// The LZ stage over segments [s0, s1) of a sub-batch.  Runs of more than 1 024 segments take the split form -- match kernel (k_lzm) + parse
// kernel (k_lzp) per run of at most `split_blocks` blocks, which meet in c->pbuf (4 bytes per input byte of the run; one run at a time on the
// stream, so runs share it) --, shorter ones and PNA_F_LZ_FUSED / PNA_LZ_SPLIT=0 the one-kernel form (k_lz<MODE 0>, no pbuf);
// PNA_F_LZ_WAVEPARSE / PNA_LZ_SPLIT=2: the split form as k_lz<MODE 1> + k_lz<MODE 2> (testing).  All forms give the same bytes.  If pbuf
// cannot be had, the run is halved down to 1 024 blocks, then fused.
// The SHORT segments among [s0, s1) behind a launch of the one-kernel form, which skips them (pna_dev.h SMALL_SEG): k_lzms + the parse kernel over their blocks, through
// the words workspace, in runs of blocks as the split form's
static int lz_small_pass(pna_gpu_ctx *c, const uint8_t *d_src, const SegDesc *segs, uint32_t nseg_all, uint32_t s0, uint32_t s1, uint32_t nblk, uint4 *ctab,
                         uint32_t flags, uint32_t max_len, hipStream_t st) {
    uint32_t run_blocks = (uint32_t)std::min<uint64_t>((uint64_t)c->tun.lz_split_blocks << (PNA_BLK_LOG - segs[s0].blk_log), 1u << 30);
    const uint32_t bps = 1u << (20 - segs[s0].blk_log);
    for (uint32_t a = s0; a < s1;) {
        const uint32_t b0 = segs[a].blk_base;
        uint32_t b = a + 1;
        {   // the first segment behind `a` whose blocks no longer fit the run (block bases grow with the index: a binary search -- 10^6 small entries made the linear walk a millisecond)
            uint32_t hi = s1;
            while (b < hi) { const uint32_t mid = b + (hi - b) / 2; if ((mid < nseg_all ? segs[mid].blk_base : nblk) - b0 + bps <= run_blocks) b = mid + 1; else hi = mid; }
        }
        const uint32_t b1 = b < nseg_all ? segs[b].blk_base : nblk;
        if (c->pbuf.ensure(((size_t)std::max<uint32_t>(b1 - b0, 1) << segs[a].blk_log) * 4)) {
            (void)hipGetLastError();
            if (run_blocks > 1024 && b - a > 1) { run_blocks /= 2; continue; }
            return fail(c, PNA_E_NOMEM, "no room for the short segments' words");
        }
        const LzParseGrid pg{c->d_segs, c->d_blk_seg, b1 - b0};
        launch_lz_small(d_src, c->d_segs + a, b - a, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, ctab, flags, max_len, st, (uint32_t *)c->pbuf.p, b0, &pg,
                        (flags & FLAG_LEN36) != 0);
        a = b;
    }
    return PNA_OK;
}

static int lz_stage(pna_gpu_ctx *c, const uint8_t *d_src, const SegDesc *segs, uint32_t nseg_all, uint32_t s0, uint32_t s1, uint32_t nblk, uint4 *ctab,
                    uint32_t flags, uint32_t max_off, uint32_t max_len, hipStream_t st, bool timed) {
    const int env_split = (int)c->tun.lz_split;
    // (the option counts blocks of 128 KiB: a run is that many BYTES of input whatever the batch's block size)
    const uint32_t env_blocks = (uint32_t)std::min<uint64_t>((uint64_t)c->tun.lz_split_blocks << (PNA_BLK_LOG - segs[s0].blk_log), 1u << 30);
    const uint32_t bps = 1u << (20 - segs[s0].blk_log);          // blocks of a full segment
    // zstd levels 10 .. 22 (the strong set): the match kernel's hash tables lie in global memory (2^19 slots per segment instead of the 24 512 LDS
    // holds; k_lz_split.hip) -- only k_lzm has that form, so those levels always take the split form, whatever the run's length
    const bool gt = !ctab && c->call_gtab;
    const bool fused = !gt && ((c->call_flags & PNA_F_LZ_FUSED) || env_split == 0 || (flags & 0x100u));   // (0x100: the phase stamps live in the fused kernel)
    const bool waveparse = !gt && ((c->call_flags & PNA_F_LZ_WAVEPARSE) || env_split == 2);
    uint32_t split_blocks = env_blocks;
    if (s1 > s0) {
        // several runs: of about equal size (whole rounds of 256 one-MiB segments) instead of full ones and a short tail -- a tail under the
        // split form's threshold would fall back to the slower one-kernel form (5 000 segments: 2 560 + 2 440 instead of 4 096 + 904)
        const uint32_t total = (s1 < nseg_all ? segs[s1].blk_base : nblk) - segs[s0].blk_base;
        const uint32_t nruns = (total + split_blocks - 1) / split_blocks;
        if (nruns > 1) {
            const uint32_t round = c->n_cus * bps, even = ((total + nruns - 1) / nruns + round - 1) / round * round;
            if (even < split_blocks) split_blocks = even;
        }
    }
    const uint32_t s1_all = s1; bool fused_tail = false;
    // The split form at every size: its parse kernel runs one wave per BLOCK (a block's parse depends on nothing outside the block), so a short run no
    // longer waits for one wave's walk over a whole segment (N x 1 MiB, one-kernel / split: 32: 2.3 / 1.6 ms, 256: 2.4 / 1.8, 1 024: 9.3 / 6.3,
    // 2 048: 18.6 / 12.3; scripts/lz_forms.py).  The one-kernel form remains for PNA_F_LZ_FUSED, for a workspace that cannot be allocated and for the
    // units of the latency mode; lz_split_min restores a threshold.
    const uint32_t min_segs = gt ? 0u : (uint32_t)c->tun.lz_split_min;
    for (uint32_t a = s0; a < s1 && !fused;) {
        const uint32_t b0 = segs[a].blk_base;
        uint32_t b = a + 1;
        {   // the first segment behind `a` whose blocks no longer fit the run (block bases grow with the index: a binary search -- 10^6 small entries made the linear walk a millisecond)
            uint32_t hi = s1;
            while (b < hi) { const uint32_t mid = b + (hi - b) / 2; if ((mid < nseg_all ? segs[mid].blk_base : nblk) - b0 + bps <= split_blocks) b = mid + 1; else hi = mid; }
        }
        const uint32_t b1 = b < nseg_all ? segs[b].blk_base : nblk;
        if (b - a < min_segs && !waveparse) { s0 = a; s1 = b; fused_tail = b < s1_all; break; }
        if ((c->tun.lz_pbuf_fail && !gt) /* testing: as if the allocation failed */ || c->pbuf.ensure(((size_t)std::max<uint32_t>(b1 - b0, 1) << segs[a].blk_log) * 4) ||
            (gt && c->gtab.ensure((size_t)(b - a) << (lz_gtab_log() + 2)))) {
            (void)hipGetLastError();                                   // (the failed allocation's sticky code)
            if (split_blocks > 1024 && b - a > 1) { split_blocks /= 2; continue; }                    // (for this call only: the next one tries the full run again)
            if (gt) return fail(c, PNA_E_NOMEM, "no room for the strong level set's hash tables");    // (the one-kernel form has LDS tables: other bytes)
            s0 = a; break;                                             // no room for the words: the rest goes through the fused kernel
        }
        hipEvent_t e1 = nullptr;
        if (timed) {
            while (c->lzm_ev.size() < c->lzm_used + 2) { hipEvent_t e = nullptr; HIPCHK(c, hipEventCreate(&e)); c->lzm_ev.push_back(e); }
            HIPCHK(c, hipEventRecord(c->lzm_ev[c->lzm_used], st)); e1 = c->lzm_ev[c->lzm_used + 1]; c->lzm_used += 2; c->lzm_nl.push_back(1);
        }
        LzParseGrid pg{c->d_segs, c->d_blk_seg, b1 - b0};                  // the parse kernel: one wave per block of the run
        pg.hist = c->lzp_hist;                                            // (the sequence codes' counters, where the entropy stage takes them from the parse kernel)
        if (waveparse) c->lzp_hist_all = false;
        // The match kernel runs one workgroup per segment and CU, all of equal length: a run of 3 334 segments is 13 full rounds of the 256 CUs and a 14th for
        // 6 of them.  The segments behind the last full round are therefore cut into UNITS of one block each (table pre-warmed: the same words, SS4a), a launch
        // of their own behind the full rounds: R x 8 short workgroups instead of R long ones next to 256 - R idle CUs.
        uint32_t R = (!gt && !waveparse && c->tun.tail_units && b - a >= 2 * c->n_cus) ? (b - a) % c->n_cus : 0;     // (n_cus: the device's compute units, 256 on an MI355X)
        if (R > 3 * c->n_cus / 8 || (flags & FLAG_ALL_SMALL)) R = 0;
        if (R) {
            const uint32_t bl = segs[a].blk_log, bs = 1u << bl;
            size_t nu = 0;
            for (uint32_t sgi = b - R; sgi < b; sgi++) nu += std::max<uint32_t>(1, (segs[sgi].len + bs - 1) >> bl);
            constexpr size_t TAIL_CAP = 16 * 96 * 128;                 // (allocated once at this size: a buffer that grew would move under the launches already queued)
            if (c->tail_used + nu > TAIL_CAP || c->h_tail.ensure(TAIL_CAP * sizeof(SegDesc)) || c->d_tail.ensure(TAIL_CAP * sizeof(SegDesc))) R = 0;
            else {
                SegDesc *hu = (SegDesc *)c->h_tail.p + c->tail_used; SegDesc *du = (SegDesc *)c->d_tail.p + c->tail_used;
                size_t k = 0;
                for (uint32_t sgi = b - R; sgi < b; sgi++)
                    for (uint32_t u = 0; u < std::max<uint32_t>(segs[sgi].len, 1); u += bs) { SegDesc us = segs[sgi]; us.u0 = u; us.u1 = std::min<uint32_t>(segs[sgi].len, u + bs); hu[k++] = us; }
                HIPCHK(c, hipMemcpyAsync(du, hu, nu * sizeof(SegDesc), hipMemcpyHostToDevice, st));
                launch_lz(d_src, c->d_segs + a, b - a - R, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, ctab, flags, max_off, max_len, st,
                          (uint32_t *)c->pbuf.p, b0, nullptr, nullptr, nullptr);                        // the full rounds: match kernel only
                launch_lz(d_src, du, (uint32_t)nu, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, ctab, flags, max_off, max_len, st,
                          (uint32_t *)c->pbuf.p, b0, e1, nullptr, &pg);                                 // the rest in units, then the parse kernel over the whole run
                c->tail_used += nu;
                if (timed) c->lzm_nl.back() = 2;                               // (the pair of events spans both launches of the match kernel)
            }
        }
        if (!R)
        launch_lz(d_src, c->d_segs + a, b - a, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, ctab, flags | (waveparse ? 0x1000u : 0u), max_off, max_len, st,
                  (uint32_t *)c->pbuf.p, b0, e1, gt ? (uint32_t *)c->gtab.p : nullptr, &pg);
        a = b;
        if (a >= s1) return PNA_OK;
    }
    c->lzp_hist_all = false;                                           // (the one-kernel form: its parse counts nothing)
    if (!(flags & FLAG_ALL_SMALL)) launch_lz(d_src, c->d_segs + s0, s1 - s0, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, ctab, flags, max_off, max_len, st, nullptr, 0, nullptr, nullptr, nullptr);
    if (flags & FLAG_HAS_SMALL) { const int rc = lz_small_pass(c, d_src, segs, nseg_all, s0, s1, nblk, ctab, flags, max_len, st); if (rc) return rc; }   // (the one-kernel form skips the short segments)
    if (fused_tail) return lz_stage(c, d_src, segs, nseg_all, s1, s1_all, nblk, ctab, flags, max_off, max_len, st, timed);   // (a short run in the middle: only with tiny PNA_LZ_SPLIT_BLOCKS)
    return PNA_OK;
}

// stage times of a finished sub-batch from its events (the stream has been waited for)
static int collect_timing(pna_gpu_ctx *c, bool defl, int nch, uint32_t nseg, uint32_t nblk, bool with_cipher) {
    float ms[6] = {0, 0, 0, 0, 0, 0}, msf = 0;
    (void)hipEventElapsedTime(&msf, c->ev[6], c->ev[7]);
    c->timing.ms_frame += msf;
    float mc = 0;                                             // the cipher kernels run inside the "pack" interval: report them apart
    if (with_cipher) { (void)hipEventElapsedTime(&mc, c->ev_ci[0], c->ev_ci[1]); c->timing.ms_cipher += mc; c->timing.ms_pack -= mc; }
    if (defl) {
        (void)hipEventElapsedTime(&ms[0], c->ev[0], c->ev[1]);
        (void)hipEventElapsedTime(&ms[1], c->ev[1], c->ev[2]);
        (void)hipEventElapsedTime(&ms[2], c->ev[2], c->ev[3]);
        (void)hipEventElapsedTime(&ms[3], c->ev[3], c->ev[4]);
        (void)hipEventElapsedTime(&ms[4], c->ev[4], c->ev[5]);
        (void)hipEventElapsedTime(&ms[5], c->ev[5], c->ev[6]);
    } else {
        // k_lz: first launch to last completion on the main stream; the entropy stages are summed over the chunks on the
        // auxiliary stream (with more than one chunk they overlap k_lz and add up to more than the wall time); "pack" = from the
        // end of the last chunk's entropy stage to the end of the write kernels (plan + scan + layout + write)
        (void)hipEventElapsedTime(&ms[0], c->ev_lz[0], c->ev_lz[nch]);
        for (int k = 0; k < nch; k++) {
            float a = 0, b2 = 0, d = 0;
            (void)hipEventElapsedTime(&a, c->ev_en[k][0], c->ev_en[k][1]);
            (void)hipEventElapsedTime(&b2, c->ev_en[k][1], c->ev_en[k][2]);
            (void)hipEventElapsedTime(&d, c->ev_en[k][2], c->ev_en[k][3]);
            ms[1] += a; ms[2] += b2; ms[3] += d;
        }
        (void)hipEventElapsedTime(&ms[4], c->ev_en[nch - 1][3], c->ev[6]);
    }
    c->timing.ms_lz += ms[0]; c->timing.ms_stats += ms[1]; c->timing.ms_lit += ms[2]; c->timing.ms_seq += ms[3];
    c->timing.ms_pack += ms[4] + ms[5];
    for (size_t i = 0; i + 1 < c->lzm_used; i += 2) { float m = 0; (void)hipEventElapsedTime(&m, c->lzm_ev[i], c->lzm_ev[i + 1]); c->timing.ms_lz_match += m; c->timing.lz_match_launches += i / 2 < c->lzm_nl.size() ? c->lzm_nl[i / 2] : 1u; }
    c->timing.n_segments += nseg; c->timing.n_blocks += nblk;
    return PNA_OK;
}

int run_subbatch(pna_gpu_ctx *c, int algo, const uint8_t *d_src, const uint64_t *src_off, const uint64_t *src_len,
                        size_t e0, size_t e1, uint8_t *d_dst, size_t dst_cap, uint64_t out_base, uint64_t *dst_off,
                        hipStream_t st, bool timed, const FrameJob *fj) {
    // LATENCY MODE: a small batch -- the CompressionWriter seam with a handful of writers in flight, one entry of
    // `pna_gpu_compress_batch` -- has fewer segments than the chip has CUs, and its time is the length of the per-segment and per-block serial
    // chains (one workgroup walks a segment's 256 tiles; one lane codes a block's sequences).  Such a batch is cut finer: blocks of 8 .. 64 KiB
    // inside the same frames (blk_log), and the LZ stage runs one workgroup per UNIT of 1 << unit_log bytes whose table is pre-warmed with
    // everything before it (lz_prewarm), which gives the very matches of the segment-long walk.  Both follow from the batch's size alone
    // (and pna_gpu_set_option), are reported by pna_gpu_last_timing, and are parameters of the oracle's model.
    // option trace: the HOST's time line of the sub-batch (what it does before and next to the kernels), printed when the call ends
    const auto t_host0 = std::chrono::steady_clock::now();
    std::vector<std::pair<const char *, double>> host_marks;
    auto mark = [&](const char *what) { if (c->tun.trace) host_marks.emplace_back(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host0).count()); };
    auto print_marks = [&]() { if (c->tun.trace && !host_marks.empty()) { fprintf(stderr, "[pna sub-batch] %zu entries, host:", e1 - e0); for (auto &m : host_marks) fprintf(stderr, "  %s %.2f", m.first, m.second); fprintf(stderr, " ms\n"); } };
    const size_t ne_all = e1 - e0;
    const unsigned host_nt = host_loop_threads(ne_all);
    struct PlanPart { uint64_t in_total = 0, nseg_est = 0, max_len = 0, n_short = 0, n_mid = 0, max_mid = 0; uint32_t sg = 0, bk = 0, un = 0; bool misaligned = false, any_empty = false; };
    std::vector<PlanPart> pp(host_nt);
    par_ranges(ne_all, host_nt, [&](unsigned t, size_t a, size_t b) {
        PlanPart q;
        for (size_t e = e0 + a; e < e0 + b; e++) {
            const uint64_t len = src_len[e];
            q.in_total += len; q.nseg_est += len ? (len + SEG_SIZE - 1) / SEG_SIZE : 1; q.max_len = std::max<uint64_t>(q.max_len, len);
            q.misaligned |= (src_off[e] & 15) != 0; q.any_empty |= len == 0;
            // SHORT segments (pna_dev.h SMALL_SEG): an entry of at most that many bytes, or the last segment of a longer one
            const uint64_t last = len ? ((len - 1) & (SEG_SIZE - 1)) + 1 : 0;
            q.n_short += (last > 0 && last <= SMALL_SEG) ? 1u : 0u; if (last > SMALL_SEG && last <= MID_SEG) { q.n_mid++; q.max_mid = std::max(q.max_mid, last); }
        }
        pp[t] = q;
    });
    uint64_t in_total = 0, nseg_est = 0, max_len = 0, n_short = 0, n_mid = 0, max_mid = 0;
    bool any_empty = false;
    for (const PlanPart &q : pp) {
        in_total += q.in_total; nseg_est += q.nseg_est; max_len = std::max(max_len, q.max_len); any_empty |= q.any_empty; n_short += q.n_short; n_mid += q.n_mid; max_mid = std::max(max_mid, q.max_mid);
        if (q.misaligned) return fail(c, PNA_E_INVAL, "entry offset not 16-byte aligned");
    }
    // an upper bound of every payload of the sub-batch when the entries are small and plain (k_frame's wave-per-entry form takes those; 0: no such bound)
    const uint32_t frame_max_payload = (fj && !fj->solid && !fj->cipher && max_len <= 16384) ? (uint32_t)std::min<size_t>(pna_gpu_bound(algo, (size_t)max_len), 0xFFFFFFFFu) : 0u;
    const bool latency = c->tun.latency_max_mib > 0 && in_total <= ((uint64_t)c->tun.latency_max_mib << 20) && nseg_est <= 1024 && !(c->call_flags & 0x100u);
    uint32_t blk_log = blk_log_for_longest(c, max_len), unit_log = 20;
    if (blk_log == PNA_BLK_LOG && algo != PNA_ALGO_ZSTD && c->tun.latency_max_mib > 0 && !(c->call_flags & 0x100u) && !(latency && in_total <= (64ull << 20))) {
        // deflate beyond 64 MiB of input: whole segments, and 64 KiB blocks up to 384 MiB of the call's input (96 / 192 / 256 MiB: 1.78 / 2.19 / 2.52 -> 1.66 / 1.99 / 2.20 ms;
        // ratio 2.540 -> 2.536: every dynamic block repeats the code description, so the blocks stay larger than zstd's)
        if (std::max<uint64_t>(in_total, c->call_total) <= (384ull << 20)) blk_log = 16;
    } else if (latency && blk_log == PNA_BLK_LOG && algo != PNA_ALGO_ZSTD) {
        // blocks: 16 KiB up to 16 MiB of input (a block's sequence chain then is ~1 000 steps), then growing with the batch so that the
        // block count -- per-block fixed costs of the entropy kernels -- stays near 1 024 .. 2 048
        blk_log = 14;
        while (blk_log < PNA_BLK_LOG && (in_total >> blk_log) > 2048) blk_log++;
        // units: about one per CU (256), never smaller than a block
        unit_log = blk_log;
        while (unit_log < 20 && (in_total >> unit_log) > 384) unit_log++;
    } else if (blk_log == PNA_BLK_LOG && algo == PNA_ALGO_ZSTD && c->tun.latency_max_mib > 0 && !(c->call_flags & 0x100u)) {       // (latency_max_mib = 0: 128 KiB blocks whatever the batch)
        // zstd, measured again on the round's final kernels (LAB_LOG.md 4.10: one device batch of n x 1 MiB): what a batch below ~1 GiB costs is its blocks' chains, so
        // the blocks stay small well beyond the latency mode -- 16 KiB up to 32 MiB of input, 32 KiB up to 384 MiB, 64 KiB up to 1 GiB (256 MiB: 3.3 -> 2.5 ms, 512: 5.2 -> 4.6;
        // ratio 2.847 -> 2.844 / 2.846), the call's whole input deciding where a large call is cut into sub-batches --, and UNITS pay only while they fit ONE round of
        // the chip's CUs (every unit replays the packed table's walk up to its start): up to 64 MiB of input; beyond, whole segments (80 MiB: 2.55 -> 2.0 ms)
        const uint64_t sz = std::max<uint64_t>(in_total, c->call_total);
        blk_log = sz <= (32ull << 20) ? 14u : (sz <= (384ull << 20) ? 15u : (sz <= (1ull << 30) ? 16u : (uint32_t)PNA_BLK_LOG));
        if (latency && in_total <= (64ull << 20)) {
            unit_log = blk_log;
            while (unit_log < 20 && (in_total >> unit_log) > c->n_cus) unit_log++;
        }
    }
    if (c->tun.unit_log) unit_log = (uint32_t)std::max<long>(c->tun.unit_log, blk_log);
    if (unit_log < blk_log) unit_log = blk_log;
    const uint32_t bsz = 1u << blk_log;
    // The plan: counted first, then written straight into the page-locked blob that travels to the device in one copy (several threads for the
    // batches of 10^5 .. 10^6 small entries, where this loop is a tenth of the call).  Layout: [segs | units | blk_seg | entry_first_seg].
    // first segment / block / unit of every range of entries (the ranges of par_ranges): counted per range, then a prefix sum over the ranges
    par_ranges(ne_all, host_nt, [&](unsigned t, size_t a, size_t b) {
        uint32_t sg = 0, bk = 0, un = 0;
        for (size_t e = e0 + a; e < e0 + b; e++) {
            const uint64_t len = src_len[e];
            if (len == 0) { sg++; continue; }
            const uint64_t full = len >> 20, rest = len & (SEG_SIZE - 1);
            sg += (uint32_t)(full + (rest ? 1 : 0));
            bk += (uint32_t)((full << (20 - blk_log)) + ((rest + bsz - 1) >> blk_log));
            if (unit_log < 20) un += (uint32_t)((full << (20 - unit_log)) + ((rest + (1u << unit_log) - 1) >> unit_log));
        }
        pp[t].sg = sg; pp[t].bk = bk; pp[t].un = un;
    });
    uint32_t nseg = 0, nblk = 0, nunits = 0;
    for (PlanPart &q : pp) { const uint32_t a = q.sg, b = q.bk, u = q.un; q.sg = nseg; q.bk = nblk; q.un = nunits; nseg += a; nblk += b; nunits += u; }

    if (nseg == 0) return PNA_OK;
    mark("counted");
    const size_t o_units = ((size_t)nseg * sizeof(SegDesc) + 15) & ~(size_t)15, o_blkseg = (o_units + (size_t)nunits * sizeof(SegDesc) + 15) & ~(size_t)15,
                 o_entry = (o_blkseg + (size_t)(nblk + 1) * 4 + 15) & ~(size_t)15, plan_bytes = o_entry + (ne_all + 2) * 4;
    if (c->plan.ensure(plan_bytes) || c->h_plan.ensure(plan_bytes)) return fail(c, PNA_E_NOMEM, "workspace allocation failed");
    SegDesc *segs = (SegDesc *)c->h_plan.p, *units = (SegDesc *)((uint8_t *)c->h_plan.p + o_units);   // (the previous sub-batch has been waited for: the staging is free)
    c->tail_used = 0;
    uint32_t *blk_seg = (uint32_t *)((uint8_t *)c->h_plan.p + o_blkseg), *entry_first_seg = (uint32_t *)((uint8_t *)c->h_plan.p + o_entry);
    // option single_frame (zstd): an entry's segments form ONE frame -- the frame header in front of the first segment only, the last-block bit on the entry's
    // last block only (SegDesc::first bit 2 tells k_plan / k_write); the blocks are what they are in the frame-per-segment form
    const uint32_t sf_bit = (algo == PNA_ALGO_ZSTD && c->tun.single_frame) ? 4u : 0u;
    {
        auto fill = [&](unsigned t, size_t a, size_t b) {
            uint32_t sg = pp[t].sg, bk = pp[t].bk, un = pp[t].un;
            for (size_t e = e0 + a; e < e0 + b; e++) {
                entry_first_seg[e - e0] = sg;
                const uint64_t len = src_len[e], off = src_off[e];
                if (len == 0) { segs[sg++] = SegDesc{off, 0, bk, (uint32_t)e, 3, 0, 0, blk_log, 0}; continue; }
                for (uint64_t p = 0; p < len; p += SEG_SIZE) {
                    const uint32_t sl = (uint32_t)std::min<uint64_t>(SEG_SIZE, len - p);
                    const SegDesc s{off + p, sl, bk, (uint32_t)e, (p == 0 ? 1u : 0u) | (p + SEG_SIZE >= len ? 2u : 0u) | sf_bit, 0, sl, blk_log, 0};
                    const uint32_t nb = (sl + bsz - 1) >> blk_log;
                    for (uint32_t b2 = 0; b2 < nb; b2++) blk_seg[bk + b2] = sg;
                    bk += nb; segs[sg++] = s;
                    if (unit_log < 20)
                        for (uint32_t u = 0; u < sl; u += 1u << unit_log) { SegDesc us = s; us.u0 = u; us.u1 = std::min<uint32_t>(sl, u + (1u << unit_log)); units[un++] = us; }
                }
            }
        };
        par_ranges(ne_all, host_nt, fill);
        entry_first_seg[ne_all] = nseg;
    }
    const bool unit_mode = unit_log < 20 && nunits > 0;
    c->last_blk_log = blk_log; c->last_units = unit_mode ? nunits : 0;
    // zstd entropy stage, two forms: statistics per block (k_hist) + tables + three-lane state chains (k_seqa) + token-parallel packing (k_seqb) while
    // the chain waves fit the chip's SIMDs (<= 40 960 blocks); beyond, statistics per segment inside k_stats and the one-kernel coder k_seq.  Flags
    // 0x1000 / 0x2000 and option hist_by_block force one.
    // (batches whose segments hold one block each -- entries of at most a block --: statistics per segment and the sequence coder with a lane per segment)
    const bool single_block = max_len <= bsz;
    const bool hist_on = algo == PNA_ALGO_ZSTD && !(c->call_flags & 0x1000u) && !single_block &&
                         ((c->call_flags & 0x2000u) || (c->tun.hist_by_block < 0 ? nblk <= 40960u : c->tun.hist_by_block != 0));
    // round 5: large zstd batches behind the split LZ stage -- the parse kernel counts the sequence codes per block into the segments' counters, k_stats walks the literals only
    const bool seq_hist = algo == PNA_ALGO_ZSTD && !hist_on && !single_block && c->tun.seq_hist != 0;
    const size_t o_hist = ((size_t)(nblk + 1) * sizeof(BlkInfo) + 15) & ~(size_t)15, blk_bytes = o_hist + ((hist_on || seq_hist) ? (size_t)nseg * 448 * 4 : 0);
    if (c->blk.ensure(blk_bytes) || c->tabs.ensure((size_t)nseg * std::max(sizeof(SegTables), sizeof(DeflTables))) ||
        (algo == PNA_ALGO_DEFLATE && c->ctab.ensure(((size_t)(nblk + 1) << (blk_log - 11)) * 16)) ||
        c->seqs.ensure((size_t)(nblk + 1) * seq_cap_of(blk_log) * 8) || c->lits.ensure((size_t)(nblk + 1) << blk_log) ||
        c->litc.ensure((size_t)(nblk + 1) << blk_log) || (algo == PNA_ALGO_ZSTD && c->seqc.ensure((size_t)(nblk + 1) << blk_log)) ||
        (algo == PNA_ALGO_ZSTD && c->seqw.ensure(hist_on ? (size_t)(nblk + 1) * seq_cap_of(blk_log) * 8 : 64)) ||
        c->seg_size.ensure((size_t)nseg * 8) || c->seg_off.ensure(((size_t)nseg + 1 + 2 * ((size_t)nseg / 4096 + 2)) * 8))       // (+ the scratch of the hierarchical scan)
        return fail(c, PNA_E_NOMEM, "workspace allocation failed");
    {
        uint8_t *hp = (uint8_t *)c->h_plan.p, *dp = (uint8_t *)c->plan.p;
        c->d_segs = (SegDesc *)dp; c->d_units = (SegDesc *)(dp + o_units); c->d_blk_seg = (uint32_t *)(dp + o_blkseg); c->d_entry_seg = (uint32_t *)(dp + o_entry);
        c->d_hist = (uint32_t *)((uint8_t *)c->blk.p + o_hist);
        HIPCHK(c, hipMemcpyAsync(dp, hp, plan_bytes, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemsetAsync(c->blk.p, 0, blk_bytes, st));                     // BlkInfo of every block and, behind them, the segments' histogram counters
    }
    c->lzm_used = 0; c->lzm_nl.clear();
    c->lzp_hist = nullptr; c->lzp_hist_all = false;
    const bool defl = algo == PNA_ALGO_DEFLATE;
    // the short segments' geometry (k_lzms): a launch flag tells the large geometry's kernels to skip them; a sub-batch of short segments only launches none of those
    const uint32_t small_fl = (c->tun.small_geometry && (n_short || n_mid))
        ? (FLAG_HAS_SMALL | (n_short ? FLAG_TIER1 : 0u) | (n_mid ? FLAG_TIER2 | ((max_mid <= 8192 ? 0u : (max_mid <= 12288 ? 1u : 2u)) << FLAG_T2_SHIFT) : 0u) | (max_len <= MID_SEG ? FLAG_ALL_SMALL : 0u)) : 0u;
    mark("plan queued");
    if (timed) HIPCHK(c, hipEventRecord(c->ev[0], st));
    int nch = 1;
    if (defl) {
        const uint32_t dfl = (c->call_flags & (F_LAZY | F_ADOPT | F_INS2 | F_STRONG | 0x300u)) | (c->call_lazy2 ? FLAG_LAZY2 : 0u) | (c->call_lazy3 ? FLAG_LAZY3 : 0u) | FLAG_LEN36 | small_fl;
        if (c->call_stored) { }                                // deflate level 0 = Compression::none(): stored blocks only, no match finder, no codes
        else if (unit_mode) {
            if (!(dfl & FLAG_ALL_SMALL)) launch_lz(d_src, c->d_units, nunits, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, (uint4 *)c->ctab.p, dfl, 32768u, 258u, st, nullptr, 0, nullptr, nullptr, nullptr);
            if (dfl & FLAG_HAS_SMALL) { const int rc = lz_small_pass(c, d_src, segs, nseg, 0, nseg, nblk, (uint4 *)c->ctab.p, dfl, 258u, st); if (rc) return rc; }
        }
        else { const int rc = lz_stage(c, d_src, segs, nseg, 0, nseg, nblk, (uint4 *)c->ctab.p, dfl, 32768u, 258u, st, timed); if (rc) return rc; }
        if (timed) HIPCHK(c, hipEventRecord(c->ev[1], st));
        launch_deflate_stage1(d_src, c->d_segs, nseg, c->d_blk_seg, nblk, (const uint64_t *)c->seqs.p,
                              (const uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, (const uint4 *)c->ctab.p, (DeflTables *)c->tabs.p, (uint8_t *)c->litc.p,
                              (uint64_t *)c->seg_size.p, (uint64_t *)c->seg_off.p, st, timed ? &c->ev[2] : nullptr, c->call_flags, c->call_stored,
                              /* a wave per segment: many small entries */ max_len <= 32768 && nseg >= 4096);
    } else {
        // zstd: the segments go through k_lz in chunks on `st`; the entropy stage of a finished chunk runs on the auxiliary
        // stream next to the following chunk's k_lz (latency-bound kernels hide in the issue slots k_lz leaves free)
        if (!c->aux) {
            { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // (lowest priority: what runs here fills in beside the main stream's kernels)
              HIPCHK(c, hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, lo)); }
            for (auto &e : c->ev_lz) HIPCHK(c, hipEventCreate(&e));
            for (auto &r : c->ev_en) for (auto &e : r) HIPCHK(c, hipEventCreate(&e));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        }
        // Measured (10 000 x 1 MiB): 4 chunks 111.8 ms vs 108.9 ms unchunked -- k_seq's duration is set by the length of one
        // block's tANS chain, not by the number of blocks, so every chunk pays it in full and the co-resident waves slow k_lz
        // by 12 %.  The chunked form therefore stays off unless PNA_PIPELINE_CHUNKS asks for it.
        nch = (int)c->tun.pipeline_chunks; if (nch < 1) nch = 1; if (nch > pna_gpu_ctx::MAXCH) nch = pna_gpu_ctx::MAXCH; if ((uint32_t)nch > nseg || latency) nch = 1;
        HIPCHK(c, hipEventRecord(c->ev_lz[0], st));
        for (int k = 0; k < nch; k++) {
            const uint32_t s0 = (uint32_t)((uint64_t)nseg * k / nch), s1 = (uint32_t)((uint64_t)nseg * (k + 1) / nch);
            const uint32_t g0 = segs[s0].blk_base, g1 = s1 < nseg ? segs[s1].blk_base : nblk;
            const uint32_t zfl = (c->call_flags & 0x3FFu) | (c->call_w32 ? (c->call_w16 ? FLAG_W16 : FLAG_W32) : 0u) | (c->call_lazy2 ? FLAG_LAZY2 : 0u) | (c->call_lazy3 ? FLAG_LAZY3 : 0u) | (c->call_gtab ? 0u : FLAG_LEN36) | (c->call_tab3 ? FLAG_TAB3 : 0u) | ((c->call_tab3 && !c->call_w16 && c->tun.far1) ? FLAG_FAR1 : 0u) | (c->call_strong2 ? FLAG_STRONG2 : 0u) | small_fl;
            const uint32_t zmax = (c->call_flags & F_FAR) ? (c->call_gtab ? MAX_OFF : MAX_OFF_W3) : NEAR_OFF;   // (3-byte words keep 19 bits of offset)
            if (unit_mode) {
                // (nch == 1: one launch over all units; the strong set: split form over the units, tables in global memory)
                const bool gt = c->call_gtab;
                const LzParseGrid pgu{c->d_segs, c->d_blk_seg, nblk};
                if (gt && (c->pbuf.ensure(((size_t)nblk << blk_log) * 4) || c->gtab.ensure((size_t)nunits << (lz_gtab_log() + 2)))) return fail(c, PNA_E_NOMEM, "no room for the strong level set's hash tables");
                if (gt || !(zfl & FLAG_ALL_SMALL))
                launch_lz(d_src, c->d_units, nunits, (uint64_t *)c->seqs.p, (uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, nullptr, zfl,
                          zmax, 0xFFFFFFFFu, st, gt ? (uint32_t *)c->pbuf.p : nullptr, 0, nullptr, gt ? (uint32_t *)c->gtab.p : nullptr, gt ? &pgu : nullptr);
                if (!gt && (zfl & FLAG_HAS_SMALL)) { const int rc = lz_small_pass(c, d_src, segs, nseg, 0, nseg, nblk, nullptr, zfl, 0xFFFFFFFFu, st); if (rc) return rc; }   // (the split form above takes them itself)
            }
            else {
                c->lzp_hist = (seq_hist && nch == 1) ? c->d_hist : nullptr; c->lzp_hist_all = c->lzp_hist != nullptr;
                const int rc = lz_stage(c, d_src, segs, nseg, s0, s1, nblk, nullptr, zfl, zmax, 0xFFFFFFFFu, st, timed); if (rc) return rc;
            }
            // (one chunk: everything stays on `st` -- a hand-over to the auxiliary stream and back costs ~45 us of idle device, a tenth of a small batch)
            hipStream_t est = nch > 1 ? c->aux : st;
            HIPCHK(c, hipEventRecord(c->ev_lz[k + 1], st));
            if (nch > 1) HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_lz[k + 1], 0));
            HIPCHK(c, hipEventRecord(c->ev_en[k][0], est));
            launch_entropy_chunk(c->d_segs, s0, s1 - s0, c->d_blk_seg, g0, g1 - g0, (const uint64_t *)c->seqs.p,
                                 (const uint8_t *)c->lits.p, (BlkInfo *)c->blk.p, (SegTables *)c->tabs.p, (uint8_t *)c->litc.p, (uint8_t *)c->seqc.p,
                                 (uint32_t *)c->seqw.p, c->call_flags, blk_log, hist_on ? c->d_hist : nullptr, est, &c->ev_en[k][1],
                                 (nch == 1 && c->tun.lit_beside_seq) ? c->aux : nullptr, c->ev_fork, c->ev_join, single_block, c->lzp_hist_all ? c->d_hist : nullptr);
        }
        if (nch > 1) { HIPCHK(c, hipEventRecord(c->ev_join, c->aux)); HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0)); }
        if (timed) HIPCHK(c, hipEventRecord(c->ev[4], st));
        launch_plan(c->d_segs, nseg, (BlkInfo *)c->blk.p, (const SegTables *)c->tabs.p, (uint64_t *)c->seg_size.p,
                    (uint64_t *)c->seg_off.p, c->call_flags, st);
        if (timed) HIPCHK(c, hipEventRecord(c->ev[5], st));
    }
    HIPCHK(c, hipGetLastError());
    // while the kernels run: the name-dependent part of every entry record (FHED and fSIZ chunks with their CRCs)
    mark("kernels queued");
    uint64_t layout_need = 0; bool layout_over = false;         // the sub-batch's worst-case size with its prefixes; an entry whose worst case exceeds one FDAT chunk
    FrameDesc *fds = nullptr; uint8_t *blob = nullptr; uint64_t *segdst = nullptr; size_t blob_len = 0;
    const bool solid = fj && fj->solid;
    const bool gcm = fj && fj->cipher && fj->cipher->cipher_mode == PNA_MODE_GCM;
    const uint32_t gcm_seg = gcm ? (fj->cipher->gcm_segment_size ? fj->cipher->gcm_segment_size : (1u << 20)) : 0u;   // default = the reference's DEFAULT_SEGMENT_SIZE (1 MiB)
    std::vector<GcmMaterial> gmat; std::vector<GcmEntry> gents;
    // GCM STREAM segments of this sub-batch, in order (an entry has ceil(payload / segment_size) of them, at least one): counter-mode IV
    // (nonce || 2) and the entry they belong to; `spread`: pieces of the compact payload that move to their place between the tags
    struct GcmSeg { uint8_t ctr_iv[16]; uint32_t entry; };
    std::vector<GcmSeg> gsegs;
    struct SpreadPiece { uint64_t src, dst; uint32_t len; };
    std::vector<SpreadPiece> spread; uint64_t spread_bytes = 0;
    std::vector<std::pair<uint64_t, uint64_t>> spread_copy;     // (archive offset, length) of the compact payloads to save first
    size_t nunit = e1 - e0;                                    // framed units: entries, or the segments of the solid stream
    if (solid) {
        if (e1 - e0 != 1) return fail(c, PNA_E_INVAL, "a solid stream is one entry");
        nunit = nseg;
        // (GCM: one SDAT chunk per GCM segment of the compressed stream -- at most the worst-case output / segment size + 1 of them)
        const size_t ucap = gcm ? (size_t)(pna_gpu_bound(algo, (size_t)src_len[e0]) / gcm_seg) + 2 : nunit;
        if (c->h_desc.ensure(ucap * sizeof(FrameDesc)) || c->h_blob.ensure(ucap * 8 + 16) || c->h_segdst.ensure((size_t)(nseg + 1) * 8))
            return fail(c, PNA_E_NOMEM, "framing staging");
        fds = (FrameDesc *)c->h_desc.p; blob = (uint8_t *)c->h_blob.p; segdst = (uint64_t *)c->h_segdst.p;
        if (gcm) {
            gmat.resize(1);
            uint8_t kc[32], ph[32];
            hkdf_sha256_32(fj->cipher->key, 32, nullptr, 0, "PNA-KC-v1", 9, kc);
            sha256_bytes(fj->cipher->phsf, strlen(fj->cipher->phsf), nullptr, 0, ph);
            gcm_entry_material(fj->cipher, kc, ph, fj->ivs, gcm_seg, nullptr, algo, gmat[0]);
        }
    } else if (fj) {
        std::vector<uint8_t> tmp;
        // bounds, one pass over the entries on the host-loop threads: the prefixes' worst case per range of entries (a range's prefixes are written
        // from its bound offset on, so the ranges need no compaction pass afterwards), the extra FDAT chunks (every chunk behind an entry's first needs a
        // descriptor and 8 prefix bytes: at most one per max_chunk_size bytes of the worst-case output), and what the device-side layout must know --
        // whether every worst-case payload fits one chunk, and the worst-case size of the sub-batch
        struct FramePart { size_t bound = 0, extra = 0; uint64_t need = 0; bool over = false; };
        std::vector<FramePart> fp(host_nt);
        const uint64_t CHb = chunk_limit(fj->max_chunk);
        par_ranges(ne_all, host_nt, [&](unsigned t, size_t a, size_t b) {
            FramePart q;
            for (size_t e = e0 + a; e < e0 + b; e++) {
                const size_t pb = (fj->cipher ? frame_entry_prefix_enc_bound(fj->names[e], fj->cipher->phsf) : frame_entry_prefix_bound(fj->names[e])) + meta_len(fj->meta, e);
                const uint64_t wb = pna_gpu_bound(algo, (size_t)src_len[e]);
                q.bound += pb; q.extra += (size_t)((wb + 64 + 16 * (src_len[e] >> 12)) / CHb); q.need += pb + wb + 16; q.over |= wb > CHb;
            }
            fp[t] = q;
        });
        size_t bound = 0, extra_chunks = 0;
        for (FramePart &q : fp) { const size_t b = q.bound; q.bound = bound; bound += b; extra_chunks += q.extra; layout_need += q.need; layout_over |= q.over; }
        if (c->h_desc.ensure(((e1 - e0) + extra_chunks + 1) * sizeof(FrameDesc)) || c->h_blob.ensure(bound + 8 * extra_chunks + 16) || c->h_segdst.ensure((size_t)(nseg + 1) * 8))
            return fail(c, PNA_E_NOMEM, "framing staging");
        fds = (FrameDesc *)c->h_desc.p; blob = (uint8_t *)c->h_blob.p; segdst = (uint64_t *)c->h_segdst.p;
        if (gcm) {
            // GCM STREAM: per entry a stream header, an HKDF stream key bound to its FHED chunk, round keys, hash subkey, E(K, J0);
            // a few host threads share the entries (SHA-256 / HKDF / key schedule: a few microseconds each) while k_lz runs
            gmat.resize(e1 - e0);
            uint8_t kc[32], ph[32];
            hkdf_sha256_32(fj->cipher->key, 32, nullptr, 0, "PNA-KC-v1", 9, kc);               // key_confirmation, aead.rs:161-163
            sha256_bytes(fj->cipher->phsf, strlen(fj->cipher->phsf), nullptr, 0, ph);
            const unsigned nt = (unsigned)std::min<size_t>(8, std::max<size_t>(1, (e1 - e0) / 256));
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t]() {
                    for (size_t e = e0 + t; e < e1; e += nt)
                        gcm_entry_material(fj->cipher, kc, ph, fj->ivs + 39 * e, gcm_seg, fj->names[e], algo, gmat[e - e0]);
                });
            for (auto &x : th) x.join();
        }
        if (!fj->cipher && !fj->meta) {
            // plain entries: the prefixes are written straight into the staging blob, every range of entries back to back from the range's bound offset
            // (the few unused bytes between two ranges travel with the blob; nothing refers to them)
            par_ranges(ne_all, host_nt, [&](unsigned t, size_t a, size_t b) {
                size_t at = fp[t].bound;
                for (size_t i = a; i < b; i++) {
                    const size_t pl = frame_entry_prefix_into(blob + at, fj->names[e0 + i], algo, src_len[e0 + i]);
                    fds[i] = FrameDesc{0, 0, (uint32_t)at, (uint32_t)pl, 0};
                    at += pl;
                }
            });
            blob_len = bound;
        } else
        for (size_t e = e0; e < e1; e++) {
            tmp.clear();
            if (gcm) frame_entry_prefix_enc(tmp, fj->names[e], algo, src_len[e], fj->cipher->encryption, PNA_MODE_GCM, fj->cipher->phsf, gmat[e - e0].header, 75);
            else if (fj->cipher) frame_entry_prefix_enc(tmp, fj->names[e], algo, src_len[e], fj->cipher->encryption, fj->cipher->cipher_mode, fj->cipher->phsf, fj->ivs + 16 * e, 16);
            else frame_entry_prefix(tmp, fj->names[e], algo, src_len[e], 0);
            splice_meta(tmp, fj->meta, e);
            memcpy(blob + blob_len, tmp.data(), tmp.size());
            fds[e - e0] = FrameDesc{0, 0, (uint32_t)blob_len, (uint32_t)tmp.size(), 0};
            blob_len += tmp.size();
        }
    }
    // Plain file entries of one FDAT chunk each (no cipher; the worst case of every payload below the chunk limit and of the whole sub-batch below the
    // destination's capacity): the archive layout is computed on the device (k_layout) and the host never waits in the middle of the sub-batch.
    mark("prefixes built");
    const bool dev_layout = fj && !solid && !fj->cipher && c->tun.dev_layout != 0 && !layout_over && out_base + layout_need + 16 <= dst_cap;
    if (dev_layout) {
        const size_t ne = e1 - e0;
        if (c->fr_desc.ensure(ne * sizeof(FrameDesc)) || c->fr_blob.ensure(blob_len + 16) || c->fr_segdst.ensure((size_t)(nseg + 1) * 8) ||
            c->fr_entoff.ensure((ne + 2 + 2 * (ne / 1024 + 2)) * 8) || c->h_entoff.ensure((ne + 2) * 8)) return fail(c, PNA_E_NOMEM, "framing workspace");
        int rcc = ensure_crc(c); if (rcc) return rcc;
        HIPCHK(c, hipMemcpyAsync(c->fr_desc.p, fds, ne * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->fr_blob.p, blob, blob_len, hipMemcpyHostToDevice, st));
        uint64_t *d_ent = (uint64_t *)c->fr_entoff.p;
        launch_layout((FrameDesc *)c->fr_desc.p, (uint8_t *)c->fr_blob.p, c->d_entry_seg, (const uint64_t *)c->seg_off.p, (uint32_t)ne, nseg, out_base,
                      (uint64_t *)c->fr_segdst.p, d_ent, d_ent + ne + 1, st);
        if (defl) launch_deflate_write(d_src, c->d_segs, c->d_blk_seg, nblk, (const BlkInfo *)c->blk.p, (const uint64_t *)c->fr_segdst.p, (const uint64_t *)c->seg_size.p,
                                       (const uint8_t *)c->litc.p, c->d_entry_seg, (uint32_t)ne, d_dst, st, c->call_stored, /* a wave per block */ max_len <= 32768 && nseg >= 4096);
        else launch_write(d_src, c->d_segs, nseg, c->d_blk_seg, nblk, (const BlkInfo *)c->blk.p, (const SegTables *)c->tabs.p, (const uint64_t *)c->fr_segdst.p,
                          (const uint8_t *)c->lits.p, (const uint8_t *)c->litc.p, (const uint8_t *)c->seqc.p, d_dst, any_empty, st, /* a wave per block */ max_len <= 32768 && nseg >= 4096);
        if (timed) HIPCHK(c, hipEventRecord(c->ev[6], st));
        launch_frame((const FrameDesc *)c->fr_desc.p, (uint32_t)ne, (const uint8_t *)c->fr_blob.p, (const CrcTabs *)c->crc_tabs.p,
                     d_dst, (uint64_t)dst_cap & ~(uint64_t)15, frame_fend_crc(), "FDAT", true, st, frame_max_payload);
        if (timed) HIPCHK(c, hipEventRecord(c->ev[7], st));
        // the entry offsets (when the caller wants them) and the sub-batch's length travel back behind the kernels: the call's one wait
        uint64_t *h_ent = (uint64_t *)c->h_entoff.p;
        if (fj->want_offsets) HIPCHK(c, hipMemcpyAsync(h_ent, d_ent, (ne + 2) * 8, hipMemcpyDeviceToHost, st));
        else HIPCHK(c, hipMemcpyAsync(h_ent + ne, d_ent + ne, 16, hipMemcpyDeviceToHost, st));
        mark("framing queued");
        HIPCHK(c, hipStreamSynchronize(st));
        mark("device done"); print_marks();
        HIPCHK(c, hipGetLastError());
        if (fj->want_offsets) memcpy(dst_off + e0, h_ent, ne * 8);
        dst_off[e1] = h_ent[ne];
        c->last_nblk = nblk;
        if (timed) {
            int rct = collect_timing(c, defl, nch, nseg, nblk, false);
            if (rct) return rct;
        }
        return PNA_OK;
    }
    // the output offsets are needed on the host before the write pass can be bounds-checked
    if (c->h_segoff.ensure((size_t)(nseg + 1) * 8)) return fail(c, PNA_E_NOMEM, "offset staging");
    const uint64_t *seg_off = (const uint64_t *)c->h_segoff.p;
    // (plain batches whose destination holds the worst case of every entry need no check against the sizes found: the write kernels go out
    // at once, the offsets travel behind them and the one wait is the call's last -- a wait in the middle of a small batch is a tenth of it)
    bool early_write = !fj;
    if (early_write) { uint64_t need = out_base; for (size_t e = e0; e < e1; e++) need += pna_gpu_bound(algo, (size_t)src_len[e]); early_write = need <= dst_cap; }
    if (!early_write) {
        HIPCHK(c, hipMemcpyAsync(c->h_segoff.p, c->seg_off.p, (size_t)(nseg + 1) * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    uint64_t total = early_write ? 0 : seg_off[nseg];
    std::vector<CipherUnit> cunits;
    const uint64_t *d_segdst = (const uint64_t *)c->seg_off.p;
    uint8_t *wbase = d_dst + out_base;
    if (fj) {
        // archive layout of this sub-batch: [prefix | payload | crc | FEND] per entry; the write kernels put every
        // segment straight at its final place, k_frame adds the rest (no second copy of the payload)
        uint64_t pos = out_base;
        if (solid) {
            // solid stream: one SDAT chunk per segment (= per zstd frame / per run of deflate blocks): [len "SDAT" | payload | crc]
            dst_off[e0] = pos;
            if (gcm) {
                // GCM STREAM over the solid stream (into_solid_archive takes any cipher, lib/src/archive/write.rs:443-470; GcmEncryptWriter, lib/src/cipher/
                // gcm.rs:48-100): the head carries the stream header as its first SDAT chunk; here one SDAT chunk per GCM segment, ciphertext || tag.  The
                // write kernels put the compressed stream down in one piece behind the first chunk header, segments k >= 1 then move forward by 28 k
                // bytes (tag and CRC of the chunk before + their own chunk header), as a GCM entry of several segments does.
                const uint64_t P = seg_off[nseg] - seg_off[0], G = gcm_seg;
                const uint64_t K = P ? (P + G - 1) / G : 1;
                if (K > 0xFFFFFFFFull) return fail(c, PNA_E_INVAL, "GCM segment counter overflow");
                const uint64_t B = pos + 8;
                for (uint32_t sg = 0; sg < nseg; sg++) segdst[sg] = B + (seg_off[sg] - seg_off[0]);
                if (K > 1) spread_copy.emplace_back(B, P);
                const GcmMaterial &gm = gmat[0];
                for (uint64_t k = 0; k < K; k++) {
                    const uint64_t sl = std::min<uint64_t>(G, P - k * G), hk = pos + k * (G + 28), so_ = hk + 8;
                    uint8_t *pf = blob + 8 * (size_t)k;
                    const uint64_t cl = sl + 16;
                    pf[0] = (uint8_t)(cl >> 24); pf[1] = (uint8_t)(cl >> 16); pf[2] = (uint8_t)(cl >> 8); pf[3] = (uint8_t)cl;
                    memcpy(pf + 4, "SDAT", 4);
                    fds[k] = FrameDesc{hk, (uint32_t)cl, (uint32_t)(8 * k), 8u, 0};
                    const uint32_t si = (uint32_t)gsegs.size();
                    uint8_t j0[16], eb[16];
                    memcpy(j0, gm.ctr_iv, 7);
                    j0[7] = (uint8_t)(k >> 24); j0[8] = (uint8_t)(k >> 16); j0[9] = (uint8_t)(k >> 8); j0[10] = (uint8_t)k; j0[11] = k + 1 == K ? 1 : 0;
                    j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
                    aes256_block_host(gm.rk, j0, eb);
                    GcmSeg gs; memcpy(gs.ctr_iv, j0, 16); gs.ctr_iv[15] = 2; gs.entry = 0;
                    gsegs.push_back(gs);
                    for (uint64_t o = 0; o < sl; o += CTR_UNIT) cunits.push_back(CipherUnit{so_ + o, o, (uint32_t)std::min<uint64_t>(CTR_UNIT, sl - o), si});
                    GcmEntry ge{so_, (uint32_t)sl, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
                    memcpy(ge.h, gm.h, 16);
                    for (int w = 0; w < 4; w++) ge.ej0[w] = ((uint32_t)eb[4 * w] << 24) | ((uint32_t)eb[4 * w + 1] << 16) | ((uint32_t)eb[4 * w + 2] << 8) | eb[4 * w + 3];
                    gents.push_back(ge);
                    if (k >= 1)
                        for (uint64_t o = 0; o < sl; o += (1u << 20))
                            spread.push_back(SpreadPiece{spread_bytes + k * G + o, so_ + o, (uint32_t)std::min<uint64_t>(1u << 20, sl - o)});
                }
                if (K > 1) spread_bytes += (P + 15) & ~(uint64_t)15;
                pos += P + 28 * K;
                nunit = (size_t)K; blob_len = 8 * (size_t)K;
            } else {
            for (uint32_t sg = 0; sg < nseg; sg++) {
                const uint64_t plen = seg_off[sg + 1] - seg_off[sg];
                uint8_t *pf = blob + 8 * (size_t)sg;
                pf[0] = (uint8_t)(plen >> 24); pf[1] = (uint8_t)(plen >> 16); pf[2] = (uint8_t)(plen >> 8); pf[3] = (uint8_t)plen;
                memcpy(pf + 4, "SDAT", 4);
                fds[sg] = FrameDesc{pos, (uint32_t)plen, 8u * sg, 8u, 0};
                segdst[sg] = pos + 8;
                if (fj->cipher)                                    // one cipher stream over all SDAT bodies: the keystream position runs on
                    for (uint64_t o = 0; o < plen; o += CTR_UNIT)
                        cunits.push_back(CipherUnit{pos + 8 + o, (seg_off[sg] - seg_off[0]) + o, (uint32_t)std::min<uint64_t>(CTR_UNIT, plen - o), 0u});
                pos += 8 + plen + 4;
            }
            blob_len = 8 * (size_t)nseg;
            }
        } else {
            // FlattenWriter cuts an entry's stream into FDAT chunks of max_chunk_size bytes, the last one holding the rest (lib/src/util/io.rs:60-77:
            // the open chunk is topped up before a new one starts; FileEntryBuilder::max_chunk_size, lib/src/entry/builder/file.rs:105-112; default
            // u32::MAX, lib/src/chunk.rs:28).  The write kernels put an entry's payload down in one piece behind the first FDAT header; for an entry of
            // K > 1 chunks the payload is saved to a scratch buffer and chunks 1 .. K - 1 move forward by 12 k bytes (CRC of the chunk before +
            // their own length / type), k_frame then takes one descriptor per chunk.  The same pass serves the GCM STREAM layout below.
            const uint64_t CH = chunk_limit(fj->max_chunk);
            std::vector<FrameDesc> units; units.reserve(e1 - e0);
            const bool cbc = fj->cipher && fj->cipher->cipher_mode == PNA_MODE_CBC;
            const bool ctr = fj->cipher && !cbc && !gcm;
            for (size_t e = e0; e < e1; e++) {
                const uint32_t s0 = entry_first_seg[e - e0], s1 = entry_first_seg[e - e0 + 1];
                const FrameDesc f0 = fds[e - e0];                  // prefix of the entry: FHED | fSIZ | ... | first FDAT header
                dst_off[e] = pos;
                const uint64_t p0 = pos + f0.prefix_len;           // where the payload starts
                uint64_t plen = seg_off[s1] - seg_off[s0];         // the entry's compressed stream, then what the cipher makes of it
                for (uint32_t sg = s0; sg < s1; sg++) segdst[sg] = p0 + (seg_off[sg] - seg_off[s0]);
                if (cbc) {
                    // CBC chains the whole entry (one lane) and appends the PKCS#7 padding block, in place
                    cunits.push_back(CipherUnit{p0, 0, (uint32_t)plen, (uint32_t)(e - e0)});
                    plen = (plen / 16 + 1) * 16;
                    if (plen > CH) return fail(c, PNA_E_UNSUPPORTED, "CBC entry beyond one FDAT chunk");
                } else if (gcm) {
                    // GCM STREAM (GcmEncryptWriter, lib/src/cipher/gcm.rs:48-100): the payload in segments of segment_size bytes, every
                    // segment followed by its 16-byte tag; all but the last carry nonce flag 0, the last one (possibly full, possibly
                    // empty) flag 1; counters 0, 1, ...  Segments k >= 1 move forward by 16 k bytes before the cipher runs.
                    const uint64_t K = plen ? (plen + gcm_seg - 1) / gcm_seg : 1;
                    if (K > 0xFFFFFFFFull) return fail(c, PNA_E_INVAL, "GCM segment counter overflow");
                    if (plen + 16 * K > CH) return fail(c, PNA_E_UNSUPPORTED, "GCM entry beyond one FDAT chunk");
                    const GcmMaterial &gm = gmat[e - e0];
                    if (K > 1) { spread_copy.emplace_back(p0, plen); }
                    for (uint64_t k = 0; k < K; k++) {
                        const uint64_t sl = std::min<uint64_t>(gcm_seg, plen - k * gcm_seg), so_ = p0 + k * ((uint64_t)gcm_seg + 16);
                        const uint32_t si = (uint32_t)gsegs.size();
                        uint8_t j0[16], eb[16];
                        memcpy(j0, gm.ctr_iv, 7);                                  // nonce prefix
                        j0[7] = (uint8_t)(k >> 24); j0[8] = (uint8_t)(k >> 16); j0[9] = (uint8_t)(k >> 8); j0[10] = (uint8_t)k; j0[11] = k + 1 == K ? 1 : 0;
                        j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
                        aes256_block_host(gm.rk, j0, eb);
                        GcmSeg gs; memcpy(gs.ctr_iv, j0, 16); gs.ctr_iv[15] = 2; gs.entry = (uint32_t)(e - e0);
                        gsegs.push_back(gs);
                        for (uint64_t o = 0; o < sl; o += CTR_UNIT)
                            cunits.push_back(CipherUnit{so_ + o, o, (uint32_t)std::min<uint64_t>(CTR_UNIT, sl - o), si});
                        GcmEntry ge{so_, (uint32_t)sl, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
                        memcpy(ge.h, gm.h, 16);
                        for (int w = 0; w < 4; w++) ge.ej0[w] = ((uint32_t)eb[4 * w] << 24) | ((uint32_t)eb[4 * w + 1] << 16) | ((uint32_t)eb[4 * w + 2] << 8) | eb[4 * w + 3];
                        gents.push_back(ge);
                        if (k >= 1)
                            for (uint64_t o = 0; o < sl; o += (1u << 20))
                                spread.push_back(SpreadPiece{spread_bytes + k * gcm_seg + o, so_ + o, (uint32_t)std::min<uint64_t>(1u << 20, sl - o)});
                    }
                    if (K > 1) spread_bytes += (plen + 15) & ~(uint64_t)15;
                    plen += 16 * K;
                }
                // the FDAT chunks: CH bytes each, the last one the rest (an empty payload is one empty chunk)
                const uint64_t K = plen ? (plen + CH - 1) / CH : 1;
                if (K > 1) spread_copy.emplace_back(p0, plen);
                for (uint64_t k = 0; k < K; k++) {
                    const uint64_t cl = std::min<uint64_t>(CH, plen - k * CH), cstart = p0 + k * (CH + 12);   // the chunk's data in the archive
                    FrameDesc u;
                    if (k == 0) { u = f0; u.arc_off = pos; }
                    else { u.prefix_off = (uint32_t)blob_len; u.prefix_len = 8; u.arc_off = cstart - 8; memcpy(blob + blob_len + 4, "FDAT", 4); blob_len += 8; }
                    uint8_t *lenf = &blob[u.prefix_off + u.prefix_len - 8];  // FDAT chunk length, big-endian
                    lenf[0] = (uint8_t)(cl >> 24); lenf[1] = (uint8_t)(cl >> 16); lenf[2] = (uint8_t)(cl >> 8); lenf[3] = (uint8_t)cl;
                    u.payload_len = (uint32_t)cl; u.pad = k + 1 < K ? 2u : 0u;
                    units.push_back(u);
                    if (k >= 1)
                        for (uint64_t o = 0; o < cl; o += (1u << 20))
                            spread.push_back(SpreadPiece{spread_bytes + k * CH + o, cstart + o, (uint32_t)std::min<uint64_t>(1u << 20, cl - o)});
                    if (ctr)                                       // CTR keeps the length and may be cut anywhere: the keystream position runs on over the chunks
                        for (uint64_t o = 0; o < cl; o += CTR_UNIT)
                            cunits.push_back(CipherUnit{cstart + o, k * CH + o, (uint32_t)std::min<uint64_t>(CTR_UNIT, cl - o), (uint32_t)(e - e0)});
                }
                if (K > 1) spread_bytes += (plen + 15) & ~(uint64_t)15;
                pos += f0.prefix_len + plen + 12 * (K - 1) + 4 + 12;
            }
            nunit = units.size();
            memcpy(fds, units.data(), nunit * sizeof(FrameDesc));
        }
        segdst[nseg] = pos;
        total = pos - out_base;
        if (pos + 16 > dst_cap) return fail(c, PNA_E_DSTSIZE, "device destination too small");
        if (c->fr_desc.ensure(nunit * sizeof(FrameDesc)) || c->fr_blob.ensure(blob_len + 16) || c->fr_segdst.ensure((size_t)(nseg + 1) * 8))
            return fail(c, PNA_E_NOMEM, "framing workspace");
        int rc = ensure_crc(c); if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->fr_desc.p, fds, nunit * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->fr_blob.p, blob, blob_len, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->fr_segdst.p, segdst, (size_t)(nseg + 1) * 8, hipMemcpyHostToDevice, st));
        d_segdst = (const uint64_t *)c->fr_segdst.p; wbase = d_dst;
    } else if (!early_write && out_base + total > dst_cap) return fail(c, PNA_E_DSTSIZE, "device destination too small");
    if (defl) launch_deflate_write(d_src, c->d_segs, c->d_blk_seg, nblk, (const BlkInfo *)c->blk.p,
                                   d_segdst, (const uint64_t *)c->seg_size.p, (const uint8_t *)c->litc.p, c->d_entry_seg,
                                   (uint32_t)(e1 - e0), wbase, st, c->call_stored, /* a wave per block */ max_len <= 32768 && nseg >= 4096);
    else launch_write(d_src, c->d_segs, nseg, c->d_blk_seg, nblk, (const BlkInfo *)c->blk.p,
                 (const SegTables *)c->tabs.p, d_segdst, (const uint8_t *)c->lits.p,
                 (const uint8_t *)c->litc.p, (const uint8_t *)c->seqc.p, wbase, any_empty, st, /* a wave per block */ max_len <= 32768 && nseg >= 4096);
    if (!spread.empty()) {
        // entries of several FDAT chunks / GCM segments: save the compact payloads, then put the pieces behind the first one at their places
        std::vector<PlaceDescH> pd(spread.size());
        for (size_t i = 0; i < spread.size(); i++) pd[i] = PlaceDescH{spread[i].src, spread[i].dst, spread[i].len, 0};
        if (c->ci_spread.ensure(spread_bytes + 64) || c->ci_spread_desc.ensure(pd.size() * sizeof(PlaceDescH) + 16)) return fail(c, PNA_E_NOMEM, "chunk workspace");
        uint64_t sp = 0;
        for (auto &cp : spread_copy) { HIPCHK(c, hipMemcpyAsync((uint8_t *)c->ci_spread.p + sp, d_dst + cp.first, cp.second, hipMemcpyDeviceToDevice, st)); sp += (cp.second + 15) & ~(uint64_t)15; }
        HIPCHK(c, hipMemcpyAsync(c->ci_spread_desc.p, pd.data(), pd.size() * sizeof(PlaceDescH), hipMemcpyHostToDevice, st));
        launch_gather(c->ci_spread_desc.p, (uint32_t)pd.size(), (const uint8_t *)c->ci_spread.p, d_dst, st);
        HIPCHK(c, hipStreamSynchronize(st));                      // pd goes out of scope
    }
    if (fj && fj->cipher) {
        if (solid && fj->cipher->cipher_mode == PNA_MODE_CBC) return fail(c, PNA_E_UNSUPPORTED, "solid archives: CTR and GCM on the device path (CBC encryption is one serial chain over the whole stream)");
        int rc = ensure_aes(c); if (rc) return rc;
        if (c->ci_units.ensure(cunits.size() * sizeof(CipherUnit) + 16) || c->ci_ivs.ensure((e1 - e0) * 16 + 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
        AesKey key; aes256_expand(fj->cipher->key, key);
        HIPCHK(c, hipMemcpyAsync(c->ci_units.p, cunits.data(), cunits.size() * sizeof(CipherUnit), hipMemcpyHostToDevice, st));
        std::vector<uint8_t> giv; std::vector<AesKey> gkeys;
        if (gcm) {
            giv.resize(gsegs.size() * 16); gkeys.resize(gsegs.size());
            for (size_t i = 0; i < gsegs.size(); i++) { memcpy(&giv[16 * i], gsegs[i].ctr_iv, 16); gkeys[i] = gmat[gsegs[i].entry].rk; }
            if (c->ci_keys.ensure(gkeys.size() * sizeof(AesKey) + 16) || c->ci_gcm.ensure(gents.size() * sizeof(GcmEntry) + 16) || c->ci_ivs.ensure(giv.size() + 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
            HIPCHK(c, hipMemcpyAsync(c->ci_ivs.p, giv.data(), giv.size(), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->ci_keys.p, gkeys.data(), gkeys.size() * sizeof(AesKey), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->ci_gcm.p, gents.data(), gents.size() * sizeof(GcmEntry), hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));                  // the host vectors above go out of scope
        } else HIPCHK(c, hipMemcpyAsync(c->ci_ivs.p, fj->ivs + 16 * e0, (e1 - e0) * 16, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipEventRecord(c->ev_ci[0], st));
        if (gcm) {
            launch_aes_ctr((const CipherUnit *)c->ci_units.p, (uint32_t)cunits.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, d_dst, key, (const AesKey *)c->ci_keys.p, st);
            launch_gcm_tag((const GcmEntry *)c->ci_gcm.p, (uint32_t)gents.size(), d_dst, st);
        } else if (fj->cipher->cipher_mode == PNA_MODE_CTR)
            launch_aes_ctr((const CipherUnit *)c->ci_units.p, (uint32_t)cunits.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, d_dst, key, nullptr, st);
        else
            launch_aes_cbc_enc((const CipherUnit *)c->ci_units.p, (uint32_t)cunits.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, d_dst, key, st);
        HIPCHK(c, hipEventRecord(c->ev_ci[1], st));
    }
    if (timed) HIPCHK(c, hipEventRecord(c->ev[6], st));
    if (fj) launch_frame((const FrameDesc *)c->fr_desc.p, (uint32_t)nunit, (const uint8_t *)c->fr_blob.p, (const CrcTabs *)c->crc_tabs.p,
                         d_dst, (uint64_t)dst_cap & ~(uint64_t)15, frame_fend_crc(), solid ? "SDAT" : "FDAT", !solid, st, frame_max_payload);
    if (timed) HIPCHK(c, hipEventRecord(c->ev[7], st));
    HIPCHK(c, hipGetLastError());
    if (early_write) {
        HIPCHK(c, hipMemcpyAsync(c->h_segoff.p, c->seg_off.p, (size_t)(nseg + 1) * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        total = seg_off[nseg];
    }
    if (!fj) for (size_t e = e0; e < e1; e++) dst_off[e] = out_base + seg_off[entry_first_seg[e - e0]];
    dst_off[e1] = out_base + total;
    c->last_nblk = nblk;
    if (fj && !timed) HIPCHK(c, hipStreamSynchronize(st));        // the staging buffers are reused by the next sub-batch
    if (timed) {
        HIPCHK(c, hipStreamSynchronize(st));
        int rct = collect_timing(c, defl, nch, nseg, nblk, fj && fj->cipher);
        if (rct) return rct;
    }
    return PNA_OK;
}

extern "C" int pna_gpu_compress_batch_device(pna_gpu_ctx *c, int algo, int level, size_t n, const void *d_src,
                                             const uint64_t *src_off, const uint64_t *src_len, void *d_dst, size_t dst_cap,
                                             uint64_t *dst_off, void *hip_stream) {
    if (!c || !src_off || !src_len || !dst_off || (!d_src && n) || (!d_dst && n)) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    set_call_level(c, algo, level);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->timing = pna_gpu_timing{};
    dst_off[0] = 0;
    uint64_t out_base = 0, in_total = 0;
    size_t e = 0;
    plan_call(c, src_len, n);
    const CallTotalScope call_total(c, src_len, n);                                              // (the block size follows the call, not its sub-batches)
    while (e < n) {
        size_t e1 = e; size_t blocks = 0;
        while (e1 < n) {
            size_t nb = plan_blocks(c, src_len[e1]);
            if (e1 > e && blocks + nb > c->max_blocks) break;
            blocks += nb; in_total += src_len[e1]; e1++;
        }
        int rc = run_subbatch(c, algo, (const uint8_t *)d_src, src_off, src_len, e, e1, (uint8_t *)d_dst, dst_cap, out_base, dst_off, st, true);
        if (rc) return rc;
        out_base = dst_off[e1];
        e = e1;
    }
    HIPCHK(c, hipStreamSynchronize(st));
    c->timing.in_bytes = in_total; c->timing.out_bytes = out_base;
    return PNA_OK;
}

extern "C" size_t pna_gpu_archive_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len) {
    size_t b = 28 + 12 + 64;                                    // signature + AHED, AEND, alignment slack of the CRC reads
    for (size_t i = 0; i < n; i++) {
        const size_t pb = pna_gpu_bound(algo, (size_t)src_len[i]);
        b += frame_entry_prefix_bound(names ? names[i] : nullptr) + pb + 16 + 12 * (src_len[i] >> 20);    // + one FDAT header / CRC per possible cut
    }
    return b;
}

// Non-solid `pna create` with the archive assembled in HBM: create_archive_file (cli/src/command/create.rs:575-635) +
// Archive::write_header / add_entry / finalize (lib/src/archive/write.rs) for file entries carrying FHED, fSIZ, FDAT, FEND.
extern "C" int pna_gpu_create_archive_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                             const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                             void *d_dst, size_t dst_cap, uint64_t *entry_off, uint64_t *archive_len,
                                             void *hip_stream) {
    return pna_gpu_create_archive_part_device(c, algo, level, n, names, d_src, src_off, src_len, d_dst, dst_cap, entry_off, archive_len,
                                              PNA_PART_HEAD | PNA_PART_TAIL, hip_stream);
}

// One shard of an archive whose entries are split over several producers (ranks): only the first shard carries the signature +
// AHED, only the last one AEND; the shards' outputs concatenated in entry order are the archive.
extern "C" int pna_gpu_create_archive_part_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                  const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                  void *d_dst, size_t dst_cap, uint64_t *entry_off, uint64_t *archive_len,
                                                  uint32_t part_flags, void *hip_stream) {
    return pna_gpu_create_archive_enc_device(c, algo, level, n, names, d_src, src_off, src_len, nullptr, d_dst, dst_cap, entry_off, archive_len,
                                             part_flags, hip_stream);
}

extern "C" size_t pna_gpu_archive_enc_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len, const pna_gpu_cipher *cipher) {
    size_t b = pna_gpu_archive_bound(algo, n, names, src_len);
    if (cipher && cipher->encryption != PNA_ENC_NONE && cipher->phsf) {
        b += n * (12 + strlen(cipher->phsf) + 12 + 75 + 16);   // PHSF, FDAT(iv | stream header), CBC padding / the final GCM tag
        if (cipher->cipher_mode == PNA_MODE_GCM) {               // one more tag per full stream segment of the (bounded) payload
            const uint64_t seg = cipher->gcm_segment_size ? cipher->gcm_segment_size : (1u << 20);
            for (size_t i = 0; i < n; i++) b += 16 * (pna_gpu_bound(algo, (size_t)src_len[i]) / seg);
        }
    }
    return b;
}

// The same with the cipher stage between the write kernels and the chunk CRC (get_writer: compress -> cipher -> sink,
// lib/src/entry/write.rs:268-274): entry record FHED | fSIZ | PHSF | FDAT(iv) | FDAT(ciphertext) | FEND.
extern "C" int pna_gpu_create_archive_enc_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                 const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                 const pna_gpu_cipher *cipher, void *d_dst, size_t dst_cap, uint64_t *entry_off,
                                                 uint64_t *archive_len, uint32_t part_flags, void *hip_stream) {
    return pna_gpu_create_archive_meta_device(c, algo, level, n, names, d_src, src_off, src_len, cipher, nullptr, d_dst, dst_cap, entry_off, archive_len,
                                              part_flags, hip_stream);
}

int check_meta(pna_gpu_ctx *c, const pna_gpu_entry_meta *meta, size_t n) {
    if (!meta) return PNA_OK;
    if ((meta->extra && !meta->extra_len) || (meta->facets && !meta->facets_len)) return fail(c, PNA_E_INVAL, "metadata blobs without lengths");
    for (size_t e = 0; e < n; e++) {
        if (meta->extra && meta->extra_len[e] && (!meta->extra[e] || !meta_blob_ok((const uint8_t *)meta->extra[e], meta->extra_len[e]))) return fail(c, PNA_E_INVAL, "extra chunks of an entry are not well-formed chunks");
        if (meta->facets && meta->facets_len[e] && (!meta->facets[e] || !meta_blob_ok((const uint8_t *)meta->facets[e], meta->facets_len[e]))) return fail(c, PNA_E_INVAL, "metadata chunks of an entry are not well-formed chunks");
    }
    return PNA_OK;
}

// ... and with per-entry metadata: chunks the host has already framed (timestamps, permissions, owner, xattr: try_for_each_metadata_facet,
// lib/src/entry.rs:124-180; user-defined extra chunks) are placed where NormalEntry::write_chunks_to puts them.
int create_archive_device_impl(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                      const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                      const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, uint32_t max_chunk, void *d_dst, size_t dst_cap,
                                      uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream);
extern "C" int pna_gpu_create_archive_meta_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                  const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                  const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, void *d_dst, size_t dst_cap,
                                                  uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream) {
    return create_archive_device_impl(c, algo, level, n, names, d_src, src_off, src_len, cipher, meta, c ? (uint32_t)c->tun.max_chunk_size : 0u, d_dst, dst_cap, entry_off, archive_len, part_flags, hip_stream);
}
extern "C" int pna_gpu_create_archive_chunked_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                     const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                     const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, uint32_t max_chunk_size, void *d_dst, size_t dst_cap,
                                                     uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream) {
    return create_archive_device_impl(c, algo, level, n, names, d_src, src_off, src_len, cipher, meta, max_chunk_size, d_dst, dst_cap, entry_off, archive_len, part_flags, hip_stream);
}
extern "C" size_t pna_gpu_archive_chunked_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len, const pna_gpu_cipher *cipher, uint32_t max_chunk_size) {
    size_t b = cipher && cipher->encryption != PNA_ENC_NONE ? pna_gpu_archive_enc_bound(algo, n, names, src_len, cipher) : pna_gpu_archive_bound(algo, n, names, src_len);
    const uint64_t CH = chunk_limit(max_chunk_size);
    for (size_t i = 0; i < n; i++) b += 12 * (size_t)((pna_gpu_bound(algo, (size_t)src_len[i]) + 64 + 16 * (src_len[i] >> 12)) / CH + 1);   // a CRC + a header per further chunk
    return b;
}
int create_archive_device_impl(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                      const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                      const pna_gpu_cipher *cipher, const pna_gpu_entry_meta *meta, uint32_t max_chunk, void *d_dst, size_t dst_cap,
                                      uint64_t *entry_off, uint64_t *archive_len, uint32_t part_flags, void *hip_stream) {
    { int rcm = check_meta(c, meta, n); if (rcm) return rcm; }
    if (!c || !archive_len || (n && (!names || !src_off || !src_len || !d_src)) || !d_dst) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    if ((uintptr_t)d_dst & 15) return fail(c, PNA_E_INVAL, "archive buffer must be 16-byte aligned");
    if (cipher && cipher->encryption == PNA_ENC_NONE) cipher = nullptr;
    const auto t_call0 = std::chrono::steady_clock::now();
    auto call_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call0).count(); };
    std::vector<uint8_t> own_ivs;
    const uint8_t *ivs = nullptr;
    if (cipher) { int rc = resolve_ivs(c, cipher, n, own_ivs, &ivs); if (rc) return rc; }
    set_call_level(c, algo, level);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->timing = pna_gpu_timing{};
    std::vector<uint8_t> head, tail;
    if (part_flags & PNA_PART_HEAD) frame_archive_head(head, 0);
    if (part_flags & PNA_PART_TAIL) frame_archive_tail(tail);
    if (head.size() + tail.size() + 16 > dst_cap) return fail(c, PNA_E_DSTSIZE, "device destination too small");
    if (!head.empty()) HIPCHK(c, hipMemcpyAsync(d_dst, head.data(), head.size(), hipMemcpyHostToDevice, st));
    // (10^6 entries: the call's own loops over them -- the total, the longest, the sub-batch cut -- ran one after the other on one thread, ~2 ms in front of the first
    // kernel and a zero-filled vector of offsets behind it; round 5: one pass on several threads, the cut by arithmetic where every entry is one block)
    std::unique_ptr<uint64_t[]> offs_mem(new uint64_t[n + 1]);
    uint64_t *offs = offs_mem.get();
    uint64_t pos = head.size(), in_total = 0, longest = 0;
    {
        const unsigned nt = host_loop_threads(n);
        std::vector<uint64_t> part_sum(nt, 0), part_max(nt, 0);
        par_ranges(n, nt, [&](unsigned t, size_t a, size_t b) { uint64_t sm = 0, mx = 0; for (size_t i = a; i < b; i++) { sm += src_len[i]; mx = std::max<uint64_t>(mx, src_len[i]); } part_sum[t] = sm; part_max[t] = mx; });
        for (unsigned t = 0; t < nt; t++) { in_total += part_sum[t]; longest = std::max(longest, part_max[t]); }
    }
    const CallTotalScope call_total(c, in_total);                                                // (the block size follows the call, not its sub-batches)
    FrameJob fj{names, 0, cipher, ivs, meta, max_chunk, entry_off != nullptr};
    size_t e = 0;
    plan_call_longest(c, longest);
    const bool one_block_each = longest <= ((uint64_t)1 << c->plan_log);
    while (e < n) {
        size_t e1 = e, blocks = 0;
        if (one_block_each) e1 = std::min(n, e + std::max<size_t>(1, c->max_blocks));           // (an empty entry counts as a block of its own here: the cut only has to respect the budget)
        else while (e1 < n) {
            size_t nb = plan_blocks(c, src_len[e1]);
            if (e1 > e && blocks + nb > c->max_blocks) break;
            blocks += nb; e1++;
        }
        if (c->tun.trace) fprintf(stderr, "[pna create_archive_device] sub-batch of %zu entries starts at %.2f ms of the call\n", e1 - e, call_ms());
        int rc = run_subbatch(c, algo, (const uint8_t *)d_src, src_off, src_len, e, e1, (uint8_t *)d_dst, dst_cap - tail.size(), pos, offs, st, true, &fj);
        if (rc) return rc;
        if (c->tun.trace) fprintf(stderr, "[pna create_archive_device] sub-batch back at %.2f ms\n", call_ms());
        pos = offs[e1];
        e = e1;
    }
    offs[n] = pos;
    if (!tail.empty()) HIPCHK(c, hipMemcpyAsync((uint8_t *)d_dst + pos, tail.data(), tail.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    pos += tail.size();
    if (entry_off) memcpy(entry_off, offs, (n + 1) * 8);
    *archive_len = pos;
    c->timing.in_bytes = in_total; c->timing.out_bytes = pos;
    if (c->tun.trace) fprintf(stderr, "[pna create_archive_device] call ends at %.2f ms\n", call_ms());
    return PNA_OK;
}

// The cipher stage alone over byte ranges of a device buffer (read side: DecryptReader::CtrAes, lib/src/entry/read.rs:83-88).
extern "C" int pna_gpu_cipher_apply_device(pna_gpu_ctx *c, const pna_gpu_cipher *cipher, int decrypt, size_t n, void *d_buf,
                                           const uint64_t *off, const uint64_t *len, void *hip_stream) {
    if (!c || !cipher || (n && (!d_buf || !off || !len || !cipher->ivs))) return fail(c, PNA_E_INVAL, "null argument");
    int rc = check_cipher(c, cipher); if (rc) return rc;
    const bool cbc = cipher->cipher_mode == PNA_MODE_CBC;
    if (cipher->cipher_mode == PNA_MODE_GCM) return fail(c, PNA_E_UNSUPPORTED, "GCM STREAM is offered by the archive entry points only");
    if (cbc && decrypt) return fail(c, PNA_E_UNSUPPORTED, "CBC decryption is not offered on the device path");
    if (n == 0) return PNA_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    rc = ensure_aes(c); if (rc) return rc;
    std::vector<CipherUnit> units;
    for (size_t i = 0; i < n; i++) {
        if (len[i] >= 0xFFFFFFF0ull) return fail(c, PNA_E_INVAL, "cipher range too long");
        if (cbc) units.push_back(CipherUnit{off[i], 0, (uint32_t)len[i], (uint32_t)i});
        else for (uint64_t o = 0; o < len[i]; o += CTR_UNIT) units.push_back(CipherUnit{off[i] + o, o, (uint32_t)std::min<uint64_t>(CTR_UNIT, len[i] - o), (uint32_t)i});
    }
    if (c->ci_units.ensure(units.size() * sizeof(CipherUnit) + 16) || c->ci_ivs.ensure(n * 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
    AesKey key; aes256_expand(cipher->key, key);
    c->timing = pna_gpu_timing{};
    HIPCHK(c, hipMemcpyAsync(c->ci_units.p, units.data(), units.size() * sizeof(CipherUnit), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->ci_ivs.p, cipher->ivs, n * 16, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev_ci[0], st));
    if (cbc) launch_aes_cbc_enc((const CipherUnit *)c->ci_units.p, (uint32_t)units.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, (uint8_t *)d_buf, key, st);
    else launch_aes_ctr((const CipherUnit *)c->ci_units.p, (uint32_t)units.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, (uint8_t *)d_buf, key, nullptr, st);
    HIPCHK(c, hipEventRecord(c->ev_ci[1], st));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    float mc = 0; (void)hipEventElapsedTime(&mc, c->ev_ci[0], c->ev_ci[1]);
    c->timing.ms_cipher = mc;
    for (size_t i = 0; i < n; i++) c->timing.in_bytes += len[i];
    return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// `pna create --solid` with the archive assembled in HBM (create_archive_file's solid branch, cli/src/command/create.rs:594-598,
// 603-617; SolidArchive / SolidEntryBuilder, lib/src/archive/write.rs:443-470,575-580,716-727):
//   1. the inner entries are serialised as STORE records FHED | fSIZ | FDAT | FEND (their FDAT CRC-32 computed by k_frame) into
//      one stream in HBM,
//   2. that stream is compressed as ONE entry (independent 1 MiB frames / zlib blocks inside the kernels),
//   3. every segment's output becomes one SDAT chunk between SHED and SEND.

extern "C" size_t pna_gpu_solid_archive_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len) {
    uint64_t plain = 0;
    for (size_t i = 0; i < n; i++) plain += frame_entry_prefix_bound(names ? names[i] : nullptr) + src_len[i] + 16;
    const uint64_t segs = (plain + SEG_SIZE - 1) / SEG_SIZE + 1;
    return 28 + 17 + pna_gpu_bound(algo, (size_t)plain) + 12 * segs + 12 + 12 + 64;
}

extern "C" size_t pna_gpu_solid_archive_enc_bound(int algo, size_t n, const char *const *names, const uint64_t *src_len, const pna_gpu_cipher *cipher) {
    size_t b = pna_gpu_solid_archive_bound(algo, n, names, src_len);
    if (!cipher || cipher->encryption == PNA_ENC_NONE) return b;
    b += 12 + (cipher->phsf ? strlen(cipher->phsf) : 0) + 12 + 75 + 64;          // PHSF chunk, the chunk of the IV / stream header
    if (cipher->cipher_mode == PNA_MODE_GCM) {
        const uint64_t seg = cipher->gcm_segment_size ? cipher->gcm_segment_size : (1u << 20);
        b += 28 * (size_t)(b / seg + 2);                                            // tag + chunk framing per GCM segment
    }
    return b;
}

extern "C" int pna_gpu_create_solid_archive_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                   const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                   void *d_dst, size_t dst_cap, uint64_t *archive_len, void *hip_stream) {
    return pna_gpu_create_solid_archive_enc_device(c, algo, level, n, names, d_src, src_off, src_len, nullptr, d_dst, dst_cap, archive_len, hip_stream);
}

// With a cipher (CTR): SHED(encryption, cipher_mode) | PHSF | SDAT(iv) | SDAT(ciphertext)* | SEND -- into_solid_archive writes the PHSF
// chunk behind SHED and the IV as the first write of the SDAT stream (lib/src/archive/write.rs:443-470); one cipher stream runs over
// all SDAT bodies.  cipher->ivs: ONE 16-byte IV (or NULL).
extern "C" int pna_gpu_create_solid_archive_enc_device(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                       const void *d_src, const uint64_t *src_off, const uint64_t *src_len,
                                                       const pna_gpu_cipher *cipher, void *d_dst, size_t dst_cap, uint64_t *archive_len,
                                                       void *hip_stream) {
    if (cipher && cipher->encryption == PNA_ENC_NONE) cipher = nullptr;
    std::vector<uint8_t> own_ivs;
    const uint8_t *ivs = nullptr;
    if (cipher) { int rc0 = resolve_ivs(c, cipher, 1, own_ivs, &ivs); if (rc0) return rc0; }
    if (cipher && cipher->cipher_mode == PNA_MODE_CBC) return fail(c, PNA_E_UNSUPPORTED, "solid archives: CTR and GCM on the device path (CBC encryption is one serial chain over the whole stream)");
    if (!c || !archive_len || (n && (!names || !src_off || !src_len || !d_src)) || !d_dst) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    if ((uintptr_t)d_dst & 15) return fail(c, PNA_E_INVAL, "archive buffer must be 16-byte aligned");
    set_call_level(c, algo, level);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    // ---- 1. layout of the serialised inner entries (all sizes are known up front)
    std::vector<FrameDesc> fds(n); std::vector<uint8_t> blob; std::vector<PlaceDescH> places;
    uint64_t pos = 0, max_inner = 0;
    for (size_t i = 0; i < n; i++) max_inner = std::max<uint64_t>(max_inner, src_len[i]);
    const uint32_t solid_max_inner = max_inner <= 16380 ? (uint32_t)std::max<uint64_t>(max_inner, 1) : 0u;      // (stored inner entries: the FDAT payload is the entry; small ones take k_frame's wave-per-entry form)
    for (size_t i = 0; i < n; i++) {
        if (src_off[i] & 15) return fail(c, PNA_E_INVAL, "entry offset not 16-byte aligned");
        if (src_len[i] >= 0x7FFF0000ull) return fail(c, PNA_E_INVAL, "inner entry too large for one FDAT chunk");
        const size_t po = blob.size();
        if (src_len[i] == 0) {
            frame_inner_entry_empty(blob, names[i]);
            fds[i] = FrameDesc{pos, 0, (uint32_t)po, (uint32_t)(blob.size() - po), 1};
            pos += blob.size() - po;
            continue;
        }
        frame_entry_prefix(blob, names[i], PNA_ALGO_STORE, src_len[i], (uint32_t)src_len[i]);
        const uint32_t pl = (uint32_t)(blob.size() - po);
        fds[i] = FrameDesc{pos, (uint32_t)src_len[i], (uint32_t)po, pl, 0};
        for (uint64_t k = 0; k < src_len[i]; k += SEG_SIZE)
            places.push_back(PlaceDescH{src_off[i] + k, pos + pl + k, (uint32_t)std::min<uint64_t>(SEG_SIZE, src_len[i] - k), 0});
        pos += pl + src_len[i] + 16;
    }
    const uint64_t plain_len = pos;
    if (c->solid_plain.ensure(plain_len + 8192) || c->solid_desc.ensure(n * sizeof(FrameDesc) + 16) || c->solid_blob.ensure(blob.size() + 16) ||
        c->solid_place.ensure(places.size() * sizeof(PlaceDescH) + 16)) return fail(c, PNA_E_NOMEM, "solid workspace");
    int rc = ensure_crc(c); if (rc) return rc;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->solid_desc.p, fds.data(), n * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->solid_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
        if (!places.empty()) HIPCHK(c, hipMemcpyAsync(c->solid_place.p, places.data(), places.size() * sizeof(PlaceDescH), hipMemcpyHostToDevice, st));
        launch_place(c->solid_place.p, (uint32_t)places.size(), (const uint8_t *)d_src, (uint8_t *)c->solid_plain.p, st);
        launch_frame((const FrameDesc *)c->solid_desc.p, (uint32_t)n, (const uint8_t *)c->solid_blob.p, (const CrcTabs *)c->crc_tabs.p,
                     (uint8_t *)c->solid_plain.p, (uint64_t)c->solid_plain.cap & ~(uint64_t)15, frame_fend_crc(), "FDAT", true, st, solid_max_inner);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));                     // the host vectors above are read by the async copies
    }
    // ---- 2 + 3. one entry -> SDAT chunks, between the fixed chunks
    c->timing = pna_gpu_timing{};
    std::vector<uint8_t> head, tail;
    frame_archive_head(head, 0);
    if (cipher && cipher->cipher_mode == PNA_MODE_GCM) {
        // the stream header (salt || nonce prefix || segment size || key confirmation: 75 bytes) is the stream's first write, a chunk of its own
        GcmMaterial gm; uint8_t kc[32], ph[32];
        hkdf_sha256_32(cipher->key, 32, nullptr, 0, "PNA-KC-v1", 9, kc);
        sha256_bytes(cipher->phsf, strlen(cipher->phsf), nullptr, 0, ph);
        gcm_entry_material(cipher, kc, ph, ivs, cipher->gcm_segment_size ? cipher->gcm_segment_size : (1u << 20), nullptr, algo, gm);
        frame_solid_head_enc(head, algo, cipher->encryption, cipher->cipher_mode, cipher->phsf, gm.header, 75);
    } else if (cipher) frame_solid_head_enc(head, algo, cipher->encryption, cipher->cipher_mode, cipher->phsf, ivs, 16); else frame_solid_head(head, algo);
    frame_solid_tail(tail); frame_archive_tail(tail);
    if (head.size() + tail.size() + 64 > dst_cap) return fail(c, PNA_E_DSTSIZE, "device destination too small");
    HIPCHK(c, hipMemcpyAsync(d_dst, head.data(), head.size(), hipMemcpyHostToDevice, st));
    const uint64_t off0 = 0, len0 = plain_len; uint64_t offs[2] = {0, 0};
    FrameJob fj{nullptr, 1, cipher, ivs};
    rc = run_subbatch(c, algo, (const uint8_t *)c->solid_plain.p, &off0, &len0, 0, 1, (uint8_t *)d_dst, dst_cap - tail.size(), head.size(), offs, st, true, &fj);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync((uint8_t *)d_dst + offs[1], tail.data(), tail.size(), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipStreamSynchronize(st));
    *archive_len = offs[1] + tail.size();
    uint64_t in_total = 0; for (size_t i = 0; i < n; i++) in_total += src_len[i];
    c->timing.in_bytes = in_total; c->timing.out_bytes = *archive_len;
    return PNA_OK;
}

extern "C" int pna_gpu_debug_block(pna_gpu_ctx *c, uint32_t block, uint64_t *seqs, uint32_t cap_seqs, uint32_t *nseq,
                                   uint8_t *lits, uint32_t cap_lits, uint32_t *nlit) {
    if (!c || block >= c->last_nblk) return PNA_E_INVAL;
    HIPCHK(c, hipSetDevice(c->device));
    BlkInfo bi;
    HIPCHK(c, hipMemcpy(&bi, (BlkInfo *)c->blk.p + block, sizeof(bi), hipMemcpyDeviceToHost));
    if (nseq) *nseq = bi.nseq;
    if (nlit) *nlit = bi.nlit;
    if (seqs) HIPCHK(c, hipMemcpy(seqs, (uint64_t *)c->seqs.p + (size_t)block * seq_cap_of(c->last_blk_log), (size_t)std::min(cap_seqs, bi.nseq) * 8, hipMemcpyDeviceToHost));
    if (lits) HIPCHK(c, hipMemcpy(lits, (uint8_t *)c->lits.p + ((size_t)block << c->last_blk_log), std::min(cap_lits, bi.nlit), hipMemcpyDeviceToHost));
    return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// corpus tables (integer-only; same construction as the checker's corpus model, written independently here)
static uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int ensure_corpus(pna_gpu_ctx *c) {
    if (c->corpus_ready) return PNA_OK;
    const int VOCAB = 50000, SLOT = 16, NPHRASE = 8192;
    static const uint16_t LCUM[26] = {817, 966, 1244, 1669, 2939, 3162, 3364, 3973, 4670, 4685, 4762, 5165, 5406,
                                      6081, 6832, 7025, 7035, 7634, 8267, 9173, 9449, 9547, 9783, 9798, 9995, 10000};
    std::vector<uint8_t> vocab((size_t)VOCAB * SLOT, 0); std::vector<uint64_t> cum(VOCAB); std::vector<uint32_t> phr((size_t)NPHRASE * 4, 0);
    uint64_t s = 0x504E41ull;
    for (int w = 0; w < VOCAB; w++) {
        uint64_t r = splitmix(s);
        int len = 2 + (int)((r & 0xFFFF) * 11 >> 16);
        if (w < 64) len = 2 + (int)((r & 0xFFFF) * 3 >> 16);
        else if (w < 1024) len = 3 + (int)((r & 0xFFFF) * 5 >> 16);
        uint8_t *slot = &vocab[(size_t)w * SLOT];
        slot[0] = (uint8_t)len;
        for (int i = 0; i < len; i++) { uint32_t x = (uint32_t)(splitmix(s) >> 33) % 10000u; int ch = 0; while (LCUM[ch] <= x) ch++; slot[1 + i] = (uint8_t)('a' + ch); }
    }
    uint64_t acc = 0;
    for (int k = 0; k < VOCAB; k++) { acc += (1ull << 40) / (uint64_t)(k + 1); cum[k] = acc; }
    auto draw = [&](uint64_t r, int n) {
        unsigned __int128 m = (unsigned __int128)r * cum[n - 1]; uint64_t x = (uint64_t)(m >> 64);
        int lo = 0, hi = n - 1; while (lo < hi) { int mid = (lo + hi) >> 1; if (cum[mid] > x) hi = mid; else lo = mid + 1; } return lo;
    };
    for (int p = 0; p < NPHRASE; p++) {
        uint64_t r = splitmix(s);
        phr[4 * p] = (uint32_t)(2 + (int)(r & 1));
        for (int i = 0; i < 3; i++) phr[4 * p + 1 + i] = (uint32_t)draw(splitmix(s), VOCAB);
    }
    if (c->c_vocab.ensure(vocab.size()) || c->c_cum.ensure(cum.size() * 8) || c->c_phr.ensure(phr.size() * 4)) return fail(c, PNA_E_NOMEM, "corpus tables");
    HIPCHK(c, hipMemcpy(c->c_vocab.p, vocab.data(), vocab.size(), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->c_cum.p, cum.data(), cum.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->c_phr.p, phr.data(), phr.size() * 4, hipMemcpyHostToDevice));
    c->corpus_ready = true;
    return PNA_OK;
}

extern "C" int pna_bench_corpus_fill_device(pna_gpu_ctx *c, int kind, uint64_t first_file, uint64_t n_files, uint64_t file_len,
                                            uint64_t stride, void *d_dst, void *hip_stream) {
    if (!c || !d_dst || kind < 0 || kind > 4 || stride < file_len) return fail(c, PNA_E_INVAL, "bad corpus argument");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_corpus(c); if (rc) return rc;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    launch_corpus(kind, first_file, n_files, file_len, stride, (const uint8_t *)c->c_vocab.p, (const uint64_t *)c->c_cum.p,
                  (const uint32_t *)c->c_phr.p, (uint8_t *)d_dst, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    return PNA_OK;
}

extern "C" int pna_gpu_debug_lz_stamps(pna_gpu_ctx *c, unsigned long long *out8) {
    if (!c || !out8) return PNA_E_INVAL;
    HIPCHK(c, hipSetDevice(c->device));
    lz_read_stamps(out8);
    return PNA_OK;
}

