// pna_pipeline.cpp -- the host-memory create pipelines of libpna_gpu.so: bounded staging || H2D || kernels || D2H (pna_gpu_create_archive_host and its
// forms, solid, multi-context, append, zero-staging host slots) and pna_gpu_compress_batch from host buffers.
#include "pna_ctx.h"
#include <memory>
#include <atomic>
static void parallel_stage(uint8_t *dst, const void *const *src, const size_t *src_len, const uint64_t *off, size_t e0, size_t e1, unsigned threads);

// `pna create --solid` from host memory (SolidArchive::add_entry streams the entries into ONE encoder, lib/src/archive/write.rs:575-580).
// zstd: the serialised inner entries -- FHED | fSIZ | FDAT(data, crc) | FEND per entry, stored -- reach the device in WINDOWS of solid_win_mib MiB of that
// stream (whole 1 MiB segments: every segment is a frame and an SDAT chunk of its own, so a window is compressed like the whole), through two page-locked
// slots each way: ~4 windows of page-locked memory whatever the archive's size (round 4: the whole stream was staged, copied and held at once).  The host
// writes the chunk framing and the entries' bytes where they stand in the stream while the device compresses the window before; the data chunks' CRC-32 are
// the device's -- the raw CRC register of every piece of a chunk inside the window (k_frame's piece mode), chained by the host with CRC(A || B) =
// x^(8|B|) CRC(A) + CRC(B) for a chunk that spans windows, written into the window (k_crc_patch) before it is compressed.  An inner entry is cut into
// FDAT chunks of at most 2^32 - 5 bytes like FlattenWriter's (lib/src/util/io.rs:60-77): any size goes through (the one-shot device path: below 2 GiB).
// The archive equals pna_gpu_create_solid_archive_device's byte for byte.  deflate (one zlib stream with one Adler-32 over everything) and the
// single_frame option keep the one-shot form below.
namespace {
struct SolidSpan { uint64_t pos, len; uint32_t kind; uint32_t chunk; uint64_t a, b; };     // kind 0: blob[a ..), 1: entry a from byte b on, 2: the CRC of chunk `chunk`
struct SolidChunk { uint32_t state = 0, crc = 0; uint64_t end = 0; bool started = false, done = false; };   // end: stream position behind the chunk's data
}
static int solid_stream_zstd(pna_gpu_ctx *c, int level, size_t n, const char *const *names, const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user) {
    const int algo = PNA_ALGO_ZSTD;
    set_call_level(c, algo, level);
    hipStream_t st = c->stream;
    // ---- the stream's layout
    std::vector<uint8_t> blob; std::vector<SolidSpan> spans; std::vector<SolidChunk> chunks;
    const uint64_t CH = 0xFFFFFFFBull;
    uint64_t pos = 0;
    uint8_t fend[12] = {0, 0, 0, 0, 'F', 'E', 'N', 'D', 0, 0, 0, 0};
    { const uint32_t fc = frame_fend_crc(); fend[8] = (uint8_t)(fc >> 24); fend[9] = (uint8_t)(fc >> 16); fend[10] = (uint8_t)(fc >> 8); fend[11] = (uint8_t)fc; }
    auto lit = [&](size_t from) { const uint64_t l = blob.size() - from; spans.push_back(SolidSpan{pos, l, 0u, 0u, (uint64_t)from, 0}); pos += l; };
    for (size_t i = 0; i < n; i++) {
        const size_t b0 = blob.size();
        if (src_len[i] == 0) { frame_inner_entry_empty(blob, names[i]); lit(b0); continue; }
        const uint64_t L = src_len[i];
        for (uint64_t o = 0; o < L; o += CH) {
            const uint64_t cl = std::min<uint64_t>(CH, L - o);
            const size_t b1 = blob.size();
            if (o == 0) frame_entry_prefix(blob, names[i], PNA_ALGO_STORE, L, (uint32_t)cl);
            else { const uint8_t h[8] = {(uint8_t)(cl >> 24), (uint8_t)(cl >> 16), (uint8_t)(cl >> 8), (uint8_t)cl, 'F', 'D', 'A', 'T'}; blob.insert(blob.end(), h, h + 8); }
            lit(b1);
            const uint32_t ci = (uint32_t)chunks.size();
            chunks.emplace_back();
            spans.push_back(SolidSpan{pos, cl, 1u, ci, (uint64_t)i, o}); pos += cl;
            chunks[ci].end = pos;
            spans.push_back(SolidSpan{pos, 4, 2u, ci, 0, 0}); pos += 4;
            if (chunks.size() > 0x7FFFFFF0u) return fail(c, PNA_E_INVAL, "too many data chunks");
        }
        const size_t b2 = blob.size();
        blob.insert(blob.end(), fend, fend + 12); lit(b2);
    }
    const uint64_t plain_len = pos;
    // ---- the fixed chunks around the stream
    std::vector<uint8_t> head, tail;
    frame_archive_head(head, 0); frame_solid_head(head, algo);
    frame_solid_tail(tail); frame_archive_tail(tail);
    if (sink(user, head.data(), head.size()) != 0) return fail(c, PNA_E_SINK, "sink failed");
    const uint64_t W = std::max<uint64_t>(1, (uint64_t)c->tun.solid_win_mib) << 20;           // a multiple of SEG_SIZE
    const uint64_t nwin = (plain_len + W - 1) / W;
    const uint64_t wcap_in = std::min<uint64_t>(W, plain_len) + 8192;
    const uint64_t wcap_out = pna_gpu_bound(algo, (size_t)std::min<uint64_t>(W, plain_len)) + (std::min<uint64_t>(W, plain_len) / SEG_SIZE + 2) * 16 + 4096;
    int rc = ensure_crc(c); if (rc) return rc;
    if (c->stage_in.ensure(2 * wcap_in + 64) || c->stage_out.ensure(2 * wcap_out + 64) || c->hp_in[0].ensure(wcap_in) || c->hp_in[1].ensure(wcap_in) ||
        c->hp_out[0].ensure(wcap_out) || c->hp_out[1].ensure(wcap_out)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    const CallTotalScope call_total(c, plain_len);                                            // every window picks the block size of the whole stream
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = c->tun.stage_threads > 0 ? (unsigned)c->tun.stage_threads : std::min(8u, std::max(1u, hw / 2));
    // assemble window k in its page-locked slot: framing bytes, entry bytes, the CRCs that are known; the pieces of data chunks inside it
    struct Win { std::vector<FrameDesc> pieces; std::vector<uint32_t> piece_chunk; std::vector<uint64_t> piece_len; std::vector<uint32_t> crc_spans; uint64_t w0 = 0, w1 = 0; };
    size_t cursor = 0;                                                                         // first span that reaches into the window
    auto assemble = [&](uint64_t k, Win &w) {
        w.pieces.clear(); w.piece_chunk.clear(); w.piece_len.clear(); w.crc_spans.clear();
        w.w0 = k * W; w.w1 = std::min(plain_len, w.w0 + W);
        uint8_t *slot = (uint8_t *)c->hp_in[k & 1].p;
        struct Job { uint8_t *d; const uint8_t *s; uint64_t l; };
        std::vector<Job> jobs;
        while (cursor < spans.size() && spans[cursor].pos + spans[cursor].len <= w.w0) cursor++;
        for (size_t j = cursor; j < spans.size() && spans[j].pos < w.w1; j++) {
            const SolidSpan &sp = spans[j];
            const uint64_t a = std::max(sp.pos, w.w0), b = std::min(sp.pos + sp.len, w.w1);
            if (a >= b) continue;
            uint8_t *d = slot + (a - w.w0);
            if (sp.kind == 0) memcpy(d, blob.data() + sp.a + (a - sp.pos), b - a);
            else if (sp.kind == 1) {
                const uint8_t *sbase = (const uint8_t *)src[sp.a] + sp.b + (a - sp.pos);
                for (uint64_t o = 0; o < b - a; o += (8u << 20)) jobs.push_back(Job{d + o, sbase + o, std::min<uint64_t>(8u << 20, b - a - o)});
                w.pieces.push_back(FrameDesc{a - w.w0, (uint32_t)(b - a), 0u, 0u, 4u | (a == sp.pos ? 0u : 8u)});
                w.piece_chunk.push_back(sp.chunk); w.piece_len.push_back(b - a);
            } else {
                const SolidChunk &cc = chunks[sp.chunk];
                if (cc.done) for (uint64_t q = a; q < b; q++) slot[q - w.w0] = (uint8_t)(cc.crc >> (24 - 8 * (q - sp.pos)));
                else { memset(d, 0, b - a); w.crc_spans.push_back((uint32_t)j); }
            }
        }
        if (w.w1 - w.w0 < wcap_in) memset(slot + (w.w1 - w.w0), 0, std::min<uint64_t>(64, wcap_in - (w.w1 - w.w0)));
        if (jobs.size() <= 1 || T <= 1) { for (const Job &jb : jobs) memcpy(jb.d, jb.s, jb.l); }
        else {
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            for (unsigned t = 0; t < std::min<size_t>(T, jobs.size()); t++)
                th.emplace_back([&]() { for (size_t q; (q = next.fetch_add(1)) < jobs.size();) memcpy(jobs[q].d, jobs[q].s, jobs[q].l); });
            for (auto &x : th) x.join();
        }
    };
    Win win[2];
    if (nwin) assemble(0, win[0]);
    DevBuf &dp = c->solid_place, &dd = c->solid_desc;                                          // patches, piece descriptors + states
    for (uint64_t k = 0; k < nwin && rc == PNA_OK; k++) {
        Win &w = win[k & 1];
        const uint64_t wl = w.w1 - w.w0;
        uint8_t *d_in = (uint8_t *)c->stage_in.p + (k & 1) * ((wcap_in + 255) & ~(uint64_t)255);
        uint8_t *d_out = (uint8_t *)c->stage_out.p + (k & 1) * ((wcap_out + 255) & ~(uint64_t)255);
        HIPCHK(c, hipMemcpyAsync(d_in, c->hp_in[k & 1].p, wl + std::min<uint64_t>(64, wcap_in - wl), hipMemcpyHostToDevice, st));
        // the CRC registers of the window's pieces
        const size_t np = w.pieces.size();
        std::vector<uint32_t> states(np);
        if (np) {
            if (dd.ensure(np * (sizeof(FrameDesc) + 4) + 64)) return fail(c, PNA_E_NOMEM, "solid workspace");
            uint32_t *d_states = (uint32_t *)((uint8_t *)dd.p + np * sizeof(FrameDesc));
            HIPCHK(c, hipMemcpyAsync(dd.p, w.pieces.data(), np * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
            launch_frame_pieces((const FrameDesc *)dd.p, (uint32_t)np, (const CrcTabs *)c->crc_tabs.p, d_in, wcap_in & ~(uint64_t)15, "FDAT", d_states, st);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(states.data(), d_states, np * 4, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(c, hipStreamSynchronize(st));
        for (size_t q = 0; q < np; q++) {
            SolidChunk &cc = chunks[w.piece_chunk[q]];
            cc.state = cc.started ? (crc_gf2_mulmod(crc_gf2_xpow(8 * w.piece_len[q]), cc.state) ^ states[q]) : states[q];
            cc.started = true;
            if (w.w0 + w.pieces[q].arc_off + w.piece_len[q] == cc.end) { cc.crc = ~cc.state; cc.done = true; }
        }
        std::vector<CrcPatchH> patches;
        for (uint32_t j : w.crc_spans) {
            const SolidSpan &sp = spans[j];
            const SolidChunk &cc = chunks[sp.chunk];
            if (!cc.done) return fail(c, PNA_E_INVAL, "internal: a chunk's CRC is due before its data is through");
            uint32_t mask = 0;
            for (int q = 0; q < 4; q++) if (sp.pos + q >= w.w0 && sp.pos + q < w.w1) mask |= 1u << q;
            patches.push_back(CrcPatchH{(int64_t)sp.pos - (int64_t)w.w0, cc.crc, mask});
        }
        if (!patches.empty()) {
            if (dp.ensure(patches.size() * sizeof(CrcPatchH) + 64)) return fail(c, PNA_E_NOMEM, "solid workspace");
            HIPCHK(c, hipMemcpyAsync(dp.p, patches.data(), patches.size() * sizeof(CrcPatchH), hipMemcpyHostToDevice, st));
            launch_crc_patch(dp.p, (uint32_t)patches.size(), d_in, st);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(st));                                               // (`patches` is read by the copy)
        }
        // the next window is assembled on a thread of its own while the device compresses this one
        std::thread next;
        if (k + 1 < nwin) next = std::thread([&, k]() { assemble(k + 1, win[(k + 1) & 1]); });
        const uint64_t off0 = 0, len0 = wl; uint64_t offs[2] = {0, 0};
        FrameJob fj{nullptr, 1, nullptr, nullptr};
        rc = run_subbatch(c, algo, d_in, &off0, &len0, 0, 1, d_out, wcap_out, 0, offs, st, false, &fj);
        if (rc == PNA_OK && hipMemcpyAsync(c->hp_out[k & 1].p, d_out, offs[1], hipMemcpyDeviceToHost, st) != hipSuccess) rc = fail(c, PNA_E_HIP, "D2H copy failed");
        if (rc == PNA_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(c, PNA_E_HIP, "D2H copy failed");
        if (next.joinable()) next.join();
        if (rc == PNA_OK && offs[1] && sink(user, c->hp_out[k & 1].p, (size_t)offs[1]) != 0) rc = fail(c, PNA_E_SINK, "sink failed");
    }
    if (rc) return rc;
    if (sink(user, tail.data(), tail.size()) != 0) return fail(c, PNA_E_SINK, "sink failed");
    return PNA_OK;
}

// Page-locked staging memory the context holds right now (the slots of the host pipelines; plan blobs and tables excluded): what tests/ pin the
// bounded-memory claims with.  Not a product path.
extern "C" uint64_t pna_gpu_debug_pinned_bytes(pna_gpu_ctx *c) {
    if (!c) return 0;
    uint64_t t = 0;
    for (auto &b : c->hp_in) t += b.cap;
    for (auto &b : c->hp_out) t += b.cap;
    return t;
}

// deflate / single_frame: one H2D of the entries, the device path, one D2H of the archive, handed to the sink in pieces of at most 16 MiB (the whole
// stream is in flight at once: a zlib stream is one compression unit with one Adler-32)
extern "C" int pna_gpu_create_solid_archive_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                 const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user) {
    if (!c || !sink || (n && (!names || !src || !src_len))) return fail(c, PNA_E_INVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (algo == PNA_ALGO_ZSTD && n && !c->tun.single_frame && c->tun.solid_win_mib > 0) return solid_stream_zstd(c, level, n, names, src, src_len, sink, user);
    std::vector<uint64_t> off(n + 1), len(n);
    uint64_t pos = 0;
    for (size_t i = 0; i < n; i++) { off[i] = pos; len[i] = src_len[i]; pos = (pos + src_len[i] + 15) & ~(uint64_t)15; }
    off[n] = pos;
    const size_t cap = pna_gpu_solid_archive_bound(algo, n, names, len.data());
    if (c->stage_in.ensure(pos + 8192) || c->stage_out.ensure(cap + 64) || c->hp_in[0].ensure(pos + 64) || c->hp_out[0].ensure(cap + 64))
        return fail(c, PNA_E_NOMEM, "staging allocation failed");
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    parallel_stage((uint8_t *)c->hp_in[0].p, src, src_len, off.data(), 0, n, std::min(8u, std::max(1u, hw / 2)));
    if (pos) HIPCHK(c, hipMemcpyAsync(c->stage_in.p, c->hp_in[0].p, pos, hipMemcpyHostToDevice, c->stream));
    uint64_t total = 0;
    int rc = pna_gpu_create_solid_archive_device(c, algo, level, n, names, c->stage_in.p, off.data(), len.data(), c->stage_out.p, cap + 64, &total, nullptr);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->hp_out[0].p, c->stage_out.p, total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint64_t p = 0; p < total; p += (16u << 20)) {
        const size_t k = (size_t)std::min<uint64_t>(16u << 20, total - p);
        if (sink(user, (const uint8_t *)c->hp_out[0].p + p, k) != 0) return fail(c, PNA_E_SINK, "sink failed");
    }
    return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Host-memory `pna create` (non-solid), bounded memory: the entries stream through two staging slots of at most
// ~1 GiB of input each.  While the GPU compresses and frames sub-batch k, helper threads stage sub-batch k+1 into
// page-locked memory and its H2D copy runs on a second stream; the archive bytes of sub-batch k-1 travel back on a
// third stream and are handed to the sink in one piece.  Replaces the reference's "every compressed entry in RAM until
// the scope ends" (cli/src/command/core.rs:496-537, create.rs:575-635) with a fixed in-flight window.
// ---- zero-staging input (round 4).  cli/src/command/core.rs:889-913 write_from_path reads every file into memory the library could own: with
// pna_gpu_host_alloc the host gets PAGE-LOCKED buffers to read its files into (read_exact into the slot instead of fs::read into a Vec), and the create
// entry points send entries that lie in such a buffer to the device straight from there -- the pageable -> page-locked copy on eight host threads is gone,
// one thread issues the copies.  Any mix works: a batch with an entry elsewhere is staged as before.
extern "C" int pna_gpu_host_alloc(pna_gpu_ctx *c, size_t bytes, void **out) {
    if (!c || !out || !bytes) return fail(c, PNA_E_INVAL, "null argument");
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return fail(c, PNA_E_NOMEM, "page-locked allocation failed");
    { std::lock_guard<std::mutex> lk(c->lent_mu); c->lent.emplace_back((const uint8_t *)p, bytes); }
    *out = p;
    return PNA_OK;
}
extern "C" int pna_gpu_host_free(pna_gpu_ctx *c, void *p) {
    if (!c || !p) return fail(c, PNA_E_INVAL, "null argument");
    {
        std::lock_guard<std::mutex> lk(c->lent_mu);
        auto it = std::find_if(c->lent.begin(), c->lent.end(), [&](const std::pair<const uint8_t *, size_t> &b) { return b.first == (const uint8_t *)p; });
        if (it == c->lent.end()) return fail(c, PNA_E_INVAL, "not a buffer of pna_gpu_host_alloc");
        c->lent.erase(it);
    }
    (void)hipSetDevice(c->device);
    (void)hipHostFree(p);
    return PNA_OK;
}
static bool entries_all_lent(pna_gpu_ctx *c, const void *const *src, const size_t *src_len, size_t n) {
    std::lock_guard<std::mutex> lk(c->lent_mu);
    if (c->lent.empty() || !n) return false;
    // (on the host-loop threads: 10^6 entries against the registry were 4 ms on one; the registry does not change while the lock is held)
    const unsigned nt = host_loop_threads(n);
    std::vector<char> ok(nt, 1);
    par_ranges(n, nt, [&](unsigned t, size_t a, size_t b) {
        size_t hint = 0;
        for (size_t i = a; i < b; i++) {
            if (!src_len[i]) continue;
            const uint8_t *p = (const uint8_t *)src[i];
            bool in = false;
            for (size_t k = 0; k < c->lent.size() && !in; k++) {                        // (entries of one call mostly share a buffer: start with the last hit)
                const auto &bf = c->lent[(hint + k) % c->lent.size()];
                if (p >= bf.first && p + src_len[i] <= bf.first + bf.second) { in = true; hint = (hint + k) % c->lent.size(); }
            }
            if (!in) { ok[t] = 0; return; }
        }
    });
    for (char v : ok) if (!v) return false;
    return true;
}

static void parallel_stage(uint8_t *dst, const void *const *src, const size_t *src_len, const uint64_t *off, size_t e0, size_t e1, unsigned threads) {
    uint64_t total = 0;
    for (size_t e = e0; e < e1; e++) total += src_len[e];
    if (threads <= 1 || total < (8u << 20)) { for (size_t e = e0; e < e1; e++) if (src_len[e]) memcpy(dst + off[e], src[e], src_len[e]); return; }
    std::vector<std::thread> th;
    const uint64_t per = (total + threads - 1) / threads;
    size_t e = e0;
    for (unsigned t = 0; t < threads && e < e1; t++) {
        size_t b = e; uint64_t acc = 0;
        while (e < e1 && (acc < per || t + 1 == threads)) acc += src_len[e++];
        th.emplace_back([=]() { for (size_t i = b; i < e; i++) if (src_len[i]) memcpy(dst + off[i], src[i], src_len[i]); });
    }
    for (auto &x : th) x.join();
}

extern "C" int pna_gpu_create_archive_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                           const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user) {
    return pna_gpu_create_archive_enc_host(c, algo, level, n, names, src, src_len, nullptr, sink, user);
}

extern "C" int pna_gpu_create_archive_enc_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                               const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                               pna_sink_fn sink, void *user) {
    return pna_gpu_create_archive_meta_host(c, algo, level, n, names, src, src_len, cipher, nullptr, sink, user);
}

static int create_archive_host_impl(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                    const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                    const pna_gpu_entry_meta *meta, uint32_t part_flags, pna_sink_fn sink, void *user, uint32_t max_chunk);
extern "C" int pna_gpu_create_archive_meta_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                                const pna_gpu_entry_meta *meta, pna_sink_fn sink, void *user) {
    return create_archive_host_impl(c, algo, level, n, names, src, src_len, cipher, meta, PNA_PART_HEAD | PNA_PART_TAIL, sink, user, c ? (uint32_t)c->tun.max_chunk_size : 0u);
}
extern "C" int pna_gpu_create_archive_chunked_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                   const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                                   const pna_gpu_entry_meta *meta, uint32_t max_chunk_size, uint32_t part_flags, pna_sink_fn sink, void *user) {
    return create_archive_host_impl(c, algo, level, n, names, src, src_len, cipher, meta, part_flags, sink, user, max_chunk_size);
}
extern "C" int pna_gpu_create_archive_part_host(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                                const void *const *src, const size_t *src_len, uint32_t part_flags, pna_sink_fn sink, void *user) {
    return create_archive_host_impl(c, algo, level, n, names, src, src_len, nullptr, nullptr, part_flags, sink, user, c ? (uint32_t)c->tun.max_chunk_size : 0u);
}
// `pna append`: Archive::seek_to_end, then the new entries and AEND where the old AEND stood (cli/src/command/append.rs:504-560)
extern "C" int pna_gpu_append_archive_host(pna_gpu_ctx *c, int algo, int level, const void *archive, size_t archive_len, size_t n,
                                           const char *const *names, const void *const *src, const size_t *src_len, uint64_t *write_at,
                                           pna_sink_fn sink, void *user) {
    if (!c || !archive || !write_at || !sink) return fail(c, PNA_E_INVAL, "null argument");
    int has_next = 0;
    if (pna_archive_seek_to_end(archive, archive_len, write_at, &has_next) != PNA_OK) return fail(c, PNA_E_INVAL, "not a PNA archive, or truncated before its AEND chunk");
    if (has_next) return fail(c, PNA_E_INVAL, "the archive continues in another part (ANXT): append to its last part");
    return create_archive_host_impl(c, algo, level, n, names, src, src_len, nullptr, nullptr, PNA_PART_TAIL, sink, user, (uint32_t)c->tun.max_chunk_size);
}
// One process, several GPUs (SURVEY §8(b)'s `device_ids, n_devices`; §8(e)'s comparison path in C): the entries are cut into contiguous
// index ranges balanced by bytes, one per context (= per device), every context runs the bounded host pipeline on its range on a thread of
// its own (part flags: the first range carries the archive header, the last AEND) into host memory, and the parts reach the sink in index
// order -- the reference's fan-out + ordered drain (cli/src/command/core.rs:496-537,471-493) with devices in place of rayon workers and no
// device-to-device traffic at all.  Contexts may share a device (that is how the one-GPU boxes test it).
extern "C" int pna_gpu_create_archive_multi_host(pna_gpu_ctx *const *ctxs, size_t n_ctx, int algo, int level, size_t n, const char *const *names,
                                                 const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user) {
    if (!ctxs || !n_ctx || !sink || (n && (!names || !src || !src_len))) return PNA_E_INVAL;
    for (size_t r = 0; r < n_ctx; r++) if (!ctxs[r]) return PNA_E_INVAL;
    if (n_ctx == 1) return pna_gpu_create_archive_host(ctxs[0], algo, level, n, names, src, src_len, sink, user);
    // contiguous ranges balanced by input bytes (shard.partition_entries)
    uint64_t total = 0; for (size_t i = 0; i < n; i++) total += src_len[i];
    std::vector<size_t> lo(n_ctx + 1, n);
    {   // range r ends where the running byte count passes r + 1 shares of the total (a range may be empty when there are few entries)
        size_t i = 0; uint64_t acc = 0;
        for (size_t r = 0; r < n_ctx; r++) {
            lo[r] = i;
            const uint64_t target = (uint64_t)((__uint128_t)total * (r + 1) / n_ctx);
            while (i < n && (r + 1 == n_ctx || acc + src_len[i] <= target)) { acc += src_len[i]; i++; }
        }
        lo[n_ctx] = n;
    }
    // Every range runs the bounded pipeline on a thread of its own; range 0 hands its pieces to the caller's sink as they come, the ranges behind it keep
    // theirs in memory until every range before them has finished (the sink sees the archive in index order, on the calling thread only).  An exception
    // inside a worker (allocation) is that range's PNA_E_NOMEM, not a terminate; the first failing range's code is returned and its message copied to
    // ctxs[0] (what pna_gpu_last_error of the first context reports).
    struct Part { std::vector<uint8_t> buf; int rc = PNA_OK; bool done = false; };
    std::vector<Part> parts(n_ctx);
    // every range picks the block size of the WHOLE call (pna_ctx.h CallTotalScope): the archive equals pna_gpu_create_archive_host's on one context
    std::vector<std::unique_ptr<CallTotalScope>> whole;
    for (size_t r = 0; r < n_ctx; r++) whole.emplace_back(new CallTotalScope(ctxs[r], total));
    std::mutex mu; std::condition_variable cv;
    auto vec_sink = [](void *u, const void *b, size_t k) -> int { auto *v = (std::vector<uint8_t> *)u; try { v->insert(v->end(), (const uint8_t *)b, (const uint8_t *)b + k); } catch (...) { return 1; } return 0; };
    std::vector<std::thread> th;
    for (size_t r = 1; r < n_ctx; r++)
        th.emplace_back([&, r]() {
            int rc;
            try {
                const uint32_t pf = r + 1 == n_ctx ? PNA_PART_TAIL : 0u;
                rc = pna_gpu_create_archive_part_host(ctxs[r], algo, level, lo[r + 1] - lo[r], names + lo[r], src + lo[r], src_len + lo[r], pf, vec_sink, &parts[r].buf);
            } catch (...) { rc = PNA_E_NOMEM; }
            std::lock_guard<std::mutex> lk(mu);
            parts[r].rc = rc; parts[r].done = true; cv.notify_all();
        });
    int rc0;
    try { rc0 = pna_gpu_create_archive_part_host(ctxs[0], algo, level, lo[1] - lo[0], names + lo[0], src + lo[0], src_len + lo[0], PNA_PART_HEAD, sink, user); }
    catch (...) { rc0 = fail(ctxs[0], PNA_E_NOMEM, "out of memory"); }
    int rc = rc0;
    for (size_t r = 1; r < n_ctx; r++) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return parts[r].done; }); }
        if (rc == PNA_OK && parts[r].rc != PNA_OK) { rc = parts[r].rc; const std::string msg = std::string("range ") + std::to_string(r) + ": " + pna_gpu_last_error(ctxs[r]); (void)fail(ctxs[0], rc, msg.c_str()); }
        if (rc == PNA_OK && !parts[r].buf.empty() && sink(user, parts[r].buf.data(), parts[r].buf.size()) != 0) rc = fail(ctxs[0], PNA_E_SINK, "sink failed");
        std::vector<uint8_t>().swap(parts[r].buf);                // handed on (or abandoned): the memory goes back at once
    }
    for (auto &t : th) t.join();
    return rc;
}
static int create_archive_host_impl(pna_gpu_ctx *c, int algo, int level, size_t n, const char *const *names,
                                    const void *const *src, const size_t *src_len, const pna_gpu_cipher *cipher,
                                    const pna_gpu_entry_meta *meta, uint32_t part_flags, pna_sink_fn sink, void *user, uint32_t max_chunk) {
    const auto t_entry = std::chrono::steady_clock::now();
    { int rcm = check_meta(c, meta, n); if (rcm) return rcm; }
    if (!c || !sink || (n && (!names || !src || !src_len))) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    set_call_level(c, algo, level);
    if (cipher && cipher->encryption == PNA_ENC_NONE) cipher = nullptr;
    std::vector<uint8_t> own_ivs;
    const uint8_t *ivs = nullptr;
    if (cipher) { int rc0 = resolve_ivs(c, cipher, n, own_ivs, &ivs); if (rc0) return rc0; }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->cp_in) {
        HIPCHK(c, hipStreamCreate(&c->cp_in)); HIPCHK(c, hipStreamCreate(&c->cp_out));
        for (auto &e : c->ev_in) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : c->ev_out) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    c->timing = pna_gpu_timing{};
    std::vector<uint8_t> head, tail;
    if (part_flags & PNA_PART_HEAD) frame_archive_head(head, 0);
    if (part_flags & PNA_PART_TAIL) frame_archive_tail(tail);
    if (!head.empty() && sink(user, head.data(), head.size()) != 0) return fail(c, PNA_E_SINK, "sink failed");
    // Sub-batches.  What bounds this path is the host link in the H2D direction (page-locked memory -> HBM: 56.8 GB/s on the MI355X boxes;
    // experiments/link_duplex.hip, profiles/r03_c_link_duplex.txt) -- the kernels take a sixth of that time, the archive bytes going back a third of the
    // volume and the link is full duplex.  So the pipeline is built around ONE rule: the H2D copy engine never waits.
    //   * a stager thread runs ahead through all sub-batches over a ring of four input slots (page-locked staging + device buffer): it copies
    //     the entries of a sub-batch into the slot's page-locked buffer (several threads, groups of ~128 MiB) and issues each group's H2D copy
    //     right behind it; it blocks only while all four slots are in use (a slot is free again when its sub-batch's kernels are done);
    //   * the main thread takes the sub-batches in order: kernels on the context's stream, then the archive bytes of the sub-batch travel to
    //     the host next to the following sub-batch's kernels and copies -- by a small copy KERNEL that stores into the page-locked buffer
    //     (d2h_wgs workgroups: ~30 GB/s, which leaves the H2D engine its full rate; the runtime's own D2H copy ran as a blit kernel at 51 GB/s
    //     and took 30 % off the H2D copies next to it) --, and are handed to the sink one sub-batch later;
    //   * sub-batch sizes grow from 64 MiB to `sub_mib` (default 256 MiB) at the start: the first kernels start after 2 ms instead of 20, and
    //     what is left to do when the last input byte has arrived is the work of one sub-batch.  256 MiB is the smallest size whose kernels
    //     (1.2 ms of fixed costs + 1 ms per 85 MiB) keep up with its H2D copy (1 ms per 53 MiB); shrinking sizes at the end only makes the
    //     kernels fall behind the copies (measured: option sub_ramp_down).
    const uint64_t SUBMAX = (uint64_t)c->tun.sub_mib << 20, SUBMIN = std::min<uint64_t>(64ull << 20, SUBMAX);
    struct Sub { size_t e0, e1; uint64_t in_bytes, out_cap; };
    std::vector<Sub> subs;
    // (the context keeps these arrays between calls: 10^6 entries are 24 MB, and fresh memory costs a page fault per 4 KiB -- milliseconds in front of the pipeline)
    if (c->pl_off.size() < n + 1) { c->pl_off.resize(n + 1); c->pl_len.resize(n + 1); c->pl_cap.resize(n + 1); }
    uint64_t *off = c->pl_off.data(), *len64 = c->pl_len.data(), *ecap = c->pl_cap.data();
    uint64_t in_total = 0, longest = 0;
    // every entry's share of its sub-batch's output capacity (name length, worst-case payload, chunk framing): on the host-loop threads -- for 10^5 .. 10^6 small
    // entries this loop, on one thread, was 8 .. 40 ms in front of the pipeline (a third of the end-to-end time of 10^6 x 4 KiB)
    const unsigned plan_nt = host_loop_threads(n);
    std::vector<std::pair<uint64_t, uint64_t>> psum(plan_nt, {0, 0});         // (bytes, longest entry) per thread
    par_ranges(n, plan_nt, [&](unsigned t, size_t a, size_t b) {
        uint64_t tot = 0, mx = 0;
        for (size_t i = a; i < b; i++) {
            const uint64_t l = src_len[i], wb = pna_gpu_bound(algo, (size_t)l);
            tot += l; mx = std::max(mx, l);
            uint64_t cap = (cipher ? frame_entry_prefix_enc_bound(names[i], cipher->phsf) + 16 : frame_entry_prefix_bound(names[i])) + meta_len(meta, i) + wb + 16;
            if (cipher && cipher->cipher_mode == PNA_MODE_GCM) cap += 16 * (wb / (cipher->gcm_segment_size ? cipher->gcm_segment_size : (1u << 20)));   // a tag per full stream segment
            // a CRC + a header per further FDAT chunk once max_chunk_size cuts the payload (the term of pna_gpu_archive_chunked_bound: without it data
            // that does not compress overran the sub-batch's device buffer by 12 bytes per chunk -- PNA_E_DSTSIZE for a 2 MiB random entry at mcs = 1000)
            cap += 12 * (uint64_t)((wb + 64 + 16 * (l >> 12)) / chunk_limit(max_chunk) + 1);
            ecap[i] = cap;
        }
        psum[t] = {tot, mx};
    });
    for (auto &q : psum) { in_total += q.first; longest = std::max(longest, q.second); }
    plan_call_longest(c, longest);
    const auto t_caps = std::chrono::steady_clock::now();
    {
        uint64_t done = 0, target = SUBMIN;
        for (size_t e = 0; e < n;) {
            const uint64_t rest = in_total - done;
            uint64_t want = std::min(target, SUBMAX);
            if (c->tun.sub_ramp_down && rest < 2 * want) want = std::max(SUBMIN, rest / 2);   // (option) ramp down: half of what is left, not below the minimum
            (void)rest;
            Sub sb{e, e, 0, 64}; uint64_t pos = 0; size_t blocks = 0;
            while (sb.e1 < n) {
                const size_t i = sb.e1; const uint64_t l = src_len[i];
                const size_t nb = plan_blocks(c, l);
                if (i > sb.e0 && (pos + l > want || blocks + nb > c->max_blocks)) break;
                off[i] = pos; len64[i] = l; pos = (pos + l + 15) & ~(uint64_t)15; blocks += nb;
                sb.out_cap += ecap[i];
                done += l; sb.e1++;
            }
            sb.in_bytes = pos; subs.push_back(sb); e = sb.e1;
            target = std::min(SUBMAX, target * 2);
        }
    }
    const auto t_cut = std::chrono::steady_clock::now();
    const bool all_lent = entries_all_lent(c, src, src_len, n);                     // (then no page-locked staging of the library's own is needed, and no staging copy)
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned threads = std::min(8u, std::max(1u, hw / 2));
    if (c->tun.stage_threads) threads = (unsigned)c->tun.stage_threads;
    FrameJob fj{names, 0, cipher, ivs, meta, max_chunk, false};
    if (c->pl_eoff.size() < n + 1) c->pl_eoff.resize(n + 1);
    uint64_t *eoff = c->pl_eoff.data();
    uint64_t out_len[2] = {0, 0}, out_total = head.size();
    constexpr int NS = 4;
    {   // slots sized once for the largest sub-batch (allocation of page-locked memory is slow: not inside the pipeline)
        uint64_t max_in = 0, max_out = 0;
        for (const Sub &sb : subs) { max_in = std::max(max_in, sb.in_bytes); max_out = std::max(max_out, sb.out_cap); }
        for (int s = 0; s < NS && s < (int)subs.size(); s++)
            if ((!all_lent && c->hp_in[s].ensure(max_in + 8192)) || c->dp_in[s].ensure(max_in + 8192)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
        for (int s = 0; s < 2 && s < (int)subs.size(); s++)
            if (c->dp_out[s].ensure(max_out + 64) || c->hp_out[s].ensure(max_out + 64)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    }
    int rc = PNA_OK;
    const auto tr0 = std::chrono::steady_clock::now();
    auto trace = [&](const char *what, size_t k) { if (c->tun.trace) fprintf(stderr, "[pna create] %8.3f ms  %s %zu\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count(), what, k); };
    if (c->tun.trace) fprintf(stderr, "[pna create] planning took %.3f ms (capacity terms at %.3f, sub-batches cut at %.3f)\n", std::chrono::duration<double, std::milli>(tr0 - t_entry).count(),
                              std::chrono::duration<double, std::milli>(t_caps - t_entry).count(), std::chrono::duration<double, std::milli>(t_cut - t_entry).count());
    trace("planned, slots ready; sub-batches:", subs.size());
    // the stager: sub-batch k into slot k % NS as soon as sub-batch k - NS has left the device
    std::mutex mu; std::condition_variable cv;
    size_t staged = 0, freed = 0; int stager_rc = PNA_OK; bool stop = false;      // sub-batches staged (copies issued) / sub-batches whose kernels are done
    const int dev_id = c->device;
    hipStream_t cp_in = c->cp_in;
    std::thread stager([&]() {
        try {
            if (hipSetDevice(dev_id) != hipSuccess) { std::lock_guard<std::mutex> lk(mu); stager_rc = PNA_E_HIP; staged = subs.size(); cv.notify_all(); return; }
            for (size_t k = 0; k < subs.size(); k++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || k < freed + NS; });
                    if (stop) return;
                }
                const Sub &nx = subs[k]; const int sl = (int)(k % NS);
                uint8_t *hb = (uint8_t *)c->hp_in[sl].p, *db = (uint8_t *)c->dp_in[sl].p;
                int r = PNA_OK;
                size_t g0 = nx.e0;
                if (all_lent) {
                    // every entry lies in page-locked memory the library lent out (pna_gpu_host_alloc): no staging copy, the copy engine reads the host's
                    // buffers themselves -- runs of entries that are contiguous there and here (16-byte stride) travel as ONE copy
                    while (g0 < nx.e1 && r == PNA_OK) {
                        size_t g1 = g0 + 1;
                        while (g1 < nx.e1 && (const uint8_t *)src[g1] == (const uint8_t *)src[g0] + (off[g1] - off[g0])) g1++;
                        const uint64_t bytes = (g1 < nx.e1 ? off[g1] : off[g1 - 1] + src_len[g1 - 1]) - off[g0];
                        const uint64_t k = std::min<uint64_t>(bytes, (const uint8_t *)src[g1 - 1] + src_len[g1 - 1] - (const uint8_t *)src[g0]);
                        if (k && hipMemcpyAsync(db + off[g0], src[g0], k, hipMemcpyHostToDevice, cp_in) != hipSuccess) r = PNA_E_HIP;
                        g0 = g1;
                    }
                }
                while (g0 < nx.e1 && r == PNA_OK) {
                    size_t g1 = g0; uint64_t acc = 0;
                    while (g1 < nx.e1 && acc < (128ull << 20)) acc += src_len[g1++];
                    parallel_stage(hb, src, src_len, off, g0, g1, threads);
                    const uint64_t b0 = off[g0], b1 = g1 < nx.e1 ? off[g1] : nx.in_bytes;
                    if (b1 > b0 && hipMemcpyAsync(db + b0, hb + b0, b1 - b0, hipMemcpyHostToDevice, cp_in) != hipSuccess) r = PNA_E_HIP;
                    g0 = g1;
                }
                if (r == PNA_OK && hipEventRecord(c->ev_in[sl], cp_in) != hipSuccess) r = PNA_E_HIP;
                trace("staged + H2D issued", k);
                std::lock_guard<std::mutex> lk(mu);
                if (r != PNA_OK) { stager_rc = r; staged = subs.size(); cv.notify_all(); return; }
                staged = k + 1; cv.notify_all();
            }
        } catch (...) { std::lock_guard<std::mutex> lk(mu); stager_rc = PNA_E_NOMEM; staged = subs.size(); cv.notify_all(); }
    });
    const CallTotalScope call_total(c, len64, n);                                                   // (the block size follows the archive, not its sub-batches)
    uint8_t *hp_out_dev[2] = {nullptr, nullptr};                 // device views of the page-locked output slots (the copy kernel's destination)
    const uint32_t d2h_wgs = (uint32_t)c->tun.d2h_wgs;
    for (size_t k = 0; k < subs.size() && rc == PNA_OK; k++) {
        const Sub &sb = subs[k]; const int sl = (int)(k % NS), so = (int)(k & 1);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return staged > k; });
            if (stager_rc != PNA_OK) { rc = fail(c, stager_rc, "staging / H2D copy failed"); break; }
        }
        if (hipEventSynchronize(c->ev_in[sl]) != hipSuccess) { rc = fail(c, PNA_E_HIP, "H2D copy failed"); break; }
        trace("H2D done, kernels start", k);
        rc = run_subbatch(c, algo, (const uint8_t *)c->dp_in[sl].p, off, len64, sb.e0, sb.e1, (uint8_t *)c->dp_out[so].p,
                          sb.out_cap + 64, 0, eoff, c->stream, true, &fj);
        { std::lock_guard<std::mutex> lk(mu); freed = k + 1; cv.notify_all(); }       // (run_subbatch has waited for its kernels: the input slot is free)
        trace("kernels done", k);
        if (rc == PNA_OK) {
            out_len[so] = eoff[sb.e1];
            bool ok = true;
            if (d2h_wgs && !hp_out_dev[so]) ok = hipHostGetDevicePointer((void **)&hp_out_dev[so], c->hp_out[so].p, 0) == hipSuccess;
            if (ok && d2h_wgs) { launch_link_copy((const uint8_t *)c->dp_out[so].p, hp_out_dev[so], out_len[so], d2h_wgs, c->cp_out); ok = hipGetLastError() == hipSuccess; }
            else if (ok) ok = hipMemcpyAsync(c->hp_out[so].p, c->dp_out[so].p, out_len[so], hipMemcpyDeviceToHost, c->cp_out) == hipSuccess;
            if (!ok || hipEventRecord(c->ev_out[so], c->cp_out) != hipSuccess) rc = fail(c, PNA_E_HIP, "D2H copy failed");
        }
        if (rc == PNA_OK && k > 0) {                             // archive bytes of the previous sub-batch -> sink
            if (hipEventSynchronize(c->ev_out[so ^ 1]) != hipSuccess) rc = fail(c, PNA_E_HIP, "D2H copy failed");
            else if (out_len[so ^ 1] && sink(user, c->hp_out[so ^ 1].p, out_len[so ^ 1]) != 0) rc = fail(c, PNA_E_SINK, "sink failed");
            out_total += out_len[so ^ 1];
        }
    }
    { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); }
    stager.join();
    trace("loop done", 0);
    if (rc != PNA_OK) { (void)hipDeviceSynchronize(); return rc; }
    if (!subs.empty()) {
        const int so = (int)((subs.size() - 1) & 1);
        HIPCHK(c, hipEventSynchronize(c->ev_out[so]));
        if (out_len[so] && sink(user, c->hp_out[so].p, out_len[so]) != 0) return fail(c, PNA_E_SINK, "sink failed");
        out_total += out_len[so];
    }
    if (!tail.empty() && sink(user, tail.data(), tail.size()) != 0) return fail(c, PNA_E_SINK, "sink failed");
    out_total += tail.size();
    trace("all bytes handed to the sink", 0);
    c->timing.in_bytes = in_total; c->timing.out_bytes = out_total;
    return PNA_OK;
}


extern "C" int pna_gpu_compress_batch(pna_gpu_ctx *c, int algo, int level, size_t n, const void *const *src,
                                      const size_t *src_len, void *const *dst, const size_t *dst_cap, size_t *dst_len) {
    if (!c || (n && (!src || !src_len || !dst || !dst_cap || !dst_len))) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> off(n + 1), len(n), doff(n + 1), obase(n + 1);       // obase: running sum of the entries' bounds
    uint64_t pos = 0, bound = 0;
    for (size_t i = 0; i < n; i++) {
        if (dst_cap[i] < pna_gpu_bound(algo, src_len[i])) return fail(c, PNA_E_DSTSIZE, "dst_cap below pna_gpu_bound");
        off[i] = pos; len[i] = src_len[i]; pos = (pos + src_len[i] + 15) & ~(uint64_t)15;
        obase[i] = bound; bound += pna_gpu_bound(algo, src_len[i]);
    }
    off[n] = pos; obase[n] = bound;
    const CallTotalScope call_total(c, src_len, n);              // (the pieces below pick the block size of the whole call: the bytes equal pna_gpu_compress_batch_device's)
    // Inputs are staged into page-locked memory by several threads and copied H2D, outputs copied D2H and scattered by several threads
    // (per-entry copies from pageable memory ran at ~1 GiB/s).  A large batch goes through in PIECES of >= 256 MiB (a round of the CUs:
    // the kernels' fixed latencies stay amortised): piece k + 1 is staged and copied while piece k is on the device, piece k - 1's results
    // travel back meanwhile -- 512 x 1 MiB: 27.5 -> see profiles/ (PNA_BATCH_PIECE_MIB; 0 = one piece).
    const uint64_t piece_bytes = c->tun.batch_piece_mib <= 0 ? ~0ull >> 1 : (uint64_t)c->tun.batch_piece_mib << 20;
    std::vector<size_t> pe{0};                                   // piece k = entries [pe[k], pe[k + 1])
    for (size_t i = 0; i < n;) {
        size_t j = i; uint64_t acc = 0;
        while (j < n && (j == i || acc + src_len[j] <= piece_bytes)) acc += src_len[j++];
        pe.push_back(j); i = j;
    }
    if (pe.size() > 2 && off[n] - off[pe[pe.size() - 2]] < piece_bytes / 2) pe.erase(pe.end() - 2);   // a short last piece joins its neighbour
    const size_t K = pe.size() - 1;
    uint64_t max_in = 0, max_out = 0;
    std::vector<uint64_t> pbase(K + 1, 0);                       // where piece k's output starts in stage_out (256-byte aligned)
    for (size_t k = 0; k < K; k++) {
        max_in = std::max(max_in, off[pe[k + 1]] - off[pe[k]]); max_out = std::max(max_out, obase[pe[k + 1]] - obase[pe[k]]);
        pbase[k + 1] = (pbase[k] + (obase[pe[k + 1]] - obase[pe[k]]) + 64 + 255) & ~(uint64_t)255;
    }
    if (c->stage_in.ensure(pos + 8192) || c->stage_out.ensure(pbase[K] + 64)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    for (int sl = 0; sl < (K > 1 ? 2 : 1); sl++) if (c->hp_in[sl].ensure(max_in + 64) || c->hp_out[sl].ensure(max_out + 64)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    if (K > 1 && !c->cp_in) {
        HIPCHK(c, hipStreamCreate(&c->cp_in)); HIPCHK(c, hipStreamCreate(&c->cp_out));
        for (auto &e : c->ev_in) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : c->ev_out) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned threads = std::min(8u, std::max(1u, hw / 2));
    hipStream_t s_in = K > 1 ? c->cp_in : c->stream, s_out = K > 1 ? c->cp_out : c->stream;
    auto stage = [&](size_t k) -> int {                          // entries of piece k -> pinned slot -> their place in stage_in
        const int sl = (int)(k & 1);
        const uint64_t b0 = off[pe[k]], nb = off[pe[k + 1]] - b0;
        std::vector<uint64_t> rel(pe[k + 1] - pe[k] + 1);
        for (size_t i = pe[k]; i <= pe[k + 1]; i++) rel[i - pe[k]] = off[i] - b0;
        parallel_stage((uint8_t *)c->hp_in[sl].p, src + pe[k], src_len + pe[k], rel.data(), 0, pe[k + 1] - pe[k], threads);
        if (nb) HIPCHK(c, hipMemcpyAsync((uint8_t *)c->stage_in.p + b0, c->hp_in[sl].p, nb, hipMemcpyHostToDevice, s_in));
        if (K > 1) HIPCHK(c, hipEventRecord(c->ev_in[sl], s_in));
        return PNA_OK;
    };
    std::vector<uint64_t> ptotal(K);
    auto scatter = [&](size_t k) -> int {                        // piece k's compressed entries: pinned slot -> the caller's buffers
        const int sl = (int)(k & 1);
        if (K > 1) HIPCHK(c, hipEventSynchronize(c->ev_out[sl])); else HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint8_t *hb = (const uint8_t *)c->hp_out[sl].p;
        const size_t e0 = pe[k], e1 = pe[k + 1];
        const unsigned T = ptotal[k] < (8u << 20) ? 1u : threads;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([=, &doff]() { for (size_t i = e0 + t; i < e1; i += T) if (dst_len[i]) memcpy(dst[i], hb + (doff[i] - doff[e0]), dst_len[i]); });
        for (auto &x : th) x.join();
        return PNA_OK;
    };
    pna_gpu_timing tsum{};
    int rc = K ? stage(0) : PNA_OK;
    if (rc) return rc;
    doff[0] = 0;
    for (size_t k = 0; k < K; k++) {
        const int sl = (int)(k & 1);
        if (k + 1 < K && (rc = stage(k + 1))) return rc;         // (its slot's previous copy, piece k - 1, was waited for by that piece's kernels)
        if (K > 1) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_in[sl], 0));
        const size_t e0 = pe[k], nk = pe[k + 1] - e0;
        std::vector<uint64_t> d(nk + 1);
        uint8_t *ob = (uint8_t *)c->stage_out.p + pbase[k];
        rc = pna_gpu_compress_batch_device(c, algo, level, nk, c->stage_in.p, off.data() + e0, len.data() + e0, ob, obase[pe[k + 1]] - obase[e0] + 64, d.data(), nullptr);
        if (rc) return rc;
        { const pna_gpu_timing &t = c->timing; tsum.ms_lz += t.ms_lz; tsum.ms_stats += t.ms_stats; tsum.ms_lit += t.ms_lit; tsum.ms_seq += t.ms_seq; tsum.ms_pack += t.ms_pack;
          tsum.in_bytes += t.in_bytes; tsum.out_bytes += t.out_bytes; tsum.n_segments += t.n_segments; tsum.n_blocks += t.n_blocks; tsum.ms_lz_match += t.ms_lz_match; tsum.lz_match_launches += t.lz_match_launches; }
        for (size_t i = 0; i < nk; i++) { doff[e0 + i + 1] = doff[e0] + d[i + 1]; dst_len[e0 + i] = (size_t)(d[i + 1] - d[i]); }
        ptotal[k] = d[nk];
        if (ptotal[k]) HIPCHK(c, hipMemcpyAsync(c->hp_out[sl].p, ob, ptotal[k], hipMemcpyDeviceToHost, s_out));   // (the kernels are done: compress_batch_device returns synchronised)
        if (K > 1) HIPCHK(c, hipEventRecord(c->ev_out[sl], s_out));
        if (k >= 1 && (rc = scatter(k - 1))) return rc;
    }
    if (K && (rc = scatter(K - 1))) return rc;
    c->timing = tsum;
    return PNA_OK;
}

