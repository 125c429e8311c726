// pna_stream.cpp -- the CompressionWriter seam: pna_gpu_stream_* (group commit as a three-stage pipeline), stream entries, pna_gpu_compress_solid.
#include "pna_ctx.h"
// ---------------------------------------------------------------------------------------------------------
// The seam is used the way the reference uses its encoders: one writer per rayon task, many tasks in flight on many host threads
// (cli/src/command/core.rs:505-517).  One entry per device batch would leave the GPU idle, so finish() is a GROUP COMMIT: the
// stream joins the context's queue; the first thread to find no leader becomes the leader, takes everything queued so far, runs
// ONE pna_gpu_compress_batch for it and wakes the owners, each of which drains its own stream into its own sink on its own thread
// (W::write is never called from a foreign thread).  While a batch runs, the finishes that arrive pile up and form the next,
// larger batch -- no timer needed under load (PNA_STREAM_LINGER_US adds an optional wait for stragglers).
// one large copy on several threads (a pageable stream of GiBs, e.g. pna_gpu_compress_solid over a whole solid archive)
static void big_memcpy(uint8_t *dst, const uint8_t *src, size_t n) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned T = n < (64u << 20) ? 1u : std::min(8u, std::max(1u, hw / 2));
    if (T == 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = ((n + T - 1) / T + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < T; t++) {
        const size_t a = std::min(n, (size_t)t * per), b = std::min(n, a + per);
        if (b > a) th.emplace_back([=]() { memcpy(dst + a, src + a, b - a); });
    }
    for (auto &x : th) x.join();
}
constexpr size_t S_SLAB = 1u << 20, S_ARENA = 64u << 20, S_MAX_SLABS = 256;   // a stream beyond 256 MiB continues in pageable memory
struct pna_gpu_stream {
    pna_gpu_ctx *ctx; int algo, level; pna_sink_fn sink; void *user;
    std::vector<uint8_t *> slabs; size_t slab_len = 0;      // page-locked mode: bytes [k * S_SLAB, ...) live in slabs[k]
    bool pageable = false; std::vector<uint8_t> buf;        // pageable mode (pool exhausted / very large stream): everything in buf
    const uint8_t *out = nullptr; size_t out_len = 0; int rc = PNA_OK, slot = 0; bool done = false, queued = false;
    size_t total() const { return pageable ? buf.size() : slab_len; }
};

static uint8_t *pool_get(pna_gpu_ctx *c) {
    std::lock_guard<std::mutex> lk(c->pool_mu);
    if (c->pool_free.empty()) {
        if (c->pool_bytes + S_ARENA > c->pool_cap) return nullptr;
        void *p = nullptr;
        if (hipSetDevice(c->device) != hipSuccess || hipHostMalloc(&p, S_ARENA, hipHostMallocDefault) != hipSuccess) return nullptr;
        c->pool_arenas.push_back(p); c->pool_bytes += S_ARENA;
        for (size_t k = 0; k < S_ARENA / S_SLAB; k++) c->pool_free.push_back((uint8_t *)p + k * S_SLAB);
    }
    uint8_t *r = c->pool_free.back(); c->pool_free.pop_back();
    return r;
}
static void pool_put(pna_gpu_ctx *c, std::vector<uint8_t *> &slabs) {
    if (slabs.empty()) return;
    std::lock_guard<std::mutex> lk(c->pool_mu);
    for (uint8_t *p : slabs) c->pool_free.push_back(p);
    slabs.clear();
}

extern "C" int pna_gpu_stream_new(pna_gpu_ctx *c, int algo, int level, pna_sink_fn sink, void *user, pna_gpu_stream **out) {
    if (!c || !sink || !out) return PNA_E_INVAL;
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return PNA_E_UNSUPPORTED;   // no fail(): other threads may be inside the context
    pna_gpu_stream *s = new (std::nothrow) pna_gpu_stream();
    if (!s) return PNA_E_NOMEM;
    s->ctx = c; s->algo = algo; s->level = level; s->sink = sink; s->user = user;
    *out = s;
    return PNA_OK;
}
extern "C" int pna_gpu_stream_write(pna_gpu_stream *s, const void *buf, size_t len) {
    if (!s || (!buf && len)) return PNA_E_INVAL;
    const uint8_t *p = (const uint8_t *)buf; size_t left = len;
    try {
        while (left && !s->pageable) {
            const size_t in_slab = s->slab_len % S_SLAB;
            if (s->slab_len == s->slabs.size() * S_SLAB) {                  // the last slab is full (or there is none yet)
                uint8_t *sl = s->slabs.size() < S_MAX_SLABS ? pool_get(s->ctx) : nullptr;
                if (!sl) {                                                  // continue in pageable memory
                    s->buf.reserve(s->slab_len + left);
                    for (size_t k = 0; k < s->slabs.size(); k++) s->buf.insert(s->buf.end(), s->slabs[k], s->slabs[k] + std::min(S_SLAB, s->slab_len - k * S_SLAB));
                    pool_put(s->ctx, s->slabs); s->slab_len = 0; s->pageable = true;
                    break;
                }
                s->slabs.push_back(sl);
            }
            const size_t k = std::min(left, S_SLAB - in_slab);
            memcpy(s->slabs.back() + in_slab, p, k); p += k; left -= k; s->slab_len += k;
        }
        if (left) s->buf.insert(s->buf.end(), p, p + left);
    } catch (const std::bad_alloc &) { return PNA_E_NOMEM; }
    return PNA_OK;
}
extern "C" int pna_gpu_stream_flush(pna_gpu_stream *s) { return s ? PNA_OK : PNA_E_INVAL; }
extern "C" void pna_gpu_stream_abort(pna_gpu_stream *s) { if (s) { pool_put(s->ctx, s->slabs); delete s; } }

// One batch of the facade = the streams a leader took, per (algo, level) group: (1) H2D copies straight from the streams' page-locked slabs (pageable
// streams are staged first) on the copy-in stream; `in_done()` then hands the leader's role on, and the next batch is copied in while this one runs;
// (2) the device batch under run_mu; (3) ONE D2H copy of the group's streams into the slot's page-locked output on the copy-out stream.
static void stream_run_batch(pna_gpu_ctx *c, const std::vector<pna_gpu_stream *> &batch, int slot, const std::function<void()> &in_done, const std::function<void()> &on_device,
                             const std::function<void()> &off_device, const std::function<void()> &in_failed) {
    auto set_err = [&](int code, const char *what) { std::lock_guard<std::mutex> lk(c->err_mu); return fail(c, code, what); };
    auto fail_all = [&](int rc) { for (pna_gpu_stream *x : batch) if (x->rc == PNA_OK && !x->out) { x->rc = rc; x->out_len = 0; } };
    struct Grp { std::vector<pna_gpu_stream *> st; std::vector<uint64_t> off, len, doff; uint64_t in_base = 0, in_bytes = 0, bound = 0, out_base = 0; int rc = PNA_OK; };
    std::vector<Grp> groups;
    static const bool trace = getenv("PNA_STREAM_TRACE") != nullptr;         // per-batch phase times on stderr
    const auto t0 = std::chrono::steady_clock::now();
    // ---- stage 1: plan + copy in (the leader still holds comb_leader: one batch at a time in this stage)
    int rc0 = PNA_OK;
    {
        if (hipSetDevice(c->device) != hipSuccess) rc0 = set_err(PNA_E_HIP, "hipSetDevice failed");
        if (rc0 == PNA_OK && !c->s_h2d) {
            if (hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->s_d2h, hipStreamNonBlocking) != hipSuccess) rc0 = set_err(PNA_E_HIP, "stream creation failed");
            for (auto &e : c->s_ev) if (rc0 == PNA_OK && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc0 = set_err(PNA_E_HIP, "event creation failed");
        }
        std::vector<char> taken(batch.size(), 0);
        uint64_t in_total = 0, bound_total = 0, out_cap = 64, page_bytes = 0;
        for (size_t i = 0; i < batch.size(); i++) {
            if (taken[i]) continue;
            Grp g;
            for (size_t q = i; q < batch.size(); q++)
                if (!taken[q] && batch[q]->algo == batch[i]->algo && batch[q]->level == batch[i]->level) { taken[q] = 1; g.st.push_back(batch[q]); }
            const size_t n = g.st.size();
            g.off.resize(n + 1); g.len.resize(n); g.doff.resize(n + 1);
            uint64_t pos = 0;
            for (size_t k = 0; k < n; k++) { g.off[k] = pos; g.len[k] = g.st[k]->total(); pos = (pos + g.len[k] + 15) & ~(uint64_t)15; g.bound += pna_gpu_bound(g.st[k]->algo, (size_t)g.len[k]) + 16; }
            g.off[n] = pos; g.in_bytes = pos;
            g.in_base = in_total; in_total += (pos + 8192 + 255) & ~(uint64_t)255;
            g.out_base = bound_total; bound_total += (g.bound + 64 + 255) & ~(uint64_t)255;
            groups.push_back(std::move(g));
        }
        for (pna_gpu_stream *x : batch) { out_cap += pna_gpu_bound(x->algo, x->total()) + 32; if (x->pageable) page_bytes += (x->buf.size() + 15) & ~(size_t)15; }
        // (buffers of a slot are sized for a full batch at once: growing them batch by batch cost the first seconds of a run 10 - 20 ms of page-locking each)
        const uint64_t capb = ((uint64_t)c->tun.stream_batch_mib << 20), cap_out = pna_gpu_bound(PNA_ALGO_DEFLATE, (size_t)capb) + (capb >> 12) + (1u << 20);
        size_t nsl = 0;
        for (pna_gpu_stream *x : batch) nsl += x->slabs.size() + 1;
        if (rc0 == PNA_OK && (c->s_out[slot].ensure(std::max<uint64_t>(out_cap + 256 * groups.size(), cap_out)) || c->st_in[slot].ensure(std::max<uint64_t>(in_total + 64, capb + (1u << 20))) ||
                              c->st_out[slot].ensure(std::max<uint64_t>(bound_total + 64, cap_out)) || c->s_segs[slot].ensure(std::max<size_t>(nsl, 4096) * 24) ||
                              (page_bytes && c->hp_in[0].ensure(page_bytes + 64)))) rc0 = set_err(PNA_E_NOMEM, "staging allocation failed");
        struct LinkSegH { const uint8_t *src; uint8_t *dst; uint64_t len; };
        LinkSegH *lsg = (LinkSegH *)c->s_segs[slot].p; uint32_t nlsg = 0;
        uint64_t ppos = 0;
        for (Grp &g : groups) {
            for (size_t k = 0; k < g.st.size() && rc0 == PNA_OK; k++) {
                pna_gpu_stream *x = g.st[k];
                uint8_t *d = (uint8_t *)c->st_in[slot].p + g.in_base + g.off[k];
                if (x->pageable) {
                    if (!x->buf.empty()) {
                        big_memcpy((uint8_t *)c->hp_in[0].p + ppos, x->buf.data(), x->buf.size());
                        if (hipMemcpyAsync(d, (uint8_t *)c->hp_in[0].p + ppos, x->buf.size(), hipMemcpyHostToDevice, c->s_h2d) != hipSuccess) rc0 = set_err(PNA_E_HIP, "H2D copy failed");
                        ppos += (x->buf.size() + 15) & ~(size_t)15;
                    }
                } else {
                    for (size_t b = 0; b < x->slabs.size() && rc0 == PNA_OK; b++) {
                        const size_t nb = std::min(S_SLAB, x->slab_len - b * S_SLAB);
                        lsg[nlsg++] = LinkSegH{x->slabs[b], d + b * S_SLAB, nb};       // (the slabs are page-locked and device-mapped: one kernel reads them all)
                    }
                }
            }
        }
        if (rc0 == PNA_OK && nlsg) {
            if (c->tun.stream_gather_wgs) { launch_link_gather(lsg, nlsg, (uint32_t)c->tun.stream_gather_wgs, c->s_h2d); if (hipGetLastError() != hipSuccess) rc0 = set_err(PNA_E_HIP, "copy-in kernel failed"); }
            else for (uint32_t q = 0; q < nlsg && rc0 == PNA_OK; q++)
                if (hipMemcpyAsync(lsg[q].dst, lsg[q].src, lsg[q].len, hipMemcpyHostToDevice, c->s_h2d) != hipSuccess) rc0 = set_err(PNA_E_HIP, "H2D copy failed");
        }
        if (rc0 == PNA_OK && hipStreamSynchronize(c->s_h2d) != hipSuccess) rc0 = set_err(PNA_E_HIP, "H2D copy failed");
    }
    in_done();                                                           // the leader's role is free (the next batch is taken once this one is on the device)
    if (rc0 != PNA_OK) { in_failed(); fail_all(rc0); return; }          // (the batch never reaches the device: it only stops waiting -- device_busy belongs to the batch that IS there)
    const auto t1 = std::chrono::steady_clock::now();
    // ---- stage 2: the device batch (the context's kernels and workspaces: one at a time)
    auto t2 = t1;
    {
        std::lock_guard<std::mutex> run(c->run_mu);
        on_device();                                                     // the next leader may take its batch and copy it in beside this one's kernels
        t2 = std::chrono::steady_clock::now();
        (void)hipSetDevice(c->device);
        for (Grp &g : groups)
            g.rc = pna_gpu_compress_batch_device(c, g.st[0]->algo, g.st[0]->level, g.st.size(), (uint8_t *)c->st_in[slot].p + g.in_base, g.off.data(), g.len.data(),
                                                 (uint8_t *)c->st_out[slot].p + g.out_base, g.bound + 64, g.doff.data(), nullptr);
        off_device();
    }
    const auto t3 = std::chrono::steady_clock::now();
    // ---- stage 3: the streams travel back (the next batch's kernels are running by now)
    uint64_t hpos = 0;
    for (Grp &g : groups) {
        const size_t n = g.st.size();
        if (g.rc == PNA_OK && g.doff[n] && hipMemcpyAsync((uint8_t *)c->s_out[slot].p + hpos, (uint8_t *)c->st_out[slot].p + g.out_base, g.doff[n], hipMemcpyDeviceToHost, c->s_d2h) != hipSuccess)
            g.rc = set_err(PNA_E_HIP, "D2H copy failed");
        for (size_t k = 0; k < n; k++) {
            g.st[k]->out = (const uint8_t *)c->s_out[slot].p + hpos + (g.rc == PNA_OK ? g.doff[k] : 0);
            g.st[k]->out_len = g.rc == PNA_OK ? (size_t)(g.doff[k + 1] - g.doff[k]) : 0;
        }
        if (g.rc == PNA_OK) hpos += (g.doff[n] + 255) & ~(uint64_t)255;
    }
    bool ok = hipEventRecord(c->s_ev[slot], c->s_d2h) == hipSuccess && hipEventSynchronize(c->s_ev[slot]) == hipSuccess;
    for (Grp &g : groups) { if (!ok && g.rc == PNA_OK) g.rc = set_err(PNA_E_HIP, "device batch failed"); for (pna_gpu_stream *x : g.st) { x->rc = g.rc; if (g.rc != PNA_OK) x->out_len = 0; } }
    if (trace) {
        const auto t4 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        uint64_t inb = 0; for (Grp &g : groups) inb += g.in_bytes;
        fprintf(stderr, "[pna stream batch] slot %d, %zu entries, %.1f MiB in: copy in %.2f ms, wait for the device %.2f ms, device batch %.2f ms, copy out %.2f ms\n",
                slot, batch.size(), inb / 1048576.0, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4));
    }
}

extern "C" int pna_gpu_stream_finish(pna_gpu_stream *s) {
    if (!s) return PNA_E_INVAL;
    pna_gpu_ctx *c = s->ctx;
    {
        std::unique_lock<std::mutex> lk(c->comb_mu);
        c->comb_queue.push_back(s); s->queued = true;
        c->gate_cv.notify_one();                                     // (the leader may be waiting for the queue to grow)
        while (!s->done) {
            // (a stream the current leader left in the queue -- its batch was full -- waits for the leader's role like a new one)
            if (c->comb_leader || !s->queued) { c->comb_cv.wait(lk); continue; }
            c->comb_leader = true;                                   // s is still queued, so the batch taken below contains it
            const int slot = (int)(c->comb_seq++ % pna_gpu_ctx::S_SLOTS);
            // the batch three before this one is still being drained from this slot; a batch is copied in and waits for the device; the device is busy and
            // the queue is still small
            for (;;) {
                uint64_t qb = 0;
                for (pna_gpu_stream *x : c->comb_queue) qb += x->total();
                if (!c->slot_pending[slot] && !c->staged_waiting && (!c->device_busy || qb >= ((uint64_t)c->tun.stream_overlap_mib << 20))) break;
                c->gate_cv.wait(lk);
            }
            {   // a short linger lets the other writers of the pool reach their finish(): with T writers in flight the batches then hold ~T
                // entries instead of T / 2 (two alternating cohorts) -- 16 threads: 1.5 -> 2.7 GiB/s, 4: 0.40 -> 0.73, 64: 4.8 -> 5.4.
                // Adaptive default: 200 us (a few % of a batch's latency) once more than one writer has been seen, none for a lone writer
                // (second half of round 4: the linger ENDS as soon as the queue holds as many streams as the largest recent batch (comb_peak) -- the cohort has arrived --; a fixed 200 us
                // was right for 16 writers and too long for 64, whose cohort of 32 then missed the device's next turn: 64 / 128 writers 18 / 21 -> 20 / 23 GiB/s)
                const uint32_t lg = c->comb_linger_us != 0xFFFFFFFFu ? c->comb_linger_us : ((c->comb_last > 1 || c->comb_queue.size() > 1) ? 200u : 0u);
                if (lg) {
                    // with a batch on the device the linger ends early (the pipeline's regime: the cohort must not miss the device's next turn); with the device idle it runs its
                    // time -- that is where two small cohorts merge into one batch again (16 writers as 8 + 8: 9.2 GiB/s; as one cohort: 10)
                    const size_t want = (c->device_busy || c->staged_waiting) ? std::max<size_t>(c->comb_peak, 2) : (size_t)-1;
                    const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(lg);       // (system_clock: pthread_cond_timedwait, which the thread sanitizer of this toolchain knows; steady_clock waits go through pthread_cond_clockwait, which it does not)
                    while (c->comb_queue.size() < want && c->gate_cv.wait_until(lk, deadline) != std::cv_status::timeout) { }
                }
            }
            // the batch: the queue's streams in arrival order up to stream_batch_mib of input -- s itself always (it may be anywhere in the queue)
            std::vector<pna_gpu_stream *> batch, rest;
            {
                const uint64_t cap = (uint64_t)c->tun.stream_batch_mib << 20;
                uint64_t bytes = s->total();
                batch.push_back(s);
                for (pna_gpu_stream *x : c->comb_queue) {
                    if (x == s) continue;
                    if (bytes + x->total() <= cap) { batch.push_back(x); bytes += x->total(); } else rest.push_back(x);
                }
                c->comb_queue.swap(rest);
            }
            c->slot_pending[slot] = batch.size(); c->comb_last = batch.size(); c->comb_peak = std::max<size_t>(batch.size(), c->comb_peak ? c->comb_peak - 1 : 0);
            for (pna_gpu_stream *x : batch) { x->slot = slot; x->queued = false; }
            lk.unlock();
            stream_run_batch(c, batch, slot, [&]() { std::lock_guard<std::mutex> g(c->comb_mu); c->comb_leader = false; c->staged_waiting++; c->comb_cv.notify_all(); },
                             [&]() { std::lock_guard<std::mutex> g(c->comb_mu); c->staged_waiting--; c->device_busy = true; c->gate_cv.notify_one(); },
                             [&]() { std::lock_guard<std::mutex> g(c->comb_mu); c->device_busy = false; c->gate_cv.notify_one(); },
                             [&]() { std::lock_guard<std::mutex> g(c->comb_mu); c->staged_waiting--; c->gate_cv.notify_one(); });
            lk.lock();
            for (pna_gpu_stream *x : batch) x->done = true;          // owners may free their streams as soon as the lock is released
            c->comb_batches++; c->comb_entries += batch.size(); c->comb_max = std::max<uint64_t>(c->comb_max, batch.size());
            c->comb_cv.notify_all();
        }
    }
    pool_put(c, s->slabs);                                           // the input has been copied to the device
    int rc = s->rc;
    if (rc == PNA_OK) {
        // the reference's zstd writer drains in bursts of at most 32 KiB (zio::Writer); keep that shape
        for (size_t p = 0; p < s->out_len && rc == PNA_OK; p += 32768) {
            const size_t k = std::min<size_t>(32768, s->out_len - p);
            if (s->sink(s->user, s->out + p, k) != 0) { std::lock_guard<std::mutex> run(c->err_mu); rc = fail(c, PNA_E_SINK, "sink failed"); }
        }
    }
    {
        std::lock_guard<std::mutex> lk(c->comb_mu);
        if (--c->slot_pending[s->slot] == 0) c->gate_cv.notify_one();
    }
    delete s;
    return rc;
}
extern "C" int pna_gpu_stream_stats(pna_gpu_ctx *c, uint64_t *batches, uint64_t *entries, uint64_t *largest_batch) {
    if (!c) return PNA_E_INVAL;
    std::lock_guard<std::mutex> lk(c->comb_mu);
    if (batches) *batches = c->comb_batches;
    if (entries) *entries = c->comb_entries;
    if (largest_batch) *largest_batch = c->comb_max;
    return PNA_OK;
}

// Benchmark support: the reference's fan-out restated on host threads (cli/src/command/core.rs:496-537) over the streaming facade --
// `threads` workers take entries FIFO, each entry = stream_new / write (whole entry in one call, core.rs:900-902) / finish into a
// counting sink.  Returns the seconds spent; *out_bytes = compressed bytes seen by the sinks.
// ---- Archive::write_file / write_stream_entry (lib/src/archive/write.rs:276-299,730-777): FHED, extra + metadata chunks, the compressed
// stream as one FDAT chunk per encoder burst (ChunkStreamWriter::write, lib/src/chunk/write.rs:32-47), FEND; no fSIZ.
struct pna_gpu_entry_writer { pna_gpu_ctx *ctx; pna_gpu_stream *st; pna_sink_fn sink; void *user; uint32_t max_chunk; };
static int entry_writer_chunk(pna_gpu_entry_writer *w, const char ty[4], const uint8_t *data, size_t len) {
    uint8_t head[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len, (uint8_t)ty[0], (uint8_t)ty[1], (uint8_t)ty[2], (uint8_t)ty[3]};
    const uint32_t crc = pna_crc32(pna_crc32(0, ty, 4), data, len);
    const uint8_t tail[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    if (w->sink(w->user, head, 8) != 0 || (len && w->sink(w->user, data, len) != 0) || w->sink(w->user, tail, 4) != 0) return 1;
    return 0;
}
static int entry_writer_burst(void *u, const void *buf, size_t len) {       // one encoder burst -> FDAT chunk(s) of at most max_chunk bytes
    pna_gpu_entry_writer *w = (pna_gpu_entry_writer *)u;
    const uint8_t *p = (const uint8_t *)buf;
    while (len) {
        const size_t k = std::min<size_t>(len, w->max_chunk);
        if (entry_writer_chunk(w, "FDAT", p, k)) return 1;
        p += k; len -= k;
    }
    return 0;
}
extern "C" int pna_gpu_stream_entry_begin(pna_gpu_ctx *c, int algo, int level, const char *name, const void *meta, size_t meta_len,
                                          uint32_t max_chunk_size, pna_sink_fn sink, void *user, pna_gpu_entry_writer **out) {
    if (!c || !name || !sink || !out || (meta_len && !meta)) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "algorithm not implemented on the device path");
    if (meta_len && !meta_blob_ok((const uint8_t *)meta, meta_len)) return fail(c, PNA_E_INVAL, "extra / metadata chunks are not well-formed chunks");
    pna_gpu_entry_writer *w = new (std::nothrow) pna_gpu_entry_writer{c, nullptr, sink, user, max_chunk_size ? max_chunk_size : 0xFFFFFFFFu};
    if (!w) return fail(c, PNA_E_NOMEM, "out of memory");
    int rc = pna_gpu_stream_new(c, algo, level, entry_writer_burst, w, &w->st);
    if (rc) { delete w; return rc; }
    const std::vector<uint8_t> fh = frame_fhed_bytes(name, algo, 0, 1);           // cipher_mode CTR (1) when unencrypted, lib/src/entry/options.rs:156-159
    if (entry_writer_chunk(w, "FHED", fh.data(), fh.size()) || (meta_len && sink(user, meta, meta_len) != 0)) {
        pna_gpu_stream_abort(w->st); delete w; return fail(c, PNA_E_SINK, "sink failed");
    }
    *out = w;
    return PNA_OK;
}
extern "C" int pna_gpu_stream_entry_write(pna_gpu_entry_writer *w, const void *buf, size_t len) { return w ? pna_gpu_stream_write(w->st, buf, len) : PNA_E_INVAL; }
extern "C" int pna_gpu_stream_entry_finish(pna_gpu_entry_writer *w) {
    if (!w) return PNA_E_INVAL;
    int rc = pna_gpu_stream_finish(w->st);                                        // consumes the stream; the bursts went through entry_writer_burst
    if (rc == PNA_OK && entry_writer_chunk(w, "FEND", nullptr, 0)) rc = fail(w->ctx, PNA_E_SINK, "sink failed");
    delete w;
    return rc;
}
extern "C" void pna_gpu_stream_entry_abort(pna_gpu_entry_writer *w) { if (w) { pna_gpu_stream_abort(w->st); delete w; } }

static int counting_sink(void *user, const void *, size_t len) { ((std::atomic<uint64_t> *)user)->fetch_add(len, std::memory_order_relaxed); return 0; }
extern "C" double pna_bench_stream_threads(pna_gpu_ctx *c, int algo, int level, unsigned threads, size_t n, const void *const *src,
                                           const size_t *src_len, uint64_t *out_bytes, int *rc_out) {
    if (!c || !threads || (n && (!src || !src_len))) { if (rc_out) *rc_out = PNA_E_INVAL; return 0.0; }
    std::atomic<size_t> next{0}; std::atomic<uint64_t> total{0}; std::atomic<int> rc_all{PNA_OK};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back([&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) break;
                pna_gpu_stream *s = nullptr;
                int rc = pna_gpu_stream_new(c, algo, level, counting_sink, &total, &s);
                if (rc == PNA_OK) { rc = pna_gpu_stream_write(s, src[i], src_len[i]); if (rc != PNA_OK) pna_gpu_stream_abort(s); }
                if (rc == PNA_OK) rc = pna_gpu_stream_finish(s);
                if (rc != PNA_OK) { rc_all.store(rc); break; }
            }
        });
    for (auto &x : th) x.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out_bytes) *out_bytes = total.load();
    if (rc_out) *rc_out = rc_all.load();
    return secs;
}

extern "C" int pna_gpu_compress_solid(pna_gpu_ctx *c, int algo, int level, const void *src, size_t src_len,
                                      pna_sink_fn sink, void *user) {
    if (!c || !sink || (!src && src_len)) return PNA_E_INVAL;
    pna_gpu_stream *s = nullptr;
    int rc = pna_gpu_stream_new(c, algo, level, sink, user, &s);
    if (rc) return rc;
    rc = pna_gpu_stream_write(s, src, src_len);
    if (rc) { pna_gpu_stream_abort(s); return rc; }
    return pna_gpu_stream_finish(s);
}

