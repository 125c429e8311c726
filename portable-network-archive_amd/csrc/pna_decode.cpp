// pna_decode.cpp -- the decoders' host side: pna_gpu_decompress_batch[_device], open-size decode, the routing of large frames.
#include "pna_ctx.h"

// The parallel executor (k_zexec_par.hip) runs a frame / stream in WINDOWS of whole blocks, at most zexec_win_mib (1 024) MiB of output each (its words count 31 bits from the
// window's start): the cuts, from the blocks' output offsets (k_zoff / the chunk decoder's count pass have set them).  One window for everything up to 1 GiB.
static int zx_windows(pna_gpu_ctx *c, const ZxFrame &h, hipStream_t st, std::vector<uint32_t> &win_blk, std::vector<uint64_t> &win_off, uint64_t *max_win) {
    win_blk.assign(1, 0u); win_off.assign(1, 0ull);
    uint64_t mx = h.dst_len;
    const uint64_t WMAX = (uint64_t)c->tun.zexec_win_mib << 20;
    if (h.dst_len > WMAX) {
        std::vector<ZBlock> hb(h.nblk);
        HIPCHK(c, hipMemcpyAsync(hb.data(), (const ZBlock *)c->z_blocks.p + h.blk_base, (size_t)h.nblk * sizeof(ZBlock), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        mx = 0;
        uint64_t start = 0;
        for (uint32_t k = 0; k < h.nblk; k++) {
            const uint64_t rel = hb[k].out_off - h.dst_off, end = rel + hb[k].out_len;
            if (end - start > WMAX && rel > start) { mx = std::max(mx, rel - start); win_blk.push_back(k); win_off.push_back(rel); start = rel; }
        }
        mx = std::max(mx, h.dst_len - start);
    }
    win_blk.push_back(h.nblk); win_off.push_back(h.dst_len);
    *max_win = mx;
    return PNA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Read side, Compression::Deflate: one zlib stream per entry (flate2::read::ZlibDecoder, lib/src/entry/read.rs:178-179).
// k_inflate turns each stream into literals + (run, length, distance) records, k_zoff / k_zexec execute them, k_iadler_* check
// the Adler-32 trailer.
// A LARGE stream the lane-per-piece decoder cannot take (a foreign encoder's: no sync flush behind every 128 KiB -- what the reference itself writes for a large
// deflate entry): block starts found by trial (k_ispec), one wave per chunk between two of them (k_inflate's chunk mode: COUNT, prefix sums here, EMIT), one ZBlock per
// chunk.  *ok = false: something did not fit (no chunk starts found, a chunk's walk did not end where the next begins, sizes that do not add up) -- the serial walk
// takes the stream, nothing is lost but time.  The records are executed by k_zexec_par afterwards (the caller).
static constexpr uint32_t SPEC_CHUNK = 16384;
static int inflate_spec_stream(pna_gpu_ctx *c, uint32_t f, const ZFrame &fr, const ZFrameX &x, const void *d_src, hipStream_t st, uint32_t *nblk_out, bool *ok,
                               bool open, uint64_t *out_len) {
    *ok = false;
    const uint32_t nch = (uint32_t)((fr.src_len + SPEC_CHUNK - 1) / SPEC_CHUNK);
    if (nch < 4 || nch > x.blk_cap) return PNA_OK;
    if (c->z_spec.ensure((size_t)nch * (8 + sizeof(ISChunkH)) + 64)) return fail(c, PNA_E_NOMEM, "decoder workspace");
    uint64_t *d_start = (uint64_t *)c->z_spec.p;
    ISChunkH *d_chunks = (ISChunkH *)((uint8_t *)c->z_spec.p + (size_t)nch * 8);
    std::vector<uint64_t> start(nch);
    launch_ispec((const uint8_t *)d_src, fr.src_off, fr.src_len, SPEC_CHUNK, nch, d_start, st);
    HIPCHK(c, hipMemcpyAsync(start.data(), d_start, (size_t)nch * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    std::vector<ISChunkH> ch;
    for (uint32_t k = 0; k < nch; k++)
        if (start[k] != ~0ull) { ISChunkH h{}; h.start_bit = start[k]; h.end_bit = ~0ull; if (!ch.empty()) ch.back().end_bit = start[k]; ch.push_back(h); }
    uint32_t m = (uint32_t)ch.size();
    auto why = [&](const char *what, uint64_t a, uint64_t b) { if (c->tun.trace) fprintf(stderr, "[pna inflate] stream %u of %llu B not decoded in chunks: %s (%llu, %llu)\n", f, (unsigned long long)fr.src_len, what, (unsigned long long)a, (unsigned long long)b); };
    if (m < 4) { why("too few block starts found", m, nch); return PNA_OK; }                                      // (stored data, or blocks of more than a chunk each: not worth the two passes)
    // one walk over the chunks ch[idx[0 ..]] (idx empty: all of them): count pass (emit = 0) or emit pass
    auto run = [&](uint32_t emit, const std::vector<uint32_t> &idx) -> int {
        const uint32_t cnt = idx.empty() ? m : (uint32_t)idx.size();
        std::vector<ISChunkH> sub;
        if (!idx.empty()) { sub.resize(cnt); for (uint32_t i = 0; i < cnt; i++) sub[i] = ch[idx[i]]; }
        ISChunkH *hp = idx.empty() ? ch.data() : sub.data();
        HIPCHK(c, hipMemcpyAsync(d_chunks, hp, (size_t)cnt * sizeof(ISChunkH), hipMemcpyHostToDevice, st));
        launch_inflate_chunks((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, f, d_chunks, cnt, emit, (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p, (uint8_t *)c->z_lit.p,
                              (uint64_t *)c->z_seqs.p, st);
        HIPCHK(c, hipMemcpyAsync(hp, d_chunks, (size_t)cnt * sizeof(ISChunkH), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (!idx.empty()) for (uint32_t i = 0; i < cnt; i++) ch[idx[i]] = sub[i];
        return PNA_OK;
    };
    // A walk that runs past its chunk's end says the NEXT chunk's start was not a block start (a well-formed header by chance: one in a few 10^9 bit positions --
    // seen in a 445 MB stream, five of them in a 2 GB one): that chunk is dropped, its predecessor runs on to the start behind it and is counted again -- only it:
    // the other chunks' counts stand (second half of round 4; before, every repair was a walk over the whole stream, one false start at a time).  A dropped chunk's
    // own walk says nothing about the chunk behind it, which is judged in the next round by its new predecessor.
    int rc = PNA_OK;
    std::vector<uint32_t> todo;                                      // (empty: everything)
    for (int round = 0;; round++) {
        rc = run(0, todo); if (rc) return rc;
        std::vector<ISChunkH> keep; keep.reserve(m);
        todo.clear();
        bool prev_dropped = false;
        for (uint32_t k = 0; k < m; k++) {
            if (k > 0 && !prev_dropped && ch[k - 1].status == 4u /* IF_CHAIN */) {           // the predecessor (kept, its walk valid) ran past this chunk's start
                if (todo.empty() || todo.back() != (uint32_t)keep.size() - 1) todo.push_back((uint32_t)keep.size() - 1);
                prev_dropped = true;
                continue;
            }
            prev_dropped = false;
            keep.push_back(ch[k]);
        }
        if (todo.empty()) break;
        if (round >= 16) { why("too many false block starts", m, round); return PNA_OK; }
        ch.swap(keep); m = (uint32_t)ch.size();
        for (uint32_t i : todo) { ISChunkH &h = ch[i]; const uint64_t sb = h.start_bit; h = ISChunkH{}; h.start_bit = sb; h.end_bit = i + 1 < m ? ch[i + 1].start_bit : ~0ull; }
    }
    uint64_t lit = 0, out = 0, rec = 0;
    for (uint32_t k = 0; k < m; k++) {
        const ISChunkH &h = ch[k];
        if (h.status) { why("a chunk's walk failed: chunk, status", k, h.status); return PNA_OK; }
        if (k + 1 < m && h.end_found != ch[k + 1].start_bit) { why("the chain is broken behind chunk: end found, next start", h.end_found, ch[k + 1].start_bit); return PNA_OK; }   // a false start, or a stream this scheme does not fit
        lit += h.nlit; out += (uint64_t)h.nlit + h.mtot; rec += h.nrec;
    }
    if (open ? out > fr.dst_len : out != fr.dst_len) { why("sizes do not add up: output, expected", out, fr.dst_len); return PNA_OK; }   // (open: dst_len is the room)
    *out_len = out;
    if (rec > x.seq_cap) { why("more records than room: records, room", rec, x.seq_cap); return PNA_OK; }
    if ((ch[m - 1].end_found + 7) / 8 != fr.src_len) { why("the last chunk does not end with the stream: end bit, stream bytes", ch[m - 1].end_found, fr.src_len); return PNA_OK; }
    lit = out = rec = 0;
    for (uint32_t k = 0; k < m; k++) { ISChunkH &h = ch[k]; h.lit_base = lit; h.out_base = out; h.rec_base = rec; lit += h.nlit; out += (uint64_t)h.nlit + h.mtot; rec += h.nrec; }
    rc = run(1, std::vector<uint32_t>()); if (rc) return rc;
    for (uint32_t k = 0; k < m; k++) if (ch[k].status) { why("a chunk's second walk failed: chunk, status", k, ch[k].status); return PNA_OK; }
    *nblk_out = m; *ok = true;
    return PNA_OK;
}

static int inflate_batch_device(pna_gpu_ctx *c, size_t n, const void *d_src, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                                const uint64_t *dst_off, const uint64_t *raw_len, hipStream_t st, bool open = false, uint64_t *raw_out = nullptr) {
    if (n > 0x3FFFFFFFull) return fail(c, PNA_E_INVAL, "batch too large for one decode call");
    std::vector<ZFrame> frs(n);
    std::vector<ZFrameX> fxs(n);
    std::vector<uint32_t> cbase(n + 1);
    uint64_t nseq_cap = 0, out_span = 0, pieces = 0, nblk = 0;
    // Streams of known size go lane-per-piece (k_vinflate): a stream of at most BLK_SIZE decoded bytes is one piece, a larger one is taken
    // to consist of ceil(raw_len / BLK_SIZE) sync-flush delimited pieces of BLK_SIZE bytes each (what this library's encoder writes) --
    // k_imark / k_vinflate / k_vfin check that and leave every stream that does not fit to the wave-per-stream kernel.  Streams of
    // unknown size (`open`) take the wave-per-stream kernel directly.
    struct VPieceH { uint32_t frame, j; };
    std::vector<VPieceH> vp;
    // pieces per stream: from the size when it is known; for streams of unknown size (solid streams, entries without fSIZ) from a count of
    // the sync-flush markers (one pass + one small read-back): markers + 1 pieces, all but the last holding BLK_SIZE bytes
    std::vector<uint64_t> npc(n);
    uint64_t tot_pieces = 0;
    bool lanes = !c->tun.inflate_serial;
    // workgroups per stream for the marker scans: one per 256 KiB of the batch's longest stream (n x G bounded)
    uint64_t max_src = 0;
    for (size_t i = 0; i < n; i++) max_src = std::max<uint64_t>(max_src, src_len[i]);
    const uint32_t scan_g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(max_src >> 18, 1024), (1ull << 24) / std::max<size_t>(n, 1)));
    // large streams that may turn out to be a foreign encoder's (inflate_spec_stream): candidates by size; their chunks' blocks live behind the batch's own
    const uint64_t spec_min = (uint64_t)c->tun.zexec_par_min_mib << 20;
    std::vector<uint32_t> cand; std::vector<uint64_t> cand_base;
    if (lanes && open) {
        std::vector<uint32_t> cnt(n);
        if (c->z_pb.ensure(n * 4 + 8) || c->z_vp.ensure(n * 16 + 16)) return fail(c, PNA_E_NOMEM, "decoder workspace");
        HIPCHK(c, hipMemcpyAsync(c->z_vp.p, src_off, n * 8, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync((uint8_t *)c->z_vp.p + n * 8, src_len, n * 8, hipMemcpyHostToDevice, st));
        launch_icount((const uint8_t *)d_src, (const uint64_t *)c->z_vp.p, (const uint64_t *)((uint8_t *)c->z_vp.p + n * 8), (uint32_t)n, (uint32_t *)c->z_pb.p, scan_g, st);
        HIPCHK(c, hipMemcpyAsync(cnt.data(), c->z_pb.p, n * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        for (size_t i = 0; i < n; i++) { npc[i] = (uint64_t)cnt[i] + 1; if ((npc[i] - 1) * BLK_SIZE > raw_len[i]) npc[i] = 1; }   // more pieces than the room allows: not this library's layout
    } else
        for (size_t i = 0; i < n; i++) npc[i] = std::max<uint64_t>(1, (raw_len[i] + BLK_SIZE - 1) / BLK_SIZE);
    for (size_t i = 0; i < n; i++) tot_pieces += npc[i];
    // a handful of pieces is served better by the wave-per-stream walk (a lane needs ~110 ms for a 128 KiB piece, however few there are)
    if (tot_pieces < 1024) lanes = false;
    for (size_t i = 0; i < n; i++) {
        // streams of 4 GiB and more: decoded by pieces (this library's layout: a sync flush behind every 128 KiB); the wave-per-stream walk counts in 32 bits
        if ((raw_len[i] > 0xFFFFFFFFull || src_len[i] > 0xFFFFFFFFull) && !lanes) return fail(c, PNA_E_UNSUPPORTED, "zlib streams of 4 GiB and more are decoded by sync-flush delimited pieces only");
        frs[i] = ZFrame{src_off[i], dst_off[i], src_len[i], raw_len[i], 0, open ? ZF_OPEN : 0u};   // open: raw_len is a capacity
        ZFrameX &x = fxs[i];
        const uint64_t P = lanes ? npc[i] : 1;
        const uint64_t pcap = std::min<uint64_t>(raw_len[i], lanes ? BLK_SIZE : raw_len[i]) / 3 + (raw_len[i] >> 16) / P + 16;   // matches are >= 3 bytes; + literal-run splits (serial walk)
        if (nblk + P > 0x7FFFFFFFull) return fail(c, PNA_E_INVAL, "batch too large for one decode call");
        x.blk_base = (uint32_t)nblk; x.blk_cap = (uint32_t)P; x.slot_base = 0; x.slot_cap = 0; x.nblk = 0;
        x.seq_base = nseq_cap; x.seq_cap = (uint32_t)std::min<uint64_t>(P * pcap, 0x7FFFFFFFu); x.pcap = (uint32_t)std::min<uint64_t>(pcap, 0x7FFFFFFFu); x.pad = 0;
        nseq_cap += P * pcap;
        if (lanes) for (uint64_t j = 0; j < P; j++) vp.push_back(VPieceH{(uint32_t)i, (uint32_t)j});
        nblk += P;
        out_span = std::max<uint64_t>(out_span, dst_off[i] + raw_len[i]);
        cbase[i] = (uint32_t)pieces;
        pieces += (raw_len[i] + 65535) >> 16;
        if (pieces > 0xFFFFFFF0ull) return fail(c, PNA_E_INVAL, "batch too large for one decode call");
        // (open: raw_len is the room, the size comes out of the count; the stream must be worth it by its compressed size then)
        if (spec_min && (open ? src_len[i] >= spec_min / 4 : raw_len[i] >= spec_min) && src_len[i] < (1ull << 32) && src_len[i] >= 8ull * SPEC_CHUNK) cand.push_back((uint32_t)i);   // (output of any size: the executor works in windows)
    }
    cbase[n] = (uint32_t)pieces;
    for (uint32_t i : cand) { cand_base.push_back(nblk); nblk += (src_len[i] + SPEC_CHUNK - 1) / SPEC_CHUNK; if (nblk > 0x7FFFFFFFull) return fail(c, PNA_E_INVAL, "batch too large for one decode call"); }
    if (c->z_frames.ensure(n * sizeof(ZFrame)) || c->z_fx.ensure(n * sizeof(ZFrameX)) || c->z_blocks.ensure(nblk * sizeof(ZBlock)) ||
        c->z_lit.ensure(out_span + 64) || c->z_seqs.ensure(nseq_cap * 8 + 64) || c->z_cbase.ensure((n + 1) * 4) || c->z_apart.ensure(pieces * 8 + 8) ||
        c->z_mode.ensure(n * 4 + 8 + (lanes ? (size_t)n * scan_g * 4 : 0)) ||
        (lanes && (c->z_vp.ensure(vp.size() * 8 + 8) || c->z_pb.ensure((nblk + n) * 8 + 8))))
        return fail(c, PNA_E_NOMEM, "decoder workspace");
    std::vector<uint32_t> modes(n, 1u);                            // 1 = the wave-per-stream walk's (VM_SERIAL); the lane-per-piece decoder decides for itself when it runs
    if (!lanes) HIPCHK(c, hipMemcpyAsync(c->z_mode.p, modes.data(), n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->z_frames.p, frs.data(), n * sizeof(ZFrame), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->z_fx.p, fxs.data(), n * sizeof(ZFrameX), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->z_cbase.p, cbase.data(), (n + 1) * 4, hipMemcpyHostToDevice, st));
    if (lanes) HIPCHK(c, hipMemcpyAsync(c->z_vp.p, vp.data(), vp.size() * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    if (lanes) launch_vinflate((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, (uint32_t)n, c->z_vp.p, (uint32_t)vp.size(), (uint64_t *)c->z_pb.p, (uint32_t *)c->z_mode.p,
                               (uint32_t *)c->z_mode.p + n + 2, scan_g, (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p, (uint8_t *)c->z_lit.p, (uint64_t *)c->z_seqs.p, st);
    // ---- large foreign streams: chunks between block starts found by trial, walked side by side; what does not fit stays with the serial walk below
    std::vector<std::pair<uint32_t, ZFrameX>> spec; std::vector<uint64_t> spec_len;
    if (!cand.empty()) {
        if (lanes) { HIPCHK(c, hipMemcpyAsync(modes.data(), c->z_mode.p, n * 4, hipMemcpyDeviceToHost, st)); HIPCHK(c, hipStreamSynchronize(st)); }
        for (size_t k = 0; k < cand.size(); k++) {
            const uint32_t i = cand[k];
            if (modes[i] != 1u) continue;                              // this library's layout: the lane-per-piece decoder has it
            ZFrameX xs = fxs[i];
            xs.blk_base = (uint32_t)cand_base[k]; xs.blk_cap = (uint32_t)((src_len[i] + SPEC_CHUNK - 1) / SPEC_CHUNK); xs.nblk = 0; xs.pad = 0;
            HIPCHK(c, hipMemcpyAsync((ZFrameX *)c->z_fx.p + i, &xs, sizeof xs, hipMemcpyHostToDevice, st));
            bool ok = false; uint32_t m = 0; uint64_t olen = 0;
            const int rcs = inflate_spec_stream(c, i, frs[i], xs, d_src, st, &m, &ok, open, &olen);
            if (rcs) return rcs;
            static const uint32_t three = 3u;                           // (not 1: the serial walk leaves the stream alone)
            if (ok) {
                xs.nblk = m; xs.pad = 1; HIPCHK(c, hipMemcpyAsync((uint32_t *)c->z_mode.p + i, &three, 4, hipMemcpyHostToDevice, st));
                if (open) {                                             // the size found: what the serial walk reports through ZFrame::dst_len
                    frs[i].dst_len = olen;
                    HIPCHK(c, hipMemcpyAsync((uint8_t *)c->z_frames.p + (size_t)i * sizeof(ZFrame) + offsetof(ZFrame, dst_len), &frs[i].dst_len, 8, hipMemcpyHostToDevice, st));
                }
            }
            else xs = fxs[i];
            HIPCHK(c, hipMemcpyAsync((ZFrameX *)c->z_fx.p + i, &xs, sizeof xs, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipStreamSynchronize(st));                        // (xs is read by the copy until then)
            if (ok) { spec.emplace_back(i, xs); spec_len.push_back(olen); }
        }
    }
    launch_inflate((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, (uint32_t)n, (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p, (uint8_t *)c->z_lit.p,
                   (uint64_t *)c->z_seqs.p, (const uint32_t *)c->z_mode.p, st);
    HIPCHK(c, hipEventRecord(c->ev[2], st));
    if (lanes) launch_zexec_groups((ZFrame *)c->z_frames.p, (const ZFrameX *)c->z_fx.p, (uint32_t)n, (ZBlock *)c->z_blocks.p, c->z_vp.p, (uint32_t)vp.size(), (const uint8_t *)d_src,
                                   (const uint8_t *)c->z_lit.p, (const uint64_t *)c->z_seqs.p, (uint8_t *)d_dst, st);   // execution groups side by side (k_vfin)
    else launch_zexec((ZFrame *)c->z_frames.p, (const ZFrameX *)c->z_fx.p, (uint32_t)n, (ZBlock *)c->z_blocks.p, (const uint8_t *)d_src,
                      (const uint8_t *)c->z_lit.p, (const uint64_t *)c->z_seqs.p, (uint8_t *)d_dst, st);
    for (size_t si = 0; si < spec.size(); si++) {                        // their chunks' records: pointer jumping over the stream's output positions (k_zexec_par.hip)
        auto &sp = spec[si];
        const uint32_t i = sp.first;
        ZxFrame h{frs[i].dst_off, spec_len[si], sp.second.blk_base, sp.second.nblk, 0, 0};
        std::vector<uint32_t> wblk; std::vector<uint64_t> woff; uint64_t wmax = 0;
        { const int rw = zx_windows(c, h, st, wblk, woff, &wmax); if (rw) return rw; }
        if (c->z_words.ensure(wmax * 4 + 4096) || c->z_rep.ensure((size_t)h.nblk * 24 + 64) || c->z_zxf.ensure(64)) return fail(c, PNA_E_NOMEM, "decoder workspace");
        HIPCHK(c, hipMemcpyAsync(c->z_zxf.p, &h, sizeof h, hipMemcpyHostToDevice, st));
        uint32_t zst = 0, rounds = 0;
        if (launch_zexec_par((ZxFrame *)c->z_zxf.p, h, (const ZBlock *)c->z_blocks.p, (const uint8_t *)d_src, (const uint8_t *)c->z_lit.p, (uint64_t *)c->z_seqs.p,
                             (uint32_t *)c->z_rep.p, (uint32_t *)c->z_words.p, (uint8_t *)d_dst, &zst, &rounds, st, (uint32_t)wblk.size() - 1, wblk.data(), woff.data()) != 0) return fail(c, PNA_E_HIP, "parallel stream execution failed");
        c->zexec_par_rounds = rounds;
        if (zst) { static const uint32_t corrupt = 1u; HIPCHK(c, hipMemcpyAsync((uint8_t *)c->z_frames.p + (size_t)i * sizeof(ZFrame) + offsetof(ZFrame, status), &corrupt, 4, hipMemcpyHostToDevice, st)); }
    }
    c->inflate_spec_streams = (uint32_t)spec.size();
    HIPCHK(c, hipEventRecord(c->ev[3], st));
    launch_iadler((ZFrame *)c->z_frames.p, (const ZFrameX *)c->z_fx.p, (const ZBlock *)c->z_blocks.p, (uint32_t)n, (const uint32_t *)c->z_cbase.p,
                  (uint32_t)pieces, (const uint8_t *)d_dst, c->z_apart.p, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, n * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
    if (lanes) HIPCHK(c, hipStreamSynchronize(st));          // (vp is read by the copy above until then)
    HIPCHK(c, hipStreamSynchronize(st));
    float ms = 0, ms_h = 0, ms_x = 0;
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); (void)hipEventElapsedTime(&ms_h, c->ev[0], c->ev[2]); (void)hipEventElapsedTime(&ms_x, c->ev[2], c->ev[3]);
    c->timing = pna_gpu_timing{}; c->timing.ms_lz = ms; c->timing.ms_stats = ms_h; c->timing.ms_lit = ms_x;   // total, Huffman walk, execution
    c->timing.lz_match_launches = spec.size();                      // (decode calls: the large foreign streams that went through the chunk decoder)
    for (size_t i = 0; i < n; i++)
        if (frs[i].status) {
            char msg[160];
            snprintf(msg, sizeof msg, "entry %zu: %s (produced %u of %llu bytes)", i,
                     frs[i].status == 2 ? "unsupported stream" : (frs[i].status == 3 ? "size mismatch" : "corrupt stream"), frs[i].out_len, (unsigned long long)frs[i].dst_len);
            return fail(c, frs[i].status == 2 ? PNA_E_UNSUPPORTED : PNA_E_INVAL, msg);
        }
    if (open && raw_out) for (size_t i = 0; i < n; i++) raw_out[i] = frs[i].dst_len;
    return PNA_OK;
}

// A zlib stream whose decoded size is recorded nowhere (deflate entries without fSIZ, deflate solid streams): decoded into dst_cap
// bytes of room, the size found is reported (PNA_E_INVAL when it does not fit).
extern "C" int pna_gpu_inflate_open_device(pna_gpu_ctx *c, const void *d_src, uint64_t src_off, uint64_t src_len, void *d_dst, uint64_t dst_off,
                                           uint64_t dst_cap, uint64_t *raw_len, void *hip_stream) {
    if (!c || !d_src || !d_dst || !raw_len) return fail(c, PNA_E_INVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return inflate_batch_device(c, 1, d_src, &src_off, &src_len, d_dst, &dst_off, &dst_cap, st, true, raw_len);
}

// ---------------------------------------------------------------------------------------------------------
// Read side: decompress_reader (lib/src/entry/read.rs:171-190); entries already in device memory.
static int zstd_decode_device(pna_gpu_ctx *c, size_t n, const void *d_src, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                              const uint64_t *dst_off, const uint64_t *raw_len, bool open, uint64_t *raw_out, hipStream_t st, bool allow_foreign = true);

extern "C" int pna_gpu_decompress_batch_device(pna_gpu_ctx *c, int algo, size_t n, const void *d_src, const uint64_t *src_off,
                                               const uint64_t *src_len, void *d_dst, const uint64_t *dst_off, const uint64_t *raw_len,
                                               void *hip_stream) {
    if (!c || (n && (!d_src || !src_off || !src_len || !d_dst || !dst_off || !raw_len))) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "only zstd and deflate streams are decoded on the device");
    if (!n) return PNA_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (algo == PNA_ALGO_DEFLATE) return inflate_batch_device(c, n, d_src, src_off, src_len, d_dst, dst_off, raw_len, st);
    return zstd_decode_device(c, n, d_src, src_off, src_len, d_dst, dst_off, raw_len, false, nullptr, st);
}

// A zstd stream whose decoded size is not recorded anywhere (the SDAT stream of a solid entry: SHED carries no size): step 1 counts
// its frames, the caller provides frames x 1 MiB (this library's segmentation; one frame of any size: `cap` bytes), step 2 decodes
// and reports the size found.
extern "C" int pna_gpu_zstd_stream_frames_device(pna_gpu_ctx *c, const void *d_src, uint64_t src_off, uint64_t src_len, uint32_t *n_frames, void *hip_stream) {
    if (!c || !d_src || !n_frames) return fail(c, PNA_E_INVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (c->z_ents.ensure(sizeof(ZEntry)) || c->z_work.ensure(64)) return fail(c, PNA_E_NOMEM, "decoder workspace");
    const ZEntry en{src_off, src_len, 0, 0, 0, 0, 1, 0};
    HIPCHK(c, hipMemcpyAsync(c->z_ents.p, &en, sizeof en, hipMemcpyHostToDevice, st));
    launch_zcount((const ZEntry *)c->z_ents.p, 1, (const uint8_t *)d_src, (uint32_t *)c->z_work.p, st);
    HIPCHK(c, hipMemcpyAsync(n_frames, c->z_work.p, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (*n_frames == 0 && src_len) return fail(c, PNA_E_INVAL, "not a sequence of zstd frames");
    return PNA_OK;
}
extern "C" int pna_gpu_zstd_decompress_open_device(pna_gpu_ctx *c, const void *d_src, uint64_t src_off, uint64_t src_len, void *d_dst, uint64_t dst_off,
                                                   uint64_t dst_cap, uint64_t *raw_len, void *hip_stream) {
    if (!c || !d_src || !d_dst || !raw_len) return fail(c, PNA_E_INVAL, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return zstd_decode_device(c, 1, d_src, &src_off, &src_len, d_dst, &dst_off, &dst_cap, true, raw_len, st);
}

// A payload k_zscan could not place -- frames of other sizes than this library's grid, skippable frames between them: anything zstd::stream::read::Decoder
// reads (lib/src/entry/read.rs:171-190) --: its frames are listed (k_zlist), every run of frames whose headers carry a content size is decoded as one batch of
// single-frame entries (the pipeline above, side by side), a frame without one on its own with an open size (its content's length is only known once it is
// decoded), one after the other.  `room` = the entry's raw length (open: its capacity); *found = the bytes produced.
static int zstd_decode_foreign(pna_gpu_ctx *c, const void *d_src, uint64_t src_off, uint64_t src_len, void *d_dst, uint64_t dst_off, uint64_t room, bool open,
                               uint64_t *found, hipStream_t st) {
    struct Item { uint64_t off, len, fcs; };
    constexpr uint32_t CAP = 4096;
    if (c->z_list.ensure(CAP * sizeof(Item) + 64)) return fail(c, PNA_E_NOMEM, "decoder workspace");
    uint64_t *d_hdr = (uint64_t *)((uint8_t *)c->z_list.p + CAP * sizeof(Item));
    std::vector<Item> items(CAP);
    uint64_t ip = 0, produced = 0;
    for (;;) {
        uint64_t hdr[3] = {0, 0, 0};
        launch_zlist((const uint8_t *)d_src, src_off, src_len, ip, c->z_list.p, CAP, d_hdr, st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hdr, d_hdr, sizeof hdr, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (hdr[2]) return fail(c, PNA_E_INVAL, "corrupt stream (not a sequence of zstd frames)");
        const size_t k = (size_t)hdr[0];
        if (k) { HIPCHK(c, hipMemcpyAsync(items.data(), c->z_list.p, k * sizeof(Item), hipMemcpyDeviceToHost, st)); HIPCHK(c, hipStreamSynchronize(st)); }
        for (size_t a = 0; a < k;) {
            if (items[a].fcs != ~0ull) {                                  // a run of frames that say what they hold: one batch
                size_t b = a; uint64_t pos = produced;
                std::vector<uint64_t> so, sl, dof, rl;
                while (b < k && items[b].fcs != ~0ull) {
                    if (items[b].fcs > room - pos) return fail(c, PNA_E_INVAL, "size mismatch: the frames hold more than the entry's size");
                    so.push_back(items[b].off); sl.push_back(items[b].len); dof.push_back(dst_off + pos); rl.push_back(items[b].fcs); pos += items[b].fcs; b++;
                }
                int rc = zstd_decode_device(c, so.size(), d_src, so.data(), sl.data(), d_dst, dof.data(), rl.data(), false, nullptr, st, false);
                if (rc) return rc;
                produced = pos; a = b;
            } else {                                                      // no content size in the header: decoded with an open size
                uint64_t cap = room - produced, got = 0, dof = dst_off + produced;
                int rc = zstd_decode_device(c, 1, d_src, &items[a].off, &items[a].len, d_dst, &dof, &cap, true, &got, st, false);
                if (rc) return rc;
                produced += got; a++;
            }
        }
        ip = hdr[1];
        if (ip >= src_len) break;
        if (k == 0) return fail(c, PNA_E_INVAL, "corrupt stream");        // (no progress: cannot happen with hdr[2] == 0)
    }
    if (!open && produced != room) return fail(c, PNA_E_INVAL, "size mismatch (the frames do not add up to the entry's size)");
    *found = produced;
    return PNA_OK;
}

static int zstd_decode_device(pna_gpu_ctx *c, size_t n, const void *d_src, const uint64_t *src_off, const uint64_t *src_len, void *d_dst,
                              const uint64_t *dst_off, const uint64_t *raw_len, bool open, uint64_t *raw_out, hipStream_t st, bool allow_foreign) {
    std::vector<ZEntry> ents(n);
    uint64_t nfr = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t k = raw_len[i] ? (raw_len[i] + SEG_SIZE - 1) / SEG_SIZE : 1;
        if (nfr + k > 0x7FFFFFFFull) return fail(c, PNA_E_INVAL, "too many frames");
        ents[i] = ZEntry{src_off[i], src_len[i], dst_off[i], raw_len[i], (uint32_t)nfr, (uint32_t)k, open ? 1u : 0u, 0u};
        nfr += k;
    }
    // per-frame bounds of the lane-parallel pipeline (frames that exceed them fall back to the one-workgroup-per-frame kernel)
    std::vector<ZFrameX> fxs(nfr);
    uint64_t nblk_cap = 0, nslot = 0, nseq_cap = 0, out_span = 0;
    for (size_t i = 0; i < n; i++) {
        out_span = std::max<uint64_t>(out_span, dst_off[i] + raw_len[i]);
        for (uint32_t f = 0; f < ents[i].n_frames; f++) {
            const uint64_t done = (uint64_t)f * SEG_SIZE;
            const uint64_t dl = (f + 1 == ents[i].n_frames) ? (raw_len[i] > done ? raw_len[i] - done : 0) : SEG_SIZE;
            ZFrameX &x = fxs[ents[i].first_frame + f];
            x.blk_base = (uint32_t)nblk_cap; x.blk_cap = (uint32_t)std::min<uint64_t>((dl >> 12) + 4, 1u << 20);
            x.slot_base = (uint32_t)nslot; x.slot_cap = (uint32_t)std::min<uint64_t>((dl >> 17) + 2, 1u << 16);
            x.seq_base = nseq_cap; x.seq_cap = (uint32_t)std::min<uint64_t>(dl / 4 + 16, 0x7FFFFFFFu); x.nblk = 0;
            nblk_cap += x.blk_cap; nslot += x.slot_cap; nseq_cap += x.seq_cap;
            if (nblk_cap > 0x3FFFFFFFull) return fail(c, PNA_E_INVAL, "batch too large for one decode call");
        }
    }
    const bool serial_only = c->tun.zdec_serial != 0;                  // diagnostics: one workgroup per frame for everything
    if (c->z_ents.ensure(n * sizeof(ZEntry)) || c->z_frames.ensure(nfr * sizeof(ZFrame)) || c->z_lit.ensure(out_span + 64) ||
        c->z_fx.ensure(nfr * sizeof(ZFrameX)) || c->z_blocks.ensure(nblk_cap * sizeof(ZBlock)) || c->z_tabs.ensure(nslot * sizeof(ZTables)) ||
        c->z_seqs.ensure(nseq_cap * 8 + 64) || c->z_hlist.ensure(nblk_cap * 16 + 16) || c->z_slist.ensure(nblk_cap * 4 + 16) || c->z_work.ensure(64))
        return fail(c, PNA_E_NOMEM, "decoder workspace");
    HIPCHK(c, hipMemcpyAsync(c->z_ents.p, ents.data(), n * sizeof(ZEntry), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->z_fx.p, fxs.data(), nfr * sizeof(ZFrameX), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(c->z_work.p, 0, 64, st));
    launch_zscan((const ZEntry *)c->z_ents.p, (uint32_t)n, (const uint8_t *)d_src, (ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, st);
    HIPCHK(c, hipEventRecord(c->ev[0], st));
    std::vector<ZFrame> frs(nfr);
    if (!serial_only) {
        // Large frames (the reference writes ONE frame per entry whatever its size): k_zscan has found them -- a frame whose content takes zexec_par_min_mib
        // and more (below 2 GiB: the parallel executor's words hold 31-bit positions).  Their blocks are PARSED side by side (k_zparse_a: the header walk,
        // k_zparse<true>: a wave per block for the tables) and their sequences EXECUTED in parallel by pointer jumping (k_zexec_par.hip) instead of by one
        // wave each; the per-frame kernels skip them (ZFrameX::pad).
        std::vector<uint32_t> big;
        std::vector<ZFrameX> fxd;
        const uint64_t big_min = (uint64_t)c->tun.zexec_par_min_mib << 20;
        if (c->tun.zexec_par_min_mib > 0) {
            bool any = false;
            for (size_t i = 0; i < n && !any; i++) any = raw_len[i] >= big_min;
            if (any) {
                HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, nfr * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                for (uint64_t f = 0; f < nfr; f++)
                    if (frs[f].status == 0 && frs[f].dst_len >= big_min) big.push_back((uint32_t)f);            // (any size: the executor works in windows of 1 GiB)
                if (!big.empty()) {
                    if (c->z_big.ensure(big.size() * 4 + 64) || c->z_one.ensure(nblk_cap * 4 + 64)) return fail(c, PNA_E_NOMEM, "decoder workspace");
                    static const uint32_t one = 1;                        // (static: the copy is asynchronous, the source must outlive this scope)
                    for (uint32_t f : big) HIPCHK(c, hipMemcpyAsync((uint8_t *)c->z_fx.p + (size_t)f * sizeof(ZFrameX) + offsetof(ZFrameX, pad), &one, 4, hipMemcpyHostToDevice, st));
                    HIPCHK(c, hipMemcpyAsync(c->z_big.p, big.data(), big.size() * 4, hipMemcpyHostToDevice, st));
                }
            }
        }
        // sequence records of frame f start at seq_base: k_zparse adds it to the block's running count
        launch_zparse((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, (uint32_t)nfr, (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p, (ZTables *)c->z_tabs.p,
                      (uint32_t *)c->z_hlist.p, (uint32_t *)c->z_slist.p, c->z_work.p, st);
        uint32_t work[4] = {0, 0, 0, 0};
        if (!big.empty()) {
            launch_zparse_big_a((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, (const uint32_t *)c->z_big.p, (uint32_t)big.size(), (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p,
                                (uint32_t *)c->z_one.p, c->z_work.p, st);
            fxd.resize(nfr);
            HIPCHK(c, hipMemcpyAsync(work, c->z_work.p, 16, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipMemcpyAsync(fxd.data(), c->z_fx.p, nfr * sizeof(ZFrameX), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            launch_zparse_big_b((ZFrame *)c->z_frames.p, (ZFrameX *)c->z_fx.p, work[2], (const uint8_t *)d_src, (ZBlock *)c->z_blocks.p, (ZTables *)c->z_tabs.p,
                                (uint32_t *)c->z_hlist.p, (uint32_t *)c->z_slist.p, c->z_work.p, (const uint32_t *)c->z_one.p, st);
        }
        HIPCHK(c, hipMemcpyAsync(work, c->z_work.p, 16, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        launch_zstreams(work[0], work[1], (const uint32_t *)c->z_hlist.p, (const uint32_t *)c->z_slist.p, c->z_work.p, (ZBlock *)c->z_blocks.p,
                        (const ZFrame *)c->z_frames.p, (const ZTables *)c->z_tabs.p, (const uint8_t *)d_src, (uint8_t *)c->z_lit.p, (uint64_t *)c->z_seqs.p, st);
        launch_zexec((ZFrame *)c->z_frames.p, (const ZFrameX *)c->z_fx.p, (uint32_t)nfr, (ZBlock *)c->z_blocks.p, (const uint8_t *)d_src,
                     (const uint8_t *)c->z_lit.p, (const uint64_t *)c->z_seqs.p, (uint8_t *)d_dst, st);
        if (!big.empty()) {
            HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, nfr * sizeof(ZFrame), hipMemcpyDeviceToHost, st));      // (k_zoff has fixed the sizes of open frames)
            HIPCHK(c, hipStreamSynchronize(st));
            for (uint32_t f : big) {
                if (frs[f].status) continue;
                ZxFrame h{frs[f].dst_off, frs[f].dst_len, fxd[f].blk_base, fxd[f].nblk, 0, 0};
                std::vector<uint32_t> wblk; std::vector<uint64_t> woff; uint64_t wmax = 0;
                { const int rw = zx_windows(c, h, st, wblk, woff, &wmax); if (rw) return rw; }
                if (c->z_words.ensure(wmax * 4 + 4096) || c->z_rep.ensure((size_t)h.nblk * 24 + 64) || c->z_zxf.ensure(64)) return fail(c, PNA_E_NOMEM, "decoder workspace");
                HIPCHK(c, hipMemcpyAsync(c->z_zxf.p, &h, sizeof h, hipMemcpyHostToDevice, st));
                uint32_t zst = 0, rounds = 0;
                if (launch_zexec_par((ZxFrame *)c->z_zxf.p, h, (const ZBlock *)c->z_blocks.p, (const uint8_t *)d_src, (const uint8_t *)c->z_lit.p, (uint64_t *)c->z_seqs.p,
                                     (uint32_t *)c->z_rep.p, (uint32_t *)c->z_words.p, (uint8_t *)d_dst, &zst, &rounds, st, (uint32_t)wblk.size() - 1, wblk.data(), woff.data()) != 0) return fail(c, PNA_E_HIP, "parallel frame execution failed");
                c->zexec_par_rounds = rounds;
                if (zst) {                                            // 2: the serial kernel takes the frame (it decodes from the source again); 3: corrupt
                    static const uint32_t codes[2] = {1u, 2u};            // (static: the copy is asynchronous)
                    HIPCHK(c, hipMemcpyAsync((uint8_t *)c->z_frames.p + (size_t)f * sizeof(ZFrame) + offsetof(ZFrame, status), &codes[zst == 2 ? 1 : 0], 4, hipMemcpyHostToDevice, st));
                }
            }
        }
        launch_zxxh((ZFrame *)c->z_frames.p, (uint32_t)nfr, (const uint8_t *)d_src, (const uint8_t *)d_dst, st);   // frames that carry a content checksum
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, nfr * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    } else {
        HIPCHK(c, hipMemcpyAsync(frs.data(), c->z_frames.p, nfr * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        for (auto &fr : frs) if (fr.status == 0) fr.status = 2;        // route every well-formed frame through the fallback below
    }
    // ---- frames the bounded pipeline could not take: one workgroup per frame
    std::vector<uint32_t> fb;
    for (uint64_t f = 0; f < nfr; f++) if (frs[f].status == 2) fb.push_back((uint32_t)f);
    if (c->tun.zdec_fallback_max_mib > 0)
        for (uint32_t f : fb) if (frs[f].dst_len > ((uint64_t)c->tun.zdec_fallback_max_mib << 20)) {
            char msg[160];
            snprintf(msg, sizeof msg, "frame %u: %llu bytes of content would be decoded by one workgroup (option zdec_fallback_max_mib)", f, (unsigned long long)frs[f].dst_len);
            return fail(c, PNA_E_UNSUPPORTED, msg);
        }
    if (!fb.empty()) {
        std::vector<ZFrame> sub(fb.size());
        for (size_t k = 0; k < fb.size(); k++) { sub[k] = frs[fb[k]]; sub[k].status = 0; sub[k].out_len = 0; }
        if (open)                                                 // the frame that closes a stream of unknown size keeps its flag
            for (size_t k = 0; k < fb.size(); k++)
                for (size_t i = 0; i < n; i++) {
                    const uint32_t f0 = ents[i].first_frame, f1 = f0 + ents[i].n_frames;
                    if (fb[k] >= f0 && fb[k] < f1 && (fb[k] + 1 == f1 || (fb[k] == f0 && f1 - f0 > 1 && frs[f0 + 1].status == 4))) sub[k].out_len = ZF_OPEN;
                }
        if (c->z_fb.ensure(sub.size() * sizeof(ZFrame)) || c->z_lit.ensure(std::max<uint64_t>(out_span + 64, sub.size() * (uint64_t)(128u << 10) + 64)))
            return fail(c, PNA_E_NOMEM, "decoder workspace");
        HIPCHK(c, hipMemcpyAsync(c->z_fb.p, sub.data(), sub.size() * sizeof(ZFrame), hipMemcpyHostToDevice, st));
        launch_zdec((ZFrame *)c->z_fb.p, (uint32_t)sub.size(), (const uint8_t *)d_src, (uint8_t *)d_dst, (uint8_t *)c->z_lit.p, (uint32_t)c->tun.zdec_dbg, st);
        launch_zxxh((ZFrame *)c->z_fb.p, (uint32_t)sub.size(), (const uint8_t *)d_src, (const uint8_t *)d_dst, st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(sub.data(), c->z_fb.p, sub.size() * sizeof(ZFrame), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        for (size_t k = 0; k < fb.size(); k++) frs[fb[k]] = sub[k];
    }
    HIPCHK(c, hipEventRecord(c->ev[1], st));
    HIPCHK(c, hipStreamSynchronize(st));
    float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    c->timing = pna_gpu_timing{}; c->timing.ms_lz = ms;            // decoder time reported in the first stage slot
    std::vector<uint64_t> foreign_len(n, ~0ull);                  // entries that went through zstd_decode_foreign: the bytes they produced
    for (size_t i = 0; i < n; i++)
        for (uint32_t f = 0; f < ents[i].n_frames; f++) {
            const ZFrame &fr = frs[ents[i].first_frame + f];
            if (allow_foreign && (fr.status == 1 || fr.status == 3)) {
                // not one of the two shapes k_zscan places (or a frame of the grid walk did not hold its MiB): the payload's frames as they are
                uint64_t got = 0;
                const int rcf = zstd_decode_foreign(c, d_src, src_off[i], src_len[i], d_dst, dst_off[i], raw_len[i], open, &got, st);
                if (rcf) return rcf;
                foreign_len[i] = got;
                break;
            }
            if (fr.status && fr.status != 4) {                    // 4: void slot behind a single frame that holds the whole entry
                char msg[160];
                snprintf(msg, sizeof msg, "entry %zu frame %u: %s (produced %u of %llu bytes)", i, f,
                         fr.status == 2 ? "unsupported stream" : (fr.status == 3 ? "size mismatch (foreign multi-frame stream?)" : "corrupt stream"), fr.out_len, (unsigned long long)fr.dst_len);
                return fail(c, fr.status == 2 ? PNA_E_UNSUPPORTED : PNA_E_INVAL, msg);
            }
        }
    if (open && raw_out)
        for (size_t i = 0; i < n; i++) {                          // sizes found by the decoder: frames in front hold SEG_SIZE each
            if (foreign_len[i] != ~0ull) { raw_out[i] = foreign_len[i]; continue; }
            uint64_t total = 0;
            for (uint32_t f = 0; f < ents[i].n_frames; f++) { const ZFrame &fr = frs[ents[i].first_frame + f]; if (fr.status != 4) total += fr.dst_len; }
            raw_out[i] = total;
        }
    return PNA_OK;
}

// The same for payloads in host memory (extract / verify of an archive read from disk).
extern "C" int pna_gpu_decompress_batch(pna_gpu_ctx *c, int algo, size_t n, const void *const *src, const size_t *src_len,
                                        void *const *dst, const size_t *raw_len) {
    if (!c || (n && (!src || !src_len || !dst || !raw_len))) return fail(c, PNA_E_INVAL, "null argument");
    if (algo != PNA_ALGO_ZSTD && algo != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "only zstd and deflate streams are decoded on the device");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> so(n), sl(n), dof(n), rl(n);
    uint64_t sp = 0, dp = 0;
    for (size_t i = 0; i < n; i++) { so[i] = sp; sl[i] = src_len[i]; sp = (sp + src_len[i] + 15) & ~(uint64_t)15; dof[i] = dp; rl[i] = raw_len[i]; dp = (dp + raw_len[i] + 15) & ~(uint64_t)15; }
    if (c->stage_in.ensure(sp + 64) || c->stage_out.ensure(dp + 64)) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    for (size_t i = 0; i < n; i++) if (src_len[i]) HIPCHK(c, hipMemcpyAsync((uint8_t *)c->stage_in.p + so[i], src[i], src_len[i], hipMemcpyHostToDevice, c->stream));
    int rc = pna_gpu_decompress_batch_device(c, algo, n, c->stage_in.p, so.data(), sl.data(), c->stage_out.p, dof.data(), rl.data(), nullptr);
    if (rc) return rc;
    for (size_t i = 0; i < n; i++) if (raw_len[i]) HIPCHK(c, hipMemcpyAsync(dst[i], (uint8_t *)c->stage_out.p + dof[i], raw_len[i], hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PNA_OK;
}

