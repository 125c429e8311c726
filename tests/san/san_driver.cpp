// tests/san/san_driver.cpp -- TEST INFRASTRUCTURE: drives the C ABI of libpna_gpu.so's HOST code (built against the CPU HIP shim and the
// device stub) under AddressSanitizer / UBSan / ThreadSanitizer: the container writer, sanitize, split / join, the password hashes, the
// batch call, the CompressionWriter facade from many threads (group commit, page-locked slab pool), the bounded host pipeline with its
// stager thread, the streaming entry writer, append.  Exit code 0 = every check passed and the sanitizer stayed silent.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "../../include/pna_gpu.h"
#include "../../include/pna_archive.h"

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); g_fail++; } } while (0)

typedef std::vector<uint8_t> Bytes;
static int vec_sink(void *u, const void *b, size_t n) { Bytes *v = (Bytes *)u; v->insert(v->end(), (const uint8_t *)b, (const uint8_t *)b + n); return 0; }

static Bytes text(size_t n, uint32_t seed) {                     // compressible-looking filler (the stub stores it anyway)
    Bytes v(n); uint32_t s = seed * 2654435761u + 1;
    for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; v[i] = (uint8_t)("etaoin shrdlu\n"[(s >> 24) % 14]); }
    return v;
}
// decoder for what the device stub writes: frames of RAW blocks (and the 9-byte empty frame)
static bool unraw(const uint8_t *p, size_t n, Bytes &out) {
    size_t pos = 0;
    while (pos < n) {
        if (n - pos < 5 || p[pos] != 0x28 || p[pos + 1] != 0xB5 || p[pos + 2] != 0x2F || p[pos + 3] != 0xFD) return false;
        pos += 4;
        if (p[pos] == 0x20) pos += 2; else if (p[pos] == 0x00) pos += 2; else return false;
        for (;;) {
            if (n - pos < 3) return false;
            const uint32_t h = p[pos] | (p[pos + 1] << 8) | (p[pos + 2] << 16); pos += 3;
            const uint32_t ty = (h >> 1) & 3, sz = h >> 3;
            if (ty != 0 || n - pos < sz) return false;
            out.insert(out.end(), p + pos, p + pos + sz); pos += sz;
            if (h & 1) break;
        }
    }
    return true;
}
struct Chunk { char ty[5]; size_t off, len; };
static bool walk(const Bytes &a, std::vector<Chunk> &out) {      // every chunk's CRC is checked
    if (a.size() < 8) return false;
    size_t pos = 8;
    while (pos < a.size()) {
        if (a.size() - pos < 12) return false;
        const size_t l = ((size_t)a[pos] << 24) | (a[pos + 1] << 16) | (a[pos + 2] << 8) | a[pos + 3];
        if (a.size() - pos - 12 < l) return false;
        Chunk c; memcpy(c.ty, &a[pos + 4], 4); c.ty[4] = 0; c.off = pos + 8; c.len = l;
        const uint32_t crc = pna_crc32(0, &a[pos + 4], 4 + l);
        const uint32_t st = ((uint32_t)a[pos + 8 + l] << 24) | (a[pos + 9 + l] << 16) | (a[pos + 10 + l] << 8) | a[pos + 11 + l];
        if (crc != st) return false;
        out.push_back(c); pos += 12 + l;
    }
    return true;
}
// entries of an archive image: (name, decoded data) for FHED .. FEND records written by the (stubbed) zstd path
static bool read_back(const Bytes &arc, std::vector<std::pair<std::string, Bytes>> &ents) {
    std::vector<Chunk> ch; if (!walk(arc, ch)) return false;
    std::string name; Bytes pay; bool open = false;
    for (const Chunk &c : ch) {
        if (!strcmp(c.ty, "FHED")) { name.assign((const char *)&arc[c.off + 6], c.len - 6); pay.clear(); open = true; }
        else if (!strcmp(c.ty, "FDAT") && open) pay.insert(pay.end(), arc.begin() + c.off, arc.begin() + c.off + c.len);
        else if (!strcmp(c.ty, "FEND") && open) { Bytes d; if (!unraw(pay.data(), pay.size(), d)) return false; ents.emplace_back(name, d); open = false; }
    }
    return !open;
}

static void test_container() {
    Bytes out; pna_archive *a = nullptr;
    CHECK(pna_archive_new(vec_sink, &out, 0, &a) == PNA_OK);
    const Bytes p1 = text(100000, 1);
    CHECK(pna_archive_add_file(a, "dir/../a.txt", 0, (int64_t)p1.size(), p1.data(), p1.size(), 30000) == PNA_OK);
    CHECK(pna_archive_add_dir(a, "/some/dir/") == PNA_OK);
    const void *pcs[2] = {p1.data(), p1.data() + 5}; const size_t pl[2] = {5, 7};
    CHECK(pna_archive_add_solid(a, 0, pcs, pl, 2) == PNA_OK);
    CHECK(pna_archive_finalize(a) == PNA_OK);
    std::vector<Chunk> ch; CHECK(walk(out, ch));
    std::string kinds; for (auto &c : ch) { kinds += c.ty; kinds += ' '; }
    CHECK(kinds == "AHED FHED fSIZ FDAT FDAT FDAT FDAT FEND FHED FEND SHED SDAT SDAT SEND AEND ");
    CHECK(std::string((const char *)&out[ch[1].off + 6], ch[1].len - 6) == "a.txt");
    CHECK(std::string((const char *)&out[ch[8].off + 6], ch[8].len - 6) == "some/dir");
    // sanitize: the reference's vectors through the inner-record writer
    const char *vec[][2] = {{"/var/../tmp/./log", "tmp/log"}, {"test/../test.txt", "test.txt"}, {"../../..", ""}, {"a/b/./../a.txt", "a/a.txt"}, {"x\\y", "x\\y"}, {"", ""}};
    for (auto &v : vec) {
        uint8_t buf[256]; const size_t n = pna_archive_inner_entry_bytes(v[0], "d", 1, buf, sizeof buf);
        CHECK(n > 20 && std::string((const char *)buf + 14, (((size_t)buf[2] << 8) | buf[3]) - 6) == v[1]);
    }
    // seek_to_end / list_entries / split / join
    uint64_t at = 0; int nxt = 0;
    CHECK(pna_archive_seek_to_end(out.data(), out.size(), &at, &nxt) == PNA_OK && at == out.size() - 12 && nxt == 0);
    for (size_t cut = 1; cut <= 8; cut++) CHECK(pna_archive_seek_to_end(out.data(), out.size() - cut, &at, &nxt) == PNA_E_INVAL);
    std::vector<std::string> names;
    auto lcb = [](void *u, size_t, const char *nm, size_t nl, int, uint64_t, uint64_t) -> int { ((std::vector<std::string> *)u)->emplace_back(nm, nl); return 0; };
    CHECK(pna_archive_list_entries(out.data(), out.size(), lcb, &names) == PNA_OK && names.size() == 3 && names[0] == "a.txt" && names[2].empty());
    std::vector<Bytes> parts;
    auto psink = [](void *u, uint32_t idx, const void *b, size_t n) -> int { auto *v = (std::vector<Bytes> *)u; if (v->size() <= idx) v->resize(idx + 1); (*v)[idx].insert((*v)[idx].end(), (const uint8_t *)b, (const uint8_t *)b + n); return 0; };
    uint32_t np = 0;
    CHECK(pna_split_archive(out.data(), out.size(), 20000, psink, &parts, &np) == PNA_OK && np == parts.size() && np >= 5);
    std::vector<const void *> pp; std::vector<size_t> pn; for (auto &p : parts) { pp.push_back(p.data()); pn.push_back(p.size()); CHECK(p.size() <= 20000); }
    Bytes joined; CHECK(pna_join_parts(pp.data(), pn.data(), pp.size(), vec_sink, &joined) == PNA_OK);
    std::vector<Chunk> c2; CHECK(walk(joined, c2));
    Bytes d1, d2; for (auto &c : ch) if (!strcmp(c.ty, "FDAT")) d1.insert(d1.end(), out.begin() + c.off, out.begin() + c.off + c.len);
    for (auto &c : c2) if (!strcmp(c.ty, "FDAT")) d2.insert(d2.end(), joined.begin() + c.off, joined.begin() + c.off + c.len);
    CHECK(d1 == d2 && d1 == p1);
}
static void test_kdf() {
    uint8_t key[32], key2[32]; char phsf[128];
    CHECK(pna_kdf_pbkdf2_sha256("password", 8, "saltsaltsaltsalt", 16, 100, key, 32, phsf, sizeof phsf) == PNA_OK && !strncmp(phsf, "$pbkdf2-sha256$i=100,l=32$", 26));
    CHECK(pna_kdf_argon2(2, "password", 8, "saltsaltsaltsalt", 16, 1, 64, 2, key2, 32) == PNA_OK);
    CHECK(memcmp(key, key2, 32) != 0);
}
static void test_batch(pna_gpu_ctx *c) {
    const size_t lens[] = {0, 1, 7, 4096, 100000, 131072, 131073, (1u << 20), (1u << 20) + 5, 2500000};
    const size_t n = sizeof lens / sizeof lens[0];
    std::vector<Bytes> in(n), out(n); std::vector<const void *> src(n); std::vector<void *> dst(n); std::vector<size_t> sl(n), cap(n), dl(n);
    for (size_t i = 0; i < n; i++) { in[i] = text(lens[i], (uint32_t)i); out[i].resize(pna_gpu_bound(PNA_ALGO_ZSTD, lens[i])); src[i] = in[i].data(); dst[i] = out[i].data(); sl[i] = lens[i]; cap[i] = out[i].size(); }
    CHECK(pna_gpu_compress_batch(c, PNA_ALGO_ZSTD, 3, n, src.data(), sl.data(), dst.data(), cap.data(), dl.data()) == PNA_OK);
    for (size_t i = 0; i < n; i++) { Bytes d; CHECK(unraw(out[i].data(), dl[i], d) && d == in[i]); }
    cap[3] = 10;
    CHECK(pna_gpu_compress_batch(c, PNA_ALGO_ZSTD, 3, n, src.data(), sl.data(), dst.data(), cap.data(), dl.data()) == PNA_E_DSTSIZE);
    CHECK(pna_gpu_compress_batch(c, 4, 3, n, src.data(), sl.data(), dst.data(), cap.data(), dl.data()) == PNA_E_UNSUPPORTED);
}
static void test_streams(pna_gpu_ctx *c) {
    const unsigned T = 24, per = 12;
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            for (unsigned k = 0; k < per; k++) {
                const Bytes in = text((size_t)((t * 7919u + k * 104729u) % 700000u), t * 100 + k);
                Bytes out; pna_gpu_stream *s = nullptr;
                if (pna_gpu_stream_new(c, PNA_ALGO_ZSTD, 3, vec_sink, &out, &s) != PNA_OK) { bad++; continue; }
                size_t pos = 0; int rc = PNA_OK;
                while (pos < in.size() && rc == PNA_OK) { const size_t w = std::min<size_t>(in.size() - pos, 1 + (pos * 31 + k) % 90000); rc = pna_gpu_stream_write(s, in.data() + pos, w); pos += w; }
                if (rc != PNA_OK) { pna_gpu_stream_abort(s); bad++; continue; }
                if (k % 5 == 4) { pna_gpu_stream_abort(s); continue; }                // a builder dropped without build()
                if (pna_gpu_stream_finish(s) != PNA_OK) { bad++; continue; }
                Bytes d; if (!unraw(out.data(), out.size(), d) || d != in) bad++;
            }
        });
    for (auto &x : th) x.join();
    CHECK(bad.load() == 0);
    uint64_t b = 0, e = 0, mx = 0;
    CHECK(pna_gpu_stream_stats(c, &b, &e, &mx) == PNA_OK && e > 0 && b <= e && mx >= 1);
}
static void test_pipeline_and_append(pna_gpu_ctx *c) {
    setenv("PNA_SUB_MIB", "16", 1);                                      // several sub-batches: the stager thread and both slots are used
    const size_t n = 150;
    std::vector<Bytes> in(n); std::vector<std::string> nm(n); std::vector<const char *> names(n); std::vector<const void *> src(n); std::vector<size_t> sl(n);
    for (size_t i = 0; i < n; i++) { in[i] = text(i % 9 == 0 ? 0 : 200000 + 1777 * i, (uint32_t)i + 1000); nm[i] = "p/" + std::to_string(i) + ".txt"; names[i] = nm[i].c_str(); src[i] = in[i].data(); sl[i] = in[i].size(); }
    Bytes whole;
    CHECK(pna_gpu_create_archive_host(c, PNA_ALGO_ZSTD, 3, n, names.data(), src.data(), sl.data(), vec_sink, &whole) == PNA_OK);
    std::vector<std::pair<std::string, Bytes>> ents;
    CHECK(read_back(whole, ents) && ents.size() == n);
    for (size_t i = 0; i < n && i < ents.size(); i++) CHECK(ents[i].first == nm[i] && ents[i].second == in[i]);
    // append: the first 100 entries, then the other 50 behind them == all at once.  (Byte equality needs the block size to be the same in both runs: the library's
    // latency mode picks it by the size of the batch an entry happens to travel in, so the mode is switched off for this comparison.)
    CHECK(pna_gpu_set_option(c, "latency_max_mib", 0) == PNA_OK);
    whole.clear();
    CHECK(pna_gpu_create_archive_host(c, PNA_ALGO_ZSTD, 3, n, names.data(), src.data(), sl.data(), vec_sink, &whole) == PNA_OK);
    Bytes base, tail; uint64_t at = 0;
    CHECK(pna_gpu_create_archive_host(c, PNA_ALGO_ZSTD, 3, 100, names.data(), src.data(), sl.data(), vec_sink, &base) == PNA_OK);
    CHECK(pna_gpu_append_archive_host(c, PNA_ALGO_ZSTD, 3, base.data(), base.size(), n - 100, names.data() + 100, src.data() + 100, sl.data() + 100, &at, vec_sink, &tail) == PNA_OK);
    Bytes got(base.begin(), base.begin() + at); got.insert(got.end(), tail.begin(), tail.end());
    CHECK(got == whole);
    // streaming entry: FHED, meta, FDAT per burst (<= max chunk), FEND, no fSIZ
    Bytes rec; pna_gpu_entry_writer *w = nullptr;
    const uint8_t meta_body[1] = {1};
    Bytes meta; { const uint8_t h[8] = {0, 0, 0, 1, 'f', 'L', 'T', 'P'}; meta.insert(meta.end(), h, h + 8); meta.push_back(1); const uint32_t crc = pna_crc32(pna_crc32(0, "fLTP", 4), meta_body, 1); const uint8_t t[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc}; meta.insert(meta.end(), t, t + 4); }
    CHECK(pna_gpu_stream_entry_begin(c, PNA_ALGO_ZSTD, 3, "./s/../streamed.txt", meta.data(), meta.size(), 5000, vec_sink, &rec, &w) == PNA_OK);
    CHECK(pna_gpu_stream_entry_write(w, in[1].data(), in[1].size()) == PNA_OK && pna_gpu_stream_entry_finish(w) == PNA_OK);
    Bytes arc; { pna_archive *a = nullptr; CHECK(pna_archive_new(vec_sink, &arc, 0, &a) == PNA_OK); arc.insert(arc.end(), rec.begin(), rec.end()); CHECK(pna_archive_finalize(a) == PNA_OK); }
    std::vector<Chunk> ch; CHECK(walk(arc, ch));
    bool has_fsiz = false; size_t nfdat = 0; for (auto &x : ch) { if (!strcmp(x.ty, "fSIZ")) has_fsiz = true; if (!strcmp(x.ty, "FDAT")) { nfdat++; CHECK(x.len <= 5000); } }
    CHECK(!has_fsiz && nfdat >= in[1].size() / 5000 && !strcmp(ch[1].ty, "FHED") && !strcmp(ch[2].ty, "fLTP"));
    ents.clear(); CHECK(read_back(arc, ents) && ents.size() == 1 && ents[0].first == "streamed.txt" && ents[0].second == in[1]);
    CHECK(pna_gpu_stream_entry_begin(c, PNA_ALGO_ZSTD, 3, "x", "garbage", 7, 0, vec_sink, &rec, &w) == PNA_E_INVAL);
}

int main() {
    pna_gpu_ctx *c = nullptr;
    CHECK(pna_gpu_init(&c, 0, PNA_F_DEFAULT) == PNA_OK);
    if (!c) return 2;
    test_container();
    test_kdf();
    test_batch(c);
    test_streams(c);
    test_pipeline_and_append(c);
    pna_gpu_shutdown(c);
    if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
    printf("san_driver: all checks passed\n");
    return 0;
}
