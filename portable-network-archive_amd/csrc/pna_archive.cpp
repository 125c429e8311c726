// pna_archive.cpp -- host-side PNA container writer (C ABI in include/pna_archive.h).
// Byte-exact restatement of the reference's writer for the chunks the compression path produces; every function
// cites the reference code it mirrors.  No compression happens here.
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <sys/random.h>
#include "../../include/pna_archive.h"

namespace {

// ---- CRC-32 (IEEE 802.3, reflected 0xEDB88320), slice-by-8 -- chunk_crc, lib/src/format/chunk.rs:7-12
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};
const CrcTables &crc_tables() { static CrcTables T; return T; }

void put_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

} // namespace

extern "C" uint32_t pna_crc32(uint32_t crc, const void *buf, size_t len) {
    const CrcTables &T = crc_tables();
    const uint8_t *p = (const uint8_t *)buf;
    uint32_t c = ~crc;
    while (len && ((uintptr_t)p & 7)) { c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8); len--; }
    while (len >= 8) {
        uint64_t v; memcpy(&v, p, 8);
        uint32_t lo = (uint32_t)v ^ c, hi = (uint32_t)(v >> 32);
        c = T.t[7][lo & 0xFF] ^ T.t[6][(lo >> 8) & 0xFF] ^ T.t[5][(lo >> 16) & 0xFF] ^ T.t[4][lo >> 24] ^
            T.t[3][hi & 0xFF] ^ T.t[2][(hi >> 8) & 0xFF] ^ T.t[1][(hi >> 16) & 0xFF] ^ T.t[0][hi >> 24];
        p += 8; len -= 8;
    }
    while (len--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}

struct pna_archive {
    pna_sink_fn sink; void *user; int err;
    int emit(const void *p, size_t n) { if (err) return err; if (n && sink(user, p, n) != 0) err = PNA_E_SINK; return err; }
    // write_chunk: length BE | type | data | crc32(type||data) BE -- lib/src/io.rs:183-197
    int chunk(const char ty[4], const void *data, size_t len) {
        if (len > 0xFFFFFFFFull) return err = PNA_E_INVAL;
        uint8_t head[8]; put_be32(head, (uint32_t)len); memcpy(head + 4, ty, 4);
        uint32_t crc = pna_crc32(pna_crc32(0, ty, 4), data, len);
        uint8_t tail[4]; put_be32(tail, crc);
        emit(head, 8); emit(data, len); return emit(tail, 4);
    }
};

// EntryName::sanitize (lib/src/entry/name.rs:148-156): the path is NORMALISED first (normalize_utf8path, lib/src/util/utf8path.rs:6-33:
// "." dropped, ".." pops the preceding normal component, "//" collapsed), then only the normal components are kept and joined by '/'.
// A ".." that finds nothing to pop (or only other ".." / the root) never survives the filter, so a stack of normal components is enough.
// '/' is the only separator: the reference splits at '\\' on Windows only (camino follows the host's std::path).
static std::string sanitize_n(const char *name, size_t n) {
    std::vector<std::pair<size_t, size_t>> st;          // (start, length) of the normal components kept so far
    size_t i = 0;
    while (i <= n) {
        size_t j = i;
        while (j < n && name[j] != '/') j++;
        const size_t len = j - i;
        if (len == 0 || (len == 1 && name[i] == '.')) { /* empty, "." */ }
        else if (len == 2 && name[i] == '.' && name[i + 1] == '.') { if (!st.empty()) st.pop_back(); }
        else st.emplace_back(i, len);
        i = j + 1;
    }
    std::string out;
    for (auto &c : st) { if (!out.empty()) out += '/'; out.append(name + c.first, c.second); }
    return out;
}
static std::string sanitize(const char *name) { return sanitize_n(name ? name : "", name ? strlen(name) : 0); }
namespace pna { std::string pna_sanitize_name(const char *name, size_t n) { return sanitize_n(name, n); } }      // the read side hands out EntryHeader::path(), the sanitised form

static std::vector<uint8_t> fhed(int kind, int compression, int encryption, int cipher_mode, const std::string &name) {
    // EntryHeader::to_bytes, lib/src/entry/header.rs:123-134
    std::vector<uint8_t> h = {0, 0, (uint8_t)kind, (uint8_t)compression, (uint8_t)encryption, (uint8_t)cipher_mode};
    h.insert(h.end(), name.begin(), name.end());
    return h;
}

static size_t fsiz(uint64_t raw, uint8_t out[16]) {        // u128 BE, leading zeros stripped -- lib/src/entry.rs:900-903
    uint8_t be[16] = {0};
    for (int i = 0; i < 8; i++) be[15 - i] = (uint8_t)(raw >> (8 * i));
    size_t skip = 0; while (skip < 16 && be[skip] == 0) skip++;
    memcpy(out, be + skip, 16 - skip);
    return 16 - skip;
}

// ---- pieces of the entry record used by the in-HBM framing path (pna_gpu_create_archive_device, pna_host.cpp)
namespace pna {
static void put_chunk(std::vector<uint8_t> &o, const char ty[4], const uint8_t *data, size_t len) {
    uint8_t head[8]; put_be32(head, (uint32_t)len); memcpy(head + 4, ty, 4);
    uint8_t tail[4]; put_be32(tail, pna_crc32(pna_crc32(0, ty, 4), data, len));
    o.insert(o.end(), head, head + 8); if (len) o.insert(o.end(), data, data + len); o.insert(o.end(), tail, tail + 4);
}
// signature + AHED -- lib/src/archive/write.rs (write_header), lib/src/archive/header.rs:27-39
void frame_archive_head(std::vector<uint8_t> &o, uint32_t archive_number) {
    static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x41, 0x0D, 0x0A, 0x1A, 0x0A};
    uint8_t ahed[8] = {0, 0, 0, 0, 0, 0, 0, 0}; put_be32(ahed + 4, archive_number);
    o.insert(o.end(), sig, sig + 8); put_chunk(o, "AHED", ahed, 8);
}
void frame_archive_tail(std::vector<uint8_t> &o) { put_chunk(o, "AEND", nullptr, 0); }
// FHED | fSIZ | FDAT length + type: everything of a file entry that precedes its payload (NormalEntry::write_chunks_to, lib/src/entry.rs:895-911)
// true when EntryName::sanitize would return the name unchanged: relative, '/'-separated, no empty / "." / ".." component
static bool name_is_clean(const char *name, size_t n) {
    if (n == 0 || name[0] == '/' || name[n - 1] == '/') return false;
    size_t comp = 0;
    for (size_t i = 0; i <= n; i++) {
        const char ch = i < n ? name[i] : '/';
        if (ch == '/') {
            if (comp == 0) return false;
            if (comp == 1 && name[i - 1] == '.') return false;
            if (comp == 2 && name[i - 1] == '.' && name[i - 2] == '.') return false;
            comp = 0;
        } else comp++;
    }
    return true;
}
// the same record pieces written straight into `out` (>= frame_entry_prefix_bound(name) bytes); returns the length.  The many-entry path
// (hundreds of thousands of 4 KiB files) builds one of these per entry on the host while the kernels run: no allocation, no string copy
// for names that are already in sanitised form.
size_t frame_entry_prefix_into(uint8_t *out, const char *name, int compression, uint64_t raw_size) {
    std::string tmp;
    const char *nm = name ? name : ""; size_t nl = strlen(nm);
    if (!name_is_clean(nm, nl)) { tmp = sanitize(nm); nm = tmp.c_str(); nl = tmp.size(); }
    uint8_t *p = out;
    put_be32(p, (uint32_t)(6 + nl)); memcpy(p + 4, "FHED", 4);
    p[8] = 0; p[9] = 0; p[10] = 0; p[11] = (uint8_t)compression; p[12] = 0; p[13] = 1;      // cipher_mode CTR (1) when unencrypted
    memcpy(p + 14, nm, nl);
    put_be32(p + 14 + nl, pna_crc32(0, p + 4, 4 + 6 + nl));
    p += 12 + 6 + nl;
    uint8_t b[16]; const size_t n = fsiz(raw_size, b);
    put_be32(p, (uint32_t)n); memcpy(p + 4, "fSIZ", 4); memcpy(p + 8, b, n);
    put_be32(p + 8 + n, pna_crc32(0, p + 4, 4 + n));
    p += 12 + n;
    put_be32(p, 0); memcpy(p + 4, "FDAT", 4);
    return (size_t)(p + 8 - out);
}
void frame_entry_prefix(std::vector<uint8_t> &o, const char *name, int compression, uint64_t raw_size, uint32_t payload_len) {
    std::vector<uint8_t> h = fhed(0, compression, 0, 1, sanitize(name));
    put_chunk(o, "FHED", h.data(), h.size());
    uint8_t b[16]; size_t n = fsiz(raw_size, b); put_chunk(o, "fSIZ", b, n);
    uint8_t head[8]; put_be32(head, payload_len); memcpy(head + 4, "FDAT", 4);
    o.insert(o.end(), head, head + 8);
}
// The same for an entry written with a cipher: FHED(encryption, cipher_mode) | fSIZ | PHSF | FDAT(iv) | FDAT length + type.  The IV is
// the data-stream prefix and becomes a data piece of its own (prepend_data_prefix, lib/src/entry/builder.rs:62-69,171-188); PHSF
// stands between the metadata and the data chunks (lib/src/entry.rs:905-910).
void frame_entry_prefix_enc(std::vector<uint8_t> &o, const char *name, int compression, uint64_t raw_size, int encryption, int cipher_mode,
                            const char *phsf, const uint8_t *prefix, size_t prefix_len) {
    std::vector<uint8_t> h = fhed(0, compression, encryption, cipher_mode, sanitize(name));
    put_chunk(o, "FHED", h.data(), h.size());
    uint8_t b[16]; size_t n = fsiz(raw_size, b); put_chunk(o, "fSIZ", b, n);
    put_chunk(o, "PHSF", (const uint8_t *)phsf, strlen(phsf));
    put_chunk(o, "FDAT", prefix, prefix_len);                  // block IV (CBC / CTR) or the 75-byte GCM stream header: prefix_bytes(), lib/src/entry/write.rs:46-51
    uint8_t head[8]; put_be32(head, 0); memcpy(head + 4, "FDAT", 4);
    o.insert(o.end(), head, head + 8);
}
size_t frame_entry_prefix_enc_bound(const char *name, const char *phsf) { return 12 + 6 + (name ? strlen(name) : 0) + 12 + 16 + 12 + strlen(phsf) + 12 + 75 + 8; }
// body of the FHED chunk of a file entry (what the GCM stream key is bound to: entry_context, lib/src/cipher/aead.rs:165-182)
std::vector<uint8_t> frame_fhed_bytes(const char *name, int compression, int encryption, int cipher_mode) { return fhed(0, compression, encryption, cipher_mode, sanitize(name)); }
// an inner entry of a solid archive without data: FHED | fSIZ | FEND, no FDAT (FlattenWriter ignores empty writes)
void frame_inner_entry_empty(std::vector<uint8_t> &o, const char *name) {
    std::vector<uint8_t> h = fhed(0, 0, 0, 1, sanitize(name));
    put_chunk(o, "FHED", h.data(), h.size());
    uint8_t b[16]; size_t n = fsiz(0, b); put_chunk(o, "fSIZ", b, n);
    put_chunk(o, "FEND", nullptr, 0);
}
// SHED chunk of an unencrypted solid entry -- lib/src/entry/header.rs:274-282; SEND -- lib/src/archive/write.rs:716-727
void frame_solid_head(std::vector<uint8_t> &o, int compression) { const uint8_t shed[5] = {0, 0, (uint8_t)compression, 0, 1}; put_chunk(o, "SHED", shed, 5); }
// SHED | PHSF | SDAT(iv) of a solid entry written with a cipher -- into_solid_archive, lib/src/archive/write.rs:443-470
void frame_solid_head_enc(std::vector<uint8_t> &o, int compression, int encryption, int cipher_mode, const char *phsf, const uint8_t *prefix, size_t prefix_len) {
    const uint8_t shed[5] = {0, 0, (uint8_t)compression, (uint8_t)encryption, (uint8_t)cipher_mode};
    put_chunk(o, "SHED", shed, 5);
    put_chunk(o, "PHSF", (const uint8_t *)phsf, strlen(phsf));
    put_chunk(o, "SDAT", prefix, prefix_len);                       // CTR: the IV; GCM STREAM: the stream header
}
void frame_solid_tail(std::vector<uint8_t> &o) { put_chunk(o, "SEND", nullptr, 0); }
size_t frame_entry_prefix_bound(const char *name) { return 12 + 6 + (name ? strlen(name) : 0) + 12 + 16 + 8; }
uint32_t frame_fend_crc() { return pna_crc32(0, "FEND", 4); }
} // namespace pna

extern "C" int pna_archive_new(pna_sink_fn sink, void *user, uint32_t archive_number, pna_archive **out) {
    if (!sink || !out) return PNA_E_INVAL;
    pna_archive *a = new pna_archive{sink, user, 0};
    static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x41, 0x0D, 0x0A, 0x1A, 0x0A};     // lib/src/format/signature.rs:6
    uint8_t ahed[8] = {0, 0, 0, 0, 0, 0, 0, 0}; put_be32(ahed + 4, archive_number);     // lib/src/archive/header.rs:27-39
    a->emit(sig, 8); a->chunk("AHED", ahed, 8);
    if (a->err) { int e = a->err; delete a; return e; }
    *out = a; return PNA_OK;
}

extern "C" int pna_archive_add_file(pna_archive *a, const char *name, int compression, int64_t raw_size,
                                    const void *payload, size_t payload_len, uint32_t max_chunk_size) {
    if (!a || (!payload && payload_len)) return PNA_E_INVAL;
    // cipher_mode is CTR (1) when unencrypted: lib/src/entry/options.rs:156-159
    std::vector<uint8_t> h = fhed(0, compression, 0, 1, sanitize(name));
    a->chunk("FHED", h.data(), h.size());
    if (raw_size >= 0) { uint8_t b[16]; size_t n = fsiz((uint64_t)raw_size, b); a->chunk("fSIZ", b, n); }
    // FlattenWriter: one write of the whole stream -> pieces of at most max_chunk_size, lib/src/util/io.rs:60-77
    size_t mx = max_chunk_size ? max_chunk_size : 0xFFFFFFFFull;
    for (size_t p = 0; p < payload_len; p += mx) a->chunk("FDAT", (const uint8_t *)payload + p, std::min(mx, payload_len - p));
    return a->chunk("FEND", nullptr, 0);
}

extern "C" int pna_archive_add_dir(pna_archive *a, const char *name) {
    if (!a) return PNA_E_INVAL;
    std::vector<uint8_t> h = fhed(1, 0, 0, 0, sanitize(name));                            // lib/src/entry/header.rs:44-52,65-67
    a->chunk("FHED", h.data(), h.size());
    return a->chunk("FEND", nullptr, 0);
}

extern "C" int pna_archive_add_solid(pna_archive *a, int compression, const void *const *pieces, const size_t *piece_len, size_t n) {
    if (!a || (n && (!pieces || !piece_len))) return PNA_E_INVAL;
    uint8_t shed[5] = {0, 0, (uint8_t)compression, 0, 1};                                 // lib/src/entry/header.rs:274-282
    a->chunk("SHED", shed, 5);
    for (size_t i = 0; i < n; i++) if (piece_len[i]) a->chunk("SDAT", pieces[i], piece_len[i]);   // chunk/write.rs:32-47
    return a->chunk("SEND", nullptr, 0);
}

extern "C" size_t pna_archive_inner_entry_bytes(const char *name, const void *data, size_t len, void *dst, size_t cap) {
    std::vector<uint8_t> h = fhed(0, 0, 0, 1, sanitize(name));
    uint8_t fs[16]; size_t fn = fsiz(len, fs);
    size_t need = (12 + h.size()) + (12 + fn) + (len ? 12 + len : 0) + 12;
    if (!dst) return need;
    if (cap < need) return 0;
    struct Mem { uint8_t *p; size_t pos; } m{(uint8_t *)dst, 0};
    auto sink = [](void *u, const void *b, size_t n) -> int { Mem *mm = (Mem *)u; memcpy(mm->p + mm->pos, b, n); mm->pos += n; return 0; };
    pna_archive tmp{sink, &m, 0};
    tmp.chunk("FHED", h.data(), h.size());
    tmp.chunk("fSIZ", fs, fn);
    if (len) tmp.chunk("FDAT", data, len);                    // an empty payload produces no FDAT (FlattenWriter ignores empty writes)
    tmp.chunk("FEND", nullptr, 0);
    return m.pos;
}

extern "C" int pna_archive_finalize(pna_archive *a) {
    if (!a) return PNA_E_INVAL;
    int rc = a->chunk("AEND", nullptr, 0);                     // lib/src/io.rs:45-51
    delete a; return rc;
}
extern "C" void pna_archive_abort(pna_archive *a) { delete a; }

// create_archive_file -- cli/src/command/create.rs:575-635
extern "C" int pna_create_archive(pna_gpu_ctx *ctx, int algo, int level, int solid, size_t n, const char *const *names,
                                  const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user) {
    if (!sink || (n && (!names || !src || !src_len))) return PNA_E_INVAL;
    if (algo != PNA_ALGO_STORE && !ctx) return PNA_E_NODEVICE;
    // non-solid compressed archives: pipelined device path, archive bytes (framing + CRC included) come back from the GPU
    if (!solid && algo != PNA_ALGO_STORE) return pna_gpu_create_archive_host(ctx, algo, level, n, names, src, src_len, sink, user);
    if (solid && algo != PNA_ALGO_STORE) return pna_gpu_create_solid_archive_host(ctx, algo, level, n, names, src, src_len, sink, user);
    pna_archive *a = nullptr;
    int rc = pna_archive_new(sink, user, 0, &a);
    if (rc) return rc;
    if (!solid) {
        for (size_t i = 0; i < n && !rc; i++)                 // STORE; drain_entry_results: index order, core.rs:471-493
            rc = pna_archive_add_file(a, names[i], algo, (int64_t)src_len[i], src[i], src_len[i], 0);
    } else {
        // solid: inner entries are STORE (create.rs:594-598), serialised as chunk bytes, then ONE compressed stream
        std::vector<uint8_t> plain;
        for (size_t i = 0; i < n; i++) {
            size_t need = pna_archive_inner_entry_bytes(names[i], src[i], src_len[i], nullptr, 0);
            size_t at = plain.size(); plain.resize(at + need);
            pna_archive_inner_entry_bytes(names[i], src[i], src_len[i], plain.data() + at, need);
        }
        struct Pieces { std::vector<std::vector<uint8_t>> v; } pcs;
        if (algo == PNA_ALGO_STORE) { if (!plain.empty()) pcs.v.push_back(plain); }
        else {
            auto psink = [](void *u, const void *b, size_t l) -> int { ((Pieces *)u)->v.emplace_back((const uint8_t *)b, (const uint8_t *)b + l); return 0; };
            rc = pna_gpu_compress_solid(ctx, algo, level, plain.data(), plain.size(), psink, &pcs);
            if (rc) { pna_archive_abort(a); return rc; }
        }
        std::vector<const void *> pp; std::vector<size_t> pl;
        for (auto &v : pcs.v) { pp.push_back(v.data()); pl.push_back(v.size()); }
        rc = pna_archive_add_solid(a, algo, pp.data(), pl.data(), pp.size());
    }
    if (rc) { pna_archive_abort(a); return rc; }
    return pna_archive_finalize(a);
}

// ---------------------------------------------------------------------------------------------------------
// Password hashing for `pna create --aes --pbkdf2` on the C++ host: PBKDF2-HMAC-SHA-256 (FIPS 180-4, RFC 2104, RFC 8018), the
// reference's hash::pbkdf2_with_salt (lib/src/hash.rs:35-45; pbkdf2 0.12 defaults: 600 000 rounds, 32-byte output) with a
// password-hash SaltString (16 random bytes, B64 without padding; the hash function is fed the decoded bytes).
namespace {
struct Sha256 {
    uint32_t h[8]; uint64_t len; uint8_t buf[64]; size_t fill;
    static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void init() {
        static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        memcpy(h, iv, sizeof iv); len = 0; fill = 0;
    }
    void block(const uint8_t *p) {
        struct KTab { uint32_t v[64]; };
        // K[i] = frac(cbrt(prime_i)) * 2^32, by integer cube root of prime << 96; built once (C++11 static: safe when several host
        // threads hash at the same time, as the per-entry GCM key derivation does)
        static const KTab KT = [] {
            KTab t; int cnt = 0;
            for (uint32_t c = 2; cnt < 64; c++) {
                bool pr = true; for (uint32_t d = 2; d * d <= c; d++) if (c % d == 0) { pr = false; break; }
                if (!pr) continue;
                unsigned __int128 target = (unsigned __int128)c << 96, lo = 0, hi = (unsigned __int128)1 << 36;
                while (hi - lo > 1) { unsigned __int128 mid = (lo + hi) / 2; if (mid * mid * mid <= target) lo = mid; else hi = mid; }
                t.v[cnt++] = (uint32_t)lo;
            }
            return t;
        }();
        const uint32_t *K = KT.v;
        uint32_t w[64], a[8];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
        memcpy(a, h, sizeof a);
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = a[7] + (ror(a[4], 6) ^ ror(a[4], 11) ^ ror(a[4], 25)) + ((a[4] & a[5]) ^ (~a[4] & a[6])) + K[i] + w[i];
            const uint32_t t2 = (ror(a[0], 2) ^ ror(a[0], 13) ^ ror(a[0], 22)) + ((a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]));
            a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1; a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
        }
        for (int i = 0; i < 8; i++) h[i] += a[i];
    }
    void update(const void *d, size_t n) {
        const uint8_t *p = (const uint8_t *)d; len += n;
        while (n) { size_t k = std::min(n, 64 - fill); memcpy(buf + fill, p, k); fill += k; p += k; n -= k; if (fill == 64) { block(buf); fill = 0; } }
    }
    void final(uint8_t out[32]) {
        const uint64_t bits = len * 8; uint8_t pad[72] = {0x80}; const size_t padn = (fill < 56 ? 56 : 120) - fill;
        update(pad, padn);
        uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(lb, 8);
        for (int i = 0; i < 8; i++) put_be32(out + 4 * i, h[i]);
    }
};
struct Hmac {                                                  // inner / outer states after the key pads: two compressions per PBKDF2 round
    Sha256 in0, out0;
    void key(const uint8_t *k, size_t n) {
        uint8_t kb[64] = {0}, pad[64];
        if (n > 64) { Sha256 t; t.init(); t.update(k, n); t.final(kb); } else memcpy(kb, k, n);
        for (int i = 0; i < 64; i++) pad[i] = kb[i] ^ 0x36;
        in0.init(); in0.update(pad, 64);
        for (int i = 0; i < 64; i++) pad[i] = kb[i] ^ 0x5c;
        out0.init(); out0.update(pad, 64);
    }
    void mac(const void *a, size_t an, const void *b, size_t bn, uint8_t out[32]) const {
        Sha256 s = in0; s.update(a, an); if (bn) s.update(b, bn);
        uint8_t inner[32]; s.final(inner);
        Sha256 o = out0; o.update(inner, 32); o.final(out);
    }
};
const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
std::string b64_nopad(const uint8_t *p, size_t n) {
    std::string o;
    for (size_t i = 0; i < n; i += 3) {
        const uint32_t v = ((uint32_t)p[i] << 16) | ((i + 1 < n ? (uint32_t)p[i + 1] : 0u) << 8) | (i + 2 < n ? (uint32_t)p[i + 2] : 0u);
        o += B64[(v >> 18) & 63]; o += B64[(v >> 12) & 63];
        if (i + 1 < n) o += B64[(v >> 6) & 63];
        if (i + 2 < n) o += B64[v & 63];
    }
    return o;
}
} // namespace

namespace pna {
void sha256_bytes(const void *a, size_t an, const void *b, size_t bn, uint8_t out[32]) { Sha256 s; s.init(); s.update(a, an); if (bn) s.update(b, bn); s.final(out); }
// HKDF-SHA-256 with one output block (RFC 5869): hkdf_sha256, lib/src/cipher/aead.rs:151-157
void hkdf_sha256_32(const void *ikm, size_t ikm_len, const void *salt, size_t salt_len, const void *info, size_t info_len, uint8_t okm[32]) {
    static const uint8_t zero[32] = {0};
    Hmac ex; if (salt_len) ex.key((const uint8_t *)salt, salt_len); else ex.key(zero, 32);
    uint8_t prk[32]; ex.mac(ikm, ikm_len, nullptr, 0, prk);
    Hmac xp; xp.key(prk, 32);
    const uint8_t one = 1;
    std::vector<uint8_t> m((const uint8_t *)info, (const uint8_t *)info + info_len); m.push_back(one);
    xp.mac(m.data(), m.size(), nullptr, 0, okm);
}
}

extern "C" int pna_kdf_pbkdf2_sha256(const void *password, size_t password_len, const void *salt, size_t salt_len, uint32_t rounds,
                                     uint8_t *key, size_t key_len, char *phsf, size_t phsf_cap) {
    if ((!password && password_len) || (!salt && salt_len) || !key || rounds == 0) return PNA_E_INVAL;
    Hmac hm; hm.key((const uint8_t *)password, password_len);
    for (uint32_t blk = 1; key_len; blk++) {
        uint8_t be[4]; put_be32(be, blk);
        uint8_t u[32], t[32];
        hm.mac(salt, salt_len, be, 4, u); memcpy(t, u, 32);
        for (uint32_t r = 1; r < rounds; r++) { hm.mac(u, 32, nullptr, 0, u); for (int i = 0; i < 32; i++) t[i] ^= u[i]; }
        const size_t k = std::min<size_t>(key_len, 32);
        memcpy(key, t, k); key += k; key_len -= k;
    }
    if (phsf) {                                                // PasswordHash::to_string() after hash.take(): "$pbkdf2-sha256$i=<rounds>,l=<len>$<salt>"
        const std::string s = "$pbkdf2-sha256$i=" + std::to_string(rounds) + ",l=32$" + b64_nopad((const uint8_t *)salt, salt_len);
        if (s.size() + 1 > phsf_cap) return PNA_E_DSTSIZE;
        memcpy(phsf, s.c_str(), s.size() + 1);
    }
    return PNA_OK;
}


// ---------------------------------------------------------------------------------------------------------
// Argon2 (RFC 9106, version 0x13, no secret / associated data) on the C++ host: the reference's default password hash
// (hash::argon2_with_salt / derive_password_hash, lib/src/hash.rs:6-33,47-70; argon2 0.5.3).  The read side needs it to open archives
// the reference wrote; the key is the raw hash output of key_size() bytes (lib/src/entry/write.rs:146-151).
namespace {
struct Blake2b {                                               // RFC 7693, unkeyed
    uint64_t h[8], t = 0; uint8_t buf[128]; size_t fill = 0, outlen = 64;
    static uint64_t ror(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    static const uint64_t *iv() {
        static const uint64_t IV[8] = {0x6A09E667F3BCC908ull, 0xBB67AE8584CAA73Bull, 0x3C6EF372FE94F82Bull, 0xA54FF53A5F1D36F1ull,
                                       0x510E527FADE682D1ull, 0x9B05688C2B3E6C1Full, 0x1F83D9ABFB41BD6Bull, 0x5BE0CD19137E2179ull};
        return IV;
    }
    explicit Blake2b(size_t out) : outlen(out) { memcpy(h, iv(), 64); h[0] ^= 0x01010000ull ^ (uint64_t)out; }
    void compress(const uint8_t *blk, bool last) {
        static const uint8_t SG[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; i++) { m[i] = 0; for (int b = 0; b < 8; b++) m[i] |= (uint64_t)blk[8 * i + b] << (8 * b); }
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = iv()[i]; }
        v[12] ^= t; if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] += v[b] + x; v[d] = ror(v[d] ^ v[a], 32); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 24);
            v[a] += v[b] + y; v[d] = ror(v[d] ^ v[a], 16); v[c] += v[d]; v[b] = ror(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; r++) {
            const uint8_t *s = SG[r % 10];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]); G(1, 5, 9, 13, m[s[2]], m[s[3]]); G(2, 6, 10, 14, m[s[4]], m[s[5]]); G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]); G(1, 6, 11, 12, m[s[10]], m[s[11]]); G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const void *d, size_t n) {
        const uint8_t *p = (const uint8_t *)d;
        while (n) {
            if (fill == 128) { t += 128; compress(buf, false); fill = 0; }
            const size_t k = std::min(n, 128 - fill);
            memcpy(buf + fill, p, k); fill += k; p += k; n -= k;
        }
    }
    void u32(uint32_t v) { uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; update(b, 4); }
    void final(uint8_t *out) {
        t += fill; memset(buf + fill, 0, 128 - fill); compress(buf, true);
        uint8_t full[64]; for (int i = 0; i < 8; i++) for (int b = 0; b < 8; b++) full[8 * i + b] = (uint8_t)(h[i] >> (8 * b));
        memcpy(out, full, outlen);
    }
};
// H' of RFC 9106 3.3: output of any length from 64-byte BLAKE2b digests, 32 bytes kept of each but the last
void argon_hprime(const uint8_t *in, size_t n, uint8_t *out, uint32_t T) {
    if (T <= 64) { Blake2b b(T); b.u32(T); b.update(in, n); b.final(out); return; }
    uint8_t v[64];
    { Blake2b b(64); b.u32(T); b.update(in, n); b.final(v); }
    size_t done = 0;
    for (;;) {
        memcpy(out + done, v, 32); done += 32;
        if (T - done <= 64) break;
        uint8_t nx[64]; { Blake2b b(64); b.update(v, 64); b.final(nx); } memcpy(v, nx, 64);
    }
    { Blake2b b(T - done); b.update(v, 64); b.final(out + done); }
}
typedef uint64_t ABlock[128];
inline uint64_t a_mix(uint64_t x, uint64_t y) { return x + y + 2 * (x & 0xFFFFFFFFull) * (y & 0xFFFFFFFFull); }
inline void a_quarter(uint64_t &a, uint64_t &b, uint64_t &c, uint64_t &d) {
    a = a_mix(a, b); d = Blake2b::ror(d ^ a, 32); c = a_mix(c, d); b = Blake2b::ror(b ^ c, 24);
    a = a_mix(a, b); d = Blake2b::ror(d ^ a, 16); c = a_mix(c, d); b = Blake2b::ror(b ^ c, 63);
}
inline void a_perm(uint64_t *v, const int idx[16]) {
    auto q = [&](int a, int b, int c, int d) { a_quarter(v[idx[a]], v[idx[b]], v[idx[c]], v[idx[d]]); };
    q(0, 4, 8, 12); q(1, 5, 9, 13); q(2, 6, 10, 14); q(3, 7, 11, 15); q(0, 5, 10, 15); q(1, 6, 11, 12); q(2, 7, 8, 13); q(3, 4, 9, 14);
}
// next = P(prev ^ ref) ^ prev ^ ref (^ old next from the second pass on), P = the BLAKE2b round on rows then columns of 8 x 8 registers
void argon_g(const uint64_t *prev, const uint64_t *ref, uint64_t *next, bool xor_old) {
    uint64_t r[128], keep[128];
    for (int i = 0; i < 128; i++) { r[i] = prev[i] ^ ref[i]; keep[i] = xor_old ? r[i] ^ next[i] : r[i]; }
    int idx[16];
    for (int row = 0; row < 8; row++) { for (int k = 0; k < 16; k++) idx[k] = 16 * row + k; a_perm(r, idx); }
    for (int col = 0; col < 8; col++) { for (int k = 0; k < 8; k++) { idx[2 * k] = 16 * k + 2 * col; idx[2 * k + 1] = 16 * k + 2 * col + 1; } a_perm(r, idx); }
    for (int i = 0; i < 128; i++) next[i] = r[i] ^ keep[i];
}
} // namespace

// kind: 0 Argon2d, 1 Argon2i, 2 Argon2id
extern "C" int pna_kdf_argon2(int kind, const void *password, size_t password_len, const void *salt, size_t salt_len,
                              uint32_t t_cost, uint32_t m_cost_kib, uint32_t lanes, uint8_t *key, size_t key_len) {
    if ((!password && password_len) || (!salt && salt_len) || !key || kind < 0 || kind > 2) return PNA_E_INVAL;
    if (lanes < 1 || lanes > 0xFFFFFF || t_cost < 1 || m_cost_kib < 8 * lanes || key_len < 4 || key_len > 1024 || m_cost_kib > (1u << 24)) return PNA_E_INVAL;
    const uint32_t blocks = 4 * lanes * (m_cost_kib / (4 * lanes)), lane_len = blocks / lanes, seg = lane_len / 4;
    std::vector<uint64_t> mem;
    try { mem.resize((size_t)blocks * 128); } catch (...) { return PNA_E_NOMEM; }
    auto B = [&](uint32_t lane, uint32_t col) -> uint64_t * { return mem.data() + ((size_t)lane * lane_len + col) * 128; };
    uint8_t h0[72];
    {
        Blake2b b(64);
        b.u32(lanes); b.u32((uint32_t)key_len); b.u32(m_cost_kib); b.u32(t_cost); b.u32(0x13); b.u32((uint32_t)kind);
        b.u32((uint32_t)password_len); b.update(password, password_len);
        b.u32((uint32_t)salt_len); b.update(salt, salt_len);
        b.u32(0); b.u32(0);
        b.final(h0);
    }
    for (uint32_t l = 0; l < lanes; l++)
        for (uint32_t j = 0; j < 2; j++) {
            uint8_t raw[1024];
            for (int k = 0; k < 4; k++) { h0[64 + k] = (uint8_t)(j >> (8 * k)); h0[68 + k] = (uint8_t)(l >> (8 * k)); }
            argon_hprime(h0, 72, raw, 1024);
            uint64_t *blk = B(l, j);
            for (int i = 0; i < 128; i++) { blk[i] = 0; for (int k = 0; k < 8; k++) blk[i] |= (uint64_t)raw[8 * i + k] << (8 * k); }
        }
    const ABlock zero = {0};
    for (uint32_t pass = 0; pass < t_cost; pass++)
        for (uint32_t slice = 0; slice < 4; slice++)
            for (uint32_t lane = 0; lane < lanes; lane++) {
                const bool indep = kind == 1 || (kind == 2 && pass == 0 && slice < 2);     // data-independent addressing
                ABlock in = {0}, addr = {0};
                if (indep) { in[0] = pass; in[1] = lane; in[2] = slice; in[3] = blocks; in[4] = t_cost; in[5] = (uint64_t)kind; }
                auto next_addr = [&]() { in[6]++; ABlock t = {0}; argon_g(zero, in, t, false); argon_g(zero, t, addr, false); };
                uint32_t first = 0;
                if (pass == 0 && slice == 0) { first = 2; if (indep) next_addr(); }
                for (uint32_t i = first; i < seg; i++) {
                    const uint32_t col = slice * seg + i, pcol = col ? col - 1 : lane_len - 1;
                    uint64_t rnd;
                    if (indep) { if (i % 128 == 0) next_addr(); rnd = addr[i % 128]; } else rnd = B(lane, pcol)[0];
                    uint32_t rl = (uint32_t)((rnd >> 32) % lanes);
                    if (pass == 0 && slice == 0) rl = lane;
                    const bool same = rl == lane;
                    // number of blocks that may be referenced (RFC 9106 3.4.1.1 / 3.4.2)
                    uint64_t area;
                    if (pass == 0) area = slice == 0 ? i - 1 : (same ? (uint64_t)slice * seg + i - 1 : (uint64_t)slice * seg - (i == 0 ? 1 : 0));
                    else area = same ? (uint64_t)lane_len - seg + i - 1 : (uint64_t)lane_len - seg - (i == 0 ? 1 : 0);
                    uint64_t x = rnd & 0xFFFFFFFFull; x = (x * x) >> 32;
                    const uint64_t rel = area - 1 - ((area * x) >> 32);
                    const uint64_t start = (pass == 0 || slice == 3) ? 0 : (uint64_t)(slice + 1) * seg;
                    const uint32_t rcol = (uint32_t)((start + rel) % lane_len);
                    argon_g(B(lane, pcol), B(rl, rcol), B(lane, col), pass != 0);
                }
            }
    uint64_t fin[128];
    memcpy(fin, B(0, lane_len - 1), sizeof fin);
    for (uint32_t l = 1; l < lanes; l++) for (int i = 0; i < 128; i++) fin[i] ^= B(l, lane_len - 1)[i];
    uint8_t raw[1024];
    for (int i = 0; i < 128; i++) for (int k = 0; k < 8; k++) raw[8 * i + k] = (uint8_t)(fin[i] >> (8 * k));
    argon_hprime(raw, 1024, key, (uint32_t)key_len);
    return PNA_OK;
}

// `pna create --aes [ctr|cbc] --password ... --pbkdf2`: one key derivation per archive (WriteOptions caches it, lib/src/entry/options.rs),
// a fresh IV per entry; non-solid zstd / deflate archives on the device path.
extern "C" int pna_create_archive_encrypted(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                            const void *const *src, const size_t *src_len, const void *password, size_t password_len,
                                            int cipher_mode, uint32_t rounds, pna_sink_fn sink, void *user) {
    if (!ctx) return PNA_E_NODEVICE;
    if (!sink || (!password && password_len)) return PNA_E_INVAL;
    uint8_t salt[16];
    if (getrandom(salt, sizeof salt, 0) != (ssize_t)sizeof salt) return PNA_E_INVAL;
    pna_gpu_cipher ci{};
    ci.encryption = PNA_ENC_AES; ci.cipher_mode = cipher_mode; ci.ivs = nullptr;
    char phsf[128];
    int rc = pna_kdf_pbkdf2_sha256(password, password_len, salt, sizeof salt, rounds ? rounds : 600000u, ci.key, 32, phsf, sizeof phsf);
    if (rc) return rc;
    ci.phsf = phsf;
    return pna_gpu_create_archive_enc_host(ctx, algo, level, n, names, src, src_len, &ci, sink, user);
}

// ---------------------------------------------------------------------------------------------------------
// Multipart archives (`pna create --split`): SplitParts, lib/src/archive/split_parts.rs.  The device paths produce ONE archive image;
// this re-frames its chunk stream into parts of at most max_part_bytes: every part opens with the signature + AHED(archive number),
// closes with [ANXT] AEND; a chunk that fits goes out untouched (its CRC is reused), a non-stream chunk that does not fit opens the next
// part, an FDAT / SDAT chunk is cut at the budget boundary and only its fragments get new CRCs (put_chunk / put_stream, :140-173).
namespace {
struct PartOut {
    pna_part_sink_fn sink; void *user; uint32_t part = 0; int err = 0;
    void emit(const void *p, size_t n) { if (!err && n && sink(user, part, p, n) != 0) err = PNA_E_SINK; }
    void header() {
        static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x41, 0x0D, 0x0A, 0x1A, 0x0A};
        uint8_t c[20]; put_be32(c, 8); memcpy(c + 4, "AHED", 4); memset(c + 8, 0, 4); put_be32(c + 12, part);
        put_be32(c + 16, pna_crc32(0, c + 4, 12));
        emit(sig, 8); emit(c, 20);
    }
    void empty_chunk(const char ty[4]) { uint8_t c[12]; put_be32(c, 0); memcpy(c + 4, ty, 4); put_be32(c + 8, pna_crc32(0, ty, 4)); emit(c, 12); }
    void fresh_chunk(const uint8_t *ty, const uint8_t *data, size_t n) {
        uint8_t h[8], t[4]; put_be32(h, (uint32_t)n); memcpy(h + 4, ty, 4);
        put_be32(t, pna_crc32(pna_crc32(0, ty, 4), data, n));
        emit(h, 8); emit(data, n); emit(t, 4);
    }
};
}

extern "C" int pna_split_archive(const void *archive, size_t len, size_t max_part_bytes, pna_part_sink_fn sink, void *user, uint32_t *n_parts) {
    if (!archive || !sink) return PNA_E_INVAL;
    const size_t MINC = 12, OVER = 8 + 20 + 2 * MINC;          // MIN_CHUNK_BYTES_SIZE, SPLIT_ARCHIVE_OVERHEAD_BYTES (split_parts.rs:14-23)
    if (max_part_bytes < OVER + MINC) return PNA_E_INVAL;      // MIN_SPLIT_PART_BYTES
    const uint8_t *a = (const uint8_t *)archive;
    static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x41, 0x0D, 0x0A, 0x1A, 0x0A};
    if (len < 8 + 20 + 12 || memcmp(a, sig, 8) != 0 || memcmp(a + 12, "AHED", 4) != 0) return PNA_E_INVAL;
    const size_t budget = max_part_bytes - OVER;
    size_t remaining = budget;
    PartOut o{sink, user};
    o.header();
    auto roll_over = [&]() -> int {
        if (o.part == 0xFFFFFFFFu) return PNA_E_INVAL;          // part_number_overflow_error
        o.empty_chunk("ANXT"); o.empty_chunk("AEND");
        o.part++; o.header(); remaining = budget;
        return o.err;
    };
    size_t pos = 8 + 20; bool ended = false;
    while (pos < len && !o.err) {
        if (len - pos < 12) return PNA_E_INVAL;
        const uint32_t dl = ((uint32_t)a[pos] << 24) | ((uint32_t)a[pos + 1] << 16) | ((uint32_t)a[pos + 2] << 8) | a[pos + 3];
        const uint8_t *ty = a + pos + 4, *data = a + pos + 8;
        if (len - pos - 12 < dl) return PNA_E_INVAL;
        const size_t clen = MINC + dl;
        if (memcmp(ty, "AEND", 4) == 0) { ended = true; break; }
        if (memcmp(ty, "ANXT", 4) == 0) return PNA_E_INVAL;     // already a part of a multipart archive
        const bool stream = memcmp(ty, "FDAT", 4) == 0 || memcmp(ty, "SDAT", 4) == 0;
        if (clen <= remaining) { o.emit(a + pos, clen); remaining -= clen; }
        else if (!stream) {
            if (clen > budget) return PNA_E_INVAL;              // chunk_does_not_fit_error
            int rc = roll_over(); if (rc) return rc;
            o.emit(a + pos, clen); remaining -= clen;
        } else if (clen <= budget && remaining <= MINC) {
            int rc = roll_over(); if (rc) return rc;
            o.emit(a + pos, clen); remaining -= clen;
        } else {
            const uint8_t *rest = data; size_t rl = dl; bool first = true;
            for (;;) {
                if (MINC + rl <= remaining) {
                    if (first) o.emit(a + pos, clen); else o.fresh_chunk(ty, rest, rl);   // (cannot be `first` here, kept for symmetry)
                    remaining -= MINC + rl; break;
                }
                if (remaining > MINC) {
                    const size_t take = remaining - MINC;
                    o.fresh_chunk(ty, rest, take); rest += take; rl -= take; remaining -= MINC + take; first = false;
                } else if (budget <= MINC) return PNA_E_INVAL;
                int rc = roll_over(); if (rc) return rc;
            }
        }
        pos += clen;
    }
    if (!ended) return o.err ? o.err : PNA_E_INVAL;
    o.empty_chunk("AEND");                                      // finalize_archive: the last part has no ANXT
    if (n_parts) *n_parts = o.part + 1;
    return o.err;
}

// The reading side (Archive::read_next_archive): parts in order -> ONE archive image (signature + AHED(0) + the concatenated chunk
// streams + AEND) that pna_gpu_extract_archive_host and every other reader take; FDAT / SDAT fragments stay separate chunks (their
// bodies are concatenated by the entry reader anyway).
extern "C" int pna_join_parts(const void *const *parts, const size_t *part_len, size_t n, pna_sink_fn sink, void *user) {
    if (!parts || !part_len || !sink || n == 0) return PNA_E_INVAL;
    static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x41, 0x0D, 0x0A, 0x1A, 0x0A};
    auto out = [&](const void *p, size_t k) { return k == 0 || sink(user, p, k) == 0; };
    for (size_t k = 0; k < n; k++) {
        const uint8_t *a = (const uint8_t *)parts[k]; const size_t len = part_len[k];
        if (!a || len < 8 + 20 + 12 || memcmp(a, sig, 8) != 0 || memcmp(a + 12, "AHED", 4) != 0) return PNA_E_INVAL;
        const uint32_t num = ((uint32_t)a[20] << 24) | ((uint32_t)a[21] << 16) | ((uint32_t)a[22] << 8) | a[23];      // AHED body: major, minor, 0, 0, archive number
        if (num != k || pna_crc32(0, a + 12, 12) != (((uint32_t)a[24] << 24) | ((uint32_t)a[25] << 16) | ((uint32_t)a[26] << 8) | a[27])) return PNA_E_INVAL;
        if (k == 0 && !out(a, 28)) return PNA_E_SINK;
        size_t pos = 28; bool has_next = false, ended = false;
        while (pos < len) {
            if (len - pos < 12) return PNA_E_INVAL;
            const uint32_t dl = ((uint32_t)a[pos] << 24) | ((uint32_t)a[pos + 1] << 16) | ((uint32_t)a[pos + 2] << 8) | a[pos + 3];
            if (len - pos - 12 < dl) return PNA_E_INVAL;
            const uint8_t *ty = a + pos + 4;
            if (memcmp(ty, "AEND", 4) == 0) { ended = true; break; }
            if (memcmp(ty, "ANXT", 4) == 0) has_next = true;
            else { if (has_next) return PNA_E_INVAL; if (!out(a + pos, 12 + (size_t)dl)) return PNA_E_SINK; }
            pos += 12 + (size_t)dl;
        }
        if (!ended || has_next != (k + 1 < n)) return PNA_E_INVAL;
    }
    uint8_t c[12]; put_be32(c, 0); memcpy(c + 4, "AEND", 4); put_be32(c + 8, pna_crc32(0, "AEND", 4));
    return out(c, 12) ? PNA_OK : PNA_E_SINK;
}


// ---- `pna append` / `pna update` plumbing on the host (include/pna_archive.h)
static bool pna_image_header_ok(const uint8_t *a, size_t len) {
    static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x41, 0x0D, 0x0A, 0x1A, 0x0A};
    // signature, then AHED as the first chunk (Archive::read_header, lib/src/archive/read.rs:26-44)
    return len >= 8 + 12 + 8 && memcmp(a, sig, 8) == 0 && memcmp(a + 12, "AHED", 4) == 0 && a[8] == 0 && a[9] == 0 && a[10] == 0 && a[11] == 8;
}
extern "C" int pna_archive_seek_to_end(const void *archive, size_t len, uint64_t *aend_off, int *has_next) {
    if (!archive || !aend_off) return PNA_E_INVAL;
    const uint8_t *a = (const uint8_t *)archive;
    if (!pna_image_header_ok(a, len)) return PNA_E_INVAL;
    if (has_next) *has_next = 0;
    size_t pos = 8;
    for (;;) {
        if (len - pos < 12) return PNA_E_INVAL;                                   // truncated: UnexpectedEof
        const uint64_t l = ((uint64_t)a[pos] << 24) | ((uint64_t)a[pos + 1] << 16) | ((uint64_t)a[pos + 2] << 8) | a[pos + 3];
        if (len - pos - 12 < l) return PNA_E_INVAL;
        if (memcmp(a + pos + 4, "AEND", 4) == 0) { *aend_off = pos; return PNA_OK; }
        if (memcmp(a + pos + 4, "ANXT", 4) == 0 && has_next) *has_next = 1;
        pos += 12 + (size_t)l;
    }
}
extern "C" int pna_archive_list_entries(const void *archive, size_t len, pna_raw_entry_fn cb, void *user) {
    if (!archive || !cb) return PNA_E_INVAL;
    const uint8_t *a = (const uint8_t *)archive;
    if (!pna_image_header_ok(a, len)) return PNA_E_INVAL;
    size_t pos = 8, idx = 0, start = 0, name_off = 0, name_len = 0;
    int open = 0, kind = 0;                                                       // 1: inside FHED..FEND, 2: inside SHED..SEND
    for (;;) {
        if (len - pos < 12) return PNA_E_INVAL;
        const uint64_t l = ((uint64_t)a[pos] << 24) | ((uint64_t)a[pos + 1] << 16) | ((uint64_t)a[pos + 2] << 8) | a[pos + 3];
        if (len - pos - 12 < l) return PNA_E_INVAL;
        const uint8_t *ty = a + pos + 4;
        if (!open) {
            if (memcmp(ty, "AEND", 4) == 0) return PNA_OK;
            if (memcmp(ty, "FHED", 4) == 0) { if (l < 6) return PNA_E_INVAL; open = 1; start = pos; kind = a[pos + 8 + 2]; name_off = pos + 8 + 6; name_len = (size_t)l - 6; }
            else if (memcmp(ty, "SHED", 4) == 0) { open = 2; start = pos; kind = -1; name_off = pos; name_len = 0; }
        } else if ((open == 1 && memcmp(ty, "FEND", 4) == 0) || (open == 2 && memcmp(ty, "SEND", 4) == 0)) {
            if (cb(user, idx++, (const char *)a + name_off, name_len, kind, start, pos + 12 + l - start) != 0) return PNA_E_SINK;
            open = 0;
        }
        pos += 12 + (size_t)l;
    }
}
