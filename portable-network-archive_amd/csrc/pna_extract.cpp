// pna_extract.cpp -- the read-side driver of libpna_gpu.so: `pna extract` / `pna verify` over archives in host memory.
#include "pna_ctx.h"
// ---------------------------------------------------------------------------------------------------------
// Read side driver: `pna extract` / `pna verify` for non-solid archives (cli/src/command/extract.rs:594-640, verify.rs:140-188;
// Archive::read_header + next_raw_item, lib/src/archive/read.rs:22-66; TryFrom<RawEntry>, lib/src/entry.rs:757-885; read_chunk with its
// mandatory CRC check, lib/src/io.rs:117-149; decrypt_reader / decompress_reader, lib/src/entry/read.rs:59-104,171-190).
// The chunk walk and the small chunks' CRCs are host work; the data chunks' CRC-32 (k_frame in verify mode), the gather of every
// entry's data pieces into one stream (k_gather), AES-CTR decryption and the zstd / deflate decoding run on the device.
// The name an entry is handed out under: EntryHeader::path() (lib/src/entry/header.rs:91-94,143-147) -- the FHED bytes must be UTF-8
// (InvalidData otherwise), and what callers see is the SANITISED form (EntryName::sanitize: no root, no "." / "..", so a crafted
// "../../etc/x" or "/abs" cannot leave the extraction directory).  The callback takes a C string, so an embedded NUL is rejected too.
static bool utf8_ok(const uint8_t *p, size_t n) {
    for (size_t i = 0; i < n;) {
        const uint8_t b = p[i];
        size_t k; uint32_t cp;
        if (b < 0x80) { i++; continue; }
        else if ((b & 0xE0) == 0xC0) { k = 1; cp = b & 0x1F; }
        else if ((b & 0xF0) == 0xE0) { k = 2; cp = b & 0x0F; }
        else if ((b & 0xF8) == 0xF0) { k = 3; cp = b & 0x07; }
        else return false;
        for (size_t j = 1; j <= k; j++) { if (i + j >= n || (p[i + j] & 0xC0) != 0x80) return false; cp = (cp << 6) | (p[i + j] & 0x3F); }
        if ((k == 1 && cp < 0x80) || (k == 2 && cp < 0x800) || (k == 3 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
        i += k + 1;
    }
    return true;
}
static int entry_path(pna_gpu_ctx *c, const std::string &raw, std::string &out) {
    if (memchr(raw.data(), 0, raw.size())) return fail(c, PNA_E_INVAL, "entry name contains a NUL byte");
    if (!utf8_ok((const uint8_t *)raw.data(), raw.size())) return fail(c, PNA_E_INVAL, "entry name is not valid UTF-8");
    out = pna::pna_sanitize_name(raw.data(), raw.size());
    return PNA_OK;
}
namespace {
struct XPiece { uint64_t off; uint32_t len; };
struct XEntry {
    std::string name; int kind = 0, compression = 0, encryption = 0, cipher_mode = 0;
    bool has_size = false; uint64_t raw_size = 0; std::string phsf;
    std::vector<XPiece> pieces; uint64_t stream_len = 0;
    uint64_t pk_off = 0, pay_len = 0, raw_off = 0;            // payload (prefix stripped) in the packed buffer; decoded bytes in the raw buffer
    std::vector<uint8_t> fhed;                                 // FHED body: the GCM stream key is bound to it
    uint32_t gcm_seg = 0;                                      // GCM STREAM: segment size of the stream header
    size_t d0 = 0, d1 = 0;                                     // its FDAT chunks in the descriptor list
    uint64_t lo = 0, hi = 0;                                   // archive bytes [lo, hi) that hold its data chunks
};
typedef std::vector<std::pair<std::string, std::vector<uint8_t>>> XKeys;
static uint32_t max_chunk_len(const std::vector<FrameDesc> &v) {       // the longest data chunk of a list (0: none below 16 380 bytes, the wave-per-chunk CRC kernel's limit)
    uint32_t m = 1;
    for (const FrameDesc &d : v) { if (d.payload_len > 16380u) return 0u; m = std::max(m, d.payload_len); }
    return m;
}
struct XSolid {                                                // SHED [PHSF] SDAT* SEND -- lib/src/entry.rs:465-484,567-583
    int compression = 0, encryption = 0, cipher_mode = 0; std::string phsf;
    std::vector<XPiece> pieces; uint64_t stream_len = 0;
    size_t order = 0;                                          // number of normal entries in front of it
    uint64_t pk_off = 0, pay_len = 0;
    std::vector<uint8_t> shed;                                 // SHED body: the GCM stream key is bound to it (entry_context, lib/src/cipher/aead.rs:167-190)
    uint32_t gcm_seg = 0;
    size_t s0 = 0, s1 = 0;                                     // its SDAT chunks in the descriptor list
    uint64_t lo = 0, hi = 0;
};
// an encrypted data stream of the archive, a normal entry's or a solid entry's: what the cipher stage needs of either
struct XCipherStream {
    const std::string *phsf; int mode; const std::vector<XPiece> *pieces; uint64_t stream_len, pk_off; uint64_t *pay_len; uint32_t gcm_seg;
    const char *htype; const std::vector<uint8_t> *hdr; uint8_t iv[16];
};
uint32_t rd_be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int b64_val(char ch) {
    if (ch >= 'A' && ch <= 'Z') return ch - 'A'; if (ch >= 'a' && ch <= 'z') return ch - 'a' + 26;
    if (ch >= '0' && ch <= '9') return ch - '0' + 52; if (ch == '+') return 62; if (ch == '/') return 63; return -1;
}
bool b64_decode_nopad(const std::string &s, std::vector<uint8_t> &out) {
    uint32_t acc = 0; int bits = 0;
    for (char ch : s) { const int v = b64_val(ch); if (v < 0) return false; acc = (acc << 6) | (uint32_t)v; bits += 6; if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); } }
    return true;
}
}

// what a window leaves behind for later: `issue` starts the D2H copy of its decoded entries (called by the NEXT window once its own bytes
// are on the device, so the copy runs next to that window's kernels), `deliver` waits for it and hands the entries out
struct XDeferred { std::function<int()> issue, deliver; bool issued = false; explicit operator bool() const { return (bool)deliver; } };
static int extract_window(pna_gpu_ctx *c, const uint8_t *a, size_t archive_len, const void *password, size_t password_len, pna_entry_fn cb, void *user,
                          std::vector<XEntry> &ents, std::vector<FrameDesc> &dchunks, std::vector<FrameDesc> &schunks, std::vector<XSolid> &solids,
                          XKeys &keys, size_t &index, int slot, XDeferred *later, XDeferred *prev);

extern "C" int pna_gpu_extract_archive_host(pna_gpu_ctx *c, const void *archive, size_t archive_len, const void *password, size_t password_len,
                                            pna_entry_fn cb, void *user) {
    if (!c || !archive || !cb || (!password && password_len)) return fail(c, PNA_E_INVAL, "null argument");
    const uint8_t *a = (const uint8_t *)archive;
    static const uint8_t sig[8] = {0x89, 0x50, 0x4E, 0x41, 0x0D, 0x0A, 0x1A, 0x0A};
    if (archive_len < 8 + 20 + 12 || memcmp(a, sig, 8) != 0) return fail(c, PNA_E_INVAL, "not a PNA archive");
    // ---- 1. chunk walk (host): structure, small-chunk CRCs, data-chunk descriptors
    std::vector<XEntry> ents; std::vector<FrameDesc> dchunks, schunks;
    std::vector<XSolid> solids; XSolid scur; bool in_solid = false;
    XEntry cur; bool in_entry = false, seen_ahed = false, ended = false;
    size_t pos = 8;
    while (pos < archive_len) {
        if (archive_len - pos < 12) return fail(c, PNA_E_INVAL, "truncated chunk header");
        const uint32_t len = rd_be32(a + pos); const uint8_t *ty = a + pos + 4, *data = a + pos + 8;
        if (archive_len - pos - 12 < len) return fail(c, PNA_E_INVAL, "truncated chunk body");
        const bool is_fdat = memcmp(ty, "FDAT", 4) == 0, is_sdat = memcmp(ty, "SDAT", 4) == 0;
        if (is_fdat || is_sdat) { if (len >= 0xFFFFFFF0u) return fail(c, PNA_E_INVAL, "data chunk too long"); (is_fdat ? dchunks : schunks).push_back(FrameDesc{pos, len, 0, 8, 0}); }
        else if (pna_crc32(pna_crc32(0, ty, 4), data, len) != rd_be32(data + len)) return fail(c, PNA_E_INVAL, "chunk CRC mismatch");
        if (!seen_ahed) {
            if (memcmp(ty, "AHED", 4) != 0 || len != 8 || data[0] != 0) return fail(c, PNA_E_INVAL, "first chunk must be AHED (major version 0)");
            seen_ahed = true;
        } else if (memcmp(ty, "AEND", 4) == 0) { ended = true; break; }
        else if (memcmp(ty, "ANXT", 4) == 0) return fail(c, PNA_E_UNSUPPORTED, "multipart archives are not read by this driver");
        else if (memcmp(ty, "SHED", 4) == 0) {
            if (in_entry || in_solid || len != 5 || data[0] != 0 || data[1] != 0) return fail(c, PNA_E_INVAL, "bad solid header");
            scur = XSolid(); in_solid = true; scur.order = ents.size(); scur.s0 = schunks.size(); scur.lo = pos;
            scur.compression = data[2]; scur.encryption = data[3]; scur.cipher_mode = data[4]; scur.shed.assign(data, data + len);
        } else if (in_solid) {
            if (memcmp(ty, "SDAT", 4) == 0) { scur.pieces.push_back(XPiece{pos + 8, len}); scur.stream_len += len; }
            else if (memcmp(ty, "PHSF", 4) == 0) scur.phsf.assign((const char *)data, len);
            else if (memcmp(ty, "SEND", 4) == 0) { scur.s1 = schunks.size(); scur.hi = pos + 12; solids.push_back(std::move(scur)); in_solid = false; }
            else if (!(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "unknown critical chunk in a solid entry");
        }
        else if (memcmp(ty, "FHED", 4) == 0) {
            if (in_entry || len < 6 || data[0] != 0 || data[1] != 0) return fail(c, PNA_E_INVAL, "bad entry header");
            cur = XEntry(); in_entry = true; cur.d0 = dchunks.size(); cur.lo = pos;
            cur.kind = data[2]; cur.compression = data[3]; cur.encryption = data[4]; cur.cipher_mode = data[5];
            cur.name.assign((const char *)data + 6, len - 6); cur.fhed.assign(data, data + len);
        } else if (!in_entry) { if (!(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "unknown critical chunk between entries"); }
        else if (is_fdat) { cur.pieces.push_back(XPiece{pos + 8, len}); cur.stream_len += len; }
        else if (memcmp(ty, "fSIZ", 4) == 0) { if (len > 8) return fail(c, PNA_E_UNSUPPORTED, "entry beyond 2^64 bytes"); cur.has_size = true; cur.raw_size = 0; for (uint32_t i = 0; i < len; i++) cur.raw_size = (cur.raw_size << 8) | data[i]; }
        else if (memcmp(ty, "PHSF", 4) == 0) cur.phsf.assign((const char *)data, len);
        else if (memcmp(ty, "FEND", 4) == 0) { cur.d1 = dchunks.size(); cur.hi = pos + 12; ents.push_back(std::move(cur)); in_entry = false; }
        else if (!(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "unknown critical chunk");      // chunk/types.rs: bit 5 of byte 0 clear = critical
        pos += 12 + (size_t)len;
    }
    if (!ended || in_entry || in_solid) return fail(c, PNA_E_INVAL, "archive not terminated by AEND");
    // ---- 2. windows: a run of entries whose archive bytes, packed payloads and decoded bytes stay within a few GiB each goes through the
    // device at a time (an archive of any size in host memory against a bounded footprint in HBM); a solid entry is a window of its own
    XKeys keys; size_t index = 0, si = 0, w0 = 0;
    const size_t n_all = ents.size();
    const uint64_t WIN = (uint64_t)c->tun.extract_win_mib << 20;  // 1 GiB of archive (and at most 3 GiB decoded) per window by default: small enough to pipeline, large enough for the kernels
    auto rebase_run = [&](size_t e0, size_t e1, std::vector<XEntry> &we, std::vector<FrameDesc> &wd, uint64_t base) {
        we.assign(std::make_move_iterator(ents.begin() + e0), std::make_move_iterator(ents.begin() + e1));
        wd.assign(dchunks.begin() + we.front().d0, dchunks.begin() + we.back().d1);
        for (auto &f : wd) f.arc_off -= base;
        for (auto &e : we) for (auto &p : e.pieces) p.off -= base;
    };
    // Windows are pipelined against each other: the decoded entries of window k travel to the host (their own stream, their own pair of
    // buffers) while window k + 1 is copied in and decoded; window k's entries are handed out once k + 1 has been launched, before k + 1's.
    XDeferred pending; int slot = 0;
    auto finish_pending = [&]() -> int {
        if (!pending) return PNA_OK;
        XDeferred f = std::move(pending); pending = XDeferred();
        if (!f.issued) { const int r = f.issue(); if (r) return r; }
        return f.deliver();
    };
    auto run_window = [&](const uint8_t *wa, size_t wlen, std::vector<XEntry> &we, std::vector<FrameDesc> &wd, std::vector<FrameDesc> &ws, std::vector<XSolid> &wso) -> int {
        XDeferred cur;
        int rc = extract_window(c, wa, wlen, password, password_len, cb, user, we, wd, ws, wso, keys, index, slot, &cur, pending ? &pending : nullptr);
        const int rc2 = finish_pending();
        if (rc == PNA_OK) rc = rc2;
        if (rc != PNA_OK) { (void)hipDeviceSynchronize(); return rc; }
        pending = std::move(cur); slot ^= 1;
        return PNA_OK;
    };
    while (w0 < n_all || si < solids.size()) {
        std::vector<XEntry> we; std::vector<FrameDesc> wd, ws; std::vector<XSolid> wso;
        if (si < solids.size() && solids[si].order <= w0) {
            XSolid so = std::move(solids[si]);
            const uint64_t base = so.lo;
            ws.assign(schunks.begin() + so.s0, schunks.begin() + so.s1);
            for (auto &f : ws) f.arc_off -= base;
            for (auto &p : so.pieces) p.off -= base;
            const uint64_t span = so.hi - base;
            so.order = 0; wso.push_back(std::move(so)); si++;
            int rc = run_window(a + base, (size_t)span, we, wd, ws, wso);
            if (rc) return rc;
            continue;
        }
        size_t w1 = w0; uint64_t raw = 0, pk = 0;
        const size_t stop = si < solids.size() ? std::min(n_all, solids[si].order) : n_all;
        while (w1 < stop) {
            const XEntry &e = ents[w1];
            const uint64_t r = e.has_size ? e.raw_size : 0;
            if (w1 > w0 && (e.hi - ents[w0].lo > WIN || raw + r > 3 * WIN || pk + e.stream_len > WIN)) break;
            raw += r; pk += e.stream_len; w1++;
        }
        const uint64_t base = ents[w0].lo, span = ents[w1 - 1].hi - base;
        rebase_run(w0, w1, we, wd, base);
        int rc = run_window(a + base, (size_t)span, we, wd, ws, wso);
        if (rc) return rc;
        w0 = w1;
    }
    return finish_pending();
}

// One window of the driver above: `a` / archive_len are the window's bytes, every offset in ents / dchunks / schunks / solids is relative to it.
static int extract_window(pna_gpu_ctx *c, const uint8_t *a, size_t archive_len, const void *password, size_t password_len, pna_entry_fn cb, void *user,
                          std::vector<XEntry> &ents, std::vector<FrameDesc> &dchunks, std::vector<FrameDesc> &schunks, std::vector<XSolid> &solids,
                          XKeys &keys, size_t &index, int slot, XDeferred *later, XDeferred *prev) {
    const size_t n = ents.size();
    // keys (one derivation per distinct PHSF string), layout of the packed payloads and of the decoded entries
    auto key_for = [&](const std::string &phsf, const uint8_t **out) -> int {
        for (auto &k : keys) if (k.first == phsf) { *out = k.second.data(); return PNA_OK; }
        // "$pbkdf2-sha256$i=<rounds>,l=<len>$<salt>" (derive_password_hash, lib/src/hash.rs:47-88); Argon2 strings need the Rust host
        if (phsf.rfind("$argon2", 0) == 0) {
            // "$argon2id$v=19$m=<KiB>,t=<passes>,p=<lanes>$<salt>" (argon2 0.5: Params::try_from(&PasswordHash), lib/src/hash.rs:56-70)
            int kind = -1; size_t p1 = 0;
            if (phsf.rfind("$argon2id$", 0) == 0) { kind = 2; p1 = 10; } else if (phsf.rfind("$argon2i$", 0) == 0) { kind = 1; p1 = 9; } else if (phsf.rfind("$argon2d$", 0) == 0) { kind = 0; p1 = 9; }
            if (kind < 0) return fail(c, PNA_E_INVAL, "malformed PHSF");
            if (phsf.compare(p1, 2, "v=") == 0) { const size_t q = phsf.find('$', p1); if (q == std::string::npos || strtoul(phsf.c_str() + p1 + 2, nullptr, 10) != 19) return fail(c, PNA_E_UNSUPPORTED, "argon2 version other than 0x13"); p1 = q + 1; }
            const size_t p2 = phsf.find('$', p1);
            if (p2 == std::string::npos) return fail(c, PNA_E_INVAL, "malformed PHSF");
            uint32_t m = 19456, t = 2, lanes = 1;                 // argon2 0.5 defaults
            const std::string prm = phsf.substr(p1, p2 - p1);
            for (size_t q = 0; q < prm.size();) {
                const size_t e2 = prm.find(',', q); const std::string kv = prm.substr(q, e2 == std::string::npos ? std::string::npos : e2 - q);
                if (kv.size() > 2 && kv[1] == '=') {
                    char *endp = nullptr; const unsigned long long v = strtoull(kv.c_str() + 2, &endp, 10);
                    if (!endp || *endp || endp == kv.c_str() + 2) return fail(c, PNA_E_INVAL, "malformed argon2 parameter in PHSF");
                    // the parameters come from an untrusted archive: refuse costs that only serve to stall / exhaust the host
                    if ((kv[0] == 'm' && v > (4ull << 20)) || (kv[0] == 't' && v > 64) || (kv[0] == 'p' && v > 256)) return fail(c, PNA_E_UNSUPPORTED, "argon2 cost beyond the accepted maximum (m <= 4 GiB, t <= 64, p <= 256)");
                    if (kv[0] == 'm') m = (uint32_t)v; else if (kv[0] == 't') t = (uint32_t)v; else if (kv[0] == 'p') lanes = (uint32_t)v;
                }
                if (e2 == std::string::npos) break; q = e2 + 1;
            }
            std::vector<uint8_t> salt;
            std::string sb = phsf.substr(p2 + 1); const size_t p3 = sb.find('$'); if (p3 != std::string::npos) sb.resize(p3);
            if (!b64_decode_nopad(sb, salt)) return fail(c, PNA_E_INVAL, "malformed PHSF");
            std::vector<uint8_t> key(32);
            int rc = pna_kdf_argon2(kind, password, password_len, salt.data(), salt.size(), t, m, lanes, key.data(), 32);
            if (rc) return fail(c, rc, "key derivation failed (argon2 parameters)");
            keys.emplace_back(phsf, std::move(key)); *out = keys.back().second.data();
            return PNA_OK;
        }
        if (phsf.rfind("$pbkdf2-sha256$", 0) != 0) return fail(c, PNA_E_UNSUPPORTED, "password hash other than argon2 / pbkdf2-sha256");
        const size_t p1 = 15, p2 = phsf.find('$', p1);
        if (p2 == std::string::npos) return fail(c, PNA_E_INVAL, "malformed PHSF");
        uint32_t rounds = 600000;
        const std::string prm = phsf.substr(p1, p2 - p1);
        const size_t ip = prm.find("i=");
        if (ip != std::string::npos) {
            char *endp = nullptr; const unsigned long long v = strtoull(prm.c_str() + ip + 2, &endp, 10);
            if (!endp || (*endp && *endp != ',') || v == 0) return fail(c, PNA_E_INVAL, "malformed pbkdf2 round count in PHSF");
            if (v > 10000000ull) return fail(c, PNA_E_UNSUPPORTED, "pbkdf2 round count beyond the accepted maximum (10 000 000)");
            rounds = (uint32_t)v;
        }
        std::vector<uint8_t> salt;
        std::string sb = phsf.substr(p2 + 1); const size_t p3 = sb.find('$'); if (p3 != std::string::npos) sb.resize(p3);
        if (!b64_decode_nopad(sb, salt) || rounds == 0) return fail(c, PNA_E_INVAL, "malformed PHSF");
        std::vector<uint8_t> key(32);
        int rc = pna_kdf_pbkdf2_sha256(password, password_len, salt.data(), salt.size(), rounds, key.data(), 32, nullptr, 0);
        if (rc) return fail(c, rc, "key derivation failed");
        keys.emplace_back(phsf, std::move(key)); *out = keys.back().second.data();
        return PNA_OK;
    };
    uint64_t pk_total = 0, raw_total = 0;
    std::vector<PlaceDescH> places; std::vector<XCipherStream> enc_list, gcm_list; std::vector<size_t> nosize_idx;
    std::vector<std::vector<uint8_t>> nosize_data;
    // A data stream (the concatenated FDAT / SDAT bodies) is laid into the packed buffer at pk_off with its cipher prefix stripped: CTR / CBC lose the
    // IV, a GCM STREAM its header and the segments' tags (only the ciphertext is gathered).  Sets pay_len (and gcm_seg) and registers the stream
    // with the cipher stage.
    auto plan_stream = [&](const std::vector<XPiece> &pieces, uint64_t stream_len, int encryption, int cipher_mode, const std::string &phsf,
                           const char *htype, const std::vector<uint8_t> &hdr, uint64_t pk_off, uint64_t &pay_len, uint32_t &gcm_seg) -> int {
        auto stream_read = [&](uint64_t lo, uint64_t n2, uint8_t *out) {        // (the prefix may span data pieces: prepend_data_prefix makes it a piece of its own)
            uint64_t at2 = 0, got = 0;
            for (const XPiece &p : pieces) { for (uint32_t k = 0; k < p.len && got < n2; k++) if (at2 + k >= lo) out[got++] = a[p.off + k]; at2 += p.len; if (got >= n2) break; }
        };
        auto stream_place = [&](uint64_t lo, uint64_t hi, uint64_t dst) {
            uint64_t at2 = 0;
            for (const XPiece &p : pieces) {
                const uint64_t s0 = std::max<uint64_t>(lo, at2), s1 = std::min<uint64_t>(hi, at2 + p.len);
                for (uint64_t k = s0; k < s1; k += (1u << 20)) places.push_back(PlaceDescH{p.off + (k - at2), dst + (k - lo), (uint32_t)std::min<uint64_t>(1u << 20, s1 - k), 0});
                at2 += p.len;
            }
        };
        if (encryption == PNA_ENC_NONE) { pay_len = stream_len; stream_place(0, stream_len, pk_off); return PNA_OK; }
        if (encryption != PNA_ENC_AES) return fail(c, PNA_E_UNSUPPORTED, "only AES entries are decrypted by this driver");
        if (!password) return fail(c, PNA_E_INVAL, "encrypted entry and no password");
        if (phsf.empty()) return fail(c, PNA_E_INVAL, "`PHSF` chunk not found");
        XCipherStream cs{&phsf, cipher_mode, &pieces, stream_len, pk_off, &pay_len, 0, htype, &hdr, {0}};
        if (cipher_mode == PNA_MODE_CTR || cipher_mode == PNA_MODE_CBC) {
            if (stream_len < 16) return fail(c, PNA_E_INVAL, "data stream shorter than the IV");
            stream_read(0, 16, cs.iv);
            pay_len = stream_len - 16;
            stream_place(16, stream_len, pk_off);
            enc_list.push_back(cs);
        } else if (cipher_mode == PNA_MODE_GCM) {
            // stream header, then segments of (segment size + 16-byte tag), the last one shorter: only the ciphertext is gathered
            if (stream_len < 75 + 16) return fail(c, PNA_E_INVAL, "datastream shorter than the stream header");
            uint8_t hd[75]; stream_read(0, 75, hd);
            gcm_seg = rd_be32(hd + 39);
            if (gcm_seg == 0 || gcm_seg > (64u << 20)) return fail(c, PNA_E_INVAL, "GCM segment size out of range");
            cs.gcm_seg = gcm_seg;
            uint64_t rest = stream_len - 75, at2 = 75, outp = pk_off;
            while (rest) {
                const uint64_t segl = std::min<uint64_t>(rest, (uint64_t)gcm_seg + 16);
                if (segl < 16) return fail(c, PNA_E_INVAL, "GCM segment shorter than a tag");
                stream_place(at2, at2 + segl - 16, outp);
                outp += segl - 16; at2 += segl; rest -= segl;
            }
            pay_len = outp - pk_off;
            gcm_list.push_back(cs);
        } else return fail(c, PNA_E_UNSUPPORTED, "unknown cipher mode");
        return PNA_OK;
    };
    for (size_t i = 0; i < n; i++) {
        XEntry &e = ents[i];
        if (e.compression != PNA_ALGO_STORE && e.compression != PNA_ALGO_ZSTD && e.compression != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "compression method not decoded on the device (xz)");
        e.pk_off = pk_total;
        { const int r = plan_stream(e.pieces, e.stream_len, e.encryption, e.cipher_mode, e.phsf, "FHED", e.fhed, e.pk_off, e.pay_len, e.gcm_seg); if (r) return r; }
        pk_total = (pk_total + e.pay_len + 15) & ~(uint64_t)15;
        if (e.compression != PNA_ALGO_STORE) {
            // fSIZ is optional (older writers omit it): the payload is then decoded like a solid stream, its size found by the decoder
            if (!e.has_size) nosize_idx.push_back(i);
            else {
                // fSIZ comes from the archive: a size no payload of this length can decode to (deflate tops out at 1032 : 1, zstd at a few
                // thousand : 1 through RLE blocks) is damage, not a reason to ask the device for exabytes
                if (e.raw_size > (1ull << 40) || e.raw_size / 65536 > e.pay_len + 1) return fail(c, PNA_E_INVAL, "fSIZ is out of proportion to the entry's data");
                e.raw_off = raw_total; raw_total = (raw_total + e.raw_size + 15) & ~(uint64_t)15;
                if (raw_total > (1ull << 42)) return fail(c, PNA_E_NOMEM, "archive decodes to more than this driver takes in one call");
            }
        }
    }
    for (XSolid &so : solids) {
        if (so.compression != PNA_ALGO_STORE && so.compression != PNA_ALGO_ZSTD && so.compression != PNA_ALGO_DEFLATE) return fail(c, PNA_E_UNSUPPORTED, "solid stream: compression method not decoded on the device (xz)");
        so.pk_off = pk_total;
        { const int r = plan_stream(so.pieces, so.stream_len, so.encryption, so.cipher_mode, so.phsf, "SHED", so.shed, so.pk_off, so.pay_len, so.gcm_seg); if (r) return r; }
        pk_total = (pk_total + so.pay_len + 15) & ~(uint64_t)15;
    }
    // ---- 3. device: upload, data-chunk CRCs, gather, decrypt, decode
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    static const bool xtrace = getenv("PNA_EXTRACT_TRACE") != nullptr;   // per-window phase times on stderr
    const auto xt0 = std::chrono::steady_clock::now();
    auto xms = [&](std::chrono::steady_clock::time_point a2) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a2).count(); };
    int rc = ensure_crc(c); if (rc) return rc;
    if (c->x_arc.ensure(archive_len + 64) || c->x_pk.ensure(pk_total + 8192) || c->x_raw[slot].ensure(raw_total + 64) || c->x_flag.ensure(64) ||
        c->x_desc.ensure(dchunks.size() * sizeof(FrameDesc) + 16) || c->x_place.ensure(places.size() * sizeof(PlaceDescH) + 16)) return fail(c, PNA_E_NOMEM, "extract workspace");
    HIPCHK(c, hipMemcpyAsync(c->x_arc.p, a, archive_len, hipMemcpyHostToDevice, st));
    const uint32_t flag0[2] = {0u, 0xFFFFFFFFu};
    HIPCHK(c, hipMemcpyAsync(c->x_flag.p, flag0, 8, hipMemcpyHostToDevice, st));
    if (!dchunks.empty()) {
        HIPCHK(c, hipMemcpyAsync(c->x_desc.p, dchunks.data(), dchunks.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        launch_frame_verify((const FrameDesc *)c->x_desc.p, (uint32_t)dchunks.size(), (const CrcTabs *)c->crc_tabs.p, (const uint8_t *)c->x_arc.p,
                            (uint64_t)c->x_arc.cap & ~(uint64_t)15, "FDAT", (uint32_t *)c->x_flag.p, st, max_chunk_len(dchunks));
    }
    if (!schunks.empty()) {
        if (c->solid_desc.ensure(schunks.size() * sizeof(FrameDesc) + 16)) return fail(c, PNA_E_NOMEM, "extract workspace");
        HIPCHK(c, hipMemcpyAsync(c->solid_desc.p, schunks.data(), schunks.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
        launch_frame_verify((const FrameDesc *)c->solid_desc.p, (uint32_t)schunks.size(), (const CrcTabs *)c->crc_tabs.p, (const uint8_t *)c->x_arc.p,
                            (uint64_t)c->x_arc.cap & ~(uint64_t)15, "SDAT", (uint32_t *)c->x_flag.p, st, max_chunk_len(schunks));
    }
    if (!places.empty()) {
        HIPCHK(c, hipMemcpyAsync(c->x_place.p, places.data(), places.size() * sizeof(PlaceDescH), hipMemcpyHostToDevice, st));
        launch_gather(c->x_place.p, (uint32_t)places.size(), (const uint8_t *)c->x_arc.p, (uint8_t *)c->x_pk.p, st);
    }
    uint32_t flag[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(flag, c->x_flag.p, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    const double x_in = xms(xt0);
    // This window's bytes are on the device: now the previous window's decoded entries start their way back, next to this window's
    // decryption and decoding.  (Issued earlier, the two copies share the link -- H2D of 1 GiB next to D2H of 2.5 GiB took 67 ms, as long
    // as one after the other -- and the kernels would again run with the link idle.)
    if (prev && *prev && !prev->issued) { prev->issued = true; const int r = prev->issue(); if (r) return r; }
    if (flag[0]) { c->err = "data chunk CRC mismatch (" + std::to_string(flag[0]) + " FDAT / SDAT chunks)"; return PNA_E_INVAL; }
    if (!enc_list.empty()) {
        // streams sharing a PHSF string and a mode share the key: one cipher call per group
        std::vector<bool> done(enc_list.size(), false);
        for (size_t j = 0; j < enc_list.size(); j++) {
            if (done[j]) continue;
            const XCipherStream &e0 = enc_list[j];
            const uint8_t *key = nullptr;
            rc = key_for(*e0.phsf, &key); if (rc) return rc;
            std::vector<uint64_t> off, len; std::vector<uint8_t> iv2; std::vector<size_t> who;
            for (size_t k = j; k < enc_list.size(); k++)
                if (!done[k] && *enc_list[k].phsf == *e0.phsf && enc_list[k].mode == e0.mode) {
                    done[k] = true; off.push_back(enc_list[k].pk_off); len.push_back(*enc_list[k].pay_len); who.push_back(k);
                    iv2.insert(iv2.end(), enc_list[k].iv, enc_list[k].iv + 16);
                }
            if (e0.mode == PNA_MODE_CTR) {
                pna_gpu_cipher ci{}; ci.encryption = PNA_ENC_AES; ci.cipher_mode = PNA_MODE_CTR; memcpy(ci.key, key, 32); ci.phsf = ""; ci.ivs = iv2.data();
                rc = pna_gpu_cipher_apply_device(c, &ci, 1, off.size(), c->x_pk.p, off.data(), len.data(), st);
                if (rc) return rc;
            } else {                                              // CBC: DecryptCbcAes256Reader, lib/src/entry/read.rs:77-82
                rc = ensure_aes_dec(c); if (rc) return rc;
                // (a block's plaintext needs its own and the previous ciphertext block only: a long stream -- a solid one -- is cut into units of 16 MiB whose
                // IV is the ciphertext block in front; the padding is read at the end of the stream's last unit)
                constexpr uint64_t CBC_UNIT = 16u << 20;
                std::vector<CipherUnit> units; std::vector<uint8_t> uiv; std::vector<size_t> last_unit(off.size());
                for (size_t q = 0; q < off.size(); q++) {
                    if (len[q] == 0 || (len[q] & 15)) return fail(c, PNA_E_INVAL, "CBC: bad length or padding (wrong password or damaged data)");
                    for (uint64_t o = 0; o < len[q]; o += CBC_UNIT) {
                        units.push_back(CipherUnit{off[q] + o, 0, (uint32_t)std::min<uint64_t>(CBC_UNIT, len[q] - o), (uint32_t)units.size()});
                        last_unit[q] = units.size() - 1;
                    }
                }
                if (c->ci_units.ensure(units.size() * sizeof(CipherUnit) + 16) || c->ci_ivs.ensure(units.size() * 16 + 16) || c->x_plen.ensure(units.size() * 4 + 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
                // the units' IVs: the stream's own for its first unit, else the 16 ciphertext bytes in front of the unit (copied on the device BEFORE the
                // kernel overwrites them: the decryption is in place)
                { size_t u = 0;
                  for (size_t q = 0; q < off.size(); q++)
                      for (uint64_t o = 0; o < len[q]; o += CBC_UNIT, u++) {
                          if (o == 0) HIPCHK(c, hipMemcpyAsync((uint8_t *)c->ci_ivs.p + 16 * u, iv2.data() + 16 * q, 16, hipMemcpyHostToDevice, st));
                          else HIPCHK(c, hipMemcpyAsync((uint8_t *)c->ci_ivs.p + 16 * u, (const uint8_t *)c->x_pk.p + off[q] + o - 16, 16, hipMemcpyDeviceToDevice, st));
                      } }
                AesKey ek, dk; aes256_expand(key, ek); aes256_dec_key(ek, dk);
                std::vector<uint32_t> plen(units.size());
                HIPCHK(c, hipMemcpyAsync(c->ci_units.p, units.data(), units.size() * sizeof(CipherUnit), hipMemcpyHostToDevice, st));
                launch_aes_cbc_dec((const CipherUnit *)c->ci_units.p, (uint32_t)units.size(), (const uint8_t *)c->ci_ivs.p, (const AesDecTabs *)c->aes_dtabs.p,
                                   (uint8_t *)c->x_pk.p, dk, (uint32_t *)c->x_plen.p, st);
                HIPCHK(c, hipMemcpyAsync(plen.data(), c->x_plen.p, units.size() * 4, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, hipStreamSynchronize(st));
                for (size_t q = 0; q < who.size(); q++) {
                    const uint32_t pl = plen[last_unit[q]];
                    if (pl == 0xFFFFFFFFu) return fail(c, PNA_E_INVAL, "CBC: bad length or padding (wrong password or damaged data)");
                    *enc_list[who[q]].pay_len = (len[q] - 1) / CBC_UNIT * CBC_UNIT + pl;
                }
            }
        }
    }
    if (!gcm_list.empty()) {
        // cipher mode 2 (decrypt_reader, (_, CipherMode::GCM): lib/src/entry/read.rs:105-140): key confirmation first -- a wrong password is
        // told apart from tampering --, then every segment's tag (k_gcm_tag in verify mode), then the CTR keystream with the stream keys
        rc = ensure_aes(c); if (rc) return rc;
        std::vector<GcmEntry> gents; std::vector<uint8_t> tags, giv; std::vector<AesKey> gkeys; std::vector<CipherUnit> units;
        for (const XCipherStream &e : gcm_list) {
            const uint8_t *km = nullptr;
            rc = key_for(*e.phsf, &km); if (rc) return rc;
            uint8_t hd[75]; { uint64_t got = 0; for (const XPiece &p : *e.pieces) { for (uint32_t k = 0; k < p.len && got < 75; k++) hd[got++] = a[p.off + k]; if (got >= 75) break; } }
            uint8_t kc[32]; hkdf_sha256_32(km, 32, nullptr, 0, "PNA-KC-v1", 9, kc);
            { uint8_t diff = 0; for (int b = 0; b < 32; b++) diff |= (uint8_t)(kc[b] ^ hd[43 + b]);      // constant time, like the reference's ct_eq
              if (diff) return fail(c, PNA_E_INVAL, "GCM STREAM: key confirmation failed (wrong password)"); }
            uint8_t info[88], ph[32], ks[32];
            memcpy(info, "PNA-STREAM-v1", 13);
            sha256_bytes(e.htype, 4, e.hdr->data(), e.hdr->size(), info + 13);      // (entry_context: the header chunk's type and body, FHED or SHED -- lib/src/cipher/aead.rs:167-190)
            sha256_bytes(e.phsf->data(), e.phsf->size(), nullptr, 0, ph); memcpy(info + 45, ph, 32);
            memcpy(info + 77, hd + 32, 7); memcpy(info + 84, hd + 39, 4);
            hkdf_sha256_32(km, 32, hd, 32, info, 88, ks);
            AesKey rk; aes256_expand(ks, rk);
            uint8_t zero[16] = {0}, hb[16]; aes256_block_host(rk, zero, hb);
            uint64_t rest = e.stream_len - 75, at2 = 75, outp = e.pk_off; uint32_t counter = 0;
            while (rest) {
                const uint64_t segl = std::min<uint64_t>(rest, (uint64_t)e.gcm_seg + 16), ctl = segl - 16;
                const bool fin = segl == rest;
                uint8_t j0[16], eb[16], tag[16];
                memcpy(j0, hd + 32, 7); j0[7] = (uint8_t)(counter >> 24); j0[8] = (uint8_t)(counter >> 16); j0[9] = (uint8_t)(counter >> 8); j0[10] = (uint8_t)counter; j0[11] = fin ? 1 : 0;
                j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;
                aes256_block_host(rk, j0, eb);
                GcmEntry ge{outp, (uint32_t)ctl, 0, {0, 0, 0, 0}, {0, 0, 0, 0}};
                for (int w = 0; w < 4; w++) {
                    ge.h[w] = ((uint32_t)hb[4 * w] << 24) | ((uint32_t)hb[4 * w + 1] << 16) | ((uint32_t)hb[4 * w + 2] << 8) | hb[4 * w + 3];
                    ge.ej0[w] = ((uint32_t)eb[4 * w] << 24) | ((uint32_t)eb[4 * w + 1] << 16) | ((uint32_t)eb[4 * w + 2] << 8) | eb[4 * w + 3];
                }
                { uint64_t p2 = 0, got = 0; const uint64_t lo = at2 + ctl;      // the stored tag, wherever the chunk boundaries fall
                  for (const XPiece &p : *e.pieces) { for (uint32_t k = 0; k < p.len && got < 16; k++) if (p2 + k >= lo) tag[got++] = a[p.off + k]; p2 += p.len; if (got >= 16) break; } }
                const uint32_t idx = (uint32_t)gents.size();
                gents.push_back(ge); tags.insert(tags.end(), tag, tag + 16); gkeys.push_back(rk);
                j0[15] = 2; giv.insert(giv.end(), j0, j0 + 16);
                for (uint64_t o = 0; o < ctl; o += CTR_UNIT) units.push_back(CipherUnit{outp + o, o, (uint32_t)std::min<uint64_t>(CTR_UNIT, ctl - o), idx});
                outp += ctl; at2 += segl; rest -= segl; counter++;
                if (!fin && segl != (uint64_t)e.gcm_seg + 16) return fail(c, PNA_E_INVAL, "GCM STREAM: short non-final segment");
            }
        }
        if (c->ci_gcm.ensure(gents.size() * sizeof(GcmEntry) + 16) || c->x_tags.ensure(tags.size() + 16) || c->ci_keys.ensure(gkeys.size() * sizeof(AesKey) + 16) ||
            c->ci_ivs.ensure(giv.size() + 16) || c->ci_units.ensure(units.size() * sizeof(CipherUnit) + 16)) return fail(c, PNA_E_NOMEM, "cipher workspace");
        HIPCHK(c, hipMemcpyAsync(c->x_flag.p, flag0, 8, hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ci_gcm.p, gents.data(), gents.size() * sizeof(GcmEntry), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->x_tags.p, tags.data(), tags.size(), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ci_keys.p, gkeys.data(), gkeys.size() * sizeof(AesKey), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipMemcpyAsync(c->ci_ivs.p, giv.data(), giv.size(), hipMemcpyHostToDevice, st));
        if (!units.empty()) HIPCHK(c, hipMemcpyAsync(c->ci_units.p, units.data(), units.size() * sizeof(CipherUnit), hipMemcpyHostToDevice, st));
        launch_gcm_verify((const GcmEntry *)c->ci_gcm.p, (uint32_t)gents.size(), (const uint8_t *)c->x_pk.p, (const uint8_t *)c->x_tags.p, (uint32_t *)c->x_flag.p, st);
        HIPCHK(c, hipMemcpyAsync(flag, c->x_flag.p, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));
        if (flag[0]) return fail(c, PNA_E_INVAL, "GCM STREAM: authentication failure (a segment tag does not match)");
        AesKey k0{};
        launch_aes_ctr((const CipherUnit *)c->ci_units.p, (uint32_t)units.size(), (const uint8_t *)c->ci_ivs.p, (const AesTabs *)c->aes_tabs.p, (uint8_t *)c->x_pk.p, k0, (const AesKey *)c->ci_keys.p, st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(st));
    }
    for (int algo : {PNA_ALGO_ZSTD, PNA_ALGO_DEFLATE}) {
        std::vector<uint64_t> so, sl, dof, rl;
        for (const XEntry &e : ents) if (e.compression == algo && e.has_size) { so.push_back(e.pk_off); sl.push_back(e.pay_len); dof.push_back(e.raw_off); rl.push_back(e.raw_size); }
        if (so.empty()) continue;
        rc = pna_gpu_decompress_batch_device(c, algo, so.size(), c->x_pk.p, so.data(), sl.data(), c->x_raw[slot].p, dof.data(), rl.data(), st);
        if (rc) return rc;
    }
    for (size_t i : nosize_idx) {                                 // compatibility path, one decode call per entry
        XEntry &e = ents[i];
        uint32_t nfr = 1; uint64_t got = 0;
        if (e.compression == PNA_ALGO_ZSTD) { rc = pna_gpu_zstd_stream_frames_device(c, c->x_pk.p, e.pk_off, e.pay_len, &nfr, st); if (rc) return rc; }
        const uint64_t cap = nfr > 1 ? (uint64_t)nfr * SEG_SIZE : std::min<uint64_t>(1ull << 30, std::max<uint64_t>(64ull << 20, 64 * e.pay_len));
        if (c->solid_plain.ensure(cap + 8192)) return fail(c, PNA_E_NOMEM, "entry buffer");
        rc = e.compression == PNA_ALGO_ZSTD ? pna_gpu_zstd_decompress_open_device(c, c->x_pk.p, e.pk_off, e.pay_len, c->solid_plain.p, 0, cap, &got, st)
                                            : pna_gpu_inflate_open_device(c, c->x_pk.p, e.pk_off, e.pay_len, c->solid_plain.p, 0, cap, &got, st);
        if (rc) return rc;
        nosize_data.emplace_back((size_t)got);
        if (got) HIPCHK(c, hipMemcpy(nosize_data.back().data(), c->solid_plain.p, got, hipMemcpyDeviceToHost));
        e.raw_size = got; e.raw_off = nosize_data.size() - 1;      // index into nosize_data
    }
    // ---- solid entries: decode a stream of unknown size, walk the inner records
    struct Inner { std::string name; int kind; std::vector<XPiece> pieces; uint64_t len; };
    std::vector<std::vector<Inner>> inner(solids.size());
    std::vector<std::vector<uint8_t>> plain(solids.size());
    for (size_t si = 0; si < solids.size(); si++) {
        XSolid &so = solids[si];
        // (an encrypted stream has been decrypted in place by the cipher stage above, with the normal entries' streams)
        uint64_t plen = so.pay_len; const void *d_plain = (const uint8_t *)c->x_pk.p + so.pk_off;
        if (so.compression != PNA_ALGO_STORE) {
            uint32_t nfr = 1;
            if (so.compression == PNA_ALGO_ZSTD) { rc = pna_gpu_zstd_stream_frames_device(c, c->x_pk.p, so.pk_off, so.pay_len, &nfr, st); if (rc) return rc; }
            // this library's zstd solid streams: frames of 1 MiB; one frame / one zlib stream: a bounded guess of its size
            const uint64_t cap = nfr > 1 ? (uint64_t)nfr * SEG_SIZE : std::min<uint64_t>(1ull << 30, std::max<uint64_t>(64ull << 20, 64 * so.pay_len));
            if (c->solid_plain.ensure(cap + 8192)) return fail(c, PNA_E_NOMEM, "solid stream buffer");
            rc = so.compression == PNA_ALGO_ZSTD ? pna_gpu_zstd_decompress_open_device(c, c->x_pk.p, so.pk_off, so.pay_len, c->solid_plain.p, 0, cap, &plen, st)
                                                 : pna_gpu_inflate_open_device(c, c->x_pk.p, so.pk_off, so.pay_len, c->solid_plain.p, 0, cap, &plen, st);
            if (rc) return rc;
            d_plain = c->solid_plain.p;
        }
        plain[si].resize(plen);
        if (plen) HIPCHK(c, hipMemcpy(plain[si].data(), d_plain, plen, hipMemcpyDeviceToHost));
        // read_next_normal_entry_from_stream over the decoded stream (lib/src/entry.rs:401-424): small chunks checked here, the
        // inner FDAT CRCs on the device over the decoded stream where it stands
        const uint8_t *b = plain[si].data();
        std::vector<FrameDesc> ichunks; Inner ic; bool in_i = false;
        for (size_t q = 0; q < plen;) {
            if (plen - q < 12) return fail(c, PNA_E_INVAL, "solid stream: truncated chunk header");
            const uint32_t len = rd_be32(b + q); const uint8_t *ty = b + q + 4, *data = b + q + 8;
            if (plen - q - 12 < len) return fail(c, PNA_E_INVAL, "solid stream: truncated chunk body");
            const bool fd = memcmp(ty, "FDAT", 4) == 0;
            if (fd) { if (len >= 0xFFFFFFF0u) return fail(c, PNA_E_INVAL, "data chunk too long"); ichunks.push_back(FrameDesc{q, len, 0, 8, 0}); }
            else if (pna_crc32(pna_crc32(0, ty, 4), data, len) != rd_be32(data + len)) return fail(c, PNA_E_INVAL, "solid stream: chunk CRC mismatch");
            if (memcmp(ty, "FHED", 4) == 0) {
                if (in_i || len < 6 || data[0] != 0 || data[1] != 0) return fail(c, PNA_E_INVAL, "solid stream: bad entry header");
                if (data[3] != PNA_ALGO_STORE || data[4] != PNA_ENC_NONE) return fail(c, PNA_E_UNSUPPORTED, "solid stream: inner entry that is not stored");
                ic = Inner(); in_i = true; ic.kind = data[2]; ic.len = 0; ic.name.assign((const char *)data + 6, len - 6);
            } else if (!in_i) { if (!(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "solid stream: unknown critical chunk"); }
            else if (fd) { ic.pieces.push_back(XPiece{q + 8, len}); ic.len += len; }
            else if (memcmp(ty, "FEND", 4) == 0) { inner[si].push_back(std::move(ic)); in_i = false; }
            else if (memcmp(ty, "fSIZ", 4) != 0 && !(ty[0] & 0x20)) return fail(c, PNA_E_INVAL, "solid stream: unknown critical chunk");
            q += 12 + (size_t)len;
        }
        if (in_i) return fail(c, PNA_E_INVAL, "solid stream: dangling chunks");
        if (!ichunks.empty()) {
            if (c->solid_desc.ensure(ichunks.size() * sizeof(FrameDesc) + 16)) return fail(c, PNA_E_NOMEM, "extract workspace");
            HIPCHK(c, hipMemcpyAsync(c->x_flag.p, flag0, 8, hipMemcpyHostToDevice, st));
            HIPCHK(c, hipMemcpyAsync(c->solid_desc.p, ichunks.data(), ichunks.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
            const DevBuf &pb = so.compression != PNA_ALGO_STORE ? c->solid_plain : c->x_pk;
            std::vector<FrameDesc> adj;
            if (so.compression == PNA_ALGO_STORE) {                // descriptors are relative to the stream's start inside the packed buffer
                adj = ichunks; for (auto &f : adj) f.arc_off += so.pk_off;
                HIPCHK(c, hipMemcpyAsync(c->solid_desc.p, adj.data(), adj.size() * sizeof(FrameDesc), hipMemcpyHostToDevice, st));
            }
            launch_frame_verify((const FrameDesc *)c->solid_desc.p, (uint32_t)ichunks.size(), (const CrcTabs *)c->crc_tabs.p, (const uint8_t *)pb.p,
                                (uint64_t)pb.cap & ~(uint64_t)15, "FDAT", (uint32_t *)c->x_flag.p, st, max_chunk_len(ichunks));
            HIPCHK(c, hipMemcpyAsync(flag, c->x_flag.p, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(st));
            if (flag[0]) return fail(c, PNA_E_INVAL, "solid stream: inner FDAT CRC mismatch");
        }
    }
    // ---- 4. back to the host, entries in archive order.  Deferred form (no stored entries in the window): the D2H copy runs on its own
    // stream behind the window's kernels and the hand-out happens later (see the driver); everything it needs moves into `D`.
    bool any_store = false; for (const XEntry &e : ents) any_store |= e.compression == PNA_ALGO_STORE && e.pay_len;
    const bool defer = later != nullptr && !any_store;
    if (xtrace) { (void)hipStreamSynchronize(st); fprintf(stderr, "[pna extract window] %zu entries, %.0f MiB in -> %.0f MiB out: H2D + CRC + gather %.1f ms, decrypt + decode %.1f ms (slot %d, %s)\n", n, archive_len / 1048576.0, raw_total / 1048576.0, x_in, xms(xt0) - x_in, slot, defer ? "deferred hand-out" : "immediate"); }
    if (c->hp_out[slot].ensure(raw_total + 64) || (any_store && c->hp_in[0].ensure(pk_total + 64))) return fail(c, PNA_E_NOMEM, "staging allocation failed");
    const uint64_t raw_bytes = raw_total;
    auto issue = [c, slot, raw_bytes]() -> int {                      // the window's kernels are complete on c->stream when this runs or are ordered before it by x_done
        if (raw_bytes && hipMemcpyAsync(c->hp_out[slot].p, c->x_raw[slot].p, raw_bytes, hipMemcpyDeviceToHost, c->x_cp) != hipSuccess) return fail(c, PNA_E_HIP, "D2H copy failed");
        return hipEventRecord(c->x_ev[slot], c->x_cp) == hipSuccess ? PNA_OK : fail(c, PNA_E_HIP, "D2H copy failed");
    };
    if (defer) {
        if (!c->x_cp) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->x_cp, hipStreamNonBlocking));
            for (auto &e : c->x_ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->x_done, hipEventDisableTiming));
        }
        HIPCHK(c, hipEventRecord(c->x_done, st));
        HIPCHK(c, hipStreamWaitEvent(c->x_cp, c->x_done, 0));
    } else {
        if (raw_total) HIPCHK(c, hipMemcpyAsync(c->hp_out[slot].p, c->x_raw[slot].p, raw_total, hipMemcpyDeviceToHost, st));
        if (any_store) HIPCHK(c, hipMemcpyAsync(c->hp_in[0].p, c->x_pk.p, pk_total, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    struct Deliver {
        std::vector<XEntry> ents; std::vector<size_t> solid_order; std::vector<std::vector<Inner>> inner; std::vector<std::vector<uint8_t>> plain, nosize_data;
        size_t index0 = 0;
    };
    auto D = std::make_shared<Deliver>();
    D->index0 = index;
    for (const XSolid &so : solids) D->solid_order.push_back(so.order);
    index += n; for (const auto &v : inner) index += v.size();
    D->ents = std::move(ents); D->inner = std::move(inner); D->plain = std::move(plain); D->nosize_data = std::move(nosize_data);
    const uint8_t *raw_host = (const uint8_t *)c->hp_out[slot].p, *pk_host = (const uint8_t *)c->hp_in[0].p;
    hipEvent_t wait_ev = defer ? c->x_ev[slot] : nullptr;
    auto deliver = [c, cb, user, D, raw_host, pk_host, wait_ev]() -> int {
        const auto dt0 = std::chrono::steady_clock::now();
        if (wait_ev && hipEventSynchronize(wait_ev) != hipSuccess) return fail(c, PNA_E_HIP, "D2H copy failed");
        const double dwait = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dt0).count();
        struct Tr { double w; std::chrono::steady_clock::time_point t; ~Tr() { if (getenv("PNA_EXTRACT_TRACE")) fprintf(stderr, "[pna extract hand-out] waited %.1f ms for the D2H copy, callbacks %.1f ms\n", w, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count()); } } tr{dwait, std::chrono::steady_clock::now()};
        size_t idx = D->index0, si = 0;
        const size_t n = D->ents.size();
        std::vector<uint8_t> joined;
        auto deliver_solids = [&](size_t upto) -> int {
            for (; si < D->solid_order.size() && D->solid_order[si] <= upto; si++)
                for (const Inner &ie : D->inner[si]) {
                    const uint8_t *d = D->plain[si].data();
                    if (ie.pieces.size() == 1) d += ie.pieces[0].off;
                    else { joined.clear(); for (const XPiece &p : ie.pieces) joined.insert(joined.end(), d + p.off, d + p.off + p.len); d = joined.data(); }
                    std::string path; { const int rp = entry_path(c, ie.name, path); if (rp) return rp; }
                    if (cb(user, idx++, path.c_str(), ie.kind, ie.len ? d : nullptr, (size_t)ie.len) != 0) return fail(c, PNA_E_SINK, "entry callback failed");
                }
            return PNA_OK;
        };
        for (size_t i = 0; i < n; i++) {
            int rc = deliver_solids(i); if (rc) return rc;
            const XEntry &e = D->ents[i];
            const uint8_t *d = e.compression == PNA_ALGO_STORE ? pk_host + e.pk_off
                             : (e.has_size ? raw_host + e.raw_off : D->nosize_data[(size_t)e.raw_off].data());
            const size_t l = e.compression == PNA_ALGO_STORE ? (size_t)e.pay_len : (size_t)e.raw_size;
            if (e.compression == PNA_ALGO_STORE && e.has_size && e.raw_size != e.pay_len) return fail(c, PNA_E_INVAL, "stored entry: fSIZ differs from the data length");
            std::string path; { const int rp = entry_path(c, e.name, path); if (rp) return rp; }
            if (cb(user, idx++, path.c_str(), e.kind, d, l) != 0) return fail(c, PNA_E_SINK, "entry callback failed");
        }
        return deliver_solids(n);
    };
    if (defer) { later->issue = issue; later->deliver = deliver; later->issued = false; return PNA_OK; }
    return deliver();
}

