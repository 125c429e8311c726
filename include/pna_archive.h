/*
 * include/pna_archive.h -- C ABI of the host-side PNA container writer in libpna_gpu.so.
 *
 * Restates, byte for byte, the writer half of libpna that the compression path feeds
 * (reference paths relative to /root/reference):
 *   Archive::write_header / add_entry / finalize          lib/src/archive/write.rs:92-101,368-370,438-440
 *   NormalEntry::write_chunks_to (FHED fSIZ FDAT* FEND)   lib/src/entry.rs:888-913
 *   write_chunk (len BE | type | data | crc32 BE)         lib/src/io.rs:183-197, lib/src/format/chunk.rs:7-12
 *   FlattenWriter splitting of the payload into FDATs     lib/src/util/io.rs:60-77
 *   Archive::write_solid_header / SolidArchive            lib/src/archive/write.rs:443-470,545-548,575-580,716-727
 *   create_archive_file driver                            cli/src/command/create.rs:575-635
 * Compression itself is done by the kernels behind pna_gpu.h; PNA_ALGO_STORE needs no GPU.
 */
#ifndef PNA_ARCHIVE_H
#define PNA_ARCHIVE_H
#include <stddef.h>
#include <stdint.h>
#include "pna_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* CRC-32 (IEEE) as crc32fast computes it for chunk_crc(type || data); start with crc = 0. */
uint32_t pna_crc32(uint32_t crc, const void *buf, size_t len);

typedef struct pna_archive pna_archive;

/* Archive::write_header: emits the signature and AHED(archive_number) into the sink at once. */
int  pna_archive_new(pna_sink_fn sink, void *user, uint32_t archive_number, pna_archive **out);
/* Archive::add_entry for a file entry whose payload is ALREADY the compressed stream (what FileEntryBuilder::build
 * produced).  compression is Compression::to_byte(); raw_size < 0 omits fSIZ; max_chunk_size 0 = u32::MAX. */
int  pna_archive_add_file(pna_archive *a, const char *name, int compression, int64_t raw_size,
                          const void *payload, size_t payload_len, uint32_t max_chunk_size);
/* Directory entry: FHED(kind 1, store) FEND, no fSIZ/FDAT (lib/src/entry/builder/dir.rs:52-54). */
int  pna_archive_add_dir(pna_archive *a, const char *name);
/* Solid entry from an already compressed stream: SHED SDAT* SEND; each piece becomes one SDAT chunk. */
int  pna_archive_add_solid(pna_archive *a, int compression, const void *const *pieces, const size_t *piece_len, size_t n_pieces);
/* Serialise one inner STORE entry (FHED fSIZ FDAT FEND as raw chunk bytes) -- what SolidArchive::add_entry feeds
 * the compressor (lib/src/archive/write.rs:575-580).  Returns the number of bytes written to dst (cap checked), or
 * the required size when dst is NULL. */
size_t pna_archive_inner_entry_bytes(const char *name, const void *data, size_t len, void *dst, size_t cap);
/* Archive::finalize: AEND.  Consumes the handle. */
int  pna_archive_finalize(pna_archive *a);
void pna_archive_abort(pna_archive *a);

/* pna create, non-solid or solid (cli/src/command/create.rs:575-635): compress the n entries on the GPU in one
 * batch (entry-parallel, like spawn_entry_results), then write them in index order.  algo PNA_ALGO_STORE works
 * without a GPU (ctx may be NULL). */
int  pna_create_archive(pna_gpu_ctx *ctx, int algo, int level, int solid, size_t n, const char *const *names,
                        const void *const *src, const size_t *src_len, pna_sink_fn sink, void *user);

/* Password hash -> cipher key on the C++ host: PBKDF2-HMAC-SHA-256 as hash::pbkdf2_with_salt drives it (lib/src/hash.rs:35-45;
 * lib/src/entry/write.rs:171-186).  `salt` is the raw salt (what the B64 SaltString decodes to).  When phsf != NULL it receives the PHC
 * string without the hash, "$pbkdf2-sha256$i=<rounds>,l=32$<salt B64, no padding>" -- the body of the PHSF chunk. */
int  pna_kdf_pbkdf2_sha256(const void *password, size_t password_len, const void *salt, size_t salt_len, uint32_t rounds,
                           uint8_t *key, size_t key_len, char *phsf, size_t phsf_cap);
/* Argon2 (RFC 9106, v0x13; kind 0 = d, 1 = i, 2 = id) on the C++ host: the reference's default password hash (hash::argon2_with_salt,
 * derive_password_hash, lib/src/hash.rs:6-33,47-70).  The read side (pna_gpu_extract_archive_host) uses it to open archives whose PHSF
 * is "$argon2id$v=19$m=..,t=..,p=..$salt". */
int  pna_kdf_argon2(int kind, const void *password, size_t password_len, const void *salt, size_t salt_len,
                    uint32_t t_cost, uint32_t m_cost_kib, uint32_t lanes, uint8_t *key, size_t key_len);
/* pna create --aes [ctr|cbc] --pbkdf2 (non-solid, zstd / deflate): one key derivation per archive (random 16-byte salt; rounds 0 =
 * the pbkdf2 crate's default 600 000), a fresh random IV per entry, cipher stage on the device (pna_gpu_create_archive_enc_host). */
int  pna_create_archive_encrypted(pna_gpu_ctx *ctx, int algo, int level, size_t n, const char *const *names,
                                  const void *const *src, const size_t *src_len, const void *password, size_t password_len,
                                  int cipher_mode, uint32_t rounds, pna_sink_fn sink, void *user);

/* `pna create --split`: re-frame ONE archive image into parts of at most max_part_bytes (SplitParts, lib/src/archive/split_parts.rs:
 * signature + AHED(archive number) ... [ANXT] AEND per part; chunks that fit keep their bytes, FDAT / SDAT chunks are cut at the budget
 * boundary and only the fragments get new CRCs).  sink(user, part_index, buf, len) receives the bytes of part `part_index` in order.
 * PNA_E_INVAL: max_part_bytes below MIN_SPLIT_PART_BYTES (64), a non-stream chunk larger than a part, malformed input. */
typedef int (*pna_part_sink_fn)(void *user, uint32_t part_index, const void *buf, size_t len);
int  pna_split_archive(const void *archive, size_t len, size_t max_part_bytes, pna_part_sink_fn sink, void *user, uint32_t *n_parts);
/* The reading side (Archive::read_next_archive): the parts in order -> one archive image for pna_gpu_extract_archive_host. */
int  pna_join_parts(const void *const *parts, const size_t *part_len, size_t n, pna_sink_fn sink, void *user);

/* ---- `pna append` / `pna update` (cli/src/command/append.rs:504-560, update.rs:620-665): the same entry producer, a different sink.
 * Archive::seek_to_end (lib/src/archive/read.rs:412-424): after the header, chunks are skipped (skip_chunk: lengths only, no CRC) up to
 * AEND; *aend_off is the offset of the AEND chunk -- where the appended entries go, overwriting it --, *has_next is 1 when an ANXT chunk
 * was passed (a multipart archive continues in the next part: append there, append.rs:533-560).  PNA_E_INVAL: not an archive, or
 * truncated before / inside the AEND chunk (UnexpectedEof in the reference, read.rs:588-604). */
int  pna_archive_seek_to_end(const void *archive, size_t len, uint64_t *aend_off, int *has_next);
/* The top-level records of an archive image in order (Archive::raw_entries, lib/src/archive/read.rs:46-66): for each normal entry
 * (FHED .. FEND) and each solid entry (SHED .. SEND) cb(user, index, name, kind, offset, length) -- `name` is the FHED name as
 * stored (empty for a solid entry, kind -1), [offset, offset + length) are its bytes in the image, which `pna update` copies unchanged for
 * entries it keeps (update.rs:620-665) while the changed ones go through pna_gpu_create_archive_part_host. */
typedef int (*pna_raw_entry_fn)(void *user, size_t index, const char *name, size_t name_len, int kind, uint64_t offset, uint64_t length);
int  pna_archive_list_entries(const void *archive, size_t len, pna_raw_entry_fn cb, void *user);

#ifdef __cplusplus
}
#endif
#endif
