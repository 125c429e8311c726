// pna_dev.h -- shared host/device declarations of the gfx950 zstd-format encoder.
//
// Normative algorithm: DESIGN.md §"Encoder specification".  The same rules are restated independently in plain C
// under oracle/ (test infrastructure); tests require the two to agree byte for byte.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace pna {

constexpr uint32_t SEG_SIZE   = 1u << 20;   // one zstd frame per segment of an entry
#ifndef PNA_BLK_LOG
#define PNA_BLK_LOG 17
#endif
constexpr uint32_t BLK_SIZE   = 1u << PNA_BLK_LOG;   // block size of both codecs (<= zstd's Block_Maximum_Size of 128 KiB)
constexpr uint32_t BLK_PER_SEG = SEG_SIZE / BLK_SIZE;

// LZ stage (k_lz)
constexpr uint32_t LZ_THREADS = 1024;       // 16 waves, one workgroup per CU (LDS-bound)
constexpr uint32_t LZ_WAVES   = LZ_THREADS / 64;
constexpr uint32_t TILE       = 2048;       // positions per entry of the deflate chunk table (k_dblock packs a block in chunks of this many positions); k_lz's own tile is 1024 G
constexpr uint32_t GROUPS_PER_WAVE = TILE / 64 / LZ_WAVES;   // 2
#ifndef LZ_HASH_ENTRIES_VALUE
#define LZ_HASH_ENTRIES_VALUE 24512
#endif
constexpr uint32_t HASH_ENTRIES = LZ_HASH_ENTRIES_VALUE;   // LDS hash table of k_lz: as many u32 entries as fit next to the 64 KiB window (index = mulhi(hash, entries))
constexpr uint32_t MIN_MATCH  = 6;
constexpr uint32_t CUT_MIN    = MIN_MATCH;  // the merge keeps a match it cuts from the front iff at least this many bytes remain (round 5: 3 until then -- a sequence of 3 - 5 bytes costs more than its bytes as literals; zstd-3 + 0.12 % of ratio, deflate + 0.06 %)
#ifndef LZ_G_ZSTD_VALUE
#define LZ_G_ZSTD_VALUE 4
#endif
constexpr uint32_t LZ_G_DEFLATE = 4;        // the same for the deflate launches (look-back 32 KiB: any G fits the window)
constexpr uint32_t LZ_G_ZSTD  = LZ_G_ZSTD_VALUE;          // positions per lane and tile of the zstd launches of k_lz (tile = 1024 x this; deflate: 2)
constexpr uint32_t MAX_OFF_G2 = 59392;      // 64 KiB window - 2 tiles of 2 048 - look-ahead - slack
constexpr uint32_t NEAR_OFF   = LZ_G_ZSTD == 2 ? MAX_OFF_G2 : 65536 - 2 * 1024 * LZ_G_ZSTD - 1024 - 16 - 240;   // candidates at most this far back are verified in the LDS window: 64 KiB - 2 tiles - look-ahead - slack (G = 4: 56 064)
constexpr uint32_t MAX_OFF    = 1u << 20;   // zstd: any earlier position of the 1 MiB segment; beyond NEAR_OFF the candidate is read from HBM / L2
constexpr uint32_t BACK_CAP   = 3;          // bytes before a match that are known to agree with its candidate (backward adoption)
#ifndef LZ_CAP1_VALUE
#define LZ_CAP1_VALUE 32
#endif
constexpr uint32_t CAP1       = LZ_CAP1_VALUE;
constexpr uint32_t LOOKAHEAD  = 1024;
constexpr uint32_t WIN_BYTES  = 65536;      // circular look-back window in LDS
constexpr uint32_t SEQ_CAP    = PNA_BLK_LOG == 17 ? 22528 : ((BLK_SIZE / MIN_MATCH + BLK_SIZE / 256 + 256 + 255) & ~255u);      // sequences per block: (BLK_SIZE - 3 * 1024) / MIN_MATCH + one front-cut match (>= 3 bytes) per wave region, rounded up to 256

constexpr uint32_t F_HUF = 1, F_FSE = 2, F_LAZY = 4, F_REP = 8;
constexpr uint32_t FLAG_LAZY2 = 0x400u;   // launch flag of the LZ kernels (with F_LAZY): two-step lazy deferral -- a start also waits for a match at q + 2 that is longer by two or more
constexpr uint32_t FLAG_LAZY3 = 0x800u;   // (with FLAG_LAZY2) three-step deferral: ... and for a match at q + 3 that is longer by three or more
constexpr uint32_t FLAG_W16 = 0x4000u;   // ... the 16 KiB-window geometry (36 800 table slots)
constexpr uint32_t FLAG_LEN36 = 0x8000u;  // adopted lengths clamped to 36: what the 3-byte words of the split form keep (5 bits: 0 or length - 5) next to 19 bits of offset
constexpr uint32_t MAX_OFF_W3 = (1u << 19) - 1;   // ... so the look-back of the sets with an LDS table ends there (the estimator: 2.7761 -> 2.7759; 2^18: 2.7678)
constexpr uint32_t FLAG_TAB3 = 0x10000u;  // (with FLAG_W32 / FLAG_W16, even inserts) the packed table: three 21-bit entries per 64-bit LDS word, 49 062 / 55 206 slots (lz_common.h)
// SHORT segments (at most SMALL_SEG bytes: less than one tile of the match finder, whose positions do not see each other's inserts -- such a segment would never
// find a match) run the SMALL geometry: one wave per segment (k_lzms, k_lz_split.hip), a table of SMALL_SLOTS 32-bit entries, look-ups and inserts alternating per
// 256 positions; the parse is the common one.  oracle/zstd_model.h: small_seg, small_slots, small_tile.
constexpr uint32_t SMALL_SEG = 4096, SMALL_SLOTS = 2048;
// ... and a second tier of it: segments of 4 097 .. MID_SEG bytes -- a segment of two to four of the large geometry's tiles finds nothing in its first one (8 KiB text
// entries 1.93 -> 2.12, 16 KiB 2.14 -> 2.24; above 16 KiB the large table wins).  Same table size (4 096 slots were + 1 % of ratio for half the waves per CU); the tiers
// differ in the LDS their window takes, which a launch sizes by its longest segment of the tier (FLAG_T2_SHIFT: 8, 12 or 16 KiB).  oracle: mid_seg, mid_slots.
constexpr uint32_t MID_SEG = 16384, MID_SLOTS = SMALL_SLOTS;
constexpr uint32_t FLAG_T2_SHIFT = 22;          // two bits: the second tier's longest segment is at most 8 / 12 / 16 KiB (0, 1, 2)
constexpr uint32_t FLAG_HAS_SMALL = 0x20000u;   // launch flag of the LZ kernels: the launch may hold short segments (at most MID_SEG bytes) -- k_lzms takes them, the other match kernels skip them
constexpr uint32_t FLAG_SMALL_ONLY = 0x40000u;  // (k_lzp) parse the blocks of short segments only: the pass behind a one-kernel launch, which skipped them
constexpr uint32_t FLAG_TIER1 = 0x100000u, FLAG_TIER2 = 0x200000u;   // (with FLAG_HAS_SMALL) the launch holds segments of the first / second tier: which of k_lzms's two forms to launch
constexpr uint32_t FLAG_ALL_SMALL = 0x80000u;   // every segment of the launch is short (or empty): the large geometry's kernels are not launched at all
constexpr uint32_t FLAG_FAR1 = 0x1000000u;   // (with FLAG_TAB3 + FLAG_W32) at most 63 far candidates per wave of 256 positions are verified -- ONE compacted round of k_lzm --, the rest dropped (oracle: far_slots, far_from)
constexpr uint32_t FLAG_STRONG2 = 0x2000000u; // (with F_STRONG; the packed 16 KiB geometry or the table in global memory: zstd 4 .. 22) a fourth adoption round over eight positions, first, and up to 15 back bytes (oracle: rounds 0x2148, back_cap 15; usable candidates lie at position 16 or beyond)
constexpr uint32_t FLAG_W32 = 0x2000u;   // launch flag of the LZ kernels: the 32 KiB-window geometry (zstd only; lz_common.h LzGeo)
constexpr uint32_t F_FAR = 0x10, F_ADOPT = 0x20, F_INS2 = 0x40, F_STRONG = 0x80;   // look-back beyond the LDS window; backward adoption; only even positions enter the table; third adoption round (7 back bytes) + two-step lazy

// packed sequence: off(20) | ml(18) << 20 | ll(18) << 38
__host__ __device__ inline uint64_t seq_pack(uint32_t ll, uint32_t ml, uint32_t off) {
    return (uint64_t)off | ((uint64_t)ml << 20) | ((uint64_t)ll << 38);
}
__host__ __device__ inline uint32_t seq_off(uint64_t s) { return (uint32_t)(s & 0xFFFFF); }
__host__ __device__ inline uint32_t seq_ml(uint64_t s)  { return (uint32_t)((s >> 20) & 0x3FFFF); }
__host__ __device__ inline uint32_t seq_ll(uint64_t s)  { return (uint32_t)((s >> 38) & 0x3FFFF); }

struct SegDesc {
    uint64_t src_off;     // byte offset of the segment in the batch input buffer (multiple of 16)
    uint32_t len;         // 1 .. SEG_SIZE
    uint32_t blk_base;    // index of the segment's first block in the per-block arrays
    uint32_t entry;       // entry the segment belongs to
    uint32_t first;       // bit0: entry's first segment, bit1: entry's last segment
    uint32_t u0, u1;      // LZ stage: the part of the segment one workgroup parses, [u0, u1): whole blocks (u1: or the segment's end).  The segment's own
                          // descriptor has [0, len); in latency mode (small batches) the LZ kernels run over UNITS, copies of it with a piece each
    uint32_t blk_log;     // block size of the batch = 1 << blk_log: 17 (BLK_SIZE), or BLK_LOG_MIN..16 (latency mode; batches of small entries).  It is also the
                          // stride of the per-block arrays: literals, bitstreams 1 << blk_log bytes, sequences seq_cap_of(blk_log) slots, chunk table 1 << (blk_log - 11).
    uint32_t pad;
};
// the parse kernel of the split LZ form runs one wave per BLOCK (a block's parse depends on nothing outside it): all segments of the sub-batch, the
// block -> segment map, and the launch's blocks [blk0, blk0 + nb)
struct LzParseGrid { const SegDesc *segs_all; const uint32_t *blk_seg; uint32_t nb;
                     uint32_t *hist = nullptr; };    // hist (round 5; zstd, large batches): the segments' histogram counters (448 words each: k_entropy.hip HIST_WORDS) -- the parse kernel adds every block's sequence codes to words 256 .. 447 of its segment, so that k_stats reads the literals only
constexpr uint32_t BLK_LOG_MIN = 13;   // smallest block of the latency mode (bounds: pna_gpu_bound)
static_assert(sizeof(SegDesc) == 40, "SegDesc layout");
// sequences a block of 1 << blk_log bytes can hold: a match is >= MIN_MATCH bytes but for one front-cut match (>= 3 bytes) per 256-position parse region
__host__ __device__ inline uint32_t seq_cap_of(uint32_t blk_log) {
    return blk_log >= PNA_BLK_LOG ? SEQ_CAP : ((((1u << blk_log) / MIN_MATCH + ((1u << blk_log) >> 8) + 256 + 255)) & ~255u);
}
__host__ __device__ inline uint32_t seg_nblk(const SegDesc &sd) { return (sd.len + (1u << sd.blk_log) - 1) >> sd.blk_log; }

// per-segment entropy tables (k_stats output; 4 KiB-ish, read by k_lit / k_seq / k_pack)
constexpr uint32_t SEQ_MAX_LOG = 8;
struct SeqSym { uint32_t delta_nb; int16_t delta_find; uint16_t first_state; };
struct SeqTable { uint16_t state[1u << SEQ_MAX_LOG]; SeqSym sym[56]; };   // 512 + 448 bytes
struct SegTables {
    uint32_t huf_ok, tree_len, max_sym, maxbits;
    uint32_t mode[3], tlog[3], desc_len[3], seq_ok;
    uint32_t nseq_seg, pad[2], pad2;
    uint32_t huf_code[256];            // code | len << 16
    uint8_t  tree[160];
    uint8_t  desc[3][96];
    SeqTable tab[3];                   // LL, OF, ML
};

// per-block results
struct BlkInfo {
    uint32_t nseq, nlit;
    uint32_t lit_body;     // bytes of Huffman body (jump table + streams), 0 if not produced
    uint32_t lit_rle;      // 1 if all literals are the same byte
    uint32_t seq_bits;     // bytes of the sequence bitstream
    uint32_t out_size;     // final block size incl. 3-byte header (k_plan)
    uint32_t plan;         // bit0 compressed block, bit1 huffman literals, bit2 carries tree, bit3 carries seq tables, bit4 rle literals
    uint32_t pad;
    uint64_t out_off;      // offset of the block header in the batch output buffer
    uint32_t adler_a, adler_b;   // deflate: Adler-32 halves of this block's input started from adler = 1
};

// deflate: per-segment Huffman tables (k_dstats output)
struct DeflTables {
    uint32_t ll_code[288];   // bit-reversed code | len << 16 (literals 0..255, EOB 256, lengths 257..285)
    uint32_t d_code[32];
    uint32_t hdr_bits, pad[3];
    uint8_t  hdr[400];       // HLIT HDIST HCLEN + code-length code + coded lengths, LSB-first
};

// in-HBM framing (k_frame): one descriptor per entry
struct FrameDesc {
    uint64_t arc_off;        // offset of the entry's first byte (FHED chunk) in the archive buffer
    uint32_t payload_len;    // bytes of the FDAT payload, already in place at arc_off + prefix_len
    uint32_t prefix_off;     // offset of the entry's prefix (FHED | fSIZ | FDAT length + type) in the prefix blob
    uint32_t prefix_len;
    uint32_t pad;
};
struct CrcTabs {
    uint32_t T[4][256];      // slice-by-4 tables of the reflected CRC-32
    uint32_t Z[4][256];      // "append 16320 zero bytes" as four byte-indexed tables
    uint32_t sh[8];          // x^(8 * 64 * 2^j) mod P, j = 0..7
    uint32_t pw[4][64];      // k_frame_wave: x^(8 * 64 m * k) mod P for m = 1..4 pieces per lane, k = 0..63 lanes behind
};

// cipher stage (k_cipher.hip)
struct AesTabs { uint32_t Te[4][256]; };   // Te[0][x] = bytes (2S, S, S, 3S) of S = sbox[x], little-endian; Te[k] = Te[0] rotated left by 8k bits
struct AesKey  { uint32_t rk[60]; };       // AES-256 round keys, word 4r + c = column c of round r, little-endian
struct AesDecTabs { uint32_t Td[4][256]; uint32_t Sd[256]; };   // equivalent inverse cipher: Td[0][x] = bytes (14, 9, 13, 11) x Si[x]; Sd = inverse S-box
struct CipherUnit {
    uint64_t off;            // first byte of the unit in the buffer
    uint64_t pos;            // CTR: its byte position in the entry's cipher stream (keystream block = pos / 16)
    uint32_t len;            // bytes (CBC: plaintext bytes of the whole entry)
    uint32_t iv_idx;         // which 16-byte IV of the IV array belongs to it
};

struct GcmEntry {            // k_gcm_tag: one GCM segment (this library writes one per entry)
    uint64_t off;            // first ciphertext byte in the buffer; the 16-byte tag goes to off + len
    uint32_t len, pad;
    uint32_t h[4];           // hash subkey H = E(K, 0^128), big-endian words (word 0 = bytes 0..3)
    uint32_t ej0[4];         // E(K, nonce || 00000001), same form
};

// zstd decoder (k_zdec): one descriptor per frame
struct ZFrame {
    uint64_t src_off;        // frame start in the compressed buffer
    uint64_t dst_off;        // where its content goes
    uint64_t src_len;        // bytes of the frame          } 64 bits for zlib streams that are decoded by pieces (k_vinflate); zstd frames and the
    uint64_t dst_len;        // bytes of content it must produce } wave-per-stream inflate walk take what fits 32 bits (status 2 otherwise)
    uint32_t status;         // out: 0 ok, 1 corrupt, 2 unsupported, 3 size mismatch
    uint32_t out_len;        // in: flags (ZF_OPEN); out: bytes produced, saturated at 2^32 - 1 (diagnostics)
};
static_assert(sizeof(ZFrame) == 40, "ZFrame layout");

// ---- lane-parallel decoder (k_zparse / k_zhuf / k_zfse / k_zoff / k_zexec)
struct ZFrameX {             // per frame: where its blocks, table slots and sequence records live
    uint64_t seq_base;       // first record of the frame in the sequence scratch
    uint32_t blk_base, blk_cap;      // its region of the block array
    uint32_t slot_base, slot_cap;    // its table slots
    uint32_t seq_cap;                // records available (saturated at 2^31 - 1)
    uint32_t nblk;                   // out
    uint32_t pcap, pad;              // k_vinflate: records available to each piece (piece j's start at seq_base + j * pcap)
};
struct ZBlock {
    uint64_t body;           // offset of the block body in the compressed buffer
    uint64_t out_off;        // absolute output offset (k_zoff)
    uint64_t seq_pos;        // first sequence record (absolute index into the scratch)
    uint32_t size, type;     // body bytes; 0 raw, 1 RLE, 2 compressed
    uint32_t ltype, regen, streams, lit_off, lit_csize;    // literals: section payload (body-relative) behind the tree description
    uint32_t lit_pos;        // where the block's decoded literals go in the frame's literal scratch (frame-relative; bits 32.. in pad[2])
    uint32_t huf_slot, slot[3];      // table slots that apply (Huffman; LL, OF, ML)
    uint32_t nseq, seq_off, seq_len; // sequence bitstream (body-relative)
    uint32_t frame, out_len, status, uses_rep;
    uint32_t pad[7];         // [0] zstd: last-block bit; [1] inflate: the Adler-32 trailer read with the stream's last piece; [2] lit_pos >> 32;
                             // inflate pieces: [3], [4] the earliest stream position a match of the piece copies from (k_vinflate), [5] the piece starts an execution group (k_vfin)
};
static_assert(sizeof(ZBlock) == 128, "ZBlock layout");
struct ZTables {             // one slot
    uint16_t huf[2048];      // sym | nbits << 8
    uint32_t fse[3][512];    // sym | nbits << 8 | base << 16
    uint32_t hufbits, alog[3];
};

struct ZEntry {              // k_zscan input: one compressed entry
    uint64_t src_off, src_len;   // its payload (concatenated frames) in the compressed buffer
    uint64_t dst_off, raw_len;   // where the content goes and how long it is (fSIZ)
    uint32_t first_frame, n_frames;
    uint32_t open, pad;          // open != 0: raw_len is only a capacity -- the last frame's size is found by decoding (streams without fSIZ: solid)
};
// Decoder records (k_zfse / k_zdec / k_inflate / k_vinflate -> k_zexec): literal run (18 bits: a zstd block and an inflate piece hold at most 128 KiB) |
// match length << 18 (18 bits: zstd's longest is 131 074) | offset value << 36 (28 bits: offset + 3 -- 1 .. 3 are zstd's repeat codes --; libzstd's
// largest window without --long, level 22's 128 MiB, fits).
constexpr uint64_t ZREC_OF_MAX = 0xFFFFFFFull;
__host__ __device__ inline uint64_t zrec_pack(uint32_t ll, uint32_t ml, uint64_t ofv) { return (uint64_t)ll | ((uint64_t)ml << 18) | (ofv << 36); }
__host__ __device__ inline uint32_t zrec_ll(uint64_t s) { return (uint32_t)(s & 0x3FFFF); }
__host__ __device__ inline uint32_t zrec_ml(uint64_t s) { return (uint32_t)((s >> 18) & 0x3FFFF); }
__host__ __device__ inline uint32_t zrec_of(uint64_t s) { return (uint32_t)(s >> 36); }
constexpr uint32_t ZF_OPEN = 0x80000000u;   // ZFrame.out_len on input: dst_len is a capacity, the decoder writes the size it found back to dst_len

} // namespace pna
