/*
 * oracle/zstd_model.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C model ("the spec") of the MI355X zstd-format encoder that replaces, behind
 * Compression::ZStandard, the third-party encoder the reference calls at
 *   lib/src/entry/write.rs:260-262  (zstd::stream::write::Encoder::new(writer, level))
 *   lib/src/compress.rs:32-41,66-75 (CompressionWriter::write / try_into_inner = finish)
 * The reference pins only DECOMPRESSED bytes at this boundary, so the model is free to
 * choose its own parse; what it fixes is a deterministic algorithm that the HIP kernels must reproduce
 * BIT-EXACTLY (tests/test_gpu_parity.py), and whose output must decode with the RFC 8878 decoder in
 * zstd_dec.c and with the system libzstd to the input (tests/test_oracle_zstd_model.py).
 */
#ifndef PNA_ORACLE_ZSTD_MODEL_H
#define PNA_ORACLE_ZSTD_MODEL_H
#include <stdint.h>
#include <stddef.h>

#define PNA_SEG_SIZE   (1u << 20)   /* one zstd frame per 1 MiB segment of an entry            */
#define PNA_BLK_SIZE   (1u << 17)   /* zstd Block_Maximum_Size: the default block size          */
#define PNA_BLK_LOG_MIN 13          /* smallest block a parameter set may ask for (blk_log)     */
#define PNA_BLK_MIN    (1u << PNA_BLK_LOG_MIN)

#define PNA_F_HUF      1u           /* Huffman-compressed literals allowed                      */
#define PNA_F_FSE      2u           /* FSE_Compressed sequence tables allowed (else predefined) */
#define PNA_F_LAZY     4u           /* one-step lazy deferral inside a 64-position group        */
#define PNA_F_REP      8u           /* repeat-offset codes (block-local history)                */
#define PNA_F_LAZY3    0x100u       /* with PNA_F_LAZY2: also defer to a match at q + 3 that is longer by three or more */
#define PNA_F_SINGLE_FRAME 0x400u   /* zstd model: an entry is ONE frame (header once, last-block bit once) instead of a frame per 1 MiB segment */
#define PNA_F_STORED   0x200u       /* deflate model only: level 0 (Compression::none()): every block a stored block, header 78 01 */
#define PNA_F_LAZY2    0x80u        /* with PNA_F_LAZY: also defer to a match at q + 2 that is longer by two or more (strong set) */

typedef struct {
    uint32_t hash_log;    /* LDS hash table: <= 31: 1 << hash_log entries (u32 each); larger: the entry count itself */
    uint32_t min_match;   /* bytes hashed and minimum match length (4..6)                       */
    uint32_t tile;        /* positions matched per synchronous step                             */
    uint32_t max_off;     /* largest usable offset (zstd: the whole 1 MiB segment; deflate: 32 KiB) */
    uint32_t cap1;        /* per-position match length cap before cooperative extension         */
    uint32_t lookahead;   /* bytes beyond the tile end that an extension may read               */
    uint32_t flags;       /* PNA_F_*                                                            */
    uint32_t max_len;     /* longest match (0 = only limited by the look-ahead); 258 for deflate          */
    uint32_t region;      /* positions parsed as one unit (128 = one GPU wave); 0 = one exact greedy parse per tile */
    uint32_t ins_mod;     /* only positions q with q % ins_mod == 0 enter the hash table (0 / 1 = all)                      */
    uint32_t back_cap;    /* bytes a match may be known to continue backwards (0 = no backward adoption)                    */
    uint32_t rounds;      /* backward adoption rounds: nibbles = lane shifts, lowest first (0x21 = shift 1, then shift 2)   */
    uint32_t near_off;    /* offsets above it are "far" (outside the GPU's LDS window): per-position cap cap_far; 0 = none  */
    uint32_t cap_far;     /* per-position match length cap of far candidates                                                */
    uint32_t blk_log;     /* block size = 1 << blk_log for 13..16 (latency mode of the device: short per-block chains); anything else: 128 KiB */
    uint32_t len_word_max; /* adopted lengths are clamped to it (0 = no clamp): the device's 3-byte words keep lengths up to 36 (plus 19 bits of offset: max_off 524 287) */
    uint32_t tab3;        /* 1: the PACKED table of the device's default / high zstd sets -- hash_log (> 31, a multiple of 3) slots in words of THREE: word =
                           * floor(h * (slots / 3) / 2^32), field = ((h & 0xFFFF) * 3) >> 16, slot = 3 * word + field (on the device a 64-bit LDS word of three
                           * 21-bit entries: 19 bits of even position, 2 of tag).  A tile's inserts into one word are ONE 64-bit maximum of "the word as it was,
                           * my field replaced": of a tile's contenders for a word only the one with the highest (field, position) is stored, the others are
                           * lost (4 - 8 % of the inserts; the estimator: -0.1 % of ratio for a half more slots in the same LDS).  Needs ins_mod = 2. */
    uint32_t mtile;       /* look-ups and inserts alternate per SUB-TILE of mtile positions inside a tile (0 = the whole tile): the positions of a sub-tile see the
                           * inserts of the sub-tiles before it.  Parse, lazy deferral and extension keep the tile. */
    uint32_t small_seg;   /* segments of at most this many bytes (0 = none) run the SMALL geometry of the device -- one wave per segment instead of a workgroup:
                           * table of small_slots 32-bit entries (not packed), sub-tiles of small_tile positions (one wave's 256).  With the large geometry a
                           * segment of one tile never finds a match (its positions do not see each other): 4 KiB text entries 1.77 -> 2.07 (zlib -6: 2.02). */
    uint32_t small_slots, small_tile;
    uint32_t mid_seg, mid_slots;   /* a second tier of the same geometry: segments above small_seg and of at most mid_seg bytes (0 = none) with a table of mid_slots entries
                                    * (the device gives both tiers the same table and sizes the window by the tier: 8 KiB text entries 1.97 -> 2.16, 16 KiB 2.16 -> 2.24; zlib -6: 2.12 / 2.22) */
    uint32_t cut_min;     /* F: a match the merge cuts from the front is kept iff at least this many bytes remain (0 = 3, the deflate minimum; the zstd sets: min_match --
                           * a sequence of 3 - 5 bytes costs more than its bytes as literals: + 0.12 % of ratio on the corpus) */
    uint32_t fixup;       /* F: 1 = where the merge drops the remainder of a straddling match (shorter than cut_min), ONE match found inside that remainder is emitted if it ends on a
                           * position the region's own walk stood on (pna_lz_block, F): + 0.07 % of ratio -- most of what parsing a tile's regions blind to each other loses */
    uint32_t far_slots, far_from;   /* far_slots != 0 (the device's default / light zstd sets, round 5): the match kernel verifies at most far_slots FAR candidates (offset >= far_from: outside
                           * its LDS window) per wave of 256 consecutive positions (tile start + 256 w ..) -- one compacted round of its 63 dense lanes instead of two.  The wave's far
                           * candidates (usable: position >= 8, offset <= max_off, and the entry's 2-bit tag equals the position's -- a foreign tag costs the device a slot like a
                           * true candidate) are numbered in the kernel's order, j-major: all positions with (q - wave start) % 4 == 0 in ascending order, then % 4 == 1, ...;
                           * those numbered far_slots and up are dropped (no candidate).  Needs tab3. */
} pna_zstd_params;
/* the parameters segment `seg_len` bytes long runs with: p itself, or *tmp = p with the small geometry */
const pna_zstd_params *pna_seg_params(const pna_zstd_params *p, uint32_t seg_len, pna_zstd_params *tmp);

typedef struct { uint32_t ll, ml, off; } pna_seq;   /* literal run, match length, offset (>=1) */

void   pna_zstd_default_params(pna_zstd_params *p);
size_t pna_zstd_bound(size_t n);
/* Compress one entry (any length) into concatenated frames.  Returns bytes written or 0 if cap too small. */
size_t pna_zstd_model_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, const pna_zstd_params *p);

/* stage-level entry points used by the parity tests */
/* LZ stage for one block of a segment: fills seqs/lits, returns nseq; *nlit_out = literal bytes. `table`
 * (1<<hash_log u32, zero at segment start) carries across the blocks of one segment. */
uint32_t pna_lz_block(const uint8_t *seg, uint32_t seg_len, uint32_t blk_start, uint32_t blk_len,
                      uint32_t *table, const pna_zstd_params *p,
                      pna_seq *seqs, uint8_t *lits, uint32_t *nlit_out);
/* entropy + framing for one segment (= one frame) from the LZ stage's per-block outputs; seqs of block b start at
 * index b*(blk_size/4), its literals at byte b*blk_size.  Returns the frame size. */
size_t pna_zstd_encode_segment(const uint8_t *seg, uint32_t seg_len, const pna_seq *seqs, const uint8_t *lits,
                               const uint32_t *blk_nseq, const uint32_t *blk_nlit, uint32_t flags, uint32_t blk_size, uint8_t *dst);
uint32_t pna_blk_size(const pna_zstd_params *p);
#endif
