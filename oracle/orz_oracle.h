/*
 * orz_oracle.h -- CPU restatement of the richox/orz v1.6.1 codec (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle of the repo: a plain-C restatement of the reference's
 * encoder and decoder, written by reading /root/reference/src/{lib,lz,matcher,symrank,
 * huffman,coder,mem,ioutil}.rs.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (orz_amd/) never links or calls it.
 *
 * Parity pinning: the reference ships no golden vectors and cannot be built here (no Rust
 * toolchain, nightly-only crate).  The oracle is pinned by (a) the reference's only unit test
 * (src/coder.rs:224-265) restated in tests/, (b) the hand-derived known-answer vectors of
 * SURVEY.md A.8, (c) encoder->decoder round trips, (d) agreement, stream for stream, with a second
 * restatement written independently in Python from the same source (tests/pyref, tests/test_pyref.py).
 * Byte-level parity with the Rust encoder itself is therefore "parity unpinned" beyond those
 * (see DESIGN.md).
 */
#ifndef ORZ_ORACLE_H
#define ORZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/lib.rs:31-34,54-55 */
#define ORC_LZ_BLOCK_SIZE ((1u << 25) - 1)
#define ORC_LZ_CHUNK_SIZE (1u << 20)
#define ORC_MATCH_MAX_LEN 240
#define ORC_MATCH_MIN_LEN 4
#define ORC_SENTINEL_LEN (ORC_MATCH_MAX_LEN * 2)
#define ORC_PREMATCH_LEN (ORC_LZ_BLOCK_SIZE / 2)
/* src/lz.rs:24-29 */
#define ORC_BUCKET_ITEMS 4094
#define ORC_NUM_SYMBOLS 389
#define ORC_WORD_SYMBOL 388
/* src/matcher.rs:18 */
#define ORC_BUCKET_HASH 4627

/* mirrors #[repr(C)] LZCfg, src/lz.rs:32-37 */
typedef struct {
    size_t match_depth, lazy_match_depth1, lazy_match_depth2;
} orc_lzcfg;

/* One parsed item, as the reference's MatchItem (src/lz.rs:100-116) plus its position.
 * Optional trace output of the encoder, used by tests to compare parses. */
typedef struct {
    uint32_t pos;         /* window offset of the item start (spos) */
    uint16_t symbol;      /* raw (pre-rank) symbol: literal, 256+roid*6+lenid, or 388 */
    uint16_t rank;        /* post-symrank symbol */
    uint16_t ctx;         /* symrank_context (9 bit) */
    uint16_t reduced_offset;
    uint8_t unlikely;     /* symrank_unlikely */
    uint8_t match_len;    /* 0 for literal/word */
    uint8_t enc_len;      /* encoded_match_len */
    uint8_t after_literal;
} orc_item;

typedef struct orc_trace {
    orc_item* items; /* caller-provided buffer */
    size_t cap;      /* capacity in items */
    size_t n;        /* filled by the encoder (stops recording when full) */
} orc_trace;

typedef struct orc_lz_encoder orc_lz_encoder;
typedef struct orc_lz_decoder orc_lz_decoder;

/* LZEncoder, src/lz.rs:69-346 */
orc_lz_encoder* orc_lz_encoder_new(void);
void orc_lz_encoder_free(orc_lz_encoder*);
void orc_lz_encoder_set_trace(orc_lz_encoder*, orc_trace* trace); /* NULL disables */
/* sbuf must point ORC_SENTINEL_LEN bytes inside an allocation (src/lib.rs:67-69). */
void orc_lz_encoder_encode(orc_lz_encoder*, const orc_lzcfg*, const uint8_t* sbuf, size_t sbuf_len,
                           uint8_t* tbuf, size_t spos, size_t* spos_out, size_t* tlen_out);
void orc_lz_encoder_forward(orc_lz_encoder*, size_t forward_len);

/* LZDecoder, src/lz.rs:348-479.  Returns 0 or -1 (InvalidData). */
orc_lz_decoder* orc_lz_decoder_new(void);
void orc_lz_decoder_free(orc_lz_decoder*);
int orc_lz_decoder_decode(orc_lz_decoder*, const uint8_t* tbuf, size_t tlen, uint8_t* sbuf,
                          size_t spos, size_t* spos_end_out);
void orc_lz_decoder_forward(orc_lz_decoder*, size_t forward_len);

/* Stream level on memory buffers, src/lib.rs:58-129.
 * *dst is malloc()ed by the callee; free with orc_free().  Return 0, or -1 on InvalidData /
 * truncated input (decode). */
int orc_encode_mem(const uint8_t* src, size_t src_len, const orc_lzcfg* cfg, uint8_t** dst,
                   size_t* dst_len, orc_trace* trace);
int orc_decode_mem(const uint8_t* src, size_t src_len, uint8_t** dst, size_t* dst_len,
                   size_t* consumed);
void orc_free(void* p);

/* Plan-driven encoder: encode a caller-supplied parse with the reference's state machine and emit
 * half; fails (-1, *err filled) on any item the format cannot express.  Checker for the GPU fast mode. */
enum { ORC_PLAN_WORD = 0, ORC_PLAN_LITERAL = 1, ORC_PLAN_MATCH = 2 };
enum { ORC_PLAN_ENOMEM = 1, ORC_PLAN_ESHORT, ORC_PLAN_EPOS, ORC_PLAN_ELEN, ORC_PLAN_EEND, ORC_PLAN_ESRC,
       ORC_PLAN_EBYTES, ORC_PLAN_ELENMIN, ORC_PLAN_EWORD, ORC_PLAN_ETYPE };
typedef struct {
    uint64_t pos;  /* item start: window offset (object level) or stream offset (orc_encode_plan_mem) */
    uint64_t src;  /* match source, same coordinate system; ignored for word / literal */
    uint8_t type;  /* ORC_PLAN_* */
    uint8_t len;   /* match length (4..240); ignored otherwise */
} orc_plan_item;
typedef struct { size_t pos; int code; } orc_plan_error;
int orc_lz_encoder_encode_plan(orc_lz_encoder*, const uint8_t* sbuf, size_t sbuf_len, uint8_t* tbuf, size_t spos,
                               const orc_plan_item* plan, size_t nplan, size_t* nused_out, size_t* spos_out,
                               size_t* tlen_out, orc_plan_error* err);
int orc_encode_plan_mem(const uint8_t* src, size_t src_len, const orc_plan_item* plan, size_t nplan,
                        uint8_t** dst, size_t* dst_len, orc_trace* trace, orc_plan_error* err);

/* Building blocks exposed for per-stage parity tests. */
/* HuffmanTable::new_from_sym_weights, src/huffman.rs:27-111; returns max code length */
int orc_huffman_lengths(const uint32_t* weights, size_t n, int max_code_len, uint8_t* lens_out);
/* HuffmanEncoding::from_huffman_table, src/huffman.rs:118-141 */
void orc_huffman_codes(const uint8_t* lens, size_t n, uint16_t* codes_out);
/* SymRankCoder, src/symrank.rs (state = 389 values + 389 indices + cnt + sum) */
typedef struct {
    uint16_t value[ORC_NUM_SYMBOLS];
    uint16_t index[ORC_NUM_SYMBOLS];
    uint32_t cnt, sum;
} orc_symrank;
void orc_symrank_new(orc_symrank*);
void orc_symrank_init(orc_symrank*, const uint16_t* values);
uint16_t orc_symrank_encode(orc_symrank*, uint16_t v, uint16_t vunlikely);
uint16_t orc_symrank_decode(orc_symrank*, uint16_t i, uint16_t vunlikely);
/* hash_dword % 4627, src/matcher.rs:256-263,117 */
uint32_t orc_hash_entry(const uint8_t* p);
/* the reference's only unit test, src/coder.rs:224-265: varint(n) + table + symbols; returns
 * encoded length, or -1 if the decode side does not reproduce the input */
long orc_coder_selftest(const uint8_t* input, size_t n, uint8_t* encoded, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
