/*
 * orz_hip.h -- C ABI of the MI355X-native orz encoder (liborz_hip.so).
 *
 * This is the drop-in boundary of the repo: plain C, pointers and sizes only.  The reference
 * (richox/orz v1.6.1, Rust) exports no C symbols (its `// pub mod ffi;` at src/lib.rs:10 is dead),
 * so each entry point below replaces the Rust call surface that `orz::encode` / `orz::decode`
 * use; the file:line of the interface it stands in for is cited per function.  INTEGRATION.md
 * shows the Rust `extern "C"` binding a maintainer would add.
 *
 * Error convention: 0 = ok, negative = failure (ORZ_E*); nothing throws across the boundary.
 * orz_last_error() returns a thread-local description of the last failure.
 */
#ifndef ORZ_HIP_H
#define ORZ_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORZ_OK 0
#define ORZ_EINVAL (-22)  /* bad argument / InvalidData (io::ErrorKind::InvalidData, src/lz.rs:413-415) */
#define ORZ_ENOMEM (-12)  /* output buffer too small / allocation failure */
#define ORZ_EIO (-5)      /* read/write callback failed (io::Error from Read/Write, src/lib.rs:58-63) */
#define ORZ_ENODEV (-19)  /* no usable HIP device, or a HIP runtime error */

/* src/lib.rs:31-34,54-55 -- window geometry the object-level API inherits */
#define ORZ_LZ_BLOCK_SIZE ((1u << 25) - 1)
#define ORZ_SBVEC_SENTINEL_LEN 480u
#define ORZ_SBVEC_PREMATCH_LEN (ORZ_LZ_BLOCK_SIZE / 2)

/* mirrors #[repr(C)] LZCfg, src/lz.rs:32-37 (three usize) */
typedef struct {
    size_t match_depth, lazy_match_depth1, lazy_match_depth2;
} orz_lzcfg;

/* level -> LZCfg, src/main.rs:97-102.  Returns ORZ_EINVAL for levels other than 0,1,2. */
int orz_lzcfg_from_level(int level, orz_lzcfg* out);

/* ---- object level: LZEncoder (src/lz.rs:69-346) --------------------------------------------- */
typedef struct orz_lz_encoder orz_lz_encoder;

/* LZEncoder::new, src/lz.rs:75-80.  `device` = HIP device ordinal.  NULL on failure. */
orz_lz_encoder* orz_lz_encoder_new(int device);
void orz_lz_encoder_free(orz_lz_encoder*);

/* LZEncoder::encode(&mut self, &LZCfg, sbuf, tbuf, spos) -> (spos, tpos), src/lz.rs:89-95,346.
 * Preconditions inherited from src/lib.rs:67-69: sbuf points ORZ_SBVEC_SENTINEL_LEN bytes inside
 * an allocation and sbuf[-480 .. sbuf_len+480) is readable (bytes past sbuf_len influence tail
 * decisions exactly as in the reference); spos >= ORZ_SBVEC_PREMATCH_LEN on the first call of a
 * block.  Produces ONE chunk (at most 2^20 items) per call like the reference.  The device parses
 * the whole block on the first call of a block and hands the remaining chunks out on the following
 * calls, which must continue at the returned *spos_out with the same sbuf contents. */
int orz_lz_encoder_encode(orz_lz_encoder*, const orz_lzcfg*, const uint8_t* sbuf, size_t sbuf_len, uint8_t* tbuf,
                          size_t tbuf_cap, size_t spos, size_t* spos_out, size_t* tlen_out);
/* LZEncoder::forward, src/lz.rs:82-87 (forward_len must be 2^24 = LZ_BLOCK_SIZE - PREMATCH, the
 * only value the reference ever passes, src/lib.rs:84). */
int orz_lz_encoder_forward(orz_lz_encoder*, size_t forward_len);

/* ---- stream level: orz::encode (src/lib.rs:58-92) ------------------------------------------- */
typedef ssize_t (*orz_read_fn)(void* ctx, uint8_t* buf, size_t cap); /* 0 = EOF, <0 = error */
typedef int (*orz_write_fn)(void* ctx, const uint8_t* buf, size_t len);
/* ProgressLogger, src/progress.rs:9-13 */
typedef void (*orz_progress_fn)(void* ctx, int is_finish, size_t in_bytes, size_t out_bytes);

int orz_encode(orz_read_fn, void* rctx, orz_write_fn, void* wctx, const orz_lzcfg*, orz_progress_fn, void* pctx,
               int device);

/* ---- decode side (host code: a stream decodes as one serial chain) ------------- */
typedef struct orz_lz_decoder orz_lz_decoder;
/* LZDecoder::new / decode / forward, src/lz.rs:352-478.  decode() writes the chunk's bytes at
 * sbuf[spos..] (sbuf laid out like the encoder's window, 480-byte pads, >= 2*LZ_BLOCK_SIZE long as in
 * src/lib.rs:102) and returns the new spos in *spos_end_out; ORZ_EINVAL = InvalidData. */
orz_lz_decoder* orz_lz_decoder_new(void);
void orz_lz_decoder_free(orz_lz_decoder*);
int orz_lz_decoder_decode(orz_lz_decoder*, const uint8_t* tbuf, size_t tlen, uint8_t* sbuf, size_t spos,
                          size_t* spos_end_out);
int orz_lz_decoder_forward(orz_lz_decoder*, size_t forward_len);
/* orz::decode, src/lib.rs:94-129 */
int orz_decode(orz_read_fn, void* rctx, orz_write_fn, void* wctx, orz_progress_fn, void* pctx);
/* whole-buffer convenience: decodes the first stream found at src; *consumed = bytes of src it used */
int orz_decode_mem(const uint8_t* src, size_t n, uint8_t** dst, size_t* dst_len, size_t* consumed);

/* ---- reusable stream encoder on caller memory (what bench.py and the Python mirror drive) ---- */
typedef struct {
    uint64_t blocks, sweeps, seg_evals, items, chunks, in_bytes, out_bytes;
    double t_prep_s, t_parse_s, t_post_s; /* host clock around device syncs */
    double parse_kernel_ms;               /* HIP-event time of the parse kernel launches (sum) */
    uint64_t parse_launches;
    double total_ms;                      /* HIP-event time of the whole encode on the stream */
    uint64_t host_syncs;                  /* times the host waited for a stream during the call (every hipStreamSynchronize of the
                                             encoder: reads of counts and sizes, copies of finished output, the closing wait) */
} orz_encode_stats;

typedef struct orz_stream orz_stream;
orz_stream* orz_stream_new(int device, const orz_lzcfg* cfg);
void orz_stream_free(orz_stream*);
/* Parse mode.  ORZ_MODE_EXACT reproduces the reference encoder's parse (src/lz.rs:131-235) item for item, so
 * the stream is byte-identical to `orz encode`; ORZ_MODE_FAST is the GPU-native parse (orz_amd/csrc/orz_fast.h):
 * same bitstream format, decodes bit-exactly with the reference decoder, size within +-0.5 % of the
 * reference's at the same level (BASELINE.json north_star), an order of magnitude faster.  New encoders start
 * in fast mode unless the environment says ORZ_MODE=exact.  tile_bytes (multiple of 4096) / rounds: 0 = keep. */
#define ORZ_MODE_EXACT 0
#define ORZ_MODE_FAST 1
int orz_stream_set_mode(orz_stream*, int mode, unsigned tile_bytes, unsigned rounds);
typedef struct {
    int mode;
    unsigned segment_bytes, window_segments;               /* exact mode: speculative segment / sweep window */
    unsigned fast_tile_bytes, fast_rounds, fast_row_entries; /* fast mode: Gauss-Seidel tile, rounds, candidates tabulated per position */
    unsigned unit_bytes;  /* bytes of a 16 MiB block encoded per pipeline unit (the whole block unless ORZ_FAST_UNIT says otherwise; a unit closes its last chunk) */
} orz_stream_config;
/* what the encoder actually runs with (bench.py reports these instead of literals) */
int orz_stream_get_config(orz_stream*, orz_stream_config* out);
/* HIP-event time and launch count of four kernels over the last orz_stream_encode call made with stats != NULL:
 * [0] the parse's per-position kernel (exact: ParseWave, fast: FastEval), [1] symbol ranking, [2] candidate table
 * build (fast mode), [3] path maps (fast mode).  What bench.py's roofline leg is computed from. */
int orz_stream_get_kernel_times(orz_stream*, double* ms4, uint64_t* launches4);
/* Profile mode: also bracket the kernels inside the fast parse's round loop ([0] and [3] above).  That loop normally
 * runs as one hipGraph replay per block (its launch sequence is the same for every full block); brackets need
 * individual launches, so a profiled encode is slower than a normal one -- use it for the roofline leg only. */
int orz_stream_set_profile(orz_stream*, int on);
/* Profile mode, all of it: EVERY kernel (and library call: sorts, scans, fills) of the last profiled orz_stream_encode made with
 * stats != NULL, by name, HIP-event time summed over its launches, largest first.  Returns the number of rows the encode
 * produced (copy at most `cap` of them into `rows`; rows == NULL just counts).  bench.py's per-kernel table and DESIGN.md 6
 * are generated from this (tools/kernel_table.py). */
typedef struct {
    char name[64];
    double ms;
    uint64_t launches;
} orz_kernel_row;
long orz_stream_get_kernel_table(orz_stream*, orz_kernel_row* rows, size_t cap);
/* tuning: bytes per speculative segment and segments per sweep window (0 = keep) */
int orz_stream_set_tuning(orz_stream*, unsigned seg_bytes, unsigned window_segs);
/* Encode `n` bytes at `src` (host memory, or device memory when src_on_device != 0) into a
 * malloc()ed orz stream (*dst, free with orz_free).  Same bytes as `orz encode` would write. */
int orz_stream_encode(orz_stream*, const void* src, size_t n, int src_on_device, uint8_t** dst, size_t* dst_len,
                      orz_encode_stats* stats);
void orz_free(void* p);
/* The same, with the finished stream left in DEVICE memory the caller owns (round 6): `d_dst` = `d_cap` bytes on the stream's
 * device; *dst_len bytes are written -- { LEB128(t) chunk[t] }* and the EOF byte, framed on the device (src/lib.rs:79-80,89).
 * The analogue of the reference handing LZEncoder::encode a caller-owned `tbuf` (src/lz.rs:89-95), for a whole stream: what the
 * multi-GPU gather sends from where it lies.  d_cap >= orz_stream_bound(n) always suffices; a buffer that turns out too small
 * fails the encode (ORZ_ENOMEM) without a byte of the overflowing block written.  One host wait per 16 MiB block and one per stream. */
size_t orz_stream_bound(size_t n);
int orz_stream_encode_to_device(orz_stream*, const void* src, size_t n, int src_on_device, uint8_t* d_dst, size_t d_cap,
                                size_t* dst_len, orz_encode_stats* stats);

/* Per-item trace of the last orz_stream_encode call (diagnostics / stage-level parity tests):
 * what the parse decided for each item, in stream order.  Mirrors the reference's MatchItem
 * (src/lz.rs:100-116) after the symrank pass. */
typedef struct {
    uint32_t block;         /* 0-based block index */
    uint32_t pos;           /* window offset of the item start (spos) */
    uint16_t symbol;        /* raw symbol: literal, 256 + roid*6 + lenid, or 388 (WORD) */
    uint16_t rank;          /* symbol after symrank */
    uint16_t ctx;           /* symrank_context (9 bit) */
    uint16_t robits;        /* robits | robitlen << 12 */
    uint8_t unlikely;       /* symrank_unlikely */
    uint8_t enc_len;        /* encoded_match_len */
    uint8_t after_literal;  /* bit 0: after_literal, bit 1: item is a match */
    uint8_t match_len;      /* match length (0 for literal / WORD) */
    uint32_t src;           /* window offset of the match source (0 for literal / WORD) */
} orz_item;
int orz_stream_set_item_trace(orz_stream*, int on);
/* copies up to cap items to out, returns the total number traced (or a negative error) */
long orz_stream_get_item_trace(orz_stream*, orz_item* out, size_t cap);

/* ---- many members per GPU (independent chunks, each a complete orz stream) ------------
 * A stream does not shard (its model state is one adaptive chain), so throughput beyond one stream
 * comes from encoding independent members concurrently: `jobs` stream encoders on one device, each fed
 * members of `member_bytes` input bytes by its own host thread.  The output is the members' streams
 * concatenated in input order; every member ends with its own EOF chunk, so the reference decoder reads
 * each piece (orz_decode_members_mem / `orz decode --members` loop over them). */
typedef struct orz_members orz_members;
orz_members* orz_members_new(int device, const orz_lzcfg* cfg, int jobs);
/* the same across several GPUs of the node: `jobs_per_device` stream encoders on each of the listed
 * devices, one host thread each; members go to whichever worker is free, the finished streams are collected in
 * host memory in member order (the only "gather" the job needs: the output lives on the host).  Input must be host
 * memory unless all workers sit on one device. */
orz_members* orz_members_new_multi(const int* devices, int n_devices, const orz_lzcfg* cfg, int jobs_per_device);
void orz_members_free(orz_members*);
int orz_members_encode(orz_members*, const void* src, size_t n, int src_on_device, size_t member_bytes, uint8_t** dst,
                       size_t* dst_len, size_t* n_members_out);
/* The same with the members' streams left in DEVICE memory the caller owns (round 6; all workers on ONE device): they are
 * written into `d_dst` (`d_cap` bytes on that device) in whatever order they finish; member k's stream is the
 * lens[k] bytes at d_dst + offs[k] (`offs`, `lens`: host arrays of at least (n + member_bytes - 1) / member_bytes entries,
 * 1 for n = 0).  d_cap >= orz_stream_bound(member_bytes) * members always suffices; ORZ_ENOMEM when the buffer fills. */
int orz_members_encode_to_device(orz_members*, const void* src, size_t n, int src_on_device, size_t member_bytes, uint8_t* d_dst,
                                 size_t d_cap, size_t* offs, size_t* lens, size_t* n_members_out);
/* decodes every stream of a concatenation (a plain single stream is the 1-member case) */
int orz_decode_members_mem(const uint8_t* src, size_t n, uint8_t** dst, size_t* dst_len, size_t* n_members_out);

/* ---- members decoded on the device (decoder on GPU, one member per wavefront) --------
 * Replaces orz::decode (/root/reference/src/lib.rs:94-129) + LZDecoder::decode (src/lz.rs:366-478) for a
 * concatenation of members: decoding one stream is a serial chain, so each member is decoded by one lane of
 * its own wavefront and the parallelism is the number of members (up to 2048 in flight).  Members of several
 * blocks decode too (round 4): a member decodes straight into its place in the output, ring nodes hold offsets into
 * the member, and the reference's window slide (src/lib.rs:119-124, src/matcher.rs:82-87) is a counter that retires
 * the nodes that left the window.  Members of 4 GiB or more fail with ORZ_EINVAL and a message naming the host
 * decoder.  Same bytes out as orz_decode_members_mem; ORZ_EINVAL for what the reference reports as InvalidData. */
typedef struct {
    uint64_t members, in_bytes, out_bytes;
    uint64_t launches;     /* kernel launches (members / 2048, rounded up) */
    double kernel_ms;      /* HIP-event time of the decode kernel launches (sum) */
    double total_s;        /* wall time incl. framing scan, uploads and the download of the result */
} orz_decode_stats;
int orz_decode_members_device(int device, const uint8_t* src, size_t n, uint8_t** dst, size_t* dst_len,
                              size_t* n_members_out, orz_decode_stats* stats);

/* The Huffman tables of `nchunks` chunks on the device, in the layout the encoder keeps them: a chunk is
 * orz_huffman_stride() = 389 + 389 + 240 entries (symbol ranks after a match / after a literal, long match lengths:
 * /root/reference/src/lz.rs:272-273,298-305), each of the three built as HuffmanTable::new_from_sym_weights(weights, 15)
 * (src/huffman.rs:27-111) followed by HuffmanEncoding::from_huffman_table (src/huffman.rs:118-141).  `lens` and `codes`
 * receive nchunks * stride entries.  Weights must be below 2^23 (a chunk holds at most 2^20 items, src/lib.rs:32);
 * ORZ_EINVAL otherwise.  `elapsed_us`, when not NULL, receives the HIP-event time of the one kernel launch. */
size_t orz_huffman_stride(void);
int orz_huffman_tables(int device, const uint32_t* weights, size_t nchunks, uint8_t* lens, uint16_t* codes, double* elapsed_us);

int orz_device_count(void);
const char* orz_last_error(void);
const char* orz_version(void);

#ifdef __cplusplus
}
#endif
#endif
