/*
 * orz_oracle.c -- CPU restatement of richox/orz v1.6.1 (TEST INFRASTRUCTURE ONLY; see header).
 *
 * Every function cites the reference lines it follows.  Data structures and the order of side
 * effects are the reference's; the literal-copy hazards of SURVEY.md A.7 are kept on purpose
 * (reads past sbuf_len, stale hash heads, pos==0 == invalid, cumulative Huffman shrinking ...).
 */
#include "orz_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ mem.rs */

static inline uint32_t ld32(const uint8_t* p) { /* src/mem.rs:17-27 get::<u32> (little endian) */
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}

/* src/mem.rs:41-51: LCP in 16-byte steps, capped at max_len (=240) */
static inline size_t common_prefix(const uint8_t* buf, size_t p1, size_t p2, size_t max_len) {
    for (size_t l = 0; l < max_len; l += 16) {
        uint64_t a0, a1, b0, b1;
        memcpy(&a0, buf + p1 + l, 8);
        memcpy(&a1, buf + p1 + l + 8, 8);
        memcpy(&b0, buf + p2 + l, 8);
        memcpy(&b1, buf + p2 + l + 8, 8);
        if (a0 != b0) return l + ((size_t)__builtin_ctzll(a0 ^ b0) >> 3);
        if (a1 != b1) return l + 8 + ((size_t)__builtin_ctzll(a1 ^ b1) >> 3);
    }
    return max_len;
}

/* src/mem.rs:55-70: last dword first, then dwords l = 0,4,.. < len-4 (any order: pure AND) */
static inline int fast_equal(const uint8_t* buf, size_t p1, size_t p2, size_t len,
                             uint32_t p2_last_dword) {
    if (p2_last_dword != ld32(buf + p1 + len - 4)) return 0;
    for (size_t l = 0; l + 4 < len; l += 4) {
        if (ld32(buf + p1 + l) != ld32(buf + p2 + l)) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ coder.rs */

typedef struct { /* src/coder.rs:159-163 */
    uint64_t value;
    unsigned len;
} bitbuf;

static inline uint64_t bb_peek(const bitbuf* b, unsigned len) { /* :167-169 */
    return (b->value >> (b->len - len)) & (((uint64_t)1 << len) - 1);
}
static inline uint64_t bb_get(bitbuf* b, unsigned len) { /* :177-181 */
    uint64_t v = bb_peek(b, len);
    b->len -= len;
    return v;
}
static inline void bb_put(bitbuf* b, unsigned len, uint64_t v) { /* :184-187 */
    b->value = (b->value << len) ^ v;
    b->len += len;
}

typedef struct { /* src/coder.rs:12-16 */
    uint8_t* out;
    size_t pos;
    bitbuf b;
} bitenc;

static inline void enc_reserve(bitenc* e) { /* :84-88 + save_u32 :199-206 (big endian) */
    if (e->b.len >= 32) {
        uint32_t w = (uint32_t)bb_get(&e->b, 32);
        e->out[e->pos + 0] = (uint8_t)(w >> 24);
        e->out[e->pos + 1] = (uint8_t)(w >> 16);
        e->out[e->pos + 2] = (uint8_t)(w >> 8);
        e->out[e->pos + 3] = (uint8_t)(w);
        e->pos += 4;
    }
}
static void enc_varint(bitenc* e, uint32_t v) { /* :27-38 */
    for (;;) {
        enc_reserve(e);
        int has_next = v > 1;
        uint32_t bits = (v & 1) | ((uint32_t)has_next << 1);
        bb_put(&e->b, 2, bits);
        v >>= 1;
        if (!has_next) break;
    }
}
static inline void enc_raw(bitenc* e, uint32_t bits, unsigned n) { /* :40-43 */
    enc_reserve(e);
    bb_put(&e->b, n, bits);
}
static void enc_huffman_table(bitenc* e, const uint8_t* lens, size_t n) { /* :45-67 */
    unsigned maxlen = 0;
    for (size_t i = 0; i < n; i++)
        if (lens[i] > maxlen) maxlen = lens[i];
    enc_varint(e, maxlen);
    size_t last = (size_t)-1;
    for (size_t sym = 0; sym < n; sym++) {
        if (lens[sym] > 0) {
            size_t delta = (last == (size_t)-1) ? sym + 1 : sym - last;
            enc_varint(e, (uint32_t)delta);
            enc_varint(e, maxlen - lens[sym]);
            last = sym;
        }
    }
    enc_varint(e, 0);
}
static size_t enc_finish(bitenc* e) { /* :75-82 + save_all :209-216 */
    enc_reserve(e);
    if (e->b.len > 0) {
        bb_put(&e->b, 32 - e->b.len, 0);
        while (e->b.len > 0) {
            e->out[e->pos++] = (uint8_t)bb_peek(&e->b, 8);
            e->b.len -= e->b.len < 8 ? e->b.len : 8;
        }
    }
    return e->pos;
}

typedef struct { /* src/coder.rs:91-95 */
    const uint8_t* in;
    size_t pos;
    bitbuf b;
} bitdec;

static inline void dec_reserve(bitdec* d) { /* :152-156 + load_u32 :190-196 */
    if (d->b.len < 32) {
        uint32_t w = ((uint32_t)d->in[d->pos] << 24) | ((uint32_t)d->in[d->pos + 1] << 16) |
                     ((uint32_t)d->in[d->pos + 2] << 8) | (uint32_t)d->in[d->pos + 3];
        bb_put(&d->b, 32, w);
        d->pos += 4;
    }
}
static uint32_t dec_varint(bitdec* d) { /* :106-118 */
    uint32_t v = 0;
    for (unsigned sh = 0;; sh++) {
        dec_reserve(d);
        uint32_t bits = (uint32_t)bb_get(&d->b, 2);
        if (sh < 32) v |= (bits & 1) << sh;
        if (!(bits > 1)) break;
    }
    return v;
}
static inline uint32_t dec_raw(bitdec* d, unsigned n) { /* :120-123 */
    dec_reserve(d);
    return (uint32_t)bb_get(&d->b, n);
}

/* ------------------------------------------------------------------ huffman.rs */

typedef struct {
    uint32_t weight;
    uint16_t index;
} hnode;

/* min-heap on (weight, index): src/huffman.rs:28-38 (Ord derived, PartialOrd reversed) */
static inline int hless(hnode a, hnode b) {
    return a.weight < b.weight || (a.weight == b.weight && a.index < b.index);
}
static void hpush(hnode* h, size_t* n, hnode v) {
    size_t i = (*n)++;
    h[i] = v;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!hless(h[i], h[p])) break;
        hnode t = h[i];
        h[i] = h[p];
        h[p] = t;
        i = p;
    }
}
static hnode hpop(hnode* h, size_t* n) {
    hnode top = h[0];
    h[0] = h[--(*n)];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && hless(h[l], h[m])) m = l;
        if (r < *n && hless(h[r], h[m])) m = r;
        if (m == i) break;
        hnode t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
    return top;
}

#define HUFF_MAX_SYMS 512

/* src/huffman.rs:27-111 */
int orc_huffman_lengths(const uint32_t* sym_weights, size_t n, int max_code_len, uint8_t* lens_out) {
    uint32_t w[HUFF_MAX_SYMS * 2];
    uint16_t c1[HUFF_MAX_SYMS * 2], c2[HUFF_MAX_SYMS * 2];
    uint8_t cl[HUFF_MAX_SYMS * 2];
    hnode heap[HUFF_MAX_SYMS];
    for (size_t i = 0; i < n; i++) w[i] = sym_weights[i];
    for (;;) {
        size_t hn = 0, nodes = n;
        for (size_t i = 0; i < n; i++) /* :56-63, filter on the ORIGINAL weights */
            if (sym_weights[i] > 0) {
                hnode v = {w[i], (uint16_t)i};
                hpush(heap, &hn, v);
            }
        if (hn <= 1) { /* :64-71 */
            memset(lens_out, 0, n);
            if (hn == 1) {
                lens_out[heap[0].index] = 1;
                return 1;
            }
            return 0;
        }
        while (hn > 1) { /* :73-88 */
            hnode n1 = hpop(heap, &hn);
            hnode n2 = hpop(heap, &hn);
            w[nodes] = n1.weight + n2.weight;
            c1[nodes] = n1.index;
            c2[nodes] = n2.index;
            hnode v = {n1.weight + n2.weight, (uint16_t)nodes};
            nodes++;
            hpush(heap, &hn, v);
        }
        memset(cl, 0, nodes); /* :91-96 */
        for (size_t i = nodes; i-- > n;) {
            cl[c1[i]] = cl[i] + 1;
            cl[c2[i]] = cl[i] + 1;
        }
        int cur_max = 0;
        for (size_t i = 0; i < n; i++)
            if (cl[i] > cur_max) cur_max = cl[i];
        if (cur_max > max_code_len) { /* :99-108, cumulative shrink of leaf weights */
            uint32_t shrink = 1u << (cur_max - max_code_len);
            for (size_t i = 0; i < n; i++)
                if (w[i] > 0) {
                    uint32_t v = w[i] / shrink;
                    w[i] = v > 1 ? v : 1;
                }
            continue;
        }
        memcpy(lens_out, cl, n);
        return cur_max;
    }
}

/* src/huffman.rs:118-141: canonical codes in (len, sym) order */
void orc_huffman_codes(const uint8_t* lens, size_t n, uint16_t* codes_out) {
    uint16_t bits = 0;
    unsigned cur = 1;
    memset(codes_out, 0, n * sizeof(uint16_t));
    for (unsigned L = 1; L <= 16; L++) {
        for (size_t sym = 0; sym < n; sym++) {
            if (lens[sym] != L) continue;
            if (L > cur) {
                bits = (uint16_t)(bits << (L - cur));
                cur = L;
            }
            codes_out[sym] = bits;
            bits++;
        }
    }
}

typedef struct { /* HuffmanDecoding, src/huffman.rs:144-167 */
    uint16_t sym[1 << 16];
    uint8_t len[1 << 16];
    unsigned max_code_len;
} hdec;

static void hdec_build(hdec* d, const uint8_t* lens, size_t n, unsigned max_code_len) {
    uint16_t codes[HUFF_MAX_SYMS];
    orc_huffman_codes(lens, n, codes);
    d->max_code_len = max_code_len;
    size_t size = (size_t)1 << max_code_len;
    memset(d->sym, 0, size * sizeof(uint16_t));
    memset(d->len, 0, size);
    for (size_t s = 0; s < n; s++) {
        if (lens[s] > 0) {
            unsigned rest = max_code_len - lens[s];
            size_t base = (size_t)codes[s] << rest;
            for (size_t k = 0; k < ((size_t)1 << rest); k++) {
                d->sym[base + k] = (uint16_t)s;
                d->len[base + k] = lens[s];
            }
        }
    }
}

/* Decoder::decode_huffman_table, src/coder.rs:125-142.  Returns number of syms, -1 if bad. */
static long dec_huffman_table(bitdec* d, uint8_t* lens, size_t cap, unsigned* max_code_len) {
    unsigned maxlen = dec_varint(d) & 0xff;
    size_t n = 0;
    if (maxlen > 16) return -1; /* HuffmanTable::new asserts <= 16, src/huffman.rs:20 */
    for (;;) {
        uint32_t delta = dec_varint(d);
        if (delta == 0) break;
        for (uint32_t i = 1; i < delta; i++) {
            if (n >= cap) return -1;
            lens[n++] = 0;
        }
        if (n >= cap) return -1;
        lens[n++] = (uint8_t)(maxlen - (dec_varint(d) & 0xff));
    }
    *max_code_len = maxlen;
    return (long)n;
}
static inline uint16_t dec_huffman_sym(bitdec* d, const hdec* h) { /* src/coder.rs:144-150 */
    dec_reserve(d);
    uint64_t peeked = bb_peek(&d->b, h->max_code_len);
    d->b.len -= h->len[peeked];
    return h->sym[peeked];
}

/* ------------------------------------------------------------------ symrank.rs */

void orc_symrank_new(orc_symrank* s) { /* :22-29 */
    memset(s, 0, sizeof(*s));
    s->cnt = 0;
    s->sum = 1000000;
}
void orc_symrank_init(orc_symrank* s, const uint16_t* values) { /* :31-36 */
    for (unsigned i = 0; i < ORC_NUM_SYMBOLS; i++) {
        s->value[i] = values[i];
        s->index[values[i]] = (uint16_t)i;
    }
}
static inline void symrank_update(orc_symrank* s, uint16_t v, uint16_t i) { /* :61-97 */
    if (s->cnt > ORC_NUM_SYMBOLS) {
        s->cnt = s->cnt * 9 / 10;
        s->sum = s->sum * 9 / 10;
    }
    s->cnt += 1;
    s->sum += i;
    uint16_t dec = (uint16_t)(i / 16 + (uint16_t)(s->sum / 16 / s->cnt));
    uint16_t next_i = i > dec ? (uint16_t)(i - dec) : 0; /* saturating_sub */
    if (next_i < i / 2) next_i = i / 2;
    uint16_t n = i - next_i;
    if (n == 0) {
    } else if (n == 1) {
        uint16_t ni1 = next_i;
        uint16_t nv1 = s->value[ni1];
        s->index[v] = ni1;
        s->value[i] = nv1;
        s->index[nv1] = i;
        s->value[ni1] = v;
    } else {
        uint16_t ni2 = next_i;
        uint16_t ni1 = next_i + n / 2;
        uint16_t nv1 = s->value[ni1];
        uint16_t nv2 = s->value[ni2];
        s->value[i] = nv1;
        s->index[nv1] = i;
        s->value[ni1] = nv2;
        s->index[nv2] = ni1;
        s->value[ni2] = v;
        s->index[v] = ni2;
    }
}
uint16_t orc_symrank_encode(orc_symrank* s, uint16_t v, uint16_t vunlikely) { /* :38-47 */
    uint16_t i = s->index[v];
    uint16_t iu = s->index[vunlikely];
    symrank_update(s, v, i);
    if (i == iu) return ORC_NUM_SYMBOLS - 1;
    return i - (i > iu);
}
uint16_t orc_symrank_decode(orc_symrank* s, uint16_t i, uint16_t vunlikely) { /* :49-59 */
    uint16_t iu = s->index[vunlikely];
    if (i == ORC_NUM_SYMBOLS - 1)
        i = iu;
    else
        i = i + !(i < iu);
    uint16_t v = s->value[i];
    symrank_update(s, v, i);
    return v;
}

/* ------------------------------------------------------------------ matcher.rs */

typedef struct { /* Bucket, src/matcher.rs:28-60 (pos:25 | len_min:7 unpacked here) */
    uint32_t pos[ORC_BUCKET_ITEMS];
    uint8_t len_min[ORC_BUCKET_ITEMS];
    uint8_t len_expected[ORC_BUCKET_ITEMS];
    size_t head;
} bucket;

typedef struct { /* BucketMatcher, src/matcher.rs:102-113 */
    int16_t heads[ORC_BUCKET_HASH];
    int16_t nexts[ORC_BUCKET_ITEMS];
} bmatcher;

typedef struct { /* Match, src/matcher.rs:20-26 */
    size_t reduced_offset, match_len, match_len_expected, match_len_min;
} lzmatch;

static inline size_t nb_add(size_t a, size_t b) { return (a + b) % ORC_BUCKET_ITEMS; } /* :246-248 */
static inline size_t nb_sub(size_t a, size_t b) {                                     /* :251-253 */
    return (a + ORC_BUCKET_ITEMS - b) % ORC_BUCKET_ITEMS;
}

uint32_t orc_hash_entry(const uint8_t* p) { /* hash_dword :256-263, % at :117,136,203 */
    uint32_t h = ((uint32_t)p[0] * 131313131u ^ 797u) + ((uint32_t)p[1] * 1313131u ^ 79797u) +
                 ((uint32_t)p[2] * 13131u ^ 7979797u) + ((uint32_t)p[3] * 131u ^ 797979797u);
    return h % ORC_BUCKET_HASH;
}

static void bucket_update(bucket* b, size_t pos, size_t reduced_offset, size_t match_len) { /* :62-80 */
    size_t new_head = nb_add(b->head, 1);
    if (match_len >= ORC_MATCH_MIN_LEN) {
        size_t ni = nb_sub(b->head, reduced_offset);
        if (b->len_min[ni] <= match_len) {
            size_t v = match_len + 1;
            b->len_min[ni] = (uint8_t)(v < 127 ? v : 127);
        }
    }
    b->pos[new_head] = (uint32_t)pos;
    b->len_min[new_head] = 0;
    b->len_expected[new_head] = (uint8_t)match_len;
    b->head = new_head;
}
static void bucket_forward(bucket* b, size_t forward_len) { /* :82-87 */
    for (size_t i = 0; i < ORC_BUCKET_ITEMS; i++)
        b->pos[i] = b->pos[i] > forward_len ? (uint32_t)(b->pos[i] - forward_len) : 0;
}
static void bm_update(bmatcher* m, const bucket* b, const uint8_t* buf, size_t pos) { /* :115-121 */
    size_t entry = orc_hash_entry(buf + pos);
    m->nexts[b->head] = m->heads[entry];
    m->heads[entry] = (int16_t)b->head;
}
static void bm_forward(bmatcher* m, const bucket* b) { /* :123-133 */
    for (size_t i = 0; i < ORC_BUCKET_HASH; i++)
        if (m->heads[i] != -1 && b->pos[m->heads[i]] == 0) m->heads[i] = -1;
    for (size_t i = 0; i < ORC_BUCKET_ITEMS; i++)
        if (m->nexts[i] != -1 && b->pos[m->nexts[i]] == 0) m->nexts[i] = -1;
}

static lzmatch bm_find_match(const bmatcher* m, const bucket* b, const uint8_t* buf, size_t buf_len,
                             size_t pos, size_t match_depth) { /* :135-192 */
    lzmatch none = {0, 0, 0, 0};
    size_t entry = orc_hash_entry(buf + pos);
    int16_t ni = m->heads[entry];
    if (ni == -1) return none;
    size_t node_index = (size_t)ni;
    size_t max_len = ORC_MATCH_MIN_LEN - 1;
    size_t max_len_min = ORC_MATCH_MIN_LEN;
    size_t max_len_expected = ORC_MATCH_MIN_LEN;
    size_t max_node_index = 0;
    size_t node_pos = b->pos[node_index];
    uint32_t max_len_dword = ld32(buf + pos + max_len - 3);

    for (size_t d = 0; d < match_depth; d++) {
        uint32_t node_dword = ld32(buf + node_pos + max_len - 3);
        if (node_dword == max_len_dword) {
            size_t lcp = common_prefix(buf, node_pos, pos, ORC_MATCH_MAX_LEN);
            if (lcp > max_len) {
                max_len_min = b->len_min[node_index];
                max_len_expected = b->len_expected[node_index];
                max_len = lcp;
                max_node_index = node_index;
                max_len_dword = ld32(buf + pos + max_len - 3);
            }
            if (lcp == ORC_MATCH_MAX_LEN) break;
            if (max_len_expected > 0 && lcp > max_len_expected) break;
        }
        int16_t nx = m->nexts[node_index];
        if (nx == -1) break;
        node_index = (size_t)nx;
        size_t node_pos_next = b->pos[node_index];
        if (node_pos <= node_pos_next) break;
        node_pos = node_pos_next;
    }
    if (max_len >= ORC_MATCH_MIN_LEN && pos + max_len < buf_len) {
        lzmatch r;
        r.reduced_offset = nb_sub(b->head, max_node_index);
        r.match_len = max_len;
        r.match_len_expected = max_len_expected > ORC_MATCH_MIN_LEN ? max_len_expected : ORC_MATCH_MIN_LEN;
        r.match_len_min = max_len_min > ORC_MATCH_MIN_LEN ? max_len_min : ORC_MATCH_MIN_LEN;
        return r;
    }
    return none;
}

static int bm_has_lazy_match(const bmatcher* m, const bucket* b, const uint8_t* buf, size_t pos,
                             size_t min_match_len, size_t depth) { /* :194-228 */
    uint32_t max_len_dword = ld32(buf + pos + min_match_len - 4);
    size_t entry = orc_hash_entry(buf + pos);
    int16_t ni = m->heads[entry];
    if (ni == -1) return 0;
    size_t node_index = (size_t)ni;
    size_t node_pos = b->pos[node_index];
    for (size_t d = 0; d < depth; d++) {
        if (fast_equal(buf, node_pos, pos, min_match_len, max_len_dword)) return 1;
        int16_t nx = m->nexts[node_index];
        if (nx == -1) break;
        node_index = (size_t)nx;
        size_t node_pos_next = b->pos[node_index];
        if (node_pos <= node_pos_next) break;
        node_pos = node_pos_next;
    }
    return 0;
}

/* ------------------------------------------------------------------ lz.rs */

#define LZ_ROID_SIZE 22
#define LZ_LENID_SIZE 6

static uint8_t g_roid_enc_id[ORC_BUCKET_ITEMS];   /* src/lz.rs:494-514 */
static uint8_t g_roid_enc_bits[ORC_BUCKET_ITEMS];
static uint16_t g_roid_enc_rest[ORC_BUCKET_ITEMS];
static uint16_t g_roid_dec_base[LZ_ROID_SIZE]; /* src/lz.rs:516-530 */
static uint8_t g_roid_dec_bits[LZ_ROID_SIZE];
static int g_roid_ready = 0;

static void roid_init(void) {
    if (g_roid_ready) return;
    size_t base = 0, id = 0, idx = 0;
    while (base < ORC_BUCKET_ITEMS) { /* :500-512 */
        size_t bit_len = id / 2;
        for (size_t rest = 0; rest != ((size_t)1 << bit_len); rest++) {
            if (base < ORC_BUCKET_ITEMS) {
                g_roid_enc_id[idx] = (uint8_t)id;
                g_roid_enc_bits[idx] = (uint8_t)bit_len;
                g_roid_enc_rest[idx] = (uint16_t)rest;
                idx++;
                base++;
            }
        }
        id++;
    }
    base = 0;
    id = 0;
    while (base < ORC_BUCKET_ITEMS) { /* :522-528 */
        size_t bit_len = id / 2;
        g_roid_dec_base[id] = (uint16_t)base;
        g_roid_dec_bits[id] = (uint8_t)bit_len;
        id++;
        base += (size_t)1 << bit_len;
    }
    g_roid_ready = 1;
}

static inline int is_alnum(uint8_t c) {
    return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z');
}
static inline size_t hash1(const uint8_t* buf, size_t pos) { /* src/lz.rs:482-486 */
    return (size_t)(buf[pos] & 0x7f) | ((size_t)is_alnum(buf[pos - 1]) << 7);
}
static inline size_t hash2(const uint8_t* buf, size_t pos) { /* src/lz.rs:489-492 */
    return (size_t)(buf[pos] & 0x7f) | (hash1(buf, pos - 1) << 7);
}

typedef struct { /* LZContext, src/lz.rs:49-67 */
    bucket* buckets;      /* 256 */
    orc_symrank* symranks; /* 512 */
    uint8_t (*words)[2];  /* 32768 */
    int first_block;
    int after_literal;
} lzctx;

static int lzctx_init(lzctx* c) {
    c->buckets = (bucket*)calloc(256, sizeof(bucket));
    c->symranks = (orc_symrank*)malloc(512 * sizeof(orc_symrank));
    c->words = (uint8_t(*)[2])calloc(32768, 2);
    if (!c->buckets || !c->symranks || !c->words) return -1;
    for (int i = 0; i < 512; i++) orc_symrank_new(&c->symranks[i]);
    c->first_block = 1;
    c->after_literal = 1;
    return 0;
}
static void lzctx_free(lzctx* c) {
    free(c->buckets);
    free(c->symranks);
    free(c->words);
}

typedef struct { /* MatchItem, src/lz.rs:100-116 */
    uint16_t symbol;
    uint16_t symrank_context;
    uint16_t robits;
    uint8_t symrank_unlikely;
    uint8_t robitlen;
    uint8_t encoded_match_len;
    uint8_t after_literal;
    uint8_t is_match;
} mitem;

struct orc_lz_encoder { /* LZEncoder, src/lz.rs:69-72 */
    lzctx ctx;
    bmatcher* matchers; /* 256 */
    mitem* items;       /* LZ_CHUNK_SIZE */
    orc_trace* trace;
    uint16_t* nodeof;   /* plan-driven encoder only: ring node index of each window position */
};

orc_lz_encoder* orc_lz_encoder_new(void) { /* :75-80 */
    roid_init();
    orc_lz_encoder* e = (orc_lz_encoder*)calloc(1, sizeof(*e));
    if (!e) return NULL;
    if (lzctx_init(&e->ctx)) return NULL;
    e->matchers = (bmatcher*)malloc(256 * sizeof(bmatcher));
    e->items = (mitem*)malloc(ORC_LZ_CHUNK_SIZE * sizeof(mitem));
    memset(e->matchers, 0xff, 256 * sizeof(bmatcher)); /* heads, nexts = -1 */
    return e;
}
void orc_lz_encoder_free(orc_lz_encoder* e) {
    if (!e) return;
    lzctx_free(&e->ctx);
    free(e->matchers);
    free(e->items);
    free(e->nodeof);
    free(e);
}
void orc_lz_encoder_set_trace(orc_lz_encoder* e, orc_trace* t) { e->trace = t; }

void orc_lz_encoder_forward(orc_lz_encoder* e, size_t forward_len) { /* :82-87 */
    for (int i = 0; i < 256; i++) {
        bucket_forward(&e->ctx.buckets[i], forward_len);
        bm_forward(&e->matchers[i], &e->ctx.buckets[i]);
    }
    if (e->nodeof) memmove(e->nodeof, e->nodeof + forward_len, ((size_t)ORC_LZ_BLOCK_SIZE + 1 - forward_len) * sizeof(uint16_t));
}

/* Second half of LZEncoder::encode (src/lz.rs:236-346): census header on the first chunk, symbol ranking,
 * Huffman tables, item emit.  Shared by the reference parse below and by the plan-driven encoder. */
static size_t emit_items(orc_lz_encoder* e, mitem* items, size_t n_items, size_t spos, size_t sbuf_len,
                         uint8_t* tbuf, size_t trace_base) {
    lzctx* c = &e->ctx;
    bitenc enc;
    enc.out = tbuf;
    enc.pos = 0;
    enc.b.value = 0;
    enc.b.len = 0;
    if (c->first_block) { /* :238-265 */
        uint32_t counts[ORC_NUM_SYMBOLS];
        memset(counts, 0, sizeof counts);
        for (size_t i = 0; i < n_items; i++) counts[items[i].symbol]++;
        uint32_t num_counted = 0;
        for (unsigned s = 0; s < ORC_NUM_SYMBOLS; s++) num_counted += counts[s] > 1;
        /* stable sort by Reverse(max(count,1)): insertion sort is stable */
        uint16_t vs[ORC_NUM_SYMBOLS];
        for (unsigned s = 0; s < ORC_NUM_SYMBOLS; s++) {
            uint32_t key = counts[s] > 1 ? counts[s] : 1;
            unsigned j = s;
            while (j > 0) {
                uint32_t kprev = counts[vs[j - 1]] > 1 ? counts[vs[j - 1]] : 1;
                if (kprev >= key) break;
                vs[j] = vs[j - 1];
                j--;
            }
            vs[j] = (uint16_t)s;
        }
        enc_varint(&enc, num_counted);
        for (uint32_t i = 0; i < num_counted; i++) enc_raw(&enc, vs[i], 9);
        orc_symrank init;
        orc_symrank_new(&init);
        orc_symrank_init(&init, vs);
        for (int i = 0; i < 512; i++) c->symranks[i] = init;
        c->first_block = 0;
    }

    enc_varint(&enc, (uint32_t)(spos < sbuf_len ? spos : sbuf_len)); /* :268-269 */
    enc_varint(&enc, (uint32_t)n_items);

    uint32_t w1[2][ORC_NUM_SYMBOLS]; /* :272-305 */
    uint32_t w2[ORC_MATCH_MAX_LEN];
    memset(w1, 0, sizeof w1);
    memset(w2, 0, sizeof w2);
    for (size_t i = 0; i < n_items; i++) {
        mitem* p = &items[i];
        uint16_t r = orc_symrank_encode(&c->symranks[p->symrank_context], p->symbol, p->symrank_unlikely);
        w1[p->after_literal][r]++;
        if (p->is_match && p->encoded_match_len >= LZ_LENID_SIZE - 1) w2[p->encoded_match_len]++;
        p->symbol = r;
        if (e->trace && trace_base + i < e->trace->n) e->trace->items[trace_base + i].rank = r;
    }
    uint8_t l0[ORC_NUM_SYMBOLS], l1[ORC_NUM_SYMBOLS], l2[ORC_MATCH_MAX_LEN]; /* :306-318 */
    uint16_t k0[ORC_NUM_SYMBOLS], k1[ORC_NUM_SYMBOLS], k2[ORC_MATCH_MAX_LEN];
    orc_huffman_lengths(w1[0], ORC_NUM_SYMBOLS, 15, l0);
    orc_huffman_lengths(w1[1], ORC_NUM_SYMBOLS, 15, l1);
    orc_huffman_lengths(w2, ORC_MATCH_MAX_LEN, 15, l2);
    enc_huffman_table(&enc, l0, ORC_NUM_SYMBOLS);
    enc_huffman_table(&enc, l1, ORC_NUM_SYMBOLS);
    enc_huffman_table(&enc, l2, ORC_MATCH_MAX_LEN);
    orc_huffman_codes(l0, ORC_NUM_SYMBOLS, k0);
    orc_huffman_codes(l1, ORC_NUM_SYMBOLS, k1);
    orc_huffman_codes(l2, ORC_MATCH_MAX_LEN, k2);

    for (size_t i = 0; i < n_items; i++) { /* :320-342 */
        const mitem* p = &items[i];
        const uint16_t* kk = p->after_literal ? k1 : k0;
        const uint8_t* ll = p->after_literal ? l1 : l0;
        enc_reserve(&enc);
        bb_put(&enc.b, ll[p->symbol], kk[p->symbol]);
        if (p->is_match) {
            enc_raw(&enc, p->robits, p->robitlen);
            if (p->encoded_match_len >= LZ_LENID_SIZE - 1) {
                enc_reserve(&enc);
                bb_put(&enc.b, l2[p->encoded_match_len], k2[p->encoded_match_len]);
            }
        }
    }
    return enc_finish(&enc);
}

void orc_lz_encoder_encode(orc_lz_encoder* e, const orc_lzcfg* cfg, const uint8_t* sbuf,
                           size_t sbuf_len, uint8_t* tbuf, size_t spos, size_t* spos_out,
                           size_t* tlen_out) { /* :89-346 */
    lzctx* c = &e->ctx;
    mitem* items = e->items;
    size_t n_items = 0;
    size_t trace_base = e->trace ? e->trace->n : 0;

    while (spos < sbuf_len && n_items < ORC_LZ_CHUNK_SIZE) { /* :131 */
        const uint8_t* lwe = c->words[hash2(sbuf, spos - 1)];
        uint8_t last_word_expected[2] = {lwe[0], lwe[1]};
        int last_word_matched = sbuf[spos] == last_word_expected[0] && sbuf[spos + 1] == last_word_expected[1];
        size_t h1 = hash1(sbuf, spos - 1);
        uint16_t symrank_context = (uint16_t)(h1 | ((size_t)c->after_literal << 8));
        uint8_t symrank_unlikely = last_word_expected[0];
        mitem it;
        orc_item tr;
        memset(&tr, 0, sizeof tr);
        tr.pos = (uint32_t)spos;

        int lazy_match_id = 0;
        lzmatch m = bm_find_match(&e->matchers[h1], &c->buckets[h1], sbuf, sbuf_len, spos, cfg->match_depth);
        if (m.match_len > 0) {
            uint8_t roid = g_roid_enc_id[m.reduced_offset];
            uint8_t robitlen = g_roid_enc_bits[m.reduced_offset];
            uint16_t robits = g_roid_enc_rest[m.reduced_offset];
            if (m.match_len < ORC_MATCH_MAX_LEN / 2) { /* :151-170 */
                size_t lazy_len1 = m.match_len + 1 + (robitlen < 8);
                size_t lazy_len2 = lazy_len1 - (size_t)last_word_matched;
                size_t hb1 = hash1(sbuf, spos);
                size_t hb2 = hash1(sbuf, spos + 1);
                if (bm_has_lazy_match(&e->matchers[hb1], &c->buckets[hb1], sbuf, spos + 1, lazy_len1,
                                      cfg->lazy_match_depth1))
                    lazy_match_id = 1;
                else if (bm_has_lazy_match(&e->matchers[hb2], &c->buckets[hb2], sbuf, spos + 2, lazy_len2,
                                           cfg->lazy_match_depth2))
                    lazy_match_id = 2;
            }
            if (lazy_match_id == 0) { /* :172-205 */
                uint8_t enc_len;
                if (m.match_len > m.match_len_expected)
                    enc_len = (uint8_t)(m.match_len - m.match_len_min);
                else if (m.match_len < m.match_len_expected)
                    enc_len = (uint8_t)(m.match_len - m.match_len_min + 1);
                else
                    enc_len = 0;
                uint8_t lenid = enc_len < LZ_LENID_SIZE - 1 ? enc_len : LZ_LENID_SIZE - 1;
                it.symbol = (uint16_t)(256 + roid * LZ_LENID_SIZE + lenid);
                it.symrank_context = symrank_context;
                it.symrank_unlikely = symrank_unlikely;
                it.robitlen = robitlen;
                it.robits = robits;
                it.encoded_match_len = enc_len;
                it.after_literal = (uint8_t)c->after_literal;
                it.is_match = 1;
                items[n_items++] = it;
                if (e->trace && e->trace->n < e->trace->cap) {
                    tr.symbol = it.symbol;
                    tr.ctx = symrank_context;
                    tr.unlikely = symrank_unlikely;
                    tr.reduced_offset = (uint16_t)m.reduced_offset;
                    tr.match_len = (uint8_t)m.match_len;
                    tr.enc_len = enc_len;
                    tr.after_literal = it.after_literal;
                    e->trace->items[e->trace->n++] = tr;
                }
                bucket_update(&c->buckets[h1], spos, m.reduced_offset, m.match_len);
                bm_update(&e->matchers[h1], &c->buckets[h1], sbuf, spos);
                spos += m.match_len;
                c->after_literal = 0;
                size_t k = hash2(sbuf, spos - 3);
                c->words[k][0] = sbuf[spos - 2];
                c->words[k][1] = sbuf[spos - 1];
                continue;
            }
        }
        bucket_update(&c->buckets[h1], spos, 0, 0); /* :207-212 */
        bm_update(&e->matchers[h1], &c->buckets[h1], sbuf, spos);

        it.symrank_context = symrank_context;
        it.symrank_unlikely = symrank_unlikely;
        it.robitlen = 0;
        it.robits = 0;
        it.encoded_match_len = 0;
        it.after_literal = (uint8_t)c->after_literal;
        it.is_match = 0;
        if (spos + 1 < sbuf_len && lazy_match_id != 1 && last_word_matched) { /* :215-223 */
            it.symbol = ORC_WORD_SYMBOL;
            items[n_items++] = it;
            spos += 2;
            c->after_literal = 0;
        } else { /* :224-234 */
            it.symbol = sbuf[spos];
            items[n_items++] = it;
            spos += 1;
            c->after_literal = 1;
            size_t k = hash2(sbuf, spos - 3);
            c->words[k][0] = sbuf[spos - 2];
            c->words[k][1] = sbuf[spos - 1];
        }
        if (e->trace && e->trace->n < e->trace->cap) {
            tr.symbol = it.symbol;
            tr.ctx = symrank_context;
            tr.unlikely = symrank_unlikely;
            tr.after_literal = it.after_literal;
            e->trace->items[e->trace->n++] = tr;
        }
    }

    *spos_out = spos;
    *tlen_out = emit_items(e, items, n_items, spos, sbuf_len, tbuf, trace_base);
}

/* ---- plan-driven encoder ------------------------------------------------------------------
 * The reference decoder accepts ANY parse that its state machine can express.
 * orc_lz_encoder_encode_plan encodes a caller-supplied parse (item starts, types, match sources and
 * lengths) with exactly the state updates of LZEncoder::encode (src/lz.rs:191-233) and the same emit
 * half (emit_items above) -- i.e. "what the reference encoder would write had its match finder made
 * these choices".  Every item is checked for representability (src/lz.rs:173-177,459-467): source is a
 * live ring node of the item's context, bytes equal, 4 <= len <= 240, len >= max(len_min,4), WORD
 * prediction true.  It is the checker for the GPU's fast parse mode; the product never calls it. */
static int plan_fail(orc_plan_error* err, size_t pos, int code) {
    if (err) { err->pos = pos; err->code = code; }
    return -1;
}
int orc_lz_encoder_encode_plan(orc_lz_encoder* e, const uint8_t* sbuf, size_t sbuf_len, uint8_t* tbuf,
                               size_t spos, const orc_plan_item* plan, size_t nplan, size_t* nused_out,
                               size_t* spos_out, size_t* tlen_out, orc_plan_error* err) {
    lzctx* c = &e->ctx;
    mitem* items = e->items;
    size_t n_items = 0, ip = 0;
    size_t trace_base = e->trace ? e->trace->n : 0;
    if (!e->nodeof) {
        e->nodeof = (uint16_t*)calloc((size_t)ORC_LZ_BLOCK_SIZE + 1, sizeof(uint16_t));
        if (!e->nodeof) return plan_fail(err, spos, ORC_PLAN_ENOMEM);
    }
    while (spos < sbuf_len && n_items < ORC_LZ_CHUNK_SIZE) {
        if (ip >= nplan) return plan_fail(err, spos, ORC_PLAN_ESHORT);
        const orc_plan_item* d = &plan[ip++];
        if (d->pos != spos) return plan_fail(err, spos, ORC_PLAN_EPOS);
        const uint8_t* lwe = c->words[hash2(sbuf, spos - 1)];
        uint8_t w0 = lwe[0], w1 = lwe[1];
        size_t h1 = hash1(sbuf, spos - 1);
        bucket* b = &c->buckets[h1];
        mitem it;
        orc_item tr;
        memset(&tr, 0, sizeof tr);
        tr.pos = (uint32_t)spos;
        it.symrank_context = (uint16_t)(h1 | ((size_t)c->after_literal << 8));
        it.symrank_unlikely = w0;
        it.robitlen = 0;
        it.robits = 0;
        it.encoded_match_len = 0;
        it.after_literal = (uint8_t)c->after_literal;
        it.is_match = 0;
        size_t adv;
        if (d->type == ORC_PLAN_MATCH) {
            size_t L = d->len, q = d->src;
            if (L < ORC_MATCH_MIN_LEN || L > ORC_MATCH_MAX_LEN) return plan_fail(err, spos, ORC_PLAN_ELEN);
            if (spos + L > sbuf_len) return plan_fail(err, spos, ORC_PLAN_EEND);
            if (q == 0 || q >= spos) return plan_fail(err, spos, ORC_PLAN_ESRC);
            size_t node = e->nodeof[q];
            if (b->pos[node] != q) return plan_fail(err, spos, ORC_PLAN_ESRC); /* not (any more) in this ring */
            if (memcmp(sbuf + q, sbuf + spos, L) != 0 && q + L <= spos) return plan_fail(err, spos, ORC_PLAN_EBYTES);
            for (size_t k = 0; k < L; k++) /* overlapping sources compare byte-wise like the decoder copies */
                if (sbuf[q + k] != sbuf[spos + k]) return plan_fail(err, spos, ORC_PLAN_EBYTES);
            size_t ro = nb_sub(b->head, node);
            size_t lmin = b->len_min[node] > ORC_MATCH_MIN_LEN ? b->len_min[node] : ORC_MATCH_MIN_LEN;
            size_t lexp = b->len_expected[node] > ORC_MATCH_MIN_LEN ? b->len_expected[node] : ORC_MATCH_MIN_LEN;
            if (L < lmin) return plan_fail(err, spos, ORC_PLAN_ELENMIN);
            uint8_t enc_len = L > lexp ? (uint8_t)(L - lmin) : (L < lexp ? (uint8_t)(L - lmin + 1) : 0);
            uint8_t lenid = enc_len < LZ_LENID_SIZE - 1 ? enc_len : LZ_LENID_SIZE - 1;
            it.symbol = (uint16_t)(256 + g_roid_enc_id[ro] * LZ_LENID_SIZE + lenid);
            it.robitlen = g_roid_enc_bits[ro];
            it.robits = g_roid_enc_rest[ro];
            it.encoded_match_len = enc_len;
            it.is_match = 1;
            tr.reduced_offset = (uint16_t)ro;
            tr.match_len = (uint8_t)L;
            tr.enc_len = enc_len;
            bucket_update(b, spos, ro, L);
            adv = L;
        } else if (d->type == ORC_PLAN_WORD) {
            if (spos + 1 >= sbuf_len) return plan_fail(err, spos, ORC_PLAN_EEND);
            if (sbuf[spos] != w0 || sbuf[spos + 1] != w1) return plan_fail(err, spos, ORC_PLAN_EWORD);
            it.symbol = ORC_WORD_SYMBOL;
            bucket_update(b, spos, 0, 0);
            adv = 2;
        } else if (d->type == ORC_PLAN_LITERAL) {
            it.symbol = sbuf[spos];
            bucket_update(b, spos, 0, 0);
            adv = 1;
        } else {
            return plan_fail(err, spos, ORC_PLAN_ETYPE);
        }
        e->nodeof[spos] = (uint16_t)b->head;
        items[n_items++] = it;
        if (e->trace && e->trace->n < e->trace->cap) {
            tr.symbol = it.symbol;
            tr.ctx = it.symrank_context;
            tr.unlikely = w0;
            tr.after_literal = it.after_literal;
            e->trace->items[e->trace->n++] = tr;
        }
        spos += adv;
        c->after_literal = d->type == ORC_PLAN_LITERAL;
        if (d->type != ORC_PLAN_WORD) { /* src/lz.rs:203,233 */
            size_t k = hash2(sbuf, spos - 3);
            c->words[k][0] = sbuf[spos - 2];
            c->words[k][1] = sbuf[spos - 1];
        }
    }
    *nused_out = ip;
    *spos_out = spos;
    *tlen_out = emit_items(e, items, n_items, spos, sbuf_len, tbuf, trace_base);
    return 0;
}

struct orc_lz_decoder { /* LZDecoder, src/lz.rs:348-350 */
    lzctx ctx;
    hdec* h0;
    hdec* h1;
    hdec* h2;
};

orc_lz_decoder* orc_lz_decoder_new(void) {
    roid_init();
    orc_lz_decoder* d = (orc_lz_decoder*)calloc(1, sizeof(*d));
    if (!d) return NULL;
    if (lzctx_init(&d->ctx)) return NULL;
    d->h0 = (hdec*)malloc(sizeof(hdec));
    d->h1 = (hdec*)malloc(sizeof(hdec));
    d->h2 = (hdec*)malloc(sizeof(hdec));
    return d;
}
void orc_lz_decoder_free(orc_lz_decoder* d) {
    if (!d) return;
    lzctx_free(&d->ctx);
    free(d->h0);
    free(d->h1);
    free(d->h2);
    free(d);
}
void orc_lz_decoder_forward(orc_lz_decoder* d, size_t forward_len) { /* :359-364 */
    for (int i = 0; i < 256; i++) bucket_forward(&d->ctx.buckets[i], forward_len);
}

/* src/mem.rs:74-92.  Byte-serial copy gives the same bytes in [pdst, pdst+len) as the
 * reference's dword copy; bytes past pdst+len (which the reference scribbles on) are always
 * rewritten by later items before anything reads them. */
static inline void match_copy(uint8_t* buf, size_t psrc, size_t pdst, size_t len) {
    for (size_t i = 0; i < len; i++) buf[pdst + i] = buf[psrc + i];
}

int orc_lz_decoder_decode(orc_lz_decoder* d, const uint8_t* tbuf, size_t tlen, uint8_t* sbuf,
                          size_t spos, size_t* spos_end_out) { /* :366-478 */
    lzctx* c = &d->ctx;
    bitdec dec;
    dec.in = tbuf;
    dec.pos = 0;
    dec.b.value = 0;
    dec.b.len = 0;
    (void)tlen;

    if (c->first_block) { /* :372-392 */
        size_t num = dec_varint(&dec);
        uint16_t vs[ORC_NUM_SYMBOLS];
        uint8_t set[ORC_NUM_SYMBOLS];
        memset(vs, 0, sizeof vs);
        memset(set, 0, sizeof set);
        if (num > ORC_NUM_SYMBOLS) return -1;
        for (size_t i = 0; i < num; i++) {
            vs[i] = (uint16_t)dec_raw(&dec, 9);
            if (vs[i] >= ORC_NUM_SYMBOLS) return -1;
            set[vs[i]] = 1;
        }
        for (unsigned i = 0; i < ORC_NUM_SYMBOLS; i++)
            if (!set[i]) {
                if (num >= ORC_NUM_SYMBOLS) return -1;
                vs[num++] = (uint16_t)i;
            }
        orc_symrank init;
        orc_symrank_new(&init);
        orc_symrank_init(&init, vs);
        for (int i = 0; i < 512; i++) c->symranks[i] = init;
        c->first_block = 0;
    }
    size_t sbuf_len = dec_varint(&dec); /* :396-397 */
    size_t n_items = dec_varint(&dec);

    uint8_t l0[HUFF_MAX_SYMS], l1[HUFF_MAX_SYMS], l2[HUFF_MAX_SYMS]; /* :400-409 */
    unsigned m0 = 0, m1 = 0, m2 = 0;
    long n0 = dec_huffman_table(&dec, l0, HUFF_MAX_SYMS, &m0);
    long n1 = dec_huffman_table(&dec, l1, HUFF_MAX_SYMS, &m1);
    long n2 = dec_huffman_table(&dec, l2, HUFF_MAX_SYMS, &m2);
    if (n0 < 0 || n1 < 0 || n2 < 0) return -1;
    hdec_build(d->h0, l0, (size_t)n0, m0);
    hdec_build(d->h1, l1, (size_t)n1, m1);
    hdec_build(d->h2, l2, (size_t)n2, m2);

    for (size_t it = 0; it < n_items; it++) { /* :411-476 */
        uint16_t symbol = dec_huffman_sym(&dec, c->after_literal ? d->h1 : d->h0);
        if (symbol > ORC_NUM_SYMBOLS) return -1; /* :413-415 */
        if (symbol >= ORC_NUM_SYMBOLS) return -1;
        if (dec.pos > tlen + 8) return -1; /* ran off the chunk: corrupt input */
        size_t h1 = hash1(sbuf, spos - 1);
        bucket* cur = &c->buckets[h1];
        uint8_t* lwe = c->words[hash2(sbuf, spos - 1)];
        uint16_t symrank_context = (uint16_t)(h1 | ((size_t)c->after_literal << 8));
        uint8_t unlikely = lwe[0];
        uint16_t v = orc_symrank_decode(&c->symranks[symrank_context], symbol, unlikely);
        if (v == ORC_WORD_SYMBOL) { /* :425-430 */
            bucket_update(cur, spos, 0, 0);
            c->after_literal = 0;
            sbuf[spos] = lwe[0];
            sbuf[spos + 1] = lwe[1];
            spos += 2;
        } else if (v < 256) { /* :431-437 */
            bucket_update(cur, spos, 0, 0);
            c->after_literal = 1;
            sbuf[spos] = (uint8_t)v;
            spos += 1;
            size_t k = hash2(sbuf, spos - 3);
            c->words[k][0] = sbuf[spos - 2];
            c->words[k][1] = sbuf[spos - 1];
        } else { /* :438-474 */
            unsigned roid = (unsigned)(v - 256) / LZ_LENID_SIZE;
            unsigned lenid = (unsigned)(v - 256) % LZ_LENID_SIZE;
            size_t reduced_offset = g_roid_dec_base[roid] + dec_raw(&dec, g_roid_dec_bits[roid]);
            if (reduced_offset >= ORC_BUCKET_ITEMS) return -1;
            size_t node = nb_sub(cur->head, reduced_offset);
            size_t enc_len = lenid == LZ_LENID_SIZE - 1 ? dec_huffman_sym(&dec, d->h2) : lenid;
            size_t match_pos = cur->pos[node];
            size_t len_min = cur->len_min[node] > ORC_MATCH_MIN_LEN ? cur->len_min[node] : ORC_MATCH_MIN_LEN;
            size_t len_exp = cur->len_expected[node] > ORC_MATCH_MIN_LEN ? cur->len_expected[node] : ORC_MATCH_MIN_LEN;
            size_t match_len;
            if (enc_len + len_min > len_exp)
                match_len = enc_len + len_min;
            else if (enc_len > 0)
                match_len = enc_len + len_min - 1;
            else
                match_len = len_exp;
            if (match_pos >= spos || spos + match_len > ORC_LZ_BLOCK_SIZE + ORC_SENTINEL_LEN) return -1;
            bucket_update(cur, spos, reduced_offset, match_len);
            c->after_literal = 0;
            match_copy(sbuf, match_pos, spos, match_len);
            spos += match_len;
            size_t k = hash2(sbuf, spos - 3);
            c->words[k][0] = sbuf[spos - 2];
            c->words[k][1] = sbuf[spos - 1];
        }
        if (spos > ORC_LZ_BLOCK_SIZE) return -1;
    }
    *spos_end_out = spos < sbuf_len ? spos : sbuf_len; /* :477 */
    return 0;
}

/* ------------------------------------------------------------------ lib.rs / ioutil.rs */

typedef struct {
    uint8_t* p;
    size_t n, cap;
} obuf;

static int ob_write(obuf* o, const uint8_t* s, size_t n) {
    if (o->n + n > o->cap) {
        size_t nc = o->cap ? o->cap : 65536;
        while (nc < o->n + n) nc *= 2;
        uint8_t* np = (uint8_t*)realloc(o->p, nc);
        if (!np) return -1;
        o->p = np;
        o->cap = nc;
    }
    memcpy(o->p + o->n, s, n);
    o->n += n;
    return 0;
}
static int ob_write_len(obuf* o, size_t len) { /* src/ioutil.rs:79-88 */
    uint8_t b;
    while (len >= 128) {
        b = (uint8_t)(128 + len % 128);
        len /= 128;
        if (ob_write(o, &b, 1)) return -1;
    }
    b = (uint8_t)len;
    return ob_write(o, &b, 1);
}

int orc_encode_mem(const uint8_t* src, size_t src_len, const orc_lzcfg* cfg, uint8_t** dst,
                   size_t* dst_len, orc_trace* trace) { /* src/lib.rs:58-92 */
    orc_lz_encoder* enc = orc_lz_encoder_new();
    uint8_t* sbvec_buf = (uint8_t*)calloc(ORC_LZ_BLOCK_SIZE + ORC_SENTINEL_LEN * 2, 1);
    uint8_t* tbvec = (uint8_t*)malloc((size_t)ORC_PREMATCH_LEN * 3);
    obuf out = {NULL, 0, 0};
    int rc = -1;
    if (!enc || !sbvec_buf || !tbvec) goto done;
    orc_lz_encoder_set_trace(enc, trace);
    if (trace) trace->n = 0;
    uint8_t* sbvec = sbvec_buf + ORC_SENTINEL_LEN;
    size_t off = 0;
    for (;;) {
        size_t room = ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN;
        size_t n = src_len - off < room ? src_len - off : room; /* read_repeatedly :42-52,72 */
        if (n == 0) break;
        memcpy(sbvec + ORC_PREMATCH_LEN, src + off, n);
        off += n;
        size_t spos = ORC_PREMATCH_LEN;
        while (spos < ORC_PREMATCH_LEN + n) { /* :76-82 */
            size_t s, t;
            orc_lz_encoder_encode(enc, cfg, sbvec, ORC_PREMATCH_LEN + n, tbvec, spos, &s, &t);
            if (ob_write_len(&out, t) || ob_write(&out, tbvec, t)) goto done;
            spos = s;
        }
        memmove(sbvec, sbvec + ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN, ORC_PREMATCH_LEN); /* :83 */
        orc_lz_encoder_forward(enc, ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN);               /* :84 */
    }
    if (ob_write_len(&out, 0)) goto done; /* :89 */
    *dst = out.p;
    *dst_len = out.n;
    out.p = NULL;
    rc = 0;
done:
    free(out.p);
    free(sbvec_buf);
    free(tbvec);
    orc_lz_encoder_free(enc);
    return rc;
}

/* orz::encode (src/lib.rs:58-92) with a caller-supplied parse in STREAM offsets. */
int orc_encode_plan_mem(const uint8_t* src, size_t src_len, const orc_plan_item* plan, size_t nplan,
                        uint8_t** dst, size_t* dst_len, orc_trace* trace, orc_plan_error* err) {
    orc_lz_encoder* enc = orc_lz_encoder_new();
    uint8_t* sbvec_buf = (uint8_t*)calloc(ORC_LZ_BLOCK_SIZE + ORC_SENTINEL_LEN * 2, 1);
    uint8_t* tbvec = (uint8_t*)malloc((size_t)ORC_PREMATCH_LEN * 3);
    orc_plan_item* wplan = (orc_plan_item*)malloc(((size_t)ORC_LZ_CHUNK_SIZE + 1) * sizeof(orc_plan_item));
    obuf out = {NULL, 0, 0};
    int rc = -1;
    if (!enc || !sbvec_buf || !tbvec || !wplan) goto done;
    orc_lz_encoder_set_trace(enc, trace);
    if (trace) trace->n = 0;
    uint8_t* sbvec = sbvec_buf + ORC_SENTINEL_LEN;
    size_t off = 0, ip = 0;
    for (;;) {
        size_t room = ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN;
        size_t n = src_len - off < room ? src_len - off : room;
        if (n == 0) break;
        memcpy(sbvec + ORC_PREMATCH_LEN, src + off, n);
        size_t spos = ORC_PREMATCH_LEN;
        while (spos < ORC_PREMATCH_LEN + n) {
            /* stream offsets -> window offsets for the items that can fall into this chunk */
            size_t k = 0;
            while (k < ORC_LZ_CHUNK_SIZE && ip + k < nplan && plan[ip + k].pos < off + n) {
                wplan[k] = plan[ip + k];
                wplan[k].pos = (uint32_t)(plan[ip + k].pos - off + ORC_PREMATCH_LEN);
                if (wplan[k].type == ORC_PLAN_MATCH) {
                    uint64_t back = (uint64_t)plan[ip + k].pos - plan[ip + k].src;
                    wplan[k].src = back < wplan[k].pos ? (uint32_t)(wplan[k].pos - back) : 0;
                }
                k++;
            }
            size_t s, t, used;
            if (orc_lz_encoder_encode_plan(enc, sbvec, ORC_PREMATCH_LEN + n, tbvec, spos, wplan, k, &used, &s, &t, err)) {
                if (err) err->pos = err->pos - ORC_PREMATCH_LEN + off;
                goto done;
            }
            ip += used;
            if (ob_write_len(&out, t) || ob_write(&out, tbvec, t)) goto done;
            spos = s;
        }
        off += n;
        memmove(sbvec, sbvec + ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN, ORC_PREMATCH_LEN);
        orc_lz_encoder_forward(enc, ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN);
    }
    if (ip != nplan) { plan_fail(err, off, ORC_PLAN_ESHORT); goto done; }
    if (ob_write_len(&out, 0)) goto done;
    *dst = out.p;
    *dst_len = out.n;
    out.p = NULL;
    rc = 0;
done:
    free(out.p);
    free(sbvec_buf);
    free(tbvec);
    free(wplan);
    orc_lz_encoder_free(enc);
    return rc;
}

int orc_decode_mem(const uint8_t* src, size_t src_len, uint8_t** dst, size_t* dst_len,
                   size_t* consumed) { /* src/lib.rs:94-129 */
    orc_lz_decoder* dec = orc_lz_decoder_new();
    uint8_t* sbvec_buf = (uint8_t*)calloc((size_t)ORC_LZ_BLOCK_SIZE * 2 + ORC_SENTINEL_LEN * 2, 1);
    size_t tcap = (size_t)ORC_PREMATCH_LEN * 3;
    uint8_t* tbvec = (uint8_t*)calloc(tcap + 16, 1);
    obuf out = {NULL, 0, 0};
    int rc = -1;
    if (!dec || !sbvec_buf || !tbvec) goto done;
    uint8_t* sbvec = sbvec_buf + ORC_SENTINEL_LEN;
    size_t spos = ORC_PREMATCH_LEN, ip = 0;
    for (;;) {
        size_t t = 0, factor = 1; /* read_len, src/ioutil.rs:60-76 */
        for (;;) {
            if (ip >= src_len) goto done; /* UnexpectedEof */
            uint8_t v = src[ip++];
            if (v < 128) {
                t += (size_t)v * factor;
                break;
            }
            t += (size_t)(v - 128) * factor;
            factor *= 128;
        }
        if (t == 0) break;
        if (t >= tcap) goto done; /* :111-113 InvalidData */
        if (ip + t > src_len) goto done;
        memcpy(tbvec, src + ip, t);
        memset(tbvec + t, 0, 16);
        ip += t;
        size_t spos_end;
        if (orc_lz_decoder_decode(dec, tbvec, t, sbvec, spos, &spos_end)) goto done;
        if (spos_end < spos) goto done;
        if (ob_write(&out, sbvec + spos, spos_end - spos)) goto done;
        spos = spos_end;
        if (spos >= ORC_LZ_BLOCK_SIZE) { /* :120-125 */
            memmove(sbvec, sbvec + ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN, ORC_PREMATCH_LEN);
            orc_lz_decoder_forward(dec, ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN);
            spos = ORC_PREMATCH_LEN;
        }
    }
    if (!out.p) out.p = (uint8_t*)malloc(1);
    *dst = out.p;
    *dst_len = out.n;
    if (consumed) *consumed = ip;
    out.p = NULL;
    rc = 0;
done:
    free(out.p);
    free(sbvec_buf);
    free(tbvec);
    orc_lz_decoder_free(dec);
    return rc;
}

void orc_free(void* p) { free(p); }

/* src/coder.rs:224-265 -- the reference's only unit test, restated */
long orc_coder_selftest(const uint8_t* input, size_t n, uint8_t* encoded, size_t cap) {
    uint32_t weights[256];
    uint8_t lens[256];
    uint16_t codes[256];
    memset(weights, 0, sizeof weights);
    for (size_t i = 0; i < n; i++) weights[input[i]]++;
    orc_huffman_lengths(weights, 256, 15, lens);
    orc_huffman_codes(lens, 256, codes);
    if (cap < n * 2 + 2048) return -1;
    memset(encoded, 0, cap);
    bitenc enc = {encoded, 0, {0, 0}};
    enc_varint(&enc, (uint32_t)n);
    enc_huffman_table(&enc, lens, 256);
    for (size_t i = 0; i < n; i++) {
        enc_reserve(&enc);
        bb_put(&enc.b, lens[input[i]], codes[input[i]]);
    }
    size_t elen = enc_finish(&enc);
    bitdec dec = {encoded, 0, {0, 0}};
    uint32_t num = dec_varint(&dec);
    uint8_t dl[HUFF_MAX_SYMS];
    unsigned maxlen;
    long ns = dec_huffman_table(&dec, dl, HUFF_MAX_SYMS, &maxlen);
    if (ns < 0 || num != n) return -1;
    hdec* h = (hdec*)malloc(sizeof(hdec));
    hdec_build(h, dl, (size_t)ns, maxlen);
    long ok = (long)elen;
    for (size_t i = 0; i < n; i++)
        if (dec_huffman_sym(&dec, h) != input[i]) ok = -1;
    free(h);
    return ok;
}
