/*
 * orz_diag.c -- forensic decoder (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).
 *
 * Decodes an orz stream with the oracle's decoder state machine (LZDecoder::decode,
 * /root/reference/src/lz.rs:366-478, driven like orz::decode, src/lib.rs:94-129) while holding the
 * EXPECTED plain bytes beside it: the first item whose output differs from the expectation -- or on
 * which the decoder gives up -- is reported with everything the decoder knew about it (context, ring
 * node, len_min / len_expected, the word prediction).  An undecodable stream is Huffman-coded noise
 * after its first wrong item; this names that item.
 *
 * Built as oracle/liborz_diag.so (make -C oracle diag); used by tools/dev/hunt.py and the GPU tests'
 * failure reports.  It includes orz_oracle.c to reach its static functions.
 */
#include "orz_oracle.c"

typedef struct {
    int32_t kind;            /* 0 = the stream decodes to `expect`; 1 = an item wrote other bytes; 2 = decoder gave up (cause) ;
                                3 = container level (length prefix / truncated / size mismatch) */
    int32_t cause;           /* kind 2: 1 symbol out of range, 2 ran off the chunk, 3 reduced offset >= ring, 4 source at or after spos
                                / beyond the window, 5 spos beyond the block, 6 table / header */
    uint64_t stream_off;     /* plain offset of the item's first byte */
    uint64_t item_index;     /* items decoded before it (whole stream) */
    uint32_t block, chunk, item_in_chunk;
    uint32_t spos;           /* window offset of the item */
    uint32_t type;           /* 0 WORD, 1 literal, 2 match */
    uint32_t symbol_rank;    /* Huffman-decoded rank */
    uint32_t symbol;         /* after the symbol ranking */
    uint32_t ctx, after_literal, unlikely;
    uint32_t reduced_offset, node, node_pos, node_len_min, node_len_expected, enc_len, match_len;
    uint32_t true_lcp;       /* common prefix of expect at the item and expect at (node_pos), capped at 240 */
    uint32_t src_ctx;        /* hash1 of the byte before node_pos (from the decoded window) */
    uint32_t word0, word1, want0, want1; /* WORD / literal: predicted / written bytes and the expected ones */
    uint32_t first_bad;      /* offset inside the item of the first wrong byte */
    uint32_t ring_count;     /* items the ring of ctx has taken so far (all blocks) */
    /* a failing match: the ring nodes at reduced offsets ro-16 .. ro+16 -- their positions and their true common prefix with the item */
    uint32_t near_pos[33], near_lcp[33], near_exp[33], near_min[33];
    /* the symbol-ranking table of the item's context BEFORE the item: the ranks the expected literal / WORD would have had, the
       symbols at ranks symbol_rank-4 .. +4 (after exclusion is undone: raw table ranks), counters */
    uint32_t lit_rank, word_rank, unl_index, tab_cnt, tab_sum, tab_near[9], tab_raw_index;
    /* the ring node (whole ring searched) with the longest true common prefix with the item */
    uint32_t best_ro, best_pos, best_lcp, best_exp, best_min;
} orc_diag;

typedef struct {
    orc_lz_decoder* d;
    uint64_t ring_items[256];
} diag_state;

/* optional item log: (window offset, Huffman-decoded rank, symbol after the ranking, context | after_literal << 8) per item */
static uint64_t g_force_at = ~0ull; static int g_force_al = 0; /* hypothesis test: decode item g_force_at as if after_literal were g_force_al */
void orc_diag_force_al(uint64_t item, int al) { g_force_at = item; g_force_al = al; }
static uint64_t g_detail_at = ~0ull; /* item whose table / ring neighbourhood is recorded (second run of a failing stream) */
static uint32_t* g_log = NULL;
static size_t g_log_cap = 0, g_log_n = 0;
void orc_diag_set_log(uint32_t* buf, size_t cap_items) { g_log = buf; g_log_cap = cap_items; g_log_n = 0; }
size_t orc_diag_log_items(void) { return g_log_n; }

static uint32_t lcp_cap(const uint8_t* a, const uint8_t* b, size_t max) {
    uint32_t l = 0;
    while (l < max && a[l] == b[l]) l++;
    return l;
}

/* one chunk; `exp_win` = expected window contents aligned with sbuf (valid for [P - hist, end)), returns 0 ok, 1 stop (out filled) */
static int diag_chunk(diag_state* st, const uint8_t* tbuf, size_t tlen, uint8_t* sbuf, const uint8_t* exp_win, size_t exp_end,
                      size_t spos, size_t* spos_end_out, orc_diag* out, uint64_t base_off, uint64_t* items_done) {
    orc_lz_decoder* d = st->d;
    lzctx* c = &d->ctx;
    bitdec dec;
    dec.in = tbuf; dec.pos = 0; dec.b.value = 0; dec.b.len = 0;
    if (c->first_block) {
        size_t num = dec_varint(&dec);
        uint16_t vs[ORC_NUM_SYMBOLS];
        uint8_t set[ORC_NUM_SYMBOLS];
        memset(vs, 0, sizeof vs); memset(set, 0, sizeof set);
        if (num > ORC_NUM_SYMBOLS) { out->kind = 2; out->cause = 6; return 1; }
        for (size_t i = 0; i < num; i++) {
            vs[i] = (uint16_t)dec_raw(&dec, 9);
            if (vs[i] >= ORC_NUM_SYMBOLS) { out->kind = 2; out->cause = 6; return 1; }
            set[vs[i]] = 1;
        }
        for (unsigned i = 0; i < ORC_NUM_SYMBOLS; i++)
            if (!set[i]) { if (num >= ORC_NUM_SYMBOLS) { out->kind = 2; out->cause = 6; return 1; } vs[num++] = (uint16_t)i; }
        orc_symrank init;
        orc_symrank_new(&init);
        orc_symrank_init(&init, vs);
        for (int i = 0; i < 512; i++) c->symranks[i] = init;
        c->first_block = 0;
    }
    size_t sbuf_len = dec_varint(&dec);
    size_t n_items = dec_varint(&dec);
    uint8_t l0[HUFF_MAX_SYMS], l1[HUFF_MAX_SYMS], l2[HUFF_MAX_SYMS];
    unsigned m0 = 0, m1 = 0, m2 = 0;
    long n0 = dec_huffman_table(&dec, l0, HUFF_MAX_SYMS, &m0);
    long n1 = dec_huffman_table(&dec, l1, HUFF_MAX_SYMS, &m1);
    long n2 = dec_huffman_table(&dec, l2, HUFF_MAX_SYMS, &m2);
    if (n0 < 0 || n1 < 0 || n2 < 0) { out->kind = 2; out->cause = 6; return 1; }
    hdec_build(d->h0, l0, (size_t)n0, m0);
    hdec_build(d->h1, l1, (size_t)n1, m1);
    hdec_build(d->h2, l2, (size_t)n2, m2);
    for (size_t it = 0; it < n_items; it++) {
        if (*items_done == g_force_at) c->after_literal = g_force_al;
        const int al = c->after_literal;
        uint16_t symbol = dec_huffman_sym(&dec, al ? d->h1 : d->h0);
        memset(out, 0, sizeof *out);
        out->item_in_chunk = (uint32_t)it;
        out->item_index = *items_done;
        out->spos = (uint32_t)spos;
        out->stream_off = base_off + (spos - ORC_PREMATCH_LEN);
        out->symbol_rank = symbol;
        out->after_literal = (uint32_t)al;
        if (symbol >= ORC_NUM_SYMBOLS) { out->kind = 2; out->cause = 1; return 1; }
        if (dec.pos > tlen + 8) { out->kind = 2; out->cause = 2; return 1; }
        size_t h1 = hash1(sbuf, spos - 1);
        bucket* cur = &c->buckets[h1];
        uint8_t* lwe = c->words[hash2(sbuf, spos - 1)];
        uint16_t symrank_context = (uint16_t)(h1 | ((size_t)al << 8));
        uint8_t unlikely = lwe[0];
        const int detail = *items_done == g_detail_at;
        orc_symrank before;
        if (detail) before = c->symranks[symrank_context];
        uint16_t v = orc_symrank_decode(&c->symranks[symrank_context], symbol, unlikely);
        out->symbol = v; out->ctx = (uint32_t)h1; out->unlikely = unlikely;
        if (g_log && g_log_n < g_log_cap) {
            uint32_t* e = g_log + 4 * g_log_n++;
            e[0] = (uint32_t)spos; e[1] = symbol; e[2] = v; e[3] = symrank_context | ((uint32_t)unlikely << 16);
        }
        if (detail) {
            const uint16_t iu = before.index[unlikely];
            const uint8_t wl = spos < exp_end ? exp_win[spos] : 0;
            const uint16_t il = before.index[wl], iw = before.index[ORC_WORD_SYMBOL];
            out->unl_index = iu;
            out->lit_rank = il == iu ? 388u : (uint32_t)(il - (il > iu));
            out->word_rank = iw == iu ? 388u : (uint32_t)(iw - (iw > iu));
            out->tab_cnt = before.cnt; out->tab_sum = before.sum;
            const uint32_t raw = symbol == 388 ? iu : (uint32_t)(symbol + (symbol >= iu));
            out->tab_raw_index = raw;
            for (int dd = -4; dd <= 4; dd++) {
                long r2 = (long)raw + dd;
                out->tab_near[dd + 4] = r2 >= 0 && r2 < ORC_NUM_SYMBOLS ? before.value[r2] : 999u;
            }
            for (size_t nd = 0; nd < ORC_BUCKET_ITEMS; nd++) {
                size_t mp = cur->pos[nd];
                if (mp < 1 || mp >= spos) continue;
                uint32_t l = lcp_cap(exp_win + mp, exp_win + spos, ORC_MATCH_MAX_LEN);
                if (l > out->best_lcp) {
                    out->best_lcp = l; out->best_pos = (uint32_t)mp; out->best_ro = (uint32_t)nb_sub(cur->head, nd);
                    out->best_exp = cur->len_expected[nd]; out->best_min = cur->len_min[nd];
                }
            }
        }
        out->ring_count = (uint32_t)st->ring_items[h1];
        size_t ilen = 0;
        if (v == ORC_WORD_SYMBOL) {
            out->type = 0; out->word0 = lwe[0]; out->word1 = lwe[1];
            bucket_update(cur, spos, 0, 0);
            c->after_literal = 0;
            sbuf[spos] = lwe[0]; sbuf[spos + 1] = lwe[1];
            ilen = 2;
        } else if (v < 256) {
            out->type = 1; out->word0 = v;
            bucket_update(cur, spos, 0, 0);
            c->after_literal = 1;
            sbuf[spos] = (uint8_t)v;
            ilen = 1;
        } else {
            out->type = 2;
            unsigned roid = (unsigned)(v - 256) / LZ_LENID_SIZE;
            unsigned lenid = (unsigned)(v - 256) % LZ_LENID_SIZE;
            size_t reduced_offset = g_roid_dec_base[roid] + dec_raw(&dec, g_roid_dec_bits[roid]);
            out->reduced_offset = (uint32_t)reduced_offset;
            if (reduced_offset >= ORC_BUCKET_ITEMS) { out->kind = 2; out->cause = 3; return 1; }
            size_t node = nb_sub(cur->head, reduced_offset);
            size_t enc_len = lenid == LZ_LENID_SIZE - 1 ? dec_huffman_sym(&dec, d->h2) : lenid;
            size_t match_pos = cur->pos[node];
            size_t len_min = cur->len_min[node] > ORC_MATCH_MIN_LEN ? cur->len_min[node] : ORC_MATCH_MIN_LEN;
            size_t len_exp = cur->len_expected[node] > ORC_MATCH_MIN_LEN ? cur->len_expected[node] : ORC_MATCH_MIN_LEN;
            size_t match_len;
            if (enc_len + len_min > len_exp) match_len = enc_len + len_min;
            else if (enc_len > 0) match_len = enc_len + len_min - 1;
            else match_len = len_exp;
            out->node = (uint32_t)node; out->node_pos = (uint32_t)match_pos; out->node_len_min = cur->len_min[node];
            out->node_len_expected = cur->len_expected[node]; out->enc_len = (uint32_t)enc_len; out->match_len = (uint32_t)match_len;
            if (match_pos >= 1 && match_pos < spos) {
                out->true_lcp = lcp_cap(exp_win + match_pos, exp_win + spos, ORC_MATCH_MAX_LEN);
                out->src_ctx = (uint32_t)hash1(sbuf, match_pos - 1);
            }
            for (int dd = -16; detail && dd <= 16; dd++) {
                long r2 = (long)reduced_offset + dd;
                if (r2 < 0 || r2 >= ORC_BUCKET_ITEMS) continue;
                size_t nd = nb_sub(cur->head, (size_t)r2);
                size_t mp = cur->pos[nd];
                out->near_pos[dd + 16] = (uint32_t)mp;
                out->near_exp[dd + 16] = cur->len_expected[nd];
                out->near_min[dd + 16] = cur->len_min[nd];
                out->near_lcp[dd + 16] = mp >= 1 && mp < spos ? lcp_cap(exp_win + mp, exp_win + spos, ORC_MATCH_MAX_LEN) : 0;
            }
            if (match_pos >= spos || spos + match_len > ORC_LZ_BLOCK_SIZE + ORC_SENTINEL_LEN) { out->kind = 2; out->cause = 4; return 1; }
            bucket_update(cur, spos, reduced_offset, match_len);
            c->after_literal = 0;
            match_copy(sbuf, match_pos, spos, match_len);
            ilen = match_len;
        }
        st->ring_items[h1]++;
        /* compare with the expectation */
        for (size_t k = 0; k < ilen; k++) {
            if (spos + k >= exp_end || sbuf[spos + k] != exp_win[spos + k]) {
                out->kind = 1; out->first_bad = (uint32_t)k;
                out->want0 = spos < exp_end ? exp_win[spos] : 256; out->want1 = spos + 1 < exp_end ? exp_win[spos + 1] : 256;
                return 1;
            }
        }
        spos += ilen;
        if (v != ORC_WORD_SYMBOL) {
            size_t k = hash2(sbuf, spos - 3);
            c->words[k][0] = sbuf[spos - 2];
            c->words[k][1] = sbuf[spos - 1];
        }
        (*items_done)++;
        if (spos > ORC_LZ_BLOCK_SIZE) { out->kind = 2; out->cause = 5; return 1; }
    }
    *spos_end_out = spos < sbuf_len ? spos : sbuf_len;
    memset(out, 0, sizeof *out);
    return 0;
}

static int diag_decode_once(const uint8_t* src, size_t src_len, const uint8_t* expect, size_t expect_len, orc_diag* out);
/* returns out->kind */
int orc_diag_decode(const uint8_t* src, size_t src_len, const uint8_t* expect, size_t expect_len, orc_diag* out) {
    g_detail_at = ~0ull;
    int kind = diag_decode_once(src, src_len, expect, expect_len, out);
    if (kind == 1 || kind == 2) { /* again, with the failing item's surroundings recorded */
        uint32_t* keep = g_log;
        g_log = NULL;
        g_detail_at = out->item_index;
        kind = diag_decode_once(src, src_len, expect, expect_len, out);
        g_detail_at = ~0ull;
        g_log = keep;
    }
    return kind;
}
static int diag_decode_once(const uint8_t* src, size_t src_len, const uint8_t* expect, size_t expect_len, orc_diag* out) {
    diag_state st;
    memset(&st, 0, sizeof st);
    st.d = orc_lz_decoder_new();
    uint8_t* sbvec_buf = (uint8_t*)calloc((size_t)ORC_LZ_BLOCK_SIZE * 2 + ORC_SENTINEL_LEN * 2, 1);
    uint8_t* exp_buf = (uint8_t*)calloc((size_t)ORC_LZ_BLOCK_SIZE * 2 + ORC_SENTINEL_LEN * 2, 1);
    size_t tcap = (size_t)ORC_PREMATCH_LEN * 3;
    uint8_t* tbvec = (uint8_t*)calloc(tcap + 16, 1);
    memset(out, 0, sizeof *out);
    out->kind = 3;
    if (!st.d || !sbvec_buf || !exp_buf || !tbvec) goto done;
    uint8_t* sbvec = sbvec_buf + ORC_SENTINEL_LEN;
    uint8_t* exp_win = exp_buf + ORC_SENTINEL_LEN;
    size_t spos = ORC_PREMATCH_LEN, ip = 0;
    uint64_t base_off = 0, items = 0, produced = 0;
    uint32_t block = 0, chunk = 0;
    const size_t room = ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN;
    size_t n = expect_len < room ? expect_len : room;
    memcpy(exp_win + ORC_PREMATCH_LEN, expect, n);
    size_t exp_end = ORC_PREMATCH_LEN + n;
    for (;;) {
        size_t t = 0, factor = 1;
        for (;;) {
            if (ip >= src_len) goto done;
            uint8_t v = src[ip++];
            if (v < 128) { t += (size_t)v * factor; break; }
            t += (size_t)(v - 128) * factor;
            factor *= 128;
        }
        if (t == 0) break;
        if (t >= tcap || ip + t > src_len) goto done;
        memcpy(tbvec, src + ip, t);
        memset(tbvec + t, 0, 16);
        ip += t;
        size_t spos_end = 0;
        if (diag_chunk(&st, tbvec, t, sbvec, exp_win, exp_end, spos, &spos_end, out, base_off, &items)) {
            out->block = block; out->chunk = chunk;
            goto done;
        }
        if (spos_end < spos) { out->kind = 3; goto done; }
        produced += spos_end - spos;
        spos = spos_end;
        chunk++;
        if (spos >= ORC_LZ_BLOCK_SIZE) {
            memmove(sbvec, sbvec + ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN, ORC_PREMATCH_LEN);
            memmove(exp_win, exp_win + ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN, ORC_PREMATCH_LEN);
            orc_lz_decoder_forward(st.d, ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN);
            spos = ORC_PREMATCH_LEN;
            base_off += room;
            block++;
            size_t left = expect_len > base_off ? expect_len - base_off : 0;
            n = left < room ? left : room;
            memcpy(exp_win + ORC_PREMATCH_LEN, expect + base_off, n);
            exp_end = ORC_PREMATCH_LEN + n;
        }
    }
    out->kind = produced == expect_len ? 0 : 3;
    out->stream_off = produced;
done:
    free(sbvec_buf);
    free(exp_buf);
    free(tbvec);
    orc_lz_decoder_free(st.d);
    return out->kind;
}
