/*
 * fast_model.c -- sequential CPU model of the GPU "fast" parse mode (TEST INFRASTRUCTURE ONLY).
 *
 * The fast mode does not reproduce the reference's parse item for item; it produces *a* parse of the
 * reference's format: R rounds of "every position decides from a frozen snapshot of the
 * item starts" -> path extraction -> new snapshot, then a validity/repair pass.  This file restates
 * that algorithm with plain loops so that (a) the compression ratio of a design variant can be
 * measured without a GPU and (b) the GPU's parse can be compared with it.  Decision rules follow
 * LZEncoder::encode (/root/reference/src/lz.rs:131-235) and find_match / has_lazy_match
 * (src/matcher.rs:135-228) on the snapshot; the stream is then written by the oracle's plan-driven
 * encoder (orc_encode_plan_mem), which checks every item against the real state machine.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/orz_oracle.h"

#define P ORC_PREMATCH_LEN
#define NEWMAX (ORC_LZ_BLOCK_SIZE - ORC_PREMATCH_LEN)
#define KHASH ORC_BUCKET_HASH
#define NKEYS (256 * KHASH)

typedef struct {
    int rounds;       /* snapshot rounds */
    int init_all;     /* round-1 snapshot: 1 = every new position is an item start, 0 = none */
    int tile;         /* positions per ordinal tile (0 = exact ordinals for every position) */
    int slack;        /* ring slack: candidates need ro_est <= 4093 - slack */
    int verbose;
    int damp;         /* percent of positions that keep their previous decision in a round (0 = Jacobi) */
    int damp_tile;    /* granularity (positions) of that choice */
    int gs_tile;      /* > 0: pipelined Gauss-Seidel schedule with tiles of this many positions */
    int region;       /* > 0: regions of this many positions run their pipelines side by side (Jacobi across regions) */
    int passes;       /* outer passes over the regions */
} orzm_params;

typedef struct {
    uint64_t items, matches, words, repairs_src, repairs_ro, repairs_lenmin, repairs_word, repair_passes;
    uint64_t changed[16]; /* item starts that differ from the previous round, per round */
} orzm_stats;

static inline int is_alnum(uint8_t c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'); }
static inline uint32_t h1(const uint8_t* b, size_t pos) { return (uint32_t)(b[pos] & 0x7f) | ((uint32_t)is_alnum(b[pos - 1]) << 7); }
static inline uint32_t h2(const uint8_t* b, size_t pos) { return (uint32_t)(b[pos] & 0x7f) | (h1(b, pos - 1) << 7); }
static inline uint32_t bkey(const uint8_t* b, size_t x) { return h1(b, x - 1) * KHASH + orc_hash_entry(b + x); }
static inline uint32_t lcp240(const uint8_t* b, size_t p1, size_t p2) {
    uint32_t l = 0;
    while (l < 240 && b[p1 + l] == b[p2 + l]) l++;
    return l;
}

/* model state of one stream */
typedef struct {
    uint8_t* winbuf;  /* sentinel + window */
    uint8_t* win;
    uint8_t* S;       /* item start flags per window position (final for history) */
    uint8_t* ML;      /* match length of the item starting there (0 = literal / word) */
    uint8_t* TY;
    uint32_t* SRC;
    uint32_t* ORD;    /* stream-cumulative ordinal within its context */
    uint8_t* LENMIN;  /* carried len_min of history sources */
    uint8_t* W0;
    uint32_t ctxcount[256];
    uint8_t wsnap[32768][2];
    int first;
} mstream;

/* per-round outputs for every new position */
typedef struct {
    uint8_t *bl, *lz1, *lz2, *lwm, *ty, *nl;
    uint32_t* bsrc;
    uint8_t *Sp, *Ep;     /* snapshot flags (window positions) */
    uint32_t* ordp;       /* snapshot ordinals of members */
    uint32_t* ordest;     /* ordinal estimate for every position (tile table or exact) */
} mround;

static void* xcalloc(size_t n, size_t s) {
    void* p = calloc(n ? n : 1, s);
    if (!p) { fprintf(stderr, "fast_model: out of memory\n"); abort(); }
    return p;
}

typedef struct { uint32_t key, pos; } kp;

static void parse_block(mstream* st, uint32_t n, const orc_lzcfg* cfg, const orzm_params* pr, orzm_stats* stats,
                        orc_plan_item* plan, size_t* nplan, uint64_t stream_off) {
    const uint8_t* win = st->win;
    const uint32_t len = P + n;
    const uint32_t depth = (uint32_t)cfg->match_depth, lazy1 = (uint32_t)cfg->lazy_match_depth1, lazy2 = (uint32_t)cfg->lazy_match_depth2;
    /* ---- slots: history item starts + every new position, ordered by (key, pos) */
    uint32_t nhist = 0;
    for (uint32_t x = 1; x < P; x++) nhist += st->S[x];
    const uint32_t nent = nhist + n;
    uint32_t* runstart = (uint32_t*)xcalloc(NKEYS + 1, 4);
    uint32_t* spos = (uint32_t*)xcalloc(nent, 4);
    uint32_t* idx = (uint32_t*)xcalloc(len, 4);
    uint32_t* keyof = (uint32_t*)xcalloc(len, 4);
    for (uint32_t x = 1; x < len; x++)
        if (x >= P || st->S[x]) { keyof[x] = bkey(win, x); runstart[keyof[x] + 1]++; }
    for (uint32_t k = 0; k < NKEYS; k++) runstart[k + 1] += runstart[k];
    {
        uint32_t* fill = (uint32_t*)xcalloc(NKEYS, 4);
        for (uint32_t x = 1; x < len; x++)
            if (x >= P || st->S[x]) { uint32_t j = runstart[keyof[x]] + fill[keyof[x]]++; spos[j] = x; idx[x] = j; }
        free(fill);
    }
    /* word predictor lists: positions u in [P-2, len) ordered by (hash2(u-1), u) */
    uint32_t* krun = (uint32_t*)xcalloc(32769, 4);
    uint32_t nk = n + 2;
    uint32_t* kpos = (uint32_t*)xcalloc(nk, 4);
    uint32_t* kidx = (uint32_t*)xcalloc(len, 4);
    for (uint32_t u = P - 2; u < len; u++) krun[h2(win, u - 1) + 1]++;
    for (uint32_t k = 0; k < 32768; k++) krun[k + 1] += krun[k];
    {
        uint32_t* fill = (uint32_t*)xcalloc(32768, 4);
        for (uint32_t u = P - 2; u < len; u++) { uint32_t k = h2(win, u - 1); uint32_t j = krun[k] + fill[k]++; kpos[j] = u; kidx[u] = j; }
        free(fill);
    }

    mround r;
    r.bl = (uint8_t*)xcalloc(len + 8, 1); r.lz1 = (uint8_t*)xcalloc(len + 8, 1); r.lz2 = (uint8_t*)xcalloc(len + 8, 1);
    r.lwm = (uint8_t*)xcalloc(len + 8, 1); r.ty = (uint8_t*)xcalloc(len + 8, 1); r.nl = (uint8_t*)xcalloc(len + 8, 1);
    r.bsrc = (uint32_t*)xcalloc(len + 8, 4);
    r.Sp = (uint8_t*)xcalloc(len + 8, 1); r.Ep = (uint8_t*)xcalloc(len + 8, 1);
    r.ordp = (uint32_t*)xcalloc(len + 8, 4); r.ordest = (uint32_t*)xcalloc(len + 8, 4);
    uint8_t* Snew = (uint8_t*)xcalloc(len + 8, 1);
    uint8_t* Enew = (uint8_t*)xcalloc(len + 8, 1);

    memcpy(r.Sp, st->S, P);
    for (uint32_t x = P; x < len; x++) r.Sp[x] = pr->init_all ? 1 : 0;
    /* Ep: u = e-2 for ends e of non-WORD items; round 1: nothing known inside the block */
    const uint32_t tile = pr->tile > 0 ? (uint32_t)pr->tile : 1;

    if (pr->gs_tile > 0) {
        /* Pipelined Gauss-Seidel schedule: the block is cut into tiles; at step s tile t runs its round s - t
         * (1..R), all active tiles at once, each seeing the other tiles as they are at the start of the step.
         * A tile's LAST round therefore sees every earlier tile in its final state. */
        const uint32_t T = (uint32_t)pr->gs_tile, R = (uint32_t)pr->rounds;
        const uint32_t ntile = (n + T - 1) / T;
        uint32_t (*cnt_at)[256] = (uint32_t(*)[256])xcalloc(ntile + 2, 256 * 4);
        uint32_t* entry = (uint32_t*)xcalloc(ntile + 2, 4);
        memcpy(cnt_at[0], st->ctxcount, 256 * 4);
        entry[0] = P;
        for (uint32_t x = 1; x < P; x++) if (st->S[x]) r.ordp[x] = st->ORD[x];
        const uint32_t tpr = pr->region > 0 ? ((uint32_t)pr->region + T - 1) / T : ntile; /* tiles per region */
        const int passes = pr->region > 0 ? pr->passes : 1;
        uint8_t* SpPrev = (uint8_t*)xcalloc(len + 8, 1);
        uint8_t* EpPrev = (uint8_t*)xcalloc(len + 8, 1);
        uint8_t* SpNext = (uint8_t*)xcalloc(len + 8, 1);
        uint8_t* EpNext = (uint8_t*)xcalloc(len + 8, 1);
        memcpy(SpPrev, r.Sp, len); memcpy(EpPrev, r.Ep, len);
        for (int pass = 1; pass <= passes; pass++)
        for (uint32_t rg = 0; rg * tpr < ntile; rg++) {
        const uint32_t tr0 = rg * tpr, tr1 = (rg + 1) * tpr < ntile ? (rg + 1) * tpr : ntile;
        if (pr->region > 0) { /* earlier regions as the previous pass left them; this region starts from its previous result */
            memcpy(r.Sp, SpPrev, len); memcpy(r.Ep, EpPrev, len);
            uint32_t cnt[256];
            memcpy(cnt, st->ctxcount, sizeof cnt);
            uint32_t x = P;
            for (; x < P + tr0 * T; x++) if (r.Sp[x]) { r.ordp[x] = cnt[h1(win, x - 1)]++; }
            memcpy(cnt_at[tr0], cnt, sizeof cnt);
            while (x < len && !r.Sp[x]) x++;
            entry[tr0] = x;
        }
        for (uint32_t step = tr0 + 1; step <= tr1 + R - 1; step++) {
            const uint32_t t_lo = step > R + tr0 ? step - R : tr0, t_hi = step - 1 < tr1 - 1 ? step - 1 : tr1 - 1;
            const uint32_t a = P + t_lo * T, b = (P + (t_hi + 1) * T < len) ? P + (t_hi + 1) * T : len;
            const uint32_t b2 = b + 2 < len ? b + 2 : len;
            {   /* ordinals over the active range (flags as they are now) */
                uint32_t cnt[256];
                memcpy(cnt, cnt_at[t_lo], sizeof cnt);
                for (uint32_t x = a; x < b2; x++) {
                    if (x < b && (x - P) % T == 0) memcpy(cnt_at[(x - P) / T], cnt, sizeof cnt);
                    const uint32_t c = h1(win, x - 1);
                    r.ordest[x] = cnt[c];
                    if (r.Sp[x]) { r.ordp[x] = cnt[c]; cnt[c]++; }
                }
            }
            for (uint32_t p = a; p < b2; p++) {
                const uint32_t key = keyof[p], lo = runstart[key];
                uint32_t best = 0, bsrc = 0, m1 = 0, m2 = 0, seen = 0, scanned = 0;
                const uint32_t op = r.ordest[p];
                for (uint32_t j = idx[p]; j > lo && seen < depth;) {
                    j--;
                    const uint32_t q = spos[j];
                    if (!r.Sp[q]) { if (++scanned > 200000) break; continue; }
                    if (op - 1 - r.ordp[q] > (uint32_t)(4093 - pr->slack)) break;
                    const uint32_t l = lcp240(win, q, p);
                    if (l > best) { best = l; bsrc = q; }
                    if (seen < lazy1 && l > m1) m1 = l;
                    if (seen < lazy2 && l > m2) m2 = l;
                    seen++;
                    if (l == 240) break;
                }
                r.bl[p] = (uint8_t)best; r.bsrc[p] = bsrc; r.lz1[p] = (uint8_t)m1; r.lz2[p] = (uint8_t)m2;
                const uint32_t k = h2(win, p - 1);
                uint8_t w0 = st->wsnap[k][0], w1 = st->wsnap[k][1];
                scanned = 0;
                for (uint32_t j = kidx[p]; j > krun[k];) {
                    j--;
                    const uint32_t u = kpos[j];
                    if (u + 2 > p) continue;
                    if (r.Ep[u]) { w0 = win[u]; w1 = win[u + 1]; break; }
                    if (++scanned > 200000) break;
                }
                r.lwm[p] = (win[p] == w0 && win[p + 1] == w1);
            }
            for (uint32_t p = a; p < b; p++) {
                uint32_t L = r.bl[p];
                if (L < 4 || p + L >= len) L = 0;
                int lazy = 0;
                if (L > 0 && L < 120) {
                    const uint32_t ro = r.ordest[p] - 1 - r.ordp[r.bsrc[p]];
                    const uint32_t l1 = L + 1 + (ro < 510), l2 = l1 - r.lwm[p];
                    if (r.lz1[p + 1] >= l1) lazy = 1;
                    else if (r.lz2[p + 2] >= l2) lazy = 2;
                }
                if (L > 0 && lazy == 0) { r.ty[p] = ORC_PLAN_MATCH; r.nl[p] = (uint8_t)L; }
                else if (p + 1 < len && lazy != 1 && r.lwm[p]) { r.ty[p] = ORC_PLAN_WORD; r.nl[p] = 2; }
                else { r.ty[p] = ORC_PLAN_LITERAL; r.nl[p] = 1; }
            }
            /* path through the active range from the (final) entry of its first tile */
            {
                uint32_t pth = entry[t_lo];
                const uint32_t ce = b + 240 < len ? b + 240 : len;
                if (pth < b) {
                    for (uint32_t x = pth; x < b; x++) r.Sp[x] = 0;
                    for (uint32_t x = pth - 1; x < ce; x++) r.Ep[x] = 0;
                }
                uint32_t tcur = t_lo;
                while (pth < b) {
                    while (tcur + 1 <= t_hi + 1 && pth >= P + (tcur + 1) * T) { tcur++; entry[tcur] = pth; }
                    r.Sp[pth] = 1;
                    const uint32_t e = pth + r.nl[pth];
                    if (r.ty[pth] != ORC_PLAN_WORD) r.Ep[e - 2] = 1;
                    pth = e;
                }
                while (tcur + 1 <= t_hi + 1) { tcur++; entry[tcur] = pth > P + tcur * T ? pth : P + tcur * T; if (entry[tcur] < pth) entry[tcur] = pth; }
            }
        }
        if (pr->region > 0) {
            const uint32_t a = P + tr0 * T, b = P + tr1 * T < len ? P + tr1 * T : len;
            memcpy(SpNext + a, r.Sp + a, b - a);
            memcpy(EpNext + a - 2, r.Ep + a - 2, b - a);
            if (tr1 == ntile) { memcpy(SpPrev + P, SpNext + P, n); memcpy(EpPrev + P - 2, EpNext + P - 2, n + 2); }
        }
        }
        if (pr->region > 0) {
            /* the final item set must be ONE path: walk the last pass's decisions from the block start */
            memset(r.Sp + P, 0, n + 8); memset(r.Ep, 0, len + 8);
            for (uint32_t pth = P; pth < len;) { r.Sp[pth] = 1; const uint32_t e = pth + r.nl[pth]; if (r.ty[pth] != ORC_PLAN_WORD) r.Ep[e - 2] = 1; pth = e; }
        }
        free(SpPrev); free(EpPrev); free(SpNext); free(EpNext);
        free(cnt_at); free(entry);
    } else
    for (int round = 1; round <= pr->rounds; round++) {
        /* ---- snapshot ordinals */
        {
            uint32_t cnt[256];
            memcpy(cnt, st->ctxcount, sizeof cnt);
            for (uint32_t x = 1; x < P; x++) if (st->S[x]) r.ordp[x] = st->ORD[x];
            if (pr->tile <= 0) {
                for (uint32_t x = P; x < len; x++) {
                    uint32_t c = h1(win, x - 1);
                    r.ordest[x] = cnt[c];
                    if (r.Sp[x]) { r.ordp[x] = cnt[c]; cnt[c]++; }
                }
            } else { /* conservative estimate: members of the context below the END of the position's tile */
                for (uint32_t t0 = P; t0 < len; t0 += tile) {
                    uint32_t t1 = t0 + tile < len ? t0 + tile : len;
                    for (uint32_t x = t0; x < t1; x++) {
                        uint32_t c = h1(win, x - 1);
                        if (r.Sp[x]) { r.ordp[x] = cnt[c]; cnt[c]++; }
                    }
                    for (uint32_t x = t0; x < t1; x++) r.ordest[x] = cnt[h1(win, x - 1)];
                }
            }
        }
        /* ---- per position: candidates of its (ctx, hash) run among the snapshot's item starts.
         * Walk the slots in (key, pos) order keeping the run's members seen so far (= the GPU's slot-order kernel). */
        {
            uint32_t* mem = (uint32_t*)xcalloc(nent, 4);
            uint32_t nm = 0, run_lo = 0, curkey = 0xffffffffu;
            for (uint32_t j = 0; j < nent; j++) {
                const uint32_t p = spos[j];
                if (keyof[p] != curkey) { curkey = keyof[p]; run_lo = nm; }
                if (p >= P) {
                    uint32_t best = 0, bsrc = 0, m1 = 0, m2 = 0, seen = 0;
                    const uint32_t op = r.ordest[p];
                    for (uint32_t i = nm; i > run_lo && seen < depth;) {
                        i--;
                        const uint32_t q = mem[i];
                        if (op - 1 - r.ordp[q] > (uint32_t)(4093 - pr->slack)) break; /* left the ring */
                        const uint32_t l = lcp240(win, q, p);
                        if (l > best) { best = l; bsrc = q; }
                        if (seen < lazy1 && l > m1) m1 = l;
                        if (seen < lazy2 && l > m2) m2 = l;
                        seen++;
                        if (l == 240) break;
                    }
                    r.bl[p] = (uint8_t)best; r.bsrc[p] = bsrc; r.lz1[p] = (uint8_t)m1; r.lz2[p] = (uint8_t)m2;
                }
                if (r.Sp[p]) mem[nm++] = p;
            }
            free(mem);
        }
        /* ---- word predictor answer per position: newest u <= p-2 with the same hash2 key whose update stuck */
        {
            uint32_t* lastu = (uint32_t*)xcalloc(32768, 4);
            for (uint32_t p = P; p < len; p++) {
                const uint32_t u = p - 2;
                if (r.Ep[u]) lastu[h2(win, u - 1)] = u;
                const uint32_t k = h2(win, p - 1);
                uint8_t w0 = st->wsnap[k][0], w1 = st->wsnap[k][1];
                if (lastu[k]) { w0 = win[lastu[k]]; w1 = win[lastu[k] + 1]; }
                r.lwm[p] = (win[p] == w0 && win[p + 1] == w1);
            }
            free(lastu);
        }
        /* ---- decisions (src/lz.rs:139-234 on snapshot quantities) */
        for (uint32_t p = P; p < len; p++) {
            if (round > 1 && pr->damp > 0) { /* asynchronous update: only part of the positions re-decide in a round */
                uint32_t hsh = ((p / (uint32_t)pr->damp_tile) * 2654435761u) ^ ((uint32_t)round * 40503u);
                hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
                if ((hsh % 100) < (uint32_t)pr->damp) continue;
            }
            uint32_t L = r.bl[p];
            if (L < 4 || p + L >= len) L = 0;
            int lazy = 0;
            if (L > 0 && L < 120) {
                const uint32_t ro = r.ordest[p] - 1 - r.ordp[r.bsrc[p]];
                const uint32_t l1 = L + 1 + (ro < 510), l2 = l1 - r.lwm[p];
                if (r.lz1[p + 1] >= l1) lazy = 1;
                else if (r.lz2[p + 2] >= l2) lazy = 2;
            }
            if (L > 0 && lazy == 0) { r.ty[p] = ORC_PLAN_MATCH; r.nl[p] = (uint8_t)L; }
            else if (p + 1 < len && lazy != 1 && r.lwm[p]) { r.ty[p] = ORC_PLAN_WORD; r.nl[p] = 2; }
            else { r.ty[p] = ORC_PLAN_LITERAL; r.nl[p] = 1; }
        }
        /* ---- path */
        memset(Snew + P, 0, n + 8);
        memset(Enew, 0, len + 8);
        for (uint32_t p = P; p < len;) {
            Snew[p] = 1;
            const uint32_t e = p + r.nl[p];
            if (r.ty[p] != ORC_PLAN_WORD) Enew[e - 2] = 1;
            p = e;
        }
        uint64_t chg = 0, chg2 = 0;
        static uint8_t* Sprev2 = NULL;
        if (!Sprev2) Sprev2 = (uint8_t*)xcalloc((size_t)ORC_LZ_BLOCK_SIZE + 16, 1);
        for (uint32_t x = P; x < len; x++) { chg += Snew[x] != r.Sp[x]; chg2 += Snew[x] != Sprev2[x]; }
        memcpy(Sprev2 + P, r.Sp + P, n);
        if (pr->verbose) fprintf(stderr, "  round %d: vs two rounds ago %llu\n", round, (unsigned long long)chg2);
        if (round < 16) stats->changed[round] += chg;
        if (pr->verbose) fprintf(stderr, "  round %d: item-start flags changed %llu\n", round, (unsigned long long)chg);
        memcpy(r.Sp + P, Snew + P, n);
        memcpy(r.Ep, Enew, len);
    }

    /* ---- the path of the last round is the item set; sources are assigned afterwards, with the item
     * boundaries frozen: every match item takes the NEWEST item start of its run (within `depth` members
     * and the ring) whose common prefix covers its length.  With that rule the lengths referring to one
     * source ascend with position, so len >= len_min holds by construction (src/matcher.rs:32-50).  An
     * item without such a source is cut to the longest prefix some member offers (or becomes a literal)
     * and the rest of its span is re-parsed from the last round's decisions, truncated at the old end:
     * item starts are only ever added.  WORD items are checked against the exact predictor state and
     * turned into two literals when it disagrees.  Repeat until nothing changes. */
    {
        uint8_t* Sf = st->S;
        uint8_t* TYf = st->TY;
        uint8_t* Lf = (uint8_t*)xcalloc(len + 8, 1);
        uint32_t* SRCf = st->SRC;
        uint32_t* mem = (uint32_t*)xcalloc(nent, 4);
        uint32_t* lastu = (uint32_t*)xcalloc(32768, 4);
        for (uint32_t x = P; x < len; x++) { Sf[x] = r.Sp[x]; TYf[x] = r.ty[x]; Lf[x] = r.nl[x]; }
        for (int pass = 0;; pass++) {
            uint64_t changes = 0;
            uint32_t cnt[256];
            memcpy(cnt, st->ctxcount, sizeof cnt);
            for (uint32_t x = P; x < len; x++) if (Sf[x]) st->ORD[x] = cnt[h1(win, x - 1)]++;
            /* source assignment in slot order */
            uint32_t nm = 0, run_lo = 0, curkey = 0xffffffffu;
            uint32_t* cut = (uint32_t*)xcalloc(n + 8, 4); /* cut[p-P] = old end of an item that was shortened in this pass */
            for (uint32_t j = 0; j < nent; j++) {
                const uint32_t p = spos[j];
                if (keyof[p] != curkey) { curkey = keyof[p]; run_lo = nm; }
                if (p >= P && Sf[p] && TYf[p] == ORC_PLAN_MATCH) {
                    const uint32_t L = Lf[p], op = st->ORD[p];
                    uint32_t best = 0, bsrc = 0, seen = 0, found = 0;
                    for (uint32_t i = nm; i > run_lo && seen < depth;) {
                        i--;
                        const uint32_t q = mem[i];
                        if (op - 1 - st->ORD[q] > 4093) break;
                        const uint32_t l = lcp240(win, q, p);
                        if (l >= L) { found = q; break; }
                        if (l > best) { best = l; bsrc = q; }
                        seen++;
                    }
                    if (found) SRCf[p] = found;
                    else {
                        changes++;
                        if (pass == 0) stats->repairs_src++; else stats->repairs_ro++;
                        cut[p - P] = p + L;
                        if (best >= 4) { Lf[p] = (uint8_t)best; SRCf[p] = bsrc; }
                        else { TYf[p] = ORC_PLAN_LITERAL; Lf[p] = 1; }
                    }
                }
                if (p < P ? st->S[p] : Sf[p]) mem[nm++] = p;
            }
            /* re-parse the uncovered rest of every shortened item from the last round's decisions */
            for (uint32_t p = P; p < len; p++) {
                if (!cut[p - P]) continue;
                const uint32_t end = cut[p - P];
                uint32_t x = p + Lf[p];
                while (x < end) {
                    uint8_t ty = r.ty[x];
                    uint32_t L = r.nl[x];
                    if (x + L > end) { /* truncate at the old end */
                        if (ty == ORC_PLAN_MATCH && end - x >= 4) L = end - x;
                        else { ty = ORC_PLAN_LITERAL; L = 1; }
                    }
                    Sf[x] = 1; TYf[x] = ty; Lf[x] = (uint8_t)L;
                    x += L;
                }
            }
            free(cut);
            /* WORD items against the exact predictor (sequential replay of words[], src/lz.rs:203,233) */
            memset(lastu, 0, 32768 * 4);
            for (uint32_t p = P; p < len;) {
                /* updates of items that ended at or before p are in lastu */
                const uint32_t k = h2(win, p - 1);
                uint8_t w0 = st->wsnap[k][0], w1 = st->wsnap[k][1];
                if (lastu[k]) { w0 = win[lastu[k]]; w1 = win[lastu[k] + 1]; }
                if (TYf[p] == ORC_PLAN_WORD && !(win[p] == w0 && win[p + 1] == w1)) {
                    TYf[p] = ORC_PLAN_LITERAL; Lf[p] = 1;
                    Sf[p + 1] = 1; TYf[p + 1] = ORC_PLAN_LITERAL; Lf[p + 1] = 1;
                    changes++; stats->repairs_word++;
                }
                st->W0[p] = w0;
                const uint32_t e = p + Lf[p];
                if (TYf[p] != ORC_PLAN_WORD) lastu[h2(win, e - 3)] = e - 2;
                p = e;
            }
            stats->repair_passes++;
            if (pr->verbose) fprintf(stderr, "  repair pass %d: %llu changes\n", pass, (unsigned long long)changes);
            if (!changes) break;
        }
        /* ---- commit: items, carried state */
        {
            uint32_t cnt[256];
            memcpy(cnt, st->ctxcount, sizeof cnt);
            uint32_t lastty = ORC_PLAN_LITERAL;
            for (uint32_t p = P; p < len;) {
                const uint32_t c = h1(win, p - 1);
                const uint8_t ty = TYf[p];
                const uint32_t L = Lf[p];
                st->ORD[p] = cnt[c]++;
                st->ML[p] = ty == ORC_PLAN_MATCH ? (uint8_t)L : 0;
                orc_plan_item* it = &plan[(*nplan)++];
                it->pos = stream_off + (p - P); it->type = ty; it->len = (uint8_t)(ty == ORC_PLAN_MATCH ? L : 0);
                it->src = ty == ORC_PLAN_MATCH ? stream_off + SRCf[p] - P : 0;
                stats->items++; stats->matches += ty == ORC_PLAN_MATCH; stats->words += ty == ORC_PLAN_WORD;
                p += L;
                if (ty != ORC_PLAN_WORD) { const uint32_t kk = h2(win, p - 3); st->wsnap[kk][0] = win[p - 2]; st->wsnap[kk][1] = win[p - 1]; }
                lastty = ty;
            }
            (void)lastty;
            memcpy(st->ctxcount, cnt, sizeof cnt);
        }
        free(Lf); free(mem); free(lastu);
    }
    free(runstart); free(spos); free(idx); free(keyof); free(krun); free(kpos); free(kidx);
    free(r.bl); free(r.lz1); free(r.lz2); free(r.lwm); free(r.ty); free(r.nl); free(r.bsrc); free(r.Sp); free(r.Ep); free(r.ordp); free(r.ordest);
    free(Snew); free(Enew);
}

/* whole stream: returns the parse in stream offsets (caller frees *plan_out with free()) */
int orzm_parse(const uint8_t* src, size_t src_len, const orc_lzcfg* cfg, const orzm_params* pr, orc_plan_item** plan_out,
               size_t* nplan_out, orzm_stats* stats) {
    mstream st;
    memset(&st, 0, sizeof st);
    const size_t W = (size_t)ORC_LZ_BLOCK_SIZE + 1;
    st.winbuf = (uint8_t*)xcalloc(W + 2 * ORC_SENTINEL_LEN + 64, 1);
    st.win = st.winbuf + ORC_SENTINEL_LEN;
    st.S = (uint8_t*)xcalloc(W + 8, 1); st.ML = (uint8_t*)xcalloc(W + 8, 1); st.TY = (uint8_t*)xcalloc(W + 8, 1);
    st.SRC = (uint32_t*)xcalloc(W + 8, 4); st.ORD = (uint32_t*)xcalloc(W + 8, 4); st.LENMIN = (uint8_t*)xcalloc(W + 8, 1); st.W0 = (uint8_t*)xcalloc(W + 8, 1);
    orc_plan_item* plan = (orc_plan_item*)xcalloc(src_len + 1, sizeof(orc_plan_item));
    size_t nplan = 0;
    memset(stats, 0, sizeof *stats);
    size_t off = 0;
    while (off < src_len) {
        uint32_t n = (uint32_t)(src_len - off < NEWMAX ? src_len - off : NEWMAX);
        memcpy(st.win + P, src + off, n);
        parse_block(&st, n, cfg, pr, stats, plan, &nplan, off);
        off += n;
        if (off < src_len) { /* slide, src/lib.rs:83-84 */
            memmove(st.win, st.win + NEWMAX, P);
            memmove(st.S, st.S + NEWMAX, P); st.S[0] = 0;
            memmove(st.ML, st.ML + NEWMAX, P);
            memmove(st.ORD, st.ORD + NEWMAX, (size_t)P * 4);
            memmove(st.LENMIN, st.LENMIN + NEWMAX, P);
            memset(st.S + P, 0, NEWMAX + 1);
        }
    }
    *plan_out = plan;
    *nplan_out = nplan;
    free(st.winbuf); free(st.S); free(st.ML); free(st.TY); free(st.SRC); free(st.ORD); free(st.LENMIN); free(st.W0);
    return 0;
}

#ifdef ORZM_MAIN
#include <time.h>
static uint8_t* slurp(const char* path, size_t* n, size_t limit) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (limit && (size_t)sz > limit) sz = (long)limit;
    uint8_t* p = (uint8_t*)malloc(sz > 0 ? (size_t)sz : 1);
    *n = fread(p, 1, (size_t)sz, f);
    fclose(f);
    return p;
}
int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: fast_model FILE LEVEL [rounds] [init_all] [tile] [slack] [limit_bytes]\n"); return 2; }
    size_t limit = argc > 7 ? (size_t)atoll(argv[7]) : 0;
    size_t n;
    uint8_t* data = slurp(argv[1], &n, limit);
    if (!data) return 1;
    int level = atoi(argv[2]);
    orc_lzcfg cfg = {level == 0 ? 5u : level == 1 ? 15u : 45u, level == 0 ? 3u : level == 1 ? 9u : 27u, level == 0 ? 2u : level == 1 ? 6u : 18u};
    orzm_params pr = {argc > 3 ? atoi(argv[3]) : 3, argc > 4 ? atoi(argv[4]) : 1, argc > 5 ? atoi(argv[5]) : 0, argc > 6 ? atoi(argv[6]) : 0, 1, argc > 8 ? atoi(argv[8]) : 0, argc > 9 ? atoi(argv[9]) : 1, argc > 10 ? atoi(argv[10]) : 0, argc > 11 ? atoi(argv[11]) : 0, argc > 12 ? atoi(argv[12]) : 1};
    orzm_stats st;
    orc_plan_item* plan; size_t nplan;
    orzm_parse(data, n, &cfg, &pr, &plan, &nplan, &st);
    uint8_t* out; size_t outn; orc_plan_error err = {0, 0};
    if (orc_encode_plan_mem(data, n, plan, nplan, &out, &outn, NULL, &err)) { fprintf(stderr, "plan rejected at %zu code %d\n", err.pos, err.code); return 1; }
    uint8_t* back; size_t backn, used;
    int rc = orc_decode_mem(out, outn, &back, &backn, &used);
    int same = rc == 0 && backn == n && memcmp(back, data, n) == 0;
    uint8_t* ref; size_t refn;
    orc_encode_mem(data, n, &cfg, &ref, &refn, NULL);
    printf("in %zu  fast %zu  oracle %zu  delta %+.3f%%  roundtrip %s  items %llu matches %llu words %llu  unsourced first pass %llu later %llu  word repairs %llu  passes %llu\n",
           n, outn, refn, 100.0 * ((double)outn - (double)refn) / (double)refn, same ? "ok" : "FAIL",
           (unsigned long long)st.items, (unsigned long long)st.matches, (unsigned long long)st.words,
           (unsigned long long)st.repairs_src, (unsigned long long)st.repairs_ro, (unsigned long long)st.repairs_word, (unsigned long long)st.repair_passes);
    return same ? 0 : 1;
}
#endif
