/* bpe_fast_oracle.c -- an EXACT restatement of minbpe's training loop that does not rescan the stream.
 *
 * TEST INFRASTRUCTURE ONLY (like bpe_oracle.c): never linked into or called from minbpe_amd/.
 *
 * Why it exists: the plain restatement (bpe_oracle.c: get_stats -> max -> merge, base.py:13-41, basic.py:31-42,
 * regex.py:49-63) costs one pass over the stream per merge -- 3.9 hours of CPU for the first 2048 merges of the 1 GB
 * one-stream input, so the other 29,696 merges of BASELINE.json's target configuration had no oracle answer.  This file
 * computes the same merges in minutes.  It is NOT the reference's algorithm restated line for line; it is pinned to it
 * instead: equal to orc_train on every small, tie-heavy and chunked case of tests/test_fast_oracle.py, on the committed
 * full-length digests the plain oracle made (12 MB one stream / 16 MB and 8 MB chunked, all 31,744 merges each: their
 * tails are hundreds of tied pairs), on all 3840 merges of the 100 MB configuration, and on the first 2048 merges of the
 * 1 GB inputs (tests/golden/big_golden.json).
 *
 * What it relies on, each a consequence of the reference's loop:
 *  (1) merge(ids, (a, b), Z) (base.py:25-41) changes, per site  L a b R -> L Z R,  exactly the pairs (L,a), (a,b),
 *      (b,R) (one occurrence less each) and (L,Z), (Z,R) (one more each); every other adjacency stands.  Sites are
 *      taken left to right, so a == b pairs up inside a run as the reference does (a site's second token is gone when the
 *      sweep reaches it).
 *  (2) get_stats (base.py:13-22) is therefore known after a merge without a recount: counts are kept in a dense table.
 *  (3) A new adjacency always involves the NEW token, so all occurrences a pair (x, y) will ever have are created during
 *      the one merge that creates max(x, y) (or stand in the input): its occurrence list is written once, in stream
 *      order, and only ever loses members.  An entry of the list is live iff the tokens at that position still read x, y.
 *  (4) max(stats, key=stats.get) (basic.py:35) takes the highest count and, among equal counts, the key inserted first
 *      into a dict that get_stats fills in stream order: the pair whose FIRST LIVE occurrence comes first (SURVEY F3, F5:
 *      chunk order is stream order).  A lazy max-heap keyed by count yields every pair at the maximum; the winner is the
 *      one whose list's first live entry has the lowest position.
 * Pairs never span chunks (regex.py:44, 60): a token that starts a chunk carries a flag, and an adjacency whose right
 * token is flagged is not a pair.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FLAG 0x80000000u
#define IDM 0x7FFFFFFFu
#define DEAD 0x7FFFFFFFu
#define NONE 0xFFFFFFFFu

typedef struct {
    uint32_t count; /* the pair's count when this entry was pushed (the table has the current one) */
    uint32_t len;   /* occurrences written when the pair was created */
    uint32_t head;  /* entries before this one are known dead */
    uint32_t a, b;
    uint32_t blk;   /* arena block of its occurrence list ... */
    uint64_t off;   /* ... and the index of its first entry there */
} hent;

typedef struct {
    hent *h;
    uint64_t n, cap;
} heap;

static int heap_push(heap *H, hent e) {
    if (H->n == H->cap) {
        uint64_t nc = H->cap ? H->cap * 2 : 1u << 16;
        hent *nh = (hent *)realloc(H->h, nc * sizeof(hent));
        if (!nh) return -1;
        H->h = nh;
        H->cap = nc;
    }
    uint64_t i = H->n++;
    while (i > 0) {
        uint64_t p = (i - 1) >> 1;
        if (H->h[p].count >= e.count) break;
        H->h[i] = H->h[p];
        i = p;
    }
    H->h[i] = e;
    return 0;
}
static hent heap_pop(heap *H) {
    hent top = H->h[0];
    hent e = H->h[--H->n];
    uint64_t i = 0;
    for (;;) {
        uint64_t l = 2 * i + 1, r = l + 1, m;
        if (l >= H->n) break;
        m = (r < H->n && H->h[r].count > H->h[l].count) ? r : l;
        if (H->h[m].count <= e.count) break;
        H->h[i] = H->h[m];
        i = m;
    }
    if (H->n) H->h[i] = e;
    return top;
}

#define ABLK (1ull << 27) /* entries per arena block (512 MiB) */
typedef struct {
    uint32_t **blk;
    uint64_t *cap, *used;
    uint32_t n, nalloc;
} arena;
static int arena_new_block(arena *A, uint64_t cap) {
    if (A->n == A->nalloc) {
        uint32_t na = A->nalloc ? A->nalloc * 2 : 64;
        A->blk = (uint32_t **)realloc(A->blk, na * sizeof(uint32_t *));
        A->cap = (uint64_t *)realloc(A->cap, na * sizeof(uint64_t));
        A->used = (uint64_t *)realloc(A->used, na * sizeof(uint64_t));
        if (!A->blk || !A->cap || !A->used) return -1;
        A->nalloc = na;
    }
    A->blk[A->n] = (uint32_t *)malloc((cap ? cap : 1) * sizeof(uint32_t));
    if (!A->blk[A->n]) return -1;
    A->cap[A->n] = cap;
    A->used[A->n] = 0;
    A->n++;
    return 0;
}
/* room for `len` contiguous entries: (*blk, *off) */
static int arena_alloc(arena *A, uint64_t len, uint32_t *blk, uint64_t *off) {
    if (A->n == 0 || A->used[A->n - 1] + len > A->cap[A->n - 1])
        if (arena_new_block(A, len > ABLK ? len : ABLK)) return -1;
    *blk = A->n - 1;
    *off = A->used[A->n - 1];
    A->used[A->n - 1] += len;
    return 0;
}

typedef struct {
    uint32_t key; /* side * V + the other token: side 0 = (other, Z), side 1 = (Z, other) */
    uint32_t pos;
} rec;

/* Same contract as orc_train (bpe_oracle.c): bytes[n], chunk offsets off[n_chunks + 1] (off[n_chunks] == n), up to
 * num_merges merges; pairs_out[2 i], counts_out[i], lens_out[i]; returns the merges done, *status 0 = ok, -2 = the
 * statistics ran empty (the reference raises ValueError from max()), -1 = out of memory, -3 = internal
 * inconsistency (a positive count without a live occurrence: never seen). */
int64_t orc_train_fast(const uint8_t *bytes, uint64_t n, const uint64_t *off, uint64_t n_chunks, int32_t num_merges,
                       int32_t *pairs_out, uint64_t *counts_out, uint64_t *lens_out, int32_t *status) {
    *status = 0;
    if (n >= 0xFFFFFFF0ull) {
        *status = -1;
        return 0;
    }
    const uint64_t V = 256u + (uint64_t)(num_merges > 0 ? num_merges : 0);
    uint32_t *tok = (uint32_t *)malloc((n ? n : 1) * 4), *nxt = (uint32_t *)malloc((n ? n : 1) * 4),
             *prv = (uint32_t *)malloc((n ? n : 1) * 4);
    uint32_t *cnt = (uint32_t *)calloc(V * V, 4);
    uint64_t *kcount = (uint64_t *)calloc(2 * V + 65536 + 1, sizeof(uint64_t));
    heap H = {0, 0, 0};
    arena A = {0, 0, 0, 0, 0};
    rec *recs = 0;
    uint64_t recs_cap = 0;
    hent *tied = 0;
    uint64_t tied_cap = 0;
    int64_t done = 0;
    if (!tok || !nxt || !prv || !cnt || !kcount) goto oom;

    /* ---- the input: tokens, links, chunk starts; the first get_stats; the byte pairs' occurrence lists ---------------- */
    for (uint64_t i = 0; i < n; i++) {
        tok[i] = bytes[i];
        nxt[i] = i + 1 < n ? (uint32_t)(i + 1) : NONE;
        prv[i] = i ? (uint32_t)(i - 1) : NONE;
    }
    for (uint64_t c = 0; c < n_chunks; c++)
        if (off[c] < n && off[c] < off[c + 1]) tok[off[c]] |= FLAG;
    if (n) tok[0] |= FLAG;
    {
        uint64_t *start = kcount; /* 65536 + 1 counters */
        for (uint64_t i = 0; i + 1 < n; i++)
            if (!(tok[i + 1] & FLAG)) start[(uint64_t)bytes[i] * 256 + bytes[i + 1] + 1]++;
        for (uint32_t k = 0; k < 65536; k++) start[k + 1] += start[k];
        uint32_t blk;
        uint64_t o;
        if (arena_new_block(&A, n ? n : 1)) goto oom;
        blk = 0;
        o = 0;
        A.used[0] = start[65536];
        uint64_t *fill = (uint64_t *)malloc(65536 * sizeof(uint64_t));
        if (!fill) goto oom;
        memcpy(fill, start, 65536 * sizeof(uint64_t));
        for (uint64_t i = 0; i + 1 < n; i++)
            if (!(tok[i + 1] & FLAG)) A.blk[0][fill[(uint64_t)bytes[i] * 256 + bytes[i + 1]]++] = (uint32_t)i;
        free(fill);
        for (uint32_t k = 0; k < 65536; k++) {
            const uint64_t len = start[k + 1] - start[k];
            if (!len) continue;
            const uint32_t a = k >> 8, b = k & 255;
            cnt[(uint64_t)a * V + b] = (uint32_t)len;
            hent e = {(uint32_t)len, (uint32_t)len, 0, a, b, blk, o + start[k]};
            if (heap_push(&H, e)) goto oom;
        }
        memset(kcount, 0, (65536 + 1) * sizeof(uint64_t));
    }
    uint64_t cur_len = n;

#define LIVE(p, x, y) ((tok[p] & IDM) == (x) && nxt[p] != NONE && !(tok[nxt[p]] & FLAG) && (tok[nxt[p]] & IDM) == (y))

    for (int32_t it = 0; it < num_merges; it++) {
        /* ---- max(stats, key=stats.get): the highest count, ties by first live occurrence ------------------------------ */
        uint32_t M = 0;
        for (;;) {
            if (!H.n) break;
            const hent t = H.h[0];
            const uint32_t cur = cnt[(uint64_t)t.a * V + t.b];
            if (cur == t.count) {
                M = cur;
                break;
            }
            hent e = heap_pop(&H);
            if (cur) {
                e.count = cur;
                if (heap_push(&H, e)) goto oom;
            }
        }
        if (!M) {
            *status = -2;
            break;
        }
        uint64_t nt = 0;
        while (H.n && H.h[0].count == M) {
            hent e = heap_pop(&H);
            const uint32_t cur = cnt[(uint64_t)e.a * V + e.b];
            if (cur != M) {
                if (cur) {
                    e.count = cur;
                    if (heap_push(&H, e)) goto oom;
                }
                continue;
            }
            if (nt == tied_cap) {
                tied_cap = tied_cap ? tied_cap * 2 : 1024;
                tied = (hent *)realloc(tied, tied_cap * sizeof(hent));
                if (!tied) goto oom;
            }
            tied[nt++] = e;
        }
        uint64_t win = 0;
        uint32_t best = NONE;
        for (uint64_t k = 0; k < nt; k++) {
            hent *e = &tied[k];
            const uint32_t *occ = A.blk[e->blk] + e->off;
            while (e->head < e->len && !LIVE(occ[e->head], e->a, e->b)) e->head++;
            if (e->head == e->len) {
                *status = -3;
                goto out;
            }
            if (nt > 1 && occ[e->head] < best) {
                best = occ[e->head];
                win = k;
            }
        }
        const hent w = tied[win];
        for (uint64_t k = 0; k < nt; k++)
            if (k != win && heap_push(&H, tied[k])) goto oom;
        const uint32_t a = w.a, b = w.b, Z = 256u + (uint32_t)it;
        pairs_out[2 * it] = (int32_t)a;
        pairs_out[2 * it + 1] = (int32_t)b;
        counts_out[it] = M;

        /* ---- merge(ids, (a, b), Z), left to right, with the table kept current ----------------------------------------- */
        uint64_t nrec = 0, sites = 0;
        const uint32_t *occ = A.blk[w.blk] + w.off;
        for (uint32_t k = w.head; k < w.len; k++) {
            const uint32_t p = occ[k];
            if (!LIVE(p, a, b)) continue;
            const uint32_t q = nxt[p], r = nxt[q];
            if (nrec + 2 > recs_cap) {
                recs_cap = recs_cap ? recs_cap * 2 : 1u << 16;
                recs = (rec *)realloc(recs, recs_cap * sizeof(rec));
                if (!recs) goto oom;
            }
            cnt[(uint64_t)a * V + b]--;
            if (!(tok[p] & FLAG)) { /* a left neighbour inside the chunk */
                const uint32_t l = prv[p], L = tok[l] & IDM;
                cnt[(uint64_t)L * V + a]--;
                cnt[(uint64_t)L * V + Z]++;
                recs[nrec].key = L;
                recs[nrec++].pos = l;
            }
            if (r != NONE && !(tok[r] & FLAG)) {
                const uint32_t R = tok[r] & IDM;
                cnt[(uint64_t)b * V + R]--;
                cnt[(uint64_t)Z * V + R]++;
                recs[nrec].key = (uint32_t)V + R;
                recs[nrec++].pos = p;
            }
            tok[p] = (tok[p] & FLAG) | Z;
            tok[q] = DEAD;
            nxt[p] = r;
            if (r != NONE) prv[r] = p;
            sites++;
        }
        cur_len -= sites;
        lens_out[it] = cur_len;
        done = it + 1;

        /* ---- the new pairs' occurrence lists (all of them involve Z): grouped by pair, each in stream order ------------- */
        if (nrec) {
            for (uint64_t k = 0; k < nrec; k++) kcount[recs[k].key + 1]++;
            /* distinct keys are few against 2V late in training: walk the records' keys, not the whole counter array */
            uint32_t blk;
            uint64_t base;
            if (arena_alloc(&A, nrec, &blk, &base)) goto oom;
            uint32_t *dst = A.blk[blk] + base;
            /* offsets: first pass assigns each key its start in order of first appearance */
            uint64_t run = 0;
            for (uint64_t k = 0; k < nrec; k++) {
                const uint32_t key = recs[k].key;
                uint64_t *c = &kcount[key + 1];
                if (*c & (1ull << 63)) continue; /* already placed */
                const uint64_t len = *c;
                *c = (1ull << 63) | run; /* the key's next free index */
                const uint32_t x = key < V ? key : Z, y = key < V ? Z : key - (uint32_t)V;
                const uint32_t cc = cnt[(uint64_t)x * V + y];
                if (cc) {
                    hent e = {cc, (uint32_t)len, 0, x, y, blk, base + run};
                    if (heap_push(&H, e)) goto oom;
                }
                run += len;
            }
            for (uint64_t k = 0; k < nrec; k++) {
                uint64_t *c = &kcount[recs[k].key + 1];
                dst[(*c)++ & ~(1ull << 63)] = recs[k].pos;
            }
            for (uint64_t k = 0; k < nrec; k++) kcount[recs[k].key + 1] = 0;
        }
    }
    goto out;
oom:
    *status = -1;
out:
    free(tok);
    free(nxt);
    free(prv);
    free(cnt);
    free(kcount);
    free(recs);
    free(tied);
    free(H.h);
    for (uint32_t k = 0; k < A.n; k++) free(A.blk[k]);
    free(A.blk);
    free(A.cap);
    free(A.used);
    return done;
}
