/* device_double.c -- runs the HOST side of the programs under AddressSanitizer/UBSan and under
 * ThreadSanitizer on the CPU:
 *   -DDOUBLE_MAIN=skh_strain_detect_main     sk_host_sd.c: reader threads, chunk queue, the replay of the
 *                                            reference's read-pair bookkeeping, the per-strain thread pool,
 *                                            the fused coverage table (+ sk_host.c, sk_host_cov.c)
 *   -DDOUBLE_MAIN=skh_kmer_scrub_count_main  sk_host.c: key-set build, order replay, the decode thread pool
 *                                            with its pinned double buffers and tickets, the table print
 * The device entry points are not linked; a plain-C test double stands in for them: a sorted array of the
 * packed keys and a byte-wise window walk.  TEST CODE only -- the product's lookups are the HIP kernels.
 * Wide (non-ACGT) strain keys, reads with U, RCCL and the scrub filter are outside the double.
 * Built and run by tests/test_sanitizers.py. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/strainer_kmer.h"
#include "../../strainer2_amd/csrc/sk_common.h"

struct sk_ctx { uint64_t *key, *rawkey; uint32_t *row, *loc; uint32_t n, ncols; uint32_t *cols; const struct sk_batch *inflight; uint32_t type_col, inf_value; uint64_t cap; };
struct sk_batch { uint8_t *bytes; uint64_t nbytes; uint32_t *start; uint32_t nrec; };

static int die(const char *what) { fprintf(stderr, "device call %s is outside the test double\n", what); abort(); return -1; }
const char *sk_strerror(int c) { (void)c; return "stub"; }
const char *sk_last_error(const sk_ctx *c) { (void)c; return "stub"; }
int sk_ctx_create(sk_ctx **o, int d) { (void)d; *o = calloc(1, sizeof **o); return *o ? SK_OK : SK_E_NOMEM; }
void sk_ctx_destroy(sk_ctx *c) { if (c) { free(c->key); free(c->row); free(c->cols); free(c->loc); free(c->rawkey); free(c); } }
uint32_t sk_table_rows(const sk_ctx *c) { return c->n; }
uint32_t sk_table_cols(const sk_ctx *c) { return c->ncols; }

typedef struct { uint64_t k; uint32_t r; } kr;
static int kr_cmp(const void *a, const void *b) { const kr *x = a, *y = b; return x->k < y->k ? -1 : x->k > y->k; }
int sk_table_load_ex(sk_ctx *c, const uint64_t *keys, uint32_t n, uint32_t ncols, const uint32_t *loc)
{
    kr *t = malloc((n + 1) * sizeof *t);
    uint32_t i;
    c->loc = malloc((n + 1) * 4); c->rawkey = malloc((n + 1) * 8);
    for (i = 0; i < n; i++) { c->loc[i] = loc ? loc[i] : i; c->rawkey[i] = keys[i]; }
    for (i = 0; i < n; i++) { t[i].k = keys[i]; t[i].r = i; }
    qsort(t, n, sizeof *t, kr_cmp);
    c->key = malloc((n + 1) * 8); c->row = malloc((n + 1) * 4); c->cols = calloc((size_t)n * ncols + 1, 4);
    for (i = 0; i < n; i++) { c->key[i] = t[i].k; c->row[i] = t[i].r; }
    c->n = n; c->ncols = ncols;
    free(t);
    return SK_OK;
}
int sk_table_load_wide(sk_ctx *c, const char *k, const uint32_t *r, uint32_t n) { (void)c; (void)k; (void)r; return n ? die("sk_table_load_wide") : SK_OK; }
/* checks the contract of sk_table_load_text on what the host builder hands over: rows with a position first in
 * locality order, positions ascending, and the text at a row's position IS the row's key (in the orientation
 * the locality flag names) */
int sk_table_load_text(sk_ctx *c, const uint32_t *text2, uint32_t nbases, const uint32_t *first_pos)
{
    uint32_t i, m = 0, *pos_by_loc = malloc((c->n + 1) * 4);
    for (i = 0; i < c->n; i++) pos_by_loc[c->loc[i] & 0x7FFFFFFFu] = first_pos[i];
    while (m < c->n && pos_by_loc[m] != 0xFFFFFFFFu) m++;
    for (i = 0; i < c->n; i++) {
        if (i >= m && pos_by_loc[i] != 0xFFFFFFFFu) return die("sk_table_load_text: rows with a position must come first");
        if (i < m && (pos_by_loc[i] + 31u > nbases || (i && pos_by_loc[i] <= pos_by_loc[i - 1]))) return die("sk_table_load_text: positions must ascend");
    }
    for (i = 0; i < c->n; i++) {
        uint64_t k = 0, rc;
        uint32_t j;
        if (first_pos[i] == 0xFFFFFFFFu) continue;
        for (j = 0; j < 31; j++) { const uint32_t q = first_pos[i] + j; k = (k << 2) | ((text2[q >> 4] >> (2 * (15 - (q & 15)))) & 3u); }
        rc = sk_revcomp62(k);
        if ((k > rc ? k : rc) != c->rawkey[i]) return die("sk_table_load_text: the text at first_pos is not the row's key");
        if (((c->loc[i] >> 31) != 0) != (k > rc)) return die("sk_table_load_text: orientation flag does not match the text");
    }
    free(pos_by_loc);
    return SK_OK;
}
/* the device-built table of strain_detect's opening, restated for the double: keys of the marked windows, rows by first occurrence */
typedef struct { uint64_t k; uint32_t p; } kp;
static int kp_cmp(const void *a, const void *b) { const kp *x = a, *y = b; return x->k < y->k ? -1 : x->k > y->k ? 1 : x->p < y->p ? -1 : x->p > y->p; }
static int u32_cmp(const void *a, const void *b) { const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }
int sk_table_build_from_text(sk_ctx *c, const uint32_t *text2, const uint32_t *ok, uint32_t nbases, uint32_t nstarts, uint32_t ncols, uint32_t col0, uint32_t *nrows)
{
    kp *w = malloc(((size_t)nstarts + 1) * sizeof *w);
    uint32_t *first = malloc(((size_t)nstarts + 1) * 4), nw = 0, nf = 0, p, i;
    uint64_t *keys;
    for (p = 0; p + 31 <= nbases; p++) {
        uint64_t k = 0, rc;
        uint32_t j;
        if (!((ok[p >> 5] >> (p & 31)) & 1u)) continue;
        for (j = 0; j < 31; j++) { const uint32_t q = p + j; k = (k << 2) | ((text2[q >> 4] >> (2 * (15 - (q & 15)))) & 3u); }
        rc = sk_revcomp62(k);
        if (nw >= nstarts) return die("sk_table_build_from_text: more marked windows than nstarts");
        w[nw].k = k > rc ? k : rc; w[nw].p = p; nw++;
    }
    if (nw != nstarts) return die("sk_table_build_from_text: nstarts does not count the marked windows");
    qsort(w, nw, sizeof *w, kp_cmp);
    for (i = 0; i < nw; i++) if (i == 0 || w[i].k != w[i - 1].k) first[nf++] = w[i].p;      /* a key's lowest position */
    qsort(first, nf, 4, u32_cmp);
    keys = malloc(((size_t)nf + 1) * 8);
    for (i = 0; i < nf; i++) {
        uint64_t k = 0, rc;
        uint32_t j;
        for (j = 0; j < 31; j++) { const uint32_t q = first[i] + j; k = (k << 2) | ((text2[q >> 4] >> (2 * (15 - (q & 15)))) & 3u); }
        rc = sk_revcomp62(k);
        keys[i] = k > rc ? k : rc;
    }
    free(w); free(first);
    free(c->key); free(c->row); free(c->cols); free(c->loc); free(c->rawkey);
    c->key = NULL; c->row = NULL; c->cols = NULL; c->loc = NULL; c->rawkey = NULL;
    sk_table_load_ex(c, keys, nf, ncols, NULL);
    for (i = 0; i < nf; i++) c->cols[i] = col0;
    free(keys);
    *nrows = nf;
    return SK_OK;
}
int sk_table_export_keys(sk_ctx *c, uint64_t *out) { memcpy(out, c->rawkey, (size_t)c->n * 8); return SK_OK; }
int sk_table_export_keys_of(sk_ctx *c, const uint32_t *rows, uint32_t n, uint64_t *out) { uint32_t i; for (i = 0; i < n; i++) { if (rows[i] >= c->n) return die("sk_table_export_keys_of: row out of range"); out[i] = c->rawkey[rows[i]]; } return SK_OK; }
int sk_counts_set_rows(sk_ctx *c, uint32_t col, const uint32_t *rows, uint32_t n, uint32_t v) { uint32_t i; for (i = 0; i < n; i++) c->cols[(size_t)col * c->n + rows[i]] = v; return SK_OK; }
int sk_counts_set(sk_ctx *c, uint32_t col, const uint32_t *in) { memcpy(c->cols + (size_t)col * c->n, in, (size_t)c->n * 4); return SK_OK; }
int sk_counts_fetch(sk_ctx *c, uint32_t col, uint32_t *out) { memcpy(out, c->cols + (size_t)col * c->n, (size_t)c->n * 4); return SK_OK; }

static int64_t find(const sk_ctx *c, uint64_t k)
{
    uint32_t lo = 0, hi = c->n;
    while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (c->key[mid] < k) lo = mid + 1; else hi = mid; }
    return lo < c->n && c->key[lo] == k ? (int64_t)c->row[lo] : -1;
}

int sk_batch_create(sk_ctx *c, sk_batch **out) { (void)c; *out = calloc(1, sizeof **out); return *out ? SK_OK : SK_E_NOMEM; }
void sk_batch_destroy(sk_batch *b) { if (b) { free(b->bytes); free(b->start); free(b); } }
int sk_batch_sync(sk_batch *b) { return b ? SK_OK : SK_E_ARG; }
int sk_batch_fill(sk_batch *b, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec);
/* a packed batch: its bytes made again (a separator for every byte that was no base: which of N, n or '\n' it was does not matter to a
 * tally -- but a record's own end must stay a '\n', and it does: the byte behind every record was one) */
int sk_batch_fill_packed(sk_batch *b, const void *packed, uint64_t n, const uint32_t *rec_start, uint32_t nrec)
{
    const uint64_t nch = (n + 15u) >> 4;
    const uint32_t *codes = (const uint32_t *)packed;
    const uint16_t *inv = (const uint16_t *)((const uint8_t *)packed + nch * 4u);
    uint8_t *bytes = (uint8_t *)malloc(n ? n : 1);
    uint64_t i;
    int rc;
    if (!bytes) return SK_E_NOMEM;
    for (i = 0; i < n; i++) {
        const uint64_t g = i >> 4;
        const unsigned k = (unsigned)(i & 15u);
        bytes[i] = (inv[g] >> k) & 1u ? (uint8_t)'\n' : (uint8_t)"ACGT"[(codes[g] >> (30u - 2u * k)) & 3u];
    }
    rc = sk_batch_fill(b, bytes, n, rec_start, nrec);
    free(bytes);
    return rc;
}
int sk_batch_fill(sk_batch *b, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec)
{
    free(b->bytes); free(b->start);
    b->bytes = malloc(nbytes); memcpy(b->bytes, stream, nbytes);
    b->start = malloc((size_t)nrec * 4); memcpy(b->start, rec_start, (size_t)nrec * 4);
    b->nbytes = nbytes; b->nrec = nrec;
    return SK_OK;
}
int sk_tally_launch(sk_ctx *c, const sk_batch *b, uint32_t type_col, uint32_t inf_value, uint64_t cap)
{
    c->inflight = b; c->type_col = type_col; c->inf_value = inf_value; c->cap = cap;
    return SK_OK;
}
int sk_tally_collect(sk_ctx *c, uint32_t *tally, sk_hit *hits, uint64_t *nhits)
{
    const sk_batch *b = c->inflight;
    uint64_t nh = 0;
    uint32_t r;
    const uint32_t *type = c->cols + (size_t)c->type_col * c->n;
    for (r = 0; r < b->nrec; r++) {
        const uint64_t beg = b->start[r], end = r + 1 < b->nrec ? b->start[r + 1] : b->nbytes;
        uint64_t f = 0, i;
        uint32_t run = 0, h = 0, inf = 0;
        for (i = beg; i < end; i++) {
            const uint8_t ch = b->bytes[i];
            if (!sk_is_acgt(ch)) { run = 0; continue; }
            f = ((f << 2) | sk_code(ch)) & ((1ull << 62) - 1);
            if (++run >= 31) {
                const uint64_t rc = sk_revcomp62(f);
                const int64_t row = find(c, f > rc ? f : rc);
                if (row >= 0) {
                    h++;
                    if (type[row] == c->inf_value) { if (nh < c->cap) { hits[nh].pos = (uint32_t)(i - 30); hits[nh].row = (uint32_t)row; } nh++; inf++; }
                }
            }
        }
        tally[2 * r] = h; tally[2 * r + 1] = inf;
    }
    *nhits = nh;
    return SK_OK;
}
int sk_tally_collect_sparse(sk_ctx *c, sk_tally_rec *out, uint64_t cap, uint64_t *n, sk_hit *hits, uint64_t *nhits)
{
    const uint32_t nrec = c->inflight->nrec;
    uint32_t *dense = calloc((size_t)nrec * 2 + 2, 4), r;
    uint64_t k = 0;
    const int rc = sk_tally_collect(c, dense, hits, nhits);
    for (r = 0; r < nrec; r++)
        if (dense[2 * r]) { if (k < cap) { out[k].rec = r; out[k].all = dense[2 * r]; out[k].inf = dense[2 * r + 1]; } k++; }
    *n = k;
    free(dense);
    return rc;
}
/* the union of several tables: here simply the members asked one after the other, results merged into the union's format */
struct sk_union { sk_ctx *m[SK_UNION_MAX]; uint32_t n; const sk_batch *b; uint64_t cap; uint32_t type_col, inf_value; };
int sk_union_create(sk_ctx *const *members, uint32_t n, uint32_t type_col, uint32_t inf_value, sk_union **out)
{
    sk_union *u;
    uint32_t i;
    if (!members || !out || n < 1 || n > SK_UNION_MAX) return SK_E_ARG;
    if (getenv("DOUBLE_NO_UNION")) return SK_E_STATE;                      /* (tests: the member-by-member way) */
    u = calloc(1, sizeof *u);
    for (i = 0; i < n; i++) u->m[i] = members[i];
    u->n = n; u->type_col = type_col; u->inf_value = inf_value;
    *out = u;
    return SK_OK;
}
void sk_union_destroy(sk_union *u) { free(u); }
int sk_union_sync(sk_union *u) { (void)u; return SK_OK; }
uint32_t sk_union_members(const sk_union *u) { return u->n; }
uint32_t sk_union_rows(const sk_union *u) { uint32_t i, r = 0; for (i = 0; i < u->n; i++) r += u->m[i]->n; return r; }
const char *sk_union_last_error(const sk_union *u) { (void)u; return "stub"; }
int sk_union_tally_launch(sk_union *u, const sk_batch *b, uint64_t cap)
{
    u->b = b; u->cap = cap;
    return SK_OK;
}
int sk_union_tally_collect(sk_union *u, sk_tally_rec *out, uint64_t cap, uint64_t *n, sk_hit *hits, uint64_t *nhits)
{
    const uint32_t nrec = u->b->nrec;
    sk_tally_rec *one = malloc(((size_t)nrec + 1) * sizeof *one);
    sk_hit *oh = malloc(((size_t)u->cap + 1) * sizeof *oh);
    uint64_t k = 0, nh = 0, i, got, goth;
    uint32_t s;
    for (s = 0; s < u->n; s++) {
        sk_tally_launch(u->m[s], u->b, u->type_col, u->inf_value, u->cap);
        sk_tally_collect_sparse(u->m[s], one, nrec, &got, oh, &goth);
        for (i = 0; i < got; i++) { if (k < cap) { out[k].rec = one[i].rec * u->n + s; out[k].all = one[i].all; out[k].inf = one[i].inf; } k++; }
        for (i = 0; i < goth; i++) { if (nh < u->cap && i < u->cap) { hits[nh].pos = oh[i].pos; hits[nh].row = oh[i].row | (s << SK_UNION_ROW_BITS); } nh++; }
    }
    free(one); free(oh);
    *n = k; *nhits = nh;
    return SK_OK;
}
int sk_tally_batch(sk_ctx *c, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec, uint32_t type_col,
                   uint32_t inf_value, uint32_t *tally, sk_hit *hits, uint64_t cap, uint64_t *nhits)
{
    sk_batch b;
    int rc;
    b.bytes = (uint8_t *)stream; b.nbytes = nbytes; b.start = (uint32_t *)rec_start; b.nrec = nrec;
    sk_tally_launch(c, &b, type_col, inf_value, cap);
    rc = sk_tally_collect(c, tally, hits, nhits);
    return rc;
}
int sk_distinct_count(sk_ctx *ctx, const uint64_t *keys, const uint32_t *sample, uint64_t n, uint32_t ns, uint64_t *uniq, uint64_t *total)
{
    uint64_t i, j;
    (void)ctx;
    memset(uniq, 0, (size_t)ns * 8); memset(total, 0, (size_t)ns * 8);
    for (i = 0; i < n; i++) {
        int seen = 0;
        total[sample[i]]++;
        for (j = 0; j < i && !seen; j++) seen = keys[j] == keys[i] && sample[j] == sample[i];
        uniq[sample[i]] += !seen;
    }
    return SK_OK;
}
/* the counting scan: every window of the record stream that is a key bumps counts[col][row] */
static pthread_mutex_t scan_mu = PTHREAD_MUTEX_INITIALIZER;
int sk_scan_stream(sk_ctx *c, const uint8_t *s, uint64_t n, uint32_t col)
{
    uint64_t f = 0, i;
    uint32_t run = 0, *cnt = c->cols + (size_t)col * c->n;
    pthread_mutex_lock(&scan_mu);
    for (i = 0; i < n; i++) {
        if (!sk_is_acgt(s[i])) { run = 0; continue; }
        f = ((f << 2) | sk_code(s[i])) & ((1ull << 62) - 1);
        if (++run >= 31) {
            const uint64_t rc = sk_revcomp62(f);
            const int64_t row = find(c, f > rc ? f : rc);
            if (row >= 0) cnt[row]++;
        }
    }
    pthread_mutex_unlock(&scan_mu);
    return SK_OK;
}
int sk_pinned_alloc(sk_ctx *c, void **p, uint64_t n) { (void)c; *p = malloc(n); return *p ? SK_OK : SK_E_NOMEM; }
int sk_pinned_free(sk_ctx *c, void *p) { (void)c; free(p); return SK_OK; }
int sk_scan_pinned(sk_ctx *c, const uint8_t *s, uint64_t n, uint32_t col, uint64_t *t) { if (t) *t = 1; return sk_scan_stream(c, s, n, col); }
int sk_ticket_wait(sk_ctx *c, uint64_t t) { (void)c; (void)t; return SK_OK; }
/* a packed batch (sk_pack_stream): its bytes made again -- a base for a code, a separator for every byte that was none (which of N,
 * n or '\n' it was makes no difference to a window count: either ends the windows that hold it) -- and scanned as bytes */
int sk_scan_pinned_packed(sk_ctx *c, const void *packed, uint64_t n, uint32_t col, uint64_t *t)
{
    const uint64_t nch = (n + 15u) >> 4;
    const uint32_t *codes = (const uint32_t *)packed;
    const uint16_t *inv = (const uint16_t *)((const uint8_t *)packed + nch * 4u);
    uint8_t *b = (uint8_t *)malloc(n ? n : 1);
    uint64_t i;
    int rc;
    if (!b) return SK_E_NOMEM;
    for (i = 0; i < n; i++) {
        const uint64_t g = i >> 4;
        const unsigned k = (unsigned)(i & 15u);
        b[i] = (inv[g] >> k) & 1u ? (uint8_t)'\n' : (uint8_t)"ACGT"[(codes[g] >> (30u - 2u * k)) & 3u];
    }
    if (t) *t = 1;
    rc = sk_scan_stream(c, b, n, col);
    free(b);
    return rc;
}
int sk_sync(sk_ctx *c) { (void)c; return SK_OK; }
/* ---- a communicator for the multi-PROCESS CPU tests (tests/test_multirank_protocol.py).  DOUBLE_COMM_DIR names a fresh
 * directory all ranks see.  Collective number k of rank r is the file c<k>.r<r> = {kind, element count, payload}; a rank
 * writes its own, then waits (bounded: DOUBLE_COMM_TIMEOUT seconds, default 20) for every other rank's file of the SAME
 * number and checks that kind and count are the same -- what RCCL needs to be true and cannot check: a mismatch (exit 98) or a
 * rank that never arrives (exit 97) is exactly the hang ADVICE r03 found.  Every call is also appended to log.r<r>, so the
 * test can compare the ranks' sequences line by line.  Without DOUBLE_COMM_DIR the collectives are outside the double, as before. */
#include <time.h>
#include <unistd.h>
static struct { int on, rank, world, seq; char dir[400]; } g_comm;
enum { DC_SUM_U32, DC_MAX_U64 };
static int dc_collective(const char *kind, int op, void *buf, size_t count)
{
    const size_t esz = op == DC_SUM_U32 ? 4 : 8, bytes = count * esz;
    char path[512], tmp[512], head[64], theirs[64];
    unsigned char *acc = malloc(bytes + 1), *in = malloc(bytes + 1);
    const double limit = getenv("DOUBLE_COMM_TIMEOUT") ? atof(getenv("DOUBLE_COMM_TIMEOUT")) : 20.0;
    FILE *f;
    int r;
    size_t i;
    g_comm.seq++;
    snprintf(head, sizeof head, "%s %zu", kind, count);
    snprintf(path, sizeof path, "%s/log.r%d", g_comm.dir, g_comm.rank);
    if ((f = fopen(path, "a")) != NULL) { fprintf(f, "%s\n", head); fclose(f); }
    snprintf(tmp, sizeof tmp, "%s/c%d.r%d.tmp", g_comm.dir, g_comm.seq, g_comm.rank);
    snprintf(path, sizeof path, "%s/c%d.r%d", g_comm.dir, g_comm.seq, g_comm.rank);
    if (!(f = fopen(tmp, "wb"))) return die("double comm: cannot write");
    fwrite(head, 1, sizeof head, f); fwrite(buf, 1, bytes, f); fclose(f);
    rename(tmp, path);
    memcpy(acc, buf, bytes);
    for (r = 0; r < g_comm.world; r++) {
        struct timespec t0, t1, nap = {0, 1000000};
        if (r == g_comm.rank) continue;
        snprintf(path, sizeof path, "%s/c%d.r%d", g_comm.dir, g_comm.seq, r);
        clock_gettime(CLOCK_MONOTONIC, &t0);
        while (!(f = fopen(path, "rb"))) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit) {
                fprintf(stderr, "double comm: rank %d waited %.0f s for rank %d in collective #%d (%s): the ranks' sequences differ\n",
                        g_comm.rank, limit, r, g_comm.seq, head);
                exit(97);
            }
            nanosleep(&nap, NULL);
        }
        if (fread(theirs, 1, sizeof theirs, f) != sizeof theirs || strcmp(theirs, head) != 0) {
            fprintf(stderr, "double comm: collective #%d is '%s' on rank %d and '%s' on rank %d\n", g_comm.seq, head, g_comm.rank, theirs, r);
            exit(98);
        }
        if (fread(in, 1, bytes, f) != bytes) { fprintf(stderr, "double comm: short payload\n"); exit(98); }
        fclose(f);
        if (op == DC_SUM_U32) for (i = 0; i < count; i++) ((uint32_t *)acc)[i] += ((uint32_t *)in)[i];
        else for (i = 0; i < count; i++) if (((uint64_t *)in)[i] > ((uint64_t *)acc)[i]) ((uint64_t *)acc)[i] = ((uint64_t *)in)[i];
    }
    memcpy(buf, acc, bytes);
    free(acc); free(in);
    return SK_OK;
}
int sk_comm_init_ex(sk_ctx *c, int r, int w, const char *f, int t, int s)
{
    uint32_t bad = s != 0;
    (void)c; (void)f; (void)t;
    if (!getenv("DOUBLE_COMM_DIR")) return die("sk_comm_init_ex");
    g_comm.on = 1; g_comm.rank = r; g_comm.world = w; g_comm.seq = 0;
    snprintf(g_comm.dir, sizeof g_comm.dir, "%s", getenv("DOUBLE_COMM_DIR"));
    dc_collective("rendezvous", DC_SUM_U32, &bad, 1);        /* (the real one: sk_rendezvous.h, every rank learns of a failed set-up) */
    return bad ? SK_E_RCCL : SK_OK;
}
int sk_comm_init(sk_ctx *c, int r, int w, const char *f, int t) { return sk_comm_init_ex(c, r, w, f, t, 0); }
int sk_comm_world(const sk_ctx *c) { (void)c; return g_comm.on ? g_comm.world : 0; }
int sk_comm_sum_u32(sk_ctx *c, uint32_t v, uint32_t *s)
{
    (void)c;
    if (!g_comm.on) return die("sk_comm_sum_u32");
    *s = v;
    return dc_collective("sum_u32", DC_SUM_U32, s, 1);
}
int sk_comm_max_u64(sk_ctx *c, uint64_t *v, uint32_t n)
{
    (void)c;
    if (!g_comm.on) return SK_OK;                             /* no communicator: the values stay */
    return dc_collective("max_u64", DC_MAX_U64, v, n);
}
int sk_comm_agree_u64(sk_ctx *c, uint64_t v, int *a)
{
    uint64_t h[2];
    h[0] = v; h[1] = ~v;
    sk_comm_max_u64(c, h, 2);
    *a = h[0] == v && h[1] == ~v;
    return SK_OK;
}
int sk_counts_zero(sk_ctx *c, uint32_t col) { memset(c->cols + (size_t)col * c->n, 0, (size_t)c->n * 4); return SK_OK; }
int sk_counts_allreduce(sk_ctx *c, void *comm)
{
    (void)comm;
    if (!g_comm.on) return die("sk_counts_allreduce");
    return dc_collective("allreduce_u32", DC_SUM_U32, c->cols, (size_t)c->n * c->ncols);
}
int skh_scrub_filter_resident(sk_ctx *c, const skh_keyset *k, int d, double m, int i, FILE *o, FILE *e) { (void)c; (void)k; (void)d; (void)m; (void)i; (void)o; (void)e; return die("skh_scrub_filter_resident"); }

int sk_first_seen_count(sk_ctx *ctx, const uint32_t *id, const uint32_t *sample, uint64_t n, uint32_t nids, uint32_t ns, uint64_t *uniq, uint64_t *total)
{
    (void)ctx; (void)id; (void)sample; (void)n; (void)nids; (void)ns; (void)uniq; (void)total;
    return die("sk_first_seen_count");
}

#ifndef DOUBLE_NO_MAIN
#ifndef DOUBLE_MAIN
#define DOUBLE_MAIN skh_strain_detect_main
#endif
int main(int argc, char **argv) { return DOUBLE_MAIN(argc, argv, stdout, stderr); }
#endif

/* the device-side gzip decoder (sk_inflate.hip) is not part of the CPU builds: "not a file this path takes" */
struct sk_inflater;
int sk_ctx_device_(const sk_ctx *c) { (void)c; return 0; }
int sk_inflater_create(int device, struct sk_inflater **out) { (void)device; *out = 0; return -1; }
void sk_inflater_destroy(struct sk_inflater *f) { (void)f; }
uint64_t sk_inflate_gz_size(const uint8_t *gz, uint64_t n) { (void)gz; (void)n; return 0; }
int sk_inflate_gz(struct sk_inflater *f, const uint8_t *gz, uint64_t n, uint8_t *t, uint64_t cap, uint64_t *len, uint32_t *crc) { (void)f; (void)gz; (void)n; (void)t; (void)cap; (void)len; (void)crc; return -100; }
int sk_scan_device_packed(sk_ctx *c, const void *p, uint64_t n, uint32_t col) { uint64_t t; return sk_scan_pinned_packed(c, p, n, col, &t); }
