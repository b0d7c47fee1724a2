/* filter_sanitize.c -- runs the HOST side of the scrub filter (strainer2_amd/csrc/sk_host_filter.c:
 * argument handling, table parsing, the dictionaries, the float arithmetic that decides how many rows
 * go, the printing) under AddressSanitizer + UBSan on the CPU, with a plain-C test double standing in
 * for the device entry points (sk_filter_*), which are not linked here.  TEST CODE: the double exists
 * only so that the host logic can be checked against the goldens where there is no GPU; the product
 * has no such path (libstrainer_kmer.so's sk_filter_* are HIP only).
 * Built and run by tests/test_sanitizers.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/strainer_kmer.h"

struct sk_filter { int64_t *pan, *meta; uint8_t *gone; uint64_t n; };
static int dummy_ctx;

int sk_ctx_create(sk_ctx **o, int d) { (void)d; *o = (sk_ctx *)&dummy_ctx; return SK_OK; }
void sk_ctx_destroy(sk_ctx *c) { (void)c; }
const char *sk_strerror(int c) { (void)c; return "stub"; }
const char *sk_last_error(const sk_ctx *c) { (void)c; return "stub"; }
void skh_keyset_key(const skh_keyset *ks, uint32_t row, char out[32]) { (void)ks; (void)row; out[0] = 0; }
uint32_t sk_table_rows(const sk_ctx *c) { (void)c; return 0; }

int sk_filter_create(sk_ctx *ctx, sk_filter **out) { (void)ctx; *out = calloc(1, sizeof **out); return *out ? SK_OK : SK_E_NOMEM; }
void sk_filter_destroy(sk_filter *f) { if (f) { free(f->pan); free(f->meta); free(f->gone); free(f); } }
int sk_filter_load(sk_filter *f, const int64_t *pan, const int64_t *meta, const uint8_t *gone, uint64_t n)
{
    f->pan = malloc(n * 8 + 8); f->meta = malloc(n * 8 + 8); f->gone = calloc(n + 1, 1);
    memcpy(f->pan, pan, n * 8); memcpy(f->meta, meta, n * 8);
    if (gone) memcpy(f->gone, gone, n);
    f->n = n;
    return SK_OK;
}
int sk_filter_load_counts(sk_filter *f, uint32_t a, uint32_t b, int32_t c) { (void)f; (void)a; (void)b; (void)c; return SK_E_STATE; }
int sk_filter_sums(sk_filter *f, int64_t *ps, int64_t *ms, uint64_t *np, uint64_t *nm, uint64_t *ng)
{
    int64_t a = 0, b = 0; uint64_t c = 0, d = 0, e = 0, i;
    for (i = 0; i < f->n; i++) {
        if (f->pan[i] > 0) { a += f->pan[i]; c++; }
        if (f->meta[i] > 0) { b += f->meta[i]; d++; }
        e += f->gone[i];
    }
    if (ps) *ps = a;
    if (ms) *ms = b;
    if (np) *np = c;
    if (nm) *nm = d;
    if (ng) *ng = e;
    return SK_OK;
}
int sk_filter_hist(sk_filter *f, int which, int64_t lo, uint32_t nbins, uint64_t *hist)
{
    const int64_t *v = which ? f->meta : f->pan;
    uint64_t i;
    memset(hist, 0, ((size_t)nbins + 1) * 8);
    for (i = 0; i < f->n; i++)
        if (v[i] > 0 && v[i] >= lo) hist[v[i] - lo < (int64_t)nbins ? v[i] - lo : (int64_t)nbins]++;
    return SK_OK;
}
static const double *g_score;
static int by_score(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    if (g_score[x] != g_score[y]) return g_score[x] > g_score[y] ? -1 : 1;
    return x < y ? -1 : 1;
}
int sk_filter_joint(sk_filter *f, int64_t ps, int64_t ms, uint64_t n_scrub, uint8_t *out)
{
    double *score = malloc((f->n + 1) * sizeof *score);
    uint64_t *idx = malloc((f->n + 1) * sizeof *idx), i, n = 0;
    for (i = 0; i < f->n; i++) {
        double s = 0;
        out[i] = f->gone[i];
        if (f->gone[i]) continue;
        if (f->meta[i] > 0 && (double)f->meta[i] / (double)ms > s) s = (double)f->meta[i] / (double)ms;
        if (f->pan[i] > 0 && (double)f->pan[i] / (double)ps > s) s = (double)f->pan[i] / (double)ps;
        score[i] = s;
        idx[n++] = i;
    }
    g_score = score;
    qsort(idx, n, sizeof *idx, by_score);
    for (i = 0; i < n_scrub && i < n; i++) out[idx[i]] = 1;
    free(score); free(idx);
    return SK_OK;
}
int sk_filter_above(sk_filter *f, int64_t tp, int64_t tm, uint8_t *out)
{
    uint64_t i;
    for (i = 0; i < f->n; i++)
        out[i] = (uint8_t)(f->gone[i] || (f->pan[i] > 0 && f->pan[i] > tp) || (f->meta[i] > 0 && f->meta[i] > tm));
    return SK_OK;
}

int main(int argc, char **argv) { return skh_scrub_filter_main(argc, argv, stdout, stderr); }
