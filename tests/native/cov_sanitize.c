/* cov_sanitize.c -- runs the HOST side of coverage_depth (strainer2_amd/csrc/sk_host_cov.c: parsing,
 * the script's dictionaries and their order, number formatting) under AddressSanitizer + UBSan on the
 * CPU, with a plain-C test double for the one device call (sk_distinct_count), which is not linked
 * here.  TEST CODE only; the product's sk_distinct_count is the HIP kernel in sk_cover.hip.
 * Built and run by tests/test_sanitizers.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/strainer_kmer.h"

static int dummy_ctx;
int sk_ctx_create(sk_ctx **o, int d) { (void)d; *o = (sk_ctx *)&dummy_ctx; return SK_OK; }
void sk_ctx_destroy(sk_ctx *c) { (void)c; }
const char *sk_strerror(int c) { (void)c; return "stub"; }
const char *sk_last_error(const sk_ctx *c) { (void)c; return "stub"; }

typedef struct { uint32_t s; uint64_t k; } pair;
static int pair_cmp(const void *a, const void *b)
{
    const pair *x = a, *y = b;
    if (x->s != y->s) return x->s < y->s ? -1 : 1;
    return x->k < y->k ? -1 : x->k > y->k;
}
int sk_distinct_count(sk_ctx *ctx, const uint64_t *keys, const uint32_t *sample, uint64_t n, uint32_t nsamples,
                      uint64_t *out_unique, uint64_t *out_total)
{
    pair *p = malloc((n + 1) * sizeof *p);
    uint64_t i;
    (void)ctx;
    memset(out_unique, 0, (size_t)nsamples * 8);
    memset(out_total, 0, (size_t)nsamples * 8);
    for (i = 0; i < n; i++) { p[i].s = sample[i]; p[i].k = keys[i]; out_total[sample[i]]++; }
    qsort(p, n, sizeof *p, pair_cmp);
    for (i = 0; i < n; i++)
        if (i == 0 || pair_cmp(&p[i - 1], &p[i])) out_unique[p[i].s]++;
    free(p);
    return SK_OK;
}

int sk_first_seen_count(sk_ctx *ctx, const uint32_t *id, const uint32_t *sample, uint64_t n, uint32_t nids, uint32_t nsamples,
                        uint64_t *out_unique, uint64_t *out_total)
{
    uint8_t *seen = calloc(nids + 1u, 1);
    uint64_t i;
    (void)ctx;
    memset(out_unique, 0, (size_t)nsamples * 8);
    memset(out_total, 0, (size_t)nsamples * 8);
    for (i = 0; i < n; i++) {
        out_total[sample[i]]++;
        if (!seen[id[i]]) { seen[id[i]] = 1; out_unique[sample[i]]++; }
    }
    free(seen);
    return SK_OK;
}

int main(int argc, char **argv) { return skh_coverage_depth_main(argc, argv, stdout, stderr); }
