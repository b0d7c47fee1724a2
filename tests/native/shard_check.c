/* shard_check.c -- the sharding identity of the list walk on the CPU (TEST CODE; the device calls are the plain-C
 * test double of tests/native/device_double.c):
 *     sum over ranks r of [ skh_scan_list(list, rank r of W) ]  ==  skh_scan_list(list, rank 0 of 1)
 * for the product's own dealing of list items to ranks (longest-first by size) and its cutting of big plain-text
 * files into byte-range pieces (sk_host.c), several decode threads per rank.
 * usage: shard_check <strain.fa> <list> <world>      prints "OK <sum of counts> <bases>" or the first difference
 *        SHARD_VARY_THREADS=1: rank r scans with SK_THREADS = 1 + (3 r mod 7) -- the plan must not follow a rank's thread count
 *        shard_check --plan <list> <world>          prints the plan's hash and the owner of every list line */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/strainer_kmer.h"

static uint32_t *scan(const skh_keyset *ks, const char *list, uint32_t rank, uint32_t world, uint64_t *bases, int *rc_out)
{
    sk_ctx *ctx = NULL;
    uint32_t *c = calloc(ks->nrows + 1, 4);
    int rc = sk_ctx_create(&ctx, 0);
    if (rc == SK_OK) rc = skh_keyset_load(ctx, ks, 4);
    if (rc == SK_OK) rc = skh_scan_list(ctx, list, NULL, 2, NULL, stderr, rank, world, bases);
    if (rc == SK_OK) rc = sk_counts_fetch(ctx, 2, c);
    sk_ctx_destroy(ctx);
    *rc_out = rc;
    return c;
}

int main(int argc, char **argv)
{
    skh_keyset ks;
    uint32_t world, r, i, *ref, *sum;
    uint64_t bases1 = 0, basesw = 0, total = 0;
    int rc;
    if (argc != 4) return 2;
    world = (uint32_t)atoi(argv[3]);
    if (!strcmp(argv[1], "--plan")) {
        uint64_t h = 0;
        uint32_t own[64], n = 0;
        if (skh_list_plan_hash(argv[2], NULL, world, &h) != SK_OK || skh_list_plan_owners(argv[2], NULL, world, own, 64, &n) != SK_OK) return 1;
        printf("%016llx", (unsigned long long)h);
        for (i = 0; i < n && i < 64; i++) printf(" %d", (int)own[i]);
        printf("\n");
        return 0;
    }
    if (skh_keyset_from_file(&ks, argv[1], SK_REF_TABLE_SLOTS, 1, 1) != SK_OK) { puts("keyset failed"); return 1; }
    ref = scan(&ks, argv[2], 0, 1, &bases1, &rc);
    if (rc != SK_OK) { printf("unsharded scan failed: %d\n", rc); return 1; }
    sum = calloc(ks.nrows + 1, 4);
    for (r = 0; r < world; r++) {
        uint32_t *c;
        if (getenv("SHARD_VARY_THREADS")) { char t[8]; snprintf(t, sizeof t, "%u", 1u + 3u * r % 7u); setenv("SK_THREADS", t, 1); }
        c = scan(&ks, argv[2], r, world, &basesw, &rc);
        if (rc != SK_OK) { printf("rank %u of %u failed: %d\n", r, world, rc); return 1; }
        for (i = 0; i < ks.nrows; i++) sum[i] += c[i];
        free(c);
    }
    for (i = 0; i < ks.nrows; i++) {
        if (sum[i] != ref[i]) { printf("row %u: sharded %u, unsharded %u\n", i, sum[i], ref[i]); return 1; }
        total += ref[i];
    }
    if (bases1 != basesw) { printf("bases: sharded %llu, unsharded %llu\n", (unsigned long long)basesw, (unsigned long long)bases1); return 1; }
    printf("OK %llu %llu\n", (unsigned long long)total, (unsigned long long)bases1);
    free(ref); free(sum);
    skh_keyset_free(&ks);
    return 0;
}
