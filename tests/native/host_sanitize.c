/* host_sanitize.c -- drives the GPU-free parts of the host layer (record parser, keyset build,
 * BIO_hash order replay, stream writer) under AddressSanitizer + UBSan on the CPU.
 * Built and run by tests/test_sanitizers.py:
 *   gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -DSK_HOST_ONLY \
 *       tests/native/host_sanitize.c strainer2_amd/csrc/sk_host.c -lz -o <tmp>/host_sanitize
 * (GPU AddressSanitizer is not available on the pool; the device code is covered by parity tests.) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/strainer_kmer.h"

/* the device layer is not linked here: stubs that must never be reached */
static int unreachable(const char *name) { fprintf(stderr, "device call %s in a host-only test\n", name); abort(); return -1; }
int sk_table_load_ex(sk_ctx *c, const uint64_t *k, uint32_t n, uint32_t nc, const uint32_t *l) { (void)c; (void)k; (void)n; (void)nc; (void)l; return unreachable("sk_table_load_ex"); }
int sk_table_load_wide(sk_ctx *c, const char *k, const uint32_t *r, uint32_t n) { (void)c; (void)k; (void)r; (void)n; return unreachable("sk_table_load_wide"); }
int sk_table_load_text(sk_ctx *c, const uint32_t *t, uint32_t n, const uint32_t *f) { (void)c; (void)t; (void)n; (void)f; return unreachable("sk_table_load_text"); }
int sk_table_build_from_text(sk_ctx *c, const uint32_t *t, const uint32_t *o, uint32_t n, uint32_t ns, uint32_t nc, uint32_t v, uint32_t *r) { (void)c; (void)t; (void)o; (void)n; (void)ns; (void)nc; (void)v; (void)r; return unreachable("sk_table_build_from_text"); }
int sk_table_export_keys(sk_ctx *c, uint64_t *k) { (void)c; (void)k; return unreachable("sk_table_export_keys"); }
int sk_table_export_keys_of(sk_ctx *c, const uint32_t *r, uint32_t n, uint64_t *k) { (void)c; (void)r; (void)n; (void)k; return unreachable("sk_table_export_keys_of"); }
int sk_counts_set(sk_ctx *c, uint32_t col, const uint32_t *in) { (void)c; (void)col; (void)in; return unreachable("sk_counts_set"); }
int sk_counts_fetch(sk_ctx *c, uint32_t col, uint32_t *out) { (void)c; (void)col; (void)out; return unreachable("sk_counts_fetch"); }
int sk_scan_stream(sk_ctx *c, const uint8_t *s, uint64_t n, uint32_t col) { (void)c; (void)s; (void)n; (void)col; return unreachable("sk_scan_stream"); }
int sk_ctx_create(sk_ctx **o, int d) { (void)o; (void)d; return unreachable("sk_ctx_create"); }
void sk_ctx_destroy(sk_ctx *c) { (void)c; unreachable("sk_ctx_destroy"); }
uint32_t sk_table_rows(const sk_ctx *c) { (void)c; return (uint32_t)unreachable("sk_table_rows"); }
uint32_t sk_table_cols(const sk_ctx *c) { (void)c; return (uint32_t)unreachable("sk_table_cols"); }
int sk_comm_init(sk_ctx *c, int r, int w, const char *f, int t) { (void)c; (void)r; (void)w; (void)f; (void)t; return unreachable("sk_comm_init"); }
int sk_comm_init_ex(sk_ctx *c, int r, int w, const char *f, int t, int s) { (void)c; (void)r; (void)w; (void)f; (void)t; (void)s; return unreachable("sk_comm_init_ex"); }
int skh_strain_detect_resident(sk_ctx *c, skh_keyset *k, const char *a, int n, char **v, FILE *o, FILE *e) { (void)c; (void)k; (void)a; (void)n; (void)v; (void)o; (void)e; return unreachable("skh_strain_detect_resident"); }
int sk_comm_sum_u32(sk_ctx *c, uint32_t v, uint32_t *s) { (void)c; (void)v; (void)s; return unreachable("sk_comm_sum_u32"); }
int sk_comm_agree_u64(sk_ctx *c, uint64_t v, int *a) { (void)c; (void)v; *a = 1; return SK_OK; }
int sk_comm_max_u64(sk_ctx *c, uint64_t *v, uint32_t n) { (void)c; (void)v; (void)n; return SK_OK; }
int sk_comm_world(const sk_ctx *c) { (void)c; return 0; }
int sk_counts_zero(sk_ctx *c, uint32_t col) { (void)c; (void)col; return unreachable("sk_counts_zero"); }
int sk_counts_allreduce(sk_ctx *c, void *comm) { (void)c; (void)comm; return unreachable("sk_counts_allreduce"); }
int sk_pinned_alloc(sk_ctx *c, void **p, uint64_t n) { (void)c; (void)p; (void)n; return unreachable("sk_pinned_alloc"); }
int sk_pinned_free(sk_ctx *c, void *p) { (void)c; (void)p; return unreachable("sk_pinned_free"); }
int sk_scan_pinned(sk_ctx *c, const uint8_t *s, uint64_t n, uint32_t col, uint64_t *t) { (void)c; (void)s; (void)n; (void)col; (void)t; return unreachable("sk_scan_pinned"); }
int sk_ticket_wait(sk_ctx *c, uint64_t t) { (void)c; (void)t; return unreachable("sk_ticket_wait"); }
int sk_scan_pinned_packed(sk_ctx *c, const void *s, uint64_t n, uint32_t col, uint64_t *t) { (void)c; (void)s; (void)n; (void)col; (void)t; return unreachable("sk_scan_pinned_packed"); }
int skh_scrub_filter_resident(sk_ctx *c, const skh_keyset *k, int d, double m, int i, FILE *o, FILE *e) { (void)c; (void)k; (void)d; (void)m; (void)i; (void)o; (void)e; return unreachable("skh_scrub_filter_resident"); }
const char *sk_strerror(int c) { (void)c; return "stub"; }
const char *sk_last_error(const sk_ctx *c) { (void)c; return "stub"; }

static unsigned long long g_bytes, g_sum;
static int sink(void *user, const uint8_t *chunk, uint64_t n)
{
    uint64_t i;
    (void)user;
    for (i = 0; i < n; i++) g_sum += chunk[i];
    g_bytes += n;
    return 0;
}

int main(int argc, char **argv)
{
    int i;
    for (i = 1; i < argc; i++) {
        skh_keyset ks;
        uint64_t bases = 0;
        char key[32];
        int64_t nrec = skh_decode_file(argv[i], 4096, sink, NULL, &bases);
        int rc = skh_keyset_from_file(&ks, argv[i], 50, 1, 1);
        if (rc == SK_OK) {
            if (ks.nrows) { skh_keyset_key(&ks, 0, key); skh_keyset_key(&ks, ks.nrows - 1, key); }
            printf("%s: records=%lld bases=%llu keys=%u wide=%u slots=%u short=%llu\n", argv[i], (long long)nrec,
                   (unsigned long long)bases, ks.nrows, ks.nwide, ks.final_slots, (unsigned long long)ks.short_records);
            skh_keyset_free(&ks);
        } else printf("%s: rc=%d\n", argv[i], rc);
    }
    printf("stream bytes=%llu checksum=%llu\n", g_bytes, g_sum);
    return 0;
}

/* the device-side gzip decoder (sk_inflate.hip) is not part of the CPU builds: "not a file this path takes" */
struct sk_inflater;
int sk_ctx_device_(const sk_ctx *c) { (void)c; return 0; }
int sk_inflater_create(int device, struct sk_inflater **out) { (void)device; *out = 0; return -1; }
void sk_inflater_destroy(struct sk_inflater *f) { (void)f; }
uint64_t sk_inflate_gz_size(const uint8_t *gz, uint64_t n) { (void)gz; (void)n; return 0; }
int sk_inflate_gz(struct sk_inflater *f, const uint8_t *gz, uint64_t n, uint8_t *t, uint64_t cap, uint64_t *len, uint32_t *crc) { (void)f; (void)gz; (void)n; (void)t; (void)cap; (void)len; (void)crc; return -100; }
int sk_scan_device_packed(sk_ctx *c, const void *s, uint64_t n, uint32_t col) { (void)c; (void)s; (void)n; (void)col; return unreachable("sk_scan_device_packed"); }
int sk_batch_fill_packed(sk_batch *b, const void *s, uint64_t n, const uint32_t *r, uint32_t nr) { (void)b; (void)s; (void)n; (void)r; (void)nr; return unreachable("sk_batch_fill_packed"); }
