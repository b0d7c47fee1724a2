/* sk_internal.h -- entry points shared between the translation units of libstrainer_kmer.so; not part
 * of the C-ABI (include/strainer_kmer.h). */
#ifndef SK_INTERNAL_H
#define SK_INTERNAL_H
#include <stdint.h>
#include <stdio.h>
#include "../../include/strainer_kmer.h"
#ifdef __cplusplus
extern "C" {
#endif
#define SK_E_UNSUPPORTED -100
/* record a message for sk_last_error and hand back `code` */
int sk_fail_(sk_ctx *ctx, int code, const char *fmt, ...);
int sk_ctx_device_(const sk_ctx *ctx);
/* column `col` of the resident counters, in the caller's row order, into a device buffer of nrows u32;
 * complete on return */
int sk_counts_rows_to_device_(sk_ctx *ctx, uint32_t col, uint32_t *d_out);
/* coverage/depth accumulator of sk_host_cov.c, fed by strain_detect when steps 3 and 4 are fused */
typedef struct skc_acc skc_acc;
skc_acc *skc_create(int64_t min_hits);
void skc_destroy(skc_acc *a);
int skc_add_hit(skc_acc *a, const char *metagenome, int64_t hits_pe1, int64_t hits_pe2, uint32_t row);
int skc_add_trailer(skc_acc *a, const char *metagenome, const char *name, int64_t value);
int skc_report(skc_acc *a, sk_ctx *ctx, const char *hits_file_name, FILE *out, FILE *err);
/* strain_detect's key set built on the device (sk_host.c): SK_OK, the ctx holds the table and *ks the keys in row order; SK_E_UNSUPPORTED: a
 * strain with letters other than A/C/G/T/N (byte-string keys) or without a key -- the caller builds it on the host */
int skh_keyset_build_on_device(skh_keyset *ks, sk_ctx *ctx, const char *path, uint32_t ncols, uint32_t col0_value);
/* ... whose host list of keys (ks->packed) is filled in on demand: the keys of these rows now */
int skh_keyset_fetch_keys(skh_keyset *ks, sk_ctx *ctx, const uint32_t *rows, uint32_t n);
/* gzip on the device (sk_inflate.hip; experimental, SK_GPU_INFLATE=1).  SK_E_UNSUPPORTED: not a file this path takes, or a check failed --
 * the caller decodes it on the host as before. */
typedef struct sk_inflater sk_inflater;
int  sk_inflater_create(int device, sk_inflater **out);
void sk_inflater_destroy(sk_inflater *f);
const char *sk_inflater_error(const sk_inflater *f);
uint64_t sk_inflate_gz_size(const uint8_t *gz, uint64_t n);
int  sk_inflate_gz(sk_inflater *f, const uint8_t *gz, uint64_t n, uint8_t *host_text, uint64_t host_cap, uint64_t *text_len, uint32_t *trailer_crc);
#ifdef __cplusplus
}
#endif
#endif
