/* sk_internal.h -- entry points shared between the translation units of libstrainer_kmer.so; not part
 * of the C-ABI (include/strainer_kmer.h). */
#ifndef SK_INTERNAL_H
#define SK_INTERNAL_H
#include <stdint.h>
#include "../../include/strainer_kmer.h"
#ifdef __cplusplus
extern "C" {
#endif
/* record a message for sk_last_error and hand back `code` */
int sk_fail_(sk_ctx *ctx, int code, const char *fmt, ...);
int sk_ctx_device_(const sk_ctx *ctx);
/* column `col` of the resident counters, in the caller's row order, into a device buffer of nrows u32;
 * complete on return */
int sk_counts_rows_to_device_(sk_ctx *ctx, uint32_t col, uint32_t *d_out);
#ifdef __cplusplus
}
#endif
#endif
