/* sk_ctxjob.h -- open the device context on a helper thread while the host builds the key set: bringing
 * up the HIP runtime takes 0.25-3 s on the boxes measured, about as long as parsing a 5 Mbp strain. */
#ifndef SK_CTXJOB_H
#define SK_CTXJOB_H
#include <pthread.h>
#include "../../include/strainer_kmer.h"

typedef struct { pthread_t th; int device, rc, started; sk_ctx *ctx; } sk_ctxjob;

static void *sk_ctxjob_run(void *arg)
{
    sk_ctxjob *j = (sk_ctxjob *)arg;
    j->rc = sk_ctx_create(&j->ctx, j->device);
    return NULL;
}

static inline void sk_ctxjob_start(sk_ctxjob *j, int device)
{
    j->device = device; j->rc = SK_E_STATE; j->ctx = NULL;
    j->started = pthread_create(&j->th, NULL, sk_ctxjob_run, j) == 0;
}

/* the context (or NULL) and sk_ctx_create's status; created here if the thread could not be started */
static inline int sk_ctxjob_join(sk_ctxjob *j, sk_ctx **out)
{
    if (j->started) { pthread_join(j->th, NULL); j->started = 0; }
    else if (!j->ctx && j->rc == SK_E_STATE) j->rc = sk_ctx_create(&j->ctx, j->device);
    *out = j->ctx;
    j->ctx = NULL;
    return j->rc;
}
#endif
