/* sk_alloc.h -- checked host allocation for the C sources that include it (after the system headers,
 * before anything that allocates): running out of host memory ends the process with a message instead of
 * a NULL dereference somewhere later (the reference dereferences the NULL). */
#ifndef SK_ALLOC_H
#define SK_ALLOC_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline void *sk_alloc_check(void *p, size_t n)
{
    if (!p && n) { fputs("libstrainer_kmer: out of host memory\n", stderr); abort(); }
    return p;
}
static inline void *sk_xmalloc(size_t n) { return sk_alloc_check(malloc(n), n); }
static inline void *sk_xcalloc(size_t a, size_t b) { return sk_alloc_check(calloc(a, b), a * b); }
static inline void *sk_xrealloc(void *q, size_t n) { return sk_alloc_check(realloc(q, n), n); }
static inline char *sk_xstrdup(const char *t) { return (char *)sk_alloc_check(strdup(t), 1); }
#define malloc(n)     sk_xmalloc(n)
#define calloc(a, b)  sk_xcalloc(a, b)
#define realloc(q, n) sk_xrealloc(q, n)
#define strdup(t)     sk_xstrdup(t)
#endif
