/* gzfast_check.c -- strainer2_amd/csrc/sk_gzfast.h against zlib: for every file named on the command line
 * the bytes delivered by skz_decode_file must equal what gzread delivers (also for files cut short, byte for
 * byte; for damaged data: equal where both have data).  Prints one line
 * per file with both rates.  --pipe: through sk_gzpipe.h's helper thread; --par T: that with T threads inflating
 * each member (sk_gzpar.h).  Built (also under ASan/UBSan) and run by tests/test_gzfast.py. */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>
#include "../../strainer2_amd/csrc/sk_gzpipe.h"

typedef struct { unsigned char *p; size_t n, cap; } buf;
static int collect(void *user, const unsigned char *d, size_t n)
{
    buf *b = (buf *)user;
    if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 4096; b->p = realloc(b->p, b->cap); }
    memcpy(b->p + b->n, d, n);
    b->n += n;
    return 0;
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* the same decode pulled through the helper-thread pipe (sk_gzpipe.h); with stop_after > 0 the consumer walks
 * away after that many bytes, which must stop the helper cleanly */
static int par_threads = 1;
static int pipe_decode(const char *path, buf *b, size_t stop_after)
{
    skzp p;
    const unsigned char *d;
    size_t n;
    int rc = skzp_open_threads(&p, path, par_threads);
    if (rc != SKZ_OK) return rc;
    while ((n = skzp_next(&p, &d)) > 0) {
        collect(b, d, n);
        if (stop_after && b->n >= stop_after) break;
    }
    pthread_mutex_lock(&p.mu);                        /* (the helper may still be running when we walked away) */
    rc = p.done ? p.rc : SKZ_STOPPED;
    pthread_mutex_unlock(&p.mu);
    skzp_close(&p);
    return rc;
}

int main(int argc, char **argv)
{
    int i, bad = 0, use_pipe = 0;
    if (argc > 1 && !strcmp(argv[1], "--pipe")) { use_pipe = 1; argv++; argc--; }
    /* --par T: the pipe with T inflating threads per member (sk_gzpar.h; segment size from SK_GZ_SEG) */
    if (argc > 2 && !strcmp(argv[1], "--par")) { use_pipe = 1; par_threads = atoi(argv[2]); argv += 2; argc -= 2; }
    for (i = 1; i < argc; i++) {
        buf a = {0}, b = {0};
        unsigned char *blk = malloc(1 << 20);
        gzFile g = gzopen(argv[i], "rb");
        int got, zerr = 0, zcut = 0, rc;
        double t0, t1, t2;
        if (!g) { printf("%s: cannot open\n", argv[i]); bad = 1; continue; }
        t0 = now();
        while ((got = gzread(g, blk, 1 << 20)) > 0) collect(&a, blk, (size_t)got);
        {   /* zlib reports a truncated or damaged stream through gzread < 0, gzerror or gzclose */
            int en = Z_OK;
            gzerror(g, &en);
            if (got < 0 || (en != Z_OK && en != Z_STREAM_END)) zerr = 1;
            zcut = en == Z_BUF_ERROR;                         /* the input ended early (as opposed to damaged data) */
            if (gzclose(g) != Z_OK) zerr = 1;
        }
        t1 = now();
        if (use_pipe) {
            buf c = {0};
            rc = pipe_decode(argv[i], &b, 0);
            if (rc != SKZ_NOT_GZIP && rc != SKZ_OPEN) {            /* and once more, abandoned half way */
                pipe_decode(argv[i], &c, b.n / 2 + 1);
                if (c.n > b.n || (c.n && memcmp(c.p, b.p, c.n))) { printf("%s: MISMATCH in the abandoned pipe run\n", argv[i]); bad = 1; }
            }
            free(c.p);
        } else rc = skz_decode_file(argv[i], collect, &b);
        t2 = now();
        if (rc == SKZ_NOT_GZIP) {
            printf("%s: not gzip (zlib passes %zu bytes through)\n", argv[i], a.n);
        } else if (rc == SKZ_OK && !zerr) {
            const int same = a.n == b.n && (a.n == 0 || !memcmp(a.p, b.p, a.n));
            printf("%s: %s %zu bytes; zlib %.0f MB/s, skz %.0f MB/s\n", argv[i], same ? "OK" : "MISMATCH", b.n,
                   a.n / (t1 - t0) / 1e6, b.n / (t2 - t1 + 1e-9) / 1e6);
            bad |= !same;
        } else {
            /* a damaged file: both must stop, and agree on the bytes both produced */
            const size_t m = a.n < b.n ? a.n : b.n;
            /* (a file cut short must give exactly zlib's bytes -- the reference parses them; after damaged data
             * zlib drops its current read call's output, and the reference's parser never returns) */
            const int same = (rc != SKZ_OK) && zerr && (m == 0 || !memcmp(a.p, b.p, m)) && (!zcut || a.n == b.n);
            printf("%s: damaged, zlib err=%d after %zu bytes, skz rc=%d after %zu bytes: %s\n", argv[i], zerr, a.n, rc, b.n,
                   same ? "OK" : "MISMATCH");
            bad |= !same;
        }
        free(a.p); free(b.p); free(blk);
    }
    if (par_threads > 1)
        printf("par: direct %llu gap %llu again %llu\n", (unsigned long long)skzq_stat_direct, (unsigned long long)skzq_stat_gap, (unsigned long long)skzq_stat_again);
    return bad;
}
