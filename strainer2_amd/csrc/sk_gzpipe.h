/* sk_gzpipe.h -- inflate on a helper thread: the consumer (the record parser) pulls decompressed pieces while
 * the next ones are being inflated, so one gzip file is decoded by two cores instead of one (inflate and
 * record parsing cost about the same per byte).  Built on sk_gzfast.h; pull-style:
 *
 *     skzp p;  if (skzp_open_threads(&p, path, threads) != SKZ_OK) ... zlib route ...
 *     while ((n = skzp_next(&p, &data)) > 0) parser_feed(ps, data, n);
 *     skzp_close(&p);                       // may be called early: stops the helper
 *
 * skzp_open_threads answers SKZ_NOT_GZIP / SKZ_OPEN exactly as skz_decode_file would, before any thread is started.
 */
#ifndef SK_GZPIPE_H
#define SK_GZPIPE_H
#include <pthread.h>
#include "sk_gzfast.h"
#include "sk_gzpar.h"

#define SKZP_NBUF 3

typedef struct {
    unsigned char *map; size_t map_len;
    pthread_t th; int started;
    pthread_mutex_t mu; pthread_cond_t cv;
    unsigned char *buf[SKZP_NBUF]; size_t len[SKZP_NBUF], cap[SKZP_NBUF];
    unsigned head, count;             /* filled pieces: buf[head], buf[head+1], ... */
    int done, cancel, rc, holding;    /* holding: the consumer still reads buf[head] */
    int threads; size_t seg_bytes;    /* > 1: the member is inflated by that many threads (sk_gzpar.h) */
} skzp;

static int skzp_sink(void *user, const unsigned char *data, size_t n)
{
    skzp *p = (skzp *)user;
    unsigned at;
    pthread_mutex_lock(&p->mu);
    while (p->count == SKZP_NBUF && !p->cancel) pthread_cond_wait(&p->cv, &p->mu);
    if (p->cancel) { pthread_mutex_unlock(&p->mu); return 1; }
    at = (p->head + p->count) % SKZP_NBUF;
    pthread_mutex_unlock(&p->mu);                      /* the slot is ours until count is raised */
    if (n > p->cap[at]) {
        free(p->buf[at]);
        p->buf[at] = (unsigned char *)malloc(n);
        p->cap[at] = p->buf[at] ? n : 0;
        if (!p->buf[at]) return 1;
    }
    memcpy(p->buf[at], data, n);
    p->len[at] = n;
    pthread_mutex_lock(&p->mu);
    p->count++;
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
    return 0;
}

static void *skzp_thread(void *arg)
{
    skzp *p = (skzp *)arg;
    const int rc = p->threads > 1 ? skzq_decode_memory(p->map, p->map_len, p->threads, p->seg_bytes, skzp_sink, p)
                                  : skz_decode_memory(p->map, p->map_len, skzp_sink, p);
    pthread_mutex_lock(&p->mu);
    p->rc = rc;
    p->done = 1;
    pthread_cond_broadcast(&p->cv);
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

/* threads: how many cores may inflate this one file (1: the helper thread alone).  SK_GZ_SEG overrides the
 * segment size of the parallel decoder (the tests use tiny ones). */
static int skzp_open_threads(skzp *p, const char *path, int threads)
{
    const int fd = open(path, O_RDONLY);
    struct stat st;
    memset(p, 0, sizeof *p);
    p->threads = threads;
    p->seg_bytes = getenv("SK_GZ_SEG") ? (size_t)strtoull(getenv("SK_GZ_SEG"), NULL, 10) : 0;
    if (fd < 0) return SKZ_OPEN;
    if (fstat(fd, &st) || !S_ISREG(st.st_mode) || st.st_size < 18) { close(fd); return SKZ_NOT_GZIP; }
    p->map = (unsigned char *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p->map == MAP_FAILED) { p->map = NULL; return SKZ_NOT_GZIP; }
    p->map_len = (size_t)st.st_size;
    if (skz_header(p->map, p->map_len) == 0) { munmap(p->map, p->map_len); p->map = NULL; return SKZ_NOT_GZIP; }
    madvise(p->map, p->map_len, MADV_SEQUENTIAL);
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv, NULL);
    if (pthread_create(&p->th, NULL, skzp_thread, p)) {
        pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv);
        munmap(p->map, p->map_len); p->map = NULL;
        return SKZ_NOT_GZIP;                           /* no helper thread: let the caller take its other route */
    }
    p->started = 1;
    return SKZ_OK;
}

/* next piece (valid until the following call), 0 at the end of the data */
static size_t skzp_next(skzp *p, const unsigned char **data)
{
    size_t n;
    pthread_mutex_lock(&p->mu);
    if (p->holding) {                                  /* give the previous piece's buffer back */
        p->head = (p->head + 1) % SKZP_NBUF;
        p->count--;
        p->holding = 0;
        pthread_cond_broadcast(&p->cv);
    }
    while (p->count == 0 && !p->done) pthread_cond_wait(&p->cv, &p->mu);
    if (p->count == 0) { pthread_mutex_unlock(&p->mu); return 0; }
    *data = p->buf[p->head];
    n = p->len[p->head];
    p->holding = 1;
    pthread_mutex_unlock(&p->mu);
    return n;
}

static void skzp_close(skzp *p)
{
    int i;
    if (!p->map) return;
    if (p->started) {
        pthread_mutex_lock(&p->mu);
        p->cancel = 1;
        pthread_cond_broadcast(&p->cv);
        pthread_mutex_unlock(&p->mu);
        pthread_join(p->th, NULL);
        pthread_mutex_destroy(&p->mu);
        pthread_cond_destroy(&p->cv);
    }
    for (i = 0; i < SKZP_NBUF; i++) free(p->buf[i]);
    munmap(p->map, p->map_len);
    memset(p, 0, sizeof *p);
}

#endif
