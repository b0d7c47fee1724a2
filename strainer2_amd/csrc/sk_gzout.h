/* sk_gzout.h -- gzip output for strain_detect's hit list: text is collected in blocks of ~1 MiB, every block is
 * compressed on a pool thread as a gzip member of its own (zlib deflate), and the members of one file are
 * written in order.  A gzip file is any concatenation of members, so the DECOMPRESSED bytes are exactly what was
 * appended -- which is what the parity tests and the downstream script (gzip.open) see.  The reference writes
 * the same text line by line through gzprintf at level 9 on its only thread: 7.5 us per hit line; with many
 * hits (a strain that really is in the metagenome) that was the whole run time.
 *
 * One pool (skzo_pool) serves any number of files (skzo_file); a file is appended to by one thread at a time.
 * SK_GZ_LEVEL sets the deflate level.  Default 4 (round 4): at level 6 the compression of the hit lines was the LARGEST single
 * consumer of CPU time in a 32-strain pass -- 35 of 75 CPU-seconds per 100 Gbase (14 M lines, 0.7 GB of text; sixteen pool threads
 * at 2.2 s each, SK_SD_TIMING=1), on a box whose quota is sixteen CPUs and whose decode side wants them.  zlib on this text
 * (tab-separated counts + a 31-mer per line): level 1 78 MB/s at 0.269 of the size, level 4 63 MB/s at 0.244, level 6 11 MB/s at
 * 0.227, level 9 (the reference's gzprintf "wb9") 2.9 MB/s at 0.220.  The decompressed bytes are the same at every level.
 */
#ifndef SK_GZOUT_H
#define SK_GZOUT_H
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define SKZO_BLOCK (1u << 20)

typedef struct skzo_job {
    struct skzo_job *next;
    struct skzo_file *file;
    uint64_t seq;
    unsigned char *in; size_t in_len;
    unsigned char *out; size_t out_len;
} skzo_job;

typedef struct skzo_pool {
    pthread_t th[16]; int nth;
    pthread_mutex_t mu; pthread_cond_t cv_work, cv_room;
    skzo_job *head, *tail;              /* blocks waiting for a compressor */
    int outstanding, max_outstanding, quit, level;
} skzo_pool;

typedef struct skzo_file {
    skzo_pool *pool;
    FILE *fp;
    unsigned char *buf; size_t len;     /* block being filled */
    uint64_t next_submit, next_write;
    skzo_job *done;                     /* compressed members not yet written, ascending seq */
    pthread_mutex_t mu; pthread_cond_t cv;
    int error;
} skzo_file;

static void skzo_compress(skzo_job *j, int level)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    j->out_len = 0;
    j->out = (unsigned char *)malloc(compressBound((uLong)j->in_len) + 64);
    if (!j->out || deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;
    z.next_in = j->in; z.avail_in = (uInt)j->in_len;
    z.next_out = j->out; z.avail_out = (uInt)(compressBound((uLong)j->in_len) + 64);
    if (deflate(&z, Z_FINISH) == Z_STREAM_END) j->out_len = z.total_out;
    deflateEnd(&z);
}

/* hand the finished member to its file; write whatever is next in line */
static void skzo_deliver(skzo_job *j)
{
    skzo_file *f = j->file;
    skzo_job **pp;
    pthread_mutex_lock(&f->mu);
    for (pp = &f->done; *pp && (*pp)->seq < j->seq; pp = &(*pp)->next) {}
    j->next = *pp;
    *pp = j;
    while (f->done && f->done->seq == f->next_write) {
        skzo_job *d = f->done;
        f->done = d->next;
        if (d->out_len == 0 || fwrite(d->out, 1, d->out_len, f->fp) != d->out_len) f->error = 1;
        free(d->in); free(d->out); free(d);
        f->next_write++;
    }
    pthread_cond_broadcast(&f->cv);
    pthread_mutex_unlock(&f->mu);
}

static void *skzo_worker(void *arg)
{
    skzo_pool *p = (skzo_pool *)arg;
    pthread_setname_np(pthread_self(), "sk-gzout");
    for (;;) {
        skzo_job *j;
        pthread_mutex_lock(&p->mu);
        while (!p->head && !p->quit) pthread_cond_wait(&p->cv_work, &p->mu);
        if (!p->head) { pthread_mutex_unlock(&p->mu); return NULL; }
        j = p->head;
        p->head = j->next;
        if (!p->head) p->tail = NULL;
        pthread_mutex_unlock(&p->mu);
        skzo_compress(j, p->level);
        skzo_deliver(j);
        pthread_mutex_lock(&p->mu);
        p->outstanding--;
        pthread_cond_broadcast(&p->cv_room);
        pthread_mutex_unlock(&p->mu);
    }
}

static void skzo_pool_start(skzo_pool *p, int nthreads)
{
    const char *lv = getenv("SK_GZ_LEVEL");
    int i;
    memset(p, 0, sizeof *p);
    p->level = lv ? atoi(lv) : 4;
    if (p->level < 0 || p->level > 9) p->level = 4;
    if (nthreads > 16) nthreads = 16;
    if (nthreads < 1) nthreads = 1;
    p->max_outstanding = 4 * nthreads;
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->cv_work, NULL);
    pthread_cond_init(&p->cv_room, NULL);
    for (i = 0; i < nthreads; i++)
        if (pthread_create(&p->th[p->nth], NULL, skzo_worker, p) == 0) p->nth++;
}

static void skzo_pool_stop(skzo_pool *p)
{
    int i;
    pthread_mutex_lock(&p->mu);
    p->quit = 1;
    pthread_cond_broadcast(&p->cv_work);
    pthread_mutex_unlock(&p->mu);
    for (i = 0; i < p->nth; i++) pthread_join(p->th[i], NULL);
    pthread_mutex_destroy(&p->mu);
    pthread_cond_destroy(&p->cv_work);
    pthread_cond_destroy(&p->cv_room);
}

static skzo_file *skzo_open(skzo_pool *pool, const char *path)
{
    skzo_file *f = (skzo_file *)calloc(1, sizeof *f);
    if (!f) return NULL;
    f->fp = fopen(path, "wb");
    f->buf = (unsigned char *)malloc(SKZO_BLOCK + 4096);
    if (!f->fp || !f->buf) { if (f->fp) fclose(f->fp); free(f->buf); free(f); return NULL; }
    f->pool = pool;
    pthread_mutex_init(&f->mu, NULL);
    pthread_cond_init(&f->cv, NULL);
    return f;
}

/* send the current block off (it becomes one gzip member) */
static void skzo_submit(skzo_file *f)
{
    skzo_pool *p = f->pool;
    skzo_job *j = (skzo_job *)calloc(1, sizeof *j);
    unsigned char *fresh = (unsigned char *)malloc(SKZO_BLOCK + 4096);
    if (!j || !fresh) { free(j); free(fresh); f->error = 1; f->len = 0; return; }
    j->file = f; j->seq = f->next_submit++; j->in = f->buf; j->in_len = f->len;
    f->buf = fresh; f->len = 0;
    if (p->nth == 0) {                                  /* no pool thread could be started: compress here */
        skzo_compress(j, p->level);
        skzo_deliver(j);
        return;
    }
    pthread_mutex_lock(&p->mu);
    while (p->outstanding >= p->max_outstanding) pthread_cond_wait(&p->cv_room, &p->mu);
    p->outstanding++;
    if (p->tail) p->tail->next = j; else p->head = j;
    p->tail = j;
    pthread_cond_signal(&p->cv_work);
    pthread_mutex_unlock(&p->mu);
}

/* room for `need` more bytes (need <= 4096) in the current block; returns where to write them */
static inline unsigned char *skzo_reserve(skzo_file *f, size_t need)
{
    if (f->len + need > SKZO_BLOCK + 4096 || f->len >= SKZO_BLOCK) skzo_submit(f);
    return f->buf + f->len;
}

static void skzo_append(skzo_file *f, const void *data, size_t n)
{
    const unsigned char *d = (const unsigned char *)data;
    while (n) {
        size_t room, take;
        if (f->len >= SKZO_BLOCK) skzo_submit(f);
        room = SKZO_BLOCK - f->len;
        take = n < room ? n : room;
        memcpy(f->buf + f->len, d, take);
        f->len += take; d += take; n -= take;
    }
}

/* 0 on success */
static int skzo_close(skzo_file *f)
{
    int err;
    if (!f) return 0;
    if (f->len || f->next_submit == 0) skzo_submit(f);  /* (an empty file is one empty member) */
    pthread_mutex_lock(&f->mu);
    while (f->next_write != f->next_submit) pthread_cond_wait(&f->cv, &f->mu);
    pthread_mutex_unlock(&f->mu);
    err = f->error;
    if (fclose(f->fp)) err = 1;
    pthread_mutex_destroy(&f->mu);
    pthread_cond_destroy(&f->cv);
    free(f->buf);
    free(f);
    return err;
}

#endif
