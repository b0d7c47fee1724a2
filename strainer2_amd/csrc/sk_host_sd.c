/* sk_host_sd.c -- host layer of the strain_detect program (reference src/strain_detect.c).
 *
 * What runs where:
 *   device  every k-mer lookup: the per-read tallies (all hits / informative hits) and the log of
 *           informative hits come from sk_tally_batch (sk_scan_grid in TALLY mode); the informative
 *           k-mer list (-a) is itself looked up as a batch of 31-byte records; the -g background
 *           counts use the ordinary counting scan.
 *   host    file grammar, the reference's sequential read-pair bookkeeping (src/strain_detect.c:443-626,
 *           including its carry-over of tallies and of the PE1 copy across reads shorter than k), the
 *           background threshold arithmetic (:160-240), gz output, messages, exit codes.
 *
 * Known, documented divergences: reads of 10,000+ bases are handled (the reference
 * overflows fixed buffers); a blank line in the -B list is reported, not a crash; after a malformed
 * FASTQ record in a PE2 file the file is treated as ended.
 */
#define _GNU_SOURCE
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/resource.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include "../../include/strainer_kmer.h"
#include "sk_alloc.h"
#include "sk_common.h"
#include "sk_parser.h"
#include "sk_internal.h"
#include "sk_ctxjob.h"
#include "sk_gzpipe.h"
#include "sk_cpus.h"
#include "sk_gzout.h"


enum { SD_TYPE = 0, SD_BACKGROUND = 5, SD_INFORMATIVE = 2, SD_PLAIN = 1, SD_NCOLS = 6 };
enum { SD_SE = 0, SD_PE = 1, SD_PEI = 2, SD_UNKNOWN = -1 };
#define SD_BATCH_BYTES (32u << 20)
#define SD_PARSERS_MAX 16                 /* parser threads on one file, at most (SK_PARSE_THREADS) */
#define SD_PARSERS_MAPPED 12              /* ... on a mapped plain file with the whole thread budget behind it, by default (8 until round 4: the main thread was what the pass waited for) */
/* ONE process, ONE decode pipeline, several devices (strain_detect -S with SK_DEVICES=n; BASELINE configs[4]: 256 strains on the 8
 * GPUs of a node).  The strains are dealt to the devices in groups of SK_UNION_MAX (one union table per group: group g lives on
 * logical device g mod n); a decoded chunk of the metagenome is uploaded ONCE PER DEVICE from the same page-locked buffer (every
 * GPU has its own PCIe link: the copies run side by side), every device scans it against its own strains, and the results come
 * back to the one host that replays each strain's bookkeeping.  Against `torchrun`-style one-process-per-GPU with the strains
 * dealt to ranks (which still works: RANK/WORLD_SIZE) the metagenome is inflated and parsed once instead of once per GPU.
 * New relative to the reference, which is one process, one strain (src/strain_detect.c:263-384).
 * phys[d] = the HIP device behind logical device d: SK_DEVICES=n -> SK_DEVICE + 0..n-1; SK_DEVICES=0,0,0 -> as listed (several
 * logical devices on one card: how the path is tested on a one-GPU box). */
#define SD_MAX_DEV 8
static struct { int n; int phys[SD_MAX_DEV]; sk_ctx *ctx[SD_MAX_DEV]; } sd_dev = { 1, {0}, {NULL} };
static uint32_t sd_group = SK_UNION_MAX;                  /* strains per union table (SK_SD_GROUP: smaller groups, for tests of the dealing) */
static int sd_dev_of_strain(uint32_t s) { return (int)((s / sd_group) % (uint32_t)sd_dev.n); }
/* SK_SD_CHUNK_BYTES: smaller chunks (tests: chunk boundaries between mates, carried state across chunks) or bigger ones (up to 1 GiB) */
/* strain_detect never shows the order of the table's rows (hits are printed read by read with the k-mer's text; the
 * trailer counts rows), so the 0.2 s replay of BIO_hash's slot order (sk_host.c) is left out: rows in strain order.
 * SK_SD_REF_ROW_ORDER=1 replays it all the same (tests compare the two). */
#define SD_ROW_ORDER (getenv("SK_SD_REF_ROW_ORDER") ? SK_REF_TABLE_SLOTS : SK_ROWS_IN_STRAIN_ORDER)

static size_t sd_chunk_bytes(void)
{
    const char *e = getenv("SK_SD_CHUNK_BYTES");
    const long v = e ? atol(e) : 0;
    return v >= 64 && v <= (1l << 30) ? (size_t)v : SD_BATCH_BYTES;
}

/* ---------------------------------------------------------------------------------------------
 * program state: one per strain (table on the device, type column, output, replay state)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    sk_ctx     *ctx;
    int         ctx_rc;        /* status of a context opened in the background (0 = fine or not tried) */
    int         table_on_device;/* the key set was built on the device (skh_keyset_build_on_device): the table is loaded already */
    skh_keyset  ks;
    uint32_t   *type;          /* host copy of the type column */
    FILE       *out, *err;
    /* quantification */
    skzo_file  *zo;            /* -o: gzip members compressed on the output pool (sk_gzout.h) */
    unsigned    genome_inf;    /* informative rows after -a / -g */
    int         h1, i1, h2, i2;            /* tallies carried from read to read (src/strain_detect.c:444-454,497-500) */
    uint32_t   *copy_rows; uint32_t copy_n, copy_cap;   /* informative rows of the PE1 read last copied (:451) */
    skc_acc    *cov; char *cov_path, *o_path;           /* --coverage-depth: step 4 of the workflow, fed at emission */
    sk_hit     *hitbuf; uint64_t hitcap;   /* landing area of the hit log */
    uint32_t   *tallybuf; uint32_t tallycap;
    int         job_rc;                    /* result of this strain's part of a pool job */
} sd_prog;

/* ---------------------------------------------------------------------------------------------
 * one metagenome file as a stream of chunks: a decode thread parses ahead, the main thread tallies
 * each chunk against every strain and walks its records
 * ------------------------------------------------------------------------------------------- */
typedef struct sd_chunk {
    uint8_t  *buf; uint64_t blen, bcap;      /* record stream: the records of length >= k, '\n' after each */
    int       pinned;                        /* buf belongs to the pool of page-locked buffers */
    int       packed;                        /* buf holds the stream in sk_pack_stream's form (blen, pstart: of the byte stream it was packed from) */
    uint32_t *pstart, *prec; uint32_t np, pcap;   /* those records: offset in buf, index in len[] */
    uint64_t *len; uint32_t nrec, rcap;      /* every record, length as the reference sees it */
    int       last, end_kind; size_t end_len;/* last chunk of the file: how the parser ended */
    /* per strain, filled by sd_tally_chunk: the records that hit the strain at all, ascending -- their tallies and their
     * informative rows (CSR).  Nothing here is proportional to records x strains: against a metagenome nearly every
     * (read, strain) pair is a blank */
    struct sd_sp *sp;
    uint32_t  nstrains;
    /* what the scan of this chunk brought back, strain by strain (round 4: owned by the chunk, so that the per-strain work on it
     * -- sort, spread, replay -- runs on the lanes while the main thread is on to the next chunk): strain s's hit records are
     * res_recs[res_off[s] .. res_off[s + 1]), its log entries res_hits[res_hoff[s] .. res_hoff[s + 1]) */
    sk_tally_rec *res_recs; sk_hit *res_hits; uint64_t *res_off, *res_hoff;
    int       refs;                          /* holders: the stream (queue, current chunk), every lane job that names it */
} sd_chunk;
typedef struct sd_sp { uint32_t n; uint32_t *rec, *all, *inf, *hbeg /* n + 1 */, *rows; } sd_sp;

typedef struct {
    const char *path;
    int gz_threads;                 /* threads inflating this file (sk_gzpar.h when > 1) */
    gzFile      g;
    pthread_t   th; int started;
    pthread_mutex_t mu; pthread_cond_t cv;
    sd_chunk   *q[3]; int qn, cancel;        /* decoded chunks waiting for the main thread */
    sd_chunk   *cur;                         /* producer: chunk under construction (serial decode) */
    size_t      chunk_bytes;
    /* several parser threads on ONE file (sd_decode_thread): the text is cut into segments at guessed record starts,
     * every segment is parsed on its own and CHECKED to end between two records, chunks are queued in segment order */
    int         par;                         /* parser threads (0/1 = the decode thread parses itself) */
    pthread_mutex_t pmu; pthread_cond_t pcv;
    struct sd_seg *segq[20]; int segn, seg_done;   /* segments waiting for a parser thread; no more will come */
    uint64_t    next_push;                   /* sequence number of the segment whose chunks go to q next */
    size_t      carry_last_len;              /* length of the last record handed out so far (for an END_STALE ending) */
    int         split_failed;                /* a segment did not end between two records */
    int         cancel_segments;             /* the file is over for the reader (truncated record): later segments are dropped */
    int         in_mode;                     /* SD_IN_*: how the parser threads get at the bytes of a mapped plain file's segments */
    /* consumer */
    sd_chunk   *c; uint32_t ci;
    int         eof, end_kind; size_t end_len;
    /* two device batches in turn: while the chunk in one is scanned, the next chunk of the queue is uploaded into the other;
     * with several devices (sd_dev) every device has its own pair -- one decoded chunk goes up to all of them */
    sk_batch   *bat[2][SD_MAX_DEV]; int bcur;
    sd_chunk   *pre;                         /* the queued chunk whose bytes are already in bat[bcur ^ 1] */
    sd_chunk   *ahead;                       /* ... and whose scan has been launched as well (solo streams: see sd_launch_ahead) */
    int         solo;                        /* the only stream that uses the tables now (SE, PEI): its next chunk may be scanned while this one is replayed */
} sd_stream;

/* Page-locked chunk buffers, recycled.  The upload of a chunk from ordinary memory goes through the runtime's
 * staging path, which collapses (measured: 17 -> 1.7 GB/s) as soon as other threads of the process fault pages
 * in at a high rate -- which is what the decode threads do when one .gz file is inflated by eight of them. */
#define SD_PIN_MAX 32                    /* (1 GiB at the default chunk size: every parser thread holds one, the queue three, the device pair two) */
static struct {
    pthread_mutex_t mu;
    sk_ctx *ctx; size_t bytes;
    void *idle[SD_PIN_MAX]; int nidle, total;
} sd_pin = { PTHREAD_MUTEX_INITIALIZER, NULL, 0, {NULL}, 0, 0 };

static unsigned long n_unpinned_chunks;     /* chunks whose text had to live in ordinary memory (their upload goes through the runtime's staging path) */
static void *sd_pin_get(void)
{
    void *q = NULL;
    pthread_mutex_lock(&sd_pin.mu);
    if (sd_pin.nidle) q = sd_pin.idle[--sd_pin.nidle];
    else if (sd_pin.ctx && sd_pin.total < SD_PIN_MAX && sk_pinned_alloc(sd_pin.ctx, &q, sd_pin.bytes) == SK_OK) sd_pin.total++;
    else q = NULL;
    pthread_mutex_unlock(&sd_pin.mu);
    return q;
}
static void sd_pin_put(void *q)
{
    pthread_mutex_lock(&sd_pin.mu);
    sd_pin.idle[sd_pin.nidle++] = q;                     /* (at most `total` <= SD_PIN_MAX are ever out) */
    pthread_mutex_unlock(&sd_pin.mu);
}
static void sd_pin_open(sk_ctx *ctx, size_t bytes) { sd_pin.ctx = ctx; sd_pin.bytes = bytes; }
static void sd_pin_close(void)
{
    while (sd_pin.nidle) { sk_pinned_free(sd_pin.ctx, sd_pin.idle[--sd_pin.nidle]); sd_pin.total--; }
    sd_pin.ctx = NULL;
}

static void sd_unmap_later(void *p, size_t n);
static double now_s(void);
static double t_pw_seg, t_pw_parse, t_pw_turn, t_pw_push;   /* SK_SD_TIMING: the parser threads' time, summed over them */
static double cpu_parsers, cpu_readers, cpu_lanes;          /* SK_SD_TIMING: CPU time (user + system) of those threads, summed when they end */
static double thread_cpu_s(void)
{
    struct rusage ru;
    if (getrusage(RUSAGE_THREAD, &ru) != 0) return 0.0;
    return (double)ru.ru_utime.tv_sec + 1e-6 * (double)ru.ru_utime.tv_usec + (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec;
}
static pthread_mutex_t t_pw_mu = PTHREAD_MUTEX_INITIALIZER;

static sd_chunk *chunk_new(void)
{
    sd_chunk *c = (sd_chunk *)calloc(1, sizeof *c);
    if (c) c->refs = 1;
    return c;
}
static void chunk_ref(sd_chunk *c) { if (c) __atomic_add_fetch(&c->refs, 1, __ATOMIC_RELAXED); }

/* drop one holder's claim; the last one frees */
static void chunk_free(sd_chunk *c)
{
    uint32_t s;
    if (!c) return;
    if (__atomic_sub_fetch(&c->refs, 1, __ATOMIC_ACQ_REL) > 0) return;
    for (s = 0; c->sp && s < c->nstrains; s++) { free(c->sp[s].rec); free(c->sp[s].rows); }   /* (rec, all, inf, hbeg: one block) */
    free(c->sp);
    free(c->res_recs); free(c->res_hits); free(c->res_off);                                    /* (res_off, res_hoff: one block) */
    if (c->pinned) sd_pin_put(c->buf); else free(c->buf);
    free(c->pstart); free(c->prec); free(c->len);
    free(c);
}

static void stream_push(sd_stream *st, sd_chunk *c)
{
    pthread_mutex_lock(&st->mu);
    while (st->qn == 3 && !st->cancel) pthread_cond_wait(&st->cv, &st->mu);
    if (st->cancel) { pthread_mutex_unlock(&st->mu); chunk_free(c); return; }
    st->q[st->qn++] = c;
    pthread_cond_broadcast(&st->cv);
    pthread_mutex_unlock(&st->mu);
}

/* chunk builder of one parser: where finished chunks go depends on who parses */
enum { SD_IN_POPULATE = 0, SD_IN_PREAD = 1, SD_IN_MAPPED = 2 };
typedef struct sd_seg { unsigned char *buf; size_t n, cap; uint64_t seq; int is_last, borrowed; int fd; uint64_t off; } sd_seg;   /* borrowed: buf points into the mapped file */
typedef struct {
    sd_stream *st;
    sd_chunk  *cur;                          /* chunk under construction */
    sd_chunk **done; int ndone, dcap;        /* parallel parse: finished chunks of this segment, pushed later in order */
    int        direct;                       /* serial decode: finished chunks go straight to the stream's queue */
} sd_builder;

/* SK_SD_PACK=1: the finished chunk packed by the thread that built it -- 6 bytes per 16 bases go up instead of 16 (sk_pack.h; the scan's
 * first phase then only copies).  Out of one pool buffer into another, the first given back at once; a chunk with a byte for the
 * byte-string kernel, or one outside the pool, stays as it is.  OFF by default: the pass is bound by these threads' CPU time as much
 * as by the link, and the packing is theirs to pay. */
static int sd_pack_on(void) { const char *e = getenv("SK_SD_PACK"); return e && e[0] == '1'; }

static void chunk_pack(sd_chunk *c)
{
    void *pk;
    int odd = 0;
    if (!c || !c->pinned || !c->np || c->packed || c->blen > sd_chunk_bytes() || !sd_pack_on()) return;
    if ((pk = sd_pin_get()) == NULL) return;
    if (sk_pack_stream(c->buf, c->blen, pk, &odd) != SK_OK || odd) { sd_pin_put(pk); return; }
    sd_pin_put(c->buf);
    c->buf = (uint8_t *)pk;
    c->packed = 1;
}

static void builder_finish_chunk(sd_builder *b)
{
    chunk_pack(b->cur);
    if (b->direct) stream_push(b->st, b->cur);
    else {
        if (b->ndone == b->dcap) { b->dcap = b->dcap ? b->dcap * 2 : 4; b->done = (sd_chunk **)realloc(b->done, (size_t)b->dcap * sizeof *b->done); }
        b->done[b->ndone++] = b->cur;
    }
    b->cur = NULL;
}

/* parser callback: one record */
static int sd_on_record(void *user, char *seq, size_t len)
{
    sd_builder *b = (sd_builder *)user;
    sd_stream *st = b->st;
    sd_chunk *c = b->cur;
    if (st->cancel) return 1;
    if (c && c->nrec && (c->blen + len + 1 > st->chunk_bytes || c->nrec >= (1u << 22))) { builder_finish_chunk(b); c = NULL; }
    if (!c) c = b->cur = chunk_new();
    if (c->nrec == c->rcap) {
        c->rcap = c->rcap ? c->rcap * 2 : 1u << 16;
        c->len = (uint64_t *)realloc(c->len, (size_t)c->rcap * sizeof *c->len);
    }
    c->len[c->nrec] = len;
    if (len >= SK_K) {
        if (c->np == c->pcap) {
            c->pcap = c->pcap ? c->pcap * 2 : 1u << 16;
            c->pstart = (uint32_t *)realloc(c->pstart, (size_t)c->pcap * sizeof *c->pstart);
            c->prec = (uint32_t *)realloc(c->prec, (size_t)c->pcap * sizeof *c->prec);
        }
        if (!c->buf && len + 1 <= st->chunk_bytes && (c->buf = (uint8_t *)sd_pin_get()) != NULL) {
            c->pinned = 1;                                 /* the usual case: a whole chunk's worth, page-locked */
            c->bcap = st->chunk_bytes;
        }
        if (!c->buf) __atomic_add_fetch(&n_unpinned_chunks, 1, __ATOMIC_RELAXED);     /* (SK_SD_TIMING: the pool was empty or the record too long) */
        if (c->blen + len + 1 > c->bcap) {                 /* a record never straddles chunks: grow instead */
            uint64_t cap = c->bcap ? c->bcap : 1u << 20;
            while (cap < c->blen + len + 1) cap *= 2;
            c->buf = (uint8_t *)realloc(c->buf, cap);      /* (never a pool buffer: those hold chunk_bytes, the most a chunk takes) */
            c->bcap = cap;
        }
        c->pstart[c->np] = (uint32_t)c->blen;
        c->prec[c->np++] = c->nrec;
        memcpy(c->buf + c->blen, seq, len);
        c->blen += len;
        c->buf[c->blen++] = '\n';
    }
    c->nrec++;
    return 0;
}

/* ---- several parser threads on one file ------------------------------------------------------------------------
 * The decode thread (below) only cuts: it gathers the decoded text into segments of about a chunk's worth, each ending
 * at the first line start in its last 64 KiB that looks like a record start (parser_guess_start), and hands them out.
 * A parser thread parses a segment as a file of its own -- the same state machine, its own chunk builder -- and then
 * CHECKS that it stands between two records (parser_between_records): the first segment starts at the file's start,
 * every other at the checked end of the one before, so if all checks hold the segments' records are the file's records
 * (src/kseq.h:171-211 reads from the top; nothing here changes what a record is).  A failed check fails the run with
 * a message (SK_NO_SPLIT=1 turns the cutting off) -- it is never papered over.  Chunks reach the queue in segment
 * order: the replay of the reference's read-after-read bookkeeping (src/strain_detect.c:443-626) sees the file's order. */
/* bytes per pread of a parser thread: SK_READ_BLOCK (64 bytes -- tests -- .. 32 MiB), default 2 MiB */
static size_t sd_read_block(void)
{
    const char *e = getenv("SK_READ_BLOCK");
    const long long v = e ? atoll(e) : 0;
    return v >= 64 && v <= (32 << 20) ? (size_t)v : (size_t)2 << 20;
}

static void *sd_parse_worker(void *arg)
{
    sd_stream *st = (sd_stream *)arg;
    unsigned char *rbuf = NULL;
    size_t rcap = 0;
    pthread_setname_np(pthread_self(), "sk-parse");
    for (;;) {
        sd_seg *sg;
        sd_builder b;
        parser ps;
        int ok = 1, i;
        double w0 = now_s(), w1, w2, w3;
        pthread_mutex_lock(&st->pmu);
        while (st->segn == 0 && !st->seg_done) pthread_cond_wait(&st->pcv, &st->pmu);
        if (st->segn == 0) {
            pthread_mutex_unlock(&st->pmu);
            pthread_mutex_lock(&t_pw_mu); cpu_parsers += thread_cpu_s(); pthread_mutex_unlock(&t_pw_mu);
            free(rbuf);
            return NULL;
        }
        sg = st->segq[0];
        for (i = 1; i < st->segn; i++) st->segq[i - 1] = st->segq[i];
        st->segn--;
        pthread_cond_broadcast(&st->pcv);
        pthread_mutex_unlock(&st->pmu);

        w1 = now_s();
        memset(&b, 0, sizeof b);
        b.st = st;
        parser_init(&ps, sd_on_record, &b);
        if (sg->borrowed && !st->cancel) {
            /* A segment of a plain file: it points into the mapping of the whole file.  Three ways to get at its bytes (round 4):
             *   SD_IN_POPULATE (default)  the segment's pages are put into the address space by ONE call, parsed where they lie, and
             *       taken out again by one call: no copy, no trap per 64 KiB of text, and the address space's lock is only ever
             *       taken for READING -- BASELINE configs[4]'s share pass 3.5-3.6 s, 40 CPU-seconds;
             *   SD_IN_PREAD               read into a buffer of this thread's, in blocks that stay in the core's cache between the
             *       copy and the parse: 3.9-4.1 s on the same box, 48 CPU-seconds -- a sampling of the program counter says 59 % of
             *       a parser thread's time is the kernel's copy (tools/probes/sigprof_preload.c); also where the kernel has no
             *       MADV_POPULATE_READ (before 5.14);
             *   SD_IN_MAPPED              parsed out of the mapping, every page faulted in where it is first touched: 5.1-6.7 s.  Not
             *       the faults themselves: the file's munmap at its end tears 2.7 M page-table entries down under the lock for
             *       WRITING, and the next file's faults wait a third of a second for it, file after file.
             * (profiles/r04_cfg5_pread.json, r04_cfg5_read_blocks.json, r04_cfg5_populate.json; SK_SD_INPUT=populate|pread|mapped) */
            int mode = st->in_mode;
            const uintptr_t pa = (uintptr_t)sg->buf & ~(uintptr_t)4095u, pe = ((uintptr_t)sg->buf + sg->n + 4095u) & ~(uintptr_t)4095u;
            if (mode == SD_IN_POPULATE) {
#ifdef MADV_POPULATE_READ
                if (madvise((void *)pa, (size_t)(pe - pa), MADV_POPULATE_READ) != 0 && errno == EINVAL) st->in_mode = mode = SD_IN_PREAD;   /* (an older kernel: from now on the copy) */
#else
                st->in_mode = mode = SD_IN_PREAD;
#endif
            }
            if (mode == SD_IN_PREAD && sg->fd >= 0) {
                size_t done = 0;
                if (!rbuf) { rcap = sd_read_block(); rbuf = (unsigned char *)malloc(rcap); }
                while (done < sg->n && ps.state != P_STOP && !st->cancel) {
                    const size_t want = sg->n - done < rcap ? sg->n - done : rcap;
                    size_t have = 0;
                    while (rbuf && have < want) {
                        const ssize_t r = pread(sg->fd, rbuf + have, want - have, (off_t)(sg->off + done + have));
                        if (r < 0 && errno == EINTR) continue;
                        if (r <= 0) break;
                        have += (size_t)r;
                    }
                    if (rbuf && have == want) parser_feed(&ps, rbuf, want); else parser_feed(&ps, sg->buf + done, want);    /* (a failed read: out of the mapping) */
                    done += want;
                }
            } else parser_feed(&ps, sg->buf, sg->n);
            if (mode == SD_IN_POPULATE) {               /* (whole pages inside the segment only: a neighbour may still be reading the ones at its ends) */
                const uintptr_t da = ((uintptr_t)sg->buf + 4095u) & ~(uintptr_t)4095u, de = ((uintptr_t)sg->buf + sg->n) & ~(uintptr_t)4095u;
                if (de > da) (void)madvise((void *)da, (size_t)(de - da), MADV_DONTNEED);
            }
        } else
        if (!st->cancel) parser_feed(&ps, sg->buf, sg->n);
        if (!sg->is_last && ps.state != P_STOP && !parser_between_records(&ps)) ok = 0;
        if (ps.state != P_STOP || ps.end_kind != SKP_END_NONE) { /* (stopped by cancel or by a truncated record: the ending stands) */ }
        parser_eof(&ps);
        if (b.cur && (sg->is_last || b.cur->nrec)) builder_finish_chunk(&b);
        else if (b.cur) { chunk_free(b.cur); b.cur = NULL; }

        w2 = now_s();
        pthread_mutex_lock(&st->pmu);                      /* in segment order */
        while (st->next_push != sg->seq && !st->cancel) pthread_cond_wait(&st->pcv, &st->pmu);
        pthread_mutex_unlock(&st->pmu);
        w3 = now_s();
        if (!ok) st->split_failed = 1;
        {
            size_t end_len = ps.end_len;
            if (ps.nrecords) st->carry_last_len = ps.last_len;
            else if (ps.end_kind == SKP_END_STALE) end_len = st->carry_last_len;       /* "the previous record" lies in an earlier segment */
            if (!st->cancel_segments && (sg->is_last || ps.end_kind == SKP_END_TRUNC || st->split_failed)) {  /* the file ends here (a truncated record ends it for the reference too) */
                sd_chunk *c;
                if (b.ndone == 0) { b.cur = chunk_new(); builder_finish_chunk(&b); }
                c = b.done[b.ndone - 1];
                c->last = 1;
                c->end_kind = ps.end_kind;
                c->end_len = end_len;
            }
        }
        {
            const int dropped = st->cancel_segments;            /* an earlier segment ended the file: this one's records do not exist */
            const int file_over = !dropped && b.ndone && b.done[b.ndone - 1]->last;      /* (read before the chunks change hands) */
            for (i = 0; i < b.ndone; i++) { if (dropped) chunk_free(b.done[i]); else stream_push(st, b.done[i]); }
            pthread_mutex_lock(&st->pmu);
            st->next_push++;
            if (file_over) st->cancel_segments = 1;
            pthread_cond_broadcast(&st->pcv);
            pthread_mutex_unlock(&st->pmu);
        }
        pthread_mutex_lock(&t_pw_mu);
        t_pw_seg += w1 - w0; t_pw_parse += w2 - w1; t_pw_turn += w3 - w2; t_pw_push += now_s() - w3;
        pthread_mutex_unlock(&t_pw_mu);
        free(b.done);
        parser_free(&ps);
        if (!sg->borrowed) free(sg->buf);
        free(sg);
    }
}

/* hand a segment to the parser threads (blocks while all of them are busy and a few wait) */
static void sd_seg_dispatch(sd_stream *st, sd_seg *sg)
{
    pthread_mutex_lock(&st->pmu);
    while ((st->segn == (int)(sizeof st->segq / sizeof st->segq[0]) || st->segn > st->par) && !st->cancel) pthread_cond_wait(&st->pcv, &st->pmu);
    if (st->cancel) { pthread_mutex_unlock(&st->pmu); if (!sg->borrowed) free(sg->buf); free(sg); return; }
    st->segq[st->segn++] = sg;
    pthread_cond_broadcast(&st->pcv);
    pthread_mutex_unlock(&st->pmu);
}

static void *sd_decode_thread(void *arg)
{
    enum { BLK = 1 << 20 };
    sd_stream *st = (sd_stream *)arg;
    const size_t TAIL = st->chunk_bytes / 2 < (64u << 10) ? (st->chunk_bytes / 2 < 256 ? 256 : st->chunk_bytes / 2) : (64u << 10);   /* text wanted behind a cut */
    unsigned char *blk = (unsigned char *)malloc(BLK);
    int got;
    skzp zp;
    const int own = !getenv("SK_ZLIB") && skzp_open_threads(&zp, st->path, st->gz_threads) == SKZ_OK;
    /* plain text is parsed straight out of the mapped file (zlib's pass-through mode copies every byte once more: 4.9 GB/s of
     * FASTA against what the parser itself does); anything that cannot be mapped, or SK_ZLIB=1: gzread */
    const unsigned char *map = NULL;
    size_t mlen = 0;
    int map_fd = -1;
    if (!own && !getenv("SK_ZLIB")) {
        const int fd = open(st->path, O_RDONLY);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            void *m = mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { map = (const unsigned char *)m; mlen = (size_t)sb.st_size; madvise(m, mlen, MADV_SEQUENTIAL); }
        }
        if (fd >= 0 && map) map_fd = fd; else if (fd >= 0) close(fd);       /* (kept open for the parser threads' pread, should they go that way) */
        if (map && mlen >= 2 && map[0] == 0x1f && map[1] == 0x8b) { munmap((void *)map, mlen); map = NULL; }   /* (gzip after all: zlib) */
    }
    if (st->par > 1 && (own || map)) {
        /* a .gz file inflated by several threads delivers text faster than one thread parses it (measured, 3 Gbase of FASTA,
         * 16 inflate threads: waiting for the decode side 0.73 s with one parser, 0.47 s with four), and so does a mapped
         * plain file when many strains wait for it (32 strains x 10 Gbase of FASTA: the one parser thread, 3 GB/s, was what the
         * whole pass waited for).  Cut the text into segments for the parser threads -- copied out of the inflate stream, or
         * simply pointing into the map: */
        pthread_t wk[SD_PARSERS_MAX];
        int nw = 0, i;
        uint64_t seq = 0;
        size_t scan_from;
        const size_t want = st->chunk_bytes;              /* about a chunk's worth of text per segment */
        sd_seg *sg = NULL;
        {   /* a mapped file costs nothing to read: with a whole thread budget behind one file, eight parsers (an inflating
             * file keeps its four: the inflate threads need the CPUs) */
            const int npar = map && !getenv("SK_PARSE_THREADS") && st->gz_threads >= 16 ? SD_PARSERS_MAPPED : st->par;
            for (i = 0; i < npar && i < SD_PARSERS_MAX; i++) if (pthread_create(&wk[nw], NULL, sd_parse_worker, st) == 0) nw++;
        }
        if (map) {
            size_t at = 0;
            while (at < mlen && nw) {
                size_t cut = mlen;
                int over;
                pthread_mutex_lock(&st->pmu);
                over = st->cancel_segments;
                pthread_mutex_unlock(&st->pmu);
                if (st->cancel || over) break;
                if (mlen - at > want + TAIL) cut = at + (size_t)parser_guess_start(map + at, mlen - at, want, 1);
                sg = (sd_seg *)calloc(1, sizeof *sg);
                sg->buf = (unsigned char *)map + at;
                sg->n = cut - at;
                sg->borrowed = 1;
                sg->fd = map_fd;
                sg->off = at;
                sg->seq = seq++;
                sg->is_last = cut == mlen;
                sd_seg_dispatch(st, sg);
                at = cut;
            }
            sg = NULL;
        } else {
        sg = (sd_seg *)calloc(1, sizeof *sg);
        sg->cap = want + (want >> 2) + BLK;
        sg->buf = (unsigned char *)malloc(sg->cap);
        scan_from = want;
        for (;;) {
            const unsigned char *data = NULL;
            size_t n = 0;
            int over;
            pthread_mutex_lock(&st->pmu);
            over = st->cancel_segments;
            pthread_mutex_unlock(&st->pmu);
            if (st->cancel || over || nw == 0) break;
            if (own) n = skzp_next(&zp, &data);
            else { got = gzread(st->g, blk, BLK); n = got > 0 ? (size_t)got : 0; data = blk; }
            if (n == 0) break;
            if (sg->n + n > sg->cap) { sg->cap = (sg->n + n) * 2; sg->buf = (unsigned char *)realloc(sg->buf, sg->cap); }
            memcpy(sg->buf + sg->n, data, n);
            sg->n += n;
            if (sg->n >= want + TAIL) {
                /* the first likely record start behind a chunk's worth of text (looked for in what has not been looked at
                 * yet; the guess wants two more lines of text behind it); none so far (very long records): keep gathering */
                const uint64_t cut = parser_guess_start(sg->buf, sg->n, scan_from, 0);
                if (cut >= sg->n) scan_from = sg->n - TAIL;
                else {
                    scan_from = want;
                    sd_seg *nx = (sd_seg *)calloc(1, sizeof *nx);
                    nx->cap = want + (want >> 2) + BLK;
                    if (nx->cap < sg->n - cut) nx->cap = (sg->n - cut) * 2;
                    nx->buf = (unsigned char *)malloc(nx->cap);
                    nx->n = sg->n - (size_t)cut;
                    memcpy(nx->buf, sg->buf + cut, nx->n);
                    sg->n = (size_t)cut;
                    sg->seq = seq++;
                    sd_seg_dispatch(st, sg);
                    sg = nx;
                }
            }
        }
        sg->seq = seq++;
        sg->is_last = 1;
        if (nw) sd_seg_dispatch(st, sg); else { free(sg->buf); free(sg); }
        }
        pthread_mutex_lock(&st->pmu);
        st->seg_done = 1;
        pthread_cond_broadcast(&st->pcv);
        pthread_mutex_unlock(&st->pmu);
        for (i = 0; i < nw; i++) pthread_join(wk[i], NULL);
        if (nw == 0) {                                     /* no thread could be started: an empty, failed file rather than a hang */
            sd_chunk *c = chunk_new();
            c->last = 1; c->end_kind = SKP_END_RESET;
            st->split_failed = 1;
            stream_push(st, c);
        }
    } else {
        /* this thread parses: gzip through the library's own inflate on a helper thread (sk_gzpipe.h); plain files, or
         * SK_ZLIB=1, through zlib */
        parser ps;
        sd_builder b;
        sd_chunk *c;
        memset(&b, 0, sizeof b);
        b.st = st; b.direct = 1;
        parser_init(&ps, sd_on_record, &b);
        if (own) {
            const unsigned char *data;
            size_t n;
            while (ps.state != P_STOP && !st->cancel && (n = skzp_next(&zp, &data)) > 0) parser_feed(&ps, data, n);
        } else {
            if (map) {
                size_t at = 0;
                while (ps.state != P_STOP && !st->cancel && at < mlen) {
                    const size_t n = mlen - at < (4u << 20) ? mlen - at : (4u << 20);
                    parser_feed(&ps, map + at, n);
                    at += n;
                }
            } else
                while (ps.state != P_STOP && !st->cancel && (got = gzread(st->g, blk, BLK)) > 0) parser_feed(&ps, blk, (size_t)got);
        }
        parser_eof(&ps);
        c = b.cur ? b.cur : chunk_new();
        b.cur = NULL;
        c->last = 1;
        c->end_kind = ps.end_kind;
        c->end_len = ps.end_len;
        stream_push(st, c);
        parser_free(&ps);
    }
    if (own) skzp_close(&zp);
    if (map) sd_unmap_later((void *)map, mlen);
    if (map_fd >= 0) close(map_fd);
    free(blk);
    pthread_mutex_lock(&t_pw_mu); cpu_readers += thread_cpu_s(); pthread_mutex_unlock(&t_pw_mu);
    return NULL;
}

/* how many threads may inflate each of the `nfiles` files read at the same time: SK_GZ_THREADS, or the host
 * thread budget (SK_THREADS, default min(16, usable CPUs)) shared out -- used when it leaves three or more per
 * file, below that the one helper thread of sk_gzpipe.h does as well */
static int sd_gz_threads(int nfiles)
{
    const long ncpu = sk_cpu_budget();
    const int budget = getenv("SK_THREADS") ? atoi(getenv("SK_THREADS")) : (int)(ncpu < 1 ? 1 : ncpu > 16 ? 16 : ncpu);
    int per = budget / nfiles;
    if (getenv("SK_GZ_THREADS")) per = atoi(getenv("SK_GZ_THREADS"));
    return per >= 3 ? (per > 16 ? 16 : per) : 1;
}

/* SK_E_OPEN if the file cannot be opened (nothing started) */
static int stream_open(sd_stream *st, const char *path, int gz_threads)
{
    memset(st, 0, sizeof *st);
    st->path = path;
    st->gz_threads = gz_threads;
    st->chunk_bytes = sd_chunk_bytes();
    {   /* SK_SD_INPUT=populate|pread|mapped (SK_SD_MAPPED=1: the round-3 way, kept for the A/B) */
        const char *e = getenv("SK_SD_INPUT");
        st->in_mode = e && !strcmp(e, "pread") ? SD_IN_PREAD : (e && !strcmp(e, "mapped")) || getenv("SK_SD_MAPPED") ? SD_IN_MAPPED : SD_IN_POPULATE;
    }
    {   /* parser threads for this file: when the budget leaves several threads per file (the case of one or two big
         * metagenomes), a quarter of them parse; SK_PARSE_THREADS sets the number, SK_NO_SPLIT=1 means one */
        const char *e = getenv("SK_PARSE_THREADS");
        st->par = e ? atoi(e) : (gz_threads >= 8 ? 4 : gz_threads >= 3 ? 2 : 1);
        if (getenv("SK_NO_SPLIT") || st->par < 2) st->par = 1;
        if (st->par > SD_PARSERS_MAX) st->par = SD_PARSERS_MAX;
    }
    pthread_mutex_init(&st->pmu, NULL);
    pthread_cond_init(&st->pcv, NULL);
    st->g = gzopen(path, "r");
    if (!st->g) return SK_E_OPEN;
    gzbuffer(st->g, 1 << 18);
    pthread_mutex_init(&st->mu, NULL);
    pthread_cond_init(&st->cv, NULL);
    if (pthread_create(&st->th, NULL, sd_decode_thread, st)) { gzclose(st->g); st->g = NULL; return SK_E_NOMEM; }
    st->started = 1;
    return SK_OK;
}

static void sd_drain_scans(void);

/* Device batches outlive the file they were made for: a `-B` list opens and closes a stream per line, and making and freeing a
 * batch (two device buffers, a stream, an event) per file cost more than scanning a small file.  A closed stream's batches wait
 * here, per device, for the next stream; sd_run frees them at its end. */
static struct { sk_batch *b[SD_MAX_DEV][4]; int n[SD_MAX_DEV]; } sd_bcache;
static int sd_batch_get(int d, sk_batch **out)
{
    if (sd_bcache.n[d] > 0) { *out = sd_bcache.b[d][--sd_bcache.n[d]]; return SK_OK; }
    return sk_batch_create(sd_dev.ctx[d], out);
}
static void sd_batch_put(int d, sk_batch *b)
{
    if (!b) return;
    if (sd_bcache.n[d] < 4 && sk_batch_sync(b) == SK_OK) sd_bcache.b[d][sd_bcache.n[d]++] = b;   /* (sync: an upload in flight must land before the chunks go) */
    else sk_batch_destroy(b);
}
static void sd_batch_cache_close(void)
{
    int d;
    for (d = 0; d < SD_MAX_DEV; d++) while (sd_bcache.n[d] > 0) sk_batch_destroy(sd_bcache.b[d][--sd_bcache.n[d]]);
}

/* Unmapping a plain file that was parsed out of its mapping (10 GB: 0.2 s of page-table work) happens behind the next file's scan,
 * on a thread of its own; one at a time, the last one joined by sd_run. */
static struct { pthread_t th; int live; void *p; size_t n; } sd_unmapper;
static pthread_mutex_t sd_unmapper_mu = PTHREAD_MUTEX_INITIALIZER;
static void *sd_unmap_thread(void *arg) { (void)arg; munmap(sd_unmapper.p, sd_unmapper.n); return NULL; }
static void sd_unmap_wait(void)
{
    pthread_mutex_lock(&sd_unmapper_mu);
    if (sd_unmapper.live) { pthread_join(sd_unmapper.th, NULL); sd_unmapper.live = 0; }
    pthread_mutex_unlock(&sd_unmapper_mu);
}
static void sd_unmap_later(void *p, size_t n)
{
    sd_unmap_wait();
    pthread_mutex_lock(&sd_unmapper_mu);
    sd_unmapper.p = p; sd_unmapper.n = n;
    if (pthread_create(&sd_unmapper.th, NULL, sd_unmap_thread, NULL) == 0) sd_unmapper.live = 1;
    else munmap(p, n);
    pthread_mutex_unlock(&sd_unmapper_mu);
}

static void stream_close(sd_stream *st)
{
    int i;
    if (!st->g) return;
    if (st->ahead) sd_drain_scans();                       /* scans started ahead on a batch that is about to go */
    if (st->started) {
        pthread_mutex_lock(&st->mu);
        st->cancel = 1;
        pthread_cond_broadcast(&st->cv);
        pthread_mutex_unlock(&st->mu);
        pthread_mutex_lock(&st->pmu);                      /* parser threads waiting for a segment or for their turn */
        pthread_cond_broadcast(&st->pcv);
        pthread_mutex_unlock(&st->pmu);
        pthread_join(st->th, NULL);
        pthread_mutex_destroy(&st->mu);
        pthread_cond_destroy(&st->cv);
        pthread_mutex_destroy(&st->pmu);
        pthread_cond_destroy(&st->pcv);
    }
    for (i = 0; i < SD_MAX_DEV; i++) {
        sd_batch_put(i, st->bat[0][i]);                 /* (waits for an upload in flight before the chunks go) */
        sd_batch_put(i, st->bat[1][i]);
    }
    for (i = 0; i < st->qn; i++) chunk_free(st->q[i]);
    chunk_free(st->c);
    chunk_free(st->cur);
    gzclose(st->g);
    memset(st, 0, sizeof *st);
}

/* SK_SD_TIMING=1: where the wall clock went, on stderr at exit */
static double t_wait, t_tally, t_setup, t_fill, t_launch, t_post, t_close, t_lens, t_replay, t_sopen, t_sclose, t_uclose, t_cfree;
static double t_open_ctx, t_open_load, t_open_flags;     /* thread time summed over the strains' worker threads (SK_SD_TIMING) */
static pthread_mutex_t t_open_mu = PTHREAD_MUTEX_INITIALIZER;
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

/* The strains are independent once a chunk is on the device: the per-strain work -- collecting and spreading a
 * chunk's tallies, replaying a run (and with it the gz compression of each strain's own output), closing a strain
 * -- is dealt out strain by strain to a few threads.  pool_run(pool, n, fn, arg) calls fn(arg, s) for every s < n,
 * the calling thread being one of the workers, and returns when all are done. */
typedef void (*sd_job_fn)(void *arg, uint32_t s);
typedef struct {
    pthread_t th[16]; int nth;
    pthread_mutex_t mu; pthread_cond_t cv_work, cv_done;
    unsigned long gen; int quit;
    sd_job_fn fn; void *arg; uint32_t ns;
    uint64_t ticket;                 /* (job number << 32) | next item: an item is claimed by a CAS that also proves the job is still the same */
    uint32_t done;
} sd_pool;

/* Items of job `job` are claimed through ONE word that holds the job's number beside the item counter: a worker that comes late
 * out of the previous job cannot take an item of the next one (nor read that job's fn/arg/ns half written) -- its CAS fails on
 * the job number.  fn, arg and ns are written before the release store that opens the job and read after an acquire load of the
 * ticket that shows the job's number.  (ThreadSanitizer found the old form -- a bare counter, ns read without order -- with five
 * strains on four threads.) */
static void pool_drain(sd_pool *pl, uint32_t job, uint32_t ns, sd_job_fn fn, void *arg)     /* (the job's own ns/fn/arg, read under the mutex) */
{
    uint32_t mine = 0;
    for (;;) {
        uint64_t t = __atomic_load_n(&pl->ticket, __ATOMIC_ACQUIRE);
        if ((uint32_t)(t >> 32) != job || (uint32_t)t >= ns) break;
        if (!__atomic_compare_exchange_n(&pl->ticket, &t, t + 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) continue;
        fn(arg, (uint32_t)t);
        mine++;
    }
    if (!mine) return;
    pthread_mutex_lock(&pl->mu);
    pl->done += mine;
    if (pl->done == pl->ns) pthread_cond_broadcast(&pl->cv_done);
    pthread_mutex_unlock(&pl->mu);
}

static void *pool_worker_sd(void *arg)
{
    sd_pool *pl = (sd_pool *)arg;
    unsigned long seen = 0;
    pthread_setname_np(pthread_self(), "sk-pool");
    for (;;) {
        pthread_mutex_lock(&pl->mu);
        while (pl->gen == seen && !pl->quit) pthread_cond_wait(&pl->cv_work, &pl->mu);
        if (pl->quit) { pthread_mutex_unlock(&pl->mu); return NULL; }
        seen = pl->gen;
        {
            const uint32_t ns = pl->ns;
            const sd_job_fn fn = pl->fn;
            void *const arg = pl->arg;
            pthread_mutex_unlock(&pl->mu);
            pool_drain(pl, (uint32_t)seen, ns, fn, arg);
        }
    }
}

static void pool_start(sd_pool *pl, uint32_t ns)
{
    int want = getenv("SK_THREADS") ? atoi(getenv("SK_THREADS")) : 8, i;
    memset(pl, 0, sizeof *pl);
    if (want > 16) want = 16;
    if ((uint32_t)want > ns) want = (int)ns;
    pthread_mutex_init(&pl->mu, NULL);
    pthread_cond_init(&pl->cv_work, NULL);
    pthread_cond_init(&pl->cv_done, NULL);
    for (i = 0; i + 1 < want; i++)                       /* the calling thread is one of the workers */
        if (pthread_create(&pl->th[pl->nth], NULL, pool_worker_sd, pl) == 0) pl->nth++;
}

static void pool_stop(sd_pool *pl)
{
    int i;
    pthread_mutex_lock(&pl->mu);
    pl->quit = 1;
    pthread_cond_broadcast(&pl->cv_work);
    pthread_mutex_unlock(&pl->mu);
    for (i = 0; i < pl->nth; i++) pthread_join(pl->th[i], NULL);
    pthread_mutex_destroy(&pl->mu);
    pthread_cond_destroy(&pl->cv_work);
    pthread_cond_destroy(&pl->cv_done);
}

static void pool_run(sd_pool *pl, uint32_t ns, sd_job_fn fn, void *arg)
{
    if (!pl || pl->nth == 0 || ns < 2) {
        uint32_t s;
        for (s = 0; s < ns; s++) fn(arg, s);
        return;
    }
    uint32_t job;
    pthread_mutex_lock(&pl->mu);                         /* (every item of the previous job is done: nobody reads fn/arg/ns now) */
    pl->fn = fn; pl->arg = arg; pl->ns = ns;
    pl->done = 0;
    pl->gen++;
    job = (uint32_t)pl->gen;
    __atomic_store_n(&pl->ticket, (uint64_t)job << 32, __ATOMIC_RELEASE);
    pthread_cond_broadcast(&pl->cv_work);
    pthread_mutex_unlock(&pl->mu);
    pool_drain(pl, job, ns, fn, arg);
    pthread_mutex_lock(&pl->mu);
    while (pl->done != pl->ns) pthread_cond_wait(&pl->cv_done, &pl->mu);
    pthread_mutex_unlock(&pl->mu);
}

static int tally_rec_cmp(const void *a, const void *b)
{
    const sk_tally_rec *x = (const sk_tally_rec *)a, *y = (const sk_tally_rec *)b;
    return x->rec < y->rec ? -1 : x->rec > y->rec;
}

static int hit_cmp(const void *a, const void *b)
{
    const sk_hit *x = (const sk_hit *)a, *y = (const sk_hit *)b;
    return x->pos < y->pos ? -1 : x->pos > y->pos;
}

/* One strain's share of a chunk's scan results (c->res_*: collected and dealt out by the main thread) -> the per-record tallies
 * and per-record lists of informative rows, in window order (c->sp[s]). */
static void tally_one(sd_prog *p, sd_chunk *c, uint32_t s)
{
    uint64_t nh = 0, h = 0;
    uint32_t n = 0, k;
    sd_sp *sp = &c->sp[s];
    memset(sp, 0, sizeof *sp);
    if (c->np == 0 || !c->res_off) return;
    /* only the pieces that hit this strain at all came back (compacted on the device): with many strains against one
     * metagenome nearly every (read, strain) pair is a blank */
    sk_tally_rec *sparse = c->res_recs + c->res_off[s];
    sk_hit *hits = c->res_hits + c->res_hoff[s];
    const uint64_t nsp = c->res_off[s + 1] - c->res_off[s];
    uint64_t e;
    nh = c->res_hoff[s + 1] - c->res_hoff[s];
    (void)p;
    /* in the order of the pieces (they come back unordered): a sort when they are few, a sweep over a piece-indexed
     * array when most pieces hit (one strain, reads from that strain) */
    if (nsp * 8 > c->np) {
        uint32_t *by = (uint32_t *)calloc((size_t)c->np, 2 * sizeof(uint32_t));
        uint64_t w = 0;
        for (e = 0; e < nsp; e++) { by[2 * (size_t)sparse[e].rec] = sparse[e].all; by[2 * (size_t)sparse[e].rec + 1] = sparse[e].inf; }
        for (k = 0; k < c->np; k++) if (by[2 * (size_t)k]) { sparse[w].rec = k; sparse[w].all = by[2 * (size_t)k]; sparse[w].inf = by[2 * (size_t)k + 1]; w++; }
        free(by);
    } else qsort(sparse, (size_t)nsp, sizeof *sparse, tally_rec_cmp);
    qsort(hits, (size_t)nh, sizeof(sk_hit), hit_cmp);
    sp->rec = (uint32_t *)malloc(((size_t)nsp * 4 + 2) * sizeof(uint32_t));
    sp->all = sp->rec + nsp; sp->inf = sp->all + nsp; sp->hbeg = sp->inf + nsp;
    sp->rows = (uint32_t *)malloc(((size_t)nh + 1) * sizeof(uint32_t));
    for (e = 0; e < nsp; ) {                              /* (a record cut into pieces: the pieces' windows add up) */
        const uint32_t rec = c->prec[sparse[e].rec];
        uint32_t last = sparse[e].rec, all = 0, inf = 0, end;
        while (e < nsp && c->prec[sparse[e].rec] == rec) { all += sparse[e].all; inf += sparse[e].inf; last = sparse[e].rec; e++; }
        while (last + 1 < c->np && c->prec[last + 1] == rec) last++;              /* the record's last piece: its rows end where the next record begins */
        end = last + 1 < c->np ? c->pstart[last + 1] : 0xFFFFFFFFu;
        sp->rec[sp->n] = rec; sp->all[sp->n] = all; sp->inf[sp->n] = inf; sp->hbeg[sp->n] = n;
        while (h < nh && hits[h].pos < end) sp->rows[n++] = hits[h++].row;
        sp->n++;
    }
    sp->hbeg[sp->n] = n;
}

/* ---------------------------------------------------------------------------------------------
 * Lanes (round 4): the per-strain work behind a chunk's scan, off the main thread.
 *
 * Round 3's main thread did, chunk after chunk and one thing after the other: upload, launch, collect, then WAIT for the pool to
 * sort and spread the chunk's results strain by strain, then walk the read lengths, then WAIT for the pool to replay the
 * reference's read-after-read bookkeeping (and compress 32 outputs) -- 3.5-4.3 s of a 7-8 s pass over 100 Gbase x 32 strains,
 * beside 2.7-3.3 s of waiting for the decode side, which in turn waited for it (profiles/r03_cfg5_share.json).  The
 * reference's bookkeeping (src/strain_detect.c:443-626) is sequential per STRAIN, not across strains: strain s owns its carried
 * tallies, its copy of the last PE1 read's rows, its output.  So: L lane threads, strain s belongs to lane s mod L, and the main
 * thread only POSTS jobs -- "spread chunk c" (tally_one), "replay this run of read pairs" -- which every lane works off in the
 * order posted, for its own strains.  A lane's strains see the chunks and runs in file order, which is all the bookkeeping needs;
 * the main thread goes on to the next chunk's collection at once.  Chunks are reference-counted: a job holds the chunks it names.
 * Back-pressure: a lane's ring holds SD_RING jobs; the main thread waits when the slowest lane is that far behind.  At the end of
 * a file the main thread waits for the lanes to run dry (the trailer lines follow the last hit line), and reads the strains'
 * error states there.
 * ------------------------------------------------------------------------------------------- */
#define SD_LANES_MAX 16
#define SD_RING 8
enum { SD_JOB_TALLY, SD_JOB_REPLAY, SD_JOB_SYNC };
typedef struct sd_job {
    int kind;
    sd_chunk *ca, *cb;                         /* TALLY: ca; REPLAY: the run's chunks (cb may be NULL) */
    const char *f1;
    uint32_t a0, b0, astep, n, first_valid;
    int have_copy;
    int pending;                               /* lanes that still have it (atomic) */
} sd_job;
struct sd_lanes;
typedef struct sd_lane {
    pthread_t th; int started;
    pthread_mutex_t mu; pthread_cond_t cv_in, cv_out;
    sd_job *ring[SD_RING]; uint32_t head, count; int quit;
    struct sd_lanes *all; uint32_t id;
    double busy;                               /* (SK_SD_TIMING) */
} sd_lane;
typedef struct sd_lanes {
    sd_lane lane[SD_LANES_MAX]; uint32_t n, ns;
    sd_prog *p;
    pthread_mutex_t smu; pthread_cond_t scv; uint64_t syncs_done;     /* SYNC jobs completed */
} sd_lanes;
static sd_lanes sd_ln;
/* device calls of the main thread (uploads, launches, collections) and the one a lane can make (a hit line's k-mer fetched on the
 * spot, emit_rows) never run at the same time: a context takes one caller at a time */
static pthread_mutex_t sd_dev_mu = PTHREAD_MUTEX_INITIALIZER;
static double t_lane_busy, t_lane_backpressure, t_lane_drain;

static void sd_replay_run(sd_prog *p, uint32_t s, const char *f1, const sd_chunk *ca, uint32_t a0, const sd_chunk *cb, uint32_t b0,
                          uint32_t astep, uint32_t n, int have_copy_in, uint32_t first_valid);

static void lane_job_done(sd_lanes *L, sd_job *j)
{
    if (__atomic_sub_fetch(&j->pending, 1, __ATOMIC_ACQ_REL) > 0) return;
    if (j->kind == SD_JOB_SYNC) {
        pthread_mutex_lock(&L->smu);
        L->syncs_done++;
        pthread_cond_broadcast(&L->scv);
        pthread_mutex_unlock(&L->smu);
    }
    chunk_free(j->ca);
    chunk_free(j->cb);
    free(j);
}

static void *lane_main(void *arg)
{
    sd_lane *ln = (sd_lane *)arg;
    sd_lanes *L = ln->all;
    pthread_setname_np(pthread_self(), "sk-lane");
    for (;;) {
        sd_job *j;
        uint32_t s;
        pthread_mutex_lock(&ln->mu);
        while (ln->count == 0 && !ln->quit) pthread_cond_wait(&ln->cv_in, &ln->mu);
        if (ln->count == 0) {
            pthread_mutex_unlock(&ln->mu);
            pthread_mutex_lock(&t_pw_mu); cpu_lanes += thread_cpu_s(); pthread_mutex_unlock(&t_pw_mu);
            return NULL;
        }
        j = ln->ring[ln->head];
        pthread_mutex_unlock(&ln->mu);
        {
            const double t0 = now_s();
            for (s = ln->id; s < L->ns; s += L->n) {
                sd_prog *p = &L->p[s];
                if (j->kind == SD_JOB_TALLY) tally_one(p, j->ca, s);
                else if (j->kind == SD_JOB_REPLAY && p->job_rc == SK_OK)
                    sd_replay_run(p, s, j->f1, j->ca, j->a0, j->cb, j->b0, j->astep, j->n, j->have_copy, j->first_valid);
            }
            ln->busy += now_s() - t0;
        }
        pthread_mutex_lock(&ln->mu);                  /* (the job leaves the ring only now: "ring empty" means "nothing in work") */
        ln->head = (ln->head + 1u) % SD_RING;
        ln->count--;
        pthread_cond_signal(&ln->cv_out);
        pthread_mutex_unlock(&ln->mu);
        lane_job_done(L, j);
    }
}

static void lanes_start(sd_prog *p, uint32_t ns)
{
    sd_lanes *L = &sd_ln;
    uint32_t want = getenv("SK_SD_LANES") ? (uint32_t)atoi(getenv("SK_SD_LANES")) : getenv("SK_THREADS") ? (uint32_t)atoi(getenv("SK_THREADS")) : 8u, i;
    memset(L, 0, sizeof *L);
    if (want < 1) want = 1;
    if (want > SD_LANES_MAX) want = SD_LANES_MAX;
    if (want > ns) want = ns;
    L->n = want; L->ns = ns; L->p = p;
    pthread_mutex_init(&L->smu, NULL);
    pthread_cond_init(&L->scv, NULL);
    for (i = 0; i < L->n; i++) {
        sd_lane *ln = &L->lane[i];
        ln->all = L; ln->id = i;
        pthread_mutex_init(&ln->mu, NULL);
        pthread_cond_init(&ln->cv_in, NULL);
        pthread_cond_init(&ln->cv_out, NULL);
        ln->started = pthread_create(&ln->th, NULL, lane_main, ln) == 0;
    }
}

/* hand a job to every lane (it takes over the caller's claims on ca and cb: pass chunks already chunk_ref'ed) */
static void lanes_post(sd_job *j)
{
    sd_lanes *L = &sd_ln;
    uint32_t i;
    j->pending = (int)L->n;
    for (i = 0; i < L->n; i++) {
        sd_lane *ln = &L->lane[i];
        if (!ln->started) {                               /* (no thread could be made for this lane: its strains are done here) */
            uint32_t s;
            for (s = ln->id; s < L->ns; s += L->n) {
                if (j->kind == SD_JOB_TALLY) tally_one(&L->p[s], j->ca, s);
                else if (j->kind == SD_JOB_REPLAY && L->p[s].job_rc == SK_OK)
                    sd_replay_run(&L->p[s], s, j->f1, j->ca, j->a0, j->cb, j->b0, j->astep, j->n, j->have_copy, j->first_valid);
            }
            lane_job_done(L, j);
            continue;
        }
        pthread_mutex_lock(&ln->mu);
        if (ln->count == SD_RING) {
            const double t0 = now_s();
            while (ln->count == SD_RING) pthread_cond_wait(&ln->cv_out, &ln->mu);
            t_lane_backpressure += now_s() - t0;
        }
        ln->ring[(ln->head + ln->count) % SD_RING] = j;
        ln->count++;
        pthread_cond_signal(&ln->cv_in);
        pthread_mutex_unlock(&ln->mu);
    }
}

static sd_job *job_new(int kind, sd_chunk *ca, sd_chunk *cb)
{
    sd_job *j = (sd_job *)calloc(1, sizeof *j);
    j->kind = kind; j->ca = ca; j->cb = cb;
    chunk_ref(ca); chunk_ref(cb);
    return j;
}

/* wait until every lane has worked off everything posted so far */
static void lanes_drain(void)
{
    sd_lanes *L = &sd_ln;
    uint64_t want;
    if (!L->n) return;
    pthread_mutex_lock(&L->smu);
    want = L->syncs_done + 1;
    pthread_mutex_unlock(&L->smu);
    lanes_post(job_new(SD_JOB_SYNC, NULL, NULL));
    pthread_mutex_lock(&L->smu);
    while (L->syncs_done < want) pthread_cond_wait(&L->scv, &L->smu);
    pthread_mutex_unlock(&L->smu);
}

static void lanes_stop(void)
{
    sd_lanes *L = &sd_ln;
    uint32_t i;
    for (i = 0; i < L->n; i++) {
        sd_lane *ln = &L->lane[i];
        if (ln->started) {
            pthread_mutex_lock(&ln->mu);
            ln->quit = 1;
            pthread_cond_signal(&ln->cv_in);
            pthread_mutex_unlock(&ln->mu);
            pthread_join(ln->th, NULL);
        }
        t_lane_busy += ln->busy;
        pthread_mutex_destroy(&ln->mu);
        pthread_cond_destroy(&ln->cv_in);
        pthread_cond_destroy(&ln->cv_out);
    }
    if (L->n) { pthread_mutex_destroy(&L->smu); pthread_cond_destroy(&L->scv); }
    L->n = 0;
}

/* Many strains on one device: one union table per group of up to SK_UNION_MAX strains (sk_union_*), so that a batch is
 * scanned once per group instead of once per strain (BASELINE configs[4]: 32 strains per GPU).  Built by sd_run when the
 * strains are open and flagged; a strain the union cannot hold (byte-string keys, no text stage), SK_SD_NO_UNION=1, or a
 * batch of 64 MiB or more (one huge record) leave the member-by-member way below, which gives the same results. */
static struct { sk_union **u; uint32_t n; sk_tally_rec *recs; uint64_t recs_cap; sk_hit *hits; uint64_t *hcap; } sd_un;

/* every scan that may still be in flight (a stream closed early with a chunk launched ahead): wait before its batch is freed */
static sd_prog *sd_all_p; static uint32_t sd_all_ns;
static void sd_drain_scans(void)
{
    uint32_t g, s;
    for (g = 0; g < sd_un.n; g++) sk_union_sync(sd_un.u[g]);
    for (s = 0; !sd_un.n && s < sd_all_ns; s++) if (sd_all_p[s].ctx) sk_sync(sd_all_p[s].ctx);
}

static void sd_unions_close(void)
{
    uint32_t g;
    for (g = 0; g < sd_un.n; g++) sk_union_destroy(sd_un.u[g]);
    free(sd_un.u); free(sd_un.recs); free(sd_un.hits); free(sd_un.hcap);
    memset(&sd_un, 0, sizeof sd_un);
}

static void sd_unions_open(sd_prog *p, uint32_t ns)
{
    uint32_t g, ng = (ns + sd_group - 1) / sd_group;
    const double t0 = now_s();
    memset(&sd_un, 0, sizeof sd_un);
    if (ns < 2 || getenv("SK_SD_NO_UNION")) return;
    sd_un.u = (sk_union **)calloc(ng, sizeof *sd_un.u);
    sd_un.hcap = (uint64_t *)calloc(ng, sizeof *sd_un.hcap);
    for (g = 0; g < ng; g++) {
        sk_ctx *m[SK_UNION_MAX];
        const uint32_t a = g * sd_group, n = ns - a < sd_group ? ns - a : sd_group;
        uint32_t k;
        int rc;
        for (k = 0; k < n; k++) m[k] = p[a + k].ctx;
        rc = sk_union_create(m, n, SD_TYPE, SD_INFORMATIVE, &sd_un.u[g]);
        if (rc != SK_OK) {
            if (getenv("SK_SD_TIMING")) fprintf(p[0].err, "strain_detect: no union table (%s: %s): strain by strain\n", sk_strerror(rc), sk_last_error(m[0]));
            sd_un.n = g;
            sd_unions_close();
            return;
        }
        sd_un.n = g + 1;
        sd_un.hcap[g] = 1u << 18;
    }
    if (getenv("SK_SD_TIMING")) fprintf(p[0].err, "strain_detect timing: %u union table(s) for %u strains in %.2f s\n", ng, ns, now_s() - t0);
}

/* while the device scans the current chunk: the next chunk of the stream's queue (if the reader is ahead) goes up into the
 * stream's other batch -- the upload of a 32 MiB chunk takes about twice as long as its scan against a union table */
static void sd_prefetch(sd_stream *st)
{
    sd_chunk *n = NULL;
    const double t0 = now_s();
    int d, ok = 1;
    if (!st) return;
    pthread_mutex_lock(&st->mu);
    if (st->qn > 0) n = st->q[0];                          /* (only this thread takes chunks off the queue: n stays) */
    pthread_mutex_unlock(&st->mu);
    if (!n || n == st->pre || !n->np) return;
    for (d = 0; d < sd_dev.n && ok; d++) {
        sk_batch **b = &st->bat[st->bcur ^ 1][d];
        if (!*b && sd_batch_get(d, b) != SK_OK) { *b = NULL; ok = 0; break; }
        ok = (n->packed ? sk_batch_fill_packed(*b, n->buf, n->blen, n->pstart, n->np) : sk_batch_fill(*b, n->buf, n->blen, n->pstart, n->np)) == SK_OK;
    }
    if (ok) st->pre = n;                                   /* (all devices or none: a chunk that is not everywhere goes up again) */
    t_fill += now_s() - t0;
}

/* start the scans of one uploaded chunk: against every union table, or strain by strain (they overlap on the devices) */
static int sd_launch(sd_prog *p, uint32_t ns, sk_batch **batches)
{
    uint32_t g, s;
    int rc;
    if (sd_un.n) {
        for (g = 0; g < sd_un.n; g++)
            if ((rc = sk_union_tally_launch(sd_un.u[g], batches[sd_dev_of_strain(g * sd_group)], sd_un.hcap[g])) != SK_OK) return rc;
        return SK_OK;
    }
    for (s = 0; s < ns; s++) {
        if (p[s].hitcap == 0) { p[s].hitcap = 1u << 16; p[s].hitbuf = (sk_hit *)malloc((size_t)p[s].hitcap * sizeof(sk_hit)); }
        if ((rc = sk_tally_launch(p[s].ctx, batches[sd_dev_of_strain(s)], SD_TYPE, SD_INFORMATIVE, p[s].hitcap)) != SK_OK) return rc;
    }
    return SK_OK;
}

/* room for n more records and m more log entries in the chunk's own result arrays */
static int chunk_res_reserve(sd_chunk *c, uint64_t *rcap, uint64_t *hcap, uint64_t nrec_total, uint64_t nhit_total)
{
    if (nrec_total > *rcap || !c->res_recs) {
        *rcap = nrec_total + nrec_total / 2 + 256;
        c->res_recs = (sk_tally_rec *)realloc(c->res_recs, (size_t)*rcap * sizeof(sk_tally_rec));
    }
    if (nhit_total > *hcap || !c->res_hits) {
        *hcap = nhit_total + nhit_total / 2 + 256;
        c->res_hits = (sk_hit *)realloc(c->res_hits, (size_t)*hcap * sizeof(sk_hit));
    }
    return c->res_recs && c->res_hits ? SK_OK : SK_E_NOMEM;
}

/* The results of the chunk whose scans are in flight, collected and laid out strain by strain IN THE CHUNK (res_recs/res_off,
 * res_hits/res_hoff) -- from the union tables (every group's results dealt to its members by a counting sort), or member by
 * member.  After this the device side may be overwritten by the next launch, and the lanes take it from here. */
static int sd_collect(sd_prog *p, uint32_t ns, sk_batch **batches, sd_chunk *c)
{
    uint32_t g, s;
    int rc;
    uint64_t rcap = 0, hcap = 0, nr = 0, nhit = 0;
    c->res_off = (uint64_t *)calloc(2 * ((size_t)ns + 1), sizeof(uint64_t));
    if (!c->res_off) return SK_E_NOMEM;
    c->res_hoff = c->res_off + ns + 1;
    if (!sd_un.n) {
        for (s = 0; s < ns; s++) {
            uint64_t nsp = 0, nh = 0;
            if (p[s].tallycap < c->np || !p[s].tallybuf) {
                p[s].tallycap = c->np + c->np / 4 + 1024;
                p[s].tallybuf = (uint32_t *)realloc(p[s].tallybuf, (size_t)p[s].tallycap * sizeof(sk_tally_rec));
            }
            if ((rc = sk_tally_collect_sparse(p[s].ctx, (sk_tally_rec *)p[s].tallybuf, c->np, &nsp, p[s].hitbuf, &nh)) != SK_OK) return rc;
            if (nh > p[s].hitcap) {                           /* the log overflowed: once more with room */
                p[s].hitcap = nh + nh / 4;
                p[s].hitbuf = (sk_hit *)realloc(p[s].hitbuf, (size_t)p[s].hitcap * sizeof(sk_hit));
                if ((rc = sk_tally_launch(p[s].ctx, batches[sd_dev_of_strain(s)], SD_TYPE, SD_INFORMATIVE, p[s].hitcap)) != SK_OK ||
                    (rc = sk_tally_collect_sparse(p[s].ctx, (sk_tally_rec *)p[s].tallybuf, c->np, &nsp, p[s].hitbuf, &nh)) != SK_OK) return rc;
            }
            if ((rc = chunk_res_reserve(c, &rcap, &hcap, nr + nsp, nhit + nh)) != SK_OK) return rc;
            memcpy(c->res_recs + nr, p[s].tallybuf, (size_t)nsp * sizeof(sk_tally_rec));
            memcpy(c->res_hits + nhit, p[s].hitbuf, (size_t)nh * sizeof(sk_hit));
            c->res_off[s] = nr; c->res_hoff[s] = nhit;
            nr += nsp; nhit += nh;
        }
        c->res_off[ns] = nr; c->res_hoff[ns] = nhit;
        return SK_OK;
    }
    for (g = 0; g < sd_un.n; g++) {
        const uint32_t a = g * sd_group, n = sk_union_members(sd_un.u[g]);
        const uint64_t worst = (uint64_t)c->np * n;
        uint64_t nsp = 0, nh = 0, e;
        uint64_t cnt[SK_UNION_MAX + 1], hcnt[SK_UNION_MAX + 1];
        if (worst > sd_un.recs_cap) {                       /* (address space only: pages are touched as far as results come back) */
            free(sd_un.recs);
            sd_un.recs_cap = worst + worst / 4 + 1024;
            sd_un.recs = (sk_tally_rec *)malloc((size_t)sd_un.recs_cap * sizeof *sd_un.recs);
        }
        for (;;) {
            sd_un.hits = (sk_hit *)realloc(sd_un.hits, (size_t)sd_un.hcap[g] * sizeof(sk_hit));
            if ((rc = sk_union_tally_collect(sd_un.u[g], sd_un.recs, sd_un.recs_cap, &nsp, sd_un.hits, &nh)) != SK_OK) return rc;
            if (nh <= sd_un.hcap[g]) break;
            sd_un.hcap[g] = nh + nh / 4;                    /* the log overflowed: once more with room */
            if ((rc = sk_union_tally_launch(sd_un.u[g], batches[sd_dev_of_strain(a)], sd_un.hcap[g])) != SK_OK) return rc;
        }
        if ((rc = chunk_res_reserve(c, &rcap, &hcap, nr + nsp, nhit + nh)) != SK_OK) return rc;
        memset(cnt, 0, sizeof cnt); memset(hcnt, 0, sizeof hcnt);
        for (e = 0; e < nsp; e++) cnt[sd_un.recs[e].rec % n]++;
        for (e = 0; e < nh; e++) hcnt[sd_un.hits[e].row >> SK_UNION_ROW_BITS]++;
        for (s = 0; s < n; s++) {                           /* the members' shares, one behind the other */
            c->res_off[a + s] = nr; c->res_hoff[a + s] = nhit;
            nr += cnt[s]; nhit += hcnt[s];
            cnt[s] = c->res_off[a + s]; hcnt[s] = c->res_hoff[a + s];      /* (now: where the member's next entry goes) */
        }
        for (e = 0; e < nsp; e++) {
            sk_tally_rec *out = c->res_recs + cnt[sd_un.recs[e].rec % n]++;
            out->rec = sd_un.recs[e].rec / n; out->all = sd_un.recs[e].all; out->inf = sd_un.recs[e].inf;
        }
        for (e = 0; e < nh; e++) {
            sk_hit *out = c->res_hits + hcnt[sd_un.hits[e].row >> SK_UNION_ROW_BITS]++;
            out->pos = sd_un.hits[e].pos; out->row = sd_un.hits[e].row & ((1u << SK_UNION_ROW_BITS) - 1u);
        }
    }
    c->res_off[ns] = nr; c->res_hoff[ns] = nhit;
    return SK_OK;
}

/* One chunk against every strain.  uploaded: its bytes are in `batches` already (prefetched while the chunk before it was scanned);
 * launched: its scans are in flight as well (started while the chunk before it was being replayed: sd_launch_ahead). */
static int sd_tally_chunk(sd_prog *p, uint32_t ns, sk_batch **batches, sd_pool *pool, sd_chunk *c, int uploaded, int launched, sd_stream *st)
{
    int rc;
    double t0 = now_s(), t1;
    (void)pool;
    c->nstrains = ns;
    c->sp = (sd_sp *)calloc(ns, sizeof *c->sp);
    if (c->np) {
        int d;
        pthread_mutex_lock(&sd_dev_mu);
        rc = SK_OK;
        for (d = 0; d < sd_dev.n && !uploaded && rc == SK_OK; d++)  /* (asynchronous copies from page-locked memory: the devices' uploads overlap) */
            rc = c->packed ? sk_batch_fill_packed(batches[d], c->buf, c->blen, c->pstart, c->np) : sk_batch_fill(batches[d], c->buf, c->blen, c->pstart, c->np);
        t1 = now_s(); t_fill += t1 - t0; t0 = t1;
        if (rc == SK_OK && !launched) rc = sd_launch(p, ns, batches);
        if (rc == SK_OK) {
            sd_prefetch(st);                                     /* the next chunk goes up while this one is scanned */
            rc = sd_collect(p, ns, batches, c);
        }
        pthread_mutex_unlock(&sd_dev_mu);
        if (rc != SK_OK) return rc;
        t1 = now_s(); t_launch += t1 - t0; t0 = t1;
    }
    lanes_post(job_new(SD_JOB_TALLY, c, NULL));                  /* sort + spread, strain by strain, behind the main thread's back */
    t_post += now_s() - t0;
    return SK_OK;
}

/* The chunk that follows is uploaded already (st->pre) and this stream is the only user of the tables: start its scans NOW, so
 * that the devices work on it while the host replays the reference's read-after-read bookkeeping over the chunk just collected
 * (and compresses the strains' output) -- the two used to take turns.  Every result buffer of the current chunk has been copied
 * out by now (collect + spread), so the launches may overwrite the device side. */
static void sd_launch_ahead(sd_stream *st, sd_prog *p, uint32_t ns)
{
    const double t0 = now_s();
    if (!st->solo || !st->pre || st->ahead || getenv("SK_SD_NO_AHEAD")) return;
    pthread_mutex_lock(&sd_dev_mu);
    if (sd_launch(p, ns, st->bat[st->bcur ^ 1]) == SK_OK) st->ahead = st->pre;
    else sd_drain_scans();
    pthread_mutex_unlock(&sd_dev_mu);
    /* (a failed launch is not an error here: the chunk is launched again, and the error reported, when its turn comes -- but
     * the launches that DID start before the one that failed are waited for first: a context takes one launch at a time, and
     * their results are not collected -- ADVICE r03) */
    t_launch += now_s() - t0;
}

/* make sure the stream's current chunk has an unread record: 1 = st->c->...[st->ci] is it, 0 = end of
 * file (end_kind, end_len set), < 0 device error */
static int stream_fill(sd_stream *st, sd_prog *p, uint32_t ns, sk_batch *batch, sd_pool *pool)
{
    for (;;) {
        sd_chunk *c;
        int rc, i;
        if (st->c && st->ci < st->c->nrec) return 1;
        if (st->eof) return 0;
        if (st->c) {
            if (st->c->last) {
                st->eof = 1; st->end_kind = st->c->end_kind; st->end_len = st->c->end_len; chunk_free(st->c); st->c = NULL;
                return st->split_failed ? SK_E_SPLIT : 0;
            }
            { const double t0 = now_s(); chunk_free(st->c); t_cfree += now_s() - t0; }
            st->c = NULL;
        }
        {
            const double t0 = now_s();
            pthread_mutex_lock(&st->mu);
            while (st->qn == 0) pthread_cond_wait(&st->cv, &st->mu);
            t_wait += now_s() - t0;
        }
        c = st->q[0];
        for (i = 1; i < st->qn; i++) st->q[i - 1] = st->q[i];
        st->qn--;
        pthread_cond_broadcast(&st->cv);
        pthread_mutex_unlock(&st->mu);
        {
            const double t0 = now_s();
            const int uploaded = st->pre == c;             /* its bytes went up while the chunk before it was scanned */
            const int launched = uploaded && st->ahead == c;   /* ... and its scans were started while that chunk was replayed */
            (void)batch;
            if (uploaded) st->bcur ^= 1;
            st->pre = NULL;
            st->ahead = NULL;
            for (i = 0; i < sd_dev.n; i++)
                if (!st->bat[st->bcur][i] && (rc = sd_batch_get(i, &st->bat[st->bcur][i])) != SK_OK) { st->bat[st->bcur][i] = NULL; chunk_free(c); return rc; }
            rc = sd_tally_chunk(p, ns, st->bat[st->bcur], pool, c, uploaded, launched, st);
            if (rc == SK_OK) sd_launch_ahead(st, p, ns);
            t_tally += now_s() - t0;
        }
        if (rc != SK_OK) { chunk_free(c); return rc; }
        st->c = c;
        st->ci = 0;
    }
}

/* take the current, fully read chunk away from the stream so that it survives the next stream_fill */
static sd_chunk *stream_steal(sd_stream *st)
{
    sd_chunk *c = st->c;
    if (c->last) { st->eof = 1; st->end_kind = c->end_kind; st->end_len = c->end_len; }
    st->c = NULL;
    return c;
}

static unsigned char *put_int(unsigned char *w, int v)
{
    char tmp[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    if (v < 0) *w++ = '-';
    do { tmp[n++] = (char)('0' + u % 10u); u /= 10u; } while (u);
    while (n) *w++ = (unsigned char)tmp[--n];
    return w;
}

/* the hit lines of one read: "<file>\t<hits PE1>\t<informative PE1>\t<hits PE2>\t<informative PE2>\t<k-mer>\n"
 * (src/strain_detect.c:567,608), formatted straight into the output block */
static void emit_rows(sd_prog *p, const uint32_t *rows, uint32_t n, const char *name)
{
    const size_t nl = strlen(name);
    uint32_t j;
    char key[32];
    for (j = 0; j < n; j++) {
        if (p->table_on_device && p->ks.packed[rows[j]] == 0) {       /* (a row whose key was not fetched with the informative ones: one at a time) */
            int frc;
            pthread_mutex_lock(&sd_dev_mu);
            frc = skh_keyset_fetch_keys(&p->ks, p->ctx, &rows[j], 1);
            pthread_mutex_unlock(&sd_dev_mu);
            if (frc != SK_OK) {
                p->job_rc = SK_E_HIP;
                return;                                      /* never print a line without its k-mer (a key of 0 would decode to thirty-one A's): the run fails */
            }
        }
        skh_keyset_key(&p->ks, rows[j], key);
        if (nl <= 3800) {
            unsigned char *w0 = skzo_reserve(p->zo, nl + 128), *w = w0;
            memcpy(w, name, nl); w += nl;
            *w++ = '\t'; w = put_int(w, p->h1);
            *w++ = '\t'; w = put_int(w, p->i1);
            *w++ = '\t'; w = put_int(w, p->h2);
            *w++ = '\t'; w = put_int(w, p->i2);
            *w++ = '\t';
            { const size_t kl = strlen(key); memcpy(w, key, kl); w += kl; }
            *w++ = '\n';
            p->zo->len += (size_t)(w - w0);
        } else {                                           /* a path longer than a block's slack: the slow way */
            char num[64];
            skzo_append(p->zo, name, nl);
            skzo_append(p->zo, num, (size_t)snprintf(num, sizeof num, "\t%d\t%d\t%d\t%d\t", p->h1, p->i1, p->h2, p->i2));
            skzo_append(p->zo, key, strlen(key));
            skzo_append(p->zo, "\n", 1);
        }
        if (p->cov) skc_add_hit(p->cov, name, p->h1, p->h2, rows[j]);
    }
}

static void emit_trailer(sd_prog *p, const char *f1, const char *what, long long v)
{
    char num[64];
    skzo_append(p->zo, "#", 1);
    skzo_append(p->zo, f1, strlen(f1));
    skzo_append(p->zo, "\t", 1);
    skzo_append(p->zo, what, strlen(what));
    skzo_append(p->zo, num, (size_t)snprintf(num, sizeof num, "\t%lld\n", v));
}

/* One strain over a run of n read pairs whose records sit in the current chunks: PE1 reads are records
 * a0, a0 + astep, ... of chunk ca; their mates (cb != NULL) records b0, b0 + astep, ... of chunk cb.
 * This is the body of the reference's read loop (src/strain_detect.c:443-626) for that strain. */
static uint32_t sp_lower(const sd_sp *q, uint32_t rec)
{
    uint32_t lo = 0, hi = q->n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (q->rec[mid] < rec) lo = mid + 1; else hi = mid; }
    return lo;
}

/* first_valid: the first pair of the run whose PE1 read has k bases (n if none) -- from there on the copy of "the PE1 read
 * last seen" exists (have_copy; the same for every strain).
 * Only the pairs that matter are walked: a pair with a hit, and the pairs behind it for as long as the carried tallies are
 * not all zero (a read shorter than k refreshes nothing, :444, so it re-emits what the read before it left).  A pair without
 * hits met with all-zero tallies leaves them all zero and emits nothing: skipped. */
static void sd_replay_run(sd_prog *p, uint32_t s, const char *f1, const sd_chunk *ca, uint32_t a0, const sd_chunk *cb, uint32_t b0,
                          uint32_t astep, uint32_t n, int have_copy_in, uint32_t first_valid)
{
    const sd_sp *A = &ca->sp[s], *B = cb ? &cb->sp[s] : NULL;
    const uint64_t a_end = (uint64_t)a0 + (uint64_t)n * astep, b_end = (uint64_t)b0 + (uint64_t)n * astep;
    uint32_t ia = sp_lower(A, a0), ib = B ? sp_lower(B, b0) : 0, j = 0;
    while (j < n) {
        uint32_t ra, rb;
        int b_valid = 0;
        if (!(p->h1 | p->i1 | p->h2 | p->i2)) {                 /* nothing carried: on to the next pair with a hit */
            uint32_t ja = n, jb = n;
            while (ia < A->n && A->rec[ia] < (uint64_t)a0 + (uint64_t)j * astep) ia++;       /* (behind pair j: done with) */
            while (B && ib < B->n && B->rec[ib] < (uint64_t)b0 + (uint64_t)j * astep) ib++;
            for (; ia < A->n && A->rec[ia] < a_end; ia++) { const uint32_t off = A->rec[ia] - a0; if (off % astep == 0) { ja = off / astep; break; } }
            if (B) for (; ib < B->n && B->rec[ib] < b_end; ib++) { const uint32_t off = B->rec[ib] - b0; if (off % astep == 0) { jb = off / astep; break; } }
            if ((ja < jb ? ja : jb) > j) j = ja < jb ? ja : jb;
            if (j >= n) break;
        }
        ra = a0 + j * astep; rb = b0 + j * astep;
        if (ca->len[ra] >= SK_K) {                       /* a read shorter than k refreshes nothing (:444) */
            uint32_t nrows = 0;
            while (ia < A->n && A->rec[ia] < ra) ia++;
            if (ia < A->n && A->rec[ia] == ra) {
                p->h1 = (int)A->all[ia]; p->i1 = (int)A->inf[ia];
                nrows = A->hbeg[ia + 1] - A->hbeg[ia];
                if (nrows) {
                    if (nrows > p->copy_cap) { p->copy_cap = nrows * 2 + 16; p->copy_rows = (uint32_t *)realloc(p->copy_rows, (size_t)p->copy_cap * 4); }
                    memcpy(p->copy_rows, A->rows + A->hbeg[ia], (size_t)nrows * 4);
                }
            } else p->h1 = p->i1 = 0;
            p->copy_n = nrows;
        }
        if (cb && cb->len[rb] >= SK_K) {
            b_valid = 1;
            while (ib < B->n && B->rec[ib] < rb) ib++;
            if (ib < B->n && B->rec[ib] == rb) { p->h2 = (int)B->all[ib]; p->i2 = (int)B->inf[ib]; }
            else { p->h2 = p->i2 = 0; b_valid = 2; }       /* (valid, no rows) */
        }
        if (p->h1 + p->h2 >= 1 && p->i1 + p->i2 >= 1) {
            if (have_copy_in || first_valid <= j) emit_rows(p, p->copy_rows, p->copy_n, f1);
            if (b_valid == 1) emit_rows(p, B->rows + B->hbeg[ib], B->hbeg[ib + 1] - B->hbeg[ib], f1);
        }
        j++;
    }
}

/* Post one run to the lanes (they replay it strain by strain, behind the runs posted before); returns the (strain-independent)
 * have_copy afterwards.  The job holds the two chunks until every lane is through with it. */
static int post_replay(const char *f1, sd_chunk *ca, uint32_t a0, sd_chunk *cb, uint32_t b0, uint32_t astep, uint32_t n, int have_copy)
{
    sd_job *j = job_new(SD_JOB_REPLAY, ca, cb);
    uint32_t k;
    j->f1 = f1; j->a0 = a0; j->b0 = b0; j->astep = astep; j->n = n; j->have_copy = have_copy;
    for (k = 0; k < n && ca->len[a0 + k * astep] < SK_K; k++) { }
    j->first_valid = k;
    lanes_post(j);
    return have_copy || k < n;
}

/* one metagenome (pair), for every strain at once.  The read lengths -- hence which read refreshes which
 * tallies, the totals and the "PE2 ended early" failure -- are the same for all strains and are handled
 * here, run by run; the per-strain part is sd_replay_run.  A read shorter than k refreshes nothing, so it
 * re-uses (and may re-emit) the previous read's tallies and sequence, as in the reference. */
static int sd_quantify(sd_prog *p, uint32_t ns, sk_batch *batch, sd_pool *pool, const char *f1, const char *f2, int mode)
{
    sd_stream A, B;
    uint32_t s, j;
    int have_copy = 0, rc, got, status = 1;
    double t_mark;
    unsigned long long evaluated = 0, reads = 0;
    FILE *err = p[0].err;

    memset(&B, 0, sizeof B);
    t_mark = now_s();
    rc = stream_open(&A, f1, sd_gz_threads(mode == SD_PE ? 2 : 1));
    if (rc == SK_E_OPEN) { fprintf(err, "could not read file (read1) %s in quantify_hits_PE() (error: %s)\n", f1, strerror(errno)); return 1; }
    if (rc) { fprintf(err, "strain_detect: cannot start the reader of %s\n", f1); return 1; }
    if (mode == SD_PE) {
        rc = stream_open(&B, f2, sd_gz_threads(2));
        if (rc == SK_E_OPEN) { fprintf(err, "could not read file (read2) is_PE %s in quantify_hits_PE() (error: (null))\n", f2); stream_close(&A); return 1; }
        if (rc) { fprintf(err, "strain_detect: cannot start the reader of %s\n", f2); stream_close(&A); return 1; }
    }
    for (s = 0; s < ns; s++) { p[s].h1 = p[s].i1 = p[s].h2 = p[s].i2 = 0; p[s].copy_n = 0; }
    t_sopen += now_s() - t_mark;
    A.solo = mode != SD_PE;                              /* (two streams take turns at the tables: no scan ahead of the replay) */

    while ((got = stream_fill(&A, p, ns, batch, pool)) == 1) {
        sd_chunk *ca = A.c, *cb = NULL, *held = NULL;
        uint32_t a0 = A.ci, b0 = 0, n = ca->nrec - a0, astep = 1;
        int mate_missing = 0;
        uint64_t stale_len2 = 0;
        if (mode == SD_PE) {
            got = stream_fill(&B, p, ns, batch, pool);
            if (got < 0) break;
            if (got == 1) { cb = B.c; b0 = B.ci; if (cb->nrec - b0 < n) n = cb->nrec - b0; }
            else { mate_missing = 1; stale_len2 = B.end_kind == SKP_END_RESET ? 0 : B.end_len; }   /* PE2 is exhausted for good */
        } else if (mode == SD_PEI) {
            if (n >= 2) { cb = ca; b0 = a0 + 1; astep = 2; n /= 2; }
            else {                                       /* the mate is the first record of the next chunk, or missing */
                held = stream_steal(&A);
                got = stream_fill(&A, p, ns, batch, pool);
                if (got < 0) { chunk_free(held); break; }
                if (got == 1) { cb = A.c; b0 = A.ci; }
                else { mate_missing = 1; stale_len2 = A.end_kind == SKP_END_RESET ? 0 : A.end_len; }
            }
        }
        /* strain-independent part of the run (:444-450,489-512) */
        t_mark = now_s();
        if (!cb && astep == 1 && a0 == 0 && n == ca->nrec) {
            /* a whole chunk of single reads (the usual case): its builder counted the records of k bases or more (np) and laid their
             * bases end to end with a '\n' after each (blen) -- the sums without a walk over 200,000 lengths per chunk */
            reads += ca->np;
            evaluated += (ca->blen - ca->np) - (uint64_t)(SK_K - 1) * ca->np;
        } else
        for (j = 0; j < n; j++) {
            const uint64_t la = ca->len[a0 + j * astep];
            if (la >= SK_K) { reads++; evaluated += la - (SK_K - 1); }
            if (cb) { const uint64_t lb = cb->len[b0 + j * astep]; if (lb >= SK_K) evaluated += lb - (SK_K - 1); }
        }
        if (mate_missing && stale_len2 >= SK_K) {        /* the mate file ended first and its last length still reads as a read (:505-512) */
            fprintf(err, "reached end of PE2 (%s) before end of PE1 (%s), check that file names are correct\n", f2 ? f2 : "(null)", f1);
            chunk_free(held);
            goto done;
        }
        t_lens += now_s() - t_mark; t_mark = now_s();
        have_copy = post_replay(f1, ca, a0, cb, b0, astep, n, have_copy);
        t_replay += now_s() - t_mark;
        if (held) { chunk_free(held); if (cb) A.ci++; }  /* PEI across a chunk boundary: the mate was A's next record */
        else {
            A.ci += n * astep;
            if (mode == SD_PE && cb) B.ci += n;
        }
    }
    {   /* the lanes run dry before anything is said about the file: the trailer lines follow the last hit line, and a strain's error
         * state is read only here */
        const double t0 = now_s();
        lanes_drain();
        t_lane_drain += now_s() - t0;
        for (s = 0; s < ns && got >= 0; s++) if (p[s].job_rc != SK_OK) got = p[s].job_rc;
    }
    if (got == SK_E_SPLIT) {
        fprintf(err, "strain_detect: %s could not be cut at record boundaries for parsing on several threads: nothing from it is reported; "
                     "run again with SK_NO_SPLIT=1\n", A.split_failed || !f2 ? f1 : f2);
        goto done;
    }
    if (got < 0) {
        fprintf(err, "strain_detect: device error on %s: %s (%s)\n", f1, sk_strerror(got), sk_last_error(p[0].ctx));
        goto done;
    }
    for (s = 0; s < ns; s++) {
        emit_trailer(&p[s], f1, "total_kmer_evaluated", (long long)evaluated);
        emit_trailer(&p[s], f1, "total_reads_evaluated", (long long)reads);
        emit_trailer(&p[s], f1, "total_genome_kmers", (long long)p[s].ks.nrows);
        emit_trailer(&p[s], f1, "total_genome_informative_kmers", (long long)p[s].genome_inf);
        if (p[s].cov) {
            skc_add_trailer(p[s].cov, f1, "total_kmer_evaluated", (int64_t)evaluated);
            skc_add_trailer(p[s].cov, f1, "total_reads_evaluated", (int64_t)reads);
            skc_add_trailer(p[s].cov, f1, "total_genome_kmers", (int64_t)p[s].ks.nrows);
            skc_add_trailer(p[s].cov, f1, "total_genome_informative_kmers", (int64_t)p[s].genome_inf);
        }
    }
    status = 0;
done:
    lanes_drain();                                       /* (every way out: nothing of this file is in work when its streams go) */
    t_mark = now_s();
    stream_close(&A);
    stream_close(&B);
    t_sclose += now_s() - t_mark;
    return status;
}

/* ---------------------------------------------------------------------------------------------
 * -a: flag the listed k-mers as informative (src/strain_detect.c:668-726)
 * ------------------------------------------------------------------------------------------- */
static int sd_flag_informative(sd_prog *p, const char *path, unsigned *n_out)
{
    gzFile g = gzopen(path, "r");
    char line[100];
    char **piece = NULL; int *kind = NULL;             /* kind: 0 wrong length, 1 looked up on the device, 2 cannot match */
    uint32_t *slot = NULL;
    size_t np = 0, cap = 0, i;
    uint8_t *stream = NULL; uint32_t *start = NULL, *tally = NULL; sk_hit *hits = NULL;
    uint32_t nq = 0;
    uint64_t nh = 0;
    unsigned n = 0;
    int rc = 0;
    if (!g) { fprintf(p->err, "could not read file %s in hash_scrubbed_kmers()\n", path); return 1; }
    while (gzgets(g, line, 100)) {                       /* pieces of at most 99 bytes, as the reference reads */
        char *nl;
        if (line[0] == '#') continue;
        if ((nl = strchr(line, '\n')) != NULL) *nl = '\0';
        if (np == cap) {
            cap = cap ? cap * 2 : 4096;
            piece = (char **)realloc(piece, cap * sizeof *piece);
            kind = (int *)realloc(kind, cap * sizeof *kind);
            slot = (uint32_t *)realloc(slot, cap * sizeof *slot);
        }
        piece[np] = strdup(line);
        kind[np] = 0;
        if (strlen(line) == SK_K) {
            const char *c;
            kind[np] = 1;
            /* the reference does not fold case here: a piece with a lower-case letter is oriented to a
             * string with a lower-case letter, which no (upper-cased) key equals */
            for (c = line; *c; c++) if (*c >= 'a' && *c <= 'z') kind[np] = 2;
            if (kind[np] == 1) slot[np] = nq++;
        }
        np++;
    }
    gzclose(g);
    if (nq) {
        uint32_t q = 0;
        stream = (uint8_t *)malloc((size_t)nq * 32);
        start = (uint32_t *)malloc((size_t)nq * 4);
        tally = (uint32_t *)malloc((size_t)nq * 8);
        hits = (sk_hit *)malloc((size_t)nq * sizeof *hits);
        for (i = 0; i < np; i++) {
            if (kind[i] != 1) continue;
            start[q] = q * 32;
            memcpy(stream + (size_t)q * 32, piece[i], SK_K);
            stream[(size_t)q * 32 + SK_K] = '\n';
            q++;
        }
        /* every row still has type PLAIN, so "informative hits" here are simply all hits */
        rc = sk_tally_batch(p->ctx, stream, (uint64_t)nq * 32, start, nq, SD_TYPE, SD_PLAIN, tally, hits, nq, &nh);
        if (rc) fprintf(p->err, "strain_detect: device error: %s (%s)\n", sk_strerror(rc), sk_last_error(p->ctx));
        else for (i = 0; i < nh && i < nq; i++) p->type[hits[i].row] = SD_INFORMATIVE;
    }
    for (i = 0; i < np && !rc; i++) {
        if (kind[i] == 0)
            fprintf(p->out, "error string length in the scrubbed kmer file (%s) must be the same size as the kmer length "
                            "(scrubbed kmer, scrubbed kmer len, seed len): %s, %d, %d\n", path, piece[i], (int)strlen(piece[i]), SK_K);
        else if (kind[i] == 1 && tally[2 * slot[i]]) n++;
        else fprintf(p->out, "error could not find informative kmer %s in the total kmer list\n", piece[i]);
    }
    if (!rc && p->ks.nrows) {
        /* every row's type is PLAIN on the device already (column 0 after the build: src/strain_detect.c:139): only the rows the
         * list named go up -- 50 k row numbers instead of the 20 MB column */
        uint32_t *rows = (uint32_t *)malloc(((size_t)nq + 1) * sizeof *rows), nr = 0;
        for (i = 0; i < nh && i < nq && rows; i++) rows[nr++] = hits[i].row;      /* (a row named twice is set twice) */
        rc = rows ? sk_counts_set_rows(p->ctx, SD_TYPE, rows, nr, SD_INFORMATIVE) : SK_E_NOMEM;
        free(rows);
        if (rc) fprintf(p->err, "strain_detect: device error: %s (%s)\n", sk_strerror(rc), sk_last_error(p->ctx));
    }
    for (i = 0; i < np; i++) free(piece[i]);
    free(piece); free(kind); free(slot); free(stream); free(start); free(tally); free(hits);
    *n_out = n;
    return rc ? 1 : 0;
}

/* ---------------------------------------------------------------------------------------------
 * -g: background filter (src/strain_detect.c:160-240)
 * ------------------------------------------------------------------------------------------- */
static int cmp_desc(const void *a, const void *b) { return (int)(*(const unsigned *)b - *(const unsigned *)a); }

static int removed_at(unsigned threshold, const unsigned *c, unsigned n)
{
    unsigned i; int r = 0;
    for (i = 0; i < n; i++) if (c[i] >= threshold) r++;
    return r;
}

static int sd_background_filter(sd_prog *p, const char *list, double fraction, unsigned n_inform)
{
    unsigned keep = (unsigned)(int)(n_inform * fraction), *c, *bg, i, n = 0, threshold = 1;
    int demoted = 0, rc;
    fprintf(p->out, "#removing %f proportion of %s kmers; informative %d keep at least %d\n", fraction, list, n_inform, keep);
    if (skh_scan_list(p->ctx, list, NULL, SD_BACKGROUND, NULL, p->err, 0, 1, NULL) != SK_OK) return 1;
    c = (unsigned *)calloc(n_inform ? n_inform : 1, sizeof *c);
    bg = (unsigned *)malloc((size_t)(p->ks.nrows ? p->ks.nrows : 1) * sizeof *bg);
    rc = sk_counts_fetch(p->ctx, SD_BACKGROUND, bg);
    if (rc) { fprintf(p->err, "strain_detect: device error: %s (%s)\n", sk_strerror(rc), sk_last_error(p->ctx)); free(c); free(bg); return 1; }
    for (i = 0; i < p->ks.nrows; i++) {
        if (p->type[i] != SD_INFORMATIVE) continue;
        if (n >= n_inform) { fputs("Error: too many background kmers\n", p->err); free(c); free(bg); return 1; }
        c[n++] = bg[i];
    }
    qsort(c, n_inform, sizeof *c, cmp_desc);
    if (keep >= 1 && c[keep - 1] > threshold) threshold = c[keep - 1];
    while ((unsigned)removed_at(threshold, c, n_inform) > keep) threshold++;
    for (i = 0; i < p->ks.nrows; i++)
        if (p->type[i] == SD_INFORMATIVE && bg[i] >= threshold) { p->type[i] = SD_PLAIN; demoted++; }
    fprintf(p->out, "#final_threshold %d removes %d background kmers %d removed\n", threshold, removed_at(threshold, c, n_inform), demoted);
    free(c);
    free(bg);
    rc = sk_counts_set(p->ctx, SD_TYPE, p->type);
    return rc ? 1 : 0;
}

/* ---------------------------------------------------------------------------------------------
 * main (src/strain_detect.c:61-158, 263-384)
 * ------------------------------------------------------------------------------------------- */
static int file_type(const char *s)
{
    if (!strcmp(s, "SE") || !strcmp(s, "se")) return SD_SE;
    if (!strcmp(s, "PE") || !strcmp(s, "pe")) return SD_PE;
    if (!strcmp(s, "PEI") || !strcmp(s, "pei") || !strcmp(s, "IPE") || !strcmp(s, "ipe")) return SD_PEI;
    return SD_UNKNOWN;
}

static void usage(FILE *err)
{
    fputs("Usage paired end with 2 files:\n\tstrain_detect -r <reference_genome.fna> -a <informative_kmer_file.txt> -b <paired-end-file1> -c <paired-end-file1> -t PE -o <kmer outfile>\n", err);
    fputs("Usage paired end interleaved 1 file:\n\tstrain_detect -r <reference_genome.fna> -a <informative_kmer_file.txt> -b <paired-end-file1>  -t PEI -o <kmer outfile>\n", err);
    fputs("Usage single end 1 file:\n\tstrain_detect -r <reference_genome.fna> -a <informative_kmer_file.txt> -b <single-end-file1>  -t SE -o <kmer outfile>\n", err);
    fputs("Usage single end 1 file:\n\tstrain_detect -r <reference_genome.fna> -a <informative_kmer_file.txt> -B <batch-list-of-metagenomes> -o <kmer outfile>\n\n", err);
    fputs("format for metagenomics batch file is:\n", err);
    fputs("PE\tfile1_PE1.fasta\tfile1_PE2.fasta\n", err);
    fputs("SE\tfile1_PE1.fasta\n", err);
    fputs("PEI\tfile1_PE1.fasta\n", err);
    fputs("\nlines that begin with # are considered comments and ignored\n", err);
    fputs("\ninformative kmer file is a list of all of the kmers left in the reference genome post scrubbing\n", err);
}

static skzo_pool *sd_zpool;             /* compressors of every -o file of the run (set by the main function) */

/* build one strain's state in two steps: the key set (host only; several strains build theirs at the
 * same time on separate threads), then the table on the device, -a flags, optional -g filter, -o file */
/* one strain of a -S list, opened on a worker thread: key set on the host, then its own context, table,
 * -a flags, -g filter and outfile.  What it would have printed is kept in two buffers and replayed by the
 * main thread in list order. */
typedef struct {
    sd_prog *p; const char *r, *a, *g, *o; int device, failed, done;
    char *out_buf, *err_buf; size_t out_len, err_len;
    sk_ctxjob *cj;                        /* the first strain takes the context that was opened while the key sets were built */
} ks_job;
typedef struct { ks_job *jobs; uint32_t njobs, next; pthread_mutex_t mu; pthread_cond_t cv; } ks_pool;

static int sd_strain_finish(sd_prog *p, int ks_rc, const char *r, const char *a, const char *g, const char *o, int device);

/* A strain's key set.  Round 3: built ON THE DEVICE from the strain's text (skh_keyset_build_on_device: the host only parses the
 * file and packs the bases; no hash table of 5 M keys on the host, no 140 MB of keys, permutations and columns over PCIe) -- the
 * context must be there first.  A strain with byte-string keys (U, IUPAC), a context that could not be opened (the caller reports it)
 * or SK_SD_HOST_KEYSET=1 (tests compare the two ways) take the host's builder as before; strain_detect never shows the order of
 * the rows, so either numbering serves. */
static int sd_keyset(sd_prog *p, const char *r, int device)
{
    if (!getenv("SK_SD_HOST_KEYSET")) {
        if (!p->ctx && !p->ctx_rc) p->ctx_rc = sk_ctx_create(&p->ctx, device);
        if (p->ctx) {
            const int rc = skh_keyset_build_on_device(&p->ks, p->ctx, r, SD_NCOLS, SD_PLAIN);
            if (rc == SK_OK) { p->table_on_device = 1; return SK_OK; }
            skh_keyset_free(&p->ks);
            if (rc != SK_E_UNSUPPORTED && rc != SK_E_OPEN) return rc;      /* (an unreadable file: the host's builder says so in the reference's words) */
        }
    }
    return skh_keyset_from_file(&p->ks, r, SD_ROW_ORDER, SD_PLAIN, 0);
}

static void *sd_keyset_pool_thread(void *arg)
{
    ks_pool *kp = (ks_pool *)arg;
    for (;;) {
        const uint32_t k = __atomic_fetch_add(&kp->next, 1u, __ATOMIC_RELAXED);
        ks_job *j;
        FILE *real_out, *real_err, *mo, *me;
        int rc;
        if (k >= kp->njobs) return NULL;
        j = &kp->jobs[k];
        if (j->cj) j->p->ctx_rc = sk_ctxjob_join(j->cj, &j->p->ctx);
        rc = sd_keyset(j->p, j->r, j->device);
        real_out = j->p->out; real_err = j->p->err;
        mo = open_memstream(&j->out_buf, &j->out_len);
        me = open_memstream(&j->err_buf, &j->err_len);
        if (mo && me) { j->p->out = mo; j->p->err = me; }
        j->failed = sd_strain_finish(j->p, rc, j->r, j->a, j->g, j->o, j->device);
        if (mo) fclose(mo);
        if (me) fclose(me);
        j->p->out = real_out; j->p->err = real_err;
        pthread_mutex_lock(&kp->mu);
        j->done = 1;
        pthread_cond_broadcast(&kp->cv);
        pthread_mutex_unlock(&kp->mu);
    }
}

/* the informative rows, the -g filter, the output file: the part of a strain's opening that follows the table load */
static int sd_strain_flags(sd_prog *p, const char *a, const char *g, const char *o)
{
    unsigned n_inform = 0, i;
    p->type = (uint32_t *)malloc((size_t)(p->ks.nrows ? p->ks.nrows : 1) * sizeof(uint32_t));
    for (i = 0; i < p->ks.nrows; i++) p->type[i] = SD_PLAIN;
    if (sd_flag_informative(p, a, &n_inform)) return 1;
    if (g && sd_background_filter(p, g, 0.5, n_inform)) return 1;
    for (i = 0; i < p->ks.nrows; i++) if (p->type[i] == SD_INFORMATIVE) p->genome_inf++;
    if (p->table_on_device && p->genome_inf) {           /* the k-mers the hit lines will print: the informative rows' keys, and no others */
        uint32_t *rows = (uint32_t *)malloc((size_t)p->genome_inf * sizeof *rows), nr = 0;
        int rc;
        for (i = 0; i < p->ks.nrows; i++) if (p->type[i] == SD_INFORMATIVE) rows[nr++] = i;
        rc = skh_keyset_fetch_keys(&p->ks, p->ctx, rows, nr);
        free(rows);
        if (rc) { fprintf(p->err, "strain_detect: device error: %s (%s)\n", sk_strerror(rc), sk_last_error(p->ctx)); return 1; }
    }
    p->zo = skzo_open(sd_zpool, o);
    if (!p->zo) { fprintf(p->err, "could not open *gzout file outfile %s in quantify_hits_all_files()\n", o); return 1; }
    p->o_path = strdup(o);
    return 0;
}

static int sd_strain_finish(sd_prog *p, int ks_rc, const char *r, const char *a, const char *g, const char *o, int device)
{   /* p->ctx may already hold a context opened in the background (single-strain start-up) */
    int rc = ks_rc;
    FILE *err = p->err;
    if (rc == SK_E_OPEN) { fprintf(err, "could not read file %s GEN_hash_sequences_set_count_vec()\n", r); return 1; }
    if (rc != SK_OK) { fprintf(err, "strain_detect: %s\n", sk_strerror(rc)); return 1; }
    if (p->ks.short_records)
        fprintf(err, "strain_detect: skipped %llu reference record(s) shorter than %d bases "
                     "(the original program crashes on these)\n", (unsigned long long)p->ks.short_records, SK_K - 1);
    {
        const double t0 = now_s();
        double t1, t2;
        int frc;
        rc = p->ctx ? SK_OK : (p->ctx_rc ? p->ctx_rc : sk_ctx_create(&p->ctx, device));
        if (rc != SK_OK) { fprintf(err, "strain_detect: cannot use HIP device %d: %s\n", device, sk_strerror(rc)); return 1; }
        t1 = now_s();
        rc = p->table_on_device ? SK_OK : skh_keyset_load(p->ctx, &p->ks, SD_NCOLS);
        if (rc != SK_OK) { fprintf(err, "strain_detect: table load failed: %s (%s)\n", sk_strerror(rc), sk_last_error(p->ctx)); return 1; }
        t2 = now_s();
        frc = sd_strain_flags(p, a, g, o);
        pthread_mutex_lock(&t_open_mu);
        t_open_ctx += t1 - t0; t_open_load += t2 - t1; t_open_flags += now_s() - t2;
        pthread_mutex_unlock(&t_open_mu);
        return frc;
    }
}

/* The strain of a kmer_scrub_count run that goes on into strain_detect in the same process (key set, row
 * order, device table, filters and text are built ONCE for steps 1 and 3; src/strain_detect.c:137-146 rebuilds what
 * src/kmer_scrub_count.c:87-89 built).  Takes over ctx and *ks (the caller's copy is cleared); the table must have been
 * loaded with at least SD_NCOLS columns.  Every column goes back to what a fresh strain_detect start has. */
static int sd_strain_adopt(sd_prog *p, sk_ctx *ctx, skh_keyset *ks, const char *a, const char *g, const char *o, FILE *out, FILE *err)
{
    uint32_t col;
    int rc = SK_OK;
    memset(p, 0, sizeof *p);
    p->out = out; p->err = err; p->ctx = ctx; p->ks = *ks;
    memset(ks, 0, sizeof *ks);
    if (sk_table_cols(ctx) < SD_NCOLS || sk_table_rows(ctx) != p->ks.nrows) { fprintf(err, "strain_detect: the resident table does not fit (columns/rows)\n"); return 1; }
    for (col = 0; col < SD_NCOLS && rc == SK_OK; col++) rc = sk_counts_zero(ctx, col);
    if (rc == SK_OK && p->ks.nrows) {
        uint32_t *plain = (uint32_t *)malloc((size_t)p->ks.nrows * sizeof(uint32_t)), i;
        for (i = 0; i < p->ks.nrows; i++) plain[i] = SD_PLAIN;
        rc = sk_counts_set(ctx, SD_TYPE, plain);               /* what skh_keyset_load(.., default SD_PLAIN, incr 0) leaves there */
        free(plain);
    }
    if (rc != SK_OK) { fprintf(err, "strain_detect: device error: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx)); return 1; }
    return sd_strain_flags(p, a, g, o);
}

/* --coverage-depth: where the table of strain p goes, and the accumulator that collects it */
static int sd_coverage_open(sd_prog *p, const char *explicit_path, int64_t min_hits)
{
    if (explicit_path && *explicit_path) p->cov_path = strdup(explicit_path);
    else {
        const size_t n = strlen(p->o_path);
        p->cov_path = (char *)malloc(n + 32);
        strcpy(p->cov_path, p->o_path);
        if (n >= 13 && !strcmp(p->cov_path + n - 13, ".kmer_hits.gz")) p->cov_path[n - 13] = 0;
        strcat(p->cov_path, ".coverage_depth");
    }
    p->cov = skc_create(min_hits);
    return p->cov ? 0 : 1;
}

static int sd_coverage_write(sd_prog *p)
{
    FILE *f;
    int rc;
    if (!p->cov) return 0;
    if (!(f = fopen(p->cov_path, "w"))) { fprintf(p->err, "strain_detect: could not write %s\n", p->cov_path); return 1; }
    rc = skc_report(p->cov, p->ctx, p->o_path, f, p->err);
    fclose(f);
    return rc;
}

static int sd_strain_open(sd_prog *p, const char *r, const char *a, const char *g, const char *o, int device, FILE *out, FILE *err)
{
    sk_ctxjob cj;
    int ks_rc;
    memset(p, 0, sizeof *p);
    p->out = out;
    p->err = err;
    sk_ctxjob_start(&cj, device);                        /* (the HIP runtime comes up on a helper thread) */
    p->ctx_rc = sk_ctxjob_join(&cj, &p->ctx);
    ks_rc = sd_keyset(p, r, device);
    return sd_strain_finish(p, ks_rc, r, a, g, o, device);
}

static void sd_strain_close(sd_prog *p)
{
    if (p->zo && skzo_close(p->zo)) fprintf(stderr, "strain_detect: error writing the -o file\n");
    /* SK_LEAK_AT_EXIT=1 (set by bin/strain_detect, whose process ends right after): the result file is complete; taking 32
     * device contexts and key sets apart one hipFree at a time (0.3-0.7 s for 32 strains) is left to the end of the process */
    if (getenv("SK_LEAK_AT_EXIT") && strcmp(getenv("SK_LEAK_AT_EXIT"), "0")) { memset(p, 0, sizeof *p); return; }
    if (p->ctx) sk_ctx_destroy(p->ctx);
    skh_keyset_free(&p->ks);
    free(p->type); free(p->copy_rows); free(p->hitbuf); free(p->tallybuf); free(p->cov_path); free(p->o_path);
    skc_destroy(p->cov);
    memset(p, 0, sizeof *p);
}

static void close_one(void *arg, uint32_t s) { sd_strain_close(&((sd_prog *)arg)[s]); }

/* the metagenome side of main (src/strain_detect.c:263-384), for ns strains at once */
static int sd_run(sd_prog *p, uint32_t ns, const char *B, const char *b, const char *b2, int mode, FILE *out, FILE *err)
{
    sk_batch *batch = NULL;                          /* (every stream has its own pair of device batches: stream_fill) */
    sd_pool pool;
    int bad = 0;
    sd_pin_open(p[0].ctx, sd_chunk_bytes());
    sd_all_p = p; sd_all_ns = ns;
    sd_unions_open(p, ns);
    pool_start(&pool, ns);
    lanes_start(p, ns);
    if (B) {
        FILE *fp = fopen(B, "r");
        char *line = NULL, *nl, *tok, *f1, *f2;
        size_t cap = 0;
        if (!fp) { fprintf(err, "could not read file file_of_filenames %s in quantify_hits_all_files()\n", B); lanes_stop(); pool_stop(&pool); sd_unions_close(); return 1; }
        while (!bad && getline(&line, &cap, fp) != -1) {
            int m;
            if ((nl = strchr(line, '\n')) != NULL) *nl = '\0';
            tok = strtok(line, "\t");
            if (!tok) { fprintf(err, "strain_detect: empty line in %s (the original program crashes here)\n", B); bad = 1; break; }
            m = file_type(tok);
            if (m == SD_UNKNOWN) { fprintf(out, "unknown file type skipping line (%s)\n", tok); continue; }
            f1 = strtok(NULL, "\t");
            if (!f1) { fprintf(out, "ERROR: no first file specified for %s\n", line); continue; }
            if (m == SD_PE) {
                f2 = strtok(NULL, "\t");
                if (!f2) { fprintf(out, "ERROR: no second file specified for PE: %s\n", line); continue; }
                bad = sd_quantify(p, ns, batch, &pool, f1, f2, m);
            } else bad = sd_quantify(p, ns, batch, &pool, f1, NULL, m);
        }
        free(line);
        fclose(fp);
    } else bad = sd_quantify(p, ns, batch, &pool, b, b2, mode);
    lanes_stop();
    pool_stop(&pool);
    sd_unmap_wait();
    sd_batch_cache_close();
    if (getenv("SK_LEAK_AT_EXIT") && strcmp(getenv("SK_LEAK_AT_EXIT"), "0")) return bad;     /* (the process is about to end: see sd_strain_close) */
    { const double t0 = now_s(); sd_unions_close(); t_uclose = now_s() - t0; }
    return bad;
}

static int sd_main_impl(int argc, char **argv, FILE *out, FILE *err, sk_ctx *adopt_ctx, skh_keyset *adopt_ks, const char *adopt_a);

int skh_strain_detect_main(int argc, char **argv, FILE *out, FILE *err) { return sd_main_impl(argc, argv, out, err, NULL, NULL, NULL); }

/* strain_detect on a strain that is already resident: ctx holds its table (>= 6 columns), *ks its key set -- both are
 * taken over and released here -- and informative_path names the informative k-mer list (what -a would name).  argv: the
 * rest of a strain_detect command line (-B/-b/-c/-t/-g/-o, --coverage-depth ...), argv[0] ignored; -r and -a are implied. */
int skh_strain_detect_resident(sk_ctx *ctx, skh_keyset *ks, const char *informative_path, int argc, char **argv, FILE *out, FILE *err)
{
    if (!ctx || !ks || !informative_path) return 1;
    return sd_main_impl(argc, argv, out, err, ctx, ks, informative_path);
}

static int sd_main_impl(int argc, char **argv, FILE *out, FILE *err, sk_ctx *adopt_ctx, skh_keyset *adopt_ks, const char *adopt_a)
{
    const char *a = NULL, *r = NULL, *b = NULL, *b2 = NULL, *B = NULL, *tt = NULL, *g = NULL, *o = NULL, *S = NULL, *env;
    int c, j, mode = SD_SE, status = 1, device = 0, n_S = 0, want_cov = 0;
    skzo_pool zpool;
    const char *cov_file = NULL;
    long long cov_min = 1;
    double t_begin = 0;
    sd_prog *p = NULL;
    char **paths = NULL;
    uint32_t ns = 0, s;

    /* Extension (not in the reference): "--coverage-depth[=FILE]" also writes the table that
     * scripts/coverage_depth.py -k <outfile> [-m N] would print (step 4 of the workflow), collected while the
     * hits are emitted; default FILE = outfile with ".kmer_hits.gz" replaced by ".coverage_depth".
     * "--min-kmer-hits N" is that script's -m.  Taken out of argv before getopt. */
    for (c = 1, j = 1; c < argc; c++) {
        if (!strncmp(argv[c], "--coverage-depth", 16) && (argv[c][16] == 0 || argv[c][16] == '=')) {
            want_cov = 1;
            cov_file = argv[c][16] ? argv[c] + 17 : NULL;
            continue;
        }
        if (!strcmp(argv[c], "--min-kmer-hits") && c + 1 < argc) { cov_min = atoll(argv[++c]); continue; }
        argv[j++] = argv[c];
    }
    argc = j;

    optind = 1;
    while ((c = getopt(argc, argv, "g:r:a:A:b:c:B:S:M:o:t:Hhuspn")) != -1) {
        switch (c) {
        case 'a': a = optarg; break;
        case 'A': break;
        case 'b': b = optarg; break;
        case 'c': b2 = optarg; break;
        case 'B': B = optarg; break;
        case 'r': r = optarg; break;
        case 'g': g = optarg; break;
        case 'o': o = optarg; break;
        case 'n': mode = SD_SE; break;
        case 't': tt = optarg; break;
        case 'S': S = optarg; n_S++; break;              /* the reference prints its usage for -S and carries on; see below */
        default:  usage(err); break;
        }
    }
    /* Extension: "-S <file>" with NO -r/-a/-o runs several strains in one pass over the metagenomes.  Each
     * line of the file is  <reference genome> TAB <informative k-mer file> TAB <outfile> [TAB <-g list>];
     * every outfile gets exactly what a separate run with that line's -r/-a/-o[/-g] would write.  All
     * tables stay resident on the device; each metagenome is decoded and uploaded once. */
    if (adopt_ctx) { a = adopt_a; r = "(resident)"; S = NULL; }
    if (S && !a && !o && !r) {
        if (!b && !B) { usage(err); return 1; }
    } else {
        S = NULL;
        while (n_S-- > 0) usage(err);
        if (!a || !o || !r) { usage(err); return 1; }
        if (!b && !B) { usage(err); return 1; }
    }
    if (tt) {
        mode = file_type(tt);
        if (mode == SD_UNKNOWN) { fputs("unknown filetype specification. allowed are SE, PE, PEI\n\n", out); usage(err); return 1; }
    }
    if (b && mode == SD_PE && !b2) {
        fputs("commandline PE mapping requires two files (-b [file1] and -c [file2])\n\n", out); usage(err); return 1;
    }
    if (b && B) {
        fputs("cannot have -B flag and -b flag\nEither have a file with metagenomics files to be detect the strain in or "
              "specify one metagenomic file to detect the strain in\n", out);
        usage(err); return 1;
    }
    if ((env = getenv("SK_DEVICE")) != NULL) device = atoi(env);
    sd_dev.n = 1; sd_dev.phys[0] = device;
    sd_group = SK_UNION_MAX;
    if ((env = getenv("SK_SD_GROUP")) != NULL && atoi(env) >= 1 && atoi(env) <= SK_UNION_MAX) sd_group = (uint32_t)atoi(env);
    memset(sd_dev.ctx, 0, sizeof sd_dev.ctx);
    t_begin = now_s();
    {
        const long ncpu = sk_cpu_budget();
        skzo_pool_start(&zpool, getenv("SK_THREADS") ? atoi(getenv("SK_THREADS")) : (int)(ncpu > 16 ? 16 : ncpu < 1 ? 1 : ncpu));
        sd_zpool = &zpool;
    }

    if (S) {
        FILE *fp = fopen(S, "r");
        char *line = NULL;
        size_t cap = 0;
        const int world = getenv("SK_WORLD_SIZE") ? atoi(getenv("SK_WORLD_SIZE")) : (getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1);
        const int rank = getenv("SK_RANK") ? atoi(getenv("SK_RANK")) : (getenv("RANK") ? atoi(getenv("RANK")) : 0);
        unsigned lineno = 0;
        if (!fp) { fprintf(err, "strain_detect: could not read the strain list %s\n", S); goto done; }
        if (!getenv("SK_DEVICE") && (env = getenv("SK_LOCAL_RANK") ? getenv("SK_LOCAL_RANK") : getenv("LOCAL_RANK")) != NULL) device = atoi(env);
        sd_dev.phys[0] = device;
        if ((env = getenv("SK_DEVICES")) != NULL && *env) {      /* one process, several devices: see sd_dev */
            if (world > 1) { fprintf(err, "strain_detect: SK_DEVICES is for ONE process that drives several devices (WORLD_SIZE is %d)\n", world); fclose(fp); goto done; }
            if (strchr(env, ',')) {                               /* the devices as listed (several logical devices may share a card) */
                const char *q = env;
                sd_dev.n = 0;
                while (*q && sd_dev.n < SD_MAX_DEV) {
                    sd_dev.phys[sd_dev.n++] = atoi(q);
                    while (*q && *q != ',') q++;
                    if (*q == ',') q++;
                }
            } else {
                int k;
                sd_dev.n = atoi(env);
                if (sd_dev.n < 1 || sd_dev.n > SD_MAX_DEV) { fprintf(err, "strain_detect: SK_DEVICES must be 1..%d or a list of devices\n", SD_MAX_DEV); fclose(fp); goto done; }
                for (k = 0; k < sd_dev.n; k++) sd_dev.phys[k] = device + k;
            }
        }
        while (getline(&line, &cap, fp) != -1) {
            char *nl, *fr, *fa, *fo, *fg;
            if ((nl = strchr(line, '\n')) != NULL) *nl = '\0';
            if (line[0] == '#' || line[0] == '\0') continue;
            if (world > 1 && (int)(lineno++ % (unsigned)world) != rank) continue;     /* strains are dealt to the ranks; no collective */
            fr = strtok(line, "\t"); fa = strtok(NULL, "\t"); fo = strtok(NULL, "\t"); fg = strtok(NULL, "\t");
            if (!fr || !fa || !fo) { fprintf(err, "strain_detect: %s: a line needs <genome> TAB <informative k-mers> TAB <outfile>\n", S); free(line); fclose(fp); goto done; }
            p = (sd_prog *)realloc(p, ((size_t)ns + 1) * sizeof *p);
            paths = (char **)realloc(paths, ((size_t)ns + 1) * 4 * sizeof *paths);
            memset(&p[ns], 0, sizeof *p);
            p[ns].out = out; p[ns].err = err;
            paths[4 * ns] = strdup(fr); paths[4 * ns + 1] = strdup(fa); paths[4 * ns + 2] = strdup(fo); paths[4 * ns + 3] = fg ? strdup(fg) : NULL;
            ns++;
        }
        free(line);
        fclose(fp);
        {   /* the strains are opened by worker threads, each strain start to finish on
             * one of them; their messages are replayed here in list order, up to the first failure
             * (SK_THREADS, default: the usable CPUs, at most 16) */
            long ncpu = sk_cpu_budget();
            int nth = getenv("SK_THREADS") ? atoi(getenv("SK_THREADS")) : (int)(ncpu < 1 ? 1 : ncpu > 16 ? 16 : ncpu), t;
            ks_pool kp;
            pthread_t th[16];
            uint32_t k, failed = 0;
            sk_ctxjob cj;
            /* the HIP runtime comes up (0.25-3 s) while the first key sets are built: the context opened here becomes the first
             * strain's -- every other sk_ctx_create then finds the runtime ready instead of all workers queueing up behind it */
            {   /* no more devices than there are groups of strains */
                const uint32_t ng = (ns + sd_group - 1) / sd_group;
                if ((uint32_t)sd_dev.n > ng) sd_dev.n = ng ? (int)ng : 1;
            }
            if (ns) sk_ctxjob_start(&cj, sd_dev.phys[0]);
            memset(&kp, 0, sizeof kp);
            kp.jobs = (ks_job *)calloc(ns ? ns : 1, sizeof *kp.jobs);
            kp.njobs = ns;
            pthread_mutex_init(&kp.mu, NULL);
            pthread_cond_init(&kp.cv, NULL);
            for (k = 0; k < ns; k++) {
                ks_job *j = &kp.jobs[k];
                j->p = &p[k]; j->r = paths[4 * k]; j->a = paths[4 * k + 1]; j->o = paths[4 * k + 2]; j->g = paths[4 * k + 3];
                j->device = sd_dev.phys[sd_dev_of_strain(k)];
                j->cj = k == 0 ? &cj : NULL;
            }
            if (nth > 16) nth = 16;
            if (nth < 1) nth = 1;
            if ((uint32_t)nth > ns) nth = (int)ns;
            for (t = 0; t < nth; t++) if (pthread_create(&th[t], NULL, sd_keyset_pool_thread, &kp)) break;
            nth = t;
            if (nth == 0) sd_keyset_pool_thread(&kp);     /* no thread could be started: open them here */
            for (k = 0; k < ns; k++) {
                ks_job *j = &kp.jobs[k];
                pthread_mutex_lock(&kp.mu);
                while (!j->done) pthread_cond_wait(&kp.cv, &kp.mu);
                pthread_mutex_unlock(&kp.mu);
                if (!failed) {
                    if (j->out_buf && j->out_len) fwrite(j->out_buf, 1, j->out_len, out);
                    if (j->err_buf && j->err_len) fwrite(j->err_buf, 1, j->err_len, err);
                    failed = (uint32_t)j->failed;
                }
                free(j->out_buf); free(j->err_buf);
            }
            for (t = 0; t < nth; t++) pthread_join(th[t], NULL);
            pthread_mutex_destroy(&kp.mu);
            pthread_cond_destroy(&kp.cv);
            free(kp.jobs);
            if (failed) goto done;
        }
        if (ns == 0) { status = 0; goto done; }          /* nothing dealt to this rank */
    } else {
        p = (sd_prog *)malloc(sizeof *p);
        ns = 1;
        if (adopt_ctx ? sd_strain_adopt(&p[0], adopt_ctx, adopt_ks, a, g, o, out, err) : sd_strain_open(&p[0], r, a, g, o, device, out, err)) goto done;
    }
    t_setup = now_s() - t_begin;
    for (s = ns; s-- > 0; ) sd_dev.ctx[sd_dev_of_strain(s)] = p[s].ctx;       /* a context on every device (its first strain's): owner of the device's batches */
    if (getenv("SK_SD_TIMING") && sd_dev.n > 1) fprintf(err, "strain_detect timing: %u strains on %d devices, one decode pipeline\n", ns, sd_dev.n);
    for (s = 0; want_cov && s < ns; s++)
        if (sd_coverage_open(&p[s], S ? NULL : cov_file, cov_min)) goto done;
    if (sd_run(p, ns, B, b, b2, mode, out, err)) goto done;
    for (s = 0; s < ns; s++)
        if (sd_coverage_write(&p[s])) goto done;
    status = 0;
done:
    if (getenv("SK_SD_TIMING") && ns > 1)
        fprintf(err, "strain_detect timing: opening the strains, thread time summed: context %.2f s, table load %.2f s, -a/-g flags + outfile %.2f s\n",
                t_open_ctx, t_open_load, t_open_flags);
    if (getenv("SK_SD_TIMING"))
        fprintf(err, "strain_detect timing: setup %.2f s, waiting for the decode thread %.2f s, tally %.2f s (upload %.2f, launch + collect %.2f, "
                     "posting to the lanes %.2f), read lengths %.2f s, posting the runs %.2f s, waiting for the lanes at the files' ends %.2f s "
                     "(and %.2f s for room in a lane's ring), chunks freed %.2f s, files opened %.2f s and closed %.2f s, union tables freed %.2f s, "
                     "total before close %.2f s; lanes' busy time, summed: %.2f s (sort + spread + replay + compression hand-over)\n",
                t_setup, t_wait, t_tally, t_fill, t_launch, t_post, t_lens, t_replay, t_lane_drain, t_lane_backpressure, t_cfree, t_sopen, t_sclose,
                t_uclose, now_s() - t_begin, t_lane_busy);
    if (getenv("SK_SD_TIMING")) {                        /* who used the CPUs: the process as a whole against the threads this file starts */
        struct rusage ru;
        if (getrusage(RUSAGE_SELF, &ru) == 0)
            fprintf(err, "strain_detect timing: CPU time of the process: user %.1f s + system %.1f s; of it parser threads %.1f s, reader threads %.1f s, "
                         "lanes %.1f s, this (main) thread %.1f s -- the rest is the output compressors, the HIP runtime's own threads and the strains' opening\n",
                    (double)ru.ru_utime.tv_sec + 1e-6 * (double)ru.ru_utime.tv_usec, (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec,
                    cpu_parsers, cpu_readers, cpu_lanes, thread_cpu_s());
    }
    if (getenv("SK_SD_TIMING")) {                        /* ... and the threads that are still there (the runtime's, the compressors', the lanes'), by name */
        DIR *td = opendir("/proc/self/task");
        struct dirent *de;
        const double tick = 1.0 / (double)sysconf(_SC_CLK_TCK);
        while (td && (de = readdir(td)) != NULL) {
            char path[300], buf[1024], *rp;
            FILE *tf;
            unsigned long ut = 0, stt = 0;
            if (de->d_name[0] == '.') continue;
            snprintf(path, sizeof path, "/proc/self/task/%s/stat", de->d_name);
            if (!(tf = fopen(path, "r"))) continue;
            if (fgets(buf, sizeof buf, tf) && (rp = strrchr(buf, ')')) != NULL &&
                sscanf(rp + 2, "%*c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %lu %lu", &ut, &stt) == 2 && (double)(ut + stt) * tick >= 0.5) {
                *rp = 0;
                fprintf(err, "strain_detect timing: thread %s (%s: user %.1f s, system %.1f s\n", de->d_name, strchr(buf, '(') ? strchr(buf, '(') + 1 : "?",
                        (double)ut * tick, (double)stt * tick);
            }
            fclose(tf);
        }
        if (td) closedir(td);
    }
    if (getenv("SK_SD_TIMING")) {                        /* was the process held back by its CPU quota?  (cgroup v2: cpu.stat of the job's group) */
        FILE *cs = fopen("/sys/fs/cgroup/cpu.stat", "r");
        char ln[128];
        while (cs && fgets(ln, sizeof ln, cs))
            if (!strncmp(ln, "nr_throttled", 12) || !strncmp(ln, "throttled_usec", 14) || !strncmp(ln, "usage_usec", 10)) {
                ln[strcspn(ln, "\n")] = 0;
                fprintf(err, "strain_detect timing: cgroup cpu.stat %s (since the group began)\n", ln);
            }
        if (cs) fclose(cs);
    }
    if (getenv("SK_SD_TIMING"))
        fprintf(err, "strain_detect timing: page-locked chunk buffers: %d of at most %d made (%zu bytes each), %lu chunks had to do without one\n",
                sd_pin.total, SD_PIN_MAX, sd_pin.bytes, n_unpinned_chunks);
    if (getenv("SK_SD_TIMING") && t_pw_parse > 0)
        fprintf(err, "strain_detect timing: parser threads, summed: parsing %.2f s, waiting for a segment %.2f s, for their turn to hand chunks on %.2f s, for room in the queue %.2f s\n",
                t_pw_parse, t_pw_seg, t_pw_turn, t_pw_push);
    sd_pin_close();
    {   /* closing a strain = finishing its gz output, freeing its device context and tables: strain by strain on threads */
        sd_pool cp;
        const double t0 = now_s();
        pool_start(&cp, ns);
        pool_run(&cp, ns, close_one, p);
        pool_stop(&cp);
        t_close = now_s() - t0;
        if (getenv("SK_SD_TIMING")) fprintf(err, "strain_detect timing: closing the strains %.2f s\n", t_close);
    }
    skzo_pool_stop(&zpool);
    sd_zpool = NULL;
    for (s = 0; paths && s < 4 * ns; s++) free(paths[s]);
    free(paths);
    free(p);
    return status;
}
