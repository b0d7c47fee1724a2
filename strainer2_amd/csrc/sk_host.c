/* sk_host.c -- host layer (plain C) of libstrainer_kmer.so.
 *
 * Keeps the reference's FILE semantics around the device layer:
 *   - FASTA/FASTQ(.gz) record grammar of the reference's parser (src/kseq.h:166-211), here as a
 *     push-style state machine over inflated blocks;
 *   - build phase (src/genome_compare.c:967-1030): oriented keys of the strain, first-occurrence
 *     order, column-0 multiplicity -- keys are 62-bit packed integers, wide (non-ACGT) keys are
 *     kept as bytes;
 *   - BIO_hash slot-order replay (src/BIO_hash.c:39-61,129-139,208-216) so that rows come out in
 *     exactly the reference's order;
 *   - file-list walk, progress log, skip rule, error texts (src/genome_compare.c:115-177);
 *   - TSV printing (src/kmer_scrub_count.c:134-156) and the program's argv contract (:29-131).
 *
 * No window is ever looked up on the CPU here: lookups happen in sk_scan_grid / sk_scan_wide.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <errno.h>
#include <pthread.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
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
#include "sk_pack.h"
#include "sk_ctxjob.h"
#include "sk_gzpipe.h"
#include "sk_cpus.h"
#include "sk_rendezvous.h"
#include "sk_internal.h"

int sk_rendezvous_exchange(int rank, int world, const char *base_path, int my_status, unsigned char *payload128, double timeout_s)
{
    return skr_exchange(rank, world, base_path, my_status, payload128, timeout_s);
}

/* tables of tens of MB that are touched at random: 2 MiB-aligned and offered to transparent huge pages
 * (fewer page faults while they are filled, fewer TLB misses while they are probed) */
static void *big_alloc(size_t bytes)
{
    void *p = NULL;
    if (bytes < (8u << 20)) return malloc(bytes);
    if (posix_memalign(&p, 2u << 20, (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1))) return malloc(bytes);
    madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }


/* run a whole (possibly gzipped) file through the parser */
/* decoded bytes of a gzip file straight into the record parser; stops the inflate once the parser has stopped */
static int parse_feed_sink(void *user, const unsigned char *data, size_t n)
{
    parser *ps = (parser *)user;
    parser_feed(ps, data, n);
    return ps->state == P_STOP;
}

/* ---- the host-side pre-pack (sk_pack.h) behind the C-ABI ---- */
uint64_t sk_packed_bytes(uint64_t nbytes) { return ((nbytes + 15u) >> 4) * 6u; }

int sk_pack_stream(const uint8_t *stream, uint64_t nbytes, void *packed, int *odd)
{
    static skp_pack_fn picked = NULL;
    skp_pack_fn fn = __atomic_load_n(&picked, __ATOMIC_RELAXED);
    const uint64_t nch = (nbytes + 15u) >> 4;
    int o = 0;
    if ((!stream && nbytes) || (!packed && nbytes) || !odd) return SK_E_ARG;
    if (!fn) { fn = skp_pack_pick(); __atomic_store_n(&picked, fn, __ATOMIC_RELAXED); }       /* (idempotent: two threads store the same pointer) */
    fn(stream, nbytes, (uint32_t *)packed, (uint16_t *)((uint8_t *)packed + nch * 4u), &o);
    *odd = o;
    return SK_OK;
}

/* pipe: inflate on a helper thread while this one parses (worth it when there are fewer files than cores);
 * pipe > 1: the helper inflates each gzip member with that many threads (sk_gzpar.h) */
/* how a plain file's bytes are got at: 0 = its pages populated into a mapping block by block (default), 1 = read() (SK_LIST_INPUT=read,
 * or set by the first thread that finds the kernel without MADV_POPULATE_READ) */
static int list_input_mode = 0;
static int list_input_read(void)
{
    const char *e = getenv("SK_LIST_INPUT");
    return __atomic_load_n(&list_input_mode, __ATOMIC_RELAXED) || (e && !strcmp(e, "read"));
}

/* this thread is parsing PLAIN text (set by parse_file / parse_range, read by the list scan's sink): only then are its chunks packed
 * before they go up -- a .gz item is bound by its inflate on these same CPUs, the link idles, and packing would cost it 7 % */
static __thread int tl_plain_text;

static int parse_file(const char *path, rec_fn fn, void *user, int64_t *nrecords, int *sink_rc, int pipe)
{
    enum { BLK = 1 << 20 };
    gzFile g;
    unsigned char *blk;
    parser ps;
    int got, zrc;
    parser_init(&ps, fn, user);
    tl_plain_text = 0;
    /* gzip files go through the library's own inflate (sk_gzfast.h, about twice zlib's rate); anything else
     * -- plain text, which gzread passes through, or SK_ZLIB=1 -- through zlib */
    if (getenv("SK_ZLIB")) zrc = SKZ_NOT_GZIP;
    else if (pipe) {
        skzp zp;
        zrc = skzp_open_threads(&zp, path, pipe);               /* pipe > 1: that many threads inflate the one file */
        if (zrc == SKZ_OK) {
            const unsigned char *data;
            size_t n;
            while (ps.state != P_STOP && (n = skzp_next(&zp, &data)) > 0) parser_feed(&ps, data, n);
            skzp_close(&zp);
        }
    } else zrc = skz_decode_file(path, parse_feed_sink, &ps);
    if (zrc == SKZ_OPEN) { parser_free(&ps); return SK_E_OPEN; }
    if (zrc == SKZ_NOT_GZIP && !getenv("SK_ZLIB")) {
        /* plain text: read() in 4 MiB blocks (round 4).  zlib's pass-through does the same copy behind more layers and cuts
         * the text every MiB (the parser's whole-record shortcuts want their records inside ONE block).  Parsing out of a
         * mapping of the file was measured and LOSES with 16 decode threads (BASELINE configs[2], 1,670 files in /dev/shm,
         * gpurun_out/r04/cfg3_where.txt -> profiles/r04_cfg3_where.txt: the threads spend 31 s on a core and 37 s asleep in
         * page faults behind the address space's lock, which every mmap/munmap of the other threads takes for writing --
         * scans 4.6 s; read(): 42 s on a core, next to none asleep -- scans 3.4 s).  What is not a regular file goes
         * through gzread below, which hands the same bytes on */
        const int fd = open(path, O_RDONLY);
        struct stat sb;
        size_t resume_at = 0;
        tl_plain_text = 1;
        /* Later in round 4: the mapping WITHOUT its two costs.  A block's pages are put into the address space by one call
         * (MADV_POPULATE_READ: no trap per 64 KiB), parsed where they lie, and taken out again by one call (MADV_DONTNEED) -- the
         * lock is only ever taken for reading, and munmap finds nothing to tear down (that teardown under the lock for WRITING is
         * what the other threads' page faults waited for).  No copy: once the chunks went up packed and the scan was bound by
         * its CPUs, a sampling of the program counter had 56 % of the decode threads' time inside read()
         * (tools/probes/sigprof_preload.c, profiles/r04_sigprof.txt).  SK_LIST_INPUT=read is the copying way, also taken where the
         * kernel has no MADV_POPULATE_READ (before 5.14) or the pages cannot be had (a file that shrank) */
#ifdef MADV_POPULATE_READ
        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0 && !list_input_read()) {
            const size_t n = (size_t)sb.st_size, PBLK = (size_t)4 << 20;
            unsigned char *t = (unsigned char *)mmap(NULL, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (t != MAP_FAILED) {
                size_t at = 0;
                while (at < n && ps.state != P_STOP) {
                    const size_t take = n - at < PBLK ? n - at : PBLK;
                    const size_t span = (take + 4095u) & ~(size_t)4095u;          /* (PBLK is a multiple of the page size: `at` stays page-aligned) */
                    if (madvise(t + at, span, MADV_POPULATE_READ) != 0) {
                        if (errno == EINVAL) __atomic_store_n(&list_input_mode, 1, __ATOMIC_RELAXED);
                        break;
                    }
                    parser_feed(&ps, t + at, take);
                    (void)madvise(t + at, span, MADV_DONTNEED);
                    at += take;
                }
                munmap(t, n);
                if (at >= n || ps.state == P_STOP) zrc = SKZ_OK;
                else resume_at = at;
            }
        }
#endif
        if (zrc == SKZ_NOT_GZIP && fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && lseek(fd, (off_t)resume_at, SEEK_SET) == (off_t)resume_at) {
            const char *e = getenv("SK_READ_BLOCK");                /* (64 bytes -- tests -- .. 32 MiB; default 4 MiB) */
            const long long ev = e ? atoll(e) : 0;
            const size_t RBLK = ev >= 64 && ev <= (32 << 20) ? (size_t)ev : (size_t)4 << 20;
            unsigned char *rb = (unsigned char *)malloc(RBLK);
            if (rb) {
                (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
                while (ps.state != P_STOP) {
                    size_t have = 0;
                    ssize_t r = 1;
                    while (have < RBLK && (r = read(fd, rb + have, RBLK - have)) != 0) {
                        if (r < 0) { if (errno == EINTR) continue; break; }
                        have += (size_t)r;
                    }
                    if (have) parser_feed(&ps, rb, have);
                    if (r <= 0) break;                  /* (end of file, or an error: what was read counts, as with gzread) */
                }
                free(rb);
                zrc = SKZ_OK;
            }
        }
        if (fd >= 0) close(fd);
    }
    if (zrc == SKZ_NOT_GZIP) {
        g = gzopen(path, "r");
        if (!g) { parser_free(&ps); return SK_E_OPEN; }
        gzbuffer(g, 1 << 18);
        blk = (unsigned char *)malloc(BLK);
        while (ps.state != P_STOP && (got = gzread(g, blk, BLK)) > 0) parser_feed(&ps, blk, (size_t)got);
        free(blk);
        gzclose(g);
    }
    if (ps.state != P_STOP) parser_eof(&ps);
    if (nrecords) *nrecords = ps.nrecords;
    if (sink_rc) *sink_rc = ps.sink_rc;
    parser_free(&ps);
    return SK_OK;
}

static int parse_memory(const char *text, size_t n, rec_fn fn, void *user)
{
    /* "stream" form: every '\n'-separated line is one decoded record */
    size_t pos = 0;
    char *tmp = NULL; size_t cap = 0;
    int rc = 0;
    while (pos <= n) {
        const char *nl = pos < n ? (const char *)memchr(text + pos, '\n', n - pos) : NULL;
        size_t len = nl ? (size_t)(nl - (text + pos)) : n - pos;
        if (!nl && len == 0 && pos > 0) break;          /* nothing after the final separator */
        grow(&tmp, &cap, len);
        memcpy(tmp, text + pos, len);
        tmp[len] = '\0';
        rc = fn(user, tmp, len);
        if (rc || !nl) break;
        pos += len + 1;
    }
    free(tmp);
    return rc;
}

/* =========================================================================================
 * record stream writer: packs decoded records into '\n'-separated chunks for the device
 * ======================================================================================= */

typedef struct {
    uint8_t    *buf;
    uint64_t    cap, len;
    skh_sink_fn sink;
    void       *user;
    uint8_t  *(*next_buf)(void *user);   /* optional: buffer to fill after a flush (double buffering) */
    uint64_t    bases;
    int         rc;
} stream_writer;

static int writer_flush(stream_writer *w)
{
    if (w->len && !w->rc) {
        w->rc = w->sink(w->user, w->buf, w->len);
        if (!w->rc && w->next_buf) w->buf = NULL;    /* the next buffer is asked for when a record needs it: not after the last flush */
    }
    w->len = 0;
    return w->rc;
}

static int writer_record(void *user, char *seq, size_t len)
{
    stream_writer *w = (stream_writer *)user;
    size_t off = 0;
    w->bases += len;
    if (len < SK_K) return 0;                        /* src/genome_compare.c:204: no window fits */
    if (w->rc) return w->rc;
    while (off < len) {
        uint64_t room, take;
        if (w->cap - w->len < 2 * SK_K + 2 && writer_flush(w)) return w->rc;
        if (!w->buf && (!w->next_buf || !(w->buf = w->next_buf(w->user)))) return w->rc = SK_E_NOMEM;
        room = w->cap - w->len - 1;
        take = len - off;
        if (take > room) take = room;
        memcpy(w->buf + w->len, seq + off, take);
        w->len += take;
        w->buf[w->len++] = '\n';
        off += take;
        if (off < len) off -= SK_OVERLAP;            /* cut record: next piece re-reads k-1 bytes */
    }
    return 0;
}

int64_t skh_decode_file(const char *path, uint64_t chunk_bytes, skh_sink_fn sink, void *user, uint64_t *bases)
{
    stream_writer w;
    int64_t nrec = 0;
    int rc;
    if (chunk_bytes < 4096) chunk_bytes = 4096;
    memset(&w, 0, sizeof w);
    w.buf = (uint8_t *)malloc(chunk_bytes);
    w.cap = chunk_bytes;
    w.sink = sink;
    w.user = user;
    rc = parse_file(path, writer_record, &w, &nrec, NULL, 0);
    if (rc == SK_OK) writer_flush(&w);
    if (bases) *bases += w.bases;
    free(w.buf);
    if (rc != SK_OK) return rc;
    if (w.rc) return w.rc;
    return nrec;
}

/* =========================================================================================
 * build phase: oriented keys in first-occurrence order
 * ======================================================================================= */

#define WIDE_FLAG 0x8000000000000000ull     /* order-list entry refers to wide key (index in low bits) */

#define SK_RING 16u
typedef struct {
    /* packed-key set: open addressing on u64 -> order index */
    uint64_t *pk; uint32_t *pv; uint64_t pmask; uint64_t pcount;
    /* wide-key set */
    char *wkeys; uint32_t *wv; uint32_t wn, wcap; uint32_t *windex; uint32_t wmask;
    /* insertion order */
    uint64_t *order; uint32_t *count; uint8_t *fwd_first; uint32_t *firstpos; uint32_t n, cap;
    uint32_t default_val, incr;
    /* the strain's text, 2 bits per base (16 per word, first base on top), records end to end: the scan's
     * seed-and-verify stage compares reads with it (sk_table_load_text) */
    uint32_t *text2; uint64_t text_n, text_cap_words;
    uint64_t short_records;
    signed char comp[256];
    /* packed keys wait here for SK_RING more windows while their table line is fetched: the distinct-key
     * set of a 5 Mbp strain (100+ MB) lives in DRAM and every insert is a cache miss otherwise */
    uint64_t ring_key[SK_RING]; uint8_t ring_fwd[SK_RING]; uint32_t ring_tpos[SK_RING]; uint32_t ring_n, ring_pos;
} builder;

#define NO_POS 0xFFFFFFFFu

/* hint_bases: about how many bases the strain has (0 = unknown): the distinct-key set starts at twice
 * that many slots instead of growing there by repeated rehashing */
static void builder_init(builder *b, uint32_t default_val, uint32_t incr, uint64_t hint_bases)
{
    memset(b, 0, sizeof *b);
    b->pmask = (1u << 16) - 1;
    while (b->pmask + 1 < 2 * hint_bases && b->pmask < (1ull << 30) - 1) b->pmask = b->pmask * 2 + 1;
    b->pk = (uint64_t *)big_alloc((b->pmask + 1) * sizeof(uint64_t));
    b->pv = (uint32_t *)big_alloc((b->pmask + 1) * sizeof(uint32_t));
    memset(b->pk, 0xFF, (b->pmask + 1) * sizeof(uint64_t));
    b->wmask = 63;
    b->windex = (uint32_t *)calloc(b->wmask + 1, sizeof(uint32_t));
    b->default_val = default_val;
    b->incr = incr;
    sk_fill_complement(b->comp);
}

static void builder_free(builder *b)
{
    free(b->pk); free(b->pv); free(b->wkeys); free(b->wv); free(b->windex);
    free(b->order); free(b->count); free(b->fwd_first); free(b->firstpos); free(b->text2);
}

static uint32_t builder_append(builder *b, uint64_t entry)
{
    if (b->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 1 << 16;
        b->order = (uint64_t *)realloc(b->order, (size_t)b->cap * sizeof(uint64_t));
        b->count = (uint32_t *)realloc(b->count, (size_t)b->cap * sizeof(uint32_t));
        b->fwd_first = (uint8_t *)realloc(b->fwd_first, (size_t)b->cap);
        b->firstpos = (uint32_t *)realloc(b->firstpos, (size_t)b->cap * sizeof(uint32_t));
    }
    b->order[b->n] = entry;
    b->count[b->n] = b->default_val;
    b->fwd_first[b->n] = 0;
    b->firstpos[b->n] = NO_POS;
    return b->n++;
}

static void builder_grow_packed(builder *b)
{
    uint64_t oldn = b->pmask + 1, i, nm = oldn * 2 - 1;
    uint64_t *ok = b->pk; uint32_t *ov = b->pv;
    b->pk = (uint64_t *)big_alloc((nm + 1) * sizeof(uint64_t));
    b->pv = (uint32_t *)big_alloc((nm + 1) * sizeof(uint32_t));
    memset(b->pk, 0xFF, (nm + 1) * sizeof(uint64_t));
    for (i = 0; i < oldn; i++) {
        uint64_t s;
        if (ok[i] == SK_EMPTY64) continue;
        s = sk_hash62(ok[i]) & nm;
        while (b->pk[s] != SK_EMPTY64) s = (s + 1) & nm;
        b->pk[s] = ok[i]; b->pv[s] = ov[i];
    }
    b->pmask = nm;
    free(ok); free(ov);
}

/* is_fwd: the key is the strain's text itself at this occurrence (not its reverse complement);
 * tpos: text position of the window's first base when the window is all ACGT, else NO_POS */
static void builder_insert_packed(builder *b, uint64_t key, int is_fwd, uint32_t tpos)
{
    uint64_t s = sk_hash62(key) & b->pmask;
    while (b->pk[s] != SK_EMPTY64) {
        if (b->pk[s] == key) { b->count[b->pv[s]] += b->incr; return; }
        s = (s + 1) & b->pmask;
    }
    b->pk[s] = key;
    b->pv[s] = builder_append(b, key);
    b->fwd_first[b->pv[s]] = (uint8_t)is_fwd;
    b->firstpos[b->pv[s]] = tpos;
    if (++b->pcount * 2 > b->pmask) builder_grow_packed(b);
}

/* queue a packed key: prefetch its table line now, insert it SK_RING windows later (order is kept) */
static void builder_add_packed(builder *b, uint64_t key, int is_fwd, uint32_t tpos)
{
    uint32_t at;
    if (b->ring_n == SK_RING) {
        builder_insert_packed(b, b->ring_key[b->ring_pos], b->ring_fwd[b->ring_pos], b->ring_tpos[b->ring_pos]);
        b->ring_pos = (b->ring_pos + 1u) & (SK_RING - 1u);
        b->ring_n--;
    }
    at = (b->ring_pos + b->ring_n) & (SK_RING - 1u);
    {
        const uint64_t s0 = sk_hash62(key) & b->pmask;
        __builtin_prefetch(&b->pk[s0], 1);
        __builtin_prefetch(&b->pv[s0], 1);
    }
    b->ring_key[at] = key;
    b->ring_fwd[at] = (uint8_t)is_fwd;
    b->ring_tpos[at] = tpos;
    b->ring_n++;
}

static void builder_drain(builder *b)
{
    while (b->ring_n) {
        builder_insert_packed(b, b->ring_key[b->ring_pos], b->ring_fwd[b->ring_pos], b->ring_tpos[b->ring_pos]);
        b->ring_pos = (b->ring_pos + 1u) & (SK_RING - 1u);
        b->ring_n--;
    }
}

static void builder_add_wide(builder *b, const char *key31)
{
    builder_drain(b);                                 /* wide and packed entries share one insertion order */
    uint32_t s = sk_hash_wide(key31) & b->wmask, i;
    while (b->windex[s]) {
        uint32_t e = b->windex[s] - 1;
        if (memcmp(b->wkeys + (size_t)e * 32, key31, SK_K) == 0) { b->count[b->wv[e]] += b->incr; return; }
        s = (s + 1) & b->wmask;
    }
    if (b->wn == b->wcap) {
        b->wcap = b->wcap ? b->wcap * 2 : 64;
        b->wkeys = (char *)realloc(b->wkeys, (size_t)b->wcap * 32);
        b->wv = (uint32_t *)realloc(b->wv, (size_t)b->wcap * sizeof(uint32_t));
    }
    memcpy(b->wkeys + (size_t)b->wn * 32, key31, SK_K);
    b->wkeys[(size_t)b->wn * 32 + SK_K] = '\0';
    b->wv[b->wn] = builder_append(b, WIDE_FLAG | b->wn);
    b->windex[s] = ++b->wn;
    if (b->wn * 2 > b->wmask) {                      /* rebuild the small index */
        uint32_t nm = b->wmask * 2 + 1;
        free(b->windex);
        b->windex = (uint32_t *)calloc((size_t)nm + 1, sizeof(uint32_t));
        for (i = 0; i < b->wn; i++) {
            uint32_t t = sk_hash_wide(b->wkeys + (size_t)i * 32) & nm;
            while (b->windex[t]) t = (t + 1) & nm;
            b->windex[t] = i + 1;
        }
        b->wmask = nm;
    }
}

/* one strain record: every window, rolling 2-bit pack; windows with a non-ACGT byte that is
 * not a hard breaker take the byte-string route (src/genome_compare.c:1000-1024) */
static int builder_record(void *user, char *seq, size_t len)
{
    builder *b = (builder *)user;
    uint64_t fwd = 0, rc = 0;
    uint32_t run = 0, soft = 0;
    size_t i;
    if (len + 1 < SK_K) { b->short_records++; return 0; }   /* reference: size_t underflow, crash */
    /* room for this record in the 2-bit text (positions beyond 2^32 - 2 are not recorded: first_pos stays NO_POS) */
    if ((b->text_n + len + 64) / 16 + 1 > b->text_cap_words) {
        uint64_t want = b->text_cap_words ? b->text_cap_words * 2 : 1u << 16;
        while (want < (b->text_n + len + 64) / 16 + 1) want *= 2;
        b->text2 = (uint32_t *)realloc(b->text2, want * sizeof(uint32_t));
        memset(b->text2 + b->text_cap_words, 0, (want - b->text_cap_words) * sizeof(uint32_t));
        b->text_cap_words = want;
    }
    const uint64_t tbase = b->text_n;
    for (i = 0; i < len; i++) {
        uint32_t c = (uint8_t)seq[i], code = sk_code(c);
        b->text2[(tbase + i) >> 4] |= (code & 3u) << (2u * (15u - (uint32_t)((tbase + i) & 15u)));
        fwd = ((fwd << 2) | code) & SK_KMASK62;
        rc = (rc >> 2) | ((uint64_t)(3u - code) << 60);
        run = sk_is_acgt(c) ? run + 1 : 0;
        soft = sk_is_hard_break(c) ? 0 : soft + 1;
        if (run >= SK_K) {
            const uint64_t tp = tbase + i - (SK_K - 1);
            builder_add_packed(b, fwd > rc ? fwd : rc, fwd > rc, tp < 0x7FFFFF00u ? (uint32_t)tp : NO_POS);
        } else if (soft >= SK_K) {
            char u[SK_K], o[SK_K + 1];
            const char *w = seq + i - (SK_K - 1);
            int j, sign = 0, pure = 1;
            for (j = 0; j < SK_K; j++) u[j] = (char)sk_upper((uint8_t)w[j]);
            for (j = 0; j < SK_K && !sign; j++) {
                signed char f = (signed char)u[j], r = b->comp[(uint8_t)u[SK_K - 1 - j]];
                sign = (f > r) - (r > f);
            }
            if (sign >= 0) memcpy(o, u, SK_K);
            else for (j = 0; j < SK_K; j++) o[SK_K - 1 - j] = (char)b->comp[(uint8_t)u[j]];
            o[SK_K] = '\0';
            for (j = 0; j < SK_K; j++) pure &= (o[j] == 'A') | (o[j] == 'C') | (o[j] == 'G') | (o[j] == 'T');
            if (pure) {                              /* e.g. U in the strain whose revcomp wins */
                uint64_t key = 0;
                for (j = 0; j < SK_K; j++) key = (key << 2) | sk_code((uint8_t)o[j]);
                builder_add_packed(b, key, sign >= 0, NO_POS);
            } else {
                builder_add_wide(b, o);
            }
        }
    }
    b->text_n = tbase + len;
    return 0;
}

/* ---- BIO_hash order replay ------------------------------------------------------------- */

static void decode_key(uint64_t key, char out[32])
{
    int i;
    for (i = 0; i < SK_K; i++) out[i] = "ACGT"[(key >> (2 * (SK_K - 1 - i))) & 3];
    out[SK_K] = '\0';
}

static uint32_t djb2_bytes(const char *s)           /* src/BIO_hash.c:208-216, signed bytes */
{
    uint32_t h = 5381u;
    for (; *s; s++) h = h * 33u + (uint32_t)(int32_t)(signed char)*s;
    return h;
}

/* djb2 of the 31 letters of a packed key without the 31 dependent multiplies:
 * h = 5381 * 33^31 + sum_i letter_i * 33^(30-i)  (mod 2^32), four letters (one byte of codes) per table lookup */
static uint32_t djb2_tab[8][256], djb2_base;
static pthread_once_t djb2_once = PTHREAD_ONCE_INIT;
static void djb2_init(void)
{
    uint32_t pw[SK_K + 1], g, v, j;
    pw[0] = 1u;
    for (j = 1; j <= SK_K; j++) pw[j] = pw[j - 1] * 33u;
    djb2_base = 5381u * pw[SK_K];
    /* the key holds letter i (0 = first) at bits 61-2i..60-2i; byte group g covers bits 8g+7..8g, i.e.
     * letters 30-4g-3 .. 30-4g (group 7 only has the first 3 letters, bits 61..56) */
    for (g = 0; g < 8; g++)
        for (v = 0; v < 256; v++) {
            uint32_t sum = 0;
            for (j = 0; j < 4; j++) {
                const int letter = 30 - 4 * (int)g - (int)j;          /* letter held in bits 2j+1..2j of the byte */
                if (letter < 0) continue;
                sum += (uint32_t)"ACGT"[(v >> (2 * j)) & 3u] * pw[30 - letter];
            }
            djb2_tab[g][v] = sum;
        }
}
static inline uint32_t djb2_packed(uint64_t key)
{
    return djb2_base + djb2_tab[0][key & 255u] + djb2_tab[1][(key >> 8) & 255u] + djb2_tab[2][(key >> 16) & 255u] +
           djb2_tab[3][(key >> 24) & 255u] + djb2_tab[4][(key >> 32) & 255u] + djb2_tab[5][(key >> 40) & 255u] +
           djb2_tab[6][(key >> 48) & 255u] + djb2_tab[7][(key >> 56) & 63u];
}

/* Replays insert-with-doubling and returns entries in ascending slot order of the final table.
 * The slot array (64 MB for a 5 Mbp strain) is probed at random: the line of the insert 16 ahead is
 * prefetched while the current one is placed. */
static uint32_t *replay_slot_order(const builder *b, uint32_t initial_slots, uint32_t *final_slots)
{
    enum { AHEAD = 48 };
    uint32_t M = initial_slots, N = 0, e, i;
    uint32_t *h32 = (uint32_t *)malloc((size_t)(b->n ? b->n : 1) * sizeof(uint32_t));
    int32_t *slot;
    uint32_t *rows;
    if (M == 0) M = 1000; else if (M < 10) M = 10;       /* src/BIO_hash.c:18-21 */
    pthread_once(&djb2_once, djb2_init);
    for (e = 0; e < b->n; e++) {
        if (b->order[e] & WIDE_FLAG) h32[e] = djb2_bytes(b->wkeys + (size_t)(b->order[e] & 0xFFFFFFFFu) * 32);
        else h32[e] = djb2_packed(b->order[e]);
    }
    slot = (int32_t *)big_alloc((size_t)M * sizeof(int32_t));
    memset(slot, 0xFF, (size_t)M * sizeof(int32_t));
    for (e = 0; e < b->n; e++) {
        uint32_t s = h32[e] % M;
        if (e + AHEAD < b->n) __builtin_prefetch(&slot[h32[e + AHEAD] % M], 1);
        while (slot[s] >= 0) s = s + 1 == M ? 0 : s + 1;
        slot[s] = (int32_t)e;
        if (N++ >= M / 2) {                              /* post-increment test: src/BIO_hash.c:138 */
            uint32_t M2 = M + M;
            int32_t *ns = (int32_t *)big_alloc((size_t)M2 * sizeof(int32_t));
            memset(ns, 0xFF, (size_t)M2 * sizeof(int32_t));
            N = 0;
            for (i = 0; i < M; i++) {                    /* old slot order: src/BIO_hash.c:54-58 */
                uint32_t t;
                if (i + 160 < M && slot[i + 160] >= 0) __builtin_prefetch(&h32[slot[i + 160]]);
                if (i + 64 < M && slot[i + 64] >= 0) __builtin_prefetch(&ns[h32[slot[i + 64]] % M2], 1);
                if (slot[i] < 0) continue;
                t = h32[slot[i]] % M2;
                while (ns[t] >= 0) t = t + 1 == M2 ? 0 : t + 1;
                ns[t] = slot[i];
                N++;
            }
            free(slot);
            slot = ns;
            M = M2;
        }
    }
    rows = (uint32_t *)malloc(((size_t)b->n + 1) * sizeof(uint32_t));
    for (i = 0, e = 0; i < M; i++) { rows[e] = (uint32_t)slot[i]; e += (uint32_t)(slot[i] >= 0); }   /* (branch-free compaction) */
    free(slot);
    free(h32);
    *final_slots = M;
    return rows;
}

static int keyset_finish(skh_keyset *ks, builder *b, uint32_t initial_slots)
{
    uint32_t r, *rows;
    builder_drain(b);
    if (initial_slots == SK_ROWS_IN_STRAIN_ORDER) {      /* rows as they first occur along the strain: no replay of the reference's table */
        rows = (uint32_t *)malloc(((size_t)b->n + 1) * sizeof(uint32_t));
        for (r = 0; r < b->n; r++) rows[r] = r;
        ks->final_slots = 0;
    } else rows = replay_slot_order(b, initial_slots, &ks->final_slots);
    uint32_t *wide_newrow = (uint32_t *)calloc(b->wn ? b->wn : 1, sizeof(uint32_t));
    ks->nrows = b->n;
    ks->nwide = b->wn;
    ks->short_records = b->short_records;
    ks->packed = (uint64_t *)malloc((size_t)(b->n ? b->n : 1) * sizeof(uint64_t));
    ks->first_count = (uint32_t *)malloc((size_t)(b->n ? b->n : 1) * sizeof(uint32_t));
    ks->locality = (uint32_t *)malloc((size_t)(b->n ? b->n : 1) * sizeof(uint32_t));
    ks->wide_keys = (char *)malloc((size_t)(b->wn ? b->wn : 1) * 32);
    ks->wide_rows = (uint32_t *)malloc((size_t)(b->wn ? b->wn : 1) * sizeof(uint32_t));
    ks->first_pos = (uint32_t *)malloc((size_t)(b->n ? b->n : 1) * sizeof(uint32_t));
    /* locality order = the device's counter order: entries with a text position first, in text order (that is
     * their insertion order), then the others (wide keys, keys met only through a U) in insertion order --
     * the contract of sk_table_load_text */
    uint32_t *newloc = (uint32_t *)malloc((size_t)(b->n ? b->n : 1) * sizeof(uint32_t));
    {
        uint32_t e, m = 0;
        for (e = 0; e < b->n; e++) if (b->firstpos[e] != NO_POS) newloc[e] = m++;
        for (e = 0; e < b->n; e++) if (b->firstpos[e] == NO_POS) newloc[e] = m++;
    }
    ks->text_bases = b->text_n < 0x7FFFFF00u ? (uint32_t)b->text_n : 0u;       /* 0: too long (a table slot has 31 bits for a position): no text stage */
    ks->text2 = NULL;
    if (ks->text_bases) {
        const size_t words = (size_t)ks->text_bases / 16 + 4;
        ks->text2 = (uint32_t *)calloc(words, sizeof(uint32_t));
        memcpy(ks->text2, b->text2, ((size_t)ks->text_bases + 15) / 16 * sizeof(uint32_t));
    }
    for (r = 0; r < b->n; r++) {
        uint64_t ent;
        if (r + 32 < b->n) { __builtin_prefetch(&b->order[rows[r + 32]]); __builtin_prefetch(&b->count[rows[r + 32]]); __builtin_prefetch(&b->fwd_first[rows[r + 32]]); }
        ent = b->order[rows[r]];
        ks->first_count[r] = b->count[rows[r]];
        ks->locality[r] = newloc[rows[r]] | (b->fwd_first[rows[r]] ? SK_LOCALITY_FWD : 0u);
        ks->first_pos[r] = ks->text_bases ? b->firstpos[rows[r]] : NO_POS;
        if (ent & WIDE_FLAG) { ks->packed[r] = SK_KEY_NONE; wide_newrow[ent & 0xFFFFFFFFu] = r; }
        else ks->packed[r] = ent;
    }
    for (r = 0; r < b->wn; r++) {
        memcpy(ks->wide_keys + (size_t)r * 32, b->wkeys + (size_t)r * 32, 32);
        ks->wide_rows[r] = wide_newrow[r];
    }
    free(wide_newrow);
    free(rows);
    free(newloc);
    return SK_OK;
}

int skh_keyset_from_file(skh_keyset *ks, const char *path, uint32_t initial_slots, uint32_t default_val, uint32_t incr)
{
    builder b;
    int rc;
    if (!ks || !path) return SK_E_ARG;
    memset(ks, 0, sizeof *ks);
    {
        struct stat st;
        const size_t pl = strlen(path);
        uint64_t hint = 0;
        if (stat(path, &st) == 0 && S_ISREG(st.st_mode))
            hint = (uint64_t)st.st_size * (pl > 3 && !strcmp(path + pl - 3, ".gz") ? 4u : 1u);
        if (hint > (1ull << 28)) hint = 1ull << 28;
        builder_init(&b, default_val, incr, hint);
    }
    {
        const double t0 = now_s();
        double t1;
        rc = parse_file(path, builder_record, &b, NULL, NULL, 0);
        t1 = now_s();
        if (rc == SK_OK) rc = keyset_finish(ks, &b, initial_slots);
        if (getenv("SK_TIMING"))
            fprintf(stderr, "key set of %s: decode + distinct k-mers %.2f s, row order replay + layout %.2f s (%u keys)\n",
                    path, t1 - t0, now_s() - t1, b.n);
    }
    builder_free(&b);
    return rc;
}

int skh_keyset_from_stream(skh_keyset *ks, const char *stream, size_t nbytes, uint32_t initial_slots,
                           uint32_t default_val, uint32_t incr)
{
    builder b;
    int rc;
    if (!ks || (!stream && nbytes)) return SK_E_ARG;
    memset(ks, 0, sizeof *ks);
    builder_init(&b, default_val, incr, nbytes);
    parse_memory(stream, nbytes, builder_record, &b);
    rc = keyset_finish(ks, &b, initial_slots);
    builder_free(&b);
    return rc;
}

/* ---- the key set built on the DEVICE (strain_detect's opening, round 3) ----------------------------------------------------
 * The host's part shrinks to what needs the file: the reference's record grammar (src/kseq.h:166-211), the strain's bases packed
 * 2 bits each, records end to end, and one bit per position "a window of 31 A/C/G/T bases of ONE record starts here" -- the
 * windows src/genome_compare.c:1000-1019 turns into keys (N breaks a window, :1007; records shorter than k - 1 are skipped and
 * counted as the host builder does).  sk_table_build_from_text makes keys, first occurrences, row numbers, rank map, filters and
 * column 0 from that; the keys come back once, in row order, for the hit lines' k-mer text.  A strain with any other letter (U,
 * IUPAC: byte-string keys) is not for this path: SK_E_UNSUPPORTED, and the caller builds the key set on the host as before. */
typedef struct { uint32_t *text2, *ok; uint64_t n, cap_words, ok_words; uint64_t nstarts, short_records; int soft; } dk_state;

static int dk_record(void *user, char *seq, size_t len)
{
    dk_state *d = (dk_state *)user;
    uint32_t run = 0;
    size_t i;
    if (len + 1 < SK_K) { d->short_records++; return 0; }
    if ((d->n + len + 64) / 16 + 1 > d->cap_words) {
        uint64_t want = d->cap_words ? d->cap_words * 2 : 1u << 16;
        while (want < (d->n + len + 64) / 16 + 1) want *= 2;
        {   /* (one bit per base: 16 bases per text word); a failed realloc leaves the old block in place for the caller to free */
            uint32_t *t2 = (uint32_t *)realloc(d->text2, want * sizeof(uint32_t)), *okb;
            if (!t2) return SK_E_NOMEM;
            d->text2 = t2;
            okb = (uint32_t *)realloc(d->ok, (want / 2 + 2) * sizeof(uint32_t));
            if (!okb) return SK_E_NOMEM;
            d->ok = okb;
        }
        memset(d->text2 + d->cap_words, 0, (want - d->cap_words) * sizeof(uint32_t));
        memset(d->ok + d->ok_words, 0, (want / 2 + 2 - d->ok_words) * sizeof(uint32_t));
        d->cap_words = want;
        d->ok_words = want / 2 + 2;
    }
    for (i = 0; i < len; i++) {
        const uint32_t c = (uint8_t)seq[i];
        const uint64_t at = d->n + i;
        if (sk_is_acgt(c)) {
            d->text2[at >> 4] |= (sk_code(c) & 3u) << (2u * (15u - (uint32_t)(at & 15u)));
            if (++run >= SK_K) { const uint64_t p = at - (SK_K - 1); d->ok[p >> 5] |= 1u << (p & 31u); d->nstarts++; }
        } else {
            run = 0;
            if (!sk_is_hard_break(c)) d->soft = 1;          /* a letter the packed table cannot hold: the host's builder decides */
        }
    }
    d->n += len;
    return 0;
}

int skh_keyset_build_on_device(skh_keyset *ks, sk_ctx *ctx, const char *path, uint32_t ncols, uint32_t col0_value)
{
    dk_state d;
    uint32_t nrows = 0;
    int rc;
    if (!ks || !ctx || !path) return SK_E_ARG;
    memset(ks, 0, sizeof *ks);
    memset(&d, 0, sizeof d);
    rc = parse_file(path, dk_record, &d, NULL, NULL, 0);
    if (rc == SK_OK && (d.soft || d.nstarts == 0 || d.n >= 0x7FFFFF00u)) rc = SK_E_UNSUPPORTED;
    if (rc == SK_OK) rc = sk_table_build_from_text(ctx, d.text2, d.ok, (uint32_t)d.n, (uint32_t)d.nstarts, ncols, col0_value, &nrows);
    free(d.text2); free(d.ok);
    if (rc != SK_OK) return rc;
    ks->nrows = nrows;
    ks->short_records = d.short_records;
    /* the keys stay on the device; the host's list is filled in for the rows somebody asks about (skh_keyset_fetch_keys: the
     * informative rows, whose k-mers the hit lines print).  0 = not fetched: no canonical key is 0 (the reverse complement of
     * thirty-one A's is thirty-one T's, the larger of the two).  calloc: pages nobody touches cost nothing */
    ks->packed = (uint64_t *)calloc((size_t)(nrows ? nrows : 1), sizeof(uint64_t));
    return ks->packed ? SK_OK : SK_E_NOMEM;
}

int skh_keyset_fetch_keys(skh_keyset *ks, sk_ctx *ctx, const uint32_t *rows, uint32_t n)
{
    uint64_t *k;
    uint32_t i;
    int rc;
    if (!ks || !ctx || (n && !rows)) return SK_E_ARG;
    if (!n) return SK_OK;
    k = (uint64_t *)malloc((size_t)n * sizeof *k);
    if (!k) return SK_E_NOMEM;
    rc = sk_table_export_keys_of(ctx, rows, n, k);
    for (i = 0; i < n && rc == SK_OK; i++) ks->packed[rows[i]] = k[i];
    free(k);
    return rc;
}

void skh_keyset_free(skh_keyset *ks)
{
    if (!ks) return;
    free(ks->packed); free(ks->first_count); free(ks->locality); free(ks->wide_keys); free(ks->wide_rows);
    free(ks->text2); free(ks->first_pos);
    memset(ks, 0, sizeof *ks);
}

void skh_keyset_key(const skh_keyset *ks, uint32_t row, char out[32])
{
    uint32_t i;
    if (ks->packed[row] != SK_KEY_NONE) { decode_key(ks->packed[row], out); return; }
    for (i = 0; i < ks->nwide; i++)
        if (ks->wide_rows[i] == row) { memcpy(out, ks->wide_keys + (size_t)i * 32, 32); return; }
    out[0] = '\0';
}

int skh_keyset_load(sk_ctx *ctx, const skh_keyset *ks, uint32_t ncols)
{
    int rc = sk_table_load_ex(ctx, ks->packed, ks->nrows, ncols, ks->locality);
    if (rc) return rc;
    rc = sk_table_load_wide(ctx, ks->wide_keys, ks->wide_rows, ks->nwide);
    if (rc) return rc;
    if (ks->text2 && ks->text_bases && ks->nrows && !getenv("SK_NO_TEXT")) rc = sk_table_load_text(ctx, ks->text2, ks->text_bases, ks->first_pos);
    if (rc) return rc;
    if (ks->nrows) rc = sk_counts_set(ctx, 0, ks->first_count);
    return rc;
}

/* =========================================================================================
 * scan phase
 * ======================================================================================= */

typedef struct { sk_ctx *ctx; uint32_t col; } scan_sink;

static int scan_sink_fn(void *user, const uint8_t *chunk, uint64_t nbytes)
{
    scan_sink *s = (scan_sink *)user;
    return sk_scan_stream(s->ctx, chunk, nbytes, s->col);
}

int skh_scan_file(sk_ctx *ctx, const char *path, uint32_t col, uint64_t *bases)
{
    scan_sink s;
    int64_t rc;
    s.ctx = ctx; s.col = col;
    rc = skh_decode_file(path, 32u << 20, scan_sink_fn, &s, bases);
    return rc < 0 ? (int)rc : SK_OK;
}

/* ---- list walk: files are decoded by a small pool of host threads (gz inflate + record parsing is
 * the slow part of the whole program), batches are submitted to the one device context under a lock.
 * SK_THREADS sets the pool size (default: min(16, usable CPUs, sk_cpus.h); 1 = the reference's strict sequence).
 *
 * Work items.  A list line is one item, or -- a big plain-text file -- several: pieces [a, b) of its bytes, each
 * starting at a record boundary (scan_item).  Items are dealt to the ranks by size (longest processing time first: the
 * biggest item to the least loaded rank), the same plan on every rank from the same list and file sizes; inside a
 * rank the decode threads take them from a queue.  The counters are sums (src/genome_compare.c:220-223), so neither
 * the dealing nor the cutting changes a count -- provided every piece really starts at a record boundary, which is
 * CHECKED, not assumed (parse_range). */
typedef struct { char *path; uint64_t a, b, size; uint32_t line; int ranged; } scan_item;

/* one line of the list as the reference reports it: "<line>\t<time>" in the progress file just before the file is opened
 * (src/genome_compare.c:133-136,167-170), "skipping ..." on stderr for the -C line that equals -r (:138-141) */
typedef struct { char *text; int skipped; long end; } list_line;

typedef struct {
    sk_ctx         *ctx;
    uint32_t        col;
    int             pipe;              /* fewer files than cores: each file's inflate gets a helper thread (> 1: that many) */
    int             gpu_inflate;       /* SK_GPU_INFLATE=1: .gz items go through the device-side decoder first (experimental) */
    int             dev_workers, worker_ids;   /* ... on this many EXTRA threads (SK_GPU_INFLATE_WORKERS, default 4), which mostly wait for the device */
    pthread_mutex_t submit_mu;         /* sk_scan_stream is one-caller-at-a-time per context */
    pthread_mutex_t queue_mu;
    scan_item      *item;              /* work list of this call (what this rank scans), in list order */
    uint32_t        nitem, next;
    int             rc;                /* first failure ...                                         */
    uint32_t        rc_index;          /* ... and the item it belongs to                            */
    uint64_t        bases;
    FILE           *progress;          /* rank 0's progress file (or NULL) ...                      */
    list_line      *ll;                /* ... the list's lines ...                                  */
    uint32_t        nll, ll_next;      /* ... and the first one not yet written there               */
    int             timing;            /* SK_TIMING: where the decode threads' time goes (seconds summed over threads, under queue_mu) */
    double          t_item, t_submit_wait, t_submit, t_ticket, t_cpu, t_pack;
    uint64_t        nchunks, npacked;
    int             pack;              /* SK_LIST_PACK: 1 (default) = the chunks of plain-text items go up packed (6 bytes per 16 bases) once the scan has shown
                                        * itself bound by the link; 0 = never; 2 = always, .gz items' too (tests) */
    int             link_bound;        /* a decode thread has spent a tenth of its time waiting for uploads (atomic) */
} scan_pool;

/* The progress file gets a list line when a decode thread TAKES the line's (first) item, and every line before it that is
 * not there yet -- the reference writes the line, then opens the file.  Called under queue_mu (or from the one thread). */
static void progress_upto(scan_pool *p, uint32_t line)
{
    if (!p->progress) { if (line + 1 > p->ll_next) p->ll_next = line + 1; return; }
    for (; p->ll_next <= line && p->ll_next < p->nll; p->ll_next++) {
        time_t now = time(NULL);
        fprintf(p->progress, "%s\t%s", p->ll[p->ll_next].text, asctime(localtime(&now)));
        p->ll[p->ll_next].end = ftell(p->progress);
    }
}

/* one decode worker: two pinned chunk buffers filled in turn; a buffer is rewritten only after the
 * DMA that read it has finished (ticket) */
typedef struct {
    scan_pool *pool;
    uint8_t   *pinned[2];
    uint64_t   ticket[2];
    int        used[2], cur;
    sk_inflater *inf;                  /* SK_GPU_INFLATE=1: this worker's device-side gzip decoder ... */
    int          dev_ok;               /* ... which only the pool's first dev_workers threads use (the others inflate on the host) */
    uint8_t   *text; uint64_t text_cap;/* ... and the page-locked buffer its text lands in */
    double     t_submit_wait, t_submit, t_ticket, t_pack;      /* SK_TIMING */
    uint64_t   nchunks, npacked;
    uint8_t   *pk[2];                  /* the chunks' packed form (sk_pack_stream), two page-locked buffers taking turns ... */
    uint64_t   pk_ticket[2];           /* ... each rewritten only after the upload that read it */
    int        pk_used[2], pk_cur;
    double     t_begin, t_wait_all;    /* since this worker began: time spent waiting for its buffers' uploads */
} scan_worker;

/* size of a worker's chunk buffer: SK_CHUNK_BYTES (4096 .. 63 MiB; tests use small ones: many flushes per file), default 32 MiB */
static uint64_t pool_chunk_bytes(void)
{
    const char *e = getenv("SK_CHUNK_BYTES");
    const long long v = e ? atoll(e) : 0;
    return v >= 4096 && v <= (63ll << 20) ? (uint64_t)v : 32ull << 20;
}
#define POOL_CHUNK pool_chunk_bytes()

static int worker_sink(void *user, const uint8_t *chunk, uint64_t nbytes)
{
    scan_worker *w = (scan_worker *)user;
    int rc;
    double t0 = w->pool->timing ? now_s() : 0.0;
    double t1;
    if (w->pool->pack > 1 || (w->pool->pack && tl_plain_text && __atomic_load_n(&w->pool->link_bound, __ATOMIC_RELAXED))) {
        /* The chunk goes up PACKED: what the scan kernel's first phase would make of its bytes is made here (sk_pack.h), 6 bytes per
         * 16 bases over the link instead of 16 -- the list scan of plain text was bound by the link, not by these threads.  A chunk
         * with a byte for the byte-string kernel (IUPAC, U, CR ...) goes up as bytes, below. */
        const int i = w->pk_cur ^= 1;
        if (!w->pk[i] && sk_pinned_alloc(w->pool->ctx, (void **)&w->pk[i], sk_packed_bytes(POOL_CHUNK)) != SK_OK) w->pk[i] = NULL;
        if (w->pk[i]) {
            int odd = 0;
            if (w->pk_used[i]) { sk_ticket_wait(w->pool->ctx, w->pk_ticket[i]); w->pk_used[i] = 0; }
            if (w->pool->timing) { t1 = now_s(); w->t_ticket += t1 - t0; t0 = t1; }
            sk_pack_stream(chunk, nbytes, w->pk[i], &odd);
            if (w->pool->timing) { t1 = now_s(); w->t_pack += t1 - t0; t0 = t1; }
            if (!odd) {
                pthread_mutex_lock(&w->pool->submit_mu);
                t1 = w->pool->timing ? now_s() : 0.0;
                rc = sk_scan_pinned_packed(w->pool->ctx, w->pk[i], nbytes, w->pool->col, &w->pk_ticket[i]);
                pthread_mutex_unlock(&w->pool->submit_mu);
                if (w->pool->timing) { w->t_submit_wait += t1 - t0; w->t_submit += now_s() - t1; w->nchunks++; w->npacked++; }
                w->pk_used[i] = rc == SK_OK;
                w->used[w->cur] = 0;                    /* (the bytes were read by this thread alone: their buffer is free at once) */
                return rc;
            }
        }
    }
    pthread_mutex_lock(&w->pool->submit_mu);
    t1 = w->pool->timing ? now_s() : 0.0;
    rc = sk_scan_pinned(w->pool->ctx, chunk, nbytes, w->pool->col, &w->ticket[w->cur]);
    pthread_mutex_unlock(&w->pool->submit_mu);
    if (w->pool->timing) { w->t_submit_wait += t1 - t0; w->t_submit += now_s() - t1; w->nchunks++; }
    w->used[w->cur] = 1;
    return rc;
}

/* the buffers come from the context's store of page-locked memory (kept from call to call) when they are first
 * needed: a worker that finds the queue empty takes none, an item that fits one buffer never takes the second */
static uint8_t *worker_buf(scan_worker *w, int i)
{
    if (!w->pinned[i] && sk_pinned_alloc(w->pool->ctx, (void **)&w->pinned[i], POOL_CHUNK) != SK_OK) w->pinned[i] = NULL;
    return w->pinned[i];
}

static uint8_t *worker_next_buf(void *user)
{
    scan_worker *w = (scan_worker *)user;
    w->cur ^= 1;
    if (w->used[w->cur]) {
        const double t0 = now_s();
        double dt;
        sk_ticket_wait(w->pool->ctx, w->ticket[w->cur]);
        dt = now_s() - t0;
        if (w->pool->timing) w->t_ticket += dt;
        /* Is this list scan bound by the link?  A thread that has spent a tenth of its time waiting for its own buffers' uploads says
         * so for all: from then on the chunks of plain-text items go up packed (worker_sink).  A scan that is short of CPUs instead
         * -- few threads per card, .gz items -- never waits here and keeps its cycles for the decode. */
        w->t_wait_all += dt;
        if (!__atomic_load_n(&w->pool->link_bound, __ATOMIC_RELAXED) && w->t_wait_all > 0.002 && w->t_wait_all > 0.1 * (now_s() - w->t_begin))
            __atomic_store_n(&w->pool->link_bound, 1, __ATOMIC_RELAXED);
    }
    return worker_buf(w, w->cur);
}

static int worker_init(scan_worker *w, scan_pool *p)
{
    memset(w, 0, sizeof *w);
    w->pool = p;
    w->t_begin = now_s();
    return SK_OK;
}

static void worker_done(scan_worker *w)
{
#ifdef SK_EXPERIMENTS
    if (w->inf) sk_inflater_destroy(w->inf);
#endif
    if (w->text) sk_pinned_free(w->pool->ctx, w->text);
    pthread_mutex_lock(&w->pool->submit_mu);
    if (w->pinned[0]) sk_pinned_free(w->pool->ctx, w->pinned[0]);      /* (synchronises the stream first) */
    if (w->pinned[1]) sk_pinned_free(w->pool->ctx, w->pinned[1]);
    if (w->pk[0]) sk_pinned_free(w->pool->ctx, w->pk[0]);
    if (w->pk[1]) sk_pinned_free(w->pool->ctx, w->pk[1]);
    pthread_mutex_unlock(&w->pool->submit_mu);
}

/* One piece [a, b) of a plain-text file: the records that START in it.  The piece's own start s(a) and end s(b) come from
 * guess_record_start; its bytes are parsed as a file of their own, and afterwards the parser must stand between two records
 * (P_SEEK after a FASTQ record, P_LINE_START after a FASTA line) with a header character next -- then s(b) is a true record
 * boundary GIVEN that s(a) was one.  Piece 0 starts at 0, which is one; every later piece's start is the previous piece's
 * checked end.  So if every piece passes, the pieces' records are exactly the file's records; if any piece fails, the
 * caller fails the run (SK_E_SPLIT) instead of counting something else. */
static int parse_range(const scan_item *it, rec_fn fn, void *user, int64_t *nrecords)
{
    enum { BLK = 4 << 20 };
    int fd = open(it->path, O_RDONLY);
    struct stat st;
    const unsigned char *t;
    uint64_t sa, sb, i;
    parser ps;
    int ok = 1;
    if (fd < 0) return SK_E_OPEN;
    if (fstat(fd, &st) != 0 || (uint64_t)st.st_size != it->size) { close(fd); return SK_E_SPLIT; }     /* (changed since the plan) */
    t = (const unsigned char *)mmap(NULL, (size_t)it->size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (t == MAP_FAILED) { close(fd); return SK_E_OPEN; }
    /* the mapping serves the two guesses and the look at the byte behind the piece (a few pages); the piece's text is read()
     * into a block of this thread's -- sixteen threads faulting mapped pages in wait for each other (parse_file above) */
    sa = parser_guess_start(t, it->size, it->a, 1);
    sb = it->b >= it->size ? it->size : parser_guess_start(t, it->size, it->b, 1);
    parser_init(&ps, fn, user);
    tl_plain_text = 1;
    {
        unsigned char *rb = (unsigned char *)malloc(BLK);
        for (i = sa; i < sb && ps.state != P_STOP; ) {
            const size_t want = (size_t)(sb - i < BLK ? sb - i : BLK);
            size_t have = 0;
#ifdef MADV_POPULATE_READ
            if (!list_input_read()) {                        /* the block's pages populated, parsed in place, dropped (parse_file above) */
                const uintptr_t pa = (uintptr_t)(t + i) & ~(uintptr_t)4095u, pe = ((uintptr_t)(t + i) + want + 4095u) & ~(uintptr_t)4095u;
                if (madvise((void *)pa, (size_t)(pe - pa), MADV_POPULATE_READ) == 0) {
                    const uintptr_t da = ((uintptr_t)(t + i) + 4095u) & ~(uintptr_t)4095u, de = ((uintptr_t)(t + i) + want) & ~(uintptr_t)4095u;
                    parser_feed(&ps, t + i, want);
                    if (de > da) (void)madvise((void *)da, (size_t)(de - da), MADV_DONTNEED);
                    i += want;
                    continue;
                }
                if (errno == EINVAL) __atomic_store_n(&list_input_mode, 1, __ATOMIC_RELAXED);
            }
#endif
            while (rb && have < want) {
                const ssize_t r = pread(fd, rb + have, want - have, (off_t)(i + have));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) break;
                have += (size_t)r;
            }
            if (rb && have == want) parser_feed(&ps, rb, want); else parser_feed(&ps, t + i, want);       /* (a failed read: out of the mapping) */
            i += want;
        }
        free(rb);
    }
    close(fd);
    if (ps.state != P_STOP && sb < it->size && sb > sa)
        ok = parser_between_records(&ps) && (t[sb] == 0x3e || t[sb] == 0x40);
    /* a FASTQ record whose quality does not match its sequence ends the FILE for the reference (src/kseq.h:205-209 returns -2,
     * src/genome_compare.c:203 stops reading): the pieces behind this one must not count, and some may have been counted
     * already -- so the run fails rather than report more than the reference would */
    if (ps.state == P_STOP && !ps.sink_rc && ps.end_kind == SKP_END_TRUNC && sb < it->size) ok = 0;
    if (ps.state != P_STOP) parser_eof(&ps);
    if (nrecords) *nrecords = ps.nrecords;
    {
        const int sink_rc = ps.sink_rc;
        parser_free(&ps);
        munmap((void *)t, (size_t)it->size);
        if (sink_rc) return sink_rc;
    }
    return ok ? SK_OK : SK_E_SPLIT;
}

/* ---- one .gz item inflated by several threads (pipe > 1): its text comes faster than one thread parses it (16 inflating
 * threads: 9.5 GB/s; the parser: 5-7 GB/s).  The item's thread only CUTS the text into segments of about a chunk's worth, each
 * ending at a line start that looks like a record start (parser_guess_start); a few helper threads parse the segments as files
 * of their own into their own page-locked buffers and CHECK that they end between two records (parser_between_records) -- the
 * first segment starts at the file's start, every other at the checked end of the one before, so if all checks hold the
 * segments' records are the file's records (src/kseq.h:171-211 reads from the top; nothing here changes what a record is).
 * Counters are sums: the order in which segments reach the device does not matter.  A failed check, or a record that ends the
 * file for the reference (FASTQ quality of the wrong length, :205-209) in a segment that is not the last, fails the scan
 * (SK_E_SPLIT) -- later segments may have been counted already; SK_NO_SPLIT=1 parses on one thread.
 * MEASURED (tools/one_gz_bench.py, profiles/r02_one_gz.txt; one 3 Gbase .gz, 16 inflating threads): it LOSES -- FASTA.gz
 * 7.4 Gbase/s with the one parser, 5.1 with four helpers; FASTQ.gz 5.0 against 2.6-4.8.  Gathering the text into segments is
 * one more copy of every byte on the cutting thread, which is about what parsing it costs, and FASTQ is bound by the inflate
 * (9.5 GB/s of text = 4.6 Gbase/s) either way.  So it runs only when SK_PARSE_THREADS asks for it (tests do). */
typedef struct gz_seg { unsigned char *buf; size_t n, cap; int is_last; struct gz_seg *next; } gz_seg;
typedef struct {
    scan_pool *pool;
    pthread_mutex_t mu; pthread_cond_t cv;
    gz_seg *head, *tail;
    int nq, done, stop, rc;
    uint64_t bases; int64_t nrec;
} gz_split;

static void *gz_split_worker(void *arg)
{
    gz_split *g = (gz_split *)arg;
    scan_worker w;
    worker_init(&w, g->pool);
    for (;;) {
        gz_seg *sg;
        int stop;
        pthread_mutex_lock(&g->mu);
        while (!g->head && !g->done) pthread_cond_wait(&g->cv, &g->mu);
        sg = g->head;
        if (!sg) { pthread_mutex_unlock(&g->mu); break; }
        g->head = sg->next;
        if (!g->head) g->tail = NULL;
        g->nq--;
        stop = g->stop;
        pthread_cond_broadcast(&g->cv);
        pthread_mutex_unlock(&g->mu);
        if (!stop) {
            stream_writer sw;
            parser ps;
            int ok = 1;
            memset(&sw, 0, sizeof sw);
            sw.cap = POOL_CHUNK; sw.sink = worker_sink; sw.user = &w; sw.next_buf = worker_next_buf;
            parser_init(&ps, writer_record, &sw);
            parser_feed(&ps, sg->buf, sg->n);
            if (!sg->is_last) {
                if (ps.state == P_STOP) ok = ps.sink_rc != 0 || ps.end_kind != SKP_END_TRUNC;   /* (a sink error is reported as itself) */
                else ok = parser_between_records(&ps);
            }
            if (ps.state != P_STOP) parser_eof(&ps);
            writer_flush(&sw);
            pthread_mutex_lock(&g->mu);
            g->bases += sw.bases;
            g->nrec += ps.nrecords;
            if (sw.rc && !g->rc) g->rc = sw.rc;
            if (!ok && !g->rc) g->rc = SK_E_SPLIT;
            if (!ok || sw.rc) g->stop = 1;
            pthread_mutex_unlock(&g->mu);
            parser_free(&ps);
        }
        free(sg->buf);
        free(sg);
    }
    worker_done(&w);
    return NULL;
}

static void gz_split_push(gz_split *g, gz_seg *sg, int limit)
{
    pthread_mutex_lock(&g->mu);
    while (g->nq >= limit && !g->stop) pthread_cond_wait(&g->cv, &g->mu);
    sg->next = NULL;
    if (g->tail) g->tail->next = sg; else g->head = sg;
    g->tail = sg;
    g->nq++;
    pthread_cond_broadcast(&g->cv);
    pthread_mutex_unlock(&g->mu);
}

/* records (or a negative SK_E_*); -100: not a gzip file / no helper could be started -- the caller parses it the ordinary way */
static int64_t parse_gz_split(scan_worker *w, const scan_item *it, uint64_t *bases)
{
    enum { BLK = 1 << 20 };
    skzp zp;
    gz_split g;
    pthread_t th[8];
    const char *e = getenv("SK_PARSE_THREADS");
    int npar = e ? atoi(e) : 0, nth = 0, i;          /* off unless asked for: measured, it loses -- see the comment above */
    const size_t want = POOL_CHUNK / 2 < (1u << 20) ? POOL_CHUNK : POOL_CHUNK / 2;
    const size_t TAIL = want / 2 < (64u << 10) ? (want / 2 < 256 ? 256 : want / 2) : (64u << 10);
    size_t scan_from = want;
    gz_seg *sg;
    if (npar < 2) return -100;
    if (npar > 8) npar = 8;
    if (skzp_open_threads(&zp, it->path, w->pool->pipe) != SKZ_OK) return -100;
    memset(&g, 0, sizeof g);
    g.pool = w->pool;
    pthread_mutex_init(&g.mu, NULL);
    pthread_cond_init(&g.cv, NULL);
    for (i = 0; i < npar; i++) if (pthread_create(&th[nth], NULL, gz_split_worker, &g) == 0) nth++;
    if (nth == 0) { skzp_close(&zp); pthread_mutex_destroy(&g.mu); pthread_cond_destroy(&g.cv); return -100; }
    sg = (gz_seg *)calloc(1, sizeof *sg);
    sg->cap = want + (want >> 2) + BLK;
    sg->buf = (unsigned char *)malloc(sg->cap);
    for (;;) {
        const unsigned char *data = NULL;
        size_t n;
        int stop;
        pthread_mutex_lock(&g.mu);
        stop = g.stop;
        pthread_mutex_unlock(&g.mu);
        if (stop) break;
        n = skzp_next(&zp, &data);
        if (n == 0) break;
        if (sg->n + n > sg->cap) { sg->cap = (sg->n + n) * 2; sg->buf = (unsigned char *)realloc(sg->buf, sg->cap); }
        memcpy(sg->buf + sg->n, data, n);
        sg->n += n;
        while (sg->n >= want + TAIL) {                      /* (a delivery may hold several segments' worth) */
            const uint64_t cut = parser_guess_start(sg->buf, sg->n, scan_from, 0);
            if (cut >= sg->n) { scan_from = sg->n - TAIL; break; }     /* no likely record start yet (very long records): keep gathering */
            else {
                gz_seg *nx = (gz_seg *)calloc(1, sizeof *nx);
                nx->cap = want + (want >> 2) + BLK;
                if (nx->cap < sg->n - cut) nx->cap = (sg->n - cut) * 2;
                nx->buf = (unsigned char *)malloc(nx->cap);
                nx->n = sg->n - (size_t)cut;
                memcpy(nx->buf, sg->buf + cut, nx->n);
                sg->n = (size_t)cut;
                gz_split_push(&g, sg, nth + 2);
                sg = nx;
                scan_from = want;
            }
        }
    }
    sg->is_last = 1;
    gz_split_push(&g, sg, nth + 2);
    pthread_mutex_lock(&g.mu);
    g.done = 1;
    pthread_cond_broadcast(&g.cv);
    pthread_mutex_unlock(&g.mu);
    for (i = 0; i < nth; i++) pthread_join(th[i], NULL);
    skzp_close(&zp);
    pthread_mutex_destroy(&g.mu);
    pthread_cond_destroy(&g.cv);
    *bases += g.bases;
    return g.rc ? (int64_t)g.rc : g.nrec;
}

#ifdef SK_EXPERIMENTS        /* (make EXPERIMENTS=1; the library that ships has no device-side inflate: DESIGN.md section 7) */
/* SK_GPU_INFLATE=1 (experimental): a .gz item of one member and dynamic blocks only is inflated ON THE DEVICE (sk_inflate.hip: the speculative
 * scheme of sk_gzpar.h, a lane per 16 KiB segment; every guess checked, the chain checked here in the library, the CRC-32 here) and its text
 * parsed by this thread.  -100: not such a file, or a check failed -- the caller decodes it on the host as before, so the bytes are zlib's
 * either way. */
static int64_t parse_gz_on_device(scan_worker *w, const scan_item *it, uint64_t *bases)
{
    int fd;
    struct stat st;
    const unsigned char *m;
    uint64_t want, len = 0;
    uint32_t crc = 0;
    int rc;
    if (!w->pool->gpu_inflate || !w->dev_ok) return -100;
    fd = open(it->path, O_RDONLY);
    if (fd < 0) return -100;                              /* (the ordinary path reports it) */
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 32) { close(fd); return -100; }
    m = (const unsigned char *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return -100;
    want = sk_inflate_gz_size(m, (uint64_t)st.st_size);
    if (!want) {
        if (w->pool->gpu_inflate > 1 && m[0] == 0x1f && m[1] == 0x8b) fprintf(stderr, "%s: left to the host decoder (not one member of a size the device path takes)\n", it->path);
        munmap((void *)m, (size_t)st.st_size);
        return -100;
    }
    if (!w->inf && sk_inflater_create(sk_ctx_device_(w->pool->ctx), &w->inf) != SK_OK) { w->inf = NULL; munmap((void *)m, (size_t)st.st_size); return -100; }
    if (want > w->text_cap) {
        const uint64_t cap = (want + (64u << 20)) & ~(uint64_t)((64u << 20) - 1);      /* (few distinct sizes: the context keeps what is given back) */
        if (w->text) sk_pinned_free(w->pool->ctx, w->text);
        w->text = NULL; w->text_cap = 0;
        if (sk_pinned_alloc(w->pool->ctx, (void **)&w->text, cap) != SK_OK) { w->text = NULL; munmap((void *)m, (size_t)st.st_size); return -100; }
        w->text_cap = cap;
    }
    {
        const double t0 = now_s();
        double t1;
        rc = sk_inflate_gz(w->inf, m, (uint64_t)st.st_size, w->text, w->text_cap, &len, &crc);
        munmap((void *)m, (size_t)st.st_size);
        t1 = now_s();
        if (rc == SK_OK) { pthread_once(&skz_crc_once, skz_crc_init); if (skz_crc32(0, w->text, (size_t)len) != crc) rc = SK_E_UNSUPPORTED; }
        if (w->pool->gpu_inflate > 1)                      /* (SK_GPU_INFLATE=2: say which way every file went) */
            fprintf(stderr, rc == SK_OK ? "%s: inflated on the device (%llu bytes of text; %.0f ms, CRC-32 %.0f ms)\n"
                                        : "%s: left to the host decoder (a check of the device path did not hold)\n",
                    it->path, (unsigned long long)len, (t1 - t0) * 1e3, (now_s() - t1) * 1e3);
    }
    if (rc != SK_OK) return -100;                         /* (unsupported, or a device hiccup: the host path decides what the file is) */
    {
        stream_writer sw;
        parser ps;
        int64_t nrec;
        uint64_t at;
        memset(&sw, 0, sizeof sw);
        sw.cap = POOL_CHUNK; sw.sink = worker_sink; sw.user = w; sw.next_buf = worker_next_buf;
        parser_init(&ps, writer_record, &sw);
        for (at = 0; at < len && ps.state != P_STOP; at += 4u << 20) parser_feed(&ps, w->text + at, (size_t)(len - at < (4u << 20) ? len - at : (4u << 20)));
        if (ps.state != P_STOP) parser_eof(&ps);
        writer_flush(&sw);
        *bases += sw.bases;
        nrec = ps.nrecords;
        parser_free(&ps);
        return sw.rc ? (int64_t)sw.rc : nrec;
    }
}

#endif

/* decode one item into the worker's pinned buffers; returns records or a negative SK_E_* */
static int64_t worker_item(scan_worker *w, const scan_item *it, uint64_t *bases)
{
    stream_writer sw;
    int64_t nrec = 0;
    int rc;
#ifdef SK_EXPERIMENTS
    if (!it->ranged) {
        const int64_t r = parse_gz_on_device(w, it, bases);
        if (r != -100) return r;
    }
#endif
    if (!it->ranged && w->pool->pipe > 1 && !getenv("SK_NO_SPLIT") && !getenv("SK_ZLIB")) {
        const int64_t r = parse_gz_split(w, it, bases);
        if (r != -100) return r;
    }
    memset(&sw, 0, sizeof sw);
    sw.buf = NULL;                                   /* (taken through next_buf by the first record: the other buffer may still be read) */
    sw.cap = POOL_CHUNK;
    sw.sink = worker_sink;
    sw.user = w;
    sw.next_buf = worker_next_buf;
    rc = it->ranged ? parse_range(it, writer_record, &sw, &nrec) : parse_file(it->path, writer_record, &sw, &nrec, NULL, w->pool->pipe);
    if (rc == SK_OK || (rc == SK_E_SPLIT && !sw.rc)) writer_flush(&sw);
    *bases += sw.bases;
    if (rc != SK_OK) return rc;
    if (sw.rc) return sw.rc;
    return nrec;
}

static void *pool_worker(void *arg)
{
    scan_pool *p = (scan_pool *)arg;
    scan_worker w;
    int wrc = worker_init(&w, p);
    pthread_setname_np(pthread_self(), "sk-decode");
    pthread_mutex_lock(&p->queue_mu);
    w.dev_ok = p->worker_ids++ < p->dev_workers;
    pthread_mutex_unlock(&p->queue_mu);
    for (;;) {
        uint32_t i;
        uint64_t bases = 0;
        int64_t rc;
        pthread_mutex_lock(&p->queue_mu);
        i = p->next;
        if (i >= p->nitem || p->rc != SK_OK) { pthread_mutex_unlock(&p->queue_mu); break; }
        p->next++;
        progress_upto(p, p->item[i].line);
        pthread_mutex_unlock(&p->queue_mu);
        {
            const double t0 = p->timing ? now_s() : 0.0;
            rc = wrc != SK_OK ? wrc : worker_item(&w, &p->item[i], &bases);
            pthread_mutex_lock(&p->queue_mu);
            if (p->timing) p->t_item += now_s() - t0;
        }
        p->bases += bases;
        if (rc < 0 && (p->rc == SK_OK || i < p->rc_index)) { p->rc = (int)rc; p->rc_index = i; }
        pthread_mutex_unlock(&p->queue_mu);
    }
    if (p->timing) {
        struct timespec ts;
        pthread_mutex_lock(&p->queue_mu);
        p->t_submit_wait += w.t_submit_wait; p->t_submit += w.t_submit; p->t_ticket += w.t_ticket; p->nchunks += w.nchunks;
        p->t_pack += w.t_pack; p->npacked += w.npacked;
        if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0) p->t_cpu += (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
        pthread_mutex_unlock(&p->queue_mu);
    }
    worker_done(&w);
    return NULL;
}

/* the plan: items of the whole list (every rank computes the same), then who takes which */
typedef struct { uint64_t est; uint32_t idx; } plan_key;
static int plan_cmp(const void *a, const void *b)
{
    const plan_key *x = (const plan_key *)a, *y = (const plan_key *)b;
    if (x->est != y->est) return x->est > y->est ? -1 : 1;
    return x->idx < y->idx ? -1 : x->idx > y->idx;
}

/* items[0..n): all items in list order; owner[i] = the rank that scans item i */
static void plan_owners(const scan_item *items, const uint64_t *est, uint32_t n, uint32_t world, uint32_t *owner)
{
    plan_key *k = (plan_key *)malloc(((size_t)n + 1) * sizeof *k);
    uint64_t *load = (uint64_t *)calloc(world, sizeof *load);
    uint32_t i, r;
    (void)items;
    for (i = 0; i < n; i++) { k[i].est = est[i]; k[i].idx = i; }
    qsort(k, n, sizeof *k, plan_cmp);
    for (i = 0; i < n; i++) {
        uint32_t best = 0;
        for (r = 1; r < world; r++) if (load[r] < load[best]) best = r;
        owner[k[i].idx] = best;
        load[best] += k[i].est ? k[i].est : 1;
    }
    free(k); free(load);
}

/* FNV-1a over everything the ranks must agree on: the items (path, byte range, size as seen by stat) and their owners */
static uint64_t fnv_bytes(uint64_t h, const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    while (n--) { h ^= *b++; h *= 0x100000001B3ull; }
    return h;
}
static uint64_t plan_digest(const scan_item *items, const uint32_t *owner, uint32_t n, uint32_t world)
{
    uint64_t h = 0xCBF29CE484222325ull;
    uint32_t i;
    h = fnv_bytes(h, &n, sizeof n);
    h = fnv_bytes(h, &world, sizeof world);
    for (i = 0; i < n; i++) {
        const uint32_t rg = (uint32_t)items[i].ranged;
        h = fnv_bytes(h, items[i].path, strlen(items[i].path) + 1);
        h = fnv_bytes(h, &items[i].a, sizeof items[i].a);
        h = fnv_bytes(h, &items[i].b, sizeof items[i].b);
        h = fnv_bytes(h, &items[i].size, sizeof items[i].size);
        h = fnv_bytes(h, &items[i].line, sizeof items[i].line);
        h = fnv_bytes(h, &rg, sizeof rg);
        h = fnv_bytes(h, &owner[i], sizeof owner[i]);
    }
    return h;
}

#define SK_PLAN_LANES 16u     /* decode lanes per rank the multi-rank plan is laid out for (the programs' thread cap) */

typedef struct { uint64_t *hash; uint32_t *owner; uint32_t cap, *nlines; } plan_report;

/* What a cut that does not hold costs: nothing but time.  Before a list with pieces is scanned the column is copied; if a
 * piece fails its check (SK_E_SPLIT: a record that ends the FILE for the reference in the middle of it, a layout the guess
 * misjudges) the column and the progress file are put back and the list is scanned again UNCUT -- the reference's strict
 * record sequence per file, which reports what the reference reports (src/kseq.h:205-209, src/genome_compare.c:203).
 * The cut is an optimisation that is checked, not a new way to fail.  With several ranks in the library's communicator
 * the same holds (round 4): the plan is global, so EVERY rank copies its column when the plan has pieces, the ranks learn of
 * a failed check through the agreement that closes every list scan, and all of them put their columns back and scan their
 * share of the whole-file plan.  Only a caller that runs ranks WITHOUT the library's communicator still gets SK_E_SPLIT
 * (its other ranks cannot be told from here): it coordinates the same steps itself (strainer2_amd/dist.py:
 * scan_list_sharded) with skh_scan_list_uncut.
 *
 * Collectives: whatever happens to a rank locally -- a list it cannot open, a file only it owns that is
 * missing, a cut that fails -- every rank issues the same sequence: per list scan ONE agreement before (plan hash, "my
 * set-up for this list failed") and ONE after (first unreadable line, "a cut failed", "a device error"), each a max
 * all-reduce of a few words (sk_comm_max_u64), and every rank returns the SAME status, so the callers' decisions
 * (skip the remaining lists, go to the big all-reduce) are the same everywhere. */
typedef struct { uint32_t *snap; long prog_at; int armed; } split_guard;

#define LIST_FAIL_OPEN  1u    /* agreement before the scan: this rank could not read the list itself */
#define LIST_FAIL_GUARD 2u    /*                            ... could not copy its column (no memory, device error) */

static int scan_list_once(sk_ctx *ctx, const char *list_path, const char *skip, uint32_t col, FILE *progress,
                          FILE *err, uint32_t rank, uint32_t world, uint64_t *bases, int plan_only, const plan_report *report,
                          int no_split, split_guard *guard)
{
    FILE *fp = fopen(list_path, "r");
    char *line = NULL, *nl;
    size_t cap = 0;
    uint32_t nline = 0, nall = 0, acap = 0, lcap = 0, i;
    int nthreads = 1;
    const char *env = getenv("SK_THREADS");
    scan_pool pool;
    scan_item *all = NULL;
    uint64_t *est = NULL, total = 0, plan_hash = 0;
    uint32_t *owner = NULL;
    uint64_t local_fail = 0;
    uint32_t open_line = UINT32_MAX;            /* the first list line whose file some rank could not open */
    int any_ranged = 0, coordinated, by_other = 0;
    if (world == 0) world = 1;
    /* several ranks act together only when all of them are in the library's communicator */
    coordinated = world <= 1 || (ctx && (uint32_t)sk_comm_world(ctx) == world);
    if (!fp) {
        if (err) fprintf(err, "could not read file %s in GEN_all_kmer_counts()\n", list_path);
        if (!ctx || plan_only || world <= 1 || !coordinated) return SK_E_OPEN;
        local_fail |= LIST_FAIL_OPEN;                       /* (the other ranks are waiting in the agreement below) */
    }
    if (no_split == 2) local_fail |= LIST_FAIL_GUARD;       /* (second scan after a failed cut: this rank could not put its column back) */
    memset(&pool, 0, sizeof pool);
    pool.timing = getenv("SK_TIMING") != NULL;
    { const char *e = getenv("SK_LIST_PACK"); pool.pack = e && e[0] == '0' ? 0 : e && e[0] == '2' ? 2 : 1; }      /* (0: never, 2: always -- tests --, default: plain-text items' chunks once the scan is bound by the link) */
    pool.ctx = ctx;
    pool.col = col;
    pthread_mutex_init(&pool.submit_mu, NULL);
    pthread_mutex_init(&pool.queue_mu, NULL);
    if (env) nthreads = atoi(env);
    else { long n = sk_cpu_budget(); nthreads = n > 16 ? 16 : (int)n; }
    if (nthreads < 1) nthreads = 1;

    /* the list's lines; what the reference says about each of them (progress line, skip message) is said in list
     * order: the progress line when a decode thread takes the line's file, the skip messages behind the scans */
    while (fp && getline(&line, &cap, fp) != -1) {
        struct stat st;
        if ((nl = strchr(line, '\n')) != NULL) *nl = '\0';
        if (nline == lcap) { lcap = lcap ? lcap * 2 : 64; pool.ll = (list_line *)realloc(pool.ll, lcap * sizeof *pool.ll); }
        memset(&pool.ll[nline], 0, sizeof pool.ll[nline]);
        pool.ll[nline].text = strdup(line);
        if (skip && strcmp(skip, line) == 0) {
            pool.ll[nline++].skipped = 1;
            continue;
        }
        if (nall == acap) { acap = acap ? acap * 2 : 64; all = (scan_item *)realloc(all, acap * sizeof *all); }
        memset(&all[nall], 0, sizeof all[nall]);
        all[nall].path = strdup(line);
        all[nall].line = nline++;
        all[nall].size = stat(line, &st) == 0 && S_ISREG(st.st_mode) ? (uint64_t)st.st_size : 0;
        nall++;
    }
    free(line);
    if (fp) fclose(fp);
    pool.nll = nline;
    pool.progress = rank == 0 ? progress : NULL;

    /* cut big plain-text files into pieces (never with one thread and one rank: that is the reference's strict sequence).
     * With several ranks the plan must be the SAME on every rank, so it is a function of the list, the files' sizes and
     * the world size only: a fixed number of decode lanes per rank stands in for the local thread count (which follows
     * SK_THREADS, the cgroup quota, LOCAL_WORLD_SIZE ... and may differ from rank to rank) -- and the ranks compare a
     * hash of their plans before any of them scans (below). */
    {
        const uint64_t lanes = world > 1 ? (uint64_t)world * SK_PLAN_LANES : (uint64_t)(nthreads > 1 ? nthreads : 1);
        uint64_t target, floor_bytes = 32u << 20;
        uint32_t n0 = nall, out = 0;
        scan_item *cut;
        uint64_t *gz = (uint64_t *)calloc((size_t)nall + 1, sizeof *gz);
        for (i = 0; i < nall; i++) {
            unsigned char magic[2] = {0, 0};
            FILE *f = all[i].size ? fopen(all[i].path, "rb") : NULL;
            if (f) { if (fread(magic, 1, 2, f) != 2) magic[0] = 0; fclose(f); }
            gz[i] = magic[0] == 0x1f && magic[1] == 0x8b;
            total += gz[i] ? all[i].size * 4 : all[i].size;           /* (a .gz FASTQ inflates about fourfold) */
        }
        target = total / (4 * lanes);
        if ((env = getenv("SK_SPLIT_FLOOR_BYTES")) != NULL && atoll(env) > 0) floor_bytes = (uint64_t)atoll(env);   /* (tests) */
        if (target < floor_bytes) target = floor_bytes;
        if ((env = getenv("SK_SPLIT_BYTES")) != NULL && atoll(env) > 0) target = (uint64_t)atoll(env);     /* (tests) */
        {   /* how many items there will be */
            size_t total_items = 0;
            for (i = 0; i < n0; i++) {
                uint64_t np = 1;
                if (lanes > 1 && !gz[i] && !no_split && !getenv("SK_NO_SPLIT") && all[i].size >= 2 * target) np = (all[i].size + target - 1) / target;
                total_items += np > 256 ? 256 : (size_t)np;
            }
            cut = (scan_item *)malloc((total_items + 1) * sizeof *cut);
            est = (uint64_t *)malloc((total_items + 1) * sizeof *est);
        }
        for (i = 0; i < n0; i++) {
            uint32_t np = 1, k;
            if (lanes > 1 && !gz[i] && !no_split && !getenv("SK_NO_SPLIT") && all[i].size >= 2 * target) {
                np = (uint32_t)((all[i].size + target - 1) / target > 256 ? 256 : (all[i].size + target - 1) / target);
            }
            for (k = 0; k < np; k++) {
                cut[out] = all[i];
                if (k) cut[out].path = strdup(all[i].path);
                cut[out].ranged = np > 1;
                cut[out].a = all[i].size / np * k;
                cut[out].b = k + 1 == np ? all[i].size : all[i].size / np * (k + 1);
                est[out] = np > 1 ? cut[out].b - cut[out].a : (gz[i] ? all[i].size * 4 : all[i].size);
                out++;
            }
        }
        free(all); free(gz);
        all = cut; nall = out;
    }
    owner = (uint32_t *)malloc(((size_t)nall + 1) * sizeof *owner);
    plan_owners(all, est, nall, world, owner);
    plan_hash = plan_digest(all, owner, nall, world);
    if (report && report->hash) *report->hash = plan_hash;
    if (report && report->nlines) *report->nlines = nline;
    if (report && report->owner) {                        /* who scans which list line (SKH_PLAN_SKIPPED / SKH_PLAN_SHARED) */
        for (i = 0; i < nline && i < report->cap; i++) report->owner[i] = SKH_PLAN_SKIPPED;
        for (i = 0; i < nall; i++)
            if (all[i].line < report->cap) {
                uint32_t *o = &report->owner[all[i].line];
                *o = *o == SKH_PLAN_SKIPPED || *o == owner[i] ? owner[i] : SKH_PLAN_SHARED;
            }
    }
    for (i = 0; i < nall; i++) any_ranged |= all[i].ranged;
    if (guard && ctx && !plan_only && !no_split && coordinated && any_ranged && sk_table_rows(ctx)) {
        /* pieces ahead (somewhere in the plan: on every rank, then): keep what a failed cut would spoil */
        guard->snap = (uint32_t *)malloc((size_t)sk_table_rows(ctx) * sizeof(uint32_t));
        if (guard->snap && sk_counts_fetch(ctx, col, guard->snap) == SK_OK) {
            guard->prog_at = progress ? (fflush(progress), ftell(progress)) : -1;
            guard->armed = 1;
        } else local_fail |= LIST_FAIL_GUARD;
    }
    if (ctx && !plan_only) {                            /* (a world of one with a communicator -- SK_FORCE_COMM -- takes the same road: tests) */
        /* The agreement before the scan.  Local settings that change the plan (SK_SPLIT_BYTES, SK_NO_SPLIT) or a file whose
         * size another rank sees differently would have byte ranges scanned twice or never and the summed table silently
         * wrong: every rank leaves instead (SK_E_PLAN) -- and so does every rank when ONE of them could not read the list or
         * copy its column.  (Without an in-library communicator -- a caller that reduces the counters itself -- the caller
         * compares skh_list_plan_hash() through its own collective.) */
        uint64_t v[3];
        int arc;
        v[0] = plan_hash; v[1] = ~plan_hash; v[2] = local_fail;
        arc = sk_comm_max_u64(ctx, v, 3);
        if (arc != SK_OK) {
            if (err) fprintf(err, "kmer_scrub_count: %s (%s)\n", sk_strerror(arc), sk_last_error(ctx));
            pool.rc = arc; plan_only = 1;
        } else if (v[2] & LIST_FAIL_OPEN) {
            if (err && !(local_fail & LIST_FAIL_OPEN)) fprintf(err, "kmer_scrub_count: another rank could not read %s; nothing is reported\n", list_path);
            pool.rc = SK_E_OPEN; plan_only = 1;
        } else if (v[2] & LIST_FAIL_GUARD) {
            if (err) fprintf(err, "kmer_scrub_count: %s could not keep a copy of its counters before cutting the files of %s\n",
                             (local_fail & LIST_FAIL_GUARD) ? "this rank" : "another rank", list_path);
            pool.rc = SK_E_NOMEM; plan_only = 1;
        } else if (v[0] != plan_hash || v[1] != ~plan_hash) {
            if (err) fprintf(err, "kmer_scrub_count: the ranks computed different work plans for %s (rank %u: %016llx) -- check "
                                  "SK_SPLIT_BYTES / SK_NO_SPLIT and that every rank sees the same files; nothing is reported\n",
                             list_path, rank, (unsigned long long)plan_hash);
            pool.rc = SK_E_PLAN; plan_only = 1;           /* (fall through to the clean-up) */
        }
    }
    pool.item = (scan_item *)malloc(((size_t)nall + 1) * sizeof *pool.item);
    for (i = 0; i < nall; i++) {
        if (owner[i] == rank) pool.item[pool.nitem++] = all[i];
        else free(all[i].path);
    }
    free(all); free(est); free(owner);

    {   /* with fewer files than half the cores, a file's inflate and its record parsing take a core each; with
         * fewer still, the threads left over inflate inside the files (a speculative segment costs about twice a
         * serial one, so it takes three threads per file to be worth it).  SK_GZ_THREADS sets the number per file. */
        const long ncpu = sk_cpu_budget();
        const char *gzt = getenv("SK_GZ_THREADS");
        pool.pipe = nthreads > 1 && (long)pool.nitem * 2 <= ncpu;
        if (pool.pipe && pool.nitem && nthreads / (int)pool.nitem >= 3) pool.pipe = nthreads / (int)pool.nitem;
        if (gzt && nthreads > 1) pool.pipe = atoi(gzt) < 1 ? 1 : atoi(gzt);
        if (pool.pipe > 16) pool.pipe = 16;
#ifdef SK_EXPERIMENTS
        pool.gpu_inflate = getenv("SK_GPU_INFLATE") ? atoi(getenv("SK_GPU_INFLATE")) : 0;
        if (pool.gpu_inflate && nthreads > 1) {               /* the device's decoder is fed by threads of its own, beside the host's decode threads */
            pool.dev_workers = getenv("SK_GPU_INFLATE_WORKERS") ? atoi(getenv("SK_GPU_INFLATE_WORKERS")) : 4;
            if (pool.dev_workers < 0) pool.dev_workers = 0;
            if (pool.dev_workers > 16) pool.dev_workers = 16;
            nthreads += pool.dev_workers;
        }
#endif
    }
    if (plan_only) {
        /* nothing is scanned */
    } else if (nthreads == 1) {                               /* strict sequence, as the reference */
        scan_worker seq;
        int seq_rc = SK_OK;
        if (pool.nitem) seq_rc = worker_init(&seq, &pool);
        for (i = 0; i < pool.nitem; i++) {
            uint64_t b = 0;
            int64_t rc;
            progress_upto(&pool, pool.item[i].line);
            rc = seq_rc != SK_OK ? seq_rc : worker_item(&seq, &pool.item[i], &b);
            pool.bases += b;
            if (rc < 0) { pool.rc = (int)rc; pool.rc_index = i; break; }
        }
        if (pool.nitem) worker_done(&seq);
    } else if (pool.nitem) {
        pthread_t *th;
        if ((uint32_t)nthreads > pool.nitem) nthreads = (int)pool.nitem;
        th = (pthread_t *)malloc((size_t)nthreads * sizeof *th);
        for (i = 0; i < (uint32_t)nthreads; i++) pthread_create(&th[i], NULL, pool_worker, &pool);
        for (i = 0; i < (uint32_t)nthreads; i++) pthread_join(th[i], NULL);
        free(th);
        if (pool.timing && err)
            fprintf(err, "kmer_scrub_count timing: %s: %d decode threads, %u items, %llu chunks; summed over the threads: in items %.2f s, "
                         "of it on a core %.2f s, waiting for the submit lock %.2f s, inside the submit %.2f s, waiting for a buffer's copy %.2f s, "
                         "packing %.2f s (%llu chunks went up packed)\n",
                    list_path, nthreads, pool.nitem, (unsigned long long)pool.nchunks, pool.t_item, pool.t_cpu, pool.t_submit_wait,
                    pool.t_submit, pool.t_ticket, pool.t_pack, (unsigned long long)pool.npacked);
    }
    if (!plan_only && ctx) {
        /* The agreement after the scan: what went wrong, anywhere.  The first list line whose file could not be opened (the
         * reference stops there: the earliest line wins), "a cut did not hold", "a device error" -- one max all-reduce, and
         * from here on every rank holds the same status. */
        uint64_t v[3] = {0, 0, 0};
        int arc;
        if (pool.rc == SK_E_OPEN) v[0] = (uint64_t)UINT32_MAX - pool.item[pool.rc_index].line + 1;       /* (larger = earlier) */
        else if (pool.rc == SK_E_SPLIT) v[1] = 1;
        else if (pool.rc != SK_OK) v[2] = (uint64_t)(uint32_t)(-pool.rc);
        arc = sk_comm_max_u64(ctx, v, 3);
        if (arc != SK_OK) {
            if (err) fprintf(err, "kmer_scrub_count: %s (%s)\n", sk_strerror(arc), sk_last_error(ctx));
            pool.rc = arc; by_other = 1;
        } else if (v[2]) {
            if (pool.rc == SK_OK || pool.rc == SK_E_OPEN || pool.rc == SK_E_SPLIT) {
                pool.rc = -(int)(uint32_t)v[2]; by_other = 1;
                if (err) fprintf(err, "kmer_scrub_count: another rank reported an error while scanning %s: %s\n", list_path, sk_strerror(pool.rc));
            }
        } else if (v[0]) {
            open_line = (uint32_t)((uint64_t)UINT32_MAX - (v[0] - 1));
            if (!(pool.rc == SK_E_OPEN && pool.item[pool.rc_index].line == open_line)) by_other = 1;   /* (the line's owner says which file) */
            pool.rc = SK_E_OPEN;
        } else if (v[1]) {
            if (pool.rc != SK_E_SPLIT) by_other = 1;
            pool.rc = SK_E_SPLIT;
        }
    }
    if (!plan_only) {
        /* What the reference has said by now.  All went well: every progress line, every skip message.  A file could not be
         * opened: it stopped right there (src/genome_compare.c:195-198) -- the progress file ends with that file's line
         * (lines of later files that other decode threads had taken meanwhile are cut off again) and no skip message of a
         * later line was printed.  (With several ranks only rank 0 writes; it knows the first failing line of ANY rank from
         * the agreement above.) */
        uint32_t upto = pool.nll;
        if (pool.rc == SK_E_OPEN) upto = (open_line != UINT32_MAX ? open_line : pool.item[pool.rc_index].line) + 1;
        if (pool.rc == SK_OK || pool.rc == SK_E_OPEN) {
            if (upto) progress_upto(&pool, upto - 1);
            if (pool.progress && upto < pool.ll_next && upto > 0 && fflush(pool.progress) == 0 &&
                ftruncate(fileno(pool.progress), (off_t)pool.ll[upto - 1].end) == 0)
                fseek(pool.progress, pool.ll[upto - 1].end, SEEK_SET);
        }
        if (err && rank == 0 && (pool.rc == SK_OK || pool.rc == SK_E_OPEN))
            for (i = 0; i < upto; i++)
                if (pool.ll[i].skipped) fprintf(err, "skipping %s (identical match)\n", pool.ll[i].text);
    }
    if (plan_only || by_other) {
        /* (said above or by the rank it happened to, or nothing to say) */
    } else if (pool.rc == SK_E_OPEN) {
        if (err) fprintf(err, "could not read file %s in GEN_calculate_kmer_count()\n", pool.item[pool.rc_index].path);
    } else if (pool.rc == SK_E_SPLIT && guard && guard->armed) {
        /* (said nothing: the caller scans the list again, uncut) */
    } else if (pool.rc == SK_E_SPLIT) {
        if (err) fprintf(err, "kmer_scrub_count: %s could not be cut at record boundaries (bytes %llu-%llu): nothing is reported; "
                              "run again with SK_NO_SPLIT=1\n", pool.item[pool.rc_index].path,
                         (unsigned long long)pool.item[pool.rc_index].a, (unsigned long long)pool.item[pool.rc_index].b);
    } else if (pool.rc != SK_OK) {
        if (err) fprintf(err, "kmer_scrub_count: device error while scanning %s: %s (%s)\n", pool.item[pool.rc_index].path,
                         sk_strerror(pool.rc), sk_last_error(ctx));
    }
    for (i = 0; i < pool.nitem; i++) free(pool.item[i].path);
    free(pool.item);
    for (i = 0; i < pool.nll; i++) free(pool.ll[i].text);
    free(pool.ll);
    pthread_mutex_destroy(&pool.submit_mu);
    pthread_mutex_destroy(&pool.queue_mu);
    if (bases && !(pool.rc == SK_E_SPLIT && guard && guard->armed)) *bases += pool.bases;
    return pool.rc;
}

static int scan_list_impl(sk_ctx *ctx, const char *list_path, const char *skip, uint32_t col, FILE *progress,
                          FILE *err, uint32_t rank, uint32_t world, uint64_t *bases, int plan_only, const plan_report *report, int no_split)
{
    split_guard guard = {NULL, -1, 0};
    int rc = scan_list_once(ctx, list_path, skip, col, progress, err, rank, world, bases, plan_only, report, no_split, &guard);
    if (rc == SK_E_SPLIT && guard.armed) {               /* (armed on every rank of a coordinated run, or on none) */
        rc = sk_counts_set(ctx, col, guard.snap);
        if (rc == SK_OK && progress && guard.prog_at >= 0 && fflush(progress) == 0 && ftruncate(fileno(progress), (off_t)guard.prog_at) == 0)
            fseek(progress, guard.prog_at, SEEK_SET);
        if (rc != SK_OK && err) fprintf(err, "kmer_scrub_count: could not put the counters back: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx));
        if (getenv("SK_TIMING") && err && rc == SK_OK) fprintf(err, "kmer_scrub_count timing: a cut of %s did not hold; the list is scanned again uncut\n", list_path);
        /* the second scan opens with an agreement too: a rank whose restore failed says so there and everybody leaves */
        rc = scan_list_once(ctx, list_path, skip, col, progress, err, rank, world, bases, plan_only, report, rc == SK_OK ? 1 : 2, NULL);
    }
    free(guard.snap);
    return rc;
}

int skh_scan_list(sk_ctx *ctx, const char *list_path, const char *skip, uint32_t col, FILE *progress,
                  FILE *err, uint32_t rank, uint32_t world, uint64_t *bases)
{
    return scan_list_impl(ctx, list_path, skip, col, progress, err, rank, world, bases, 0, NULL, 0);
}

int skh_scan_list_uncut(sk_ctx *ctx, const char *list_path, const char *skip, uint32_t col, FILE *progress,
                        FILE *err, uint32_t rank, uint32_t world, uint64_t *bases)
{
    return scan_list_impl(ctx, list_path, skip, col, progress, err, rank, world, bases, 0, NULL, 1);
}

int skh_list_plan_hash(const char *list_path, const char *skip, uint32_t world, uint64_t *hash)
{
    plan_report rp = {hash, NULL, 0, NULL};
    if (!list_path || !hash) return SK_E_ARG;
    return scan_list_impl(NULL, list_path, skip, 0, NULL, NULL, 0, world ? world : 1, NULL, 1, &rp, 0);
}

int skh_list_plan_owners(const char *list_path, const char *skip, uint32_t world, uint32_t *owner, uint32_t cap, uint32_t *nlines)
{
    plan_report rp = {NULL, owner, cap, nlines};
    if (!list_path || !nlines || (cap && !owner)) return SK_E_ARG;
    return scan_list_impl(NULL, list_path, skip, 0, NULL, NULL, 0, world ? world : 1, NULL, 1, &rp, 0);
}

/* =========================================================================================
 * output
 * ======================================================================================= */

static char *put_i32(char *p, int32_t v)             /* "%d" of an unsigned counter */
{
    char tmp[12];
    int n = 0;
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    if (v < 0) *p++ = '-';
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}

static int cmp_u64(const void *a, const void *b) { const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return x < y ? -1 : x > y; }

int skh_print_counts(sk_ctx *ctx, const skh_keyset *ks, FILE *out, int with_drug_column)
{
    const uint32_t ncols = with_drug_column ? 4 : 3, n = ks->nrows;
    uint32_t *cols[4] = {NULL, NULL, NULL, NULL};
    uint32_t c, r, w = 0, *worder;
    char *buf, *p;
    int rc = SK_OK;
    const size_t BUF = 1 << 20;
    if (sk_table_cols(ctx) < ncols || sk_table_rows(ctx) != n) return SK_E_STATE;
    for (c = 0; c < ncols && rc == SK_OK; c++) {
        cols[c] = (uint32_t *)malloc((size_t)(n ? n : 1) * sizeof(uint32_t));
        rc = sk_counts_fetch(ctx, c, cols[c]);
    }
    /* wide keys in ascending row order (an IUPAC-riddled strain has hundreds of thousands: sort, not insert) */
    worder = (uint32_t *)malloc((size_t)(ks->nwide ? ks->nwide : 1) * sizeof(uint32_t));
    {
        uint64_t *pair = (uint64_t *)malloc((size_t)(ks->nwide ? ks->nwide : 1) * sizeof(uint64_t));
        for (c = 0; c < ks->nwide; c++) pair[c] = ((uint64_t)ks->wide_rows[c] << 32) | c;
        qsort(pair, ks->nwide, sizeof *pair, cmp_u64);
        for (c = 0; c < ks->nwide; c++) worder[c] = (uint32_t)pair[c];
        free(pair);
    }
    if (rc == SK_OK) {
        buf = (char *)malloc(BUF + 256);
        p = buf;
        fputs("#kmer\treference_count\tpangenome_count\tmetagenome_count\tdrug_count\n", out);
        for (r = 0; r < n; r++) {
            if (ks->packed[r] != SK_KEY_NONE) { decode_key(ks->packed[r], p); p += SK_K; }
            else {
                const char *wk = ks->wide_keys + (size_t)worder[w++] * 32;
                size_t l = strlen(wk);
                memcpy(p, wk, l);
                p += l;
            }
            for (c = 0; c < ncols; c++) { *p++ = '\t'; p = put_i32(p, (int32_t)cols[c][r]); }
            *p++ = '\n';
            if ((size_t)(p - buf) >= BUF) { fwrite(buf, 1, (size_t)(p - buf), out); p = buf; }
        }
        fwrite(buf, 1, (size_t)(p - buf), out);
        free(buf);
    }
    for (c = 0; c < 4; c++) free(cols[c]);
    free(worder);
    return rc;
}

/* =========================================================================================
 * the program
 * ======================================================================================= */

static void usage(FILE *err)
{
    fputs("Usage: kmer_scrub_count -r <reference genome>  -A <file with multiple genome filenames> "
          "-B <file with multiple metagenome filenames> -C <(optional) file with multiple genome "
          "filenames of drug strains> -p [progress output file, optional]\n", err);
}

static int env_int(const char *a, const char *b, const char *c3, int dflt)
{
    const char *v = getenv(a);
    if (!v && b) v = getenv(b);
    if (!v && c3) v = getenv(c3);
    return v ? atoi(v) : dflt;
}

/* Multi-GPU use (new; one process per GPU, any launcher that sets the usual variables):
 *   SK_WORLD_SIZE | WORLD_SIZE | OMPI_COMM_WORLD_SIZE,  SK_RANK | RANK | OMPI_COMM_WORLD_RANK,
 *   SK_LOCAL_RANK | LOCAL_RANK | OMPI_COMM_WORLD_LOCAL_RANK (HIP device; SK_DEVICE overrides),
 *   SK_RCCL_ID_FILE (rendezvous file for the RCCL unique id; default /tmp/sk_rccl_id.<MASTER_PORT|uid>).
 * List items (files, byte ranges of big plain-text files) are dealt to the ranks by size -- the same plan on every rank,
 * compared before anyone scans (SK_E_PLAN) --, every rank's counters are summed with one RCCL all-reduce, rank 0 alone
 * writes stdout, the progress file and the skip/progress messages.  The sequence of collectives is the same on every rank
 * whatever fails where: set-up (in the rendezvous), table load (one sum), per list scan two agreements, one sum, the all-reduce. */
int skh_kmer_scrub_count_main(int argc, char **argv, FILE *out, FILE *err)
{
    const char *A = NULL, *B = NULL, *C = NULL, *R = NULL, *P = NULL, *env;
    FILE *progress = NULL;
    skh_keyset ks;
    sk_ctx *ctx = NULL;
    int c, rc, status = 1, failed = 0;
    const int world = env_int("SK_WORLD_SIZE", "WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", 1);
    const int rank = env_int("SK_RANK", "RANK", "OMPI_COMM_WORLD_RANK", 0);
    int device = env_int("SK_LOCAL_RANK", "LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", 0);
    const int use_comm = world > 1 || getenv("SK_FORCE_COMM") != NULL;
    uint32_t nfailed = 0;
    sk_ctxjob cj;
    int crc;
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
    double scrub_fraction = -1.0;                    /* >= 0: print the scrub filter's result instead of the table */
    int scrub_independent = 0, j;
    const char *scrub_out = NULL;                    /* --scrub-out FILE: the filter's result goes there instead of stdout */
    int detect_argc = 0;                             /* --detect ...: strain_detect's arguments, run on the resident table */
    char **detect_argv = NULL;

    /* Extension (not in the reference): "--scrub <min_fraction>" [--independent] runs step 2 of the
     * workflow (scripts/kmer_scrub_filter.py -s <table> -m <min_fraction> [-i]) on the counters while
     * they are still on the device and prints ITS output; the 38-bytes-per-k-mer table is never written.
     * The words are taken out of argv before getopt, which only knows the reference's letters. */
    /* "--detect <strain_detect arguments>" (extension; needs --scrub): after steps 1 and 2 the same process goes on into
     * step 3 on the SAME resident table (key set, row order, device table, filters and strain text built once; SURVEY
     * 8(f3)): -r is this run's, -a is the list --scrub just produced; everything behind --detect is strain_detect's
     * command line (-B/-b/-c/-t/-g/-o, --coverage-depth ...).  "--scrub-out FILE" keeps the informative list. */
    for (c = 1; c < argc; c++)
        if (!strcmp(argv[c], "--detect")) {
            detect_argv = argv + c;                  /* argv[0] of the second command line: ignored there */
            detect_argc = argc - c;
            argc = c;
            break;
        }
    for (c = 1, j = 1; c < argc; c++) {
        if (!strcmp(argv[c], "--independent")) { scrub_independent = 1; continue; }
        if (!strcmp(argv[c], "--scrub-out") && c + 1 < argc) { scrub_out = argv[++c]; continue; }
        if (!strncmp(argv[c], "--scrub", 7) && (argv[c][7] == 0 || argv[c][7] == '=')) {
            const char *v = argv[c][7] ? argv[c] + 8 : (c + 1 < argc ? argv[++c] : "");
            char *e;
            scrub_fraction = strtod(v, &e);
            if (e == v || *e || scrub_fraction < 0.0 || scrub_fraction > 1.0) {
                fprintf(err, "kmer_scrub_count: --scrub needs a fraction between 0.0 and 1.0\n");
                return 1;
            }
            continue;
        }
        argv[j++] = argv[c];
    }
    argc = j;

    optind = 1;
    while ((c = getopt(argc, argv, "A:B:C:r:p:Hhud")) != -1) {
        switch (c) {
        case 'A': A = optarg; break;
        case 'B': B = optarg; break;
        case 'C': C = optarg; break;
        case 'r': R = optarg; break;
        case 'p': P = optarg; break;
        case 'd': break;
        default:  usage(err); break;                 /* -h/-u/-H/unknown: print and carry on */
        }
    }
    if (!R || !A || !B) { usage(err); return 1; }
    if (detect_argv && (scrub_fraction < 0.0 || world > 1)) {
        fprintf(err, "kmer_scrub_count: --detect needs --scrub <min_fraction> and a single process\n");
        return 1;
    }
    if (world < 1 || rank < 0 || rank >= world) { fprintf(err, "kmer_scrub_count: bad rank %d of %d\n", rank, world); return 1; }
    if (P && rank == 0) {
        progress = fopen(P, "w");
        if (!progress) { fprintf(err, "could not open progress file %s\n", P); return 1; }
        fputs("adding kmer counts for:\n", progress);
    }
    if ((env = getenv("SK_DEVICE")) != NULL) device = atoi(env);

    memset(&ks, 0, sizeof ks);
    t0 = now_s();
    sk_ctxjob_start(&cj, device);                      /* the HIP runtime comes up while the key set is built */
    rc = skh_keyset_from_file(&ks, R, SK_REF_TABLE_SLOTS, 1, 1);
    t1 = now_s();
    crc = sk_ctxjob_join(&cj, &ctx);
    t2 = now_s();
    {   /* set-up problems are reported here; with several ranks the others are told before anyone enters the collective */
        int setup_failed = 0;
        if (rc == SK_E_OPEN) { fprintf(err, "could not read file %s GEN_hash_sequences_set_count_vec()\n", R); setup_failed = 1; }
        else if (rc != SK_OK) { fprintf(err, "kmer_scrub_count: %s\n", sk_strerror(rc)); setup_failed = 1; }
        else if (ks.short_records && rank == 0)
            fprintf(err, "kmer_scrub_count: skipped %llu reference record(s) shorter than %d bases "
                         "(the original program crashes on these)\n", (unsigned long long)ks.short_records, SK_K - 1);
        if (crc != SK_OK) {
            if (!setup_failed) fprintf(err, "kmer_scrub_count: cannot use HIP device %d: %s\n", device, sk_strerror(crc));
            setup_failed = 1; ctx = NULL;
        }
        if (use_comm) {
            char path[256];
            int timeout_s = 120;
            if ((env = getenv("SK_RCCL_ID_FILE")) != NULL) snprintf(path, sizeof path, "%s", env);
            else if ((env = getenv("MASTER_PORT")) != NULL) snprintf(path, sizeof path, "/tmp/sk_rccl_id.%s", env);
            else snprintf(path, sizeof path, "/tmp/sk_rccl_id.%d", (int)getuid());
            if ((env = getenv("SK_RENDEZVOUS_TIMEOUT")) != NULL && atoi(env) > 0) timeout_s = atoi(env);
            {   /* RCCL prints its version banner on stdout, which here is the TSV: park fd 1 on stderr meanwhile */
                int saved;
                fflush(stdout);
                saved = dup(1);
                dup2(2, 1);
                rc = sk_comm_init_ex(ctx, rank, world, path, timeout_s, setup_failed);
                fflush(stdout);
                dup2(saved, 1);
                close(saved);
            }
            if (rc != SK_OK && !setup_failed)
                fprintf(err, "kmer_scrub_count: RCCL rendezvous failed: %s (%s)\n", sk_strerror(rc), ctx ? sk_last_error(ctx) : "no context");
            if (rc != SK_OK) goto done;
        } else if (setup_failed) goto done;
    }
    rc = skh_keyset_load(ctx, &ks, detect_argv ? 6 : 4);        /* (strain_detect's table has six columns) */
    t3 = now_s();
    if (rc != SK_OK) { fprintf(err, "kmer_scrub_count: table load failed: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx)); failed = 1; }
    if (rc == SK_OK && world > 1) rc = sk_counts_zero(ctx, 0);      /* column 0 must not be summed world times: keep it on rank 0 */
    if (rc == SK_OK && world > 1 && rank == 0) rc = sk_counts_set(ctx, 0, ks.first_count);
    if (rc != SK_OK && !failed) {                                   /* (a wrong column 0 must not reach the all-reduce) */
        fprintf(err, "kmer_scrub_count: could not set up the reference_count column: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx));
        failed = 1;
    }

    if (use_comm) {
        /* a rank whose table did not load must not leave the others alone in the lists' agreements: from here to the big
         * all-reduce every rank issues the same collectives (skh_scan_list returns the same status on all of them) */
        rc = sk_comm_sum_u32(ctx, (uint32_t)failed, &nfailed);
        if (rc != SK_OK) { fprintf(err, "kmer_scrub_count: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx)); goto done; }
        if (nfailed) {
            if (!failed) fprintf(err, "kmer_scrub_count: another rank could not load the table; nothing is reported\n");
            goto done;
        }
    }
    if (!failed && skh_scan_list(ctx, A, NULL, 1, progress, err, (uint32_t)rank, (uint32_t)world, NULL) != SK_OK) failed = 1;
    if (!failed && skh_scan_list(ctx, B, NULL, 2, progress, err, (uint32_t)rank, (uint32_t)world, NULL) != SK_OK) failed = 1;
    if (!failed && C && skh_scan_list(ctx, C, R, 3, progress, err, (uint32_t)rank, (uint32_t)world, NULL) != SK_OK) failed = 1;
    if (use_comm) {
        /* agree once more (the lists' own agreements have made `failed` the same everywhere: a belt to those braces) */
        rc = sk_comm_sum_u32(ctx, (uint32_t)failed, &nfailed);
        if (rc != SK_OK) { fprintf(err, "kmer_scrub_count: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx)); goto done; }
        if (nfailed) goto done;
        rc = sk_counts_allreduce(ctx, NULL);
        if (rc != SK_OK) { fprintf(err, "kmer_scrub_count: all-reduce failed: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx)); goto done; }
    } else if (failed) goto done;
    t4 = now_s();
    if (rank == 0 && scrub_fraction >= 0.0) {
        FILE *so = out;
        char tmp_path[64] = "";
        const char *list_path = scrub_out;
        if (detect_argv && !scrub_out) {              /* step 3 reads the list from a file: a temporary one, copied to stdout */
            int fd;
            snprintf(tmp_path, sizeof tmp_path, "/tmp/sk_scrubbed.XXXXXX");
            fd = mkstemp(tmp_path);
            if (fd < 0) { fprintf(err, "kmer_scrub_count: cannot create a temporary file for the informative k-mers\n"); goto done; }
            close(fd);
            list_path = tmp_path;
        }
        if (list_path && !(so = fopen(list_path, "w"))) { fprintf(err, "kmer_scrub_count: cannot write %s\n", list_path); goto done; }
        status = skh_scrub_filter_resident(ctx, &ks, C != NULL, scrub_fraction, scrub_independent, so, err);
        if (so != out) fclose(so);
        if (tmp_path[0] && status == 0) {             /* no --scrub-out: the list still goes to stdout, as --scrub alone does */
            FILE *f = fopen(tmp_path, "r");
            char blk[65536];
            size_t got;
            while (f && (got = fread(blk, 1, sizeof blk, f)) > 0) fwrite(blk, 1, got, out);
            if (f) fclose(f);
            fflush(out);
        }
        t5 = now_s();
        if (status == 0 && detect_argv) {
            sk_ctx *c2 = ctx;
            ctx = NULL;                               /* ctx and ks now belong to strain_detect, which releases them */
            status = skh_strain_detect_resident(c2, &ks, list_path, detect_argc, detect_argv, out, err);
        }
        if (tmp_path[0]) unlink(tmp_path);
        goto done;
    }
    if (rank == 0) {
        rc = skh_print_counts(ctx, &ks, out, C != NULL);
        if (rc != SK_OK) { fprintf(err, "kmer_scrub_count: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx)); goto done; }
    }
    t5 = now_s();
    status = 0;
done:
    if (getenv("SK_TIMING") && t5 > 0)
        fprintf(err, "kmer_scrub_count timing: key set %.2f s (+%.2f s more for the HIP context), table load %.2f s, scans %.2f s, "
                     "%s %.2f s\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, scrub_fraction >= 0.0 ? "filter + print" : "print", t5 - t4);
    if (ctx) sk_ctx_destroy(ctx);
    skh_keyset_free(&ks);
    if (progress) fclose(progress);
    return status;
}
