/* inflate_pair_probe.c -- EXPERIMENT (not product code): how much does one thread gain by decoding TWO gzip streams
 * interleaved symbol by symbol?  The serial inflate loop of strainer2_amd/csrc/sk_gzfast.h is a chain of dependent
 * table loads; tools/smt_probe.sh shows a core has room for a second chain (1.56x with two decoders on its two SMT
 * threads).  This program decodes the same single-member .gz file (a) once, (b) as two streams one after the other,
 * (c) as two streams interleaved in one loop (skz_block_pair below), and prints the rates.  Outputs are checked by
 * CRC-32 and length against the gzip trailer.
 *     gcc -O2 -o /tmp/pair tools/inflate_pair_probe.c -lpthread && /tmp/pair reads.fq.gz
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <time.h>
#include "../strainer2_amd/csrc/sk_gzfast.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static int sink(void *u, const unsigned char *d, size_t n) { (void)d; *(size_t *)u += n; return 0; }

typedef struct {
    skz_stream s;
    skz_tables *dyn;
    const skz_tables *fixed, *cur;
    int final, done, bad;
    size_t got;
} job;

/* from "between blocks" to "inside a Huffman block" (cur set) or done */
static void job_next_block(job *j)
{
    skz_stream *s = &j->s;
    for (;;) {
        uint32_t final, type;
        if (j->final) {                                     /* the member is over: trailer */
            skz_flush(s, 0);
            s->in -= s->bitcnt >> 3;
            if (s->in + 8 > s->in_end) { j->bad = 1; j->done = 1; return; }
            {
                const uint32_t crc = (uint32_t)s->in[0] | ((uint32_t)s->in[1] << 8) | ((uint32_t)s->in[2] << 16) | ((uint32_t)s->in[3] << 24);
                const uint32_t isz = (uint32_t)s->in[4] | ((uint32_t)s->in[5] << 8) | ((uint32_t)s->in[6] << 16) | ((uint32_t)s->in[7] << 24);
                if (crc != s->crc || isz != (uint32_t)s->total) j->bad = 1;
            }
            j->done = 1;
            return;
        }
        SKZ_REFILL(s);
        final = SKZ_BITS(s, 1); SKZ_DROP(s, 1);
        type = SKZ_BITS(s, 2); SKZ_DROP(s, 2);
        j->final = (int)final;
        if (type == 0) {
            uint32_t len, nlen;
            SKZ_DROP(s, s->bitcnt & 7u);
            s->in -= s->bitcnt >> 3;
            s->bitbuf = 0; s->bitcnt = 0;
            if (s->in + 4 > s->in_end) { j->bad = j->done = 1; return; }
            len = (uint32_t)s->in[0] | ((uint32_t)s->in[1] << 8);
            nlen = (uint32_t)s->in[2] | ((uint32_t)s->in[3] << 8);
            s->in += 4;
            if ((len ^ 0xFFFFu) != nlen || (size_t)(s->in_end - s->in) < len) { j->bad = j->done = 1; return; }
            while (len) {
                size_t room = (size_t)(s->out_end - s->out), take;
                if (room < 4096) { skz_flush(s, 1); room = (size_t)(s->out_end - s->out); }
                take = len < room ? len : room;
                memcpy(s->out, s->in, take);
                s->out += take; s->in += take; len -= (uint32_t)take;
            }
            continue;
        }
        if (type == 1) { j->cur = j->fixed; return; }
        if (type == 2) { if (skz_read_dynamic(s, j->dyn, 0)) { j->bad = j->done = 1; return; } j->cur = j->dyn; return; }
        j->bad = j->done = 1;
        return;
    }
}

static void job_open(job *j, const unsigned char *data, size_t n, const skz_tables *fixed)
{
    memset(j, 0, sizeof *j);
    j->dyn = (skz_tables *)malloc(sizeof *j->dyn);
    j->fixed = fixed;
    j->s.out_base = (unsigned char *)malloc(SKZ_WINDOW + SKZ_OUT_CHUNK + 1024);
    j->s.out_end = j->s.out_base + SKZ_WINDOW + SKZ_OUT_CHUNK;
    j->s.out = j->s.out_flushed = j->s.out_base;
    j->s.sink = sink; j->s.user = &j->got;
    j->s.data = data; j->s.in = data + skz_header(data, n); j->s.in_end = data + n;
    job_next_block(j);
}
static void job_close(job *j) { free(j->dyn); free(j->s.out_base); }

/* one stream's block the ordinary way */
static void job_block_single(job *j)
{
    const int rc = skz_block(&j->s, j->cur);
    if (rc) { j->bad = j->done = 1; return; }
    job_next_block(j);
}

/* ---- the interleaved loop: one decode step of A, one of B, ... ---------------------------------------------
 * A step = refill + up to three literal entries, or one match.  Leaves (with both states saved) when either stream
 * reaches its end-of-block code, runs low on output room, or comes within 16 bytes of the end of its input.
 * Returns: 0 A at end of block, 1 B at end of block, 2 A needs room, 3 B needs room, 4 A near its end, 5 B near its end,
 * -1 / -2 A / B corrupt. */
#define KIND(e) (((e) >> 4) & 15u)
#define LITS(e) (KIND(e) <= (uint32_t)SKZ_K_LIT2)
#define FILL(X) do { uint64_t w_; memcpy(&w_, in##X, 8); bb##X |= w_ << bc##X; in##X += (63u - bc##X) >> 3; bc##X |= 56u; } while (0)
#define EMIT(X, e) do { bb##X >>= (e) & 15u; bc##X -= (e) & 15u; out##X[0] = (unsigned char)((e) >> 16); out##X[1] = (unsigned char)((e) >> 24); \
                        out##X += 1u + (KIND(e) == SKZ_K_LIT2); } while (0)
#define STEP(X, EOB_EVENT, BAD_EVENT) do {                                                                     \
        uint32_t e;                                                                                            \
        FILL(X);                                                                                               \
        e = lt##X[bb##X & ((1u << SKZ_LITLEN_BITS) - 1u)];                                                     \
        if (LITS(e)) {                                                                                         \
            EMIT(X, e); e = lt##X[bb##X & ((1u << SKZ_LITLEN_BITS) - 1u)];                                     \
            if (LITS(e)) {                                                                                     \
                EMIT(X, e); e = lt##X[bb##X & ((1u << SKZ_LITLEN_BITS) - 1u)];                                 \
                if (LITS(e)) { EMIT(X, e); break; }                                                            \
            }                                                                                                  \
            FILL(X);                                                                                           \
        }                                                                                                      \
        if (KIND(e) == SKZ_K_SUB) {                                                                            \
            bb##X >>= SKZ_LITLEN_BITS; bc##X -= SKZ_LITLEN_BITS;                                               \
            e = lt##X[(e >> 16) + (uint32_t)(bb##X & (((uint64_t)1 << ((e >> 8) & 255u)) - 1u))];              \
            if (KIND(e) == SKZ_K_LIT) { bb##X >>= e & 15u; bc##X -= e & 15u; *out##X++ = (unsigned char)(e >> 16); break; } \
        }                                                                                                      \
        bb##X >>= e & 15u; bc##X -= e & 15u;                                                                   \
        if (KIND(e) == SKZ_K_LEN) {                                                                            \
            const uint32_t xb = (e >> 8) & 255u;                                                               \
            uint32_t len, dist, d, db;                                                                         \
            unsigned char *dst; const unsigned char *src;                                                      \
            len = (e >> 16) + (uint32_t)(bb##X & (((uint64_t)1 << xb) - 1u)); bb##X >>= xb; bc##X -= xb;       \
            d = dt##X[bb##X & ((1u << SKZ_DIST_BITS) - 1u)];                                                   \
            if (KIND(d) == SKZ_K_SUB) {                                                                        \
                bb##X >>= SKZ_DIST_BITS; bc##X -= SKZ_DIST_BITS;                                               \
                d = dt##X[(d >> 16) + (uint32_t)(bb##X & (((uint64_t)1 << ((d >> 8) & 255u)) - 1u))];          \
            }                                                                                                  \
            bb##X >>= d & 15u; bc##X -= d & 15u;                                                               \
            if (KIND(d) != SKZ_K_DIST) { ev = BAD_EVENT; goto leave; }                                         \
            db = (d >> 8) & 255u;                                                                              \
            dist = (d >> 16) + (uint32_t)(bb##X & (((uint64_t)1 << db) - 1u)); bb##X >>= db; bc##X -= db;      \
            if (dist > (size_t)(out##X - base##X)) { ev = BAD_EVENT; goto leave; }                             \
            dst = out##X; src = dst - dist; out##X += len;                                                     \
            if (dist >= 8) { do { uint64_t w; memcpy(&w, src, 8); memcpy(dst, &w, 8); src += 8; dst += 8; } while (dst < out##X); } \
            else if (dist == 1) memset(dst, *src, len);                                                        \
            else { do { *dst++ = *src++; } while (dst < out##X); }                                             \
            break;                                                                                             \
        }                                                                                                      \
        ev = KIND(e) == SKZ_K_EOB ? EOB_EVENT : BAD_EVENT;                                                     \
        goto leave;                                                                                            \
    } while (0)

static int skz_block_pair(skz_stream *a, const skz_tables *ta, skz_stream *b, const skz_tables *tb)
{
    const uint32_t *const ltA = ta->litlen, *const dtA = ta->dist, *const ltB = tb->litlen, *const dtB = tb->dist;
    const unsigned char *inA = a->in, *inB = b->in;
    const unsigned char *const fastA = a->in_end - 16, *const fastB = b->in_end - 16;
    uint64_t bbA = a->bitbuf, bbB = b->bitbuf;
    unsigned bcA = a->bitcnt, bcB = b->bitcnt;
    unsigned char *outA = a->out, *outB = b->out;
    unsigned char *const baseA = a->out_base, *const baseB = b->out_base;
    const unsigned char *const limA = a->out_end - (6 + 2 * 258 + 16), *const limB = b->out_end - (6 + 2 * 258 + 16);
    int ev;
    for (;;) {
        if (outA > limA) { ev = 2; break; }
        if (outB > limB) { ev = 3; break; }
        if (inA > fastA) { ev = 4; break; }
        if (inB > fastB) { ev = 5; break; }
        STEP(A, 0, -1);
        STEP(B, 1, -2);
    }
leave:
    a->in = inA; a->bitbuf = bbA; a->bitcnt = bcA; a->out = outA;
    b->in = inB; b->bitbuf = bbB; b->bitcnt = bcB; b->out = outB;
    return ev;
}

static void run_pair(job *a, job *b)
{
    while (!a->done || !b->done) {
        if (a->done) { job_block_single(b); continue; }
        if (b->done) { job_block_single(a); continue; }
        switch (skz_block_pair(&a->s, a->cur, &b->s, b->cur)) {
        case 0: job_next_block(a); break;
        case 1: job_next_block(b); break;
        case 2: skz_flush(&a->s, 1); break;
        case 3: skz_flush(&b->s, 1); break;
        case 4: job_block_single(a); break;              /* the last bytes of the input: the ordinary decoder finishes the block */
        case 5: job_block_single(b); break;
        case -1: a->bad = a->done = 1; break;
        default: b->bad = b->done = 1; break;
        }
    }
}

int main(int argc, char **argv)
{
    struct stat st;
    unsigned char *m;
    skz_tables *fixed = (skz_tables *)malloc(sizeof *fixed);
    int fd, rep;
    if (argc < 2 || (fd = open(argv[1], O_RDONLY)) < 0 || fstat(fd, &st)) { fprintf(stderr, "usage: inflate_pair_probe file.gz\n"); return 2; }
    m = (unsigned char *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    if (m == MAP_FAILED || !skz_header(m, (size_t)st.st_size)) return 2;
    pthread_once(&skz_crc_once, skz_crc_init);
    skz_fixed_tables(fixed);
    for (rep = 0; rep < 3; rep++) {
        job a, b;
        double t0, t1, t2;
        size_t one;
        t0 = now();
        job_open(&a, m, (size_t)st.st_size, fixed);
        while (!a.done) job_block_single(&a);
        t1 = now();
        one = a.got;
        if (a.bad) { printf("single: BAD\n"); return 1; }
        job_close(&a);
        job_open(&a, m, (size_t)st.st_size, fixed);
        job_open(&b, m, (size_t)st.st_size, fixed);
        run_pair(&a, &b);
        t2 = now();
        printf("one stream: %.0f MB/s;  two streams interleaved in one thread: %.0f MB/s in total (%s, %zu + %zu bytes)  => x%.2f\n",
               (double)one / (t1 - t0) / 1e6, (double)(a.got + b.got) / (t2 - t1) / 1e6, a.bad || b.bad || a.got != one || b.got != one ? "WRONG" : "both verified",
               a.got, b.got, ((double)(a.got + b.got) / (t2 - t1)) / ((double)one / (t1 - t0)));
        job_close(&a); job_close(&b);
    }
    return 0;
}
