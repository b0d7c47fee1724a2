/* kso_oracle.c -- CPU ORACLE (test infrastructure; see kso_oracle.h for the rules).
 *
 * A from-scratch restatement of the reference's kmer_scrub_count path, kept deliberately
 * "faithful": keys are NUL-terminated ASCII strings, hashing is 32-bit djb2 over the key
 * bytes, collisions are resolved by linear probing with strcmp, and the table doubles when
 * half full.  It does the same amount of work per window as the reference, so it doubles
 * as the CPU baseline ("port") in bench.py.
 *
 * Reference citations are relative to /root/reference/.
 *
 * Deliberately DEFINED where the reference has undefined behaviour:
 *   - bytes >= 0x80 index the reference's COMPLEMENT[] with a negative subscript
 *     (src/genome_compare.c:1114,1131).  Here their complement is (char)-1.
 *   - a strain record shorter than k-1 makes the reference's window loop underflow
 *     (src/genome_compare.c:1000) and crash; here that is reported (KSO_E_SHORT_CONTIG)
 *     or, with short_policy=1, the record is skipped.
 */
#define _GNU_SOURCE
#include "kso_oracle.h"
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

/* ------------------------------------------------------------------------------------ */
/* complement map: src/BIO_sequence.c:203-213 (IUPAC table, including its K -> '.' slip)  */

static signed char g_comp[256];
static int g_comp_ready = 0;

static void comp_init(void)
{
    /* pairs "xy" mean complement(x) = y */
    static const char *pairs[] = {
        "--", "..", "^^",
        "AT", "TA", "CG", "GC", "BV", "VB", "DH", "HD", "K.", "MK", "NN",
        "RY", "YR", "SS", "UA", "WW", "XX",
        "at", "ta", "cg", "gc", "bv", "vb", "dh", "hd", "km", "mk", "nn",
        "ry", "yr", "ss", "ua", "ww", "xx", NULL };
    int i;
    if (g_comp_ready) return;
    memset(g_comp, -1, sizeof g_comp);
    for (i = 0; pairs[i]; i++) g_comp[(unsigned char)pairs[i][0]] = (signed char)pairs[i][1];
    g_comp_ready = 1;
}

/* ------------------------------------------------------------------------------------ */
/* string table: src/BIO_hash.c:14-37 (init), :129-139 (insert), :39-61 (doubling),
 * :161-172 (search), :174-188 (key order), :208-216 (djb2)                              */

typedef struct { unsigned *vec; char *key; } kso_slot;

struct kso_table {
    unsigned M;        /* slots            */
    unsigned N;        /* live keys        */
    kso_slot *slot;
    int ncols;
};

static unsigned djb2_mod(const char *s, unsigned M)
{
    unsigned h = 5381u;
    for (; *s; s++) h = h * 33u + (unsigned)(int)(signed char)*s;   /* signed bytes */
    return h % M;
}

kso_table *kso_table_new(unsigned capacity)
{
    kso_table *t = (kso_table *)malloc(sizeof *t);
    if (capacity == 0) capacity = 1000;          /* src/BIO_hash.c:9,18-19 */
    else if (capacity < 10) capacity = 10;       /* src/BIO_hash.c:6,20-21 */
    t->M = capacity;
    t->N = 0;
    t->ncols = 0;
    t->slot = (kso_slot *)calloc(t->M, sizeof *t->slot);
    comp_init();
    return t;
}

void kso_table_free(kso_table *t)
{
    unsigned i;
    if (!t) return;
    for (i = 0; i < t->M; i++)
        if (t->slot[i].vec) { free(t->slot[i].key); free(t->slot[i].vec); }
    free(t->slot);
    free(t);
}

unsigned kso_table_size(const kso_table *t)     { return t->N; }
unsigned kso_table_capacity(const kso_table *t) { return t->M; }
int      kso_table_ncols(const kso_table *t)    { return t->ncols; }

static void table_place(kso_table *t, char *owned_key, unsigned *vec);

static void table_double(kso_table *t)
{
    kso_slot *old = t->slot;
    unsigned oldM = t->M, i;
    t->M = oldM + oldM;
    t->N = 0;
    t->slot = (kso_slot *)calloc(t->M, sizeof *t->slot);
    for (i = 0; i < oldM; i++)                  /* re-insert in old slot order */
        if (old[i].vec) table_place(t, old[i].key, old[i].vec);
    free(old);
}

static void table_place(kso_table *t, char *owned_key, unsigned *vec)
{
    unsigned i = djb2_mod(owned_key, t->M);
    while (t->slot[i].vec) i = (i + 1) % t->M;
    t->slot[i].key = owned_key;
    t->slot[i].vec = vec;
    if (t->N++ >= t->M / 2) table_double(t);     /* post-increment: src/BIO_hash.c:138 */
}

static unsigned *table_find(const kso_table *t, const char *key)
{
    unsigned i = djb2_mod(key, t->M);
    while (t->slot[i].vec) {
        if (strcmp(key, t->slot[i].key) == 0) return t->slot[i].vec;
        i = (i + 1) % t->M;
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------ */
/* FASTA/FASTQ record reader with the reference parser's observable behaviour
 * (src/kseq.h:166-211 record grammar, :90-141 line reads).  Written as an explicit
 * reader over a zlib stream or a memory block.                                          */

typedef struct { char *p; size_t len, cap; } kso_buf;

typedef struct {
    gzFile gz;                 /* NULL when reading from memory */
    const unsigned char *mem; size_t mem_len, mem_pos;
    unsigned char chunk[65536];
    int beg, end, drained;
    int pending_header;        /* header char already consumed by the previous record */
    kso_buf seq, qual;
} kso_reader;

static void buf_reserve(kso_buf *b, size_t extra)
{
    if (b->len + extra + 1 > b->cap) {
        size_t nc = b->cap ? b->cap : 256;
        while (nc < b->len + extra + 1) nc *= 2;
        b->p = (char *)realloc(b->p, nc);
        b->cap = nc;
    }
}

static int rd_fill(kso_reader *r)
{
    int got;
    if (r->drained) return 0;
    if (r->gz) got = gzread(r->gz, r->chunk, sizeof r->chunk);
    else {
        size_t left = r->mem_len - r->mem_pos;
        got = (int)(left < sizeof r->chunk ? left : sizeof r->chunk);
        memcpy(r->chunk, r->mem + r->mem_pos, (size_t)got);
        r->mem_pos += (size_t)got;
    }
    r->beg = 0;
    if (got <= 0) { r->end = 0; r->drained = 1; return 0; }
    r->end = got;
    return 1;
}

static int rd_byte(kso_reader *r)
{
    if (r->beg >= r->end && !rd_fill(r)) return -1;
    return r->chunk[r->beg++];
}

/* Append one line (without its '\n') to b.  Returns -1 when no byte at all was available;
 * otherwise the new length.  A trailing '\r' is dropped when the accumulated length
 * exceeds one (src/kseq.h:136). */
static long rd_line_into(kso_reader *r, kso_buf *b)
{
    int any = 0;
    for (;;) {
        unsigned char *nl;
        int avail;
        if (r->beg >= r->end && !rd_fill(r)) break;
        any = 1;
        avail = r->end - r->beg;
        nl = (unsigned char *)memchr(r->chunk + r->beg, '\n', (size_t)avail);
        if (nl) {
            size_t n = (size_t)(nl - (r->chunk + r->beg));
            buf_reserve(b, n);
            memcpy(b->p + b->len, r->chunk + r->beg, n);
            b->len += n;
            r->beg += (int)n + 1;
            break;
        }
        buf_reserve(b, (size_t)avail);
        memcpy(b->p + b->len, r->chunk + r->beg, (size_t)avail);
        b->len += (size_t)avail;
        r->beg = r->end;
    }
    if (!any) return -1;
    buf_reserve(b, 0);
    if (b->len > 1 && b->p[b->len - 1] == '\r') b->len--;
    b->p[b->len] = '\0';
    return (long)b->len;
}

/* Consume the record name (up to the first whitespace byte).  Returns -1 when the input
 * was already exhausted, else 0; *delim gets the whitespace byte (0 at end of input). */
static int rd_skip_name(kso_reader *r, int *delim)
{
    int any = 0;
    *delim = 0;
    for (;;) {
        if (r->beg >= r->end && !rd_fill(r)) break;
        any = 1;
        while (r->beg < r->end) {
            int c = r->chunk[r->beg++];
            if (isspace(c)) { *delim = c; return 0; }
        }
    }
    return any ? 0 : -1;
}

/* Next record.  >=0: sequence length (r->seq holds it), -1: end of input,
 * -2: quality string length mismatch (src/kseq.h:166-170). */
static long rd_record(kso_reader *r)
{
    int c, delim;
    kso_buf scratch = { NULL, 0, 0 };

    if (!r->pending_header) {
        do c = rd_byte(r); while (c != -1 && c != '>' && c != '@');
        if (c == -1) return -1;
        r->pending_header = c;
    }
    r->seq.len = 0;
    r->qual.len = 0;
    if (rd_skip_name(r, &delim) < 0) return -1;
    if (delim != '\n') { rd_line_into(r, &scratch); free(scratch.p); }   /* comment */

    buf_reserve(&r->seq, 0);
    while ((c = rd_byte(r)) != -1 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;
        buf_reserve(&r->seq, 1);
        r->seq.p[r->seq.len++] = (char)c;
        rd_line_into(r, &r->seq);
    }
    if (c == '>' || c == '@') r->pending_header = c;
    buf_reserve(&r->seq, 0);
    r->seq.p[r->seq.len] = '\0';
    if (c != '+') return (long)r->seq.len;

    do c = rd_byte(r); while (c != -1 && c != '\n');   /* rest of the '+' line */
    if (c == -1) return -2;
    while (rd_line_into(r, &r->qual) >= 0 && r->qual.len < r->seq.len) { }
    r->pending_header = 0;
    if (r->qual.len != r->seq.len) return -2;
    return (long)r->seq.len;
}

static kso_reader *rd_open_file(const char *path)
{
    gzFile g = gzopen(path, "r");
    kso_reader *r;
    if (!g) return NULL;
    r = (kso_reader *)calloc(1, sizeof *r);
    r->gz = g;
    return r;
}

static void rd_close(kso_reader *r)
{
    if (!r) return;
    if (r->gz) gzclose(r->gz);
    free(r->seq.p);
    free(r->qual.p);
    free(r);
}

/* ------------------------------------------------------------------------------------ */
/* per-window primitives                                                                 */

static void upcase(char *s)                    /* src/BIO_sequence.c:228-234 */
{
    size_t i, n = strlen(s);
    for (i = 0; i < n; i++) s[i] = (char)toupper((unsigned char)s[i]);
}

static int has_enn(const char *s)              /* src/genome_compare.c:443-451 */
{
    for (; *s; s++) if (*s == 'N') return 1;
    return 0;
}

/* sign of (window - reverse complement) in signed-char order: src/genome_compare.c:1122-1141 */
static int window_vs_rc(const char *w, int k)
{
    int i;
    for (i = 0; i < k; i++) {
        signed char f = (signed char)w[i];
        signed char r = g_comp[(unsigned char)w[k - 1 - i]];
        if (f > r) return 1;
        if (r > f) return -1;
    }
    return 0;
}

/* canonical orientation = the larger of the two, forward on ties: src/genome_compare.c:1100-1120 */
static const char *orient(const char *w, char *rcbuf, int k)
{
    int i;
    if (window_vs_rc(w, k) >= 0) return w;
    rcbuf[k] = '\0';
    for (i = 0; i < k; i++) rcbuf[k - 1 - i] = (char)g_comp[(unsigned char)w[i]];
    return rcbuf;
}

/* a1 inner loop: src/genome_compare.c:204-230 */
static void count_record(kso_table *t, char *s, size_t l, int k, int col, char *rcbuf)
{
    size_t i;
    int rec_has_n;
    if (l < (size_t)k) return;
    upcase(s);
    rec_has_n = has_enn(s);
    for (i = 0; i + (size_t)k <= l; i++) {
        char *w = s + i, keep = w[k];
        const char *o;
        w[k] = '\0';
        o = orient(w, rcbuf, k);
        if (!rec_has_n || !has_enn(o)) {
            unsigned *vec = table_find(t, o);
            if (vec) vec[col] += 1;
        }
        w[k] = keep;
    }
}

/* a3 inner loop: src/genome_compare.c:995-1024 */
static int build_record(kso_table *t, char *s, size_t l, int k, unsigned default_val,
                        unsigned incr, int idx, int ncols, int short_policy, char *rcbuf)
{
    size_t i;
    upcase(s);
    if (l + 1 < (size_t)k) {                    /* l - k + 1 underflows in the reference */
        if (short_policy) return KSO_OK;
        return KSO_E_SHORT_CONTIG;
    }
    for (i = 0; i + (size_t)k <= l; i++) {
        char *w = s + i, keep = w[k];
        const char *o;
        w[k] = '\0';
        o = orient(w, rcbuf, k);
        if (!has_enn(o)) {
            unsigned *vec = table_find(t, o);
            if (!vec) {
                vec = (unsigned *)calloc((size_t)ncols, sizeof *vec);
                vec[idx] = default_val;
                table_place(t, strdup(o), vec);
            } else {
                vec[idx] += incr;
            }
        }
        w[k] = keep;
    }
    return KSO_OK;
}

/* ------------------------------------------------------------------------------------ */

int kso_build_from_file(kso_table *t, const char *path, int k, unsigned default_val,
                        unsigned incr, int idx, int ncols, int short_policy)
{
    kso_reader *r = rd_open_file(path);
    char *rcbuf;
    long l;
    int rc = KSO_OK;
    if (!r) return KSO_E_OPEN;
    t->ncols = ncols;
    rcbuf = (char *)malloc((size_t)k + 1);
    while ((l = rd_record(r)) >= 0) {
        rc = build_record(t, r->seq.p, (size_t)l, k, default_val, incr, idx, ncols,
                          short_policy, rcbuf);
        if (rc != KSO_OK) break;
    }
    free(rcbuf);
    rd_close(r);
    return rc;
}

int kso_scan_file(kso_table *t, const char *path, int k, int col, uint64_t *bases_seen)
{
    kso_reader *r = rd_open_file(path);
    char *rcbuf;
    long l;
    if (!r) return KSO_E_OPEN;
    rcbuf = (char *)malloc((size_t)k + 1);
    while ((l = rd_record(r)) >= 0) {
        if (bases_seen) *bases_seen += (uint64_t)l;
        count_record(t, r->seq.p, (size_t)l, k, col, rcbuf);
    }
    free(rcbuf);
    rd_close(r);
    return KSO_OK;
}

char *kso_decode_file(const char *path, size_t *out_len, long *nrecords, int *status)
{
    kso_reader *r = rd_open_file(path);
    kso_buf out = { NULL, 0, 0 };
    long l, n = 0;
    if (!r) return NULL;
    buf_reserve(&out, 0);
    while ((l = rd_record(r)) >= 0) {
        buf_reserve(&out, (size_t)l + 1);
        memcpy(out.p + out.len, r->seq.p, (size_t)l);
        out.len += (size_t)l;
        out.p[out.len++] = '\n';
        n++;
    }
    if (status) *status = (int)l;
    if (nrecords) *nrecords = n;
    if (out_len) *out_len = out.len;
    rd_close(r);
    return out.p;
}

void kso_free(void *p) { free(p); }

/* a2: src/genome_compare.c:115-146 (skip variant) and :149-177 */
int kso_scan_list(kso_table *t, const char *list_path, const char *skip, int k, int col,
                  FILE *progress, FILE *err, uint64_t *bases_seen)
{
    FILE *fp = fopen(list_path, "r");
    char *line = NULL, *nl;
    size_t cap = 0;
    if (!fp) {
        if (err) fprintf(err, "could not read file %s in GEN_all_kmer_counts()\n", list_path);
        return KSO_E_OPEN;
    }
    while (getline(&line, &cap, fp) != -1) {
        if ((nl = strchr(line, '\n')) != NULL) *nl = '\0';
        if (progress) {
            time_t now = time(NULL);
            fprintf(progress, "%s\t%s", line, asctime(localtime(&now)));
        }
        if (skip && strcmp(skip, line) == 0) {
            if (err) fprintf(err, "skipping %s (identical match)\n", line);
            continue;
        }
        if (kso_scan_file(t, line, k, col, bases_seen) != KSO_OK) {
            if (err) fprintf(err, "could not read file %s in GEN_calculate_kmer_count()\n", line);
            free(line);
            fclose(fp);
            return KSO_E_OPEN;
        }
    }
    free(line);
    fclose(fp);
    return KSO_OK;
}

/* in-memory variants: each '\n'-separated line is one already-decoded record */
void kso_scan_stream(kso_table *t, const char *stream, size_t len, int k, int col)
{
    char *rcbuf = (char *)malloc((size_t)k + 1);
    kso_buf rec = { NULL, 0, 0 };
    size_t pos = 0;
    while (pos <= len) {
        const char *nl = (pos < len) ? (const char *)memchr(stream + pos, '\n', len - pos) : NULL;
        size_t n = nl ? (size_t)(nl - (stream + pos)) : len - pos;
        if (!nl && n == 0 && pos > 0) break;            /* nothing after the final separator */
        rec.len = 0;
        buf_reserve(&rec, n);
        memcpy(rec.p, stream + pos, n);
        rec.p[n] = '\0';
        count_record(t, rec.p, n, k, col, rcbuf);
        if (!nl) break;
        pos += n + 1;
    }
    free(rec.p);
    free(rcbuf);
}

int kso_build_from_stream(kso_table *t, const char *stream, size_t len, int k,
                          unsigned default_val, unsigned incr, int idx, int ncols,
                          int short_policy)
{
    char *rcbuf = (char *)malloc((size_t)k + 1);
    kso_buf rec = { NULL, 0, 0 };
    size_t pos = 0;
    int rc = KSO_OK;
    t->ncols = ncols;
    while (pos <= len) {
        const char *nl = (pos < len) ? (const char *)memchr(stream + pos, '\n', len - pos) : NULL;
        size_t n = nl ? (size_t)(nl - (stream + pos)) : len - pos;
        if (!nl && n == 0 && pos > 0) break;            /* nothing after the final separator */
        rec.len = 0;
        buf_reserve(&rec, n);
        memcpy(rec.p, stream + pos, n);
        rec.p[n] = '\0';
        rc = build_record(t, rec.p, n, k, default_val, incr, idx, ncols, short_policy, rcbuf);
        if (rc != KSO_OK || !nl) break;
        pos += n + 1;
    }
    free(rec.p);
    free(rcbuf);
    return rc;
}

void kso_table_rows(const kso_table *t, int k, char *keys_out, unsigned *counts_out)
{
    unsigned i, row = 0;
    for (i = 0; i < t->M; i++) {
        if (!t->slot[i].vec) continue;
        if (keys_out) {
            strncpy(keys_out + (size_t)row * ((size_t)k + 1), t->slot[i].key, (size_t)k);
            keys_out[(size_t)row * ((size_t)k + 1) + (size_t)k] = '\0';
        }
        if (counts_out)
            memcpy(counts_out + (size_t)row * (size_t)t->ncols, t->slot[i].vec,
                   (size_t)t->ncols * sizeof(unsigned));
        row++;
    }
}

/* a9: src/kmer_scrub_count.c:134-156.  Header always names five columns; "%d" of unsigned. */
void kso_print(const kso_table *t, FILE *out, int with_drug_column)
{
    unsigned i;
    fputs("#kmer\treference_count\tpangenome_count\tmetagenome_count\tdrug_count\n", out);
    for (i = 0; i < t->M; i++) {
        const unsigned *v = t->slot[i].vec;
        if (!v) continue;
        if (with_drug_column)
            fprintf(out, "%s\t%d\t%d\t%d\t%d\n", t->slot[i].key, (int)v[0], (int)v[1], (int)v[2], (int)v[3]);
        else
            fprintf(out, "%s\t%d\t%d\t%d\n", t->slot[i].key, (int)v[0], (int)v[1], (int)v[2]);
    }
}

/* whole program: src/kmer_scrub_count.c:29-131 */
static void say_usage(FILE *err)
{
    fputs("Usage: kmer_scrub_count -r <reference genome>  -A <file with multiple genome filenames> "
          "-B <file with multiple metagenome filenames> -C <(optional) file with multiple genome "
          "filenames of drug strains> -p [progress output file, optional]\n", err);
}

int kso_main(int argc, char **argv, FILE *out, FILE *err)
{
    const int k = 31;
    const char *A = NULL, *B = NULL, *C = NULL, *R = NULL, *P = NULL;
    FILE *progress = NULL;
    kso_table *t;
    int c, rc;

    optind = 1;
    while ((c = getopt(argc, argv, "A:B:C:r:p:Hhud")) != -1) {
        switch (c) {
        case 'A': A = optarg; break;
        case 'B': B = optarg; break;
        case 'C': C = optarg; break;
        case 'r': R = optarg; break;
        case 'p': P = optarg; break;
        case 'd': break;
        default:  say_usage(err); break;     /* -h -u -H and unknown flags: print, carry on */
        }
    }
    if (!R || !A || !B) { say_usage(err); return 1; }

    if (P) {
        progress = fopen(P, "w");
        if (!progress) { fprintf(err, "could not open progress file %s\n", P); return 1; }
        fputs("adding kmer counts for:\n", progress);
    }
    t = kso_table_new(KSO_DEFAULT_CAPACITY);
    rc = kso_build_from_file(t, R, k, 1, 1, 0, 4, 0);
    if (rc == KSO_E_OPEN) {
        fprintf(err, "could not read file %s GEN_hash_sequences_set_count_vec()\n", R);
        return 1;
    }
    if (rc == KSO_E_SHORT_CONTIG) return 139;   /* the reference dies of SIGSEGV here */
    if (kso_scan_list(t, A, NULL, k, 1, progress, err, NULL) != KSO_OK) return 1;
    if (kso_scan_list(t, B, NULL, k, 2, progress, err, NULL) != KSO_OK) return 1;
    if (C && kso_scan_list(t, C, R, k, 3, progress, err, NULL) != KSO_OK) return 1;
    kso_print(t, out, C != NULL);
    kso_table_free(t);
    if (progress) fclose(progress);
    return 0;
}

#ifdef KSO_MAIN
int main(int argc, char **argv)
{
    static char obuf[1 << 20];
    setvbuf(stdout, obuf, _IOFBF, sizeof obuf);
    return kso_main(argc, argv, stdout, stderr);
}
#endif
