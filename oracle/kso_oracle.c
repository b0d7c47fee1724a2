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
#include <errno.h>
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

/* ==================================================================================== */
/* strain_detect restatement: src/strain_detect.c (whole file)                          */

enum { SD_TYPE = 0, SD_BACKGROUND = 5, SD_INFORMATIVE = 2, SD_PLAIN = 1, SD_NCOLS = 6 };
enum { SD_SE = 0, SD_PE = 1, SD_PEI = 2, SD_UNKNOWN = -1 };

static int sd_file_type(const char *s)                    /* src/strain_detect.c:728-747 */
{
    if (!strcmp(s, "SE") || !strcmp(s, "se")) return SD_SE;
    if (!strcmp(s, "PE") || !strcmp(s, "pe")) return SD_PE;
    if (!strcmp(s, "PEI") || !strcmp(s, "pei") || !strcmp(s, "IPE") || !strcmp(s, "ipe")) return SD_PEI;
    return SD_UNKNOWN;
}

static void sd_usage(FILE *err)                           /* src/strain_detect.c:750-764 */
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

/* flag the listed k-mers as informative: src/strain_detect.c:668-726 (gzgets in 99-byte pieces,
 * no case folding, '#' lines skipped, every piece whose length is not k reported on stdout) */
static int sd_flag_informative(kso_table *t, const char *path, int k, FILE *out, FILE *err, unsigned *n_out)
{
    gzFile g = gzopen(path, "r");
    char line[100], *rcbuf = (char *)malloc(101), *nl;
    unsigned n = 0;
    if (!g) { fprintf(err, "could not read file %s in hash_scrubbed_kmers()\n", path); free(rcbuf); return 1; }
    while (gzgets(g, line, 100)) {
        if (line[0] == '#') continue;
        if ((nl = strchr(line, '\n')) != NULL) *nl = '\0';
        if ((int)strlen(line) == k) {
            unsigned *vec = table_find(t, orient(line, rcbuf, k));
            if (vec) { vec[SD_TYPE] = SD_INFORMATIVE; n++; }
            else fprintf(out, "error could not find informative kmer %s in the total kmer list\n", line);
        } else {
            fprintf(out, "error string length in the scrubbed kmer file (%s) must be the same size as the kmer length "
                         "(scrubbed kmer, scrubbed kmer len, seed len): %s, %d, %d\n", path, line, (int)strlen(line), k);
        }
    }
    free(rcbuf);
    gzclose(g);
    *n_out = n;
    return 0;
}

static int sd_cmp_desc(const void *a, const void *b) { return (int)(*(const unsigned *)b - *(const unsigned *)a); }

static int sd_removed(unsigned threshold, const unsigned *c, unsigned n)
{
    unsigned i; int r = 0;
    for (i = 0; i < n; i++) if (c[i] >= threshold) r++;
    return r;
}

/* -g: src/strain_detect.c:160-240 */
static int sd_background_filter(kso_table *t, const char *list, double fraction, unsigned n_inform, int k,
                                FILE *out, FILE *err)
{
    unsigned keep = (unsigned)(int)(n_inform * fraction), *c = (unsigned *)calloc(n_inform ? n_inform : 1, sizeof *c);
    unsigned i, n = 0, threshold = 1;
    int demoted = 0;
    fprintf(out, "#removing %f proportion of %s kmers; informative %d keep at least %d\n", fraction, list, n_inform, keep);
    if (kso_scan_list(t, list, NULL, k, SD_BACKGROUND, NULL, err, NULL) != KSO_OK) { free(c); return 1; }
    for (i = 0; i < t->M; i++) {
        const unsigned *v = t->slot[i].vec;
        if (!v || v[SD_TYPE] != SD_INFORMATIVE) continue;
        if (n >= n_inform) { fputs("Error: too many background kmers\n", err); free(c); return 1; }
        c[n++] = v[SD_BACKGROUND];
    }
    qsort(c, n_inform, sizeof *c, sd_cmp_desc);
    if (keep >= 1 && c[keep - 1] > threshold) threshold = c[keep - 1];
    while ((unsigned)sd_removed(threshold, c, n_inform) > keep) threshold++;
    for (i = 0; i < t->M; i++) {
        unsigned *v = t->slot[i].vec;
        if (v && v[SD_TYPE] == SD_INFORMATIVE && v[SD_BACKGROUND] >= threshold) { v[SD_TYPE] = SD_PLAIN; demoted++; }
    }
    fprintf(out, "#final_threshold %d removes %d background kmers %d removed\n", threshold, sd_removed(threshold, c, n_inform), demoted);
    free(c);
    return 0;
}

/* whole-read reverse complement through the complement map: src/BIO_sequence.c:244-253 */
static void sd_revcomp(char *dst, const char *src, size_t l)
{
    size_t i;
    for (i = 0; i < l; i++) dst[l - 1 - i] = (char)g_comp[(unsigned char)src[i]];
    dst[l] = '\0';
}

/* tally pass over one read: src/strain_detect.c:455-492 / 509-540.  Orientation here is
 * strcmp(window, rc_window) > 0 ? window : rc_window on the pre-reversed read. */
static void sd_tally(kso_table *t, char *s, size_t l, int k, char *rc, int *hits, int *inf, unsigned long long *evaluated)
{
    size_t i;
    int has_n;
    char *w = s, *r;
    sd_revcomp(rc, s, l);
    r = rc + (l - (size_t)k);
    has_n = has_enn(s);
    for (i = 0; i + (size_t)k <= l; i++) {
        char keep = w[k];
        const char *o;
        w[k] = '\0';
        r[k] = '\0';
        o = strcmp(w, r) > 0 ? w : r;
        if (!has_n || !has_enn(o)) {
            const unsigned *vec = table_find(t, o);
            if (vec) { (*hits)++; if (vec[SD_TYPE] == SD_INFORMATIVE) (*inf)++; }
        }
        w[k] = keep;
        w++;
        r--;
        (*evaluated)++;
    }
}

/* emission pass over one read: src/strain_detect.c:552-585 / 590-621 */
static void sd_emit(kso_table *t, char *s, size_t l, int k, char *rcbuf, gzFile gz, const char *name,
                    int h1, int i1, int h2, int i2)
{
    size_t i;
    if (l < (size_t)k) return;
    for (i = 0; i + (size_t)k <= l; i++) {
        char *w = s + i, keep = w[k];
        const char *o;
        w[k] = '\0';
        o = orient(w, rcbuf, k);
        if (!has_enn(o)) {
            const unsigned *vec = table_find(t, o);
            if (vec && vec[SD_TYPE] == SD_INFORMATIVE) gzprintf(gz, "%s\t%d\t%d\t%d\t%d\t%s\n", name, h1, i1, h2, i2, o);
        }
        w[k] = keep;
    }
}

/* one metagenome (pair): src/strain_detect.c:387-663, including its carry-over behaviour: the
 * per-read tallies and the PE1 copy are only refreshed for reads of at least k bases, so a short
 * read inherits the previous read's tallies and sequence. */
static int sd_quantify(kso_table *t, const char *f1, const char *f2, int k, gzFile gz, int mode,
                       unsigned genome_kmers, unsigned genome_inf, FILE *err)
{
    kso_reader *r1 = rd_open_file(f1), *r2 = NULL;
    kso_buf copy = { NULL, 0, 0 }, rc = { NULL, 0, 0 };
    char *rcbuf = (char *)malloc((size_t)k + 1);
    int h1 = 0, i1 = 0, h2 = 0, i2 = 0;
    size_t copy_len = 0;
    unsigned long long evaluated = 0, reads = 0;
    long l, l2;
    if (!r1) { fprintf(err, "could not read file (read1) %s in quantify_hits_PE() (error: %s)\n", f1, strerror(errno)); return 1; }
    if (mode == SD_PE) {
        r2 = rd_open_file(f2);
        if (!r2) { fprintf(err, "could not read file (read2) is_PE %s in quantify_hits_PE() (error: (null))\n", f2); return 1; }
    } else if (mode == SD_PEI) r2 = r1;
    while ((l = rd_record(r1)) >= 0) {
        if (r1->seq.len >= (size_t)k) {
            reads++;
            h1 = i1 = 0;
            copy_len = r1->seq.len;
            upcase(r1->seq.p);
            copy.len = 0; buf_reserve(&copy, copy_len); memcpy(copy.p, r1->seq.p, copy_len + 1);
            rc.len = 0; buf_reserve(&rc, copy_len + 1);
            sd_tally(t, r1->seq.p, copy_len, k, rc.p, &h1, &i1, &evaluated);
        }
        if (mode != SD_SE) {
            l2 = rd_record(r2);
            if (r2->seq.len >= (size_t)k) {
                h2 = i2 = 0;
                if (l2 < 0) {
                    fprintf(err, "reached end of PE2 (%s) before end of PE1 (%s), check that file names are correct\n",
                            f2 ? f2 : "(null)", f1);
                    return 1;
                }
                upcase(r2->seq.p);
                rc.len = 0; buf_reserve(&rc, r2->seq.len + 1);
                sd_tally(t, r2->seq.p, r2->seq.len, k, rc.p, &h2, &i2, &evaluated);
            }
        }
        if (h1 + h2 >= 1 && i1 + i2 >= 1) {
            if (copy.p) sd_emit(t, copy.p, copy_len, k, rcbuf, gz, f1, h1, i1, h2, i2);
            if (mode != SD_SE) sd_emit(t, r2->seq.p, r2->seq.len, k, rcbuf, gz, f1, h1, i1, h2, i2);
        }
    }
    gzprintf(gz, "#%s\ttotal_kmer_evaluated\t%lld\n", f1, evaluated);
    gzprintf(gz, "#%s\ttotal_reads_evaluated\t%lld\n", f1, reads);
    gzprintf(gz, "#%s\ttotal_genome_kmers\t%lld\n", f1, (long long)genome_kmers);
    gzprintf(gz, "#%s\ttotal_genome_informative_kmers\t%lld\n", f1, (long long)genome_inf);
    free(copy.p); free(rc.p); free(rcbuf);
    if (r2 && r2 != r1) rd_close(r2);
    rd_close(r1);
    return 0;
}

int ksd_main(int argc, char **argv, FILE *out, FILE *err)
{
    const int k = 31;
    const char *a = NULL, *r = NULL, *b = NULL, *b2 = NULL, *B = NULL, *tt = NULL, *g = NULL, *o = NULL;
    int c, mode = SD_SE, rc;
    unsigned n_inform = 0, i, genome_inf = 0;
    kso_table *t;
    gzFile gz;

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
        default:  sd_usage(err); break;
        }
    }
    if (!a || !o || !r) { sd_usage(err); return 1; }
    if (!b && !B) { sd_usage(err); return 1; }
    if (tt) {
        mode = sd_file_type(tt);
        if (mode == SD_UNKNOWN) { fputs("unknown filetype specification. allowed are SE, PE, PEI\n\n", out); sd_usage(err); return 1; }
    }
    if (b && mode == SD_PE && !b2) {
        fputs("commandline PE mapping requires two files (-b [file1] and -c [file2])\n\n", out); sd_usage(err); return 1;
    }
    if (b && B) {
        fputs("cannot have -B flag and -b flag\nEither have a file with metagenomics files to be detect the strain in or "
              "specify one metagenomic file to detect the strain in\n", out);
        sd_usage(err); return 1;
    }
    t = kso_table_new(KSO_DEFAULT_CAPACITY);
    rc = kso_build_from_file(t, r, k, SD_PLAIN, 0, 0, SD_NCOLS, 0);
    if (rc == KSO_E_OPEN) { fprintf(err, "could not read file %s GEN_hash_sequences_set_count_vec()\n", r); return 1; }
    if (rc == KSO_E_SHORT_CONTIG) return 139;
    if (sd_flag_informative(t, a, k, out, err, &n_inform)) return 1;
    if (g && sd_background_filter(t, g, 0.5, n_inform, k, out, err)) return 1;

    for (i = 0; i < t->M; i++) if (t->slot[i].vec && t->slot[i].vec[SD_TYPE] == SD_INFORMATIVE) genome_inf++;
    gz = gzopen(o, "wb9");
    if (!gz) { fprintf(err, "could not open *gzout file outfile %s in quantify_hits_all_files()\n", o); return 1; }
    if (B) {
        FILE *fp = fopen(B, "r");
        char *line = NULL, *nl, *tok, *f1, *f2;
        size_t cap = 0;
        if (!fp) { fprintf(err, "could not read file file_of_filenames %s in quantify_hits_all_files()\n", B); return 1; }
        while (getline(&line, &cap, fp) != -1) {
            int m;
            if ((nl = strchr(line, '\n')) != NULL) *nl = '\0';
            tok = strtok(line, "\t");
            if (!tok) { gzclose(gz); return 139; }          /* reference: strcmp(NULL, ...) */
            m = sd_file_type(tok);
            if (m == SD_UNKNOWN) { fprintf(out, "unknown file type skipping line (%s)\n", tok); continue; }
            f1 = strtok(NULL, "\t");
            if (!f1) { fprintf(out, "ERROR: no first file specified for %s\n", line); continue; }
            if (m == SD_PE) {
                f2 = strtok(NULL, "\t");
                if (!f2) { fprintf(out, "ERROR: no second file specified for PE: %s\n", line); continue; }
                if (sd_quantify(t, f1, f2, k, gz, m, t->N, genome_inf, err)) { gzclose(gz); return 1; }
            } else if (sd_quantify(t, f1, NULL, k, gz, m, t->N, genome_inf, err)) { gzclose(gz); return 1; }
        }
        free(line);
        fclose(fp);
    } else if (sd_quantify(t, b, b2, k, gz, mode, t->N, genome_inf, err)) { gzclose(gz); return 1; }
    gzclose(gz);
    kso_table_free(t);
    return 0;
}

#ifdef KSD_MAIN
int main(int argc, char **argv) { return ksd_main(argc, argv, stdout, stderr); }
#endif

#ifdef KSO_MAIN
int main(int argc, char **argv)
{
    static char obuf[1 << 20];
    setvbuf(stdout, obuf, _IOFBF, sizeof obuf);
    return kso_main(argc, argv, stdout, stderr);
}
#endif
