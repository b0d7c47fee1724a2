/* sk_host_sd.c -- host layer of the strain_detect program (reference src/strain_detect.c).
 *
 * What runs where:
 *   device  every k-mer lookup: the per-read tallies (all hits / informative hits) and the log of
 *           informative hits come from sk_tally_batch (sk_scan_main in TALLY mode); the informative
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
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include "../../include/strainer_kmer.h"
#include "sk_common.h"
#include "sk_parser.h"

enum { SD_TYPE = 0, SD_BACKGROUND = 5, SD_INFORMATIVE = 2, SD_PLAIN = 1, SD_NCOLS = 6 };
enum { SD_SE = 0, SD_PE = 1, SD_PEI = 2, SD_UNKNOWN = -1 };
#define SD_BATCH_BYTES (32u << 20)

/* ---------------------------------------------------------------------------------------------
 * one metagenome file, decoded and tallied on the device
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t len;           /* sequence length as the reference sees it                          */
    uint32_t hits, inf;     /* tallies (meaningful when len >= k)                                */
    uint64_t hit_begin;     /* its informative hits, in window order: hit_rows[hit_begin .. +inf) */
} sd_rec;

typedef struct {
    sd_rec   *rec; size_t n, cap;
    uint32_t *hit_rows; size_t nh, hcap;
    int       end_kind; size_t end_len;
    /* batch under construction */
    sk_ctx   *ctx;
    uint8_t  *buf; uint64_t blen;
    uint32_t *bstart; uint64_t *bindex; uint32_t bn, bcap;
    int       rc;
} sd_file;

static int hit_cmp(const void *a, const void *b)
{
    const sk_hit *x = (const sk_hit *)a, *y = (const sk_hit *)b;
    return x->pos < y->pos ? -1 : x->pos > y->pos;
}

static int sd_flush(sd_file *f)
{
    uint32_t *tally;
    sk_hit *hits = NULL;
    uint64_t cap = 1u << 20, nh = 0;
    uint32_t r;
    size_t h = 0;
    if (f->bn == 0 || f->rc) { f->blen = 0; f->bn = 0; return f->rc; }
    tally = (uint32_t *)malloc((size_t)f->bn * 8);
    for (;;) {
        hits = (sk_hit *)realloc(hits, (size_t)cap * sizeof *hits);
        f->rc = sk_tally_batch(f->ctx, f->buf, f->blen, f->bstart, f->bn, SD_TYPE, SD_INFORMATIVE, tally, hits, cap, &nh);
        if (f->rc || nh <= cap) break;
        cap = nh;                                        /* the log overflowed: once more with room */
    }
    if (!f->rc) {
        qsort(hits, (size_t)nh, sizeof *hits, hit_cmp);
        if (f->nh + nh > f->hcap) {
            f->hcap = (f->nh + nh) * 2 + 1024;
            f->hit_rows = (uint32_t *)realloc(f->hit_rows, f->hcap * sizeof(uint32_t));
        }
        for (r = 0; r < f->bn; r++) {
            sd_rec *rec = &f->rec[f->bindex[r]];
            const uint32_t end = r + 1 < f->bn ? f->bstart[r + 1] : 0xFFFFFFFFu;
            if (rec->hits == 0 && rec->inf == 0) rec->hit_begin = f->nh;    /* first piece of this record */
            rec->hits += tally[2 * r];
            rec->inf += tally[2 * r + 1];
            while (h < nh && hits[h].pos < end) f->hit_rows[f->nh++] = hits[h++].row;
        }
    }
    free(hits);
    free(tally);
    f->blen = 0;
    f->bn = 0;
    return f->rc;
}

static void sd_batch_add(sd_file *f, uint64_t index, const char *seq, size_t len)
{
    if (f->bn == f->bcap) {
        f->bcap = f->bcap ? f->bcap * 2 : 1 << 16;
        f->bstart = (uint32_t *)realloc(f->bstart, (size_t)f->bcap * sizeof(uint32_t));
        f->bindex = (uint64_t *)realloc(f->bindex, (size_t)f->bcap * sizeof(uint64_t));
    }
    f->bstart[f->bn] = (uint32_t)f->blen;
    f->bindex[f->bn++] = index;
    memcpy(f->buf + f->blen, seq, len);
    f->blen += len;
    f->buf[f->blen++] = '\n';
}

static int sd_on_record(void *user, char *seq, size_t len)
{
    sd_file *f = (sd_file *)user;
    size_t off = 0;
    if (f->n == f->cap) {
        f->cap = f->cap ? f->cap * 2 : 1 << 16;
        f->rec = (sd_rec *)realloc(f->rec, f->cap * sizeof(sd_rec));
    }
    memset(&f->rec[f->n], 0, sizeof(sd_rec));
    f->rec[f->n].len = len;
    f->rec[f->n].hit_begin = f->nh;
    if (len >= SK_K) {
        while (off < len) {                              /* pieces of at most one batch, k-1 overlap */
            size_t room, take;
            if (SD_BATCH_BYTES - f->blen < 4096 && sd_flush(f)) return f->rc;
            room = SD_BATCH_BYTES - f->blen - 1;
            take = len - off < room ? len - off : room;
            sd_batch_add(f, f->n, seq + off, take);
            off += take;
            if (off < len) { off -= SK_OVERLAP; if (sd_flush(f)) return f->rc; }
        }
    }
    f->n++;
    return 0;
}

static void sd_file_free(sd_file *f)
{
    free(f->rec); free(f->hit_rows); free(f->buf); free(f->bstart); free(f->bindex);
    memset(f, 0, sizeof *f);
}

/* decode + tally a whole file; SK_E_OPEN if it cannot be opened */
static int sd_file_load(sk_ctx *ctx, const char *path, sd_file *f)
{
    enum { BLK = 1 << 20 };
    gzFile g;
    unsigned char *blk;
    parser ps;
    int got;
    memset(f, 0, sizeof *f);
    g = gzopen(path, "r");
    if (!g) return SK_E_OPEN;
    gzbuffer(g, 1 << 18);
    f->ctx = ctx;
    f->buf = (uint8_t *)malloc(SD_BATCH_BYTES);
    blk = (unsigned char *)malloc(BLK);
    parser_init(&ps, sd_on_record, f);
    while (ps.state != P_STOP && (got = gzread(g, blk, BLK)) > 0) parser_feed(&ps, blk, (size_t)got);
    parser_eof(&ps);
    f->end_kind = ps.end_kind;
    f->end_len = ps.end_len;
    if (!f->rc) sd_flush(f);
    parser_free(&ps);
    free(blk);
    gzclose(g);
    return f->rc;
}

/* ---------------------------------------------------------------------------------------------
 * program state
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    sk_ctx     *ctx;
    skh_keyset  ks;
    uint32_t   *type;          /* host copy of the type column */
    FILE       *out, *err;
} sd_prog;

static void emit_hits(sd_prog *p, gzFile gz, const sd_file *f, const sd_rec *rec, const char *name,
                      int h1, int i1, int h2, int i2)
{
    uint32_t j;
    char key[32];
    for (j = 0; j < rec->inf; j++) {
        skh_keyset_key(&p->ks, f->hit_rows[rec->hit_begin + j], key);
        gzprintf(gz, "%s\t%d\t%d\t%d\t%d\t%s\n", name, h1, i1, h2, i2, key);
    }
}

/* one metagenome (pair): the reference's read loop, src/strain_detect.c:443-626, replayed over the
 * device tallies.  A read shorter than k refreshes nothing, so it re-uses (and may re-emit) the
 * previous read's tallies and sequence -- as the reference does. */
static int sd_quantify(sd_prog *p, gzFile gz, const char *f1, const char *f2, int mode,
                       unsigned genome_kmers, unsigned genome_inf)
{
    sd_file A, B, *pb = NULL;
    size_t ia = 0, ib = 0, *cursor_b;
    const sd_rec *copy = NULL, *cur_b = NULL;
    int h1 = 0, i1 = 0, h2 = 0, i2 = 0, rc;
    unsigned long long evaluated = 0, reads = 0;

    rc = sd_file_load(p->ctx, f1, &A);
    if (rc == SK_E_OPEN) { fprintf(p->err, "could not read file (read1) %s in quantify_hits_PE() (error: %s)\n", f1, strerror(errno)); return 1; }
    if (rc) { fprintf(p->err, "strain_detect: device error on %s: %s (%s)\n", f1, sk_strerror(rc), sk_last_error(p->ctx)); return 1; }
    memset(&B, 0, sizeof B);
    if (mode == SD_PE) {
        rc = sd_file_load(p->ctx, f2, &B);
        if (rc == SK_E_OPEN) { fprintf(p->err, "could not read file (read2) is_PE %s in quantify_hits_PE() (error: (null))\n", f2); sd_file_free(&A); return 1; }
        if (rc) { fprintf(p->err, "strain_detect: device error on %s: %s (%s)\n", f2, sk_strerror(rc), sk_last_error(p->ctx)); sd_file_free(&A); return 1; }
        pb = &B;
        cursor_b = &ib;
    } else {
        pb = &A;                                         /* PEI: the mate is the next record of the same file */
        cursor_b = &ia;
    }

    while (ia < A.n) {
        const sd_rec *a = &A.rec[ia++];
        if (a->len >= SK_K) {
            reads++;
            h1 = (int)a->hits;
            i1 = (int)a->inf;
            copy = a;
            evaluated += a->len - (SK_K - 1);
        }
        if (mode != SD_SE) {
            uint64_t len2;
            int failed = 0;
            if (*cursor_b < pb->n) { cur_b = &pb->rec[(*cursor_b)++]; len2 = cur_b->len; }
            else { cur_b = NULL; failed = 1; len2 = pb->end_kind == SKP_END_RESET ? 0 : pb->end_len; }
            if (len2 >= SK_K) {
                h2 = i2 = 0;
                if (failed) {
                    fprintf(p->err, "reached end of PE2 (%s) before end of PE1 (%s), check that file names are correct\n",
                            f2 ? f2 : "(null)", f1);
                    sd_file_free(&A); sd_file_free(&B);
                    return 1;
                }
                h2 = (int)cur_b->hits;
                i2 = (int)cur_b->inf;
                evaluated += cur_b->len - (SK_K - 1);
            }
        }
        if (h1 + h2 >= 1 && i1 + i2 >= 1) {
            if (copy) emit_hits(p, gz, &A, copy, f1, h1, i1, h2, i2);
            if (mode != SD_SE && cur_b && cur_b->len >= SK_K) emit_hits(p, gz, pb, cur_b, f1, h1, i1, h2, i2);
        }
    }
    gzprintf(gz, "#%s\ttotal_kmer_evaluated\t%lld\n", f1, evaluated);
    gzprintf(gz, "#%s\ttotal_reads_evaluated\t%lld\n", f1, reads);
    gzprintf(gz, "#%s\ttotal_genome_kmers\t%lld\n", f1, (long long)genome_kmers);
    gzprintf(gz, "#%s\ttotal_genome_informative_kmers\t%lld\n", f1, (long long)genome_inf);
    sd_file_free(&A);
    sd_file_free(&B);
    return 0;
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
    for (i = 0; i < np; i++) free(piece[i]);
    free(piece); free(kind); free(slot); free(stream); free(start); free(tally); free(hits);
    if (!rc && p->ks.nrows) {
        rc = sk_counts_set(p->ctx, SD_TYPE, p->type);
        if (rc) fprintf(p->err, "strain_detect: device error: %s (%s)\n", sk_strerror(rc), sk_last_error(p->ctx));
    }
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

int skh_strain_detect_main(int argc, char **argv, FILE *out, FILE *err)
{
    const char *a = NULL, *r = NULL, *b = NULL, *b2 = NULL, *B = NULL, *tt = NULL, *g = NULL, *o = NULL, *env;
    int c, mode = SD_SE, rc, status = 1, device = 0;
    unsigned n_inform = 0, i, genome_inf = 0;
    sd_prog p;
    gzFile gz = NULL;

    memset(&p, 0, sizeof p);
    p.out = out;
    p.err = err;
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
        default:  usage(err); break;
        }
    }
    if (!a || !o || !r) { usage(err); return 1; }
    if (!b && !B) { usage(err); return 1; }
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

    rc = skh_keyset_from_file(&p.ks, r, SK_REF_TABLE_SLOTS, SD_PLAIN, 0);
    if (rc == SK_E_OPEN) { fprintf(err, "could not read file %s GEN_hash_sequences_set_count_vec()\n", r); goto done; }
    if (rc != SK_OK) { fprintf(err, "strain_detect: %s\n", sk_strerror(rc)); goto done; }
    if (p.ks.short_records)
        fprintf(err, "strain_detect: skipped %llu reference record(s) shorter than %d bases "
                     "(the original program crashes on these)\n", (unsigned long long)p.ks.short_records, SK_K - 1);
    rc = sk_ctx_create(&p.ctx, device);
    if (rc != SK_OK) { fprintf(err, "strain_detect: cannot use HIP device %d: %s\n", device, sk_strerror(rc)); goto done; }
    rc = skh_keyset_load(p.ctx, &p.ks, SD_NCOLS);
    if (rc != SK_OK) { fprintf(err, "strain_detect: table load failed: %s (%s)\n", sk_strerror(rc), sk_last_error(p.ctx)); goto done; }
    p.type = (uint32_t *)malloc((size_t)(p.ks.nrows ? p.ks.nrows : 1) * sizeof(uint32_t));
    for (i = 0; i < p.ks.nrows; i++) p.type[i] = SD_PLAIN;

    if (sd_flag_informative(&p, a, &n_inform)) goto done;
    if (g && sd_background_filter(&p, g, 0.5, n_inform)) goto done;
    for (i = 0; i < p.ks.nrows; i++) if (p.type[i] == SD_INFORMATIVE) genome_inf++;

    gz = gzopen(o, "wb9");
    if (!gz) { fprintf(err, "could not open *gzout file outfile %s in quantify_hits_all_files()\n", o); goto done; }
    if (B) {
        FILE *fp = fopen(B, "r");
        char *line = NULL, *nl, *tok, *f1, *f2;
        size_t cap = 0;
        int bad = 0;
        if (!fp) { fprintf(err, "could not read file file_of_filenames %s in quantify_hits_all_files()\n", B); goto done; }
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
                bad = sd_quantify(&p, gz, f1, f2, m, p.ks.nrows, genome_inf);
            } else bad = sd_quantify(&p, gz, f1, NULL, m, p.ks.nrows, genome_inf);
        }
        free(line);
        fclose(fp);
        if (bad) goto done;
    } else if (sd_quantify(&p, gz, b, b2, mode, p.ks.nrows, genome_inf)) goto done;
    status = 0;
done:
    if (gz) gzclose(gz);
    if (p.ctx) sk_ctx_destroy(p.ctx);
    skh_keyset_free(&p.ks);
    free(p.type);
    return status;
}
