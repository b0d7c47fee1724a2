/* sk_host_cov.c -- host side of the coverage/depth table: the drop-in for reference
 * scripts/coverage_depth.py (step 4 of test/example.sh).
 *
 * The script reads strain_detect's hit list (<metagenome> <hits PE1> <informative PE1> <hits PE2>
 * <informative PE2> <k-mer>, plus four "#<metagenome> <name> <value>" trailer lines per metagenome) and
 * prints, per metagenome, how many hit lines pass the read filter, how many different k-mers they name
 * and the two ratios to the strain's informative k-mer count.  Here the lines are parsed into
 * (sample, packed k-mer) pairs and both counts are taken on the device (sk_distinct_count,
 * sk_cover.hip); this file keeps the script's dictionaries (their insertion order is the order of the
 * printed rows) and its number formatting.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "../../include/strainer_kmer.h"
#include "sk_common.h"
#include "sk_pyfmt.h"
#include "sk_internal.h"

/* ------------------------------------------------------------------ a small insertion-ordered name table */
typedef struct {
    char   **name; uint32_t n, cap;
    /* per sample, the script's dictionaries; have_* = key present */
    int64_t *kmer_eval, *read_eval, *g_total, *g_inf;
    uint8_t *have_eval, *have_reads, *have_total, *have_inf, *in_depth;
    uint32_t *slot; uint32_t nslot;
} names;

static uint64_t name_hash(const char *s, size_t len)
{
    uint64_t h = 1469598103934665603ull;
    size_t i;
    for (i = 0; i < len; i++) h = (h ^ (unsigned char)s[i]) * 1099511628211ull;
    return h;
}

static int64_t name_get(names *t, const char *s, size_t len)
{
    uint32_t i, g;
    if (!t->slot) {
        t->nslot = 256;
        if (!(t->slot = calloc(t->nslot, sizeof *t->slot))) return -1;
    }
    i = (uint32_t)name_hash(s, len) & (t->nslot - 1);
    while ((g = t->slot[i]) != 0) {
        if (strlen(t->name[g - 1]) == len && !memcmp(t->name[g - 1], s, len)) return g - 1;
        i = (i + 1) & (t->nslot - 1);
    }
    if (t->n == t->cap) {
        const uint32_t cap = t->cap ? t->cap * 2 : 64;
#define GROW(field) do { void *p_ = realloc(t->field, (size_t)cap * sizeof *t->field); if (!p_) return -1; t->field = p_; } while (0)
        GROW(name); GROW(kmer_eval); GROW(read_eval); GROW(g_total); GROW(g_inf);
        GROW(have_eval); GROW(have_reads); GROW(have_total); GROW(have_inf); GROW(in_depth);
#undef GROW
        t->cap = cap;
    }
    g = t->n++;
    if (!(t->name[g] = malloc(len + 1))) return -1;
    memcpy(t->name[g], s, len);
    t->name[g][len] = 0;
    t->kmer_eval[g] = t->read_eval[g] = t->g_total[g] = t->g_inf[g] = 0;
    t->have_eval[g] = t->have_reads[g] = t->have_total[g] = t->have_inf[g] = t->in_depth[g] = 0;
    t->slot[i] = g + 1;
    if (t->n * 2 > t->nslot) {                        /* rebuild at four times the size */
        uint32_t ns = t->nslot * 4, *s2 = calloc(ns, sizeof *s2), k;
        if (!s2) return -1;
        for (k = 0; k < t->n; k++) {
            uint32_t j = (uint32_t)name_hash(t->name[k], strlen(t->name[k])) & (ns - 1);
            while (s2[j]) j = (j + 1) & (ns - 1);
            s2[j] = k + 1;
        }
        free(t->slot);
        t->slot = s2; t->nslot = ns;
    }
    return g;
}

static void names_free(names *t)
{
    uint32_t i;
    for (i = 0; i < t->n; i++) free(t->name[i]);
    free(t->name); free(t->kmer_eval); free(t->read_eval); free(t->g_total); free(t->g_inf);
    free(t->have_eval); free(t->have_reads); free(t->have_total); free(t->have_inf); free(t->in_depth); free(t->slot);
    memset(t, 0, sizeof *t);
}

static const char *base_of(const char *s, const char *end)
{
    const char *p = end;
    while (p > s && p[-1] != '/') p--;
    return p;
}

/* ------------------------------------------------------------------ the hit list */
typedef struct {
    names     smp;
    uint64_t *key; uint32_t *sample; uint64_t n, cap;   /* hit lines that passed the read filter */
    uint32_t *depth_order; uint32_t ndepth;            /* samples in the order their first passing line came */
    uint32_t *eval_order; uint32_t neval;              /* samples in the order their total_kmer_evaluated line came */
    names     text;                                    /* general mode: the script's unique strings -> ids */
    uint32_t *gid, *gsample; uint64_t gn, gcap;        /* general mode: per passing line, in file order: joined-string number, sample */
    int       general;
    size_t    klen;                                    /* k-mer field length seen so far (0 = none yet) */
} hitlist;

enum { COV_OK = 0, COV_OPEN, COV_FIELDS, COV_INT, COV_NOMEM };

/* the script keys its uniqueness test by <sample name><k-mer text> joined without a separator (:88-93).
 * With k-mer fields of one length made of A/C/G/T -- all strain_detect ever writes -- that is the pair
 * (sample, packed k-mer), counted on the device.  Any other k-mer text (ragged lengths, other letters)
 * switches the file to "general" mode, because two different pairs may then join to the same string: the
 * joined strings are numbered here (a dictionary of texts, as the parse needs anyway) and the device credits
 * every string to the sample of the first line that shows it (sk_first_seen_count), which is what the
 * script's global "seen" dictionary does. */
static int general_count(hitlist *h, uint32_t g, const char *ktext, size_t klen)
{
    const char *sn = h->smp.name[g];
    const size_t sl = strlen(sn);
    char *u = malloc(sl + klen + 1);
    int64_t id;
    if (!u) return COV_NOMEM;
    memcpy(u, sn, sl); memcpy(u + sl, ktext, klen);
    id = name_get(&h->text, u, sl + klen);            /* number of the joined string */
    free(u);
    if (id < 0) return COV_NOMEM;
    if (h->gn == h->gcap) {
        const uint64_t cap = h->gcap ? h->gcap * 2 : 1u << 12;
        uint32_t *a = realloc(h->gid, cap * sizeof *a), *b;
        if (!a) return COV_NOMEM;
        h->gid = a;
        if (!(b = realloc(h->gsample, cap * sizeof *b))) return COV_NOMEM;
        h->gsample = b;
        h->gcap = cap;
    }
    h->gid[h->gn] = (uint32_t)id;
    h->gsample[h->gn++] = g;
    return COV_OK;
}

static int to_general(hitlist *h)
{
    uint64_t i;
    char buf[32];
    for (i = 0; i < h->n; i++) {                      /* replay what was packed so far, in file order */
        size_t j;
        int rc;
        for (j = 0; j < h->klen; j++) buf[j] = "ACGT"[(h->key[i] >> (2 * (h->klen - 1 - j))) & 3u];
        if ((rc = general_count(h, h->sample[i], buf, h->klen)) != COV_OK) return rc;
    }
    h->n = 0;
    h->general = 1;
    return COV_OK;
}

/* one hit that passed the read filter: sample = basename of the metagenome, key = packed k-mer or row */
static int64_t cov_sample(hitlist *h, const char *sn, size_t slen)
{
    const int64_t g = name_get(&h->smp, sn, slen);
    if (g >= 0 && !h->smp.in_depth[g]) {
        uint32_t *o = realloc(h->depth_order, ((size_t)h->ndepth + 1) * sizeof *o);
        if (!o) return -1;
        h->depth_order = o;
        o[h->ndepth++] = (uint32_t)g;
        h->smp.in_depth[g] = 1;
    }
    return g;
}

static int cov_push(hitlist *h, uint32_t g, uint64_t key)
{
    if (h->n == h->cap) {
        const uint64_t cap = h->cap ? h->cap * 2 : 1u << 16;
        uint64_t *k2 = realloc(h->key, cap * sizeof *k2);
        uint32_t *s2;
        if (!k2) return COV_NOMEM;
        h->key = k2;
        if (!(s2 = realloc(h->sample, cap * sizeof *s2))) return COV_NOMEM;
        h->sample = s2;
        h->cap = cap;
    }
    h->key[h->n] = key;
    h->sample[h->n++] = g;
    return COV_OK;
}

static int hit_line(hitlist *h, char *s, size_t len, int64_t min_hits)
{
    char *f[7], *end = s + len, *p;
    int nf = 1;
    int64_t a, b, c, d, g;
    const char *sn, *kend;
    size_t klen, j;
    uint64_t key = 0;
    f[0] = s;
    for (p = s; p < end; p++)
        if (*p == '\t') { if (nf < 7) f[nf] = p + 1; nf++; }
    if (nf < 6) return COV_FIELDS;
    if (!skp_int(f[1], f[2] - 1, &a) || !skp_int(f[2], f[3] - 1, &b) || !skp_int(f[3], f[4] - 1, &c) || !skp_int(f[4], f[5] - 1, &d))
        return COV_INT;
    if (!(a + c > min_hits)) return COV_OK;           /* hits of the read pair, informative or not (:83-86) */
    sn = base_of(f[0], f[1] - 1);
    if ((g = cov_sample(h, sn, (size_t)(f[1] - 1 - sn))) < 0) return COV_NOMEM;
    kend = nf > 6 ? f[6] - 1 : end;
    klen = (size_t)(kend - f[5]);
    if (!h->general) {
        int plain = klen >= 1 && klen <= 31 && (h->klen == 0 || h->klen == klen);
        for (j = 0; plain && j < klen; j++) {
            const unsigned cde = sk_code((uint8_t)f[5][j]);
            if (f[5][j] != "ACGT"[cde]) plain = 0;     /* upper-case A/C/G/T only */
            key = (key << 2) | (cde & 3u);
        }
        if (plain) h->klen = klen;
        else { int rc = to_general(h); if (rc) return rc; }
    }
    if (h->general) return general_count(h, (uint32_t)g, f[5], klen);
    return cov_push(h, (uint32_t)g, key);
}

/* one trailer value of a sample (:104-116) */
static int cov_trailer(hitlist *h, const char *sn, size_t slen, const char *var, size_t vlen, int64_t v)
{
    const int64_t g = name_get(&h->smp, sn, slen);
    if (g < 0) return COV_NOMEM;
#define IS(word) (vlen == sizeof(word) - 1 && !memcmp(var, word, sizeof(word) - 1))
    if (IS("total_kmer_evaluated")) {
        if (!h->smp.have_eval[g]) {
            uint32_t *o = realloc(h->eval_order, ((size_t)h->neval + 1) * sizeof *o);
            if (!o) return COV_NOMEM;
            h->eval_order = o;
            o[h->neval++] = (uint32_t)g;
        }
        h->smp.kmer_eval[g] = v; h->smp.have_eval[g] = 1;
    }
    else if (IS("total_reads_evaluated")) { h->smp.read_eval[g] = v; h->smp.have_reads[g] = 1; }
    else if (IS("total_genome_kmers")) { h->smp.g_total[g] = v; h->smp.have_total[g] = 1; }
    else if (IS("total_genome_informative_kmers")) { h->smp.g_inf[g] = v; h->smp.have_inf[g] = 1; }
#undef IS
    return COV_OK;
}

/* "#<metagenome>\t<name>\t<value>" (:101-116) */
static int trailer_line(hitlist *h, char *s, size_t len)
{
    char *f[4], *end, *p;
    int nf = 1;
    int64_t v;
    const char *sn;
    while (len && isspace((unsigned char)s[len - 1])) len--;      /* line.rstrip() */
    end = s + len;
    f[0] = s;
    for (p = s; p < end; p++)
        if (*p == '\t') { if (nf < 4) f[nf] = p + 1; nf++; }
    if (nf < 3) return COV_FIELDS;
    sn = base_of(f[0], f[1] - 1);
    if (sn < f[1] - 1 && *sn == '#') sn++;
    if (!skp_int(f[2], nf > 3 ? f[3] - 1 : end, &v)) return COV_INT;
    return cov_trailer(h, sn, (size_t)(f[1] - 1 - sn), f[1], (size_t)(f[2] - 1 - f[1]), v);
}

static int read_hits(hitlist *h, const char *path, int64_t min_hits)
{
    gzFile gz = gzopen(path, "rb");
    size_t cap = 1u << 20, have = 0;
    char *buf;
    int n, rc = COV_OK;
    if (!gz) return COV_OPEN;
    gzbuffer(gz, 1u << 20);
    if (!(buf = malloc(cap))) { gzclose(gz); return COV_NOMEM; }
    for (;;) {
        char *line, *nl, *end;
        if (have == cap) {
            char *b = realloc(buf, cap * 2);
            if (!b) { rc = COV_NOMEM; break; }
            buf = b; cap *= 2;
        }
        n = gzread(gz, buf + have, (unsigned)(cap - have));
        if (n < 0) { rc = COV_OPEN; break; }
        have += (size_t)n;
        end = buf + have;
        line = buf;
        while (line < end && ((nl = memchr(line, '\n', (size_t)(end - line))) != NULL || n == 0)) {
            size_t len = nl ? (size_t)(nl - line) : (size_t)(end - line);
            if (len && line[len - 1] == '\r' && nl) len--;
            rc = (len && line[0] == '#') ? trailer_line(h, line, len) : hit_line(h, line, len, min_hits);
            if (rc) goto out;
            line = nl ? nl + 1 : end;
        }
        have = (size_t)(end - line);
        memmove(buf, line, have);
        if (n == 0) break;
    }
out:
    free(buf);
    gzclose(gz);
    return rc;
}

/* ------------------------------------------------------------------ the program */
static int cov_opt(int argc, char **argv, int *i, const char *shortn, const char *longn, const char **val, FILE *err, int *bad)
{
    const char *a = argv[*i];
    const size_t ll = strlen(longn);
    if (!strcmp(a, shortn) || !strcmp(a, longn)) {
        if (*i + 1 >= argc) { fprintf(err, "coverage_depth: error: argument %s/%s: expected one argument\n", longn, shortn); *bad = 1; return 1; }
        *val = argv[++*i];
        return 1;
    }
    if (!strncmp(a, longn, ll) && a[ll] == '=') { *val = a + ll + 1; return 1; }
    if (!strncmp(a, shortn, 2) && a[2]) { *val = a + 2; return 1; }
    return 0;
}

/* everything after the hit list is read: trailer-only samples, the counts (device), the table (:121-124,200-271).
 * `kfile` only names the strain.  Returns the exit status. */
static int cov_report(hitlist *h, sk_ctx *ctx, const char *kfile, const char *bfile, FILE *out, FILE *err)
{
    names bg;
    uint64_t *uniq = NULL, *total = NULL;
    char *strain = NULL, *species = NULL, *genus = NULL, *us;
    size_t sl;
    uint32_t s, k;
    int rc, status = 1;
    memset(&bg, 0, sizeof bg);
    /* metagenomes that only have trailer lines come after those with passing hits (:121-124) */
    for (k = 0; k < h->neval; k++)
        if (!h->smp.in_depth[s = h->eval_order[k]]) {
            uint32_t *o = realloc(h->depth_order, ((size_t)h->ndepth + 1) * sizeof *o);
            if (!o) { fprintf(err, "coverage_depth: out of memory\n"); goto done; }
            h->depth_order = o;
            o[h->ndepth++] = s;
            h->smp.in_depth[s] = 1;
        }
    if (h->smp.n) {
        if (!(uniq = calloc(h->smp.n, sizeof *uniq)) || !(total = calloc(h->smp.n, sizeof *total))) { fprintf(err, "coverage_depth: out of memory\n"); goto done; }
        if (h->general) {                            /* out-of-domain k-mer text: first sightings of the joined strings */
            rc = sk_first_seen_count(ctx, h->gid, h->gsample, h->gn, h->text.n, h->smp.n, uniq, total);
            if (rc != SK_OK) { fprintf(err, "coverage_depth: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx)); goto done; }
        } else {
            rc = sk_distinct_count(ctx, h->key, h->sample, h->n, h->smp.n, uniq, total);
            if (rc != SK_OK) { fprintf(err, "coverage_depth: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx)); goto done; }
        }
    }
    /* names derived from the file name (:203-213): strip "<any>kmer_hits<any>gz" at the end, split at '_' */
    strain = strdup(base_of(kfile, kfile + strlen(kfile)));
    sl = strlen(strain);
    if (sl >= 13 && !memcmp(strain + sl - 12, "kmer_hits", 9) && !memcmp(strain + sl - 2, "gz", 2)) strain[sl - 13] = 0;
    species = strdup(strain);
    genus = strdup(strain);
    if ((us = strchr(genus, '_')) != NULL) {
        *us = 0;
        if ((us = strchr(species + strlen(genus) + 1, '_')) != NULL) *us = 0;
    }
    if (bfile) {                                     /* :131-140 */
        FILE *bf = fopen(bfile, "r");
        char *line = NULL;
        size_t cap = 0;
        ssize_t n;
        if (!bf) { fprintf(err, "coverage_depth: could not read %s\n", bfile); goto done; }
        while ((n = getline(&line, &cap, bf)) >= 0) {
            while (n > 0 && line[n - 1] == '\n') line[--n] = 0;
            if (name_get(&bg, line, (size_t)n) < 0) { fclose(bf); free(line); goto done; }
        }
        free(line);
        fclose(bf);
    }
    fputs("strain_name\tspecies_name\tgenus_name\tgenome_num_total_kmers\tgenome_num_informative_kmers\tmetagenome\t"
          "num_metagenomic_reads\tnum_metagenome_kmers\tunique_observed_informative_kmers\ttotal_observed_informative_kmers\t"
          "kmer_coverage\tkmer_depth\tkmer_depth_per_20B_kmer\tbackground\n", out);
    for (k = 0; k < h->ndepth; k++) {                /* :227-271 */
        const names *t = &h->smp;
        const char *m;
        int64_t observed, unique, evaluated, reads, gt, gi;
        int in_bg = 0;
        uint32_t j;
        char c1[32], c2[32], c3[32];
        s = h->depth_order[k];
        m = t->name[s];
        observed = (int64_t)total[s];
        /* general mode only: a sample all of whose joined strings were first seen under another sample
         * never enters the script's coverage dictionary and prints the -1 placeholder (:236-243) */
        unique = (h->general && total[s] && !uniq[s]) ? -1 : (int64_t)uniq[s];
        evaluated = t->have_eval[s] ? t->kmer_eval[s] : -1;
        reads = t->have_eval[s] ? (t->have_reads[s] ? t->read_eval[s] : 0) : -1;
        gt = t->have_total[s] ? t->g_total[s] : -1;
        gi = t->have_inf[s] ? t->g_inf[s] : -1;
        if (gi == 0) { fflush(out); fprintf(err, "ZeroDivisionError: float division by zero\n"); goto done; }
        skp_float_str((double)unique / (double)gi, c1);
        skp_float_str((double)observed / (double)gi, c2);
        if (evaluated == 0) strcpy(c3, "0");
        else skp_float_str(((double)observed / (double)gi) * (2000000000 / (double)evaluated), c3);
        for (j = 0; j < bg.n && !in_bg; j++) in_bg = !strcmp(bg.name[j], m);
        fprintf(out, "%s\t%s\t%s\t%lld\t%lld\t%s\t%lld\t%lld\t%lld\t%lld\t%s\t%s\t%s\t%d\n", strain, species, genus, (long long)gt,
                (long long)gi, m, (long long)reads, (long long)evaluated, (long long)unique, (long long)observed, c1, c2, c3, in_bg);
    }
    status = 0;
done:
    fflush(out);
    free(uniq); free(total); free(strain); free(species); free(genus);
    names_free(&bg);
    return status;
}

static void hitlist_free(hitlist *h)
{
    free(h->key); free(h->sample); free(h->depth_order); free(h->eval_order); free(h->gid); free(h->gsample);
    names_free(&h->smp); names_free(&h->text);
    memset(h, 0, sizeof *h);
}

/* ---- the same accumulator fed directly by strain_detect (fused steps 3 -> 4; sk_internal.h) ---- */
struct skc_acc { hitlist h; int64_t min_hits; };

skc_acc *skc_create(int64_t min_hits)
{
    skc_acc *a = calloc(1, sizeof *a);
    if (a) a->min_hits = min_hits;
    return a;
}

void skc_destroy(skc_acc *a) { if (a) { hitlist_free(&a->h); free(a); } }

/* what one hit line of the -o file would contribute: metagenome path, the pair's hit counts of PE1 and PE2
 * (fields 2 and 4), and the k-mer named by its row */
int skc_add_hit(skc_acc *a, const char *metagenome, int64_t hits_pe1, int64_t hits_pe2, uint32_t row)
{
    const char *sn;
    int64_t g;
    if (!(hits_pe1 + hits_pe2 > a->min_hits)) return 0;
    sn = base_of(metagenome, metagenome + strlen(metagenome));
    if ((g = cov_sample(&a->h, sn, strlen(sn))) < 0) return -1;
    return cov_push(&a->h, (uint32_t)g, (uint64_t)row) == COV_OK ? 0 : -1;
}

int skc_add_trailer(skc_acc *a, const char *metagenome, const char *name, int64_t value)
{
    const char *sn = base_of(metagenome, metagenome + strlen(metagenome));
    return cov_trailer(&a->h, sn, strlen(sn), name, strlen(name), value) == COV_OK ? 0 : -1;
}

int skc_report(skc_acc *a, sk_ctx *ctx, const char *hits_file_name, FILE *out, FILE *err)
{
    return cov_report(&a->h, ctx, hits_file_name, NULL, out, err);
}

int skh_coverage_depth_main(int argc, char **argv, FILE *out, FILE *err)
{
    const char *kfile = NULL, *mtext = NULL, *bfile = NULL, *env;
    int64_t min_hits = 1;
    int i, bad = 0, rc, status = 1, device = 0;
    hitlist h;
    sk_ctx *ctx = NULL;

    for (i = 1; i < argc; i++) {                     /* :27-41 */
        if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) {
            fprintf(out, "usage: coverage_depth [-h] --kmer_hits_file FILE [--min_kmer_hits N] [--background_metagenomes_file FILE]\n");
            return 0;
        }
        if (cov_opt(argc, argv, &i, "-k", "--kmer_hits_file", &kfile, err, &bad)) { if (bad) return 2; continue; }
        if (cov_opt(argc, argv, &i, "-m", "--min_kmer_hits", &mtext, err, &bad)) { if (bad) return 2; continue; }
        if (cov_opt(argc, argv, &i, "-b", "--background_metagenomes_file", &bfile, err, &bad)) { if (bad) return 2; continue; }
        fprintf(err, "coverage_depth: error: unrecognized arguments: %s\n", argv[i]);
        return 2;
    }
    if (!kfile) { fprintf(err, "coverage_depth: error: the following arguments are required: --kmer_hits_file/-k\n"); return 2; }
    if (mtext && !skp_int(mtext, mtext + strlen(mtext), &min_hits)) {
        fprintf(err, "coverage_depth: error: argument --min_kmer_hits/-m: invalid int value: '%s'\n", mtext);
        return 2;
    }
    memset(&h, 0, sizeof h);
    rc = read_hits(&h, kfile, min_hits);
    if (rc == COV_OPEN) { fprintf(err, "coverage_depth: could not read %s\n", kfile); goto done; }
    if (rc == COV_FIELDS) { fprintf(err, "coverage_depth: %s: a line has too few tab-separated fields\n", kfile); goto done; }
    if (rc == COV_INT) { fprintf(err, "coverage_depth: %s: a count field is not an integer\n", kfile); goto done; }
    if (rc == COV_NOMEM) { fprintf(err, "coverage_depth: out of memory\n"); goto done; }
    if ((env = getenv("SK_DEVICE")) != NULL) device = atoi(env);
    rc = sk_ctx_create(&ctx, device);
    if (rc != SK_OK) { fprintf(err, "coverage_depth: cannot use HIP device %d: %s\n", device, sk_strerror(rc)); goto done; }
    status = cov_report(&h, ctx, kfile, bfile, out, err);
done:
    fflush(out);
    sk_ctx_destroy(ctx);
    hitlist_free(&h);
    return status;
}
