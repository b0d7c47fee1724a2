/* oracle/ksf_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into, called by, or shipped with the
 * product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *
 * CPU restatement, in plain C, of the two Python scripts either side of the scan path
 * (SURVEY.md §8f rows 2 and 4):
 *
 *   ksf_main   reference scripts/kmer_scrub_filter.py   (step 2 of test/example.sh)
 *   kcd_main   reference scripts/coverage_depth.py      (step 4 of test/example.sh)
 *
 * The restatement keeps the scripts' data model -- insertion-ordered dictionaries keyed by the
 * k-mer / sample TEXT, Python integer and double arithmetic, Python's float repr -- so that it can be
 * compared byte for byte with what the scripts print.  It is written for clarity, not speed.
 *
 * Parity status: PINNED.  tests/test_filter_oracle.py checks it against every fixture under
 * tests/golden/filter_cases and tests/golden/cov_cases; those fixtures were produced by running the
 * reference's own scripts (tests/golden/make_golden_filter.py).
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ Python's repr() of a float */
/* shortest decimal string that reads back as the same double, laid out as float.__repr__ does:
 * fixed notation for 1e-4 <= |x| < 1e16, otherwise d.ddde+XX */
static void py_float_repr(double x, char *out)
{
    if (x != x) { strcpy(out, "nan"); return; }
    if (x == 1.0 / 0.0) { strcpy(out, "inf"); return; }
    if (x == -1.0 / 0.0) { strcpy(out, "-inf"); return; }
    char buf[64];
    int prec;
    for (prec = 1; prec <= 17; prec++) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, x);
        if (strtod(buf, NULL) == x) break;
    }
    /* buf = [-]d[.ddd]e[+-]XX */
    char digits[32];
    int nd = 0, neg = buf[0] == '-';
    const char *p = buf + neg;
    for (; *p && *p != 'e'; p++)
        if (isdigit((unsigned char)*p)) digits[nd++] = *p;
    int exp10 = atoi(p + 1);
    while (nd > 1 && digits[nd - 1] == '0') nd--;
    digits[nd] = 0;
    char *o = out;
    if (neg) *o++ = '-';
    int decpt = exp10 + 1;                        /* digits = 0.d1d2.. x 10^decpt */
    if (decpt > 16 || decpt < -3) {
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; memcpy(o, digits + 1, (size_t)nd - 1); o += nd - 1; }
        sprintf(o, "e%c%02d", exp10 < 0 ? '-' : '+', abs(exp10));
        return;
    }
    if (decpt <= 0) {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -decpt; i++) *o++ = '0';
        memcpy(o, digits, (size_t)nd); o += nd;
    } else if (decpt >= nd) {
        memcpy(o, digits, (size_t)nd); o += nd;
        for (int i = nd; i < decpt; i++) *o++ = '0';
        *o++ = '.'; *o++ = '0';
    } else {
        memcpy(o, digits, (size_t)decpt); o += decpt;
        *o++ = '.';
        memcpy(o, digits + decpt, (size_t)(nd - decpt)); o += nd - decpt;
    }
    *o = 0;
}

/* ------------------------------------------------------------------ an insertion-ordered dict */
typedef struct { char *key; int64_t iv; double dv; int alive; } ditem;
typedef struct { ditem *it; size_t n, cap, live; int64_t *slot; size_t nslot; } dict;

static uint64_t str_hash(const char *s)
{
    uint64_t h = 1469598103934665603ull;
    for (; *s; s++) h = (h ^ (unsigned char)*s) * 1099511628211ull;
    return h;
}
static void dict_init(dict *d) { memset(d, 0, sizeof *d); d->nslot = 1024; d->slot = malloc(d->nslot * sizeof *d->slot); for (size_t i = 0; i < d->nslot; i++) d->slot[i] = -1; }
static void dict_free(dict *d) { for (size_t i = 0; i < d->n; i++) free(d->it[i].key); free(d->it); free(d->slot); memset(d, 0, sizeof *d); }
static void dict_rehash(dict *d)
{
    d->nslot *= 2;
    d->slot = realloc(d->slot, d->nslot * sizeof *d->slot);
    for (size_t i = 0; i < d->nslot; i++) d->slot[i] = -1;
    for (size_t i = 0; i < d->n; i++) {
        size_t s = str_hash(d->it[i].key) & (d->nslot - 1);
        while (d->slot[s] >= 0) s = (s + 1) & (d->nslot - 1);
        d->slot[s] = (int64_t)i;
    }
}
/* a deleted key keeps its item (alive = 0); setting it again appends a NEW item at the end, as a
 * Python dict would */
static ditem *dict_find(const dict *d, const char *key)
{
    size_t s = str_hash(key) & (d->nslot - 1);
    ditem *hit = NULL;
    while (d->slot[s] >= 0) {
        ditem *it = &d->it[d->slot[s]];
        if (it->alive && !strcmp(it->key, key)) hit = it;
        s = (s + 1) & (d->nslot - 1);
    }
    return hit;
}
static ditem *dict_set(dict *d, const char *key)
{
    ditem *it = dict_find(d, key);
    if (it) return it;
    if ((d->n + 1) * 2 > d->nslot) dict_rehash(d);
    if (d->n == d->cap) { d->cap = d->cap ? d->cap * 2 : 1024; d->it = realloc(d->it, d->cap * sizeof *d->it); }
    it = &d->it[d->n];
    it->key = strdup(key); it->iv = 0; it->dv = 0; it->alive = 1;
    size_t s = str_hash(key) & (d->nslot - 1);
    while (d->slot[s] >= 0) s = (s + 1) & (d->nslot - 1);
    d->slot[s] = (int64_t)d->n;
    d->n++; d->live++;
    return it;
}
static void dict_del(dict *d, const char *key) { ditem *it = dict_find(d, key); if (it) { it->alive = 0; d->live--; } }
/* Python dict equality: same live key set, equal values */
static int dict_equal(const dict *a, const dict *b)
{
    if (a->live != b->live) return 0;
    for (size_t i = 0; i < a->n; i++) {
        if (!a->it[i].alive) continue;
        const ditem *o = dict_find(b, a->it[i].key);
        if (!o || o->iv != a->it[i].iv) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ text helpers */
/* gzip.open(file, 'rt') line iteration: returns a malloc'd line WITHOUT its newline ("\r\n" counts as
 * a newline, as universal-newline text mode has it), or NULL at end of file */
typedef struct { gzFile f; char *buf; size_t cap; } lines;
static char *next_line(lines *L, size_t *len)
{
    size_t n = 0;
    for (;;) {
        if (n + 2 > L->cap) { L->cap = L->cap ? L->cap * 2 : 256; L->buf = realloc(L->buf, L->cap); }
        if (!gzgets(L->f, L->buf + n, (int)(L->cap - n))) { if (n == 0) return NULL; break; }
        n += strlen(L->buf + n);
        if (n && L->buf[n - 1] == '\n') { n--; if (n && L->buf[n - 1] == '\r') n--; break; }
    }
    L->buf[n] = 0;
    *len = n;
    return L->buf;
}
/* split on tabs in place */
static int split_tabs(char *s, char **f, int maxf)
{
    int n = 0;
    f[n++] = s;
    for (; *s; s++)
        if (*s == '\t') { *s = 0; if (n < maxf) f[n++] = s + 1; else n++; }
    return n;
}
/* Python int(): optional blanks, sign, decimal digits, optional blanks */
static int py_int(const char *s, int64_t *out)
{
    while (isspace((unsigned char)*s)) s++;
    errno = 0;
    char *e;
    if (!(isdigit((unsigned char)*s) || ((*s == '-' || *s == '+') && isdigit((unsigned char)s[1])))) return 0;
    long long v = strtoll(s, &e, 10);
    if (errno) return 0;
    while (isspace((unsigned char)*e)) e++;
    if (*e) return 0;
    *out = v;
    return 1;
}
static const char *base_name(const char *p) { const char *s = strrchr(p, '/'); return s ? s + 1 : p; }

/* ================================================================== kmer_scrub_filter.py */
/* scrub_max_kmers (scripts/kmer_scrub_filter.py:30-58): raise the threshold until at most
 * (1 - min_frac) of ALL k-mers exceed it; every step is reported on stderr */
static int64_t scrub_max_kmers(double min_frac, const dict *h, double total, int *zero_div)
{
    int64_t thr = -1;
    double kept = -1.0;
    char r[64];
    while (kept < min_frac) {
        thr++;
        int64_t hits = 0;
        for (size_t i = 0; i < h->n; i++)
            if (h->it[i].alive && h->it[i].iv > thr) hits++;
        if (total == 0.0) { *zero_div = 1; return 0; }
        kept = 1 - ((double)hits / total);
        py_float_repr(kept, r);
        fprintf(stderr, "kept %s with threshold %lld\n", r, (long long)thr);
    }
    size_t left = 0;
    for (size_t i = 0; i < h->n; i++)
        if (h->it[i].alive && h->it[i].iv > thr) left++;
    py_float_repr(total, r);
    fprintf(stderr, "threshold was %lld left with %zu out of %s that will be scrubbed\n", (long long)thr, left, r);
    return thr;
}

static void merge_sort_desc(size_t *idx, size_t *tmp, size_t n, const double *v)
{
    if (n < 2) return;
    size_t h = n / 2;
    merge_sort_desc(idx, tmp, h, v);
    merge_sort_desc(idx + h, tmp, n - h, v);
    size_t a = 0, b = h, o = 0;
    while (a < h && b < n) tmp[o++] = v[idx[b]] > v[idx[a]] ? idx[b++] : idx[a++];   /* stable: left wins ties */
    while (a < h) tmp[o++] = idx[a++];
    while (b < n) tmp[o++] = idx[b++];
    memcpy(idx, tmp, n * sizeof *idx);
}

int ksf_main(int argc, char **argv)
{
    const char *sfile = NULL, *lfile = NULL;
    double min_fraction = 0.04;
    int independent = 0;
    for (int i = 1; i < argc; i++) {                 /* the argparse surface the workflow uses (:14-28) */
        const char *a = argv[i], *v = NULL;
        int kind = 0;
        if (!strcmp(a, "-i") || !strcmp(a, "--independent")) { independent = 1; continue; }
        if (!strcmp(a, "-s") || !strcmp(a, "--scrub_count_file")) kind = 1;
        else if (!strcmp(a, "-l") || !strcmp(a, "--scrub_count_list")) kind = 2;
        else if (!strcmp(a, "-m") || !strcmp(a, "--min_fraction")) kind = 3;
        else if (!strncmp(a, "--scrub_count_file=", 19)) { kind = 1; v = a + 19; }
        else if (!strncmp(a, "--scrub_count_list=", 19)) { kind = 2; v = a + 19; }
        else if (!strncmp(a, "--min_fraction=", 15)) { kind = 3; v = a + 15; }
        else { fprintf(stderr, "error: unrecognized arguments: %s\n", a); return 2; }
        if (!v) { if (++i >= argc) { fprintf(stderr, "error: argument %s: expected one argument\n", a); return 2; } v = argv[i]; }
        if (kind == 1) sfile = v;
        else if (kind == 2) lfile = v;
        else {
            char *e;
            min_fraction = strtod(v, &e);
            if (e == v || *e) { fprintf(stderr, "error: argument --min_fraction/-m: invalid float value: '%s'\n", v); return 2; }
        }
    }
    if (min_fraction < 0.0 || min_fraction > 1.0) {  /* :142-143 builds str + float and dies with TypeError */
        fprintf(stderr, "TypeError: can only concatenate str (not \"float\") to str\n");
        return 1;
    }
    if (!sfile && !lfile) fprintf(stderr, "error: one of scrub_count_file or scrub_count_list must be provided.");
    if (sfile && lfile) fprintf(stderr, "error: can provide only one of either scrub_count_file or scrub_count_list.");

    char **files = NULL;
    size_t nfiles = 0;
    if (sfile) { files = malloc(sizeof *files); files[nfiles++] = strdup(sfile); }
    else if (lfile) {                                /* :152-155, each line rstrip()ped */
        FILE *f = fopen(lfile, "r");
        if (!f) { fprintf(stderr, "FileNotFoundError: [Errno 2] No such file or directory: '%s'\n", lfile); return 1; }
        char *line = NULL; size_t cap = 0; ssize_t n;
        while ((n = getline(&line, &cap, f)) >= 0) {
            while (n > 0 && isspace((unsigned char)line[n - 1])) line[--n] = 0;
            files = realloc(files, (nfiles + 1) * sizeof *files);
            files[nfiles++] = strdup(line);
        }
        free(line); fclose(f);
    }

    dict strain, prev, meta, pan, drug;
    dict_init(&strain); dict_init(&prev); dict_init(&meta); dict_init(&pan); dict_init(&drug);
    int drug_filter = 0;
    int64_t all_kmers = 0;
    for (size_t i = 0; i < nfiles; i++) {            /* :164-201 */
        if (i > 1) { dict_free(&prev); prev = strain; dict_init(&strain); }
        else { dict_free(&strain); dict_init(&strain); }
        all_kmers = 0;
        lines L = {0};
        L.f = gzopen(files[i], "rb");
        if (!L.f) { fprintf(stderr, "FileNotFoundError: [Errno 2] No such file or directory: '%s'\n", files[i]); return 1; }
        char *line; size_t len;
        while ((line = next_line(&L, &len))) {
            if (line[0] == '#') continue;
            char *f[6];
            int nf = split_tabs(line, f, 6);
            int64_t c1, c2, c3, c4 = 0;
            if (nf < 4) { fprintf(stderr, "IndexError: list index out of range\n"); return 1; }
            if (!py_int(f[1], &c1) || !py_int(f[2], &c2) || !py_int(f[3], &c3)) { fprintf(stderr, "ValueError: invalid literal for int() with base 10\n"); return 1; }
            all_kmers++;
            dict_set(&strain, f[0])->iv = c1;
            if (c2 > 0) dict_set(&pan, f[0])->iv += c2;
            if (c3 > 0) dict_set(&meta, f[0])->iv += c3;
            if (nf == 5) {
                drug_filter = 1;
                if (!py_int(f[4], &c4)) { fprintf(stderr, "ValueError: invalid literal for int() with base 10\n"); return 1; }
                if (c4 > 0) dict_set(&drug, f[0])->iv += c3;      /* sic: adds the metagenome field (:194) */
            }
        }
        gzclose(L.f); free(L.buf);
        if (i > 1 && !dict_equal(&strain, &prev)) {
            fprintf(stderr, "error: input files do not have identical hash and strain hash values.\n");
            return 1;
        }
    }
    printf("#total kmers in strain:%lld,%zu pangenome: %zu metagenome: %zu\n", (long long)all_kmers, strain.live, pan.live, meta.live);

    int64_t drug_scrubbed = 0;
    char r[64];
    if (drug_filter) {                               /* :207-218 */
        printf("#total kmers cross drug:%zu\n", drug.live);
        for (size_t i = 0; i < drug.n; i++) if (drug.it[i].alive) dict_del(&strain, drug.it[i].key);
        if (all_kmers == 0) { fflush(stdout); fprintf(stderr, "ZeroDivisionError: float division by zero\n"); return 1; }
        double frac = (double)strain.live / (double)all_kmers;
        drug_scrubbed = all_kmers - (int64_t)strain.live;
        py_float_repr(frac, r);
        printf("#fraction kmers remaining drug post scrub:%s\n", r);
        printf("#drug_scrubbed kmers:%lld\n", (long long)drug_scrubbed);
        if (frac < min_fraction * 2) {
            fflush(stdout);
            fprintf(stderr, "Exception: ERROR: too few kmers remain after drug scrub. Are your drug strains too similar?\n");
            return 1;
        }
    }

    if (independent) {                               /* independent_scrub (:72-84) */
        int zd = 0;
        int64_t tp = scrub_max_kmers(min_fraction, &pan, (double)all_kmers, &zd);
        if (zd) { fflush(stdout); fprintf(stderr, "ZeroDivisionError: float division by zero\n"); return 1; }
        int64_t tm = scrub_max_kmers(min_fraction, &meta, (double)all_kmers, &zd);
        for (size_t i = 0; i < pan.n; i++) if (pan.it[i].alive && pan.it[i].iv > tp) dict_del(&strain, pan.it[i].key);
        for (size_t i = 0; i < meta.n; i++) if (meta.it[i].alive && meta.it[i].iv > tm) dict_del(&strain, meta.it[i].key);
    } else {                                         /* joint_scrub (:87-137) */
        int64_t msum = 0, psum = 0;
        for (size_t i = 0; i < meta.n; i++) msum += meta.it[i].iv;
        for (size_t i = 0; i < meta.n; i++) meta.it[i].dv = (double)meta.it[i].iv / (double)msum;
        for (size_t i = 0; i < pan.n; i++) psum += pan.it[i].iv;
        for (size_t i = 0; i < pan.n; i++) pan.it[i].dv = (double)pan.it[i].iv / (double)psum;
        size_t n = 0, *idx = malloc((strain.live + 1) * sizeof *idx), *tmp = malloc((strain.live + 1) * sizeof *tmp);
        double *score = malloc((strain.n + 1) * sizeof *score);
        for (size_t i = 0; i < strain.n; i++) {
            if (!strain.it[i].alive) continue;
            double v = 0;
            const ditem *m = dict_find(&meta, strain.it[i].key), *p = dict_find(&pan, strain.it[i].key);
            if (m && m->dv > v) v = m->dv;
            if (p && p->dv > v) v = p->dv;
            score[i] = v;
            idx[n++] = i;
        }
        merge_sort_desc(idx, tmp, n, score);         /* sorted(..., reverse=True) is stable */
        double num_scrubbed = (double)drug_scrubbed;
        for (size_t j = 0; j < n; j++)
            if ((1 - ((num_scrubbed + 1) / (double)all_kmers)) > min_fraction) {
                num_scrubbed += 1.0;
                strain.it[idx[j]].alive = 0; strain.live--;
            }
        free(idx); free(tmp); free(score);
    }
    printf("#post scrub kmers %zu out of %lld\n", strain.live, (long long)all_kmers);
    for (size_t i = 0; i < strain.n; i++)
        if (strain.it[i].alive) { fputs(strain.it[i].key, stdout); fputc('\n', stdout); }
    return 0;
}

/* ================================================================== coverage_depth.py */
int kcd_main(int argc, char **argv)
{
    const char *kfile = NULL, *bfile = NULL;
    int64_t min_hits = 1;
    for (int i = 1; i < argc; i++) {                 /* :27-41 */
        const char *a = argv[i], *v = NULL;
        int kind = 0;
        if (!strcmp(a, "-k") || !strcmp(a, "--kmer_hits_file")) kind = 1;
        else if (!strcmp(a, "-m") || !strcmp(a, "--min_kmer_hits")) kind = 2;
        else if (!strcmp(a, "-b") || !strcmp(a, "--background_metagenomes_file")) kind = 3;
        else if (!strncmp(a, "--kmer_hits_file=", 17)) { kind = 1; v = a + 17; }
        else if (!strncmp(a, "--min_kmer_hits=", 16)) { kind = 2; v = a + 16; }
        else if (!strncmp(a, "--background_metagenomes_file=", 30)) { kind = 3; v = a + 30; }
        else { fprintf(stderr, "error: unrecognized arguments: %s\n", a); return 2; }
        if (!v) { if (++i >= argc) { fprintf(stderr, "error: argument %s: expected one argument\n", a); return 2; } v = argv[i]; }
        if (kind == 1) kfile = v;
        else if (kind == 3) bfile = v;
        else if (!py_int(v, &min_hits)) { fprintf(stderr, "error: argument --min_kmer_hits/-m: invalid int value: '%s'\n", v); return 2; }
    }
    if (!kfile) { fprintf(stderr, "error: the following arguments are required: --kmer_hits_file/-k\n"); return 2; }

    /* count_passed_kmers (:62-129).  depth's insertion order is the order of the printed rows. */
    dict depth, cover, uniq, kmer_eval, read_eval, g_total, g_inf;
    dict_init(&depth); dict_init(&cover); dict_init(&uniq); dict_init(&kmer_eval); dict_init(&read_eval);
    dict_init(&g_total); dict_init(&g_inf);
    lines L = {0};
    L.f = gzopen(kfile, "rb");
    if (!L.f) { fprintf(stderr, "FileNotFoundError: [Errno 2] No such file or directory: '%s'\n", kfile); return 1; }
    char *line; size_t len;
    while ((line = next_line(&L, &len))) {
        char *f[7];
        if (line[0] != '#') {
            int nf = split_tabs(line, f, 7);
            if (nf < 6) { fprintf(stderr, "IndexError: list index out of range\n"); return 1; }
            const char *sample = base_name(f[0]);
            int64_t a, b, c, d;
            if (!py_int(f[1], &a) || !py_int(f[2], &b) || !py_int(f[3], &c) || !py_int(f[4], &d)) { fprintf(stderr, "ValueError: invalid literal for int() with base 10\n"); return 1; }
            if (a + c > min_hits) {                  /* total k-mer hits of the read pair, not informative ones (:83-86) */
                char *u = malloc(strlen(sample) + strlen(f[5]) + 1);
                strcpy(u, sample); strcat(u, f[5]);
                if (!dict_find(&uniq, u)) { dict_set(&cover, sample)->iv += 1; dict_set(&uniq, u)->iv = 1; }
                free(u);
                dict_set(&depth, sample)->iv += 1;
            }
        } else {                                     /* trailer lines "#<file>\t<name>\t<value>" (:101-116) */
            while (len && isspace((unsigned char)line[len - 1])) line[--len] = 0;
            int nf = split_tabs(line, f, 7);
            if (nf < 3) { fprintf(stderr, "IndexError: list index out of range\n"); return 1; }
            const char *sample = base_name(f[0]);
            if (*sample == '#') sample++;
            int64_t v;
            if (!py_int(f[2], &v)) { fprintf(stderr, "ValueError: invalid literal for int() with base 10\n"); return 1; }
            if (!strcmp(f[1], "total_kmer_evaluated")) dict_set(&kmer_eval, sample)->iv = v;
            else if (!strcmp(f[1], "total_reads_evaluated")) dict_set(&read_eval, sample)->iv = v;
            else if (!strcmp(f[1], "total_genome_kmers")) dict_set(&g_total, sample)->iv = v;
            else if (!strcmp(f[1], "total_genome_informative_kmers")) dict_set(&g_inf, sample)->iv = v;
        }
    }
    gzclose(L.f); free(L.buf);
    for (size_t i = 0; i < kmer_eval.n; i++) {       /* samples with a trailer but no passing hit (:121-124) */
        ditem *dd = dict_set(&depth, kmer_eval.it[i].key);
        if (!dd->iv) { dict_set(&cover, kmer_eval.it[i].key)->iv = 0; dd->iv = 0; }
    }

    /* main (:200-271) */
    char *strain = strdup(base_name(kfile));
    size_t sl = strlen(strain);
    if (sl >= 13 && !strncmp(strain + sl - 12, "kmer_hits", 9) && !strcmp(strain + sl - 2, "gz"))
        strain[sl - 13] = 0;                         /* regex ".kmer_hits.gz$": each '.' is any one character */
    char *species = strdup(strain), *genus = strdup(strain);
    char *u1 = strchr(genus, '_');
    if (u1) {
        *u1 = 0;
        char *u2 = strchr(species + (u1 - genus) + 1, '_');
        if (u2) *u2 = 0;
    }
    dict bg; dict_init(&bg);
    if (bfile) {
        FILE *f = fopen(bfile, "r");
        if (!f) { fprintf(stderr, "FileNotFoundError: [Errno 2] No such file or directory: '%s'\n", bfile); return 1; }
        char *ln = NULL; size_t cap = 0; ssize_t n;
        while ((n = getline(&ln, &cap, f)) >= 0) {
            while (n > 0 && ln[n - 1] == '\n') ln[--n] = 0;
            dict_set(&bg, ln);
        }
        free(ln); fclose(f);
    }
    printf("strain_name\tspecies_name\tgenus_name\tgenome_num_total_kmers\tgenome_num_informative_kmers\tmetagenome\t"
           "num_metagenomic_reads\tnum_metagenome_kmers\tunique_observed_informative_kmers\ttotal_observed_informative_kmers\t"
           "kmer_coverage\tkmer_depth\tkmer_depth_per_20B_kmer\tbackground\n");
    for (size_t i = 0; i < depth.n; i++) {
        const char *m = depth.it[i].key;
        const ditem *t;
        int64_t observed = depth.it[i].iv;
        int64_t unique = (t = dict_find(&cover, m)) ? t->iv : -1;
        int have_eval = (t = dict_find(&kmer_eval, m)) != NULL;
        int64_t evaluated = have_eval ? t->iv : -1;
        int64_t reads = have_eval ? ((t = dict_find(&read_eval, m)) ? t->iv : 0) : -1;
        int64_t gt = (t = dict_find(&g_total, m)) ? t->iv : -1;
        int64_t gi = (t = dict_find(&g_inf, m)) ? t->iv : -1;
        if (gi == 0) { fflush(stdout); fprintf(stderr, "ZeroDivisionError: float division by zero\n"); return 1; }
        double coverage = (double)unique / (double)gi, dep = (double)observed / (double)gi;
        char rc_[64], rd[64], rs[64];
        py_float_repr(coverage, rc_);
        py_float_repr(dep, rd);
        if (evaluated == 0) strcpy(rs, "0");
        else py_float_repr(dep * (2000000000 / (double)evaluated), rs);
        printf("%s\t%s\t%s\t%lld\t%lld\t%s\t%lld\t%lld\t%lld\t%lld\t%s\t%s\t%s\t%d\n", strain, species, genus, (long long)gt,
               (long long)gi, m, (long long)reads, (long long)evaluated, (long long)unique, (long long)observed, rc_, rd, rs,
               dict_find(&bg, m) ? 1 : 0);
    }
    return 0;
}

#ifdef KSF_MAIN
int main(int argc, char **argv) { return ksf_main(argc, argv); }
#endif
#ifdef KCD_MAIN
int main(int argc, char **argv) { return kcd_main(argc, argv); }
#endif
