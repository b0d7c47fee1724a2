/* sk_host_filter.c -- host side of the scrub filter: the drop-in for reference
 * scripts/kmer_scrub_filter.py (step 2 of test/example.sh).
 *
 * The script keeps four Python dictionaries keyed by the k-mer text (strain, pangenome, metagenome,
 * drug).  Here every distinct key gets a number (gid) in one hash table; the dictionaries become
 * columns indexed by gid; the rows handed to the device are "the strain dictionary of the last file, in
 * its insertion order" followed by the keys only other files had (those still weigh in the column sums
 * and the threshold walk, as their dictionary entries do in the script, but are never in the result).
 * Ranking, sums and histograms run on the device (sk_filter.hip); this file parses, decides how many
 * rows go (the script's float arithmetic, :122-130) and prints.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <zlib.h>
#include "../../include/strainer_kmer.h"
#include "sk_pyfmt.h"
#include "sk_gzfast.h"

/* ------------------------------------------------------------------ distinct keys */
typedef struct {
    char     *arena;                  /* key texts, NUL-terminated, back to back */
    size_t    arena_len, arena_cap;
    uint64_t *off;                    /* per gid: offset into arena */
    int64_t  *pan, *meta;             /* sums of the positive fields over all files and lines (:186-189) */
    uint8_t  *drug;                   /* some line had a positive 5th field (:191-194) */
    int32_t  *cur_file, *prev_file;   /* last file (and the one before) in which the key was a row */
    int64_t  *cur_val, *prev_val;     /* its reference_count there (strain_hash[key], :183) */
    uint32_t  n, cap;
    uint32_t *slot;                   /* gid + 1, 0 = empty */
    uint64_t  nslot;
} keydict;

static uint64_t key_hash(const char *s, size_t len)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ len, w;
    while (len >= 8) { memcpy(&w, s, 8); h = (h ^ w) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; s += 8; len -= 8; }
    if (len) { w = 0; memcpy(&w, s, len); h = (h ^ w) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 29; }
    return h ^ (h >> 31);
}

/* hint_keys: about how many distinct keys to expect (0 = unknown); the slot array starts big enough for them
 * instead of getting there by repeated rehashing */
static int kd_init(keydict *d, uint64_t hint_keys)
{
    memset(d, 0, sizeof *d);
    d->nslot = 1u << 16;
    while (d->nslot < 2 * hint_keys && d->nslot < (1ull << 31)) d->nslot <<= 1;
    d->slot = calloc(d->nslot, sizeof *d->slot);
    return d->slot ? 0 : -1;
}

static void kd_free(keydict *d)
{
    free(d->arena); free(d->off); free(d->pan); free(d->meta); free(d->drug);
    free(d->cur_file); free(d->prev_file); free(d->cur_val); free(d->prev_val); free(d->slot);
    memset(d, 0, sizeof *d);
}

static int kd_grow_rows(keydict *d)
{
    const uint32_t cap = d->cap ? d->cap * 2 : 1u << 16;
#define GROW(field) do { void *p_ = realloc(d->field, (size_t)cap * sizeof *d->field); if (!p_) return -1; d->field = p_; } while (0)
    GROW(off); GROW(pan); GROW(meta); GROW(drug); GROW(cur_file); GROW(prev_file); GROW(cur_val); GROW(prev_val);
#undef GROW
    d->cap = cap;
    return 0;
}

static int kd_grow_slots(keydict *d)
{
    uint64_t ns = d->nslot * 4, i;
    uint32_t *s = calloc(ns, sizeof *s), g;
    if (!s) return -1;
    for (g = 0; g < d->n; g++) {
        const char *k = d->arena + d->off[g];
        i = key_hash(k, strlen(k)) & (ns - 1);
        while (s[i]) i = (i + 1) & (ns - 1);
        s[i] = g + 1;
    }
    free(d->slot);
    d->slot = s; d->nslot = ns;
    return 0;
}

/* gid of the key, adding it when new; -1 = out of memory */
static int64_t kd_get(keydict *d, const char *key, size_t len)
{
    uint64_t i = key_hash(key, len) & (d->nslot - 1);
    uint32_t g;
    while ((g = d->slot[i]) != 0) {
        const char *k = d->arena + d->off[g - 1];
        if (!memcmp(k, key, len) && k[len] == 0) return g - 1;
        i = (i + 1) & (d->nslot - 1);
    }
    if (d->n == UINT32_MAX - 1) return -1;
    if (d->n == d->cap && kd_grow_rows(d)) return -1;
    if (d->arena_len + len + 1 > d->arena_cap) {
        size_t cap = d->arena_cap ? d->arena_cap * 2 : 1u << 20;
        char *a;
        while (cap < d->arena_len + len + 1) cap *= 2;
        if (!(a = realloc(d->arena, cap))) return -1;
        d->arena = a; d->arena_cap = cap;
    }
    g = d->n++;
    d->off[g] = d->arena_len;
    memcpy(d->arena + d->arena_len, key, len);
    d->arena[d->arena_len + len] = 0;
    d->arena_len += len + 1;
    d->pan[g] = d->meta[g] = 0; d->drug[g] = 0;
    d->cur_file[g] = d->prev_file[g] = -1; d->cur_val[g] = d->prev_val[g] = 0;
    d->slot[i] = g + 1;
    if ((uint64_t)d->n * 2 > d->nslot && kd_grow_slots(d)) return -1;
    return g;
}

/* ------------------------------------------------------------------ reading the tables */
typedef struct {
    keydict   keys;
    uint32_t *order;                  /* gids of the current file's strain dictionary, insertion order */
    uint32_t  norder, order_cap;
    int64_t   all_kmers;              /* data lines of the current file (:181) */
    int       drug_filter;            /* some line had exactly five fields (:190) */
} tables;

enum { FLT_OK = 0, FLT_OPEN, FLT_FIELDS, FLT_INT, FLT_NOMEM, FLT_MISMATCH };

/* one data line (no newline).  Python: content = line.rstrip('\n').split('\t') (:179-195) */
static int take_line(tables *t, int file_idx, char *s, size_t len)
{
    char *f[6], *end = s + len, *p;
    int nf = 1;
    int64_t c1, c2, c3, c4, g;
    keydict *d = &t->keys;
    f[0] = s;
    for (p = s; p < end; p++)
        if (*p == '\t') { if (nf < 6) f[nf] = p + 1; nf++; }
    if (nf < 4) return FLT_FIELDS;
#define FIELD_END(i) ((i) + 1 < nf && (i) + 1 < 6 ? f[(i) + 1] - 1 : end)
    if (!skp_int(f[1], FIELD_END(1), &c1) || !skp_int(f[2], FIELD_END(2), &c2) || !skp_int(f[3], FIELD_END(3), &c3)) return FLT_INT;
    if ((g = kd_get(d, f[0], (size_t)(f[1] - 1 - f[0]))) < 0) return FLT_NOMEM;
    t->all_kmers++;
    if (d->cur_file[g] != file_idx) {
        if (t->norder == t->order_cap) {
            uint32_t cap = t->order_cap ? t->order_cap * 2 : 1u << 16, *o = realloc(t->order, (size_t)cap * sizeof *o);
            if (!o) return FLT_NOMEM;
            t->order = o; t->order_cap = cap;
        }
        t->order[t->norder++] = (uint32_t)g;
        d->cur_file[g] = file_idx;
    }
    d->cur_val[g] = c1;
    if (c2 > 0) d->pan[g] += c2;
    if (c3 > 0) d->meta[g] += c3;
    if (nf == 5) {
        t->drug_filter = 1;
        if (!skp_int(f[4], end, &c4)) return FLT_INT;
        if (c4 > 0) d->drug[g] = 1;
    }
#undef FIELD_END
    return FLT_OK;
}

/* decompressed text arrives in pieces of any size (the library's own inflate hands out a few MB at a time,
 * gzread whatever fits): complete lines are taken as they come, the tail waits for the next piece */
typedef struct { tables *t; int file_idx, rc; char *carry; size_t have, cap; } line_feed;

static int feed_lines(void *user, const unsigned char *data, size_t n)
{
    line_feed *f = (line_feed *)user;
    const char *p = (const char *)data, *end = p + n, *nl;
    if (f->have) {                                    /* finish the line left over from the previous piece */
        nl = memchr(p, '\n', n);
        {
            const size_t add = nl ? (size_t)(nl - p) : n;
            if (f->have + add + 1 > f->cap) {
                size_t cap = f->cap ? f->cap : 4096;
                char *c;
                while (cap < f->have + add + 1) cap *= 2;
                if (!(c = realloc(f->carry, cap))) { f->rc = FLT_NOMEM; return 1; }
                f->carry = c; f->cap = cap;
            }
            memcpy(f->carry + f->have, p, add);
            f->have += add;
        }
        if (!nl) return 0;
        {
            size_t len = f->have;
            if (len && f->carry[len - 1] == '\r') len--;          /* text mode reads "\r\n" as one newline */
            if ((f->carry[0] != '#' || len == 0) && (f->rc = take_line(f->t, f->file_idx, f->carry, len)) != FLT_OK) return 1;
        }
        f->have = 0;
        p = nl + 1;
    }
    for (;;) {                                        /* lines in batches: the dictionary slots of a batch are
                                                       * prefetched before its lines are taken (every new key is a
                                                       * cache miss in a table of tens of MB otherwise) */
        enum { BATCH = 32 };
        const char *lp[BATCH];
        size_t ll[BATCH];
        int nb = 0, i;
        while (nb < BATCH && (nl = memchr(p, '\n', (size_t)(end - p))) != NULL) {
            size_t len = (size_t)(nl - p);
            if (len && p[len - 1] == '\r') len--;
            if (p[0] != '#' || len == 0) {
                const char *tab = memchr(p, '\t', len);
                const keydict *d = &f->t->keys;
                lp[nb] = p; ll[nb] = len; nb++;
                if (tab) __builtin_prefetch(&d->slot[key_hash(p, (size_t)(tab - p)) & (d->nslot - 1)]);
            }
            p = nl + 1;
        }
        for (i = 0; i < nb; i++)
            if ((f->rc = take_line(f->t, f->file_idx, (char *)lp[i], ll[i])) != FLT_OK) return 1;
        if (nb < BATCH) break;
    }
    if (p < end) {                                    /* keep the unfinished line */
        const size_t add = (size_t)(end - p);
        if (add + 1 > f->cap) {
            size_t cap = f->cap ? f->cap : 4096;
            char *c;
            while (cap < add + 1) cap *= 2;
            if (!(c = realloc(f->carry, cap))) { f->rc = FLT_NOMEM; return 1; }
            f->carry = c; f->cap = cap;
        }
        memcpy(f->carry, p, add);
        f->have = add;
    }
    return 0;
}

static int read_table(tables *t, int file_idx, const char *path)
{
    line_feed f;
    keydict *d = &t->keys;
    int zrc;
    memset(&f, 0, sizeof f);
    f.t = t; f.file_idx = file_idx; f.rc = FLT_OK;
    if (file_idx > 0 && d->n) {       /* the previous file's strain dictionary, for the identity check (:169-170,199-201) */
        memcpy(d->prev_file, d->cur_file, (size_t)d->n * sizeof *d->cur_file);
        memcpy(d->prev_val, d->cur_val, (size_t)d->n * sizeof *d->cur_val);
    }
    t->norder = 0;
    t->all_kmers = 0;
    zrc = getenv("SK_ZLIB") ? SKZ_NOT_GZIP : skz_decode_file(path, feed_lines, &f);     /* own inflate; zlib for anything that is not gzip */
    if (zrc == SKZ_OPEN) { free(f.carry); return FLT_OPEN; }
    if (zrc == SKZ_NOT_GZIP) {
        gzFile gz = gzopen(path, "rb");
        unsigned char *buf;
        int n;
        if (!gz) { free(f.carry); return FLT_OPEN; }
        gzbuffer(gz, 1u << 20);
        if (!(buf = malloc(1u << 20))) { gzclose(gz); free(f.carry); return FLT_NOMEM; }
        while ((n = gzread(gz, buf, 1u << 20)) > 0)
            if (feed_lines(&f, buf, (size_t)n)) break;
        free(buf);
        gzclose(gz);
    }
    if (f.rc == FLT_OK && f.have && f.carry[0] != '#')         /* last line without a newline */
        f.rc = take_line(t, file_idx, f.carry, f.have);
    free(f.carry);
    return f.rc;
}

/* strain_hash != previous_strain_hash (:199-201): both dictionaries hold the same keys with the same values */
static int same_strain_dict(const tables *t, int file_idx)
{
    const keydict *d = &t->keys;
    uint32_t g, nprev = 0;
    for (g = 0; g < d->n; g++) {
        const int in_cur = d->cur_file[g] == file_idx, in_prev = d->prev_file[g] == file_idx - 1;
        nprev += (uint32_t)in_prev;
        if (in_cur && (!in_prev || d->prev_val[g] != d->cur_val[g])) return 0;
    }
    return nprev == t->norder;
}

/* ------------------------------------------------------------------ the decision, shared by both routes */
typedef const char *(*key_fn)(void *user, uint64_t row, char tmp[32]);

/* scrub_max_kmers (:30-58): the smallest threshold t with 1 - #{v > t}/total >= min_frac, every step
 * reported on stderr exactly as the script words it */
static int walk_threshold(sk_filter *f, int which, double min_frac, double total, FILE *err, int64_t *thr_out)
{
    enum { W = 65536 };
    uint64_t *hist = malloc((W + 1) * sizeof *hist), hits = 0;
    int64_t thr = 0, lo = 1;
    char r[32];
    int rc, b;
    if (!hist) return SK_E_NOMEM;
    if ((rc = sk_filter_hist(f, which, lo, W, hist)) != SK_OK) { free(hist); return rc; }
    for (b = 0; b <= W; b++) hits += hist[b];
    for (;;) {
        const double kept = 1 - ((double)hits / total);
        skp_float_str(kept, r);
        fprintf(err, "kept %s with threshold %lld\n", r, (long long)thr);
        if (!(kept < min_frac)) break;
        thr++;
        if (thr >= lo + W) {
            lo = thr;
            if ((rc = sk_filter_hist(f, which, lo, W, hist)) != SK_OK) { free(hist); return rc; }
        }
        hits -= hist[thr - lo];
    }
    skp_float_str(total, r);
    fprintf(err, "threshold was %lld left with %llu out of %s that will be scrubbed\n", (long long)thr, (unsigned long long)hits, r);
    free(hist);
    *thr_out = thr;
    return SK_OK;
}

/* Everything after the tables are read (:204-231).  `n` rows are on the device, the first `n_strain` of
 * them are the strain dictionary in order; `n_drug_keys` = len(drug_genome_hash). */
static int decide_and_print(sk_ctx *ctx, sk_filter *f, uint64_t n, uint64_t n_strain, int64_t all_kmers, int drug_filter,
                            uint64_t n_drug_keys, uint64_t n_strain_gone, double min_fraction, int independent,
                            key_fn key_of, void *user, FILE *out, FILE *err)
{
    int64_t pan_sum, meta_sum, drug_scrubbed = 0;
    uint64_t n_pan, n_meta, n_alive = n_strain, left = 0, r;
    uint8_t *gone_or_scrubbed = NULL;
    char fr[32], tmp[32];
    int rc;
    if ((rc = sk_filter_sums(f, &pan_sum, &meta_sum, &n_pan, &n_meta, NULL)) != SK_OK) goto dev_fail;
    fprintf(out, "#total kmers in strain:%lld,%llu pangenome: %llu metagenome: %llu\n", (long long)all_kmers,
            (unsigned long long)n_strain, (unsigned long long)n_pan, (unsigned long long)n_meta);
    if (drug_filter) {                                /* :207-218 */
        double frac;
        n_alive = n_strain - n_strain_gone;
        fprintf(out, "#total kmers cross drug:%llu\n", (unsigned long long)n_drug_keys);
        if (all_kmers == 0) { fflush(out); fprintf(err, "ZeroDivisionError: float division by zero\n"); return 1; }
        frac = (double)n_alive / (double)all_kmers;
        drug_scrubbed = all_kmers - (int64_t)n_alive;
        skp_float_str(frac, fr);
        fprintf(out, "#fraction kmers remaining drug post scrub:%s\n#drug_scrubbed kmers:%lld\n", fr, (long long)drug_scrubbed);
        if (frac < min_fraction * 2) {
            fflush(out);
            fprintf(err, "Exception: ERROR: too few kmers remain after drug scrub. Are your drug strains too similar?\n");
            return 1;
        }
    }
    if (n && !(gone_or_scrubbed = malloc(n))) { fprintf(err, "kmer_scrub_filter: out of memory\n"); return 1; }
    if (independent) {                                /* :72-84 */
        int64_t tp, tm;
        if (all_kmers == 0) { fflush(out); fprintf(err, "ZeroDivisionError: float division by zero\n"); free(gone_or_scrubbed); return 1; }
        if ((rc = walk_threshold(f, 0, min_fraction, (double)all_kmers, err, &tp)) != SK_OK) goto dev_fail;
        if ((rc = walk_threshold(f, 1, min_fraction, (double)all_kmers, err, &tm)) != SK_OK) goto dev_fail;
        if ((rc = sk_filter_above(f, tp, tm, gone_or_scrubbed)) != SK_OK) goto dev_fail;
    } else {                                          /* :87-137: rows go, most frequent first, while more than min_fraction would remain */
        double num_scrubbed = (double)drug_scrubbed;
        uint64_t n_scrub = 0;
        while (n_scrub < n_alive && (1 - ((num_scrubbed + 1) / (double)all_kmers)) > min_fraction) { num_scrubbed += 1.0; n_scrub++; }
        if ((rc = sk_filter_joint(f, pan_sum, meta_sum, n_scrub, gone_or_scrubbed)) != SK_OK) goto dev_fail;
    }
    for (r = 0; r < n_strain; r++) left += !gone_or_scrubbed[r];
    fprintf(out, "#post scrub kmers %llu out of %lld\n", (unsigned long long)left, (long long)all_kmers);
    for (r = 0; r < n_strain; r++)
        if (!gone_or_scrubbed[r]) { fputs(key_of(user, r, tmp), out); fputc('\n', out); }
    free(gone_or_scrubbed);
    return 0;
dev_fail:
    free(gone_or_scrubbed);
    fprintf(err, "kmer_scrub_filter: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx));
    return 1;
}

/* ------------------------------------------------------------------ route 1: the script's command line */
typedef struct { const keydict *d; const uint32_t *rows; } file_keys;
static const char *file_key_of(void *user, uint64_t row, char tmp[32])
{
    const file_keys *k = user;
    (void)tmp;
    return k->d->arena + k->d->off[k->rows[row]];
}

static int opt_value(int argc, char **argv, int *i, const char *shortn, const char *longn, const char **val, FILE *err, int *bad)
{
    const char *a = argv[*i];
    const size_t ll = strlen(longn);
    if (!strcmp(a, shortn) || !strcmp(a, longn)) {
        if (*i + 1 >= argc) { fprintf(err, "kmer_scrub_filter: error: argument %s/%s: expected one argument\n", longn, shortn); *bad = 1; return 1; }
        *val = argv[++*i];
        return 1;
    }
    if (!strncmp(a, longn, ll) && a[ll] == '=') { *val = a + ll + 1; return 1; }
    if (!strncmp(a, shortn, 2) && a[2]) { *val = a + 2; return 1; }          /* -m0.01 */
    return 0;
}

int skh_scrub_filter_main(int argc, char **argv, FILE *out, FILE *err)
{
    const char *sfile = NULL, *lfile = NULL, *mtext = NULL, *env;
    double min_fraction = 0.04;
    int independent = 0, i, bad = 0, status = 1, rc, device = 0;
    char **files = NULL;
    size_t nfiles = 0, fi;
    tables t;
    sk_ctx *ctx = NULL;
    sk_filter *flt = NULL;
    int64_t *pan = NULL, *meta = NULL;
    uint8_t *gone = NULL;
    uint32_t *rows = NULL;

    for (i = 1; i < argc; i++) {                     /* argparse surface of the script (:14-28) */
        if (!strcmp(argv[i], "-i") || !strcmp(argv[i], "--independent")) { independent = 1; continue; }
        if (!strcmp(argv[i], "-h") || !strcmp(argv[i], "--help")) {
            fprintf(out, "usage: kmer_scrub_filter [-h] [--scrub_count_file FILE | --scrub_count_list LIST] [--min_fraction M] [--independent]\n");
            return 0;
        }
        if (opt_value(argc, argv, &i, "-s", "--scrub_count_file", &sfile, err, &bad)) { if (bad) return 2; continue; }
        if (opt_value(argc, argv, &i, "-l", "--scrub_count_list", &lfile, err, &bad)) { if (bad) return 2; continue; }
        if (opt_value(argc, argv, &i, "-m", "--min_fraction", &mtext, err, &bad)) { if (bad) return 2; continue; }
        fprintf(err, "kmer_scrub_filter: error: unrecognized arguments: %s\n", argv[i]);
        return 2;
    }
    if (mtext) {
        char *e;
        min_fraction = strtod(mtext, &e);
        if (e == mtext || *e) { fprintf(err, "kmer_scrub_filter: error: argument --min_fraction/-m: invalid float value: '%s'\n", mtext); return 2; }
    }
    if (min_fraction < 0.0 || min_fraction > 1.0) {  /* the script dies while wording this complaint (:142-143) */
        fprintf(err, "error --min_fraction (-m) must be between 0.0 and 1.0 (%s)\n", mtext);
        return 1;
    }
    if (!sfile && !lfile) fputs("error: one of scrub_count_file or scrub_count_list must be provided.", err);
    if (sfile && lfile) fputs("error: can provide only one of either scrub_count_file or scrub_count_list.", err);

    if (sfile) {
        if (!(files = malloc(sizeof *files)) || !(files[0] = strdup(sfile))) return 1;
        nfiles = 1;
    } else if (lfile) {                              /* one path per line, trailing blanks dropped (:152-155) */
        FILE *lf = fopen(lfile, "r");
        char *line = NULL;
        size_t cap = 0;
        ssize_t n;
        if (!lf) { fprintf(err, "kmer_scrub_filter: could not open list %s\n", lfile); return 1; }
        while ((n = getline(&line, &cap, lf)) >= 0) {
            char **nf;
            while (n > 0 && isspace((unsigned char)line[n - 1])) line[--n] = 0;
            if (!(nf = realloc(files, (nfiles + 1) * sizeof *files))) { fclose(lf); return 1; }
            files = nf;
            files[nfiles++] = strdup(line);
        }
        free(line);
        fclose(lf);
    }

    memset(&t, 0, sizeof t);
    {   /* a table line is ~38 bytes and gzip shrinks the tables about nine-fold */
        struct stat st;
        uint64_t hint = 0;
        if (nfiles && stat(files[0], &st) == 0 && S_ISREG(st.st_mode)) hint = (uint64_t)st.st_size * 9u / 38u;
        if (hint > (1ull << 28)) hint = 1ull << 28;
        if (kd_init(&t.keys, hint)) { fprintf(err, "kmer_scrub_filter: out of memory\n"); return 1; }
    }
    for (fi = 0; fi < nfiles; fi++) {                /* :164-201 */
        rc = read_table(&t, (int)fi, files[fi]);
        if (rc == FLT_OPEN) { fprintf(err, "kmer_scrub_filter: could not read %s\n", files[fi]); goto done; }
        if (rc == FLT_FIELDS) { fprintf(err, "kmer_scrub_filter: %s: a line has fewer than four tab-separated fields\n", files[fi]); goto done; }
        if (rc == FLT_INT) { fprintf(err, "kmer_scrub_filter: %s: a count field is not an integer\n", files[fi]); goto done; }
        if (rc == FLT_NOMEM) { fprintf(err, "kmer_scrub_filter: out of memory\n"); goto done; }
        if (fi > 1 && !same_strain_dict(&t, (int)fi)) {
            fputs("error: input files do not have identical hash and strain hash values.\n", err);
            goto done;
        }
    }

    {   /* rows for the device: the strain dictionary in order, then the keys only other files had */
        const keydict *d = &t.keys;
        const uint64_t n = d->n;
        const int last = (int)nfiles - 1;
        uint64_t r = t.norder, n_drug = 0, n_strain_gone = 0;
        uint32_t g;
        file_keys fk;
        if (!(rows = malloc((n + 1) * sizeof *rows)) || !(pan = malloc((n + 1) * sizeof *pan)) ||
            !(meta = malloc((n + 1) * sizeof *meta)) || !(gone = malloc(n + 1))) { fprintf(err, "kmer_scrub_filter: out of memory\n"); goto done; }
        if (t.norder) memcpy(rows, t.order, (size_t)t.norder * sizeof *rows);
        for (g = 0; g < d->n; g++) {
            if (nfiles == 0 || d->cur_file[g] != last) rows[r++] = g;
            n_drug += d->drug[g];
        }
        for (r = 0; r < n; r++) {
            g = rows[r];
            pan[r] = d->pan[g]; meta[r] = d->meta[g];
            gone[r] = (uint8_t)(r >= t.norder || (t.drug_filter && d->drug[g]));
            if (r < t.norder) n_strain_gone += gone[r];
        }
        if ((env = getenv("SK_DEVICE")) != NULL) device = atoi(env);
        rc = sk_ctx_create(&ctx, device);
        if (rc != SK_OK) { fprintf(err, "kmer_scrub_filter: cannot use HIP device %d: %s\n", device, sk_strerror(rc)); goto done; }
        if ((rc = sk_filter_create(ctx, &flt)) != SK_OK || (rc = sk_filter_load(flt, pan, meta, gone, n)) != SK_OK) {
            fprintf(err, "kmer_scrub_filter: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx));
            goto done;
        }
        fk.d = d; fk.rows = rows;
        status = decide_and_print(ctx, flt, n, t.norder, t.all_kmers, t.drug_filter, n_drug, n_strain_gone, min_fraction, independent,
                                  file_key_of, &fk, out, err);
    }
done:
    fflush(out);
    sk_filter_destroy(flt);
    sk_ctx_destroy(ctx);
    free(rows); free(pan); free(meta); free(gone); free(t.order);
    kd_free(&t.keys);
    for (fi = 0; fi < nfiles; fi++) free(files[fi]);
    free(files);
    return status;
}

/* ------------------------------------------------------------------ route 2: straight from the scan's counters */
static const char *keyset_key_of(void *user, uint64_t row, char tmp[32])
{
    skh_keyset_key((const skh_keyset *)user, (uint32_t)row, tmp);
    return tmp;
}

int skh_scrub_filter_resident(sk_ctx *ctx, const skh_keyset *ks, int with_drug_column, double min_fraction,
                              int independent, FILE *out, FILE *err)
{
    sk_filter *flt = NULL;
    uint64_t n_gone = 0;
    int rc, status;
    if (!ctx || !ks || min_fraction < 0.0 || min_fraction > 1.0) return 1;
    if ((rc = sk_filter_create(ctx, &flt)) != SK_OK || (rc = sk_filter_load_counts(flt, 1, 2, with_drug_column ? 3 : -1)) != SK_OK ||
        (rc = sk_filter_sums(flt, NULL, NULL, NULL, NULL, &n_gone)) != SK_OK) {
        fprintf(err, "kmer_scrub_filter: %s (%s)\n", sk_strerror(rc), sk_last_error(ctx));
        sk_filter_destroy(flt);
        return 1;
    }
    /* a printed table has one line per row and no repeated key: all_kmers = len(strain_hash) = rows;
     * len(drug_genome_hash) = rows with a positive drug count */
    status = decide_and_print(ctx, flt, ks->nrows, ks->nrows, (int64_t)ks->nrows, with_drug_column, n_gone, n_gone, min_fraction,
                              independent, keyset_key_of, (void *)ks, out, err);
    fflush(out);
    sk_filter_destroy(flt);
    return status;
}
