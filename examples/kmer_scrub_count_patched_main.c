/* examples/kmer_scrub_count_patched_main.c -- INTEGRATION.md section 2 as a complete program: the shape of
 * the reference's own main() (src/kmer_scrub_count.c:29-124) with its five hot calls replaced by calls
 * into libstrainer_kmer.  Each replaced reference line is quoted above its replacement.
 *
 *   gcc -O2 -Iinclude examples/kmer_scrub_count_patched_main.c -Lstrainer2_amd/lib -lstrainer_kmer \
 *       -Wl,-rpath,$PWD/strainer2_amd/lib -o /tmp/ksc_patched
 *
 * tests/test_examples.py builds it and compares its output with the reference's golden output. */
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include "strainer_kmer.h"

static void usage(void)
{
    fprintf(stderr, "Usage: kmer_scrub_count -r <reference genome>  -A <file with multiple genome filenames> "
                    "-B <file with multiple metagenome filenames> -C <(optional) file with multiple genome "
                    "filenames of drug strains> -p [progress output file, optional]\n");
}

int main(int argc, char *argv[])
{
    char *A_file = NULL, *B_file = NULL, *C_file = NULL, *r_file = NULL, *progress_file = NULL;
    FILE *progress = NULL;
    skh_keyset ks;
    sk_ctx *ctx;
    int c;

    while ((c = getopt(argc, argv, "A:B:C:r:p:Hhud")) != EOF)
        switch (c) {
        case 'A': A_file = optarg; break;
        case 'B': B_file = optarg; break;
        case 'C': C_file = optarg; break;
        case 'r': r_file = optarg; break;
        case 'p': progress_file = optarg; break;
        case 'd': break;
        default: usage(); break;
        }
    if (!r_file || !A_file || !B_file) { usage(); exit(1); }
    if (progress_file) {
        progress = fopen(progress_file, "w");
        if (!progress) { fprintf(stderr, "could not open progress file %s\n", progress_file); exit(1); }
        fprintf(progress, "adding kmer counts for:\n");
    }

    /* seqHash = BIO_initHash(DEFAULT_GENOME_HASH_SIZE);                                    :87 */
    /* GEN_hash_sequences_set_count_vec(r_file, seed, seqHash, 1, 1, 0, 4);                 :89 */
    if (skh_keyset_from_file(&ks, r_file, SK_REF_TABLE_SLOTS, 1, 1) == SK_E_OPEN) {
        fprintf(stderr, "could not read file %s GEN_hash_sequences_set_count_vec()\n", r_file);
        exit(EXIT_FAILURE);
    }
    if (sk_ctx_create(&ctx, 0) != SK_OK) { fprintf(stderr, "no usable HIP device\n"); exit(EXIT_FAILURE); }
    if (skh_keyset_load(ctx, &ks, 4) != SK_OK) { fprintf(stderr, "table load failed: %s\n", sk_last_error(ctx)); exit(EXIT_FAILURE); }

    /* GEN_all_kmer_counts(A_file, seed, seqHash, 1, progress);                             :90 */
    if (skh_scan_list(ctx, A_file, NULL, 1, progress, stderr, 0, 1, NULL)) exit(EXIT_FAILURE);
    /* GEN_all_kmer_counts(B_file, seed, seqHash, 2, progress);                             :91 */
    if (skh_scan_list(ctx, B_file, NULL, 2, progress, stderr, 0, 1, NULL)) exit(EXIT_FAILURE);
    /* if (C_file) GEN_all_kmer_counts_skip_file(C_file, r_file, seed, seqHash, 3, progress);   :93-94 */
    if (C_file && skh_scan_list(ctx, C_file, r_file, 3, progress, stderr, 0, 1, NULL)) exit(EXIT_FAILURE);

    /* print_hash_counts(seqHash, C_file);                                                  :98 */
    if (skh_print_counts(ctx, &ks, stdout, C_file != NULL) != SK_OK) exit(EXIT_FAILURE);

    /* BIO_destroyHashD(seqHash);                                                           :112 */
    sk_ctx_destroy(ctx);
    skh_keyset_free(&ks);
    if (progress) fclose(progress);
    return 0;
}
