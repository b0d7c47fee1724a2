/* kso_oracle.h -- CPU ORACLE for the k-mer scrub/count hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a from-scratch restatement, in plain C, of the algorithm the reference runs on
 * its kmer_scrub_count path (string-keyed table, djb2, linear probing, strcmp; see the
 * per-function citations in kso_oracle.c).  It exists to CHECK the HIP product path and to
 * serve as the timed "port" CPU baseline in bench.py.  Nothing under strainer2_amd/ may
 * include, link or call it: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks it byte-for-byte against
 *   (1) the unmodified reference binary built into oracle/_ref (when present), and
 *   (2) committed golden outputs produced by that binary (tests/golden/), including the
 *       md5 75989a9bc31ef0b6f53a5112a60920bd of the reference's bundled test/example.sh step 1.
 */
#ifndef KSO_ORACLE_H
#define KSO_ORACLE_H
#include <stddef.h>
#include <stdio.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSO_OK                 0
#define KSO_E_OPEN            -1   /* reference: "could not read file ..." + exit(1)        */
#define KSO_E_SHORT_CONTIG    -2   /* reference: size_t underflow -> SIGSEGV (exit 139)    */

#define KSO_DEFAULT_CAPACITY 8000000u   /* reference src/genome_compare.h:20 */

typedef struct kso_table kso_table;

kso_table *kso_table_new(unsigned capacity);
void       kso_table_free(kso_table *t);
unsigned   kso_table_size(const kso_table *t);       /* number of keys (N)          */
unsigned   kso_table_capacity(const kso_table *t);   /* number of slots (M)         */
int        kso_table_ncols(const kso_table *t);

/* a3: build/update the table from a FASTA/FASTQ(.gz) file.
 * short_policy: 0 = behave like the reference (stop and report KSO_E_SHORT_CONTIG when a
 * record shorter than k-1 is met: the reference crashes there), 1 = skip such records. */
int kso_build_from_file(kso_table *t, const char *path, int k, unsigned default_val,
                        unsigned incr, int idx, int ncols, int short_policy);

/* a1: scan one file, bumping column `col` for every window whose oriented form is a key. */
int kso_scan_file(kso_table *t, const char *path, int k, int col, uint64_t *bases_seen);

/* a2: walk a newline-separated list of files; `skip` (may be NULL) is compared with strcmp
 * against each line; progress (may be NULL) receives "<line>\t<asctime>" lines.
 * Messages go to `err` exactly as the reference words them.  Returns KSO_OK or KSO_E_OPEN
 * (with *failed_is_list telling which message the caller should expect was written). */
int kso_scan_list(kso_table *t, const char *list_path, const char *skip, int k, int col,
                  FILE *progress, FILE *err, uint64_t *bases_seen);

/* In-memory scan of already-decoded records separated by '\n' (each line = one record's
 * sequence).  Same per-record logic as kso_scan_file.  Used by bench.py's cpu_baseline and
 * by parity tests against the device ABI, which takes the same stream layout. */
void kso_scan_stream(kso_table *t, const char *stream, size_t len, int k, int col);

/* Same, but build-phase semantics (a3) over an in-memory stream. */
int kso_build_from_stream(kso_table *t, const char *stream, size_t len, int k,
                          unsigned default_val, unsigned incr, int idx, int ncols,
                          int short_policy);

/* Enumerate rows in ascending slot order (= the reference's output order).
 * keys_out: n * (k+1) bytes (NUL-terminated each); counts_out: n * ncols u32. */
void kso_table_rows(const kso_table *t, int k, char *keys_out, unsigned *counts_out);

/* a9: print the TSV exactly as the reference's print_hash_counts does. */
void kso_print(const kso_table *t, FILE *out, int with_drug_column);

/* Test helper: decode a FASTA/FASTQ(.gz) file with the reference parser's grammar and return
 * every record's sequence, each followed by '\n', in one malloc'd buffer (free with
 * kso_free).  *status = last parser return (-1 end of file, -2 truncated quality). */
char *kso_decode_file(const char *path, size_t *out_len, long *nrecords, int *status);
void  kso_free(void *p);

/* Whole-program restatement: argv as the reference's kmer_scrub_count. Returns exit status. */
int kso_main(int argc, char **argv, FILE *out, FILE *err);

/* Whole-program restatement of the reference's strain_detect (src/strain_detect.c): argv as the
 * reference, messages the reference prints on stdout go to `out`, stderr texts to `err`, the
 * gz result file is written with zlib level 9 like the reference.  Returns the exit status
 * (139 where the reference would crash on a NULL token). */
int ksd_main(int argc, char **argv, FILE *out, FILE *err);

#ifdef __cplusplus
}
#endif
#endif
