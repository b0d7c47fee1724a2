/* strainer_kmer.h -- C-ABI of libstrainer_kmer.so: the MI355X (gfx950) k-mer scrub/count path.
 *
 * This is the drop-in boundary.  The reference (jeremiahfaith/strainer2, a plain C program)
 * has no FFI: its seam for this path is the internal call chain
 *
 *     GEN_hash_sequences_set_count_vec()   src/genome_compare.c:967-1030   (build the strain table)
 *     GEN_all_kmer_counts[_skip_file]()    src/genome_compare.c:149-177,115-146 (walk a file list)
 *     GEN_calculate_kmer_count()           src/genome_compare.c:179-236     (THE hot loop)
 *     BIO_searchHash()/BIO_getHashKeys()   src/BIO_hash.c:161-172,174-188   (lookup / row order)
 *     print_hash_counts()                  src/kmer_scrub_count.c:134-156   (TSV)
 *     quantify_hits_PE()/hash_scrubbed_kmers()/background_filter()   src/strain_detect.c:387-663,668-726,160-240
 *
 * and, either side of it in the workflow (test/example.sh), the two scripts scripts/kmer_scrub_filter.py and
 * scripts/coverage_depth.py, which are covered here too (sk_filter_*, sk_distinct_count, skh_*_main).
 *
 * The entry points below replace those calls one for one (each cites what it replaces).
 * Plain pointers and sizes only; no C++ or torch types.  Two layers:
 *
 *   sk_*   device layer  -- context, table residency, batch scan, counters, collective.
 *   skh_*  host layer    -- the reference's file/list/record semantics in C (kseq grammar,
 *                           canonical 2-bit packing, BIO_hash slot-order replay, TSV print),
 *                           driving the device layer.  kmer_scrub_count's main() is ~100 lines
 *                           on top of it.
 *
 * There is NO CPU compute fallback: every window lookup runs in a HIP kernel.  Without a
 * usable GPU sk_ctx_create() fails with SK_E_NODEVICE and callers must stop.
 *
 * Stream layout ("record stream") used by the scan entry points: the sequence bytes of the
 * records, verbatim as decoded (any case), records separated by one '\n'.  Bytes that are not
 * A/C/G/T (any case) break windows; '\n', 'N'/'n' and NUL never take part in any window.
 *
 * Thread-safety: one caller per sk_ctx at a time; distinct contexts are independent.
 */
#ifndef STRAINER_KMER_H
#define STRAINER_KMER_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SK_K                 31            /* seed length: src/kmer_scrub_count.c:39, src/strain_detect.c:78 */
#define SK_REF_TABLE_SLOTS   8000000u      /* DEFAULT_GENOME_HASH_SIZE: src/genome_compare.h:20 */
#define SK_KEY_NONE          UINT64_MAX    /* "no packed key for this row" (row is a wide key) */
#define SK_ROWS_IN_STRAIN_ORDER UINT32_MAX /* skh_keyset_from_*: initial_slots value meaning "do not replay the reference's hash table: rows in
                                            * first-occurrence order along the strain" -- for callers whose output does not show the row order
                                            * (strain_detect: src/strain_detect.c prints k-mer text per hit, never walks the table) */
#define SK_LOCALITY_FWD      0x80000000u   /* locality[] flag: the key is the strain text itself at its first occurrence */

/* error codes (0 = success) */
#define SK_OK            0
#define SK_E_NODEVICE   -1   /* no HIP device / HIP runtime failure at init                  */
#define SK_E_HIP        -2   /* a HIP call failed; sk_last_error() has the text              */
#define SK_E_ARG        -3   /* bad argument                                                 */
#define SK_E_NOMEM      -4
#define SK_E_OPEN       -5   /* file could not be opened (host layer)                        */
#define SK_E_DUPKEY     -6   /* duplicate key passed to sk_table_load                        */
#define SK_E_STATE      -7   /* call out of order (e.g. scan before table load)              */
#define SK_E_RCCL       -8
#define SK_E_SPLIT      -9   /* a big text file could not be cut at record boundaries: the run fails rather than count something else */
#define SK_E_PLAN       -10  /* the ranks of a multi-GPU run computed different work plans for a list: every rank leaves */

typedef struct sk_ctx sk_ctx;

/* ----------------------------------------------------------------------------------------
 * device layer
 * -------------------------------------------------------------------------------------- */

/* Create a context on HIP device `device` (0-based).  Replaces BIO_initHash(): src/BIO_hash.c:14-37. */
int  sk_ctx_create(sk_ctx **out, int device);
/* Replaces BIO_destroyHashD(): src/BIO_hash.c:77-92. */
void sk_ctx_destroy(sk_ctx *ctx);
const char *sk_last_error(const sk_ctx *ctx);
const char *sk_strerror(int code);

/* Make a strain's key set resident in HBM and allocate ncols zeroed u32 counters per row.
 * keys[r] is the canonical 31-mer of row r packed 2 bits/base, A=0 C=1 G=2 T=3, first base in
 * the most significant position (bits 61..60), i.e. canonical = max(forward, revcomp) as
 * integers == the reference's orient_string() choice (src/genome_compare.c:1100-1141).
 * Rows whose key is not pure ACGT carry SK_KEY_NONE here and are supplied through
 * sk_table_load_wide().  Row order is the caller's (the host layer passes BIO_hash slot order).
 * Replaces the insert half of GEN_hash_sequences_set_count_vec(): src/genome_compare.c:1007-1019. */
int sk_table_load(sk_ctx *ctx, const uint64_t *keys, uint32_t nrows, uint32_t ncols);
/* Same, with a "locality" permutation: locality[r] = position of row r's key in an order in which
 * keys that follow each other in the strain are neighbours (the host layer passes first-occurrence
 * order), in the low 31 bits; bit 31 (SK_LOCALITY_FWD) says whether the key equals the strain text at
 * that occurrence or its reverse complement.  The device keeps its counters (and a copy of the keys)
 * in that order: the hits of one read land on adjacent counters (their atomics coalesce), and the
 * neighbours of a window that hit are first looked for next to it instead of through the hash.
 * Invisible through sk_counts_fetch/set and sk_tally_batch (they speak caller rows);
 * sk_counts_device_ptr/sk_counts_allreduce see the block in locality order, which is the same on
 * every rank that loaded the same key set. */
int sk_table_load_ex(sk_ctx *ctx, const uint64_t *keys, uint32_t nrows, uint32_t ncols, const uint32_t *locality);

/* Strain text for the scan's "seed and verify" stage (optional; after sk_table_load_ex).  text2 = the
 * strain's bases as the build phase read them, 2 bits each in the code of the keys, 16 bases per word with
 * the first base in the most significant position, records laid end to end, nbases in all (bytes that are
 * not A/C/G/T take any code).  first_pos[r] = text position of the first base of an all-ACGT window whose
 * canonical form is row r's key, or UINT32_MAX for rows without one (wide keys; keys that only occur
 * through a U).  Contract on the locality order given to sk_table_load_ex: the rows WITH a position come
 * first, in ascending position order (then the counter index of such a row is the rank of its position, which
 * the device recomputes from a bit per text position).  With the text resident, a window that hit the table
 * tells where the read lies on the strain; its neighbours are then compared with the strain's text at the
 * expected places (a 62-bit compare each, as exact as a table probe) instead of being probed one by one.
 * Replaces nothing in the reference: BIO_searchHash per window (src/BIO_hash.c:161-172) stays the meaning. */
int sk_table_load_text(sk_ctx *ctx, const uint32_t *text2, uint32_t nbases, const uint32_t *first_pos);
/* The whole table made ON THE DEVICE from the strain's text (new; strain_detect's opening): text2 = the bases, 2 bits each, records end
 * to end (as for sk_table_load_text); startok = one bit per position, set where a window of 31 A/C/G/T bases of one record starts
 * (nstarts of them) -- the windows src/genome_compare.c:1000-1019 makes keys of.  Rows are numbered by first occurrence along the
 * text; column 0 of every row = col0_value (src/strain_detect.c:139: default 1, increment 0); *nrows = distinct keys.  No byte-string
 * keys: a strain with U/IUPAC letters goes through skh_keyset_from_file + skh_keyset_load.  sk_table_export_keys: the keys in
 * row order, once.  Replaces GEN_hash_sequences_set_count_vec (src/genome_compare.c:967-1030) for strain_detect. */
int sk_table_build_from_text(sk_ctx *ctx, const uint32_t *text2, const uint32_t *startok, uint32_t nbases, uint32_t nstarts,
                             uint32_t ncols, uint32_t col0_value, uint32_t *nrows);
int sk_table_export_keys(sk_ctx *ctx, uint64_t *keys_out /* nrows */);
int sk_table_export_keys_of(sk_ctx *ctx, const uint32_t *rows, uint32_t n, uint64_t *keys_out /* n */);   /* ... of n chosen rows */

/* Wide keys: rows whose 31-byte upper-cased oriented key contains bytes other than ACGT
 * (IUPAC letters in the strain: SURVEY 8(a) a3/a6).  keys31 = nwide * 32 bytes, each key
 * 31 bytes + NUL; rows[i] = row index of key i. */
int sk_table_load_wide(sk_ctx *ctx, const char *keys31, const uint32_t *rows, uint32_t nwide);

/* Scan one batch of the record stream held in HOST memory: copy to HBM and count, both
 * asynchronous on the context's stream (the call returns once the batch is staged; the
 * caller may reuse `stream` immediately).  Adds 1 to counter column `col` of every row whose
 * key equals the canonical form of a window.  Replaces the window loop of
 * GEN_calculate_kmer_count(): src/genome_compare.c:213-229 + BIO_searchHash(). */
int sk_scan_stream(sk_ctx *ctx, const uint8_t *stream, uint64_t nbytes, uint32_t col);

/* Zero-copy variant for callers that fill pinned host buffers themselves (the host layer's decode
 * threads do): `pinned` comes from sk_pinned_alloc, holds at most 64 MiB - 64 bytes of record stream
 * made of WHOLE records or pieces cut with the k-1 overlap, and is DMA-read in place; it may be
 * rewritten once sk_ticket_wait(ctx, *ticket) has returned. */
int sk_pinned_alloc(sk_ctx *ctx, void **p, uint64_t nbytes);
int sk_pinned_free(sk_ctx *ctx, void *p);
int sk_scan_pinned(sk_ctx *ctx, const uint8_t *pinned, uint64_t nbytes, uint32_t col, uint64_t *ticket);
int sk_ticket_wait(sk_ctx *ctx, uint64_t ticket);

/* The host-side 2-bit pre-pack of a record stream (new; the reference moves bytes through memory only,
 * src/kseq.h:90-141).  sk_pack_stream makes, for every 16-byte chunk of `stream`, what the scan kernel's own first phase would
 * make of it -- a 32-bit word of sixteen 2-bit codes (A 0, C 1, G 2, T 3 in either case, first byte highest; 0 for any other byte)
 * and a 16-bit mask of the bytes that are no A/C/G/T -- into `packed` (sk_packed_bytes(nbytes) bytes: the code words, then the
 * masks): 6 bytes per 16 bases cross the PCIe link instead of 16.  *odd is set when the stream holds a byte that is neither
 * A/C/G/T, N/n nor '\n' (an IUPAC letter, U, '\r': only the exact byte-string kernel can judge its windows, a6 of SURVEY 8):
 * such a batch must go to sk_scan_pinned as bytes.  sk_scan_pinned_packed is sk_scan_pinned for a packed batch (`packed` in
 * memory from sk_pinned_alloc; nbytes = the length of the byte stream it was packed from); counts are the same, bit for bit. */
uint64_t sk_packed_bytes(uint64_t nbytes);
int sk_pack_stream(const uint8_t *stream, uint64_t nbytes, void *packed, int *odd);
int sk_scan_pinned_packed(sk_ctx *ctx, const void *packed, uint64_t nbytes, uint32_t col, uint64_t *ticket);
int sk_scan_device_packed(sk_ctx *ctx, const void *dev_packed, uint64_t nbytes, uint32_t col);   /* the packed batch already in device memory (4-byte aligned) */

/* Same, for a batch already resident in HBM (device pointer). */
int sk_scan_device(sk_ctx *ctx, const void *dev_stream, uint64_t nbytes, uint32_t col);

/* strain_detect's per-READ view of the same scan (src/strain_detect.c:443-541): one batch of the
 * record stream whose records start at rec_start[0..nrec) (byte offsets into `stream`, ascending;
 * a record ends at the next start).  For record r: out_tally[2r] = windows that hit any key,
 * out_tally[2r+1] = windows that hit a key whose counter in column `type_col` equals
 * `informative_value`; each of the latter is also logged as {window-end offset, row} in out_hits
 * (unordered; *out_nhits may exceed hits_cap: then only hits_cap entries were stored).  Synchronous. */
typedef struct sk_hit { uint32_t pos, row; } sk_hit;
int sk_tally_batch(sk_ctx *ctx, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec,
                   uint32_t type_col, uint32_t informative_value, uint32_t *out_tally /* 2*nrec */,
                   sk_hit *out_hits, uint64_t hits_cap, uint64_t *out_nhits);

/* The same in pieces, for tallying ONE batch against MANY tables (several strains resident on one
 * device, one context each): upload the batch once, launch on every context (each on its own HIP
 * stream, so the launches overlap), then collect.  A batch may be refilled only after every launch on
 * it has been collected; a context has at most one launch in flight.  New: the reference runs one
 * strain per process and re-reads the metagenome for each. */
typedef struct sk_batch sk_batch;
int  sk_batch_create(sk_ctx *ctx, sk_batch **out);          /* on ctx's device; errors are reported through ctx */
void sk_batch_destroy(sk_batch *b);
int  sk_batch_sync(sk_batch *b);                               /* its uploads are done: the host memory it was filled from may be reused, the batch kept for the next file */
int  sk_batch_fill(sk_batch *b, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec);
int  sk_batch_fill_packed(sk_batch *b, const void *packed, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec);   /* the batch in sk_pack_stream's form (nbytes, rec_start: of the byte stream it was packed from) */
int  sk_tally_launch(sk_ctx *ctx, const sk_batch *b, uint32_t type_col, uint32_t informative_value, uint64_t hits_cap);
int  sk_tally_collect(sk_ctx *ctx, uint32_t *out_tally /* 2*nrec */, sk_hit *out_hits /* hits_cap */, uint64_t *out_nhits);
/* The same collection, sparse: only records with at least one hit, compacted on the device, as {record, all hits,
 * informative hits} in no particular order (*n of them; at most cap are stored).  What comes back over PCIe is then
 * proportional to the reads that hit this strain, not to the reads of the batch (many strains x one metagenome:
 * src/strain_detect.c:443-626 keeps two counters per read and strain). */
typedef struct sk_tally_rec { uint32_t rec, all, inf; } sk_tally_rec;
int  sk_tally_collect_sparse(sk_ctx *ctx, sk_tally_rec *out, uint64_t cap, uint64_t *n, sk_hit *out_hits /* hits_cap */, uint64_t *out_nhits);

/* ONE table for several resident strains: a batch is then scanned once, not once per strain (BASELINE configs[4]: 32
 * strains per GPU against one metagenome).  `members`: 1..SK_UNION_MAX contexts on one device, each with its table, text
 * stage and type column as they are to be used (the union is a snapshot: build it after the -a/-g flags are final; the
 * members stay as they are and must outlive it).  SK_E_STATE if a member has byte-string keys or no text stage -- the
 * caller then tallies member by member (sk_tally_launch).
 * sk_union_tally_collect: out[i] = {record * n + member, windows that hit a key of that member, those whose key is
 * informative_value in the member's type_col} for the (record, member) pairs with at least one hit, unordered;
 * out_hits[j] = {window-end offset, member << SK_UNION_ROW_BITS | the MEMBER's own row}, one per informative hit and member
 * (a member has fewer than 2^27 - 1 rows, else SK_E_STATE).  Batches of any size below 4 GiB.  New: the reference holds one
 * strain per process
 * (src/strain_detect.c:137-146) and re-reads the metagenome for each (:263-384). */
#define SK_UNION_MAX 32
#define SK_UNION_ROW_BITS 27
typedef struct sk_union sk_union;
int  sk_union_create(sk_ctx *const *members, uint32_t n, uint32_t type_col, uint32_t informative_value, sk_union **out); /* errors: members[0] */
void sk_union_destroy(sk_union *u);
int  sk_union_tally_launch(sk_union *u, const sk_batch *b, uint64_t hits_cap);
int  sk_union_tally_collect(sk_union *u, sk_tally_rec *out, uint64_t cap, uint64_t *n, sk_hit *out_hits /* hits_cap */, uint64_t *out_nhits);
int  sk_union_sync(sk_union *u);                               /* wait for a launch whose results will not be collected */
int  sk_union_scan_timing(sk_union *u, double *total_ms, uint64_t *launches, int reset);   /* sk_scan_timing of the union's scans */
const char *sk_union_last_error(const sk_union *u);          /* of launch/collect */
uint32_t sk_union_members(const sk_union *u);
uint32_t sk_union_rows(const sk_union *u);                   /* the members' rows added up */

/* Wait for all queued work of the context. */
int sk_sync(sk_ctx *ctx);

/* Counter access, in the caller's row order; u32, wrapping.  (On the device a column is
 * counts[col * nrows + i] with i = row, or i = locality[row] after sk_table_load_ex.)
 * fetch/set replace reads/writes of the reference's per-key count vectors
 * (src/kmer_scrub_count.c:144-151; src/genome_compare.c:1011-1016). */
int sk_counts_fetch(sk_ctx *ctx, uint32_t col, uint32_t *out /* nrows */);
int sk_counts_set(sk_ctx *ctx, uint32_t col, const uint32_t *in /* nrows */);
/* counts[col][rows[i]] = value, i < n: a sparse update (new; strain_detect's type column names 1 % of the rows informative) */
int sk_counts_set_rows(sk_ctx *ctx, uint32_t col, const uint32_t *rows, uint32_t n, uint32_t value);
int sk_counts_zero(sk_ctx *ctx, uint32_t col);
/* Device address of the whole counter block (ncols * nrows u32) for a caller-run collective
 * (torch.distributed / RCCL all-reduce over xGMI).  New: the reference is single-process. */
void    *sk_counts_device_ptr(sk_ctx *ctx);
uint32_t sk_table_rows(const sk_ctx *ctx);
uint32_t sk_table_cols(const sk_ctx *ctx);
/* Multi-GPU, one process per GPU (new: the reference is single-process).
 * sk_comm_init: rank 0 creates an RCCL unique id and publishes it through `id_file`, the other
 * ranks wait (at most timeout_s) for it; every rank then joins the communicator.
 * sk_comm_sum_u32: tiny all-reduce used to agree on failure before the big one.
 * sk_counts_allreduce: in-place sum (u32, wrapping) of the whole counter block over xGMI;
 * rccl_comm = an ncclComm_t, or NULL for the context's own communicator. */
int  sk_comm_init(sk_ctx *ctx, int rank, int world, const char *id_file, int timeout_s);
/* The same, with this rank's set-up status in the exchange that precedes the collective: setup_failed != 0 (ctx may then
 * be NULL: no device, no key set) makes EVERY rank return SK_E_RCCL before anyone blocks in RCCL.  The exchange proves
 * freshness (a file left by a crashed launch is ignored), every wait is bounded by timeout_s, and a watchdog ends the
 * process (exit status 3) if ncclCommInitRank itself does not return in that time.  id_file must be visible to all
 * ranks (node-local /tmp serves one node only). */
int  sk_comm_init_ex(sk_ctx *ctx, int rank, int world, const char *id_file, int timeout_s, int setup_failed);
/* The file exchange alone (no GPU, no RCCL): returns 0 = go, 1 = somebody failed, 2 = timed out, 3 = cannot write.
 * payload128: in on rank 0, out elsewhere.  Exported for the multi-process CPU tests of the protocol. */
int  sk_rendezvous_exchange(int rank, int world, const char *base_path, int my_status, unsigned char *payload128, double timeout_s);
void sk_comm_destroy(sk_ctx *ctx);
int  sk_comm_sum_u32(sk_ctx *ctx, uint32_t value, uint32_t *sum);
/* *agree = 1 iff every rank passed the same value (one max all-reduce of {v, ~v}); without a communicator: 1.
 * skh_scan_list uses it to compare the ranks' work plans before any of them scans. */
int  sk_comm_agree_u64(sk_ctx *ctx, uint64_t value, int *agree);
/* vals[i] = the maximum of vals[i] over all ranks, i < n <= 8 (one all-reduce); without a communicator: unchanged.  The list
 * walk's two agreements per list are built on it (skh_scan_list below): whatever happens to a rank locally, every rank issues
 * the same sequence of collectives.  sk_comm_world: ranks of the context's communicator, 0 without one. */
int  sk_comm_max_u64(sk_ctx *ctx, uint64_t *vals, uint32_t n);
int  sk_comm_world(const sk_ctx *ctx);
int  sk_counts_allreduce(sk_ctx *ctx, void *rccl_comm);

/* Device-side timing of the scan kernel, from HIP events recorded on the context's stream
 * around every scan kernel since the last reset: total milliseconds and launch count. */
int sk_scan_timing(sk_ctx *ctx, double *total_ms, uint64_t *launches, int reset);

/* Tunables (before sk_table_load).  name: "table_load_pct" (max load factor in percent), "grid_kib" (size of
 * the level-1 filter in KiB, -1 = automatic), "text_stage" (0 = stage 2 probes every window on its own even
 * when the strain's text is resident; for A/B runs and tests), "pipeline" (2 = the partitioned pipeline sk_bin ->
 * sk_lds_probe -> candidates-only scan, an experiment that measured slower than the default single kernel: DESIGN.md
 * section 4), "odd_list_cap" (tests), "dev_alloc_uncached" (experiment: no effect), "ablate" (timing
 * experiments only: kernel variants that skip memory stages and give WRONG counts).
 * Unknown name -> SK_E_ARG. */
int sk_set_option(sk_ctx *ctx, const char *name, long value);

/* Device buffer helpers so that FFI callers need no HIP binding of their own. */
int sk_dev_alloc(sk_ctx *ctx, void **dev, uint64_t nbytes);
int sk_dev_free(sk_ctx *ctx, void *dev);
int sk_dev_upload(sk_ctx *ctx, void *dev, const void *host, uint64_t nbytes);
int sk_dev_download(sk_ctx *ctx, void *host, const void *dev, uint64_t nbytes);

/* ----------------------------------------------------------------------------------------
 * host layer (C): the reference's file semantics around the device layer
 * -------------------------------------------------------------------------------------- */

typedef struct skh_keyset {
    uint32_t  nrows;        /* distinct oriented keys, in OUTPUT (BIO_hash slot) order          */
    uint32_t  nwide;        /* how many of them are wide (non-ACGT bytes)                       */
    uint64_t *packed;       /* [nrows] canonical packed key or SK_KEY_NONE                      */
    uint32_t *first_count;  /* [nrows] column-0 value after the build phase                     */
    uint32_t *locality;     /* [nrows] first-occurrence rank of each row's key along the strain    */
    char     *wide_keys;    /* [nwide*32]                                                       */
    uint32_t *wide_rows;    /* [nwide]                                                          */
    uint32_t  final_slots;  /* M of the replayed reference table                                */
    uint64_t  short_records;/* records skipped because shorter than k-1 (reference crashes)     */
    uint32_t *text2;        /* the strain's bases, 2 bits each (sk_table_load_text)             */
    uint32_t  text_bases;   /* how many                                                         */
    uint32_t *first_pos;    /* [nrows] text position of the row's first all-ACGT occurrence, or UINT32_MAX */
} skh_keyset;

/* Build phase.  Reads a FASTA/FASTQ(.gz) strain file with the reference parser's grammar
 * (src/kseq.h:166-211), upper-cases, extracts every window's oriented key that contains no
 * 'N', de-duplicates in first-occurrence order with column 0 = default_val + incr*(repeats)
 * and replays BIO_hash's insertion/doubling (src/BIO_hash.c:39-61,129-139,208-216) from
 * `initial_slots` to put the rows in the reference's output order.
 * Replaces GEN_hash_sequences_set_count_vec(): src/genome_compare.c:967-1030. */
int  skh_keyset_from_file(skh_keyset *ks, const char *path, uint32_t initial_slots,
                          uint32_t default_val, uint32_t incr);
int  skh_keyset_from_stream(skh_keyset *ks, const char *stream, size_t nbytes,
                            uint32_t initial_slots, uint32_t default_val, uint32_t incr);
void skh_keyset_free(skh_keyset *ks);
/* Decode row r's key to 31 ASCII bytes + NUL. */
void skh_keyset_key(const skh_keyset *ks, uint32_t row, char out[32]);
/* Load a keyset into a context (sk_table_load + sk_table_load_wide + column 0). */
int  skh_keyset_load(sk_ctx *ctx, const skh_keyset *ks, uint32_t ncols);

/* Scan one FASTA/FASTQ(.gz) file into column `col`; *bases (may be NULL) accumulates the
 * sequence bytes seen.  Replaces GEN_calculate_kmer_count(): src/genome_compare.c:179-236. */
int skh_scan_file(sk_ctx *ctx, const char *path, uint32_t col, uint64_t *bases);

/* Walk a newline-separated file list.  `skip` (may be NULL): a line equal to it is not
 * scanned and "skipping %s (identical match)" goes to `err`.  `progress` (may be NULL) gets
 * "<line>\t<asctime>".  On an unreadable list or file the reference's message is written to
 * `err` and SK_E_OPEN returned (the progress file then ends with the unreadable file's line and no later
 * skip message is printed, as in the reference, which exits there).  With world > 1 the list's items -- files,
 * or byte ranges of big plain-text files -- are dealt to the ranks by size and this rank scans its own; rank 0
 * writes the progress file and the messages (world=1, rank=0 for the single-GPU program).
 * Replaces GEN_all_kmer_counts(): src/genome_compare.c:149-177 and
 * GEN_all_kmer_counts_skip_file(): src/genome_compare.c:115-146. */
int skh_scan_list(sk_ctx *ctx, const char *list_path, const char *skip, uint32_t col,
                  FILE *progress, FILE *err, uint32_t rank, uint32_t world, uint64_t *bases);
/* What happens when a big text file's cut does not hold (a piece does not end between two records: SK_E_SPLIT inside).  One
 * process, or ranks that are all in the context's communicator (sk_comm_init): nothing the caller sees -- the column is put
 * back on every rank and the list scanned again uncut, every rank returning the same status (two small agreements per list
 * scan, sk_comm_max_u64: the same sequence of collectives on every rank whatever fails where).  Ranks WITHOUT the library's
 * communicator (a caller that reduces through torch.distributed): SK_E_SPLIT comes back and the caller does the same steps
 * itself -- copy the column before, agree after, sk_counts_set + skh_scan_list_uncut on every rank (strainer2_amd/dist.py:
 * scan_list_sharded).  skh_scan_list_uncut = skh_scan_list over the plan without byte-range pieces (SK_NO_SPLIT=1). */
int skh_scan_list_uncut(sk_ctx *ctx, const char *list_path, const char *skip, uint32_t col,
                        FILE *progress, FILE *err, uint32_t rank, uint32_t world, uint64_t *bases);
/* Hash of the work plan skh_scan_list(list, skip, .., world) follows (items, byte ranges, file sizes, owners): a function
 * of the list, the files and `world` (and of SK_SPLIT_BYTES / SK_NO_SPLIT) only, never of a rank's thread count.  With the
 * library's own communicator skh_scan_list compares it across ranks itself (SK_E_PLAN); a caller that reduces the counters
 * through a collective of its own (torch.distributed) compares this value first.  New (the reference is one process). */
int skh_list_plan_hash(const char *list_path, const char *skip, uint32_t world, uint64_t *hash);
/* The same plan, line by line: owner[i] = the rank that scans list line i, SKH_PLAN_SKIPPED for the line equal to `skip`,
 * SKH_PLAN_SHARED for a big plain-text file whose byte ranges go to several ranks; *nlines = lines in the list (owner[]
 * is filled up to `cap`). */
#define SKH_PLAN_SKIPPED 0xFFFFFFFFu
#define SKH_PLAN_SHARED  0xFFFFFFFEu
int skh_list_plan_owners(const char *list_path, const char *skip, uint32_t world, uint32_t *owner, uint32_t cap, uint32_t *nlines);

/* Print the TSV: src/kmer_scrub_count.c:134-156 (5 header names always; 4 or 5 fields). */
int skh_print_counts(sk_ctx *ctx, const skh_keyset *ks, FILE *out, int with_drug_column);

/* The whole program with the reference's argv contract (src/kmer_scrub_count.c:29-131).
 * Extra environment: SK_DEVICE (default 0).  Returns the process exit status. */
int skh_kmer_scrub_count_main(int argc, char **argv, FILE *out, FILE *err);

/* The whole strain_detect program with the reference's argv contract (src/strain_detect.c:61-158):
 * -r -a -o and one of -b [-c] [-t SE|PE|PEI] / -B, optional -g.  Messages the reference prints on
 * stdout go to `out`, stderr texts to `err`; the -o file is gzip (members compressed in parallel; the
 * decompressed bytes are the reference's).  Returns the exit status. */
int skh_strain_detect_main(int argc, char **argv, FILE *out, FILE *err);
/* The same on a strain that is ALREADY resident (the table of step 1 serves step 3): ctx holds its table,
 * loaded with at least 6 columns, *ks its key set -- both are taken over and released by the call --; informative_path is
 * what -a would name.  argv = the rest of a strain_detect command line (-B/-b/-c/-t/-g/-o, --coverage-depth ...), argv[0]
 * ignored; -r and -a are implied.  `kmer_scrub_count ... --scrub m --detect <those arguments>` runs the reference's
 * steps 1 to 3 (4 with --coverage-depth) in one process this way. */
int skh_strain_detect_resident(sk_ctx *ctx, skh_keyset *ks, const char *informative_path, int argc, char **argv, FILE *out, FILE *err);

/* Record reader exposed for tests: decode `path` into the record stream, calling `sink` with
 * successive chunks (records separated by '\n'; a long record may be cut with a k-1 overlap).
 * Returns number of records, or SK_E_OPEN. */
typedef int (*skh_sink_fn)(void *user, const uint8_t *chunk, uint64_t nbytes);
int64_t skh_decode_file(const char *path, uint64_t chunk_bytes, skh_sink_fn sink, void *user,
                        uint64_t *bases);

/* ------------------------------------------------------------------------------------------------
 * Scrub filter: the consumer of the count table (reference scripts/kmer_scrub_filter.py, step 2 of
 * test/example.sh).  The count columns live on the device, either uploaded from a parsed table
 * (sk_filter_load) or taken straight from the counters a scan has just filled (sk_filter_load_counts:
 * the table never becomes text); the ranking the script does with a full sort is a radix selection
 * there.  A "row" is one k-mer of the strain; rows marked gone do not take part (the script's drug
 * scrub, :62-69) but still count towards the column sums, as their dictionary entries do in the script.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sk_filter sk_filter;
int  sk_filter_create(sk_ctx *ctx, sk_filter **out);
void sk_filter_destroy(sk_filter *f);
/* Counts as the table's text carries them (signed: a counter >= 2^31 was printed negative by %d and is
 * "not > 0" to the script). */
int  sk_filter_load(sk_filter *f, const int64_t *pan, const int64_t *meta, const uint8_t *gone /* may be NULL */, uint64_t n);
/* Columns of the context's resident counters, in row order; drug_col < 0: no drug column, else rows
 * with a positive drug count are gone.  Replaces print_hash_counts + the script's parse (:164-201). */
int  sk_filter_load_counts(sk_filter *f, uint32_t pan_col, uint32_t meta_col, int32_t drug_col);
/* Sums and sizes of the script's pangenome_hash / metagenome_hash (positive entries only) (:91-106,204). */
int  sk_filter_sums(sk_filter *f, int64_t *pan_sum, int64_t *meta_sum, uint64_t *n_pan, uint64_t *n_meta, uint64_t *n_gone);
/* Value histogram of one column (0 = pan, 1 = meta) over its positive entries: hist[b] = #{v == lo + b}
 * for b < nbins, hist[nbins] = #{v >= lo + nbins}.  Feeds scrub_max_kmers' threshold walk (:30-58). */
int  sk_filter_hist(sk_filter *f, int which, int64_t lo, uint32_t nbins, uint64_t *hist /* nbins + 1 */);
/* joint_scrub (:87-137): remove the n_scrub rows with the largest max(pan/pan_sum, meta/meta_sum),
 * equal scores in row order.  out[r] = 1 for every row that is NOT in the result (gone before, or
 * removed now). */
int  sk_filter_joint(sk_filter *f, int64_t pan_sum, int64_t meta_sum, uint64_t n_scrub, uint8_t *out /* n */);
/* independent_scrub (:72-84): remove rows whose pan count exceeds pan_thr or meta count meta_thr. */
int  sk_filter_above(sk_filter *f, int64_t pan_thr, int64_t meta_thr, uint8_t *out /* n */);

/* The script's command line (-s/--scrub_count_file, -l/--scrub_count_list, -m/--min_fraction,
 * -i/--independent): same stdout bytes, same exit status, the same "kept ..." lines on stderr. */
int skh_scrub_filter_main(int argc, char **argv, FILE *out, FILE *err);
/* Fused step 1 -> step 2: what the script would print for the table skh_print_counts would print,
 * computed from the resident counters (columns 1, 2 and, with_drug_column, 3). */
int skh_scrub_filter_resident(sk_ctx *ctx, const skh_keyset *ks, int with_drug_column, double min_fraction,
                              int independent, FILE *out, FILE *err);

/* ------------------------------------------------------------------------------------------------
 * Coverage / depth: the consumer of strain_detect's hit list (reference scripts/coverage_depth.py,
 * step 4 of test/example.sh).
 * ---------------------------------------------------------------------------------------------- */
/* Per sample s < nsamples: out_total[s] = number of (sample, key) pairs with that sample,
 * out_unique[s] = number of different keys among them.  Replaces the script's unique_kmers_by_metagenome
 * / kmer_depth_count_by_metagenome bookkeeping (:64-93).  Keys are any u64 but 2^64-1.  Synchronous. */
int sk_distinct_count(sk_ctx *ctx, const uint64_t *keys, const uint32_t *sample, uint64_t n, uint32_t nsamples,
                      uint64_t *out_unique, uint64_t *out_total);
/* The same for hit lists whose k-mer text is not a fixed-length ACGT word: id[i] < nids numbers line i's
 * joined string <sample><k-mer> (file order); a string counts once, for the sample of the first line showing it,
 * as the script's global dictionary has it.  Synchronous. */
int sk_first_seen_count(sk_ctx *ctx, const uint32_t *id, const uint32_t *sample, uint64_t n, uint32_t nids, uint32_t nsamples,
                        uint64_t *out_unique, uint64_t *out_total);
/* The script's command line (-k/--kmer_hits_file, -m/--min_kmer_hits, -b/--background_metagenomes_file):
 * same stdout bytes and exit status. */
int skh_coverage_depth_main(int argc, char **argv, FILE *out, FILE *err);

#ifdef __cplusplus
}
#endif
#endif /* STRAINER_KMER_H */
