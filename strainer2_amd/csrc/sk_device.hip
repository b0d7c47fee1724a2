// sk_device.hip -- device layer of libstrainer_kmer.so: hand-written HIP for CDNA4 / gfx950.
//
// Kernels (all integer/byte work; bound by L2 request rate and random-access rate, no MFMA):
//   sk_scan_grid      THE hot kernel (replaces reference src/genome_compare.c:213-229 +
//                     src/BIO_hash.c:161-172; in TALLY mode src/strain_detect.c:471-485):
//                       phase 1  bytes -> 2-bit code words + "not ACGT" masks in LDS (SWAR)
//                       phase 2  one filter lookup per aligned 16-base chunk (a 31-base window holds exactly one)
//                       stage 2  the windows of the surviving chunks: a few of them (anchors) are probed in the
//                                HBM table; a hit tells where the read lies on the strain, and the other windows
//                                are compared with the strain's 2-bit text there (seed and verify: 62-bit
//                                compares, as exact as a probe); atomicAdd on the row counter on a hit
//   sk_scan_wide      exact byte-string path for the rare windows that contain bytes other than
//                     ACGT (U / IUPAC / junk): the reference's signed-char orientation compare
//                     through its COMPLEMENT map.  Early exit unless phase 1 saw such a byte.
//   sk_table_insert   open-addressed key table (atomicCAS on the key word)
//   sk_grid_insert    the two filter levels (canonical 16-mers / 24-mers of every key)
//   sk_gather/scatter counters <-> caller row order
//
// Data layout in HBM:
//   slots  [S]  16 B  {u64 key, u32 counter index, u32 text position << 1 | orientation}: open addressing,
//                     linear probing, S = 2^s >= 2 nrows, empty = key all ones (one random line per hit)
//   grid1       8 B   blocks of the Bloom set of the strain's canonical 16-mers (level 1, sized for the L2)
//   grid2       8 B   blocks of the Bloom set of the strain's canonical 24-mers (level 2: the chunk + the 8 bases on either side)
//   text2       u32   the strain's bases, 2 bits each, records end to end (1.25 MB for 5 Mbp)
//   rank        16 B  per 64 text positions: {counter index of the first one, 64 "a row starts here" bits}
//   counts [ncols][nrows] u32, in locality (first-occurrence) order: one read's hits are adjacent
//   stream      u8    record stream: sequence bytes, records separated by '\n'
//
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <string>
#include <vector>
#include <unistd.h>
#include <pthread.h>
#include <errno.h>
#include <time.h>

#include "../../include/strainer_kmer.h"
#include "sk_rendezvous.h"
#include "sk_common.h"
#include "sk_internal.h"

#include "sk_dev_scan.hip.h"          // the scan kernel (sk_scan_grid) and everything it is made of
#include "sk_dev_pipeline.hip.h"      // the partitioned pipeline (experiment, selectable)
#include "sk_dev_kernels.hip.h"       // byte-string kernel, table load / build / filter kernels

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
#define SK_STAGE_BYTES   (64ull << 20)
#define SK_NSTAGE        2

struct sk_pin { void *p; size_t n; bool used; bool registered; };    // registered: malloc'd + hipHostRegister; else hipHostMalloc

struct sk_ctx {
    int          device;
    hipStream_t  stream;
    std::vector<sk_pin> pins;         // page-locked host buffers of sk_pinned_alloc, kept for re-use
    pthread_mutex_t pin_mu;
    // table
    sk_u4       *d_keys;              // the slot array (name kept: "is a table loaded" checks)
    uint32_t     slots_log2;
    uint32_t    *d_text2;                 // seed and verify: the strain's text and the rank map (sk_table_load_text)
    sk_u4       *d_rank;
    uint32_t     text_bases;
    uint2       *d_grid1, *d_grid2;       // grid kernel's two filter levels
    uint32_t     grid1_blocks, grid2_blocks_log2;
    long         grid_kib;                // option: size of level 1 in KiB (-1 = automatic)
    uint64_t    *d_keys_by_row;           // sk_table_build_from_text: the keys in row order, until sk_table_export_keys fetched them
    bool         grid_pending;            // the two levels are allocated and zeroed, not filled yet (sk_grid_ensure)
    bool         all_rows_in_text;        // every key is a window of the text stage (sk_table_load_text)
    long         odd_cap;                 // option (tests): usable length of the odd-chunk list, 0 = all of it
    uint32_t     nrows, ncols;
    uint32_t    *d_counts;
    uint32_t    *d_perm, *d_inv;      // locality order of the counters (NULL = caller's row order)
    uint32_t    *d_locality;          // the caller's locality[] as given (with the orientation bit)
    uint32_t    *d_tmp;               // [nrows] scratch for fetch/set through the permutation
    uint32_t    *d_infbits;           // TALLY: informative-row bitmap of (infbits_col, infbits_val); valid while infbits_ok
    uint32_t     infbits_col, infbits_val; bool infbits_ok;
    uint32_t    *d_diff, *d_diff_sums;// difference array of the column being scanned [nrows + 1] and its block sums
    int          diff_col;            // the column d_diff belongs to, -1 = nothing pending
    std::vector<uint32_t> h_perm;     // host copy of the permutation (empty = identity)
    // wide keys
    char        *d_wide_keys;
    uint32_t    *d_wide_rows;
    uint32_t    *d_wide_index;
    uint32_t     wide_mask, nwide;
    // staging for host-resident streams
    uint8_t     *h_stage[SK_NSTAGE];
    uint8_t     *d_stage[SK_NSTAGE];
    hipEvent_t   stage_done[SK_NSTAGE];
    int          stage_next;
    hipStream_t  copy_stream;          // sk_scan_pinned's uploads: the next chunk's copy runs while this chunk is scanned
    hipEvent_t   copied[64];           // ring of "host buffer of ticket t has been read" events
    uint64_t     tickets;              // tickets issued so far
    // flags: [0] wide windows seen in the current batch, [1] table build errors
    uint32_t    *d_flags;
    uint32_t     flag_set;            // which of the two sets of scan flags (words 0..3 / 4..7) the launch in hand uses
    bool         flags_ready;         // both sets are zero (false after anything else wrote to the flag block)
    uint32_t    *d_oddlist;           // chunks with a byte for the byte-string kernel (SK_ODDCAP entries)
    // timing
    std::vector<hipEvent_t> ev;        // begin/end pairs of launches not yet added up: a ring of at most SK_EV_PAIRS
    std::vector<hipEvent_t> ev_free;   // pairs that have been added up, for the next launches
    double       timed_ms;
    uint64_t     timed_launches;
    // options
    long         table_load_pct;
    long         ablate;              // timing experiments: kernel variants that skip memory stages
    long         dev_uncached;        // experiment: sk_dev_alloc hands out memory the L2 does not keep
    uint32_t    *d_grid3;             // partitioned pipeline: the SK_BIN_P filter slices (bitmaps) of sk_lds_probe
    void        *p_bins, *p_binn, *p_cand;    // its grow-only scratch: segments, segment fills, candidate bytes
    size_t       p_bins_cap, p_binn_cap, p_cand_cap;
    long         pipeline;            // option: 0/1 = the single kernel (default), 2 = the partitioned pipeline (experiment)
    long         no_text;             // option "text_stage"=0: stage 2 probes every window on its own (A/B, tests)
    void        *t_tally, *t_hits, *t_compact;    // grow-only device scratch of the tally path
    size_t       t_tally_cap, t_hits_cap, t_compact_cap;
    uint8_t     *h_tally;             // pinned landing area of the tallies (+ the hit counter)
    size_t       h_tally_cap;
    uint32_t     t_inflight_nrec;     // a sk_tally_launch waiting for its sk_tally_collect
    uint64_t     t_inflight_cap;
    struct sk_batch *own_batch;       // sk_tally_batch's private batch
    void        *comm;                // RCCL communicator from sk_comm_init (NULL: single process)
    int          comm_rank, comm_world;
    char         err[512];
};

static int sk_fail(sk_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define SK_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return sk_fail((ctx), SK_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                  \
    } while (0)

extern "C" const char *sk_strerror(int code)
{
    switch (code) {
    case SK_OK: return "ok";
    case SK_E_NODEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case SK_E_HIP: return "HIP runtime error";
    case SK_E_ARG: return "bad argument";
    case SK_E_NOMEM: return "out of memory";
    case SK_E_OPEN: return "could not open file";
    case SK_E_DUPKEY: return "duplicate or malformed key in table load";
    case SK_E_STATE: return "call out of order";
    case SK_E_RCCL: return "RCCL error";
    case SK_E_SPLIT: return "a file could not be cut at record boundaries";
    case SK_E_PLAN: return "the ranks computed different work plans";
    default: return "unknown error";
    }
}

extern "C" const char *sk_last_error(const sk_ctx *ctx) { return ctx ? ctx->err : "no context"; }

extern "C" int sk_ctx_create(sk_ctx **out, int device)
{
    if (!out) return SK_E_ARG;
    *out = NULL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SK_E_NODEVICE;
    if (device < 0 || device >= ndev) return SK_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return SK_E_NODEVICE;
    {   // (experiment) SK_SYNC=blocking: waits for the device sleep instead of spinning -- the hosts here are short of CPUs, not of latency
        const char *e = getenv("SK_SYNC");
        if (e && !strcmp(e, "blocking")) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
        else if (e && !strcmp(e, "yield")) (void)hipSetDeviceFlags(hipDeviceScheduleYield);
    }
    sk_ctx *c = new (std::nothrow) sk_ctx();
    if (!c) return SK_E_NOMEM;
    c->device = device;
    pthread_mutex_init(&c->pin_mu, NULL);
    c->table_load_pct = 50;
    c->grid_kib = -1;
    c->diff_col = -1;
    c->err[0] = 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SK_E_NODEVICE; }
    if (hipMalloc((void **)&c->d_flags, 32 * sizeof(uint32_t)) != hipSuccess) { delete c; return SK_E_NOMEM; }   // (words 16..31: scratch of the small collectives)
    if (hipMalloc((void **)&c->d_oddlist, (size_t)SK_ODDCAP * sizeof(uint32_t)) != hipSuccess) { hipFree(c->d_flags); delete c; return SK_E_NOMEM; }
    hipMemsetAsync(c->d_flags, 0, 32 * sizeof(uint32_t), c->stream);
    {   // the complement map goes to the device once per process and device, not once per context: the copy to a symbol
        // waits for the device, and 32 strains opened at once (strain_detect -S) spent 0.19 s each in here
        static pthread_mutex_t once_mu = PTHREAD_MUTEX_INITIALIZER;
        static unsigned long long done_mask[4];
        pthread_mutex_lock(&once_mu);
        const bool have = device < 256 && ((done_mask[device >> 6] >> (device & 63)) & 1ull);
        hipError_t e = hipSuccess;
        if (!have) {
            signed char comp[256];
            sk_fill_complement(comp);
            e = hipMemcpyToSymbol(HIP_SYMBOL(sk_comp_dev), comp, sizeof comp);
            if (e == hipSuccess && device < 256) done_mask[device >> 6] |= 1ull << (device & 63);
        }
        pthread_mutex_unlock(&once_mu);
        if (e != hipSuccess) { delete c; return SK_E_NODEVICE; }
    }
    *out = c;
    return SK_OK;
}

static void sk_table_release(sk_ctx *c)
{
    hipFree(c->d_keys); c->d_keys = NULL;
    hipFree(c->d_text2); c->d_text2 = NULL;
    hipFree(c->d_rank); c->d_rank = NULL;
    c->text_bases = 0;
    hipFree(c->d_grid1); c->d_grid1 = NULL;
    hipFree(c->d_grid2); c->d_grid2 = NULL;
    hipFree(c->d_grid3); c->d_grid3 = NULL;
    hipFree(c->d_keys_by_row); c->d_keys_by_row = NULL;
    c->grid_pending = false; c->all_rows_in_text = false;
    hipFree(c->d_counts); c->d_counts = NULL;
    hipFree(c->d_perm); c->d_perm = NULL;
    c->h_perm.clear();
    hipFree(c->d_inv); c->d_inv = NULL;
    hipFree(c->d_locality); c->d_locality = NULL;
    hipFree(c->d_tmp); c->d_tmp = NULL;
    hipFree(c->d_diff); c->d_diff = NULL;
    hipFree(c->d_diff_sums); c->d_diff_sums = NULL;
    c->diff_col = -1;
    hipFree(c->d_infbits); c->d_infbits = NULL; c->infbits_ok = false;
    hipFree(c->d_wide_keys); c->d_wide_keys = NULL;
    hipFree(c->d_wide_rows); c->d_wide_rows = NULL;
    hipFree(c->d_wide_index); c->d_wide_index = NULL;
    c->nrows = c->ncols = c->nwide = 0;
}

extern "C" void sk_comm_destroy(sk_ctx *c);
extern "C" void sk_batch_destroy(struct sk_batch *b);

extern "C" void sk_ctx_destroy(sk_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    sk_comm_destroy(c);
    sk_table_release(c);
    for (int i = 0; i < SK_NSTAGE; i++) {
        if (c->h_stage[i]) hipHostFree(c->h_stage[i]);
        if (c->d_stage[i]) hipFree(c->d_stage[i]);
        if (c->stage_done[i]) hipEventDestroy(c->stage_done[i]);
    }
    if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
    for (hipEvent_t e : c->ev) hipEventDestroy(e);
    for (hipEvent_t e : c->ev_free) hipEventDestroy(e);
    for (int i = 0; i < 64; i++) if (c->copied[i]) hipEventDestroy(c->copied[i]);
    if (c->own_batch) sk_batch_destroy(c->own_batch);
    hipFree(c->t_tally); hipFree(c->t_hits); hipFree(c->t_compact);
    hipFree(c->p_bins); hipFree(c->p_binn); hipFree(c->p_cand);
    if (c->h_tally) hipHostFree(c->h_tally);
    for (sk_pin &q : c->pins) { if (q.registered) { hipHostUnregister(q.p); free(q.p); } else hipHostFree(q.p); }
    pthread_mutex_destroy(&c->pin_mu);
    hipFree(c->d_flags);
    hipFree(c->d_oddlist);
    hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int sk_set_option(sk_ctx *c, const char *name, long value)
{
    if (!c || !name) return SK_E_ARG;
    if (!strcmp(name, "table_load_pct")) { if (value < 5 || value > 90) return SK_E_ARG; c->table_load_pct = value; return SK_OK; }
    if (!strcmp(name, "grid_kib")) { if (value < -1 || value == 0 || value > (1 << 22)) return SK_E_ARG; c->grid_kib = value; return SK_OK; }
    if (!strcmp(name, "odd_list_cap")) { if (value < 0 || value > (long)SK_ODDCAP) return SK_E_ARG; c->odd_cap = value; return SK_OK; }
    if (!strcmp(name, "dev_alloc_uncached")) { c->dev_uncached = value != 0; return SK_OK; }
    if (!strcmp(name, "text_stage")) { c->no_text = value == 0; return SK_OK; }
    if (!strcmp(name, "pipeline")) { if (value < 0 || value > 2) return SK_E_ARG; c->pipeline = value; return SK_OK; }
#ifdef SK_EXPERIMENTS
    if (!strcmp(name, "ablate")) { c->ablate = value; return SK_OK; }
#else
    // the timing variants of the hot kernel that leave memory stages out (some of them count wrongly) are compiled
    // only into an experiments build (make EXPERIMENTS=1): the product library has no switch that changes a count
    if (!strcmp(name, "ablate")) return value == 0 ? SK_OK : sk_fail(c, SK_E_ARG, "option \"ablate\" needs a library built with make EXPERIMENTS=1");
#endif
    return sk_fail(c, SK_E_ARG, "unknown option %s", name);
}

extern "C" int sk_table_load(sk_ctx *c, const uint64_t *keys, uint32_t nrows, uint32_t ncols)
{
    return sk_table_load_ex(c, keys, nrows, ncols, NULL);
}

extern "C" int sk_table_load_ex(sk_ctx *c, const uint64_t *keys, uint32_t nrows, uint32_t ncols, const uint32_t *locality)
{
    if (!c || (!keys && nrows) || ncols == 0 || ncols > 16) return SK_E_ARG;
    if (locality) {                                    // must be a permutation of 0..nrows-1
        std::vector<uint8_t> seen(nrows, 0);
        for (uint32_t i = 0; i < nrows; i++) {
            const uint32_t l = locality[i] & 0x7FFFFFFFu;
            if (l >= nrows || seen[l]) return sk_fail(c, SK_E_ARG, "locality is not a permutation");
            seen[l] = 1;
        }
    }
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    sk_table_release(c);

    uint32_t lg = 10;
    while (((uint64_t)1 << lg) * (uint64_t)c->table_load_pct < (uint64_t)nrows * 100ull && lg < 31) lg++;
    const uint64_t slots = (uint64_t)1 << lg;
    c->slots_log2 = lg;
    SK_HIP(c, hipMalloc((void **)&c->d_keys, slots * sizeof(sk_u4)));
    const size_t cbytes = (size_t)(nrows ? nrows : 1) * ncols * sizeof(uint32_t);
    SK_HIP(c, hipMalloc((void **)&c->d_counts, cbytes));
    SK_HIP(c, hipMemsetAsync(c->d_counts, 0, cbytes, c->stream));
    hipLaunchKernelGGL(sk_fill64, dim3(2048), dim3(256), 0, c->stream, (uint64_t *)c->d_keys, 2 * slots, SK_EMPTY64);
    SK_HIP(c, hipMemsetAsync(c->d_flags, 0, 16 * sizeof(uint32_t), c->stream));
    c->flags_ready = false;
    if (nrows) {
        uint64_t *d_in = NULL;
        SK_HIP(c, hipMalloc((void **)&d_in, (size_t)nrows * sizeof(uint64_t)));
        SK_HIP(c, hipMemcpyAsync(d_in, keys, (size_t)nrows * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        if (locality) {
            SK_HIP(c, hipMalloc((void **)&c->d_perm, (size_t)nrows * 4));
            SK_HIP(c, hipMalloc((void **)&c->d_inv, (size_t)nrows * 4));
            SK_HIP(c, hipMalloc((void **)&c->d_tmp, (size_t)nrows * 4));
            c->h_perm.resize(nrows);
            for (uint32_t i = 0; i < nrows; i++) c->h_perm[i] = locality[i] & 0x7FFFFFFFu;
            SK_HIP(c, hipMalloc((void **)&c->d_locality, (size_t)nrows * 4));
            SK_HIP(c, hipMemcpyAsync(c->d_locality, locality, (size_t)nrows * 4, hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(sk_perm_from_locality, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, c->d_perm, (const uint32_t *)c->d_locality, nrows);   // (not a second 4 N bytes over PCIe)
            SK_HIP(c, hipMemsetAsync(c->d_inv, 0xFF, (size_t)nrows * 4, c->stream));
            hipLaunchKernelGGL(sk_invert_perm, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, c->d_inv, c->d_perm, nrows, c->d_flags);
        }
        hipLaunchKernelGGL(sk_table_insert, dim3((nrows + 255) / 256), dim3(256), 0, c->stream,
                           d_in, nrows, c->d_keys, (uint32_t)(slots - 1), c->d_flags, c->d_perm, c->d_locality);
        {   // grid filters.  Level 1: 5.75 bits per key -- 3.4 MiB for a 5 Mbp strain, which stays in the L2 (4 MiB per XCD,
            // shared with everything else; round 3, once the kernel was bound by vector instructions: 3.0 -> 3.5 MiB saves more in
            // false positives' rounds than it costs in L2 misses: -1 % / -3 % / -6 % at 0 / 2 / 30 % strain reads; 4.5 MiB and up lose).  For bigger strains it is better to keep the 5 bits per key and leave the L2
            // than to keep the size and let the filter fill up (tools/grid_size_sweep.sh, 20 Mbp strain: 3 MiB 540,
            // 6 MiB 731, 12 MiB 850, 24 MiB 815 Gbase/s; 100 Mbp: 3 MiB 192, 64 MiB 465): misses of a sparse level 1
            // are served by the 256 MB Infinity Cache, the level-2 lookups a full one lets through are not.
            // Level 2 settles what level 1 lets through: >= 32 bits per key, false positives ~1e-5.
            uint64_t kib = c->grid_kib > 0 ? (uint64_t)c->grid_kib : ((uint64_t)nrows * 23ull / 32ull + 1023ull) / 1024ull;
            if (kib > (1ull << 22)) kib = 1ull << 22;
            if (kib < 4ull) kib = 4ull;
            c->grid1_blocks = (uint32_t)(kib * 1024ull / sizeof(uint2));
            uint32_t g2 = 12;
            while (g2 < 34 && ((uint64_t)1 << g2) < (uint64_t)nrows * 32ull) g2++;
            c->grid2_blocks_log2 = g2 - 6u;
            const size_t b1 = (size_t)c->grid1_blocks * sizeof(uint2), b2 = ((size_t)1 << c->grid2_blocks_log2) * sizeof(uint2);
            SK_HIP(c, hipMalloc((void **)&c->d_grid1, b1));
            SK_HIP(c, hipMalloc((void **)&c->d_grid2, b2));
            SK_HIP(c, hipMemsetAsync(c->d_grid1, 0, b1, c->stream));
            SK_HIP(c, hipMemsetAsync(c->d_grid2, 0, b2, c->stream));
            // filled when it is known from what: by sk_table_load_text from the strain's text (one insert per base instead of
            // sixteen per key), or -- no text stage -- from the slots when the first scan asks (sk_grid_ensure)
            c->grid_pending = true;
            c->all_rows_in_text = false;
        }
        uint32_t flags[2] = {0, 0};
        SK_HIP(c, hipMemcpyAsync(flags, c->d_flags, sizeof flags, hipMemcpyDeviceToHost, c->stream));
        SK_HIP(c, hipStreamSynchronize(c->stream));
        hipFree(d_in);
        if (flags[1]) { sk_table_release(c); return sk_fail(c, SK_E_DUPKEY, "%u duplicate/malformed keys", flags[1]); }
    }
    SK_HIP(c, hipGetLastError());
    c->nrows = nrows;
    c->ncols = ncols;
    return SK_OK;
}

extern "C" int sk_table_load_wide(sk_ctx *c, const char *keys31, const uint32_t *rows, uint32_t nwide)
{
    if (!c) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "sk_table_load first");
    if (nwide == 0) return SK_OK;
    if (!keys31 || !rows) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    uint32_t lg = 4;
    while (((uint32_t)1 << lg) < nwide * 2u) lg++;
    const uint32_t wslots = (uint32_t)1 << lg;
    std::vector<uint32_t> index(wslots, 0u);
    for (uint32_t i = 0; i < nwide; i++) {
        if (rows[i] >= c->nrows) return sk_fail(c, SK_E_ARG, "wide row %u out of range", rows[i]);
        uint32_t slot = sk_hash_wide(keys31 + (size_t)i * 32) & (wslots - 1);
        while (index[slot]) slot = (slot + 1) & (wslots - 1);
        index[slot] = i + 1;
    }
    SK_HIP(c, hipMalloc((void **)&c->d_wide_keys, (size_t)nwide * 32));
    SK_HIP(c, hipMalloc((void **)&c->d_wide_rows, (size_t)nwide * sizeof(uint32_t)));
    SK_HIP(c, hipMalloc((void **)&c->d_wide_index, (size_t)wslots * sizeof(uint32_t)));
    SK_HIP(c, hipMemcpy(c->d_wide_keys, keys31, (size_t)nwide * 32, hipMemcpyHostToDevice));
    std::vector<uint32_t> wrows(rows, rows + nwide);
    if (!c->h_perm.empty()) for (uint32_t i = 0; i < nwide; i++) wrows[i] = c->h_perm[rows[i]];
    SK_HIP(c, hipMemcpy(c->d_wide_rows, wrows.data(), (size_t)nwide * sizeof(uint32_t), hipMemcpyHostToDevice));
    SK_HIP(c, hipMemcpy(c->d_wide_index, index.data(), (size_t)wslots * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->wide_mask = wslots - 1;
    c->nwide = nwide;
    return SK_OK;
}

extern "C" int sk_table_load_text(sk_ctx *c, const uint32_t *text2, uint32_t nbases, const uint32_t *first_pos)
{
    if (!c || !text2 || !first_pos) return SK_E_ARG;
    if (!c->d_keys || !c->nrows) return sk_fail(c, SK_E_STATE, "sk_table_load_ex first");
    if (c->h_perm.empty()) return sk_fail(c, SK_E_STATE, "the text stage needs the locality order of sk_table_load_ex");
    if (nbases < SK_K || nbases > 0x7FFFFF00u) return sk_fail(c, SK_E_ARG, "text of %u bases (a table slot holds 31 bits of position)", nbases);
    const uint32_t nrows = c->nrows;
    // by counter index; the contract: rows with a position first, positions ascending
    std::vector<uint32_t> pos_by_idx(nrows, 0xFFFFFFFFu);
    for (uint32_t r = 0; r < nrows; r++) pos_by_idx[c->h_perm[r]] = first_pos[r];
    uint32_t m = 0;
    while (m < nrows && pos_by_idx[m] != 0xFFFFFFFFu) m++;
    for (uint32_t i = 0; i < nrows; i++) {
        const uint32_t p = pos_by_idx[i];
        if (i >= m) { if (p != 0xFFFFFFFFu) return sk_fail(c, SK_E_ARG, "rows with a text position must come first in locality order"); continue; }
        if (p > nbases - SK_K || (i && p <= pos_by_idx[i - 1])) return sk_fail(c, SK_E_ARG, "text positions must ascend with the locality order and lie inside the text");
    }
    // rank map: per 64 positions {counter index of the first row starting in the block, the 64 start bits}
    const size_t nblk = (size_t)nbases / 64 + 2;
    std::vector<sk_u4> rank(nblk, (sk_u4){0u, 0u, 0u, 0u});
    for (uint32_t i = 0; i < m; i++) {
        const uint32_t p = pos_by_idx[i];
        if (p & 32u) rank[p >> 6].z |= 1u << (p & 31u); else rank[p >> 6].y |= 1u << (p & 31u);
    }
    uint32_t run = 0;
    for (size_t b = 0; b < nblk; b++) { rank[b].x = run; run += (uint32_t)__builtin_popcount(rank[b].y) + (uint32_t)__builtin_popcount(rank[b].z); }
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    hipFree(c->d_text2); c->d_text2 = NULL;
    hipFree(c->d_rank); c->d_rank = NULL;
    c->text_bases = 0;
    const size_t words = (size_t)nbases / 16 + 4, have = ((size_t)nbases + 15) / 16;
    SK_HIP(c, hipMalloc((void **)&c->d_text2, words * 4));
    SK_HIP(c, hipMalloc((void **)&c->d_rank, nblk * sizeof(sk_u4)));
    hipFree(c->d_diff); c->d_diff = NULL;
    hipFree(c->d_diff_sums); c->d_diff_sums = NULL;
    c->diff_col = -1;
    SK_HIP(c, hipMalloc((void **)&c->d_diff, ((size_t)nrows + 2) * 4));
    SK_HIP(c, hipMalloc((void **)&c->d_diff_sums, ((size_t)nrows / SK_DIFF_PER_BLOCK + 2) * 4));
    SK_HIP(c, hipMemsetAsync(c->d_diff, 0, ((size_t)nrows + 2) * 4, c->stream));
    SK_HIP(c, hipMemsetAsync(c->d_text2, 0, words * 4, c->stream));
    SK_HIP(c, hipMemcpyAsync(c->d_text2, text2, have * 4, hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipMemcpyAsync(c->d_rank, rank.data(), nblk * sizeof(sk_u4), hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipMemcpyAsync(c->d_tmp, pos_by_idx.data(), (size_t)nrows * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(sk_table_setpos, dim3(4096), dim3(256), 0, c->stream, c->d_keys, (uint64_t)1 << c->slots_log2, c->d_tmp);
    c->all_rows_in_text = m == nrows;
    if (c->grid_pending && c->all_rows_in_text) {          // (rows without a place in the text -- a U in the strain -- : from the slots, sk_grid_ensure)
        hipLaunchKernelGGL(sk_grid_insert_text, dim3((nbases + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)c->d_text2, nbases,
                           (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
        c->grid_pending = false;
    }
    SK_HIP(c, hipStreamSynchronize(c->stream));
    SK_HIP(c, hipGetLastError());
    c->text_bases = nbases;
    return SK_OK;
}

// The whole table from the strain's text, on the device (round 3; strain_detect's opening: src/strain_detect.c:137-139 builds it with
// GEN_hash_sequences_set_count_vec(r, 31, h, 1, 0, 0, 6), src/genome_compare.c:967-1030).  The host parses the file and hands over the
// bases (2 bits each, records end to end) and, per position, whether a window of 31 A/C/G/T bases of one record starts there; keys,
// first occurrences, row numbers (by first occurrence: "strain order"), rank map, both filter levels and column 0 are made here --
// no 40 MB of keys, no permutations and no column 0 over PCIe, no hash table on the host.  *nrows_out = distinct keys.
static int sk_table_build_from_text_steps(sk_ctx *c, const uint32_t *text2, const uint32_t *startok, uint32_t nbases, uint32_t nstarts,
                                          uint32_t ncols, uint32_t col0_value, uint32_t *nrows_out, uint32_t *&d_ok);
extern "C" int sk_table_build_from_text(sk_ctx *c, const uint32_t *text2, const uint32_t *startok, uint32_t nbases, uint32_t nstarts,
                                        uint32_t ncols, uint32_t col0_value, uint32_t *nrows_out)
{
    if (!c || !text2 || !startok || !nrows_out || ncols == 0 || ncols > 16) return SK_E_ARG;
    if (nbases < SK_K || nbases > 0x7FFFFF00u) return sk_fail(c, SK_E_ARG, "text of %u bases (a table slot holds 31 bits of position)", nbases);
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    sk_table_release(c);
    // a build that fails midway (out of memory with many strains on one device) must leave neither its scratch nor half a table
    // behind: the caller falls back to the host builder on this same context
    uint32_t *d_ok = NULL;
    const int rc = sk_table_build_from_text_steps(c, text2, startok, nbases, nstarts, ncols, col0_value, nrows_out, d_ok);
    if (d_ok) (void)hipFree(d_ok);
    if (rc != SK_OK) {
        (void)hipStreamSynchronize(c->stream);
        sk_table_release(c);
        c->nrows = 0; c->ncols = 0;
    }
    return rc;
}

static int sk_table_build_from_text_steps(sk_ctx *c, const uint32_t *text2, const uint32_t *startok, uint32_t nbases, uint32_t nstarts,
                                          uint32_t ncols, uint32_t col0_value, uint32_t *nrows_out, uint32_t *&d_ok)
{
    uint32_t lg = 10;
    while (((uint64_t)1 << lg) * (uint64_t)c->table_load_pct < (uint64_t)nstarts * 100ull && lg < 31) lg++;
    const uint64_t slots = (uint64_t)1 << lg;
    const uint32_t mask = (uint32_t)(slots - 1);
    c->slots_log2 = lg;
    const size_t words = (size_t)nbases / 16 + 4, have = ((size_t)nbases + 15) / 16, nblk = (size_t)nbases / 64 + 2, bwords = ((size_t)nbases + 31) / 32;
    uint32_t *d_total = NULL;
    SK_HIP(c, hipMalloc((void **)&c->d_keys, slots * sizeof(sk_u4)));
    SK_HIP(c, hipMalloc((void **)&c->d_text2, words * 4));
    SK_HIP(c, hipMalloc((void **)&c->d_rank, nblk * sizeof(sk_u4)));
    SK_HIP(c, hipMalloc((void **)&d_ok, bwords * 4 + 4));
    d_total = d_ok + bwords;
    hipLaunchKernelGGL(sk_fill64, dim3(2048), dim3(256), 0, c->stream, (uint64_t *)c->d_keys, 2 * slots, SK_EMPTY64);
    SK_HIP(c, hipMemsetAsync(c->d_text2, 0, words * 4, c->stream));
    SK_HIP(c, hipMemsetAsync(c->d_rank, 0, nblk * sizeof(sk_u4), c->stream));
    SK_HIP(c, hipMemsetAsync(c->d_flags, 0, 16 * sizeof(uint32_t), c->stream));
    c->flags_ready = false;
    SK_HIP(c, hipMemcpyAsync(c->d_text2, text2, have * 4, hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipMemcpyAsync(d_ok, startok, bwords * 4, hipMemcpyHostToDevice, c->stream));
    const dim3 grid((nbases + 255) / 256), block(256);
    hipLaunchKernelGGL(sk_build_insert, grid, block, 0, c->stream, (const uint32_t *)c->d_text2, (const uint32_t *)d_ok, nbases, c->d_keys, mask);
    hipLaunchKernelGGL(sk_build_first, grid, block, 0, c->stream, (const uint32_t *)c->d_text2, (const uint32_t *)d_ok, nbases, (const sk_u4 *)c->d_keys, mask, c->d_rank);
    hipLaunchKernelGGL(sk_build_rank_scan, dim3(1), dim3(1024), 0, c->stream, c->d_rank, (uint32_t)nblk, d_total);
    uint32_t nrows = 0;
    SK_HIP(c, hipMemcpyAsync(&nrows, d_total, 4, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d_ok);
    d_ok = NULL;
    *nrows_out = nrows;
    c->nrows = nrows; c->ncols = ncols;
    c->text_bases = nbases;
    c->all_rows_in_text = true;
    const size_t cbytes = (size_t)(nrows ? nrows : 1) * ncols * sizeof(uint32_t);
    SK_HIP(c, hipMalloc((void **)&c->d_counts, cbytes));
    SK_HIP(c, hipMemsetAsync(c->d_counts, 0, cbytes, c->stream));
    if (nrows) {
        SK_HIP(c, hipMalloc((void **)&c->d_keys_by_row, (size_t)nrows * 8));
        hipLaunchKernelGGL(sk_build_index, grid, block, 0, c->stream, (const uint32_t *)c->d_text2, nbases, c->d_keys, mask, (const sk_u4 *)c->d_rank, c->d_keys_by_row);
        if (col0_value) hipLaunchKernelGGL(sk_fill32, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, c->d_counts, nrows, col0_value);
        // the filter levels, sized as sk_table_load_ex sizes them, filled from the text
        uint64_t kib = c->grid_kib > 0 ? (uint64_t)c->grid_kib : ((uint64_t)nrows * 23ull / 32ull + 1023ull) / 1024ull;
        if (kib > (1ull << 22)) kib = 1ull << 22;
        if (kib < 4ull) kib = 4ull;
        c->grid1_blocks = (uint32_t)(kib * 1024ull / sizeof(uint2));
        uint32_t g2 = 12;
        while (g2 < 34 && ((uint64_t)1 << g2) < (uint64_t)nrows * 32ull) g2++;
        c->grid2_blocks_log2 = g2 - 6u;
        const size_t b1 = (size_t)c->grid1_blocks * sizeof(uint2), b2 = ((size_t)1 << c->grid2_blocks_log2) * sizeof(uint2);
        SK_HIP(c, hipMalloc((void **)&c->d_grid1, b1));
        SK_HIP(c, hipMalloc((void **)&c->d_grid2, b2));
        SK_HIP(c, hipMemsetAsync(c->d_grid1, 0, b1, c->stream));
        SK_HIP(c, hipMemsetAsync(c->d_grid2, 0, b2, c->stream));
        hipLaunchKernelGGL(sk_grid_insert_text, grid, block, 0, c->stream, (const uint32_t *)c->d_text2, nbases,
                           (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
        c->grid_pending = false;
        c->diff_col = -1;
        SK_HIP(c, hipMalloc((void **)&c->d_diff, ((size_t)nrows + 2) * 4));
        SK_HIP(c, hipMalloc((void **)&c->d_diff_sums, ((size_t)nrows / SK_DIFF_PER_BLOCK + 2) * 4));
        SK_HIP(c, hipMemsetAsync(c->d_diff, 0, ((size_t)nrows + 2) * 4, c->stream));
    } else {                                             // (no key at all: an empty table the scans skip)
        c->grid1_blocks = 512; c->grid2_blocks_log2 = 6;
        SK_HIP(c, hipMalloc((void **)&c->d_grid1, 512 * sizeof(uint2)));
        SK_HIP(c, hipMalloc((void **)&c->d_grid2, 64 * sizeof(uint2)));
        SK_HIP(c, hipMemsetAsync(c->d_grid1, 0, 512 * sizeof(uint2), c->stream));
        SK_HIP(c, hipMemsetAsync(c->d_grid2, 0, 64 * sizeof(uint2), c->stream));
    }
    SK_HIP(c, hipStreamSynchronize(c->stream));
    SK_HIP(c, hipGetLastError());
    return SK_OK;
}

// the keys of a table built by sk_table_build_from_text, in row order (the device keeps its list: 8 bytes per row)
extern "C" int sk_table_export_keys(sk_ctx *c, uint64_t *keys_out)
{
    if (!c || (!keys_out && c->nrows)) return SK_E_ARG;
    if (!c->d_keys_by_row && c->nrows) return sk_fail(c, SK_E_STATE, "no key list to export (sk_table_build_from_text first)");
    SK_HIP(c, hipSetDevice(c->device));
    if (c->nrows) SK_HIP(c, hipMemcpy(keys_out, c->d_keys_by_row, (size_t)c->nrows * 8, hipMemcpyDeviceToHost));
    return SK_OK;
}
// ... of n chosen rows only (strain_detect prints the k-mers of informative rows -- 1 % of them -- and of no others: 40 MB per
// strain stay where they are)
extern "C" int sk_table_export_keys_of(sk_ctx *c, const uint32_t *rows, uint32_t n, uint64_t *keys_out)
{
    if (!c || (n && (!rows || !keys_out))) return SK_E_ARG;
    if (!n) return SK_OK;
    if (!c->d_keys_by_row) return sk_fail(c, SK_E_STATE, "no key list to export (sk_table_build_from_text first)");
    for (uint32_t i = 0; i < n; i++) if (rows[i] >= c->nrows) return sk_fail(c, SK_E_ARG, "row %u out of range", rows[i]);
    SK_HIP(c, hipSetDevice(c->device));
    void *d = NULL;
    SK_HIP(c, hipMalloc(&d, (size_t)n * 12));
    uint64_t *d_out = (uint64_t *)d;
    uint32_t *d_rows = (uint32_t *)(d_out + n);
    hipError_t e = hipMemcpyAsync(d_rows, rows, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sk_gather_keys, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_out, (const uint64_t *)c->d_keys_by_row, (const uint32_t *)d_rows, n);
        e = hipMemcpyAsync(keys_out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    SK_HIP(c, e);
    return SK_OK;
}

// the filter levels of a table that has no text stage (or keys outside it): from the slots, once, before the first scan
static int sk_grid_ensure(sk_ctx *c)
{
    if (!c->grid_pending) return SK_OK;
    hipLaunchKernelGGL(sk_grid_insert_slots, dim3(4096), dim3(256), 0, c->stream, (const sk_u4 *)c->d_keys, (uint64_t)1 << c->slots_log2,
                       (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
    SK_HIP(c, hipGetLastError());
    c->grid_pending = false;
    return SK_OK;
}

// Fold the pending difference array into its column (asynchronous on the context's stream).  Every entry point that
// reads or writes the counters calls this first; scans into another column do too.
static int sk_diff_flush(sk_ctx *c)
{
    if (c->diff_col < 0 || !c->d_diff) { c->diff_col = -1; return SK_OK; }
    const uint32_t n = c->nrows + 1u, nb = (n + SK_DIFF_PER_BLOCK - 1u) / SK_DIFF_PER_BLOCK;
    hipLaunchKernelGGL(sk_diff_block_sums, dim3(nb), dim3(256), 0, c->stream, c->d_diff, n, c->d_diff_sums);
    hipLaunchKernelGGL(sk_diff_scan_sums, dim3(1), dim3(1024), 0, c->stream, c->d_diff_sums, nb);
    hipLaunchKernelGGL(sk_diff_apply, dim3(nb), dim3(256), 0, c->stream, c->d_diff, n, c->d_diff_sums,
                       c->d_counts + (size_t)c->diff_col * c->nrows, c->nrows);
    c->diff_col = -1;
    SK_HIP(c, hipGetLastError());
    return SK_OK;
}

static int sk_scratch(sk_ctx *c, void **p, size_t *cap, size_t need);

// launch main + wide kernels over one device-resident batch
// packed_inv: the batch is in the host-packed form (sk_pack_stream) -- d_stream points at its code words, packed_inv at its masks
static int sk_launch_scan(sk_ctx *c, const uint8_t *d_stream, uint64_t nbytes, uint64_t emit_begin, uint32_t col,
                          const sk_sink *tally_sink = NULL, const void *packed_inv = NULL)
{
    if (nbytes <= emit_begin || c->nrows == 0) return SK_OK;        // (an empty key set: nothing can be counted)
    const uint64_t ntiles = (nbytes + SK_TILE - 1) / SK_TILE;
    if (ntiles > 0x7FFFFFFFull) return sk_fail(c, SK_E_ARG, "batch too large");
    sk_table_view tv;
    tv.nrows = c->nrows;
    tv.text2 = c->no_text ? NULL : c->d_text2; tv.rank = c->d_rank; tv.text_bases = c->text_bases;
    tv.slots = c->d_keys; tv.mask = (uint32_t)(((uint64_t)1 << c->slots_log2) - 1);
    tv.oddlist = c->d_oddlist; tv.oddcap = c->odd_cap ? (uint32_t)c->odd_cap : SK_ODDCAP;
    tv.grid1 = c->d_grid1; tv.grid2 = c->d_grid2;
    tv.grid1_blocks = c->grid1_blocks; tv.grid2_shift = 32u - c->grid2_blocks_log2;
    sk_wide_view wv;
    wv.keys31 = c->d_wide_keys; wv.rows = c->d_wide_rows; wv.index = c->d_wide_index;
    wv.wmask = c->wide_mask; wv.nwide = c->nwide;
    sk_sink sink;
    memset(&sink, 0, sizeof sink);
    if (tally_sink) sink = *tally_sink;
    else {
        c->infbits_ok = false;                                 // (a column is about to change)
        sink.counts = c->d_counts + (size_t)col * c->nrows;
        sink.diff = c->d_diff;
        if (tv.text2 && c->diff_col != (int)col) {            // the difference array serves one column at a time
            int rc = sk_diff_flush(c);
            if (rc) return rc;
            c->diff_col = (int)col;
        }
    }
    const dim3 grid((uint32_t)ntiles), block(SK_THREADS);
    if (!c->d_grid1) return sk_fail(c, SK_E_STATE, "no table loaded");
    { int grc = sk_grid_ensure(c); if (grc) return grc; }

    // [0] odd bytes seen, [2] listed chunks: two sets of four words taking turns -- this launch's byte-string kernel zeroes the other set
    // for the next launch (both sets are zero after a table load)
    if (!c->flags_ready) {                                    // (the first scan after a table was built: its kernels used the same words)
        SK_HIP(c, hipMemsetAsync(c->d_flags, 0, 8 * sizeof(uint32_t), c->stream));
        c->flags_ready = true;
    }
    c->flag_set ^= 1u;
    uint32_t *const d_fl = c->d_flags + 4u * c->flag_set, *const d_fl_next = c->d_flags + 4u * (c->flag_set ^ 1u);
    // timing: a begin/end event pair per launch from a small ring -- the oldest pair is added to the totals (its launch is
    // long over, SK_EV_PAIRS launches later) and used again, so a program that never asks for the timing holds 128 events,
    // not two per launch
    hipEvent_t e0 = NULL, e1 = NULL;
    const bool timed = true;
    if (c->ev.size() >= 2 * SK_EV_PAIRS) {
        float ms = 0.f;
        SK_HIP(c, hipEventSynchronize(c->ev[1]));
        SK_HIP(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
        c->timed_ms += ms;
        c->timed_launches++;
        c->ev_free.push_back(c->ev[0]);
        c->ev_free.push_back(c->ev[1]);
        c->ev.erase(c->ev.begin(), c->ev.begin() + 2);
    }
    if (c->ev_free.size() >= 2) {
        e1 = c->ev_free.back(); c->ev_free.pop_back();
        e0 = c->ev_free.back(); c->ev_free.pop_back();
    } else {
        SK_HIP(c, hipEventCreate(&e0));
        SK_HIP(c, hipEventCreate(&e1));
    }
    SK_HIP(c, hipEventRecord(e0, c->stream));
    // The partitioned pipeline (the level-1 question answered from LDS) is an experiment kept selectable (option
    // "pipeline" = 2; parity-tested): as measured it LOSES to the single kernel (0.43 against 0.34 ms per 0.6 Gbase at
    // 2 % strain reads; profiles/r02_lds_pipeline.txt) -- its one full pass over the stream plus the
    // bin write already costs half of the single kernel's time, and the re-read of the candidates' neighbourhoods pays
    // the fabric's random-line rate.  The default is the single kernel for every batch size.
    const bool piped = !packed_inv && (!c->ablate || c->ablate >= 7) && c->pipeline == 2;        // (ablations 7-9 exist for both forms)
    const uint8_t *d_cand = NULL;
    if (piped && !c->d_grid3) {
        // the partitioned pipeline's filter slices, built from the resident table the first time they are wanted:
        // SK_BIN_P bitmaps of 2^20 bits (16 MiB)
        const size_t b3 = (size_t)SK_BIN_P * SK_BIN_WORDS * 4;
        const uint64_t nslots = (uint64_t)1 << c->slots_log2;
        SK_HIP(c, hipMalloc((void **)&c->d_grid3, b3));
        SK_HIP(c, hipMemsetAsync(c->d_grid3, 0, b3, c->stream));
        hipLaunchKernelGGL(sk_grid3_insert, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, c->stream, (const sk_u4 *)c->d_keys, nslots, c->d_grid3);
    }
    if (piped) {
        const uint64_t ntiles_bin = (nbytes + SK_BIN_TILE - 1) / SK_BIN_TILE;
        int rc;
        if ((rc = sk_scratch(c, &c->p_bins, &c->p_bins_cap, (size_t)ntiles_bin * SK_BIN_P * SK_BIN_CAP * 4)) != SK_OK) return rc;
        if ((rc = sk_scratch(c, &c->p_binn, &c->p_binn_cap, (size_t)ntiles_bin * SK_BIN_P)) != SK_OK) return rc;
        if ((rc = sk_scratch(c, &c->p_cand, &c->p_cand_cap, (size_t)ntiles_bin * SK_BIN_CH + 64)) != SK_OK) return rc;
        SK_HIP(c, hipMemsetAsync(c->p_cand, 0, (size_t)ntiles_bin * SK_BIN_CH + 64, c->stream));
        hipLaunchKernelGGL(sk_bin, dim3((uint32_t)ntiles_bin), dim3(256), 0, c->stream, d_stream, nbytes, tv,
                           (uint32_t *)c->p_bins, (uint8_t *)c->p_binn, (uint32_t)ntiles_bin, (uint8_t *)c->p_cand, d_fl);
        // one workgroup per CU at a time (128 KiB of LDS each): a few shares per partition keep all 256 CUs busy
        uint32_t splits = ntiles_bin >= 4096 ? 4u : ntiles_bin >= 1024 ? 2u : 1u;
        static bool lds_attr = false;
        if (!lds_attr) {
            SK_HIP(c, hipFuncSetAttribute((const void *)sk_lds_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SK_BIN_WORDS * 4)));
            lds_attr = true;
        }
        hipLaunchKernelGGL(sk_lds_probe, dim3(SK_BIN_P * splits), dim3(1024), SK_BIN_WORDS * 4, c->stream, (const uint32_t *)c->d_grid3,
                           (const uint32_t *)c->p_bins, (const uint8_t *)c->p_binn, (uint32_t)ntiles_bin, splits, (uint8_t *)c->p_cand);
        d_cand = (const uint8_t *)c->p_cand;
    }
#define SK_LAUNCH_GRID(T, A, C) hipLaunchKernelGGL((sk_scan_grid<T, A, C>), grid, block, 0, c->stream, \
                                                   d_stream, nbytes, emit_begin, tv, sink, d_fl, d_cand)
    if (packed_inv) {
        if (c->ablate) return sk_fail(c, SK_E_UNSUPPORTED, "no ablations on packed batches");
        if (tally_sink && tally_sink->ns)
            hipLaunchKernelGGL((sk_scan_grid<true, 0, false, true, true>), grid, block, 0, c->stream, d_stream, nbytes, emit_begin, tv, sink, d_fl, (const uint8_t *)packed_inv);
        else if (tally_sink)
            hipLaunchKernelGGL((sk_scan_grid<true, 0, false, false, true>), grid, block, 0, c->stream, d_stream, nbytes, emit_begin, tv, sink, d_fl, (const uint8_t *)packed_inv);
        else
            hipLaunchKernelGGL((sk_scan_grid<false, 0, false, false, true>), grid, block, 0, c->stream, d_stream, nbytes, emit_begin, tv, sink, d_fl, (const uint8_t *)packed_inv);
    }
    else if (tally_sink && tally_sink->ns)
        hipLaunchKernelGGL((sk_scan_grid<true, 0, false, true>), grid, block, 0, c->stream, d_stream, nbytes, emit_begin, tv, sink, d_fl, d_cand);
    else if (piped && tally_sink) SK_LAUNCH_GRID(true, 0, true);
#ifdef SK_EXPERIMENTS
    else if (piped && c->ablate == 7) SK_LAUNCH_GRID(false, 7, true);
    else if (piped && c->ablate == 8) SK_LAUNCH_GRID(false, 8, true);
    else if (piped && c->ablate == 9) SK_LAUNCH_GRID(false, 9, true);
#endif
    else if (piped)          SK_LAUNCH_GRID(false, 0, true);
    else if (tally_sink)     SK_LAUNCH_GRID(true, 0, false);
#ifdef SK_EXPERIMENTS
    else if (c->ablate == 1) SK_LAUNCH_GRID(false, 1, false);
    else if (c->ablate == 2) SK_LAUNCH_GRID(false, 2, false);
    else if (c->ablate == 3) SK_LAUNCH_GRID(false, 3, false);
    else if (c->ablate == 4) SK_LAUNCH_GRID(false, 4, false);
    else if (c->ablate == 5) SK_LAUNCH_GRID(false, 5, false);
    else if (c->ablate == 6) SK_LAUNCH_GRID(false, 6, false);
    else if (c->ablate == 10) SK_LAUNCH_GRID(false, 10, false);
#endif
    else                     SK_LAUNCH_GRID(false, 0, false);
#undef SK_LAUNCH_GRID
    if (timed) {
        SK_HIP(c, hipEventRecord(e1, c->stream));
        c->ev.push_back(e0);
        c->ev.push_back(e1);
    }
    uint64_t wblocks = (nbytes - emit_begin + 255) / 256;
    if (wblocks > 2048) wblocks = 2048;                        // (grid-stride: eight workgroups per CU; with nothing to do -- the usual case -- the launch is over in 3 us)
    if (wblocks == 0) { wblocks = 1; }                        // (the kernel also readies the next launch's flag words)
    if (tally_sink && tally_sink->ns)
        hipLaunchKernelGGL((sk_scan_wide<true, true>), dim3((uint32_t)wblocks), dim3(256), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, wv, sink, d_fl, d_fl_next);
    else if (tally_sink)
        hipLaunchKernelGGL(sk_scan_wide<true>, dim3((uint32_t)wblocks), dim3(256), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, wv, sink, d_fl, d_fl_next);
    else
        hipLaunchKernelGGL(sk_scan_wide<false>, dim3((uint32_t)wblocks), dim3(256), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, wv, sink, d_fl, d_fl_next);
    SK_HIP(c, hipGetLastError());
    return SK_OK;
}

// grow-only scratch buffer of the context
static int sk_scratch(sk_ctx *c, void **p, size_t *cap, size_t need)
{
    if (need <= *cap) return SK_OK;
    if (*p) { SK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(*p); *p = NULL; *cap = 0; }
    size_t want = need + need / 4 + 4096;
    SK_HIP(c, hipMalloc(p, want));
    *cap = want;
    return SK_OK;
}

// ---------------------------------------------------------------------------------------------
// strain_detect: a batch of records resident on the device, tallied against any number of tables
// ---------------------------------------------------------------------------------------------
struct sk_batch {
    sk_ctx      *owner;
    hipStream_t  stream;
    hipEvent_t   ready;                 // the upload of the current contents
    void        *d_stream, *d_rec;      // bytes; rec_start[nrec] followed by tile_first[ntiles + 2]
    size_t       stream_cap, rec_cap;
    uint64_t     nbytes;
    uint32_t     nrec, ntiles;
    bool         packed;                // d_stream holds the host-packed form (sk_pack_stream) of nbytes bytes
    uint32_t    *h_rec;                 // page-locked staging of rec_start[nrec] + tile_first[ntiles + 2]: ONE copy from pinned memory instead of
    size_t       h_rec_cap;             // two from the caller's pageable arrays (which the runtime stages, chunk by chunk, on its own threads)
};

extern "C" int sk_batch_create(sk_ctx *c, sk_batch **out)
{
    if (!c || !out) return SK_E_ARG;
    *out = NULL;
    SK_HIP(c, hipSetDevice(c->device));
    sk_batch *b = new (std::nothrow) sk_batch();
    if (!b) return SK_E_NOMEM;
    b->owner = c;
    b->d_stream = b->d_rec = NULL; b->stream_cap = b->rec_cap = 0; b->nbytes = 0; b->nrec = b->ntiles = 0;
    b->h_rec = NULL; b->h_rec_cap = 0; b->packed = false;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { delete b; return sk_fail(c, SK_E_HIP, "stream"); }
    if (hipEventCreateWithFlags(&b->ready, hipEventDisableTiming) != hipSuccess) { hipStreamDestroy(b->stream); delete b; return sk_fail(c, SK_E_HIP, "event"); }
    *out = b;
    return SK_OK;
}

// waits until the batch's uploads are done: its device buffers may be refilled and the host memory it was filled from reused
extern "C" int sk_batch_sync(sk_batch *b)
{
    if (!b) return SK_E_ARG;
    SK_HIP(b->owner, hipSetDevice(b->owner->device));
    SK_HIP(b->owner, hipStreamSynchronize(b->stream));
    return SK_OK;
}

extern "C" void sk_batch_destroy(sk_batch *b)
{
    if (!b) return;
    hipSetDevice(b->owner->device);
    hipStreamSynchronize(b->stream);
    hipFree(b->d_stream); hipFree(b->d_rec);
    if (b->h_rec) hipHostFree(b->h_rec);
    hipEventDestroy(b->ready);
    hipStreamDestroy(b->stream);
    delete b;
}

// Upload new contents.  Every tally launched on the previous contents must have been collected.
static int sk_batch_fill_any(sk_batch *b, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec, bool packed);
extern "C" int sk_batch_fill(sk_batch *b, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec)
{
    return sk_batch_fill_any(b, stream, nbytes, rec_start, nrec, false);
}
// the batch in the host-packed form (sk_pack_stream: 6 bytes per 16 of the byte stream; no byte for the byte-string kernel in it);
// nbytes and rec_start are those of the BYTE stream it was packed from
extern "C" int sk_batch_fill_packed(sk_batch *b, const void *packed, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec)
{
    return sk_batch_fill_any(b, (const uint8_t *)packed, nbytes, rec_start, nrec, true);
}
static int sk_batch_fill_any(sk_batch *b, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec, bool packed)
{
    if (!b || !stream || !rec_start) return SK_E_ARG;
    sk_ctx *c = b->owner;
    if (nbytes == 0 || nrec == 0 || nbytes > 0xFFFFFFF0ull) return sk_fail(c, SK_E_ARG, "bad batch size");
    SK_HIP(c, hipSetDevice(c->device));
    const uint32_t ntiles = (uint32_t)((nbytes + 32767u) >> 15);
    if (nbytes + 16 > b->stream_cap) {
        SK_HIP(c, hipStreamSynchronize(b->stream));
        (void)hipFree(b->d_stream); b->d_stream = NULL; b->stream_cap = 0;
        const size_t want = nbytes + nbytes / 4 + 4096;
        SK_HIP(c, hipMalloc(&b->d_stream, want));
        b->stream_cap = want;
    }
    const size_t rec_need = ((size_t)nrec + ntiles + 2) * 4;
    if (rec_need > b->rec_cap) {
        SK_HIP(c, hipStreamSynchronize(b->stream));
        (void)hipFree(b->d_rec); b->d_rec = NULL; b->rec_cap = 0;
        const size_t want = rec_need + rec_need / 4 + 4096;
        SK_HIP(c, hipMalloc(&b->d_rec, want));
        b->rec_cap = want;
    }
    SK_HIP(c, hipStreamSynchronize(b->stream));            // the staging of the previous upload is free again
    if (rec_need > b->h_rec_cap) {
        if (b->h_rec) (void)hipHostFree(b->h_rec);
        b->h_rec = NULL; b->h_rec_cap = 0;
        const size_t want = rec_need + rec_need / 4 + 4096;
        SK_HIP(c, hipHostMalloc((void **)&b->h_rec, want, hipHostMallocDefault));
        b->h_rec_cap = want;
    }
    memcpy(b->h_rec, rec_start, (size_t)nrec * 4);
    uint32_t *const tile_first = b->h_rec + nrec;
    for (uint32_t t = 0, r = 0; t < ntiles + 2; t++) {     // first record starting at or after the tile's first byte
        const uint64_t edge = (uint64_t)t << 15;
        while (r < nrec && rec_start[r] < edge) r++;
        tile_first[t] = r;
    }
    SK_HIP(c, hipMemcpyAsync(b->d_stream, stream, packed ? ((nbytes + 15u) >> 4) * 6u : nbytes, hipMemcpyHostToDevice, b->stream));
    b->packed = packed;
    SK_HIP(c, hipMemcpyAsync(b->d_rec, b->h_rec, ((size_t)nrec + ntiles + 2) * 4, hipMemcpyHostToDevice, b->stream));
    SK_HIP(c, hipEventRecord(b->ready, b->stream));
    b->nbytes = nbytes; b->nrec = nrec; b->ntiles = ntiles;
    return SK_OK;
}

// Start the tallies of batch `b` against the table of context `c` (any context on the batch's device);
// returns at once.  One launch per context may be in flight.
extern "C" int sk_tally_launch(sk_ctx *c, const sk_batch *b, uint32_t type_col, uint32_t informative_value, uint64_t hits_cap)
{
    if (!c || !b) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (type_col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", type_col);
    if (b->owner->device != c->device) return sk_fail(c, SK_E_ARG, "batch lives on another device");
    if (b->nrec == 0) return sk_fail(c, SK_E_STATE, "empty batch");
    SK_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = sk_diff_flush(c)) != SK_OK) return rc;              // (the type column is read by the kernel)
    const uint32_t nrec = b->nrec;
    if ((rc = sk_scratch(c, &c->t_tally, &c->t_tally_cap, (size_t)nrec * 8 + 16)) != SK_OK) return rc;
    if ((rc = sk_scratch(c, &c->t_compact, &c->t_compact_cap, (size_t)nrec * 12 + 16)) != SK_OK) return rc;
    if ((rc = sk_scratch(c, &c->t_hits, &c->t_hits_cap, (size_t)(hits_cap ? hits_cap : 1) * sizeof(uint2))) != SK_OK) return rc;
    if ((size_t)nrec * 8 + 16 > c->h_tally_cap) {
        if (c->h_tally) { SK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_tally); c->h_tally = NULL; c->h_tally_cap = 0; }
        const size_t want = (size_t)nrec * 10 + 4096;
        SK_HIP(c, hipHostMalloc((void **)&c->h_tally, want, hipHostMallocDefault));
        c->h_tally_cap = want;
    }
    unsigned long long *d_n = (unsigned long long *)((uint8_t *)c->t_tally + (size_t)nrec * 8);   // hit counter behind the tallies
    SK_HIP(c, hipStreamWaitEvent(c->stream, b->ready, 0));
    SK_HIP(c, hipMemsetAsync(c->t_tally, 0, (size_t)nrec * 8 + 16, c->stream));
    sk_sink sink;
    memset(&sink, 0, sizeof sink);
    sink.rec_start = (const uint32_t *)b->d_rec; sink.nrec = nrec; sink.tally = (uint32_t *)c->t_tally;
    sink.tile_first = (const uint32_t *)b->d_rec + nrec;
    sink.type = c->d_counts + (size_t)type_col * c->nrows; sink.inf_value = informative_value;
    sink.hits = (uint2 *)c->t_hits; sink.nhits = d_n; sink.hits_cap = hits_cap; sink.inv = c->d_inv;
    if (!c->infbits_ok || c->infbits_col != type_col || c->infbits_val != informative_value) {
        if (!c->d_infbits) SK_HIP(c, hipMalloc((void **)&c->d_infbits, ((size_t)c->nrows / 32 + 8) * 4));
        SK_HIP(c, hipMemsetAsync(c->d_infbits, 0, ((size_t)c->nrows / 32 + 8) * 4, c->stream));
        hipLaunchKernelGGL(sk_inf_bitmap, dim3((c->nrows + 255) / 256), dim3(256), 0, c->stream, sink.type, c->nrows, informative_value, c->d_infbits);
        c->infbits_ok = true; c->infbits_col = type_col; c->infbits_val = informative_value;
    }
    sink.infbits = c->d_infbits;
    rc = sk_launch_scan(c, (const uint8_t *)b->d_stream, b->nbytes, 0, 0, &sink, b->packed ? (const uint8_t *)b->d_stream + ((b->nbytes + 15u) >> 4) * 4u : NULL);
    if (rc) return rc;
    // the records that were hit at all, compacted on the device (sk_tally_collect_sparse); only the two counters come
    // back now, the tallies themselves when they are asked for
    hipLaunchKernelGGL(sk_tally_compact, dim3((nrec + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)c->t_tally, nrec,
                       (uint32_t *)c->t_compact, d_n + 1);
    SK_HIP(c, hipMemcpyAsync(c->h_tally, d_n, 16, hipMemcpyDeviceToHost, c->stream));
    c->t_inflight_nrec = nrec;
    c->t_inflight_cap = hits_cap;
    return SK_OK;
}

// Wait for the launch of this context and hand out its results.  *out_nhits may exceed hits_cap (log
// overflow): launch again with more room.
extern "C" int sk_tally_collect(sk_ctx *c, uint32_t *out_tally, sk_hit *out_hits, uint64_t *out_nhits)
{
    if (!c || !out_tally || !out_nhits) return SK_E_ARG;
    if (!c->t_inflight_nrec) return sk_fail(c, SK_E_STATE, "no tally in flight");
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t nrec = c->t_inflight_nrec;
    c->t_inflight_nrec = 0;
    SK_HIP(c, hipMemcpy(out_tally, c->t_tally, (size_t)nrec * 8, hipMemcpyDeviceToHost));
    unsigned long long nh;
    memcpy(&nh, c->h_tally, 8);
    const unsigned long long take = nh < c->t_inflight_cap ? nh : c->t_inflight_cap;
    if (take && !out_hits) return SK_E_ARG;
    if (take) SK_HIP(c, hipMemcpy(out_hits, c->t_hits, (size_t)take * sizeof(uint2), hipMemcpyDeviceToHost));
    *out_nhits = nh;
    return SK_OK;
}

// Like sk_tally_collect, but only the records with at least one hit come back: out[i] = {record, all hits, informative
// hits}, unordered, *n of them (if *n > cap only cap were stored: collect again is not possible -- size cap to the batch's
// record count to be safe, or to what the workload allows).
extern "C" int sk_tally_collect_sparse(sk_ctx *c, sk_tally_rec *out, uint64_t cap, uint64_t *n, sk_hit *out_hits, uint64_t *out_nhits)
{
    if (!c || !n || !out_nhits || (cap && !out)) return SK_E_ARG;
    if (!c->t_inflight_nrec) return sk_fail(c, SK_E_STATE, "no tally in flight");
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    c->t_inflight_nrec = 0;
    unsigned long long two[2];
    memcpy(two, c->h_tally, 16);
    const unsigned long long take_r = two[1] < cap ? two[1] : cap;
    if (take_r) SK_HIP(c, hipMemcpy(out, c->t_compact, (size_t)take_r * sizeof(sk_tally_rec), hipMemcpyDeviceToHost));
    *n = two[1];
    const unsigned long long take = two[0] < c->t_inflight_cap ? two[0] : c->t_inflight_cap;
    if (take && !out_hits) return SK_E_ARG;
    if (take) SK_HIP(c, hipMemcpy(out_hits, c->t_hits, (size_t)take * sizeof(uint2), hipMemcpyDeviceToHost));
    *out_nhits = two[0];
    return SK_OK;
}

// Per-record tallies of one batch, synchronous: upload + launch + collect on the context's own batch.
extern "C" int sk_tally_batch(sk_ctx *c, const uint8_t *stream, uint64_t nbytes, const uint32_t *rec_start, uint32_t nrec,
                              uint32_t type_col, uint32_t informative_value, uint32_t *out_tally,
                              sk_hit *out_hits, uint64_t hits_cap, uint64_t *out_nhits)
{
    if (!c || !stream || !rec_start || !out_tally || !out_nhits || (hits_cap && !out_hits)) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    int rc;
    if (!c->own_batch && (rc = sk_batch_create(c, &c->own_batch)) != SK_OK) return rc;
    if ((rc = sk_batch_fill(c->own_batch, stream, nbytes, rec_start, nrec)) != SK_OK) return rc;
    if ((rc = sk_tally_launch(c, c->own_batch, type_col, informative_value, hits_cap)) != SK_OK) return rc;
    return sk_tally_collect(c, out_tally, out_hits, out_nhits);
}


#include "sk_dev_union.hip.h"         // the union table's kernels

struct sk_union {
    sk_ctx   *uc;                    // the union as a context of its own (table, text, filters, scratch, stream)
    uint32_t  n;
    void     *d_tally, *d_flag, *d_raw;       // dense (record, strain) tallies (all zero between launches), touched marks, raw hit log
    size_t    tally_cap, flag_cap, raw_cap;
    unsigned long long *d_cnt;       // two sets of four taking turns: [0] raw log entries, [1] compacted pairs, [2] dealt-out log entries
    uint32_t  cnt_set;               // the set the launch in flight (or the last one) counts in
    bool      cnt_clean;             // both sets are zero but for what the last launch's sk_union_ship leaves zero for the next
    uint64_t *d_ukeys;               // [rows] key of every global row
    uint2    *d_umask;               // [rows]
    sk_union_member *d_members;
};

extern "C" void sk_union_destroy(sk_union *u)
{
    if (!u) return;
    if (u->uc) {
        hipSetDevice(u->uc->device);
        hipStreamSynchronize(u->uc->stream);
        hipFree(u->d_ukeys); hipFree(u->d_umask); hipFree(u->d_members);
        hipFree(u->d_tally); hipFree(u->d_flag); hipFree(u->d_raw); hipFree(u->d_cnt);
        sk_ctx_destroy(u->uc);
    }
    delete u;
}

extern "C" int sk_union_create(sk_ctx *const *members, uint32_t n, uint32_t type_col, uint32_t informative_value, sk_union **out)
{
    if (!members || !out || n < 1 || n > SK_UNION_MAX) return SK_E_ARG;
    *out = NULL;
    sk_ctx *first = members[0];
    if (!first) return SK_E_ARG;
    uint64_t rows = 0, tbases = 0;
    for (uint32_t s = 0; s < n; s++) {
        sk_ctx *m = members[s];
        if (!m) return SK_E_ARG;
        if (m->device != first->device) return sk_fail(first, SK_E_ARG, "the members of a union live on one device");
        if (!m->d_keys || !m->nrows) return sk_fail(first, SK_E_STATE, "member %u has no table", s);
        if (type_col >= m->ncols) return sk_fail(first, SK_E_ARG, "column %u out of range", type_col);
        if (m->nwide) return sk_fail(first, SK_E_STATE, "member %u has byte-string keys: no union", s);
        if (!m->d_text2 || !m->d_rank || m->no_text) return sk_fail(first, SK_E_STATE, "member %u has no text stage: no union", s);
        if (m->nrows >= (1u << SK_UNION_ROW_BITS) - 1u) return sk_fail(first, SK_E_STATE, "member %u has too many rows for the union's hit log", s);
        rows += m->nrows;
        tbases += ((uint64_t)m->text_bases + 63u) / 64u * 64u + 64u;
    }
    if (rows > 0x7FFFFFF0ull || tbases > 0x7FFFFF00ull) return sk_fail(first, SK_E_STATE, "union too large (%llu rows, %llu bases)", (unsigned long long)rows, (unsigned long long)tbases);
    SK_HIP(first, hipSetDevice(first->device));
    sk_union *u = new (std::nothrow) sk_union();
    if (!u) return SK_E_NOMEM;
    u->uc = NULL; u->n = n; u->d_ukeys = NULL; u->d_umask = NULL; u->d_members = NULL;
    u->d_tally = u->d_flag = u->d_raw = NULL; u->tally_cap = u->flag_cap = u->raw_cap = 0; u->d_cnt = NULL;
    int rc = sk_ctx_create(&u->uc, first->device);
    if (rc != SK_OK) { delete u; return rc; }
    sk_ctx *c = u->uc;
#define SK_U(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { sk_fail(first, SK_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); sk_union_destroy(u); return SK_E_HIP; } } while (0)
    const uint32_t nrows = (uint32_t)rows;
    uint32_t lg = 10;
    while (((uint64_t)1 << lg) * (uint64_t)first->table_load_pct < rows * 100ull && lg < 31) lg++;
    const uint64_t slots = (uint64_t)1 << lg;
    c->slots_log2 = lg;
    c->nrows = nrows; c->ncols = 1;
    SK_U(hipMalloc((void **)&c->d_keys, slots * sizeof(sk_u4)));
    SK_U(hipMalloc((void **)&u->d_ukeys, (size_t)nrows * 8));
    SK_U(hipMalloc((void **)&u->d_umask, (size_t)nrows * 8));
    uint32_t *d_canon = NULL;
    SK_U(hipMalloc((void **)&d_canon, (size_t)nrows * 4));
    hipLaunchKernelGGL(sk_fill64, dim3(2048), dim3(256), 0, c->stream, (uint64_t *)c->d_keys, 2 * slots, SK_EMPTY64);
    hipLaunchKernelGGL(sk_fill64, dim3(2048), dim3(256), 0, c->stream, u->d_ukeys, (uint64_t)nrows, SK_EMPTY64);
    SK_U(hipMemsetAsync(u->d_umask, 0, (size_t)nrows * 8, c->stream));
    // text and rank map, member behind member
    const uint32_t text_bases = (uint32_t)tbases;
    const size_t words = (size_t)text_bases / 16 + 4, nblk = (size_t)text_bases / 64 + 2;
    SK_U(hipMalloc((void **)&c->d_text2, words * 4));
    SK_U(hipMalloc((void **)&c->d_rank, nblk * sizeof(sk_u4)));
    SK_U(hipMemsetAsync(c->d_text2, 0, words * 4, c->stream));
    std::vector<sk_union_member> hm(n);
    uint32_t base = 0, tbase = 0;
    for (uint32_t s = 0; s < n; s++) {
        sk_ctx *m = members[s];
        int frc = sk_diff_flush(m);                         // (the type column is read below)
        if (frc != SK_OK) { hipFree(d_canon); sk_union_destroy(u); return frc; }
        SK_U(hipStreamSynchronize(m->stream));
        const uint64_t mslots = (uint64_t)1 << m->slots_log2;
        hipLaunchKernelGGL(sk_union_insert, dim3(4096), dim3(256), 0, c->stream, (const sk_u4 *)m->d_keys, mslots, base, tbase,
                           c->d_keys, (uint32_t)(slots - 1), u->d_ukeys);
        const uint32_t span = (uint32_t)(((uint64_t)m->text_bases + 63u) / 64u * 64u + 64u);
        SK_U(hipMemcpyAsync(c->d_text2 + tbase / 16u, m->d_text2, (((size_t)m->text_bases + 15) / 16) * 4, hipMemcpyDeviceToDevice, c->stream));
        const uint32_t nsrc = m->text_bases / 64u + 1u, ndst = span / 64u;
        hipLaunchKernelGGL(sk_union_rank_copy, dim3((ndst + 255) / 256), dim3(256), 0, c->stream, c->d_rank + tbase / 64u, (const sk_u4 *)m->d_rank,
                           nsrc < ndst ? nsrc : ndst, ndst, base, base + m->nrows);
        hm[s].slots = m->d_keys; hm[s].mask = (uint32_t)(mslots - 1); hm[s].inv = m->d_inv;
        base += m->nrows;
        tbase += span;
    }
    hipLaunchKernelGGL(sk_union_rank_copy, dim3(1), dim3(256), 0, c->stream, c->d_rank + tbase / 64u, (const sk_u4 *)c->d_rank, 0u, (uint32_t)(nblk - tbase / 64u), 0u, nrows);
    base = 0;
    for (uint32_t s = 0; s < n; s++) {
        sk_ctx *m = members[s];
        hipLaunchKernelGGL(sk_union_mask_a, dim3(4096), dim3(256), 0, c->stream, (const sk_u4 *)m->d_keys, (uint64_t)1 << m->slots_log2, base, s,
                           (const uint32_t *)(m->d_counts + (size_t)type_col * m->nrows), informative_value,
                           (const sk_u4 *)c->d_keys, (uint32_t)(slots - 1), u->d_umask, d_canon);
        base += m->nrows;
    }
    hipLaunchKernelGGL(sk_union_mask_b, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, u->d_umask, (const uint32_t *)d_canon, nrows);
    {   // the filters, sized for all the keys (level 1 no longer fits the L2: its misses go to the Infinity Cache)
        uint64_t kib = first->grid_kib > 0 ? (uint64_t)first->grid_kib * n : ((uint64_t)nrows * 23ull / 32ull + 1023ull) / 1024ull;
        if (kib > (1ull << 22)) kib = 1ull << 22;
        if (kib < 4ull) kib = 4ull;
        c->grid1_blocks = (uint32_t)(kib * 1024ull / sizeof(uint2));
        uint32_t g2 = 12;
        while (g2 < 37 && ((uint64_t)1 << g2) < (uint64_t)nrows * 32ull) g2++;
        c->grid2_blocks_log2 = g2 - 6u;
        const size_t b1 = (size_t)c->grid1_blocks * sizeof(uint2), b2 = ((size_t)1 << c->grid2_blocks_log2) * sizeof(uint2);
        SK_U(hipMalloc((void **)&c->d_grid1, b1));
        SK_U(hipMalloc((void **)&c->d_grid2, b2));
        SK_U(hipMemsetAsync(c->d_grid1, 0, b1, c->stream));
        SK_U(hipMemsetAsync(c->d_grid2, 0, b2, c->stream));
        bool from_text = true;                              // (every member's keys are windows of its text: one insert per base of the union's text)
        for (uint32_t s2 = 0; s2 < n; s2++) from_text = from_text && members[s2]->all_rows_in_text;
        if (from_text)
            hipLaunchKernelGGL(sk_grid_insert_text, dim3((text_bases + 255) / 256), dim3(256), 0, c->stream, (const uint32_t *)c->d_text2, text_bases,
                               (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
        else
            hipLaunchKernelGGL(sk_grid_insert, dim3((nrows + 255) / 256), dim3(256), 0, c->stream, (const uint64_t *)u->d_ukeys, nrows,
                               (uint32_t *)c->d_grid1, c->grid1_blocks, (uint32_t *)c->d_grid2, 32u - c->grid2_blocks_log2);
    }
    SK_U(hipMalloc((void **)&u->d_members, n * sizeof(sk_union_member)));
    SK_U(hipMemcpyAsync(u->d_members, hm.data(), n * sizeof(sk_union_member), hipMemcpyHostToDevice, c->stream));
    SK_U(hipStreamSynchronize(c->stream));
    hipFree(d_canon);
    SK_U(hipGetLastError());
#undef SK_U
    c->text_bases = text_bases;
    *out = u;
    return SK_OK;
}

// grow-only device buffer of a union that must read as zero when it is handed out
static int sk_union_zeroed(sk_union *u, void **p, size_t *cap, size_t need)
{
    sk_ctx *c = u->uc;
    if (need <= *cap) return SK_OK;
    if (*p) { SK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(*p); *p = NULL; *cap = 0; }
    const size_t want = need + need / 4 + 4096;
    SK_HIP(c, hipMalloc(p, want));
    SK_HIP(c, hipMemsetAsync(*p, 0, want, c->stream));
    *cap = want;
    return SK_OK;
}

// Start the tallies of batch `b` against every member at once; returns at once.  Results: sk_union_tally_collect.
extern "C" int sk_union_tally_launch(sk_union *u, const sk_batch *b, uint64_t hits_cap)
{
    if (!u || !b) return SK_E_ARG;
    sk_ctx *c = u->uc;
    if (b->owner->device != c->device) return sk_fail(c, SK_E_ARG, "batch lives on another device");
    if (b->nrec == 0) return sk_fail(c, SK_E_STATE, "empty batch");
    const uint64_t vrec = (uint64_t)b->nrec * u->n;
    if (vrec > 0x7FFFFFF0ull) return sk_fail(c, SK_E_ARG, "too many records x members");
    SK_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = sk_union_zeroed(u, &u->d_tally, &u->tally_cap, (size_t)vrec * 8 + 16)) != SK_OK) return rc;
    if ((rc = sk_union_zeroed(u, &u->d_flag, &u->flag_cap, (size_t)b->nrec + 16)) != SK_OK) return rc;
    if (!u->d_cnt) { SK_HIP(c, hipMalloc((void **)&u->d_cnt, 64)); u->cnt_clean = false; }
    if ((rc = sk_scratch(c, &u->d_raw, &u->raw_cap, (size_t)(hits_cap ? hits_cap : 1) * sizeof(uint2))) != SK_OK) return rc;
    if ((rc = sk_scratch(c, &c->t_compact, &c->t_compact_cap, (size_t)vrec * 12 + 16)) != SK_OK) return rc;
    if ((rc = sk_scratch(c, &c->t_hits, &c->t_hits_cap, (size_t)(hits_cap ? hits_cap : 1) * sizeof(uint2))) != SK_OK) return rc;
    // the page-locked landing area: the three counters, then room for the first SK_UNION_EAGER records and log entries, which come
    // back WITH the counters (round 4: collecting used to be three dependent trips -- wait, copy the records, copy the log, the
    // last two through pageable memory; a 32 MiB chunk against 32 strains brings ~4 K records and ~5 K entries, far below the room)
    const size_t land = 64 + (size_t)SK_UNION_EAGER * (sizeof(sk_tally_rec) + sizeof(uint2));
    if (c->h_tally_cap < land) {
        if (c->h_tally) { SK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_tally); c->h_tally = NULL; c->h_tally_cap = 0; }
        SK_HIP(c, hipHostMalloc((void **)&c->h_tally, land, hipHostMallocDefault));
        c->h_tally_cap = land;
    }
    SK_HIP(c, hipStreamWaitEvent(c->stream, b->ready, 0));
    if (!u->cnt_clean) {                                          // (the first launch, or one after a launch that failed half-way)
        SK_HIP(c, hipMemsetAsync(u->d_cnt, 0, 64, c->stream));
        u->cnt_set = 1;
    }
    u->cnt_clean = false;
    u->cnt_set ^= 1u;
    unsigned long long *const d_cnt = u->d_cnt + 4u * u->cnt_set, *const d_cnt_next = u->d_cnt + 4u * (u->cnt_set ^ 1u);
    uint8_t *d_land = NULL;
    SK_HIP(c, hipHostGetDevicePointer((void **)&d_land, c->h_tally, 0));
    sk_sink sink;
    memset(&sink, 0, sizeof sink);
    sink.rec_start = (const uint32_t *)b->d_rec; sink.nrec = b->nrec; sink.tally = (uint32_t *)u->d_tally;
    sink.tile_first = (const uint32_t *)b->d_rec + b->nrec;
    sink.hits = (uint2 *)u->d_raw; sink.nhits = d_cnt; sink.hits_cap = hits_cap;
    sink.type = (const uint32_t *)u->d_umask; sink.ns = u->n;
    sink.infbits = (const uint32_t *)u->d_flag;                   // (sk_uflag: the records' touched marks)
    rc = sk_launch_scan(c, (const uint8_t *)b->d_stream, b->nbytes, 0, 0, &sink, b->packed ? (const uint8_t *)b->d_stream + ((b->nbytes + 15u) >> 4) * 4u : NULL);
    if (rc) return rc;
    {   // compact the tallies and deal the log out, in one launch
        uint32_t sets = (b->nrec + 256u * 256u - 1u) / (256u * 256u);          // (about 256 compacting workgroups)
        if (sets < 1u) sets = 1u;
        if (sets > SK_UC_SETS_MAX) sets = SK_UC_SETS_MAX;
        const uint32_t ncompact = (b->nrec + 256u * sets - 1u) / (256u * sets);
        sk_union_resolve_args ra;
        ra.raw = (const uint2 *)u->d_raw; ra.nraw = d_cnt; ra.cap_raw = hits_cap;
        ra.out = (uint2 *)c->t_hits; ra.nout = d_cnt + 2; ra.cap_out = hits_cap;
        ra.ukeys = (const uint64_t *)u->d_ukeys; ra.umask = (const uint2 *)u->d_umask; ra.mem = (const sk_union_member *)u->d_members;
        hipLaunchKernelGGL(sk_union_post, dim3(ncompact + 256u), dim3(256), 0, c->stream, (uint2 *)u->d_tally, (uint8_t *)u->d_flag,
                           b->nrec, u->n, (uint32_t *)c->t_compact, d_cnt + 1, sets, ncompact, ra);
    }
    // the counters and the first SK_UNION_EAGER results go home in one kernel (and the other set of counters is zeroed for the next launch)
    hipLaunchKernelGGL(sk_union_ship, dim3(32), dim3(256), 0, c->stream, (const unsigned long long *)d_cnt, d_cnt_next,
                       (const uint32_t *)c->t_compact, (unsigned long long)vrec, (const uint2 *)c->t_hits, (unsigned long long)hits_cap,
                       (uint32_t)SK_UNION_EAGER, d_land);
    SK_HIP(c, hipGetLastError());
    u->cnt_clean = true;
    c->t_inflight_nrec = (uint32_t)vrec;
    c->t_inflight_cap = hits_cap;
    return SK_OK;
}

// out[i] = {record * members + member, all hits, informative hits} for the pairs with at least one hit (unordered);
// out_hits[j] = {window-end offset, member << 27 | the member's own row}.  *out_nhits may exceed the launch's hits_cap:
// launch again with more room.
extern "C" int sk_union_tally_collect(sk_union *u, sk_tally_rec *out, uint64_t cap, uint64_t *n, sk_hit *out_hits, uint64_t *out_nhits)
{
    if (!u || !n || !out_nhits || (cap && !out)) return SK_E_ARG;
    sk_ctx *c = u->uc;
    if (!c->t_inflight_nrec) return sk_fail(c, SK_E_STATE, "no tally in flight");
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    c->t_inflight_nrec = 0;
    unsigned long long cnt[3];
    memcpy(cnt, c->h_tally, 24);
    const unsigned long long take_r = cnt[1] < cap ? cnt[1] : cap;
    {   // what came back with the counters, then (rarely) the rest
        const unsigned long long have = take_r < SK_UNION_EAGER ? take_r : SK_UNION_EAGER;
        if (have) memcpy(out, c->h_tally + 64, (size_t)have * sizeof(sk_tally_rec));
        if (take_r > have) SK_HIP(c, hipMemcpy(out + have, (const sk_tally_rec *)c->t_compact + have, (size_t)(take_r - have) * sizeof(sk_tally_rec), hipMemcpyDeviceToHost));
    }
    *n = cnt[1];
    // the raw log holds one entry per hit, the caller gets one per hit AND strain: if the raw log itself ran over, entries are
    // missing from the count below -- report at least one more than the room there was
    unsigned long long nh = cnt[2];
    if (cnt[0] > c->t_inflight_cap && nh <= c->t_inflight_cap) nh = cnt[0] > nh ? cnt[0] : c->t_inflight_cap + 1;
    const unsigned long long take = nh < c->t_inflight_cap ? nh : c->t_inflight_cap;
    if (take && !out_hits) return SK_E_ARG;
    {
        const unsigned long long have = take < SK_UNION_EAGER ? take : SK_UNION_EAGER;
        if (have) memcpy(out_hits, c->h_tally + 64 + (size_t)SK_UNION_EAGER * sizeof(sk_tally_rec), (size_t)have * sizeof(uint2));
        if (take > have) SK_HIP(c, hipMemcpy(out_hits + have, (const uint2 *)c->t_hits + have, (size_t)(take - have) * sizeof(uint2), hipMemcpyDeviceToHost));
    }
    *out_nhits = nh;
    return SK_OK;
}

// wait for a launch whose results nobody will collect (the batch it reads is about to be freed)
extern "C" int sk_union_sync(sk_union *u)
{
    if (!u) return SK_E_ARG;
    sk_ctx *c = u->uc;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    c->t_inflight_nrec = 0;
    return SK_OK;
}

extern "C" int sk_union_scan_timing(sk_union *u, double *total_ms, uint64_t *launches, int reset)
{
    return u ? sk_scan_timing(u->uc, total_ms, launches, reset) : SK_E_ARG;
}
extern "C" const char *sk_union_last_error(const sk_union *u) { return u ? u->uc->err : "no union"; }
extern "C" uint32_t sk_union_members(const sk_union *u) { return u ? u->n : 0u; }
extern "C" uint32_t sk_union_rows(const sk_union *u) { return u ? u->uc->nrows : 0u; }

extern "C" int sk_scan_device(sk_ctx *c, const void *dev_stream, uint64_t nbytes, uint32_t col)
{
    if (!c || (!dev_stream && nbytes)) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", col);
    if (((uintptr_t)dev_stream & 15u) != 0) return sk_fail(c, SK_E_ARG, "device stream must be 16-byte aligned");
    SK_HIP(c, hipSetDevice(c->device));
    return sk_launch_scan(c, (const uint8_t *)dev_stream, nbytes, 0, col);
}

// sk_scan_device for a batch that lies in device memory in the PACKED form (sk_pack_stream's layout: the code words, then the masks;
// 4-byte aligned): how fast the kernel is when its first phase only copies -- a side measurement of bench.py, not the headline's input.
extern "C" int sk_scan_device_packed(sk_ctx *c, const void *dev_packed, uint64_t nbytes, uint32_t col)
{
    if (!c || (!dev_packed && nbytes)) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", col);
    if (((uintptr_t)dev_packed & 3u) != 0) return sk_fail(c, SK_E_ARG, "device batch must be 4-byte aligned");
    SK_HIP(c, hipSetDevice(c->device));
    const uint64_t nch = (nbytes + 15u) >> 4;
    return sk_launch_scan(c, (const uint8_t *)dev_packed, nbytes, 0, col, NULL, (const uint8_t *)dev_packed + nch * 4u);
}

static int sk_stage_init(sk_ctx *c)
{
    if (c->h_stage[0]) return SK_OK;
    for (int i = 0; i < SK_NSTAGE; i++) {
        SK_HIP(c, hipHostMalloc((void **)&c->h_stage[i], SK_STAGE_BYTES, hipHostMallocDefault));
        SK_HIP(c, hipMalloc((void **)&c->d_stage[i], SK_STAGE_BYTES));
        SK_HIP(c, hipEventCreateWithFlags(&c->stage_done[i], hipEventDisableTiming));
    }
    SK_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    return SK_OK;
}

extern "C" int sk_scan_stream(sk_ctx *c, const uint8_t *stream, uint64_t nbytes, uint32_t col)
{
    if (!c || (!stream && nbytes)) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", col);
    SK_HIP(c, hipSetDevice(c->device));
    int rc = sk_stage_init(c);
    if (rc) return rc;
    // cut into staging-sized pieces; a piece after the first re-sends the k-1 bytes before it
    uint64_t done = 0;
    while (done < nbytes) {
        const uint64_t lead = done ? SK_OVERLAP : 0;
        uint64_t take = nbytes - done;
        if (take > SK_STAGE_BYTES - lead) take = SK_STAGE_BYTES - lead;
        const int b = c->stage_next;
        c->stage_next = (b + 1) % SK_NSTAGE;
        SK_HIP(c, hipEventSynchronize(c->stage_done[b]));       // buffer free again?
        memcpy(c->h_stage[b], stream + done - lead, lead + take);
        SK_HIP(c, hipMemcpyAsync(c->d_stage[b], c->h_stage[b], lead + take, hipMemcpyHostToDevice, c->stream));
        rc = sk_launch_scan(c, c->d_stage[b], lead + take, lead, col);
        if (rc) return rc;
        SK_HIP(c, hipEventRecord(c->stage_done[b], c->stream));
        done += take;
    }
    return SK_OK;
}

// ---- zero-copy variant for callers that fill PINNED host buffers themselves ---------------------
// Page-locked buffers are kept by the context and handed out again: locking pages is slow and serialised in the driver
// (measured on the MI355X box, tools/probes/pin_probe.hip: hipHostMalloc 6.1 ms per 32 MiB plus 3.7 ms to free it, 32 buffers
// from 16 threads 189 ms -- a list scan with 16 decode threads spent 0.3 s on its buffers, every call); registering ordinary
// memory costs 2.7 ms per 32 MiB, and a buffer given back stays registered until the context goes.  May be called from
// several threads at once.
extern "C" int sk_pinned_alloc(sk_ctx *c, void **p, uint64_t nbytes)
{
    if (!c || !p) return SK_E_ARG;
    const size_t want = (size_t)((nbytes ? nbytes : 16) + 4095u) & ~(size_t)4095u;
    pthread_mutex_lock(&c->pin_mu);
    for (sk_pin &q : c->pins)
        if (!q.used && q.n == want) { q.used = true; *p = q.p; pthread_mutex_unlock(&c->pin_mu); return SK_OK; }
    pthread_mutex_unlock(&c->pin_mu);
    if (hipSetDevice(c->device) != hipSuccess) return SK_E_HIP;
    void *m = NULL;
    bool registered = true;
    if (posix_memalign(&m, 4096, want) != 0) return SK_E_NOMEM;
    // (portable: every device of the process may DMA from it -- one decoded chunk goes up to several GPUs, sk_host_sd.c)
    if (hipHostRegister(m, want, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) {      // (a limit on registered memory, say): the slower way
        free(m);
        m = NULL;
        registered = false;
        (void)hipGetLastError();
        if (hipHostMalloc(&m, want, hipHostMallocPortable) != hipSuccess) return sk_fail(c, SK_E_NOMEM, "no page-locked memory (%zu bytes)", want);
    }
    pthread_mutex_lock(&c->pin_mu);
    c->pins.push_back((sk_pin){m, want, true, registered});
    pthread_mutex_unlock(&c->pin_mu);
    *p = m;
    return SK_OK;
}

extern "C" int sk_pinned_free(sk_ctx *c, void *p)
{
    if (!c) return SK_E_ARG;
    if (!p) return SK_OK;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return SK_E_HIP;   // (nothing reads it any more)
    int rc = SK_E_ARG;
    pthread_mutex_lock(&c->pin_mu);
    for (sk_pin &q : c->pins) if (q.p == p && q.used) { q.used = false; rc = SK_OK; break; }
    pthread_mutex_unlock(&c->pin_mu);
    return rc;
}

// Like sk_scan_stream, but `pinned` (from sk_pinned_alloc, at most 64 MiB - 64 bytes) is DMA-read in
// place: the caller must leave it alone until sk_ticket_wait(*ticket) returns.
extern "C" int sk_scan_pinned(sk_ctx *c, const uint8_t *pinned, uint64_t nbytes, uint32_t col, uint64_t *ticket)
{
    if (!c || (!pinned && nbytes) || !ticket) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", col);
    if (nbytes > SK_STAGE_BYTES - 64) return sk_fail(c, SK_E_ARG, "pinned batch larger than the staging buffer");
    SK_HIP(c, hipSetDevice(c->device));
    int rc = sk_stage_init(c);
    if (rc) return rc;
    const uint64_t t = c->tickets++;
    hipEvent_t &ev = c->copied[t & 63u];
    if (!ev) SK_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming | (getenv("SK_SYNC") ? hipEventBlockingSync : 0u)));
    else SK_HIP(c, hipEventSynchronize(ev));                     // ring slot of ticket t-64
    const int b = c->stage_next;
    c->stage_next = (b + 1) % SK_NSTAGE;
    // The copy goes on a stream of its own and waits THERE for the scan that read this device buffer last; the scan waits for the
    // copy.  So chunk n+1 is on the link while chunk n is scanned, and the caller (a decode thread holding the submit lock) does
    // not wait for the device at all -- its back-pressure is the ticket of its own two buffers.  (Round 4: copy and scan took
    // turns on one stream, 0.65 + 0.05 ms a chunk, and the submit waited on the host for the buffer two chunks back.)
    SK_HIP(c, hipStreamWaitEvent(c->copy_stream, c->stage_done[b], 0));       // (never recorded yet = done)
    if (nbytes) SK_HIP(c, hipMemcpyAsync(c->d_stage[b], pinned, nbytes, hipMemcpyHostToDevice, c->copy_stream));
    SK_HIP(c, hipEventRecord(ev, c->copy_stream));
    SK_HIP(c, hipStreamWaitEvent(c->stream, ev, 0));                          // (before anything else: sk_sync / sk_pinned_free wait on c->stream alone)
    rc = sk_launch_scan(c, c->d_stage[b], nbytes, 0, col);
    if (rc) return rc;
    SK_HIP(c, hipEventRecord(c->stage_done[b], c->stream));
    *ticket = t;
    return SK_OK;
}

// The same for a batch the host has PACKED (sk_pack_stream, sk_host.c: per 16-byte chunk of the byte stream the 32-bit code word and
// the 16-bit mask the scan kernel's own decode would make of it -- `packed` holds the (nbytes + 15) / 16 code words, then the masks):
// 6 bytes per 16 bases cross the PCIe link instead of 16, and the kernel's phase 1 only copies them into place.  nbytes is the
// length of the BYTE stream the batch was packed from (offsets, record ends and the batch's end mean what they mean there).  A batch
// with a byte that only the byte-string kernel can judge (sk_pack_stream says so) must be sent as bytes.
extern "C" int sk_scan_pinned_packed(sk_ctx *c, const void *packed, uint64_t nbytes, uint32_t col, uint64_t *ticket)
{
    if (!c || (!packed && nbytes) || !ticket) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", col);
    const uint64_t nch = (nbytes + 15u) >> 4;
    if (nch * 6u > SK_STAGE_BYTES) return sk_fail(c, SK_E_ARG, "packed batch larger than the staging buffer");
    SK_HIP(c, hipSetDevice(c->device));
    int rc = sk_stage_init(c);
    if (rc) return rc;
    const uint64_t t = c->tickets++;
    hipEvent_t &ev = c->copied[t & 63u];
    if (!ev) SK_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming | (getenv("SK_SYNC") ? hipEventBlockingSync : 0u)));
    else SK_HIP(c, hipEventSynchronize(ev));                     // ring slot of ticket t-64
    const int b = c->stage_next;
    c->stage_next = (b + 1) % SK_NSTAGE;
    SK_HIP(c, hipStreamWaitEvent(c->copy_stream, c->stage_done[b], 0));
    if (nch) SK_HIP(c, hipMemcpyAsync(c->d_stage[b], packed, nch * 6u, hipMemcpyHostToDevice, c->copy_stream));
    SK_HIP(c, hipEventRecord(ev, c->copy_stream));
    SK_HIP(c, hipStreamWaitEvent(c->stream, ev, 0));
    rc = sk_launch_scan(c, c->d_stage[b], nbytes, 0, col, NULL, c->d_stage[b] + nch * 4u);
    if (rc) return rc;
    SK_HIP(c, hipEventRecord(c->stage_done[b], c->stream));
    *ticket = t;
    return SK_OK;
}

extern "C" int sk_ticket_wait(sk_ctx *c, uint64_t ticket)
{
    if (!c || ticket >= c->tickets) return SK_E_ARG;
    if (c->tickets - ticket > 64) return SK_OK;                  // its ring slot was recycled only after it completed
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipEventSynchronize(c->copied[ticket & 63u]));
    return SK_OK;
}

extern "C" int sk_sync(sk_ctx *c)
{
    if (!c) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_counts_fetch(sk_ctx *c, uint32_t col, uint32_t *out)
{
    if (!c || !out) return SK_E_ARG;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    const uint32_t *src = c->d_counts + (size_t)col * c->nrows;
    if (c->d_perm && c->nrows) {
        hipLaunchKernelGGL(sk_gather_u32, dim3((c->nrows + 255) / 256), dim3(256), 0, c->stream, c->d_tmp, src, c->d_perm, c->nrows);
        src = c->d_tmp;
    }
    SK_HIP(c, hipMemcpyAsync(out, src, (size_t)c->nrows * 4, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_counts_set(sk_ctx *c, uint32_t col, const uint32_t *in)
{
    if (!c || !in) return SK_E_ARG;
    c->infbits_ok = false;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    uint32_t *dst = c->d_counts + (size_t)col * c->nrows;
    if (c->d_perm && c->nrows) {
        SK_HIP(c, hipMemcpyAsync(c->d_tmp, in, (size_t)c->nrows * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(sk_scatter_u32, dim3((c->nrows + 255) / 256), dim3(256), 0, c->stream, dst, c->d_tmp, c->d_perm, c->nrows);
    } else {
        SK_HIP(c, hipMemcpyAsync(dst, in, (size_t)c->nrows * 4, hipMemcpyHostToDevice, c->stream));
    }
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

// counts[col][rows[i]] = value for n rows (caller's row numbers): what a column of mostly one value needs instead of 4 bytes per row
// over PCIe (strain_detect's type column: "informative" for the 1 % of rows the -a list names; src/strain_detect.c:668-726)
extern "C" int sk_counts_set_rows(sk_ctx *c, uint32_t col, const uint32_t *rows, uint32_t n, uint32_t value)
{
    if (!c || (n && !rows)) return SK_E_ARG;
    c->infbits_ok = false;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    for (uint32_t i = 0; i < n; i++) if (rows[i] >= c->nrows) return sk_fail(c, SK_E_ARG, "row %u out of range", rows[i]);
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    if (!n) return SK_OK;
    uint32_t *d_rows = NULL;
    SK_HIP(c, hipMalloc((void **)&d_rows, (size_t)n * 4));
    hipError_t e = hipMemcpyAsync(d_rows, rows, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sk_set_rows_u32, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_counts + (size_t)col * c->nrows, (const uint32_t *)d_rows, n,
                           (const uint32_t *)c->d_perm, value);
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(d_rows);
    SK_HIP(c, e);
    return SK_OK;
}

extern "C" int sk_counts_zero(sk_ctx *c, uint32_t col)
{
    if (!c) return SK_E_ARG;
    c->infbits_ok = false;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    SK_HIP(c, hipMemsetAsync(c->d_counts + (size_t)col * c->nrows, 0, (size_t)c->nrows * 4, c->stream));
    return SK_OK;
}

extern "C" void *sk_counts_device_ptr(sk_ctx *c)
{
    if (!c) return NULL;
    // whoever takes the pointer reads the block on a stream of their own: fold the pending increments in and wait
    if (hipSetDevice(c->device) != hipSuccess || sk_diff_flush(c) != SK_OK || hipStreamSynchronize(c->stream) != hipSuccess) return NULL;
    return (void *)c->d_counts;
}

// ---- for the other translation units of the library (sk_internal.h)
extern "C" int sk_fail_(sk_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}
extern "C" int sk_ctx_device_(const sk_ctx *c) { return c->device; }
extern "C" int sk_counts_rows_to_device_(sk_ctx *c, uint32_t col, uint32_t *d_out)
{
    if (!c || !d_out) return SK_E_ARG;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    { int rc_ = sk_diff_flush(c); if (rc_) return rc_; }
    const uint32_t *src = c->d_counts + (size_t)col * c->nrows;
    if (c->d_perm && c->nrows)
        hipLaunchKernelGGL(sk_gather_u32, dim3((c->nrows + 255) / 256), dim3(256), 0, c->stream, d_out, src, c->d_perm, c->nrows);
    else
        SK_HIP(c, hipMemcpyAsync(d_out, src, (size_t)c->nrows * 4, hipMemcpyDeviceToDevice, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}
extern "C" uint32_t sk_table_rows(const sk_ctx *c) { return c ? c->nrows : 0; }
extern "C" uint32_t sk_table_cols(const sk_ctx *c) { return c ? c->ncols : 0; }

extern "C" int sk_scan_timing(sk_ctx *c, double *total_ms, uint64_t *launches, int reset)
{
    if (!c) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i + 1 < c->ev.size(); i += 2) {
        float ms = 0.f;
        SK_HIP(c, hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        c->timed_ms += ms;
        c->timed_launches++;
        c->ev_free.push_back(c->ev[i]);
        c->ev_free.push_back(c->ev[i + 1]);
    }
    c->ev.clear();
    if (total_ms) *total_ms = c->timed_ms;
    if (launches) *launches = c->timed_launches;
    if (reset) { c->timed_ms = 0; c->timed_launches = 0; }
    return SK_OK;
}

// ---------------------------------------------------------------------------------------------
// RCCL (resolved lazily so that the library loads on hosts without it)
// ---------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } sk_nccl_id;                       // ncclUniqueId
typedef int (*sk_nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*sk_nccl_getid_fn)(sk_nccl_id *);
typedef int (*sk_nccl_initrank_fn)(void **, int, sk_nccl_id, int);
typedef int (*sk_nccl_destroy_fn)(void *);
static struct { void *lib; sk_nccl_allreduce_fn allreduce; sk_nccl_getid_fn getid; sk_nccl_initrank_fn initrank; sk_nccl_destroy_fn destroy; } g_rccl;

static int sk_rccl_load(sk_ctx *c)
{
    if (g_rccl.lib) return SK_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return sk_fail(c, SK_E_RCCL, "cannot load librccl: %s", dlerror());
    g_rccl.allreduce = (sk_nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
    g_rccl.getid = (sk_nccl_getid_fn)dlsym(h, "ncclGetUniqueId");
    g_rccl.initrank = (sk_nccl_initrank_fn)dlsym(h, "ncclCommInitRank");
    g_rccl.destroy = (sk_nccl_destroy_fn)dlsym(h, "ncclCommDestroy");
    if (!g_rccl.allreduce || !g_rccl.getid || !g_rccl.initrank || !g_rccl.destroy) return sk_fail(c, SK_E_RCCL, "RCCL symbols missing");
    g_rccl.lib = h;
    return SK_OK;
}

// One process per GPU.  The ranks first exchange, through files next to `id_file` (sk_rendezvous.h: fresh-token handshake,
// every wait bounded), whether their own set-up worked and rank 0's RCCL unique id; only if everybody is fine does anyone
// enter ncclCommInitRank, and a watchdog ends the process if that call does not return in time (a rank that died after
// the exchange would otherwise hang the rest for good).  New relative to the reference, which is single-process.
struct sk_watchdog { pthread_mutex_t mu; pthread_cond_t cv; int done; int timeout_s; int rank; };
static void *sk_watchdog_main(void *arg)
{
    sk_watchdog *w = (sk_watchdog *)arg;
    struct timespec until;
    clock_gettime(CLOCK_REALTIME, &until);
    until.tv_sec += w->timeout_s;
    pthread_mutex_lock(&w->mu);
    while (!w->done) {
        if (pthread_cond_timedwait(&w->cv, &w->mu, &until) == ETIMEDOUT && !w->done) {
            fprintf(stderr, "sk_comm_init: ncclCommInitRank did not return within %d s on rank %d (did a rank die after the rendezvous?) -- giving up\n",
                    w->timeout_s, w->rank);
            fflush(stderr);
            _exit(3);                                  // (never an exec: this process has touched the GPU)
        }
    }
    pthread_mutex_unlock(&w->mu);
    return NULL;
}

extern "C" int sk_comm_init_ex(sk_ctx *c, int rank, int world, const char *id_file, int timeout_s, int setup_failed)
{
    if (world < 1 || world > SKR_MAX_WORLD || rank < 0 || rank >= world || !id_file || (!c && !setup_failed)) return SK_E_ARG;
    if (timeout_s < 1) timeout_s = 1;
    sk_nccl_id id;
    memset(&id, 0, sizeof id);
    int failed = setup_failed != 0;
    if (!failed && sk_rccl_load(c) != SK_OK) failed = 1;
    if (!failed && hipSetDevice(c->device) != hipSuccess) failed = 1;
    if (!failed && rank == 0 && g_rccl.getid(&id) != 0) { sk_fail(c, SK_E_RCCL, "ncclGetUniqueId failed"); failed = 1; }
    static_assert(sizeof id == SKR_PAYLOAD, "RCCL unique id size");
    const int x = skr_exchange(rank, world, id_file, failed, (unsigned char *)&id, (double)timeout_s);
    if (rank == 0) {                                   // (the others remove their own hello files)
        // the board stays until the ranks have read it; it carries this launch's tokens, so a later launch ignores it
    }
    if (x == SKR_ABORT) return sk_fail(c, SK_E_RCCL, failed ? "set-up failed on this rank; the other ranks were told to leave"
                                                            : "another rank reported a failed set-up: leaving before the collective");
    if (x == SKR_TIMEOUT) return sk_fail(c, SK_E_RCCL, "rendezvous through %s timed out after %d s (rank %d of %d)", id_file, timeout_s, rank, world);
    if (x != SKR_OK) return sk_fail(c, SK_E_RCCL, "rendezvous through %s failed (cannot write there?)", id_file);
    sk_watchdog w;
    pthread_mutex_init(&w.mu, NULL); pthread_cond_init(&w.cv, NULL);
    w.done = 0; w.timeout_s = timeout_s; w.rank = rank;
    pthread_t th;
    const bool watched = pthread_create(&th, NULL, sk_watchdog_main, &w) == 0;
    void *comm = NULL;
    const int nrc = g_rccl.initrank(&comm, world, id, rank);
    if (watched) {
        pthread_mutex_lock(&w.mu); w.done = 1; pthread_cond_signal(&w.cv); pthread_mutex_unlock(&w.mu);
        pthread_join(th, NULL);
    }
    pthread_mutex_destroy(&w.mu); pthread_cond_destroy(&w.cv);
    if (nrc != 0) return sk_fail(c, SK_E_RCCL, "ncclCommInitRank failed (rank %d of %d)", rank, world);
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    return SK_OK;
}

extern "C" int sk_comm_init(sk_ctx *c, int rank, int world, const char *id_file, int timeout_s)
{
    if (!c) return SK_E_ARG;
    return sk_comm_init_ex(c, rank, world, id_file, timeout_s, 0);
}

extern "C" void sk_comm_destroy(sk_ctx *c)
{
    if (c && c->comm && g_rccl.destroy) { g_rccl.destroy(c->comm); c->comm = NULL; }
}

// Sum a small host value over all ranks (agreement on "did anyone fail" before the big collective).
extern "C" int sk_comm_sum_u32(sk_ctx *c, uint32_t value, uint32_t *sum)
{
    if (!c || !sum) return SK_E_ARG;
    if (!c->comm) { *sum = value; return SK_OK; }
    SK_HIP(c, hipSetDevice(c->device));
    uint32_t *d = c->d_flags + 12;                     // spare words of the flag block
    SK_HIP(c, hipMemcpyAsync(d, &value, 4, hipMemcpyHostToDevice, c->stream));
    const int ncclUint32 = 3, ncclSum = 0;
    if (g_rccl.allreduce(d, d, 1, ncclUint32, ncclSum, c->comm, c->stream) != 0) return sk_fail(c, SK_E_RCCL, "ncclAllReduce failed");
    SK_HIP(c, hipMemcpyAsync(sum, d, 4, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

// Element-wise maximum of up to 8 host words over all ranks, in place: the one small collective the list walk is built on
// (plan hash + local failure flags before a list is scanned, what went wrong where after it: every rank issues the same
// sequence of these whatever happens to it locally).  No communicator: the values stay as they are.
extern "C" int sk_comm_max_u64(sk_ctx *c, uint64_t *vals, uint32_t n)
{
    if (!vals || n == 0 || n > 8) return SK_E_ARG;
    if (!c || !c->comm) return SK_OK;
    SK_HIP(c, hipSetDevice(c->device));
    uint64_t *d = (uint64_t *)(c->d_flags + 16);      // scratch words of the flag block (64 bytes in: 8-byte aligned)
    SK_HIP(c, hipMemcpyAsync(d, vals, 8u * n, hipMemcpyHostToDevice, c->stream));
    const int ncclUint64 = 5, ncclMax = 2;            // rccl.h: ncclDataType_t / ncclRedOp_t
    if (g_rccl.allreduce(d, d, n, ncclUint64, ncclMax, c->comm, c->stream) != 0) return sk_fail(c, SK_E_RCCL, "ncclAllReduce failed");
    SK_HIP(c, hipMemcpyAsync(vals, d, 8u * n, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

// ranks of the context's communicator (0: none) -- the list walk coordinates its fall-backs only when every rank of the run is in it
extern "C" int sk_comm_world(const sk_ctx *c) { return c && c->comm ? c->comm_world : 0; }

// Do all ranks hold the same 64-bit value (a hash of the work plan, before anyone scans)?  One max all-reduce over
// {v, ~v}: the values agree iff max(v) == min(v) == ~max(~v).  No communicator: a world of one agrees with itself.
extern "C" int sk_comm_agree_u64(sk_ctx *c, uint64_t value, int *agree)
{
    if (!agree) return SK_E_ARG;
    *agree = 1;
    uint64_t h[2] = {value, ~value};
    const int rc = sk_comm_max_u64(c, h, 2);
    if (rc != SK_OK) return rc;
    *agree = h[0] == value && h[1] == ~value;         // (max v == v and min v == v on this rank <=> on every rank)
    return SK_OK;
}

// In-library sum all-reduce of the whole counter block; rccl_comm == NULL uses sk_comm_init's.
extern "C" int sk_counts_allreduce(sk_ctx *c, void *rccl_comm)
{
    if (!c) return SK_E_ARG;
    if (!c->d_counts) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (!rccl_comm) rccl_comm = c->comm;
    if (!rccl_comm) return sk_fail(c, SK_E_STATE, "no communicator");
    int rc = sk_rccl_load(c);
    if (rc) return rc;
    SK_HIP(c, hipSetDevice(c->device));
    if ((rc = sk_diff_flush(c)) != SK_OK) return rc;
    const int ncclUint32 = 3, ncclSum = 0;            // rccl.h: ncclDataType_t / ncclRedOp_t
    if (g_rccl.allreduce(c->d_counts, c->d_counts, (size_t)c->nrows * c->ncols, ncclUint32, ncclSum, rccl_comm, c->stream) != 0)
        return sk_fail(c, SK_E_RCCL, "ncclAllReduce failed");
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_dev_alloc(sk_ctx *c, void **dev, uint64_t nbytes)
{
    if (!c || !dev) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    if (c->dev_uncached) SK_HIP(c, hipExtMallocWithFlags(dev, nbytes ? nbytes : 16, hipDeviceMallocUncached));
    else SK_HIP(c, hipMalloc(dev, nbytes ? nbytes : 16));
    return SK_OK;
}

extern "C" int sk_dev_free(sk_ctx *c, void *dev)
{
    if (!c) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    SK_HIP(c, hipFree(dev));
    return SK_OK;
}

extern "C" int sk_dev_upload(sk_ctx *c, void *dev, const void *host, uint64_t nbytes)
{
    if (!c || !dev || !host) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipMemcpyAsync(dev, host, nbytes, hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_dev_download(sk_ctx *c, void *host, const void *dev, uint64_t nbytes)
{
    if (!c || !dev || !host) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipMemcpyAsync(host, dev, nbytes, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

#if SK_PHASE_CLOCK
// experiment builds only: the eight phase sums (wave-cycles of s_memtime) since the last call, then zeroed
extern "C" int sk_debug_phase_clock(sk_ctx *c, unsigned long long out[8])
{
    if (!c || !out) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    unsigned long long *d = (unsigned long long *)(c->d_oddlist + (c->odd_cap ? c->odd_cap : SK_ODDCAP) - 1024u), h[512];
    SK_HIP(c, hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    SK_HIP(c, hipMemset(d, 0, sizeof h));
    for (int k = 0; k < 8; k++) { out[k] = 0; for (int s = 0; s < 64; s++) out[k] += h[s * 8 + k]; }
    return SK_OK;
}
#endif
