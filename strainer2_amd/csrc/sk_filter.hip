// sk_filter.hip -- device side of the scrub filter (the consumer of the count table: reference
// scripts/kmer_scrub_filter.py, step 2 of test/example.sh) for gfx950.
//
// The script ranks every k-mer of the strain by max(pangenome share, metagenome share) and removes the
// most frequent ones until only min_fraction of the strain is left ("joint"), or removes everything
// above a per-column count threshold ("independent").  Here the count columns stay on the device
// (uploaded from a parsed table, or taken straight from the counters a scan just filled) and the
// ranking is a selection, not a sort:
//
//   flt_score     one 64-bit key per row: the IEEE-754 bits of the score (doubles >= 0 order like
//                 integers) + 1, or 0 for rows that no longer take part
//   flt_digit/flt_pick   MSB-first radix select, 8 passes of 8 bits: the key T of the n_scrub-th largest
//                 row and how many rows equal to T are still to be taken
//   flt_eq_count/flt_eq_scan/flt_mark   rows with key > T go; rows with key == T go in row order until
//                 the quota is used (what a stable descending sort does with ties)
//   flt_hist      value histogram of one column (LDS-privatised) for the independent mode's threshold walk
//
// All of it is HBM-streaming integer/double work over N rows (N ~ 5-7 M: a few tens of MB per pass);
// at that size every kernel is launch-latency bound, the point of running it here is that the fused
// path (scan -> filter in one process) never moves the counters off the device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include "sk_internal.h"

#define FLT_THREADS 256
#define FLT_ITEMS   16                       // consecutive rows per thread in the tie-rank kernels
#define FLT_CHUNK   (FLT_THREADS * FLT_ITEMS)
#define FLT_LDS_BINS 2048

struct flt_state {                           // lives on the device between the select passes
    unsigned long long prefix, mask, need;
    unsigned long long hist[8][256];
};

struct sk_filter {
    sk_ctx      *ctx;
    int          device;
    hipStream_t  stream;
    uint64_t     n, cap;
    int64_t     *d_pan, *d_meta;
    uint8_t     *d_gone, *d_out;
    uint64_t    *d_keys;
    uint32_t    *d_u32;                      // 3 * cap staging for sk_filter_load_counts
    unsigned long long *d_blk;               // per-chunk tie counts / offsets
    unsigned long long *d_hist;              // value histogram
    size_t       hist_cap;
    flt_state   *d_state;
    unsigned long long *d_sums;              // pan sum, meta sum, #pan > 0, #meta > 0, #gone
};

#define FLT_HIP(f, call)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return sk_fail_((f)->ctx, SK_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                                   \
    } while (0)

// counters as the table prints them: %d of an unsigned (src/kmer_scrub_count.c:146-151)
__global__ void flt_from_u32(const uint32_t *__restrict__ pan, const uint32_t *__restrict__ meta, const uint32_t *__restrict__ drug,
                             uint64_t n, int64_t *__restrict__ o_pan, int64_t *__restrict__ o_meta, uint8_t *__restrict__ o_gone)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_pan[i] = (int64_t)(int32_t)pan[i];
    o_meta[i] = (int64_t)(int32_t)meta[i];
    o_gone[i] = drug ? ((int32_t)drug[i] > 0 ? 1 : 0) : 0;
}

__device__ __forceinline__ unsigned long long flt_wave_sum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = (unsigned)__shfl_down((int)(unsigned)v, off), hi = (unsigned)__shfl_down((int)(unsigned)(v >> 32), off);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// sums[0..4] += sum of positive pan, sum of positive meta, #pan > 0, #meta > 0, #gone
__global__ void flt_sums(const int64_t *__restrict__ pan, const int64_t *__restrict__ meta, const uint8_t *__restrict__ gone,
                         uint64_t n, unsigned long long *__restrict__ sums)
{
    unsigned long long a[5] = {0, 0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t p = pan[i], m = meta[i];
        if (p > 0) { a[0] += (unsigned long long)p; a[2]++; }
        if (m > 0) { a[1] += (unsigned long long)m; a[3]++; }
        a[4] += gone[i] ? 1u : 0u;
    }
    for (int k = 0; k < 5; k++) {
        const unsigned long long s = flt_wave_sum(a[k]);
        if ((threadIdx.x & 63u) == 0 && s) atomicAdd(&sums[k], s);
    }
}

// hist[b] += #{v > 0 : v == lo + b} for b < nbins; hist[nbins] += #{v > 0 : v >= lo + nbins}; values below lo are ignored
__global__ void flt_hist(const int64_t *__restrict__ vals, uint64_t n, int64_t lo, uint32_t nbins, unsigned long long *__restrict__ hist)
{
    __shared__ uint32_t lds[FLT_LDS_BINS];
    for (uint32_t b = threadIdx.x; b < FLT_LDS_BINS; b += blockDim.x) lds[b] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const int64_t v = vals[i];
        if (v <= 0 || v < lo) continue;
        const uint64_t d = (uint64_t)(v - lo);
        const uint32_t b = d < nbins ? (uint32_t)d : nbins;
        if (b < FLT_LDS_BINS) atomicAdd(&lds[b], 1u);
        else atomicAdd(&hist[b], 1ull);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < FLT_LDS_BINS && b <= nbins; b += blockDim.x)
        if (lds[b]) atomicAdd(&hist[b], (unsigned long long)lds[b]);
}

// score of a row (scripts/kmer_scrub_filter.py:91-115): the larger of its two shares, shares being
// count / column sum in double arithmetic; only positive counts have a share
__global__ void flt_score(const int64_t *__restrict__ pan, const int64_t *__restrict__ meta, const uint8_t *__restrict__ gone,
                          uint64_t n, double pan_sum, double meta_sum, uint64_t *__restrict__ keys)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (gone[i]) { keys[i] = 0; return; }
    double s = 0.0;
    const int64_t m = meta[i], p = pan[i];
    if (m > 0) { const double v = (double)m / meta_sum; if (v > s) s = v; }
    if (p > 0) { const double v = (double)p / pan_sum; if (v > s) s = v; }
    keys[i] = (uint64_t)__double_as_longlong(s) + 1ull;
}

__global__ void flt_digit(const uint64_t *__restrict__ keys, uint64_t n, flt_state *__restrict__ st, int pass)
{
    __shared__ uint32_t lds[256];
    lds[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix, mask = st->mask;
    const int shift = 56 - 8 * pass;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        if (k != 0 && (k & mask) == prefix) atomicAdd(&lds[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (lds[threadIdx.x]) atomicAdd(&st->hist[pass][threadIdx.x], (unsigned long long)lds[threadIdx.x]);
}

// one thread: the digit in which the need-th largest key (among those matching the prefix) falls
__global__ void flt_pick(flt_state *st, int pass)
{
    if (threadIdx.x || blockIdx.x) return;
    const int shift = 56 - 8 * pass;
    unsigned long long need = st->need, above = 0;
    int d = 255;
    for (; d > 0; d--) {
        const unsigned long long h = st->hist[pass][d];
        if (above + h >= need) break;
        above += h;
    }
    st->need = need - above;
    st->prefix |= (unsigned long long)d << shift;
    st->mask |= 255ull << shift;
}

// per chunk of FLT_CHUNK rows: how many keys equal the threshold
__global__ void flt_eq_count(const uint64_t *__restrict__ keys, uint64_t n, const flt_state *__restrict__ st, unsigned long long *__restrict__ blk)
{
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const unsigned long long T = st->prefix;
    const uint64_t base = (uint64_t)blockIdx.x * FLT_CHUNK + (uint64_t)threadIdx.x * FLT_ITEMS;
    uint32_t c = 0;
    for (int j = 0; j < FLT_ITEMS; j++)
        if (base + j < n && keys[base + j] == T) c++;
    if (c) atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = total;
}

// one block: exclusive prefix sum over the chunk counts, in place
__global__ void flt_eq_scan(unsigned long long *blk, uint32_t nblk)
{
    __shared__ unsigned long long part[FLT_THREADS];
    const uint32_t per = (nblk + FLT_THREADS - 1) / FLT_THREADS;
    const uint32_t a = threadIdx.x * per, b = a + per < nblk ? a + per : nblk;
    unsigned long long s = 0;
    for (uint32_t i = a; i < b; i++) s += blk[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < FLT_THREADS; i++) { const unsigned long long v = part[i]; part[i] = run; run += v; }
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint32_t i = a; i < b; i++) { const unsigned long long v = blk[i]; blk[i] = run; run += v; }
}

__global__ void flt_mark(const uint64_t *__restrict__ keys, uint64_t n, const flt_state *__restrict__ st, const unsigned long long *__restrict__ blk,
                         uint8_t *__restrict__ out)
{
    __shared__ uint32_t cnt[FLT_THREADS];
    const unsigned long long T = st->prefix, need = st->need;
    const uint64_t base = (uint64_t)blockIdx.x * FLT_CHUNK + (uint64_t)threadIdx.x * FLT_ITEMS;
    uint32_t c = 0;
    for (int j = 0; j < FLT_ITEMS; j++)
        if (base + j < n && keys[base + j] == T) c++;
    cnt[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < FLT_THREADS; i++) { const uint32_t v = cnt[i]; cnt[i] = run; run += v; }
    }
    __syncthreads();
    unsigned long long rank = blk[blockIdx.x] + cnt[threadIdx.x];
    for (int j = 0; j < FLT_ITEMS; j++) {
        if (base + j >= n) break;
        const uint64_t k = keys[base + j];
        uint8_t o = 0;
        if (k == 0 || k > T) o = 1;                  // already gone, or above the cut
        else if (k == T) { o = rank < need ? 1 : 0; rank++; }
        out[base + j] = o;
    }
}

__global__ void flt_above(const int64_t *__restrict__ pan, const int64_t *__restrict__ meta, const uint8_t *__restrict__ gone, uint64_t n,
                          int64_t pan_thr, int64_t meta_thr, uint8_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t p = pan[i], m = meta[i];
    out[i] = (gone[i] || (p > 0 && p > pan_thr) || (m > 0 && m > meta_thr)) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
static void flt_release(sk_filter *f)
{
    hipFree(f->d_pan); hipFree(f->d_meta); hipFree(f->d_gone); hipFree(f->d_out); hipFree(f->d_keys);
    hipFree(f->d_u32); hipFree(f->d_blk);
    f->d_pan = f->d_meta = NULL; f->d_gone = f->d_out = NULL; f->d_keys = NULL; f->d_u32 = NULL; f->d_blk = NULL;
    f->cap = 0;
}

static int flt_reserve(sk_filter *f, uint64_t n)
{
    FLT_HIP(f, hipSetDevice(f->device));
    if (n <= f->cap) return SK_OK;
    flt_release(f);
    const uint64_t cap = n + n / 8 + 1024;
    FLT_HIP(f, hipMalloc((void **)&f->d_pan, cap * 8));
    FLT_HIP(f, hipMalloc((void **)&f->d_meta, cap * 8));
    FLT_HIP(f, hipMalloc((void **)&f->d_keys, cap * 8));
    FLT_HIP(f, hipMalloc((void **)&f->d_gone, cap));
    FLT_HIP(f, hipMalloc((void **)&f->d_out, cap));
    FLT_HIP(f, hipMalloc((void **)&f->d_blk, (cap / FLT_CHUNK + 2) * 8));
    f->cap = cap;
    return SK_OK;
}

extern "C" int sk_filter_create(sk_ctx *ctx, sk_filter **out)
{
    if (!ctx || !out) return SK_E_ARG;
    *out = NULL;
    sk_filter *f = new (std::nothrow) sk_filter();
    if (!f) return SK_E_NOMEM;
    memset(f, 0, sizeof *f);
    f->ctx = ctx;
    f->device = sk_ctx_device_(ctx);
    if (hipSetDevice(f->device) != hipSuccess || hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&f->d_state, sizeof(flt_state)) != hipSuccess || hipMalloc((void **)&f->d_sums, 5 * 8) != hipSuccess) {
        delete f;
        return sk_fail_(ctx, SK_E_HIP, "could not set up the filter on device %d", sk_ctx_device_(ctx));
    }
    *out = f;
    return SK_OK;
}

extern "C" void sk_filter_destroy(sk_filter *f)
{
    if (!f) return;
    hipSetDevice(f->device);
    hipStreamSynchronize(f->stream);
    flt_release(f);
    hipFree(f->d_hist); hipFree(f->d_state); hipFree(f->d_sums);
    hipStreamDestroy(f->stream);
    delete f;
}

extern "C" int sk_filter_load(sk_filter *f, const int64_t *pan, const int64_t *meta, const uint8_t *gone, uint64_t n)
{
    if (!f || (n && (!pan || !meta))) return SK_E_ARG;
    int rc = flt_reserve(f, n);
    if (rc) return rc;
    f->n = n;
    if (!n) return SK_OK;
    FLT_HIP(f, hipMemcpyAsync(f->d_pan, pan, n * 8, hipMemcpyHostToDevice, f->stream));
    FLT_HIP(f, hipMemcpyAsync(f->d_meta, meta, n * 8, hipMemcpyHostToDevice, f->stream));
    if (gone) FLT_HIP(f, hipMemcpyAsync(f->d_gone, gone, n, hipMemcpyHostToDevice, f->stream));
    else FLT_HIP(f, hipMemsetAsync(f->d_gone, 0, n, f->stream));
    FLT_HIP(f, hipStreamSynchronize(f->stream));
    return SK_OK;
}

extern "C" int sk_filter_load_counts(sk_filter *f, uint32_t pan_col, uint32_t meta_col, int32_t drug_col)
{
    if (!f) return SK_E_ARG;
    const uint64_t n = sk_table_rows(f->ctx);
    const uint32_t ncols = sk_table_cols(f->ctx);
    if (pan_col >= ncols || meta_col >= ncols || (drug_col >= 0 && (uint32_t)drug_col >= ncols))
        return sk_fail_(f->ctx, SK_E_ARG, "filter: column out of range");
    int rc = flt_reserve(f, n);
    if (rc) return rc;
    f->n = n;
    if (!n) return SK_OK;
    if (!f->d_u32) FLT_HIP(f, hipMalloc((void **)&f->d_u32, f->cap * 12));
    uint32_t *p = f->d_u32, *m = p + f->cap, *d = m + f->cap;
    if ((rc = sk_counts_rows_to_device_(f->ctx, pan_col, p)) != SK_OK) return rc;
    if ((rc = sk_counts_rows_to_device_(f->ctx, meta_col, m)) != SK_OK) return rc;
    if (drug_col >= 0 && (rc = sk_counts_rows_to_device_(f->ctx, (uint32_t)drug_col, d)) != SK_OK) return rc;
    hipLaunchKernelGGL(flt_from_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, f->stream, p, m, drug_col >= 0 ? d : (uint32_t *)NULL, n,
                       f->d_pan, f->d_meta, f->d_gone);
    FLT_HIP(f, hipStreamSynchronize(f->stream));
    return SK_OK;
}

extern "C" int sk_filter_sums(sk_filter *f, int64_t *pan_sum, int64_t *meta_sum, uint64_t *n_pan, uint64_t *n_meta, uint64_t *n_gone)
{
    if (!f) return SK_E_ARG;
    FLT_HIP(f, hipSetDevice(f->device));
    unsigned long long h[5] = {0, 0, 0, 0, 0};
    FLT_HIP(f, hipMemsetAsync(f->d_sums, 0, sizeof h, f->stream));
    if (f->n) hipLaunchKernelGGL(flt_sums, dim3(1024), dim3(256), 0, f->stream, f->d_pan, f->d_meta, f->d_gone, f->n, f->d_sums);
    FLT_HIP(f, hipMemcpyAsync(h, f->d_sums, sizeof h, hipMemcpyDeviceToHost, f->stream));
    FLT_HIP(f, hipStreamSynchronize(f->stream));
    if (pan_sum) *pan_sum = (int64_t)h[0];
    if (meta_sum) *meta_sum = (int64_t)h[1];
    if (n_pan) *n_pan = h[2];
    if (n_meta) *n_meta = h[3];
    if (n_gone) *n_gone = h[4];
    return SK_OK;
}

extern "C" int sk_filter_hist(sk_filter *f, int which, int64_t lo, uint32_t nbins, uint64_t *hist)
{
    if (!f || !hist || nbins == 0 || nbins > (1u << 24) || (which != 0 && which != 1)) return SK_E_ARG;
    FLT_HIP(f, hipSetDevice(f->device));
    const size_t need = ((size_t)nbins + 1) * 8;
    if (need > f->hist_cap) {
        if (f->d_hist) { hipFree(f->d_hist); f->d_hist = NULL; f->hist_cap = 0; }
        FLT_HIP(f, hipMalloc((void **)&f->d_hist, need));
        f->hist_cap = need;
    }
    FLT_HIP(f, hipMemsetAsync(f->d_hist, 0, need, f->stream));
    if (f->n) hipLaunchKernelGGL(flt_hist, dim3(512), dim3(256), 0, f->stream, which ? f->d_meta : f->d_pan, f->n, lo, nbins, f->d_hist);
    FLT_HIP(f, hipMemcpyAsync(hist, f->d_hist, need, hipMemcpyDeviceToHost, f->stream));
    FLT_HIP(f, hipStreamSynchronize(f->stream));
    return SK_OK;
}

extern "C" int sk_filter_joint(sk_filter *f, int64_t pan_sum, int64_t meta_sum, uint64_t n_scrub, uint8_t *out_scrub)
{
    if (!f || (f->n && !out_scrub)) return SK_E_ARG;
    FLT_HIP(f, hipSetDevice(f->device));
    const uint64_t n = f->n;
    if (!n) return SK_OK;
    if (n_scrub == 0) {                           // nothing to rank: the result is the rows that were already gone
        FLT_HIP(f, hipMemcpyAsync(out_scrub, f->d_gone, n, hipMemcpyDeviceToHost, f->stream));
        FLT_HIP(f, hipStreamSynchronize(f->stream));
        return SK_OK;
    }
    const unsigned nb = (unsigned)((n + 255) / 256), nchunk = (unsigned)((n + FLT_CHUNK - 1) / FLT_CHUNK);
    flt_state init;
    memset(&init, 0, sizeof init);
    init.need = n_scrub;
    FLT_HIP(f, hipMemcpyAsync(f->d_state, &init, sizeof init, hipMemcpyHostToDevice, f->stream));
    hipLaunchKernelGGL(flt_score, dim3(nb), dim3(256), 0, f->stream, f->d_pan, f->d_meta, f->d_gone, n, (double)pan_sum, (double)meta_sum, f->d_keys);
    for (int pass = 0; pass < 8; pass++) {
        hipLaunchKernelGGL(flt_digit, dim3(1024), dim3(256), 0, f->stream, f->d_keys, n, f->d_state, pass);
        hipLaunchKernelGGL(flt_pick, dim3(1), dim3(64), 0, f->stream, f->d_state, pass);
    }
    hipLaunchKernelGGL(flt_eq_count, dim3(nchunk), dim3(FLT_THREADS), 0, f->stream, f->d_keys, n, f->d_state, f->d_blk);
    hipLaunchKernelGGL(flt_eq_scan, dim3(1), dim3(FLT_THREADS), 0, f->stream, f->d_blk, nchunk);
    hipLaunchKernelGGL(flt_mark, dim3(nchunk), dim3(FLT_THREADS), 0, f->stream, f->d_keys, n, f->d_state, f->d_blk, f->d_out);
    flt_state fin;
    FLT_HIP(f, hipMemcpyAsync(out_scrub, f->d_out, n, hipMemcpyDeviceToHost, f->stream));
    FLT_HIP(f, hipMemcpyAsync(&fin, f->d_state, 3 * 8, hipMemcpyDeviceToHost, f->stream));
    FLT_HIP(f, hipStreamSynchronize(f->stream));
    if (fin.prefix == 0)                          // would mean n_scrub exceeded the rows that take part
        return sk_fail_(f->ctx, SK_E_ARG, "filter: asked to scrub %llu rows, fewer take part", (unsigned long long)n_scrub);
    return SK_OK;
}

extern "C" int sk_filter_above(sk_filter *f, int64_t pan_thr, int64_t meta_thr, uint8_t *out_scrub)
{
    if (!f || (f->n && !out_scrub)) return SK_E_ARG;
    FLT_HIP(f, hipSetDevice(f->device));
    const uint64_t n = f->n;
    if (!n) return SK_OK;
    hipLaunchKernelGGL(flt_above, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, f->stream, f->d_pan, f->d_meta, f->d_gone, n, pan_thr, meta_thr, f->d_out);
    FLT_HIP(f, hipMemcpyAsync(out_scrub, f->d_out, n, hipMemcpyDeviceToHost, f->stream));
    FLT_HIP(f, hipStreamSynchronize(f->stream));
    return SK_OK;
}
