// sk_cover.hip -- device side of the coverage/depth aggregation (the consumer of strain_detect's hit
// list: reference scripts/coverage_depth.py, step 4 of test/example.sh) for gfx950.
//
// Per metagenome the script counts the hit lines that pass its read filter ("depth") and how many
// DIFFERENT k-mers they name ("coverage").  Here a hit is a (sample, key) pair -- key = the k-mer packed
// to 62 bits, or a row number -- and one kernel does both counts: every sample owns a power-of-two
// segment of one open-addressed u64 set; a lane inserts its key with one atomicCAS (a fresh slot = one
// more distinct k-mer for that sample) and runs of lanes with the same sample add their line count with
// one atomic.  HBM-bound integer work: 12 B read per hit + one random 8-B probe.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <new>
#include <vector>
#include "sk_internal.h"

#define COV_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint64_t cov_mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
    return x;
}

__global__ void cov_count(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ sample, uint64_t n,
                          const uint64_t *__restrict__ seg_base, const uint64_t *__restrict__ seg_mask,
                          unsigned long long *__restrict__ table, unsigned long long *__restrict__ uniq, unsigned long long *__restrict__ total)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool live = i < n;
    const uint32_t s = live ? sample[i] : 0xFFFFFFFFu;
    bool fresh = false;
    if (live) {
        const uint64_t k = keys[i], base = seg_base[s], mask = seg_mask[s];
        uint64_t slot = cov_mix(k) & mask;
        for (;;) {
            const unsigned long long old = atomicCAS(&table[base + slot], COV_EMPTY, (unsigned long long)k);
            if (old == COV_EMPTY) { fresh = true; break; }
            if (old == k) break;
            slot = (slot + 1) & mask;
        }
    }
    // runs of equal sample within the wave: one atomic per run and counter
    const uint32_t prev = (uint32_t)__shfl_up((int)s, 1);
    const bool first = (lane == 0u) | (s != prev);
    const unsigned long long fm = __ballot(first), lm = __ballot(live), um = __ballot(fresh);
    if (first & live) {
        const unsigned long long above = lane == 63u ? 0ull : fm & ~((2ull << lane) - 1ull);
        const uint32_t end = above ? (uint32_t)__builtin_ctzll(above) : 64u;
        const unsigned long long seg = (end == 64u ? ~0ull : ((1ull << end) - 1ull)) & ~((1ull << lane) - 1ull);
        atomicAdd(&total[s], (unsigned long long)__popcll(lm & seg));
        const uint32_t nu = (uint32_t)__popcll(um & seg);
        if (nu) atomicAdd(&uniq[s], (unsigned long long)nu);
    }
}

// "first sighting" counting, for hit lists whose k-mer text is not a fixed-length ACGT word: the script then
// keys its uniqueness test by the joined string <sample><k-mer>, globally, and credits the sample of the FIRST
// line that shows a string.  id[i] = number of line i's joined string (dense, < nids), in file order.
__global__ void cov_first_min(const uint32_t *__restrict__ id, uint64_t n, unsigned long long *__restrict__ first)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicMin(&first[id[i]], (unsigned long long)i);
}
__global__ void cov_first_count(const uint32_t *__restrict__ id, const uint32_t *__restrict__ sample, uint64_t n,
                                const unsigned long long *__restrict__ first, unsigned long long *__restrict__ uniq,
                                unsigned long long *__restrict__ total)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicAdd(&total[sample[i]], 1ull);
    if (first[id[i]] == (unsigned long long)i) atomicAdd(&uniq[sample[i]], 1ull);
}

extern "C" int sk_first_seen_count(sk_ctx *ctx, const uint32_t *id, const uint32_t *sample, uint64_t n, uint32_t nids, uint32_t nsamples,
                                   uint64_t *out_unique, uint64_t *out_total)
{
    if (!ctx || !out_unique || !out_total || (n && (!id || !sample))) return SK_E_ARG;
    if (nsamples == 0) return n ? SK_E_ARG : SK_OK;
    memset(out_unique, 0, (size_t)nsamples * 8);
    memset(out_total, 0, (size_t)nsamples * 8);
    if (!n) return SK_OK;
    for (uint64_t i = 0; i < n; i++)
        if (id[i] >= nids || sample[i] >= nsamples) return sk_fail_(ctx, SK_E_ARG, "id or sample out of range");
    int rc = SK_OK;
    void *d_id = NULL, *d_sample = NULL, *d_first = NULL, *d_out = NULL;
#define COV_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = sk_fail_(ctx, SK_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); goto out; } } while (0)
    COV_HIP(hipSetDevice(sk_ctx_device_(ctx)));
    COV_HIP(hipMalloc(&d_id, n * 4));
    COV_HIP(hipMalloc(&d_sample, n * 4));
    COV_HIP(hipMalloc(&d_first, (size_t)nids * 8));
    COV_HIP(hipMalloc(&d_out, (size_t)nsamples * 16));
    COV_HIP(hipMemcpy(d_id, id, n * 4, hipMemcpyHostToDevice));
    COV_HIP(hipMemcpy(d_sample, sample, n * 4, hipMemcpyHostToDevice));
    COV_HIP(hipMemset(d_first, 0xFF, (size_t)nids * 8));
    COV_HIP(hipMemset(d_out, 0, (size_t)nsamples * 16));
    hipLaunchKernelGGL(cov_first_min, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint32_t *)d_id, n, (unsigned long long *)d_first);
    hipLaunchKernelGGL(cov_first_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint32_t *)d_id, (const uint32_t *)d_sample, n,
                       (const unsigned long long *)d_first, (unsigned long long *)d_out, (unsigned long long *)d_out + nsamples);
    COV_HIP(hipGetLastError());
    COV_HIP(hipMemcpy(out_unique, d_out, (size_t)nsamples * 8, hipMemcpyDeviceToHost));
    COV_HIP(hipMemcpy(out_total, (uint64_t *)d_out + nsamples, (size_t)nsamples * 8, hipMemcpyDeviceToHost));
#undef COV_HIP
out:
    (void)hipFree(d_id); (void)hipFree(d_sample); (void)hipFree(d_first); (void)hipFree(d_out);
    return rc;
}

// Distinct and total number of keys per sample.  Keys must not be 0xFFFFFFFFFFFFFFFF.  Synchronous.
extern "C" int sk_distinct_count(sk_ctx *ctx, const uint64_t *keys, const uint32_t *sample, uint64_t n, uint32_t nsamples,
                                 uint64_t *out_unique, uint64_t *out_total)
{
    if (!ctx || !out_unique || !out_total || (n && (!keys || !sample))) return SK_E_ARG;
    if (nsamples == 0) return n ? SK_E_ARG : SK_OK;
    memset(out_unique, 0, (size_t)nsamples * 8);
    memset(out_total, 0, (size_t)nsamples * 8);
    if (!n) return SK_OK;
    // segment sizes: next power of two >= 2 * (keys of that sample)
    std::vector<uint64_t> cnt(nsamples, 0), base(nsamples), mask(nsamples);
    for (uint64_t i = 0; i < n; i++) {
        if (sample[i] >= nsamples) return sk_fail_(ctx, SK_E_ARG, "sample %u out of range", sample[i]);
        if (keys[i] == COV_EMPTY) return sk_fail_(ctx, SK_E_ARG, "reserved key value");
        cnt[sample[i]]++;
    }
    uint64_t slots = 0;
    for (uint32_t s = 0; s < nsamples; s++) {
        uint64_t c = 2;
        while (c < 2 * cnt[s]) c <<= 1;
        base[s] = slots; mask[s] = c - 1; slots += c;
    }
    int rc = SK_OK;
    void *d_keys = NULL, *d_sample = NULL, *d_seg = NULL, *d_table = NULL, *d_out = NULL;
#define COV_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = sk_fail_(ctx, SK_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); goto out; } } while (0)
    COV_HIP(hipSetDevice(sk_ctx_device_(ctx)));
    COV_HIP(hipMalloc(&d_keys, n * 8));
    COV_HIP(hipMalloc(&d_sample, n * 4));
    COV_HIP(hipMalloc(&d_seg, (size_t)nsamples * 16));
    COV_HIP(hipMalloc(&d_table, slots * 8));
    COV_HIP(hipMalloc(&d_out, (size_t)nsamples * 16));
    COV_HIP(hipMemcpy(d_keys, keys, n * 8, hipMemcpyHostToDevice));
    COV_HIP(hipMemcpy(d_sample, sample, n * 4, hipMemcpyHostToDevice));
    COV_HIP(hipMemcpy(d_seg, base.data(), (size_t)nsamples * 8, hipMemcpyHostToDevice));
    COV_HIP(hipMemcpy((uint64_t *)d_seg + nsamples, mask.data(), (size_t)nsamples * 8, hipMemcpyHostToDevice));
    COV_HIP(hipMemset(d_table, 0xFF, slots * 8));
    COV_HIP(hipMemset(d_out, 0, (size_t)nsamples * 16));
    hipLaunchKernelGGL(cov_count, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint64_t *)d_keys, (const uint32_t *)d_sample, n,
                       (const uint64_t *)d_seg, (const uint64_t *)d_seg + nsamples, (unsigned long long *)d_table,
                       (unsigned long long *)d_out, (unsigned long long *)d_out + nsamples);
    COV_HIP(hipGetLastError());
    COV_HIP(hipMemcpy(out_unique, d_out, (size_t)nsamples * 8, hipMemcpyDeviceToHost));
    COV_HIP(hipMemcpy(out_total, (uint64_t *)d_out + nsamples, (size_t)nsamples * 8, hipMemcpyDeviceToHost));
#undef COV_HIP
out:
    (void)hipFree(d_keys); (void)hipFree(d_sample); (void)hipFree(d_seg); (void)hipFree(d_table); (void)hipFree(d_out);
    return rc;
}
