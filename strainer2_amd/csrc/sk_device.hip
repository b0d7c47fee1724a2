// sk_device.hip -- device layer of libstrainer_kmer.so: hand-written HIP for CDNA4 / gfx950.
//
// Kernels (all integer work; HBM/cache bound, no MFMA):
//   sk_table_insert   build the open-addressed key table in HBM (atomicCAS on 64-bit slots)
//   sk_scan_main      THE hot kernel: slide the k=31 window over a record stream,
//                     rolling forward + reverse-complement 2-bit packing in registers,
//                     canonical = max, probe the table, atomicAdd the row counter.
//                     Replaces reference src/genome_compare.c:213-229 + src/BIO_hash.c:161-172.
//   sk_scan_wide      exact byte-string path for the rare windows that contain bytes other
//                     than ACGT (U / IUPAC / junk): reproduces the reference's signed-char
//                     orientation compare through its COMPLEMENT map.  Runs only when
//                     sk_scan_main saw such a window in the batch (device-side early exit).
//
// Data layout in HBM:
//   keys   [S]  u64   open addressing, linear probing, S = 2^s slots, empty = all ones
//   rowid  [S]  u32   row index of the key in that slot (read on hits only)
//   counts [ncols][nrows] u32
//   stream      u8    record stream: sequence bytes, records separated by '\n'
//
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

#include "../../include/strainer_kmer.h"
#include "sk_common.h"

// ---------------------------------------------------------------------------------------------
// scan kernel geometry
// ---------------------------------------------------------------------------------------------
#define SK_THREADS      256                 // 4 waves of 64
#define SK_SPAN         128                 // window-end positions per thread
#define SK_TILE         (SK_THREADS * SK_SPAN)
#define SK_CH           16                  // bytes per LDS chunk (one ds_read_b128)
#define SK_SPAN_CH      (SK_SPAN / SK_CH)   // 8 chunks per span
#define SK_LEAD         32                  // bytes staged in front of the tile (>= k-1, 16-aligned)
#define SK_TILE_CH      ((SK_TILE + SK_LEAD) / SK_CH)
// one 16-byte pad after every SK_SPAN_CH chunks: lane stride 36 dwords -> conflict-free b128 reads
#define SK_LDS_CH       (SK_TILE_CH + SK_TILE_CH / SK_SPAN_CH + 1)

typedef uint32_t sk_u4 __attribute__((ext_vector_type(4)));

// per-wave queue of windows that passed stage 1 and still need their table probe
#define SK_WAVES        (SK_THREADS / 64)
#define SK_QCAP         (64 + 4 * 64)       // drained below 64 before every group of 4 windows

__device__ __forceinline__ uint32_t sk_lds_slot(uint32_t chunk) { return chunk + chunk / SK_SPAN_CH; }

struct sk_table_view {
    const uint64_t *keys;
    const uint32_t *rowid;
    uint32_t        mask;
    // L2-resident prefilter: a Bloom set of the MINIMIZER hashes that occur in the strain
    // (about nrows/8 items).  64-bit blocks chosen by the low bits of the minimizer hash,
    // two bits in each 32-bit half.
    const uint2    *bloom;
    uint32_t        bloom_mask;      // number of 64-bit blocks - 1
};

// follow the probe sequence of `canon` from `slot` (first key already loaded)
__device__ __forceinline__ void sk_resolve(uint64_t canon, uint32_t slot, uint64_t key,
                                           const sk_table_view &t, uint32_t *counts)
{
    for (;;) {
        if (key == canon) { atomicAdd(&counts[t.rowid[slot]], 1u); return; }
        if (key == SK_EMPTY64) return;
        slot = (slot + 1u) & t.mask;
        key = t.keys[slot];
    }
}

// is minimizer hash `mz` (possibly) one of the strain's?  blk = its filter block
// stage 2 for one queued window: slot from the k-mer hash, 62-bit compare, atomicAdd on a hit
__device__ __forceinline__ void sk_probe(uint64_t canon, const sk_table_view &t, uint32_t *counts)
{
    const uint32_t slot = sk_slot0(0u, sk_khash(canon), t.mask);
    const uint64_t key = t.keys[slot];
    if (key != SK_EMPTY64) sk_resolve(canon, slot, key, t, counts);
}

__device__ __forceinline__ bool sk_filter_test(const uint2 blk, uint32_t mz)
{
    const uint32_t g = mz * 0x9E3779B1u;
    const uint32_t t = (blk.x >> (g >> 27)) & (blk.x >> ((g >> 22) & 31u)) &
                       (blk.y >> ((g >> 17) & 31u)) & (blk.y >> ((g >> 12) & 31u));
    return (t & 1u) != 0u;
}

struct sk_roll {
    uint64_t fwd, rc;       // forward / reverse-complement packed windows (62 bits)
    uint32_t run;           // consecutive ACGT bytes ending here
    uint32_t soft;          // consecutive bytes that are not hard breakers (N, '\n', NUL)
};

__device__ __forceinline__ void sk_step(sk_roll &s, uint32_t b)
{
    const uint32_t code = sk_code(b);
    s.fwd = ((s.fwd << 2) | code) & SK_KMASK62;
    s.rc  = (s.rc >> 2) | ((uint64_t)(3u - code) << 60);
    s.run  = sk_is_acgt(b) ? s.run + 1u : 0u;
    s.soft = sk_is_hard_break(b) ? 0u : s.soft + 1u;
}

// hash of the canonical 16-mer that ends at the current base
__device__ __forceinline__ uint32_t sk_mmer_hash(const sk_roll &s)
{
    const uint32_t f16 = (uint32_t)s.fwd;
    const uint32_t r16 = (uint32_t)(s.rc >> 30);
    return sk_mhash(f16 < r16 ? f16 : r16);
}

// THE hot kernel.  One thread owns SK_SPAN consecutive window-end positions and rolls over
// them in 16-base chunks held in registers (bytes staged through LDS, coalesced 16 B loads):
//   roll      fwd/rc 2-bit packing, ACGT run length                      (registers)
//   minimizer sliding minimum of the 16-mer hashes over the window        (registers;
//             block-decomposed: prefix minima of this chunk + suffix minima of the last)
//   stage 1   "does the strain contain this minimizer at all?"  One 8-byte load from the
//             L2-resident minimizer filter, only when the lane's minimizer changes (about once
//             per 8 windows); the verdict is kept in a register until it changes again.
//             Windows of reads unrelated to the strain stop here (false positives ~0.1 %).
//   stage 2   table probe (HBM / Infinity Cache), only for windows whose minimizer passed:
//             slot from the k-mer hash, full 62-bit compare, atomicAdd on a hit
template <bool BLOOM, bool STATS, int ABLATE>      // ABLATE (timing experiments only; wrong counts):
__global__ __launch_bounds__(SK_THREADS)           // 1 = no filter/table memory at all, 2 = no table probes
void sk_scan_main(const uint8_t *__restrict__ stream, uint64_t nbytes, uint64_t emit_begin,
                  sk_table_view table, uint32_t *__restrict__ counts, uint32_t *__restrict__ flags)
{
    __shared__ sk_u4 lds[SK_LDS_CH];
    __shared__ uint64_t queue[SK_WAVES][SK_QCAP];

    const uint64_t tile0 = (uint64_t)blockIdx.x * SK_TILE;        // first window-end position
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint64_t *const wq = queue[tid >> 6];
    uint32_t qn = 0;                                              // queued windows (wave-uniform)

    // ---- stage bytes [tile0 - LEAD, tile0 + TILE) into LDS, 16 B per lane, coalesced --------
    for (uint32_t c = tid; c < SK_TILE_CH; c += SK_THREADS) {
        const int64_t off = (int64_t)tile0 - SK_LEAD + (int64_t)c * SK_CH;
        sk_u4 v;
        if (off >= 0 && (uint64_t)off + SK_CH <= nbytes) {
            v = __builtin_nontemporal_load((const sk_u4 *)(stream + off));
        } else {
            uint32_t w[4] = {0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au};   // '\n' fill
            for (int i = 0; i < SK_CH; i++) {
                const int64_t p = off + i;
                if (p >= 0 && (uint64_t)p < nbytes) {
                    w[i >> 2] = (w[i >> 2] & ~(0xFFu << ((i & 3) * 8))) | ((uint32_t)stream[p] << ((i & 3) * 8));
                }
            }
            v = (sk_u4){w[0], w[1], w[2], w[3]};
        }
        lds[sk_lds_slot(c)] = v;
    }
    __syncthreads();

    sk_roll s;
    s.fwd = 0; s.rc = 0; s.run = 0; s.soft = 0;
    uint32_t wide_seen = 0;
    uint32_t n_live = 0, n_load = 0, n_probe = 0;             // STATS only
    const uint32_t chunk0 = tid * SK_SPAN_CH;
    const uint64_t pos0 = tile0 + (uint64_t)tid * SK_SPAN;       // window-end position of span byte 0

    uint32_t S[17];                     // suffix minima of the previous chunk's 16-mer hashes
#pragma unroll
    for (int i = 0; i < 17; i++) S[i] = 0xFFFFFFFFu;
    uint32_t cur_mz = 0xFFFFFFFFu;      // minimizer hash the verdict below belongs to
    bool     cur_pass = false;          // stage-1 verdict for cur_mz

    // ---- warm-up chunk 0: rolling state only ---------------------------------------------
    {
        const sk_u4 v = lds[sk_lds_slot(chunk0)];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int o = 0; o < 16; o++) sk_step(s, (w[o >> 2] >> (8 * (o & 3))) & 0xFFu);
    }

#pragma unroll 1
    for (uint32_t j = 1; j < SK_SPAN_CH + 2; j++) {
        const sk_u4 v = lds[sk_lds_slot(chunk0 + j)];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t H[16];
        const bool emit_chunk = j >= 2;                        // chunk 1 only warms the minima up
        const uint64_t pbase = pos0 + (uint64_t)(j - 2) * SK_CH;
        uint32_t P = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint64_t canon[4];
            uint32_t mz[4];
            bool     live[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int o = q * 4 + i;
                sk_step(s, (w[q] >> (8 * i)) & 0xFFu);
                const uint32_t h = sk_mmer_hash(s);
                H[o] = h;
                P = h < P ? h : P;
                mz[i] = S[o + 1] < P ? S[o + 1] : P;
                const uint64_t p = pbase + (uint64_t)o;
                const bool in_range = emit_chunk & (p >= emit_begin) & (p < nbytes);
                canon[i] = s.fwd > s.rc ? s.fwd : s.rc;
                live[i] = in_range & (s.run >= (uint32_t)SK_K);
                wide_seen |= (uint32_t)(in_range & (s.run < (uint32_t)SK_K) & (s.soft >= (uint32_t)SK_K));
            }
            if (!emit_chunk) continue;
            if (BLOOM) {
                // stage 1: look the minimizer up only where it differs from the one already judged
                uint32_t m[4];
                bool     need[4];
                uint32_t prev = cur_mz;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    m[i] = live[i] ? mz[i] : prev;
                    need[i] = m[i] != prev;
                    prev = m[i];
                }
                uint2 b[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    b[i] = make_uint2(0u, 0u);
                    if (need[i] && ABLATE != 1) b[i] = table.bloom[m[i] & table.bloom_mask];
                    if (STATS) { n_live += live[i]; n_load += need[i]; }
                }
                bool pass = cur_pass;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (need[i]) pass = sk_filter_test(b[i], m[i]) || (ABLATE == 1 && m[i] == 0x12345u);
                    live[i] = live[i] & pass;
                }
                cur_pass = pass;
                cur_mz = prev;
            }
            if (ABLATE == 2) {
#pragma unroll
                for (int i = 0; i < 4; i++) live[i] = live[i] & (canon[i] == 0x123456789ull);
            }
            // queue the windows that passed; stage 2 runs on full waves when 64 are waiting
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned long long mask = __ballot(live[i]);
                if (mask) {
                    if (live[i]) wq[qn + __popcll(mask & ((1ull << lane) - 1ull))] = canon[i];
                    qn += (uint32_t)__popcll(mask);
                    if (STATS) n_probe += live[i];
                }
            }
            if (qn >= 64u) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                do {
                    qn -= 64u;
                    sk_probe(wq[qn + lane], table, counts);
                } while (qn >= 64u);
                __builtin_amdgcn_wave_barrier();
            }
        }
        // suffix minima of this chunk for the next one
        S[15] = H[15];
#pragma unroll
        for (int i = 14; i >= 0; i--) S[i] = H[i] < S[i + 1] ? H[i] : S[i + 1];
    }
    // drain what is left in the queue
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < qn) sk_probe(wq[lane], table, counts);
    if (wide_seen) atomicAdd(&flags[0], 1u);
    if (STATS) {
        atomicAdd((unsigned long long *)&flags[4], (unsigned long long)n_live);
        atomicAdd((unsigned long long *)&flags[6], (unsigned long long)n_load);
        atomicAdd((unsigned long long *)&flags[8], (unsigned long long)n_probe);
    }
}

// ---------------------------------------------------------------------------------------------
// wide (byte-string) path
// ---------------------------------------------------------------------------------------------
__constant__ signed char sk_comp_dev[256];

struct sk_wide_view {
    const char     *keys31;     // [nwide][32]
    const uint32_t *rows;       // [nwide]
    const uint32_t *index;      // [wmask+1]  0 = empty, else key index + 1
    uint32_t        wmask;
    uint32_t        nwide;
};

__global__ __launch_bounds__(256)
void sk_scan_wide(const uint8_t *__restrict__ stream, uint64_t nbytes, uint64_t emit_begin,
                  sk_table_view table, sk_wide_view wide, uint32_t *__restrict__ counts,
                  const uint32_t *__restrict__ flags)
{
    if (flags[0] == 0u) return;                        // no window with a non-ACGT byte in this batch
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = emit_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nbytes; p += stride) {
        if (p < (uint64_t)(SK_K - 1)) continue;
        const uint8_t *w = stream + (p - (SK_K - 1));
        char u[SK_K];
        bool hard = false, pure = true;
        for (int i = 0; i < SK_K; i++) {
            const uint32_t b = w[i];
            hard |= (bool)sk_is_hard_break(b);
            pure &= (bool)sk_is_acgt(b);
            u[i] = (char)sk_upper(b);
        }
        if (hard || pure) continue;                    // skipped by the reference / done by sk_scan_main
        // orientation: sign of (window - revcomp) in signed-char order (src/genome_compare.c:1122-1141)
        int sign = 0;
        for (int i = 0; i < SK_K && sign == 0; i++) {
            const signed char f = (signed char)u[i];
            const signed char r = sk_comp_dev[(uint8_t)u[SK_K - 1 - i]];
            sign = (f > r) - (r > f);
        }
        char o[SK_K + 1];
        if (sign >= 0) { for (int i = 0; i < SK_K; i++) o[i] = u[i]; }
        else           { for (int i = 0; i < SK_K; i++) o[SK_K - 1 - i] = (char)sk_comp_dev[(uint8_t)u[i]]; }
        o[SK_K] = 0;
        bool opure = true, has_n = false, has_nul = false;
        for (int i = 0; i < SK_K; i++) {
            opure &= (bool)sk_is_acgt((uint8_t)o[i]) & ((uint8_t)o[i] < 'a');
            has_n |= (o[i] == 'N');
            has_nul |= (o[i] == 0);
        }
        if (has_n || has_nul) continue;
        if (opure) {                                   // e.g. a window with U whose revcomp wins
            uint64_t key = 0;
            for (int i = 0; i < SK_K; i++) key = (key << 2) | sk_code((uint8_t)o[i]);
            const uint32_t slot = sk_slot0(sk_minimizer62(key), sk_khash(key), table.mask);
            const uint64_t first = table.keys[slot];
            if (first != SK_EMPTY64) sk_resolve(key, slot, first, table, counts);
        } else if (wide.nwide) {
            uint32_t slot = sk_hash_wide(o) & wide.wmask;
            for (;;) {
                const uint32_t e = wide.index[slot];
                if (e == 0u) break;
                const char *cand = wide.keys31 + (size_t)(e - 1u) * 32u;
                bool same = true;
                for (int i = 0; i < SK_K; i++) same &= (cand[i] == o[i]);
                if (same) { atomicAdd(&counts[wide.rows[e - 1u]], 1u); break; }
                slot = (slot + 1u) & wide.wmask;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// table build
// ---------------------------------------------------------------------------------------------
__global__ void sk_fill64(uint64_t *p, uint64_t n, uint64_t v)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

__global__ void sk_table_insert(const uint64_t *__restrict__ in, uint32_t n, uint64_t *keys,
                                uint32_t *rowid, uint32_t mask, uint32_t *flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = in[i];
    if (k == SK_EMPTY64) return;                       // wide row: not in this table
    if (k > SK_KMASK62) { atomicAdd(&flags[1], 1u); return; }
    uint32_t slot = sk_slot0(sk_minimizer62(k), sk_khash(k), mask);
    for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long *)&keys[slot],
                                                (unsigned long long)SK_EMPTY64, (unsigned long long)k);
        if (old == SK_EMPTY64) { rowid[slot] = i; return; }
        if (old == k) { atomicAdd(&flags[1], 1u); return; }     // duplicate key
        slot = (slot + 1u) & mask;
    }
}

__global__ void sk_bloom_insert(const uint64_t *__restrict__ in, uint32_t n, uint32_t *bloom_words, uint32_t bloom_mask)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = in[i];
    if (k == SK_EMPTY64) return;
    const uint32_t mz = sk_minimizer62(k);
    const uint32_t g = mz * 0x9E3779B1u;
    uint32_t *blk = bloom_words + 2u * (size_t)(mz & bloom_mask);
    atomicOr(&blk[0], (1u << (g >> 27)) | (1u << ((g >> 22) & 31u)));
    atomicOr(&blk[1], (1u << ((g >> 17) & 31u)) | (1u << ((g >> 12) & 31u)));
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
#define SK_STAGE_BYTES   (64ull << 20)
#define SK_NSTAGE        2

struct sk_ctx {
    int          device;
    hipStream_t  stream;
    // table
    uint64_t    *d_keys;
    uint32_t    *d_rowid;
    uint32_t     slots_log2;
    uint2       *d_bloom;
    uint32_t     bloom_blocks_log2;       // 0 = no prefilter
    uint32_t     nrows, ncols;
    uint32_t    *d_counts;
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
    // flags: [0] wide windows seen in the current batch, [1] table build errors
    uint32_t    *d_flags;
    // timing
    std::vector<hipEvent_t> ev;        // begin/end pairs
    double       timed_ms;
    uint64_t     timed_launches;
    // options
    long         table_load_pct;
    long         bloom_bits_log2;
    long         stats;               // debug: count live windows / filter loads / table probes
    long         ablate;              // timing experiments: kernel variants that skip memory stages
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
    sk_ctx *c = new (std::nothrow) sk_ctx();
    if (!c) return SK_E_NOMEM;
    c->device = device;
    c->table_load_pct = 50;
    c->bloom_bits_log2 = -1;          // -1 = automatic (sized for the L2), 0 = off
    c->err[0] = 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return SK_E_NODEVICE; }
    if (hipMalloc((void **)&c->d_flags, 16 * sizeof(uint32_t)) != hipSuccess) { delete c; return SK_E_NOMEM; }
    hipMemsetAsync(c->d_flags, 0, 16 * sizeof(uint32_t), c->stream);
    signed char comp[256];
    sk_fill_complement(comp);
    if (hipMemcpyToSymbol(HIP_SYMBOL(sk_comp_dev), comp, sizeof comp) != hipSuccess) { delete c; return SK_E_NODEVICE; }
    *out = c;
    return SK_OK;
}

static void sk_table_release(sk_ctx *c)
{
    hipFree(c->d_keys); c->d_keys = NULL;
    hipFree(c->d_rowid); c->d_rowid = NULL;
    hipFree(c->d_bloom); c->d_bloom = NULL;
    hipFree(c->d_counts); c->d_counts = NULL;
    hipFree(c->d_wide_keys); c->d_wide_keys = NULL;
    hipFree(c->d_wide_rows); c->d_wide_rows = NULL;
    hipFree(c->d_wide_index); c->d_wide_index = NULL;
    c->nrows = c->ncols = c->nwide = 0;
}

extern "C" void sk_ctx_destroy(sk_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    sk_table_release(c);
    for (int i = 0; i < SK_NSTAGE; i++) {
        if (c->h_stage[i]) hipHostFree(c->h_stage[i]);
        if (c->d_stage[i]) hipFree(c->d_stage[i]);
        if (c->stage_done[i]) hipEventDestroy(c->stage_done[i]);
    }
    for (hipEvent_t e : c->ev) hipEventDestroy(e);
    hipFree(c->d_flags);
    hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int sk_set_option(sk_ctx *c, const char *name, long value)
{
    if (!c || !name) return SK_E_ARG;
    if (!strcmp(name, "table_load_pct")) { if (value < 5 || value > 90) return SK_E_ARG; c->table_load_pct = value; return SK_OK; }
    if (!strcmp(name, "bloom_bits_log2")) { if (value < -1 || value > 34 || (value > 0 && value < 10)) return SK_E_ARG; c->bloom_bits_log2 = value; return SK_OK; }
    if (!strcmp(name, "stats")) { c->stats = value != 0; return SK_OK; }
    if (!strcmp(name, "ablate")) { c->ablate = value; return SK_OK; }
    return sk_fail(c, SK_E_ARG, "unknown option %s", name);
}

extern "C" int sk_table_load(sk_ctx *c, const uint64_t *keys, uint32_t nrows, uint32_t ncols)
{
    if (!c || (!keys && nrows) || ncols == 0 || ncols > 16) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    sk_table_release(c);

    uint32_t lg = 10;
    while (((uint64_t)1 << lg) * (uint64_t)c->table_load_pct < (uint64_t)nrows * 100ull && lg < 31) lg++;
    const uint64_t slots = (uint64_t)1 << lg;
    c->slots_log2 = lg;
    SK_HIP(c, hipMalloc((void **)&c->d_keys, slots * sizeof(uint64_t)));
    SK_HIP(c, hipMalloc((void **)&c->d_rowid, slots * sizeof(uint32_t)));
    const size_t cbytes = (size_t)(nrows ? nrows : 1) * ncols * sizeof(uint32_t);
    SK_HIP(c, hipMalloc((void **)&c->d_counts, cbytes));
    SK_HIP(c, hipMemsetAsync(c->d_counts, 0, cbytes, c->stream));
    SK_HIP(c, hipMemsetAsync(c->d_rowid, 0, slots * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(sk_fill64, dim3(2048), dim3(256), 0, c->stream, c->d_keys, slots, SK_EMPTY64);
    SK_HIP(c, hipMemsetAsync(c->d_flags, 0, 16 * sizeof(uint32_t), c->stream));
    if (nrows) {
        uint64_t *d_in = NULL;
        SK_HIP(c, hipMalloc((void **)&d_in, (size_t)nrows * sizeof(uint64_t)));
        SK_HIP(c, hipMemcpyAsync(d_in, keys, (size_t)nrows * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(sk_table_insert, dim3((nrows + 255) / 256), dim3(256), 0, c->stream,
                           d_in, nrows, c->d_keys, c->d_rowid, (uint32_t)(slots - 1), c->d_flags);
        // minimizer filter: automatic size = smallest power of two >= 4 bits per key (the set holds
        // ~nrows/8 minimizers, i.e. ~32 bits each: false positives ~0.1 %); 2 MiB for a 5 Mbp strain
        long bb = c->bloom_bits_log2;
        if (bb < 0) { bb = 12; while (bb < 30 && ((uint64_t)1 << bb) < (uint64_t)nrows * 4ull) bb++; }
        c->bloom_blocks_log2 = 0;
        if (bb > 0) {
            const uint32_t blocks_log2 = (uint32_t)bb - 6u;
            const size_t bbytes = ((size_t)1 << blocks_log2) * sizeof(uint2);
            SK_HIP(c, hipMalloc((void **)&c->d_bloom, bbytes));
            SK_HIP(c, hipMemsetAsync(c->d_bloom, 0, bbytes, c->stream));
            hipLaunchKernelGGL(sk_bloom_insert, dim3((nrows + 255) / 256), dim3(256), 0, c->stream,
                               d_in, nrows, (uint32_t *)c->d_bloom, (uint32_t)(((uint64_t)1 << blocks_log2) - 1));
            c->bloom_blocks_log2 = blocks_log2;
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
    SK_HIP(c, hipMemcpy(c->d_wide_rows, rows, (size_t)nwide * sizeof(uint32_t), hipMemcpyHostToDevice));
    SK_HIP(c, hipMemcpy(c->d_wide_index, index.data(), (size_t)wslots * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->wide_mask = wslots - 1;
    c->nwide = nwide;
    return SK_OK;
}

// launch main + wide kernels over one device-resident batch
static int sk_launch_scan(sk_ctx *c, const uint8_t *d_stream, uint64_t nbytes, uint64_t emit_begin, uint32_t col)
{
    if (nbytes <= emit_begin) return SK_OK;
    const uint64_t ntiles = (nbytes + SK_TILE - 1) / SK_TILE;
    if (ntiles > 0x7FFFFFFFull) return sk_fail(c, SK_E_ARG, "batch too large");
    sk_table_view tv;
    tv.keys = c->d_keys; tv.rowid = c->d_rowid; tv.mask = (uint32_t)(((uint64_t)1 << c->slots_log2) - 1);
    tv.bloom = c->d_bloom;
    tv.bloom_mask = c->bloom_blocks_log2 ? (uint32_t)(((uint64_t)1 << c->bloom_blocks_log2) - 1) : 0u;
    sk_wide_view wv;
    wv.keys31 = c->d_wide_keys; wv.rows = c->d_wide_rows; wv.index = c->d_wide_index;
    wv.wmask = c->wide_mask; wv.nwide = c->nwide;
    uint32_t *counts = c->d_counts + (size_t)col * c->nrows;

    SK_HIP(c, hipMemsetAsync(c->d_flags, 0, sizeof(uint32_t), c->stream));
    hipEvent_t e0 = NULL, e1 = NULL;
    const bool timed = c->ev.size() < 2 * 8192;
    if (timed) {
        SK_HIP(c, hipEventCreate(&e0));
        SK_HIP(c, hipEventCreate(&e1));
        SK_HIP(c, hipEventRecord(e0, c->stream));
    }
    if (c->ablate == 1 && c->bloom_blocks_log2)
        hipLaunchKernelGGL((sk_scan_main<true, false, 1>), dim3((uint32_t)ntiles), dim3(SK_THREADS), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, counts, c->d_flags);
    else if (c->ablate == 2 && c->bloom_blocks_log2)
        hipLaunchKernelGGL((sk_scan_main<true, false, 2>), dim3((uint32_t)ntiles), dim3(SK_THREADS), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, counts, c->d_flags);
    else if (c->stats && c->bloom_blocks_log2)
        hipLaunchKernelGGL((sk_scan_main<true, true, 0>), dim3((uint32_t)ntiles), dim3(SK_THREADS), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, counts, c->d_flags);
    else if (c->bloom_blocks_log2)
        hipLaunchKernelGGL((sk_scan_main<true, false, 0>), dim3((uint32_t)ntiles), dim3(SK_THREADS), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, counts, c->d_flags);
    else
        hipLaunchKernelGGL((sk_scan_main<false, false, 0>), dim3((uint32_t)ntiles), dim3(SK_THREADS), 0, c->stream,
                           d_stream, nbytes, emit_begin, tv, counts, c->d_flags);
    if (timed) {
        SK_HIP(c, hipEventRecord(e1, c->stream));
        c->ev.push_back(e0);
        c->ev.push_back(e1);
    }
    uint64_t wblocks = (nbytes - emit_begin + 255) / 256;
    if (wblocks > 16384) wblocks = 16384;
    hipLaunchKernelGGL(sk_scan_wide, dim3((uint32_t)wblocks), dim3(256), 0, c->stream,
                       d_stream, nbytes, emit_begin, tv, wv, counts, c->d_flags);
    SK_HIP(c, hipGetLastError());
    return SK_OK;
}

extern "C" int sk_scan_device(sk_ctx *c, const void *dev_stream, uint64_t nbytes, uint32_t col)
{
    if (!c || (!dev_stream && nbytes)) return SK_E_ARG;
    if (!c->d_keys) return sk_fail(c, SK_E_STATE, "no table loaded");
    if (col >= c->ncols) return sk_fail(c, SK_E_ARG, "column %u out of range", col);
    if (((uintptr_t)dev_stream & 15u) != 0) return sk_fail(c, SK_E_ARG, "device stream must be 16-byte aligned");
    SK_HIP(c, hipSetDevice(c->device));
    return sk_launch_scan(c, (const uint8_t *)dev_stream, nbytes, 0, col);
}

static int sk_stage_init(sk_ctx *c)
{
    if (c->h_stage[0]) return SK_OK;
    for (int i = 0; i < SK_NSTAGE; i++) {
        SK_HIP(c, hipHostMalloc((void **)&c->h_stage[i], SK_STAGE_BYTES, hipHostMallocDefault));
        SK_HIP(c, hipMalloc((void **)&c->d_stage[i], SK_STAGE_BYTES));
        SK_HIP(c, hipEventCreateWithFlags(&c->stage_done[i], hipEventDisableTiming));
    }
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
    SK_HIP(c, hipMemcpyAsync(out, c->d_counts + (size_t)col * c->nrows, (size_t)c->nrows * 4, hipMemcpyDeviceToHost, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_counts_set(sk_ctx *c, uint32_t col, const uint32_t *in)
{
    if (!c || !in) return SK_E_ARG;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipMemcpyAsync(c->d_counts + (size_t)col * c->nrows, in, (size_t)c->nrows * 4, hipMemcpyHostToDevice, c->stream));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_counts_zero(sk_ctx *c, uint32_t col)
{
    if (!c) return SK_E_ARG;
    if (!c->d_counts || col >= c->ncols) return sk_fail(c, SK_E_ARG, "bad column");
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipMemsetAsync(c->d_counts + (size_t)col * c->nrows, 0, (size_t)c->nrows * 4, c->stream));
    return SK_OK;
}

extern "C" void *sk_counts_device_ptr(sk_ctx *c) { return c ? (void *)c->d_counts : NULL; }
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
        hipEventDestroy(c->ev[i]);
        hipEventDestroy(c->ev[i + 1]);
    }
    c->ev.clear();
    if (total_ms) *total_ms = c->timed_ms;
    if (launches) *launches = c->timed_launches;
    if (reset) { c->timed_ms = 0; c->timed_launches = 0; }
    return SK_OK;
}

// debug statistics accumulated by sk_scan_main<.., STATS=true> since the table was loaded:
// out[0] = windows looked up, out[1] = prefilter block loads, out[2] = table probes
extern "C" int sk_scan_stats(sk_ctx *c, uint64_t out[3])
{
    if (!c || !out) return SK_E_ARG;
    uint64_t raw[4];
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipStreamSynchronize(c->stream));
    SK_HIP(c, hipMemcpy(raw, c->d_flags + 4, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    out[0] = raw[0]; out[1] = raw[1]; out[2] = raw[2];
    return SK_OK;
}

// RCCL is resolved lazily so that the library loads on hosts without it
extern "C" int sk_counts_allreduce(sk_ctx *c, void *rccl_comm)
{
    if (!c || !rccl_comm) return SK_E_ARG;
    if (!c->d_counts) return sk_fail(c, SK_E_STATE, "no table loaded");
    typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    static allreduce_fn fn = NULL;
    if (!fn) {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return sk_fail(c, SK_E_RCCL, "cannot load librccl: %s", dlerror());
        fn = (allreduce_fn)dlsym(h, "ncclAllReduce");
        if (!fn) return sk_fail(c, SK_E_RCCL, "ncclAllReduce not found");
    }
    SK_HIP(c, hipSetDevice(c->device));
    const int ncclUint32 = 3, ncclSum = 0;            // rccl.h: ncclDataType_t / ncclRedOp_t
    const int rc = fn(c->d_counts, c->d_counts, (size_t)c->nrows * c->ncols, ncclUint32, ncclSum, rccl_comm, c->stream);
    if (rc != 0) return sk_fail(c, SK_E_RCCL, "ncclAllReduce returned %d", rc);
    SK_HIP(c, hipStreamSynchronize(c->stream));
    return SK_OK;
}

extern "C" int sk_dev_alloc(sk_ctx *c, void **dev, uint64_t nbytes)
{
    if (!c || !dev) return SK_E_ARG;
    SK_HIP(c, hipSetDevice(c->device));
    SK_HIP(c, hipMalloc(dev, nbytes ? nbytes : 16));
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
