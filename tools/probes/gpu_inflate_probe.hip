// tools/probes/gpu_inflate_probe.hip -- EXPERIMENT (not product code): how fast does an MI355X run the DEFLATE symbol loop?
// The host half (gz_blocks_dump.c) has decoded the file once and written, per dynamic block, the bit where its symbols
// start, its decode tables and where its bytes go.  Here every block is decoded independently, ONE LANE PER BLOCK
// (v1: tables and output in global memory), and the lengths are checked against the host's.  Matches that reach back
// before the block's own start cannot be resolved this way (the bytes belong to another lane): they are written as zeros
// and counted -- a real device inflate would carry placeholders as sk_gzpar.h does.  What is measured is the rate of the
// loop (symbols/s, bytes of text/s), i.e. the ceiling of a lane-per-block design, to decide whether a device-side inflate
// is worth building.
//     hipcc -O3 --offload-arch=gfx950 -o gpu_inflate_probe tools/probes/gpu_inflate_probe.hip ; ./gpu_inflate_probe file.gz blocks.bin
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#define LITLEN_BITS 11
#define DIST_BITS 8
#define NLIT 6624
#define NDIST 4096
enum { K_LIT = 0, K_LIT2 = 1, K_LEN = 2, K_EOB = 3, K_SUB = 4, K_BAD = 5, K_DIST = 6 };
struct blk { uint64_t sym_bit, out_off; uint32_t out_len, pad; };

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void inflate_blocks(const uint32_t *__restrict__ comp32, uint64_t nwords, const blk *__restrict__ b, const uint32_t *__restrict__ tabs, uint32_t nb,
                               uint8_t *text, uint32_t *got_len, unsigned long long *stats)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const uint32_t *lt = tabs + (size_t)i * (NLIT + NDIST), *dt = lt + NLIT;
    uint64_t wp = b[i].sym_bit >> 5;
    uint64_t bitbuf = (uint64_t)comp32[wp] >> (b[i].sym_bit & 31u);
    uint32_t bitcnt = 32u - (uint32_t)(b[i].sym_bit & 31u);
    wp++;
    uint8_t *const base = text + b[i].out_off;
    const uint32_t cap = b[i].out_len + 258u;
    uint32_t n = 0, nsym = 0, nfar = 0;
    int rc = 0;
#define REFILL() do { if (bitcnt <= 32u) { bitbuf |= (uint64_t)(wp < nwords ? comp32[wp] : 0u) << bitcnt; wp++; bitcnt += 32u; } } while (0)
    for (;;) {
        REFILL();
        uint32_t e = lt[bitbuf & ((1u << LITLEN_BITS) - 1u)];
        uint32_t kind = (e >> 4) & 15u;
        nsym++;
        if (n > cap) { rc = 2; break; }
        if (kind <= K_LIT2) {
            bitbuf >>= e & 15u; bitcnt -= e & 15u;
            base[n] = (uint8_t)(e >> 16);
            if (kind == K_LIT2) base[n + 1] = (uint8_t)(e >> 24);
            n += 1u + (kind == K_LIT2);
            continue;
        }
        if (kind == K_SUB) {
            bitbuf >>= LITLEN_BITS; bitcnt -= LITLEN_BITS;
            e = lt[(e >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << ((e >> 8) & 255u)) - 1u))];
            kind = (e >> 4) & 15u;
            if (kind == K_LIT) { bitbuf >>= e & 15u; bitcnt -= e & 15u; base[n++] = (uint8_t)(e >> 16); continue; }
        }
        bitbuf >>= e & 15u; bitcnt -= e & 15u;
        if (kind == K_LEN) {
            const uint32_t xb = (e >> 8) & 255u;
            uint32_t len = (e >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << xb) - 1u));
            bitbuf >>= xb; bitcnt -= xb;
            REFILL();
            uint32_t d = dt[bitbuf & ((1u << DIST_BITS) - 1u)];
            if (((d >> 4) & 15u) == K_SUB) {
                bitbuf >>= DIST_BITS; bitcnt -= DIST_BITS;
                d = dt[(d >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << ((d >> 8) & 255u)) - 1u))];
            }
            bitbuf >>= d & 15u; bitcnt -= d & 15u;
            if (((d >> 4) & 15u) != K_DIST) { rc = 3; break; }
            const uint32_t db = (d >> 8) & 255u;
            const uint32_t dist = (d >> 16) + (uint32_t)(bitbuf & (((uint64_t)1 << db) - 1u));
            bitbuf >>= db; bitcnt -= db;
            if (dist > n) {                                   // reaches into another block's bytes
                const uint32_t far = dist - n < len ? dist - n : len;
                for (uint32_t k = 0; k < far; k++) base[n + k] = 0;
                for (uint32_t k = far; k < len; k++) base[n + k] = base[n + k - dist];
                nfar++;
            } else {
                for (uint32_t k = 0; k < len; k++) base[n + k] = base[n + k - dist];
            }
            n += len;
            continue;
        }
        if (kind == K_EOB) break;
        rc = 4;
        break;
    }
    got_len[i] = rc ? 0xFFFFFFFFu : n;
    atomicAdd(&stats[0], (unsigned long long)nsym);
    atomicAdd(&stats[1], (unsigned long long)nfar);
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: gpu_inflate_probe file.gz blocks.bin\n"); return 2; }
    struct stat st;
    int fd = open(argv[1], O_RDONLY);
    if (fd < 0 || fstat(fd, &st)) return 2;
    const size_t ncomp = (size_t)st.st_size;
    void *m = mmap(NULL, ncomp, PROT_READ, MAP_PRIVATE, fd, 0);
    FILE *f = fopen(argv[2], "rb");
    uint64_t h[2];
    if (!f || fread(h, 8, 2, f) != 2) return 2;
    const uint32_t nb = (uint32_t)h[0];
    std::vector<blk> b(nb);
    std::vector<uint32_t> tabs((size_t)nb * (NLIT + NDIST));
    if (fread(b.data(), sizeof(blk), nb, f) != nb || fread(tabs.data(), (NLIT + NDIST) * 4, nb, f) != nb) return 2;
    fclose(f);
    uint64_t text_bytes = 0;
    for (auto &x : b) text_bytes += x.out_len;
    const uint64_t text_cap = b[nb - 1].out_off + b[nb - 1].out_len + 4096;
    uint32_t *d_comp, *d_tabs, *d_len;
    blk *d_b;
    uint8_t *d_text;
    unsigned long long *d_stats;
    const size_t nwords = (ncomp + 3) / 4;
    CK(hipMalloc(&d_comp, nwords * 4 + 16));
    CK(hipMemset(d_comp, 0, nwords * 4 + 16));
    CK(hipMemcpy(d_comp, m, ncomp, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_tabs, tabs.size() * 4));
    CK(hipMemcpy(d_tabs, tabs.data(), tabs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_b, nb * sizeof(blk)));
    CK(hipMemcpy(d_b, b.data(), nb * sizeof(blk), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_text, text_cap));
    CK(hipMalloc(&d_len, nb * 4));
    CK(hipMalloc(&d_stats, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int threads : {64, 256}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(d_stats, 0, 16));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(inflate_blocks, dim3((nb + threads - 1) / threads), dim3(threads), 0, 0, d_comp, (uint64_t)nwords, d_b, d_tabs, nb, d_text, d_len, d_stats);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint32_t> got(nb);
            unsigned long long stats[2];
            CK(hipMemcpy(got.data(), d_len, nb * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(stats, d_stats, 16, hipMemcpyDeviceToHost));
            uint32_t bad = 0;
            for (uint32_t i = 0; i < nb; i++) bad += got[i] != b[i].out_len;
            printf("lane per block, %3d threads per workgroup: %u blocks, %.2f GB of text in %.2f ms = %.1f GB/s of text, %.2f G symbols/s; "
                   "%llu matches reached before their block (%.1f %% of the symbols); lengths wrong in %u blocks\n",
                   threads, nb, text_bytes / 1e9, ms, text_bytes / (ms * 1e-3) / 1e9, stats[0] / (ms * 1e-3) / 1e9, stats[1], 100.0 * stats[1] / stats[0], bad);
        }
    }
    return 0;
}
