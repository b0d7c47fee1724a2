// tools/probes/gather_policy.hip -- does the cache policy of a load change the rate at which a CU gets random 8-byte words out of an
// L2-resident table (the level-1 filter's lookups: 0.5 per clock and CU with plain loads)?  plain / nt / sc0 / sc1 / sc0 sc1 / sc0 sc1 nt
//   hipcc -O3 --offload-arch=gfx950 tools/probes/gather_policy.hip -o tools/probes/gather_policy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16; return h; }
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
// Four loads and their wait in ONE asm statement, outputs early-clobber: the compiler does not know that an asm load's result arrives
// later -- left to itself it re-used a destination pair as the next load's address while the first was still in flight (the data
// landed on the address: a wild load; the first version of this probe faulted that way).
#define LD4(POLSTR) asm volatile("global_load_dwordx2 %0, %4, off " POLSTR "\n\tglobal_load_dwordx2 %1, %5, off " POLSTR "\n\t" \
                                 "global_load_dwordx2 %2, %6, off " POLSTR "\n\tglobal_load_dwordx2 %3, %7, off " POLSTR "\n\ts_waitcnt vmcnt(0)" \
                                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory")
template <int POL>
__global__ __launch_bounds__(256) void gather(const u2 *__restrict__ tab, uint32_t mask, uint32_t iters, uint64_t *out)
{
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint64_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const u2 *p[4];
        u2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { x = mix(x + 0x9E3779B9u); p[u] = tab + (x % mask); }
        if (POL == 0) LD4("");
        if (POL == 1) LD4("nt");
        if (POL == 2) LD4("sc0");
        if (POL == 3) LD4("sc1");
        if (POL == 4) LD4("sc0 sc1");
        if (POL == 5) LD4("sc0 sc1 nt");
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x + v[u].y;
    }
    if (acc == 0x1234567) out[0] = acc;
}
template <int POL> static void run(size_t bytes, const char *name)
{
    size_t n = bytes / 8;
    u2 *tab; uint64_t *out;
    CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(tab, 1, bytes));
    const uint32_t iters = 512; const int blocks = 256 * 8;        // (8 workgroups = 32 waves per CU, 4 loads in flight per lane)
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<POL>), dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)n, iters, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<POL>), dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)n, iters, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-12s table %6.1f MiB : %7.1f G loads/s\n", name, bytes / 1048576.0, (double)blocks * 256 * iters * 4 / ms / 1e6);
    CK(hipFree(tab)); CK(hipFree(out));
}
int main()
{
    for (size_t s : {(size_t)16 << 10, (size_t)3 << 20, (size_t)32 << 20}) {
        run<0>(s, "plain"); run<1>(s, "nt"); run<2>(s, "sc0"); run<3>(s, "sc1"); run<4>(s, "sc0 sc1"); run<5>(s, "sc0 sc1 nt");
    }
    return 0;
}
