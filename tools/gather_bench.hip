// gather_bench.hip -- calibrates the memory system for the probe step of sk_scan_main:
// random independent loads (4 or 8 bytes) from a table of a given size, U loads in flight per lane.
// Not part of the product; numbers recorded in DESIGN.md.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o tools/gather_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16; return h; }

template <typename T, int U>
__global__ __launch_bounds__(256) void gather(const T *__restrict__ tab, uint32_t mask, uint32_t iters, uint64_t *out)
{
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint64_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        uint32_t idx[U];
#pragma unroll
        for (int u = 0; u < U; u++) { x = mix(x + 0x9E3779B9u); idx[u] = x & mask; }
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = tab[idx[u]];
#pragma unroll
        for (int u = 0; u < U; u++) acc += (uint64_t)v[u];
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <typename T, int U>
static void run(size_t bytes, int waves_per_cu)
{
    size_t n = bytes / sizeof(T);
    T *tab; uint64_t *out;
    CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(tab, 1, bytes));
    uint32_t iters = 2048 / U;
    int blocks = 256 * waves_per_cu / 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather<T, U>), dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)(n - 1), iters, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather<T, U>), dim3(blocks), dim3(256), 0, 0, tab, (uint32_t)(n - 1), iters, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double loads = (double)blocks * 256 * iters * U;
    printf("%2zu-byte loads  table %7.1f MiB  U=%d  waves/CU=%2d : %8.1f G loads/s  (%.3f ms)\n", sizeof(T), bytes / 1048576.0, U,
           waves_per_cu, loads / ms / 1e6, ms);
    CK(hipFree(tab)); CK(hipFree(out));
}

int main()
{
    size_t sizes[] = {64ull << 10, 1ull << 20, 2ull << 20, 4ull << 20, 8ull << 20, 16ull << 20, 32ull << 20, 64ull << 20, 128ull << 20, 256ull << 20, 1024ull << 20};
    for (size_t s : sizes) { run<uint32_t, 4>(s, 16); run<uint64_t, 4>(s, 16); }
    printf("-- occupancy / ILP sweep on 128 MiB, 8-byte --\n");
    run<uint64_t, 1>(128ull << 20, 16); run<uint64_t, 2>(128ull << 20, 16); run<uint64_t, 8>(128ull << 20, 16);
    run<uint64_t, 4>(128ull << 20, 8); run<uint64_t, 4>(128ull << 20, 32); run<uint64_t, 8>(128ull << 20, 32);
    printf("-- 4-byte, 4 MiB and 2 MiB (L2-resident filter) sweep --\n");
    run<uint32_t, 8>(2ull << 20, 32); run<uint32_t, 8>(4ull << 20, 32); run<uint32_t, 4>(2ull << 20, 8);
    return 0;
}
