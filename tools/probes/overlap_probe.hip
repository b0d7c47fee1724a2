// overlap_probe.hip -- can a CU overlap the scan kernel's two memory streams?  (round 4)
// The scan kernel's launch time is, by ablation, the SUM of "read the record stream" (HBM-bound alone: ~6 TB/s) and "one random 8-byte
// lookup per 16 bytes in a 3 MiB L2-resident filter" (L2-request-bound alone), although eight workgroups per CU are in different phases
// at any time.  This probe runs the two access patterns bare, alone and together, to see what the memory pipeline allows:
//   mode 0  stream only      every thread loads 8 x 16 B (nt) of its workgroup's 32 KiB tile, xor-reduces
//   mode 1  lookups only     every thread does 8 independent 8-byte loads at hashed places of the table
//   mode 2  both, independent: the lookups' addresses do not depend on the stream (issued right behind the stream loads)
//   mode 3  both, dependent:   a lookup's address is a hash of the 16 bytes just loaded (the real kernel's shape)
//   mode 4  both, by wave:     waves 0-1 of a workgroup stream the whole tile, waves 2-3 do all its lookups
//   hipcc -O3 --offload-arch=gfx950 tools/probes/overlap_probe.hip -o tools/probes/overlap_probe && tools/probes/overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define TILE 32768u

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 15; x *= 0x9E3779B1u; x ^= x >> 13; return x; }

__global__ void fill(uint32_t *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = mix((uint32_t)i) ^ mix((uint32_t)(i >> 32) + 77u); }

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t *__restrict__ stream, const uint2 *__restrict__ table, uint32_t blocks, uint32_t *out)
{
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const uint64_t t0 = (uint64_t)blockIdx.x * TILE;
    uint32_t acc = 0;
    u4 v[8];
    uint2 b[8];
    if (MODE == 4) {
        if (wave < 2) {                                          // 128 threads stream the tile: 16 loads each
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = __builtin_nontemporal_load((const u4 *)(stream + t0 + ((uint64_t)(h * 8 + i) * 128u + tid) * 16u));
#pragma unroll
                for (int i = 0; i < 8; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
            }
        } else {                                                 // 128 threads do the tile's 2048 lookups: 16 each
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int i = 0; i < 8; i++) b[i] = table[__umulhi(mix(blockIdx.x * 2048u + (h * 8 + i) * 128u + tid), blocks)];
#pragma unroll
                for (int i = 0; i < 8; i++) acc ^= b[i].x ^ b[i].y;
            }
        }
    } else {
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = __builtin_nontemporal_load((const u4 *)(stream + t0 + ((uint64_t)i * 256u + tid) * 16u));
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = table[__umulhi(mix(blockIdx.x * 2048u + i * 256u + tid), blocks)];
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = table[__umulhi(mix(v[i].x ^ v[i].y ^ v[i].z ^ v[i].w), blocks)];
        }
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        }
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= b[i].x ^ b[i].y;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static int run(const char *what, const uint8_t *s, const uint2 *t, uint32_t blocks, uint32_t *out, uint32_t ntiles)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(ntiles), dim3(256), 0, 0, s, t, blocks, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<MODE>, dim3(ntiles), dim3(256), 0, 0, s, t, blocks, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("%-28s %8.4f ms   stream %6.2f TB/s   lookups %6.1f G/s\n", what, ms, MODE == 1 ? 0.0 : (double)ntiles * TILE / ms / 1e9,
           MODE == 0 ? 0.0 : (double)ntiles * 2048.0 / ms / 1e6);
    return 0;
}

int main(int argc, char **argv)
{
    const uint32_t ntiles = 46080, table_kib = argc > 1 ? (uint32_t)atoi(argv[1]) : 3072;
    const uint32_t blocks = table_kib * 1024u / 8u;
    uint8_t *s; uint2 *t; uint32_t *out;
    CK(hipMalloc(&s, (size_t)ntiles * TILE + 4096));
    CK(hipMalloc(&t, (size_t)blocks * 8));
    CK(hipMalloc(&out, 64));
    { const size_t nw = ((size_t)ntiles * TILE + 4096) / 4; hipLaunchKernelGGL(fill, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, 0, (uint32_t *)s, nw); CK(hipDeviceSynchronize()); }
    CK(hipMemset(t, 0x5A, (size_t)blocks * 8));
    printf("tiles %u x 32 KiB = %.2f GB, table %u KiB, 8 loads + 8 lookups per thread\n", ntiles, (double)ntiles * TILE / 1e9, table_kib);
    if (run<0>("0 stream only", s, t, blocks, out, ntiles)) return 1;
    if (run<1>("1 lookups only", s, t, blocks, out, ntiles)) return 1;
    if (run<2>("2 both, independent", s, t, blocks, out, ntiles)) return 1;
    if (run<3>("3 both, lookups depend", s, t, blocks, out, ntiles)) return 1;
    if (run<4>("4 both, split by wave", s, t, blocks, out, ntiles)) return 1;
    return 0;
}
