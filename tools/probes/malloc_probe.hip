// malloc_probe.hip -- what device allocations cost on the box: the ~20 hipMalloc calls of one strain_detect table load (sizes of a
// 5 Mbp strain), 32 strains, one thread and 16 threads; against the same bytes taken as 1 GiB chunks.
//   hipcc --offload-arch=gfx950 -O2 -o malloc_probe malloc_probe.hip -lpthread && ./malloc_probe
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <time.h>
#include <vector>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static const size_t sizes[] = {256u << 20, 120u << 20, 40u << 20, 20u << 20, 20u << 20, 20u << 20, 20u << 20, 3u << 20, 32u << 20, 2u << 20, 2u << 20,
                               20u << 20, 1u << 20, 4u << 20, 64, 1u << 20};
static void *one_strain(void *arg)
{
    std::vector<void *> *keep = (std::vector<void *> *)arg;
    hipSetDevice(0);
    for (size_t s : sizes) { void *p = NULL; if (hipMalloc(&p, s) == hipSuccess) keep->push_back(p); }
    return NULL;
}
int main()
{
    hipSetDevice(0);
    void *w; hipMalloc(&w, 1 << 20); hipFree(w);
    size_t per = 0; for (size_t s : sizes) per += s;
    {   // 32 strains, one thread
        std::vector<void *> keep; double t0 = now();
        for (int i = 0; i < 32; i++) one_strain(&keep);
        double t1 = now();
        printf("one thread : 32 x %zu allocations (%.2f GB): %.3f s = %.2f ms per call\n", sizeof sizes / sizeof *sizes, 32 * per / 1e9, t1 - t0, (t1 - t0) * 1e3 / keep.size());
        t0 = now(); for (void *p : keep) hipFree(p); printf("             freeing them: %.3f s\n", now() - t0);
    }
    {   // 32 strains on 16 threads
        std::vector<void *> keep[32]; pthread_t th[16]; double t0 = now();
        for (int r = 0; r < 2; r++) { for (int i = 0; i < 16; i++) pthread_create(&th[i], NULL, one_strain, &keep[r * 16 + i]); for (int i = 0; i < 16; i++) pthread_join(th[i], NULL); }
        printf("16 threads : the same: %.3f s wall\n", now() - t0);
        t0 = now(); for (auto &k : keep) for (void *p : k) hipFree(p); printf("             freeing them: %.3f s\n", now() - t0);
    }
    {   // the same bytes as 1 GiB chunks
        std::vector<void *> keep; double t0 = now();
        const int n = (int)((32 * per + (1u << 30) - 1) >> 30);
        for (int i = 0; i < n; i++) { void *p = NULL; if (hipMalloc(&p, (size_t)1 << 30) == hipSuccess) keep.push_back(p); }
        printf("chunks     : %d x 1 GiB: %.3f s\n", n, now() - t0);
        t0 = now(); for (void *p : keep) hipMemset(p, 0, (size_t)1 << 30); hipDeviceSynchronize(); printf("             touching them (memset): %.3f s\n", now() - t0);
        t0 = now(); for (void *p : keep) hipFree(p); printf("             freeing them: %.3f s\n", now() - t0);
    }
    return 0;
}
