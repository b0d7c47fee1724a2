// tools/probes/pin_probe.hip -- what does page-locked host memory cost on this box?  hipHostMalloc of 8/16/32 MiB, one
// after the other and from 16 threads at once; hipHostFree; first touch of the buffer.  (Measurement only.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipSetDevice(0);
    void *w; hipHostMalloc(&w, 1 << 20, hipHostMallocDefault); hipHostFree(w);
    for (size_t mib : {8, 16, 32}) {
        const size_t n = mib << 20;
        std::vector<void *> p(8);
        double t0 = now();
        for (auto &q : p) hipHostMalloc(&q, n, hipHostMallocDefault);
        double t1 = now();
        for (auto &q : p) memset(q, 1, n);
        double t2 = now();
        for (auto &q : p) hipHostFree(q);
        double t3 = now();
        printf("%2zu MiB x 8 sequential: alloc %.2f ms each, first touch %.2f ms each, free %.2f ms each\n", mib, (t1 - t0) / 8 * 1e3, (t2 - t1) / 8 * 1e3, (t3 - t2) / 8 * 1e3);
        std::vector<std::thread> th;
        std::vector<void *> q(32);
        t0 = now();
        for (int t = 0; t < 16; t++) th.emplace_back([&, t] { hipSetDevice(0); hipHostMalloc(&q[2 * t], n, hipHostMallocDefault); hipHostMalloc(&q[2 * t + 1], n, hipHostMallocDefault); });
        for (auto &x : th) x.join();
        t1 = now();
        printf("%2zu MiB x 32 from 16 threads: %.2f ms in all (%.2f ms per buffer)\n", mib, (t1 - t0) * 1e3, (t1 - t0) / 32 * 1e3);
        for (auto &x : q) hipHostFree(x);
    }
    // registering ordinary memory instead
    {
        const size_t n = 32u << 20;
        void *m = aligned_alloc(4096, n);
        memset(m, 1, n);
        double t0 = now();
        hipHostRegister(m, n, hipHostRegisterDefault);
        double t1 = now();
        hipHostUnregister(m);
        printf("hipHostRegister of 32 MiB touched memory: %.2f ms, unregister %.2f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
        free(m);
    }
    return 0;
}
