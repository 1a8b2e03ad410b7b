#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <vector>
#include <ctime>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main() {
    (void)hipInit(0); (void)hipSetDevice(0); (void)hipFree(nullptr);
    const size_t B = 425u << 20;
    void* p[3];
    double t0 = now();
    for (int k = 0; k < 3; ++k) (void)hipHostMalloc(&p[k], B, hipHostMallocDefault);
    printf("sequential 3 x 425 MB hipHostMalloc: %.3f s\n", now() - t0);
    t0 = now();
    for (int k = 0; k < 3; ++k) (void)hipHostFree(p[k]);
    printf("free: %.3f s\n", now() - t0);
    t0 = now();
    std::vector<std::thread> th;
    for (int k = 0; k < 3; ++k) th.emplace_back([&, k] { (void)hipSetDevice(0); (void)hipHostMalloc(&p[k], B, hipHostMallocDefault); });
    for (auto& t : th) t.join();
    printf("three threads at once: %.3f s\n", now() - t0);
    t0 = now();
    for (int k = 0; k < 3; ++k) (void)hipHostFree(p[k]);
    printf("free: %.3f s\n", now() - t0);
    t0 = now();
    void* q; (void)hipHostMalloc(&q, 3 * B, hipHostMallocDefault);
    printf("one 1275 MB hipHostMalloc: %.3f s\n", now() - t0);
    t0 = now(); (void)hipHostFree(q); printf("free: %.3f s\n", now() - t0);
    void* d; t0 = now(); (void)hipMalloc(&d, (size_t)2 << 30); printf("hipMalloc 2 GB: %.3f s\n", now() - t0);
    t0 = now(); (void)hipFree(d); printf("hipFree: %.3f s\n", now() - t0);
    return 0;
}
