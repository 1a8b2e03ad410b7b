// tools/h2d_probe.hip -- what the box's host -> device path gives (GPU box): page-locked host memory to HBM by hipMemcpyAsync,
// one to four streams side by side, 8 MB (a 30x sample's BGZF payloads) to 256 MB per copy; and the other direction.  The
// from-BAM leg sends 8.7 MB of compressed payloads per sample: at B GB/s it cannot pass B / 8.7 MB samples a second.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_probe tools/h2d_probe.hip && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <vector>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(int argc, char** argv) {
    const size_t MAXB = 256u << 20;
    const int NS = 4;
    uint8_t* h[NS]; uint8_t* d[NS]; hipStream_t st[NS];
    for (int k = 0; k < NS; ++k) {
        if (hipHostMalloc((void**)&h[k], MAXB, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&d[k], MAXB) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
        memset(h[k], k + 1, MAXB);
        (void)hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
    }
    printf("{\"rows\": [");
    bool first = true;
    for (int dir = 0; dir < 2; ++dir)
        for (int streams : {1, 2, 4})
            for (size_t bytes : {(size_t)8 << 20, (size_t)32 << 20, (size_t)256 << 20}) {
                const int reps = bytes >= (64u << 20) ? 6 : 24;
                double best = 1e30;
                for (int trial = 0; trial < 3; ++trial) {
                    for (int k = 0; k < streams; ++k) (void)hipStreamSynchronize(st[k]);
                    const double t0 = now();
                    for (int r = 0; r < reps; ++r)
                        for (int k = 0; k < streams; ++k)
                            (void)(dir == 0 ? hipMemcpyAsync(d[k], h[k], bytes, hipMemcpyHostToDevice, st[k]) : hipMemcpyAsync(h[k], d[k], bytes, hipMemcpyDeviceToHost, st[k]));
                    for (int k = 0; k < streams; ++k) (void)hipStreamSynchronize(st[k]);
                    const double dt = now() - t0;
                    if (dt < best) best = dt;
                }
                printf("%s{\"dir\": \"%s\", \"streams\": %d, \"MB_per_copy\": %zu, \"GB_per_s\": %.2f}", first ? "" : ", ", dir ? "d2h" : "h2d", streams, bytes >> 20,
                       (double)bytes * reps * streams / best * 1e-9);
                first = false;
            }
    printf("]}\n");
    return 0;
}
