// tools/inflate_prof.hip -- where a wavefront of inflate_kernel spends its cycles.
// Builds the decoder with per-phase cycle counters (-DINFLATE_PROF: s_memtime marks in lane 0's control flow) next to a
// plain build of the same source, runs the BGZF blocks of a BAM file m times per call through the C ABI and prints
// the call's device time, the decode launches' time and -- profiled build -- the mean cycles per block and phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DINFLATE_PROF] -o inflate_prof tools/inflate_prof.hip
//   ./inflate_prof file.bam [samples per call = 1] [calls = 5] [crc = 1]
#include "../tredparse_amd/csrc/inflate_decode.hip"
#include "../tredparse_amd/csrc/walk.hip"
#include "../tredparse_amd/csrc/inflater_api.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s file.bam [samples per call] [calls] [crc]\n", argv[0]); return 2; }
    const int m = argc > 2 ? atoi(argv[2]) : 1, calls = argc > 3 ? atoi(argv[3]) : 5, want_crc = argc > 4 ? atoi(argv[4]) : 1;
    FILE* fp = fopen(argv[1], "rb");
    if (!fp) { perror(argv[1]); return 2; }
    std::vector<uint8_t> raw;
    uint8_t buf[1 << 16];
    for (size_t n; (n = fread(buf, 1, sizeof buf, fp)) > 0;) raw.insert(raw.end(), buf, buf + n);
    fclose(fp);
    struct Blk { size_t at, len; uint32_t crc, isize; };
    std::vector<Blk> blocks;
    for (size_t pos = 0; pos + 18 <= raw.size();) {
        const int xlen = raw[pos + 10] | raw[pos + 11] << 8, bsize = (raw[pos + 16] | raw[pos + 17] << 8) + 1;
        uint32_t crc, isize;
        memcpy(&crc, &raw[pos + bsize - 8], 4);
        memcpy(&isize, &raw[pos + bsize - 4], 4);
        if (isize > 0) blocks.push_back({pos + 12 + xlen, (size_t)bsize - 12 - xlen - 8, crc, isize});
        pos += bsize;
    }
    const int nb = (int)blocks.size(), n = nb * m;
    size_t cbytes = 0, obytes = 0;
    for (const Blk& b : blocks) { cbytes += (b.len + 3) & ~(size_t)3; obytes += b.isize; }
    tredgpu_inflater* f = nullptr;
    if (tredgpu_inflater_create(0, &f) != 0) { fprintf(stderr, "%s\n", tredgpu_inflater_last_error(nullptr)); return 1; }
    uint8_t *comp, *out;
    int64_t *coff, *ooff;
    if (tredgpu_inflater_reserve(f, (int64_t)cbytes * m, (int64_t)obytes * m, n, &comp, &out, &coff, &ooff) != 0) { fprintf(stderr, "%s\n", tredgpu_inflater_last_error(f)); return 1; }
    coff[0] = ooff[0] = 0;
    for (int k = 0; k < n; ++k) {
        const Blk& b = blocks[k % nb];
        memcpy(comp + coff[k], &raw[b.at], b.len);
        coff[k + 1] = coff[k] + (int64_t)((b.len + 3) & ~(size_t)3);
        ooff[k + 1] = ooff[k] + b.isize;
    }
#ifdef INFLATE_PROF
    unsigned long long* d_prof = nullptr;
    (void)hipMalloc((void**)&d_prof, (size_t)n * 8 * sizeof(unsigned long long));
    (void)hipMemset(d_prof, 0, (size_t)n * 8 * sizeof(unsigned long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), &d_prof, sizeof d_prof);
#endif
    std::vector<int32_t> status(n);
    std::vector<uint32_t> crc(n);
    double best_total = 1e30, best_kernel = 1e30;
    timespec w0, w1, c0, c1;
    clock_gettime(CLOCK_MONOTONIC, &w0);
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &c0);
    for (int c = 0; c < calls; ++c) {
        const int bad = tredgpu_inflate_blocks_crc(f, n, status.data(), want_crc ? crc.data() : nullptr);
        if (bad != 0) { fprintf(stderr, "call %d: %d bad blocks (%s)\n", c, bad, tredgpu_inflater_last_error(f)); return 1; }
        if (want_crc) for (int k = 0; k < n; ++k) if (crc[k] != blocks[k % nb].crc) { fprintf(stderr, "block %d: CRC mismatch\n", k); return 1; }
        double total, kernel;
        tredgpu_inflater_timing(f, &total, &kernel);
        if (c > 0 || calls == 1) { best_total = std::min(best_total, total); best_kernel = std::min(best_kernel, kernel); }
    }
    clock_gettime(CLOCK_MONOTONIC, &w1);
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &c1);
    const double wall = (w1.tv_sec - w0.tv_sec) * 1e3 + (w1.tv_nsec - w0.tv_nsec) * 1e-6, cpu = (c1.tv_sec - c0.tv_sec) * 1e3 + (c1.tv_nsec - c0.tv_nsec) * 1e-6;
    printf("{\"wall_ms_per_call\": %.3f, \"host_cpu_ms_per_call\": %.3f, ", wall / calls, cpu / calls);
    printf("\"blocks\": %d, \"samples\": %d, \"device_ms\": %.3f, \"kernel_ms\": %.3f, \"kernel_blocks_per_s\": %.0f, \"kernel_output_GBps\": %.2f", n, m, best_total,
           best_kernel, n / (best_kernel * 1e-3), obytes * (double)m / best_kernel * 1e-6);
#ifdef INFLATE_PROF
    std::vector<unsigned long long> prof((size_t)n * 8);
    (void)hipMemcpy(prof.data(), d_prof, prof.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const char* names[8] = {"header_and_tables", "window_lookups", "walk", "append", "run_queue", "crc", "windows", "total"};
    printf(", \"mean_per_block\": {");
    for (int k = 0; k < 8; ++k) {
        double sum = 0;
        for (int b = 0; b < n; ++b) sum += (double)prof[(size_t)b * 8 + k];
        printf("%s\"%s\": %.0f", k ? ", " : "", names[k], sum / n);
    }
    printf("}");
#endif
    printf("}\n");
    tredgpu_inflater_destroy(f);
    return 0;
}
