// ubench_valu.hip -- issue rate of the integer VALU instructions the SW kernel is made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip && ./ubench_valu
// Each kernel runs ITER x 64 independent instructions of one kind per wave, 8 waves per SIMD on every CU;
// the result is SIMD cycles per wave64 instruction at the clock reported by the device.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
constexpr int ITER = 4000;

#define KERNEL(NAME, ASM)                                                                      \
    __global__ __launch_bounds__(256) void NAME(int* out, int seed) {                          \
        int a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = seed * 3, c = seed * 5; \
        for (int i = 0; i < ITER; ++i) {                                                       \
            asm volatile(REP8(REP8(ASM)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc"); \
        }                                                                                      \
        if (a0 + a1 + a2 + a3 == 12345) out[0] = a0;                                           \
    }
// one "ASM" = 1 instruction; 4 accumulators rotate so consecutive instructions are independent
KERNEL(k_max,    "v_max_i32 %0, %0, %4\n\t")
KERNEL(k_max3,   "v_max3_i32 %0, %0, %4, %5\n\t")
KERNEL(k_add,    "v_add_u32 %0, %0, %4\n\t")
KERNEL(k_add3,   "v_add3_u32 %0, %0, %4, %5\n\t")
KERNEL(k_andor,  "v_and_or_b32 %0, %0, %4, %5\n\t")
KERNEL(k_sub,    "v_subrev_u32 %0, %4, %0\n\t")
KERNEL(k_maxf,   "v_max_f32 %0, %0, %4\n\t")
KERNEL(k_max3f,  "v_max3_f32 %0, %0, %4, %5\n\t")
KERNEL(k_maxu,   "v_max_u32 %0, %0, %4\n\t")
KERNEL(k_cmpgt,  "v_cmp_gt_i32 vcc, %0, %4\n\t")
KERNEL(k_cndm,   "v_cndmask_b32 %0, %0, %4, vcc\n\t")
KERNEL(k_andb,   "v_and_b32 %0, %0, %4\n\t")
KERNEL(k_mov,    "v_mov_b32 %0, %4\n\t")
KERNEL(k_maxdpp, "v_max_i32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t")
KERNEL(k_pkmax,  "v_pk_max_i16 %0, %0, %4\n\t")
KERNEL(k_pkadd,  "v_pk_add_i16 %0, %0, %4\n\t")
KERNEL(k_max_r,  "v_max_i32 %0, %0, %4\n\tv_max_i32 %1, %1, %4\n\tv_max_i32 %2, %2, %5\n\tv_max_i32 %3, %3, %5\n\t")
KERNEL(k_max3_r, "v_max3_i32 %0, %0, %4, %5\n\tv_max3_i32 %1, %1, %4, %5\n\tv_max3_i32 %2, %2, %5, %4\n\tv_max3_i32 %3, %3, %5, %4\n\t")

template <typename K>
void run(const char* name, K k, int per_asm, int* d, int cus, double mhz) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus * 8;   // 8 x 256 threads = 32 waves per CU = 8 per SIMD
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)ITER * 64 * per_asm * 8;   // 8 waves per SIMD
    const double cycles = ms * 1e-3 * mhz * 1e6;
    printf("%-10s %8.3f ms  %.2f SIMD cycles per wave64 instruction (at %.0f MHz)\n", name, ms, cycles / instr_per_simd, mhz);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int* d;
    hipMalloc(&d, 64);
    const double mhz = p.clockRate / 1000.0;
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    run("v_max_i32", k_max, 1, d, p.multiProcessorCount, mhz);
    run("v_max3_i32", k_max3, 1, d, p.multiProcessorCount, mhz);
    run("v_add_u32", k_add, 1, d, p.multiProcessorCount, mhz);
    run("v_add3_u32", k_add3, 1, d, p.multiProcessorCount, mhz);
    run("v_and_or", k_andor, 1, d, p.multiProcessorCount, mhz);
    run("v_subrev", k_sub, 1, d, p.multiProcessorCount, mhz);
    run("v_max_f32", k_maxf, 1, d, p.multiProcessorCount, mhz);
    run("v_max3_f32", k_max3f, 1, d, p.multiProcessorCount, mhz);
    run("v_max_u32", k_maxu, 1, d, p.multiProcessorCount, mhz);
    run("v_cmp_gt", k_cmpgt, 1, d, p.multiProcessorCount, mhz);
    run("v_cndmask", k_cndm, 1, d, p.multiProcessorCount, mhz);
    run("v_and_b32", k_andb, 1, d, p.multiProcessorCount, mhz);
    run("v_mov_b32", k_mov, 1, d, p.multiProcessorCount, mhz);
    run("max_dpp", k_maxdpp, 1, d, p.multiProcessorCount, mhz);
    run("pk_max_i16", k_pkmax, 1, d, p.multiProcessorCount, mhz);
    run("pk_add_i16", k_pkadd, 1, d, p.multiProcessorCount, mhz);
    run("max x4", k_max_r, 4, d, p.multiProcessorCount, mhz);
    run("max3 x4", k_max3_r, 4, d, p.multiProcessorCount, mhz);
    return 0;
}
