// ubench_decoder.hip -- issue rate of the VALU instructions inflate_kernel's look-ups are made of (gfx950): what a 64-bit shift
// costs against the 32-bit forms that could stand in for it.  Same method as ubench_valu.hip (64 independent instructions of one
// kind per trip, 8 waves per SIMD on every CU).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_decoder tools/ubench_decoder.hip && /tmp/ubench_decoder
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(X) X X X X X X X X
constexpr int ITER = 4000;

#define KERNEL32(NAME, ASM)                                                                    \
    __global__ __launch_bounds__(256) void NAME(int* out, int seed) {                          \
        int a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = seed * 3, c = seed * 5; \
        for (int i = 0; i < ITER; ++i) {                                                       \
            asm volatile(REP8(REP8(ASM)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20", "s21"); \
        }                                                                                      \
        if (a0 + a1 + a2 + a3 == 12345) out[0] = a0;                                           \
    }
#define KERNEL64(NAME, ASM)                                                                    \
    __global__ __launch_bounds__(256) void NAME(int* out, int seed) {                          \
        uint64_t a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; int b = seed & 7, c = seed * 5; \
        for (int i = 0; i < ITER; ++i) {                                                       \
            asm volatile(REP8(REP8(ASM)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc"); \
        }                                                                                      \
        if (a0 + a1 + a2 + a3 == 12345) out[0] = (int)a0;                                      \
    }
// (each ASM string is FOUR instructions on four accumulators, so that consecutive ones are independent)
KERNEL64(k_shr64,  "v_lshrrev_b64 %0, %4, %0\n\tv_lshrrev_b64 %1, %4, %1\n\tv_lshrrev_b64 %2, %4, %2\n\tv_lshrrev_b64 %3, %4, %3\n\t")
KERNEL64(k_shl64,  "v_lshlrev_b64 %0, %4, %0\n\tv_lshlrev_b64 %1, %4, %1\n\tv_lshlrev_b64 %2, %4, %2\n\tv_lshlrev_b64 %3, %4, %3\n\t")
KERNEL32(k_shr32,  "v_lshrrev_b32 %0, %4, %0\n\tv_lshrrev_b32 %1, %4, %1\n\tv_lshrrev_b32 %2, %4, %2\n\tv_lshrrev_b32 %3, %4, %3\n\t")
KERNEL32(k_align,  "v_alignbit_b32 %0, %0, %5, %4\n\tv_alignbit_b32 %1, %1, %5, %4\n\tv_alignbit_b32 %2, %2, %5, %4\n\tv_alignbit_b32 %3, %3, %5, %4\n\t")
KERNEL32(k_bfe,    "v_bfe_u32 %0, %0, %4, %5\n\tv_bfe_u32 %1, %1, %4, %5\n\tv_bfe_u32 %2, %2, %4, %5\n\tv_bfe_u32 %3, %3, %4, %5\n\t")
KERNEL32(k_and,    "v_and_b32 %0, %0, %4\n\tv_and_b32 %1, %1, %4\n\tv_and_b32 %2, %2, %4\n\tv_and_b32 %3, %3, %4\n\t")
KERNEL32(k_min,    "v_min_u32 %0, %0, %4\n\tv_min_u32 %1, %1, %4\n\tv_min_u32 %2, %2, %4\n\tv_min_u32 %3, %3, %4\n\t")
// (v_cndmask_b32 with the mask in vcc, 64 of them back to back, measures 22.9 cycles each -- and 2.1 as one in four among v_and_b32, 4.0
//  behind a v_cmp each: an artefact of a stream of nothing but vcc readers, not a cost real code pays; the mask in an SGPR pair here)
KERNEL32(k_cndm,   "v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n\tv_cndmask_b32_e64 %1, %1, %4, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %4, s[20:21]\n\tv_cndmask_b32_e64 %3, %3, %4, s[20:21]\n\t")
KERNEL32(k_addlsh, "v_add_lshl_u32 %0, %0, %4, 8\n\tv_add_lshl_u32 %1, %1, %4, 8\n\tv_add_lshl_u32 %2, %2, %4, 8\n\tv_add_lshl_u32 %3, %3, %4, 8\n\t")
KERNEL32(k_sdwa,   "v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\tv_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\tv_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\tv_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t")
KERNEL32(k_shl,    "v_lshlrev_b32 %0, 2, %0\n\tv_lshlrev_b32 %1, 2, %1\n\tv_lshlrev_b32 %2, 2, %2\n\tv_lshlrev_b32 %3, 2, %3\n\t")
KERNEL32(k_mbcnt,  "v_mbcnt_lo_u32_b32 %0, %4, %0\n\tv_mbcnt_lo_u32_b32 %1, %4, %1\n\tv_mbcnt_lo_u32_b32 %2, %4, %2\n\tv_mbcnt_lo_u32_b32 %3, %4, %3\n\t")

template <typename K>
void run(const char* name, K k, int* d, int cus, double mhz) {
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
    const double instr_per_simd = (double)ITER * 64 * 4 * 8;
    printf("%-16s %8.3f ms  %.2f SIMD cycles per wave64 instruction (at %.0f MHz)\n", name, ms, ms * 1e-3 * mhz * 1e6 / instr_per_simd, mhz);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int* d;
    hipMalloc(&d, 64);
    const double mhz = p.clockRate / 1000.0;
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    run("v_lshrrev_b64", k_shr64, d, p.multiProcessorCount, mhz);
    run("v_lshlrev_b64", k_shl64, d, p.multiProcessorCount, mhz);
    run("v_lshrrev_b32", k_shr32, d, p.multiProcessorCount, mhz);
    run("v_alignbit_b32", k_align, d, p.multiProcessorCount, mhz);
    run("v_bfe_u32", k_bfe, d, p.multiProcessorCount, mhz);
    run("v_and_b32", k_and, d, p.multiProcessorCount, mhz);
    run("v_min_u32", k_min, d, p.multiProcessorCount, mhz);
    run("v_cndmask_b32", k_cndm, d, p.multiProcessorCount, mhz);
    run("v_add_lshl_u32", k_addlsh, d, p.multiProcessorCount, mhz);
    run("v_add_u32_sdwa", k_sdwa, d, p.multiProcessorCount, mhz);
    run("v_lshlrev_b32", k_shl, d, p.multiProcessorCount, mhz);
    run("v_mbcnt_lo", k_mbcnt, d, p.multiProcessorCount, mhz);
    return 0;
}
