// Micro-benchmark: sustained issue rate of the VALU instructions the byte / i16 kernels of the front end are made of, at 1, 2, 4 and 8
// waves per SIMD (gfx950).  Each wave runs a block of 64 INDEPENDENT instances of one opcode (hand-placed inline asm) in a loop;
// result = SIMD cycles per wave-instruction = wall time x clock / (instructions per SIMD), clock measured with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)

#define KERNEL(NAME, INSTR)                                                                                         \
__global__ void NAME(unsigned* out, unsigned long long* cyc, int iters, unsigned a0)                                \
{                                                                                                                   \
    unsigned r0 = a0 + threadIdx.x, r1 = a0 * 3 + 1, r2 = a0 ^ 0x55aa, r3 = a0 + 7, x = threadIdx.x * 0x01010101u, y = 0x04030201u; \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
    for (int it = 0; it < iters; it++)                                                                               \
        asm volatile(REP16(INSTR) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(x), "v"(y));                        \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3;                                                  \
}
// four independent destinations per group of four instructions
KERNEL(k_add_u32,   "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
KERNEL(k_fma_f32,   "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
KERNEL(k_pk_sub,    "v_pk_sub_i16 %0, %0, %4\n v_pk_sub_i16 %1, %1, %4\n v_pk_sub_i16 %2, %2, %4\n v_pk_sub_i16 %3, %3, %4\n")
KERNEL(k_pk_min,    "v_pk_min_i16 %0, %0, %4\n v_pk_min_i16 %1, %1, %4\n v_pk_min_i16 %2, %2, %4\n v_pk_min_i16 %3, %3, %4\n")
KERNEL(k_perm,      "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %4, 1\n v_alignbyte_b32 %1, %1, %4, 2\n v_alignbyte_b32 %2, %2, %4, 3\n v_alignbyte_b32 %3, %3, %4, 1\n")
KERNEL(k_dot4,      "v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3\n")
KERNEL(k_min3,      "v_min3_u32 %0, %0, %4, %5\n v_min3_u32 %1, %1, %4, %5\n v_min3_u32 %2, %2, %4, %5\n v_min3_u32 %3, %3, %4, %5\n")
KERNEL(k_sad_u8,    "v_sad_u8 %0, %4, %5, %0\n v_sad_u8 %1, %4, %5, %1\n v_sad_u8 %2, %4, %5, %2\n v_sad_u8 %3, %4, %5, %3\n")
KERNEL(k_and_or,    "v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5\n")
KERNEL(k_cmp,       "v_cmp_gt_u32 vcc, %0, %4\n v_cmp_gt_u32 vcc, %1, %4\n v_cmp_gt_u32 vcc, %2, %4\n v_cmp_gt_u32 vcc, %3, %4\n")
KERNEL(k_pk_sub_u8sdwa, "v_sub_u16_sdwa %0, %0, %4 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_2\n v_sub_u16_sdwa %1, %1, %4 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_2\n v_sub_u16_sdwa %2, %2, %4 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_2\n v_sub_u16_sdwa %3, %3, %4 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_2\n")

template <class K>
void run(const char* name, K kern)
{
    unsigned* d; unsigned long long* c;
    (void)hipMalloc(&d, 256 * 1024 * 4 * sizeof(unsigned)); (void)hipMalloc(&c, 1024 * sizeof(unsigned long long));
    const int iters = 20000;
    printf("%-18s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {            // waves per SIMD: blocks of 256 * wps threads, one block per CU
        const int threads = 256 * (wps > 4 ? 4 : wps), blocks = 256 * (wps > 4 ? 2 : 1);      // 8 w/SIMD: two 1024-thread blocks per CU
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, c, 50, 3u);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, c, iters, 3u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), c, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        // wave 0's loop took h ticks (shader cycles) for iters * 64 instructions, while its SIMD hosted wps waves doing the same
        // wall clock: every SIMD of the chip executed wps * iters * 64 wave-instructions in ms
        printf("  %d w/SIMD: %5.2f ticks %5.3f ns", wps, (double)h[blocks / 2] / (iters * 64.0) / wps, ms * 1e6 / ((double)iters * 64.0 * wps));
    }
    printf("\n");
    (void)hipFree(d); (void)hipFree(c);
}

int main()
{
    run("v_add_u32", k_add_u32); run("v_fma_f32", k_fma_f32); run("v_pk_sub_i16", k_pk_sub); run("v_pk_min_i16", k_pk_min);
    run("v_perm_b32", k_perm); run("v_alignbyte_b32", k_alignbyte); run("v_dot4_u32_u8", k_dot4); run("v_min3_u32", k_min3);
    run("v_sad_u8", k_sad_u8); run("v_and_or_b32", k_and_or); run("v_cmp_gt_u32", k_cmp); run("v_sub_u16_sdwa", k_pk_sub_u8sdwa);
    return 0;
}
