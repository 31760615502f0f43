// Micro-benchmark with HAND-PLACED instruction streams (inline asm, nothing for the compiler to reschedule): what a gfx950 SIMD
// issues beside v_mfma_f32_32x32x16_f16.
//   (a) one wave per SIMD: K independent v_fma_f32 (or v_pk_fma_f32) placed after EVERY MFMA, K = 0 .. 8
//   (b) two waves per SIMD, both running stream (a)
//   (c) two waves per SIMD: one MFMA-only, the other VALU-only (does the partner's VALU hide under this wave's MFMAs?)
// Workgroups of 512 threads (8 waves, wave w and w + 4 share a SIMD), one workgroup per CU, 256 workgroups.
// Output: SIMD cycles per MFMA slot from s_memtime deltas of wave 0 (shader clock), median over workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MF(ACC) "v_mfma_f32_32x32x16_f16 %" #ACC ", %8, %9, %" #ACC "\n"
#define F1 "v_fma_f32 %4, %4, %10, %11\n"
#define F2 F1 "v_fma_f32 %5, %5, %10, %11\n"
#define F3 F2 "v_fma_f32 %6, %6, %10, %11\n"
#define F4 F3 "v_fma_f32 %7, %7, %10, %11\n"
#define F5 F4 "v_fma_f32 %12, %12, %10, %11\n"
#define F6 F5 "v_fma_f32 %13, %13, %10, %11\n"
#define F8 F6 "v_fma_f32 %14, %14, %10, %11\n" "v_fma_f32 %15, %15, %10, %11\n"
#define P1 "v_pk_fma_f32 %16, %16, %18, %19\n"
#define P2 P1 "v_pk_fma_f32 %17, %17, %18, %19\n"
#define F0 ""

// one MFMA + filler, on 4 rotating accumulators, 16 MFMAs per asm block
#define BLOCK4(FILL) MF(0) FILL MF(1) FILL MF(2) FILL MF(3) FILL
#define BLOCK16(FILL) BLOCK4(FILL) BLOCK4(FILL) BLOCK4(FILL) BLOCK4(FILL)
#define VONLY16(FILL) FILL FILL FILL FILL FILL FILL FILL FILL FILL FILL FILL FILL FILL FILL FILL FILL

#define ASM(BODY)                                                                                                      \
    asm volatile(BODY : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)          \
                 : "v"(a), "v"(b), "v"(fa), "v"(fb), "v"(f4), "v"(f5), "v"(f6), "v"(f7), "v"(p0), "v"(p1), "v"(pa), "v"(pb))

template <int MODE, int K>   // MODE 0: all waves stream (a); 1: waves 0-3 stream (a), 4-7 idle; 2: waves 0-3 MFMA only, waves 4-7 VALU only (K per slot)
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int iters, float a0)
{
    const int wave = threadIdx.x >> 6;
    f32x16 acc0, acc1, acc2, acc3;
    for (int q = 0; q < 16; q++) { acc0[q] = 0.f; acc1[q] = 1.f; acc2[q] = 2.f; acc3[q] = 3.f; }
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(a0 + (threadIdx.x & 7) * 0.01f + i * 0.001f); b[i] = (_Float16)(0.5f + i * 0.01f); }
    float f0 = a0, f1 = a0 + 1, f2 = a0 + 2, f3 = a0 + 3, f4 = a0 + 4, f5 = a0 + 5, f6 = a0 + 6, f7 = a0 + 7, fa = 0.999f, fb = 0.001f;
    f32x2 p0 = {a0, a0 + 1}, p1 = {a0 + 2, a0 + 3}, pa = {0.999f, 0.999f}, pb = {0.001f, 0.001f};
    const bool mf = MODE == 0 || wave < 4, idle = MODE == 1 && wave >= 4, vonly = MODE == 2 && wave >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (!idle) {
        if (vonly) {
            for (int it = 0; it < iters; it++) {
                if (K == 2) ASM(VONLY16(F2)); else if (K == 4) ASM(VONLY16(F4)); else if (K == 5) ASM(VONLY16(F5)); else if (K == 6) ASM(VONLY16(F6));
                else if (K == 8) ASM(VONLY16(F8));
            }
        } else if (mf) {
            for (int it = 0; it < iters; it++) {
                if (MODE == 2 || K == 0) ASM(BLOCK16(F0));
                else if (K == 1) ASM(BLOCK16(F1)); else if (K == 2) ASM(BLOCK16(F2)); else if (K == 3) ASM(BLOCK16(F3));
                else if (K == 4) ASM(BLOCK16(F4)); else if (K == 5) ASM(BLOCK16(F5)); else if (K == 6) ASM(BLOCK16(F6));
                else if (K == 8) ASM(BLOCK16(F8)); else if (K == 101) ASM(BLOCK16(P1)); else if (K == 102) ASM(BLOCK16(P2));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (threadIdx.x == 256) cyc[256 + blockIdx.x] = t1 - t0;
    float s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p0.y + p1.x + p1.y;
    for (int q = 0; q < 16; q++) s += acc0[q] + acc1[q] + acc2[q] + acc3[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int K>
void run(const char* name)
{
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 256 * 512 * sizeof(float)); (void)hipMalloc(&c, 512 * sizeof(unsigned long long));
    const int iters = 4000;
    hipLaunchKernelGGL((probe<MODE, K>), dim3(256), dim3(512), 0, 0, d, c, 200, 1.f);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, K>), dim3(256), dim3(512), 0, 0, d, c, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    (void)hipMemcpy(h.data(), c, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + 256); std::sort(h.begin() + 256, h.end());
    // s_memtime counts at 100 MHz on gfx9 (REFCLK): convert with the wall time of the kernel instead -- cycles = ms * clock; report both
    printf("%-78s wall %.3f ms  | per MFMA slot: %6.1f ns (wave 0) %6.1f ns (wave 4) | memtime ticks/slot %.2f / %.2f\n", name, ms,
           ms * 1e6 / (iters * 16.0), ms * 1e6 / (iters * 16.0), (double)h[128] / (iters * 16.0), (double)h[384] / (iters * 16.0));
    (void)hipFree(d); (void)hipFree(c);
}

int main()
{
    run<1, 0>("1 wave/SIMD: MFMA only");
    run<1, 1>("1 wave/SIMD: MFMA + 1 v_fma per gap");
    run<1, 2>("1 wave/SIMD: MFMA + 2 v_fma per gap");
    run<1, 3>("1 wave/SIMD: MFMA + 3 v_fma per gap");
    run<1, 4>("1 wave/SIMD: MFMA + 4 v_fma per gap");
    run<1, 5>("1 wave/SIMD: MFMA + 5 v_fma per gap");
    run<1, 6>("1 wave/SIMD: MFMA + 6 v_fma per gap");
    run<1, 8>("1 wave/SIMD: MFMA + 8 v_fma per gap");
    run<1, 101>("1 wave/SIMD: MFMA + 1 v_pk_fma_f32 per gap");
    run<1, 102>("1 wave/SIMD: MFMA + 2 v_pk_fma_f32 per gap");
    run<0, 0>("2 waves/SIMD: MFMA only (both)");
    run<0, 2>("2 waves/SIMD: both MFMA + 2 v_fma per gap");
    run<0, 4>("2 waves/SIMD: both MFMA + 4 v_fma per gap");
    run<0, 102>("2 waves/SIMD: both MFMA + 2 v_pk_fma per gap");
    run<2, 2>("2 waves/SIMD: wave A MFMA only, wave B 2 v_fma per slot (VALU only)");
    run<2, 4>("2 waves/SIMD: wave A MFMA only, wave B 4 v_fma per slot (VALU only)");
    run<2, 6>("2 waves/SIMD: wave A MFMA only, wave B 6 v_fma per slot (VALU only)");
    run<2, 8>("2 waves/SIMD: wave A MFMA only, wave B 8 v_fma per slot (VALU only)");
    return 0;
}
