// Feasibility probe for a one-wave-per-SIMD form of k_fcn_irbd4: inline-asm MFMAs whose accumulators AND B operands live in AGPRs
// (gfx90a+: srcA / srcB / srcC / vDst may each be an AGPR), 256-thread workgroup, > 256 registers per wave.
//   hipcc --offload-arch=gfx950 -O3 -o agpr_mfma agpr_mfma.hip && ./agpr_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// C[16x16] += A[16x32] * B[32x16]; acc in AGPR ("+a"), B in AGPR ("a"), A in VGPR ("v")
__device__ __forceinline__ void mfma16_aB(f4& acc, h8 a, h8 bAg)
{ asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(bAg)); }
__device__ __forceinline__ void mfma32_acc(f16v& acc, h8 a, h8 b)
{ asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }

__global__ __launch_bounds__(256, 1) void k(const h8* A, const h8* B, float* C, float* C2, int iters)
{
    const int lane = threadIdx.x & 63;
    h8 a = A[lane];
    h8 b[20];                       // 80 registers of B operands: ask for AGPRs through the asm constraint
#pragma unroll
    for (int i = 0; i < 20; i++) b[i] = B[i * 64 + lane];
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f16v big[10];                   // 160 accumulator registers
#pragma unroll
    for (int t = 0; t < 10; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) big[t][q] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 20; i++) mfma16_aB(acc[i & 3], a, b[i]);
#pragma unroll
        for (int t = 0; t < 10; t++) mfma32_acc(big[t], a, b[t]);
    }
    f4 s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int r = 0; r < 4; r++) C[(blockIdx.x * 256 + threadIdx.x) * 4 + r] = s[r];
    float z = 0;
#pragma unroll
    for (int t = 0; t < 10; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) z += big[t][q];
    C2[blockIdx.x * 256 + threadIdx.x] = z;
}

int main()
{
    std::vector<_Float16> hA(64 * 8), hB(20 * 64 * 8);
    for (size_t i = 0; i < hA.size(); i++) hA[i] = (_Float16)((int)(i % 7) - 3);
    for (size_t i = 0; i < hB.size(); i++) hB[i] = (_Float16)((int)(i % 5) - 2);
    h8 *dA, *dB; float *dC, *dC2;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, 256 * 256 * 4 * 4); hipMalloc(&dC2, 256 * 256 * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, dA, dB, dC, dC2, 1);
    hipDeviceSynchronize();
    // reference for lane 0..63 of wave 0: C[16x16] of sum_i A * B_i  (layout: lane: col = lane & 15, rows 4 (lane >> 4) + r; A[row l&15][k = 8 (l>>4) + j])
    std::vector<float> hC(256 * 4);
    hipMemcpy(hC.data(), dC, 256 * 4 * 4, hipMemcpyDeviceToHost);
    double maxd = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int row = 4 * (l >> 4) + r, col = l & 15;
            double ref = 0;
            for (int i = 0; i < 20; i++)
                for (int kk = 0; kk < 32; kk++) {
                    const float av = (float)hA[(size_t)((kk >> 3) * 16 + row) * 8 + (kk & 7)];
                    const float bv = (float)hB[((size_t)i * 64 + (kk >> 3) * 16 + col) * 8 + (kk & 7)];
                    ref += av * bv;
                }
            maxd = fmax(maxd, fabs(ref - hC[l * 4 + r]));
        }
    printf("max |diff| of the AGPR-operand 16x16x32 chain: %g\n", maxd);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, dA, dB, dC, dC2, 2000);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, dA, dB, dC, dC2, 2000);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = 2000.0 * (20 * 16 + 10 * 32);
    printf("2000 iterations of 20 x 16x16x32 + 10 x 32x32x16 per wave: %.3f ms (pure matrix-pipe time at 2.1 GHz: %.3f ms)\n", ms, cyc / 2.1e6);
    return maxd == 0 ? 0 : 1;
}
