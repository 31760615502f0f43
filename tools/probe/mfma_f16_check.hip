// Checks on gfx950: operand layout of v_mfma_f32_32x32x16_f16 and whether f16 subnormal inputs are preserved.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void k(const _Float16* A, const _Float16* B, float* C)
{
    const int lane = threadIdx.x, i = lane & 31, kg = lane >> 5;
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = A[i * 16 + 8 * kg + j]; b[j] = B[(8 * kg + j) * 32 + i]; }
    f32x16 acc;
    for (int q = 0; q < 16; q++) acc[q] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int q = 0; q < 16; q++) C[((q & 3) + 8 * (q >> 2) + 4 * kg) * 32 + i] = acc[q];
}

int main()
{
    std::vector<_Float16> A(32 * 16), B(16 * 32);
    std::vector<float> C(32 * 32), R(32 * 32);
    _Float16 *dA, *dB; float* dC;
    (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, C.size() * 4);
    for (int pass = 0; pass < 2; pass++) {
        unsigned s = 12345;
        for (auto& v : A) { s = s * 1664525u + 1013904223u; v = pass == 0 ? (_Float16)(float)((int)(s >> 24) % 17 - 8) : (_Float16)ldexpf((float)((s >> 24) % 15 + 1), -24); }
        for (auto& v : B) { s = s * 1664525u + 1013904223u; v = pass == 0 ? (_Float16)(float)((int)(s >> 24) % 13 - 6) : (_Float16)(float)((s >> 24) % 7 + 1) * (_Float16)256.f; }
        (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int m = 0; m < 32; m++)
            for (int n = 0; n < 32; n++) {
                double r = 0;
                for (int kk = 0; kk < 16; kk++) r += (double)(float)A[m * 16 + kk] * (double)(float)B[kk * 32 + n];
                maxerr = fmax(maxerr, fabs(r - C[m * 32 + n])); maxref = fmax(maxref, fabs(r));
            }
        printf("%s: max |C - ref| = %.3g (max |ref| = %.3g)\n", pass == 0 ? "layout check (small integers)" : "subnormal f16 A x 256..1792 B", maxerr, maxref);
    }
    return 0;
}
