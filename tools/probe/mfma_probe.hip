// Micro-benchmark: f32 MFMA issue rate on gfx950 and how much plain VALU / LDS work overlaps with it.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NACC, int NVALU, int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; t++) for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = a0 * i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < NACC; t++) {
            if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NVALU; j++) v[j % 8] = __builtin_fmaf(v[j % 8], a, b);
        }
    }
    float s = 0.f;
    for (int t = 0; t < NACC; t++) for (int q = 0; q < 16; q++) s += acc[t][q];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int NVALU, int MODE>
void run(const char* name, int wgPerCu)
{
    float* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256 * wgPerCu);
    hipLaunchKernelGGL((probe<NACC, NVALU, MODE>), grid, dim3(256), 0, 0, d, 100, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, NVALU, MODE>), grid, dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = MODE == 0 ? (double)grid.x * 4 * iters * NACC * 4096.0 : 0;
    const double vf = (double)grid.x * 256 * iters * NACC * NVALU * 2.0;
    printf("%-34s wg/cu=%d  %.3f ms  mfma %.1f TF/s  valu %.1f TF/s  cycles/iter/wave@2.4GHz %.0f\n", name, wgPerCu, ms,
           mf / ms * 1e-9, vf / ms * 1e-9, ms * 1e-3 * 2.4e9 / iters);
    hipFree(d);
}

int main()
{
    run<4, 0, 0>("mfma only, 4 acc", 1);
    run<4, 0, 0>("mfma only, 4 acc", 2);
    run<1, 0, 0>("mfma only, 1 acc (dependent)", 1);
    run<1, 0, 0>("mfma only, 1 acc (dependent)", 2);
    run<4, 4, 0>("mfma + 4 fma each", 1);
    run<4, 4, 0>("mfma + 4 fma each", 2);
    run<4, 8, 0>("mfma + 8 fma each", 1);
    run<4, 8, 0>("mfma + 8 fma each", 2);
    run<4, 14, 0>("mfma + 14 fma each", 1);
    run<4, 14, 0>("mfma + 14 fma each", 2);
    run<4, 8, 1>("8 fma only", 1);
    run<4, 8, 1>("8 fma only", 2);
    return 0;
}
