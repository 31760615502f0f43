// Micro-benchmark: do f16 MFMAs and plain VALU work overlap on a gfx950 SIMD
//   (a) inside one wave (independent instructions interleaved), and
//   (b) across the two waves that share a SIMD (wave w: MFMA only, wave w+4: VALU only)?
// Workgroups of 512 threads = 8 waves = 2 per SIMD, one workgroup per CU.  Times are per loop iteration per wave in SIMD cycles
// (GRBM clock measured under MFMA load: ~1.95 GHz; printed at 2.0 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// role of a wave: 0 = idle, 1 = NM MFMAs per iteration, 2 = NV FMAs per iteration, 3 = both interleaved
template <int NM, int NV, int ROLE_LO, int ROLE_HI>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float a0, float b0)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_LO : ROLE_HI;
    f32x16 acc[4];
    for (int t = 0; t < 4; t++) for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(a0 + (threadIdx.x & 7) + i); b[i] = (_Float16)(b0 + i); }
    float fa = a0 + (threadIdx.x & 3), fb = b0;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = a0 * i;
    if (role == 1) {
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int t = 0; t < NM; t++) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t & 3], 0, 0, 0);
    } else if (role == 2) {
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int j = 0; j < NV; j++) v[j & 7] = __builtin_fmaf(v[j & 7], fa, fb);
    } else if (role == 3) {
        constexpr int PER = NM ? NV / (NM ? NM : 1) : 0;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int t = 0; t < NM; t++) {
                acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t & 3], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < PER; j++) v[j & 7] = __builtin_fmaf(v[j & 7], fa, fb);
            }
    }
    float s = 0.f;
    for (int t = 0; t < 4; t++) for (int q = 0; q < 16; q++) s += acc[t][q];
    for (int i = 0; i < 8; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int RL, int RH>
void run(const char* name)
{
    float* d; (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NM, NV, RL, RH>), dim3(256), dim3(512), 0, 0, d, 100, 1.f, 2.f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NM, NV, RL, RH>), dim3(256), dim3(512), 0, 0, d, iters, 1.f, 2.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %.3f ms = %6.0f cycles/iter @2.0GHz   (MFMA %d x 32 = %d, VALU %d x 4 = %d)\n", name, ms, ms * 1e-3 * 2.0e9 / iters,
           NM, NM * 32, NV, NV * 4);
    (void)hipFree(d);
}

int main()
{
    run<16, 0, 1, 0>("waves 0-3: 16 MFMA; waves 4-7 idle");
    run<16, 0, 1, 1>("waves 0-3: 16 MFMA; waves 4-7: 16 MFMA");
    run<0, 128, 0, 2>("waves 0-3 idle; waves 4-7: 128 FMA");
    run<0, 128, 2, 2>("waves 0-3: 128 FMA; waves 4-7: 128 FMA");
    run<16, 128, 1, 2>("waves 0-3: 16 MFMA; waves 4-7: 128 FMA  (cross-wave overlap?)");
    run<16, 64, 1, 2>("waves 0-3: 16 MFMA; waves 4-7: 64 FMA");
    run<16, 128, 3, 0>("waves 0-3: 16 MFMA + 128 FMA interleaved 1:8; waves 4-7 idle");
    run<16, 64, 3, 0>("waves 0-3: 16 MFMA + 64 FMA interleaved 1:4; waves 4-7 idle");
    run<16, 128, 3, 3>("all 8 waves: 16 MFMA + 128 FMA interleaved 1:8");
    run<16, 64, 3, 3>("all 8 waves: 16 MFMA + 64 FMA interleaved 1:4");
    return 0;
}
