// Micro-benchmark: HBM write bandwidth on MI355X as a function of the contiguous run each wave store produces and
// of the stride between consecutive stores of one workgroup (the FCN's [B][C][HW] planes are written as C short runs).
#include <hip/hip_runtime.h>
#include <cstdio>

// grid: (runs per plane, images); each WG (256 thr) writes, for c in 0..C-1, a run of RUN bytes in plane c.
// RUN = 512 B: 4 waves write 4 different planes (32 lanes x 16 B, 2 rows per wave-store like the MFMA epilogue)
template <int RUNF4>   // float4s per run handled by one wave-store half (32 lanes) -> run bytes = 512
__global__ __launch_bounds__(256) void k_planes(float4* Y, int C, int planeF4, int wgRunsPerPlane)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    const size_t img = (size_t)blockIdx.y * C * planeF4;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    // tiles of 32 channels per wave: wave w takes tiles w, w+4, ...; each store covers rows (r, r+4) x 512 B
    for (int t = wave; t < C / 32; t += 4)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            Y[img + (size_t)c * planeF4 + blockIdx.x * 32 + col] = v;
        }
}
// same bytes, but every workgroup writes whole consecutive 2 KB pieces: WG = (plane group, 2 KB slice)
__global__ __launch_bounds__(256) void k_linear(float4* Y, size_t nF4)
{
    size_t i = (size_t)blockIdx.x * 256 * 16 + threadIdx.x;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
    for (int k = 0; k < 16; k++) { if (i < nF4) Y[i] = v; i += 256; }
}

int main()
{
    const int B = 32;
    float4* d; (void)hipMalloc(&d, (size_t)1 << 30);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { int C, HW; } cases[] = {{960, 4096}, {576, 4096}, {384, 4096}, {96, 65536}, {144, 16384}};
    for (auto cs : cases) {
        const int planeF4 = cs.HW / 4;
        const size_t bytes = (size_t)B * cs.C * cs.HW * 4;
        float ms;
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_planes<32>, dim3(planeF4 / 32, B), dim3(256), 0, 0, d, cs.C, planeF4, planeF4 / 32);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("planes C=%4d HW=%6d (%4.0f MB): 512-B runs per plane  %.1f us  %.2f TB/s", cs.C, cs.HW, bytes / 1e6, ms * 1e3, bytes / ms * 1e-9);
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_linear, dim3((unsigned)((bytes / 16 + 4095) / 4096)), dim3(256), 0, 0, d, bytes / 16);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("   | linear %.1f us  %.2f TB/s\n", ms * 1e3, bytes / ms * 1e-9);
    }
    return 0;
}
