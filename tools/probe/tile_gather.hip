// Micro-benchmark (r04): what bounds the input gather / residual load of the whole-block FCN kernels (k_fcn_irbd4: one 512-thread
// workgroup per CU; every lane reads 80 values of its tile [160 channels][256 sub-image pixels], all requests in flight, then waits)?
// Variants of WHERE those 80 values sit, same bytes per workgroup (164 KB useful):
//   0  planes [c][64][64], sub-image pixels 4 apart (the kernel today): lane = dword, 16 lanes span 256 B, 4 channel planes per instruction
//   1  planes [c][16 sub-images][256]: lane = dword, 16 lanes = 64 contiguous bytes, 4 channel planes (16 KB apart) per instruction
//   2  tile-major [16 sub-images][160 c][256]: the workgroup's tile is ONE contiguous 164 KB piece; lane = dword
//   3  tile-major, lane = dwordx4 (20 loads per lane)
//   4  planes [c][64][64], lane = dwordx4 of 4 ADJACENT pixels (a quarter of them belongs to the tile: 4 x the bytes; what a layout-free
//      "read whole rows" gather costs)
// Prints cycles (s_memtime) from first issue to last landing, per workgroup (wave 0), for a footprint far beyond the Infinity Cache
// (images = 128) and for an L2-resident one (images = 2), with every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int C = 160, HW = 4096;
__device__ unsigned long long g_cyc[2];

template <int V>
__global__ __launch_bounds__(512) void k_gather(const float* __restrict__ X, int nImg, float* __restrict__ sink)
{
    extern __shared__ float lds[];                      // 100 KB: one workgroup per CU, like the kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = gridDim.x, L = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    const int b = L >> 4, py = (L >> 2) & 3, px = L & 3, si = L & 15;
    const float* Xb = X + (size_t)b * C * HW;
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (V == 0 || V == 1 || V == 2) {
        float v[2][5][8];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int c = 32 * s5 + 8 * (lane >> 4) + j, sr = 2 * wave + u, sc = lane & 15;
                    size_t a;
                    if (V == 0) a = (size_t)c * HW + (4 * sr + py) * 64 + 4 * sc + px;
                    else if (V == 1) a = (size_t)c * HW + si * 256 + sr * 16 + sc;
                    else a = ((size_t)si * C + c) * 256 + sr * 16 + sc;
                    v[u][s5][j] = Xb[a];
                }
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc += v[u][s5][j];
    } else if (V == 3) {
        float4 v[20];
#pragma unroll
        for (int k = 0; k < 20; k++) v[k] = *(const float4*)(Xb + ((size_t)si * C * 256) + ((size_t)k * 512 + tid) * 4);
#pragma unroll
        for (int k = 0; k < 20; k++) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    } else {
        float4 v[2][5][8];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int c = 32 * s5 + 8 * (lane >> 4) + j, sr = 2 * wave + u, sc = lane & 15;
                    v[u][s5][j] = *(const float4*)(Xb + (size_t)c * HW + (4 * sr + py) * 64 + 4 * sc);
                }
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
                for (int j = 0; j < 8; j++) acc += v[u][s5][j].x + v[u][s5][j].w;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 123.456f) sink[tid] = acc + lds[tid];
    if (tid == 0) { atomicAdd(&g_cyc[0], t1 - t0); atomicAdd(&g_cyc[1], 1ull); }
}

template <int V>
void run(const float* dX, int nImg, float* dSink, const char* what)
{
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gather<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    unsigned long long z[2] = {0, 0};
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof z);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather<V>, dim3(16 * nImg), dim3(512), 100 * 1024, 0, dX, nImg, dSink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cyc), sizeof c);
        if (rep == 2)
            printf("  variant %d (%s): %8.0f cycles per workgroup, launch %.3f ms = %.2f TB/s useful\n", V, what, (double)c[0] / c[1], ms,
                   (double)nImg * C * HW * 4 / (ms * 1e-3) / 1e12);
    }
}

int main()
{
    for (int nImg : {128, 16, 2}) {
        float *dX, *dSink;
        hipMalloc(&dX, (size_t)nImg * C * HW * 4 + 4096); hipMalloc(&dSink, 4096);
        hipMemset(dX, 0, (size_t)nImg * C * HW * 4 + 4096);
        printf("images %d (%.0f MB, %d workgroups):\n", nImg, nImg * C * HW * 4 / 1e6, 16 * nImg);
        run<0>(dX, nImg, dSink, "planes, pixels 4 apart, dword");
        run<1>(dX, nImg, dSink, "planes, sub-image-major pixels, dword");
        run<2>(dX, nImg, dSink, "tile-major, dword");
        run<3>(dX, nImg, dSink, "tile-major, dwordx4");
        run<4>(dX, nImg, dSink, "planes, dwordx4 of adjacent pixels (4x bytes)");
        hipFree(dX); hipFree(dSink);
    }
    return 0;
}
