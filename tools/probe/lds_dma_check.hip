// Checks on gfx950: global_load_lds with 16 B per lane lands lane i's bytes at lds_base + 16*i, and later plain loads
// whose data has arrived imply the earlier LDS-DMA has completed (in-order vmcnt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void k(const uint4* src, uint4* dst, const float* other, float* sink)
{
    __shared__ uint4 buf[4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + wave * 64 + lane), (lptr_t)(buf + wave * 64), 16, 0, 0);
    float v = other[threadIdx.x];          // a later ordinary load
    sink[threadIdx.x] = v * 2.f;           // its use makes the compiler wait for it (and, in order, for the DMA)
    __syncthreads();
    dst[threadIdx.x] = buf[(threadIdx.x + 64) % 256];     // read another wave's part
}

int main()
{
    std::vector<uint4> h(256), o(256);
    for (int i = 0; i < 256; i++) h[i] = make_uint4(i, i * 3 + 1, i * 7 + 2, ~i);
    uint4 *ds, *dd; float *dof, *dsk;
    (void)hipMalloc(&ds, 4096); (void)hipMalloc(&dd, 4096); (void)hipMalloc(&dof, 1024); (void)hipMalloc(&dsk, 1024);
    (void)hipMemset(dof, 0, 1024);
    (void)hipMemcpy(ds, h.data(), 4096, hipMemcpyHostToDevice);
    int bad = 0;
    for (int rep = 0; rep < 200; rep++) {
        (void)hipMemset(dd, 0, 4096);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, ds, dd, dof, dsk);
        (void)hipMemcpy(o.data(), dd, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 256; i++) { const uint4 e = h[(i + 64) % 256]; if (o[i].x != e.x || o[i].y != e.y || o[i].z != e.z || o[i].w != e.w) bad++; }
    }
    printf("lds dma 16B/lane: %d mismatches over 200 launches\n", bad);
    return bad != 0;
}
