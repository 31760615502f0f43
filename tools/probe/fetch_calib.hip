// FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern").  Three streaming readers of the same 1 GiB buffer (far beyond the 256 MiB Infinity
// Cache): 4, 8 and 16 bytes per lane per load.  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`; the ratio
// FETCH_SIZE * 1024 / 2^30 per kernel is the correction factor to apply to kernels with that load width.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __global__ void k_read(const T* __restrict__ p, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const unsigned* w = (const unsigned*)&v;
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345678u) out[0] = acc;       // keeps the loads alive
}
int main()
{
    const size_t bytes = (size_t)1 << 30;
    void* d = nullptr; unsigned* o = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    hipMemset(d, 1, bytes);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read<unsigned>, dim3(4096), dim3(256), 0, 0, (const unsigned*)d, bytes / 4, o);
        hipLaunchKernelGGL(k_read<uint2>, dim3(4096), dim3(256), 0, 0, (const uint2*)d, bytes / 8, o);
        hipLaunchKernelGGL(k_read<uint4>, dim3(4096), dim3(256), 0, 0, (const uint4*)d, bytes / 16, o);
    }
    hipDeviceSynchronize();
    printf("read 1 GiB x 3 widths x 3 reps\n");
    return 0;
}
