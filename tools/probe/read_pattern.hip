// Micro-benchmark: HBM read bandwidth of the access pattern of k_fcn_dwpw (hidden tensor [B][960][64][64] f32, one workgroup = two
// image rows y, y+4 of one image, per 16-channel chunk the four tap rows y-4, y, y+4, y+8 as 256-B pieces 16 KB apart) against
// (a) a plain streaming read of the same bytes and (b) the same workgroup tiling on a chunk-major layout
// [B][60][64 rows][16 ch][64 px] where a chunk's row is one 4-KB piece.  No arithmetic beyond keeping the loads alive.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int K = 960, HW = 4096, Wd = 64, DIL = 4;

__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ p, size_t n, unsigned* out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// LAYOUT 0: planes [b][c][y][x]; 1: chunk-major [b][c/16][y][c%16][x].  PF = chunks in flight.  ROWS = tap rows loaded per chunk
// (4 = the kernel's pattern; 2 = only the workgroup's own two rows: the algorithmic minimum)
template <int LAYOUT, int PF, int ROWS>
__global__ __launch_bounds__(256, 2) void k_tiles(const float* __restrict__ X, unsigned* out)
{
    const int tid = threadIdx.x, kc = tid >> 4, g = tid & 15;
    const int nwg = gridDim.x, L = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    const int b = L / 32, p128 = L % 32;
    const int yA = (p128 / DIL) * 2 * DIL + p128 % DIL;
    const bool halfB = g >= 8;
    const int x0 = (g & 7) * 8;
    int rowY[2];
    if (ROWS == 4) { rowY[0] = halfB ? yA + 2 * DIL : yA - DIL; rowY[1] = halfB ? yA + DIL : yA; }
    else { rowY[0] = halfB ? yA + DIL : yA; rowY[1] = rowY[0]; }
    for (int r = 0; r < 2; r++) rowY[r] = min(max(rowY[r], 0), Wd - 1);
    const float* Xb = X + (size_t)b * K * HW;
    float4 buf[PF][2][2];
    auto issue = [&](int slot, int c) {
        c = min(c, K / 16 - 1);
#pragma unroll
        for (int r = 0; r < (ROWS == 4 ? 2 : 1); r++) {
            const float* P = LAYOUT == 0 ? Xb + (size_t)(16 * c + kc) * HW + rowY[r] * Wd + x0
                                         : Xb + (size_t)c * 16 * HW + (size_t)rowY[r] * 16 * Wd + kc * Wd + x0;
            buf[slot][r][0] = *(const float4*)P; buf[slot][r][1] = *(const float4*)(P + 4);
        }
    };
#pragma unroll
    for (int d = 0; d < PF; d++) issue(d, d);
    float acc = 0.f;
    for (int c = 0; c < K / 16; c += PF) {
#pragma unroll
        for (int d = 0; d < PF; d++) {
#pragma unroll
            for (int r = 0; r < (ROWS == 4 ? 2 : 1); r++) acc += buf[d][r][0].x + buf[d][r][0].w + buf[d][r][1].y + buf[d][r][1].z;
            issue(d, c + d + PF);
        }
    }
    if (acc == 12345.678f) out[0] = 1;
}

template <typename F> void timeit(const char* name, double bytes, F launch)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-72s %8.1f us  %.2f TB/s (useful bytes)\n", name, ms * 1e3, bytes / ms * 1e-9);
}

int main()
{
    const int B = 128;
    const size_t n = (size_t)B * K * HW;            // floats: 2.01 GB
    float* d; unsigned* o;
    if (hipMalloc(&d, n * 4) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 0, n * 4);
    const double bytes = (double)n * 4;
    timeit("streaming uint4 read, 4096 workgroups", bytes, [&] { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const uint4*)d, n / 4, o); });
    timeit("streaming uint4 read, 16384 workgroups", bytes, [&] { hipLaunchKernelGGL(k_stream, dim3(16384), dim3(256), 0, 0, (const uint4*)d, n / 4, o); });
#define T(LAY, PF, ROWS, NAME) timeit(NAME, bytes, [&] { hipLaunchKernelGGL((k_tiles<LAY, PF, ROWS>), dim3(32 * B), dim3(256), 0, 0, d, o); })
    T(0, 2, 4, "planes, 4 tap rows, 2 chunks in flight (= k_fcn_dwpw)");
    T(0, 4, 4, "planes, 4 tap rows, 4 chunks in flight");
    T(0, 2, 2, "planes, own 2 rows only, 2 chunks in flight");
    T(0, 4, 2, "planes, own 2 rows only, 4 chunks in flight");
    T(1, 2, 4, "chunk-major, 4 tap rows, 2 chunks in flight");
    T(1, 4, 4, "chunk-major, 4 tap rows, 4 chunks in flight");
    T(1, 2, 2, "chunk-major, own 2 rows only, 2 chunks in flight");
    T(1, 4, 2, "chunk-major, own 2 rows only, 4 chunks in flight");
    return 0;
}
