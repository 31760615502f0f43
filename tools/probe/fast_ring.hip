// Micro-benchmark (r04, verdict item 3b): the measured ceiling of the FAST kernel's tile scheme, stripped to its parts.
// Same tile as k_fast_nms (128 x 32 output pixels, (128 + 8) x (32 + 8) raw bytes staged in LDS by aligned dword loads), same packed-i16
// ring arithmetic (the device functions of iv_slam_amd/csrc/ivf_kernels.hip are included, not copied), on 256 planes of 1242 x 1160
// bytes (= the 1,441,432 pixels of an 8-level 1242 x 375 pyramid, one launch = what k_fast_nms sees per 128 stereo pairs):
//   mode 0  staging only: global -> LDS, barrier, every thread reads its dwords back once (no test)
//   mode 1  + the compass pre-check (rows -3, 0, +3) on EVERY pixel pair, result OR-ed into a register: pass A without the cell
//           classification, the score-plane zeroing, the ballots and the queue
//   mode 2  + the full 16-pixel ring score on every pair that passes (in place, divergent: no queue, no compaction)
//   mode 3  the full ring score on EVERY pair (what a kernel without a pre-check would pay)
// Prints us per launch, pixels per ns, and the launch's HBM-roofline fraction for the algorithmic bytes (1 B per pixel read).
#include "../../iv_slam_amd/csrc/ivf_kernels.hip"
#include <cstdio>
#include <vector>
using namespace ivf;
void ivf::count_launch() {}                              // the launchers of the included file count their launches through the C-ABI object
int ivf::set_error(int code, const char*, ...) { return code; }
constexpr int PW = 1242, PH = 1160, PP = 1280;          // plane width, height, pitch

template <int MODE>
__global__ __launch_bounds__(256) void k_ring(const uint8_t* __restrict__ planes, int nImg, int t, unsigned* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) unsigned raw[(kFastTH + 8) * (kRawP / 4) + 4];
    const int tilesX = (PW - 32 + kFastTW - 1) / kFastTW, tilesY = (PH - 32 + kFastTH - 1) / kFastTH;
    int img, bx;
    if (!xcd_tile_image(tilesX * tilesY, nImg, bx, img)) return;
    const int tid = threadIdx.x, tx = bx % tilesX, ty = bx / tilesX;
    const int x0 = 16 + tx * kFastTW, y0 = 16 + ty * kFastTH;
    const uint8_t* src = planes + (size_t)img * PP * PH;
    constexpr int kRawQ = kRawP / 4, kRawRPP = 256 / kRawQ, kRawIt = (kFastTH + 8 + kRawRPP - 1) / kRawRPP;
    const int rawQ = tid % kRawQ, rawR = tid / kRawQ;
    unsigned rv_[kRawIt];
#pragma unroll
    for (int k = 0; k < kRawIt; k++) {
        const int ry = rawR + kRawRPP * k, gy = min(y0 - 4 + ry, PH - 1), gx = min(x0 - 4 + 4 * rawQ, PP - 4);
        rv_[k] = *(const unsigned*)(src + (size_t)gy * PP + gx);
    }
#pragma unroll
    for (int k = 0; k < kRawIt; k++)
        if (rawR < kRawRPP && rawR + kRawRPP * k < kFastTH + 8) raw[(rawR + kRawRPP * k) * kRawQ + rawQ] = rv_[k];
    __syncthreads();
    unsigned acc = 0;
    constexpr int kQuads = (kScW + 3) / 4;
    for (int i = tid; i < kScH * kQuads; i += 256) {
        const int sy = i / kQuads, sx = (i % kQuads) * 4;
        const unsigned* base = raw + (sy * kRawP + sx) / 4;
        if (MODE == 0) { acc ^= base[3 * (kRawP / 4)]; continue; }
        unsigned t0 = base[0], t1 = base[1];
        unsigned m0 = base[3 * (kRawP / 4)], m1 = base[3 * (kRawP / 4) + 1], m2 = base[3 * (kRawP / 4) + 2];
        unsigned b0 = base[6 * (kRawP / 4)], b1 = base[6 * (kRawP / 4) + 1];
        bool pA = true, pB = true;
        if (MODE != 3) { pA = fast_precheck_pair(t0, t1, m0, m1, b0, b1, t); pB = fast_precheck_pair_b(t0, t1, m0, m1, m2, b0, b1, t); }
        if (MODE == 1) { acc += (pA ? 1u : 0u) + (pB ? 2u : 0u); continue; }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (!(h ? pB : pA)) continue;
            unsigned lo[7], hi[7];
#pragma unroll
            for (int r = 0; r < 7; r++) {
                const unsigned w0 = base[r * (kRawP / 4)], w1 = base[r * (kRawP / 4) + 1], w2 = base[r * (kRawP / 4) + 2];
                lo[r] = __builtin_amdgcn_alignbyte(w1, w0, 2 * h);
                hi[r] = __builtin_amdgcn_alignbyte(w2, w1, 2 * h);
            }
            acc += fast_score_pair(lo, hi, t);
        }
    }
    if (acc == 0x12345677u) out[0] = acc;
}

template <int MODE>
void run(const uint8_t* d, int nImg, unsigned* dOut, const char* what)
{
    const int tilesX = (PW - 32 + kFastTW - 1) / kFastTW, tilesY = (PH - 32 + kFastTH - 1) / kFastTH;
    const dim3 grid((nImg + 7) / 8 * 8 * tilesX * tilesY);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ring<MODE>, grid, dim3(256), 0, 0, d, nImg, 20, dOut);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double px = (double)nImg * (PW - 32) * (PH - 32);
    printf("mode %d (%s): %7.1f us per launch of %d planes, %.1f pixels/ns, %.3f of the 8 TB/s roofline (1 B per pixel)\n", MODE, what,
           best * 1e3, nImg, px / (best * 1e6), px / (best * 1e-3) / 8e12);
}

int main()
{
    const int nImg = 256;
    std::vector<uint8_t> h((size_t)PP * PH);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); i++) {                 // smooth gradient + blocks + noise: ~10 % of the pairs pass the pre-check, like the bench's scenes
        s = s * 1664525u + 1013904223u;
        const int x = (int)(i % PP), y = (int)(i / PP);
        int v = (x / 3 + y / 2) & 255;
        if (((x / 37) ^ (y / 29)) & 1) v = (v + 90) & 255;
        v += (int)((s >> 28) & 3) - 1;
        h[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
    uint8_t* d; unsigned* dOut;
    hipMalloc(&d, (size_t)nImg * PP * PH + 64); hipMalloc(&dOut, 64);
    for (int i = 0; i < nImg; i++) hipMemcpy(d + (size_t)i * PP * PH, h.data(), h.size(), hipMemcpyHostToDevice);
    run<0>(d, nImg, dOut, "staging only");
    run<1>(d, nImg, dOut, "+ compass pre-check on every pair");
    run<2>(d, nImg, dOut, "+ ring score where it passes, in place");
    run<3>(d, nImg, dOut, "ring score on every pair");
    return 0;
}
