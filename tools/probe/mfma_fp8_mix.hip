// Gate (ii) of the r05 verdict's item 1: can the two CORRECTION products of the split-f16 scheme (a_hi*w_lo + a_lo*w_hi) run on the
// block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3: twice the f16 rate; fp6 e2m3: four times) into the SAME accumulators as
// the f16 hi*hi product, next to the VALU stream a whole-block FCN kernel carries -- and what does it buy?
//   part 1: semantics, with exact data: the cvt_scalef32 conversions (is it src * scale or src / scale?), the A/B operand lane maps of
//           the scaled MFMA (lane (r, h) holds k = 32 h + j in byte j of its 8 registers?), the E8M0 scale operands, fp6 packing.
//   part 2: rates.  One "unit" = one B fragment set of 64 f32-equivalent K against T = 5 output tiles (block 15 / 16's projection):
//           f16x3: 60 v_mfma_f32_32x32x16_f16;  fp8: 20 f16 + 10 scaled fp8 (K = 64 = [a_hi | a_lo] of 32 channels);  fp6: 20 f16 + 10 fp6.
//           NV independent v_fma_f32 per unit interleaved (the kernel's stencil / split / epilogue stream), plus the conversions the
//           variant needs (fp8: 32 v_cvt_scalef32_pk_fp8_f16 per unit and lane; fp6: 2 v_cvt_scalef32_pk32_fp6_f16).
// Build: make -C tools/probe mfma_fp8_mix ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));

// ------------------------------------------------------------------------------------------------------------ part 1: semantics
__global__ void k_cvt(int* o, const float* xs, const float* scs, int n)
{
    // one thread: for every (x, scale): fp8 bytes of (x, 2x) from f16 and from f32 sources
    for (int i = 0; i < n; i++) {
        s16x2 old = {0, 0};
        f16x2 h = {(_Float16)xs[i], (_Float16)(2.f * xs[i])};
        s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, h, scs[i], false);
        s16x2 r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, xs[i], 2.f * xs[i], scs[i], false);
        o[2 * i] = (unsigned short)r[0]; o[2 * i + 1] = (unsigned short)r2[0];
    }
}
__global__ void k_cvt6(int* o, float step, float sc)
{
    f16x32 hh;
    for (int i = 0; i < 32; i++) hh[i] = (_Float16)(step * i);
    i32x6 p = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hh, sc);
    for (int i = 0; i < 6; i++) o[i] = p[i];
}
// D = A.B with A [32][64], B [64][32] given as per-lane register images (8 dwords per lane each); FMT 0 = fp8 e4m3, 2 = fp6 e2m3
template <int FMT>
__global__ void k_mfma8(float* d, const int* A, const int* B, int sa, int sb, int with_f16)
{
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = A[l * 8 + i]; b[i] = B[l * 8 + i]; }
    f32x16 c = {};
    if (with_f16) {                                        // mix: an f16 product into the same accumulators first: ones x ones, K = 16
        f16x8 x, y;
        for (int i = 0; i < 8; i++) { x[i] = (_Float16)1.f; y[i] = (_Float16)1.f; }
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT, FMT, 0, sa, 0, sb);
    for (int i = 0; i < 16; i++) d[l * 16 + i] = c[i];
}
// fp6 operands made on the device by the pk32 conversion from f16 rows (32 values per lane)
__global__ void k_mfma6(float* d, const _Float16* A, const _Float16* B)
{
    const int l = threadIdx.x;
    f16x32 x, y;
    for (int i = 0; i < 32; i++) { x[i] = A[l * 32 + i]; y[i] = B[l * 32 + i]; }
    i32x6 pa = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x, 1.0f);
    i32x6 pb = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(y, 1.0f);
    i32x8 a = {pa[0], pa[1], pa[2], pa[3], pa[4], pa[5], 0, 0}, b = {pb[0], pb[1], pb[2], pb[3], pb[4], pb[5], 0, 0};
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, 127, 0, 127);
    for (int i = 0; i < 16; i++) d[l * 16 + i] = c[i];
}

static uint8_t e4m3(int n)                                  // exact small integers 0..15 (and negatives)
{
    if (n == 0) return 0;
    uint8_t s = n < 0 ? 0x80 : 0; n = std::abs(n);
    int e = 0; while ((n >> (e + 1)) != 0) e++;
    int m = (n * 8 >> e) - 8;
    return (uint8_t)(s | ((e + 7) << 3) | m);
}

static void semantics()
{
    // (a) conversions
    {
        const float xs[] = {1.f, 1.f, 1.f, 1.f, 3.f, 0.01f, 0.01f, 300.f, 500.f}, sc[] = {1.f, 2.f, 0.5f, 4.f, 1.f, 1.f, 1.f / 64.f, 1.f, 1.f};
        const int n = 9;
        float *dx, *ds; int* d_o; int ho[2 * n];
        (void)hipMalloc(&dx, sizeof(xs)); (void)hipMalloc(&ds, sizeof(sc)); (void)hipMalloc(&d_o, sizeof(ho));
        (void)hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice); (void)hipMemcpy(ds, sc, sizeof(sc), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(1), 0, 0, d_o, dx, ds, n);
        (void)hipMemcpy(ho, d_o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("cvt_scalef32_pk_fp8 of (x, 2x) with scale s  (e4m3: 0x38 = 1, 0x40 = 2, 0x30 = 0.5, 0x48 = 4; low byte = x):\n");
        for (int i = 0; i < n; i++) printf("   x = %-7g s = %-8g from f16: 0x%04x   from f32: 0x%04x\n", xs[i], sc[i], ho[2 * i], ho[2 * i + 1]);
        int h6[6];
        for (float step : {0.25f, 0.125f}) {
            hipLaunchKernelGGL(k_cvt6, dim3(1), dim3(1), 0, 0, d_o, step, 1.0f);
            (void)hipMemcpy(h6, d_o, sizeof(h6), hipMemcpyDeviceToHost);
            printf("cvt_scalef32_pk32_fp6_f16 of i * %g, i = 0..31, scale 1:", step);
            for (int i = 0; i < 6; i++) printf(" %08x", (unsigned)h6[i]);
            printf("\n   as 6-bit fields, element i at bits [6i, 6i+6):");
            for (int i = 0; i < 32; i++) {
                unsigned bit = 6 * i, w = bit >> 5, sft = bit & 31;
                uint64_t two = (uint64_t)(unsigned)h6[w] | (w + 1 < 6 ? (uint64_t)(unsigned)h6[w + 1] << 32 : 0);
                printf(" %02x", (unsigned)((two >> sft) & 63));
            }
            printf("\n");
        }
        (void)hipFree(dx); (void)hipFree(ds); (void)hipFree(d_o);
    }
    // (b) fp8 scaled MFMA: operand lane map + scales, exact integers
    {
        std::vector<int> Am(32 * 64), Bm(64 * 32);
        for (int i = 0; i < 32; i++) for (int k = 0; k < 64; k++) Am[i * 64 + k] = ((i * 7 + k * 3) % 9) - 4;
        for (int k = 0; k < 64; k++) for (int j = 0; j < 32; j++) Bm[k * 32 + j] = ((k * 5 + j * 11) % 7) - 3;
        std::vector<uint8_t> Ar(64 * 32), Br(64 * 32);
        for (int l = 0; l < 64; l++) for (int e = 0; e < 32; e++) {
            const int r = l & 31, h = l >> 5, k = 32 * h + e;
            Ar[l * 32 + e] = e4m3(Am[r * 64 + k]); Br[l * 32 + e] = e4m3(Bm[k * 32 + r]);
        }
        int *dA, *dB; float* dD; std::vector<float> D(64 * 16);
        (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dD, 4096);
        (void)hipMemcpy(dA, Ar.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Br.data(), 2048, hipMemcpyHostToDevice);
        struct { int sa, sb, f16; double mul, add; const char* what; } cases[] = {
            {127, 127, 0, 1.0, 0.0, "scales 127 / 127 (2^0)"}, {128, 127, 0, 2.0, 0.0, "scale_a 128 (x2)"}, {127, 124, 0, 0.125, 0.0, "scale_b 124 (x 2^-3)"},
            {127 | (130 << 8), 127, 0, 1.0, 0.0, "scale_a byte 1 = 130 ignored with opsel 0"}, {127, 127, 1, 1.0, 16.0, "after an f16 ones x ones product (+16)"}};
        for (auto& cs : cases) {
            hipLaunchKernelGGL(k_mfma8<0>, dim3(1), dim3(64), 0, 0, dD, dA, dB, cs.sa, cs.sb, cs.f16);
            (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int l = 0; l < 64; l++) for (int q = 0; q < 16; q++) {
                const int col = l & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (l >> 5);
                double ref = 0; for (int k = 0; k < 64; k++) ref += (double)Am[row * 64 + k] * Bm[k * 32 + col];
                worst = std::fmax(worst, std::fabs(D[l * 16 + q] - (ref * cs.mul + cs.add)));
            }
            printf("scaled fp8 MFMA, lane (r, h) element e = k 32h + e, %-48s max |D - ref| = %g\n", cs.what, worst);
        }
        // (c) fp6 through the device conversion
        std::vector<_Float16> A6(64 * 32), B6(64 * 32);
        std::vector<float> Af(32 * 64), Bf(64 * 32);
        for (int i = 0; i < 32; i++) for (int k = 0; k < 64; k++) Af[i * 64 + k] = (((i * 7 + k * 3) % 13) - 6) * 0.5f;       // multiples of 0.5 in [-3, 3]: exact in e2m3
        for (int k = 0; k < 64; k++) for (int j = 0; j < 32; j++) Bf[k * 32 + j] = (((k * 5 + j * 11) % 9) - 4) * 0.25f;       // multiples of 0.25 in [-1, 1]: exact
        for (int l = 0; l < 64; l++) for (int e = 0; e < 32; e++) {
            const int r = l & 31, h = l >> 5, k = 32 * h + e;
            A6[l * 32 + e] = (_Float16)Af[r * 64 + k]; B6[l * 32 + e] = (_Float16)Bf[k * 32 + r];
        }
        _Float16 *dA6, *dB6;
        (void)hipMalloc(&dA6, 4096); (void)hipMalloc(&dB6, 4096);
        (void)hipMemcpy(dA6, A6.data(), 4096, hipMemcpyHostToDevice); (void)hipMemcpy(dB6, B6.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma6, dim3(1), dim3(64), 0, 0, dD, dA6, dB6);
        (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int l = 0; l < 64; l++) for (int q = 0; q < 16; q++) {
            const int col = l & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (l >> 5);
            double ref = 0; for (int k = 0; k < 64; k++) ref += (double)Af[row * 64 + k] * Bf[k * 32 + col];
            worst = std::fmax(worst, std::fabs(D[l * 16 + q] - ref));
        }
        printf("scaled fp6 (e2m3) MFMA on operands from v_cvt_scalef32_pk32_fp6_f16, same lane map:                max |D - ref| = %g\n", worst);
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD); (void)hipFree(dA6); (void)hipFree(dB6);
    }
}

// ------------------------------------------------------------------------------------------------------------ part 2: rates
// MODE 0: f16 x 3;  1: f16 + fp8 corrections;  2: f16 + fp6 corrections;  3: f16 hi*hi only (the floor).  T = 5 output tiles.
template <int MODE, int NV, int CVT>
__global__ __launch_bounds__(512) void k_rate(float* out, int iters, float a0, float b0)
{
    constexpr int T = 5;
    f32x16 acc[T];
    for (int t = 0; t < T; t++) for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
    f16x8 ah[T], al[T];                                   // "weights" hi / lo per tile (loop-invariant, like the kernel's A fragments)
    for (int t = 0; t < T; t++) for (int i = 0; i < 8; i++) { ah[t][i] = (_Float16)(a0 + t + i); al[t][i] = (_Float16)(a0 * 0.001f + i); }
    i32x8 a8[T];
    for (int t = 0; t < T; t++) for (int i = 0; i < 8; i++) a8[t][i] = 0x38383838 + t + (threadIdx.x & 3);
    float fa = a0 + (threadIdx.x & 3), fb = b0;
    float v[32];
    for (int i = 0; i < 32; i++) v[i] = a0 * i + threadIdx.x;
    const int sa = 127, sb = 120;
    constexpr int NMF = (MODE == 0 ? 12 : MODE == 3 ? 4 : 6) * T;           // matrix instructions per unit
    constexpr int PER = NV / NMF, REM = NV - PER * NMF;
    for (int it = 0; it < iters; it++) {
        // this unit's B operand: 4 K-steps of 8 f16 (hi) and 8 f16 (lo) per lane, made from the VALU stream's registers
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                auto p = __builtin_amdgcn_cvt_pkrtz(v[8 * s + i], v[8 * s + i + 1]);                      // 16 distinct pairs per unit
                bh[s][i] = p[0]; bh[s][i + 1] = p[1];
                auto q = __builtin_amdgcn_cvt_pkrtz(v[8 * s + i] * 0.001f, v[8 * s + i + 1] * 0.001f);      // stands for x - float(hi)
                bl[s][i] = q[0]; bl[s][i + 1] = q[1];
            }
        i32x8 b8[2];
        if (MODE == 1) {
            if (CVT) {
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int w = 0; w < 8; w++) {          // one dword = 4 fp8 = two conversions of an f16 pair
                        s16x2 r = {0, 0};
                        const f16x8& src = (w & 4) ? bl[2 * u + ((w >> 1) & 1)] : bh[2 * u + ((w >> 1) & 1)];
                        f16x2 p0 = {src[(w & 1) * 4 + 0], src[(w & 1) * 4 + 1]}, p1 = {src[(w & 1) * 4 + 2], src[(w & 1) * 4 + 3]};
                        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, p0, 1.0f, false);
                        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, p1, 1.0f, true);
                        b8[u][w] = (int)(unsigned short)r[0] | ((int)r[1] << 16);
                    }
            } else {
                for (int u = 0; u < 2; u++) for (int w = 0; w < 8; w++) b8[u][w] = __builtin_bit_cast(int, v[u * 8 + w]) & 0x3f3f3f3f;
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (CVT) {
                    f16x32 src;
#pragma unroll
                    for (int i = 0; i < 8; i++) { src[i] = bh[2 * u][i]; src[8 + i] = bl[2 * u][i]; src[16 + i] = bh[2 * u + 1][i]; src[24 + i] = bl[2 * u + 1][i]; }
                    i32x6 p = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(src, 1.0f);
                    b8[u] = i32x8{p[0], p[1], p[2], p[3], p[4], p[5], 0, 0};
                } else {
                    for (int w = 0; w < 8; w++) b8[u][w] = __builtin_bit_cast(int, v[u * 8 + w]) & 0x1b6db6db;
                }
            }
        }
        int nv = 0, nm = 0;
        auto valu = [&]() {
#pragma unroll
            for (int j = 0; j < PER + (nm < REM ? 1 : 0); j++) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[nv & 31]) : "v"(fa), "v"(fb)); nv++; }
            nm++;
        };
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int t = 0; t < T; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bh[s], acc[t], 0, 0, 0); valu();
                if (MODE == 0) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], bh[s], acc[t], 0, 0, 0); valu();
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], bl[s], acc[t], 0, 0, 0); valu();
                }
            }
            if ((MODE == 1 || MODE == 2) && (s & 1)) {
#pragma unroll
                for (int t = 0; t < T; t++) {
                    acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[t], b8[s >> 1], acc[t], MODE == 1 ? 0 : 2, MODE == 1 ? 0 : 2, 0, sa, 0, sb); valu();
                }
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < T; t++) for (int q = 0; q < 16; q++) s += acc[t][q];
    for (int i = 0; i < 32; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV, int CVT>
static double run(int threads, const char* name)
{
    float* d; (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<MODE, NV, CVT>), dim3(256), dim3(threads), 0, 0, d, 50, 1.f, 0.5f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<MODE, NV, CVT>), dim3(256), dim3(threads), 0, 0, d, iters, 1.f, 0.5f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.0e9 / iters;
    printf("   %-58s %d waves/SIMD  %8.3f ms = %6.0f cycles/unit @2.0GHz\n", name, threads / 256, ms, cyc);
    (void)hipFree(d);
    return cyc;
}

template <int NV>
static void rates()
{
    printf("VALU stream: %d v_fma_f32 per unit (+ the f16 split of the B operand; + the fp8 / fp6 conversions where marked)\n", NV);
    for (int threads : {256, 512}) {
        const double f3 = run<0, NV, 0>(threads, "f16 x 3 (60 MFMA)");
        const double f1 = run<3, NV, 0>(threads, "f16 hi*hi only (20 MFMA; the floor)");
        const double p8n = run<1, NV, 0>(threads, "f16 + fp8 corrections (20 + 10), no conversions");
        const double p8 = run<1, NV, 1>(threads, "f16 + fp8 corrections (20 + 10), 32 cvt_pk_fp8_f16");
        const double p6n = run<2, NV, 0>(threads, "f16 + fp6 corrections (20 + 10), no conversions");
        const double p6 = run<2, NV, 1>(threads, "f16 + fp6 corrections (20 + 10), 2 cvt_pk32_fp6_f16");
        printf("   => f16x3 / fp8: %.2fx (%.2fx without its conversions);  f16x3 / fp6: %.2fx (%.2fx);  f16x3 / floor: %.2fx\n",
               f3 / p8, f3 / p8n, f3 / p6, f3 / p6n, f3 / f1);
    }
}

// ------------------------------------------------------------------------------------------------------------ part 3: the form the kernel uses
// v_mfma_scale_f32_16x16x128_f8f6f4 with A = host-packed bf6 (e3m2) rows, B = fp6 (e2m3) made by v_cvt_scalef32_pk32_fp6_f16 with a
// PER-LANE scale (the lane's own 2^sb, from its largest element), scale bytes selected by opsel from a packed register.
typedef float f32x4_ __attribute__((ext_vector_type(4)));
static float dec6(int code, bool bf6)                         // value of a 6-bit code
{
    const int sgn = code & 32; code &= 31;
    float v;
    if (bf6) { const int e = code >> 2, m = code & 3; v = e ? (1.f + m * 0.25f) * std::ldexp(1.f, e - 3) : m * 0.0625f; }
    else     { const int e = code >> 3, m = code & 7; v = e ? (1.f + m * 0.125f) * std::ldexp(1.f, e - 1) : m * 0.125f; }
    return sgn ? -v : v;
}
static int enc6(float x, bool bf6)                            // nearest code, ties to the even code, saturating
{
    const float a = std::fabs(x);
    int best = 0; float bd = 1e30f;
    for (int c = 0; c < 32; c++) {
        const float d = std::fabs(dec6(c, bf6) - a);
        if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; }
    }
    return best | (x < 0.f ? 32 : 0);
}
static void pack6(uint32_t* dst6, const int* codes32)         // element i at bits [6 i, 6 i + 6) of 6 dwords
{
    for (int i = 0; i < 6; i++) dst6[i] = 0;
    for (int i = 0; i < 32; i++) {
        const unsigned bit = 6 * i, w = bit >> 5, sft = bit & 31;
        dst6[w] |= (uint32_t)codes32[i] << sft;
        if (sft > 26) dst6[w + 1] |= (uint32_t)codes32[i] >> (32 - sft);
    }
}
// one wave: D[16][16] = sum_k A[m][k] B[k][n], K = 128; lane (r = l & 15, kq = l >> 4) holds k = 32 kq + e of row / column r
__global__ void k_mfma6x128(float* d, const int* A6, const _Float16* B, int* sbOut, int scaleA)
{
    const int l = threadIdx.x;
    i32x8 a = {A6[l * 6], A6[l * 6 + 1], A6[l * 6 + 2], A6[l * 6 + 3], A6[l * 6 + 4], A6[l * 6 + 5], 0, 0};
    f16x32 y; float amax = 0.f;
    for (int i = 0; i < 32; i++) { y[i] = B[l * 32 + i]; amax = fmaxf(amax, fabsf((float)y[i])); }
    const int sb = __builtin_amdgcn_frexp_expf(amax) - 3;                  // amax / 2^sb in [4, 8)
    const float scl = __builtin_bit_cast(float, (127 + sb) << 23);
    i32x6 pb = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(y, scl);
    i32x8 b = {pb[0], pb[1], pb[2], pb[3], pb[4], pb[5], 0, 0};
    const int sbReg = (77 << 0) | ((127 + sb) << 16) | (55 << 24);         // the lane's byte sits in byte 2: opsel 2
    f32x4_ c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 3, 2, 0, scaleA, 2, sbReg);
    for (int i = 0; i < 4; i++) d[l * 4 + i] = c[i];
    sbOut[l] = sb;
}
template <int CB, int BL>
__global__ __launch_bounds__(256) void k_rate128(float* out, int iters)
{
    i32x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0x08208208 + threadIdx.x; b[i] = 0x10410410 + i; }
    f32x4_ acc[8];
    for (int t = 0; t < 8; t++) acc[t] = f32x4_{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[t], CB, BL, 0, 120, 0, 127);
    float s = 0.f;
    for (int t = 0; t < 8; t++) for (int q = 0; q < 4; q++) s += acc[t][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CB, int BL>
static void rate128(const char* name)
{
    float* d; (void)hipMalloc(&d, 256 * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate128<CB, BL>), dim3(256), dim3(256), 0, 0, d, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate128<CB, BL>), dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("   v_mfma_scale_f32_16x16x128_f8f6f4 %-28s %.1f cycles per instruction @2.0GHz (one wave per SIMD, 8 independent accumulators)\n", name, ms * 1e-3 * 2.0e9 / iters / 8);
    (void)hipFree(d);
}
static void kernel_form()
{
    std::vector<float> Af(16 * 128), Bf(128 * 16);
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 8) & 0xffff) / 65536.f; };
    for (auto& v : Af) v = (rnd() - 0.5f) * 30.f;                                   // weights-like, in +-15: bf6 range (28)
    for (int k = 0; k < 128; k++) for (int n = 0; n < 16; n++) Bf[k * 16 + n] = (rnd() - 0.3f) * std::ldexp(1.f, (n * 3 + k / 32 * 5) % 19 - 6);   // magnitudes differ per (column, K block)
    std::vector<uint32_t> A6(64 * 6); std::vector<_Float16> Bh(64 * 32);
    std::vector<float> Aq(16 * 128);
    for (int l = 0; l < 64; l++) {
        int codes[32];
        for (int e = 0; e < 32; e++) { const int m = l & 15, k = 32 * (l >> 4) + e; codes[e] = enc6(Af[m * 128 + k], true); Aq[m * 128 + k] = dec6(codes[e], true); }
        pack6(&A6[l * 6], codes);
        for (int e = 0; e < 32; e++) Bh[l * 32 + e] = (_Float16)Bf[(32 * (l >> 4) + e) * 16 + (l & 15)];
    }
    int* dA; _Float16* dB; float* dD; int* dS;
    (void)hipMalloc(&dA, 64 * 24); (void)hipMalloc(&dB, 64 * 64); (void)hipMalloc(&dD, 1024); (void)hipMalloc(&dS, 256);
    (void)hipMemcpy(dA, A6.data(), 64 * 24, hipMemcpyHostToDevice); (void)hipMemcpy(dB, Bh.data(), 64 * 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma6x128, dim3(1), dim3(64), 0, 0, dD, dA, dB, dS, 127 - 3);
    std::vector<float> D(256); std::vector<int> S(64);
    (void)hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(S.data(), dS, 256, hipMemcpyDeviceToHost);
    // reference: A dequantised (host codes), B quantised per lane block with the lane's scale (host enc6 = the device conversion?), x 2^-3
    double worst = 0, big = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        const int n = l & 15, m = 4 * (l >> 4) + r;
        double ref = 0;
        for (int k = 0; k < 128; k++) {
            const int sb = S[(k >> 5) * 16 + n];
            const float bq = dec6(enc6((float)(_Float16)Bf[k * 16 + n] * std::ldexp(1.f, -sb), false), false) * std::ldexp(1.f, sb);
            ref += (double)Aq[m * 128 + k] * bq;
        }
        ref *= 0.125;
        worst = std::fmax(worst, std::fabs(D[l * 4 + r] - ref)); big = std::fmax(big, std::fabs(ref));
    }
    printf("16x16x128: A bf6 host-packed (lane (m, kq): k = 32 kq + e), B fp6 by the device conversion with PER-LANE scales (opsel 2), scale_a 2^-3:\n"
           "   max |D - ref| = %g (largest |ref| %g; f32 accumulation order only)   lane scales sb: %d %d %d %d ... %d\n", worst, big, S[0], S[1], S[2], S[3], S[63]);
    rate128<0, 0>("fp8 x fp8:"); rate128<2, 2>("fp6 x fp6:"); rate128<3, 2>("bf6 (A) x fp6 (B):"); rate128<3, 3>("bf6 x bf6:"); rate128<0, 2>("fp8 (A) x fp6 (B):");
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD); (void)hipFree(dS);
}

// the same harness for every matrix instruction in question: NACC independent accumulators, back to back, one wave per SIMD
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k_rate_any(float* out, int iters)
{
    i32x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0x08208208 + threadIdx.x; b[i] = 0x10410410 + i; }
    f16x8 ha, hb;
    for (int i = 0; i < 8; i++) { ha[i] = (_Float16)(1.f + i); hb[i] = (_Float16)(0.5f * i); }
    const int sa = 120, sb = 127;
    float s = 0.f;
    if (KIND == 0 || (KIND >= 2 && KIND <= 4)) {
        f32x16 acc[NACC];
        for (int t = 0; t < NACC; t++) for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int t = 0; t < NACC; t++) {
                if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[t], 0, 0, 0);
                if (KIND == 2) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 0, 0, 0, sa, 0, sb);
                if (KIND == 3) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 2, 2, 0, sa, 0, sb);
                if (KIND == 4) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 3, 2, 0, sa, 0, sb);
            }
        for (int t = 0; t < NACC; t++) for (int q = 0; q < 16; q++) s += acc[t][q];
    } else {
        // the 16 x 16 forms through inline asm: the compiler's allocation of the four-register accumulators shuffled them through
        // v_accvgpr moves every iteration (measured 35 cycles for v_mfma_f32_16x16x32_f16)
        f32x4_ acc4[NACC];
        for (int t = 0; t < NACC; t++) acc4[t] = f32x4_{0.f, 0.f, 0.f, 0.f};
        const i32x6 a6 = {a[0], a[1], a[2], a[3], a[4], a[5]}, b6 = {b[0], b[1], b[2], b[3], b[4], b[5]};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int t = 0; t < NACC; t++) {
                if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4[t]) : "v"(ha), "v"(hb));
                if (KIND == 5) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc4[t]) : "v"(a), "v"(b), "v"(sa), "v"(sb));
                if (KIND == 6) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(acc4[t]) : "v"(a6), "v"(b6), "v"(sa), "v"(sb));
                if (KIND == 7) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:3 blgp:2" : "+v"(acc4[t]) : "v"(a6), "v"(b6), "v"(sa), "v"(sb));
            }
        for (int t = 0; t < NACC; t++) for (int q = 0; q < 4; q++) s += acc4[t][q];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int NACC>
static void rate_any(const char* name, double macs)
{
    float* d; (void)hipMalloc(&d, 256 * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate_any<KIND, NACC>), dim3(256), dim3(256), 0, 0, d, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate_any<KIND, NACC>), dim3(256), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.0e9 / iters / NACC;
    printf("   %-44s %d accumulators: %6.1f cycles @2.0GHz per instruction = %6.0f MAC per cycle and SIMD\n", name, NACC, cyc, macs / cyc);
    (void)hipFree(d);
}
static void rates_plain()
{
    printf("matrix instruction issue, back to back (one wave per SIMD):\n");
    rate_any<0, 4>("v_mfma_f32_32x32x16_f16", 16384); rate_any<0, 8>("v_mfma_f32_32x32x16_f16", 16384);
    rate_any<1, 4>("v_mfma_f32_16x16x32_f16", 8192); rate_any<1, 8>("v_mfma_f32_16x16x32_f16", 8192);
    rate_any<2, 4>("scale 32x32x64 fp8 x fp8", 65536); rate_any<3, 4>("scale 32x32x64 fp6 x fp6", 65536); rate_any<3, 8>("scale 32x32x64 fp6 x fp6", 65536);
    rate_any<4, 4>("scale 32x32x64 bf6 x fp6", 65536);
    rate_any<5, 4>("scale 16x16x128 fp8 x fp8", 32768); rate_any<6, 2>("scale 16x16x128 fp6 x fp6", 32768); rate_any<6, 4>("scale 16x16x128 fp6 x fp6", 32768);
    rate_any<6, 8>("scale 16x16x128 fp6 x fp6", 32768); rate_any<7, 8>("scale 16x16x128 bf6 x fp6", 32768);
}

int main(int argc, char** argv)
{
    semantics();
    kernel_form();
    rates_plain();
    if (argc > 1) return 0;
    rates<0>();
    rates<200>();
    rates<420>();       // block 15 / 16's mix: matrix pipe busy 0.50, VALU issue 0.43 of the kernel's time => ~84 VALU per 12 MFMAs
    return 0;
}
