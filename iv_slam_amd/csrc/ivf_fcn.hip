// ivf_fcn.hip -- introspection FCN forward pass on gfx950.
//
// Reference: IntrospectionModule.forward (IF/networks/models_light/models_light.py:18-28) = bilinear resize to
// 512x512 -> MobileNetV2 dilated to output stride 8 (models_light.py:99-172, mobilenet.py:35-110) -> C1 decoder
// (models_light.py:176-204) -> bilinear to out_size -> sigmoid(20(x-0.5)); called from C++ at
// ORB/Examples/Stereo/stereo_kitti.cc:493-514 (BGR->RGB, /255, -mean, /std, NCHW; (y*255).to(u8) truncation).
//
// Layout: NCHW f32 planes (pixels contiguous).  That makes the activation operand of a 1x1 convolution a
// coalesced 128-byte row per channel and needs no LDS transposes:
//   D[cout][pixel] += W[cout][k] * X[k][pixel]   on v_mfma_f32_32x32x16_f16 with BOTH operands split into two f16
//   halves (x = hi + lo, 22 significant bits) and three MFMAs per product: hi*hi + hi*lo + lo*hi, f32 accumulate.
//   That reproduces the f32 product to ~2^-21 (the 1e-3 bound after the slope-5 logistic leaves no room for plain
//   f16/bf16 inputs, and three bf16 terms measure 1.1e-3), runs 4x faster than v_mfma_f32_32x32x2_f32 (measured:
//   tools/probe/) and, unlike the f32 MFMA, lets ordinary VALU work issue underneath it.  gfx950 keeps f16
//   subnormals in MFMA inputs (tools/probe/mfma_f16_check.hip), which the lo halves rely on.
//   A = weights split and pre-shuffled on the host into the MFMA A-fragment order, B = activations loaded straight
//   from global as float/float2/float4 per lane and split in registers, C/D rows = output channels, columns =
//   pixels, so each accumulator register stores a contiguous pixel run.
// BN (eval mode, eps 1e-5) is folded into per-channel scale/shift at load; ReLU6 / ReLU / residual add are fused
// into the GEMM and depthwise epilogues.  Depthwise 3x3 (stride/dilation) is an HBM-bound VALU stencil.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "ivf_device.h"

#define ffail ivf::set_error

// kernels live in a NAMED namespace: __global__ functions with internal linkage (anonymous namespace)
// failed to resolve at launch on ROCm 7 when the .so holds several HIP translation units
#include <type_traits>
namespace ivffcn {

#define FHIP(expr)                                                                                   \
    do { hipError_t e_ = (expr);                                                                      \
         if (e_ != hipSuccess) return ffail(IVF_E_NO_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
union HFrag { f16x8 v; uint4 q; uint32_t u[4]; };       // one lane's 8 f16 of an MFMA 32x32x16 A or B operand
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));

// two f32 -> packed f16 hi pair + packed f16 lo pair, x = hi + lo (hi, lo rounded toward zero; x - hi is exact)
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& hi, uint32_t& lo)
{
    const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]));
}

// ---- f16 range guard (r05) ----
// split_pair clamps |x| >= 65,504 to the largest f16, and the lo half with it: a silently wrong product, no flag.  Weights are pre-scaled
// per output channel and every hidden tensor is bounded by ReLU6; the UN-CLAMPED activations are the linear-bottleneck outputs (BN, no
// activation, + residual), which the next block's expansion splits.  Every kernel that stores one keeps the largest magnitude it stored
// and ORs bit 0 into g_fcnRange when that reaches 65,504 (inf included; a NaN can only follow an inf, and the blob is checked for
// non-finite weights at ivf_fcn_create).  k_fcn_out, the last kernel of every forward, moves the word into the HANDLE's status word;
// ivf_fcn_forward / ivf_fcn_status report it as IVF_E_STATE -- like the front end's consistency flags.  Cost: one v_max per stored value.
__device__ int g_fcnRange = 0;
__device__ __forceinline__ void range_note(float& amax, float v) { amax = fmaxf(amax, fabsf(v)); }
__device__ __forceinline__ void range_flag(float amax) { if (!(amax < 65504.f)) atomicOr(&g_fcnRange, 1); }

constexpr int kEnc = 512;                      // encoder input size (IF/config: enc_input_size)
constexpr int kSubBatch = 64;                  // images per launch sequence of ivf_fcn_forward_device (r06: see there)

// ---- pre-processing + bilinear resize to 512x512 (stereo_kitti.cc:494-506, models_light.py:19-21) ----
#ifndef IVF_PREP_ROWS
#define IVF_PREP_ROWS 2          // measured 1 / 2 / 4 / 8: 176-178 / 151-154 / 153-157 / 161-170 us per 128 images
#endif
constexpr int kPrepRows = IVF_PREP_ROWS;       // output rows per thread of k_fcn_prep (grid y = kEnc / kPrepRows)
__global__ void k_fcn_prep(const uint8_t* __restrict__ bgr, size_t imageStride, int rowStride, int w, int h,
                           float* __restrict__ out)
{
    // (r05: an XCD-aware block order -- XCD k takes images k, k + 8, ... whole, so that the two output rows that share an input row
    // meet in one L2 -- brings FETCH_SIZE down from 2.9x the input but not the time: 187 vs 180 us per 128 images; not kept.  The kernel
    // is bound by its 403 MB of f32 stores and its VALU work, not by the u8 fetches.)
    // r05: a thread walks kPrepRows consecutive output rows at its column: the horizontal taps, weights and byte offsets are computed once,
    // and consecutive output rows share source rows (h < kEnc: 0.73 source rows per output row at 375) in this workgroup's L1.  Per element
    // the arithmetic and its order are unchanged.
    const int x = blockIdx.x * blockDim.x + threadIdx.x, yA = blockIdx.y * kPrepRows, b = blockIdx.z;
    if (x >= kEnc) return;
    const float sy_ = (float)h / (float)kEnc, sx_ = (float)w / (float)kEnc;
    float fx = sx_ * ((float)x + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
    int x0 = (int)fx; if (x0 > w - 1) x0 = w - 1;
    const int x1 = x0 + (x0 < w - 1);
    const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
    const uint8_t* I = bgr + (size_t)b * imageStride;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, istd[3] = {1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f};
    const bool wide = x0 + 3 <= w - 1;
    if (w >= 3) {
        // r06: branch-free taps, every row of the thread in flight at once.  A row's two taps are the 6 bytes from 3 x0 (3 when x1 = x0, the last column): ONE unaligned
        // 8-byte load at min(3 x0, 3 w - 8) -- never past the row's end -- shifted down by the bytes in front of 3 x0.  The form below (`wide ? two dwords : twelve byte
        // loads`, row after row) was a load -> wait -> compute -> store round trip per output row; the bytes, the arithmetic and its order are the same.
        const int xb = min(3 * x0, 3 * w - 8), sh = 8 * (3 * x0 - xb), o1 = 24 * (x1 - x0);
        unsigned long long wt[kPrepRows], wb[kPrepRows];
        float ly0s[kPrepRows], ly1s[kPrepRows];
#pragma unroll
        for (int r = 0; r < kPrepRows; r++) {
            const int y = yA + r;
            float fy = sy_ * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
            int y0 = (int)fy; if (y0 > h - 1) y0 = h - 1;
            const int y1 = y0 + (y0 < h - 1);
            ly1s[r] = fy - (float)y0; ly0s[r] = 1.f - ly1s[r];
            __builtin_memcpy(&wt[r], I + (size_t)y0 * rowStride + xb, 8);
            __builtin_memcpy(&wb[r], I + (size_t)y1 * rowStride + xb, 8);
        }
#pragma unroll
        for (int r = 0; r < kPrepRows; r++) {
            unsigned tl = (unsigned)wt[r], th = (unsigned)(wt[r] >> 32), bl = (unsigned)wb[r], bh = (unsigned)(wb[r] >> 32);
            asm volatile("" : "+v"(tl), "+v"(th), "+v"(bl), "+v"(bh));            // (all rows requested before the first is consumed)
            wt[r] = ((unsigned long long)th << 32) | tl; wb[r] = ((unsigned long long)bh << 32) | bl;
        }
#pragma unroll
        for (int r = 0; r < kPrepRows; r++) {
            const int y = yA + r;
            const unsigned long long T = wt[r] >> sh, B = wb[r] >> sh;
            const float ly0 = ly0s[r], ly1 = ly1s[r];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int sc = 2 - c;                       // BGR -> RGB
                auto nz = [&](unsigned q) { return ((float)q * (1.0f / 255.0f) - mean[c]) * istd[c]; };
                const unsigned t0 = (unsigned)(T >> (8 * sc)) & 0xffu, t1 = (unsigned)(T >> (o1 + 8 * sc)) & 0xffu;
                const unsigned b0 = (unsigned)(B >> (8 * sc)) & 0xffu, b1 = (unsigned)(B >> (o1 + 8 * sc)) & 0xffu;
                const float top = nz(t0) * lx0 + nz(t1) * lx1;
                const float bot = nz(b0) * lx0 + nz(b1) * lx1;
                out[(((size_t)b * 3 + c) * kEnc + y) * kEnc + x] = top * ly0 + bot * ly1;
            }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < kPrepRows; r++) {
        const int y = yA + r;
        float fy = sy_ * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
        int y0 = (int)fy; if (y0 > h - 1) y0 = h - 1;
        const int y1 = y0 + (y0 < h - 1);
        const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
        // the four taps' three bytes each, loaded once; the normalisation `(p/255 - mean) / std` multiplies by the reciprocal of the
        // constant (one ulp from the true quotient, far inside the 1e-3 bar): the IEEE division sequence was most of this kernel
        // (r01: 80 VALU lane-instructions per output element, VALU-bound)
        const uint8_t *p00 = I + (size_t)y0 * rowStride + x0 * 3, *p01 = I + (size_t)y0 * rowStride + x1 * 3;
        const uint8_t *p10 = I + (size_t)y1 * rowStride + x0 * 3, *p11 = I + (size_t)y1 * rowStride + x1 * 3;
        // r02: twelve single-byte gathers per element made this kernel address-rate-bound (4 lane addresses per clock and CU).  The
        // two taps of a row are 6 adjacent bytes (B G R B G R) whenever x1 = x0 + 1: two unaligned dword loads per row, except in
        // the last columns, where the second dword would run past the row
        unsigned t0[3], t1[3], b0[3], b1[3];            // taps (row, column) per colour byte
        if (wide) {
            unsigned a0, a1, c0, c1;
            __builtin_memcpy(&a0, p00, 4); __builtin_memcpy(&a1, p00 + 4, 4);
            __builtin_memcpy(&c0, p10, 4); __builtin_memcpy(&c1, p10 + 4, 4);
            const unsigned a01 = __builtin_amdgcn_alignbyte(a1, a0, 3), c01 = __builtin_amdgcn_alignbyte(c1, c0, 3);   // bytes 3 .. 6
#pragma unroll
            for (int k = 0; k < 3; k++) {
                t0[k] = (a0 >> (8 * k)) & 0xffu; t1[k] = (a01 >> (8 * k)) & 0xffu;
                b0[k] = (c0 >> (8 * k)) & 0xffu; b1[k] = (c01 >> (8 * k)) & 0xffu;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) { t0[k] = p00[k]; t1[k] = p01[k]; b0[k] = p10[k]; b1[k] = p11[k]; }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int sc = 2 - c;                       // BGR -> RGB
            auto nz = [&](unsigned q) { return ((float)q * (1.0f / 255.0f) - mean[c]) * istd[c]; };
            const float top = nz(t0[sc]) * lx0 + nz(t1[sc]) * lx1;
            const float bot = nz(b0[sc]) * lx0 + nz(b1[sc]) * lx1;
            out[(((size_t)b * 3 + c) * kEnc + y) * kEnc + x] = top * ly0 + bot * ly1;
        }
    }
}

// ---- features[0]: conv 3x3 stride 2 pad 1, 3 -> 32, + BN + ReLU6 (mobilenet.py:19-24) ----
__global__ __launch_bounds__(256) void k_fcn_conv0(const float* __restrict__ X, const float* __restrict__ Wt,
                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                  float* __restrict__ Y)
{
    __shared__ float sw[32 * 27];
    __shared__ float ss[64];
    for (int i = threadIdx.x; i < 32 * 27; i += 256) sw[i] = Wt[i];
    if (threadIdx.x < 32) { ss[threadIdx.x] = scale[threadIdx.x]; ss[32 + threadIdx.x] = shift[threadIdx.x]; }
    __syncthreads();
    const int O = kEnc / 2;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= O) return;
    float v[27];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int yy = 2 * y - 1 + ky, xx = 2 * x - 1 + kx;
                v[(c * 3 + ky) * 3 + kx] = (yy >= 0 && yy < kEnc && xx >= 0 && xx < kEnc)
                                               ? X[(((size_t)b * 3 + c) * kEnc + yy) * kEnc + xx] : 0.f;
            }
    for (int co = 0; co < 32; co++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 27; k++) acc += sw[co * 27 + k] * v[k];
        float r = acc * ss[co] + ss[32 + co];
        r = fminf(fmaxf(r, 0.f), 6.f);
        Y[(((size_t)b * 32 + co) * O + y) * O + x] = r;
    }
}

// ---- stem: features[0] (conv 3x3 s2, 3 -> 32, BN, ReLU6) fused with block 1's depthwise 3x3 (+ BN + ReLU6) ----
// The 32 x 256 x 256 map between the two layers (8.4 MB per image, written and read back: the largest round trip of the
// 256x256 stage) stays in LDS.  One workgroup = a 32 x 8 output tile, all 32 channels:
//   1. the 69 x 21 x 3 input window goes to LDS (zero outside the 512 x 512 image = the conv's padding);
//   2. features[0] on the 34 x 10 positions the depthwise layer reads (1.33x recompute for the halo), one work item =
//      one position x 8 output channels, results (zero outside the 256 x 256 map = the depthwise padding) to LDS;
//   3. the depthwise stencil: thread = channel (tid >> 3) x 4 adjacent pixels ((tid & 7) * 4), rolling down the 8 rows.
constexpr int kStemTW = 32, kStemTH = 8, kStemCW = kStemTW + 2, kStemCH = kStemTH + 2;          // conv0 positions per tile
constexpr int kStemIW = 2 * kStemCW + 1, kStemIH = 2 * kStemCH + 1, kStemIP = 76;                // input window 69 x 21; LDS rows hold 72 floats from x = 2*ox0 - 4 (16-byte aligned), pitch 76
constexpr int kStemCP = kStemCW * kStemCH + 4;                                                   // floats per channel plane in LDS
__global__ __launch_bounds__(512, 4) void k_fcn_stem(const float* __restrict__ X, const float* __restrict__ W0,
                                                 const float* __restrict__ s0, const float* __restrict__ b0,
                                                 const float* __restrict__ Wd, const float* __restrict__ sd,
                                                 const float* __restrict__ bd, const float* __restrict__ Wp,
                                                 const float* __restrict__ sp, const float* __restrict__ bp, float* __restrict__ Y,
                                                 const uint4* __restrict__ Wf, int ilOut)
{
    // Wf (r05): the A operands of both GEMMs split into f16 hi / lo fragments ONCE on the host ([conv0 | projection][K step][hi, lo][64 lanes]):
    // four 16-byte loads per GEMM instead of 16 scalar loads + 8 f16 splits per lane and workgroup (a sixth of this kernel's vector instructions)
    constexpr int O = kEnc / 2, NT = 512;
    __shared__ __attribute__((aligned(16))) float sIn[3 * kStemIH * kStemIP];
    __shared__ __attribute__((aligned(16))) float sC[32 * kStemCP];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int ox0 = blockIdx.x * kStemTW, oy0 = blockIdx.y * kStemTH;
    const int cx0 = ox0 - 1, cy0 = oy0 - 1;                     // first conv0 position of the tile
    const int iy0 = 2 * cy0 - 1;                                // first input row (first input column: 2 * cx0 - 1 = LDS column 1)
    const float* Xb = X + (size_t)b * 3 * kEnc * kEnc;
    // r05: every per-workgroup constant is requested HERE, before the window loads: each used to be a global round trip in the middle
    // of the workgroup (the taps after the second barrier, the projection's BN in front of the final stores) with nothing to hide it --
    // the 32 strided scalar loads + 16 splits of the two weight matrices alone were a third of this kernel's time (692 -> 468 us per 128
    // images with host-made fragments, -> 450 with these)
    const int dwc = tid >> 4;                                    // the stencil's channel of this thread
    float wkS[9];
#pragma unroll
    for (int k = 0; k < 9; k++) wkS[k] = Wd[dwc * 9 + k];
    const float dscS = sd[dwc], dshS = bd[dwc];
    float spS[8], bpS[8];                                        // projection BN of this lane's 8 output rows: channels (r & 3) + 8 (r >> 2) + 4 hh
    {
        const int hh_ = (tid & 63) >> 5;
        const float4 a0 = *(const float4*)(sp + 4 * hh_), a1 = *(const float4*)(sp + 8 + 4 * hh_);
        const float4 c0 = *(const float4*)(bp + 4 * hh_), c1 = *(const float4*)(bp + 8 + 4 * hh_);
        spS[0] = a0.x; spS[1] = a0.y; spS[2] = a0.z; spS[3] = a0.w; spS[4] = a1.x; spS[5] = a1.y; spS[6] = a1.z; spS[7] = a1.w;
        bpS[0] = c0.x; bpS[1] = c0.y; bpS[2] = c0.z; bpS[3] = c0.w; bpS[4] = c1.x; bpS[5] = c1.y; bpS[6] = c1.z; bpS[7] = c1.w;
    }
    {   // 1. aligned float4 loads, all of a thread's loads in flight before the first LDS store
        constexpr int Q4 = 18, N4 = 3 * kStemIH * Q4, IT = (N4 + NT - 1) / NT;
        float4 v4[IT]; bool ok[IT]; int dst[IT];
        // item i = tid + NT k -> (plane c, row r, float4 q4): decomposed ONCE (three divisions by constants), then advanced by the
        // constant step NT = DC planes + DR rows + DQ float4s with two carries (r05: the divisions per item were 40 of this kernel's ~750
        // vector instructions per wave)
        constexpr int DC = NT / (kStemIH * Q4), DR = (NT % (kStemIH * Q4)) / Q4, DQ = NT % Q4;
        int c = tid / (kStemIH * Q4), r = (tid / Q4) % kStemIH, q4 = tid % Q4;
#pragma unroll
        for (int k = 0; k < IT; k++) {
            if (k > 0) {
                q4 += DQ; const int cq = q4 >= Q4 ? 1 : 0; q4 -= cq * Q4;
                r += DR + cq; const int cr = r >= kStemIH ? 1 : 0; r -= cr * kStemIH;
                c = min(c + DC + cr, 2);                           // items past the end (not stored) stay inside the three planes
            }
            const int yy = iy0 + r, xx = 2 * ox0 - 4 + 4 * q4;
            ok[k] = yy >= 0 && yy < kEnc && xx >= 0 && xx < kEnc;
            dst[k] = (c * kStemIH + r) * kStemIP + 4 * q4;
            v4[k] = *(const float4*)(Xb + ((size_t)c * kEnc + (ok[k] ? yy : 0)) * kEnc + (ok[k] ? xx : 0));
        }
#pragma unroll
        for (int k = 0; k < IT; k++) asm volatile("" : "+v"(v4[k].x), "+v"(v4[k].y), "+v"(v4[k].z), "+v"(v4[k].w));      // r06: see k_fcn_irb -- the last item's load had been sunk into its store's branch
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < IT; k++)
            if (tid + NT * k < N4) *(float4*)&sIn[dst[k]] = ok[k] ? v4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // 2. conv0 as an MFMA GEMM (split-f16, f32 accumulate like every other convolution here): M = 32 output channels,
    //    K = 27 taps padded to 32 (two K steps), N = the 340 positions in tiles of 32; wave w takes tiles w and w + 8.
    //    The B operand is gathered straight from the LDS window (lane = position, 8 consecutive taps per K step and
    //    lane half); the C/D layout (lane = position, registers = channels) is what the LDS planes want.
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hh = lane >> 5, col = lane & 31;
    {
        auto tap_off = [](int k) { return k < 27 ? ((k / 9) * kStemIH + (k / 3) % 3) * kStemIP + k % 3 : 0; };
        HFrag ah[2], al[2];
#pragma unroll
        for (int st = 0; st < 2; st++) { ah[st].q = Wf[((0 * 2 + st) * 2 + 0) * 64 + lane]; al[st].q = Wf[((0 * 2 + st) * 2 + 1) * 64 + lane]; }
        float sc[16], sh[16];                       // channels (r & 3) + 8 (r >> 2) + 4 hh: four runs of four -> 16-byte loads
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const float4 a = *(const float4*)(s0 + 8 * r4 + 4 * hh), c = *(const float4*)(b0 + 8 * r4 + 4 * hh);
            sc[4 * r4] = a.x; sc[4 * r4 + 1] = a.y; sc[4 * r4 + 2] = a.z; sc[4 * r4 + 3] = a.w;
            sh[4 * r4] = c.x; sh[4 * r4 + 1] = c.y; sh[4 * r4 + 2] = c.z; sh[4 * r4 + 3] = c.w;
        }
        constexpr int NPOS = kStemCW * kStemCH, NTILE = (NPOS + 31) / 32;
        for (int t = wv; t < NTILE; t += 8) {
            const int p = 32 * t + col, pc = min(p, NPOS - 1);
            const int py = pc / kStemCW, px = pc % kStemCW;
            const float* base = sIn + 2 * py * kStemIP + 2 * px + 1;         // window column q = LDS column q + 1
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.f;
#pragma unroll
            for (int st = 0; st < 2; st++) {
                HFrag bh, bl;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int kA = 16 * st + 2 * jj, kB = kA + 8;
                    const float x0 = base[hh ? tap_off(kB) : tap_off(kA)], x1 = base[hh ? tap_off(kB + 1) : tap_off(kA + 1)];
                    split_pair(x0, x1, bh.u[jj], bl.u[jj]);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st].v, bh.v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st].v, bl.v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st].v, bh.v, acc, 0, 0, 0);
            }
            if (p < NPOS) {
                // conv0 positions outside the 256 x 256 map are the depthwise layer's zero padding: only the workgroups on the map's edge have any
                const bool edgeWg = cy0 < 0 || cx0 < 0 || cy0 + kStemCH > O || cx0 + kStemCW > O;      // workgroup-uniform
                if (!edgeWg) {
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        sC[((r & 3) + 8 * (r >> 2) + 4 * hh) * kStemCP + p] = __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[r], sc[r], sh[r]), 0.f, 6.f);
                } else {
                    const bool inside = (unsigned)(cy0 + py) < (unsigned)O && (unsigned)(cx0 + px) < (unsigned)O;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float v = __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[r], sc[r], sh[r]), 0.f, 6.f);
                        sC[((r & 3) + 8 * (r >> 2) + 4 * hh) * kStemCP + p] = inside ? v : 0.f;
                    }
                }
            }
        }
    }
    __syncthreads();
    // 3. depthwise 3x3 + BN + ReLU6: thread = channel x 4 adjacent pixels x 4 rows, rolling window down the rows.  The
    //    results return to LDS in place (plane rows 0..7, columns 0..31) once every thread has finished reading
    {
        const int c = dwc, x4 = (tid & 7) * 4, r0 = ((tid >> 3) & 1) * (kStemTH / 2);
        const float (&wk)[9] = wkS;
        const float dsc = dscS, dsh = dshS;
        const float* plane = sC + c * kStemCP + r0 * kStemCW + x4;      // conv0 position (row r, col x4 + k) = tile pixel (r - 1, x4 + k - 1)
        float win[3][6], o[kStemTH / 2][4];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 6; k++) win[r][k] = plane[r * kStemCW + k];
#pragma unroll
        for (int y = 0; y < kStemTH / 2; y++) {
#pragma unroll
            for (int k = 0; k < 6; k++) win[(y + 2) % 3][k] = plane[(y + 2) * kStemCW + k];
#pragma unroll
            for (int p = 0; p < 4; p++) o[y][p] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int kx = 0; kx < 3; kx++)
#pragma unroll
                    for (int p = 0; p < 4; p++) o[y][p] = __builtin_fmaf(wk[ky * 3 + kx], win[(y + ky) % 3][p + kx], o[y][p]);
        }
        __syncthreads();
        float* dwo = sC + c * kStemCP + r0 * kStemCW + x4;
#pragma unroll
        for (int y = 0; y < kStemTH / 2; y++)
#pragma unroll
            for (int p = 0; p < 4; p++) dwo[y * kStemCW + p] = __builtin_amdgcn_fmed3f(__builtin_fmaf(o[y][p], dsc, dsh), 0.f, 6.f);
    }
    __syncthreads();
    // 4. block 1's projection 32 -> 16 + BN (no activation, no residual), also on MFMA: M = 16 channels (rows 16..31 of
    //    the tile are zero weights), K = 32 hidden channels, N = the tile's 256 pixels: wave w = tile row w
    {
        HFrag ah[2], al[2];
#pragma unroll
        for (int st = 0; st < 2; st++) { ah[st].q = Wf[((1 * 2 + st) * 2 + 0) * 64 + lane]; al[st].q = Wf[((1 * 2 + st) * 2 + 1) * 64 + lane]; }
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = 0.f;
        const float* src = sC + wv * kStemCW + col;
#pragma unroll
        for (int st = 0; st < 2; st++) {
            HFrag bh, bl;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int k0 = 16 * st + 8 * hh + 2 * jj;
                split_pair(src[k0 * kStemCP], src[(k0 + 1) * kStemCP], bh.u[jj], bl.u[jj]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st].v, bh.v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st].v, bl.v, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st].v, bh.v, acc, 0, 0, 0);
        }
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 8; r++) {                       // rows (r & 3) + 8 (r >> 2) + 4 hh < 16
            const int ch = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float v = __builtin_fmaf(acc[r], spS[r], bpS[r]);
            range_note(amax, v);
            // ilOut: rows of all channels interleaved ([y][channel][x]) -- the consumer's window then touches a few contiguous regions instead of one per plane
            Y[(size_t)b * 16 * O * O + (ilOut ? ((size_t)(oy0 + wv) * 16 + ch) * O : ((size_t)ch * O + oy0 + wv) * O) + ox0 + col] = v;
        }
        range_flag(amax);
    }
}

// ---- blocks 2-4: a whole inverted-residual block (expand 1x1 -> depthwise 3x3 -> project 1x1) per launch ----
// Same recipe as k_fcn_stem: the hidden tensor of these blocks (96 x 256 x 256 = 25 MB per image for block 2; 144
// channels for blocks 3 and 4) never exists in HBM.  One workgroup = a 32 x TH output tile (TH = 2 at stride 2, 4 at
// stride 1), 512 threads:
//   A. the input window (CIN channels, the positions the stencil reads) goes to LDS with aligned float4 loads;
//   B. per group of 32 hidden channels:  1. expansion as a split-f16 MFMA GEMM over the window positions (B operand read
//      straight from the LDS window, lane = position), BN + ReLU6, ZERO outside the image (the depthwise layer pads
//      the hidden map, not the input), to LDS planes;  2. depthwise stencil + BN + ReLU6, thread = channel x 4 or 8
//      pixels, to a second LDS tile;  3. projection MFMAs of that K slice accumulate the output tile in registers
//      (waves 0 .. TW*TH/32 - 1, one 32-pixel N tile each);
//   C. BN, residual, store.
// Weights: the expansion's and the projection's A operands come as the f16 hi / lo fragments k_fcn_gemm uses (one
// 16-byte load per lane and fragment, L2-resident; the projection's are fetched at the top of a group and used after two
// barriers, the expansion's one group ahead); depthwise taps and all BN scale / shift pairs sit in an LDS table.
// diagnostic build (tools/build_variant.sh x -DIVF_IRB_TIMING=<input width of the instance: 256 block 2, 128 blocks 3 and 4>): s_memtime sums of wave 0 per
// phase -- [0] window / table loads issued -> landed in LDS (first barrier), [1] B fragments of the window, [2] B1 expansion, [3] wait at the first
// barrier of a group, [4] B2 stencil, [5] wait at the second barrier, [6] B3 projection, [7] epilogue; printed by ivf_fcn_destroy
#ifdef IVF_IRB_TIMING
__device__ unsigned long long g_irbTim[10];
#define IRB_TIM(i) do { if (WI == IVF_IRB_TIMING && S == IVF_IRB_TIMING_S) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    tacc[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define IRB_TIM(i) do { } while (0)
#endif
// output rows per tile of blocks 2 / 3 / 4 (compile-time: tools/build_variant.sh measures others)
#ifndef IVF_IRB_TH2
#define IVF_IRB_TH2 2
#endif
#ifndef IVF_IRB_TH3
#define IVF_IRB_TH3 4
#endif
#ifndef IVF_FCN_HEAD_IL_DEFAULT
#define IVF_FCN_HEAD_IL_DEFAULT 0
#endif
#ifndef IVF_FCN_HEADCHUNK_DEFAULT
#define IVF_FCN_HEADCHUNK_DEFAULT 0
#endif
#ifndef IVF_IRB_TH4
#define IVF_IRB_TH4 4         // r05, measured per 128 images: block 4 at 1 / 2 / 4 rows 387 / 404 / 327 us; block 3 at 2 / 4 / 8 rows 708 / 468 / 569;
#endif                        // block 2 at 2 / 4 rows 610 / 730 (profiles/r05_irb_tile_heights.txt): 2, 4, 4

#ifndef IVF_IRB_REP
#define IVF_IRB_REP 16
#endif
constexpr int kIrbRep = IVF_IRB_REP;          // copies of the A fragments of blocks 2-4 in global memory
template <int S, int CIN, int HID, int COUT, bool RES, int WI, int TH>
__global__ __launch_bounds__(512, 4) void k_fcn_irb(const float* __restrict__ X, const uint4* __restrict__ WqE_,
                                                const float* __restrict__ se, const float* __restrict__ be,
                                                const float* __restrict__ Wd, const float* __restrict__ sd,
                                                const float* __restrict__ bd, const uint4* __restrict__ WqP_,
                                                const float* __restrict__ sp, const float* __restrict__ bp, float* __restrict__ Y,
                                                const float4* __restrict__ tab4, int ilIn, int ilOut)
{
    // ilIn / ilOut: the input / output tensor is row-interleaved ([y][channel][x]) instead of planes ([channel][y][x])
    // tab4 (r05): the LDS image of sT | sPB ([NG * 32][TP] per-channel parameters, then the projection's BN scale[32] | shift[32]) packed ONCE on the
    // host: one 16-byte load per thread, requested together with the window -- the thirteen strided scalar loads per channel and the
    // conditional loads around them were two to three DEPENDENT global round trips in front of the first barrier (phase timers: 19-21k of a
    // workgroup's 54-61k cycles in blocks 2 / 3)
    constexpr int NT = 512, WO = WI / S, TW = 32;
    constexpr int RW = (TW - 1) * S + 3, RH = (TH - 1) * S + 3, NPOS = RW * RH;       // window of hidden / input positions
    constexpr int Q4 = (RW + 3 + 3) / 4, RP = 4 * Q4, XPL = RH * RP + 4;                // input rows start 3 floats left of the window (16-byte aligned)
    constexpr int HPL = (NPOS + 3) / 4 * 4 + 4, DPL = TW * TH + 4;
    constexpr int K16 = (CIN + 15) / 16, NG = (HID + 31) / 32, NTE = (NPOS + 31) / 32, NTP = TW * TH / 32;
    constexpr int TP = 13;                                      // table: expansion scale, shift, 9 taps, depthwise scale, shift
    __shared__ __attribute__((aligned(16))) float sX[CIN * XPL];
    __shared__ __attribute__((aligned(16))) float sH[32 * HPL];
    __shared__ __attribute__((aligned(16))) float sD[32 * DPL];
    constexpr int NTAB = NG * 32 * TP + 64, NTAB4 = NTAB / 4, ITT = (NTAB4 + NT - 1) / NT;
    static_assert(NTAB % 4 == 0, "table is copied in 16-byte pieces");
    __shared__ __attribute__((aligned(16))) float sTP[NTAB];
    float* const sT = sTP;                                      // [NG * 32][TP]
    float* const sPB = sTP + NG * 32 * TP;                      // the projection's BN scale[32] | shift[32]
    const int tid = threadIdx.x, b = blockIdx.z;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const int rx0 = ox0 * S - 1, ry0 = oy0 * S - 1;             // window origin in the input map
    const float* Xb = X + (size_t)b * CIN * WI * WI;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hh = lane >> 5, col = lane & 31;
#ifdef IVF_IRB_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    // r05: every wave of every resident workgroup reads the SAME few KB of A fragments per hidden group -- 512 waves per XCD on the same
    // cache lines, i.e. on a few L2 channels: measured, the wait for fragments requested a whole B2 + B3 earlier was 19k of a workgroup's
    // 61k cycles (block 3).  The host uploads kIrbRep copies; workgroups on one XCD (linear id = XCD mod 8) take different copies.
    const uint4* WqE = WqE_; const uint4* WqP = WqP_;
    {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned rep_ = (lin >> 3) % (unsigned)kIrbRep;
        WqE += (size_t)rep_ * (K16 * NG * 128);
        WqP += (size_t)rep_ * (2 * NG * 128);
    }
    HFrag eh[K16], el[K16];                                     // expansion fragments of the current group
#pragma unroll
    for (int st = 0; st < K16; st++) { eh[st].q = WqE[((st * NG + 0) * 2 + 0) * 64 + lane]; el[st].q = WqE[((st * NG + 0) * 2 + 1) * 64 + lane]; }
    {   // A. aligned float4 loads, all of a thread's loads in flight before the first LDS store
        constexpr int N4 = CIN * RH * Q4, IT = (N4 + NT - 1) / NT;
        float4 v4[IT]; bool ok[IT]; int dst[IT];
        // item i = tid + NT k -> (channel c, row r, float4 q4): decomposed once, then advanced by the constant step NT with two carries
        // (r05, as in k_fcn_stem: block 4's 4-row tile has eight items per thread)
        constexpr int DC = NT / (RH * Q4), DR = (NT % (RH * Q4)) / Q4, DQ = NT % Q4;
        int c = tid / (RH * Q4), r = (tid / Q4) % RH, q4 = tid % Q4;
#pragma unroll
        for (int k = 0; k < IT; k++) {
            if (k > 0) {
                q4 += DQ; const int cq = q4 >= Q4 ? 1 : 0; q4 -= cq * Q4;
                r += DR + cq; const int cr = r >= RH ? 1 : 0; r -= cr * RH;
                c = min(c + DC + cr, CIN - 1);                     // items past the end (not stored) stay inside the tensor
            }
            const int yy = ry0 + r, xx = ox0 * S - 4 + 4 * q4;
            ok[k] = yy >= 0 && yy < WI && xx >= 0 && xx < WI;
            dst[k] = c * XPL + r * RP + 4 * q4;
#ifdef IVF_IRB_ABL_PLANE      // timing-only probe (results wrong): every channel's window rows from plane 0 -- is the load phase bound by how many planes a window touches?
            v4[k] = *(const float4*)(Xb + ((size_t)0 * WI + (ok[k] ? yy : 0)) * WI + (ok[k] ? xx : 0));
#else
            v4[k] = *(const float4*)(Xb + (ilIn ? ((size_t)(ok[k] ? yy : 0) * CIN + c) * WI : ((size_t)c * WI + (ok[k] ? yy : 0)) * WI) + (ok[k] ? xx : 0));
#endif
        }
        float4 t4[ITT];                                         // the parameter table: branch-free clamped loads, in flight with the window
#pragma unroll
        for (int k = 0; k < ITT; k++) t4[k] = tab4[min(tid + NT * k, NTAB4 - 1)];
        // r06: every loaded value passes through an empty asm before the first store.  sched_barrier does not stop the compiler's IR passes from moving a load into the
        // branch of its conditional store: the table's second piece (blocks 3 / 4: ITT = 2, stored by 24 threads) was loaded AFTER the first piece's wait -- a second
        // round trip at the head of every workgroup
#pragma unroll
        for (int k = 0; k < ITT; k++) asm volatile("" : "+v"(t4[k].x), "+v"(t4[k].y), "+v"(t4[k].z), "+v"(t4[k].w));
#pragma unroll
        for (int k = 0; k < IT; k++) asm volatile("" : "+v"(v4[k].x), "+v"(v4[k].y), "+v"(v4[k].z), "+v"(v4[k].w));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < ITT; k++)
            if (tid + NT * k < NTAB4) ((float4*)sTP)[tid + NT * k] = t4[k];
#pragma unroll
        for (int k = 0; k < IT; k++)
            if (tid + NT * k < N4) *(float4*)&sX[dst[k]] = ok[k] ? v4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x16 accO;
#pragma unroll
    for (int q = 0; q < 16; q++) accO[q] = 0.f;
    __syncthreads();
    IRB_TIM(0);
    // the expansion's B operand (the input window, split into f16 hi / lo) is the same for every hidden group: a wave
    // builds the fragments of its (at most TPW) position tiles once
    constexpr int TPW = (NTE + 7) / 8;
    HFrag xh[TPW][K16], xl[TPW][K16];
#pragma unroll
    for (int ti = 0; ti < TPW; ti++) {
        const int nc = min(32 * (wv + 8 * ti) + col, NPOS - 1);
        const float* base = sX + (nc / RW) * RP + nc % RW + 3;
#pragma unroll
        for (int st = 0; st < K16; st++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int k0 = 16 * st + 8 * hh + 2 * jj;
                const float x0 = k0 < CIN ? base[k0 * XPL] : 0.f, x1 = k0 + 1 < CIN ? base[(k0 + 1) * XPL] : 0.f;
                split_pair(x0, x1, xh[ti][st].u[jj], xl[ti][st].u[jj]);
            }
    }
    IRB_TIM(1);
#pragma unroll 1
    for (int g = 0; g < NG; g++) {
        HFrag ph[2], pl[2];                                     // projection fragments of this group: needed two barriers from now
        {   // B1. expansion of hidden channels 32g .. 32g+31 at every window position
            float sc[16], sh[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float* t = sT + (32 * g + (r & 3) + 8 * (r >> 2) + 4 * hh) * TP;
                sc[r] = t[0]; sh[r] = t[1];
            }
#pragma unroll
            for (int ti = 0; ti < TPW; ti++) {
                const int t = wv + 8 * ti;
                if (t >= NTE) break;
                const int n = 32 * t + col, nc = min(n, NPOS - 1);
                const int r_ = nc / RW, q_ = nc % RW;
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; q++) acc[q] = 0.f;
#pragma unroll
                for (int st = 0; st < K16; st++) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(el[st].v, xh[ti][st].v, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh[st].v, xl[ti][st].v, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh[st].v, xh[ti][st].v, acc, 0, 0, 0);
                }
                if (n < NPOS) {
                    // window positions outside the map are the depthwise layer's zero padding: only workgroups on the map's edge have any (r05:
                    // a workgroup-uniform branch instead of a select per stored value)
                    const bool edgeWg = ry0 < 0 || rx0 < 0 || ry0 + RH > WI || rx0 + RW > WI;
                    if (!edgeWg) {
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            sH[((r & 3) + 8 * (r >> 2) + 4 * hh) * HPL + n] = __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[r], sc[r], sh[r]), 0.f, 6.f);
                    } else {
                        const bool inside = (unsigned)(ry0 + r_) < (unsigned)WI && (unsigned)(rx0 + q_) < (unsigned)WI;
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const float v = __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[r], sc[r], sh[r]), 0.f, 6.f);
                            sH[((r & 3) + 8 * (r >> 2) + 4 * hh) * HPL + n] = inside ? v : 0.f;
                        }
                    }
                }
            }
            // r05: the projection's fragments of THIS group are requested here, behind the expansion's MFMAs: requested at the top of the
            // group they made the compiler's wait in front of the first MFMA a vmcnt(0) (a control-flow join loses the count) -- a fresh global
            // round trip at the start of every group, although the expansion's own fragments had been loaded a group earlier
            // (inside `if (wv < NTP)` like their use: loaded unconditionally the compiler SINKS them into B3's block, two barriers down, right in
            // front of the MFMAs that need them.  The conservative vmcnt(0) the join costs now falls on waits that have nothing young to wait for)
#ifdef IVF_IRB_ABL_FRAG        // timing-only probe (results wrong): fragments of group 0 only, loaded once -- what do the per-group fragment loads cost?
            if (g == 0)
#endif
            if (wv < NTP) {
#pragma unroll
                for (int st = 0; st < 2; st++) { ph[st].q = WqP[((2 * g + st) * 2 + 0) * 64 + lane]; pl[st].q = WqP[((2 * g + st) * 2 + 1) * 64 + lane]; }
            }
#ifndef IVF_IRB_ABL_FRAG
            {                                                    // next group's expansion fragments: in flight during B2 / B3 (past the end: group 0's again, unused)
                const int gn = g + 1 < NG ? g + 1 : 0;
#pragma unroll
                for (int st = 0; st < K16; st++) { eh[st].q = WqE[((st * NG + gn) * 2 + 0) * 64 + lane]; el[st].q = WqE[((st * NG + gn) * 2 + 1) * 64 + lane]; }
            }
#endif
        }
        IRB_TIM(2);
        __syncthreads();
        IRB_TIM(3);
        {   // B2. depthwise 3x3 (stride S) + BN + ReLU6 of the group
            const int chl = tid >> 4, sub = tid & 15;
            constexpr int PX = TW * TH / 16;                    // pixels per thread: 2, 4 or 8
            const int oy = sub / (TW / PX), x0 = (sub % (TW / PX)) * PX;
            const float* tb = sT + (32 * g + chl) * TP;
            float wk[9];
#pragma unroll
            for (int k = 0; k < 9; k++) wk[k] = tb[2 + k];
            const float dsc = tb[11], dsh = tb[12];
            const float* hp = sH + chl * HPL + (oy * S) * RW + x0 * S;
            float o[PX];
#pragma unroll
            for (int p = 0; p < PX; p++) o[p] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                float row[(PX - 1) * S + 3];
#pragma unroll
                for (int k = 0; k < (PX - 1) * S + 3; k++) row[k] = hp[ky * RW + k];
#pragma unroll
                for (int kx = 0; kx < 3; kx++)
#pragma unroll
                    for (int p = 0; p < PX; p++) o[p] = __builtin_fmaf(wk[ky * 3 + kx], row[p * S + kx], o[p]);
            }
            float* dp = sD + chl * DPL + oy * TW + x0;
#pragma unroll
            for (int p = 0; p < PX; p++) dp[p] = __builtin_amdgcn_fmed3f(__builtin_fmaf(o[p], dsc, dsh), 0.f, 6.f);
        }
        IRB_TIM(4);
        __syncthreads();
        IRB_TIM(5);
        if (wv < NTP) {   // B3. projection: K slice = this group's 32 hidden channels, N tile = pixels 32 wv .. 32 wv + 31
            const float* src = sD + 32 * wv + col;
#pragma unroll
            for (int st = 0; st < 2; st++) {
                HFrag bh, bl;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int k0 = 16 * st + 8 * hh + 2 * jj;
                    split_pair(src[k0 * DPL], src[(k0 + 1) * DPL], bh.u[jj], bl.u[jj]);
                }
                accO = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl[st].v, bh.v, accO, 0, 0, 0);
                accO = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[st].v, bl.v, accO, 0, 0, 0);
                accO = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[st].v, bh.v, accO, 0, 0, 0);
            }
        }
        IRB_TIM(6);
    }
    if (wv < NTP) {   // C. BN (+ residual), store
        const int n = 32 * wv + col, y = oy0 + n / TW, x = ox0 + n % TW;
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ch = (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (ch >= COUT) continue;
            float v = __builtin_fmaf(accO[r], sPB[ch], sPB[32 + ch]);
            // the residual is the block's own input: the centre of the window this workgroup holds in LDS (stride 1, CIN == COUT) -- r05: it
            // was 16 global loads per lane in front of the final stores, a round trip nothing hid
            if (RES) v += sX[ch * XPL + (n / TW + 1) * RP + (n % TW + 1) + 3];
            range_note(amax, v);
            Y[(size_t)b * COUT * WO * WO + (ilOut ? ((size_t)y * COUT + ch) * WO : ((size_t)ch * WO + y) * WO) + x] = v;
        }
        range_flag(amax);
    }
#ifdef IVF_IRB_TIMING
    if (WI == IVF_IRB_TIMING && S == IVF_IRB_TIMING_S) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IRB_TIM(7);
        if (tid == 0) { for (int i = 0; i < 8; i++) atomicAdd(&g_irbTim[i], tacc[i]); atomicAdd(&g_irbTim[8], 1ull); }
    }
#endif
}

// ---- blocks 5-11 (64 x 64 maps, stride 1, dilation 1 or 2, up to 64 input channels): whole block per launch ----
// k_fcn_irb without the LDS copy of the input window: at 64 input channels the window would not fit beside the hidden
// planes, and it is only ever used as the expansion's B operand -- so every wave gathers the f16 hi / lo fragments of
// its ONE tile of window positions straight from global memory (lane = position: coalesced rows) and keeps them in
// registers for all hidden groups.  Output tile 32 x 2; window (32 + 2 DIL) x (2 + 2 DIL) positions; the depthwise
// taps / BN table of a group is double-buffered in LDS (group g + 1 is fetched during group g).
// Projection: wave w = N tile (w & 1) x output-channel tile (w >> 1).
template <int DIL, int CIN, int HID, int COUT, bool RES>
__global__ __launch_bounds__(512, 4) void k_fcn_irb64(const float* __restrict__ X, const uint4* __restrict__ WqE,
                                                  const float* __restrict__ se, const float* __restrict__ be,
                                                  const float* __restrict__ Wd, const float* __restrict__ sd,
                                                  const float* __restrict__ bd, const uint4* __restrict__ WqP,
                                                  const float* __restrict__ sp, const float* __restrict__ bp, float* __restrict__ Y)
{
    constexpr int NT = 512, WI = 64, TW = 32, TH = 2;
    constexpr int RW = TW + 2 * DIL, RH = TH + 2 * DIL, NPOS = RW * RH;
    constexpr int HPL = (NPOS + 3) / 4 * 4 + 4, DPL = TW * TH + 4;
    constexpr int K16 = (CIN + 15) / 16, NG = (HID + 31) / 32, NTE = (NPOS + 31) / 32, MT = (COUT + 31) / 32, TP = 13;
    static_assert(NTE <= 8, "one window tile per wave");
    static_assert(MT <= 4, "one output-channel tile per wave");
    __shared__ __attribute__((aligned(16))) float sH[32 * HPL];
    __shared__ __attribute__((aligned(16))) float sD[32 * DPL];
    __shared__ float sT[2][32 * TP];
    const int tid = threadIdx.x, b = blockIdx.z;
    const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
    const int rx0 = ox0 - DIL, ry0 = oy0 - DIL;
    const float* Xb = X + (size_t)b * CIN * WI * WI;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, hh = lane >> 5, col = lane & 31;
    // table entry of this thread: expansion scale / shift, 9 taps, depthwise scale / shift per channel.  Fetched one group
    // ahead into a register at the top of a group and stored to LDS at its end, so that nothing waits on the load
    const float* tsrc; int tstride;
    {
        const int f = tid % TP;
        tsrc = f == 0 ? se : f == 1 ? be : f < 11 ? Wd + (f - 2) : f == 11 ? sd : bd;
        tstride = (f >= 2 && f < 11) ? 9 : 1;
    }
    auto fetch_table = [&](int g) {
        const int c = 32 * g + tid / TP;
        return (tid < 32 * TP && c < HID) ? tsrc[(size_t)c * tstride] : 0.f;
    };
    HFrag eh[K16], el[K16];
#pragma unroll
    for (int st = 0; st < K16; st++) { eh[st].q = WqE[((st * NG + 0) * 2 + 0) * 64 + lane]; el[st].q = WqE[((st * NG + 0) * 2 + 1) * 64 + lane]; }
    if (tid < 32 * TP) sT[0][tid] = fetch_table(0);
    // the wave's window tile: position n = 32 wv + col
    const int n = 32 * wv + col, nc = min(n, NPOS - 1);
    const int r_ = nc / RW, q_ = nc % RW;
    const bool inside = (unsigned)(ry0 + r_) < (unsigned)WI && (unsigned)(rx0 + q_) < (unsigned)WI;
    HFrag xh[K16], xl[K16];
    {
        const float* src = Xb + (size_t)(inside ? ry0 + r_ : 0) * WI + (inside ? rx0 + q_ : 0);
        float xv[K16][8];
#pragma unroll
        for (int st = 0; st < K16; st++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 16 * st + 8 * hh + j;
                xv[st][j] = k < CIN ? src[(size_t)k * WI * WI] : 0.f;
            }
#pragma unroll
        for (int st = 0; st < K16; st++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++)
                split_pair(inside ? xv[st][2 * jj] : 0.f, inside ? xv[st][2 * jj + 1] : 0.f, xh[st].u[jj], xl[st].u[jj]);
    }
    f32x16 accO;
#pragma unroll
    for (int q = 0; q < 16; q++) accO[q] = 0.f;
    const int ntP = wv & 1, mtP = wv >> 1;
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < NG; g++) {
        const float* T = sT[g & 1];
        const float tnext = g + 1 < NG ? fetch_table(g + 1) : 0.f;
        if (wv < NTE) {   // B1. expansion of hidden channels 32g .. 32g+31 on the wave's window tile
            // two independent accumulation chains (hi*hi; the two cross terms) instead of one of 3 K16 dependent MFMAs
            f32x16 acc, acc1;
#pragma unroll
            for (int q = 0; q < 16; q++) { acc[q] = 0.f; acc1[q] = 0.f; }
#pragma unroll
            for (int st = 0; st < K16; st++) {
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(el[st].v, xh[st].v, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh[st].v, xh[st].v, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(eh[st].v, xl[st].v, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] += acc1[q];
            if (n < NPOS) {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int cl = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const float v = __builtin_amdgcn_fmed3f(__builtin_fmaf(acc[r], T[cl * TP], T[cl * TP + 1]), 0.f, 6.f);
                    sH[cl * HPL + n] = inside ? v : 0.f;        // the depthwise layer pads the HIDDEN map with zeros
                }
            }
        }
        if (g + 1 < NG) {                                        // next group's fragments and table: in flight during B2 / B3
#pragma unroll
            for (int st = 0; st < K16; st++) { eh[st].q = WqE[((st * NG + g + 1) * 2 + 0) * 64 + lane]; el[st].q = WqE[((st * NG + g + 1) * 2 + 1) * 64 + lane]; }
        }
        __syncthreads();
        {   // B2. depthwise 3x3 (dilation DIL) + BN + ReLU6 of the group: thread = channel x 4 adjacent pixels
            const int chl = tid >> 4, sub = tid & 15;
            const int oy = sub >> 3, x0 = (sub & 7) * 4;
            const float* tb = T + chl * TP;
            float wk[9];
#pragma unroll
            for (int k = 0; k < 9; k++) wk[k] = tb[2 + k];
            const float dsc = tb[11], dsh = tb[12];
            const float* hp = sH + chl * HPL + oy * RW + x0;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                float row[4 + 2 * DIL];
#pragma unroll
                for (int k = 0; k < 4 + 2 * DIL; k++) row[k] = hp[ky * DIL * RW + k];
#pragma unroll
                for (int kx = 0; kx < 3; kx++)
#pragma unroll
                    for (int p = 0; p < 4; p++) o[p] = __builtin_fmaf(wk[ky * 3 + kx], row[p + kx * DIL], o[p]);
            }
            float* dp = sD + chl * DPL + oy * TW + x0;
#pragma unroll
            for (int p = 0; p < 4; p++) dp[p] = __builtin_amdgcn_fmed3f(__builtin_fmaf(o[p], dsc, dsh), 0.f, 6.f);
        }
        if (g + 1 < NG && tid < 32 * TP) sT[(g + 1) & 1][tid] = tnext;     // first read after this barrier (B1 of the next group)
        __syncthreads();
        if (mtP < MT) {   // B3. projection: K slice = this group's 32 hidden channels
            HFrag ph[2], pl[2];
#pragma unroll
            for (int st = 0; st < 2; st++) {
                ph[st].q = WqP[(((2 * g + st) * MT + mtP) * 2 + 0) * 64 + lane];
                pl[st].q = WqP[(((2 * g + st) * MT + mtP) * 2 + 1) * 64 + lane];
            }
            const float* src = sD + 32 * ntP + col;
#pragma unroll
            for (int st = 0; st < 2; st++) {
                HFrag bh, bl;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const int k0 = 16 * st + 8 * hh + 2 * jj;
                    split_pair(src[k0 * DPL], src[(k0 + 1) * DPL], bh.u[jj], bl.u[jj]);
                }
                accO = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl[st].v, bh.v, accO, 0, 0, 0);
                accO = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[st].v, bl.v, accO, 0, 0, 0);
                accO = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[st].v, bh.v, accO, 0, 0, 0);
            }
        }
    }
    if (mtP < MT) {   // C. BN (+ residual), store
        const int m = 32 * ntP + col, y = oy0 + m / TW, x = ox0 + m % TW;
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ch = 32 * mtP + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (ch >= COUT) continue;
            float v = __builtin_fmaf(accO[r], sp[ch], bp[ch]);
            if (RES) v += Xb[((size_t)ch * WI + y) * WI + x];
            range_note(amax, v);
            Y[(((size_t)b * COUT + ch) * WI + y) * WI + x] = v;
        }
        range_flag(amax);
    }
}

// ---- depthwise 3x3 (stride s, dilation d, pad d) + BN + ReLU6 (mobilenet.py:46,54; models_light.py:139-152) ----
// HBM-bound stencil.  One workgroup = one 64 x TH output tile of one channel plane (TH = 64 when the plane is 64 rows,
// else 16): the input window ((TH-1)*s + 2d + 1) x ((64-1)*s + 2d + 1) is staged in LDS once (zero padding materialised there), every
// thread then produces 4 horizontally adjacent outputs and stores them as one float4.
constexpr int kDwTW = 64;
template <int S, int kDwTH>
__global__ __launch_bounds__(256) void k_fcn_dw(const float* __restrict__ X, const float* __restrict__ Wt,
                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                               float* __restrict__ Y, int C, int Hi, int Wi, int Ho, int Wo, int dil)
{
    extern __shared__ float tile[];
    const int bc = blockIdx.z, c = bc % C;
    const int ox0 = blockIdx.x * kDwTW, oy0 = blockIdx.y * kDwTH;
    const int IW = (kDwTW - 1) * S + 2 * dil + 1, IH = (kDwTH - 1) * S + 2 * dil + 1;
    const int IWp = IW | 1;                                   // odd pitch: conflict-free column access
    const int ix0 = ox0 * S - dil, iy0 = oy0 * S - dil;
    const float* I = X + (size_t)bc * Hi * Wi;
    for (int i = threadIdx.x; i < IH * IW; i += 256) {
        const int r = i / IW, q = i % IW;
        const int yy = iy0 + r, xx = ix0 + q;
        tile[r * IWp + q] = (yy >= 0 && yy < Hi && xx >= 0 && xx < Wi) ? I[(size_t)yy * Wi + xx] : 0.f;
    }
    float wk[9];
#pragma unroll
    for (int k = 0; k < 9; k++) wk[k] = Wt[c * 9 + k];
    const float sc = scale[c], sh = shift[c];
    __syncthreads();
    const int tx4 = (threadIdx.x & 15) * 4;
    if (ox0 + tx4 >= Wo) return;
#pragma unroll
    for (int rr = 0; rr < kDwTH / 16; rr++) {
        const int ty = (threadIdx.x >> 4) + 16 * rr;
        const int oy = oy0 + ty;
        if (oy >= Ho) break;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const float* row = tile + (ty * S + ky * dil) * IWp + tx4 * S;
#pragma unroll
            for (int kx = 0; kx < 3; kx++)
#pragma unroll
                for (int p = 0; p < 4; p++) o[p] += wk[ky * 3 + kx] * row[p * S + kx * dil];
        }
#pragma unroll
        for (int p = 0; p < 4; p++) o[p] = fminf(fmaxf(o[p] * sc + sh, 0.f), 6.f);
        float* out = Y + (size_t)bc * Ho * Wo + (size_t)oy * Wo + ox0 + tx4;
        if (ox0 + tx4 + 3 < Wo) *(float4*)out = make_float4(o[0], o[1], o[2], o[3]);
        else for (int p = 0; p < 4 && ox0 + tx4 + p < Wo; p++) out[p] = o[p];
    }
}

// ---- 1x1 (TAPS=1) / dense 3x3 pad 1 (TAPS=9) convolution as an MFMA GEMM ----
//   wave tile: 32*NT output channels x 32*PT pixels; workgroup = 4 waves along the pixel axis.
//   Wq: A fragments, 16 B per lane: index ((((tap*K16 + s) * nTiles + tile) * 2 + part) * 64 + lane), part 0 = hi,
//       1 = lo, holding W[tile*32 + (lane&31)][16*s + 8*(lane>>5) + 0..7][tap] (zero beyond Cout / Cin)
//   act: 0 none, 1 ReLU6, 2 ReLU.  res: optional residual (same shape as Y).
template <int PT> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<4> { typedef float4 T; };
template <int PT> __device__ __forceinline__ float vget(const typename VecT<PT>::T& v, int i);
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <> __device__ __forceinline__ float vget<2>(const float2& v, int i) { return i ? v.y : v.x; }
template <> __device__ __forceinline__ float vget<4>(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

template <int PT, int NT, int TAPS>
__global__ __launch_bounds__(256) void k_fcn_gemm(const float* __restrict__ X, const uint4* __restrict__ Wq,
                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                 const float* __restrict__ res, float* __restrict__ Y,
                                                 int Cin, int Cout, int Hd, int Wd, int nTiles, int act)
{
    typedef typename VecT<PT>::T vec;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = lane >> 5, col = lane & 31;
    const int HW = Hd * Wd;
    const int b = blockIdx.z;
    const int p0 = (blockIdx.x * 4 + wave) * 32 * PT + PT * col;      // first pixel of this lane
    const int ct0 = blockIdx.y * NT;
    const int K16 = (Cin + 15) / 16;
    f32x16 acc[NT][PT];
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int p = 0; p < PT; p++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[n][p][r] = 0.f;
    const float* Xb = X + (size_t)b * Cin * HW;
    const uint4* wq = Wq + (size_t)ct0 * 128 + lane;

    // registers of the step in flight: raw activations of the lane's 8 channels + the A fragments
    vec xb[8];
    uint4 aq[NT][2];
    HFrag bh[PT], bl[PT], ah[NT], al[NT];
    auto consume = [&]() {                          // split the loaded step into f16 operands, freeing xb / aq
#pragma unroll
        for (int p = 0; p < PT; p++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++)
                split_pair(vget<PT>(xb[2 * jj], p), vget<PT>(xb[2 * jj + 1], p), bh[p].u[jj], bl[p].u[jj]);
#pragma unroll
        for (int n = 0; n < NT; n++) { ah[n].q = aq[n][0]; al[n].q = aq[n][1]; }
    };
    auto multiply = [&]() {                         // small terms first; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int p = 0; p < PT; p++) acc[n][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[n].v, bh[p].v, acc[n][p], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int p = 0; p < PT; p++) acc[n][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[n].v, bl[p].v, acc[n][p], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int p = 0; p < PT; p++) acc[n][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[n].v, bh[p].v, acc[n][p], 0, 0, 0);
    };

    if constexpr (TAPS == 1) {
        // channels past Cin are clamped to the last one: their weights are zero and the values finite
        auto load = [&](int s) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int kch = min(16 * s + 8 * kg + j, Cin - 1);
                xb[j] = *(const vec*)(Xb + (size_t)kch * HW + p0);
            }
#pragma unroll
            for (int n = 0; n < NT; n++) {
                aq[n][0] = wq[((size_t)s * nTiles + n) * 128];
                aq[n][1] = wq[((size_t)s * nTiles + n) * 128 + 64];
            }
        };
        load(0);
        // branch-free body: the loads of step s+1 are issued right after step s has been split into f16 operands and
        // stay in flight under its 3*NT*PT MFMAs (the last refill is a redundant re-load)
        for (int s = 0; s < K16; s++) {
            consume();
            load(min(s + 1, K16 - 1));
            __builtin_amdgcn_sched_barrier(0);
            multiply();
        }
    } else {
        static_assert(TAPS == 1 || PT == 1, "the 3x3 path handles one pixel per lane");
        const int py = p0 / Wd, pxx = p0 % Wd;
        int tapN = 0, sN = 0;                       // (tap, step) the next load() fetches
        auto load = [&]() {
            const int yy = py + tapN / 3 - 1, xx = pxx + tapN % 3 - 1;
            const bool ok = yy >= 0 && yy < Hd && xx >= 0 && xx < Wd;
            const int off = ok ? yy * Wd + xx : p0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int kch = min(16 * sN + 8 * kg + j, Cin - 1);
                const float t = Xb[(size_t)kch * HW + off];
                xb[j] = ok ? t : 0.f;
            }
#pragma unroll
            for (int n = 0; n < NT; n++) {
                aq[n][0] = wq[((size_t)(tapN * K16 + sN) * nTiles + n) * 128];
                aq[n][1] = wq[((size_t)(tapN * K16 + sN) * nTiles + n) * 128 + 64];
            }
            if (++sN == K16) { sN = 0; if (tapN < 8) ++tapN; else sN = K16 - 1; }
        };
        load();
        for (int i = 0; i < 9 * K16; i++) {
            consume();
            load();
            __builtin_amdgcn_sched_barrier(0);
            multiply();
        }
    }
    // epilogue: C/D layout col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (output channel).  All loads of
    // a tile (BN scale/shift as float4 per row quad -- the arrays are padded to whole tiles -- and the residual) are
    // issued as one batch before the first use, so the tile costs one memory round trip instead of sixteen.
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const int cb = (ct0 + n) * 32 + 4 * kg;
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4); }
        const size_t ob = ((size_t)b * Cout + cb) * HW + p0;
        vec rv[16];
        if (res) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ro = (r & 3) + 8 * (r >> 2);
                if (cb + ro < Cout) rv[r] = *(const vec*)(res + ob + (size_t)ro * HW);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ro = (r & 3) + 8 * (r >> 2);
            if (cb + ro >= Cout) continue;
            const float sc = vget<4>(sc4[r >> 2], r & 3), sh = vget<4>(sh4[r >> 2], r & 3);
            float o[PT];
#pragma unroll
            for (int p = 0; p < PT; p++) {
                float v = acc[n][p][r] * sc + sh;
                if (act == 1) v = __builtin_amdgcn_fmed3f(v, 0.f, 6.f);
                else if (act == 2) v = fmaxf(v, 0.f);
                if (res) v += vget<PT>(rv[r], p);
                if (act != 1 && !(fabsf(v) < 65504.f)) atomicOr(&g_fcnRange, 1);      // un-clamped output: f16 range guard (never taken in a sane network)
                o[p] = v;
            }
            float* yo = Y + ob + (size_t)ro * HW;
            if constexpr (PT == 4) *(float4*)yo = make_float4(o[0], o[1], o[2], o[3]);
            else if constexpr (PT == 2) *(float2*)yo = make_float2(o[0], o[1]);
            else *yo = o[0];
        }
    }
}

// ---- dense 3x3, pad 1, on a 64-wide map (decoder cbr: 320 -> 80 at 64x64) + BN + ReLU ----
// A wave owns two image rows (128 pixels) and one 32-channel output tile; a lane holds four horizontally adjacent pixels
// of eight input channels, so the 16 lanes of a DPP row span exactly one image row.  Per (dy, K step) the lane loads its
// eight float4 once and splits them into the four pixel-tile B fragments; the dx = -1 / +1 taps re-use the SAME
// fragments shifted by one pixel tile, the edge tile coming from the neighbouring lane by a DPP row shift whose
// out-of-row zero fill is the image's zero padding.  36 MFMAs per 8 loads (k_fcn_gemm<1,3,9>: 9 per 8).
__global__ __launch_bounds__(256) void k_fcn_conv3x3(const float* __restrict__ X, const uint4* __restrict__ Wq,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    float* __restrict__ Y, int Cin, int Cout, int nTiles)
{
    constexpr int HW = 64 * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kg = lane >> 5, col = lane & 31;
    // 1-D grid, renumbered so that all workgroups of an image (8 row groups x nTiles) run on one XCD and share its L2
    const int nwg = gridDim.x, L = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    const int b = L / (8 * nTiles), n = (L / 8) % nTiles;
    const int rp = (L % 8) * 4 + wave;                               // row pair of the image
    const int y = 2 * rp + (col >> 4), x = 4 * (col & 15);
    const int K16 = Cin / 16;
    const float* Xb = X + (size_t)b * Cin * HW + (size_t)8 * kg * HW + x;
    const uint4* wq = Wq + (size_t)n * 128 + lane;
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][r] = 0.f;

    // two register sets P / Q alternate (loop unrolled by two, no copies): a set is refilled for step i+2 right after
    // the MFMAs of step i have been issued, so every load has one whole step (36 MFMAs) to land
    struct Set { float4 x[8]; uint4 a[3][2]; float m; };      // m: 1, or 0 when the tap row is outside the image
    int dyN = 0, sN = 0;
    auto load = [&](Set& S) {
        const int yy = y + dyN - 1;
        const bool ok = yy >= 0 && yy < 64;
        const float* P = Xb + (size_t)16 * sN * HW + (ok ? yy : y) * 64;
        S.m = ok ? 1.f : 0.f;                       // applied when the values are consumed: no wait on the load here
#pragma unroll
        for (int j = 0; j < 8; j++) S.x[j] = *(const float4*)(P + (size_t)j * HW);
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const uint4* w = wq + ((size_t)((dyN * 3 + dx) * K16 + sN) * nTiles) * 128;
            S.a[dx][0] = w[0]; S.a[dx][1] = w[64];
        }
        if (++sN == K16) { sN = 0; if (dyN < 2) ++dyN; else sN = K16 - 1; }     // past the end: redundant re-load
    };
    auto dpp4 = [](const HFrag& f, int shr) {
        HFrag o;
#pragma unroll
        for (int i = 0; i < 4; i++)
            o.u[i] = shr ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.u[i], 0x111, 0xF, 0xF, true)
                         : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.u[i], 0x101, 0xF, 0xF, true);
        return o;
    };
    auto step = [&](Set& S) {
        HFrag bh[6], bl[6];                         // pixel tiles -1 .. 4: [0] = left neighbour's tile 3, [5] = right neighbour's tile 0
#pragma unroll
        for (int pt = 0; pt < 4; pt++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++)
                split_pair(vget<4>(S.x[2 * jj], pt) * S.m, vget<4>(S.x[2 * jj + 1], pt) * S.m, bh[pt + 1].u[jj], bl[pt + 1].u[jj]);
        bh[0] = dpp4(bh[4], 1); bl[0] = dpp4(bl[4], 1);              // row_shr:1  lane c receives lane c-1
        bh[5] = dpp4(bh[1], 0); bl[5] = dpp4(bl[1], 0);              // row_shl:1  lane c receives lane c+1
        HFrag ah[3], al[3];
#pragma unroll
        for (int dx = 0; dx < 3; dx++) { ah[dx].q = S.a[dx][0]; al[dx].q = S.a[dx][1]; }
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
            for (int pt = 0; pt < 4; pt++) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[dx].v, bh[pt + dx].v, acc[pt], 0, 0, 0);
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
            for (int pt = 0; pt < 4; pt++) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[dx].v, bl[pt + dx].v, acc[pt], 0, 0, 0);
#pragma unroll
        for (int dx = 0; dx < 3; dx++)
#pragma unroll
            for (int pt = 0; pt < 4; pt++) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[dx].v, bh[pt + dx].v, acc[pt], 0, 0, 0);
    };
    Set SP, SQ;
    load(SP);
    load(SQ);
    for (int i = 0; i < 3 * K16; i += 2) {          // 3*K16 is even (K16 = 20)
        step(SP);
        __builtin_amdgcn_sched_barrier(0);
        load(SP);
        __builtin_amdgcn_sched_barrier(0);
        step(SQ);
        __builtin_amdgcn_sched_barrier(0);
        load(SQ);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int cb = n * 32 + 4 * kg;
    float4 sc4[4], sh4[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4); }
    float* yb = Y + ((size_t)b * Cout + cb) * HW + y * 64 + x;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int ro = (r & 3) + 8 * (r >> 2);
        if (cb + ro >= Cout) continue;
        const float sc = vget<4>(sc4[r >> 2], r & 3), sh = vget<4>(sh4[r >> 2], r & 3);
        *(float4*)(yb + (size_t)ro * HW) = make_float4(fmaxf(acc[0][r] * sc + sh, 0.f), fmaxf(acc[1][r] * sc + sh, 0.f),
                                                       fmaxf(acc[2][r] * sc + sh, 0.f), fmaxf(acc[3][r] * sc + sh, 0.f));
    }
}

// ---- the same convolution with ALL output-channel tiles in one wave (r02) ----
// k_fcn_conv3x3 gives every 32-channel output tile its own workgroup, so the 320-channel input is read once per tile
// (three times) on top of the three tap rows: 5.6x its size from HBM (PMC), and the kernel ran fetch-bound at 957 us per 128
// images against an MFMA floor of ~400.  Here a wave keeps the accumulators of all NT tiles (NT x 4 x 16 registers) and
// multiplies the SAME B fragments against them: one pass over the input.  A fragments: one register set, each tile's slice
// refilled for the next step right after its MFMAs have been issued (the other tiles' MFMAs cover the load); x: two sets.
template <int NT>
__global__ __launch_bounds__(256, 1) void k_fcn_conv3x3_all(const float* __restrict__ X, const uint4* __restrict__ Wq,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ Y, int Cin, int Cout,
                                                           const float* __restrict__ lastW, float lastBias, float* __restrict__ logits,
                                                           float* __restrict__ part)
{
    // part != nullptr (small batches, r04): 8 workgroups per image fill a thirty-second of the chip at batch 1.  The 3 x K16 (tap row,
    // K step) pairs are cut into gridDim.y contiguous ranges; a workgroup stores its RAW sums into part [gridDim.y][images][Cout][4096],
    // k_fcn_dec_reduce adds them in index order and applies BN + ReLU + conv_last.
    constexpr int HW = 64 * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kg = lane >> 5, col = lane & 31;
    // 1-D grid, renumbered so that the 8 workgroups of an image run on one XCD and share its L2 (tap rows of neighbouring row groups)
    const int nwg = gridDim.x, L = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    const int b = L / 8;
    const int rp = (L % 8) * 4 + wave;                               // row pair of the image
    const int y = 2 * rp + (col >> 4), x = 4 * (col & 15);
    const int K16 = Cin / 16;
    const float* Xb = X + (size_t)b * Cin * HW + (size_t)8 * kg * HW + x;
    const uint4* wq = Wq + lane;
    f32x16 acc[NT][4];
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[n][p][r] = 0.f;
    struct XSet { float4 x[8]; float m; };
    uint4 A[NT][3][2];                                               // [tile][dx][hi, lo] of the CURRENT step
    const int nSteps = 3 * K16 / (int)gridDim.y, i0 = (int)blockIdx.y * nSteps, i1 = i0 + nSteps;     // this workgroup's (tap row, K step) pairs
    int dyN = i0 % 3, sN = i0 / 3;                                   // (tap row, K step) the next x load belongs to
    auto load_x = [&](XSet& S) {
        const int yy = y + dyN - 1;
        const bool ok = yy >= 0 && yy < 64;
        S.m = ok ? 1.f : 0.f;
        const float* P = Xb + (size_t)16 * sN * HW + (ok ? yy : y) * 64;
#pragma unroll
        for (int j = 0; j < 8; j++) S.x[j] = *(const float4*)(P + (size_t)j * HW);
        // r04: the three tap rows of a K step are consecutive steps (step = 3 s + dy).  With the tap row as the OUTER loop a row was
        // re-read a whole sweep over the 320 channels later (26 MB per XCD in between: gone from L2), 3.0x the input from the
        // memory side (profiles/r03_pmc_hbm_traffic.json); now rows y - 1, y, y + 1 of one channel chunk follow each other.
        if (++dyN == 3) { dyN = 0; if (sN < K16 - 1) ++sN; else dyN = 2; }      // past the end: redundant re-load of the last step
    };
    auto load_a = [&](int n, int step) {                             // step = 3 s + dy, clamped at the last one
        step = min(step, 3 * K16 - 1);
        const int dy = step % 3, s_ = step / 3;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const uint4* w = wq + ((size_t)((dy * 3 + dx) * K16 + s_) * NT + n) * 128;
            A[n][dx][0] = w[0]; A[n][dx][1] = w[64];
        }
    };
    auto dpp4 = [](const HFrag& f, int shr) {
        HFrag o;
#pragma unroll
        for (int i = 0; i < 4; i++)
            o.u[i] = shr ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.u[i], 0x111, 0xF, 0xF, true)
                         : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f.u[i], 0x101, 0xF, 0xF, true);
        return o;
    };
    auto step = [&](XSet& S, int stepIdx) {
        HFrag bh[6], bl[6];                         // pixel tiles -1 .. 4: [0] = left neighbour's tile 3, [5] = right neighbour's tile 0
#pragma unroll
        for (int pt = 0; pt < 4; pt++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++)
                split_pair(vget<4>(S.x[2 * jj], pt) * S.m, vget<4>(S.x[2 * jj + 1], pt) * S.m, bh[pt + 1].u[jj], bl[pt + 1].u[jj]);
        bh[0] = dpp4(bh[4], 1); bl[0] = dpp4(bl[4], 1);
        bh[5] = dpp4(bh[1], 0); bl[5] = dpp4(bl[1], 0);
#pragma unroll
        for (int n = 0; n < NT; n++) {
            HFrag ah[3], al[3];
#pragma unroll
            for (int dx = 0; dx < 3; dx++) { ah[dx].q = A[n][dx][0]; al[dx].q = A[n][dx][1]; }
#pragma unroll
            for (int dx = 0; dx < 3; dx++)
#pragma unroll
                for (int pt = 0; pt < 4; pt++) acc[n][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[dx].v, bh[pt + dx].v, acc[n][pt], 0, 0, 0);
#pragma unroll
            for (int dx = 0; dx < 3; dx++)
#pragma unroll
                for (int pt = 0; pt < 4; pt++) acc[n][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[dx].v, bl[pt + dx].v, acc[n][pt], 0, 0, 0);
#pragma unroll
            for (int dx = 0; dx < 3; dx++)
#pragma unroll
                for (int pt = 0; pt < 4; pt++) acc[n][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[dx].v, bh[pt + dx].v, acc[n][pt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_a(n, stepIdx + 1);                 // this tile's A slice of the NEXT step: lands under the other tiles' MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    XSet SP, SQ;
#pragma unroll
    for (int n = 0; n < NT; n++) load_a(n, i0);
    load_x(SP);
    load_x(SQ);
    for (int i = i0; i < i1; i += 2) {              // ranges of an even number of steps (3 * K16 = 60: 1, 2, 3, 5, 6, 10, 15 ranges)
        step(SP, i);
        __builtin_amdgcn_sched_barrier(0);
        load_x(SP);
        __builtin_amdgcn_sched_barrier(0);
        step(SQ, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        load_x(SQ);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (part) {
        float* pb = part + ((size_t)blockIdx.y * (nwg / 8) + b) * Cout * HW + (size_t)y * 64 + x;
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ch = n * 32 + 4 * kg + (r & 3) + 8 * (r >> 2);
                if (ch < Cout) *(float4*)(pb + (size_t)ch * HW) = make_float4(acc[n][0][r], acc[n][1][r], acc[n][2][r], acc[n][3][r]);
            }
        return;
    }
    if (lastW) {
        // decoder: conv_last (1x1, Cout -> 1, + bias; models_light.py:196) folded into the epilogue.  A lane holds 16 channels per
        // tile of its 4 pixels; the other 16 sit in lane ^ 32: the Cout-channel map (1.3 MB per image written and read back) and
        // the k_fcn_last launch disappear
        float lg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int cb = n * 32 + 4 * kg;
            float4 sc4[4], sh4[4], lw4[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4);
#pragma unroll
                for (int i = 0; i < 4; i++) {                           // lastW holds Cout entries only
                    const int ch = cb + 8 * g4 + i;
                    (&lw4[g4].x)[i] = ch < Cout ? lastW[ch] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ro = (r & 3) + 8 * (r >> 2);
                if (cb + ro >= Cout) continue;
                const float sc = vget<4>(sc4[r >> 2], r & 3), sh = vget<4>(sh4[r >> 2], r & 3), lw = vget<4>(lw4[r >> 2], r & 3);
#pragma unroll
                for (int p = 0; p < 4; p++) lg[p] += lw * fmaxf(acc[n][p][r] * sc + sh, 0.f);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) lg[p] += __shfl_xor(lg[p], 32, 64);
        if (kg == 0)
            *(float4*)(logits + (size_t)b * HW + y * 64 + x) = make_float4(lg[0] + lastBias, lg[1] + lastBias, lg[2] + lastBias, lg[3] + lastBias);
        return;
    }
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const int cb = n * 32 + 4 * kg;
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4); }
        float* yb = Y + ((size_t)b * Cout + cb) * HW + y * 64 + x;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ro = (r & 3) + 8 * (r >> 2);
            if (cb + ro >= Cout) continue;
            const float sc = vget<4>(sc4[r >> 2], r & 3), sh = vget<4>(sh4[r >> 2], r & 3);
            *(float4*)(yb + (size_t)ro * HW) = make_float4(fmaxf(acc[n][0][r] * sc + sh, 0.f), fmaxf(acc[n][1][r] * sc + sh, 0.f),
                                                           fmaxf(acc[n][2][r] * sc + sh, 0.f), fmaxf(acc[n][3][r] * sc + sh, 0.f));
        }
    }
}

// ---- the decoder's 3x3 with the CORRECTION products on the block-scaled 6-bit matrix instruction (r06) ----
// k_fcn_conv3x3_all is the one FCN kernel that is bound by the matrix pipe (0.69 busy, r05 counters), and two of its three split-f16 products --
// x_hi w_lo + x_lo w_hi, 2^-11 of the result -- need only ~4 bits.  Two K steps of 16 channels make ONE K = 64 product
//   [x_hi | x_lo 2^10]_s [x_hi | x_lo 2^10]_{s+1}  .  [w_lo 2^13 ; w_hi 2^3]_s [w_lo 2^13 ; w_hi 2^3]_{s+1}  x 2^-13
// on v_mfma_scale_f32_32x32x64_f8f6f4 (A bf6 / e3m2 packed on the host, B fp6 / e2m3 converted here with a power-of-two scale per lane = per
// (pixel, 16 channels) from the block's largest |x|; the instruction's E8M0 operands undo both scales) at ~31 cycles against 4 x 27 for the four
// f16 products it replaces (tools/probe/mfma_fp8_mix.hip): per unit (two K steps, one tap row) 72 f16 + 36 scaled instructions instead of 216 f16.
// Reference error of the scheme against the six goldens, host emulation: + 4e-5 at most (tools/fcn_fp8_corrections.py, profiles/r06_*).
// Geometry, epilogues and the small-batch ranges are k_fcn_conv3x3_all<3>'s; a unit's x window is loaded while the previous unit's MFMAs run, the A
// operands of a (tile, dx) group two groups ahead.  Wq6 (make_gemm): per unit u = 3 s2 + dy, tile n, dx: [hi of step 2 s2][hi of step 2 s2 + 1][bf6
// dwords 0-3][bf6 dwords 4-5 in .x .y], 64 lanes x uint4 each.
#ifndef IVF_DEC_FP6
#define IVF_DEC_FP6 2       // 0: three f16 products (k_fcn_conv3x3_all); 1: k_fcn_conv3x3_f6; 2: k_fcn_conv3x3_f6r where the ranges are whole K-step pairs, k_fcn_conv3x3_f6 elsewhere
#endif
#ifndef IVF_DEC6_ABL
#define IVF_DEC6_ABL 0       // timing-only ablations (results wrong): 1 no x loads in the loop, 2 no A loads, 4 no f16 MFMAs, 8 no scaled MFMAs
#endif
constexpr int kDecSH = 3;                      // weights of the correction operand: w_hi 2^3 (< 16), w_lo 2^13 (<= 4) in e3m2 (largest 28)
__global__ __launch_bounds__(256, 1) void k_fcn_conv3x3_f6(const float* __restrict__ X, const uint4* __restrict__ Wq6,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ Y, int Cin, int Cout,
                                                          const float* __restrict__ lastW, float lastBias, float* __restrict__ logits,
                                                          float* __restrict__ part)
{
    constexpr int HW = 64 * 64, NT = 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kg = lane >> 5, col = lane & 31;
    const int nwg = gridDim.x, L = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    const int b = L / 8;
    const int rp = (L % 8) * 4 + wave;                               // row pair of the image
    const int y = 2 * rp + (col >> 4), x = 4 * (col & 15);
    const int K32 = Cin / 32;
    const float* Xb = X + (size_t)b * Cin * HW + (size_t)8 * kg * HW + x;
    const uint4* wq = Wq6 + lane;
    f32x16 acc[NT][4];
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[n][p][r] = 0.f;
    const int nUnits = 3 * K32 / (int)gridDim.y, u0 = (int)blockIdx.y * nUnits, u1 = u0 + nUnits;     // unit = (pair of K steps s2, tap row dy), u = 3 s2 + dy
    struct XSet2 { float4 x[2][8]; float m; };
    auto load_x = [&](XSet2& S, int u) {
        u = min(u, 3 * K32 - 1);                                     // past the end: a redundant re-load of the last unit
        const int dy = u % 3, s2 = u / 3;
        const int yy = y + dy - 1;
        const bool ok = yy >= 0 && yy < 64;
        S.m = ok ? 1.f : 0.f;
        const float* P = Xb + (size_t)32 * s2 * HW + (ok ? yy : y) * 64;
#pragma unroll
        for (int st = 0; st < 2; st++)
#pragma unroll
            for (int j = 0; j < 8; j++) S.x[st][j] = *(const float4*)(P + (size_t)(16 * st + j) * HW);
    };
    struct AGrp { HFrag h0, h1; uint4 q; uint4 r; };               // A operands of one (tile, dx) group of a unit
    // (r06b measured: the four waves' common A operands staged through LDS -- a quarter fetched per wave, a barrier per unit -- 375 us per 64 images against
    // 257 with every wave loading its own from L2 two groups ahead: the quarter parks in scratch in this 256 + 196 register body and the barrier couples the waves)
    auto load_a = [&](AGrp& a, int u, int g) {                       // g = 3 n + dx
        u = min(u, 3 * K32 - 1);
        const uint4* w = wq + ((size_t)u * 9 + g) * 256;
        a.h0.q = w[0]; a.h1.q = w[64]; a.q = w[128]; a.r = w[192];
    };
    auto dppz = [](int v, int shr, int z) {      // lane - 1 (row_shr:1) / lane + 1 (row_shl:1) within the row of 16 = one image row; zero at the image edge
        return shr ? __builtin_amdgcn_update_dpp(z, v, 0x111, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(z, v, 0x101, 0xF, 0xF, true);
    };
    XSet2 S;
    AGrp ring[3];
    load_x(S, u0);
    load_a(ring[0], u0, 0);
    load_a(ring[1], u0, 1);
    for (int u = u0; u < u1; u++) {
        // ---- this unit's B operands: pixel tiles -1 .. 4 ([0] = left neighbour's tile 3, [5] = right neighbour's tile 0)
        HFrag bh[2][6];
        i32x6 b6[6]; int sbv[6];
#pragma unroll
        for (int pt = 0; pt < 4; pt++) {
            float amax = 0.f;
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int j = 0; j < 8; j++) amax = fmaxf(amax, fabsf(vget<4>(S.x[st][j], pt)));
            amax *= S.m;
            const int sb = min(max(__builtin_amdgcn_frexp_expf(amax) - 3, -40), 20);      // the block's largest |x| / 2^sb in [4, 8) (e2m3: largest 7.5)
            HFrag lo[2];
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
                    split_pair(vget<4>(S.x[st][2 * jj], pt) * S.m, vget<4>(S.x[st][2 * jj + 1], pt) * S.m, bh[st][pt + 1].u[jj], lo[st].u[jj]);
            const f16x8 k1024 = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024};
            const f16x8 l0 = lo[0].v * k1024, l1 = lo[1].v * k1024, h0 = bh[0][pt + 1].v, h1 = bh[1][pt + 1].v;
            const f16x32 src = {h0[0], h0[1], h0[2], h0[3], h0[4], h0[5], h0[6], h0[7], l0[0], l0[1], l0[2], l0[3], l0[4], l0[5], l0[6], l0[7],
                                h1[0], h1[1], h1[2], h1[3], h1[4], h1[5], h1[6], h1[7], l1[0], l1[1], l1[2], l1[3], l1[4], l1[5], l1[6], l1[7]};
            b6[pt + 1] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(src, __builtin_bit_cast(float, (127 + sb) << 23));
            sbv[pt + 1] = 127 + sb;
        }
        // the neighbours' tiles ([0] = left neighbour's tile 3, [5] = right neighbour's tile 0) are NOT kept: 30 registers this one-wave-per-SIMD body does
        // not have beside the next unit's window; the groups with dx = 0 / dx = 2 pull them by DPP when they need them (36 moves per tile instead of 30 per unit)
        __builtin_amdgcn_sched_barrier(0);
        if (!(IVF_DEC6_ABL & 1)) load_x(S, u + 1);      // the window of the next unit: lands under this unit's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 9; g++) {
            const int n = g / 3, dx = g % 3;
            if (!(IVF_DEC6_ABL & 2)) { if (g + 2 < 9) load_a(ring[(g + 2) % 3], u, g + 2); else load_a(ring[(g + 2) % 3], u + 1, g + 2 - 9); }
            const AGrp& a = ring[g % 3];
            const i32x8 a6 = {(int)a.q.x, (int)a.q.y, (int)a.q.z, (int)a.q.w, (int)a.r.x, (int)a.r.y, 0, 0};
            int zz = 0;
            asm volatile("" : "+v"(zz));            // opaque per group: the DPP pulls below are not merged across groups (and kept live)
#pragma unroll
            for (int pt = 0; pt < 4; pt++) {
                const int idx = pt + dx;            // 0 .. 5
                HFrag f0, f1; i32x8 bq; int sbq;
                if (idx >= 1 && idx <= 4) {
                    f0 = bh[0][idx]; f1 = bh[1][idx]; sbq = sbv[idx];
                    bq = i32x8{b6[idx][0], b6[idx][1], b6[idx][2], b6[idx][3], b6[idx][4], b6[idx][5], 0, 0};
                } else {
                    const int srcI = idx == 0 ? 4 : 1, shr = idx == 0 ? 1 : 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) { f0.u[i] = (uint32_t)dppz((int)bh[0][srcI].u[i], shr, zz); f1.u[i] = (uint32_t)dppz((int)bh[1][srcI].u[i], shr, zz); }
                    bq = i32x8{dppz(b6[srcI][0], shr, zz), dppz(b6[srcI][1], shr, zz), dppz(b6[srcI][2], shr, zz), dppz(b6[srcI][3], shr, zz),
                               dppz(b6[srcI][4], shr, zz), dppz(b6[srcI][5], shr, zz), 0, 0};
                    sbq = dppz(sbv[srcI], shr, zz);
                }
                if (!(IVF_DEC6_ABL & 4)) {
                acc[n][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h0.v, f0.v, acc[n][pt], 0, 0, 0);
                acc[n][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h1.v, f1.v, acc[n][pt], 0, 0, 0);
                }
                if (!(IVF_DEC6_ABL & 8))
                acc[n][pt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, bq, acc[n][pt], 3, 2, 0, 127 - 10 - kDecSH, 0, sbq);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (part) {
        float* pb = part + ((size_t)blockIdx.y * (nwg / 8) + b) * Cout * HW + (size_t)y * 64 + x;
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ch = n * 32 + 4 * kg + (r & 3) + 8 * (r >> 2);
                if (ch < Cout) *(float4*)(pb + (size_t)ch * HW) = make_float4(acc[n][0][r], acc[n][1][r], acc[n][2][r], acc[n][3][r]);
            }
        return;
    }
    if (lastW) {                                // conv_last folded into the epilogue (see k_fcn_conv3x3_all)
        float lg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int cb = n * 32 + 4 * kg;
            float4 sc4[4], sh4[4], lw4[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int ch = cb + 8 * g4 + i;
                    (&lw4[g4].x)[i] = ch < Cout ? lastW[ch] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ro = (r & 3) + 8 * (r >> 2);
                if (cb + ro >= Cout) continue;
                const float sc = vget<4>(sc4[r >> 2], r & 3), sh = vget<4>(sh4[r >> 2], r & 3), lw = vget<4>(lw4[r >> 2], r & 3);
#pragma unroll
                for (int p = 0; p < 4; p++) lg[p] += lw * fmaxf(acc[n][p][r] * sc + sh, 0.f);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) lg[p] += __shfl_xor(lg[p], 32, 64);
        if (kg == 0)
            *(float4*)(logits + (size_t)b * HW + y * 64 + x) = make_float4(lg[0] + lastBias, lg[1] + lastBias, lg[2] + lastBias, lg[3] + lastBias);
        return;
    }
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const int cb = n * 32 + 4 * kg;
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4); }
        float* yb = Y + ((size_t)b * Cout + cb) * HW + y * 64 + x;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ro = (r & 3) + 8 * (r >> 2);
            if (cb + ro >= Cout) continue;
            const float sc = vget<4>(sc4[r >> 2], r & 3), sh = vget<4>(sh4[r >> 2], r & 3);
            *(float4*)(yb + (size_t)ro * HW) = make_float4(fmaxf(acc[n][0][r] * sc + sh, 0.f), fmaxf(acc[n][1][r] * sc + sh, 0.f),
                                                           fmaxf(acc[n][2][r] * sc + sh, 0.f), fmaxf(acc[n][3][r] * sc + sh, 0.f));
        }
    }
}

// ---- k_fcn_conv3x3_f6r (r06c): the same arithmetic with the two output rows of a wave sharing their input rows ----
// k_fcn_conv3x3_f6 is bound by its window loads (983 MB per launch through L2 -> L1: every input row is fetched for three tap rows; timing ablations above).  There a
// lane's column is (row of the pair, 4-pixel group), so the lanes of one B fragment sit on two input rows and each (unit, row) loads its own.  Here a lane is a
// TWO-pixel group of ONE row (32 lanes x 2 pixels = the 64 pixels of a row) and the wave keeps the accumulators of both output rows side by side: input row r serves
// output row r with tap row 1 and output row r - 1 with tap row 2 ... -- in unit (s2, dy) output row y0 draws from input row y0 + dy - 1 and output row y0 + 1 from
// y0 + dy: the second is the first of the next unit.  Two fragment sets alternate (by unrolling the three tap rows of a K-step pair); per pair of K steps FOUR input
// rows are loaded, split and converted instead of SIX.  The neighbour pixel of a group comes from the next lane: DPP wave shifts (the 32 lanes of a row span two DPP
// rows), zeroed at the image edge.  Per accumulator the products arrive in the order of k_fcn_conv3x3_f6 with the same operands: the maps are bit-identical.
// Batched form and ranges that are whole K-step pairs (gridDim.y 1, 2, 5, 10); other ranges run k_fcn_conv3x3_f6.
__global__ __launch_bounds__(256, 1) void k_fcn_conv3x3_f6r(const float* __restrict__ X, const uint4* __restrict__ Wq6,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ Y, int Cin, int Cout,
                                                           const float* __restrict__ lastW, float lastBias, float* __restrict__ logits,
                                                           float* __restrict__ part)
{
    constexpr int HW = 64 * 64, NT = 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kg = lane >> 5, col = lane & 31;
    const int nwg = gridDim.x, L = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    const int b = L / 8;
    const int y0 = 2 * ((L % 8) * 4 + wave), x = 2 * col;           // the wave's output rows y0, y0 + 1; this lane's pixels x, x + 1 of both
    const int K32 = Cin / 32;
    const float* Xb = X + (size_t)b * Cin * HW + (size_t)8 * kg * HW + x;
    const uint4* wq = Wq6 + lane;
    f32x16 acc[NT][2][2];                                            // [tile][output row][pixel of the pair]
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[n][p >> 1][p & 1][r] = 0.f;
    const int nUnits = 3 * K32 / (int)gridDim.y, u0 = (int)blockIdx.y * nUnits, u1 = u0 + nUnits;     // whole K-step pairs: u0 % 3 == 0, nUnits % 3 == 0
    struct XRow { float2 x[2][8]; float m; };                       // one input row of a K-step pair: 2 steps x 8 channels x 2 pixels
    auto load_row = [&](XRow& S, int s2, int yy) __attribute__((always_inline)) {
        s2 = min(s2, K32 - 1);
        const bool ok = yy >= 0 && yy < 64;
        S.m = ok ? 1.f : 0.f;
        const float* P = Xb + (size_t)32 * s2 * HW + (ok ? yy : y0) * 64;
#pragma unroll
        for (int st = 0; st < 2; st++)
#pragma unroll
            for (int j = 0; j < 8; j++) S.x[st][j] = *(const float2*)(P + (size_t)(16 * st + j) * HW);
    };
    struct FSet { uint32_t bh[2][2][4]; int b6[2][6]; int sbv[2]; };      // (plain scalars: with HFrag / i32x6 members the sets went through scratch)       // the B operands of one input row: [K step][pixel], fp6 image + scale byte per pixel
    auto build = [&](FSet& F, const XRow& S) __attribute__((always_inline)) {
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
            float amax = 0.f;
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int j = 0; j < 8; j++) amax = fmaxf(amax, fabsf(pt ? S.x[st][j].y : S.x[st][j].x));
            amax *= S.m;
            const int sb = min(max(__builtin_amdgcn_frexp_expf(amax) - 3, -40), 20);
            HFrag lo[2];
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
                    split_pair((pt ? S.x[st][2 * jj].y : S.x[st][2 * jj].x) * S.m, (pt ? S.x[st][2 * jj + 1].y : S.x[st][2 * jj + 1].x) * S.m, F.bh[st][pt][jj], lo[st].u[jj]);
            const f16x8 k1024 = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024};
            HFrag hh0, hh1;
#pragma unroll
            for (int i = 0; i < 4; i++) { hh0.u[i] = F.bh[0][pt][i]; hh1.u[i] = F.bh[1][pt][i]; }
            const f16x8 l0 = lo[0].v * k1024, l1 = lo[1].v * k1024, h0 = hh0.v, h1 = hh1.v;
            const f16x32 src = {h0[0], h0[1], h0[2], h0[3], h0[4], h0[5], h0[6], h0[7], l0[0], l0[1], l0[2], l0[3], l0[4], l0[5], l0[6], l0[7],
                                h1[0], h1[1], h1[2], h1[3], h1[4], h1[5], h1[6], h1[7], l1[0], l1[1], l1[2], l1[3], l1[4], l1[5], l1[6], l1[7]};
            const i32x6 cq = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(src, __builtin_bit_cast(float, (127 + sb) << 23));
#pragma unroll
            for (int i = 0; i < 6; i++) F.b6[pt][i] = cq[i];
            F.sbv[pt] = 127 + sb;
        }
    };
    struct AGrp { HFrag h0, h1; uint4 q; uint4 r; };
    auto load_a = [&](AGrp& a, int u, int g) {
        u = min(u, 3 * K32 - 1);
        const uint4* w = wq + ((size_t)u * 9 + g) * 256;
        a.h0.q = w[0]; a.h1.q = w[64]; a.q = w[128]; a.r = w[192];
    };
    // the pixel left of a lane's pair is the previous lane's second pixel, the one right of it the next lane's first: shifts over the whole wave (a row of the image is two
    // DPP rows), zero where the neighbour would be outside the image (col 0 / col 31: the shift would bring the other channel half's lane, or nothing)
    const int edgeL = col == 0 ? 0 : -1, edgeR = col == 31 ? 0 : -1;
    auto nbL = [&](int v, int z) { return __builtin_amdgcn_update_dpp(z, v, 0x138, 0xF, 0xF, true) & edgeL; };      // wave_shr:1
    auto nbR = [&](int v, int z) { return __builtin_amdgcn_update_dpp(z, v, 0x130, 0xF, 0xF, true) & edgeR; };      // wave_shl:1
    AGrp ring[3];
    // one unit: output row 0 from FA, output row 1 from FB
    auto unit = [&](int u, const FSet& FA, const FSet& FB) __attribute__((always_inline)) {      // (called three times per loop trip: not inlined on its own, and then both sets live in scratch)
#pragma unroll
        for (int g = 0; g < 9; g++) {
            const int n = g / 3, dx = g % 3;
            if (g + 2 < 9) load_a(ring[(g + 2) % 3], u, g + 2); else load_a(ring[(g + 2) % 3], u + 1, g + 2 - 9);
            const AGrp& a = ring[g % 3];
            const i32x8 a6 = {(int)a.q.x, (int)a.q.y, (int)a.q.z, (int)a.q.w, (int)a.r.x, (int)a.r.y, 0, 0};
            int zz = 0;
            asm volatile("" : "+v"(zz));
            auto do_row = [&](auto ROW, const FSet& F) __attribute__((always_inline)) {      // (no `row ? FB : FA`: a select of references keeps both sets in scratch)
                constexpr int row = decltype(ROW)::value;
#pragma unroll
                for (int pt = 0; pt < 2; pt++) {
                    const int idx = pt + dx - 1;    // -1 .. 2: pixel of the pair, or the neighbour lane's
                    HFrag f0, f1; i32x8 bq; int sbq;
                    if (idx == 0 || idx == 1) {
#pragma unroll
                        for (int i = 0; i < 4; i++) { f0.u[i] = F.bh[0][idx][i]; f1.u[i] = F.bh[1][idx][i]; }
                        sbq = F.sbv[idx];
                        bq = i32x8{F.b6[idx][0], F.b6[idx][1], F.b6[idx][2], F.b6[idx][3], F.b6[idx][4], F.b6[idx][5], 0, 0};
                    } else if (idx < 0) {
#pragma unroll
                        for (int i = 0; i < 4; i++) { f0.u[i] = (uint32_t)nbL((int)F.bh[0][1][i], zz); f1.u[i] = (uint32_t)nbL((int)F.bh[1][1][i], zz); }
                        bq = i32x8{nbL(F.b6[1][0], zz), nbL(F.b6[1][1], zz), nbL(F.b6[1][2], zz), nbL(F.b6[1][3], zz), nbL(F.b6[1][4], zz), nbL(F.b6[1][5], zz), 0, 0};
                        sbq = nbL(F.sbv[1], zz);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; i++) { f0.u[i] = (uint32_t)nbR((int)F.bh[0][0][i], zz); f1.u[i] = (uint32_t)nbR((int)F.bh[1][0][i], zz); }
                        bq = i32x8{nbR(F.b6[0][0], zz), nbR(F.b6[0][1], zz), nbR(F.b6[0][2], zz), nbR(F.b6[0][3], zz), nbR(F.b6[0][4], zz), nbR(F.b6[0][5], zz), 0, 0};
                        sbq = nbR(F.sbv[0], zz);
                    }
                    acc[n][row][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h0.v, f0.v, acc[n][row][pt], 0, 0, 0);
                    acc[n][row][pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h1.v, f1.v, acc[n][row][pt], 0, 0, 0);
                    acc[n][row][pt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, bq, acc[n][row][pt], 3, 2, 0, 127 - 10 - kDecSH, 0, sbq);
                }
            };
            do_row(std::integral_constant<int, 0>{}, FA);
            do_row(std::integral_constant<int, 1>{}, FB);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    XRow R0, R1;                                 // raw rows in flight
    FSet F0, F1;
    load_row(R0, u0 / 3, y0 - 1);
    load_row(R1, u0 / 3, y0);
    load_a(ring[0], u0, 0);
    load_a(ring[1], u0, 1);
    for (int u = u0; u < u1; u += 3) {
        const int s2 = u / 3;
        // tap row 0: output row y0 <- input row y0 - 1 (F0), output row y0 + 1 <- input row y0 (F1)
        build(F0, R0); build(F1, R1);
        __builtin_amdgcn_sched_barrier(0);
        load_row(R0, s2, y0 + 1);                // lands under this unit's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        unit(u, F0, F1);
        // tap row 1: y0 <- input row y0 (F1), y0 + 1 <- input row y0 + 1 (into F0)
        build(F0, R0);
        __builtin_amdgcn_sched_barrier(0);
        load_row(R0, s2, y0 + 2);
        __builtin_amdgcn_sched_barrier(0);
        unit(u + 1, F1, F0);
        // tap row 2: y0 <- input row y0 + 1 (F0), y0 + 1 <- input row y0 + 2 (into F1)
        build(F1, R0);
        __builtin_amdgcn_sched_barrier(0);
        load_row(R0, s2 + 1, y0 - 1); load_row(R1, s2 + 1, y0);      // the next K-step pair's first two rows
        __builtin_amdgcn_sched_barrier(0);
        unit(u + 2, F0, F1);
    }
    // ---- epilogues: k_fcn_conv3x3_all's with this kernel's pixel map (row y0 + p / 2, pixels x + (p & 1))
    if (part) {
        float* pb = part + ((size_t)blockIdx.y * (nwg / 8) + b) * Cout * HW + (size_t)y0 * 64 + x;
#pragma unroll
        for (int n = 0; n < NT; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ch = n * 32 + 4 * kg + (r & 3) + 8 * (r >> 2);
                if (ch < Cout) {
                    *(float2*)(pb + (size_t)ch * HW) = make_float2(acc[n][0][0][r], acc[n][0][1][r]);
                    *(float2*)(pb + (size_t)ch * HW + 64) = make_float2(acc[n][1][0][r], acc[n][1][1][r]);
                }
            }
        return;
    }
    if (lastW) {
        float lg[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int cb = n * 32 + 4 * kg;
            float4 sc4[4], sh4[4], lw4[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int ch = cb + 8 * g4 + i;
                    (&lw4[g4].x)[i] = ch < Cout ? lastW[ch] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ro = (r & 3) + 8 * (r >> 2);
                if (cb + ro >= Cout) continue;
                const float sc = vget<4>(sc4[r >> 2], r & 3), sh = vget<4>(sh4[r >> 2], r & 3), lw = vget<4>(lw4[r >> 2], r & 3);
#pragma unroll
                for (int p = 0; p < 4; p++) lg[p >> 1][p & 1] += lw * fmaxf(acc[n][p >> 1][p & 1][r] * sc + sh, 0.f);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) lg[p >> 1][p & 1] += __shfl_xor(lg[p >> 1][p & 1], 32, 64);
        if (kg == 0) {
            *(float2*)(logits + (size_t)b * HW + y0 * 64 + x) = make_float2(lg[0][0] + lastBias, lg[0][1] + lastBias);
            *(float2*)(logits + (size_t)b * HW + (y0 + 1) * 64 + x) = make_float2(lg[1][0] + lastBias, lg[1][1] + lastBias);
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NT; n++) {
        const int cb = n * 32 + 4 * kg;
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4); }
        float* yb = Y + ((size_t)b * Cout + cb) * HW + y0 * 64 + x;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ro = (r & 3) + 8 * (r >> 2);
            if (cb + ro >= Cout) continue;
            const float sc = vget<4>(sc4[r >> 2], r & 3), sh = vget<4>(sh4[r >> 2], r & 3);
            *(float2*)(yb + (size_t)ro * HW) = make_float2(fmaxf(acc[n][0][0][r] * sc + sh, 0.f), fmaxf(acc[n][0][1][r] * sc + sh, 0.f));
            *(float2*)(yb + (size_t)ro * HW + 64) = make_float2(fmaxf(acc[n][1][0][r] * sc + sh, 0.f), fmaxf(acc[n][1][1][r] * sc + sh, 0.f));
        }
    }
}

// second half of the decoder's small-batch schedule: sums of the ranges (index order) -> BN -> ReLU -> conv_last (+ bias) -> logits.
// Workgroup = 16 pixel quads x 16 channel groups (all of a channel's ranges requested together); the groups meet in LDS.
__global__ __launch_bounds__(256) void k_fcn_dec_reduce(const float* __restrict__ part, int nSplit, size_t splitStride, int Cout,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ lastW, float lastBias, float* __restrict__ logits)
{
    constexpr int kMaxSplit = 15;
    __shared__ float4 red[16][16];
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int quad = blockIdx.x * 16 + q, b = blockIdx.y;            // 1024 quads per image
    const int cPer = (Cout + 15) / 16;
    float4 lg = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = g * cPer; c < min((g + 1) * cPer, Cout); c++) {
        const float* p = part + ((size_t)b * Cout + c) * 4096 + 4 * quad;
        float4 v[kMaxSplit];
#pragma unroll
        for (int s = 0; s < kMaxSplit; s++) v[s] = *(const float4*)(p + (size_t)(s < nSplit ? s : 0) * splitStride);
        float4 a = v[0];
#pragma unroll
        for (int s = 1; s < kMaxSplit; s++) if (s < nSplit) { a.x += v[s].x; a.y += v[s].y; a.z += v[s].z; a.w += v[s].w; }
        const float sc = scale[c], sh = shift[c], lw = lastW[c];
        lg.x += lw * fmaxf(a.x * sc + sh, 0.f); lg.y += lw * fmaxf(a.y * sc + sh, 0.f);
        lg.z += lw * fmaxf(a.z * sc + sh, 0.f); lg.w += lw * fmaxf(a.w * sc + sh, 0.f);
    }
    red[g][q] = lg;
    __syncthreads();
    if (g == 0) {
        float4 o = red[0][q];
#pragma unroll
        for (int k = 1; k < 16; k++) { const float4 v = red[k][q]; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; }
        *(float4*)(logits + (size_t)b * 4096 + 4 * quad) = make_float4(o.x + lastBias, o.y + lastBias, o.z + lastBias, o.w + lastBias);
    }
}

// ---- 1x1 expansion (Cin -> 6*Cin) + BN + ReLU6 with the activation tile stationary in LDS ----
// An expansion re-uses every activation for up to 30 output-channel tiles.  k_fcn_gemm fetches and splits the B operand
// once per tile; here a workgroup owns 32*PXT pixels, loads X[Cin][pixels] ONCE, splits it into f16 hi/lo MFMA B
// fragments in LDS (Cin/16 * PXT * 2 KB), and its four waves then walk the output-channel tiles (wave w: tiles w, w+4,
// ...) with A fragments streamed from L2 and B fragments read from LDS: no redundant split VALU, no B re-reads from L2,
// and the kernel is left with its HBM write stream.
template <int PXT, int KS, int RDEPTH = 0>
__global__ __launch_bounds__(256, 2) void k_fcn_expand(const float* __restrict__ X, const uint4* __restrict__ Wq,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ Y, int Cin, int Cout, int HW, int nTiles)
{
    typedef typename VecT<PXT>::T vec;
    extern __shared__ __attribute__((aligned(16))) uint4 sB[];       // [(s*PXT + pt)*2 + part][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kg = lane >> 5, col = lane & 31;
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32 * PXT + PXT * col;                // this lane's PXT adjacent pixels (pixel tile pt = pixel p0+pt)
    const float* Xb = X + (size_t)b * Cin * HW + p0;
    // stage: wave w splits the K steps w, w+4, ...
    for (int s = wave; s < KS; s += 4) {
        vec xb[8];
#pragma unroll
        for (int j = 0; j < 8; j++) xb[j] = *(const vec*)(Xb + (size_t)min(16 * s + 8 * kg + j, Cin - 1) * HW);
#pragma unroll
        for (int pt = 0; pt < PXT; pt++) {
            HFrag h, l;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) split_pair(vget<PXT>(xb[2 * jj], pt), vget<PXT>(xb[2 * jj + 1], pt), h.u[jj], l.u[jj]);
            sB[((s * PXT + pt) * 2 + 0) * 64 + lane] = h.q;
            sB[((s * PXT + pt) * 2 + 1) * 64 + lane] = l.q;
        }
    }
    __syncthreads();
    // A fragments stream from L2 through a ring of RD = KS/2 register slots (slot = step % RD), each refilled right
    // after use with the step RD ahead -- across tile boundaries, clamped at the end.  The refills for the first RD
    // steps of the next tile are therefore issued BEFORE this tile's epilogue stores: on gfx9 loads and stores share
    // the in-order vmcnt, so a load issued behind 16 stores would wait for their write acknowledgements.  The tile
    // body stays branch-free (counted vmcnt waits instead of drains at control-flow joins).
    constexpr int RD = RDEPTH ? KS : KS / 2;       // RDEPTH: a slot per K step -- every refill is for the NEXT tile (see launch_expand)
    static_assert(KS % RD == 0, "ring slots must repeat per tile");
    const int myTiles = (nTiles - wave + 3) / 4;
    uint4 ring[RD][2];
    auto aload = [&](int it, int st, uint4 (&r)[2]) {
#if IVF_EXPAND_ABL & 4
        if (it > 0) return;                         // timing-only: the A fragments of the first tile are reused for every tile
#endif
        const uint4* w = Wq + ((size_t)st * nTiles + wave + 4 * min(it, myTiles - 1)) * 128 + lane;
        r[0] = w[0]; r[1] = w[64];
    };
    if (myTiles > 0) {
#pragma unroll
        for (int d = 0; d < RD; d++) aload(0, d, ring[d]);
    }
    // The epilogue of tile it-1 (BN, ReLU6, 16 row stores) is spread over the K steps of tile it: inside ONE wave the
    // stores then overlap the MFMAs, whatever phase the other wave of the SIMD is in (two waves that both alternate
    // K loop / epilogue fall into lock step: the 160-channel expansion measured 122 us of matrix pipe + 190 us of
    // stores = 312 us, i.e. no overlap at all).  Two accumulator sets alternate; the tile loop is unrolled by two.
    auto kstep = [&](int it, int s, int sl, f32x16 (&acc)[PXT]) {
        HFrag ah, al;
        ah.q = ring[s % RD][0]; al.q = ring[s % RD][1];
        if (s + RD < KS) aload(it, s + RD, ring[s % RD]); else aload(it + 1, s + RD - KS, ring[s % RD]);
        HFrag bh[PXT], bl[PXT];
#pragma unroll
        for (int pt = 0; pt < PXT; pt++) { bh[pt].q = sB[((s * PXT + pt) * 2 + 0) * 64 + sl]; bl[pt].q = sB[((s * PXT + pt) * 2 + 1) * 64 + sl]; }
#if IVF_EXPAND_ABL & 2
#pragma unroll
        for (int pt = 0; pt < PXT; pt++) acc[pt][s & 15] += __builtin_bit_cast(float, al.u[0] ^ bh[pt].u[1] ^ ah.u[2] ^ bl[pt].u[3]);
#else
#pragma unroll
        for (int pt = 0; pt < PXT; pt++) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, bh[pt].v, acc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < PXT; pt++) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bl[pt].v, acc[pt], 0, 0, 0);
#pragma unroll
        for (int pt = 0; pt < PXT; pt++) acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bh[pt].v, acc[pt], 0, 0, 0);
#endif
    };
    auto store_rows = [&](int n, int r0, int r1, const f32x16 (&acc)[PXT], const float4 (&sc4)[4], const float4 (&sh4)[4]) {
        const int cb = n * 32 + 4 * kg;
        float* yb = Y + ((size_t)b * Cout + cb) * HW + p0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r < r0 || r >= r1) continue;
            const int ro = (r & 3) + 8 * (r >> 2);
            if (cb + ro >= Cout) continue;
            const float sc = vget<4>(sc4[r >> 2], r & 3), sh = vget<4>(sh4[r >> 2], r & 3);
            float o[PXT];
#pragma unroll
            for (int p = 0; p < PXT; p++) o[p] = __builtin_amdgcn_fmed3f(acc[p][r] * sc + sh, 0.f, 6.f);
            float* yo = yb + (size_t)ro * HW;
#if IVF_EXPAND_ABL & 1
            asm volatile("" :: "v"(o[0]), "v"(o[PXT - 1]), "v"(yo));
#else
            if constexpr (PXT == 4) *(float4*)yo = make_float4(o[0], o[1], o[2], o[3]);
            else *(float2*)yo = make_float2(o[0], o[1]);
#endif
        }
    };
    auto load_bn = [&](int n, float4 (&sc4)[4], float4 (&sh4)[4]) {
        const int cb = n * 32 + 4 * kg;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4); }
    };
    // one tile: K loop into `cur`, with the epilogue of the previous tile (accumulators `prev`) folded in
    auto tile = [&](int it, f32x16 (&cur)[PXT], const f32x16 (&prev)[PXT], bool havePrev) {
        int sl = lane;                              // opaque per tile: keeps the B fragment reads inside the tile loop
        asm volatile("" : "+v"(sl));                // (hoisted, they would pin KS*PXT*8 registers and spill)
#pragma unroll
        for (int p = 0; p < PXT; p++)
#pragma unroll
            for (int r = 0; r < 16; r++) cur[p][r] = 0.f;
        float4 sc4[4], sh4[4];
        const int nPrev = wave + 4 * (it - 1);
        if (havePrev) load_bn(nPrev, sc4, sh4);
#pragma unroll
        for (int s = 0; s < KS; s++) {
            kstep(it, s, sl, cur);
            if (havePrev) store_rows(nPrev, 16 * s / KS, 16 * (s + 1) / KS, prev, sc4, sh4);
        }
    };
    f32x16 accA[PXT], accB[PXT];
    int it = 0;
    if (myTiles > 0) { tile(0, accA, accB, false); it = 1; }
    for (; it + 1 < myTiles; it += 2) {
        tile(it, accB, accA, true);
        tile(it + 1, accA, accB, true);
    }
    if (it < myTiles) {                             // even tile count: one more tile into B, then B is the last one
        tile(it, accB, accA, true);
        float4 sc4[4], sh4[4];
        load_bn(wave + 4 * it, sc4, sh4);
        store_rows(wave + 4 * it, 0, 16, accB, sc4, sh4);
    } else if (myTiles > 0) {
        float4 sc4[4], sh4[4];
        load_bn(wave + 4 * (myTiles - 1), sc4, sh4);
        store_rows(wave + 4 * (myTiles - 1), 0, 16, accA, sc4, sh4);
    }
}

// ---- fused depthwise 3x3 (stride 1, dilation DIL) + BN + ReLU6  ->  1x1 projection (MFMA) + BN (+ residual) ----
// For the 64x64 stages of the encoder the depthwise output never goes to HBM: a workgroup owns 128 pixels (two image
// rows) and TILES*32 output channels; per 16-channel chunk of the hidden tensor (= one MFMA K step) its 256 threads
// compute the depthwise values (8 horizontally adjacent pixels of one channel per thread, packed-f32 FMAs) into LDS,
// the projection's A fragments of that chunk are staged beside them, and every wave multiplies its own 32 pixels
// against all TILES channel tiles.
//   * a thread loads only its own 8 pixels of the three tap rows; the dilation halo comes from the neighbouring lanes
//     by DPP row shifts (the 16 threads of a channel are one DPP row), which halves the L1 traffic and the registers
//     of a window, so that TWO chunks of loads are kept in flight (register sets A/B, loop unrolled by two);
//   * both LDS stages are double buffered, one barrier per chunk;
//   * workgroups are renumbered so that the row pairs of one image run on the same XCD (shared L2 for halo rows).
// dwP: per hidden channel 12 floats = 9 taps, BN scale, BN shift, pad.
#ifdef IVF_DWPW_TIMING
// diagnostic build only (make TIMING=1): per-phase cycle sums of wave 0 of every workgroup of k_fcn_dwpw<5,4,6,1>
__device__ unsigned long long g_dwpwTim[8];
#define DWPW_TIM(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    tacc[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DWPW_TIM(i) do { } while (0)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ f32x2 pkfma(f32x2 a, float w, f32x2 c) { return __builtin_elementwise_fma(a, (f32x2){w, w}, c); }
// timing-only ablations of k_fcn_dwpw, compile-time so that the product build carries none of it (make EXTRA=-DIVF_DWPW_ABL=<mask>,
// tools/dwpw_ablate.sh): 1 window loads hit chunk 0 (no HBM), 4 no barriers, 8 no A-fragment loads in the loop, 16 no window loads
// in the loop, 32 no MFMAs, 64 no stencil arithmetic, 128 no A-fragment stores to LDS, 256 no B reads from LDS / split, 512 no
// depthwise stores to LDS.  Results are wrong by construction.
#ifndef IVF_DWPW_ABL
#define IVF_DWPW_ABL 0
#endif
#ifndef IVF_EXPAND_ABL
#define IVF_EXPAND_ABL 0      // timing-only ablations of k_fcn_expand: 1 no stores, 2 no MFMAs, 4 no A-fragment loads after the first tile
#endif
constexpr int kAbl = IVF_DWPW_ABL;
#ifndef IVF_DWPW_OCC
#define IVF_DWPW_OCC 2
#endif
#ifndef IVF_DWPW_PACKED
#define IVF_DWPW_PACKED 1     // 0: the stencil's in-thread taps as plain v_fma_f32 instead of v_pk_fma_f32 (tools/probe/issue_model.hip:
                              // packed f32 VALU does not hide under MFMAs; in THIS kernel the stencil and the MFMAs of a workgroup are
                              // separated by barriers anyway, measured 100.7 vs 101.1 us per image, so the packed form stays)
#endif
template <int S> struct DwSetT { float4 own[3][2 * S]; float par; float hl[3], hr[3]; };   // par: parameter (tid & 15) of this thread's channel;
                                                                     // hl / hr: halo pixels across the workgroup edge (256-wide maps only)

template <int TILES, int DIL, int LW, int S_, int NW = 4>    // LW = log2 of the (square) OUTPUT map size: 6, 7 (one row per workgroup),
                                                 // 8 (half a row); S_ = stride of the depthwise layer (input map = S_ x larger);
                                                 // NW = waves = 32-pixel tiles per workgroup (8: four image rows, 64 x 64 maps only)
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : (TILES >= 5 ? IVF_DWPW_OCC : 3)) void k_fcn_dwpw(const float* __restrict__ X, const float* __restrict__ dwP,
                                                    const uint4* __restrict__ Wq, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, const float* __restrict__ res,
                                                    float* __restrict__ Y, int K, int Cout, int nTiles, int abl)
{
    constexpr int NT = 64 * NW, NPX = 32 * NW, TPC = NPX / 8;       // threads, pixels per workgroup; threads per channel of a chunk
    constexpr int kPitch = NPX + 4;                 // floats per hidden channel row in LDS: the pixels + bank skew
                                                    // (8 rows apart = 32 banks apart: the two k-groups never collide)
    static_assert(NW == 4 || (NW == 8 && LW == 6 && S_ == 1), "eight waves: the row-pair layout of the 64 x 64 maps");
    static_assert(LW == 6 || DIL == 1, "the wider maps of the network are not dilated");
    static_assert(S_ == 1 || (S_ == 2 && DIL == 1 && LW <= 7), "stride 2: blocks 2 and 4");
    typedef DwSetT<S_> DwSet;
    constexpr int Wd = 1 << LW, HW = Wd * Wd, WGPI = HW / NPX;      // output map; workgroups per image
    constexpr int Wi = Wd * S_, HWi = Wi * Wi;                       // input map
    __shared__ __attribute__((aligned(16))) float sD[2][16 * kPitch];
    __shared__ __attribute__((aligned(16))) float sW[2][TILES * 512];    // per tile: hi fragment, lo fragment (1 KB each)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 5, col = lane & 31;
    const int nwg = gridDim.x, L = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    const int b = L / WGPI, p128 = L % WGPI;
    const int tile0 = blockIdx.y * TILES;
    const int kc = tid / TPC, g = tid % TPC;
    // PAIR (64-wide stride-1 maps): the workgroup's two rows are y and y + DIL, not y and y + 1.  Their tap rows are
    // y-D, y, y+D and y, y+D, y+2D: four distinct rows instead of six, and the two shared ones travel between the halves
    // of the DPP row (lane ^ 8) instead of being loaded twice.  A thread loads its OTHER row (slot 0: y - D for half A,
    // y + D for half B) and its CENTRE row (slot 1); slot 2 is the partner's centre row (row_ror:8), i.e. tap row 2 for A
    // and tap row 0 for B.  Only the tap weights of slots 0 and 2 depend on the half.
    constexpr bool PAIR = LW == 6 && S_ == 1;
    // (eight waves: rows y, y + D, y + 2D, y + 3D as two such pairs; the rows the pairs share with each other are loaded by both
    // and hit in the vector L1, so the workgroup pulls six tap rows from L2 for four output rows instead of eight, and one set
    // of A fragments for 256 pixels instead of two)
    const bool halfB = PAIR && (g & 8);
    const int yPairA = PAIR ? (p128 / DIL) * (NPX / 64) * DIL + p128 % DIL : 0;       // first row of the workgroup
    const int y = PAIR ? yPairA + (g >> 3) * DIL : (128 * p128 + 8 * g) >> LW;
    const int x0 = PAIR ? (g & 7) * 8 : (128 * p128 + 8 * g) & (Wd - 1);
    int rowOff[3]; float rowM[3];
#pragma unroll
    for (int sl = 0; sl < 3; sl++) {
        const int ky = PAIR ? (sl == 1 ? 1 : ((sl == 0) != halfB ? 0 : 2)) : sl;      // slot -> tap row
        const int yy = y * S_ + (ky - 1) * DIL;     // input row of the slot's tap row
        const bool ok = yy >= 0 && yy < Wi;
        rowM[sl] = ok ? 1.f : 0.f;
        rowOff[sl] = (ok ? yy : y * S_) * Wi + x0 * S_;
    }
    // the DPP row (16 lanes) covers two image rows (LW 6), one (LW 7) or half of one (LW 8).  mL / mR switch the halo taps off
    // where lane-1 / lane+1 is not the horizontal neighbour (image border, or the other row of the pair); for LW 8 the
    // neighbour across the workgroup edge is fetched from memory instead (edgeL / edgeR lanes)
    const float mL = x0 > 0 ? 1.f : 0.f, mR = x0 + 8 < Wd ? 1.f : 0.f;
    const bool edgeL = LW == 8 && g == 0 && x0 > 0, edgeR = LW == 8 && g == 15 && x0 + 8 < Wd;
    const float* Xb = X + (size_t)b * K * HWi;
    const float* Wf = (const float*)Wq;
    const int nChunks = K / 16;                      // odd for the 144-channel block: see the tail after the loop
#ifdef IVF_DWPW_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif

    // window and depthwise-parameter loads run two chunks ahead; each of the 16 threads of a channel fetches ONE of its
    // 12 parameters and the stencil broadcasts them with DPP row_share.  The A fragments (L2-resident, needed only at
    // publish time) run one chunk ahead and are issued first so that waiting for them leaves the rest in flight.
    const int parIdx = kc * 12 + min(g & 15, 11);
    auto issue = [&](DwSet& S, int c) {
        c = min(c, nChunks - 1);                    // refills past the end are redundant re-loads (branch-free loop)
        if (kAbl & 1) c = 0;
        if ((kAbl & 16) && c >= 3) return;
        const float* P = Xb + (size_t)(16 * c + kc) * HWi;
        S.par = dwP[c * 192 + parIdx];
#pragma unroll
        for (int ky = 0; ky < (PAIR ? 2 : 3); ky++) {
#pragma unroll
            for (int j = 0; j < 2 * S_; j++) S.own[ky][j] = *(const float4*)(P + rowOff[ky] + 4 * j);
            if constexpr (LW == 8) {
                const float l = P[rowOff[ky] - (edgeL ? 1 : 0)], r8 = P[rowOff[ky] + (edgeR ? 8 : 0)];
                S.hl[ky] = edgeL ? l : 0.f; S.hr[ky] = edgeR ? r8 : 0.f;
            }
        }
    };
    constexpr int NWREG = (TILES * 256 + NT - 1) / NT;              // float2 of A fragments per thread and chunk
    constexpr bool WTAIL = TILES * 256 % NT != 0;                   // the last round covers half of the threads
    float2 wreg[NWREG];
    auto issue_w = [&](int c) {
        c = min(c, nChunks - 1);
        if ((kAbl & 8) && c >= 2) return;
#pragma unroll
        for (int j = 0; j < NWREG; j++)
            if (!WTAIL || j + 1 < NWREG || tid + NT * j < TILES * 256)
                wreg[j] = *(const float2*)(Wf + ((size_t)c * nTiles + tile0) * 512 + (tid + NT * j) * 2);
    };
    auto shr1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, true)); };
    auto shl1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true)); };
    float o[8];
    auto stencil = [&](const DwSet& S) {
        if (kAbl & 64) {
            o[0] = S.own[1][0].x; o[1] = S.own[1][0].y; o[2] = S.own[1][0].z; o[3] = S.own[1][0].w;
            o[4] = S.own[1][1].x; o[5] = S.own[1][1].y; o[6] = S.own[1][1].z; o[7] = S.own[0][0].x + S.par;
            return;
        }
        float wk[9];
        const int pi = __builtin_bit_cast(int, S.par);
#define ROW_SHARE(k) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, pi, 0x150 + (k), 0xF, 0xF, true))
        wk[0] = ROW_SHARE(0); wk[1] = ROW_SHARE(1); wk[2] = ROW_SHARE(2); wk[3] = ROW_SHARE(3); wk[4] = ROW_SHARE(4);
        wk[5] = ROW_SHARE(5); wk[6] = ROW_SHARE(6); wk[7] = ROW_SHARE(7); wk[8] = ROW_SHARE(8);
        const float dsc = ROW_SHARE(9), dsh = ROW_SHARE(10);
#undef ROW_SHARE
#pragma unroll
        for (int p = 0; p < 8; p++) o[p] = 0.f;
        f32x2 O[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        float4 part[2];
        if constexpr (PAIR) {
            auto ror8 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)); };
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float4 c = S.own[1][j];
                part[j] = make_float4(ror8(c.x), ror8(c.y), ror8(c.z), ror8(c.w));
            }
            // tap weights by slot: slot 0 = tap row 0 (A) / 2 (B), slot 1 = tap row 1, slot 2 = tap row 2 (A) / 0 (B)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const float lo = wk[kx], hi = wk[6 + kx];
                wk[kx] = halfB ? hi : lo; wk[6 + kx] = halfB ? lo : hi;
            }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            // plain FMAs, one per tap and pixel.  A tap that falls outside the thread's own 8 pixels reads the neighbour
            // lane's register through a DPP row shift folded into the FMA itself (v_fmac_f32_dpp, written as inline asm:
            // the compiler's DPP combiner leaves MAC-type instructions alone), so the halo costs no instruction.  Zero padding: rows through the tap weights (rowM),
            // the image's left / right border through the halo taps' weights (mL / mR; the DPP's own zero fill covers
            // the ends of the 16-lane row).
            // PAIR: only slot 0 (the row y - D of half A, y + 2D of half B) can lie outside the image
            const bool masked = !PAIR || ky == 0;
            const float w0 = masked ? wk[ky * 3] * rowM[ky] : wk[ky * 3], w1 = masked ? wk[ky * 3 + 1] * rowM[ky] : wk[ky * 3 + 1],
                        w2 = masked ? wk[ky * 3 + 2] * rowM[ky] : wk[ky * 3 + 2];
            const float w0L = w0 * mL, w2R = w2 * mR;
            if constexpr (IVF_DWPW_PACKED && PAIR && (DIL == 2 || DIL == 4)) {
                // packed-f32 FMAs (v_pk_fma_f32: two pixels per instruction, full rate) for every tap that stays inside the
                // thread's 8 pixels; the halo taps stay DPP FMAs.  Per pixel the order centre, left, right is that of the scalar path.
                const float4 a = ky == 2 ? part[0] : S.own[ky][0], c4 = ky == 2 ? part[1] : S.own[ky][1];
                const float own[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
                const f32x2 A[4] = {{a.x, a.y}, {a.z, a.w}, {c4.x, c4.y}, {c4.z, c4.w}};
                constexpr int H = DIL / 2;
#pragma unroll
                for (int i = 0; i < 4; i++) O[i] = pkfma(A[i], w1, O[i]);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (i - H >= 0) O[i] = pkfma(A[i - H], w0, O[i]);
                    else {
                        float t0 = O[i].x, t1 = O[i].y;
                        asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(t0) : "v"(own[8 + 2 * i - DIL]), "v"(w0L));
                        asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(t1) : "v"(own[9 + 2 * i - DIL]), "v"(w0L));
                        O[i].x = t0; O[i].y = t1;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (i + H < 4) O[i] = pkfma(A[i + H], w2, O[i]);
                    else {
                        float t0 = O[i].x, t1 = O[i].y;
                        asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(t0) : "v"(own[2 * i + DIL - 8]), "v"(w2R));
                        asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(t1) : "v"(own[2 * i + 1 + DIL - 8]), "v"(w2R));
                        O[i].x = t0; O[i].y = t1;
                    }
                }
                continue;
            }
            if constexpr (S_ == 2) {
                // stride 2: output pixel p reads input columns 2p-1, 2p, 2p+1 of the thread's 16; only column -1 of p = 0
                // belongs to the left neighbour (its column 15)
                float in16[16];
#pragma unroll
                for (int j = 0; j < 4; j++) { const float4 v = S.own[ky][j]; in16[4 * j] = v.x; in16[4 * j + 1] = v.y; in16[4 * j + 2] = v.z; in16[4 * j + 3] = v.w; }
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    o[p] = __builtin_fmaf(in16[2 * p], w1, o[p]);
                    o[p] = __builtin_fmaf(in16[2 * p + 1], w2, o[p]);
                    if (p > 0) o[p] = __builtin_fmaf(in16[2 * p - 1], w0, o[p]);
                    else asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p]) : "v"(in16[15]), "v"(w0L));
                }
                (void)w2R;
                continue;
            }
            const float4 a = (PAIR && ky == 2) ? part[0] : S.own[ky][0], c4 = (PAIR && ky == 2) ? part[1] : S.own[ky][1];
            const float own[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int p = 0; p < 8; p++) {
                o[p] = __builtin_fmaf(own[p], w1, o[p]);
                if (p - DIL >= 0) o[p] = __builtin_fmaf(own[p - DIL], w0, o[p]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p]) : "v"(own[8 + p - DIL]), "v"(w0L));
                if (p + DIL < 8) o[p] = __builtin_fmaf(own[p + DIL], w2, o[p]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p]) : "v"(own[p + DIL - 8]), "v"(w2R));
            }
            if constexpr (LW == 8) {          // the neighbour pixel across the workgroup edge (zero in every other lane)
                o[0] = __builtin_fmaf(S.hl[ky], w0, o[0]);
                o[7] = __builtin_fmaf(S.hr[ky], w2, o[7]);
            }
        }
        if constexpr (IVF_DWPW_PACKED && PAIR && (DIL == 2 || DIL == 4)) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const f32x2 v = pkfma(O[i], dsc, (f32x2){dsh, dsh});
                o[2 * i] = __builtin_amdgcn_fmed3f(v.x, 0.f, 6.f); o[2 * i + 1] = __builtin_amdgcn_fmed3f(v.y, 0.f, 6.f);
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < 8; p++) o[p] = __builtin_amdgcn_fmed3f(__builtin_fmaf(o[p], dsc, dsh), 0.f, 6.f);
    };
    auto publish = [&](int buf) {
        float* dst = &sD[buf][kc * kPitch + 8 * g];
        if (!(kAbl & 512)) {
            *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
            *(float4*)(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
        } else asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]));
        if (kAbl & 128) { asm volatile("" :: "v"(wreg[0].x), "v"(wreg[NWREG - 1].y)); return; }
#pragma unroll
        for (int j = 0; j < NWREG; j++)
            if (!WTAIL || j + 1 < NWREG || tid + NT * j < TILES * 256) *(float2*)&sW[buf][(tid + NT * j) * 2] = wreg[j];
    };

    f32x16 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
    auto multiply = [&](int cur) {
        // B operand: this lane's pixel, hidden channels 8*kg .. 8*kg+7 of the chunk, split into f16 hi / lo
        HFrag bh, bl;
        const float* dB = &sD[cur][8 * kg * kPitch + 32 * wave + col];
        if (kAbl & 256) {
#pragma unroll
            for (int jj = 0; jj < 4; jj++) { bh.u[jj] = 0x3c003c00u + cur + lane; bl.u[jj] = 0x1c001c00u + cur; }
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; jj++) split_pair(dB[2 * jj * kPitch], dB[(2 * jj + 1) * kPitch], bh.u[jj], bl.u[jj]);
        }
        DWPW_TIM(5);
        const uint4* wA = (const uint4*)&sW[cur][0] + lane;
        if (kAbl & 32) {
            acc[0][0] += __builtin_bit_cast(float, bh.u[0] ^ bl.u[1]) + __builtin_bit_cast(float, bh.u[2] ^ bl.u[3]);
            acc[0][1] += __builtin_bit_cast(float, bh.u[1] ^ bl.u[0]) + __builtin_bit_cast(float, bh.u[3] ^ bl.u[2]);
            return;
        }
#pragma unroll
        for (int t = 0; t < TILES; t += 2) {        // two tiles at a time: consecutive MFMAs hit different accumulators
            HFrag ah[2], al[2];
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (t + u < TILES) { ah[u].q = wA[(t + u) * 128]; al[u].q = wA[(t + u) * 128 + 64]; }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (t + u < TILES) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u].v, bh.v, acc[t + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (t + u < TILES) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u].v, bl.v, acc[t + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (t + u < TILES) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u].v, bh.v, acc[t + u], 0, 0, 0);
        }
    };

    DwSet SA, SB;
    issue(SA, 0);
    issue_w(0);
    issue(SB, 1);
    stencil(SA);
    publish(0);
    issue_w(1);                                     // weights before the window: in-order return lets the next
    issue(SA, 2);                                   // stencil wait for them with the window loads still in flight
    __syncthreads();
#ifdef IVF_DWPW_TIMING
    tlast = __builtin_amdgcn_s_memtime();
#endif
    for (int c = 0; c + 1 < nChunks; c += 2) {
        // buffer 0 holds chunk c; set B = window of chunk c+1, set A = window of chunk c+2 (both in flight)
        multiply(0);
        DWPW_TIM(0);
        stencil(SB);
        DWPW_TIM(1);
        publish(1);
        DWPW_TIM(4);
        __builtin_amdgcn_sched_barrier(0);          // pin the issue order: A fragments, then the far-ahead window
        issue_w(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        issue(SB, c + 3);
        __builtin_amdgcn_sched_barrier(0);
        DWPW_TIM(2);
        if (!(kAbl & 4)) __syncthreads();
        DWPW_TIM(3);
        multiply(1);
        DWPW_TIM(0);
        stencil(SA);
        DWPW_TIM(1);
        publish(0);
        DWPW_TIM(4);
        __builtin_amdgcn_sched_barrier(0);          // pin the issue order: A fragments, then the far-ahead window
        issue_w(c + 3);
        __builtin_amdgcn_sched_barrier(0);
        issue(SA, c + 4);
        __builtin_amdgcn_sched_barrier(0);
        DWPW_TIM(2);
        if (!(kAbl & 4)) __syncthreads();
        DWPW_TIM(3);
    }
#ifdef IVF_DWPW_TIMING
    if (TILES == 5 && DIL == 4 && tid == 0) {
        for (int i = 0; i < 6; i++) atomicAdd(&g_dwpwTim[i], tacc[i]);
        atomicAdd(&g_dwpwTim[7], 1ull);
    }
#endif
    if (nChunks & 1) multiply(0);                   // odd chunk count (the 144-channel block): the last chunk sits in buffer 0.
                                                    // A peeled tail, not an exit inside the loop: that costs the loop its
                                                    // counted vmcnt waits (measured on the 64-wide kernels: 226 -> 357 us)
    // epilogue: per tile, BN scale/shift (float4 per row quad, arrays padded to whole tiles) and the residual are
    // loaded as one batch before the first use
    const int pix = PAIR ? (yPairA + (wave >> 1) * DIL) * Wd + 32 * (wave & 1) + col : 128 * p128 + 32 * wave + col;   // wave = pixel tile
    float amaxOut = 0.f;
#pragma unroll
    for (int t = 0; t < TILES; t++) {
        const int cb = (tile0 + t) * 32 + 4 * kg;
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4); }
        const size_t ob = ((size_t)b * Cout + cb) * HW + pix;
        float rv[16];
        if (res) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int ro = (q & 3) + 8 * (q >> 2);
                if (cb + ro < Cout) rv[q] = res[ob + (size_t)ro * HW];
            }
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int ro = (q & 3) + 8 * (q >> 2);
            if (cb + ro >= Cout) continue;
            float v = acc[t][q] * vget<4>(sc4[q >> 2], q & 3) + vget<4>(sh4[q >> 2], q & 3);
            if (res) v += rv[q];
            range_note(amaxOut, v);
            Y[ob + (size_t)ro * HW] = v;
        }
    }
    range_flag(amaxOut);
}

// ---- k_fcn_dwpw with eight waves per workgroup (64 x 64 maps, stride 1) ----
// k_fcn_dwpw<5, 4, 6, 1> covers 160 output channels per workgroup, so block 17 (320) ran it twice per pixel tile (grid
// y = 2): the hidden tensor was read and the stencil computed twice.  Here a workgroup has EIGHT waves on the same 128 pixels
// and splits the TT output tiles between two wave sets (each wave reads only its set's A fragments from LDS):
// all 512 threads share the stencil of a 16-channel chunk (channel = tid >> 5, 4 adjacent pixels per thread; a DPP row of
// 16 lanes is exactly one 64-pixel image row, so the dilation halo comes from the neighbour lanes and the DPP zero fill IS
// the zero padding), wave w multiplies pixel tile (w & 3) against output tiles 5 (w >> 2) .. 5 (w >> 2) + 4.
// The body is instantiated once per wave set (NTW = output tiles of the set, TS = its index) so that the tile count of a
// wave is a compile-time constant: a run-time count costs the MFMA schedule (725 -> 833 us at TT = 10).
template <int DIL, int TT, int NTW, int TS>
__device__ __forceinline__ void dwpw8_body(float (&sD)[2][16 * 132], float* sWp, const float* __restrict__ X, const float* __restrict__ dwP,
                                           const uint4* __restrict__ Wq, const float* __restrict__ scale,
                                           const float* __restrict__ shift, const float* __restrict__ res,
                                           float* __restrict__ Y, int K, int Cout, int nTiles)
{
    static_assert(DIL == 1 || DIL == 2 || DIL == 4, "a tap at distance DIL is in this lane or the next one");
    static_assert(NTW >= 1, "every wave set owns at least one output tile");
    constexpr int kPitch = 132, Wd = 64, HW = Wd * Wd, WGPI = HW / 128, TILES = NTW, T0 = (TT + 1) / 2;     // set 0 holds tiles 0 .. T0-1
    float (&sW)[2][TT * 512] = *reinterpret_cast<float (*)[2][TT * 512]>(sWp);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 5, col = lane & 31;
    const int nwg = gridDim.x, L = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;
    const int b = L / WGPI, p128 = L % WGPI;
    const int kc = tid >> 5, g = tid & 31;                      // channel of the chunk; 32 threads x 4 pixels = two image rows
    const int y = 2 * p128 + (g >> 4), x0 = (g & 15) * 4;
    int rowOff[3]; float rowM[3];
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
        const int yy = y + (ky - 1) * DIL;
        const bool ok = yy >= 0 && yy < Wd;
        rowM[ky] = ok ? 1.f : 0.f;
        rowOff[ky] = (ok ? yy : y) * Wd + x0;
    }
    const float* Xb = X + (size_t)b * K * HW;
    const float* Wf = (const float*)Wq;
    const int nChunks = K / 16;                                 // even for every 64 x 64 block (12, 24, 36, 60)
    struct Win { float4 r[3]; float par; };
    const int parIdx = kc * 12 + min(g & 15, 11);               // the 16 lanes of a DPP row fetch the channel's 12 parameters
    auto issue = [&](Win& S, int c) {
        c = min(c, nChunks - 1);
        const float* P = Xb + (size_t)(16 * c + kc) * HW;
        S.par = dwP[c * 192 + parIdx];
#pragma unroll
        for (int ky = 0; ky < 3; ky++) S.r[ky] = *(const float4*)(P + rowOff[ky]);
    };
    constexpr int NW = (TT * 256 + 511) / 512, NWF = TT * 256 / 512;   // float2 per thread per chunk: NWF full rounds (+ a half one, threads 0..255)
    float2 wreg[NW];
    auto issue_w = [&](int c) {
        c = min(c, nChunks - 1);
#pragma unroll
        for (int j = 0; j < NW; j++)
            if (j < NWF || TS == 0) wreg[j] = *(const float2*)(Wf + (size_t)c * nTiles * 512 + (tid + 512 * j) * 2);      // the half round: set 0's 256 threads
    };
    float o[4];
    auto stencil = [&](const Win& S) {
        const int pi = __builtin_bit_cast(int, S.par);
#define ROW_SHARE(k) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, pi, 0x150 + (k), 0xF, 0xF, true))
        float wk[9];
        wk[0] = ROW_SHARE(0); wk[1] = ROW_SHARE(1); wk[2] = ROW_SHARE(2); wk[3] = ROW_SHARE(3); wk[4] = ROW_SHARE(4);
        wk[5] = ROW_SHARE(5); wk[6] = ROW_SHARE(6); wk[7] = ROW_SHARE(7); wk[8] = ROW_SHARE(8);
        const float dsc = ROW_SHARE(9), dsh = ROW_SHARE(10);
#undef ROW_SHARE
#pragma unroll
        for (int p = 0; p < 4; p++) o[p] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const float w0 = wk[ky * 3] * rowM[ky], w1 = wk[ky * 3 + 1] * rowM[ky], w2 = wk[ky * 3 + 2] * rowM[ky];
            const float own[4] = {S.r[ky].x, S.r[ky].y, S.r[ky].z, S.r[ky].w};
#pragma unroll
            for (int p = 0; p < 4; p++) {
                o[p] = __builtin_fmaf(own[p], w1, o[p]);
                // a tap outside the thread's 4 pixels is a pixel of the left / right neighbour lane (zero past the ends of the
                // image row: the DPP row IS the image row)
                if (p - DIL >= 0) o[p] = __builtin_fmaf(own[p - DIL], w0, o[p]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p]) : "v"(own[4 + p - DIL]), "v"(w0));
                if (p + DIL < 4) o[p] = __builtin_fmaf(own[p + DIL], w2, o[p]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p]) : "v"(own[p + DIL - 4]), "v"(w2));
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) o[p] = __builtin_amdgcn_fmed3f(__builtin_fmaf(o[p], dsc, dsh), 0.f, 6.f);
    };
    auto publish = [&](int buf) {
        *(float4*)&sD[buf][kc * kPitch + 4 * g] = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int j = 0; j < NW; j++)
            if (j < NWF || TS == 0) *(float2*)&sW[buf][(tid + 512 * j) * 2] = wreg[j];
    };
    f32x16 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[t][q] = 0.f;
    const int pt = wave & 3;                                    // pixel tile
    constexpr int ts = TS;
    auto multiply = [&](int cur) {
        HFrag bh, bl;
        const float* dB = &sD[cur][8 * kg * kPitch + 32 * pt + col];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) split_pair(dB[2 * jj * kPitch], dB[(2 * jj + 1) * kPitch], bh.u[jj], bl.u[jj]);
        const uint4* wA = (const uint4*)&sW[cur][0] + (size_t)T0 * ts * 128 + lane;
#pragma unroll
        for (int t = 0; t < TILES; t += 2) {
            HFrag ah[2], al[2];
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (t + u < TILES) { ah[u].q = wA[(t + u) * 128]; al[u].q = wA[(t + u) * 128 + 64]; }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (t + u < TILES) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[u].v, bh.v, acc[t + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (t + u < TILES) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u].v, bl.v, acc[t + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (t + u < TILES) acc[t + u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u].v, bh.v, acc[t + u], 0, 0, 0);
        }
    };
    Win SA, SB;
    issue(SA, 0);
    issue_w(0);
    issue(SB, 1);
    stencil(SA);
    publish(0);
    issue_w(1);
    issue(SA, 2);
    __syncthreads();
    for (int c = 0; c + 1 < nChunks; c += 2) {
        multiply(0);
        stencil(SB);
        publish(1);
        __builtin_amdgcn_sched_barrier(0);
        issue_w(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        issue(SB, c + 3);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        multiply(1);
        stencil(SA);
        publish(0);
        __builtin_amdgcn_sched_barrier(0);
        issue_w(c + 3);
        __builtin_amdgcn_sched_barrier(0);
        issue(SA, c + 4);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    const int pix = 128 * p128 + 32 * pt + col;
    float amaxOut = 0.f;
#pragma unroll
    for (int t = 0; t < TILES; t++) {
        const int cb = (T0 * ts + t) * 32 + 4 * kg;
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(scale + cb + 8 * g4); sh4[g4] = *(const float4*)(shift + cb + 8 * g4); }
        const size_t ob = ((size_t)b * Cout + cb) * HW + pix;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int ro = (q & 3) + 8 * (q >> 2);
            if (cb + ro >= Cout) continue;
            float v = acc[t][q] * vget<4>(sc4[q >> 2], q & 3) + vget<4>(sh4[q >> 2], q & 3);
            if (res) v += res[ob + (size_t)ro * HW];
            range_note(amaxOut, v);
            Y[ob + (size_t)ro * HW] = v;
        }
    }
    range_flag(amaxOut);
}

template <int DIL, int TT>
__global__ __launch_bounds__(512, 2) void k_fcn_dwpw8(const float* __restrict__ X, const float* __restrict__ dwP,
                                                     const uint4* __restrict__ Wq, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const float* __restrict__ res,
                                                     float* __restrict__ Y, int K, int Cout, int nTiles)
{
    __shared__ __attribute__((aligned(16))) float sD[2][16 * 132];
    __shared__ __attribute__((aligned(16))) float sW[2 * TT * 512];
    if (threadIdx.x < 256) dwpw8_body<DIL, TT, (TT + 1) / 2, 0>(sD, sW, X, dwP, Wq, scale, shift, res, Y, K, Cout, nTiles);
    else dwpw8_body<DIL, TT, TT / 2, 1>(sD, sW, X, dwP, Wq, scale, shift, res, Y, K, Cout, nTiles);
}

// ---- blocks 15 - 17 as ONE kernel each: expand 160 -> 960, depthwise 3x3 dilation 4, project 960 -> 160 / 320 (+ residual) ----
// (IF/networks/models_light/mobilenet.py:35-64 with the dilation of models_light.py:139-152.)  The 960-channel hidden tensors of
// these blocks (15.7 MB per image, written by k_fcn_expand and read back by k_fcn_dwpw*: 94 MB per image for the three blocks)
// never exist: a depthwise 3x3 with DILATION 4 on a 64 x 64 map only ever combines pixels of the same residue (y mod 4, x mod 4),
// i.e. it is sixteen independent plain 3x3 convolutions on 16 x 16 sub-images with zero padding -- no halo between them.  One
// workgroup owns one sub-image (256 pixels) of one image for all 960 hidden channels:
//   * the input tile X[160][256] is gathered once and kept in REGISTERS as split-f16 B fragments (80 VGPRs per lane);
//   * per group of 16 hidden channels (60 groups): E = expansion on v_mfma_f32_16x16x32_f16 (M = the 16 hidden channels) + BN +
//     ReLU6 -> LDS planes; S = the 3x3 stencil on the 16 x 16 planes (thread = channel x sub-row x half row, halo pixel by DPP) +
//     BN + ReLU6 -> LDS; P = projection on v_mfma_f32_32x32x16_f16 (the group is one K step) into 5 resident accumulator tiles;
//   * software pipeline over the groups, ONE barrier per interval: E(g), S(g - 1), P(g - 2) work on different LDS buffers;
//   * the weights of an interval (10 KB expansion + 10 KB projection fragments + 768 B of depthwise / BN parameters) arrive by
//     LDS-DMA (global_load_lds_dwordx4), no registers, no VALU.  A piece of 1 KB holds the issuing wave for ~260 cycles (the path
//     moves about one dword per clock and wave), so the 21 pieces of interval it + 2 are requested at the END of interval it, five
//     per wave by the half of the workgroup that reaches the barrier first, into the third of three buffers;
//   * waves 0-3 run [MFMA phase, stencil phase], waves 4-7 [stencil phase, MFMA phase]: wave w and w + 4 share a SIMD, so one
//     partner's VALU issues under the other's MFMAs (tools/probe/issue_model.hip: 5 four-cycle VALU instructions hide per
//     32-cycle MFMA slot, also across the two waves of a SIMD; packed-f32 VALU does not -- the stencil uses plain v_fma_f32);
//   * inside the MFMA phase (branch-free: intervals 0, 1 and 60, 61 run on zeroed / unread operands) the f16 splits of P's B
//     values sit one instruction behind each of E's MFMAs and E's BN + ReLU6 + stores two behind each of P's
//     (sched_group_barrier); the phase's first LDS reads are requested before the stencil phase by waves 4-7.
// Measured and removed: weights through registers, pieces spread over the MFMA steps, a three-deep fragment
// ring, four 64-pixel waves, the chunked schedule.  Ablation switches of the first version: commit 0a06b2f.
// Output / residual: the sub-image's pixels are 4 apart in x, so stores are 4-byte pieces of rows that the three sibling workgroups
// (same image, same y phase, same XCD) complete in L2.
#if defined(IVF_F4_TIMING) || defined(IVF_D2_TIMING)
__device__ unsigned long long g_f4Tim[16];      // diagnostic build (make EXTRA=-DIVF_F4_TIMING): cycle sums per phase, waves 0 and 4
__device__ unsigned long long g_f4Whole[8];     // wave 0: prologue (input gather ... first barrier), interval loop, epilogue; workgroups
#endif
#ifdef IVF_F4_TIMING
#define F4_TIM(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                       tacc[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define F4_TIM(i) do { } while (0)
#endif
// Activation layout between the whole-block kernels of the 64 x 64 stage (r04): TILE-MAJOR.  Every workgroup of these kernels owns
// a tile of 256 pixels of one image (16 tiles per image) for ALL channels.  In planes (NCHW) those 256 pixels are scattered over
// each channel plane (dilation 4: every fourth pixel of every fourth row), so a lane's dword touches a 64-byte line of which the
// workgroup uses 16 bytes: the input gather and the residual loads of k_fcn_irbd4 are bound by the 10,240 lines a workgroup pulls
// through its L1 -- 10.7k cycles even when every line hits in L2, 13.8k from HBM (tools/probe/tile_gather.hip), 35k + 21k cycles of a
// workgroup's 360k in the kernel (one workgroup per CU: nothing overlaps them).  With the tile contiguous ([image][tile][channel][256]:
// 164 KB in one piece) the same gather takes 2.0k cycles from L2 and 9k when all CUs pull from HBM at once (4.5 TB/s: the chip's
// limit).  Measured and rejected on the way: all loads in flight before the first use (no change), 8-channel interleaving with
// dwordx4 accesses (NHWC8: 64 lines per instruction, + 7 %), staggered workgroup starts (no change while the lines are the bound).
//   lay 0  planes [image][channel][64][64]                         (ends of the chain, every fallback switch)
//   lay 1  tile = 4 rows x 64 columns (tile = y >> 2)               k_fcn_irbd2<DIL 1>, blocks 5-7
//   lay 2  tile = 8 x 32 strip of a (y & 1, x & 1) sub-image        k_fcn_irbd2<DIL 2>, blocks 8-14
//   lay 4  tile = the (y & 3, x & 3) sub-image, 16 x 16             k_fcn_irbd4, blocks 15-17
// A kernel reads its input and residual in ITS OWN layout and writes in the layout of its consumer (plan_layouts on the host).
// Element (channel c) of pixel (y, x) of image b at base + c * cs:
__device__ __forceinline__ void lay_addr(int lay, int C, int b, int y, int x, size_t& base, int& cs)
{
    if (lay == 0) { base = (size_t)b * C * 4096 + y * 64 + x; cs = 4096; return; }
    int tile, off;
    if (lay == 1) { tile = y >> 2; off = (y & 3) * 64 + x; }
    else if (lay == 2) { tile = ((y & 1) * 2 + (x & 1)) * 4 + (y >> 4); off = ((y >> 1) & 7) * 32 + (x >> 1); }
    else { tile = (y & 3) * 4 + (x & 3); off = (y >> 2) * 16 + (x >> 2); }
    base = ((size_t)(b * 16 + tile) * C) * 256 + off; cs = 256;
}
// inverse: pixel of (tile, offset inside the tile)
__device__ __forceinline__ void lay_pixel(int lay, int tile, int off, int& y, int& x)
{
    if (lay == 1) { y = tile * 4 + (off >> 6); x = off & 63; }
    else if (lay == 2) { const int sub = tile >> 2, r = (tile & 3) * 8 + (off >> 5); y = 2 * r + (sub >> 1); x = 2 * (off & 31) + (sub & 1); }
    else { y = 4 * (off >> 4) + (tile >> 2); x = 4 * (off & 15) + (tile & 3); }
}

// issue priority of the two halves of a whole-block workgroup (waves w and w + 4 share a SIMD; its vector issue is arbitrated by priority, then age):
//   0 none   1 / 2 static: waves 4-7 / waves 0-3 at priority 1   3 / 4 per phase: priority 1 during the MFMA phase / during the stencil phase
#ifndef IVF_PRIO
#define IVF_PRIO 0
#endif
#define IVF_PRIO_STATIC() do { if (IVF_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1); if (IVF_PRIO == 2 && wave < 4) __builtin_amdgcn_s_setprio(1); } while (0)
#define IVF_PRIO_MFMA(on) do { if (IVF_PRIO == 3) __builtin_amdgcn_s_setprio(on); } while (0)
#define IVF_PRIO_STEN(on) do { if (IVF_PRIO == 4) __builtin_amdgcn_s_setprio(on); } while (0)
constexpr int kF4Cin = 160, kF4Hid = 960, kF4Groups = 60;
constexpr int kF4HP = 20;                         // floats per 16-pixel sub-row of a hidden plane in LDS (80 B rows)
// LDS-DMA addressing per kernel: 1 = wave-uniform base in an SGPR pair + one 32-bit lane offset, 0 = a 64-bit pointer per lane.  Measured (us per 128 images, alternated): k_fcn_irbd4<true>
// 961 / 966 -> 946 / 946 (its 24 B of scratch go); k_fcn_irbd4h 1,790 / 1,796 -> 1,813 / 1,827; k_fcn_irbd2 +-0.5 %.
#ifndef IVF_F4_DMA_SADDR
#define IVF_F4_DMA_SADDR 1
#endif
#ifndef IVF_H4_DMA_SADDR
#define IVF_H4_DMA_SADDR 0
#endif
#ifndef IVF_W4_DMA_SADDR
#define IVF_W4_DMA_SADDR 0
#endif
#ifndef IVF_D2_DMA_SADDR
#define IVF_D2_DMA_SADDR 0
#endif
#ifndef IVF_F4_WALK
#define IVF_F4_WALK 0         // blocks 15 / 16 as a persistent grid (see k_fcn_irbd2): 953 -> 1,138 us, the tile loop makes the 256-register body spill; off
#endif
#ifndef IVF_RES_FROM_FRAGS
#define IVF_RES_FROM_FRAGS 1
#endif
#ifndef IVF_D2_WALK_ALL
#define IVF_D2_WALK_ALL 0
#endif
#ifndef IVF_D2_RES_FROM_FRAGS
#define IVF_D2_RES_FROM_FRAGS IVF_RES_FROM_FRAGS       // the same for k_fcn_irbd2's residual instances (blocks 6, 7, 9-11, 13, 14)
#endif
#ifndef IVF_F4_CS
#define IVF_F4_CS 324         // floats per channel plane of sH: 16 rows x kF4HP + 4, so that the four 16-lane groups of E's b32 stores hit different banks
#endif
constexpr int kF4CS = IVF_F4_CS;
constexpr int kF4DP = 260;                        // floats per channel of the depthwise output in LDS
constexpr int kF4ParB = 1024;                     // bytes per parameter slot (16 channels x 12 floats = 768 used)
#ifndef IVF_F4_DMA_A
#define IVF_F4_DMA_A 5        // weight pieces per wave of the half that reaches the barrier first (waves 0-3); waves 4-7 share the rest.
#endif                        // Measured 2 / 3 / 4 / 5 / 6: 100.5 / 99.9 / 99.1 / 97.4 / 98.0 us per image
constexpr int kF4WSlots = 3, kF4PSlots = 4;  // weight / parameter buffers: consumed in interval it, landed for it + 1, arriving for it + 2
constexpr size_t kF4Lds = (size_t)2 * 16 * kF4CS * 4 + (size_t)2 * 16 * kF4DP * 4 + 2 * kF4WSlots * 10240 + kF4PSlots * kF4ParB +
                          2 * 160 * 4;        // + the projection's BN scale / shift of this workgroup's 160 output channels
typedef float f32x4 __attribute__((ext_vector_type(4)));

// FP6 (r06, the r05 verdict's item 1): the two CORRECTION products of the split-f16 scheme on the block-scaled matrix instruction at four times the f16 rate.
//   x . w = x_hi w_hi + (x_hi w_lo + x_lo w_hi) + O(2^-22): the bracket is 2^-11 of the result, so its operands need ~4 bits -- it is ONE product
//   [x_hi | x_lo 2^10] . [w_lo 2^13 ; w_hi 2^3] 2^-13 of twice the K, B in fp6 (e2m3), A in bf6 (e3m2), on v_mfma_scale_f32_16x16x128_f8f6f4 into the
//   SAME accumulators as the f16 hi * hi product (tools/probe/mfma_fp8_mix.hip: lane maps, scale semantics, 17.6 cycles per K = 128 against 15.0 for
//   v_mfma_f32_16x16x32_f16).  Built for the EXPANSION of k_fcn_irbd4: its B operand is the block's input (unbounded), converted ONCE per workgroup in
//   the prologue with a power-of-two scale per lane block (pixel, 16 channels: hi and lo of 2 x 8) from the block's largest |x_hi| -- the instruction's
//   E8M0 scale operand undoes it; the A operand is packed on the host (make_fused4: 6-bit codes, element e at bits [6 e, 6 e + 6) of 6 dwords).
//   Measured: k_fcn_irbd4<true> 985 -> 937-945 us per 128 images on one box (963 -> 888 on another), worst golden 9.1e-5 -> 2.2e-4
//   (bar 3e-4, contract 1e-3).  The projection's corrections in the same form (pairs of hidden groups, K = 64 on 32x32x64) were built and measured too
//   (commit "WIP: FP6 correction products"): they need 11 more live registers than the 256 the two-waves-per-SIMD body has -- 916-986 us without spills,
//   2.8 ms with -- and are not in the tree.  NOT the default: one percent of the forward does not pay for 2.4 x the reference error; the experiment build
//   carries the kernels (IVF_FCN_FP6=1) and tests/test_gpu_fcn.py keeps them at the goldens' bar.
#ifndef IVF_F4_FP6
#define IVF_F4_FP6 0
#endif
#if IVF_F4_FP6 || defined(IVF_EXPERIMENT)
#define IVF_F4_FP6_BUILT 1
#else
#define IVF_F4_FP6_BUILT 0
#endif
constexpr int kF6SH = 3;                          // weights: w_hi 2^3 (< 16), w_lo 2^13 (<= 4) in e3m2 (largest 28)

// one sub-row's split input fragments (five K steps of 32 channels: lane (column, kq) holds 8 channels of each) -> the three fp6 (e2m3) B operands of
// v_mfma_scale_f32_16x16x128_f8f6f4 for the correction product, [x_hi (K step 2c) | x_lo 2^10 | x_hi (K step 2c + 1) | x_lo 2^10] / 2^sb per lane, and the
// lane's three E8M0 scale bytes (byte c = 127 + sb of operand c).  x_lo < ulp(x_hi) = 2^-10 2^floor(log2 x_hi), so a block's largest element is an x_hi:
// sb = floor(log2 max |x_hi|) - 2 puts it in [4, 8) (e2m3: largest 7.5).  Operand 2 has K step 4 only (the rest zero).
__device__ __forceinline__ void fp6_pack(const HFrag (&bh)[5], const HFrag (&bl)[5], i32x6 (&xq)[3], int& sbBytes)
{
    sbBytes = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const f16x8 k1024 = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024};
        const f16x8 h0 = bh[2 * c].v, l0 = bl[2 * c].v * k1024;
        const f16x8 h1 = c < 2 ? bh[c < 2 ? 2 * c + 1 : 0].v : z8, l1 = c < 2 ? bl[c < 2 ? 2 * c + 1 : 0].v * k1024 : z8;
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) amax = fmaxf(amax, fmaxf(fabsf((float)h0[j]), fabsf((float)h1[j])));
        const int sb = min(max(__builtin_amdgcn_frexp_expf(amax) - 3, -40), 20);
        const f16x32 src = {h0[0], h0[1], h0[2], h0[3], h0[4], h0[5], h0[6], h0[7], l0[0], l0[1], l0[2], l0[3], l0[4], l0[5], l0[6], l0[7],
                            h1[0], h1[1], h1[2], h1[3], h1[4], h1[5], h1[6], h1[7], l1[0], l1[1], l1[2], l1[3], l1[4], l1[5], l1[6], l1[7]};
        xq[c] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(src, __builtin_bit_cast(float, (127 + sb) << 23));
        sbBytes |= (127 + sb) << (8 * c);
    }
}

// SPLIT (small batches, r04): 16 workgroups per image leave most of the chip idle at batch 1 (the per-call drop-in path).  The hidden
// groups are cut into gridDim.z contiguous ranges; a workgroup runs the same pipeline over its range only and stores its RAW projection
// accumulators into `part` [gridDim.z][images][Cout][4096]; k_fcn_split_reduce adds the ranges in index order (deterministic) and applies
// the projection's BN (+ residual).  The batched instantiation (SPLIT = false) is unchanged: g0 / g1 are constants there.
template <bool RES, bool SPLIT = false, bool FP6 = false>      // FP6: the expansion's correction products on bf6 x fp6 (see above)
__global__ __launch_bounds__(512, 2) void k_fcn_irbd4(const float* __restrict__ X, const uint4* __restrict__ WE, const float* __restrict__ par,
                                                     const uint4* __restrict__ WP, const float* __restrict__ scP, const float* __restrict__ shP,
                                                     const float* __restrict__ res, float* __restrict__ Y, int Cout, int tilesP, float* __restrict__ part,
                                                     int layIn, int layOut, int nT)
{
    const int g0 = SPLIT ? (int)(blockIdx.z * kF4Groups / gridDim.z) : 0, g1 = SPLIT ? (int)((blockIdx.z + 1) * kF4Groups / gridDim.z) : kF4Groups;
    extern __shared__ __attribute__((aligned(16))) uint4 f4smem[];
    // nT = tiles of the launch (16 per image); blocks 15 / 16 may run as a persistent grid that walks them (see k_fcn_irbd2)
    constexpr bool kWalk = IVF_F4_WALK && RES && !SPLIT;
    const int perX_ = nT >> 3, strideX_ = (int)gridDim.x >> 3;
    int slot_ = (int)blockIdx.x >> 3;
    do {
#ifdef IVF_F4_TIMING
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#endif
    float* const sH = (float*)f4smem;                               // [2][16 ch][kF4CS >= 16 rows x kF4HP]
    float* const sD = sH + 2 * 16 * kF4CS;                          // [2][16 ch][kF4DP]
    uint4* const sWE = (uint4*)(sD + 2 * 16 * kF4DP);               // [slots][5 K steps][hi, lo][64 lanes]
    uint4* const sWP = sWE + kF4WSlots * 640;                       // [slots][5 tiles][hi, lo][64 lanes]
    float* const sPar = (float*)(sWP + kF4WSlots * 640);            // [slots][16 ch][12]: 9 taps (x dw BN scale), dw BN shift, expansion BN scale, shift
    float* const sBN = sPar + kF4PSlots * (kF4ParB / 4);            // [scale 160 | shift 160] of the projection (epilogue)
#if IVF_F4_WALK
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));      // opaque per tile: otherwise every lane-dependent constant of the body is hoisted out of the tile walk and kept live across it (spills)
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
#else
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#endif
    const int nwg = nT, L = (int)(blockIdx.x & 7) * perX_ + slot_;                    // consecutive L on one XCD (nT = 16 * images)
    const int b = L >> 4, py = (L >> 2) & 3, px = L & 3;
    const int tile0 = blockIdx.y * 5;
    constexpr int HW = 4096;

    // ---- weights / parameters of one interval by LDS-DMA: 21 pieces of <= 1 KB, piece c by wave c % 8.  Issued through inline
    // asm: the compiler's wait-count pass treats its own LDS-DMA builtin as a store to ALL of LDS and puts s_waitcnt vmcnt(0) in
    // front of every later ds_read, which would expose the whole L2 latency of the prefetch every interval.  Here the only
    // consumer-side wait is the explicit vmcnt(0) in front of the interval's barrier (extra in-flight VMEM operations can only make
    // the compiler's own counted waits longer, never too short: vmcnt retires in order).
    const unsigned ldsBase = (unsigned)(uintptr_t)f4smem;
    const unsigned ldsWE = ldsBase + (unsigned)((uint8_t*)sWE - (uint8_t*)f4smem), ldsWP = ldsBase + (unsigned)((uint8_t*)sWP - (uint8_t*)f4smem),
                   ldsPar = ldsBase + (unsigned)((uint8_t*)sPar - (uint8_t*)f4smem);
#if IVF_F4_DMA_SADDR      // r05: wave-uniform base in an SGPR pair + ONE 32-bit lane offset (16 lane): the per-lane 64-bit pointers of WE / WP / par (6 registers, kept live across the
                       // whole kernel) and their 64-bit vector adds per piece are gone
    const unsigned voff16 = (unsigned)lane * 16u;
    auto dma16 = [voff16](const void* sbase, unsigned ldsAddr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff16), "s"(sbase), "s"(ldsAddr) : "memory");
    };
#undef IVF_DMA_LANE
#define IVF_DMA_LANE(x) 0
#else
    auto dma16 = [](const void* src, unsigned ldsAddr) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(ldsAddr) : "memory");
    };
#undef IVF_DMA_LANE
#define IVF_DMA_LANE(x) (x)
#endif
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    auto piece = [&](int it, int c) {           // piece c of what interval `it` consumes: WE[it] (c < 10), WP[it - 2] (c < 20), par[it] (c = 20)
        const int nb = it % kF4WSlots;
        if (c < 10) {
            if (it < g1) dma16(WE + ((size_t)it * 10 + c) * 64 + IVF_DMA_LANE(lane), ldsWE + (unsigned)(nb * 640 + c * 64) * 16u);
        } else if (c < 20) {
            const int c2 = c - 10, gp = it - 2;
            if (gp >= g0 && gp < g1)
                dma16(WP + (((size_t)gp * tilesP + tile0) * 2 + c2) * 64 + IVF_DMA_LANE(lane), ldsWP + (unsigned)(nb * 640 + c2 * 64) * 16u);
        } else if (c == 20) {
            if (it < g1 && lane < 48) dma16(par + (size_t)it * 192 + IVF_DMA_LANE(lane * 4), ldsPar + (unsigned)((it % kF4PSlots) * kF4ParB));
        }
    };
    auto dma = [&](int it) {                    // all 21 pieces (<= 1 KB each), piece c by wave c % 8
#pragma unroll
        for (int r = 0; r < 3; r++) piece(it, uwave + 8 * r);
    };
    auto dma_late = [&](int it) {               // the same pieces, most of them by waves 0-3, which reach the barrier first
        constexpr int NA = IVF_F4_DMA_A, NB = 4 * NA >= 21 ? 0 : (21 - 4 * NA + 3) / 4;
        if (uwave < 4) {
#pragma unroll
            for (int r = 0; r < NA; r++) { const int c = uwave + 4 * r; if (c < 21) piece(it, c); }
        } else {
#pragma unroll
            for (int r = 0; r < NB; r++) { const int c = 4 * NA + (uwave - 4) + 4 * r; if (c < 21) piece(it, c); }
        }
    };
    dma(g0);
    dma(g0 + 1);

    // ---- the input tile: this wave's 32 sub-image pixels (sub-rows 2w, 2w+1) x 160 channels as B fragments of the 16x16x32 MFMA
    // lane: column n = lane & 15 (sub-column), k = 8 (lane >> 4) + j
    // r04: all 80 loads of a lane are requested before the first value is used (the accumulators are not live yet, so the registers
    // are there): one memory round trip instead of ten dependent ones -- with one workgroup per CU nothing else hides them
    // (measured, batch 128: 36k of the workgroup's 360k cycles were this gather, 44k the epilogue's residual loads, tile by tile).
    HFrag bh[5][2], bl[5][2];
    {
        float xv[2][5][8];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            size_t xb; int cs;
            lay_addr(layIn, kF4Cin, b, 4 * (2 * wave + u) + py, 4 * (lane & 15) + px, xb, cs);
            const float* Xp = X + xb + (size_t)(8 * (lane >> 4)) * cs;
#pragma unroll
            for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
                for (int j = 0; j < 8; j++) xv[u][s5][j] = Xp[(size_t)(32 * s5 + j) * cs];
        }
        if (tid < 160) { sBN[tid] = scP[tile0 * 32 + tid]; sBN[160 + tid] = shP[tile0 * 32 + tid]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++) split_pair(xv[u][s5][2 * jj], xv[u][s5][2 * jj + 1], bh[s5][u].u[jj], bl[s5][u].u[jj]);
    }
    // FP6: the correction operand of the expansion, made once: instruction c (K = 128: 4 lane groups x 32 elements) holds, per lane,
    // [x_hi (8 channels of K step 2c) | x_lo 2^10 | x_hi (K step 2c + 1) | x_lo 2^10] / 2^sb; c = 2 has K step 4 only (the rest zero).
    // x_lo < ulp(x_hi) = 2^-10 2^floor(log2 x_hi), so the block's largest element is an x_hi: sb = floor(log2 max |x_hi|) - 2 puts it in [4, 8).
    i32x6 xq[3][2]; int sbB[2] = {0, 0};
    if constexpr (FP6) {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
                const f16x8 k1024 = {1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024};
                const f16x8 h0 = bh[2 * c][u].v, l0 = bl[2 * c][u].v * k1024;
                const f16x8 h1 = c < 2 ? bh[c < 2 ? 2 * c + 1 : 0][u].v : z8, l1 = c < 2 ? bl[c < 2 ? 2 * c + 1 : 0][u].v * k1024 : z8;
                float amax = 0.f;
#pragma unroll
                for (int j = 0; j < 8; j++) amax = fmaxf(amax, fmaxf(fabsf((float)h0[j]), fabsf((float)h1[j])));
                const int sb = min(max(__builtin_amdgcn_frexp_expf(amax) - 3, -40), 20);
                const f16x32 src = {h0[0], h0[1], h0[2], h0[3], h0[4], h0[5], h0[6], h0[7], l0[0], l0[1], l0[2], l0[3], l0[4], l0[5], l0[6], l0[7],
                                    h1[0], h1[1], h1[2], h1[3], h1[4], h1[5], h1[6], h1[7], l1[0], l1[1], l1[2], l1[3], l1[4], l1[5], l1[6], l1[7]};
                xq[c][u] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(src, __builtin_bit_cast(float, (127 + sb) << 23));
                sbB[u] |= (127 + sb) << (8 * c);
            }
    }
    f32x16 pacc[5];
#pragma unroll
    for (int t = 0; t < 5; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) pacc[t][q] = 0.f;

    // stencil thread: channel sch (0..15), sub-row ssr, half row sh_
    const int sch = tid >> 5, ssr = (tid >> 1) & 15, sh_ = tid & 1;
    const float rowM0 = ssr > 0 ? 1.f : 0.f, rowM2 = ssr < 15 ? 1.f : 0.f;
    const int srow0 = sch * kF4CS + (ssr > 0 ? ssr - 1 : ssr) * kF4HP + 8 * sh_, srow1 = sch * kF4CS + ssr * kF4HP + 8 * sh_,
              srow2 = sch * kF4CS + (ssr < 15 ? ssr + 1 : ssr) * kF4HP + 8 * sh_;
    const float mL = sh_ ? 1.f : 0.f, mR = sh_ ? 0.f : 1.f;          // the halo pixel comes from the row's other half (lane -1 / +1)

    for (int i = tid; i < 2 * 16 * kF4DP / 4; i += 512) ((uint4*)sD)[i] = make_uint4(0u, 0u, 0u, 0u);      // P(-2), P(-1): zero operands
    for (int i = tid; i < (SPLIT ? 3 : 2) * 640; i += 512) sWP[i] = make_uint4(0u, 0u, 0u, 0u);      // slots g0 % 3, (g0 + 1) % 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef IVF_F4_TIMING
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tk1 = tlast;
#endif
    // ---- MFMA phase, branch-free form.  Intervals 0, 1 run P on zeroed operands (sD and the first two sWP slots are cleared in the
    // prologue), intervals 60, 61 run E into planes nobody reads: no `it`-dependent control flow inside the phase.
    struct MPre { float dv[8]; float2 eb[4]; HFrag ea0[2], pa0[2]; };
    auto mfma_pre = [&](int it, MPre& m) {      // every LDS read of the phase that depends on no MFMA
        const int cur = it & 1, ws = it % kF4WSlots;
        const float* dB = sD + cur * (16 * kF4DP) + (8 * (lane >> 5)) * kF4DP + 32 * wave + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; j++) m.dv[j] = dB[j * kF4DP];
        const float* pp = sPar + (it % kF4PSlots) * (kF4ParB / 4) + (4 * (lane >> 4)) * 12 + 10;
#pragma unroll
        for (int r = 0; r < 4; r++) m.eb[r] = *(const float2*)(pp + r * 12);
        const uint4* wE = sWE + ws * 640 + lane;
        const uint4* wPq = sWP + ws * 640 + lane;
        m.ea0[0].q = wE[0]; m.ea0[1].q = wE[64];
        m.pa0[0].q = wPq[0]; m.pa0[1].q = wPq[64];
    };
    auto mfma_main = [&](int it, MPre& m) {
        const int cur = it & 1, ws = it % kF4WSlots;
        const uint4* wE = sWE + ws * 640 + lane;
        const uint4* wPq = sWP + ws * 640 + lane;
        HFrag ea[2][2], pa[2][2], ph, pl;
        ea[0][0] = m.ea0[0]; ea[0][1] = m.ea0[1]; pa[0][0] = m.pa0[0]; pa[0][1] = m.pa0[1];
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
        F4_TIM(4);
#pragma unroll
        for (int s5 = 0; s5 < 5; s5++) {        // E(it): hidden group `it` = W_E[16 x 160] . X[160 x 32 pixels of this wave]
            if (s5 + 1 < 5) { ea[(s5 + 1) & 1][0].q = wE[(2 * s5 + 2) * 64]; ea[(s5 + 1) & 1][1].q = wE[(2 * s5 + 3) * 64]; }
            const HFrag &ah = ea[s5 & 1][0], &al = ea[s5 & 1][1];
            e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.v, bh[s5][0].v, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.v, bh[s5][1].v, e1, 0, 0, 0);
            e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bl[s5][0].v, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bl[s5][1].v, e1, 0, 0, 0);
            e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bh[s5][0].v, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bh[s5][1].v, e1, 0, 0, 0);
            if (s5 < 4) split_pair(m.dv[2 * s5], m.dv[2 * s5 + 1], ph.u[s5], pl.u[s5]);      // P's B fragment, one pair per step
            // order inside the step: the next fragments' reads first, then one VALU instruction behind each MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int i = 0; i < 6; i++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
        F4_TIM(5);
        // C layout of E: column = lane & 15 (sub-column), row = 4 (lane >> 4) + r (hidden channel of the group)
        float* hp = sH + cur * (16 * kF4CS) + (4 * (lane >> 4)) * kF4CS + (2 * wave) * kF4HP + (lane & 15);
#pragma unroll
        for (int t = 0; t < 5; t++) {           // P(it - 2): out[160 x 32 pixels] += W_P[160 x 16] . D[16 x 32 pixels]
            if (t + 1 < 5) { pa[(t + 1) & 1][0].q = wPq[(2 * t + 2) * 64]; pa[(t + 1) & 1][1].q = wPq[(2 * t + 3) * 64]; }
            const HFrag &ah = pa[t & 1][0], &al = pa[t & 1][1];
            pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, ph.v, pacc[t], 0, 0, 0);
            pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, pl.v, pacc[t], 0, 0, 0);
            pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, ph.v, pacc[t], 0, 0, 0);
            if (t >= 1) {                       // E's epilogue, one row per step (E's last MFMA retired during step 0): BN + ReLU6 -> planes
                const int r = t - 1;
                hp[r * kF4CS] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e0[r], m.eb[r].x, m.eb[r].y), 0.f, 6.f);
                hp[r * kF4CS + kF4HP] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e1[r], m.eb[r].x, m.eb[r].y), 0.f, 6.f);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        F4_TIM(6);
    };
    // FP6 form of the MFMA phase: E = 10 x 16x16x32 f16 (hi * hi; P's B splits behind them) + 6 x 16x16x128 bf6 x fp6 (both corrections); P as in r05.
    auto mfma_main6 = [&](int it, MPre& m) {
        const int cur = it & 1, ws = it % kF4WSlots;
        const uint4* wE = sWE + ws * 640 + lane;
        const uint2* wE2 = (const uint2*)(sWE + ws * 640 + 8 * 64) + lane;
        const uint4* wPq = sWP + ws * 640 + lane;
        HFrag ea[2], ph, pl;
        ea[0] = m.ea0[0];
        uint4 q6[3]; uint2 r6[3];
#pragma unroll
        for (int c = 0; c < 3; c++) { q6[c] = wE[(5 + c) * 64]; r6[c] = wE2[c * 64]; }
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
        F4_TIM(4);
#pragma unroll
        for (int s5 = 0; s5 < 5; s5++) {
            if (s5 + 1 < 5) ea[(s5 + 1) & 1].q = wE[(s5 + 1) * 64];
            e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea[s5 & 1].v, bh[s5][0].v, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea[s5 & 1].v, bh[s5][1].v, e1, 0, 0, 0);
            if (s5 < 4) split_pair(m.dv[2 * s5], m.dv[2 * s5 + 1], ph.u[s5], pl.u[s5]);
        }
        auto ecorr = [&](int c, auto OPS) {
            constexpr int OP = decltype(OPS)::value;
            const i32x8 a6 = {(int)q6[c].x, (int)q6[c].y, (int)q6[c].z, (int)q6[c].w, (int)r6[c].x, (int)r6[c].y, 0, 0};
            const i32x8 b0 = {xq[c][0][0], xq[c][0][1], xq[c][0][2], xq[c][0][3], xq[c][0][4], xq[c][0][5], 0, 0};
            const i32x8 b1 = {xq[c][1][0], xq[c][1][1], xq[c][1][2], xq[c][1][3], xq[c][1][4], xq[c][1][5], 0, 0};
            e0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b0, e0, 3, 2, 0, 127 - 10 - kF6SH, OP, sbB[0]);
            e1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b1, e1, 3, 2, 0, 127 - 10 - kF6SH, OP, sbB[1]);
        };
        ecorr(0, std::integral_constant<int, 0>{});
        ecorr(1, std::integral_constant<int, 1>{});
        ecorr(2, std::integral_constant<int, 2>{});
        F4_TIM(5);
        float* hp = sH + cur * (16 * kF4CS) + (4 * (lane >> 4)) * kF4CS + (2 * wave) * kF4HP + (lane & 15);
        auto eepi = [&](int r) {                // E's epilogue, one row: BN + ReLU6 -> planes
            hp[r * kF4CS] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e0[r], m.eb[r].x, m.eb[r].y), 0.f, 6.f);
            hp[r * kF4CS + kF4HP] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e1[r], m.eb[r].x, m.eb[r].y), 0.f, 6.f);
        };
        {                                       // the projection as three f16 products (the r05 loop)
            HFrag pa[2][2];
            pa[0][0] = m.pa0[0]; pa[0][1] = m.pa0[1];
#pragma unroll
            for (int t = 0; t < 5; t++) {
                if (t + 1 < 5) { pa[(t + 1) & 1][0].q = wPq[(2 * t + 2) * 64]; pa[(t + 1) & 1][1].q = wPq[(2 * t + 3) * 64]; }
                const HFrag &ah = pa[t & 1][0], &al = pa[t & 1][1];
                pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, ph.v, pacc[t], 0, 0, 0);
                pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, pl.v, pacc[t], 0, 0, 0);
                pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, ph.v, pacc[t], 0, 0, 0);
                if (t >= 1) eepi(t - 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        F4_TIM(6);
    };
    auto stencil_phase = [&](int it) {          // S(it - 1): 3x3 on the 16 x 16 planes of group it - 1, + BN + ReLU6
        const int g = it - 1;
        if (g < g0 || g >= g1) return;
        const float* hp = sH + (g & 1) * (16 * kF4CS);
        const float4* pq = (const float4*)(sPar + (g % kF4PSlots) * (kF4ParB / 4) + sch * 12);
        const float4 w03 = pq[0], w47 = pq[1], w8s = pq[2];          // taps 0-3 | 4-7 | tap 8, shift, (expansion BN)
        float o[8];
#pragma unroll
        for (int p8 = 0; p8 < 8; p8++) o[p8] = w8s.y;
        const int ro[3] = {srow0, srow1, srow2};
        const float wk[9] = {w03.x * rowM0, w03.y * rowM0, w03.z * rowM0, w03.w, w47.x, w47.y, w47.z * rowM2, w47.w * rowM2, w8s.x * rowM2};
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const float4 a = *(const float4*)(hp + ro[ky]), c4 = *(const float4*)(hp + ro[ky] + 4);
            const float own[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
            const float w0 = wk[3 * ky], w1 = wk[3 * ky + 1], w2 = wk[3 * ky + 2];
            const float w0L = w0 * mL, w2R = w2 * mR;
#pragma unroll
            for (int p8 = 0; p8 < 8; p8++) {
                o[p8] = __builtin_fmaf(own[p8], w1, o[p8]);
                if (p8 > 0) o[p8] = __builtin_fmaf(own[p8 - 1], w0, o[p8]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p8]) : "v"(own[7]), "v"(w0L));
                if (p8 < 7) o[p8] = __builtin_fmaf(own[p8 + 1], w2, o[p8]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p8]) : "v"(own[0]), "v"(w2R));
            }
        }
        float* dp = sD + (g & 1) * (16 * kF4DP) + sch * kF4DP + ssr * 16 + 8 * sh_;
        *(float4*)dp = make_float4(__builtin_amdgcn_fmed3f(o[0], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[1], 0.f, 6.f),
                                   __builtin_amdgcn_fmed3f(o[2], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[3], 0.f, 6.f));
        *(float4*)(dp + 4) = make_float4(__builtin_amdgcn_fmed3f(o[4], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[5], 0.f, 6.f),
                                         __builtin_amdgcn_fmed3f(o[6], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[7], 0.f, 6.f));
    };

    IVF_PRIO_STATIC();
    for (int it = g0; it < g1 + 2; it++) {
        F4_TIM(0);
        {
            MPre m;
            mfma_pre(it, m); __builtin_amdgcn_sched_barrier(0);
            auto mm = [&]() { if constexpr (FP6) mfma_main6(it, m); else mfma_main(it, m); };
            if (wave < 4) { IVF_PRIO_MFMA(1); mm(); IVF_PRIO_MFMA(0); F4_TIM(1); IVF_PRIO_STEN(1); stencil_phase(it); IVF_PRIO_STEN(0); F4_TIM(2); }
            else { IVF_PRIO_STEN(1); stencil_phase(it); IVF_PRIO_STEN(0); F4_TIM(2); __builtin_amdgcn_sched_barrier(0); IVF_PRIO_MFMA(1); mm(); IVF_PRIO_MFMA(0); F4_TIM(1); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the pieces of it + 1, requested an interval ago
        dma_late(it + 2); F4_TIM(0);                             // land during it + 1; their slots were last read in it - 1
        __syncthreads();
        F4_TIM(3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef IVF_F4_TIMING
    const unsigned long long tk2 = __builtin_amdgcn_s_memtime();
#ifndef IVF_F4_TIM_H4ONLY        // (a variant build that times k_fcn_irbd4h alone)
    if (lane == 0 && (wave == 0 || wave == 4)) {
        for (int i = 0; i < 7; i++) atomicAdd(&g_f4Tim[(wave ? 8 : 0) + i], tacc[i]);
        atomicAdd(&g_f4Tim[(wave ? 8 : 0) + 7], 1ull);
    }
#endif
#endif

    // ---- epilogue: BN (+ residual) of the projection, 4-byte pieces (the sub-image's pixels are 4 apart)
    const int n = lane & 31;
    const int oy = 4 * (2 * wave + (n >> 4)) + py, ox = 4 * (n & 15) + px;      // this lane's pixel
    // the residual values of all five tiles are requested at once (the input fragments are dead: 80 registers), BN parameters from LDS.
    // Lane's channels of tile t: cb + (q & 3) + 8 (q >> 2), cb = 32 (tile0 + t) + 4 (lane >> 5)
    // r05 (IVF_RES_FROM_FRAGS): the residual IS the block's input, which this wave still holds as split-f16 B fragments of the same 32 pixels
    // (hi + lo = the 22-bit value the expansion multiplied; the residual add then differs from the exact f32 by <= 2^-22 relative).  Re-reading
    // it from memory cost the epilogue a round trip with every CU of the chip asking at once (phase timers: 34k of a workgroup's ~230k cycles).
    // The fragments are transposed to the accumulator layout tile by tile through a wave-private 4.6 KB piece of the (now idle) hidden planes.
    constexpr bool kResFrags = RES && !SPLIT && IVF_RES_FROM_FRAGS && !FP6;      // FP6: the exact lo halves are gone after the prologue
    float* const sxw = (float*)f4smem + wave * (32 * 36);
    float rvAll[5][16];
    if (RES && !SPLIT && !kResFrags) {
        size_t rb; int rcs;
        lay_addr(layIn, Cout, b, oy, ox, rb, rcs);
        const float* rp = res + rb + (size_t)(tile0 * 32 + 4 * (lane >> 5)) * rcs;
#pragma unroll
        for (int t = 0; t < 5; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) rvAll[t][q] = rp[(size_t)(t * 32 + (q & 3) + 8 * (q >> 2)) * rcs];
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef IVF_F4_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tkA = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    size_t ob; int ocs;
    lay_addr(layOut, Cout, b, oy, ox, ob, ocs);
    float* const yp = (SPLIT ? part + (size_t)blockIdx.z * (nwg / 16) * Cout * HW : Y) + ob + (size_t)(tile0 * 32 + 4 * (lane >> 5)) * ocs;
    float amaxOut = 0.f;
#pragma unroll
    for (int t = 0; t < 5; t++) {
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(sBN + t * 32 + 4 * (lane >> 5) + 8 * g4); sh4[g4] = *(const float4*)(sBN + 160 + t * 32 + 4 * (lane >> 5) + 8 * g4); }
        float rvT[16];
        if (kResFrags) {
            // channels 32 t .. 32 t + 31 of this wave's 32 pixels: fragment lane (column lane & 15, channel group lane >> 4) -> [pixel][channel] in LDS
            // -> accumulator lane (pixel lane & 31, channels 4 (lane >> 5) + (q & 3) + 8 (q >> 2)).  LDS executes a wave's instructions in order;
            // the fences only keep the compiler from moving the reads above the writes (and the next tile's writes above these reads)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                float f[8];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    uint32_t hu = bh[t][u].u[jj], lu = bl[t][u].u[jj];
                    asm volatile("" : "+v"(hu), "+v"(lu));      // opaque: otherwise the compiler computes these sums in the PROLOGUE (they depend on nothing
                                                                // in the loop) and parks all 80 in scratch across it -- 360 B per lane spilled and reloaded
                    const f16x2 h = __builtin_bit_cast(f16x2, hu), l = __builtin_bit_cast(f16x2, lu);
                    f[2 * jj] = (float)h[0] + (float)l[0]; f[2 * jj + 1] = (float)h[1] + (float)l[1];
                }
                float* d = sxw + (16 * u + (lane & 15)) * 36 + 8 * (lane >> 4);
                *(float4*)d = make_float4(f[0], f[1], f[2], f[3]); *(float4*)(d + 4) = make_float4(f[4], f[5], f[6], f[7]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            const float* sr = sxw + n * 36 + 4 * (lane >> 5);
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) { const float4 r = *(const float4*)(sr + 8 * g4); rvT[4 * g4] = r.x; rvT[4 * g4 + 1] = r.y; rvT[4 * g4 + 2] = r.z; rvT[4 * g4 + 3] = r.w; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            float v = pacc[t][q];                   // SPLIT: the raw sums of this range of hidden groups, in the OUTPUT layout
            if (!SPLIT) {
                v = v * vget<4>(sc4[q >> 2], q & 3) + vget<4>(sh4[q >> 2], q & 3);
                if (RES) v += kResFrags ? rvT[q] : rvAll[t][q];
                range_note(amaxOut, v);
            }
            yp[(size_t)(t * 32 + (q & 3) + 8 * (q >> 2)) * ocs] = v;
        }
    }
    if (!SPLIT) range_flag(amaxOut);            // SPLIT: k_fcn_split_reduce checks the finished sums
#ifdef IVF_F4_TIMING
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tkB = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        const unsigned long long tk3 = __builtin_amdgcn_s_memtime();
        atomicAdd(&g_f4Whole[0], tk1 - tk0); atomicAdd(&g_f4Whole[1], tk2 - tk1); atomicAdd(&g_f4Whole[2], tk3 - tk2); atomicAdd(&g_f4Whole[3], 1ull);
        atomicAdd(&g_f4Whole[4], tkA - tk2); atomicAdd(&g_f4Whole[5], tkB - tkA); atomicAdd(&g_f4Whole[6], tk3 - tkB);
    }
#endif
    } while (kWalk && (slot_ += strideX_) < perX_);      // tiles of this workgroup
}

// ---- k_fcn_irbd4h (r05): block 17 (160 -> 960 -> 320, no residual) in ONE pass over the hidden groups ----
// k_fcn_irbd4 holds the projection accumulators of 160 output channels (80 registers beside the 80 of the input fragments), so block 17
// (320 outputs) ran it as TWO workgroups per 256-pixel tile, each computing the whole 160 -> 960 expansion and the 960-channel stencil for
// its half of the outputs: a third of the block's matrix work and all of its stencil twice (r04 verdict: 0.098 of the f16 peak
// algorithmic, the single most expensive launch).  A 256-pixel tile with 320 outputs does not fit a CU's register file (input 164 KB +
// accumulators 328 KB of 512).  HALF a sub-image does: one workgroup = 8 of the 16 rows of a (y & 3, x & 3) sub-image (128 pixels) for
// ALL 320 outputs,
//   * E: wave w expands its own sub-row (15 x v_mfma_f32_16x16x32_f16 per group); the stencil also needs the sub-row just outside the
//     half (row 8 for the upper half, row 7 for the lower one): its split input fragments sit in LDS (10 KB, written once) and wave 0
//     expands it as well -- 9 row expansions per 128 pixels instead of 2 x 8 per 128;
//   * S: thread = (channel, row, quarter row): 4 pixels, the pixel beside the quarter from the neighbouring lane (DPP);
//   * P: wave w = pixel tile w & 3 (two sub-rows, 32 pixels) x output half w >> 2 (five 32-channel tiles, 80 accumulator registers);
//   * the projection's A fragments (20 KB per group for 320 outputs) do NOT go through LDS: with only one sub-row of input in registers
//     (40 instead of 80) a wave has room for its five tiles' fragments (40 registers), loaded straight from L2 one interval ahead -- tile t
//     of interval it + 1 right behind the MFMAs that used tile t of interval it.  The LDS-DMA stream, the costliest part of the loop
//     around the MFMAs in k_fcn_irbd4, shrinks from 21 pieces per 256 pixels and interval to 11 per 128
//     (the expansion's fragments + the parameters); vmcnt retires in order, so the wait in front of the interval's barrier is
//     vmcnt(10): everything older than this interval's ten fragment loads = the DMA pieces of the next interval.
// Everything else is k_fcn_irbd4's: groups of 16 hidden channels, E(g) / S(g - 1) / P(g - 2) on different LDS buffers, one barrier per
// interval, the two halves of the workgroup in opposite phase order.  Input in `lay 4` (tile-major), output in `layOut` (planes for the
// decoder).  Batched form only (the small-batch schedule keeps the SPLIT instance of k_fcn_irbd4).
constexpr int kH4Rows = 10;                       // row slots of a hidden plane: the 8 own sub-rows + the one above + the one below
constexpr int kH4CS = kH4Rows * kF4HP + 4;        // floats per channel plane of sH (204: the four 16-lane groups of E's b32 stores hit different banks)
constexpr int kH4DP = 128 + 4;                    // floats per channel of the depthwise output
constexpr int kH4Cout = 320, kH4TilesP = 10;
constexpr size_t kH4Lds = (size_t)2 * 16 * kH4CS * 4 + (size_t)2 * 16 * kH4DP * 4 + kF4WSlots * 10240 + kF4PSlots * kF4ParB + 2 * kH4Cout * 4 + 10240;
#ifndef IVF_H4_ABL
#define IVF_H4_ABL 0          // timing-only ablations (compile time; results wrong): 1 no projection-fragment loads in the loop, 2 E's A fragments read once,
#endif                        // 4 no LDS-DMA in the loop, 8 no stencil, 16 no halo expansion, 32 no P MFMAs, 64 no E MFMAs
#ifndef IVF_H4_DMA_SKIP0
#define IVF_H4_DMA_SKIP0 1    // the wave that also expands the halo row (wave 0) issues no LDS-DMA pieces: 1,846 / 1,861 vs 1,874 / 1,911 us per 128 images
                              // (a wave's pieces consecutive, one M0 + one address per wave and the instruction's immediate offset for the rest -- the offset
                              // moves the LDS address too -- is no faster: 1,823 / 1,842 vs 1,812 / 1,829: the cost of a piece is not its scalar set-up)
#endif
#ifndef IVF_H4_HALO_WAVE
#define IVF_H4_HALO_WAVE 0
#endif
#ifndef IVF_H4_WALK
#define IVF_H4_WALK 0         // block 17 as a persistent grid: 1,785 -> 1,888 us (the tile loop around a 242-register body spills 104 B per lane); off
#endif
#ifndef IVF_H4_DMA_A
#define IVF_H4_DMA_A 3        // expansion-weight pieces per wave of the half that reaches the barrier first (waves 0-3); waves 4-7 share the rest of the 11.
                              // Measured 1 / 2 / 3: 1,905 / 1,881 / 1,849 us per 128 images
#endif
template <bool FP6 = false>      // FP6 (r06): the expansion's correction products on bf6 x fp6, see k_fcn_irbd4; WE then points to the FP6 form (make_fused4: dWE6)
__global__ __launch_bounds__(512, 2) void k_fcn_irbd4h(const float* __restrict__ X, const uint4* __restrict__ WE, const float* __restrict__ par,
                                                      const uint4* __restrict__ WP, const float* __restrict__ scP, const float* __restrict__ shP,
                                                      float* __restrict__ Y, int layIn, int layOut, int nT)
{
    constexpr int g1 = kF4Groups;
    extern __shared__ __attribute__((aligned(16))) uint4 f4smem[];
    // nT = tiles of the launch (32 per image); the grid may be one workgroup per CU that walks them (see k_fcn_irbd2)
    const int perX_ = nT >> 3, strideX_ = (int)gridDim.x >> 3;
    int slot_ = (int)blockIdx.x >> 3;
    do {
    float* const sH = (float*)f4smem;                               // [2][16 ch][kH4CS]
    float* const sD = sH + 2 * 16 * kH4CS;                          // [2][16 ch][kH4DP]
    uint4* const sWE = (uint4*)(sD + 2 * 16 * kH4DP);               // [slots][5 K steps][hi, lo][64 lanes]
    float* const sPar = (float*)(sWE + kF4WSlots * 640);            // [slots][16 ch][12]
    float* const sBN = sPar + kF4PSlots * (kF4ParB / 4);            // [scale 320 | shift 320] of the projection (epilogue)
    uint4* const sXH = (uint4*)(sBN + 2 * kH4Cout);                 // [5 K steps][hi, lo][64 lanes]: input fragments of the halo sub-row
#if IVF_H4_WALK
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));      // opaque per tile: otherwise every lane-dependent constant of the body is hoisted out of the tile walk and kept live across it (spills)
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
#else
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#endif
    const int L = (int)(blockIdx.x & 7) * perX_ + slot_;                              // consecutive L on one XCD (nT = 32 * images)
    const int b = L >> 5, py = (L >> 3) & 3, px = (L >> 1) & 3, half = L & 1;
    const int r0 = 8 * half;                                        // first own sub-row
    const int haloRow = half ? 7 : 8, haloSlot = half ? 0 : 9;      // row slot s holds sub-row r0 - 1 + s

    const unsigned ldsBase = (unsigned)(uintptr_t)f4smem;
    const unsigned ldsWE = ldsBase + (unsigned)((uint8_t*)sWE - (uint8_t*)f4smem), ldsPar = ldsBase + (unsigned)((uint8_t*)sPar - (uint8_t*)f4smem);
#if IVF_H4_DMA_SADDR      // r05: wave-uniform base in an SGPR pair + ONE 32-bit lane offset (16 lane): the per-lane 64-bit pointers of WE / WP / par (6 registers, kept live across the
                       // whole kernel) and their 64-bit vector adds per piece are gone
    const unsigned voff16 = (unsigned)lane * 16u;
    auto dma16 = [voff16](const void* sbase, unsigned ldsAddr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff16), "s"(sbase), "s"(ldsAddr) : "memory");
    };
#undef IVF_DMA_LANE
#define IVF_DMA_LANE(x) 0
#else
    auto dma16 = [](const void* src, unsigned ldsAddr) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(ldsAddr) : "memory");
    };
#undef IVF_DMA_LANE
#define IVF_DMA_LANE(x) (x)
#endif
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    auto piece = [&](int it, int c) {           // piece c of what interval `it` consumes: WE[it] (c < 10), par[it] (c = 10)
        if (it >= g1) return;
        if (c < 10) dma16(WE + ((size_t)it * 10 + c) * 64 + IVF_DMA_LANE(lane), ldsWE + (unsigned)((it % kF4WSlots) * 640 + c * 64) * 16u);
        else if (c == 10 && lane < 48) dma16(par + (size_t)it * 192 + IVF_DMA_LANE(lane * 4), ldsPar + (unsigned)((it % kF4PSlots) * kF4ParB));
    };
    auto dma = [&](int it) { piece(it, uwave); if (uwave < 3) piece(it, uwave + 8); };
    auto dma_late = [&](int it) {               // the same 11 pieces, most of them by waves 0-3, which reach the barrier first
#if IVF_H4_DMA_SKIP0            // wave 0 also expands the halo row: the pieces go to waves 1-3 (4 + 4 + 3)
        if (uwave >= 1 && uwave < 4) {
#pragma unroll
            for (int r = 0; r < 4; r++) { const int c = __builtin_amdgcn_readfirstlane((uwave - 1) + 3 * r); if (c < 11) piece(it, c); }
        }
        return;
#endif
        constexpr int NA = IVF_H4_DMA_A, NB = 4 * NA >= 11 ? 0 : (11 - 4 * NA + 3) / 4;
        if (uwave < 4) {
#pragma unroll
            for (int r = 0; r < NA; r++) { const int c = uwave + 4 * r; if (c < 11) piece(it, c); }
        } else {
#pragma unroll
            for (int r = 0; r < NB; r++) { const int c = 4 * NA + (uwave - 4) + 4 * r; if (c < 11) piece(it, c); }
        }
    };
    dma(0);
    dma(1);

    // ---- input: this wave's own sub-row (r0 + wave) as B fragments of the 16x16x32 MFMA (lane: column n = lane & 15, k = 8 (lane >> 4) + j);
    // wave 7 also gathers the halo sub-row and leaves its split fragments in LDS for wave 0
    HFrag bh[5], bl[5];
    {
        float xv[5][8], xh[5][8];
        size_t xb; int cs;
        lay_addr(layIn, kF4Cin, b, 4 * (r0 + wave) + py, 4 * (lane & 15) + px, xb, cs);
        const float* Xp = X + xb + (size_t)(8 * (lane >> 4)) * cs;
#pragma unroll
        for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
            for (int j = 0; j < 8; j++) xv[s5][j] = Xp[(size_t)(32 * s5 + j) * cs];
        if (uwave == 7) {
            lay_addr(layIn, kF4Cin, b, 4 * haloRow + py, 4 * (lane & 15) + px, xb, cs);
            const float* Xq = X + xb + (size_t)(8 * (lane >> 4)) * cs;
#pragma unroll
            for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
                for (int j = 0; j < 8; j++) xh[s5][j] = Xq[(size_t)(32 * s5 + j) * cs];
        }
        for (int i = tid; i < 2 * kH4Cout; i += 512) sBN[i] = i < kH4Cout ? scP[i] : shP[i - kH4Cout];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
            for (int jj = 0; jj < 4; jj++) split_pair(xv[s5][2 * jj], xv[s5][2 * jj + 1], bh[s5].u[jj], bl[s5].u[jj]);
        if (uwave == 7) {
            HFrag hh[5], hl[5];
#pragma unroll
            for (int s5 = 0; s5 < 5; s5++) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++) split_pair(xh[s5][2 * jj], xh[s5][2 * jj + 1], hh[s5].u[jj], hl[s5].u[jj]);
                if constexpr (!FP6) { sXH[(2 * s5) * 64 + lane] = hh[s5].q; sXH[(2 * s5 + 1) * 64 + lane] = hl[s5].q; }
            }
            if constexpr (FP6) {
                // halo sub-row in the FP6 form: slots 0-4 the hi fragments, 5-7 dwords 0-3 of the three fp6 operands, then [c][lane] uint2 (dwords 4-5),
                // then the lane's three scale bytes
                i32x6 q[3]; int sb = 0;
                fp6_pack(hh, hl, q, sb);
#pragma unroll
                for (int s5 = 0; s5 < 5; s5++) sXH[s5 * 64 + lane] = hh[s5].q;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    sXH[(5 + c) * 64 + lane] = make_uint4((unsigned)q[c][0], (unsigned)q[c][1], (unsigned)q[c][2], (unsigned)q[c][3]);
                    ((uint2*)(sXH + 8 * 64))[c * 64 + lane] = make_uint2((unsigned)q[c][4], (unsigned)q[c][5]);
                }
                ((int*)(sXH + 9 * 64 + 32))[lane] = sb;
            }
        }
    }
    i32x6 xq[3]; int sbB = 0;
    if constexpr (FP6) fp6_pack(bh, bl, xq, sbB);
    f32x16 pacc[5];
#pragma unroll
    for (int t = 0; t < 5; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) pacc[t][q] = 0.f;
    const int pt = wave & 3, oh = wave >> 2;                        // P: pixel tile (sub-rows r0 + 2 pt, + 1), output half (tiles 5 oh .. 5 oh + 4)

    // projection fragments of this wave's five tiles, straight from L2: group gp's live in pw while P(gp) runs, then tile by tile those of gp + 1
    HFrag pw[5][2];
    const uint4* const wpBase = WP + ((size_t)(5 * oh) * 2) * 64 + lane;
    auto wp_load = [&](int gp, int t) {
        const int g = min(max(gp, 0), g1 - 1);                      // intervals outside [0, g1) load valid, unused fragments: the count per interval stays 10
        const uint4* p = wpBase + ((size_t)g * kH4TilesP + t) * 2 * 64;
        pw[t][0].q = p[0]; pw[t][1].q = p[64];
    };
#pragma unroll
    for (int t = 0; t < 5; t++) wp_load(-2, t);                     // P(-2) runs on zeroed B operands: any finite fragments do

    // stencil thread: channel sch (0..15), own row srow (0..7), quarter row sq (4 pixels)
    const int sch = tid >> 5, srow = (tid >> 2) & 7, sq = tid & 3;
    const bool topOk = !(half == 0 && srow == 0), botOk = !(half == 1 && srow == 7);          // sub-rows -1 / 16 are the depthwise layer's zero padding
    const float rowM0 = topOk ? 1.f : 0.f, rowM2 = botOk ? 1.f : 0.f;
    const int srowT = sch * kH4CS + (topOk ? srow : srow + 1) * kF4HP + 4 * sq, srowM = sch * kH4CS + (srow + 1) * kF4HP + 4 * sq,
              srowB = sch * kH4CS + (botOk ? srow + 2 : srow + 1) * kF4HP + 4 * sq;
    const float mL = sq > 0 ? 1.f : 0.f, mR = sq < 3 ? 1.f : 0.f;   // the pixel beside the quarter comes from lane -1 / +1 of the same image row

    for (int i = tid; i < 2 * 16 * kH4DP / 4; i += 512) ((uint4*)sD)[i] = make_uint4(0u, 0u, 0u, 0u);      // P(-2), P(-1): zero operands
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef IVF_F4_TIMING
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    struct MPre { float dv[8]; float2 eb[4]; HFrag ea0[2]; };
    auto mfma_pre = [&](int it, MPre& m) {      // every LDS read of the phase that depends on no MFMA
        const int cur = it & 1, ws = it % kF4WSlots;
        const float* dB = sD + cur * (16 * kH4DP) + (8 * (lane >> 5)) * kH4DP + 32 * pt + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; j++) m.dv[j] = dB[j * kH4DP];
        const float* pp = sPar + (it % kF4PSlots) * (kF4ParB / 4) + (4 * (lane >> 4)) * 12 + 10;
#pragma unroll
        for (int r = 0; r < 4; r++) m.eb[r] = *(const float2*)(pp + r * 12);
        const uint4* wE = sWE + ws * 640 + lane;
        m.ea0[0].q = wE[0]; m.ea0[1].q = wE[64];
    };
    auto mfma_main = [&](int it, MPre& m) {
        const int cur = it & 1, ws = it % kF4WSlots;
        const uint4* wE = sWE + ws * 640 + lane;
        HFrag ea[2][2], ph, pl;
        ea[0][0] = m.ea0[0]; ea[0][1] = m.ea0[1];
        uint4 q6[2]; uint2 r6[2];                   // FP6: the correction operands of the current / next step
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
        const bool haloWave = uwave == IVF_H4_HALO_WAVE;
        const int gp = it - 2;
        F4_TIM(4);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) split_pair(m.dv[2 * jj], m.dv[2 * jj + 1], ph.u[jj], pl.u[jj]);      // P's B fragment
        // E(it) and P(it - 2) step by step: K step s5 of the expansion, then output tile s5 of the projection -- the expansion's A
        // fragments of step s5 + 1 (LDS, requested at the top of step s5) then have the projection tile's 96 matrix cycles to arrive
        // (r05 phase timers: with the five E steps back to back, one step = 48 matrix cycles ahead, the E loop took 900-1200 cycles for
        // its 240-480)
#pragma unroll
        for (int s5 = 0; s5 < 5; s5++) {
            if constexpr (FP6) {
                // slot layout (make_fused4, dWE6): pieces 0-4 the hi fragments of the five K steps, 5-7 dwords 0-3 of the three bf6 correction operands,
                // 8-9 their dwords 4-5 ([c][lane] uint2).  Step s5 < 3 also issues correction c = s5 (operand requested a step ahead)
                const uint2* wE2 = (const uint2*)(sWE + ws * 640 + 8 * 64) + lane;
                if (s5 + 1 < 5) ea[(s5 + 1) & 1][0].q = wE[(s5 + 1) * 64];
                if (s5 == 0) { q6[0] = wE[5 * 64]; r6[0] = wE2[0]; }
                if (s5 + 1 < 3) { q6[(s5 + 1) & 1] = wE[(6 + s5) * 64]; r6[(s5 + 1) & 1] = wE2[(s5 + 1) * 64]; }
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea[s5 & 1][0].v, bh[s5].v, e0, 0, 0, 0);
                if (haloWave) {
                    HFrag xh_; xh_.q = sXH[s5 * 64 + lane];
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ea[s5 & 1][0].v, xh_.v, e1, 0, 0, 0);
                }
                if (s5 < 3) {
                    const int c = s5;
                    const i32x8 a6 = {(int)q6[c & 1].x, (int)q6[c & 1].y, (int)q6[c & 1].z, (int)q6[c & 1].w, (int)r6[c & 1].x, (int)r6[c & 1].y, 0, 0};
                    const i32x8 b0 = {xq[c][0], xq[c][1], xq[c][2], xq[c][3], xq[c][4], xq[c][5], 0, 0};
                    if (c == 0) e0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b0, e0, 3, 2, 0, 127 - 10 - kF6SH, 0, sbB);
                    if (c == 1) e0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b0, e0, 3, 2, 0, 127 - 10 - kF6SH, 1, sbB);
                    if (c == 2) e0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b0, e0, 3, 2, 0, 127 - 10 - kF6SH, 2, sbB);
                    if (haloWave) {
                        const uint4 hq = sXH[(5 + c) * 64 + lane]; const uint2 hr = ((const uint2*)(sXH + 8 * 64))[c * 64 + lane];
                        const int hsb = ((const int*)(sXH + 9 * 64 + 32))[lane];
                        const i32x8 b1 = {(int)hq.x, (int)hq.y, (int)hq.z, (int)hq.w, (int)hr.x, (int)hr.y, 0, 0};
                        if (c == 0) e1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b1, e1, 3, 2, 0, 127 - 10 - kF6SH, 0, hsb);
                        if (c == 1) e1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b1, e1, 3, 2, 0, 127 - 10 - kF6SH, 1, hsb);
                        if (c == 2) e1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a6, b1, e1, 3, 2, 0, 127 - 10 - kF6SH, 2, hsb);
                    }
                }
            } else {
            if (s5 + 1 < 5) {
                if (IVF_H4_ABL & 2) { ea[(s5 + 1) & 1][0] = ea[s5 & 1][0]; ea[(s5 + 1) & 1][1] = ea[s5 & 1][1]; }
                else { ea[(s5 + 1) & 1][0].q = wE[(2 * s5 + 2) * 64]; ea[(s5 + 1) & 1][1].q = wE[(2 * s5 + 3) * 64]; }
            }
            if (!(IVF_H4_ABL & 64))
            {   // E(it): hidden group `it` = W_E[16 x 160] . X[160 x 16 pixels of this wave's sub-row]
                const HFrag &ah = ea[s5 & 1][0], &al = ea[s5 & 1][1];
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.v, bh[s5].v, e0, 0, 0, 0);
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bl[s5].v, e0, 0, 0, 0);
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bh[s5].v, e0, 0, 0, 0);
                if (haloWave && !(IVF_H4_ABL & 16)) {                     // wave-uniform: the sub-row just outside the half, fragments from LDS
                    HFrag xh_, xl_;                 // (requested a K step ahead like the A fragments: 1,881-1,887 vs 1,841-1,868 us, not kept)
                    xh_.q = sXH[(2 * s5) * 64 + lane]; xl_.q = sXH[(2 * s5 + 1) * 64 + lane];
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.v, xh_.v, e1, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, xl_.v, e1, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, xh_.v, e1, 0, 0, 0);
                }
            }
            }
            {   // P(it - 2), tile s5: out[32 channels x 32 pixels] += W_P[32 x 16] . D[16 x 32 pixels]
                const int t = s5;
                const HFrag &ah = pw[t][0], &al = pw[t][1];
                if (!(IVF_H4_ABL & 32)) {
                pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, ph.v, pacc[t], 0, 0, 0);
                pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, pl.v, pacc[t], 0, 0, 0);
                pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, ph.v, pacc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(IVF_H4_ABL & 1)) wp_load(gp + 1, t);                 // the same tile of the next group, behind the MFMAs that read this one
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        F4_TIM(5);
        // E's epilogue: BN + ReLU6 -> planes.  C layout of E: column = lane & 15 (sub-column), row = 4 (lane >> 4) + r (hidden channel of the group)
        float* hp = sH + cur * (16 * kH4CS) + (4 * (lane >> 4)) * kH4CS + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            hp[r * kH4CS + (wave + 1) * kF4HP] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e0[r], m.eb[r].x, m.eb[r].y), 0.f, 6.f);
            if (haloWave) hp[r * kH4CS + haloSlot * kF4HP] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e1[r], m.eb[r].x, m.eb[r].y), 0.f, 6.f);
        }
        F4_TIM(6);
    };
    auto stencil_phase = [&](int it) {          // S(it - 1): 3x3 on the 8 own rows of group it - 1, + BN + ReLU6
        const int g = it - 1;
        if (g < 0 || g >= g1 || (IVF_H4_ABL & 8)) return;
        const float* hp = sH + (g & 1) * (16 * kH4CS);
        const float4* pq = (const float4*)(sPar + (g % kF4PSlots) * (kF4ParB / 4) + sch * 12);
        const float4 w03 = pq[0], w47 = pq[1], w8s = pq[2];          // taps 0-3 | 4-7 | tap 8, shift, (expansion BN)
        float o[4];
#pragma unroll
        for (int p4 = 0; p4 < 4; p4++) o[p4] = w8s.y;
        const int ro[3] = {srowT, srowM, srowB};
        const float wk[9] = {w03.x * rowM0, w03.y * rowM0, w03.z * rowM0, w03.w, w47.x, w47.y, w47.z * rowM2, w47.w * rowM2, w8s.x * rowM2};
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const float4 a = *(const float4*)(hp + ro[ky]);
            const float own[4] = {a.x, a.y, a.z, a.w};
            const float w0 = wk[3 * ky], w1 = wk[3 * ky + 1], w2 = wk[3 * ky + 2];
            const float w0L = w0 * mL, w2R = w2 * mR;
#pragma unroll
            for (int p4 = 0; p4 < 4; p4++) {
                o[p4] = __builtin_fmaf(own[p4], w1, o[p4]);
                if (p4 > 0) o[p4] = __builtin_fmaf(own[p4 - 1], w0, o[p4]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p4]) : "v"(own[3]), "v"(w0L));
                if (p4 < 3) o[p4] = __builtin_fmaf(own[p4 + 1], w2, o[p4]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p4]) : "v"(own[0]), "v"(w2R));
            }
        }
        float* dp = sD + (g & 1) * (16 * kH4DP) + sch * kH4DP + srow * 16 + 4 * sq;
        *(float4*)dp = make_float4(__builtin_amdgcn_fmed3f(o[0], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[1], 0.f, 6.f),
                                   __builtin_amdgcn_fmed3f(o[2], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[3], 0.f, 6.f));
    };

    IVF_PRIO_STATIC();
    for (int it = 0; it < g1 + 2; it++) {
        F4_TIM(0);
        {
            MPre m;
            mfma_pre(it, m); __builtin_amdgcn_sched_barrier(0);
            if (wave < 4) { IVF_PRIO_MFMA(1); mfma_main(it, m); IVF_PRIO_MFMA(0); F4_TIM(1); IVF_PRIO_STEN(1); stencil_phase(it); IVF_PRIO_STEN(0); F4_TIM(2); }
            else { IVF_PRIO_STEN(1); stencil_phase(it); IVF_PRIO_STEN(0); F4_TIM(2); __builtin_amdgcn_sched_barrier(0); IVF_PRIO_MFMA(1); mfma_main(it, m); IVF_PRIO_MFMA(0); F4_TIM(1); }
        }
        // everything older than this interval's ten projection-fragment loads has landed: the DMA pieces of it + 1, requested an interval ago
        if (IVF_H4_ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        F4_TIM(3);
        if (!(IVF_H4_ABL & 4)) dma_late(it + 2);
        F4_TIM(0);                               // land during it + 1; their slots were last read in it - 1
        __syncthreads();
        F4_TIM(3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if defined(IVF_F4_TIMING) && !defined(IVF_F4_TIM_D4ONLY)
    if (lane == 0 && (wave == 0 || wave == 4)) {
        for (int i = 0; i < 7; i++) atomicAdd(&g_f4Tim[(wave ? 8 : 0) + i], tacc[i]);
        atomicAdd(&g_f4Tim[(wave ? 8 : 0) + 7], 1ull);
    }
#endif

    // ---- epilogue: BN of the projection; the sub-image's pixels are 4 apart in planes (layOut 0: the decoder reads planes)
    const int n = lane & 31;
    const int oy = 4 * (r0 + 2 * pt + (n >> 4)) + py, ox = 4 * (n & 15) + px;      // this lane's pixel
    size_t ob; int ocs;
    lay_addr(layOut, kH4Cout, b, oy, ox, ob, ocs);
    float* const yp = Y + ob + (size_t)(160 * oh + 4 * (lane >> 5)) * ocs;
    float amaxOut = 0.f;
#pragma unroll
    for (int t = 0; t < 5; t++) {
        float4 sc4[4], sh4[4];
        const int cb = 160 * oh + t * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(sBN + cb + 8 * g4); sh4[g4] = *(const float4*)(sBN + kH4Cout + cb + 8 * g4); }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const float v = pacc[t][q] * vget<4>(sc4[q >> 2], q & 3) + vget<4>(sh4[q >> 2], q & 3);
            range_note(amaxOut, v);
            yp[(size_t)(t * 32 + (q & 3) + 8 * (q >> 2)) * ocs] = v;
        }
    }
    range_flag(amaxOut);
    } while (IVF_H4_WALK && (slot_ += strideX_) < perX_);      // tiles of this workgroup
}

#ifdef IVF_EXPERIMENT      // opt-in variant (IVF_FCN_ROLES=1) with a run-time ablation mask: experiment builds only
// ---- k_fcn_irbd4w (r04): the same block with ROLE-SPECIALISED waves ----
// k_fcn_irbd4 runs eight waves that each hold the input fragments (80 registers) AND the projection accumulators (80): two waves per
// SIMD, whose MFMA phase, stencil phase and weight requests add up instead of overlapping (interval = MFMA cycles of ONE wave + ~2,200;
// matrix pipe busy 0.41-0.48).  Here a 1,024-thread workgroup has sixteen waves of <= 128 registers, four per SIMD:
//   waves 0-7   E role: hold the input fragments; per interval E(it) = 30 x v_mfma_f32_16x16x32_f16 + BN + ReLU6 -> sH
//   waves 8-15  P role: hold the accumulators; per interval P(it - 2) = 15 x v_mfma_f32_32x32x16_f16; request the weights by LDS-DMA;
//               prologue (LDS set-up) and epilogue (residual, BN, stores)
//   all sixteen S(it - 1): wave w = hidden channel w of the group, lane = (sub-row, quarter row of 4 pixels)
// Each role is its own loop (the register allocator sees the fragments and the accumulators in different branches); both loops meet at
// ONE s_barrier per interval.  Every SIMD hosts two E and two P waves: each needs the matrix pipe for 480 of the interval's cycles, so
// the pipe's 1,920 busy cycles come from four instruction streams instead of two, and the stencil's VALU work is spread over all of them.
__device__ int g_wAbl = 0;      // timing experiments only (IVF_FCN_WABL; results wrong): 1 no stencil, 2 no weight DMA in the loop, 4 E fragments read once,
                                // 8 no E MFMAs, 16 no P MFMAs, 32 no E epilogue, 64 P fragments read once, 128 no barrier in the loop
template <bool RES, bool SPLIT = false>
__global__ __launch_bounds__(1024) void k_fcn_irbd4w(const float* __restrict__ X, const uint4* __restrict__ WE, const float* __restrict__ par,
                                                    const uint4* __restrict__ WP, const float* __restrict__ scP, const float* __restrict__ shP,
                                                    const float* __restrict__ res, float* __restrict__ Y, int Cout, int tilesP, float* __restrict__ part,
                                                    int layIn, int layOut)
{
    const int g0 = SPLIT ? (int)(blockIdx.z * kF4Groups / gridDim.z) : 0, g1 = SPLIT ? (int)((blockIdx.z + 1) * kF4Groups / gridDim.z) : kF4Groups;
    extern __shared__ __attribute__((aligned(16))) uint4 f4smem[];
    float* const sH = (float*)f4smem;                               // [2][16 ch][kF4CS >= 16 rows x kF4HP]
    float* const sD = sH + 2 * 16 * kF4CS;                          // [2][16 ch][kF4DP]
    uint4* const sWE = (uint4*)(sD + 2 * 16 * kF4DP);               // [slots][5 K steps][hi, lo][64 lanes]
    uint4* const sWP = sWE + kF4WSlots * 640;                       // [slots][5 tiles][hi, lo][64 lanes]
    float* const sPar = (float*)(sWP + kF4WSlots * 640);            // [slots][16 ch][12]
    float* const sBN = sPar + kF4PSlots * (kF4ParB / 4);            // [scale 160 | shift 160] of the projection (epilogue)
    const int tid = threadIdx.x, lane = tid & 63;
    const int uwave = __builtin_amdgcn_readfirstlane(tid >> 6), w8 = uwave & 7;
    const int nwg = gridDim.x, L = (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8;     // consecutive L on one XCD (grid x = 16 * images)
    const int b = L >> 4, py = (L >> 2) & 3, px = L & 3;
    const int tile0 = blockIdx.y * 5;
    constexpr int HW = 4096;
    const int abl = g_wAbl;

    // ---- S(it - 1), all sixteen waves: 3x3 on the 16 x 16 plane of channel `uwave` of group it - 1, + BN + ReLU6.  Lane = (sub-row, quarter);
    // a DPP row of 16 lanes = four sub-rows x four quarters: the pixel beside a quarter comes from lane -1 / +1, none at the row's ends
    const int ssr = lane >> 2, sq = lane & 3;
    const float rowM0 = ssr > 0 ? 1.f : 0.f, rowM2 = ssr < 15 ? 1.f : 0.f;
    const int srow0 = uwave * kF4CS + (ssr > 0 ? ssr - 1 : ssr) * kF4HP + 4 * sq, srow1 = uwave * kF4CS + ssr * kF4HP + 4 * sq,
              srow2 = uwave * kF4CS + (ssr < 15 ? ssr + 1 : ssr) * kF4HP + 4 * sq;
    const float mL = sq > 0 ? 1.f : 0.f, mR = sq < 3 ? 1.f : 0.f;
    auto stencil_phase = [&](int it) {
        const int g = it - 1;
        if (g < g0 || g >= g1 || (abl & 1)) return;
        const float* hp = sH + (g & 1) * (16 * kF4CS);
        const float4* pq = (const float4*)(sPar + (g % kF4PSlots) * (kF4ParB / 4) + uwave * 12);
        const float4 w03 = pq[0], w47 = pq[1], w8s = pq[2];          // taps 0-3 | 4-7 | tap 8, shift, (expansion BN)
        float o[4] = {w8s.y, w8s.y, w8s.y, w8s.y};
        const int ro[3] = {srow0, srow1, srow2};
        const float wk[9] = {w03.x * rowM0, w03.y * rowM0, w03.z * rowM0, w03.w, w47.x, w47.y, w47.z * rowM2, w47.w * rowM2, w8s.x * rowM2};
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const float4 a = *(const float4*)(hp + ro[ky]);
            const float own[4] = {a.x, a.y, a.z, a.w};
            const float w0 = wk[3 * ky], w1 = wk[3 * ky + 1], w2 = wk[3 * ky + 2];
            const float w0L = w0 * mL, w2R = w2 * mR;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                o[p] = __builtin_fmaf(own[p], w1, o[p]);
                if (p > 0) o[p] = __builtin_fmaf(own[p - 1], w0, o[p]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p]) : "v"(own[3]), "v"(w0L));
                if (p < 3) o[p] = __builtin_fmaf(own[p + 1], w2, o[p]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p]) : "v"(own[0]), "v"(w2R));
            }
        }
        *(float4*)(sD + (g & 1) * (16 * kF4DP) + uwave * kF4DP + ssr * 16 + 4 * sq) =
            make_float4(__builtin_amdgcn_fmed3f(o[0], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[1], 0.f, 6.f),
                        __builtin_amdgcn_fmed3f(o[2], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[3], 0.f, 6.f));
    };

    if (uwave < 8) {
        // ================= E role =================
        HFrag bh[5][2], bl[5][2];
        {
            float xv[2][5][8];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                size_t xb; int cs;
                lay_addr(layIn, kF4Cin, b, 4 * (2 * w8 + u) + py, 4 * (lane & 15) + px, xb, cs);
                const float* Xp = X + xb + (size_t)(8 * (lane >> 4)) * cs;
#pragma unroll
                for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
                    for (int j = 0; j < 8; j++) xv[u][s5][j] = Xp[(size_t)(32 * s5 + j) * cs];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int s5 = 0; s5 < 5; s5++)
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) split_pair(xv[u][s5][2 * jj], xv[u][s5][2 * jj + 1], bh[s5][u].u[jj], bl[s5][u].u[jj]);
        }
        __syncthreads();
        for (int it = g0; it < g1 + 2; it++) {
            if (it < g1) {                      // E(it): hidden group `it` = W_E[16 x 160] . X[160 x 32 pixels of this wave]
                const int cur = it & 1, ws = it % kF4WSlots;
                const uint4* wE = sWE + ws * 640 + lane;
                const float* pp = sPar + (it % kF4PSlots) * (kF4ParB / 4) + (4 * (lane >> 4)) * 12 + 10;
                float2 eb[4];
#pragma unroll
                for (int r = 0; r < 4; r++) eb[r] = *(const float2*)(pp + r * 12);
                HFrag ea[2][2];
                ea[0][0].q = wE[0]; ea[0][1].q = wE[64];
                f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s5 = 0; s5 < 5; s5++) {
                    if (s5 + 1 < 5) {
                        if (abl & 4) { ea[(s5 + 1) & 1][0] = ea[s5 & 1][0]; ea[(s5 + 1) & 1][1] = ea[s5 & 1][1]; }
                        else { ea[(s5 + 1) & 1][0].q = wE[(2 * s5 + 2) * 64]; ea[(s5 + 1) & 1][1].q = wE[(2 * s5 + 3) * 64]; }
                    }
                    if (abl & 8) continue;
                    const HFrag &ah = ea[s5 & 1][0], &al = ea[s5 & 1][1];
                    e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.v, bh[s5][0].v, e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.v, bh[s5][1].v, e1, 0, 0, 0);
                    e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bl[s5][0].v, e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bl[s5][1].v, e1, 0, 0, 0);
                    e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bh[s5][0].v, e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bh[s5][1].v, e1, 0, 0, 0);
                }
                // C layout of E: column = lane & 15 (sub-column), row = 4 (lane >> 4) + r (hidden channel of the group)
                float* hp = sH + cur * (16 * kF4CS) + (4 * (lane >> 4)) * kF4CS + (2 * w8) * kF4HP + (lane & 15);
                if (!(abl & 32)) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        hp[r * kF4CS] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e0[r], eb[r].x, eb[r].y), 0.f, 6.f);
                        hp[r * kF4CS + kF4HP] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e1[r], eb[r].x, eb[r].y), 0.f, 6.f);
                    }
                } else if (e0[0] + e1[0] == 123.25f) hp[0] = e0[1];
            }
            stencil_phase(it);
            if (!(abl & 128)) __syncthreads();
        }
        return;
    }

    // ================= P role =================
    const unsigned ldsBase = (unsigned)(uintptr_t)f4smem;
    const unsigned ldsWE = ldsBase + (unsigned)((uint8_t*)sWE - (uint8_t*)f4smem), ldsWP = ldsBase + (unsigned)((uint8_t*)sWP - (uint8_t*)f4smem),
                   ldsPar = ldsBase + (unsigned)((uint8_t*)sPar - (uint8_t*)f4smem);
#if IVF_W4_DMA_SADDR      // r05: wave-uniform base in an SGPR pair + ONE 32-bit lane offset (16 lane): the per-lane 64-bit pointers of WE / WP / par (6 registers, kept live across the
                       // whole kernel) and their 64-bit vector adds per piece are gone
    const unsigned voff16 = (unsigned)lane * 16u;
    auto dma16 = [voff16](const void* sbase, unsigned ldsAddr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff16), "s"(sbase), "s"(ldsAddr) : "memory");
    };
#undef IVF_DMA_LANE
#define IVF_DMA_LANE(x) 0
#else
    auto dma16 = [](const void* src, unsigned ldsAddr) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(ldsAddr) : "memory");
    };
#undef IVF_DMA_LANE
#define IVF_DMA_LANE(x) (x)
#endif
    // Every piece is ALWAYS issued (a group index outside [g0, g1) is clamped: its slot is one nobody reads in that interval), so a wave's
    // count of outstanding requests is known: the three pieces of a wave are spread over its interval (before the stencil, before P, at
    // the end) and the wait in front of the barrier lets exactly the two youngest stay in flight.
    auto piece = [&](int it, int c) {           // piece c of what interval `it` consumes: WE[it] (c < 10), WP[it - 2] (c < 20), par[it] (c = 20)
        const int nb = it % kF4WSlots;
        if (c < 10) {
            const int ge = min(it, g1 - 1);
            dma16(WE + ((size_t)ge * 10 + c) * 64 + IVF_DMA_LANE(lane), ldsWE + (unsigned)(nb * 640 + c * 64) * 16u);
        } else if (c < 20) {
            const int c2 = c - 10, gp = min(max(it - 2, g0), g1 - 1);
            dma16(WP + (((size_t)gp * tilesP + tile0) * 2 + c2) * 64 + IVF_DMA_LANE(lane), ldsWP + (unsigned)(nb * 640 + c2 * 64) * 16u);
        } else if (c == 20) {
            const int ge = min(it, g1 - 1);
            if (lane < 48) dma16(par + (size_t)ge * 192 + IVF_DMA_LANE(lane * 4), ldsPar + (unsigned)((it % kF4PSlots) * kF4ParB));
        }
    };
    auto dma = [&](int it) {                    // all 21 pieces (<= 1 KB each), piece c by P wave c % 8
#pragma unroll
        for (int r = 0; r < 3; r++) piece(it, w8 + 8 * r);
    };
    dma(g0);
    dma(g0 + 1);
    if (tid - 512 < 160) { sBN[tid - 512] = scP[tile0 * 32 + tid - 512]; sBN[160 + tid - 512] = shP[tile0 * 32 + tid - 512]; }
    f32x16 pacc[5];
#pragma unroll
    for (int t = 0; t < 5; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) pacc[t][q] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = g0; it < g1 + 2; it++) {
        if (!(abl & 2)) piece(it + 2, w8);                       // lands during it + 1; the slots were last read in it - 1
        stencil_phase(it);
        if (!(abl & 2)) piece(it + 2, w8 + 8);
        if (it >= g0 + 2) {                     // P(it - 2): out[160 x 32 pixels] += W_P[160 x 16] . D[16 x 32 pixels]
            const int cur = it & 1, ws = it % kF4WSlots;
            const float* dB = sD + cur * (16 * kF4DP) + (8 * (lane >> 5)) * kF4DP + 32 * w8 + (lane & 31);
            float dv[8];
#pragma unroll
            for (int j = 0; j < 8; j++) dv[j] = dB[j * kF4DP];
            const uint4* wPq = sWP + ws * 640 + lane;
            HFrag pa[2][2], ph, pl;
            pa[0][0].q = wPq[0]; pa[0][1].q = wPq[64];
#pragma unroll
            for (int jj = 0; jj < 4; jj++) split_pair(dv[2 * jj], dv[2 * jj + 1], ph.u[jj], pl.u[jj]);
#pragma unroll
            for (int t = 0; t < 5; t++) {
                if (t + 1 < 5) {
                    if (abl & 64) { pa[(t + 1) & 1][0] = pa[t & 1][0]; pa[(t + 1) & 1][1] = pa[t & 1][1]; }
                    else { pa[(t + 1) & 1][0].q = wPq[(2 * t + 2) * 64]; pa[(t + 1) & 1][1].q = wPq[(2 * t + 3) * 64]; }
                }
                if (abl & 16) continue;
                const HFrag &ah = pa[t & 1][0], &al = pa[t & 1][1];
                pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, ph.v, pacc[t], 0, 0, 0);
                pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, pl.v, pacc[t], 0, 0, 0);
                pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, ph.v, pacc[t], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");        // everything but this interval's two pieces: the pieces of it + 1 have landed
        if (!(abl & 2)) piece(it + 2, w8 + 16);
        if (!(abl & 128)) __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue (P waves): BN (+ residual) of the projection.  128 registers: the residual tile by tile (two tiles in flight), BN
    // parameters straight from LDS
    const int n = lane & 31;
    const int oy = 4 * (2 * w8 + (n >> 4)) + py, ox = 4 * (n & 15) + px;      // this lane's pixel
    size_t rb = 0; int rcs = 0;
    if (RES && !SPLIT) lay_addr(layIn, Cout, b, oy, ox, rb, rcs);
    const float* const rp = res + rb + (size_t)(tile0 * 32 + 4 * (lane >> 5)) * rcs;
    size_t ob; int ocs;
    lay_addr(layOut, Cout, b, oy, ox, ob, ocs);
    float* const yp = (SPLIT ? part + (size_t)blockIdx.z * (nwg / 16) * Cout * HW : Y) + ob + (size_t)(tile0 * 32 + 4 * (lane >> 5)) * ocs;
    float rv[2][16];
    if (RES && !SPLIT) {
#pragma unroll
        for (int q = 0; q < 16; q++) rv[0][q] = rp[(size_t)((q & 3) + 8 * (q >> 2)) * rcs];
    }
#pragma unroll
    for (int t = 0; t < 5; t++) {
        if (RES && !SPLIT && t + 1 < 5) {
#pragma unroll
            for (int q = 0; q < 16; q++) rv[(t + 1) & 1][q] = rp[(size_t)((t + 1) * 32 + (q & 3) + 8 * (q >> 2)) * rcs];
        }
        const float* bn = sBN + t * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            float v = pacc[t][q];
            if (!SPLIT) {
                v = v * bn[(q & 3) + 8 * (q >> 2)] + bn[160 + (q & 3) + 8 * (q >> 2)];
                if (RES) v += rv[t & 1][q];
                if (!(fabsf(v) < 65504.f)) atomicOr(&g_fcnRange, 1);
            }
            yp[(size_t)(t * 32 + (q & 3) + 8 * (q >> 2)) * ocs] = v;
        }
    }
}
#endif  // IVF_EXPERIMENT

// the second half of a SPLIT launch: Y = (sum over the ranges, in index order) * scale + shift (+ residual); 4 pixels per thread
// part and Y are in the output layout `layOut`, the residual (the block's input) in `layIn`.  In every layout the four elements of
// a float4 are four pixels of ONE channel.
__global__ __launch_bounds__(256) void k_fcn_split_reduce(const float* __restrict__ part, int nSplit, size_t splitStride, int Cout,
                                                          const float* __restrict__ scP, const float* __restrict__ shP,
                                                          const float* __restrict__ res, float* __restrict__ Y, size_t total4,
                                                          int layIn, int layOut)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    float4 a = ((const float4*)part)[i];
    for (int s = 1; s < nSplit; s++) {
        const float4 v = ((const float4*)(part + (size_t)s * splitStride))[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const size_t e = 4 * i, perImg = (size_t)Cout * 4096;
    const int b = (int)(e / perImg);
    const size_t w = e % perImg;
    const int c = layOut ? (int)((w >> 8) % (size_t)Cout) : (int)(w >> 12);
    const float sc = scP[c], sh = shP[c];
    float4 o = make_float4(a.x * sc + sh, a.y * sc + sh, a.z * sc + sh, a.w * sc + sh);
    if (res) {
        float4 r;
        if (layIn == layOut) r = ((const float4*)res)[i];
        else {                                                       // the four pixels one by one through the input layout
            float rr[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int y, x;
                if (layOut) lay_pixel(layOut, (int)(w / ((size_t)Cout * 256)), (int)(w & 255) + k, y, x);
                else { y = (int)((w & 4095) >> 6); x = (int)(w & 63) + k; }
                size_t rb; int rcs;
                lay_addr(layIn, Cout, b, y, x, rb, rcs);
                rr[k] = res[rb + (size_t)c * rcs];
            }
            r = make_float4(rr[0], rr[1], rr[2], rr[3]);
        }
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    range_flag(fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
    ((float4*)Y)[i] = o;
}

// ---- blocks 5 - 14 (dilation 1 and 2 at 64 x 64) as ONE kernel each: expand CIN -> 6 CIN, depthwise 3x3, project -> COUT (+ residual) ----
// The scheme of k_fcn_irbd4 one dilation down: a depthwise 3x3 with dilation 2 on a 64 x 64 map is four independent plain 3x3
// convolutions on the 32 x 32 sub-images of equal (y mod 2, x mod 2).  A sub-image does not fit one workgroup's registers, so a
// workgroup owns a STRIP of 8 rows x 32 columns of one sub-image (256 output pixels, the full width: no halo left or right) and
// computes the expansion for the row above and the row below as well (10 rows: 25 % more expansion, 12 % more MFMA work overall --
// against the 576- / 384-channel hidden tensors going to HBM and back).  Rows outside the sub-image are the depthwise layer's zero
// padding: their hidden values are stored as zeros.
//   * wave w: output row w of the strip (32 pixels: two 16-pixel expansion blocks, one 32-pixel projection block); waves 4-7 also own
//     one 16-pixel block of a halo row each (waves 4, 5: the row above; 6, 7: the row below) -- every SIMD hosts one wave with three
//     expansion blocks and one with two;
//   * everything else as in k_fcn_irbd4: groups of 16 hidden channels, one barrier per interval, E(g) / S(g - 1) / P(g - 2) on
//     different LDS buffers, weights by LDS-DMA requested two intervals ahead (most pieces by waves 0-3, which have no halo block),
//     waves 0-3 [MFMA phase, stencil phase], waves 4-7 [stencil, MFMA], VALU placed behind the MFMAs.
//   * stencil thread = (channel, row, 8-pixel segment): the pixel left / right of a segment comes from the neighbouring lane (DPP).
// Geometry by dilation.  DIL 2: strip = 8 rows x 32 columns of a 32 x 32 sub-image, wave w = row w, halo blocks on waves 4-7.
// DIL 1 (blocks 5-7, ONE 64 x 64 "sub-image"): strip = 4 rows x 64 columns, wave w = half a row (row w >> 1, columns 32 (w & 1) ..),
// the two halo rows are eight 16-pixel blocks, one per wave.  PITCH = floats per plane row (+ 4 pad), CS = floats per channel plane
// (+ 4 / + 4: the four 16-lane groups of E's stores, 4 channels apart, start 16 banks apart).
#ifndef IVF_D2_HALO_LOW
#define IVF_D2_HALO_LOW 1     // r05: 542 / 330 / 622 / 348 -> 524 / 315 / 609 / 338 us per 128 images (<96,96> / <64,64> / <96,160> / <64,96>)
#endif
template <int CIN, int COUT, int DIL = 2>
struct D2Cfg {
    static constexpr int ROWS = DIL == 2 ? 8 : 4, COLS = DIL == 2 ? 32 : 64, PITCH = COLS + 4, CS = (ROWS + 2) * PITCH + 4;
    static constexpr int KS = CIN / 32, TILES = COUT / 32, HID = 6 * CIN, NG = HID / 16;
    static constexpr int NPE = 2 * KS, NPP = 2 * TILES, NP = NPE + NPP + 1;              // 1 KB pieces per interval (+ parameters)
    static constexpr int WSLOT = (NPE + NPP) * 64;                                        // uint4 per weight slot
    static constexpr bool WALK = IVF_D2_WALK_ALL || !(CIN == 96 && COUT == 160);                          // may run as a persistent grid (k_fcn_irbd2)
    static constexpr size_t LDS = (size_t)2 * 16 * CS * 4 + (size_t)2 * 16 * kF4DP * 4 + (size_t)3 * WSLOT * 16 + 4 * kF4ParB + 2 * COUT * 4;
};

template <int CIN, int COUT, bool RES, int DIL = 2, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void k_fcn_irbd2(const float* __restrict__ X, const uint4* __restrict__ WE, const float* __restrict__ par,
                                                     const uint4* __restrict__ WP, const float* __restrict__ scP, const float* __restrict__ shP,
                                                     const float* __restrict__ res, float* __restrict__ Y, float* __restrict__ part,
                                                     int layIn, int layOut, int nT)
{
    // nT = tiles of the launch (16 per image).  r05: the grid may be SMALLER than that -- one workgroup per CU that walks its share of the tiles (persistent form):
    // the stores of a tile's epilogue drain while the next tile's input gather and first LDS-DMA pieces are already in flight, and the per-tile dispatch of a
    // fresh 8-wave workgroup (register file + LDS of a whole CU) disappears.  grid = nT is the one-tile-per-workgroup form of r03 / r04.
    using C = D2Cfg<CIN, COUT, DIL>;
    constexpr int kD2CS = C::CS, PITCH = C::PITCH, ROWS = C::ROWS, COLS = C::COLS;
    constexpr int KS = C::KS, TILES = C::TILES, NG = C::NG, NPE = C::NPE, NPP = C::NPP, NP = C::NP, WSLOT = C::WSLOT;
    // SPLIT: this workgroup's contiguous range of hidden groups (see k_fcn_irbd4)
    const int g0 = SPLIT ? (int)(blockIdx.z * NG / gridDim.z) : 0, g1 = SPLIT ? (int)((blockIdx.z + 1) * NG / gridDim.z) : NG;
    extern __shared__ __attribute__((aligned(16))) uint4 d2smem[];
    // tiles of XCD x (= blockIdx % 8: consecutive tiles of an image share an L2): [x * nT / 8, (x + 1) * nT / 8), walked with the stride of the XCD's workgroups
    // (compile-time off for the SPLIT instances -- their grid is always nT -- and for <96,160>, whose 240 registers spill with the loop around them: 600 -> 671 us)
    constexpr bool kWalk = D2Cfg<CIN, COUT, DIL>::WALK && !SPLIT;
    const int perX_ = nT >> 3, strideX_ = (int)gridDim.x >> 3;
    int slot_ = (int)blockIdx.x >> 3;
    do {
#ifdef IVF_D2_TIMING      // diagnostic build: -DIVF_D2_TIMING=<CIN * 1000 + COUT> times that instance like IVF_F4_TIMING times k_fcn_irbd4
    constexpr bool kTimed = CIN * 1000 + COUT == IVF_D2_TIMING && !SPLIT;
    const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
    unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = tk0, tk1 = 0;
#define D2_TIM(i) do { if (kTimed) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    tacc[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define D2_TIM(i) do { } while (0)
#endif
    float* const sH = (float*)d2smem;                               // [2][16 ch][kD2CS]
    float* const sD = sH + 2 * 16 * kD2CS;                          // [2][16 ch][kF4DP]
    uint4* const sW = (uint4*)(sD + 2 * 16 * kF4DP);                // [3 slots][E: KS x (hi, lo) | P: TILES x (hi, lo)][64 lanes]
    float* const sPar = (float*)(sW + 3 * WSLOT);                   // [4 slots][16 ch][12]
    float* const sBN = sPar + 4 * (kF4ParB / 4);                    // [scale COUT | shift COUT] of the projection (epilogue)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));      // opaque per tile: otherwise every lane-dependent constant of the body is hoisted out of the tile walk and kept live across it (spills)
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const int nwg = nT, L = (int)(blockIdx.x & 7) * perX_ + slot_;                    // consecutive L on one XCD (nT = 16 * images)
    const int b = L >> 4, py = DIL == 2 ? (L >> 3) & 1 : 0, px = DIL == 2 ? (L >> 2) & 1 : 0, strip = DIL == 2 ? L & 3 : L & 15;
    constexpr int HW = 4096;
    const unsigned ldsBase = (unsigned)(uintptr_t)d2smem;
    const unsigned ldsW = ldsBase + (unsigned)((uint8_t*)sW - (uint8_t*)d2smem), ldsPar = ldsBase + (unsigned)((uint8_t*)sPar - (uint8_t*)d2smem);
#if IVF_D2_DMA_SADDR      // r05: wave-uniform base in an SGPR pair + ONE 32-bit lane offset (16 lane): the per-lane 64-bit pointers of WE / WP / par (6 registers, kept live across the
                       // whole kernel) and their 64-bit vector adds per piece are gone
    const unsigned voff16 = (unsigned)lane * 16u;
    auto dma16 = [voff16](const void* sbase, unsigned ldsAddr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff16), "s"(sbase), "s"(ldsAddr) : "memory");
    };
#undef IVF_DMA_LANE
#define IVF_DMA_LANE(x) 0
#else
    auto dma16 = [](const void* src, unsigned ldsAddr) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(ldsAddr) : "memory");
    };
#undef IVF_DMA_LANE
#define IVF_DMA_LANE(x) (x)
#endif
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    auto piece = [&](int it, int c) {           // piece c of what interval `it` consumes: WE[it] (c < NPE), WP[it - 2] (c < NPE + NPP), par[it]
        const int nb = it % 3;
        if (c < NPE) {
            if (it < g1) dma16(WE + ((size_t)it * NPE + c) * 64 + IVF_DMA_LANE(lane), ldsW + (unsigned)(nb * WSLOT + c * 64) * 16u);
        } else if (c < NPE + NPP) {
            const int gp = it - 2;
            if (gp >= g0 && gp < g1) dma16(WP + ((size_t)gp * NPP + (c - NPE)) * 64 + IVF_DMA_LANE(lane), ldsW + (unsigned)(nb * WSLOT + c * 64) * 16u);
        } else if (c == NPE + NPP) {
            if (it < g1 && lane < 48) dma16(par + (size_t)it * 192 + IVF_DMA_LANE(lane * 4), ldsPar + (unsigned)((it & 3) * kF4ParB));
        }
    };
    auto dma_all = [&](int it) {
#pragma unroll
        for (int r = 0; r < (NP + 7) / 8; r++) { const int c = uwave + 8 * r; if (c < NP) piece(it, c); }
    };
    auto dma_late = [&](int it) {               // waves 0-3 (no halo block, first at the barrier) take NA pieces each, waves 4-7 the rest
        constexpr int NA = NP / 4;              // measured NA - 1 / NA / NA + 1: 85.8 / 84.3 / 84.1 us per image of the whole network
        if (uwave < 4) {
#pragma unroll
            for (int r = 0; r < NA; r++) piece(it, uwave + 4 * r);
        } else {
#pragma unroll
            for (int r = 0; r < (NP - 4 * NA + 3) / 4; r++) { const int c = 4 * NA + (uwave - 4) + 4 * r; if (c < NP) piece(it, c); }
        }
    };
    dma_all(g0);
    dma_all(g0 + 1);

    // ---- input: blocks 0, 1 = this wave's row (columns 0-15, 16-31); block 2 (waves 4-7) = 16 pixels of a halo row
    // lane: column n = lane & 15, k = 8 (lane >> 4) + j; rows outside the sub-image read as zeros
    const int r0 = DIL == 2 ? 8 * strip + wave : 4 * strip + (wave >> 1);        // (sub-)image row of blocks 0, 1
    const int c0 = DIL == 2 ? 0 : 32 * (wave & 1);                                // their first column
    // r05: which half of the workgroup expands the four halo blocks (DIL 2).  Phase timers: with waves 4-7 (stencil first, MFMA phase last -- the half that
    // reaches the barrier last) waves 0-3 idled 960 of an interval's 3,040 cycles at the barrier (k_fcn_irbd2<64,64>); with waves 0-3 the halves arrive together
    const bool hasHalo = DIL == 1 || (IVF_D2_HALO_LOW ? wave < 4 : wave >= 4);
    const bool above = DIL == 2 ? (IVF_D2_HALO_LOW ? wave < 2 : wave < 6) : wave < 4;
    const int rH = above ? ROWS * strip - 1 : ROWS * strip + ROWS, cH = DIL == 2 ? 16 * (wave & 1) : 16 * (wave & 3);   // halo block
    constexpr int SUB = 64 / DIL;                                                 // rows / columns of a sub-image
    const bool haloIn = hasHalo && rH >= 0 && rH < SUB;
    // all of a lane's loads are requested before the first value is used (see k_fcn_irbd4)
    HFrag bh[KS][3], bl[KS][3];
    {
        float xv[3][KS][8];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int r = u < 2 ? r0 : rH, c = (u < 2 ? c0 + 16 * u : cH) + (lane & 15);
            const bool ok = u < 2 || haloIn;
            size_t xb; int cs;
            lay_addr(layIn, CIN, b, ok ? DIL * r + py : 0, ok ? DIL * c + px : 0, xb, cs);
            const float* Xp = X + xb + (size_t)(8 * (lane >> 4)) * cs;
#pragma unroll
            for (int s = 0; s < KS; s++)
#pragma unroll
                for (int j = 0; j < 8; j++) xv[u][s][j] = Xp[(size_t)(32 * s + j) * cs];
        }
        if (tid < COUT) { sBN[tid] = scP[tid]; sBN[COUT + tid] = shP[tid]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const bool ok = u < 2 || haloIn;
#pragma unroll
            for (int s = 0; s < KS; s++)
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
                    split_pair(ok ? xv[u][s][2 * jj] : 0.f, ok ? xv[u][s][2 * jj + 1] : 0.f, bh[s][u].u[jj], bl[s][u].u[jj]);
        }
    }
    f32x16 pacc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) pacc[t][q] = 0.f;

    // stencil thread: channel sch, output row srw of the strip, segment sg (pixels 8 sg .. 8 sg + 7); tap rows = plane rows srw .. srw + 2
    constexpr int SEGS = COLS / 8;
    const int sch = tid >> 5, srw = (tid & 31) / SEGS, sg = tid % SEGS;
    const int sbase = sch * kD2CS + srw * PITCH + 8 * sg;
    const float mL = sg > 0 ? 1.f : 0.f, mR = sg < SEGS - 1 ? 1.f : 0.f;   // the pixel beside a segment comes from lane -1 / +1; none at the row's ends

    for (int i = tid; i < 2 * 16 * kF4DP / 4; i += 512) ((uint4*)sD)[i] = make_uint4(0u, 0u, 0u, 0u);      // P(-2), P(-1): zero operands
    for (int i = tid; i < (SPLIT ? 3 : 2) * NPP * 64; i += 512)                                           // P fragments of slots g0 % 3, (g0 + 1) % 3
        sW[(i / (NPP * 64)) * WSLOT + NPE * 64 + i % (NPP * 64)] = make_uint4(0u, 0u, 0u, 0u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // hidden rows outside the sub-image are zero padding of the depthwise layer
    const float haloKeep = haloIn ? 1.f : 0.f;
    const int lrowH = above ? 0 : ROWS + 1;

    struct MPre { float dv[8]; float2 eb[4]; HFrag ea0[2], pa0[2]; };
    auto mfma_pre = [&](int it, MPre& m) {      // every LDS read of the phase that depends on no MFMA
        const int cur = it & 1, ws = it % 3;
        const float* dB = sD + cur * (16 * kF4DP) + (8 * (lane >> 5)) * kF4DP + 32 * wave + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; j++) m.dv[j] = dB[j * kF4DP];
        const float* pp = sPar + (it & 3) * (kF4ParB / 4) + (4 * (lane >> 4)) * 12 + 10;
#pragma unroll
        for (int r = 0; r < 4; r++) m.eb[r] = *(const float2*)(pp + r * 12);
        const uint4* wE = sW + ws * WSLOT + lane;
        const uint4* wPq = wE + NPE * 64;
        m.ea0[0].q = wE[0]; m.ea0[1].q = wE[64];
        m.pa0[0].q = wPq[0]; m.pa0[1].q = wPq[64];
    };
    auto mfma_main = [&](int it, MPre& m) {
        const int cur = it & 1, ws = it % 3;
        const uint4* wE = sW + ws * WSLOT + lane;
        const uint4* wPq = wE + NPE * 64;
        HFrag ea[2][2], pa[2][2], ph, pl;
        ea[0][0] = m.ea0[0]; ea[0][1] = m.ea0[1]; pa[0][0] = m.pa0[0]; pa[0][1] = m.pa0[1];
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; s++) {          // E(it), blocks 0 and 1: hidden group `it` = W_E[16 x CIN] . X[CIN x 32 pixels of this wave's row]
            if (s + 1 < KS) { ea[(s + 1) & 1][0].q = wE[(2 * s + 2) * 64]; ea[(s + 1) & 1][1].q = wE[(2 * s + 3) * 64]; }
            const HFrag &ah = ea[s & 1][0], &al = ea[s & 1][1];
            e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.v, bh[s][0].v, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.v, bh[s][1].v, e1, 0, 0, 0);
            e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bl[s][0].v, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bl[s][1].v, e1, 0, 0, 0);
            e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bh[s][0].v, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bh[s][1].v, e1, 0, 0, 0);
            // P's B fragment: the four f16 splits spread over the K steps, two VALU instructions behind each MFMA
#pragma unroll
            for (int j = 0; j < 4; j++) if (j * KS / 4 == s) split_pair(m.dv[2 * j], m.dv[2 * j + 1], ph.u[j], pl.u[j]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int i = 0; i < 6; i++) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // C layout of E: column = lane & 15, row = 4 (lane >> 4) + r (hidden channel of the group)
        float* hpl = sH + cur * (16 * kD2CS) + (4 * (lane >> 4)) * kD2CS + (lane & 15);
        if (DIL == 1 || (IVF_D2_HALO_LOW ? uwave < 4 : uwave >= 4)) {           // E(it), block 2: this wave's 16 pixels of a halo row (its fragments are read once more)
            f32x4 e2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; s++) {
                HFrag ah, al; ah.q = wE[(2 * s) * 64]; al.q = wE[(2 * s + 1) * 64];
                e2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.v, bh[s][2].v, e2, 0, 0, 0);
                e2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bl[s][2].v, e2, 0, 0, 0);
                e2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.v, bh[s][2].v, e2, 0, 0, 0);
            }
            float* hp2 = hpl + lrowH * PITCH + cH;
#pragma unroll
            for (int r = 0; r < 4; r++) hp2[r * kD2CS] = haloKeep * __builtin_amdgcn_fmed3f(__builtin_fmaf(e2[r], m.eb[r].x, m.eb[r].y), 0.f, 6.f);
        }
        __builtin_amdgcn_sched_barrier(0);
        float* hp = hpl + ((DIL == 2 ? wave : wave >> 1) + 1) * PITCH + c0;
        constexpr int RPS = TILES > 1 ? (4 + TILES - 2) / (TILES - 1) : 4;      // epilogue rows per projection step (steps 1 .. TILES - 1)
#pragma unroll
        for (int t = 0; t < TILES; t++) {       // P(it - 2): out[COUT x 32 pixels] += W_P[COUT x 16] . D[16 x 32 pixels]
            if (t + 1 < TILES) { pa[(t + 1) & 1][0].q = wPq[(2 * t + 2) * 64]; pa[(t + 1) & 1][1].q = wPq[(2 * t + 3) * 64]; }
            const HFrag &ah = pa[t & 1][0], &al = pa[t & 1][1];
            pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, ph.v, pacc[t], 0, 0, 0);
            pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, pl.v, pacc[t], 0, 0, 0);
            pacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, ph.v, pacc[t], 0, 0, 0);
            if (TILES == 1 || t >= 1) {         // E's epilogue of blocks 0, 1 under P's MFMAs: BN + ReLU6 -> planes
#pragma unroll
                for (int r = (TILES == 1 ? 0 : (t - 1) * RPS); r < (TILES == 1 ? 4 : t * RPS) && r < 4; r++) {
                    hp[r * kD2CS] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e0[r], m.eb[r].x, m.eb[r].y), 0.f, 6.f);
                    hp[r * kD2CS + 16] = __builtin_amdgcn_fmed3f(__builtin_fmaf(e1[r], m.eb[r].x, m.eb[r].y), 0.f, 6.f);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2 * RPS, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2 * RPS, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 2 * RPS, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stencil_phase = [&](int it) {          // S(it - 1): 3x3 on the 10 x 32 planes of group it - 1 -> 8 x 32, + BN + ReLU6
        const int g = it - 1;
        if (g < g0 || g >= g1) return;
        const float* hp = sH + (g & 1) * (16 * kD2CS) + sbase;
        const float4* pq = (const float4*)(sPar + (g & 3) * (kF4ParB / 4) + sch * 12);
        const float4 w03 = pq[0], w47 = pq[1], w8s = pq[2];          // taps 0-3 | 4-7 | tap 8, shift, (expansion BN)
        float o[8];
#pragma unroll
        for (int p8 = 0; p8 < 8; p8++) o[p8] = w8s.y;
        const float wk[9] = {w03.x, w03.y, w03.z, w03.w, w47.x, w47.y, w47.z, w47.w, w8s.x};
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const float4 a = *(const float4*)(hp + ky * PITCH), c4 = *(const float4*)(hp + ky * PITCH + 4);
            const float own[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
            const float w0 = wk[3 * ky], w1 = wk[3 * ky + 1], w2 = wk[3 * ky + 2];
            const float w0L = w0 * mL, w2R = w2 * mR;
#pragma unroll
            for (int p8 = 0; p8 < 8; p8++) {
                o[p8] = __builtin_fmaf(own[p8], w1, o[p8]);
                if (p8 > 0) o[p8] = __builtin_fmaf(own[p8 - 1], w0, o[p8]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p8]) : "v"(own[7]), "v"(w0L));
                if (p8 < 7) o[p8] = __builtin_fmaf(own[p8 + 1], w2, o[p8]);
                else asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(o[p8]) : "v"(own[0]), "v"(w2R));
            }
        }
        float* dp = sD + (g & 1) * (16 * kF4DP) + sch * kF4DP + srw * COLS + 8 * sg;
        *(float4*)dp = make_float4(__builtin_amdgcn_fmed3f(o[0], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[1], 0.f, 6.f),
                                   __builtin_amdgcn_fmed3f(o[2], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[3], 0.f, 6.f));
        *(float4*)(dp + 4) = make_float4(__builtin_amdgcn_fmed3f(o[4], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[5], 0.f, 6.f),
                                         __builtin_amdgcn_fmed3f(o[6], 0.f, 6.f), __builtin_amdgcn_fmed3f(o[7], 0.f, 6.f));
    };

#ifdef IVF_D2_TIMING
    tk1 = tlast = __builtin_amdgcn_s_memtime();
#endif
    IVF_PRIO_STATIC();
    for (int it = g0; it < g1 + 2; it++) {
        D2_TIM(0);
        {
            MPre m;
            mfma_pre(it, m); __builtin_amdgcn_sched_barrier(0);
            if (wave < 4) { IVF_PRIO_MFMA(1); mfma_main(it, m); IVF_PRIO_MFMA(0); D2_TIM(1); IVF_PRIO_STEN(1); stencil_phase(it); IVF_PRIO_STEN(0); D2_TIM(2); }
            else { IVF_PRIO_STEN(1); stencil_phase(it); IVF_PRIO_STEN(0); D2_TIM(2); __builtin_amdgcn_sched_barrier(0); IVF_PRIO_MFMA(1); mfma_main(it, m); IVF_PRIO_MFMA(0); D2_TIM(1); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the pieces of it + 1, requested an interval ago
        D2_TIM(3);
        dma_late(it + 2);                                        // land during it + 1; their slots were last read in it - 1
        D2_TIM(0);
        __syncthreads();
        D2_TIM(3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef IVF_D2_TIMING
    const unsigned long long tk2 = __builtin_amdgcn_s_memtime();
    if (kTimed && lane == 0 && (wave == 0 || wave == 4)) {
        for (int i = 0; i < 7; i++) atomicAdd(&g_f4Tim[(wave ? 8 : 0) + i], tacc[i]);
        atomicAdd(&g_f4Tim[(wave ? 8 : 0) + 7], 1ull);
    }
#endif

    // ---- epilogue: BN (+ residual) of the projection; pixel n of the wave's row -> image (2 r0 + py, 2 n + px)
    const int n = lane & 31;
    const int oy = DIL * r0 + py, ox = DIL * (c0 + n) + px;         // this lane's pixel
    // the residual values of all tiles are requested at once (the input fragments are dead), BN parameters from LDS
    // r05 (IVF_RES_FROM_FRAGS, see k_fcn_irbd4): the residual is this wave's own input fragments (blocks 0, 1 = its 32 pixels, CIN = COUT), transposed per tile
    // through a wave-private 4.6 KB piece of the hidden planes, which nobody touches after the loop's last barrier
    constexpr bool kResFrags = RES && !SPLIT && IVF_D2_RES_FROM_FRAGS && CIN == COUT;
    float* const sxw = (float*)d2smem + wave * (32 * 36);
    static_assert(!kResFrags || (size_t)2 * 16 * kD2CS * 4 >= (size_t)8 * 32 * 36 * 4, "the transposition scratch lives in the hidden planes");
    float rvAll[TILES][16];
    if (RES && !SPLIT && !kResFrags) {
        size_t rb; int rcs;
        lay_addr(layIn, COUT, b, oy, ox, rb, rcs);
        const float* rp = res + rb + (size_t)(4 * (lane >> 5)) * rcs;
#pragma unroll
        for (int t = 0; t < TILES; t++)
#pragma unroll
            for (int q = 0; q < 16; q++) rvAll[t][q] = rp[(size_t)(t * 32 + (q & 3) + 8 * (q >> 2)) * rcs];
        __builtin_amdgcn_sched_barrier(0);
    }
    size_t ob; int ocs;
    lay_addr(layOut, COUT, b, oy, ox, ob, ocs);
    float* const yp = (SPLIT ? part + (size_t)blockIdx.z * (nwg / 16) * COUT * HW : Y) + ob + (size_t)(4 * (lane >> 5)) * ocs;
    float amaxOut = 0.f;
#pragma unroll
    for (int t = 0; t < TILES; t++) {
        const int cb = t * 32 + 4 * (lane >> 5);
        float4 sc4[4], sh4[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) { sc4[g4] = *(const float4*)(sBN + cb + 8 * g4); sh4[g4] = *(const float4*)(sBN + COUT + cb + 8 * g4); }
        float rvT[16];
        if (kResFrags) {
            // fragment lane (column lane & 15 of block u, channels 32 t + 8 (lane >> 4) + j) -> [pixel][channel] -> accumulator lane (pixel lane & 31,
            // channels 32 t + 4 (lane >> 5) + (q & 3) + 8 (q >> 2)); a wave's LDS instructions execute in order, the fences only stop the compiler
            constexpr int tt = kResFrags ? 1 : 0;   // (keeps bh / bl out of instances that do not use them here)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                float f[8];
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    uint32_t hu = bh[t * tt][u].u[jj], lu = bl[t * tt][u].u[jj];
                    asm volatile("" : "+v"(hu), "+v"(lu));      // opaque: pins these sums to the epilogue (see k_fcn_irbd4)
                    const f16x2 h = __builtin_bit_cast(f16x2, hu), l = __builtin_bit_cast(f16x2, lu);
                    f[2 * jj] = (float)h[0] + (float)l[0]; f[2 * jj + 1] = (float)h[1] + (float)l[1];
                }
                float* d = sxw + (16 * u + (lane & 15)) * 36 + 8 * (lane >> 4);
                *(float4*)d = make_float4(f[0], f[1], f[2], f[3]); *(float4*)(d + 4) = make_float4(f[4], f[5], f[6], f[7]);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            const float* sr = sxw + n * 36 + 4 * (lane >> 5);
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) { const float4 r = *(const float4*)(sr + 8 * g4); rvT[4 * g4] = r.x; rvT[4 * g4 + 1] = r.y; rvT[4 * g4 + 2] = r.z; rvT[4 * g4 + 3] = r.w; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            float v = pacc[t][q];                   // SPLIT: raw sums of this range of hidden groups, in the OUTPUT layout
            if (!SPLIT) {
                v = v * vget<4>(sc4[q >> 2], q & 3) + vget<4>(sh4[q >> 2], q & 3);
                if (RES) v += kResFrags ? rvT[q] : rvAll[t][q];
                range_note(amaxOut, v);
            }
            yp[(size_t)(t * 32 + (q & 3) + 8 * (q >> 2)) * ocs] = v;
        }
    }
    if (!SPLIT) range_flag(amaxOut);            // SPLIT: k_fcn_split_reduce checks the finished sums
#ifdef IVF_D2_TIMING
    if (kTimed) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) {
            const unsigned long long tk3 = __builtin_amdgcn_s_memtime();
            atomicAdd(&g_f4Whole[0], tk1 - tk0); atomicAdd(&g_f4Whole[1], tk2 - tk1); atomicAdd(&g_f4Whole[2], tk3 - tk2); atomicAdd(&g_f4Whole[3], 1ull);
        }
    }
#endif
    } while (kWalk && (slot_ += strideX_) < perX_);      // tiles of this workgroup
}

// ---- conv_last 1x1 80 -> 1 + bias (models_light.py:196) ----
__global__ void k_fcn_last(const float* __restrict__ X, const float* __restrict__ w, float bias, float* __restrict__ Y,
                           int C, int HW)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (p >= HW) return;
    float acc = 0.f;
    for (int c = 0; c < C; c++) acc += w[c] * X[((size_t)b * C + c) * HW + p];
    Y[(size_t)b * HW + p] = acc + bias;
}

// ---- bilinear to out_size, logistic, u8 truncation (models_light.py:198-199, :25-26; stereo_kitti.cc:511) ----
#ifndef IVF_OUT_ROWS
#define IVF_OUT_ROWS 8
#endif
constexpr int kOutRows = IVF_OUT_ROWS;        // output rows per workgroup of k_fcn_out
__global__ __launch_bounds__(256) void k_fcn_out(const float* __restrict__ L, int lh, int lw, int oh, int ow, float sy_, float sx_,
                                                float* __restrict__ costF, uint8_t* __restrict__ costU, int* __restrict__ status, int rowsPerWg,
                                                size_t uImageStride, int uRowStride)       // the u8 map's strides (r06: it may be a pitched plane of the front end)
{
    // the last kernel of a forward: every earlier kernel of it has finished (stream order), so one thread moves the f16 range flag
    // they may have raised into the handle's status word
    if ((blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x) == 0) {
        const int v = atomicExch(&g_fcnRange, 0);
        if (v) atomicOr(status, v);
    }
    // a workgroup = 1024 pixels (4 adjacent per thread) of rowsPerWg consecutive output rows (kOutRows = 8 for batches; 98 -> 73 us per 128 images; 2 for the per-call path, which has a chip to fill with one image) (r05; one row before): the logit rows they read go through LDS once,
    // a thread's horizontal taps and weights are computed once for all its rows, and the launch has an eighth of the workgroups.  sy_ = (float)lh / (float)oh and
    // sx_ come from the host (the same IEEE f32 quotient, computed once instead of by two division sequences per thread); the u8 map leaves as one dword per
    // thread where the row allows it.  Per pixel the arithmetic and its order are those of the one-row form (bit-identical maps).
    __shared__ float rows[(kEnc / 8) * (kEnc / 8)];                  // at most the whole logit map (lh, lw <= kEnc / 8)
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, yA = blockIdx.y * rowsPerWg, yB = min(yA + rowsPerWg, oh), b = blockIdx.z;
    auto src_row = [&](int y, int& y0, int& y1, float& fy) {
        fy = sy_ * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
        y0 = (int)fy; if (y0 > lh - 1) y0 = lh - 1;
        y1 = y0 + (y0 < lh - 1);
    };
    int rFirst, rLast, t0, t1; float tf;
    src_row(yA, rFirst, t1, tf); src_row(yB - 1, t0, rLast, tf);
    if (rLast < rFirst) rLast = rFirst;
    const float* P = L + (size_t)b * lh * lw + (size_t)rFirst * lw;
    for (int i = threadIdx.x; i < (rLast - rFirst + 1) * lw; i += blockDim.x) rows[i] = P[i];
    __syncthreads();
    if (x4 >= ow) return;
    int x0[4], x1[4]; float lx0[4], lx1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x4 + k;
        float fx = sx_ * ((float)x + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
        x0[k] = (int)fx; if (x0[k] > lw - 1) x0[k] = lw - 1;
        x1[k] = x0[k] + (x0[k] < lw - 1);
        lx1[k] = fx - (float)x0[k]; lx0[k] = 1.f - lx1[k];
    }
    const int nv = min(4, ow - x4);
    for (int y = yA; y < yB; y++) {
        int y0, y1; float fy;
        src_row(y, y0, y1, fy);
        const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
        const float* rowT = rows + (y0 - rFirst) * lw;
        const float* rowB = rows + (y1 - rFirst) * lw;
        float c[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float top = rowT[x0[k]] * lx0[k] + rowT[x1[k]] * lx1[k];
            const float bot = rowB[x0[k]] * lx0[k] + rowB[x1[k]] * lx1[k];
            const float v = top * ly0 + bot * ly1;
            const float z = 20.f * (v - 0.5f);
            // logistic through the hardware exp2 / rcp (1 ulp each; the libm expf + IEEE division were most of this kernel's instructions)
            c[k] = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-z * 1.44269504088896341f));
        }
        const size_t o = ((size_t)b * oh + y) * ow + x4;
        if (costF) {
            if (nv == 4 && (o & 3) == 0) *(float4*)(costF + o) = make_float4(c[0], c[1], c[2], c[3]);
            else for (int k = 0; k < nv; k++) costF[o + k] = c[k];
        }
        if (costU) {
            const unsigned u0 = (unsigned)(uint8_t)(c[0] * 255.0f), u1 = (unsigned)(uint8_t)(c[1] * 255.0f),
                           u2 = (unsigned)(uint8_t)(c[2] * 255.0f), u3 = (unsigned)(uint8_t)(c[3] * 255.0f);
            const size_t ou = (size_t)b * uImageStride + (size_t)y * uRowStride + x4;
            if (nv == 4 && (ou & 3) == 0) *(unsigned*)(costU + ou) = u0 | (u1 << 8) | (u2 << 16) | (u3 << 24);
            else { const unsigned u[4] = {u0, u1, u2, u3}; for (int k = 0; k < nv; k++) costU[ou + k] = (uint8_t)u[k]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side: architecture walk, BN folding, A-fragment weight shuffle, launch plan
// ------------------------------------------------------------------------------------------------
struct Block { int inp, oup, t, stride, dil; bool res; };
const Block kBlocks[17] = {
    {32, 16, 1, 1, 1, false},
    {16, 24, 6, 2, 1, false}, {24, 24, 6, 1, 1, true},
    {24, 32, 6, 2, 1, false}, {32, 32, 6, 1, 1, true}, {32, 32, 6, 1, 1, true},
    {32, 64, 6, 1, 1, false},
    {64, 64, 6, 1, 2, true}, {64, 64, 6, 1, 2, true}, {64, 64, 6, 1, 2, true},
    {64, 96, 6, 1, 2, false}, {96, 96, 6, 1, 2, true}, {96, 96, 6, 1, 2, true},
    {96, 160, 6, 1, 2, false},
    {160, 160, 6, 1, 4, true}, {160, 160, 6, 1, 4, true},
    {160, 320, 6, 1, 4, false},
};

struct Gemm {            // one MFMA convolution
    int cin, cout, taps, nTiles, NT, PT, act;
    uint4* dWq;             // f16 hi / lo A fragments (see k_fcn_gemm)
    float *dScale, *dShift;
    uint4* dWq6 = nullptr;  // 3x3 only (r06): the operands of k_fcn_conv3x3_f6 -- hi fragments + bf6 correction operands per (unit, tile, dx)
};
struct Dw { int c, stride, dil; float *dW, *dScale, *dShift, *dPack; };   // dPack: 12 floats per channel (k_fcn_dwpw)

struct Reader {
    const float* p; size_t left;
    const float* take(size_t n) { if (n > left) return nullptr; const float* r = p; p += n; left -= n; return r; }
};

// IEEE binary16 <-> binary32 on the host (round to nearest even, subnormals kept): the weight halves of the split
uint16_t f32_to_f16(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));     // overflow / inf / nan
    if (x < 0x38800000u) {                                   // below the smallest normal half: subnormal or zero
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23);
        const uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
        const int shift = 126 - e;                           // 14 .. 24
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = (x - 0x38000000u) >> 13;
    const uint32_t rem = x & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(sign | r);
}
float f16_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 31; const uint32_t m = h & 0x3FFu;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 0x400u), e - 25);
    uint32_t x; memcpy(&x, &v, 4); x |= sign; memcpy(&v, &x, 4);
    return v;
}

void fold_bn(const float* g, const float* b, const float* m, const float* v, int c, std::vector<float>& sc, std::vector<float>& sh)
{
    sc.resize(c); sh.resize(c);
    for (int i = 0; i < c; i++) {
        const float inv = 1.0f / std::sqrt(v[i] + 1e-5f);
        sc[i] = g[i] * inv;
        sh[i] = b[i] - m[i] * sc[i];
    }
}

template <int PT, int NT, int TAPS>
void launch_gemm_t(const Gemm& g, const float* X, const float* res, float* Y, int H, int W, int B, hipStream_t s)
{
    const int HW = H * W;
    dim3 grid(HW / (128 * PT), g.nTiles / NT, B);
    hipLaunchKernelGGL((k_fcn_gemm<PT, NT, TAPS>), grid, dim3(256), 0, s, X, g.dWq, g.dScale, g.dShift, res, Y, g.cin,
                       g.cout, H, W, g.nTiles, g.act);
}
void launch_gemm(const Gemm& g, const float* X, const float* res, float* Y, int H, int W, int B, hipStream_t s)
{
    if (g.taps == 9) {
        static const bool old9 = IVF_EXP_ENV("IVF_FCN_OLD3X3") != nullptr;
        static const bool split9 = IVF_EXP_ENV("IVF_FCN_3X3_SPLIT") != nullptr;      // the r01 kernel: one workgroup per output-channel tile
        static const int dec6 = IVF_EXP_ENV("IVF_FCN_DEC6") ? atoi(IVF_EXP_ENV("IVF_FCN_DEC6")) : IVF_DEC_FP6;
        if (!old9 && !split9 && dec6 >= 2 && g.dWq6 && H == 64 && W == 64 && g.act == 2 && !res)
            hipLaunchKernelGGL(k_fcn_conv3x3_f6r, dim3(8 * B), dim3(256), 0, s, X, g.dWq6, g.dScale, g.dShift, Y, g.cin, g.cout,
                               (const float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr);
        else if (!old9 && !split9 && dec6 && g.dWq6 && H == 64 && W == 64 && g.act == 2 && !res)
            hipLaunchKernelGGL(k_fcn_conv3x3_f6, dim3(8 * B), dim3(256), 0, s, X, g.dWq6, g.dScale, g.dShift, Y, g.cin, g.cout,
                               (const float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr);
        else if (!old9 && !split9 && H == 64 && W == 64 && g.cin % 32 == 0 && g.act == 2 && !res && g.nTiles == 3)
            hipLaunchKernelGGL((k_fcn_conv3x3_all<3>), dim3(8 * B), dim3(256), 0, s, X, g.dWq, g.dScale, g.dShift, Y, g.cin, g.cout,
                               (const float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr);
        else if (!old9 && H == 64 && W == 64 && g.cin % 32 == 0 && g.act == 2 && !res)
            hipLaunchKernelGGL(k_fcn_conv3x3, dim3(8 * g.nTiles * B), dim3(256), 0, s, X, g.dWq, g.dScale, g.dShift, Y, g.cin, g.cout, g.nTiles);
        else launch_gemm_t<1, 3, 9>(g, X, res, Y, H, W, B, s);
        return;
    }
    if (g.PT == 4 && g.NT == 1) launch_gemm_t<4, 1, 1>(g, X, res, Y, H, W, B, s);
    else if (g.PT == 4 && g.NT == 2) launch_gemm_t<4, 2, 1>(g, X, res, Y, H, W, B, s);
    else if (g.PT == 4 && g.NT == 3) launch_gemm_t<4, 3, 1>(g, X, res, Y, H, W, B, s);
    else if (g.PT == 2 && g.NT == 1) launch_gemm_t<2, 1, 1>(g, X, res, Y, H, W, B, s);
    else if (g.PT == 2 && g.NT == 2) launch_gemm_t<2, 2, 1>(g, X, res, Y, H, W, B, s);
    else if (g.PT == 2 && g.NT == 3) launch_gemm_t<2, 3, 1>(g, X, res, Y, H, W, B, s);
    else if (g.PT == 1 && g.NT == 1) launch_gemm_t<1, 1, 1>(g, X, res, Y, H, W, B, s);
    else if (g.PT == 1 && g.NT == 2) launch_gemm_t<1, 2, 1>(g, X, res, Y, H, W, B, s);
    else launch_gemm_t<2, 5, 1>(g, X, res, Y, H, W, B, s);
}

// expansion with the B tile stationary in LDS; false = not applicable, caller uses k_fcn_gemm
bool launch_expand(const Gemm& g, const float* X, float* Y, int H, int W, int B, hipStream_t s)
{
    static const int mode = IVF_EXP_ENV("IVF_FCN_EXPAND") ? atoi(IVF_EXP_ENV("IVF_FCN_EXPAND")) : 4;     // 0 off, 2 / 4 = PXT
    static const int minCin = IVF_EXP_ENV("IVF_FCN_EXPAND_MINCIN") ? atoi(IVF_EXP_ENV("IVF_FCN_EXPAND_MINCIN")) : 64;
    const int tiles = (g.cout + 31) / 32, HW = H * W, K16 = (g.cin + 15) / 16;
    if (!mode || g.taps != 1 || g.act != 1 || g.nTiles != tiles || g.cin < minCin || tiles < 4) return false;
    int pxt = mode;
    static const bool bigLds = [] {                             // 160 input channels x 128 pixels need 80 KB of LDS
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fcn_expand<4, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    }();
    if ((size_t)K16 * pxt * 2048 > (bigLds ? 80 : 64) * 1024) pxt = 2;
    if (HW % (32 * pxt)) return false;
    const size_t lds = (size_t)K16 * pxt * 2048;
    // IVF_FCN_EXPAND_RING=1: full-depth A ring (a register slot per K step: every refill belongs to the NEXT tile).  It
    // helped the un-pipelined tile loop at 10 K steps (329 -> 314 us); with the epilogue folded into the next tile's K
    // loop the second accumulator set leaves no room for it (77 spills, 424 us), so the half-depth ring is the default.
    static const bool fullRing = IVF_EXP_ENV("IVF_FCN_EXPAND_RING") && atoi(IVF_EXP_ENV("IVF_FCN_EXPAND_RING")) == 1;
    static const bool bigLdsR = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fcn_expand<4, 10, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    }();
#define EXPAND(P, KSV) do { if (fullRing && bigLdsR && KSV == 10)                                                                          \
        hipLaunchKernelGGL((k_fcn_expand<P, KSV, 1>), dim3(HW / (32 * P), 1, B), dim3(256), lds, s, X, g.dWq, g.dScale, g.dShift, Y, g.cin,   \
                           g.cout, HW, g.nTiles);                                                                             \
    else hipLaunchKernelGGL((k_fcn_expand<P, KSV, 0>), dim3(HW / (32 * P), 1, B), dim3(256), lds, s, X, g.dWq, g.dScale, g.dShift, Y,       \
                            g.cin, g.cout, HW, g.nTiles); } while (0)
    if (pxt == 4 && K16 == 4) EXPAND(4, 4);
    else if (pxt == 4 && K16 == 6) EXPAND(4, 6);
    else if (pxt == 4 && K16 == 10) EXPAND(4, 10);
    else if (pxt == 2 && K16 == 10) EXPAND(2, 10);
    else return false;
#undef EXPAND
    return true;
}

// name of the kernel the last launch_dwpw of this thread dispatched (ivf_fcn_probe_info: the probe reports what it timed)
static thread_local char g_lastDwpw[96] = "";
// fused depthwise + projection for the 64x64 stride-1 stages; false = shape not covered, caller runs the two kernels
bool launch_dwpw(const Dw& d, const Gemm& g, const float* X, const float* res, float* Y, int H, int W, int B, hipStream_t s)
{
    static const bool off = IVF_EXP_ENV("IVF_FCN_NOFUSE") != nullptr;
    const int abl = kAbl;                                                   // ablation builds keep the 4-wave kernels
    const int tiles = (g.cout + 31) / 32;
    if (off || H != W || (H != 64 && H != 128 && H != 256) || d.c % 16 || g.taps != 1 || g.nTiles != tiles || g.act != 0) return false;
    if (d.stride == 2) {                                                    // blocks 2 (256 -> 128) and 4 (128 -> 64)
        static const bool s2 = IVF_EXP_ENV("IVF_FCN_NOSTRIDE2") == nullptr;
        if (!s2 || d.dil != 1 || tiles != 1 || (H != 256 && H != 128)) return false;
        const int Ho = H / 2, wg = Ho * Ho / 128;
        if (Ho == 128)
            hipLaunchKernelGGL((k_fcn_dwpw<1, 1, 7, 2>), dim3(wg * B, 1), dim3(256), 0, s, X, d.dPack, g.dWq, g.dScale, g.dShift, res, Y, d.c, g.cout, g.nTiles, abl);
        else
            hipLaunchKernelGGL((k_fcn_dwpw<1, 1, 6, 2>), dim3(wg * B, 1), dim3(256), 0, s, X, d.dPack, g.dWq, g.dScale, g.dShift, res, Y, d.c, g.cout, g.nTiles, abl);
        return true;
    }
    if (d.stride != 1) return false;
    static const bool wide = IVF_EXP_ENV("IVF_FCN_NOWIDE") == nullptr;           // 128-wide maps (block 3: 297 -> 114 us)
    static const bool wide256 = IVF_EXP_ENV("IVF_FCN_WIDE256") != nullptr;       // block 1 has only two K chunks: the fused kernel is
                                                                            // slower there (335 vs 270 us), off unless asked for
    if (H != 64 && (!wide || d.dil != 1 || tiles != 1)) return false;
    const dim3 blk(256);
    const int wgpi = H * W / 128;
    // eight waves (four image rows) per workgroup on the 64 x 64 maps.  Measured per 128 images, 4 vs 8 waves: 1 tile 125 -> 117,
    // 2 tiles 263 -> 251 / 144 -> 128, 3 tiles 388 -> 395, 5 tiles 915 -> 919 us: on for <= 2 tiles; IVF_FCN_NW = 4 / 8 forces one
    static const int nwEnv = IVF_EXP_ENV("IVF_FCN_NW") ? atoi(IVF_EXP_ENV("IVF_FCN_NW")) : 0;
#define DWPW(T, D, LWV, GY) do {                                                                                          \
    snprintf(g_lastDwpw, sizeof g_lastDwpw, "ivffcn::k_fcn_dwpw<" #T ", " #D "> %d->%d", d.c, g.cout);                      \
    if (LWV == 6 && (nwEnv == 8 || (nwEnv != 4 && T <= 2)))                                                                 \
        hipLaunchKernelGGL((k_fcn_dwpw<T, D, LWV, 1, (LWV == 6 ? 8 : 4)>), dim3(wgpi / 2 * B, GY), dim3(512), 0, s, X, d.dPack, g.dWq, g.dScale,    \
                           g.dShift, res, Y, d.c, g.cout, g.nTiles, abl);                                                   \
    else hipLaunchKernelGGL((k_fcn_dwpw<T, D, LWV, 1>), dim3(wgpi * B, GY), blk, 0, s, X, d.dPack, g.dWq, g.dScale, g.dShift, res, Y, d.c, \
                       g.cout, g.nTiles, abl); } while (0)
#define DWPW8(D, T) do { snprintf(g_lastDwpw, sizeof g_lastDwpw, "ivffcn::k_fcn_dwpw8<" #D ", " #T "> %d->%d", d.c, g.cout);             \
    hipLaunchKernelGGL((k_fcn_dwpw8<D, T>), dim3(wgpi * B), dim3(512), 0, s, X, d.dPack, g.dWq, g.dScale, g.dShift, res, Y, d.c, \
                       g.cout, g.nTiles); } while (0)
    // 8-wave kernel per shape (measured per 64 images, 4-wave vs 8-wave): 576->160 dil 2: 294 vs 258 (on by default);
    // 960->160 dil 4: 460 vs 472, 576->96: 213 vs 228, 384->64: 148 vs 167 (off).  IVF_FCN_DWPW8 = bit mask by tile count.
    static const int w8 = IVF_EXP_ENV("IVF_FCN_DWPW8") ? (int)strtol(IVF_EXP_ENV("IVF_FCN_DWPW8"), nullptr, 0) : -1;
    if (H == 64 && !abl) {
        auto on = [&](bool dflt) { return w8 < 0 ? dflt : (w8 >> tiles & 1) != 0; };
        if (tiles == 5 && d.dil == 2 && on(true)) { DWPW8(2, 5); return true; }
        if (tiles == 5 && d.dil == 4 && on(false)) { DWPW8(4, 5); return true; }
        if (tiles == 3 && d.dil == 2 && on(false)) { DWPW8(2, 3); return true; }
        if (tiles == 2 && d.dil == 2 && on(false)) { DWPW8(2, 2); return true; }
    }
    if (H == 128) DWPW(1, 1, 7, 1);
    else if (H == 256) { if (!wide256) return false; DWPW(1, 1, 8, 1); }
    else if (tiles == 1 && d.dil == 1) DWPW(1, 1, 6, 1);
    else if (tiles == 2 && d.dil == 1) DWPW(2, 1, 6, 1);
    else if (tiles == 2 && d.dil == 2) DWPW(2, 2, 6, 1);
    else if (tiles == 3 && d.dil == 2) DWPW(3, 2, 6, 1);
    else if (tiles == 5 && d.dil == 2) DWPW(5, 2, 6, 1);
    else if (tiles == 5 && d.dil == 4) DWPW(5, 4, 6, 1);
    else if (tiles == 10 && d.dil == 4) {
        static const bool one = IVF_EXP_ENV("IVF_FCN_NODWPW10") == nullptr;      // one pass over the hidden tensor with an 8-wave workgroup
        if (one && !abl) DWPW8(4, 10); else DWPW(5, 4, 6, 2);
    }
    else return false;
#undef DWPW
#undef DWPW8
    return true;
}

}  // namespace ivffcn
using namespace ivffcn;

struct ivf_fcn {
    int device = 0, inW = 0, inH = 0, outW = 0, outH = 0, maxBatch = 0;
    int numCU = 0;              // compute units of the device (persistent grids of the whole-block kernels)
    float *dConv0W = nullptr, *dConv0S = nullptr, *dConv0B = nullptr;
    float* dProj0W = nullptr;   // block 1's 16 x 32 projection in f32 (k_fcn_stem)
    uint4* dStemFrag = nullptr; // conv0's and that projection's A operands as f16 hi / lo MFMA fragments (k_fcn_stem)
    float *dIrbWE[3] = {}, *dIrbWP[3] = {};   // blocks 2-4: kIrbRep copies of the expansion's / projection's A fragments (k_fcn_irb)
    float* dIrbTab[3] = {};     // blocks 2-4: the LDS parameter table of k_fcn_irb ([NG * 32][13] per hidden channel + projection BN scale[32] | shift[32])
    std::vector<float> hConv0W; // conv0's pre-scaled rows (host copy, until dStemFrag is built)
    std::vector<Gemm> pw;        // in forward order: per block expand (t>1), project; then decoder cbr
    std::vector<Dw> dw;
    float* dLastW = nullptr; float lastBias = 0.f;
    struct Fused4 { uint4 *dWE = nullptr, *dWP = nullptr; float* dPar = nullptr; int cout = 0, tilesP = 0;
                    uint4* dWE6 = nullptr; } f4[3];   // blocks 15-17 (k_fcn_irbd4); dWE6: the expansion's operands in the FP6 form (r06)
    Fused4 f2[7];                                                                                                          // blocks 8-14 (k_fcn_irbd2)
    Fused4 f1[3];                                                                                                          // blocks 5-7 (k_fcn_irbd2, DIL = 1)
    float *bufIn = nullptr, *bufA = nullptr, *bufB = nullptr, *bufH1 = nullptr, *bufH2 = nullptr, *bufLogits = nullptr;
    float* bufPart = nullptr;             // partial projection sums of the small-batch (SPLIT) launches
    int* dStatus = nullptr;               // [1] device-side flags of this handle: bit 0 = an un-clamped activation left the f16 range (see g_fcnRange)
    uint8_t *dStageIn = nullptr, *dStageU8 = nullptr; float* dStageF = nullptr;
    void* hPin = nullptr;        // pinned host staging of the per-call path (ivf_fcn_forward)
    std::vector<void*> allocs;
    // measurement probe: HIP events around the first 960 -> 160 fused depthwise+projection launch (block 15) of each
    // forward, the single most expensive kernel of the network (bench.py's roofline line)
    static constexpr int kProbe = 64;
    hipEvent_t probe0[kProbe] = {}, probe1[kProbe] = {};
    int probeBatch[kProbe] = {};
    long probeCount = 0;
    char probeName[96] = "";       // the kernel the probe brackets, as dispatched
    double probeAlgoBytes = 0;     // its algorithmic HBM bytes per image (whole-block kernels: block input + output; the 960 -> 160 kernel: hidden tensor + residual + output)
    // r05: a second probe around block 17 (the largest single launch); ivf_fcn_probe_select picks which one probe_info / probe_stats report
    hipEvent_t probeB0[kProbe] = {}, probeB1[kProbe] = {};
    long probeBCount = 0;
    char probeBName[96] = "";
    double probeBAlgoBytes = 0;
    int probeSel = 0;
};

namespace {

int upload(ivf_fcn* f, const std::vector<float>& h, float** d)
{
    FHIP(hipMalloc(d, std::max<size_t>(h.size(), 1) * sizeof(float)));
    f->allocs.push_back(*d);
    FHIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return IVF_OK;
}

// Per-output-channel power-of-two pre-scaling of a convolution's weights, folded into the BatchNorm scale that follows it
// (both exact: only exponents change).  The split-f16 products keep 22 significant bits of a weight only while its `lo`
// half is a NORMAL f16, i.e. while |w| >~ 2^-3: the small pointwise weights of a trained checkpoint (~1e-3) would keep
// ~14 bits (lo in f16 subnormals).  Row r is scaled so that its largest |w| lands in [1, 2); the accumulator then comes out
// 2^e times larger and scale[r] * 2^-e undoes it in the epilogue.  Returns the scaled copy of w; sc is updated in place.
std::vector<float> prescale_rows(const float* w, int cout, int rowLen, std::vector<float>& sc)
{
    std::vector<float> ws((size_t)cout * rowLen);
    for (int co = 0; co < cout; co++) {
        float m = 0.f;
        for (int k = 0; k < rowLen; k++) m = std::max(m, std::fabs(w[(size_t)co * rowLen + k]));
        int e = 0;
        if (m > 0.f && std::isfinite(m)) { int ex; (void)std::frexp(m, &ex); e = 1 - ex; }      // m = f * 2^ex, f in [0.5, 1)
        e = std::min(std::max(e, -100), 100);
        for (int k = 0; k < rowLen; k++) ws[(size_t)co * rowLen + k] = std::ldexp(w[(size_t)co * rowLen + k], e);
        sc[co] = std::ldexp(sc[co], -e);
    }
    return ws;
}

// 6-bit operand codes of v_mfma_scale_f32_*_f8f6f4 (OCP MX element formats): e3m2 ("bf6": 1 + 3 + 2 bits, bias 3, largest 28) and e2m3 ("fp6":
// 1 + 2 + 3 bits, bias 1, largest 7.5), no inf / NaN.  enc6: nearest code, ties to the even code, saturating -- what v_cvt_scalef32_pk32_*6_f16
// does on the device (tools/probe/mfma_fp8_mix.hip checks host-packed operands against device-converted ones).
float dec6(int code, bool bf6)
{
    const int sgn = code & 32; code &= 31;
    float v;
    if (bf6) { const int e = code >> 2, m = code & 3; v = e ? (1.f + m * 0.25f) * std::ldexp(1.f, e - 3) : m * 0.0625f; }
    else     { const int e = code >> 3, m = code & 7; v = e ? (1.f + m * 0.125f) * std::ldexp(1.f, e - 1) : m * 0.125f; }
    return sgn ? -v : v;
}
int enc6(float x, bool bf6)
{
    const float a = std::fabs(x);
    int best = 0; float bd = 1e30f;
    for (int c = 0; c < 32; c++) {
        const float d = std::fabs(dec6(c, bf6) - a);
        if (d < bd || (d == bd && !(c & 1))) { bd = d; best = c; }
    }
    return best | (x < 0.f ? 32 : 0);
}
void pack6(uint32_t* dst6, const int* codes32)         // element i at bits [6 i, 6 i + 6) of 6 dwords
{
    for (int i = 0; i < 6; i++) dst6[i] = 0;
    for (int i = 0; i < 32; i++) {
        const unsigned bit = 6 * i, w = bit >> 5, sft = bit & 31;
        dst6[w] |= (uint32_t)codes32[i] << sft;
        if (sft > 26) dst6[w + 1] |= (uint32_t)codes32[i] >> (32 - sft);
    }
}

// w / sc: already pre-scaled (prescale_rows)
int make_gemm(ivf_fcn* f, const float* w, int cout, int cin, int taps, const std::vector<float>& sc,
              const std::vector<float>& sh, int act, Gemm& g)
{
    g.cin = cin; g.cout = cout; g.taps = taps; g.act = act;
    const int tiles = (cout + 31) / 32;
    // wave tile = 32*NT output channels x 32*PT pixels.  Few input channels (short K loop, output-write bound):
    // small accumulator tiles so several waves share a SIMD and hide the load/store latency; long K loops: bigger
    // tiles for operand reuse.  IVF_FCN_NT_SMALL / IVF_FCN_NT_BIG / IVF_FCN_USE25 override for tuning runs.
    static const int ntSmall = IVF_EXP_ENV("IVF_FCN_NT_SMALL") ? atoi(IVF_EXP_ENV("IVF_FCN_NT_SMALL")) : 1;
    static const int ntBig = IVF_EXP_ENV("IVF_FCN_NT_BIG") ? atoi(IVF_EXP_ENV("IVF_FCN_NT_BIG")) : 1;
    static const int use25 = IVF_EXP_ENV("IVF_FCN_USE25") ? atoi(IVF_EXP_ENV("IVF_FCN_USE25")) : 0;
    static const int ptSmall = IVF_EXP_ENV("IVF_FCN_PT_SMALL") ? atoi(IVF_EXP_ENV("IVF_FCN_PT_SMALL")) : 4;
    static const int ptBig = IVF_EXP_ENV("IVF_FCN_PT_BIG") ? atoi(IVF_EXP_ENV("IVF_FCN_PT_BIG")) : 4;
    if (taps == 9) { g.NT = 3; g.PT = 1; }
    else if (cin <= 32) { g.NT = std::min(tiles, ntSmall); g.PT = ptSmall; }
    else if (tiles == 5 && use25) { g.NT = 5; g.PT = 2; }
    else { g.NT = std::min(tiles, ntBig); g.PT = ptBig; }
    g.nTiles = (tiles + g.NT - 1) / g.NT * g.NT;
    const int K16 = (cin + 15) / 16;
    // one extra all-zero K step when K16 is odd: k_fcn_irb / k_fcn_irb64 consume the projection in 32-channel chunks
    std::vector<float> wq((size_t)taps * (K16 + (taps == 1 ? (K16 & 1) : 0)) * g.nTiles * 2 * 64 * 4, 0.f);     // 8 f16 = 4 dwords per lane and part
    uint16_t* q = reinterpret_cast<uint16_t*>(wq.data());
    for (int tap = 0; tap < taps; tap++)
        for (int s = 0; s < K16; s++)
            for (int t = 0; t < g.nTiles; t++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int co = t * 32 + (lane & 31), k = 16 * s + 8 * (lane >> 5) + j;
                        if (co >= cout || k >= cin) continue;
                        const float v = w[((size_t)co * cin + k) * taps + tap];
                        const uint16_t hi = f32_to_f16(v);
                        const uint16_t lo = f32_to_f16(v - f16_to_f32(hi));
                        const size_t frag = (((size_t)tap * K16 + s) * g.nTiles + t) * 2;
                        q[((frag + 0) * 64 + lane) * 8 + j] = hi;
                        q[((frag + 1) * 64 + lane) * 8 + j] = lo;
                    }
    float* dq = nullptr;
    int rc = upload(f, wq, &dq); if (rc) return rc;
    g.dWq = reinterpret_cast<uint4*>(dq);
    if (taps == 9 && cin % 32 == 0 && g.nTiles == 3) {
        // k_fcn_conv3x3_f6: unit u = 3 s2 + dy (K steps 2 s2, 2 s2 + 1 of 16 channels; tap row dy), group g = 3 n + dx: four 1 KB pieces --
        // hi fragments of the two steps (lane: row = lane & 31, k = 8 (lane >> 5) + j), then the bf6 correction operand of the pair: element e = 8 seg + j,
        // step seg >> 1; seg even -> w_lo 2^(SH + 10) (meets x_hi), seg odd -> w_hi 2^SH (meets x_lo 2^10); dwords 0-3, then dwords 4-5 in .x .y
        const int units = 3 * (cin / 32);
        std::vector<float> w6((size_t)units * 9 * 4 * 64 * 4, 0.f);
        uint32_t* d32 = reinterpret_cast<uint32_t*>(w6.data());
        uint16_t* d16 = reinterpret_cast<uint16_t*>(w6.data());
        for (int u = 0; u < units; u++)
            for (int gq = 0; gq < 9; gq++) {
                const int s2 = u / 3, dy = u % 3, n = gq / 3, dx = gq % 3, tap = dy * 3 + dx;
                const size_t base = ((size_t)u * 9 + gq) * 4 * 64 * 4;          // dwords
                for (int lane = 0; lane < 64; lane++) {
                    const int co = n * 32 + (lane & 31), kgh = lane >> 5;
                    int codes[32];
                    for (int e = 0; e < 32; e++) {
                        const int seg = e >> 3, j = e & 7, k = 16 * (2 * s2 + (seg >> 1)) + 8 * kgh + j;
                        float v = 0.f;
                        if (co < cout) {
                            const float wv = w[((size_t)co * cin + k) * taps + tap];
                            const float hi = f16_to_f32(f32_to_f16(wv)), lo = f16_to_f32(f32_to_f16(wv - hi));
                            v = (seg & 1) ? std::ldexp(hi, kDecSH) : std::ldexp(lo, kDecSH + 10);
                            if (seg == 0) d16[(base + (size_t)lane * 4) * 2 + j] = f32_to_f16(wv);
                            if (seg == 2) d16[(base + (size_t)(64 + lane) * 4) * 2 + j] = f32_to_f16(wv);
                        }
                        codes[e] = enc6(v, true);
                    }
                    uint32_t six[6]; pack6(six, codes);
                    for (int q = 0; q < 4; q++) d32[base + (size_t)(128 + lane) * 4 + q] = six[q];
                    d32[base + (size_t)(192 + lane) * 4 + 0] = six[4]; d32[base + (size_t)(192 + lane) * 4 + 1] = six[5];
                }
            }
        float* d6 = nullptr;
        rc = upload(f, w6, &d6); if (rc) return rc;
        g.dWq6 = reinterpret_cast<uint4*>(d6);
    }
    std::vector<float> scp((size_t)g.nTiles * 32, 0.f), shp((size_t)g.nTiles * 32, 0.f);   // padded: float4 loads per tile
    std::copy(sc.begin(), sc.begin() + cout, scp.begin());
    std::copy(sh.begin(), sh.begin() + cout, shp.begin());
    rc = upload(f, scp, &g.dScale); if (rc) return rc;
    return upload(f, shp, &g.dShift);
}

// Operands of k_fcn_irbd4 (blocks 15-17): expansion rows as A fragments of v_mfma_f32_16x16x32_f16 (lane: row = lane & 15,
// k = 8 (lane >> 4) + j) per group of 16 hidden channels and K step of 32 input channels; projection rows as A fragments of
// v_mfma_f32_32x32x16_f16 (lane: row = lane & 31, k = 8 (lane >> 5) + j) per group (= one K step) and output tile; per hidden
// channel 12 parameters: the nine depthwise taps times the depthwise BN scale, its shift, the expansion's BN scale and shift.
// we / wp: the pre-scaled rows (prescale_rows), scE: the expansion's BN scale after the pre-scaling.
int make_fused4(ivf_fcn* f, ivf_fcn::Fused4& F, const float* we, const std::vector<float>& scE, const std::vector<float>& shE, const float* wd,
                const std::vector<float>& scD, const std::vector<float>& shD, const float* wp, int cout, int cin = kF4Cin, int hid = kF4Hid)
{
    const int groups = hid / 16, ksteps = cin / 32;
    auto put = [](uint16_t* q, size_t frag, int lane, int j, float v) {
        const uint16_t hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
        q[((frag + 0) * 64 + lane) * 8 + j] = hi; q[((frag + 1) * 64 + lane) * 8 + j] = lo;
    };
    F.cout = cout; F.tilesP = cout / 32;
    std::vector<float> qe((size_t)groups * ksteps * 2 * 64 * 4, 0.f), qp((size_t)groups * F.tilesP * 2 * 64 * 4, 0.f), par((size_t)hid * 12, 0.f);
    uint16_t* e16 = reinterpret_cast<uint16_t*>(qe.data()); uint16_t* p16 = reinterpret_cast<uint16_t*>(qp.data());
    for (int g = 0; g < groups; g++) {
        for (int s5 = 0; s5 < ksteps; s5++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++)
                    put(e16, ((size_t)g * ksteps + s5) * 2, lane, j, we[(size_t)(16 * g + (lane & 15)) * cin + 32 * s5 + 8 * (lane >> 4) + j]);
        for (int t = 0; t < F.tilesP; t++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++)
                    put(p16, ((size_t)g * F.tilesP + t) * 2, lane, j, wp[(size_t)(32 * t + (lane & 31)) * hid + 16 * g + 8 * (lane >> 5) + j]);
    }
    for (int c = 0; c < hid; c++) {
        for (int q = 0; q < 9; q++) par[(size_t)c * 12 + q] = wd[(size_t)c * 9 + q] * scD[c];
        par[(size_t)c * 12 + 9] = shD[c]; par[(size_t)c * 12 + 10] = scE[c]; par[(size_t)c * 12 + 11] = shE[c];
    }
    float *de = nullptr, *dp = nullptr;
    int rc;
    if ((rc = upload(f, qe, &de)) || (rc = upload(f, qp, &dp)) || (rc = upload(f, par, &F.dPar))) return rc;
    F.dWE = reinterpret_cast<uint4*>(de); F.dWP = reinterpret_cast<uint4*>(dp);
    if (IVF_F4_FP6_BUILT && cin == kF4Cin && hid == kF4Hid) {
        // FP6 form of the expansion (k_fcn_irbd4<.., FP6>): per group ten 1 KB pieces -- 0-4: the hi fragments of the five K steps; 5-7: dwords 0-3 of
        // the bf6 correction operand of instruction c (lane (row m = lane & 15, kq = lane >> 4), element e = 8 seg + j: channel 32 (2c + (seg >> 1)) +
        // 8 kq + j; seg even -> w_lo 2^(SH + 10) (meets x_hi), seg odd -> w_hi 2^SH (meets x_lo 2^10); K step 5 does not exist: zero); 8-9: dwords 4-5
        // as [c][lane] uint2 (the last half piece is padding)
        std::vector<float> q6((size_t)groups * 10 * 64 * 4, 0.f);
        uint32_t* d32 = reinterpret_cast<uint32_t*>(q6.data());
        uint16_t* d16 = reinterpret_cast<uint16_t*>(q6.data());
        for (int g = 0; g < groups; g++) {
            const size_t base = (size_t)g * 10 * 64 * 4;                 // dwords
            for (int s5 = 0; s5 < 5; s5++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++)
                        d16[(base + ((size_t)s5 * 64 + lane) * 4) * 2 + j] = f32_to_f16(we[(size_t)(16 * g + (lane & 15)) * cin + 32 * s5 + 8 * (lane >> 4) + j]);
            for (int c = 0; c < 3; c++)
                for (int lane = 0; lane < 64; lane++) {
                    int codes[32];
                    for (int e = 0; e < 32; e++) {
                        const int seg = e >> 3, j = e & 7, s5 = 2 * c + (seg >> 1);
                        float v = 0.f;
                        if (s5 < 5) {
                            const float w = we[(size_t)(16 * g + (lane & 15)) * cin + 32 * s5 + 8 * (lane >> 4) + j];
                            const float hi = f16_to_f32(f32_to_f16(w)), lo = f16_to_f32(f32_to_f16(w - hi));
                            v = (seg & 1) ? std::ldexp(hi, kF6SH) : std::ldexp(lo, kF6SH + 10);
                        }
                        codes[e] = enc6(v, true);
                    }
                    uint32_t six[6]; pack6(six, codes);
                    for (int q = 0; q < 4; q++) d32[base + ((size_t)(5 + c) * 64 + lane) * 4 + q] = six[q];
                    d32[base + (size_t)8 * 64 * 4 + ((size_t)c * 64 + lane) * 2 + 0] = six[4];
                    d32[base + (size_t)8 * 64 * 4 + ((size_t)c * 64 + lane) * 2 + 1] = six[5];
                }
        }
        float* d6 = nullptr;
        if ((rc = upload(f, q6, &d6))) return rc;
        F.dWE6 = reinterpret_cast<uint4*>(d6);
    }
    return IVF_OK;
}

// Dynamic LDS above 64 KB must be reserved per KERNEL (hipFuncAttributeMaxDynamicSharedMemorySize).  Every instantiation of the
// whole-block kernels is listed here once, at ivf_fcn_create: all k_fcn_irbd2<...> share one function-pointer type, so a flag inside a
// generic launch lambda would cover only the first instance launched.
int reserve_lds()
{
    struct { const void* fn; size_t lds; const char* name; } ks[] = {
        {reinterpret_cast<const void*>(&k_fcn_irbd2<32, 32, true, 1>), D2Cfg<32, 32, 1>::LDS, "k_fcn_irbd2<32,32,true,1>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd2<32, 64, false, 1>), D2Cfg<32, 64, 1>::LDS, "k_fcn_irbd2<32,64,false,1>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd2<64, 64, true>), D2Cfg<64, 64>::LDS, "k_fcn_irbd2<64,64,true>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd2<64, 96, false>), D2Cfg<64, 96>::LDS, "k_fcn_irbd2<64,96,false>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd2<96, 96, true>), D2Cfg<96, 96>::LDS, "k_fcn_irbd2<96,96,true>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd2<96, 160, false>), D2Cfg<96, 160>::LDS, "k_fcn_irbd2<96,160,false>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd4<true>), kF4Lds, "k_fcn_irbd4<true>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd4<false>), kF4Lds, "k_fcn_irbd4<false>"},
#if IVF_F4_FP6_BUILT
        {reinterpret_cast<const void*>(&k_fcn_irbd4<true, false, true>), kF4Lds, "k_fcn_irbd4<true,fp6>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd4<false, false, true>), kF4Lds, "k_fcn_irbd4<false,fp6>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd4<false, true, true>), kF4Lds, "k_fcn_irbd4<split,fp6>"},
#endif
        // the small-batch (SPLIT) instances
        {reinterpret_cast<const void*>(&k_fcn_irbd2<64, 64, true, 2, true>), D2Cfg<64, 64>::LDS, "k_fcn_irbd2<64,64,split>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd2<64, 96, false, 2, true>), D2Cfg<64, 96>::LDS, "k_fcn_irbd2<64,96,split>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd2<96, 96, true, 2, true>), D2Cfg<96, 96>::LDS, "k_fcn_irbd2<96,96,split>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd2<96, 160, false, 2, true>), D2Cfg<96, 160>::LDS, "k_fcn_irbd2<96,160,split>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd4<false, true>), kF4Lds, "k_fcn_irbd4<split>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd4h<false>), kH4Lds, "k_fcn_irbd4h"},
#if IVF_F4_FP6_BUILT
        {reinterpret_cast<const void*>(&k_fcn_irbd4h<true>), kH4Lds, "k_fcn_irbd4h<fp6>"},
#endif
#ifdef IVF_EXPERIMENT
        {reinterpret_cast<const void*>(&k_fcn_irbd4w<true>), kF4Lds, "k_fcn_irbd4w<true>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd4w<false>), kF4Lds, "k_fcn_irbd4w<false>"},
        {reinterpret_cast<const void*>(&k_fcn_irbd4w<false, true>), kF4Lds, "k_fcn_irbd4w<split>"},
#endif
    };
    for (auto& k : ks)
        if (hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.lds) != hipSuccess)
            return ffail(IVF_E_NO_DEVICE, "cannot reserve %zu bytes of LDS for %s", k.lds, k.name);
#ifdef IVF_EXPERIMENT
    if (const char* e = IVF_EXP_ENV("IVF_FCN_WABL")) { const int v = atoi(e); FHIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wAbl), &v, sizeof v)); }
#endif
    return IVF_OK;
}

// Small batches (the per-call drop-in path runs ONE image): a whole-block kernel launches 16 workgroups per image, a sixteenth of the
// chip at batch 1.  There the hidden groups of blocks 8-17 are cut into `ns` ranges over gridDim.z: ns ~ 16 / images, at least 4 groups
// per range, partial sums within f->bufPart.  ns depends on the batch size only, so a cost map is reproducible for a given batch size; it
// differs from the one computed in a large batch by the f32 summation order of the projection (1e-6 relative; tests bound it at 3e-4).
constexpr size_t kPartFloats = (size_t)16 * 320 * 4096;         // bufPart: ns x images x Cout x 4096 <= this
int split_ways(int n, int groups, int cout)
{
    static const int mode = IVF_EXP_ENV("IVF_FCN_SPLIT") ? atoi(IVF_EXP_ENV("IVF_FCN_SPLIT")) : 1;      // 0 = never (every batch through the batched form)
    if (!mode || n >= 16) return 1;
    int ns = std::min(16 / n, groups / 4);
    while (ns > 1 && (size_t)ns * n * cout * 4096 > kPartFloats) ns--;
    return std::max(ns, 1);
}
// Persistent form of a whole-block kernel (r05): one workgroup per CU that walks its XCD's tiles, when every workgroup gets at least two.
#ifndef IVF_PERSIST
#define IVF_PERSIST 1
#endif
int persistent_grid(const ivf_fcn* f, int tiles)
{
    static const int mode = IVF_EXP_ENV("IVF_FCN_PERSIST") ? atoi(IVF_EXP_ENV("IVF_FCN_PERSIST")) : IVF_PERSIST;      // 0 = one tile per workgroup
    const int g = f->numCU & ~7;                                                                                       // whole XCD slots
    return (mode && g >= 8 && tiles >= 2 * g && tiles % 8 == 0) ? g : tiles;
}
void launch_split_reduce(ivf_fcn* f, int ns, int n, int cout, const Gemm& pj, const float* res, float* y, int layIn, int layOut, hipStream_t s)
{
    const size_t total4 = (size_t)n * cout * 1024;
    hipLaunchKernelGGL(k_fcn_split_reduce, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, (const float*)f->bufPart, ns, (size_t)n * cout * 4096, cout,
                       (const float*)pj.dScale, (const float*)pj.dShift, res, y, total4, layIn, layOut);
}

// IVF_FCN_DEBUG=1: synchronise and check after every launch, naming the stage that failed
#define STAGE(name)                                                                                          \
    do { if (dbg) { hipError_t e_ = hipStreamSynchronize(s); if (e_ == hipSuccess) e_ = hipGetLastError();   \
                    fprintf(stderr, "[ivf_fcn] %s: %s\n", name, hipGetErrorString(e_));                      \
                    if (e_ != hipSuccess) return ffail(IVF_E_NO_DEVICE, "stage %s: %s", name, hipGetErrorString(e_)); } } while (0)

int forward_device(ivf_fcn* f, const uint8_t* dBgr, size_t imageStride, int rowStride, int n, uint8_t* dU8, float* dF,
                   hipStream_t s, size_t uImageStride = 0, int uRowStride = 0)
{
    if (!uRowStride) { uRowStride = f->outW; uImageStride = (size_t)f->outW * f->outH; }
    static const bool dbg = getenv("IVF_FCN_DEBUG") != nullptr;
    char nm[64];
    // r05: the 512^2 / 256^2 / 128^2 stage (prep, stem, blocks 2-4) runs in CHUNKS of images, back to back per chunk: a chunk's tensors
    // (12.6 MB f32 input planes, 4.2 MB stem output, 1.6 MB block outputs per image) are written and read back while they are still in the
    // 256 MB Infinity Cache instead of after the whole batch (403 MB + 537 MB + ... at 128 images) has gone through HBM
    static const int headChunkEnv = IVF_EXP_ENV("IVF_FCN_HEADCHUNK") ? atoi(IVF_EXP_ENV("IVF_FCN_HEADCHUNK")) : IVF_FCN_HEADCHUNK_DEFAULT;
    int headChunk = headChunkEnv;
    if (headChunk <= 0 || headChunk >= n) headChunk = 0;
    if (!headChunk) {
        hipLaunchKernelGGL(k_fcn_prep, dim3(kEnc / 256, kEnc / kPrepRows, n), dim3(256), 0, s, dBgr, imageStride, rowStride, f->inW, f->inH, f->bufIn);
        STAGE("prep");
    }
    // whole-block kernels, bit i = block i + 2: blocks 2-4 (k_fcn_irb) by default; bits 3-9 = blocks 5-11 through k_fcn_irb64, which
    // is correct but measures slower than expand + dwpw there (337 vs 245 us for the 64->384->64 blocks): opt-in.  Off under the
    // layer-by-layer / other-kernel experiment switches
    static const unsigned irbMask = (IVF_EXP_ENV("IVF_FCN_NOFUSE") || IVF_EXP_ENV("IVF_FCN_NOSTRIDE2")) ? 0u
                                    : IVF_EXP_ENV("IVF_FCN_IRBMASK") ? (unsigned)strtoul(IVF_EXP_ENV("IVF_FCN_IRBMASK"), nullptr, 0) : 7u;
    // conv0 + block 1's depthwise layer in one kernel, unless an experiment switch asks for another kernel on block 1
    static const bool stem = IVF_EXP_ENV("IVF_FCN_NOSTEM") == nullptr && IVF_EXP_ENV("IVF_FCN_WIDE256") == nullptr && IVF_EXP_ENV("IVF_FCN_NOFUSE") == nullptr;
    // r05: the three tensors between the stem and block 4 row-interleaved ([y][channel][x]) when all four kernels are the whole-block ones
    static const int headIlEnv = IVF_EXP_ENV("IVF_FCN_HEAD_IL") ? atoi(IVF_EXP_ENV("IVF_FCN_HEAD_IL")) : IVF_FCN_HEAD_IL_DEFAULT;
    const int headIl = (headIlEnv && stem && (irbMask & 7u) == 7u) ? 1 : 0;
    const bool chunkedHead = headChunk > 0 && stem && (irbMask & 7u) == 7u;
    if (headChunk > 0 && !chunkedHead) {            // an experiment switch took a head kernel away: the plain schedule
        hipLaunchKernelGGL(k_fcn_prep, dim3(kEnc / 256, kEnc / kPrepRows, n), dim3(256), 0, s, dBgr, imageStride, rowStride, f->inW, f->inH, f->bufIn);
        STAGE("prep");
    }
    if (chunkedHead) {
        const Dw& d0 = f->dw[0];
        for (int c0 = 0; c0 < n; c0 += headChunk) {
            const int nb = std::min(headChunk, n - c0);
            float* in = f->bufIn + (size_t)c0 * 3 * kEnc * kEnc;
            float* o1 = f->bufB + (size_t)c0 * 16 * 256 * 256;          // stem (conv0 + block 1)
            float* o2 = f->bufA + (size_t)c0 * 24 * 128 * 128;          // block 2
            float* o3 = f->bufB + (size_t)c0 * 24 * 128 * 128;          // block 3 (its chunk's stem output is dead by then; other chunks' live regions lie elsewhere)
            float* o4 = f->bufA + (size_t)c0 * 32 * 64 * 64;            // block 4: the layout the 64 x 64 stage expects
            hipLaunchKernelGGL(k_fcn_prep, dim3(kEnc / 256, kEnc / kPrepRows, nb), dim3(256), 0, s, dBgr + (size_t)c0 * imageStride, imageStride, rowStride, f->inW, f->inH, in);
            hipLaunchKernelGGL(k_fcn_stem, dim3(kEnc / 2 / kStemTW, kEnc / 2 / kStemTH, nb), dim3(512), 0, s, (const float*)in, f->dConv0W, f->dConv0S,
                               f->dConv0B, d0.dW, d0.dScale, d0.dShift, f->dProj0W, f->pw[0].dScale, f->pw[0].dShift, o1, f->dStemFrag, headIl);
#define IRBC(S_, CIN_, HID_, COUT_, RES_, WI_, TH_, IP_, ID_, X_, Y_, T_)                                                                          \
            hipLaunchKernelGGL((k_fcn_irb<S_, CIN_, HID_, COUT_, RES_, WI_, TH_>), dim3(WI_ / S_ / 32, WI_ / S_ / TH_, nb), dim3(512), 0, s, (const float*)(X_), \
                               (const uint4*)f->dIrbWE[T_], f->pw[IP_].dScale, f->pw[IP_].dShift, f->dw[ID_].dW, f->dw[ID_].dScale, f->dw[ID_].dShift, (const uint4*)f->dIrbWP[T_], \
                               f->pw[IP_ + 1].dScale, f->pw[IP_ + 1].dShift, (Y_), (const float4*)f->dIrbTab[T_], headIl, (T_ < 2 ? headIl : 0))
            IRBC(2, 16, 96, 24, false, 256, IVF_IRB_TH2, 1, 1, o1, o2, 0);
            IRBC(1, 24, 144, 24, true, 128, IVF_IRB_TH3, 3, 2, o2, o3, 1);
            IRBC(2, 24, 144, 32, false, 128, IVF_IRB_TH4, 5, 3, o3, o4, 2);
#undef IRBC
        }
        STAGE("prep + stem + blocks 2-4, chunked");
    } else if (stem) {
        const Dw& d0 = f->dw[0];
        hipLaunchKernelGGL(k_fcn_stem, dim3(kEnc / 2 / kStemTW, kEnc / 2 / kStemTH, n), dim3(512), 0, s, f->bufIn, f->dConv0W, f->dConv0S,
                           f->dConv0B, d0.dW, d0.dScale, d0.dShift, f->dProj0W, f->pw[0].dScale, f->pw[0].dShift, f->bufB, f->dStemFrag, headIl);
        STAGE("stem (conv0 + block 1)");
    } else {
        hipLaunchKernelGGL(k_fcn_conv0, dim3(1, kEnc / 2, n), dim3(256), 0, s, f->bufIn, f->dConv0W, f->dConv0S, f->dConv0B, f->bufA);
        STAGE("conv0");
    }
    // which blocks run as ONE whole-block kernel of the 64 x 64 stage (k_fcn_irbd2 / k_fcn_irbd4), and the layout of every tensor
    // between them: lay[i] = layout of block i's OUTPUT (0 planes, 1 NHWC8: producer AND consumer are such kernels, or the decoder)
    static const int fused1 = IVF_EXP_ENV("IVF_FCN_NOFUSE") ? 0 : IVF_EXP_ENV("IVF_FCN_FUSED1") ? atoi(IVF_EXP_ENV("IVF_FCN_FUSED1")) : 1;
    static const int fused2 = IVF_EXP_ENV("IVF_FCN_NOFUSE") ? 0 : IVF_EXP_ENV("IVF_FCN_FUSED2") ? atoi(IVF_EXP_ENV("IVF_FCN_FUSED2")) : 1;
    static const int fused4 = IVF_EXP_ENV("IVF_FCN_NOFUSE") ? 0 : IVF_EXP_ENV("IVF_FCN_FUSED4") ? atoi(IVF_EXP_ENV("IVF_FCN_FUSED4")) : 1;
    static const bool fuseLast = IVF_EXP_ENV("IVF_FCN_NOFUSELAST") == nullptr && IVF_EXP_ENV("IVF_FCN_OLD3X3") == nullptr &&
                                 IVF_EXP_ENV("IVF_FCN_3X3_SPLIT") == nullptr;
    static const int tileMask = IVF_EXP_ENV("IVF_FCN_TILED") ? atoi(IVF_EXP_ENV("IVF_FCN_TILED")) : 6;      // bit 0: lay 1 (measured slower: off), bit 1: lay 2, bit 2: lay 4
    auto whole = [&](int i) {
        if (i >= 4 && i <= 10 && (irbMask >> (i - 1) & 1)) return false;           // k_fcn_irb64 (opt-in) takes the block
        if (i >= 4 && i <= 6) return fused1 && f->f1[i - 4].dWE != nullptr;
        if (i >= 7 && i <= 13) return fused2 && f->f2[i - 7].dWE != nullptr;
        if (i >= 14 && i <= 16) return fused4 && f->f4[i - 14].dWE != nullptr;
        return false;
    };
    // lay[i] = layout of block i's OUTPUT = the tile layout of block i + 1 when both are whole-block kernels, planes otherwise
    // (block 17 writes planes for the decoder, block 5 reads the planes of block 4)
    int lay[18] = {};
    for (int i = 4; i <= 15; i++)
        if (whole(i) && whole(i + 1)) {
            const int want = i + 1 <= 6 ? 1 : i + 1 <= 13 ? 2 : 4;
            if (tileMask & want) lay[i] = want;
        }
    float *x = f->bufA, *y = f->bufB;
    int H = kEnc / 2, W = kEnc / 2;
    size_t ip = 0, id = 0;
    int iFirst = 0;
    if (chunkedHead) { iFirst = 4; ip = 7; id = 4; H = W = 64; }      // blocks 1-4 are done: block 4's output is in bufA = x
    for (int i = iFirst; i < 17; i++) {
        const Block& bk = kBlocks[i];
        const int hid = bk.inp * bk.t;
        const float* h = x;
        if (i == 0 && stem) {                       // block 1 (t = 1, stride 1, no residual) ran inside the stem kernel: its output is in y
            id++; ip++;
            std::swap(x, y);
            continue;
        }
        if (i >= 4 && i <= 10 && (irbMask >> (i - 1) & 1)) {     // blocks 5-11 (64 x 64, <= 64 input channels): one kernel per block
            const Gemm& ex = f->pw[ip]; const Gemm& pj = f->pw[ip + 1]; const Dw& d = f->dw[id];
#define IRB64(DIL_, CIN_, HID_, COUT_, RES_)                                                                                \
            hipLaunchKernelGGL((k_fcn_irb64<DIL_, CIN_, HID_, COUT_, RES_>), dim3(2, 32, n), dim3(512), 0, s, x, ex.dWq, ex.dScale,      \
                               ex.dShift, d.dW, d.dScale, d.dShift, pj.dWq, pj.dScale, pj.dShift, y)
            if (i == 4 || i == 5) IRB64(1, 32, 192, 32, true);
            else if (i == 6) IRB64(1, 32, 192, 64, false);
            else if (i <= 9) IRB64(2, 64, 384, 64, true);
            else IRB64(2, 64, 384, 96, false);
#undef IRB64
            ip += 2; id++;
            snprintf(nm, sizeof nm, "block %d whole", i + 1); STAGE(nm);
            std::swap(x, y);
            continue;
        }
        if (i >= 1 && i <= 3 && (irbMask >> (i - 1) & 1)) {      // blocks 2-4: one kernel per block, hidden tensor in LDS
            const Gemm& ex = f->pw[ip]; const Gemm& pj = f->pw[ip + 1]; const Dw& d = f->dw[id];
#define IRB(S_, CIN_, HID_, COUT_, RES_, WI_, TH_)                                                                          \
            hipLaunchKernelGGL((k_fcn_irb<S_, CIN_, HID_, COUT_, RES_, WI_, TH_>), dim3(WI_ / S_ / 32, WI_ / S_ / TH_, n), dim3(512), 0, s, x, \
                               (const uint4*)f->dIrbWE[i - 1], ex.dScale, ex.dShift, d.dW, d.dScale, d.dShift, (const uint4*)f->dIrbWP[i - 1], pj.dScale, pj.dShift, y, \
                               (const float4*)f->dIrbTab[i - 1], headIl, (i < 3 ? headIl : 0))
            if (i == 1) IRB(2, 16, 96, 24, false, 256, IVF_IRB_TH2);
            else if (i == 2) IRB(1, 24, 144, 24, true, 128, IVF_IRB_TH3);
            else IRB(2, 24, 144, 32, false, 128, IVF_IRB_TH4);
#undef IRB
            ip += 2; id++;
            H = (H - 1) / d.stride + 1; W = (W - 1) / d.stride + 1;
            snprintf(nm, sizeof nm, "block %d whole", i + 1); STAGE(nm);
            std::swap(x, y);
            continue;
        }
        const int layIn = i >= 1 ? lay[i - 1] : 0, layOut = lay[i];
        if (fused1 && i >= 4 && i <= 6 && f->f1[i - 4].dWE && H == 64 && W == 64) {           // blocks 5-7: the same kernel on 4-row strips of the whole map
            const ivf_fcn::Fused4& F = f->f1[i - 4];
            const Gemm& pj = f->pw[ip + 1];
            bool ok = true;
            auto go = [&](auto kern, size_t lds) {                                               // LDS reserved per instantiation by reserve_lds()
                hipLaunchKernelGGL(kern, dim3(persistent_grid(f, 16 * n)), dim3(512), lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale, pj.dShift, bk.res ? x : (const float*)nullptr, y,
                                   (float*)nullptr, layIn, layOut, 16 * n);
            };
            if (bk.oup == 32 && bk.res) go(&k_fcn_irbd2<32, 32, true, 1>, D2Cfg<32, 32, 1>::LDS);
            else if (bk.oup == 64 && !bk.res) go(&k_fcn_irbd2<32, 64, false, 1>, D2Cfg<32, 64, 1>::LDS);
            else ok = false;
            if (!ok) return ffail(IVF_E_NO_DEVICE, "block %d: no k_fcn_irbd2<DIL 1> instance / LDS reservation failed", i + 1);
            ip += 2; id++;
            snprintf(nm, sizeof nm, "block %d whole (4-row strips)", i + 1); STAGE(nm);
            std::swap(x, y);
            continue;
        }
        if (fused2 && i >= 7 && i <= 13 && f->f2[i - 7].dWE && H == 64 && W == 64) {           // blocks 8-14: one kernel, no hidden tensor in HBM
            const ivf_fcn::Fused4& F = f->f2[i - 7];
            const Gemm& pj = f->pw[ip + 1];
            bool ok = true;
            // small batches: the hidden groups are cut into `ns` ranges over gridDim.z, partial sums through f->bufPart (see k_fcn_irbd4)
            const int ns = split_ways(n, hid / 16, bk.oup);
            auto go = [&](auto kern, auto kernSplit, size_t lds, bool walk = true) {             // LDS reserved per instantiation by reserve_lds()
                if (ns > 1) {
                    hipLaunchKernelGGL(kernSplit, dim3(16 * n, 1, ns), dim3(512), lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale, pj.dShift, (const float*)nullptr, y,
                                       f->bufPart, layIn, layOut, 16 * n);
                    launch_split_reduce(f, ns, n, bk.oup, pj, bk.res ? x : nullptr, y, layIn, layOut, s);
                } else
                    hipLaunchKernelGGL(kern, dim3(walk ? persistent_grid(f, 16 * n) : 16 * n), dim3(512), lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale, pj.dShift,
                                       bk.res ? x : (const float*)nullptr, y, (float*)nullptr, layIn, layOut, 16 * n);
            };
            if (bk.inp == 64 && bk.oup == 64 && bk.res) go(&k_fcn_irbd2<64, 64, true>, &k_fcn_irbd2<64, 64, true, 2, true>, D2Cfg<64, 64>::LDS);
            else if (bk.inp == 64 && bk.oup == 96 && !bk.res) go(&k_fcn_irbd2<64, 96, false>, &k_fcn_irbd2<64, 96, false, 2, true>, D2Cfg<64, 96>::LDS);
            else if (bk.inp == 96 && bk.oup == 96 && bk.res) go(&k_fcn_irbd2<96, 96, true>, &k_fcn_irbd2<96, 96, true, 2, true>, D2Cfg<96, 96>::LDS);
            else if (bk.inp == 96 && bk.oup == 160 && !bk.res) go(&k_fcn_irbd2<96, 160, false>, &k_fcn_irbd2<96, 160, false, 2, true>, D2Cfg<96, 160>::LDS, D2Cfg<96, 160>::WALK);
            else ok = false;
            if (!ok) return ffail(IVF_E_NO_DEVICE, "block %d: no k_fcn_irbd2 instance / LDS reservation failed", i + 1);
            ip += 2; id++;
            snprintf(nm, sizeof nm, "block %d whole (dilation-2 strips)", i + 1); STAGE(nm);
            std::swap(x, y);
            continue;
        }
        if (fused4 && i >= 14 && f->f4[i - 14].dWE && H == 64 && W == 64) {      // blocks 15-17: one kernel, no hidden tensor in HBM
            const ivf_fcn::Fused4& F = f->f4[i - 14];
            const Gemm& pj = f->pw[ip + 1];
            const bool probe4 = i == 14 && f->probe0[0];
            const int slot4 = (int)(f->probeCount % ivf_fcn::kProbe);
            if (probe4) FHIP(hipEventRecord(f->probe0[slot4], s));
            const bool probeB = i == 16 && f->probeB0[0];
            const int slotB = (int)(f->probeBCount % ivf_fcn::kProbe);
            if (probeB) FHIP(hipEventRecord(f->probeB0[slotB], s));
            const char* kname = "k_fcn_irbd4";
            const int ns = split_ways(n, kF4Groups, F.cout);
            const dim3 grid(16 * n, F.cout / 160, ns);
            static const int fp6 = IVF_EXP_ENV("IVF_FCN_FP6") ? atoi(IVF_EXP_ENV("IVF_FCN_FP6")) : IVF_F4_FP6;     // 1: the expansion's correction products on bf6 x fp6 (r06; experiment build)
            static const int half4 = IVF_EXP_ENV("IVF_FCN_HALF4") ? atoi(IVF_EXP_ENV("IVF_FCN_HALF4")) : 1;     // 0: block 17 as two workgroups per 256-pixel tile (r03 / r04)
#ifdef IVF_EXPERIMENT
            static const int roles = IVF_EXP_ENV("IVF_FCN_ROLES") ? atoi(IVF_EXP_ENV("IVF_FCN_ROLES")) : 0;      // 1: role-specialised waves (k_fcn_irbd4w)
            if (roles) {
                if (ns > 1) {
                    hipLaunchKernelGGL((k_fcn_irbd4w<false, true>), grid, dim3(1024), kF4Lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale, pj.dShift, (const float*)nullptr, y,
                                       F.cout, F.tilesP, f->bufPart, layIn, layOut);
                    launch_split_reduce(f, ns, n, F.cout, pj, bk.res ? x : nullptr, y, layIn, layOut, s);
                } else if (bk.res)
                    hipLaunchKernelGGL((k_fcn_irbd4w<true>), grid, dim3(1024), kF4Lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale, pj.dShift, x, y, F.cout, F.tilesP, (float*)nullptr,
                                       layIn, layOut);
                else
                    hipLaunchKernelGGL((k_fcn_irbd4w<false>), grid, dim3(1024), kF4Lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale, pj.dShift, (const float*)nullptr, y, F.cout, F.tilesP,
                                       (float*)nullptr, layIn, layOut);
            } else
#endif
            if (ns > 1) {
#if IVF_F4_FP6_BUILT
                if (fp6 && F.dWE6)      // the same arithmetic as the batched form: a batch of 1 and a batch of 128 differ by summation order only
                    hipLaunchKernelGGL((k_fcn_irbd4<false, true, true>), grid, dim3(512), kF4Lds, s, x, F.dWE6, F.dPar, F.dWP, pj.dScale, pj.dShift, (const float*)nullptr, y,
                                       F.cout, F.tilesP, f->bufPart, layIn, layOut, 16 * n);
                else
#endif
                hipLaunchKernelGGL((k_fcn_irbd4<false, true>), grid, dim3(512), kF4Lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale, pj.dShift, (const float*)nullptr, y,
                                   F.cout, F.tilesP, f->bufPart, layIn, layOut, 16 * n);
                launch_split_reduce(f, ns, n, F.cout, pj, bk.res ? x : nullptr, y, layIn, layOut, s);
            } else if (half4 && !bk.res && F.cout == kH4Cout && F.tilesP == kH4TilesP)
            {   // block 17 in ONE pass: half a sub-image (128 pixels) x all 320 outputs per workgroup
#if IVF_F4_FP6_BUILT
                if (fp6 && F.dWE6)
                    hipLaunchKernelGGL(k_fcn_irbd4h<true>, dim3(32 * n), dim3(512), kH4Lds, s, x, F.dWE6, F.dPar, F.dWP, pj.dScale, pj.dShift, y, layIn, layOut, 32 * n);
                else
#endif
                hipLaunchKernelGGL(k_fcn_irbd4h<false>, dim3(IVF_H4_WALK ? persistent_grid(f, 32 * n) : 32 * n), dim3(512), kH4Lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale, pj.dShift, y,
                                   layIn, layOut, 32 * n);
                kname = "k_fcn_irbd4h";
            }
#if IVF_F4_FP6_BUILT
            else if (fp6 && F.dWE6 && bk.res)
                hipLaunchKernelGGL((k_fcn_irbd4<true, false, true>), dim3(16 * n, grid.y, 1), dim3(512), kF4Lds, s, x, F.dWE6, F.dPar, F.dWP, pj.dScale,
                                   pj.dShift, x, y, F.cout, F.tilesP, (float*)nullptr, layIn, layOut, 16 * n);
            else if (fp6 && F.dWE6)
                hipLaunchKernelGGL((k_fcn_irbd4<false, false, true>), grid, dim3(512), kF4Lds, s, x, F.dWE6, F.dPar, F.dWP, pj.dScale, pj.dShift, (const float*)nullptr, y, F.cout, F.tilesP,
                                   (float*)nullptr, layIn, layOut, 16 * n);
#endif
            else if (bk.res)
                hipLaunchKernelGGL((k_fcn_irbd4<true>), dim3(IVF_F4_WALK && grid.y == 1 ? persistent_grid(f, 16 * n) : 16 * n, grid.y, 1), dim3(512), kF4Lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale,
                                   pj.dShift, x, y, F.cout, F.tilesP, (float*)nullptr, layIn, layOut, 16 * n);
            else
                hipLaunchKernelGGL((k_fcn_irbd4<false>), grid, dim3(512), kF4Lds, s, x, F.dWE, F.dPar, F.dWP, pj.dScale, pj.dShift, (const float*)nullptr, y, F.cout, F.tilesP,
                                   (float*)nullptr, layIn, layOut, 16 * n);
            if (probe4) {
                FHIP(hipEventRecord(f->probe1[slot4], s)); f->probeBatch[slot4] = n; f->probeCount++;
                snprintf(f->probeName, sizeof f->probeName, "ivffcn::k_fcn_irbd4<%s> %d->%d->%d", bk.res ? "true" : "false", bk.inp, hid, bk.oup);
                f->probeAlgoBytes = (double)(bk.inp + bk.oup) * H * W * sizeof(float);       // the residual IS the input (read once: r05)
            }
            if (probeB) {
                FHIP(hipEventRecord(f->probeB1[slotB], s)); f->probeBCount++;
                if (strcmp(kname, "k_fcn_irbd4h") == 0) snprintf(f->probeBName, sizeof f->probeBName, "ivffcn::k_fcn_irbd4h %d->%d->%d", bk.inp, hid, bk.oup);
                else snprintf(f->probeBName, sizeof f->probeBName, "ivffcn::k_fcn_irbd4<%s> %d->%d->%d", bk.res ? "true" : "false", bk.inp, hid, bk.oup);
                f->probeBAlgoBytes = (double)(bk.inp + bk.oup) * H * W * sizeof(float);
            }
            ip += 2; id++;
            snprintf(nm, sizeof nm, "block %d whole (dilation-4 phases)", i + 1); STAGE(nm);
            std::swap(x, y);
            continue;
        }
        // IVF_FCN_CHUNK = images per chunk (0 = off): expansion and fused depthwise + projection of a 64 x 64 block run chunk by chunk,
        // back to back, through ONE hidden-tensor region of `chunk` images that every chunk reuses -- the region (8 images x 15.7 MB
        // for the 960-channel blocks) stays in the 256 MB Infinity Cache between its write and its read
        static const int chunkEnv = IVF_EXP_ENV("IVF_FCN_CHUNK") ? atoi(IVF_EXP_ENV("IVF_FCN_CHUNK")) : 0;
        if (chunkEnv > 0 && bk.t != 1 && H == 64 && W == 64 && n > chunkEnv && bk.stride == 1) {
            const Dw& d = f->dw[id];
            bool ok = true;
            for (int c0 = 0; c0 < n && ok; c0 += chunkEnv) {
                const int nb = std::min(chunkEnv, n - c0);
                const float* xo = x + (size_t)c0 * bk.inp * H * W;
                float* yo = y + (size_t)c0 * bk.oup * H * W;
                if (!launch_expand(f->pw[ip], xo, f->bufH1, H, W, nb, s)) launch_gemm(f->pw[ip], xo, nullptr, f->bufH1, H, W, nb, s);
                ok = launch_dwpw(d, f->pw[ip + 1], f->bufH1, bk.res ? xo : nullptr, yo, H, W, nb, s);
            }
            if (!ok) return ffail(IVF_E_STATE, "chunked schedule: block %d has no fused depthwise + projection kernel", i + 1);
            ip += 2; id++;
            snprintf(nm, sizeof nm, "block %d chunked", i + 1); STAGE(nm);
            std::swap(x, y);
            continue;
        }
        if (bk.t != 1) {
            if (!launch_expand(f->pw[ip], x, f->bufH1, H, W, n, s)) launch_gemm(f->pw[ip], x, nullptr, f->bufH1, H, W, n, s);
            ip++; h = f->bufH1; snprintf(nm, sizeof nm, "block %d expand", i + 1); STAGE(nm); }
        const Dw& d = f->dw[id++];
        const bool probe = i == 14 && f->probe0[0];
        const int slot = (int)(f->probeCount % ivf_fcn::kProbe);
        if (probe) FHIP(hipEventRecord(f->probe0[slot], s));
        if (launch_dwpw(d, f->pw[ip], h, bk.res ? x : nullptr, y, H, W, n, s)) {
            if (probe) {
                FHIP(hipEventRecord(f->probe1[slot], s)); f->probeBatch[slot] = n; f->probeCount++;
                snprintf(f->probeName, sizeof f->probeName, "%s", g_lastDwpw);
                f->probeAlgoBytes = (double)(hid + f->pw[ip].cout * (bk.res ? 2 : 1)) * H * W * sizeof(float);
            }
            ip++;
            H = (H - 1) / d.stride + 1; W = (W - 1) / d.stride + 1;
            snprintf(nm, sizeof nm, "block %d depthwise+project", i + 1); STAGE(nm);
            std::swap(x, y);
            continue;
        }
        const int Ho = (H + 2 * d.dil - 2 * d.dil - 1) / d.stride + 1, Wo = (W + 2 * d.dil - 2 * d.dil - 1) / d.stride + 1;
        {
            const int TH = (d.stride == 1 && Ho <= 64) ? 64 : 16;
            const dim3 grid((Wo + kDwTW - 1) / kDwTW, (Ho + TH - 1) / TH, n * hid);
            const int IW = (kDwTW - 1) * d.stride + 2 * d.dil + 1, IH = (TH - 1) * d.stride + 2 * d.dil + 1;
            const size_t lds = (size_t)IH * (IW | 1) * sizeof(float);
            if (d.stride == 1 && TH == 64)
                hipLaunchKernelGGL((k_fcn_dw<1, 64>), grid, dim3(256), lds, s, h, d.dW, d.dScale, d.dShift, f->bufH2, hid, H, W, Ho, Wo, d.dil);
            else if (d.stride == 1)
                hipLaunchKernelGGL((k_fcn_dw<1, 16>), grid, dim3(256), lds, s, h, d.dW, d.dScale, d.dShift, f->bufH2, hid, H, W, Ho, Wo, d.dil);
            else
                hipLaunchKernelGGL((k_fcn_dw<2, 16>), grid, dim3(256), lds, s, h, d.dW, d.dScale, d.dShift, f->bufH2, hid, H, W, Ho, Wo, d.dil);
        }
        snprintf(nm, sizeof nm, "block %d depthwise", i + 1); STAGE(nm);
        H = Ho; W = Wo;
        launch_gemm(f->pw[ip++], f->bufH2, bk.res ? x : nullptr, y, H, W, n, s);
        snprintf(nm, sizeof nm, "block %d project", i + 1); STAGE(nm);
        std::swap(x, y);
    }
    {
        // decoder cbr (3x3 320->80 + BN + ReLU) and conv_last: one kernel when the all-tiles 3x3 kernel applies
        const Gemm& g = f->pw[ip++];
        if (fuseLast && g.taps == 9 && H == 64 && W == 64 && g.cin % 32 == 0 && g.act == 2 && g.nTiles == 3) {
            // small batches: the 60 (tap row, K step) pairs in `nd` ranges over gridDim.y + k_fcn_dec_reduce (same rule as split_ways)
            static const int splitMode = IVF_EXP_ENV("IVF_FCN_SPLIT") ? atoi(IVF_EXP_ENV("IVF_FCN_SPLIT")) : 1;
            int nd = 1;
            if (splitMode && g.cin == 320) { const int ways[] = {15, 10, 6, 5, 3, 2}; for (int w : ways) if (w * n <= 16) { nd = w; break; } }
            static const int dec6 = IVF_EXP_ENV("IVF_FCN_DEC6") ? atoi(IVF_EXP_ENV("IVF_FCN_DEC6")) : IVF_DEC_FP6;      // 0: three f16 products (k_fcn_conv3x3_all, r02-r05)
            const bool use6 = dec6 && g.dWq6 != nullptr;
            if (nd > 1) {
                if (use6 && dec6 >= 2 && (30 / nd) % 3 == 0)      // ranges of whole K-step pairs
                    hipLaunchKernelGGL(k_fcn_conv3x3_f6r, dim3(8 * n, nd), dim3(256), 0, s, x, g.dWq6, g.dScale, g.dShift, f->bufH1, g.cin, g.cout,
                                       (const float*)nullptr, 0.f, (float*)nullptr, f->bufPart);
                else if (use6)     // the same arithmetic as the batched form: the 30 units in `nd` ranges
                    hipLaunchKernelGGL(k_fcn_conv3x3_f6, dim3(8 * n, nd), dim3(256), 0, s, x, g.dWq6, g.dScale, g.dShift, f->bufH1, g.cin, g.cout,
                                       (const float*)nullptr, 0.f, (float*)nullptr, f->bufPart);
                else
                hipLaunchKernelGGL((k_fcn_conv3x3_all<3>), dim3(8 * n, nd), dim3(256), 0, s, x, g.dWq, g.dScale, g.dShift, f->bufH1, g.cin, g.cout,
                                   (const float*)nullptr, 0.f, (float*)nullptr, f->bufPart);
                hipLaunchKernelGGL(k_fcn_dec_reduce, dim3(64, n), dim3(256), 0, s, (const float*)f->bufPart, nd, (size_t)n * g.cout * 4096, g.cout,
                                   (const float*)g.dScale, (const float*)g.dShift, (const float*)f->dLastW, f->lastBias, f->bufLogits);
            } else if (use6 && dec6 >= 2)
                hipLaunchKernelGGL(k_fcn_conv3x3_f6r, dim3(8 * n), dim3(256), 0, s, x, g.dWq6, g.dScale, g.dShift, f->bufH1, g.cin, g.cout,
                                   (const float*)f->dLastW, f->lastBias, f->bufLogits, (float*)nullptr);
            else if (use6)
                hipLaunchKernelGGL(k_fcn_conv3x3_f6, dim3(8 * n), dim3(256), 0, s, x, g.dWq6, g.dScale, g.dShift, f->bufH1, g.cin, g.cout,
                                   (const float*)f->dLastW, f->lastBias, f->bufLogits, (float*)nullptr);
            else
                hipLaunchKernelGGL((k_fcn_conv3x3_all<3>), dim3(8 * n), dim3(256), 0, s, x, g.dWq, g.dScale, g.dShift, f->bufH1, g.cin, g.cout,
                                   (const float*)f->dLastW, f->lastBias, f->bufLogits, (float*)nullptr);
            STAGE("decoder cbr + conv_last");
        } else {
            launch_gemm(g, x, nullptr, f->bufH1, H, W, n, s);
            STAGE("decoder cbr");
            hipLaunchKernelGGL(k_fcn_last, dim3((H * W + 255) / 256, n), dim3(256), 0, s, f->bufH1, f->dLastW, f->lastBias,
                               f->bufLogits, 80, H * W);
            STAGE("conv_last");
        }
    }
    const int outRows = n >= 4 ? kOutRows : 2;
    hipLaunchKernelGGL(k_fcn_out, dim3((f->outW + 1023) / 1024, (f->outH + outRows - 1) / outRows, n), dim3(256), 0, s, f->bufLogits, H, W, f->outH, f->outW,
                       (float)H / (float)f->outH, (float)W / (float)f->outW, dF, dU8, f->dStatus, outRows, uImageStride, uRowStride);
    FHIP(hipGetLastError());
    return IVF_OK;
}

}  // namespace

extern "C" {

int ivf_fcn_create(const float* weights_blob, size_t n_floats, int in_width, int in_height, int out_width, int out_height,
                   int max_batch, int device_id, ivf_fcn** out)
{
    if (!weights_blob || !out) return ffail(IVF_E_INVALID, "null argument");
    *out = nullptr;
    if (in_width < 2 || in_height < 2 || out_width < 1 || out_height < 1 || max_batch < 1)
        return ffail(IVF_E_INVALID, "bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ffail(IVF_E_NO_DEVICE, "no HIP device available; libivfront has no CPU path");
    if (device_id < 0 || device_id >= ndev) return ffail(IVF_E_INVALID, "device_id %d outside [0,%d)", device_id, ndev);
    FHIP(hipSetDevice(device_id));
    { const int lrc = reserve_lds(); if (lrc) return lrc; }
    for (size_t i = 0; i < n_floats; i++)      // before the handle exists: nothing to release on this way out
        if (!std::isfinite(weights_blob[i])) return ffail(IVF_E_INVALID, "weight blob holds a non-finite value at float %zu", i);
    ivf_fcn* f = new ivf_fcn();
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess) f->numCU = cus; }
    f->device = device_id; f->inW = in_width; f->inH = in_height; f->outW = out_width; f->outH = out_height; f->maxBatch = max_batch;
    Reader rd{weights_blob, n_floats};
    auto bad = [&]() { ivf_fcn_destroy(f); return ffail(IVF_E_INVALID, "weight blob too short for the architecture"); };
    std::vector<float> sc, sh;
    auto read_bn = [&](int c) -> bool {
        const float *g = rd.take(c), *b = rd.take(c), *m = rd.take(c), *v = rd.take(c);
        if (!g || !b || !m || !v) return false;
        fold_bn(g, b, m, v, c, sc, sh);
        return true;
    };
    int rc;
    {   // features[0]
        const float* w = rd.take(32 * 27);
        if (!w || !read_bn(32)) return bad();
        std::vector<float> hw = prescale_rows(w, 32, 27, sc);
        if ((rc = upload(f, hw, &f->dConv0W)) || (rc = upload(f, sc, &f->dConv0S)) || (rc = upload(f, sh, &f->dConv0B))) { ivf_fcn_destroy(f); return rc; }
        f->hConv0W = hw;
    }
    for (int i = 0; i < 17; i++) {
        const Block& bk = kBlocks[i];
        const int hid = bk.inp * bk.t;
        const bool f4 = i >= 14 && bk.inp == kF4Cin && hid == kF4Hid && bk.dil == 4 && bk.stride == 1 && bk.oup % 160 == 0;   // blocks 15-17
        const bool f2 = i >= 7 && i <= 13 && bk.dil == 2 && bk.stride == 1 && bk.t == 6 && (bk.inp == 64 || bk.inp == 96);        // blocks 8-14
        const bool f1 = i >= 4 && i <= 6 && bk.dil == 1 && bk.stride == 1 && bk.t == 6 && bk.inp == 32;                          // blocks 5-7
        std::vector<float> f4we, f4scE, f4shE, f4scD, f4shD; const float* f4wd = nullptr;
        if (bk.t != 1) {
            const float* w = rd.take((size_t)hid * bk.inp);
            if (!w || !read_bn(hid)) return bad();
            const std::vector<float> ws = prescale_rows(w, hid, bk.inp, sc);
            Gemm g; if ((rc = make_gemm(f, ws.data(), hid, bk.inp, 1, sc, sh, 1, g))) { ivf_fcn_destroy(f); return rc; }
            f->pw.push_back(g);
            if (f4 || f2 || f1) { f4we = ws; f4scE = sc; f4shE = sh; }
        }
        {
            const float* w = rd.take((size_t)hid * 9);
            if (!w || !read_bn(hid)) return bad();
            if (f4 || f2 || f1) { f4wd = w; f4scD = sc; f4shD = sh; }
            Dw d; d.c = hid; d.stride = bk.stride; d.dil = bk.dil;
            std::vector<float> hw(w, w + (size_t)hid * 9);
            std::vector<float> pk((size_t)((hid + 31) / 32 * 32) * 12, 0.f);    // padded to whole 32-channel chunks
            for (int c = 0; c < hid; c++) {
                for (int q = 0; q < 9; q++) pk[(size_t)c * 12 + q] = w[(size_t)c * 9 + q];
                pk[(size_t)c * 12 + 9] = sc[c]; pk[(size_t)c * 12 + 10] = sh[c];
            }
            if ((rc = upload(f, hw, &d.dW)) || (rc = upload(f, sc, &d.dScale)) || (rc = upload(f, sh, &d.dShift)) ||
                (rc = upload(f, pk, &d.dPack))) { ivf_fcn_destroy(f); return rc; }
            f->dw.push_back(d);
        }
        {
            const float* w = rd.take((size_t)bk.oup * hid);
            if (!w || !read_bn(bk.oup)) return bad();
            const std::vector<float> ws = prescale_rows(w, bk.oup, hid, sc);
            Gemm g; if ((rc = make_gemm(f, ws.data(), bk.oup, hid, 1, sc, sh, 0, g))) { ivf_fcn_destroy(f); return rc; }
            if (i == 0 && (rc = upload(f, ws, &f->dProj0W))) { ivf_fcn_destroy(f); return rc; }      // same pre-scaled rows, f32 (k_fcn_stem)
            if (i == 0) {
                // k_fcn_stem's A fragments (32x32x16: lane = row col & 31, k = 16 st + 8 (lane >> 5) + j): conv0 [32][27 -> 32], projection [16 -> 32][32]
                std::vector<float> q((size_t)2 * 2 * 2 * 64 * 4, 0.f);
                uint16_t* q16 = reinterpret_cast<uint16_t*>(q.data());
                for (int conv = 0; conv < 2; conv++)
                    for (int st = 0; st < 2; st++)
                        for (int lane = 0; lane < 64; lane++)
                            for (int j = 0; j < 8; j++) {
                                const int col = lane & 31, k = 16 * st + 8 * (lane >> 5) + j;
                                const float v = conv == 0 ? (k < 27 ? f->hConv0W[(size_t)col * 27 + k] : 0.f) : (col < 16 ? ws[(size_t)col * 32 + k] : 0.f);
                                const uint16_t hi = f32_to_f16(v), lo = f32_to_f16(v - f16_to_f32(hi));
                                const size_t frag = (size_t)(conv * 2 + st) * 2;
                                q16[((frag + 0) * 64 + lane) * 8 + j] = hi; q16[((frag + 1) * 64 + lane) * 8 + j] = lo;
                            }
                float* dq = nullptr;
                if ((rc = upload(f, q, &dq))) { ivf_fcn_destroy(f); return rc; }
                f->dStemFrag = reinterpret_cast<uint4*>(dq);
            }
            f->pw.push_back(g);
            if (f4 && (rc = make_fused4(f, f->f4[i - 14], f4we.data(), f4scE, f4shE, f4wd, f4scD, f4shD, ws.data(), bk.oup))) { ivf_fcn_destroy(f); return rc; }
            if (f1 && (rc = make_fused4(f, f->f1[i - 4], f4we.data(), f4scE, f4shE, f4wd, f4scD, f4shD, ws.data(), bk.oup, bk.inp, hid))) { ivf_fcn_destroy(f); return rc; }
            if (f2 && (rc = make_fused4(f, f->f2[i - 7], f4we.data(), f4scE, f4shE, f4wd, f4scD, f4shD, ws.data(), bk.oup, bk.inp, hid))) { ivf_fcn_destroy(f); return rc; }
        }
    }
    {   // decoder: cbr (3x3 320->80 + BN + ReLU), cbr_deepsup (unused at inference), conv_last, conv_last_deepsup (unused)
        const float* w = rd.take((size_t)80 * 320 * 9);
        if (!w || !read_bn(80)) return bad();
        const std::vector<float> ws = prescale_rows(w, 80, 320 * 9, sc);
        Gemm g; if ((rc = make_gemm(f, ws.data(), 80, 320, 9, sc, sh, 2, g))) { ivf_fcn_destroy(f); return rc; }
        f->pw.push_back(g);
        if (!rd.take((size_t)80 * 160 * 9) || !rd.take(4 * 80)) return bad();
        const float* lw = rd.take(80); const float* lb = rd.take(1);
        if (!lw || !lb) return bad();
        std::vector<float> hw(lw, lw + 80);
        if ((rc = upload(f, hw, &f->dLastW))) { ivf_fcn_destroy(f); return rc; }
        f->lastBias = lb[0];
        if (!rd.take(80) || !rd.take(1)) return bad();
        if (rd.left != 0) { ivf_fcn_destroy(f); return ffail(IVF_E_INVALID, "weight blob has %zu trailing floats", rd.left); }
    }
    const size_t B = (size_t)max_batch;
    auto dalloc = [&](float** p, size_t n) -> int { FHIP(hipMalloc(p, n * sizeof(float))); f->allocs.push_back(*p); return IVF_OK; };
    const size_t big = (size_t)96 * 256 * 256;           // largest activation: 96 x 256 x 256
    if ((rc = dalloc(&f->bufIn, B * 3 * kEnc * kEnc)) || (rc = dalloc(&f->bufA, B * 32 * 256 * 256)) ||
        (rc = dalloc(&f->bufB, B * 32 * 256 * 256)) || (rc = dalloc(&f->bufH1, B * big)) || (rc = dalloc(&f->bufH2, B * big)) ||
        (rc = dalloc(&f->bufLogits, B * 64 * 64)) || (rc = dalloc(&f->bufPart, kPartFloats))) { ivf_fcn_destroy(f); return rc; }
    {   // k_fcn_irb's parameter tables (blocks 2-4): gathered from the per-layer arrays uploaded above
        size_t ipw = 1, idw = 1;                                // block 1 (t = 1) holds one pointwise and one depthwise layer
        for (int i = 1; i <= 3; i++) {
            const Block& bk = kBlocks[i];
            const int hid = bk.inp * bk.t, ng = (hid + 31) / 32, tp = 13;
            const Gemm& ex = f->pw[ipw]; const Gemm& pj = f->pw[ipw + 1]; const Dw& d = f->dw[idw];
            std::vector<float> se(hid), be(hid), wd((size_t)hid * 9), sdv(hid), bdv(hid), spv(bk.oup), bpv(bk.oup);
            if (hipMemcpy(se.data(), ex.dScale, hid * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(be.data(), ex.dShift, hid * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(wd.data(), d.dW, (size_t)hid * 36, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(sdv.data(), d.dScale, hid * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(bdv.data(), d.dShift, hid * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(spv.data(), pj.dScale, bk.oup * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(bpv.data(), pj.dShift, bk.oup * 4, hipMemcpyDeviceToHost) != hipSuccess) { ivf_fcn_destroy(f); return ffail(IVF_E_NO_DEVICE, "parameter table read-back failed"); }
            std::vector<float> tab((size_t)ng * 32 * tp + 64, 0.f);
            for (int c = 0; c < hid; c++) {
                float* t = tab.data() + (size_t)c * tp;
                t[0] = se[c]; t[1] = be[c]; t[11] = sdv[c]; t[12] = bdv[c];
                for (int k = 0; k < 9; k++) t[2 + k] = wd[(size_t)c * 9 + k];
            }
            for (int c = 0; c < bk.oup && c < 32; c++) { tab[(size_t)ng * 32 * tp + c] = spv[c]; tab[(size_t)ng * 32 * tp + 32 + c] = bpv[c]; }
            if ((rc = upload(f, tab, &f->dIrbTab[i - 1]))) { ivf_fcn_destroy(f); return rc; }
            {   // replicated A fragments: [kIrbRep][K16 * NG * 128] / [kIrbRep][2 * NG * 128] uint4
                const int k16 = (bk.inp + 15) / 16;
                const size_t ne = (size_t)k16 * ng * 128 * 4, np = (size_t)2 * ng * 128 * 4;       // floats
                std::vector<float> he(ne), hp(np), re(ne * kIrbRep), rp(np * kIrbRep);
                if (hipMemcpy(he.data(), ex.dWq, ne * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(hp.data(), pj.dWq, np * 4, hipMemcpyDeviceToHost) != hipSuccess) {
                    ivf_fcn_destroy(f); return ffail(IVF_E_NO_DEVICE, "fragment read-back failed");
                }
                for (int r = 0; r < kIrbRep; r++) { memcpy(re.data() + (size_t)r * ne, he.data(), ne * 4); memcpy(rp.data() + (size_t)r * np, hp.data(), np * 4); }
                if ((rc = upload(f, re, &f->dIrbWE[i - 1])) || (rc = upload(f, rp, &f->dIrbWP[i - 1]))) { ivf_fcn_destroy(f); return rc; }
            }
            ipw += 2; idw++;
        }
    }
    if (hipMalloc(&f->dStatus, 4 * sizeof(int)) != hipSuccess || hipMemset(f->dStatus, 0, 4 * sizeof(int)) != hipSuccess) {
        ivf_fcn_destroy(f);
        return ffail(IVF_E_NO_DEVICE, "status word allocation failed");
    }
    f->allocs.push_back(f->dStatus);
    *out = f;
    return IVF_OK;
}

void ivf_fcn_destroy(ivf_fcn* f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    (void)hipDeviceSynchronize();
#if defined(IVF_F4_TIMING) || defined(IVF_D2_TIMING)
    {
        unsigned long long t[16] = {};
        if (hipMemcpyFromSymbol(t, HIP_SYMBOL(ivffcn::g_f4Tim), sizeof t) == hipSuccess && t[7]) {
            for (int w = 0; w < 2; w++) {
                const double n = (double)t[8 * w + 7];
                fprintf(stderr, "[irbd4 timing] wave %d: workgroups %.0f; cycles per workgroup: dma issue %.0f  mfma phase %.0f (first reads %.0f, E loop %.0f, split + P loop %.0f, epilogue %.0f)  stencil phase %.0f  wait + barrier %.0f\n",
                        4 * w, n, t[8 * w] / n, (t[8 * w + 1] + t[8 * w + 4] + t[8 * w + 5] + t[8 * w + 6]) / n, t[8 * w + 4] / n, t[8 * w + 5] / n, t[8 * w + 6] / n, t[8 * w + 1] / n, t[8 * w + 2] / n, t[8 * w + 3] / n);
            }
            unsigned long long z[16] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(ivffcn::g_f4Tim), z, sizeof z);
            unsigned long long wv[8] = {};
            if (hipMemcpyFromSymbol(wv, HIP_SYMBOL(ivffcn::g_f4Whole), sizeof wv) == hipSuccess && wv[3])
                fprintf(stderr, "[irbd4 whole] workgroups %llu; s_memtime ticks per workgroup (wave 0): prologue %.0f  interval loop %.0f  epilogue %.0f "
                                "(residual loads issued -> landed %.0f, BN + stores issued %.0f, stores drained %.0f)\n",
                        wv[3], (double)wv[0] / wv[3], (double)wv[1] / wv[3], (double)wv[2] / wv[3], (double)wv[4] / wv[3], (double)wv[5] / wv[3], (double)wv[6] / wv[3]);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(ivffcn::g_f4Whole), z, sizeof wv);
        }
    }
#endif
#ifdef IVF_IRB_TIMING
    {
        unsigned long long t[10] = {};
        if (hipMemcpyFromSymbol(t, HIP_SYMBOL(ivffcn::g_irbTim), sizeof t) == hipSuccess && t[8]) {
            const double n = (double)t[8];
            fprintf(stderr, "[irb timing WI=%d S=%d] workgroups %.0f; cycles per workgroup (wave 0): loads -> LDS %.0f  window fragments %.0f  B1 expansion %.0f  "
                            "barrier-1 wait %.0f  B2 stencil %.0f  barrier-2 wait %.0f  B3 projection %.0f  epilogue %.0f\n", IVF_IRB_TIMING, IVF_IRB_TIMING_S, n,
                    t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n, t[5] / n, t[6] / n, t[7] / n);
            unsigned long long z[10] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(ivffcn::g_irbTim), z, sizeof z);
        }
    }
#endif
#ifdef IVF_DWPW_TIMING
    {
        unsigned long long t[8] = {};
        if (hipMemcpyFromSymbol(t, HIP_SYMBOL(ivffcn::g_dwpwTim), sizeof t) == hipSuccess && t[7]) {
            const double n = (double)t[7];
            fprintf(stderr, "[dwpw timing] workgroups %llu; cycles per workgroup: LDS read + split %.0f  MFMA %.0f  stencil %.0f  publish %.0f  issue %.0f  barrier %.0f\n",
                    t[7], t[5] / n, t[0] / n, t[1] / n, t[4] / n, t[2] / n, t[3] / n);
            unsigned long long z[8] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(ivffcn::g_dwpwTim), z, sizeof z);
        }
    }
#endif
    for (int i = 0; i < ivf_fcn::kProbe; i++) {
        if (f->probe0[i]) (void)hipEventDestroy(f->probe0[i]);
        if (f->probe1[i]) (void)hipEventDestroy(f->probe1[i]);
        if (f->probeB0[i]) (void)hipEventDestroy(f->probeB0[i]);
        if (f->probeB1[i]) (void)hipEventDestroy(f->probeB1[i]);
    }
    for (void* p : f->allocs) (void)hipFree(p);
    if (f->dStageIn) (void)hipFree(f->dStageIn);
    if (f->dStageU8) (void)hipFree(f->dStageU8);
    if (f->dStageF) (void)hipFree(f->dStageF);
    if (f->hPin) (void)hipHostFree(f->hPin);
    delete f;
}

int ivf_fcn_probe_enable(ivf_fcn* f)
{
    if (!f) return ffail(IVF_E_INVALID, "null argument");
    FHIP(hipSetDevice(f->device));
    for (int i = 0; i < ivf_fcn::kProbe; i++) {
        if (!f->probe0[i]) FHIP(hipEventCreate(&f->probe0[i]));
        if (!f->probe1[i]) FHIP(hipEventCreate(&f->probe1[i]));
        if (!f->probeB0[i]) FHIP(hipEventCreate(&f->probeB0[i]));
        if (!f->probeB1[i]) FHIP(hipEventCreate(&f->probeB1[i]));
    }
    f->probeCount = 0; f->probeBCount = 0;
    return IVF_OK;
}

int ivf_fcn_probe_select(ivf_fcn* f, int which)
{
    if (!f || which < 0 || which > 1) return ffail(IVF_E_INVALID, "probe 0 = block 15, probe 1 = block 17");
    f->probeSel = which;
    return IVF_OK;
}

int ivf_fcn_probe_info(const ivf_fcn* f, char* name, int name_cap, double* algorithmic_bytes_per_image)
{
    if (!f || !name || name_cap < 1) return ffail(IVF_E_INVALID, "bad argument");
    const char* pn = f->probeSel ? f->probeBName : f->probeName;
    if (!pn[0]) return ffail(IVF_E_STATE, "no probed forward has run");
    snprintf(name, (size_t)name_cap, "%s", pn);
    if (algorithmic_bytes_per_image) *algorithmic_bytes_per_image = f->probeSel ? f->probeBAlgoBytes : f->probeAlgoBytes;
    return IVF_OK;
}

int ivf_fcn_probe_stats(ivf_fcn* f, int last_n, double* sum_ms, int* n_out, int* batch)
{
    if (!f || !sum_ms || !n_out) return ffail(IVF_E_INVALID, "null argument");
    FHIP(hipSetDevice(f->device));
    const long count = f->probeSel ? f->probeBCount : f->probeCount;
    hipEvent_t* const e0 = f->probeSel ? f->probeB0 : f->probe0;
    hipEvent_t* const e1 = f->probeSel ? f->probeB1 : f->probe1;
    const long have = std::min<long>(count, ivf_fcn::kProbe);
    const long take = last_n > 0 ? std::min<long>(last_n, have) : have;
    double sum = 0; int n = 0;
    for (long k = 0; k < take; k++) {
        const int slot = (int)((count - 1 - k) % ivf_fcn::kProbe);
        FHIP(hipEventSynchronize(e1[slot]));
        float ms = 0.f;
        FHIP(hipEventElapsedTime(&ms, e0[slot], e1[slot]));
        sum += ms; n++;
        if (batch) *batch = f->probeBatch[slot];      // both probes bracket a launch of every forward: their slots advance together
    }
    *sum_ms = sum; *n_out = n;
    return IVF_OK;
}

int ivf_fcn_forward_device(ivf_fcn* f, const uint8_t* d_bgr, size_t image_stride, int row_stride, int n,
                           uint8_t* d_cost_u8, float* d_cost_f32, void* hip_stream)
{
    if (!f || !d_bgr || (!d_cost_u8 && !d_cost_f32)) return ffail(IVF_E_INVALID, "null argument");
    if (n < 1 || n > f->maxBatch) return ffail(IVF_E_INVALID, "batch %d outside [1,%d]", n, f->maxBatch);
    if (row_stride < 3 * f->inW) return ffail(IVF_E_INVALID, "row_stride too small");
    FHIP(hipSetDevice(f->device));
    // r06: a large batch runs as sub-batches of 64 images, back to back on the same stream.  The tensors between two launches of a 64-image forward (up to
    // 2 x 64 x 1.3 MB at the 64 x 64 stage, 4 x that at the head) then fit the 256 MB Infinity Cache; measured 75.0 / 75.5 vs 75.5 / 76.2 us per image
    // for 64 vs 128 images per launch sequence on one box (tools/time_fcn_batch.py; 32: 76.2-76.5, 16: 80.3), results identical (images are independent).
    static const int sub = IVF_EXP_ENV("IVF_FCN_SUBBATCH") ? std::max(1, atoi(IVF_EXP_ENV("IVF_FCN_SUBBATCH"))) : kSubBatch;
    const size_t outPx = (size_t)f->outW * f->outH;
    for (int i0 = 0; i0 < n; i0 += sub) {
        const int nb = std::min(sub, n - i0);
        const int rc = forward_device(f, d_bgr + (size_t)i0 * image_stride, image_stride, row_stride, nb, d_cost_u8 ? d_cost_u8 + (size_t)i0 * outPx : nullptr,
                                      d_cost_f32 ? d_cost_f32 + (size_t)i0 * outPx : nullptr, (hipStream_t)hip_stream);
        if (rc) return rc;
    }
    return IVF_OK;
}

int ivf_fcn_forward_device_strided(ivf_fcn* f, const uint8_t* d_bgr, size_t image_stride, int row_stride, int n,
                                   uint8_t* d_cost_u8, size_t cost_image_stride, int cost_row_stride, void* hip_stream)
{
    if (!f || !d_bgr || !d_cost_u8) return ffail(IVF_E_INVALID, "null argument");
    if (n < 1 || n > f->maxBatch) return ffail(IVF_E_INVALID, "batch %d outside [1,%d]", n, f->maxBatch);
    if (row_stride < 3 * f->inW) return ffail(IVF_E_INVALID, "row_stride too small");
    if (cost_row_stride < f->outW || cost_image_stride < (size_t)cost_row_stride * (f->outH - 1) + f->outW) return ffail(IVF_E_INVALID, "cost strides too small for %dx%d", f->outW, f->outH);
    FHIP(hipSetDevice(f->device));
    static const int sub = IVF_EXP_ENV("IVF_FCN_SUBBATCH") ? std::max(1, atoi(IVF_EXP_ENV("IVF_FCN_SUBBATCH"))) : kSubBatch;
    for (int i0 = 0; i0 < n; i0 += sub) {
        const int nb = std::min(sub, n - i0);
        const int rc = forward_device(f, d_bgr + (size_t)i0 * image_stride, image_stride, row_stride, nb, d_cost_u8 + (size_t)i0 * cost_image_stride, nullptr,
                                      (hipStream_t)hip_stream, cost_image_stride, cost_row_stride);
        if (rc) return rc;
    }
    return IVF_OK;
}

int ivf_fcn_status(ivf_fcn* f, void* hip_stream)
{
    if (!f) return ffail(IVF_E_INVALID, "null handle");
    FHIP(hipSetDevice(f->device));
    int s = 0;
    FHIP(hipMemcpyAsync(&s, f->dStatus, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    FHIP(hipStreamSynchronize((hipStream_t)hip_stream));
    if (!s) return IVF_OK;
    FHIP(hipMemsetAsync(f->dStatus, 0, sizeof(int), (hipStream_t)hip_stream));
    FHIP(hipStreamSynchronize((hipStream_t)hip_stream));
    return ffail(IVF_E_STATE, "ivf_fcn: an un-clamped activation left the f16 range (|x| >= 65504, flags 0x%x) in a forward since the last "
                              "check: cost maps computed since then are not to be trusted", s);
}

int ivf_fcn_forward(ivf_fcn* f, const uint8_t* bgr, int width, int height, int stride, uint8_t* cost_u8, int cost_stride,
                    float* cost_f32)
{
    if (!f || !bgr || (!cost_u8 && !cost_f32)) return ffail(IVF_E_INVALID, "null argument");
    if (width != f->inW || height != f->inH) return ffail(IVF_E_INVALID, "image is %dx%d, handle was created for %dx%d", width, height, f->inW, f->inH);
    if (stride < 3 * width || (cost_u8 && cost_stride < f->outW)) return ffail(IVF_E_INVALID, "stride too small");
    FHIP(hipSetDevice(f->device));
    const size_t inBytes = (size_t)width * 3 * height, outPx = (size_t)f->outW * f->outH;
    if (!f->dStageIn) {
        FHIP(hipMalloc(&f->dStageIn, inBytes)); FHIP(hipMalloc(&f->dStageU8, outPx)); FHIP(hipMalloc(&f->dStageF, outPx * sizeof(float)));
    }
    // The caller's buffers are pageable: hipMemcpy2D from / to them runs row by row (measured 6.5 ms for this 1.4 MB image, against
    // 0.84 ms for the whole forward).  Rows go through a pinned staging buffer of the handle instead: one host memcpy + one DMA.
    const size_t pinBytes = inBytes + outPx + outPx * sizeof(float) + 16;
    if (!f->hPin) FHIP(hipHostMalloc(&f->hPin, pinBytes, hipHostMallocDefault));
    uint8_t* hIn = (uint8_t*)f->hPin; uint8_t* hU8 = hIn + inBytes; float* hF = (float*)(hU8 + outPx);
    int* hSt = (int*)((uint8_t*)f->hPin + ((inBytes + outPx + outPx * sizeof(float) + 3) & ~(size_t)3));
    for (int y = 0; y < height; y++) memcpy(hIn + (size_t)y * width * 3, bgr + (size_t)y * stride, (size_t)width * 3);
    FHIP(hipMemcpyAsync(f->dStageIn, hIn, inBytes, hipMemcpyHostToDevice, nullptr));
    int rc = forward_device(f, f->dStageIn, inBytes, width * 3, 1, cost_u8 ? f->dStageU8 : nullptr, cost_f32 ? f->dStageF : nullptr, nullptr);
    if (rc) return rc;
    if (cost_u8) FHIP(hipMemcpyAsync(hU8, f->dStageU8, outPx, hipMemcpyDeviceToHost, nullptr));
    if (cost_f32) FHIP(hipMemcpyAsync(hF, f->dStageF, outPx * sizeof(float), hipMemcpyDeviceToHost, nullptr));
    FHIP(hipMemcpyAsync(hSt, f->dStatus, sizeof(int), hipMemcpyDeviceToHost, nullptr));
    FHIP(hipStreamSynchronize(nullptr));
    if (*hSt) {                                 // no plausible-looking cost map leaves the call (the output buffers are not written)
        FHIP(hipMemset(f->dStatus, 0, sizeof(int)));
        return ffail(IVF_E_STATE, "ivf_fcn_forward: an un-clamped activation left the f16 range (|x| >= 65504, flags 0x%x): the split-f16 "
                                  "products of the next layer would be wrong; these weights need activations re-scaled", *hSt);
    }
    if (cost_u8) for (int y = 0; y < f->outH; y++) memcpy(cost_u8 + (size_t)y * cost_stride, hU8 + (size_t)y * f->outW, (size_t)f->outW);
    if (cost_f32) memcpy(cost_f32, hF, outPx * sizeof(float));
    return IVF_OK;
}

}  // extern "C"
