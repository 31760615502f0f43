// ivf_rectify.hip -- stereo rectification in front of the extractor.
//
// The reference's driver builds undistort/rectify maps once (cv::initUndistortRectifyMap, CV_32F maps,
// introspective_ORB_SLAM/Examples/Stereo/stereo_kitti.cc:285-343) and then, per frame, runs
// cv::remap(..., INTER_LINEAR) on the left image, the right image and the predicted cost image
// (:462-468, :519-521).  Here the map is converted ONCE into the fixed-point form OpenCV's remap derives
// on every call (5-bit sub-pixel positions: int16 x, int16 y, 10-bit weight index) and stays in HBM; the
// per-frame work is one gather kernel.  Arithmetic: DESIGN.md A-9 / A-10 (OpenCV 4.x plain C++ paths;
// OpenCV is un-vendored, parity unpinned like the other OpenCV primitives).
//
// HBM traffic per destination pixel (1 channel): 4 B positions + 2 B weight index + 1 B written, and the
// source pixels through L2 (each source line is touched by neighbouring destination pixels).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "ivf_device.h"

#define rfail ivf::set_error
#define RHIPCHK(expr)                                                                                  \
    do { hipError_t e_ = (expr);                                                                        \
         if (e_ != hipSuccess) return rfail(IVF_E_NO_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

struct ivf_remap {
    int device = 0;
    int w = 0, h = 0, wp = 0;           // destination size = map size; wp = map row pitch in pixels (multiple of 4)
    int sw = 0, sh = 0, cn = 1;         // source size, channels
    uint32_t* dXY = nullptr;            // [h][wp] (int16 x) | (int16 y) << 16 : integer source position
    uint16_t* dA = nullptr;             // [h][wp] (fy5 << 5) | fx5 : weight index
    uint8_t *dSrc = nullptr, *dDst = nullptr;   // staging of the host-buffer entry point
    uint8_t* hPin = nullptr;                    // ... and its pinned host twin (source rows, then destination rows)
    hipStream_t stream = nullptr;
};

namespace {

// fixed-point bilinear weights of OpenCV's BilinearTab_i in closed form: the float products (1-fy)(1-fx) ... of
// multiples of 1/32 are exact, so entry k = 32 * (5-bit factors); alpha 0 is the one table OpenCV's sum repair
// touches: {32767, 0, 0, 1}  (oracle/ivf_oracle.c builds the table the long way; tests compare)
__device__ __forceinline__ void remap_weights(unsigned a, int& w00, int& w01, int& w10, int& w11)
{
    const int fx = a & 31, fy = a >> 5;
    w00 = (32 - fy) * (32 - fx) * 32; w01 = (32 - fy) * fx * 32; w10 = fy * (32 - fx) * 32; w11 = fy * fx * 32;
    if (a == 0) { w00 = 32767; w11 = 1; }
}

// workgroup = 256 x 4 destination pixels; a wave owns 256 consecutive pixels of one row and lane i takes pixels
// i, i+64, i+128, i+192 of them, so that every map load, gather and store of the wave touches consecutive addresses
// (rectification maps are smooth: neighbouring destination pixels read neighbouring source pixels)
template <int CN>
__global__ __launch_bounds__(256) void k_remap(const uint32_t* __restrict__ xy, const uint16_t* __restrict__ al, int wp,
                                               const uint8_t* __restrict__ src, size_t srcImage, int sstride, int sw, int sh,
                                               uint8_t* __restrict__ dst, size_t dstImage, int dstride, int w, int h)
{
    const int xb = blockIdx.x * 256 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= h) return;
    const uint8_t* S = src + (size_t)blockIdx.z * srcImage;
    uint8_t* D = dst + (size_t)blockIdx.z * dstImage + (size_t)y * dstride;
    unsigned pk[4], ak[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = min(xb + 64 * k, wp - 1);                 // the padded map row is readable up to wp
        pk[k] = xy[(size_t)y * wp + x];
        ak[k] = al[(size_t)y * wp + x];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = xb + 64 * k;
        const int sx = (short)(pk[k] & 0xffffu), sy = (short)(pk[k] >> 16);
        int w00, w01, w10, w11;
        remap_weights(ak[k], w00, w01, w10, w11);
        const bool x0 = (unsigned)sx < (unsigned)sw, x1 = (unsigned)(sx + 1) < (unsigned)sw;
        const bool y0 = (unsigned)sy < (unsigned)sh, y1 = (unsigned)(sy + 1) < (unsigned)sh;
        const uint8_t* r0 = S + (size_t)(y0 ? sy : 0) * sstride;
        const uint8_t* r1 = S + (size_t)(y1 ? sy + 1 : 0) * sstride;
        const int c0 = (x0 ? sx : 0) * CN, c1 = (x1 ? sx + 1 : 0) * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) {
            // BORDER_CONSTANT 0: a tap outside the source contributes nothing
            const int t00 = (x0 && y0) ? r0[c0 + c] : 0, t01 = (x1 && y0) ? r0[c1 + c] : 0;
            const int t10 = (x0 && y1) ? r1[c0 + c] : 0, t11 = (x1 && y1) ? r1[c1 + c] : 0;
            const unsigned v = (unsigned)(t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11 + (1 << 14)) >> 15;   // <= 255
            if (x < w) D[x * CN + c] = (uint8_t)v;
        }
    }
}

inline int16_t sat_s16(int v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

int launch_remap(const ivf_remap* r, const uint8_t* dsrc, int sstride, size_t srcImage, uint8_t* ddst, int dstride,
                 size_t dstImage, int nImg, hipStream_t st)
{
    const dim3 grid((r->w + 255) / 256, (r->h + 3) / 4, nImg);
    if (r->cn == 1) k_remap<1><<<grid, 256, 0, st>>>(r->dXY, r->dA, r->wp, dsrc, srcImage, sstride, r->sw, r->sh, ddst, dstImage, dstride, r->w, r->h);
    else k_remap<3><<<grid, 256, 0, st>>>(r->dXY, r->dA, r->wp, dsrc, srcImage, sstride, r->sw, r->sh, ddst, dstImage, dstride, r->w, r->h);
    RHIPCHK(hipGetLastError());
    return IVF_OK;
}

}  // namespace

extern "C" {

int ivf_init_undistort_rectify_map(const double* K, const double* dist, int n_dist, const double* R, const double* P,
                                   int width, int height, float* map1, float* map2)
{
    if (!K || !P || !map1 || !map2) return rfail(IVF_E_INVALID, "null argument");
    if (width < 1 || height < 1) return rfail(IVF_E_INVALID, "bad map size %dx%d", width, height);
    if (!(n_dist == 0 || n_dist == 4 || n_dist == 5 || n_dist == 8 || n_dist == 12) || (n_dist > 0 && !dist))
        return rfail(IVF_E_INVALID, "distortion vector of %d coefficients (0, 4, 5, 8 or 12; tilt terms are not supported)", n_dist);
    // iR = (P * R)^-1 in double (3x3: cv::invert's closed form)
    double Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (R) memcpy(Rm, R, sizeof Rm);
    double A[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P[3 * i + k] * Rm[3 * k + j];
            A[3 * i + j] = s;
        }
    const double m00 = A[4] * A[8] - A[5] * A[7], m01 = A[3] * A[8] - A[5] * A[6], m02 = A[3] * A[7] - A[4] * A[6];
    double det = A[0] * m00 - A[1] * m01 + A[2] * m02;
    if (det == 0) return rfail(IVF_E_GEOMETRY, "P * R is singular");
    det = 1. / det;
    const double ir[9] = {m00 * det,                         (A[2] * A[7] - A[1] * A[8]) * det, (A[1] * A[5] - A[2] * A[4]) * det,
                          (A[5] * A[6] - A[3] * A[8]) * det, (A[0] * A[8] - A[2] * A[6]) * det, (A[2] * A[3] - A[0] * A[5]) * det,
                          (A[3] * A[7] - A[4] * A[6]) * det, (A[1] * A[6] - A[0] * A[7]) * det, (A[0] * A[4] - A[1] * A[3]) * det};
    double d[12] = {0};
    for (int i = 0; i < n_dist; i++) d[i] = dist[i];
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    for (int i = 0; i < height; i++) {
        double X = i * ir[1] + ir[2], Y = i * ir[4] + ir[5], W = i * ir[7] + ir[8];
        float* m1 = map1 + (size_t)i * width;
        float* m2 = map2 + (size_t)i * width;
        for (int j = 0; j < width; j++, X += ir[0], Y += ir[3], W += ir[6]) {
            const double iw = 1. / W, x = X * iw, y = Y * iw;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, xy2 = 2 * x * y;
            const double kr = (1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2) / (1 + ((d[7] * r2 + d[6]) * r2 + d[5]) * r2);
            const double xd = (x * kr + d[2] * xy2 + d[3] * (r2 + 2 * x2) + d[8] * r2 + d[9] * r2 * r2);
            const double yd = (y * kr + d[2] * (r2 + 2 * y2) + d[3] * xy2 + d[10] * r2 + d[11] * r2 * r2);
            m1[j] = (float)(fx * xd + u0);
            m2[j] = (float)(fy * yd + v0);
        }
    }
    return IVF_OK;
}

void ivf_remap_destroy(ivf_remap* r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    if (r->dXY) (void)hipFree(r->dXY);
    if (r->dA) (void)hipFree(r->dA);
    if (r->dSrc) (void)hipFree(r->dSrc);
    if (r->dDst) (void)hipFree(r->dDst);
    if (r->hPin) (void)hipHostFree(r->hPin);
    delete r;
}

int ivf_remap_create(const float* map1, const float* map2, int width, int height, int src_width, int src_height,
                     int channels, int device_id, ivf_remap** out)
{
    if (!out) return rfail(IVF_E_INVALID, "null argument");
    *out = nullptr;
    if (!map1 || !map2) return rfail(IVF_E_INVALID, "null map");
    if (width < 1 || height < 1 || src_width < 1 || src_height < 1 || src_width > 32767 || src_height > 32767)
        return rfail(IVF_E_INVALID, "bad sizes (map %dx%d, source %dx%d)", width, height, src_width, src_height);
    if (channels != 1 && channels != 3) return rfail(IVF_E_INVALID, "channels must be 1 or 3");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return rfail(IVF_E_NO_DEVICE, "no HIP device available (%s); libivfront has no CPU path",
                                                e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return rfail(IVF_E_INVALID, "device_id %d outside [0,%d)", device_id, n);
    RHIPCHK(hipSetDevice(device_id));
    ivf_remap* r = new ivf_remap();
    r->device = device_id; r->w = width; r->h = height; r->wp = (width + 3) & ~3;
    r->sw = src_width; r->sh = src_height; r->cn = channels;
    // what cv::remap's RemapInvoker derives from CV_32FC1 maps on every call, done once:
    // sx = cvRound(map1 * 32), sy = cvRound(map2 * 32); integer part saturated to short, 5 + 5 fraction bits
    std::vector<uint32_t> xy((size_t)r->wp * height, 0x80008000u);     // padding: far outside => 0
    std::vector<uint16_t> al((size_t)r->wp * height, 0);
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const int fx = (int)lrintf(map1[(size_t)y * width + x] * 32), fy = (int)lrintf(map2[(size_t)y * width + x] * 32);
            const uint16_t sx = (uint16_t)sat_s16(fx >> 5), sy = (uint16_t)sat_s16(fy >> 5);
            xy[(size_t)y * r->wp + x] = (uint32_t)sx | ((uint32_t)sy << 16);
            al[(size_t)y * r->wp + x] = (uint16_t)(((fy & 31) << 5) | (fx & 31));
        }
    const size_t srcBytes = (size_t)src_width * src_height * channels, dstBytes = (size_t)width * height * channels;
    if (hipMalloc(&r->dXY, xy.size() * 4) != hipSuccess || hipMalloc(&r->dA, al.size() * 2) != hipSuccess ||
        hipMalloc(&r->dSrc, srcBytes) != hipSuccess || hipMalloc(&r->dDst, dstBytes) != hipSuccess ||
        hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) {
        ivf_remap_destroy(r);
        return rfail(IVF_E_NO_DEVICE, "device allocation failed for a %dx%d remap", width, height);
    }
    if (hipMemcpy(r->dXY, xy.data(), xy.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(r->dA, al.data(), al.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
        ivf_remap_destroy(r);
        return rfail(IVF_E_NO_DEVICE, "map upload failed");
    }
    *out = r;
    return IVF_OK;
}

int ivf_remap_apply(ivf_remap* r, const uint8_t* src, int src_stride, uint8_t* dst, int dst_stride)
{
    if (!r || !src || !dst) return rfail(IVF_E_INVALID, "null argument");
    if (src_stride < r->sw * r->cn || dst_stride < r->w * r->cn) return rfail(IVF_E_INVALID, "stride smaller than a row");
    RHIPCHK(hipSetDevice(r->device));
    const size_t srow = (size_t)r->sw * r->cn, drow = (size_t)r->w * r->cn;
    // caller buffers are pageable (cv::Mat): rows go through a pinned staging buffer -- a pageable 2-D copy runs row by row
    const size_t sBytes = srow * r->sh, dBytes = drow * r->h;
    if (!r->hPin) RHIPCHK(hipHostMalloc((void**)&r->hPin, sBytes + dBytes, hipHostMallocDefault));
    for (int y = 0; y < r->sh; y++) memcpy(r->hPin + (size_t)y * srow, src + (size_t)y * src_stride, srow);
    RHIPCHK(hipMemcpyAsync(r->dSrc, r->hPin, sBytes, hipMemcpyHostToDevice, r->stream));
    int rc = launch_remap(r, r->dSrc, (int)srow, 0, r->dDst, (int)drow, 0, 1, r->stream);
    if (rc) return rc;
    RHIPCHK(hipMemcpyAsync(r->hPin + sBytes, r->dDst, dBytes, hipMemcpyDeviceToHost, r->stream));
    RHIPCHK(hipStreamSynchronize(r->stream));
    for (int y = 0; y < r->h; y++) memcpy(dst + (size_t)y * dst_stride, r->hPin + sBytes + (size_t)y * drow, drow);
    return IVF_OK;
}

int ivf_remap_apply_device(ivf_remap* r, const uint8_t* d_src, int src_stride, size_t src_image_stride, uint8_t* d_dst,
                     int dst_stride, size_t dst_image_stride, int n_images, void* hip_stream)
{
    if (!r || !d_src || !d_dst) return rfail(IVF_E_INVALID, "null argument");
    if (n_images < 1 || n_images > 65535) return rfail(IVF_E_INVALID, "n_images %d outside [1,65535]", n_images);
    if (src_stride < r->sw * r->cn || dst_stride < r->w * r->cn) return rfail(IVF_E_INVALID, "stride smaller than a row");
    RHIPCHK(hipSetDevice(r->device));
    return launch_remap(r, d_src, src_stride, src_image_stride, d_dst, dst_stride, dst_image_stride, n_images,
                        (hipStream_t)hip_stream);
}

int ivf_remap_get_fixed_maps(const ivf_remap* r, int16_t* xy, uint16_t* alpha)
{
    if (!r || !xy || !alpha) return rfail(IVF_E_INVALID, "null argument");
    RHIPCHK(hipSetDevice(r->device));
    std::vector<uint32_t> hx((size_t)r->wp * r->h);
    std::vector<uint16_t> ha((size_t)r->wp * r->h);
    RHIPCHK(hipMemcpy(hx.data(), r->dXY, hx.size() * 4, hipMemcpyDeviceToHost));
    RHIPCHK(hipMemcpy(ha.data(), r->dA, ha.size() * 2, hipMemcpyDeviceToHost));
    for (int y = 0; y < r->h; y++)
        for (int x = 0; x < r->w; x++) {
            const uint32_t p = hx[(size_t)y * r->wp + x];
            xy[((size_t)y * r->w + x) * 2] = (int16_t)(p & 0xffffu);
            xy[((size_t)y * r->w + x) * 2 + 1] = (int16_t)(p >> 16);
            alpha[(size_t)y * r->w + x] = ha[(size_t)y * r->wp + x];
        }
    return IVF_OK;
}

}  // extern "C"
