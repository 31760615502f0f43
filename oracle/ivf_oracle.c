/*
 * ivf_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see ivf_oracle.h for the parity statement).
 *
 * Plain-C restatement of the reference CPU algorithm.  Every function cites the reference
 * file:line it follows (ORB/ = /root/reference/introspective_ORB_SLAM/).  OpenCV primitives the
 * reference calls are restated from OpenCV 4.x's published plain-C++ algorithms ("frozen
 * semantics") because OpenCV is absent from this image: PARITY UNPINNED there.
 *
 * Build: gcc -O3 -ffp-contract=off (no FMA contraction: SURVEY.md Appendix D-10).
 */
#include "ivf_oracle.h"
#include <math.h>
#include <float.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * A-1  cvRound / cvFloor / cvCeil  (round-half-to-even; OpenCV fast_math.hpp)
 * ---------------------------------------------------------------------------------------------- */
int orc_cv_round_d(double v) { return (int)lrint(v); }
int orc_cv_round_f(float v) { return (int)lrintf(v); }
static int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
static int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
static int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }

/* ------------------------------------------------------------------------------------------------
 * A-8  cosf / sinf: glibc >= 2.28 single-precision sin/cos (ARM optimized-routines "sincosf"
 * algorithm, f64 polynomial after a fast reduction by pi/2).  Restated so the SAME arithmetic
 * runs on the device; checked exhaustively against this image's glibc 2.35 for every float in
 * [0, 6.2832] (tests/test_oracle_primitives.py::test_trig_matches_glibc).  Called at
 * ORB/src/ORBextractor.cc:113-114 through std::cos(float)/std::sin(float).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } sincos_tab;
static const sincos_tab SC[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0,
     -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16,
     -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static inline uint32_t abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }
static inline float sincos_poly(double x, double x2, const sincos_tab* p, int n)
{
    if ((n & 1) == 0) {
        double x3 = x * x2, s1 = p->s2 + x2 * p->s3, x7 = x3 * x2, s = x + x3 * p->s1;
        return (float)(s + x7 * s1);
    } else {
        double x4 = x2 * x2, c2 = p->c3 + x2 * p->c4, c1 = p->c0 + x2 * p->c1, x6 = x4 * x2, c = c1 + x4 * p->c2;
        return (float)(c + x6 * c2);
    }
}
static inline double reduce_fast(double x, const sincos_tab* p, int* np)
{
    double r = x * p->hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return x - n * p->hpi;
}
/* valid for |y| < 120 (the path only feeds [0, 2*pi)); larger |y| falls back to libm */
float orc_sinf(float y)
{
    double x = y; const sincos_tab* p = &SC[0]; int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sincos_poly(x, x * x, p, 0);
    }
    if (!(abstop12(y) < abstop12(120.0f))) return sinf(y);
    x = reduce_fast(x, p, &n);
    double s = p->sign[n & 3];
    if (n & 2) p = &SC[1];
    return sincos_poly(x * s, x * x, p, n);
}
float orc_cosf(float y)
{
    double x = y; const sincos_tab* p = &SC[0]; int n;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sincos_poly(x, x * x, p, 1);
    }
    if (!(abstop12(y) < abstop12(120.0f))) return cosf(y);
    x = reduce_fast(x, p, &n);
    double s = p->sign[n & 3];
    if (n & 2) p = &SC[1];
    return sincos_poly(x * s, x * x, p, n ^ 1);
}

/* ------------------------------------------------------------------------------------------------
 * A-12 logf: glibc >= 2.27 single-precision log (ARM optimized-routines "logf": 16-entry table of 1/c and log c, degree-3
 * polynomial in f64).  MapPoint::PredictScale (ORB/src/MapPoint.cc:398,415) calls an unqualified log(float) under the
 * `using namespace std` that DBoW2/TemplatedVocabulary.h:36 puts into every translation unit that includes Frame.h, i.e.
 * std::log(float) = logf, divides by the float mfLogScaleFactor = logf(1.2f) (Frame.cc:106) and takes std::ceil(float).
 * Restated so the SAME arithmetic runs on the device; checked against this image's glibc 2.35 for EVERY positive normal float
 * (tests/test_oracle_primitives.py::test_logf_matches_glibc_exhaustive).  Table = glibc's __logf_data (the values, not code).
 * ---------------------------------------------------------------------------------------------- */
static const double LOGF_T[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
/* valid for positive normal x (a distance ratio); anything else goes to libm */
float orc_logf(float x)
{
    uint32_t ix; memcpy(&ix, &x, 4);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) return logf(x);
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    float zf; memcpy(&zf, &iz, 4);
    const double z = (double)zf, invc = LOGF_T[i][0], logc = LOGF_T[i][1];
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

/* ------------------------------------------------------------------------------------------------
 * A-5  cv::fastAtan2 (OpenCV 3.x/4.x mathfuncs_core: degree-7 odd polynomial, f32, degrees).
 * Called at ORB/src/ORBextractor.cc:104.
 * ---------------------------------------------------------------------------------------------- */
/* OpenCV-version switches of the un-pinned primitives: the default is OpenCV >= 3.4.2 / 4.x;
 * the alternatives restate what older releases computed, so a maintainer whose OpenCV differs can check both sides.
 *   blur   0: [18,34,48,56,48,34,18]/256 (error-diffused fixed point)   1: cvRound(k*256) = [18,34,49,55,49,34,18] (<= 3.4.1)
 *   retain 0: nth_element(begin, begin + n - 1, end)                    1: nth_element(begin, begin + n, end) (2.4 / 3.x)
 *   atan   0: degree-7 polynomial (2.4.4+, 3.x, 4.x)                    1: x*y/(x^2 + 0.28 y^2) rational form (<= 2.4.3) */
#ifndef ORC_BUILD_ID
#define ORC_BUILD_ID "unstamped"
#endif
/* build provenance: hash of the checker's sources at compile time (oracle/Makefile); tests/oracle_lib.py compares it with the sources on disk */
const char* orc_build_id(void) { return ORC_BUILD_ID; }

static int g_var_blur = 0, g_var_retain = 0, g_var_atan = 0;
void orc_set_opencv_variant(int blur, int retain, int atan) { g_var_blur = blur; g_var_retain = retain; g_var_atan = atan; }

static float fast_atan2_legacy(float y, float x)
{
    double a, x2 = (double)x * x, y2 = (double)y * y;
    if (y2 <= x2) {
        a = (180. / 3.14159265358979323846) * x * y / (x2 + 0.28 * y2 + DBL_EPSILON);
        return (float)(x < 0 ? a + 180 : y >= 0 ? a : 360 + a);
    }
    a = (180. / 3.14159265358979323846) * x * y / (y2 + 0.28 * x2 + DBL_EPSILON);
    return (float)(y > 0 ? 90 - a : 270 - a);
}

float orc_fast_atan2(float y, float x)
{
    if (g_var_atan) return fast_atan2_legacy(y, x);
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ------------------------------------------------------------------------------------------------
 * A-2  cv::FAST(img, kps, t, true), TYPE_9_16 (OpenCV features2d/fast.cpp + fast_score.cpp).
 * Called at ORB/src/ORBextractor.cc:1045,1051 on a per-cell sub-image.
 * ---------------------------------------------------------------------------------------------- */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* cornerScore<16>: max over the 16 arcs of 9 contiguous ring pixels, both polarities, of the
 * minimum signed difference, floored at `threshold`, minus 1. */
static int fast_corner_score(const uint8_t* ptr, const int* pixel, int threshold)
{
    int d[25], v = ptr[0], k;
    for (k = 0; k < 25; k++) d[k] = v - ptr[pixel[k & 15]];
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) if (d[k + j] < a) a = d[k + j];
        int m = a < d[k] ? a : d[k];
        if (m > a0) a0 = m;
        m = a < d[k + 9] ? a : d[k + 9];
        if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] > b) b = d[k + 3];
        if (b >= b0) continue;
        for (int j = 4; j <= 8; j++) if (d[k + j] > b) b = d[k + j];
        int m = b > d[k] ? b : d[k];
        if (m < b0) b0 = m;
        m = b > d[k + 9] ? b : d[k + 9];
        if (m < b0) b0 = m;
    }
    return -b0 - 1;
}

/* segment test: >= 9 contiguous ring pixels all < v-t or all > v+t */
static int fast_is_corner(const uint8_t* ptr, const int* pixel, int t)
{
    int v = ptr[0], lo = v - t, hi = v + t;
#define CLS(k) ((ptr[pixel[k]] < lo) ? 1 : (ptr[pixel[k]] > hi) ? 2 : 0)
    int d = CLS(0) | CLS(8);
    if (!d) return 0;
    d &= CLS(2) | CLS(10); d &= CLS(4) | CLS(12); d &= CLS(6) | CLS(14);
    if (!d) return 0;
    d &= CLS(1) | CLS(9); d &= CLS(3) | CLS(11); d &= CLS(5) | CLS(13); d &= CLS(7) | CLS(15);
#undef CLS
    if (d & 1) {
        int count = 0;
        for (int k = 0; k < 25; k++) {
            if (ptr[pixel[k & 15]] < lo) { if (++count > 8) return 1; } else count = 0;
        }
    }
    if (d & 2) {
        int count = 0;
        for (int k = 0; k < 25; k++) {
            if (ptr[pixel[k & 15]] > hi) { if (++count > 8) return 1; } else count = 0;
        }
    }
    return 0;
}

void orc_fast_score_map(const uint8_t* img, int stride, int cols, int rows, int threshold, uint8_t* out)
{
    int pixel[16];
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    for (int k = 0; k < 16; k++) pixel[k] = RING_DX[k] + RING_DY[k] * stride;
    memset(out, 0, (size_t)cols * rows);
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (fast_is_corner(p, pixel, threshold))
                out[(size_t)y * cols + x] = (uint8_t)fast_corner_score(p, pixel, threshold);
        }
}

int orc_fast_detect(const uint8_t* img, int stride, int cols, int rows, int threshold, orc_keypoint* out, int cap)
{
    if (cols < 7 || rows < 7) return 0;
    uint8_t* s = (uint8_t*)malloc((size_t)cols * rows);
    orc_fast_score_map(img, stride, cols, rows, threshold, s);
    int n = 0;
    /* 3x3 strict non-max suppression; scores outside the detection domain are 0 (fast.cpp row buffers) */
    for (int y = 3; y < rows - 3; y++)
        for (int x = 3; x < cols - 3; x++) {
            int sc = s[(size_t)y * cols + x];
            if (!sc) continue;
            const uint8_t* r0 = s + (size_t)(y - 1) * cols + x;
            const uint8_t* r1 = r0 + cols;
            const uint8_t* r2 = r1 + cols;
            if (sc > r1[-1] && sc > r1[1] && sc > r0[-1] && sc > r0[0] && sc > r0[1] &&
                sc > r2[-1] && sc > r2[0] && sc > r2[1]) {
                if (n < cap) {
                    out[n].x = (float)x; out[n].y = (float)y; out[n].size = 7.f; out[n].angle = -1.f;
                    out[n].response = (float)sc; out[n].octave = 0;
                }
                n++;
            }
        }
    free(s);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * A-3  cv::resize(..., INTER_LINEAR), CV_8UC1 (OpenCV imgproc/resize.cpp: 11-bit fixed-point
 * coefficients, HResizeLinear into int, VResizeLinear<uchar,int,short,FixedPtCast<..,22>>).
 * Called at ORB/src/ORBextractor.cc:1311,1341.
 * ---------------------------------------------------------------------------------------------- */
static inline short sat_short_round(float v)
{
    int i = orc_cv_round_f(v);
    return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}
/* cv::resize with dsize given computes inv_scale = (double)dsize / ssize and then scale = 1. / inv_scale (imgproc/src/resize.cpp,
 * cv::resize + hal::resize): NOT (double)ssize / dsize -- the two doubles differ in the last bit for some (ssize, dsize). */
static inline double resize_scale(int ssize, int dsize) { const double inv_scale = (double)dsize / ssize; return 1. / inv_scale; }

/* KAT hook (tests/test_oracle_primitives.py): number of destination indices of a ssize -> dsize axis whose (floor, 11-bit
 * coefficient pair) differ between OpenCV's scale = 1. / ((double)dsize / ssize) and the direct quotient (double)ssize / dsize
 * that rounds 1-3 of this repository used.  0 for every pair the pyramid can produce => goldens and fixtures of those rounds stand. */
int orc_resize_coef_mismatches(int ssize, int dsize)
{
    const double s0 = resize_scale(ssize, dsize), s1 = (double)ssize / dsize;
    int bad = 0;
    for (int d = 0; d < dsize; d++) {
        float f0 = (float)((d + 0.5) * s0 - 0.5), f1 = (float)((d + 0.5) * s1 - 0.5);
        const int i0 = cv_floor_f(f0), i1 = cv_floor_f(f1);
        f0 -= i0; f1 -= i1;
        if (i0 != i1 || orc_cv_round_f((1.f - f0) * 2048.f) != orc_cv_round_f((1.f - f1) * 2048.f) ||
            orc_cv_round_f(f0 * 2048.f) != orc_cv_round_f(f1 * 2048.f)) bad++;
    }
    return bad;
}

void orc_resize_linear_8u(const uint8_t* src, int sstride, int sw, int sh, uint8_t* dst, int dstride, int dw, int dh)
{
    double scale_x = resize_scale(sw, dw), scale_y = resize_scale(sh, dh);
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* alpha = (short*)malloc(sizeof(short) * 2 * dw);
    int* row0 = (int*)malloc(sizeof(int) * dw);
    int* row1 = (int*)malloc(sizeof(int) * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = sat_short_round((1.f - fx) * 2048.f);
        alpha[2 * dx + 1] = sat_short_round(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        short b0 = sat_short_round((1.f - fy) * 2048.f), b1 = sat_short_round(fy * 2048.f);
        int y0 = sy < 0 ? 0 : sy > sh - 1 ? sh - 1 : sy;
        int y1 = sy + 1 < 0 ? 0 : sy + 1 > sh - 1 ? sh - 1 : sy + 1;
        const uint8_t* S0 = src + (size_t)y0 * sstride;
        const uint8_t* S1 = src + (size_t)y1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
            row0[dx] = S0[sx] * alpha[2 * dx] + S0[sx1] * alpha[2 * dx + 1];
            row1[dx] = S1[sx] * alpha[2 * dx] + S1[sx1] * alpha[2 * dx + 1];
        }
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(alpha); free(row0); free(row1);
}

/* ------------------------------------------------------------------------------------------------
 * A-4  cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101), CV_8U (OpenCV >= 3.4.2 / 4.x fixed-point
 * path: 8.8 kernel with error diffusion [18,34,48,56,48,34,18]/256, exact horizontal pass,
 * 16.16 vertical pass rounded (+32768)>>16).  Called at ORB/src/ORBextractor.cc:1277.
 * ---------------------------------------------------------------------------------------------- */
static const int GK4[7] = {18, 34, 48, 56, 48, 34, 18}, GK3[7] = {18, 34, 49, 55, 49, 34, 18};
static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}
void orc_gauss7_8u(const uint8_t* src, int sstride, int w, int h, uint8_t* dst, int dstride)
{
    const int* GK = g_var_blur ? GK3 : GK4;
    uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint16_t* t = tmp + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            if (x >= 3 && x < w - 3) for (int k = 0; k < 7; k++) acc += GK[k] * s[x + k - 3];
            else for (int k = 0; k < 7; k++) acc += GK[k] * s[reflect101(x + k - 3, w)];
            t[x] = (uint16_t)acc;
        }
    }
    for (int y = 0; y < h; y++) {
        const uint16_t* r[7];
        for (int k = 0; k < 7; k++) r[k] = tmp + (size_t)reflect101(y + k - 3, h) * w;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int k = 0; k < 7; k++) acc += (uint32_t)GK[k] * r[k][x];
            const uint32_t v = (acc + 32768u) >> 16;
            d[x] = (uint8_t)(v > 255u ? 255u : v);             /* the sum-257 table can reach 257 on a 255 plateau: saturate_cast */
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * A-6  std::nth_element (libstdc++ bits/stl_algo.h: __introselect, median-of-3 to first,
 * __unguarded_partition, depth limit 2*lg(n), __heap_select fallback, __insertion_sort for <= 3)
 * with cv::KeypointResponseGreater, then cv::KeyPointsFilter::retainBest (OpenCV 4.x) and the
 * reference's resize (ORB/src/ORBextractor.cc:1146-1148, 1164-1165).
 * Pinned against this image's libstdc++ by oracle/stl_pin.cpp.
 * ---------------------------------------------------------------------------------------------- */
#define GT(a, b) ((a).response > (b).response)
static inline void kswap(orc_keypoint* a, orc_keypoint* b) { orc_keypoint t = *a; *a = *b; *b = t; }

static void move_median_to_first(orc_keypoint* result, orc_keypoint* a, orc_keypoint* b, orc_keypoint* c)
{
    if (GT(*a, *b)) {
        if (GT(*b, *c)) kswap(result, b);
        else if (GT(*a, *c)) kswap(result, c);
        else kswap(result, a);
    } else if (GT(*a, *c)) kswap(result, a);
    else if (GT(*b, *c)) kswap(result, c);
    else kswap(result, b);
}
static orc_keypoint* unguarded_partition(orc_keypoint* first, orc_keypoint* last, orc_keypoint* pivot)
{
    for (;;) {
        while (GT(*first, *pivot)) ++first;
        --last;
        while (GT(*pivot, *last)) --last;
        if (!(first < last)) return first;
        kswap(first, last);
        ++first;
    }
}
static void push_heap_(orc_keypoint* first, long hole, long top, orc_keypoint value)
{
    long parent = (hole - 1) / 2;
    while (hole > top && GT(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
static void adjust_heap_(orc_keypoint* first, long hole, long len, orc_keypoint value)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (GT(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value);
}
static void heap_select_(orc_keypoint* first, orc_keypoint* middle, orc_keypoint* last)
{
    long len = middle - first;
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            orc_keypoint v = first[parent];
            adjust_heap_(first, parent, len, v);
            if (parent == 0) break;
            parent--;
        }
    }
    for (orc_keypoint* i = middle; i < last; ++i)
        if (GT(*i, *first)) {
            orc_keypoint v = *i;
            *i = *first;
            adjust_heap_(first, 0, len, v);
        }
}
static void insertion_sort_(orc_keypoint* first, orc_keypoint* last)
{
    if (first == last) return;
    for (orc_keypoint* i = first + 1; i != last; ++i) {
        if (GT(*i, *first)) {
            orc_keypoint v = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(orc_keypoint));
            *first = v;
        } else {
            orc_keypoint v = *i;
            orc_keypoint* l = i;
            orc_keypoint* nx = i - 1;
            while (GT(v, *nx)) { *l = *nx; l = nx; --nx; }
            *l = v;
        }
    }
}
void orc_nth_element_resp(orc_keypoint* v, int n, int nth)
{
    if (n <= 0 || nth >= n) return;
    orc_keypoint *first = v, *last = v + n, *pn = v + nth;
    long depth = 0;
    for (long t = n; t > 1; t >>= 1) depth++;   /* std::__lg(n) */
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {
            heap_select_(first, pn + 1, last);
            kswap(first, pn);
            return;
        }
        --depth;
        orc_keypoint* mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1);
        orc_keypoint* cut = unguarded_partition(first + 1, last, first);
        if (cut <= pn) first = cut; else last = cut;
    }
    insertion_sort_(first, last);
}
int orc_retain_best(orc_keypoint* v, int n, int n_points)
{
    if (n_points >= 0 && n > n_points) {
        if (n_points == 0) return 0;
        orc_nth_element_resp(v, n, g_var_retain ? n_points : n_points - 1);
        /* std::partition of the tail only reorders elements past n_points, which the reference's
         * resize(n_points) then drops (ORBextractor.cc:1147-1148,1165): the survivors are v[0..n_points) */
        return n_points;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * a11  ORBmatcher::DescriptorDistance (ORB/src/ORBmatcher.cc:1700-1716): SWAR popcount of 8 x u32
 * ---------------------------------------------------------------------------------------------- */
int orc_hamming256(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

static const int8_t PATTERN31[1024] = {
#include "../include/ivf_pattern31.inc"
};
const int8_t* orc_bit_pattern_31(void) { return PATTERN31; }

/* ------------------------------------------------------------------------------------------------
 * Extractor state
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int w, h; uint8_t* data; } plane;
struct orc_extractor {
    orc_params p;
    double scale_factor_d;                 /* the member is `double scaleFactor` (ORB/include/ORBextractor.h:110) */
    float scale[ORC_MAX_LEVELS], inv_scale[ORC_MAX_LEVELS], sigma2[ORC_MAX_LEVELS], inv_sigma2[ORC_MAX_LEVELS];
    int nfeat[ORC_MAX_LEVELS];
    int umax[16];
    plane pyr[ORC_MAX_LEVELS], qpyr[ORC_MAX_LEVELS];
    int quality_available;
    int level_count[ORC_MAX_LEVELS];
};

/* ORB/src/ORBextractor.cc:411-476 */
orc_extractor* orc_extractor_create(const orc_params* p)
{
    if (!p || p->nlevels < 1 || p->nlevels > ORC_MAX_LEVELS) return NULL;
    orc_extractor* e = (orc_extractor*)calloc(1, sizeof(*e));
    e->p = *p;
    e->scale_factor_d = (double)p->scale_factor;
    const int nl = p->nlevels;
    e->scale[0] = 1.0f; e->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        e->scale[i] = (float)((double)e->scale[i - 1] * e->scale_factor_d);   /* :424 float*double */
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < nl; i++) {
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    float factor = (float)(1.0 / e->scale_factor_d);                                              /* :437 */
    float nDesired = (float)p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl)); /* :438 */
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        e->nfeat[l] = orc_cv_round_f(nDesired);
        sum += e->nfeat[l];
        nDesired *= factor;
    }
    e->nfeat[nl - 1] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0;
    /* umax (:458-475), HALF_PATCH_SIZE = 15 */
    int v, v0, vmax = cv_floor_d(15 * sqrtf(2.f) / 2 + 1), vmin = cv_ceil_d(15 * sqrtf(2.f) / 2);
    const double hp2 = 15 * 15;
    for (v = 0; v <= vmax; ++v) e->umax[v] = orc_cv_round_d(sqrt(hp2 - v * v));
    for (v = 15, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}
void orc_extractor_destroy(orc_extractor* e)
{
    if (!e) return;
    for (int l = 0; l < ORC_MAX_LEVELS; l++) { free(e->pyr[l].data); free(e->qpyr[l].data); }
    free(e);
}
int orc_extractor_levels(const orc_extractor* e) { return e->p.nlevels; }
void orc_extractor_tables(const orc_extractor* e, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                          int* features_per_level, int* umax16)
{
    for (int l = 0; l < e->p.nlevels; l++) {
        if (scale) scale[l] = e->scale[l];
        if (inv_scale) inv_scale[l] = e->inv_scale[l];
        if (sigma2) sigma2[l] = e->sigma2[l];
        if (inv_sigma2) inv_sigma2[l] = e->inv_sigma2[l];
        if (features_per_level) features_per_level[l] = e->nfeat[l];
    }
    if (umax16) memcpy(umax16, e->umax, sizeof(int) * 16);
}
int orc_pyramid_level(const orc_extractor* e, int level, const uint8_t** data, int* w, int* h)
{
    if (level < 0 || level >= e->p.nlevels || !e->pyr[level].data) return -1;
    *data = e->pyr[level].data; *w = e->pyr[level].w; *h = e->pyr[level].h;
    return 0;
}
int orc_quality_level(const orc_extractor* e, int level, const uint8_t** data, int* w, int* h)
{
    if (level < 0 || level >= e->p.nlevels || !e->qpyr[level].data) return -1;
    *data = e->qpyr[level].data; *w = e->qpyr[level].w; *h = e->qpyr[level].h;
    return 0;
}
int orc_level_count(const orc_extractor* e, int level) { return e->level_count[level]; }

/* ComputePyramid / ComputeQualityImagePyramid (ORB/src/ORBextractor.cc:1298-1357).  The 19-px
 * reflect-101 border the reference adds is never read by the live path, so
 * planes are stored un-padded. */
static void build_pyramid(const orc_extractor* e, plane* pyr, const uint8_t* img, int w, int h, int stride)
{
    for (int l = 0; l < e->p.nlevels; l++) {
        float s = e->inv_scale[l];
        int lw = orc_cv_round_f((float)w * s), lh = orc_cv_round_f((float)h * s);
        free(pyr[l].data);
        pyr[l].w = lw; pyr[l].h = lh;
        pyr[l].data = (uint8_t*)malloc((size_t)(lw > 0 ? lw : 1) * (lh > 0 ? lh : 1));
        if (l == 0) for (int y = 0; y < h; y++) memcpy(pyr[0].data + (size_t)y * w, img + (size_t)y * stride, w);
        else orc_resize_linear_8u(pyr[l - 1].data, pyr[l - 1].w, pyr[l - 1].w, pyr[l - 1].h, pyr[l].data, lw, lw, lh);
    }
}

/* IC_Angle (ORB/src/ORBextractor.cc:78-105) */
static float ic_angle(const plane* im, float px, float py, const int* umax)
{
    int m_01 = 0, m_10 = 0;
    const int step = im->w;
    const uint8_t* center = im->data + (size_t)orc_cv_round_f(py) * step + orc_cv_round_f(px);
    for (int u = -15; u <= 15; ++u) m_10 += u * center[u];
    for (int v = 1; v <= 15; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor (ORB/src/ORBextractor.cc:109-148): x*b + y*a is evaluated WITHOUT fma */
static void orb_descriptor(const orc_keypoint* kp, const uint8_t* img, int step, uint8_t* desc)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = kp->angle * factorPI;
    float a = orc_cosf(angle), b = orc_sinf(angle);
    const uint8_t* center = img + (size_t)orc_cv_round_f(kp->y) * step + orc_cv_round_f(kp->x);
    const int8_t* pat = PATTERN31;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; k++, pat += 4) {
            float x0 = pat[0], y0 = pat[1], x1 = pat[2], y1 = pat[3];
            int t0 = center[orc_cv_round_f(x0 * b + y0 * a) * step + orc_cv_round_f(x0 * a - y0 * b)];
            int t1 = center[orc_cv_round_f(x1 * b + y1 * a) * step + orc_cv_round_f(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

static unsigned long roi_sum(const plane* im, int x0, int y0, int x1, int y1)
{
    unsigned long s = 0;
    for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) s += im->data[(size_t)y * im->w + x];
    return s;
}

typedef struct { orc_keypoint* v; int n, cap; } kvec;
static void kvec_reserve(kvec* k, int cap) { if (cap > k->cap) { k->v = (orc_keypoint*)realloc(k->v, sizeof(orc_keypoint) * cap); k->cap = cap; } }

/* ComputeKeyPointsOld (ORB/src/ORBextractor.cc:880-1213) for one level; returns -3 if a cell window
 * leaves the image (OpenCV would throw there). */
static int keypoints_level(orc_extractor* e, int level, kvec* out)
{
    const plane* im = &e->pyr[level];
    const int introspect = e->quality_available && e->p.enable_introspection;
    const float imageRatio = (float)e->pyr[0].w / e->pyr[0].h;
    const int nDesired = e->nfeat[level];
    const int levelCols = (int)sqrtf((float)nDesired / (5 * imageRatio));
    const int levelRows = (int)(imageRatio * levelCols);
    out->n = 0;
    if (levelCols <= 0 || levelRows <= 0) return 0;
    const int minBorderX = 19, minBorderY = 19, maxBorderX = im->w - 19, maxBorderY = im->h - 19;
    const int W = maxBorderX - minBorderX, H = maxBorderY - minBorderY;
    const int cellW = (int)ceilf((float)W / levelCols), cellH = (int)ceilf((float)H / levelRows);
    const int nCells = levelRows * levelCols;
    const int nfeaturesCell = (int)ceilf((float)nDesired / nCells);

    kvec* cellKP = (kvec*)calloc(nCells, sizeof(kvec));
    int* nToRetain = (int*)calloc(nCells, sizeof(int));
    int* nTotal = (int*)calloc(nCells, sizeof(int));
    char* bNoMore = (char*)calloc(nCells, 1);
    int* iniXCol = (int*)calloc(levelCols, sizeof(int));
    int* iniYRow = (int*)calloc(levelRows, sizeof(int));
    float* nfeatures_cell = (float*)malloc(sizeof(float) * nCells);
    float* cell_weights = (float*)calloc(nCells, sizeof(float));
    for (int c = 0; c < nCells; c++) nfeatures_cell[c] = (float)nfeaturesCell;
    int nNoMore = 0, nToDistribute = 0, rc = 0;
    float hY = (float)(cellH + 6);            /* :935 -- NOT reset by the main loop (Appendix D-2) */
    float cell_weights_sum = 0.0f;

    if (introspect) {                          /* :946-987 */
        const plane* q = &e->qpyr[level];
        for (int i = 0; i < levelRows; i++) {
            const float iniY = (float)(minBorderY + i * cellH - 3);
            iniYRow[i] = (int)iniY;
            if (i == levelRows - 1) { hY = maxBorderY + 3 - iniY; if (hY <= 0) continue; }
            float hX = (float)(cellW + 6);
            for (int j = 0; j < levelCols; j++) {
                float iniX;
                if (i == 0) { iniX = (float)(minBorderX + j * cellW - 3); iniXCol[j] = (int)iniX; }
                else iniX = (float)iniXCol[j];
                if (j == levelCols - 1) { hX = maxBorderX + 3 - iniX; if (hX <= 0) continue; }
                int y0 = (int)iniY, y1 = (int)(iniY + hY), x0 = (int)iniX, x1 = (int)(iniX + hX);
                if (x0 < 0 || y0 < 0 || x1 > q->w || y1 > q->h || x1 < x0 || y1 < y0) { rc = -3; goto done; }
                unsigned long sum = roi_sum(q, x0, y0, x1, y1);
                float cost = (float)sum / (float)(hX * hY);
                float qual_score = (float)(1.0 / (1.0 + (double)(cost / 255)));      /* :981 double then narrowed */
                float qual_score_norm = 2 * qual_score - 1;
                cell_weights[i * levelCols + j] = qual_score_norm;
                cell_weights_sum += qual_score_norm;
            }
        }
    }

    for (int i = 0; i < levelRows; i++) {      /* :989-1101 */
        const float iniY = (float)(minBorderY + i * cellH - 3);
        iniYRow[i] = (int)iniY;
        if (i == levelRows - 1) { hY = maxBorderY + 3 - iniY; if (hY <= 0) continue; }
        float hX = (float)(cellW + 6);
        for (int j = 0; j < levelCols; j++) {
            const int c = i * levelCols + j;
            float iniX;
            if (i == 0) { iniX = (float)(minBorderX + j * cellW - 3); iniXCol[j] = (int)iniX; }
            else iniX = (float)iniXCol[j];
            if (j == levelCols - 1) { hX = maxBorderX + 3 - iniX; if (hX <= 0) continue; }
            if (introspect)
                nfeatures_cell[c] = fmaxf(1.0f, ceilf((float)nDesired * cell_weights[c] / cell_weights_sum));
            int y0 = (int)iniY, y1 = (int)(iniY + hY), x0 = (int)iniX, x1 = (int)(iniX + hX);
            /* a cell whose FAST domain leaves [19, dim-19) makes the reference read outside the
             * blurred clone (UB) or throw in rowRange/colRange: reported as unsupported geometry */
            if (x0 < 0 || y0 < 0 || x1 > maxBorderX + 3 || y1 > maxBorderY + 3 || x1 < x0 || y1 < y0) { rc = -3; goto done; }
            const uint8_t* cell = im->data + (size_t)y0 * im->w + x0;
            const int cw = x1 - x0, ch = y1 - y0;
            kvec* kv = &cellKP[c];
            int maxk = ((cw > 6 ? cw - 6 : 0) * (ch > 6 ? ch - 6 : 0)) + 1;
            kvec_reserve(kv, maxk);
            kv->n = orc_fast_detect(cell, im->w, cw, ch, e->p.ini_th_fast, kv->v, kv->cap);
            if (kv->n <= 3) kv->n = orc_fast_detect(cell, im->w, cw, ch, e->p.min_th_fast, kv->v, kv->cap);
            if (introspect) {                  /* :1058-1080 */
                const plane* q = &e->qpyr[level];
                for (int k = 0; k < kv->n; k++) {
                    float cost = (float)q->data[(size_t)(y0 + (int)kv->v[k].y) * q->w + (x0 + (int)kv->v[k].x)];
                    kv->v[k].response *= 2 * (1.0f / (1.0f + cost / 255.0f)) - 1;
                }
            }
            const int nKeys = kv->n;
            nTotal[c] = nKeys;
            if ((float)nKeys > nfeatures_cell[c]) { nToRetain[c] = (int)nfeatures_cell[c]; bNoMore[c] = 0; }
            else {
                nToRetain[c] = nKeys;
                nToDistribute = (int)((float)nToDistribute + (nfeatures_cell[c] - (float)nKeys));
                bNoMore[c] = 1; nNoMore++;
            }
        }
    }

    while (nToDistribute > 0 && nNoMore < nCells) {      /* :1103-1133 (runs at most once) */
        for (int c = 0; c < nCells; c++) {
            if (!bNoMore[c]) {
                int nNew = (int)(nfeatures_cell[c] + ceilf((float)nToDistribute / (nCells - nNoMore)));
                if (nTotal[c] > nNew) { nToRetain[c] = nNew; bNoMore[c] = 0; }
                else {
                    nToRetain[c] = nTotal[c];
                    nToDistribute += nNew - nTotal[c];
                    bNoMore[c] = 1; nNoMore++;
                }
            }
        }
        nToDistribute = 0;
    }

    {
        int total = 0;
        for (int c = 0; c < nCells; c++) total += cellKP[c].n;
        kvec_reserve(out, total + 1);
    }
    const int scaledPatchSize = (int)(31 * e->scale[level]);          /* :1142 */
    for (int i = 0; i < levelRows; i++)
        for (int j = 0; j < levelCols; j++) {
            kvec* kv = &cellKP[i * levelCols + j];
            kv->n = orc_retain_best(kv->v, kv->n, nToRetain[i * levelCols + j]);
            for (int k = 0; k < kv->n; k++) {
                orc_keypoint kp = kv->v[k];
                kp.x += iniXCol[j]; kp.y += iniYRow[i];
                kp.octave = level; kp.size = (float)scaledPatchSize;
                out->v[out->n++] = kp;
            }
        }
    if (out->n > nDesired) out->n = orc_retain_best(out->v, out->n, nDesired);   /* :1162-1166 */
done:
    for (int c = 0; c < nCells; c++) free(cellKP[c].v);
    free(cellKP); free(nToRetain); free(nTotal); free(bNoMore); free(iniXCol); free(iniYRow);
    free(nfeatures_cell); free(cell_weights);
    return rc;
}

/* ORBextractor::operator() (ORB/src/ORBextractor.cc:1224-1296) */
int orc_extract(orc_extractor* e, const uint8_t* img, int w, int h, int stride,
                const uint8_t* cost, int cost_stride, orc_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    *n_out = 0;
    if (!img || w <= 0 || h <= 0) return 0;                /* :1227 empty image -> silent return */
    if (cost && e->p.enable_introspection) {               /* :1231-1238 */
        e->quality_available = 1;
        build_pyramid(e, e->qpyr, cost, w, h, cost_stride);
    } else e->quality_available = 0;
    build_pyramid(e, e->pyr, img, w, h, stride);

    const int nl = e->p.nlevels;
    kvec lv[ORC_MAX_LEVELS];
    memset(lv, 0, sizeof(lv));
    int rc = 0, total = 0;
    for (int l = 0; l < nl && rc == 0; l++) {
        if (e->pyr[l].w < 39 || e->pyr[l].h < 39) { e->level_count[l] = 0; continue; }
        rc = keypoints_level(e, l, &lv[l]);
        e->level_count[l] = lv[l].n;
        total += lv[l].n;
    }
    if (rc == 0 && total > cap) rc = -2;
    if (rc == 0) {
        for (int l = 0; l < nl; l++)                        /* computeOrientation :1209-1210 */
            for (int k = 0; k < lv[l].n; k++) lv[l].v[k].angle = ic_angle(&e->pyr[l], lv[l].v[k].x, lv[l].v[k].y, e->umax);
        int offset = 0;
        for (int l = 0; l < nl; l++) {
            if (lv[l].n == 0) continue;
            const plane* im = &e->pyr[l];
            uint8_t* blur = (uint8_t*)malloc((size_t)im->w * im->h);
            orc_gauss7_8u(im->data, im->w, im->w, im->h, blur, im->w);      /* :1276-1277 */
            for (int k = 0; k < lv[l].n; k++) orb_descriptor(&lv[l].v[k], blur, im->w, desc + (size_t)(offset + k) * 32);
            free(blur);
            if (l != 0) {
                float sc = e->scale[l];
                for (int k = 0; k < lv[l].n; k++) { lv[l].v[k].x *= sc; lv[l].v[k].y *= sc; }
            }
            memcpy(kps + offset, lv[l].v, sizeof(orc_keypoint) * lv[l].n);
            offset += lv[l].n;
        }
        *n_out = offset;
    }
    for (int l = 0; l < nl; l++) free(lv[l].v);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * a12  Frame::ComputeStereoMatches (ORB/src/Frame.cc:758-932)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int dist, idx; } dist_idx;
static int cmp_dist_idx(const void* a, const void* b)
{
    const dist_idx *x = (const dist_idx*)a, *y = (const dist_idx*)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : x->idx > y->idx;
}
int orc_stereo_match(const orc_extractor* eL, const orc_extractor* eR,
                     const orc_keypoint* kpL, int nL, const uint8_t* descL,
                     const orc_keypoint* kpR, int nR, const uint8_t* descR,
                     float bf, float b, float* u_right, float* depth)
{
    for (int i = 0; i < nL; i++) { u_right[i] = -1.0f; depth[i] = -1.0f; }
    if (!eL->pyr[0].data || !eR->pyr[0].data) return -1;
    const int thOrbDist = (100 + 50) / 2;
    const int nRows = eL->pyr[0].h;
    /* row table: candidates per row in increasing iR (:767-785) */
    int* rowCount = (int*)calloc(nRows, sizeof(int));
    int* minr = (int*)malloc(sizeof(int) * (nR > 0 ? nR : 1));
    int* maxr = (int*)malloc(sizeof(int) * (nR > 0 ? nR : 1));
    for (int iR = 0; iR < nR; iR++) {
        const float kpY = kpR[iR].y;
        const float r = 2.0f * eL->scale[kpR[iR].octave];
        maxr[iR] = (int)ceilf(kpY + r);
        minr[iR] = (int)floorf(kpY - r);
        for (int yi = minr[iR]; yi <= maxr[iR]; yi++) if (yi >= 0 && yi < nRows) rowCount[yi]++;
    }
    int* rowStart = (int*)malloc(sizeof(int) * (nRows + 1));
    rowStart[0] = 0;
    for (int y = 0; y < nRows; y++) rowStart[y + 1] = rowStart[y] + rowCount[y];
    int* rowIdx = (int*)malloc(sizeof(int) * (rowStart[nRows] > 0 ? rowStart[nRows] : 1));
    memset(rowCount, 0, sizeof(int) * nRows);
    for (int iR = 0; iR < nR; iR++)
        for (int yi = minr[iR]; yi <= maxr[iR]; yi++) if (yi >= 0 && yi < nRows) rowIdx[rowStart[yi] + rowCount[yi]++] = iR;

    const float minZ = b, minD = 0, maxD = bf / minZ;
    dist_idx* vDistIdx = (dist_idx*)malloc(sizeof(dist_idx) * (nL > 0 ? nL : 1));
    int nDist = 0;

    for (int iL = 0; iL < nL; iL++) {
        const int levelL = kpL[iL].octave;
        const float vL = kpL[iL].y, uL = kpL[iL].x;
        const int row = (int)vL;
        if (row < 0 || row >= nRows) continue;
        const int nC = rowCount[row];
        if (nC == 0) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100;                        /* ORBmatcher::TH_HIGH */
        int bestIdxR = 0;
        const uint8_t* dL = descL + (size_t)iL * 32;
        for (int iC = 0; iC < nC; iC++) {
            const int iR = rowIdx[rowStart[row] + iC];
            if (kpR[iR].octave < levelL - 1 || kpR[iR].octave > levelL + 1) continue;
            const float uR = kpR[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_hamming256(dL, descR + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {                /* :844 sub-pixel match by correlation */
            const float uR0 = kpR[bestIdxR].x;
            const float scaleFactor = eL->inv_scale[levelL];
            const float scaleduL = roundf(kpL[iL].x * scaleFactor);
            const float scaledvL = roundf(kpL[iL].y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const plane* PL = &eL->pyr[levelL];
            const plane* PR = &eR->pyr[levelL];
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= PR->w) continue;
            const int yl = (int)(scaledvL - w), xl = (int)(scaleduL - w);
            /* windows must lie inside the level (OpenCV would throw otherwise) */
            if (yl < 0 || yl + 11 > PL->h || xl < 0 || xl + 11 > PL->w) continue;
            if ((int)(scaleduR0 - L - w) < 0) continue;
            const int cLv = PL->data[(size_t)(yl + w) * PL->w + xl + w];
            int bestD = INT_MAX, bestincR = 0;
            float vDists[11];
            for (int incR = -L; incR <= L; incR++) {
                const int xr = (int)(scaleduR0 + incR - w);
                const int cRv = PR->data[(size_t)(yl + w) * PR->w + xr + w];
                int acc = 0;
                for (int yy = 0; yy < 11; yy++)
                    for (int xx = 0; xx < 11; xx++) {
                        int dl = PL->data[(size_t)(yl + yy) * PL->w + xl + xx] - cLv;
                        int dr = PR->data[(size_t)(yl + yy) * PR->w + xr + xx] - cRv;
                        acc += abs(dl - dr);
                    }
                float dist = (float)acc;
                if (dist < (float)bestD) { bestD = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = eL->scale[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01);      /* Frame.cc:909 computes uL-0.01 in double and narrows */ }
                depth[iL] = bf / disparity;
                u_right[iL] = bestuR;
                vDistIdx[nDist].dist = bestD; vDistIdx[nDist].idx = iL; nDist++;
            }
        }
    }
    if (nDist > 0) {                               /* :918-931; empty list => no gate (Appendix D-8) */
        qsort(vDistIdx, nDist, sizeof(dist_idx), cmp_dist_idx);
        const float median = (float)vDistIdx[nDist / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nDist - 1; i >= 0; i--) {
            if ((float)vDistIdx[i].dist < thDist) break;
            u_right[vDistIdx[i].idx] = -1; depth[vDistIdx[i].idx] = -1;
        }
    }
    free(rowCount); free(minr); free(maxr); free(rowStart); free(rowIdx); free(vDistIdx);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a16  Frame grid: AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea (ORB/src/Frame.cc:415-430,
 * 615-680; grid 64 x 48, ORB/include/Frame.h:43-44)
 * ---------------------------------------------------------------------------------------------- */
#define GRID_COLS 64
#define GRID_ROWS 48
typedef struct { int* start; int* idx; float inv_w, inv_h; } grid_t;
static void grid_build(grid_t* g, const orc_keypoint* kps, int n, const orc_bounds* bd)
{
    g->inv_w = (float)GRID_COLS / (bd->max_x - bd->min_x);
    g->inv_h = (float)GRID_ROWS / (bd->max_y - bd->min_y);
    g->start = (int*)calloc(GRID_COLS * GRID_ROWS + 1, sizeof(int));
    g->idx = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    int* cell = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        int px = (int)roundf((kps[i].x - bd->min_x) * g->inv_w), py = (int)roundf((kps[i].y - bd->min_y) * g->inv_h);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) { cell[i] = -1; continue; }
        cell[i] = px * GRID_ROWS + py;
        g->start[cell[i] + 1]++;
    }
    for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) g->start[c + 1] += g->start[c];
    int* fill = (int*)calloc(GRID_COLS * GRID_ROWS, sizeof(int));
    for (int i = 0; i < n; i++) if (cell[i] >= 0) g->idx[g->start[cell[i]] + fill[cell[i]]++] = i;
    free(fill); free(cell);
}
static void grid_free(grid_t* g) { free(g->start); free(g->idx); }
static int grid_query(const grid_t* g, const orc_keypoint* kps, const orc_bounds* bd,
                      float x, float y, float r, int minLevel, int maxLevel, int32_t* out, int cap)
{
    int n = 0;
    int nMinCellX = (int)floorf((x - bd->min_x - r) * g->inv_w); if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - bd->min_x + r) * g->inv_w); if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - bd->min_y - r) * g->inv_h); if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - bd->min_y + r) * g->inv_h); if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * GRID_ROWS + iy;
            for (int j = g->start[c]; j < g->start[c + 1]; j++) {
                const orc_keypoint* kp = &kps[g->idx[j]];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float distx = kp->x - x, disty = kp->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) { if (n < cap) out[n] = g->idx[j]; n++; }
            }
        }
    return n;
}
int orc_features_in_area(const orc_keypoint* kps, int n, const orc_bounds* bounds,
                         float x, float y, float r, int min_level, int max_level, int32_t* out, int cap)
{
    grid_t g;
    grid_build(&g, kps, n, bounds);
    int c = grid_query(&g, kps, bounds, x, y, r, min_level, max_level, out, cap);
    grid_free(&g);
    return c;
}

/* a17  ORBmatcher::ComputeThreeMaxima (ORB/src/ORBmatcher.cc:1654-1695) */
void orc_three_maxima(const int* histo, int L, int* ind1, int* ind2, int* ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* a13  ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORB/src/ORBmatcher.cc:1372-1518)
 * on already-projected queries.  Per query i (a last-frame map point that passed the frustum tests):
 *   q_u,q_v   projected pixel; q_ur = u - bf*invzc; q_radius = th*scale[lastOctave];
 *   q_min_level/q_max_level  the GetFeaturesInArea level arguments chosen at :1429-1434;
 *   q_angle   LastFrame.mvKeysUn[i].angle; q_desc the map point descriptor;
 *   q_blocks  1 when pMP->Observations()>0 (its assignment blocks later queries, :1447-1449).
 * cur_assign[i2]: in/out; -1 free, -2 pre-occupied by a blocking map point, >=0 query index. */
int orc_search_by_projection(const orc_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                             const orc_bounds* bounds,
                             int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                             const int32_t* q_min_level, const int32_t* q_max_level,
                             const float* q_angle, const uint8_t* q_desc,
                             const uint8_t* q_valid, const uint8_t* q_blocks,
                             int check_orientation, int32_t* cur_assign, int* nmatches_out)
{
    int nmatches = 0;
    enum { HISTO_LENGTH = 30 };
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n_q > 0 ? n_q : 1)); rotN[i] = 0; }
    const float factor = 1.0f / HISTO_LENGTH;
    grid_t g;
    grid_build(&g, cur_kps, n_cur, bounds);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n_cur > 0 ? n_cur : 1));
    for (int i = 0; i < n_q; i++) {
        if (q_valid && !q_valid[i]) continue;
        const float u = q_u[i], v = q_v[i], radius = q_radius[i];
        int nc = grid_query(&g, cur_kps, bounds, u, v, radius, q_min_level[i], q_max_level[i], cand, n_cur);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = cand[k];
            if (cur_assign[i2] == -2) continue;
            if (cur_assign[i2] >= 0 && (!q_blocks || q_blocks[cur_assign[i2]])) continue;
            if (cur_uright[i2] > 0) {
                const float er = fabsf(q_ur[i] - cur_uright[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_hamming256(q_desc + (size_t)i * 32, cur_desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= 100) {
            cur_assign[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = q_angle[i] - cur_kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin][rotN[bin]++] = bestIdx2;
            }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < rotN[i]; j++) { cur_assign[rotHist[i][j]] = -1; nmatches--; }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(cand);
    grid_free(&g);
    *nmatches_out = nmatches;
    return 0;
}

/* f1  ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORB/src/ORBmatcher.cc:410-519).
 * Only octave-0 keypoints of F1 search (:426-428), in the window of F2's grid around prev_xy[i1] at level 0 (:430);
 * best / second best skip candidates already matched at a smaller-or-equal distance (:449-450); a match steals the
 * candidate from its previous owner (:468-472); rotation histogram with the reference's 1/HISTO_LENGTH factor and C
 * round() (:480-487); prev_xy is updated from the surviving matches (:514-516).  Returns nmatches. */
int orc_search_for_initialization(const orc_keypoint* k1, const uint8_t* d1, int n1,
                                  const orc_keypoint* k2, const uint8_t* d2, int n2, const orc_bounds* bounds2,
                                  float* prev_xy, int window_size, float nn_ratio, int check_orientation,
                                  int32_t* matches12, int* nmatches_out)
{
    enum { HISTO_LENGTH = 30, TH_LOW = 50 };
    int nmatches = 0;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1)); rotN[i] = 0; }
    const float factor = 1.0f / HISTO_LENGTH;
    int* matchedDist = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int* matches21 = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    for (int i = 0; i < n2; i++) { matchedDist[i] = 2147483647; matches21[i] = -1; }
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    grid_t g;
    grid_build(&g, k2, n2, bounds2);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1));
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = k1[i1].octave;
        if (level1 > 0) continue;
        const int nc = grid_query(&g, k2, bounds2, prev_xy[2 * i1], prev_xy[2 * i1 + 1], (float)window_size, level1, level1, cand, n2);
        if (nc == 0) continue;
        int bestDist = 2147483647, bestDist2 = 2147483647, bestIdx2 = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = cand[k];
            const int dist = orc_hamming256(d1 + (size_t)i1 * 32, d2 + (size_t)i2 * 32);
            if (matchedDist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * nn_ratio) {
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
            matches12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matchedDist[bestIdx2] = bestDist;
            nmatches++;
            if (check_orientation) {
                float rot = k1[i1].angle - k2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin][rotN[bin]++] = i1;
            }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < rotN[i]; j++) {
                    const int idx1 = rotHist[i][j];
                    if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
                }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) { prev_xy[2 * i1] = k2[matches12[i1]].x; prev_xy[2 * i1 + 1] = k2[matches12[i1]].y; }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(cand); free(matchedDist); free(matches21);
    grid_free(&g);
    *nmatches_out = nmatches;
    return 0;
}

/* f3  ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORB/src/ORBmatcher.cc:296-404) on
 * already-projected candidates (the caller runs :318-365: Sim3 transform, depth / image / distance / viewing-angle tests,
 * PredictScale).  Per query: window KeyFrame::GetFeaturesInArea(u, v, radius) (no level arguments, KeyFrame.cc:606-645),
 * candidates already holding a map point are skipped (:379-380), octave in [level-1, level] (:384-385), first minimum,
 * accepted when <= TH_LOW (:399-403).  matched[idx]: in/out, -1 free, -2 occupied on entry, >= 0 query index. */
int orc_search_keyframe_points(const orc_keypoint* kf_kps, const uint8_t* kf_desc, int n_kf, const orc_bounds* bounds,
                               int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                               const uint8_t* q_desc, const uint8_t* q_valid, int32_t* matched, int* nmatches_out)
{
    enum { TH_LOW = 50 };
    int nmatches = 0;
    grid_t g;
    grid_build(&g, kf_kps, n_kf, bounds);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n_kf > 0 ? n_kf : 1));
    for (int i = 0; i < n_q; i++) {
        if (q_valid && !q_valid[i]) continue;
        const int nc = grid_query(&g, kf_kps, bounds, q_u[i], q_v[i], q_radius[i], -1, -1, cand, n_kf);
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = cand[k];
            if (matched[idx] != -1) continue;
            const int kpLevel = kf_kps[idx].octave;
            if (kpLevel < q_level[i] - 1 || kpLevel > q_level[i]) continue;
            const int dist = orc_hamming256(q_desc + (size_t)i * 32, kf_desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { matched[bestIdx] = i; nmatches++; }
    }
    free(cand);
    grid_free(&g);
    *nmatches_out = nmatches;
    return 0;
}

/* f4  ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (ORB/src/ORBmatcher.cc:831-982), the matching core :893-955 on
 * already-projected map points: window, octave in [level-1, level], chi-square gate on the reprojection error with
 * mvInvLevelSigma2 (7.8 with a right coordinate, 5.99 without, :918-938), first minimum; best_idx = -1 unless
 * best_dist <= TH_LOW.  inv_level_sigma2 == NULL (then kf_uright / q_ur may be NULL too): the core of
 * Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint) (:983-1106, :1064-1084), which has no gate.  The Replace / AddObservation bookkeeping (:958-977) stays with the caller. */
int orc_fuse_candidates(const orc_keypoint* kf_kps, const uint8_t* kf_desc, const float* kf_uright, int n_kf,
                        const orc_bounds* bounds, const float* inv_level_sigma2,
                        int n_q, const float* q_u, const float* q_v, const float* q_ur, const float* q_radius,
                        const int32_t* q_level, const uint8_t* q_desc, const uint8_t* q_valid,
                        int32_t* best_idx, int32_t* best_dist)
{
    enum { TH_LOW = 50 };
    grid_t g;
    grid_build(&g, kf_kps, n_kf, bounds);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n_kf > 0 ? n_kf : 1));
    for (int i = 0; i < n_q; i++) {
        best_idx[i] = -1; best_dist[i] = 256;
        if (q_valid && !q_valid[i]) continue;
        const float u = q_u[i], v = q_v[i], ur = q_ur ? q_ur[i] : 0.0f;
        const int nc = grid_query(&g, kf_kps, bounds, u, v, q_radius[i], -1, -1, cand, n_kf);
        int bestDist = 256, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = cand[k];
            const orc_keypoint* kp = &kf_kps[idx];
            const int kpLevel = kp->octave;
            if (kpLevel < q_level[i] - 1 || kpLevel > q_level[i]) continue;
            if (!inv_level_sigma2) {
                /* Fuse(KF, Scw, ...) (:983-1106): no reprojection gate */
            } else if (kf_uright[idx] >= 0) {
                const float ex = u - kp->x, ey = v - kp->y, er = ur - kf_uright[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - kp->x, ey = v - kp->y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
            }
            const int dist = orc_hamming256(q_desc + (size_t)i * 32, kf_desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_dist[i] = bestDist;
        if (bestDist <= TH_LOW) best_idx[i] = bestIdx;
    }
    free(cand);
    grid_free(&g);
    return 0;
}

/* f5  ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORB/src/ORBmatcher.cc:1145-1254): the
 * caller projects KF1's unmatched map points into KF2 and vice versa (:1193-1230, :1273-1310); each direction is a
 * window search with octave in [level-1, level], first minimum, accepted when <= TH_HIGH (:1233-1264); a pair is a
 * match only when both directions agree (:1336-1349).  matches12[i1] = index in KF2 or -1; returns nFound. */
static void window_best(const grid_t* g, const orc_keypoint* kps, const uint8_t* desc, int n, const orc_bounds* bd, int n_q,
                        const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level, const uint8_t* q_desc,
                        const uint8_t* q_valid, int th, int32_t* cand, int32_t* best)
{
    for (int i = 0; i < n_q; i++) {
        best[i] = -1;
        if (q_valid && !q_valid[i]) continue;
        const int nc = grid_query(g, kps, bd, q_u[i], q_v[i], q_radius[i], -1, -1, cand, n);
        int bestDist = 2147483647, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = cand[k];
            if (kps[idx].octave < q_level[i] - 1 || kps[idx].octave > q_level[i]) continue;
            const int dist = orc_hamming256(q_desc + (size_t)i * 32, desc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= th) best[i] = bestIdx;
    }
}
int orc_search_by_sim3(const orc_keypoint* k1, const uint8_t* d1, int n1, const orc_bounds* b1,
                       const orc_keypoint* k2, const uint8_t* d2, int n2, const orc_bounds* b2,
                       const float* q12_u, const float* q12_v, const float* q12_radius, const int32_t* q12_level,
                       const uint8_t* q12_desc, const uint8_t* q12_valid,
                       const float* q21_u, const float* q21_v, const float* q21_radius, const int32_t* q21_level,
                       const uint8_t* q21_desc, const uint8_t* q21_valid, int32_t* matches12, int* nfound)
{
    enum { TH_HIGH = 100 };
    grid_t g1, g2;
    grid_build(&g1, k1, n1, b1); grid_build(&g2, k2, n2, b2);
    const int nmax = n1 > n2 ? n1 : n2;
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (nmax > 0 ? nmax : 1));
    int32_t* m1 = (int32_t*)malloc(sizeof(int32_t) * (n1 > 0 ? n1 : 1));
    int32_t* m2 = (int32_t*)malloc(sizeof(int32_t) * (n2 > 0 ? n2 : 1));
    window_best(&g2, k2, d2, n2, b2, n1, q12_u, q12_v, q12_radius, q12_level, q12_desc, q12_valid, TH_HIGH, cand, m1);
    window_best(&g1, k1, d1, n1, b1, n2, q21_u, q21_v, q21_radius, q21_level, q21_desc, q21_valid, TH_HIGH, cand, m2);
    int nf = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        matches12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { matches12[i1] = idx2; nf++; }
    }
    free(cand); free(m1); free(m2);
    grid_free(&g1); grid_free(&g2);
    *nfound = nf;
    return 0;
}

/* f6  ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vpMapPointMatches) (ORB/src/ORBmatcher.cc:165-294).
 * DBoW2::FeatureVector (std::map<NodeId, vector<unsigned>>) of each side in CSR form: node ids ascending (map order),
 * start offsets, feature indices.  Nodes present on both sides are walked in ascending order (:187-265); inside a node
 * every KF feature with a good map point looks for its best / second best among the frame's features of that node
 * that are still unmatched (:215-217), TH_LOW + ratio test (:236-238), rotation histogram of the FRAME indices with
 * the reference's 1/HISTO_LENGTH factor (:244-253) and the final three-maxima filter (:268-288).
 * f_match[iF] = KF keypoint index whose map point the frame keypoint receives, or -1. */
static int rot_bin(float a1, float a2)
{
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * (1.0f / 30));
    if (bin == 30) bin = 0;
    return bin;
}
int orc_search_by_bow(const orc_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_has_mp, int n_kf,
                      const int32_t* kf_node, const int32_t* kf_start, const int32_t* kf_idx, int kf_nodes,
                      const orc_keypoint* f_kps, const uint8_t* f_desc, int n_f,
                      const int32_t* f_node, const int32_t* f_start, const int32_t* f_idx, int f_nodes,
                      float nn_ratio, int check_orientation, int32_t* f_match, int* nmatches_out)
{
    enum { HISTO_LENGTH = 30, TH_LOW = 50 };
    int nmatches = 0;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n_f > 0 ? n_f : 1)); rotN[i] = 0; }
    for (int i = 0; i < n_f; i++) f_match[i] = -1;
    int a = 0, b = 0;
    while (a < kf_nodes && b < f_nodes) {
        if (kf_node[a] == f_node[b]) {
            for (int p = kf_start[a]; p < kf_start[a + 1]; p++) {
                const int realIdxKF = kf_idx[p];
                if (!kf_has_mp[realIdxKF]) continue;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int q = f_start[b]; q < f_start[b + 1]; q++) {
                    const int realIdxF = f_idx[q];
                    if (f_match[realIdxF] >= 0) continue;
                    const int dist = orc_hamming256(kf_desc + (size_t)realIdxKF * 32, f_desc + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW && (float)bestDist1 < nn_ratio * (float)bestDist2) {
                    f_match[bestIdxF] = realIdxKF;
                    if (check_orientation) {
                        const int bin = rot_bin(kf_kps[realIdxKF].angle, f_kps[bestIdxF].angle);
                        rotHist[bin][rotN[bin]++] = bestIdxF;
                    }
                    nmatches++;
                }
            }
            a++; b++;
        } else if (kf_node[a] < f_node[b]) {
            while (a < kf_nodes && kf_node[a] < f_node[b]) a++;          /* lower_bound */
        } else {
            while (b < f_nodes && f_node[b] < kf_node[a]) b++;
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < rotN[i]; j++) { f_match[rotHist[i][j]] = -1; nmatches--; }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    *nmatches_out = nmatches;
    return 0;
}

/* f7  ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (ORB/src/ORBmatcher.cc:528-661): as f6 but
 * both sides need a good map point (:560-564, :577-582), side-2 features are claimed through vbMatched2 (:604), the
 * acceptance is bestDist1 < TH_LOW (strict, :598), the histogram and the output are indexed by the KF1 feature
 * (:606, :619).  matches12[idx1] = idx2 or -1. */
int orc_search_by_bow_keyframes(const orc_keypoint* k1, const uint8_t* d1, const uint8_t* has_mp1, int n1,
                                const int32_t* node1, const int32_t* start1, const int32_t* idx1v, int nodes1,
                                const orc_keypoint* k2, const uint8_t* d2, const uint8_t* has_mp2, int n2,
                                const int32_t* node2, const int32_t* start2, const int32_t* idx2v, int nodes2,
                                float nn_ratio, int check_orientation, int32_t* matches12, int* nmatches_out)
{
    enum { HISTO_LENGTH = 30, TH_LOW = 50 };
    int nmatches = 0;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1)); rotN[i] = 0; }
    uint8_t* matched2 = (uint8_t*)calloc(n2 > 0 ? n2 : 1, 1);
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    int a = 0, b = 0;
    while (a < nodes1 && b < nodes2) {
        if (node1[a] == node2[b]) {
            for (int p = start1[a]; p < start1[a + 1]; p++) {
                const int i1 = idx1v[p];
                if (!has_mp1[i1]) continue;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int q = start2[b]; q < start2[b + 1]; q++) {
                    const int i2 = idx2v[q];
                    if (matched2[i2] || !has_mp2[i2]) continue;
                    const int dist = orc_hamming256(d1 + (size_t)i1 * 32, d2 + (size_t)i2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = i2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW && (float)bestDist1 < nn_ratio * (float)bestDist2) {
                    matches12[i1] = bestIdx2;
                    matched2[bestIdx2] = 1;
                    if (check_orientation) {
                        const int bin = rot_bin(k1[i1].angle, k2[bestIdx2].angle);
                        rotHist[bin][rotN[bin]++] = i1;
                    }
                    nmatches++;
                }
            }
            a++; b++;
        } else if (node1[a] < node2[b]) {
            while (a < nodes1 && node1[a] < node2[b]) a++;
        } else {
            while (b < nodes2 && node2[b] < node1[a]) b++;
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < rotN[i]; j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(matched2);
    *nmatches_out = nmatches;
    return 0;
}

/* f8  ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (ORB/src/ORBmatcher.cc:663-829)
 * with CheckDistEpipolarLine (:146-163).  Features WITHOUT a map point on both sides (:697-701, :721-725); a candidate
 * must not be worse than the best so far nor than TH_LOW (:736-737: ties replace), mono-mono pairs too close to the
 * epipole are dropped (:741-747), the epipolar distance gate is 3.84 * mvLevelSigma2[octave2] (:162).  vbMatched2 is
 * never set in this reference (:765-767), so several KF1 features may share a KF2 feature.  matches12[idx1] = idx2 / -1. */
static int epipolar_ok(const orc_keypoint* kp1, const orc_keypoint* kp2, const float* F, const float* level_sigma2)
{
    const float a = kp1->x * F[0] + kp1->y * F[3] + F[6];
    const float b = kp1->x * F[1] + kp1->y * F[4] + F[7];
    const float c = kp1->x * F[2] + kp1->y * F[5] + F[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * level_sigma2[kp2->octave];
}
int orc_search_for_triangulation(const orc_keypoint* k1, const uint8_t* d1, const uint8_t* has_mp1, const uint8_t* stereo1, int n1,
                                 const int32_t* node1, const int32_t* start1, const int32_t* idx1v, int nodes1,
                                 const orc_keypoint* k2, const uint8_t* d2, const uint8_t* has_mp2, const uint8_t* stereo2, int n2,
                                 const int32_t* node2, const int32_t* start2, const int32_t* idx2v, int nodes2,
                                 const float* F12, float ex, float ey, const float* scale_factors2, const float* level_sigma2_2,
                                 int only_stereo, int check_orientation, int32_t* matches12, int* nmatches_out)
{
    enum { HISTO_LENGTH = 30, TH_LOW = 50 };
    int nmatches = 0;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1)); rotN[i] = 0; }
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    int a = 0, b = 0;
    while (a < nodes1 && b < nodes2) {
        if (node1[a] == node2[b]) {
            for (int p = start1[a]; p < start1[a + 1]; p++) {
                const int i1 = idx1v[p];
                if (has_mp1[i1]) continue;
                const int bStereo1 = stereo1[i1];
                if (only_stereo && !bStereo1) continue;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int q = start2[b]; q < start2[b + 1]; q++) {
                    const int i2 = idx2v[q];
                    if (has_mp2[i2]) continue;
                    const int bStereo2 = stereo2[i2];
                    if (only_stereo && !bStereo2) continue;
                    const int dist = orc_hamming256(d1 + (size_t)i1 * 32, d2 + (size_t)i2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - k2[i2].x, distey = ey - k2[i2].y;
                        if (distex * distex + distey * distey < 100 * scale_factors2[k2[i2].octave]) continue;
                    }
                    if (epipolar_ok(&k1[i1], &k2[i2], F12, level_sigma2_2)) { bestIdx2 = i2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    matches12[i1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        const int bin = rot_bin(k1[i1].angle, k2[bestIdx2].angle);
                        rotHist[bin][rotN[bin]++] = i1;
                    }
                }
            }
            a++; b++;
        } else if (node1[a] < node2[b]) {
            while (a < nodes1 && node1[a] < node2[b]) a++;
        } else {
            while (b < nodes2 && node2[b] < node1[a]) b++;
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < rotN[i]; j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    *nmatches_out = nmatches;
    return 0;
}

/* f9  ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist)
 * (ORB/src/ORBmatcher.cc:1520-1652, relocalisation) on projected map points: window with levels [pred-1, pred+1] (:1571),
 * any occupied keypoint is skipped (:1585-1586), first minimum accepted when <= ORBdist (:1597), rotation histogram
 * against the keyframe keypoint's angle (:1604-1612), three-maxima filter (:1618-1637).
 * cur_assign[i2]: in/out, -1 free, -2 occupied on entry, >= 0 query index. */
int orc_search_by_projection_reloc(const orc_keypoint* cur_kps, const uint8_t* cur_desc, int n_cur, const orc_bounds* bounds,
                                   int n_q, const float* q_u, const float* q_v, const float* q_radius, const int32_t* q_level,
                                   const float* q_angle, const uint8_t* q_desc, const uint8_t* q_valid,
                                   int orb_dist, int check_orientation, int32_t* cur_assign, int* nmatches_out)
{
    enum { HISTO_LENGTH = 30 };
    int nmatches = 0;
    int* rotHist[HISTO_LENGTH]; int rotN[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) { rotHist[i] = (int*)malloc(sizeof(int) * (n_q > 0 ? n_q : 1)); rotN[i] = 0; }
    grid_t g;
    grid_build(&g, cur_kps, n_cur, bounds);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n_cur > 0 ? n_cur : 1));
    for (int i = 0; i < n_q; i++) {
        if (q_valid && !q_valid[i]) continue;
        const int nc = grid_query(&g, cur_kps, bounds, q_u[i], q_v[i], q_radius[i], q_level[i] - 1, q_level[i] + 1, cand, n_cur);
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = cand[k];
            if (cur_assign[i2] != -1) continue;
            const int dist = orc_hamming256(q_desc + (size_t)i * 32, cur_desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist && bestIdx2 >= 0) {
            cur_assign[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                const int bin = rot_bin(q_angle[i], cur_kps[bestIdx2].angle);
                rotHist[bin][rotN[bin]++] = bestIdx2;
            }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        orc_three_maxima(rotN, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j = 0; j < rotN[i]; j++) { cur_assign[rotHist[i][j]] = -1; nmatches--; }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(rotHist[i]);
    free(cand);
    grid_free(&g);
    *nmatches_out = nmatches;
    return 0;
}

/* f10  DBoW2 TemplatedVocabulary<FORB>::transform(feature, word_id, weight, nid, levelsup)
 * (ORB/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1259) as called for every descriptor by Frame::ComputeBoW /
 * KeyFrame::ComputeBoW (ORB/src/Frame.cc:683-694, KeyFrame.cc:66-77: levelsup = 4): from the root, at every level the child
 * with the smallest Hamming distance (first minimum, :1237-1248); the node reached at level L - levelsup is recorded
 * (:1227-1228: root when that level is <= 0; :1250-1251); the leaf gives word id and weight (:1255-1257).
 * The tree as flat arrays: children of node i = child[child_start[i] .. child_start[i+1]) (none = leaf). */
int orc_bow_transform(int n_nodes, const int32_t* child_start, const int32_t* child, const uint8_t* node_desc,
                      const int32_t* node_word, const double* node_weight, int depth_L,
                      const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id, double* weight)
{
    const int nid_level = depth_L - levelsup;
    for (int f = 0; f < n; f++) {
        int nid = 0, final_id = 0, current_level = 0;
        if (child_start[1] == child_start[0]) return -1;                   /* a root without children is not a vocabulary */
        do {
            ++current_level;
            const int c0 = child_start[final_id], c1 = child_start[final_id + 1];
            int best = child[c0];
            int best_d = orc_hamming256(desc + (size_t)f * 32, node_desc + (size_t)best * 32);
            for (int c = c0 + 1; c < c1; c++) {
                const int id = child[c];
                const int d = orc_hamming256(desc + (size_t)f * 32, node_desc + (size_t)id * 32);
                if (d < best_d) { best_d = d; best = id; }
            }
            final_id = best;
            if (current_level == nid_level) nid = final_id;
        } while (child_start[final_id + 1] != child_start[final_id]);
        (void)n_nodes;
        word_id[f] = node_word[final_id];
        weight[f] = node_weight[final_id];
        node_id[f] = nid_level <= 0 ? 0 : nid;
    }
    return 0;
}

/* f2  MapPoint::ComputeDistinctiveDescriptors (ORB/src/MapPoint.cc:247-312): among n observed descriptors the one with
 * the least median Hamming distance to the rest; median = sorted row [ (int)(0.5*(n-1)) ] (the row holds the 0 of the
 * diagonal), first minimum wins (:294-305). */
static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }
int orc_distinctive_descriptor(const uint8_t* desc, int n, int* best_idx, int* best_median)
{
    if (n <= 0) return -1;
    int* row = (int*)malloc(sizeof(int) * n);
    int bestMedian = 2147483647, bestIdx = 0;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) row[j] = i == j ? 0 : orc_hamming256(desc + (size_t)i * 32, desc + (size_t)j * 32);
        qsort(row, n, sizeof(int), cmp_int);
        const int median = row[(int)(0.5 * (n - 1))];
        if (median < bestMedian) { bestMedian = median; bestIdx = i; }
    }
    free(row);
    *best_idx = bestIdx;
    if (best_median) *best_median = bestMedian;
    return 0;
}

/* a14  ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th) (ORB/src/ORBmatcher.cc:45-135)
 * on already-projected queries.  Per query i (a map point with mbTrackInView and !isBad()):
 *   q_u,q_v = mTrackProjX/Y; q_ur = mTrackProjXR; q_radius = r*mvScaleFactors[nPredictedLevel] with
 *   r = RadiusByViewingCos(mTrackViewCos) (*th when th != 1) (:63-70); q_level = nPredictedLevel (window levels
 *   [level-1, level]); best / second-best with the ratio test only when both sit in the same octave (:117-121). */
int orc_search_map_points(const orc_keypoint* cur_kps, const uint8_t* cur_desc, const float* cur_uright, int n_cur,
                          const orc_bounds* bounds, int n_q, const float* q_u, const float* q_v, const float* q_ur,
                          const float* q_radius, const int32_t* q_level, const uint8_t* q_desc,
                          const uint8_t* q_valid, const uint8_t* q_blocks, float nn_ratio,
                          int32_t* cur_assign, int* nmatches_out)
{
    int nmatches = 0;
    grid_t g;
    grid_build(&g, cur_kps, n_cur, bounds);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (n_cur > 0 ? n_cur : 1));
    for (int i = 0; i < n_q; i++) {
        if (q_valid && !q_valid[i]) continue;
        const int level = q_level[i];
        int nc = grid_query(&g, cur_kps, bounds, q_u[i], q_v[i], q_radius[i], level - 1, level, cand, n_cur);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = cand[k];
            if (cur_assign[idx] == -2) continue;
            if (cur_assign[idx] >= 0 && (!q_blocks || q_blocks[cur_assign[idx]])) continue;
            if (cur_uright[idx] > 0) {
                const float er = fabsf(q_ur[i] - cur_uright[idx]);
                if (er > q_radius[i]) continue;
            }
            const int dist = orc_hamming256(q_desc + (size_t)i * 32, cur_desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = cur_kps[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = cur_kps[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= 100) {
            if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;
            cur_assign[bestIdx] = i;
            nmatches++;
        }
    }
    free(cand);
    grid_free(&g);
    *nmatches_out = nmatches;
    return 0;
}

/* a18  ORBmatcher::UpdateQualityScores(Frame &F) (ORB/src/ORBmatcher.cc:1108-1121): for every keypoint i with a map point
 * (assign[i] >= 0 = index into mp_quality): q = min(mapPointQ, kpQ); the map point is updated only when it changes by
 * more than 0.01; the keypoint score always becomes q.  Sequential: later keypoints see earlier updates. */
void orc_update_quality_scores(const int32_t* assign, int n, float* kp_quality, float* mp_quality)
{
    const float kDeltaThresh = 0.01f;
    for (int i = 0; i < n; i++) {
        if (assign[i] < 0) continue;
        const float mpt = mp_quality[assign[i]];
        const float upd = mpt < kp_quality[i] ? mpt : kp_quality[i];
        if (fabsf(upd - mpt) > kDeltaThresh) mp_quality[assign[i]] = upd;
        kp_quality[i] = upd;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Rectification: cv::initUndistortRectifyMap + cv::remap as the reference's driver calls them
 * (introspective_ORB_SLAM/Examples/Stereo/stereo_kitti.cc:285-343, :462-468, :519-521).  OpenCV is un-vendored and
 * absent here, so these follow the plain C++ code paths of OpenCV 4.x as recalled: parity
 * unpinned, like the other OpenCV primitives.
 * ---------------------------------------------------------------------------------------------- */
static void orc_mat3_mul(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a[3 * i + k] * b[3 * k + j];
            c[3 * i + j] = s;
        }
}
/* cv::invert, 3x3 double, DECOMP_LU: closed form through the determinant */
static int orc_mat3_inv(const double* S, double* t)
{
#define S_(i, j) S[3 * (i) + (j)]
    double d = S_(0, 0) * (S_(1, 1) * S_(2, 2) - S_(1, 2) * S_(2, 1)) - S_(0, 1) * (S_(1, 0) * S_(2, 2) - S_(1, 2) * S_(2, 0)) +
               S_(0, 2) * (S_(1, 0) * S_(2, 1) - S_(1, 1) * S_(2, 0));
    if (d == 0) return -1;
    d = 1. / d;
    t[0] = (S_(1, 1) * S_(2, 2) - S_(1, 2) * S_(2, 1)) * d;
    t[1] = (S_(0, 2) * S_(2, 1) - S_(0, 1) * S_(2, 2)) * d;
    t[2] = (S_(0, 1) * S_(1, 2) - S_(0, 2) * S_(1, 1)) * d;
    t[3] = (S_(1, 2) * S_(2, 0) - S_(1, 0) * S_(2, 2)) * d;
    t[4] = (S_(0, 0) * S_(2, 2) - S_(0, 2) * S_(2, 0)) * d;
    t[5] = (S_(0, 2) * S_(1, 0) - S_(0, 0) * S_(1, 2)) * d;
    t[6] = (S_(1, 0) * S_(2, 1) - S_(1, 1) * S_(2, 0)) * d;
    t[7] = (S_(0, 1) * S_(2, 0) - S_(0, 0) * S_(2, 1)) * d;
    t[8] = (S_(0, 0) * S_(1, 1) - S_(0, 1) * S_(1, 0)) * d;
#undef S_
    return 0;
}

int orc_init_undistort_rectify_map(const double* K, const double* dist, int n_dist, const double* R, const double* P,
                                   int w, int h, float* map1, float* map2)
{
    if (!K || !P || !map1 || !map2 || w < 1 || h < 1) return -1;
    if (!(n_dist == 0 || n_dist == 4 || n_dist == 5 || n_dist == 8 || n_dist == 12) || (n_dist && !dist)) return -1;
    static const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double PR[9], ir[9];
    orc_mat3_mul(P, R ? R : eye, PR);
    if (orc_mat3_inv(PR, ir)) return -1;
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    double k[12] = {0};
    for (int i = 0; i < n_dist; i++) k[i] = dist[i];
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double s1 = k[8], s2 = k[9], s3 = k[10], s4 = k[11];
    for (int i = 0; i < h; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double iw = 1. / _w, x = _x * iw, y = _y * iw;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2);
            const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2);
            /* no sensor tilt: matTilt = I, invProj = 1 */
            const double u = fx * xd + u0, v = fy * yd + v0;
            map1[(size_t)i * w + j] = (float)u;
            map2[(size_t)i * w + j] = (float)v;
        }
    }
    return 0;
}

static int16_t orc_sat_s16(long v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }

void orc_remap_weight_table(int16_t* T)
{
    /* initInterTab2D(INTER_LINEAR, fixpt): the 1-D table holds (1 - f, f), f = k/32 in float; the 2x2 products are
     * scaled by 2^15 and saturated to short; a table whose entries do not sum to 2^15 gets the difference added to
     * its extreme entry, searched over rows/cols ksize/2 .. ksize/2+1 of a ksize = 2 table -- i.e. over flat indices
     * 3..6, three of which belong to the NEXT (not yet written, still zero) table.  Only alpha = 0 is affected
     * (32768 saturates to 32767): it becomes {32767, 0, 0, 1}. */
    float tab1[32][2];
    for (int i = 0; i < 32; i++) { const float f = (float)i * (1.f / 32); tab1[i][0] = 1.f - f; tab1[i][1] = f; }
    int16_t buf[1024 * 4 + 8];
    memset(buf, 0, sizeof buf);
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int16_t* itab = buf + (i * 32 + j) * 4;
            int isum = 0;
            for (int k1 = 0; k1 < 2; k1++) {
                const float vy = tab1[i][k1];
                for (int k2 = 0; k2 < 2; k2++) {
                    const float v = vy * tab1[j][k2];
                    itab[k1 * 2 + k2] = orc_sat_s16(lrintf(v * 32768));
                    isum += itab[k1 * 2 + k2];
                }
            }
            if (isum != 32768) {
                const int diff = isum - 32768;
                int Mk1 = 1, Mk2 = 1, mk1 = 1, mk2 = 1;
                for (int k1 = 1; k1 < 3; k1++)
                    for (int k2 = 1; k2 < 3; k2++) {
                        if (itab[k1 * 2 + k2] < itab[mk1 * 2 + mk2]) mk1 = k1, mk2 = k2;
                        else if (itab[k1 * 2 + k2] > itab[Mk1 * 2 + Mk2]) Mk1 = k1, Mk2 = k2;
                    }
                if (diff < 0) itab[Mk1 * 2 + Mk2] = (int16_t)(itab[Mk1 * 2 + Mk2] - diff);
                else itab[mk1 * 2 + mk2] = (int16_t)(itab[mk1 * 2 + mk2] - diff);
            }
        }
    memcpy(T, buf, 1024 * 4 * sizeof(int16_t));
}

void orc_remap_bilinear_u8(const uint8_t* src, int sw, int sh, int sstride, int cn, const float* map1, const float* map2,
                           int w, int h, uint8_t* dst, int dstride)
{
    int16_t T[4096];
    orc_remap_weight_table(T);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            /* RemapInvoker, CV_32FC1 maps: positions to 1/32 px, round-half-even */
            const int fx = (int)lrintf(map1[(size_t)y * w + x] * 32), fy = (int)lrintf(map2[(size_t)y * w + x] * 32);
            const int sx = orc_sat_s16(fx >> 5), sy = orc_sat_s16(fy >> 5);
            const int16_t* wt = T + ((fy & 31) * 32 + (fx & 31)) * 4;
            for (int c = 0; c < cn; c++) {
                int tap[4];
                for (int t = 0; t < 4; t++) {
                    const int xx = sx + (t & 1), yy = sy + (t >> 1);
                    tap[t] = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[(size_t)yy * sstride + xx * cn + c] : 0;
                }
                const int v = (tap[0] * wt[0] + tap[1] * wt[1] + tap[2] * wt[2] + tap[3] * wt[3] + (1 << 14)) >> 15;
                dst[(size_t)y * dstride + x * cn + c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
}


/* A-13  cv::cvtColor(src, dst, CV_BGR2GRAY / CV_RGB2GRAY) on 8UC3 (ORB/src/Tracking.cc:272-295: GrabImageStereo converts mImGray by mbRGB).
 * "parity unpinned" like A-1 .. A-11: restated from OpenCV's published RGB2Gray<uchar>.  Fixed point:
 *   OpenCV 4.x  (color.hpp: gray_shift = 15, RY15 = 9798, GY15 = 19235, BY15 = 3735):  (R*9798 + G*19235 + B*3735 + (1 << 14)) >> 15
 *   OpenCV <= 3.x (yuv_shift = 14, R2Y = 4899, G2Y = 9617, B2Y = 1868):                (R*4899 + G*9617  + B*1868 + (1 << 13)) >> 14
 * rgb: 0 = bytes B,G,R (CV_BGR2GRAY), 1 = bytes R,G,B (CV_RGB2GRAY); cv3: 1 = the <= 3.x coefficients. */
void orc_gray_from_color(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int rgb, int cv3)
{
    const int kR = cv3 ? 4899 : 9798, kG = cv3 ? 9617 : 19235, kB = cv3 ? 1868 : 3735, sh = cv3 ? 14 : 15;
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            const int c0 = s[3 * x], g = s[3 * x + 1], c2 = s[3 * x + 2];
            const int r = rgb ? c0 : c2, b = rgb ? c2 : c0;
            d[x] = (uint8_t)((r * kR + g * kG + b * kB + (1 << (sh - 1))) >> sh);
        }
    }
}
