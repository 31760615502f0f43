// stl_pin.cpp -- TEST INFRASTRUCTURE: pins oracle/ivf_oracle.c's restatement of
// cv::KeyPointsFilter::retainBest (std::nth_element + std::partition + resize, OpenCV 4.x
// features2d/keypoint.cpp as called from ORB/src/ORBextractor.cc:1146-1148,1164-1165) against the
// REAL libstdc++ of this image.  Exposes the real-STL version through a C symbol so the Python test
// can compare both on tie-heavy random inputs.
#include <algorithm>
#include <vector>
#include <cstdint>
#include "ivf_oracle.h"

namespace {
struct RespGreater { bool operator()(const orc_keypoint& a, const orc_keypoint& b) const { return a.response > b.response; } };
struct RespGE { float v; bool operator()(const orc_keypoint& k) const { return k.response >= v; } };
}

extern "C" int stl_retain_best(orc_keypoint* v, int n, int n_points)
{
    std::vector<orc_keypoint> k(v, v + n);
    if (n_points >= 0 && k.size() > (size_t)n_points) {
        if (n_points == 0) { k.clear(); }
        else {
            std::nth_element(k.begin(), k.begin() + n_points - 1, k.end(), RespGreater());
            float amb = k[n_points - 1].response;
            auto new_end = std::partition(k.begin() + n_points, k.end(), RespGE{amb});
            k.resize(new_end - k.begin());
        }
    }
    if ((int)k.size() > n_points) k.resize(n_points);      // the reference's own resize
    std::copy(k.begin(), k.end(), v);
    return (int)k.size();
}

extern "C" void stl_nth_element(orc_keypoint* v, int n, int nth)
{
    std::nth_element(v, v + nth, v + n, RespGreater());
}

// glibc pin: counts floats u in [lo, hi) (bit patterns, step `stride`) whose restated
// orc_cosf/orc_sinf differ from this image's libm cosf/sinf (what the reference calls at
// ORB/src/ORBextractor.cc:113-114).
#include <cmath>
#include <cstring>
extern "C" long glibc_trig_mismatches(uint32_t lo, uint32_t hi, uint32_t stride)
{
    long bad = 0;
    for (uint64_t u = lo; u < hi; u += stride) {
        uint32_t b = (uint32_t)u; float x; std::memcpy(&x, &b, 4);
        if (orc_cosf(x) != cosf(x) || orc_sinf(x) != sinf(x)) bad++;
    }
    return bad;
}

// how many floats in [lo, hi) (bit patterns, step `stride`) where orc_logf differs from this image's libm logf (what
// MapPoint::PredictScale calls, ORB/src/MapPoint.cc:398,415)
extern "C" long glibc_logf_mismatches(uint32_t lo, uint32_t hi, uint32_t stride)
{
    long bad = 0;
    for (uint64_t u = lo; u < hi; u += stride) {
        uint32_t b = (uint32_t)u; float x; std::memcpy(&x, &b, 4);
        if (orc_logf(x) != logf(x)) bad++;
    }
    return bad;
}
