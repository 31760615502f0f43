// TEST-ONLY driver: RUNS the adapter's ORB_SLAM2::ORBextractor (include/ivfront_orbslam.hpp; the reference's class surface,
// ORB/include/ORBextractor.h:57-92) the way Frame's stereo constructor does -- left and right operator() on two std::threads, the
// LEFT cost map handed to both as `mask` (ORB/src/Frame.cc:116-124) -- then ivf::ComputeStereoMatches (Frame.cc:758-932), against the
// mock cv types of tests/cv_mock.  Everything it produces is dumped as a flat file; tests/test_gpu_adapter.py compares it byte
// for byte with the ctypes path and the oracle.
#include "ivfront_orbslam.hpp"
#include <cstdio>
#include <thread>

static std::vector<uint8_t> g_out;
template <class T> static void put(const T& v) { const uint8_t* p = (const uint8_t*)&v; g_out.insert(g_out.end(), p, p + sizeof(T)); }
static void put_bytes(const void* p, size_t n) { put((int32_t)n); g_out.insert(g_out.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
static void put_kps(const std::vector<cv::KeyPoint>& k)
{
    put((int32_t)k.size());
    for (const cv::KeyPoint& p : k) { put(p.pt.x); put(p.pt.y); put(p.size); put(p.angle); put(p.response); put((int32_t)p.octave); }
}
static void put_mat(const cv::Mat& m)
{
    put((int32_t)m.rows); put((int32_t)m.cols);
    for (int y = 0; y < m.rows; y++) g_out.insert(g_out.end(), m.data + y * m.step, m.data + y * m.step + m.cols);
}
static cv::Mat read_img(FILE* f, int w, int h)
{
    cv::Mat m(h, w, CV_8U);
    if (fread(m.data, 1, (size_t)w * h, f) != (size_t)w * h) throw std::runtime_error("short scenario file");
    return m;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: extractor_driver scenario.bin result.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    try {
        int32_t hdr[6];                                       // w, h, nfeatures, iniTh, minTh, nlevels
        float fl[3];                                          // scaleFactor, bf, b
        if (fread(hdr, 4, 6, f) != 6 || fread(fl, 4, 3, f) != 3) throw std::runtime_error("short header");
        const int w = hdr[0], h = hdr[1], N = hdr[2];
        cv::Mat imL = read_img(f, w, h), imR = read_img(f, w, h), cost = read_img(f, w, h);
        fclose(f);
        using ORB_SLAM2::ORBextractor;
        // Tracking.cc:174-191: the left extractor with the YAML's enableIntrospection, the right one without
        ORBextractor left(N, fl[0], hdr[5], hdr[3], hdr[4], true), right(N, fl[0], hdr[5], hdr[3], hdr[4]);
        std::vector<cv::KeyPoint> kL, kR; cv::Mat dL, dR;
        // 1. Frame.cc:116-124: two threads, both given the left cost map
        std::thread tl([&] { left(imL, cost, kL, dL); });
        std::thread tr([&] { right(imR, cost, kR, dR); });
        tl.join(); tr.join();
        put_kps(kL); put_mat(dL); put_kps(kR); put_mat(dR);
        put((int32_t)left.GetLevels());
        for (int l = 0; l < left.GetLevels(); l++) { put_mat(left.mvImagePyramid[l]); put_mat(left.mvQualityImagePyramid[l]); put_mat(right.mvImagePyramid[l]); }
        { std::vector<float> s = left.GetScaleFactors(), is = left.GetInverseScaleFactors(), g = left.GetScaleSigmaSquares(), ig = left.GetInverseScaleSigmaSquares();
          put_bytes(s.data(), s.size() * 4); put_bytes(is.data(), is.size() * 4); put_bytes(g.data(), g.size() * 4); put_bytes(ig.data(), ig.size() * 4);
          put(left.GetScaleFactor()); }
        // 2. ivf::ComputeStereoMatches on what the two handles hold (Frame.cc:758-932)
        std::vector<float> uR, depth;
        ivf::ComputeStereoMatches(&left, &right, kL, dL, kR, dR, fl[1], fl[2], uR, depth);
        put_bytes(uR.data(), uR.size() * 4); put_bytes(depth.data(), depth.size() * 4);
        // 3. the same left extractor WITHOUT a mask (ORBextractor.cc:1231-1238: no quality pyramid, plain FAST responses)
        std::vector<cv::KeyPoint> k2; cv::Mat d2;
        left(imL, cv::Mat(), k2, d2);
        put_kps(k2); put_mat(d2);
        // 4. mbCopyPyramids = false: the public pyramids keep the previous call's content, single levels on demand
        left.mbCopyPyramids = false;
        const cv::Mat before = left.mvImagePyramid[2].clone();
        std::vector<cv::KeyPoint> k3; cv::Mat d3;
        left(imR, cv::Mat(), k3, d3);
        put_kps(k3);
        put((int32_t)(memcmp(before.data, left.mvImagePyramid[2].data, (size_t)before.rows * before.step) == 0));
        cv::Mat lvl; left.CopyPyramidLevel(2, false, lvl);
        put_mat(lvl);
        // 5. empty image: silent return, outputs untouched (ORBextractor.cc:1227-1228)
        std::vector<cv::KeyPoint> k4(3); cv::Mat d4(2, 32, CV_8U);
        left(cv::Mat(), cv::Mat(), k4, d4);
        put((int32_t)k4.size()); put((int32_t)d4.rows);
    } catch (const std::exception& e) { fprintf(stderr, "extractor_driver: %s\n", e.what()); return 1; }
    FILE* o = fopen(argv[2], "wb");
    if (!o || fwrite(g_out.data(), 1, g_out.size(), o) != g_out.size()) return 2;
    fclose(o);
    return 0;
}
