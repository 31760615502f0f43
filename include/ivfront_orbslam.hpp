// ivfront_orbslam.hpp -- header-only adapters that re-create the reference's C++ class surfaces on top
// of the C-ABI (include/ivfront.h), so Tracking / Frame / LocalMapping keep calling them unchanged.
//
//   ORB_SLAM2::ORBextractor   replaces ORB/include/ORBextractor.h:51-126 + ORB/src/ORBextractor.cc
//   ORB_SLAM2::ORBmatcher     replaces the Hamming core of ORB/include/ORBmatcher.h:37-108
//   ivf::ComputeStereoMatches replaces the body of Frame::ComputeStereoMatches (ORB/src/Frame.cc:758-932)
//
// Needs OpenCV *headers* only for the types in the signatures (cv::Mat, cv::KeyPoint, cv::InputArray);
// no OpenCV function does any of the work.  Link with -livfront.  See INTEGRATION.md for the three-line
// change in the reference's CMakeLists.txt.
#pragma once
#include <opencv2/core/core.hpp>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "ivfront.h"

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // same argument list as ORB/include/ORBextractor.h:57-58; device_id is the only addition (defaulted)
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 bool enableIntrospection = false, int device_id = 0)
        : nlevels_(nlevels), nfeatures_(nfeatures)
    {
        ivf_extractor_params p = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, enableIntrospection ? 1 : 0};
        if (ivf_extractor_create(&p, device_id, &h_) != IVF_OK)
            throw std::runtime_error(std::string("ivf_extractor_create: ") + ivf_last_error());
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        ivf_extractor_get_scale_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                                       mvInvLevelSigma2.data());
        mvImagePyramid.resize(nlevels); mvQualityImagePyramid.resize(nlevels);
    }
    ~ORBextractor() { ivf_extractor_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // ORB/src/ORBextractor.cc:1224-1296.  Empty image -> silent return (:1227); callee allocates descriptors.
    void operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints,
                    cv::OutputArray _descriptors)
    {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        CV_Assert(image.type() == CV_8UC1);
        cv::Mat mask;
        if (!_mask.empty()) { mask = _mask.getMat(); CV_Assert(mask.type() == CV_8UC1 && mask.size() == image.size()); }
        std::vector<ivf_keypoint> kps(nfeatures_);
        cv::Mat desc(nfeatures_, 32, CV_8U);
        int n = 0;
        const int rc = ivf_extract(h_, image.data, image.cols, image.rows, (int)image.step,
                                   mask.empty() ? nullptr : mask.data, mask.empty() ? 0 : (int)mask.step,
                                   kps.data(), desc.data, nfeatures_, &n);
        if (rc != IVF_OK) throw std::runtime_error(std::string("ivf_extract: ") + ivf_last_error());
        _keypoints.clear();
        _keypoints.reserve(n);
        for (int i = 0; i < n; i++)
            _keypoints.push_back(cv::KeyPoint(kps[i].x, kps[i].y, kps[i].size, kps[i].angle, kps[i].response, kps[i].octave));
        if (n == 0) _descriptors.release();
        else desc.rowRange(0, n).copyTo(_descriptors);
        // public data members read by Frame::ComputeStereoMatches when it is NOT replaced (Frame.cc:765,855,867,872)
        for (int l = 0; l < nlevels_; l++) {
            int w = 0, hgt = 0;
            ivf_extractor_pyramid_level(h_, l, nullptr, 0, &w, &hgt);
            mvImagePyramid[l].create(hgt, w, CV_8U);
            ivf_extractor_pyramid_level(h_, l, mvImagePyramid[l].data, (int)mvImagePyramid[l].step, &w, &hgt);
        }
    }

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return ivf_extractor_get_scale_factor(h_); }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    std::vector<cv::Mat> mvImagePyramid;
    std::vector<cv::Mat> mvQualityImagePyramid;

    ivf_extractor* handle() const { return h_; }     // for ivf::ComputeStereoMatches below

protected:
    ivf_extractor* h_ = nullptr;
    int nlevels_, nfeatures_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;     // ORB/src/ORBmatcher.cc:37-39
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    // ORB/src/ORBmatcher.cc:1700-1716
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return ivf_hamming(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }
    // ORB/src/ORBmatcher.cc:1372-1518 on flat projected queries (the adapter inside Tracking projects
    // LastFrame's map points exactly as :1399-1434 does, then calls this)
    int SearchByProjectionFlat(const std::vector<ivf_keypoint>& curKeysUn, const cv::Mat& curDescriptors,
                               const std::vector<float>& curURight, const ivf_bounds& bounds,
                               const std::vector<float>& u, const std::vector<float>& v, const std::vector<float>& ur,
                               const std::vector<float>& radius, const std::vector<int32_t>& minLevel,
                               const std::vector<int32_t>& maxLevel, const std::vector<float>& angle,
                               const cv::Mat& queryDescriptors, const std::vector<uint8_t>& blocks,
                               std::vector<int32_t>& curAssign, int device_id = 0) const
    {
        int nm = 0;
        const int rc = ivf_search_by_projection(curKeysUn.data(), curDescriptors.ptr<uint8_t>(), curURight.data(),
                                                (int)curKeysUn.size(), &bounds, (int)u.size(), u.data(), v.data(), ur.data(),
                                                radius.data(), minLevel.data(), maxLevel.data(), angle.data(),
                                                queryDescriptors.ptr<uint8_t>(), nullptr, blocks.empty() ? nullptr : blocks.data(),
                                                mbCheckOrientation ? 1 : 0, curAssign.data(), &nm, device_id);
        if (rc != IVF_OK) throw std::runtime_error(std::string("ivf_search_by_projection: ") + ivf_last_error());
        return nm;
    }
    // ORB/src/ORBmatcher.cc:410-519, same arguments on the frames' public members: mvKeysUn / mDescriptors of F1 and F2,
    // F2's image bounds (Frame::mnMinX ...).  Called from Tracking::MonocularInitialization (ORB/src/Tracking.cc:1036).
    int SearchForInitialization(const std::vector<cv::KeyPoint>& keysUn1, const cv::Mat& descriptors1,
                                const std::vector<cv::KeyPoint>& keysUn2, const cv::Mat& descriptors2, const ivf_bounds& bounds2,
                                std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10,
                                int device_id = 0) const
    {
        auto conv = [](const std::vector<cv::KeyPoint>& in) {
            std::vector<ivf_keypoint> out(in.size());
            for (size_t i = 0; i < in.size(); i++) out[i] = {in[i].pt.x, in[i].pt.y, in[i].size, in[i].angle, in[i].response, in[i].octave};
            return out;
        };
        const std::vector<ivf_keypoint> k1 = conv(keysUn1), k2 = conv(keysUn2);
        std::vector<float> prev(2 * k1.size());
        for (size_t i = 0; i < k1.size(); i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
        std::vector<int32_t> m12(k1.size(), -1);
        int nm = 0;
        const int rc = ivf_search_for_initialization(k1.data(), descriptors1.ptr<uint8_t>(), (int)k1.size(), k2.data(),
                                                     descriptors2.ptr<uint8_t>(), (int)k2.size(), &bounds2, prev.data(), windowSize,
                                                     mfNNratio, mbCheckOrientation ? 1 : 0, m12.data(), &nm, device_id);
        if (rc != IVF_OK) throw std::runtime_error(std::string("ivf_search_for_initialization: ") + ivf_last_error());
        vnMatches12.assign(m12.begin(), m12.end());
        for (size_t i = 0; i < k1.size(); i++) { vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1]; }
        return nm;
    }

protected:
    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM2

namespace ivf {
// Body replacement for Frame::ComputeStereoMatches (ORB/src/Frame.cc:758-932): fills mvuRight / mvDepth.
inline void ComputeStereoMatches(ORB_SLAM2::ORBextractor* left, ORB_SLAM2::ORBextractor* right,
                                 const std::vector<cv::KeyPoint>& mvKeys, const cv::Mat& mDescriptors,
                                 const std::vector<cv::KeyPoint>& mvKeysRight, const cv::Mat& mDescriptorsRight,
                                 float mbf, float mb, std::vector<float>& mvuRight, std::vector<float>& mvDepth)
{
    auto conv = [](const std::vector<cv::KeyPoint>& in) {
        std::vector<ivf_keypoint> out(in.size());
        for (size_t i = 0; i < in.size(); i++) out[i] = {in[i].pt.x, in[i].pt.y, in[i].size, in[i].angle, in[i].response, in[i].octave};
        return out;
    };
    const std::vector<ivf_keypoint> kl = conv(mvKeys), kr = conv(mvKeysRight);
    mvuRight.assign(kl.size(), -1.0f);
    mvDepth.assign(kl.size(), -1.0f);
    const int rc = ivf_stereo_match(left->handle(), right->handle(), kl.data(), (int)kl.size(), mDescriptors.ptr<uint8_t>(),
                                    kr.data(), (int)kr.size(), mDescriptorsRight.ptr<uint8_t>(), mbf, mb,
                                    mvuRight.data(), mvDepth.data());
    if (rc != IVF_OK) throw std::runtime_error(std::string("ivf_stereo_match: ") + ivf_last_error());
}
// Core of MapPoint::ComputeDistinctiveDescriptors (ORB/src/MapPoint.cc:247-312): index of the observed descriptor to keep.
inline int DistinctiveDescriptorIndex(const std::vector<cv::Mat>& vDescriptors, int device_id = 0)
{
    std::vector<uint8_t> flat(vDescriptors.size() * 32);
    for (size_t i = 0; i < vDescriptors.size(); i++) std::memcpy(&flat[i * 32], vDescriptors[i].ptr<uint8_t>(), 32);
    int best = 0;
    const int rc = ivf_distinctive_descriptor(flat.data(), (int)vDescriptors.size(), &best, nullptr, device_id);
    if (rc != IVF_OK) throw std::runtime_error(std::string("ivf_distinctive_descriptor: ") + ivf_last_error());
    return best;
}
}  // namespace ivf
