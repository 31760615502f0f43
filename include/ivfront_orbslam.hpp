// ivfront_orbslam.hpp -- header-only adapters that re-create the reference's C++ class surfaces on top
// of the C-ABI (include/ivfront.h), so Tracking / Frame / LocalMapping keep calling them unchanged.
//
//   ORB_SLAM2::ORBextractor   replaces ORB/include/ORBextractor.h:51-126 + ORB/src/ORBextractor.cc
//   ORB_SLAM2::ORBmatcherT    replaces ORB/include/ORBmatcher.h:37-108 + ORB/src/ORBmatcher.cc (all 11 searches, reference signatures)
//   ivf::ComputeStereoMatches replaces the body of Frame::ComputeStereoMatches (ORB/src/Frame.cc:758-932)
//
// Needs OpenCV *headers* only for the types in the signatures (cv::Mat, cv::KeyPoint, cv::InputArray);
// no OpenCV function does any of the work.  Link with -livfront.  See INTEGRATION.md for the three-line
// change in the reference's CMakeLists.txt.
#pragma once
#include <opencv2/core/core.hpp>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>
#include <stdexcept>
#include <utility>
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
        : nlevels_(nlevels), nfeatures_(nfeatures), introspection_(enableIntrospection)
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
        // public data members (ORBextractor.h:91-92).  mvImagePyramid is read by Frame::ComputeStereoMatches when that is NOT
        // replaced by ivf::ComputeStereoMatches (Frame.cc:765,855,867,872); mvQualityImagePyramid is filled like the reference does
        // (only with a mask and enableIntrospection, :1231-1238).  The copies are 8 device-to-host transfers per call: a caller that
        // has replaced every reader sets mbCopyPyramids = false and fetches single levels with CopyPyramidLevel when it needs one.
        if (mbCopyPyramids)
            for (int l = 0; l < nlevels_; l++) {
                CopyPyramidLevel(l, false, mvImagePyramid[l]);
                if (!mask.empty() && introspection_) CopyPyramidLevel(l, true, mvQualityImagePyramid[l]);
            }
    }

    // one level of the last call's image (quality = false) or cost-map (true) pyramid, device -> host
    void CopyPyramidLevel(int level, bool quality, cv::Mat& dst)
    {
        int w = 0, hgt = 0;
        int rc = quality ? ivf_extractor_quality_level(h_, level, nullptr, 0, &w, &hgt) : ivf_extractor_pyramid_level(h_, level, nullptr, 0, &w, &hgt);
        if (rc == IVF_OK) {
            dst.create(hgt, w, CV_8U);
            rc = quality ? ivf_extractor_quality_level(h_, level, dst.data, (int)dst.step, &w, &hgt)
                         : ivf_extractor_pyramid_level(h_, level, dst.data, (int)dst.step, &w, &hgt);
        }
        if (rc != IVF_OK) throw std::runtime_error(std::string("pyramid level: ") + ivf_last_error());
    }

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return ivf_extractor_get_scale_factor(h_); }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    std::vector<cv::Mat> mvImagePyramid;
    std::vector<cv::Mat> mvQualityImagePyramid;
    bool mbCopyPyramids = true;                       // addition: see operator()

    ivf_extractor* handle() const { return h_; }     // for ivf::ComputeStereoMatches below

protected:
    ivf_extractor* h_ = nullptr;
    int nlevels_, nfeatures_;
    bool introspection_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM2

// ---- small-matrix arithmetic of the matcher's projection loops ---------------------------------------------------------
// The reference writes these steps with cv::Mat expressions (Rcw*x3Dw+tcw, -Rcw.t()*tcw, cv::norm(PO), PO.dot(Pn),
// sRcw/scw ...) whose arithmetic lives in un-vendored OpenCV.  Frozen here ("parity unpinned" like the other
// OpenCV primitives) as OpenCV 4.x's plain C++ paths evaluate them for CV_32F operands:
//   A*B (+C)   cv::gemm, GEMMSingleMul<float,double>: every output = (float)(sum_k (double)a*(double)b [+ (double)c])
//   -A.t()*b   one gemm with alpha = -1 and GEMM_1_T: (float)(-(sum_k (double)a_kj*(double)b_k))
//   a.dot(b), cv::norm(a)   double accumulation of (double)a*(double)b; norm = sqrt of it; narrowed where the source assigns to float
//   s*A, A/s   cvtScale f32 -> f32: a * (float)alpha with alpha = s resp. 1.0/s (double, narrowed once)
// Everything else (u = fx*xc*invzc + cx ...) is the float expression the reference's source spells out; compile the
// translation unit that includes this header with -ffp-contract=off.
namespace ivf { namespace mat {
struct V3 { float x, y, z; };
struct M3 { float m[9]; };       // row-major
template <class Mat> inline V3 vec3(const Mat& v) { return V3{v.template at<float>(0, 0), v.template at<float>(1, 0), v.template at<float>(2, 0)}; }
template <class Mat> inline M3 rot_of(const Mat& T)                  // T.rowRange(0,3).colRange(0,3)
{ M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[3 * i + j] = T.template at<float>(i, j); return r; }
template <class Mat> inline V3 trans_of(const Mat& T) { return V3{T.template at<float>(0, 3), T.template at<float>(1, 3), T.template at<float>(2, 3)}; }
inline V3 mul_add(const M3& R, const V3& p, const V3& t)            // R*p + t
{
    const double X = p.x, Y = p.y, Z = p.z;
    return V3{(float)((double)R.m[0] * X + (double)R.m[1] * Y + (double)R.m[2] * Z + (double)t.x),
              (float)((double)R.m[3] * X + (double)R.m[4] * Y + (double)R.m[5] * Z + (double)t.y),
              (float)((double)R.m[6] * X + (double)R.m[7] * Y + (double)R.m[8] * Z + (double)t.z)};
}
inline V3 neg_rt_mul(const M3& R, const V3& t)                       // -R.t()*t
{
    const double X = t.x, Y = t.y, Z = t.z;
    return V3{(float)(-((double)R.m[0] * X + (double)R.m[3] * Y + (double)R.m[6] * Z)),
              (float)(-((double)R.m[1] * X + (double)R.m[4] * Y + (double)R.m[7] * Z)),
              (float)(-((double)R.m[2] * X + (double)R.m[5] * Y + (double)R.m[8] * Z))};
}
inline V3 neg_mul(const M3& R, const V3& t)                          // -R*t
{
    const double X = t.x, Y = t.y, Z = t.z;
    return V3{(float)(-((double)R.m[0] * X + (double)R.m[1] * Y + (double)R.m[2] * Z)),
              (float)(-((double)R.m[3] * X + (double)R.m[4] * Y + (double)R.m[5] * Z)),
              (float)(-((double)R.m[6] * X + (double)R.m[7] * Y + (double)R.m[8] * Z))};
}
inline V3 sub(const V3& a, const V3& b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(const V3& a, const V3& b) { return (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z; }
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
inline M3 scaled(const M3& R, double alpha, bool transposed = false)
{
    M3 o; const float a = (float)alpha;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.m[3 * i + j] = (transposed ? R.m[3 * j + i] : R.m[3 * i + j]) * a;
    return o;
}
inline V3 scaled(const V3& v, double alpha) { const float a = (float)alpha; return V3{v.x * a, v.y * a, v.z * a}; }
}}  // namespace ivf::mat

namespace ORB_SLAM2 {

// ORB_SLAM2::ORBmatcher with the reference's own signatures (ORB/include/ORBmatcher.h:44-89), written once against the
// MEMBER NAMES of the caller's Frame / KeyFrame / MapPoint (ORB/include/Frame.h, KeyFrame.h, MapPoint.h): the class is a
// template over those three types, so this header needs none of the reference's headers, and a method is instantiated only
// where a call site uses it.  The projection / bookkeeping loops run here, on the host, in the reference's order; every
// window search + Hamming distance runs on the device behind the C-ABI (ivf_search_* in ivfront.h).
//   typedef ORB_SLAM2::ORBmatcherT<Frame, KeyFrame, MapPoint> ORBmatcher;     // what ORBmatcher.h becomes
template <class Frame, class KeyFrame, class MapPoint>
class ORBmatcherT {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;     // ORB/src/ORBmatcher.cc:37-39
    ORBmatcherT(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // the gflag FLAGS_ivslam_propagate_keyptqual (ORB/src/ORBmatcher.cc:130,1513,1647; default false, MapPoint.cc:26) and the
    // GPU the searches run on: process-wide settings of the adapter
    static bool& PropagateKeyptQual() { static bool v = false; return v; }
    static int& DeviceId() { static int v = 0; return v; }

    // ORB/src/ORBmatcher.cc:1700-1716
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return ivf_hamming(a.template ptr<uint8_t>(), b.template ptr<uint8_t>()); }

    // ---- ORB/src/ORBmatcher.cc:45-135 (Tracking::SearchLocalPoints) ----
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3)
    {
        const bool bFactor = th != 1.0;
        Queries q; std::vector<int> src;                             // query -> index into vpMapPoints
        for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
            MapPoint* pMP = vpMapPoints[iMP];
            if (!pMP->mbTrackInView || pMP->isBad()) continue;
            const int nPredictedLevel = pMP->mnTrackScaleLevel;
            float r = RadiusByViewingCos(pMP->mTrackViewCos);
            if (bFactor) r *= th;
            q.add(pMP->mTrackProjX, pMP->mTrackProjY, pMP->mTrackProjXR, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel, 0, 0.f,
                  pMP->GetDescriptor(), pMP->Observations() > 0);
            src.push_back((int)iMP);
        }
        std::vector<int32_t> assign = occupancy(F.mvpMapPoints, false);
        int nm = 0;
        if (q.n() > 0 && F.N > 0) {
            const FrameArrays fa(F.mvKeysUn, F.mDescriptors);
            const ivf_bounds bd = bounds_of(F);
            check(ivf_search_map_points(fa.kps.data(), fa.desc, F.mvuRight.data(), F.N, &bd, q.n(), q.u.data(), q.v.data(), q.ur.data(),
                                        q.radius.data(), q.level.data(), q.desc.data(), nullptr, q.blocks.data(), mfNNratio,
                                        assign.data(), &nm, DeviceId()), "ivf_search_map_points");
            for (int i = 0; i < F.N; i++) if (assign[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[src[assign[i]]];
        }
        if (PropagateKeyptQual()) UpdateQualityScores(F);
        return nm;
    }

    // ---- ORB/src/ORBmatcher.cc:1372-1518 (Tracking::TrackWithMotionModel) ----
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
    {
        using namespace ivf::mat;
        const M3 Rcw = rot_of(CurrentFrame.mTcw); const V3 tcw = trans_of(CurrentFrame.mTcw);
        const V3 twc = neg_rt_mul(Rcw, tcw);
        const M3 Rlw = rot_of(LastFrame.mTcw); const V3 tlw = trans_of(LastFrame.mTcw);
        const V3 tlc = mul_add(Rlw, twc, tlw);
        const bool bForward = tlc.z > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc.z > CurrentFrame.mb && !bMono;
        Queries q; std::vector<int> src;                             // query -> keypoint index in LastFrame
        for (int i = 0; i < LastFrame.N; i++) {
            MapPoint* pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            const V3 x3Dc = mul_add(Rcw, vec3(pMP->GetWorldPos()), tcw);
            const float xc = x3Dc.x, yc = x3Dc.y;
            const float invzc = 1.0 / x3Dc.z;
            if (invzc < 0) continue;
            const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
            const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
            if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
            if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
            const int nLastOctave = LastFrame.mvKeys[i].octave;
            const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
            int minL, maxL;                                          // GetFeaturesInArea level arguments (:1429-1434)
            if (bForward) { minL = nLastOctave; maxL = -1; }
            else if (bBackward) { minL = 0; maxL = nLastOctave; }
            else { minL = nLastOctave - 1; maxL = nLastOctave + 1; }
            const float ur = u - CurrentFrame.mbf * invzc;
            q.add(u, v, ur, radius, minL, maxL, LastFrame.mvKeysUn[i].angle, pMP->GetDescriptor(), pMP->Observations() > 0);
            src.push_back(i);
        }
        std::vector<int32_t> assign = occupancy(CurrentFrame.mvpMapPoints, false);
        int nm = 0;
        if (q.n() > 0 && CurrentFrame.N > 0) {
            const FrameArrays fa(CurrentFrame.mvKeysUn, CurrentFrame.mDescriptors);
            const ivf_bounds bd = bounds_of(CurrentFrame);
            std::vector<uint8_t> removed(CurrentFrame.N, 0);
            check(ivf_search_by_projection_ex(fa.kps.data(), fa.desc, CurrentFrame.mvuRight.data(), CurrentFrame.N, &bd, q.n(), q.u.data(),
                                              q.v.data(), q.ur.data(), q.radius.data(), q.level.data(), q.maxLevel.data(), q.angle.data(),
                                              q.desc.data(), nullptr, q.blocks.data(), mbCheckOrientation ? 1 : 0, assign.data(),
                                              removed.data(), &nm, DeviceId()), "ivf_search_by_projection");
            for (int i = 0; i < CurrentFrame.N; i++) {
                if (assign[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[src[assign[i]]];
                else if (removed[i]) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint*>(NULL);      // rotation filter (:1504)
            }
        }
        if (PropagateKeyptQual()) UpdateQualityScores(CurrentFrame);
        return nm;
    }

    // ---- ORB/src/ORBmatcher.cc:1520-1652 (Tracking::Relocalization) ----
    template <class MapPointSet>
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const MapPointSet& sAlreadyFound, const float th, const int ORBdist)
    {
        using namespace ivf::mat;
        const M3 Rcw = rot_of(CurrentFrame.mTcw); const V3 tcw = trans_of(CurrentFrame.mTcw);
        const V3 Ow = neg_rt_mul(Rcw, tcw);
        const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
        Queries q; std::vector<int> src;
        for (size_t i = 0; i < vpMPs.size(); i++) {
            MapPoint* pMP = vpMPs[i];
            if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
            const V3 x3Dw = vec3(pMP->GetWorldPos());
            const V3 x3Dc = mul_add(Rcw, x3Dw, tcw);
            const float xc = x3Dc.x, yc = x3Dc.y;
            const float invzc = 1.0 / x3Dc.z;
            const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
            const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
            if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
            if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
            const float dist3D = (float)norm(sub(x3Dw, Ow));
            const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
            const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
            q.add(u, v, 0.f, radius, nPredictedLevel, 0, pKF->mvKeysUn[i].angle, pMP->GetDescriptor(), true);
            src.push_back((int)i);
        }
        std::vector<int32_t> assign = occupancy(CurrentFrame.mvpMapPoints, true);      // any occupant blocks (:1591-1592)
        int nm = 0;
        if (q.n() > 0 && CurrentFrame.N > 0) {
            const FrameArrays fa(CurrentFrame.mvKeysUn, CurrentFrame.mDescriptors);
            const ivf_bounds bd = bounds_of(CurrentFrame);
            check(ivf_search_by_projection_reloc(fa.kps.data(), fa.desc, CurrentFrame.N, &bd, q.n(), q.u.data(), q.v.data(), q.radius.data(),
                                                 q.level.data(), q.angle.data(), q.desc.data(), nullptr, ORBdist, mbCheckOrientation ? 1 : 0,
                                                 assign.data(), &nm, DeviceId()), "ivf_search_by_projection_reloc");
            for (int i = 0; i < CurrentFrame.N; i++) if (assign[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[src[assign[i]]];
        }
        if (PropagateKeyptQual()) UpdateQualityScores(CurrentFrame);
        return nm;
    }

    // ---- ORB/src/ORBmatcher.cc:296-404 (LoopClosing::ComputeSim3) ----
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th)
    {
        using namespace ivf::mat;
        M3 Rcw; V3 tcw, Ow;
        decompose_sim3(Scw, Rcw, tcw, Ow);
        std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
        Queries q; std::vector<int> src;
        for (int iMP = 0, iend = (int)vpPoints.size(); iMP < iend; iMP++) {
            MapPoint* pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            float u, v, dist; V3 p3Dc;
            if (!project_kf(pKF, Rcw, tcw, Ow, pMP, true, u, v, dist, p3Dc)) continue;
            const int nPredictedLevel = pMP->PredictScale(dist, pKF);
            q.add(u, v, 0.f, th * pKF->mvScaleFactors[nPredictedLevel], nPredictedLevel, 0, 0.f, pMP->GetDescriptor(), true);
            src.push_back(iMP);
        }
        std::vector<int32_t> matched = occupancy(vpMatched, true);
        int nm = 0;
        if (q.n() > 0 && pKF->N > 0) {
            const FrameArrays fa(pKF->mvKeysUn, pKF->mDescriptors);
            const ivf_bounds bd = bounds_of(*pKF);
            check(ivf_search_keyframe_points(fa.kps.data(), fa.desc, pKF->N, &bd, q.n(), q.u.data(), q.v.data(), q.radius.data(),
                                             q.level.data(), q.desc.data(), nullptr, matched.data(), &nm, DeviceId()),
                  "ivf_search_keyframe_points");
            for (int i = 0; i < pKF->N; i++) if (matched[i] >= 0) vpMatched[i] = vpPoints[src[matched[i]]];
        }
        return nm;
    }

    // ---- ORB/src/ORBmatcher.cc:165-294 (Tracking::TrackReferenceKeyFrame / Relocalization) ----
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches)
    {
        const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
        const FrameArrays kf(pKF->mvKeysUn, pKF->mDescriptors), fr(F.mvKeys, F.mDescriptors);
        const std::vector<uint8_t> has = good_points(vpMapPointsKF);
        const Csr a(pKF->mFeatVec), b(F.mFeatVec);
        std::vector<int32_t> fmatch(std::max(F.N, 1), -1);
        int nm = 0;
        if (F.N > 0 && !vpMapPointsKF.empty())
            check(ivf_search_by_bow(kf.kps.data(), kf.desc, has.data(), (int)vpMapPointsKF.size(), a.node.data(), a.start.data(), a.idx.data(),
                                    a.n(), fr.kps.data(), fr.desc, F.N, b.node.data(), b.start.data(), b.idx.data(), b.n(), mfNNratio,
                                    mbCheckOrientation ? 1 : 0, fmatch.data(), &nm, DeviceId()), "ivf_search_by_bow");
        for (int i = 0; i < F.N; i++) if (fmatch[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[fmatch[i]];
        return nm;
    }

    // ---- ORB/src/ORBmatcher.cc:528-661 (LoopClosing::ComputeSim3) ----
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12)
    {
        const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
        const FrameArrays k1(pKF1->mvKeysUn, pKF1->mDescriptors), k2(pKF2->mvKeysUn, pKF2->mDescriptors);
        const std::vector<uint8_t> has1 = good_points(vpMapPoints1), has2 = good_points(vpMapPoints2);
        const Csr a(pKF1->mFeatVec), b(pKF2->mFeatVec);
        std::vector<int32_t> m12(std::max<size_t>(vpMapPoints1.size(), 1), -1);
        int nm = 0;
        if (!vpMapPoints1.empty() && !vpMapPoints2.empty())
            check(ivf_search_by_bow_keyframes(k1.kps.data(), k1.desc, has1.data(), (int)vpMapPoints1.size(), a.node.data(), a.start.data(),
                                              a.idx.data(), a.n(), k2.kps.data(), k2.desc, has2.data(), (int)vpMapPoints2.size(), b.node.data(),
                                              b.start.data(), b.idx.data(), b.n(), mfNNratio, mbCheckOrientation ? 1 : 0, m12.data(), &nm,
                                              DeviceId()), "ivf_search_by_bow_keyframes");
        for (size_t i = 0; i < vpMapPoints1.size(); i++) if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
        return nm;
    }

    // ---- ORB/src/ORBmatcher.cc:410-519 (Tracking::MonocularInitialization) ----
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10)
    {
        const FrameArrays k1(F1.mvKeysUn, F1.mDescriptors), k2(F2.mvKeysUn, F2.mDescriptors);
        const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
        std::vector<float> prev(2 * (size_t)std::max(n1, 1));
        for (int i = 0; i < n1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
        std::vector<int32_t> m12(std::max(n1, 1), -1);
        const ivf_bounds bd = bounds_of(F2);
        int nm = 0;
        if (n1 > 0 && n2 > 0)
            check(ivf_search_for_initialization(k1.kps.data(), k1.desc, n1, k2.kps.data(), k2.desc, n2, &bd, prev.data(), windowSize, mfNNratio,
                                                mbCheckOrientation ? 1 : 0, m12.data(), &nm, DeviceId()), "ivf_search_for_initialization");
        vnMatches12.assign(m12.begin(), m12.begin() + n1);
        for (int i = 0; i < n1; i++) { vbPrevMatched[i].x = prev[2 * i]; vbPrevMatched[i].y = prev[2 * i + 1]; }
        return nm;
    }

    // ---- ORB/src/ORBmatcher.cc:663-829 (LocalMapping::CreateNewMapPoints) ----
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo)
    {
        using namespace ivf::mat;
        // epipole of camera 1 in image 2 (:670-676)
        const V3 C2 = mul_add(rot_of(pKF2->GetRotation()), vec3(pKF1->GetCameraCenter()), vec3(pKF2->GetTranslation()));
        const float invz = 1.0f / C2.z;
        const float ex = pKF2->fx * C2.x * invz + pKF2->cx, ey = pKF2->fy * C2.y * invz + pKF2->cy;
        const FrameArrays k1(pKF1->mvKeysUn, pKF1->mDescriptors), k2(pKF2->mvKeysUn, pKF2->mDescriptors);
        std::vector<uint8_t> has1(std::max(pKF1->N, 1)), st1(std::max(pKF1->N, 1)), has2(std::max(pKF2->N, 1)), st2(std::max(pKF2->N, 1));
        for (int i = 0; i < pKF1->N; i++) { has1[i] = pKF1->GetMapPoint(i) ? 1 : 0; st1[i] = pKF1->mvuRight[i] >= 0 ? 1 : 0; }
        for (int i = 0; i < pKF2->N; i++) { has2[i] = pKF2->GetMapPoint(i) ? 1 : 0; st2[i] = pKF2->mvuRight[i] >= 0 ? 1 : 0; }
        const Csr a(pKF1->mFeatVec), b(pKF2->mFeatVec);
        float f12[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) f12[3 * i + j] = F12.template at<float>(i, j);
        std::vector<int32_t> m12(std::max(pKF1->N, 1), -1);
        int nm = 0;
        if (pKF1->N > 0 && pKF2->N > 0)
            check(ivf_search_for_triangulation(k1.kps.data(), k1.desc, has1.data(), st1.data(), pKF1->N, a.node.data(), a.start.data(), a.idx.data(),
                                               a.n(), k2.kps.data(), k2.desc, has2.data(), st2.data(), pKF2->N, b.node.data(), b.start.data(),
                                               b.idx.data(), b.n(), f12, ex, ey, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(),
                                               (int)pKF2->mvScaleFactors.size(), bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, m12.data(), &nm,
                                               DeviceId()), "ivf_search_for_triangulation");
        vMatchedPairs.clear();
        vMatchedPairs.reserve(std::max(nm, 0));
        for (int i = 0; i < pKF1->N; i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return nm;
    }

    // ---- ORB/src/ORBmatcher.cc:1145-1370 (LoopClosing::ComputeSim3) ----
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                     const cv::Mat& t12, const float th)
    {
        using namespace ivf::mat;
        const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;       // KF1's intrinsics in BOTH directions (:1148-1151)
        const M3 R1w = rot_of(pKF1->GetRotation()); const V3 t1w = vec3(pKF1->GetTranslation());
        const M3 R2w = rot_of(pKF2->GetRotation()); const V3 t2w = vec3(pKF2->GetTranslation());
        const M3 r12 = rot_of(R12); const V3 T12 = vec3(t12);
        const M3 sR12 = scaled(r12, (double)s12), sR21 = scaled(r12, 1.0 / s12, true);
        const V3 t21 = neg_mul(sR21, T12);
        const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
        std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
        for (int i = 0; i < N1; i++) {
            MapPoint* pMP = vpMatches12[i];
            if (!pMP) continue;
            vbAlreadyMatched1[i] = true;
            const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
        // one query slot per keypoint of the source keyframe (valid = its map point is projected)
        auto side = [&](const std::vector<MapPoint*>& pts, const std::vector<bool>& done, const M3& Ra, const V3& ta, const M3& Rb, const V3& tb,
                        KeyFrame* target, Queries& q) {
            for (size_t i = 0; i < pts.size(); i++) {
                MapPoint* pMP = pts[i];
                bool ok = pMP && !done[i] && !pMP->isBad();
                float u = 0, v = 0, radius = 0; int level = 0;
                if (ok) {
                    const V3 pa = mul_add(Ra, vec3(pMP->GetWorldPos()), ta);
                    const V3 pb = mul_add(Rb, pa, tb);
                    ok = !(pb.z < 0.0);
                    if (ok) {
                        const float invz = 1.0 / pb.z;
                        const float x = pb.x * invz, y = pb.y * invz;
                        u = fx * x + cx; v = fy * y + cy;
                        ok = target->IsInImage(u, v);
                        if (ok) {
                            const float dist3D = (float)norm(pb);
                            ok = !(dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance());
                            if (ok) { level = pMP->PredictScale(dist3D, target); radius = th * target->mvScaleFactors[level]; }
                        }
                    }
                }
                q.add(u, v, 0.f, radius, level, 0, 0.f, ok ? pMP->GetDescriptor() : cv::Mat(), true, ok);
            }
        };
        Queries q12, q21;
        side(vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, q12);
        side(vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, T12, pKF1, q21);
        const FrameArrays k1(pKF1->mvKeysUn, pKF1->mDescriptors), k2(pKF2->mvKeysUn, pKF2->mDescriptors);
        const ivf_bounds b1 = bounds_of(*pKF1), b2 = bounds_of(*pKF2);
        std::vector<int32_t> m12(std::max(N1, 1), -1);
        int nFound = 0;
        if (N1 > 0 && N2 > 0)
            check(ivf_search_by_sim3(k1.kps.data(), k1.desc, N1, &b1, k2.kps.data(), k2.desc, N2, &b2, q12.u.data(), q12.v.data(), q12.radius.data(),
                                     q12.level.data(), q12.desc.data(), q12.valid.data(), q21.u.data(), q21.v.data(), q21.radius.data(),
                                     q21.level.data(), q21.desc.data(), q21.valid.data(), m12.data(), &nFound, DeviceId()), "ivf_search_by_sim3");
        for (int i1 = 0; i1 < N1; i1++) if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];
        return nFound;
    }

    // ---- ORB/src/ORBmatcher.cc:831-981 (LocalMapping::SearchInNeighbors) ----
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0)
    {
        using namespace ivf::mat;
        const M3 Rcw = rot_of(pKF->GetRotation()); const V3 tcw = vec3(pKF->GetTranslation()), Ow = vec3(pKF->GetCameraCenter());
        const float bf = pKF->mbf;
        Queries q; std::vector<int> src;
        for (int i = 0, n = (int)vpMapPoints.size(); i < n; i++) {
            MapPoint* pMP = vpMapPoints[i];
            if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            float u, v, dist3D; V3 p3Dc;
            if (!project_kf(pKF, Rcw, tcw, Ow, pMP, true, u, v, dist3D, p3Dc)) continue;
            const float invz = 1 / p3Dc.z;
            const float ur = u - bf * invz;
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
            q.add(u, v, ur, th * pKF->mvScaleFactors[nPredictedLevel], nPredictedLevel, 0, 0.f, pMP->GetDescriptor(), true);
            src.push_back(i);
        }
        int nFused = 0;
        if (q.n() == 0 || pKF->N == 0) return 0;
        const FrameArrays fa(pKF->mvKeysUn, pKF->mDescriptors);
        const ivf_bounds bd = bounds_of(*pKF);
        std::vector<int32_t> best(q.n(), -1);
        check(ivf_fuse_candidates(fa.kps.data(), fa.desc, pKF->mvuRight.data(), pKF->N, &bd, pKF->mvInvLevelSigma2.data(),
                                  (int)pKF->mvInvLevelSigma2.size(), q.n(), q.u.data(), q.v.data(), q.ur.data(), q.radius.data(), q.level.data(),
                                  q.desc.data(), nullptr, best.data(), nullptr, DeviceId()), "ivf_fuse_candidates");
        for (int k = 0; k < q.n(); k++) {                            // Replace / AddObservation bookkeeping in query order (:958-977)
            if (best[k] < 0) continue;
            MapPoint* pMP = vpMapPoints[src[k]];
            MapPoint* pMPinKF = pKF->GetMapPoint(best[k]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, best[k]);
                pKF->AddMapPoint(pMP, best[k]);
            }
            nFused++;
        }
        return nFused;
    }

    // ---- ORB/src/ORBmatcher.cc:983-1106 (LoopClosing::SearchAndFuse) ----
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint)
    {
        using namespace ivf::mat;
        M3 Rcw; V3 tcw, Ow;
        decompose_sim3(Scw, Rcw, tcw, Ow);
        const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
        Queries q; std::vector<int> src;
        for (int iMP = 0, n = (int)vpPoints.size(); iMP < n; iMP++) {
            MapPoint* pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            float u, v, dist3D; V3 p3Dc;
            if (!project_kf(pKF, Rcw, tcw, Ow, pMP, false, u, v, dist3D, p3Dc)) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
            q.add(u, v, 0.f, th * pKF->mvScaleFactors[nPredictedLevel], nPredictedLevel, 0, 0.f, pMP->GetDescriptor(), true);
            src.push_back(iMP);
        }
        int nFused = 0;
        if (q.n() == 0 || pKF->N == 0) return 0;
        const FrameArrays fa(pKF->mvKeysUn, pKF->mDescriptors);
        const ivf_bounds bd = bounds_of(*pKF);
        std::vector<int32_t> best(q.n(), -1);
        check(ivf_fuse_candidates(fa.kps.data(), fa.desc, nullptr, pKF->N, &bd, nullptr, 0, q.n(), q.u.data(), q.v.data(), nullptr, q.radius.data(),
                                  q.level.data(), q.desc.data(), nullptr, best.data(), nullptr, DeviceId()), "ivf_fuse_candidates");
        for (int k = 0; k < q.n(); k++) {                            // :1085-1100
            if (best[k] < 0) continue;
            MapPoint* pMP = vpPoints[src[k]];
            MapPoint* pMPinKF = pKF->GetMapPoint(best[k]);
            if (pMPinKF) { if (!pMPinKF->isBad()) vpReplacePoint[src[k]] = pMPinKF; }
            else { pMP->AddObservation(pKF, best[k]); pKF->AddMapPoint(pMP, best[k]); }
            nFused++;
        }
        return nFused;
    }

    // ---- ORB/src/ORBmatcher.cc:1108-1121 ----
    void UpdateQualityScores(Frame& F)
    {
        const float kDeltaThresh = 0.01;
        for (size_t i = 0; i < F.mvpMapPoints.size(); i++) {
            if (!F.mvpMapPoints[i]) continue;
            const float mpt_qual = F.mvpMapPoints[i]->GetQualityScore();
            const float updated_qual = std::min(mpt_qual, F.mvKeyQualScore[i]);
            if (std::fabs(updated_qual - mpt_qual) > kDeltaThresh) F.mvpMapPoints[i]->SetQualityScore(updated_qual);
            F.mvKeyQualScore[i] = updated_qual;
        }
    }
    // ---- ORB/src/ORBmatcher.cc:1123-1143 ----
    void UpdateQualityScores(KeyFrame& KF)
    {
        const float kDeltaThresh = 0.01;
        const std::vector<MapPoint*> vpMPs = KF.GetMapPointMatches();
        for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
            MapPoint* pMP = vpMPs[i];
            if (!pMP) continue;
            const int keypt_idx = pMP->GetIndexInKeyFrame(&KF);
            const float mpt_qual = pMP->GetQualityScore();
            const float updated_qual = std::min(mpt_qual, KF.mvKeyQualScore[keypt_idx]);
            if (std::fabs(updated_qual - mpt_qual) > kDeltaThresh) pMP->SetQualityScore(updated_qual);
            KF.mvKeyQualScore[keypt_idx] = updated_qual;
        }
    }

protected:
    // ORB/src/ORBmatcher.cc:137-143
    float RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }

    static void check(int rc, const char* what) { if (rc != IVF_OK) throw std::runtime_error(std::string(what) + ": " + ivf_last_error()); }

    // flat query arrays of one search
    struct Queries {
        std::vector<float> u, v, ur, radius, angle; std::vector<int32_t> level, maxLevel; std::vector<uint8_t> desc, blocks, valid;
        int n() const { return (int)u.size(); }
        void add(float u_, float v_, float ur_, float r_, int level_, int maxLevel_, float angle_, const cv::Mat& d, bool blocks_, bool valid_ = true)
        {
            u.push_back(u_); v.push_back(v_); ur.push_back(ur_); radius.push_back(r_); level.push_back(level_); maxLevel.push_back(maxLevel_);
            angle.push_back(angle_); blocks.push_back(blocks_ ? 1 : 0); valid.push_back(valid_ ? 1 : 0);
            const size_t o = desc.size(); desc.resize(o + 32, 0);
            if (valid_ && !d.empty()) std::memcpy(&desc[o], d.template ptr<uint8_t>(), 32);
        }
    };
    // keypoints of a frame / keyframe as the C-ABI's plain structs; descriptors are used in place (N x 32 CV_8U, continuous rows)
    struct FrameArrays {
        std::vector<ivf_keypoint> kps; const uint8_t* desc;
        FrameArrays(const std::vector<cv::KeyPoint>& in, const cv::Mat& descriptors) : kps(std::max<size_t>(in.size(), 1)), desc(nullptr)
        {
            for (size_t i = 0; i < in.size(); i++) kps[i] = ivf_keypoint{in[i].pt.x, in[i].pt.y, in[i].size, in[i].angle, in[i].response, in[i].octave};
            if (!descriptors.empty()) desc = descriptors.template ptr<uint8_t>();
        }
    };
    // DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) in CSR form, nodes in map order
    struct Csr {
        std::vector<int32_t> node, start, idx;
        template <class FeatureVector> explicit Csr(const FeatureVector& fv)
        {
            start.push_back(0);
            for (typename FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
                node.push_back((int32_t)it->first);
                for (size_t k = 0; k < it->second.size(); k++) idx.push_back((int32_t)it->second[k]);
                start.push_back((int32_t)idx.size());
            }
            if (node.empty()) node.push_back(0);
            if (idx.empty()) idx.push_back(0);
        }
        int n() const { return (int)start.size() - 1; }
    };
    template <class F> static ivf_bounds bounds_of(const F& f) { return ivf_bounds{(float)f.mnMinX, (float)f.mnMinY, (float)f.mnMaxX, (float)f.mnMaxY}; }
    // -1 free / -2 blocked on entry: a slot holding a map point blocks always (anyOccupant) or only when that point has observations
    static std::vector<int32_t> occupancy(const std::vector<MapPoint*>& pts, bool anyOccupant)
    {
        std::vector<int32_t> a(std::max<size_t>(pts.size(), 1), -1);
        for (size_t i = 0; i < pts.size(); i++) if (pts[i] && (anyOccupant || pts[i]->Observations() > 0)) a[i] = -2;
        return a;
    }
    static std::vector<uint8_t> good_points(const std::vector<MapPoint*>& pts)
    {
        std::vector<uint8_t> h(std::max<size_t>(pts.size(), 1), 0);
        for (size_t i = 0; i < pts.size(); i++) h[i] = (pts[i] && !pts[i]->isBad()) ? 1 : 0;
        return h;
    }
    // Scw -> Rcw, tcw, Ow (:304-309, :990-995)
    static void decompose_sim3(const cv::Mat& Scw, ivf::mat::M3& Rcw, ivf::mat::V3& tcw, ivf::mat::V3& Ow)
    {
        using namespace ivf::mat;
        const M3 sRcw = rot_of(Scw);
        const V3 row0{sRcw.m[0], sRcw.m[1], sRcw.m[2]};
        const float scw = (float)std::sqrt(dot(row0, row0));
        Rcw = scaled(sRcw, 1.0 / scw);
        tcw = scaled(trans_of(Scw), 1.0 / scw);
        Ow = neg_rt_mul(Rcw, tcw);
    }
    // the common projection of a map point into a keyframe (:318-365, :859-905, :1011-1060): depth > 0, inside the image,
    // inside the scale-invariance range, viewing angle < 60 deg.  invzIsFloatOne: `1/z` (float) at :333,:876 vs `1.0/z` (double) at :1024.
    static bool project_kf(KeyFrame* pKF, const ivf::mat::M3& Rcw, const ivf::mat::V3& tcw, const ivf::mat::V3& Ow, MapPoint* pMP,
                           bool invzIsFloatOne, float& u, float& v, float& dist, ivf::mat::V3& p3Dc)
    {
        using namespace ivf::mat;
        const V3 p3Dw = vec3(pMP->GetWorldPos());
        p3Dc = mul_add(Rcw, p3Dw, tcw);
        if (p3Dc.z < 0.0) return false;
        const float invz = invzIsFloatOne ? 1 / p3Dc.z : (float)(1.0 / p3Dc.z);
        const float x = p3Dc.x * invz, y = p3Dc.y * invz;
        u = pKF->fx * x + pKF->cx; v = pKF->fy * y + pKF->cy;
        if (!pKF->IsInImage(u, v)) return false;
        const V3 PO = sub(p3Dw, Ow);
        dist = (float)norm(PO);
        if (dist < pMP->GetMinDistanceInvariance() || dist > pMP->GetMaxDistanceInvariance()) return false;
        if (dot(PO, vec3(pMP->GetNormal())) < 0.5 * dist) return false;
        return true;
    }

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
