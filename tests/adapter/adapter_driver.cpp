// TEST-ONLY driver: compiles include/ivfront_orbslam.hpp against MOCK Frame / KeyFrame / MapPoint types that carry the
// reference's member names (ORB/include/Frame.h, KeyFrame.h, MapPoint.h) and RUNS the adapter's ORBmatcher methods end
// to end on a scenario file written by tests/test_gpu_adapter.py; results go back as a flat int32 file.  The Python side
// holds the expected values (independent numpy restatement of the projection loops + the C oracle's searches).
#include "ivfront_orbslam.hpp"
#include <cstdio>
#include <map>

namespace ORB_SLAM2 {
struct KeyFrame;
struct Frame;
struct MapPoint {
    cv::Mat mWorldPos, mDescriptor, mNormal;
    int nObs = 0; bool bad = false; float minDist = 0.f, maxDist = 1e9f, quality = 1.f;
    bool mbTrackInView = false; int mnTrackScaleLevel = 0; float mTrackViewCos = 1.f, mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    std::map<const void*, int> obs;
    MapPoint* replacedBy = nullptr;
    bool isBad() { return bad; }
    cv::Mat GetWorldPos() { return mWorldPos; }
    cv::Mat GetDescriptor() { return mDescriptor; }
    cv::Mat GetNormal() { return mNormal; }
    int Observations() { return nObs; }
    float GetMaxDistanceInvariance() { return 1.2f * maxDist; }      // MapPoint.cc:374-384
    float GetMinDistanceInvariance() { return 0.8f * minDist; }
    template <class K> int PredictScale(const float& currentDist, K* k)   // MapPoint.cc:386-420
    {
        const float ratio = maxDist / currentDist;
        int nScale = (int)std::ceil(std::log(ratio) / k->mfLogScaleFactor);
        if (nScale < 0) nScale = 0; else if (nScale >= k->mnScaleLevels) nScale = k->mnScaleLevels - 1;
        return nScale;
    }
    bool IsInKeyFrame(KeyFrame* k) { return obs.count(k) != 0; }
    int GetIndexInKeyFrame(const KeyFrame* k) { return obs.count(k) ? obs[k] : -1; }
    void AddObservation(KeyFrame* k, size_t idx) { obs[k] = (int)idx; nObs++; }
    void Replace(MapPoint* p) { replacedBy = p; bad = true; }
    float GetQualityScore() { return quality; }
    void SetQualityScore(float q) { quality = q; }
};
struct Base {       // what Frame and KeyFrame share in the reference
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn; std::vector<float> mvuRight, mvKeyQualScore; cv::Mat mDescriptors;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0, mfLogScaleFactor = 0; int mnScaleLevels = 8;
    float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;
};
struct Frame : Base {
    std::vector<MapPoint*> mvpMapPoints; std::vector<bool> mvbOutlier; cv::Mat mTcw;
};
struct KeyFrame : Base {
    std::vector<MapPoint*> mvpMapPoints; cv::Mat Tcw;
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint* GetMapPoint(size_t i) { return mvpMapPoints[i]; }
    std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }
    void AddMapPoint(MapPoint* p, size_t i) { mvpMapPoints[i] = p; }
    cv::Mat GetRotation() { cv::Mat R(3, 3, CV_32F); for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R.at<float>(i, j) = Tcw.at<float>(i, j); return R; }
    cv::Mat GetTranslation() { cv::Mat t(3, 1, CV_32F); for (int i = 0; i < 3; i++) t.at<float>(i, 0) = Tcw.at<float>(i, 3); return t; }
    cv::Mat GetCameraCenter() { cv::Mat c(3, 1, CV_32F); for (int i = 0; i < 3; i++) c.at<float>(i, 0) = Ow[i]; return c; }
    float Ow[3] = {0, 0, 0};
    bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }   // KeyFrame.cc:597-600
};
typedef ORBmatcherT<Frame, KeyFrame, MapPoint> ORBmatcher;      // what ORB/include/ORBmatcher.h becomes
}  // namespace ORB_SLAM2
using namespace ORB_SLAM2;

// ---- scenario file: little-endian, every array as {int32 count; payload} --------------------------------------------
struct Reader {
    FILE* f;
    template <class T> std::vector<T> arr() { int32_t n = 0; if (fread(&n, 4, 1, f) != 1) abort(); std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != (size_t)n) abort(); return v; }
    float f32() { float v; if (fread(&v, 4, 1, f) != 1) abort(); return v; }
    int i32() { int32_t v; if (fread(&v, 4, 1, f) != 1) abort(); return v; }
};
static cv::Mat mat_f(const std::vector<float>& v, int r, int c) { cv::Mat m(r, c, CV_32F); for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) m.at<float>(i, j) = v[(size_t)i * c + j]; return m; }

static void read_base(Reader& R, Base& b)
{
    std::vector<float> k = R.arr<float>();             // n x 6: x y size angle response octave
    b.N = (int)k.size() / 6;
    b.mvKeysUn.resize(b.N);
    for (int i = 0; i < b.N; i++) b.mvKeysUn[i] = cv::KeyPoint(k[6 * i], k[6 * i + 1], k[6 * i + 2], k[6 * i + 3], k[6 * i + 4], (int)k[6 * i + 5]);
    b.mvKeys = b.mvKeysUn;
    std::vector<uint8_t> d = R.arr<uint8_t>();
    b.mDescriptors.create(b.N, 32, CV_8U);
    if (b.N) memcpy(b.mDescriptors.data, d.data(), d.size());
    b.mvuRight = R.arr<float>(); b.mvKeyQualScore.assign(b.N, 1.f);
    b.mvScaleFactors = R.arr<float>(); b.mvLevelSigma2 = R.arr<float>(); b.mvInvLevelSigma2 = R.arr<float>();
    b.mnScaleLevels = (int)b.mvScaleFactors.size();
    b.fx = R.f32(); b.fy = R.f32(); b.cx = R.f32(); b.cy = R.f32(); b.mbf = R.f32(); b.mb = R.f32(); b.mfLogScaleFactor = R.f32();
    b.mnMinX = R.f32(); b.mnMinY = R.f32(); b.mnMaxX = R.f32(); b.mnMaxY = R.f32();
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: adapter_driver scenario.bin result.bin\n"); return 2; }
    Reader R{fopen(argv[1], "rb")};
    if (!R.f) return 2;
    std::vector<int32_t> out;
    auto put_assign = [&](const std::vector<MapPoint*>& v, const std::vector<MapPoint>& pool) {
        out.push_back((int32_t)v.size());
        for (MapPoint* p : v) out.push_back(p ? (int32_t)(p - pool.data()) : -1);
    };
    try {
        // map points (world position, descriptor, normal, observations, distance range, tracking fields)
        const int nMP = R.i32();
        std::vector<MapPoint> pool(nMP);
        for (int i = 0; i < nMP; i++) {
            MapPoint& m = pool[i];
            std::vector<float> p = R.arr<float>();     // 3 pos, 3 normal, minDist, maxDist, viewCos, projX, projY, projXR
            m.mWorldPos = mat_f(std::vector<float>(p.begin(), p.begin() + 3), 3, 1);
            m.mNormal = mat_f(std::vector<float>(p.begin() + 3, p.begin() + 6), 3, 1);
            m.minDist = p[6]; m.maxDist = p[7]; m.mTrackViewCos = p[8]; m.mTrackProjX = p[9]; m.mTrackProjY = p[10]; m.mTrackProjXR = p[11];
            std::vector<int32_t> q = R.arr<int32_t>(); // nObs, bad, trackInView, trackScaleLevel
            m.nObs = q[0]; m.bad = q[1] != 0; m.mbTrackInView = q[2] != 0; m.mnTrackScaleLevel = q[3];
            std::vector<uint8_t> d = R.arr<uint8_t>();
            m.mDescriptor.create(1, 32, CV_8U); memcpy(m.mDescriptor.data, d.data(), 32);
        }
        std::vector<char> poolBad0(nMP); std::vector<int> poolObs0(nMP);
        for (int i = 0; i < nMP; i++) { poolBad0[i] = pool[i].bad; poolObs0[i] = pool[i].nObs; }
        Frame last, cur; KeyFrame kf;
        read_base(R, last); read_base(R, cur); read_base(R, kf);
        last.mTcw = mat_f(R.arr<float>(), 4, 4); cur.mTcw = mat_f(R.arr<float>(), 4, 4); kf.Tcw = mat_f(R.arr<float>(), 4, 4);
        { std::vector<float> ow = R.arr<float>(); for (int i = 0; i < 3; i++) kf.Ow[i] = ow[i]; }
        std::vector<int32_t> lastMp = R.arr<int32_t>(), lastOut = R.arr<int32_t>(), curMp = R.arr<int32_t>(), kfMp = R.arr<int32_t>();
        last.mvpMapPoints.resize(last.N); last.mvbOutlier.resize(last.N);
        for (int i = 0; i < last.N; i++) { last.mvpMapPoints[i] = lastMp[i] >= 0 ? &pool[lastMp[i]] : nullptr; last.mvbOutlier[i] = lastOut[i] != 0; }
        auto set_cur = [&]() { cur.mvpMapPoints.resize(cur.N); for (int i = 0; i < cur.N; i++) cur.mvpMapPoints[i] = curMp[i] >= 0 ? &pool[curMp[i]] : nullptr; };
        kf.mvpMapPoints.resize(kf.N);
        for (int i = 0; i < kf.N; i++) {
            kf.mvpMapPoints[i] = kfMp[i] >= 0 ? &pool[kfMp[i]] : nullptr;
            if (kfMp[i] >= 0) pool[kfMp[i]].obs[&kf] = i;           // MapPoint::IsInKeyFrame(pKF) (the observation count is scenario data)
        }
        const float th = R.f32(); const int bMono = R.i32();
        std::vector<int32_t> localIdx = R.arr<int32_t>();       // indices of the "local map" points for SearchByProjection(F, vpMapPoints)
        std::vector<float> Scw = R.arr<float>();
        Reader& extra = R;                                           // the rest of the file feeds steps 6-11

        ORBmatcher matcher(0.9f, true);
        // 1. SearchByProjection(CurrentFrame, LastFrame, th, bMono)      (Tracking.cc:1303-1342 call site)
        set_cur();
        out.push_back(matcher.SearchByProjection(cur, last, th, bMono != 0));
        put_assign(cur.mvpMapPoints, pool);
        // 2. SearchByProjection(F, vpMapPoints, th) on top of that state      (Tracking::SearchLocalPoints)
        std::vector<MapPoint*> local;
        for (int i : localIdx) local.push_back(&pool[i]);
        ORBmatcher matcher2(0.8f);
        out.push_back(matcher2.SearchByProjection(cur, local, 3.0f));
        put_assign(cur.mvpMapPoints, pool);
        // 3. SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)   (Tracking::Relocalization)
        set_cur();
        std::set<MapPoint*> found;
        for (int i = 0; i < cur.N; i++) if (cur.mvpMapPoints[i]) found.insert(cur.mvpMapPoints[i]);
        out.push_back(matcher.SearchByProjection(cur, &kf, found, 10.0f, 100));
        put_assign(cur.mvpMapPoints, pool);
        // 4. SearchByProjection(pKF, Scw, vpPoints, vpMatched, th)      (LoopClosing)
        std::vector<MapPoint*> vpMatched(kf.N, nullptr);
        out.push_back(matcher.SearchByProjection(&kf, mat_f(Scw, 4, 4), local, vpMatched, 10));
        put_assign(vpMatched, pool);
        // 5. Fuse(pKF, vpMapPoints, th): bookkeeping visible through the mock      (LocalMapping::SearchInNeighbors)
        out.push_back(matcher.Fuse(&kf, local, 3.0f));
        put_assign(kf.mvpMapPoints, pool);
        out.push_back(nMP);
        for (int i = 0; i < nMP; i++) out.push_back(pool[i].replacedBy ? (int32_t)(pool[i].replacedBy - pool.data()) : -1);
        // ---- the BoW / initialisation / triangulation / Sim3 searches: a second keyframe from the current frame's data ----
        KeyFrame kf2;
        static_cast<Base&>(kf2) = static_cast<const Base&>(cur);
        kf2.Tcw = cur.mTcw;
        { std::vector<float> ow2 = extra.arr<float>(); for (int i = 0; i < 3; i++) kf2.Ow[i] = ow2[i]; }
        std::vector<int32_t> kf2Mp = extra.arr<int32_t>(), kfMp2 = extra.arr<int32_t>();
        kf2.mvpMapPoints.resize(kf2.N);
        std::vector<MapPoint> pool2(pool.size());                    // fresh copies: the Fuse above replaced / re-observed some points
        for (size_t i = 0; i < pool.size(); i++) { pool2[i] = pool[i]; pool2[i].bad = poolBad0[i]; pool2[i].nObs = poolObs0[i]; pool2[i].obs.clear(); pool2[i].replacedBy = nullptr; }
        for (int i = 0; i < kf.N; i++) { kf.mvpMapPoints[i] = kfMp2[i] >= 0 ? &pool2[kfMp2[i]] : nullptr; if (kfMp2[i] >= 0) pool2[kfMp2[i]].obs[&kf] = i; }
        for (int i = 0; i < kf2.N; i++) { kf2.mvpMapPoints[i] = kf2Mp[i] >= 0 ? &pool2[kf2Mp[i]] : nullptr; if (kf2Mp[i] >= 0) pool2[kf2Mp[i]].obs[&kf2] = i; }
        // DBoW2::FeatureVector stand-ins: node = first descriptor byte & 31, features in index order (what transform() would give)
        auto featvec = [](Base& b) { b.mFeatVec.clear(); for (int i = 0; i < b.N; i++) b.mFeatVec[b.mDescriptors.at<uint8_t>(i, 0) & 31u].push_back((unsigned)i); };
        featvec(kf); featvec(kf2); featvec(cur); featvec(last);
        // 6. SearchByBoW(pKF, F, vpMapPointMatches)
        std::vector<MapPoint*> bowF;
        ORBmatcher m7(0.7f, true);
        out.push_back(m7.SearchByBoW(&kf, cur, bowF));
        put_assign(bowF, pool2);
        // 7. SearchByBoW(pKF1, pKF2, vpMatches12)
        std::vector<MapPoint*> bowKK;
        ORBmatcher m75(0.75f, true);
        out.push_back(m75.SearchByBoW(&kf, &kf2, bowKK));
        put_assign(bowKK, pool2);
        // 8. SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, 100)
        std::vector<cv::Point2f> prev(last.N);
        for (int i = 0; i < last.N; i++) prev[i] = last.mvKeysUn[i].pt;
        std::vector<int> m12;
        out.push_back(matcher.SearchForInitialization(last, cur, prev, m12, 100));
        out.push_back((int32_t)m12.size());
        for (int v : m12) out.push_back(v);
        for (int i = 0; i < last.N; i++) { int32_t a, b2; memcpy(&a, &prev[i].x, 4); memcpy(&b2, &prev[i].y, 4); out.push_back(a); out.push_back(b2); }
        // 9. SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, false)
        std::vector<float> F12 = extra.arr<float>();
        std::vector<std::pair<size_t, size_t> > pairs;
        ORBmatcher m6(0.6f, false);
        out.push_back(m6.SearchForTriangulation(&kf, &kf2, mat_f(F12, 3, 3), pairs, false));
        out.push_back((int32_t)pairs.size());
        for (auto& pr : pairs) { out.push_back((int32_t)pr.first); out.push_back((int32_t)pr.second); }
        // 10. SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)
        std::vector<float> s12v = extra.arr<float>();                // s12, R12 (9), t12 (3)
        std::vector<MapPoint*> vpM12(kf.N, nullptr);
        { std::vector<int32_t> pre = extra.arr<int32_t>(); for (int i = 0; i < kf.N; i++) if (pre[i] >= 0) vpM12[i] = &pool2[pre[i]]; }
        out.push_back(matcher.SearchBySim3(&kf, &kf2, vpM12, s12v[0], mat_f(std::vector<float>(s12v.begin() + 1, s12v.begin() + 10), 3, 3),
                                           mat_f(std::vector<float>(s12v.begin() + 10, s12v.begin() + 13), 3, 1), 7.5f));
        put_assign(vpM12, pool2);
        // 11. Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
        std::vector<MapPoint*> local2, repl;
        for (int i : localIdx) local2.push_back(&pool2[i]);
        repl.assign(local2.size(), nullptr);
        out.push_back(matcher.Fuse(&kf2, mat_f(Scw, 4, 4), local2, 4.0f, repl));
        put_assign(repl, pool2);
        put_assign(kf2.mvpMapPoints, pool2);
        // 12. UpdateQualityScores(Frame&) / (KeyFrame&) and DescriptorDistance
        for (int i = 0; i < cur.N; i++) cur.mvKeyQualScore[i] = 0.5f + 0.001f * (i % 400);
        for (size_t i = 0; i < pool2.size(); i++) pool2[i].quality = 0.4f + 0.002f * (i % 300);
        set_cur();
        for (int i = 0; i < cur.N; i++) if (cur.mvpMapPoints[i]) cur.mvpMapPoints[i] = &pool2[cur.mvpMapPoints[i] - pool.data()];
        matcher.UpdateQualityScores(cur);
        for (int i = 0; i < cur.N; i++) { int32_t a; memcpy(&a, &cur.mvKeyQualScore[i], 4); out.push_back(a); }
        for (size_t i = 0; i < pool2.size(); i++) { int32_t a; memcpy(&a, &pool2[i].quality, 4); out.push_back(a); }
        out.push_back(ORBmatcher::DescriptorDistance(pool[0].mDescriptor, pool[1 % nMP].mDescriptor));
        fclose(R.f);
    } catch (const std::exception& e) { fprintf(stderr, "adapter_driver: %s\n", e.what()); return 1; }
    FILE* o = fopen(argv[2], "wb");
    fwrite(out.data(), 4, out.size(), o); fclose(o);
    return 0;
}

// UpdateQualityScores(KeyFrame&) is not reached by main(): instantiate it
void instantiate_rest(ORBmatcher& m, KeyFrame* a) { m.UpdateQualityScores(*a); }
