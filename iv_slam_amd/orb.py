"""Host-side mirror of the reference's class surfaces for the hot path, on top of the C-ABI.

Same names, argument meaning and error behaviour as
  ORB_SLAM2::ORBextractor   (ORB/include/ORBextractor.h:51-126, ORB/src/ORBextractor.cc:411-476,1224-1296)
  ORB_SLAM2::ORBmatcher     (ORB/include/ORBmatcher.h:37-108, ORB/src/ORBmatcher.cc:1372-1518,1700-1716)
  Frame::ComputeStereoMatches / GetFeaturesInArea (ORB/src/Frame.cc:758-932, 615-680)
so parity tests read like tests of the reference classes.  Every method calls libivfront.so (HIP);
nothing here computes on the CPU except trivial glue.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, Bounds, ExtractorParams, check, ptr


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, enableIntrospection=False)."""

    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, enableIntrospection=False, device_id=0):
        self._lib = _lib.load()
        self.params = ExtractorParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, int(bool(enableIntrospection)))
        h = C.c_void_p()
        check(self._lib.ivf_extractor_create(C.byref(self.params), device_id, C.byref(h)))
        self._h = h
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ivf_extractor_destroy(self._h)
            self._h = None

    # --- getters (ORBextractor.h:69-89)
    def GetLevels(self):
        return self._lib.ivf_extractor_get_levels(self._h)

    def GetScaleFactor(self):
        return self._lib.ivf_extractor_get_scale_factor(self._h)

    def _tables(self):
        n = self.nlevels
        t = [np.zeros(n, np.float32) for _ in range(4)]
        check(self._lib.ivf_extractor_get_scale_tables(self._h, *[ptr(a) for a in t]))
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def feature_tables(self):
        nf = np.zeros(self.nlevels, np.int32); um = np.zeros(16, np.int32)
        check(self._lib.ivf_extractor_get_feature_tables(self._h, ptr(nf), ptr(um)))
        return nf, um

    # --- operator()(image, mask, keypoints, descriptors)
    def __call__(self, image, mask=None, cap=None):
        """Returns (keypoints[KP_DTYPE], descriptors[n,32] u8).  Empty image -> empty outputs (ORBextractor.cc:1227)."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise AssertionError("image.type() == CV_8UC1")              # ORBextractor.cc:1241
        image = np.ascontiguousarray(image)
        h, w = image.shape
        if mask is not None and mask.size:
            if mask.dtype != np.uint8 or mask.shape != image.shape:
                raise AssertionError("mask.type() == CV_8UC1 and same size as the image")   # :1234
            mask = np.ascontiguousarray(mask)
        else:
            mask = None
        cap = cap or self.nfeatures
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        check(self._lib.ivf_extract(self._h, ptr(image), w, h, w, ptr(mask), w, ptr(kps), ptr(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    # --- public data members mvImagePyramid / mvQualityImagePyramid (ORBextractor.h:91-92)
    def _level(self, fn, level):
        w = C.c_int(); h = C.c_int()
        check(fn(self._h, level, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        check(fn(self._h, level, ptr(out), w.value, C.byref(w), C.byref(h)))
        return out

    @property
    def mvImagePyramid(self):
        return [self._level(self._lib.ivf_extractor_pyramid_level, l) for l in range(self.nlevels)]

    @property
    def mvQualityImagePyramid(self):
        return [self._level(self._lib.ivf_extractor_quality_level, l) for l in range(self.nlevels)]

    def set_opencv_variant(self, blur=0, retain_best=0, atan2=0):
        """OpenCV-version switches (include/ivfront.h: ivf_extractor_set_opencv_variant); 0,0,0 = OpenCV >= 3.4.2 / 4.x."""
        check(self._lib.ivf_extractor_set_opencv_variant(self._h, int(blur), int(retain_best), int(atan2)))

    def blur_level(self, level):
        """7x7 sigma-2 blurred copy of mvImagePyramid[level] (the reference's local workingMat, ORBextractor.cc:1276-1277)."""
        return self._level(self._lib.ivf_extractor_blur_level, level)

    def level_counts(self):
        c = np.zeros(self.nlevels, np.int32)
        check(self._lib.ivf_extractor_level_counts(self._h, ptr(c)))
        return c.tolist()


def ComputeStereoMatches(extractorLeft, extractorRight, mvKeys, mDescriptors, mvKeysRight, mDescriptorsRight, mbf, mb):
    """Frame::ComputeStereoMatches (ORB/src/Frame.cc:758-932) -> (mvuRight, mvDepth)."""
    lib = _lib.load()
    kl = np.ascontiguousarray(mvKeys, KP_DTYPE); kr = np.ascontiguousarray(mvKeysRight, KP_DTYPE)
    dl = np.ascontiguousarray(mDescriptors, np.uint8); dr = np.ascontiguousarray(mDescriptorsRight, np.uint8)
    ur = np.full(len(kl), -1.0, np.float32); dp = np.full(len(kl), -1.0, np.float32)
    check(lib.ivf_stereo_match(extractorLeft._h, extractorRight._h, ptr(kl), len(kl), ptr(dl), ptr(kr), len(kr), ptr(dr),
                               mbf, mb, ptr(ur), ptr(dp)))
    return ur, dp


def GetFeaturesInArea(mvKeysUn, bounds, x, y, r, minLevel=-1, maxLevel=-1):
    """Frame::GetFeaturesInArea (ORB/src/Frame.cc:615-668); bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY)."""
    lib = _lib.load()
    k = np.ascontiguousarray(mvKeysUn, KP_DTYPE)
    out = np.zeros(max(len(k), 1), np.int32)
    n = C.c_int(0)
    bd = Bounds(*bounds)
    check(lib.ivf_features_in_area(ptr(k), len(k), C.byref(bd), x, y, r, minLevel, maxLevel, ptr(out), len(out), C.byref(n)))
    return out[:n.value].copy()


class ORBmatcher:
    """ORBmatcher(nnratio=0.6, checkOri=True) (ORB/include/ORBmatcher.h:41)."""

    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio=0.6, checkOri=True, device_id=0):
        self._lib = _lib.load()
        self.mfNNratio, self.mbCheckOrientation, self.device_id = nnratio, checkOri, device_id

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        return _lib.load().ivf_hamming(ptr(a), ptr(b))

    def DescriptorDistances(self, descA, descB, pairs):
        """Batch of DescriptorDistance on the device: pairs[n,2] of (rowA,rowB)."""
        a = np.ascontiguousarray(descA, np.uint8); b = np.ascontiguousarray(descB, np.uint8)
        p = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        d = np.zeros(len(p), np.int32)
        check(self._lib.ivf_hamming_pairs(ptr(a), len(a), ptr(b), len(b), ptr(p), len(p), ptr(d), self.device_id))
        return d

    def SearchByProjection(self, cur_kps, cur_desc, cur_uright, bounds, q, cur_assign=None):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1372-1518) on projected queries
        (see include/ivfront.h).  Returns (CurrentFrame.mvpMapPoints as query indices, nmatches)."""
        cur_kps = np.ascontiguousarray(cur_kps, KP_DTYPE); cur_desc = np.ascontiguousarray(cur_desc, np.uint8)
        cur_uright = np.ascontiguousarray(cur_uright, np.float32)
        n_cur = len(cur_kps); n_q = len(q["u"])
        assign = np.full(n_cur, -1, np.int32) if cur_assign is None else np.ascontiguousarray(cur_assign, np.int32).copy()
        types = dict(u=np.float32, v=np.float32, ur=np.float32, radius=np.float32, min_level=np.int32,
                     max_level=np.int32, angle=np.float32, desc=np.uint8, valid=np.uint8, blocks=np.uint8)
        qq = {k: np.ascontiguousarray(q[k], t) for k, t in types.items()}
        nm = C.c_int(0)
        bd = Bounds(*bounds)
        check(self._lib.ivf_search_by_projection(ptr(cur_kps), ptr(cur_desc), ptr(cur_uright), n_cur, C.byref(bd), n_q,
                                                 ptr(qq["u"]), ptr(qq["v"]), ptr(qq["ur"]), ptr(qq["radius"]),
                                                 ptr(qq["min_level"]), ptr(qq["max_level"]), ptr(qq["angle"]),
                                                 ptr(qq["desc"]), ptr(qq["valid"]), ptr(qq["blocks"]),
                                                 int(self.mbCheckOrientation), ptr(assign), C.byref(nm), self.device_id))
        return assign, nm.value

    @staticmethod
    def RadiusByViewingCos(viewCos):
        """ORBmatcher::RadiusByViewingCos (ORB/src/ORBmatcher.cc:137-143)."""
        return 2.5 if viewCos > 0.998 else 4.0

    def SearchByProjectionMapPoints(self, cur_kps, cur_desc, cur_uright, bounds, q, cur_assign=None):
        """SearchByProjection(Frame &F, const vector<MapPoint*>&, th) (ORBmatcher.cc:45-135) on projected map points:
        q has u, v, ur, radius, level, desc, valid, blocks.  Returns (F.mvpMapPoints as query indices, nmatches)."""
        cur_kps = np.ascontiguousarray(cur_kps, KP_DTYPE); cur_desc = np.ascontiguousarray(cur_desc, np.uint8)
        cur_uright = np.ascontiguousarray(cur_uright, np.float32)
        n_cur = len(cur_kps); n_q = len(q["u"])
        assign = np.full(n_cur, -1, np.int32) if cur_assign is None else np.ascontiguousarray(cur_assign, np.int32).copy()
        types = dict(u=np.float32, v=np.float32, ur=np.float32, radius=np.float32, level=np.int32, desc=np.uint8,
                     valid=np.uint8, blocks=np.uint8)
        qq = {k: np.ascontiguousarray(q[k], t) for k, t in types.items()}
        nm = C.c_int(0)
        bd = Bounds(*bounds)
        check(self._lib.ivf_search_map_points(ptr(cur_kps), ptr(cur_desc), ptr(cur_uright), n_cur, C.byref(bd), n_q,
                                              ptr(qq["u"]), ptr(qq["v"]), ptr(qq["ur"]), ptr(qq["radius"]), ptr(qq["level"]),
                                              ptr(qq["desc"]), ptr(qq["valid"]), ptr(qq["blocks"]), self.mfNNratio,
                                              ptr(assign), C.byref(nm), self.device_id))
        return assign, nm.value

    def UpdateQualityScores(self, mvpMapPoints, mvKeyQualScore, mapPointQuality):
        """UpdateQualityScores(Frame &F) (ORBmatcher.cc:1108-1121): returns (mvKeyQualScore, mapPointQuality) updated."""
        a = np.ascontiguousarray(mvpMapPoints, np.int32)
        kq = np.ascontiguousarray(mvKeyQualScore, np.float32).copy(); mq = np.ascontiguousarray(mapPointQuality, np.float32).copy()
        check(self._lib.ivf_update_quality_scores(ptr(a), len(a), ptr(kq), ptr(mq), len(mq)))
        return kq, mq

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, bounds2, vbPrevMatched, windowSize=10):
        """SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:410-519) on the frames'
        mvKeysUn / mDescriptors.  Returns (vnMatches12, vbPrevMatched updated, nmatches)."""
        k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        prev = np.ascontiguousarray(vbPrevMatched, np.float32).reshape(-1, 2).copy()
        if len(prev) != len(k1):
            raise AssertionError("vbPrevMatched.size() == F1.mvKeysUn.size()")
        m12 = np.full(len(k1), -1, np.int32); nm = C.c_int(0); bd = Bounds(*bounds2)
        check(self._lib.ivf_search_for_initialization(ptr(k1), ptr(d1), len(k1), ptr(k2), ptr(d2), len(k2), C.byref(bd), ptr(prev),
                                                      int(windowSize), self.mfNNratio, int(self.mbCheckOrientation), ptr(m12),
                                                      C.byref(nm), self.device_id))
        return m12, prev, nm.value

    def SearchByProjectionKeyFrame(self, kf_kps, kf_desc, bounds, q, vpMatched=None):
        """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:296-404) on projected candidates:
        q has u, v, radius, level, desc, valid.  Returns (vpMatched as query indices, nmatches)."""
        k = np.ascontiguousarray(kf_kps, KP_DTYPE); d = np.ascontiguousarray(kf_desc, np.uint8)
        m = np.full(len(k), -1, np.int32) if vpMatched is None else np.ascontiguousarray(vpMatched, np.int32).copy()
        t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, desc=np.uint8, valid=np.uint8)
        qq = {a: np.ascontiguousarray(q[a], b) for a, b in t.items()}
        nm = C.c_int(0); bd = Bounds(*bounds)
        check(self._lib.ivf_search_keyframe_points(ptr(k), ptr(d), len(k), C.byref(bd), len(qq["u"]), ptr(qq["u"]), ptr(qq["v"]),
                                                   ptr(qq["radius"]), ptr(qq["level"]), ptr(qq["desc"]), ptr(qq["valid"]), ptr(m),
                                                   C.byref(nm), self.device_id))
        return m, nm.value

    def FuseCandidates(self, kf_kps, kf_desc, kf_uright, bounds, mvInvLevelSigma2, q):
        """Matching core of Fuse(KeyFrame*, vpMapPoints, th) (ORBmatcher.cc:893-955): q has u, v, ur, radius, level, desc,
        valid.  Returns (best_idx, best_dist) per map point; -1 = nothing within TH_LOW."""
        k = np.ascontiguousarray(kf_kps, KP_DTYPE); d = np.ascontiguousarray(kf_desc, np.uint8)
        gate = mvInvLevelSigma2 is not None          # None: Fuse(KF, Scw, ...) (:983-1106), no reprojection gate
        ur = np.ascontiguousarray(kf_uright, np.float32) if gate else None
        sg = np.ascontiguousarray(mvInvLevelSigma2, np.float32) if gate else None
        t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, desc=np.uint8, valid=np.uint8)
        if gate: t["ur"] = np.float32
        qq = {a: np.ascontiguousarray(q[a], b) for a, b in t.items()}
        qq.setdefault("ur", None)
        n = len(qq["u"]); bi = np.full(n, -1, np.int32); bdist = np.full(n, 256, np.int32); bd = Bounds(*bounds)
        check(self._lib.ivf_fuse_candidates(ptr(k), ptr(d), ptr(ur), len(k), C.byref(bd), ptr(sg), len(sg) if gate else 0, n, ptr(qq["u"]),
                                            ptr(qq["v"]), ptr(qq["ur"]), ptr(qq["radius"]), ptr(qq["level"]), ptr(qq["desc"]),
                                            ptr(qq["valid"]), ptr(bi), ptr(bdist), self.device_id))
        return bi, bdist

    def SearchBySim3(self, kps1, desc1, bounds1, kps2, desc2, bounds2, q12, q21):
        """SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:1145-1254) on the projected map points:
        q12 / q21 have u, v, radius, level, desc, valid per keypoint slot of KF1 / KF2.  Returns (matches12, nFound)."""
        k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, desc=np.uint8, valid=np.uint8)
        a = {x: np.ascontiguousarray(q12[x], y) for x, y in t.items()}; b = {x: np.ascontiguousarray(q21[x], y) for x, y in t.items()}
        if len(a["u"]) != len(k1) or len(b["u"]) != len(k2):
            raise AssertionError("one query slot per keypoint of each keyframe")
        m = np.full(len(k1), -1, np.int32); nf = C.c_int(0); bd1 = Bounds(*bounds1); bd2 = Bounds(*bounds2)
        check(self._lib.ivf_search_by_sim3(ptr(k1), ptr(d1), len(k1), C.byref(bd1), ptr(k2), ptr(d2), len(k2), C.byref(bd2),
                                           ptr(a["u"]), ptr(a["v"]), ptr(a["radius"]), ptr(a["level"]), ptr(a["desc"]), ptr(a["valid"]),
                                           ptr(b["u"]), ptr(b["v"]), ptr(b["radius"]), ptr(b["level"]), ptr(b["desc"]), ptr(b["valid"]),
                                           ptr(m), C.byref(nf), self.device_id))
        return m, nf.value

    @staticmethod
    def _csr(fv):
        nodes = sorted(fv)
        start = np.zeros(len(nodes) + 1, np.int32); idx = []
        for k, nd in enumerate(nodes):
            idx.extend(fv[nd]); start[k + 1] = len(idx)
        return np.array(nodes, np.int32), start, np.array(idx, np.int32)

    def SearchByBoW(self, kf_kps, kf_desc, kf_has_map_point, kf_feat_vec, f_kps, f_desc, f_feat_vec):
        """SearchByBoW(KeyFrame* pKF, Frame &F, vpMapPointMatches) (ORBmatcher.cc:165-294).  *_feat_vec: the
        DBoW2::FeatureVector as {node id: [feature indices]}.  Returns (for every keypoint of F the index of the keyframe
        keypoint whose map point it receives, or -1; nmatches)."""
        kk = np.ascontiguousarray(kf_kps, KP_DTYPE); kd = np.ascontiguousarray(kf_desc, np.uint8)
        hm = np.ascontiguousarray(kf_has_map_point, np.uint8)
        fk = np.ascontiguousarray(f_kps, KP_DTYPE); fd = np.ascontiguousarray(f_desc, np.uint8)
        kn, ks, ki = self._csr(kf_feat_vec); fn, fs, fi = self._csr(f_feat_vec)
        m = np.full(len(fk), -1, np.int32); nm = C.c_int(0)
        check(self._lib.ivf_search_by_bow(ptr(kk), ptr(kd), ptr(hm), len(kk), ptr(kn), ptr(ks), ptr(ki), len(kn), ptr(fk), ptr(fd),
                                          len(fk), ptr(fn), ptr(fs), ptr(fi), len(fn), self.mfNNratio, int(self.mbCheckOrientation),
                                          ptr(m), C.byref(nm), self.device_id))
        return m, nm.value

    def SearchByBoWKeyFrames(self, kps1, desc1, has_map_point1, feat_vec1, kps2, desc2, has_map_point2, feat_vec2):
        """SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12) (ORBmatcher.cc:528-661).  Returns (matches12 = index in
        KF2 per KF1 keypoint or -1, nmatches)."""
        k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        h1 = np.ascontiguousarray(has_map_point1, np.uint8); h2 = np.ascontiguousarray(has_map_point2, np.uint8)
        a, b, c = self._csr(feat_vec1); e, f, g = self._csr(feat_vec2)
        m = np.full(len(k1), -1, np.int32); nm = C.c_int(0)
        check(self._lib.ivf_search_by_bow_keyframes(ptr(k1), ptr(d1), ptr(h1), len(k1), ptr(a), ptr(b), ptr(c), len(a), ptr(k2), ptr(d2),
                                                    ptr(h2), len(k2), ptr(e), ptr(f), ptr(g), len(e), self.mfNNratio,
                                                    int(self.mbCheckOrientation), ptr(m), C.byref(nm), self.device_id))
        return m, nm.value

    def SearchForTriangulation(self, kps1, desc1, has_map_point1, stereo1, feat_vec1, kps2, desc2, has_map_point2, stereo2, feat_vec2,
                               F12, ex, ey, mvScaleFactors2, mvLevelSigma2_2, bOnlyStereo=False):
        """SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) (ORBmatcher.cc:663-829).  Returns
        (matches12, nmatches); vMatchedPairs = [(i, matches12[i]) for i where matches12[i] >= 0]."""
        k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        h1 = np.ascontiguousarray(has_map_point1, np.uint8); h2 = np.ascontiguousarray(has_map_point2, np.uint8)
        s1 = np.ascontiguousarray(stereo1, np.uint8); s2 = np.ascontiguousarray(stereo2, np.uint8)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sc = np.ascontiguousarray(mvScaleFactors2, np.float32); sg = np.ascontiguousarray(mvLevelSigma2_2, np.float32)
        a, b, c = self._csr(feat_vec1); e, f, g = self._csr(feat_vec2)
        m = np.full(len(k1), -1, np.int32); nm = C.c_int(0)
        check(self._lib.ivf_search_for_triangulation(ptr(k1), ptr(d1), ptr(h1), ptr(s1), len(k1), ptr(a), ptr(b), ptr(c), len(a),
                                                     ptr(k2), ptr(d2), ptr(h2), ptr(s2), len(k2), ptr(e), ptr(f), ptr(g), len(e),
                                                     ptr(F), float(ex), float(ey), ptr(sc), ptr(sg), min(len(sc), len(sg)),
                                                     int(bool(bOnlyStereo)), int(self.mbCheckOrientation), ptr(m), C.byref(nm),
                                                     self.device_id))
        return m, nm.value

    def SearchByProjectionReloc(self, cur_kps, cur_desc, bounds, q, ORBdist, cur_assign=None):
        """SearchByProjection(CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1520-1652): q has u, v,
        radius, level, angle, desc, valid.  Returns (CurrentFrame.mvpMapPoints as query indices, nmatches)."""
        k = np.ascontiguousarray(cur_kps, KP_DTYPE); d = np.ascontiguousarray(cur_desc, np.uint8)
        a = np.full(len(k), -1, np.int32) if cur_assign is None else np.ascontiguousarray(cur_assign, np.int32).copy()
        t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, angle=np.float32, desc=np.uint8, valid=np.uint8)
        qq = {x: np.ascontiguousarray(q[x], y) for x, y in t.items()}
        nm = C.c_int(0); bd = Bounds(*bounds)
        check(self._lib.ivf_search_by_projection_reloc(ptr(k), ptr(d), len(k), C.byref(bd), len(qq["u"]), ptr(qq["u"]), ptr(qq["v"]),
                                                       ptr(qq["radius"]), ptr(qq["level"]), ptr(qq["angle"]), ptr(qq["desc"]),
                                                       ptr(qq["valid"]), int(ORBdist), int(self.mbCheckOrientation), ptr(a),
                                                       C.byref(nm), self.device_id))
        return a, nm.value


def ComputeDistinctiveDescriptors(vDescriptors, device_id=0):
    """MapPoint::ComputeDistinctiveDescriptors (ORB/src/MapPoint.cc:247-312) on the observed descriptors [n,32]:
    returns (index of the descriptor to keep, its median distance to the others)."""
    lib = _lib.load()
    d = np.ascontiguousarray(vDescriptors, np.uint8).reshape(-1, 32)
    if len(d) == 0:
        raise AssertionError("vDescriptors is empty (the reference returns before this point)")
    bi = C.c_int(0); bm = C.c_int(0)
    check(lib.ivf_distinctive_descriptor(ptr(d), len(d), C.byref(bi), C.byref(bm), device_id))
    return bi.value, bm.value


class ORBVocabulary:
    """DBoW2 ORB vocabulary on the device (TemplatedVocabulary<FORB::TDescriptor, FORB>): transform() as
    Frame::ComputeBoW / KeyFrame::ComputeBoW call it (ORB/src/Frame.cc:683-694)."""

    def __init__(self, child_start, child, node_desc, node_word, node_weight, depth_L, device_id=0):
        self._lib = _lib.load()
        cs = np.ascontiguousarray(child_start, np.int32); ch = np.ascontiguousarray(child, np.int32)
        nd = np.ascontiguousarray(node_desc, np.uint8).reshape(-1, 32)
        wd = np.ascontiguousarray(node_word, np.int32); wt = np.ascontiguousarray(node_weight, np.float64)
        if not (len(cs) == len(nd) + 1 == len(wd) + 1 == len(wt) + 1):
            raise AssertionError("one child_start / descriptor / word id / weight entry per node")
        h = C.c_void_p()
        check(self._lib.ivf_vocabulary_create(len(nd), ptr(cs), ptr(ch), ptr(nd), ptr(wd), ptr(wt), int(depth_L), device_id, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ivf_vocabulary_destroy(self._h)
            self._h = None

    def transform_features(self, descriptors, levelsup=4):
        """Per descriptor (word id, node id at level L - levelsup, weight) (TemplatedVocabulary.h:1217-1259)."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32); n = len(d)
        wid = np.zeros(n, np.int32); nid = np.zeros(n, np.int32); wt = np.zeros(n, np.float64)
        check(self._lib.ivf_bow_transform(self._h, ptr(d), n, int(levelsup), ptr(wid), ptr(nid), ptr(wt)))
        return wid, nid, wt

    def transform(self, descriptors, levelsup=4):
        """transform(features, BowVector&, FeatureVector&, levelsup) (:1126-1204): returns (mBowVec as {word: value},
        mFeatVec as {node: [feature indices]})."""
        wid, nid, wt = self.transform_features(descriptors, levelsup)
        n = len(wid); cap = max(n, 1)
        bw = np.zeros(cap, np.int32); bv = np.zeros(cap, np.float64); fn = np.zeros(cap, np.int32)
        fs = np.zeros(cap + 1, np.int32); fi = np.zeros(cap, np.int32); nb = C.c_int(0); nf = C.c_int(0)
        check(self._lib.ivf_bow_vectors(ptr(wid), ptr(nid), ptr(wt), n, ptr(bw), ptr(bv), cap, C.byref(nb), ptr(fn), ptr(fs), ptr(fi),
                                        cap, C.byref(nf)))
        bow = {int(bw[k]): float(bv[k]) for k in range(nb.value)}
        fv = {int(fn[k]): [int(x) for x in fi[fs[k]:fs[k + 1]]] for k in range(nf.value)}
        return bow, fv


class DeviceFrame:
    """A frame whose mvKeysUn / mDescriptors / mGrid live on the device (Frame::AssignFeaturesToGrid, ORB/src/Frame.cc:415-430):
    SearchByProjection against it runs the windows and distances on the GPU, the greedy replay on the host."""

    def __init__(self, kps, desc, uright, bounds, device_id=0):
        self._lib = _lib.load()
        k = np.ascontiguousarray(kps, KP_DTYPE); d = np.ascontiguousarray(desc, np.uint8); u = np.ascontiguousarray(uright, np.float32)
        self.n = len(k)
        h = C.c_void_p(); bd = Bounds(*bounds)
        check(self._lib.ivf_frame_create(ptr(k), ptr(d), ptr(u), self.n, C.byref(bd), device_id, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ivf_frame_destroy(self._h)
            self._h = None

    def grid(self):
        """(cell_start [64*48+1], cell_index [n]) as built on the device; cell = ix*48 + iy."""
        st = np.zeros(64 * 48 + 1, np.int32); ix = np.zeros(max(self.n, 1), np.int32)
        check(self._lib.ivf_frame_grid(self._h, ptr(st), ptr(ix)))
        return st, ix[:self.n]

    def SearchByProjection(self, q, check_orientation=True, cur_assign=None):
        """Same queries and results as ORBmatcher.SearchByProjection(cur_kps, cur_desc, cur_uright, bounds, q)."""
        n_q = len(q["u"])
        assign = np.full(self.n, -1, np.int32) if cur_assign is None else np.ascontiguousarray(cur_assign, np.int32).copy()
        types = dict(u=np.float32, v=np.float32, ur=np.float32, radius=np.float32, min_level=np.int32,
                     max_level=np.int32, angle=np.float32, desc=np.uint8, valid=np.uint8, blocks=np.uint8)
        qq = {k: np.ascontiguousarray(q[k], t) for k, t in types.items()}
        nm = C.c_int(0)
        check(self._lib.ivf_frame_search_by_projection(self._h, n_q, ptr(qq["u"]), ptr(qq["v"]), ptr(qq["ur"]), ptr(qq["radius"]),
                                                       ptr(qq["min_level"]), ptr(qq["max_level"]), ptr(qq["angle"]), ptr(qq["desc"]),
                                                       ptr(qq["valid"]), ptr(qq["blocks"]), int(bool(check_orientation)), ptr(assign),
                                                       C.byref(nm)))
        return assign, nm.value

    def SearchByProjectionMapPoints(self, q, nn_ratio=0.6, cur_assign=None):
        """Same queries and results as ORBmatcher(nn_ratio).SearchByProjectionMapPoints(cur_kps, cur_desc, cur_uright, bounds, q)."""
        n_q = len(q["u"])
        assign = np.full(self.n, -1, np.int32) if cur_assign is None else np.ascontiguousarray(cur_assign, np.int32).copy()
        types = dict(u=np.float32, v=np.float32, ur=np.float32, radius=np.float32, level=np.int32, desc=np.uint8,
                     valid=np.uint8, blocks=np.uint8)
        qq = {k: np.ascontiguousarray(q[k], t) for k, t in types.items()}
        nm = C.c_int(0)
        check(self._lib.ivf_frame_search_map_points(self._h, n_q, ptr(qq["u"]), ptr(qq["v"]), ptr(qq["ur"]), ptr(qq["radius"]),
                                                    ptr(qq["level"]), ptr(qq["desc"]), ptr(qq["valid"]), ptr(qq["blocks"]),
                                                    float(nn_ratio), ptr(assign), C.byref(nm)))
        return assign, nm.value

    # --- frames made straight from a front-end batch, and the remaining window searches on a resident frame
    @classmethod
    def from_frontend(cls, fe, pair, side=0, bounds=None, age=0):
        """Resident frame of one image of a StereoFrontend batch: device-to-device, only the keypoint count is read back."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        bd = Bounds(*(bounds if bounds is not None else (0.0, 0.0, float(fe.width), float(fe.height))))
        h = C.c_void_p()
        check(self._lib.ivf_frame_create_from_frontend(fe._h, int(age), int(pair), int(side), C.byref(bd), C.byref(h)))
        self._h = h
        self.n = int(self._lib.ivf_frame_count(self._h))     # Frame::N: every keypoint, also those PosInGrid leaves out of the grid
        if self.n < 0:
            check(self.n)
        return self

    @staticmethod
    def _q(q, types):
        return {k: np.ascontiguousarray(q[k], t) for k, t in types.items()}

    def SearchKeyFramePoints(self, q, matched=None, n=None):
        """ivf_frame_search_keyframe_points: q = u, v, radius, level, desc, valid."""
        n = self.n if n is None else n
        m = np.full(n, -1, np.int32) if matched is None else np.ascontiguousarray(matched, np.int32).copy()
        qq = self._q(q, dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, desc=np.uint8, valid=np.uint8))
        nm = C.c_int(0)
        check(self._lib.ivf_frame_search_keyframe_points(self._h, len(qq["u"]), ptr(qq["u"]), ptr(qq["v"]), ptr(qq["radius"]), ptr(qq["level"]),
                                                         ptr(qq["desc"]), ptr(qq["valid"]), ptr(m), C.byref(nm)))
        return m, nm.value

    def FuseCandidates(self, inv_level_sigma2, q):
        """ivf_frame_fuse_candidates: q = u, v, ur, radius, level, desc, valid -> (best_idx, best_dist)."""
        gate = inv_level_sigma2 is not None
        sg = np.ascontiguousarray(inv_level_sigma2, np.float32) if gate else None
        t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, desc=np.uint8, valid=np.uint8)
        if gate:
            t["ur"] = np.float32
        qq = self._q(q, t); qq.setdefault("ur", None)
        nq = len(qq["u"]); bi = np.full(nq, -1, np.int32); bd = np.full(nq, 256, np.int32)
        check(self._lib.ivf_frame_fuse_candidates(self._h, ptr(sg), 0 if sg is None else len(sg), nq, ptr(qq["u"]), ptr(qq["v"]), ptr(qq["ur"]),
                                                  ptr(qq["radius"]), ptr(qq["level"]), ptr(qq["desc"]), ptr(qq["valid"]), ptr(bi), ptr(bd)))
        return bi, bd

    def SearchByProjectionReloc(self, q, orb_dist, check_orientation=True, cur_assign=None, n=None):
        n = self.n if n is None else n
        a = np.full(n, -1, np.int32) if cur_assign is None else np.ascontiguousarray(cur_assign, np.int32).copy()
        qq = self._q(q, dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, angle=np.float32, desc=np.uint8, valid=np.uint8))
        nm = C.c_int(0)
        check(self._lib.ivf_frame_search_by_projection_reloc(self._h, len(qq["u"]), ptr(qq["u"]), ptr(qq["v"]), ptr(qq["radius"]), ptr(qq["level"]),
                                                             ptr(qq["angle"]), ptr(qq["desc"]), ptr(qq["valid"]), int(orb_dist),
                                                             int(bool(check_orientation)), ptr(a), C.byref(nm)))
        return a, nm.value

    def SearchBySim3(self, other, q12, q21, n1=None):
        n1 = self.n if n1 is None else n1
        t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, desc=np.uint8, valid=np.uint8)
        a = self._q(q12, t); b = self._q(q21, t)
        m = np.full(max(n1, 1), -1, np.int32); nf = C.c_int(0)
        check(self._lib.ivf_frame_search_by_sim3(self._h, other._h, ptr(a["u"]), ptr(a["v"]), ptr(a["radius"]), ptr(a["level"]), ptr(a["desc"]),
                                                 ptr(a["valid"]), ptr(b["u"]), ptr(b["v"]), ptr(b["radius"]), ptr(b["level"]), ptr(b["desc"]),
                                                 ptr(b["valid"]), ptr(m), C.byref(nf)))
        return m[:n1], nf.value
