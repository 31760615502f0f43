"""projection_oracle.py -- CPU ORACLE (TEST INFRASTRUCTURE ONLY) of the PROJECTION / BOOKKEEPING loops that surround the
window searches of ORB_SLAM2::ORBmatcher: numpy restatement, plain Python loops (small cases), of

  SearchByProjection(Frame&, const Frame&, th, bMono)            ORB/src/ORBmatcher.cc:1372-1518
  SearchByProjection(Frame&, vector<MapPoint*>&, th)             ORB/src/ORBmatcher.cc:45-135
  SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, d)    ORB/src/ORBmatcher.cc:1520-1652
  SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)    ORB/src/ORBmatcher.cc:296-404
  Fuse(KeyFrame*, vpMapPoints, th)                               ORB/src/ORBmatcher.cc:831-981

The window search + greedy replay of each is the C oracle's (oracle/ivf_oracle.c through tests/oracle_lib.py); this file adds
what the reference does around it with cv::Mat expressions.  Those expressions' arithmetic lives in un-vendored OpenCV:
PARITY UNPINNED, frozen as DESIGN.md A-11 (gemm: double accumulation then one narrowing; norm / dot in double; scalar scale in
float).  Independent of include/ivfront_orbslam.hpp (different language, different structure): the adapter test compares the two.
"""
import math

import numpy as np

F = np.float32
D = np.float64


def mul_add(R, p, t):
    """cv::gemm A*B + C for CV_32F: (float)(sum (double)a*(double)b + (double)c)."""
    R = R.astype(D); p = p.astype(D); t = t.astype(D)
    return np.array([R[i, 0] * p[0] + R[i, 1] * p[1] + R[i, 2] * p[2] + t[i] for i in range(3)], D).astype(F)


def neg_rt_mul(R, t):
    R = R.astype(D); t = t.astype(D)
    return np.array([-(R[0, j] * t[0] + R[1, j] * t[1] + R[2, j] * t[2]) for j in range(3)], D).astype(F)


def norm(v):
    v = v.astype(D)
    return math.sqrt(float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))


def dot(a, b):
    a = a.astype(D); b = b.astype(D)
    return float(a[0] * b[0] + a[1] * b[1] + a[2] * b[2])


def predict_scale(mp, dist, frame):
    """MapPoint::PredictScale (ORB/src/MapPoint.cc:386-420)."""
    ratio = F(mp["maxDist"]) / F(dist)
    n = int(math.ceil(math.log(float(ratio)) / float(frame["logScale"])))        # float log in the source: the quotient is far from
    return min(max(n, 0), len(frame["scale"]) - 1)                                # an integer in every test scenario


def _occupancy(mps, pool, any_occupant):
    a = np.full(len(mps), -1, np.int32)
    for i, m in enumerate(mps):
        if m >= 0 and (any_occupant or pool[m]["nObs"] > 0):
            a[i] = -2
    return a


def _q(rows, keys):
    return {k: np.array([r[k] for r in rows]) if rows else np.zeros((0,) + ((32,) if k == "desc" else ())) for k in keys}


def search_cur_last(O, cur, last, pool, cur_mps, th, mono):
    Rcw, tcw = cur["T"][:3, :3], cur["T"][:3, 3]
    twc = neg_rt_mul(Rcw, tcw)
    tlc = mul_add(last["T"][:3, :3], twc, last["T"][:3, 3])
    fwd = bool(tlc[2] > F(cur["mb"])) and not mono
    bwd = bool(-tlc[2] > F(cur["mb"])) and not mono
    rows, src = [], []
    for i in range(len(last["kps"])):
        m = last["mps"][i]
        if m < 0 or last["outlier"][i]:
            continue
        x = mul_add(Rcw, pool[m]["pos"], tcw)
        invz = F(D(1.0) / D(x[2]))
        if invz < 0:
            continue
        u = F(F(F(cur["fx"]) * x[0]) * invz) + F(cur["cx"]); v = F(F(F(cur["fy"]) * x[1]) * invz) + F(cur["cy"])
        u = F(u); v = F(v)
        if u < cur["bounds"][0] or u > cur["bounds"][2] or v < cur["bounds"][1] or v > cur["bounds"][3]:
            continue
        o = int(last["kps"]["octave"][i])
        radius = F(F(th) * cur["scale"][o])
        lo, hi = (o, -1) if fwd else (0, o) if bwd else (o - 1, o + 1)
        ur = F(u - F(F(cur["mbf"]) * invz))
        rows.append(dict(u=u, v=v, ur=ur, radius=radius, min_level=lo, max_level=hi, angle=last["kps"]["angle"][i],
                         desc=pool[m]["desc"], valid=1, blocks=1 if pool[m]["nObs"] > 0 else 0))
        src.append(i)
    out = list(cur_mps)
    if not rows:
        return 0, out
    q = _q(rows, ("u", "v", "ur", "radius", "min_level", "max_level", "angle", "desc", "valid", "blocks"))
    occ = _occupancy(cur_mps, pool, False)
    a0, _ = O.search_by_projection(cur["kps"], cur["desc"], cur["uright"], cur["bounds"], q, False, occ)
    a1, nm = O.search_by_projection(cur["kps"], cur["desc"], cur["uright"], cur["bounds"], q, True, occ)
    for i in range(len(out)):
        if a1[i] >= 0:
            out[i] = last["mps"][src[a1[i]]]
        elif a0[i] >= 0:
            out[i] = -1                     # matched, then dropped by the rotation filter: NULL (:1504)
    return nm, out


def search_local_points(O, cur, pool, cur_mps, local, th, nn_ratio):
    rows, src = [], []
    for k, m in enumerate(local):
        p = pool[m]
        if not p["inView"] or p["bad"]:
            continue
        r = F(2.5) if F(p["viewCos"]) > 0.998 else F(4.0)
        if th != 1.0:
            r = F(r * F(th))
        lv = p["trackLevel"]
        rows.append(dict(u=p["projX"], v=p["projY"], ur=p["projXR"], radius=F(r * cur["scale"][lv]), level=lv, desc=p["desc"], valid=1,
                         blocks=1 if p["nObs"] > 0 else 0))
        src.append(m)
    out = list(cur_mps)
    if not rows:
        return 0, out
    q = _q(rows, ("u", "v", "ur", "radius", "level", "desc", "valid", "blocks"))
    a, nm = O.search_map_points(cur["kps"], cur["desc"], cur["uright"], cur["bounds"], q, nn_ratio, _occupancy(cur_mps, pool, False))
    for i in range(len(out)):
        if a[i] >= 0:
            out[i] = src[a[i]]
    return nm, out


def search_reloc(O, cur, kf, pool, cur_mps, found, th, orb_dist):
    Rcw, tcw = cur["T"][:3, :3], cur["T"][:3, 3]
    Ow = neg_rt_mul(Rcw, tcw)
    rows, src = [], []
    for i, m in enumerate(kf["mps"]):
        if m < 0 or pool[m]["bad"] or m in found:
            continue
        xw = pool[m]["pos"]
        x = mul_add(Rcw, xw, tcw)
        invz = F(D(1.0) / D(x[2]))
        u = F(F(F(F(cur["fx"]) * x[0]) * invz) + F(cur["cx"])); v = F(F(F(F(cur["fy"]) * x[1]) * invz) + F(cur["cy"]))
        if u < cur["bounds"][0] or u > cur["bounds"][2] or v < cur["bounds"][1] or v > cur["bounds"][3]:
            continue
        dist = F(norm((xw - Ow).astype(F)))
        if dist < F(F(0.8) * F(pool[m]["minDist"])) or dist > F(F(1.2) * F(pool[m]["maxDist"])):
            continue
        lv = predict_scale(pool[m], dist, cur)
        rows.append(dict(u=u, v=v, radius=F(F(th) * cur["scale"][lv]), level=lv, angle=kf["kps"]["angle"][i], desc=pool[m]["desc"], valid=1))
        src.append(m)
    out = list(cur_mps)
    if not rows:
        return 0, out
    q = _q(rows, ("u", "v", "radius", "level", "angle", "desc", "valid"))
    a, nm = O.search_by_projection_reloc(cur["kps"], cur["desc"], cur["bounds"], q, orb_dist, True, _occupancy(cur_mps, pool, True))
    for i in range(len(out)):
        if a[i] >= 0:
            out[i] = src[a[i]]
    return nm, out


def _project_kf(kf, Rcw, tcw, Ow, p, invz_float_one):
    """the common keyframe projection (:318-365, :859-905): returns (u, v, dist, p3Dc) or None."""
    xw = p["pos"]
    x = mul_add(Rcw, xw, tcw)
    if x[2] < 0.0:
        return None
    invz = F(F(1) / x[2]) if invz_float_one else F(D(1.0) / D(x[2]))
    u = F(F(F(kf["fx"]) * F(x[0] * invz)) + F(kf["cx"])); v = F(F(F(kf["fy"]) * F(x[1] * invz)) + F(kf["cy"]))
    b = kf["bounds"]
    if not (u >= b[0] and u < b[2] and v >= b[1] and v < b[3]):                     # KeyFrame::IsInImage
        return None
    PO = (xw - Ow).astype(F)
    dist = F(norm(PO))
    if dist < F(F(0.8) * F(p["minDist"])) or dist > F(F(1.2) * F(p["maxDist"])):
        return None
    if dot(PO, p["normal"]) < 0.5 * float(dist):
        return None
    return u, v, dist, x


def search_kf_sim3(O, kf, Scw, pool, points, th):
    s = Scw[:3, :3]
    scw = F(math.sqrt(dot(s[0], s[0])))
    a = F(D(1.0) / D(scw))
    Rcw = (s * a).astype(F); tcw = (Scw[:3, 3] * a).astype(F)
    Ow = neg_rt_mul(Rcw, tcw)
    rows, src = [], []
    for m in points:
        if pool[m]["bad"]:
            continue
        pr = _project_kf(kf, Rcw, tcw, Ow, pool[m], True)
        if pr is None:
            continue
        u, v, dist, _ = pr
        lv = predict_scale(pool[m], dist, kf)
        rows.append(dict(u=u, v=v, radius=F(F(th) * kf["scale"][lv]), level=lv, desc=pool[m]["desc"], valid=1))
        src.append(m)
    out = [-1] * len(kf["kps"])
    if not rows:
        return 0, out
    q = _q(rows, ("u", "v", "radius", "level", "desc", "valid"))
    a, nm = O.search_keyframe_points(kf["kps"], kf["desc"], kf["bounds"], q)
    for i in range(len(out)):
        if a[i] >= 0:
            out[i] = src[a[i]]
    return nm, out


def fuse(O, kf, pool, points, th):
    """returns (nFused, kf map points after, replacedBy per map point); mutates copies only."""
    Rcw, tcw, Ow = kf["T"][:3, :3], kf["T"][:3, 3], kf["Ow"]
    kf_mps = list(kf["mps"])
    in_kf = {m: i for i, m in enumerate(kf_mps) if m >= 0}
    nobs = {m: pool[m]["nObs"] for m in range(len(pool))}
    bad = {m: pool[m]["bad"] for m in range(len(pool))}
    replaced = [-1] * len(pool)
    rows, src = [], []
    for m in points:
        if bad[m] or m in in_kf:
            continue
        pr = _project_kf(kf, Rcw, tcw, Ow, pool[m], True)
        if pr is None:
            continue
        u, v, dist, x = pr
        invz = F(F(1) / x[2])
        lv = predict_scale(pool[m], dist, kf)
        rows.append(dict(u=u, v=v, ur=F(u - F(F(kf["mbf"]) * invz)), radius=F(F(th) * kf["scale"][lv]), level=lv, desc=pool[m]["desc"], valid=1))
        src.append(m)
    if not rows:
        return 0, kf_mps, replaced
    q = _q(rows, ("u", "v", "ur", "radius", "level", "desc", "valid"))
    best, _ = O.fuse_candidates(kf["kps"], kf["desc"], kf["uright"], kf["bounds"], kf["invSigma2"], q)
    n = 0
    for k, m in enumerate(src):
        if best[k] < 0:
            continue
        other = kf_mps[best[k]]
        if other >= 0:
            if not bad[other]:
                if nobs[other] > nobs[m]:
                    replaced[m] = other; bad[m] = True
                else:
                    replaced[other] = m; bad[other] = True
        else:
            nobs[m] += 1; kf_mps[best[k]] = m
        n += 1
    return n, kf_mps, replaced


# ---- r02: the remaining ORBmatcher methods (restated for tests/test_gpu_adapter.py) ----------------------------------
def feature_vector(desc):
    """the tests' stand-in for a DBoW2::FeatureVector: node = first descriptor byte & 31, features in index order."""
    fv = {}
    for i, d in enumerate(desc):
        fv.setdefault(int(d[0]) & 31, []).append(i)
    return fv


def epipole(kf1, kf2):
    """ORBmatcher.cc:670-676: C2 = R2w*Cw + t2w; ex = fx*C2x*invz + cx with invz = 1.0f / C2z (float)."""
    C2 = mul_add(kf2["T"][:3, :3], kf1["Ow"], kf2["T"][:3, 3])
    invz = F(F(1) / C2[2])
    return F(F(F(F(kf2["fx"]) * C2[0]) * invz) + F(kf2["cx"])), F(F(F(F(kf2["fy"]) * C2[1]) * invz) + F(kf2["cy"]))


def _sim3_side(pts, done, pool, Ra, ta, Rb, tb, target, fx, fy, cx, cy, th):
    """one direction of SearchBySim3 (:1193-1230 / :1273-1310): one query slot per keypoint of the source keyframe."""
    n = len(pts)
    q = dict(u=np.zeros(n, F), v=np.zeros(n, F), radius=np.zeros(n, F), level=np.zeros(n, np.int32), desc=np.zeros((n, 32), np.uint8),
             valid=np.zeros(n, np.uint8))
    b = target["bounds"]
    for i, m in enumerate(pts):
        if m < 0 or done[i] or pool[m]["bad"]:
            continue
        pa = mul_add(Ra, pool[m]["pos"], ta)
        pb = mul_add(Rb, pa, tb)
        if pb[2] < 0.0:
            continue
        invz = F(D(1.0) / D(pb[2]))
        u = F(F(F(fx) * F(pb[0] * invz)) + F(cx)); v = F(F(F(fy) * F(pb[1] * invz)) + F(cy))
        if not (u >= b[0] and u < b[2] and v >= b[1] and v < b[3]):
            continue
        dist = F(norm(pb))
        if dist < F(F(0.8) * F(pool[m]["minDist"])) or dist > F(F(1.2) * F(pool[m]["maxDist"])):
            continue
        lv = predict_scale(pool[m], dist, target)
        q["u"][i] = u; q["v"][i] = v; q["radius"][i] = F(F(th) * target["scale"][lv]); q["level"][i] = lv; q["desc"][i] = pool[m]["desc"]; q["valid"][i] = 1
    return q


def search_by_sim3(O, kf1, kf2, pool, pre12, s12, R12, t12, th):
    """ORBmatcher.cc:1145-1370 -> (nFound, vpMatches12 as pool indices)."""
    R12 = R12.astype(F); t12 = t12.astype(F)
    sR12 = (R12 * F(s12)).astype(F)
    sR21 = (R12.T * F(D(1.0) / D(F(s12)))).astype(F)
    t21 = np.array([-(D(sR21[i, 0]) * D(t12[0]) + D(sR21[i, 1]) * D(t12[1]) + D(sR21[i, 2]) * D(t12[2])) for i in range(3)], D).astype(F)
    m1, m2 = kf1["mps"], kf2["mps"]
    done1 = [p >= 0 for p in pre12]; done2 = [False] * len(m2)
    where2 = {}
    for i, m in enumerate(m2):
        if m >= 0:
            where2[m] = i                     # GetIndexInKeyFrame: the mock keeps the LAST slot a point was registered at
    for i, p in enumerate(pre12):
        if p >= 0 and p in where2:
            done2[where2[p]] = True
    fx, fy, cx, cy = kf1["fx"], kf1["fy"], kf1["cx"], kf1["cy"]                 # KF1's intrinsics in both directions (:1148-1151)
    q12 = _sim3_side(m1, done1, pool, kf1["T"][:3, :3], kf1["T"][:3, 3], sR21, t21, kf2, fx, fy, cx, cy, th)
    q21 = _sim3_side(m2, done2, pool, kf2["T"][:3, :3], kf2["T"][:3, 3], sR12, t12, kf1, fx, fy, cx, cy, th)
    m12, nf = O.search_by_sim3(kf1["kps"], kf1["desc"], kf1["bounds"], kf2["kps"], kf2["desc"], kf2["bounds"], q12, q21)
    out = list(pre12)
    for i in range(len(out)):
        if m12[i] >= 0:
            out[i] = m2[m12[i]]
    return nf, out


def fuse_sim3(O, kf, Scw, pool, points, th):
    """Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:983-1106) -> (nFused, vpReplacePoint, keyframe map points after)."""
    s = Scw[:3, :3]
    scw = F(math.sqrt(dot(s[0], s[0])))
    a = F(D(1.0) / D(scw))
    Rcw = (s * a).astype(F); tcw = (Scw[:3, 3] * a).astype(F)
    Ow = neg_rt_mul(Rcw, tcw)
    kf_mps = list(kf["mps"])
    found = set(m for m in kf_mps if m >= 0 and not pool[m]["bad"])
    rows, src = [], []
    for k, m in enumerate(points):
        if pool[m]["bad"] or m in found:
            continue
        pr = _project_kf(kf, Rcw, tcw, Ow, pool[m], False)
        if pr is None:
            continue
        u, v, dist, _ = pr
        lv = predict_scale(pool[m], dist, kf)
        rows.append(dict(u=u, v=v, radius=F(F(th) * kf["scale"][lv]), level=lv, desc=pool[m]["desc"], valid=1))
        src.append(k)
    repl = [-1] * len(points)
    if not rows:
        return 0, repl, kf_mps
    q = _q(rows, ("u", "v", "radius", "level", "desc", "valid"))
    best, _ = O.fuse_candidates(kf["kps"], kf["desc"], None, kf["bounds"], None, q)
    n = 0
    for j, k in enumerate(src):
        if best[j] < 0:
            continue
        other = kf_mps[best[j]]
        if other >= 0:
            if not pool[other]["bad"]:
                repl[k] = other
        else:
            kf_mps[best[j]] = points[k]
        n += 1
    return n, repl, kf_mps


def update_quality_scores(frame_mps, kp_quality, mp_quality):
    """ORBmatcher::UpdateQualityScores(Frame&) (:1108-1121)."""
    kq = np.array(kp_quality, F); mq = np.array(mp_quality, F)
    for i, m in enumerate(frame_mps):
        if m < 0:
            continue
        upd = min(mq[m], kq[i])
        if abs(F(upd - mq[m])) > F(0.01):
            mq[m] = upd
        kq[i] = upd
    return kq, mq


# ---- Tracking::TrackWithMotionModel, matcher part (ORB/src/Tracking.cc:1303-1330), on one (last, cur) frame pair ------------
def unproject_stereo(frame, i):
    """Frame::UnprojectStereo (ORB/src/Frame.cc:958-972): x = (u - cx) * z * invfx in float, then mRwc * x3Dc + mOw (gemm);
    mRwc = mRcw.t(), mOw = -mRcw.t() * mtcw (Frame::UpdatePoseMatrices, Frame.cc:549-555)."""
    z = F(frame["depth"][i])
    u = F(frame["kps"]["x"][i]); v = F(frame["kps"]["y"][i])
    invfx = F(F(1.0) / F(frame["fx"])); invfy = F(F(1.0) / F(frame["fy"]))          # Frame.cc:203-204
    x = F(F(F(u - F(frame["cx"])) * z) * invfx); y = F(F(F(v - F(frame["cy"])) * z) * invfy)
    Rcw, tcw = frame["T"][:3, :3], frame["T"][:3, 3]
    Ow = neg_rt_mul(Rcw, tcw)
    return mul_add(np.ascontiguousarray(Rcw.T), np.array([x, y, z], F), Ow)


def update_last_frame_points(last, th_depth):
    """Tracking::UpdateLastFrame's "visual odometry" points (Tracking.cc:1256-1300): stereo points sorted by (depth, index); every
    point is taken until one lies beyond mThDepth AND more than 100 have been taken.  Returns the selected keypoint indices in
    ascending index order.  th_depth <= 0: every stereo point."""
    idx = [(float(last["depth"][i]), i) for i in range(len(last["kps"])) if last["depth"][i] > 0]
    if th_depth <= 0:
        return [i for _, i in idx]
    idx.sort()
    out = []
    n_points = 0
    for z, i in idx:
        out.append(i); n_points += 1
        if z > th_depth and n_points > 100:
            break
    return sorted(out)


def track_with_motion_model_matches(O, cur, last, th, th_retry, retry_below, check_orientation=True, th_depth=0.0, points_block=True,
                                    point_flags=None, quality=None):
    """(nmatches, CurrentFrame.mvpMapPoints as last-frame keypoint indices or -1) of the matcher part of TrackWithMotionModel.
    point_flags (per last keypoint: bit 0 = has a map point, bit 1 = Observations() > 0) overrides th_depth / points_block.
    quality = (point_q per LAST keypoint, key_q per CURRENT keypoint), float32 arrays updated IN PLACE the way every
    SearchByProjection call ends under --ivslam_propagate_keyptqual (ORBmatcher.cc:1513-1515 -> UpdateQualityScores, :1108-1121)."""
    if point_flags is not None:
        sel = [i for i in range(len(last["kps"])) if last["depth"][i] > 0 and (int(point_flags[i]) & 1)]
        obs = {i: (int(point_flags[i]) >> 1) & 1 for i in sel}
    else:
        sel = update_last_frame_points(last, th_depth)
        obs = {i: (0 if th_depth > 0 else int(bool(points_block))) for i in sel}       # new VO points have no observations
    pool = [dict(pos=unproject_stereo(last, i), desc=last["desc"][i], nObs=obs[i]) for i in sel]
    lm = dict(last)
    mps = np.full(len(last["kps"]), -1, np.int64)
    for k, i in enumerate(sel):
        mps[i] = k
    lm["mps"] = mps; lm["outlier"] = np.zeros(len(last["kps"]), bool)
    cur_mps = [-1] * len(cur["kps"])                                                   # fill(mvpMapPoints, NULL) (Tracking.cc:1311)

    def one(th_):
        if check_orientation:
            return search_cur_last(O, cur, lm, pool, cur_mps, th_, False)
        return _search_cur_last_no_ori(O, cur, lm, pool, cur_mps, th_)
    def propagate(out_):
        if quality is not None:
            pq, kq = quality
            kq2, pq2 = update_quality_scores([sel[m] if m >= 0 else -1 for m in out_], kq, pq)
            kq[:] = kq2; pq[:] = pq2
    nm, out = one(th)
    propagate(out)
    if nm < retry_below:                                                               # Tracking.cc:1320-1330
        nm, out = one(th_retry)
        propagate(out)
    return nm, np.array([sel[m] if m >= 0 else -1 for m in out], np.int32)


def _search_cur_last_no_ori(O, cur, last, pool, cur_mps, th):
    """search_cur_last with mbCheckOrientation = false (no rotation histogram)."""
    import types
    shim = types.SimpleNamespace(search_by_projection=lambda k, d, u, b, q, chk, occ: O.search_by_projection(k, d, u, b, q, False, occ))
    return search_cur_last(shim, cur, last, pool, cur_mps, th, False)


# ---- Tracking::SearchLocalPoints from the projection on (ORB/src/Tracking.cc:2088-2132), one frame -----------------------------
def predict_scale_f32(O, max_dist, dist, frame):
    """MapPoint::PredictScale (ORB/src/MapPoint.cc:407-422) AS COMPILED: the unqualified log(float) resolves to std::log(float) = logf
    under the `using namespace std` of DBoW2/TemplatedVocabulary.h:36, the quotient by the float mfLogScaleFactor is a float and
    std::ceil(float) rounds it (glibc's logf restated in oracle/ivf_oracle.c:orc_logf)."""
    ratio = F(F(max_dist) / F(dist))
    q = F(F(O.lib.orc_logf(float(ratio))) / F(frame["logScale"]))
    n = int(math.ceil(float(q)))
    return min(max(n, 0), len(frame["scale"]) - 1)


def is_in_frustum(O, frame, mp, cos_limit=0.5):
    """Frame::isInFrustum (ORB/src/Frame.cc:557-613): None, or the tracking fields it leaves on the map point."""
    Rcw, tcw = frame["T"][:3, :3], frame["T"][:3, 3]
    P = np.asarray(mp["pos"], F)
    Pc = mul_add(Rcw, P, tcw)                                                          # :565
    if Pc[2] < F(0.0):
        return None
    with np.errstate(all="ignore"):
        invz = F(F(1.0) / Pc[2])
        u = F(F(F(F(frame["fx"]) * Pc[0]) * invz) + F(frame["cx"])); v = F(F(F(F(frame["fy"]) * Pc[1]) * invz) + F(frame["cy"]))
    minx, miny, maxx, maxy = [F(b) for b in frame["bounds"]]
    if u < minx or u > maxx or v < miny or v > maxy:
        return None
    PO = (P - neg_rt_mul(Rcw, tcw)).astype(F)                                          # P - mOw (:587)
    dist = F(norm(PO))                                                                 # cv::norm -> const float (:588)
    if dist < F(F(0.8) * F(mp["minDist"])) or dist > F(F(1.2) * F(mp["maxDist"])):     # MapPoint.cc:378-388, Frame.cc:590
        return None
    view_cos = F(dot(PO, np.asarray(mp["normal"], F)) / float(dist))                   # :596
    if view_cos < F(cos_limit):
        return None
    return dict(projX=u, projY=v, projXR=F(u - F(F(frame["mbf"]) * invz)), trackLevel=predict_scale_f32(O, mp["maxDist"], dist, frame),
                viewCos=view_cos)


def search_local_points_frame(O, cur, points, occupied, th, nn_ratio, cos_limit=0.5):
    """points: list of dict(pos, normal, minDist, maxDist, desc, skip, nObs) in mvpLocalMapPoints order; occupied: per keypoint, True =
    holds a map point with observations.  Returns (nmatches, per keypoint the index of the point it received in this call or -1)."""
    rows, src = [], []
    for k, p in enumerate(points):
        if p["skip"]:
            continue
        tr = is_in_frustum(O, cur, p, cos_limit)
        if tr is None:
            continue
        r = F(2.5) if float(tr["viewCos"]) > 0.998 else F(4.0)                         # RadiusByViewingCos (ORBmatcher.cc:137-143)
        if float(th) != 1.0:
            r = F(r * F(th))
        lv = tr["trackLevel"]
        rows.append(dict(u=tr["projX"], v=tr["projY"], ur=tr["projXR"], radius=F(r * F(cur["scale"][lv])), level=lv, desc=p["desc"], valid=1,
                         blocks=1 if p["nObs"] > 0 else 0))
        src.append(k)
    n = len(cur["kps"])
    out = np.full(n, -1, np.int32)
    if not rows:
        return 0, out
    q = _q(rows, ("u", "v", "ur", "radius", "level", "desc", "valid", "blocks"))
    occ = np.where(np.asarray(occupied, bool), -2, -1).astype(np.int32) if occupied is not None else np.full(n, -1, np.int32)
    a, nm = O.search_map_points(cur["kps"], cur["desc"], cur["uright"], cur["bounds"], q, nn_ratio, occ)
    for i in range(n):
        if a[i] >= 0:
            out[i] = src[a[i]]
    return nm, out
