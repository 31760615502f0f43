"""CPU checks of the tracker's oracle (oracle/projection_oracle.py): the checker of ivf_tracker_run / ivf_tracker_search_local is itself
pinned here against hand-computed cases and against an independent route through the older, pre-projected entry points.
  Frame::isInFrustum            ORB/src/Frame.cc:557-613
  MapPoint::PredictScale        ORB/src/MapPoint.cc:407-422 (logf: DESIGN.md A-12)
  Tracking::UpdateLastFrame     ORB/src/Tracking.cc:1256-1300 (point selection)
  ORBmatcher::SearchByProjection(F, vpMapPoints, th)   ORB/src/ORBmatcher.cc:45-135
"""
import math
import os
import sys

import numpy as np

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import projection_oracle as PO  # noqa: E402

F = np.float32


def scale_table(n=8, sf=1.2):
    s = [F(1.0)]
    for _ in range(1, n):
        s.append(F(np.float64(s[-1]) * np.float64(sf)))
    return np.array(s, F)


def frame(T=None, n=0):
    sc = scale_table()
    kps = np.zeros(n, O.KP_DTYPE)
    return dict(kps=kps, desc=np.zeros((n, 32), np.uint8), uright=np.full(n, -1, F), depth=np.full(n, -1, F),
                T=np.eye(4, dtype=F) if T is None else T, scale=sc, logScale=F(O.lib.orc_logf(float(sc[1]))), fx=F(500.0), fy=F(500.0),
                cx=F(320.0), cy=F(240.0), mbf=F(40.0), mb=F(0.08), bounds=(0.0, 0.0, 640.0, 480.0))


def point(pos, normal=(0, 0, 1), lo=1.0, hi=20.0):
    return dict(pos=np.array(pos, F), normal=np.array(normal, F), minDist=F(lo), maxDist=F(hi), desc=np.zeros(32, np.uint8), skip=False, nObs=1)


def test_is_in_frustum_hand_cases():
    fr = frame()
    # on the optical axis, 10 m ahead, normal towards +z (the camera looks along +z from the origin: viewing ray = +z)
    t = PO.is_in_frustum(O, fr, point((0, 0, 10)))
    assert t is not None and t["projX"] == F(320.0) and t["projY"] == F(240.0)
    assert t["projXR"] == F(F(320.0) - F(F(40.0) * F(0.1))) and t["viewCos"] == F(1.0)
    # mfMaxDistance 20, distance 10: ratio 2 -> ceil(logf(2) / logf(1.2)) = ceil(3.80) = 4
    assert t["trackLevel"] == 4
    assert PO.is_in_frustum(O, fr, point((0, 0, -1))) is None                       # behind the camera (:571)
    assert PO.is_in_frustum(O, fr, point((10, 0, 10))) is None                      # u = 820 > mnMaxX (:579)
    assert PO.is_in_frustum(O, fr, point((0, -10, 10))) is None                     # v < mnMinY (:581)
    assert PO.is_in_frustum(O, fr, point((0, 0, 25))) is None                       # 25 > 1.2 * 20 (:590)
    assert PO.is_in_frustum(O, fr, point((0, 0, 24))) is not None                   # 24 <= 1.2 * 20: the bound is the scaled one
    assert PO.is_in_frustum(O, fr, point((0, 0, 0.79))) is None                     # 0.79 < 0.8 * 1
    assert PO.is_in_frustum(O, fr, point((0, 0, 10), normal=(1, 0, 0))) is None     # viewCos 0 < 0.5 (:598)
    n = (math.sin(math.radians(59)), 0.0, math.cos(math.radians(59)))
    assert PO.is_in_frustum(O, fr, point((0, 0, 10), normal=n)) is not None         # cos 59 deg = 0.515
    n = (math.sin(math.radians(61)), 0.0, math.cos(math.radians(61)))
    assert PO.is_in_frustum(O, fr, point((0, 0, 10), normal=n)) is None             # cos 61 deg = 0.485
    # a translated camera: Tcw = [I | t], camera centre at -t
    T = np.eye(4, dtype=F); T[:3, 3] = (1.0, 0.0, -2.0)
    t2 = PO.is_in_frustum(O, frame(T), point((0, 0, 10)))
    assert t2 is not None and t2["projX"] == F(F(F(F(500.0) * F(1.0)) * F(F(1.0) / F(8.0))) + F(320.0))


def test_predict_scale_uses_float_log_and_clamps():
    fr = frame()
    sc = fr["scale"]
    for lv in range(8):
        for d in (0.5, 3.0, 17.0):
            # far from the knife edge: between two powers of 1.2
            mx = F(F(d) * sc[lv] * F(1.09))
            assert PO.predict_scale_f32(O, mx, F(d), fr) == min(lv + 1, 7)
    assert PO.predict_scale_f32(O, F(1.0), F(5.0), fr) == 0                          # ratio < 1: negative -> 0
    assert PO.predict_scale_f32(O, F(1000.0), F(1.0), fr) == 7                       # -> nlevels - 1
    # the float path and a double path differ exactly on the knife edge: at least one (distance, level) pair shows it
    differs = 0
    for lv in range(1, 8):
        for d in np.linspace(0.7, 40.0, 400).astype(F):
            mx = F(d * sc[lv])
            a = PO.predict_scale_f32(O, mx, d, fr)
            r = float(F(mx / d))
            b = min(max(int(math.ceil(math.log(r) / float(fr["logScale"]))), 0), 7)
            differs += a != b
            assert a in (lv, min(lv + 1, 7)) and b in (lv, min(lv + 1, 7))
    assert differs > 0


def test_update_last_frame_point_rule():
    """Tracking.cc:1256-1300: sorted by depth, every point closer than mThDepth and at least the 100 closest."""
    rng = np.random.default_rng(0)
    n = 400
    depth = np.where(rng.random(n) < 0.8, rng.uniform(1, 80, n), -1).astype(F)
    last = dict(kps=np.zeros(n, O.KP_DTYPE), depth=depth)
    pos = np.sort(depth[depth > 0])
    for th in (0.0, 5.0, 30.0, 200.0):
        sel = PO.update_last_frame_points(last, th)
        assert sel == sorted(sel) and all(depth[i] > 0 for i in sel)
        if th <= 0:
            assert len(sel) == len(pos)
            continue
        n_close = int((pos <= th).sum())
        # the walk stops after the first point beyond th once more than 100 were taken (that point is included)
        expect = len(pos) if n_close >= len(pos) else max(n_close + 1, 101)
        assert len(sel) == min(expect, len(pos)), (th, len(sel), n_close)
        assert set(sel) == set(np.argsort(np.where(depth > 0, depth, np.inf), kind="stable")[:len(sel)].tolist())


def test_search_local_points_frame_equals_the_pre_projected_route():
    """search_local_points_frame (projection inside) == search_local_points (the caller projects: what the per-call adapters'
    oracle has used since r02) on the same scene."""
    rng = np.random.default_rng(4)
    n = 300
    kps = np.zeros(n, O.KP_DTYPE)
    kps["x"] = rng.uniform(20, 620, n).astype(F); kps["y"] = rng.uniform(20, 460, n).astype(F); kps["octave"] = rng.integers(0, 8, n)
    desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
    depth = rng.uniform(2, 30, n).astype(F)
    fr = frame(n=n)
    fr.update(kps=kps, desc=desc, depth=depth, uright=(kps["x"] - fr["mbf"] / depth).astype(F))
    pts = []
    for i in range(n):
        if rng.random() < 0.3:
            continue
        z = depth[i]
        pos = np.array([(kps["x"][i] - fr["cx"]) * z / fr["fx"], (kps["y"][i] - fr["cy"]) * z / fr["fy"], z], F)
        dist = F(np.sqrt((pos.astype(np.float64) ** 2).sum()))
        d = desc[i].copy(); d[rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
        pts.append(dict(pos=pos, normal=(pos / dist).astype(F), minDist=F(dist * fr["scale"][kps["octave"][i]] / fr["scale"][-1]),
                        maxDist=F(dist * fr["scale"][kps["octave"][i]]), desc=d, skip=bool(rng.random() < 0.1), nObs=int(rng.random() < 0.7)))
    occ = rng.random(n) < 0.15
    for th, ratio in ((1.0, 0.8), (3.0, 0.6)):
        nm, got = PO.search_local_points_frame(O, fr, pts, occ, F(th), F(ratio))
        # the older route: the caller runs isInFrustum and hands the tracking fields over
        pool = []
        for p in pts:
            tr = None if p["skip"] else PO.is_in_frustum(O, fr, p)
            q = dict(desc=p["desc"], nObs=p["nObs"], bad=False, inView=tr is not None)
            if tr is not None:
                q.update(tr)
            pool.append(q)
        cur_mps = [len(pool) if o else -1 for o in occ]                     # an occupant WITH observations (a pool entry of its own)
        pool.append(dict(nObs=1))
        nm2, out2 = PO.search_local_points(O, fr, pool, cur_mps, list(range(len(pts))), F(th), F(ratio))
        exp = np.array([m if (m >= 0 and m < len(pts) and not occ[i]) else -1 for i, m in enumerate(out2)], np.int32)
        assert nm == nm2 and nm > 20
        assert np.array_equal(got, exp)
