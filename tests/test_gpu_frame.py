"""Device-resident frame: the 64x48 grid built on the device and SearchByProjection against it,
vs the oracle's AssignFeaturesToGrid / GetFeaturesInArea / SearchByProjection restatement."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from iv_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(iv, seed, n, w=640, h=240, radius=7.0, big=False):
    rng = np.random.default_rng(seed)
    img = synth.make_left(w, h, seed=seed, idx=1)
    g = iv.ORBextractor(n, 1.2, 8, 20, 7)
    kps, desc = g(img)
    nq = len(kps)
    qd = desc.copy()
    for i in range(nq):
        for bpos in rng.integers(0, 256, 5):
            qd[i, bpos // 8] ^= np.uint8(1 << (bpos % 8))
    oct_ = kps["octave"]; sc = g.GetScaleFactors()
    q = dict(u=kps["x"] + rng.uniform(-3, 3, nq).astype(np.float32), v=kps["y"] + rng.uniform(-3, 3, nq).astype(np.float32),
             ur=(kps["x"] - 20).astype(np.float32), radius=(radius * sc[oct_]).astype(np.float32),
             min_level=(oct_ - 1).astype(np.int32) if not big else np.full(nq, -1, np.int32),
             max_level=(oct_ + 1).astype(np.int32) if not big else np.full(nq, -1, np.int32),
             angle=(kps["angle"] + rng.choice([0, 0, 0, 90], nq)).astype(np.float32) % 360, desc=qd,
             valid=(rng.uniform(size=nq) > 0.1).astype(np.uint8), blocks=(rng.uniform(size=nq) > 0.2).astype(np.uint8))
    q["u"][:3] = [-50.0, w + 80.0, 5.0]; q["v"][:3] = [10.0, 20.0, -40.0]          # windows partly / fully off the grid
    uright = np.where(rng.uniform(size=nq) > 0.5, kps["x"] - 20 + rng.uniform(-10, 10, nq), -1).astype(np.float32)
    pre = np.full(nq, -1, np.int32); pre[rng.integers(0, nq, 20)] = -2
    return kps, desc, uright, (0.0, 0.0, float(w), float(h)), q, pre


def _np_grid(kps, bounds):
    """AssignFeaturesToGrid (Frame.cc:415-430) in numpy: stable by (cell, insertion index)"""
    iw = np.float32(64) / np.float32(bounds[2] - bounds[0]); ih = np.float32(48) / np.float32(bounds[3] - bounds[1])
    fx = (kps["x"] - np.float32(bounds[0])) * iw; fy = (kps["y"] - np.float32(bounds[1])) * ih
    rnd = lambda v: np.where(v >= 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5))).astype(np.int64)     # C round()
    px = rnd(fx); py = rnd(fy)
    ok = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
    cell = np.where(ok, px * 48 + py, -1)
    order = [i for i in np.argsort(cell, kind="stable") if cell[i] >= 0]
    start = np.zeros(64 * 48 + 1, np.int64)
    for c in cell[ok]:
        start[c + 1] += 1
    return np.cumsum(start), np.asarray(order, np.int64)


def test_grid_built_on_device(iv):
    for seed, n in ((5, 500), (6, 3000)):
        kps, desc, uright, bounds, q, pre = _case(iv, seed, n, 1242 if n > 1000 else 640, 375 if n > 1000 else 240)
        f = iv.DeviceFrame(kps, desc, uright, bounds)
        st, ix = f.grid()
        es, ei = _np_grid(kps, bounds)
        assert np.array_equal(st, es) and np.array_equal(ix, ei)
    # empty frame
    f = iv.DeviceFrame(kps[:0], desc[:0], uright[:0], bounds)
    st, ix = f.grid()
    assert not st.any() and len(ix) == 0


def test_search_by_projection_on_resident_frame(iv):
    m = iv.ORBmatcher(0.9, True)
    for seed, n, radius, big in ((11, 500, 7.0, False), (12, 1500, 15.0, False), (13, 800, 40.0, True)):
        kps, desc, uright, bounds, q, pre = _case(iv, seed, n, radius=radius, big=big)
        f = iv.DeviceFrame(kps, desc, uright, bounds)
        for chk in (True, False):
            ga, gn = f.SearchByProjection(q, chk, pre)
            oa, on = O.search_by_projection(kps, desc, uright, bounds, q, chk, pre)
            assert gn == on and np.array_equal(ga, oa), (seed, chk)
        ha, hn = m.SearchByProjection(kps, desc, uright, bounds, q, pre)           # the host-grid entry point agrees too
        ga, gn = f.SearchByProjection(q, True, pre)
        assert hn == gn and np.array_equal(ha, ga)


def test_search_map_points_on_resident_frame(iv):
    for seed, n, radius in ((21, 600, 4.0), (22, 1200, 10.0)):
        kps, desc, uright, bounds, q, pre = _case(iv, seed, n, radius=radius)
        rng = np.random.default_rng(seed)
        qm = dict(u=q["u"], v=q["v"], ur=q["ur"], radius=q["radius"], level=np.clip(kps["octave"] + rng.integers(-1, 2, len(kps)), 0, 7).astype(np.int32),
                  desc=q["desc"], valid=q["valid"], blocks=q["blocks"])
        f = iv.DeviceFrame(kps, desc, uright, bounds)
        for ratio in (0.6, 0.8, 1.0):
            ga, gn = f.SearchByProjectionMapPoints(qm, ratio, pre)
            oa, on = O.search_map_points(kps, desc, uright, bounds, qm, ratio, pre)
            assert gn == on and np.array_equal(ga, oa), (seed, ratio)


_OVERFLOW = r"""
import sys, os
sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, %r)
import numpy as np
import iv_slam_amd as iv
import oracle_lib as O
import test_gpu_frame as T
kps, desc, uright, bounds, q, pre = T._case(iv, 13, 800, radius=40.0, big=True)
f = iv.DeviceFrame(kps, desc, uright, bounds)
ga, gn = f.SearchByProjection(q, True, pre)
oa, on = O.search_by_projection(kps, desc, uright, bounds, q, True, pre)
assert gn == on and np.array_equal(ga, oa)
print("OK")
"""


def test_window_list_overflow_falls_back_per_query():
    """A window with more candidates than the device-side list holds is redone through the host grid for that query only:
    force it with a list capacity of 4."""
    e = dict(os.environ, IVF_FRAME_WINDOW_CAP="4")
    r = subprocess.run([sys.executable, "-c", _OVERFLOW % (ROOT, ROOT)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@pytest.fixture(scope="module")
def iv():
    import iv_slam_amd
    lib = iv_slam_amd.load()
    assert lib.ivf_device_count() >= 1, "no HIP device: libivfront has no CPU fallback"
    return iv_slam_amd


def test_frames_straight_from_a_frontend_batch_and_all_window_searches(iv):
    """Batch -> resident frames with no host copy (ivf_frame_create_from_frontend) -> every window search of ORBmatcher on
    the device grid: SearchByProjection(cur,last) across two consecutive frames of the batch, and the keyframe / fuse /
    Sim3 / relocalisation searches, each against the oracle on the fetched copies of the same frames."""
    import torch
    w, h, n, pairs = 640, 240, 500, 3
    base_l, base_r = synth.make_pair(w, h, seed=83, idx=0)
    lefts = np.stack([np.roll(base_l, 3 * k, axis=1) for k in range(pairs)])
    rights = np.stack([np.roll(base_r, 3 * k, axis=1) for k in range(pairs)])
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, pairs, nfeatures=n, bf=386.1448, b=386.1448 / 718.856)
    fe.run(torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev))
    bounds = (0.0, 0.0, float(w), float(h))
    frames = [iv.DeviceFrame.from_frontend(fe, p, 0, bounds) for p in range(pairs)]
    host = [fe.fetch(p, 0) for p in range(pairs)]
    sc = iv.ORBextractor(n, 1.2, 8, 20, 7).GetScaleFactors()
    inv2 = (1.0 / (sc * sc)).astype(np.float32)
    rng = np.random.default_rng(7)
    for k in range(1, pairs):
        last, cur = host[k - 1], host[k]
        assert frames[k].n == len(cur["kps"])
        sel = last["uright"] >= 0
        lk = last["kps"][sel]
        disp = lk["x"] - last["uright"][sel]
        q = dict(u=(lk["x"] + 3).astype(np.float32), v=lk["y"].astype(np.float32), ur=(lk["x"] + 3 - disp).astype(np.float32),
                 radius=(7 * sc[lk["octave"]]).astype(np.float32), min_level=(lk["octave"] - 1).astype(np.int32),
                 max_level=(lk["octave"] + 1).astype(np.int32), angle=lk["angle"].copy(), desc=last["desc"][sel].copy(),
                 valid=np.ones(len(lk), np.uint8), blocks=np.ones(len(lk), np.uint8), level=lk["octave"].astype(np.int32))
        ga, gn = frames[k].SearchByProjection(q)
        oa, on = O.search_by_projection(cur["kps"], cur["desc"], cur["uright"], bounds, q, True)
        assert gn == on and np.array_equal(ga, oa) and gn > 0.5 * len(lk)
        # keyframe points / fuse / reloc / sim3 on the same projected queries
        m, nm = frames[k].SearchKeyFramePoints(q)
        om, onm = O.search_keyframe_points(cur["kps"], cur["desc"], bounds, q)
        assert nm == onm and np.array_equal(m, om) and nm > 20
        bi, bd = frames[k].FuseCandidates(inv2, q)
        obi, obd = O.fuse_candidates(cur["kps"], cur["desc"], cur["uright"], bounds, inv2, q)
        assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and (bi >= 0).sum() > 20
        bi2, _ = frames[k].FuseCandidates(None, q)
        obi2, _ = O.fuse_candidates(cur["kps"], cur["desc"], None, bounds, None, q)
        assert np.array_equal(bi2, obi2)
        pre = np.full(len(cur["kps"]), -1, np.int32); pre[rng.integers(0, len(pre), 15)] = -2
        ra, rn = frames[k].SearchByProjectionReloc(q, 100, True, pre)
        ora, orn = O.search_by_projection_reloc(cur["kps"], cur["desc"], bounds, q, 100, True, pre)
        assert rn == orn and np.array_equal(ra, ora)
    # Sim3: queries of frame 0 into frame 1 and back (one query slot per keypoint of the source keyframe)
    def slots(src, shift):
        kp = src["kps"]
        return dict(u=(kp["x"] + shift).astype(np.float32), v=kp["y"].astype(np.float32), radius=(8 * sc[kp["octave"]]).astype(np.float32),
                    level=kp["octave"].astype(np.int32), desc=src["desc"].copy(), valid=(rng.uniform(size=len(kp)) > 0.2).astype(np.uint8))
    q12, q21 = slots(host[0], 3), slots(host[1], -3)
    gm, gf = frames[0].SearchBySim3(frames[1], q12, q21)
    om, of = O.search_by_sim3(host[0]["kps"], host[0]["desc"], bounds, host[1]["kps"], host[1]["desc"], bounds, q12, q21)
    assert gf == of and np.array_equal(gm, om) and gf > 50


def test_config4_cross_frame_matching_4000_features_on_resident_frames(iv):
    """BASELINE configs[4] shape end to end after the front end: two consecutive 1920x1200 stereo pairs at 4000 features with
    the cost map on, batched extraction + stereo, resident frames straight from the batch, the tracker's
    SearchByProjection(cur, last) with th = 15 on the device grid -- against the oracle on the fetched frames."""
    import torch
    w, h, n = 1920, 1200, 4000
    bf, fx = 69.690815 * 2, 528.955512 * 2
    base_l, base_r = synth.make_pair(w, h, seed=151, idx=0)
    cost0 = synth.make_cost_map(w, h, seed=151, idx=0)
    lefts = np.stack([base_l, np.roll(base_l, 4, axis=1)]); rights = np.stack([base_r, np.roll(base_r, 4, axis=1)])
    costs = np.stack([cost0, np.roll(cost0, 4, axis=1)])
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, 2, nfeatures=n, iniThFAST=12, minThFAST=7, enableIntrospection=True, bf=bf, b=bf / fx, fx=fx)
    fe.run(torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev), torch.from_numpy(costs).to(dev))
    bounds = (0.0, 0.0, float(w), float(h))
    cur_frame = iv.DeviceFrame.from_frontend(fe, 1, 0, bounds)
    last, cur = fe.fetch(0, 0), fe.fetch(1, 0)
    sc = iv.ORBextractor(n, 1.2, 8, 12, 7).GetScaleFactors()
    sel = last["uright"] >= 0
    lk = last["kps"][sel]
    q = dict(u=(lk["x"] + 4).astype(np.float32), v=lk["y"].astype(np.float32), ur=(last["uright"][sel] + 4).astype(np.float32),
             radius=(15 * sc[lk["octave"]]).astype(np.float32), min_level=(lk["octave"] - 1).astype(np.int32),
             max_level=(lk["octave"] + 1).astype(np.int32), angle=lk["angle"].copy(), desc=last["desc"][sel].copy(),
             valid=np.ones(len(lk), np.uint8), blocks=np.ones(len(lk), np.uint8))
    ga, gn = cur_frame.SearchByProjection(q)
    oa, on = O.search_by_projection(cur["kps"], cur["desc"], cur["uright"], bounds, q, True)
    assert gn == on and np.array_equal(ga, oa)
    assert len(cur["kps"]) > 3000 and gn > 0.7 * len(lk) > 500
