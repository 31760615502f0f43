"""GPU parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs.  Bar: BIT-EXACT keypoints (all 6 fields), descriptors, pyramids, mvuRight/mvDepth."""
import numpy as np
import pytest

import oracle_lib as O
from iv_slam_amd import synth

pytestmark = pytest.mark.gpu

BF = 386.1448
B = BF / 718.856


@pytest.fixture(scope="module")
def iv():
    import iv_slam_amd
    iv_slam_amd.load()
    lib = iv_slam_amd.load()
    assert lib.ivf_device_count() >= 1, "no HIP device: libivfront has no CPU fallback"
    return iv_slam_amd


def assert_kps_equal(a, b, what=""):
    assert len(a) == len(b), "%s: count %d vs %d" % (what, len(a), len(b))
    for f in a.dtype.names:
        if not np.array_equal(a[f], b[f]):
            bad = np.nonzero(a[f] != b[f])[0]
            raise AssertionError("%s: field %s differs at %d positions, first %d: %r vs %r" %
                                 (what, f, len(bad), bad[0], a[f][bad[0]], b[f][bad[0]]))
    assert a.tobytes() == b.tobytes()


def extract_both(iv, img, cost=None, n=1000, nlevels=8, ini=20, mn=7, sf=1.2, introspection=False):
    g = iv.ORBextractor(n, sf, nlevels, ini, mn, introspection)
    o = O.Extractor(n, sf, nlevels, ini, mn, introspection)
    gk, gd = g(img, cost)
    ok, od = o(img, cost)
    return g, o, gk, gd, ok, od


def test_tables_match_oracle(iv):
    for n in (500, 1000, 2000, 4000):
        g = iv.ORBextractor(n, 1.2, 8, 20, 7)
        t = O.Extractor(n, 1.2, 8, 20, 7).tables()
        assert np.array_equal(g.GetScaleFactors(), t["scale"]) and np.array_equal(g.GetInverseScaleFactors(), t["inv_scale"])
        assert np.array_equal(g.GetScaleSigmaSquares(), t["sigma2"]) and np.array_equal(g.GetInverseScaleSigmaSquares(), t["inv_sigma2"])
        nf, um = g.feature_tables()
        assert np.array_equal(nf, t["features_per_level"]) and np.array_equal(um, t["umax"])
        assert g.GetLevels() == 8 and abs(g.GetScaleFactor() - 1.2) < 1e-6


@pytest.mark.parametrize("size,n", [((320, 200), 300), ((640, 240), 500), ((1242, 375), 1000), ((1242, 375), 2000)])
def test_extract_bit_exact(iv, size, n):
    w, h = size
    for idx in range(2):
        img = synth.make_left(w, h, seed=11, idx=idx)
        g, o, gk, gd, ok, od = extract_both(iv, img, n=n)
        for l in range(8):
            assert np.array_equal(g.mvImagePyramid[l], o.pyramid(l)), "pyramid level %d" % l
        assert g.level_counts() == o.level_counts()
        assert_kps_equal(gk, ok, "keypoints %s n=%d idx=%d" % (size, n, idx))
        assert np.array_equal(gd, od), "descriptors"
        assert len(gk) > n // 2


def test_extract_with_cost_map_bit_exact(iv):
    w, h = 1242, 375
    img = synth.make_left(w, h, seed=5, idx=0)
    cost = synth.make_cost_map(w, h, seed=5, idx=0)
    g, o, gk, gd, ok, od = extract_both(iv, img, cost, n=1000, introspection=True)
    for l in range(8):
        assert np.array_equal(g.mvQualityImagePyramid[l], o.quality_pyramid(l)), "quality level %d" % l
    assert g.level_counts() == o.level_counts()
    assert_kps_equal(gk, ok, "introspection keypoints")
    assert np.array_equal(gd, od)
    # the cost map must actually change the outcome (stale-hY + response weighting)
    _, _, pk, _, _, _ = extract_both(iv, img, None, n=1000, introspection=True)
    assert pk.tobytes() != gk.tobytes()
    # a non-introspective extractor ignores the mask (Tracking.cc:182-183 right extractor)
    g2 = iv.ORBextractor(1000, 1.2, 8, 20, 7, False)
    k2, _ = g2(img, cost)
    assert_kps_equal(k2, pk, "mask ignored")


def test_extract_edge_cases(iv):
    g = iv.ORBextractor(1000, 1.2, 8, 20, 7)
    k, d = g(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)                       # empty image: silent return
    flat = np.full((200, 320), 77, np.uint8)
    gk, gd = g(flat)
    assert len(gk) == 0                                              # no corners anywhere
    # tiny image: upper levels smaller than the 19-px border produce nothing, like the oracle
    img = synth.make_left(120, 90, seed=2, idx=0)
    g, o, gk, gd, ok, od = extract_both(iv, img, n=200)
    assert_kps_equal(gk, ok, "tiny")
    assert np.array_equal(gd, od)
    # reuse of one handle with a different size rebuilds geometry
    img2 = synth.make_left(400, 300, seed=2, idx=1)
    gk2, gd2 = g(img2)
    ok2, od2 = o(img2)
    assert_kps_equal(gk2, ok2, "resize handle")
    # jackal-style thresholds / other pyramid params
    g, o, gk, gd, ok, od = extract_both(iv, img2, n=700, nlevels=6, ini=12, mn=7, sf=1.3)
    assert_kps_equal(gk, ok, "params")
    assert np.array_equal(gd, od)


def test_random_noise_image_many_ties(iv):
    """(also the one image here whose corner-dense tiles overflow k_fast_nms's list of scored pixels, so that the kernel scans the score
    plane instead: the forced variant of that path went away with the run-time ablation mask in r05)"""
    rng = np.random.default_rng(9)
    img = (rng.integers(0, 4, size=(240, 400)) * 60 + 20).astype(np.uint8)     # 4 grey levels: massive response ties
    g, o, gk, gd, ok, od = extract_both(iv, img, n=800)
    assert_kps_equal(gk, ok, "ties")
    assert np.array_equal(gd, od)


@pytest.mark.parametrize("size,n", [((640, 240), 500), ((1242, 375), 1000)])
def test_stereo_matches_bit_exact(iv, size, n):
    w, h = size
    L, R = synth.make_pair(w, h, seed=21, idx=0)
    gL = iv.ORBextractor(n, 1.2, 8, 20, 7); gR = iv.ORBextractor(n, 1.2, 8, 20, 7)
    kL, dL = gL(L); kR, dR = gR(R)
    oL = O.Extractor(n, 1.2, 8, 20, 7); oR = O.Extractor(n, 1.2, 8, 20, 7)
    okL, odL = oL(L); okR, odR = oR(R)
    assert_kps_equal(kL, okL, "L"); assert_kps_equal(kR, okR, "R")
    ur, dp = iv.ComputeStereoMatches(gL, gR, kL, dL, kR, dR, BF, B)
    our, odp = O.stereo_match(oL, oR, okL, odL, okR, odR, BF, B)
    assert ur.tobytes() == our.tobytes(), "mvuRight differs at %r" % (np.nonzero(ur != our)[0][:5],)
    assert dp.tobytes() == odp.tobytes()
    assert (ur >= 0).sum() > n // 10                                # matching really happens
    # no right keypoints / no candidates -> all -1, no gate (Appendix D-8)
    ur0, dp0 = iv.ComputeStereoMatches(gL, gR, kL, dL, kR[:0], dR[:0], BF, B)
    assert (ur0 == -1).all() and (dp0 == -1).all()


def test_frontend_batch_matches_oracle(iv):
    import torch
    w, h, n, pairs = 640, 240, 500, 3
    stream = synth.make_stream(pairs, w, h, seed=31)
    dev = torch.device("cuda:0")
    left = torch.from_numpy(stream[:, 0].copy()).to(dev); right = torch.from_numpy(stream[:, 1].copy()).to(dev)
    fe = iv.StereoFrontend(w, h, 4, nfeatures=n, bf=BF, b=B)
    fe.run(left, right)
    fe.sync()
    assert fe.last_fast_ms() > 0
    for p in range(pairs):
        oL = O.Extractor(n, 1.2, 8, 20, 7); oR = O.Extractor(n, 1.2, 8, 20, 7)
        okL, odL = oL(stream[p, 0]); okR, odR = oR(stream[p, 1])
        our, odp = O.stereo_match(oL, oR, okL, odL, okR, odR, BF, B)
        rl = fe.fetch(p, 0); rr = fe.fetch(p, 1)
        assert_kps_equal(rl["kps"], okL, "pair %d L" % p); assert_kps_equal(rr["kps"], okR, "pair %d R" % p)
        assert np.array_equal(rl["desc"], odL) and np.array_equal(rr["desc"], odR)
        assert rl["uright"].tobytes() == our.tobytes() and rl["depth"].tobytes() == odp.tobytes()
        assert (rl["quality"] == 1.0).all()
    # determinism: a second run of the same batch is identical
    a = fe.fetch(1, 0)
    fe.run(left, right); fe.sync()
    b = fe.fetch(1, 0)
    assert a["kps"].tobytes() == b["kps"].tobytes() and np.array_equal(a["desc"], b["desc"])


def test_frontend_with_cost_maps(iv):
    import torch
    w, h, n, pairs = 640, 240, 500, 2
    stream = synth.make_stream(pairs, w, h, seed=41)
    cost = np.stack([synth.make_cost_map(w, h, seed=41, idx=i) for i in range(pairs)])
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, 2, nfeatures=n, enableIntrospection=True, bf=BF, b=B)
    fe.run(torch.from_numpy(stream[:, 0].copy()).to(dev), torch.from_numpy(stream[:, 1].copy()).to(dev),
           torch.from_numpy(cost).to(dev))
    fe.sync()
    for p in range(pairs):
        oL = O.Extractor(n, 1.2, 8, 20, 7, introspection=True); oR = O.Extractor(n, 1.2, 8, 20, 7, introspection=False)
        okL, odL = oL(stream[p, 0], cost[p]); okR, odR = oR(stream[p, 1], cost[p])       # right ignores the map (D-7)
        rl = fe.fetch(p, 0); rr = fe.fetch(p, 1)
        assert_kps_equal(rl["kps"], okL, "L"); assert_kps_equal(rr["kps"], okR, "R")
        assert np.array_equal(rl["desc"], odL) and np.array_equal(rr["desc"], odR)
        # mvKeyQualScore (Frame.cc:130-143): cost/256, double division narrowed to float
        px = O.c_round(okL["x"]); py = O.c_round(okL["y"])
        c = cost[p][py, px].astype(np.float32)
        q = (np.float64(1.0) / (np.float64(1.0) + (c / np.float32(256)).astype(np.float64))).astype(np.float32)
        assert np.array_equal(rl["quality"], (np.float32(2) * q - np.float32(1)).astype(np.float32))


def test_hamming_pairs_and_search_by_projection(iv):
    rng = np.random.default_rng(77)
    w, h, n = 640, 240, 500
    img = synth.make_left(w, h, seed=51, idx=0)
    g = iv.ORBextractor(n, 1.2, 8, 20, 7)
    kps, desc = g(img)
    m = iv.ORBmatcher(0.9, True)
    pairs = rng.integers(0, len(kps), (2000, 2)).astype(np.int32)
    d = m.DescriptorDistances(desc, desc, pairs)
    exp = np.array([O.hamming(desc[a], desc[b]) for a, b in pairs])
    assert np.array_equal(d, exp)
    assert iv.ORBmatcher.DescriptorDistance(desc[0], desc[1]) == O.hamming(desc[0], desc[1])
    # queries = the frame's own keypoints displaced by a few pixels, descriptors with flipped bits
    nq = len(kps)
    qd = desc.copy()
    flip = rng.integers(0, 256, (nq, 6))
    for i in range(nq):
        for bpos in flip[i]:
            qd[i, bpos // 8] ^= np.uint8(1 << (bpos % 8))
    oct_ = kps["octave"]
    sc = g.GetScaleFactors()
    q = dict(u=kps["x"] + rng.uniform(-3, 3, nq).astype(np.float32), v=kps["y"] + rng.uniform(-3, 3, nq).astype(np.float32),
             ur=(kps["x"] - 20).astype(np.float32), radius=(7 * sc[oct_]).astype(np.float32),
             min_level=(oct_ - 1).astype(np.int32), max_level=(oct_ + 1).astype(np.int32),
             angle=(kps["angle"] + rng.choice([0, 0, 0, 90], nq)).astype(np.float32) % 360, desc=qd,
             valid=(rng.uniform(size=nq) > 0.1).astype(np.uint8), blocks=(rng.uniform(size=nq) > 0.2).astype(np.uint8))
    uright = np.where(rng.uniform(size=nq) > 0.5, kps["x"] - 20 + rng.uniform(-10, 10, nq), -1).astype(np.float32)
    bounds = (0.0, 0.0, float(w), float(h))
    pre = np.full(nq, -1, np.int32); pre[rng.integers(0, nq, 20)] = -2
    ga, gn = m.SearchByProjection(kps, desc, uright, bounds, q, pre)
    oa, on = O.search_by_projection(kps, desc, uright, bounds, q, True, pre)
    assert gn == on and np.array_equal(ga, oa) and gn > nq // 4
    for x, y, r, lo, hi in [(300, 120, 40, -1, -1), (10, 10, 30, 0, 3), (630, 230, 60, 2, 7)]:
        assert np.array_equal(iv.GetFeaturesInArea(kps, bounds, x, y, r, lo, hi), O.features_in_area(kps, bounds, x, y, r, lo, hi))
    # a14: SearchByProjection(F, mapPoints) (Tracking::SearchLocalPoints) with the same perturbed queries
    qm = dict(u=q["u"], v=q["v"], ur=q["ur"], radius=(4.0 * sc[oct_]).astype(np.float32),
              level=np.clip(oct_ + rng.integers(-1, 2, nq), 0, 7).astype(np.int32), desc=qd, valid=q["valid"], blocks=q["blocks"])
    for ratio in (0.6, 0.8, 1.0):
        mm = iv.ORBmatcher(ratio, True)
        ga, gn = mm.SearchByProjectionMapPoints(kps, desc, uright, bounds, qm, pre)
        oa, on = O.search_map_points(kps, desc, uright, bounds, qm, ratio, pre)
        assert gn == on and np.array_equal(ga, oa)
    assert iv.ORBmatcher.RadiusByViewingCos(0.999) == 2.5 and iv.ORBmatcher.RadiusByViewingCos(0.9) == 4.0


def test_search_for_initialization_and_distinctive_descriptor(iv):
    """SearchForInitialization (ORBmatcher.cc:410-519) and the core of
    MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:247-312), HIP path vs oracle on the same inputs."""
    all_searches_scenario(iv, 91, 640, 240, 800, 52, True)


def all_searches_scenario(iv, seed, w, h, nfeat, img_seed, strict):
    """every remaining ORBmatcher search on one two-frame scenario; `strict` adds the yield checks that hold for the 640 x 240
    default scenario (tests/test_gpu_fuzz.py calls this with random sizes and seeds, equality checks only)"""
    rng = np.random.default_rng(seed)
    g = iv.ORBextractor(nfeat, 1.2, 8, 20, 7)
    k1, d1 = g(synth.make_left(w, h, seed=img_seed, idx=0))
    # second frame = the first displaced by a few pixels, descriptors with a few flipped bits, order shuffled
    perm = rng.permutation(len(k1))
    k2 = k1[perm].copy(); d2 = d1[perm].copy()
    k2["x"] += rng.uniform(-4, 4, len(k2)).astype(np.float32); k2["y"] += rng.uniform(-4, 4, len(k2)).astype(np.float32)
    k2["angle"] = (k2["angle"] + rng.choice([0, 0, 0, 0, 120], len(k2))).astype(np.float32) % 360
    for i in range(len(d2)):
        for bpos in rng.integers(0, 256, rng.integers(0, 30)):
            d2[i, bpos // 8] ^= np.uint8(1 << (bpos % 8))
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
    bounds = (0.0, 0.0, float(w), float(h))
    if len(k1) < 8:
        return False                                      # too few keypoints for a scenario (random sizes only)
    total = 0
    for ratio, ori, win in [(0.9, True, 10), (0.9, False, 10), (0.6, True, 25), (1.0, True, 100), (0.9, True, 0)]:
        m = iv.ORBmatcher(ratio, ori)
        gm, gp, gn = m.SearchForInitialization(k1, d1, k2, d2, bounds, prev, win)
        om, op, on = O.search_for_initialization(k1, d1, k2, d2, bounds, prev, win, ratio, ori)
        assert gn == on and np.array_equal(gm, om) and gp.tobytes() == op.tobytes()
        total += gn
    assert not strict or total > 300
    # empty frames
    gm, gp, gn = iv.ORBmatcher(0.9, True).SearchForInitialization(k1[:0], d1[:0], k2, d2, bounds, prev[:0], 10)
    assert gn == 0 and len(gm) == 0
    gm, gp, gn = iv.ORBmatcher(0.9, True).SearchForInitialization(k1, d1, k2[:0], d2[:0], bounds, prev, 10)
    assert gn == 0 and (gm == -1).all()
    # distinctive descriptor: observation sets of several sizes, with exact duplicates (median ties -> first minimum)
    for n in (1, 2, 5, 64, 257, 1000):
        obs = d1[rng.integers(0, min(len(d1), 40), n)].copy()
        for i in range(n):
            for bpos in rng.integers(0, 256, rng.integers(0, 24)):
                obs[i, bpos // 8] ^= np.uint8(1 << (bpos % 8))
        assert iv.ComputeDistinctiveDescriptors(obs) == O.distinctive_descriptor(obs)
    with pytest.raises(AssertionError):
        iv.ComputeDistinctiveDescriptors(np.zeros((0, 32), np.uint8))
    # SearchByProjection(KF, Scw) and the Fuse core on perturbed projections of frame 2's own keypoints
    nq = len(k2); oct2 = k2["octave"]; sc = g.GetScaleFactors(); inv_s2 = g.GetInverseScaleSigmaSquares()
    qd = d2.copy()
    for i in range(nq):
        for bpos in rng.integers(0, 256, rng.integers(0, 40)):
            qd[i, bpos // 8] ^= np.uint8(1 << (bpos % 8))
    q = dict(u=(k2["x"] + rng.uniform(-2, 2, nq)).astype(np.float32), v=(k2["y"] + rng.uniform(-2, 2, nq)).astype(np.float32),
             ur=(k2["x"] - 15 + rng.uniform(-2, 2, nq)).astype(np.float32), radius=(4 * sc[oct2]).astype(np.float32),
             level=np.clip(oct2 + rng.integers(-1, 2, nq), 0, 7).astype(np.int32), desc=qd,
             valid=(rng.uniform(size=nq) > 0.1).astype(np.uint8))
    pre = np.full(nq, -1, np.int32); pre[rng.integers(0, nq, 40)] = -2
    m = iv.ORBmatcher(0.75, True)
    gm, gn = m.SearchByProjectionKeyFrame(k2, d2, bounds, q, pre)
    om, on = O.search_keyframe_points(k2, d2, bounds, q, pre)
    assert gn == on and np.array_equal(gm, om) and (not strict or gn > nq // 5)
    ur2 = np.where(rng.uniform(size=nq) > 0.4, k2["x"] - 15, -1).astype(np.float32)
    gb, gd = m.FuseCandidates(k2, d2, ur2, bounds, inv_s2, q)
    ob, od = O.fuse_candidates(k2, d2, ur2, bounds, inv_s2, q)
    assert np.array_equal(gb, ob) and np.array_equal(gd, od) and (not strict or (gb >= 0).sum() > nq // 5)
    # SearchBySim3: KF1 = frame 1, KF2 = frame 2 (a displaced, shuffled copy): project each keypoint's map point to where its
    # partner sits in the other frame (+ noise), partner known from the permutation
    inv = np.argsort(perm)
    q12 = dict(u=(k2["x"][inv] + rng.uniform(-2, 2, nq)).astype(np.float32), v=(k2["y"][inv] + rng.uniform(-2, 2, nq)).astype(np.float32),
               radius=(7.5 * sc[k1["octave"]]).astype(np.float32), level=np.clip(k1["octave"] + rng.integers(0, 2, nq), 0, 7).astype(np.int32),
               desc=d1, valid=(rng.uniform(size=nq) > 0.15).astype(np.uint8))
    q21 = dict(u=(k1["x"][perm] + rng.uniform(-2, 2, nq)).astype(np.float32), v=(k1["y"][perm] + rng.uniform(-2, 2, nq)).astype(np.float32),
               radius=(7.5 * sc[oct2]).astype(np.float32), level=np.clip(oct2 + rng.integers(0, 2, nq), 0, 7).astype(np.int32),
               desc=qd, valid=(rng.uniform(size=nq) > 0.15).astype(np.uint8))
    gs, gf = m.SearchBySim3(k1, d1, bounds, k2, d2, bounds, q12, q21)
    os_, of = O.search_by_sim3(k1, d1, bounds, k2, d2, bounds, q12, q21)
    assert gf == of and np.array_equal(gs, os_) and (not strict or gf > nq // 4)
    assert not strict or (gs[gs >= 0] == inv[gs >= 0]).mean() > 0.9         # mostly the true partners
    # SearchByBoW(KF, F): a synthetic vocabulary level = 64 nodes keyed by descriptor bits, so that most true partners share
    # a node; frame 2's keypoints carry a few flipped bits and some land in other nodes
    def feat_vec(desc, drop):
        fv = {}
        for i in range(len(desc)):
            if drop[i]: continue
            fv.setdefault(int(desc[i, 0] & 0x3F) * 7 + 3, []).append(i)      # sparse, non-contiguous node ids
        return fv
    fv1 = feat_vec(d1, rng.uniform(size=nq) < 0.05); fv2 = feat_vec(d2, rng.uniform(size=nq) < 0.05)
    has_mp = (rng.uniform(size=nq) > 0.2).astype(np.uint8)
    for ratio, ori in [(0.7, True), (0.9, False), (0.6, True)]:
        mm = iv.ORBmatcher(ratio, ori)
        gm_, gn_ = mm.SearchByBoW(k1, d1, has_mp, fv1, k2, d2, fv2)
        om_, on_ = O.search_by_bow(k1, d1, has_mp, fv1, k2, d2, fv2, ratio, ori)
        assert gn_ == on_ and np.array_equal(gm_, om_) and (not strict or gn_ > 50)
        assert has_mp[gm_[gm_ >= 0]].all()                                  # only keyframe features that own a map point
    gm_, gn_ = iv.ORBmatcher(0.7, True).SearchByBoW(k1, d1, has_mp, {}, k2, d2, fv2)
    assert gn_ == 0 and (gm_ == -1).all()
    has2 = (rng.uniform(size=nq) > 0.25).astype(np.uint8)
    for ratio, ori in [(0.75, True), (0.9, False)]:
        mm = iv.ORBmatcher(ratio, ori)
        gk, gkn = mm.SearchByBoWKeyFrames(k1, d1, has_mp, fv1, k2, d2, has2, fv2)
        ok_, okn = O.search_by_bow_keyframes(k1, d1, has_mp, fv1, k2, d2, has2, fv2, ratio, ori)
        assert gkn == okn and np.array_equal(gk, ok_) and (not strict or gkn > 30)
        assert has_mp[np.nonzero(gk >= 0)[0]].all() and has2[gk[gk >= 0]].all()
        assert len(np.unique(gk[gk >= 0])) == (gk >= 0).sum()                # vbMatched2: a KF2 feature is claimed once
    # SearchForTriangulation: pure horizontal translation between the cameras => F12 = [t]x with t = (1,0,0): epipolar lines are
    # the rows y2 = y1; the epipole is at infinity (put far outside the image) -- plus a tilted F to exercise the general case
    sig2 = g.GetScaleSigmaSquares()
    st1 = (rng.uniform(size=nq) > 0.5).astype(np.uint8); st2 = (rng.uniform(size=nq) > 0.5).astype(np.uint8)
    nomp1 = (rng.uniform(size=nq) > 0.4).astype(np.uint8); nomp2 = (rng.uniform(size=nq) > 0.4).astype(np.uint8)
    for F12, ex, ey in [(np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32), 1e6, 120.0),
                        (np.array([[0, 0, 0.002], [0, 0, -1], [-0.002, 1, 0.3]], np.float32), 320.0, 118.0)]:
        for only_st, ori in [(False, True), (True, True), (False, False)]:
            mm = iv.ORBmatcher(0.6, ori)
            gt, gtn = mm.SearchForTriangulation(k1, d1, 1 - nomp1, st1, fv1, k2, d2, 1 - nomp2, st2, fv2, F12, ex, ey, sc, sig2, only_st)
            ot, otn = O.search_for_triangulation(k1, d1, 1 - nomp1, st1, fv1, k2, d2, 1 - nomp2, st2, fv2, F12, ex, ey, sc, sig2, only_st, ori)
            assert gtn == otn and np.array_equal(gt, ot)
            sel = np.nonzero(gt >= 0)[0]
            assert nomp1[sel].all() and nomp2[gt[sel]].all()                 # only features without a map point
            if only_st: assert st1[sel].all() and st2[gt[sel]].all()
        assert not strict or gtn > 10
    # relocalisation SearchByProjection(CurrentFrame, KF, ...): same perturbed projections, angles from "the keyframe"
    qr = dict(q); qr["angle"] = ((k2["angle"] + rng.choice([0, 0, 0, 0, 75], nq)) % 360).astype(np.float32)
    for orbd, ori in [(100, True), (64, True), (64, False), (0, True)]:
        ga_, gn2 = iv.ORBmatcher(0.9, ori).SearchByProjectionReloc(k2, d2, bounds, qr, orbd, pre)
        oa_, on2 = O.search_by_projection_reloc(k2, d2, bounds, qr, orbd, ori, pre)
        assert gn2 == on2 and np.array_equal(ga_, oa_)
    assert (not strict or gn2 < nq // 4) and (ga_[pre == -2] == -2).all()
    gb2, gd2 = m.FuseCandidates(k2, d2, None, bounds, None, q)              # Fuse(KF, Scw, ...): no chi-square gate
    ob2, od2 = O.fuse_candidates(k2, d2, None, bounds, None, q)
    assert np.array_equal(gb2, ob2) and np.array_equal(gd2, od2) and (gb2 >= 0).sum() >= (gb >= 0).sum()
    return True


def test_bow_transform_and_vectors(iv):
    """DBoW2 transform (descriptor -> word / node / weight), BowVector, FeatureVector, and the
    feature vectors feeding SearchByBoW end to end."""
    rng = np.random.default_rng(17)
    g = iv.ORBextractor(800, 1.2, 8, 20, 7)
    k1, d1 = g(synth.make_left(640, 240, seed=53, idx=0))
    for k, depth, early, levelsup in [(10, 3, 0.0, 1), (10, 4, 0.1, 4), (5, 6, 0.15, 4), (2, 1, 0.0, 4), (17, 2, 0.0, 1)]:
        voc = O.make_vocabulary(k, depth, seed=100 + k + depth, early_leaf_frac=early, stop_frac=0.1)
        V = iv.ORBVocabulary(voc["child_start"], voc["child"], voc["desc"], voc["word"], voc["weight"], voc["depth"])
        desc = np.concatenate([d1, voc["desc"][rng.integers(1, len(voc["desc"]), 50)]])      # incl. exact node descriptors (ties)
        gw, gn, gt = V.transform_features(desc, levelsup)
        ow, on, ot = O.bow_transform(voc, desc, levelsup)
        assert np.array_equal(gw, ow) and np.array_equal(gn, on) and np.array_equal(gt, ot)
        bow, fv = V.transform(desc, levelsup)
        # restatement of BowVector::addWeight + L1 normalize and FeatureVector::addFeature (DBoW2 BowVector.cpp, FeatureVector.cpp)
        eb, ef = {}, {}
        for f in range(len(desc)):
            if ot[f] > 0:
                eb[int(ow[f])] = eb.get(int(ow[f]), 0.0) + float(ot[f]); ef.setdefault(int(on[f]), []).append(f)
        norm = 0.0
        for w_ in sorted(eb): norm += abs(eb[w_])
        eb = {w_: v_ / norm for w_, v_ in eb.items()} if norm > 0 else eb
        assert bow == eb and fv == ef and abs(sum(bow.values()) - 1.0) < 1e-12
    # end to end: two frames through the same vocabulary, then SearchByBoW on their feature vectors
    voc = O.make_vocabulary(10, 3, seed=7)
    V = iv.ORBVocabulary(voc["child_start"], voc["child"], voc["desc"], voc["word"], voc["weight"], voc["depth"])
    d2 = d1.copy()
    for i in range(len(d2)):
        for bpos in rng.integers(0, 256, 4):
            d2[i, bpos // 8] ^= np.uint8(1 << (bpos % 8))
    _, fv1 = V.transform(d1, 1); _, fv2 = V.transform(d2, 1)
    has = np.ones(len(k1), np.uint8)
    m = iv.ORBmatcher(0.7, True)
    gm, gn_ = m.SearchByBoW(k1, d1, has, fv1, k1, d2, fv2)
    om, on_ = O.search_by_bow(k1, d1, has, fv1, k1, d2, fv2, 0.7, True)
    assert gn_ == on_ and np.array_equal(gm, om) and gn_ > len(k1) // 4
    assert (gm[gm >= 0] == np.nonzero(gm >= 0)[0]).mean() > 0.95
    with pytest.raises(iv.IvfError):                                      # a child listed twice is not a tree
        iv.ORBVocabulary(np.array([0, 2, 2, 2], np.int32), np.array([1, 1], np.int32), voc["desc"][:3], np.zeros(3, np.int32), np.ones(3), 1)


def test_full_size_properties(iv):
    """BASELINE full size: properties that need no oracle (idempotence, level order, border, uniqueness)."""
    img = synth.make_left(1242, 375, seed=61, idx=0)
    g = iv.ORBextractor(1000, 1.2, 8, 20, 7)
    k1, d1 = g(img); k2, d2 = g(img)
    assert k1.tobytes() == k2.tobytes() and np.array_equal(d1, d2)           # idempotent
    assert (np.diff(k1["octave"]) >= 0).all()                                # levels concatenated in order
    sc = g.GetScaleFactors()
    lw = [p.shape[1] for p in g.mvImagePyramid]; lh = [p.shape[0] for p in g.mvImagePyramid]
    for l in range(8):
        s = k1[k1["octave"] == l]
        x = np.rint(s["x"] / sc[l]); y = np.rint(s["y"] / sc[l])
        assert (x >= 19).all() and (x < lw[l] - 19).all() and (y >= 19).all() and (y < lh[l] - 19).all()
        assert len(s) <= [217, 181, 151, 126, 105, 87, 73, 60][l]
        assert len(set(zip(x.tolist(), y.tolist()))) == len(s)                # NMS: no duplicate positions
    assert ((k1["angle"] >= 0) & (k1["angle"] < 360.0001)).all() and (k1["response"] >= 7).all()


def test_committed_golden_fixtures(iv):
    """HIP path against the committed fixtures (tests/golden, frozen oracle outputs on seeded inputs)."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "mini_320x200.npz"))
    for prefix, cost, intro in (("plain_", None, False), ("intro_", g["cost"], True)):
        eL = iv.ORBextractor(300, 1.2, 8, 20, 7, intro); eR = iv.ORBextractor(300, 1.2, 8, 20, 7, False)
        kL, dL = eL(g["left"], cost); kR, dR = eR(g["right"], cost)
        ur, dp = iv.ComputeStereoMatches(eL, eR, kL, dL, kR, dR, BF, B)
        assert kL.tobytes() == g[prefix + "kpsL"].tobytes() and kR.tobytes() == g[prefix + "kpsR"].tobytes()
        assert np.array_equal(dL, g[prefix + "descL"]) and np.array_equal(dR, g[prefix + "descR"])
        assert ur.tobytes() == g[prefix + "uright"].tobytes() and dp.tobytes() == g[prefix + "depth"].tobytes()
        assert eL.level_counts() == g[prefix + "level_counts"].tolist()
    k = np.load(os.path.join(gold, "kitti_1242x375.npz"))
    L, R = synth.make_pair(1242, 375, seed=int(k["seed"][0]), idx=int(k["seed"][1]))
    cost = synth.make_cost_map(1242, 375, seed=int(k["seed"][0]), idx=int(k["seed"][1]))
    for prefix, c, intro, n in (("plain_", None, False, 1000), ("intro_", cost, True, 1000), ("n2000_", None, False, 2000)):
        eL = iv.ORBextractor(n, 1.2, 8, 20, 7, intro); eR = iv.ORBextractor(n, 1.2, 8, 20, 7, False)
        kL, dL = eL(L, c); kR, dR = eR(R, c)
        ur, dp = iv.ComputeStereoMatches(eL, eR, kL, dL, kR, dR, BF, B)
        assert kL.tobytes() == k[prefix + "kpsL"].tobytes() and np.array_equal(dL, k[prefix + "descL"])
        assert kR.tobytes() == k[prefix + "kpsR"].tobytes() and np.array_equal(dR, k[prefix + "descR"])
        assert ur.tobytes() == k[prefix + "uright"].tobytes() and dp.tobytes() == k[prefix + "depth"].tobytes()


def test_device_retain_best_matches_libstdcxx(iv):
    """The wave-cooperative introselect on the device against the REAL libstdc++ nth_element (oracle/stl_pin.cpp):
    tie-heavy random inputs, sorted/reversed runs, and organ-pipe inputs that hit the depth-limit heap-select."""
    import ctypes as C
    from iv_slam_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(2024)

    def check(resp, k):
        n = len(resp)
        resp = np.ascontiguousarray(resp, np.float32)
        order = np.zeros(n, np.int32)
        _lib.check(lib.ivf_test_retain_best(_lib.ptr(resp), n, k, _lib.ptr(order), 0))
        v = np.zeros(n, O.KP_DTYPE); v["response"] = resp; v["x"] = np.arange(n)
        if 0 < k < n:
            O.pin.stl_nth_element(O.ptr(v), n, k - 1)
        assert np.array_equal(order, v["x"].astype(np.int32)), (n, k)

    for _ in range(150):
        n = int(rng.integers(1, 3000))
        mode = int(rng.integers(0, 4))
        resp = rng.integers(7, 7 + int(rng.integers(1, 80)), n).astype(np.float32)
        if mode == 1:
            resp *= rng.choice(np.array([0.25, 0.5, 1.0], np.float32), n)
        elif mode == 2:
            resp = np.sort(resp)[::-1].copy() if rng.integers(0, 2) else np.sort(resp)
        elif mode == 3:
            resp[:] = 42.0
        check(resp, int(rng.integers(0, n + 2)))
    for n in (64, 257, 1000, 4096):
        base = np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)[::-1]]).astype(np.float32)
        for k in (1, 2, n // 3, n // 2, n - 1):
            check(base, k)


def test_jackal_shape_4000_features(iv):
    """BASELINE configs[4] shape: 1920x1200, 4000 features/frame, FAST 12/7, with a cost map (one image: the oracle
    needs ~1 s for it)."""
    w, h, n = 1920, 1200, 4000
    img = synth.make_left(w, h, seed=71, idx=0)
    cost = synth.make_cost_map(w, h, seed=71, idx=0)
    g = iv.ORBextractor(n, 1.2, 8, 12, 7, True)
    o = O.Extractor(n, 1.2, 8, 12, 7, True)
    gk, gd = g(img, cost)
    ok, od = o(img, cost, cap=2 * n)
    assert g.level_counts() == o.level_counts()
    assert_kps_equal(gk, ok, "1920x1200 N=4000")
    assert np.array_equal(gd, od)
    assert len(gk) > 3000
    # and the 960x600 / 2000-feature Jackal YAML shape without a cost map
    img2 = synth.make_left(960, 600, seed=72, idx=0)
    g2, o2, gk2, gd2, ok2, od2 = extract_both(iv, img2, n=2000, ini=12, mn=7)
    assert_kps_equal(gk2, ok2, "960x600 N=2000")
    assert np.array_equal(gd2, od2)


def test_cross_frame_matching_through_gather_records(iv):
    """SURVEY §8(e) flow on one GPU: batch of consecutive frames -> device-packed gather records (what RCCL all-gathers)
    -> cross-frame SearchByProjection(cur, last) on the unpacked records, against the oracle end to end."""
    import torch
    from iv_slam_amd.frontend import unpack_gather_records
    w, h, n, pairs = 640, 240, 500, 3
    base_l, base_r = synth.make_pair(w, h, seed=81, idx=0)
    # consecutive "frames": the scene shifts 3 px per frame (left AND right), so last-frame points reproject nearby
    lefts = np.stack([np.roll(base_l, 3 * k, axis=1) for k in range(pairs)])
    rights = np.stack([np.roll(base_r, 3 * k, axis=1) for k in range(pairs)])
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, pairs, nfeatures=n, bf=BF, b=B)
    fe.run(torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev))
    rec = fe.gather_record_bytes()
    block = torch.zeros(pairs * rec, dtype=torch.uint8, device=dev)
    assert fe.pack_gather_block(block) == rec
    torch.cuda.synchronize()
    frames = unpack_gather_records(block.cpu().numpy(), n)
    sc = iv.ORBextractor(n, 1.2, 8, 20, 7).GetScaleFactors()
    m = iv.ORBmatcher(0.9, True)
    bounds = (0.0, 0.0, float(w), float(h))
    for k in range(1, pairs):
        last, cur = frames[k - 1], frames[k]
        # oracle extraction of the same frames must equal the gathered records
        oL = O.Extractor(n, 1.2, 8, 20, 7); oR = O.Extractor(n, 1.2, 8, 20, 7)
        okL, odL = oL(lefts[k]); okR, odR = oR(rights[k])
        our, _ = O.stereo_match(oL, oR, okL, odL, okR, odR, BF, B)
        assert cur["kps"].tobytes() == okL.tobytes() and np.array_equal(cur["desc"], odL) and cur["uright"].tobytes() == our.tobytes()
        sel = last["uright"] >= 0                                     # "map points" = last frame's stereo points
        lk = last["kps"][sel]
        disp = lk["x"] - last["uright"][sel]
        q = dict(u=(lk["x"] + 3).astype(np.float32), v=lk["y"].astype(np.float32),
                 ur=(lk["x"] + 3 - disp).astype(np.float32), radius=(7 * sc[lk["octave"]]).astype(np.float32),
                 min_level=(lk["octave"] - 1).astype(np.int32), max_level=(lk["octave"] + 1).astype(np.int32),
                 angle=lk["angle"].copy(), desc=last["desc"][sel].copy(), valid=np.ones(len(lk), np.uint8),
                 blocks=np.ones(len(lk), np.uint8))
        ga, gn = m.SearchByProjection(cur["kps"], cur["desc"], cur["uright"], bounds, q)
        oa, on = O.search_by_projection(cur["kps"], cur["desc"], cur["uright"], bounds, q, True)
        assert gn == on and np.array_equal(ga, oa)
        assert gn > 0.5 * len(lk)                                     # the shifted scene really re-matches


@pytest.mark.parametrize("size,n,ini", [((1242, 375), 1000, 20), ((640, 240), 500, 20), ((960, 600), 2000, 12)])
def test_blur_planes_bit_exact(iv, size, n, ini):
    """Row a7 directly: every blurred level the descriptors were sampled from (GaussianBlur 7x7 sigma 2, REFLECT_101 of the
    un-padded level, ORBextractor.cc:1276-1277) byte for byte against the oracle's blur of the oracle's own pyramid."""
    w, h = size
    img = synth.make_left(w, h, seed=91, idx=1)
    g, o, gk, gd, ok, od = extract_both(iv, img, n=n, ini=ini)
    counts = g.level_counts()
    assert counts == o.level_counts()
    checked = 0
    for l in range(8):
        if counts[l] == 0:
            continue
        want = O.gauss7(np.ascontiguousarray(o.pyramid(l)))
        got = g.blur_level(l)
        assert got.shape == want.shape
        assert np.array_equal(got, want), "blurred level %d differs at %d pixels" % (l, int((got != want).sum()))
        checked += 1
    assert checked >= 6


def test_config3_batched_frontend_2000_features(iv):
    """BASELINE configs[3] on one rank: batched 8-level pyramid at 1242x375 with 2000 features/frame through the batched
    front end (4 pairs in one launch sequence), incl. the gather records the ranks exchange, all vs the oracle."""
    import torch
    from iv_slam_amd.frontend import unpack_gather_records
    w, h, n, pairs = 1242, 375, 2000, 4
    stream = synth.make_stream(pairs, w, h, seed=131)
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, pairs, nfeatures=n, bf=BF, b=B)
    fe.run(torch.from_numpy(stream[:, 0].copy()).to(dev), torch.from_numpy(stream[:, 1].copy()).to(dev))
    rec = fe.gather_record_bytes()
    assert rec == 16 + n * 64
    block = torch.zeros(pairs * rec, dtype=torch.uint8, device=dev)
    fe.pack_gather_block(block)
    fe.sync()
    torch.cuda.synchronize()
    frames = unpack_gather_records(block.cpu().numpy(), n)
    for p in range(pairs):
        oL = O.Extractor(n, 1.2, 8, 20, 7); oR = O.Extractor(n, 1.2, 8, 20, 7)
        okL, odL = oL(stream[p, 0]); okR, odR = oR(stream[p, 1])
        our, odp = O.stereo_match(oL, oR, okL, odL, okR, odR, BF, B)
        rl = fe.fetch(p, 0); rr = fe.fetch(p, 1)
        assert_kps_equal(rl["kps"], okL, "pair %d L" % p); assert_kps_equal(rr["kps"], okR, "pair %d R" % p)
        assert np.array_equal(rl["desc"], odL) and np.array_equal(rr["desc"], odR)
        assert rl["uright"].tobytes() == our.tobytes() and rl["depth"].tobytes() == odp.tobytes()
        assert len(okL) > 1500 and (our >= 0).sum() > 100
        f = frames[p]
        assert f["n"] == len(okL) and f["kps"].tobytes() == okL.tobytes() and np.array_equal(f["desc"], odL)
        assert f["uright"].tobytes() == our.tobytes()


def test_config4_jackal_stereo_frontend_4000_features_introspection(iv):
    """BASELINE configs[4] on one rank: 1920x1200 stereo pairs, 4000 features/frame, FAST 12/7, cost map ON, through the
    BATCHED front end incl. mvuRight / mvDepth / mvKeyQualScore (Frame.cc:89-230), vs the oracle."""
    import torch
    w, h, n, pairs = 1920, 1200, 4000, 2
    bf, fx = 69.690815 * 2, 528.955512 * 2          # jackal_visual_odom_stereo_inference.yaml:8,27 scaled to the un-binned sensor
    b = bf / fx
    stream = synth.make_stream(pairs, w, h, seed=141)
    cost = np.stack([synth.make_cost_map(w, h, seed=141, idx=i) for i in range(pairs)])
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, pairs, nfeatures=n, iniThFAST=12, minThFAST=7, enableIntrospection=True, bf=bf, b=b, fx=fx)
    fe.run(torch.from_numpy(stream[:, 0].copy()).to(dev), torch.from_numpy(stream[:, 1].copy()).to(dev),
           torch.from_numpy(cost).to(dev))
    fe.sync()
    for p in range(pairs):
        oL = O.Extractor(n, 1.2, 8, 12, 7, introspection=True); oR = O.Extractor(n, 1.2, 8, 12, 7, introspection=False)
        okL, odL = oL(stream[p, 0], cost[p], cap=2 * n); okR, odR = oR(stream[p, 1], cost[p], cap=2 * n)
        our, odp = O.stereo_match(oL, oR, okL, odL, okR, odR, bf, b)
        rl = fe.fetch(p, 0); rr = fe.fetch(p, 1)
        assert_kps_equal(rl["kps"], okL, "pair %d L" % p); assert_kps_equal(rr["kps"], okR, "pair %d R" % p)
        assert np.array_equal(rl["desc"], odL) and np.array_equal(rr["desc"], odR)
        assert rl["uright"].tobytes() == our.tobytes() and rl["depth"].tobytes() == odp.tobytes()
        assert len(okL) > 3000 and (our >= 0).sum() > 200
        px = O.c_round(okL["x"]); py = O.c_round(okL["y"])
        c = cost[p][py, px].astype(np.float32)
        q = (np.float64(1.0) / (np.float64(1.0) + (c / np.float32(256)).astype(np.float64))).astype(np.float32)
        assert np.array_equal(rl["quality"], (np.float32(2) * q - np.float32(1)).astype(np.float32))


def test_huge_cells_take_the_global_fallback(iv):
    """Few features on a big corner-dense image: 1920x1200 at N=500 has 627x290-pixel cells holding far more than the 4096
    survivors the LDS selection paths handle; those cells go through k_cell_select_huge (global scratch).  The handle must
    stay usable afterwards, and a later ordinary image must not see stale flags."""
    rng = np.random.default_rng(19)
    img = (rng.integers(0, 4, size=(1200, 1920)) * 60 + 20).astype(np.uint8)
    g = iv.ORBextractor(500, 1.2, 8, 20, 7)
    o = O.Extractor(500, 1.2, 8, 20, 7)
    gk, gd = g(img)
    ok, od = o(img, cap=2000)
    assert g.level_counts() == o.level_counts()
    assert_kps_equal(gk, ok, "huge cells")
    assert np.array_equal(gd, od)
    img2 = synth.make_left(1920, 1200, seed=3, idx=0)
    gk2, gd2 = g(img2)
    ok2, od2 = o(img2, cap=2000)
    assert_kps_equal(gk2, ok2, "after huge")
    assert np.array_equal(gd2, od2)


def test_quality_scores_without_extractor_introspection(iv):
    """Frame.cc:130-143 fills mvKeyQualScore whenever a cost image comes with the frame, also when the extractor was
    built with enableIntrospection = 0 (then the keypoints are the plain ones and only the scores use the map)."""
    import torch
    w, h, n = 640, 240, 500
    stream = synth.make_stream(1, w, h, seed=43)
    cost = np.stack([synth.make_cost_map(w, h, seed=43, idx=0)])
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, 1, nfeatures=n, enableIntrospection=False, bf=BF, b=B)
    fe.run(torch.from_numpy(stream[:, 0].copy()).to(dev), torch.from_numpy(stream[:, 1].copy()).to(dev), torch.from_numpy(cost).to(dev))
    fe.sync()
    okL, odL = O.Extractor(n, 1.2, 8, 20, 7)(stream[0, 0])
    rl = fe.fetch(0, 0)
    assert_kps_equal(rl["kps"], okL, "plain keypoints")
    px = O.c_round(okL["x"]); py = O.c_round(okL["y"])
    c = cost[0][py, px].astype(np.float32)
    q = (np.float64(1.0) / (np.float64(1.0) + (c / np.float32(256)).astype(np.float64))).astype(np.float32)
    assert np.array_equal(rl["quality"], (np.float32(2) * q - np.float32(1)).astype(np.float32))
    assert (fe.fetch(0, 1)["quality"] == 1.0).all()
    # and without a cost image the scores stay 1 (Frame.cc:137-141 else branch)
    fe.run(torch.from_numpy(stream[:, 0].copy()).to(dev), torch.from_numpy(stream[:, 1].copy()).to(dev))
    fe.sync()
    assert (fe.fetch(0, 0)["quality"] == 1.0).all()


@pytest.mark.parametrize("variant", [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)])
def test_opencv_variant_switches(iv, variant):
    """Older-OpenCV forms of the three un-pinned primitives (blur table, retainBest nth position, fastAtan2): device ==
    oracle under each switch, and each switch really changes the result."""
    import torch
    w, h, n = 640, 240, 500
    img = synth.make_left(w, h, seed=61, idx=0)
    base_k, base_d = iv.ORBextractor(n, 1.2, 8, 20, 7)(img)
    try:
        O.set_opencv_variant(*variant)
        g = iv.ORBextractor(n, 1.2, 8, 20, 7)
        g.set_opencv_variant(*variant)
        gk, gd = g(img)
        ok, od = O.Extractor(n, 1.2, 8, 20, 7)(img)
        assert_kps_equal(gk, ok, "variant %r" % (variant,))
        assert np.array_equal(gd, od)
        assert gk.tobytes() != base_k.tobytes() or not np.array_equal(gd, base_d)
        # the batched front end takes the same switches
        stream = synth.make_stream(1, w, h, seed=61)
        fe = iv.StereoFrontend(w, h, 1, nfeatures=n, bf=BF, b=B)
        fe.set_opencv_variant(*variant)
        dev = torch.device("cuda:0")
        fe.run(torch.from_numpy(stream[:, 0].copy()).to(dev), torch.from_numpy(stream[:, 1].copy()).to(dev))
        fe.sync()
        assert_kps_equal(fe.fetch(0, 0)["kps"], ok, "front end variant")
        # switching back restores the default
        g.set_opencv_variant()
        k0, d0 = g(img)
        assert_kps_equal(k0, base_k, "default restored") 
    finally:
        O.set_opencv_variant()


def test_frontend_strided_inputs(iv):
    """the batched front end on padded rows / images (views of larger device tensors, odd byte offsets): k_ingest reads through
    the caller's strides; results equal the contiguous run"""
    import torch
    w, h, n, pairs = 637, 241, 400, 3
    stream = synth.make_stream(pairs, w, h, seed=77)
    cost = np.stack([synth.make_cost_map(w, h, seed=77, idx=i) for i in range(pairs)])
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, pairs, nfeatures=n, enableIntrospection=True, bf=BF, b=B)
    L = torch.from_numpy(stream[:, 0].copy()).to(dev); R = torch.from_numpy(stream[:, 1].copy()).to(dev); Cm = torch.from_numpy(cost).to(dev)
    fe.run(L, R, Cm); fe.sync()
    ref = [(fe.fetch(p, 0), fe.fetch(p, 1)) for p in range(pairs)]
    big = [torch.zeros((pairs, h + 5, w + 43), dtype=torch.uint8, device=dev) for _ in range(3)]
    views = [b[:, 2:2 + h, 7:7 + w] for b in big]
    for v, src in zip(views, (L, R, Cm)):
        v.copy_(src)
    assert not views[0].is_contiguous()
    fe.run(views[0], views[1], views[2]); fe.sync()
    for p in range(pairs):
        for side in (0, 1):
            a = fe.fetch(p, side); b_ = ref[p][side]
            assert_kps_equal(a["kps"], b_["kps"], "strided pair %d side %d" % (p, side))
            assert np.array_equal(a["desc"], b_["desc"]) and np.array_equal(a["quality"], b_["quality"])
        assert fe.fetch(p, 0)["uright"].tobytes() == ref[p][0]["uright"].tobytes()


@pytest.mark.parametrize("rgb,cv3", [(False, False), (True, False), (False, True), (True, True)])
def test_frontend_color_inputs_gray_conversion_fused_into_the_ingest(iv, rgb, cv3):
    """r06: ivf_frontend_run_color = Tracking::GrabImageStereo's cvtColor (Tracking.cc:272-295, by mbRGB) + ivf_frontend_run.  The left side arrives as
    interleaved colour (what the FCN reads too), the right side grey or colour, with different strides per side, odd widths (the row's last 16-byte piece),
    padded views; level 0 must equal the oracle's cvtColor byte for byte, i.e. keypoints / descriptors / stereo matches / quality equal the oracle chain on
    the converted images -- for both byte orders and both OpenCV coefficient generations."""
    import torch
    w, h, n, pairs = 637, 241, 400, 3
    rng = np.random.default_rng(5 + 2 * int(rgb) + int(cv3))
    stream = synth.make_stream(pairs, w, h, seed=78)
    cost = np.stack([synth.make_cost_map(w, h, seed=78, idx=i) for i in range(pairs)])

    def colourise(g):          # a colour image whose grey conversion is textured like g: channels = g +- small, clipped
        c = np.stack([np.clip(g.astype(int) + rng.integers(-40, 41, g.shape), 0, 255) for _ in range(3)], axis=-1)
        return c.astype(np.uint8)
    Lc = np.stack([colourise(stream[p, 0]) for p in range(pairs)]); Rc = np.stack([colourise(stream[p, 1]) for p in range(pairs)])
    Lg = np.stack([O.gray_from_color(Lc[p], rgb, cv3) for p in range(pairs)]); Rg = np.stack([O.gray_from_color(Rc[p], rgb, cv3) for p in range(pairs)])
    assert np.array_equal(Lg[0], kitti_to_gray(Lc[0], rgb, cv3))                   # the host loader's conversion is the same function
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, pairs, nfeatures=n, enableIntrospection=True, bf=BF, b=B)
    dLc = torch.from_numpy(Lc).to(dev); dRc = torch.from_numpy(Rc).to(dev); dRg = torch.from_numpy(Rg).to(dev); dC = torch.from_numpy(cost).to(dev)
    # reference: the plain run on the oracle-converted grey images
    fe.run(torch.from_numpy(Lg).to(dev), dRg, dC); fe.sync()
    ref = [(fe.fetch(p, 0), fe.fetch(p, 1)) for p in range(pairs)]
    oL = O.Extractor(n, 1.2, 8, 20, 7, introspection=True); oR = O.Extractor(n, 1.2, 8, 20, 7)
    okL, odL = oL(Lg[1], cost[1]); okR, odR = oR(Rg[1], cost[1])
    assert_kps_equal(ref[1][0]["kps"], okL, "grey L vs oracle"); assert np.array_equal(ref[1][0]["desc"], odL)
    assert_kps_equal(ref[1][1]["kps"], okR, "grey R vs oracle"); assert np.array_equal(ref[1][1]["desc"], odR)

    def same(tag):
        for p in range(pairs):
            for side in (0, 1):
                a = fe.fetch(p, side); b_ = ref[p][side]
                assert_kps_equal(a["kps"], b_["kps"], "%s pair %d side %d" % (tag, p, side))
                assert np.array_equal(a["desc"], b_["desc"]) and np.array_equal(a["quality"], b_["quality"])
            assert fe.fetch(p, 0)["uright"].tobytes() == ref[p][0]["uright"].tobytes() and fe.fetch(p, 0)["depth"].tobytes() == ref[p][0]["depth"].tobytes()
    fe.run_color(dLc, dRg, dC, rgb=rgb, cv3=cv3); fe.sync(); same("colour left, grey right")
    fe.run_color(dLc, dRc, dC, rgb=rgb, cv3=cv3); fe.sync(); same("colour left and right")
    # padded views with different strides per side, odd byte offsets
    bigL = torch.zeros((pairs, h + 3, w + 11, 3), dtype=torch.uint8, device=dev); vL = bigL[:, 1:1 + h, 5:5 + w]; vL.copy_(dLc)
    bigR = torch.zeros((pairs, h + 7, w + 29), dtype=torch.uint8, device=dev); vR = bigR[:, 4:4 + h, 3:3 + w]; vR.copy_(dRg)
    bigC = torch.zeros((pairs, h + 2, w + 1), dtype=torch.uint8, device=dev); vC = bigC[:, 1:1 + h, :w]; vC.copy_(dC)
    fe.run_color(vL, vR, vC, rgb=rgb, cv3=cv3); fe.sync(); same("padded views")
    # argument checks
    with pytest.raises(Exception):
        fe._lib.ivf_frontend_run_color.restype
        from iv_slam_amd._lib import check
        check(fe._lib.ivf_frontend_run_color(fe._h, dLc.data_ptr(), 3, dLc.stride()[0], dLc.stride()[1], dRg.data_ptr(), 0, dRg.stride()[0], dRg.stride()[1],
                                             None, 0, 0, pairs, None))
    with pytest.raises(Exception):
        from iv_slam_amd._lib import check
        check(fe._lib.ivf_frontend_run_color(fe._h, dLc.data_ptr(), 1, dLc.stride()[0], w, dRg.data_ptr(), 0, dRg.stride()[0], dRg.stride()[1],
                                             None, 0, 0, pairs, None))          # a colour row needs 3 w bytes


def kitti_to_gray(img, rgb, cv3):
    from iv_slam_amd import kitti
    return kitti.to_gray(img, rgb, cv3)


def test_cost_maps_written_into_the_front_ends_plane_skip_the_ingest(iv):
    """r06: ivf_frontend_cost_plane + ivf_fcn_forward_device_strided -- the FCN writes its u8 cost maps into the pitched level-0 cost plane of the batch
    context the next run uses, and the run skips their ingest.  Five runs through the three contexts (every plane is reused) give exactly the results of the
    plain path (FCN -> contiguous buffer -> ivf_frontend_run ingests it) on the same inputs, and equal the oracle chain on the FCN's map."""
    import torch
    from iv_slam_amd import fcn_weights
    w, h, n, pairs = 640, 240, 400, 3
    dev = torch.device("cuda:0")
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(fcn_weights.make_seeded_weights(11)), (h, w), (h, w), max_batch=pairs)
    fe = iv.StereoFrontend(w, h, pairs, nfeatures=n, enableIntrospection=True, bf=BF, b=B)
    ref_fe = iv.StereoFrontend(w, h, pairs, nfeatures=n, enableIntrospection=True, bf=BF, b=B)
    st = torch.cuda.current_stream(dev)
    for it in range(5):
        stream = synth.make_stream(pairs, w, h, seed=200 + it)
        L = torch.from_numpy(stream[:, 0].copy()).to(dev); R = torch.from_numpy(stream[:, 1].copy()).to(dev)
        bgr = torch.stack([L, L // 2 + 40, 255 - L // 2], dim=-1).contiguous()
        plane = fe.cost_plane(pairs, st.cuda_stream)
        assert not plane.is_contiguous() and tuple(plane.shape) == (pairs, h, w)
        fcn.forward_device(bgr, cost_u8=plane, stream_ptr=st.cuda_stream)
        fe.run_color(L, R, plane, st.cuda_stream)
        cost = torch.empty((pairs, h, w), dtype=torch.uint8, device=dev)
        fcn.forward_device(bgr, cost_u8=cost, stream_ptr=st.cuda_stream)
        ref_fe.run(L, R, cost, st.cuda_stream)
        fe.sync(); ref_fe.sync(); torch.cuda.synchronize()
        assert torch.equal(plane, cost)                                   # the same map, through the strides
        for p in range(pairs):
            for side in (0, 1):
                a = fe.fetch(p, side); b_ = ref_fe.fetch(p, side)
                assert_kps_equal(a["kps"], b_["kps"], "run %d pair %d side %d" % (it, p, side))
                assert np.array_equal(a["desc"], b_["desc"]) and np.array_equal(a["quality"], b_["quality"])
            assert fe.fetch(p, 0)["uright"].tobytes() == ref_fe.fetch(p, 0)["uright"].tobytes()
        if it == 4:
            hc = cost.cpu().numpy()
            okL, odL = O.Extractor(n, 1.2, 8, 20, 7, introspection=True)(stream[1, 0], hc[1])
            r = fe.fetch(1, 0)
            assert_kps_equal(r["kps"], okL, "vs oracle"); assert np.array_equal(r["desc"], odL)
            assert len(okL) > 100


def test_two_front_ends_from_two_host_threads(iv):
    """two independent handles driven concurrently from two host threads (a two-camera rig): same results as one after the
    other (per-thread scratch, no shared mutable state in the library; r05: both handles enqueue on the SAME three pooled internal streams)"""
    import threading
    import torch
    w, h, n, pairs = 640, 240, 500, 2
    dev = torch.device("cuda:0")
    streams = [synth.make_stream(pairs, w, h, seed=s) for s in (201, 202)]
    ten = [(torch.from_numpy(st[:, 0].copy()).to(dev), torch.from_numpy(st[:, 1].copy()).to(dev)) for st in streams]
    fes = [iv.StereoFrontend(w, h, pairs, nfeatures=n, bf=BF, b=B) for _ in range(2)]
    ext = [iv.ORBextractor(n, 1.2, 8, 20, 7) for _ in range(2)]
    ref = []
    for k in range(2):
        fes[k].run(*ten[k]); fes[k].sync()
        ref.append(([fes[k].fetch(p, 0) for p in range(pairs)], ext[k](streams[k][0, 0])))
    out = [None, None]; err = []

    def work(k):
        try:
            for _ in range(8):
                fes[k].run(*ten[k]); fes[k].sync()
                a = [fes[k].fetch(p, 0) for p in range(pairs)]
                b_ = ext[k](streams[k][0, 0])
            out[k] = (a, b_)
        except Exception as e:                     # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not err, err
    for k in range(2):
        for p in range(pairs):
            assert out[k][0][p]["kps"].tobytes() == ref[k][0][p]["kps"].tobytes()
            assert np.array_equal(out[k][0][p]["desc"], ref[k][0][p]["desc"])
            assert out[k][0][p]["uright"].tobytes() == ref[k][0][p]["uright"].tobytes()
        assert out[k][1][0].tobytes() == ref[k][1][0].tobytes() and np.array_equal(out[k][1][1], ref[k][1][1])
