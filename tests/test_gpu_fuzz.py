"""Seeded random geometries / parameters through the per-call extractor and the batched stereo front end against the CPU
oracle (bit-exact bar).  The fixed-size parity tests cover the benchmark shapes; this one walks the corners of the geometry
code: odd widths and pitches, planes whose upper levels vanish, level counts and scale factors other than 8 / 1.2, thresholds,
feature counts far below and above what the image yields, cost maps on and off."""
import os

import numpy as np
import pytest

import oracle_lib as O
from iv_slam_amd import synth
from test_gpu_parity import assert_kps_equal, iv  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _case(rng):
    w = int(rng.integers(97, 1400)); h = int(rng.integers(81, 620))
    n = int(rng.choice([60, 200, 500, 1000, 2500, 6000]))
    nlevels = int(rng.integers(1, 9))
    sf = float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0]))
    ini = int(rng.choice([10, 20, 35])); mn = int(rng.choice([3, 7, ini]))
    return w, h, n, nlevels, sf, ini, min(mn, ini)


IVF_E_GEOMETRY = -3
SEED0 = int(os.environ.get("IVF_FUZZ_SEED0", "0"))          # long runs on seeds no earlier round has seen: IVF_FUZZ_SEED0=100000 IVF_FUZZ_EXTRACT=2000 ...


def _seeds(var, default):
    return range(SEED0, SEED0 + int(os.environ.get(var, default)))


def _valid_case(iv, rng, intro):
    """draws cases until one has a cell grid the reference can run: where a cell window would leave its level the reference
    throws in rowRange / colRange (ORBextractor.cc:1033); product and oracle must BOTH report that geometry as unsupported."""
    from iv_slam_amd._lib import IvfError
    for _ in range(40):
        w, h, n, nlevels, sf, ini, mn = _case(rng)
        img = synth.make_left(w, h, seed=int(rng.integers(0, 1 << 20)), idx=0)
        cost = synth.make_cost_map(w, h, seed=3, idx=1) if intro else None
        g = iv.ORBextractor(n, sf, nlevels, ini, mn, intro)
        o = O.Extractor(n, sf, nlevels, ini, mn, intro)
        try:
            gk, gd = g(img, cost)
        except IvfError as e:
            assert e.code == IVF_E_GEOMETRY, str(e)
            with pytest.raises(RuntimeError):
                o(img, cost)
            continue
        if intro:
            # rows that overshoot a level are defined only on the introspection path (stale hY): without the cost map both sides
            # must refuse exactly when the oracle does
            try:
                o(img, None); oracle_ok = True
            except RuntimeError:
                oracle_ok = False
            try:
                g(img, None); product_ok = True
            except IvfError as e:
                assert e.code == IVF_E_GEOMETRY, str(e); product_ok = False
            assert oracle_ok == product_ok, "%dx%d n=%d levels=%d sf=%.1f without cost map: oracle %s, product %s" % (w, h, n, nlevels, sf, oracle_ok, product_ok)
            gk, gd = g(img, cost)
        return (w, h, n, nlevels, sf, ini, mn), img, cost, g, o, gk, gd
    raise AssertionError("no valid geometry in 40 draws")


@pytest.mark.parametrize("seed", _seeds("IVF_FUZZ_EXTRACT", "10"))
def test_extract_random_geometry(iv, seed):
    rng = np.random.default_rng(1000 + seed)
    intro = bool(seed & 1)
    (w, h, n, nlevels, sf, ini, mn), img, cost, g, o, gk, gd = _valid_case(iv, rng, intro)
    ok, od = o(img, cost)
    what = "%dx%d n=%d levels=%d sf=%.1f th=%d/%d intro=%d" % (w, h, n, nlevels, sf, ini, mn, intro)
    assert g.level_counts() == o.level_counts(), what
    assert_kps_equal(gk, ok, what)
    assert np.array_equal(gd, od), what
    for l in range(nlevels):
        assert np.array_equal(g.mvImagePyramid[l], o.pyramid(l)), "%s: pyramid level %d" % (what, l)
    if seed % 3 == 2:                                   # the same geometry on a noisy image: ties, dense corners
        noisy = np.clip(img.astype(np.int32) + rng.integers(-40, 41, img.shape), 0, 255).astype(np.uint8)
        gk, gd = g(noisy, cost); ok, od = o(noisy, cost)
        assert_kps_equal(gk, ok, what + " noisy"); assert np.array_equal(gd, od), what


@pytest.mark.parametrize("seed", _seeds("IVF_FUZZ_FRONTEND", "4"))
def test_frontend_random_geometry(iv, seed):
    import torch
    rng = np.random.default_rng(2000 + seed)
    intro = bool(seed & 1)
    (w, h, n, nlevels, sf, ini, mn), _img, _cost, _g, _o, _gk, _gd = _valid_case(iv, rng, False)   # the right side never sees a cost map
    pairs = 2
    bf = 386.1448; b = bf / 718.856
    stream = synth.make_stream(pairs, w, h, seed=70 + seed)
    cost = np.stack([synth.make_cost_map(w, h, seed=70 + seed, idx=i) for i in range(pairs)]) if intro else None
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, pairs, nfeatures=n, scaleFactor=sf, nlevels=nlevels, iniThFAST=ini, minThFAST=mn,
                           enableIntrospection=intro, bf=bf, b=b)
    fe.run(torch.from_numpy(stream[:, 0].copy()).to(dev), torch.from_numpy(stream[:, 1].copy()).to(dev),
           None if cost is None else torch.from_numpy(cost).to(dev))
    fe.sync()
    what = "%dx%d n=%d levels=%d sf=%.1f th=%d/%d intro=%d" % (w, h, n, nlevels, sf, ini, mn, intro)
    for p in range(pairs):
        oL = O.Extractor(n, sf, nlevels, ini, mn, intro); oR = O.Extractor(n, sf, nlevels, ini, mn, False)
        okL, odL = oL(stream[p, 0], None if cost is None else cost[p]); okR, odR = oR(stream[p, 1], None)
        our, odp = O.stereo_match(oL, oR, okL, odL, okR, odR, bf, b)
        rl = fe.fetch(p, 0); rr = fe.fetch(p, 1)
        assert_kps_equal(rl["kps"], okL, what + " L%d" % p); assert_kps_equal(rr["kps"], okR, what + " R%d" % p)
        assert np.array_equal(rl["desc"], odL) and np.array_equal(rr["desc"], odR), what
        assert rl["uright"].tobytes() == our.tobytes() and rl["depth"].tobytes() == odp.tobytes(), what


@pytest.mark.parametrize("seed", _seeds("IVF_FUZZ_FCN", "3"))
def test_fcn_random_sizes_and_batches(iv, seed):
    """the FCN's input / output stages (bilinear to 512 x 512, bilinear to out_size, logistic, u8 truncation) at sizes other
    than the two benchmark shapes, batched, against the torch-CPU layer list (1e-3 bar; u8 within one LSB)."""
    import torch
    import fcn_common as FC
    import fcn_oracle_torch as FT
    from iv_slam_amd import fcn_weights
    rng = np.random.default_rng(3000 + seed)
    g, W, _bgr, _ = FC.load_case(["kitti", "jackal_smallw", "kitti_bigw"][seed % 3])     # weights with a calibrated conv_last
    ih, iw = int(rng.integers(33, 900)), int(rng.integers(33, 1500))
    oh, ow = (ih, iw) if seed % 2 == 0 else (int(rng.integers(17, 700)), int(rng.integers(17, 1300)))
    nb = int(rng.integers(1, 4))
    imgs = np.stack([FC.bgr_image(iw, ih, 900 + 7 * seed + k) for k in range(nb)])
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), (ih, iw), (oh, ow), max_batch=nb)
    dev = torch.device("cuda:0")
    cu8 = torch.empty((nb, oh, ow), dtype=torch.uint8, device=dev); cf = torch.empty((nb, oh, ow), dtype=torch.float32, device=dev)
    fcn.forward_device(torch.from_numpy(imgs).to(dev), cost_u8=cu8, cost_f32=cf)
    torch.cuda.synchronize()
    T = FT.prepare(W)
    oc, ou8 = FT.forward(T, imgs, (oh, ow))
    err = float(np.abs(cf.cpu().numpy() - oc).max())
    assert err < 1e-3, "in %dx%d out %dx%d batch %d: %.3g" % (iw, ih, ow, oh, nb, err)
    d = np.abs(cu8.cpu().numpy().astype(int) - ou8.astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.02
    u8_single = fcn(imgs[0])                             # the per-call host path = a batch of one: the first slot of this batch up to
    ds = np.abs(u8_single.astype(int) - cu8[0].cpu().numpy().astype(int))      # the summation order of the small-batch schedule (nb > 1)
    assert ds.max() <= (0 if nb == 1 else 1) and (ds != 0).mean() < 0.01


@pytest.mark.parametrize("seed", _seeds("IVF_FUZZ_SEARCH", "4"))
def test_window_searches_random_scenarios(iv, seed):
    """SearchByProjection(cur,last) / SearchByProjection(F, mapPoints) on a resident frame, through the host-grid entry point and in
    the oracle, on random frame sizes, feature counts, radii and level windows (incl. windows off the grid, dense windows that
    overflow the device-side candidate lists, empty query sets)."""
    import test_gpu_frame as TF
    from iv_slam_amd._lib import IvfError
    rng = np.random.default_rng(4000 + seed)
    radius = float(rng.choice([1.0, 4.0, 7.0, 15.0, 60.0]))
    big = bool(rng.integers(0, 2))
    for _ in range(40):                                 # the scenario's keypoints come from the extractor: skip unsupported cell grids
        w = int(rng.integers(200, 1300)); h = int(rng.integers(120, 500))
        n = int(rng.choice([40, 300, 1000, 2500]))
        try:
            kps, desc, uright, bounds, q, pre = TF._case(iv, 500 + seed, n, w, h, radius=radius, big=big)
            break
        except IvfError as e:
            assert e.code == IVF_E_GEOMETRY
        except ValueError:                              # fewer than three keypoints: the scenario builder needs three
            pass
    if seed % 5 == 4:                                   # no query survives
        q["valid"][:] = 0
    f = iv.DeviceFrame(kps, desc, uright, bounds)
    check_ori = bool(rng.integers(0, 2))
    m = iv.ORBmatcher(float(rng.choice([0.6, 0.75, 0.9])), check_ori)
    what = "%dx%d n=%d r=%.0f big=%d" % (w, h, len(kps), radius, big)
    for chk in (True, False):
        ga, gn = f.SearchByProjection(q, chk, pre)
        oa, on = O.search_by_projection(kps, desc, uright, bounds, q, chk, pre)
        assert gn == on and np.array_equal(ga, oa), what
    ha, hn = m.SearchByProjection(kps, desc, uright, bounds, q, pre)
    ga, gn = f.SearchByProjection(q, check_ori, pre)
    assert hn == gn and np.array_equal(ha, ga), what
    qm = dict(u=q["u"], v=q["v"], ur=q["ur"], radius=q["radius"],
              level=np.clip(kps["octave"] + rng.integers(-1, 2, len(kps)), 0, 7).astype(np.int32),
              desc=q["desc"], valid=q["valid"], blocks=q["blocks"])
    for ratio in (0.6, 1.0):
        ga, gn = f.SearchByProjectionMapPoints(qm, ratio, pre)
        oa, on = O.search_map_points(kps, desc, uright, bounds, qm, ratio, pre)
        assert gn == on and np.array_equal(ga, oa), what


@pytest.mark.parametrize("seed", _seeds("IVF_FUZZ_ALLSEARCH", "3"))
def test_remaining_searches_random_scenarios(iv, seed):
    """SearchForInitialization, ComputeDistinctiveDescriptors, SearchByProjection(KF, Scw), Fuse x2, SearchBySim3, SearchByBoW x2,
    SearchForTriangulation and the relocalisation search on two-frame scenarios of random size / feature count / seed: equality
    with the oracle only (the yield checks belong to the fixed scenario of tests/test_gpu_parity.py)."""
    import test_gpu_parity as TP
    from iv_slam_amd._lib import IvfError
    rng = np.random.default_rng(5000 + seed)
    for _ in range(40):
        w = int(rng.integers(160, 1300)); h = int(rng.integers(120, 500)); n = int(rng.choice([100, 400, 800, 2000]))
        try:
            if TP.all_searches_scenario(iv, 6000 + seed, w, h, n, 40 + seed, False):
                return
        except IvfError as e:
            assert e.code == IVF_E_GEOMETRY
    raise AssertionError("no usable scenario in 40 draws")


@pytest.mark.parametrize("pattern", ["noise", "checker2", "checker3", "stripes", "ramp", "blobs", "salt", "pink", "foliage", "pink-steep", "foliage-jackal"])
def test_extract_stress_patterns(iv, pattern):
    """image content at the extremes: dense corners (survivor-list / cell capacities, the plane-scan NMS path, ties everywhere),
    none at all, saturated blobs -- KITTI-sized, 1000 and 4000 features, with and without a cost map.  r06: natural-image statistics
    (synth.make_natural: 1 / f amplitude spectrum = corners at every scale and no flat region; thresholded occlusion structure = the corner
    density of foliage), also at the Jackal size of configs[4]"""
    w, h = (1920, 1200) if pattern.endswith("jackal") else (1242, 375)
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:h, 0:w]
    if pattern == "noise":
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
    elif pattern == "checker2":
        img = ((((yy // 2) + (xx // 2)) & 1) * 255).astype(np.uint8)
    elif pattern == "checker3":
        img = ((((yy // 3) + (xx // 3)) & 1) * 200 + 20).astype(np.uint8)
    elif pattern == "stripes":
        img = (((xx // 5) & 1) * 180 + 30).astype(np.uint8)
    elif pattern == "ramp":
        img = ((xx * 255) // (w - 1)).astype(np.uint8)
    elif pattern == "blobs":
        img = np.zeros((h, w), np.uint8)
        for _ in range(300):
            cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(2, 9)
            img[max(cy - r, 0):cy + r, max(cx - r, 0):cx + r] = 255
    elif pattern == "pink":
        img = synth.make_natural(w, h, seed=21, idx=0, beta=1.0)
    elif pattern == "pink-steep":
        img = synth.make_natural(w, h, seed=22, idx=0, beta=1.5)
    elif pattern.startswith("foliage"):
        img = synth.make_natural(w, h, seed=23, idx=0, beta=1.0, foliage=True)
    else:
        img = np.full((h, w), 128, np.uint8)
        idx = rng.integers(0, h * w, 20000)
        img.reshape(-1)[idx] = rng.choice([0, 255], 20000).astype(np.uint8)
    cost = synth.make_cost_map(w, h, seed=9, idx=0)
    for n, intro in ((1000, False), (4000, True)):
        g = iv.ORBextractor(n, 1.2, 8, 20, 7, intro)
        o = O.Extractor(n, 1.2, 8, 20, 7, intro)
        gk, gd = g(img, cost if intro else None)
        ok, od = o(img, cost if intro else None)
        assert g.level_counts() == o.level_counts(), (pattern, n)
        assert_kps_equal(gk, ok, "%s n=%d" % (pattern, n))
        assert np.array_equal(gd, od), (pattern, n)


@pytest.mark.parametrize("pattern", ["noise", "checker3", "blobs", "foliage"])
def test_stereo_stress_patterns(iv, pattern):
    """the stereo matcher on repetitive / noisy content (Hamming ties, many candidates per row, SAD plateaus): the batched front end
    against the oracle chain, right image = left shifted by a few pixels"""
    import torch
    w, h, n = 800, 300, 1500
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:h, 0:w + 16]
    if pattern == "noise":
        base = rng.integers(0, 256, (h, w + 16)).astype(np.uint8)
    elif pattern == "checker3":
        base = ((((yy // 3) + (xx // 3)) & 1) * 200 + 20).astype(np.uint8)
    elif pattern == "foliage":
        base = synth.make_natural(w + 16, h, seed=24, idx=0, beta=1.0, foliage=True)
    else:
        base = np.zeros((h, w + 16), np.uint8)
        for _ in range(400):
            cy, cx, r = rng.integers(0, h), rng.integers(0, w + 16), rng.integers(2, 9)
            base[max(cy - r, 0):cy + r, max(cx - r, 0):cx + r] = rng.integers(60, 256)
    left = base[:, 16:16 + w].copy(); right = base[:, 9:9 + w].copy()       # disparity 7
    bf = 386.1448; b = bf / 718.856
    dev = torch.device("cuda:0")
    fe = iv.StereoFrontend(w, h, 1, nfeatures=n, bf=bf, b=b)
    fe.run(torch.from_numpy(left[None].copy()).to(dev), torch.from_numpy(right[None].copy()).to(dev))
    fe.sync()
    oL = O.Extractor(n, 1.2, 8, 20, 7); oR = O.Extractor(n, 1.2, 8, 20, 7)
    okL, odL = oL(left); okR, odR = oR(right)
    our, odp = O.stereo_match(oL, oR, okL, odL, okR, odR, bf, b)
    rl = fe.fetch(0, 0); rr = fe.fetch(0, 1)
    assert_kps_equal(rl["kps"], okL, pattern + " L"); assert_kps_equal(rr["kps"], okR, pattern + " R")
    assert np.array_equal(rl["desc"], odL) and np.array_equal(rr["desc"], odR)
    assert rl["uright"].tobytes() == our.tobytes() and rl["depth"].tobytes() == odp.tobytes(), pattern


@pytest.fixture(scope="module")
def adapter_driver(tmp_path_factory):
    import adapter_scenario as AS
    return AS.build_driver(tmp_path_factory.mktemp("adapter_fuzz") / "adapter_driver")


@pytest.mark.parametrize("seed", _seeds("IVF_FUZZ_ADAPTER", "3"))
def test_adapter_random_scenarios(iv, adapter_driver, tmp_path, seed):
    """the compiled C++ adapter (all eleven ORBmatcher signatures on mock Frame / KeyFrame / MapPoint types) on further seeded
    scenarios: every result equal to the projection oracle (the yield checks belong to tests/test_gpu_adapter.py)"""
    import subprocess
    import adapter_scenario as AS
    S, blob = AS.make(O, synth, 100 + seed, bool(seed & 1))
    (tmp_path / "s.bin").write_bytes(blob)
    r = subprocess.run([adapter_driver, str(tmp_path / "s.bin"), str(tmp_path / "r.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "r.bin", np.int32).astype(np.int64)
    want, _counts = AS.expected(O, S)
    assert got.shape == want.shape
    assert np.array_equal(got, want), "seed %d: first difference at %d" % (seed, int(np.nonzero(got != want)[0][0]))
