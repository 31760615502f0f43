"""GPU parity of the rectification step: ivf_init_undistort_rectify_map / ivf_remap_* through the
C-ABI against the CPU oracle.  Bar: bit-exact maps (same double arithmetic) and bit-exact remapped images."""
import numpy as np
import pytest

import oracle_lib as O
from iv_slam_amd import synth
from test_oracle_rectify import K_L, D_L, R_L, P_L

pytestmark = pytest.mark.gpu

K_R = [530.158021, 0.0, 475.540633, 0.0, 529.682234, 299.995465, 0.0, 0.0, 1.0]
D_R = [-0.156833, 0.081841, -0.000779, -0.000356, 0.0]
R_R = [0.999661, -0.024534, 0.008699, 0.024595, 0.999673, -0.006974, -0.008525, 0.007186, 0.999938]
P_R = [528.955512, 0.0, 479.748173, -69.690815, 0.0, 528.955512, 298.607571, 0.0, 0.0, 0.0, 1.0, 0.0]


@pytest.fixture(scope="module")
def iv():
    import iv_slam_amd
    lib = iv_slam_amd.load()
    assert lib.ivf_device_count() >= 1, "no HIP device: libivfront has no CPU fallback"
    return iv_slam_amd


def test_maps_bit_exact(iv):
    for (K, D, R, P) in [(K_L, D_L, R_L, P_L), (K_R, D_R, R_R, P_R), (K_L, None, R_L, P_L), (K_L, D_L, None, K_L),
                         (K_L, D_L[:4], R_L, P_L), (K_R, D_R + [0.01, -0.02, 0.003], R_R, P_R),
                         (K_R, D_R + [0.01, -0.02, 0.003, 1e-4, -2e-4, 3e-4, -1e-4], R_R, P_R)]:
        g1, g2 = iv.initUndistortRectifyMap(K, D, R, P, (960, 600))
        o1, o2 = O.init_undistort_rectify_map(K, D, R, P, (960, 600))
        assert g1.tobytes() == o1.tobytes() and g2.tobytes() == o2.tobytes()
    with pytest.raises(iv.IvfError):
        iv.initUndistortRectifyMap(K_L, [0.1] * 14, None, K_L, (64, 48))          # tilt terms: refused, not ignored
    with pytest.raises(iv.IvfError):
        iv.initUndistortRectifyMap(K_L, None, None, [1, 0, 0, 2, 0, 0, 0, 0, 1], (64, 48))   # singular P


def test_fixed_maps_and_remap_random_maps(iv):
    rng = np.random.default_rng(5)
    for (sh, sw, h, w, cn) in [(64, 80, 50, 71, 1), (120, 160, 120, 160, 1), (97, 131, 63, 250, 3), (33, 35, 5, 3, 1)]:
        shape = (sh, sw) if cn == 1 else (sh, sw, 3)
        img = rng.integers(0, 256, size=shape).astype(np.uint8)
        m1 = (rng.random((h, w)) * (sw + 8) - 4).astype(np.float32)
        m2 = (rng.random((h, w)) * (sh + 8) - 4).astype(np.float32)
        m1[0, :3] = [-1.0, sw - 0.5, 1e6]; m2[0, :3] = [0, 0, -1e6]
        m1[1, :2] = [3 + 1 / 64, 5 + 3 / 64]; m2[1, :2] = [2.5, 7 + 1 / 64]     # ties of the 1/32-px rounding
        r = iv.Remap(m1, m2, (sh, sw), cn)
        xy, al = r.fixed_maps()
        fx = np.rint(m1 * np.float32(32)).astype(np.int64); fy = np.rint(m2 * np.float32(32)).astype(np.int64)
        assert np.array_equal(xy[..., 0], np.clip(fx >> 5, -32768, 32767)) and np.array_equal(xy[..., 1], np.clip(fy >> 5, -32768, 32767))
        assert np.array_equal(al, ((fy & 31) << 5) | (fx & 31))
        assert np.array_equal(r(img), O.remap_bilinear(img, m1, m2)), (sh, sw, h, w, cn)
    with pytest.raises(iv.IvfError):
        iv.Remap(m1, m2, (40000, 10), 1)
    with pytest.raises(iv.IvfError):
        iv.Remap(m1, m2, (10, 10), 2)


def test_rectified_stereo_pair_full_size(iv):
    """Jackal-shaped 960x600 pair: maps from the calibration, remap on the device, then the extractor -- the whole
    chain equals the oracle chain bit for bit."""
    L, R = synth.make_pair(960, 600, seed=31, idx=1)
    for img, (K, D, Rm, P) in ((L, (K_L, D_L, R_L, P_L)), (R, (K_R, D_R, R_R, P_R))):
        m1, m2 = iv.initUndistortRectifyMap(K, D, Rm, P, (960, 600))
        g = iv.Remap(m1, m2, img.shape)(img)
        o = O.remap_bilinear(img, m1, m2)
        assert np.array_equal(g, o)
        assert (g[:, :3] == 0).any() or (g[:3] == 0).any() or True          # borders may or may not be visible; no assumption
        gk, gd = iv.ORBextractor(1200, 1.2, 8, 12, 7)(g)
        ok, od = O.Extractor(1200, 1.2, 8, 12, 7)(o)
        assert gk.tobytes() == ok.tobytes() and np.array_equal(gd, od)


def test_remap_device_batch(iv):
    import torch
    rng = np.random.default_rng(8)
    h, w = 375, 1242
    m1, m2 = iv.initUndistortRectifyMap([718.856, 0, 607.1928, 0, 718.856, 185.2157, 0, 0, 1], [-0.2, 0.05, 1e-3, -1e-3, 0.0],
                                        None, [718.856, 0, 607.1928, 0, 718.856, 185.2157, 0, 0, 1], (w, h))
    imgs = rng.integers(0, 256, size=(6, h, w)).astype(np.uint8)
    r = iv.Remap(m1, m2, (h, w))
    out = r.apply_device(torch.from_numpy(imgs).cuda())
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for k in range(len(imgs)):
        assert np.array_equal(out[k], O.remap_bilinear(imgs[k], m1, m2))
    # size-independent property: a map of integer positions is a pure gather (weights {32767, 0, 0, 1} never change a value)
    perm_x = rng.permutation(w).astype(np.float32)[None, :].repeat(h, 0); perm_y = rng.permutation(h).astype(np.float32)[:, None].repeat(w, 1)
    g = iv.Remap(perm_x, perm_y, (h, w))(imgs[0])
    assert np.array_equal(g, imgs[0][perm_y.astype(int), perm_x.astype(int)])


def test_replay_harness_matches_oracle_chain(iv, tmp_path):
    """tools/replay_kitti.py on a synthetic KITTI-layout sequence (PNG files, times.txt, settings.yaml, cost images for
    two frames out of three): load -> rectify on the device -> extract L/R (+ cost map) -> stereo match equals the oracle
    chain frame by frame."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import replay_kitti
    from iv_slam_amd import kitti
    seq = str(tmp_path / "seq")
    settings = replay_kitti.make_synthetic(seq, 6)
    S = kitti.Settings.load(settings)
    left, right, ts = kitti.LoadImages(seq)
    qual, found = kitti.GetImageQualFileNames(os.path.join(seq, "qual"), len(ts))
    assert len(ts) == 6 and found == 4
    nf, sf, nl, ini, mn, _ = S.extractor_params(); bf, b = S.stereo()
    maps = {}
    for side in ("LEFT", "RIGHT"):
        K, D, R, P, size = S.rectification(side)
        maps[side] = O.init_undistort_rectify_map(K, D, R, P, size)
    for introspect in (False, True):
        rp = replay_kitti.Replay(S, rectify=True, undistort=True, introspect=introspect, batch=1 if introspect else 4)
        for i0 in range(0, 6, rp.batch):
            idx = list(range(i0, min(i0 + rp.batch, 6)))
            Ls = [kitti.imread(left[i]) for i in idx]; Rs = [kitti.imread(right[i]) for i in idx]
            Cs = [kitti.imread(qual[i]) if qual[i] else None for i in idx] if introspect else None
            res = rp.run(Ls, Rs, Cs)
            for k, i in enumerate(idx):
                oL = O.remap_bilinear(Ls[k], *maps["LEFT"]); oR = O.remap_bilinear(Rs[k], *maps["RIGHT"])
                oC = O.remap_bilinear(Cs[k], *maps["LEFT"]) if (introspect and Cs[k] is not None) else None
                eL = O.Extractor(nf, sf, nl, ini, mn, introspection=introspect); eR = O.Extractor(nf, sf, nl, ini, mn)
                okL, odL = eL(oL, oC); okR, odR = eR(oR)
                our, odp = O.stereo_match(eL, eR, okL, odL, okR, odR, bf, b)
                l, r = res[k]
                assert l["kps"].tobytes() == okL.tobytes() and np.array_equal(l["desc"], odL), (introspect, i)
                assert r["kps"].tobytes() == okR.tobytes() and np.array_equal(r["desc"], odR), (introspect, i)
                assert l["uright"].tobytes() == our.tobytes() and l["depth"].tobytes() == odp.tobytes(), (introspect, i)


def test_replay_harness_cross_frame_tracking(iv, tmp_path):
    """--track: after each batch the tracker's matcher call (SearchByProjection(cur, last), zero-motion prior) is replayed on
    device-resident frames made straight from the batch, also across batch boundaries; equals the oracle chain
    (oracle extraction + stereo of both frames -> oracle SearchByProjection) frame by frame."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import replay_kitti
    from iv_slam_amd import kitti
    seq = str(tmp_path / "seq")
    settings = replay_kitti.make_synthetic(seq, 5, with_qual=False)
    S = kitti.Settings.load(settings)
    left, right, ts = kitti.LoadImages(seq)
    nf, sf, nl, ini, mn, _ = S.extractor_params(); bf, b = S.stereo()
    rp = replay_kitti.Replay(S, batch=2, track=True)
    w, h = rp.size
    prev = None
    for i0 in range(0, 5, 2):
        idx = list(range(i0, min(i0 + 2, 5)))
        Ls = [kitti.imread(left[i]) for i in idx]; Rs = [kitti.imread(right[i]) for i in idx]
        res = rp.run(Ls, Rs)
        for k, i in enumerate(idx):
            eL = O.Extractor(nf, sf, nl, ini, mn); eR = O.Extractor(nf, sf, nl, ini, mn)
            okL, odL = eL(Ls[k]); okR, odR = eR(Rs[k])
            our, _ = O.stereo_match(eL, eR, okL, odL, okR, odR, bf, b)
            cur = dict(kps=okL, desc=odL, uright=our)
            assign, nm = res[k][0]["tracked"]
            if prev is None:
                assert nm == 0 and (assign == -1).all()
            else:
                oa, on = O.search_by_projection(okL, odL, our, (0.0, 0.0, float(w), float(h)), rp.track_queries(prev), True)
                assert nm == on and np.array_equal(assign, oa), i
            prev = cur
    # the synthetic frames are independent scenes, so matches are few: identical frames must re-match almost everything
    rp2 = replay_kitti.Replay(S, batch=2, track=True)
    L0 = kitti.imread(left[0]); R0 = kitti.imread(right[0])
    res = rp2.run([L0, L0], [R0, R0])
    nst = int((res[0][0]["uright"] >= 0).sum())
    assert res[1][0]["tracked"][1] > 0.9 * nst > 50


def test_replay_harness_online_network(iv, tmp_path):
    """--fcn: the introspection network runs on the un-remapped left frame, its cost map is remapped like the left image
    and gates the left extractor (stereo_kitti.cc:493-521); equals FCN (device) -> oracle remap -> oracle extractor."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import replay_kitti
    from iv_slam_amd import kitti, fcn_weights
    seq = str(tmp_path / "seq")
    S = kitti.Settings.load(replay_kitti.make_synthetic(seq, 2, with_qual=False))
    left, right, ts = kitti.LoadImages(seq)
    blob = fcn_weights.pack_blob(fcn_weights.make_seeded_weights(5))
    rp = replay_kitti.Replay(S, rectify=True, undistort=True, batch=2, fcn_blob=blob)
    Ls = [kitti.imread(p) for p in left]; Rs = [kitti.imread(p) for p in right]
    res = rp.run(Ls, Rs, None, Ls)
    nf, sf, nl, ini, mn, _ = S.extractor_params(); bf, b = S.stereo()
    K, D, R, P, size = S.rectification("LEFT"); mL = O.init_undistort_rectify_map(K, D, R, P, size)
    K, D, R, P, size = S.rectification("RIGHT"); mR = O.init_undistort_rectify_map(K, D, R, P, size)
    # the cost maps the way the replay computed them: ONE batch of two images (the small-batch schedule depends on the batch size);
    # the device network itself is checked in test_gpu_fcn.py
    import torch
    fcn = iv.IntrospectionFCN(blob, Ls[0].shape, Ls[0].shape, max_batch=2)
    dcost = torch.empty((2,) + tuple(Ls[0].shape), dtype=torch.uint8, device="cuda:0")
    fcn.forward_device(torch.from_numpy(np.stack([np.repeat(Ls[k][..., None], 3, axis=2) for k in range(2)])).to("cuda:0"), cost_u8=dcost)
    torch.cuda.synchronize()
    costs = dcost.cpu().numpy()
    for k in range(2):
        cost = costs[k]
        oL = O.remap_bilinear(Ls[k], *mL); oR = O.remap_bilinear(Rs[k], *mR); oC = O.remap_bilinear(cost, *mL)
        eL = O.Extractor(nf, sf, nl, ini, mn, introspection=True); eR = O.Extractor(nf, sf, nl, ini, mn)
        okL, odL = eL(oL, oC); okR, odR = eR(oR)
        l, r = res[k]
        assert l["kps"].tobytes() == okL.tobytes() and np.array_equal(l["desc"], odL)
        assert r["kps"].tobytes() == okR.tobytes() and np.array_equal(r["desc"], odR)
