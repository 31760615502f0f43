"""Batched Tracking::SearchLocalPoints (ivf_tracker_search_local) against the oracle: Frame::isInFrustum of every local map point
(ORB/src/Frame.cc:557-613, MapPoint::PredictScale MapPoint.cc:407-422 with glibc's logf) and
ORBmatcher::SearchByProjection(F, vpMapPoints, th) (ORB/src/ORBmatcher.cc:45-135): windows on levels [level - 1, level], stereo
check, best / second best in candidate order, the ratio test inside one octave, and the order-dependent occupancy rule.
Oracle = oracle/projection_oracle.search_local_points_frame (numpy projection around the C oracle's orc_search_map_points).
Bar: the map point every keypoint received and nmatches IDENTICAL for every frame."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from test_gpu_track import extracted_sequence, frame_dict, pose, iv  # noqa: F401  (iv: the module fixture)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
F = np.float32


def map_points_from(cam, rec, T, rng, flip_bits=6, dup=0.15, jitter=0.02):
    """A local map seen from another frame: the stereo points of `rec` (pose T), with the fields UpdateNormalAndDepth leaves
    (MapPoint.cc:330-376: mfMaxDistance = dist * scale[level], mfMinDistance = mfMaxDistance / scale[nLevels - 1], normal = unit
    vector from the reference camera to the point), descriptors with a few flipped bits, some points doubled (rivals for one
    keypoint), random skip / no-observation flags, in shuffled order."""
    import projection_oracle as PO
    fr = frame_dict(rec, T, cam)
    Ow = PO.neg_rt_mul(T[:3, :3], T[:3, 3])
    pts = []
    for i in range(len(rec["kps"])):
        if not rec["depth"][i] > 0:
            continue
        Pw = PO.unproject_stereo(fr, i)
        d = (Pw - Ow).astype(np.float64)
        dist = float(np.sqrt((d * d).sum()))
        if not dist > 0.1:
            continue
        lvl = int(rec["kps"]["octave"][i])
        maxd = F(F(dist) * cam["scale"][lvl]); mind = F(maxd / cam["scale"][-1])
        n = (d / dist + rng.normal(0, 0.2, 3)); n /= np.linalg.norm(n)
        desc = rec["desc"][i].copy()
        for b in rng.integers(0, 256, int(rng.integers(0, flip_bits + 1))):
            desc[b >> 3] ^= np.uint8(1 << (b & 7))
        p = dict(pos=(Pw + rng.normal(0, jitter, 3)).astype(F), normal=n.astype(F), minDist=mind, maxDist=maxd, desc=desc,
                 skip=bool(rng.random() < 0.05), nObs=int(rng.random() < 0.8))
        pts.append(p)
        if rng.random() < dup:                                  # a rival: same place, another few bits off
            q = dict(p); q["desc"] = desc.copy()
            for b in rng.integers(0, 256, int(rng.integers(0, 4))):
                q["desc"][b >> 3] ^= np.uint8(1 << (b & 7))
            q["pos"] = (p["pos"] + rng.normal(0, jitter, 3)).astype(F); q["nObs"] = int(rng.random() < 0.8); q["skip"] = False
            pts.append(q)
    order = rng.permutation(len(pts))
    return [pts[k] for k in order]


def pack_points(iv, per_frame):
    from iv_slam_amd._lib import LOCAL_POINT_DTYPE
    tot = sum(len(p) for p in per_frame)
    a = np.zeros(max(tot, 1), LOCAL_POINT_DTYPE)
    off = [0]
    k = 0
    for pts in per_frame:
        for p in pts:
            a[k]["pos"] = p["pos"]; a[k]["normal"] = p["normal"]; a[k]["min_distance"] = p["minDist"]; a[k]["max_distance"] = p["maxDist"]
            a[k]["desc"] = p["desc"]; a[k]["flags"] = (1 if p["skip"] else 0) | (2 if p["nObs"] > 0 else 0)
            k += 1
        off.append(k)
    return a, np.array(off, np.int32)


def run_local(iv, cam, recs, frames, per_frame, poses, occupied, th, nn_ratio, max_points=None, cos_limit=0.5, tracker=None, quality=None):
    """poses: list per RECORD (the test's world) -> uploaded per frame SLOT, the layout of ivf_tracker_search_local's d_poses.
    quality = (point_q flat like the packed points, key_q [n_frames, nf]): updated on the device and returned as two more arrays."""
    import torch
    from iv_slam_amd import dist as ivd
    nf = cam["nf"]
    dev = torch.device("cuda:0")
    block = torch.from_numpy(ivd.pack_records(recs, nf).reshape(-1)).to(dev)
    tr = tracker or iv.BatchTracker(nf, cam["scale"], float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"]), float(cam["bf"]),
                                   cam["bounds"], max_pairs=len(frames), b=float(cam["b"]))
    pts, off = pack_points(iv, per_frame)
    M = max_points or max(1, max(len(p) for p in per_frame))
    dpts = torch.from_numpy(pts.view(np.uint8).reshape(-1)).to(dev)
    doff = torch.from_numpy(off).to(dev)
    dfr = torch.tensor(frames, dtype=torch.int32, device=dev)
    dposes = None if poses is None else torch.from_numpy(np.stack([poses[ri][:3, :4].reshape(12) for ri in frames]).astype(F)).to(dev)
    docc = None
    if occupied is not None:
        oc = np.zeros((len(frames), nf), np.uint8)
        for i, o in enumerate(occupied):
            oc[i, :len(o)] = o
        docc = torch.from_numpy(oc).to(dev)
    assign = torch.full((len(frames), nf), -7, dtype=torch.int32, device=dev); nm = torch.full((len(frames),), -7, dtype=torch.int32, device=dev)
    dpq = dkq = None
    if quality is not None:
        dpq = torch.from_numpy(np.ascontiguousarray(quality[0], F)).to(dev); dkq = torch.from_numpy(np.ascontiguousarray(quality[1], F)).to(dev)
    tr.search_local(block, dfr, dpts, doff, M, assign, nm, poses=dposes, occupied=docc, th=th, nn_ratio=nn_ratio, cos_limit=cos_limit,
                    point_quality=dpq, key_quality=dkq)
    torch.cuda.synchronize()
    if quality is not None:
        return assign.cpu().numpy(), nm.cpu().numpy(), dpq.cpu().numpy(), dkq.cpu().numpy()
    return assign.cpu().numpy(), nm.cpu().numpy()


def check_local(cam, recs, frames, per_frame, poses, occupied, th, nn_ratio, got_a, got_nm, max_points=None, cos_limit=0.5, what=""):
    import projection_oracle as PO
    I = np.eye(4, dtype=F)
    logscale = F(O.lib.orc_logf(float(cam["scale"][1])))
    total = 0
    for k, ri in enumerate(frames):
        cur = frame_dict(recs[ri], I if poses is None else poses[ri], cam); cur["logScale"] = logscale
        pts = per_frame[k] if max_points is None else per_frame[k][:max_points]
        nm, exp = PO.search_local_points_frame(O, cur, pts, None if occupied is None else occupied[k], F(th), F(nn_ratio), cos_limit)
        nC = len(cur["kps"])
        assert got_nm[k] == nm, "%s frame %d: nmatches %d vs oracle %d" % (what, k, got_nm[k], nm)
        assert np.array_equal(got_a[k, :nC], exp), "%s frame %d: assignment differs at %r" % (what, k, np.nonzero(got_a[k, :nC] != exp)[0][:8])
        assert (got_a[k, nC:] == -1).all()
        total += nm
    return total


def test_local_points_kitti_shape(iv):
    """1242x375 / 1000 features: every frame searches the map made from the frame before it (and from itself: the stationary case,
    PredictScale on the knife edge), th = 1 / 3 / 5 (Tracking.cc:2125-2128), nn_ratio 0.8 and 0.64, with and without occupancy."""
    cam, recs, _, _ = extracted_sequence(iv, 1242, 375, 1000, 5, seed=191)
    rng = np.random.default_rng(7)
    I = np.eye(4, dtype=F)
    frames = [1, 2, 3, 4, 2, 0]
    src = [0, 1, 2, 3, 2, 0]                                               # frame 2 also searches its own points, frame 0 likewise
    per_frame = [map_points_from(cam, recs[s], I, rng) for s in src]
    occ = [rng.random(len(recs[f]["kps"])) < 0.1 for f in frames]
    tr = None
    for th, ratio, oc in ((1.0, 0.8, None), (3.0, 0.8, occ), (5.0, 0.64, occ), (1.0, 0.64, occ)):
        a, nm = run_local(iv, cam, recs, frames, per_frame, None, oc, th, ratio)
        tot = check_local(cam, recs, frames, per_frame, None, oc, th, ratio, a, nm, what="th %g ratio %g" % (th, ratio))
        assert tot > 500, tot


def test_local_points_with_poses_and_truncation(iv):
    """moving camera (forward / backward / rotation), points outside the frustum, behind the camera, out of the distance range and
    beyond the viewing-angle limit; max_points_per_frame smaller than a frame's list; an empty list; cos_limit 0.8."""
    cam, recs, _, _ = extracted_sequence(iv, 640, 240, 500, 4, seed=192, shift=2)
    cam["fx"] = cam["fy"] = F(370.0); cam["cx"] = F(320.0); cam["cy"] = F(120.0); cam["bf"] = F(198.75); cam["b"] = F(F(198.75) / F(370.0))
    for r in recs:
        d = r["kps"]["x"] - r["uright"]
        r["depth"] = np.where(r["uright"] >= 0, cam["bf"] / np.maximum(d, F(1e-3)), F(-1)).astype(F)
    rng = np.random.default_rng(8)
    poses = [pose(0.0, [0, 0, 0]), pose(1.5, [0.05, 0.0, -0.9]), pose(-2.0, [-0.1, 0.02, 1.2]), pose(12.0, [0.3, 0.0, 0.1], deg_x=2.0)]
    frames = [1, 2, 3, 0, 1]
    src = [0, 1, 2, 3, 1]
    per_frame = [map_points_from(cam, recs[s], poses[s], rng, jitter=0.05) for s in src]
    # some hopeless points: behind the camera, far off axis, wrong distance range, normal turned away
    for k in range(len(per_frame)):
        base = per_frame[k][:12]
        for j, p in enumerate(base):
            q = dict(p); q["desc"] = p["desc"].copy()
            if j % 4 == 0: q["pos"] = (p["pos"] * F(-1.0)).astype(F)
            elif j % 4 == 1: q["minDist"] = F(p["maxDist"] * F(3.0)); q["maxDist"] = F(p["maxDist"] * F(9.0))
            elif j % 4 == 2: q["normal"] = (-p["normal"]).astype(F)
            else: q["pos"] = (p["pos"] + np.array([400, 0, 0], F)).astype(F)
            per_frame[k].insert(int(rng.integers(0, len(per_frame[k]))), q)
    per_frame[3] = []                                                       # nToMatch == 0
    occ = [rng.random(len(recs[f]["kps"])) < 0.15 for f in frames]
    a, nm = run_local(iv, cam, recs, frames, per_frame, poses, occ, 3.0, 0.8)
    tot = check_local(cam, recs, frames, per_frame, poses, occ, 3.0, 0.8, a, nm, what="poses")
    assert tot > 100 and nm[3] == 0
    cap = min(len(p) for p in per_frame if p) // 2
    a, nm = run_local(iv, cam, recs, frames, per_frame, poses, None, 1.0, 0.8, max_points=cap, cos_limit=0.8)
    check_local(cam, recs, frames, per_frame, poses, None, 1.0, 0.8, a, nm, max_points=cap, cos_limit=0.8, what="truncated")


def test_local_points_quality_propagation_on_the_device(iv):
    """--ivslam_propagate_keyptqual: UpdateQualityScores(F) at the end of SearchByProjection(F, vpMapPoints, th)
    (ORB/src/ORBmatcher.cc:128-132, :1108-1121) for the keypoints that received a point in the call, on the device."""
    import projection_oracle as PO
    cam, recs, _, _ = extracted_sequence(iv, 640, 240, 500, 4, seed=192, shift=2)
    rng = np.random.default_rng(17)
    I = np.eye(4, dtype=F)
    frames = [1, 2, 3, 0]
    per_frame = [map_points_from(cam, recs[s], I, rng) for s in (0, 1, 2, 0)]
    off = np.cumsum([0] + [len(p) for p in per_frame])
    nf = cam["nf"]
    pq = (rng.integers(0, 250, int(off[-1])) * F(0.004)).astype(F); kq = (rng.integers(0, 250, (len(frames), nf)) * F(0.004)).astype(F)
    occ = [rng.random(len(recs[f]["kps"])) < 0.1 for f in frames]
    a, nm, gpq, gkq = run_local(iv, cam, recs, frames, per_frame, None, occ, 3.0, 0.8, quality=(pq, kq))
    check_local(cam, recs, frames, per_frame, None, occ, 3.0, 0.8, a, nm, what="quality")
    for k, ri in enumerate(frames):
        nC = len(recs[ri]["kps"])
        ekq, epq = PO.update_quality_scores([int(v) for v in a[k, :nC]], kq[k, :nC], pq[off[k]:off[k + 1]])
        assert gkq[k, :nC].tobytes() == ekq.tobytes() and gpq[off[k]:off[k + 1]].tobytes() == epq.tobytes(), k
        assert np.array_equal(gkq[k, nC:], kq[k, nC:])
    assert (gkq != kq).any() and (gpq != pq).any()


def test_local_frame_table_outside_the_block_is_refused_per_frame(iv):
    """a record index outside the block or a decreasing offset: that frame reports nmatches = -1 / assign = -1, the others are served."""
    import torch
    from iv_slam_amd import dist as ivd
    cam, recs, _, _ = extracted_sequence(iv, 640, 240, 500, 3, seed=193, shift=2)
    rng = np.random.default_rng(3)
    I = np.eye(4, dtype=F)
    per_frame = [map_points_from(cam, recs[0], I, rng), map_points_from(cam, recs[1], I, rng), map_points_from(cam, recs[1], I, rng)]
    nf = cam["nf"]; dev = torch.device("cuda:0")
    block = torch.from_numpy(ivd.pack_records(recs, nf).reshape(-1)).to(dev)
    pts, off = pack_points(iv, per_frame)
    tr = iv.BatchTracker(nf, cam["scale"], float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"]), float(cam["bf"]),
                         cam["bounds"], max_pairs=3, b=float(cam["b"]))
    dpts = torch.from_numpy(pts.view(np.uint8).reshape(-1)).to(dev)
    a = torch.full((3, nf), -7, dtype=torch.int32, device=dev); nm = torch.full((3,), -7, dtype=torch.int32, device=dev)
    tr.search_local(block, torch.tensor([1, 9, 2], dtype=torch.int32, device=dev), dpts, torch.from_numpy(off).to(dev), max(len(p) for p in per_frame), a, nm)
    torch.cuda.synchronize()
    a, nm = a.cpu().numpy(), nm.cpu().numpy()
    assert nm[1] == -1 and (a[1] == -1).all()
    check_local(cam, recs, [1, 2], [per_frame[0], per_frame[2]], None, None, 1.0, 0.8, a[[0, 2]], nm[[0, 2]], what="valid frames beside an invalid one")
    bad_off = off.copy(); bad_off[2] = bad_off[1] - 5                       # frame 1's range runs backwards, frame 2 starts below frame 1's start
    a2 = torch.full((3, nf), -7, dtype=torch.int32, device=dev); nm2 = torch.full((3,), -7, dtype=torch.int32, device=dev)
    tr.search_local(block, torch.tensor([1, 2, 2], dtype=torch.int32, device=dev), dpts, torch.from_numpy(bad_off).to(dev), max(len(p) for p in per_frame), a2, nm2)
    torch.cuda.synchronize()
    assert int(nm2[1]) == -1 and int(nm2[0]) == int(nm[0])


def test_local_points_overflowing_windows_and_ties(iv):
    """a crowd of near-identical keypoints (tiled texture -> equal descriptors, ties in distance) and th = 40: windows hold far more
    than 64 candidates, so the greedy kernel re-walks them against the live occupancy state; one handle, two calls (scratch grows)."""
    import torch
    w, h, n = 640, 240, 2000
    tile = np.random.default_rng(5).integers(0, 256, (24, 32)).astype(np.uint8)
    L = np.tile(tile, (h // 24, w // 32)); R = np.roll(L, -4, axis=1)
    dev = torch.device("cuda:0")
    bf, fx = 386.1448, 718.856
    fe = iv.StereoFrontend(w, h, 2, nfeatures=n, bf=bf, fx=fx)
    fe.run(torch.from_numpy(np.stack([L, np.roll(L, 2, axis=1)])).to(dev), torch.from_numpy(np.stack([R, np.roll(R, 2, axis=1)])).to(dev))
    fe.sync()
    recs = [fe.fetch(k, 0) for k in range(2)]
    from test_gpu_track import scale_table
    cam = dict(nf=n, scale=scale_table(), fx=F(fx), fy=F(fx), cx=F(w / 2 + 0.5), cy=F(h / 2 - 0.25), bf=F(bf), b=F(F(bf) / F(fx)),
               bounds=(0.0, 0.0, float(w), float(h)))
    assert min(len(r["kps"]) for r in recs) > 800
    rng = np.random.default_rng(9)
    I = np.eye(4, dtype=F)
    per_frame = [map_points_from(cam, recs[0], I, rng, flip_bits=2, dup=0.3), map_points_from(cam, recs[1], I, rng, flip_bits=0, dup=0.0)]
    frames = [1, 0]
    tr = iv.BatchTracker(n, cam["scale"], float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"]), float(cam["bf"]),
                         cam["bounds"], max_pairs=2, b=float(cam["b"]))
    a, nm = run_local(iv, cam, recs, frames, [p[:40] for p in per_frame], None, None, 1.0, 0.9, tracker=tr)           # small first call
    check_local(cam, recs, frames, [p[:40] for p in per_frame], None, None, 1.0, 0.9, a, nm, what="small")
    a, nm = run_local(iv, cam, recs, frames, per_frame, None, None, 40.0, 0.9, tracker=tr)
    tot = check_local(cam, recs, frames, per_frame, None, None, 40.0, 0.9, a, nm, what="overflow")
    assert tot > 50


def test_local_points_argument_errors(iv):
    import torch
    from iv_slam_amd import dist as ivd
    dev = torch.device("cuda:0")
    from test_gpu_track import scale_table
    tr = iv.BatchTracker(100, scale_table(), 500.0, 500.0, 320.0, 120.0, 200.0, (0.0, 0.0, 640.0, 240.0), max_pairs=2)
    rec = torch.zeros(2 * ivd.record_bytes(100), dtype=torch.uint8, device=dev)
    fr = torch.tensor([0, 1, 0], dtype=torch.int32, device=dev)
    pts = torch.zeros(80 * 4, dtype=torch.uint8, device=dev); off = torch.tensor([0, 2, 4, 4], dtype=torch.int32, device=dev)
    a = torch.empty((3, 100), dtype=torch.int32, device=dev); nm = torch.empty(3, dtype=torch.int32, device=dev)
    with pytest.raises(Exception):
        tr.search_local(rec, fr, pts, off, 4, a, nm)                        # three frames, max_pairs 2
    with pytest.raises(Exception):
        tr.search_local(rec, fr[:2], pts, off, 0, a, nm)                    # no capacity
    with pytest.raises(Exception):
        tr.search_local(rec, fr[:2], pts, off, 4, a, nm, th=0.0)
    tr.search_local(rec, fr[:2], pts, off, 4, a, nm)                        # empty records: nothing to match, no fault
    torch.cuda.synchronize()
    assert (nm[:2].cpu().numpy() == 0).all() and (a[:2].cpu().numpy() == -1).all()


@pytest.mark.parametrize("seed", range(4))
def test_local_points_random_synthetic(iv, seed):
    """keypoints that never came from the extractor: random positions (also outside the grid's rounding range), all eight octaves,
    descriptors with few distinct values (many exact ties in distance), frames with no keypoints, frames with no points, a point
    list far longer than the keypoint list; th 1 and 7 on one handle."""
    from test_gpu_track import scale_table
    rng = np.random.default_rng(500 + seed)
    w, h, nf = 752, 480, int(rng.choice([64, 700, 2100]))
    cam = dict(nf=nf, scale=scale_table(), fx=F(458.0), fy=F(457.0), cx=F(367.0), cy=F(248.0), bf=F(50.0), b=F(F(50.0) / F(458.0)),
               bounds=(0.0, 0.0, float(w), float(h)))
    base = rng.integers(0, 256, (6, 32)).astype(np.uint8)
    recs = []
    for k in range(4):
        n = 0 if k == 2 else int(rng.integers(nf // 2, nf + 1))
        kps = np.zeros(n, O.KP_DTYPE)
        kps["x"] = rng.uniform(-0.4, w + 0.4, n).astype(F); kps["y"] = rng.uniform(-0.4, h + 0.4, n).astype(F)
        kps["octave"] = rng.integers(0, 8, n); kps["angle"] = rng.uniform(0, 360, n).astype(F); kps["size"] = 31; kps["response"] = 20
        desc = base[rng.integers(0, 6, n)].copy()
        flip = rng.random(n) < 0.5
        desc[flip, rng.integers(0, 32, flip.sum())] ^= np.uint8(1) << rng.integers(0, 8, flip.sum()).astype(np.uint8)
        depth = np.where(rng.random(n) < 0.7, rng.uniform(2, 40, n), -1).astype(F)
        ur = np.where(depth > 0, kps["x"] - cam["bf"] / np.maximum(depth, F(1e-3)), F(-1)).astype(F)
        recs.append(dict(kps=kps, desc=desc, uright=ur, depth=depth))
    poses = [pose(0.0, [0, 0, 0]), pose(3.0, [0.2, 0.0, -0.5]), pose(0.0, [0, 0, 0]), pose(-4.0, [0.0, 0.1, 0.8], deg_x=1.0)]
    frames = [1, 0, 2, 3, 1]
    per_frame = [map_points_from(cam, recs[0], poses[0], rng, flip_bits=3, dup=0.4, jitter=0.1),
                 map_points_from(cam, recs[1], poses[1], rng, flip_bits=3, dup=0.4, jitter=0.1) * 3,      # list longer than the keypoints
                 map_points_from(cam, recs[0], poses[0], rng),                                            # frame without keypoints
                 [],                                                                                       # frame without points
                 map_points_from(cam, recs[3], poses[3], rng, flip_bits=1, dup=0.5, jitter=0.2)]
    occ = [rng.random(len(recs[f]["kps"])) < 0.2 for f in frames]
    tr = iv.BatchTracker(nf, cam["scale"], float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"]), float(cam["bf"]),
                         cam["bounds"], max_pairs=len(frames), b=float(cam["b"]))
    for th in (1.0, 7.0):
        a, nm = run_local(iv, cam, recs, frames, per_frame, poses, occ, th, 0.8, tracker=tr)
        check_local(cam, recs, frames, per_frame, poses, occ, th, 0.8, a, nm, what="synthetic th %g" % th)
        assert nm[2] == 0 and nm[3] == 0
