"""Batched, device-resident tracker step (ivf_tracker_run) against the oracle: the matcher part of Tracking::TrackWithMotionModel
(ORB/src/Tracking.cc:1303-1330) = UpdateLastFrame's stereo points (Tracking.cc:1256-1300, Frame::UnprojectStereo Frame.cc:958-972)
-> ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, false) (ORB/src/ORBmatcher.cc:1372-1518) -> retry with 2 * th.
Oracle = oracle/projection_oracle.track_with_motion_model_matches (numpy projection loops around the C oracle's window search
and greedy replay).  Bar: CurrentFrame.mvpMapPoints (as last-keypoint indices) and nmatches IDENTICAL for every frame pair."""
import math
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from iv_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
F = np.float32


@pytest.fixture(scope="module")
def iv():
    import iv_slam_amd
    assert iv_slam_amd.load().ivf_device_count() >= 1, "no HIP device: libivfront has no CPU fallback"
    return iv_slam_amd


def scale_table(nlevels=8, sf=1.2):
    s = [F(1.0)]
    for _ in range(1, nlevels):
        s.append(F(np.float64(s[-1]) * np.float64(F(sf))))                 # ORBextractor.cc:419-425: scaleFactor = (double)(float)1.2
    return np.array(s, F)


def pose(deg_y, t, deg_x=0.0):
    a = math.radians(deg_y); b = math.radians(deg_x)
    Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    T = np.eye(4); T[:3, :3] = Ry @ Rx; T[:3, 3] = t
    return T.astype(F)


def frame_dict(rec, T, cam):
    return dict(kps=rec["kps"], desc=rec["desc"], uright=rec["uright"], depth=rec["depth"], T=T, scale=cam["scale"], fx=cam["fx"],
                fy=cam["fy"], cx=cam["cx"], cy=cam["cy"], mbf=cam["bf"], mb=cam["b"], bounds=cam["bounds"])


def pair_poses(poses, pairs):
    """poses: None, a list per RECORD (pair (a, b) then carries (poses[a], poses[b])) or a list per PAIR of (T_last, T_cur)
    -> list per pair of (T_last, T_cur): the layout of ivf_tracker_run's d_poses [n_pairs][2][12]."""
    if poses is None:
        return None
    if isinstance(poses[0], tuple):
        assert len(poses) == len(pairs)
        return list(poses)
    return [(poses[a], poses[b]) for a, b in pairs]


def run_tracker(iv, cam, recs, pairs, poses=None, flags=None, quality=None, **kw):
    """recs: list of dict(kps, desc, uright, depth) -> (assign [n_pairs, nf], nmatches [n_pairs]) from the device.
    quality = (point_q [n_pairs, nf], key_q [n_pairs, nf]) float32: updated on the device, returned as two more arrays."""
    import torch
    from iv_slam_amd import dist as ivd
    nf = cam["nf"]
    dev = torch.device("cuda:0")
    block = torch.from_numpy(ivd.pack_records(recs, nf).reshape(-1)).to(dev)
    tr = iv.BatchTracker(nf, cam["scale"], float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"]), float(cam["bf"]),
                         cam["bounds"], max_pairs=len(pairs), b=float(cam["b"]), **kw)
    assert tr.record_bytes == ivd.record_bytes(nf)
    dp = torch.tensor(pairs, dtype=torch.int32, device=dev).reshape(-1, 2)
    assign = torch.full((len(pairs), nf), -7, dtype=torch.int32, device=dev); nm = torch.full((len(pairs),), -7, dtype=torch.int32, device=dev)
    pp = pair_poses(poses, pairs)
    dposes = None if pp is None else torch.from_numpy(np.stack([np.stack([tl[:3, :4].reshape(12), tc[:3, :4].reshape(12)]) for tl, tc in pp]).astype(F)).to(dev)
    dflags = None
    if flags is not None:
        fl = np.zeros((len(recs), nf), np.uint8)
        for i, f in enumerate(flags):
            fl[i, :len(f)] = f
        dflags = torch.from_numpy(fl).to(dev)
    dpq = dkq = None
    if quality is not None:
        dpq = torch.from_numpy(np.ascontiguousarray(quality[0], F)).to(dev); dkq = torch.from_numpy(np.ascontiguousarray(quality[1], F)).to(dev)
    tr.run(block, dp, assign, nm, poses=dposes, point_flags=dflags, point_quality=dpq, key_quality=dkq)
    torch.cuda.synchronize()
    if quality is not None:
        return assign.cpu().numpy(), nm.cpu().numpy(), dpq.cpu().numpy(), dkq.cpu().numpy()
    return assign.cpu().numpy(), nm.cpu().numpy()


def check_pairs(cam, recs, pairs, got_assign, got_nm, poses=None, flags=None, th=7.0, th_retry=None, retry_below=20,
                check_orientation=True, th_depth=0.0, points_block=True, what="", quality=None, got_quality=None):
    import projection_oracle as PO
    I = np.eye(4, dtype=F)
    total = 0
    pp = pair_poses(poses, pairs)
    for k, (a, b) in enumerate(pairs):
        last = frame_dict(recs[a], I if pp is None else pp[k][0], cam); cur = frame_dict(recs[b], I if pp is None else pp[k][1], cam)
        q = None
        if quality is not None:
            q = (np.array(quality[0][k], F), np.array(quality[1][k], F))
        nm, exp = PO.track_with_motion_model_matches(O, cur, last, F(th), F(2 * th if th_retry is None else th_retry), retry_below,
                                                     check_orientation, th_depth, points_block, None if flags is None else flags[a], quality=q)
        if q is not None:
            nL = len(last["kps"])
            assert got_quality[0][k].tobytes() == q[0].tobytes(), "%s pair %d: map-point quality differs at %r" % (what, k, np.nonzero(got_quality[0][k] != q[0])[0][:8])
            assert got_quality[1][k].tobytes() == q[1].tobytes(), "%s pair %d: mvKeyQualScore differs at %r" % (what, k, np.nonzero(got_quality[1][k] != q[1])[0][:8])
        nC = len(cur["kps"])
        assert got_nm[k] == nm, "%s pair %d (%d -> %d): nmatches %d vs oracle %d" % (what, k, a, b, got_nm[k], nm)
        assert np.array_equal(got_assign[k, :nC], exp), "%s pair %d: assignment differs at %r" % (what, k, np.nonzero(got_assign[k, :nC] != exp)[0][:8])
        assert (got_assign[k, nC:] == -1).all()
        total += nm
    return total


def extracted_sequence(iv, w, h, n, frames, seed, shift=3, cost=False):
    """consecutive "frames": the scene shifts `shift` px per frame (left AND right), extracted by the batched front end and
    packed into gather records on the device; returns the unpacked records (kps, desc, uright, depth)."""
    import torch
    from iv_slam_amd.frontend import unpack_gather_records
    L, R = synth.make_pair(w, h, seed=seed, idx=0)
    lefts = np.stack([np.roll(L, shift * k, axis=1) for k in range(frames)]); rights = np.stack([np.roll(R, shift * k, axis=1) for k in range(frames)])
    dev = torch.device("cuda:0")
    bf, fx = 386.1448, 718.856
    fe = iv.StereoFrontend(w, h, frames, nfeatures=n, bf=bf, fx=fx)
    fe.run(torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev))
    rec = fe.gather_record_bytes()
    block = torch.zeros(frames * rec, dtype=torch.uint8, device=dev)
    fe.pack_gather_block(block)
    fe.sync(); torch.cuda.synchronize()
    recs = unpack_gather_records(block.cpu().numpy(), n)
    for k in range(frames):                                                 # the records carry mvDepth as the front end computed it
        r = fe.fetch(k, 0)
        assert recs[k]["depth"].tobytes() == r["depth"].tobytes() and recs[k]["uright"].tobytes() == r["uright"].tobytes()
    cam = dict(nf=n, scale=scale_table(), fx=F(fx), fy=F(fx), cx=F(w / 2 + 0.5), cy=F(h / 2 - 0.25), bf=F(bf), b=F(F(bf) / F(fx)),
               bounds=(0.0, 0.0, float(w), float(h)))
    return cam, recs, block, fe


def test_zero_motion_kitti_shape_from_a_frontend_batch(iv):
    """configs[1]/[2] shape: 1242x375, 1000 features, 6 consecutive frames straight from a front-end batch; zero-motion prior."""
    cam, recs, block, fe = extracted_sequence(iv, 1242, 375, 1000, 6, seed=91)
    pairs = [(k - 1, k) for k in range(1, 6)] + [(0, 0), (5, 2)]
    for kw in (dict(th=7.0), dict(th=15.0, retry_below=0, check_orientation=False), dict(th=7.0, th_depth=40.0), dict(th=7.0, points_block=False)):
        a, nm = run_tracker(iv, cam, recs, pairs, **kw)
        tot = check_pairs(cam, recs, pairs, a, nm, what=str(kw), **kw)
        assert tot > 1000                                                   # the shifted scene really re-matches
    # the same through the device block the front end packed itself (no host round trip of the records)
    import torch
    dev = torch.device("cuda:0")
    tr = iv.BatchTracker(1000, cam["scale"], float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"]), float(cam["bf"]),
                         cam["bounds"], max_pairs=len(pairs), b=float(cam["b"]))
    dp = torch.tensor(pairs, dtype=torch.int32, device=dev)
    assign = torch.empty((len(pairs), 1000), dtype=torch.int32, device=dev); nmm = torch.empty(len(pairs), dtype=torch.int32, device=dev)
    tr.run(block, dp, assign, nmm)
    torch.cuda.synchronize()
    check_pairs(cam, recs, pairs, assign.cpu().numpy(), nmm.cpu().numpy(), what="device block")


def test_boundary_pair_between_two_batches_through_the_carry_record(iv):
    """r05: the pair (last frame of batch k, first frame of batch k + 1) is tracked too (Tracking.cc:1303-1330 runs for every frame).
    Three batches of 4 consecutive frames through ONE front end (three batch contexts, three internal streams): each batch is packed on
    its own stream into [4 records | carry record], BoundaryCarry hands the last record over to the next batch's buffer, the tracker
    step of batch k runs track_pairs(1, 0, 4, carry=True) on batch k's stream.  Every pair -- the boundary pairs first -- equals the
    oracle on the same records; the first batch's carry record is empty (0 matches)."""
    import torch
    from iv_slam_amd import dist as ivd
    from iv_slam_amd.frontend import unpack_gather_records
    w, h, n, P, NB = 640, 240, 500, 4, 3
    L, R = synth.make_pair(w, h, seed=95, idx=0)
    lefts = np.stack([np.roll(L, 3 * k, axis=1) for k in range(P * NB)]); rights = np.stack([np.roll(R, 3 * k, axis=1) for k in range(P * NB)])
    dev = torch.device("cuda:0")
    bf, fx = 386.1448, 718.856
    fe = iv.StereoFrontend(w, h, P, nfeatures=n, bf=bf, fx=fx)
    rec = fe.gather_record_bytes()
    cam = dict(nf=n, scale=scale_table(), fx=F(fx), fy=F(fx), cx=F(w / 2 + 0.5), cy=F(h / 2 - 0.25), bf=F(bf), b=F(F(bf) / F(fx)),
               bounds=(0.0, 0.0, float(w), float(h)))
    # NB + 1 buffers here, so that every batch's buffer -- carry record included -- is still intact when the host reads it back after
    # the last batch (bench.py rings three: batch k + 2's hand-over overwrites the carry slot batch k read, long after it was read)
    bufs = [torch.zeros((P + 1) * rec, dtype=torch.uint8, device=dev) for _ in range(NB + 1)]
    carry = ivd.BoundaryCarry(bufs, 1, P, rec)
    pairs = ivd.track_pairs(1, 0, P, carry=True)
    assert len(pairs) == P and pairs[0] == (P, 0)
    dp = torch.tensor(pairs, dtype=torch.int32, device=dev)
    trackers = [iv.BatchTracker(n, cam["scale"], float(fx), float(fx), float(cam["cx"]), float(cam["cy"]), float(bf), cam["bounds"], max_pairs=P,
                                b=float(cam["b"])) for _ in range(3)]
    assign = [torch.full((P, n), -7, dtype=torch.int32, device=dev) for _ in range(NB)]
    nm = [torch.full((P,), -7, dtype=torch.int32, device=dev) for _ in range(NB)]
    dl = torch.from_numpy(lefts).to(dev); dr = torch.from_numpy(rights).to(dev)
    for k in range(NB):                                            # enqueued back to back, nothing waits on the host
        fe.run(dl[k * P:(k + 1) * P], dr[k * P:(k + 1) * P])
        bs = torch.cuda.ExternalStream(fe.batch_stream(0), device=dev)
        fe.pack_gather_block(bufs[k], fe.STREAM_OF_BATCH)
        carry.publish(k, bs)
        carry.acquire(k, bs)
        trackers[k % 3].run(bufs[k], dp, assign[k], nm[k], stream_ptr=bs.cuda_stream)
        carry.release(k, bs)
    fe.sync(); torch.cuda.synchronize()
    recs_b = [unpack_gather_records(bufs[k].cpu().numpy(), n) for k in range(NB)]
    for k in range(NB):
        recs = recs_b[k]
        if k == 0:
            assert recs[P]["n"] == 0 and int(nm[0][0]) == 0 and (assign[0][0].cpu().numpy() == -1).all()     # nothing before the first frame
        else:
            assert recs[P]["n"] > 100 and recs[P]["kps"].tobytes() == recs_b[k - 1][P - 1]["kps"].tobytes()  # the carry IS the previous batch's last record
            assert recs[P]["desc"].tobytes() == recs_b[k - 1][P - 1]["desc"].tobytes() and recs[P]["depth"].tobytes() == recs_b[k - 1][P - 1]["depth"].tobytes()
        tot = check_pairs(cam, recs, pairs, assign[k].cpu().numpy(), nm[k].cpu().numpy(), what="batch %d" % k)
        assert tot > 100 * (P - 1 if k == 0 else P)
        if k > 0:
            assert int(nm[k][0]) > 100                             # the boundary pair really re-matches the shifted scene


def test_cpp_exchange_step_against_the_c_abi(iv, tmp_path):
    """r06 (the r05 verdict's item 7): the exchange step a C++ host writes -- tests/adapter/exchange_rccl.cpp = INTEGRATION.md section 6: pack on
    the batch's stream, ncclAllGather into [G * P records | carry], the carry hand-over, ivf_tracker_run -- built with g++ against
    include/ivfront.h + <rccl/rccl.h>, linked with libivfront / libamdhip64 / librccl and RUN at world = 1 (the collective is skipped,
    everything else is the multi-rank code): three batches of four frames through one front end; every pair incl. the two boundary pairs
    equals the oracle on the records the C++ side holds."""
    import ctypes as C
    import subprocess
    import torch
    from iv_slam_amd.frontend import unpack_gather_records
    so = str(tmp_path / "libivx.so")
    lib_dir = os.path.join(ROOT, "iv_slam_amd")
    rccl = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(rccl):
        pytest.skip("no RCCL headers on this box")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-shared", "-fPIC", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "adapter", "exchange_rccl.cpp"), "-o", so, "-L", lib_dir, "-livfront", "-L", "/opt/rocm/lib", "-lamdhip64", "-lrccl",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    X = C.CDLL(so)
    X.ivx_c_create.restype = C.c_void_p; X.ivx_c_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    X.ivx_c_step.restype = C.c_int; X.ivx_c_step.argtypes = [C.c_void_p] * 6
    X.ivx_c_records.restype = C.c_void_p; X.ivx_c_records.argtypes = [C.c_void_p, C.c_int]
    X.ivx_c_pairs.restype = C.c_int; X.ivx_c_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    X.ivx_c_destroy.argtypes = [C.c_void_p]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    w, h, n, P, NB = 640, 240, 500, 4, 3
    L, R = synth.make_pair(w, h, seed=96, idx=0)
    lefts = np.stack([np.roll(L, 3 * k, axis=1) for k in range(P * NB)]); rights = np.stack([np.roll(R, 3 * k, axis=1) for k in range(P * NB)])
    dev = torch.device("cuda:0")
    bf, fx = 386.1448, 718.856
    fe = iv.StereoFrontend(w, h, P, nfeatures=n, bf=bf, fx=fx)
    rec = fe.gather_record_bytes()
    cam = dict(nf=n, scale=scale_table(), fx=F(fx), fy=F(fx), cx=F(w / 2 + 0.5), cy=F(h / 2 - 0.25), bf=F(bf), b=F(F(bf) / F(fx)),
               bounds=(0.0, 0.0, float(w), float(h)))
    trackers = [iv.BatchTracker(n, cam["scale"], float(fx), float(fx), float(cam["cx"]), float(cam["cy"]), float(bf), cam["bounds"], max_pairs=P,
                                b=float(cam["b"])) for _ in range(3)]
    x = X.ivx_c_create(1, 0, P, n, None)
    assert x
    hp = np.zeros((P, 2), np.int32)
    assert X.ivx_c_pairs(x, hp.ctypes.data, hp.size) == P
    pairs = [tuple(int(v) for v in r) for r in hp]
    from iv_slam_amd import dist as ivd
    assert pairs == ivd.track_pairs(1, 0, P, carry=True)                  # the C++ pair table IS dist.track_pairs
    assign = [torch.full((P, n), -7, dtype=torch.int32, device=dev) for _ in range(NB)]
    nm = [torch.full((P,), -7, dtype=torch.int32, device=dev) for _ in range(NB)]
    dl = torch.from_numpy(lefts).to(dev); dr = torch.from_numpy(rights).to(dev)
    snaps = []
    for k in range(NB):
        fe.run(dl[k * P:(k + 1) * P], dr[k * P:(k + 1) * P])
        assert X.ivx_c_step(x, fe._h, trackers[k % 3]._h, None, assign[k].data_ptr(), nm[k].data_ptr()) == 0
        # snapshot of batch k's record buffer on its own stream, behind the tracker step (the ring of three is reused by batch k + 2's hand-over)
        bs = torch.cuda.ExternalStream(fe.batch_stream(0), device=dev)
        snap = torch.empty((P + 1) * rec, dtype=torch.uint8, device=dev)
        snaps.append((snap, bs))
        assert hip.hipMemcpyAsync(snap.data_ptr(), X.ivx_c_records(x, k), (P + 1) * rec, 3, bs.cuda_stream) == 0      # 3 = hipMemcpyDeviceToDevice
    fe.sync(); torch.cuda.synchronize()
    recs_b = [unpack_gather_records(s.cpu().numpy(), n) for s, _ in snaps]
    for k in range(NB):
        recs = recs_b[k]
        if k == 0:
            assert recs[P]["n"] == 0 and int(nm[0][0]) == 0 and (assign[0][0].cpu().numpy() == -1).all()
        else:
            assert recs[P]["n"] > 100 and recs[P]["kps"].tobytes() == recs_b[k - 1][P - 1]["kps"].tobytes()      # the carry IS the previous batch's last record
        tot = check_pairs(cam, recs, pairs, assign[k].cpu().numpy(), nm[k].cpu().numpy(), what="C++ exchange, batch %d" % k)
        assert tot > 100 * (P - 1 if k == 0 else P)
        if k > 0:
            assert int(nm[k][0]) > 100
    X.ivx_c_destroy(x)


def test_poses_forward_backward_and_flags(iv):
    """supplied poses: forward motion (levels [o, inf)), backward motion ([0, o]), small motion (+-1), rotations; explicit point flags."""
    cam, recs, _, _ = extracted_sequence(iv, 640, 240, 500, 5, seed=92, shift=2)
    cam["fx"] = cam["fy"] = F(370.0); cam["cx"] = F(320.0); cam["cy"] = F(120.0); cam["bf"] = F(198.75); cam["b"] = F(F(198.75) / F(370.0))
    # depths must follow the camera: recompute mvDepth = mbf / disparity for the test camera (inputs, any positive values do)
    for r in recs:
        d = r["kps"]["x"] - r["uright"]
        r["depth"] = np.where(r["uright"] >= 0, cam["bf"] / np.maximum(d, F(0.01)), F(-1)).astype(F)
    poses = [pose(0.0, [0, 0, 0]), pose(0.15, [0.03, -0.01, -0.9]), pose(0.1, [0.0, 0.0, -0.1]), pose(-0.2, [0.01, 0.02, 0.8], 0.1), pose(0.0, [0, 0, 0.8])]
    pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 2), (4, 0)]
    rng = np.random.default_rng(5)
    flags = [(rng.integers(0, 4, len(r["kps"]))).astype(np.uint8) for r in recs]
    for kw, fl in ((dict(th=7.0), None), (dict(th=15.0, retry_below=1000, th_retry=25.0), None), (dict(th=10.0, th_depth=12.0), None),
                   (dict(th=7.0), flags)):
        a, nm = run_tracker(iv, cam, recs, pairs, poses=poses, flags=fl, **kw)
        check_pairs(cam, recs, pairs, a, nm, poses=poses, flags=fl, what=str(kw), **kw)


def test_per_pair_poses_prior_and_optimised(iv):
    """INTEGRATION.md's example: d_poses[p] = {Tcw_last (optimised), Tcw_cur (motion-model prior)}.  Frame k is "cur" in pair
    (k-1, k) under its PRIOR and "last" in pair (k, k+1) under its OPTIMISED pose -- two different poses for one record in one
    call, which a per-record pose table could not express (ORB/src/Tracking.cc:1303-1311)."""
    cam, recs, _, _ = extracted_sequence(iv, 640, 240, 500, 5, seed=93, shift=2)
    opt = [pose(0.03 * k, [0.02 * k, 0.0, -0.25 * k]) for k in range(5)]
    pri = [pose(0.03 * k + 0.02, [0.02 * k + 0.01, 0.005, -0.25 * k - 0.07]) for k in range(5)]
    pairs = [(k - 1, k) for k in range(1, 5)]
    pp = [(opt[a], pri[b]) for a, b in pairs]
    a, nm = run_tracker(iv, cam, recs, pairs, poses=pp, th=7.0)
    tot = check_pairs(cam, recs, pairs, a, nm, poses=pp, th=7.0, what="per-pair poses")
    assert tot > 200
    # and it matters: with the priors replaced by the optimised poses at least one pair matches differently
    a2, nm2 = run_tracker(iv, cam, recs, pairs, poses=[(opt[x], opt[y]) for x, y in pairs], th=7.0)
    assert not np.array_equal(a, a2)


def test_quality_propagation_on_the_device(iv):
    """--ivslam_propagate_keyptqual: UpdateQualityScores(CurrentFrame) at the end of every SearchByProjection call
    (ORB/src/ORBmatcher.cc:1108-1121, :1513-1515) inside the batched tracker, incl. the second update of a retried pair."""
    cam, recs, _, _ = extracted_sequence(iv, 640, 240, 500, 5, seed=94, shift=2)
    pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 3), (4, 4)]
    rng = np.random.default_rng(11)
    nf = cam["nf"]
    # scores on a 0.004 lattice around each other: differences on both sides of kDeltaThresh = 0.01, exact ties included
    pq = (rng.integers(0, 250, (len(pairs), nf)) * F(0.004)).astype(F); kq = (rng.integers(0, 250, (len(pairs), nf)) * F(0.004)).astype(F)
    close = rng.uniform(size=pq.shape) < 0.5
    kq[close] = (pq[close] + rng.integers(-4, 5, int(close.sum())) * F(0.004)).astype(F)
    for kw in (dict(th=7.0), dict(th=3.0, retry_below=1000, th_retry=9.0), dict(th=7.0, points_block=False)):
        a, nm, gpq, gkq = run_tracker(iv, cam, recs, pairs, quality=(pq, kq), **kw)
        check_pairs(cam, recs, pairs, a, nm, what="quality %r" % kw, quality=(pq, kq), got_quality=(gpq, gkq), **kw)
        assert (gkq != kq).any() and (gpq != pq).any()
    # without matches nothing moves; padding beyond the keypoint count is never touched
    a, nm, gpq, gkq = run_tracker(iv, cam, recs, [(4, 4)], quality=(pq[:1], kq[:1]), th=7.0, check_orientation=False)
    n4 = len(recs[4]["kps"])
    assert np.array_equal(gkq[0, n4:], kq[0, n4:])


def test_pair_table_outside_the_block_is_refused_per_pair(iv):
    """a pair table built for another block (more records) must not read outside this one: nmatches = -1, assign = -1 for
    those pairs, the valid ones unaffected."""
    cam, recs, _, _ = extracted_sequence(iv, 640, 240, 500, 3, seed=95, shift=2)
    pairs = [(0, 1), (1, 7), (-1, 2), (1, 2), (3, 0)]
    a, nm = run_tracker(iv, cam, recs, pairs, th=7.0)
    good = [0, 3]
    check_pairs(cam, recs, [pairs[k] for k in good], a[good], nm[good], th=7.0, what="valid pairs beside invalid ones")
    for k in (1, 2, 4):
        assert nm[k] == -1 and (a[k] == -1).all()


def _random_records(rng, nf, n_frames, w, h, cluster):
    """frames of random keypoints: frame k+1 = frame k's points jittered (descriptors with a few flipped bits, many exact
    duplicates -> distance ties), plus clutter; `cluster` squeezes them into a small region (windows of hundreds of candidates)."""
    from iv_slam_amd._lib import KP_DTYPE
    recs = []
    n0 = int(rng.integers(nf // 2, nf + 1))
    x = rng.uniform(20, (w - 20) * cluster, n0); y = rng.uniform(20, (h - 20) * cluster, n0)
    desc = rng.integers(0, 256, (n0, 32)).astype(np.uint8)
    desc[rng.integers(0, n0, n0 // 4)] = desc[rng.integers(0, n0, n0 // 4)]          # duplicates
    octv = rng.integers(0, 8, n0); ang = rng.uniform(0, 360, n0)
    for k in range(n_frames):
        n = len(x)
        kp = np.zeros(n, KP_DTYPE)
        kp["x"] = x.astype(F); kp["y"] = y.astype(F); kp["octave"] = octv; kp["angle"] = ang.astype(F); kp["size"] = 31; kp["response"] = 50
        disp = rng.uniform(2, 60, n)
        stereo = rng.uniform(size=n) < 0.8
        ur = np.where(stereo, kp["x"] - disp, -1).astype(F)
        depth = np.where(stereo, F(198.75) / disp.astype(F), F(-1)).astype(F)
        recs.append(dict(kps=kp, desc=desc.copy(), uright=ur, depth=depth))
        # next frame: permute, jitter, flip bits, drop some, add clutter
        perm = rng.permutation(n)[:int(n * 0.9)]
        x = x[perm] + rng.uniform(-4, 4, len(perm)); y = y[perm] + rng.uniform(-3, 3, len(perm))
        octv = np.clip(octv[perm] + rng.integers(-1, 2, len(perm)), 0, 7); ang = (ang[perm] + rng.choice([0, 0, 0, 2, 45, 200], len(perm))) % 360
        desc = desc[perm].copy()
        flip = rng.integers(0, 256, (len(perm), 3))
        for j in range(3):
            sel = rng.uniform(size=len(perm)) < 0.5
            desc[sel, flip[sel, j] // 8] ^= (1 << (flip[sel, j] % 8)).astype(np.uint8)
        extra = min(nf - len(perm), int(rng.integers(0, nf // 8 + 1)))
        x = np.concatenate([x, rng.uniform(0, w * cluster, extra)]); y = np.concatenate([y, rng.uniform(0, h * cluster, extra)])
        octv = np.concatenate([octv, rng.integers(0, 8, extra)]); ang = np.concatenate([ang, rng.uniform(0, 360, extra)])
        desc = np.concatenate([desc, rng.integers(0, 256, (extra, 32)).astype(np.uint8)])
        x = np.clip(x, 0, w - 1); y = np.clip(y, 0, h - 1)
    return recs


@pytest.mark.parametrize("seed,nf,cluster,th", [(1, 300, 1.0, 7.0), (2, 600, 0.25, 12.0), (3, 1500, 0.12, 30.0), (4, 64, 1.0, 7.0), (5, 4096, 0.5, 9.0)])
def test_random_records_ties_and_overflowing_windows(iv, seed, nf, cluster, th):
    """seeded random frames: duplicate descriptors (first-minimum tie-break), re-assignment by points without observations,
    windows far beyond the 64-entry candidate list (re-walked against the live state), empty frames, nfeatures up to the cap."""
    rng = np.random.default_rng(seed)
    w, h = 640, 240
    cam = dict(nf=nf, scale=scale_table(), fx=F(370.0), fy=F(370.0), cx=F(320.0), cy=F(120.0), bf=F(198.75), b=F(F(198.75) / F(370.0)),
               bounds=(0.0, 0.0, float(w), float(h)))
    recs = _random_records(rng, nf, 5, w, h, cluster)
    from iv_slam_amd._lib import KP_DTYPE
    recs.append(dict(kps=np.zeros(0, KP_DTYPE), desc=np.zeros((0, 32), np.uint8), uright=np.zeros(0, F), depth=np.zeros(0, F)))   # empty frame
    pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 4), (4, 5), (5, 0), (2, 2)]
    if cluster < 0.3:
        # these cases must really overflow the 64-entry candidate lists (the in-place re-walk of k_track_greedy)
        k0, k1 = recs[0]["kps"], recs[1]["kps"]
        big = max(len(O.features_in_area(k1, cam["bounds"], float(k0["x"][i]), float(k0["y"][i]), float(F(th) * cam["scale"][k0["octave"][i]]),
                                         int(k0["octave"][i]) - 1, int(k0["octave"][i]) + 1)) for i in range(0, len(k0), 7))
        assert big > 64, big
    poses = [pose(0.02 * k, [0.01 * k, 0.0, -0.05 * k]) for k in range(6)]
    for kw, ps in ((dict(th=th), None), (dict(th=th, points_block=False, retry_below=0), poses), (dict(th=th, th_depth=6.0), None)):
        a, nm = run_tracker(iv, cam, recs, pairs, poses=ps, **kw)
        check_pairs(cam, recs, pairs, a, nm, poses=ps, what="seed %d %r" % (seed, kw), **kw)


@pytest.mark.parametrize("seed", range(1000 + int(os.environ.get("IVF_FUZZ_SEED0", "0")), 1000 + int(os.environ.get("IVF_FUZZ_SEED0", "0")) + int(os.environ.get("IVF_FUZZ_TRACK", "3"))))
def test_random_records_more_seeds(iv, seed):
    """the random-record scenario of the test above with geometry drawn from the seed (IVF_FUZZ_TRACK = number of seeds for soak runs)"""
    rng = np.random.default_rng(seed)
    nf = int(rng.choice([100, 300, 700, 1200, 2500])); cluster = float(rng.choice([1.0, 0.5, 0.25, 0.15])); th = float(rng.choice([5.0, 7.0, 12.0, 20.0]))
    w, h = 640, 240
    cam = dict(nf=nf, scale=scale_table(), fx=F(370.0), fy=F(370.0), cx=F(320.0), cy=F(120.0), bf=F(198.75), b=F(F(198.75) / F(370.0)),
               bounds=(0.0, 0.0, float(w), float(h)))
    recs = _random_records(rng, nf, 4, w, h, cluster)
    pairs = [(0, 1), (1, 2), (2, 3), (0, 3), (3, 1)]
    flags = [rng.choice(np.array([0, 1, 3, 3], np.uint8), size=len(r["kps"])) for r in recs]
    for kw, fl in ((dict(th=th), None), (dict(th=th, retry_below=0), flags)):
        a, nm = run_tracker(iv, cam, recs, pairs, flags=fl, **kw)
        check_pairs(cam, recs, pairs, a, nm, flags=fl, what="seed %d %r" % (seed, kw), **kw)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_consecutive_queries_claim_the_same_keypoints(iv, seed):
    """k_track_greedy commits a group of 16 queries at once unless two of them chose the same keypoint: here runs of CONSECUTIVE last
    keypoints sit on the same spot with near-identical descriptors (every member of a run wants the same few current keypoints), runs
    straddle the 16-query groups, some points have no observations (their assignment does not block: the next query may take the same
    keypoint), some windows hold more than 16 and more than 64 candidates.  Bar: identical to the serial oracle."""
    from iv_slam_amd._lib import KP_DTYPE
    rng = np.random.default_rng(seed)
    w, h, nf = 640, 240, 512
    cam = dict(nf=nf, scale=scale_table(), fx=F(370.0), fy=F(370.0), cx=F(320.0), cy=F(120.0), bf=F(198.75), b=F(F(198.75) / F(370.0)),
               bounds=(0.0, 0.0, float(w), float(h)))

    def make(n_spots, run_len, cand_per_spot):
        sx = rng.uniform(40, w - 40, n_spots); sy = rng.uniform(30, h - 30, n_spots)
        sd = rng.integers(0, 256, (n_spots, 32)).astype(np.uint8)
        so = rng.integers(0, 4, n_spots)
        # last frame: run_len consecutive keypoints per spot (same position up to 0.5 px, descriptor up to 2 flipped bits)
        lx, ly, ld, lo = [], [], [], []
        for s_ in range(n_spots):
            for _ in range(int(run_len[s_])):
                lx.append(sx[s_] + rng.uniform(-0.5, 0.5)); ly.append(sy[s_] + rng.uniform(-0.5, 0.5)); lo.append(so[s_])
                d = sd[s_].copy()
                for _b in range(int(rng.integers(0, 3))):
                    bit = int(rng.integers(0, 256)); d[bit // 8] ^= np.uint8(1 << (bit % 8))
                ld.append(d)
        n = min(len(lx), nf)
        kl = np.zeros(n, KP_DTYPE); kl["x"] = np.array(lx[:n], F); kl["y"] = np.array(ly[:n], F); kl["octave"] = np.array(lo[:n]); kl["angle"] = 10; kl["size"] = 31
        disp = rng.uniform(4, 40, n)
        last = dict(kps=kl, desc=np.array(ld[:n], np.uint8), uright=(kl["x"] - disp).astype(F), depth=(F(198.75) / disp.astype(F)).astype(F))
        # current frame: cand_per_spot keypoints around every spot, descriptors at growing distance from the spot's
        cx_, cy_, cd, co = [], [], [], []
        for s_ in range(n_spots):
            for j in range(int(cand_per_spot[s_])):
                cx_.append(sx[s_] + rng.uniform(-3, 3)); cy_.append(sy[s_] + rng.uniform(-3, 3)); co.append(so[s_])
                d = sd[s_].copy()
                for bit in rng.choice(256, size=min(3 * (j % 9), 60), replace=False):
                    d[bit // 8] ^= np.uint8(1 << (bit % 8))
                cd.append(d)
        order = rng.permutation(len(cx_))[:nf]
        kc = np.zeros(len(order), KP_DTYPE); kc["x"] = np.array(cx_, F)[order]; kc["y"] = np.array(cy_, F)[order]; kc["octave"] = np.array(co)[order]
        kc["angle"] = 12; kc["size"] = 31
        cur = dict(kps=kc, desc=np.array(cd, np.uint8)[order], uright=np.full(len(order), -1, F), depth=np.full(len(order), -1, F))
        return last, cur

    a0, b0 = make(40, rng.integers(1, 12, 40), rng.integers(1, 9, 40))               # short runs, a handful of candidates: the row-packed path
    a1, b1 = make(12, rng.integers(10, 40, 12), rng.integers(12, 30, 12))            # long runs over several groups, lists of 12..30 (> a row)
    a2, b2 = make(4, rng.integers(30, 100, 4), rng.integers(70, 120, 4))             # lists beyond the 64-entry cap: re-walked windows
    recs = [a0, b0, a1, b1, a2, b2]
    pairs = [(0, 1), (2, 3), (4, 5), (0, 3), (2, 1)]
    flags = [rng.choice(np.array([0, 1, 3, 3], np.uint8), size=len(r["kps"])) for r in recs]   # bit 0: a map point; bit 1: with observations (it blocks)
    for kw, fl in ((dict(th=7.0), None), (dict(th=7.0, retry_below=0), flags), (dict(th=7.0, points_block=False, check_orientation=False), None)):
        a, nm = run_tracker(iv, cam, recs, pairs, flags=fl, **kw)
        total = check_pairs(cam, recs, pairs, a, nm, flags=fl, what="seed %d %r" % (seed, kw), **kw)
        assert total > 50, total


def test_tracker_argument_checks(iv):
    import torch
    sc = scale_table()
    with pytest.raises(iv.IvfError):
        iv.BatchTracker(5000, sc, 700.0, 700.0, 600.0, 180.0, 386.0, (0.0, 0.0, 1242.0, 375.0), max_pairs=4)        # beyond the LDS state
    with pytest.raises(iv.IvfError):
        iv.BatchTracker(1000, sc, 700.0, 700.0, 600.0, 180.0, 386.0, (0.0, 0.0, 0.0, 375.0), max_pairs=4)           # empty bounds
    tr = iv.BatchTracker(100, sc, 700.0, 700.0, 600.0, 180.0, 386.0, (0.0, 0.0, 1242.0, 375.0), max_pairs=2)
    dev = torch.device("cuda:0")
    blk = torch.zeros(3 * tr.record_bytes, dtype=torch.uint8, device=dev)
    pairs = torch.zeros((3, 2), dtype=torch.int32, device=dev)
    out = torch.zeros((3, 100), dtype=torch.int32, device=dev); nm = torch.zeros(3, dtype=torch.int32, device=dev)
    with pytest.raises(iv.IvfError):
        tr.run(blk, pairs, out, nm)                                         # more pairs than max_pairs
    tr.run(blk, pairs[:2], out, nm)                                         # empty records: no matches, nothing written out of range
    torch.cuda.synchronize()
    assert (nm.cpu().numpy()[:2] == 0).all() and (out.cpu().numpy()[:2] == -1).all()


def test_one_handle_on_several_streams_serialises_its_runs(iv):
    """the handle's scratch belongs to one run at a time: runs enqueued back to back on DIFFERENT streams (what a pipelined
    caller does) must not overlap in it -- same results as one run at a time."""
    import torch
    from iv_slam_amd import dist as ivd
    rng = np.random.default_rng(77)
    nf, w, h = 1200, 640, 240
    cam = dict(nf=nf, scale=scale_table(), fx=F(370.0), fy=F(370.0), cx=F(320.0), cy=F(120.0), bf=F(198.75), b=F(F(198.75) / F(370.0)),
               bounds=(0.0, 0.0, float(w), float(h)))
    dev = torch.device("cuda:0")
    sets = [_random_records(np.random.default_rng(100 + s), nf, 9, w, h, 0.6) for s in range(3)]
    blocks = [torch.from_numpy(ivd.pack_records(r, nf).reshape(-1)).to(dev) for r in sets]
    pairs = torch.tensor([(k, k + 1) for k in range(8)], dtype=torch.int32, device=dev)
    tr = iv.BatchTracker(nf, cam["scale"], 370.0, 370.0, 320.0, 120.0, 198.75, cam["bounds"], max_pairs=8, b=float(cam["b"]), th=12.0)
    ref = []
    for b in blocks:
        a = torch.empty((8, nf), dtype=torch.int32, device=dev); n = torch.empty(8, dtype=torch.int32, device=dev)
        tr.run(b, pairs, a, n); torch.cuda.synchronize()
        ref.append((a.cpu().numpy(), n.cpu().numpy()))
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    for rep in range(5):
        outs = [(torch.empty((8, nf), dtype=torch.int32, device=dev), torch.empty(8, dtype=torch.int32, device=dev)) for _ in range(3)]
        for i in range(3):
            tr.run(blocks[i], pairs, outs[i][0], outs[i][1], stream_ptr=streams[i].cuda_stream)
        torch.cuda.synchronize()
        for i in range(3):
            assert np.array_equal(outs[i][0].cpu().numpy(), ref[i][0]) and np.array_equal(outs[i][1].cpu().numpy(), ref[i][1]), (rep, i)
    assert sum(int(r[1].sum()) for r in ref) > 1000
