"""Scenario builder shared by the adapter tests: a synthetic tracking situation (last frame with map points, current frame,
a keyframe, a local map) from real extracted keypoints, written in the flat binary format tests/adapter/adapter_driver.cpp
reads, plus the same data as Python dicts for oracle/projection_oracle.py."""
import math
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def build_driver(out_path):
    """g++ build of the driver against the mock cv / mock Frame types; links libivfront.so."""
    lib_dir = os.path.join(ROOT, "iv_slam_amd")
    cmd = ["g++", "-std=c++14", "-O1", "-ffp-contract=off", "-Wall", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "tests", "cv_mock"), os.path.join(ROOT, "tests", "adapter", "adapter_driver.cpp"), "-o", str(out_path),
           "-L", lib_dir, "-livfront", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return str(out_path)


def _rot_y(deg):
    a = math.radians(deg); c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def _pose(deg, t):
    T = np.eye(4, dtype=np.float64); T[:3, :3] = _rot_y(deg); T[:3, 3] = t
    return T.astype(F)


def make(O, synth, seed, forward):
    """returns (scenario dict for the oracle, bytes for the driver)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import projection_oracle as PO
    rng = np.random.default_rng(seed)
    w, h, n = 640, 240, 500
    fx = fy = F(370.0); cx, cy = F(320.0), F(120.0); bf = F(198.75); mb = F(bf / fx)
    L, R = synth.make_pair(w, h, seed=seed, idx=0)
    eL = O.Extractor(n, 1.2, 8, 20, 7); eR = O.Extractor(n, 1.2, 8, 20, 7)
    kL, dL = eL(L); kR, dR = eR(R)
    ur, depth = O.stereo_match(eL, eR, kL, dL, kR, dR, float(bf), float(mb))
    L2 = np.roll(L, 2, axis=1); R2 = np.roll(R, 2, axis=1)
    eL2 = O.Extractor(n, 1.2, 8, 20, 7); eR2 = O.Extractor(n, 1.2, 8, 20, 7)
    kC, dC = eL2(L2); kCr, dCr = eR2(R2)
    urC, _ = O.stereo_match(eL2, eR2, kC, dC, kCr, dCr, float(bf), float(mb))
    tab = eL.tables()
    scale = tab["scale"].astype(F); sig2 = tab["sigma2"].astype(F); inv2 = tab["inv_sigma2"].astype(F)
    bounds = (0.0, 0.0, float(w), float(h))

    def frame(kps, desc, uright, T):
        return dict(kps=kps, desc=desc, uright=uright.astype(F), scale=scale, sigma2=sig2, invSigma2=inv2, fx=fx, fy=fy, cx=cx, cy=cy,
                    mbf=bf, mb=mb, logScale=F(math.log(1.2)), bounds=bounds, T=T)

    last = frame(kL, dL, ur, _pose(0.0, [0, 0, 0]))
    cur = frame(kC, dC, urC, _pose(0.15, [0.03, -0.01, -0.9 if forward else -0.1]))
    kf = frame(kL, dL, ur, _pose(-0.1, [-0.02, 0.0, 0.05]))
    kf["Ow"] = PO.neg_rt_mul(kf["T"][:3, :3], kf["T"][:3, 3])
    # map points = last-frame stereo points back-projected (T_lw = identity: world = last camera frame)
    pool = []; last_mps = np.full(len(kL), -1, np.int32)
    for i in range(len(kL)):
        if depth[i] <= 0 or rng.uniform() < 0.1:
            continue
        z = F(depth[i]); pos = np.array([(kL["x"][i] - cx) * z / fx, (kL["y"][i] - cy) * z / fy, z], F)
        d = F(np.linalg.norm(pos.astype(np.float64)))
        maxd = F(d * scale[kL["octave"][i]])
        nrm = (pos / d).astype(F)
        last_mps[i] = len(pool)
        pool.append(dict(pos=pos, normal=nrm, minDist=F(maxd / scale[-1]), maxDist=maxd, desc=dL[i].copy(),
                         nObs=int(rng.integers(0, 4)), bad=bool(rng.uniform() < 0.03), inView=False, trackLevel=0, viewCos=F(1.0),
                         projX=F(0), projY=F(0), projXR=F(0)))
    last["mps"] = last_mps; last["outlier"] = (rng.uniform(size=len(kL)) < 0.05)
    # tracking fields of the "local map" (Frame::isInFrustum, Frame.cc:579-613, evaluated here in double precision: they are INPUTS)
    local = []
    Rc, tc = cur["T"][:3, :3].astype(np.float64), cur["T"][:3, 3].astype(np.float64)
    for m, p in enumerate(pool):
        pc = Rc @ p["pos"].astype(np.float64) + tc
        if pc[2] <= 0:
            continue
        u = fx * pc[0] / pc[2] + cx; v = fy * pc[1] / pc[2] + cy
        if not (0 <= u < w and 0 <= v < h) or rng.uniform() < 0.3:
            continue
        dist = np.linalg.norm(pc)
        p.update(inView=True, projX=F(u), projY=F(v), projXR=F(u - bf / pc[2]), viewCos=F(0.9995 if rng.uniform() < 0.5 else 0.9),
                 trackLevel=PO.predict_scale(p, F(dist), cur))
        local.append(m)
    cur_mps = np.full(len(kC), -1, np.int32)
    for i in rng.choice(len(kC), size=25, replace=False):
        cur_mps[i] = int(rng.integers(0, len(pool)))
    kf_mps = np.full(len(kL), -1, np.int32)
    for i in range(len(kL)):
        if rng.uniform() < 0.4:
            kf_mps[i] = int(rng.integers(0, len(pool)))
    kf["mps"] = kf_mps
    Scw = kf["T"].copy(); Scw[:3, :] = (Scw[:3, :] * F(1.1)).astype(F)
    S = dict(pool=pool, last=last, cur=cur, kf=kf, cur_mps=cur_mps, local=local, th=15.0, mono=0, Scw=Scw)

    # ---- binary for the driver
    b = bytearray()

    def arr(a, dt):
        a = np.ascontiguousarray(a, dt).reshape(-1)
        b.extend(struct.pack("<i", a.size)); b.extend(a.tobytes())

    b.extend(struct.pack("<i", len(pool)))
    for p in pool:
        arr(np.concatenate([p["pos"], p["normal"], [p["minDist"], p["maxDist"], p["viewCos"], p["projX"], p["projY"], p["projXR"]]]), F)
        arr([p["nObs"], int(p["bad"]), int(p["inView"]), p["trackLevel"]], np.int32)
        arr(p["desc"], np.uint8)
    for fr in (last, cur, kf):
        k = fr["kps"]
        arr(np.stack([k["x"], k["y"], k["size"], k["angle"], k["response"], k["octave"].astype(F)], axis=1), F)
        arr(fr["desc"], np.uint8); arr(fr["uright"], F); arr(fr["scale"], F); arr(fr["sigma2"], F); arr(fr["invSigma2"], F)
        b.extend(struct.pack("<11f", fr["fx"], fr["fy"], fr["cx"], fr["cy"], fr["mbf"], fr["mb"], fr["logScale"], *fr["bounds"]))
    for fr in (last, cur, kf):
        arr(fr["T"], F)
    arr(kf["Ow"], F)
    arr(last["mps"], np.int32); arr(last["outlier"].astype(np.int32), np.int32); arr(cur_mps, np.int32); arr(kf_mps, np.int32)
    b.extend(struct.pack("<fi", S["th"], S["mono"]))
    arr(local, np.int32); arr(Scw, F)
    # ---- second part: a second keyframe (the current frame's data), fresh map-point assignments, F12, the Sim3
    cur["Ow"] = PO.neg_rt_mul(cur["T"][:3, :3], cur["T"][:3, 3])
    # the current image is the last one shifted by 2 px: a current keypoint that sits on a shifted last keypoint observes the
    # same map point (consistent observations make the two directions of SearchBySim3 agree); the rest are random
    kf2_mps = np.full(len(kC), -1, np.int32)
    for j in range(len(kC)):
        near = np.nonzero((np.abs(kL["x"] + 2 - kC["x"][j]) <= 1.0) & (np.abs(kL["y"] - kC["y"][j]) <= 1.0) & (kL["octave"] == kC["octave"][j]))[0]
        if len(near) and last_mps[near[0]] >= 0 and rng.uniform() < 0.8:
            kf2_mps[j] = last_mps[near[0]]
        elif rng.uniform() < 0.1:
            kf2_mps[j] = int(rng.integers(0, len(pool)))
    kf_mps2 = np.full(len(kL), -1, np.int32)
    for i in range(len(kL)):
        if rng.uniform() < 0.35:
            kf_mps2[i] = last_mps[i] if (last_mps[i] >= 0 and rng.uniform() < 0.7) else int(rng.integers(0, len(pool)))
    # relative pose kf2 <- kf (p2 = R21 p1 + t21) and the fundamental matrix F12 with x1^T F12 x2 = 0 (ORB-SLAM's convention)
    T1 = kf["T"].astype(np.float64); T2 = cur["T"].astype(np.float64)
    T12 = T1 @ np.linalg.inv(T2)                                  # p1 = R12 p2 + t12
    R12, t12 = T12[:3, :3], T12[:3, 3]
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F12 = (np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)).astype(F)
    s12 = F(1.05)
    pre12 = np.full(len(kL), -1, np.int32)
    for i in rng.choice(len(kL), size=10, replace=False):
        pre12[i] = int(rng.integers(0, len(pool)))
    arr(cur["Ow"], F); arr(kf2_mps, np.int32); arr(kf_mps2, np.int32); arr(F12, F)
    arr(np.concatenate([[s12], R12.astype(F).reshape(-1), t12.astype(F)]), F); arr(pre12, np.int32)
    S.update(kf2_mps=kf2_mps, kf_mps2=kf_mps2, F12=F12, s12=s12, R12=R12.astype(F), t12=t12.astype(F), pre12=pre12)
    return S, bytes(b)


def expected(O, S):
    """the driver's output sequence, computed by the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import projection_oracle as PO
    pool, last, cur, kf = S["pool"], S["last"], S["cur"], S["kf"]
    out = []
    nm1, mps1 = PO.search_cur_last(O, cur, last, pool, list(S["cur_mps"]), S["th"], bool(S["mono"]))
    out += [nm1, len(mps1)] + list(mps1)
    nm2, mps2 = PO.search_local_points(O, cur, pool, mps1, S["local"], 3.0, 0.8)
    out += [nm2, len(mps2)] + list(mps2)
    found = set(int(m) for m in S["cur_mps"] if m >= 0)
    nm3, mps3 = PO.search_reloc(O, cur, kf, pool, list(S["cur_mps"]), found, 10.0, 100)
    out += [nm3, len(mps3)] + list(mps3)
    nm4, m4 = PO.search_kf_sim3(O, kf, S["Scw"], pool, S["local"], 10)
    out += [nm4, len(m4)] + list(m4)
    nf, kfm, rep = PO.fuse(O, kf, pool, S["local"], 3.0)
    out += [nf, len(kfm)] + list(kfm) + [len(rep)] + list(rep)
    # ---- steps 6-12 of the driver: fresh map points (original bad / observation state), second keyframe = current frame's data
    kf1 = dict(kf); kf1["mps"] = S["kf_mps2"]
    kf2 = dict(cur); kf2["mps"] = S["kf2_mps"]
    has1 = np.array([m >= 0 and not pool[m]["bad"] for m in kf1["mps"]], np.uint8)
    has2 = np.array([m >= 0 and not pool[m]["bad"] for m in kf2["mps"]], np.uint8)
    fv_kf, fv_cur, fv_last = PO.feature_vector(kf["desc"]), PO.feature_vector(cur["desc"]), PO.feature_vector(last["desc"])
    fm, n6 = O.search_by_bow(kf["kps"], kf["desc"], has1, fv_kf, cur["kps"], cur["desc"], fv_cur, 0.7, True)
    out += [n6, len(fm)] + [int(kf1["mps"][j]) if j >= 0 else -1 for j in fm]
    m12, n7 = O.search_by_bow_keyframes(kf["kps"], kf["desc"], has1, fv_kf, cur["kps"], cur["desc"], has2, fv_cur, 0.75, True)
    out += [n7, len(m12)] + [int(kf2["mps"][j]) if j >= 0 else -1 for j in m12]
    prev = np.stack([last["kps"]["x"], last["kps"]["y"]], axis=1).astype(F)
    im12, prev2, n8 = O.search_for_initialization(last["kps"], last["desc"], cur["kps"], cur["desc"], cur["bounds"], prev, 100, 0.9, True)
    out += [n8, len(im12)] + [int(v) for v in im12] + [int(v) for v in prev2.astype(F).reshape(-1).view(np.int32)]
    ex, ey = PO.epipole(kf, kf2)
    st1 = (kf["uright"] >= 0).astype(np.uint8); st2 = (cur["uright"] >= 0).astype(np.uint8)
    hm1 = (np.asarray(kf1["mps"]) >= 0).astype(np.uint8); hm2 = (np.asarray(kf2["mps"]) >= 0).astype(np.uint8)
    tm, n9 = O.search_for_triangulation(kf["kps"], kf["desc"], hm1, st1, fv_kf, cur["kps"], cur["desc"], hm2, st2, fv_cur, S["F12"], float(ex), float(ey),
                                        cur["scale"], cur["sigma2"], False, False)
    prs = [(i, int(j)) for i, j in enumerate(tm) if j >= 0]
    out += [n9, len(prs)] + [v for pr in prs for v in pr]
    n10, vm12 = PO.search_by_sim3(O, kf1, kf2, pool, [int(v) for v in S["pre12"]], S["s12"], S["R12"], S["t12"], 7.5)
    out += [n10, len(vm12)] + [int(v) for v in vm12]
    n11, repl, kf2_after = PO.fuse_sim3(O, kf2, S["Scw"], pool, S["local"], 4.0)
    out += [n11, len(repl)] + [int(v) for v in repl] + [len(kf2_after)] + [int(v) for v in kf2_after]
    kq0 = (F(0.5) + F(0.001) * (np.arange(len(cur["kps"])) % 400).astype(F)).astype(F)
    mq0 = (F(0.4) + F(0.002) * (np.arange(len(pool)) % 300).astype(F)).astype(F)
    kq, mq = PO.update_quality_scores([int(v) for v in S["cur_mps"]], kq0, mq0)
    out += [int(v) for v in kq.view(np.int32)] + [int(v) for v in mq.view(np.int32)]
    out.append(O.hamming(pool[0]["desc"], pool[1 % len(pool)]["desc"]))
    counts = dict(cur_last=nm1, local=nm2, reloc=nm3, kf_sim3=nm4, fused=nf, bow=n6, bow_kf=n7, init=n8, triangulation=n9, sim3=n10, fuse_sim3=n11)
    return np.array(out, np.int64), counts
