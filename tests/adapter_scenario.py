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
    out.append(O.hamming(pool[0]["desc"], pool[1 % len(pool)]["desc"]))
    return np.array(out, np.int64), dict(cur_last=nm1, local=nm2, reloc=nm3, kf_sim3=nm4, fused=nf)
