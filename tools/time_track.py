#!/usr/bin/env python3
"""Time the batched tracker step (ivf_tracker_run) on consecutive frames of the benchmark's synthetic stream: P frames from one
front-end batch, frame pairs (k-1, k), zero-motion prior, th = 7 / retry 14 below 20 matches (Tracking.cc:1313-1330).
Prints one JSON line: us per frame pair (HIP events over `reps` launch sequences) next to the oracle on ONE host core."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--oracle-pairs", type=int, default=8)
    ap.add_argument("--width", type=int, default=0, help="image size (default: the benchmark's 1242 x 375; configs[4]: 1920 x 1200 with --nfeatures 4000 --fast 12 7)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--fast", type=int, nargs=2, default=[20, 7])
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import iv_slam_amd as iv
    from iv_slam_amd.frontend import unpack_gather_records
    W, H, N, P = a.width or bench.W, a.height or bench.H, a.nfeatures, a.pairs
    dev = torch.device("cuda:0")
    from iv_slam_amd import synth
    L0, R0 = synth.make_pair(W, H, seed=100, idx=0)                                   # one scene moving 3 px per frame: consecutive frames
    bl = torch.from_numpy(L0).to(dev); br = torch.from_numpy(R0).to(dev)
    left = torch.stack([torch.roll(bl, 3 * k, dims=1) for k in range(P)]); right = torch.stack([torch.roll(br, 3 * k, dims=1) for k in range(P)])
    fe = iv.StereoFrontend(W, H, P, nfeatures=N, iniThFAST=a.fast[0], minThFAST=a.fast[1], bf=bench.BF, fx=bench.FX)
    fe.run(left, right)
    rec = fe.gather_record_bytes()
    block = torch.zeros(P * rec, dtype=torch.uint8, device=dev)
    fe.pack_gather_block(block); fe.sync(); torch.cuda.synchronize()
    sc = iv.ORBextractor(N, 1.2, 8, a.fast[0], a.fast[1]).GetScaleFactors()
    cam = dict(fx=bench.FX, fy=bench.FX, cx=W / 2 + 0.5, cy=H / 2 - 0.25, bf=bench.BF)
    bounds = (0.0, 0.0, float(W), float(H))
    tr = iv.BatchTracker(N, sc, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"], bounds, max_pairs=P - 1)
    pairs = torch.tensor([(k - 1, k) for k in range(1, P)], dtype=torch.int32, device=dev)
    assign = torch.empty((P - 1, N), dtype=torch.int32, device=dev); nm = torch.empty(P - 1, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev)
    for _ in range(3):
        tr.run(block, pairs, assign, nm, stream_ptr=st.cuda_stream)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(a.reps):
        tr.run(block, pairs, assign, nm, stream_ptr=st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.reps / (P - 1)
    nmh = nm.cpu().numpy()
    # oracle on one core: the projection + window search + greedy replay of the same pairs
    import oracle_lib as O
    import projection_oracle as PO
    F = np.float32
    recs = unpack_gather_records(block.cpu().numpy(), N)
    I = np.eye(4, dtype=F)

    def fd(r):
        return dict(kps=r["kps"], desc=r["desc"], uright=r["uright"], depth=r["depth"], T=I, scale=sc, fx=F(cam["fx"]), fy=F(cam["fy"]),
                    cx=F(cam["cx"]), cy=F(cam["cy"]), mbf=F(cam["bf"]), mb=F(F(cam["bf"]) / F(cam["fx"])), bounds=bounds)
    ah = assign.cpu().numpy()
    t0 = time.perf_counter(); ok = True
    for k in range(1, min(P, a.oracle_pairs + 1)):
        onm, oa = PO.track_with_motion_model_matches(O, fd(recs[k]), fd(recs[k - 1]), F(7.0), F(14.0), 20)
        ok &= onm == nmh[k - 1] and np.array_equal(oa, ah[k - 1, :len(oa)])
    t_or = (time.perf_counter() - t0) / max(1, min(P - 1, a.oracle_pairs))
    # the C oracle's search alone (what r02 quoted: 90 us at N = 1000), on the flat queries of one pair
    last, cur = recs[0], recs[1]
    sel = last["uright"] >= 0; lk = last["kps"][sel]
    q = dict(u=lk["x"], v=lk["y"], ur=last["uright"][sel], radius=(F(7.0) * sc[lk["octave"]]).astype(F), min_level=(lk["octave"] - 1).astype(np.int32),
             max_level=(lk["octave"] + 1).astype(np.int32), angle=lk["angle"].copy(), desc=last["desc"][sel].copy(), valid=np.ones(len(lk), np.uint8),
             blocks=np.ones(len(lk), np.uint8))
    t0 = time.perf_counter()
    for _ in range(50):
        O.search_by_projection(cur["kps"], cur["desc"], cur["uright"], bounds, q, True)
    t_c = (time.perf_counter() - t0) / 50
    out = {"us_per_frame_pair": round(us, 3), "frame_pairs_per_launch_sequence": P - 1, "N": N, "reps": a.reps,
           "mean_matches": float(nmh.mean()), "min_matches": int(nmh.min()), "parity_vs_oracle_on_first_pairs": bool(ok),
           "oracle_us_per_frame_pair_one_core": {"python_projection_plus_c_search": round(t_or * 1e6, 1), "c_search_only": round(t_c * 1e6, 1)}}

    # ---- Tracking::SearchLocalPoints for every frame at once (ivf_tracker_search_local): the local map of frame k = the stereo points
    # of frames k-1 and k-2 (~2 x 600 points), identity poses
    from iv_slam_amd._lib import LOCAL_POINT_DTYPE
    rng = np.random.default_rng(3)
    per = []
    for k in range(P):
        pts = []
        for s in (max(k - 1, 0), max(k - 2, 0)):
            r = recs[s]; fr = fd(r)
            for i in np.nonzero(r["depth"] > 0)[0]:
                Pw = PO.unproject_stereo(fr, int(i)) if k < a.oracle_pairs + 1 else None
                pts.append((s, int(i), Pw))
        per.append(pts)
    M = max(len(p) for p in per)
    tot = sum(len(p) for p in per)
    arr = np.zeros(tot, LOCAL_POINT_DTYPE); off = np.zeros(P + 1, np.int32)
    kk = 0
    for k, pts in enumerate(per):
        for (s, i, Pw) in pts:
            r = recs[s]; z = r["depth"][i]
            x = F(F(F(r["kps"]["x"][i] - F(cam["cx"])) * z) * F(F(1.0) / F(cam["fx"]))); y = F(F(F(r["kps"]["y"][i] - F(cam["cy"])) * z) * F(F(1.0) / F(cam["fy"])))
            pos = np.array([x, y, z], F)                                  # identity pose: world = camera coordinates (UnprojectStereo)
            dist = F(np.sqrt(np.float64(x) ** 2 + np.float64(y) ** 2 + np.float64(z) ** 2))
            lv = int(r["kps"]["octave"][i])
            arr[kk]["pos"] = pos; arr[kk]["normal"] = (pos / dist).astype(F)
            arr[kk]["max_distance"] = F(dist * sc[lv]); arr[kk]["min_distance"] = F(arr[kk]["max_distance"] / sc[-1])
            arr[kk]["desc"] = r["desc"][i]; arr[kk]["flags"] = 2
            kk += 1
        off[k + 1] = kk
    dpts = torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(dev); doff = torch.from_numpy(off).to(dev)
    dfr = torch.arange(P, dtype=torch.int32, device=dev)
    trl = iv.BatchTracker(N, sc, cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["bf"], bounds, max_pairs=P)
    la = torch.empty((P, N), dtype=torch.int32, device=dev); lnm = torch.empty(P, dtype=torch.int32, device=dev)
    for _ in range(3):
        trl.search_local(block, dfr, dpts, doff, M, la, lnm, th=1.0, nn_ratio=0.8, stream_ptr=st.cuda_stream)
    e0.record(st)
    for _ in range(a.reps):
        trl.search_local(block, dfr, dpts, doff, M, la, lnm, th=1.0, nn_ratio=0.8, stream_ptr=st.cuda_stream)
    e1.record(st); torch.cuda.synchronize()
    us_l = e0.elapsed_time(e1) * 1e3 / a.reps / P
    lah, lnh = la.cpu().numpy(), lnm.cpu().numpy()
    logscale = F(O.lib.orc_logf(float(sc[1])))
    okl = True; t0 = time.perf_counter(); t_c2 = 0.0
    for k in range(min(P, a.oracle_pairs)):
        cur = fd(recs[k]); cur["logScale"] = logscale
        pts = [dict(pos=arr[j]["pos"], normal=arr[j]["normal"], minDist=arr[j]["min_distance"], maxDist=arr[j]["max_distance"], desc=arr[j]["desc"],
                    skip=False, nObs=1) for j in range(off[k], off[k + 1])]
        onm, oa = PO.search_local_points_frame(O, cur, pts, None, F(1.0), F(0.8))
        okl &= onm == lnh[k] and np.array_equal(oa, lah[k, :len(oa)])
    t_l = (time.perf_counter() - t0) / max(1, min(P, a.oracle_pairs))
    out["search_local_points"] = {"us_per_frame": round(us_l, 3), "frames_per_launch_sequence": P, "map_points_per_frame_mean": round(tot / P, 1),
                                  "map_points_per_frame_max": int(M), "mean_matches": float(lnh.mean()), "parity_vs_oracle_on_first_frames": bool(okl),
                                  "oracle_python_projection_plus_c_search_us_per_frame": round(t_l * 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
