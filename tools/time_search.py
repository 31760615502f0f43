"""Latency of the window searches (ORBmatcher::SearchByProjection(cur,last) and SearchByProjection(F, mapPoints)) at
N = 1000 / 2000 features per 1242x375 frame: resident frame made straight from a front-end batch (windows + Hamming on the
device, greedy replay on the host) vs the host-array entry point (host grid + device Hamming) vs the CPU oracle on one core.
Writes gpurun_out/search_latency.json (copied to profiles/ by hand)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import iv_slam_amd as iv
from iv_slam_amd import synth
import oracle_lib as O

W, H = 1242, 375
dev = torch.device("cuda:0")
out = {"image": [W, H], "unit": "us per call (median of 30)", "cases": []}


def med(fn, reps=30):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
    return round(float(np.median(ts)), 1)


for n in (1000, 2000):
    base_l, base_r = synth.make_pair(W, H, seed=300 + n, idx=0)
    lefts = np.stack([np.roll(base_l, 3 * k, axis=1) for k in range(2)]); rights = np.stack([np.roll(base_r, 3 * k, axis=1) for k in range(2)])
    fe = iv.StereoFrontend(W, H, 2, nfeatures=n)
    fe.run(torch.from_numpy(lefts).to(dev), torch.from_numpy(rights).to(dev)); fe.sync()
    bounds = (0.0, 0.0, float(W), float(H))
    last, cur = fe.fetch(0, 0), fe.fetch(1, 0)
    sc = iv.ORBextractor(n, 1.2, 8, 20, 7).GetScaleFactors()
    sel = last["uright"] >= 0
    lk = last["kps"][sel]; disp = lk["x"] - last["uright"][sel]
    q = dict(u=(lk["x"] + 3).astype(np.float32), v=lk["y"].astype(np.float32), ur=(lk["x"] + 3 - disp).astype(np.float32),
             radius=(15 * sc[lk["octave"]]).astype(np.float32), min_level=(lk["octave"] - 1).astype(np.int32),
             max_level=(lk["octave"] + 1).astype(np.int32), angle=lk["angle"].copy(), desc=last["desc"][sel].copy(),
             valid=np.ones(len(lk), np.uint8), blocks=np.ones(len(lk), np.uint8), level=lk["octave"].astype(np.int32))
    m = iv.ORBmatcher(0.9, True)
    frame = iv.DeviceFrame.from_frontend(fe, 1, 0, bounds)
    ga, gn = frame.SearchByProjection(q)
    ha, hn = m.SearchByProjection(cur["kps"], cur["desc"], cur["uright"], bounds, q)
    oa, on = O.search_by_projection(cur["kps"], cur["desc"], cur["uright"], bounds, q, True)
    assert gn == on == hn and np.array_equal(ga, oa) and np.array_equal(ha, oa)
    case = {"nfeatures": n, "keypoints_cur": int(len(cur["kps"])), "queries": int(len(lk)), "matches": int(gn),
            "frame_from_frontend_create": med(lambda: iv.DeviceFrame.from_frontend(fe, 1, 0, bounds)),
            "frame_from_host_arrays_create": med(lambda: iv.DeviceFrame(cur["kps"], cur["desc"], cur["uright"], bounds)),
            "search_by_projection": {"resident_frame": med(lambda: frame.SearchByProjection(q)),
                                     "host_arrays": med(lambda: m.SearchByProjection(cur["kps"], cur["desc"], cur["uright"], bounds, q)),
                                     "oracle_one_core": med(lambda: O.search_by_projection(cur["kps"], cur["desc"], cur["uright"], bounds, q, True))},
            "search_map_points": {"resident_frame": med(lambda: frame.SearchByProjectionMapPoints(q, 0.8)),
                                  "host_arrays": med(lambda: iv.ORBmatcher(0.8).SearchByProjectionMapPoints(cur["kps"], cur["desc"], cur["uright"], bounds, q)),
                                  "oracle_one_core": med(lambda: O.search_map_points(cur["kps"], cur["desc"], cur["uright"], bounds, q, 0.8))}}
    out["cases"].append(case)
    print(json.dumps(case))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "search_latency.json"), "w"), indent=1)
