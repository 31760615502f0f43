#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the hot path (introspection FCN + ORB extract L+R + L/R stereo match) on MI355X.

Contract: python bench.py --gpus N --steps K --warmup W.  With N > 1 and no WORLD_SIZE in the environment bench.py starts
the N ranks itself (a fresh `python -m torch.distributed.run` child, before this process touches the GPU); under a launcher
it checks WORLD_SIZE == N.  One rank per GPU, RCCL ("nccl") between them.

One step = one pass of the hot path over one batch of synthetic 1242x375 stereo pairs that are already resident in HBM:
`--batches-per-step` launch sequences of `--pairs` pairs each (default 16 x 128 = 2048 pairs per step and GPU, so that the
driver's 20 timed steps last seconds, not a third of a second).  value = pairs all ranks processed / time (max over ranks).
Workload = BASELINE.json configs[2], the configuration the metric "extract+match+introspect" names: the introspection FCN
runs on every left image inside the timed step and its cost map gates the left extractor.  --no-introspect measures
configs[1] (extract + match only); the default run also reports that figure in `extract_match_only`, measured after the
timed region.  Prints ONE JSON line on rank 0, carrying `roofline` for the dominant kernel (HIP events on its stream inside
the timed region), `cpu_baseline` (ports of the reference CPU path timed on this box's host cores), and after the timed
region: `latency_ms_batch1` (the per-call drop-in API on host buffers, the mode the reference runs at 10 fps) and
`h2d_included` (the same stream fed from pinned host memory over PCIe, copies overlapped with compute).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "stereo pairs/s (extract+match+introspect) @1242×375, 1000 feat; 1/2/4/8 GPU"
W, H, NFEAT = 1242, 375, 1000
INI_TH, MIN_TH = 20, 7            # ORBextractor.iniThFAST / minThFAST (KITTI00-02.yaml:42-52)
BF, FX = 386.1448, 718.856
LEVEL_PX_SUM = 1441432            # sum of pixels over the 8 levels
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec

# BASELINE.json configs[k] as bench workloads (--config k).  configs[2] is the configuration the metric is quoted on: the
# default run.  configs[0] is the reference's CPU-runnable case (a parity-test case, not a bench line).
#   3: 2000 features per frame on the batched 8-level pyramid (the 8-rank half: --gpus N)
#   4: the un-binned Jackal stream: 1920x1200, 4000 features, FAST 12 / 7, introspection ON, cost map at the image size
#      (ORB/Examples/Stereo/jackal_visual_odom_stereo_inference.yaml:91-105 holds the ORB keys; its camera is the 2x2-binned
#      960x600 one, fx 528.955512 / bf 69.690815 (:8-27) -- doubled here for the full-resolution image)
CONFIGS = {
    1: dict(w=1242, h=375, n=1000, ini=20, mn=7, bf=386.1448, fx=718.856, introspect=False, pairs=128, stream=512,
            name="configs[1]: 1242x375 stereo pair stream, ORB extract + L/R Hamming match, introspection OFF"),
    2: dict(w=1242, h=375, n=1000, ini=20, mn=7, bf=386.1448, fx=718.856, introspect=True, pairs=128, stream=512,
            name="configs[2]: 1242x375 stereo stream with introspection FCN cost-map forward (MFMA convs) gating keypoints, "
                 "ORB extract + L/R Hamming match"),
    3: dict(w=1242, h=375, n=2000, ini=20, mn=7, bf=386.1448, fx=718.856, introspect=True, pairs=128, stream=512,
            name="configs[3]: batched 8-level pyramid, 2000 features/frame, 1242x375 stereo stream, introspection FCN + ORB extract + "
                 "L/R Hamming match (frames shard across ranks with the RCCL descriptor all-gather under --gpus N)"),
    4: dict(w=1920, h=1200, n=4000, ini=12, mn=7, bf=139.38163, fx=1057.911024, introspect=True, pairs=128, stream=256,      # r05: 128 pairs per launch sequence
            # (one box: 32 pairs 6,343 pairs/s and 35 us per frame pair in the tracker step, 64: 6,711-6,735 / 18.7, 128: 6,855 / 10.3 -- the tracker's greedy walk is one wave per pair)
            name="configs[4]: 1920x1200 Jackal stereo stream, 4000 features/frame, FAST 12/7, introspection ON (cost map 1920x1200), "
                 "ORB extract + L/R Hamming match + tracker step (the Tracking loop's matcher calls)"),
}
WORKLOAD = CONFIGS[2]["name"]


def apply_config(k):
    """Select BASELINE.json configs[k]: sets the module-level workload constants every leg of the bench reads."""
    global W, H, NFEAT, INI_TH, MIN_TH, BF, FX, LEVEL_PX_SUM, WORKLOAD
    c = CONFIGS[k]
    W, H, NFEAT, INI_TH, MIN_TH, BF, FX, WORKLOAD = c["w"], c["h"], c["n"], c["ini"], c["mn"], c["bf"], c["fx"], c["name"]
    import numpy as np
    F = np.float32
    sc = [F(1.0)]
    for _ in range(7):
        sc.append(F(np.float64(sc[-1]) * np.float64(F(1.2))))                  # ORBextractor.cc:419-425
    LEVEL_PX_SUM = sum(int(np.rint(F(W) * F(F(1.0) / v))) * int(np.rint(F(H) * F(F(1.0) / v))) for v in sc)   # :1303 (cvRound = rint)
    return c


RUN = 32                          # consecutive frames per visit of a scene


def make_device_stream(torch, dev, n_pairs, seed, base_pairs=16, rank=0, world=1):
    """>=256 distinct pairs without minutes of host synthesis, as a SEQUENCE: local frame i of this rank is global frame
    g = i * world + rank (frame k -> GPU k mod G); the global sequence visits `base_pairs` seeded host scenes for
    RUN consecutive frames each, the scene moving 3 px per frame (left AND right: disparity preserved) with +-1 seeded noise --
    so that consecutive frames re-match under the tracker's window search, wherever they were extracted."""
    from iv_slam_amd import synth
    base = synth.make_stream(base_pairs, W, H, seed=seed)
    bl = torch.from_numpy(base[:, 0].copy()).to(dev); br = torch.from_numpy(base[:, 1].copy()).to(dev)
    g_ = torch.Generator(device=dev); g_.manual_seed(1234 + seed + 7919 * rank)
    left = torch.empty((n_pairs, H, W), dtype=torch.uint8, device=dev); right = torch.empty_like(left)
    for i in range(n_pairs):
        g = i * world + rank
        b = (g // RUN) % base_pairs
        t = g % RUN + RUN * (g // (RUN * base_pairs))
        dx, dy = (3 * t) % 240, (t // 4) % 24
        for src, dst in ((bl, left), (br, right)):
            img = torch.roll(src[b], shifts=(dy, dx), dims=(0, 1)).to(torch.int16)
            img += torch.randint(-1, 2, img.shape, generator=g_, device=dev, dtype=torch.int16) * (t > 0)
            dst[i] = img.clamp_(0, 255).to(torch.uint8)
    return left, right


def track_pairs(world, rank, P, carry=False):
    from iv_slam_amd import dist as ivd
    return ivd.track_pairs(world, rank, P, carry=carry)


def cpu_baseline(pairs_sample, cores, threads_per_pair=1):
    """Oracle (bit-exact scalar restatement of the reference CPU path; kind = 'port') on host threads.
    threads_per_pair = 1: one pair per worker thread, `cores` pairs in flight (ctypes releases the GIL).
    threads_per_pair = 2: the reference's own threading model -- left and right extraction on two threads, joined, then
    the stereo matcher (ORB/src/Frame.cc:116-124) -- with cores // 2 pairs in flight."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import concurrent.futures as cf
    import oracle_lib as O          # checker / baseline only -- never on the product path
    from iv_slam_amd import synth
    imgs = [synth.make_pair(W, H, seed=900, idx=i) for i in range(min(pairs_sample, 8))]
    b = BF / FX
    inner = cf.ThreadPoolExecutor(max(cores, 2)) if threads_per_pair == 2 else None

    def work(i):
        L, R = imgs[i % len(imgs)]
        eL = O.Extractor(NFEAT, 1.2, 8, INI_TH, MIN_TH); eR = O.Extractor(NFEAT, 1.2, 8, INI_TH, MIN_TH)
        if inner is not None:
            fl = inner.submit(eL, L); fr = inner.submit(eR, R)
            (kL, dL), (kR, dR) = fl.result(), fr.result()
        else:
            kL, dL = eL(L); kR, dR = eR(R)
        O.stereo_match(eL, eR, kL, dL, kR, dR, BF, b)
        return 1

    work(0)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max(1, cores // threads_per_pair)) as ex:
        done = sum(ex.map(work, range(pairs_sample)))
    dt = time.perf_counter() - t0
    if inner is not None:
        inner.shutdown()
    return done / dt, dt


def cpu_fcn_baseline(threads, images=8, reps=8):
    """The introspection FCN on host cores the way the reference runs it: a FROZEN TorchScript module (torch.jit.trace +
    torch.jit.freeze of the layer list, oracle/fcn_oracle_torch.frozen -- what torch::jit::load + forward execute at
    ORB/Examples/Stereo/stereo_kitti.cc:236, :508), PyTorch's CPU kernels with `threads` intra-op threads, batches of `images`.
    Returns (images/s, seconds)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch
    import fcn_oracle_torch         # checker / baseline only -- never on the product path
    from iv_slam_amd import fcn_weights, synth
    torch.set_num_threads(threads)
    T = fcn_oracle_torch.prepare(fcn_weights.make_seeded_weights(7))
    img = np.stack([np.stack([synth.make_left(W, H, seed=901, idx=c + 3 * k) for c in range(3)], axis=-1) for k in range(2)])
    batch = np.concatenate([img] * (images // 2))
    run = fcn_oracle_torch.frozen(T, batch, (H, W))
    run(batch)
    t0 = time.perf_counter()
    for _ in range(reps):
        run(batch)
    dt = time.perf_counter() - t0
    return reps * len(batch) / dt, dt


def oracle_checksum():
    """sha1 over what the loaded oracle produces for one seeded pair: the -march=native build must equal the portable one."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hashlib
    import oracle_lib as O
    from iv_slam_amd import synth
    L, R = synth.make_pair(W, H, seed=900, idx=0)
    eL = O.Extractor(NFEAT, 1.2, 8, INI_TH, MIN_TH); eR = O.Extractor(NFEAT, 1.2, 8, INI_TH, MIN_TH)
    kL, dL = eL(L); kR, dR = eR(R)
    ur, dp = O.stereo_match(eL, eR, kL, dL, kR, dR, BF, BF / FX)
    return hashlib.sha1(kL.tobytes() + dL.tobytes() + kR.tobytes() + dR.tobytes() + ur.tobytes() + dp.tobytes()).hexdigest()


def cpu_baseline_worker(introspect):
    """Runs in a FRESH process (bench.py --cpu-baseline-worker): the oracle library named by IVF_ORACLE_SO -- the reference's
    own build flags, -O3 -march=native (ORB/CMakeLists.txt:16-17), compiled on this host -- and the frozen-TorchScript FCN leg."""
    cores = max(1, min(os.cpu_count() or 1, 32))
    # ~4 s on 32 cores at 1242x375: with the FCN leg (~9 s) a bounded sample of 10-30 s of CPU work; scaled by the pyramid size
    sample = max(4 * cores, int(max(256, 64 * cores) * 1441432 / LEVEL_PX_SUM))
    v, secs = cpu_baseline(sample, cores)
    v2, secs2 = cpu_baseline(max(2 * cores, int(max(64, 8 * cores) * 1441432 / LEVEL_PX_SUM)), cores, threads_per_pair=2)
    out = {"cores": cores, "sample_pairs": sample, "extract_match_all_cores": v, "secs": secs, "extract_match_two_threads_per_pair": v2,
           "checksum": oracle_checksum(), "oracle_so": os.environ.get("IVF_ORACLE_SO", "oracle/libivf_oracle.so")}
    if introspect:
        fv, fsecs = cpu_fcn_baseline(cores)
        out.update(fcn_images_per_s=fv, fcn_secs=fsecs)
    print("CPU_BASELINE " + json.dumps(out), flush=True)


def run_cpu_baseline(introspect, config=2):
    """cpu_baseline object of the JSON line: builds the oracle -march=native for THIS host (fallback: the portable -march=x86-64-v2
    library that travels with the repo), times it in a fresh process, checks the native build against the portable one."""
    import subprocess
    native = os.path.join(ROOT, "oracle", "_native")
    so = os.path.join(native, "libivf_oracle.so")
    flags = "-O3 -march=native -ffp-contract=off"
    try:
        os.makedirs(native, exist_ok=True)
        import hashlib
        hid = hashlib.sha256()                   # same provenance stamp as oracle/Makefile (tests/oracle_lib.py refuses an unstamped checker)
        for f in ("ivf_oracle.c", "ivf_oracle.h", "stl_pin.cpp", os.path.join("..", "include", "ivf_pattern31.inc")):
            with open(os.path.join(ROOT, "oracle", f), "rb") as fh:
                hid.update(fh.read())
        subprocess.check_call(["gcc"] + flags.split() + ["-DORC_BUILD_ID=\"%s\"" % hid.hexdigest()[:16],
                               "-fPIC", "-std=gnu11", "-shared", "-o", so, os.path.join(ROOT, "oracle", "ivf_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        so, flags = None, "-O3 -march=x86-64-v2 -ffp-contract=off (no compiler on this host: the portable build)"
    env = dict(os.environ)
    if so:
        env["IVF_ORACLE_SO"] = so
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--config", str(config)] +
                           ([] if introspect else ["--no-introspect"]), env=env, capture_output=True, text=True, timeout=1200)
        line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
        if r.returncode != 0 or not line:
            raise RuntimeError("worker exit code %d: %s" % (r.returncode, r.stderr[-1500:]))
    except Exception as e:                       # the GPU measurements above must still be printed: report the failure in the line
        return {"value": None, "unit": "pairs/s", "cores": None, "kind": "port", "sample": None,
                "error": "cpu baseline worker failed: %s" % (str(e)[-1500:],)}
    w = json.loads(line[0][len("CPU_BASELINE "):])
    same = w["checksum"] == oracle_checksum()
    v = w["extract_match_all_cores"]
    txt = ("%d pairs of the same %dx%d/%d-feature workload, one pair per thread, %.1f s wall (oracle/ivf_oracle.c, scalar C, %s)"
           % (w["sample_pairs"], W, H, NFEAT, w["secs"], flags))
    extra = {"extract_match_all_cores": round(v, 2), "extract_match_two_threads_per_pair": round(w["extract_match_two_threads_per_pair"], 2),
             "oracle_build": flags, "native_build_equals_portable_build": bool(same)}
    if introspect:
        fv = w["fcn_images_per_s"]
        txt += ("; + introspection FCN as a frozen TorchScript module (torch.jit.trace + freeze of the layer list, oracle/fcn_oracle_torch.py; "
                "PyTorch CPU kernels, %d intra-op threads, batches of 8), %.1f s wall = %.2f images/s; extract+match alone = %.1f pairs/s; "
                "value = 1/(1/a + 1/b)" % (w["cores"], w["fcn_secs"], fv, v))
        extra["fcn_images_per_s"] = round(fv, 2); extra["fcn_threads"] = w["cores"]
        extra["reference_threading_model"] = round(1.0 / (1.0 / w["extract_match_two_threads_per_pair"] + 1.0 / fv), 3)
        v = 1.0 / (1.0 / v + 1.0 / fv)
    return dict({"value": round(v, 3), "unit": "pairs/s", "cores": w["cores"], "kind": "port", "sample": txt}, **extra)


def parity_spot_check(spot, introspect):
    """The oracle (checker only) on the inputs of 4 pairs of the last TIMED sub-batch against what the timed launch sequence
    left in HBM for them: keypoints, descriptors, mvuRight / mvDepth bit for bit, mvKeyQualScore from the same cost map."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O          # checker only -- never on the product path
    b = BF / FX
    bad = []
    for s in spot:
        oL = O.Extractor(NFEAT, 1.2, 8, INI_TH, MIN_TH, introspection=bool(introspect)); oR = O.Extractor(NFEAT, 1.2, 8, INI_TH, MIN_TH)
        kL, dL = oL(s["L"], s["cost"]); kR, dR = oR(s["R"], None)
        ur, dp = O.stereo_match(oL, oR, kL, dL, kR, dR, BF, b)
        ok = (s["l"]["kps"].tobytes() == kL.tobytes() and s["r"]["kps"].tobytes() == kR.tobytes() and
              np.array_equal(s["l"]["desc"], dL) and np.array_equal(s["r"]["desc"], dR) and
              s["l"]["uright"].tobytes() == ur.tobytes() and s["l"]["depth"].tobytes() == dp.tobytes())
        if ok and s["cost"] is not None:                                   # mvKeyQualScore (Frame.cc:130-143)
            px = np.rint(kL["x"]).astype(int); py = np.rint(kL["y"]).astype(int)
            c = s["cost"][py, px].astype(np.float32)
            q = (np.float64(1.0) / (np.float64(1.0) + (c / np.float32(256)).astype(np.float64))).astype(np.float32)
            ok = np.array_equal(s["l"]["quality"], (np.float32(2) * q - np.float32(1)).astype(np.float32))
        if not ok:
            bad.append(s["pair"])
    return {"pairs": len(spot), "ok": not bad, "mismatching_pairs": bad, "sub_batch": "last timed launch sequence",
            "compared": "keypoints (6 fields), descriptors, mvuRight, mvDepth" + (", mvKeyQualScore; left extractor gated by the FCN's u8 cost map"
                                                                                 if introspect else ""),
            "checker": "oracle/libivf_oracle.so"}


def local_points_report(torch, iv, rec_buf, recs, tpairs_h, sc, cam, stream, O, PO):
    """Tracking::SearchLocalPoints, batched (ivf_tracker_search_local): frame b of every (a, b) pair searches a local map made of
    frame a's stereo points (world = camera frame, identity poses; mfMaxDistance = dist * scale[octave] as UpdateNormalAndDepth
    leaves it).  Duration by HIP events over 10 launch sequences; 2 frames checked against the oracle."""
    import numpy as np
    from iv_slam_amd._lib import LOCAL_POINT_DTYPE
    F = np.float32
    dev = rec_buf.device
    chunks, off = [], [0]
    invfx = F(F(1.0) / F(cam["fx"])); invfy = F(F(1.0) / F(cam["fy"]))
    for a, _ in tpairs_h:
        r = recs[a]
        sel = np.nonzero(r["depth"] > 0)[0]
        z = r["depth"][sel].astype(F)
        x = ((r["kps"]["x"][sel] - F(cam["cx"])) * z * invfx).astype(F); y = ((r["kps"]["y"][sel] - F(cam["cy"])) * z * invfy).astype(F)
        pos = np.stack([x, y, z], 1).astype(F)
        dist = np.sqrt((pos.astype(np.float64) ** 2).sum(1)).astype(F)
        arr = np.zeros(len(sel), LOCAL_POINT_DTYPE)
        arr["pos"] = pos; arr["normal"] = (pos / dist[:, None]).astype(F)
        arr["max_distance"] = (dist * sc[r["kps"]["octave"][sel]]).astype(F); arr["min_distance"] = (arr["max_distance"] / sc[-1]).astype(F)
        arr["desc"] = r["desc"][sel]; arr["flags"] = 2
        chunks.append(arr); off.append(off[-1] + len(sel))
    pts = np.concatenate(chunks) if chunks else np.zeros(1, LOCAL_POINT_DTYPE)
    nfr = len(tpairs_h)
    M = max(1, max(len(c) for c in chunks))
    dpts = torch.from_numpy(pts.view(np.uint8).reshape(-1)).to(dev); doff = torch.tensor(off, dtype=torch.int32, device=dev)
    dfr = torch.tensor([b for _, b in tpairs_h], dtype=torch.int32, device=dev)
    trl = iv.BatchTracker(NFEAT, sc, cam["fx"], cam["fy"], cam["cx"], cam["cy"], BF, (0.0, 0.0, float(W), float(H)), max_pairs=nfr,
                          device_id=dev.index or 0)
    la = torch.empty((nfr, NFEAT), dtype=torch.int32, device=dev); ln = torch.empty(nfr, dtype=torch.int32, device=dev)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    trl.search_local(rec_buf, dfr, dpts, doff, M, la, ln, th=1.0, nn_ratio=0.8, stream_ptr=stream.cuda_stream)
    e0.record(stream)
    for _ in range(10):
        trl.search_local(rec_buf, dfr, dpts, doff, M, la, ln, th=1.0, nn_ratio=0.8, stream_ptr=stream.cuda_stream)
    e1.record(stream); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10 / max(nfr, 1)
    lah, lnh = la.cpu().numpy(), ln.cpu().numpy()
    ok = True
    I = np.eye(4, dtype=F)
    for k in sorted({0, nfr - 1}):
        b = tpairs_h[k][1]; r = recs[b]
        cur = dict(kps=r["kps"], desc=r["desc"], uright=r["uright"], depth=r["depth"], T=I, scale=sc, fx=F(cam["fx"]), fy=F(cam["fy"]),
                   cx=F(cam["cx"]), cy=F(cam["cy"]), mbf=F(BF), mb=F(F(BF) / F(cam["fx"])), bounds=(0.0, 0.0, float(W), float(H)),
                   logScale=F(O.lib.orc_logf(float(sc[1]))))
        c = chunks[k]
        plist = [dict(pos=c[j]["pos"], normal=c[j]["normal"], minDist=c[j]["min_distance"], maxDist=c[j]["max_distance"], desc=c[j]["desc"],
                      skip=False, nObs=1) for j in range(len(c))]
        onm, oa = PO.search_local_points_frame(O, cur, plist, None, F(1.0), F(0.8))
        ok = ok and onm == int(lnh[k]) and np.array_equal(oa, lah[k, :len(oa)])
    return {"us_per_frame": round(us, 3), "frames_per_launch_sequence": nfr, "map_points_per_frame": round(float(np.mean([len(c) for c in chunks])), 1),
            "mean_matches": round(float(lnh.mean()), 1), "parity_ok": bool(ok), "frames_checked_vs_oracle": 2,
            "what": "Tracking::SearchLocalPoints (isInFrustum + PredictScale + SearchByProjection(F, mapPoints), th 1, ratio 0.8) for every "
                    "frame at once, device-resident (ivf_tracker_search_local)"}


def track_report(torch, iv, tracker, rec_buf, tpairs, tpairs_h, assign_h, nm_h, sc, cam, stream):
    """The batched tracker step on the last timed sub-batch's records: (i) what it left in HBM inside the timed region against
    the oracle for 3 frame pairs (projection loops: oracle/projection_oracle.py; window search + greedy replay: the C oracle),
    (ii) its own duration (HIP events, 10 launch sequences, nothing else on the GPU), (iii) the C oracle's
    SearchByProjection(cur, last) on ONE host core for the same frame pairs."""
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_lib as O          # checker only
    import projection_oracle as PO  # checker only
    from iv_slam_amd.frontend import unpack_gather_records
    F = np.float32
    recs = unpack_gather_records(rec_buf.cpu().numpy(), NFEAT)
    I = np.eye(4, dtype=F)
    bounds = (0.0, 0.0, float(W), float(H))

    def fd(r):
        return dict(kps=r["kps"], desc=r["desc"], uright=r["uright"], depth=r["depth"], T=I, scale=sc, fx=F(cam["fx"]), fy=F(cam["fy"]),
                    cx=F(cam["cx"]), cy=F(cam["cy"]), mbf=F(BF), mb=F(F(BF) / F(cam["fx"])), bounds=bounds)
    ok = True
    sample = sorted({0, len(tpairs_h) // 2, len(tpairs_h) - 1})
    for k in sample:
        a, b = tpairs_h[k]
        onm, oa = PO.track_with_motion_model_matches(O, fd(recs[b]), fd(recs[a]), F(7.0), F(14.0), 20)
        ok = ok and onm == int(nm_h[k]) and np.array_equal(oa, assign_h[k, :len(oa)])
    a3 = torch.empty_like(torch.from_numpy(assign_h)).to(rec_buf.device); n3 = torch.empty(len(tpairs_h), dtype=torch.int32, device=rec_buf.device)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    tracker.run(rec_buf, tpairs, a3, n3, stream_ptr=stream.cuda_stream)
    e0.record(stream)
    for _ in range(10):
        tracker.run(rec_buf, tpairs, a3, n3, stream_ptr=stream.cuda_stream)
    e1.record(stream); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10 / len(tpairs_h)
    ok = ok and np.array_equal(a3.cpu().numpy(), assign_h) and np.array_equal(n3.cpu().numpy(), nm_h[:len(tpairs_h)])
    # one core, C oracle: flat queries of the same pairs (zero-motion prior), search + greedy + rotation filter
    t_c = []
    for k in sample:
        la, cu = recs[tpairs_h[k][0]], recs[tpairs_h[k][1]]
        sel = la["depth"] > 0; lk = la["kps"][sel]
        q = dict(u=lk["x"], v=lk["y"], ur=la["uright"][sel], radius=(F(7.0) * sc[lk["octave"]]).astype(F), min_level=(lk["octave"] - 1).astype(np.int32),
                 max_level=(lk["octave"] + 1).astype(np.int32), angle=lk["angle"].copy(), desc=la["desc"][sel].copy(),
                 valid=np.ones(len(lk), np.uint8), blocks=np.ones(len(lk), np.uint8))
        t0 = time.perf_counter()
        for _ in range(20):
            O.search_by_projection(cu["kps"], cu["desc"], cu["uright"], bounds, q, True)
        t_c.append((time.perf_counter() - t0) / 20)
    local = local_points_report(torch, iv, rec_buf, recs, tpairs_h, sc, cam, stream, O, PO)
    return {"search_local_points": local, "us_per_frame_pair": round(us, 3), "N": NFEAT, "frame_pairs_per_launch_sequence": len(tpairs_h),
            "mean_matches": round(float(nm_h[:len(tpairs_h)].mean()), 1), "parity_ok": bool(ok), "pairs_checked_vs_oracle": len(sample),
            "oracle_one_core_us_per_frame_pair": round(float(np.mean(t_c)) * 1e6, 1),
            "what": "Tracking::TrackWithMotionModel's matcher call (UpdateLastFrame stereo points -> SearchByProjection(cur, last), th 7, retry 14 "
                    "below 20 matches) for every consecutive frame pair, device-resident (ivf_tracker_run)"}


# algorithmic HBM bytes of the probed FCN launch per image: hidden tensor read once
# (960 x 64 x 64 f32), output written and residual read (160 x 64 x 64 f32 each); weights are L2-resident
FCN_PROBE_BYTES_PER_IMAGE = (960 + 160 + 160) * 64 * 64 * 4


def latency_batch1(iv, blob, introspect, iters=30):
    """Single-pair latency of the DROP-IN per-call path, host buffers in and out: ivf_fcn_forward -> cost map (host) ->
    ivf_extract left (with the map) and right on two host threads (ORB/src/Frame.cc:116-124; here: this thread and one
    persistent worker -- a Python thread start costs ~0.1 ms, a std::thread ~0.02) -> ivf_stereo_match.  Also the four calls
    timed on their own (median of `iters`), so that the host-side share of `value` is visible."""
    import concurrent.futures as cf
    import numpy as np
    from iv_slam_amd import synth
    L, R = synth.make_pair(W, H, seed=77, idx=0)
    bgr = np.stack([L, L // 2 + 40, 255 - L // 2], axis=-1).astype(np.uint8)
    eL = iv.ORBextractor(NFEAT, 1.2, 8, INI_TH, MIN_TH, bool(introspect)); eR = iv.ORBextractor(NFEAT, 1.2, 8, INI_TH, MIN_TH, False)
    fcn1 = iv.IntrospectionFCN(blob, (H, W), (H, W), max_batch=1) if introspect else None
    pool = cf.ThreadPoolExecutor(1)
    ms = []
    for it in range(iters + 3):
        t0 = time.perf_counter()
        cost = fcn1(bgr) if fcn1 is not None else None
        fr = pool.submit(eR, R)
        kL, dL = eL(L, cost)
        kR, dR = fr.result()
        ur, dp = iv.ComputeStereoMatches(eL, eR, kL, dL, kR, dR, BF, BF / FX)
        if it >= 3:
            ms.append((time.perf_counter() - t0) * 1e3)
    ms.sort()

    def alone(fn):
        fn(); ts = []
        for _ in range(iters):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return round(ts[len(ts) // 2], 3)
    parts = {"extract_left": alone(lambda: eL(L, cost)), "extract_right": alone(lambda: eR(R)),
             "stereo_match": alone(lambda: iv.ComputeStereoMatches(eL, eR, kL, dL, kR, dR, BF, BF / FX))}
    if fcn1 is not None:
        parts["fcn_forward"] = alone(lambda: fcn1(bgr))
    pool.shutdown()
    return {"value": round(ms[len(ms) // 2], 3), "min": round(ms[0], 3), "unit": "ms per stereo pair", "iters": iters,
            "keypoints_left": int(len(kL)), "stereo_matches": int((ur >= 0).sum()), "calls_alone_ms": parts,
            "mode": "drop-in per-call C-ABI on host buffers: " + ("ivf_fcn_forward + " if introspect else "") +
                    "ivf_extract x2 (two host threads, each handle on its own HIP stream) + ivf_stereo_match; blocking copies included"}


def link_probe(torch, dev, nbytes=256 << 20, reps=4):
    """what the host link itself does, next to `pcie_gb_s` (r06): plain pinned hipMemcpyAsync, H2D, D2H and both at once on two streams"""
    hp = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True); hq = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    dp = torch.empty(nbytes, dtype=torch.uint8, device=dev); dq = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    s1 = torch.cuda.Stream(dev); s2 = torch.cuda.Stream(dev)

    def timed(fn):
        fn(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps

    def h2d():
        with torch.cuda.stream(s1):
            dp.copy_(hp, non_blocking=True)

    def d2h():
        with torch.cuda.stream(s2):
            hq.copy_(dq, non_blocking=True)

    def both():
        h2d(); d2h()
    th, td, tb = timed(h2d), timed(d2h), timed(both)
    return {"h2d_gb_s": round(nbytes / th / 1e9, 2), "d2h_gb_s": round(nbytes / td / 1e9, 2), "both_directions_gb_s": round(2 * nbytes / tb / 1e9, 2),
            "bytes": nbytes, "note": "pinned host memory, one hipMemcpyAsync of 256 MiB per direction, nothing else running"}


def h2d_included(torch, iv, fe, fcn, dev, P, n_batches, left, right, bgr, cost, rec, sets=2, nocopy=False, two_streams=False):
    """Throughput with the inputs coming from PINNED HOST memory: per batch the images cross PCIe on a copy stream into one of two staging
    sets while the previous batch computes; the packed result records {n, kps, desc, uRight} go back to pinned host memory.  Never the
    headline `value`.  r06: with the FCN the left image crosses ONCE, as the colour image the FCN reads; the extractor's grey left image is
    made from it on the device (ivf_frontend_run_color = Tracking::GrabImageStereo's cvtColor, Tracking.cc:272-295, fused into the ingest):
    4 planes per pair instead of 5.  The colour image of this leg is B = G = R = the resident grey left image, whose conversion is that grey
    image again (the coefficients sum to 2^15), so the extraction works on the same pixels as the timed region."""
    n_host = min(left.shape[0], 2 * P)
    hr = torch.empty((n_host, H, W), dtype=torch.uint8, pin_memory=True); hr.copy_(right[:n_host])
    hl = hb = None
    if fcn is not None:
        hb = torch.empty((n_host, H, W, 3), dtype=torch.uint8, pin_memory=True); hb.copy_(left[:n_host].unsqueeze(-1).expand(-1, -1, -1, 3))
    else:
        hl = torch.empty((n_host, H, W), dtype=torch.uint8, pin_memory=True); hl.copy_(left[:n_host])
    stage = [dict(l=torch.empty((P, H, W), dtype=torch.uint8, device=dev) if fcn is None else None, r=torch.empty((P, H, W), dtype=torch.uint8, device=dev),
                  b=torch.empty((P, H, W, 3), dtype=torch.uint8, device=dev) if fcn is not None else None) for _ in range(sets)]
    out_dev = [torch.empty(P * rec, dtype=torch.uint8, device=dev) for _ in range(3)]
    out_host = [torch.empty(P * rec, dtype=torch.uint8, pin_memory=True) for _ in range(3)]
    copy = torch.cuda.Stream(dev); copy2 = torch.cuda.Stream(dev) if two_streams else None
    main = torch.cuda.current_stream(dev)
    ready = [torch.cuda.Event() for _ in range(sets)]; ready2 = [torch.cuda.Event() for _ in range(sets)]; free = [torch.cuda.Event() for _ in range(sets)]
    nsl = n_host // P

    def run(nb):
        for k in range(nb):
            st = stage[k % sets]; s0 = (k % nsl) * P
            with torch.cuda.stream(copy):
                if k >= sets:
                    copy.wait_event(free[k % sets])
                if not nocopy:
                    if copy2 is None:
                        st["r"].copy_(hr[s0:s0 + P], non_blocking=True)
                    if hb is not None:
                        st["b"].copy_(hb[s0:s0 + P], non_blocking=True)
                    else:
                        st["l"].copy_(hl[s0:s0 + P], non_blocking=True)
                ready[k % sets].record(copy)
            if copy2 is not None:
                with torch.cuda.stream(copy2):
                    if k >= sets:
                        copy2.wait_event(free[k % sets])
                    if not nocopy:
                        st["r"].copy_(hr[s0:s0 + P], non_blocking=True)
                    ready2[k % sets].record(copy2)
                main.wait_event(ready2[k % sets])
            main.wait_event(ready[k % sets])
            if fcn is not None:
                plane = fe.cost_plane(P, main.cuda_stream)              # the FCN writes its maps into the front end's own cost plane: no ingest copy
                fcn.forward_device(st["b"], cost_u8=plane, stream_ptr=main.cuda_stream)
                fe.run_color(st["b"], st["r"], plane, main.cuda_stream)
            else:
                fe.run(st["l"], st["r"], None, main.cuda_stream)
            free[k % sets].record(main)                    # fe.run made `main` wait until the inputs were ingested
            bs = torch.cuda.ExternalStream(fe.batch_stream(0), device=dev)
            fe.pack_gather_block(out_dev[k % 3], fe.STREAM_OF_BATCH)
            with torch.cuda.stream(bs):
                out_host[k % 3].copy_(out_dev[k % 3], non_blocking=True)
        fe.sync(); torch.cuda.synchronize(dev)

    run(3)
    t0 = time.perf_counter()
    run(n_batches)
    dt = time.perf_counter() - t0
    in_bytes = P * H * W * (1 + (3 if fcn is not None else 1))
    return {"value": round(P * n_batches / dt, 2), "unit": "pairs/s", "batches": n_batches, "pairs_per_batch": P,
            "h2d_bytes_per_pair": in_bytes // P, "d2h_bytes_per_pair": rec, "planes_per_pair": 4 if fcn is not None else 2,
            "pcie_gb_s": round((in_bytes + P * rec) * n_batches / dt / 1e9, 2),
            "link": link_probe(torch, dev),
            "note": "inputs in pinned host memory, H2D on a copy stream double-buffered against compute, packed result records D2H; with the FCN the left "
                    "image crosses once, as colour (grey conversion on the device: ivf_frontend_run_color)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[k]: 2 (default) = the configuration the "
                    "metric is quoted on; 1 = the same without introspection; 3 = 2000 features; 4 = 1920x1200 / 4000 features / FAST 12/7")
    ap.add_argument("--pairs", type=int, default=0, help="stereo pairs per launch sequence (sub-batch) per GPU (default: 128; 32 for --config 4)")
    ap.add_argument("--batches-per-step", type=int, default=0, help="sub-batches per step (default 16 with introspection, 64 without)")
    ap.add_argument("--stream", type=int, default=0, help="distinct pairs resident per GPU (default: 512; 128 for --config 4)")
    ap.add_argument("--introspect", action="store_true", help="(default) configs[2]: run the introspection FCN on every left image and gate keypoints with it")
    ap.add_argument("--no-introspect", action="store_true", help="configs[1]: extract + match only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help="skip the post-run latency / PCIe-inclusive / configs[1] measurements")
    ap.add_argument("--force-gather", action="store_true", help="test aid: run the multi-GPU exchange step (RCCL all-gather of "
                    "the descriptor records) even with one rank")
    ap.add_argument("--track", action="store_true", help=argparse.SUPPRESS)      # r03 flag, now the default
    ap.add_argument("--no-track", action="store_true", help="leave the batched tracker step (pack of the result records + SearchByProjection(cur, last) "
                    "for every consecutive frame pair, ivf_tracker_run) out of the timed step.  By default it runs inside it at EVERY rank count, "
                    "so that the 1/2/4/8-GPU lines time the same work per frame (with ranks > 1 it consumes the all-gathered records)")
    ap.add_argument("--no-cost-plane", action="store_true", help="A/B aid: the FCN writes its cost maps into a buffer of the caller and ivf_frontend_run ingests them (a copy "
                    "into the pitched plane, the r01-r05 path) instead of writing them into the front end's plane itself (r06 default)")
    ap.add_argument("--no-carry", action="store_true", help="A/B aid: leave out the frame pair at every batch boundary (the r04 behaviour: world * P - 1 "
                    "pairs per launch sequence).  Default: the last record of a batch is carried into the next batch's buffer and tracked (r05)")
    ap.add_argument("--serial", action="store_true", help="profiling aid: wait for each batch before enqueuing the next, so "
                    "rocprofv3 kernel durations are not inflated by the overlap of consecutive batches")
    args = ap.parse_args()
    cfg = apply_config(args.config)
    args.introspect = cfg["introspect"] and not args.no_introspect
    if args.pairs <= 0:
        args.pairs = cfg["pairs"]
    if args.stream <= 0:
        args.stream = cfg["stream"]
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args.introspect)
    BPS = args.batches_per_step if args.batches_per_step > 0 else (16 if args.introspect else 64)

    # N>1 without a launcher: start one rank per GPU as a FRESH child (torch.distributed.run) before this process has
    # imported torch or touched the GPU -- never re-exec a process that has initialised HIP -- and pass its exit code on.
    # Rank 0 of the child prints the single JSON line (stdout is inherited).
    backend = os.environ.get("IVF_BENCH_BACKEND", "nccl")              # "gloo": TEST AID -- ranks may share one device, blocks cross the host
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        import torch as _t                      # device_count() alone does not initialise the GPU (no HIP context in this parent)
        if _t.cuda.device_count() < args.gpus and backend != "gloo":
            raise SystemExit("bench.py: --gpus %d but only %d device(s) are visible" % (args.gpus, _t.cuda.device_count()))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # --standalone: torchrun picks its own rendezvous port (no bind / close / re-bind race)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    numa = None
    if world > 1 and backend != "gloo":
        # one process per GPU: keep this rank's host threads on the NUMA node of ITS device.  Done here, in the fresh rank process,
        # from sysfs alone -- before torch / HIP are imported or the GPU is touched.
        from iv_slam_amd import dist as ivd
        numa = ivd.bind_rank_to_numa(local_rank)
    import torch
    import torch.distributed as dist
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d ...) or drop WORLD_SIZE and let bench.py spawn them"
                         % (args.gpus, world, args.gpus, args.gpus))
    exchange = world > 1 or args.force_gather
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)     # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()             # test aid: ranks share the visible device(s)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import iv_slam_amd as iv
    P = args.pairs
    n_stream = max(args.stream, P)
    n_stream = (n_stream + P - 1) // P * P
    left, right = make_device_stream(torch, dev, n_stream, seed=100, rank=rank, world=world)
    cost = None
    fcn = None
    blob = None
    bgr = None
    if args.introspect:
        # configs[2]: the introspection FCN (random-init weights of the deployed architecture; no checkpoints
        # exist offline) runs on the left colour image of every pair inside the timed step and its u8 cost map
        # gates keypoint selection in the left extractor.
        from iv_slam_amd import fcn_weights
        blob = fcn_weights.pack_blob(fcn_weights.make_seeded_weights(7))
        fcn = iv.IntrospectionFCN(blob, (H, W), (H, W), max_batch=P, device_id=local_rank)
        fcn.probe_enable()
        bgr = torch.stack([left, left // 2 + 40, 255 - left // 2], dim=-1).contiguous()      # [n,H,W,3] colour-ish
        cost = torch.empty((P, H, W), dtype=torch.uint8, device=dev)
    fe = iv.StereoFrontend(W, H, P, nfeatures=NFEAT, iniThFAST=INI_TH, minThFAST=MIN_TH, enableIntrospection=args.introspect, bf=BF, fx=FX,
                           device_id=local_rank)
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream
    rec = fe.gather_record_bytes()
    nslices = n_stream // P
    # The exchange step (pack + all-gather) is enqueued on the internal stream the batch itself runs on -- in order behind
    # it, so no stream ever waits on a batch-completion event (a stream that does costs 8 % of configs[2]: DESIGN.md
    # section 6) and the batch that reuses the context three steps later simply queues behind the collective.  One block /
    # gather buffer per internal stream.
    # every record buffer the tracker reads has ONE MORE record behind the world * P gathered ones: the carry record = the last global
    # frame of the previous batch (iv_slam_amd.dist.BoundaryCarry), so that the pair at the batch boundary is tracked too
    blocks3 = [torch.zeros(P * rec, dtype=torch.uint8, device=dev) for _ in range(3)] if exchange else None
    gathered3 = [torch.zeros((world * P + 1) * rec, dtype=torch.uint8, device=dev) for _ in range(3)] if exchange else None
    NG = world * P * rec                 # bytes of the gathered records proper
    nsub = [0]
    track = exchange or not args.no_track
    tracker = None
    carry = None
    if track:
        # the exchange step's consumer: Tracking::TrackWithMotionModel's matcher call for every frame this rank extracted against
        # the frame before it, wherever that one was extracted (ivf_tracker_run on the gathered records; zero-motion prior,
        # th = 7, retry with 14 below 20 matches: Tracking.cc:1313-1330)
        sc = iv.ORBextractor(NFEAT, 1.2, 8, INI_TH, MIN_TH, device_id=local_rank).GetScaleFactors()
        cam = dict(fx=FX, fy=FX, cx=W / 2 + 0.5, cy=H / 2 - 0.25)
        use_carry = not args.no_carry
        tpairs_h = track_pairs(world, rank, P, carry=use_carry)
        # one tracker per internal stream of the front end: a handle owns the scratch of one run at a time
        trackers = [iv.BatchTracker(NFEAT, sc, cam["fx"], cam["fy"], cam["cx"], cam["cy"], BF, (0.0, 0.0, float(W), float(H)),
                                    max_pairs=max(len(tpairs_h), 1), device_id=local_rank) for _ in range(3)]
        tracker = trackers[0]
        tpairs = torch.tensor(tpairs_h, dtype=torch.int32, device=dev).reshape(-1, 2)
        if blocks3 is None:
            blocks3 = [torch.zeros((P + 1) * rec, dtype=torch.uint8, device=dev) for _ in range(3)]
        from iv_slam_amd import dist as ivd
        # only rank 0 tracks a pair that reaches into the previous batch (frame k -> rank k mod world)
        carry = ivd.BoundaryCarry(gathered3 if exchange else blocks3, world, P, rec) if (use_carry and rank == 0) else None
        assign3 = [torch.full((max(len(tpairs_h), 1), NFEAT), -1, dtype=torch.int32, device=dev) for _ in range(3)]
        nm3 = [torch.zeros(max(len(tpairs_h), 1), dtype=torch.int32, device=dev) for _ in range(3)]

    coll_stream = [None]                 # the HIP stream the last collective was enqueued on (reported in `exchange`)

    def all_gather_block(bs, k):
        coll_stream[0] = bs.cuda_stream
        if backend == "gloo":
            # test aid: the block crosses the host (gloo has no device path); blocking, never a measured configuration
            bs.synchronize()
            hb = blocks3[k % 3].cpu()
            parts = [torch.empty_like(hb) for _ in range(world)]
            dist.all_gather(parts, hb)
            with torch.cuda.stream(bs):
                gathered3[k % 3][:NG].copy_(torch.cat(parts), non_blocking=False)
        else:
            with torch.cuda.stream(bs):
                dist.all_gather_into_tensor(gathered3[k % 3][:NG], blocks3[k % 3])

    cost_plane_mode = fcn is not None and not args.no_cost_plane
    last_plane = [None]

    def sub_batch(i):
        s = (i % nslices) * P
        k = nsub[0]; nsub[0] += 1
        if fcn is not None and cost_plane_mode:
            # r06: the FCN writes its u8 maps straight into the level-0 cost plane of the batch context this run uses; the run skips their ingest
            plane = fe.cost_plane(P, sptr)
            last_plane[0] = plane
            fcn.forward_device(bgr[s:s + P], cost_u8=plane, stream_ptr=sptr)
            fe.run_color(left[s:s + P], right[s:s + P], plane, sptr)
        else:
            if fcn is not None:
                fcn.forward_device(bgr[s:s + P], cost_u8=cost, stream_ptr=sptr)
            fe.run(left[s:s + P], right[s:s + P], cost, sptr)
        if exchange or track:
            # the path's one exchange step: all-gather of {n, kps, desc, uRight, depth} for cross-frame matching ...
            bs = torch.cuda.ExternalStream(fe.batch_stream(0), device=dev)
            fe.pack_gather_block(blocks3[k % 3], fe.STREAM_OF_BATCH)
            if exchange:
                all_gather_block(bs, k)
            if track and len(tpairs_h):
                # ... and its consumer, in order behind the collective on the batch's own stream.  The last record of this batch goes
                # into the carry slot of the NEXT batch's buffer; this batch's tracker waits for the previous batch's hand-over (the
                # one cross-stream wait of a launch sequence: the boundary pair needs both batches)
                if carry is not None:
                    carry.publish(k, bs)
                    carry.acquire(k, bs)
                trackers[k % 3].run(gathered3[k % 3] if exchange else blocks3[k % 3], tpairs, assign3[k % 3], nm3[k % 3], stream_ptr=bs.cuda_stream)
                if carry is not None:
                    carry.release(k, bs)
        if args.serial:
            fe.sync()

    def step(i):
        for j in range(BPS):
            sub_batch(i * BPS + j)

    for i in range(args.warmup):
        step(i)
    fe.sync()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    launches0 = iv.load().ivf_debug_launch_count()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    host_enqueue_bp_ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)   # host time in the enqueue loop of one step: includes
    launches = iv.load().ivf_debug_launch_count() - launches0                    # waiting on the full HIP queue (back-pressure)
    fe.sync()                                           # batches run on the front end's own streams: wait for all of
    torch.cuda.synchronize(dev)                         # them (also checks the device-side consistency flags)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    n_timed = min(args.steps * BPS, 64)
    if fcn is not None:
        fcn.status(sptr)                 # raises if any forward of the timed region drove an un-clamped activation out of the f16 range
    fast_sum_ms, fast_n = fe.fast_ms_stats(n_timed)
    if fcn is not None:
        probe_sets = {}
        for which in (0, 1):
            fcn.probe_select(which); probe_sets[which] = fcn.probe_stats(n_timed)
        fcn.probe_select(0)
        probe_sum_ms, probe_n, probe_batch = probe_sets[0]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_pairs = P * BPS * args.steps * world
    value = total_pairs / dt

    # sanity on the last batch: keypoints were really produced and matched
    r0 = fe.fetch(0, 0)
    assert len(r0["kps"]) > NFEAT // 4 and (r0["uright"] >= 0).sum() > 20, "degenerate output"
    # parity spot check: 4 pairs of the LAST timed sub-batch -- results, inputs and (configs[2]) the cost maps that gated them
    # are copied out now, before anything else runs, and compared with the oracle below (never inside the timed region)
    s_last = ((nsub[0] - 1) % nslices) * P
    spot_pairs = sorted({0, 1, P // 2, P - 1})
    if last_plane[0] is not None:
        cost.copy_(last_plane[0])                  # the maps that gated the last sub-batch, out of the context's plane (after the timed region)
    spot = [dict(pair=p, L=left[s_last + p].cpu().numpy(), R=right[s_last + p].cpu().numpy(),
                 cost=cost[p].cpu().numpy() if fcn is not None else None, l=fe.fetch(p, 0), r=fe.fetch(p, 1)) for p in spot_pairs]
    exch = None
    if exchange:
        # the last collective really delivered every rank's records: own slot == own block, every count plausible
        import numpy as np
        last = (nsub[0] - 1) % 3
        g = gathered3[last][:NG].cpu().numpy().reshape(world, P, rec)
        own = blocks3[last][:P * rec].cpu().numpy().reshape(P, rec)
        assert np.array_equal(g[rank], own), "all-gather: own slot differs from the packed block"
        counts = g[:, :, :4].copy().view(np.int32)[:, :, 0]
        assert ((counts > NFEAT // 4) & (counts <= NFEAT)).all(), "all-gather: implausible keypoint counts %r" % counts
        exch = {"world": world, "world_size_seen_by_backend": dist.get_world_size(), "record_bytes": rec, "bytes_per_rank_per_batch": P * rec,
                "records_checked": int(counts.size), "backend": backend, "consumed": False,
                "enqueued_on_stream": "0x%x" % coll_stream[0] if coll_stream[0] is not None else None,
                "batch_stream_of_that_run": "0x%x" % fe.batch_stream(0),
                "rank0_numa_binding": numa,
                "collective": ("all_gather_into_tensor (RCCL) on the batch's own internal stream" if backend != "gloo" else
                               "TEST AID: gloo all_gather of host copies (ranks may share a device)")}
    trk = None
    if track and len(tpairs_h):
        # what the tracker left in HBM for the LAST timed sub-batch, against the oracle on the same records (3 frame pairs)
        last = (nsub[0] - 1) % 3
        rec_buf = (gathered3[last] if exchange else blocks3[last]).clone()
        trk = track_report(torch, iv, tracker, rec_buf, tpairs, tpairs_h, assign3[last].cpu().numpy(), nm3[last].cpu().numpy(), sc, cam, stream)
        if not (trk["parity_ok"] and trk["search_local_points"]["parity_ok"]):
            print("bench.py: rank %d: TRACKER PARITY CHECK FAILED: %s" % (rank, json.dumps(trk)), file=sys.stderr, flush=True)
            raise SystemExit(4)
        if exch is not None:
            exch["consumed"] = True
            exch["consumer"] = "ivf_tracker_run: SearchByProjection(cur, last) of every frame this rank extracted against the frame before it, out of the gathered records"

    # host cost of enqueuing one step when nothing pushes back: two launch sequences into EMPTY queues (they fit), timed on the host
    fe.sync(); torch.cuda.synchronize(dev)
    enq = []
    for rep in range(3):
        t1 = time.perf_counter()
        sub_batch(2 * rep); sub_batch(2 * rep + 1)
        enq.append((time.perf_counter() - t1) / 2)
        fe.sync(); torch.cuda.synchronize(dev)
    host_enqueue_ms = min(enq) * 1e3 * BPS

    # ---- after the timed region -------------------------------------------------------------------------------------
    # The front end overlaps consecutive batches on its own streams, so inside the timed region the probed kernel shares
    # the GPU with other kernels and its event-measured duration is inflated.  For the kernel's own figure, 5 more
    # sub-batches are run one at a time (sync between them) and reported separately as `isolated`.
    for i in range(5):
        sub_batch(i)
        fe.sync()
        torch.cuda.synchronize(dev)
    iso_sum_ms, iso_n = fe.fast_ms_stats(5)
    if fcn is not None:
        iso_sets = {}
        for which in (0, 1):
            fcn.probe_select(which); iso_sets[which] = fcn.probe_stats(5)
        fcn.probe_select(0)
        iso_probe_ms, iso_probe_n, _ = iso_sets[0]
    extras = world == 1 and not args.no_extras
    # the network alone (MFMA leg of north_star): 3 forwards with nothing else on the GPU, HIP events on its stream
    fcn_alone = None
    if fcn is not None and extras:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        fcn.forward_device(bgr[:P], cost_u8=cost, stream_ptr=sptr)
        e0.record(stream)
        for _ in range(3):
            fcn.forward_device(bgr[:P], cost_u8=cost, stream_ptr=sptr)
        e1.record(stream); torch.cuda.synchronize(dev)
        us_img = e0.elapsed_time(e1) * 1e3 / 3 / P
        # 17.229 GFLOP per forward, of which the 1x1 convs (14.61) and the decoder 3x3 (1.89) run on MFMA, each
        # product as three f16 MFMAs (split-f16: hi*hi + hi*lo + lo*hi, f32 accumulate)
        fcn_alone = {"us_per_image": round(us_img, 2), "batch": P,
                     "f32_equivalent_tflops": round(17.229e9 / us_img / 1e6, 1),
                     "mfma_f16_tflops_issued": round(3 * (14.61 + 1.89) * 1e9 / us_img / 1e6, 1), "mfma_f16_dense_peak_tflops": 2500.0,
                     "mfma_frac": round(3 * (14.61 + 1.89) * 1e9 / us_img / 1e6 / 2500.0, 4),
                     "mfma_frac_algorithmic": round(17.229e9 / us_img / 1e6 / 2500.0, 4)}
    # configs[1] (no introspection) on the same stream of pairs, for reference next to the headline number
    em_only = None
    if fcn is not None and extras:
        fe1 = iv.StereoFrontend(W, H, P, nfeatures=NFEAT, iniThFAST=INI_TH, minThFAST=MIN_TH, enableIntrospection=False, bf=BF, fx=FX, device_id=local_rank)
        for i in range(3):
            s0 = (i % nslices) * P; fe1.run(left[s0:s0 + P], right[s0:s0 + P], None, sptr)
        fe1.sync(); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(64):
            s0 = (i % nslices) * P; fe1.run(left[s0:s0 + P], right[s0:s0 + P], None, sptr)
        fe1.sync(); torch.cuda.synchronize(dev)
        em_only = P * 64 / (time.perf_counter() - t1)
        del fe1
    if trk is None and extras:
        # the tracker step on its own (not part of `value`): the records of the batch the front end holds, consecutive frames of the
        # stream, against the oracle and next to the oracle's search on one core
        sc = iv.ORBextractor(NFEAT, 1.2, 8, INI_TH, MIN_TH, device_id=local_rank).GetScaleFactors()
        cam = dict(fx=FX, fy=FX, cx=W / 2 + 0.5, cy=H / 2 - 0.25)
        tp_h = track_pairs(1, 0, P)
        tr1 = iv.BatchTracker(NFEAT, sc, cam["fx"], cam["fy"], cam["cx"], cam["cy"], BF, (0.0, 0.0, float(W), float(H)), max_pairs=len(tp_h), device_id=local_rank)
        tp = torch.tensor(tp_h, dtype=torch.int32, device=dev).reshape(-1, 2)
        blk = torch.zeros(P * rec, dtype=torch.uint8, device=dev)
        a1 = torch.full((len(tp_h), NFEAT), -1, dtype=torch.int32, device=dev); n1 = torch.zeros(len(tp_h), dtype=torch.int32, device=dev)
        fe.sync()
        fe.pack_gather_block(blk, sptr)
        tr1.run(blk, tp, a1, n1, stream_ptr=sptr)
        torch.cuda.synchronize(dev)
        trk = track_report(torch, iv, tr1, blk, tp, tp_h, a1.cpu().numpy(), n1.cpu().numpy(), sc, cam, stream)
        trk["in_timed_region"] = False
        if not (trk["parity_ok"] and trk["search_local_points"]["parity_ok"]):
            print("bench.py: rank %d: TRACKER PARITY CHECK FAILED: %s" % (rank, json.dumps(trk)), file=sys.stderr, flush=True)
            raise SystemExit(4)
    h2d = None
    lat = None
    if extras:
        h2d = h2d_included(torch, iv, fe, fcn, dev, P, 16 if args.introspect else 48, left, right, bgr, cost, rec)
        if os.environ.get("IVF_BENCH_H2D_AB"):        # A/B aid (r06): the same leg without its copies, with three staging sets, with two copy streams
            h2d["ab"] = {"nocopy": h2d_included(torch, iv, fe, fcn, dev, P, 16 if args.introspect else 48, left, right, bgr, cost, rec, nocopy=True)["value"],
                         "three_sets": h2d_included(torch, iv, fe, fcn, dev, P, 16 if args.introspect else 48, left, right, bgr, cost, rec, sets=3)["value"],
                         "two_streams": h2d_included(torch, iv, fe, fcn, dev, P, 16 if args.introspect else 48, left, right, bgr, cost, rec, two_streams=True)["value"],
                         "three_sets_two_streams": h2d_included(torch, iv, fe, fcn, dev, P, 16 if args.introspect else 48, left, right, bgr, cost, rec, sets=3, two_streams=True)["value"],
                         "again": h2d_included(torch, iv, fe, fcn, dev, P, 16 if args.introspect else 48, left, right, bgr, cost, rec)["value"]}
        lat = latency_batch1(iv, blob, args.introspect)

    parity = parity_spot_check(spot, args.introspect)
    if not parity["ok"]:
        print("bench.py: rank %d: PARITY SPOT CHECK FAILED: %s" % (rank, json.dumps(parity)), file=sys.stderr, flush=True)
        raise SystemExit(3)
    if rank == 0:
        def load_pmc(kernel_key):
            # HBM traffic REPLAYED from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE are collected in separate
            # runs because counters cannot be read inside this process), per image, rescaled to this launch size.  The
            # entry is used only when it was recorded for the kernel the probe actually timed (name match); the source string
            # names the file, the commit it was collected at (when recorded) and the file's own hash.
            import hashlib
            if (W, H) != (1242, 375):
                return None, None                  # the committed counter passes were collected at 1242x375 only
            for name in ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
                try:
                    path = os.path.join(ROOT, "profiles", name)
                    raw = open(path, "rb").read()
                    pmc = json.loads(raw)
                    k = pmc["kernels"][kernel_key]
                    src = "replayed from profiles/%s (sha1 %s%s): separate rocprofv3 --pmc passes, not measured in this run" % (
                        name, hashlib.sha1(raw).hexdigest()[:12], (", collected on build %s" % pmc["build"]["libivfront"]) if pmc.get("build", {}).get("libivfront") else
                        (", collected at commit %s" % pmc["commit"] if "commit" in pmc else ""))
                    return (k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]) / k.get("images_per_launch", pmc["images_per_launch"]), src
                except Exception:
                    continue
            return None, None

        def roof(kernel, algo_bytes, sum_ms, n, iso_ms, iso_n, traffic, src):
            ms = sum_ms / max(n, 1); ims = iso_ms / max(iso_n, 1)
            ach = algo_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            iach = algo_bytes / (ims * 1e-3) / 1e9 if ims > 0 else 0.0
            return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": src,
                    "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(ms, 5), "launches_timed": n,
                    "isolated": {"note": "same kernel, 5 launches after the timed region with no other batch in flight",
                                 "avg_launch_ms": round(ims, 5), "achieved": round(iach, 2), "frac": round(iach / HBM_PEAK_GBS, 5)}}

        imgs_per_launch = 2 * P
        fast_algo = (LEVEL_PX_SUM + NFEAT * 8) * imgs_per_launch      # pyramid read + candidate output
        per_img, src = load_pmc("ivf::k_fast_nms")
        fast_roof = roof("k_fast_nms", fast_algo, fast_sum_ms, fast_n, iso_sum_ms, iso_n,
                         None if per_img is None else int(per_img * imgs_per_launch), src)
        if fcn is not None:
            # r05: `roofline` names the kernel with the largest TOTAL time per forward among the probed ones: block 15's kernel runs twice
            # (blocks 15 and 16), block 17's once; the other one is printed next to it as `roofline_other_block`
            # (same pricing: frac_issued / frac_algorithmic of the f16 matrix peak)
            cands = []
            for which in (0, 1):
                fcn.probe_select(which)
                try:
                    nm_, algo_ = fcn.probe_info()
                except Exception:
                    continue
                cands.append(dict(which=which, name=nm_, algo=algo_, timed=probe_sets[which], iso=iso_sets[which],
                                  launches_per_forward=2 if which == 0 else 1))
            fcn.probe_select(0)
            def total_ms(c):
                s_, n_, _b = c["timed"]
                return c["launches_per_forward"] * s_ / max(n_, 1)
            cands.sort(key=total_ms, reverse=True)
            top = cands[0]
            pname, palgo = top["name"], top["algo"]
            probe_sum_ms, probe_n, probe_batch = top["timed"]
            iso_probe_ms, iso_probe_n = top["iso"][0], top["iso"][1]
            per_img, src = load_pmc(pname)
            roofline = roof("%s (%d images)" % (pname, probe_batch), int(palgo * probe_batch), probe_sum_ms, probe_n,
                            iso_probe_ms, iso_probe_n, None if per_img is None else int(per_img * probe_batch), src)
            roofline["launches_per_forward"] = top["launches_per_forward"]
            roofline["total_ms_per_forward"] = round(total_ms(top), 5)
            # The same launch priced two more ways (r01 verdict): (i) at the BLOCK's boundary -- the hidden tensor this kernel
            # reads is a product of the expand / depthwise split, not of the network: block input + residual + output is all an
            # ideal whole-block kernel would move; (ii) against the matrix pipe -- three f16 MFMAs per f32 product.
            import re
            m3 = re.search(r"(\d+)->(\d+)->(\d+)", pname)
            m = re.search(r"(\d+)->(\d+)", pname)
            if m3 and roofline["avg_launch_ms"] > 0:
                # a whole inverted-residual block in one kernel (k_fcn_irbd4): the hidden tensor never reaches HBM, the launch moves
                # only the block's input and output (r05: the residual is rebuilt from the input fragments) (a few percent of the HBM roofline by construction) -- it is priced
                # against the MATRIX pipe: both 1x1 convolutions, three f16 MFMAs per f32 product (hi*hi + hi*lo + lo*hi).
                cin, hid, cout = (int(v) for v in m3.groups())
                ms = roofline["avg_launch_ms"]; ims = roofline["isolated"]["avg_launch_ms"]
                fl = 2.0 * (cin * hid + hid * cout) * 64 * 64 * 3 * probe_batch
                hbm = {k: roofline[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")}
                hbm["note"] = "block input + output at 64x64 f32 (the residual is the input, held in registers): what this launch moves algorithmically"
                # r06 (verdict item 4): `achieved` / `frac` are SURVEY 8(d)'s ALGORITHMIC figures -- 2 x MAC of both 1x1 convolutions,
                # counted once; the three f16 MFMAs every f32 product costs (what the matrix pipe executes) are `*_issued` beside them
                ach_algo = round(fl / 3.0 / (ms * 1e-3) / 1e12, 2)
                roofline.update({"bound": "mfma", "achieved": ach_algo, "peak": 2500.0, "unit": "TFLOP/s",
                                 "frac": round(ach_algo / 2500.0, 5), "flops_per_launch": fl / 3.0,
                                 "frac_algorithmic": round(ach_algo / 2500.0, 5),
                                 "achieved_issued": round(fl / (ms * 1e-3) / 1e12, 1),
                                 "frac_issued": round(fl / (ms * 1e-3) / 1e12 / 2500.0, 5),
                                 "issued_flops_per_launch": fl,
                                 "note": "algorithmic flops of this launch (expansion + projection, 2 x MAC) against the 2.5 PFLOP/s dense f16 peak; "
                                         "the matrix pipe executes three times that (hi*hi + hi*lo + lo*hi): *_issued; the launch is neither HBM- nor "
                                         "matrix-bound: DESIGN.md section 7 (r03) has its phase timing",
                                 "hbm": hbm})
                roofline["isolated"].update({"achieved": round(fl / 3.0 / (ims * 1e-3) / 1e12, 2) if ims > 0 else 0.0,
                                             "frac": round(fl / 3.0 / (ims * 1e-3) / 1e12 / 2500.0, 5) if ims > 0 else 0.0})
                if len(cands) > 1:
                    o = cands[1]
                    mo = re.search(r"(\d+)->(\d+)->(\d+)", o["name"])
                    s_, n_, b_ = o["timed"]
                    oms = s_ / max(n_, 1)
                    if mo and oms > 0:
                        ci, hi_, co = (int(v) for v in mo.groups())
                        ofl = 2.0 * (ci * hi_ + hi_ * co) * 64 * 64 * 3 * b_
                        out_other = {"kernel": "%s (%d images)" % (o["name"], b_), "bound": "mfma", "avg_launch_ms": round(oms, 5), "launches_timed": n_,
                                     "launches_per_forward": o["launches_per_forward"], "total_ms_per_forward": round(total_ms(o), 5),
                                     "achieved": round(ofl / 3.0 / (oms * 1e-3) / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s",
                                     "frac": round(ofl / 3.0 / (oms * 1e-3) / 1e12 / 2500.0, 5),
                                     "achieved_issued": round(ofl / (oms * 1e-3) / 1e12, 1),
                                     "frac_issued": round(ofl / (oms * 1e-3) / 1e12 / 2500.0, 5),
                                     "frac_algorithmic": round(ofl / 3.0 / (oms * 1e-3) / 1e12 / 2500.0, 5)}
                        roofline["other_block"] = out_other
            elif m and roofline["avg_launch_ms"] > 0:
                hid, cout = int(m.group(1)), int(m.group(2))
                cin = hid // 6                                     # MobileNetV2 expansion factor (mobilenet.py:36)
                ms = roofline["avg_launch_ms"]
                bb = (cin + cout + cout) * 64 * 64 * 4 * probe_batch
                fl = 2.0 * hid * cout * 64 * 64 * 3 * probe_batch
                roofline["block_boundary"] = {
                    "bytes_per_launch": bb, "achieved": round(bb / (ms * 1e-3) / 1e9, 2), "unit": "GB/s",
                    "frac": round(bb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "note": "block input (%d ch) + residual + output (%d ch) at 64x64 f32; the %d-channel hidden tensor is not counted" % (cin, cout, hid)}
                roofline["mfma"] = {"bound": "mfma", "achieved": round(fl / (ms * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                    "frac": round(fl / (ms * 1e-3) / 1e12 / 2500.0, 5), "flops_per_launch": fl,
                                    "note": "f16 MFMA flops issued by this launch (hi*hi + hi*lo + lo*hi); sustained clock under this kernel ~1.95 GHz"}
        else:
            roofline = fast_roof
        out = {
            "metric": METRIC, "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "host_enqueue_ms_per_step": round(host_enqueue_ms, 4),
            "host_enqueue_ms_per_step_incl_backpressure": round(host_enqueue_bp_ms, 4),
            "launches_per_step": round(launches / max(args.steps, 1), 1), "launches_per_launch_sequence": round(launches / max(args.steps * BPS, 1), 1),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 (ORB) + f32 via split-f16 MFMA (FCN)" if args.introspect else "u8",
            "data": "synthetic",
            "config": {"workload": (WORKLOAD if args.introspect == CONFIGS[args.config]["introspect"] else
                                    CONFIGS[1]["name"] if args.config == 2 else WORKLOAD + " -- run with --no-introspect"),
                       "baseline_config_index": args.config if (args.introspect or args.config != 2) else 1, "width": W, "height": H,
                       "pairs_per_step_per_gpu": P * BPS, "pairs_per_launch_sequence": P, "launch_sequences_per_step": BPS,
                       "distinct_pairs_per_gpu": n_stream, "nfeatures": NFEAT,
                       "nlevels": 8, "scale_factor": 1.2, "fast_thresholds": [INI_TH, MIN_TH],
                       "parallelism": "frames sharded %d-way, RCCL all-gather of descriptor blocks" % world if world > 1 else "1 GPU"},
            "timed_region_s": round(dt, 3),
            "track_in_timed_region": bool(track and len(tpairs_h)),
            "cost_maps": ("written by the FCN into the front end's level-0 cost plane (ivf_frontend_cost_plane): no ingest copy" if cost_plane_mode else
                          "FCN -> caller's buffer -> ingest copy" if fcn is not None else None),
            "build": {"libivfront": iv.load().ivf_build_id().decode(), "sources": iv._lib.source_build_id()},
            "roofline": roofline,
            "parity_spot_check": parity,
        }
        if exch is not None:
            out["exchange"] = exch
        if trk is not None:
            trk.setdefault("in_timed_region", True)
            out["track"] = trk
        if fcn is not None:
            out["roofline_fast_nms"] = fast_roof
            if fcn_alone is not None:
                out["fcn_forward"] = fcn_alone
            if em_only is not None:
                out["extract_match_only"] = {"value": round(em_only, 2), "unit": "pairs/s",
                                             "note": "configs[1] (introspection OFF), 64 launch sequences on the same resident pairs after the timed region"}
        if lat is not None:
            out["latency_ms_batch1"] = lat
        if h2d is not None:
            out["h2d_included"] = h2d
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline(args.introspect, args.config)
        print(json.dumps(out), flush=True)
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
