#!/usr/bin/env python3
"""Replay a KITTI-layout stereo sequence through the device front end: the per-frame part of the
reference's stereo driver (introspective_ORB_SLAM/Examples/Stereo/stereo_kitti.cc:437-573) up to the point where
System::TrackStereo would take over -- load pair, optional undistort/rectify remap, optional cost image (predicted
heat maps from disk, remapped like the left image, :470-521), extract L/R, stereo match.

  python tools/replay_kitti.py SEQUENCE_DIR SETTINGS.yaml [--rectify] [--undistort] [--qual DIR | --fcn WEIGHTS.bin] [--batch 16] [--track]
  python tools/replay_kitti.py --make-synthetic DIR --frames 12        # writes a small synthetic sequence + settings

Prints one line per frame (keypoints L/R, stereo matches, median depth) and the pairs/s of the device part.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SYNTH_SETTINGS = """%YAML:1.0
Camera.fx: 718.856
Camera.fy: 718.856
Camera.cx: 320.0
Camera.cy: 120.0
Camera.k1: 0.0
Camera.k2: 0.0
Camera.p1: 0.0
Camera.p2: 0.0
Camera.width: 640
Camera.height: 240
Camera.fps: 10.0
Camera.bf: 386.1448
Camera.RGB: 1
ThDepth: 35
LEFT.height: 240
LEFT.width: 640
LEFT.D: !!opencv-matrix
   rows: 1
   cols: 5
   dt: d
   data: [-0.05, 0.01, 0.0002, -0.0003, 0.0]
LEFT.K: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [716.0, 0.0, 322.0, 0.0, 717.0, 119.0, 0.0, 0.0, 1.0]
LEFT.R: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [0.999990, -0.002, -0.004, 0.002, 0.999995, 0.001, 0.004, -0.001, 0.999991]
LEFT.P: !!opencv-matrix
   rows: 3
   cols: 4
   dt: d
   data: [718.856, 0.0, 320.0, 0.0, 0.0, 718.856, 120.0, 0.0, 0.0, 0.0, 1.0, 0.0]
RIGHT.height: 240
RIGHT.width: 640
RIGHT.D: !!opencv-matrix
   rows: 1
   cols: 5
   dt: d
   data: [-0.04, 0.012, -0.0001, 0.0002, 0.0]
RIGHT.K: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [719.0, 0.0, 318.0, 0.0, 718.0, 121.0, 0.0, 0.0, 1.0]
RIGHT.R: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [0.999992, 0.003, 0.002, -0.003, 0.999994, -0.0015, -0.002, 0.0015, 0.999997]
RIGHT.P: !!opencv-matrix
   rows: 3
   cols: 4
   dt: d
   data: [718.856, 0.0, 320.0, -386.1448, 0.0, 718.856, 120.0, 0.0, 0.0, 0.0, 1.0, 0.0]
ORBextractor.nFeatures: 600
ORBextractor.scaleFactor: 1.2
ORBextractor.nLevels: 8
ORBextractor.iniThFAST: 20
ORBextractor.minThFAST: 7
"""


def make_synthetic(root, frames, with_qual=True):
    from iv_slam_amd import kitti, synth
    os.makedirs(os.path.join(root, "image_0"), exist_ok=True); os.makedirs(os.path.join(root, "image_1"), exist_ok=True)
    if with_qual:
        os.makedirs(os.path.join(root, "qual"), exist_ok=True)
    with open(os.path.join(root, "times.txt"), "w") as f:
        for i in range(frames):
            f.write("%e\n" % (0.1 * i))
    for i in range(frames):
        L, R = synth.make_pair(640, 240, seed=77, idx=i)
        kitti.imwrite(os.path.join(root, "image_0", "%06d.png" % i), L)
        kitti.imwrite(os.path.join(root, "image_1", "%06d.png" % i), R)
        if with_qual and i % 3 != 2:                               # every third frame has no predicted cost image
            kitti.imwrite(os.path.join(root, "qual", "%06d.png" % i), synth.make_cost_map(640, 240, seed=77, idx=i))
    with open(os.path.join(root, "settings.yaml"), "w") as f:
        f.write(SYNTH_SETTINGS)
    return os.path.join(root, "settings.yaml")


class Replay:
    """The per-frame device work of the driver, batched: remap (optional) -> StereoFrontend."""

    def __init__(self, settings, rectify=False, undistort=False, introspect=False, batch=16, device_id=0, fcn_blob=None, track=False):
        import torch
        import iv_slam_amd as iv
        self.torch = torch; self.iv = iv
        self.track = track; self.prev_left = None
        self.S = settings
        self.dev = torch.device("cuda:%d" % device_id)
        nf, sf, nl, ini, mn, _ = settings.extractor_params()
        bf, b = settings.stereo()
        self.rgb = bool(int(settings.get("Camera.RGB", 0)))
        self.remapL = self.remapR = None
        self.size = (int(settings["Camera.width"]), int(settings["Camera.height"]))
        if rectify or undistort:                                   # stereo_kitti.cc:285-343
            maps = []
            for side in ("LEFT", "RIGHT"):
                K, D, R, P, size = settings.rectification(side)
                if rectify and not undistort:
                    D = np.zeros(4)
                if undistort and not rectify:
                    R = np.eye(3)
                maps.append((iv.initUndistortRectifyMap(K, D, R, P, size), size))
            (mL, sL), (mR, sR) = maps
            self.remapL = iv.Remap(mL[0], mL[1], (sL[1], sL[0]), 1, device_id)
            self.remapR = iv.Remap(mR[0], mR[1], (sR[1], sR[0]), 1, device_id)
            self.size = sL
        self.fe = iv.StereoFrontend(self.size[0], self.size[1], batch, nf, sf, nl, ini, mn, enableIntrospection=introspect or fcn_blob is not None,
                                    bf=bf, b=b, device_id=device_id)
        self.batch = batch
        self.scale_factors = iv.ORBextractor(nf, sf, nl, ini, mn).GetScaleFactors()
        # online inference of the introspection network (stereo_kitti.cc:493-514): the UN-remapped left image goes in,
        # the cost map comes out at the same size and is then remapped like the left image (:519-521)
        self.fcn = None
        if fcn_blob is not None:
            src = (int(settings["Camera.height"]), int(settings["Camera.width"]))
            self.fcn = iv.IntrospectionFCN(fcn_blob, src, src, max_batch=batch, device_id=device_id)

    def run(self, lefts, rights, costs=None, raw_lefts=None):
        """lists of host grey images (cost entries may be None; raw_lefts = the left frames as loaded, B,G,R or grey, for the
        network) -> list of per-pair dicts (left/right fetch results)."""
        torch = self.torch
        n = len(lefts)
        L = torch.from_numpy(np.stack(lefts)).to(self.dev); R = torch.from_numpy(np.stack(rights)).to(self.dev)
        if self.remapL is not None:
            L = self.remapL.apply_device(L); R = self.remapR.apply_device(R)
        C = None
        if self.fcn is not None:
            Lraw = torch.from_numpy(np.stack(raw_lefts if raw_lefts is not None else lefts)).to(self.dev)
            bgr = Lraw if Lraw.dim() == 4 else Lraw.unsqueeze(-1).expand(-1, -1, -1, 3)      # grey frames: the same plane three times
            C = torch.empty(Lraw.shape[:3], dtype=torch.uint8, device=self.dev)
            self.fcn.forward_device(bgr.contiguous(), cost_u8=C, stream_ptr=torch.cuda.current_stream().cuda_stream)
            if self.remapL is not None:
                C = self.remapL.apply_device(C)
        elif costs is not None and any(c is not None for c in costs):
            if any(c is None for c in costs):
                raise ValueError("a batch mixes frames with and without a cost image: replay those frames with --batch 1")
            C = torch.from_numpy(np.stack(costs)).to(self.dev)
            if self.remapL is not None:
                C = self.remapL.apply_device(C)                    # :519-521
        torch.cuda.current_stream().synchronize()
        self.fe.run(L, R, C)
        self.fe.sync()
        res = [(self.fe.fetch(k, 0), self.fe.fetch(k, 1)) for k in range(n)]
        if self.track:
            self._track(res)
        return res

    # ---- pose-free replay of the tracker's matcher call (Tracking::TrackWithMotionModel, ORB/src/Tracking.cc:1303-1342):
    # the previous frame's stereo points are searched for in the current frame around their OLD image position (zero-motion
    # prior instead of the velocity model's projection; th = 15, octave window +-1, stereo consistency, rotation histogram:
    # ORBmatcher::SearchByProjection(cur, last), ORB/src/ORBmatcher.cc:1372-1518).  The current frame is a device-resident
    # frame made straight from the batch (no host copy of its keypoints / descriptors); results land in res[k][0]["tracked"].
    TRACK_TH = 15.0

    def track_queries(self, last):
        """flat query arrays of SearchByProjection(cur, last) from the previous left frame's fetch() result."""
        sel = last["uright"] >= 0
        lk = last["kps"][sel]
        sc = self.scale_factors
        return dict(u=lk["x"].astype(np.float32), v=lk["y"].astype(np.float32), ur=last["uright"][sel].astype(np.float32),
                    radius=(np.float32(self.TRACK_TH) * sc[lk["octave"]]).astype(np.float32), min_level=(lk["octave"] - 1).astype(np.int32),
                    max_level=(lk["octave"] + 1).astype(np.int32), angle=lk["angle"].copy(), desc=last["desc"][sel].copy(),
                    valid=np.ones(len(lk), np.uint8), blocks=np.ones(len(lk), np.uint8))

    def _track(self, res):
        bounds = (0.0, 0.0, float(self.size[0]), float(self.size[1]))
        for k in range(len(res)):
            last = self.prev_left if k == 0 else res[k - 1][0]
            if last is not None and (last["uright"] >= 0).any() and len(res[k][0]["kps"]):
                frame = self.iv.DeviceFrame.from_frontend(self.fe, k, 0, bounds)
                assign, nm = frame.SearchByProjection(self.track_queries(last))
                res[k][0]["tracked"] = (assign, nm)
            else:
                res[k][0]["tracked"] = (np.full(len(res[k][0]["kps"]), -1, np.int32), 0)
        self.prev_left = res[-1][0]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("sequence", nargs="?"); ap.add_argument("settings", nargs="?")
    ap.add_argument("--rectify", action="store_true"); ap.add_argument("--undistort", action="store_true")
    ap.add_argument("--qual", help="directory of predicted cost images (%%06d.*): enables the introspection-weighted extractor")
    ap.add_argument("--fcn", help="weights blob (tools/export_fcn_weights.py) or 'seeded': run the introspection network on every left image")
    ap.add_argument("--batch", type=int, default=16); ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--track", action="store_true", help="also replay the tracker's cross-frame matcher call (zero-motion prior) on resident frames")
    ap.add_argument("--make-synthetic"); ap.add_argument("--frames", type=int, default=12)
    a = ap.parse_args()
    from iv_slam_amd import kitti
    if a.make_synthetic:
        print("wrote", make_synthetic(a.make_synthetic, a.frames))
        return 0
    if not a.sequence or not a.settings:
        ap.error("SEQUENCE_DIR and SETTINGS.yaml are required")
    S = kitti.Settings.load(a.settings)
    left, right, ts = kitti.LoadImages(a.sequence)
    n = len(ts) if not a.max_frames else min(len(ts), a.max_frames)
    qual = None
    if a.qual:
        qual, found = kitti.GetImageQualFileNames(a.qual, len(ts))
        print("%d predicted cost images found for %d frames" % (found, len(ts)))
    batch = 1 if a.qual else a.batch          # frames without a cost image run the plain extractor (System ignores empty images)
    blob = None
    if a.fcn:
        from iv_slam_amd import fcn_weights
        blob = fcn_weights.pack_blob(fcn_weights.make_seeded_weights(7)) if a.fcn == "seeded" else np.fromfile(a.fcn, np.float32)
    rp = Replay(S, a.rectify, a.undistort, introspect=bool(a.qual), batch=batch, fcn_blob=blob, track=a.track)
    t_dev = 0.0; done = 0
    for i0 in range(0, n, batch):
        idx = [i for i in range(i0, min(i0 + batch, n)) if left[i]]
        if not idx:
            continue
        raw = [kitti.imread(left[i]) for i in idx]
        Ls = [kitti.to_gray(im, rp.rgb) for im in raw]
        Rs = [kitti.to_gray(kitti.imread(right[i]), rp.rgb) for i in idx]
        Cs = None
        if qual is not None:
            Cs = [kitti.to_gray(kitti.imread(qual[i]), rp.rgb) if qual[i] else None for i in idx]
        t0 = time.perf_counter()
        res = rp.run(Ls, Rs, Cs, raw)
        t_dev += time.perf_counter() - t0; done += len(idx)
        for i, (l, r) in zip(idx, res):
            m = l["uright"] >= 0
            med = float(np.median(l["depth"][m])) if m.any() else float("nan")
            print("frame %6d t=%.3f  kps L/R %4d/%4d  stereo matches %4d  median depth %.2f%s" %
                  (i, ts[i], len(l["kps"]), len(r["kps"]), int(m.sum()), med,
                   "  tracked from previous frame %4d" % l["tracked"][1] if "tracked" in l else ""))
    if done:
        print("%d pairs, device part (upload + remap + extract + match + fetch) %.1f pairs/s" % (done, done / t_dev))
    return 0


if __name__ == "__main__":
    sys.exit(main())
