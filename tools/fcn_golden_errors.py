"""max |cost - reference golden| of the HIP FCN for every committed golden case (tests/golden/fcn_*.npz)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fcn_common as FC
import iv_slam_amd as iv
from iv_slam_amd import fcn_weights
for tag in ("kitti", "jackal", "jackal_full", "kitti_smallw", "jackal_smallw", "kitti_bigw"):
    g, W, bgr, out_size = FC.load_case(tag)
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size)
    u8, cost = fcn(bgr, want_f32=True)
    sub = int(g["sub"][0]) if "sub" in g.files else 6
    err = np.abs(cost[::sub, ::sub] - g["cost_sub"])
    print("%-14s max|d| %.3g  mean|d| %.3g  finite %s" % (tag, err.max(), err.mean(), np.isfinite(cost).all()))
