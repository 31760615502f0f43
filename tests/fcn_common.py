"""Shared helpers of the FCN tests: rebuild the seeded weights / inputs a golden fixture was made with."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLD = os.path.join(ROOT, "tests", "golden")


def bgr_image(w, h, seed):
    from iv_slam_amd import synth
    return np.stack([synth.make_left(w, h, seed=seed, idx=c) for c in range(3)], axis=-1)


def load_case(tag):
    from iv_slam_amd import fcn_weights
    g = np.load(os.path.join(GOLD, "fcn_%s.npz" % tag))
    seed = int(g["seed"][0]); w, h = (int(v) for v in g["size"])
    W = fcn_weights.make_seeded_weights(seed)
    if "bn_stats" in g.files:        # "smallw" cases: trained-checkpoint magnitudes (tests/golden/make_fcn_golden.py)
        pw, dw = np.float32(g["pw_scale"][0]), np.float32(g["dw_scale"][0])
        for k in W:
            if k.endswith(".weight") and W[k].ndim == 4 and "conv_last" not in k:
                depthwise = W[k].shape[1] == 1 and W[k].shape[2] == 3
                W[k] = (W[k] * (dw if depthwise else pw)).astype(np.float32)
        st = g["bn_stats"]; o = 0
        for name, shape in fcn_weights.tensor_specs():
            if (name.endswith("running_mean") or name.endswith("running_var")) and "deepsup" not in name:
                n = int(np.prod(shape)); W[name] = st[o:o + n].astype(np.float32).copy(); o += n
        assert o == st.size
    W["decoder.conv_last.weight"] = g["conv_last_weight"]
    W["decoder.conv_last.bias"] = g["conv_last_bias"]
    return g, W, bgr_image(w, h, 50 + seed), (h, w)


def check_against_golden(g, cost, u8, tol=1e-3):
    """cost maps within `tol` of the reference CPU path (north-star bar 1e-3); u8 = trunc(cost*255) may differ by one
    LSB from the reference only where the reference's value sits within tol*255 of a truncation boundary."""
    sub = int(g["sub"][0]) if "sub" in g.files else 6
    ref = g["cost_sub"]; got = cost[::sub, ::sub]
    err = float(np.abs(got - ref).max())
    assert err < tol, "cost map differs from the reference by %.3g" % err
    ru8 = g["u8_sub"].astype(np.int32); gu8 = u8[::sub, ::sub].astype(np.int32)
    d = np.abs(ru8 - gu8)
    assert d.max() <= 1
    frac = (ref.astype(np.float64) * 255.0) % 1.0
    near = np.minimum(frac, 1.0 - frac) < tol * 255.0
    assert (d[~near] == 0).all()
    return err
