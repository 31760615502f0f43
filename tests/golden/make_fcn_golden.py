"""Generates tests/golden/fcn_*.npz by IMPORTING THE REFERENCE'S OWN PYTHON MODEL (IF/networks/models_light) in the
build container.  The reference cannot travel to the GPU box, so only small vectors are committed: seeds (weights
and inputs are regenerated from iv_slam_amd.fcn_weights / iv_slam_amd.synth), the calibrated conv_last, and the
reference's outputs (sub-sampled f32 map, full 64x64 logits, tap statistics, u8 sub-sample).

    python tests/golden/make_fcn_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference/introspection_function")
from networks.models_light import models_light as ML, mobilenet  # noqa: E402  (the reference, read-only)
from iv_slam_amd import fcn_weights, synth  # noqa: E402
import fcn_oracle  # noqa: E402


def bgr_image(w, h, seed):
    """seeded colour image: three differently-seeded synthetic planes (B, G, R)."""
    return np.stack([synth.make_left(w, h, seed=seed, idx=c) for c in range(3)], axis=-1)


def reference_model(W, out_size):
    enc = ML.MobileNetV2Dilated(mobilenet.mobilenetv2(pretrained=False), 8)
    dec = ML.C1DeepSup(num_class=1, fc_dim=320, regression_mode=True, inference_mode=True, out_size=out_size)
    m = ML.IntrospectionModule(enc, dec, (512, 512), logistic_func=True).eval()
    sd = m.state_dict()
    for k in sd:
        if not k.endswith("num_batches_tracked"):
            sd[k] = torch.from_numpy(W[k].copy())
    m.load_state_dict(sd)
    return m


def run_reference(m, bgr):
    x = torch.from_numpy(fcn_oracle.preprocess(bgr))          # the C++ call contract's preprocessing
    taps = {}
    with torch.no_grad():
        xr = torch.nn.functional.interpolate(x, size=(512, 512), mode="bilinear", align_corners=False)
        f = xr
        for i, mod in enumerate(m.encoder.features):
            f = mod(f)
            if i in (0, 7, 17):
                taps["f%d" % i] = f.numpy()
        logits = m.decoder.conv_last(m.decoder.cbr(f))
        y = m(x)                                               # the deployed graph end to end
    u8 = (y * 255.0).to(torch.uint8)                           # stereo_kitti.cc:511
    return y[0, 0].numpy(), u8[0, 0].numpy(), logits[0, 0].numpy(), taps


def main():
    torch.set_num_threads(8)
    # (tag, image size, seed, sub-sampling step of the committed maps); `python make_fcn_golden.py TAG...` regenerates only those
    # "smallw" cases: the magnitudes of a TRAINED checkpoint instead of He-normal init -- pointwise weights ~1e-3, depthwise
    # weights ~1e-2, and every BatchNorm's running statistics set to the actual statistics of its input (one train-mode pass
    # of the reference model with momentum 1), which puts running_var anywhere between ~1e-8 and ~1e2 and the BN scales
    # between ~1e-1 and ~1e4.  The calibrated statistics are data produced by the reference, so they are committed.
    cases = (("kitti", (1242, 375), 3, 6, None), ("jackal", (960, 600), 4, 6, None), ("jackal_full", (1920, 1200), 6, 12, None),
             ("kitti_smallw", (1242, 375), 8, 6, (1e-2, 5e-2)), ("jackal_smallw", (960, 600), 9, 6, (3e-3, 2e-2)),
             ("kitti_bigw", (1242, 375), 10, 6, (8.0, 3.0)))
    only = set(sys.argv[1:])
    for tag, (w, h), seed, sub, small in cases:
        if only and tag not in only:
            continue
        W = fcn_weights.make_seeded_weights(seed)
        bgr = bgr_image(w, h, 50 + seed)
        extra = {}
        if small is not None:
            pw_scale, dw_scale = small
            for k in W:
                if k.endswith(".weight") and W[k].ndim == 4 and "conv_last" not in k:
                    depthwise = W[k].shape[1] == 1 and W[k].shape[2] == 3
                    W[k] = (W[k] * np.float32(dw_scale if depthwise else pw_scale)).astype(np.float32)
            m = reference_model(W, (h, w))
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.momentum = 1.0
            m.train()
            with torch.no_grad():
                m.encoder.train(); m.decoder.train()
                x = torch.from_numpy(fcn_oracle.preprocess(bgr))
                f = torch.nn.functional.interpolate(x, size=(512, 512), mode="bilinear", align_corners=False)
                f = m.encoder.features(f)
                m.decoder.cbr(f)
            m.eval()
            sd = m.state_dict()
            stats = []
            for name, _shape in fcn_weights.tensor_specs():
                if name.endswith("running_mean") or name.endswith("running_var"):
                    if "deepsup" in name:
                        continue
                    W[name] = sd[name].numpy().astype(np.float32).copy()
                    stats.append(W[name].reshape(-1))
            extra = dict(bn_stats=np.concatenate(stats), pw_scale=np.array([pw_scale], np.float32), dw_scale=np.array([dw_scale], np.float32))
            rv = np.concatenate([W[k].reshape(-1) for k in W if k.endswith("running_var") and "deepsup" not in k])
            print(tag, "running_var range %.3g .. %.3g, median %.3g" % (rv.min(), rv.max(), np.median(rv)))
        # calibrate conv_last so the logits straddle 0.5 (otherwise the logistic saturates to 0/1 everywhere)
        m = reference_model(W, (h, w))
        _, _, logits, _ = run_reference(m, bgr)
        pre = (logits - W["decoder.conv_last.bias"][0]) / 0.05           # un-scaled response
        last_w = (W["decoder.conv_last.weight"] / np.float32(0.05) * np.float32(0.06 / max(float(pre.std()), 1e-9))).astype(np.float32)
        W["decoder.conv_last.weight"] = last_w
        m = reference_model(W, (h, w))
        _, _, logits, _ = run_reference(m, bgr)
        W["decoder.conv_last.bias"] = np.array([0.5 - float(logits.mean() - W["decoder.conv_last.bias"][0])], np.float32)
        m = reference_model(W, (h, w))
        y, u8, logits, taps = run_reference(m, bgr)
        print(tag, "logits mean/std", logits.mean(), logits.std(), "cost range", y.min(), y.max(), "u8 hist",
              np.percentile(u8, [1, 25, 50, 75, 99]))
        # the numpy oracle must agree with the reference before anything is frozen
        oc, ou8, otaps = fcn_oracle.forward(W, bgr, (h, w), return_taps=True)
        print("  oracle vs reference: max|d cost| = %.3g, max|d logits| = %.3g, u8 mismatches = %d" %
              (np.abs(oc - y).max(), np.abs(otaps["logits"][0, 0] - logits).max(), int((ou8 != u8).sum())))
        assert np.abs(oc - y).max() < 2e-4      # f32 summation-order noise x logistic slope 5; bar is 1e-3
        np.savez_compressed(os.path.join(HERE, "fcn_%s.npz" % tag), seed=np.array([seed]), size=np.array([w, h]),
                            conv_last_weight=W["decoder.conv_last.weight"], conv_last_bias=W["decoder.conv_last.bias"],
                            cost_sub=y[::sub, ::sub].copy(), u8_sub=u8[::sub, ::sub].copy(), sub=np.array([sub]), logits=logits.copy(),
                            tap_stats=np.array([[taps[k].mean(), np.abs(taps[k]).mean(), taps[k].std()] for k in ("f0", "f7", "f17")], np.float64),
                            f17_sub=taps["f17"][0, ::16, ::8, ::8].copy(), **extra)
    for f in sorted(os.listdir(HERE)):
        if f.startswith("fcn_"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
