"""Experiment (host only): how much cost-map error does storing the HIDDEN tensors (post-expansion, post-ReLU6) of the inverted
residual blocks as f16 instead of f32 add, against the reference goldens?  Everything else stays f32 (torch CPU).
    python tools/fcn_hidden_precision.py"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fcn_common
import fcn_oracle_torch as O
from iv_slam_amd.fcn_weights import BLOCKS


@torch.no_grad()
def forward(T, bgr, out_size, first_q, mode):
    a = np.asarray(bgr)[None]
    x = torch.from_numpy(np.ascontiguousarray(a[..., ::-1])).to(torch.float32).permute(0, 3, 1, 2) * (1.0 / 255.0)
    x = (x - torch.tensor(O.MEAN).view(1, 3, 1, 1)) / torch.tensor(O.STD).view(1, 3, 1, 1)
    x = F.interpolate(x, size=(512, 512), mode="bilinear", align_corners=False)
    x = F.relu6(O._bn(F.conv2d(x, T["encoder.features.0.0.weight"], None, 2, 1), T, "encoder.features.0.1"))

    def q(v):
        if mode == "f16":
            return v.half().float()
        if mode == "bf16":
            return v.bfloat16().float()
        if mode == "u16":                      # 16-bit fixed point on [0, 6]
            return torch.round(v * (65535.0 / 6.0)) * (6.0 / 65535.0)
        return v
    for i, (inp, oup, t, s, d, res) in enumerate(BLOCKS, start=1):
        p = "encoder.features.%d.conv" % i
        y = x
        if t == 1:
            y = F.relu6(O._bn(F.conv2d(y, T[p + ".0.weight"], None, s, d, d, inp * t), T, p + ".1"))
            y = O._bn(F.conv2d(y, T[p + ".3.weight"]), T, p + ".4")
        else:
            y = F.relu6(O._bn(F.conv2d(y, T[p + ".0.weight"]), T, p + ".1"))
            if i >= first_q:
                y = q(y)
            y = F.relu6(O._bn(F.conv2d(y, T[p + ".3.weight"], None, s, d, d, inp * t), T, p + ".4"))
            y = O._bn(F.conv2d(y, T[p + ".6.weight"]), T, p + ".7")
        x = x + y if res else y
    y = F.relu(O._bn(F.conv2d(x, T["decoder.cbr.0.weight"], None, 1, 1), T, "decoder.cbr.1"))
    y = F.conv2d(y, T["decoder.conv_last.weight"], T["decoder.conv_last.bias"])
    y = F.interpolate(y, size=tuple(out_size), mode="bilinear", align_corners=False)
    return torch.sigmoid(20.0 * (y - 0.5))[0, 0].numpy()


torch.set_num_threads(8)
for tag in ("kitti", "jackal", "kitti_smallw", "jackal_smallw", "kitti_bigw"):
    g, W, bgr, (h, w) = fcn_common.load_case(tag)
    T = O.prepare(W)
    sub = int(g["sub"][0]) if "sub" in g.files else 6
    ref = g["cost_sub"]
    out = []
    for first_q, mode in ((99, "f32"), (5, "f16"), (15, "f16"), (5, "u16"), (5, "bf16")):
        c = forward(T, bgr, (h, w), first_q, mode)
        out.append("%s from block %d: %.2e" % (mode, first_q, float(np.abs(c[::sub, ::sub] - ref).max())))
    print("%-14s max |cost - reference|:  %s" % (tag, "   ".join(out)), flush=True)
