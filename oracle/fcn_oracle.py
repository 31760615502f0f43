"""fcn_oracle.py -- CPU ORACLE of the introspection FCN forward pass (TEST INFRASTRUCTURE ONLY).

numpy f32 restatement of
  * the C++ call contract  ORB/Examples/Stereo/stereo_kitti.cc:493-514  (BGR->RGB, /255, -mean, /std, HWC->NCHW,
    forward, (y*255).to(u8) truncation)
  * IntrospectionModule.forward   IF/networks/models_light/models_light.py:18-28  (bilinear to 512x512,
    encoder, decoder, sigmoid(20*(pred-0.5)))
  * MobileNetV2Dilated / InvertedResidual   models_light.py:99-172, mobilenet.py:35-64  (eval-mode BN eps=1e-5, ReLU6)
  * C1DeepSup.forward   models_light.py:192-204  (cbr 3x3 + BN + ReLU, conv_last 1x1 + bias, bilinear to out_size)
Parity: PINNED -- tests/golden/fcn_*.npz hold outputs of the reference's own Python model (imported from
/root/reference in the build container by tests/golden/make_fcn_golden.py) for the same seeded weights/inputs;
tests/test_fcn_oracle.py checks this restatement against them (tolerance 2e-4 on the f32 map; the bar is 1e-3).
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)
F = np.float32


def preprocess(bgr_u8):
    """HxWx3 u8 (BGR, as cv::imread gives) -> 1x3xHxW f32, RGB, normalised (stereo_kitti.cc:494-506)."""
    rgb = bgr_u8[:, :, ::-1].astype(np.float32) * F(1.0 / 255.0)
    rgb = (rgb - MEAN) / STD
    return np.ascontiguousarray(rgb.transpose(2, 0, 1)[None]).astype(np.float32)


def bilinear(x, oh, ow):
    """torch.nn.functional.interpolate(mode='bilinear', align_corners=False) on NCHW f32."""
    n, c, ih, iw = x.shape

    def axis(isz, osz):
        scale = F(isz) / F(osz)
        src = scale * (np.arange(osz, dtype=np.float32) + F(0.5)) - F(0.5)
        src = np.maximum(src, F(0)).astype(np.float32)
        i0 = np.minimum(src.astype(np.int64), isz - 1)
        i1 = i0 + (i0 < isz - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, (F(1) - l1).astype(np.float32), l1

    y0, y1, hy0, hy1 = axis(ih, oh)
    x0, x1, wx0, wx1 = axis(iw, ow)
    top = x[:, :, y0][:, :, :, x0] * wx0 + x[:, :, y0][:, :, :, x1] * wx1
    bot = x[:, :, y1][:, :, :, x0] * wx0 + x[:, :, y1][:, :, :, x1] * wx1
    return (top * hy0[None, None, :, None] + bot * hy1[None, None, :, None]).astype(np.float32)


def conv2d(x, w, stride=1, pad=0, dil=1, groups=1):
    n, cin, ih, iw = x.shape
    cout, cing, kh, kw = w.shape
    oh = (ih + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    ow = (iw + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad))) if pad else x
    if groups == cin and cing == 1:                       # depthwise
        out = np.zeros((n, cout, oh, ow), np.float32)
        for ky in range(kh):
            for kx in range(kw):
                sl = xp[:, :, ky * dil:ky * dil + (oh - 1) * stride + 1:stride, kx * dil:kx * dil + (ow - 1) * stride + 1:stride]
                out += sl * w[None, :, 0, ky, kx, None, None]
        return out
    assert groups == 1
    if kh == 1 and kw == 1 and stride == 1:
        return np.einsum("oc,nchw->nohw", w[:, :, 0, 0], x, optimize=True).astype(np.float32)
    cols = np.empty((n, cin, kh, kw, oh, ow), np.float32)
    for ky in range(kh):
        for kx in range(kw):
            cols[:, :, ky, kx] = xp[:, :, ky * dil:ky * dil + (oh - 1) * stride + 1:stride, kx * dil:kx * dil + (ow - 1) * stride + 1:stride]
    out = w.reshape(cout, -1) @ cols.reshape(n, cin * kh * kw, oh * ow)
    return out.reshape(n, cout, oh, ow).astype(np.float32)


def bn(x, W, prefix, eps=1e-5):
    g, b, m, v = (W[prefix + s] for s in (".weight", ".bias", ".running_mean", ".running_var"))
    inv = (F(1) / np.sqrt(v + F(eps))).astype(np.float32)
    return ((x - m[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]).astype(np.float32)


def relu6(x):
    return np.clip(x, F(0), F(6))


def forward(W, bgr_u8, out_size, enc_size=(512, 512), return_taps=False):
    """W: {state_dict name: f32 array}.  Returns (cost_f32 HxW, cost_u8 HxW[, taps])."""
    from iv_slam_amd.fcn_weights import BLOCKS          # architecture table (data)
    taps = {}
    x = bilinear(preprocess(bgr_u8), *enc_size)
    taps["resized"] = x
    x = relu6(bn(conv2d(x, W["encoder.features.0.0.weight"], 2, 1), W, "encoder.features.0.1"))
    taps["f0"] = x
    for i, (inp, oup, t, s, d, res) in enumerate(BLOCKS, start=1):
        p = "encoder.features.%d.conv" % i
        y = x
        if t == 1:
            y = relu6(bn(conv2d(y, W[p + ".0.weight"], s, d, d, groups=inp * t), W, p + ".1"))
            y = bn(conv2d(y, W[p + ".3.weight"]), W, p + ".4")
        else:
            y = relu6(bn(conv2d(y, W[p + ".0.weight"]), W, p + ".1"))
            y = relu6(bn(conv2d(y, W[p + ".3.weight"], s, d, d, groups=inp * t), W, p + ".4"))
            y = bn(conv2d(y, W[p + ".6.weight"]), W, p + ".7")
        x = (x + y).astype(np.float32) if res else y
        taps.setdefault("block_absmax", {})[i] = float(np.abs(x).max())     # the un-clamped activations (linear bottleneck + residual)
        if i in (7, 17):
            taps["f%d" % i] = x
    y = np.maximum(bn(conv2d(x, W["decoder.cbr.0.weight"], 1, 1), W, "decoder.cbr.1"), F(0))
    y = conv2d(y, W["decoder.conv_last.weight"]) + W["decoder.conv_last.bias"][None, :, None, None]
    taps["logits"] = y.astype(np.float32)
    y = bilinear(y.astype(np.float32), *out_size)
    z = (F(20) * (y - F(0.5))).astype(np.float32)
    cost = (F(1) / (F(1) + np.exp(-z))).astype(np.float32)[0, 0]
    u8 = (cost * F(255.0)).astype(np.float32).astype(np.uint8)      # (cost_img * 255.0).to(torch::kByte): truncation
    return (cost, u8, taps) if return_taps else (cost, u8)
