"""Weight-blob tooling for the introspection FCN (MobileNetV2-dilated encoder + C1 decoder).

Blob format consumed by ivf_fcn_create: the model's `state_dict` tensors in state_dict order, f32, concatenated,
with the int64 `num_batches_tracked` entries skipped (the loader walks the fixed architecture and knows every
size).  `ARCH` below is that walk; it mirrors IF/networks/models_light/mobilenet.py:66-100 (inverted-residual
settings) and models_light.py:108-152 (output-stride-8 dilation rule) as DATA: (inp, oup, t, stride, dilation, res).
"""
import numpy as np

# features[1..17]: (inp, oup, expand t, depthwise stride, depthwise dilation, residual)
BLOCKS = [
    (32, 16, 1, 1, 1, False),
    (16, 24, 6, 2, 1, False), (24, 24, 6, 1, 1, True),
    (24, 32, 6, 2, 1, False), (32, 32, 6, 1, 1, True), (32, 32, 6, 1, 1, True),
    (32, 64, 6, 1, 1, False),                                   # orig stride 2 -> stride 1, dilation 2//2 = 1
    (64, 64, 6, 1, 2, True), (64, 64, 6, 1, 2, True), (64, 64, 6, 1, 2, True),
    (64, 96, 6, 1, 2, False), (96, 96, 6, 1, 2, True), (96, 96, 6, 1, 2, True),
    (96, 160, 6, 1, 2, False),                                  # orig stride 2 -> stride 1, dilation 4//2 = 2
    (160, 160, 6, 1, 4, True), (160, 160, 6, 1, 4, True),
    (160, 320, 6, 1, 4, False),
]
FC_DIM = 320


def tensor_specs():
    """[(name, shape)] in state_dict order without num_batches_tracked."""
    specs = []

    def bn(prefix, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            specs.append(("%s.%s" % (prefix, s), (c,)))

    specs.append(("encoder.features.0.0.weight", (32, 3, 3, 3))); bn("encoder.features.0.1", 32)
    for i, (inp, oup, t, _s, _d, _r) in enumerate(BLOCKS, start=1):
        hid = inp * t
        p = "encoder.features.%d.conv" % i
        if t == 1:
            specs.append((p + ".0.weight", (hid, 1, 3, 3))); bn(p + ".1", hid)
            specs.append((p + ".3.weight", (oup, hid, 1, 1))); bn(p + ".4", oup)
        else:
            specs.append((p + ".0.weight", (hid, inp, 1, 1))); bn(p + ".1", hid)
            specs.append((p + ".3.weight", (hid, 1, 3, 3))); bn(p + ".4", hid)
            specs.append((p + ".6.weight", (oup, hid, 1, 1))); bn(p + ".7", oup)
    specs.append(("decoder.cbr.0.weight", (FC_DIM // 4, FC_DIM, 3, 3))); bn("decoder.cbr.1", FC_DIM // 4)
    specs.append(("decoder.cbr_deepsup.0.weight", (FC_DIM // 4, FC_DIM // 2, 3, 3))); bn("decoder.cbr_deepsup.1", FC_DIM // 4)
    specs.append(("decoder.conv_last.weight", (1, FC_DIM // 4, 1, 1))); specs.append(("decoder.conv_last.bias", (1,)))
    specs.append(("decoder.conv_last_deepsup.weight", (1, FC_DIM // 4, 1, 1))); specs.append(("decoder.conv_last_deepsup.bias", (1,)))
    return specs


def blob_floats():
    return int(sum(int(np.prod(s)) for _, s in tensor_specs()))


def make_seeded_weights(seed=0, last_scale=None, last_bias=None):
    """Random-init weights of the architecture (no checkpoints exist offline): He-normal convs, BN statistics
    randomised so every BN actually transforms its input, conv_last centred on 0.5 so the logistic
    sigmoid(20(x-0.5)) is exercised instead of saturating.  Returns {name: f32 array}."""
    rng = np.random.Generator(np.random.PCG64([0xFC0000 + int(seed)]))
    out = {}
    for name, shape in tensor_specs():
        if name.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shape)
        elif name.endswith("running_mean"):
            a = rng.normal(0.0, 0.1, shape)
        elif len(shape) == 1 and name.endswith(".weight"):
            a = rng.uniform(0.7, 1.3, shape)
        elif len(shape) == 1 and name.endswith(".bias"):
            a = rng.normal(0.0, 0.1, shape)
        else:
            fan = shape[1] * shape[2] * shape[3]
            a = rng.normal(0.0, np.sqrt(2.0 / fan), shape)
        out[name] = a.astype(np.float32)
    w = out["decoder.conv_last.weight"]
    out["decoder.conv_last.weight"] = (w * np.float32(0.05 if last_scale is None else last_scale)).astype(np.float32)
    out["decoder.conv_last.bias"] = np.array([0.5 if last_bias is None else last_bias], np.float32)
    return out


def pack_blob(weights):
    """{name: array} (e.g. a state_dict converted to numpy) -> flat f32 blob for ivf_fcn_create."""
    parts = []
    for name, shape in tensor_specs():
        a = np.asarray(weights[name], np.float32)
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        parts.append(a.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), np.float32)


def unpack_blob(blob):
    blob = np.asarray(blob, np.float32).reshape(-1)
    assert blob.size == blob_floats(), "blob has %d floats, architecture needs %d" % (blob.size, blob_floats())
    out, o = {}, 0
    for name, shape in tensor_specs():
        n = int(np.prod(shape))
        out[name] = blob[o:o + n].reshape(shape).copy()
        o += n
    return out
