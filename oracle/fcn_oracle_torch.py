"""fcn_oracle_torch.py -- the introspection FCN's layer list executed through torch.nn.functional on HOST cores
(TEST / BASELINE INFRASTRUCTURE ONLY: bench.py's cpu_baseline leg and tests/ may import it, nothing else).

Same graph as oracle/fcn_oracle.py (which restates IF/networks/models_light/models_light.py:18-28, 99-204 and
mobilenet.py:35-64, call contract ORB/Examples/Stereo/stereo_kitti.cc:493-514) but with the convolutions done by
PyTorch's CPU kernels (oneDNN) -- i.e. what the reference's libtorch CPU path runs, minus TorchScript.  It is the CPU
baseline's FCN leg: the numpy oracle is a checker, far slower than anything the reference would execute.
tests/test_fcn_oracle.py checks it against the numpy oracle and the reference goldens.
"""
import numpy as np
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def prepare(W):
    """{state_dict name: np.float32 array} -> the same as torch tensors (done once, outside any timed loop)."""
    return {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for k, v in W.items()}


def _bn(x, T, p):
    return F.batch_norm(x, T[p + ".running_mean"], T[p + ".running_var"], T[p + ".weight"], T[p + ".bias"], False, 0.0, 1e-5)


def _graph(T, x, out_size, enc_size=(512, 512)):
    """the layer list on a normalised NCHW f32 tensor (RGB) -> cost map [N, H, W] f32"""
    from iv_slam_amd.fcn_weights import BLOCKS
    x = F.interpolate(x, size=enc_size, mode="bilinear", align_corners=False)
    x = F.relu6(_bn(F.conv2d(x, T["encoder.features.0.0.weight"], None, 2, 1), T, "encoder.features.0.1"))
    for i, (inp, oup, t, s, d, res) in enumerate(BLOCKS, start=1):
        p = "encoder.features.%d.conv" % i
        y = x
        if t == 1:
            y = F.relu6(_bn(F.conv2d(y, T[p + ".0.weight"], None, s, d, d, inp * t), T, p + ".1"))
            y = _bn(F.conv2d(y, T[p + ".3.weight"]), T, p + ".4")
        else:
            y = F.relu6(_bn(F.conv2d(y, T[p + ".0.weight"]), T, p + ".1"))
            y = F.relu6(_bn(F.conv2d(y, T[p + ".3.weight"], None, s, d, d, inp * t), T, p + ".4"))
            y = _bn(F.conv2d(y, T[p + ".6.weight"]), T, p + ".7")
        x = x + y if res else y
    y = F.relu(_bn(F.conv2d(x, T["decoder.cbr.0.weight"], None, 1, 1), T, "decoder.cbr.1"))
    y = F.conv2d(y, T["decoder.conv_last.weight"], T["decoder.conv_last.bias"])
    y = F.interpolate(y, size=tuple(out_size), mode="bilinear", align_corners=False)
    return torch.sigmoid(20.0 * (y - 0.5))[:, 0]


def _preprocess(bgr_u8):
    a = np.asarray(bgr_u8)
    x = torch.from_numpy(np.ascontiguousarray(a[..., ::-1])).to(torch.float32).permute(0, 3, 1, 2) * (1.0 / 255.0)
    return (x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)


@torch.no_grad()
def forward(T, bgr_u8, out_size, enc_size=(512, 512)):
    """T = prepare(W); bgr_u8 HxWx3 u8 (or a batch NxHxWx3).  Returns (cost f32 [N,]H,W, u8)."""
    a = np.asarray(bgr_u8)
    single = a.ndim == 3
    if single:
        a = a[None]
    cost = _graph(T, _preprocess(a), out_size, enc_size)
    u8 = (cost * 255.0).to(torch.uint8)
    cost, u8 = cost.numpy(), u8.numpy()
    return (cost[0], u8[0]) if single else (cost, u8)


class _Net(torch.nn.Module):
    def __init__(self, T, out_size, enc_size):
        super().__init__()
        self.names = list(T)
        for i, k in enumerate(self.names):
            self.register_buffer("t%d" % i, T[k])
        self.out_size, self.enc_size = tuple(out_size), tuple(enc_size)

    def forward(self, x):
        T = {k: getattr(self, "t%d" % i) for i, k in enumerate(self.names)}
        return _graph(T, x, self.out_size, self.enc_size)


@torch.no_grad()
def frozen(T, example_bgr_u8, out_size, enc_size=(512, 512)):
    """The same layer list as a FROZEN TorchScript module -- torch.jit.trace + torch.jit.freeze, i.e. what the reference's C++
    front end executes after torch::jit::load of the archive IF/training/export_model_light.py:114-121 traced
    (ORB/Examples/Stereo/stereo_kitti.cc:236, :508): constants folded, BatchNorm folded into the convolutions by the freezing
    passes.  Returns run(bgr_u8 batch) -> (cost f32, u8).  The traced graph is specialised to the example's batch shape."""
    net = _Net(T, out_size, enc_size).eval()
    ex = _preprocess(np.asarray(example_bgr_u8))
    mod = torch.jit.freeze(torch.jit.trace(net, ex, check_trace=False))
    mod(ex); mod(ex)                                        # let the profiling executor specialise before anything is timed

    @torch.no_grad()
    def run(bgr_u8):
        cost = mod(_preprocess(np.asarray(bgr_u8)))
        return cost.numpy(), (cost * 255.0).to(torch.uint8).numpy()
    return run
