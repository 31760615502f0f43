"""The numpy FCN oracle against golden vectors produced by the reference's own Python model
(tests/golden/make_fcn_golden.py), plus weight-blob tooling.  CPU only."""
import numpy as np
import pytest

import fcn_common as FC
from iv_slam_amd import fcn_weights


@pytest.mark.parametrize("tag", ["kitti", "jackal", "jackal_full", "kitti_smallw", "jackal_smallw", "kitti_bigw"])
def test_oracle_matches_reference_goldens(tag):
    import fcn_oracle
    g, W, bgr, out_size = FC.load_case(tag)
    cost, u8, taps = fcn_oracle.forward(W, bgr, out_size, return_taps=True)
    err = FC.check_against_golden(g, cost, u8, tol=2e-4)
    assert np.abs(taps["logits"][0, 0] - g["logits"]).max() < 5e-5
    assert np.abs(taps["f17"][0, ::16, ::8, ::8] - g["f17_sub"]).max() < 5e-4
    stats = np.array([[taps[k].mean(), np.abs(taps[k]).mean(), taps[k].std()] for k in ("f0", "f7", "f17")])
    assert np.allclose(stats, g["tap_stats"], rtol=1e-4, atol=1e-5)
    assert cost.shape == out_size and u8.dtype == np.uint8
    # the logistic is exercised, not saturated
    assert cost.min() < 0.1 and cost.max() > 0.9 and 0.3 < cost.mean() < 0.7
    assert err >= 0


def test_blob_roundtrip_and_size():
    W = fcn_weights.make_seeded_weights(1)
    blob = fcn_weights.pack_blob(W)
    assert blob.dtype == np.float32 and blob.size == fcn_weights.blob_floats()
    back = fcn_weights.unpack_blob(blob)
    assert all(np.array_equal(W[k], back[k]) for k in W)
    # parameters only (no running stats, no unused buffers) = 2,157,794
    nparam = sum(int(np.prod(s)) for n, s in fcn_weights.tensor_specs() if "running_" not in n)
    assert nparam == 2157794
    assert len(fcn_weights.tensor_specs()) == 322 - 53          # state_dict minus the 53 int64 num_batches_tracked scalars


def test_bilinear_and_preprocess_kats():
    import fcn_oracle
    x = np.arange(16, dtype=np.float32).reshape(1, 1, 4, 4)
    assert np.array_equal(fcn_oracle.bilinear(x, 4, 4), x)
    up = fcn_oracle.bilinear(x, 8, 8)
    assert up[0, 0, 0, 0] == 0 and up[0, 0, -1, -1] == 15 and abs(up[0, 0, 0, 1] - 0.25) < 1e-6
    img = np.zeros((2, 2, 3), np.uint8); img[..., 2] = 255          # pure red in BGR
    p = fcn_oracle.preprocess(img)
    assert p.shape == (1, 3, 2, 2)
    assert abs(p[0, 0, 0, 0] - (1 - 0.485) / 0.229) < 1e-5 and abs(p[0, 2, 0, 0] - (0 - 0.406) / 0.225) < 1e-5


def test_torch_baseline_graph_matches_numpy_oracle_and_golden():
    """bench.py's CPU-baseline FCN leg (the layer list through torch.nn.functional) is the same function."""
    import fcn_oracle
    import fcn_oracle_torch
    g, W, bgr, out_size = FC.load_case("kitti")
    T = fcn_oracle_torch.prepare(W)
    cost, u8 = fcn_oracle_torch.forward(T, bgr, out_size)
    FC.check_against_golden(g, cost, u8, tol=2e-4)
    oc, ou8 = fcn_oracle.forward(W, bgr, out_size)
    assert np.abs(cost - oc).max() < 2e-4
    cb, ub = fcn_oracle_torch.forward(T, np.stack([bgr, bgr[::-1].copy()]), out_size)
    assert cb.shape == (2,) + tuple(out_size) and np.abs(cb[0] - cost).max() < 1e-5
