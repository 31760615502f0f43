"""GPU parity of the introspection FCN: HIP path (C-ABI) vs golden vectors of the reference's own Python model
(1e-3 bar from BASELINE.json:north_star) and vs the numpy oracle."""
import numpy as np
import pytest

import fcn_common as FC
from iv_slam_amd import fcn_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def iv():
    import iv_slam_amd
    assert iv_slam_amd.load().ivf_device_count() >= 1
    return iv_slam_amd


@pytest.mark.parametrize("tag", ["kitti", "jackal"])
def test_fcn_matches_reference_goldens(iv, tag):
    g, W, bgr, out_size = FC.load_case(tag)
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size)
    u8, cost = fcn(bgr, want_f32=True)
    err = FC.check_against_golden(g, cost, u8, tol=1e-3)
    assert err < 3e-4, "f32-MFMA path should sit well inside the 1e-3 bar (got %.3g)" % err
    assert np.array_equal(u8, (cost * np.float32(255.0)).astype(np.uint8))     # truncation, not rounding
    # idempotent
    u8b = fcn(bgr)
    assert np.array_equal(u8, u8b)


def test_fcn_matches_numpy_oracle_full_map(iv):
    import fcn_oracle
    g, W, bgr, out_size = FC.load_case("kitti")
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size)
    u8, cost = fcn(bgr, want_f32=True)
    oc, ou8 = fcn_oracle.forward(W, bgr, out_size)
    assert np.abs(cost - oc).max() < 3e-4
    assert (np.abs(u8.astype(int) - ou8.astype(int)) <= 1).all() and (u8 != ou8).mean() < 0.01


def test_fcn_batch_device_path_and_extractor_coupling(iv):
    """configs[2]: FCN cost map (device) gating keypoints in the extractor, all resident in HBM."""
    import torch
    import oracle_lib as O
    g, W, bgr, out_size = FC.load_case("kitti")
    dev = torch.device("cuda:0")
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size, max_batch=2)
    batch = torch.from_numpy(np.stack([bgr, bgr[:, ::-1].copy()])).to(dev)
    cost_u8 = torch.empty((2,) + tuple(out_size), dtype=torch.uint8, device=dev)
    fcn.forward_device(batch, cost_u8=cost_u8)
    torch.cuda.synchronize()
    host = cost_u8.cpu().numpy()
    assert np.array_equal(host[0], fcn(bgr))
    assert not np.array_equal(host[0], host[1])
    # feed the device cost maps straight into the batched front end
    grey = np.stack([bgr[..., 1], bgr[:, ::-1, 1]]).copy()
    left = torch.from_numpy(grey).to(dev)
    fe = iv.StereoFrontend(out_size[1], out_size[0], 2, nfeatures=1000, enableIntrospection=True)
    fe.run(left, left.clone(), cost_u8)
    fe.sync()
    for p in range(2):
        ok, od = O.Extractor(1000, 1.2, 8, 20, 7, introspection=True)(grey[p], host[p])
        r = fe.fetch(p, 0)
        assert r["kps"].tobytes() == ok.tobytes() and np.array_equal(r["desc"], od)


def test_fcn_rejects_bad_blob(iv):
    W = fcn_weights.make_seeded_weights(0)
    blob = fcn_weights.pack_blob(W)
    with pytest.raises(iv.IvfError):
        iv.IntrospectionFCN(blob[:-5], (375, 1242))
    with pytest.raises(iv.IvfError):
        iv.IntrospectionFCN(np.concatenate([blob, blob[:3]]), (375, 1242))
