"""tools/export_fcn_weights.py on the artefact formats the reference ships / produces: a TorchScript archive traced and saved the way
IF/training/export_model_light.py:114-121 does (what stereo_kitti.cc:236 `torch::jit::load`s), a whole-module state_dict, and
separate encoder / decoder checkpoints (models_light.py:57-60, 81-84).  The reference model is IMPORTED here (build container
only; nothing of it is committed) -- the test is skipped where /root/reference is absent."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/introspection_function"
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's Python model is only present in the build container")


@pytest.fixture(scope="module")
def model_and_weights():
    import torch
    sys.path.insert(0, REF)
    from networks.models_light import models_light as ML, mobilenet
    from iv_slam_amd import fcn_weights
    W = fcn_weights.make_seeded_weights(11)
    enc = ML.MobileNetV2Dilated(mobilenet.mobilenetv2(pretrained=False), 8)
    dec = ML.C1DeepSup(num_class=1, fc_dim=320, regression_mode=True, inference_mode=True, out_size=(96, 128))
    m = ML.IntrospectionModule(enc, dec, (512, 512), logistic_func=True).eval()
    sd = m.state_dict()
    for k in sd:
        if not k.endswith("num_batches_tracked"):
            sd[k] = torch.from_numpy(W[k].copy())
    m.load_state_dict(sd)
    return m, W


def test_torchscript_archive_state_dict_and_split_checkpoints(model_and_weights, tmp_path):
    import torch
    import export_fcn_weights as X
    from iv_slam_amd import fcn_weights
    m, W = model_and_weights
    want = fcn_weights.pack_blob(W)
    # (1) TorchScript archive, traced and saved as export_model_light.py does
    torch.manual_seed(0)
    with torch.no_grad():
        sm = torch.jit.trace(m, torch.rand(1, 3, 96, 128))
    pt = str(tmp_path / "iv_model_light.pt")
    sm.save(pt)
    out = str(tmp_path / "w1.bin")
    X.main([pt, out])
    assert np.array_equal(np.fromfile(out, np.float32), want)
    # (2) whole-module state_dict, bare and wrapped
    pth = str(tmp_path / "model.pth")
    torch.save(m.state_dict(), pth)
    assert np.array_equal(X.export(str(tmp_path / "w2.bin"), model=pth), want)
    torch.save({"state_dict": m.state_dict(), "epoch": 3}, pth)
    assert np.array_equal(X.export(str(tmp_path / "w3.bin"), model=pth), want)
    # (3) separate encoder / decoder checkpoints (bare keys), one of them from DataParallel ("module." prefix)
    e = str(tmp_path / "encoder_epoch_1.pth"); d = str(tmp_path / "decoder_epoch_1.pth")
    torch.save(m.encoder.state_dict(), e)
    torch.save({"module." + k: v for k, v in m.decoder.state_dict().items()}, d)
    X.main(["--encoder", e, "--decoder", d, str(tmp_path / "w4.bin")])
    assert np.array_equal(np.fromfile(str(tmp_path / "w4.bin"), np.float32), want)


def test_mismatching_checkpoints_fail_loudly(model_and_weights, tmp_path):
    import torch
    import export_fcn_weights as X
    m, _ = model_and_weights
    sd = dict(m.state_dict())
    del sd["encoder.features.7.conv.3.weight"]
    p = str(tmp_path / "short.pth"); torch.save(sd, p)
    with pytest.raises(X.WeightFormatError, match="missing"):
        X.export(str(tmp_path / "x.bin"), model=p)
    sd = dict(m.state_dict()); sd["decoder.cbr.0.weight"] = sd["decoder.cbr.0.weight"][:, :100]
    torch.save(sd, p)
    with pytest.raises(X.WeightFormatError, match="shape"):
        X.export(str(tmp_path / "x.bin"), model=p)
    sd = dict(m.state_dict()); sd["decoder.ppm.0.weight"] = torch.zeros(3)
    torch.save(sd, p)
    with pytest.raises(X.WeightFormatError, match="unexpected"):
        X.export(str(tmp_path / "x.bin"), model=p)
    # encoder and decoder paths swapped (the reference's own Jackal config does this and loads with strict=False: SURVEY D-15)
    e = str(tmp_path / "e.pth"); d = str(tmp_path / "d.pth")
    torch.save(m.encoder.state_dict(), e); torch.save(m.decoder.state_dict(), d)
    with pytest.raises(X.WeightFormatError):
        X.export(str(tmp_path / "x.bin"), encoder=d, decoder=e)
    with pytest.raises(SystemExit):
        X.main(["--encoder", e, str(tmp_path / "x.bin")])
