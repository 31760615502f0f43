"""Introspection-function weights, in any of the forms the reference ships or produces -> flat f32 blob for ivf_fcn_create.

    python tools/export_fcn_weights.py iv_jackal_mobilenet_c1deepsup_light.pt weights.bin            # TorchScript archive
    python tools/export_fcn_weights.py model_state.pth weights.bin                                  # IntrospectionModule state_dict
    python tools/export_fcn_weights.py --encoder encoder_epoch_N.pth --decoder decoder_epoch_N.pth weights.bin

* TorchScript archive: what the C++ front end loads (`torch::jit::load`, ORB/Examples/Stereo/stereo_kitti.cc:236; fetched by
  introspective_ORB_SLAM/download_pretrained_models.bash), produced by IF/training/export_model_light.py:114-121
  (`torch.jit.trace(net, img)` + `save`).  Read with `torch.jit.load(...).state_dict()`.
* state_dict of the whole IntrospectionModule (keys `encoder.features...`, `decoder.cbr...`), bare or under "state_dict".
* separate encoder / decoder checkpoints as training writes them and ModelBuilder loads them
  (IF/networks/models_light/models_light.py:57-60, 81-84): keys without the `encoder.` / `decoder.` prefix.

The blob is the f32 tensors in state_dict order with `num_batches_tracked` skipped; iv_slam_amd/fcn_weights.py:tensor_specs is
the authoritative walk.  The key walk is VERIFIED against it: a missing tensor, an unexpected tensor or a shape mismatch is an
error, never a silent default (the reference itself loads with strict=False and would run on random weights).  Only torch is
needed, not the reference's Python.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class WeightFormatError(ValueError):
    pass


def _to_numpy(sd, prefix=""):
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        out[prefix + k] = v.detach().cpu().numpy().astype(np.float32)
    return out


def load_any(path):
    """{name: f32 array} from a TorchScript archive or a pickled state_dict (optionally under "state_dict")."""
    import torch
    try:
        m = torch.jit.load(path, map_location="cpu")            # TorchScript archive (zip with code/ and data/)
        return _to_numpy(m.state_dict())
    except RuntimeError:
        pass
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    if hasattr(sd, "state_dict"):
        sd = sd.state_dict()
    return _to_numpy(sd)


def with_prefix(weights, prefix):
    """encoder / decoder checkpoints carry bare keys (`features.0.0.weight`, `cbr.0.weight`); a `module.` prefix from
    DataParallel is dropped."""
    out = {}
    for k, v in weights.items():
        if k.startswith("module."):
            k = k[len("module."):]
        out[k if k.startswith(prefix) else prefix + k] = v
    return out


def verify(weights):
    """every tensor of the fixed architecture present with its shape, nothing else: returns the dict restricted and ordered
    like fcn_weights.tensor_specs(); raises WeightFormatError otherwise."""
    from iv_slam_amd import fcn_weights
    specs = fcn_weights.tensor_specs()
    names = [n for n, _ in specs]
    missing = [n for n in names if n not in weights]
    extra = [k for k in weights if k not in set(names)]
    wrong = [(n, tuple(weights[n].shape), tuple(s)) for n, s in specs if n in weights and tuple(weights[n].shape) != tuple(s)]
    if missing or extra or wrong:
        msg = []
        if missing:
            msg.append("%d tensors missing (first: %s)" % (len(missing), ", ".join(missing[:4])))
        if extra:
            msg.append("%d unexpected tensors (first: %s)" % (len(extra), ", ".join(extra[:4])))
        if wrong:
            msg.append("%d shape mismatches (first: %s is %r, expected %r)" % ((len(wrong),) + wrong[0]))
        raise WeightFormatError("not the mobilenetv2dilated + c1_deepsup architecture ivf_fcn_create implements: " + "; ".join(msg))
    return {n: weights[n] for n in names}


def export(dst, model=None, encoder=None, decoder=None):
    from iv_slam_amd import fcn_weights
    if model is not None:
        if encoder or decoder:
            raise WeightFormatError("give either one model file or --encoder and --decoder")
        w = load_any(model)
        if not any(k.startswith("encoder.") for k in w) and any(k.startswith("module.") for k in w):
            w = {k[len("module."):]: v for k, v in w.items()}
    else:
        if not (encoder and decoder):
            raise WeightFormatError("--encoder and --decoder must be given together")
        w = dict(with_prefix(load_any(encoder), "encoder."))
        w.update(with_prefix(load_any(decoder), "decoder."))
    blob = fcn_weights.pack_blob(verify(w))
    assert blob.size == fcn_weights.blob_floats()
    blob.tofile(dst)
    return blob


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("model", nargs="?", help="TorchScript archive (.pt) or IntrospectionModule state_dict (.pth)")
    ap.add_argument("output")
    ap.add_argument("--encoder"); ap.add_argument("--decoder")
    a = ap.parse_args(argv)
    try:
        blob = export(a.output, a.model, a.encoder, a.decoder)
    except WeightFormatError as e:
        raise SystemExit("export_fcn_weights: " + str(e))
    print("wrote %s: %d floats" % (a.output, blob.size))


if __name__ == "__main__":
    main()
