"""state_dict (.pth of IntrospectionModule, or encoder+decoder checkpoints) -> flat f32 blob for ivf_fcn_create.

    python tools/export_fcn_weights.py model_state.pth weights.bin

The blob is the state_dict's f32 tensors in state_dict order with `num_batches_tracked` skipped
(iv_slam_amd/fcn_weights.py:tensor_specs is the authoritative walk).  Only torch is needed, not the reference.
"""
import sys

import numpy as np


def state_dict_to_numpy(sd):
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in sd.items() if not k.endswith("num_batches_tracked")}


def main(src, dst):
    import torch
    sys.path.insert(0, __file__.rsplit("/", 2)[0])
    from iv_slam_amd import fcn_weights
    sd = torch.load(src, map_location="cpu")
    if "state_dict" in sd:
        sd = sd["state_dict"]
    blob = fcn_weights.pack_blob(state_dict_to_numpy(sd))
    blob.tofile(dst)
    print("wrote %s: %d floats" % (dst, blob.size))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
