"""Batch determinism of the FCN on the device: three forwards of the same IVF_B-image batch must be identical, and image 0
must equal the single-image forward (the test that exposed co-residency defects in earlier whole-block kernels)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import fcn_common as FC, iv_slam_amd as iv
from iv_slam_amd import fcn_weights
g, W, bgr, out = FC.load_case("kitti")
B = int(os.environ.get("IVF_B", "3"))
f = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out, max_batch=B)
u8, cost = f(bgr, want_f32=True)
dev = torch.device("cuda:0")
batch = torch.from_numpy(np.stack([bgr if i % 2 == 0 else bgr[:, ::-1].copy() for i in range(B)])).to(dev)
outs = []
for rep in range(3):
    c = torch.empty((B,) + tuple(out), dtype=torch.uint8, device=dev)
    f.forward_device(batch, cost_u8=c); torch.cuda.synchronize()
    outs.append(c.cpu().numpy())
d = [int((outs[0][i] != outs[1][i]).sum() + (outs[1][i] != outs[2][i]).sum()) for i in range(B)]
print("B=%d: pixels differing between repeats, summed over images: %d (max per image %d); image 0 vs single-image forward: %d"
      % (B, sum(d), max(d), int((outs[0][0] != u8).sum())))
sys.exit(1 if sum(d) or (outs[0][0] != u8).any() else 0)
