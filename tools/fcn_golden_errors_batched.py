"""max |cost - reference golden| of the HIP FCN through the BATCHED device path (batch 16: the whole-block kernels without the
small-batch split) for every committed golden case; IVF_FCN_FP6=0/1 with the experiment build selects the expansion's form."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fcn_common as FC
import iv_slam_amd as iv
from iv_slam_amd import fcn_weights
dev = torch.device("cuda:0")
NB = int(os.environ.get("IVF_B", "16"))
worst = 0.0
for tag in ("kitti", "jackal", "jackal_full", "kitti_smallw", "jackal_smallw", "kitti_bigw"):
    g, W, bgr, out_size = FC.load_case(tag)
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out_size, max_batch=NB)
    batch = torch.from_numpy(np.stack([bgr] * NB)).to(dev)
    cf = torch.empty((NB,) + tuple(out_size), dtype=torch.float32, device=dev)
    fcn.forward_device(batch, cost_f32=cf); fcn.status()
    cost = cf[NB - 1].cpu().numpy()
    sub = int(g["sub"][0]) if "sub" in g.files else 6
    err = np.abs(cost[::sub, ::sub] - g["cost_sub"])
    worst = max(worst, float(err.max()))
    print("%-14s batch %d  max|d| %.3g  mean|d| %.3g  finite %s" % (tag, NB, err.max(), err.mean(), np.isfinite(cost).all()), flush=True)
print("FP6=%s worst %.3g" % (os.environ.get("IVF_FCN_FP6", "default"), worst))
