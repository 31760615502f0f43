import sys, os; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, fcn_common as FC, iv_slam_amd as iv
from iv_slam_amd import fcn_weights
g, W, bgr, out = FC.load_case("kitti")
dev = torch.device("cuda:0")
f2 = iv.IntrospectionFCN(fcn_weights.pack_blob(W), bgr.shape[:2], out, max_batch=2)
batch = torch.from_numpy(np.stack([bgr, bgr[:, ::-1].copy()])).to(dev)
outs = []
for rep in range(3):
    c = torch.empty((2,) + tuple(out), dtype=torch.uint8, device=dev)
    cf = torch.empty((2,) + tuple(out), dtype=torch.float32, device=dev)
    f2.forward_device(batch, cost_u8=c, cost_f32=cf); torch.cuda.synchronize()
    outs.append((c.cpu().numpy(), cf.cpu().numpy()))
print("batch rep0 vs rep1 u8 diff px:", (outs[0][0] != outs[1][0]).sum(), " f32 max diff", np.abs(outs[0][1] - outs[1][1]).max())
print("batch rep1 vs rep2 u8 diff px:", (outs[1][0] != outs[2][0]).sum())
u1, c1 = f2(bgr, want_f32=True)
print("single vs batch[0] u8 diff px:", (u1 != outs[0][0][0]).sum(), " f32 max diff", np.abs(c1 - outs[0][1][0]).max())
u1b, c1b = f2(bgr, want_f32=True)
print("single vs single:", (u1 != u1b).sum(), np.abs(c1 - c1b).max())
