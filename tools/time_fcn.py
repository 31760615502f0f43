"""Timing experiment helper: FCN forward ms per image (batch 32) with the current env overrides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import iv_slam_amd as iv
from iv_slam_amd import fcn_weights
dev = torch.device("cuda:0")
B = int(os.environ.get("IVF_B", "32"))
fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(fcn_weights.make_seeded_weights(7)), (375, 1242), (375, 1242), max_batch=B)
bgr = torch.randint(0, 256, (B, 375, 1242, 3), dtype=torch.uint8, device=dev)
out = torch.empty((B, 375, 1242), dtype=torch.uint8, device=dev)
for _ in range(2): fcn.forward_device(bgr, cost_u8=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): fcn.forward_device(bgr, cost_u8=out)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / 5 * 1e3
print("NT_SMALL=%s NT_BIG=%s USE25=%s : %.3f ms/forward(32) = %.1f us/image, %.1f TFLOP/s" % (
    os.environ.get("IVF_FCN_NT_SMALL", "-"), os.environ.get("IVF_FCN_NT_BIG", "-"), os.environ.get("IVF_FCN_USE25", "-"),
    ms, ms / B * 1e3, 17.229 * B / ms))
