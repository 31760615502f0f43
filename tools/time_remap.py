"""Timing helper: k_remap on a batch of 128 1242x375 images (HIP events through torch on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import iv_slam_amd as iv
W, H, N = 1242, 375, 128
K = [718.856, 0, 607.1928, 0, 718.856, 185.2157, 0, 0, 1]
m1, m2 = iv.initUndistortRectifyMap(K, [-0.2, 0.05, 1e-3, -1e-3, 0.0], [0.99999, -0.003, 0.002, 0.003, 0.99999, -0.001, -0.002, 0.001, 0.99999], K, (W, H))
r = iv.Remap(m1, m2, (H, W))
src = torch.randint(0, 256, (N, H, W), dtype=torch.uint8, device="cuda")
out = torch.empty_like(src)
for _ in range(3):
    r.apply_device(src, out)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    r.apply_device(src, out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
alg = N * W * H * (1 + 1) + W * H * 6          # source read + destination write per image; the 6 B/px map is shared by the batch (L2)
print("k_remap %d x %dx%d: %.1f us per launch, %.0f GB/s algorithmic (%.3f of 8 TB/s)" % (N, W, H, us, alg / us / 1e3, alg / us / 1e3 / 8000))
