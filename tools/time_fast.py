"""Timing experiment helper: average k_fast_nms launch time (HIP events) for the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iv_slam_amd as iv
from bench import make_device_stream, W, H
dev = torch.device("cuda:0")
left, right = make_device_stream(torch, dev, 64, seed=100)
fe = iv.StereoFrontend(W, H, 64)
for i in range(6):
    fe.run(left, right)
    fe.sync()                      # no overlap between batches: isolate the kernel
s, n = fe.fast_ms_stats(4)
print("k_fast_nms avg %.1f us" % (1e3 * s / n))
