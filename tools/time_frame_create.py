#!/usr/bin/env python3
"""us per ivf_frame_create_from_frontend + destroy (pooled arenas; r02: 0.5 ms = five hipMalloc + a stream per frame)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import iv_slam_amd as iv
dev = torch.device("cuda:0")
left, right = bench.make_device_stream(torch, dev, 8, seed=5)
fe = iv.StereoFrontend(bench.W, bench.H, 8, nfeatures=1000)
fe.run(left, right); fe.sync()
for it in range(3):
    t0 = time.perf_counter()
    for k in range(64):
        f = iv.DeviceFrame.from_frontend(fe, k % 8, 0)
        del f
    print("round %d: %.1f us per create + destroy" % (it, (time.perf_counter() - t0) / 64 * 1e6))
