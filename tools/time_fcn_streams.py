"""Experiment: does running the FCN on two half-batches concurrently (two handles, two streams) beat one full batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iv_slam_amd as iv
from iv_slam_amd import fcn_weights
from bench import make_device_stream, W, H
dev = torch.device("cuda:0")
P = 128
left, _ = make_device_stream(torch, dev, P, seed=100)
bgr = torch.stack([left, left // 2 + 40, 255 - left // 2], dim=-1).contiguous()
blob = fcn_weights.pack_blob(fcn_weights.make_seeded_weights(7))
cost = torch.empty((P, H, W), dtype=torch.uint8, device=dev)
def bench(nsplit, reps=6):
    n = P // nsplit
    fcns = [iv.IntrospectionFCN(blob, (H, W), (H, W), max_batch=n) for _ in range(nsplit)]
    streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
    def run():
        for k in range(nsplit):
            fcns[k].forward_device(bgr[k * n:(k + 1) * n], cost_u8=cost[k * n:(k + 1) * n], stream_ptr=streams[k].cuda_stream)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / P * 1e6
for ns in (1, 2, 4):
    print("FCN %d image(s) per handle x %d concurrent streams: %.1f us per image" % (P // ns, ns, bench(ns)))
