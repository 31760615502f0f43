"""Experiment: FCN time per image as a function of the batch one launch sequence covers (a hidden tensor of 8..16 images
fits the 256 MB Infinity Cache; does the expand -> depthwise round trip stay out of HBM then?)."""
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
s = torch.cuda.Stream(dev)
for n in [int(v) for v in os.environ.get("FCN_CHUNKS", "4,8,16,32,64,128").split(",")]:
    f = iv.IntrospectionFCN(blob, (H, W), (H, W), max_batch=n)
    def run():
        for k in range(P // n):
            f.forward_device(bgr[k * n:(k + 1) * n], cost_u8=cost[k * n:(k + 1) * n], stream_ptr=s.cuda_stream)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    print("chunk %3d images: %.1f us per image" % (n, (time.perf_counter() - t0) / 5 / P * 1e6), flush=True)
    if len(sys.argv) > 1 and n == int(sys.argv[1]):
        f.probe_enable(); run(); torch.cuda.synchronize()
        for r in f.probe_stats(0)[: 60]:
            print("   ", r)
    del f
