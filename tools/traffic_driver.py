"""Counter-pass driver (tools/pmc_traffic.sh): the FCN on 128 left images and the front end on 128 pairs, strictly one after
the other with a device synchronisation in between (PMC passes serialise dispatches; bench.py's overlapping streams can stall
a FETCH_SIZE pass).  Same inputs and handles as bench.py's default workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iv_slam_amd as iv
from iv_slam_amd import fcn_weights
from bench import make_device_stream, W, H, NFEAT, BF, FX
dev = torch.device("cuda:0")
P = 128
left, right = make_device_stream(torch, dev, P, seed=100)
bgr = torch.stack([left, left // 2 + 40, 255 - left // 2], dim=-1).contiguous()
cost = torch.empty((P, H, W), dtype=torch.uint8, device=dev)
fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(fcn_weights.make_seeded_weights(7)), (H, W), (H, W), max_batch=P)
fe = iv.StereoFrontend(W, H, P, nfeatures=NFEAT, enableIntrospection=True, bf=BF, fx=FX)
for _ in range(3):
    fcn.forward_device(bgr, cost_u8=cost); torch.cuda.synchronize()
    fe.run(left, right, cost); fe.sync(); torch.cuda.synchronize()
print("done")
