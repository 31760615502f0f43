"""Times the 960 -> 160 launch the FCN probe brackets (k_fcn_dwpw<5,4>) at batch 128.  tools/dwpw_ablate.sh runs it once per
ablation build (compile-time mask IVF_DWPW_ABL in ivf_fcn.hip: results of such builds are wrong by construction, only the
launch time is read)."""
import os, sys
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
f = iv.IntrospectionFCN(blob, (H, W), (H, W), max_batch=P)
for _ in range(2):
    f.forward_device(bgr, cost_u8=cost, stream_ptr=s.cuda_stream)
torch.cuda.synchronize()
f.probe_enable()
for _ in range(5):
    f.forward_device(bgr, cost_u8=cost, stream_ptr=s.cuda_stream)
torch.cuda.synchronize()
ms, n, b = f.probe_stats(0)
print("%s%s: %.1f us per launch (%d launches of %d images)" % ("", f.probe_info()[0], ms / n * 1e3, n, b), flush=True)
