import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import iv_slam_amd as iv
from iv_slam_amd import synth, fcn_weights
W,H,N=1242,375,1000
L,R=synth.make_pair(W,H,seed=77,idx=0)
bgr=np.stack([L,L//2+40,255-L//2],axis=-1).astype(np.uint8)
blob=fcn_weights.pack_blob(fcn_weights.make_seeded_weights(7))
eL=iv.ORBextractor(N,1.2,8,20,7,True); eR=iv.ORBextractor(N,1.2,8,20,7,False)
fcn=iv.IntrospectionFCN(blob,(H,W),(H,W),max_batch=1)
def t(fn,reps=20):
    fn(); ts=[]
    for _ in range(reps):
        t0=time.perf_counter(); r=fn(); ts.append((time.perf_counter()-t0)*1e3)
    return np.median(ts), r
tf,cost=t(lambda: fcn(bgr))
tl,(kL,dL)=t(lambda: eL(L,cost))
tr,(kR,dR)=t(lambda: eR(R))
ts_,_=t(lambda: iv.ComputeStereoMatches(eL,eR,kL,dL,kR,dR,386.1448,386.1448/718.856))
print("fcn %.2f ms  extract L (cost) %.2f  extract R %.2f  stereo %.2f"%(tf,tl,tr,ts_))
eL.mvImagePyramid  # touch
import torch
dev = torch.device("cuda:0")
bd = torch.from_numpy(bgr[None].copy()).to(dev); cu8 = torch.empty((1, H, W), dtype=torch.uint8, device=dev)
def fdev():
    fcn.forward_device(bd, cost_u8=cu8); torch.cuda.synchronize()
td, _ = t(fdev)
fe = iv.StereoFrontend(W, H, 1, nfeatures=N, enableIntrospection=True)
Ld = torch.from_numpy(L[None].copy()).to(dev); Rd = torch.from_numpy(R[None].copy()).to(dev)
def fedev():
    fe.run(Ld, Rd, cu8); fe.sync()
tfe, _ = t(fedev)
def both():
    fcn.forward_device(bd, cost_u8=cu8); fe.run(Ld, Rd, cu8); fe.sync()
tb, _ = t(both)
print("device-resident batch 1: fcn %.2f ms  front end (L+R+stereo) %.2f ms  both %.2f ms" % (td, tfe, tb))
