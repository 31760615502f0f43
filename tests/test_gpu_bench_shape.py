"""Parity at the launch shape bench.py runs (configs[2]): 1242x375, 1000 features, introspection ON, the FCN at the batch
size of the front end writing the cost maps the left extractor is gated with, FIVE consecutive launch sequences enqueued
without a synchronisation in between (all three batch contexts of the front end in flight, each context reused), batches
of 21 pairs (42 images: XCD groups 0..5, not a multiple of 8) and of 128 pairs (256 images: groups 0..31, what the
benchmark launches).  Sampled pairs of the three runs still held -- first / last / XCD-group boundaries -- are compared
with the oracle: keypoints (6 fields), descriptors, mvuRight, mvDepth, mvKeyQualScore bit for bit; the FCN cost map within
1e-3 of the CPU layer list (north_star's bar) and the u8 map within one LSB at truncation boundaries.

Reference: ORBextractor::operator() (ORB/src/ORBextractor.cc:1224-1296), Frame::ComputeStereoMatches (ORB/src/Frame.cc:758-932),
mvKeyQualScore (Frame.cc:130-143), the FCN call (ORB/Examples/Stereo/stereo_kitti.cc:493-514)."""
import concurrent.futures as cf
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

W, H, N = 1242, 375, 1000
BF, FX = 386.1448, 718.856
B = BF / FX


@pytest.fixture(scope="module")
def iv():
    import iv_slam_amd
    assert iv_slam_amd.load().ivf_device_count() >= 1, "no HIP device: libivfront has no CPU fallback"
    return iv_slam_amd


def quality_from_cost(kps, cost):
    """mvKeyQualScore (Frame.cc:130-143): cost/256, double division narrowed to float, 2q - 1."""
    px = O.c_round(kps["x"]); py = O.c_round(kps["y"])
    c = cost[py, px].astype(np.float32)
    q = (np.float64(1.0) / (np.float64(1.0) + (c / np.float32(256)).astype(np.float64))).astype(np.float32)
    return (np.float32(2) * q - np.float32(1)).astype(np.float32)


def oracle_pair(L, R, cost):
    oL = O.Extractor(N, 1.2, 8, 20, 7, introspection=True); oR = O.Extractor(N, 1.2, 8, 20, 7, introspection=False)
    kL, dL = oL(L, cost); kR, dR = oR(R, cost)                     # the right extractor ignores the map (Appendix D-7)
    ur, dp = O.stereo_match(oL, oR, kL, dL, kR, dR, BF, B)
    return kL, dL, kR, dR, ur, dp


def sample_pairs(P):
    """first / last pairs and the pairs either side of XCD-group boundaries (group = image // 8, image = 2*pair + side)."""
    want = [0, 1, 3, 4, 7, 8, P // 2 - 1, P // 2, P - 5, P - 4, P - 2, P - 1]
    if P > 64:
        want += [31, 32, 63, 64]
    out = sorted({p for p in want if 0 <= p < P})
    assert len(out) >= 12
    return out


@pytest.mark.parametrize("P", [21, 128])
def test_frontend_at_the_benchmark_launch_shape(iv, P):
    import torch
    import bench                    # the benchmark's own stream generator
    import fcn_oracle_torch         # checker only
    from iv_slam_amd import fcn_weights
    dev = torch.device("cuda:0")
    RUNS = 5
    n_stream = 2 * P if P > 64 else RUNS * P                          # P = 128: runs cycle through two slices like the benchmark
    left, right = bench.make_device_stream(torch, dev, n_stream, seed=300 + P)
    bgr = torch.stack([left, left // 2 + 40, 255 - left // 2], dim=-1).contiguous()
    Wt = fcn_weights.make_seeded_weights(7)
    fcn = iv.IntrospectionFCN(fcn_weights.pack_blob(Wt), (H, W), (H, W), max_batch=P)
    fe = iv.StereoFrontend(W, H, P, nfeatures=N, enableIntrospection=True, bf=BF, fx=FX)
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream
    # one cost buffer per run so that every run's maps can still be read afterwards; the benchmark reuses one buffer, which the
    # front end allows as soon as a batch has been ingested -- the LAST two runs share a buffer to exercise exactly that
    costs = [torch.empty((P, H, W), dtype=torch.uint8, device=dev) for _ in range(RUNS - 1)]
    cost_f32 = torch.empty((P, H, W), dtype=torch.float32, device=dev)
    nsl = n_stream // P
    slices = [(r % nsl) * P for r in range(RUNS)]
    for r in range(RUNS):
        s = slices[r]
        cbuf = costs[min(r, RUNS - 2)]
        fcn.forward_device(bgr[s:s + P], cost_u8=cbuf, cost_f32=cost_f32 if r == RUNS - 1 else None, stream_ptr=sptr)
        fe.run(left[s:s + P], right[s:s + P], cbuf, sptr)             # no sync between the five launch sequences
    fe.sync(); torch.cuda.synchronize(dev)

    pairs = sample_pairs(P)
    T = fcn_oracle_torch.prepare(Wt)
    pool = cf.ThreadPoolExecutor(min(os.cpu_count() or 1, 16))
    checked = 0
    for age in (2, 1, 0):                                             # the three runs still held: runs 2, 3, 4
        r = RUNS - 1 - age
        s = slices[r]
        # runs 3 and 4 shared a cost buffer: run 3's maps were overwritten after its ingest, so its gate is re-derived from the
        # FCN on the same input (bit-identical between launches: test_fcn_batch_is_deterministic), run 4's are read back
        if r == RUNS - 2:
            again = torch.empty((P, H, W), dtype=torch.uint8, device=dev)
            fcn.forward_device(bgr[s:s + P], cost_u8=again, stream_ptr=sptr)
            torch.cuda.synchronize(dev)
            cost_host = again.cpu().numpy()
        else:
            cost_host = costs[min(r, RUNS - 2)].cpu().numpy()
        Lh = left[s:s + P].cpu().numpy(); Rh = right[s:s + P].cpu().numpy()
        futs = {p: pool.submit(oracle_pair, Lh[p], Rh[p], cost_host[p]) for p in pairs}
        for p in pairs:
            kL, dL, kR, dR, ur, dp = futs[p].result()
            rl = fe.fetch(p, 0, age=age); rr = fe.fetch(p, 1, age=age)
            what = "P=%d run %d pair %d" % (P, r, p)
            assert len(kL) > N // 2, what
            assert rl["kps"].tobytes() == kL.tobytes(), what + ": left keypoints"
            assert rr["kps"].tobytes() == kR.tobytes(), what + ": right keypoints"
            assert np.array_equal(rl["desc"], dL) and np.array_equal(rr["desc"], dR), what + ": descriptors"
            assert rl["uright"].tobytes() == ur.tobytes() and rl["depth"].tobytes() == dp.tobytes(), what + ": stereo"
            assert np.array_equal(rl["quality"], quality_from_cost(kL, cost_host[p])), what + ": mvKeyQualScore"
            checked += 1
        if age == 0:
            # the cost maps themselves, FCN at batch P: within 1e-3 of the CPU layer list on the sampled images
            f32 = cost_f32.cpu().numpy()
            sel = pairs[:6] + pairs[-6:]
            ref, ref_u8 = fcn_oracle_torch.forward(T, bgr[s:s + P].cpu().numpy()[sel], (H, W))
            err = float(np.abs(f32[sel] - ref).max())
            assert err < 1e-3, "FCN at batch %d: cost map differs from the CPU layer list by %.3g" % (P, err)
            d = np.abs(cost_host[sel].astype(np.int32) - ref_u8.astype(np.int32))
            assert d.max() <= 1
            frac = (ref.astype(np.float64) * 255.0) % 1.0
            near = np.minimum(frac, 1.0 - frac) < 1e-3 * 255.0
            assert (d[~near] == 0).all()
            assert np.array_equal(cost_host, (f32 * np.float32(255)).astype(np.uint8)), "u8 map = trunc(cost * 255) (stereo_kitti.cc:511)"
    assert checked >= 3 * 12
