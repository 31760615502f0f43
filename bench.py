#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the hot path (introspection FCN + ORB extract L+R + L/R stereo match) on MI355X.

Contract: python bench.py --gpus N --steps K --warmup W  (N>1 is launched by torch.distributed.run,
one rank per GPU).  One step = one pass of the hot path over one batch of `--pairs` synthetic
1242x375 stereo pairs that are already resident in HBM; value = pairs all ranks processed / time
(max over ranks).  Workload = BASELINE.json configs[2], the configuration the metric "extract+match+introspect" names:
the introspection FCN runs on every left image inside the timed step and its cost map gates the left extractor.
--no-introspect measures configs[1] (extract + match only); the default run also reports that figure in
`extract_match_only`, measured after the timed region.  Prints ONE JSON line on rank 0, carrying `roofline` for the
dominant kernel (HIP events on its stream inside the timed region: the FCN's 960->160 fused depthwise+projection
launch with introspection, k_fast_nms without) and `cpu_baseline` (the oracles = ports of the reference CPU path,
timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "stereo pairs/s (extract+match+introspect) @1242×375, 1000 feat; 1/2/4/8 GPU"
W, H, NFEAT = 1242, 375, 1000
BF, FX = 386.1448, 718.856
LEVEL_PX_SUM = 1441432            # sum of pixels over the 8 levels
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_device_stream(torch, dev, n_pairs, seed, base_pairs=16):
    """>=256 distinct pairs without minutes of host synthesis: `base_pairs` seeded host pairs, each
    expanded on the GPU by a per-image (dx,dy) roll applied to left AND right (disparity preserved)
    plus +-1 seeded noise."""
    from iv_slam_amd import synth
    base = synth.make_stream(base_pairs, W, H, seed=seed)
    bl = torch.from_numpy(base[:, 0].copy()).to(dev); br = torch.from_numpy(base[:, 1].copy()).to(dev)
    g = torch.Generator(device=dev); g.manual_seed(1234 + seed)
    left = torch.empty((n_pairs, H, W), dtype=torch.uint8, device=dev); right = torch.empty_like(left)
    for i in range(n_pairs):
        b = i % base_pairs; k = i // base_pairs
        dx, dy = (37 * k) % 200, (11 * k) % 40
        for src, dst in ((bl, left), (br, right)):
            img = torch.roll(src[b], shifts=(dy, dx), dims=(0, 1)).to(torch.int16)
            img += torch.randint(-1, 2, img.shape, generator=g, device=dev, dtype=torch.int16) * (k > 0)
            dst[i] = img.clamp_(0, 255).to(torch.uint8)
    return left, right


def cpu_baseline(pairs_sample, cores):
    """Oracle (bit-exact scalar restatement of the reference CPU path; kind = 'port') on host threads.
    One pair per worker thread (ctypes releases the GIL); the reference itself uses 2 threads per pair."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import concurrent.futures as cf
    import oracle_lib as O          # checker / baseline only -- never on the product path
    from iv_slam_amd import synth
    imgs = [synth.make_pair(W, H, seed=900, idx=i) for i in range(min(pairs_sample, 8))]
    b = BF / FX

    def work(i):
        L, R = imgs[i % len(imgs)]
        eL = O.Extractor(NFEAT, 1.2, 8, 20, 7); eR = O.Extractor(NFEAT, 1.2, 8, 20, 7)
        kL, dL = eL(L); kR, dR = eR(R)
        O.stereo_match(eL, eR, kL, dL, kR, dR, BF, b)
        return 1

    work(0)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        done = sum(ex.map(work, range(pairs_sample)))
    dt = time.perf_counter() - t0
    return done / dt, dt


def cpu_fcn_baseline(threads):
    """numpy oracle of the introspection FCN (oracle/fcn_oracle.py, f32): one forward per worker thread (numpy releases
    the GIL inside its kernels; BLAS keeps its default threading)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import concurrent.futures as cf
    import numpy as np
    import fcn_oracle               # checker / baseline only -- never on the product path
    from iv_slam_amd import fcn_weights, synth
    Wt = fcn_weights.make_seeded_weights(7)
    img = np.stack([synth.make_left(W, H, seed=901, idx=c) for c in range(3)], axis=-1)
    fcn_oracle.forward(Wt, img, (H, W))
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(threads) as ex:
        done = sum(1 for _ in ex.map(lambda i: fcn_oracle.forward(Wt, img, (H, W)), range(threads)))
    dt = time.perf_counter() - t0
    return done / dt, dt


# algorithmic HBM bytes of the probed FCN launch per image: hidden tensor read once
# (960 x 64 x 64 f32), output written and residual read (160 x 64 x 64 f32 each); weights are L2-resident
FCN_PROBE_BYTES_PER_IMAGE = (960 + 160 + 160) * 64 * 64 * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=128, help="stereo pairs per step per GPU")
    ap.add_argument("--stream", type=int, default=256, help="distinct pairs resident per GPU")
    ap.add_argument("--introspect", action="store_true", help="(default) configs[2]: run the introspection FCN on every left image and gate keypoints with it")
    ap.add_argument("--no-introspect", action="store_true", help="configs[1]: extract + match only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-gather", action="store_true", help="test aid: run the multi-GPU exchange step (RCCL all-gather of "
                    "the descriptor records) even with one rank")
    ap.add_argument("--serial", action="store_true", help="profiling aid: wait for each batch before enqueuing the next, so "
                    "rocprofv3 kernel durations are not inflated by the overlap of consecutive batches")
    args = ap.parse_args()
    args.introspect = not args.no_introspect

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    exchange = world > 1 or args.force_gather
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)      # "nccl" is RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import iv_slam_amd as iv
    from iv_slam_amd import synth
    P = args.pairs
    n_stream = max(args.stream, P)
    n_stream = (n_stream + P - 1) // P * P
    left, right = make_device_stream(torch, dev, n_stream, seed=100 + rank)
    cost = None
    fcn = None
    if args.introspect:
        # configs[2]: the introspection FCN (random-init weights of the deployed architecture; no checkpoints
        # exist offline) runs on the left colour image of every pair inside the timed step and its u8 cost map
        # gates keypoint selection in the left extractor.
        from iv_slam_amd import fcn_weights
        blob = fcn_weights.pack_blob(fcn_weights.make_seeded_weights(7))
        fcn = iv.IntrospectionFCN(blob, (H, W), (H, W), max_batch=P, device_id=local_rank)
        fcn.probe_enable()
        bgr = torch.stack([left, left // 2 + 40, 255 - left // 2], dim=-1).contiguous()      # [n,H,W,3] colour-ish
        cost = torch.empty((P, H, W), dtype=torch.uint8, device=dev)
    fe = iv.StereoFrontend(W, H, P, nfeatures=NFEAT, enableIntrospection=args.introspect, bf=BF, fx=FX,
                           device_id=local_rank)
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream
    rec = fe.gather_record_bytes()
    block = torch.empty(P * rec, dtype=torch.uint8, device=dev)
    gathered = torch.empty(world * P * rec, dtype=torch.uint8, device=dev) if exchange else None
    nslices = n_stream // P
    # The exchange step (pack + all-gather) is enqueued on the internal stream the batch itself runs on -- in order behind
    # it, so no stream ever waits on a batch-completion event (a stream that does costs 8 % of configs[2]: DESIGN.md
    # section 6) and the batch that reuses the context three steps later simply queues behind the collective.  One block /
    # gather buffer per internal stream.  IVF_BENCH_EXCHANGE=side selects the earlier variant (own stream + event waits).
    side = torch.cuda.Stream(dev) if exchange else None
    packed = [torch.cuda.Event() for _ in range(3)] if exchange else None
    blocks3 = [torch.empty_like(block) for _ in range(3)] if exchange else None          # one per internal stream
    gathered3 = [torch.empty_like(gathered) for _ in range(3)] if exchange else None
    nstep = [0]

    def step(i):
        s = (i % nslices) * P
        k = nstep[0]; nstep[0] += 1
        if fcn is not None:
            fcn.forward_device(bgr[s:s + P], cost_u8=cost, stream_ptr=sptr)
        mode = os.environ.get("IVF_BENCH_EXCHANGE", "batch-stream")
        if exchange and mode == "side" and k >= 3:
            stream.wait_event(packed[k % 3])                     # the pack of three steps ago has read the context run(k) reuses
        fe.run(left[s:s + P], right[s:s + P], cost, sptr)
        if exchange and mode == "side":
            with torch.cuda.stream(side):
                fe.pack_gather_block(block, side.cuda_stream)
                packed[k % 3].record(side)
                dist.all_gather_into_tensor(gathered, block)
        elif exchange:
            # the path's one exchange step: all-gather of {n, kps, desc, uRight} for cross-frame matching.  Pack and
            # collective go onto the internal stream the batch itself runs on: in order behind it, so no stream ever waits
            # on a batch-completion event, and the batch that reuses the context three steps later queues behind them
            bs = torch.cuda.ExternalStream(fe.batch_stream(0), device=dev)
            fe.pack_gather_block(blocks3[k % 3], fe.STREAM_OF_BATCH)
            with torch.cuda.stream(bs):
                dist.all_gather_into_tensor(gathered3[k % 3], blocks3[k % 3])
        if args.serial:
            fe.sync()
            if exchange:
                side.synchronize()

    for i in range(args.warmup):
        step(i)
    fe.sync()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    host_enqueue_ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)      # host time to enqueue one step
    fe.sync()                                           # batches run on the front end's own streams: wait for all of
    torch.cuda.synchronize(dev)                         # them (also checks the device-side consistency flag)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    fast_sum_ms, fast_n = fe.fast_ms_stats(min(args.steps, 64))
    if fcn is not None:
        probe_sum_ms, probe_n, probe_batch = fcn.probe_stats(min(args.steps, 64))
    # The front end overlaps consecutive batches on its own streams, so inside the timed region the probed kernel shares
    # the GPU with other kernels and its event-measured duration is inflated.  For the kernel's own figure, 5 more steps
    # are run one at a time (sync between them) AFTER the timed region and reported separately as `isolated`.
    for i in range(5):
        step(args.warmup + args.steps + i)
        fe.sync()
        torch.cuda.synchronize(dev)
    iso_sum_ms, iso_n = fe.fast_ms_stats(5)
    if fcn is not None:
        iso_probe_ms, iso_probe_n, _ = fcn.probe_stats(5)
    # the network alone (MFMA leg of north_star): 3 forwards with nothing else on the GPU, HIP events on its stream
    fcn_alone = None
    if fcn is not None and world == 1:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        fcn.forward_device(bgr[:P], cost_u8=cost, stream_ptr=sptr)
        e0.record(stream)
        for _ in range(3):
            fcn.forward_device(bgr[:P], cost_u8=cost, stream_ptr=sptr)
        e1.record(stream); torch.cuda.synchronize(dev)
        us_img = e0.elapsed_time(e1) * 1e3 / 3 / P
        # 17.229 GFLOP per forward, of which the 1x1 convs (14.61) and the decoder 3x3 (1.89) run on MFMA, each
        # product as three f16 MFMAs (split-f16: hi*hi + hi*lo + lo*hi, f32 accumulate)
        fcn_alone = {"us_per_image": round(us_img, 2), "batch": P,
                     "f32_equivalent_tflops": round(17.229e9 / us_img / 1e6, 1),
                     "mfma_f16_tflops_issued": round(3 * (14.61 + 1.89) * 1e9 / us_img / 1e6, 1), "mfma_f16_dense_peak_tflops": 2500.0,
                     "mfma_frac": round(3 * (14.61 + 1.89) * 1e9 / us_img / 1e6 / 2500.0, 4)}
    # configs[1] (no introspection) on the same stream of pairs, for reference next to the headline number
    em_only = None
    if fcn is not None and world == 1:
        fe1 = iv.StereoFrontend(W, H, P, nfeatures=NFEAT, enableIntrospection=False, bf=BF, fx=FX, device_id=local_rank)
        for i in range(3):
            s0 = (i % nslices) * P; fe1.run(left[s0:s0 + P], right[s0:s0 + P], None, sptr)
        fe1.sync(); torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(10):
            s0 = (i % nslices) * P; fe1.run(left[s0:s0 + P], right[s0:s0 + P], None, sptr)
        fe1.sync(); torch.cuda.synchronize(dev)
        em_only = P * 10 / (time.perf_counter() - t1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_pairs = P * args.steps * world
    value = total_pairs / dt

    # sanity on the last batch: keypoints were really produced and matched
    r0 = fe.fetch(0, 0)
    assert len(r0["kps"]) > NFEAT // 2 and (r0["uright"] >= 0).sum() > 20, "degenerate output"

    if rank == 0:
        def load_pmc(kernel_key):
            # HBM traffic from the committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE collected in separate runs
            # because counters cannot be read inside this process), per image, rescaled to this launch size
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
                k = pmc["kernels"][kernel_key]
                return (k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]) / k.get("images_per_launch", pmc["images_per_launch"]), \
                    "profiles/r01_pmc_hbm_traffic.json"
            except Exception:
                return None, None

        def roof(kernel, algo_bytes, sum_ms, n, iso_ms, iso_n, traffic, src):
            ms = sum_ms / max(n, 1); ims = iso_ms / max(iso_n, 1)
            ach = algo_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            iach = algo_bytes / (ims * 1e-3) / 1e9 if ims > 0 else 0.0
            return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": src,
                    "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": round(ms, 5), "launches_timed": n,
                    "isolated": {"note": "same kernel, 5 launches after the timed region with no other batch in flight",
                                 "avg_launch_ms": round(ims, 5), "achieved": round(iach, 2), "frac": round(iach / HBM_PEAK_GBS, 5)}}

        imgs_per_launch = 2 * P
        fast_algo = (LEVEL_PX_SUM + NFEAT * 8) * imgs_per_launch      # pyramid read + candidate output
        per_img, src = load_pmc("ivf::k_fast_nms")
        fast_roof = roof("k_fast_nms", fast_algo, fast_sum_ms, fast_n, iso_sum_ms, iso_n,
                         None if per_img is None else int(per_img * imgs_per_launch), src)
        if fcn is not None:
            per_img, src = load_pmc("ivffcn::k_fcn_dwpw<5, 4> 960->160")
            roofline = roof("k_fcn_dwpw<5,4> (fused depthwise 3x3 + 1x1 projection 960->160, %d images)" % probe_batch,
                            FCN_PROBE_BYTES_PER_IMAGE * probe_batch, probe_sum_ms, probe_n, iso_probe_ms, iso_probe_n,
                            None if per_img is None else int(per_img * probe_batch), src)
        else:
            roofline = fast_roof
        out = {
            "metric": METRIC, "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "host_enqueue_ms_per_step": round(host_enqueue_ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8 (ORB) + f32 via split-f16 MFMA (FCN)" if args.introspect else "u8",
            "data": "synthetic",
            "config": {"workload": ("configs[2]: 1242x375 stereo stream with introspection FCN cost-map forward (MFMA convs) gating keypoints, "
                                    "ORB extract + L/R Hamming match"
                                    if args.introspect else
                                    "configs[1]: 1242x375 stereo pair stream, ORB extract + L/R Hamming match, introspection OFF"),
                       "pairs_per_step_per_gpu": P, "distinct_pairs_per_gpu": n_stream, "nfeatures": NFEAT,
                       "nlevels": 8, "scale_factor": 1.2, "fast_thresholds": [20, 7],
                       "parallelism": "frames sharded %d-way, RCCL all-gather of descriptor blocks" % world if world > 1 else "1 GPU"},
            "roofline": roofline,
        }
        if fcn is not None:
            out["roofline_fast_nms"] = fast_roof
            if fcn_alone is not None:
                out["fcn_forward"] = fcn_alone
            if em_only is not None:
                out["extract_match_only"] = {"value": round(em_only, 2), "unit": "pairs/s",
                                             "note": "configs[1] (introspection OFF), 10 steps on the same resident pairs after the timed region"}
        if not args.no_cpu_baseline:
            cores = max(1, min(os.cpu_count() or 1, 32))
            sample = max(128, 8 * cores)
            v, secs = cpu_baseline(sample, cores)
            txt = ("%d pairs of the same 1242x375/1000-feature workload, one pair per thread, %.1f s wall "
                   "(oracle/libivf_oracle.so, scalar C, -O3 -ffp-contract=off)" % (sample, secs))
            if args.introspect:
                fv, fsecs = cpu_fcn_baseline(cores)
                txt += ("; + introspection FCN: %d forwards of the numpy oracle (oracle/fcn_oracle.py), one per thread, "
                        "%.1f s wall = %.2f images/s; extract+match alone = %.1f pairs/s; value = 1/(1/a + 1/b)" % (cores, fsecs, fv, v))
                v = 1.0 / (1.0 / v + 1.0 / fv)
            out["cpu_baseline"] = {"value": round(v, 3), "unit": "pairs/s", "cores": cores, "kind": "port", "sample": txt}
        print(json.dumps(out), flush=True)
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
