"""world_size-2 gloo test of the N>1 path: frame sharding (frame k -> rank k mod G) and the all-gather of
per-pair descriptor records, with the record layout the C-ABI packs on the device."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NF = 64


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _fake_result(frame):
    from iv_slam_amd._lib import KP_DTYPE
    rng = np.random.default_rng(1000 + frame)
    n = int(rng.integers(10, NF + 1))
    k = np.zeros(n, KP_DTYPE)
    k["x"] = rng.uniform(0, 1242, n); k["y"] = rng.uniform(0, 375, n); k["octave"] = rng.integers(0, 8, n)
    k["response"] = frame
    return dict(kps=k, desc=rng.integers(0, 256, (n, 32)).astype(np.uint8), uright=rng.uniform(-1, 900, n).astype(np.float32),
                depth=rng.uniform(-1, 80, n).astype(np.float32))


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from iv_slam_amd import dist as ivd
    from iv_slam_amd.frontend import unpack_gather_records
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = ivd.shard_frames(n_frames, rank, world)
    block = torch.from_numpy(ivd.pack_records([_fake_result(f) for f in mine], NF).reshape(-1))
    gathered = ivd.all_gather_blocks(block, world)
    ok = gathered.shape == (world, len(mine) * ivd.record_bytes(NF))
    recs = [unpack_gather_records(gathered[r].numpy(), NF) for r in range(world)]
    for frame, r, j in ivd.frames_in_order(gathered, world, len(mine)):
        exp = _fake_result(frame); got = recs[r][j]
        ok &= got["n"] == len(exp["kps"]) and got["kps"].tobytes() == exp["kps"].tobytes()
        ok &= np.array_equal(got["desc"], exp["desc"]) and got["uright"].tobytes() == exp["uright"].tobytes()
        ok &= got["depth"].tobytes() == exp["depth"].tobytes()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok), mine))


def test_shard_and_all_gather_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world, n_frames = 2, 8
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs: p.join(60)
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5, 7]        # round-robin, disjoint, complete


def test_record_layout_matches_c_abi_formula():
    from iv_slam_amd import dist as ivd
    assert ivd.record_bytes(1000) == 16 + 1000 * (24 + 32 + 4 + 4)
    from iv_slam_amd.frontend import unpack_gather_records
    r = _fake_result(3)
    back = unpack_gather_records(ivd.pack_records([r], NF), NF)[0]
    assert back["n"] == len(r["kps"]) and back["kps"].tobytes() == r["kps"].tobytes()
