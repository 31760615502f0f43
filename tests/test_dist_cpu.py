"""world_size-2 / -4 gloo tests of the N>1 path: frame sharding (frame k -> rank k mod G) and the all-gather of
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


@pytest.mark.parametrize("world", [2, 4])
def test_shard_and_all_gather(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_frames = 4 * world
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs: p.join(60)
    assert [r[1] for r in res] == [True] * world
    for r in range(world):
        assert res[r][2] == list(range(r, n_frames, world))                # round-robin, disjoint, complete
    assert sorted(f for r in res for f in r[2]) == list(range(n_frames))


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("P", [1, 2, 16, 128])
def test_track_pairs_cover_every_consecutive_global_pair_exactly_once(world, P):
    """bench.py's pair tables (iv_slam_amd.dist.track_pairs): over all ranks every consecutive pair (g - 1, g) of the world * P
    global frames of one batch appears exactly once, each on the rank that extracted frame g (rank 0's wrap to the last rank's
    previous slot included); frame 0 has no predecessor inside the batch; every index stays inside the gathered buffer."""
    from iv_slam_amd import dist as ivd
    seen = {}
    for rank in range(world):
        for last, cur in ivd.track_pairs(world, rank, P):
            assert 0 <= last < world * P and 0 <= cur < world * P
            assert cur // P == rank, "a rank tracks only frames it extracted itself"
            g_last, g_cur = ivd.slot_frame(last, world, P), ivd.slot_frame(cur, world, P)
            assert g_cur == g_last + 1, (world, P, rank, last, cur)
            assert g_cur not in seen
            seen[g_cur] = rank
    assert sorted(seen) == list(range(1, world * P))
    # slot_frame is the inverse of frames_in_order
    for g, r, j in ivd.frames_in_order(None, world, P):
        assert ivd.slot_frame(r * P + j, world, P) == g


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("P", [1, 2, 16, 128])
def test_track_pairs_with_carry_cover_every_pair_of_the_stream_across_batches(world, P):
    """r05 (Tracking::TrackWithMotionModel runs for EVERY frame, Tracking.cc:1303-1330): with the carry record -- the last global
    frame of the previous batch in the slot behind the gathered ones -- every consecutive pair (g - 1, g) of the whole STREAM is
    produced exactly once over three consecutive batches, on the rank that extracted frame g; world * P pairs per batch over all
    ranks; the carry's source is the slot of a batch's last global frame."""
    from iv_slam_amd import dist as ivd
    B = world * P
    assert ivd.slot_frame(ivd.carry_source_slot(world, P), world, P) == B - 1
    assert ivd.slot_frame(ivd.carry_slot(world, P), world, P) == -1
    seen = {}
    for batch in range(3):
        per_batch = 0
        for rank in range(world):
            for last, cur in ivd.track_pairs(world, rank, P, carry=True):
                assert 0 <= last <= B and 0 <= cur < B          # `last` may be the carry slot, `cur` never
                assert cur // P == rank
                g_last = batch * B + ivd.slot_frame(last, world, P); g_cur = batch * B + ivd.slot_frame(cur, world, P)
                assert g_cur == g_last + 1, (world, P, rank, last, cur)
                assert g_cur not in seen
                seen[g_cur] = rank
                assert rank == g_cur % world
                per_batch += 1
        assert per_batch == B
    # frame 0's pair is (frame -1, frame 0): the first batch's carry record is empty (no keypoints, no matches)
    assert sorted(seen) == list(range(0, 3 * B))
    # and without the carry nothing changes for existing callers
    assert ivd.track_pairs(world, 0, P) == [p for p in ivd.track_pairs(world, 0, P, carry=True) if p[0] != ivd.carry_slot(world, P)]


def test_rank_to_numa_binding_from_sysfs(tmp_path, monkeypatch):
    """bind_rank_to_numa reads the KFD topology + PCI + node cpulists from sysfs only (no HIP call): a fake 2-socket tree with four
    GPUs, two per socket; visible-device remapping; a device without a NUMA node; nothing applied to this process (apply=False)."""
    from iv_slam_amd import dist as ivd
    sysfs = tmp_path
    nodes = sysfs / "class" / "kfd" / "kfd" / "topology" / "nodes"
    gpus = [(0x0500, 0), (0x2600, 0), (0x8500, 1), (0xa600, -1)]
    for i in range(2):                                                     # CPU nodes first, like a real box
        (nodes / str(i)).mkdir(parents=True); (nodes / str(i) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\n")
    for k, (loc, numa) in enumerate(gpus):
        d = nodes / str(2 + k); d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n" % loc)
        bdf = "0000:%02x:%02x.%x" % ((loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        pd = sysfs / "bus" / "pci" / "devices" / bdf; pd.mkdir(parents=True)
        (pd / "numa_node").write_text("%d\n" % numa)
    for n, cl in ((0, "0-3,64-67"), (1, "4-7")):
        nd = sysfs / "devices" / "system" / "node" / ("node%d" % n); nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False); monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    assert ivd.gpu_numa_nodes(str(sysfs)) == [("0000:05:00.0", 0), ("0000:26:00.0", 0), ("0000:85:00.0", 1), ("0000:a6:00.0", -1)]
    r = ivd.bind_rank_to_numa(0, str(sysfs), apply=False)
    assert r == {"bound": True, "device": 0, "pci": "0000:05:00.0", "numa_node": 0, "cpus": 8}
    assert ivd.bind_rank_to_numa(2, str(sysfs), apply=False)["numa_node"] == 1
    assert ivd.bind_rank_to_numa(3, str(sysfs), apply=False)["bound"] is False
    assert ivd.bind_rank_to_numa(7, str(sysfs), apply=False)["bound"] is False
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert ivd.bind_rank_to_numa(0, str(sysfs), apply=False)["pci"] == "0000:85:00.0"
    # the two lists compose: ROCr exposes KFD devices (3, 2, 0) as 0, 1, 2; HIP then takes (2, 0) of THOSE -> rank 0 = KFD 0, rank 1 = KFD 3
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "3,2,0")
    assert ivd.bind_rank_to_numa(0, str(sysfs), apply=False)["pci"] == "0000:05:00.0"
    assert ivd.bind_rank_to_numa(1, str(sysfs), apply=False)["pci"] == "0000:a6:00.0"
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert ivd.bind_rank_to_numa(1, str(sysfs), apply=False)["pci"] == "0000:85:00.0"      # ROCr list alone
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-deadbeef")
    assert ivd.bind_rank_to_numa(0, str(sysfs), apply=False)["bound"] is False             # UUID entries: not mapped, unbound
    assert ivd.parse_cpulist("0-2,8,10-11") == {0, 1, 2, 8, 10, 11} and ivd.parse_cpulist("") == set()


def test_record_layout_matches_c_abi_formula():
    from iv_slam_amd import dist as ivd
    assert ivd.record_bytes(1000) == 16 + 1000 * (24 + 32 + 4 + 4)
    from iv_slam_amd.frontend import unpack_gather_records
    r = _fake_result(3)
    back = unpack_gather_records(ivd.pack_records([r], NF), NF)[0]
    assert back["n"] == len(r["kps"]) and back["kps"].tobytes() == r["kps"].tobytes()
