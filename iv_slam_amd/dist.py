"""Multi-GPU sharding of the stereo stream: frame k -> rank k mod G, plus the path's
one exchange step -- an all-gather of fixed-size per-pair records {n, kps, desc, uRight, depth} so every rank can
run cross-frame matching (ORBmatcher::SearchByProjection(cur,last), ORB/src/ORBmatcher.cc:1372) against the
previous frame wherever it was extracted.  torch.distributed is the transport ("nccl" = RCCL over xGMI on
the GPU box, "gloo" in CPU tests); the record layout is the C-ABI's (ivf_frontend_pack_gather_block).
"""
import numpy as np

from ._lib import KP_DTYPE


def record_bytes(nfeatures):
    return 16 + nfeatures * 24 + nfeatures * 32 + nfeatures * 8


def shard_frames(n_frames, rank, world):
    """Global frame indices owned by `rank` (round-robin: consecutive frames land on different GPUs)."""
    return list(range(rank, n_frames, world))


def pack_records(results, nfeatures):
    """Host-side twin of ivf_frontend_pack_gather_block: list of dict(kps, desc, uright[, depth]) -> uint8 [n, record]."""
    rec = record_bytes(nfeatures)
    out = np.zeros((len(results), rec), np.uint8)
    for i, r in enumerate(results):
        n = len(r["kps"])
        assert n <= nfeatures
        out[i, :4] = np.array([n], np.int32).view(np.uint8)
        out[i, 16:16 + n * 24] = np.ascontiguousarray(r["kps"], KP_DTYPE).view(np.uint8)
        out[i, 16 + nfeatures * 24:16 + nfeatures * 24 + n * 32] = np.ascontiguousarray(r["desc"], np.uint8).reshape(-1)
        out[i, 16 + nfeatures * 56:16 + nfeatures * 56 + n * 4] = np.ascontiguousarray(r["uright"], np.float32).view(np.uint8)
        if "depth" in r:
            out[i, 16 + nfeatures * 60:16 + nfeatures * 60 + n * 4] = np.ascontiguousarray(r["depth"], np.float32).view(np.uint8)
    return out


def all_gather_blocks(block, world):
    """block: uint8 tensor [pairs_per_rank * record] on this rank's device -> [world, pairs_per_rank * record]."""
    import torch
    import torch.distributed as dist
    out = torch.empty((world,) + tuple(block.shape), dtype=block.dtype, device=block.device)
    if world == 1:
        out[0] = block
        return out
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(parts, block)
        return torch.stack(parts)
    dist.all_gather_into_tensor(out.view(-1), block.view(-1))
    return out


def frames_in_order(gathered, world, pairs_per_rank):
    """gathered[r][j] holds global frame r + j*world; return record views ordered by global frame index."""
    order = []
    for j in range(pairs_per_rank):
        for r in range(world):
            order.append((r + j * world, r, j))
    return order
