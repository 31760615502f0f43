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


def track_pairs(world, rank, pairs_per_rank, carry=False):
    """(last, cur) record indices into the all-gathered buffer [world][pairs_per_rank] for the frames THIS rank extracted: slot
    (r, j) holds global frame j * world + r, so the frame before (r, j) is (r - 1, j), or (world - 1, j - 1) for r = 0.
    Rank 0's first frame of a batch has its predecessor in the PREVIOUS batch.  carry = False: that pair is left out (world * P - 1
    pairs per batch).  carry = True (r05; Tracking::TrackWithMotionModel runs for EVERY frame, Tracking.cc:1303-1330): the buffer
    holds one more record at index carry_slot() = world * P -- the last global frame of the previous batch, put there by
    BoundaryCarry -- and rank 0 tracks (carry, first): world * P pairs per batch over all ranks, every consecutive global pair of
    the stream exactly once (the very first batch finds an empty carry record: no keypoints, no matches)."""
    P = pairs_per_rank
    out = []
    for j in range(P):
        if rank > 0:
            out.append(((rank - 1) * P + j, rank * P + j))
        elif j > 0:
            out.append(((world - 1) * P + j - 1, j))
        elif carry:
            out.append((carry_slot(world, P), 0))
    return out


def carry_slot(world, pairs_per_rank):
    """index of the extra record behind the world * P gathered ones: the last global frame of the previous batch"""
    return world * pairs_per_rank


def carry_source_slot(world, pairs_per_rank):
    """slot of a batch's LAST global frame (rank world - 1, j = P - 1): what the next batch needs as its carry record"""
    return world * pairs_per_rank - 1


class BoundaryCarry:
    """Hands the last record of batch k to batch k + 1 across the batch contexts' streams (torch plumbing around one
    device-to-device copy).  Batch k's records live in buffers[k % len(buffers)] = [world * P records | 1 carry record], complete on
    the stream batch k ran on once its pack / all-gather is enqueued there.

        carry.publish(k, stream_k)        # behind batch k's pack / all-gather, on batch k's stream
        carry.acquire(k, stream_k)        # in front of batch k's tracker step (waits for batch k - 1's publish)
        carry.release(k, stream_k)        # behind batch k's tracker step

    publish copies record carry_source_slot of buffer k into the carry slot of buffer k + 1 ON BATCH k's STREAM and records an event;
    acquire makes batch k + 1's stream wait for it -- the one real cross-stream wait per batch: the boundary pair needs both batches.
    The carry slot it writes was last read by the tracker step of batch k + 1 - len(buffers), whose release event is long complete in
    steady state: publish queries it on the host and inserts a wait only if it is not (formally safe, free in practice)."""

    def __init__(self, buffers, world, pairs_per_rank, record_bytes):
        import torch
        self.torch = torch
        self.buffers = buffers
        self.n = len(buffers)
        self.rec = record_bytes
        self.src = carry_source_slot(world, pairs_per_rank) * record_bytes
        self.dst = carry_slot(world, pairs_per_rank) * record_bytes
        for b in buffers:
            assert b.numel() >= self.dst + record_bytes, "record buffers need one carry record behind the world * P gathered ones"
        self.published = [torch.cuda.Event() for _ in range(self.n)]      # [k % n]: carry slot of buffer k written (by batch k - 1)
        self.released = [torch.cuda.Event() for _ in range(self.n)]       # [k % n]: tracker step of batch k has read buffer k
        self.have_published = [False] * self.n
        self.have_released = [False] * self.n

    def publish(self, k, stream):
        nxt = (k + 1) % self.n
        if self.have_released[nxt] and not self.released[nxt].query():
            stream.wait_event(self.released[nxt])
        with self.torch.cuda.stream(stream):
            self.buffers[nxt][self.dst:self.dst + self.rec].copy_(self.buffers[k % self.n][self.src:self.src + self.rec], non_blocking=True)
        self.published[nxt].record(stream)
        self.have_published[nxt] = True

    def acquire(self, k, stream):
        if self.have_published[k % self.n]:
            stream.wait_event(self.published[k % self.n])

    def release(self, k, stream):
        self.released[k % self.n].record(stream)
        self.have_released[k % self.n] = True


def slot_frame(slot, world, pairs_per_rank):
    """global frame index (within one batch) held by slot index `slot` = r * pairs_per_rank + j of the gathered buffer; the carry
    slot holds frame -1 = the last frame of the previous batch."""
    if slot == carry_slot(world, pairs_per_rank):
        return -1
    r, j = divmod(slot, pairs_per_rank)
    return j * world + r


# ---- rank -> device -> NUMA node ------------------------------------------------------------------------------------------
def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every GPU in KFD topology order (the order HIP enumerates devices in when HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES do not reorder them), read from sysfs only -- no HIP call, so a rank can use it before it touches the GPU.
    Returns a list of (pci_bdf, numa_node or -1)."""
    import os
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    out = []
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return out
    for n in nodes:
        txt = _read(os.path.join(base, str(n), "properties"))
        if not txt:
            continue
        props = dict(l.split(None, 1) for l in txt.splitlines() if len(l.split(None, 1)) == 2)
        if int(props.get("simd_count", "0")) == 0:
            continue                                   # a CPU node
        loc = int(props.get("location_id", "0")); dom = int(props.get("domain", "0"))
        bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        numa = _read(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node"))
        out.append((bdf, int(numa) if numa is not None and numa.lstrip("-").isdigit() else -1))
    return out


def parse_cpulist(txt):
    cpus = set()
    for part in (txt or "").split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def bind_rank_to_numa(local_rank, sysfs="/sys", apply=True):
    """Pin this rank's host threads to the CPUs of the NUMA node its GPU hangs off (one process per GPU: the rank's enqueue loop
    and its pinned staging buffers then sit next to the device).  Call it in a FRESH rank process before anything initialises
    the GPU.  Best effort: returns a dict saying what was done or why nothing was."""
    import os
    gpus = gpu_numa_nodes(sysfs)
    # the two lists COMPOSE (r04 ADVICE): ROCR_VISIBLE_DEVICES filters / reorders what the ROCr runtime exposes (indices into the
    # KFD order), HIP_VISIBLE_DEVICES (or CUDA_VISIBLE_DEVICES) then indexes into what ROCr left.  Only integer lists are mapped
    # (UUID entries: give up, unbound).  Assumption that remains: KFD topology order = ROCr enumeration order, which holds on the
    # driver's boxes; bench.py reports the PCI address it bound to, so a mismatch with hipDeviceGetPCIBusId is visible in the line.
    idx = local_rank
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):            # innermost mapping first
        vis = os.environ.get(var) if var != "HIP_VISIBLE_DEVICES" else (os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES"))
        if not vis:
            continue
        try:
            idx = [int(v) for v in vis.split(",")][idx]
        except (ValueError, IndexError):
            return {"bound": False, "why": "cannot map local rank %d through %s=%r" % (local_rank, var, vis)}
    if not (0 <= idx < len(gpus)):
        return {"bound": False, "why": "no KFD topology entry for device %d (%d GPU nodes found)" % (idx, len(gpus))}
    bdf, numa = gpus[idx]
    if numa < 0:
        return {"bound": False, "device": idx, "pci": bdf, "why": "the device reports no NUMA node"}
    cpus = parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", "node%d" % numa, "cpulist")))
    if apply and hasattr(os, "sched_getaffinity"):
        cpus &= os.sched_getaffinity(0)                # never widen what the launcher / container allows
    if not cpus:
        return {"bound": False, "device": idx, "pci": bdf, "numa_node": numa, "why": "no usable CPU on that node"}
    if apply:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError as e:
            return {"bound": False, "device": idx, "pci": bdf, "numa_node": numa, "why": "sched_setaffinity: %s" % e}
    return {"bound": True, "device": idx, "pci": bdf, "numa_node": numa, "cpus": len(cpus)}
