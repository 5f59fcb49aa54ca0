"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

The path shards two ways:
  * a batch of images  -> independent per-rank sub-batches, NO collective (greedy LPT on ECS bytes);
  * ONE huge image with restart intervals -> contiguous restart-segment ranges per rank; the only
    shared state is the descriptor (quantisation + Huffman tables, segment offsets: ~20 KB), which
    rank 0 -- the rank that scanned the file -- broadcasts; each rank's slice of the entropy-coded
    bytes is scattered from rank 0.  Results stay on their GPU (no gather).
"""
import ctypes as C

import numpy as np

from . import ImageDesc

_BLOB_HDR = C.sizeof(ImageDesc)


def lpt_assign(costs, world):
    """Greedy longest-processing-time assignment. -> list (per rank) of item indices.
    The reference sorts its inputs by file size for the same reason (decoder_host.cpp:46-61)."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    load = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        out[r].append(i)
        load[r] += costs[i]
    for r in range(world):
        out[r].sort()
    return out


def segment_range(n_segments, rank, world):
    """Contiguous restart-segment range [first, first+count) of `rank`."""
    base, rem = divmod(n_segments, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def pack_descriptor(desc: ImageDesc, seg_offsets: np.ndarray) -> np.ndarray:
    """Descriptor blob = the ImageDesc bytes (pointers zeroed) + the u64 segment offsets + u64 ecs_len."""
    d = ImageDesc()
    C.memmove(C.byref(d), C.byref(desc), _BLOB_HDR)
    d.ecs = None
    d.seg_offsets = None
    hdr = np.frombuffer(bytes(d), np.uint8)
    segs = np.ascontiguousarray(seg_offsets, np.uint64)
    return np.concatenate([hdr, segs.view(np.uint8)])


def unpack_descriptor(blob: np.ndarray):
    """-> (ImageDesc without ecs, seg_offsets array).  Caller attaches its ECS slice."""
    raw = np.ascontiguousarray(blob, np.uint8)
    d = ImageDesc.from_buffer_copy(raw[:_BLOB_HDR].tobytes())
    segs = raw[_BLOB_HDR:].view(np.uint64).copy()
    assert len(segs) == d.n_segments
    return d, segs


def broadcast_descriptor(blob, src=0, device=None):
    """Broadcast the descriptor blob from `src` (one collective, latency-bound).  With the nccl
    backend the tensor lives on `device` and travels over xGMI; with gloo it stays on the host."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    if rank == src:
        t.copy_(torch.from_numpy(np.ascontiguousarray(blob)))
    dist.broadcast(t, src)
    return t.cpu().numpy()


def scatter_ecs(ecs, seg_offsets, ecs_len, src=0, device=None):
    """Each rank receives the entropy-coded bytes of its own restart-segment range."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    nseg = len(seg_offsets)
    bounds = []
    for r in range(world):
        f, c = segment_range(nseg, r, world)
        lo = int(seg_offsets[f]) if c else 0
        hi = (int(seg_offsets[f + c]) if f + c < nseg else int(ecs_len)) if c else 0
        bounds.append((lo, hi))
    lo, hi = bounds[rank]
    out = torch.empty(hi - lo, dtype=torch.uint8, device=device)
    if rank == src:
        src_t = torch.from_numpy(np.ascontiguousarray(ecs)).to(device) if device is not None else torch.from_numpy(np.ascontiguousarray(ecs))
        reqs = []
        for r in range(world):
            a, b = bounds[r]
            if r == src:
                out.copy_(src_t[a:b])
            elif b > a:
                reqs.append(dist.isend(src_t[a:b].contiguous(), r))
        for q in reqs:
            q.wait()
    elif hi > lo:
        dist.recv(out, src)
    return out.cpu().numpy(), lo


def shard_descriptor(desc: ImageDesc, seg_offsets: np.ndarray, ecs_slice: np.ndarray, slice_lo: int, rank, world):
    """Build this rank's ImageDesc: full geometry, only its segments' bytes.
    The C ABI takes segment offsets relative to the ecs pointer it is given, so a rank that holds only
    its slice passes rebased offsets for its own range and marks the rest as empty."""
    f, c = segment_range(len(seg_offsets), rank, world)
    if c == 0:
        # more ranks than restart segments: this rank has nothing to decode (shard_n_segs == 0 would mean "all")
        return None, None
    d = ImageDesc()
    C.memmove(C.byref(d), C.byref(desc), _BLOB_HDR)
    segs = np.zeros(len(seg_offsets), np.uint64)
    own = seg_offsets[f:f + c].astype(np.int64) - slice_lo
    segs[:f] = 0
    segs[f:f + c] = own
    segs[f + c:] = len(ecs_slice)
    keep = (np.ascontiguousarray(ecs_slice, np.uint8), segs)
    d.ecs = keep[0].ctypes.data if len(ecs_slice) else None
    d.ecs_len = len(ecs_slice)
    d.seg_offsets = segs.ctypes.data
    d.n_segments = len(segs)
    d.shard_first_seg, d.shard_n_segs = f, c
    return d, keep


def distribute_image(scanned, src=0, device=None):
    """The whole exchange of the split-image case (BASELINE config 5): the rank that scanned the file broadcasts the
    descriptor (one collective) and sends every rank the entropy-coded bytes of its restart-segment range; each rank gets
    the ImageDesc of its shard.  `scanned`: a pjd_amd.Scanned on rank `src`, None elsewhere.
    -> (desc or None if this rank has no segments, keep-alive tuple, bytes of the descriptor blob)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == src:
        blob = pack_descriptor(scanned.desc, scanned.seg_offsets())
        ecs, n = scanned.ecs(), int(scanned.desc.ecs_len)
    else:
        blob, ecs, n = None, None, 0
    blob = broadcast_descriptor(blob, src=src, device=device)
    desc, segs = unpack_descriptor(blob)
    sl, lo = scatter_ecs(ecs, segs, n if rank == src else int(desc.ecs_len), src=src, device=device)
    d, keep = shard_descriptor(desc, segs, sl, lo, rank, world)
    return d, keep, len(blob)
