"""One process per GPU: shard the stream's buffers over ranks, gather records to rank 0.

The path shards by whole 131072-sample buffers (all scan state is buffer-local in the reference,
dump1090.c:1567-1568); the only overlap between shards is the 476-byte carry each rank reads from
the input (dump1090.c:481), so there is NO data-path collective.  The single exchange is the
gather of the (tiny, variable-length) record lists to rank 0, which owns the one piece of
cross-buffer state - the ICAO whitelist - and runs the sequential resolve.  With backend "nccl"
(= RCCL on ROCm, over xGMI) the payload travels as CUDA uint8 tensors; with "gloo" (CPU tests) as
CPU tensors.
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .demod import shard_blocks, shard_byte_range  # noqa: F401  (re-exported)


_SIZE_BUFFERS = {}


def _size_buffers(device: str, world: int):
    """(host size, device size, device sizes of all ranks, host copy of those); on the CPU the pairs coincide."""
    import torch
    key = (device, world)
    if key not in _SIZE_BUFFERS:
        if device.startswith("cuda"):
            size_h = torch.zeros(1, dtype=torch.int64).pin_memory()
            all_h = torch.zeros(world, dtype=torch.int64).pin_memory()
            _SIZE_BUFFERS[key] = (size_h, torch.zeros(1, dtype=torch.int64, device=device),
                                  torch.zeros(world, dtype=torch.int64, device=device), all_h)
        else:
            size = torch.zeros(1, dtype=torch.int64)
            every = torch.zeros(world, dtype=torch.int64)
            _SIZE_BUFFERS[key] = (size, size, every, every)
    return _SIZE_BUFFERS[key]


def gather_arrays(arr: np.ndarray, dst: int = 0, group=None, device="cpu"):
    """Gather variable-length 1-D numpy arrays (any dtype) to rank `dst` in rank order.
    Returns the concatenation on `dst`, None elsewhere.  Two collectives: sizes, then padded data."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    # sizes: one collective into one tensor, one device-to-host copy (a .item() per rank costs a
    # synchronisation each - at 0.25 ms per step that alone would make the host the bottleneck);
    # the small tensors are allocated once per (device, world size)
    size_h, size_d, all_d, all_h = _size_buffers(str(device), world)
    size_h[0] = raw.size
    if size_d is size_h:                                   # CPU tensors (gloo)
        dist.all_gather_into_tensor(all_d, size_d, group=group)
    else:
        size_d.copy_(size_h, non_blocking=True)
        dist.all_gather_into_tensor(all_d, size_d, group=group)
        all_h.copy_(all_d, non_blocking=True)
        torch.cuda.current_stream(size_d.device).synchronize()
    sizes = [int(v) for v in all_h.tolist()]
    maxb = max(sizes)
    if maxb == 0:
        return arr[:0].copy() if rank == dst else None
    pad = torch.zeros(maxb, dtype=torch.uint8, device=device)
    if raw.size:
        pad[: raw.size] = torch.from_numpy(raw.copy()).to(device)
    bufs = [torch.empty(maxb, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    parts = [b[:n].cpu().numpy() for b, n in zip(bufs, sizes)]
    return np.concatenate(parts).view(arr.dtype)


def gather_records(records: np.ndarray, candidates, dst: int = 0, group=None, device="cpu"):
    """Rank-ordered concatenation of every rank's records (and candidate list) on `dst`.
    Ranks own ascending, disjoint buffer ranges, so the result is already in stream order."""
    recs = gather_arrays(np.ascontiguousarray(records, dtype=N.RECORD_DTYPE), dst, group, device)
    cands = None
    if candidates is not None:
        cands = gather_arrays(np.ascontiguousarray(candidates, dtype=np.uint64), dst, group, device)
    return recs, cands
