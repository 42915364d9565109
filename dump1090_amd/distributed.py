"""One process per GPU: shard the stream's buffers over ranks, gather the record lists to rank 0.

The path shards by whole 131072-sample buffers (all scan state is buffer-local in the reference,
dump1090.c:1567-1568); the only overlap between shards is the 476-byte carry each rank reads from
the input (dump1090.c:481), so there is NO data-path collective.  The single exchange is the
gather of the variable-length record lists to rank 0, which owns the one piece of cross-buffer
state - the ICAO whitelist - and runs the sequential resolve.

`RecordGather` does that exchange from DEVICE memory, in two asynchronous phases per call, so that the
host never waits for a collective it has just queued:

  counts   the kernels of modes_gpu_detect leave the ordered list and its length in caller-owned device
           buffers (modes_gpu_set_output); an all_gather of the 8-byte lengths is queued right behind them
           on the same stream, followed by a copy to pinned host memory - no host round trip in between;
  records  once the lengths are on the host (the step's one synchronisation), every rank with records
           sends exactly its n * 64 bytes and rank 0 receives each list at its final offset of one
           contiguous device buffer (rank order = stream order; rank 0's own list is already at the
           front): grouped point-to-point, 7 peers -> root over 7 distinct xGMI links with RCCL, no
           padding, no staging copy.  One device-to-host copy of the whole list follows on rank 0.

With backend "nccl" (= RCCL on ROCm) the tensors are CUDA tensors; with "gloo" (CPU tests, and
bench.py --backend gloo on a box with fewer GPUs than ranks) the same code moves CPU tensors.
"""
from __future__ import annotations

import numpy as np

from . import _native as N
from .demod import shard_blocks, shard_byte_range  # noqa: F401  (re-exported)


class GatherSlot:
    """Buffers of one call in flight (a pipelined host keeps several)."""

    def __init__(self, owner: "RecordGather"):
        import torch
        g = self.g = owner
        dev = g.device
        own = g.cap * 64
        # rank dst: room for every rank's list, its own first; the others: their own list only.  A group of ONE rank
        # (bench.py --force-gather on a one-GPU box) sends its list to itself into a second half - the point-to-point
        # calls of the N > 1 path executed over RCCL on the hardware at hand (loopback)
        self.records = torch.zeros(own * (max(2, g.world) if g.rank == g.dst else 1), dtype=torch.uint8, device=dev)
        self.count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.all_counts = torch.zeros(g.world, dtype=torch.int64, device=dev)
        if g.on_gpu:
            self.all_counts_h = torch.zeros(g.world, dtype=torch.int64).pin_memory()
            self.records_h = torch.zeros(self.records.numel(), dtype=torch.uint8).pin_memory() if g.rank == g.dst else None
            self.ev_counts = torch.cuda.Event(enable_timing=True)
            self.ev_records = torch.cuda.Event(enable_timing=True)
            self.ev_c0 = torch.cuda.Event(enable_timing=True)       # in front of the count all_gather / of the transfers:
            self.ev_r0 = torch.cuda.Event(enable_timing=True)       # on the communication stream, never in a launch stream
        else:
            self.all_counts_h = self.all_counts
            self.records_h = self.records
            self.ev_counts = self.ev_records = None
        self.counts = None          # list[int] once exchange_records() has run
        self._works = []
        self.p2p_ops = 0            # point-to-point operations this rank issued for the call (exchange_records)
        self.loopback = None        # group of one: did the list that travelled through send / recv arrive unchanged?

    @property
    def own_records(self):
        """Where this rank's ordered list goes (modes_gpu_set_output): the first cap * 64 bytes."""
        return self.records[: self.g.cap * 64]

    def exchange_counts(self, stream=None):
        """Queue the all_gather of the record counts behind whatever `stream` holds (the detect's kernels)."""
        import torch
        import torch.distributed as dist
        g = self.g
        self.counts = None
        if g.on_gpu:
            with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream(g.device)):
                self.ev_c0.record()
                dist.all_gather_into_tensor(self.all_counts, self.count, group=g.group)
                self.all_counts_h.copy_(self.all_counts, non_blocking=True)
                self.ev_counts.record()
        else:
            dist.all_gather_into_tensor(self.all_counts, self.count, group=g.group)

    def exchange_records(self, stream=None):
        """Wait for the counts; queue the point-to-point transfers of the lists (and the copy to the host on dst)."""
        import torch
        import torch.distributed as dist
        g = self.g
        if g.on_gpu:
            self.ev_counts.synchronize()
        self.counts = [int(v) for v in self.all_counts_h.tolist()]
        for r, n in enumerate(self.counts):
            if n > g.cap:
                raise N.ModesError(-4, "rank %d produced %d records, the gather buffers hold %d per rank" % (r, n, g.cap))
        ops = []
        if g.rank == g.dst:
            # rank order = stream order: the root's own list already sits at the front of its buffer (it owns the
            # first buffers of the stream), every other list lands right behind its predecessor's
            offs = self.counts[g.dst] * 64
            for r in range(g.world):
                if r == g.dst or self.counts[r] == 0:
                    continue
                nb = self.counts[r] * 64
                ops.append(dist.P2POp(dist.irecv, self.records[offs: offs + nb], g.global_rank(r), group=g.group))
                offs += nb
            if g.world == 1 and g.on_gpu and self.counts[0]:          # loopback: the root's list through isend / irecv to itself
                nb = self.counts[0] * 64
                me = g.global_rank(g.dst)
                ops.append(dist.P2POp(dist.irecv, self.records[g.cap * 64: g.cap * 64 + nb], me, group=g.group))
                ops.append(dist.P2POp(dist.isend, self.records[:nb], me, group=g.group))
        elif self.counts[g.rank]:
            ops.append(dist.P2POp(dist.isend, self.records[: self.counts[g.rank] * 64], g.global_rank(g.dst), group=g.group))
        self.p2p_ops = len(ops)
        self.loopback = None
        if g.on_gpu:
            with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream(g.device)):
                self.ev_r0.record()
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()                                  # stream-ordered (the host does not block)
                if g.world == 1 and ops:
                    nb = self.counts[0] * 64
                    # compared on the device, in stream order; the verdict is read in wait()
                    self.loopback = (self.records[:nb] == self.records[g.cap * 64: g.cap * 64 + nb]).all()
                if g.rank == g.dst:
                    total = sum(self.counts) * 64
                    if total:
                        self.records_h[:total].copy_(self.records[:total], non_blocking=True)
                self.ev_records.record()
        else:
            self._works = [op.op(op.tensor, op.peer, group=g.group) for op in ops]

    def wait(self):
        """-> the concatenated record list (numpy, RECORD_DTYPE, a view valid until the slot is reused) on dst, else None."""
        g = self.g
        if g.on_gpu:
            self.ev_records.synchronize()
        else:
            for w in self._works:
                w.wait()
            self._works = []
        if self.loopback is not None and not bool(self.loopback):
            raise N.ModesError(-2, "loopback: the record list that went through isend / irecv differs from the one sent")
        if g.rank != g.dst:
            return None
        total = sum(self.counts) * 64
        return self.records_h[:total].numpy().view(N.RECORD_DTYPE)

    def comm_ms(self):
        """GPU time of this call's two exchanges on the communication stream (count all_gather + its copy; list transfers +
        the copy to the host), ms - after wait().  0.0 on the CPU path."""
        if not self.g.on_gpu:
            return 0.0
        return float(self.ev_c0.elapsed_time(self.ev_counts) + self.ev_r0.elapsed_time(self.ev_records))

    def gathered_bytes(self):
        """Bytes that travelled between ranks for this call: every list but the root's own (after exchange_records)."""
        g = self.g
        if self.counts is None:
            return 0
        if g.world == 1:
            return self.counts[0] * 64 if self.p2p_ops else 0
        return 64 * (sum(self.counts) - self.counts[g.dst])


class RecordGather:
    def __init__(self, cap_records: int, device="cpu", dst: int = 0, group=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        assert dst == 0, "the gather root is rank 0: it owns the first buffers of the stream and the whitelist"
        self.dst = dst
        self.cap = int(cap_records)
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"

    def global_rank(self, r: int) -> int:
        import torch.distributed as dist
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def slot(self) -> GatherSlot:
        return GatherSlot(self)


def gather_arrays(arr: np.ndarray, dst: int = 0, group=None, device="cpu"):
    """Gather variable-length 1-D numpy arrays (any dtype) to rank `dst` in rank order (sizes first, then
    padded data).  Host-side convenience for small side lists (the --stats candidate positions); the record
    lists go through RecordGather."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    size = torch.tensor([raw.size], dtype=torch.int64, device=device)
    sizes_t = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes_t, size, group=group)
    sizes = [int(v) for v in sizes_t.tolist()]
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
    """Rank-ordered concatenation of every rank's HOST record array (and candidate list) on `dst`: the simple,
    blocking form for callers that already hold numpy records.  Ranks own ascending, disjoint buffer ranges,
    so the result is already in stream order."""
    recs = gather_arrays(np.ascontiguousarray(records, dtype=N.RECORD_DTYPE), dst, group, device)
    cands = None
    if candidates is not None:
        cands = gather_arrays(np.ascontiguousarray(candidates, dtype=np.uint64), dst, group, device)
    return recs, cands
