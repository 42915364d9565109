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


# ---------------------------------------------------------------------------------------------------------------------
# Resolve on the ranks that demodulated.
#
# The gather above brings every rank's records to rank 0, whose host then runs the one sequential piece of the path for
# all N GPUs (the ICAO whitelist, dump1090.c:896-925, :1183-1210).  RankResolve leaves the records where they are: the
# speculation of modes_host_resolve_raw_mt (modes_host.cpp) one level up, a rank = a piece of the step.
#
#   1  every rank lists what its clean DF11/17/18 frames would write to the whitelist (whitelist_guess: 4 KiB);
#      one all_gather (host memory) - with rank 0's clock, which all ranks use for the step;
#   2  rank r resolves its own records - in stream order they follow those of ranks 0 .. r-1 - from the state the step
#      started with, overlaid with the guesses of the ranks before it, logging the slots it wrote and every whitelist
#      question it answered from that start state (raw_listing_spec);
#   3  one all_gather of what every rank wrote (8 KiB) and its line / byte / counter totals; every rank rebuilds the
#      state that REALLY preceded it from the tables of the ranks before it and checks its logged answers against it;
#   4  one all_gather of the verdicts.  All good (the rule): the texts are final.  Else the first rank with a wrong
#      answer - the ranks before it are final, so the state it has just rebuilt is the true one - resolves again from
#      that state, and 3 - 4 repeat for the ranks behind it; every round settles at least one more rank;
#   5  the texts travel to rank 0 (31 bytes a line instead of 64 a record), which concatenates them in rank order.
#
# Same answers -> same control flow -> same output: the listing is byte-identical to the sequential resolve of the
# gathered records (tests/test_host.py: every stream, 2 .. 8 ranks, spoiled guesses; tests/test_distributed.py over gloo).
#
# The protocol is a generator (rank_resolve_step) that yields what it wants exchanged: LocalRanks runs N of them in one
# process (tests), RankResolve runs one per process over torch.distributed.
# ---------------------------------------------------------------------------------------------------------------------
_HDR = 2 + len(N.STAT_NAMES)          # lines, bytes, the --stats counters; then one word per whitelist slot


def _pack_writes(written, addr, seen):
    """int64[ICAO_SLOTS]: -1 = slot untouched, else seen << 24 | addr (24-bit addresses, a non-negative clock)."""
    assert int(seen.min()) >= 0 and int(seen.max()) < (1 << 38)
    return np.where(written != 0, (seen.astype(np.int64) << 24) | addr.astype(np.int64), np.int64(-1))


def _apply_writes(addr, seen, w):
    m = w >= 0
    addr[m] = (w[m] & 0xFFFFFF).astype(np.uint32)
    seen[m] = w[m] >> 24


def rank_resolve_step(make_resolver, rank, world, start, segments, threads=1, now=0, spoil=False, text_buffer=None):
    """One step of the protocol on one rank, as a generator.
    yields ("gather", int64 array)            -> send() it the list of every rank's array, in rank order
           ("text", uint8 view, [nbytes ...]) -> send() rank 0 the list of every rank's text (its own first), the others None
    returns (StopIteration.value) a dict: lines / nbytes / stats of the whole step, texts (rank 0), the whitelist the
    step leaves (identical on every rank), reruns of this rank, rounds, the resolver (for its listing buffer).
    start = (addr, seen): the true whitelist the step starts from, the same on every rank.
    spoil: make this rank's guess of its start state wrong on purpose (tests: the re-run path)."""
    addr0, seen0 = start
    res = make_resolver(text_buffer)
    guess = res.whitelist_guess(segments, threads)
    gs = yield ("gather", np.concatenate([np.array([now], dtype=np.int64), guess.astype(np.int64)]))
    now = int(gs[0][0])                                  # one clock for the step: rank 0's
    res.set_time(now)
    addr, seen = addr0.copy(), seen0.copy()
    for r in range(rank):
        g = gs[r][1:]
        m = g != N.ICAO_NONE
        addr[m] = g[m].astype(np.uint32)
        seen[m] = now
    if spoil:                                            # forget everything: every "known" answer of the true run comes out wrong
        addr[:] = 0
        seen[:] = 0
    res.set_whitelist(addr, seen)
    lines, nbytes, written, lookups = res.raw_listing_spec(segments, threads)
    reruns = rounds = 0
    while True:
        rounds += 1
        a, s = res.whitelist()
        st = res.stats()
        head = np.array([lines, nbytes] + [st[k] for k in N.STAT_NAMES], dtype=np.int64)
        ws = yield ("gather", np.concatenate([head, _pack_writes(written, a, s)]))
        addr, seen = addr0.copy(), seen0.copy()
        for r in range(rank):
            _apply_writes(addr, seen, ws[r][_HDR:])
        probe = make_resolver(None)
        probe.set_time(now)
        probe.set_whitelist(addr, seen)
        ok = probe.whitelist_check(lookups)
        probe.close()
        oks = yield ("gather", np.array([1 if ok else 0], dtype=np.int64))
        bad = [r for r in range(world) if not int(oks[r][0])]
        if not bad:
            break
        if bad[0] == rank:                               # the ranks before this one are final: `addr, seen` is the true start
            buf = res.take_text_buffer()
            res.close()
            res = make_resolver(buf)
            res.set_time(now)
            res.set_whitelist(addr, seen)
            lines, nbytes, written, lookups = res.raw_listing_spec(segments, threads)
            reruns += 1
    for r in range(rank, world):                         # the whitelist the step leaves: every rank's writes, in order
        _apply_writes(addr, seen, ws[r][_HDR:])
    sizes = [int(w[1]) for w in ws]
    texts = yield ("text", res.text_view(nbytes), sizes)
    stats = {k: int(sum(int(w[2 + i]) for w in ws)) for i, k in enumerate(N.STAT_NAMES)}
    stats["valid_preamble"] = -1                         # (records only: like modes_host_get_stats without candidates)
    return {"lines": int(sum(int(w[0]) for w in ws)), "nbytes": int(sum(sizes)), "stats": stats, "texts": texts,
            "whitelist": (addr, seen), "reruns": reruns, "rounds": rounds, "resolver": res, "own_lines": lines}


class LocalRanks:
    """N ranks' protocol generators run in lockstep in ONE process (tests, and a host that holds several GPUs' lists):
    step(segments_per_rank) -> (listing bytes, the per-rank result dicts)."""

    def __init__(self, world, flags, threads=1):
        from .demod import HostResolver
        self.world, self.threads = world, threads
        self.make = lambda buf=None: HostResolver(text_buffer=buf, **flags)
        self.state = (np.zeros(N.ICAO_SLOTS, dtype=np.uint32), np.zeros(N.ICAO_SLOTS, dtype=np.int64))

    def step(self, segments_per_rank, now=0, spoil=()):
        gens = [rank_resolve_step(self.make, r, self.world, self.state, segments_per_rank[r], self.threads, now, r in spoil)
                for r in range(self.world)]
        asks = [next(g) for g in gens]
        results = [None] * self.world
        while any(r is None for r in results):
            kind = asks[0][0]
            assert all(a[0] == kind for a in asks)          # every rank is at the same exchange
            if kind == "gather":
                reply = [[a[1].copy() for a in asks]] * self.world
            else:
                texts = [a[1].copy() for a in asks]
                assert [t.size for t in texts] == asks[0][2]
                reply = [texts] + [None] * (self.world - 1)
            nxt = []
            for r, g in enumerate(gens):
                try:
                    nxt.append(g.send(reply[r]))
                except StopIteration as e:
                    results[r] = e.value
            asks = nxt
        for r in results:
            r["resolver"].close()
        self.state = results[0]["whitelist"]
        for r in results[1:]:                               # every rank ends the step with the same whitelist
            assert np.array_equal(r["whitelist"][0], self.state[0]) and np.array_equal(r["whitelist"][1], self.state[1])
        return b"".join(t.tobytes() for t in results[0]["texts"]), results


class RankResolve:
    """The protocol of one rank of a torch.distributed job.  ctl: a process group over HOST memory (gloo) for the three small
    all_gathers, used by the calling thread only; the texts travel through it too (CPU tensors), or - device given - over the
    default group from device memory (RCCL: host -> device on the sender, exact-size point-to-point transfers into one buffer on
    rank 0, one copy back).  fresh=True: every step starts from an empty whitelist (a bench step replays its stream from the
    beginning); False: from the one the step before left."""

    def __init__(self, flags, threads, ctl, device=None, fresh=True, cap_bytes=1 << 22):
        import torch
        import torch.distributed as dist
        from .demod import HostResolver
        self.dist, self.torch = dist, torch
        self.ctl = ctl
        self.world = dist.get_world_size(ctl)
        self.rank = dist.get_rank(ctl)
        self.threads = threads
        self.fresh = fresh
        self.make = lambda buf=None: HostResolver(text_buffer=buf, **flags)
        self.state = self._empty()
        self.device = torch.device(device) if device is not None else None
        self.on_gpu = self.device is not None and self.device.type == "cuda"
        self._prev = None                                   # the resolver of the step before (its listing buffer is handed on)
        self._dev = self._pin = None
        self._cap = 0
        if self.on_gpu:
            self.stream = torch.cuda.Stream(device=self.device)
            self._grow(cap_bytes)
        self.work_s = self.exchange_s = 0.0                 # seconds in the C calls / inside the exchanges (waiting for peers included)
        self.phase_s = [0.0] * 8                            # work_s by stretch between exchanges: guess, resolve, check, totals, end (+ re-runs)
        self.steps = self.reruns = self.rounds = 0
        self.bytes_moved = self.p2p_ops = 0

    @staticmethod
    def _empty():
        return np.zeros(N.ICAO_SLOTS, dtype=np.uint32), np.zeros(N.ICAO_SLOTS, dtype=np.int64)

    def _grow(self, nbytes):
        torch = self.torch
        if nbytes > self._cap:
            self._cap = int(nbytes * 5 // 4)
            self._dev = torch.empty(self._cap, dtype=torch.uint8, device=self.device)
            self._pin = torch.empty(self._cap, dtype=torch.uint8).pin_memory()

    # Every gather of the protocol travels in ONE fixed shape - word 0 = "this rank is well", word 1 = the payload's length, the payload
    # padded to the longest one: a rank whose own work fails between two exchanges (a ModesError in its resolve, an assertion) still
    # takes part in the next gather, with word 0 = 0, and EVERY rank raises there and then - instead of the peers sitting in an
    # all_gather the failed rank never joins until the group's timeout.
    _WORDS = 2 + _HDR + N.ICAO_SLOTS

    def _gather(self, arr, failed=None):
        torch, dist = self.torch, self.dist
        mine = torch.zeros(self._WORDS, dtype=torch.int64)
        if failed is None:
            a = np.ascontiguousarray(arr, dtype=np.int64)
            assert a.size <= self._WORDS - 2
            mine[0], mine[1] = 1, a.size
            mine[2: 2 + a.size] = torch.from_numpy(a)
        out = torch.empty(self.world * self._WORDS, dtype=torch.int64)
        dist.all_gather_into_tensor(out, mine, group=self.ctl)
        rows = out.numpy().reshape(self.world, -1)
        bad = [r for r in range(self.world) if not int(rows[r][0])]
        if bad:
            if failed is not None:
                raise failed
            raise N.ModesError(-5, "resolve on the ranks: rank(s) %s failed inside the step; every rank stops here" % bad)
        return [rows[r][2: 2 + int(rows[r][1])] for r in range(self.world)]

    def _texts(self, view, sizes):
        """-> rank 0: [uint8 array per rank] (views of buffers that live until the next step), others None"""
        torch, dist = self.torch, self.dist
        me, n = self.rank, int(view.size)
        assert sizes[me] == n
        peers = [r for r in range(self.world) if r != 0 and sizes[r]]
        moved = sum(sizes[r] for r in peers)
        if not self.on_gpu:
            if me == 0:
                self.bytes_moved += moved
                self.p2p_ops += len(peers)
                parts = {r: torch.empty(sizes[r], dtype=torch.uint8) for r in peers}
                works = [dist.irecv(parts[r], dist.get_global_rank(self.ctl, r), group=self.ctl) for r in peers]
                for w in works:
                    w.wait()
                return [view] + [parts[r].numpy() if r in parts else np.zeros(0, dtype=np.uint8) for r in range(1, self.world)]
            if n:
                dist.send(torch.from_numpy(view), dist.get_global_rank(self.ctl, 0), group=self.ctl)
                self.p2p_ops += 1
            return None
        # device memory: the default group is RCCL; its peers are GLOBAL ranks - `ctl` may number the ranks its own way
        g = lambda r: dist.get_global_rank(self.ctl, r)
        with torch.cuda.stream(self.stream):
            if me == 0:
                self._grow(moved + (n if self.world == 1 else 0))
                ops, offs, where = [], 0, {}
                for r in peers:
                    where[r] = (offs, sizes[r])
                    ops.append(dist.P2POp(dist.irecv, self._dev[offs: offs + sizes[r]], g(r)))
                    offs += sizes[r]
                if self.world == 1 and n:                   # a group of one: the text goes through isend / irecv to itself (the calls of
                    src = torch.from_numpy(view).to(self.device, non_blocking=False)        # the N > 1 path on the hardware at hand)
                    ops += [dist.P2POp(dist.irecv, self._dev[:n], g(0)), dist.P2POp(dist.isend, src, g(0))]
                    offs = n
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                    self._pin[:offs].copy_(self._dev[:offs], non_blocking=True)
                self.stream.synchronize()
                self.bytes_moved += offs
                self.p2p_ops += len(ops)
                if self.world == 1:
                    if n and not np.array_equal(self._pin[:n].numpy(), view):
                        raise N.ModesError(-2, "loopback: the text that went through isend / irecv differs from the one sent")
                    return [view]
                host = self._pin.numpy()
                return [view] + [host[where[r][0]: where[r][0] + where[r][1]] if r in where else np.zeros(0, dtype=np.uint8)
                                 for r in range(1, self.world)]
            if n:
                self._grow(n)
                self._pin[:n].copy_(torch.from_numpy(view))
                self._dev[:n].copy_(self._pin[:n], non_blocking=True)
                for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, self._dev[:n], g(0))]):
                    w.wait()
                self.p2p_ops += 1
            self.stream.synchronize()
            return None

    def step(self, segments, now=0, spoil=False):
        """Resolve this rank's records of one step (arrays in stream order, whole buffers each).  -> the result dict of
        rank_resolve_step; on rank 0 `texts` holds every rank's listing in rank order (views: valid until the next step)."""
        import time
        if self.fresh:
            self.state = self._empty()
        buf = None
        if self._prev is not None:                          # the step before is consumed: its listing buffer serves again
            buf = self._prev.take_text_buffer()
            self._prev.close()
            self._prev = None
        gen = rank_resolve_step(self.make, self.rank, self.world, self.state, segments, self.threads, now, spoil, buf)
        t = time.perf_counter()
        try:
            ask = next(gen)
        except Exception as e:                              # noqa: BLE001 - this rank's own work failed before the first exchange:
            self._gather(None, failed=e)                    # the peers are told in it (raises e here, ModesError there)
            raise
        ph = 0
        while True:
            t1 = time.perf_counter()
            self.work_s += t1 - t
            self.phase_s[min(ph, 7)] += t1 - t
            ph += 1
            reply = self._gather(ask[1]) if ask[0] == "gather" else self._texts(ask[1], ask[2])
            t = time.perf_counter()
            self.exchange_s += t - t1
            try:
                ask = gen.send(reply)
            except StopIteration as e:
                out = e.value
                break
            except Exception as e:                          # noqa: BLE001 - ... or between two exchanges: the next one is a gather
                # (behind the text exchange nothing is left to exchange; behind the round's last gather - every rank's check came
                #  back good - the peers' next exchange is the texts, not a gather: nothing but two array copies lies in between)
                last_check = ask[0] == "gather" and ask[1].size == 1 and all(int(x[0]) for x in reply)
                if ask[0] == "gather" and not last_check:
                    self._gather(None, failed=e)
                raise
        self.work_s += time.perf_counter() - t
        self.phase_s[min(ph, 7)] += time.perf_counter() - t
        self.state = out["whitelist"]
        self._prev = out.pop("resolver")                    # rank 0's own text is a view of its buffer: alive until the next step
        self.steps += 1
        self.reruns += out["reruns"]
        self.rounds += out["rounds"]
        return out

    def close(self):
        if self._prev is not None:
            self._prev.close()
            self._prev = None
