"""The pipelined step loop of a host that keeps a GPU (one rank of N) busy: several detect calls in flight,
the record lists fetched - or gathered to rank 0 from device memory - behind them, the sequential resolve on its
own thread.  bench.py drives it with the HIP Demodulator; tests/test_pipeline.py drives the same code on CPU
(gloo, world_size 2) with a stand-in detector fed by the oracle."""
from __future__ import annotations

import gc
import queue
import threading
import time

import numpy as np


class Resolver(threading.Thread):
    """Rank 0's sequential half on its own thread (the C host does the same): record lists in, --raw listing out.
    A fresh whitelist per step: every step demodulates the same stream from its beginning; the calls of one step
    share it (like the batches of one file)."""

    def __init__(self, flags, threads=1, hold_views=True, rank_resolve=None, root=True):
        super().__init__(daemon=True)
        self.rr = rank_resolve          # distributed.RankResolve: every rank runs a Resolver and resolves its OWN records; the ranks
        self.root = root                # confirm each other and the texts travel to rank 0 (root) - no record leaves its rank
        self.exchange_s = 0.0           # seconds of the timed steps inside that protocol's exchanges (waiting for the peers included)
        self.flags = flags
        self.hold_views = hold_views    # keep the gather buffers of a step's calls until the step is resolved (else: copy)
        self.threads = threads          # modes_host_resolve_raw_mt: pieces of a long list resolved in parallel (exact)
        self.q = queue.Queue()
        self.msgs = 0                   # messages of the timed steps
        self.step_text = []             # listing of the step in progress, one piece per call (steps whose text is kept)
        self.step_lines = 0
        self.last_text = b""            # listing of the last complete step
        self.last_lines = 0
        self.error = None
        self._res = None
        self._stale = True              # the next record starts from a fresh whitelist
        self.submitted = 0              # items handed in (main thread) / worked off completely (resolver thread)
        self.completed = 0
        self.resolve_s = 0.0            # seconds inside the resolve + formatting of the TIMED steps (this thread)
        self.start()

    def submit(self, recs, counts, first_call, last_call, timed, done_event, keep_text=True):
        """recs: the records of one call - this rank's (counts None) or the gathered lists of all ranks, rank after
        rank (counts = records per rank).  keep_text: the step's listing is wanted as a Python object (last_text);
        otherwise it is formatted all the same, and only counted."""
        self.submitted += 1
        self.q.put((recs, counts, first_call, last_call, timed, done_event, keep_text))

    def take_empty(self, first_call, last_call, keep_text=True):
        """A call WITHOUT a record on a host that resolves its own lists, taken by the launching thread itself when this thread has
        nothing queued (then it touches none of this until the next put): the step's bookkeeping and no hand-over - two thread
        wake-ups per call are 30 us of the ~100 us a timed region of the record-free workload idles the chip at its end.
        -> False: queue it like any other call (the order of a step's calls is the order of the queue)."""
        if self.rr is not None or self.completed < self.submitted:
            return False
        if first_call:
            self._stale = True
            self.step_text = []
            self.step_lines = 0
            self._parts = []
        if last_call:
            self.last_lines = self.step_lines
            if keep_text:
                self.last_text = b"".join(self.step_text)
        return True

    def _fresh_host(self):
        """The step's whitelist, made when the step's first record arrives: a fresh one, with the old listing buffer (a new one per
        step would be zero-filled and faulted in under the GIL: 7 ms for the 34 MB of an 8-GPU step - the launching thread stalls,
        the GPU runs dry).  A step without a record never makes one: on the record-free headline workload the resolver's share of a
        timed region's last moments - four hand-overs, each with a host made and closed - was 0.11 ms of a 20-step region's 4.3."""
        if self._stale:
            from .demod import HostResolver
            buf = None
            if self._res is not None:
                buf = self._res.take_text_buffer()
                self._res.close()
            self._res = HostResolver(text_buffer=buf, **self.flags)
            self._stale = False
        return self._res

    def _resolve(self, recs, timed, keep):
        if len(recs) == 0:                                      # nothing to resolve, nothing to print
            return
        t_a = time.perf_counter()
        n, text = self._fresh_host().raw_listing(recs, None, threads=self.threads, text=keep)
        if timed:
            self.resolve_s += time.perf_counter() - t_a
        if keep:
            self.step_text.append(text)
        self.step_lines += n
        if timed:
            self.msgs += n

    def run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, threading.Event):               # drain(): everything submitted before it is resolved
                item.set()
                continue
            recs, counts, first_call, last_call, timed, done, keep = item
            try:
                if self.rr is not None:
                    self._rank_step(recs, first_call, last_call, timed, done, keep)
                    continue
                if first_call:
                    self._stale = True                          # a fresh whitelist for this step (_fresh_host)
                    self.step_text = []
                    self.step_lines = 0
                    self._parts = []
                if counts is None or (first_call and last_call):
                    self._resolve(recs, timed, keep)            # one rank, or one call per step: already in stream order
                    done.set()
                else:
                    # several calls per step and several ranks: stream order is rank-major (rank 0's calls, then
                    # rank 1's ...), the gathered lists arrive call-major - keep the pieces (views of the gather buffers:
                    # their slots stay busy until the step is resolved), resolve at the step's end
                    offs = np.concatenate([[0], np.cumsum(counts)])
                    if self.hold_views:
                        self._parts.append(([recs[offs[r]: offs[r + 1]] for r in range(len(counts))], done))
                    else:                                       # fewer buffers than calls per step: release this one now
                        self._parts.append(([recs[offs[r]: offs[r + 1]].copy() for r in range(len(counts))], None))
                        done.set()
                    if last_call:
                        # rank-major = stream order; ONE parallel resolve over all the pieces where they lie
                        # (modes_host_resolve_raw_mtv): with a call per (rank, call) piece rank 0 of an 8-GPU run spent
                        # 16 x 0.5 ms per step here against 2.2 ms of kernels
                        segs = [call[r] for r in range(len(counts)) for call, _ in self._parts]
                        t_a = time.perf_counter()
                        n, text = self._fresh_host().raw_listing_segments(segs, threads=self.threads, text=keep)
                        if timed:
                            self.resolve_s += time.perf_counter() - t_a
                        if keep:
                            self.step_text.append(text)
                        self.step_lines += n
                        if timed:
                            self.msgs += n
                        for _, ev in self._parts:
                            if ev is not None:
                                ev.set()
                        self._parts = []
                if last_call:
                    self.last_lines = self.step_lines
                    if keep:                                    # (joining and counting 12 MB under the GIL stalls the launches)
                        self.last_text = b"".join(self.step_text)
            except Exception as e:          # noqa: BLE001 - reported by the main thread
                self.error = e
                done.set()
                for _, ev in getattr(self, "_parts", []):
                    if ev is not None:
                        ev.set()
            finally:
                self.completed += 1

    def _rank_step(self, recs, first_call, last_call, timed, done, keep):
        """A call's records on a rank that resolves its own: kept (views of the context's list, or copies when there are fewer
        contexts than a step has calls) until the step's last call is here - the protocol runs once per step, over the rank's
        records of the step in stream order."""
        if first_call:
            self._parts = []
        if self.hold_views:
            self._parts.append((recs, done))
        else:
            self._parts.append((recs.copy(), None))
            done.set()
        if not last_call:
            return
        rr = self.rr
        w0, x0 = rr.work_s, rr.exchange_s
        out = rr.step([p for p, _ in self._parts])
        if timed:
            self.resolve_s += rr.work_s - w0
            self.exchange_s += rr.exchange_s - x0
            self.msgs += out["lines"]
        self.step_lines = self.last_lines = out["lines"]
        if keep and self.root:
            self.last_text = b"".join(t.tobytes() for t in out["texts"])
        for _, ev in self._parts:
            if ev is not None:
                ev.set()
        self._parts = []

    def drain(self):
        """Block until everything submitted so far is resolved (a released buffer only says its records were taken over).
        Normally the resolver is a few microseconds from done when this is called (the caller has just seen its last
        buffer released): yield to it a few times before paying for a round trip through the queue (two thread wake-ups,
        ~80 us - 1.5 % of a 20-step timed region)."""
        for _ in range(200):
            if self.completed >= self.submitted:
                return
            time.sleep(0)
        ev = threading.Event()
        self.q.put(ev)
        ev.wait()

    def stop(self):
        self.q.put(None)
        self.join()


def split_calls(first_block, nblocks, ncalls, lo, total_bytes):
    """A rank's buffers in `ncalls` contiguous GPU calls (one call holds at most 8 GiB - 64 KiB of samples):
    -> [(first_block, nblocks, byte_lo, byte_hi)] with the bytes each call needs (its buffers + the 476-byte carry)."""
    from .demod import shard_blocks, shard_byte_range
    out = []
    for c in range(ncalls):
        b0, nb = shard_blocks(nblocks, ncalls, c)
        clo, chi = shard_byte_range(first_block + b0, nb, total_bytes)
        out.append((first_block + b0, nb, clo, chi))
    return out


def run_steps(*args, **kwargs):
    """_run_steps with the garbage collector's state restored afterwards (it is switched off for the timed regions)."""
    was = gc.isenabled()
    try:
        return _run_steps(*args, **kwargs)
    finally:
        if was:
            gc.enable()


def _run_steps(make_demod, iq, lo, calls, flags, steps, warm, depth, world=1, rank=0, dist=None, coll_device=None,
              cap_records=1 << 16, streams=(None,), device_sync=lambda: None, time_every=8, resolve_threads=1,
              gather=None, oplog=None, lag=None, resolve_on="root", ctl_group=None, regions=1):
    """`warm` untimed + `steps` timed steps of the hot path over this rank's shard `iq` (stream bytes from `lo`;
    a CUDA uint8 tensor - anything sliceable that make_demod()'s detect accepts); a step is the sequence `calls`
    of GPU calls (split_calls).  `depth` contexts (make_demod() each) are used in rotation, so that the GPU always
    has the next call queued while the host fetches, gathers and resolves the previous ones:

        call i           detect queued on streams[i mod len] (three kernels, nothing else in that stream)
        call i - lag     N > 1: its kernels are done (a word in pinned memory) -> all_gather of the record counts queued
        call i - lag - 1 N > 1: counts on the host -> exact-size transfers of the device-resident lists to rank 0 queued
        call i - lag - 2 records on rank 0's host -> resolver thread (sequential resolve + --raw formatting)
                         (N = 1: call i - 2 goes straight to the resolver)
    lag (default: the number of launch streams) = the calls that stay queued on the GPU while the host waits for an older
    one: with two launch streams two calls run concurrently, and the host must not wait for the older of them before
    the next one is queued.

    Kernel times cost idle GPU time (events around the kernels: ~9 us per boundary), so only one call in `time_every`
    carries them (modes_gpu_set_timing); the averages returned are over those calls of the timed steps.

    regions > 1: the timed part is `regions` regions of `steps` steps each, back to back behind ONE warm-up, every one of them
    bracketed by the flush + barrier + device sync of a single region; "elapsed" is then the MEDIAN region (max over the ranks
    per region first), "elapsed_regions" has them all - one region of 20 steps is 4 ms, a sample of a distribution whose spread
    is larger than a round's gain.  Per-step figures are averages over all regions' steps.

    gather: exchange the record lists through RecordGather (default: when world > 1; True with world == 1 runs the whole
    N > 1 code path - device output buffers, count all_gather, transfers - on a single rank, which is how the RCCL calls
    are exercised on a one-GPU box).
    world > 1: `dist` = torch.distributed (initialised), coll_device = where the gathered bytes travel (the CUDA
    device with RCCL; "cpu" with gloo: the lists are fetched to the host first).  Returns a dict of measurements
    (rank 0: also the --raw listing of the last step).

    resolve_on="ranks" (world > 1, or gather=True on one rank): no record leaves its rank - every rank fetches its own list
    like a single GPU's host does and resolves it on its own resolver thread; the ranks exchange whitelist guesses, confirm each
    other and send their TEXT to rank 0 (distributed.RankResolve; ctl_group: a gloo group for its host-memory all_gathers, made
    here when not given).  The launching threads then issue no communication call at all between the barriers.

    Every rank issues its communication calls in the SAME order - per call n: detect(n), count all_gather(n - 1), list
    transfers(n - 2); at a flush: what is left of the newest calls - whatever its resolver thread is doing: RCCL
    executes a communicator's operations in issue order, so a rank that queued "transfers(n - 1), all_gather(n)" against
    peers that queued "all_gather(n), transfers(n - 1)" would deadlock.  oplog (a list): receives ("detect" | "counts" |
    "records", call number) as they are issued (tests/test_pipeline.py compares the ranks' logs)."""
    import torch
    from ._native import ModesError
    from .distributed import RecordGather

    dist_on = world > 1 if gather is None else bool(gather)          # records travel through RecordGather
    ranks_mode = resolve_on == "ranks" and dist_on
    if ranks_mode:
        dist_on = False                                              # ... unless every rank resolves its own (RankResolve)
    sync_ranks = world > 1 or ranks_mode or dist_on
    demods = [make_demod() for _ in range(depth)]
    works = list(streams)
    slots = None
    on_gpu = coll_device is not None and torch.device(coll_device).type == "cuda"
    # the detector writes its ordered list and count straight into the gather buffers (modes_gpu_set_output): always on
    # the GPU; a CPU stand-in may offer the same (tests: the schedule of the RCCL path on gloo)
    inplace = on_gpu or bool(getattr(demods[0], "cpu_output", False))
    log = oplog.append if oplog is not None else (lambda item: None)
    if dist_on:
        rg = RecordGather(cap_records, device=coll_device)
        slots = [rg.slot() for _ in range(depth)]
        comms = [None] * depth
        if inplace:
            for d, s in zip(demods, slots):
                d.set_output(s.own_records, s.count)
        if on_gpu:
            # the exchanges of a call are queued on a stream of their own, after the host has seen the call's results
            # complete: no event, no order kernel in the detect's stream (they cost it ~11 us per call)
            # ONE stream for all calls: HIP multiplexes its streams onto a few hardware queues, and a communication stream
            # that lands in the queue of a launch stream puts its copies between the scans (measured with one stream per
            # context: +60 us per step on two launch streams)
            comms = [torch.cuda.Stream(device=coll_device)] * depth
    # (the resolver may keep a step's gather buffers until the step's last call arrives only if that call's transfers are
    # queued before the buffers are needed again: see the wait below)
    lag = max(1, len(works) if lag is None else int(lag))
    rr = None
    if ranks_mode:
        from .distributed import RankResolve
        if ctl_group is None:
            ctl_group = dist.new_group(backend="gloo")
        rr = RankResolve(dict(fix=flags["fix"], aggressive=flags["aggressive"], check_crc=True), resolve_threads, ctl_group,
                         device=coll_device if on_gpu else None, fresh=True)
    resolver = Resolver(dict(fix=flags["fix"], aggressive=flags["aggressive"], check_crc=True), resolve_threads,
                        hold_views=depth >= len(calls) + (1 + lag if dist_on else 1), rank_resolve=rr,
                        root=rank == 0) if (rank == 0 or ranks_mode) else None
    free = [threading.Event() for _ in range(depth)]          # the resolver is done with context k's record buffer
    for e in free:
        e.set()

    def sync_all():
        device_sync()
        if sync_ranks:
            dist.barrier()
            device_sync()

    scan_ms, demod_ms, order_ms = [], [], []
    scan_region = []                                                         # the region each timed call's sample belongs to
    region_elapsed = []
    cur_region = [0]
    regions = max(1, int(regions))
    last = {}
    host = dict(wait_kernels=0.0, queue_counts=0.0, queue_records=0.0, wait_records=0.0, detect=0.0, wait_free=0.0,
                fetch=0.0)                                                   # host seconds by phase
    comm = dict(calls=0, p2p_ops=0, bytes=0, ms=0.0)                         # the exchanges of the timed calls (this rank's view)

    def note(info, timed):
        last.update({k: v for k, v in info.items() if not k.endswith("_ms")})
        if timed and info.get("scan_ms", 0.0) > 0.0:                        # a call that carried timing events
            scan_ms.append(info["scan_ms"])
            scan_region.append(cur_region[0])
            demod_ms.append(info["demod_ms"])
            order_ms.append(info["order_ms"])

    # phase 1 of a call in flight (N > 1): wait for its kernels (the host-visible word; a long list is put in order here),
    # then queue the all_gather of the record counts - on the communication stream, with no event in the detect's stream
    def phase_counts(k, timed, n):
        d, s = demods[k], slots[k]
        t_a = time.perf_counter()
        # A list that outgrew the gather buffers must fail the JOB, not this rank alone: the other ranks are about to
        # queue this call's all_gather and would wait for it until the watchdog fires.  So the overflow is swallowed here,
        # the true count (> cap) still goes through the all_gather, and every rank raises from the shared check in
        # GatherSlot.exchange_records.
        info = {}
        if inplace:
            try:
                _, info = d.fetch_device()
            except ModesError as e:
                if e.code != -4:
                    raise                                       # (the kernels left the true count in s.count)
        else:                                                   # --backend gloo smoke mode: the lists travel as CPU tensors
            try:
                recs, _, info = d.fetch()
                n_recs = recs.size
            except ModesError as e:
                if e.code != -4:
                    raise
                n_recs = s.g.cap + 1
            if n_recs <= s.g.cap:
                s.own_records[: n_recs * 64] = torch.from_numpy(recs.view(np.uint8).reshape(-1))
            s.count[0] = n_recs
        t_b = time.perf_counter()
        log(("counts", n))
        s.exchange_counts(stream=comms[k])
        host["wait_kernels"] += t_b - t_a
        host["queue_counts"] += time.perf_counter() - t_b
        note(info, timed)

    # phase 2 (N > 1): its counts are on the host -> queue the transfers of the lists
    def phase_records(k, n):
        log(("records", n))
        t_a = time.perf_counter()
        slots[k].exchange_records(stream=comms[k])
        host["queue_records"] += time.perf_counter() - t_a

    # last phase: the records are on rank 0's host -> resolver thread
    def phase_resolve(k, tag, timed):
        counts = None
        if dist_on:
            t_a = time.perf_counter()
            recs = slots[k].wait()
            host["wait_records"] += time.perf_counter() - t_a
            counts = slots[k].counts
            if timed:
                comm["calls"] += 1
                comm["p2p_ops"] += slots[k].p2p_ops
                comm["bytes"] += slots[k].gathered_bytes()
                comm["ms"] += slots[k].comm_ms()
        else:
            t_a = time.perf_counter()
            recs, _, info = demods[k].fetch(copy=False)         # a view of the context's pinned list
            host["fetch"] += time.perf_counter() - t_a
            note(info, timed)
        if resolver is not None:
            # (gathered lists: only a step of ONE call - then the resolver would resolve the list where it stands, like a rank's own)
            if len(recs) == 0 and (counts is None or (tag[0] and tag[1])) and resolver.take_empty(tag[0], tag[1], keep_text=tag[2]):
                return                                          # nothing to resolve, nothing to print: the buffer stays free
            free[k].clear()
            resolver.submit(recs, counts, tag[0], tag[1], timed, free[k], keep_text=tag[2])

    stage = {}                                                  # context -> [phase, timed, (first call, last call of its step), call number]
    order = []                                                  # contexts with a call in flight, oldest first
    QUEUED = 2 if dist_on else 0                                # phase in which nothing is left to queue but the hand-over

    def advance(k, upto):
        """upto: 1 = counts queued, 2 = transfers queued, 3 = handed to the resolver"""
        st = stage.get(k)
        if st is None:
            return
        if dist_on and st[0] == 0 and upto >= 1:
            phase_counts(k, st[1], st[3])
            st[0] = 1
        if dist_on and st[0] == 1 and upto >= 2:
            phase_records(k, st[3])
            st[0] = 2
        if upto >= 3:
            phase_resolve(k, st[2], st[1])
            stage.pop(k)
            order.remove(k)

    def flush():
        """Everything in flight to the resolver.  Phase by phase over all calls, not call by call: the count all_gathers of all of them
        are queued before the first list transfer is, so the exchanges of the last calls overlap instead of standing behind one another at
        the end of a timed region (a 20-step region of the headline leg ends with up to three calls in flight; every rank flushes at the
        same step with the same calls in flight, so the order of the communication calls stays the same everywhere)."""
        for upto in (1, 2, 3):
            for k in list(order):
                advance(k, upto)

    t0 = None
    prof0 = []
    ncall = 0
    first_timed_call = warm * len(calls)
    timed_phase = (min(time_every, steps * len(calls)) - 1) % time_every
    t0_region = None
    # No collector pauses inside the timed regions: a pass over the interpreter's objects is milliseconds - one region of five read 0.2875 ms
    # per step next to four at 0.214-0.217 (profiles/r09/bench_line_gc.json); the loop allocates no cycles.  Collected HERE, in front of the
    # warm-up: a pause between the warm-up and the first region idles the chip and restarts its clock transient (that region then read
    # 0.245 next to 0.216-0.219).
    gc.collect()
    gc.disable()                                               # (run_steps puts it back, whatever happens in here)
    for step in range(warm + regions * steps):
        if step > warm and (step - warm) % steps == 0:          # a region ends here, the next one begins: the same bracket as at the very end
            flush()
            for e in free:
                e.wait()
            if resolver is not None:
                resolver.drain()
            sync_all()
            t_now = time.perf_counter()
            region_elapsed.append(t_now - t0_region)
            t0_region = t_now
            cur_region[0] += 1
        if step == warm:
            flush()
            for e in free:
                e.wait()
            if resolver is not None:
                resolver.drain()
            sync_all()
            prof0 = [d.host_profile() for d in demods if hasattr(d, "host_profile")]
            if rr is not None:                                  # (its thread is idle: drained above)
                rr0 = dict(steps=rr.steps, rounds=rr.rounds, reruns=rr.reruns, p2p_ops=rr.p2p_ops, bytes=rr.bytes_moved, phase=list(rr.phase_s))
            for key in host:                                    # host time by phase: of the timed calls only (the warm-up
                host[key] = 0.0                                 # holds one-off costs: list growth, RCCL's connection set-up)
            t0 = t0_region = time.perf_counter()
        timed = step >= warm
        for ci, (b0, nb, clo, chi) in enumerate(calls):
            k = ncall % depth
            if k in stage:                                      # its previous call must be complete first
                for kk in list(order):
                    advance(kk, 3)
                    if kk == k:
                        break
            # ... and resolved: the record buffer is reused.  (Rank 0 with several calls per step: the resolver releases a
            # step's buffers together, when it has the step's last call - hand it everything that is still in flight.)
            # Only calls that have nothing left to queue: the communication calls of the newer ones are issued at THEIR
            # point of the schedule (below) on every rank, never from this rank-0-only wait.  The buffer waited for never
            # depends on those: the resolver holds a step's buffers only when depth >= len(calls) + 1 + lag, and then the
            # step that last used context k ended at least 2 + lag calls ago - its transfers are queued.
            while not free[k].is_set() and order and stage[order[0]][0] >= QUEUED:
                advance(order[0], 3)
            t_a = time.perf_counter()
            free[k].wait()
            host["wait_free"] += time.perf_counter() - t_a
            stream = works[ncall % len(works)]
            # the LAST call of every group of time_every calls of the timed region carries timing events (of a region shorter
            # than that: its last call; none while warming up) - never the first call after the flush, which starts on an
            # idle chip and runs up to 30 % longer
            timed_call = timed and (ncall - first_timed_call) % time_every == timed_phase and hasattr(demods[k], "set_timing")
            if hasattr(demods[k], "set_timing"):
                demods[k].set_timing(timed_call)
            # With several launch streams the kernels of consecutive calls overlap (the next scan fills the gaps and the
            # tail of this call's latency-bound kernels) and a kernel's duration then includes the time it shared the
            # chip.  The calls that carry timing events are therefore run ALONE: they start when everything queued
            # before has finished, and nothing queued later starts before they are done - their times are the kernels'.
            alone = timed_call and len(works) > 1 and stream is not None
            if alone:
                for w in works:
                    if w is not stream:
                        stream.wait_stream(w)
            log(("detect", ncall))
            t_a = time.perf_counter()
            demods[k].detect(iq[clo - lo: chi - lo], stream_byte0=clo, first_block=b0, nblocks=nb, stream=stream)
            host["detect"] += time.perf_counter() - t_a
            if alone:
                for w in works:
                    if w is not stream:
                        demods[k].stream_wait(w)
            # (first call, last call of its step, the step's listing is wanted: the last step's - it is what the caller checks)
            stage[k] = [0, timed, (ci == 0, ci == len(calls) - 1, step == warm + regions * steps - 1), ncall]
            order.append(k)
            ncall += 1
            # keep the older calls moving - the same communication calls at the same point on every rank
            if dist_on:
                if len(order) >= 1 + lag:
                    advance(order[-1 - lag], 1)
                if len(order) >= 2 + lag:
                    advance(order[-2 - lag], 2)
                if len(order) >= 3 + lag:
                    advance(order[0], 3)
            elif len(order) >= 3:
                advance(order[0], 3)
    t_loop = time.perf_counter()
    flush()
    t_adv = time.perf_counter()
    for e in free:
        e.wait()
    if resolver is not None:
        resolver.drain()                                        # the timed region ends when the last message is out
    t_drain = time.perf_counter()
    sync_all()
    t_end = time.perf_counter()
    region_elapsed.append(t_end - t0_region)

    # where the (last) region's final moments go (ms): the calls still in flight when the loop ends, the resolver, the final sync
    tail_ms = {"in_flight": round((t_adv - t_loop) * 1e3, 4), "resolver": round((t_drain - t_adv) * 1e3, 4),
               "sync": round((t_end - t_drain) * 1e3, 4)}
    if sync_ranks:
        t = torch.tensor(region_elapsed, dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_elapsed = [float(x) for x in t.tolist()]
    elapsed = float(np.median(region_elapsed))                               # (one region: that region)
    all_steps = steps * regions
    mean = lambda v: float(np.mean(v)) if v else 0.0
    out = {"elapsed": elapsed, "elapsed_regions": region_elapsed, "regions": regions,
           "scan_ms": mean(scan_ms), "demod_ms": mean(demod_ms), "order_ms": mean(order_ms),
           "scan_ms_median": float(np.median(scan_ms)) if scan_ms else 0.0,
           # the scan kernel's average per region (what the headline's roofline.frac spread comes from)
           "scan_ms_regions": [mean([v for v, r in zip(scan_ms, scan_region) if r == g]) for g in range(regions)],
           "timed_calls": len(scan_ms), "last": dict(last), "depth": depth, "calls_per_step": len(calls),
           "host_ms_per_call": {k: round(v / max(1, all_steps * len(calls)) * 1e3, 4) for k, v in host.items()},
           "call_bytes": float(np.mean([chi - clo for _, _, clo, chi in calls])), "steps": steps, "comm": comm,
           "region_tail_ms": tail_ms}
    prof = [d.host_profile() for d in demods if hasattr(d, "host_profile")]
    if prof:                                                     # inside modes_gpu_detect, by section, microseconds per timed call
        if len(prof0) == len(prof):
            prof = [{k: p[k] - q[k] for k in p} for p, q in zip(prof, prof0)]
        ncalls_p = max(1, sum(p["calls"] for p in prof))
        out["detect_us_per_call"] = {k: round(sum(p[k] for p in prof) / ncalls_p * 1e6, 2) for k in prof[0] if k != "calls"}
    if rr is not None:
        # the protocol's facts over the timed steps (this rank's view; `comm` in the shape the record gather reports it in:
        # three host-memory all_gathers a step, the texts as point-to-point transfers)
        nst = max(1, rr.steps - rr0["steps"])
        out["rank_resolve"] = {"steps": rr.steps - rr0["steps"], "rounds_per_step": round((rr.rounds - rr0["rounds"]) / nst, 3),
                               "reruns": rr.reruns - rr0["reruns"], "threads": resolve_threads,
                               "work_ms_per_step": round(resolver.resolve_s / max(1, all_steps) * 1e3, 4),
                               "exchange_ms_per_step": round(resolver.exchange_s / max(1, all_steps) * 1e3, 4),
                               "text_bytes_per_step": int((rr.bytes_moved - rr0["bytes"]) / nst),
                               # the work by stretch between the exchanges: [guess, resolve, check, totals, end] (+ re-run rounds)
                               "work_ms_by_phase": [round((a - b) / nst * 1e3, 4) for a, b in zip(rr.phase_s, rr0["phase"])]}
        comm.update(calls=(rr.rounds - rr0["rounds"]) * 2 + (rr.steps - rr0["steps"]), p2p_ops=rr.p2p_ops - rr0["p2p_ops"], bytes=rr.bytes_moved - rr0["bytes"])
        if rank != 0:
            err = resolver.error
            resolver.stop()
            rr.close()
            if err is not None:
                raise err
    if rank == 0:
        if resolver.error is not None:
            raise resolver.error
        out.update(msgs=resolver.msgs / regions, listing=resolver.last_text, lines=resolver.last_lines)   # (messages of one region's steps)
        # rank 0's host half per STEP (its own thread, next to the launches): the resolve + --raw formatting of everything the
        # step gathered - what bounds an N-GPU step when it exceeds the kernels' time per rank
        out["host_ms_per_call"]["resolve_per_step"] = round(resolver.resolve_s / max(1, all_steps) * 1e3, 4)
        resolver.stop()
        if rr is not None:
            rr.close()
    for d in demods:
        d.close()
    return out
