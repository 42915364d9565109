"""dump1090_amd/pipeline.py on CPU: the step loop bench.py runs on every rank (several calls in flight, the
record lists gathered to rank 0, the resolve on its own thread), driven here by a stand-in detector that gets
its records from the oracle - world_size 1 in-process, world_size 2 over gloo.  The listing rank 0 ends up with
must be the reference's, whatever the split into ranks and calls."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleDetector:
    """Quacks like dump1090_amd.Demodulator for run_steps' host-list mode: detect() remembers the buffers,
    fetch() returns their records (ascending, RECORD_DTYPE) from the oracle."""

    def __init__(self, data, maxfix, cpu_output=False):
        self.data, self.maxfix, self.span, self.calls = data, maxfix, None, 0
        self.cpu_output = cpu_output        # run_steps: give me the gather buffers (the schedule of the RCCL path, on gloo)
        self.out = None

    def set_output(self, records, count):
        self.out = (records, count)

    def fetch_device(self):
        recs, _, info = self.fetch()
        if self.out is not None and recs.size > self.out[0].numel() // 64:
            from dump1090_amd._native import ModesError
            raise ModesError(-4, "%d records exceed max_records=%d" % (recs.size, self.out[0].numel() // 64))
        return recs.size, info

    def detect(self, iq, stream_byte0=0, first_block=0, nblocks=None, stream=None):
        from dump1090_amd import shard_byte_range
        assert self.span is None, "one detect per context at a time"
        lo, hi = shard_byte_range(first_block, nblocks, self.data.size)
        assert (stream_byte0, len(iq)) == (lo, hi - lo), "the call got exactly the bytes its buffers need"
        assert np.array_equal(iq, self.data[lo:hi])
        self.span = (first_block, nblocks)
        self.calls += 1
        if self.out is not None:            # "the kernels" leave the ordered list and its length in the caller's buffers
            import torch
            recs = self.fetch()[0]
            self.span = (first_block, nblocks)
            fit = min(recs.size, self.out[0].numel() // 64)     # like the kernels: the true count, no store past the capacity
            self.out[0][: fit * 64] = torch.from_numpy(recs[:fit].view(np.uint8).reshape(-1).copy())
            self.out[1][0] = recs.size

    def fetch(self, copy=True):
        from helpers import oracle_records
        first, n = self.span
        self.span = None
        recs, _ = oracle_records(self.data, self.maxfix, blocks=range(first, first + n))
        return recs, None, dict(n_records=recs.size, n_forwarded=0, n_preambles=0, scan_ms=0.0, demod_ms=0.0, order_ms=0.0)

    def close(self):
        pass


def _sparse():
    """Buffers without a single record around three buffers of frames: calls of one buffer each - the step's first, its last and
    some in between hand the resolver an EMPTY list."""
    import synth
    quiet = np.full(synth.DATA_LEN, 127, dtype=np.uint8)
    return np.concatenate([quiet, synth.case_frames(), quiet, quiet])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, case, ncalls, depth, outdir, inplace=False, slow_resolver=0.0, lag=None, resolve_on="root", regions=1):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import synth
    from dump1090_amd import block_count, shard_blocks, shard_byte_range
    from dump1090_amd.pipeline import run_steps, split_calls
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    data = {"frames": synth.case_frames, "edges": synth.case_edges, "sparse": _sparse}[case]()
    total = block_count(data.size)
    first, n = shard_blocks(total - 1, world, rank)            # bench.py's sharding: the EOF buffer goes to the last rank
    if rank == world - 1:
        n += 1
    lo, hi = shard_byte_range(first, n, data.size)
    calls = split_calls(first, n, ncalls, lo, data.size)
    made = []

    def make():
        made.append(OracleDetector(data, 1, cpu_output=inplace))
        return made[-1]

    if slow_resolver:                                            # rank 0's sequential half falls behind its GPU
        import time
        from dump1090_amd import demod
        fast1, fastv, fasts = demod.HostResolver.raw_listing, demod.HostResolver.raw_listing_segments, demod.HostResolver.raw_listing_spec
        demod.HostResolver.raw_listing = lambda self, *a, **k: (time.sleep(slow_resolver), fast1(self, *a, **k))[1]
        demod.HostResolver.raw_listing_segments = lambda self, *a, **k: (time.sleep(2 * slow_resolver), fastv(self, *a, **k))[1]
        # (resolve on the ranks: one rank's resolver slower than the other's - the ranks wait for each other in the exchanges only)
        demod.HostResolver.raw_listing_spec = lambda self, *a, **k: (time.sleep(slow_resolver * (1 + rank)), fasts(self, *a, **k))[1]
    oplog = []
    out = run_steps(make, data[lo:hi], lo, calls, dict(fix=True, aggressive=False), steps=2, warm=1, depth=depth,
                    world=world, rank=rank, dist=dist, coll_device="cpu", cap_records=4096, oplog=oplog, lag=lag, resolve_on=resolve_on,
                    regions=regions)
    if regions > 1:
        # `regions` timed regions of 2 steps behind ONE warm-up step: every region has its own bracket and its own clock, `elapsed` is
        # the median; what is counted per step is counted over all of them
        assert sum(d.calls for d in made) == (1 + 2 * regions) * ncalls and out["regions"] == regions and len(out["elapsed_regions"]) == regions
        assert out["elapsed"] == float(np.median(out["elapsed_regions"])) and min(out["elapsed_regions"]) > 0
        if rank == 0:
            with open(os.path.join(outdir, "out.txt"), "wb") as f:
                f.write(out["listing"])
            assert out["msgs"] == 2 * out["lines"]               # the messages of ONE region's two steps
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    assert sum(d.calls for d in made) == 3 * ncalls and out["calls_per_step"] == ncalls
    if resolve_on == "ranks":
        # no record left its rank: the launching threads issued no communication call at all; three all_gathers a step and the texts
        assert [op for op, _ in oplog] == ["detect"] * (3 * ncalls)
        rr = out["rank_resolve"]
        assert rr["steps"] == 2 and rr["rounds_per_step"] == 1.0 and rr["reruns"] == 0
        if rank == 0:
            assert out["comm"]["bytes"] > 0 and out["comm"]["p2p_ops"] >= 2 and rr["text_bytes_per_step"] > 0
    elif world > 1:
        # every rank issued its communication calls in the same order (RCCL executes them in issue order: a rank that
        # deviates deadlocks the job) - per call n: detect(n), counts(n - 1), records(n - 2)
        logs = [None] * world
        dist.all_gather_object(logs, oplog)
        assert all(l == logs[0] for l in logs), "ranks issued their communication calls in different orders"
        at = {op: i for i, op in enumerate(oplog)}
        for n in range(3 * ncalls):
            assert at[("detect", n)] < at[("counts", n)] < at[("records", n)]
    if rank == 0:
        with open(os.path.join(outdir, "out.txt"), "wb") as f:
            f.write(out["listing"])
        assert out["msgs"] == 2 * out["lines"]                   # two timed steps, the same listing each
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _spawn2(args, deadline=180.0, nprocs=2):
    """Two ranks; a hang (ranks waiting for each other) is a failure, not a stuck test run."""
    import time
    ctx = mp.spawn(_run, args=args, nprocs=nprocs, join=False)
    t0 = time.time()
    while not ctx.join(timeout=2.0):
        if time.time() - t0 > deadline:
            for p in ctx.processes:
                p.terminate()
            pytest.fail("the ranks hung - communication calls issued in different orders?")


@pytest.mark.parametrize("case,ncalls,depth", [("frames", 1, 3), ("edges", 2, 3), ("edges", 3, 2), ("frames", 2, 1)])
def test_single_rank_pipeline(tmp_path, golden, case, ncalls, depth):
    _run(0, 1, 0, case, ncalls, depth, str(tmp_path))
    assert open(tmp_path / "out.txt").read() == golden[case]["raw"]["default"]["text"]


@pytest.mark.parametrize("depth", [1, 2, 4])
def test_calls_without_a_record_are_taken_by_the_launching_thread(tmp_path, depth):
    """Resolver.take_empty: a call whose list is empty never goes through the resolver thread when that thread has nothing queued
    (bench.py's record-free headline workload: no hand-over at all) - and goes through it like any other call when it has, so that
    the step's first / last bookkeeping keeps its order.  Seven calls of one buffer, the first, the last and one in the middle empty:
    the listing is the oracle's for the whole stream, on every depth (depth 1: the resolver is always idle when a call ends)."""
    sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    import oracle as orc
    from dump1090_amd import pipeline
    taken = []
    orig = pipeline.Resolver.take_empty
    pipeline.Resolver.take_empty = lambda self, *a, **k: (taken.append(orig(self, *a, **k)), taken[-1])[1]
    try:
        _run(0, 1, 0, "sparse", 7, depth, str(tmp_path))
    finally:
        pipeline.Resolver.take_empty = orig
    msgs, _ = orc.run_stream(_sparse(), fix=True)
    assert open(tmp_path / "out.txt").read() == orc.raw_text(msgs) and len(msgs) >= 100
    assert len(taken) >= 3 * 3 and any(taken)                  # three steps x (at least) three empty calls; some taken inline


def test_several_timed_regions_behind_one_warm_up(tmp_path, golden):
    """run_steps(regions=3) - bench.py's headline: three regions of K steps back to back, each between its own flush + barrier + sync;
    the listing is the last step's, the median region is `elapsed`.  One rank, and two over gloo (the regions' clocks are max-reduced)."""
    _run(0, 1, 0, "edges", 2, 3, str(tmp_path), regions=3)
    assert open(tmp_path / "out.txt").read() == golden["edges"]["raw"]["default"]["text"]
    os.remove(tmp_path / "out.txt")
    _spawn2((2, _free_port(), "edges", 2, 3, str(tmp_path), False, 0.0, None, "root", 3))
    assert open(tmp_path / "out.txt").read() == golden["edges"]["raw"]["default"]["text"]


@pytest.mark.parametrize("case,ncalls,depth", [("frames", 1, 3), ("edges", 2, 3), ("edges", 2, 2), ("edges", 2, 1), ("edges", 3, 4)])
def test_two_ranks_gather_over_gloo(tmp_path, golden, case, ncalls, depth):
    _spawn2((2, _free_port(), case, ncalls, depth, str(tmp_path)))
    assert open(tmp_path / "out.txt").read() == golden[case]["raw"]["default"]["text"]


@pytest.mark.parametrize("case,ncalls,depth,slow,lag", [("edges", 1, 3, 0.05, 1), ("edges", 2, 4, 0.05, 1), ("edges", 3, 4, 0.03, 1),
                                                         ("frames", 2, 2, 0.03, 1), ("edges", 3, 2, 0.0, 1),
                                                         ("edges", 2, 6, 0.04, 2), ("edges", 2, 5, 0.04, 2), ("edges", 3, 4, 0.03, 2),
                                                         ("frames", 1, 6, 0.03, 3)])
def test_two_ranks_same_communication_order(tmp_path, golden, case, ncalls, depth, slow, lag):
    """The schedule of the RCCL path (lists written in place; per call n: detect(n), counts(n - lag), transfers(n - lag - 1))
    on gloo, with rank 0's resolver slower than its detector, buffers held by the resolver (depth >= calls + 1 + lag) or
    copied: the ranks' sequences of communication calls must not diverge."""
    _spawn2((2, _free_port(), case, ncalls, depth, str(tmp_path), True, slow, lag))
    assert open(tmp_path / "out.txt").read() == golden[case]["raw"]["default"]["text"]


@pytest.mark.parametrize("world,case,ncalls,depth,slow", [(2, "frames", 1, 3, 0.0), (2, "edges", 2, 3, 0.03), (2, "edges", 3, 2, 0.0),
                                                          (3, "edges", 2, 4, 0.02), (2, "frames", 2, 1, 0.0)])
def test_ranks_resolve_their_own_records(tmp_path, golden, world, case, ncalls, depth, slow):
    """run_steps(resolve_on="ranks"): every rank fetches its own list and resolves it on its own resolver thread
    (distributed.RankResolve); rank 0 ends up with the reference's listing, whatever the split into ranks and calls, with
    buffers held until a step is resolved (depth > calls) or copied, and with resolvers of different speed."""
    _spawn2((world, _free_port(), case, ncalls, depth, str(tmp_path), False, slow, None, "ranks"), nprocs=world)
    assert open(tmp_path / "out.txt").read() == golden[case]["raw"]["default"]["text"]


def _run_overflow(rank, world, port, outdir, inplace):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import synth
    import torch.distributed as dist
    from dump1090_amd import block_count, shard_blocks, shard_byte_range
    from dump1090_amd._native import ModesError
    from dump1090_amd.pipeline import run_steps, split_calls
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    # frames only in the LAST rank's buffers: rank 1's list outgrows the gather buffers, rank 0's is empty
    data = synth.case_noise(seed=18, nblocks=6)
    data[4 * 262144: 6 * 262144] = synth.case_frames(seed=15, nblocks=2)
    total = block_count(data.size)
    first, n = shard_blocks(total - 1, world, rank)
    if rank == world - 1:
        n += 1
    lo, hi = shard_byte_range(first, n, data.size)
    calls = split_calls(first, n, 1, lo, data.size)
    verdict = "no error"
    try:
        run_steps(lambda: OracleDetector(data, 1, cpu_output=inplace), data[lo:hi], lo, calls, dict(fix=True, aggressive=False), steps=2,
                  warm=1, depth=3, world=world, rank=rank, dist=dist, coll_device="cpu", cap_records=8)
    except ModesError as e:
        verdict = "ModesError %d: %s" % (e.code, e)
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write(verdict)
    dist.destroy_process_group()


@pytest.mark.parametrize("inplace", [False, True])
def test_a_list_that_outgrows_the_gather_buffers_fails_every_rank(tmp_path, inplace):
    """One rank's record list does not fit the gather buffers: the job must END - every rank raising the same overflow
    from the shared check behind the count exchange - instead of the other ranks waiting in an all_gather the
    overflowing rank never joins (until the watchdog fires)."""
    import time
    ctx = mp.spawn(_run_overflow, args=(2, _free_port(), str(tmp_path), inplace), nprocs=2, join=False)
    t0 = time.time()
    while not ctx.join(timeout=2.0):
        if time.time() - t0 > 120:
            for p in ctx.processes:
                p.terminate()
            pytest.fail("the ranks hung on a one-sided overflow")
    for r in (0, 1):
        text = open(tmp_path / ("rank%d.txt" % r)).read()
        assert text.startswith("ModesError -4") and "rank 1 produced" in text, (r, text)
