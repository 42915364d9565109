"""N > 1 host path on CPU: world_size-2/3 gloo processes shard the buffers, gather the records to
rank 0 (dump1090_amd.distributed) and rank 0 resolves - the listing must equal the reference's.
The per-rank records come from the oracle here (no GPU in this suite); on the GPU box the same code
path is fed by libmodes_gfx950.so (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, fs, outdir):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle as orc
    import synth
    from dump1090_amd import HostResolver, block_count, raw_text, shard_blocks
    from dump1090_amd.distributed import gather_records
    from helpers import maxfix_of, oracle_records

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    makers = {"frames": synth.case_frames, "edges_smear": lambda: synth.case_edges(seed=23, smear16=6),
              "modes1": lambda: synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin"))}
    data = makers[case]()
    flags = orc.FLAGSETS[fs]
    first, n = shard_blocks(block_count(data.size), world, rank)
    recs, cands = oracle_records(data, maxfix_of(flags), blocks=range(first, first + n))
    # the product path of bench.py --gpus N: device-resident lists (CPU tensors under gloo), counts first, then
    # exact-size point-to-point transfers into rank 0's contiguous buffer - two calls in flight, like the bench
    import torch
    from dump1090_amd.distributed import RecordGather
    g = RecordGather(cap_records=4096, device="cpu")        # the same capacity on every rank
    slots = [g.slot(), g.slot()]
    for k, s in enumerate(slots):                         # the second call carries only every other record
        mine = np.ascontiguousarray(recs if k == 0 else recs[::2])
        s.own_records[: mine.size * 64] = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy())
        s.count[0] = mine.size
        s.exchange_counts()
    for s in slots:
        s.exchange_records()
    got = [s.wait() for s in slots]
    _, cands = gather_records(recs[:0], cands, dst=0)
    if rank == 0:
        assert got[1].size == sum((c + 1) // 2 for c in slots[0].counts)
        recs = got[0].copy()
    else:
        assert got == [None, None]
        recs = None
    if rank == 0:
        r = HostResolver(**flags)
        text = raw_text(r.resolve(recs, cands))
        with open(os.path.join(outdir, "out.txt"), "w") as f:
            f.write(text)
        with open(os.path.join(outdir, "stats.txt"), "w") as f:
            f.write(r.stats_text())
    else:
        assert recs is None and cands is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case,fs", [(2, "frames", "default"), (3, "edges_smear", "aggressive_nocrc"),
                                           (2, "modes1", "default")])
def test_sharded_gather_resolve(tmp_path, golden, world, case, fs):
    mp.spawn(_worker, args=(world, _free_port(), case, fs, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "out.txt").read() == golden[case]["raw"][fs]["text"]
    if fs in golden[case]["stats"]:
        assert open(tmp_path / "stats.txt").read() == golden[case]["stats"][fs]["text"]


def test_gather_arrays_empty_and_ragged(tmp_path):
    mp.spawn(_ragged_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    got = np.load(tmp_path / "g.npy")
    assert np.array_equal(got, np.array([10, 11, 12, 30], dtype=np.uint64))


def _ragged_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dump1090_amd.distributed import gather_arrays
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    mine = {0: [10, 11, 12], 1: [], 2: [30]}[rank]
    out = gather_arrays(np.array(mine, dtype=np.uint64), dst=0)
    empty = gather_arrays(np.zeros(0, dtype=np.uint64), dst=0)
    if rank == 0:
        assert empty.size == 0
        np.save(os.path.join(outdir, "g.npy"), out)
    dist.barrier()
    dist.destroy_process_group()


def _rank_resolve_worker(rank, world, port, case, fs, outdir):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle as orc
    import synth
    from dump1090_amd import HostResolver, block_count, shard_blocks
    from dump1090_amd.distributed import RankResolve, gather_records
    from helpers import maxfix_of, oracle_records

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ctl = dist.new_group(backend="gloo")                  # host memory: the protocol's small all_gathers (and, on CPU, the texts)
    makers = {"frames": synth.case_frames, "edges_smear": lambda: synth.case_edges(seed=23, smear16=6),
              "modes1": lambda: synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin"))}
    data = makers[case]()
    flags = orc.FLAGSETS[fs]
    first, n = shard_blocks(block_count(data.size), world, rank)
    recs, _ = oracle_records(data, maxfix_of(flags), blocks=range(first, first + n))
    # the rank's records as its GPU calls leave them: two arrays of whole buffers
    cut = int(np.searchsorted(recs["block"], first + n // 2))
    segs = [recs[:cut].copy(), recs[cut:].copy()]
    rr = RankResolve(flags, threads=1, ctl=ctl, fresh=False)
    out1 = rr.step(segs)
    text1 = b"".join(t.tobytes() for t in out1["texts"]) if rank == 0 else None
    out2 = rr.step(segs, spoil=rank > 0)                  # the same records again from the whitelist the first step left; bad starts
    text2 = b"".join(t.tobytes() for t in out2["texts"]) if rank == 0 else None
    assert (out1["texts"] is None) == (rank != 0)
    everything, _ = gather_records(recs, None, dst=0)
    if rank == 0:
        seq = HostResolver(**flags)
        want1 = seq.raw_listing(everything, None)
        want2 = seq.raw_listing(everything, None)
        st = seq.stats()
        assert (out1["lines"], text1) == want1 and (out2["lines"], text2) == want2
        assert {k: out1["stats"][k] + out2["stats"][k] for k in st if k != "valid_preamble"} == {k: v for k, v in st.items() if k != "valid_preamble"}
        wl = seq.whitelist()
        assert np.array_equal(rr.state[0], wl[0]) and np.array_equal(rr.state[1], wl[1])
        with open(os.path.join(outdir, "out.txt"), "wb") as f:
            f.write(text1)
        with open(os.path.join(outdir, "facts.txt"), "w") as f:
            f.write("%d %d %d" % (rr.steps, rr.rounds, rr.p2p_ops))
    rr.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case,fs", [(2, "frames", "default"), (3, "edges_smear", "aggressive_nocrc"), (3, "modes1", "default")])
def test_resolve_on_the_ranks_over_gloo(tmp_path, golden, world, case, fs):
    """RankResolve: every rank resolves the records of its own buffers, three small all_gathers and the texts travel - the
    listing rank 0 ends up with is the reference's, over two steps that share the whitelist, the second from spoiled starts."""
    mp.spawn(_rank_resolve_worker, args=(world, _free_port(), case, fs, str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "out.txt").read() == golden[case]["raw"][fs]["text"]
    steps, rounds, p2p = [int(v) for v in open(tmp_path / "facts.txt").read().split()]
    assert steps == 2 and rounds >= 2 and p2p >= 1


def _failing_rank_worker(rank, world, port, where, outdir):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import time
    import torch.distributed as dist
    import oracle as orc
    import synth
    from dump1090_amd import block_count, shard_blocks
    from dump1090_amd import demod
    from dump1090_amd._native import ModesError
    from dump1090_amd.distributed import RankResolve
    from helpers import oracle_records

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ctl = dist.new_group(backend="gloo")
    data = synth.case_frames()
    flags = orc.FLAGSETS["default"]
    first, n = shard_blocks(block_count(data.size), world, rank)
    recs, _ = oracle_records(data, 1, blocks=range(first, first + n))
    if rank == 1:                                         # this rank's own work fails: before the first exchange / between two of them
        def boom(self, *a, **k):
            raise RuntimeError("rank 1's resolve failed on purpose")
        if where == "guess":
            demod.HostResolver.whitelist_guess = boom
        else:
            demod.HostResolver.raw_listing_spec = boom
    rr = RankResolve(flags, threads=1, ctl=ctl, fresh=True)
    t0 = time.time()
    try:
        rr.step([recs])
        verdict = "no error"
    except RuntimeError as e:                             # (ModesError is a RuntimeError)
        verdict = "%s: %s" % ("ModesError" if isinstance(e, ModesError) else "RuntimeError", str(e)[:60])
    with open(os.path.join(outdir, "rank%d.txt" % rank), "w") as f:
        f.write("%.1f %s" % (time.time() - t0, verdict))
    # (no barrier: a rank that failed is not expected to take part in anything else)


@pytest.mark.parametrize("where", ["guess", "resolve"])
def test_a_rank_that_fails_inside_the_step_stops_every_rank_at_the_next_exchange(tmp_path, where):
    """an exception on one rank's resolver inside RankResolve.step used to leave its peers in an all_gather it never joined
    until the group's 300 s timeout.  Every gather carries a "this rank is well" word now: the failing rank still takes part in the next
    one, with the word down, raises its own error, and every other rank raises a ModesError there and then - within seconds."""
    ctx = mp.spawn(_failing_rank_worker, args=(3, _free_port(), where, str(tmp_path)), nprocs=3, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=2.0):
        assert time.time() - t0 < 60, "the ranks hung"
    out = [open(tmp_path / ("rank%d.txt" % r)).read() for r in range(3)]
    assert out[1].split(" ", 1)[1].startswith("RuntimeError: rank 1's resolve failed on purpose"), out
    for r in (0, 2):
        secs, verdict = out[r].split(" ", 1)
        assert float(secs) < 20 and verdict.startswith("ModesError") and "rank(s) [1] failed" in verdict, out
