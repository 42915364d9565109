"""bench.py's own checking code, on CPU: the analytic expectation of the frames leg (which frames a rank's buffers must
contribute) and the listing check rank 0 runs before the leg's numbers are printed - against the oracle's listing of a small
stream of the same generator, whole and split into ranks as `bench.py --gpus N` splits it; and that the check does fail on
a listing that lost lines or their order."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import synth  # noqa: E402
from dump1090_amd import HostResolver, block_count, shard_blocks, shard_byte_range  # noqa: E402
from helpers import oracle_records  # noqa: E402


@pytest.fixture(scope="module")
def small():
    # 40 buffers, a frame per 32,768 samples, every 5th at a buffer seam (offsets that are never tested included); seed
    # and spacing chosen so that no seam frame lands in a neighbour's skip window (the reference would drop it)
    st = synth.config3_stream(7, 40, per=32768, edge_every=5)
    data = st.window(0, st.nbytes)
    recs, _ = oracle_records(data, 1)
    r = HostResolver(fix=True)
    n, text = r.raw_listing(recs, None)
    r.close()
    return st, data, text


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_expectation_of_the_ranks_is_what_the_stream_decodes_to(small, world):
    st, data, text = small
    total = block_count(st.nbytes)
    expected = []
    for rank in range(world):
        first, n = shard_blocks(total - 1, world, rank)          # bench.py's shard(): the EOF buffer goes to the last rank
        if rank == world - 1:
            n += 1
        lo, hi = shard_byte_range(first, n, st.nbytes)
        # the rank builds its part of the stream from its own frames only: same bytes as the whole stream's
        mine = synth.config3_stream(7, 40, per=32768, edge_every=5, only_samples=(lo // 2, (hi + 1) // 2))
        assert np.array_equal(mine.window(lo, hi), data[lo:hi]), (world, rank)
        expected += bench.frames_expectation(mine, first, n)
    res = bench.check_listing(text, expected)
    assert res["missing"] == 0 and res["lines"] >= res["expected_frames"] > 140
    whole = bench.frames_expectation(st, 0, total)
    assert expected == whole                                     # ranks own disjoint, ascending buffer ranges


def test_listing_check_rejects_a_damaged_listing(small):
    st, data, text = small
    expected = bench.frames_expectation(st, 0, block_count(st.nbytes))
    lines = text.decode().split()
    join = lambda ls: ("\n".join(ls) + "\n").encode()
    bench.check_listing(join(lines), expected)
    with pytest.raises(AssertionError, match="not in the listing|lines for"):
        bench.check_listing(join(lines[: len(lines) // 2]), expected)           # half of the stream is missing
    with pytest.raises(AssertionError, match="stream order"):
        half = len(lines) // 2
        bench.check_listing(join(lines[half:] + lines[:half]), expected)        # two ranks' lists swapped
    with pytest.raises(AssertionError, match="lines for"):
        bench.check_listing(join(lines + lines), expected)                      # every list delivered twice


def test_listing_check_rejects_lines_that_are_no_frame_of_the_stream(small):
    st, data, text = small
    expected = bench.frames_expectation(st, 0, block_count(st.nbytes))
    lines = text.decode().split()
    join = lambda ls: ("\n".join(ls) + "\n").encode()
    ok = bench.check_listing(join(lines), expected)
    assert ok["spurious"] <= 2 and ok["missing"] == 0
    junk = ["*5d%06x%06x;" % (k, k * 7919) for k in range(3)]
    with pytest.raises(AssertionError, match="no frame of the stream"):
        bench.check_listing(join(lines[:50] + junk + lines[50:]), expected)         # three invented messages: inside every other bound
    bench.check_listing(join(lines[:50] + junk[:2] + lines[50:]), expected)          # a noise-born message or two do occur


def test_launcher_command_is_the_documented_one():
    cmd = bench.launcher_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    at = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[at + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    auto = bench.launcher_command([], 2)
    assert 1024 < int(auto[auto.index("--master-port") + 1]) < 65536


def test_self_launch_propagates_the_ranks_status(tmp_path):
    """bench.py --gpus 2 without WORLD_SIZE re-runs itself under torch.distributed.run; here (no GPU) both ranks fail their
    `needs a GPU` assertion at once - the parent must end non-zero and print no line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_bench.py runs the launcher for real")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"], capture_output=True,
                       timeout=300, env=env)
    assert p.returncode != 0
    assert b"needs a GPU" in p.stderr and not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


def test_committed_reference_listings_cover_the_default_runs():
    """Every stream a default `bench.py --gpus N` (N = 1, 2, 4, 8) resolves has the reference's own verdict committed."""
    for n in (1, 2, 4, 8):
        g = bench.golden_listing("frames", 3 if n == 1 else 4, 32768 * n)
        assert g and len(g["md5"]) == 32 and g["lines"] > 65000 * n
        low = bench.golden_listing("lowsnr", 5, 4096 * n)
        assert low and low["lines"] > 2000 * n and "--aggressive" in low["flags"]
    assert bench.golden_listing("frames", 4, 262144)["lines"] > 520000          # the strong-scaling leg's stream at every N
    assert bench.golden_listing("frames", 3, 32768)["md5"] == "71d130a9532b26ccef50568cc9966b53"   # = tests/golden/config2_listing.json


def test_fetch_size_rows_become_bytes_per_launch(tmp_path):
    """bench.py's own PMC pass: the scan kernel's FETCH_SIZE rows (KB) of rocprofv3's counter_collection CSV, averaged over
    the launches and doubled (gfx950: 128-byte requests of 16 B-per-lane reads count as 64 bytes); other kernels and other
    counters do not enter; no row, no number."""
    import bench
    d = tmp_path / "pass" / "box"
    d.mkdir(parents=True)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,"
            "Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    def row(kernel, counter, value):
        return '1,1,4,1,10,10,4194304,7,"%s",128,12800,0,56,0,112,%s,%s,100,200' % (kernel, counter, value)
    rows += [row("(anonymous namespace)::scan_kernel((anonymous namespace)::ScanParams)", "FETCH_SIZE", v) for v in (536000.0, 537000.0)]
    rows += [row("void (anonymous namespace)::demod_kernel<8, LutFull>(DemodParams)", "FETCH_SIZE", 147000.0),
             row("(anonymous namespace)::scan_kernel((anonymous namespace)::ScanParams)", "WRITE_SIZE", 2400.0)]
    (d / "fetch_counter_collection.csv").write_text("\n".join(rows) + "\n")
    assert bench.fetch_bytes_per_launch(str(tmp_path)) == (int(536500.0 * 1024 * 2), 2)
    assert bench.fetch_bytes_per_launch(str(tmp_path), kernel="demod_kernel") == (int(147000.0 * 2048), 1)
    assert bench.fetch_bytes_per_launch(str(tmp_path), kernel="order_kernel") is None


def test_weak_listing_check_still_rejects_foreign_lines():
    """The low-SNR leg's check (most injected frames are undecodable: `missing` is unbounded) keeps the other two bounds: what is
    listed must be frames of the stream (1 % + 2 may not be) and in stream order."""
    import oracle as orc
    st = synth.config3_stream(5, 96, **bench.LOWSNR)
    msgs, _ = orc.run_stream(st.window(0, st.nbytes), **orc.FLAGSETS["aggressive"])
    lines = orc.raw_text(msgs).split()
    expected = bench.frames_expectation(st, 0, 97)
    join = lambda ls: ("\n".join(ls) + "\n").encode()
    ok = bench.check_listing(join(lines), expected, weak=True)
    assert ok["lines"] > 40 and ok["spurious"] <= 1 and ok["missing"] > ok["lines"]
    junk = ["*5d%06x%06x;" % (k, k * 7919) for k in range(3)]
    with pytest.raises(AssertionError, match="no frame of the stream"):
        bench.check_listing(join(lines[:20] + junk + lines[20:]), expected, weak=True)
    with pytest.raises(AssertionError, match="stream order"):
        half = len(lines) // 2
        bench.check_listing(join(lines[half:] + lines[:half]), expected, weak=True)


def test_committed_counter_pass_belongs_to_these_kernel_sources():
    """profiles/traffic_latest.json (the committed rocprofv3 FETCH_SIZE pass bench.py falls back on, and reports next to its live
    pass as roofline.committed_traffic) is stamped with the hash of the kernel sources it was measured on: a kernel edit without
    a new pass would silently turn the field into null."""
    got, note = bench.measured_traffic(1024)
    assert got is not None, note
    assert 1.0 < got / 2 ** 30 < 1.08, (got, note)                 # the scan kernel reads every sample once (+ look-backs)
    assert bench.TRACE_AVG_MS and 0.15 < bench.TRACE_AVG_MS < 0.25
