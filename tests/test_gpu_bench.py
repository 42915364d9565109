"""-m gpu: bench.py's own code paths at small sizes - the single-rank pipeline, the N > 1 path forced onto one rank over
RCCL (process group of one: device output buffers, count all_gather, fetch_device, stream_wait), and two ranks sharing
the one GPU over gloo (the lists travel as CPU tensors).  Each run checks its gathered, resolved listing against the
analytic expectation inside bench.py (listing_check) before it prints its line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--mib", "128", "--frames-mib", "256", "--steps", "3", "--warmup", "1", "--settle", "4", "--frames-steps", "3",
         "--no-cpu-baseline", "--no-end-to-end"]


def parse(out: bytes):
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-600:]
    return json.loads(lines[0])


def check(j, n_gpus):
    assert j["n_gpus"] == n_gpus and j["value"] > 0 and j["unit"] == "Msamples/s"
    assert 0.05 < j["roofline"]["frac"] < 1.0 and j["roofline"]["bound"] == "hbm"
    f = j["frames"]
    lc = f["listing_check"]
    assert lc["missing"] == 0 and lc["lines"] >= lc["expected_frames"] > 0.9 * 256 * n_gpus * 2 ** 20 / 2 / 65536
    assert f["msgs_per_step"] == lc["lines"]
    if "one_launch_stream" in j:
        one = j["one_launch_stream"]
        k = j["kernel_ms"]
        assert one["Msamples_per_s"] > 0 and k["scan"] + k["demod"] <= one["ms_per_step"] * 1.02   # in order on one stream


@pytest.mark.parametrize("extra", [[], ["--streams", "1"], ["--force-gather"], ["--force-gather", "--overlap", "2"]])
def test_bench_small_single_rank(extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    check(parse(p.stdout), 1)


def test_bench_two_ranks_on_one_gpu_over_gloo():
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29581", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"] + SMALL,
                       capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr[-1500:]
    check(parse(p.stdout), 2)
