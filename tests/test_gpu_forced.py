"""-m gpu: the evidence behind rows a3-a6 that used to live in builder-run logs only, where the driver runs it:
each demodulation path - one kernel (demod_variant 3), select + record (2) - FORCED for the whole differential suite of
tests/test_gpu_parity.py (MODES_GPU_DEMOD_VARIANT applies to every context created with the automatic choice, so the listing, sharding,
overflow and host-buffer tests run on that path too), and a fixed-seed slice of the fuzz: 200 random streams x 3 flag sets x both paths,
every record field, class byte, whitelist slot and preamble position against the oracle."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant,name", [("2", "select_kernel + record_kernel"), ("3", "demod_kernel")])
def test_differential_suite_on_one_forced_demod_path(variant, name):
    env = dict(os.environ, MODES_GPU_DEMOD_VARIANT=variant)
    keep = "records_and_candidates or listing_matches or published_hash or sharded or tuning or ragged or slot_overflow or every_lane or dense_capture or no_retry"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", keep], capture_output=True, timeout=900, env=env, cwd=ROOT)
    tail = p.stdout.decode()[-1500:]
    assert p.returncode == 0, (name, tail, p.stderr.decode()[-800:])
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 40 and "failed" not in tail, (name, tail)


def test_fuzz_slice_of_200_random_streams():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "7000", "200"], capture_output=True, timeout=600, cwd=ROOT)
    out = p.stdout.decode()
    assert p.returncode == 0, (out[-600:], p.stderr.decode()[-1200:])
    m = re.search(r"streams 7000\.\.7199: (\d+) records, (\d+) preamble positions, (\d+) two-bit repairs compared on 2 paths, all equal", out)
    assert m and int(m.group(1)) > 50000 and int(m.group(3)) > 500, out[-400:]
