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
