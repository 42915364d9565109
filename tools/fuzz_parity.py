#!/usr/bin/env python3
"""One-off wider run of tests/test_gpu_parity.py::test_randomized_streams_match_oracle: streams [first, first + count) of the
same seeded generator, both demodulation paths, three flag sets, every record and preamble position against the oracle.
    python tools/fuzz_parity.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import oracle as orc
from helpers import assert_records_equal, maxfix_of, oracle_records
from test_gpu_parity import random_stream
from dump1090_amd import Demodulator

first = int(sys.argv[1]) if len(sys.argv) > 1 else 64
count = int(sys.argv[2]) if len(sys.argv) > 2 else 512
names = ("default", "aggressive", "nofix")
demods = {(v, n): Demodulator(keep_candidates=True, demod_variant=v, **orc.FLAGSETS[n]) for v in (2, 3) for n in names}
t0 = time.time()
records = preambles = repaired2 = 0
for i in range(first, first + count):
    data, kw = random_stream(i)
    iq = torch.from_numpy(data).to("cuda:0")
    for n in names:
        want, cands = oracle_records(data, maxfix_of(orc.FLAGSETS[n]))
        records += want.size
        preambles += cands.size
        repaired2 += int((want["att"]["nfix"] == 2).sum())
        for v in (2, 3):
            d = demods[(v, n)]
            d.detect(iq)
            recs, got_c, _ = d.fetch()
            assert np.array_equal(got_c, cands), (i, kw, n, v)
            assert_records_equal(recs, want, ctx=(i, kw, n, v))
print("streams %d..%d: %d records, %d preamble positions, %d two-bit repairs compared on 2 paths, all equal; %.0f s" % (
    first, first + count - 1, records, preambles, repaired2, time.time() - t0))
