#!/usr/bin/env python3
"""Streams that are DENSE in ordering survivors (the scan kernel's queue path under pressure: every lane of a wavefront pushing, more
entries from one chunk than the queue holds, passes at run ends) against the oracle: segments of periodic level patterns with two
valid preambles per 16 samples (9 + 7 apart), randomly shifted, cut, perturbed and mixed with noise; run lengths 0 (automatic), 2, 6.
    python tools/fuzz_scan_dense.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import synth
from helpers import assert_records_equal, oracle_records
from dump1090_amd import Demodulator

LEVEL = np.array([2, 0, 14, 20, 2, 16, 0, 2, 1, 0, 10, 1, 20, 1, 10, 0])        # preambles at 3 and 12 (mod 16)


def dense_stream(i):
    rng = np.random.default_rng(1000 + i)
    n = 2 * synth.DATA_LEN
    ns = n // 2
    a = np.zeros(ns, dtype=np.int64)
    pos = 0
    while pos < ns:
        ln = int(rng.integers(64, 40000))
        kind = rng.integers(0, 4)
        s = np.arange(min(ln, ns - pos))
        if kind == 0:                                                    # the pattern, shifted
            seg = LEVEL[(s + rng.integers(0, 16)) % 16] * int(rng.integers(2, 7))
        elif kind == 1:                                                  # the pattern with a few samples knocked out
            seg = LEVEL[(s + rng.integers(0, 16)) % 16] * 4
            hit = rng.random(seg.size) < rng.choice([0.002, 0.02, 0.2])
            seg = np.where(hit, rng.integers(0, 90, seg.size), seg)
        elif kind == 2:                                                  # quiet
            seg = np.zeros(s.size, dtype=np.int64)
        else:                                                            # small noise
            seg = np.abs(rng.normal(0, 3, s.size)).astype(np.int64)
        a[pos:pos + s.size] = seg
        pos += s.size
    iq = np.full(n, 127, dtype=np.uint8)
    iq[0::2] = np.clip(127 + a, 0, 255)
    iq[1::2] = np.clip(127 - (a // 3) * rng.integers(0, 2), 0, 255)
    iq[-480:] = 127
    return iq


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
demods = {rc: Demodulator(keep_candidates=True, check_crc=False, run_chunks=rc) for rc in (0, 2, 6)}
t0 = time.time()
records = preambles = 0
for i in range(first, first + count):
    data = dense_stream(i)
    iq = torch.from_numpy(data).to("cuda:0")
    want, cands = oracle_records(data, 1)
    records += want.size
    preambles += cands.size
    for rc, d in demods.items():
        d.detect(iq)
        recs, got_c, _ = d.fetch()
        assert np.array_equal(got_c, cands), (i, rc, "preamble positions")
        assert_records_equal(recs, want, ctx=(i, rc))
print("dense streams %d..%d: %d records, %d preamble positions (%.1f %% of all positions) compared at 3 run lengths, all equal; %.0f s" % (
    first, first + count - 1, records, preambles, 100.0 * preambles / (count * synth.DATA_LEN), time.time() - t0))
