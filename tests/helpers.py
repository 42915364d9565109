"""Shared test helpers.  `oracle_records` turns the oracle's stateless per-position records into
the product's RECORD_DTYPE so the host half (libmodes_host.so) can be tested on a CPU, and so GPU
records can be compared field by field."""
import numpy as np

import oracle as orc
from dump1090_amd import RECORD_DTYPE, BLOCK_STRIDE, _native as N


def oracle_records(data: np.ndarray, maxfix: int, blocks=None):
    """-> (records with att[0].gate_ok (RECORD_DTYPE, ascending), all preamble positions as framed g)."""
    recs, cands = [], []
    blocks = range(orc.block_count(data.size)) if blocks is None else blocks
    for k in blocks:
        mag = orc.block_magnitude(data, k)
        js = orc.block_candidates(mag)
        o = orc.records(mag, js, maxfix)
        cands.append(js.astype(np.uint64) + np.uint64(k * BLOCK_STRIDE))
        keep = o[o["att"]["gate_ok"][:, 0] == 1]
        r = np.zeros(keep.size, dtype=RECORD_DTYPE)
        r["block"] = k
        r["j"] = keep["j"]
        for f in ("msg", "errors", "gate_ok", "nfix", "fixpos", "syndrome"):
            r["att"][f] = keep["att"][f]
        # the product leaves syndrome / repair of a gate-failed retry at zero
        bad = r["att"]["gate_ok"][:, 1] == 0
        r["att"]["syndrome"][bad, 1] = 0
        r["att"]["nfix"][bad, 1] = 0
        r["att"]["fixpos"][bad, 1] = 0xFF
        recs.append(r)
    return (np.concatenate(recs) if recs else np.zeros(0, dtype=RECORD_DTYPE),
            np.concatenate(cands) if cands else np.zeros(0, dtype=np.uint64))


def maxfix_of(flags) -> int:
    return 0 if not flags["fix"] else (2 if flags["aggressive"] else 1)


def assert_records_equal(got: np.ndarray, want: np.ndarray, ctx=""):
    assert got.size == want.size, "%s: %d records, oracle has %d" % (ctx, got.size, want.size)
    assert np.array_equal(got["block"], want["block"]) and np.array_equal(got["j"], want["j"]), ctx
    for a in (0, 1):
        gate = want["att"]["gate_ok"][:, a] == 1
        assert np.array_equal(got["att"]["gate_ok"][:, a], want["att"]["gate_ok"][:, a]), (ctx, a)
        if a == 1:
            # a retry whose gate fails is never looked at (dump1090.c:1723): compare only live ones
            sel = gate
        else:
            sel = np.ones(want.size, dtype=bool)
        for f in ("msg", "errors", "nfix", "fixpos", "syndrome"):
            assert np.array_equal(got["att"][f][sel, a], want["att"][f][sel, a]), (ctx, a, f)
    if got.size and got["att"]["cls"].any():
        # the class byte and whitelist slot the kernels leave (include/modes_gfx950.h MODES_CLS_*): every attempt carries them, all
        # for one repair policy, and they are what modes_classify makes of the ORACLE's attempt for that policy
        c = got["att"]["cls"]
        assert (c & 0x80).all() and len(set((c & 0x60).reshape(-1).tolist())) == 1, (ctx, "class bytes: valid, one policy")
        fix, aggressive = bool(c[0, 0] & 0x20), bool(c[0, 0] & 0x40)
        ref = N.classify_records(want, fix=fix, aggressive=aggressive)
        for a in (0, 1):
            live = want["att"]["gate_ok"][:, a] == 1
            assert np.array_equal(c[live, a], ref["att"]["cls"][live, a]), (ctx, a, "cls")
            assert np.array_equal(got["att"]["slot"][live, a], ref["att"]["slot"][live, a]), (ctx, a, "slot")
            assert ((c[~live, a] & 7) == 1).all(), (ctx, a, "a failed gate is class GATE")
