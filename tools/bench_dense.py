#!/usr/bin/env python3
"""Informational: the reference's own (unnaturally message-dense, --snip'ed) capture tiled to 1 GiB."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import synth
from dump1090_amd import Demodulator, HostResolver
one = synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin"))
reps = (1 << 30) // one.size
iq = torch.from_numpy(np.tile(one, reps)).to("cuda:0")
for flags in (dict(), dict(aggressive=True)):
    d = Demodulator(**flags)
    out = {}
    for rep in range(4):
        t0 = time.perf_counter()
        d.detect(iq); recs, cands, info = d.fetch(copy=False)
        t1 = time.perf_counter()
        r = HostResolver(**flags); n, text1 = r.raw_listing(recs, None); r.close()
        t2 = time.perf_counter()
        r = HostResolver(**flags); n16, text16 = r.raw_listing(recs, None, threads=16); r.close()
        t3 = time.perf_counter()
        assert (n16, text16) == (n, text1), "the 16-thread resolve differs from the sequential one"
        out = {"flags": flags, "gib": iq.numel() / 2**30, "records": len(recs), "messages": n, "scan_ms": round(info["scan_ms"], 3),
               "demod_ms": round(info["demod_ms"], 3), "order_ms": round(info["order_ms"], 3), "gpu_call_s": round(t1 - t0, 4), "resolve_s": round(t2 - t1, 4), "resolve_16_threads_s": round(t3 - t2, 4),
               "preambles": info["n_preambles"], "forwarded": info["n_forwarded"]}
    print(json.dumps(out), flush=True)
    d.close()
