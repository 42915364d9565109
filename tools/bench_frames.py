#!/usr/bin/env python3
"""Informational timing of BASELINE configs[2] (frames + --fix) and [4] (low SNR, --aggressive) on one GPU:
kernel times, host resolve time and messages per second.  bench.py stays the headline (configs[1])."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch
import synth
from dump1090_amd import Demodulator, HostResolver, block_count, shard_byte_range
from test_gpu_fullsize import build_on_device

def run(name, st, batch_blocks, **flags):
    d = Demodulator(**flags)
    iq = build_on_device(torch, d, st)
    total = block_count(st.nbytes)
    out = {}
    for rep in range(3):
        res = HostResolver(**flags)
        t_gpu = t_res = 0.0
        scan = demod = 0.0
        nmsg = nrec = 0
        t0 = time.perf_counter()
        for b0 in range(0, total, batch_blocks):
            nb = min(batch_blocks, total - b0)
            lo, hi = shard_byte_range(b0, nb, st.nbytes)
            a = time.perf_counter()
            d.detect(iq[lo:hi], stream_byte0=lo, first_block=b0, nblocks=nb)
            recs, cands, info = d.fetch()
            b = time.perf_counter()
            nmsg += res.count(recs, cands)
            c = time.perf_counter()
            t_gpu += b - a; t_res += c - b; scan += info["scan_ms"]; demod += info["demod_ms"]; nrec += len(recs)
        wall = time.perf_counter() - t0
        res.close()
        out = {"config": name, "gib": st.nbytes / 2**30, "frames": len(st.placements), "messages": nmsg, "records": nrec,
               "wall_s": round(wall, 4), "gpu_calls_s": round(t_gpu, 4), "resolve_s": round(t_res, 4),
               "scan_ms": round(scan, 3), "demod_ms": round(demod, 3),
               "Msamples_per_s": round(st.nbytes / 2 / wall / 1e6, 1), "msgs_per_s": round(nmsg / wall, 1)}
    d.close()
    del iq
    torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)

run("configs[2]: 8 GiB, DF11/DF17 frames, --fix", synth.config3_stream(3, 32768), 12288)
run("configs[4]: 1 GiB, low SNR, --aggressive", synth.config3_stream(5, 4096, per=16384, amp=(8, 15), smear=(3, 4, 5, 6), flip1=10, flip2=20, edge_every=61), 4096, aggressive=True)
