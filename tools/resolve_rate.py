#!/usr/bin/env python3
"""Rate of the host half on the record list of BASELINE configs[2] (8 GiB, ~65,500 frames, --fix): records fetched once from
the GPU, then modes_host_resolve_raw_mt timed for several thread counts, the list tiled to the 524,000 records a
--gpus 8 step of configs[3] hands rank 0.  Prints one JSON line."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from dump1090_amd import Demodulator, HostResolver, BLOCK_STRIDE

dev = torch.device("cuda:0")
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
iq, st = bench.build_frames_shard(torch, dev, nblocks, 0, nblocks * 262144, seed=3)
d = Demodulator()
recs = []
for b0 in range(0, nblocks, 16384):
    nb = min(16384, nblocks - b0)
    lo, hi = max(0, b0 * 262144 - 476), min(iq.numel(), (b0 + nb) * 262144)
    d.detect(iq[lo:hi], stream_byte0=lo, first_block=b0, nblocks=nb)
    recs.append(d.fetch()[0].copy())
d.close()
recs = np.concatenate(recs)
big = np.tile(recs, 8)
for r in range(8):
    big["block"][r * recs.size:(r + 1) * recs.size] += r * nblocks
out = {"records_8gib": int(recs.size), "records_tiled": int(big.size), "host_cores": os.cpu_count(), "host_cpus": bench.host_cpus(), "runs": []}
for name, arr in (("8gib", recs), ("64gib", big)):
    for th in (1, 4, 8, 16, 32, 64):
        res = HostResolver()                         # one resolver, like a step loop: its text buffer is allocated once
        n0, text0 = res.raw_listing(arr, None, threads=th)       # (the listing as a Python object once: its md5 must not depend on the threads)
        md5 = hashlib.md5(text0).hexdigest()
        best = 1e9
        for _ in range(7):
            # what a step loop pays (pipeline.Resolver): the listing stays in the library - in one buffer from one thread, in the pieces
            # its threads wrote from several (modes_host_resolve_raw_pieces: no gathering copy) - and is not copied into a Python object
            t0 = time.perf_counter()
            n, _ = res.raw_listing(arr, None, threads=th, text=False)
            best = min(best, time.perf_counter() - t0)
        res.close()
        assert n == n0
        out["runs"].append({"list": name, "threads": th, "ms": round(best * 1e3, 3), "lines": int(n), "md5": md5, "ns_per_record": round(best * 1e9 / arr.size, 2),
                            "Mmsgs_per_s": round(n / best / 1e6, 1)})
print(json.dumps(out))
