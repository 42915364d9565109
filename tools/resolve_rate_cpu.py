#!/usr/bin/env python3
"""Rate of the host half WITHOUT a GPU: the record list of a configs[3]-style stream comes from the oracle (test
infrastructure), is tiled to the 557,760 records a 64 GiB step hands one resolver, and modes_host_resolve_raw_mt is timed
for several thread counts - the development loop of the lean resolve (tools/resolve_rate.py is the same measurement on
records fetched from the GPU).  Prints one JSON line; the listing's md5 is compared with the one-thread listing."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import synth
from helpers import oracle_records
from dump1090_amd import HostResolver
from dump1090_amd import _native as N

nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 512
target = int(sys.argv[2]) if len(sys.argv) > 2 else 557760
cache = "/tmp/resolve_rate_cpu_%d.npy" % nblocks
if os.path.exists(cache):
    recs = np.load(cache)
else:
    st = synth.config3_stream(3, nblocks, per=8192)          # 8 x denser than configs[3]: the tile count stays small
    recs, _ = oracle_records(st.window(0, st.nbytes), 1)
    np.save(cache, recs)
if hasattr(N, "classify_records"):
    recs = N.classify_records(recs, 1)                        # what the GPU leaves in the attempts' class bytes
reps = max(1, target // recs.size)
big = np.tile(recs, reps)
for r in range(reps):
    big["block"][r * recs.size:(r + 1) * recs.size] += r * nblocks
out = {"records": int(big.size), "host_cpus": os.cpu_count(), "runs": []}
ref_md5 = None
for th in (1, 2, 4, 8):
    res = HostResolver()
    n, text = res.raw_listing(big, None, threads=th)
    md5 = hashlib.md5(text).hexdigest()
    ref_md5 = ref_md5 or md5
    assert md5 == ref_md5, (th, md5, ref_md5)
    best = 1e9
    for _ in range(7):
        res2 = HostResolver(text_buffer=res.take_text_buffer())
        t0 = time.perf_counter()
        n, _ = res2.raw_listing(big, None, threads=th, text=False)
        best = min(best, time.perf_counter() - t0)
        res.close()
        res = res2
    res.close()
    out["runs"].append({"threads": th, "ms": round(best * 1e3, 3), "lines": int(n), "ns_per_record": round(best * 1e9 / big.size, 2)})
out["md5"] = ref_md5
print(json.dumps(out))
