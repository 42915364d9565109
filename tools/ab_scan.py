"""A/B of kernel builds on one box: python tools/ab_scan.py [--rounds R] [--workload noise|lowsnr|frames] [--demod-variant V] [--run-chunks R] lib1.so lib2.so ...
(--workload lowsnr: 1 GiB of BASELINE configs[4]'s stream, --aggressive; frames: 1 GiB of configs[2]'s, --fix; default: the
bench noise, --fix.  The same library may be named twice with different MODES_* knobs only through separate builds.)
All libraries are loaded into ONE process (each dlopen has its own namespace) and measured on the same 1 GiB of the
bench noise, R rounds interleaved (A B C A B C ...: clocks drift, and where a buffer lands in HBM moves the scan time
by +-2.5 % from process to process), detects back to back on two contexts per build (sustained clocks): 60 detects
with kernel events per turn -> mean scan / demod ms, and the counts
that must not change between builds (records, preambles) or may only grow (forwarded)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, json, hashlib, time
sys.path.insert(0, '@ROOT@'); sys.path.insert(0, '@ROOT@/tests')
import numpy as np, torch
from dump1090_amd import _native as N
from dump1090_amd import Demodulator
libs = sys.argv[5:]
rounds, workload, variant, run_chunks = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
flags = dict(fix=True, aggressive=(workload == "lowsnr"), demod_variant=variant, run_chunks=run_chunks)
if workload == "noise":
    iq = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
    g = Demodulator(fix=True)
    g.synth_noise(iq, 0, seed=20260922, sigma_q16=941)
    g.fill(iq[-480:], 127)
    g.close()
else:
    import bench
    kw = bench.LOWSNR if workload == "lowsnr" else {}
    iq, _ = bench.build_frames_shard(torch, torch.device("cuda", 0), 4096, 0, 1 << 30, seed=5 if workload == "lowsnr" else 3, **kw)
ds = []
for lib in libs:                       # every build in THIS process, on the same input buffer; two contexts each
    N._gpu = None
    N.GPU_LIB = lib
    ds.append((Demodulator(**flags), Demodulator(**flags)))
res = [dict(scan=[], demod=[], third=[]) for _ in libs]

def burst(pair, n, timing):
    """n detects back to back - the next one is queued before the previous one is fetched, so the chip stays loaded
    and at its SUSTAINED clocks (with a host round trip between kernels it boosts, and everything looks 8 percent faster)"""
    a, b = pair
    a.set_timing(timing); b.set_timing(timing)
    sc, dm, th, info, recs = [], [], [], None, None
    a.detect(iq)
    for k in range(n):
        cur, nxt = (a, b) if k % 2 == 0 else (b, a)
        nxt.detect(iq)
        recs, _, info = cur.fetch()
        sc.append(info["scan_ms"]); dm.append(info["demod_ms"]); th.append(info["order_ms"])
    (b if n % 2 == 0 else a).fetch() if False else None
    last = a if n % 2 == 0 else b
    last.fetch()
    return sc, dm, th, info, recs

for pair in ds:
    burst(pair, 120, False)
for r in range(rounds):
    for i, pair in enumerate(ds):
        burst(pair, 40, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        burst(pair, 300, False)                      # no events at all: 100 percent duty, what bench.py's step sees
        torch.cuda.synchronize(); res[i].setdefault("wall", []).append((time.perf_counter() - t0) / 300 * 1e3)
        sc, dm, th, info, recs = burst(pair, 60, True)
        res[i]["scan"].append(float(np.mean(sc[10:]))); res[i]["demod"].append(float(np.mean(dm[10:]))); res[i]["third"].append(float(np.mean(th[10:])))
        res[i].update(n_records=int(info["n_records"]), n_forwarded=int(info["n_forwarded"]), n_preambles=int(info["n_preambles"]),
                      md5=hashlib.md5(recs.tobytes()).hexdigest())
print(json.dumps(res))
'''.replace('@ROOT@', ROOT)


def main():
    args = sys.argv[1:]
    rounds, workload, variant, run_chunks = 4, "noise", 0, 0
    while args and args[0].startswith("--"):
        if args[0] == "--rounds": rounds = int(args[1])
        elif args[0] == "--workload": workload = args[1]
        elif args[0] == "--demod-variant": variant = int(args[1])
        elif args[0] == "--run-chunks": run_chunks = int(args[1])
        else: sys.exit("unknown option " + args[0])
        args = args[2:]
    libs = [os.path.abspath(a) for a in args]
    out = subprocess.run([sys.executable, "-c", CHILD, str(rounds), workload, str(variant), str(run_chunks)] + libs, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("[")]
    if not line:
        print("FAILED", out.stderr[-1500:]); sys.exit(1)
    for lib, r in zip(libs, json.loads(line[-1])):
        print("%-44s scan %s | mean %.4f  demod %.4f  third %.4f  step(no events) %s  fwd %d pre %d rec %d %s" % (os.path.basename(lib), " ".join("%.4f" % x for x in r["scan"]),
              sum(r["scan"]) / len(r["scan"]), sum(r["demod"]) / len(r["demod"]), sum(r["third"]) / len(r["third"]), " ".join("%.4f" % x for x in r["wall"]), r["n_forwarded"], r["n_preambles"], r["n_records"], r["md5"][:8]))


if __name__ == "__main__":
    main()
