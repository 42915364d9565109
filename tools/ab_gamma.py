#!/usr/bin/env python3
"""The in-scan select experiment (-DSCAN_GAMMA build: every scan wavefront also runs the exact preamble predicate and the
noise-gate pre-test on its own forwarded positions at the end of its run): its own counts against the demodulation kernels' on
three streams, then its price - tools/ab_scan.py of the production and the experiment build on each stream.
    python tools/ab_gamma.py dump1090_amd/libmodes_gfx950.so dump1090_amd/libmodes_gfx950_gamma.so"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
prod, gamma = [os.path.abspath(a) for a in sys.argv[1:3]]
import torch
import bench
from dump1090_amd import _native as N
N.GPU_LIB = gamma
from dump1090_amd import Demodulator
lib = N.gpu_lib()
dev = torch.device("cuda", 0)
for name, kw, flags in (("noise", None, dict(fix=True)), ("frames", {}, dict(fix=True)), ("lowsnr", bench.LOWSNR, dict(fix=True, aggressive=True))):
    if kw is None:
        iq = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        g = Demodulator(fix=True); g.synth_noise(iq, 0, seed=20260922, sigma_q16=941); g.fill(iq[-480:], 127); g.close()
    else:
        iq, _ = bench.build_frames_shard(torch, dev, 4096, 0, 1 << 30, seed=5 if name == "lowsnr" else 3, **kw)
    d = Demodulator(**flags)
    d.detect(iq)
    recs, _, info = d.fetch()
    pre, surv = C.c_ulonglong(), C.c_ulonglong()
    assert lib.modes_gpu_gamma_totals(C.byref(pre), C.byref(surv)) == 0
    print("%-7s demodulation kernels: %d preambles, %d records | counted inside the scan (all runs but the guarded first / last ones): %d preambles, %d gate survivors" % (
        name, info["n_preambles"], info["n_records"], pre.value, surv.value), flush=True)
    d.close()
    del iq
    torch.cuda.empty_cache()
for wl, dv in (("noise", 0), ("lowsnr", 2), ("frames", 2)):
    print("== tools/ab_scan.py --workload %s" % wl, flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_scan.py"), "--rounds", "3", "--workload", wl, "--demod-variant", str(dv), prod, gamma])
