#!/usr/bin/env python3
"""What the PATCHED REFERENCE (oracle/_ref/dump1090_dropin: the reference's own main(), reader thread and sink on the two
libraries, one 256 KiB buffer per GPU call) makes of 1 GiB of the headline noise, next to the unmodified reference and to the
C++ host on the same file.  Prints one JSON line."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import oracle as orc
from dump1090_amd import Demodulator

path = "/dev/shm/modes_dropin_1g.bin"
d = Demodulator(fix=False)
iq = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
d.synth_noise(iq, 0, seed=20260922, sigma_q16=941)
d.fill(iq[-480:], 127)
iq.cpu().numpy().tofile(path)
d.close()
del iq
env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME)
out = {"file_mib": 1024}
for name, exe in (("reference", orc.REF_BIN), ("patched_reference", os.path.join(ROOT, "oracle", "_ref", "dump1090_dropin")),
                  ("cxx_host", os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd"))):
    if not os.path.exists(exe):
        continue
    best, lines = 1e9, 0
    for _ in range(2 if name == "reference" else 4):
        t0 = time.perf_counter()
        p = subprocess.run([exe, "--ifile", path, "--raw", "--no-fix"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           env=env if name != "cxx_host" else os.environ, check=True)
        best = min(best, time.perf_counter() - t0)
        lines = p.stdout.count(b"\n")
    out[name] = {"seconds": round(best, 3), "Msamples_per_s": round((1 << 29) / best / 1e6, 1), "lines": lines}
os.remove(path)
print(json.dumps(out))
