#!/usr/bin/env python3
"""End-to-end (file -> stdout) rate of the C host, PCIe and file reading included.
Builds a <GiB> GiB synthetic stream in /dev/shm with the GPU generator, then times
    dump1090_amd/bin/dump1090_amd --ifile <file> --raw > /dev/null
and, for comparison, the compiled reference on the first GiB.  Prints one JSON line."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle as orc
from dump1090_amd import Demodulator

gib = int(sys.argv[1]) if len(sys.argv) > 1 else 4
path = "/dev/shm/modes_e2e.bin"
d = Demodulator(fix=False)
with open(path, "wb") as f:
    for k in range(gib):
        iq = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
        d.synth_noise(iq, k << 30, seed=77, sigma_q16=941)
        if k == gib - 1:
            d.fill(iq[-480:], 127)
        iq.cpu().numpy().tofile(f)
d.close()
exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
out = {"file_gib": gib}
for name, cmd in (("cli_raw", [exe, "--ifile", path, "--raw"]), ("cli_raw_again", [exe, "--ifile", path, "--raw"]),
                  ("cli_stats", [exe, "--ifile", path, "--stats"])):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, check=True)
    dt = time.perf_counter() - t0
    out[name] = {"seconds": round(dt, 3), "Msamples_per_s": round(gib * (1 << 29) / dt / 1e6, 1), "stdout_bytes": len(p.stdout)}
if orc.have_ref():
    one = "/dev/shm/modes_e2e_1g.bin"
    with open(path, "rb") as f, open(one, "wb") as g:
        buf = bytearray(f.read(1 << 30)); buf[-480:] = b"\x7f" * 480; g.write(buf)
    env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME)
    t0 = time.perf_counter()
    subprocess.run([orc.REF_BIN, "--ifile", one, "--raw"], stdout=subprocess.DEVNULL, env=env, check=True)
    dt = time.perf_counter() - t0
    out["reference_1gib"] = {"seconds": round(dt, 3), "Msamples_per_s": round((1 << 29) / dt / 1e6, 1)}
    os.remove(one)
os.remove(path)
print(json.dumps(out))
