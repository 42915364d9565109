#!/usr/bin/env python3
"""The C host's ways to N GPUs on ONE device (file -> stdout, wall clock of the whole process and the host's own --timing line):
one process, --ranks N over RCCL (records to rank 0; N = 1 is all a one-GPU box allows: RCCL refuses two ranks on one device),
--ranks N --resolve-on-ranks (no communicator: the ranks may share the device, N = 1, 2, 4).  What one GPU can show: the start-up
each mode puts in front of the first byte, and that N processes feeding one device cost nothing against one.  A <GiB> GiB synthetic
stream in /dev/shm (noise + a DF17 frame every 65,536 samples of the first GiB).  Prints one JSON line."""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import synth as sy
from dump1090_amd import Demodulator

gib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
path = "/dev/shm/modes_e2e_ranks.bin"
d = Demodulator(fix=False)
with open(path, "wb") as f:
    for k in range(gib):
        iq = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
        d.synth_noise(iq, k << 30, seed=78, sigma_q16=941)
        if k == gib - 1:
            d.fill(iq[-480:], 127)
        h = iq.cpu().numpy()
        if k == 0:
            for i in range(1, 8192):
                sy.add_frame(h, i * 65536 + 1234, sy.make_frame(17, sy._payload(5, 14, i)), 70, i)
        h.tofile(f)
d.close()
del iq
torch.cuda.empty_cache()
exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
out = {"file_gib": gib, "host_cores": os.cpu_count(), "runs": []}
env = dict(os.environ, MODES_RANKS_QUIET="1")


def run(extra):
    time.sleep(1.0)
    t0 = time.perf_counter()
    p = subprocess.run([exe, "--ifile", path, "--raw", "--timing"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, env=env)
    dt = time.perf_counter() - t0
    tim = {}
    for ln in p.stderr.decode().splitlines():
        if ln.startswith("{"):
            tim = json.loads(ln)
    return {"args": " ".join(extra) or "(one process)", "wall_s": round(dt, 3), "wall_GBps": round(gib * 2**30 / dt / 1e9, 2), "init_s": tim.get("init_s"),
            "stream_s": tim.get("stream_s"), "stream_GBps": tim.get("stream_GBps"), "lines": p.stdout.count(b"\n"), "md5": hashlib.md5(p.stdout).hexdigest()}


run([])
for extra in ([], ["--gpu-list", "0,0"], ["--ranks", "1"], ["--ranks", "1", "--resolve-on-ranks"], ["--ranks", "2", "--gpu-list", "0,0", "--resolve-on-ranks"],
              ["--ranks", "4", "--gpu-list", "0,0,0,0", "--resolve-on-ranks"], ["--ranks", "1"], ["--ranks", "2", "--gpu-list", "0,0", "--resolve-on-ranks"], []):
    out["runs"].append(run(extra))
out["same_listing"] = len({r["md5"] for r in out["runs"]}) == 1
os.remove(path)
print(json.dumps(out))
