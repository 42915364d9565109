#!/usr/bin/env python3
"""End-to-end (file -> stdout) rate of the C host, PCIe and file reading included.
Builds a <GiB> GiB synthetic stream in /dev/shm with the GPU generator, then times
    dump1090_amd/bin/dump1090_amd --ifile <file> --raw --timing > /dev/null
for a few reader-thread / batch-size / lane settings (wall clock of the whole process AND the stream time the
host reports itself: first read -> last message, i.e. without HIP start-up and the allocation of the pinned
buffers), and, for comparison, the compiled reference on the first GiB.  Prints one JSON line."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle as orc
from dump1090_amd import Demodulator

gib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
del iq
torch.cuda.empty_cache()
exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
out = {"file_gib": gib, "host_cores": os.cpu_count(), "runs": []}


def run(extra, mode="--raw"):
    time.sleep(1.0)          # the kernel is still tearing the previous process down (its GPU queues, pinned pages) when subprocess.run returns:
                             # a process started right behind it waits ~0.1 s longer for the HIP runtime to come up
    t0 = time.perf_counter()
    p = subprocess.run([exe, "--ifile", path, mode, "--no-fix", "--timing"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    dt = time.perf_counter() - t0
    tim = {}
    for ln in p.stderr.decode().splitlines():
        if ln.startswith("{"):
            tim = json.loads(ln)
    return {"args": " ".join([mode] + extra), "wall_s": round(dt, 3), "wall_GBps": round(gib * 2**30 / dt / 1e9, 2),
            "stream_s": tim.get("stream_s"), "stream_GBps": tim.get("stream_GBps"), "init_s": tim.get("init_s"), "total_s": tim.get("total_s"),
            "init": tim.get("init"), "teardown": tim.get("teardown"), "create_trace": [ln for ln in p.stderr.decode().splitlines() if ln.startswith("modes_gpu_create:")][:1],
            "stdout_bytes": len(p.stdout)}


run([])                                                     # page cache / driver warm-up, not reported
for extra in ([], ["--read-threads", "8"], ["--read-threads", "32"], ["--no-mmap"], ["--no-mmap", "--read-threads", "8"],
              ["--no-mmap", "--read-threads", "32"], ["--batch-blocks", "256"], ["--batch-blocks", "1024"], ["--depth", "2"],
              ["--depth", "4"], ["--gpu-list", "0,0"], ["--ranks", "1"], []):
    out["runs"].append(run(extra))
out["stats_mode"] = run([], mode="--stats")
best = max(out["runs"], key=lambda r: r["stream_GBps"] or 0)
out["best"] = best
if orc.have_ref():
    one = "/dev/shm/modes_e2e_1g.bin"
    with open(path, "rb") as f, open(one, "wb") as g:
        buf = bytearray(f.read(1 << 30)); buf[-480:] = b"\x7f" * 480; g.write(buf)
    env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME)
    t0 = time.perf_counter()
    subprocess.run([orc.REF_BIN, "--ifile", one, "--raw", "--no-fix"], stdout=subprocess.DEVNULL, env=env, check=True)
    dt = time.perf_counter() - t0
    out["reference_1gib"] = {"seconds": round(dt, 3), "Msamples_per_s": round((1 << 29) / dt / 1e6, 1)}
    os.remove(one)
os.remove(path)
print(json.dumps(out))
