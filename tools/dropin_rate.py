#!/usr/bin/env python3
"""What the PATCHED REFERENCE makes of a file of the headline noise in /dev/shm, next to the unmodified reference and to the C++
host on the same file - whole processes, wall clock, `--raw --no-fix > /dev/null`:

    reference                   oracle/_ref/dump1090_ref            (first GiB only: ~2.5 s per GiB)
    patched_reference           oracle/_ref/dump1090_dropin         integration/dump1090_gfx950.patch: one 256 KiB buffer per GPU call (first GiB)
    patched_reference_batched   oracle/_ref/dump1090_dropin_batched integration/dump1090_gfx950_batched.patch: K buffers per hand-off,
                                                                    K = 64, 512 (default), 2048
    cxx_host                    dump1090_amd/bin/dump1090_amd

    python tools/dropin_rate.py [GiB]        (default 8)           prints one JSON line"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import oracle as orc
from dump1090_amd import Demodulator

gib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
path, one = "/dev/shm/modes_dropin.bin", "/dev/shm/modes_dropin_1g.bin"
d = Demodulator(fix=False)
with open(path, "wb") as f:
    for k in range(gib):
        iq = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
        d.synth_noise(iq, k << 30, seed=20260922, sigma_q16=941)
        if k == gib - 1:
            d.fill(iq[-480:], 127)
        a = iq.cpu().numpy()
        a.tofile(f)
        if k == 0:
            b = a.copy(); b[-480:] = 127; b.tofile(one)
d.close()
del iq
torch.cuda.empty_cache()
env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME)
out = {"file_gib": gib}


def rate(name, exe, file, nbytes, runs, extra_env=None):
    if not os.path.exists(exe):
        return
    best, lines, inner = 1e9, 0, None
    e = dict(env if "dump1090_amd/bin" not in exe else os.environ, MODES_DROPIN_TIMING="1", MODES_DROPIN_FAST_EXIT="1", **(extra_env or {}))
    args = ["--timing"] if "dump1090_amd/bin" in exe else []
    for _ in range(runs):
        time.sleep(1.0)      # (a process started right behind another GPU process's exit waits ~0.1 s longer for the HIP runtime: tools/e2e_cli.py)
        t0 = time.perf_counter()
        p = subprocess.run([exe, "--ifile", file, "--raw", "--no-fix"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, check=True)
        dt = time.perf_counter() - t0
        if dt < best:
            best = dt
            js = [ln for ln in p.stderr.decode().splitlines() if ln.startswith("{")]
            inner = json.loads(js[-1]) if js else None
        lines = p.stdout.count(b"\n")
    out[name] = {"seconds": round(best, 3), "Msamples_per_s": round(nbytes / 2 / best / 1e6, 1), "GB_per_s": round(nbytes / best / 1e9, 2),
                 "lines": lines, "file_gib": nbytes >> 30}
    if inner:                                             # the process's own account: start-up apart, first read .. last message
        out[name]["stream_Msamples_per_s"] = inner.get("stream_Msamples_per_s")
        out[name]["stream_s"] = inner.get("stream_s")
        out[name]["start_up_s"] = round(inner.get("gpu_init_s", inner.get("init_s", 0.0)) + inner.get("reader_setup_s", 0.0), 4)


REFDIR = os.path.join(ROOT, "oracle", "_ref")
rate("reference", orc.REF_BIN, one, 1 << 30, 1)
rate("patched_reference", os.path.join(REFDIR, "dump1090_dropin"), one, 1 << 30, 2)
for k in (64, 512, 2048):
    rate("patched_reference_batched_k%d" % k, os.path.join(REFDIR, "dump1090_dropin_batched"), path, gib << 30, 4, {"MODES_DROPIN_BLOCKS": str(k)})
rate("cxx_host", os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd"), path, gib << 30, 4)
os.remove(path)
os.remove(one)
print(json.dumps(out))
