#!/usr/bin/env python3
"""tests/golden/config_listings.json: what the UNMODIFIED reference prints for the big synthetic streams bench.py
demodulates at N = 1, 2, 4, 8 GPUs (BASELINE.json configs[2], [3], [4]) - line count and md5 of `--raw` stdout.

The streams are tests/synth.py:config3_stream(...) - too big to materialise in numpy (12 MB/s), so gen_stream.c
(same integers: checked below against synth on a small stream) writes them into the stdin of
oracle/_ref/dump1090_ref (compiled from /root/reference by oracle/Makefile, constant clock interposed).  Only runnable
where /root/reference exists; bench.py and tests/test_gpu_fullsize.py consume the committed JSON.

    python tests/golden/make_listings.py [key ...]          (default: every stream; ~15 min on 8 cores)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "dump1090_ref")
FIXED_TIME = os.path.join(ROOT, "oracle", "_ref", "libfixedtime.so")
OUT = os.path.join(HERE, "config_listings.json")
GEN = os.path.join(tempfile.gettempdir(), "gen_stream")

LOWSNR = dict(per=16384, amp=(8, 15), smear=(3, 4, 5, 6), flip1=10, flip2=20, edge_every=61)


def streams():
    """key -> (seed, buffers, config3_stream keywords, reference flags).  The keys are what bench.py looks up."""
    s = {}
    for gib in (8, 16, 32, 64):                       # frames leg: 8 GiB per GPU; seed 3 at N = 1, 4 beyond (bench.py)
        seed = 3 if gib == 8 else 4
        s["frames:%d:%d" % (seed, gib * 4096)] = (seed, gib * 4096, {}, [])
    for gib in (1, 2, 4, 8):                          # low-SNR leg: 1 GiB per GPU, --aggressive
        s["lowsnr:5:%d" % (gib * 4096)] = (5, gib * 4096, LOWSNR, ["--aggressive"])
    return s


def build_gen():
    subprocess.run(["gcc", "-O3", "-march=native", "-fopenmp", "-o", GEN, os.path.join(HERE, "gen_stream.c")], check=True)


def write_patches(st, path):
    first, data = st.patches()
    with open(path, "wb") as f:
        np.array([len(first), data.shape[1] if len(first) else 0], dtype=np.int64).tofile(f)
        first.astype(np.int64).tofile(f)
        data.tofile(f)


def self_check():
    """gen_stream's bytes == synth's on a stream small enough for numpy"""
    st = synth.config3_stream(7, 40, per=32768, edge_every=5)
    with tempfile.NamedTemporaryFile(suffix=".patch") as pf:
        write_patches(st, pf.name)
        got = subprocess.run([GEN, "7", str(st.nbytes), str(st.sigma_q16), pf.name], capture_output=True, check=True).stdout
    assert np.array_equal(np.frombuffer(got, dtype=np.uint8), st.window(0, st.nbytes)), "gen_stream.c and tests/synth.py disagree"


def listing(seed, nblocks, kw, flags):
    st = synth.config3_stream(seed, nblocks, **kw)
    with tempfile.NamedTemporaryFile(suffix=".patch") as pf:
        write_patches(st, pf.name)
        env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
        gen = subprocess.Popen([GEN, str(seed), str(st.nbytes), str(st.sigma_q16), pf.name], stdout=subprocess.PIPE, env=env)
        ref = subprocess.Popen([REF_BIN, "--ifile", "-", "--raw"] + flags, stdin=gen.stdout, stdout=subprocess.PIPE,
                               env=dict(os.environ, LD_PRELOAD=FIXED_TIME))
        gen.stdout.close()
        h, lines = hashlib.md5(), 0
        for chunk in iter(lambda: ref.stdout.read(1 << 20), b""):
            h.update(chunk)
            lines += chunk.count(b"\n")
        assert ref.wait() == 0 and gen.wait() == 0
    return {"stream": "synth.config3_stream(%d, %d%s)" % (seed, nblocks, "".join(", %s=%r" % kv for kv in sorted(kw.items()))),
            "flags": " ".join(["--raw"] + flags), "lines": lines, "md5": h.hexdigest(), "frames": len(st.placements),
            "by": "oracle/_ref/dump1090_ref (the compiled reference), stream by tests/golden/gen_stream.c"}


def main():
    assert os.path.exists(REF_BIN), "build oracle/_ref first (make -C oracle)"
    build_gen()
    self_check()
    have = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            have = json.load(f)
    todo = streams()
    for key in (sys.argv[1:] or sorted(todo, key=lambda k: todo[k][1])):
        have[key] = listing(*todo[key])
        print(key, have[key]["lines"], have[key]["md5"], flush=True)
        with open(OUT, "w") as f:
            json.dump(have, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
