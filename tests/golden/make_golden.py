#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the UNMODIFIED reference.

Runs oracle/_ref/dump1090_ref (the reference's own dump1090.c compiled by
oracle/Makefile from /root/reference, under the constant-clock interposer
oracle/_ref/libfixedtime.so) on every named stream of tests/synth.py plus the
reference's only fixture (testfiles/modes1.bin, padded - copied to
tests/golden/modes1.bin), for every flag set of the hot path, and stores the
exact stdout.  Only runnable where /root/reference exists (the build
container); the GPU box and CI consume the committed JSON.

    python tests/golden/make_golden.py
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "dump1090_ref")
FIXED_TIME = os.path.join(ROOT, "oracle", "_ref", "libfixedtime.so")
REF_FIXTURE = "/root/reference/testfiles/modes1.bin"

RAW_FLAGSETS = {
    "default": [],
    "nofix": ["--no-fix"],
    "aggressive": ["--aggressive"],
    "nocrc": ["--no-crc-check"],
    "nofix_nocrc": ["--no-fix", "--no-crc-check"],
    "aggressive_nocrc": ["--aggressive", "--no-crc-check"],
}
STATS_FLAGSETS = {"default": [], "nofix": ["--no-fix"], "aggressive": ["--aggressive"]}
VERBOSE_FLAGSETS = {"default": [], "aggressive_nocrc": ["--aggressive", "--no-crc-check"]}   # no --raw: full text dump

CASES = {
    "modes1": lambda: synth.modes1_padded(os.path.join(HERE, "modes1.bin")),
    "uniform": synth.case_uniform,
    "coarse": synth.case_coarse,
    "edges": synth.case_edges,
    "edges_smear": lambda: synth.case_edges(seed=23, smear16=6),
    "frames": synth.case_frames,
    "smear": synth.case_smear,
    "lowsnr": synth.case_lowsnr,
    "noise": synth.case_noise,
    "saturated": synth.case_saturated,
}


def run_ref(path, flags):
    env = dict(os.environ, LD_PRELOAD=FIXED_TIME)
    r = subprocess.run([REF_BIN, "--ifile", path] + flags, capture_output=True, env=env, check=True)
    return r.stdout.decode()


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    if os.path.exists(REF_FIXTURE):
        shutil.copyfile(REF_FIXTURE, os.path.join(HERE, "modes1.bin"))
    golden = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, make in CASES.items():
            data = make()
            path = os.path.join(tmp, name + ".bin")
            data.tofile(path)
            entry = {"nbytes": int(len(data)), "input_md5": hashlib.md5(data.tobytes()).hexdigest(),
                     "raw": {}, "onlyaddr": {}, "stats": {}, "verbose": {}}
            for fname, flags in RAW_FLAGSETS.items():
                out = run_ref(path, ["--raw"] + flags)
                entry["raw"][fname] = {"lines": out.count("\n"), "md5": hashlib.md5(out.encode()).hexdigest(),
                                       "text": out}
            out = run_ref(path, ["--onlyaddr"])
            entry["onlyaddr"]["default"] = {"lines": out.count("\n"), "md5": hashlib.md5(out.encode()).hexdigest(),
                                            "text": out}
            for fname, flags in STATS_FLAGSETS.items():
                out = run_ref(path, ["--stats"] + flags)
                entry["stats"][fname] = {"md5": hashlib.md5(out.encode()).hexdigest(), "text": out}
            for fname, flags in VERBOSE_FLAGSETS.items():
                out = run_ref(path, flags)
                entry["verbose"][fname] = {"lines": out.count("\n"), "md5": hashlib.md5(out.encode()).hexdigest(),
                                           "text": out}
            golden[name] = entry
            print(name, {k: v["lines"] for k, v in entry["raw"].items()})
    with gzip.open(os.path.join(HERE, "golden.json.gz"), "wt", compresslevel=9) as f:
        json.dump(golden, f, sort_keys=True)
    summary = {c: {"nbytes": e["nbytes"], "input_md5": e["input_md5"],
                   "raw": {k: {"lines": v["lines"], "md5": v["md5"]} for k, v in e["raw"].items()},
                   "verbose": {k: {"lines": v["lines"], "md5": v["md5"]} for k, v in e["verbose"].items()},
                   "stats": {k: v["text"] for k, v in e["stats"].items()}} for c, e in golden.items()}
    with open(os.path.join(HERE, "golden_summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
