#!/usr/bin/env python3
"""Golden lines of the reference's two network sinks for tests/golden/modes1.bin.

Runs the compiled reference (oracle/_ref/dump1090_ref, built from /root/reference by oracle/Makefile)
with --net and the constant-clock interposer, input on stdin, and records what it writes to a client
of its raw-output port (modesSendRawOutput, dump1090.c:2381) and of its SBS port (modesSendSBSOutput,
dump1090.c:2397).  The reference accepts clients only between buffers (backgroundTasks,
dump1090.c:2831), so an all-127 buffer is fed first, the clients are given time to be accepted, and
only then the capture follows.  Output: modes1_rawnet[_<flags>].txt, modes1_sbs[_<flags>].txt.
"""
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc          # noqa: E402
import synth                  # noqa: E402

PORTS = dict(ro=31002, ri=31001, http=31080, sbs=31003)


def capture(data: np.ndarray, flags):
    env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME)
    args = [orc.REF_BIN, "--net", "--net-ro-port", str(PORTS["ro"]), "--net-ri-port", str(PORTS["ri"]),
            "--net-http-port", str(PORTS["http"]), "--net-sbs-port", str(PORTS["sbs"]), "--ifile", "-", "--raw"] + flags
    proc = subprocess.Popen(args, stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
    out = {}

    def reader(name, port):
        for _ in range(100):
            try:
                s = socket.create_connection(("127.0.0.1", port), timeout=5)
                break
            except OSError:
                time.sleep(0.05)
        s.settimeout(30)
        buf = b""
        while True:
            chunk = s.recv(1 << 16)
            if not chunk:
                break
            buf += chunk
        out[name] = buf

    threads = [threading.Thread(target=reader, args=(n, PORTS[n])) for n in ("ro", "sbs")]
    for t in threads:
        t.start()
    time.sleep(0.5)                                             # both connections are in the listen backlog
    proc.stdin.write(bytes([127]) * synth.DATA_LEN)             # one empty buffer -> backgroundTasks() accepts them
    proc.stdin.flush()
    time.sleep(0.5)
    stdout_chunks = []
    drain = threading.Thread(target=lambda: stdout_chunks.append(proc.stdout.read()))
    drain.start()
    proc.stdin.write(data.tobytes())
    proc.stdin.close()
    drain.join()
    assert proc.wait() == 0
    for t in threads:
        t.join()
    return out["ro"].decode(), out["sbs"].decode(), stdout_chunks[0].decode()


def main():
    one = synth.modes1_padded(os.path.join(HERE, "modes1.bin"))
    for tag, flags in (("", []), ("_aggressive", ["--aggressive"])):
        ro, sbs, raw = capture(one, flags)
        # the raw port carries exactly the --raw listing in upper case: the capture lost nothing
        assert ro.lower() == raw, "raw port and stdout disagree"
        open(os.path.join(HERE, "modes1_rawnet%s.txt" % tag), "w").write(ro)
        open(os.path.join(HERE, "modes1_sbs%s.txt" % tag), "w").write(sbs)
        print(tag or "default", len(ro.splitlines()), "raw lines,", len(sbs.splitlines()), "SBS lines")


if __name__ == "__main__":
    main()
