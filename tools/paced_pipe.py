#!/usr/bin/env python3
"""A radio's cadence on a pipe: writes <file> into the stdin of a command one 256 KiB buffer every <ms> milliseconds (one buffer
is 65.5 ms of air time at 2 Msps; dump1090.c:460-512 hands the decoder one such buffer at a time) and reports WHEN the command's
first output arrived, when the writer had finished, and the md5 of everything it printed:
    tools/paced_pipe.py <ms per buffer> <file> [--repeat N] -- <command ...>
-> one JSON line {"first_output_s", "writer_done_s", "buffers", "lines", "md5", "status"}.  A host that waits for a full batch before
its first GPU call prints nothing before the writer is done (33.5 s of silence at the radio's rate)."""
import hashlib, json, subprocess, sys, threading, time

def main():
    sep = sys.argv.index("--")
    args, cmd = sys.argv[1:sep], sys.argv[sep + 1:]
    ms, path = float(args[0]), args[1]
    repeat = int(args[args.index("--repeat") + 1]) if "--repeat" in args else 1
    data = open(path, "rb").read() * repeat
    p = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    t0 = time.monotonic()
    out = {"first": None, "chunks": []}

    def reader():
        while True:
            b = p.stdout.read1(1 << 16)
            if not b:
                return
            if out["first"] is None:
                out["first"] = time.monotonic() - t0
            out["chunks"].append(b)

    th = threading.Thread(target=reader)
    th.start()
    n = 0
    for lo in range(0, len(data), 262144):
        if lo:
            time.sleep(max(0.0, t0 + (n * ms) / 1e3 - time.monotonic()))
        try:
            p.stdin.write(data[lo:lo + 262144])
            p.stdin.flush()
        except BrokenPipeError:
            break
        n += 1
    done = time.monotonic() - t0
    p.stdin.close()
    th.join()
    rc = p.wait()
    text = b"".join(out["chunks"])
    print(json.dumps({"first_output_s": None if out["first"] is None else round(out["first"], 4), "writer_done_s": round(done, 4),
                      "buffers": n, "ms_per_buffer": ms, "lines": text.count(b"\n"), "md5": hashlib.md5(text).hexdigest(), "status": rc}))
    return 0 if rc == 0 else 1

if __name__ == "__main__":
    sys.exit(main())
