#!/usr/bin/env python3
"""The kernels of a few consecutive steps in the middle of a rocprofv3 --kernel-trace of bench.py, on one time axis: which queue (launch
stream) a kernel ran on, when it started and ended relative to the first scan shown, how long it took and how much of it overlapped
the other queue's kernels - VERDICT r4 item 2c, the two-stream region.   python tools/region_timeline.py <trace dir> [steps]"""
import csv, glob, os, statistics, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
HOT = ("scan_kernel", "demod_kernel", "select_kernel", "record_kernel", "finalize2_kernel", "finalize_kernel", "order_kernel")
ev = []
for r in rows:
    for k in HOT:
        if k in r["Kernel_Name"]:
            ev.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]))
            break
scans = [i for i, e in enumerate(ev) if e[0] == "scan_kernel"]
# the timed region of a two-stream run: the longest stretch in which consecutive scans alternate between two queues (the settling
# steps and the isolated timing calls run on one); a one-stream run: the last 60 % of the trace
alt, best, start = [], (0, 0), None
for n, (a, b) in enumerate(zip(scans, scans[1:])):
    if ev[a][3] != ev[b][3]:
        start = n if start is None else start
        if n + 1 - start > best[1] - best[0]:
            best = (start, n + 1)
    else:
        start = None
if best[1] - best[0] > 20:
    scans = scans[best[0]:best[1]]
else:
    scans = scans[len(scans) * 4 // 10:]
mid = scans[len(scans) // 2]
t0 = ev[mid][1]
last = scans[scans.index(mid) + nshow]
print("%-18s %-6s %10s %10s %9s   overlap with the other queue's kernels" % ("kernel", "queue", "start us", "end us", "dur us"))
for i in range(mid, last + 1):
    k, a, b, q = ev[i]
    ov = 0
    for k2, a2, b2, q2 in ev[max(0, i - 8):i + 8]:
        if q2 != q:
            ov += max(0, min(b, b2) - max(a, a2))
    print("%-18s %-6s %10.1f %10.1f %9.1f   %6.1f us" % (k, q, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, ov / 1e3))
# steady state: the 40 % .. 90 % stretch of the scans
seg = [ev[i] for i in range(scans[len(scans) // 10], scans[len(scans) * 9 // 10])]
per = [b[1] - a[1] for a, b in zip([e for e in seg if e[0] == "scan_kernel"], [e for e in seg if e[0] == "scan_kernel"][1:])]
dur = {}
for e in seg:
    dur.setdefault(e[0], []).append(e[2] - e[1])
print("steady state: scan-to-scan period mean %.1f us (median %.1f); " % (statistics.mean(per) / 1e3, statistics.median(per) / 1e3) +
      ", ".join("%s mean %.1f us" % (k, statistics.mean(v) / 1e3) for k, v in dur.items()))
tail = [e for e in seg if e[0] != "scan_kernel"]
print("sum of the kernels' own durations per step: %.1f us -> %.1f us of every step run under another kernel" % (
    sum(statistics.mean(v) for v in dur.values()) / 1e3, (sum(statistics.mean(v) for v in dur.values()) - statistics.mean(per)) / 1e3))
