#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (steady state: the last 60 % of the trace)."""
import csv, glob, os, statistics, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    for k in ("scan_kernel", "demod_kernel", "select_kernel", "record_kernel", "finalize2_kernel", "finalize_kernel", "order_kernel", "prefix_kernel"):
        if k in n:
            return k
    return n[:24]


ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ev = [e for e in ev if e[0] in ("scan_kernel", "demod_kernel", "select_kernel", "record_kernel", "finalize2_kernel", "finalize_kernel", "order_kernel")]
ev = ev[len(ev) * 4 // 10:]
gaps, dur = {}, {}
for a, b in zip(ev, ev[1:]):
    gaps.setdefault(a[0] + " -> " + b[0], []).append(b[1] - a[2])
for e in ev:
    dur.setdefault(e[0], []).append(e[2] - e[1])
for k, v in dur.items():
    print("%-28s n=%4d  median %8.1f us  mean %8.1f" % (k, len(v), statistics.median(v) / 1e3, statistics.mean(v) / 1e3))
for k, v in gaps.items():
    print("gap %-36s n=%4d  median %6.1f us  mean %6.1f  min %6.1f  max %6.1f" % (k, len(v), statistics.median(v) / 1e3, statistics.mean(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
# how much of the time is more than one of these kernels running?
edges = sorted([(e[1], 1) for e in ev] + [(e[2], -1) for e in ev])
busy = over = 0
depth = 0
for (t, d), (t2, _) in zip(edges, edges[1:]):
    depth += d
    if depth >= 1:
        busy += t2 - t
    if depth >= 2:
        over += t2 - t
span = edges[-1][0] - edges[0][0]
print("of %.1f ms traced: a kernel is running %.1f %% of the time, two or more %.1f %%" % (span / 1e6, 100.0 * busy / span, 100.0 * over / span))
scans = [e for e in ev if e[0] == "scan_kernel"]
per = [b[1] - a[1] for a, b in zip(scans, scans[1:])]
print("scan-to-scan period: median %.1f us, mean %.1f us" % (statistics.median(per) / 1e3, statistics.mean(per) / 1e3))
