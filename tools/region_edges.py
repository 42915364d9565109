#!/usr/bin/env python3
"""Around the idle gaps (> 40 us) of a rocprofv3 --kernel-trace of bench.py: the kernels before and after each gap - what a
timed region pays at its two ends.  python tools/region_edges.py <trace dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    for k in ("scan_kernel", "demod_kernel", "finalize_kernel", "stream_read_kernel", "synth_noise"):
        if k in n:
            ev.append((k, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
            break
busy_end = ev[0][2]
t00 = ev[0][1]
for i in range(1, len(ev)):
    gap = ev[i][1] - busy_end
    if gap > 40000:
        print("--- idle %.1f us at %.3f ms" % (gap / 1e3, (ev[i][1] - t00) / 1e6))
        for e in ev[max(0, i - 4):i]:
            print("   before: %-18s start %+9.1f us  dur %7.1f us" % (e[0], (e[1] - ev[i][1]) / 1e3, (e[2] - e[1]) / 1e3))
        for e in ev[i:i + 10]:
            print("   after : %-18s start %+9.1f us  dur %7.1f us" % (e[0], (e[1] - ev[i][1]) / 1e3, (e[2] - e[1]) / 1e3))
    busy_end = max(busy_end, ev[i][2])
