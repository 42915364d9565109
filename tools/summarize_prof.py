#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile.sh) into a short text summary per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


def short(name):
    for k in ("scan_kernel", "demod_kernel", "finalize_kernel", "compact_candidates", "synth_noise", "fill_kernel",
              "magnitude_kernel", "power_kernel"):
        if k in name:
            return k
    return name[:60]


for f in find("*kernel_stats.csv"):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-24s calls %6s  total %12s ns  avg %10s ns  min %10s  max %10s  %6s%%" % (
            short(row.get("Name", "")), row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"),
            row.get("MinNs"), row.get("MaxNs"), row.get("Percentage")))

for f in find("*kernel_trace.csv"):
    if os.sep + "kt" + os.sep not in f:
        continue
    d = defaultdict(list)
    meta = {}
    for row in csv.DictReader(open(f)):
        n = short(row["Kernel_Name"])
        d[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[n] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                   row.get("Scratch_Size"), row.get("Workgroup_Size"), row.get("Grid_Size"))
    print("== kernel trace durations (%s)" % os.path.relpath(f, out))
    for n, v in d.items():
        v.sort()
        print("  %-24s n=%4d  median %9.1f us  min %9.1f  max %9.1f   vgpr/agpr/sgpr/lds/scratch/wg/grid=%s" % (
            n, len(v), v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3, meta[n]))

for f in find("*counter_collection.csv"):
    agg = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== counters (%s): mean per dispatch" % os.path.relpath(f, out))
    for k, cs in agg.items():
        if k not in ("scan_kernel", "demod_kernel", "finalize_kernel"):
            continue
        print("  %s" % k)
        for c, v in cs.items():
            big = max(v)
            print("    %-24s mean %16.1f   max %16.1f   n=%d" % (c, sum(v) / len(v), big, len(v)))
