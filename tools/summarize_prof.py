#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile.sh) into a short text summary per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict
trace_avg_us = {}

out = sys.argv[1]
HOT = ("scan_kernel", "demod_kernel", "select_kernel", "record_kernel", "finalize2_kernel", "finalize_kernel", "order_kernel")
SKIP = int(sys.argv[2]) if len(sys.argv) > 2 else 82       # launches in front of the timed region (bench.py: settle + warmup)


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


def short(name):
    for k in ("scan_kernel", "demod_kernel", "select_kernel", "record_kernel", "finalize2_kernel", "finalize_kernel", "order_kernel",
              "prefix_kernel", "compact_candidates", "synth_noise", "fill_kernel", "magnitude_kernel", "power_kernel", "stream_read_kernel"):
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
    order = defaultdict(list)
    meta = {}
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for row in rows:
        n = short(row["Kernel_Name"])
        d[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        order[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[n] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                   row.get("Scratch_Size"), row.get("Workgroup_Size"), row.get("Grid_Size"))
    print("== kernel trace durations (%s)" % os.path.relpath(f, out))
    for n, v in d.items():
        v.sort()
        print("  %-24s n=%4d  median %9.1f us  min %9.1f  max %9.1f   vgpr/agpr/sgpr/lds/scratch/wg/grid=%s" % (
            n, len(v), v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3, meta[n]))
        if n in order and len(order[n]) > SKIP + 10:
            tail = order[n][SKIP:]                      # bench.py's timed steps (after its settle + warmup steps)
            print("  %-24s the %d launches of the timed region: avg %9.1f us" % (n, len(tail), sum(tail) / len(tail) / 1e3))
            trace_avg_us[n] = sum(tail) / len(tail) / 1e3
    kt_log = os.path.join(out, "kt.log")
    if os.path.exists(kt_log):
        import json
        for line in open(kt_log):
            if line.startswith("{") and "kernel_ms" in line:
                j = json.loads(line)
                print("  bench.py in the same run (HIP events on the launch stream, timed region): %s, ms_per_step %s" % (
                    json.dumps(j["kernel_ms"]), j["ms_per_step"]))

for f in find("*counter_collection.csv"):
    agg = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== counters (%s): mean per dispatch" % os.path.relpath(f, out))
    for k, cs in agg.items():
        if k not in HOT:
            continue
        print("  %s" % k)
        for c, v in cs.items():
            big = max(v)
            print("    %-24s mean %16.1f   max %16.1f   n=%d" % (c, sum(v) / len(v), big, len(v)))

# HBM traffic per scan launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
# FETCH_SIZE (KB) counts 128-byte requests as 64 bytes on wide coalesced streaming reads -> x2.
import json
traffic = {}
for f in find("*counter_collection.csv"):
    agg = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in HOT:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if c in agg.get(k, {}):
                v = agg[k][c]
                traffic.setdefault(k, {})[c + "_KB_mean"] = sum(v) / len(v)
# The factor on FETCH_SIZE: x 2 for the scan kernel (the guide's gfx950 correction for wide coalesced streaming reads); for the demodulation
# kernels - scattered 16-byte pieces, "uncalibrated" by the guide's own words - the factor MEASURED on their access pattern with known line
# traffic (tools/fetch_calib.sh -> profiles/fetch_size_calibration.json: stage-2 pattern), x 2 with a note when that file is missing.
calib = None
try:
    calib = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "fetch_size_calibration.json")))
except (OSError, ValueError):
    pass
for k, d in traffic.items():
    if "FETCH_SIZE_KB_mean" in d:
        factor, how = 2.0, "guide: x 2 (streaming reads)"
        if k != "scan_kernel":
            if calib and "calib_stage2" in calib:
                factor, how = float(calib["calib_stage2"]["factor_on_KB_x_1024"]), "measured on the stage-2 access pattern (profiles/fetch_size_calibration.json)"
            else:
                how = "x 2 UNCALIBRATED for this access pattern"
        d["fetch_size_factor"] = factor
        d["fetch_size_factor_source"] = how
        d["hbm_read_bytes_per_launch"] = int(d["FETCH_SIZE_KB_mean"] * 1024 * factor)
    if "WRITE_SIZE_KB_mean" in d:
        d["hbm_write_bytes_per_launch_uncalibrated"] = int(d["WRITE_SIZE_KB_mean"] * 1024)
if traffic:
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rel in ("dump1090_amd/csrc/modes_gfx950.hip", "dump1090_amd/csrc/modes_core.h"):
        h.update(open(os.path.join(root, rel), "rb").read())
    traffic["kernel_source_sha256_16"] = h.hexdigest()[:16]      # bench.py reports the counters only for these very sources
    traffic["tag"] = os.path.basename(os.path.normpath(out))
    for k, us in trace_avg_us.items():                           # the kernel trace's own average over bench.py's timed launches
        traffic.setdefault(k, {})["trace_avg_us_timed_region"] = round(us, 2)
    traffic["workload"] = sys.argv[3] if len(sys.argv) > 3 else "noise"
    traffic["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on bench.py's %s workload; FETCH_SIZE (KB) x 1024 x "
                        "fetch_size_factor per kernel" % traffic["workload"])
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
