#!/bin/bash
# FETCH_SIZE (and WRITE_SIZE) of kernels whose line traffic is known: tools/fetch_calib.sh -> gpurun_out/fetch_calib/{summary.json,...}
R=$PWD; O=$R/gpurun_out/fetch_calib; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o fetch -f csv -- $R/tools/ubench_fetch_calib > $O/expected.json 2> $O/fetch.log
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/rdreq -o rdreq -f csv -- $R/tools/ubench_fetch_calib > /dev/null 2> $O/rdreq.log
cd $R
python - <<'PY'
import csv, glob, json
from collections import defaultdict
exp = json.loads([l for l in open("gpurun_out/fetch_calib/expected.json") if l.startswith("{")][0])
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/fetch_calib/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        for k in exp:
            if k in row["Kernel_Name"]:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, want in exp.items():
    if k == "launches" or "FETCH_SIZE" not in acc[k]:
        continue
    v = acc[k]["FETCH_SIZE"][1:]                      # (the first launch also pays the cold caches)
    kb = sum(v) / len(v)
    out[k] = {"line_bytes_per_launch": want, "FETCH_SIZE_KB": round(kb, 1), "bytes_per_counted_KB": round(want / kb, 1),
              "factor_on_KB_x_1024": round(want / (kb * 1024), 3)}
    for c in acc[k]:
        if c != "FETCH_SIZE":
            w = acc[k][c][1:]
            out[k][c] = round(sum(w) / len(w), 1)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/fetch_calib/summary.json", "w"), indent=1)
PY
find $O -name "*.csv" -size +1M -delete
