#!/bin/bash
# Hardware counters of the scan / demodulation kernels for several builds of the library, one rocprofv3 --pmc pass per counter set
# and build (never combined with other trace domains):   tools/pmc_ab.sh <tag> lib1.so lib2.so ...
# -> gpurun_out/pmc_<tag>/<lib>/<set>/..., summary on stdout (mean per dispatch of the hot kernels)
TAG=$1; shift
R=$PWD
export TMPDIR=/tmp
cd /tmp
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
      "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS")
for lib in "$@"; do
  n=$(basename $lib .so)
  for i in 0 1; do
    O=$R/gpurun_out/pmc_$TAG/$n/set$i
    mkdir -p $O
    MODES_GPU_LIB=$R/$lib timeout 300 rocprofv3 --kernel-trace --pmc ${SETS[$i]} -d $O -o pmc -f csv -- python $R/bench.py --workload ${WL:-noise} --no-end-to-end \
        --no-ceiling --no-cpu-baseline --no-live-traffic --streams 1 --leg-streams 1 --settle 20 --steps 3 --warmup 1 --depth 1 --time-every 100000 $BENCH_EXTRA > $O/log.txt 2>&1
    echo "$n set$i rc=$?"
  done
done
cd $R
python - "$TAG" "$@" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
tag, libs = sys.argv[1], sys.argv[2:]
for lib in libs:
    n = os.path.basename(lib)[:-3]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob("gpurun_out/pmc_%s/%s/set*/**/*counter_collection.csv" % (tag, n), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            for short in ("scan_kernel", "demod_kernel", "select_kernel"):
                if short in k:
                    acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in acc:
        print("%-28s %-14s %s" % (n, k, "  ".join("%s %.4g" % (c, sum(v[-3:]) / len(v[-3:])) for c, v in sorted(acc[k].items()))))
PY
find gpurun_out/pmc_$TAG -name "*.csv" -size +2M -delete
