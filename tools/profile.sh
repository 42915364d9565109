#!/bin/bash
# rocprofv3 recipe for the bench workload (run on the GPU box from the repo root):
#   tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/{kt,pmc1,pmc2,fetch,write}/...
# Kernel-trace + stats in one run; counters in their own runs (never with --sys-trace etc.).
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"            # extra bench.py arguments, e.g. --scan-variant 2
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --workload noise --no-end-to-end --no-live-traffic --steps 300 --warmup 2 --no-cpu-baseline --depth 4 --streams 1 --time-every 8 $EXTRA"   # the headline leg of the default `python bench.py`   # the timed region dominates the --stats average
SHORT="python $R/bench.py --workload noise --no-end-to-end --no-live-traffic --settle 20 --steps 3 --warmup 1 --no-cpu-baseline --depth 1 --streams 1 --time-every 100000 $EXTRA"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -f csv -- $BENCH > "$OUT/kt.log" 2>&1
echo "kt rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    -d "$OUT/pmc1" -o pmc1 -f csv -- $SHORT > "$OUT/pmc1.log" 2>&1
echo "pmc1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE \
    -d "$OUT/pmc2" -o pmc2 -f csv -- $SHORT > "$OUT/pmc2.log" 2>&1
echo "pmc2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -f csv -- $SHORT > "$OUT/fetch.log" 2>&1
echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -f csv -- $SHORT > "$OUT/write.log" 2>&1
echo "write rc=$?"
cd "$R"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep the merge-back small: raw per-dispatch CSVs of the counter runs can be large
find "$OUT" -name "*.csv" -size +4M -delete
