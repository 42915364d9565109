#!/bin/bash
# rocprofv3 recipe for the bench workloads (run on the GPU box from the repo root):
#   tools/profile.sh <tag> [workload [extra bench.py arguments]]   workload: noise (default) | lowsnr | frames | strong
#   -> gpurun_out/prof_<tag>/{kt,pmc1,pmc2,fetch,write}/... + summary.txt (+ traffic.json)
# Kernel-trace + stats in one run; counters in their own runs (never with --sys-trace etc.).
set -u
TAG=${1:-r01}
shift || true
WL=${1:-noise}
shift || true
EXTRA="$*"            # extra bench.py arguments, e.g. --demod-variant 2
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
case $WL in
  noise)  STEPS=300; SETTLE=240 ;;       # 1 GiB steps
  lowsnr) STEPS=200; SETTLE=240 ;;       # 1 GiB steps
  frames) STEPS=40;  SETTLE=240 ;;      # 8 GiB steps (bench.py turns --settle into ceil(240 / 8) = 30 of them)
  strong) STEPS=8;   SETTLE=240 ;;      # 64 GiB steps of 9 calls (configs[3]'s stream on one GPU; at least 6 settle steps)
  *) echo "workload $WL?"; exit 2 ;;
esac
COMMON="--workload $WL --regions 1 --no-end-to-end --no-live-traffic --no-cpu-baseline --no-ceiling --streams 1 --leg-streams 1"
BENCH="python $R/bench.py $COMMON --steps $STEPS --warmup 2 --settle $SETTLE --depth 4 --time-every 8 $EXTRA"   # the timed region dominates the --stats average
SHORT="python $R/bench.py $COMMON --settle 20 --steps 3 --warmup 1 --depth 1 --time-every 100000 $EXTRA"
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
# launches in front of the timed region: the leg's settle steps (ceil(SETTLE / GiB per step), at least 6) + warmup... bench.py's
# noise leg adds its warmup to --settle; the other legs settle max(6, ceil(settle / GiB)) steps of `calls` launches each
case $WL in noise) SKIP=$((SETTLE + 2)) ;; lowsnr) SKIP=$SETTLE ;; frames) SKIP=60 ;; strong) SKIP=54 ;; esac
python tools/summarize_prof.py "$OUT" $SKIP $WL > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep the merge-back small: raw per-dispatch CSVs of the counter runs can be large
find "$OUT" -name "*.csv" -size +4M -delete
