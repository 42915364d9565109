#!/bin/bash
# Scan time against the run length (chunks per wavefront) on one box: tools/ab_runlen.sh [run_chunks ...]
# (the shipped library, the bench noise; every value in its own tools/ab_scan.py process, 2 interleaved rounds)
R=$PWD
O=$R/gpurun_out/ab_runlen
mkdir -p "$O"
for rc in "$@"; do
  echo "== run_chunks $rc"
  timeout 300 python tools/ab_scan.py --rounds 2 --run-chunks $rc dump1090_amd/libmodes_gfx950.so 2>&1 | tail -n 2
done | tee "$O/ab_runlen.txt"
