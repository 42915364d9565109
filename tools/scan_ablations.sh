#!/bin/bash
# The timing-only builds behind profiles/HISTORY.md B.1 ("where the scan's last 10 % are"), all in ONE process, interleaved
# (tools/ab_scan.py): the production library next to builds of the same sources with parts of the scan kernel taken out.
#   SCAN_ABL_NOPERM      the eight odd-pair v_perm per chunk cost nothing (some other dword of the window stands in: same survivor rate)
#   SCAN_ABL             no hand-off: ordering survivors are counted, not pushed; no level pass
#   SCAN_ABL_NOBETA      no level pass (the compiler then drops the push's stores as dead: equals SCAN_ABL)
#   SCAN_ABL_PUSH=n      n fewer window dwords per push (the level pass then works on stale windows and forwards more)
#   SCAN_ABL_EXTRA=n     n more full-rate vector instructions per chunk
# None of them produces the right positions (the counts in the output say how far off); run on the GPU box:
#   tools/scan_ablations.sh [rounds]      -> gpurun_out/scan_ablations.txt
R=${1:-3}
D=$(mktemp -d /tmp/scan_abl.XXXXXX)
cd "$(dirname "$0")/../dump1090_amd/csrc" || exit 1
build() { hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $2 -shared -o "$D/lib_$1.so" modes_gfx950.hip -lpthread 2>/dev/null & }
build production ""
build noperm "-DSCAN_ABL_NOPERM"
build noperm_extra4 "-DSCAN_ABL_NOPERM -DSCAN_ABL_EXTRA=4"
build no_handoff "-DSCAN_ABL"
build noperm_no_handoff "-DSCAN_ABL_NOPERM -DSCAN_ABL"
build push4 "-DSCAN_ABL_PUSH=4"
wait
cd ../.. && mkdir -p gpurun_out
python tools/ab_scan.py --rounds "$R" "$D"/lib_production.so "$D"/lib_noperm.so "$D"/lib_noperm_extra4.so "$D"/lib_no_handoff.so \
    "$D"/lib_noperm_no_handoff.so "$D"/lib_push4.so 2>&1 | tee gpurun_out/scan_ablations.txt
rm -rf "$D"
