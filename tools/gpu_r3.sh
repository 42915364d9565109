#!/bin/bash
# Round-3 GPU call: named steps, each with its own timeout and log under gpurun_out/<tag>/.
#   tools/gpu_r3.sh <tag> step [step ...]
TAG=${1:-r3}
shift || true
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name"; ( time timeout "$@" ) > "$O/$name.log" 2>&1; echo "   rc=$? $(tail -n 4 "$O/$name.log" | tr '\n' ' ' | cut -c1-400)"; }
NOISE="python bench.py --workload noise --no-end-to-end --no-cpu-baseline --no-ceiling --no-live-traffic"
for s in "$@"; do
  case $s in
    smoke)     run smoke 400 python __graft_entry__.py smoke ;;
    parity)    run parity 900 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -m gpu -q --maxfail=6 -p no:cacheprovider ;;
    benchtest) run benchtest 1500 python -m pytest tests/test_gpu_bench.py -m gpu -q --maxfail=6 -p no:cacheprovider ;;
    full8)     run full8 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider -k "not sixty_four" ;;
    full)      run full 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider ;;
    bench)     run bench 900 python bench.py ;;
    variants)  for v in 0 2 3; do run noise_v$v 200 $NOISE --demod-variant $v; done
               for v in 3 2; do run noise_v${v}_s1 200 $NOISE --demod-variant $v --streams 1; done ;;
    frames_v)  for v in 0 2; do run frames_v$v 300 python bench.py --workload frames --steps 40 --demod-variant $v; done
               for v in 0 2; do run lowsnr_v$v 300 python bench.py --workload lowsnr --steps 40 --demod-variant $v; done ;;
    parity3)   MODES_GPU_DEMOD_VARIANT=2 run parity3 900 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -m gpu -q --maxfail=6 -p no:cacheprovider ;;
    full3)     MODES_GPU_DEMOD_VARIANT=2 run full3 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider -k "not sixty_four" ;;
    v03)       for v in 0 2; do run noise_v$v 200 $NOISE --demod-variant $v; done
               for v in 0 2; do run frames_v$v 300 python bench.py --workload frames --steps 40 --demod-variant $v; done
               for v in 0 2; do run lowsnr_v$v 300 python bench.py --workload lowsnr --steps 40 --demod-variant $v; done ;;
    ab)        run ab 600 python tools/ab_scan.py --rounds 3 $AB_LIBS ;;
    prof)      bash tools/profile.sh ${TAG} > "$O/prof.log" 2>&1; tail -n 40 "$O/prof.log" ;;
    *)         echo "unknown step $s" ;;
  esac
done
for f in "$O"/bench.log "$O"/noise_v*.log "$O"/frames_v*.log "$O"/lowsnr_v*.log; do
  [ -f "$f" ] && grep '^{' "$f" > "${f%.log}.json"
done
true
