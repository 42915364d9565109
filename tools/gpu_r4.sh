#!/bin/bash
# Round-4 GPU call: named steps, each with its own timeout and log under gpurun_out/<tag>/.
#   tools/gpu_r4.sh <tag> step [step ...]
TAG=${1:-r4}
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
    parity2)   MODES_GPU_DEMOD_VARIANT=2 run parity2 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=6 -p no:cacheprovider ;;
    forced)    MODES_GPU_DEMOD_VARIANT=2 run pytest_gpu_forced_two_kernel_path 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_dropin.py -m gpu -q -p no:cacheprovider -k "not sixty_four"
               MODES_GPU_DEMOD_VARIANT=3 run pytest_gpu_forced_one_kernel_path 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_dropin.py -m gpu -q -p no:cacheprovider -k "not sixty_four" ;;
    fuzz)      run fuzz_parity 900 python tools/fuzz_parity.py 2000 500 ;;
    benchtest) run benchtest 1500 python -m pytest tests/test_gpu_bench.py -m gpu -q --maxfail=6 -p no:cacheprovider ;;
    full8)     run full8 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider -k "not sixty_four" ;;
    full)      run full 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider ;;
    all)       run pytest_gpu_all 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider ;;
    bench)     run bench 900 python bench.py ;;
    bench20)   run bench20 900 python bench.py --steps 20 --warmup 5 ;;
    lowsnr)    run lowsnr 300 python bench.py --workload lowsnr --steps 100
               run lowsnr_v3 300 python bench.py --workload lowsnr --steps 100 --demod-variant 3
               run lowsnr_s2 300 python bench.py --workload lowsnr --steps 100 --leg-streams 2 ;;
    frames)    run frames 300 python bench.py --workload frames --steps 40
               run frames_s2 300 python bench.py --workload frames --steps 40 --leg-streams 2 ;;
    ab_low)    run ab_low 900 python tools/ab_scan.py --rounds 3 --workload lowsnr --demod-variant 2 $AB_LIBS ;;
    ab_noise)  run ab_noise 900 python tools/ab_scan.py --rounds 3 --workload noise $AB_LIBS ;;
    ab_frames) run ab_frames 900 python tools/ab_scan.py --rounds 3 --workload frames --demod-variant 2 $AB_LIBS ;;
    prof)      bash tools/profile.sh ${TAG} noise > "$O/prof_noise.log" 2>&1; tail -n 30 "$O/prof_noise.log" ;;
    prof_low)  bash tools/profile.sh ${TAG}_lowsnr lowsnr > "$O/prof_lowsnr.log" 2>&1; tail -n 60 "$O/prof_lowsnr.log" ;;
    prof_frames) bash tools/profile.sh ${TAG}_frames frames > "$O/prof_frames.log" 2>&1; tail -n 60 "$O/prof_frames.log" ;;
    e2e)       MODES_GPU_CREATE_TRACE=1 run e2e 600 python tools/e2e_cli.py 8 ;;
    dropin)    run dropin 600 python tools/dropin_rate.py ;;
    rccl_init) run rccl_init 300 bash tools/rccl_init_time.sh ;;
    *)         echo "unknown step $s" ;;
  esac
done
for f in "$O"/bench*.log "$O"/noise*.log "$O"/frames*.log "$O"/lowsnr*.log; do
  [ -f "$f" ] && grep '^{' "$f" > "${f%.log}.json"
done
true
