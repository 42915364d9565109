#!/bin/bash
# Round-5 GPU call: named steps, each with its own timeout and log under gpurun_out/<tag>/.
#   tools/gpu_r5.sh <tag> step [step ...]         (AB_LIBS: the builds tools/ab_scan.py compares)
TAG=${1:-r5}
shift || true
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
L=dump1090_amd
run() { name=$1; shift; echo "== $name"; ( time timeout "$@" ) > "$O/$name.log" 2>&1; echo "   rc=$? $(tail -n 4 "$O/$name.log" | tr '\n' ' ' | cut -c1-400)"; }
for s in "$@"; do
  case $s in
    smoke)     run smoke 400 python -c "import __graft_entry__ as g; g.smoke()" ;;
    parity)    run parity 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider ;;
    parity_win) run parity_win 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "windows or oracle" ;;
    dropin)    run dropin 900 python -m pytest tests/test_dropin.py -m gpu -q -x -p no:cacheprovider ;;
    forced)    MODES_GPU_DEMOD_VARIANT=2 run pytest_gpu_forced_two_kernel_path 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_dropin.py -m gpu -q -p no:cacheprovider -k "not sixty_four"
               MODES_GPU_DEMOD_VARIANT=3 run pytest_gpu_forced_one_kernel_path 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_dropin.py -m gpu -q -p no:cacheprovider -k "not sixty_four" ;;
    fuzz)      run fuzz_parity 900 python tools/fuzz_parity.py 5000 600 ;;
    full8)     run full8 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider -k "not sixty_four" ;;
    all)       run pytest_gpu_all 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider ;;
    bench)     run bench 900 python bench.py ;;
    bench20)   run bench20 900 python bench.py --steps 20 --warmup 5 ;;
    ab_noise)  run ab_noise 900 python tools/ab_scan.py --rounds 3 --workload noise $AB_LIBS ;;
    ab_noise2) run ab_noise2 900 python tools/ab_scan.py --rounds 3 --workload noise --demod-variant 2 $AB_LIBS ;;
    ab_low)    run ab_low 900 python tools/ab_scan.py --rounds 3 --workload lowsnr --demod-variant 2 $AB_LIBS ;;
    ab_frames) run ab_frames 900 python tools/ab_scan.py --rounds 3 --workload frames --demod-variant 2 $AB_LIBS ;;
    trace)     run trace_demod_kernel 300 python tools/trace_demod.py $L/libmodes_gfx950_trace.so 1024 3
               run trace_select_kernel 300 python tools/trace_demod.py $L/libmodes_gfx950_trace.so 1024 2
               [ -f $L/libmodes_gfx950_base_trace.so ] && run trace_demod_kernel_base 300 python tools/trace_demod.py $L/libmodes_gfx950_base_trace.so 1024 3
               [ -f $L/libmodes_gfx950_base_trace.so ] && run trace_select_kernel_base 300 python tools/trace_demod.py $L/libmodes_gfx950_base_trace.so 1024 2 ;;
    prof)      bash tools/profile.sh ${TAG} noise > "$O/prof_noise.log" 2>&1; tail -n 30 "$O/prof_noise.log" ;;
    prof_low)  bash tools/profile.sh ${TAG}_lowsnr lowsnr > "$O/prof_lowsnr.log" 2>&1; tail -n 60 "$O/prof_lowsnr.log" ;;
    prof_frames) bash tools/profile.sh ${TAG}_frames frames > "$O/prof_frames.log" 2>&1; tail -n 60 "$O/prof_frames.log" ;;
    *)         echo "unknown step $s" ;;
  esac
done
for f in "$O"/bench*.log; do
  [ -f "$f" ] && grep '^{' "$f" > "${f%.log}.json"
done
true
