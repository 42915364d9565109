#!/bin/bash
# One gpurun call's worth of checks, cheapest and most telling first; every step has its own timeout and log.
#   tools/gpu_round.sh <tag> [steps...]      steps: smoke parity full bench ab gloo prof e2e dense (default: all but full)
TAG=${1:-r02}
shift || true
STEPS=${*:-smoke parity bench ab gloo prof e2e dense}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name"; ( time timeout "$@" ) > "$O/$name.log" 2>&1; echo "   rc=$? $(tail -n 3 "$O/$name.log" | tr '\n' ' ' | cut -c1-300)"; }
for s in $STEPS; do
  case $s in
    smoke)  run smoke 300 python __graft_entry__.py smoke ; run smoke2 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    parity) run parity 900 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -m gpu -q --maxfail=6 -p no:cacheprovider ;;
    benchtest) run benchtest 1200 python -m pytest tests/test_gpu_bench.py -m gpu -q --maxfail=6 -p no:cacheprovider ;;
    full)   run full 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider ;;
    full8)  run full8 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider -k "not sixty_four" ;;
    bench)  run bench 600 python bench.py ;;
    ab)     B="python bench.py --workload noise --no-end-to-end --no-live-traffic --no-cpu-baseline --steps 60"
            run ab_s2 200 $B
            MODES_DEMOD_PER_CU=1 run ab_s2_cu1 200 $B
            MODES_DEMOD_PER_CU=1 run ab_s2_cu1_R16 200 $B --run-chunks 16
            MODES_DEMOD_PER_CU=1 run ab_s3_cu1 200 $B --streams 3 --depth 6
            run ab_s3 200 $B --streams 3 --depth 6
            run ab_s1 200 $B --streams 1
            MODES_DEMOD_PER_CU=1 run ab_s1_cu1 200 $B --streams 1 ;;
    gloo)   run gloo2 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 \
                bench.py --gpus 2 --backend gloo --frames-mib 1024 --steps 5 --warmup 2 --settle 10 ;;
    prof)   run prof 900 bash tools/profile.sh "$TAG" ; mkdir -p "$O/prof"; cp -r gpurun_out/prof_$TAG/summary.txt gpurun_out/prof_$TAG/traffic.json "$O/prof/" 2>/dev/null
            cp gpurun_out/prof_$TAG/kt/*kernel_stats.csv "$O/prof/" 2>/dev/null; find gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} "$O/prof/kt_kernel_stats.csv" \; ;;
    e2e)    run e2e 600 python tools/e2e_cli.py 8 ;;
    dense)  run dense 300 python tools/bench_dense.py ;;
    gaps)   B="$R/bench.py --workload noise --no-end-to-end --no-live-traffic --no-cpu-baseline --settle 40 --steps 100"
            ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$O/overlap_kt" -o kt -f csv -- python $B --time-every 100000 > "$O/overlap_kt.log" 2>&1 )
            ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$O/serial_kt" -o kt -f csv -- python $B --time-every 100000 --streams 1 > "$O/serial_kt.log" 2>&1 )
            python tools/kernel_gaps.py "$O/overlap_kt" > "$O/overlap_kt_summary.txt" 2>&1
            python tools/kernel_gaps.py "$O/serial_kt" > "$O/serial_kt_summary.txt" 2>&1 ;;
    trace)  run trace_v0 300 python tools/trace_demod.py dump1090_amd/libmodes_gfx950_trace.so 1024 0
            run trace_v1 300 python tools/trace_demod.py dump1090_amd/libmodes_gfx950_trace.so 1024 1 ;;
  esac
done
echo "== done"
