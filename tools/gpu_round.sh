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
    smoke)  run smoke 300 python __graft_entry__.py smoke ;;
    parity) run parity 900 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -m gpu -q --maxfail=6 -p no:cacheprovider ;;
    full)   run full 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider ;;
    full8)  run full8 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=3 -p no:cacheprovider -k "not sixty_four" ;;
    bench)  run bench 600 python bench.py ;;
    ab)     run ab_overlap0 200 python bench.py --workload noise --no-end-to-end --no-cpu-baseline --overlap 0
            run ab_overlap1 200 python bench.py --workload noise --no-end-to-end --no-cpu-baseline --overlap 1
            run ab_overlap1_s2 200 python bench.py --workload noise --no-end-to-end --no-cpu-baseline --overlap 1 --streams 2
            run ab_depth2 200 python bench.py --workload noise --no-end-to-end --no-cpu-baseline --overlap 1 --depth 2 ;;
    gloo)   run gloo2 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 \
                bench.py --gpus 2 --backend gloo --frames-mib 1024 --steps 5 --warmup 2 --settle 10 ;;
    prof)   run prof 900 bash tools/profile.sh "$TAG" ; mkdir -p "$O/prof"; cp -r gpurun_out/prof_$TAG/summary.txt gpurun_out/prof_$TAG/traffic.json "$O/prof/" 2>/dev/null
            cp gpurun_out/prof_$TAG/kt/*kernel_stats.csv "$O/prof/" 2>/dev/null; find gpurun_out/prof_$TAG -name "*kernel_stats.csv" -exec cp {} "$O/prof/kt_kernel_stats.csv" \; ;;
    e2e)    run e2e 600 python tools/e2e_cli.py 8 ;;
    dense)  run dense 300 python tools/bench_dense.py ;;
  esac
done
echo "== done"
