#!/bin/bash
# Round-6 GPU call: named steps, each with its own timeout and log under gpurun_out/<tag>/.
#   tools/gpu_r6.sh <tag> step [step ...]
TAG=${1:-r6}
shift || true
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name"; ( time timeout "$@" ) > "$O/$name.log" 2>&1; echo "   rc=$? $(tail -n 4 "$O/$name.log" | tr '\n' ' ' | cut -c1-500)"; }
for s in "$@"; do
  case $s in
    smoke)     run smoke 400 python -c "import __graft_entry__ as g; g.smoke()" ;;
    parity)    run parity 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider ;;
    all)       run pytest_gpu_all 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider ;;
    resolve)   MODES_HOST_MT_DEBUG=1 run resolve_rate 600 python tools/resolve_rate.py; grep '^{' "$O/resolve_rate.log" > "$O/resolve_rate.json" ;;
    ranks8)    run rank_resolve_8ranks 1200 python -m pytest tests/test_gpu_bench.py -m gpu -q -x -p no:cacheprovider -k eight_ranks -s ;;
    bench)     run bench 900 python bench.py ;;
    bench20)   run bench20 900 python bench.py --steps 20 --warmup 5 ;;
    prof)      bash tools/profile.sh ${TAG} noise > "$O/prof_noise.log" 2>&1; tail -n 30 "$O/prof_noise.log" ;;
    prof_low)  bash tools/profile.sh ${TAG}_lowsnr lowsnr > "$O/prof_lowsnr.log" 2>&1; tail -n 40 "$O/prof_lowsnr.log" ;;
    prof_frames) bash tools/profile.sh ${TAG}_frames frames > "$O/prof_frames.log" 2>&1; tail -n 40 "$O/prof_frames.log" ;;
    prof_strong) bash tools/profile.sh ${TAG}_strong strong > "$O/prof_strong.log" 2>&1; tail -n 40 "$O/prof_strong.log" ;;
    e2e)       run e2e_cli 900 python tools/e2e_cli.py 8 ;;
    pipe)      run pipe_cadence 600 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -m gpu -q -x -p no:cacheprovider -k "pipe or paced" ;;
    fuzz)      run fuzz_parity 900 python tools/fuzz_parity.py 5000 600 ;;
    forced)    run pytest_gpu_forced_and_fuzz 1500 python -m pytest tests/test_gpu_forced.py -m gpu -q -x -p no:cacheprovider ;;
    legs)      for wl in frames lowsnr frames lowsnr; do
                 python bench.py --workload $wl --steps 40 --no-end-to-end --no-live-traffic --no-cpu-baseline --no-ceiling 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$wl: ms_per_step %.4f kernel_ms %s' % (d['ms_per_step'], {k: d['kernel_ms'][k] for k in ('scan', 'demod', 'order')}))"
               done > "$O/legs.txt" 2>&1; cat "$O/legs.txt" ;;
    threads_ab) for rt in 2 4 8 15 2 4 8 15; do
                 python bench.py --workload frames --steps 40 --no-end-to-end --no-live-traffic --no-cpu-baseline --no-ceiling --resolve-threads $rt 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('frames --resolve-threads $rt: ms_per_step %.4f resolve_per_step %.4f fetch %.4f' % (d['ms_per_step'], d['host_ms_per_call']['resolve_per_step'], d['host_ms_per_call']['fetch']))"
               done > "$O/resolve_threads_ab.txt" 2>&1; cat "$O/resolve_threads_ab.txt" ;;
    *)         echo "unknown step $s" ;;
  esac
done
for f in "$O"/bench*.log; do
  [ -f "$f" ] && grep '^{' "$f" > "${f%.log}.json"
done
for d in $R/gpurun_out/prof_${TAG}*; do
  [ -d "$d" ] && mkdir -p "$O/$(basename $d)" && cp "$d"/summary.txt "$d"/traffic.json "$O/$(basename $d)/" 2>/dev/null
  [ -d "$d" ] && find "$d" -name "*kernel_stats.csv" -exec cp {} "$O/$(basename $d)/kt_kernel_stats.csv" \;
done
echo "== done"
