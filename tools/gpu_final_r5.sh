R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5final
for v in 0 2 0 2; do
  python bench.py --workload noise --no-end-to-end --no-live-traffic --no-cpu-baseline --no-ceiling --demod-variant $v --steps 300 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('noise demod_variant $v: ms_per_step %.4f kernel_ms %s' % (d['ms_per_step'], {k: d['kernel_ms'][k] for k in ('scan','demod','order')}))"
done > gpurun_out/r5final/noise_variant_ab.txt 2>&1
python bench.py > gpurun_out/r5final/bench_line.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r5final/bench_line_steps20.log 2>&1
( time python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > gpurun_out/r5final/pytest_gpu_all.log 2>&1
cat gpurun_out/r5final/noise_variant_ab.txt; grep "passed\|failed" gpurun_out/r5final/pytest_gpu_all.log
