cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for s in 2 1; do
rocprofv3 --kernel-trace -d $R/gpurun_out/tl_s$s -o tl -f csv -- python $R/bench.py --workload noise --no-end-to-end --no-live-traffic --no-cpu-baseline --no-ceiling --streams $s --steps 300 --warmup 2 --settle 240 --time-every 100000 > $R/gpurun_out/tl_s$s.log 2>&1
python $R/tools/region_timeline.py $R/gpurun_out/tl_s$s 3 > $R/gpurun_out/tl_s$s.txt 2>&1
python $R/tools/kernel_gaps.py $R/gpurun_out/tl_s$s >> $R/gpurun_out/tl_s$s.txt 2>&1
grep '^{' $R/gpurun_out/tl_s$s.log | cut -c1-400 >> $R/gpurun_out/tl_s$s.txt
done
find $R/gpurun_out/tl_s* -name "*.csv" -size +2M -delete
cat $R/gpurun_out/tl_s2.txt
