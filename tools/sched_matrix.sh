R=$GRAFT_REPO_ROOT; cd $R
for cfg in "2 4" "2 6" "2 8" "3 6" "3 9" "4 4" "4 8" "4 12" "6 12" "2 4"; do
  set -- $cfg
  python bench.py --workload noise --no-end-to-end --no-live-traffic --no-cpu-baseline --no-ceiling --streams $1 --depth $2 --steps 400 --warmup 5 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('streams $1 depth $2: ms_per_step %.4f  value %.0f  kernel_ms %s' % (d['ms_per_step'], d['value'], d.get('kernel_ms')))"
done
