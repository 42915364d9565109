R=$GRAFT_REPO_ROOT; cd $R
for ov in 0 1 2 0 1; do
  python bench.py --workload noise --no-end-to-end --no-live-traffic --no-cpu-baseline --no-ceiling --overlap $ov --steps 400 --warmup 5 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('overlap $ov: ms_per_step %.4f' % d['ms_per_step'])"
done
