# one vs two launch streams for the record-bearing legs, whole default bench each time, same box:  bash tools/leg_streams_ab.sh
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for ls in 1 2 1 2; do
  python bench.py --leg-streams $ls --no-cpu-baseline --no-end-to-end --no-live-traffic --no-ceiling 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('leg-streams $ls: noise %.4f | frames %.4f  lowsnr %.4f  strong %.4f ms per step | scan %s %s %s' % (d['ms_per_step'], d['frames']['ms_per_step'], d['lowsnr']['ms_per_step'], d['frames_strong']['ms_per_step'], d['frames']['kernel_ms']['scan'], d['lowsnr']['kernel_ms']['scan'], d['frames_strong']['kernel_ms']['scan']))"
done
