#!/bin/bash
# Does reading librccl.so sequentially (0.57 GB) before the first communicator shorten a COLD start?  (62 - 435 s have been seen for
# the first ncclCommInitRank of a box: page faults all over a library nobody has read yet.)
#   tools/rccl_cold_preread.sh [nopreread]      -> timings of: the read, the first --ranks 1 run, the second
F=tests/golden/modes1.bin
now() { python3 -c 'import time; print("%.3f" % time.time())'; }
if [ "$1" != "nopreread" ]; then
  t0=$(now); cat /opt/rocm/lib/librccl.so.1 > /dev/null; t1=$(now)
  python3 -c "print('sequential read of /opt/rocm/lib/librccl.so.1: %.2f s' % ($t1 - $t0))"
fi
for label in first again; do
  t0=$(now)
  out=$(dump1090_amd/bin/dump1090_amd --ifile $F --raw --ranks 1 --timing 2>&1 >/dev/null | grep '^{' | tail -1)
  t1=$(now)
  python3 -c "print('$label: wall %.2f s' % ($t1 - $t0))"
  echo "   $out"
done
