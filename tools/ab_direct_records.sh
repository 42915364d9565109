#!/bin/bash
# A/B: the record lists of the record-bearing legs reach the host as ONE device-to-host copy behind the kernels (default: lists longer than
# direct_records = 4096) or are written to the pinned host list by record_kernel itself (--direct-records 1048576: no copy at all).
for rep in 1 2; do
for wl in frames strong lowsnr; do
for dr in 0 1048576; do
  python bench.py --workload $wl --no-end-to-end --no-live-traffic --no-cpu-baseline --no-ceiling --direct-records $dr 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$wl direct_records=$dr: ms_per_step %.4f kernel_ms %s fetch %.4f resolve %.4f md5 %s' % (d['ms_per_step'], {k: d['kernel_ms'][k] for k in ('scan', 'demod', 'order')}, d['host_ms_per_call']['fetch'], d['host_ms_per_call']['resolve_per_step'], (d.get('listing_check') or {}).get('equals_reference_md5')))"
done; done; done
