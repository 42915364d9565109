#!/bin/bash
# First contact with an N-GPU node (nothing of this has run on more than one device: every piece was proven on one GPU or on CPU with
# stubs).  The exact sequence for the lease, cheapest and most telling first.  SELF-VERIFYING: every step has its own timeout, its log
# under gpurun_out/first_contact/, prints PASS or FAIL (the md5 / listing comparison is part of the step, not left to the reader) and a
# line that says what a failure means; the script ends with ONE JSON line the driver can keep and exits 0 only when everything passed.
#   tools/first_contact.sh [N]            (default: every GPU of the box; N = 1 runs the same steps over a group of one)
N=${1:-$(python -c "import torch; print(torch.cuda.device_count())")}
R=$PWD
O=$R/gpurun_out/first_contact
mkdir -p "$O"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}      # dmabuf IPC (this pool's hosts support nothing else)
REF_RAW=4a81758c8bec5e45ffa8541c5622938a        # ./dump1090 --ifile testfiles/modes1.bin --raw (padded capture)
REF_STATS=bc3d1c04b24f4989f0fc4a2d1f45abdd      # ... --stats
EXE=dump1090_amd/bin/dump1090_amd
STEPS=()
FAILED=0
record() { local d=${4//\\/\\\\}; d=${d//\"/\\\"}; STEPS+=("{\"step\": \"$1\", \"pass\": $2, \"seconds\": $3, \"detail\": \"$d\"}"); [ "$2" = true ] && echo "   PASS  $1 ($3 s) $4" || { echo "   FAIL  $1 ($3 s) $4"; FAILED=$((FAILED + 1)); }; }
means() { echo "         if this fails: $*"; }
# run <name> <timeout> <command...>: log, wall seconds in $SECS, status in $RC
run() { name=$1; limit=$2; shift 2; echo "== $name: $*"; t0=$(date +%s.%N); timeout $limit "$@" > "$O/$name.out" 2> "$O/$name.err"; RC=$?; SECS=$(python -c "import time; print('%.1f' % (time.time() - $t0))"); }
# the JSON line a bench.py run printed (stdout), field by python expression on `d`
jget() { python -c "
import json, sys
try:
    d = json.loads([l for l in open('$1') if l.startswith('{')][-1])
    print($2)
except Exception as e:
    print('unreadable: %s' % e)"; }

echo "first contact: N = $N, HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY, $(rocm-smi --showcomputepartition 2>/dev/null | grep -m1 'GPU\[' | sed 's/.*: //') partition"
NR=$((N < 2 ? N : 2))

# (first of all the mode that needs nothing from RCCL: N processes, a device each, confirmation and texts through shared memory - if THIS fails
#  the devices or the library are the matter, not the transport)
run rr${N}_small 200 $EXE --ifile tests/golden/modes1.bin --raw --ranks $N --batch-blocks 1 --resolve-on-ranks --timing
md5=$(md5sum < "$O/rr${N}_small.out" | cut -c1-32)
record rr${N}_small $([ $RC = 0 ] && [ "$md5" = $REF_RAW ] && echo true || echo false) $SECS "status $RC, listing md5 $md5 (reference $REF_RAW)"
means "$N PROCESSES, one device each, NO communicator (--resolve-on-ranks): every rank resolves its own batches, rank 0 prints the texts." \
      "A failure here is a device that does not come up or a context that cannot be made on it - nothing RCCL could be blamed for."

run rr${N}_stats 200 $EXE --ifile tests/golden/modes1.bin --stats --ranks $N --batch-blocks 1 --resolve-on-ranks
md5=$(md5sum < "$O/rr${N}_stats.out" | cut -c1-32)
record rr${N}_stats $([ $RC = 0 ] && [ "$md5" = $REF_STATS ] && echo true || echo false) $SECS "status $RC, --stats md5 $md5 (reference $REF_STATS)"
means "the same mode for --stats (round 6): every rank counts its own batches, rank 0 adds the nine counters up - still no communicator."

run ranks${NR}_small 1200 $EXE --ifile tests/golden/modes1.bin --raw --ranks $NR --batch-blocks 1 --timing
md5=$(md5sum < "$O/ranks${NR}_small.out" | cut -c1-32)
record ranks${NR}_small $([ $RC = 0 ] && [ "$md5" = $REF_RAW ] && echo true || echo false) $SECS "status $RC, listing md5 $md5 (reference $REF_RAW)"
means "$NR PROCESS(ES), one device each, 3 batches: the unique id pipe, ncclCommInitRank across processes, the first ncclSend/ncclRecv between two GPUs." \
      "'hipIpcGetMemHandle: invalid argument' = the IPC mode (the host restarts itself once with the other one: look for 'starting over' in the .err);" \
      "'probe' after 300 s = the communicator came up but the first transfer did not complete (xGMI / P2P access between the two devices)."

run ranks${NR}_stats 400 $EXE --ifile tests/golden/modes1.bin --stats --ranks $NR --batch-blocks 1
md5=$(md5sum < "$O/ranks${NR}_stats.out" | cut -c1-32)
record ranks${NR}_stats $([ $RC = 0 ] && [ "$md5" = $REF_STATS ] && echo true || echo false) $SECS "status $RC, --stats md5 $md5 (reference $REF_STATS)"
means "the gather's second list (every rank's preamble positions) - the same transfers as above with a second buffer."

run bench${NR}_frames 1200 python bench.py --gpus $NR --workload frames --frames-mib 1024 --steps 10 $([ $NR = 1 ] && echo --force-gather)
ok=$(jget "$O/bench${NR}_frames.out" "d['listing_check'].get('equals_reference_md5') in (True, None) and d['n_gpus'] == $NR and d['rccl']['p2p_ops_per_step'] > 0")
record bench${NR}_frames $([ $RC = 0 ] && [ "$ok" = True ] && echo true || echo false) $SECS "status $RC, $(jget "$O/bench${NR}_frames.out" "'%.0f Msamples/s, %d msgs per step, rccl_start %s' % (d['value'], d['listing_check']['lines'], d.get('rccl_start'))")"
means "torch.distributed over RCCL: init_process_group, the count all_gather, exact-size isend/irecv of device lists; the listing is checked" \
      "inside (an assertion text names what differs); a start-up failure leaves a {\"rccl_start\": ...} line in the .err."

if [ $N -gt 2 ]; then
run bench${N}_frames 900 python bench.py --gpus $N --workload frames --steps 20
ok=$(jget "$O/bench${N}_frames.out" "d['listing_check'].get('equals_reference_md5') is True and d['n_gpus'] == $N")
record bench${N}_frames $([ $RC = 0 ] && [ "$ok" = True ] && echo true || echo false) $SECS "status $RC, $(jget "$O/bench${N}_frames.out" "'%.0f Msamples/s, listing == the reference md5: %s, rank0_resolve_ms_per_step %s vs kernels %s' % (d['value'], d['listing_check'].get('equals_reference_md5'), d.get('rank0_resolve_ms_per_step'), d.get('kernel_ms_per_step_max_rank'))")"
means "the same at N = $N with BASELINE's 8 GiB per GPU (configs[3]): $((N - 1)) lists per call to rank 0 over $((N - 1)) links; rank0_resolve_ms_per_step" \
      "against kernel_ms_per_step_max_rank says whether rank 0's host half bounds the step."
fi

run bench${N}_frames_rr 900 python bench.py --gpus $N --workload frames --steps 20 --resolve-on ranks $([ $N = 1 ] && echo --force-gather)
ok=$(jget "$O/bench${N}_frames_rr.out" "d['listing_check'].get('equals_reference_md5') in (True, None) and d['n_gpus'] == $N and d['rank_resolve']['steps'] > 0")
record bench${N}_frames_rr $([ $RC = 0 ] && [ "$ok" = True ] && echo true || echo false) $SECS "status $RC, $(jget "$O/bench${N}_frames_rr.out" "'%.0f Msamples/s, rank_resolve %s' % (d['value'], d['rank_resolve'])")"
means "the same leg with every rank resolving its own records (DESIGN.md 5.4): three host all_gathers a step over gloo, the texts over RCCL." \
      "Compare its Msamples/s and rank_resolve.work_ms_per_step with bench${N}_frames above: which mode is the default for N devices is decided here."

run bench${N}_all 1500 python bench.py --gpus $N --steps 20 --warmup 5
ok=$(jget "$O/bench${N}_all.out" "all(d[k]['listing_check'].get('equals_reference_md5') in (True, None) for k in ('frames', 'lowsnr', 'frames_strong')) and d['n_gpus'] == $N")
record bench${N}_all $([ $RC = 0 ] && [ "$ok" = True ] && echo true || echo false) $SECS "status $RC, $(jget "$O/bench${N}_all.out" "'value %.0f Msamples/s over %d GPU(s); frames %.0f, low SNR %.0f, 64 GiB strong %.0f' % (d['value'], d['n_gpus'], d['frames']['Msamples_per_s'], d['lowsnr']['Msamples_per_s'], d['frames_strong']['Msamples_per_s'])")"
means "the driver's command: every leg (noise, frames, low SNR, the 64 GiB stream at every N)."
# round 6: that ONE invocation carries both answers - the strong leg's steps in the given resolve mode and, over the same resident shards, in
# the other, each with its listing check and its scaling_breakdown {slowest rank's kernels, rank 0's resolve, exchange, step, efficiency}
ok=$(jget "$O/bench${N}_all.out" "$N == 1 or ('frames_strong_resolve_on_ranks' in d and d['frames_strong_resolve_on_ranks']['listing_check']['md5'] == d['frames_strong']['listing_check']['md5'])")
record bench${N}_all_both_modes $([ "$ok" = True ] && echo true || echo false) 0 "$(jget "$O/bench${N}_all.out" "'root: %s | ranks: %s | second-pass error: %s' % (d['frames_strong'].get('scaling_breakdown'), d.get('frames_strong_resolve_on_ranks', {}).get('scaling_breakdown'), d.get('frames_strong_other_mode_error'))")"
means "the second pass (resolve_on = ranks) of the strong leg: a failure there is reported in the line (frames_strong_other_mode_error), not fatal;" \
      "the two scaling_breakdown objects say where a sub-linear step went - kernels, rank 0's host half, or the exchange."

if [ -w /dev/shm ]; then
  python - <<'PY'
import sys
sys.path[:0] = [".", "tests"]
import torch, bench
total_blocks = 262144 // 16                          # 4 GiB of configs[3]'s generator (the 64 GiB stream needs 64 GiB of /dev/shm)
dev = torch.device("cuda", 0)
with open("/dev/shm/modes_fc.bin", "wb") as f:       # built on the GPU (the numpy generator takes minutes for this), 1 GiB at a time
    for lo in range(0, total_blocks * 262144, 1 << 30):
        iq, _ = bench.build_frames_shard(torch, dev, total_blocks, lo, lo + (1 << 30), seed=4)
        iq.cpu().numpy().tofile(f)
PY
  one=$($EXE --ifile /dev/shm/modes_fc.bin --raw | md5sum | cut -c1-32)
  st1=$($EXE --ifile /dev/shm/modes_fc.bin --stats | md5sum | cut -c1-32)
  run ranks${N}_file 600 $EXE --ifile /dev/shm/modes_fc.bin --raw --ranks $N --timing
  many=$(md5sum < "$O/ranks${N}_file.out" | cut -c1-32)
  record ranks${N}_file $([ $RC = 0 ] && [ "$one" = "$many" ] && echo true || echo false) $SECS "status $RC, 4 GiB of frames: one process md5 $one, --ranks $N md5 $many"
  means "the C host's one-process-per-GPU mode at full width: batches dealt round-robin, $N lists per round."
  run rr${N}_file 600 $EXE --ifile /dev/shm/modes_fc.bin --raw --ranks $N --resolve-on-ranks --timing
  many=$(md5sum < "$O/rr${N}_file.out" | cut -c1-32)
  record rr${N}_file $([ $RC = 0 ] && [ "$one" = "$many" ] && echo true || echo false) $SECS "status $RC, the same file with --resolve-on-ranks: md5 $many; $(grep -o '"init_s": [0-9.]*, "stream_s": [0-9.]*' "$O/rr${N}_file.err" | tail -1) (gather: $(grep -o '"init_s": [0-9.]*, "stream_s": [0-9.]*' "$O/ranks${N}_file.err" | tail -1))"
  means "the C host without the gather: compare init_s / stream_s with ranks${N}_file (the communicator's start is the difference in init_s)."
  run ranks${N}_file_stats 600 $EXE --ifile /dev/shm/modes_fc.bin --stats --ranks $N
  st=$(md5sum < "$O/ranks${N}_file_stats.out" | cut -c1-32)
  record ranks${N}_file_stats $([ $RC = 0 ] && [ "$st" = "$st1" ] && echo true || echo false) $SECS "status $RC, --stats: one process $st1, --ranks $N $st"
  rm -f /dev/shm/modes_fc.bin
fi
echo "== logs in $O"
echo "{\"first_contact\": {\"n_gpus\": $N, \"ipc_mode\": \"$HSA_ENABLE_IPC_MODE_LEGACY\", \"failed\": $FAILED, \"all_pass\": $([ $FAILED = 0 ] && echo true || echo false), \"steps\": [$(IFS=,; echo "${STEPS[*]}")]}}" | tee "$O/verdict.json"
[ $FAILED = 0 ]
