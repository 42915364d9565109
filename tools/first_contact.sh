#!/bin/bash
# First contact with an N-GPU node (nothing of this has run on more than one device: every piece was proven on one GPU or on
# CPU with stubs).  The exact sequence for the lease, cheapest and most telling first; every step has its own timeout, its log
# under gpurun_out/first_contact/, and a line that says what a failure means.
#   tools/first_contact.sh [N]            (default: every GPU of the box)
N=${1:-$(python -c "import torch; print(torch.cuda.device_count())")}
R=$PWD
O=$R/gpurun_out/first_contact
mkdir -p "$O"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}      # dmabuf IPC (this pool's hosts support nothing else)
step() { name=$1; limit=$2; shift 2; echo "== $name: $*"; ( time timeout $limit "$@" ) > "$O/$name.log" 2>&1; rc=$?; echo "   rc=$rc  $(grep -m1 '^{' "$O/$name.log" | cut -c1-300)"; return $rc; }
means() { echo "   -> if this fails: $*"; }

step ranks2_small 400 dump1090_amd/bin/dump1090_amd --ifile tests/golden/modes1.bin --raw --ranks 2 --batch-blocks 1 --timing
means "two PROCESSES, two devices, 3 batches: the unique id pipe, ncclCommInitRank across processes, the first ncclSend/ncclRecv between two GPUs." \
      "'hipIpcGetMemHandle: invalid argument' = the IPC mode (try HSA_ENABLE_IPC_MODE_LEGACY=1); a hang that ends after 120 s with 'probe' = the" \
      "communicator came up but the first transfer did not complete (xGMI / P2P access between the two devices)."
md5=$(dump1090_amd/bin/dump1090_amd --ifile tests/golden/modes1.bin --raw --ranks 2 --batch-blocks 1 2>/dev/null | md5sum | cut -c1-32)
echo "   --ranks 2 listing md5 $md5 (the reference's: 4a81758c8bec5e45ffa8541c5622938a)"

step bench2_frames 600 python bench.py --gpus 2 --workload frames --frames-mib 1024 --steps 10
means "torch.distributed over RCCL with two ranks: init_process_group, the count all_gather, exact-size isend/irecv of device lists;" \
      "the listing is checked against the analytic expectation inside (an assertion text names what differs)."

step bench${N}_frames 900 python bench.py --gpus $N --workload frames --steps 20
means "the same at N = $N with BASELINE's 8 GiB per GPU (configs[3]): 7 lists per call to rank 0 over 7 links; listing == the committed" \
      "reference md5.  rank0_resolve_ms_per_step against kernel_ms_per_step_max_rank says whether rank 0's host half bounds the step."

step bench${N}_all 1500 python bench.py --gpus $N --steps 20 --warmup 5
means "the driver's command: every leg (noise, frames, low SNR, the 64 GiB stream at every N)."

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
  one=$(dump1090_amd/bin/dump1090_amd --ifile /dev/shm/modes_fc.bin --raw | md5sum | cut -c1-32)
  step ranks${N}_file 600 dump1090_amd/bin/dump1090_amd --ifile /dev/shm/modes_fc.bin --raw --ranks $N --timing
  many=$(dump1090_amd/bin/dump1090_amd --ifile /dev/shm/modes_fc.bin --raw --ranks $N 2>/dev/null | md5sum | cut -c1-32)
  echo "   4 GiB of frames: one process md5 $one, --ranks $N md5 $many  $([ "$one" = "$many" ] && echo SAME || echo DIFFERENT)"
  means "the C host's one-process-per-GPU mode at full width: batches dealt round-robin, $N lists per round."
  st=$(dump1090_amd/bin/dump1090_amd --ifile /dev/shm/modes_fc.bin --stats --ranks $N 2>/dev/null | md5sum | cut -c1-32)
  st1=$(dump1090_amd/bin/dump1090_amd --ifile /dev/shm/modes_fc.bin --stats | md5sum | cut -c1-32)
  echo "   --stats: one process $st1, --ranks $N $st  $([ "$st" = "$st1" ] && echo SAME || echo DIFFERENT)"
  rm -f /dev/shm/modes_fc.bin
fi
echo "== done: logs in $O"
