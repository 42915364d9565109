#!/bin/bash
# End-to-end time of the C host on an 8 GiB file in /dev/shm for several reader-thread counts and batch sizes.
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from dump1090_amd import Demodulator
d = Demodulator(fix=False)
with open("/dev/shm/modes_e2e.bin", "wb") as f:
    for k in range(8):
        iq = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
        d.synth_noise(iq, k << 30, seed=77, sigma_q16=941)
        if k == 7: d.fill(iq[-480:], 127)
        iq.cpu().numpy().tofile(f)
d.close()
PY
TIMEFORMAT="%R s"
for t in 8 8 16; do for b in 128 256 512 1024; do echo -n "threads $t batch $b: "; { time dump1090_amd/bin/dump1090_amd --ifile /dev/shm/modes_e2e.bin --raw --no-fix --read-threads $t --batch-blocks $b > /dev/null; } 2>&1; done; done
rm -f /dev/shm/modes_e2e.bin
