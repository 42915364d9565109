#!/bin/bash
# How long the communicator of `dump1090_amd --ranks 1` takes to come up (ncclGetUniqueId + ncclCommInitRank + the 64-byte probe),
# under a few RCCL environment settings.  One GPU is enough: the cost is RCCL's own start-up, not the exchange.
#   tools/rccl_init_time.sh          -> one JSON-ish line per setting
F=tests/golden/modes1.bin
run() {
    label=$1; shift
    out=$(env "$@" dump1090_amd/bin/dump1090_amd --ifile $F --raw --ranks 1 --timing 2>&1 >/dev/null | grep '^{' | tail -1)
    echo "$label: $(echo "$out" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["init"], "total_s", j["total_s"])' 2>/dev/null || echo "FAILED $out")"
}
run default X=1
run default_again X=1
run msccl_off RCCL_MSCCL_ENABLE=0
run mscclpp_off RCCL_MSCCLPP_ENABLE=0 RCCL_MSCCL_ENABLE=0
run no_ib NCCL_IB_DISABLE=1 NCCL_SOCKET_IFNAME=lo RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0
run no_probe MODES_GATHER_PROBE_SECONDS=0
run p2p_only NCCL_SHM_DISABLE=1 NCCL_NET_DISABLE=1 RCCL_MSCCL_ENABLE=0
