#!/bin/bash
# Runs the CPU tests of the host library and of the header-only device arithmetic against sanitizer builds:
#   tools/sanitize_host.sh                 UBSan, then ASan, then TSan
#   tools/sanitize_host.sh ubsan|asan|tsan|tsan-host|ranks-host|loop-host|pipe-host one of them
# tsan: the multi-threaded resolve (modes_host_resolve_raw_mt: worker pool, speculative pieces) in a C++ harness built
# together with the host sources under -fsanitize=thread, on the records of the reference's capture.
# (-fno-sanitize-recover / abort_on_error: any finding kills the test process).  The host library is built with
# gcc, the shim around modes_core.h needs clang (vector extensions) and its shared sanitizer runtime; under ASan
# python itself has to run with the runtimes preloaded.  The normal builds are restored whatever happens.
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
CLANG=/opt/rocm/lib/llvm/bin/clang++
# the C host: the option parser and its three hosts (dump1090_amd/csrc/Makefile HOST_SRCS)
HOST_SRCS="dump1090_amd/csrc/main.cpp dump1090_amd/csrc/host_single.cpp dump1090_amd/csrc/host_ranks.cpp dump1090_amd/csrc/host_ranks_rccl.cpp dump1090_amd/csrc/host_ranks_shared.cpp"
RT=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
# ubsan / asan replace dump1090_amd/libmodes_host.so and the core shim by sanitizer builds: the normal ones are put back
# whatever happens - through a rename, never by writing into the file (a process that has the library mapped, e.g. the pytest
# run that started this script, would see its pages change)
SAVED=0
save() { cp dump1090_amd/libmodes_host.so /tmp/libmodes_host_orig.so; SAVED=1; }
restore() {
    [ "$SAVED" = 1 ] || return 0
    cp /tmp/libmodes_host_orig.so "$R/dump1090_amd/libmodes_host.so.tmp" && mv "$R/dump1090_amd/libmodes_host.so.tmp" "$R/dump1090_amd/libmodes_host.so"
    (cd "$R" && python -c "import sys; sys.path.insert(0, 'tests'); from native.build import build; build(force=True)")
}
trap restore EXIT
set -e
TESTS="tests/test_host.py tests/test_track.py tests/test_core.py"
run_one() {
    local kind=$1 san
    [ "$SAVED" = 1 ] || save
    if [ "$kind" = ubsan ]; then
        san="-fsanitize=undefined -fno-sanitize-recover=undefined"
    else
        san="-fsanitize=address -fno-omit-frame-pointer"
    fi
    (cd dump1090_amd/csrc && g++ -O1 -g -std=c++17 -fPIC $san -I../../include -shared -o ../libmodes_host.so modes_host.cpp modes_track.cpp -lm)
    (cd tests/native && $CLANG -O1 -g -std=c++17 -fPIC -shared $san -shared-libsan -Wl,-rpath,$RT -I ../../dump1090_amd/csrc -o libcore_shim.so core_shim.cpp)
    echo "== $kind =="
    if [ "$kind" = asan ]; then
        # gcc's runtime first: it owns malloc; python's own leaks are not ours to report
        LD_PRELOAD="$(gcc -print-file-name=libasan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 \
            python -m pytest tests/test_host.py tests/test_track.py -x -q -p no:cacheprovider
    else
        python -m pytest $TESTS -x -q -p no:cacheprovider
    fi
}
run_tsan() {
    echo "== tsan =="
    python - <<'PY'
import sys
sys.path[:0] = [".", "tests", "oracle"]
import numpy as np, synth
from helpers import oracle_records
recs, _ = oracle_records(synth.modes1_padded("tests/golden/modes1.bin"), 1)
big = np.tile(recs, 12)
nb = int(recs["block"].max()) + 1
for r in range(12):
    big["block"][r * recs.size:(r + 1) * recs.size] += r * nb
big.tofile("/tmp/modes_mt_records.bin")
PY
    g++ -O1 -g -std=c++17 -fsanitize=thread -Iinclude -o /tmp/modes_mt_harness tests/native/mt_harness.cpp \
        dump1090_amd/csrc/modes_host.cpp dump1090_amd/csrc/modes_track.cpp -lpthread -lm
    TSAN_OPTIONS=halt_on_error=1 /tmp/modes_mt_harness /tmp/modes_mt_records.bin
}
# tsan-host: the C host itself (dump1090_amd/csrc/host_single.cpp: reader thread, resolver thread, lanes of two "devices" handed
# between them) under ThreadSanitizer, the GPU library replaced by tests/native/gpu_stub.cpp (the oracle's stateless
# functions behind the same entry points) - listing and --stats must be the reference's, and no race may be reported.
run_tsan_host() {
    echo "== tsan-host =="
    gcc -O1 -g -fsanitize=thread -c -o /tmp/modes_oracle_tsan.o oracle/modes_oracle.c
    g++ -O1 -g -std=c++17 -fsanitize=thread -Iinclude -o /tmp/dump1090_amd_tsan $HOST_SRCS tests/native/gpu_stub.cpp \
        dump1090_amd/csrc/modes_host.cpp dump1090_amd/csrc/modes_track.cpp /tmp/modes_oracle_tsan.o -lpthread -lm -ldl
    # the reference's md5s of BASELINE.md section 4 (--raw, --stats, --onlyaddr)
    check() { want=$1; shift
        TSAN_OPTIONS=halt_on_error=1 /tmp/dump1090_amd_tsan --ifile tests/golden/modes1.bin "$@" > /tmp/tsan_host.out
        got=$(md5sum < /tmp/tsan_host.out | cut -c1-32)
        echo "   $*: $(wc -l < /tmp/tsan_host.out) lines, md5 $got"
        [ "$got" = "$want" ] || { echo "   expected $want"; exit 1; }
    }
    check 4a81758c8bec5e45ffa8541c5622938a --raw
    check 4a81758c8bec5e45ffa8541c5622938a --raw --gpu-list 0,0 --batch-blocks 1 --depth 2
    check bc3d1c04b24f4989f0fc4a2d1f45abdd --stats --gpu-list 0,0,0 --batch-blocks 2
    check bab0f055e262e216208a5cbbdf63fe24 --onlyaddr --gpus 2 --batch-blocks 1 --read-threads 3
}
# ranks-host: the one-process-per-GPU mode of the C host (dump1090_amd --ranks N) with N = 1, 2, 3 PROCESSES on this machine:
# the GPU stub above plus tests/native/gather_stub.cpp (include/modes_gather.h over shared memory instead of RCCL, built as
# the libmodes_gather.so the host dlopens) - fork, id pipes, round-robin batches, gather rounds, ranks without a batch, the EOF
# batch; stdout must be the reference's for every N and batch size.
run_ranks_host() {
    export MODES_RANKS_QUIET=1                    # (the '--gpus N is faster below 96 GB' line: once per run is enough for a log)
    echo "== ranks-host =="
    D=/tmp/modes_ranks_host
    mkdir -p $D
    gcc -O1 -g -c -o $D/modes_oracle.o oracle/modes_oracle.c
    g++ -O1 -g -std=c++17 -DMODES_TEST_HOOKS -Iinclude -o $D/dump1090_amd_stub $HOST_SRCS tests/native/gpu_stub.cpp \
        dump1090_amd/csrc/modes_host.cpp dump1090_amd/csrc/modes_track.cpp $D/modes_oracle.o -lpthread -lm -ldl -rdynamic
    g++ -O1 -g -std=c++17 -fPIC -shared -Iinclude -o $D/libmodes_gather.so tests/native/gather_stub.cpp -lpthread -lrt
    for n in 1 2 3; do for bb in 1 2 5; do
        got=$($D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw --ranks $n --batch-blocks $bb | md5sum | cut -c1-32)
        echo "   --ranks $n --batch-blocks $bb: md5 $got"
        [ "$got" = 4a81758c8bec5e45ffa8541c5622938a ] || { echo "   expected 4a81758c8bec5e45ffa8541c5622938a"; exit 1; }
    done; done
    got=$($D/dump1090_amd_stub --ifile tests/golden/modes1.bin --onlyaddr --ranks 3 --batch-blocks 1 | md5sum | cut -c1-32)
    [ "$got" = bab0f055e262e216208a5cbbdf63fe24 ] || { echo "   --onlyaddr: $got"; exit 1; }
    echo "   --onlyaddr --ranks 3: md5 $got"
    # --stats: every rank's preamble positions travel to rank 0 as the gather's second list (dump1090.c:2993-3006 needs them all)
    for n in 1 2 3; do for bb in 1 2; do
        got=$($D/dump1090_amd_stub --ifile tests/golden/modes1.bin --stats --ranks $n --batch-blocks $bb | md5sum | cut -c1-32)
        echo "   --stats --ranks $n --batch-blocks $bb: md5 $got"
        [ "$got" = bc3d1c04b24f4989f0fc4a2d1f45abdd ] || { echo "   expected bc3d1c04b24f4989f0fc4a2d1f45abdd"; exit 1; }
    done; done
    # eight ranks, three batches: five ranks never have a batch (every round they report an empty list and still take part)
    for mode in --raw --stats; do
        got=$($D/dump1090_amd_stub --ifile tests/golden/modes1.bin $mode --ranks 8 --batch-blocks 1 | md5sum | cut -c1-32)
        echo "   $mode --ranks 8 --batch-blocks 1: md5 $got"
        case $mode in --raw) want=4a81758c8bec5e45ffa8541c5622938a ;; --stats) want=bc3d1c04b24f4989f0fc4a2d1f45abdd ;; esac
        [ "$got" = "$want" ] || { echo "   expected $want for $mode"; exit 1; }
    done
    # a second list that outgrows its buffers fails every rank together (status 1), it is not truncated
    set +e
    timeout 20 $D/dump1090_amd_stub --ifile tests/golden/modes1.bin --stats --ranks 2 --batch-blocks 1 --gather-candidates 8 > /dev/null 2> $D/fail.err
    rc=$?
    set -e
    echo "   --stats with 8 positions of room: exit status $rc"
    [ "$rc" = 1 ] && grep -q "exceeds the gather buffers\|gather" $D/fail.err || { cat $D/fail.err; exit 1; }
    # a pipe or --loop has ONE reader: rank 0 reads and hands every rank its batches through shared memory (round 5; refused before)
    for n in 1 2 3; do for bb in 1 2; do
        got=$($D/dump1090_amd_stub --ifile - --raw --ranks $n --batch-blocks $bb < tests/golden/modes1.bin | md5sum | cut -c1-32)
        echo "   --ifile - --ranks $n --batch-blocks $bb: md5 $got"
        [ "$got" = 4a81758c8bec5e45ffa8541c5622938a ] || { echo "   expected 4a81758c8bec5e45ffa8541c5622938a"; exit 1; }
    done; done
    got=$(cat tests/golden/modes1.bin | $D/dump1090_amd_stub --ifile - --stats --ranks 2 --batch-blocks 1 | md5sum | cut -c1-32)
    [ "$got" = bc3d1c04b24f4989f0fc4a2d1f45abdd ] || { echo "   --ifile - --stats --ranks 2: $got"; exit 1; }
    echo "   --ifile - --stats --ranks 2: md5 $got"
    # --loop: the first 2.5 laps are the single-process host's (which are the reference's: loop-host below)
    python - <<'PY'
import sys
sys.path[:0] = [".", "tests", "oracle"]
import synth
synth.modes1_padded("tests/golden/modes1.bin").tofile("/tmp/modes_ranks_host/pad.bin")
PY
    one=$($D/dump1090_amd_stub --ifile $D/pad.bin --raw | wc -c)
    n=$(( one * 5 / 2 ))
    set +e
    timeout 60 $D/dump1090_amd_stub --ifile $D/pad.bin --raw --loop --batch-blocks 1 2> /dev/null | head -c $n > $D/loop_1proc.txt
    for r in 2 3; do
        timeout 60 $D/dump1090_amd_stub --ifile $D/pad.bin --raw --loop --ranks $r --batch-blocks 1 2> /dev/null | head -c $n > $D/loop_ranks$r.txt
    done
    set -e
    cmp $D/loop_1proc.txt $D/loop_ranks2.txt && cmp $D/loop_1proc.txt $D/loop_ranks3.txt || { echo "   --loop --ranks differs from the one-process replay"; exit 1; }
    echo "   --loop --ranks 2 / 3: the first $n bytes (2.5 laps) equal the one-process host's"
    # --resolve-on-ranks: every rank resolves its own batches from a guessed whitelist, the ranks confirm each other through shared memory,
    # rank 0 prints their texts (no gather library is loaded at all: the stub is moved away for these runs) - the same listing for every N and
    # batch size, from a file, a pipe and a replay; with every rank but the first started from a wrong state on purpose (MODES_RR_SPOIL: the
    # logged answers do not hold, the rank resolves again); ranks without a batch; a mode it does not serve is refused
    mv $D/libmodes_gather.so $D/libmodes_gather.so.away
    for k in 1 2 3 8; do for bb in 1 2; do
        got=$($D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw --ranks $k --batch-blocks $bb --resolve-on-ranks | md5sum | cut -c1-32)
        echo "   --resolve-on-ranks --ranks $k --batch-blocks $bb: md5 $got"
        [ "$got" = 4a81758c8bec5e45ffa8541c5622938a ] || { echo "   expected 4a81758c8bec5e45ffa8541c5622938a"; exit 1; }
    done; done
    got=$($D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw --ranks 3 --batch-blocks 1 --resolve-on-ranks --clean-exit | md5sum | cut -c1-32)
    echo "   --resolve-on-ranks, orderly teardown (--clean-exit): md5 $got"
    [ "$got" = 4a81758c8bec5e45ffa8541c5622938a ] || { echo "   expected 4a81758c8bec5e45ffa8541c5622938a"; exit 1; }
    for k in 2 3; do
        got=$(MODES_RR_SPOIL=1 $D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw --ranks $k --batch-blocks 1 --resolve-on-ranks --timing 2> $D/rr_spoil.err | md5sum | cut -c1-32)
        echo "   --resolve-on-ranks --ranks $k, wrong starts: md5 $got, $(grep -o '"reruns": [0-9]*' $D/rr_spoil.err)"
        [ "$got" = 4a81758c8bec5e45ffa8541c5622938a ] || { echo "   expected 4a81758c8bec5e45ffa8541c5622938a"; exit 1; }
    done
    # ... and a stream on which a wrong start SHOWS: an aircraft's DF17 in buffer 0, a DF4 and a DF5 whose parity only validates against
    # that address at the head of buffers 1 and 2 (dump1090.c:942-983) - a rank started from an empty whitelist logs "unknown", the state the
    # ranks before it really left says "known": it must resolve again, and the listing must still be the one-process host's
    python - <<'PY'
import sys
sys.path[:0] = [".", "tests", "oracle"]
import numpy as np, synth as sy
iq = sy.noise_bytes(31, 0, 3 * 262144, sigma_q16=300)
a17 = sy.make_frame(17, sy._payload(1, 14, 1))
addr = int.from_bytes(a17[1:4], "big")
sy.add_frame(iq, 1000, a17, 70, 3)
sy.add_frame(iq, 131072 + 900, sy.make_frame(4, sy._payload(2, 7, 2), xor_parity=addr), 70, 9)
sy.add_frame(iq, 131072 + 9000, a17, 70, 5)
sy.add_frame(iq, 2 * 131072 + 700, sy.make_frame(5, sy._payload(2, 7, 3), xor_parity=addr), 70, 1)
sy.add_frame(iq, 2 * 131072 + 5000, sy.make_frame(20, sy._payload(2, 14, 4), xor_parity=addr), 70, 1)
sy.finish_stream(iq).tofile("/tmp/modes_ranks_host/ap.bin")
PY
    want=$($D/dump1090_amd_stub --ifile $D/ap.bin --raw | md5sum | cut -c1-32)
    lines=$($D/dump1090_amd_stub --ifile $D/ap.bin --raw | wc -l)
    # (524f28a5...: what the unmodified reference prints for this stream, oracle/_ref/dump1090_ref --ifile ap.bin --raw)
    [ "$lines" = 5 ] && [ "$want" = 524f28a5613c2468123104e0a17e61f8 ] || { echo "   ap.bin: $lines lines, md5 $want from the one-process host; 5 lines, 524f28a5... expected"; exit 1; }
    for k in 2 3; do for spoil in "" 1; do
        got=$(env ${spoil:+MODES_RR_SPOIL=1} $D/dump1090_amd_stub --ifile $D/ap.bin --raw --ranks $k --batch-blocks 1 --resolve-on-ranks --timing 2> $D/rr_ap.err | md5sum | cut -c1-32)
        re=$(grep -o '"reruns": [0-9]*' $D/rr_ap.err | cut -d' ' -f2)
        echo "   --resolve-on-ranks --ranks $k on AP-validated frames${spoil:+, wrong starts}: md5 $got, $re re-run(s)"
        [ "$got" = "$want" ] || { echo "   expected $want"; exit 1; }
        if [ -n "$spoil" ]; then [ "$re" -ge $((k - 1)) ] || { echo "   a wrong start went unnoticed"; exit 1; }; else [ "$re" = 0 ] || { echo "   a right guess was rejected"; exit 1; }; fi
    done; done
    got=$($D/dump1090_amd_stub --ifile - --raw --ranks 3 --batch-blocks 1 --resolve-on-ranks < tests/golden/modes1.bin | md5sum | cut -c1-32)
    echo "   --resolve-on-ranks --ifile - --ranks 3: md5 $got"
    [ "$got" = 4a81758c8bec5e45ffa8541c5622938a ] || { echo "   expected 4a81758c8bec5e45ffa8541c5622938a"; exit 1; }
    set +e
    for r in 2 3; do
        timeout 60 $D/dump1090_amd_stub --ifile $D/pad.bin --raw --loop --ranks $r --batch-blocks 1 --resolve-on-ranks 2> /dev/null | head -c $n > $D/loop_rr$r.txt
    done
    $D/dump1090_amd_stub --ifile tests/golden/modes1.bin --sbs --ranks 2 --resolve-on-ranks > /dev/null 2> $D/rr_refused.err
    rc=$?
    set -e
    cmp $D/loop_1proc.txt $D/loop_rr2.txt && cmp $D/loop_1proc.txt $D/loop_rr3.txt || { echo "   --loop --resolve-on-ranks differs from the one-process replay"; exit 1; }
    echo "   --resolve-on-ranks --loop --ranks 2 / 3: the first $n bytes (2.5 laps) equal the one-process host's"
    echo "   --resolve-on-ranks --sbs: exit status $rc"
    [ "$rc" = 1 ] && grep -q "needs the aircraft table in stream order" $D/rr_refused.err || { cat $D/rr_refused.err; exit 1; }
    # round 6: the other sinks whose output is a function of the message (--onlyaddr, --raw-net) or a sum over the batches (--stats: every
    # rank counts its own batches - preamble positions included - and rank 0 adds the nine counters up), right and wrong starts, a pipe
    for k in 1 2 3 8; do for spoil in "" 1; do
        for mode in --stats --onlyaddr --raw-net; do
            case $mode in --stats) want=bc3d1c04b24f4989f0fc4a2d1f45abdd ;; --onlyaddr) want=bab0f055e262e216208a5cbbdf63fe24 ;;
                          --raw-net) want=$($D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw-net | md5sum | cut -c1-32) ;; esac
            got=$(env ${spoil:+MODES_RR_SPOIL=1} $D/dump1090_amd_stub --ifile tests/golden/modes1.bin $mode --ranks $k --batch-blocks 1 --resolve-on-ranks | md5sum | cut -c1-32)
            [ "$got" = "$want" ] || { echo "   $mode --ranks $k --resolve-on-ranks${spoil:+ (wrong starts)}: $got, want $want"; exit 1; }
        done
        echo "   --stats / --onlyaddr / --raw-net --ranks $k --resolve-on-ranks${spoil:+, wrong starts}: the one-process host's output"
    done; done
    wants=$($D/dump1090_amd_stub --ifile $D/ap.bin --stats | md5sum | cut -c1-32)
    for k in 2 3; do for spoil in "" 1; do
        got=$(env ${spoil:+MODES_RR_SPOIL=1} $D/dump1090_amd_stub --ifile $D/ap.bin --stats --ranks $k --batch-blocks 1 --resolve-on-ranks --timing 2> $D/rr_ap.err | md5sum | cut -c1-32)
        re=$(grep -o '"reruns": [0-9]*' $D/rr_ap.err | cut -d' ' -f2)
        echo "   --stats --resolve-on-ranks --ranks $k on AP-validated frames${spoil:+, wrong starts}: md5 $got, $re re-run(s)"
        [ "$got" = "$wants" ] || { echo "   expected $wants (a repeated resolve must not count twice)"; exit 1; }
        if [ -n "$spoil" ]; then [ "$re" -ge $((k - 1)) ] || { echo "   a wrong start went unnoticed"; exit 1; }; fi
    done; done
    got=$(cat tests/golden/modes1.bin | $D/dump1090_amd_stub --ifile - --stats --ranks 3 --batch-blocks 1 --resolve-on-ranks | md5sum | cut -c1-32)
    [ "$got" = bc3d1c04b24f4989f0fc4a2d1f45abdd ] || { echo "   --ifile - --stats --ranks 3 --resolve-on-ranks: $got"; exit 1; }
    echo "   --ifile - --stats --ranks 3 --resolve-on-ranks: md5 $got"
    mv $D/libmodes_gather.so.away $D/libmodes_gather.so
    # a rank whose GPU does not come up while its peers already wait in the gather: the job ends with status 1, it does not hang
    # (rank 0's watchdog kills the other ranks; a rank never outlives rank 0)
    for bad in 0 1 2; do
        set +e
        MODES_STUB_FAIL_DEVICE=$bad timeout 20 $D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw --ranks 3 --batch-blocks 1 > /dev/null 2> $D/fail.err
        rc=$?
        set -e
        echo "   rank $bad fails to start: exit status $rc"
        [ "$rc" = 1 ] || { cat $D/fail.err; exit 1; }
    done
    # the start-up probe: the first transfer over the new communicator does not complete (on every rank / on one peer only /
    # on rank 0 only) - rank 0 starts the job over ONCE with the other IPC mode, and that run prints the listing
    for who in all 1 0; do
        got=$(MODES_STUB_PROBE_FAIL=$who timeout 60 $D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw --ranks 3 --batch-blocks 1 2> $D/probe.err | md5sum | cut -c1-32)
        echo "   probe fails on rank(s) $who: md5 $got, $(grep -c 'starting over with' $D/probe.err) restart"
        [ "$got" = 4a81758c8bec5e45ffa8541c5622938a ] && [ "$(grep -c 'starting over with' $D/probe.err)" = 1 ] || { cat $D/probe.err; exit 1; }
    done
    # a rank that fails MID-STREAM (two ranks, three batches: rank 0's second GPU call / rank 1's only one): the peer is already
    # in that round's exchange, nobody tears an RCCL communicator down on that path (host_ranks.cpp run_ranks) - status 1, promptly
    for bad in 0:1 1:0; do
        set +e
        MODES_STUB_FAIL_SUBMIT=$bad timeout 20 $D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw --ranks 2 --batch-blocks 1 --depth 4 > /dev/null 2> $D/fail.err
        rc=$?
        set -e
        echo "   rank ${bad%%:*} fails in GPU call ${bad##*:}: exit status $rc"
        [ "$rc" = 1 ] || { cat $D/fail.err; exit 1; }
    done
    # a rank KILLED between its guess and its final tables (round 6): its peers wait for final_seq in shared memory, nobody
    # sets `failed` - rank 0's watchdog (or, rank 0 being the one, the parent-death signal) must end the job within a second
    for die in 1:0 2:0 0:0; do
        set +e
        t0=$(date +%s.%N)
        MODES_RR_DIE=$die timeout 20 $D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw --ranks 3 --batch-blocks 1 --resolve-on-ranks > /dev/null 2> $D/fail.err
        rc=$?
        dt=$(python -c "import time,sys; print('%.2f' % (time.time() - float(sys.argv[1])))" $t0)
        set -e
        echo "   resolving on the ranks, rank ${die%%:*} killed in round ${die##*:} between guess and final: status $rc after $dt s"
        [ "$rc" != 0 ] && [ "$rc" != 124 ] && python -c "import sys; sys.exit(0 if float(sys.argv[1]) < 2.0 else 1)" $dt || { cat $D/fail.err; exit 1; }
    done
    # the same under --resolve-on-ranks: the peers wait for that rank's tables in shared memory - they must see the failure and end
    for bad in 0:1 1:0; do
        set +e
        MODES_STUB_FAIL_SUBMIT=$bad timeout 20 $D/dump1090_amd_stub --ifile tests/golden/modes1.bin --raw --ranks 2 --batch-blocks 1 --resolve-on-ranks > /dev/null 2> $D/fail.err
        rc=$?
        set -e
        echo "   resolving on the ranks, rank ${bad%%:*} fails in its GPU call ${bad##*:}: status $rc"
        [ "$rc" = 1 ] || { cat $D/fail.err; exit 1; }
    done
}
# loop-host: --loop (dump1090.c:488-494: at the end of the file the reader seeks back and keeps filling the SAME buffer - the
# stream is the file repeated for ever, the whitelist carries over from lap to lap) and --clean-exit, on the stubbed host: the
# first 2.5 laps' worth of output bytes must be the unmodified reference's (where oracle/_ref/dump1090_ref exists: this
# container and the GPU box), and in any case start with one lap's listing.
run_loop_host() {
    echo "== loop-host =="
    D=/tmp/modes_loop_host
    mkdir -p $D
    gcc -O1 -g -c -o $D/modes_oracle.o oracle/modes_oracle.c
    g++ -O1 -g -std=c++17 -DMODES_TEST_HOOKS -Iinclude -o $D/dump1090_amd_stub $HOST_SRCS tests/native/gpu_stub.cpp \
        dump1090_amd/csrc/modes_host.cpp dump1090_amd/csrc/modes_track.cpp $D/modes_oracle.o -lpthread -lm -ldl -rdynamic
    python - <<'PY'
import sys
sys.path[:0] = [".", "tests", "oracle"]
import synth
synth.modes1_padded("tests/golden/modes1.bin").tofile("/tmp/modes_loop_host/pad.bin")
PY
    one=$($D/dump1090_amd_stub --ifile $D/pad.bin --raw --clean-exit | tee $D/one.txt | md5sum | cut -c1-32)
    echo "   one lap, --clean-exit: md5 $one"
    [ "$one" = 4a81758c8bec5e45ffa8541c5622938a ] || exit 1
    # --clean-exit on a file of many batches: the pages of the mapping are handed back batch by batch while the stream runs, and the
    # orderly teardown must unmap only what is left (unmapping the whole original range again takes whatever was
    # allocated in the freed holes since - output buffers, lanes - away from under its owners)
    python - <<'PY'
import numpy as np
one = np.fromfile("/tmp/modes_loop_host/pad.bin", dtype=np.uint8)
np.tile(one, 48).tofile("/tmp/modes_loop_host/pad48.bin")
PY
    for bb in 2 7; do
        $D/dump1090_amd_stub --ifile $D/pad48.bin --raw --clean-exit --batch-blocks $bb --depth 3 > $D/many.txt || { echo "   --clean-exit on 48 laps: status $?"; exit 1; }
        many=$(wc -l < $D/many.txt)
        head -c $(wc -c < $D/one.txt) $D/many.txt | cmp - $D/one.txt || { echo "   --clean-exit, 48 laps in one file: the first lap differs"; exit 1; }
        echo "   --clean-exit, 48 laps as one file, --batch-blocks $bb: $many lines, status 0"
    done
    # ... and at a size where it MATTERS: a 1 GiB sparse file (zeros: no preamble anywhere) with the capture at its end, under
    # AddressSanitizer, 64 batches of 16 MiB over six lanes that are set up while the stream already runs - the unmapper has handed ~1 GiB of
    # the mapping back by the time the last lanes' buffers, the output and the pools are allocated (some of them INTO the freed range), and the
    # orderly teardown unmaps what is left.  Unmapping the whole original range there would pull memory from under its owners: ASan or a
    # segfault says so.  The output must be that of the default exit path (which unmaps nothing).
    g++ -O1 -g -std=c++17 -DMODES_TEST_HOOKS -fsanitize=address -fno-omit-frame-pointer -Iinclude -o $D/dump1090_amd_asan $HOST_SRCS tests/native/gpu_stub.cpp \
        dump1090_amd/csrc/modes_host.cpp dump1090_amd/csrc/modes_track.cpp $D/modes_oracle.o -lpthread -lm -ldl -rdynamic
    python - <<'PY'
import numpy as np
pad = np.fromfile("/tmp/modes_loop_host/pad.bin", dtype=np.uint8)
with open("/tmp/modes_loop_host/sparse.bin", "wb") as f:
    f.truncate(1 << 30)
    f.seek((1 << 30) - pad.size)
    f.write(pad.tobytes())
PY
    $D/dump1090_amd_stub --ifile $D/sparse.bin --raw --batch-blocks 64 > $D/sparse_want.txt
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 $D/dump1090_amd_asan --ifile $D/sparse.bin --raw --clean-exit --batch-blocks 64 --depth 3 --gpu-list 0,0 > $D/sparse_got.txt \
        || { echo "   --clean-exit on a 1 GiB sparse file under ASan: status $?"; exit 1; }
    cmp $D/sparse_want.txt $D/sparse_got.txt || { echo "   --clean-exit on a 1 GiB sparse file: the output differs from the default exit path's"; exit 1; }
    echo "   --clean-exit, 1 GiB sparse file + capture, 64 batches over six lanes, under ASan: $(wc -l < $D/sparse_got.txt) lines = the default exit path's, status 0"
    rm -f $D/sparse.bin
    n=$(( $(wc -c < $D/one.txt) * 5 / 2 ))
    set +e
    for bb in 1 3 512; do
        timeout 60 $D/dump1090_amd_stub --ifile $D/pad.bin --raw --loop --batch-blocks $bb 2> /dev/null | head -c $n > $D/loop_$bb.txt
    done
    set -e
    head -c $(wc -c < $D/one.txt) $D/loop_1.txt | cmp - $D/one.txt || { echo "   the first lap differs from a plain run"; exit 1; }
    cmp $D/loop_1.txt $D/loop_3.txt && cmp $D/loop_1.txt $D/loop_512.txt || { echo "   the replay depends on the batch size"; exit 1; }
    echo "   --loop: $n bytes (2.5 laps), the same for --batch-blocks 1, 3, 512; first lap = the plain listing; md5 $(md5sum < $D/loop_1.txt | cut -c1-32)"
    if [ -x oracle/_ref/dump1090_ref ]; then
        set +e
        LD_PRELOAD=$PWD/oracle/_ref/libfixedtime.so timeout 60 oracle/_ref/dump1090_ref --ifile $D/pad.bin --raw --loop 2> /dev/null | head -c $n > $D/loop_ref.txt
        set -e
        cmp $D/loop_1.txt $D/loop_ref.txt || { echo "   --loop differs from the reference's"; exit 1; }
        echo "   --loop == oracle/_ref/dump1090_ref --loop over the same 2.5 laps"
    fi
}
# pipe-host: a pipe is served at the pace it delivers (host_common.h read_paced; dump1090.c:460-512, :2969-2990 print a buffer's messages
# within that buffer): the reference's capture written one 256 KiB buffer every 150 ms into `dump1090_amd --ifile -` with the DEFAULT
# batch size (512 buffers: a host that waits for a full batch prints nothing before the writer is done) - the first line must be out
# before the writer has finished, the listing must be the file run's; the same through --ranks 2 (rank 0's reader deals the batches
# out) in both resolve modes, and unpaced (cat |): full speed, same bytes.
run_pipe_host() {
    export MODES_RANKS_QUIET=1
    echo "== pipe-host =="
    D=/tmp/modes_pipe_host
    mkdir -p $D
    gcc -O1 -g -c -o $D/modes_oracle.o oracle/modes_oracle.c
    g++ -O1 -g -std=c++17 -DMODES_TEST_HOOKS -Iinclude -o $D/dump1090_amd_stub $HOST_SRCS tests/native/gpu_stub.cpp \
        dump1090_amd/csrc/modes_host.cpp dump1090_amd/csrc/modes_track.cpp $D/modes_oracle.o -lpthread -lm -ldl -rdynamic
    g++ -O1 -g -std=c++17 -fPIC -shared -Iinclude -o $D/libmodes_gather.so tests/native/gather_stub.cpp -lpthread -lrt
    check() { what=$1; want=$2; shift 2
        python tools/paced_pipe.py 150 tests/golden/modes1.bin --repeat 4 -- "$@" > $D/paced.json || { cat $D/paced.json; echo "   $what: the host failed"; exit 1; }
        python - "$what" "$want" $D/paced.json <<'PY'
import json, sys
what, want, d = sys.argv[1], sys.argv[2], json.load(open(sys.argv[3]))
ok = d["md5"] == want and d["first_output_s"] is not None and d["first_output_s"] < d["writer_done_s"] - 0.2
print("   %s: first output after %.2f s, writer done after %.2f s (%d buffers, one per %g ms), %d lines, md5 %s%s" % (
    what, d["first_output_s"] or -1, d["writer_done_s"], d["buffers"], d["ms_per_buffer"], d["lines"], d["md5"], "" if ok else "  <-- FAIL (want %s)" % want))
sys.exit(0 if ok else 1)
PY
    }
    python - <<'PY'
import numpy as np
one = np.fromfile("tests/golden/modes1.bin", dtype=np.uint8)
np.tile(one, 4).tofile("/tmp/modes_pipe_host/four.bin")
PY
    want=$($D/dump1090_amd_stub --ifile $D/four.bin --raw | md5sum | cut -c1-32)
    echo "   the file run (4 x the reference's capture): md5 $want"
    check "paced pipe, one process" $want $D/dump1090_amd_stub --ifile - --raw
    check "paced pipe, --ranks 2" $want $D/dump1090_amd_stub --ifile - --raw --ranks 2
    check "paced pipe, --ranks 2 --resolve-on-ranks" $want $D/dump1090_amd_stub --ifile - --raw --ranks 2 --resolve-on-ranks
    wants=$($D/dump1090_amd_stub --ifile $D/four.bin --stats | md5sum | cut -c1-32)
    python tools/paced_pipe.py 20 tests/golden/modes1.bin --repeat 4 -- $D/dump1090_amd_stub --ifile - --stats > $D/paced.json
    grep -q "\"md5\": \"$wants\"" $D/paced.json || { cat $D/paced.json; echo "   --stats through a paced pipe differs from the file run ($wants)"; exit 1; }
    echo "   paced pipe, --stats: md5 $wants (the file run's)"
    for fl in 0 5 66 1000; do
        got=$(cat $D/four.bin | $D/dump1090_amd_stub --ifile - --raw --flush-ms $fl --batch-blocks 3 | md5sum | cut -c1-32)
        [ "$got" = "$want" ] || { echo "   cat | --flush-ms $fl: $got, want $want"; exit 1; }
    done
    echo "   unpaced pipe (cat |), --flush-ms 0 / 5 / 66 / 1000, --batch-blocks 3: md5 $want"
}
case "${1:-all}" in
    pipe-host) run_pipe_host ;;
    ubsan) run_one ubsan ;;
    asan)  run_one asan ;;
    tsan)  run_tsan ;;
    tsan-host) run_tsan_host ;;
    ranks-host) run_ranks_host ;;
    loop-host) run_loop_host ;;
    *)     run_one ubsan; run_one asan; run_tsan; run_tsan_host; run_ranks_host; run_loop_host; run_pipe_host ;;
esac
