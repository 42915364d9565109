"""CPU tests of the host half (libmodes_host.so via dump1090_amd.HostResolver): fed with the
oracle's stateless records it must print exactly what the reference prints."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as orc
from dump1090_amd import (HostResolver, _native as N, block_count, onlyaddr_text, raw_text, shard_blocks,
                          shard_byte_range, verbose_text)
from helpers import maxfix_of, oracle_records

CASES = ["modes1", "uniform", "coarse", "edges", "edges_smear", "frames", "smear", "lowsnr", "saturated"]


@pytest.mark.parametrize("case", CASES)
def test_resolve_reproduces_reference_listing(golden, streams, case):
    data = streams[case]
    for fs, flags in orc.FLAGSETS.items():
        recs, cands = oracle_records(data, maxfix_of(flags))
        r = HostResolver(**flags)
        msgs = r.resolve(recs, cands)
        assert raw_text(msgs) == golden[case]["raw"][fs]["text"], (case, fs)
        if fs in golden[case]["stats"]:
            assert r.stats_text() == golden[case]["stats"][fs]["text"], (case, fs)
        if fs == "default":
            assert onlyaddr_text(msgs) == golden[case]["onlyaddr"]["default"]["text"]
        if fs in golden[case]["verbose"]:
            # the full text dump pins every decoded field of struct modesMessage
            assert verbose_text(msgs) == golden[case]["verbose"][fs]["text"], (case, fs)
        r.close()


def test_resolve_without_candidates_gives_same_messages(streams):
    data = streams["smear"]
    flags = orc.FLAGSETS["aggressive_nocrc"]
    recs, cands = oracle_records(data, 2)
    a, b = HostResolver(**flags), HostResolver(**flags)
    assert raw_text(a.resolve(recs, cands)) == raw_text(b.resolve(recs, None))
    sa, sb = a.stats(), b.stats()
    assert sb["valid_preamble"] == -1
    sa.pop("valid_preamble"), sb.pop("valid_preamble")
    assert sa == sb


def test_resolve_in_batches_keeps_icao_state(streams, golden):
    """Feeding the records buffer by buffer (as the C host does per GPU batch) changes nothing."""
    data = streams["modes1"]
    recs, cands = oracle_records(data, 1)
    r = HostResolver()
    msgs = []
    for k in range(block_count(data.size)):
        sel = recs["block"] == k
        csel = (cands // N.BLOCK_STRIDE) == k
        msgs += r.resolve(recs[sel], cands[csel])
    assert raw_text(msgs) == golden["modes1"]["raw"]["default"]["text"]
    assert r.stats_text() == golden["modes1"]["stats"]["default"]["text"]


def test_message_fields_against_oracle(streams):
    data = streams["frames"]
    want, _ = orc.run_stream(data, **orc.FLAGSETS["default"])
    recs, cands = oracle_records(data, 1)
    got = HostResolver().resolve(recs, cands)
    assert len(got) == len(want) > 50
    for g, w in zip(got, want):
        assert (g.msg, g.msgbits, g.msgtype, g.crcok, g.crc, g.errorbit, g.aa1, g.aa2, g.aa3, g.phase_corrected,
                g.iid, g.block, g.j) == (bytes(w.msg), w.msgbits, w.msgtype, w.crcok, w.crc, w.errorbit, w.aa1, w.aa2,
                                         w.aa3, w.phase_corrected, w.iid, w.block, w.j)


def test_crc_helpers():
    L = N.host_lib()
    rng = np.random.default_rng(3)
    for bits in (56, 112):
        for _ in range(200):
            msg = rng.integers(0, 256, 14, dtype=np.uint8)
            assert L.modes_checksum(msg.ctypes.data, bits) == orc.lib().orc_checksum(msg.ctypes.data, bits)
            orc.lib().orc_compute_crc.restype = C.c_uint32
            orc.lib().orc_compute_crc.argtypes = [C.c_void_p, C.c_int]
            assert L.modes_compute_crc(msg.ctypes.data, bits) == orc.lib().orc_compute_crc(msg.ctypes.data, bits)
    assert [L.modes_message_len_by_type(t) for t in (0, 4, 11, 16, 17, 21, 22, 24)] == [56, 56, 56, 112, 112, 112, 56, 56]
    assert L.modes_block_count(0) == 1 and L.modes_block_count(262144) == 2 and L.modes_block_count(262145) == 2


def test_sharding_helpers():
    for nblocks in (1, 7, 8, 9, 4097):
        for world in (1, 2, 3, 8):
            parts = [shard_blocks(nblocks, world, r) for r in range(world)]
            assert parts[0][0] == 0 and sum(n for _, n in parts) == nblocks
            for (f0, n0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + n0 == f1
    assert shard_byte_range(0, 2, 10 ** 9) == (0, 2 * 262144)
    assert shard_byte_range(2, 2, 10 ** 9) == (2 * 262144 - 476, 4 * 262144)
    assert shard_byte_range(3, 1, 3 * 262144 + 10) == (3 * 262144 - 476, 3 * 262144 + 10)


def test_cli_fails_loudly_without_gpu():
    """No CPU fallback: on a machine without a HIP device the CLI must refuse, not emulate."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(N.PKG_DIR, "bin", "dump1090_amd")
    fixture = os.path.join(os.path.dirname(__file__), "golden", "modes1.bin")
    p = subprocess.run([exe, "--ifile", fixture, "--raw"], capture_output=True)
    assert p.returncode == 1 and p.stdout == b"" and b"no HIP device" in p.stderr


def test_raw_net_line_is_the_raw_line_in_upper_case(streams):
    """modesSendRawOutput (dump1090.c:2381-2393) differs from --raw only in the case of the hex digits."""
    data = streams["frames"]
    recs, cands = oracle_records(data, 1)
    lib = N.host_lib()
    cfg = N.HostConfig(1, 0, 1, 0)
    h = lib.modes_host_create(C.byref(cfg))
    out = (N.Emitted * (2 * recs.size + 16))()
    n = lib.modes_host_resolve_to_array(h, recs.ctypes.data, recs.size, cands.ctypes.data, cands.size, out, len(out))
    assert n > 50
    a, b = C.create_string_buffer(64), C.create_string_buffer(64)
    for i in range(n):
        lib.modes_format_raw(C.byref(out[i].mm), a)
        lib.modes_format_raw_net(C.byref(out[i].mm), b)
        assert b.value == a.value.upper() and b.value != a.value or a.value.upper() == a.value
    lib.modes_host_destroy(h)


def test_icao_whitelist_ttl_follows_the_callers_clock(streams, golden):
    """dump1090.c:913,924: whitelist entries die after 60 s of the caller's clock.  A file run never advances
    it (nothing expires: the goldens above).  Advanced by 100 s per record, every entry is expired by the
    time it is looked up - the reference run under a clock that advances 100 s per time() call prints
    223 lines, md5 9069aad8... (hard part 3: every AP-validated message is lost)."""
    import hashlib
    data = streams["modes1"]
    recs, _ = oracle_records(data, 1)
    r = HostResolver()
    got = []
    for i in range(recs.size):
        r.set_time(100 * i)
        got += r.resolve(recs[i:i + 1], None)
    text = raw_text(got)
    assert text.count("\n") == 223 and hashlib.md5(text.encode()).hexdigest().startswith("9069aad8")
    r.close()
    r = HostResolver()
    kept = []
    for i in range(recs.size):
        r.set_time(i // 100)                                                # well inside the TTL: nothing expires
        kept += r.resolve(recs[i:i + 1], None)
    assert raw_text(kept) == golden["modes1"]["raw"]["default"]["text"]
    r.close()


@pytest.mark.parametrize("case", CASES)
def test_multithreaded_resolve_equals_the_sequential_one(streams, golden, case):
    """modes_host_resolve_raw_mt: pieces resolved speculatively from guessed whitelists and confirmed in order must give
    the sequential listing, the same counters and the same whitelist afterwards - for any number of pieces, on the
    reference's own capture (AP-validated frames depend on addresses from earlier pieces) as on the synthetic ones."""
    data = streams[case]
    for fs in ("default", "aggressive", "nofix"):
        flags = orc.FLAGSETS[fs]
        recs, _ = oracle_records(data, maxfix_of(flags))
        seq = HostResolver(**flags)
        n0, text0 = seq.raw_listing(recs, None)
        assert text0.decode() == golden[case]["raw"][fs]["text"]
        follow0 = seq.raw_listing(recs, None)                      # the whitelist the first pass left behind
        for pieces in (2, 3, 7, 32):
            mt = HostResolver(**flags)
            n1, text1 = mt.raw_listing(recs, None, threads=-pieces)
            assert (n1, text1) == (n0, text0), (case, fs, pieces)
            s0, s1 = seq.stats(), mt.stats()
            assert mt.raw_listing(recs, None, threads=-pieces) == follow0, (case, fs, pieces, "state after the batch")
            mt.close()
        seq.close()


@pytest.mark.parametrize("case", ["modes1", "frames", "smear", "saturated"])
def test_segmented_resolve_equals_the_sequential_one(streams, golden, case):
    """modes_host_resolve_raw_mtv: the batch handed over as several arrays of whole buffers (how the gathered lists of N
    ranks and several calls lie in rank 0's buffers) - one parallel resolve over all of them gives the sequential listing
    and leaves the same whitelist, however the buffers are distributed over the arrays and the pieces over the threads."""
    data = streams[case]
    rng = np.random.default_rng(5)
    for fs in ("default", "aggressive"):
        flags = orc.FLAGSETS[fs]
        recs, _ = oracle_records(data, maxfix_of(flags))
        seq = HostResolver(**flags)
        want = seq.raw_listing(recs, None)
        assert want[1].decode() == golden[case]["raw"][fs]["text"]
        follow = seq.raw_listing(recs, None)
        seq.close()
        blocks = np.unique(recs["block"])
        for nseg, pieces in ((1, 3), (2, 2), (3, 5), (len(blocks), 2), (4, 64)):
            # cut points at buffer boundaries; empty segments are legal
            cuts = sorted(rng.choice(np.concatenate([blocks, blocks[-1:] + 1]), size=nseg - 1, replace=True)) if nseg > 1 else []
            bounds = [0] + [int(np.searchsorted(recs["block"], c)) for c in cuts] + [recs.size]
            segs = [recs[bounds[i]: bounds[i + 1]].copy() for i in range(len(bounds) - 1)]     # separate allocations
            mt = HostResolver(**flags)
            assert mt.raw_listing_segments(segs, threads=1) == want, (case, fs, nseg, "one thread")
            mt.close()
            mt = HostResolver(**flags)
            got = mt.raw_listing_segments(segs, threads=pieces) if recs.size >= 2048 * pieces else None
            mt.close()
            assert got is None or got == want
            mt = HostResolver(**flags)
            lens = (C.c_uint64 * len(segs))(*[a.size for a in segs])
            ptrs = (C.c_void_p * len(segs))(*[a.ctypes.data for a in segs])
            buf = C.create_string_buffer(62 * recs.size + 64)
            nb = C.c_uint64()
            n = N.host_lib().modes_host_resolve_raw_mtv(mt._h, ptrs, lens, len(segs), buf, len(buf), C.byref(nb), -pieces)   # forced pieces
            assert (int(n), buf.raw[: nb.value]) == want, (case, fs, nseg, pieces)
            assert mt.raw_listing(recs, None) == follow, "whitelist after the batch"
            mt.close()


def test_multithreaded_resolve_rejects_wrong_guesses():
    """A piece whose speculation was wrong is resolved again: an AP-validated frame (DF4) whose address is only known
    from a frame that sits INSIDE another frame's skip window of an earlier piece looks validated to the guess (the guess
    ignores skip windows) but is not; and the other way round for an address repaired (not clean) earlier."""
    from dump1090_amd import RECORD_DTYPE
    import synth as sy

    def rec(block, j, frame, syndrome=0, nfix=0, pos=0xFF):
        r = np.zeros(1, dtype=RECORD_DTYPE)
        r["block"], r["j"] = block, j
        for a in (0, 1):
            r["att"]["msg"][0, a, :len(frame)] = np.frombuffer(frame, dtype=np.uint8)
            r["att"]["gate_ok"][0, a] = 1
            r["att"]["syndrome"][0, a] = syndrome
            r["att"]["nfix"][0, a] = nfix
            r["att"]["fixpos"][0, a] = (pos, 0xFF)
        return r

    a17 = sy.make_frame(17, sy._payload(1, 14, 1))                 # aircraft A, clean
    b17 = sy.make_frame(17, sy._payload(1, 14, 2))                 # aircraft B, clean - but inside A's skip window
    addr_b = b17[1:4]
    # a DF4 whose AP field validates against B: data bytes 0..3, parity = crc(data) xor address
    data4 = bytes([4 << 3, 0x12, 0x34, 0x56])
    lib = N.host_lib()
    crc = lib.modes_compute_crc(data4 + bytes(3), 56)
    ap = bytes([((crc >> 16) & 0xFF) ^ addr_b[0], ((crc >> 8) & 0xFF) ^ addr_b[1], (crc & 0xFF) ^ addr_b[2]])
    df4 = data4 + ap
    syn4 = lib.modes_checksum(df4, 56)
    recs = np.concatenate([rec(0, 100, a17), rec(0, 150, b17),     # piece 0: B is skipped (j = 150 < 100 + 241)
                           rec(1, 100, df4, syndrome=syn4)])       # piece 1: DF4 "from B": nobody remembered B
    seq = HostResolver()
    want = seq.raw_listing(recs, None)
    assert want[0] == 1                                            # only A's frame
    mt = HostResolver()
    assert mt.raw_listing(recs, None, threads=-2) == want
    seq.close()
    mt.close()


def _rank_pieces(recs, world, rng, calls=2):
    """recs cut at buffer boundaries into `world` contiguous pieces (stream order = rank order; empty ones are legal), each
    handed over as up to `calls` arrays of whole buffers - how a rank's GPU calls leave its records."""
    blocks = np.unique(recs["block"])
    edges = np.concatenate([blocks, blocks[-1:] + 1]) if blocks.size else np.zeros(1, dtype=np.uint32)
    cuts = sorted(rng.choice(edges, size=world - 1, replace=True))
    bounds = [0] + [int(np.searchsorted(recs["block"], c)) for c in cuts] + [recs.size]
    out = []
    for r in range(world):
        piece = recs[bounds[r]: bounds[r + 1]]
        inner = sorted(rng.choice(edges, size=calls - 1, replace=True))
        ib = [0] + [int(np.searchsorted(piece["block"], c)) for c in inner] + [piece.size]
        out.append([piece[ib[i]: ib[i + 1]].copy() for i in range(len(ib) - 1)])
    return out


@pytest.mark.parametrize("case", CASES)
def test_resolve_on_the_ranks_equals_the_sequential_one(streams, golden, case):
    """distributed.rank_resolve_step (N ranks' generators in lockstep, LocalRanks): every rank resolves its own records from a
    guessed whitelist, the ranks confirm each other in stream order - the concatenated texts, the counters and the whitelist
    the step leaves are those of the sequential resolve of all the records, for any number of ranks, however the buffers
    fall on them, with every guess spoiled on purpose, and over two steps that share the whitelist."""
    from dump1090_amd.distributed import LocalRanks
    data = streams[case]
    rng = np.random.default_rng(11)
    for fs in ("default", "aggressive", "nofix"):
        flags = orc.FLAGSETS[fs]
        recs, _ = oracle_records(data, maxfix_of(flags))
        seq = HostResolver(**flags)
        n0, text0 = seq.raw_listing(recs, None)
        assert text0.decode() == golden[case]["raw"][fs]["text"]
        st0 = seq.stats()
        wl0 = seq.whitelist()
        n1, text1 = seq.raw_listing(recs, None)                    # a second batch from the whitelist the first left
        wl1 = seq.whitelist()
        seq.close()
        for world, threads, spoil in ((2, 1, ()), (3, -2, ()), (8, 1, ()), (3, 1, (1, 2)), (8, -3, (1, 2, 3, 4, 5, 6, 7))):
            segs = _rank_pieces(recs, world, rng)
            lr = LocalRanks(world, flags, threads)
            text, res = lr.step(segs, spoil=spoil)
            assert text == text0 and res[0]["lines"] == n0, (case, fs, world, spoil)
            assert {k: v for k, v in res[0]["stats"].items() if k != "valid_preamble"} == \
                   {k: v for k, v in st0.items() if k != "valid_preamble"}, (case, fs, world)
            assert all(r["stats"] == res[0]["stats"] and r["lines"] == n0 for r in res)
            assert np.array_equal(lr.state[0], wl0[0]) and np.array_equal(lr.state[1], wl0[1]), "the whitelist after the step"
            text, res = lr.step(_rank_pieces(recs, world, rng), spoil=spoil)
            assert text == text1 and res[0]["lines"] == n1, (case, fs, world, "second step")
            assert np.array_equal(lr.state[0], wl1[0]) and np.array_equal(lr.state[1], wl1[1])


def test_resolve_on_the_ranks_rejects_wrong_guesses():
    """The ranks' form of test_multithreaded_resolve_rejects_wrong_guesses: rank 0's clean frame B sits inside A's skip window,
    so rank 0 never remembers B; rank 1's DF4 validates against B under the guess and must come out unvalidated - rank 1 sees
    a wrong answer in its log, resolves again from the true state, and the step takes a second round."""
    from dump1090_amd import RECORD_DTYPE
    from dump1090_amd.distributed import LocalRanks
    import synth as sy

    def rec(block, j, frame, syndrome=0):
        r = np.zeros(1, dtype=RECORD_DTYPE)
        r["block"], r["j"] = block, j
        for a in (0, 1):
            r["att"]["msg"][0, a, :len(frame)] = np.frombuffer(frame, dtype=np.uint8)
            r["att"]["gate_ok"][0, a] = 1
            r["att"]["syndrome"][0, a] = syndrome
            r["att"]["fixpos"][0, a] = (0xFF, 0xFF)
        return r

    a17 = sy.make_frame(17, sy._payload(1, 14, 1))
    b17 = sy.make_frame(17, sy._payload(1, 14, 2))
    addr_b = b17[1:4]
    data4 = bytes([4 << 3, 0x12, 0x34, 0x56])
    lib = N.host_lib()
    crc = lib.modes_compute_crc(data4 + bytes(3), 56)
    df4 = data4 + bytes([((crc >> 16) & 0xFF) ^ addr_b[0], ((crc >> 8) & 0xFF) ^ addr_b[1], (crc & 0xFF) ^ addr_b[2]])
    syn4 = lib.modes_checksum(df4, 56)
    r0 = np.concatenate([rec(0, 100, a17), rec(0, 150, b17)])
    r1 = rec(1, 100, df4, syndrome=syn4)
    seq = HostResolver()
    want = seq.raw_listing(np.concatenate([r0, r1]), None)
    seq.close()
    assert want[0] == 1
    lr = LocalRanks(2, dict(fix=True, aggressive=False, check_crc=True))
    text, res = lr.step([[r0], [r1]])
    assert text == want[1] and res[0]["lines"] == 1
    assert res[1]["reruns"] == 1 and res[0]["reruns"] == 0 and res[0]["rounds"] == 2
    # the other way round: the guess is right, nothing runs twice
    text, res = lr.step([[np.concatenate([rec(2, 100, b17)])], [rec(3, 100, df4, syndrome=syn4)]])
    assert text.count(b"\n") == 2 and [r["reruns"] for r in res] == [0, 0] and res[0]["rounds"] == 1
    # a rank that starts from a wrong state although the guess was right (spoiled): same text, that rank runs twice
    lr = LocalRanks(3, dict(fix=True, aggressive=False, check_crc=True))
    text2, res = lr.step([[rec(2, 100, b17)], [rec(3, 100, df4, syndrome=syn4)], [rec(4, 7, df4, syndrome=syn4), rec(5, 9, a17)]], spoil=(1, 2))
    assert text2 == text + text[text.index(b"\n") + 1:] + want[1] and [r["reruns"] for r in res] == [0, 1, 1] and res[0]["rounds"] == 3


def test_a_listing_that_outgrows_its_buffer_is_cut_at_a_line(streams):
    """include/modes_host.h: when the listing is longer than `cap`, *nbytes still reports the whole length and out holds
    whole lines only, as many as fit - the same for the one-thread and the multi-threaded resolve (which used to copy
    later, smaller pieces behind a gap)."""
    flags = orc.FLAGSETS["default"]
    recs, _ = oracle_records(streams["frames"], maxfix_of(flags))
    seq = HostResolver(**flags)
    n0, text0 = seq.raw_listing(recs, None)
    seq.close()
    assert n0 > 20
    lib = N.host_lib()
    for cap in (len(text0) // 2, len(text0) // 3 + 7, 40, 1):
        for pieces in (0, 3, 7):
            h = HostResolver(**flags)
            buf = C.create_string_buffer(b"\xee" * (len(text0) + 64), len(text0) + 64)
            nb = C.c_uint64()
            if pieces:
                n = lib.modes_host_resolve_raw_mt(h._h, recs.ctypes.data, recs.size, buf, cap, C.byref(nb), -pieces)
            else:
                n = lib.modes_host_resolve_raw(h._h, recs.ctypes.data, recs.size, None, 0, buf, cap, C.byref(nb))
            h.close()
            assert (int(n), nb.value) == (n0, len(text0)), (cap, pieces)
            stored = buf.raw[:cap].split(b"\x00")[0]
            assert text0.startswith(stored) and (stored == b"" or stored.endswith(b"\n")), (cap, pieces, stored[-40:])
            assert len(stored) <= cap - 1 or cap <= 1
            assert buf.raw[cap:cap + 8] == b"\xee" * 8, "wrote past cap"
            if pieces:
                # as many whole lines as fit: the next line would not have
                nxt = text0[len(stored):].split(b"\n")[0] + b"\n"
                assert len(stored) + len(nxt) + 1 > cap, (cap, pieces)


def test_c_host_under_thread_sanitizer():
    """tools/sanitize_host.sh tsan-host: dump1090_amd/csrc/host_single.cpp - reader thread, resolver thread, lanes of two and three
    "devices" handed between them - built under -fsanitize=thread with the GPU library replaced by tests/native/gpu_stub.cpp
    (the oracle's stateless functions behind the same entry points).  No race, and the reference's md5s for --raw, --stats
    and --onlyaddr (the script checks both)."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS")}       # (this test may itself run under ASan)
    p = subprocess.run(["bash", os.path.join(root, "tools", "sanitize_host.sh"), "tsan-host"], capture_output=True, timeout=600, env=env)
    assert p.returncode == 0 and b"ThreadSanitizer" not in p.stderr + p.stdout, (p.stdout[-800:], p.stderr[-800:])
    assert p.stdout.count(b"md5 4a81758c8bec5e45ffa8541c5622938a") == 2


def test_c_host_one_process_per_gpu_with_two_and_three_processes():
    """tools/sanitize_host.sh ranks-host: `dump1090_amd --ranks N` with N = 1, 2, 3 real processes on this machine - the GPU
    library stubbed by the oracle (tests/native/gpu_stub.cpp), include/modes_gather.h implemented over shared memory
    (tests/native/gather_stub.cpp, the libmodes_gather.so the host dlopens): the fork, the id pipes, round-robin batches,
    gather rounds in three rotating slots, ranks that have no batch in a round, the EOF batch.  stdout is the reference's
    for every N and batch size (the script compares the md5s).  And the failure path: one rank's "GPU" does not come up
    (MODES_STUB_FAIL_DEVICE) - the job must end with status 1 instead of leaving the other ranks in the gather for ever."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS")}
    p = subprocess.run(["bash", os.path.join(root, "tools", "sanitize_host.sh"), "ranks-host"], capture_output=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-800:], p.stderr[-800:])
    # the reference's listing: 9 files x (N, batch size), N = 8, 6 pipes (--ifile -, round 5) and 3 restarts
    assert p.stdout.count(b"md5 4a81758c8bec5e45ffa8541c5622938a") == 19 + 12 and b"--onlyaddr --ranks 3: md5 bab0f055" in p.stdout
    # (three of them: the start-up probe of the communicator fails - on every rank, on a peer, on rank 0 - and rank 0 starts the
    # job over once with the other IPC mode; that second run prints the listing)
    assert p.stdout.count(b", 1 restart") == 3
    # a rank that fails to start while its peers wait in the gather ends the job (status 1), whichever rank it is
    assert p.stdout.count(b"fails to start: exit status 1") == 3
    # ... and so does a rank whose GPU call fails mid-stream, with its peer already inside that round's exchange (no teardown of
    # the communicator on that path: ADVICE round 3)
    assert p.stdout.count(b"fails in GPU call") == 2 and p.stdout.count(b": exit status 1") == 7      # (+ --resolve-on-ranks --sbs, refused)
    # --stats through the gather's second list (every rank's preamble positions on rank 0): the reference's nine lines for N = 1, 2, 3;
    # a list that outgrows its buffers fails the job
    assert p.stdout.count(b"md5 bc3d1c04b24f4989f0fc4a2d1f45abdd") == 9        # N = 1, 2, 3 x two batch sizes, N = 8, a pipe with N = 2, and (round 6) a pipe with N = 3 resolving on the ranks
    assert b"--raw --ranks 8 --batch-blocks 1: md5 4a81758c" in p.stdout              # eight processes, five of them without a batch
    assert b"--stats with 8 positions of room: exit status 1" in p.stdout
    # round 5: a pipe and --loop through --ranks (rank 0 reads, shared-memory slots): the pipe's listing for N = 1, 2, 3 x two batch sizes
    # is counted above; the replay's first 2.5 laps equal the one-process host's
    assert p.stdout.count(b"   --ifile - --ranks") == 6 and b"   --loop --ranks 2 / 3: the first" in p.stdout
    # --resolve-on-ranks (every rank resolves its own batches, the ranks confirm each other through shared memory, rank 0 prints the texts;
    # no gather library): N = 1, 2, 3, 8 x two batch sizes, a pipe, a replay; on a stream whose DF4 / DF5 / DF20 frames only validate against
    # an address an earlier rank remembered: right guesses are kept (0 re-runs), wrong starts are noticed and resolved again - same listing
    assert p.stdout.count(b"--resolve-on-ranks --ranks") == 8 + 2 + 4 + 4 and b"orderly teardown (--clean-exit): md5 4a81758c" in p.stdout and b"--resolve-on-ranks --ifile - --ranks 3: md5 4a81758c" in p.stdout
    assert p.stdout.count(b"on AP-validated frames: md5 524f28a5613c2468123104e0a17e61f8, 0 re-run(s)") == 2
    assert b"--ranks 2 on AP-validated frames, wrong starts: md5 524f28a5613c2468123104e0a17e61f8, 1 re-run(s)" in p.stdout
    assert b"--ranks 3 on AP-validated frames, wrong starts: md5 524f28a5613c2468123104e0a17e61f8, 2 re-run(s)" in p.stdout
    assert b"--resolve-on-ranks --loop --ranks 2 / 3: the first" in p.stdout and b"--resolve-on-ranks --sbs: exit status 1" in p.stdout
    # round 6: --stats (the ranks' counters added up by rank 0; a repeated resolve does not count twice), --onlyaddr and --raw-net resolved
    # on the ranks: N = 1, 2, 3, 8 with right and wrong starts, and a pipe
    assert p.stdout.count(b"--stats / --onlyaddr / --raw-net --ranks") == 8 and b"--ifile - --stats --ranks 3 --resolve-on-ranks: md5 bc3d1c04" in p.stdout
    assert p.stdout.count(b"--stats --resolve-on-ranks --ranks") == 4
    assert p.stdout.count(b"fails in its GPU call") == 2 and p.stdout.count(b": status 1\n") == 2       # a failing rank ends the job there too
    # ... and a rank KILLED between publishing its guess and its final tables (nobody sets `failed`): the watchdog / the parent-death signal
    # end the job within two seconds, whichever rank it is (the script checks status and time)
    assert p.stdout.count(b"between guess and final: status") == 3


def test_c_host_loop_replays_the_file_like_the_reference():
    """tools/sanitize_host.sh loop-host: `dump1090_amd --loop` (dump1090.c:488-494) on the stubbed host - the first 2.5 laps of
    output are the unmodified reference's bytes (it replays the file into the same buffer, the whitelist carries over),
    whatever the batch size; and --clean-exit (the orderly teardown the default exit skips) prints the plain listing."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS")}
    p = subprocess.run(["bash", os.path.join(root, "tools", "sanitize_host.sh"), "loop-host"], capture_output=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-800:], p.stderr[-800:])
    assert b"first lap = the plain listing" in p.stdout and b"--clean-exit: md5 4a81758c8bec5e45ffa8541c5622938a" in p.stdout
    # --clean-exit where it matters: 1 GiB of mapping handed back batch by batch, six lanes allocated into the holes, under ASan
    assert b"1 GiB sparse file + capture, 64 batches over six lanes, under ASan: 284 lines" in p.stdout
    if os.path.exists(os.path.join(root, "oracle", "_ref", "dump1090_ref")):
        assert b"--loop == oracle/_ref/dump1090_ref --loop" in p.stdout


def test_c_host_serves_a_pipe_at_the_pace_it_delivers():
    """tools/sanitize_host.sh pipe-host (dump1090.c:460-512, :2969-2990: the reference prints a buffer's messages
    within that buffer): the capture written one 256 KiB buffer every 150 ms into `dump1090_amd --ifile -` with the DEFAULT batch of
    512 buffers - the first line is out long before the writer has finished (one process, --ranks 2 in both resolve modes), the listing
    and --stats are the file run's, and an unpaced pipe prints the same bytes for any --flush-ms."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS")}
    p = subprocess.run(["bash", os.path.join(root, "tools", "sanitize_host.sh"), "pipe-host"], capture_output=True, timeout=600, env=env)
    assert p.returncode == 0 and b"FAIL" not in p.stdout, (p.stdout[-1200:], p.stderr[-800:])
    assert p.stdout.count(b"paced pipe, ") == 4 and b"unpaced pipe (cat |)" in p.stdout
