"""-m gpu: the HIP path (through the C ABI) against the oracle, bit for bit."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import oracle as orc
import synth
from helpers import assert_records_equal, maxfix_of, oracle_records
from dump1090_amd import _native as N

pytestmark = pytest.mark.gpu

CASES = ["modes1", "uniform", "coarse", "edges", "edges_smear", "frames", "smear", "lowsnr", "noise", "saturated"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


def to_dev(torch, data):
    return torch.from_numpy(np.ascontiguousarray(data)).to("cuda:0")


def test_native_library_is_loaded(torch_cuda):
    """The product libraries - not a fallback - are what this process runs: both are mapped, a context exists
    on the GPU, and the ABI version is the header's."""
    from dump1090_amd import Demodulator, HostResolver, _native as N
    d, r = Demodulator(), HostResolver()
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(N.GPU_LIB) in maps, "libmodes_gfx950.so is not mapped"
    assert os.path.realpath(N.HOST_LIB) in maps, "libmodes_host.so is not mapped"
    assert N.gpu_lib().modes_gpu_abi_version() == 5
    r.close()
    d.close()


def test_one_detect_per_context_at_a_time(torch_cuda, streams):
    """A second detect before the fetch would relaunch over live scratch: MODES_ERR_STATE, and the first one survives."""
    from dump1090_amd import Demodulator, ModesError
    iq = to_dev(torch_cuda, streams["frames"])
    d = Demodulator()
    d.detect(iq)
    with pytest.raises(ModesError, match="MODES_ERR_STATE"):
        d.detect(iq)
    recs, _, _ = d.fetch()
    want, _ = oracle_records(streams["frames"], 1)
    assert_records_equal(recs, want, "after the refused second detect")
    with pytest.raises(ModesError, match="MODES_ERR_STATE"):
        d.fetch()
    d.close()


def test_device_output_and_direct_host_path_agree(torch_cuda, streams):
    """The ordered list in a caller-owned device buffer (modes_gpu_set_output + fetch_device: what the RCCL gather
    consumes), the zero-copy host path (short lists) and the copy path (long lists) are the same records."""
    from dump1090_amd import Demodulator, ModesError, RECORD_DTYPE
    torch = torch_cuda
    data = streams["modes1"]
    iq = to_dev(torch, data)
    want, _ = oracle_records(data, 1)
    assert want.size > 500
    for direct in (0, 1, 1 << 20):                      # default (4096), everything through the copy, everything direct
        d = Demodulator(direct_records=direct)
        d.detect(iq)
        recs, _, info = d.fetch()
        assert_records_equal(recs, want, ("direct_records", direct))
        d.close()
    out = torch.zeros(8192 * 64, dtype=torch.uint8, device="cuda:0")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    # short list: in order with the kernels; "long" list (direct_records = 1): put in order inside fetch_device, or
    # by a kernel that follows the detect in its stream (MODES_GPU_ORDER_IN_STREAM: complete in stream order)
    for direct, eager in ((0, False), (1, False), (1, True)):
        d = Demodulator(direct_records=direct, order_in_stream=eager)
        out.zero_()
        d.set_output(out, cnt)
        d.detect(iq)
        if eager:
            torch.cuda.synchronize()                     # no fetch yet: the list is already complete
            got = out[: want.size * 64].cpu().numpy().view(RECORD_DTYPE)
            assert_records_equal(got, want, "device list in stream order")
        n, info = d.fetch_device()
        assert n == want.size and int(cnt.item()) == n
        got = out[: n * 64].cpu().numpy().view(RECORD_DTYPE)
        assert_records_equal(got, want, ("device list", direct, eager))
        d.close()
    d = Demodulator()
    small = torch.zeros(64 * 64, dtype=torch.uint8, device="cuda:0")      # 64 records of room for 559
    d.set_output(small, cnt)
    d.detect(iq)
    with pytest.raises(ModesError, match="MODES_ERR_OVERFLOW"):
        d.fetch_device()
    assert int(cnt.item()) == want.size                  # the device word still says what the call needed
    d.close()


def test_magnitude_all_byte_pairs(torch_cuda):
    """K1: every (I,Q) byte pair (SURVEY.md 4.1 ii) plus ragged lengths."""
    from dump1090_amd import Demodulator
    d = Demodulator()
    iq = np.stack(np.meshgrid(np.arange(256), np.arange(256), indexing="ij"), -1).astype(np.uint8).reshape(-1)
    got = d.compute_magnitude_vector(to_dev(torch_cuda, iq)).cpu().numpy()
    assert np.array_equal(got, orc.magnitude(iq))
    for n in (2, 14, 16, 18, 4094, 262620):
        part = np.resize(iq[::7], n)
        got = d.compute_magnitude_vector(to_dev(torch_cuda, part)).cpu().numpy()
        assert np.array_equal(got, orc.magnitude(part)), n
    s = d.compute_power(to_dev(torch_cuda, iq)).cpu().numpy()
    i = iq[0::2].astype(np.int64) - 127
    q = iq[1::2].astype(np.int64) - 127
    assert np.array_equal(s, np.minimum(i * i + q * q, 32767).astype(np.uint16))      # the scan's saturated powers
    d.close()


def test_table_free_magnitude_equals_the_table_on_the_device(torch_cuda):
    """modes_mag_exact with the DEVICE's single-precision square root (the demod kernel's magnitude for powers beyond
    its 512-entry LDS table) equals the reference's LUT for every saturated power."""
    from dump1090_amd import Demodulator
    d = Demodulator()
    lut, exact = d.debug_tables()
    s = np.arange(32768, dtype=np.float64)
    s[32767] = 32768.0
    real = np.zeros(32768, dtype=bool)                       # the powers that occur: sums of two squares (the LUT's other entries are 0)
    i, q = np.meshgrid(np.arange(129), np.arange(129))
    real[np.minimum(i * i + q * q, 32767).reshape(-1)] = True
    assert np.array_equal(exact, np.round(np.sqrt(s) * 360.0).astype(np.uint16))
    assert np.array_equal(exact[real], lut[real]) and real.sum() == 5924
    d.close()


def test_synth_noise_matches_host_generator(torch_cuda):
    from dump1090_amd import Demodulator
    d = Demodulator()
    for first, n, seed, sig in ((0, 1 << 16, 18, 941), (12345, 4099, 7, 627), (1 << 33, 1 << 12, 99, 941)):
        out = torch_cuda.empty(n, dtype=torch_cuda.uint8, device="cuda:0")
        d.synth_noise(out, first, seed, sig)
        assert np.array_equal(out.cpu().numpy(), synth.noise_bytes(seed, first, n, sig)), (first, n)
    d.close()


@pytest.mark.parametrize("demod_variant", [0, 2, 3])
@pytest.mark.parametrize("case", CASES)
def test_records_and_candidates_match_oracle(torch_cuda, streams, case, demod_variant):
    """Both demodulation paths - one kernel (3), select + record (2) - and the automatic choice between them (0): two
    independent implementations of stages 1-3 against the oracle."""
    from dump1090_amd import Demodulator
    data = streams[case]
    iq = to_dev(torch_cuda, data)
    for flags in (orc.FLAGSETS["default"], orc.FLAGSETS["aggressive"], orc.FLAGSETS["nofix"]):
        mf = maxfix_of(flags)
        d = Demodulator(keep_candidates=True, demod_variant=demod_variant, **flags)
        d.detect(iq)
        recs, cands, info = d.fetch()
        want, want_cands = oracle_records(data, mf)
        assert np.array_equal(cands, want_cands), (case, "preamble positions")
        assert info["n_preambles"] == want_cands.size and info["n_forwarded"] >= want_cands.size
        assert_records_equal(recs, want, ctx=(case, mf))
        if recs.size:       # the class bytes are for THIS context's repair policy (the host believes no others)
            assert ((recs["att"]["cls"] & 0xE0) == (0x80 | (0x20 if flags["fix"] else 0) | (0x40 if flags["aggressive"] else 0))).all()
        d.close()


def test_removed_demod_variant_is_refused(torch_cuda):
    from dump1090_amd import Demodulator, ModesError
    with pytest.raises(ModesError, match="demod_variant 1"):      # (an explicit variant is never overridden by the environment)
        Demodulator(demod_variant=1)


def test_environment_knob_only_fills_in_the_automatic_choice(torch_cuda, streams, monkeypatch):
    """MODES_GPU_DEMOD_VARIANT (tools/gpu_round.sh: a whole suite on one path) applies to contexts created with demod_variant 0
    only, and must be 0, 2 or 3: garbage or the removed variant 1 fails the create instead of silently meaning `automatic`."""
    from dump1090_amd import Demodulator, ModesError
    iq = to_dev(torch_cuda, streams["frames"])
    want, _ = oracle_records(streams["frames"], 1)
    for value in ("banana", "1", "2x", ""):
        monkeypatch.setenv("MODES_GPU_DEMOD_VARIANT", value)
        with pytest.raises(ModesError, match="MODES_GPU_DEMOD_VARIANT"):
            Demodulator()
    monkeypatch.setenv("MODES_GPU_DEMOD_VARIANT", "2")
    for explicit, third_kernel in ((3, False), (0, True), (2, True)):          # 3 stays the one-kernel path: no record kernel time
        d = Demodulator(demod_variant=explicit)
        d.detect(iq)
        recs, _, info = d.fetch()
        assert_records_equal(recs, want, ctx=("env", explicit))
        assert (info["order_ms"] > 0) == third_kernel, (explicit, info["order_ms"])
        d.close()


def test_automatic_demod_path_follows_the_record_density(torch_cuda, streams):
    """demod_variant 0: a context's first call and calls behind record-free ones run the one-kernel path; a call that follows
    a record-rich call (> 4096 records per GiB) runs select + record.  Same records either way; which path ran shows in the
    third kernel's time (the record kernel is timed on the two-kernel path; a short list needs no order kernel on the other)."""
    from dump1090_amd import Demodulator
    if os.environ.get("MODES_GPU_DEMOD_VARIANT"):
        pytest.skip("the environment overrides the variant")
    rich, quiet = to_dev(torch_cuda, streams["frames"]), to_dev(torch_cuda, streams["noise"])
    want, _ = oracle_records(streams["frames"], 1)
    d = Demodulator()
    seen = []
    for iq, expect in ((rich, want), (rich, want), (quiet, None), (rich, want), (rich, want)):
        d.detect(iq)
        recs, _, info = d.fetch()
        seen.append(info["order_ms"] > 0)
        if expect is not None:
            assert_records_equal(recs, expect, ctx=("auto", len(seen)))
        else:
            assert recs.size == 0
    assert seen == [False, True, True, False, True], seen       # the call AFTER a rich one is the first on the other path
    d.close()


@pytest.mark.parametrize("case", CASES)
def test_listing_matches_reference(torch_cuda, golden, streams, case):
    from dump1090_amd import Demodulator, onlyaddr_text, raw_text
    iq = to_dev(torch_cuda, streams[case])
    for fs, flags in orc.FLAGSETS.items():
        d = Demodulator(keep_candidates=True, **flags)
        msgs = d.demodulate(iq)
        assert raw_text(msgs) == golden[case]["raw"][fs]["text"], (case, fs)
        if fs in golden[case]["stats"]:
            assert d.last["stats_text"] == golden[case]["stats"][fs]["text"], (case, fs)
        if fs == "default":
            assert onlyaddr_text(msgs) == golden[case]["onlyaddr"]["default"]["text"]
        d.close()


def test_modes1_published_hash_from_host_buffer(torch_cuda, streams):
    """The reference's own fixture through modes_gpu_demod_host (the C host's entry point)."""
    from dump1090_amd import Demodulator, raw_text
    d = Demodulator()
    raw = np.fromfile(os.path.join(ROOT, "tests", "golden", "modes1.bin"), dtype=np.uint8)   # NOT padded
    text = raw_text(d.demodulate(raw))
    assert (text.count("\n"), hashlib.md5(text.encode()).hexdigest()) == (284, "4a81758c8bec5e45ffa8541c5622938a")
    d.close()


@pytest.mark.parametrize("flags,lines,md5", [
    (["--raw"], 284, "4a81758c8bec5e45ffa8541c5622938a"),
    (["--raw", "--no-fix"], 283, "ac539444a66eb99a7f04affa95c55079"),
    (["--raw", "--aggressive", "--no-crc-check"], 824, "bec25488d6b84e9b0703d164de1cc873"),
    (["--onlyaddr"], 284, "bab0f055e262e216208a5cbbdf63fe24"),
    (["--stats"], 9, "bc3d1c04b24f4989f0fc4a2d1f45abdd"),
    ([], 3202, "0bf2290fa954f1675437e52508ea8aa3"),                       # the verbose dump of every field
    (["--aggressive", "--no-crc-check"], 5357, "b4e11b2e0017cdf772e49d300ab19158"),
    (["--raw", "--batch-blocks", "1"], 284, "4a81758c8bec5e45ffa8541c5622938a"),
    # the N-GPU split of the C host, emulated on one device: batch b runs on "GPU" b mod N, each with its own
    # contexts; the listing and the counters must not notice
    (["--raw", "--batch-blocks", "1", "--gpu-list", "0,0"], 284, "4a81758c8bec5e45ffa8541c5622938a"),
    (["--raw", "--batch-blocks", "1", "--gpu-list", "0,0,0", "--depth", "1"], 284, "4a81758c8bec5e45ffa8541c5622938a"),
    (["--stats", "--batch-blocks", "1", "--gpu-list", "0,0"], 9, "bc3d1c04b24f4989f0fc4a2d1f45abdd"),
    (["--batch-blocks", "2", "--gpu-list", "0,0", "--aggressive", "--no-crc-check"], 5357, "b4e11b2e0017cdf772e49d300ab19158"),
    (["--raw", "--gpus", "1"], 284, "4a81758c8bec5e45ffa8541c5622938a"),
])
def test_cli_reproduces_reference_stdout(torch_cuda, flags, lines, md5):
    """dump1090_amd --ifile testfiles/modes1.bin: BASELINE.md section 4 hashes of the reference."""
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    p = subprocess.run([exe, "--ifile", os.path.join(ROOT, "tests", "golden", "modes1.bin")] + flags,
                       capture_output=True, check=True)
    assert (p.stdout.count(b"\n"), hashlib.md5(p.stdout).hexdigest()) == (lines, md5), p.stderr[-400:]
    # stdin works too
    data = open(os.path.join(ROOT, "tests", "golden", "modes1.bin"), "rb").read()
    p2 = subprocess.run([exe, "--ifile", "-"] + flags, input=data, capture_output=True, check=True)
    assert p2.stdout == p.stdout


def test_cli_loop_and_clean_exit(torch_cuda, tmp_path):
    """dump1090_amd --loop on the GPU (dump1090.c:488-494): the first 2.5 laps of output over the padded capture are the
    compiled reference's bytes (oracle/_ref travels with the snapshot), for two batch sizes; --clean-exit (the orderly teardown
    the default exit skips) prints the plain listing and ends with status 0."""
    import synth
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    pad = tmp_path / "pad.bin"
    synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin")).tofile(pad)
    one = subprocess.run([exe, "--ifile", str(pad), "--raw", "--clean-exit"], capture_output=True, check=True).stdout
    assert hashlib.md5(one).hexdigest() == "4a81758c8bec5e45ffa8541c5622938a"
    n = len(one) * 5 // 2

    def laps(cmd, env=None):
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
        got = b""
        try:
            while len(got) < n:
                chunk = p.stdout.read(n - len(got))
                if not chunk:
                    break
                got += chunk
        finally:
            p.kill()
            p.wait()
        return got

    a = laps([exe, "--ifile", str(pad), "--raw", "--loop"])
    b = laps([exe, "--ifile", str(pad), "--raw", "--loop", "--batch-blocks", "1"])
    assert len(a) == n and a == b and a[:len(one)] == one
    if orc.have_ref():
        ref = laps([orc.REF_BIN, "--ifile", str(pad), "--raw", "--loop"], env=dict(os.environ, LD_PRELOAD=orc.FIXED_TIME))
        assert ref == a


def test_cli_two_ranks_on_two_gpus(torch_cuda, streams, tmp_path):
    """dump1090_amd --ranks 2 where the box has two GPUs: the lists of two PROCESSES travel over RCCL between two devices -
    stdout must be the single-process listing (on a one-GPU box the same command must fail fast: next test)."""
    if torch_cuda.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    for case, batch in (("modes1", 1), ("frames", 2)):
        path = tmp_path / (case + ".bin")
        streams[case].tofile(path)
        one = subprocess.run([exe, "--ifile", str(path), "--raw", "--batch-blocks", str(batch)], capture_output=True, check=True)
        two = subprocess.run([exe, "--ifile", str(path), "--raw", "--batch-blocks", str(batch), "--ranks", "2"], capture_output=True, timeout=300)
        assert two.returncode == 0 and two.stdout == one.stdout and len(one.stdout) > 0, two.stderr[-600:]


@pytest.mark.parametrize("flags,name", [(["--raw"], "default"), (["--raw", "--aggressive"], "aggressive"), (["--raw", "--no-fix"], "nofix"),
                                        (["--onlyaddr"], None), ([], None)])
def test_cli_one_process_per_gpu_gathers_over_rccl(torch_cuda, golden, streams, tmp_path, flags, name):
    """dump1090_amd --ranks 1: the one-process-per-GPU host - unique id, communicator, device output buffers, length
    all-gather, the list through ncclSend / ncclRecv (to itself: the group has one rank), rank 0's resolve - on the GPU at
    hand.  Same stdout as the single-process host and, for --raw, as the reference; several rounds (small batches, three
    slots in rotation), the EOF batch included."""
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    for case, batch in (("modes1", 1), ("frames", 2), ("edges_smear", 4)):
        path = tmp_path / (case + ".bin")
        streams[case].tofile(path)
        one = subprocess.run([exe, "--ifile", str(path), "--batch-blocks", str(batch)] + flags, capture_output=True, check=True)
        p = subprocess.run([exe, "--ifile", str(path), "--batch-blocks", str(batch), "--ranks", "1", "--timing"] + flags, capture_output=True)
        assert p.returncode == 0, p.stderr[-600:]
        assert p.stdout == one.stdout and len(p.stdout) > 0, (case, flags)
        if name is not None:
            assert p.stdout.decode() == golden[case]["raw"][name]["text"], (case, name)
        t = json.loads([ln for ln in p.stderr.decode().splitlines() if ln.startswith("{")][-1])
        assert t["ranks"] == 1 and t["rounds"] == streams[case].size // (batch * 262144) + 1
        assert t["rccl"]["p2p_ops"] >= 2 and t["rccl"]["bytes_received"] >= 64 and t["rccl"]["version"] > 20000


def test_cli_one_process_per_gpu_prints_the_reference_stats(torch_cuda, streams, tmp_path):
    """dump1090_amd --ranks 1 --stats: the preamble positions of every batch go through the gather's second list (H2D from the
    context's sorted host list, the lengths in the same all-gather, D2H on rank 0) - the nine lines of the reference
    (md5 bc3d1c04...) for several batch sizes, and the same text as the single-process host on a generated stream."""
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    raw = os.path.join(ROOT, "tests", "golden", "modes1.bin")
    for batch in ("1", "2", "512"):
        p = subprocess.run([exe, "--ifile", raw, "--stats", "--ranks", "1", "--batch-blocks", batch], capture_output=True)
        assert p.returncode == 0, p.stderr[-600:]
        assert (p.stdout.count(b"\n"), hashlib.md5(p.stdout).hexdigest()) == (9, "bc3d1c04b24f4989f0fc4a2d1f45abdd"), (batch, p.stdout)
    path = tmp_path / "frames.bin"
    streams["frames"].tofile(path)
    one = subprocess.run([exe, "--ifile", str(path), "--stats", "--batch-blocks", "2"], capture_output=True, check=True).stdout
    two = subprocess.run([exe, "--ifile", str(path), "--stats", "--batch-blocks", "2", "--ranks", "1"], capture_output=True, check=True).stdout
    assert one == two and one.count(b"\n") == 9
    small = subprocess.run([exe, "--ifile", raw, "--stats", "--ranks", "1", "--gather-candidates", "8"], capture_output=True)
    assert small.returncode == 1 and b"preamble positions" in small.stderr and small.stdout == b""


def test_cli_one_process_per_gpu_reads_a_pipe_and_replays_a_file(torch_cuda, streams, tmp_path):
    """dump1090_amd --ranks 1 with --ifile - and with --loop (round 5: rank 0 reads, the ranks take their batches from shared memory):
    the pipe's listing is the file's, the replay's first 2.5 laps are the one-process host's (which are the reference's:
    test_cli_loop_and_clean_exit) - through real RCCL on the GPU at hand."""
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    pad = tmp_path / "pad.bin"
    synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin")).tofile(pad)
    one = subprocess.run([exe, "--ifile", str(pad), "--raw"], capture_output=True, check=True).stdout
    assert hashlib.md5(one).hexdigest() == "4a81758c8bec5e45ffa8541c5622938a"
    with open(pad, "rb") as f:
        p = subprocess.run([exe, "--ifile", "-", "--raw", "--ranks", "1", "--batch-blocks", "1"], stdin=f, capture_output=True)
    assert p.returncode == 0 and p.stdout == one, p.stderr[-600:]
    n = len(one) * 5 // 2
    outs = []
    for extra in ([], ["--ranks", "1"]):
        q = subprocess.Popen([exe, "--ifile", str(pad), "--raw", "--loop", "--batch-blocks", "1"] + extra, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        got = b""
        while len(got) < n:
            chunk = q.stdout.read(n - len(got))
            if not chunk:
                break
            got += chunk
        q.kill()
        q.wait()
        outs.append(got)
    assert outs[0] == outs[1] and len(outs[0]) == n and outs[0][:len(one)] == one


@pytest.mark.parametrize("nranks", [1, 2, 3])
def test_cli_ranks_resolve_their_own_batches(torch_cuda, golden, streams, tmp_path, nranks):
    """dump1090_amd --ranks N --resolve-on-ranks: N real PROCESSES, each with its own contexts on the GPU at hand (--gpu-list 0,0,...:
    no communicator exists in this mode, so the ranks may share a device), every rank resolving its own batches from a guessed
    whitelist, the confirmation in rank order and the texts through shared memory - stdout is the reference's listing, for a file,
    a pipe and a replay; the low-SNR stream under --aggressive (retries, two-bit repairs) included."""
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    ranks = ["--ranks", str(nranks), "--gpu-list", ",".join(["0"] * nranks), "--resolve-on-ranks"]
    for case, batch, flags, name in (("modes1", 1, [], "default"), ("frames", 2, [], "default"), ("edges_smear", 1, ["--aggressive"], "aggressive"),
                                     ("lowsnr", 1, ["--aggressive"], "aggressive"), ("frames", 1, ["--no-fix"], "nofix")):
        path = tmp_path / (case + ".bin")
        streams[case].tofile(path)
        p = subprocess.run([exe, "--ifile", str(path), "--raw", "--batch-blocks", str(batch), "--timing"] + flags + ranks, capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr[-600:]
        assert p.stdout.decode() == golden[case]["raw"][name]["text"] and len(p.stdout) > 0, (case, flags, nranks)
        t = json.loads([ln for ln in p.stderr.decode().splitlines() if ln.startswith("{")][-1])
        assert t["ranks"] == nranks and t["resolve_on"] == "ranks" and t["sink_calls"] == p.stdout.count(b"\n")
    pad = tmp_path / "pad.bin"
    synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin")).tofile(pad)
    one = subprocess.run([exe, "--ifile", str(pad), "--raw"], capture_output=True, check=True).stdout
    with open(pad, "rb") as f:
        p = subprocess.run([exe, "--ifile", "-", "--raw", "--batch-blocks", "1"] + ranks, stdin=f, capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout == one and hashlib.md5(one).hexdigest() == "4a81758c8bec5e45ffa8541c5622938a", p.stderr[-600:]
    n = len(one) * 5 // 2
    outs = []
    for extra in ([], ranks):
        q = subprocess.Popen([exe, "--ifile", str(pad), "--raw", "--loop", "--batch-blocks", "1"] + extra, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        got = b""
        while len(got) < n:
            chunk = q.stdout.read(n - len(got))
            if not chunk:
                break
            got += chunk
        q.kill()
        q.wait()
        outs.append(got)
    assert outs[0] == outs[1] and len(outs[0]) == n


@pytest.mark.parametrize("nranks", [1, 3])
def test_cli_ranks_resolve_stats_and_onlyaddr(torch_cuda, golden, streams, tmp_path, nranks):
    """--resolve-on-ranks for the sinks beyond --raw (round 6; dump1090.c:2993-3006, :1317-1331): --stats - every rank counts its own
    batches, preamble positions included, rank 0 adds the nine counters up - and --onlyaddr / --raw-net, as real processes sharing the
    device; the reference's md5s on its capture, the one-process host's output on the frames stream."""
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    ranks = ["--ranks", str(nranks), "--gpu-list", ",".join(["0"] * nranks), "--resolve-on-ranks", "--batch-blocks", "1"]
    cap = os.path.join(ROOT, "tests", "golden", "modes1.bin")
    for flags, lines, md5 in ((["--stats"], 9, "bc3d1c04b24f4989f0fc4a2d1f45abdd"), (["--onlyaddr"], 284, "bab0f055e262e216208a5cbbdf63fe24")):
        p = subprocess.run([exe, "--ifile", cap] + flags + ranks, capture_output=True, timeout=300)
        assert p.returncode == 0 and (p.stdout.count(b"\n"), hashlib.md5(p.stdout).hexdigest()) == (lines, md5), (flags, p.stderr[-400:])
    path = tmp_path / "frames.bin"
    streams["frames"].tofile(path)
    for flags in (["--stats"], ["--stats", "--aggressive"], ["--raw-net"], ["--onlyaddr", "--no-fix"]):
        want = subprocess.run([exe, "--ifile", str(path)] + flags, capture_output=True, check=True).stdout
        got = subprocess.run([exe, "--ifile", str(path)] + flags + ranks, capture_output=True, timeout=300)
        assert got.returncode == 0 and got.stdout == want and len(want) > 0, (flags, got.stderr[-400:])
    p = subprocess.run([exe, "--ifile", cap, "--sbs"] + ranks, capture_output=True, timeout=60)
    assert p.returncode == 1 and b"aircraft table" in p.stderr


def test_cli_more_ranks_than_gpus_fails_instead_of_hanging(torch_cuda, streams, tmp_path):
    """dump1090_amd --ranks 2 on a box with one GPU: rank 1 has no device, rank 0 already waits in the communicator's
    rendezvous.  Rank 0's watchdog ends the job: status 1 and a message, not a hang."""
    if torch_cuda.cuda.device_count() >= 2:
        pytest.skip("needs a box with a single GPU")
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    path = tmp_path / "modes1.bin"
    streams["modes1"].tofile(path)
    p = subprocess.run([exe, "--ifile", str(path), "--raw", "--ranks", "2"], capture_output=True, timeout=120)
    assert p.returncode == 1 and p.stdout == b"", (p.returncode, p.stderr[-600:])
    assert b"rank 1" in p.stderr, p.stderr[-600:]


def test_gather_library_overflow_and_state_errors(torch_cuda, streams):
    """include/modes_gather.h through ctypes, one rank: a list longer than the gather buffers is MODES_ERR_OVERFLOW from
    modes_gather_records (where every rank would see it), calls out of order are MODES_ERR_STATE, and after an exchange
    rank 0 holds the records the detect produced."""
    import ctypes as C
    from dump1090_amd import Demodulator
    L = N.gather_lib()
    iq = to_dev(torch_cuda, streams["frames"])
    want = None
    for cap in (1 << 14, 8):
        ident = (C.c_ubyte * 128)()
        assert L.modes_gather_unique_id(ident) == 0                        # one id per communicator
        cfg = N.GatherConfig(0, 0, 1, cap, 2, 0)
        h = C.c_void_p()
        assert L.modes_gather_create(C.byref(cfg), ident, C.byref(h)) == 0, L.modes_gather_last_error(None)
        d_rec, d_cnt, capacity = C.c_void_p(), C.c_void_p(), C.c_uint64()
        assert L.modes_gather_output(h, 1, C.byref(d_rec), C.byref(capacity), C.byref(d_cnt)) == 0 and capacity.value == cap
        assert L.modes_gather_output(h, 2, C.byref(d_rec), C.byref(capacity), C.byref(d_cnt)) == -1            # no such slot
        assert L.modes_gather_records(h, 1) == -5 and L.modes_gather_wait(h, 1, None, None, None) == -5           # nothing queued
        d = Demodulator()
        assert N.gpu_lib().modes_gpu_set_output(d._h, d_rec, cap, d_cnt) == 0
        d.detect(iq)
        if cap > 8:
            n, _ = d.fetch_device()
            assert L.modes_gather_counts(h, 1) == 0 and L.modes_gather_counts(h, 1) == -5                        # one exchange per slot at a time
            assert L.modes_gather_records(h, 1) == 0
            recs, nrec, counts = C.c_void_p(), C.c_uint64(), C.POINTER(C.c_uint64)()
            assert L.modes_gather_wait(h, 1, C.byref(recs), C.byref(nrec), C.byref(counts)) == 0, L.modes_gather_last_error(h)
            assert nrec.value == n == counts[0] and n > 100
            got = np.frombuffer((C.c_uint8 * (n * 64)).from_address(recs.value), dtype=N.RECORD_DTYPE).copy()
            d2 = Demodulator()
            d2.detect(iq)
            want = d2.fetch()[0]
            d2.close()
            assert np.array_equal(got, want)
            st = N.GatherStats()
            assert L.modes_gather_get_stats(h, C.byref(st)) == 0 and st.p2p_ops == 2 and st.bytes_received == n * 64 and st.gather_ms > 0
        else:
            with pytest.raises(N.ModesError, match="OVERFLOW"):
                d.fetch_device()
            assert L.modes_gather_counts(h, 1) == 0                       # the true length still travels ...
            assert L.modes_gather_records(h, 1) == -4                     # ... and fails the exchange on every rank
            assert b"gather buffers hold 8" in L.modes_gather_last_error(h)
        d.close()
        L.modes_gather_destroy(h)


@pytest.mark.parametrize("flag,golden_file", [("--sbs", "modes1_sbs.txt"), ("--raw-net", "modes1_rawnet.txt")])
def test_cli_network_sink_lines_match_reference_capture(torch_cuda, flag, golden_file):
    """--sbs / --raw-net print what the reference wrote to clients of its ports 30003 / 30002 for the same
    capture (tests/golden/make_net_golden.py).  The constant-clock interposer of the oracle pins which CPR
    frame counts as newer, as it did for the reference when the golden was recorded."""
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME) if os.path.exists(orc.FIXED_TIME) else dict(os.environ)
    p = subprocess.run([exe, "--ifile", os.path.join(ROOT, "tests", "golden", "modes1.bin"), flag],
                       capture_output=True, check=True, env=env)
    want = open(os.path.join(ROOT, "tests", "golden", golden_file), "rb").read()
    if flag == "--sbs" and not os.path.exists(orc.FIXED_TIME):
        assert p.stdout.count(b"\n") == want.count(b"\n")          # real clock: positions may pick the other frame
    else:
        assert p.stdout == want, p.stderr[-400:]


def test_sharded_detection_equals_whole(torch_cuda, streams):
    """Buffers split over 'ranks' with only the 476-byte carry shared: same records."""
    from dump1090_amd import Demodulator, block_count, shard_blocks, shard_byte_range
    data = streams["edges_smear"]
    d = Demodulator(keep_candidates=True, aggressive=True)
    whole_dev = to_dev(torch_cuda, data)
    d.detect(whole_dev)
    whole, whole_c, _ = d.fetch()
    nb = block_count(data.size)
    for world in (2, 3, 8):
        parts, cparts = [], []
        for r in range(world):
            first, n = shard_blocks(nb, world, r)
            if n == 0:
                continue
            lo, hi = shard_byte_range(first, n, data.size)
            assert hi > lo
            # once as the rank's own (aligned) allocation, once as an unaligned view of the big buffer
            d.detect(to_dev(torch_cuda, data[lo:hi]), stream_byte0=lo, first_block=first, nblocks=n)
            r_own, c_own, _ = d.fetch()
            d.detect(whole_dev[lo:hi], stream_byte0=lo, first_block=first, nblocks=n)
            rr, cc, _ = d.fetch()
            assert np.array_equal(rr, r_own) and np.array_equal(cc, c_own)
            parts.append(rr)
            cparts.append(cc)
        assert np.array_equal(np.concatenate(parts), whole), world
        assert np.array_equal(np.concatenate(cparts), whole_c), world
    d.close()


def test_tuning_parameters_do_not_change_results(torch_cuda, streams):
    from dump1090_amd import Demodulator, ModesError
    data = streams["frames"]
    iq = to_dev(torch_cuda, data)
    base = None
    for rc, dv in ((0, 0), (1, 0), (3, 3), (16, 2), (64, 3), (5, 2), (64, 2), (2, 0), (7, 3)):     # run length x demodulation path
        d = Demodulator(keep_candidates=True, run_chunks=rc, demod_variant=dv)
        d.detect(iq)
        recs, cands, _ = d.fetch()
        if base is None:
            base = (recs, cands)
        assert np.array_equal(recs, base[0]) and np.array_equal(cands, base[1]), (rc, dv)
        d.close()
    with pytest.raises(ModesError, match="scan_variant 1"):      # the single-pass first version of the scan is gone (round 4)
        Demodulator(scan_variant=1)
    d = Demodulator(run_chunks=64, slot_cap=1)          # far too few slots: must fail, not drop
    d.detect(iq)
    with pytest.raises(ModesError, match="MODES_ERR_OVERFLOW"):
        d.fetch()
    d.close()


def test_ragged_and_empty_streams(torch_cuda):
    from dump1090_amd import Demodulator, raw_text
    d = Demodulator(keep_candidates=True, check_crc=False)
    for n in (0, 2, 30, 478, 1024, 262144, 262146, 300001):
        data = synth.frames_stream(77, 2, spacing=900, amp=(50, 90))[0][:n].copy()
        want, st = orc.run_stream(data, **orc.FLAGSETS["nocrc"])
        msgs = d.demodulate(data)                       # host-buffer path handles any length
        assert raw_text(msgs) == orc.raw_text(want), n
        assert d.last["stats_text"] == orc.stats_text(st), n
    d.close()


def test_full_size_noise_properties(torch_cuda):
    """BASELINE config 2 at full size (1 GiB of sigma=3 noise, --no-fix): size-independent checks.
    (a) preamble positions of the first 8 MiB equal the oracle's; (b) the whole-stream result is the
    concatenation of 4 independent quarter-stream results (shard additivity); (c) no messages."""
    from dump1090_amd import Demodulator, HostResolver, block_count, shard_blocks, shard_byte_range
    torch = torch_cuda
    n = 1 << 30
    d = Demodulator(keep_candidates=True, fix=False)
    iq = torch.empty(n, dtype=torch.uint8, device="cuda:0")
    d.synth_noise(iq, 0, seed=2024, sigma_q16=941)
    d.fill(iq[-480:], 127)
    d.detect(iq)
    recs, cands, info = d.fetch()
    head = synth.noise_bytes(2024, 0, 8 << 20, 941)
    _, want_c = oracle_records(head, 0, blocks=range(31))
    assert np.array_equal(cands[cands < 31 * 131072], want_c)
    nb = block_count(n)
    got_r, got_c = [], []
    for r in range(4):
        first, cnt = shard_blocks(nb, 4, r)
        lo, hi = shard_byte_range(first, cnt, n)
        d.detect(iq[lo:hi], stream_byte0=lo, first_block=first, nblocks=cnt)
        rr, cc, _ = d.fetch()
        got_r.append(rr)
        got_c.append(cc)
    assert np.array_equal(np.concatenate(got_c), cands) and np.array_equal(np.concatenate(got_r), recs)
    assert 2.0e-4 < cands.size / (n / 2) < 1.0e-3          # ~5e-4 preambles per sample on this noise
    assert HostResolver(fix=False).resolve(recs, cands) == []
    d.close()


def test_slot_overflow_is_retried_not_dropped(torch_cuda):
    """A periodic preamble-like signal (period 15 samples: pulses at 0, 2, 7, 9) makes 1/15 of ALL positions
    valid preambles - more than the automatic per-run slot lists (1/16) hold.  The library must notice,
    repeat the call with worst-case lists and still return exactly the oracle's records."""
    from dump1090_amd import Demodulator
    n = 2 * synth.DATA_LEN
    iq = np.full(n, 127, dtype=np.uint8)
    s = np.arange(n // 2)
    pulse = np.isin(s % 15, (0, 2, 7, 9))
    iq[0::2][pulse] = 210
    iq[1::2][pulse] = 60
    iq[-480:] = 127
    d = Demodulator(keep_candidates=True, check_crc=False)
    d.detect(to_dev(torch_cuda, iq))
    recs, cands, info = d.fetch()
    want_r, want_c = oracle_records(iq, 1)
    assert want_c.size > (n // 2) // 16                                  # the scenario really overflows 1/16
    assert np.array_equal(cands, want_c)
    assert_records_equal(recs, want_r, "periodic")
    d.detect(to_dev(torch_cuda, iq))                                     # the context keeps the big lists
    recs2, cands2, _ = d.fetch()
    assert np.array_equal(cands2, want_c) and recs2.size == recs.size
    d.close()


def test_every_lane_pushes(torch_cuda):
    """A period-16 signal with TWO valid preambles per period (9 and 7 samples apart): every lane of every scan wavefront owns an
    ordering survivor in every chunk - 64 queue entries from one chunk, more than the scan kernel's queue holds (59 since round 4:
    the push hands the first 59 to the level pass and queues the rest).  Shifted copies move the pattern across the lanes' 8-sample
    windows; 1/8 of all positions are preambles, so the per-run slot lists overflow and are retried too."""
    from dump1090_amd import Demodulator
    level = np.array([2, 0, 14, 20, 2, 16, 0, 2, 1, 0, 10, 1, 20, 1, 10, 0])        # preambles at 3 and 12 (mod 16)
    n = 2 * synth.DATA_LEN
    iq = np.full(n, 127, dtype=np.uint8)
    ns = n // 2
    seg = ns // 8
    for k in range(8):                                                   # 8 segments, the pattern shifted by k (and k + 8 inside)
        s = np.arange(seg - 64)
        iq[0::2][k * seg:k * seg + seg - 64] = 127 + 4 * level[(s + k) % 16]
    iq[-480:] = 127
    for run_chunks in (0, 2):                                            # 2: every run is two chunks - the pass at the end of a run too
        d = Demodulator(keep_candidates=True, check_crc=False, run_chunks=run_chunks)
        d.detect(to_dev(torch_cuda, iq))
        recs, cands, info = d.fetch()
        want_r, want_c = oracle_records(iq, 1)
        assert want_c.size > ns // 9                                     # two preambles per 16 samples
        assert np.array_equal(cands, want_c)
        assert_records_equal(recs, want_r, "every lane pushes")
        d.close()


def test_dense_capture_grows_record_list_and_matches_reference(torch_cuda):
    """The reference's own capture is far denser than any real feed (~750 records per MiB).  Tiled to
    1.5 GiB it yields more records than the automatic list capacity (2^20): the library must grow the
    list and repeat the call, and the ordered --raw listing must still equal the reference's.  A context
    with an explicit, too small max_records must fail with MODES_ERR_OVERFLOW instead (never truncate)."""
    import hashlib
    import os
    from dump1090_amd import Demodulator, raw_text
    torch = torch_cuda
    one = synth.modes1_padded(os.path.join(os.path.dirname(__file__), "golden", "modes1.bin"))
    reps = (3 << 29) // one.size
    host = np.tile(one, reps)
    iq = torch.from_numpy(host).to("cuda:0")
    d = Demodulator()
    msgs = d.demodulate(iq, batch_blocks=8192)
    assert d.last["n_records"] > (1 << 20)
    got = raw_text(msgs)
    if orc.have_ref():
        want = orc.run_ref_bytes(host, ["--raw"]).decode()
    else:
        want = orc.raw_text(orc.run_stream(host, cap=1 << 22, **orc.FLAGSETS["default"])[0])
    assert got.count("\n") == want.count("\n")
    assert hashlib.md5(got.encode()).hexdigest() == hashlib.md5(want.encode()).hexdigest()
    d.close()
    small = Demodulator(max_records=4096)
    small.detect(iq[:64 * synth.DATA_LEN])
    with pytest.raises(RuntimeError, match="OVERFLOW"):
        small.fetch()
    small.close()


def test_no_retry_reports_overflow_and_the_resubmitted_call_succeeds(torch_cuda):
    """MODES_GPU_NO_RETRY (a host that cannot keep its input alive until fetch): both overflow cases come back as
    MODES_ERR_OVERFLOW with the lists already enlarged; the host resubmits the span and gets the full result."""
    from dump1090_amd import Demodulator, ModesError
    torch = torch_cuda
    # (a) slot lists: the periodic preamble-like signal of test_slot_overflow_is_retried_not_dropped
    n = 2 * synth.DATA_LEN
    iq = np.full(n, 127, dtype=np.uint8)
    s = np.arange(n // 2)
    pulse = np.isin(s % 15, (0, 2, 7, 9))
    iq[0::2][pulse] = 210
    iq[1::2][pulse] = 60
    iq[-480:] = 127
    dev = to_dev(torch, iq)
    d = Demodulator(keep_candidates=True, check_crc=False, no_retry=True)
    d.detect(dev)
    with pytest.raises(ModesError, match="MODES_ERR_OVERFLOW"):
        d.fetch()
    d.detect(dev)
    recs, cands, _ = d.fetch()
    want_r, want_c = oracle_records(iq, 1)
    assert np.array_equal(cands, want_c)
    assert_records_equal(recs, want_r, "periodic, resubmitted")
    d.close()
    # (b) record list: the reference's capture tiled until it holds more records than the automatic capacity (2^18)
    one = synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin"))
    host = np.tile(one, (512 << 20) // one.size)
    dev = torch.from_numpy(host).to("cuda:0")
    d = Demodulator(no_retry=True)
    d.detect(dev)
    with pytest.raises(ModesError, match="MODES_ERR_OVERFLOW"):
        d.fetch()
    d.detect(dev)
    recs, _, info = d.fetch()
    assert info["n_records"] > (1 << 18)
    per_tile, _ = oracle_records(one, 1)
    # every tile but the first repeats the same records (the first tile has no carry-in from a previous one)
    nb = one.size // synth.DATA_LEN
    tile5 = recs[(recs["block"] >= 5 * nb) & (recs["block"] < 6 * nb)].copy()
    tile9 = recs[(recs["block"] >= 9 * nb) & (recs["block"] < 10 * nb)].copy()
    tile5["block"] -= 5 * nb
    tile9["block"] -= 9 * nb
    assert np.array_equal(tile5, tile9) and abs(int(tile5.size) - int(per_tile.size)) < 8
    assert np.all(np.diff(recs["block"].astype(np.int64) * 131072 + recs["j"]) > 0)      # strictly ascending: device order
    d.close()


random_stream = synth.random_stream          # the generator of the randomized differential tests (tests/synth.py)


@pytest.mark.parametrize("group", range(8))
def test_randomized_streams_match_oracle(torch_cuda, group):
    """64 seeded random streams (8 per group) x 3 flag sets x both demodulation paths: every record, every preamble
    position, bit for bit - whatever mix of density, SNR, leak, bit errors and hostile stretches the generator draws."""
    from dump1090_amd import Demodulator, raw_text
    demods = {(v, name): Demodulator(keep_candidates=True, demod_variant=v, **orc.FLAGSETS[name])
              for v in (2, 3) for name in ("default", "aggressive", "nofix")}
    total = 0
    for i in range(group * 8, group * 8 + 8):
        data, kw = random_stream(i)
        iq = to_dev(torch_cuda, data)
        for name in ("default", "aggressive", "nofix"):
            want, want_cands = oracle_records(data, maxfix_of(orc.FLAGSETS[name]))
            total += want.size
            for v in (2, 3):
                d = demods[(v, name)]
                d.detect(iq)
                recs, cands, info = d.fetch()
                assert np.array_equal(cands, want_cands), (i, kw, name, v, "preamble positions")
                assert_records_equal(recs, want, ctx=(i, kw, name, v))
            if orc.have_ref():                                # and end to end: the listing the compiled reference prints
                cli = {"default": [], "aggressive": ["--aggressive"], "nofix": ["--no-fix"]}[name]
                msgs = demods[(3 - (i & 1), name)].demodulate(iq)      # the two paths in turn
                assert raw_text(msgs) == orc.run_ref_bytes(data, ["--raw"] + cli).decode(), (i, kw, name, "listing")
    for d in demods.values():
        d.close()
    assert total > 500                                        # the group did exercise the demodulator
