"""-m gpu: BASELINE.json configs at their FULL sizes, through the C ABI, against the reference.

The reference binary (oracle/_ref, compiled from /root/reference by oracle/Makefile and shipped with
the snapshot) is fed the very same bytes through stdin; when it is absent the C restatement
(oracle/liboracle.so) stands in.  These are the only tests that move whole streams back to the host.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle as orc
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch


def reference_stdout(data: np.ndarray, flagset: str, mode: str) -> str:
    """What `dump1090 --ifile <data> <mode> <flags>` prints (mode: --raw | --stats)."""
    fl = orc.FLAGSETS[flagset]
    if orc.have_ref():
        args = [mode] + ([] if fl["fix"] else ["--no-fix"]) + (["--aggressive"] if fl["aggressive"] else []) + \
               ([] if fl["check_crc"] else ["--no-crc-check"])
        return orc.run_ref_bytes(data, args).decode()
    msgs, st = orc.run_stream(data, cap=1 << 22, **fl)
    return orc.raw_text(msgs) if mode == "--raw" else orc.stats_text(st)


def build_on_device(torch, d, st: synth.SparseFrameStream, by_deltas=False):
    """synth_noise + scatter of the frames' footprints + 127-tail: the stream `st`, resident in HBM.  The footprints come
    finished from the host (st.patches(): numpy's copy of the noise under them) or - by_deltas, what bench.py does: seconds
    instead of a minute for 524,287 frames - as what the frames ADD to the device's own noise (st.deltas())."""
    iq = torch.empty(st.nbytes, dtype=torch.uint8, device="cuda:0")
    d.synth_noise(iq, 0, seed=st.seed, sigma_q16=st.sigma_q16)
    if by_deltas:
        first, delta = st.deltas()
        cols = torch.arange(delta.shape[1], device="cuda:0")[None, :]
        for a in range(0, len(first), 65536):
            idx = torch.from_numpy(first[a:a + 65536]).to("cuda:0")[:, None] + cols
            keep = idx < st.nbytes
            at = idx[keep]
            iq[at] = (iq[at].to(torch.int16) + torch.from_numpy(delta[a:a + 65536]).to("cuda:0")[keep]).clamp_(0, 255).to(torch.uint8)
        first = first[:0]
    else:
        first, data = st.patches()
    if len(first):
        idx = torch.from_numpy(first).to("cuda:0")[:, None] + torch.arange(data.shape[1], device="cuda:0")[None, :]
        iq[idx.reshape(-1)] = torch.from_numpy(data).to("cuda:0").reshape(-1)
    d.fill(iq[-480:], 127)
    torch.cuda.synchronize()
    return iq


def test_config2_one_gib_noise_stats_match_reference(torch_cuda):
    """configs[1]: 1 GiB sigma=3 noise, --no-fix.  Every counter of --stats (valid preambles, phase
    corrected retries, demodulated, good/bad CRC ...) and the (empty) --raw listing equal the reference's."""
    from dump1090_amd import Demodulator, raw_text
    torch = torch_cuda
    d = Demodulator(keep_candidates=True, fix=False)
    iq = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
    d.synth_noise(iq, 0, seed=20260922, sigma_q16=941)
    d.fill(iq[-480:], 127)
    msgs = d.demodulate(iq)
    host = iq.cpu().numpy()
    assert d.last["stats_text"] == reference_stdout(host, "nofix", "--stats")
    assert raw_text(msgs) == reference_stdout(host, "nofix", "--raw")
    assert d.last["n_preambles"] > 200000
    d.close()


@pytest.mark.parametrize("nblocks", [32768])
def test_config3_eight_gib_frames_listing_matches_reference(torch_cuda, nblocks):
    """configs[2]: 8 GiB of sigma=3 noise with ~65,000 DF11/DF17 frames at hashed offsets (amplitude
    40..100, random phase, one in ten with a flipped data bit, some on the buffer seams of SURVEY 3.3),
    --fix.  The ordered --raw listing equals the reference's byte for byte, and (analytic check)
    all frames away from the two never-tested seam offsets come out, repaired where a bit was flipped."""
    from dump1090_amd import Demodulator, raw_text
    torch = torch_cuda
    free, _ = torch.cuda.mem_get_info()
    if free < (nblocks * synth.DATA_LEN) * 1.3:
        pytest.skip("not enough free HBM for the 8 GiB stream")
    st = synth.config3_stream(3, nblocks)
    d = Demodulator()
    iq = build_on_device(torch, d, st)
    # spot check: device bytes == host definition around a few frames and at the ends
    for lo in (0, 2 * st.placements[777][0] - 64, st.nbytes - 4096):
        lo -= lo % 2
        assert np.array_equal(iq[lo:lo + 4096].cpu().numpy(), st.window(lo, lo + 4096))
    # one GPU call of 32767 buffers (8 GiB - 256 KiB + carry: positions up to 2^32 - 2^17, the 32-bit
    # limit of a call) and one with the remaining two buffers
    msgs = d.demodulate(iq, batch_blocks=32767)
    got = raw_text(msgs)
    want = reference_stdout(iq.cpu().numpy(), "default", "--raw")
    assert hashlib.md5(got.encode()).hexdigest() == hashlib.md5(want.encode()).hexdigest(), \
        "listing differs from the reference (%d vs %d lines)" % (got.count("\n"), want.count("\n"))
    # analytic expectation: original (unflipped) frame bytes, in stream order
    listed = set(got.split())
    missing = testable = 0
    for sample, fb, amp, phase, smear in st.placements:
        j = (sample + synth.CARRY) % synth.BLOCK_STRIDE
        if j >= 131070:
            continue                                                 # never tested (Q1)
        testable += 1
        missing += ("*" + st.clean[sample].hex() + ";") not in listed
    assert missing <= testable // 200, "%d of %d injected frames not decoded" % (missing, testable)
    assert len(msgs) >= 0.99 * testable
    d.close()
    # bench.py's frames leg demodulates this very stream: leave the reference's verdict where it can be committed
    # as tests/golden/config2_listing.json (the leg then checks its listing against the reference's md5)
    if orc.have_ref() and nblocks == 32768:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "config2_listing.json"), "w") as f:
            json.dump({"stream": "synth.config3_stream(3, 32768)", "flags": "--raw", "lines": want.count("\n"),
                       "md5": hashlib.md5(want.encode()).hexdigest(), "by": "oracle/_ref/dump1090_ref (the compiled reference)"}, f)


def test_config5_low_snr_aggressive_matches_reference(torch_cuda):
    """configs[4] on one GPU: 1 GiB of sigma=3 noise with ~32,000 weak frames (amplitude 8..15 LSB,
    20-40 % inter-sample leak so the phase-corrected retry matters, 5 % two-bit errors), --aggressive.
    --raw listing and every --stats counter equal the reference's."""
    from dump1090_amd import Demodulator, raw_text
    torch = torch_cuda
    st = synth.config3_stream(5, 4096, per=16384, amp=(8, 15), smear=(3, 4, 5, 6), flip1=10, flip2=20, edge_every=61)
    d = Demodulator(keep_candidates=True, aggressive=True)
    iq = build_on_device(torch, d, st)
    msgs = d.demodulate(iq, batch_blocks=1500)                     # three GPU calls; whitelist carried across
    host = iq.cpu().numpy()
    assert raw_text(msgs) == reference_stdout(host, "aggressive", "--raw")
    assert d.last["stats_text"] == reference_stdout(host, "aggressive", "--stats")
    assert len(msgs) > 1000                                         # the path is exercised, not vacuous
    assert any(m.phase_corrected for m in msgs) and any(m.errorbit >= 0 for m in msgs)
    # configs[4] names 8 GPUs: the same stream cut into the 8 buffer ranges bench.py --gpus 8 gives its ranks, each range
    # demodulated from exactly the bytes that rank would hold, the lists concatenated in rank order and resolved once
    # (on 16 threads, like rank 0 does) - the listing must not notice
    from dump1090_amd import HostResolver, block_count, shard_blocks, shard_byte_range
    total = block_count(st.nbytes)
    parts = []
    d8 = Demodulator(aggressive=True)
    for rank in range(8):
        first, n = shard_blocks(total - 1, 8, rank)
        if rank == 7:
            n += 1
        lo, hi = shard_byte_range(first, n, st.nbytes)
        d8.detect(iq[lo:hi].clone(), stream_byte0=lo, first_block=first, nblocks=n)     # the rank's own allocation
        parts.append(d8.fetch()[0])
    res = HostResolver(aggressive=True)
    n_lines, text = res.raw_listing(np.concatenate(parts), None, threads=16)
    res.close()
    assert text.decode() == raw_text(msgs) and n_lines == len(msgs)
    d8.close()
    d.close()


def test_config4_sixty_four_gib_in_eight_shards_matches_reference(torch_cuda):
    """configs[3]'s data size on ONE GPU: the 64 GiB stream (524,287 frames) is cut into the 8 buffer ranges
    bench.py --gpus 8 would give its ranks, each range is demodulated on its own from exactly the bytes
    that rank would hold (its buffers + the 476-byte carry), the record lists are concatenated in rank
    order and resolved once - the N = 8 data path without the 8 GPUs.  The listing must equal the
    reference's on the whole stream."""
    from dump1090_amd import Demodulator, HostResolver, block_count, raw_text, shard_blocks, shard_byte_range
    torch = torch_cuda
    nblocks = 262144
    free, _ = torch.cuda.mem_get_info()
    if free < nblocks * synth.DATA_LEN * 1.2:
        pytest.skip("not enough free HBM for the 64 GiB stream")
    st = synth.config3_stream(4, nblocks)
    d = Demodulator()
    iq = build_on_device(torch, d, st, by_deltas=True)
    total = block_count(st.nbytes)
    recs = []
    for rank in range(8):
        first, n = shard_blocks(total - 1, 8, rank)
        if rank == 7:
            n += 1                                               # the EOF buffer goes to the last rank (bench.py)
        lo, hi = shard_byte_range(first, n, st.nbytes)
        shard = iq[lo:hi]                                        # what this rank would hold
        for b0 in range(first, first + n, 16384):                # <= 4 GiB per GPU call
            nb = min(16384, first + n - b0)
            blo, bhi = shard_byte_range(b0, nb, st.nbytes)
            d.detect(shard[blo - lo:bhi - lo], stream_byte0=blo, first_block=b0, nblocks=nb)
            r, _, _ = d.fetch()
            recs.append(r)
    res = HostResolver()
    got = raw_text(res.resolve(np.concatenate(recs), None))
    res.close()
    # The reference's verdict on this very stream is committed (tests/golden/config_listings.json: the compiled reference fed
    # the same bytes by tests/golden/gen_stream.c in the build container); MODES_LIVE_REFERENCE=1 also runs the binary here
    # (3 more minutes and 64 GiB of host memory).
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_listings.json")) as f:
        gold = json.load(f)["frames:4:%d" % nblocks]
    assert got.count("\n") > 500000
    assert (got.count("\n"), hashlib.md5(got.encode()).hexdigest()) == (gold["lines"], gold["md5"]), \
        "listing differs from the reference's (%d vs %d lines)" % (got.count("\n"), gold["lines"])
    if os.environ.get("MODES_LIVE_REFERENCE") == "1":
        want = reference_stdout(iq.cpu().numpy(), "default", "--raw")
        assert hashlib.md5(got.encode()).hexdigest() == hashlib.md5(want.encode()).hexdigest()
    d.close()


def test_hosts_on_the_eight_gib_file_print_the_reference_listing(torch_cuda):
    """BASELINE configs[2] as a FILE (8 GiB in /dev/shm, the generator's stream of seed 3) through the two hosts a user would run -
    the C++ host and the reference's own main() with the batched patch (integration/dump1090_gfx950_batched.patch, K = 512 buffers per
    hand-off) - and, with --aggressive, configs[4]'s 1 GiB low-SNR file: stdout == the compiled reference's listing of the very
    stream (tests/golden/config_listings.json: 65,519 / 2,519 lines), byte for byte."""
    import hashlib
    import json
    import shutil
    import subprocess
    import bench
    torch = torch_cuda
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.disk_usage("/dev/shm").free < 10 * 2 ** 30:
        pytest.skip("needs 9 GiB of /dev/shm")
    batched = os.path.join(root, "oracle", "_ref", "dump1090_dropin_batched")
    cxx = os.path.join(root, "dump1090_amd", "bin", "dump1090_amd")
    assert os.path.exists(batched), "oracle/_ref/dump1090_dropin_batched must travel with the snapshot"
    with open(os.path.join(root, "tests", "golden", "config_listings.json")) as f:
        gold = json.load(f)
    dev = torch.device("cuda", 0)
    for key, kw, flags in (("frames:3:32768", {}, ["--raw"]), ("lowsnr:5:4096", bench.LOWSNR, ["--raw", "--aggressive"])):
        kind, seed, nblocks = key.split(":")
        seed, nblocks = int(seed), int(nblocks)
        path = "/dev/shm/modes_fullsize_%s.bin" % kind
        try:
            with open(path, "wb") as f:                                      # built on the GPU, 1 GiB at a time
                for lo in range(0, nblocks * 262144, 1 << 30):
                    iq, _ = bench.build_frames_shard(torch, dev, nblocks, lo, min(nblocks * 262144, lo + (1 << 30)), seed=seed, **kw)
                    iq.cpu().numpy().tofile(f)
                    del iq
            torch.cuda.empty_cache()
            for exe, env in ((cxx, None), (batched, dict(os.environ, MODES_DROPIN_BLOCKS="512"))):
                p = subprocess.run([exe, "--ifile", path] + flags, capture_output=True, env=env, timeout=600)
                assert p.returncode == 0, p.stderr[-400:]
                got = (p.stdout.count(b"\n"), hashlib.md5(p.stdout).hexdigest())
                assert got == (gold[key]["lines"], gold[key]["md5"]), (key, os.path.basename(exe), got)
        finally:
            if os.path.exists(path):
                os.remove(path)
