"""The CPU oracle (oracle/modes_oracle.c) pinned against the reference.

Pins, in order of strength:
  1. the md5 / line counts SURVEY.md 4.2 and BASELINE.md 4 record for the
     reference on the padded testfiles/modes1.bin (6 --raw flag sets, --onlyaddr,
     3 --stats flag sets);
  2. exact stdout of oracle/_ref/dump1090_ref (the unmodified reference) on 8
     seeded synthetic streams, committed as tests/golden/golden.json.gz by
     tests/golden/make_golden.py;
  3. (build container only) a live re-run of oracle/_ref, and the reference's
     literal CRC table parsed from /root/reference/dump1090.c.
"""
import hashlib
import os
import re

import numpy as np
import pytest

import oracle as orc
import synth

CASES = ["modes1", "uniform", "coarse", "edges", "edges_smear", "frames", "smear", "lowsnr", "noise", "saturated"]

# BASELINE.md section 4 / SURVEY.md 4.2 (reference stdout on the padded fixture)
PUBLISHED_MODES1 = {
    "default": (284, "4a81758c8bec5e45ffa8541c5622938a"),
    "nofix": (283, "ac539444a66eb99a7f04affa95c55079"),
    "aggressive": (284, "4a81758c8bec5e45ffa8541c5622938a"),
    "nocrc": (765, "a6092d178fcf0d09a730ab67f2a43724"),
    "nofix_nocrc": (770, "d4d71bcac73e54d5d58c660346bae0cd"),
    "aggressive_nocrc": (824, "bec25488d6b84e9b0703d164de1cc873"),
}
PUBLISHED_MODES1_STATS = {
    "default": "bc3d1c04b24f4989f0fc4a2d1f45abdd",
    "nofix": "4212141c5fcc7e9b07455f778d6e2a04",
    "aggressive": "6497039fa2c045d3e64ad85ab68edaef",
}


def md5(s: str) -> str:
    return hashlib.md5(s.encode()).hexdigest()


@pytest.mark.parametrize("flagset", list(orc.FLAGSETS))
def test_modes1_published_hashes(streams, flagset):
    msgs, _ = orc.run_stream(streams["modes1"], **orc.FLAGSETS[flagset])
    text = orc.raw_text(msgs)
    assert (text.count("\n"), md5(text)) == PUBLISHED_MODES1[flagset]


def test_modes1_onlyaddr_and_stats(streams):
    msgs, st = orc.run_stream(streams["modes1"], **orc.FLAGSETS["default"])
    assert md5(orc.onlyaddr_text(msgs)) == "bab0f055e262e216208a5cbbdf63fe24"
    for fs in ("default", "nofix", "aggressive"):
        _, st = orc.run_stream(streams["modes1"], **orc.FLAGSETS[fs])
        assert md5(orc.stats_text(st)) == PUBLISHED_MODES1_STATS[fs]


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_committed_reference_output(golden, streams, case):
    data = streams[case]
    g = golden[case]
    assert data.size == g["nbytes"] and hashlib.md5(data.tobytes()).hexdigest() == g["input_md5"], \
        "synthetic stream generator drifted from the committed goldens"
    for fs, flags in orc.FLAGSETS.items():
        msgs, st = orc.run_stream(data, **flags)
        assert orc.raw_text(msgs) == g["raw"][fs]["text"], (case, fs)
        if fs in g["stats"]:
            assert orc.stats_text(st) == g["stats"][fs]["text"], (case, fs)
    msgs, _ = orc.run_stream(data, **orc.FLAGSETS["default"])
    assert orc.onlyaddr_text(msgs) == g["onlyaddr"]["default"]["text"]


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
def test_live_reference_agrees(tmp_path, streams):
    """Fresh stream the goldens do not contain, straight through the compiled reference."""
    data = synth.frames_stream(4242, 2, spacing=1800, sigma_q16=941, amp=(20, 80),
                               smear=(0, 3, 6), flip1=4, flip2=6)[0]
    path = tmp_path / "live.bin"
    data.tofile(path)
    for fs, cli in (("default", []), ("aggressive_nocrc", ["--aggressive", "--no-crc-check"]),
                    ("nofix", ["--no-fix"])):
        msgs, st = orc.run_stream(data, **orc.FLAGSETS[fs])
        assert orc.raw_text(msgs) == orc.run_ref(str(path), ["--raw"] + cli)
        if fs != "aggressive_nocrc":
            assert orc.stats_text(st) == orc.run_ref(str(path), ["--stats"] + cli)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("group", range(4))
def test_restatement_equals_compiled_reference_on_random_streams(group):
    """The streams of the GPU suite's randomized differential test (synth.random_stream: density, SNR, leak, bit errors, seam
    frames, hostile stretches - all drawn from a seeded generator), through the compiled reference and through the
    restatement: the same --raw listing and the same --stats counters.  Reference = restatement here, restatement = HIP
    path there (tests/test_gpu_parity.py::test_randomized_streams_match_oracle), on the very same bytes."""
    total = 0
    for i in range(group * 16, group * 16 + 16):
        data, kw = synth.random_stream(i)
        for fs, cli in (("default", []), ("aggressive", ["--aggressive"]), ("nofix", ["--no-fix"])):
            msgs, st = orc.run_stream(data, **orc.FLAGSETS[fs])
            assert orc.raw_text(msgs) == orc.run_ref_bytes(data, ["--raw"] + cli).decode(), (i, kw, fs)
            assert orc.stats_text(st) == orc.run_ref_bytes(data, ["--stats"] + cli).decode(), (i, kw, fs)
            total += len(msgs)
    assert total > 200


def test_maglut_matches_formula_and_is_monotone_in_s():
    lut = orc.maglut().reshape(129, 129)
    i, q = np.meshgrid(np.arange(129), np.arange(129), indexing="ij")
    s = i * i + q * q
    want = np.floor(np.sqrt(s.astype(np.float64)) * 360.0 + 0.5).astype(np.uint16)
    assert np.array_equal(lut, want)
    # exact integer form: (isqrt(518400*s)+1)//2
    import math
    exact = np.array([(math.isqrt(518400 * int(v)) + 1) // 2 for v in s.ravel()], dtype=np.uint16).reshape(129, 129)
    assert np.array_equal(lut, exact)
    # strictly monotone in s: ordering tests on s are ordering tests on magnitude
    order = np.argsort(s.ravel(), kind="stable")
    ss, mm = s.ravel()[order], lut.ravel()[order].astype(np.int64)
    ds, dm = np.diff(ss), np.diff(mm)
    assert np.all(dm[ds > 0] > 0) and np.all(dm[ds == 0] == 0)


def test_magnitude_all_byte_pairs():
    iq = np.stack(np.meshgrid(np.arange(256), np.arange(256), indexing="ij"), -1).astype(np.uint8).reshape(-1)
    mag = orc.magnitude(iq)
    i = np.abs(iq[0::2].astype(np.int64) - 127)
    q = np.abs(iq[1::2].astype(np.int64) - 127)
    assert np.array_equal(mag, orc.maglut()[i * 129 + q])
    assert mag.max() == 65167 and mag[127 * 256 + 127] == 0


def test_crc_table_against_reference_source():
    tab = [orc.lib().orc_crc_table_entry(i) for i in range(112)]
    assert tab[87] == 0xFFF409 and tab[0] == 0x3935EA and all(t == 0 for t in tab[88:])
    assert tab == synth.CRC_TABLE
    src = "/root/reference/dump1090.c"
    if os.path.exists(src):
        text = open(src).read()
        body = text[text.index("modes_checksum_table[112]"):]
        body = body[: body.index("};")]
        ref = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{6}", body)]
        assert ref == tab


def test_framing_and_block_count():
    n = 3 * orc.DATA_LEN + 1000
    stream = (np.arange(n) % 251).astype(np.uint8)
    assert orc.block_count(n) == 4 and orc.block_count(3 * orc.DATA_LEN) == 4 and orc.block_count(0) == 1
    b0 = orc.frame_block(stream, 0)
    assert np.all(b0[:476] == 127) and np.array_equal(b0[476:], stream[: orc.DATA_LEN])
    b1 = orc.frame_block(stream, 1)
    assert np.array_equal(b1, stream[orc.DATA_LEN - 476: 2 * orc.DATA_LEN])
    b3 = orc.frame_block(stream, 3)
    assert np.array_equal(b3[: 476 + 1000], stream[3 * orc.DATA_LEN - 476:]) and np.all(b3[476 + 1000:] == 127)


def test_records_reproduce_detect(streams):
    """The stateless per-position records + a sequential resolve written here in
    Python give the same listing as the oracle's reference-shaped loop: the
    decomposition the GPU path relies on is itself pinned."""
    data = streams["smear"]
    flags = orc.FLAGSETS["aggressive_nocrc"]
    want, _ = orc.run_stream(data, **flags)
    got = []
    for k in range(orc.block_count(data.size)):
        mag = orc.block_magnitude(data, k)
        js = orc.block_candidates(mag)
        recs = orc.records(mag, js, 2)
        got.append((k, js, recs))
    # candidates that pass attempt-0's gate and are demodulable are a superset of emitted positions
    emitted = {(m.block, m.j) for m in want}
    cands = {(k, int(j)) for k, js, _ in got for j in js}
    assert emitted <= cands
    for k, js, recs in got:
        for r in recs:
            if (k, int(r["j"])) in emitted:
                assert r["att"][0]["gate_ok"] == 1
