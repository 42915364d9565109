"""The drop-in claim, executed: integration/dump1090_gfx950.patch applied to the reference's dump1090.c (a
temporary copy; oracle/Makefile does it where /root/reference exists) gives oracle/_ref/dump1090_dropin - the
reference's OWN main(), option parser, reader thread and sink (displayModesMessage / useModesMessage) running on
top of libmodes_gfx950.so + libmodes_host.so instead of computeMagnitudeVector() + detectModeS()
(dump1090.c:2974, :2986).  On the GPU box it must print what the unmodified reference prints."""
import hashlib
import os
import shutil
import subprocess

import pytest

import oracle as orc
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, "oracle", "_ref", "dump1090_dropin")
PATCH = os.path.join(ROOT, "integration", "dump1090_gfx950.patch")
REF = "/root/reference"


BATCHED = os.path.join(ROOT, "oracle", "_ref", "dump1090_dropin_batched")
PATCH_BATCHED = os.path.join(ROOT, "integration", "dump1090_gfx950_batched.patch")
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "dump1090.c")), reason="needs the reference source tree")
CC_ARGS = ["-I", os.path.join(ROOT, "oracle", "stub"), "-I", REF, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "integration")]


@needs_ref
@pytest.mark.parametrize("patch,extra_lines", [(PATCH, 3), (PATCH_BATCHED, 3)])
def test_patch_applies_and_builds_against_the_c_abi(tmp_path, patch, extra_lines):
    """Build box: the committed patches apply to a pristine copy without fuzz, touch exactly the places INTEGRATION.md names
    (four edits; the batched variant a fifth: the file reader's call), and the result compiles (-Werror) and links against the
    two product libraries with gcc."""
    shutil.copy(os.path.join(REF, "dump1090.c"), tmp_path / "dump1090.c")
    p = subprocess.run(["patch", "-p1", "--fuzz=0", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode == 0 and "fuzz" not in p.stdout and "offset" not in p.stdout, p.stdout + p.stderr
    patched = open(tmp_path / "dump1090.c").read()
    orig = open(os.path.join(REF, "dump1090.c")).read()
    assert patched.count("modesGpuDemod();") == 1 and patched.count("modesGpuResolve();") == 1
    assert patched.count("modesInitGpu();") == 1 and patched.count('#include "modes_dropin.c"') == 1
    assert patched.count("modesGpuReadFile();") == (1 if patch == PATCH_BATCHED else 0)
    assert ("    readDataFromFile();" in patched) == (patch == PATCH)           # the reference's reader is still called, or not at all
    assert len(patched.splitlines()) - len(orig.splitlines()) == extra_lines
    import difflib
    changed = [ln for ln in difflib.unified_diff(orig.splitlines(), patched.splitlines(), n=0, lineterm="") if ln[:1] in "+-" and ln[:3] not in ("+++", "---")]
    assert len([ln for ln in changed if ln.startswith("-")]) == (3 if patch == PATCH_BATCHED else 2), changed
    exe = tmp_path / "dropin"
    cc = subprocess.run(["gcc", "-O2", "-Wall", "-W", "-Werror"] + CC_ARGS + ["-o", str(exe),
                         str(tmp_path / "dump1090.c"), os.path.join(REF, "anet.c"), "-L", os.path.join(ROOT, "dump1090_amd"),
                         "-lmodes_gfx950", "-lmodes_host", "-lpthread", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "dump1090_amd"),
                         "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    syms = subprocess.run(["nm", "-u", str(exe)], capture_output=True, text=True).stdout
    for s in ("modes_gpu_create", "modes_gpu_demod_host", "modes_host_resolve", "modes_host_get_stats") + (
            ("modes_gpu_submit_host", "modes_gpu_fetch", "modes_gpu_host_alloc") if patch == PATCH_BATCHED else ()):
        assert s in syms, "the drop-in does not bind " + s


def test_committed_patches_are_what_make_patch_derives(tmp_path):
    """integration/make_patch.py regenerates both patches from the reference source: the committed files are its output."""
    if not os.path.exists(os.path.join(REF, "dump1090.c")):
        pytest.skip("needs the reference source tree")
    work = tmp_path / "integration"
    shutil.copytree(os.path.join(ROOT, "integration"), work)
    subprocess.run(["python3", str(work / "make_patch.py")], check=True, capture_output=True)
    for name in ("dump1090_gfx950.patch", "dump1090_gfx950_batched.patch"):
        assert open(work / name).read() == open(os.path.join(ROOT, "integration", name)).read(), name


REFERENCE_MD5 = [
    (["--raw"], 284, "4a81758c8bec5e45ffa8541c5622938a"),
    (["--raw", "--no-fix"], 283, "ac539444a66eb99a7f04affa95c55079"),
    (["--raw", "--aggressive", "--no-crc-check"], 824, "bec25488d6b84e9b0703d164de1cc873"),
    (["--onlyaddr"], 284, "bab0f055e262e216208a5cbbdf63fe24"),
    (["--stats"], 9, "bc3d1c04b24f4989f0fc4a2d1f45abdd"),
    ([], 3202, "0bf2290fa954f1675437e52508ea8aa3"),
]


@pytest.fixture(scope="module")
def stub_hosted(tmp_path_factory):
    """The patched reference (both patches) linked against tests/native/gpu_stub.cpp - the GPU library's entry points on the
    oracle's stateless functions, TEST SCAFFOLDING - so that the reference's own main loop, the batched reader and the
    hand-off protocol between them run on a machine without a GPU.  -> {"plain": exe, "batched": exe, "batched_tsan": exe}"""
    if not os.path.exists(os.path.join(REF, "dump1090.c")) or not shutil.which("g++"):
        pytest.skip("needs the reference source tree and g++")
    d = tmp_path_factory.mktemp("dropin_stub")
    out = {}
    for name, patch, san in (("plain", PATCH, []), ("batched", PATCH_BATCHED, []), ("batched_tsan", PATCH_BATCHED, ["-fsanitize=thread"])):
        w = d / name
        w.mkdir()
        shutil.copy(os.path.join(REF, "dump1090.c"), w / "dump1090.c")
        subprocess.run(["patch", "-s", "-p1", "--fuzz=0", "-i", patch], cwd=w, check=True)
        run = lambda cmd: subprocess.run(cmd, cwd=w, check=True, capture_output=True)
        run(["gcc", "-O1", "-g", "-Wall", "-W", "-Werror"] + san + CC_ARGS + ["-c", "dump1090.c", "-o", "d.o"])
        run(["gcc", "-O1", "-g"] + san + ["-c", os.path.join(REF, "anet.c"), "-o", "anet.o"])
        run(["gcc", "-O1", "-g"] + san + ["-c", os.path.join(ROOT, "oracle", "modes_oracle.c"), "-o", "orc.o"])
        run(["g++", "-O1", "-g", "-std=c++17"] + san + ["-I", os.path.join(ROOT, "include"), "-o", "dropin", "d.o", "anet.o", "orc.o",
             os.path.join(ROOT, "tests", "native", "gpu_stub.cpp"), os.path.join(ROOT, "dump1090_amd", "csrc", "modes_host.cpp"),
             os.path.join(ROOT, "dump1090_amd", "csrc", "modes_track.cpp"), "-lpthread", "-lm"])
        out[name] = str(w / "dropin")
    pad = d / "modes1_pad.bin"
    synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin")).tofile(pad)
    out["padded"] = str(pad)
    return out


def run_md5(exe, args, env=None, stdin=None):
    p = subprocess.run([exe] + args, capture_output=True, timeout=300, env=env, stdin=stdin)
    assert p.returncode == 0, p.stderr[-600:]
    return p.stdout.count(b"\n"), hashlib.md5(p.stdout).hexdigest()


@pytest.mark.parametrize("k", [1, 2, 7, 64])
def test_batched_reader_hands_over_k_buffers_and_prints_the_reference_listing(stub_hosted, k):
    """The batched patch on CPU (GPU stubbed by the oracle): K buffers per hand-off - K = 1 (the reference's granularity), 2 (the
    padded capture is 3 buffers: a full hand-off, then a short one that carries the EOF buffer), 7 and 64 (the whole file in
    one hand-off) - every flag set of BASELINE.md section 4 prints the reference's bytes; --stats too (the preamble positions
    of a batch travel with its records)."""
    env = dict(os.environ, MODES_DROPIN_BLOCKS=str(k))
    for flags, lines, md5 in REFERENCE_MD5:
        assert run_md5(stub_hosted["batched"], ["--ifile", stub_hosted["padded"]] + flags, env) == (lines, md5), (k, flags)
    if k == 1:
        for flags, lines, md5 in REFERENCE_MD5[:1] + REFERENCE_MD5[4:5]:
            assert run_md5(stub_hosted["plain"], ["--ifile", stub_hosted["padded"]] + flags) == (lines, md5), flags


def test_batched_reader_on_ragged_input_stdin_and_under_tsan(stub_hosted, tmp_path):
    """The reference's own capture as it is (699,392 bytes: two buffers and a ragged third - its unmodified main loop usually
    drops that one; here the reader raises Modes.exit only when the last hand-off has been taken, so every run
    prints the race-free 284 lines), from a file and from a pipe (--ifile -: sequential reads, the same hand-offs); a length
    that is an exact multiple of K buffers (the EOF buffer then travels alone); and the hand-off protocol under
    ThreadSanitizer."""
    raw = os.path.join(ROOT, "tests", "golden", "modes1.bin")
    want = (284, "4a81758c8bec5e45ffa8541c5622938a")
    for k in (1, 2, 3, 64):
        env = dict(os.environ, MODES_DROPIN_BLOCKS=str(k))
        assert run_md5(stub_hosted["batched"], ["--ifile", raw, "--raw"], env) == want, k
        with open(raw, "rb") as f:
            assert run_md5(stub_hosted["batched"], ["--ifile", "-", "--raw"], env, stdin=f) == want, ("stdin", k)
    exact = tmp_path / "exact.bin"                                             # 4 buffers: K = 2 and K = 4 divide it
    data = synth.modes1_padded(raw)
    import numpy as np
    np.concatenate([data, np.full(262144, 127, np.uint8)]).tofile(exact)
    for k in (2, 4):
        assert run_md5(stub_hosted["batched"], ["--ifile", str(exact), "--raw"], dict(os.environ, MODES_DROPIN_BLOCKS=str(k))) == want, k
    # (the reference never joins its reader thread - dump1090.c:2966 - which is a leak report, not a race)
    env = dict(os.environ, MODES_DROPIN_BLOCKS="1", TSAN_OPTIONS="halt_on_error=1:report_thread_leaks=0")
    for k in ("1", "2"):
        env["MODES_DROPIN_BLOCKS"] = k
        assert run_md5(stub_hosted["batched_tsan"], ["--ifile", raw, "--raw"], env) == want


def test_dropin_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    if not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/dump1090_dropin not built (no reference tree on this machine)")
    p = subprocess.run([DROPIN, "--ifile", os.path.join(ROOT, "tests", "golden", "modes1.bin"), "--raw"], capture_output=True, text=True)
    assert p.returncode == 1 and p.stdout == "" and "no HIP device" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("flags,lines,md5", REFERENCE_MD5)
def test_reference_main_on_the_gpu_path_reproduces_reference_stdout(tmp_path, flags, lines, md5):
    """GPU box: the patched reference on its own capture (padded to whole buffers, like every oracle run: SURVEY.md
    3.4) - the md5s of BASELINE.md section 4, which the unmodified binary produces too when it is present."""
    assert os.path.exists(DROPIN), "oracle/_ref/dump1090_dropin must travel with the snapshot (built by __graft_entry__.build())"
    path = tmp_path / "modes1_pad.bin"
    synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin")).tofile(path)
    p = subprocess.run([DROPIN, "--ifile", str(path)] + flags, capture_output=True, check=True)
    assert (p.stdout.count(b"\n"), hashlib.md5(p.stdout).hexdigest()) == (lines, md5), p.stderr[-400:]
    if orc.have_ref():
        env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME)
        ref = subprocess.run([orc.REF_BIN, "--ifile", str(path)] + flags, capture_output=True, check=True, env=env)
        assert ref.stdout == p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 7, 64])
def test_batched_reference_main_on_the_gpu_path_reproduces_reference_stdout(tmp_path, k):
    """GPU box: the reference with the BATCHED patch (K buffers per hand-off, two contexts alternating) - the same six md5s for
    every K, on the padded capture; the ragged capture itself (file and pipe) prints the race-free 284 lines."""
    assert os.path.exists(BATCHED), "oracle/_ref/dump1090_dropin_batched must travel with the snapshot"
    path = tmp_path / "modes1_pad.bin"
    synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin")).tofile(path)
    env = dict(os.environ, MODES_DROPIN_BLOCKS=str(k))
    for flags, lines, md5 in REFERENCE_MD5:
        assert run_md5(BATCHED, ["--ifile", str(path)] + flags, env) == (lines, md5), (k, flags)
    raw = os.path.join(ROOT, "tests", "golden", "modes1.bin")
    assert run_md5(BATCHED, ["--ifile", raw, "--raw"], env) == (284, "4a81758c8bec5e45ffa8541c5622938a")
    with open(raw, "rb") as f:
        assert run_md5(BATCHED, ["--ifile", "-", "--raw"], env, stdin=f) == (284, "4a81758c8bec5e45ffa8541c5622938a")


@pytest.mark.gpu
def test_batched_dropin_on_a_generated_stream_equals_the_cxx_host(tmp_path):
    """64 MiB of the frames generator (seam frames included), K = 7 (not a divisor of its 256 buffers) and the default K: the
    batched drop-in prints what dump1090_amd prints for the same file (which the full-size tests pin to the reference)."""
    st = synth.config3_stream(11, 256)
    path = tmp_path / "frames.bin"
    st.window(0, st.nbytes).tofile(path)
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    want = subprocess.run([exe, "--ifile", str(path), "--raw"], capture_output=True, check=True).stdout
    assert want.count(b"\n") > 200
    for k in ("7", "512"):
        got = subprocess.run([BATCHED, "--ifile", str(path), "--raw"], capture_output=True, check=True, env=dict(os.environ, MODES_DROPIN_BLOCKS=k)).stdout
        assert got == want, k


def _paced(cmd, path, ms=150, repeat=4, env=None):
    """tools/paced_pipe.py: `path` (x repeat) into cmd's stdin one 256 KiB buffer every `ms` milliseconds -> its JSON report."""
    import json
    p = subprocess.run([os.sys.executable, os.path.join(ROOT, "tools", "paced_pipe.py"), str(ms), path, "--repeat", str(repeat), "--"] + cmd,
                       capture_output=True, timeout=300, env=env)
    assert p.returncode == 0, (p.stdout[-400:], p.stderr[-600:])
    return json.loads(p.stdout)


def test_batched_reader_serves_a_pipe_at_the_pace_it_delivers(stub_hosted, tmp_path):
    """The batched patch with its default K = 512 on a pipe that delivers one buffer every 150 ms: the hand-off is what
    has arrived - whole buffers - 66 ms after the batch began, so the first line is printed long before the writer is done; the bytes
    are the file run's (the capture four times over), paced or not."""
    raw = os.path.join(ROOT, "tests", "golden", "modes1.bin")
    four = tmp_path / "four.bin"
    four.write_bytes(open(raw, "rb").read() * 4)
    want = run_md5(stub_hosted["batched"], ["--ifile", str(four), "--raw"])
    d = _paced([stub_hosted["batched"], "--ifile", "-", "--raw"], raw)
    assert (d["lines"], d["md5"]) == want and d["first_output_s"] is not None and d["first_output_s"] < d["writer_done_s"] - 0.2, d
    with open(four, "rb") as f:
        assert run_md5(stub_hosted["batched"], ["--ifile", "-", "--raw"], dict(os.environ, MODES_DROPIN_BLOCKS="3"), stdin=f) == want
    d = _paced([stub_hosted["batched_tsan"], "--ifile", "-", "--raw"], raw, ms=40, repeat=2, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:report_thread_leaks=0"))
    assert d["status"] == 0 and d["lines"] == want[0] // 2


@pytest.mark.gpu
def test_paced_pipe_through_both_hosts_on_the_gpu(tmp_path):
    """GPU box: the C++ host and the batched drop-in on a pipe at the radio's cadence (one buffer per 66 ms) with their default batch
    sizes - first output before the writer is done, stdout byte-identical to the file run; an unpaced pipe too."""
    raw = os.path.join(ROOT, "tests", "golden", "modes1.bin")
    four = tmp_path / "thirty.bin"
    four.write_bytes(open(raw, "rb").read() * 30)
    exe = os.path.join(ROOT, "dump1090_amd", "bin", "dump1090_amd")
    want = run_md5(exe, ["--ifile", str(four), "--raw"])
    assert want[0] > 7500
    for cmd in ([exe, "--ifile", "-", "--raw"], [BATCHED, "--ifile", "-", "--raw"], [exe, "--ifile", "-", "--raw", "--ranks", "1", "--resolve-on-ranks"]):
        d = _paced(cmd, raw, ms=66, repeat=30)
        assert (d["lines"], d["md5"]) == want, (cmd, d)
        # (the process starts the HIP runtime, a second context and two pinned buffers while the first buffers arrive: 0.3-0.5 s of the
        #  writer's 5.9 s - seconds on a box whose libraries are not in the page cache yet; the drop-in prints a hand-off when the next
        #  one is submitted)
        assert d["first_output_s"] is not None and d["first_output_s"] < d["writer_done_s"] - 0.5, (cmd, d)
        with open(four, "rb") as f:
            assert run_md5(cmd[0], cmd[1:], stdin=f) == want, cmd


def _first_bytes(cmd, n, env=None):
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


def test_batched_reader_replays_the_file_like_the_reference(stub_hosted):
    """--loop through the batched reader (dump1090.c:488-494: at the end of the file the reader seeks back and keeps filling the same
    hand-off): the first 2.5 laps of output are the unmodified reference's bytes (where oracle/_ref/dump1090_ref exists), for K = 1
    and for a K that spans several laps."""
    one = subprocess.run([stub_hosted["batched"], "--ifile", stub_hosted["padded"], "--raw"], capture_output=True, check=True,
                         env=dict(os.environ, MODES_DROPIN_BLOCKS="64")).stdout
    n = len(one) * 5 // 2
    laps = {k: _first_bytes([stub_hosted["batched"], "--ifile", stub_hosted["padded"], "--raw", "--loop"], n,
                            dict(os.environ, MODES_DROPIN_BLOCKS=str(k))) for k in (1, 7)}
    assert len(laps[1]) == n and laps[1] == laps[7] and laps[1][:len(one)] == one
    if orc.have_ref():
        ref = _first_bytes([orc.REF_BIN, "--ifile", stub_hosted["padded"], "--raw", "--loop"], n, dict(os.environ, LD_PRELOAD=orc.FIXED_TIME))
        assert ref == laps[1]
