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


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "dump1090.c")), reason="needs the reference source tree")
def test_patch_applies_and_builds_against_the_c_abi(tmp_path):
    """Build box: the committed patch applies to a pristine copy without fuzz, touches exactly the four places
    INTEGRATION.md names, and the result compiles and links against the two product libraries with gcc."""
    shutil.copy(os.path.join(REF, "dump1090.c"), tmp_path / "dump1090.c")
    p = subprocess.run(["patch", "-p1", "--fuzz=0", "-i", PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode == 0 and "fuzz" not in p.stdout and "offset" not in p.stdout, p.stdout + p.stderr
    patched = open(tmp_path / "dump1090.c").read()
    orig = open(os.path.join(REF, "dump1090.c")).read()
    assert patched.count("modesGpuDemod();") == 1 and patched.count("modesGpuResolve();") == 1
    assert patched.count("modesInitGpu();") == 1 and patched.count('#include "modes_dropin.c"') == 1
    assert len(patched.splitlines()) - len(orig.splitlines()) == 3
    exe = tmp_path / "dropin"
    cc = subprocess.run(["gcc", "-O2", "-Wall", "-W", "-Werror", "-I", os.path.join(ROOT, "oracle", "stub"), "-I", REF,
                         "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "integration"), "-o", str(exe),
                         str(tmp_path / "dump1090.c"), os.path.join(REF, "anet.c"), "-L", os.path.join(ROOT, "dump1090_amd"),
                         "-lmodes_gfx950", "-lmodes_host", "-lpthread", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "dump1090_amd"),
                         "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    syms = subprocess.run(["nm", "-u", str(exe)], capture_output=True, text=True).stdout
    for s in ("modes_gpu_create", "modes_gpu_demod_host", "modes_host_resolve", "modes_host_get_stats"):
        assert s in syms, "the drop-in does not bind " + s


def test_dropin_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    if not os.path.exists(DROPIN):
        pytest.skip("oracle/_ref/dump1090_dropin not built (no reference tree on this machine)")
    p = subprocess.run([DROPIN, "--ifile", os.path.join(ROOT, "tests", "golden", "modes1.bin"), "--raw"], capture_output=True, text=True)
    assert p.returncode == 1 and p.stdout == "" and "no HIP device" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("flags,lines,md5", [
    (["--raw"], 284, "4a81758c8bec5e45ffa8541c5622938a"),
    (["--raw", "--no-fix"], 283, "ac539444a66eb99a7f04affa95c55079"),
    (["--raw", "--aggressive", "--no-crc-check"], 824, "bec25488d6b84e9b0703d164de1cc873"),
    (["--onlyaddr"], 284, "bab0f055e262e216208a5cbbdf63fe24"),
    (["--stats"], 9, "bc3d1c04b24f4989f0fc4a2d1f45abdd"),
    ([], 3202, "0bf2290fa954f1675437e52508ea8aa3"),
])
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
