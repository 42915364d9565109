"""The C-ABI libraries load and export every symbol include/*.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from dump1090_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(modes_[a-z0-9_]+)\s*\(", text)) - {"modes_sink_fn"})


def test_host_library_exports_header():
    names = declared("modes_host.h")
    assert sorted(N.HOST_SYMBOLS) == names
    L = N.host_lib()
    for n in names:
        assert getattr(L, n)


def test_gpu_library_exports_header():
    names = declared("modes_gfx950.h")
    assert sorted(N.GPU_SYMBOLS) == names
    assert os.path.exists(N.GPU_LIB), "libmodes_gfx950.so must be built in-tree"
    L = N.gpu_lib()
    for n in names:
        assert getattr(L, n)
    assert L.modes_gpu_abi_version() == 5


def test_gather_library_exports_header():
    """libmodes_gather.so (the record gather over RCCL a C host binds): built in-tree, loads (it links librccl, no GPU
    needed for that) and exports what include/modes_gather.h declares."""
    names = declared("modes_gather.h")
    assert sorted(N.GATHER_SYMBOLS) == names
    assert os.path.exists(N.GATHER_LIB), "libmodes_gather.so must be built in-tree"
    L = N.gather_lib()
    for n in names:
        assert getattr(L, n)
    assert L.modes_gather_abi_version() == 2
    assert C.sizeof(N.GatherConfig) == 24 and C.sizeof(N.GatherStats) == 56
    # without a device the create fails loudly, with a text
    import torch
    if not torch.cuda.is_available():
        ident = (C.c_ubyte * 128)()
        h = C.c_void_p()
        cfg = N.GatherConfig(0, 0, 1, 1024, 0, 0)
        assert L.modes_gather_create(C.byref(cfg), ident, C.byref(h)) < 0 and not h.value
        assert L.modes_gather_last_error(None)


def test_struct_layouts():
    assert C.sizeof(N.Attempt) == 28 and C.sizeof(N.Record) == 64
    assert N.RECORD_DTYPE.itemsize == 64 and N.RECORD_DTYPE.fields["att"][1] == 8
    assert C.sizeof(N.GpuConfig) == 48 and C.sizeof(N.Span) == 40 and C.sizeof(N.GpuResult) == 64


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dump1090_amd import Demodulator, ModesError
    with pytest.raises(ModesError, match="no HIP device"):
        Demodulator()


def test_no_kernel_of_the_shipped_library_spills():
    """Every kernel of dump1090_amd/libmodes_gfx950.so has private_segment_fixed_size == 0 in its gfx950 code object's
    metadata (tools/kernel_resources.py reads the AMDGPU notes of the very file that ships): select_kernel is held to
    64 VGPRs for two 16-wave workgroups per CU and used to pay for it with 52 bytes of scratch per lane."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    if not os.path.exists(kr.READELF):
        pytest.skip("no llvm-readelf")
    rows = kr.kernel_resources(N.GPU_LIB)
    names = {r["name"] for r in rows}
    assert {"scan_kernel", "demod_kernel", "select_kernel", "record_kernel", "finalize_kernel", "finalize2_kernel"} <= names, names
    spills = {r["name"]: r["scratch"] for r in rows if r["scratch"]}
    assert not spills, "kernels with a private segment (bytes per lane): %s" % spills
    by = {r["name"]: r for r in rows}
    assert by["select_kernel"]["vgpr"] <= 64 and by["scan_kernel"]["vgpr"] <= 64      # the occupancy the launch geometry counts on
