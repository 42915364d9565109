#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel in a gfx950 code object.

    python tools/kernel_resources.py [dump1090_amd/libmodes_gfx950.so]     (default: the shipped library)

Reads the clang offload bundle out of the shared library, takes the gfx950 ELF and prints the AMDGPU metadata notes
(llvm-readelf --notes): one row per kernel.  tests/test_abi.py asserts that no kernel of the shipped library has a
private segment (scratch): a spill in a kernel that is held to a register budget is a bug here, not a tuning choice.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_object(so_path, arch="gfx950"):
    """bytes of the `arch` ELF inside the fat binary of so_path"""
    data = open(so_path, "rb").read()
    at = data.find(MAGIC)
    if at < 0:
        raise ValueError("%s: no offload bundle" % so_path)
    n = struct.unpack_from("<Q", data, at + 24)[0]
    off = at + 32
    for _ in range(n):
        o, size, tlen = struct.unpack_from("<QQQ", data, off)
        off += 24
        triple = data[off:off + tlen].decode()
        off += tlen
        if arch in triple:
            return data[at + o:at + o + size]
    raise ValueError("%s: no %s code object" % (so_path, arch))


def kernel_resources(so_path, arch="gfx950"):
    """[{name, vgpr, agpr, sgpr, lds, scratch, max_flat_workgroup_size}] for every kernel of the library"""
    with tempfile.NamedTemporaryFile(suffix=".elf", dir="/tmp") as f:
        f.write(code_object(so_path, arch))
        f.flush()
        notes = subprocess.run([READELF, "--notes", f.name], check=True, stdout=subprocess.PIPE, text=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", "\n" + notes)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk)
        name = get("name")
        if not name or not get("vgpr_count"):
            continue
        short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name.group(1))
        short = re.match(r"[A-Za-z0-9_]+?kernel", short).group(0) if re.match(r"[A-Za-z0-9_]+?kernel", short) else name.group(1)
        out.append({"name": short, "symbol": name.group(1), "vgpr": int(get("vgpr_count").group(1)), "agpr": int(get("agpr_count").group(1)),
                    "sgpr": int(get("sgpr_count").group(1)), "lds": int(get("group_segment_fixed_size").group(1)),
                    "scratch": int(get("private_segment_fixed_size").group(1)),
                    "max_flat_workgroup_size": int(get("max_flat_workgroup_size").group(1))})
    return out


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "dump1090_amd", "libmodes_gfx950.so")
    rows = kernel_resources(so)
    print("%-28s %5s %5s %5s %7s %8s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "LDS", "scratch", "wg"))
    for r in rows:
        print("%-28s %5d %5d %5d %7d %8d %6d" % (r["name"], r["vgpr"], r["agpr"], r["sgpr"], r["lds"], r["scratch"], r["max_flat_workgroup_size"]))
    return 1 if any(r["scratch"] for r in rows) else 0


if __name__ == "__main__":
    sys.exit(main())
