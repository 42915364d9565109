"""Builds tests/native/libcore_shim.so (clang++ from the ROCm LLVM, host only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libcore_shim.so")
SRC = os.path.join(HERE, "core_shim.cpp")
CORE = os.path.join(ROOT, "dump1090_amd", "csrc", "modes_core.h")


def clangxx():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++", "clang++"):
        if os.path.sep not in c or os.path.exists(c):
            return c
    raise RuntimeError("no clang++")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(SRC), os.path.getmtime(CORE)):
        return OUT
    subprocess.run([clangxx(), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall",
                    "-I", os.path.join(ROOT, "dump1090_amd", "csrc"), "-o", OUT, SRC], check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))

