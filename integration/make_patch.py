#!/usr/bin/env python3
"""Regenerates the two patches from the reference source (needs /root/reference), each as a unified diff with one line of
context, edits anchored on the statements they replace:

  dump1090_gfx950.patch           four edits of dump1090.c: the include, the init call, the two hot-path calls of the main
                                  loop (dump1090.c:2974, :2986).  One 256 KiB buffer per GPU call, the reference's reader.
  dump1090_gfx950_batched.patch   the same four plus a fifth: the file reader called at dump1090.c:524 becomes
                                  modesGpuReadFile() (integration/modes_dropin.c) - K buffers per hand-off.  Live RTL-SDR
                                  input is untouched.

A patch is what a maintainer applies (patch -p1 in the reference's directory); this script only exists so that the patches
can be re-derived and reviewed."""
import difflib
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(REF, "dump1090.c")).read().split("\n")


def make(batched):
    out = list(src)

    def only(pred, what):
        hits = [i for i, ln in enumerate(out) if pred(ln)]
        assert len(hits) == 1, (what, hits)
        return hits[0]

    if batched:
        i = only(lambda ln: ln.strip() == "readDataFromFile();", "readDataFromFile call")
        out[i] = out[i].replace("readDataFromFile();", "modesGpuReadFile();   /* K buffers per hand-off (modes_dropin.c) */")
        # the reader is defined before main(): the include moves in front of it
        i = only(lambda ln: ln.startswith("void *readerThreadEntryPoint(void *arg) {"), "readerThreadEntryPoint")
        # (useModesMessage(), which modes_dropin.c calls, is declared at dump1090.c:276)
        out[i:i] = ['#include "modes_dropin.c"   /* the gfx950 path: modesInitGpu, modesGpuDemod, modesGpuResolve, modesGpuReadFile */', ""]
    else:
        i = only(lambda ln: ln.startswith("int main(int argc, char **argv) {"), "main")
        out[i:i] = ['#include "modes_dropin.c"   /* the gfx950 path: modesInitGpu, modesGpuDemod, modesGpuResolve */', ""]
    i = only(lambda ln: ln.strip() == "modesInit();", "modesInit call")
    out.insert(i + 1, "    if (!Modes.net_only) modesInitGpu();   /* --net-only never demodulates: no GPU needed */")
    i = only(lambda ln: ln.strip() == "computeMagnitudeVector();", "computeMagnitudeVector call")
    out[i] = out[i].replace("computeMagnitudeVector();", "modesGpuDemod();")
    i = only(lambda ln: ln.strip() == "detectModeS(Modes.magnitude, Modes.data_len/2);", "detectModeS call")
    out[i] = out[i].replace("detectModeS(Modes.magnitude, Modes.data_len/2);", "modesGpuResolve();")
    diff = difflib.unified_diff(src, out, "a/dump1090.c", "b/dump1090.c", n=1, lineterm="")
    name = "dump1090_gfx950_batched.patch" if batched else "dump1090_gfx950.patch"
    with open(os.path.join(HERE, name), "w") as f:
        f.write("\n".join(diff) + "\n")
    print("==", name)
    print(open(os.path.join(HERE, name)).read())


make(False)
make(True)
