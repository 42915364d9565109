#!/usr/bin/env python3
"""Regenerates integration/dump1090_gfx950.patch from the reference source (needs /root/reference):
four edits of dump1090.c, anchored on the statements they replace, written as a unified diff with one line of
context.  The patch is what a maintainer applies (patch -p1 in the reference's directory); this script only
exists so that the patch can be re-derived and reviewed."""
import difflib
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(REF, "dump1090.c")).read().split("\n")
out = list(src)


def only(pred, what):
    hits = [i for i, ln in enumerate(out) if pred(ln)]
    assert len(hits) == 1, (what, hits)
    return hits[0]


i = only(lambda ln: ln.startswith("int main(int argc, char **argv) {"), "main")
out[i:i] = ['#include "modes_dropin.c"   /* the gfx950 path: modesInitGpu, modesGpuDemod, modesGpuResolve */', ""]
i = only(lambda ln: ln.strip() == "modesInit();", "modesInit call")
out.insert(i + 1, "    if (!Modes.net_only) modesInitGpu();   /* --net-only never demodulates: no GPU needed */")
i = only(lambda ln: ln.strip() == "computeMagnitudeVector();", "computeMagnitudeVector call")
out[i] = out[i].replace("computeMagnitudeVector();", "modesGpuDemod();")
i = only(lambda ln: ln.strip() == "detectModeS(Modes.magnitude, Modes.data_len/2);", "detectModeS call")
out[i] = out[i].replace("detectModeS(Modes.magnitude, Modes.data_len/2);", "modesGpuResolve();")
diff = difflib.unified_diff(src, out, "a/dump1090.c", "b/dump1090.c", n=1, lineterm="")
with open(os.path.join(HERE, "dump1090_gfx950.patch"), "w") as f:
    f.write("\n".join(diff) + "\n")
print(open(os.path.join(HERE, "dump1090_gfx950.patch")).read())
