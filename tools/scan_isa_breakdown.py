#!/usr/bin/env python3
"""Where the scan kernel's instructions come from: compiles modes_gfx950.hip with line tables,
takes the two unrolled halves of scan_run<false>'s chunk loop out of the ISA and attributes every instruction to the source
construct its debug location (innermost line + inlining chain) belongs to.

    python tools/scan_isa_breakdown.py [VALU wave-instructions per launch of 1 GiB, default: profiles/r08's SQ_INSTS_VALU]

Per chunk (512 samples, one wavefront iteration) = the static count of the always-executed code of one half; what the level pass
(scan_beta: once per ~8.4 chunks) and the divergent push add on top is the counter's total minus that."""
import os, re, subprocess, sys, tempfile
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "dump1090_amd", "csrc", "modes_gfx950.hip")
CORE = os.path.join(ROOT, "dump1090_amd", "csrc", "modes_core.h")


def line_of(path, needle, nth=0):
    hits = [i + 1 for i, l in enumerate(open(path)) if needle in l]
    return hits[nth]


def span_of(path, start_needle):
    """(first, last) line of the brace-balanced block that starts at the line containing start_needle"""
    lines = open(path).read().split("\n")
    a = next(i for i, l in enumerate(lines) if start_needle in l)
    depth, seen = 0, False
    for i in range(a, len(lines)):
        depth += lines[i].count("{") - lines[i].count("}")
        seen = seen or "{" in lines[i]
        if seen and depth == 0:
            return a + 1, i + 1
    raise ValueError(start_needle)


hip = {
    "powers (power16_scan: xor, and, v_dot4_i32_i8 clamp, v_perm)": span_of(SRC, "__device__ __forceinline__ uint4 power16_scan("),
    "level pass (scan_beta)": span_of(SRC, "__device__ __forceinline__ void scan_beta("),
    "push into the queue (push_entry, ranks)": span_of(SRC, "auto push_entry = [&]"),
}
core = {"ordering relations (modes_order8_swar + packed helpers)": span_of(CORE, "MODES_HD void modes_order8_swar(")}
pk_helpers = (line_of(CORE, "MODES_HD uint32_t pk_max("), line_of(CORE, "MODES_HD uint32_t pk_shr1("))
half_sites = (line_of(SRC, "step(sx, wr0, rd0a, rd0b, c0 + k);"), line_of(SRC, "step(sy, wr1, rd1a, rd1b, c0 + k + 1);"))
pow_sites = (line_of(SRC, "const uint4 sx = power16_scan(x);"), line_of(SRC, "const uint4 sy = power16_scan(y);"))
load_sites = (line_of(SRC, "x = load_at(off);"), line_of(SRC, "y = load_at(off + kChunkBytes);"))
loop_line = line_of(SRC, "for (; k + 2 <= nk; k += 2, off += 2 * kChunkBytes) {")
unguarded_site = line_of(SRC, "scan_run<false>(P, run, lane, ring_all[wave], queue_all[wave]);")

tmp = tempfile.mkdtemp(prefix="scan_isa_")
asm = os.path.join(tmp, "k.s")
subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-gline-tables-only", "-I" + os.path.join(ROOT, "include"),
                "-S", "--cuda-device-only", "-o", asm, SRC], check=True, stderr=subprocess.DEVNULL)
body, on = [], False
for l in open(asm):
    if re.match(r"^_ZN\S*scan_kernel\S*:", l):
        on = True
    if on:
        body.append(l.rstrip("\n"))
        if "s_endpgm" in l:
            break

loc_re = re.compile(r"\.loc\s+(\d+)\s+(\d+)\s+\d+.*?;\s*(.*)$")
chain_re = re.compile(r"([\w./-]+):(\d+):\d+")
cat_count = {h: Counter() for h in (0, 1)}
kinds = defaultdict(Counter)
cur = None
for l in body:
    m = loc_re.search(l)
    if m:
        chain = [(os.path.basename(f), int(n)) for f, n in chain_re.findall(m.group(3))]
        cur = chain
        continue
    ins = l.strip()
    if not ins or ins.startswith((";", ".", "_Z")) or ins.endswith(":") or cur is None:
        continue
    op = ins.split()[0]
    if not re.match(r"^(v_|s_|ds_|buffer_|global_)", op):
        continue
    lines_hip = [n for f, n in cur if f == "modes_gfx950.hip"]
    if unguarded_site not in lines_hip:
        continue
    half = None
    for h in (0, 1):
        if half_sites[h] in lines_hip or pow_sites[h] in lines_hip or load_sites[h] in lines_hip:
            half = h
    if half is None:
        if loop_line in lines_hip and not any(s in lines_hip for s in ()):      # loop control is attributed to the for statement
            half = 0 if True else None
            cat = "loop control (counters, prefetch offsets, branch)"
        else:
            continue
    else:
        cat = None
    inner_f, inner_n = cur[0]
    if cat is None:
        if inner_f == "modes_core.h" and (core[next(iter(core))][0] <= inner_n <= core[next(iter(core))][1] or pk_helpers[0] <= inner_n <= pk_helpers[1]):
            cat = next(iter(core))
            if any(hip["level pass (scan_beta)"][0] <= n <= hip["level pass (scan_beta)"][1] for n in lines_hip):
                cat = "level pass (scan_beta)"
        else:
            cat = None
            for name, (a, b) in hip.items():
                if any(a <= n <= b for n in lines_hip):
                    cat = name
                    if name.startswith("level"):
                        break
            if cat is None:
                cat = "exchange through the LDS ring, survivor test, ballot, queue bookkeeping (step)"
    unit = "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM"
    cat_count[half][(cat, unit)] += 1
    kinds[cat][op] += 1

total_valu = float(sys.argv[1]) if len(sys.argv) > 1 else None
if total_valu is None:
    try:
        for l in open(os.path.join(ROOT, "profiles", "r08", "summary.txt")):
            if "SQ_INSTS_VALU" in l and total_valu is None:
                total_valu = float(l.split("mean")[1].split()[0])
    except OSError:
        pass
chunks = (1 << 30) / 1024
print("scan_kernel, scan_run<false>: the chunk loop's two unrolled halves (static instruction counts from the ISA, per half = per chunk)\n")
cats = sorted({c for h in cat_count for (c, _) in cat_count[h]})
print("%-86s %6s %6s %6s %6s" % ("source construct", "VALU", "SALU", "LDS", "VMEM"))
always = 0.0
for c in cats:
    row = [(cat_count[0][(c, u)] + cat_count[1][(c, u)]) / 2.0 for u in ("VALU", "SALU", "LDS", "VMEM")]
    print("%-86s %6.1f %6.1f %6.1f %6.1f" % (c, *row))
    if not c.startswith(("level pass", "push")):
        always += row[0]
print("\nalways executed (every chunk, every lane): %.1f VALU per chunk" % always)
push = (cat_count[0][("push into the queue (push_entry, ranks)", "VALU")] + cat_count[1][("push into the queue (push_entry, ranks)", "VALU")]) / 2.0
print("the push (executed once per chunk under the mask of the ~11 %% of lanes that own an ordering survivor): %.1f VALU per chunk" % push)
if total_valu:
    per_chunk = total_valu / chunks
    print("SQ_INSTS_VALU %.4g per 1 GiB launch = %.1f per chunk -> the level pass (static %.0f VALU, a per-survivor loop inside; once per ~8.4 chunks) "
          "and the rare paths: %.1f per chunk" % (total_valu, per_chunk, (cat_count[0][("level pass (scan_beta)", "VALU")] + cat_count[1][("level pass (scan_beta)", "VALU")]) / 2.0,
                                                  per_chunk - always - push))
print("\nthe instructions behind each construct (both halves):")
for c in cats:
    print("  %s\n      %s" % (c, ", ".join("%s x%d" % (o, n) for o, n in kinds[c].most_common(40))))
