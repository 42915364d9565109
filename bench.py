#!/usr/bin/env python3
"""bench.py - IQ Msamples/s demodulated on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of HBM-resident synthetic input: scan kernel + demod
kernel + order kernel + record fetch (+ gather of the device-resident record lists to rank 0 over RCCL when
N > 1) + the sequential host resolve and --raw formatting on rank 0.

HEADLINE (the JSON line's metric/value/roofline): BASELINE.json configs[1]: 1 GiB of synthetic uint8 I/Q in
2 Msps file format (sigma = 3 integer noise, tests/synth.py), --no-fix, PER GPU (weak scaling: the stream is
sharded into per-GPU buffer ranges, each rank's shard is 1 GiB).

SECOND LEG (object "frames" in the same line): BASELINE.json configs[2] at N = 1 / configs[3] at N = 8: 8 GiB
per GPU of the same noise with DF11/DF17 frames at known offsets (about one per 65,536 samples, 10 % with a
flipped bit, some on the buffer seams), --fix.  The record lists are non-empty here: at N > 1 the RCCL gather
carries real payload, and rank 0 checks the gathered listing against the analytic expectation (every testable
frame, in stream order) before the leg's numbers are reported.  `--workload frames` makes it the only leg.

THIRD LEG (object "lowsnr"): BASELINE.json configs[4]: 1 GiB per GPU of the same noise with weak frames (amplitude 8-15 LSB,
20-40 % inter-sample leak, 5 % two-bit errors, one per 16,384 samples), --aggressive: the phase-corrected retry and the
two-bit repair search of the demod kernel are on the path here.

FOURTH LEG (object "frames_strong"): STRONG scaling - the SAME 64 GiB stream (configs[3]'s: 524,287 frames) at every N, each
rank holding 1/N of its buffers ("also run at 1/2/4 GPUs on the same stream").  At N = 8 this is the
frames leg's own workload and its numbers are reused.

Every listing rank 0 ends up with is checked before a leg's numbers are printed: byte for byte (md5) against what the
compiled reference printed for the very same stream (tests/golden/config_listings.json, made by
tests/golden/make_listings.py) whenever the stream is one of the default sizes at N = 1, 2, 4, 8; against the analytic
expectation (every testable frame, in stream order, next to nothing else) otherwise.

LAST LEG (object "end_to_end", N = 1): the headline workload again, but starting in pinned HOST memory:
modes_gpu_submit_host (H2D over PCIe + kernels) in rotating contexts - the PCIe-inclusive rate, never `value`.

    python bench.py [--gpus N] [--steps K] [--warmup W]          N > 1: starts its own N ranks (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...        (the same thing)

Before the W warmup steps the headline leg runs `--settle` (240) more untimed steps: the chip's power management
needs ~160 back-to-back steps (35 ms) to reach its sustained clocks (profiles/r07/scan_us_per_launch.txt); the K timed
steps are therefore the sustained rate, which is what a stream of many batches sees.

Prints ONE JSON line on rank 0 - and nothing else on stdout: whatever libraries print there (RCCL's version banner) is
sent to stderr.  `roofline` is for the scan kernel (the only stage that reads every sample): algorithmic bytes = 2 per
sample, duration = HIP events recorded around the kernel on its launch stream inside libmodes_gfx950.so;
`roofline.traffic` = the kernel's HBM read bytes per launch from a `rocprofv3 --kernel-trace --pmc FETCH_SIZE` pass that
bench.py runs itself at N = 1 (a child process, three steps of the headline workload, counters only; the committed pass
of the same kernel sources - `roofline.committed_traffic` - where rocprofv3 is missing); `roofline.measured_ceiling` =
the same loads with no arithmetic behind them (stream_read_kernel), measured in this run.  `cpu_baseline` (N == 1 only)
times the compiled reference (oracle/_ref) - or the C restatement if that binary is absent - on the host: on the whole
headline workload (--raw --no-fix, four passes), and inside the `frames` and `lowsnr` objects on the first GiB of THEIR
streams with THEIR flags (--raw / --raw --aggressive; Msamples/s and msgs/s), SURVEY.md 8d.  Every leg settles before
its timed steps (--settle counts 1 GiB steps; a leg reports its own `settle_steps`).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
KERNEL_SOURCES = ("dump1090_amd/csrc/modes_gfx950.hip", "dump1090_amd/csrc/modes_core.h")


def kernel_source_hash():
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(mib):
    """(HBM bytes per scan launch, note) from the committed rocprofv3 PMC passes (profiles/traffic_latest.json,
    written by tools/profile.sh on this same default workload).  The file is stamped with the hash of the kernel
    sources it was measured on: a different hash means the counters are stale and `traffic` is reported as null."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if mib != 1024 or not os.path.exists(path):
        return None, "no PMC pass for this workload"
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("kernel_source_sha256_16") != kernel_source_hash():
            return None, "stale: PMC pass was taken on kernel sources %s, these are %s" % (
                t.get("kernel_source_sha256_16"), kernel_source_hash())
        global TRACE_AVG_MS
        us = t["scan_kernel"].get("trace_avg_us_timed_region")
        TRACE_AVG_MS = round(us / 1e3, 4) if us else None
        return int(t["scan_kernel"]["hbm_read_bytes_per_launch"]), "rocprofv3 FETCH_SIZE pass %s (%s)" % (
            t.get("tag", "?"), t.get("kernel_source_sha256_16"))
    except (KeyError, ValueError):
        return None, "unreadable"


def committed_demod_traffic(mib):
    """HBM bytes the demodulation kernel reads per launch of the headline workload (the committed PMC pass of these kernel sources), or None."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if mib != 1024 or t.get("kernel_source_sha256_16") != kernel_source_hash():
            return None
        return int(t["demod_kernel"]["hbm_read_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def leg_traffic(kind):
    """Committed PMC pass of a record-bearing leg (profiles/traffic_latest.json["legs"][kind], written by tools/profile.sh <tag>
    lowsnr | frames + tools/merge_traffic.py): {kernel: HBM read bytes per launch}, or None when there is none for these kernel
    sources.  The scan kernel's FETCH_SIZE carries the guide's x 2; the demodulation kernels' the factor measured on their own
    access pattern (profiles/fetch_size_calibration.json)."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            t = json.load(f)["legs"][kind]
        if t.get("kernel_source_sha256_16") != kernel_source_hash():
            return None
        # (the kernels of the leg's steady state: those the kernel trace saw in the timed region - the first calls of a context run the
        #  one-kernel path before the record density is known)
        return {k: int(v["hbm_read_bytes_per_launch"]) for k, v in t.items()
                if isinstance(v, dict) and "hbm_read_bytes_per_launch" in v and v.get("trace_avg_us_timed_region")}
    except (OSError, KeyError, ValueError):
        return None


def fetch_bytes_per_launch(directory, kernel="scan_kernel"):
    """(HBM read bytes per launch, launches) from the counter_collection CSVs rocprofv3 left under `directory`: FETCH_SIZE is in
    KB and counts the 128-byte requests of 16 B-per-lane streaming reads as 64 bytes on gfx950 (the guide's correction: x 2).
    None when the kernel has no FETCH_SIZE row."""
    import csv
    import glob
    vals = []
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if kernel in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        return None
    return int(sum(vals) / len(vals) * 1024 * 2), len(vals)


def live_traffic(mib, timeout_s=240):
    """(HBM bytes per scan launch, note) measured NOW: a child `rocprofv3 --kernel-trace --pmc FETCH_SIZE` run (a counter pass of
    its own, no other trace domain) of three steps of the headline workload on this GPU, FETCH_SIZE (KB) averaged over the
    scan kernel's dispatches and doubled (the guide's gfx950 correction for 16 B-per-lane streaming reads).  None when
    rocprofv3 is not there, when this process already runs under it, or when the pass fails - the caller then falls back on
    the committed pass of the same sources."""
    import shutil
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    out = tempfile.mkdtemp(prefix="modes_fetch_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc", "FETCH_SIZE", "-d", out, "-o", "fetch", "-f", "csv", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--workload", "noise", "--mib", str(mib), "--no-end-to-end", "--no-cpu-baseline", "--no-ceiling",
           "--no-live-traffic", "--settle", "4", "--steps", "3", "--warmup", "1", "--depth", "1", "--streams", "1", "--time-every", "100000"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    try:
        t0 = time.perf_counter()
        child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            if child.wait(timeout=timeout_s) != 0:
                return None
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, 9)                      # the profiler AND the bench under it: its own process group
            child.wait()
            return None
        got = fetch_bytes_per_launch(out)
        if got is None:
            return None
        return got[0], "rocprofv3 --kernel-trace --pmc FETCH_SIZE pass run by this bench.py (%d scan launches, %.0f s; KB x 1024 x 2)" % (
            got[1], time.perf_counter() - t0)
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


TRACE_AVG_MS = None      # the committed kernel trace's average scan launch (same sources, same workload), for comparison


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            return next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "?")
    except OSError:
        return "?"


def cpu_baseline(iq, nbytes_sample, flagset="nofix", passes=4, what="the workload"):
    """Reference single-threaded C path on the host cores, on the first nbytes_sample bytes of the same workload (SURVEY.md 8d:
    `./dump1090 --ifile <file> --raw [flags] > /dev/null`, wall clock, one decode thread).  flagset: oracle.FLAGSETS name -
    "nofix" (configs[1]), "default" (configs[2]/[3], --fix), "aggressive" (configs[4]).  Checker code (oracle/) is used here
    only as the thing being timed.  -> the object of the JSON line (Msamples/s and msgs/s)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    sample = iq[:nbytes_sample].cpu().numpy()
    sample[-480:] = 127
    nsamp = nbytes_sample // 2
    cli = {"nofix": ["--no-fix"], "default": [], "aggressive": ["--aggressive"]}[flagset]
    if orc.have_ref():
        with tempfile.NamedTemporaryFile(suffix=".bin", dir="/tmp") as f:
            sample.tofile(f.name)
            env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME)
            dt, out = 0.0, b""
            for _ in range(passes):
                t0 = time.perf_counter()
                out = subprocess.run([orc.REF_BIN, "--ifile", f.name, "--raw"] + cli, stdout=subprocess.PIPE,
                                     env=env, check=True).stdout
                dt += time.perf_counter() - t0
        kind, lines = "reference", out.count(b"\n")
        desc = "oracle/_ref/dump1090_ref --ifile <first %d MiB of %s> --raw%s, %d pass%s" % (
            nbytes_sample >> 20, what, "".join(" " + c for c in cli), passes, "es" if passes > 1 else "")
        nsamp *= passes
    else:
        passes = 1
        t0 = time.perf_counter()
        msgs, _ = orc.run_stream(sample, **orc.FLAGSETS[flagset])
        dt = time.perf_counter() - t0
        kind, lines = "port", len(msgs)
        desc = "oracle/liboracle.so orc_run_stream on the first %d MiB of %s, flag set %s" % (nbytes_sample >> 20, what, flagset)
    return {"value": round(nsamp / dt / 1e6, 2), "unit": "Msamples/s", "msgs_per_s": round(lines * passes / dt, 1), "cores": 1, "kind": kind,
            "sample": "%s; %.2f s wall, %d messages per pass; 1 decode thread (host: %d x %s)" % (
                desc, dt, lines, os.cpu_count() or 0, host_cpu_model())}


LOWSNR = dict(per=16384, amp=(8, 15), smear=(3, 4, 5, 6), flip1=10, flip2=20, edge_every=61)      # configs[4]'s stream (tests/synth.py)


def golden_listing(kind, seed, nblocks):
    """What the compiled reference printed for this very stream, if it is one of the committed ones
    (tests/golden/config_listings.json, made in the build container by tests/golden/make_listings.py): {lines, md5} or None."""
    path = os.path.join(ROOT, "tests", "golden", "config_listings.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get("%s:%d:%d" % (kind, seed, nblocks))


def build_frames_shard(torch, dev, total_blocks, lo, hi, seed, **kw):
    """Bytes [lo, hi) of the stream tests/synth.py:config3_stream(seed, total_blocks, **kw) in HBM, built from this rank's
    frames only: the noise by the device's generator, the frames added to it on the device (SparseFrameStream.deltas).
    -> (tensor, the stream object: placements and clean frames of this rank's part)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    from dump1090_amd import Demodulator
    st = synth.config3_stream(seed, total_blocks, only_samples=(lo // 2, (hi + 1) // 2), **kw)
    d = Demodulator(device=dev.index)
    iq = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
    d.synth_noise(iq, lo, seed=st.seed, sigma_q16=st.sigma_q16)
    first, delta = st.deltas()
    cols = torch.arange(delta.shape[1], device=dev)[None, :]
    for a in range(0, len(first), 65536):                       # footprints are disjoint: a plain read-modify-write
        idx = torch.from_numpy(first[a:a + 65536]).to(dev)[:, None] + cols
        dl = torch.from_numpy(delta[a:a + 65536]).to(dev)
        keep = (idx >= lo) & (idx < hi)
        at = idx[keep] - lo
        iq[at] = (iq[at].to(torch.int16) + dl[keep]).clamp_(0, 255).to(torch.uint8)
    if hi == st.nbytes:
        d.fill(iq[-480:], 127)
    torch.cuda.synchronize(dev)
    d.close()
    return iq, st


def frames_expectation(st, first_block, nblocks):
    """The lines this rank's buffers must contribute: every frame whose preamble offset lies in one of the rank's
    buffers at a tested offset (j < 131070, dump1090.c:1593), in stream order."""
    import synth
    want = []
    for sample, fb, amp, phase, smear in st.placements:
        g = sample + synth.CARRY
        blk, j = divmod(g, synth.BLOCK_STRIDE)
        if first_block <= blk < first_block + nblocks and j < 131070:
            want.append("*" + st.clean[sample].hex() + ";")
    return want


def check_listing(listing: bytes, expected: list[str], weak: bool = False):
    """Rank 0: the gathered, resolved --raw listing against the analytic expectation.  Raises on failure.
    Strong frames (default): every injected frame comes out (<= 0.5 % may be lost to a neighbour's skip window or a
    noise hit), in stream order, and next to NOTHING else does - a line that is not an expected frame is a noise-born or
    mis-repaired message, a handful per stream (<= 0.05 % + 2).  weak (the low-SNR stream, --aggressive): most frames are
    below the demodulator's reach, so what is MISSING cannot be bounded - but what is found must still be frames of the
    stream (<= 1 % + 2 lines that are not: the two-bit repair invents a valid-looking message now and then) in stream order
    (the exact check is the reference's md5 where a listing is committed: golden_listing)."""
    lines = listing.decode().split()
    listed = set(lines)
    want = set(expected)
    missing = sum(1 for e in expected if e not in listed)
    spurious = sum(1 for ln in lines if ln not in want)
    if weak:
        # (measured: 0-1 invented lines per 2,500 - tests/test_bench_helpers.py; the bound leaves room for a noise-born message
        # or a mis-repair per hundred, not for a second listing mixed into the first)
        assert len(lines) > 0 and spurious <= len(lines) // 100 + 2, "%d lines, %d of them no frame of the stream" % (len(lines), spurious)
    else:
        assert missing <= len(expected) // 200, "%d of %d injected frames are not in the listing" % (missing, len(expected))
        assert 0.99 * len(expected) <= len(lines) <= 1.02 * len(expected) + 64, "%d lines for %d frames" % (len(lines), len(expected))
        assert spurious <= len(expected) // 2000 + 2, "%d of %d lines are no frame of the stream" % (spurious, len(lines))
    # stream order: walking the expected frames in the order of their offsets, each one is found at or after the
    # line of its predecessor.  Frames whose bytes occur more than once in the stream stay out of the walk (a 56-bit
    # DF11 has 27 free bits: among the 131,000 of an 8 GiB low-SNR stream ~60 pairs collide, and when the first of a
    # pair is too weak to decode its twin's line, far ahead, would be taken for it).
    from bisect import bisect_left
    from collections import Counter
    twice = {e for e, c in Counter(expected).items() if c > 1}
    where = {}
    for i, ln in enumerate(lines):
        where.setdefault(ln, []).append(i)
    at = out_of_order = 0
    for e in expected:
        occ = where.get(e)
        if not occ or e in twice:
            continue
        k = bisect_left(occ, at)
        if k == len(occ):
            out_of_order += 1
        else:
            at = occ[k]
    assert out_of_order <= (len(lines) // 100 if weak else 0), \
        "%d expected frames appear before their predecessors: the listing is not in stream order" % out_of_order
    return {"lines": len(lines), "expected_frames": len(expected), "missing": missing, "spurious": spurious,
            "md5": hashlib.md5(listing).hexdigest()}


def launcher_command(argv, gpus, port=None):
    """`python bench.py --gpus N` without a launcher environment starts its own N ranks: the command the driver documents
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...).
    torch.distributed.run ends with a non-zero status when any rank does."""
    if port is None:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def host_cpus():
    """What the host side of this run can use: logical CPUs, the affinity mask, and the container's CPU-time quota in CPUs
    (cgroup v2 cpu.max / v1 cfs quota; None = no limit).  The resolver's worker threads share the quota with the launch threads:
    on round 5's boxes 256 logical CPUs sit behind a quota of 16 (profiles/r08/rank_resolve_8ranks.txt)."""
    out = {"logical": os.cpu_count(), "affinity": None, "quota": None}
    try:
        out["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            out["quota"] = round(int(q) / int(p), 2)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                out["quota"] = round(q / p, 2)
        except (OSError, ValueError):
            pass
    try:                                             # what the host library sizes its pools by (modes_host_cpu_budget)
        from dump1090_amd import _native as N
        out["budget"] = int(N.host_lib().modes_host_cpu_budget())
    except Exception:                                # noqa: BLE001
        out["budget"] = None
    return out


def auto_resolve_threads(world, ranks_resolve):
    """Threads of a rank's resolve when --resolve-threads is not given: what the process may really run at once - the smallest of
    online CPUs, affinity and cgroup quota (modes_host_cpu_budget; round 5's min(32, cores / 4) asked a 16-CPU container for 32) - less
    one per rank for the launching threads that poll for their kernels, shared out among the ranks when every rank resolves."""
    from dump1090_amd import _native as N
    spare = max(1, N.host_lib().modes_host_cpu_budget() - world)
    return max(1, min(32, spare // world if ranks_resolve else spare))


def n1_reference(leg):
    """The committed ONE-GPU rate of a leg (profiles/n1_reference.json, written from this round's `python bench.py` on the lease) - what an
    N > 1 line's efficiency is priced against inside the record; the driver computes its own from its own N = 1 run."""
    if leg is None:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "n1_reference.json")) as f:
            return json.load(f).get(leg)
    except (OSError, ValueError):
        return None


def visible_gpus():
    """GPUs this process can use: hipGetDeviceCount through torch (it honours the *_VISIBLE_DEVICES lists and what the container
    may open).  Called by the parent of a self-launched N-rank run only - the ranks are separate processes."""
    import torch
    return torch.cuda.device_count()


def ipc_probe(dist, torch, dev, rank, world, seconds=300.0):
    """The first transfer between two ranks' devices is where a wrong IPC mode or a missing peer-to-peer path shows - as a hang.
    A 64-byte ring (rank r -> r + 1) right behind init_process_group, watched: when it does not complete in `seconds` (generous:
    on a fresh box RCCL's first communicator alone has taken 60-435 s while the image pages in, and the first send / recv between
    two ranks sets up its channels on top of that - a slow start must not be taken for a hang) the rank
    says which HSA_ENABLE_IPC_MODE_LEGACY it ran with, leaves $MODES_PROBE_MARK for a self-launched parent (which then starts the
    job over with the other value) and ends the process - a wrong guess costs seconds, not the lease."""
    a = torch.full((64,), 0x5a, dtype=torch.uint8, device=dev)
    b = torch.zeros(64, dtype=torch.uint8, device=dev)
    ops = [dist.P2POp(dist.isend, a, (rank + 1) % world), dist.P2POp(dist.irecv, b, (rank - 1) % world)]
    works = dist.batch_isend_irecv(ops)
    t0 = time.perf_counter()
    while not all(w.is_completed() for w in works):
        if time.perf_counter() - t0 > seconds:
            print("bench.py rank %d of %d: the first 64-byte isend/irecv ring did not complete in %.0f s (HSA_ENABLE_IPC_MODE_LEGACY=%s)" % (
                rank, world, seconds, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset")), file=sys.stderr, flush=True)
            # one machine-readable line for whoever keeps the log (the driver, tools/first_contact.sh): the failure path says what
            # the success path's `rccl_start` object says
            print(json.dumps({"rccl_start": dict(RCCL_START, failed="probe", rank=rank, world=world, probe_s=round(time.perf_counter() - t0, 1))}),
                  file=sys.stderr, flush=True)
            mark = os.environ.get("MODES_PROBE_MARK")
            if mark:
                open(mark, "w").close()
            os._exit(75)
        time.sleep(0.002)
    torch.cuda.synchronize(dev)
    assert bool((b == 0x5a).all()), "the probe ring delivered other bytes than were sent"
    return time.perf_counter() - t0


# How the process group came up (rank 0's view; in the JSON line as `rccl_start`, and on stderr when the start fails)
RCCL_START = {"ipc_mode": None, "restarts": 0, "init_s": None, "probe_s": None}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=5,
                    help="timed regions of the headline leg, back to back behind one warm-up, each exactly --steps steps between a barrier + "
                         "device sync on both sides: `value` / `ms_per_step` are the MEDIAN region, `value_min` / `value_max` and "
                         "`ms_per_step_regions` say how far one region is from another (a 20-step region is 4 ms)")
    ap.add_argument("--workload", default="all", choices=("all", "noise", "frames", "lowsnr", "strong"),
                    help="all (default): every leg, the noise leg's numbers in the headline fields; one name: that leg only (a leg "
                         "other than noise then fills the headline fields)")
    ap.add_argument("--mib", type=int, default=1024, help="MiB of I/Q per GPU of the noise leg (default: the 1 GiB workload)")
    ap.add_argument("--frames-mib", type=int, default=8192, help="MiB of I/Q per GPU of the frames leg (default: 8 GiB)")
    ap.add_argument("--frames-steps", type=int, default=40,
                    help="timed steps of the frames leg (8 GiB each: with 6 the fill and drain of the pipeline were a fifth of the time)")
    ap.add_argument("--lowsnr-mib", type=int, default=1024, help="MiB of I/Q per GPU of the low-SNR leg (configs[4])")
    ap.add_argument("--lowsnr-steps", type=int, default=200,
                    help="timed steps of the low-SNR leg (1 GiB each: the fill and drain of the four-deep pipeline is ~0.3 ms per timed region - "
                         "3 %% of 40 steps, 0.6 %% of 200)")
    ap.add_argument("--frames-total-mib", type=int, default=65536,
                    help="MiB of the strong-scaling leg's stream, the same at every N (default: configs[3]'s 64 GiB); 0 = no such leg")
    ap.add_argument("--strong-steps", type=int, default=8, help="timed steps of the strong-scaling leg (64 GiB each)")
    ap.add_argument("--cpu-mib", type=int, default=1024, help="MiB of the workload timed on the CPU baseline")
    ap.add_argument("--run-chunks", type=int, default=0)
    ap.add_argument("--demod-variant", type=int, default=0, help="modes_gpu_config.demod_variant (include/modes_gfx950.h)")
    ap.add_argument("--direct-records", type=int, default=0,
                    help="modes_gpu_config.direct_records: lists of up to this many records are written to the host's pinned copy by the "
                         "kernels themselves (0 = the library's default, 4096); longer lists follow as one device-to-host copy")
    ap.add_argument("--depth", type=int, default=0, help="detect calls in flight (contexts used in rotation); default 4, and 6 when "
                                                        "the record lists are gathered (N > 1): that pipeline has two more stages")
    ap.add_argument("--settle", type=int, default=240,
                    help="extra untimed steps before the W warmup steps: the chip's power management needs ~160 "
                         "back-to-back steps (35 ms) to settle - the scan kernel runs 0.20, 0.27, 0.24, 0.20, 0.19, 0.185 ms at "
                         "launches 1, 15, 30, 80, 140, 200 of a sustained run (profiles/r07/scan_us_per_launch.txt)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="modes_gpu_config.overlap: 0 = scan, demod and order kernels in order on one stream; 2 = the "
                         "order kernel (a few microseconds, no LDS) runs on the context's own stream next to the following "
                         "step's scan; 1 = the demod kernel too")
    ap.add_argument("--time-every", type=int, default=8,
                    help="one call in this many carries HIP timing events around its kernels (they cost ~9 us of idle GPU per "
                         "kernel boundary); 1 = every call")
    ap.add_argument("--resolve-threads", type=int, default=0,
                    help="threads of rank 0's resolve (modes_host_resolve_raw_mt: exact, speculative pieces confirmed in order); "
                         "0 = the process's CPU budget (affinity, cgroup quota) less one launching thread per rank, at most 32")
    ap.add_argument("--streams", type=int, default=2,
                    help="launch streams of the throughput region.  2 (default): the scan kernel of call i+1 does not queue behind the "
                         "demod / finalize kernels of call i.  Kernel durations are measured in a second timed region on ONE stream, "
                         "where every kernel runs alone (`one_launch_stream`).  1: one region, one stream")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to smoke-test "
                                                      "the N > 1 control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--resolve-on", default="root", choices=("root", "ranks"),
                    help="N > 1 (or --force-gather): root (default) = the record lists travel to rank 0, which resolves them all; ranks = "
                         "every rank resolves its own records from a guessed whitelist, the ranks confirm each other and only the TEXT "
                         "travels (dump1090_amd/distributed.py RankResolve; DESIGN.md 5.4)")
    ap.add_argument("--force-gather", action="store_true",
                    help="with one rank: run the N > 1 code path anyway (process group of one, device output buffers, count "
                         "all_gather, the list sent to itself through isend / irecv) - exercises the RCCL calls on a one-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--leg-streams", type=int, default=0, help="launch streams of the frames / lowsnr / strong legs; 0 (default): two for legs "
                    "whose calls are at most 2 GiB, one for bigger calls (the kernels are timed in a one-stream region of their own either way)")
    ap.add_argument("--call-blocks", type=int, default=32767, help="buffers per GPU call of the frames / strong legs (at most 32767 = "
                    "8 GiB - 256 KiB; the same number of calls on every rank)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC pass only: no child "
                    "rocprofv3 --pmc FETCH_SIZE run (N = 1, ~40 s)")
    ap.add_argument("--no-ceiling", action="store_true", help="skip roofline.measured_ceiling (the read-only streaming kernel)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one (one process per GPU; rank 0 of the children prints the line).  RCCL wants a device per
        # rank: say so here, in one line and at once, instead of N ranks failing somewhere inside the rendezvous
        if args.backend == "nccl":
            have = visible_gpus()
            if have < args.gpus:
                sys.exit("bench.py --gpus %d: %d GPU(s) visible on this box (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); RCCL needs one "
                         "device per rank (--backend gloo runs the ranks' control flow on fewer)" % (args.gpus, have))
        # one restart: a rank whose first transfer over the new process group does not complete (ipc_probe) leaves a mark; HSA reads
        # HSA_ENABLE_IPC_MODE_LEGACY when a process starts, so the other value needs new ranks
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        mark = os.path.join(tempfile.gettempdir(), "modes_ipc_probe_%d" % os.getpid())
        rc = subprocess.call(launcher_command(sys.argv[1:], args.gpus), env=dict(os.environ, MODES_PROBE_MARK=mark))
        if rc != 0 and os.path.exists(mark):
            os.remove(mark)
            other = "1" if os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" else "0"
            print("bench.py: the first transfer between the ranks did not complete with HSA_ENABLE_IPC_MODE_LEGACY=%s; starting over with %s" % (
                os.environ["HSA_ENABLE_IPC_MODE_LEGACY"], other), file=sys.stderr, flush=True)
            rc = subprocess.call(launcher_command(sys.argv[1:], args.gpus), env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=other, MODES_IPC_RETRIED="1"))
        if rc != 0:
            print(json.dumps({"rccl_start": {"failed": "launcher status %d" % rc, "world": args.gpus,
                                             "ipc_mode_tried_first": os.environ["HSA_ENABLE_IPC_MODE_LEGACY"]}}), file=sys.stderr, flush=True)
        sys.exit(rc)

    # stdout carries ONE line: the JSON.  Libraries write there too (RCCL prints a five-line version banner with printf when a
    # communicator is made, flushed at exit): keep the real stdout for the line and give everything else stderr as fd 1.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool's hosts (RCCL across processes)
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from dump1090_amd import Demodulator, block_count, shard_blocks, shard_byte_range
    from dump1090_amd.pipeline import run_steps, split_calls

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world                                       # the launcher's word counts
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")     # where the gathered bytes travel
    dist_on = world > 1 or args.force_gather
    if args.depth <= 0:
        args.depth = 6 if dist_on else 4
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        # a rank that stops answering ends the job with an error after five minutes (the default is ten) instead of
        # holding the node (not three: the first RCCL start on a fresh box has taken 60-100 s by itself)
        import datetime
        limit = datetime.timedelta(seconds=300)
        RCCL_START.update(ipc_mode=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset"), restarts=1 if os.environ.get("MODES_IPC_RETRIED") else 0)
        if args.backend == "nccl":
            t_init = time.perf_counter()
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=limit)
            except Exception as e:                              # the rendezvous or the communicator: say it in one line, then fail as before
                print(json.dumps({"rccl_start": dict(RCCL_START, failed="init_process_group: %s" % str(e)[:200], rank=rank, world=world,
                                                     init_s=round(time.perf_counter() - t_init, 1))}), file=sys.stderr, flush=True)
                raise
            RCCL_START["init_s"] = round(time.perf_counter() - t_init, 3)
            RCCL_START["probe_s"] = round(ipc_probe(dist, torch, dev, rank, world), 3)   # (a group of one sends to itself: the same calls on a one-GPU box)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=limit)
    ranks_resolve = dist_on and args.resolve_on == "ranks"
    # host memory for the ranks' whitelist exchanges (three small all_gathers a step), whatever the records' backend is
    # (made whenever lists travel: the strong leg runs in BOTH resolve modes behind each other - one SCALE invocation, both answers)
    ctl_group = dist.new_group(backend="gloo", timeout=limit) if dist_on else None

    def shard(total_bytes):
        """this rank's contiguous buffer range of the whole stream, and the bytes it needs (476-byte carry in front)"""
        nblocks_total = block_count(total_bytes)
        first_block, nblocks = shard_blocks(nblocks_total - 1, world, rank)   # the EOF buffer goes to the last rank
        if rank == world - 1:
            nblocks += 1
        lo, hi = shard_byte_range(first_block, nblocks, total_bytes)
        return first_block, nblocks, lo, hi

    def leg(iq, lo, calls, flags, steps, warm, cap_records, nstreams, timing, time_every=None, regions=1, resolve_on=None):
        """K timed steps over this rank's HBM-resident shard (dump1090_amd/pipeline.py).  timing: one call in --time-every
        carries HIP timing events around its kernels (one launch stream only: the times are then the kernels' own); without:
        no events at all - pure throughput."""
        works = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        resolve_on = resolve_on or args.resolve_on
        ranks_resolve = dist_on and resolve_on == "ranks"          # (this leg's: the strong leg runs once in each mode)

        class NoTiming:                                   # run_steps switches timing per call only on objects that have set_timing
            def __init__(self, d):
                self._d = d
                d.set_timing(False)

            def __getattr__(self, k):
                if k == "set_timing":
                    raise AttributeError(k)
                return getattr(self._d, k)

        def make():
            d = Demodulator(device=local, run_chunks=args.run_chunks, overlap=args.overlap,
                            demod_variant=args.demod_variant, direct_records=args.direct_records, max_records=cap_records if dist_on and not ranks_resolve else 0, **flags)
            return d if timing else NoTiming(d)
        return run_steps(make, iq, lo, calls, flags, steps, warm, args.depth, world=world, rank=rank, dist=dist,
                         coll_device=coll_dev, cap_records=cap_records, streams=works,
                         device_sync=lambda: torch.cuda.synchronize(dev), time_every=max(1, time_every or args.time_every),
                         # (root: rank 0 resolves every rank's records on up to 32 threads; ranks: every rank its own, all at the same
                         #  time - half the hardware threads shared out among them)
                         resolve_threads=args.resolve_threads or auto_resolve_threads(world, ranks_resolve),
                         gather=dist_on, resolve_on=resolve_on, ctl_group=ctl_group, regions=regions)

    def gathered(obj):
        """[obj of rank 0, of rank 1, ...] on rank 0 (None elsewhere)"""
        if world == 1:
            return [obj]
        parts = [None] * world if rank == 0 else None
        dist.gather_object(obj, parts, dst=0)
        return parts

    def comm_facts(res, steps):
        """What a reader needs to trust an N > 1 line: rank 0's view of the exchanges of the timed steps."""
        c = res["comm"]
        version = None
        if args.backend == "nccl":
            try:
                version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:       # noqa: BLE001
                version = "?"
        return {"nranks": world, "backend": "RCCL" if args.backend == "nccl" else args.backend, "version": version,
                "count_all_gathers_per_step": round(c["calls"] / steps, 2), "p2p_ops_per_step": round(c["p2p_ops"] / steps, 2),
                "bytes_gathered_per_step": int(c["bytes"] / steps),
                "gather_ms": round(c["ms"] / max(1, c["calls"]), 4),        # GPU time of one call's two exchanges (communication stream)
                "loopback": (world == 1) or None}

    def frames_leg(kind, seed, total_bytes, flags, steps, kw, cap_records, both_modes=False):
        """A leg over the stream config3_stream(seed, total_bytes / 262144, **kw), sharded over the ranks: K timed steps, the
        last step's gathered listing checked (reference md5 where committed, analytic expectation otherwise)."""
        total_blocks = total_bytes // 262144
        first_block, nblocks, lo, hi = shard(total_bytes)
        iq_f, st = build_frames_shard(torch, dev, total_blocks, lo, hi, seed=seed, **kw)
        mine = frames_expectation(st, first_block, nblocks)
        # one detect call holds at most 8 GiB - 64 KiB of samples: the same number of calls per step on every rank (the
        # collectives pair up)
        max_blocks = total_blocks // world + 1
        calls = split_calls(first_block, nblocks, (max_blocks + args.call_blocks - 1) // args.call_blocks, lo, total_bytes)
        # settle like the headline leg: the chip needs ~35 ms of back-to-back kernels to reach its sustained clocks, and the
        # seconds of host-side stream building in front of this leg restart that transient.  --settle counts
        # steps of the 1 GiB workload; a leg with G GiB per step and GPU gets ceil(settle / G) of its own steps, at least 6.
        gib_per_step = max((hi - lo) / 2 ** 30, 1e-3)
        settle_steps = max(6, int(-(-args.settle // gib_per_step)))
        # Like the headline leg: the THROUGHPUT region runs on --leg-streams launch streams with no event in them (2 since round 5: the
        # scan of call i + 1 fills the tail of call i's demodulation kernels - low SNR 0.2448 -> 0.2370 ms per step, frames 1.8475 ->
        # 1.8335; profiles/r08/leg_streams_ab.txt), and the kernels are timed in a short region of their own on ONE stream (one call in
        # four carries events: every kernel alone, at sustained clocks) - two regions, one listing check.
        # (0 = automatic: two for calls of up to 2 GiB - the next scan has a tail to fill - one for bigger calls, whose kernels are long enough
        #  to leave nothing to fill and whose big record copies only get in each other's way: frames 1.861 / 1.866 ms on one stream, 1.894 / 2.203 on two)
        # (decided from rank-independent numbers: the launch streams set the order in which a rank issues its collectives)
        nls = args.leg_streams if args.leg_streams > 0 else (2 if total_bytes // world // len(calls) <= (2 << 30) else 1)
        res = leg(iq_f, lo, calls, flags, steps, settle_steps, cap_records, nls, nls == 1)
        if nls > 1:
            tsteps = max(4, -(-48 // len(calls)))                # ~48 calls, 12 of them timed
            # (settled like every region: the pause between two regions - the resolver drains, contexts are torn down and made - restarts
            #  the chip's clock transient)
            kt = leg(iq_f, lo, calls, flags, tsteps, settle_steps, cap_records, 1, True, time_every=4)
            res.update(scan_ms=kt["scan_ms"], demod_ms=kt["demod_ms"], order_ms=kt["order_ms"], scan_ms_median=kt["scan_ms_median"],
                       timed_calls=kt["timed_calls"], kernel_steps=tsteps,
                       kernel_timing="a region of %d steps of the same workload on ONE launch stream, HIP events on one call in 4" % tsteps)
        res.update(total=total_bytes, span=hi - lo, per_gpu=total_bytes // world, settle_steps=settle_steps, resolve_on=args.resolve_on if dist_on else None)
        other = None
        if both_modes and dist_on:
            # the same steps over the same resident shard with the OTHER resolve mode (root: the lists travel to rank 0, which resolves them all;
            # ranks: every rank resolves its own, the texts travel): a sub-linear point of the curve is attributed in the record itself
            mode2 = "root" if args.resolve_on == "ranks" else "ranks"
            # (the first pass is the line's; a second pass that raises - on every rank or on one - is reported, not fatal: the ranks
            #  agree on the outcome before anything else is exchanged)
            err2 = None
            try:
                other = leg(iq_f, lo, calls, flags, steps, settle_steps, cap_records, nls, nls == 1, resolve_on=mode2)
            except Exception as e:                               # noqa: BLE001
                other, err2 = None, "%s: %s" % (type(e).__name__, str(e)[:300])
            verdicts = [None] * world
            if world > 1:
                dist.all_gather_object(verdicts, err2)
            else:
                verdicts = [err2]
            if any(v is not None for v in verdicts):
                other = None
                res["other_mode_error"] = {"resolve_on": mode2, "errors": {r: v for r, v in enumerate(verdicts) if v is not None}}
        if other is not None:
            other.update(total=total_bytes, span=hi - lo, per_gpu=total_bytes // world, settle_steps=settle_steps, resolve_on=mode2,
                         scan_ms=res["scan_ms"], demod_ms=res["demod_ms"], order_ms=res["order_ms"], scan_ms_median=res["scan_ms_median"],
                         timed_calls=res["timed_calls"], kernel_timing="the first pass's (the same kernels on the same input)")
        if rank == 0 and world == 1 and not args.no_cpu_baseline and kind != "strong":     # (strong: the frames leg's stream at another seed)
            # the compiled reference on the first GiB of THIS leg's stream, with this leg's flags ("for >= 8 GiB
            # configs ... on the first 1 GiB with the extrapolation stated": the stream is statistically uniform - one frame
            # per `per` samples throughout - so the rate of the first GiB is the rate of the whole)
            nb = min(args.cpu_mib << 20, (hi - lo) // 262144 * 262144)
            res["cpu_baseline"] = cpu_baseline(iq_f, nb, "aggressive" if flags.get("aggressive") else "default", passes=1,
                                               what="this leg's stream (the whole stream has the same frame density)")
        del iq_f
        torch.cuda.empty_cache()
        parts = gathered(mine)
        if rank == 0:
            expected = [e for p in parts for e in p]
            chk = check_listing(res["listing"], expected, weak=(kind == "lowsnr"))
            gold = golden_listing(kind if kind != "strong" else "frames", seed, total_blocks)
            if gold is not None:
                assert (chk["lines"], chk["md5"]) == (gold["lines"], gold["md5"]), \
                    "the listing differs from the reference's: %s vs %s" % (chk, gold)
            chk["equals_reference_md5"] = True if gold is not None else None      # None: no committed listing for this size
            res["check"] = chk
            if other is not None:
                chk2 = check_listing(other["listing"], expected, weak=(kind == "lowsnr"))
                assert (chk2["lines"], chk2["md5"]) == (chk["lines"], chk["md5"]), "resolve_on=%s prints another listing: %s vs %s" % (other["resolve_on"], chk2, chk)
                chk2["equals_reference_md5"] = chk["equals_reference_md5"]
                other["check"] = chk2
        res["other_mode"] = other
        return res

    line = {}
    noise = noise1 = noise_s1 = None
    iq_noise = None
    ceiling = None
    want = lambda name: args.workload in ("all", name)
    if want("noise"):
        per_gpu = args.mib << 20
        total = per_gpu * world
        first_block, nblocks, lo, hi = shard(total)
        gen = Demodulator(device=local, fix=False)
        iq_noise = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
        gen.synth_noise(iq_noise, first_byte=lo, seed=20260922, sigma_q16=941)
        if hi == total:
            gen.fill(iq_noise[-480:], 127)              # oracle-safe tail
        torch.cuda.synchronize(dev)
        calls = split_calls(first_block, nblocks, 1, lo, total)
        # Timed regions over the same resident input:
        #   throughput  K steps, --streams (2) launch streams, no events in the streams: `value`, `ms_per_step`
        #   one stream  K steps, ONE launch stream, no events either: `one_launch_stream` (what a single-stream caller gets)
        #   kernels     max(K, 96) steps on ONE launch stream, timing events on one call in 4: every kernel runs alone, so
        #               the durations are the kernels' own -> `kernel_ms`, `roofline` (>= 24 launches in the average; each
        #               timed call costs the stream a host round trip, so this region's step time means nothing)
        # (--streams 1: one region serves all three; its kernel times come from one call in --time-every.)
        noflags = dict(fix=False, aggressive=False)
        settle = args.settle + args.warmup
        # R regions each (--regions): one region of 20 steps is 4 ms - a single sample; the line carries the median and the spread
        R = max(1, args.regions)
        if args.streams <= 1:
            noise = noise1 = noise_s1 = leg(iq_noise, lo, calls, noflags, args.steps, settle, 1 << 16, 1, True, regions=R)
        else:
            noise1 = leg(iq_noise, lo, calls, noflags, max(args.steps, 32 if R > 1 else 96), settle, 1 << 16, 1, True, time_every=4, regions=R)
            noise_s1 = leg(iq_noise, lo, calls, noflags, args.steps, settle, 1 << 16, 1, False)
            noise = leg(iq_noise, lo, calls, noflags, args.steps, settle, 1 << 16, args.streams, False, regions=R)
        for x in (noise, noise1, noise_s1):
            x.update(total=total, span=hi - lo, per_gpu=per_gpu)
        if not args.no_ceiling:
            # the chip's own read-only streaming rate, right behind the legs (sustained clocks): the scan kernel's loads and
            # launch geometry, no arithmetic - 96 launches back to back, one in 4 timed like the scan kernel's launches are
            off = (-iq_noise.data_ptr()) % 16
            c_avg, c_min, c_bytes = gen.stream_ceiling(iq_noise[off:], launches=96, time_every=4)
            ceiling = {"GB_per_s": round(c_bytes / (c_avg * 1e-3) / 1e9, 1), "ms_per_launch": round(c_avg, 4), "min_ms": round(c_min, 4),
                       "bytes_per_launch": int(c_bytes),
                       "what": "stream_read_kernel: the scan kernel's loads (16 B per lane, nt | sc1, two chunks in flight, runs of 32 "
                               "chunks per wavefront) over the same resident input, nothing else; 24 timed launches of 96 back to back"}
        gen.close()

    frames = lowsnr = strong = None
    if want("frames"):
        fsteps = args.frames_steps if args.workload == "all" else args.steps
        frames = frames_leg("frames", 3 if world == 1 else 4, (args.frames_mib << 20) * world, dict(fix=True, aggressive=False),
                            fsteps, {}, 1 << 17, both_modes=(args.frames_mib << 20) * world == args.frames_total_mib << 20)
    if want("lowsnr"):
        lsteps = args.lowsnr_steps if args.workload == "all" else args.steps
        lowsnr = frames_leg("lowsnr", 5, (args.lowsnr_mib << 20) * world, dict(fix=True, aggressive=True), lsteps, LOWSNR, 1 << 16)
    strong_is_frames = False
    if want("strong") and args.frames_total_mib:
        total_strong = args.frames_total_mib << 20
        if frames is not None and frames["total"] == total_strong and world > 1:
            strong, strong_is_frames = frames, True             # N = 8: the frames leg already is this stream on these shards
        else:
            ssteps = args.strong_steps if args.workload == "all" else args.steps
            strong = frames_leg("strong", 4, total_strong, dict(fix=True, aggressive=False), ssteps, {}, 1 << 17, both_modes=True)

    def leg_summary(leg, name, scaling):
        steps = leg["steps"]
        samples_per_step = leg["total"] // 2
        el = leg["elapsed"]
        d = {"workload": name, "scaling": scaling, "Msamples_per_s": round(samples_per_step * steps / el / 1e6, 1),
             "ms_per_step": round(el / steps * 1e3, 4), "steps": steps, "calls_per_step": leg["calls_per_step"],
             "bytes_per_gpu": leg["per_gpu"],
             "kernel_ms": {"scan": round(leg["scan_ms"], 4), "demod": round(leg["demod_ms"], 4), "order": round(leg["order_ms"], 4),
                           "measured_in": leg.get("kernel_timing", "the timed region")},
             "records_per_step_rank0": int(leg["last"].get("n_records", 0)), "host_ms_per_call": leg.get("host_ms_per_call"),
             "settle_steps": leg.get("settle_steps")}
        if leg.get("cpu_baseline") is not None:
            d["cpu_baseline"] = leg["cpu_baseline"]
        ksum = leg["calls_per_step"] * (leg["scan_ms"] + leg["demod_ms"] + leg["order_ms"])
        d["wall_over_kernels"] = round(d["ms_per_step"] / ksum, 3) if ksum > 0 else None
        # the leg's own roofline: algorithmic bytes = 2 B per sample = this rank's bytes per step; `scan` prices the
        # scan kernel's launches (HIP events), `step` the whole step (every kernel, the fetch and the resolve behind it: the leg's clock)
        if leg["scan_ms"] > 0:
            kind = name.split(":")[0].replace("BASELINE.json ", "")
            tkind = "lowsnr" if "configs[4]" in kind else "frames"
            if scaling != "weak":
                tkind = "strong"                                     # its own pass (tools/profile.sh <tag> strong): 7.1 GiB calls
            tr = leg_traffic(tkind)
            per_launch = leg["call_bytes"]
            a_scan = per_launch / (leg["scan_ms"] * 1e-3) / 1e9
            a_step = leg["per_gpu"] / (d["ms_per_step"] * 1e-3) / 1e9
            d["roofline"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "scan_kernel",
                             "algorithmic_bytes_per_launch": int(per_launch), "algorithmic_bytes_per_step": int(leg["per_gpu"]),
                             "achieved_scan": round(a_scan, 1), "frac_scan": round(a_scan / HBM_PEAK_GBS, 4),
                             "achieved_step": round(a_step, 1), "frac_step": round(a_step / HBM_PEAK_GBS, 4),
                             # HBM bytes per launch of every kernel of a call, from the committed counter pass of THIS workload
                             "traffic": tr, "traffic_per_launch_total": sum(tr.values()) if tr else None,
                             "traffic_source": ("profiles/traffic_latest.json legs.%s (rocprofv3 FETCH_SIZE passes of this workload on these "
                                                "kernel sources; profiles/README.md)" % tkind) if tr else "no committed PMC pass of this workload for these kernel sources"}
        per_rank = gathered({"scan": round(leg["scan_ms"], 4), "demod": round(leg["demod_ms"], 4)})
        if rank == 0:
            d["msgs_per_s"] = round(leg["msgs"] / el, 1)
            d["msgs_per_step"] = int(leg["lines"])
            d["listing_check"] = leg["check"]
            if world > 1:
                d["kernel_ms_per_rank"] = per_rank
                # a sub-linear step is either a slow rank or rank 0's host half: both are in the line
                per_step = lambda r: leg["calls_per_step"] * (r["scan"] + r["demod"])
                d["kernel_ms_per_step_max_rank"] = round(max(per_step(r) for r in per_rank), 4)
                d["kernel_ms_per_step_min_rank"] = round(min(per_step(r) for r in per_rank), 4)
                d["rank0_resolve_ms_per_step"] = (leg.get("host_ms_per_call") or {}).get("resolve_per_step")
            if leg.get("rank_resolve"):
                d["rank_resolve"] = leg["rank_resolve"]     # rank 0's view: its own share of the resolve, the protocol's exchanges
            if dist_on:
                d["rccl"] = comm_facts(leg, steps)
                # where a step's time can go at N > 1, in one place: the slowest rank's kernels, rank 0's host half, the
                # exchange - and the step itself; efficiency against the committed one-GPU rate of the same stream (profiles/n1_reference.json)
                per_step = lambda r: leg["calls_per_step"] * (r["scan"] + r["demod"])
                rr_facts = leg.get("rank_resolve") or {}
                n1 = n1_reference("frames_strong" if scaling == "strong" else None)
                d["scaling_breakdown"] = {"resolve_on": leg.get("resolve_on"), "n_gpus": world, "ms_per_step": d["ms_per_step"],
                                "kernel_ms_max_rank": round(max(per_step(r) for r in per_rank), 4),
                                "rank0_resolve_ms": rr_facts.get("work_ms_per_step", (leg.get("host_ms_per_call") or {}).get("resolve_per_step")),
                                "exchange_ms": rr_facts.get("exchange_ms_per_step", round(d["rccl"]["gather_ms"] * leg["calls_per_step"], 4)),
                                "Msamples_per_s": d["Msamples_per_s"],
                                "efficiency_vs_committed_n1": round(d["Msamples_per_s"] / (world * n1["Msamples_per_s"]), 4) if n1 else None,
                                "committed_n1": n1}
                # a leg with records to gather whose gather moved nothing did not measure the N > 1 path
                assert not (leg["lines"] > 0 and d["rccl"]["p2p_ops_per_step"] == 0 and (world > 1 or args.backend == "nccl")), \
                    "%s: %d messages per step but no point-to-point transfer was issued" % (name, leg["lines"])
        return d

    legs = {}
    if frames is not None:
        legs["frames"] = leg_summary(frames, "BASELINE.json configs[%d]: %d MiB per GPU, sigma=3 noise + DF11/DF17 frames (1 per 65,536 samples, "
                                     "10 %% with a flipped bit, seam offsets), --fix" % (2 if world == 1 else 3, args.frames_mib), "weak")
    if lowsnr is not None:
        legs["lowsnr"] = leg_summary(lowsnr, "BASELINE.json configs[4]: %d MiB per GPU, sigma=3 noise + weak frames (amplitude 8-15, 20-40 %% leak, "
                                     "5 %% two-bit errors, 1 per 16,384 samples), --aggressive" % args.lowsnr_mib, "weak")
    if strong is not None:
        sname = "BASELINE.json configs[3]'s stream (%d MiB, 1 frame per 65,536 samples) over %d GPU(s): the same stream at every N, --fix" % (
            args.frames_total_mib, world)
        if strong_is_frames:
            legs["frames_strong"] = dict(legs["frames"], scaling="strong", same_run_as="frames")
        else:
            legs["frames_strong"] = leg_summary(strong, sname, "strong")
        if strong.get("other_mode_error"):
            legs["frames_strong_other_mode_error"] = strong["other_mode_error"]
        if strong.get("other_mode") is not None:                # the same steps with the other resolve mode, behind the first pass
            o = strong["other_mode"]
            legs["frames_strong_resolve_on_%s" % o["resolve_on"]] = leg_summary(o, sname + " - resolve_on=%s" % o["resolve_on"], "strong")

    names = {"noise": noise, "frames": frames, "lowsnr": lowsnr, "strong": strong}
    head = noise if noise is not None else next(v for k, v in names.items() if v is not None)
    head_kind = next(k for k, v in names.items() if v is head)
    kern = noise1 if noise1 is not None else head              # the region whose kernel times are reported
    head_steps = head["steps"]
    head_name = ("%d MiB synthetic uint8 IQ @ 2 Msps format per GPU (sigma=3 integer noise, seed 20260922), --no-fix, "
                 "HBM-resident; BASELINE.json configs[1]" % args.mib) if noise is not None else legs[
                     "frames_strong" if head_kind == "strong" else head_kind]["workload"]
    samples_per_step = head["total"] // 2                                     # the whole stream: every rank's shard
    value = samples_per_step * head_steps / head["elapsed"] / 1e6
    region_s = head.get("elapsed_regions") or [head["elapsed"]]
    assert kern["timed_calls"] > 0 and kern["scan_ms"] > 0, "no call of the timed region carried timing events"
    achieved = kern["call_bytes"] / (kern["scan_ms"] * 1e-3) / 1e9             # this rank's launches: 2 B per sample
    traffic, traffic_note = measured_traffic(args.mib) if noise is not None else (None, "no PMC pass for this workload")
    committed_traffic = traffic
    if rank == 0 and world == 1 and noise is not None and not args.no_live_traffic:
        live = live_traffic(args.mib)
        if live is not None:
            traffic, traffic_note = live
    per_rank_kernels = gathered({"scan": round(kern["scan_ms"], 4), "demod": round(kern["demod_ms"], 4)})
    # the step in the lines it really moves: the scan's traffic (live or committed counter pass) + the demodulation kernel's (committed)
    demod_traffic = committed_demod_traffic(args.mib) if noise is not None else None
    step_traffic = traffic + demod_traffic if (traffic and demod_traffic and head["calls_per_step"] == 1) else None
    step_traffic_gbs = step_traffic / (head["elapsed"] / head_steps) / 1e9 if step_traffic else None
    line = {
        "metric": "IQ Msamples/s demodulated", "value": round(value, 1), "unit": "Msamples/s",
        "n_gpus": world, "steps": head_steps, "warmup": args.warmup,
        "ms_per_step": round(head["elapsed"] / head_steps * 1e3, 4), "higher_is_better": True,
        # `value` / `ms_per_step`: the MEDIAN of `regions` timed regions of exactly `steps` steps each (barrier + device sync on both
        # sides of every one, max over the ranks per region), back to back behind one warm-up; the others are here
        "regions": head.get("regions", 1),
        "ms_per_step_regions": [round(e / head_steps * 1e3, 4) for e in region_s],
        "value_min": round(samples_per_step * head_steps / max(region_s) / 1e6, 1),
        "value_max": round(samples_per_step * head_steps / min(region_s) / 1e6, 1),
        "value_spread_pct": round((max(region_s) - min(region_s)) / head["elapsed"] * 50, 2),      # +- half the range, in percent of the median
        "scaling": "strong" if head_kind == "strong" else "weak",
        "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": head_name, "bytes_per_gpu": head["per_gpu"],
                   "flags": {"noise": "--raw --no-fix", "lowsnr": "--raw --aggressive"}.get(head_kind, "--raw"),
                   "sharding": "buffers over %d rank(s)" % world, "settle_steps": args.settle if noise is not None else head.get("settle_steps"),
                   "demod_variant": args.demod_variant, "resolve_on": args.resolve_on if dist_on else None,
                   "step": "scan + demod + order kernels, record fetch%s, host resolve + --raw formatting on a second thread; "
                           "%d detect(s) in flight; overlap=%d; completion by a host-visible word (no event in the stream), kernel timing events on "
                           "one call in %d of the kernel-timing region only; %d launch stream(s) in the throughput region" % (
                               (", every rank resolves its own list (whitelist guesses and confirmations: three host all_gathers a step), the "
                                "texts travel to rank 0 over %s" if ranks_resolve else
                                ", device-resident lists gathered to rank 0 over %s (counts all_gather + exact-size send/recv)") % (
                                   "RCCL" if args.backend == "nccl" else args.backend) if world > 1 else "",
                               head["depth"], args.overlap, 4 if (noise is not None and args.streams > 1) else args.time_every,
                               max(1, args.streams))},
        "msgs_per_s": round(head.get("msgs", 0) / head["elapsed"], 2) if rank == 0 else None,
        "host_cpus": host_cpus(),                                  # what the host half (launch threads, resolver, its workers) had
        "host_ms_per_call": head.get("host_ms_per_call"),          # rank 0's host thread, by phase of the step loop
        "detect_us_per_call": head.get("detect_us_per_call"),      # inside modes_gpu_detect, by section (modes_gpu_host_profile)
        "region_tail_ms": head.get("region_tail_ms"),              # the end of the timed region: calls in flight, resolver, final sync
        "preambles_per_step_rank0": int(head["last"].get("n_preambles", 0)),
        "forwarded_per_step_rank0": int(head["last"].get("n_forwarded", 0)),
        "kernel_ms": {"scan": round(kern["scan_ms"], 4), "demod": round(kern["demod_ms"], 4), "order": round(kern["order_ms"], 4),
                      "scan_median": round(kern["scan_ms_median"], 4),
                      # the scan kernel's average in each of the kernel-timing regions (`scan` is over all of them)
                      "scan_regions": [round(v, 4) for v in kern.get("scan_ms_regions", []) if v > 0],
                      "timed_calls": kern["timed_calls"], "of_calls": kern.get("kernel_steps", kern["steps"]) * kern["calls_per_step"],
                      "measured_in": "a region of %d steps of the same workload on ONE launch stream, HIP events on one call in 4 (every "
                                     "kernel alone; averages over the timed calls)" % kern["steps"] if kern is not head
                      else kern.get("kernel_timing", "the timed region")},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "frac_regions": [round(kern["call_bytes"] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for v in kern.get("scan_ms_regions", []) if v > 0],
                     "traffic": traffic, "traffic_source": traffic_note,
                     "kernel": "scan_kernel", "algorithmic_bytes_per_launch": int(kern["call_bytes"]),
                     # the same bytes over the WHOLE step (every kernel, the fetch and the resolve behind it: the clock `value` is on)
                     "achieved_step": round(head["per_gpu"] / (head["elapsed"] / head_steps) / 1e9, 1),
                     "frac_step": round(head["per_gpu"] / (head["elapsed"] / head_steps) / 1e9 / HBM_PEAK_GBS, 4),
                     # ... and the HBM lines the step's kernels really read (scan + demodulation, FETCH_SIZE) over the same clock
                     "traffic_step": step_traffic,
                     "achieved_traffic_step": round(step_traffic_gbs, 1) if step_traffic_gbs else None,
                     "frac_traffic_step": round(step_traffic_gbs / HBM_PEAK_GBS, 4) if step_traffic_gbs else None,
                     # the box's own read-only streaming rate next to the specification
                     "measured_ceiling": ceiling,
                     "frac_traffic_step_of_measured_ceiling": round(step_traffic_gbs / ceiling["GB_per_s"], 4) if (step_traffic_gbs and ceiling) else None,
                     "frac_of_measured_ceiling": round(achieved / ceiling["GB_per_s"], 4) if ceiling else None,
                     # `achieved` is from the HIP events of THIS run; the committed rocprofv3 trace of the same sources
                     # (profiles/): its events read ~3 % above its own kernel durations (the dispatch's ~5 us lead-in)
                     "committed_trace_avg_ms": TRACE_AVG_MS, "committed_traffic": committed_traffic},
    }
    if rank == 0 and dist_on:
        line["rccl_start"] = dict(RCCL_START, backend=args.backend)
    if rank == 0 and world > 1:
        line["kernel_ms_per_rank"] = per_rank_kernels
    if rank == 0 and dist_on and noise is not None:
        line["rccl"] = comm_facts(noise, noise["steps"] * noise.get("regions", 1))     # the headline leg's lists are empty: the count exchange only
    if noise1 is not None and noise_s1 is not noise:
        line["one_launch_stream"] = {
            "Msamples_per_s": round(noise_s1["total"] // 2 * noise_s1["steps"] / noise_s1["elapsed"] / 1e6, 1),
            "ms_per_step": round(noise_s1["elapsed"] / noise_s1["steps"] * 1e3, 4),
            "what": "the same K steps with every call on ONE launch stream (scan, demod, finalize strictly in order, no events): "
                    "the kernel times plus the ~6 us in front of every scan kernel add up to this step"}
    if noise is not None:
        line.update(legs)
    elif rank == 0:
        only = legs["frames_strong" if head_kind == "strong" else head_kind]
        line["listing_check"] = only["listing_check"]
        if "rccl" in only:
            line["rccl"] = only["rccl"]
        if "rank_resolve" in only:
            line["rank_resolve"] = only["rank_resolve"]

    if rank == 0 and world == 1 and noise is not None and not args.no_end_to_end:
        line["end_to_end"] = end_to_end(torch, dev, iq_noise, args)
    if rank == 0 and world == 1 and noise is not None and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(iq_noise, min(args.cpu_mib << 20, noise["span"] // 262144 * 262144), "nofix", passes=4)
    if rank == 0:
        print(json.dumps(line), file=line_out, flush=True)
    if dist_on:
        dist.destroy_process_group()


def end_to_end(torch, dev, iq, args, batch_blocks=1024, passes=6):
    """The noise workload starting in pinned HOST memory: batches of `batch_blocks` buffers through
    modes_gpu_submit_host (H2D + kernels, asynchronous) in three rotating contexts, fetch + resolve in order - the
    loop of dump1090_amd/csrc/main.cpp without the file reads.  PCIe-inclusive; never the headline."""
    from dump1090_amd import Demodulator, HostResolver, block_count
    n = iq.numel()
    total_blocks = block_count(n)
    demods = [Demodulator(device=dev.index, fix=False) for _ in range(3)]
    batches = []
    for b0 in range(0, total_blocks, batch_blocks):
        nb = min(batch_blocks, total_blocks - b0)
        lo = max(0, b0 * 262144 - 476)
        hi = min(n, (b0 + nb) * 262144)
        buf = demods[0].host_alloc(hi - lo)
        buf[:] = iq[lo:hi].cpu().numpy()
        batches.append((buf, lo, b0, nb))
    res = HostResolver(fix=False)

    def one_pass():
        inflight = []
        for i, (buf, lo, b0, nb) in enumerate(batches):
            d = demods[i % 3]
            if len(inflight) == 3:
                x = inflight.pop(0)
                recs, _, _ = x.fetch(copy=False)
                res.raw_listing(recs, None)
            d.submit_host(buf, stream_byte0=lo, first_block=b0, nblocks=nb)
            inflight.append(d)
        for x in inflight:
            recs, _, _ = x.fetch(copy=False)
            res.raw_listing(recs, None)

    one_pass()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(passes):
        one_pass()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    for buf, *_ in batches:
        demods[0].host_free(buf)
    res.close()
    for d in demods:
        d.close()
    return {"Msamples_per_s": round(n / 2 * passes / dt / 1e6, 1), "GB_per_s": round(n * passes / dt / 1e9, 2),
            "what": "%d MiB of the headline workload in pinned host memory -> modes_gpu_submit_host in %d-buffer batches, 3 contexts "
                    "in rotation, fetch + resolve in order; %d passes, %.3f s (PCIe-inclusive; the C host adds the file reads: "
                    "tools/e2e_cli.py)" % (n >> 20, batch_blocks, passes, dt)}


if __name__ == "__main__":
    main()
