#!/usr/bin/env python3
"""bench.py - IQ Msamples/s demodulated on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of HBM-resident synthetic input:
scan kernel + demod kernel + record fetch (+ gather of records to rank 0 when N > 1) + the
sequential host resolve on rank 0.  Workload = BASELINE.json configs[1]: 1 GiB of synthetic
uint8 I/Q in 2 Msps file format (sigma = 3 integer noise, tests/synth.py), --no-fix, PER GPU
(weak scaling: the stream is sharded into per-GPU buffer ranges, each rank's shard is 1 GiB).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Before the W warmup steps the bench runs `--settle` (80) more untimed steps: the chip's power management
needs ~40 back-to-back steps to reach its sustained clocks (tools/scan_steps.py: the scan kernel takes 0.22,
0.27 and 0.21 ms at steps 1, 10 and 60 of an uninterrupted run); the K timed steps are therefore the
sustained rate, which is what a stream of many batches sees.

Prints ONE JSON line on rank 0.  `roofline` is for the scan kernel (the only stage that reads
every sample): algorithmic bytes = 2 per sample, duration = HIP events recorded around the kernel
on its launch stream inside libmodes_gfx950.so.  `cpu_baseline` (N == 1 only) times the compiled
reference (oracle/_ref) - or the C restatement if that binary is absent - on the host.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def measured_traffic(mib):
    """HBM bytes per scan launch from the committed rocprofv3 PMC passes (profiles/traffic_latest.json,
    written by tools/profile.sh on this same default workload); None for any other workload."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if mib != 1024 or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return int(json.load(f)["scan_kernel"]["hbm_read_bytes_per_launch"])
    except (KeyError, ValueError):
        return None


def cpu_baseline(iq, nbytes_sample):
    """Reference single-threaded C path on the host cores, on the first nbytes_sample bytes of the
    same workload.  Checker code (oracle/) is used here only as the thing being timed."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    sample = iq[:nbytes_sample].cpu().numpy()
    sample[-480:] = 127
    nsamp = nbytes_sample // 2
    passes = 4                                           # ~10 s of single-threaded CPU work on the 1 GiB workload
    if orc.have_ref():
        with tempfile.NamedTemporaryFile(suffix=".bin", dir="/tmp") as f:
            sample.tofile(f.name)
            env = dict(os.environ, LD_PRELOAD=orc.FIXED_TIME)
            dt, out = 0.0, b""
            for _ in range(passes):
                t0 = time.perf_counter()
                out = subprocess.run([orc.REF_BIN, "--ifile", f.name, "--raw", "--no-fix"], stdout=subprocess.PIPE,
                                     env=env, check=True).stdout
                dt += time.perf_counter() - t0
        kind, lines = "reference", out.count(b"\n")
        what = "oracle/_ref/dump1090_ref --ifile <first %d MiB of the workload> --raw --no-fix, %d passes" % (
            nbytes_sample >> 20, passes)
        nsamp *= passes
    else:
        t0 = time.perf_counter()
        msgs, _ = orc.run_stream(sample, **orc.FLAGSETS["nofix"])
        dt = time.perf_counter() - t0
        kind, lines = "port", len(msgs)
        what = "oracle/liboracle.so orc_run_stream on the first %d MiB of the workload, --no-fix" % (nbytes_sample >> 20)
    return {"value": round(nsamp / dt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": kind,
            "sample": "%s; %.2f s wall, %d messages; 1 decode thread (host has %d cores)" % (
                what, dt, lines, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mib", type=int, default=1024, help="MiB of I/Q per GPU (default: the 1 GiB workload)")
    ap.add_argument("--cpu-mib", type=int, default=1024, help="MiB of the workload timed on the CPU baseline")
    ap.add_argument("--run-chunks", type=int, default=0)
    ap.add_argument("--scan-variant", type=int, default=0, help="0 = production scan kernel, 1 = fused single-pass scan")
    ap.add_argument("--depth", type=int, default=0,
                    help="detect calls in flight (contexts used alternately); default 2, 3 with several ranks "
                         "(the record gather adds host time per step: one more step of slack)")
    ap.add_argument("--settle", type=int, default=80,
                    help="extra untimed steps before the W warmup steps: the chip's power management needs ~40 "
                         "back-to-back steps (13 ms) to settle - the scan kernel runs 0.22, 0.27, 0.21 ms at steps "
                         "1, 10, 60 of a sustained run (tools/scan_steps.py)")
    ap.add_argument("--overlap", type=int, default=0,
                    help="1: a step's demod kernel runs on a second stream, concurrent with the next step's scan "
                         "(measured: +3 %% value, but the scan kernel then shares the chip: -10 %% on its own time)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the detects are spread over (1: all kernels in order on one stream; "
                         "2: the two contexts' kernels may overlap at their edges)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to smoke-test "
                                                      "the N > 1 control flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dump1090_amd import Demodulator, HostResolver, block_count, shard_blocks, shard_byte_range
    from dump1090_amd.distributed import gather_records

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (
                args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")     # where the gathered bytes travel
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # the global stream = world x (--mib) MiB; this rank demodulates its contiguous buffer range and
    # holds exactly the bytes that range needs (its buffers + the 476-byte carry in front).
    per_gpu = args.mib << 20
    total = per_gpu * world
    nblocks_total = block_count(total)
    first_block, nblocks = shard_blocks(nblocks_total - 1, world, rank)     # the EOF buffer goes to the last rank
    if rank == world - 1:
        nblocks += 1
    lo, hi = shard_byte_range(first_block, nblocks, total)

    # Two contexts, used alternately: step i+1's kernels are queued before step i's records are fetched
    # and resolved, so the GPU never waits for the host (the C host double-buffers the same way).
    # Every step still does all of its work; K steps = K detects + K fetches + K resolves.
    demods = [Demodulator(device=local, fix=False, run_chunks=args.run_chunks, scan_variant=args.scan_variant,
                          overlap=bool(args.overlap)) for _ in range(args.depth if args.depth > 0 else (2 if world == 1 else 3))]
    demod = demods[0]
    iq = torch.empty(hi - lo, dtype=torch.uint8, device=dev)
    demod.synth_noise(iq, first_byte=lo, seed=20260922, sigma_q16=941)
    if hi == total:
        demod.fill(iq[-480:], 127)                  # oracle-safe tail
    torch.cuda.synchronize(dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    scan_ms, demod_ms, n_pre, n_fwd = [], [], 0, 0
    n_msgs = [0]
    # the kernels go to their own stream, so that the (tiny) RCCL size exchange of step i, issued on
    # torch's current stream, does not queue behind the kernels of step i+1
    works = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    def finish(d, timed):
        """fetch + (gather) + sequential host resolve of the detect in flight on context d"""
        nonlocal n_pre, n_fwd
        recs, cands, info = d.fetch()
        if world > 1:
            recs, cands = gather_records(recs, cands, dst=0, device=coll_dev)
        if rank == 0:
            res = HostResolver(fix=False)
            m = res.count(recs, cands)
            res.close()
            if timed:
                n_msgs[0] += m
        if timed:
            scan_ms.append(info["scan_ms"])
            demod_ms.append(info["demod_ms"])
        n_pre, n_fwd = info["n_preambles"], info.get("n_forwarded", 0)

    in_flight = []                                  # contexts with a detect queued, oldest first
    t0 = None
    warm = args.settle + args.warmup
    for step in range(warm + args.steps):
        if step == warm:
            while in_flight:
                finish(in_flight.pop(0), False)
            sync_all()
            t0 = time.perf_counter()
        d = demods[step % len(demods)]
        if d in in_flight:                          # its previous detect must be fetched first
            while in_flight:
                x = in_flight.pop(0)
                finish(x, step > warm)
                if x is d:
                    break
        d.detect(iq, stream_byte0=lo, first_block=first_block, nblocks=nblocks, stream=works[step % len(works)])
        in_flight.append(d)
    while in_flight:
        finish(in_flight.pop(0), True)
    sync_all()
    elapsed = time.perf_counter() - t0
    n_msgs = n_msgs[0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_per_step = total // 2
    value = samples_per_step * args.steps / elapsed / 1e6
    scan_avg_ms = float(np.mean(scan_ms))
    achieved = (2.0 * (hi - lo) / 2) / (scan_avg_ms * 1e-3) / 1e9       # this rank's launch: 2 B per sample
    line = {
        "metric": "IQ Msamples/s demodulated", "value": round(value, 1), "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": "%d MiB synthetic uint8 IQ @ 2 Msps format per GPU (sigma=3 integer noise, seed "
                               "20260922), --no-fix, HBM-resident; BASELINE.json configs[1]" % args.mib,
                   "bytes_per_gpu": per_gpu, "flags": "--raw --no-fix", "sharding": "buffers over %d rank(s)" % world,
                   "settle_steps": args.settle,
                   "step": "scan + demod kernels, record fetch%s, host resolve; %d detect(s) in flight" % (
                       ", RCCL gather to rank 0" if world > 1 else "", len(demods))},
        "msgs_per_s": round(n_msgs / elapsed, 2),
        "preambles_per_step_rank0": int(n_pre), "forwarded_per_step_rank0": int(n_fwd),
        "kernel_ms": {"scan": round(scan_avg_ms, 4), "demod_finalize": round(float(np.mean(demod_ms)), 4)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args.mib),
                     "kernel": "scan_kernel", "algorithmic_bytes_per_launch": int(hi - lo)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(iq, min(args.cpu_mib << 20, (hi - lo) // 262144 * 262144))
    if rank == 0:
        print(json.dumps(line), flush=True)
    for d in demods:
        d.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
