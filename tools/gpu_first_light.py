#!/usr/bin/env python3
"""Stage-by-stage bring-up on a GPU box: prints diffs instead of stopping at the first assert.
Not a test (tests/ has those); a debugging aid whose output goes to gpurun_out/."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np
import torch

import oracle as orc
import synth
from dump1090_amd import Demodulator, HostResolver, raw_text
from helpers import oracle_records

print("torch", torch.__version__, "cuda", torch.cuda.is_available(), torch.cuda.get_device_name(0))
props = torch.cuda.get_device_properties(0)
print("CUs", props.multi_processor_count, "mem GiB", props.total_memory / 2**30)


def stage(name):
    def deco(fn):
        t = time.time()
        try:
            fn()
            print("[ok  ] %-28s %.2fs" % (name, time.time() - t), flush=True)
        except Exception:
            print("[FAIL] %-28s" % name, flush=True)
            traceback.print_exc()
        return fn
    return deco


def forward_mask(iq):
    i = iq[0::2].astype(np.int64) - 127
    q = iq[1::2].astype(np.int64) - 127
    s = np.concatenate([i * i + q * q, np.zeros(24, dtype=np.int64)])
    n = iq.size // 2
    S = lambda k: s[k:k + n]
    ok = (S(0) > np.maximum.reduce([S(1), S(3), S(4), S(5), S(6)])) & (S(2) > np.maximum(S(1), S(3))) \
        & (S(7) > S(8)) & (S(9) > np.maximum(S(8), S(6)))
    sumh = (S(0) >> 2) + (S(2) >> 2) + (S(7) >> 2) + (S(9) >> 2)
    quiet = np.maximum.reduce([S(4), S(5), S(11), S(12), S(13), S(14)])
    return ok & (quiet <= ((sumh + 4) >> 1))


data = synth.modes1_padded(os.path.join(ROOT, "tests", "golden", "modes1.bin"))
iq = torch.from_numpy(data).cuda()


@stage("power tap")
def _():
    d = Demodulator()
    s = d.compute_power(iq).cpu().numpy()
    i = data[0::2].astype(np.int64) - 127
    q = data[1::2].astype(np.int64) - 127
    bad = np.flatnonzero(s != (i * i + q * q))
    print("   mismatches:", bad.size, bad[:10])
    assert bad.size == 0


@stage("magnitude")
def _():
    d = Demodulator()
    m = d.compute_magnitude_vector(iq).cpu().numpy()
    bad = np.flatnonzero(m != orc.magnitude(data))
    print("   mismatches:", bad.size, bad[:10])
    assert bad.size == 0


@stage("synth noise")
def _():
    d = Demodulator()
    out = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")
    d.synth_noise(out, 100, 5, 941)
    want = synth.noise_bytes(5, 100, 1 << 16, 941)
    bad = np.flatnonzero(out.cpu().numpy() != want)
    print("   mismatches:", bad.size, bad[:10], out[:8].cpu().numpy(), want[:8])
    assert bad.size == 0


@stage("candidates modes1")
def _():
    for rc in (0, 1, 64):
        d = Demodulator(keep_candidates=True, run_chunks=rc)
        d.detect(iq)
        recs, cands, info = d.fetch()
        want, want_c = oracle_records(data, 1)
        fm = forward_mask(data)
        g = np.arange(fm.size) + 238
        valid = (g % 131072) < 131070
        print("   run_chunks", rc, info, "numpy forwarded:", int((fm & valid).sum()), "oracle preambles:", want_c.size,
              "oracle records:", want.size)
        missing = np.setdiff1d(want_c, cands)
        extra = np.setdiff1d(cands, want_c)
        print("   missing", missing.size, missing[:10], "extra", extra.size, extra[:10])
        if cands.size:
            print("   sorted:", bool(np.all(np.diff(cands.astype(np.int64)) > 0)))
        ok = recs.size == want.size and np.array_equal(recs["block"], want["block"]) and np.array_equal(recs["j"], want["j"])
        print("   record positions equal:", ok)
        if ok:
            for a in (0, 1):
                for f in ("msg", "errors", "gate_ok", "nfix", "fixpos", "syndrome"):
                    x, y = recs["att"][f][:, a], want["att"][f][:, a]
                    if a == 1:
                        live = want["att"]["gate_ok"][:, 1] == 1
                        x, y = x[live], y[live]
                    neq = np.flatnonzero((x != y).reshape(len(x), -1).any(axis=1))
                    if neq.size:
                        print("   att", a, f, "differs at", neq.size, "records, first:", neq[:5], x[neq[0]], y[neq[0]])
        d.close()


@stage("listing modes1 (all flag sets)")
def _():
    for fs, flags in orc.FLAGSETS.items():
        d = Demodulator(keep_candidates=True, **flags)
        msgs = d.demodulate(iq)
        want, st = orc.run_stream(data, **flags)
        same = raw_text(msgs) == orc.raw_text(want)
        print("   %-18s %4d lines, oracle %4d, equal=%s stats_equal=%s" % (
            fs, len(msgs), len(want), same, d.last["stats_text"] == orc.stats_text(st)))
        d.close()


@stage("timing 1 GiB noise")
def _():
    n = 1 << 30
    big = torch.empty(n, dtype=torch.uint8, device="cuda")
    d0 = Demodulator(fix=False)
    d0.synth_noise(big, 0, 20260922, 941)
    d0.fill(big[-480:], 127)
    torch.cuda.synchronize()
    d0.close()
    for rc in (0, 8, 16, 32, 64, 128, 256):
        d = Demodulator(fix=False, run_chunks=rc)
        ts, td = [], []
        for it in range(6):
            d.detect(big)
            recs, _, info = d.fetch()
            if it >= 2:
                ts.append(info["scan_ms"])
                td.append(info["demod_ms"])
        print("   run_chunks %3d: scan %.3f ms (%.0f GB/s, %.1f%% of 8 TB/s)  demod+finalize %.3f ms  fwd %d pre %d rec %d" % (
            rc, np.mean(ts), n / np.mean(ts) / 1e6, 100 * n / np.mean(ts) / 1e6 / 8000, np.mean(td),
            info["n_forwarded"], info["n_preambles"], info["n_records"]), flush=True)
        d.close()
    # a plain device-to-device copy of the same buffer as the achievable-bandwidth yardstick
    dst = torch.empty_like(big)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        dst.copy_(big)
    e0.record()
    for _ in range(5):
        dst.copy_(big)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("   torch copy 1 GiB: %.3f ms -> read %.0f GB/s (+ same written)" % (ms, n / ms / 1e6))
    # read-only yardstick: sum reduction
    for _ in range(2):
        big.view(torch.int64).sum()
    e0.record()
    for _ in range(5):
        big.view(torch.int64).sum()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("   torch int64 sum 1 GiB: %.3f ms -> read %.0f GB/s" % (ms, n / ms / 1e6))
